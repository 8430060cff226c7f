#!/usr/bin/env python3
"""Headline benchmark: output megapixels/s of the 8x EMSR stage-3 forward
(5-slice 1x128x128 LR windows -> 1024x1024 HR tiles) on 1/2/4/8 MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env), or started
plainly -- then this process only PARSES the arguments and starts N fresh rank processes of itself (one per GPU,
env:// rendezvous on 127.0.0.1: the one-process-per-GPU launch of R:train_stage3.py:20-27); the parent never touches
the GPU, forwards rank 0's JSON line and exits non-zero if any rank fails.

A "step" is one pass of the hot path over one batch of synthetic tiles: at N=1 the workload is BASELINE.json configs[1]
(batch = 16 tiles, fp32, forward only); at N>1 every rank runs the same 16 tiles/GPU (weak scaling) and the step ends
with the RCCL all-gather of the HR output slabs (the north_star's only exchange step).  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line with the `roofline` (dominant kernel family: algorithmic FLOPs /
HIP-event time on the launch stream, against the matrix-pipe peak of the dtype) and `cpu_baseline` (the CPU oracle timed
on the host cores) objects.  `extras` carries the bf16 configurations (BASELINE configs[2]) measured in the same run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MATRIX_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak
ESSENTIAL_GFLOP_PER_TILE = {8: 5901.7, 16: 4882.3}
CHILD_ENV = "GPEMSR_BENCH_CHILD"


def effective_cores() -> int:
    """Host cores this process may actually use: min(affinity, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tiles", type=int, default=16, help="tiles (5-slice windows) per GPU per step")
    ap.add_argument("--lr", type=int, default=128)
    ap.add_argument("--scale", type=int, default=8, choices=(8, 16))
    ap.add_argument("--precision", type=str, default="fp32", choices=("fp32", "bf16x3", "bf16", "bf16op"),
                    help="fp32 = exact fp32 MFMA (BASELINE configs[1], the default); bf16 = bf16 activations in HBM + bf16 MFMA "
                         "(BASELINE configs[2]); bf16x3 = fp32 activations, convs on the bf16 pipe with split hi+lo operands "
                         "(fp32-grade); bf16op = fp32 activations, bf16 operands (round 1's bf16 mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra (untimed-by-the-driver) measurements")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the BASELINE configs[3] / configs[4] legs of the default line (x16 forward in both precisions, training step)")
    ap.add_argument("--extras", type=str, default="bf16", help="comma list of precisions measured beside the official fp32 number")
    ap.add_argument("--cpu-lr", type=int, default=128, help="LR size of the CPU-oracle sample tile")
    ap.add_argument("--layer-report", type=str, default="", help="write a per-layer conv timing table to this file")
    ap.add_argument("--mode", type=str, default="forward", choices=("forward", "train", "train2", "train1"),
                    help="forward = BASELINE configs[1]/[2]/[3] (the headline metric, default); train = configs[4], the stage-3 "
                         "training step; train2 = the stage-2 (indexer) training step")
    ap.add_argument("--no-profile", action="store_true", help="--mode train: no per-launch HIP events (roofline fields become 0)")
    ap.add_argument("--train-batch", type=int, default=8, help="--mode train: samples per GPU per step")
    ap.add_argument("--stage2-lr", type=int, default=128, help="--mode train2: LR size (GT is scale x larger)")
    ap.add_argument("--stage1-size", type=int, default=512, help="--mode train1: crop size (option/train_stage1.yml GT_size)")
    ap.add_argument("--train-lr", type=int, default=32, help="--mode train: LR crop size (option/train_stage3_x8.yml LQ_size)")
    ap.add_argument("--backend", type=str, default="", help="torch.distributed backend (default: nccl = RCCL on GPUs)")
    ap.add_argument("--stub", action="store_true",
                    help="launcher self-test (tests/test_bench_launcher_cpu.py): a per-tile stand-in model on the CPU with "
                         "--backend gloo; exercises spawn, rendezvous, barrier, all-gather and max-over-ranks timing only")
    ap.add_argument("--rank-timeout", type=float, default=3000.0, help="seconds the launcher waits for its ranks")
    ap.add_argument("--detail", type=str, default="bench_detail.json",
                    help="file that receives the FULL measurement document (per-kernel tables, full legs, extras, prose); the ONE stdout line stays compact")
    ap.add_argument("--gather", type=str, default="f32", choices=("f32", "u8"),
                    help="what the step's all-gather exchanges: the fp32 HR slabs (default; 4 MiB per 1024^2 tile, the north_star's slabs) or the "
                         "8-bit image the last kernel writes (1 MiB per tile: what output_GPEMSR.py saves)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own N ranks
# ----------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args, argv) -> int:
    """Start args.gpus rank processes of this script (fresh interpreters; this parent initialises no GPU state), wait,
    forward rank 0's stdout.  Returns the exit code: 0 only if every rank exited 0 and rank 0 printed its JSON line."""
    import socket
    import subprocess
    import tempfile
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ)
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    base[CHILD_ENV] = "1"
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")                 # rank 0's stdout (a file, so a long line can never block it)
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + args.rank_timeout
    rc = 0
    try:
        pending = set(range(n))
        while pending and rc == 0:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0:
                        print(f"[bench launcher] rank {r} exited with code {code}", file=sys.stderr, flush=True)
                        rc = code if code > 0 else 1
            if pending and rc == 0:
                if time.time() > deadline:
                    print(f"[bench launcher] ranks {sorted(pending)} still running after {args.rank_timeout:.0f} s", file=sys.stderr, flush=True)
                    rc = 124
                else:
                    time.sleep(0.2)
    finally:
        for p in procs:                      # the exact PIDs started above, never a pattern
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:
                p.kill()
    out0.seek(0)
    text = out0.read()
    out0.close()
    json_line = None
    for ln in text.splitlines():
        if ln.startswith("{"):
            try:
                if json.loads(ln).get("n_gpus") == n:
                    json_line = ln
            except Exception:
                pass
        elif ln.strip():
            print(ln, file=sys.stderr)          # library chatter on rank 0's stdout (e.g. gloo's connection notice): not part of the ONE line
    if rc == 0 and json_line is None:
        print("[bench launcher] rank 0 printed no JSON line for n_gpus=%d" % n, file=sys.stderr, flush=True)
        rc = 1
    if json_line is not None:
        sys.stdout.write(json_line + "\n")
    sys.stdout.flush()
    return rc


# ----------------------------------------------------------------------------------------------------------------------
# rank body
# ----------------------------------------------------------------------------------------------------------------------
class _StubModel:
    """Per-tile stand-in (launcher self-test on the CPU): out depends on the tile only, so sharded == unsharded."""
    precision = "stub"

    def __call__(self, x, want_u8=False):
        b = x.shape[0]
        out = (x[:, 2].mean(dim=(1, 2, 3)).view(b, 1, 1, 1) + x[:, 2, :, :8, :8]).contiguous()
        if want_u8:
            return out, x, (out[:, 0].clamp(0, 1) * 255.0).round().to(__import__("torch").uint8)
        return out, x


def timed_steps(step, steps: int, world: int, dev, sync):
    """EXACTLY `steps` steps between barrier + synchronize pairs; returns (max over ranks of the wall time, last output)."""
    import torch
    if world > 1:
        torch.distributed.barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    sync()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, out


PROFILED_STEPS = 3                  # steps of the second, event-carrying pass the kernel tables come from


def measure_leg(step, args, world, dev, sync, stub=False):
    """The OFFICIAL figure of a leg: exactly args.steps steps between barrier + synchronize pairs with NO per-launch events.  The kernel
    tables (roofline, per-kernel rows, layer report) come from a SECOND pass of min(steps, PROFILED_STEPS) steps with HIP events around every
    matrix-kernel launch (ops.LaunchProfiler; events on the launch stream).  Returns (dt, last output, profiler or None, profiled-pass seconds,
    profiled steps)."""
    dt, out = timed_steps(step, args.steps, world, dev, sync)
    if stub:
        return dt, out, None, dt, args.steps
    from gpemsr_amd import ops
    psteps = max(1, min(args.steps, PROFILED_STEPS))
    prof = ops.LaunchProfiler()
    ops.PROFILER = prof
    try:
        dtp, _ = timed_steps(step, psteps, world, dev, sync)
    finally:
        ops.PROFILER = None
    return dt, out, prof, dtp, psteps


def _family(summ, names):
    d = {"launches": 0, "ms": 0.0, "flops": 0.0, "executed": 0.0}
    for k in names:
        if k in summ:
            for f in d:
                d[f] += summ[k][f]
    return d


def _tf(d):
    return d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0


PMC_ROUNDS = ("r06", "r05")         # profiles/<round>_<precision>_pmc_summary.json, newest first: the counter summary taken on THIS tree


def _pmc_doc(tag: str):
    for rnd in PMC_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{tag}_pmc_summary.json")
        try:
            return json.load(open(path)), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def pmc_traffic(tag: str, kernel_name: str, launches_per_step: float, fam_launches_per_step: int):
    """HBM bytes per launch from the rocprofv3 --pmc summary committed for THIS tree (scripts/pmc_round2.sh; separate passes, FETCH_SIZE x2
    gfx950 correction): of the ONE dominant kernel (matched by base name + launches per step) and of the family.  A launch-count mismatch
    means the summary was taken on another tree -> null."""
    doc, path = _pmc_doc(tag)
    if doc is None:
        return None, None, None
    src = {"file": path, "hbm_bytes_per_step_all_kernels": doc.get("hbm_bytes_per_step_all_kernels")}
    fam = doc.get("dominant_family", {})
    fam_traffic = round(fam["hbm_bytes_per_launch"]) if int(fam.get("launches_per_step", -1)) == int(fam_launches_per_step) else None
    if fam_traffic is not None:
        src["family_mfma_util_pmc"] = round(fam.get("mfma_util", 0.0), 4)
    base = kernel_name.split("<")[0]
    k_traffic = None
    if kernel_name.endswith("<*>"):       # every instantiation of one kernel template (epilogue flavours of one main loop): counters summed
        ds = [d for nm, d in doc.get("per_kernel", {}).items() if nm.split("<")[0] == base]
        if ds and abs(sum(d.get("dispatches_per_step", 0) for d in ds) - launches_per_step) < 0.01:
            k_traffic = round(1e9 * sum(d["hbm_read_GB_per_step"] + d["hbm_write_GB_per_step"] for d in ds) / max(launches_per_step, 1e-9))
            src["kernel_counter_name"] = base + "<*> (%d instantiations)" % len(ds)
            gui = sum(d.get("gui_active", 0.0) for d in ds)
            if gui:
                src["kernel_mfma_util_pmc"] = round((sum(d.get("mfma_busy_cycles", 0.0) for d in ds) / 1024.0) / (gui / 8.0), 4)
        return k_traffic, fam_traffic, src
    for nm, d in doc.get("per_kernel", {}).items():
        if nm.split("<")[0] == base and abs(d.get("dispatches_per_step", -1) - launches_per_step) < 0.01:
            k_traffic = round(1e9 * (d["hbm_read_GB_per_step"] + d["hbm_write_GB_per_step"]) / max(launches_per_step, 1e-9))
            src["kernel_counter_name"] = nm
            # SQ_VALU_MFMA_BUSY_CYCLES sums over 4 SIMDs x 256 CUs; GRBM_GUI_ACTIVE over 8 XCDs
            if d.get("gui_active"):
                src["kernel_mfma_util_pmc"] = round((d.get("mfma_busy_cycles", 0.0) / 1024.0) / (d["gui_active"] / 8.0), 4)
            break
    return k_traffic, fam_traffic, src


FLAVOURED = ("conv_wino4_f32_kernel",)     # kernel templates whose parameters only select the epilogue

FAMILIES = {
    "fp32": (("conv_mfma",), "fp32 MFMA family (v_mfma_f32_32x32x2_f32): conv_wino4_f32_kernel (3x3 stride-1 layers in the Winograd F(4x4,3x3) form: 1/4 of the "
             "multiplies, fp32 arithmetic; its template parameter selects the epilogue), conv_wino2*_f32_kernel (F(2x2,3x3): 16/36), conv7_wino2d_f32_kernel "
             "(7x7 in F(2x2,7x7): 64/196), conv_mfma_kernel (implicit-GEMM conv / GEMM, direct form); per kernel: `kernels`", PEAK_F32_MATRIX_TFLOPS),
    "bf16": (("conv_bf16", "vgg_mask"),
             "bf16 MFMA family (v_mfma_f32_32x32x16_bf16 / 16x16x32): conv_bf16_kernel / conv64_resident2_kernel / convt64_resident_kernel (implicit-GEMM "
             "conv / 1x1 / transposed), conv7_c32_cout16_kernel + conv7_c8_cout32_kernel (SpyNet 7x7), flash_attn512_kernel (q.k^T + online softmax + P.v "
             "of the NonLocalBlock), vgg_mask2_kernel (fused VGG relu1_2 + 16x16 patch cosine); per kernel: `kernels`",
             PEAK_BF16_MATRIX_TFLOPS),
    "bf16x3": (("conv_split",), "conv_split_kernel (v_mfma_f32_32x32x16_bf16, 3 split products per algorithmic product)", PEAK_BF16_MATRIX_TFLOPS),
    "bf16op": (("conv_split",), "conv_split_kernel (v_mfma_f32_32x32x16_bf16, bf16 operands rounded in LDS)", PEAK_BF16_MATRIX_TFLOPS),
}


def _etf(d):
    """EXECUTED TFLOP/s: the multiplies the matrix pipe really performed / the HIP-event time."""
    return d["executed"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0


def build_roofline(args, prof, dtp, psteps, B, s, precision):
    """`roofline` = the ONE dominant kernel (most HIP-event time in the profiled pass): `achieved` / `frac` are EXECUTED matrix FLOPs per launch
    / its average launch duration / the dtype's dense matrix peak -- always <= 1.  Where a kernel removes arithmetic (Winograd F(2x2,3x3):
    16 of every 36 multiplies of the direct form), the ALGORITHMIC rate (direct-convolution FLOP count of SURVEY 8(d) / time) is stated
    beside it as `effective_tflops` / `effective_frac` and may exceed the peak.  `family` = the same figures over every launch of the matrix
    kernel family; `kernels` = the top instantiations one by one."""
    summ = prof.summary()
    by_name = prof.summary(by_name=True)
    fam_names, fam_desc, peak = FAMILIES[precision]
    fam = _family(summ, fam_names)
    steps = max(psteps, 1)
    fam_lps = fam["launches"] // steps
    # the dominant kernel: most time among the family's instantiations.  Instantiations of a template in FLAVOURED differ only in their epilogue
    # (store / residual / PixelShuffle / GroupNorm sums / patch cosine around ONE main loop): they count as one kernel, `kernel` = "<name><*>"
    cand = {}
    for k, v in by_name.items():
        if v["family"] not in fam_names:
            continue
        base = k.split("<")[0]
        key = base + "<*>" if base in FLAVOURED else k
        if key in cand:
            for f in ("launches", "ms", "flops", "bytes", "executed"):
                cand[key][f] += v[f]
            cand[key]["instantiations"].append(k)
        else:
            cand[key] = {**{f: v[f] for f in ("launches", "ms", "flops", "bytes", "executed")}, "family": v["family"], "instantiations": [k]}
    dom_name, dom = max(cand.items(), key=lambda kv: kv[1]["ms"], default=("?", None))
    if dom is None:
        dom = {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "executed": 0.0}
    dom_lps = dom["launches"] / steps
    k_traffic, fam_traffic, traffic_src = pmc_traffic(precision, dom_name, dom_lps, fam_lps)
    nl = max(dom["launches"], 1)
    r = {
        "bound": "mfma", "kernel": dom_name, "kernel_instantiations": sorted(dom.get("instantiations", [dom_name])),
        "achieved": round(_etf(dom), 2), "peak": peak, "unit": "TFLOP/s", "frac": round(_etf(dom) / peak, 4),
        "achieved_is": "EXECUTED matrix FLOPs per launch / average launch duration (HIP events on the launch stream, second pass of "
                       f"{steps} steps; the official `value` is timed in a pass without events)",
        "traffic": k_traffic, "traffic_unit": "HBM bytes per launch (PMC: FETCH_SIZE x 2 + WRITE_SIZE)", "traffic_source": traffic_src,
        "avg_launch_us": round(1e3 * dom["ms"] / nl, 2), "launches_per_step": round(dom_lps, 2),
        "executed_gflop_per_launch": round(dom["executed"] / 1e9 / nl, 2),
        "algorithmic_gflop_per_launch": round(dom["flops"] / 1e9 / nl, 2),
        "algorithmic_bytes_per_launch": round(dom["bytes"] / nl),
        "effective_tflops": round(_tf(dom), 2), "effective_frac": round(_tf(dom) / peak, 4),
        "executed_over_algorithmic_flops": round(dom["executed"] / dom["flops"], 4) if dom["flops"] > 0 else 1.0,
        "time_share_of_step": round(dom["ms"] * 1e-3 / dtp, 3),
        "family": {
            "kernels": fam_desc,
            "achieved": round(_etf(fam), 2), "frac": round(_etf(fam) / peak, 4),
            "effective_tflops": round(_tf(fam), 2), "effective_frac": round(_tf(fam) / peak, 4),
            "executed_over_algorithmic_flops": round(fam["executed"] / fam["flops"], 4) if fam["flops"] > 0 else 1.0,
            "launches_per_step": fam_lps, "avg_launch_us": round(1e3 * fam["ms"] / max(fam["launches"], 1), 2),
            "algorithmic_gflop_per_launch": round(fam["flops"] / 1e9 / max(fam["launches"], 1), 2),
            "traffic": fam_traffic,
            "algorithmic_gflop_per_tile_in_kernel": round(fam["flops"] / 1e9 / max(steps * B, 1), 1),
            "essential_gflop_per_tile_survey": ESSENTIAL_GFLOP_PER_TILE[s],
            "time_share_of_step": round(fam["ms"] * 1e-3 / dtp, 3),
        },
        "whole_path_tflops_essential": round(ESSENTIAL_GFLOP_PER_TILE[s] * B * steps / dtp / 1e3, 2),
        "profiled_pass_ms_per_step": round(1e3 * dtp / steps, 2),
    }
    others = {k: {"tflops": round(_tf(v), 2), "launches_per_step": v["launches"] // steps,
                  "time_share_of_step": round(v["ms"] * 1e-3 / dtp, 3)} for k, v in summ.items() if k not in fam_names}
    if others:
        r["other_profiled_kernels"] = others
    # what a reader needs to audit the figures: algorithmic bytes beside the counter bytes, and the kernels one by one
    alg_bytes = sum(v["bytes"] for v in summ.values()) / steps
    r["algorithmic_bytes_per_step"] = round(alg_bytes)
    r["algorithmic_bytes_note"] = ("sum over every profiled launch (all convolution / product / fused-mask layers) of its unique inputs + outputs + weights "
                                   "at the dtypes stored; element-wise passes between layers (GroupNorm apply, resamplers, gathers) are not in it")
    all_b = (traffic_src or {}).get("hbm_bytes_per_step_all_kernels")
    r["counter_bytes_per_step_all_kernels"] = all_b
    r["counter_over_algorithmic_bytes"] = round(all_b / alg_bytes, 2) if (all_b and alg_bytes) else None
    r["kernels"] = kernel_table(prof, steps, peak)
    return r


def kernel_table(prof, steps: int, peak_tflops: float, top: int = 8):
    """The `top` kernel instantiations of a leg by time: name, launches/step, ms/step, EXECUTED TFLOP/s and its fraction of the leg's matrix
    peak (`tflops`, `frac`), the algorithmic rate beside it (`effective_tflops`), algorithmic GB/step and GB/s (HIP events on the launch
    stream, same data as `roofline`)."""
    rows = sorted(prof.summary(by_name=True).items(), key=lambda kv: -kv[1]["ms"])[:top]
    out = []
    for name, d in rows:
        out.append({"name": name, "family": d["family"], "launches_per_step": round(d["launches"] / steps, 2), "ms_per_step": round(d["ms"] / steps, 3),
                    "executed_gflop_per_step": round(d["executed"] / 1e9 / steps, 1), "tflops": round(_etf(d), 1), "frac": round(_etf(d) / peak_tflops, 4),
                    "algorithmic_gflop_per_step": round(d["flops"] / 1e9 / steps, 1), "effective_tflops": round(_tf(d), 1),
                    "algorithmic_gb_per_step": round(d["bytes"] / 1e9 / steps, 3),
                    "algorithmic_gbps": round(d["bytes"] / 1e9 / (d["ms"] * 1e-3), 1) if d["ms"] > 0 else 0.0})
    return out


def write_layer_report(prof, path):
    rows = sorted(prof.summary(by_tag=True).items(), key=lambda kv: -kv[1]["ms"])
    with open(path, "w") as f:
        f.write("kernel\ttag\tlaunches\tms_total\tGFLOP\tTFLOP/s\tinstantiation\talgorithmic_GB\tGB/s\n")
        for (kern, tag), d in rows:
            gbps = d["bytes"] / 1e9 / (d["ms"] * 1e-3) if d["ms"] > 0 else 0.0
            f.write(f"{kern}\t{tag}\t{d['launches']}\t{d['ms']:.3f}\t{d['flops'] / 1e9:.1f}\t{_tf(d):.1f}\t{d['name']}\t{d['bytes'] / 1e9:.3f}\t{gbps:.0f}\n")


def _volume_bench(model, fr, win, steps):
    import torch
    model.forward_volume(fr, win)
    torch.cuda.synchronize()
    tv = time.perf_counter()
    for _ in range(steps):
        model.forward_volume(fr, win)
    torch.cuda.synchronize()
    return (time.perf_counter() - tv) / steps


def leg_entry(args, dt, steps, world, B, opix, roofline, phases=None):
    e = {"value": round(world * B * opix / 1e6 * steps / dt, 3), "unit": "MP/s", "n_gpus": world,
         "ms_per_step": round(1e3 * dt / steps, 2), "tiles_per_gpu": B, "scaling": "weak", "steps": steps,
         "timed_region": "barrier + synchronize, steps x (forward of this rank's tiles + all-gather of the " +
                         ("uint8 HR images [tiles,H,W] (tensor2img of SR, from the last kernel)" if args.gather == "u8" else "fp32 HR slabs [tiles,1,H,W]") +
                         " at N > 1), synchronize + barrier; max over ranks; no per-launch events in this pass", "gather": args.gather,
         "roofline": roofline}
    if phases is not None:
        e["rank_phases"] = phases
    return e


def run_precision_leg(args, opt, x, rank, world, dev, mode, sync, scale=None, steps=None, warmup=None):
    """One more configuration of the forward, measured the same way as the headline at the same N: every rank runs its tiles, the step
    ends with the all-gather of the HR slabs, barrier + synchronize on both sides, max over ranks (BASELINE configs[2] = "batch=128 tiles
    sharded over 8 MI355X, bf16 MFMA, RCCL all-gather of HR slabs" is the bf16 leg at --gpus 8; configs[3] = the x16 legs).
    Returns (entry, model, last output)."""
    import copy
    import torch
    from gpemsr_amd import dist as gdist
    B, s = x.shape[0], (scale or args.scale)
    largs = copy.copy(args)
    largs.steps = steps or args.steps
    if args.stub:
        m3 = _StubModel()
    else:
        from gpemsr_amd.config import build_model
        torch.cuda.empty_cache()
        m3 = build_model(opt, load_prior_files=False, precision=mode).eval().to(dev)
    timing = [] if world > 1 else None

    def step3():
        return gdist.forward_sharded(m3, x, rank, world, already_local=True, gather=True, gather_u8=args.gather == "u8", timing=timing)[0]
    for _ in range(max(args.warmup if warmup is None else warmup, 2)):          # the allocator re-grows its pools after empty_cache(): keep that out of the timing
        step3()
    if timing is not None:
        timing.clear()
    d3, o3, prof, dtp, psteps = measure_leg(step3, largs, world, dev, sync, stub=args.stub)
    phases = gdist.step_phase_times(timing[:largs.steps], world, dev) if timing is not None else None
    if prof is not None and args.layer_report and rank == 0:
        write_layer_report(prof, args.layer_report + "." + mode + (f".x{s}" if s != args.scale else ""))
    assert o3.shape[0] == B * world
    opix = o3.shape[-1] * o3.shape[-2]
    roof = None if args.stub else build_roofline(largs, prof, dtp, psteps, B, s, precision=mode)
    return leg_entry(args, d3, largs.steps, world, B, opix, roof, phases), m3, o3


MAX_LINE_BYTES = 8192              # the ONE stdout line; round 5's 34.7 KB line came back from the driver unparsed (BENCH_r05.parsed = null)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_roofline(r):
    """The contract's `roofline` object of the stdout line: the dominant kernel's figures + the family's, no prose and no tables
    (those are in the --detail document)."""
    if not r:
        return None
    c = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches_per_step",
                  "executed_gflop_per_launch", "algorithmic_gflop_per_launch", "algorithmic_bytes_per_launch", "effective_frac",
                  "time_share_of_step", "whole_path_tflops_essential", "algorithmic_bytes_per_step", "counter_bytes_per_step_all_kernels",
                  "counter_over_algorithmic_bytes"))
    c.setdefault("traffic", None)
    src = r.get("traffic_source") or {}
    if src:
        c["traffic_source"] = src.get("file")
        if src.get("kernel_mfma_util_pmc") is not None:
            c["mfma_util_pmc"] = src["kernel_mfma_util_pmc"]
    fam = r.get("family") or {}
    c["family"] = _pick(fam, ("achieved", "frac", "effective_frac", "launches_per_step", "avg_launch_us", "traffic", "time_share_of_step"))
    if r.get("legs"):
        c["legs"] = r["legs"]
    return c


def compact_line(d):
    """The ONE stdout line: the contract's keys, the compact `roofline` / `cpu_baseline`, the other BASELINE configurations as
    value_* / ms_per_step_* pairs.  Everything else stays in the --detail document."""
    legs = d.get("legs") or {}
    bf = ((d.get("extras") or {}).get("bf16") or {}) if isinstance(d.get("extras"), dict) else {}
    line = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "rccl_world", "dist_backend", "steps", "warmup", "ms_per_step",
                                   "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "gather")}
    line["dtype"] = str(line["dtype"]).split(" ")[0]
    line["config"] = _pick(d.get("config") or {}, ("workload", "tiles_per_gpu", "global_tiles", "lr", "scale", "parallelism"))
    roof = compact_roofline(d.get("roofline"))
    compact = {k: (compact_leg(v) if k != "train" else _pick(v, ("value", "unit", "ms_per_step", "steps", "n_gpus", "kernel", "frac", "family_frac")))
               for k, v in legs.items()}
    if bf:
        compact["x8_bf16"] = compact_leg(bf)
    if roof is not None and compact:
        roof["legs"] = compact              # inside the contract's `roofline` object: survives a driver that drops unknown top-level keys
    line["roofline"] = roof
    cb = d.get("cpu_baseline")
    line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "gpu_vs_cpu_rel_err_free_running")) if cb else None
    ph = d.get("rank_phases")
    if ph:
        line["rank_phases"] = _pick(ph, ("forward_ms_max", "forward_ms_min", "gather_ms_max", "gather_ms_min", "steps_recorded"))
    if bf:
        line["value_bf16"], line["ms_per_step_bf16"] = bf.get("value"), bf.get("ms_per_step")
        rb = bf.get("roofline") or {}
        if rb:
            line["roofline_bf16"] = {**_pick(rb, ("kernel", "achieved", "peak", "frac", "traffic", "avg_launch_us", "launches_per_step", "time_share_of_step")),
                                     "family": _pick(rb.get("family") or {}, ("frac", "launches_per_step", "time_share_of_step"))}
        if bf.get("rank_phases"):
            line["rank_phases_bf16"] = _pick(bf["rank_phases"], ("forward_ms_max", "forward_ms_min", "gather_ms_max", "gather_ms_min"))
        for k in ("rel_err_vs_fp32_path_teacher_forced_2_windows", "code_index_agreement_free_running_2_windows", "free_running_dpsnr"):
            if k in bf:
                line.setdefault("bf16_fidelity", {})[k] = float(f"{bf[k]:.4g}")
    for name in ("x16_fp32", "x16_bf16", "train"):
        if name in legs:
            line["value_" + name], line["ms_per_step_" + name] = legs[name].get("value"), legs[name].get("ms_per_step")
    vol = (d.get("extras") or {}).get("volume_mode") if isinstance(d.get("extras"), dict) else None
    if vol:
        line["value_volume_fp32"] = vol.get("value")
        if bf.get("volume_mode_value") is not None:
            line["value_volume_bf16"] = bf["volume_mode_value"]
    return line


def compact_leg(entry):
    """The few figures of a leg that ride at the top level of the line (and inside `roofline.legs`, which a driver that keeps only the
    contract's keys still keeps): value, ms/step, the dominant kernel and its executed fraction, the family's."""
    r = entry.get("roofline") or {}
    return {"value": entry["value"], "unit": entry["unit"], "ms_per_step": entry["ms_per_step"], "steps": entry.get("steps"), "n_gpus": entry.get("n_gpus"),
            "kernel": r.get("kernel"), "frac": r.get("frac"), "family_frac": (r.get("family") or {}).get("frac"),
            "effective_frac": r.get("effective_frac")}


def run_config_legs(args, rank, world, dev, sync):
    """BASELINE configs[3] and configs[4] on the default line: 16x EMSR, 8 windows per GPU of 5x1x64x64 -> 1024^2
    (R:option/output_GPEMSR_x16.yml) in fp32 and on the bf16 data path, and the stage-3 training step (batch 8 per GPU of 32^2 -> 256^2
    crops, R:train_stage3.py:343-366).  Fewer steps than the headline (the whole default command has to stay within minutes)."""
    import torch
    from gpemsr_amd.config import load_options
    from gpemsr_amd.synth import synth_lr_tiles
    legs = {}
    steps, warm = max(1, min(args.steps, 10)), 2
    opt16 = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x16.yml"))
    x16 = synth_lr_tiles(8, 5, 64, 64, seed=1600 + rank, kind="uniform").to(dev)
    for mode in ("bf16", "fp32"):
        e, m, o = run_precision_leg(args, opt16, x16, rank, world, dev, mode, sync, scale=16, steps=steps, warmup=warm)
        e["config"] = f"16x EMSR stage-3 forward, 8 windows per GPU of 5x1x64x64 -> 1024x1024, {mode} (BASELINE.json configs[3], one GPU's share)"
        legs["x16_" + mode] = e
        del m, o
        torch.cuda.empty_cache()
    import bench_train
    legs["train"] = bench_train.train_leg(ROOT, 8, rank, world, dev, steps=steps, warmup=warm)
    torch.cuda.empty_cache()
    return legs


def run_extras(args, model, opt, x, out, dt, dev, rank=0, world=1, sync=None):
    """Measured beside the official number, never replacing it.  At every N: the bf16 configuration (BASELINE configs[2]) as
    its own leg with the all-gather inside its timed region.  At N = 1 additionally: volume mode (what
    output_GPEMSR.py runs) and the bf16 leg's error against this run's fp32 outputs (teacher-forced = the fp32 run's code
    indices, free-running = its own) incl. the free-running PSNR difference."""
    import torch
    B, s, lr = args.tiles, args.scale, args.lr
    modes = [m for m in args.extras.split(",") if m]
    if world > 1 or args.stub or args.gather == "u8":      # (u8 gather: `out` is the 8-bit image, the fp32 comparisons below do not apply)
        extras = {}
        if args.precision == "fp32" or args.stub:
            del model
            for mode in modes:
                extras[mode], m3, o3 = run_precision_leg(args, opt, x, rank, world, dev, mode, sync)
                del m3, o3
        return extras
    from gpemsr_amd import ops
    from gpemsr_amd.imgutil import calculate_psnr
    from gpemsr_amd.synth import synth_lr_tiles
    T = B + 4
    fr = synth_lr_tiles(1, T, lr, lr, seed=77, kind="smooth")[0].to(dev)
    rows = [[min(max(c + o, 0), T - 1) for o in (-2, -1, 0, 1, 2)] for c in range(T)]
    win = torch.tensor(rows, dtype=torch.int32, device=dev)
    value = B * (lr * s) ** 2 / 1e6 * args.steps / dt
    dv = _volume_bench(model, fr, win, args.steps)
    extras = {"volume_mode": {"value": round(T * (lr * s) ** 2 / 1e6 / dv, 3), "unit": "MP/s", "ms_per_volume": round(1e3 * dv, 2),
                              "workload": f"{T} consecutive {lr}x{lr} LR slices -> {T} HR slices of {lr * s}^2 (sliding 5-slice windows, "
                                          "per-slice features cached; output_GPEMSR.py's loop)", "precision": args.precision,
                              "speedup_vs_independent_windows": round(T * (lr * s) ** 2 / 1e6 / dv / value, 3)}}
    if args.precision != "fp32":
        return extras
    out_ref = out[:B].clone()
    tr_ref = {}
    o2_ref, _ = model(x[:2], trace=tr_ref)                  # two windows, for the teacher-forced comparison
    idx_ref = torch.cat(tr_ref["code_idx"])
    del model

    def psnr_vs_base(o, k):                                 # the reference's image metric (R:util/util.py:253-260) against the bilinear base
        u8 = ops.tensor2img_u8(o[k, 0]).cpu().numpy()
        base = torch.nn.functional.interpolate(x[k:k + 1, 2], scale_factor=s, mode="bilinear", align_corners=False)
        b8 = (base.squeeze().clamp(0, 1) * 255.0).round().to(torch.uint8).cpu().numpy()
        return calculate_psnr(u8, b8), u8
    for mode in modes:
        entry, m3, o3 = run_precision_leg(args, opt, x, 0, 1, dev, mode, torch.cuda.synchronize)
        rel = float((o3[:B] - out_ref).abs().max() / out_ref.abs().max())
        tr3 = {}
        o2_tf, _ = m3(x[:2], forced_code_idx=idx_ref)
        m3(x[:2], trace=tr3)
        rel_tf = float((o2_tf - o2_ref).abs().max() / o2_ref.abs().max())
        agree = float((torch.cat(tr3["code_idx"]) == idx_ref).float().mean())
        dps, du8 = [], 0
        for k in range(min(B, 4)):                          # free-running image quality, this leg against the fp32 path, first windows
            p3, u3 = psnr_vs_base(o3, k)
            p0, u0 = psnr_vs_base(out_ref, k)
            dps.append(abs(p3 - p0))
            du8 = max(du8, int(abs(u3.astype("int32") - u0.astype("int32")).max()))
        vol = _volume_bench(m3, fr, win, args.steps)
        entry.update({"volume_mode_value": round(T * (lr * s) ** 2 / 1e6 / vol, 3),
                      "rel_err_vs_fp32_path_teacher_forced_2_windows": rel_tf,
                      "code_index_agreement_free_running_2_windows": agree,
                      "rel_err_vs_fp32_path_free_running_all_windows": rel,
                      "free_running_dpsnr": round(max(dps), 5),
                      "free_running_u8_max_level_diff": du8,
                      "speedup_vs_fp32_path": round(dt / (entry["ms_per_step"] * 1e-3 * args.steps), 3)})
        extras[mode] = entry
        del m3, o3
    return extras


def run_cpu_baseline(args, model_sd, x, out):
    """Bounded sample of the same workload: ONE 5-slice window through the CPU oracle on the host cores, a same-size
    warm-up pass + 2 timed passes."""
    import torch
    from gpemsr_amd.synth import synth_lr_tiles
    from oracle import gpemsr_oracle as orc
    s, lr = args.scale, args.lr
    cores = effective_cores()
    torch.set_num_threads(cores)
    xc = x[:1].cpu() if args.cpu_lr == lr else synth_lr_tiles(1, 5, args.cpu_lr, args.cpu_lr, seed=1000)
    times = []
    with torch.no_grad():
        for i in range(3):
            t1 = time.perf_counter()
            o_cpu, _ = orc.gpemsr_forward(model_sd, xc, scale=s)
            if i > 0:
                times.append(time.perf_counter() - t1)
    cdt = sum(times) / len(times)
    cpu_mp = (args.cpu_lr * s) ** 2 / 1e6 / cdt
    cb = {"value": round(cpu_mp, 5), "unit": "output megapixels/s", "cores": torch.get_num_threads(), "kind": "port",
          "sample": f"1 window [1,5,1,{args.cpu_lr},{args.cpu_lr}] -> {args.cpu_lr * s}^2: warm-up pass + 2 timed passes "
                    f"({times[0]:.1f} s, {times[1]:.1f} s) of oracle/gpemsr_oracle.py (torch CPU fp32)"}
    if args.cpu_lr == lr and args.precision == "fp32" and out.dtype == torch.float32:
        err = float((out[:1].cpu() - o_cpu).abs().max() / o_cpu.abs().max())
        cb["gpu_vs_cpu_rel_err_free_running"] = float(f"{err:.3e}")
    return cb


def run_forward(args) -> int:
    import torch
    from gpemsr_amd import dist as gdist

    t_start = time.perf_counter()
    wall = {}

    def lap(name):                      # wall seconds of each section of the command (the default line has to fit the driver's run)
        nonlocal t_start
        now = time.perf_counter()
        wall[name] = round(now - t_start, 1)
        t_start = now

    rank, world, local = gdist.init_from_env(backend=args.backend or None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    B, s, lr = args.tiles, args.scale, args.lr
    if args.stub and os.environ.get("GPEMSR_BENCH_TEST_HANG_RANK") == str(rank):
        time.sleep(3600)                    # launcher self-test: this rank never arrives (tests/test_bench_launcher_cpu.py)
    if world > 1 and args.backend != "gloo":
        # a silent gloo fall-back on a multi-GPU box would read as bad scaling: RCCL ("nccl" on ROCm) unless gloo was asked for explicitly
        assert torch.distributed.get_backend() == "nccl", f"N > 1 runs over RCCL; got backend {torch.distributed.get_backend()!r} (pass --backend gloo for a rehearsal)"
    if args.stub:
        dev = torch.device("cpu")
        sync = lambda: None                                                        # noqa: E731
        model = _StubModel()
        x = torch.rand(B, 5, 1, lr, lr, generator=torch.Generator().manual_seed(1000 + rank))
        opt = None
    else:
        from gpemsr_amd.config import build_model, load_options
        from gpemsr_amd.synth import synth_lr_tiles
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU path)"
        if os.environ.get("GPEMSR_BENCH_SHARE_GPU") == "1":       # rehearsal on a box with fewer GPUs than ranks (use with --backend gloo)
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        sync = torch.cuda.synchronize
        opt = load_options(os.path.join(ROOT, "option", f"output_GPEMSR_x{s}.yml"))
        model = build_model(opt, load_prior_files=False, precision=args.precision).eval().to(dev)
        x = synth_lr_tiles(B, 5, lr, lr, seed=1000 + rank, kind="uniform").to(dev)     # resident in HBM before timing
    timing = [] if world > 1 else None

    def step():
        out, _ = gdist.forward_sharded(model, x, rank, world, already_local=True, gather=True, gather_u8=args.gather == "u8", timing=timing)
        return out

    lap("setup")
    for _ in range(args.warmup):
        step()
    if timing is not None:
        timing.clear()
    # the official number: K steps with NO per-launch events; the tables: a second, profiled pass (measure_leg)
    dt, out, prof, dtp, psteps = measure_leg(step, args, world, dev, sync, stub=args.stub)
    rank_phases = gdist.step_phase_times(timing[:args.steps], world, dev) if timing is not None else None
    assert out.shape[0] == B * world
    rccl_world = torch.distributed.get_world_size() if world > 1 else 1           # the group the all-gathers above ran on
    dist_backend = torch.distributed.get_backend() if world > 1 else None         # "nccl" = RCCL on ROCm; "gloo" in rehearsals / self-tests

    lap("headline")
    opix = out.shape[-1] * out.shape[-2]
    mp_per_step = world * B * opix / 1e6
    value = mp_per_step * args.steps / dt
    roofline, extras, cpu_baseline, legs = None, None, None, None
    if args.stub and not args.no_extras:
        extras = run_extras(args, model, None, x, out, dt, dev, rank, world, sync)
    if not args.stub:
        roofline = build_roofline(args, prof, dtp, psteps, B, s, args.precision)
        if args.layer_report and rank == 0:
            write_layer_report(prof, args.layer_report)
        model_sd = {k: v.detach().cpu() for k, v in model.state_dict().items()} if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
        if not args.no_extras:
            extras = run_extras(args, model, opt, x, out, dt, dev, rank, world, sync)
        del model
        lap("extras")
        if not args.no_extras and not args.no_config_legs and s == 8 and args.precision == "fp32" and args.gather == "f32":
            legs = run_config_legs(args, rank, world, dev, sync)
            lap("config_legs")
        if model_sd is not None:
            cpu_baseline = run_cpu_baseline(args, model_sd, x, out)
            lap("cpu_baseline")

    io_edges = None
    if not args.stub and rank == 0 and world == 1 and not args.no_extras:
        io_edges = measure_io_edges(out, dev)
    if rank == 0:
        cfg_name = {8: "BASELINE.json configs[1]" if args.precision == "fp32" else "BASELINE.json configs[2], one GPU's share",
                    16: "BASELINE.json configs[3], one GPU's share"}[s]
        dtype = {"fp32": "f32", "bf16": "bf16 (bf16 activations in HBM, bf16 MFMA, fp32 accumulate; indexer logits at fp32 precision = 3 bf16 products of hi+lo operands, argmax on fp32 logits)",
                 "bf16x3": "bf16x3 (fp32 activations; convs as 3 split hi+lo bf16 MFMA products, fp32 accumulate)",
                 "bf16op": "bf16 operands rounded in the kernel (fp32 activations in HBM), fp32 accumulate"}[args.precision]
        bf = ((extras or {}).get("bf16") or {}) if isinstance(extras, dict) else {}
        detail = {
            "metric": "output megapixels/sec, 8x EMSR 128->1024 tiles" if s == 8 else "output megapixels/sec, 16x EMSR 64->1024 tiles",
            "value": round(value, 3), "unit": "MP/s", "n_gpus": world, "rccl_world": rccl_world, "dist_backend": dist_backend, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak",
            "timed_region": "barrier + synchronize, steps x (forward of this rank's tiles + all-gather of the " +
                            ("uint8 HR images" if args.gather == "u8" else "fp32 HR slabs") + " at N > 1), synchronize + barrier; max over ranks; "
                            "no per-launch events in this pass (the roofline tables come from a second pass; at N > 1 three phase marks per step)",
            "gather": args.gather,
            "vs_baseline": None, "dtype": "stub" if args.stub else dtype, "data": "synthetic",
            "config": {"workload": ("launcher self-test (stub model, CPU, gloo)" if args.stub else
                                    f"{s}x EMSR stage-3 forward, batch={B} synthetic 5x1x{lr}x{lr} LR windows per GPU -> "
                                    f"{lr * s}x{lr * s} HR tiles, {args.precision} ({cfg_name})"),
                       "tiles_per_gpu": B, "global_tiles": B * world, "lr": lr, "scale": s,
                       "weights": "deterministic synthetic init (reference checkpoints are not redistributable)",
                       "parallelism": f"tiles sharded over {world} GPU(s), one process per GPU, RCCL all-gather of HR slabs" if world > 1 else "single GPU"},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
            "rank_phases": rank_phases,
            "legs": legs,
            "io_edges": io_edges,
            "extras": extras,
            "wall_s_by_section": wall,
        }
        line = compact_line(detail)
        line["detail"] = args.detail or None
        text = json.dumps(line, separators=(",", ":"))
        assert len(text) < MAX_LINE_BYTES, f"bench line is {len(text)} bytes (limit {MAX_LINE_BYTES}): the driver could not parse a 34.7 KB line in round 5"
        detail["line_bytes"] = len(text)
        if args.detail:
            try:
                with open(os.path.join(ROOT, args.detail) if not os.path.isabs(args.detail) else args.detail, "w") as f:
                    json.dump(detail, f, indent=1)
                print(f"[bench] full measurement document: {args.detail} ({len(text)}-byte line on stdout)", file=sys.stderr, flush=True)
            except OSError as e:
                print(f"[bench] could not write {args.detail}: {e}", file=sys.stderr, flush=True)
        print(text, flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


def measure_io_edges(out, dev):
    """SURVEY 8(f)3, outside the timed region: the step's HR tiles as 8-bit images -> complete PNG files on the device (csrc/png.hip, png_huff.hip):
    tensor2img, then the stored-block and the Huffman-compressing encoder; bytes moved = image read by the assemble / histogram / pack and
    checksum passes + file written and read once (DESIGN_HISTORY.md §3.12)."""
    import torch
    from gpemsr_amd import ops, png
    u8 = ops.tensor2img_u8(out.reshape(-1, out.shape[-2], out.shape[-1]))
    n = u8.shape[0]
    res = {"images": int(n), "image_shape": [int(u8.shape[-2]), int(u8.shape[-1])]}
    for name, fn in (("stored", lambda: png.encode_gray8(u8)), ("huffman", lambda: png.encode_gray8_compressed(u8))):
        for _ in range(2):
            r = fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            r = fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        nbytes = float(r.numel()) / n if name == "stored" else float(r[1].float().mean().item())
        res[f"png_encode_{name}"] = {"ms_per_batch": round(ms, 4), "images_per_s": round(n / ms * 1e3, 0), "bytes_per_file": round(nbytes, 0)}
    return res


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and env_world == 1 and not os.environ.get(CHILD_ENV):
        return spawn_ranks(args, argv)                   # plain `python bench.py --gpus N`: start our own ranks
    if args.mode == "train":
        import bench_train
        bench_train.run(args, ROOT, effective_cores)
        return 0
    if args.mode == "train2":
        import bench_train
        bench_train.run_stage2(args, ROOT, effective_cores)
        return 0
    if args.mode == "train1":
        import bench_train
        bench_train.run_stage1(args, ROOT, effective_cores)
        return 0
    return run_forward(args)


if __name__ == "__main__":
    sys.exit(main())
