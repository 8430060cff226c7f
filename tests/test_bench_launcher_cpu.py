"""`python bench.py --gpus N` must start its own N ranks (the one-process-per-GPU launch of
R:train_stage3.py:20-27).  Driven here on the CPU: --backend gloo with the per-tile stub model, so the launcher, the
env:// rendezvous, the barrier + max-over-ranks timing and the all-gather of the output slabs all really run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=300, detail=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GPEMSR_BENCH_CHILD")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--backend", "gloo", "--tiles", "3", "--lr", "16",
                           "--steps", "2", "--warmup", "1", "--detail", detail or ""] + extra, env=env, capture_output=True, text=True, timeout=timeout)


def _one_line(stdout):
    """The contract: ONE JSON line on stdout, short enough for the driver to keep (BENCH_r05.parsed was null on a 34.7 KB line)."""
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    assert len(lines[0]) < 8192
    return json.loads(lines[0])


def test_plain_invocation_spawns_two_ranks(tmp_path):
    detail = str(tmp_path / "detail.json")
    r = _run(["--gpus", "2"], detail=detail)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_line(r.stdout)
    assert line["n_gpus"] == 2 and line["value_bf16"] > 0 and line["ms_per_step_bf16"] > 0
    for ph in (line["rank_phases"], line["rank_phases_bf16"]):          # the compact form on the line: max / min over ranks
        assert ph["forward_ms_max"] >= ph["forward_ms_min"] > 0 and ph["gather_ms_max"] >= ph["gather_ms_min"] > 0
    d = json.load(open(detail))                                          # the full document (per-rank lists, full legs)
    assert d["value"] == line["value"] and d["line_bytes"] < 8192
    assert d["n_gpus"] == 2 and d["rccl_world"] == 2 and d["config"]["global_tiles"] == 6
    assert d["dist_backend"] == "gloo"                      # `rccl_world` alone cannot tell a rehearsal from an RCCL run
    assert d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    # BASELINE configs[2] (bf16, tiles sharded over the GPUs, all-gather of the slabs) rides along at EVERY N as its own leg
    leg = d["extras"]["bf16"]
    assert leg["n_gpus"] == 2 and leg["value"] > 0 and leg["ms_per_step"] > 0 and leg["tiles_per_gpu"] == 3
    assert "all-gather" in leg["timed_region"]
    # at N > 1 every leg says where a step's time went, rank by rank: forward and all-gather separately, max / min over ranks
    for ph in (d["rank_phases"], leg["rank_phases"]):
        assert len(ph["forward_ms_per_rank"]) == 2 and len(ph["gather_ms_per_rank"]) == 2 and ph["steps_recorded"] == 2
        assert ph["forward_ms_max"] >= ph["forward_ms_min"] > 0 and ph["gather_ms_max"] >= ph["gather_ms_min"] > 0
    assert "no per-launch events" in d["timed_region"]


def test_a_silent_gloo_fallback_is_refused():
    """Without `--backend gloo` an N > 1 run must be on RCCL ("nccl"): on this GPU-less container torch.distributed would pick gloo by
    itself, which on an 8-GPU box would read as bad scaling -- the rank asserts instead."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GPEMSR_BENCH_CHILD")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--tiles", "2", "--lr", "16", "--steps", "1", "--warmup", "0", "--gpus", "2",
                        "--rank-timeout", "120"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_single_rank_runs_in_process():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 1 and d["rccl_world"] == 1 and d["dist_backend"] is None
    assert d["value_bf16"] > 0 and "rank_phases" not in d


def test_failing_rank_fails_the_launcher():
    # a bogus backend makes every rank raise in init_process_group: the parent must report failure, not hang or print a line
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GPEMSR_BENCH_CHILD")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--backend", "no_such_backend", "--gpus", "2", "--tiles", "2",
                        "--lr", "16", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_under_torchrun_env_it_is_a_rank_not_a_launcher():
    # the driver's N>1 form sets WORLD_SIZE itself (python -m torch.distributed.run ...): no second level of spawning
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    base = dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    env0 = {k: v for k, v in os.environ.items() if k != "GPEMSR_BENCH_CHILD"}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--backend", "gloo", "--gpus", "2", "--tiles", "2", "--lr", "16",
           "--steps", "1", "--warmup", "0"]
    procs = [subprocess.Popen(cmd, env=dict(env0, **base, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    d = _one_line(outs[0][0])
    assert d["n_gpus"] == 2 and d["rccl_world"] == 2
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]


def test_eight_ranks_rehearsal_gloo(tmp_path):
    """The driver's N = 8 form cannot be rehearsed on hardware from here (no 8-GPU node): eight gloo ranks of the stub model exercise what
    is NOT kernel work at that width -- port choice, rendezvous, barrier, the all-gather of 8 slabs, max-over-ranks timing, the bf16 leg."""
    detail = str(tmp_path / "detail.json")
    r = _run(["--gpus", "8", "--rank-timeout", "240"], timeout=400, detail=detail)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_line(r.stdout)
    assert line["n_gpus"] == 8 and line["rank_phases"]["gather_ms_max"] > 0 and line["value_bf16"] > 0
    d = json.load(open(detail))
    assert d["n_gpus"] == 8 and d["rccl_world"] == 8 and d["dist_backend"] == "gloo" and d["config"]["global_tiles"] == 24
    assert d["gather"] == "f32" and "fp32 HR slabs" in d["timed_region"]
    assert d["extras"]["bf16"]["n_gpus"] == 8 and d["extras"]["bf16"]["gather"] == "f32"
    assert len(d["rank_phases"]["forward_ms_per_rank"]) == 8 and d["rank_phases"]["gather_ms_max"] > 0


def test_uint8_gather_option_two_ranks(tmp_path):
    """--gather u8: the step exchanges the 8-bit image of the last kernel (1/4 of the bytes); stated in `gather` on the line and in the
    document's `timed_region`."""
    detail = str(tmp_path / "detail.json")
    r = _run(["--gpus", "2", "--gather", "u8"], detail=detail)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _one_line(r.stdout)["gather"] == "u8"
    d = json.load(open(detail))
    assert d["gather"] == "u8" and "uint8" in d["timed_region"] and d["value"] > 0
    assert "uint8" in d["extras"]["bf16"]["timed_region"]


def test_rank_timeout_stops_a_hung_launch():
    """A rank that never finishes must not hang the launcher: --rank-timeout ends it with a non-zero code and no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GPEMSR_BENCH_CHILD")}
    env["GPEMSR_BENCH_TEST_HANG_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--backend", "gloo", "--gpus", "2", "--tiles", "2", "--lr", "16",
                        "--steps", "1", "--warmup", "0", "--rank-timeout", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_the_full_measurement_document_compacts_to_a_line_the_driver_keeps():
    """Round 5's own full document (34.7 KB printed as the line then; `BENCH_r05.parsed` = null) through compact_line: contract keys,
    `roofline` with the six contract fields + family, `cpu_baseline`, the other configurations' values -- under 8 KB, also at N = 2 where
    every leg carries rank phases."""
    sys.path.insert(0, ROOT)
    import bench
    for name in ("r05_default_bench.json", "r05_gpus2_rehearsal.json"):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        line = bench.compact_line(d)
        text = json.dumps(line, separators=(",", ":"))
        assert len(text) < bench.MAX_LINE_BYTES == 8192, (name, len(text))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                  "roofline", "cpu_baseline"):
            assert k in line, k
        assert "workload" in line["config"] and "model" not in line["config"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "family", "legs"):
            assert k in line["roofline"], k
        assert 0 < line["roofline"]["frac"] <= 1 and line["roofline"]["family"]["frac"] > 0
        for k in ("value_bf16", "value_x16_fp32", "value_x16_bf16", "value_train"):
            assert line[k] > 0, k
        if d["n_gpus"] == 1:
            assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
