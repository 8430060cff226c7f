"""Build libgpemsr_hip.so (gfx950) in-tree with hipcc.  No torch involved: the
library is a plain C-ABI shared object (include/gpemsr_hip.h)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgpemsr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
         "-I" + os.path.join(REPO, "include"), "-I" + CSRC]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    # every header and generated include (csrc/*.h, csrc/*.inc -- e.g. attn_agpr.inc from scripts/gen_attn_agpr.py) is a dependency of every object
    hdrs = [os.path.join(REPO, "include", "gpemsr_hip.h")] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")))
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print("[gpemsr build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    # the flash-attention object is checked whenever it is newer than the stamp of its last successful check (not only when something was
    # compiled in THIS call: a failed check must not be forgotten by the next call); a rejected object and the library are removed
    attn_obj, stamp = os.path.join(LIBDIR, "attn_bf16.o"), os.path.join(LIBDIR, "attn_bf16.checked")
    if os.path.exists(attn_obj) and _stale(stamp, [attn_obj]):
        try:
            check_flash_attention_object(attn_obj)
        except Exception:
            for f in (attn_obj, LIB, stamp):
                if os.path.exists(f):
                    os.remove(f)
            raise
        with open(stamp, "w") as f:
            f.write("flash_attn512_kernel: 768 / 768 AGPR accesses, no scratch\n")
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[gpemsr build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def check_flash_attention_object(obj: str) -> None:
    """(obj: the object just built; the check recompiles the source to device assembly, ~2 s.)  flash_attn512_kernel keeps O^T in a[0:255] across inline-asm statements that declare those registers only as clobbers, so the
    compiler is free to use AGPRs for copies or spills in between -- a different hipcc could do so silently.  Checked on the object that
    was just built: no scratch traffic, and every v_accvgpr access is one of the generated macros' (768 writes = zero-fill + rescale path,
    768 reads = rescale + epilogue; scripts/gen_attn_agpr.py).  A failing check stops the build: set GPEMSR_FLASH_ATTN=0 to run the
    layered attention instead and report the compiler version."""
    src = os.path.join(CSRC, "attn_bf16.hip")
    if not os.path.exists(src):
        return
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([HIPCC] + FLAGS + ["-S", "--cuda-device-only", "-Wno-unused-command-line-argument", src, "-o", f"{td}/attn.s"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed for {src}:\n{r.stderr}")
        body = open(f"{td}/attn.s").read()
    if "flash_attn512_kernel" not in body:
        raise RuntimeError("attn_bf16.hip: flash_attn512_kernel not found in the device assembly")
    wr, rd = body.count("v_accvgpr_write"), body.count("v_accvgpr_read")
    scratch = body.count("scratch_load") + body.count("scratch_store") + (0 if "; ScratchSize: 0" in body else 1)
    if scratch or wr != 768 or rd != 768:
        raise RuntimeError(f"attn_bf16.o: flash_attn512_kernel was compiled with {wr} AGPR writes / {rd} reads (expected 768 / 768 from "
                           f"attn_agpr.inc) and {scratch} scratch instructions: the compiler touched the accumulator half; "
                           "do not ship this object (GPEMSR_FLASH_ATTN=0 selects the layered attention)")


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
