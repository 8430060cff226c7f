import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
try:
    print("cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
from gpemsr_amd.arch import param_specs
from gpemsr_amd.synth import synth_state_dict, synth_lr_tiles
from gpemsr_amd.config import load_options
from oracle import gpemsr_oracle as orc
opt = load_options("option/output_GPEMSR_x8.yml")
kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
sd = synth_state_dict(param_specs(scale=8, **kw))
for thr in (8, 16, 32, 64):
    torch.set_num_threads(thr)
    x = synth_lr_tiles(1, 5, 32, 32, seed=1)
    with torch.no_grad():
        t = time.perf_counter(); orc.gpemsr_forward(sd, x, scale=8); dt = time.perf_counter() - t
    print("threads", thr, "LR32 window", round(dt, 2), "s", flush=True)
