"""Dev helper (GPU box, GPEMSR_LIB_PATH = a -DGP_STAMP build): per-block s_memtime stamps of conv_mfma_kernel
(entry, main loop start, epilogue start, exit) -> median phase durations in 100 MHz ticks (10 ns)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gpemsr_amd import ops, _abi
from gpemsr_amd.packing import pack_conv, pack_convT
dev = torch.device("cuda", 0)
lib = _abi.load()
lib.gpemsr_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
CASES = [("rb64_512", 20, 64, 64, 512, "conv"), ("vq256_128", 20, 256, 256, 128, "conv"), ("convT64_512", 20, 64, 64, 512, "convT")]
for name, n, cin, cout, hw, kind in CASES:
    x = ops.from_nhwc(torch.randn(n, hw, hw, cin, device=dev))
    pc = pack_convT(torch.randn(cin, cout, 3, 3) * 0.05, torch.randn(cout), dev) if kind == "convT" else \
        pack_conv(torch.randn(cout, cin, 3, 3) * 0.05, torch.randn(cout), dev)
    for _ in range(3):
        out = ops.conv2d([x], pc, ops.ACT_LRELU)
    torch.cuda.synchronize()
    th = 8 if (kind == "convT" or cout <= 64) else 4
    bn = 128 if (kind == "convT" or cout > 64) else 64
    nb = min(65536, n * (hw // th) * (hw // 32) * (((4 * cout) if kind == "convT" else cout) + bn - 1) // bn)
    buf = np.zeros(8 * nb, dtype=np.uint64)
    assert lib.gpemsr_debug_read_stamps(buf.ctypes.data, nb) == 0
    st = buf.reshape(nb, 8).astype(np.int64)
    med = lambda a: float(np.median(a)) / 1e3
    # stamps: 0 entry, 1 main loop start, 2 epilogue start, 4 pass-0 prefetch issued, 5 pass-0 acc->LDS done, 6 pass-0 stores issued, 3 exit
    print(f"{name:12s} blocks {nb} (kcycles, median): prologue {med(st[:,1]-st[:,0]):.1f} | main {med(st[:,2]-st[:,1]):.1f} | "
          f"epilogue {med(st[:,3]-st[:,2]):.1f} = [sync+addr+prefetch {med(st[:,4]-st[:,2]):.1f}, acc->LDS+barrier {med(st[:,5]-st[:,4]):.1f}, "
          f"readout+store {med(st[:,6]-st[:,5]):.1f}, other passes {med(st[:,3]-st[:,6]):.1f}] | total {med(st[:,3]-st[:,0]):.1f}")
