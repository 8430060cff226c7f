"""Dev helper (GPU box): run-to-run bit-stability screen of gpemsr_conv2d on many shapes (races show up as diffs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpemsr_amd import ops
from gpemsr_amd.packing import pack_conv, pack_convT
import torch.nn.functional as F
dev = torch.device("cuda", 0)
bad = 0
CASES = [(20, 64, 64, 3, 1, 128, 128), (4, 64, 64, 3, 1, 512, 512), (8, 256, 256, 3, 1, 128, 128), (8, 512, 512, 3, 1, 64, 64),
         (4, 32, 64, 7, 1, 256, 256), (8, 512, 512, 1, 1, 64, 64), (8, 64, 64, 3, 2, 256, 256), (3, 64, 216, 3, 1, 100, 76),
         (5, 128, 64, 3, 1, 36, 44), (2, 64, 32, 3, 1, 130, 94), (4, 64, 64, 0, 1, 128, 128), (4, 512, 256, 0, 1, 32, 32)]
for (n, cin, cout, k, stride, h, w) in CASES:
    x = torch.randn(n, cin, h, w)
    if k == 0:
        wt = torch.randn(cin, cout, 3, 3) * 0.05; b = torch.randn(cout)
        pc = pack_convT(wt, b, dev); want = F.conv_transpose2d(x, wt, b, stride=2, padding=1, output_padding=1)
    else:
        wt = torch.randn(cout, cin, k, k) * (1.0 / (cin * k * k) ** 0.5); b = torch.randn(cout)
        pc = pack_conv(wt, b, dev); want = F.conv2d(x, wt, b, stride, k // 2)
    xa = ops.from_nchw(x.to(dev))
    ref = None
    for rep in range(6):
        out = ops.conv2d([xa], pc, 0, stride=stride if k else 1)
        torch.cuda.synchronize()
        o = out.torch().clone()
        if ref is None:
            ref = o
            err = float((out.nchw().cpu() - want).abs().max() / want.abs().max())
        elif not torch.equal(o, ref):
            bad += 1
            print("NONDETERMINISTIC", (n, cin, cout, k, stride, h, w), float((o - ref).abs().max()))
            break
    print((n, cin, cout, k, stride, h, w), "rel err %.2e" % err, flush=True)
    if err > 1e-4:
        bad += 1
print("BAD" if bad else "CLEAN", bad)
