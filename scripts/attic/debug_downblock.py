import sys, time, torch
sys.path.insert(0, ".")
from gpemsr_amd import ops
from gpemsr_amd.ops import ACT_NONE
from gpemsr_amd.packing import pack_conv
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for (n, c, h) in ((8, 128, 256), (8, 128, 512), (8, 128, 513), (4, 128, 512), (8, 128, 384)):
    wt = (torch.rand(c, c, 3, 3, generator=g) * 2 - 1) / (c * 9) ** 0.5
    pc = pack_conv(wt, torch.rand(c), dev)
    x = ops.from_nhwc((torch.rand(n, h, h, c, generator=g) * 2 - 1).to(dev))
    for _ in range(2):
        y = ops.conv2d([x], pc, ACT_NONE, stride=2, precision="fp32")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        y = ops.conv2d([x], pc, ACT_NONE, stride=2, precision="fp32")
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print(n, c, h, "->", y.h, y.w, f"{ms:.3f} ms", f"{2 * 9 * c * c * n * y.h * y.w / ms / 1e9:.1f} TF")
