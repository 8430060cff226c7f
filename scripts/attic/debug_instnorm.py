import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from gpemsr_amd import ops
from gpemsr_amd.ops import Act
DEV = torch.device("cuda", 0)
def act(t):
    n, h, w, c = t.shape
    return Act(t.to(DEV, torch.float32).contiguous(), n, h, w, c, c, 0)
for (n, h, w, c) in [(1, 14, 14, 256), (1, 30, 30, 128), (2, 30, 30, 128), (1, 16, 16, 128), (1, 20, 20, 128), (1, 30, 30, 64), (1,30,30,256)]:
    gen = torch.Generator().manual_seed(n * 1000 + c)
    x = torch.randn(n, h, w, c, generator=gen, dtype=torch.float64) * 1.7 + 0.3
    dy = torch.randn(n, h, w, c, generator=gen, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    y = F.instance_norm(xr.permute(0, 3, 1, 2), eps=1e-5).permute(0, 2, 3, 1)
    (dx,) = torch.autograd.grad(y, xr, dy)
    xa = act(x)
    ya, mr = ops.instnorm(xa)
    got = ops.instnorm_bwd(xa, mr, act(dy)).torch().view(x.shape).cpu().double()
    print((n, h, w, c), "rel err", ((got - dx).abs().max() / dx.abs().max()).item(), "fwd", ((ya.torch().view(x.shape).cpu().double() - y).abs().max()).item())
