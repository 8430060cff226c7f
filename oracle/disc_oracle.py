"""TEST INFRASTRUCTURE ONLY (imported by tests/ alone): CPU restatement, in torch float64 under autograd, of the discriminator side of the
adversarial phase of stage-1 training -- what gpemsr_amd/discriminator.py + csrc/stage1_adv.hip compute on the GPU.

  * ``DiscOracle``           R:model/discriminator.py:13-32: Conv2d(im_channel, nf, 4, 2, 0) + LeakyReLU(0.2); n_layers times
                             Conv2d(.., 4, 2 (1 for the last), 0, bias=False) + InstanceNorm2d + LeakyReLU(0.2); Conv2d(.., 1, 4, 1, 0).
  * ``generator_gan_term``   R:train_stage1.py:305-306: g_loss = mean(-D(decoded)).
  * ``discriminator_losses`` R:train_stage1.py:333-345,360-372: d_loss = 0.5 * (mean(-D(imgs)) + mean(D(decoded))), and the R1 penalty
                             mean_b sum (d sum(D(imgs)) / d imgs)^2 with create_graph=True, scaled r1_reg_weight / 2 * net_d_reg_every.

Pinned by tests/test_oracle_golden.py against tests/golden/stage1_adv.npz, which the UNMODIFIED reference ``train_vqgan_onestep`` emitted
(oracle/gen_golden_stage1_adv.py): logged losses and the gradient statistics of every discriminator tensor."""
import torch
import torch.nn.functional as F


def layer_plan(im_channel: int, num_filters_last: int, n_layers: int):
    """[(state-dict index, cin, cout, stride, bias, instance norm, leaky relu)]"""
    plan = [(0, im_channel, num_filters_last, 2, True, False, True)]
    mult, idx = 1, 2
    for i in range(1, n_layers + 1):
        last, mult = mult, min(2 ** i, 8)
        plan.append((idx, num_filters_last * last, num_filters_last * mult, 2 if i < n_layers else 1, False, True, True))
        idx += 3
    plan.append((idx, num_filters_last * mult, 1, 1, True, False, False))
    return plan


class DiscOracle(torch.nn.Module):
    def __init__(self, state_dict, im_channel=1, num_filters_last=64, n_layers=3):
        super().__init__()
        self.plan = layer_plan(im_channel, num_filters_last, n_layers)
        self.p = torch.nn.ParameterDict({k.replace(".", "_"): torch.nn.Parameter(v.detach().double().cpu().clone()) for k, v in state_dict.items()})

    def grad(self, key):
        return self.p[key.replace(".", "_")].grad

    def forward(self, x):
        for idx, cin, cout, stride, bias, inorm, lrelu in self.plan:
            x = F.conv2d(x, self.p[f"model_{idx}_weight"], self.p[f"model_{idx}_bias"] if bias else None, stride=stride)
            if inorm:
                x = F.instance_norm(x, eps=1e-5)
            if lrelu:
                x = F.leaky_relu(x, 0.2)
        return x


def generator_gan_term(disc: DiscOracle, decoded: torch.Tensor):
    """-> (g_loss, d g_loss / d decoded)"""
    dec = decoded.double().clone().requires_grad_(True)
    g_loss = torch.mean(-disc(dec))
    (g,) = torch.autograd.grad(g_loss, dec)
    return g_loss.detach(), g


def discriminator_losses(disc: DiscOracle, imgs: torch.Tensor, decoded: torch.Tensor, current_step: int, r1_reg_weight: float, net_d_reg_every: int):
    """Accumulates the gradients into disc's parameters (.grad); -> dict of the loss values."""
    for p in disc.parameters():
        p.grad = None
    real, fake = disc(imgs.double()), disc(decoded.double().detach())
    d_loss_real, d_loss_fake = torch.mean(-real), torch.mean(fake)
    (0.5 * (d_loss_real + d_loss_fake)).backward()
    out = {"d_loss_real": d_loss_real.item(), "d_loss_fake": d_loss_fake.item()}
    if current_step % net_d_reg_every == 0:
        x = imgs.double().clone().requires_grad_(True)
        (grad_real,) = torch.autograd.grad(disc(x).sum(), x, create_graph=True)
        penalty = grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()
        r1 = r1_reg_weight / 2 * penalty * net_d_reg_every
        r1.backward()
        out["r1_penalty"], out["r1_loss"] = penalty.item(), r1.item()
    return out
