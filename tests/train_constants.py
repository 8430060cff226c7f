"""Constants shared by the training tests and by the golden-vector generators (oracle/gen_golden_train.py,
oracle/gen_golden_stage2.py import them from HERE): the stage-3 optimiser options of option/train_stage3_x8.yml:90-108, the
tensors whose gradients are stored in full, and the seeded +-1 projection every other gradient is reduced with."""
import torch

TRAIN_OPT = dict(lr_G=4e-4, beta1=0.9, beta2=0.99, T_period=[40000, 80000, 120000, 120000, 120000],
                 restarts=[40000, 120000, 240000, 360000], restart_weights=[1, 1, 1, 1], eta_min=1e-7,
                 rec_loss_factor=1, ref_loss_factor=0.001)   # option/train_stage3_x8.yml:90-108
FULL = ("conv_last.bias", "conv_first.bias", "refmaskconv3.weight", "ThreeDA.conv3D_1.weight", "ThreeDA.conv3D_1.bias",
        "align_module.flowdsconv0_1.weight", "align_module.L1_dcnpack.conv_offset.bias", "upconv3.bias",
        "feature_extraction.0.conv1.bias", "reffea_L2_conv1.bias", "align_module.cas_dcnpack.bias", "recon_trunk.9.conv2.bias")


def hash_name(name: str) -> int:
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def projection(name: str, numel: int) -> torch.Tensor:
    """Seeded +-1 vector: regenerated from the tensor name on both sides."""
    g = torch.Generator().manual_seed(abs(hash_name(name)) % (2 ** 31))
    return (torch.randint(0, 2, (numel,), generator=g).to(torch.float64) * 2 - 1)
