import numpy as np
import torch

BF16 = torch.bfloat16


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b):
    return float((torch.as_tensor(a).detach().double().cpu() - torch.as_tensor(b).detach().double().cpu()).abs().max())


def randn_bf16(shape, seed, scale=1.0, device='cuda'):
    """bf16-representable random tensor: returns (device bf16 tensor, CPU fp32 copy of the same values)."""
    rs = np.random.RandomState(seed)
    t = torch.from_numpy(rs.standard_normal(shape).astype(np.float32) * scale).to(BF16)
    return t.to(device), t.float()


def randn_f32(shape, seed, scale=1.0, device='cuda'):
    rs = np.random.RandomState(seed)
    t = torch.from_numpy(rs.standard_normal(shape).astype(np.float32) * scale)
    return t.to(device), t.clone()
