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


def encoder_keep_masks(model, step, B, T, R, p, p_attn):
    """The dropout keep masks the encoder kernels draw for forward pass number ``step`` of ``model``, rebuilt on
    the host with the NumPy twin of the device RNG (m3p_amd/rng.py) in the layout oracle.ref_cpu.jointfwd's
    ``keeps`` takes.  Element indices: image rows (r*B + b)*d + c, everything else (b*S + s)*d + c, attention
    probabilities ((b*H + h)*S + q)*S + k."""
    from m3p_amd import functional as Fn, rng
    S, d, H = R + T, model.dim, model.n_heads
    seed = lambda kind, i=0: rng.stream_seed(model.base_seed, step, Fn._site(kind, i))   # noqa: E731
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))                               # noqa: E731
    keeps = {'emb': t(rng.keep_mask(B * S * d, seed('emb'), p, (B, S, d)))}
    if R:
        keeps['img'] = t(rng.keep_mask(R * B * d, seed('img'), p, (R, B, d)).transpose(1, 0, 2))
    for i in range(model.n_layers):
        keeps[('attn_p', i)] = t(rng.keep_mask(B * H * S * S, seed('attn_p', i), p_attn, (B, H, S, S)))
        keeps[('attn_out', i)] = t(rng.keep_mask(B * S * d, seed('attn_out', i), p, (B, S, d)))
        keeps[('ffn', i)] = t(rng.keep_mask(B * S * d, seed('ffn', i), p, (B, S, d)))
    return keeps
