"""fp8 GEMM path (BASELINE configs[3]: "fp8 MFMA GEMMs"): the quantisation kernel against PyTorch's OCP float8 casts, the
K = 128 MFMA kernel against an fp32 product of the very same 8-bit operands (so only accumulation order differs), every
epilogue it carries, and the per-tensor scale plumbing."""
import math

import numpy as np
import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu


def _bf16(shape, seed, scale=1.0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return (torch.randn(shape, device='cuda', generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize('bf8', [False, True])
def test_quant_fp8_matches_torch_casts(bf8):
    from m3p_amd import ops
    x = _bf16((300, 512), 1, 3.0)
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-4, -2e-3, 500.0, -1000.0, 447.9, 60000.0], device='cuda').to(torch.bfloat16)
    scale = torch.tensor([0.75], device='cuda')
    amax = torch.zeros(1, device='cuda')
    q = ops.quant_fp8(x, scale=scale, amax=amax, bf8=bf8)
    dt = torch.float8_e5m2 if bf8 else torch.float8_e4m3fn
    lim = 57344.0 if bf8 else 448.0
    ref = (x.float() * 0.75).clamp(-lim, lim).to(dt)
    assert torch.equal(q.view(dt).float(), ref.float())
    assert float(amax) == float(x.float().abs().max())
    q1 = ops.quant_fp8(x, bf8=bf8)                # no scale, no amax
    assert torch.equal(q1.view(dt).float(), x.float().clamp(-lim, lim).to(dt).float())


@pytest.mark.parametrize('M,N,K', [(256, 256, 128), (1024, 768, 768), (2560, 3072, 1024), (4096, 1024, 4096)])
@pytest.mark.parametrize('a_bf8', [False, True])
def test_gemm_nt_fp8_vs_fp32_product_of_the_same_operands(M, N, K, a_bf8):
    from m3p_amd import ops, rng, lib as L
    a, w = _bf16((M, K), 2, 1.0), _bf16((N, K), 3, 0.05)
    sa = torch.tensor([16.0], device='cuda')
    sw = torch.tensor([64.0], device='cuda')
    a8 = ops.quant_fp8(a, scale=sa, bf8=a_bf8)
    w8 = ops.quant_fp8(w, scale=sw)
    da, dw = 1.0 / sa, 1.0 / sw
    af = a8.view(torch.float8_e5m2 if a_bf8 else torch.float8_e4m3fn).float() / 16.0
    wf = w8.view(torch.float8_e4m3fn).float() / 64.0
    prod = af @ wf.t()
    # the quantised product is close to the bf16 one (sanity of the scale plumbing) ...
    assert rel_l2(prod, a.float() @ w.float().t()) < (0.2 if a_bf8 else 0.08)
    # ... and the kernel reproduces the product of the 8-bit operands to bf16 output rounding
    c = ops.gemm_nt_fp8(a8, w8, L.EPI_NONE, a_is_bf8=a_bf8, descale_a=da, descale_b=dw)
    assert rel_l2(c.float(), prod) < 4e-3
    bias = torch.randn(N, device='cuda')
    r = _bf16((M, N), 4)
    c = ops.gemm_nt_fp8(a8, w8, L.EPI_BIAS, a_is_bf8=a_bf8, descale_a=da, descale_b=dw, bias=bias, scale_cols=N // 3, scale=0.125)
    ref = prod + bias
    ref[:, :N // 3] *= 0.125
    assert rel_l2(c.float(), ref) < 4e-3
    c = ops.gemm_nt_fp8(a8, w8, L.EPI_RES, a_is_bf8=a_bf8, descale_a=da, descale_b=dw, aux=r)
    assert rel_l2(c.float(), prod + r.float()) < 4e-3
    c = ops.gemm_nt_fp8(a8, w8, L.EPI_BIAS_DROP_RES, a_is_bf8=a_bf8, descale_a=da, descale_b=dw, bias=bias, aux=r, seed=99, p_drop=0.1)
    keep = torch.from_numpy(rng.keep_mask(M * N, 99, 0.1, (M, N))).cuda()
    assert rel_l2(c.float(), (prod + bias) * keep / 0.9 + r.float()) < 4e-3
    cs = torch.zeros(N, device='cuda')
    c = ops.gemm_nt_fp8(a8, w8, L.EPI_DGELU, a_is_bf8=a_bf8, descale_a=da, descale_b=dw, aux=r, colsum=cs)
    x = r.double()
    dg = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    assert rel_l2(c.float(), prod.double() * dg) < 6e-3
    assert rel_l2(cs, c.float().sum(0)) < 1e-3


def test_gemm_nt_fp8_rejects_what_it_does_not_take():
    from m3p_amd import ops, lib as L
    a8 = torch.zeros((300, 128), dtype=torch.uint8, device='cuda')
    w8 = torch.zeros((256, 128), dtype=torch.uint8, device='cuda')
    with pytest.raises(L.M3PError):
        ops.gemm_nt_fp8(a8, w8)                       # M % 256 != 0
