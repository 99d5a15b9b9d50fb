import numpy as np
import pytest
import torch

from tests.util import randn_bf16, randn_f32, rel_l2, max_abs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('rows,d', [(37, 128), (1000, 768), (513, 1024), (64, 2048), (5, 64)])
@pytest.mark.parametrize('masked', [False, True])
def test_layernorm_fwd_bwd(rows, d, masked):
    from m3p_amd import ops
    from oracle import ref_cpu as O
    x, xc = randn_bf16((rows, d), 1, 2.0)
    g, gc = randn_f32((d,), 2, 0.5)
    g += 1.0; gc += 1.0
    b, bc = randn_f32((d,), 3, 0.5)
    rm = None
    rmc = torch.ones(rows)
    if masked:
        rmc = (torch.from_numpy(np.random.RandomState(4).rand(rows)) > 0.3).float()
        rm = rmc.to(torch.uint8).cuda()
    y, mean, rstd = ops.layernorm_fwd(x, g, b, rm)
    xr = xc.clone().requires_grad_(True)
    gr = gc.clone().requires_grad_(True)
    br = bc.clone().requires_grad_(True)
    yr = O.layer_norm(xr, gr, br) * rmc[:, None]
    assert rel_l2(y.float(), yr) < 4e-3          # bf16 output rounding (2^-9 per element)
    assert max_abs(mean, xc.mean(-1)) < 1e-5
    # backward: dy = dy_a + dy_b
    dya, dyac = randn_bf16((rows, d), 5)
    dyb, dybc = randn_bf16((rows, d), 6)
    dg = torch.zeros(d, device='cuda'); db = torch.zeros(d, device='cuda'); dbias = torch.zeros(d, device='cuda')
    dx, dxd = ops.layernorm_bwd(dya, dyb, x, g, mean, rstd, rm, dg, db, dbias_drop=dbias)
    yr.backward(dyac + dybc)
    assert dxd is None
    assert rel_l2(dx.float(), xr.grad) < 6e-3
    assert rel_l2(dg, gr.grad) < 1e-4
    assert rel_l2(db, br.grad) < 1e-4
    assert rel_l2(dbias, dx.float().sum(0)) < 1e-5   # column sum of the bf16 dx it wrote


def test_layernorm_bwd_dropout_branch():
    from m3p_amd import ops, rng
    rows, d, p, seed = 300, 768, 0.1, 12345
    x, xc = randn_bf16((rows, d), 1)
    g, gc = randn_f32((d,), 2, 0.1); g += 1; gc += 1
    b, _ = randn_f32((d,), 3, 0.1)
    y, mean, rstd = ops.layernorm_fwd(x, g, b)
    dy, _ = randn_bf16((rows, d), 5)
    dg = torch.zeros(d, device='cuda'); db = torch.zeros(d, device='cuda'); dbias = torch.zeros(d, device='cuda')
    dx, dxd = ops.layernorm_bwd(dy, None, x, g, mean, rstd, None, dg, db, dbias_drop=dbias, want_drop=True, seed=seed, p_drop=p)
    keep = torch.from_numpy(rng.keep_mask(rows * d, seed, p, (rows, d)))
    exp = (dx.float().cpu() * keep / (1 - p)).to(torch.bfloat16)
    assert torch.equal(dxd.cpu(), exp)
    assert abs(float(keep.float().mean()) - (1 - p)) < 5e-3
    assert rel_l2(dbias, exp.float().sum(0)) < 1e-5


def test_layernorm_perf_smoke():
    from m3p_amd import ops
    rows, d = 41984, 768
    x, _ = randn_bf16((rows, d), 1)
    g, _ = randn_f32((d,), 2, 0.1); g += 1
    b, _ = randn_f32((d,), 3, 0.1)
    y, mean, rstd = ops.layernorm_fwd(x, g, b)
    dy, _ = randn_bf16((rows, d), 5)
    dg = torch.zeros(d, device='cuda'); db = torch.zeros(d, device='cuda'); dbias = torch.zeros(d, device='cuda')
    for name, fn in (('ln_fwd', lambda: ops.layernorm_fwd(x, g, b)),
                     ('ln_bwd+drop', lambda: ops.layernorm_bwd(dy, None, x, g, mean, rstd, None, dg, db, dbias_drop=dbias, want_drop=True, seed=3, p_drop=0.1))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        print('%s [%d x %d]: %.1f us' % (name, rows, d, e0.elapsed_time(e1) * 100))
