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
    cs.zero_()
    c = ops.gemm_nt_fp8(a8, w8, L.EPI_MUL, a_is_bf8=a_bf8, descale_a=da, descale_b=dw, aux=r, colsum=cs)
    assert rel_l2(c.float(), prod * r.float()) < 4e-3
    assert rel_l2(cs, c.float().sum(0)) < 1e-3


def test_quant_fp8_running_max_accumulates_over_launches_and_shapes():
    from m3p_amd import ops
    amax = torch.zeros(1, device='cuda')
    worst = 0.0
    for seed, shape, s in ((1, (8, 128), 0.1), (2, (4096, 1024), 1.0), (3, (22784, 4096), 2.0), (4, (256, 256), 0.5)):
        x = _bf16(shape, seed, s)
        ops.quant_fp8(x, amax=amax)
        worst = max(worst, float(x.float().abs().max()))
        assert float(amax) == worst


def test_gemm_nt_fp8_rejects_what_it_does_not_take():
    from m3p_amd import ops, lib as L
    a8 = torch.zeros((300, 128), dtype=torch.uint8, device='cuda')
    w8 = torch.zeros((256, 128), dtype=torch.uint8, device='cuda')
    with pytest.raises(L.M3PError):
        ops.gemm_nt_fp8(a8, w8)                       # M % 256 != 0


def _train_curve(cfg, fp8, steps, lr='0.001,warmup_updates=8', dropout=0.1, n_batches=4):
    """Loss curve of `steps` pre-training steps (MLM + ITM, Adam inverse-sqrt + clip) on cycling synthetic batches."""
    import bench
    from m3p_amd import synth
    np.random.seed(11)
    torch.manual_seed(11)
    trainer, tup0 = bench.build(cfg, dropout, 1, 0, 0, fp8=fp8, lr=lr)
    tups = [tup0]
    for s in range(1, n_batches):
        b = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'], seed=2000 + s, ragged=False)
        img = b['x_img'].transpose(0, 1).contiguous().cuda()
        loc = b['image_loc'].transpose(0, 1).contiguous().cuda()
        tups.append(((b['x'].cuda(), b['lengths'].cuda(), b['x_labels']),
                     (img, torch.ones(cfg['B'], cfg['R'], dtype=torch.long, device='cuda'), loc, None, b['pos_labels'].tolist(), None, None)))
    for k in range(steps):
        trainer.pretrain_under_step(tups[k % n_batches], 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0)
        trainer.n_iter += 1
    torch.cuda.synchronize()
    mlm = torch.stack([v.float() for v in trainer.stats['CMLM-google']]).cpu().numpy()
    itm = torch.stack([v.float() for v in trainer.stats['t2i-google']]).cpu().numpy() if 't2i-google' in trainer.stats else None
    return trainer, mlm, itm


def _curve_check(cfg, steps):
    _, mlm16, itm16 = _train_curve(cfg, False, steps)
    tr8, mlm8, itm8 = _train_curve(cfg, True, steps)
    assert tr8.model.fp8 and tr8.model.fp8_state().weights, 'the fp8 path did not run'
    assert np.isfinite(mlm8).all()
    # the run learns something (else "equal curves" says nothing) ...
    assert mlm16[-4:].mean() < mlm16[:4].mean() - 0.3, (mlm16[:4], mlm16[-4:])
    # ... and the fp8 curve follows the bf16 one within 2 %, step by step on a 4-step running mean
    # (single-step losses of different batches are noisy in both runs) and over the whole run
    run = lambda v: np.convolve(v, np.ones(4) / 4, mode='valid')      # noqa: E731
    rel = np.abs(run(mlm8) - run(mlm16)) / run(mlm16)
    assert rel.max() < 0.02, (rel.max(), mlm16, mlm8)
    assert abs(mlm8.mean() - mlm16.mean()) / mlm16.mean() < 0.01
    if itm16 is not None:
        # the image-text matching loss (BCE around ln 2 = 0.69 on random pairs: nothing to learn in the synthetic batch) spikes
        # in BOTH runs when Adam overshoots at this learning rate, to a different height each time - in a bf16 run too if a
        # weight changes in its last bit.  tools/fp8_curve_stats.py over repeated runs: the 4-step mean of the step's total
        # loss differs by 0.9-1.0 % between the fp8 and the bf16 run, all of it from those spikes (the MLM term alone: 0.2 %),
        # and about one run in ten takes another spike pattern (1.9 %).  The total loss - the quantity the step descends on -
        # is therefore compared on an 8-step mean, the ITM term on its mean over the run.
        run8 = lambda v: np.convolve(v, np.ones(8) / 8, mode='valid')      # noqa: E731
        tot16, tot8 = run8(mlm16 + itm16), run8(mlm8 + itm8)
        assert (np.abs(tot8 - tot16) / tot16).max() < 0.02, (tot16, tot8)
        assert abs(itm8.mean() - itm16.mean()) / itm16.mean() < 0.10, (itm16, itm8)
    return mlm16, mlm8


def test_fp8_training_loss_curve_follows_bf16_small():
    """4 layers / 256 wide, 16 regions + 48 tokens x 16 sequences (M = 1024 = 4 row tiles), V = 8192, 40 steps."""
    cfg = dict(emb_dim=256, n_heads=4, n_layers=4, n_words=8192, T=48, R=16, B=16, n_pred=8)
    _curve_check(cfg, 40)


def test_fp8_training_loss_curve_follows_bf16_cfg4_geometry():
    """BASELINE configs[3] geometry (1024 wide, 16 heads, 100 regions + 256 tokens, batch 64, V = 250 002) with 6 of its
    24 layers so that two 24-step runs fit the suite's time budget; the full 24-layer step runs in bench.py --config cfg4 --fp8."""
    from m3p_amd import synth
    cfg = dict(synth.CONFIGS['cfg4'])
    cfg['n_layers'] = 6
    _curve_check(cfg, 24)


def test_fp8_full_depth_cfg4_three_steps():
    """BASELINE configs[3] at its real depth: 24 layers / 1024 wide / 16 heads, 100 regions + 256 tokens, batch 64, V = 250 002
    (M = 64 x 356 = 89 row tiles).  Three optimizer steps in bf16 and with the fp8 layer GEMMs from the same initial weights and
    batches: the total loss of every step within 2 %, every fp8 site of all 24 layers exercised."""
    from m3p_amd import synth
    cfg = dict(synth.CONFIGS['cfg4'])
    assert cfg['n_layers'] == 24 and cfg['B'] == 64
    _, mlm16, itm16 = _train_curve(cfg, False, 3, n_batches=3)
    tr8, mlm8, itm8 = _train_curve(cfg, True, 3, n_batches=3)
    st8 = tr8.model.fp8_state()
    assert tr8.model.fp8 and len({k[0] for k in st8.weights}) == 24, 'not every layer ran on 8-bit weights'
    tot16, tot8 = mlm16 + itm16, mlm8 + itm8
    assert np.isfinite(tot8).all()
    assert (np.abs(tot8 - tot16) / tot16).max() < 0.02, (tot16, tot8)


def test_fp8_needs_whole_row_tiles():
    import bench
    cfg = dict(emb_dim=256, n_heads=4, n_layers=2, n_words=4096, T=40, R=10, B=6, n_pred=4)      # M = 300
    trainer, tup = bench.build(cfg, 0.0, 1, 0, 0, fp8=True)
    with pytest.raises(AssertionError, match='256-row tiles'):
        trainer.pretrain_under_step(tup, 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0)


def test_gelu_pass_with_8bit_copy_equals_gelu_then_quant():
    from m3p_amd import ops
    u = _bf16((1000, 512), 5, 2.0)
    scale = torch.tensor([24.0], device='cuda')
    am1, am2 = torch.zeros(1, device='cuda'), torch.zeros(1, device='cuda')
    h, h8 = ops.gelu_fwd_q8(u.clone(), scale, am1)
    h_ref = ops.gelu_fwd(u.clone())
    q_ref = ops.quant_fp8(h_ref, scale=scale, amax=am2)
    assert torch.equal(h, h_ref) and torch.equal(h8, q_ref) and float(am1) == float(am2) == float(h_ref.float().abs().max())


def test_byte_derivative_training_curve_follows_the_stored_pre_activation(monkeypatch):
    """Round 4 keeps gelu'(u) as ONE byte per element (256 levels, |error| <= 2.5e-3) instead of the bf16 pre-activation.
    The same 40-step run as above - 4 layers / 256 wide, M = 1024 rows, dropout 0.1, Adam + clip - with the byte form
    (lin1's epilogue computes GELU and the byte, dU decodes it) against the round-3 form (bias epilogue, GELU pass, derivative
    recomputed from the stored u): the byte must actually be in use, and the MLM curve must follow within 1 % on a 4-step
    running mean and 0.5 % over the run - half the window the 8-bit GEMM path is allowed."""
    from m3p_amd import functional as Fn, ops
    cfg = dict(emb_dim=256, n_heads=4, n_layers=4, n_words=8192, T=48, R=16, B=16, n_pred=8)
    assert ops.gq_eligible(cfg['B'] * (cfg['T'] + cfg['R']), 4 * cfg['emb_dim'])
    calls = []
    real = ops.gemm_nt

    def spy(a, w, epilogue=0, **kw):
        calls.append(epilogue)
        return real(a, w, epilogue, **kw)
    monkeypatch.setattr(ops, 'gemm_nt', spy)
    monkeypatch.setattr(Fn, '_GELU_BYTE_GRAD', 0)
    _, ref, _ = _train_curve(cfg, False, 40)
    from m3p_amd import lib as L
    assert L.EPI_MULQ not in calls and L.EPI_DGELU in calls
    calls.clear()
    monkeypatch.setattr(Fn, '_GELU_BYTE_GRAD', 2)
    _, byte, _ = _train_curve(cfg, False, 40)
    assert L.EPI_BIAS_GELUQ in calls and L.EPI_MULQ in calls and L.EPI_DGELU not in calls
    assert np.isfinite(byte).all() and ref[-4:].mean() < ref[:4].mean() - 0.3
    run = lambda v: np.convolve(v, np.ones(4) / 4, mode='valid')      # noqa: E731
    rel = np.abs(run(byte) - run(ref)) / run(ref)
    assert rel.max() < 0.01, (rel.max(), ref, byte)
    assert abs(byte.mean() - ref.mean()) / ref.mean() < 0.005


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,K', [(1024, 512, 64), (2048, 1024, 256)])
def test_byte_epilogues_leave_the_8bit_copy_of_their_output(M, N, K):
    """Round 6: lin1 + GELU + byte (EPI_BIAS_GELUQ) and the byte-decode data gradient (EPI_MULQ) also leave the e4m3 / e5m2 copy
    of their output for the fp8 product that consumes it (M3PEpilogue::out8): the bf16 output and the codes are bit for bit what
    the launch without the copy leaves; the copy is the OCP cast of (output * scale) within one 8-bit rounding of the cast
    of the stored bf16 output (it is taken from the fp32 value), saturating; the running maximum is raised to max |output|."""
    from m3p_amd import ops, lib as L
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    a = torch.randn((M, K), device='cuda', generator=g).to(torch.bfloat16)
    w = (torch.randn((N, K), device='cuda', generator=g) * (1.5 / math.sqrt(K))).to(torch.bfloat16)
    bias = torch.randn((N,), device='cuda', generator=g)
    # forward: h, codes, h8
    q0 = torch.empty(M * N, dtype=torch.uint8, device='cuda')
    h0 = ops.gemm_nt(a, w, L.EPI_BIAS_GELUQ, bias=bias, out2=q0)
    scale = torch.tensor([37.0], device='cuda')           # (large enough that the biggest activations saturate at 448)
    amax = torch.zeros(1, device='cuda')
    q1 = torch.empty(M * N, dtype=torch.uint8, device='cuda')
    h8 = torch.empty((M, N), dtype=torch.uint8, device='cuda')
    h1 = ops.gemm_nt(a, w, L.EPI_BIAS_GELUQ, bias=bias, out2=q1, out8=h8, scale8=scale, amax8=amax)
    assert torch.equal(h1, h0) and torch.equal(q1, q0)
    ref8 = (h0.float() * 37.0).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    got8 = h8.view(torch.float8_e4m3fn).float()
    assert torch.isfinite(got8).all() and float(got8.abs().max()) <= 448.0
    # e4m3 has 3 mantissa bits: one code apart is <= 12.5 % relative; the fp32 source may round to the neighbour of the bf16's
    close = (got8 - ref8).abs() <= 0.126 * ref8.abs().clamp_min(2.0 ** -9 * 37.0)
    assert float(close.float().mean()) > 0.9999 and float((got8 == ref8).float().mean()) > 0.97
    assert abs(float(amax) - float(h0.float().abs().max())) <= 2.0 ** -7 * float(amax)
    # backward: dU, column sums, dU8 in e5m2
    dy = (torch.randn((M, K), device='cuda', generator=g) * 0.01).to(torch.bfloat16)
    cs0, cs1 = torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda')
    du0 = ops.gemm_nt(dy, w, L.EPI_MULQ, aux=q0, colsum=cs0)
    s2 = torch.tensor([4096.0], device='cuda')
    amax2 = torch.zeros(1, device='cuda')
    du8 = torch.empty((M, N), dtype=torch.uint8, device='cuda')
    du1 = ops.gemm_nt(dy, w, L.EPI_MULQ, aux=q0, colsum=cs1, out8=du8, scale8=s2, amax8=amax2, out8_bf8=True)
    assert torch.equal(du1, du0) and rel_l2(cs1, cs0) < 1e-6
    ref = (du0.float() * 4096.0).clamp(-57344, 57344).to(torch.float8_e5m2).float()
    got = du8.view(torch.float8_e5m2).float()
    close = (got - ref).abs() <= 0.26 * ref.abs().clamp_min(2.0 ** -9 * 4096.0 * 1e-3)
    assert float(close.float().mean()) > 0.9999 and float((got == ref).float().mean()) > 0.95
    assert abs(float(amax2) - float(du0.float().abs().max())) <= 2.0 ** -7 * float(amax2)
