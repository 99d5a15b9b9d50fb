import math

import numpy as np
import pytest
import torch

from tests.util import randn_bf16, randn_f32, rel_l2, max_abs

pytestmark = pytest.mark.gpu

SHAPES = [(128, 128, 64), (592, 384, 128), (300, 1000, 128), (41, 130, 192), (1024, 768, 768), (256, 3072, 768),
          (520, 768, 3072), (2048 + 40, 1000, 128), (4096, 2304, 768), (9216, 768, 2048), (70000, 128, 64)]


def _tol(K):
    return 4e-3


@pytest.mark.parametrize('M,N,K', SHAPES)
def test_gemm_nt_none_and_bias(M, N, K):
    from m3p_amd import ops, lib as L
    a, ac = randn_bf16((M, K), 1)
    w, wc = randn_bf16((N, K), 2, 0.05)
    bias, bc = randn_f32((N,), 3)
    ldc = (N + 3) // 4 * 4
    out = torch.full((M, ldc), 7.0, dtype=torch.bfloat16, device='cuda')
    c = ops.gemm_nt(a, w, L.EPI_NONE, out=out, n=N)
    ref = ac @ wc.t()
    assert rel_l2(c[:, :N].float(), ref) < _tol(K)
    if ldc > N:
        assert bool((c[:, N:] == 7.0).all()), 'wrote outside the N range'
    c = ops.gemm_nt(a, w, L.EPI_BIAS, bias=bias, scale_cols=N // 3, scale=0.125, out=out, n=N)
    ref = ac @ wc.t() + bc
    ref[:, :N // 3] *= 0.125
    assert rel_l2(c[:, :N].float(), ref) < _tol(K)


def test_gemm_nt_transpose_detecting():
    """A = I-like, asymmetric W: catches a swapped result map."""
    from m3p_amd import ops, lib as L
    M = N = K = 128
    a = torch.eye(M, dtype=torch.bfloat16, device='cuda')
    w = (torch.arange(N * K, dtype=torch.float32).view(N, K) % 37 - 18).to(torch.bfloat16).cuda()
    c = ops.gemm_nt(a, w, L.EPI_NONE)
    assert torch.equal(c.float().cpu(), w.float().cpu().t())


def test_gemm_nt_bias_gelu():
    from m3p_amd import ops, lib as L
    from oracle import ref_cpu as O
    M, N, K = 300, 512, 128
    a, ac = randn_bf16((M, K), 1)
    w, wc = randn_bf16((N, K), 2, 0.1)
    bias, bc = randn_f32((N,), 3)
    u = torch.empty((M, N), dtype=torch.bfloat16, device='cuda')
    h = ops.gemm_nt(a, w, L.EPI_BIAS_GELU, bias=bias, out2=u)
    uref = ac @ wc.t() + bc
    assert rel_l2(u.float(), uref) < 4e-3
    assert rel_l2(h.float(), O.gelu_erf(u.float().cpu())) < 4e-3   # gelu of the stored pre-activation
    assert rel_l2(h.float(), O.gelu_erf(uref)) < 8e-3


@pytest.mark.parametrize('p', [0.0, 0.1])
def test_gemm_nt_bias_dropout_residual(p):
    from m3p_amd import ops, rng, lib as L
    M, N, K, seed = 333, 768, 256, 777
    a, ac = randn_bf16((M, K), 1)
    w, wc = randn_bf16((N, K), 2, 0.1)
    bias, bc = randn_f32((N,), 3)
    r, rc = randn_bf16((M, N), 4)
    c = ops.gemm_nt(a, w, L.EPI_BIAS_DROP_RES, bias=bias, aux=r, seed=seed, p_drop=p)
    y = ac @ wc.t() + bc
    if p > 0:
        keep = torch.from_numpy(rng.keep_mask(M * N, seed, p, (M, N)))
        y = y * keep / (1 - p)
    assert rel_l2(c.float(), y + rc) < 4e-3


def test_gemm_nt_res_and_dgelu():
    from m3p_amd import ops, lib as L
    M, N, K = 260, 512, 128
    a, ac = randn_bf16((M, K), 1)
    w, wc = randn_bf16((N, K), 2, 0.1)
    r, rc = randn_bf16((M, N), 4)
    c = ops.gemm_nt(a, w, L.EPI_RES, aux=r, alpha=0.5)
    assert rel_l2(c.float(), 0.5 * (ac @ wc.t()) + rc) < 4e-3
    cs = torch.zeros(N, device='cuda')
    c = ops.gemm_nt(a, w, L.EPI_DGELU, aux=r, colsum=cs)
    x = rc.double().requires_grad_(True)
    g = 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))
    g.sum().backward()
    ref = (ac @ wc.t()).double() * x.grad
    assert rel_l2(c.float(), ref) < 4e-3
    assert rel_l2(cs, c.float().sum(0)) < 1e-5


@pytest.mark.parametrize('M,N,K', [(64, 128, 128), (592, 384, 128), (1000, 100, 72), (4100, 768, 768), (131, 40, 264),
                                   (8200, 3072, 768), (8192, 768, 768), (4288, 2304, 768), (4096, 1000, 72),
                                   (16384, 256, 128), (4864, 5008, 768), (4096, 3072, 768), (8192, 768, 3072),
                                   (4160, 1024, 1024)])
def test_gemm_wgrad(M, N, K):
    from m3p_amd import ops
    dy, dyc = randn_bf16((M, (N + 7) // 8 * 8), 1)
    x, xc = randn_bf16((M, (K + 7) // 8 * 8), 2)
    dw = torch.ones((N, K), dtype=torch.float32, device='cuda')
    ops.gemm_wgrad(dy, x, dw, alpha=0.5, n=N, k=K)
    ref = 1.0 + 0.5 * (dyc[:, :N].double().t() @ xc[:, :K].double())
    assert rel_l2(dw, ref) < 1e-5


def test_gemm_perf_smoke():
    """Not a benchmark: prints achieved TF for the cfg2 shapes so the GPU log shows them."""
    from m3p_amd import ops, lib as L
    M = 41984
    for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
        a, _ = randn_bf16((M, K), 1)
        w, _ = randn_bf16((N, K), 2, 0.05)
        out = torch.empty((M, N), dtype=torch.bfloat16, device='cuda')
        for _ in range(3):
            ops.gemm_nt(a, w, L.EPI_NONE, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm_nt(a, w, L.EPI_NONE, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('gemm_nt M=%d N=%d K=%d: %.3f ms  %.1f TF' % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
        dw = torch.zeros((N, K), dtype=torch.float32, device='cuda')
        dy = out
        for _ in range(2):
            ops.gemm_wgrad(dy, a, dw)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            ops.gemm_wgrad(dy, a, dw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('gemm_wgrad M=%d N=%d K=%d: %.3f ms  %.1f TF' % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))


@pytest.mark.parametrize('M,N,K', [(300, 768, 4096), (4864, 768, 25024), (1000, 130, 640)])
def test_gemm_nt_streamk(M, N, K):
    from m3p_amd import ops
    a, ac = randn_bf16((M, K), 1, 0.5)
    w, wc = randn_bf16((N, K), 2, 0.05)
    out = torch.ones((M, N), dtype=torch.float32, device='cuda')
    ops.gemm_nt_streamk(a, w, out, alpha=0.5)
    ref = 1.0 + 0.5 * (ac.double() @ wc.double().t())
    assert rel_l2(out, ref) < 1e-5


@pytest.mark.parametrize('variant', [2, 6])
@pytest.mark.parametrize('M,N,K', [(1024, 768, 3072), (1024 + 40, 1000, 128), (2560, 512, 2304), (1300, 2304 + 8, 1536),
                                   (1280, 2304, 768), (4096, 768, 768), (8192, 3072, 768), (1280, 640, 64), (33024, 384, 192)])
def test_gemm_nt_256x256_kernels(M, N, K, variant):
    """The 256x256-tile kernels (variant 2: four waves of 128x128; variant 6: eight waves of 128x64) against the fp32
    product, every epilogue, full and ragged tiles; forced through the developer switch so every shape runs them
    (shapes a kernel does not take fall through to the next one, as in production)."""
    from m3p_amd import ops, rng, lib as L
    lib = L.load()
    a, ac = randn_bf16((M, K), 1)
    w, wc = randn_bf16((N, K), 2, 0.05)
    bias, bc = randn_f32((N,), 3)
    r, rc = randn_bf16((M, N), 4)
    prod = ac @ wc.t()
    lib.m3p_debug_set_variant(variant)
    try:
        c = ops.gemm_nt(a, w, L.EPI_NONE)
        assert rel_l2(c.float(), prod) < 4e-3
        c = ops.gemm_nt(a, w, L.EPI_BIAS, bias=bias, scale_cols=N // 3, scale=0.125)
        ref = prod + bc
        ref[:, :N // 3] *= 0.125
        assert rel_l2(c.float(), ref) < 4e-3
        u = torch.empty((M, N), dtype=torch.bfloat16, device='cuda')
        h = ops.gemm_nt(a, w, L.EPI_BIAS_GELU, bias=bias, out2=u)
        assert rel_l2(u.float(), prod + bc) < 4e-3
        uf = u.float().cpu().double()
        assert rel_l2(h.float(), (0.5 * uf * (1 + torch.erf(uf / math.sqrt(2)))).float()) < 4e-3
        c = ops.gemm_nt(a, w, L.EPI_BIAS_DROP_RES, bias=bias, aux=r, seed=99, p_drop=0.1)
        keep = torch.from_numpy(rng.keep_mask(M * N, 99, 0.1, (M, N)))
        assert rel_l2(c.float(), (prod + bc) * keep / 0.9 + rc) < 4e-3
        c = ops.gemm_nt(a, w, L.EPI_RES, aux=r, alpha=0.5)
        assert rel_l2(c.float(), 0.5 * prod + rc) < 4e-3
        cs = torch.zeros(N, device='cuda')
        c = ops.gemm_nt(a, w, L.EPI_DGELU, aux=r, colsum=cs)
        x = rc.double()
        dg = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
        assert rel_l2(c.float(), prod.double() * dg) < 4e-3
        assert rel_l2(cs, c.float().sum(0)) < 1e-4
        cs = torch.zeros(N, device='cuda')
        c = ops.gemm_nt(a, w, L.EPI_MUL, aux=r, colsum=cs)
        assert rel_l2(c.float(), prod * rc) < 4e-3
        assert rel_l2(cs, c.float().sum(0)) < 1e-4
    finally:
        lib.m3p_debug_set_variant(1)


@pytest.mark.parametrize('M,N,K', [(1, 768, 768), (5, 256, 128), (16, 2304, 768), (33, 1000, 3072), (64, 768, 3072),
                                   (100, 3072, 768), (128, 1536, 1024), (32, 250002, 768), (128, 20003, 256)])
def test_gemm_nt_skinny_rows(M, N, K):
    """M <= 128 rows (a decoding step, head GEMMs on [CLS] rows): the fragment-from-global kernel, split-K for small N and
    one wave per 16 columns for the vocabulary width; every epilogue it takes, ragged N, against the fp32 product."""
    from m3p_amd import ops, rng, lib as L
    a, ac = randn_bf16((M, K), 11)
    w, wc = randn_bf16((N, K), 12, 0.05)
    bias, bc = randn_f32((N,), 13)
    r, rc = randn_bf16((M, N), 14)
    prod = ac @ wc.t()
    if N % 8:       # a row pitch must be a multiple of 4 elements: odd widths (the vocabulary) live in padded buffers
        NP = (N + 63) // 64 * 64
        buf = torch.zeros((M, NP), dtype=torch.bfloat16, device='cuda')
        ops.gemm_nt(a, w, L.EPI_BIAS, bias=bias, out=buf, n=N)
        assert rel_l2(buf[:, :N].float(), prod + bc) < 4e-3 and float(buf[:, N:].abs().max()) == 0.0
        return
    c = ops.gemm_nt(a, w, L.EPI_NONE)
    assert rel_l2(c.float(), prod) < 4e-3
    c = ops.gemm_nt(a, w, L.EPI_BIAS, bias=bias, scale_cols=N // 3, scale=0.125)
    ref = prod + bc
    ref[:, :N // 3] *= 0.125
    assert rel_l2(c.float(), ref) < 4e-3
    u = torch.empty((M, N), dtype=torch.bfloat16, device='cuda')
    h = ops.gemm_nt(a, w, L.EPI_BIAS_GELU, bias=bias, out2=u)
    assert rel_l2(u.float(), prod + bc) < 4e-3
    uf = u.float().cpu().double()
    assert rel_l2(h.float(), (0.5 * uf * (1 + torch.erf(uf / math.sqrt(2)))).float()) < 4e-3
    c = ops.gemm_nt(a, w, L.EPI_BIAS_DROP_RES, bias=bias, aux=r, seed=99, p_drop=0.1)
    keep = torch.from_numpy(rng.keep_mask(M * N, 99, 0.1, (M, N)))
    assert rel_l2(c.float(), (prod + bc) * keep / 0.9 + rc) < 4e-3
    c = ops.gemm_nt(a, w, L.EPI_RES, aux=r, alpha=0.5)
    assert rel_l2(c.float(), 0.5 * prod + rc) < 4e-3
    # a strided output (the vocabulary logits live in a padded buffer) and a row-sliced weight (k | v rows of the fused matrix)
    if N >= 512:
        buf = torch.zeros((M, N + 64), dtype=torch.bfloat16, device='cuda')
        ops.gemm_nt(a, w, L.EPI_BIAS, bias=bias, out=buf, n=N)
        assert rel_l2(buf[:, :N].float(), prod + bc) < 4e-3 and float(buf[:, N:].abs().max()) == 0.0
        c = ops.gemm_nt(a, w[256:], L.EPI_NONE)
        assert rel_l2(c.float(), prod[:, 256:]) < 4e-3


@pytest.mark.parametrize('M,N,K,kv', [(300, 768, 4096, 4096), (4864, 768, 25024, 25002), (1000, 130, 640, 601)])
def test_gemm_nn_streamk(M, N, K, kv):
    """Cf += alpha * A[M,K] W[K,N] with W row-major in the contraction index (the vocabulary data gradient reads the
    embedding matrix in place); rows kv..K of W do not exist, the matching columns of A are zero."""
    from m3p_amd import ops
    a, ac = randn_bf16((M, K), 1)
    a[:, kv:] = 0
    ac[:, kv:] = 0
    w, wc = randn_bf16((kv, (N + 7) // 8 * 8), 2, 0.05)
    w = w[:, :N] if N % 8 == 0 else w
    out = torch.ones((M, N), dtype=torch.float32, device='cuda')
    if N % 8 == 0:
        ops.gemm_nn_streamk(a, w, out, alpha=0.5)
        ref = 1.0 + 0.5 * (ac[:, :kv].double() @ wc[:, :N].double())
        assert rel_l2(out, ref) < 1e-5
    else:
        wv = w[:, :N]                       # view with pitch (N + 7) // 8 * 8
        ops.gemm_nn_streamk(a, wv, out, alpha=0.5)
        ref = 1.0 + 0.5 * (ac[:, :kv].double() @ wc[:, :N].double())
        assert rel_l2(out, ref) < 1e-5



@pytest.mark.parametrize('M,N,K', [(256, 256, 4096), (512, 768, 8192), (4864, 768, 250112)])
def test_gemm_nn_on_the_four_wave_kernel(M, N, K):
    """C[M,N] += alpha * A[M,K] x W[K,N] with A contraction-contiguous and W row-major over the contraction (the vocabulary
    data gradient dH = dlogits x E): the four-wave (tile, K-chunk) kernel with its first operand staged as an NT panel,
    against an fp64 product in slices and against the stream-K form; accumulates into a non-zero C."""
    from m3p_amd import ops
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    a = (torch.randn((M, K), device='cuda', generator=g) * 0.05).to(torch.bfloat16)
    w = torch.randn((K, N), device='cuda', generator=g).to(torch.bfloat16)
    out = torch.full((M, N), 0.5, device='cuda')
    ops.gemm_nn(a, w, out, alpha=2.0)
    ref = torch.full((M, N), 0.5, dtype=torch.float64, device='cuda')
    for k0 in range(0, K, 16384):
        ref += 2.0 * (a[:, k0:k0 + 16384].double() @ w[k0:k0 + 16384].double())
    assert rel_l2(out.double(), ref) < 1e-5
    out2 = torch.full((M, N), 0.5, device='cuda')
    ops.gemm_nn_streamk(a, w, out2, alpha=2.0)
    assert rel_l2(out.double(), out2.double()) < 1e-5
    # a second operand shorter than the contraction goes to the stream-K form unless the caller vouches for the rows behind it
    out3 = torch.zeros((M, N), device='cuda')
    a2 = a.clone(); a2[:, K - 64:] = 0
    ops.gemm_nn(a2, w[:K - 64], out3)
    assert rel_l2(out3.double(), a2[:, :K - 64].double() @ w[:K - 64].double()) < 1e-5


def _ref_on_gpu(a, w):
    """fp32 product of the bf16 operands with PyTorch on the device (the checker at sizes a CPU product takes minutes)."""
    return a.float() @ w.float().t()


@pytest.mark.parametrize('M', [41984, 167936])       # B = 256 (configs[1]) and B = 1024 (per-GPU share of configs[2]) x S = 164
@pytest.mark.parametrize('N,K', [(2304, 768), (768, 768), (3072, 768), (768, 3072)])
def test_gemm_nt_at_the_benchmarked_sizes(M, N, K):
    """The persistent four-wave / ring kernels at the exact (M, N, K) the cfg2 step launches them with, every epilogue
    the layer uses, against a plain fp32 product: the per-kernel parity shapes above stop at M = 4100."""
    from m3p_amd import ops, lib as L
    if M > 41984 and (N, K) != (3072, 768):
        pytest.skip('one shape at the 1024-sequence row count')
    g = torch.Generator(device='cuda').manual_seed(M + N)
    a = (torch.randn((M, K), device='cuda', generator=g)).to(torch.bfloat16)
    w = (torch.randn((N, K), device='cuda', generator=g) * 0.05).to(torch.bfloat16)
    r = (torch.randn((M, N), device='cuda', generator=g)).to(torch.bfloat16)
    bias = torch.randn((N,), device='cuda', generator=g)
    ref = _ref_on_gpu(a, w)
    c = ops.gemm_nt(a, w, L.EPI_BIAS, bias=bias)
    assert rel_l2(c.float(), ref + bias) < 4e-3
    c = ops.gemm_nt(a, w, L.EPI_RES, aux=r)
    assert rel_l2(c.float(), ref + r.float()) < 4e-3
    c = ops.gemm_nt(a, w, L.EPI_BIAS_DROP_RES, bias=bias, aux=r, seed=5, p_drop=0.0)
    assert rel_l2(c.float(), ref + bias + r.float()) < 4e-3
    cs = torch.zeros(N, device='cuda')
    c = ops.gemm_nt(a, w, L.EPI_DGELU, aux=r, colsum=cs)
    x = r.float()
    dg = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    assert rel_l2(c.float(), ref * dg) < 6e-3
    assert rel_l2(cs, (ref * dg).sum(0)) < 2e-2
    del c, ref


@pytest.mark.parametrize('M,N,K', [(1024, 512, 64), (2048, 768, 256), (41984, 3072, 768)])
def test_gelu_byte_derivative_and_its_dgrad(M, N, K):
    """FFN activation with gelu'(u) kept as ONE byte per element (m3p_gelu_fwd_gq) and the data gradient that consumes it
    (M3P_EPI_MULQ: dU = (dY W) * decode(code), column sums): h against torch's erf-GELU, the decoded derivative against
    the exact one within the code's half step (2.5e-3; dead and saturated units decode to exactly 0 and 1), the product tightly against the fp32 product times
    the decoded derivative and - the bar that matters for training - against the exact derivative."""
    from m3p_amd import ops, lib as L
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    u = (torch.randn((M, N), device='cuda', generator=g) * 1.5).to(torch.bfloat16)
    u[0, :8] = torch.tensor([0.0, -0.0, 1e-4, -1e-4, 9.0, -9.0, 0.7518, -0.7518], device='cuda').to(torch.bfloat16)
    x = u.float()
    h, gq = ops.gelu_fwd_gq(u.clone())
    assert rel_l2(h.float(), torch.nn.functional.gelu(x)) < 3e-3
    exact = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    dec = ops.gq_unpack(gq, M, N)
    assert float((dec - exact).abs().max()) <= 0.5 * ops.GQ_STEP + 1e-5
    assert abs(float(dec[0, 4]) - 1.0) < 1e-6 and abs(float(dec[0, 5])) < 1e-6          # u = +9 / -9: the grid contains both ends
    a = (torch.randn((M, K), device='cuda', generator=g)).to(torch.bfloat16)
    w = (torch.randn((N, K), device='cuda', generator=g) * 0.05).to(torch.bfloat16)
    ref = _ref_on_gpu(a, w)
    cs = torch.zeros(N, device='cuda')
    c = ops.gemm_nt(a, w, L.EPI_MULQ, aux=gq, colsum=cs)
    assert rel_l2(c.float(), ref * dec) < 4e-3
    assert rel_l2(cs, (ref * dec).sum(0)) < 2e-2
    assert rel_l2(c.float(), ref * exact) < 8e-3          # (the byte adds ~3e-3 of relative error to the bf16 output's ~2e-3)
    # shapes outside whole eight-wave tiles are refused, not mis-read
    with pytest.raises(L.M3PError):
        ops.gemm_nt(a[:1000], w, L.EPI_MULQ, aux=gq[:1000 * N])
    # the producer fused into the lin1 GEMM (M3P_EPI_BIAS_GELUQ): h = gelu(a w^T + b) and the same byte layout
    bias = torch.randn((N,), device='cuda', generator=g)
    q2 = torch.empty(M * N, dtype=torch.uint8, device='cuda')
    h2 = ops.gemm_nt(a, w, L.EPI_BIAS_GELUQ, bias=bias, out2=q2)
    pre = ref + bias
    assert rel_l2(h2.float(), torch.nn.functional.gelu(pre)) < 4e-3
    exact2 = 0.5 * (1 + torch.erf(pre / math.sqrt(2))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
    # (the kernel's u is its own fp32 accumulation: against the library product's the codes may sit one level off where
    #  the derivative is steep - 1.5 steps covers it; on average they agree to the code's own rms)
    d2 = (ops.gq_unpack(q2, M, N) - exact2).abs()
    assert float(d2.max()) <= 1.5 * ops.GQ_STEP and float(d2.pow(2).mean().sqrt()) < 0.4 * ops.GQ_STEP


@pytest.mark.parametrize('M,N,V,K', [(1024, 1280, 1217, 128), (1024, 512, 512, 64), (2048, 2560, 2500, 768)])
def test_projection_with_block_statistics_and_the_cross_entropy_from_them(M, N, V, K):
    """The vocabulary projection that also leaves (max, sum exp) per row and 64-column block (M3P_EPI_BIAS_LSE), and the
    cross-entropy built on them (m3p_ce_lse_from_blocks + m3p_ce_bwd_colsum), against the two-pass form on the plain
    projection and against torch: same logits bit for bit, log-sum-exp / loss / gradient / bias gradient within fp32 / bf16
    rounding.  V < N leaves pad columns (and, in the first case, a whole 64-column block) out of the statistics."""
    from m3p_amd import ops, lib as L
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    a = torch.randn((M, K), device='cuda', generator=g).to(torch.bfloat16)
    w = (torch.randn((N, K), device='cuda', generator=g) * (1.2 / math.sqrt(K))).to(torch.bfloat16)      # logits ~ N(0, 1.2^2) + bias
    bias = torch.randn((N,), device='cuda', generator=g) * 0.5
    y = torch.randint(0, V, (M,), device='cuda', generator=g)
    plain = ops.gemm_nt(a, w, L.EPI_BIAS, bias=bias)
    stats = torch.empty((N // 64, M, 2), dtype=torch.float32, device='cuda')
    logits = ops.gemm_nt(a, w, L.EPI_BIAS_LSE, bias=bias, out2=stats, scale_cols=V)
    assert torch.equal(plain, logits)
    ref = torch.nn.functional.cross_entropy(plain[:, :V].float(), y, reduction='none')
    ref_lse = torch.logsumexp(plain[:, :V].float(), dim=1)
    l2 = plain.clone()
    loss_a, rows_a, cs_a = ops.ce_fwd_bwd_colsum(l2, V, y, 1.0 / M, 1.0 / M)
    loss_b, rows_b, cs_b = ops.ce_from_block_stats(logits, V, y, stats, 1.0 / M, 1.0 / M)
    # (the block statistics are taken on the fp32 accumulators, the two-pass form reads the bf16 logits back: the rows'
    #  log-sum-exp differ by the logits' rounding, a few 1e-3 at most)
    assert float((rows_b - ref).abs().max()) < 2e-2 and abs(float(loss_b) - float(ref.mean())) < 2e-3
    assert abs(float(loss_b) - float(loss_a)) < 2e-3
    assert rel_l2(logits[:, :V].float(), l2[:, :V].float()) < 1e-2           # both now hold the gradient
    assert float(logits[:, V:].float().abs().max()) == 0.0 if V < N else True
    p = torch.softmax(plain[:, :V].float(), dim=1)
    p[torch.arange(M, device='cuda'), y] -= 1.0
    assert rel_l2(logits[:, :V].float(), p / M) < 1e-2
    assert rel_l2(cs_b[:V], (p / M).sum(0)) < 2e-2
    del ref_lse


@pytest.mark.parametrize('N,K', [(2304, 768), (768, 768), (3072, 768), (768, 3072)])
def test_gemm_wgrad_at_the_benchmarked_sizes(N, K):
    from m3p_amd import ops
    M = 41984
    g = torch.Generator(device='cuda').manual_seed(N + K)
    dy = (torch.randn((M, N), device='cuda', generator=g) * 0.1).to(torch.bfloat16)
    x = torch.randn((M, K), device='cuda', generator=g).to(torch.bfloat16)
    dw = torch.ones((N, K), device='cuda')
    ops.gemm_wgrad(dy, x, dw)
    # reference: the fp64 product in slices of M.  (The once-in-twenty mismatch this test used to diagnose was the round-2
    # early-clobber race of the fragment macro: root-caused, fixed, checked on the ISA by tests/test_kernel_isa.py and
    # stressed by test_gemm_wgrad_with_fresh_operands_of_changing_shapes below.)
    ref64 = torch.ones((N, K), dtype=torch.float64, device='cuda')
    for m0 in range(0, M, 8192):
        ref64 += dy[m0:m0 + 8192].double().t() @ x[m0:m0 + 8192].double()
    assert rel_l2(dw.double(), ref64) < 1e-5


@pytest.mark.parametrize('M,Na,Ka,Nb,Kb', [(41984, 2304, 768, 768, 768), (8192, 768, 768, 256, 512), (4096, 3072, 768, 768, 3072),
                                           (1000, 100, 64, 768, 768)])
def test_gemm_wgrad_pair(M, Na, Ka, Nb, Kb):
    """Two weight gradients over the same rows in one launch (the attention sub-layer's q/k/v + out_lin pair at the benchmarked
    size; a small pair; a pair with too many tiles and a ragged one, which fall back to two launches): both against fp64
    products, accumulating into non-zero gradients."""
    from m3p_amd import ops
    g = torch.Generator(device='cuda').manual_seed(M + Na)
    mk = lambda n, sc: (torch.randn((M, (n + 7) // 8 * 8), device='cuda', generator=g) * sc).to(torch.bfloat16)   # noqa: E731
    dya, xa, dyb, xb = mk(Na, 0.1), mk(Ka, 1.0), mk(Nb, 0.1), mk(Kb, 1.0)
    dwa, dwb = torch.ones((Na, Ka), device='cuda'), torch.full((Nb, Kb), 2.0, device='cuda')
    ops.gemm_wgrad_pair(dya[:, :Na], xa[:, :Ka], dwa, dyb[:, :Nb], xb[:, :Kb], dwb)
    for dw, dy, x, n, k, c in ((dwa, dya, xa, Na, Ka, 1.0), (dwb, dyb, xb, Nb, Kb, 2.0)):
        ref = torch.full((n, k), c, dtype=torch.float64, device='cuda')
        for m0 in range(0, M, 8192):
            ref += dy[m0:m0 + 8192, :n].double().t() @ x[m0:m0 + 8192, :k].double()
        assert rel_l2(dw.double(), ref) < 1e-5, (n, k)


def test_gemm_wgrad_stored_into_a_zero_gradient():
    """dw_is_zero: shapes the four-wave kernel deals out as whole tiles (>= 4 tiles per CU - the vocabulary matrix's form)
    are stored instead of accumulated with atomics; the result must equal the accumulating call's on zeros, and a shape
    that is not dealt out that way must quietly accumulate (here: into zeros, so the same answer)."""
    from m3p_amd import ops
    for M, N, K in ((4096, 16384, 4096), (4096, 768, 768)):
        g = torch.Generator(device='cuda').manual_seed(N + K)
        dy = (torch.randn((M, N), device='cuda', generator=g) * 0.1).to(torch.bfloat16)
        x = torch.randn((M, K), device='cuda', generator=g).to(torch.bfloat16)
        stored = torch.full((N, K), float('nan'), device='cuda') if N > 768 else torch.zeros((N, K), device='cuda')
        ops.gemm_wgrad(dy, x, stored, alpha=0.5, dw_is_zero=True)       # (NaN-filled: a store must not read what was there)
        added = torch.zeros((N, K), device='cuda')
        ops.gemm_wgrad(dy, x, added, alpha=0.5)
        assert torch.isfinite(stored).all()
        assert rel_l2(stored.double(), added.double()) < 1e-6
        ref = 0.5 * (dy[:, :256].double().t() @ x.double())
        assert rel_l2(stored[:256].double(), ref) < 1e-5
        del stored, added, ref


def test_gemm_wgrad_pair_with_gradients_the_workspace_form_cannot_take():
    """A pair whose gradient views are 4 bytes off 16-byte alignment (or have an odd pitch): the entry point must fall back to
    two single launches (each of which flushes with atomics) instead of pairing on the atomic path - both against fp64."""
    from m3p_amd import ops
    M, Na, Ka, Nb, Kb = 8192, 768, 768, 256, 512
    g = torch.Generator(device='cuda').manual_seed(11)
    mk = lambda n, sc: (torch.randn((M, n), device='cuda', generator=g) * sc).to(torch.bfloat16)   # noqa: E731
    dya, xa, dyb, xb = mk(Na, 0.1), mk(Ka, 1.0), mk(Nb, 0.1), mk(Kb, 1.0)
    for off, pitch_pad in ((1, 0), (0, 1)):
        bufa = torch.ones(off + Na * (Ka + pitch_pad), device='cuda')
        bufb = torch.full((off + Nb * (Kb + pitch_pad),), 2.0, device='cuda')
        dwa = bufa[off:].view(Na, Ka + pitch_pad)[:, :Ka]
        dwb = bufb[off:].view(Nb, Kb + pitch_pad)[:, :Kb]
        ops.gemm_wgrad_pair(dya, xa, dwa, dyb, xb, dwb)
        for dw, dy, x, c in ((dwa, dya, xa, 1.0), (dwb, dyb, xb, 2.0)):
            ref = dy.double().t() @ x.double() + c
            assert rel_l2(dw.double(), ref) < 1e-5, (off, pitch_pad)


def test_gemm_wgrad_with_fresh_operands_of_changing_shapes():
    """~900 weight-gradient launches, each on freshly allocated operands of another shape, against fp64 products: the pattern
    under which one launch in ~300 returned a wrong half-fragment's worth of one tile before the fragment asm's outputs were
    made early-clobber (tools/wgrad_stress2.py) - warm, repeated operands never showed it."""
    from m3p_amd import ops
    shapes = [(4288, 2304, 768), (4100, 768, 768), (8192, 768, 768), (4096, 3072, 768), (8192, 768, 3072), (4160, 1024, 1024),
              (16384, 256, 128), (20992, 2304, 768), (20992, 3072, 768)]
    bad = []
    for rnd in range(100):
        for M, N, K in shapes:
            g = torch.Generator(device='cuda').manual_seed(N + K + rnd)
            dy = (torch.randn((M, N), device='cuda', generator=g) * 0.1).to(torch.bfloat16)
            x = torch.randn((M, K), device='cuda', generator=g).to(torch.bfloat16)
            dw = torch.ones((N, K), device='cuda')
            ops.gemm_wgrad(dy, x, dw)
            ref = torch.ones((N, K), dtype=torch.float64, device='cuda')
            for m0 in range(0, M, 8192):
                ref += dy[m0:m0 + 8192].double().t() @ x[m0:m0 + 8192].double()
            err = float((dw.double() - ref).norm() / ref.norm())
            if err > 1e-5:
                bad.append((rnd, M, N, K, err))
            del dy, x, dw, ref
    assert not bad, bad
