"""Pin the gfx950 instruction semantics the kernels are built on (MFMA operand/result
lane maps, ds_read_b64_tr_b16 gather).  If one of these fails, every GEMM/attention
result is suspect — read this first."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_mfma_16x16x32_lane_maps():
    from m3p_amd import lib as L
    lib = L.load()
    rs = np.random.RandomState(0)
    a = torch.from_numpy(rs.randint(-4, 5, size=(16, 32)).astype(np.float32))
    w = torch.from_numpy(rs.randint(-4, 5, size=(16, 32)).astype(np.float32))  # asymmetric on purpose
    ad, wd = a.to(torch.bfloat16).cuda(), w.to(torch.bfloat16).cuda()
    d = torch.zeros(16, 16, dtype=torch.float32, device='cuda')
    rc = torch.zeros(256, 2, dtype=torch.int32, device='cuda')
    L.check(lib.m3p_probe_mfma_16x16x32(ad.data_ptr(), wd.data_ptr(), d.data_ptr(), rc.data_ptr(), L.stream()), 'probe')
    torch.cuda.synchronize()
    ref = a @ w.t()
    assert torch.equal(d.cpu(), ref), 'MFMA lane map differs from the documented one:\n%s\n%s' % (d.cpu(), ref)


@pytest.mark.parametrize('a_is_bf8', [0, 1])
def test_mfma_fp8_16x16x128_lane_maps(a_is_bf8):
    from m3p_amd import lib as L
    lib = L.load()
    rs = np.random.RandomState(2 + a_is_bf8)
    a = torch.from_numpy(rs.randint(-4, 5, size=(16, 128)).astype(np.float32))     # small integers: exact in e4m3 and e5m2
    w = torch.from_numpy(rs.randint(-4, 5, size=(16, 128)).astype(np.float32))
    ad = a.to(torch.float8_e5m2 if a_is_bf8 else torch.float8_e4m3fn).cuda()      # (named: a temporary would be freed and reused)
    wd = w.to(torch.float8_e4m3fn).cuda()
    d = torch.zeros(16, 16, dtype=torch.float32, device='cuda')
    L.check(lib.m3p_probe_mfma_fp8_16x16x128(ad.data_ptr(), wd.data_ptr(), d.data_ptr(), a_is_bf8, L.stream()), 'probe')
    torch.cuda.synchronize()
    assert torch.equal(d.cpu(), a @ w.t()), 'fp8 MFMA lane map / format selector differs from the documented one'


def test_ds_read_tr16_gather():
    from m3p_amd import lib as L
    lib = L.load()
    tile = torch.arange(64 * 16, dtype=torch.int16).view(64, 16)
    out = torch.zeros(64, 4, dtype=torch.int16, device='cuda')
    L.check(lib.m3p_probe_tr16(tile.cuda().data_ptr(), out.data_ptr(), L.stream()), 'probe')
    torch.cuda.synchronize()
    exp = torch.zeros(64, 4, dtype=torch.int16)
    for l in range(64):
        g, t = l >> 4, l & 15
        # the probe points group g at rows 4g..4g+3 (16 rows apart would be other groups)
        for j in range(4):
            exp[l, j] = tile[4 * g + j, t]
    assert torch.equal(out.cpu(), exp), 'tr16 gather differs:\n%s' % out.cpu()


def test_permlane16_swap_trades_odd_rows_of_x_for_even_rows_of_y():
    """v_permlane16_swap_b32 x, y: afterwards x = [x.row0, y.row0, x.row2, y.row2] and y = [x.row1, y.row1, x.row3, y.row3]
    (rows of 16 lanes) - four lanes that hold the 8-byte quarters of one 32-byte run then hold its 16-byte halves."""
    from m3p_amd import lib as L
    lib = L.load()
    out = torch.zeros(128, dtype=torch.int32, device='cuda')
    L.check(lib.m3p_probe_permlane16_swap(out.data_ptr(), L.stream()), 'probe')
    torch.cuda.synchronize()
    x, y = out[:64].cpu().tolist(), out[64:].cpu().tolist()
    lanes = list(range(64))
    xin, yin = lanes, [100 + l for l in lanes]
    row = lambda v, r: v[16 * r:16 * r + 16]      # noqa: E731
    assert x == row(xin, 0) + row(yin, 0) + row(xin, 2) + row(yin, 2), x
    assert y == row(xin, 1) + row(yin, 1) + row(xin, 3) + row(yin, 3), y
