#!/usr/bin/env python3
"""The weight-gradient tests of tests/test_gemm.py as a loop with fresh tensors of changing shapes per call (the pattern under
which a rare wrong block showed up), with a detailed report of every bad result.  usage: python tools/wgrad_stress2.py [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops

SHAPES = [(4288, 2304, 768), (4100, 768, 768), (8200, 3072, 768), (8192, 768, 768), (4288, 2304, 768), (16384, 256, 128), (4864, 5008, 768), (4096, 3072, 768),
          (8192, 768, 3072), (4160, 1024, 1024), (41984, 2304, 768), (41984, 768, 768), (41984, 3072, 768), (41984, 768, 3072)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = n = 0
for rnd in range(rounds):
    for M, N, K in SHAPES:
        g = torch.Generator(device='cuda').manual_seed(N + K + rnd)
        dy = (torch.randn((M, N), device='cuda', generator=g) * 0.1).to(torch.bfloat16)
        x = torch.randn((M, K), device='cuda', generator=g).to(torch.bfloat16)
        dw = torch.ones((N, K), device='cuda')
        ops.gemm_wgrad(dy, x, dw)
        ref = torch.ones((N, K), dtype=torch.float64, device='cuda')
        for m0 in range(0, M, 8192):
            ref += dy[m0:m0 + 8192].double().t() @ x[m0:m0 + 8192].double()
        err = float((dw.double() - ref).norm() / ref.norm())
        n += 1
        if err != err:
            err = 1.0
        if err > 1e-5:
            bad += 1
            d = (dw.double() - ref).abs()
            rows = (d.max(dim=1).values > 1e-2).nonzero().view(-1)
            cols = (d.max(dim=0).values > 1e-2).nonzero().view(-1)
            blk_o, blk_r = dw.double()[rows][:, cols] - 1, ref[rows][:, cols] - 1
            # is the bad block the product over a SUBSET of the rows of M (a missing / doubled partial)?
            C = max(256 // ((N // 256) * (K // 256)), 1)
            print('BAD round %d M=%d N=%d K=%d rel %.3e: rows %d..%d (%d) cols %s; ours/ref in block: mean %.4f, (ours-ref)/ref rms %.4f; C=%d chunks'
                  % (rnd, M, N, K, err, int(rows.min()), int(rows.max()), len(rows), cols.tolist(), float((blk_o * blk_r).sum() / (blk_r * blk_r).sum()),
                     float(((blk_o - blk_r).norm() / blk_r.norm())), C), flush=True)
            # where did it go wrong: the partial tiles the w4 kernel left in the workspace, or their reduction?
            wsb = list(ops._WGRAD_WS.values())[0]
            ntile_ = (N // 256) * (K // 256)
            Cc = 256 // ntile_
            wsf = wsb[:256 * (65536 + 272) * 4].view(torch.float32).view(256, 65536 + 272)
            tiles = wsb[256 * (65536 + 272) * 4:256 * (65536 + 272) * 4 + 1024].view(torch.int32)
            host = torch.ones((N, K), dtype=torch.float64, device='cuda')
            nmt_ = M // 64
            for s_ in range(Cc * ntile_):
                t_ = s_ % ntile_
                c_ = s_ // ntile_
                assert int(tiles[s_]) == t_, (s_, int(tiles[s_]))
                ti2, tj2 = t_ // (K // 256), t_ % (K // 256)
                part = wsf[s_, :65536].view(256, 256).double()
                host[ti2 * 256:(ti2 + 1) * 256, tj2 * 256:(tj2 + 1) * 256] += part
                m0_, m1_ = (c_ * nmt_ // Cc) * 64, ((c_ + 1) * nmt_ // Cc) * 64
                true = dy[m0_:m1_, ti2 * 256:(ti2 + 1) * 256].double().t() @ x[m0_:m1_, tj2 * 256:(tj2 + 1) * 256].double()
                pe = float((part - true).norm() / true.norm())
                if pe > 1e-5:
                    dd = (part - true).abs()
                    rr_ = (dd.max(dim=1).values > 1e-3).nonzero().view(-1)
                    cc_ = (dd.max(dim=0).values > 1e-3).nonzero().view(-1)
                    print('   WORKSPACE slot %d (chunk %d rows %d..%d, tile %d): partial off by rel %.3e; rows %d..%d (%d) cols %s'
                          % (s_, c_, m0_, m1_, t_, pe, int(rr_.min()), int(rr_.max()), len(rr_), cc_.tolist()), flush=True)
                    # is the wrong block the product over a sub-range of the chunk's K-tiles?
                    blk_t = true[rr_][:, cc_]; blk_p = part[rr_][:, cc_]
                    for kt in range(m0_ // 64, m1_ // 64):
                        for half in range(4):
                            r0 = kt * 64 + half * 16
                            sl = dy[r0:r0 + 16, ti2 * 256:(ti2 + 1) * 256][:, rr_].double().t() @ x[r0:r0 + 16, tj2 * 256:(tj2 + 1) * 256][:, cc_].double()
                            miss = float(((blk_t - blk_p) - sl).norm() / sl.norm())
                            extra = float(((blk_p - blk_t) - sl).norm() / sl.norm())
                            if miss < 0.2 or extra < 0.2:
                                print('      = rows of M %d..%d %s (residual %.3f)' % (r0, r0 + 15, 'MISSING' if miss < extra else 'COUNTED TWICE', min(miss, extra)), flush=True)
            print('   host reduction of the workspace against the reference: rel %.3e (device result: %.3e)'
                  % (float((host - ref).norm() / ref.norm()), err), flush=True)
            e = (dw.double() - ref)
            for ti_ in range(N // 256):
                line = []
                for tj_ in range(K // 256):
                    blk = e[ti_ * 256:(ti_ + 1) * 256, tj_ * 256:(tj_ + 1) * 256]
                    rb = ref[ti_ * 256:(ti_ + 1) * 256, tj_ * 256:(tj_ + 1) * 256] - 1
                    # per 16-column group: rms error relative to rms value; rows with error
                    grp = [float(blk[:, c0:c0 + 16].norm() / rb[:, c0:c0 + 16].norm()) for c0 in range(0, 256, 16)]
                    nrows = int((blk.abs().max(dim=1).values > 1e-2).sum())
                    line.append('t(%d,%d) rows %3d grp[%s]' % (ti_, tj_, nrows, ' '.join('%.2f' % v if v > 1e-4 else '.' for v in grp)))
                print('   ' + ' | '.join(line), flush=True)
            # one bad tile in detail: which rows, and is a bad row the reference of ANOTHER row / column group?
            ti_, tj_ = int(rows.min()) // 256, int(cols.min()) // 256
            c0 = int(cols.min())
            blk = e[ti_ * 256:(ti_ + 1) * 256, c0:c0 + 16]
            print('   tile (%d,%d) cols %d..%d: |err| per row (x100 of row rms): %s' % (ti_, tj_, c0, c0 + 15,
                  ' '.join('%d' % int(100 * float(blk[r].norm() / (ref[ti_ * 256 + r, c0:c0 + 16] - 1).norm())) for r in range(256))), flush=True)
            nmt = M // 64
            for c in range(C):
                m0, m1 = (c * nmt // C) * 64, ((c + 1) * nmt // C) * 64
                part = dy[m0:m1][:, rows].double().t() @ x[m0:m1][:, cols].double()
                resid = (blk_r - blk_o)
                print('   chunk %d rows of M %d..%d: |missing - part| / |part| = %.4f   |missing + part|/|part| = %.4f'
                      % (c, m0, m1, float((resid - part).norm() / part.norm()), float((resid + part).norm() / part.norm())), flush=True)
        del dy, x, dw, ref
print('%d bad of %d' % (bad, n))
