#!/usr/bin/env python3
"""What does it cost to leave CUs to RCCL, and what does it cost not to?  (one MI355X, no second GPU needed)

Every hot GEMM is a persistent grid of one workgroup per CU that fills the CU's LDS and registers, so a collective's
kernel arriving mid-backward finds no CU it could share: it waits for a GEMM to end, or - once resident - makes the next
GEMM run a second round on the CUs it holds.  Under data parallelism the GEMM grids therefore leave r CUs free
(m3p_set_persistent_grid(num_cus - r)).  This tool prices both sides on the benchmarked step: a stand-in for the
collectives (m3p_debug_side_copy: `--channels` workgroups streaming `--mb` MB per launch, `--launches` launches per step on
a side stream, issued from the backward hooks' position: right after the step's forward) next to the step, with the GEMM
grids at 256 and at 256 - r workgroups.

    python tools/cu_reserve_ab.py > gpurun_out/cu_reserve.txt      (copy the summary to profiles/r03_cu_reserve.txt)
"""
import argparse
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--channels', type=int, default=16)
    ap.add_argument('--mb', type=int, default=96, help='MB copied per side launch (a layer bucket is 28 MB reduced = ~2 x that moved)')
    ap.add_argument('--launches', type=int, default=14)
    ap.add_argument('--queue-ab', action='store_true', help='A/B of the dynamic tile queues (m3p_set_tile_queue) instead of the grid sizes')
    ap.add_argument('--profile-arm', default=None, help="'side' / 'none': only run that arm at the full grid (for rocprofv3 --stats)")
    args = ap.parse_args()
    torch.set_num_threads(4)
    import bench
    from m3p_amd import synth, lib as L
    lib = L.load()
    lib.m3p_debug_side_copy.restype = C.c_int
    lib.m3p_debug_side_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    cfg = dict(synth.CONFIGS['cfg2'])
    cfg['B'] = args.batch
    trainer, tup = bench.build(cfg, 0.1, 1, 0, 0)
    side = torch.cuda.Stream()
    src = torch.empty(args.mb << 20, dtype=torch.uint8, device='cuda')
    dst = torch.empty_like(src)
    side_ms = []

    def step(with_side):
        if with_side:
            ev = torch.cuda.Event(); ev.record()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                e0.record()
                for _ in range(args.launches):
                    L.check(lib.m3p_debug_side_copy(src.data_ptr(), dst.data_ptr(), src.numel(), args.channels, side.cuda_stream), 'side_copy')
                e1.record()
            side_ms.append((e0, e1))
        trainer.pretrain_under_step(tup, 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0)
        trainer.n_iter += 1
        if with_side:
            torch.cuda.current_stream().wait_stream(side)

    def run(grid, with_side):
        lib.m3p_set_persistent_grid(grid)
        del side_ms[:]
        for _ in range(4):
            step(with_side)
        torch.cuda.synchronize()
        del side_ms[:]
        t = time.perf_counter()
        for _ in range(args.steps):
            step(with_side)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / args.steps * 1e3
        sm = sum(a.elapsed_time(b) for a, b in side_ms) / max(len(side_ms), 1) if with_side else 0.0
        return ms, sm

    ncu = L.num_cus()
    for _ in range(6):
        step(False)
    if args.profile_arm:
        print(run(ncu, args.profile_arm == 'side'))
        return
    if args.queue_ab:
        from m3p_amd import ops
        print('# step = cfg2 B=%d; side traffic = %d launches x %d MB per step on %d workgroups; tile queue off / on' % (args.batch, args.launches, args.mb, args.channels))
        print('# queue side  ms/step  side ms/step')
        for rnd in range(3):
            for q in (False, True):
                ops.set_tile_queue(q)
                for with_side in (False, True):
                    ms, sm = run(ncu, with_side)
                    print('%-5s %-5s %7.2f  %8.2f' % ('on' if q else 'off', 'yes' if with_side else 'no', ms, sm), flush=True)
        ops.set_tile_queue(False)
        return
    print('# step = cfg2 B=%d; side traffic = %d launches x %d MB per step on %d workgroups (stand-in for RCCL channels)'
          % (args.batch, args.launches, args.mb, args.channels))
    print('# grid  side  ms/step  side-stream ms/step (first side launch -> last done)  side GB/s (read + write)')
    rows = []
    for rnd in range(2):
        for r in (0, 8, 16, 32):
            for with_side in (False, True):
                ms, sm = run(ncu - r, with_side)
                gbs = 2.0 * args.launches * args.mb * (1 << 20) / (sm * 1e-3) / 1e9 if sm else 0.0
                rows.append((ncu - r, with_side, ms, sm, gbs))
                print('%5d  %-5s %7.2f  %8.2f  %7.0f' % (ncu - r, 'yes' if with_side else 'no', ms, sm, gbs), flush=True)
    lib.m3p_set_persistent_grid(0)


if __name__ == '__main__':
    main()
