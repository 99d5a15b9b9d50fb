#!/usr/bin/env python3
"""Drives tools/probes/mfma_rate.so: bf16 16x16x32 against 32x32x16, one / two waves per SIMD, random / zero operands."""
import ctypes as C, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, 'mfma_rate.so'))
lib.mfma_rate.restype = C.c_float
lib.mfma_rate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
out = torch.zeros(1024, device='cuda')
iters = 20000
for fill in ('random', 'zero'):
    src = ((torch.randn(4096 * 8, device='cuda') * 0.5) if fill == 'random' else torch.zeros(4096 * 8, device='cuda')).to(torch.bfloat16)
    for threads in (256, 512):
        for mode, nm in ((0, 'bf16 16x16x32'), (1, 'bf16 32x32x16'), (2, 'fp8 16x16x128 (x4 FLOP)')):
            best = min(lib.mfma_rate(mode, 256, threads, iters, src.data_ptr(), out.data_ptr()) for _ in range(3))
            fl = 256 * (threads // 64) * iters * 524288.0 * (4 if mode == 2 else 1)
            print('%-7s %d waves/SIMD  %s  %8.3f ms  %7.1f TFLOP/s' % (fill, threads // 256, nm, best, fl / best / 1e9))
