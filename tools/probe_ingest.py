#!/usr/bin/env python3
"""L2 -> CU ingest rate (bytes per clock per CU) for LDS-DMA and plain vector loads."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import lib as L
lib = L.load()
f = lib.m3p_debug_probe_ingest
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
nb, rounds = 256, 512
src = torch.randn(8 * (2 << 20) // 4 + 1024, device='cuda')
out = torch.zeros(nb * 4, dtype=torch.int64, device='cuda')
names = ['LDS-DMA 128-B row segments', 'LDS-DMA 64-B row segments', 'global_load_dwordx4 -> VGPR, 128-B segments', 'LDS-DMA contiguous 1 KB']
for mode in range(4):
    for _ in range(2):
        rc = f(mode, src.data_ptr(), out.data_ptr(), rounds, nb, L.stream())
    torch.cuda.synchronize()
    t = out.double()
    print('%-44s rc=%d  %.1f B/clk/CU (mean), %.1f (slowest wave)' % (names[mode], rc, rounds * 32768 / t.mean(), rounds * 32768 / t.max()), flush=True)
