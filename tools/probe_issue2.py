#!/usr/bin/env python3
"""Issue cost of one memory instruction per 8 MFMAs for a wave alone on its SIMD (clean version)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import lib as L
lib = L.load()
f = lib.m3p_debug_probe_issue2
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
nb, trips = 256, 256
src = torch.randn(nb * 4 * 4096 + 65536, device='cuda').to(torch.bfloat16)
out = torch.zeros(nb * 4 * 2, dtype=torch.int64, device='cuda')
names = ['none', 'LDS-DMA (invariant)', 'global_load_dwordx4 saddr+voff', 'ds_read_b128', 'ds_write_b128', 'global_load_dwordx4 vaddr64', 'LDS-DMA + v_lshl_add_u64']
for mode in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else range(7)):
    for _ in range(2):
        rc = f(mode, src.data_ptr(), out.data_ptr(), trips, nb, L.stream())
    torch.cuda.synchronize()
    t = out.view(-1, 2)[:, 0].double() / (trips * 8)
    print('%-34s rc=%d  clocks per round of 8 MFMAs: mean %.1f  max %.1f' % (names[mode], rc, t.mean(), t.max()), flush=True)
