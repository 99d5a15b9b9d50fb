#!/usr/bin/env python3
"""Issue cost of one memory instruction between 8 MFMAs for a wave that is alone on its SIMD."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import lib as L
lib = L.load()
f = lib.m3p_debug_probe_issue
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = 2048
src = torch.randn(nb * 4 * 4096 + 65536, device='cuda').to(torch.bfloat16)
out = torch.zeros(nb * 4 * 2, dtype=torch.int64, device='cuda')
names = ['none', 'global_load_lds vaddr64', 'global_load_lds saddr+voff32', 'global_load_dwordx4 -> vgpr', 'buffer_load_dwordx4 lds', 'ds_read_b128', '1 valu', 'glds invariant addr+m0', 'ds_read invariant addr', 's_mov m0 + s_nop', '4 valu', '4 v_mul_lo_u32', '4 v_mul_u32_u24', '4 v_exp_f32', '4 v_mad_u32_u24', '4 v_mul_hi_u32', 'ds_write_b128', 'global_load_dwordx4 saddr+voff -> vgpr']
modes = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else range(18)
for mode in modes:
    for _ in range(2):
        rc = f(mode, src.data_ptr(), out.data_ptr(), iters, nb, L.stream())
    torch.cuda.synchronize()
    t = out.view(-1, 2)[:, 0].double()
    print('%-30s rc=%d  ticks/round: mean %.1f  min %.1f  max %.1f   (8 MFMAs = 128 ideal)' % (names[mode], rc, t.mean() / iters, t.min() / iters, t.max() / iters), flush=True)
