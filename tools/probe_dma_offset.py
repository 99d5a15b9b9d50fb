#!/usr/bin/env python3
"""Where does global_load_lds_dwordx4 ... offset:1024 put its data, and which source bytes does it read?"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import lib as L
lib = L.load()
f = lib.m3p_debug_probe_dma_offset
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
src = torch.arange(4096, dtype=torch.int32, device='cuda')      # word i holds i
out = torch.zeros(2048, dtype=torch.int32, device='cuda')
print('rc', f(src.data_ptr(), out.data_ptr(), L.stream()))
torch.cuda.synchronize()
o = out.cpu().numpy()
import numpy as np
hit = np.nonzero(o != np.int32(-559038737))[0]
print('LDS words written: %d..%d (window word index; M0 base = word 128)' % (hit.min(), hit.max()))
print('first values', o[hit[:4]], ' -> source word index; pointer passed = word 512, +offset 1024 B = word 768')
