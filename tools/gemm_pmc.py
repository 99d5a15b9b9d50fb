#!/usr/bin/env python3
"""One GEMM shape per process, one kernel flavour: for rocprofv3 --pmc passes that compare the vendor's kernel with ours.
    python tools/gemm_pmc.py <vendor|1|2> <N> <K> [epi]       (1 = eight-wave, 2 = four-wave)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
M = 41984
a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
if which == 'vendor':
    for _ in range(25):
        c = a @ w.t()
else:
    from m3p_amd import lib as L
    lib = L.load()
    lib.m3p_debug_set_variant(int(which))
    out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    aux = torch.randn(M, N, device='cuda').to(torch.bfloat16)
    ep = L.Epilogue()
    ep.aux = aux.data_ptr() if epi in (3, 4) else None
    ep.ld_aux = N
    ep.alpha = 1.0
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(25):
        rc = lib.m3p_gemm_nt_bf16(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, epi, C.byref(ep), st)
        assert rc == 0
torch.cuda.synchronize()
