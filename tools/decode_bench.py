#!/usr/bin/env python3
"""Decode-step timing of the causal decoder (m3p_amd/decoder.py) at the M3P-base size: 12 layers / 768 wide / 12 heads,
V = 250 002, a source of 36 regions + 100 tokens; greedy and beam search.   python tools/decode_bench.py [bs] [beam]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import synth
from m3p_amd.model.transformer import TransformerModel

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
beam = int(sys.argv[2]) if len(sys.argv) > 2 else 4
P = synth.model_params(768, 12, 12, 250002, n_dec_layers=12, n_langs=2, id2lang={0: 'en', 1: 'zh'}, lang2id={'en': 0, 'zh': 1})
torch.manual_seed(0)
m = TransformerModel(P, is_encoder=False, with_output=True, is_crossModal=True).cuda().eval()
with torch.no_grad():
    m.pred_layer.proj.bias[synth.EOS] = -1e4          # nobody stops early: every run decodes max_len - 1 steps
S, max_len = 136, 33
src = torch.randn(bs, S, 768, device='cuda')
src_len = torch.full((bs,), S, dtype=torch.long, device='cuda')


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


with torch.no_grad():
    dt = timed(lambda: m.generate(src, src_len, 1, max_len=max_len))
    print('greedy  bs=%d: %.2f ms per step, %.0f tokens/s' % (bs, dt / (max_len - 1) * 1e3, bs * (max_len - 1) / dt))
    dt = timed(lambda: m.generate_beam(src, src_len, 1, beam, 1.0, False, max_len=max_len))
    print('beam %d  bs=%d (%d rows): %.2f ms per step, %.0f sentences x tokens/s'
          % (beam, bs, bs * beam, dt / (max_len - 1) * 1e3, bs * (max_len - 1) / dt))
