#!/usr/bin/env python3
"""Register / scratch / occupancy figures of the kernels in one csrc file (cross-compiles, no GPU):
   python tools/kernel_info.py gemm.hip w8 [-DFLAG=1 ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
out = os.path.join(tempfile.mkdtemp(), 'k.s')
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-ffp-contract=fast',
       '-Wno-unused-result', '--cuda-device-only', '-S', os.path.join(ROOT, 'm3p_amd', 'csrc', src), '-o', out] + sys.argv[3:]
subprocess.run(cmd, check=True, capture_output=True)
body = open(out).read()
for m in re.finditer(r'^(_Z\S*):\s*; @', body, re.M):
    name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    if pat not in name:
        continue
    k = body.find('; Kernel info:', m.end())
    if k < 0:
        continue
    info = dict(re.findall(r'; (\w+): (\d+)', body[k:k + 700]))
    short = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print('%-44s vgpr %3s agpr %3s scratch %4s occ %s' % (short[:44], info.get('NumVgprs'), info.get('NumAgprs'),
                                                             info.get('ScratchSize'), info.get('Occupancy')))
