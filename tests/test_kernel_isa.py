"""Build-time invariants of the generated ISA (no GPU needed: hipcc cross-compiles gfx950).

The four-wave GEMM keeps its 256 accumulators in AGPRs through inline asm with literal register
numbers.  That is only sound while the compiler itself never touches an AGPR (spill-to-AGPR) or
scratch in that kernel; this test disassembles every instantiation and checks exactly that."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = '/opt/rocm/bin/hipcc'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_w4_gemm_accumulators_are_ours(tmp_path):
    out = tmp_path / 'gemm.s'
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-ffp-contract=fast',
           '-Wno-unused-result', '--cuda-device-only', '-S', os.path.join(ROOT, 'm3p_amd', 'csrc', 'gemm.hip'), '-o', str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    text = out.read_text().splitlines()
    starts = [i for i, l in enumerate(text) if re.match(r'^_ZN\S*gemm_nt_w4_kernelILi\dELb0ELi0E\S*:', l)]
    assert len(starts) == 7, 'expected the seven epilogue instantiations of the production kernel'
    for st in starts:
        name = text[st].split(':')[0]
        in_asm, bad, n_mfma = False, [], 0
        for l in text[st + 1:]:
            if l.startswith('\t.amdhsa_kernel') or l.startswith('.Lfunc_end'):
                break
            if 'ASMSTART' in l:
                in_asm = True
            elif 'ASMEND' in l:
                in_asm = False
            elif in_asm and 'v_mfma' in l:
                n_mfma += 1
            elif not in_asm and ('v_accvgpr' in l or 'scratch_' in l):
                bad.append(l.strip())
        assert not bad, '%s: compiler-generated AGPR / scratch traffic: %s' % (name, bad[:3])
        assert n_mfma >= 3 * 64, name     # phase 1 (first / accumulate) + phase 2
    # the persistent eight-wave kernels (one workgroup per CU, every stall exposed) must not spill either: a
    # 16-byte-struct temporary array once put 144 bytes of them into scratch and cost 15 % of the dGELU launch
    body = '\n'.join(text)
    for m in re.finditer(r'^(_ZN\S*gemm_(?:nt|wgrad)_ring_kernel\S*):', body, re.M):
        k = body.index('; Kernel info:', body.index('.Lfunc_end', m.end()))
        info = dict(re.findall(r'; (\w+): (\d+)', body[k:k + 600]))
        assert info['ScratchSize'] == '0' and info['NumAgprs'] == '0', (m.group(1), info['ScratchSize'], info['NumAgprs'])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_eight_wave_gemm_fits_two_waves_per_simd(tmp_path):
    """The eight-wave 256x256 NT kernel lives on 256 registers per wave (two waves per SIMD): 128 accumulators + a
    quarter K-tile of fragments.  A whole k-step of fragments ahead spilled inside the K loop - this pins the budget: no
    scratch at all in the plain / bias / residual instantiations, a few dwords at most (epilogue only) in the
    dGELU / multiply ones, occupancy 2 everywhere; the fp8 body may keep its small epilogue spill."""
    out = tmp_path / 'gemm.s'
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-ffp-contract=fast',
           '-Wno-unused-result', '--cuda-device-only', '-S', os.path.join(ROOT, 'm3p_amd', 'csrc', 'gemm.hip'), '-o', str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    body = out.read_text()
    seen = 0
    queued = 0
    copies = 0
    for m in re.finditer(r'^(_ZN\S*gemm_nt_w8_kernelILi(\d)ELb([01])ELb([01])E\S*):', body, re.M):
        k = body.index('; Kernel info:', body.index('.Lfunc_end', m.end()))
        info = dict(re.findall(r'; (\w+): (\d+)', body[k:k + 700]))
        epi, dyn, o8 = int(m.group(2)), m.group(3) == '1', m.group(4) == '1'
        assert int(info['Occupancy']) >= 2 and int(info['NumVgprs']) <= 256, (m.group(1), info)
        if o8:
            # (round 6, fp8 path only: the byte epilogues that also leave the 8-bit copy of their output.  GELU + byte: no
            #  scratch; byte-decode: loop-invariant addresses parked in front of the K loop and fetched back in the epilogue -
            #  a dozen dwords, none of them touched inside the K loop)
            assert epi in (7, 8) and not dyn, m.group(1)
            assert int(info['ScratchSize']) <= (64 if epi == 7 else 0), (m.group(1), info['ScratchSize'])
            fn = body[m.end():body.index('.Lfunc_end', m.end())]
            mf = [x.start() for x in re.finditer(r'v_mfma', fn)]
            assert not re.search(r'scratch_', fn[mf[0]:mf[-1]]), 'scratch access inside the K loop of %s' % m.group(1)
            copies += 1
            continue
        if dyn:
            # the tile-queue instantiations (data parallelism only) carry a dozen registers of bookkeeping: a few dwords of
            # scratch at most, and none exists for the multiply epilogues (they spilled ~100 bytes: the launcher keeps those
            # on the static schedule)
            assert epi <= 4, m.group(1)
            assert int(info['ScratchSize']) <= 32, (m.group(1), info['ScratchSize'])
            queued += 1
        else:
            # (7 - 9: the round-4 epilogues - byte-derivative multiply, bias-GELU + byte, bias + block statistics - no scratch)
            assert int(info['ScratchSize']) <= (32 if epi in (5, 6) else 0), (m.group(1), info['ScratchSize'])
            seen += 1
    assert seen == 10 and queued == 5 and copies == 2, (seen, queued, copies)
    for m in re.finditer(r'^(_ZN\S*gemm_nt_w8f8_kernel\S*):', body, re.M):
        k = body.index('; Kernel info:', body.index('.Lfunc_end', m.end()))
        info = dict(re.findall(r'; (\w+): (\d+)', body[k:k + 700]))
        assert int(info['Occupancy']) >= 2 and int(info['ScratchSize']) <= 320, (m.group(1), info)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_attention_kernels_keep_their_occupancy(tmp_path):
    """No production instantiation of the attention kernels spills to scratch, the M3P-sequence instantiations (S = 164: six
    32-row steps, eleven 16-row tiles) fit three waves per SIMD (three ~48-KB workgroups per CU) and the long-sequence
    ones (eight waves per workgroup) two."""
    out = tmp_path / 'attention.s'
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-ffp-contract=fast',
           '-Wno-unused-result', '--cuda-device-only', '-S', os.path.join(ROOT, 'm3p_amd', 'csrc', 'attention.hip'), '-o', str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    body = out.read_text()
    seen = 0
    for m in re.finditer(r'^(_ZN\S*attn_(fwd|bwd)_kernelI(\S*?)EEv\S*):', body, re.M):
        k = body.index('; Kernel info:', body.index('.Lfunc_end', m.end()))
        info = dict(re.findall(r'; (\w+): (\d+)', body[k:k + 600]))
        args = m.group(3).rstrip('E')
        rehash = m.group(2) == 'bwd' and 'Lb1ELb0E' in args    # dropout without the forward's keep bits: legacy path
        assert info['ScratchSize'] == '0' or rehash, (m.group(1), info['ScratchSize'])
        # (forward: <DH, KT, DROP, NTC = 11, NW = 4, EVENS>; backward: <DH, 16, DROP, MASK, 6, 11, NW = 4, KB, KBQ>)
        if m.group(2) == 'fwd' and 'Li11ELi4ELb' in args or m.group(2) == 'bwd' and 'Li6ELi11ELi4ELi' in args:
            assert int(info['Occupancy']) >= 3, (m.group(1), info['Occupancy'], info['NumVgprs'])
            seen += 1
        if re.search(r'ELi8(ELb[01]|ELi1ELi1)$', args):
            assert int(info['Occupancy']) >= 2, (m.group(1), info['Occupancy'])
    assert seen >= 8, seen


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_no_lds_read_may_overwrite_the_address_of_the_next_one(tmp_path):
    """Two LDS reads from one inline-asm statement share their address register; if the first read's destination includes
    that register (outputs not marked early-clobber) the second read's address is the first read's DATA whenever the wave
    stalls between the two for longer than the LDS latency.  The weight-gradient kernel had 21 such pairs: one wrong
    half-fragment in about one launch out of 300 with freshly allocated operands (tools/wgrad_stress2.py), found by
    tests/test_gemm.py failing once in ~20 runs of the suite.  Checked over every csrc file: consecutive LDS reads that use
    the same address register must not write it."""
    pat = re.compile(r'\bds_read\w*\s+(v\[(\d+):(\d+)\]|v(\d+)),\s+v(\d+)\b')
    n_pairs = 0
    for src in sorted(os.listdir(os.path.join(ROOT, 'm3p_amd', 'csrc'))):
        if not src.endswith('.hip'):
            continue
        out = tmp_path / (src[:-4] + '.s')
        cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-ffp-contract=fast',
               '-Wno-unused-result', '--cuda-device-only', '-S', os.path.join(ROOT, 'm3p_amd', 'csrc', src), '-o', str(out)]
        subprocess.run(cmd, check=True, capture_output=True, timeout=900)
        text = out.read_text().splitlines()
        prev = None
        for i, l in enumerate(text):
            m = pat.search(l)
            if m is None:
                if l.strip() and not l.strip().startswith(';'):
                    prev = None
                continue
            lo, hi = (int(m.group(2)), int(m.group(3))) if m.group(2) else (int(m.group(4)), int(m.group(4)))
            addr = int(m.group(5))
            if prev is not None and prev[2] == addr:
                n_pairs += 1
                assert not (prev[0] <= addr <= prev[1]), '%s:%d: %s  then  %s' % (src, i, text[i - 1].strip(), l.strip())
            prev = (lo, hi, addr)
    assert n_pairs >= 80          # the tr16 pairs of the weight-gradient kernel are seen at all


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_nothing_touches_an_inline_lds_reads_destination_before_its_wait(tmp_path):
    """The four-wave kernels read LDS through inline asm the compiler does not see as loads (fragment reads of the K loop,
    the epilogue's staging rows): the destination registers hold the data only after OUR s_waitcnt lgkmcnt.  To the compiler
    they are defined at once - given control flow between a read and its wait it copies them (phi moves) in front of the
    wait, which is how a pipelined form of the epilogue went wrong in round 4.  Checked on the ISA: between an inline
    ds_read and the next s_waitcnt lgkmcnt no compiler-generated instruction may name a register of its destination."""
    out = tmp_path / 'gemm.s'
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-ffp-contract=fast',
           '-Wno-unused-result', '--cuda-device-only', '-S', os.path.join(ROOT, 'm3p_amd', 'csrc', 'gemm.hip'), '-o', str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    text = out.read_text().splitlines()
    rd = re.compile(r'^\s*ds_read\w*\s+v\[(\d+):(\d+)\]')
    reg = re.compile(r'\bv(?:\[(\d+):(\d+)\]|(\d+))')
    starts = [i for i, l in enumerate(text) if re.match(r'^_ZN\S*gemm_(?:nt|wgrad)_w4_kernel\S*:', l)]
    assert len(starts) >= 9
    n_reads = 0
    for st in starts:
        in_asm, pending = False, []        # pending: destination ranges of inline reads not yet waited for
        for i in range(st + 1, len(text)):
            l = text[i]
            if l.startswith('.Lfunc_end'):
                break
            s = l.strip()
            if 'ASMSTART' in l:
                in_asm = True
                continue
            if 'ASMEND' in l:
                in_asm = False
                continue
            if not s or s.startswith(';') or s.startswith('.'):
                continue
            if s.startswith('s_waitcnt') and 'lgkmcnt' in s:
                pending = []
                continue
            m = rd.match(l)
            if in_asm and m:
                pending.append((int(m.group(1)), int(m.group(2))))
                n_reads += 1
                continue
            if in_asm or not pending:
                continue
            if s.endswith(':') or s.startswith('s_cbranch') or s.startswith('s_branch'):
                # (a label or branch: the linear scan no longer follows the program; reads of the K loop are always waited
                #  for inside their straight-line K-tile, so anything still pending here is a finding)
                assert not pending, '%s: control flow between an inline LDS read and its wait (line %d)' % (text[st].split(':')[0], i)
            for a, b, c in reg.findall(s):
                lo, hi = (int(a), int(b)) if a else (int(c), int(c))
                for plo, phi in pending:
                    assert hi < plo or lo > phi, '%s line %d: `%s` names v[%d:%d] before its lgkmcnt wait' % (
                        text[st].split(':')[0], i, s, plo, phi)
    assert n_reads > 1000


def test_generated_kernel_bodies_are_up_to_date():
    """The two four-wave GEMM bodies in gemm.hip are generated (tools/gen/gen_w4.py: the MFMA / memory-instruction schedule is
    a table there); the committed file must be what the generator makes of the committed templates."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen', 'gen_w4.py'), '--check'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr or r.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_persistent_attention_backward_fits_three_waves_per_simd(tmp_path):
    """attn_bwd_p_kernel runs one twelve-wave workgroup per CU: three waves per SIMD means at most 170 registers and no scratch;
    and its per-head bias sums are the halving butterfly of round 6 (bank-masked v_add_f32_dpp from inline asm - 48 of them per
    instantiation with dropout: two flushes of 16 + 8), not four full butterflies per value."""
    out = tmp_path / 'attention.s'
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-ffp-contract=fast',
           '-Wno-unused-result', '--cuda-device-only', '-S', os.path.join(ROOT, 'm3p_amd', 'csrc', 'attention.hip'), '-o', str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    body = out.read_text()
    found = 0
    for m in re.finditer(r'^(_ZN\S*attn_bwd_p_kernelILb[01]ELb[01]E\S*):', body, re.M):
        end = body.index('.Lfunc_end', m.end())
        k = body.index('; Kernel info:', end)
        info = dict(re.findall(r'; (\w+): (\d+)', body[k:k + 600]))
        assert info['ScratchSize'] == '0' and int(info['NumVgprs']) <= 170, (m.group(1), info['ScratchSize'], info['NumVgprs'])
        text = body[m.end():end]
        masked = len(re.findall(r'v_add_f32_dpp [^\n]*bank_mask:0x(?:3|c|5|a)\b', text))
        assert masked == 48, (m.group(1), masked)
        found += 1
    assert found == 2
