#!/usr/bin/env python3
"""Generates the body of gemm_nt_w4_kernel (the unrolled MFMA / memory-instruction schedule) into
m3p_amd/csrc/gemm.hip between the W4-BEGIN / W4-END markers.  The schedule is data here:
which MFMA index is followed by which fragment read or LDS-DMA piece."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P2_LOADS = int(os.environ.get('W4_P2_LOADS', '11'))     # LDS-DMAs of K-tile +2 issued in phase 2 (rest: next phase 1)

def mfma_lines(frag_a, frag_w, extra):
    out, n = [], 0
    for i in range(8):
        for j in range(8):
            l = '    W4_M(%s, %s, %d, %d);' % (frag_a, frag_w, i, j)
            if n in extra:
                l += ' ' + ' '.join(extra[n])
            out.append(l)
            n += 1
    return "\n".join(out)

def add(d, n, stmt):
    d.setdefault(n, []).append(stmt)

# phase 1: k-step 0 from registers; reads of k-step 1 at odd MFMA slots 1..31; the tail of the
# previous phase 2's LDS-DMA list in the first slots
ex1 = {}
for r in range(16):
    add(ex1, 2 * r + 1, ('W4_DSR(fw1[%d], rb1, %d);' % (r, r * 2048)) if r < 8 else ('W4_DSR(fa1[%d], ra1, %d);' % (r - 8, (r - 8) * 2048)))
rest = list(range(P2_LOADS, 16))
for k, q in enumerate(rest):
    add(ex1, 4 * k + 2, 'W4_LP(%d);' % q)
# phase 2: k-step 1; reads of the next K-tile's k-step 0 at odd slots 1..31; LDS-DMAs spread over the phase
ex2 = {}
for r in range(16):
    add(ex2, 2 * r + 1, ('W4_DSR(fw0[%d], rb0n, %d);' % (r, r * 2048)) if r < 8 else ('W4_DSR(fa0[%d], ra0n, %d);' % (r - 8, (r - 8) * 2048)))
for k in range(P2_LOADS):
    add(ex2, int(k * 62 / max(P2_LOADS - 1, 1)) // 2 * 2, 'W4_L(%d);' % k)

# ---- weight-gradient kernel: same two phases; a "read" here is one fragment = two tr16 reads
def wg_mfma_lines(yf, xf, extra):
    out, n = [], 0
    for b in range(8):
        for a in range(8):
            l = '    WG_M(%s, %s, %d, %d);' % (yf, xf, b, a)
            if n in extra:
                l += ' ' + ' '.join(extra[n])
            out.append(l)
            n += 1
    return "\n".join(out)

def wg_reads(off, stage):
    r = []
    for c in range(8):
        r.append('WG_YRD(%d, %s, %d);' % (c, stage, 1 if off else 0))
    for c in range(8):
        r.append('WG_TR2(xl[%d], xh[%d], x_addr[%s][%d], %d);' % (c, c, stage, c, off))
    return r

# (weight-gradient kernel: its own knobs - WG_P2_LOADS LDS-DMAs of K-tile +2 in phase 2, a fragment read every WG_READ_STRIDE
#  MFMAs so that the last one is issued well before the phase's closing lgkmcnt(0))
WG_P2 = int(os.environ.get('WG_P2_LOADS', str(P2_LOADS)))
WG_RS = int(os.environ.get('WG_READ_STRIDE', '4'))
wrest = list(range(WG_P2, 16))
wex1 = {}
for k, st in enumerate(wg_reads(16384, 's_cur')):
    add(wex1, WG_RS * k + 1, st)
for k, q in enumerate(wrest):
    add(wex1, 4 * k + 2, 'WG_LP(%d);' % q)
wex2 = {}
for k, st in enumerate(wg_reads(0, 's_cur ^ 1')):
    add(wex2, WG_RS * k + 1, st)
for k in range(WG_P2):
    add(wex2, int(k * 62 / max(WG_P2 - 1, 1)) // 2 * 2, 'WG_L(%d);' % k)
wbody = open(os.path.join(ROOT, 'tools', 'gen', 'wgrad_w4_template.hip')).read()
wbody = wbody.replace('@PHASE1@', wg_mfma_lines('yf0', 'xf0', wex1)).replace('@PHASE2@', wg_mfma_lines('yf1', 'xf1', wex2))

# ---- the NT kernel's one-piece K-tile (M3P_W4_SCHED2): slots are MFMA indices 0..127 (k-step 0 = 0..63)
KT_BAR1 = int(os.environ.get('W4_BAR1', '20'))          # after this MFMA: W of the stage is dead
KT_BAR2 = int(os.environ.get('W4_BAR2', '50'))          # ... and A
KT_BAR3 = int(os.environ.get('W4_BAR3', '107'))         # the next K-tile has landed
KT_DMA_W = [int(x) for x in os.environ.get('W4_DMA_W', '21,26,31,36,41,46,52,57').split(',')]
KT_DMA_A = [int(x) for x in os.environ.get('W4_DMA_A', '62,67,72,77,82,87,92,97').split(',')]
kt = {}
for r in range(8):                                      # W fragments of k-step 1: MFMAs 1..15
    add(kt, 2 * r + 1, 'W4_DSR(fw1[%d], rb1, %d);' % (r, r * 2048))
a_slots = [17, 19] + [23 + 4 * k for k in range(6)]     # A fragments: two before the first barrier, the rest between the W transfers
for r, sl in enumerate(a_slots):
    add(kt, sl, 'W4_DSR(fa1[%d], ra1, %d);' % (r, r * 2048))
n_after = sum(1 for sl in a_slots if sl <= KT_BAR1)
BAR_LAG = int(os.environ.get('W4_BAR_LAG', '0'))        # 1: one MFMA between a wait and its barrier (the pipe has work while the wave waits)
if BAR_LAG:
    add(kt, KT_BAR1, 'W4_WAIT_LGKM(%d);' % n_after); add(kt, KT_BAR1 + 1, 'W4_BAR(); W4_TSEG(0);')
else:
    add(kt, KT_BAR1, 'W4_WAIT_LGKM(%d); W4_BAR(); W4_TSEG(0);' % n_after)
for k, sl in enumerate(KT_DMA_W):
    add(kt, sl, 'W4_L(%d);' % (8 + k))
assert max(a_slots) < KT_BAR2
if BAR_LAG:
    add(kt, KT_BAR2, 'W4_WAIT_LGKM(0);'); add(kt, KT_BAR2 + 1, 'W4_BAR(); W4_TSEG(1);')
else:
    add(kt, KT_BAR2, 'W4_WAIT_LGKM(0); W4_BAR(); W4_TSEG(1);')
for k, sl in enumerate(KT_DMA_A):
    add(kt, sl, 'W4_L(%d);' % k)
add(kt, max(KT_DMA_A) + 1, 'W4_LD();')
assert max(KT_DMA_A) + 1 < KT_BAR3
if BAR_LAG:
    add(kt, KT_BAR3 - 1, 'W4_TSEG(2); W4_WAIT_VM(16); W4_TSEG(3);'); add(kt, KT_BAR3, 'W4_BAR(); W4_TSEG(4);')
else:
    add(kt, KT_BAR3, 'W4_TSEG(2); W4_WAIT_VM(16); W4_TSEG(3); W4_BAR(); W4_TSEG(4);')
for r in range(16):                                     # k-step 0 of the next K-tile
    add(kt, KT_BAR3 + 1 + r, ('W4_DSR(fw0[%d], rb0n, %d);' % (r, r * 2048)) if r < 8 else ('W4_DSR(fa0[%d], ra0n, %d);' % (r - 8, (r - 8) * 2048)))
assert KT_BAR3 + 16 <= 127
add(kt, 127, 'W4_WAIT_LGKM(0); W4_TSEG(5);')
kt_a = {n: v for n, v in kt.items() if n < 64}
kt_b = {n - 64: v for n, v in kt.items() if n >= 64}

# ---- the weight-gradient kernel's one-piece K-tile (M3P_WG_SCHED2): the NT kernel's slots, a "read" = one fragment
wkt = {}
for c in range(8):                                      # first operand's fragments of k-step 1
    add(wkt, 2 * c + 1, 'WG_YRD(%d, s_cur, 1);' % c)
for c, sl in enumerate(a_slots):                        # second operand's: two before the first barrier (= 4 read instructions)
    add(wkt, sl, 'WG_TR2(xl[%d], xh[%d], x_addr[s_cur][%d], 16384);' % (c, c, c))
add(wkt, KT_BAR1, 'WG_WAIT_LGKM(%d); WG_BAR();' % (2 * n_after))
for k, sl in enumerate(KT_DMA_W):
    add(wkt, sl, 'WG_L(%d);' % k)                       # (the first operand's region is pieces 0..7 here)
add(wkt, KT_BAR2, 'WG_WAIT_LGKM(0); WG_SET1(); WG_BAR();')
for k, sl in enumerate(KT_DMA_A):
    add(wkt, sl, 'WG_L(%d);' % (8 + k))
add(wkt, max(KT_DMA_A) + 1, 'WG_LD();')
WGK_BAR3 = int(os.environ.get('WGK_BAR3', str(KT_BAR3)))    # (a fragment is two read instructions here: the tail may want more room)
WGK_RSTEP = int(os.environ.get('WGK_RSTEP', '1'))
assert max(KT_DMA_A) + 1 < WGK_BAR3 and WGK_BAR3 + 1 + 15 * WGK_RSTEP <= 126
add(wkt, WGK_BAR3, 'WG_WAIT_VM(16); WG_BAR();')
for r in range(16):
    add(wkt, WGK_BAR3 + 1 + r * WGK_RSTEP, ('WG_YRD(%d, s_cur ^ 1, 0);' % r) if r < 8 else ('WG_TR2(xl[%d], xh[%d], x_addr[s_cur ^ 1][%d], 0);' % (r - 8, r - 8, r - 8)))
add(wkt, 127, 'WG_WAIT_LGKM(0); WG_SET0();')
wkt_a = {n: v for n, v in wkt.items() if n < 64}
wkt_b = {n - 64: v for n, v in wkt.items() if n >= 64}
wbody = wbody.replace('@WKT_A@', wg_mfma_lines('yf0', 'xf0', wkt_a).replace('\n    ', '\n      ')).replace('@WKT_B@', wg_mfma_lines('yf1', 'xf1', wkt_b).replace('\n    ', '\n      '))

body = open(os.path.join(ROOT, 'tools', 'gen', 'w4_template.hip')).read()
body = body.replace('@PHASE1@', mfma_lines('fa0', 'fw0', ex1)).replace('@PHASE2@', mfma_lines('fa1', 'fw1', ex2))
body = body.replace('@KT_A@', mfma_lines('fa0', 'fw0', kt_a).replace('\n    ', '\n      ')).replace('@KT_B@', mfma_lines('fa1', 'fw1', kt_b).replace('\n    ', '\n      '))
p = os.path.join(ROOT, 'm3p_amd', 'csrc', 'gemm.hip')
s = open(p).read()
a = s.index('// W4-BEGIN')
b = s.index('// W4-END')
s = s[:a] + '// W4-BEGIN (generated by tools/gen/gen_w4.py from tools/gen/w4_template.hip - edit those)\n' + body + s[b:]
a = s.index('// WGW4-BEGIN')
b = s.index('// WGW4-END')
s = s[:a] + '// WGW4-BEGIN (generated by tools/gen/gen_w4.py from tools/gen/wgrad_w4_template.hip - edit those)\n' + wbody + s[b:]
if '--check' in sys.argv:       # (tests: the committed kernel bodies are what the generator makes of the committed templates)
    sys.exit(0 if s == open(p).read() else 'm3p_amd/csrc/gemm.hip is out of date: run tools/gen/gen_w4.py')
open(p, 'w').write(s)
print('generated', len(body.splitlines()), 'lines; phase-2 loads:', P2_LOADS)
