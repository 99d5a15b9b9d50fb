"""Per-tensor scaling state of the fp8 GEMM path (BASELINE.json configs[3]: "fp8 MFMA GEMMs").

Which products run in 8 bits: the four projections of every encoder layer (fused QKV, attention output, FFN lin1 /
lin2) forward - activations and weights in fp8 e4m3 - and their four data gradients - gradients in bf8 e5m2, transposed
weights in e4m3 - on ``m3p_gemm_nt_fp8`` (fp32 accumulate, bf16 out, every epilogue of the bf16 path).  Weight
gradients, attention, LayerNorm, the embedding assembly and the vocabulary block stay bf16: the reference itself runs
fp16/fp32 only (Apex AMP O1/O2), so this is the build's own recipe, pinned by a loss-curve test against the bf16 path
("fp8 (cfg4): loss-curve parity over N steps within 2 % rather than per-tensor").

Scaling is *delayed*, per tensor site (layer x {x, ctx, x1, hact, four weights, dy2, du, dao, dqkv}): the quantisation
kernel multiplies by the site's scale, saturates, and raises the site's running max |x|; at the start of the next
training forward every site's scale becomes  fmax / (2 * amax)  (a factor 2 of headroom for growth between two
consecutive steps; fmax = 448 for e4m3, 57344 for e5m2).  Everything stays on the device - no host reads.  A site used
for the first time has no history: its max is measured once with a reduction before it is quantised.
"""
import torch

from . import ops

SITES = ('x', 'ctx', 'x1', 'hact', 'wqkv', 'wout', 'w1', 'w2', 'dy2', 'du', 'dao', 'dqkv')
# Which of a layer's eight products actually run in 8 bits, named by their weight.  A product pays for itself only if the
# GEMM saves more than the quantisation pass of its activation / gradient operand costs (both measured on the cfg4 shapes,
# profiles/r02_fp8_gemm_shapes.txt): the three wide or deep forward products and the two data gradients with a 4096- /
# 3072-deep contraction do (+11 .. +43 us each); the two 1024 x 1024 products break even, and the dU product would need
# the derivative pre-computed by the forward GELU pass (+40 us there) to save 18 us.  M3P_FP8_ALL=1: all eight (A/B runs).
import os as _os
# Round 6: the two 4096-deep products take their 8-bit operand from the PRODUCER's epilogue - lin1 + GELU + byte leaves the e4m3
# copy of h for lin2 forward, the byte-decode dU product the e5m2 copy of dU for dx1 (M3PEpilogue::out8) - so they pay no
# quantisation pass at all, and lin1 itself stays on the bf16 GELU + byte kernel (its 8-bit form needed a GELU pass of its own
# and the pre-activation stored for backward: 1.81 + 0.75 ms per cfg4 step).  The default runs exactly these two products in 8
# bits: zero activation / gradient quantisation launches.  M3P_FP8_SITES picks another recipe for A/B runs: "r5" = the round-5
# one (lin1 in 8 bits, GELU pass with the 8-bit copy), "qkv" = the default plus q/k/v forward and its data gradient
# (profiles/r06_fp8_cfg4_ab.txt: all three within the box's noise of each other and of bf16).
_recipe = _os.environ.get('M3P_FP8_SITES', '')
if _os.environ.get('M3P_FP8_ALL', '0') != '0':
    FWD_SITES, BWD_SITES = {'wqkv', 'wout', 'w1', 'w2'}, {'wqkv', 'wout', 'w1', 'w2'}
elif _recipe == 'r5':
    FWD_SITES, BWD_SITES = {'wqkv', 'w1', 'w2'}, {'wqkv', 'w1'}
elif _recipe == 'qkv':      # + the fused q/k/v projection and its data gradient, their operands quantised by passes of their own
    FWD_SITES, BWD_SITES = {'wqkv', 'w2'}, {'wqkv', 'w1'}
else:
    FWD_SITES, BWD_SITES = {'w2'}, {'w1'}
_BF8 = {'dy2', 'du', 'dao', 'dqkv'}
E4M3_MAX, E5M2_MAX = 448.0, 57344.0


class Fp8State:
    def __init__(self, n_layers, device):
        n = n_layers * len(SITES)
        self.n_layers = n_layers
        self.scale = torch.ones(n, dtype=torch.float32, device=device)
        self.descale = torch.ones(n, dtype=torch.float32, device=device)
        self.amax = torch.zeros(n, dtype=torch.float32, device=device)
        self.fmax = torch.tensor([E5M2_MAX if s in _BF8 else E4M3_MAX for _ in range(n_layers) for s in SITES],
                                 dtype=torch.float32, device=device)
        self.seen = [False] * n
        self.weights = {}           # (layer, site) -> (w8, 8-bit transposed copy, descale [1])
        self.weights_version = None
        self._wdesc = None

    def index(self, layer, site):
        return layer * len(SITES) + SITES.index(site)

    def roll(self):
        """New scales from the maxima the last pass recorded (sites that were not used keep theirs)."""
        used = self.amax > 0
        new = torch.where(used, self.fmax / (2.0 * self.amax.clamp_min(1e-30)), self.scale)
        self.scale.copy_(new)
        self.descale.copy_(1.0 / new)
        self.amax.zero_()

    def _first_use(self, x, i):
        amax = x.detach().float().abs().max().clamp_min(1e-30)
        self.scale[i] = self.fmax[i] / (2.0 * amax)
        self.descale[i] = 1.0 / self.scale[i]
        self.seen[i] = True

    def quant(self, x, layer, site, record=True):
        """-> (8-bit tensor, descale device scalar [1]) for bf16 ``x`` at this site."""
        i = self.index(layer, site)
        if not self.seen[i]:
            self._first_use(x, i)
        x8 = ops.quant_fp8(x, scale=self.scale[i:i + 1], amax=self.amax[i:i + 1] if record else None, bf8=site in _BF8)
        return x8, self.descale[i:i + 1]

    def gelu_quant(self, u, layer):
        """hact = gelu(u) together with its 8-bit copy for the lin2 product (one pass instead of GELU + quantisation); the
        site's first use - no scale history yet - takes the two-pass route.  -> (hact, (hact8, descale) or None)"""
        i = self.index(layer, 'hact')
        if not self.seen[i]:
            return ops.gelu_fwd(u), None
        h, h8 = ops.gelu_fwd_q8(u, self.scale[i:i + 1], self.amax[i:i + 1])
        return h, (h8, self.descale[i:i + 1])

    def quant_weights(self, arena):
        """8-bit copies of the layers' weight matrices that run in fp8 and of the transposes their data gradients need,
        re-made when the bf16 working copy changed (once per optimizer step) - ONE batched launch (a descriptor row per
        matrix; 120 separate launches cost more than the 1.3 GB they move)."""
        version = (arena.epoch, arena.master._version)
        if self.weights_version == version and self.weights:
            return
        first = not self.weights
        used = [(i, site) for i in range(self.n_layers) for site in ('wqkv', 'wout', 'w1', 'w2')
                if site in FWD_SITES or site in BWD_SITES]
        if first:
            self._wk = torch.tensor([self.index(i, s) for i, s in used], dtype=torch.long, device=arena.device)
            self._wdsc = torch.empty(len(used), dtype=torch.float32, device=arena.device)
        rows = []
        for j, (i, site) in enumerate(used):
            w, wt = {'wqkv': (arena.qkv_w16(i), arena.wt[('qkv', i)]),
                     'wout': (arena.w('attentions.%d.out_lin.weight' % i), arena.wt[('out', i)]),
                     'w1': (arena.w('ffns.%d.lin1.weight' % i), arena.wt[('lin1', i)]),
                     'w2': (arena.w('ffns.%d.lin2.weight' % i), arena.wt[('lin2', i)])}[site]
            k = self.index(i, site)
            if not self.seen[k]:
                self._first_use(w, k)
            if first:
                w8 = torch.empty(w.shape, dtype=torch.uint8, device=w.device) if site in FWD_SITES else None
                wt8 = torch.empty(wt.shape, dtype=torch.uint8, device=w.device) if site in BWD_SITES else None
                self.weights[(i, site)] = (w8, wt8, self._wdsc[j:j + 1])
            w8, wt8, _ = self.weights[(i, site)]
            assert w.is_contiguous() and wt.is_contiguous() and w.numel() % 8 == 0
            # (only the copies a product reads are made; the matrix's running maximum rides on the first of them)
            if w8 is not None:
                rows.append([w.data_ptr(), w8.data_ptr(), self.scale[k:k + 1].data_ptr(), self.amax[k:k + 1].data_ptr(), w.numel() // 8])
            if wt8 is not None:
                rows.append([wt.data_ptr(), wt8.data_ptr(), self.scale[k:k + 1].data_ptr(),
                             0 if w8 is not None else self.amax[k:k + 1].data_ptr(), wt.numel() // 8])
        # (roll() rewrites descale in place; the copies made now stay valid for the 8-bit weights quantised now)
        torch.index_select(self.descale, 0, self._wk, out=self._wdsc)
        if self._wdesc is None or self._wdesc[1] != rows:
            self._wdesc = (torch.tensor(rows, dtype=torch.int64, device=arena.device), rows)
        ops.quant_fp8_batch(self._wdesc[0], len(rows))
        self.weights_version = version
