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

    def quant_weights(self, arena):
        """8-bit copies of every layer's four weight matrices and of their transposes (the data-gradient operands),
        re-made when the bf16 working copy changed (once per optimizer step)."""
        version = (arena.epoch, arena.master._version)
        if self.weights_version == version and self.weights:
            return
        for i in range(self.n_layers):
            for site, w, wt in (('wqkv', arena.qkv_w16(i), arena.wt[('qkv', i)]),
                                ('wout', arena.w('attentions.%d.out_lin.weight' % i), arena.wt[('out', i)]),
                                ('w1', arena.w('ffns.%d.lin1.weight' % i), arena.wt[('lin1', i)]),
                                ('w2', arena.w('ffns.%d.lin2.weight' % i), arena.wt[('lin2', i)])):
                w8, dsc = self.quant(w, i, site)
                wt8, _ = self.quant(wt, i, site, record=False)        # same values, same scale
                self.weights[(i, site)] = (w8, wt8, dsc.clone())     # roll() rewrites descale in place; these copies keep theirs
        self.weights_version = version
