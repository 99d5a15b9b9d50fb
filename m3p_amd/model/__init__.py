"""``build_model`` of the reference's ``src/model/__init__.py`` (:85-132) for the encoder-only
cross-modal model: construct, optionally reload a checkpoint's ``'model'`` entry, move to the GPU.

Reload semantics kept (:96-124): ``module.`` prefixes of a DDP-saved file are stripped; parameters
the file does not hold (heads added after a checkpoint was written, e.g. the CLCM pair) are
BACK-FILLED from the freshly initialised model so the strict load still succeeds; with
``multi_reload_model`` the listed checkpoints are averaged and blended 0.6 / 0.4 with the main
one (:106-122)."""
from logging import getLogger

import torch

logger = getLogger()


def _load_model_entry(path):
    sd = torch.load(path, map_location='cpu', weights_only=False)['model']
    if all(k.startswith('module.') for k in sd):
        sd = {k[len('module.'):]: v for k, v in sd.items()}
    return sd


def build_model(params):
    from .transformer import TransformerModel
    assert getattr(params, 'encoder_only', True), 'the MI355X build is the cross-modal encoder (encoder_only)'
    model = TransformerModel(params, is_encoder=True, is_crossModal=getattr(params, 'is_cross_modal', True), with_output=True)
    path = getattr(params, 'reload_model', '')
    if path != '':
        logger.info('Reloading model from %s ...' % path)
        reloaded = _load_model_entry(path)
        own = model.state_dict()
        for k, v in own.items():                    # back-fill what the file lacks
            if k not in reloaded:
                reloaded[k] = v
        multi = getattr(params, 'multi_reload_model', '')
        if multi != '':
            paths = [s for s in (multi.split(',') if isinstance(multi, str) else multi) if len(s) > 0]
            params.multi_reload_model = paths
            mean = None
            for p in paths:
                sd = _load_model_entry(p)
                mean = {k: v.clone().float() for k, v in sd.items()} if mean is None else \
                    {k: mean[k] + sd[k].float() for k in mean}
            for k in reloaded:
                reloaded[k] = reloaded[k] * 0.6 + (mean[k] / len(paths)).to(reloaded[k].dtype) * 0.4
        model.load_state_dict(reloaded)
    logger.info('Number of parameters (model): %i' % sum(p.numel() for p in model.parameters() if p.requires_grad))
    if torch.cuda.is_available():
        model = model.cuda()
    return model
