"""A minimal stand-in for ``pytorch_lightning`` (absent from this image) — just the slice of Lightning 1.0's ``LightningModule``
the reference's modules touch (capstone/volumetric/base_trainer.py:21-132, capstone/training/base_trainer.py:22-148), written
from Lightning 1.0's documented behaviour so the product's ``_Base = pl.LightningModule`` branch can execute in a test:

  * ``save_hyperparameters(*names)`` with FRAME INSPECTION (``core/lightning.py`` + ``utilities/parsing.get_init_args``): the
    arguments are read from the caller's ``__init__`` frame — which must hold a ``__class__`` cell, i.e. use zero-argument
    ``super()`` — and ``**kwargs`` are merged in; ``hparams`` is an attribute dict;
  * ``log(name, value, ..., on_step, on_epoch)``: a no-op outside a Trainer loop (``self._results is None``); inside one the value
    must be a number or a ONE-element tensor (Lightning raises on anything else);
  * ``device``: tracked through ``.to() / .cuda() / .cpu()`` (``DeviceDtypeModuleMixin``), NOT derived from the parameters;
  * checkpoint hooks ``on_save_checkpoint`` / ``on_load_checkpoint`` exist and do nothing.
Test infrastructure only."""
import inspect
import sys
import types

import torch
import torch.nn as nn


class AttributeDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _get_init_args(frame):
    local_vars = frame.f_locals
    if "__class__" not in local_vars:
        return {}
    cls = local_vars["__class__"]
    params = inspect.signature(cls.__init__).parameters
    self_name, *rest = list(params)
    varargs = [n for n, p in params.items() if p.kind == p.VAR_POSITIONAL]
    varkw = [n for n, p in params.items() if p.kind == p.VAR_KEYWORD]
    out = {k: local_vars[k] for k in params if k in local_vars}
    for kw in varkw:
        out.update(out.get(kw, {}))
    for k in [self_name] + varargs + varkw:
        out.pop(k, None)
    return out


class LightningModule(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        self._device = torch.device("cpu")
        self._results = None            # a Trainer loop sets this; tests set it to a list to record log() calls
        self._hparams = AttributeDict()
        self.trainer = None

    # -- hparams ------------------------------------------------------------------------------------------------
    def save_hyperparameters(self, *args, frame=None):
        if not frame:
            frame = inspect.currentframe().f_back
        init_args = _get_init_args(frame)
        assert init_args, "failed to inspect the self init (no zero-argument super() in __init__?)"
        if not args:
            hp = init_args
        else:
            isx_non_str = [i for i, arg in enumerate(args) if not isinstance(arg, str)]
            assert not isx_non_str
            hp = {arg: init_args[arg] for arg in args}        # KeyError if __init__ never received it: as Lightning
        self._hparams = AttributeDict(hp)

    @property
    def hparams(self):
        return self._hparams

    # -- logging ------------------------------------------------------------------------------------------------
    def log(self, name, value, prog_bar=False, logger=True, on_step=None, on_epoch=None, reduce_fx=torch.mean, **kw):
        if self._results is None:
            return
        if torch.is_tensor(value) and value.numel() != 1:
            raise ValueError(f"self.log({name!r}, ...) needs a number or a one-element tensor, got shape {tuple(value.shape)}")
        self._results.append((name, float(value.detach() if torch.is_tensor(value) else value), on_step, on_epoch))

    # -- device -------------------------------------------------------------------------------------------------
    @property
    def device(self):
        return self._device

    def to(self, *args, **kwargs):
        dev = torch._C._nn._parse_to(*args, **kwargs)[0]
        if dev is not None:
            self._device = dev
        return super().to(*args, **kwargs)

    def cuda(self, device=None):
        self._device = torch.device("cuda", device if isinstance(device, int) else torch.cuda.current_device())
        return super().cuda(device)

    def cpu(self):
        self._device = torch.device("cpu")
        return super().cpu()

    # -- hooks --------------------------------------------------------------------------------------------------
    def on_save_checkpoint(self, checkpoint):
        pass

    def on_load_checkpoint(self, checkpoint):
        pass


def seed_everything(seed):
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed


def install():
    """register the stand-in as ``pytorch_lightning`` — BEFORE capstone_amd.volumetric.base_trainer is imported"""
    assert "capstone_amd.volumetric.base_trainer" not in sys.modules, "install() must run before the product module is imported"
    m = types.ModuleType("pytorch_lightning")
    m.LightningModule, m.seed_everything, m.__version__ = LightningModule, seed_everything, "1.0.0-stub"
    m.Trainer = type("Trainer", (), {})
    sys.modules["pytorch_lightning"] = m
    return m
