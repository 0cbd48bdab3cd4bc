"""Sliding-window inference on the MI355X engine (BASELINE.json configs[4]).

The reference has no inferer (``grep sliding_window|inferer capstone/`` -> 0 hits); this is the companion of the MONAI ``UNet`` it
builds (capstone/volumetric/base_trainer.py:65-72) with the call signature and semantics of MONAI 0.3
``monai.inferers.sliding_window_inference`` / ``SlidingWindowInferer``:

* windows of ``roi_size`` on a regular grid with stride ``int(roi * (1 - overlap))``, the last one of each axis shifted back
  inside the volume; volumes smaller than the ROI are padded symmetrically with ``cval`` and the result cropped back;
* every window prediction is weighted by an importance map (``"constant"``: ones; ``"gaussian"``: an impulse at ``roi // 2``
  blurred by MONAI's erf-integrated Gaussian, sigma = ``sigma_scale * roi``, truncated at 4 sigma, scaled to max 1, zeros
  replaced by the smallest positive weight) and the weighted sum is divided by the sum of weights.

Device work = three C-ABI calls per window batch: ``ctseg_window_gather`` (volume -> the plan's channels-last input, storage
dtype, padding on the fly), the recorded forward program of an inference-only plan, ``ctseg_window_blend`` (weighted
accumulate into the channels-last fp32 output).  The weight normalisation is input independent, so its reciprocal is computed
once on the host and folded into the blend.  No CPU fallback: a predictor that is not backed by the engine raises.
"""
import ctypes
import math
from typing import Sequence, Union

import os

import numpy as np
import torch

from . import _native as nat
from .engine import rup

__all__ = ["sliding_window_inference", "SlidingWindowInferer"]


def _scan_interval(image_size, roi_size, overlap):
    return tuple(r if r == i else max(int(r * (1 - overlap)), 1) for i, r in zip(image_size, roi_size))


def _window_starts(image_size, roi_size, interval):
    """start corners in MONAI's ``dense_patch_slices`` order (last axis fastest)"""
    per_axis = []
    for i, r, s in zip(image_size, roi_size, interval):
        n = int(math.ceil((i - r) / s)) + 1 if i > r else 1
        per_axis.append([min(k * s, i - r) for k in range(n)])
    return [(a, b, c) for a in per_axis[0] for b in per_axis[1] for c in per_axis[2]]


def _gauss_1d(n, sigma):
    """response at 0..n-1 of MONAI's ``gaussian_1d`` (erf-integrated, truncated at 4 sigma) to an impulse at n // 2"""
    tail = int(sigma * 4.0 + 0.5)
    d = torch.arange(n, dtype=torch.float) - (n // 2)
    t = 0.70710678 / sigma
    g = (0.5 * ((t * (d + 0.5)).erf() - (t * (d - 0.5)).erf())).clamp(min=0)      # float32 erf, as MONAI evaluates it
    g[d.abs() > tail] = 0
    return g.numpy()


def _importance_map(roi_size, mode, sigma_scale):
    if mode == "constant":
        return np.ones(roi_size, np.float32)
    if mode != "gaussian":
        raise ValueError(f'mode must be "constant" or "gaussian", got {mode!r}')
    gx, gy, gz = (_gauss_1d(n, sigma_scale * n) if n > 1 else np.ones(1, np.float32) for n in roi_size)
    w = gx[:, None, None] * gy[None, :, None] * gz[None, None, :]
    w = (w / w.max()).astype(np.float32)
    w[w == 0] = w[w != 0].min()
    return w


_MAPS = {}        # (device, volume, roi, window grid, mode, sigma) -> (importance, 1 / sum of weights): input independent


def _blend_maps(dev, img, roi, lo, starts, mode, sigma_scale):
    """importance map and reciprocal weight sum on the device; the sum is accumulated by the blend kernel itself (unit
    logits), window by window in the order the predictions are blended"""
    key = (str(dev), img, roi, starts, mode, float(sigma_scale))
    if key not in _MAPS:
        if len(_MAPS) >= 4:
            _MAPS.pop(next(iter(_MAPS)))
        imp = torch.from_numpy(_importance_map(roi, mode, sigma_scale)).to(dev)
        ones = torch.ones(roi, dtype=torch.float32, device=dev)
        count = torch.zeros(img, dtype=torch.float32, device=dev)
        for a, b, c in starts:
            nat.call("ctseg_window_blend", ones.data_ptr(), 1, 1, *roi, a - lo[0], b - lo[1], c - lo[2], imp.data_ptr(), None,
                     count.data_ptr(), *img, 1)
        _MAPS[key] = (imp, count.reciprocal_())
    return _MAPS[key]


def _engine_of(predictor):
    net = getattr(predictor, "unet", predictor)            # a Base* LightningModule or the UNet itself
    if not hasattr(net, "engine"):
        raise TypeError("sliding_window_inference runs on the MI355X engine: predictor must be a capstone_amd UNet "
                        "(or a module holding one as .unet); arbitrary callables have no device plan")
    return net, net.engine()


LAST_DEVICE_BATCH = None     # windows per forward of the most recent call (reporting only)


def _device_plan(engine, dev, sw_batch_size, n_windows, roi):
    """(plan, windows per forward).  Windows are independent samples (InstanceNorm is per sample) and are blended in scan order
    whatever the grouping, so on a 288 GB device the forward batch need not stay at the caller's ``sw_batch_size`` -- a bound
    chosen for the memory of smaller GPUs that leaves the 12^3-voxel levels of a 4-window batch on 24 workgroups.
    CTSEG_SW_DEVICE_BATCH = "auto" (default: up to 48 windows per forward, halved while the plan does not fit), an integer
    cap, or "0" to run exactly ``sw_batch_size`` windows per forward."""
    want = os.environ.get("CTSEG_SW_DEVICE_BATCH", "auto")
    nb = sw_batch_size
    if dev.type == "cuda" and want != "0":
        cap = 48 if want == "auto" else max(1, int(want))
        nb = max(sw_batch_size, min(n_windows, cap))
    global LAST_DEVICE_BATCH
    while True:
        try:
            plan = engine.plan_for_shape(dev, nb, roi, inference=True)
            LAST_DEVICE_BATCH = nb
            return plan, nb
        except torch.cuda.OutOfMemoryError:
            if nb <= sw_batch_size:
                raise
            torch.cuda.empty_cache()
            nb = max(sw_batch_size, nb // 2)


@torch.no_grad()
def sliding_window_inference(inputs: torch.Tensor, roi_size: Union[Sequence[int], int], sw_batch_size: int, predictor,
                             overlap: float = 0.25, mode: str = "constant", sigma_scale: float = 0.125,
                             padding_mode: str = "constant", cval: float = 0.0) -> torch.Tensor:
    """inputs (1, Cin, H, W[, D]) fp32 on the GPU -> (1, Cout, H, W[, D]) fp32 blended logits (a view of channels-last storage)."""
    net, engine = _engine_of(predictor)
    nat.require_gpu(inputs, "sliding_window_inference")
    nd = net.dimensions
    if inputs.ndim != nd + 2 or inputs.shape[1] != net.in_channels:
        raise ValueError(f"expected inputs (1,{net.in_channels},*spatial[{nd}]), got {tuple(inputs.shape)}")
    if inputs.shape[0] != 1:
        raise NotImplementedError("Currently only inputs with batch size = 1 are supported.")      # as MONAI 0.3
    if not 0 <= overlap < 1:
        raise AssertionError("overlap must be >= 0 and < 1.")
    if str(getattr(padding_mode, "value", padding_mode)) != "constant":
        raise NotImplementedError("only padding_mode='constant' is implemented on the device")
    mode = str(getattr(mode, "value", mode))
    if sw_batch_size < 1:
        raise ValueError("sw_batch_size must be >= 1")

    img = tuple(inputs.shape[2:]) + ((1,) if nd == 2 else ())
    roi = (roi_size,) * nd if isinstance(roi_size, int) else tuple(roi_size)
    if len(roi) != nd:
        raise ValueError(f"roi_size needs {nd} entries")
    roi = tuple(r if r and r > 0 else i for r, i in zip(roi, img)) + ((1,) if nd == 2 else ())    # MONAI fall_back_tuple
    padded = tuple(max(i, r) for i, r in zip(img, roi))
    lo = tuple((p - i) // 2 for p, i in zip(padded, img))
    starts = _window_starts(padded, roi, _scan_interval(padded, roi, overlap))
    dev = inputs.device
    imp_d, inv_count = _blend_maps(dev, img, roi, lo, tuple(starts), mode, sigma_scale)

    plan, sw_batch_size = _device_plan(engine, dev, sw_batch_size, len(starts), roi[:nd])
    vol = inputs.reshape(net.in_channels, *img)
    if vol.dtype != torch.float32 or not vol.is_contiguous():
        vol = vol.float().contiguous()
    C = net.out_channels
    out_ld = rup(C, 4)
    out = torch.zeros(img + (out_ld,), dtype=torch.float32, device=dev)
    xin, logits = plan.x, plan.logits
    win_elems = roi[0] * roi[1] * roi[2]
    esz = xin.t.element_size()
    for g in range(0, len(starts), sw_batch_size):
        batch = starts[g:g + sw_batch_size]               # a short last batch re-runs stale windows; they are not blended
        rel = [[a - lo[0], bb - lo[1], c - lo[2]] for a, bb, c in batch]
        on_gpu = dev.type == "cuda"
        if on_gpu:                                            # one gather launch for the batch
            st_d = torch.tensor(rel, dtype=torch.int32).to(dev, non_blocking=True)
            nat.call("ctseg_window_gather_batch", vol.data_ptr(), net.in_channels, *img, st_d.data_ptr(), len(batch), *roi, float(cval),
                     xin.t.data_ptr(), plan.dt, xin.ld)
        else:
            for b, (a, bb, c) in enumerate(rel):
                nat.call("ctseg_window_gather", vol.data_ptr(), net.in_channels, *img, a, bb, c, *roi,
                         float(cval), xin.t.data_ptr() + b * win_elems * xin.ld * esz, plan.dt, xin.ld)
        plan.forward()
        if on_gpu and logits.ld == out_ld and os.environ.get("CTSEG_SW_BLEND", "batch") != "per_window":
            # one output-centric launch for the whole batch: same sums in the same (scan) order, the output read / written once
            b0 = [max(0, min(r[k] for r in rel)) for k in range(3)]
            b1 = [min(img[k], max(r[k] for r in rel) + roi[k]) for k in range(3)]
            bbox = (ctypes.c_int32 * 6)(*b0, *[b1[k] - b0[k] for k in range(3)])       # voxels the batch can touch
            nat.call("ctseg_window_blend_batch", logits.t.data_ptr(), logits.ld, C, *roi, st_d.data_ptr(), len(batch),
                     imp_d.data_ptr(), inv_count.data_ptr(), out.data_ptr(), *img, out_ld, bbox)
            continue
        for b, (a, bb, c) in enumerate(batch):
            nat.call("ctseg_window_blend", logits.t.data_ptr() + b * win_elems * logits.ld * 4, logits.ld, C, *roi,
                     a - lo[0], bb - lo[1], c - lo[2], imp_d.data_ptr(), inv_count.data_ptr(), out.data_ptr(), *img, out_ld)
    res = out[..., :C].permute(3, 0, 1, 2).unsqueeze(0)
    return res[..., 0] if nd == 2 else res


class SlidingWindowInferer:
    """``monai.inferers.SlidingWindowInferer`` surface: ``inferer(inputs, network)``."""

    def __init__(self, roi_size, sw_batch_size: int = 1, overlap: float = 0.25, mode: str = "constant", sigma_scale: float = 0.125,
                 padding_mode: str = "constant", cval: float = 0.0):
        self.roi_size, self.sw_batch_size, self.overlap, self.mode = roi_size, sw_batch_size, overlap, mode
        self.sigma_scale, self.padding_mode, self.cval = sigma_scale, padding_mode, cval

    def __call__(self, inputs: torch.Tensor, network) -> torch.Tensor:
        return sliding_window_inference(inputs, self.roi_size, self.sw_batch_size, network, self.overlap, self.mode,
                                        self.sigma_scale, self.padding_mode, self.cval)
