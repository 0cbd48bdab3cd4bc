"""ORACLE (test infrastructure only) — CPU restatement of MONAI 0.3 ``sliding_window_inference``.

PARITY UNPINNED: the reference repository holds no inferer (build-defined feature) and ``monai==0.3``
(reference README.md:39) is not installable here, so this restates the published MONAI 0.3 algorithm
(monai/inferers/utils.py ``sliding_window_inference`` / ``_get_scan_interval``, monai/data/utils.py ``dense_patch_slices`` /
``compute_importance_map``, monai/networks/layers/simplelayers.py ``GaussianFilter`` + ``gaussian_1d``) in the order MONAI
executes it: pad -> slices -> predictor on stacked windows -> ``output += importance * prob``, ``count += importance`` ->
``output / count`` -> crop.  The Gaussian importance map is produced the way MONAI does it — an impulse image pushed through
separable zero-padded 1-D convolutions — NOT by the closed form the product uses, so the two are independent.
"""
import math

import torch
import torch.nn.functional as F


def get_scan_interval(image_size, roi_size, overlap):
    out = []
    for i, r in zip(image_size, roi_size):
        out.append(int(r) if r == i else int(r * (1 - overlap)))
    return tuple(out)


def dense_patch_slices(image_size, patch_size, scan_interval):
    nd = len(image_size)
    scan_num = []
    for i in range(nd):
        if scan_interval[i] == 0:
            scan_num.append(1)
        else:
            num = int(math.ceil(float(image_size[i]) / scan_interval[i]))
            first = next(d for d in range(num) if d * scan_interval[i] + patch_size[i] >= image_size[i])
            scan_num.append(first + 1)

    def axis(i, k):
        s = k * scan_interval[i]
        s -= max(s + patch_size[i] - image_size[i], 0)
        return slice(s, s + patch_size[i])

    slices = []
    if nd == 3:
        for i in range(scan_num[0]):
            for j in range(scan_num[1]):
                for k in range(scan_num[2]):
                    slices.append((axis(0, i), axis(1, j), axis(2, k)))
    else:
        for i in range(scan_num[0]):
            for j in range(scan_num[1]):
                slices.append((axis(0, i), axis(1, j)))
    return slices


def gaussian_1d(sigma, truncated=4.0):
    tail = int(sigma * truncated + 0.5)
    x = torch.arange(-tail, tail + 1, dtype=torch.float)
    t = 0.70710678 / sigma
    return (0.5 * ((t * (x + 0.5)).erf() - (t * (x - 0.5)).erf())).clamp(min=0)


def gaussian_filter(x, sigmas):
    """separable 'same' convolution, zero padded, of x (1,1,*sp)"""
    nd = x.dim() - 2
    conv = (F.conv1d, F.conv2d, F.conv3d)[nd - 1]
    for d, s in enumerate(sigmas):
        k = gaussian_1d(s)
        shape = [1, 1] + [1] * nd
        shape[2 + d] = k.numel()
        pad = [0] * nd
        pad[d] = (k.numel() - 1) // 2
        x = conv(x, k.reshape(shape), padding=pad)
    return x


def compute_importance_map(patch_size, mode="constant", sigma_scale=0.125):
    if mode == "constant":
        return torch.ones(tuple(patch_size), dtype=torch.float)
    center = [i // 2 for i in patch_size]
    sigmas = [i * sigma_scale for i in patch_size]
    imp = torch.zeros(tuple(patch_size))
    imp[tuple(center)] = 1
    imp = gaussian_filter(imp[None, None], sigmas)[0, 0]
    imp = (imp / imp.max()).float()
    imp[imp == 0] = imp[imp != 0].min()
    return imp


def sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap=0.25, mode="constant", sigma_scale=0.125, cval=0.0):
    nd = inputs.dim() - 2
    assert 0 <= overlap < 1 and inputs.shape[0] == 1
    image_size_ = list(inputs.shape[2:])
    roi_size = tuple(r if r and r > 0 else i for r, i in zip(roi_size, image_size_))
    image_size = tuple(max(image_size_[i], roi_size[i]) for i in range(nd))
    pad_size = []
    for k in range(inputs.dim() - 1, 1, -1):
        diff = max(roi_size[k - 2] - inputs.shape[k], 0)
        half = diff // 2
        pad_size.extend([half, diff - half])
    inputs = F.pad(inputs, pad=pad_size, mode="constant", value=cval)
    slices = dense_patch_slices(image_size, roi_size, get_scan_interval(image_size, roi_size, overlap))
    importance = compute_importance_map(roi_size, mode, sigma_scale)
    output, count = None, None
    for g in range(0, len(slices), sw_batch_size):
        sl = slices[g:g + sw_batch_size]
        window = torch.cat([inputs[(slice(None), slice(None)) + s] for s in sl])
        prob = predictor(window)
        if output is None:
            output = torch.zeros((1, prob.shape[1]) + image_size)
            count = torch.zeros((1, prob.shape[1]) + image_size)
        for b, s in enumerate(sl):
            idx = (slice(None), slice(None)) + s
            output[idx] += importance * prob[b:b + 1]
            count[idx] += importance
    output = output / count
    crop = [slice(None), slice(None)]
    for d in range(nd):
        lo = pad_size[2 * (nd - 1 - d)]
        crop.append(slice(lo, lo + image_size_[d]))
    return output[tuple(crop)]
