"""Sliding-window (tiled) inference with everything resident on the device (BASELINE.json config 5).

Two reference procedures, restated for the GPU:
  * ``predict_3d_tiled`` — nnU-Net's ``_internal_predict_3D_3Dconv_tiled`` (3D/d_lka_former/network_architecture/neural_network.py:
    292-428) with ``_compute_steps_for_sliding_window`` (:266-290) and the Gaussian importance map ``_get_gaussian`` (:250-263);
  * ``predict_single_case`` — the pancreas evaluation ``test_single_case`` (3D/pancreas_code/test_util.py:45-111): fixed strides,
    softmax per tile, plain averaging.
The reference moves every tile host -> device and its prediction device -> host (test_util.py:88,92; neural_network.py:383-386 unless
``all_in_gpu``); here the padded volume, the score map and the weight map live in HBM for the whole volume (a 288 GB device holds any
clinical volume many times over), tiles are gathered and blended on the device, several tiles go through the network per call, and
only the final label / probability maps leave.  Accumulation is fp32 (the reference's ``all_in_gpu`` branch uses fp16)."""
from __future__ import annotations

import math
from typing import Callable, List, Sequence, Tuple

import torch
import torch.nn.functional as F


def compute_steps_for_sliding_window(patch_size: Sequence[int], image_size: Sequence[int], step_size: float) -> List[List[int]]:
    """neural_network.py:266-290: at most ``patch * step_size`` apart, evenly spread so that the last tile ends at the border."""
    assert all(i >= j for i, j in zip(image_size, patch_size)), "image size must be as large or larger than patch_size"
    assert 0 < step_size <= 1, "step_size must be larger than 0 and smaller or equal to 1"
    target = [i * step_size for i in patch_size]
    num_steps = [int(math.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, patch_size)]
    steps = []
    for dim in range(len(patch_size)):
        max_step = image_size[dim] - patch_size[dim]
        actual = max_step / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(round(actual * i)) for i in range(num_steps[dim])])
    return steps


def gaussian_importance_map(patch_size: Sequence[int], sigma_scale: float = 1. / 8, device=None) -> torch.Tensor:
    """neural_network.py:250-263: a unit impulse at the patch centre filtered with ``scipy.ndimage.gaussian_filter(sigma = size *
    sigma_scale, mode='constant')``, normalised to max 1, zeros replaced by the smallest non-zero value.  The filter is separable and the
    input an impulse, so the map is the outer product of three sampled 1-D kernels (scipy truncates them at 4 sigma)."""
    axes = []
    for n in patch_size:
        sigma = n * sigma_scale
        radius = int(4.0 * sigma + 0.5)
        xs = torch.arange(-radius, radius + 1, dtype=torch.float64)
        k = torch.exp(-0.5 * (xs / sigma) ** 2)
        k = k / k.sum()
        line = torch.zeros(n, dtype=torch.float64)
        c = n // 2
        lo, hi = max(0, c - radius), min(n, c + radius + 1)
        line[lo:hi] = k[lo - (c - radius):hi - (c - radius)]
        axes.append(line)
    g = axes[0]
    for a in axes[1:]:
        g = g.unsqueeze(-1) * a
    g = (g / g.max()).to(torch.float32)
    g[g == 0] = g[g != 0].min()
    return g.to(device) if device is not None else g


def _pad_to_patch(x: torch.Tensor, patch_size: Sequence[int]) -> Tuple[torch.Tensor, Tuple[slice, ...]]:
    """Symmetric constant padding up to the patch size (batchgenerators' ``pad_nd_image(..., 'constant')`` as called at :308, and
    test_util.py:49-71); returns the padded volume and the slicer that undoes it."""
    pads, slicer = [], []
    for n, p in zip(x.shape[-3:], patch_size):
        d = max(p - n, 0)
        lo = d // 2
        pads.append((lo, d - lo))
        slicer.append(slice(lo, lo + n))
    if any(a or b for a, b in pads):
        x = F.pad(x, [v for a, b in reversed(pads) for v in (a, b)], mode="constant", value=0)
    return x, tuple(slicer)


def _run_tiles(net: Callable, data: torch.Tensor, origins: List[Tuple[int, int, int]], patch_size, tile_batch: int, post: Callable,
               score: torch.Tensor, weight: torch.Tensor, tile_weight: torch.Tensor):
    pd, ph, pw = patch_size
    for i in range(0, len(origins), tile_batch):
        chunk = origins[i:i + tile_batch]
        tiles = torch.stack([data[:, x:x + pd, y:y + ph, z:z + pw] for x, y, z in chunk])       # gathered on the device
        pred = net(tiles)
        if isinstance(pred, (list, tuple)):   # deep supervision: the full-resolution head
            pred = pred[0]
        pred = post(pred).float()
        for t, (x, y, z) in enumerate(chunk):   # overlapping tiles of one chunk must be blended one after the other
            score[:, x:x + pd, y:y + ph, z:z + pw] += pred[t] * tile_weight
            weight[x:x + pd, y:y + ph, z:z + pw] += tile_weight


@torch.no_grad()
def predict_3d_tiled(net: Callable, x: torch.Tensor, patch_size: Sequence[int], step_size: float = 0.5, use_gaussian: bool = True,
                     num_classes: int = None, tile_batch: int = 4, nonlin: Callable = None):
    """x: (C, X, Y, Z) on the device.  Returns (predicted_segmentation (X, Y, Z) int64, class_probabilities (K, X, Y, Z) fp32), both on
    the device.  ``nonlin`` = the network's ``inference_apply_nonlin`` (softmax over classes in the reference trainer,
    d_lka_former_trainer_synapse.py:185); mirroring (test-time flips) is left to the caller."""
    assert x.ndim == 4, "x must be (c, x, y, z)"
    nonlin = nonlin if nonlin is not None else (lambda t: torch.softmax(t, 1))
    data, slicer = _pad_to_patch(x, patch_size)
    steps = compute_steps_for_sliding_window(patch_size, data.shape[1:], step_size)
    origins = [(a, b, c) for a in steps[0] for b in steps[1] for c in steps[2]]
    if use_gaussian and len(origins) > 1:
        tw = gaussian_importance_map(patch_size, 1. / 8, device=x.device)
    else:
        tw = torch.ones(tuple(patch_size), device=x.device)
    if num_classes is None:
        probe = net(data[None, :, :patch_size[0], :patch_size[1], :patch_size[2]])
        num_classes = (probe[0] if isinstance(probe, (list, tuple)) else probe).shape[1]
    score = torch.zeros((num_classes,) + tuple(data.shape[1:]), device=x.device, dtype=torch.float32)
    weight = torch.zeros(tuple(data.shape[1:]), device=x.device, dtype=torch.float32)
    _run_tiles(net, data, origins, patch_size, tile_batch, nonlin, score, weight, tw)
    probs = (score / weight)[(slice(None),) + slicer]
    return probs.argmax(0), probs


@torch.no_grad()
def predict_single_case(net: Callable, image: torch.Tensor, stride_xy: int, stride_z: int, patch_size: Sequence[int], num_classes: int = 1,
                        tile_batch: int = 4):
    """test_util.py:45-111.  image: (W, H, D) on the device.  Returns (label_map (W, H, D) int64, score_map (K, W, H, D) fp32)."""
    data, slicer = _pad_to_patch(image[None], patch_size)
    ww, hh, dd = data.shape[1:]
    sx = math.ceil((ww - patch_size[0]) / stride_xy) + 1
    sy = math.ceil((hh - patch_size[1]) / stride_xy) + 1
    sz = math.ceil((dd - patch_size[2]) / stride_z) + 1
    origins = [(min(stride_xy * a, ww - patch_size[0]), min(stride_xy * b, hh - patch_size[1]), min(stride_z * c, dd - patch_size[2]))
               for a in range(sx) for b in range(sy) for c in range(sz)]
    score = torch.zeros((num_classes, ww, hh, dd), device=image.device, dtype=torch.float32)
    cnt = torch.zeros((ww, hh, dd), device=image.device, dtype=torch.float32)
    ones = torch.ones(tuple(patch_size), device=image.device)
    _run_tiles(net, data, origins, patch_size, tile_batch, lambda t: torch.softmax(t, 1), score, cnt, ones)
    score = (score / cnt.unsqueeze(0))[(slice(None),) + slicer]
    return score.argmax(0), score


def num_tiles(image_size: Sequence[int], patch_size: Sequence[int], stride_xy: int, stride_z: int) -> int:
    """Tiles ``predict_single_case`` runs (test_util.py:73-75)."""
    sizes = [max(i, p) for i, p in zip(image_size, patch_size)]
    return (math.ceil((sizes[0] - patch_size[0]) / stride_xy) + 1) * (math.ceil((sizes[1] - patch_size[1]) / stride_xy) + 1) * \
           (math.ceil((sizes[2] - patch_size[2]) / stride_z) + 1)
