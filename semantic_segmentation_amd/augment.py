"""On-device mask augmentation.

The reference augments the masks that feed the generator on the HOST with imgaug
(running_files/train_end2end_jsrt.py:99-112: Fliplr(0.5), CropAndPad(percent=(0, 0.1)), Affine scale (0.8, 1.2) per
axis, translate_percent +-0.1, rotate +-15 deg, shear +-8 deg, applied in random order) and pays a GPU -> CPU -> GPU
round trip per Unet step (:186-190).  Every stage is an affine map of the image plane, so the whole pipeline is ONE
affine map per sample: `MaskAugmenter` samples the stages and their order on the host (a handful of scalars per
sample), composes the 3x3 matrices, and one HIP kernel (gs_affine_warp) warps + re-binarises the batch on the device.

Not bit-compatible with imgaug (different random stream; imgaug resamples after every stage, this resamples once --
less blur), same family and parameter ranges.  Deterministic given the seed.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from .ops import _stream


def _translate(tx, ty):
    return np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], dtype=np.float64)


def _about_centre(m, w, h):
    return _translate(w / 2, h / 2) @ m @ _translate(-w / 2, -h / 2)


def sample_forward_matrix(rng: np.random.Generator, h: int, w: int) -> np.ndarray:
    """3x3 FORWARD map (source pixel coordinates -> augmented coordinates, origin at the top-left corner) of one draw
    of the reference pipeline."""
    stages = []
    if rng.random() < 0.5:                                   # iaa.Fliplr(0.5)
        stages.append(np.array([[-1, 0, w], [0, 1, 0], [0, 0, 1]], dtype=np.float64))
    # iaa.CropAndPad(percent=(0, 0.1)): pad every side by 0..10 % and resize back to (h, w)
    top, right, bottom, left = rng.uniform(0.0, 0.1, size=4)
    nw, nh = w * (1 + left + right), h * (1 + top + bottom)
    stages.append(np.array([[w / nw, 0, left * w * (w / nw)], [0, h / nh, top * h * (h / nh)], [0, 0, 1]], dtype=np.float64))
    sx, sy = rng.uniform(0.8, 1.2, size=2)                   # iaa.Affine(scale={"x": .., "y": ..})
    stages.append(_about_centre(np.diag([sx, sy, 1.0]), w, h))
    tx, ty = rng.uniform(-0.1, 0.1, size=2)                  # iaa.Affine(translate_percent=...)
    stages.append(_translate(tx * w, ty * h))
    a = math.radians(rng.uniform(-15.0, 15.0))               # iaa.Affine(rotate=(-15, 15))
    stages.append(_about_centre(np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]]), w, h))
    s = math.radians(rng.uniform(-8.0, 8.0))                 # iaa.Affine(shear=(-8, 8)): x-shear
    stages.append(_about_centre(np.array([[1, -math.tan(s), 0], [0, 1, 0], [0, 0, 1]], dtype=np.float64), w, h))
    order = rng.permutation(len(stages))                     # random_order=True
    m = np.eye(3)
    for i in order:
        m = stages[i] @ m
    return m


class MaskAugmenter:
    """`mask_augment` callback of steps.unet_step_loss / harness.EndToEndTrainer: float masks [N,1,H,W] on the GPU in,
    augmented {0,1} masks out (threshold 0.1 as train_end2end_jsrt.py:191-193)."""

    def __init__(self, seed: int = 0, threshold: float = 0.1):
        self.rng = np.random.default_rng(seed)
        self.threshold = threshold

    def matrices(self, n: int, h: int, w: int) -> np.ndarray:
        """[n, 6] destination -> source maps (the inverse of the sampled forward maps)."""
        out = np.empty((n, 6), dtype=np.float32)
        for i in range(n):
            inv = np.linalg.inv(sample_forward_matrix(self.rng, h, w))
            out[i] = inv[:2].reshape(6)
        return out

    def __call__(self, mask: torch.Tensor) -> torch.Tensor:
        if not mask.is_cuda:
            raise RuntimeError("MaskAugmenter runs on the MI355X only (no CPU path)")
        squeeze = mask.dim() == 3
        m4 = (mask.unsqueeze(1) if squeeze else mask).contiguous().float()
        n, c, h, w = m4.shape
        mats = torch.from_numpy(np.repeat(self.matrices(n, h, w), 1, axis=0)).to(m4.device)
        out = torch.empty_like(m4)
        _lib.call("gs_affine_warp", m4.data_ptr(), out.data_ptr(), mats.data_ptr(), n, c, h, w, float(self.threshold), _stream())
        return out[:, 0] if squeeze else out


def affine_warp(src: torch.Tensor, mats: torch.Tensor, threshold: float = -1.0) -> torch.Tensor:
    """Raw kernel access: src fp32 [N,C,H,W], mats fp32 [N,6] (destination -> source)."""
    src = src.contiguous().float()
    n, c, h, w = src.shape
    mats = mats.contiguous().float()
    if tuple(mats.shape) != (n, 6) or not src.is_cuda or mats.device != src.device:
        raise ValueError("affine_warp: mats must be [N,6] on the device of src")
    out = torch.empty_like(src)
    _lib.call("gs_affine_warp", src.data_ptr(), out.data_ptr(), mats.data_ptr(), n, c, h, w, float(threshold), _stream())
    return out
