"""On-device mask augmentation (semantic_segmentation_amd/augment.py, gs_affine_warp) against a numpy statement of
the same bilinear warp, plus the properties of the sampled pipeline (train_end2end_jsrt.py:99-112, 186-193)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def warp_numpy(src, mats, thresh):
    n, c, h, w = src.shape
    out = np.zeros_like(src)
    ys, xs = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")
    for i in range(n):
        m = mats[i].astype(np.float32)
        sx = (m[0] * xs.astype(np.float32) + m[1] * ys.astype(np.float32) + m[2] - np.float32(0.5)).astype(np.float32)
        sy = (m[3] * xs.astype(np.float32) + m[4] * ys.astype(np.float32) + m[5] - np.float32(0.5)).astype(np.float32)
        x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
        lx, ly = sx - x0, sy - y0

        def tap(yy, xx):
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            return np.where(ok, src[i][:, np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0.0)

        v = (1 - ly) * ((1 - lx) * tap(y0, x0) + lx * tap(y0, x0 + 1)) + ly * ((1 - lx) * tap(y0 + 1, x0) + lx * tap(y0 + 1, x0 + 1))
        out[i] = (v > thresh).astype(np.float32) if thresh >= 0 else v
    return out


def test_affine_warp_matches_numpy():
    from semantic_segmentation_amd.augment import MaskAugmenter, affine_warp
    rng = np.random.default_rng(0)
    src = rng.random((3, 2, 37, 53), dtype=np.float32)
    mats = MaskAugmenter(seed=5).matrices(3, 37, 53)
    got = affine_warp(torch.from_numpy(src).cuda(), torch.from_numpy(mats).cuda(), -1.0).cpu().numpy()
    ref = warp_numpy(src, mats, -1.0)
    assert np.abs(got - ref).max() < 2e-5
    ident = np.tile(np.array([1, 0, 0, 0, 1, 0], dtype=np.float32), (3, 1))
    same = affine_warp(torch.from_numpy(src).cuda(), torch.from_numpy(ident).cuda(), -1.0).cpu().numpy()
    assert np.array_equal(same, src)                                     # identity map: exact copy


def test_mask_augmenter_pipeline_properties():
    from semantic_segmentation_amd.augment import MaskAugmenter
    from semantic_segmentation_amd.harness import SyntheticLungDataset
    ds = SyntheticLungDataset(8, 256, 4)
    mask = torch.stack([ds[i]["mask"] for i in range(8)]).float().cuda()    # [8,1,256,256]
    a1, a2 = MaskAugmenter(seed=11), MaskAugmenter(seed=11)
    o1, o2 = a1(mask), a2(mask)
    assert torch.equal(o1, o2)                                            # deterministic given the seed
    assert set(torch.unique(o1).tolist()) <= {0.0, 1.0}                   # re-binarised (threshold 0.1)
    assert not torch.equal(o1, mask)
    area_in, area_out = mask.mean(dim=(1, 2, 3)), o1.mean(dim=(1, 2, 3))
    ratio = (area_out / area_in).cpu().numpy()
    assert (ratio > 0.4).all() and (ratio < 1.8).all(), ratio            # scale 0.8..1.2 per axis, pad <= 20 %
    assert not torch.equal(a1(mask), o1)                                  # the stream advances
    flat = MaskAugmenter(seed=3)(mask[:, 0])                              # [N,H,W] in -> [N,H,W] out
    assert flat.shape == (8, 256, 256)
    with pytest.raises(RuntimeError):
        MaskAugmenter()(mask.cpu())
