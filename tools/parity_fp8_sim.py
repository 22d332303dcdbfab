#!/usr/bin/env python3
"""CPU simulation of the pair forward with FP8 correction segments.

A stage in mode "q" computes  x_hi.w_hi  (16-bit MFMA, K = C)  +  [q8(x_lo) | q8(x_hi)] . [q8(w_hi) | q8(w_lo)]  (ONE block-scaled
e4m3 MFMA segment, K = 2C at twice the 16-bit rate): the two correction terms of the "xw" form at the cost of one more segment
instead of two.  The simulation rounds exactly what the kernels would round (power-of-two scales, e4m3 saturating at 448) and
reports max / mean |logit - fp32 logit| for the U-Net (oracle weights, synthetic batch) and for UNet3D.

    python tools/parity_fp8_sim.py [--size 256] [--batch 2] [--net unet|unet3d] [--out profiles/r04_fp8_sim.json]
"""
import argparse
import json
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402

E4M3_MAX = 448.0


def r16(t, tdt):
    return t.to(tdt).float()


def split(t, tdt):
    hi = t.to(tdt).float()
    lo = (t - hi).to(tdt).float()
    return hi, lo


def q8(t, k):
    """e4m3 value of t * 2^k (saturating), returned un-scaled"""
    s = 2.0 ** k
    return (t * s).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float() / s


def wscale_exp(w):
    """power of two that puts max |w| into [128, 256): what the pack kernel would derive from the layer's amax"""
    m = float(w.abs().max())
    return 7 - math.ceil(math.log2(m)) if m > 0 else 0


def pair_conv(conv, inp, w, mode, tdt, x_exp=0):
    """conv: callable (x, w) -> y without bias.  inp: fp32 "exact" pair value; w fp32.  x_exp: power-of-two exponent for q8(x_hi)."""
    xh, xl = split(inp, tdt)
    wh, wl = split(w, tdt)
    lo_shift = 11 if tdt == torch.float16 else 8          # lo <= 2^-lo_shift-1 * |hi|
    y = conv(xh, wh)
    if mode == "1":
        return y
    if mode in ("x", "xw"):
        y = y + conv(xl, wh)
    if mode in ("w", "xw"):
        y = y + conv(xh, wl)
    if mode == "q":
        kw = wscale_exp(w)
        y = y + conv(q8(xl, x_exp + lo_shift + 1), q8(wh, kw)) + conv(q8(xh, x_exp), q8(wl, kw + lo_shift + 1))
    if mode == "qx":          # fp8 x_lo correction only
        kw = wscale_exp(w)
        y = y + conv(q8(xl, x_exp + lo_shift + 1), q8(wh, kw))
    return y


# ------------------------------------------------------------------------------------------------ 2-D U-Net
STAGES = (["inc.0", "inc.3"] + [f"down{i}.{k}" for i in range(1, 5) for k in (0, 3)] +
          [s for j in range(1, 5) for s in (f"up{j}.up", f"up{j}.conv.0", f"up{j}.conv.3")])


def unet_forward(sd, x, plan, tdt, bilinear=False):
    """pair forward: y, z stored as hi/lo pairs everywhere; plan[stage] in "1" | "x" | "w" | "xw" | "q" | "f32"."""
    def sp(t):
        a, b = split(t, tdt)
        return a + b

    def stage(name, inp, wkey, bnkey, image=False):
        m = plan[name]
        if image or m == "f32":
            y = F.conv2d(inp, sd[wkey], None, padding=1)
        else:
            y = pair_conv(lambda a, b: F.conv2d(a, b, None, padding=1), inp, sd[wkey], m, tdt)
        mean, var = y.mean([0, 2, 3]), y.var([0, 2, 3], unbiased=False)
        yq = y if m == "f32" else sp(y)
        inv = torch.rsqrt(var + oracle.BN_EPS) * sd[bnkey + ".weight"]
        z = torch.relu((yq - mean.view(1, -1, 1, 1)) * inv.view(1, -1, 1, 1) + sd[bnkey + ".bias"].view(1, -1, 1, 1))
        return z if m == "f32" else sp(z)

    def dconv(name, inp, prefix, image=False):
        p = prefix + ".double_conv"
        z = stage(name + ".0", inp, p + ".0.weight", p + ".1", image)
        return stage(name + ".3", z, p + ".3.weight", p + ".4")

    x1 = dconv("inc", x, "inc", True)
    skips, cur = [x1], x1
    for i in range(1, 5):
        cur = dconv(f"down{i}", F.max_pool2d(cur, 2), f"down{i}.maxpool_conv.1")
        skips.append(cur)
    y = skips[4]
    for j in range(1, 5):
        m = plan[f"up{j}.up"]
        if bilinear:
            up = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True)
            up = up if m == "f32" else sp(up)
        elif m == "f32":
            up = F.conv_transpose2d(y, sd[f"up{j}.up.weight"], sd[f"up{j}.up.bias"], stride=2)
        else:
            up = pair_conv(lambda a, b: F.conv_transpose2d(a, b, None, stride=2), y, sd[f"up{j}.up.weight"], m, tdt)
            up = up + sd[f"up{j}.up.bias"].view(1, -1, 1, 1)
            up = r16(up, tdt) if m == "1" else sp(up)             # a "1" up-conv writes the hi plane only
        y = dconv(f"up{j}.conv", torch.cat([skips[4 - j], up], 1), f"up{j}.conv")
    return F.conv2d(y, sd["outc.conv.weight"], sd["outc.conv.bias"])


# ------------------------------------------------------------------------------------------------ UNet3D
STAGES3D = ([f"a_block{k}.conv{c}" for k in (1, 2, 3) for c in (1, 2)] + ["bottleNeck.conv1", "bottleNeck.conv2"] +
            [s for k in (3, 2, 1) for s in (f"s_block{k}.upconv1", f"s_block{k}.conv1", f"s_block{k}.conv2")])


def unet3d_forward(sd, x, plan, tdt):
    def sp(t):
        a, b = split(t, tdt)
        return a + b

    def bn_relu(y, p, m):
        dims = [0, 2, 3, 4]
        mean, var = y.mean(dims), y.var(dims, unbiased=False)
        yq = y if m == "f32" else sp(y)
        inv = torch.rsqrt(var + oracle.BN_EPS) * sd[p + ".weight"]
        z = torch.relu((yq - mean.view(1, -1, 1, 1, 1)) * inv.view(1, -1, 1, 1, 1) + sd[p + ".bias"].view(1, -1, 1, 1, 1))
        return z if m == "f32" else sp(z)

    def conv(name, h, first=False):
        m = plan[name]
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        if first or m == "f32":
            y = F.conv3d(h, w, None, padding=1)
        else:
            y = pair_conv(lambda a, c: F.conv3d(a, c, None, padding=1), h, w, m, tdt)
        return y + b.view(1, -1, 1, 1, 1), m

    def a_block(h, p, pool, first=False):
        y, m = conv(p + ".conv1", h, first)
        h = bn_relu(y, p + ".bn1", m)
        y, m = conv(p + ".conv2", h)
        res = bn_relu(y, p + ".bn2", m)
        return (F.max_pool3d(res, 2) if pool else res), res

    def s_block(h, res, p, last):
        m = plan[p + ".upconv1"]
        w, b = sd[p + ".upconv1.weight"], sd[p + ".upconv1.bias"]
        if m == "f32":
            u = F.conv_transpose3d(h, w, b, stride=2)
        else:
            u = pair_conv(lambda a, c: F.conv_transpose3d(a, c, None, stride=2), h, w, m, tdt) + b.view(1, -1, 1, 1, 1)
            u = r16(u, tdt) if m == "1" else sp(u)
        h = torch.cat((u, res), 1)
        y, m = conv(p + ".conv1", h)
        h = bn_relu(y, p + ".bn", m)
        y, m = conv(p + ".conv2", h)
        h = bn_relu(y, p + ".bn", m)
        if last:
            h = F.conv3d(h, sd[p + ".conv3.weight"], sd[p + ".conv3.bias"])
        return h

    out, r1 = a_block(x, "a_block1", True, first=True)
    out, r2 = a_block(out, "a_block2", True)
    out, r3 = a_block(out, "a_block3", True)
    out, _ = a_block(out, "bottleNeck", False)
    out = s_block(out, r3, "s_block3", False)
    out = s_block(out, r2, "s_block2", False)
    return s_block(out, r1, "s_block1", True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="unet", choices=["unet", "unet3d"])
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--bilinear", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    tdt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    out = {"config": vars(a)}
    with torch.no_grad():
        if a.net == "unet":
            from semantic_segmentation_amd.unet.unet_engine import MIXED_XW
            size = a.size or 256
            sd = oracle.unet_state_dict(1, 2, seed=7, bilinear=a.bilinear)
            x, _ = oracle.synthetic_batch(a.batch, size, seed=3)
            ref = oracle.unet_forward(sd, x, True, bilinear=a.bilinear)
            fwd = lambda plan: unet_forward(sd, x, plan, tdt, a.bilinear)
            S = STAGES
            mixed = {s: ("xw" if s in MIXED_XW else "1") for s in S}
            cand = {
                "pairs, 1 everywhere": {s: "1" for s in S},
                "mixed (xw on MIXED_XW)": mixed,
                "mixed-q (q on MIXED_XW)": {s: ("q" if s in MIXED_XW else "1") for s in S},
                "q everywhere": {s: "q" for s in S},
                "q on MIXED_XW + down3, up2": {s: ("q" if (s in MIXED_XW or s.startswith(("down3", "up2.conv"))) else "1") for s in S},
                "xw everywhere": {s: "xw" for s in S},
            }
        else:
            size = a.size or 32
            sd = oracle.unet3d_state_dict(1, 2, seed=5)
            g = torch.Generator().manual_seed(11)
            x = torch.randn(a.batch, 1, size, size, size, generator=g)
            ref = oracle.unet3d_forward(sd, x, True)
            fwd = lambda plan: unet3d_forward(sd, x, plan, tdt)
            S = STAGES3D
            lvl01 = [s for s in S if s.startswith(("a_block1", "a_block2", "s_block1", "s_block2")) and "upconv" not in s]
            lvl0 = [s for s in S if s.startswith(("a_block1", "s_block1")) and "upconv" not in s]
            lvl012 = [s for s in S if not s.startswith("bottleNeck") and "upconv" not in s]
            cand = {
                "pairs, 1 everywhere": {s: "1" for s in S},
                "xw level 0": {s: ("xw" if s in lvl0 else "1") for s in S},
                "xw levels 0-1": {s: ("xw" if s in lvl01 else "1") for s in S},
                "xw levels 0-2": {s: ("xw" if s in lvl012 else "1") for s in S},
                "q levels 0-1": {s: ("q" if s in lvl01 else "1") for s in S},
                "q levels 0-2": {s: ("q" if s in lvl012 else "1") for s in S},
                "q everywhere (convs)": {s: ("1" if "upconv" in s else "q") for s in S},
                "q everywhere": {s: "q" for s in S},
                "xw everywhere": {s: "xw" for s in S},
            }
        out["logit_abs_max"] = float(ref.abs().max())
        out["plans"] = {}
        for k, plan in cand.items():
            d = (fwd(plan) - ref).abs()
            out["plans"][k] = {"max": float(d.max()), "mean": float(d.mean())}
            print(f"{k:36s} max {float(d.max()):.3e}  mean {float(d.mean()):.3e}", flush=True)
        # one stage at "1", all others exact (f32): each stage's own contribution in the pair forward
        out["one_stage_1"] = {}
        for s in S:
            plan = {t: ("1" if t == s else "f32") for t in S}
            d = (fwd(plan) - ref).abs()
            out["one_stage_1"][s] = {"max": float(d.max()), "mean": float(d.mean())}
            print(f"only {s:22s} at '1' (rest exact): max {float(d.max()):.3e} mean {float(d.mean()):.3e}", flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
