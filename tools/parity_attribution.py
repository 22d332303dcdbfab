#!/usr/bin/env python3
"""Where does the 16-bit engine's logit error come from?  

CPU-only simulation: the oracle's U-Net forward (oracle/oracle.py, train-mode BN) with the engine's storage points
rounded per stage -- 16-bit weight packs, raw conv output y, activation z = relu(bn(y)), transposed-conv output -- to
fp16 / bf16 ("16"), to a hi+lo pair of 16-bit values ("split": what the precise mode stores) or not at all ("f32").
BN statistics always come from the un-rounded accumulators, as in the kernels.  Reports max / mean |logit - fp32 logit|:
  * every stage 16-bit (the default engine), every stage split (the precise mode),
  * ONE stage 16-bit, all others exact  -> that stage's own contribution,
  * the last n stages exact, the rest 16-bit -> what a "precise tail" would buy.
  * --wyz (round 3): per stage, which ROUNDING makes the error -- the 16-bit weight pack (w), the stored conv output (y) or
    the stored activation (z) -- one source at a time, everything else exact;
  * --plans (round 3): the pair forward of unet_engine.forward_precise (y and z travel as hi/lo pairs everywhere) with the
    MFMA segments of a plan: "1" = x_hi.w_hi (weights and the conv input rounded), "x" = + x_lo.w_hi, "w" = + x_hi.w_lo,
    "xw" = all three -- the simulation the "mixed" plan (unet_engine.MIXED_XW) was chosen with.
Usage: python tools/parity_attribution.py [--size 256] [--batch 2] [--dtype f16] [--wyz] [--plans] [--out profiles/...json]
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402

STAGES = (["inc.0", "inc.3"] + [f"down{i}.{k}" for i in range(1, 5) for k in (0, 3)] +
          [s for j in range(1, 5) for s in (f"up{j}.up", f"up{j}.conv.0", f"up{j}.conv.3")])


def rounder(mode, tdt):
    if mode == "f32":
        return lambda t: t
    if mode == "16":
        return lambda t: t.to(tdt).float()
    if mode == "split":
        def f(t):
            hi = t.to(tdt)
            return hi.float() + (t - hi.float()).to(tdt).float()
        return f
    raise ValueError(mode)


def forward(sd, x, modes, tdt):
    """modes: stage name -> 'f32' | '16' | 'split'."""
    def stage(name, inp, wkey, bnkey, image=False):
        r = rounder(modes[name], tdt)
        w = sd[wkey] if image else r(sd[wkey])             # the first conv reads fp32 image and fp32 weights
        y = F.conv2d(inp, w, None, padding=1)
        dims = [0, 2, 3]
        mean, var = y.mean(dims), y.var(dims, unbiased=False)        # fp32 accumulators
        yq = r(y)
        inv = torch.rsqrt(var + oracle.BN_EPS) * sd[bnkey + ".weight"]
        z = torch.relu((yq - mean.view(1, -1, 1, 1)) * inv.view(1, -1, 1, 1) + sd[bnkey + ".bias"].view(1, -1, 1, 1))
        return r(z)

    def dconv(name, inp, prefix, image=False):
        p = prefix + ".double_conv"
        z = stage(name + ".0", inp, p + ".0.weight", p + ".1", image)
        return stage(name + ".3", z, p + ".3.weight", p + ".4")

    x1 = dconv("inc", x, "inc", image=True)
    skips = [x1]
    cur = x1
    for i in range(1, 5):
        cur = dconv(f"down{i}", F.max_pool2d(cur, 2), f"down{i}.maxpool_conv.1")
        skips.append(cur)
    y = skips[4]
    for j in range(1, 5):
        r = rounder(modes[f"up{j}.up"], tdt)
        up = r(F.conv_transpose2d(y, r(sd[f"up{j}.up.weight"]), sd[f"up{j}.up.bias"], stride=2))
        y = dconv(f"up{j}.conv", torch.cat([skips[4 - j], up], 1), f"up{j}.conv")
    return F.conv2d(y, sd["outc.conv.weight"], sd["outc.conv.bias"])


def forward_wyz(sd, x, modes, tdt, pair_plan=None):
    """modes: stage -> (w, y, z) rounding modes ('f32' | '16' | 'split').  pair_plan: stage -> "1" | "x" | "w" | "xw": the pair
    forward (y, z split everywhere; the conv input / weights rounded where the plan has no x_lo / w_lo segment)."""
    def cfg(name):
        if pair_plan is None:
            return modes[name] + ("f32",)
        m = pair_plan[name]
        return ("split" if "w" in m else "16", "split", "split", "f32" if "x" in m else "16")

    def stage(name, inp, wkey, bnkey, image=False):
        mw, my, mz, mx = cfg(name)
        w = sd[wkey] if image else rounder(mw, tdt)(sd[wkey])
        if not image:
            inp = rounder(mx, tdt)(inp)
        y = F.conv2d(inp, w, None, padding=1)
        mean, var = y.mean([0, 2, 3]), y.var([0, 2, 3], unbiased=False)
        yq = rounder(my, tdt)(y)
        inv = torch.rsqrt(var + oracle.BN_EPS) * sd[bnkey + ".weight"]
        z = torch.relu((yq - mean.view(1, -1, 1, 1)) * inv.view(1, -1, 1, 1) + sd[bnkey + ".bias"].view(1, -1, 1, 1))
        return rounder(mz, tdt)(z)

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
        mw, my, _, mx = cfg(f"up{j}.up")
        up = rounder(my, tdt)(F.conv_transpose2d(rounder(mx, tdt)(y), rounder(mw, tdt)(sd[f"up{j}.up.weight"]),
                                                 sd[f"up{j}.up.bias"], stride=2))
        y = dconv(f"up{j}.conv", torch.cat([skips[4 - j], up], 1), f"up{j}.conv")
    return F.conv2d(y, sd["outc.conv.weight"], sd["outc.conv.bias"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--quick", action="store_true", help="skip the one-stage-at-a-time sweep")
    ap.add_argument("--wyz", action="store_true", help="split every stage's error into weight / conv-output / activation rounding")
    ap.add_argument("--plans", action="store_true", help="simulate the pair forward under candidate segment plans")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    tdt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    sd = oracle.unet_state_dict(1, a.classes, seed=7)
    x, _ = oracle.synthetic_batch(a.batch, a.size, seed=3)
    with torch.no_grad():
        ref = forward(sd, x, {s: "f32" for s in STAGES}, tdt)
        chk = oracle.unet_forward(sd, x, True)
        assert (ref - chk).abs().max() < 1e-5, "simulation harness != oracle"

        def err(modes):
            d = (forward(sd, x, modes, tdt) - ref).abs()
            return {"max": float(d.max()), "mean": float(d.mean())}

        out = {"config": vars(a), "logit_abs_mean": float(ref.abs().mean()), "logit_abs_max": float(ref.abs().max())}
        out["all_16"] = err({s: "16" for s in STAGES})
        out["all_split"] = err({s: "split" for s in STAGES})
        print("all 16-bit :", out["all_16"], flush=True)
        print("all split  :", out["all_split"], flush=True)
        out["tail_exact"] = {}
        for n in (1, 2, 3, 6, 9, 12):
            modes = {s: ("f32" if i >= len(STAGES) - n else "16") for i, s in enumerate(STAGES)}
            out["tail_exact"][str(n)] = err(modes)
            print(f"last {n:2d} stages exact:", out["tail_exact"][str(n)], flush=True)
        if not a.quick:
            out["one_stage_16"] = {}
            for s in STAGES:
                modes = {t: ("16" if t == s else "f32") for t in STAGES}
                out["one_stage_16"][s] = err(modes)
                print(f"only {s:12s} 16-bit:", out["one_stage_16"][s], flush=True)
        if a.wyz:
            exact = {s: ("f32", "f32", "f32") for s in STAGES}

            def err3(modes):
                d = (forward_wyz(sd, x, modes, tdt) - ref).abs()
                return {"max": float(d.max()), "mean": float(d.mean())}
            out["wyz_everywhere"] = {}
            for idx, which in enumerate("wyz"):
                m = {s: tuple("16" if i == idx else "f32" for i in range(3)) for s in STAGES}
                out["wyz_everywhere"][which] = err3(m)
                print(f"only {which} rounded, every stage:", out["wyz_everywhere"][which], flush=True)
            out["wyz_one_stage"] = {}
            for s in STAGES:
                if s.endswith(".up"):
                    continue
                out["wyz_one_stage"][s] = {}
                for idx, which in enumerate("wyz"):
                    m = dict(exact)
                    m[s] = tuple("16" if i == idx else "f32" for i in range(3))
                    out["wyz_one_stage"][s][which] = err3(m)
                print(f"{s:12s}", {k: "%.2e" % v["mean"] for k, v in out["wyz_one_stage"][s].items()}, flush=True)
        if a.plans:
            from semantic_segmentation_amd.unet.unet_engine import MIXED_XW
            base = {s: ("xw" if s in MIXED_XW else "1") for s in STAGES}

            def errp(plan):
                d = (forward_wyz(sd, x, None, tdt, pair_plan=plan) - ref).abs()
                return {"max": float(d.max()), "mean": float(d.mean())}
            cand = {"pairs, one segment everywhere": {s: "1" for s in STAGES}, "pairs, xw everywhere (precise=True)": {s: "xw" for s in STAGES},
                    "mixed (MIXED_XW)": base,
                    "mixed without down2 / up3": {s: ("xw" if s in MIXED_XW and not s.startswith(("down2", "up3")) else "1") for s in STAGES},
                    "mixed, up3 -> x": {**base, "up3.conv.0": "x", "up3.conv.3": "x"},
                    "mixed, up3 -> 1": {**base, "up3.conv.0": "1", "up3.conv.3": "1"},
                    "mixed, down2 -> x": {**base, "down2.0": "x", "down2.3": "x"},
                    "mixed, x only": {s: ("x" if s in MIXED_XW else "1") for s in STAGES},
                    "mixed, w only": {s: ("w" if s in MIXED_XW else "1") for s in STAGES}}
            out["plans"] = {}
            for k, plan in cand.items():
                out["plans"][k] = errp(plan)
                print(f"{k:40s}", out["plans"][k], flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
