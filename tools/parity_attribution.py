#!/usr/bin/env python3
"""Where does the 16-bit engine's logit error come from?  

CPU-only simulation: the oracle's U-Net forward (oracle/oracle.py, train-mode BN) with the engine's storage points
rounded per stage -- 16-bit weight packs, raw conv output y, activation z = relu(bn(y)), transposed-conv output -- to
fp16 / bf16 ("16"), to a hi+lo pair of 16-bit values ("split": what the precise mode stores) or not at all ("f32").
BN statistics always come from the un-rounded accumulators, as in the kernels.  Reports max / mean |logit - fp32 logit|:
  * every stage 16-bit (the default engine), every stage split (the precise mode),
  * ONE stage 16-bit, all others exact  -> that stage's own contribution,
  * the last n stages exact, the rest 16-bit -> what a "precise tail" would buy.
Usage: python tools/parity_attribution.py [--size 256] [--batch 2] [--dtype f16] [--out profiles/r02_parity_attribution.json]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--quick", action="store_true", help="skip the one-stage-at-a-time sweep")
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
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
