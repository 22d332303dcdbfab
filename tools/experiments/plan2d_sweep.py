"""U-Net mixed-plan sweep on the MI355X: one MIXED_XW stage at a time set to "1" (or the decoder entries to "xw-"): max / mean |dlogit| on
the reference fixtures (train-mode forward) and the bs 32 256^2 step time.  (Round 3 swept by CPU simulation: profiles/r03_mixed_plan_sweep.txt.)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle
from semantic_segmentation_amd.unet import UNet, unet_engine as ue
from semantic_segmentation_amd.losses import seg_loss
G = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
base = ue.resolve_plan("mixed", "f16")
variants = {"default": {}}
for st in ue.MIXED_XW:
    variants["1 on " + st] = {st: "1"}
variants["xw- on up4.conv.0"] = {"up4.conv.0": "xw-"}
variants["xw- on up3.conv.0"] = {"up3.conv.0": "xw-"}
variants["x on up3.conv.3"] = {"up3.conv.3": "x"}
variants["w on up3.conv.3"] = {"up3.conv.3": "w"}
xb, mb = oracle.synthetic_batch(32, 256, seed=1234)
xb, mb = xb.cuda(), mb.cuda()
for name, ch in variants.items():
    plan = dict(base); plan.update(ch)
    mx, mn = [], []
    for fx in ("unet_c2_128_b4", "unet_c1_64", "unet_c2_64", "unet_c1_odd", "unet_c1_ones"):
        z = np.load(os.path.join(G, fx + ".npz"))
        ncls, seed = int(z["n_classes"]), int(z["seed"])
        net = UNet(1, ncls, precise=plan); net.load_state_dict(oracle.unet_state_dict(1, ncls, seed=seed), strict=True)
        net = net.cuda().train()
        with torch.no_grad():
            d = np.abs(net(torch.from_numpy(z["x"]).cuda()).cpu().numpy() - z["logits"])
        mx.append(float(d.max())); mn.append(float(d.mean()))
    torch.manual_seed(0)
    net = UNet(1, 2, precise=plan).cuda().train()
    def step():
        for p in net.parameters(): p.grad = None
        net.engine.invalidate_packs()
        seg_loss(net(xb), mb).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(8): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 8 * 1e3
    print(f"{name:22s} max {max(mx):.2e} ({' '.join('%.1e' % e for e in mx)}) mean {max(mn):.2e}   step {ms:6.2f} ms", flush=True)
