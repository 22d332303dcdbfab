"""UNet3D mixed-plan sweep: which stages need the correction segments?  max |dlogit| on the fixtures + step time at 128^3."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle
from semantic_segmentation_amd.unet3d import UNet3D, unet3d_engine as u3
from semantic_segmentation_amd.losses import seg_loss
G = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
base = u3.resolve_plan3d("mixed", "f16")
def variant(drop):
    p = dict(base)
    for k in drop: p[k] = "1"
    return {k: v for k, v in p.items() if not k.endswith("upconv1")}
plans = {"default": variant([]),
         "no a_block2": variant(["a_block2.conv1", "a_block2.conv2"]),
         "no s_block2": variant(["s_block2.conv1", "s_block2.conv2"]),
         "level 0 only": variant(["a_block2.conv1", "a_block2.conv2", "s_block2.conv1", "s_block2.conv2"]),
         "no s_block2.conv1": variant(["s_block2.conv1"]),
         "no a_block2.conv1, s_block2.conv2": variant(["a_block2.conv1", "s_block2.conv2"])}
for name, plan in plans.items():
    errs = []
    for fx in ("unet3d_c2_16", "unet3d_c1_16", "unet3d_c2_128"):
        z = np.load(os.path.join(G, fx + ".npz"))
        ncls, seed = int(z["n_classes"]), int(z["seed"])
        net = UNet3D(1, ncls, precise=plan); net.load_state_dict(oracle.unet3d_state_dict(1, ncls, seed=seed), strict=True)
        net = net.cuda().train()
        if "x" in z.files:
            x = torch.from_numpy(z["x"]).cuda()
        else:                                               # the 128^3 fixture regenerates its input from a seed
            continue
        with torch.no_grad():
            lg = net(x)
        errs.append(float(np.abs(lg.cpu().numpy() - z["logits"]).max()))
    net = UNet3D(1, 2, precise=plan).cuda().train()
    xv = torch.randn(1, 1, 128, 128, 128, device="cuda"); mv = (torch.rand(1, 128, 128, 128, device="cuda") > 0.5).long()
    def step():
        for p in net.parameters(): p.grad = None
        lg = net(xv); n, c, d, h, w = lg.shape
        seg_loss(lg.reshape(n, c, d * h, w), mv.reshape(n, d * h, w)).backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(4): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 4 * 1e3
    print(f"{name:36s} max |dlogit| {' '.join('%.2e' % e for e in errs)}   128^3 step {ms:6.2f} ms", flush=True)
