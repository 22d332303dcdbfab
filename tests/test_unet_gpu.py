"""Whole-network parity: UNet on the HIP engine vs the golden vectors generated from the reference and
vs the CPU oracle on seeded inputs.  GPU only.

Tolerances (north star: 1e-3 on logits / Dice vs the reference CPU path):
  * loss and Dice: |delta| <= 1e-3 asserted in both modes (measured <= 7e-6).
  * logits, DEFAULT mode: the engine stores activations in 16 bits (fp16: 11-bit significand); 23 stacked convolutions
    give mean |delta logit| 5-6e-4 and max 2.3-4.3e-3 in fp16 (3.5e-2 in bf16) -- the same numbers come out of a CPU
    simulation of 16-bit storage (tools/parity_attribution.py), which also shows that the error is made at the
    full-resolution stages.  Asserted at 1.5x the measured values.
  * logits, PRECISE mode (UNet(precise=True): hi/lo 16-bit pairs): max 6.4e-6 - 9.5e-6 (bf16 pairs: 5.2e-5): the north star's
    1e-3 with two orders of magnitude to spare; asserted at 3e-5 (test_precise_mode_*).
  * gradients: 16-bit activations flip the ReLU gate (and the max-pool arg-max) of the ~8e-4 of elements whose
    pre-activation lies within the rounding error of zero; each flip changes its gradient by 100 %, i.e. ~2.8 % relative L2
    per ReLU layer, ~11 % through 18 layers (fp16; ~30 % bf16; 2.4 % with the precise forward).  Whole-net gradients are
    checked (a) against the oracle / fixtures at 1.5x measured and (b) as the first-order descent direction of OUR forward.
"""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import grad_summary
from oracle import oracle

pytestmark = pytest.mark.gpu
REPORT = {}


def _dump():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_unet.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def build_net(n_classes, seed, dtype="f16", bilinear=False, precise=False):
    """precise=False: the FAST 16-bit engine (what most tests of this file were written against, with its bounds); precise=None:
    what `UNet(n_channels, n_classes)` builds -- the "mixed" pair forward, which must meet the north star's 1e-3 on logits."""
    from semantic_segmentation_amd.unet import UNet
    sd = oracle.unet_state_dict(1, n_classes, seed=seed, bilinear=bilinear)
    net = UNet(1, n_classes, bilinear=bilinear, compute_dtype=dtype, precise=precise)
    net.load_state_dict(sd, strict=True)
    return net.cuda(), sd


@pytest.mark.parametrize("mode", ["default", "fast"])
@pytest.mark.parametrize("name", ["unet_c1_64", "unet_c2_64", "unet_c1_odd", "unet_c1_zeros", "unet_c1_ones",
                                  "unet_c2_128_b4", "unet_c1_bilinear_64", "unet_c2_bilinear_odd"])
def test_unet_step_vs_golden(golden_dir, name, mode):
    """Every reference fixture in both numerics modes.  "default" = UNet(...) exactly as the reference's scripts build it
    (train_end2end_jsrt.py:62,67): max |dlogit| < 1e-3, the north star's tolerance (bilinear included).  "fast" =
    UNet(..., precise=False), the 16-bit engine, asserted at 1.5x its measured error."""
    from semantic_segmentation_amd.losses import seg_loss
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    n_classes, seed = int(z["n_classes"]), int(z["seed"])
    bilinear = bool(int(z["bilinear"])) if "bilinear" in z.files else False
    net, sd = build_net(n_classes, seed, bilinear=bilinear, precise=None if mode == "default" else False)
    assert (net.engine.plan is not None) == (mode == "default")
    net.train()
    x = torch.from_numpy(z["x"]).cuda()
    mask = torch.from_numpy(z["mask"].astype(np.int64)).cuda()
    logits = net(x)
    loss, parts = seg_loss(logits, mask, return_parts=True)
    loss.backward()
    torch.cuda.synchronize()
    lg = logits.detach().cpu().numpy()
    d = np.abs(lg - z["logits"])
    rep = {"logit_max_abs": float(d.max()), "logit_mean_abs": float(d.mean()),
           "loss": float(loss.item()), "loss_ref": float(z["loss"]),
           "dice_loss": float(parts[2].item()), "dice_loss_ref": float(z["loss_dice"])}
    worst = 0.0
    for k, p in net.named_parameters():
        ref = z["gsum/" + k]
        got = grad_summary(p.grad.cpu())
        scale = max(ref[1], 1e-12)
        worst = max(worst, abs(got[1] - ref[1]) / scale)
    rep["grad_norm_rel_err_worst"] = worst
    bworst = 0.0
    for k, b in net.named_buffers():
        if "num_batches" in k:
            assert int(b) == 1
        else:
            ref = z["buf/" + k]
            bworst = max(bworst, float(np.abs(b.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-6)))
    rep["bn_buffer_rel_err_worst"] = bworst
    key = name if mode == "fast" else "default_" + name
    REPORT[key] = rep
    _dump()
    assert np.isfinite(lg).all()
    if mode == "default":
        # the north star's bar: measured 5.4e-4 .. 6.7e-4 (bilinear, all stages corrected: 2e-5), loss / Dice <= 1e-5
        assert abs(rep["loss"] - rep["loss_ref"]) < 2e-5 and abs(rep["dice_loss"] - rep["dice_loss_ref"]) < 2e-5, rep
        assert rep["logit_max_abs"] < 1e-3 and rep["logit_mean_abs"] < 1.6e-4, rep
        assert worst < 1.6e-2 and bworst < 6e-4, rep       # gradient norms: measured <= 1.0 %; BN buffers <= 3.8e-4
    else:
        assert abs(rep["loss"] - rep["loss_ref"]) < 1e-3, rep
        assert abs(rep["dice_loss"] - rep["dice_loss_ref"]) < 1e-3, rep
        # FAST mode (single 16-bit storage): asserted at 1.5x the worst value measured on MI355X -- max 4.12e-3 / mean 5.6e-4
        # (bilinear: four more 16-bit roundings and half as many channels to average over: 9.94e-3 / 1.57e-3).  OUTSIDE the north
        # star's 1e-3 on logits (the default mode meets it); loss / Dice meet it in both.
        assert rep["logit_max_abs"] < (1.5e-2 if bilinear else 6.2e-3) and rep["logit_mean_abs"] < (2.4e-3 if bilinear else 8.5e-4), rep
        assert worst < 4e-2, rep                  # gradient norms: measured <= 2.7 %
        assert bworst < 2.6e-3, rep               # measured <= 1.7e-3
    # eval mode with the updated running statistics + evaluate.py Dice
    net.eval()
    with torch.no_grad():
        le = net(x)
    de = np.abs(le.cpu().numpy() - z["logits_eval"])
    REPORT[key]["eval_logit_max_abs"] = float(de.max())
    REPORT[key]["eval_dice_delta"] = abs(float(oracle.evaluate_dice(le.cpu(), mask.cpu())) - float(z["eval_dice"]))
    _dump()
    assert REPORT[key]["eval_dice_delta"] < 1e-3
    # (eval mode with barely-moved running statistics squashes the activations: measured 2.6e-5 .. 3.9e-5 in the fast mode)
    assert de.max() < (1e-5 if mode == "default" else 6e-5) * max(1.0, np.abs(z["logits_eval"]).max())      # measured 2.6e-6 / 3.9e-5


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_unet_vs_oracle_256(dtype):
    """BASELINE config 2 shape (256x256) at batch 2 against the oracle, both compute dtypes."""
    from semantic_segmentation_amd.losses import seg_loss
    net, sd = build_net(2, seed=11, dtype=dtype)
    net.train()
    x, mask = oracle.synthetic_batch(2, 256, seed=99)
    ref_logits, ref_loss, ref_grads, _ = oracle.unet_step(sd, x, mask, train=True)
    logits = net(x.cuda())
    loss = seg_loss(logits, mask.cuda())
    loss.backward()
    torch.cuda.synchronize()
    d = (logits.detach().cpu() - ref_logits).abs()
    rel = {}
    for k, p in net.named_parameters():
        g, r = p.grad.cpu().double(), ref_grads[k].double()
        rel[k] = float((g - r).norm() / max(r.norm().item(), 1e-20))
    worst_k = max(rel, key=rel.get)
    REPORT["oracle256_" + dtype] = {"logit_max_abs": float(d.max()), "logit_mean_abs": float(d.mean()),
                                    "loss": float(loss.item()), "loss_ref": float(ref_loss.item()),
                                    "grad_rel_l2_worst": rel[worst_k], "grad_rel_l2_worst_key": worst_k,
                                    "grad_rel_l2_median": float(np.median(list(rel.values())))}
    _dump()
    assert abs(loss.item() - ref_loss.item()) < 1e-3
    # 1.5x the measured values (f16: max 3.94e-3 / mean 5.3e-4, gradients worst 16.6 %; bf16: 2.63e-2 / 4.19e-3, 45.8 %)
    lim_mean, lim_max = (8e-4, 5.9e-3) if dtype == "f16" else (6.3e-3, 4e-2)
    assert d.mean() < lim_mean and d.max() < lim_max, REPORT
    assert rel[worst_k] < (0.25 if dtype == "f16" else 0.69), (worst_k, rel[worst_k])


def test_unet_forward_is_deterministic_and_retain_graph():
    from semantic_segmentation_amd.losses import seg_loss
    net, _ = build_net(1, seed=12)
    net.train()
    x, mask = oracle.synthetic_batch(2, 64, seed=5)
    x, mask = x.cuda(), mask.cuda()
    a = net(x)
    b = net(x)
    assert torch.equal(a, b)
    loss = seg_loss(a, mask)
    loss.backward(retain_graph=True)
    g1 = {k: p.grad.clone() for k, p in net.named_parameters()}
    net.zero_grad()
    loss.backward()
    for k, p in net.named_parameters():
        # every weight gradient of the U-Net is summed in a fixed order (split-K slabs + ordered reduction for the 3x3
        # convs and the transposed convs, two-stage block reductions elsewhere): bit-reproducible
        assert torch.equal(p.grad, g1[k]), k


@pytest.mark.parametrize("n_classes,B,S", [(2, 4, 128), (1, 3, 80)])
def test_unet_backward_is_bit_reproducible(n_classes, B, S):
    """Two independent forward+backward passes on the same inputs give identical gradients (no fp32 atomics anywhere on
    the U-Net path), at sizes where the weight-gradient launches split K into many parts."""
    from semantic_segmentation_amd.losses import seg_loss
    net, _ = build_net(n_classes, seed=4)
    net.train()
    x, mask = oracle.synthetic_batch(B, S, seed=15)
    x, mask = x.cuda(), mask.cuda()
    runs = []
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        seg_loss(net(x), mask).backward()
        runs.append({k: p.grad.clone() for k, p in net.named_parameters()})
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k


def test_unet_input_gradient():
    net, sd = build_net(1, seed=13)
    net.train()
    x, _ = oracle.synthetic_batch(2, 32, seed=6)
    xc = x.clone().requires_grad_(True)
    p = {k: v.clone() for k, v in sd.items()}
    oracle.unet_forward(p, xc, True).square().mean().backward()
    xg = x.cuda().requires_grad_(True)
    net(xg).square().mean().backward()
    r = (xg.grad.cpu() - xc.grad).norm() / xc.grad.norm()
    assert r < 0.3, float(r)


def test_no_cpu_fallback():
    from semantic_segmentation_amd.unet import UNet
    net = UNet(1, 1)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 1, 32, 32))


@pytest.mark.parametrize("n_classes", [1, 2])
def test_unet_gradient_is_descent_direction(n_classes):
    """loss(w - eps*g) - loss(w) ~= -eps*|g|^2 through OUR forward: validates the whole backward wiring
    (scales, skip/pool routing, BN backward, wgrad/dgrad pairing) independently of activation rounding."""
    from semantic_segmentation_amd.losses import seg_loss
    net, _ = build_net(n_classes, seed=21)
    net.train()
    x, mask = oracle.synthetic_batch(4, 64, seed=8)
    x, mask = x.cuda(), mask.cuda()
    loss0 = seg_loss(net(x), mask)
    loss0.backward()
    g2 = sum(float((p.grad.double() ** 2).sum()) for p in net.parameters())
    ratios = []
    for target in (2e-3, 8e-3):
        eps = target / g2
        with torch.no_grad():
            for p in net.parameters():
                p.add_(p.grad, alpha=-eps)
            loss1 = seg_loss(net(x), mask)
            for p in net.parameters():
                p.add_(p.grad, alpha=eps)
        ratios.append((loss0.item() - loss1.item()) / target)
    REPORT[f"descent_c{n_classes}"] = ratios
    _dump()
    assert 0.8 < ratios[0] < 1.15, ratios
    assert 0.6 < ratios[1] < 1.15, ratios


@pytest.mark.parametrize("mode", ["default", "fast"])
@pytest.mark.parametrize("name", ["unet_c1_64", "unet_c2_64"])
def test_evaluate_matches_reference(golden_dir, name, mode):
    """unet/evaluate.py:10-43 semantics through the compat import path the reference scripts use (both numerics modes: the
    default one evaluates on the folded pair forward)."""
    import semantic_segmentation_amd.compat as compat
    compat.install()
    from unet import UNet                     # noqa: F401  (resolves to this package)
    from unet.evaluate import evaluate
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    n_classes, seed = int(z["n_classes"]), int(z["seed"])
    net, _ = build_net(n_classes, seed, precise=None if mode == "default" else False)
    net.train()
    x = torch.from_numpy(z["x"])
    mask = torch.from_numpy(z["mask"].astype(np.int64))
    net(x.cuda())                              # one train-mode step updates the running statistics like the fixture
    loader = [{"image": x, "mask": mask}]      # same keys as util/JSRT_loader.py:96-101
    score = evaluate(net, loader, torch.device("cuda:0"), amp=True)
    assert net.training                        # evaluate() restores train mode (evaluate.py:42)
    assert abs(float(score) - float(z["eval_dice"])) < 1e-3


@pytest.mark.parametrize("mode", ["default", "fast"])
def test_full_size_properties_bs32_256(mode):
    """Size-independent properties at BASELINE's full configuration (UNet(1,2), 256x256, batch 32), where the oracle
    is too slow to run as a checker:
      * eval-mode shard equivalence: the logits of a sample do not depend on which other samples share its batch
        (bit-exact) -- the property the data-parallel parity definition of SURVEY 8e rests on;
      * the loss gradient is linear in the loss scale: grads(4*loss) == 4*grads(loss) (a power of two passes through
        the 16-bit gradient tensors exactly; the backward itself carries a power-of-two scale S and divides at the end);
      * train-mode statistics: after one step the running mean of the first BatchNorm equals momentum * batch mean
        of the first conv's output, checked against a direct fp32 evaluation of that one layer."""
    from semantic_segmentation_amd.losses import seg_loss
    net, sd = build_net(2, seed=21, precise=None if mode == "default" else False)
    x, mask = oracle.synthetic_batch(32, 256, seed=77)
    x, mask = x.cuda(), mask.cuda()
    net.eval()
    with torch.no_grad():
        full = net(x)
        part = net(x[8:16].contiguous())
        single = net(x[31:32].contiguous())
    assert torch.equal(full[8:16], part)
    assert torch.equal(full[31:32], single)
    net.train()
    g1, g3 = {}, {}
    for scale, store in ((1.0, g1), (4.0, g3)):
        for p in net.parameters():
            p.grad = None
        net.load_state_dict(sd, strict=True)            # same running statistics for both passes
        (seg_loss(net(x), mask) * scale).backward()
        for k, p in net.named_parameters():
            store[k] = p.grad.detach().clone()
    # whole-gradient relative L2 (single tensors such as the up-conv biases, whose exact gradient nearly cancels
    # in front of a BatchNorm, move by ~1e-2 of their own tiny norm when the fp16 subnormal tail shifts by 4x)
    num = sum(float(((g3[k] - 4.0 * g1[k]).double() ** 2).sum()) for k in g1)
    den = sum(float(((4.0 * g1[k]).double() ** 2).sum()) for k in g1)
    assert (num / den) ** 0.5 < 2e-3, (num / den) ** 0.5
    # running mean of inc.double_conv.1 after exactly one training step from zero: 0.1 * batch mean of conv(x)
    net.load_state_dict(sd, strict=True)
    net(x)
    w = sd["inc.double_conv.0.weight"].cuda()
    ref_mean = torch.nn.functional.conv2d(x, w, padding=1).mean(dim=(0, 2, 3)) * 0.1
    got = net.inc.double_conv[1].running_mean
    assert float((got - ref_mean).abs().max()) < 2e-5 + 2e-3 * float(ref_mean.abs().max())


@pytest.mark.parametrize("n_classes,bilinear,H,W", [(2, False, 64, 64), (1, True, 48, 80), (1, False, 33, 47), (2, False, 256, 256)])
def test_pair_inference_folded_bn_matches_two_pass_and_oracle(n_classes, bilinear, H, W):
    """The same for what `UNet()` builds (the pair forward): eval-mode forward with BatchNorm folded into the SEGMENT packs (conv +
    bias + ReLU leave the z pair, `gs_maxpool2x2_fwd_pair` pools it) against the two-pass pair forward and against the oracle --
    the north star's 1e-3 on logits holds for inference too (measured: folded vs two-pass <= 2.3e-6, vs oracle <= 3.5e-6: running
    statistics two steps away from (0, 1) squash the activations)."""
    from semantic_segmentation_amd.unet import unet_engine
    net, sd = build_net(n_classes, seed=13, bilinear=bilinear, precise=None)
    x, mask = oracle.synthetic_batch(3, max(H, W), seed=9)
    x = x[:, :, :H, :W].contiguous().cuda()
    net.train()
    with torch.no_grad():
        for _ in range(2):
            net(x)                                       # move the running statistics away from (0, 1)
    net.eval()
    assert unet_engine.FOLD_BN_INFERENCE
    with torch.no_grad():
        folded = net(x)
    assert any(k.endswith("|fsegs") for k in net.engine._packs), "the folded segment packs were not built"
    unet_engine.FOLD_BN_INFERENCE = False
    try:
        with torch.no_grad():
            two_pass = net(x)
    finally:
        unet_engine.FOLD_BN_INFERENCE = True
    ref = oracle.unet_forward({k: v.detach().cpu() for k, v in net.state_dict().items()}, x.cpu(), False, {}, bilinear)
    scale = max(1.0, float(ref.abs().max()))
    REPORT[f"pair_folded_{n_classes}_{int(bilinear)}_{H}x{W}"] = {
        "folded_vs_two_pass_max": float((folded - two_pass).abs().max()), "folded_vs_oracle_max": float((folded.cpu() - ref).abs().max()),
        "two_pass_vs_oracle_max": float((two_pass.cpu() - ref).abs().max()), "logit_scale": scale}
    assert float((folded - two_pass).abs().max()) < 4e-6 * scale
    assert float((folded.cpu() - ref).abs().max()) < 6e-6 * scale
    # another training step changes the running statistics: the folded packs must follow
    net.train()
    with torch.no_grad():
        net(x)
    net.eval()
    with torch.no_grad():
        again = net(x)
    ref2 = oracle.unet_forward({k: v.detach().cpu() for k, v in net.state_dict().items()}, x.cpu(), False, {}, bilinear)
    assert float((again.cpu() - ref2).abs().max()) < 1e-5 * max(1.0, float(ref2.abs().max()))


@pytest.mark.parametrize("n_classes,bilinear,H,W", [(2, False, 64, 64), (1, True, 48, 80), (1, False, 33, 47)])
def test_inference_folded_bn_matches_two_pass(n_classes, bilinear, H, W):
    """eval-mode forward with BatchNorm folded into the convs (one kernel per conv block, stand-alone
    pool) against the two-pass eval forward (conv -> bn_act_apply with running statistics) and against the oracle."""
    from semantic_segmentation_amd.unet import unet_engine
    net, sd = build_net(n_classes, seed=13, bilinear=bilinear)
    x, mask = oracle.synthetic_batch(3, max(H, W), seed=9)
    x = x[:, :, :H, :W].contiguous().cuda()
    net.train()
    with torch.no_grad():
        for _ in range(2):
            net(x)                                       # move the running statistics away from (0, 1)
    net.eval()
    assert unet_engine.FOLD_BN_INFERENCE
    with torch.no_grad():
        folded = net(x)
    unet_engine.FOLD_BN_INFERENCE = False
    try:
        with torch.no_grad():
            two_pass = net(x)
    finally:
        unet_engine.FOLD_BN_INFERENCE = True
    ref = oracle.unet_forward({k: v.detach().cpu() for k, v in net.state_dict().items()}, x.cpu(), False, {}, bilinear)
    scale = max(1.0, float(ref.abs().max()))
    assert float((folded - two_pass).abs().max()) < 1e-2 * scale
    assert float((folded.cpu() - ref).abs().mean()) < 2.5e-3 * scale
    assert float((folded.cpu() - ref).abs().max()) < 2e-2 * scale
    # the weight packs are cached per (parameter, running statistics) version: another training step must refresh them
    net.train()
    with torch.no_grad():
        net(x)
    net.eval()
    with torch.no_grad():
        again = net(x)
    ref2 = oracle.unet_forward({k: v.detach().cpu() for k, v in net.state_dict().items()}, x.cpu(), False, {}, bilinear)
    assert float((again.cpu() - ref2).abs().max()) < 2e-2 * max(1.0, float(ref2.abs().max()))


@pytest.mark.parametrize("N,H,W,C", [(1, 16, 16, 1), (3, 17, 31, 2), (1, 48, 200, 3), (2, 384, 512, 1)])
def test_unet_extreme_shapes_vs_oracle(N, H, W, C):
    """Edge shapes of the reference's `forward` (unet_model.py:26-37): the minimum 16x16 (1x1 bottleneck), odd sizes
    that need the F.pad of Up at every level, a wide strip, and a large non-square frame; batch 1 and a 3-class head.
    Train-mode forward + loss against the oracle; the backward must be finite and descend."""
    from semantic_segmentation_amd.losses import seg_loss
    net, sd = build_net(C, seed=31 + C)
    net.train()
    x, mask = oracle.synthetic_batch(N, max(H, W), seed=5 + H)
    x, mask = x[:, :, :H, :W].contiguous(), mask[:, :, :H, :W].contiguous()
    if C == 3:
        mask = (mask + (x[:, :1] > 0.8).long()).clamp_(0, 2)             # three classes present
    ref_logits, ref_loss, ref_grads, _ = oracle.unet_step(sd, x, mask, train=True)
    logits = net(x.cuda())
    loss = seg_loss(logits, mask.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert logits.shape == (N, C, H, W)
    d = (logits.detach().cpu() - ref_logits).abs()
    scale = max(1.0, float(ref_logits.abs().max()))
    assert abs(loss.item() - ref_loss.item()) < 2e-3 * max(1.0, abs(ref_loss.item())), (loss.item(), ref_loss.item())
    # a 1x1 bottleneck with batch 1 normalises a single value per channel: BN output is exactly beta there
    assert float(d.mean()) < 3e-3 * scale and float(d.max()) < 3e-2 * scale, (float(d.mean()), float(d.max()))
    num = den = 0.0
    for k, p in net.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        num += float((p.grad.cpu().double() * ref_grads[k].double()).sum())
        den += float((ref_grads[k].double() ** 2).sum())
    assert num / den > 0.5, num / den                    # projection on the fp32 gradient: same direction, same scale


def test_inference_forward_is_hipgraph_capturable():
    """The eval-mode forward (folded BatchNorm: ~27 launches through the C ABI on torch's current stream, no host-side
    dependence on device values) can be captured in a HIP graph and replayed on new inputs with identical results."""
    net, _ = build_net(2, seed=17)
    net.eval()
    x = torch.randn(2, 1, 64, 96, device="cuda")
    with torch.no_grad():
        for _ in range(2):
            net(x)                                   # warm-up: weight packs, workspaces
        torch.cuda.synchronize()
        sx = x.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            sy = net(sx)
        x2 = torch.randn(2, 1, 64, 96, device="cuda")
        sx.copy_(x2)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(sy, net(x2))


# ------------------------------------------------------------------------------------------------ precise mode
def build_precise(n_classes, seed, dtype="f16"):
    from semantic_segmentation_amd.unet import UNet
    sd = oracle.unet_state_dict(1, n_classes, seed=seed)
    net = UNet(1, n_classes, compute_dtype=dtype, precise=True)
    net.load_state_dict(sd, strict=True)
    return net.cuda(), sd


@pytest.mark.parametrize("name", ["unet_c2_128_b4", "unet_c1_64", "unet_c2_64", "unet_c1_odd", "unet_c1_zeros"])
def test_precise_mode_meets_the_north_star_bound_vs_golden(golden_dir, name):
    """UNet(precise=True) -- hi/lo 16-bit pairs -- against the reference-generated fixtures: max |dlogit| below the north
    star's 1e-3 (asserted at 3e-5 = 3x the measured 6.4e-6 .. 8.9e-6), loss / Dice / running statistics tight, eval mode too."""
    from semantic_segmentation_amd.losses import seg_loss
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    n_classes, seed = int(z["n_classes"]), int(z["seed"])
    net, sd = build_precise(n_classes, seed)
    net.train()
    x = torch.from_numpy(z["x"]).cuda()
    mask = torch.from_numpy(z["mask"].astype(np.int64)).cuda()
    logits = net(x)
    loss, parts = seg_loss(logits, mask, return_parts=True)
    loss.backward()
    torch.cuda.synchronize()
    d = np.abs(logits.detach().cpu().numpy() - z["logits"])
    worst = 0.0
    for k, p in net.named_parameters():
        ref = z["gsum/" + k]
        worst = max(worst, abs(grad_summary(p.grad.cpu())[1] - ref[1]) / max(ref[1], 1e-12))
    bworst = 0.0
    for k, b in net.named_buffers():
        if "num_batches" not in k:
            ref = z["buf/" + k]
            bworst = max(bworst, float(np.abs(b.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-6)))
    net.eval()
    with torch.no_grad():
        le = net(x)
    de = np.abs(le.cpu().numpy() - z["logits_eval"])
    REPORT["precise_" + name] = {"logit_max_abs": float(d.max()), "logit_mean_abs": float(d.mean()),
                                 "loss_abs_err": abs(float(loss.item()) - float(z["loss"])),
                                 "dice_abs_err": abs(float(parts[2].item()) - float(z["loss_dice"])),
                                 "grad_norm_rel_err_worst": worst, "bn_buffer_rel_err_worst": bworst,
                                 "eval_logit_max_abs": float(de.max())}
    _dump()
    assert d.max() < 3e-5 and d.mean() < 4e-6, REPORT["precise_" + name]
    assert abs(float(loss.item()) - float(z["loss"])) < 2e-5
    assert abs(float(parts[2].item()) - float(z["loss_dice"])) < 2e-5
    assert bworst < 1e-4
    assert worst < 8e-2                      # backward = the default 16-bit one
    assert de.max() < 1e-4 * max(1.0, np.abs(z["logits_eval"]).max())


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_precise_mode_vs_oracle_256(dtype):
    """256x256 (BASELINE config 2 shape) batch 2 against the oracle in precise mode."""
    from semantic_segmentation_amd.losses import seg_loss
    net, sd = build_precise(2, seed=11, dtype=dtype)
    net.train()
    x, mask = oracle.synthetic_batch(2, 256, seed=99)
    ref_logits, ref_loss, ref_grads, _ = oracle.unet_step(sd, x, mask, train=True)
    logits = net(x.cuda())
    loss = seg_loss(logits, mask.cuda())
    loss.backward()
    torch.cuda.synchronize()
    d = (logits.detach().cpu() - ref_logits).abs()
    rel = {}
    for k, p in net.named_parameters():
        g, r = p.grad.cpu().double(), ref_grads[k].double()
        rel[k] = float((g - r).norm() / max(r.norm().item(), 1e-20))
    REPORT["precise_oracle256_" + dtype] = {"logit_max_abs": float(d.max()), "logit_mean_abs": float(d.mean()),
                                            "loss_abs_err": abs(float(loss.item()) - float(ref_loss.item())),
                                            "grad_rel_l2_median": float(np.median(list(rel.values()))),
                                            "grad_rel_l2_worst": max(rel.values())}
    _dump()
    # bf16 pairs carry 16 significand bits: ~2^-17 per rounding instead of 2^-23.  Measured: f16 max 9.5e-6 / mean 1.3e-6,
    # bf16 max 5.2e-5 / mean 8.1e-6 -- both inside the north star's 1e-3; asserted at ~3x the measurement
    lim_max, lim_mean = (3e-5, 4e-6) if dtype == "f16" else (1.5e-4, 2.5e-5)
    assert d.max() < lim_max and d.mean() < lim_mean, REPORT["precise_oracle256_" + dtype]
    assert abs(loss.item() - ref_loss.item()) < 2e-5
    assert max(rel.values()) < (0.1 if dtype == "f16" else 0.25)      # measured 4.3 % / 11.5 % worst, 2.4 % / 7.0 % median


def test_precise_mode_is_deterministic_and_matches_default_backward_path():
    """Two precise passes are bit-identical (logits and every gradient); eval-mode precise logits agree with the default
    engine within the default engine's own error."""
    from semantic_segmentation_amd.losses import seg_loss
    net, sd = build_precise(2, seed=5)
    net.train()
    x, mask = oracle.synthetic_batch(2, 64, seed=8)
    x, mask = x.cuda(), mask.cuda()
    outs = []
    for _ in range(2):
        net.load_state_dict(sd, strict=True)
        net.zero_grad(set_to_none=True)
        lg = net(x)
        seg_loss(lg, mask).backward()
        outs.append((lg.detach().clone(), [p.grad.clone() for p in net.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
    ref, _ = build_net(2, seed=5)
    ref.train()
    assert (ref(x) - outs[0][0]).abs().max() < 1e-2


@pytest.mark.parametrize("precise", [None, False])
def test_data_writes_between_training_forwards_are_seen_by_default(precise):
    """(precise=None: the default pair forward -- its segment packs are rebuilt too; False: the fast 16-bit engine.)
    Betty's darts hypergradient perturbs the parameters IN PLACE THROUGH `p.data` between two forward/backward passes
    (running_files/train_end2end_jsrt.py:287-292, Config(type="darts", roll_back=True)) -- a write no version counter sees.
    With default settings a training forward must use the perturbed weights (the 16-bit packs are rebuilt by every forward
    that keeps a graph), and the gradients must be those of the perturbed network."""
    from semantic_segmentation_amd.losses import seg_loss
    net, sd = build_net(2, seed=23, precise=precise)
    net.train()
    x, mask = oracle.synthetic_batch(2, 64, seed=6)
    x, mask = x.cuda(), mask.cuda()
    l0 = net(x)
    seg_loss(l0, mask).backward()
    g = torch.Generator(device="cuda").manual_seed(3)
    versions = [p._version for p in net.parameters()]
    for p in net.parameters():
        p.data.add_(torch.randn(p.shape, generator=g, device="cuda") * 0.05 * p.data.abs().mean())     # the darts "+eps * v"
    assert versions == [p._version for p in net.parameters()], "the test must not bump the version counters"
    net.zero_grad(set_to_none=True)
    l1 = net(x)
    seg_loss(l1, mask).backward()
    g1 = [p.grad.clone() for p in net.parameters()]
    assert (l1 - l0).abs().max() > 1e-3, "perturbation too small to tell"
    # a fresh module holding the same (perturbed) values
    from semantic_segmentation_amd.unet import UNet
    ref = UNet(1, 2, precise=precise).cuda()
    ref.load_state_dict({k: v.detach().clone() for k, v in net.state_dict().items()}, strict=True)
    for k, b in ref.named_buffers():                     # same running statistics as `net` had BEFORE its second pass is irrelevant:
        pass                                             # train-mode logits do not read them
    ref.train()
    lr = ref(x)
    seg_loss(lr, mask).backward()
    assert torch.equal(l1, lr)
    assert all(torch.equal(a, p.grad) for a, p in zip(g1, ref.parameters()))
    # and the explicit opt-out still exists for loops that own their updates
    # (only the packed MFMA weights are cached; biases, BatchNorm parameters and the stem's fp32 weights are read directly)
    net.engine.trust_versions = True
    net(x)                                               # fills the version-keyed cache
    net.down1.maxpool_conv[1].double_conv[0].weight.data.mul_(1.5)
    l2 = net(x)
    assert torch.equal(l2, l1), "trust_versions=True reuses the version-keyed packs (documented opt-in)"
    net.engine.trust_versions = False
    l3 = net(x)
    assert not torch.equal(l3, l1)


@pytest.mark.parametrize("n_channels,n_classes,need_dx", [(6, 7, False), (8, 2, True), (3, 9, False)])
def test_unet_wide_ends_vs_oracle(n_channels, n_classes, need_dx):
    """`UNet(n_channels, n_classes)` takes ANY channel / class count in the reference (unet/unet_model.py:8-12); the direct end
    kernels cover 1..4.  More input channels enter the MFMA path as a zero-padded 16-bit NHWC image, more classes run the
    pointwise head in groups of four: forward, loss and every gradient (incl. the image gradient) against the oracle."""
    from semantic_segmentation_amd.losses import seg_loss
    from semantic_segmentation_amd.unet import UNet
    sd = oracle.unet_state_dict(n_channels, n_classes, seed=41)
    net = UNet(n_channels, n_classes)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, n_channels, 48, 64, generator=g)
    mask = torch.randint(0, n_classes, (2, 1, 48, 64), generator=g)
    xr = x.clone().requires_grad_(need_dx)
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    ref_logits = oracle.unet_forward(params, xr, True, {})
    ref_loss = oracle.seg_loss(ref_logits, mask)
    leaves = {k: v for k, v in params.items() if v.requires_grad}
    ref_g = torch.autograd.grad(ref_loss, list(leaves.values()) + ([xr] if need_dx else []))
    ref_grads = dict(zip(leaves.keys(), ref_g))
    xd = x.cuda().requires_grad_(need_dx)
    logits = net(xd)
    loss = seg_loss(logits, mask.cuda())
    loss.backward()
    torch.cuda.synchronize()
    d = (logits.detach().cpu() - ref_logits.detach()).abs()
    rel = {k: float((p.grad.cpu().double() - ref_grads[k].double()).norm() / max(ref_grads[k].double().norm().item(), 1e-20))
           for k, p in net.named_parameters()}
    REPORT[f"wide_{n_channels}_{n_classes}"] = {"logit_max_abs": float(d.max()), "logit_mean_abs": float(d.mean()),
                                                "loss_abs_err": abs(loss.item() - ref_loss.item()),
                                                "grad_rel_l2_worst": max(rel.values()), "worst_key": max(rel, key=rel.get)}
    _dump()
    assert tuple(logits.shape) == (2, n_classes, 48, 64)
    assert abs(loss.item() - ref_loss.item()) < 1e-3
    assert d.mean() < 9.5e-4 and d.max() < 8.2e-3, REPORT[f"wide_{n_channels}_{n_classes}"]      # 1.5x measured (6.35e-4 / 5.45e-3)
    assert max(rel.values()) < 0.27, (max(rel, key=rel.get), max(rel.values()))                      # measured 17.7 %
    if need_dx:
        gx, rx = xd.grad.cpu().double(), ref_g[-1].double()
        assert float((gx - rx).norm() / rx.norm()) < 0.35


# ------------------------------------------------------------------------------------------------ mixed mode: 1e-3 on logits
def build_mode(n_classes, seed, mode, dtype="f16"):
    from semantic_segmentation_amd.unet import UNet
    sd = oracle.unet_state_dict(1, n_classes, seed=seed)
    net = UNet(1, n_classes, compute_dtype=dtype, precise=mode)
    net.load_state_dict(sd, strict=True)
    return net.cuda(), sd


@pytest.mark.parametrize("name", ["unet_c2_128_b4", "unet_c1_64", "unet_c2_64", "unet_c1_odd"])
def test_mixed_mode_meets_1e3_vs_golden(golden_dir, name):
    """UNet(precise="mixed") -- pairs everywhere, three MFMA segments on the nine stages that make the 16-bit error
    (unet_engine.MIXED_XW) -- against the reference-generated fixtures: the north star's max |dlogit| < 1e-3, asserted AT 1e-3
    (CPU simulation of the plan: 5.6e-4 .. 6.5e-4), loss / Dice at 2e-5."""
    from semantic_segmentation_amd.losses import seg_loss
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    n_classes, seed = int(z["n_classes"]), int(z["seed"])
    net, sd = build_mode(n_classes, seed, "mixed")
    assert sum(v == "xw" for v in net.engine.plan.values()) == 9
    net.train()
    x = torch.from_numpy(z["x"]).cuda()
    mask = torch.from_numpy(z["mask"].astype(np.int64)).cuda()
    logits = net(x)
    loss, parts = seg_loss(logits, mask, return_parts=True)
    loss.backward()
    torch.cuda.synchronize()
    d = np.abs(logits.detach().cpu().numpy() - z["logits"])
    worst = 0.0
    for k, p in net.named_parameters():
        ref = z["gsum/" + k]
        worst = max(worst, abs(grad_summary(p.grad.cpu())[1] - ref[1]) / max(ref[1], 1e-12))
    bworst = 0.0
    for k, b in net.named_buffers():
        if "num_batches" not in k:
            ref = z["buf/" + k]
            bworst = max(bworst, float(np.abs(b.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-6)))
    REPORT["mixed_" + name] = {"logit_max_abs": float(d.max()), "logit_mean_abs": float(d.mean()),
                               "loss_abs_err": abs(float(loss.item()) - float(z["loss"])),
                               "dice_abs_err": abs(float(parts[2].item()) - float(z["loss_dice"])),
                               "grad_norm_rel_err_worst": worst, "bn_buffer_rel_err_worst": bworst}
    _dump()
    assert d.max() < 1e-3 and d.mean() < 1.6e-4, REPORT["mixed_" + name]      # measured 5.4e-4 .. 6.7e-4 / 1.05e-4
    assert abs(float(loss.item()) - float(z["loss"])) < 2e-5
    assert abs(float(parts[2].item()) - float(z["loss_dice"])) < 2e-5
    assert bworst < 5.7e-4                   # measured <= 3.8e-4
    assert worst < 1.5e-2                    # gradient norms, measured <= 1.0 % (backward = the 16-bit one on the hi planes)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_mixed_mode_meets_1e3_vs_oracle_256(dtype):
    """256x256 (BASELINE config 2 shape) batch 2 against the oracle: max |dlogit| < 1e-3 in BOTH 16-bit dtypes (for bf16
    "mixed" resolves to three segments on every stage: bf16 pairs carry 16 significand bits)."""
    from semantic_segmentation_amd.losses import seg_loss
    net, sd = build_mode(2, 11, "mixed", dtype)
    net.train()
    x, mask = oracle.synthetic_batch(2, 256, seed=99)
    ref_logits, ref_loss, ref_grads, _ = oracle.unet_step(sd, x, mask, train=True)
    logits = net(x.cuda())
    loss = seg_loss(logits, mask.cuda())
    loss.backward()
    torch.cuda.synchronize()
    d = (logits.detach().cpu() - ref_logits).abs()
    rel = {k: float((p.grad.cpu().double() - ref_grads[k].double()).norm() / max(ref_grads[k].double().norm().item(), 1e-20))
           for k, p in net.named_parameters()}
    REPORT["mixed_oracle256_" + dtype] = {"logit_max_abs": float(d.max()), "logit_mean_abs": float(d.mean()),
                                          "loss_abs_err": abs(float(loss.item()) - float(ref_loss.item())),
                                          "grad_rel_l2_median": float(np.median(list(rel.values()))),
                                          "grad_rel_l2_worst": max(rel.values())}
    _dump()
    assert d.max() < 1e-3 and d.mean() < 1.5e-4, REPORT["mixed_oracle256_" + dtype]
    assert abs(loss.item() - ref_loss.item()) < 2e-5
    assert max(rel.values()) < (0.11 if dtype == "f16" else 0.17)      # per-tensor rel-L2, measured 7.3 % / 11.5 %


@pytest.mark.parametrize("mode", ["default", "mixed", "full"])
def test_config2_bs32_256_vs_reference_fixture(golden_dir, mode):
    """BASELINE config 2 AT ITS OWN SIZE -- UNet(1,2), 256x256, batch 32, train mode -- against the compact fixture generated
    from the imported reference (tests/golden/make_golden.py --only unet_b32: logits sub-sampled 8x8 + the 64 largest-|logit|
    positions, the losses, all 118 gradient summaries, the BatchNorm buffers).  The inputs are regenerated from the seed
    (their checksums are in the fixture).  "default" here = the FAST 16-bit mode (UNet(precise=False)), asserted at 1.5x its measured
    error; "mixed" = what UNet() builds: AT the north star's 1e-3, over the sub-sample, the 64 largest logits AND the full logits of
    four images; full: 3e-5."""
    from golden_util import tensor_checksum
    from semantic_segmentation_amd.losses import seg_loss
    z = np.load(os.path.join(golden_dir, "unet_c2_256_b32.npz"))
    n_classes, seed, batch, size = int(z["n_classes"]), int(z["seed"]), int(z["batch"]), int(z["size"])
    x, mask = oracle.synthetic_batch(batch, size, seed=int(z["data_seed"]))
    assert np.allclose(tensor_checksum(x), z["xsum"], rtol=1e-6, atol=1e-6), "the synthetic input stream drifted"
    assert np.allclose(tensor_checksum(mask.float()), z["masksum"], rtol=0, atol=0)
    net, sd = build_mode(n_classes, seed, {"default": False, "mixed": "mixed", "full": True}[mode])
    for k, v in sd.items():
        if v.is_floating_point():
            assert np.allclose(tensor_checksum(v), z["wsum/" + k], rtol=1e-6, atol=1e-7), k
    net.train()
    logits = net(x.cuda())
    loss, parts = seg_loss(logits, mask.cuda(), return_parts=True)
    loss.backward()
    torch.cuda.synchronize()
    lg = logits.detach().cpu()
    d_sub = (lg[:, :, ::8, ::8] - torch.from_numpy(z["logits_sub"])).abs()
    d_top = (lg.flatten()[torch.from_numpy(z["top_idx"])] - torch.from_numpy(z["top_val"])).abs()
    worst, wk = 0.0, None
    for k, p in net.named_parameters():
        ref = z["gsum/" + k]
        e = abs(grad_summary(p.grad.cpu())[1] - ref[1]) / max(ref[1], 1e-12)
        if e > worst:
            worst, wk = e, k
    bworst = 0.0
    for k, b in net.named_buffers():
        if "num_batches" not in k:
            ref = z["buf/" + k]
            bworst = max(bworst, float(np.abs(b.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-6)))
    rep = {"logit_sub_max_abs": float(d_sub.max()), "logit_sub_mean_abs": float(d_sub.mean()), "logit_top64_max_abs": float(d_top.max()),
           "logit_absmean_rel_err": abs(float(lg.abs().mean()) - float(z["logits_absmean"])) / float(z["logits_absmean"]),
           "loss_abs_err": abs(float(loss.item()) - float(z["loss"])), "dice_abs_err": abs(float(parts[2].item()) - float(z["loss_dice"])),
           "grad_norm_rel_err_worst": worst, "grad_norm_worst_key": wk, "bn_buffer_rel_err_worst": bworst}
    REPORT["config2_bs32_" + mode] = rep
    _dump()
    # the FULL logits of the first four images: 0.5 M contiguous logits at the size the metric is quoted on
    d_full = (lg[:4] - torch.from_numpy(z["logits_full4"])).abs()
    rep["logit_full4_max_abs"], rep["logit_full4_mean_abs"] = float(d_full.max()), float(d_full.mean())
    _dump()
    # fast (UNet(precise=False)): 1.5x measured (4.30e-3 over the full logits of four images / 5.5e-4); mixed (what UNet() builds): AT the north star's 1e-3; full: 3e-5
    lim_max, lim_mean, lim_loss = {"default": (6.5e-3, 8.2e-4, 1e-3), "mixed": (1e-3, 1.5e-4, 2e-5), "full": (3e-5, 4e-6, 2e-5)}[mode]
    assert d_sub.max() < lim_max and d_top.max() < lim_max and d_sub.mean() < lim_mean, rep
    assert d_full.max() < lim_max and d_full.mean() < lim_mean, rep
    assert rep["loss_abs_err"] < lim_loss and rep["dice_abs_err"] < lim_loss, rep
    # gradient norms: measured 2.3 % / 1.0 % / 0.4 %; BatchNorm buffers 2.8e-4 / 2.8e-4 / 4e-6
    assert worst < {"default": 3.5e-2, "mixed": 1.5e-2, "full": 6e-3}[mode] and bworst < (1e-5 if mode == "full" else 4.2e-4), rep
    net.eval()
    with torch.no_grad():
        le = net(x.cuda()).cpu()
    de = (le[:, :, ::8, ::8] - torch.from_numpy(z["logits_eval_sub"])).abs()
    REPORT["config2_bs32_" + mode]["eval_logit_sub_max_abs"] = float(de.max())
    REPORT["config2_bs32_" + mode]["eval_dice_delta"] = abs(float(oracle.evaluate_dice(le, mask)) - float(z["eval_dice"]))
    _dump()
    assert REPORT["config2_bs32_" + mode]["eval_dice_delta"] < 1e-3
    assert de.max() < 5e-5 * max(1.0, float(np.abs(z["logits_eval_sub"]).max()))      # measured 3.3e-5 (fast) / 2e-6 (mixed)


def test_mixed_plan_choices():
    """the plan resolver: False / True / "mixed" / dict; bf16 "mixed" = every stage; unknown names raise"""
    from semantic_segmentation_amd.unet import unet_engine as ue
    assert ue.resolve_plan(False, "f16") is None and ue.resolve_plan(None, "f16") is None
    assert set(ue.resolve_plan(True, "f16").values()) == {"xw"}
    m = ue.resolve_plan("mixed", "f16")
    assert [k for k, v in m.items() if v == "xw"] == [s_ for s_ in ue.STAGES if s_ in ue.MIXED_XW] and len(m) == 22
    assert set(ue.resolve_plan("mixed", "bf16").values()) == {"xw"}
    assert ue.resolve_plan("auto", "f16") == m                                            # what UNet() builds
    assert set(ue.resolve_plan("auto", "f16", bilinear=True).values()) == {"xw"}          # bilinear: every stage corrected
    from semantic_segmentation_amd.unet import UNet
    assert UNet(1, 2).engine.plan == m and UNet(1, 2).engine.auto and UNet(1, 2, precise=False).engine.plan is None
    from semantic_segmentation_amd.unet3d import unet3d_engine as u3
    m3 = u3.resolve_plan3d("auto", "f16")
    assert [k for k, v in m3.items() if v in ("xw", "xw-")] == [s_ for s_ in u3.STAGES3D if s_ in u3.MIXED3D_XW]
    assert [k for k, v in m3.items() if v == "xw-"] == ["s_block1.conv1"]      # the level-0 decoder entry: w_lo on the residual half
    assert u3.segs3d("xw-", 192, lo0=128, lo_len=64) == ([(0, 0, 192), (0, 128, 64), (1, 128, 64)], 320, 256)
    assert u3.segs3d("xw", 32) == ([(0, 0, 32), (0, 0, 32), (1, 0, 32), (2, 0, 32)], 128, 64)      # the 32-channel conv: K padded to 128
    assert u3.segs3d("1", 32) == ([(0, 0, 32), (2, 0, 32)], 64, 64)
    assert u3.segs3d("xw", 192, lo0=128, lo_len=64) == ([(0, 0, 192), (0, 128, 64), (1, 0, 192)], 448, 256)
    assert u3.segs3d("1", 768, lo0=512, lo_len=256) == ([(0, 0, 768)], 768, 768)
    assert ue.resolve_plan({"inc.3": "w"}, "f16")["inc.3"] == "w"
    # "xw-": the w_lo segment only on the channels whose x_lo is valid (a round-4 experiment on the decoder-entry convs, opt-in)
    assert ue._segs("xw-", 128, 64) == ([(0, 0, 128), (0, 0, 64), (1, 0, 64)], 256, 192) and ue._segs("xw-", 64) == ue._segs("xw", 64)
    with pytest.raises(ValueError):
        ue.resolve_plan({"inc.7": "w"}, "f16")
    with pytest.raises(ValueError):
        ue.resolve_plan("fast", "f16")


def test_decoder_entry_partial_w_lo_plan_vs_golden(golden_dir):
    """The opt-in plan with the decoder-entry convs' w_lo segment on the skip half only (`GSSEG_MIXED_UPW=skip`, mode "xw-";
    DESIGN.md section 2.2: -0.21 ms per step, not the default because it spends margin): still inside the north star's 1e-3 on
    the hardest fixture (measured 7.8e-4 against 6.7e-4 for the default plan)."""
    from semantic_segmentation_amd.unet import unet_engine as ue
    z = np.load(os.path.join(golden_dir, "unet_c2_128_b4.npz"))
    plan = dict(ue.resolve_plan("mixed", "f16"))
    plan["up3.conv.0"] = plan["up4.conv.0"] = "xw-"
    net, _ = build_mode(int(z["n_classes"]), int(z["seed"]), plan)
    net.train()
    d = np.abs(net(torch.from_numpy(z["x"]).cuda()).detach().cpu().numpy() - z["logits"])
    REPORT["partial_w_lo_unet_c2_128_b4"] = {"logit_max_abs": float(d.max()), "logit_mean_abs": float(d.mean())}
    _dump()
    assert d.max() < 1e-3 and d.mean() < 1.6e-4


@pytest.mark.parametrize("mode", ["1", "x", "w"])
def test_pair_forward_segment_modes_run(mode):
    """every segment mode of the pair forward ("1" = x_hi.w_hi only, "x" = + x_lo.w_hi, "w" = + x_hi.w_lo) on every stage: the
    logits stay within the default mode's error of the oracle (they differ from it only in which rounding is removed)"""
    from semantic_segmentation_amd.unet import unet_engine as ue
    net, sd = build_mode(2, 19, {s_: mode for s_ in ue.STAGES})
    net.train()
    x, mask = oracle.synthetic_batch(2, 96, seed=21)
    ref = oracle.unet_forward(sd, x, True)
    d = (net(x.cuda()).detach().cpu() - ref).abs()
    REPORT["segmode_" + mode] = {"logit_max_abs": float(d.max()), "logit_mean_abs": float(d.mean())}
    _dump()
    assert d.max() < 6.5e-3 and d.mean() < 9e-4


# ------------------------------------------------------------------------------------------------ RGB input, fused eval
@pytest.mark.parametrize("precise", [None, False, True])
def test_unet_rgb_input_vs_oracle(precise):
    """UNet(n_channels=3, n_classes=1) -- the ISIC / RGB configuration (train_end2end_isic.py) -- forward + backward against
    the oracle at 96x80 (the direct first-layer kernels read 1..4 input channels).  precise=None = what UNet(3, 1) builds: the
    mixed pair forward, the north star's 1e-3 on logits."""
    from semantic_segmentation_amd.losses import seg_loss
    from semantic_segmentation_amd.unet import UNet
    sd = oracle.unet_state_dict(3, 1, seed=17)
    net = UNet(3, 1, precise=precise)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 96, 80, generator=g)
    mask = (torch.rand(2, 1, 96, 80, generator=g) > 0.6).long()
    ref_logits, ref_loss, ref_grads, _ = oracle.unet_step(sd, x, mask, train=True)
    logits = net(x.cuda())
    loss = seg_loss(logits, mask.cuda())
    loss.backward()
    torch.cuda.synchronize()
    d = (logits.detach().cpu() - ref_logits).abs()
    rel = {k: float((p.grad.cpu().double() - ref_grads[k].double()).norm() / max(ref_grads[k].double().norm().item(), 1e-20))
           for k, p in net.named_parameters()}
    REPORT["rgb_c1" + {None: "_default", False: "", True: "_precise"}[precise]] = {"logit_max_abs": float(d.max()), "logit_mean_abs": float(d.mean()),
                                                         "loss_abs_err": abs(loss.item() - ref_loss.item()),
                                                         "grad_rel_l2_worst": max(rel.values()),
                                                         "grad_rel_l2_first_conv": rel["inc.double_conv.0.weight"]}
    _dump()
    assert abs(loss.item() - ref_loss.item()) < (1e-3 if precise is False else 2e-5)
    if precise is None:
        assert net.engine.plan is not None and d.max() < 1e-3 and d.mean() < 1.6e-4      # the default mode (measured: see parity_unet.json rgb_c1_default)
    else:
        assert d.max() < (3e-5 if precise else 7.5e-3) and d.mean() < (4e-6 if precise else 1.05e-3)     # fast: 4.97e-3 / 6.8e-4 measured
    assert max(rel.values()) < 0.3


@pytest.mark.parametrize("C", [1, 2, 3])
def test_fused_eval_dice_matches_oracle(C):
    """gs_eval_dice (threshold / arg-max + per-sample Dice + mean in one pass) against the oracle's restatement of
    unet/evaluate.py:29-43, incl. a sample with an empty mask and empty prediction (Dice = 1 through the sets == 0 rule)."""
    from semantic_segmentation_amd.losses import eval_dice
    g = torch.Generator().manual_seed(30 + C)
    N, H, W = 5, 70, 52
    logits = torch.randn(N, C, H, W, generator=g) * 3
    mask = torch.randint(0, max(2, C), (N, 1, H, W), generator=g)
    mask[1] = 0
    if C == 1:
        logits[1] = -5.0                                  # nothing predicted, nothing true: dice == 1
    else:
        logits[1, 0] = 9.0
    mask[2] = 1
    want = oracle.evaluate_dice(logits, mask)
    got = eval_dice(logits.cuda(), mask.cuda())
    assert abs(got.item() - want.item()) < 2e-7, (got.item(), want.item())
    got3 = eval_dice(logits.cuda(), mask[:, 0].cuda())    # [N,H,W] masks too
    assert got3.item() == got.item()


def test_dice_coeff_per_item_is_one_launch_pair_and_exact():
    from semantic_segmentation_amd.util.dice_score import dice_coeff, multiclass_dice_coeff
    g = torch.Generator().manual_seed(2)
    p = torch.rand(6, 40, 56, generator=g)
    t = (torch.rand(6, 40, 56, generator=g) > 0.5).float()
    t[3] = 0
    p[3] = 0
    want = oracle.dice_coeff(p, t, reduce_batch_first=False)
    got = dice_coeff(p.cuda(), t.cuda(), reduce_batch_first=False)
    assert abs(got.item() - want.item()) < 2e-7
    pm, tm = p.reshape(2, 3, 40, 56), t.reshape(2, 3, 40, 56)
    assert abs(multiclass_dice_coeff(pm.cuda(), tm.cuda(), False).item()
               - oracle.multiclass_dice_coeff(pm, tm, False).item()) < 2e-7


def test_dynamic_loss_scale_survives_a_1e4_loss_weight():
    """fp16 backward with the loss multiplied by 1e4 (and by 1e-4): the static loss scale assumes a mean-reduced loss and
    overflows / underflows fp16; dynamic_loss_scale renormalises on the device and gives the scaled baseline gradient."""
    from semantic_segmentation_amd.losses import seg_loss
    from semantic_segmentation_amd.unet import UNet
    sd = oracle.unet_state_dict(1, 1, seed=9)
    x, mask = oracle.synthetic_batch(2, 64, seed=6)
    x, mask = x.cuda(), mask.cuda()

    def grads(factor, dynamic):
        net = UNet(1, 1, dynamic_loss_scale=dynamic)
        net.load_state_dict(sd, strict=True)
        net = net.cuda().train()
        (seg_loss(net(x), mask) * factor).backward()
        return {k: p.grad.clone() for k, p in net.named_parameters()}

    base = grads(1.0, False)
    for factor in (1e4, 1e-4):
        got = grads(factor, True)
        for k in base:
            ref = base[k] * factor
            assert torch.isfinite(got[k]).all(), k
            err = float((got[k] - ref).norm() / ref.norm().clamp_min(1e-30))
            assert err < 2e-2, (factor, k, err)      # same 16-bit roundings up to the power-of-two renormalisation
    big = grads(1e4, False)                      # documents the limit of the static scale: not finite (or saturated)
    bad = sum(int((~torch.isfinite(v)).sum()) for v in big.values())
    REPORT["static_scale_x1e4_nonfinite_grad_elements"] = bad
    _dump()


def test_fused_eval_jaccard_matches_oracle():
    """gs_eval_jaccard against the oracle's restatement of the ISIC evaluate (train_end2end_isic.py:58-84), incl. an empty
    sample (Jaccard = 1 through smooth = 1)."""
    from semantic_segmentation_amd.losses import eval_jaccard
    g = torch.Generator().manual_seed(77)
    N, H, W = 4, 70, 52
    logits = torch.randn(N, 1, H, W, generator=g) * 3
    mask = torch.randint(0, 2, (N, 1, H, W), generator=g)
    mask[1] = 0
    logits[1] = -5.0
    pred = (torch.sigmoid(logits) > 0.5).float()
    want = oracle.jaccard_index(pred.squeeze(), mask.float().squeeze())
    got = eval_jaccard(logits.cuda(), mask.cuda())
    assert abs(got.item() - want.item()) < 2e-7, (got.item(), want.item())


def test_weight_gradients_on_side_stream_are_identical(monkeypatch):
    """GSSEG_WGRAD_STREAM=1 runs the MFMA weight gradients on a second HIP stream (events + record_stream hand-over): the
    gradients must be bit-identical to the single-stream backward, for every parameter, over repeated steps"""
    import torch
    from oracle import oracle
    from semantic_segmentation_amd.losses import seg_loss
    from semantic_segmentation_amd.unet import UNet, unet_engine
    dev = torch.device("cuda:0")
    sd = oracle.unet_state_dict(1, 2, seed=11)
    x, mask = oracle.synthetic_batch(4, 64, seed=5)

    def grads(side):
        monkeypatch.setattr(unet_engine, "WGRAD_SIDE_STREAM", side)
        net = UNet(1, 2).to(dev)
        net.load_state_dict(sd, strict=True)
        net.train()
        out = []
        for _ in range(3):
            for p in net.parameters():
                p.grad = None
            loss = seg_loss(net(x.to(dev)), mask.to(dev))
            loss.backward()
            torch.cuda.synchronize()
            out.append({n: p.grad.clone() for n, p in net.named_parameters()})
        return out

    a, b = grads(False), grads(True)
    for ga, gb in zip(a, b):
        for n in ga:
            assert torch.equal(ga[n], gb[n]), n


@pytest.mark.parametrize("N,H,W,need_dx", [(2, 64, 64, False), (1, 16, 10000, False), (2, 48, 80, True)])
def test_stem_and_head_fusions_match_the_stored_tensor_paths(monkeypatch, N, H, W, need_dx):
    """The stem without its convolution output in memory and the head on the last stage's conv output (unet_parts.py:16-21,
    74) against the paths that store those tensors: same logits, loss and gradients up to the 16-bit rounding the stored
    tensors add.  A 10000-pixel-wide strip does not fit the stem backward's LDS strip (the engine re-forms y and takes the
    tensor path), and an input that needs a gradient takes the tensor path in the stem's backward as well."""
    from semantic_segmentation_amd.losses import seg_loss
    from semantic_segmentation_amd.unet import UNet, unet_engine
    dev = torch.device("cuda:0")
    sd = oracle.unet_state_dict(1, 2, seed=23)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, 1, H, W, generator=g)
    mask = (torch.rand(N, H, W, generator=g) > 0.6).long()

    def run(fused):
        for flag in ("FUSED_STEM_FWD", "FUSED_HEAD_FWD", "FUSED_HEAD_BWD", "FUSED_STEM_BWD"):
            monkeypatch.setattr(unet_engine, flag, fused)
        net = UNet(1, 2, precise=False).to(dev)          # the fusions of the 16-bit engine (the pair forward has its own forms)
        net.load_state_dict(sd, strict=True)
        net.train()
        xd = x.to(dev).requires_grad_(need_dx)
        logits = net(xd)
        loss = seg_loss(logits, mask.to(dev))
        loss.backward()
        torch.cuda.synchronize()
        gr = {n: p.grad.detach().float().cpu() for n, p in net.named_parameters()}
        bufs = {n: b.detach().float().cpu() for n, b in net.named_buffers() if b.dtype.is_floating_point}
        return logits.detach().cpu(), loss.item(), gr, bufs, (xd.grad.detach().cpu() if need_dx else None)

    la, lossa, ga, ba, dxa = run(True)
    lb, lossb, gb, bb, dxb = run(False)
    assert float((la - lb).abs().max()) < 1.5e-2 * max(1.0, float(lb.abs().max())) and float((la - lb).abs().mean()) < 1.5e-3
    assert abs(lossa - lossb) < 1e-3 * max(1.0, abs(lossb))
    for n in ba:                                              # running statistics: same batch statistics
        assert float((ba[n] - bb[n]).abs().max()) < 2e-3 * max(1.0, float(bb[n].abs().max())), n
    for n in ga:
        assert torch.isfinite(ga[n]).all(), n
        num = float((ga[n].double() * gb[n].double()).sum()); den = float((gb[n].double() ** 2).sum())
        if den > 1e-20:
            assert 0.7 < num / den < 1.3, (n, num / den)
    if need_dx:
        num = float((dxa.double() * dxb.double()).sum()); den = float((dxb.double() ** 2).sum())
        assert 0.8 < num / den < 1.2, num / den


@pytest.mark.parametrize("name", ["unet_c2_128_b4", "unet_c1_odd"])
def test_q_plan_fp8_correction_segment_vs_golden(golden_dir, name):
    """The opt-in "q" form of the mixed plan (GSSEG_MIXED_SEG=q): the correction terms of the nine MIXED_XW stages as ONE FP8
    block-scaled MFMA segment.  Round-4 experiment with the kill criterion "every fixture < 8.5e-4 and >= 1 ms per step": parity met
    (measured 7.7e-4 / 6.1e-4 here, simulated 6.3e-4: tools/parity_fp8_sim.py), the step gained 0.3 ms -- not the default."""
    from semantic_segmentation_amd.losses import seg_loss
    from semantic_segmentation_amd.unet import UNet, unet_engine as ue
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    n_classes, seed = int(z["n_classes"]), int(z["seed"])
    sd = oracle.unet_state_dict(1, n_classes, seed=seed)
    net = UNet(1, n_classes, precise={s_: ("q" if s_ in ue.MIXED_XW else "1") for s_ in ue.STAGES})
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    x = torch.from_numpy(z["x"]).cuda()
    mask = torch.from_numpy(z["mask"].astype(np.int64)).cuda()
    logits = net(x)
    loss = seg_loss(logits, mask)
    loss.backward()
    torch.cuda.synchronize()
    d = np.abs(logits.detach().cpu().numpy() - z["logits"])
    REPORT["q_" + name] = {"logit_max_abs": float(d.max()), "logit_mean_abs": float(d.mean())}
    _dump()
    assert d.max() < 8.5e-4 and d.mean() < 1.6e-4, REPORT["q_" + name]
    assert abs(float(loss.item()) - float(z["loss"])) < 2e-5
    for k, p in net.named_parameters():
        ref = z["gsum/" + k]
        assert abs(grad_summary(p.grad.cpu())[1] - ref[1]) / max(ref[1], 1e-12) < 1.5e-2, k


_CURVE_ORACLE = {}


@pytest.mark.parametrize("mode", ["default", "fast"])
def test_training_curve_tracks_the_fp32_oracle(mode):
    """What the 16-bit backward (and, in the fast mode, the 16-bit forward) does to a TRAINING CURVE: 24 RMSprop
    steps (the script's optimiser and learning rate: train_end2end_jsrt.py:69-70, lr 1e-5, momentum 0.9, weight decay 1e-8) of
    UNet(1,2) on three alternating 64x64 batches of 4, this engine against the fp32 CPU oracle stepping the same initial weights
    with the same optimiser, while the loss falls from 0.98 to 0.18.  The per-step losses must track (bounds = 1.5x the measured
    maximum relative deviation, see the assert)."""
    from semantic_segmentation_amd.losses import seg_loss
    steps, lr = 24, 1e-5
    net, sd = build_net(2, seed=23, precise=None if mode == "default" else False)
    net.train()
    batches = [oracle.synthetic_batch(4, 64, seed=100 + i) for i in range(3)]
    opt = torch.optim.RMSprop(net.parameters(), lr=lr, weight_decay=1e-8, momentum=0.9, foreach=True)
    got = []
    for k in range(steps):
        x, m = batches[k % 3]
        opt.zero_grad(set_to_none=True)
        loss = seg_loss(net(x.cuda()), m.cuda())
        loss.backward()
        opt.step()
        got.append(float(loss.item()))
    # the oracle: the same loop on the CPU in fp32 (once for both modes: ~20 s)
    if "ref" not in _CURVE_ORACLE:
        ref_sd = {k: v.clone() for k, v in sd.items()}
        leaves = {k: v.clone().requires_grad_(True) for k, v in ref_sd.items() if v.is_floating_point() and "running" not in k}
        ropt = torch.optim.RMSprop(list(leaves.values()), lr=lr, weight_decay=1e-8, momentum=0.9, foreach=True)
        ref = []
        for k in range(steps):
            x, m = batches[k % 3]
            cur = dict(ref_sd)
            cur.update({kk: v.detach() for kk, v in leaves.items()})
            _, loss, grads, updates = oracle.unet_step(cur, x, m, True)
            for kk, v in leaves.items():
                v.grad = grads[kk]
            ropt.step()
            ref_sd.update(updates)
            ref.append(float(loss))
        _CURVE_ORACLE["ref"], _CURVE_ORACLE["leaves"] = ref, leaves
    ref, leaves = _CURVE_ORACLE["ref"], _CURVE_ORACLE["leaves"]
    dev_ = max(abs(a - b) / abs(b) for a, b in zip(got, ref))
    wrel = max(float((p.detach().cpu() - leaves[k].detach()).norm() / (leaves[k].detach().norm() + 1e-12))
               for k, p in net.named_parameters() if p.dim() == 4)
    REPORT["training_curve_" + mode] = {"loss_first": ref[0], "loss_last": ref[-1], "loss_last_engine": got[-1],
                                        "max_rel_loss_deviation": dev_, "worst_conv_weight_rel_l2": wrel}
    _dump()
    assert ref[-1] < 0.3 * ref[0], "the oracle's loss did not move: the test would not see a broken backward"
    # measured: default 1.9e-2 (final loss 0.1827 against 0.1807), fast 3.0e-2 (0.1843); final conv weights within 16 % / 18 % rel-L2 (RMSprop
    # turns every gradient component, however small, into a step of ~lr: two fp32 runs with different summation orders drift apart too)
    assert dev_ < (2.9e-2 if mode == "default" else 4.5e-2), REPORT["training_curve_" + mode]
    assert abs(got[-1] - ref[-1]) < 0.03 * ref[-1]
