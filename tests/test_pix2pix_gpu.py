"""Pix2Pix generator / PatchGAN discriminator on the HIP engine vs the golden vectors generated from the
reference (tests/golden/pix2pix_256.npz: seeded weights, dropout disabled) and vs the CPU oracle.  GPU only.
Measured deltas are written to gpurun_out/parity_pix2pix.json."""
import json
import math
import os

import numpy as np
import pytest
import torch

from golden_util import grad_summary, seeded_discriminator_state_dict, seeded_generator_state_dict
from oracle import oracle

pytestmark = pytest.mark.gpu
REPORT = {}


def _dump():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_pix2pix.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def build(golden_dir):
    from semantic_segmentation_amd.models_pix2pix import networks
    z = np.load(os.path.join(golden_dir, "pix2pix_256.npz"))
    norm = networks.get_norm_layer("batch")
    G = networks.UnetGenerator(1, 1, 8, 64, norm_layer=norm, use_dropout=True)
    D = networks.NLayerDiscriminator(2, 64, 3, norm)
    sdG, sdD = seeded_generator_state_dict(seed=21), seeded_discriminator_state_dict(seed=22)
    G.load_state_dict(sdG, strict=True)
    D.load_state_dict(sdD, strict=True)
    no_dropout(G)
    arch = torch.from_numpy(z["arch"]).cuda().requires_grad_(True)
    networks.upconv_arch = arch                      # the reference re-binds the module global too (pix2pix_model.py:64)
    mask = torch.from_numpy(z["mask"].astype(np.float32)).cuda()
    real = torch.from_numpy(z["real_image"]).cuda()
    return networks, G.cuda(), D.cuda(), z, mask, real, sdG, sdD


def no_dropout(G):
    """The fixtures were generated with the dropout probability forced to 0 (make_golden.py): do the same here.  (All-ones
    keep masks are NOT equivalent: kept values are scaled by 1/(1-p) = 2.)"""
    n = 0
    for m in G.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
            n += 1
    assert n == 3, n


def ones_masks(N):
    shapes = [(N, 4, 4, 512), (N, 8, 8, 512), (N, 16, 16, 512)]      # dropout blocks, innermost-adjacent first
    return [torch.ones(s, dtype=torch.uint8, device="cuda") for s in shapes]


def test_generator_discriminator_eval_forward(golden_dir):
    networks, G, D, z, mask, real, _, _ = build(golden_dir)
    G.eval(); D.eval()
    with torch.no_grad():
        fe = G(mask)
        pe = D(torch.cat((mask, fe), 1))
    dg = (fe.cpu() - torch.from_numpy(z["fake_eval"])).abs()
    dd = (pe.cpu() - torch.from_numpy(z["pred_eval"])).abs()
    REPORT["eval"] = {"fake_max_abs": float(dg.max()), "fake_mean_abs": float(dg.mean()),
                      "pred_max_abs": float(dd.max()), "pred_scale": float(np.abs(z["pred_eval"]).max())}
    _dump()
    assert fe.shape == (2, 1, 256, 256) and pe.shape == (2, 1, 30, 30)
    assert dg.mean() < 2e-3 and dg.max() < 3e-2, REPORT
    assert dd.max() < 2e-2 * max(1.0, float(np.abs(z["pred_eval"]).max())), REPORT


def test_gan_steps_train_mode(golden_dir):
    """Generator.training_step / Discriminator.training_step (train_end2end_jsrt.py:141-172) with dropout
    disabled, against the reference's losses, gradients and the architecture-tensor gradient."""
    from semantic_segmentation_amd.losses import l1_loss
    networks, G, D, z, mask, real, sdG, sdD = build(golden_dir)
    G.train(); D.train()
    crit = networks.GANLoss("vanilla").cuda()
    fake = G(mask)
    dfk = (fake.detach().cpu() - torch.from_numpy(z["fake_train"])).abs()
    pred_fake = D(torch.cat((mask, fake), 1))
    loss_G = crit(pred_fake, True) + l1_loss(fake, real) * 100.0
    loss_G.backward()
    torch.cuda.synchronize()
    rep = {"fake_train_max_abs": float(dfk.max()), "fake_train_mean_abs": float(dfk.mean()),
           "loss_G": float(loss_G.item()), "loss_G_ref": float(z["loss_G"])}
    worst, worst_k, errs = 0.0, "", []
    for k, p in G.named_parameters():
        ref = z["gsumG/" + k]
        got = grad_summary(p.grad.cpu())
        e = abs(got[1] - ref[1]) / max(ref[1], 1e-12)
        errs.append(e)
        if e > worst:
            worst, worst_k = e, k
    rep["G_grad_norm_rel_err_worst"], rep["G_grad_norm_worst_key"] = worst, worst_k
    rep["G_grad_norm_rel_err_median"] = float(np.median(errs))
    rep["G_grad_norm_rel_err_p90"] = float(np.quantile(errs, 0.9))
    ag = z["arch_grad_G"]
    got = networks.upconv_arch.grad.cpu().numpy()
    rep["arch_grad_rel_err"] = float(np.abs(got - ag).max() / np.abs(ag).max())
    # D step
    for p in D.parameters():
        p.grad = None
    D2 = D
    D2.load_state_dict(sdD, strict=True)             # fresh BN buffers like the fixture
    pf = D2(torch.cat((mask, fake), 1).detach())
    pr = D2(torch.cat((mask, real), 1))
    loss_D = (crit(pf, False) + crit(pr, True)) * 0.5
    loss_D.backward()
    torch.cuda.synchronize()
    rep["loss_D"], rep["loss_D_ref"] = float(loss_D.item()), float(z["loss_D"])
    dworst = 0.0
    for k, p in D2.named_parameters():
        ref = z["gsumD/" + k]
        got = grad_summary(p.grad.cpu())
        dworst = max(dworst, abs(got[1] - ref[1]) / max(ref[1], 1e-12))
    rep["D_grad_norm_rel_err_worst"] = dworst
    REPORT["train"] = rep
    _dump()
    # Measured on MI355X (fp16): loss_G 2.4e-6 relative, loss_D 4e-5, fake image mean |d| 1.5e-4 / max 1.4e-3, generator
    # gradient norms median 5e-4 / worst 1.2e-2, architecture gradient 1.1e-3, discriminator gradient norms 1.0e-3 --
    # asserted at ~3-4x those values.  (Round 1 reported 3e-2 / 118 % here and blamed BatchNorm over 2..8 values: the test
    # had fed all-ones dropout masks at p = 0.5, i.e. scaled the kept activations by 2, while the fixture has p = 0.)
    assert abs(rep["loss_G"] - rep["loss_G_ref"]) < 1e-4 * abs(rep["loss_G_ref"]), rep
    assert abs(rep["loss_D"] - rep["loss_D_ref"]) < 2e-4, rep
    assert rep["fake_train_mean_abs"] < 2.3e-4 and rep["fake_train_max_abs"] < 2e-3, rep       # 1.5x measured (1.5e-4 / 1.31e-3)
    assert rep["G_grad_norm_rel_err_median"] < 3e-3 and rep["G_grad_norm_rel_err_worst"] < 5e-2, rep
    assert dworst < 5e-3, rep
    assert rep["arch_grad_rel_err"] < 5e-3, rep


def test_gan_steps_train_mode_batch8_vs_golden(golden_dir):
    """The same two steps at batch 8 against the REFERENCE-generated fixture pix2pix_256_b8.npz (well-conditioned
    BatchNorm): fake image, both losses, every generator / discriminator gradient norm, sampled gradient elements, the
    architecture gradient and the BatchNorm running statistics."""
    from semantic_segmentation_amd.losses import l1_loss
    from semantic_segmentation_amd.models_pix2pix import networks
    z = np.load(os.path.join(golden_dir, "pix2pix_256_b8.npz"))
    norm = networks.get_norm_layer("batch")
    G = networks.UnetGenerator(1, 1, 8, 64, norm_layer=norm, use_dropout=True)
    D = networks.NLayerDiscriminator(2, 64, 3, norm)
    sdG, sdD = seeded_generator_state_dict(seed=21), seeded_discriminator_state_dict(seed=22)
    G.load_state_dict(sdG, strict=True); D.load_state_dict(sdD, strict=True)
    no_dropout(G)
    G, D = G.cuda().train(), D.cuda().train()
    arch = torch.from_numpy(z["arch"]).cuda().requires_grad_(True)
    networks.upconv_arch = arch
    mask = torch.from_numpy(z["mask"].astype(np.float32)).cuda()
    real = (torch.from_numpy(z["real_image_u8"]).float() / 255.0).cuda()
    crit = networks.GANLoss("vanilla").cuda()
    fake = G(mask)
    dfk = (fake.detach().cpu()[:, :, ::4, ::4] - torch.from_numpy(z["fake_train"])).abs()
    pred_fake = D(torch.cat((mask, fake), 1))
    loss_G = crit(pred_fake, True) + l1_loss(fake, real) * 100.0
    loss_G.backward()
    torch.cuda.synchronize()
    nerr, serr = {}, {}
    for k, p in G.named_parameters():
        ref, got = z["gsumG/" + k], grad_summary(p.grad.cpu())
        nerr[k] = abs(got[1] - ref[1]) / max(ref[1], 1e-12)
        rms = ref[1] / math.sqrt(p.numel())                      # sampled elements against the gradient's RMS
        serr[k] = float(np.abs(got[2:] - ref[2:]).max() / max(rms, 1e-20))
    ag = z["arch_grad_G"]
    arel = float(np.abs(arch.grad.cpu().numpy() - ag).max() / np.abs(ag).max())
    bworst = 0.0
    for k, b in G.named_buffers():
        if "num_batches" not in k:
            ref = z["bufG/" + k]
            bworst = max(bworst, float(np.abs(b.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-6)))
    D.load_state_dict(sdD, strict=True)
    for p in D.parameters():
        p.grad = None
    pf = D(torch.cat((mask, fake), 1).detach())
    pr = D(torch.cat((mask, real), 1))
    loss_D = (crit(pf, False) + crit(pr, True)) * 0.5
    loss_D.backward()
    torch.cuda.synchronize()
    dn = max(abs(grad_summary(p.grad.cpu())[1] - z["gsumD/" + k][1]) / max(z["gsumD/" + k][1], 1e-12)
             for k, p in D.named_parameters())
    dbw = 0.0
    for k, b in D.named_buffers():
        if "num_batches" not in k:
            ref = z["bufD/" + k]
            dbw = max(dbw, float(np.abs(b.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-6)))
    rep = {"fake_mean_abs": float(dfk.mean()), "fake_max_abs": float(dfk.max()),
           "loss_G": float(loss_G.item()), "loss_G_ref": float(z["loss_G"]),
           "loss_D": float(loss_D.item()), "loss_D_ref": float(z["loss_D"]),
           "G_grad_norm_rel_err_median": float(np.median(list(nerr.values()))),
           "G_grad_norm_rel_err_worst": max(nerr.values()), "G_grad_norm_worst_key": max(nerr, key=nerr.get),
           "G_grad_sample_err_over_rms_median": float(np.median(list(serr.values()))),
           "G_grad_sample_err_over_rms_worst": max(serr.values()),
           "arch_grad_rel_err": arel, "G_bn_buffer_rel_err_worst": bworst, "D_grad_norm_rel_err_worst": dn,
           "D_bn_buffer_rel_err_worst": dbw}
    REPORT["train_b8_golden"] = rep
    _dump()
    # measured: fake mean 1.3e-4 / max 1.2e-3, loss_G 4.4e-6 rel, loss_D 3e-5, G gradient norms median 3.9e-4 / worst
    # 8.2e-3, sampled elements 5 % of the tensor RMS (median), arch 3.8e-4, BN buffers 5.9e-4 (G) / 2.5e-4 (D), D norms 1.3e-3
    assert rep["fake_mean_abs"] < 1.9e-4 and rep["fake_max_abs"] < 1.8e-3, rep                 # 1.5x measured (1.26e-4 / 1.16e-3)
    assert abs(rep["loss_G"] - rep["loss_G_ref"]) < 1e-4 * abs(rep["loss_G_ref"]), rep
    assert abs(rep["loss_D"] - rep["loss_D_ref"]) < 2e-4, rep
    assert rep["G_grad_norm_rel_err_median"] < 2e-3 and rep["G_grad_norm_rel_err_worst"] < 3e-2, rep
    assert rep["G_grad_sample_err_over_rms_median"] < 0.2, rep
    assert arel < 2e-3 and bworst < 3e-3 and dbw < 1e-3 and dn < 5e-3, rep


def test_discriminator_gradients_vs_oracle():
    """D alone on a larger batch (well-conditioned BatchNorm): every gradient against the CPU oracle."""
    from semantic_segmentation_amd.models_pix2pix import networks
    sdD = seeded_discriminator_state_dict(seed=5)
    D = networks.NLayerDiscriminator(2, 64, 3, networks.get_norm_layer("batch"))
    D.load_state_dict(sdD, strict=True)
    D = D.cuda().train()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(4, 2, 128, 128, generator=g)
    p = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sdD.items()}
    xc = x.clone().requires_grad_(True)
    ref = oracle.nlayer_discriminator_forward(p, xc, train=True)
    oracle.gan_loss(ref, True).backward()
    xg = x.cuda().requires_grad_(True)
    pred = D(xg)
    networks.GANLoss("vanilla")(pred, True).backward()
    torch.cuda.synchronize()
    assert (pred.detach().cpu() - ref.detach()).abs().max() < 2e-2
    rel = {}
    for k, q in D.named_parameters():
        a, b = q.grad.cpu().double(), p[k].grad.double()
        rel[k] = float((a - b).norm() / b.norm())
    REPORT["D_oracle_grad_rel_l2"] = rel
    REPORT["D_oracle_dx_rel_l2"] = float((xg.grad.cpu() - xc.grad).norm() / xc.grad.norm())
    _dump()
    assert max(rel.values()) < 0.15, rel
    assert REPORT["D_oracle_dx_rel_l2"] < 0.15


def test_generator_descent_direction(golden_dir):
    """first-order check of the generator backward (weights AND architecture tensor) through our own forward"""
    from semantic_segmentation_amd.losses import l1_loss
    networks, G, D, z, mask, real, _, _ = build(golden_dir)
    G.train()
    masks = ones_masks(2)

    def loss_fn():
        return l1_loss(G(mask, dropout_masks=masks), real)

    l0 = loss_fn()
    l0.backward()
    params = list(G.parameters()) + [networks.upconv_arch]
    g2 = sum(float((p.grad.double() ** 2).sum()) for p in params)
    target = 0.02 * float(l0.item())
    eps = target / g2
    with torch.no_grad():
        for p in params:
            p.add_(p.grad, alpha=-eps)
        l1 = loss_fn()
    ratio = (l0.item() - l1.item()) / target
    REPORT["G_descent_ratio"] = ratio
    _dump()
    assert 0.5 < ratio < 1.3, ratio


def test_merged_upconv_equals_three_way_sum(golden_dir):
    from semantic_segmentation_amd import ops
    z = np.load(os.path.join(golden_dir, "ops_micro.npz"))
    sd = {k[len("cell/sd/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("cell/sd/")}
    arch = torch.from_numpy(z["cell/arch"])
    w4, w6, w8 = (sd[f"_ops._ops.{j}.op.weight"].cuda().contiguous() for j in range(3))
    sm = torch.softmax(arch[2], -1).cuda().contiguous()
    merged = torch.empty(6, 5, 8, 8, device="cuda")
    ops.upconv_merge_pack(w4, w6, w8, sm, None, None, merged)
    Wm, _ = oracle.merged_upconv_weight({"c." + k: v for k, v in sd.items()}, "c", arch[2])
    assert (merged.cpu() - Wm).abs().max() < 1e-6


def test_generator_train_forward_vs_oracle_batch8():
    """train-mode generator at batch 8 (better conditioned deep BatchNorm) against the CPU oracle"""
    from semantic_segmentation_amd.models_pix2pix import networks
    sdG = seeded_generator_state_dict(seed=31)
    G = networks.UnetGenerator(1, 1, 8, 64, norm_layer=networks.get_norm_layer("batch"), use_dropout=True)
    G.load_state_dict(sdG, strict=True)
    G = G.cuda().train()
    g = torch.Generator().manual_seed(9)
    arch = 0.5 * torch.randn(8, 3, generator=g)
    networks.upconv_arch = arch.cuda()
    _, mask = oracle.synthetic_batch(8, 256, seed=41)
    mask = mask.float()
    keep = [(torch.rand(s, generator=g) > 0.5) for s in [(8, 4, 4, 512), (8, 8, 8, 512), (8, 16, 16, 512)]]
    with torch.no_grad():
        ref = oracle.unet_generator_forward(sdG, arch, mask, train=True,
                                            dropout_masks=[k.permute(0, 3, 1, 2).float() for k in keep])
        got = G(mask.cuda(), dropout_masks=[k.to(torch.uint8).cuda() for k in keep])
    d = (got.cpu() - ref).abs()
    REPORT["G_train_b8_vs_oracle"] = {"max_abs": float(d.max()), "mean_abs": float(d.mean()),
                                     "ref_abs_mean": float(ref.abs().mean())}
    _dump()
    assert d.mean() < 1e-3 and d.max() < 1e-2, REPORT["G_train_b8_vs_oracle"]
    # gradients (weights + architecture tensor) of an L1 objective against the oracle
    from semantic_segmentation_amd.losses import l1_loss
    real = torch.rand(8, 1, 256, 256, generator=g)
    pG = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sdG.items()}
    a = arch.clone().requires_grad_(True)
    refo = oracle.unet_generator_forward(pG, a, mask, train=True,
                                         dropout_masks=[k.permute(0, 3, 1, 2).float() for k in keep])
    oracle.l1_loss(refo, real).backward()
    ag = arch.cuda().requires_grad_(True)
    networks.upconv_arch = ag
    l1_loss(G(mask.cuda(), dropout_masks=[k.to(torch.uint8).cuda() for k in keep]), real.cuda()).backward()
    torch.cuda.synchronize()
    rel = []
    for k, q in G.named_parameters():
        x1, x2 = q.grad.cpu().double(), pG[k].grad.double()
        rel.append(float((x1 - x2).norm() / max(x2.norm().item(), 1e-20)))
    arel = float((ag.grad.cpu() - a.grad).norm() / a.grad.norm())
    REPORT["G_train_b8_grad_rel_l2"] = {"median": float(np.median(rel)), "p90": float(np.quantile(rel, 0.9)),
                                       "worst": float(max(rel)), "arch": arel}
    _dump()
    assert np.median(rel) < 0.15 and arel < 0.1, REPORT["G_train_b8_grad_rel_l2"]


def _pix2pix_opt(tmp_path, ngf=64):
    import argparse
    return argparse.Namespace(model="pix2pix", input_nc=1, output_nc=1, ngf=ngf, ndf=64, netG="unet_256", netD="basic",
                              n_layers_D=3, norm="batch", no_dropout=False, init_type="normal", init_gain=0.02,
                              gpu_ids=[0], cuda_index=0, isTrain=True, gan_mode="vanilla", lr=2e-4, beta1=0.5,
                              arch_lr=3e-4, lambda_L1=100.0, lr_policy="linear", n_epochs=2, n_epochs_decay=2,
                              epoch_count=1, checkpoints_dir=str(tmp_path), name="p2p", continue_train=False,
                              verbose=False)


def test_create_model_surface_and_stage1_steps(tmp_path):
    """`create_model(opt)` object of the reference (models_pix2pix/pix2pix_model.py, base_model.py): one
    optimize_parameters + optimize_architect iteration runs on the HIP engines, updates G, D and the up-conv
    architecture tensor, and save_model / load_model round-trips bit-exactly (train_pix2pix_lung.py:65-103)."""
    import semantic_segmentation_amd.compat as compat
    compat.install()
    from models_pix2pix import create_model, networks as nw
    torch.manual_seed(3)
    model = create_model(_pix2pix_opt(tmp_path))
    model.setup(model.opt)
    for attr in ("netG", "netD", "optimizer_G", "optimizer_D", "optimizer_arch_upconv", "optimizer_arch_conv",
                 "criterionGAN", "set_input", "set_input_1", "optimize_parameters", "optimize_architect",
                 "save_model", "load_model", "eval", "test", "update_learning_rate", "get_current_losses"):
        assert hasattr(model, attr), attr
    g = torch.Generator().manual_seed(5)
    mask = (torch.rand(2, 1, 256, 256, generator=g) > 0.6).float()
    image = torch.rand(2, 1, 256, 256, generator=g) * 2 - 1
    g0 = [p.detach().clone() for p in list(model.netG.parameters())[:3]]
    d0 = [p.detach().clone() for p in list(model.netD.parameters())[:3]]
    a0 = nw.upconv_arch.detach().clone()
    model.set_input_1({"image": image, "mask": mask})
    model.optimize_parameters()
    losses = model.get_current_losses()
    assert set(losses) == {"G_GAN", "G_L1", "D_real", "D_fake"} and all(math.isfinite(v) for v in losses.values())
    assert any(not torch.equal(a, b.detach()) for a, b in zip(g0, list(model.netG.parameters())[:3]))
    assert any(not torch.equal(a, b.detach()) for a, b in zip(d0, list(model.netD.parameters())[:3]))
    model.optimize_architect(image, mask)
    assert not torch.equal(a0, nw.upconv_arch.detach())            # the arch optimiser really owns the leaf tensor
    model.update_learning_rate()
    # checkpoint round trip
    out = tmp_path / "ckpt"
    model.save_model(str(out))
    sdG = {k: v.detach().clone() for k, v in model.netG.state_dict().items()}
    arch_saved = nw.upconv_arch.detach().clone()
    with torch.no_grad():
        for p in model.netG.parameters():
            p.add_(1.0)
        nw.upconv_arch.add_(1.0)
    model.load_model(str(out / "pix2pix_discriminator.pkl"), str(out / "pix2pix_generator.pkl"))
    for k, v in model.netG.state_dict().items():
        assert torch.equal(v, sdG[k]), k
    assert torch.equal(nw.upconv_arch.detach(), arch_saved)
    model.eval()
    model.test()
    assert model.fake_image.shape == (2, 1, 256, 256) and torch.isfinite(model.fake_image).all()


def test_discriminator_backward_is_bit_reproducible():
    """NLayerDiscriminator (networks.py:620-665): every weight gradient is summed in a fixed order (split-K slabs +
    ordered reduction, block reductions for the 2-channel first and 1-channel last conv)."""
    from semantic_segmentation_amd.models_pix2pix import networks
    D = networks.NLayerDiscriminator(2, 64, 3, networks.get_norm_layer("batch"))
    D.load_state_dict(seeded_discriminator_state_dict(seed=6), strict=True)
    D = D.cuda().train()
    x = torch.rand(4, 2, 256, 256, generator=torch.Generator().manual_seed(1)).cuda()
    crit = networks.GANLoss("vanilla")
    runs = []
    for _ in range(2):
        D.zero_grad(set_to_none=True)
        crit(D(x), True).backward()
        runs.append({k: p.grad.clone() for k, p in D.named_parameters()})
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k


def test_generator_backward_is_bit_reproducible_batch8():
    """UnetGenerator with the 3-way mixed transposed-conv cells at a batch where the merged-kernel weight gradients split
    K: slabs + ordered sums everywhere, so two passes give identical gradients (weights and architecture parameters)."""
    from semantic_segmentation_amd.models_pix2pix import networks
    torch.manual_seed(11)
    G = networks.define_G(1, 1, 64, "unet_256", "batch", False).cuda().train()     # no dropout: same mask-free graph twice
    networks.upconv_arch = (1e-3 * torch.randn(8, 3)).cuda().requires_grad_(True)
    x = (torch.rand(8, 1, 256, 256, generator=torch.Generator().manual_seed(2)) > 0.5).float().cuda()
    runs = []
    for _ in range(2):
        G.zero_grad(set_to_none=True)
        networks.upconv_arch.grad = None
        G(x).square().mean().backward()
        r = {k: p.grad.clone() for k, p in G.named_parameters()}
        r["arch"] = networks.upconv_arch.grad.clone()
        runs.append(r)
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k


def test_direct_image_layer_matches_mfma_path():
    """gs_upconv8_image_fwd (outermost generator layer as a direct kernel: merged 8x8 transposed conv + bias + tanh ->
    fp32 NCHW) against the same layer on the MFMA engine (4 class launches + tanh + layout pass): image and gradients."""
    from semantic_segmentation_amd.models_pix2pix import networks, pix2pix_engine
    torch.manual_seed(21)
    G = networks.define_G(1, 1, 64, "unet_256", "batch", False).cuda().train()
    networks.upconv_arch = (0.3 * torch.randn(8, 3)).cuda().requires_grad_(True)
    x = (torch.rand(2, 1, 256, 256, generator=torch.Generator().manual_seed(4)) > 0.5).float().cuda()
    res = {}
    for direct in (True, False):
        pix2pix_engine.DIRECT_IMAGE_LAYER = direct
        try:
            G.zero_grad(set_to_none=True)
            networks.upconv_arch.grad = None
            out = G(x)
            (out * torch.linspace(-1, 1, out.numel(), device=out.device).view_as(out)).mean().backward()   # mean-type loss: the
            # backward carries gradients times S ~ N*H*W in 16 bits
            res[direct] = (out.detach().clone(), {k: p.grad.clone() for k, p in G.named_parameters()})
        finally:
            pix2pix_engine.DIRECT_IMAGE_LAYER = True
    a, b = res[True], res[False]
    assert float((a[0] - b[0]).abs().max()) < 4e-3            # 16-bit rounding of the pre-activation on the MFMA path
    num = sum(float(((a[1][k] - b[1][k]).double() ** 2).sum()) for k in a[1])
    den = sum(float((b[1][k].double() ** 2).sum()) for k in a[1])
    assert (num / den) ** 0.5 < 2e-2


def test_rgb_generator_and_discriminator_vs_oracle():
    """The ISIC / RGB configuration of the GAN (scripts/train_end2end_isic.sh: --input_nc 1 --output_nc 3): UnetGenerator(1, 3)
    and NLayerDiscriminator(1 + 3) forward, GAN-step losses and gradients against the oracle at batch 4, 256x256."""
    from semantic_segmentation_amd import steps
    from semantic_segmentation_amd.models_pix2pix import networks
    N = 4
    sdG = seeded_generator_state_dict(seed=61, input_nc=1, output_nc=3)
    sdD = seeded_discriminator_state_dict(seed=62, input_nc=4)
    norm = networks.get_norm_layer("batch")
    G = networks.UnetGenerator(1, 3, 8, 64, norm_layer=norm, use_dropout=False)
    D = networks.NLayerDiscriminator(4, 64, 3, norm)
    G.load_state_dict(sdG, strict=True); D.load_state_dict(sdD, strict=True)
    G, D = G.cuda().train(), D.cuda().train()
    g = torch.Generator().manual_seed(6)
    arch = 0.4 * torch.randn(8, 3, generator=g)
    _, mask = oracle.synthetic_batch(N, 256, seed=13)
    maskf = mask.float()
    real = torch.rand(N, 3, 256, 256, generator=g)
    crit = networks.GANLoss("vanilla").cuda()
    ag = arch.cuda().requires_grad_(True)
    networks.upconv_arch = ag
    with torch.no_grad():
        ref_fake = oracle.unet_generator_forward(sdG, arch, maskf, train=True)
        fake = G(maskf.cuda())
    d = (fake.cpu() - ref_fake).abs()
    assert fake.shape == (N, 3, 256, 256)
    assert d.mean() < 5e-4 and d.max() < 1e-2, (float(d.mean()), float(d.max()))
    G.load_state_dict(sdG, strict=True)
    lG = steps.generator_step_loss(G, D, crit, maskf.cuda(), real.cuda(), 100.0)
    lG.backward()
    pG = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sdG.items()}
    a = arch.clone().requires_grad_(True)
    rG = oracle.generator_step_loss(pG, sdD, a, maskf, real, 100.0)
    rG.backward()
    assert abs(lG.item() - rG.item()) < 2e-4 * abs(rG.item()), (lG.item(), rG.item())
    nerr = [abs(float(q.grad.norm()) - float(pG[k].grad.norm())) / max(float(pG[k].grad.norm()), 1e-20)
            for k, q in G.named_parameters()]
    assert np.median(nerr) < 5e-3 and max(nerr) < 8e-2, (float(np.median(nerr)), max(nerr))
    assert float((ag.grad.cpu() - a.grad).abs().max() / a.grad.abs().max()) < 1e-2
    G.load_state_dict(sdG, strict=True); D.load_state_dict(sdD, strict=True)
    lD = steps.discriminator_step_loss(G, D, crit, maskf.cuda(), real.cuda())
    with torch.no_grad():
        rD = oracle.discriminator_step_loss(sdG, sdD, arch, maskf, real)
    assert abs(lD.item() - rD.item()) < 5e-4, (lD.item(), rD.item())
