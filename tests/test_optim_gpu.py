"""Fused multi-tensor optimisers (csrc/optim.hip through the C ABI) against torch.optim's own single-tensor
implementations on the CPU, same parameters / gradients / hyper-parameters, several steps."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g) for s in shapes]


SHAPES = [(64, 1, 3, 3), (64,), (128, 64, 3, 3), (1024, 512, 3, 3), (7,), (3, 5, 2), (2, 64, 1, 1), (65537,)]


def _run(make_ref, make_fused, steps=4, skip_every=None):
    dev = torch.device("cuda:0")
    p_ref = [torch.nn.Parameter(t.clone()) for t in _params(1, SHAPES)]
    p_gpu = [torch.nn.Parameter(t.clone().to(dev)) for t in _params(1, SHAPES)]
    o_ref, o_gpu = make_ref(p_ref), make_fused(p_gpu)
    for k in range(steps):
        grads = _params(100 + k, SHAPES)
        for i, (a, b, g) in enumerate(zip(p_ref, p_gpu, grads)):
            if skip_every is not None and (i + k) % skip_every == 0:      # a parameter without gradient this step
                a.grad, b.grad = None, None
            else:
                a.grad, b.grad = g.clone(), g.clone().to(dev)
        o_ref.step()
        o_gpu.step()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(p_ref, p_gpu)):
        err = (a.detach() - b.detach().cpu()).abs().max().item()
        scale = a.detach().abs().max().item()
        assert err <= 2e-6 * max(scale, 1.0), (i, err)
    return o_ref, o_gpu


@pytest.mark.parametrize("momentum,wd", [(0.9, 1e-8), (0.0, 0.0), (0.5, 1e-2)])
def test_rmsprop_matches_torch(momentum, wd):
    """optim.RMSprop(lr, weight_decay=1e-8, momentum=0.9) of running_files/train_end2end_jsrt.py:69-70"""
    from semantic_segmentation_amd.optim import RMSprop
    o_ref, o_gpu = _run(lambda p: torch.optim.RMSprop(p, lr=1e-3, weight_decay=wd, momentum=momentum, foreach=False),
                        lambda p: RMSprop(p, lr=1e-3, weight_decay=wd, momentum=momentum, foreach=True), skip_every=5)
    sd_ref, sd_gpu = o_ref.state_dict(), o_gpu.state_dict()
    assert set(sd_ref["state"].keys()) == set(sd_gpu["state"].keys())
    k0 = next(iter(sd_ref["state"]))
    assert set(sd_ref["state"][k0].keys()) == set(sd_gpu["state"][k0].keys())
    for k in sd_ref["state"]:
        a, b = sd_ref["state"][k]["square_avg"], sd_gpu["state"][k]["square_avg"].cpu()
        assert (a - b).abs().max().item() <= 2e-6 * max(a.abs().max().item(), 1.0)


@pytest.mark.parametrize("betas,wd", [((0.5, 0.999), 0.0), ((0.5, 0.999), 1e-3), ((0.9, 0.999), 1e-5)])
def test_adam_matches_torch(betas, wd):
    """torch.optim.Adam of models_pix2pix/pix2pix_model.py:69-72 and train_end2end_jsrt.py:318"""
    from semantic_segmentation_amd.optim import Adam
    _run(lambda p: torch.optim.Adam(p, lr=2e-4, betas=betas, weight_decay=wd, foreach=False),
         lambda p: Adam(p, lr=2e-4, betas=betas, weight_decay=wd), skip_every=4)


def test_state_dict_round_trip_and_lr_schedule():
    from semantic_segmentation_amd.optim import RMSprop
    dev = torch.device("cuda:0")
    p = [torch.nn.Parameter(t.to(dev)) for t in _params(3, SHAPES[:4])]
    opt = RMSprop(p, lr=1e-3, weight_decay=1e-8, momentum=0.9)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "max", patience=0)      # train_end2end_jsrt.py:71
    for k in range(3):
        for q, g in zip(p, _params(50 + k, SHAPES[:4])):
            q.grad = g.to(dev)
        opt.step()
        sched.step(0.5)              # no improvement -> the LR is cut; the fused step must read the new value
    assert opt.param_groups[0]["lr"] < 1e-3
    sd = copy.deepcopy(opt.state_dict())
    opt2 = RMSprop([torch.nn.Parameter(q.detach().clone()) for q in p], lr=1.0, momentum=0.9)
    opt2.load_state_dict(sd)
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
    torch.optim.RMSprop(p, lr=1e-3, momentum=0.9).load_state_dict(sd)               # torch accepts the same layout


def test_cpu_parameters_raise():
    from semantic_segmentation_amd.optim import Adam
    q = torch.nn.Parameter(torch.zeros(4))
    q.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        Adam([q]).step()


@pytest.mark.parametrize("which", ["rmsprop", "adam"])
def test_fused_step_refreshes_engine_weight_packs(which):
    """The optimiser kernels write the parameters through raw pointers; the engines cache 16-bit packs of the weights
    keyed on the tensors' version counters.  After a fused step the network must compute with the NEW weights: its
    output has to equal that of a freshly built network loaded with the updated state dict (train mode and the
    folded-BatchNorm inference path)."""
    from semantic_segmentation_amd.losses import seg_loss
    from semantic_segmentation_amd.optim import Adam, RMSprop
    from semantic_segmentation_amd.unet import UNet
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    net = UNet(1, 2).to(dev).train()
    x = torch.randn(2, 1, 32, 32, device=dev)
    mask = (torch.rand(2, 1, 32, 32, device=dev) > 0.5).long()
    opt = RMSprop(net.parameters(), lr=1e-4, momentum=0.9) if which == "rmsprop" else Adam(net.parameters(), lr=1e-3)
    v0 = [p._version for p in net.parameters()]
    with torch.no_grad():
        net.eval()
        before = net(x).clone()
        net.train()
    seg_loss(net(x), mask).backward()
    opt.step()
    assert all(p._version > v for p, v in zip(net.parameters(), v0))
    fresh = UNet(1, 2).to(dev)
    fresh.load_state_dict(net.state_dict(), strict=True)
    for mode in ("eval", "train"):
        getattr(net, mode)()
        getattr(fresh, mode)()
        with torch.no_grad():
            a, b = net(x), fresh(x)
        assert torch.equal(a, b), mode
        if mode == "eval":
            assert torch.isfinite(a).all()
            assert float((a - before).abs().max()) > 1e-4      # the step moved every weight by ~1e-3: stale packs would give `before`
