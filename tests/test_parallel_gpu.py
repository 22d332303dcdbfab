"""The data-parallel engine hooks on a real GPU: a 1-rank RCCL group (backend "nccl") with the GradReducer attached
to the U-Net engine -- gradients are produced inside the reducer's flat buckets, every bucket goes through an
asynchronous RCCL all-reduce issued from inside backward, and the result must equal the plain single-GPU gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_grad_reducer_on_unet_engine_rccl_single_rank():
    from oracle import oracle
    from semantic_segmentation_amd.losses import seg_loss
    from semantic_segmentation_amd.parallel import GradReducer, all_reduce_gradients, broadcast_module_state
    from semantic_segmentation_amd.unet import UNet
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sd = oracle.unet_state_dict(1, 2, seed=5)
        x, mask = oracle.synthetic_batch(4, 64, seed=9)
        x, mask = x.to(dev), mask.to(dev)

        def run(with_reducer):
            net = UNet(1, 2)
            net.load_state_dict(sd, strict=True)
            net = net.to(dev).train()
            broadcast_module_state(net)
            red = None
            if with_reducer:
                red = GradReducer(net.named_parameters(), bucket_bytes=8 << 20, force_collective=True).attach(net.engine)
                assert len(red.buckets) >= 3
            out = []
            for _ in range(2):                                  # the buckets are reused step after step
                for p in net.parameters():
                    p.grad = None
                loss = seg_loss(net(x), mask, global_dice=True)
                loss.backward()
                torch.cuda.synchronize()
                out.append({k: p.grad.detach().clone() for k, p in net.named_parameters()})
                if red is not None:                             # what autograd hands out is the reduced bucket content
                    for k, p in net.named_parameters():
                        assert torch.equal(p.grad, red.views[k]), k
            return float(loss.detach()), out

        l0, g0 = run(False)
        l1, g1 = run(True)
        assert abs(l0 - l1) < 1e-6
        for step in range(2):
            for k in g0[step]:
                a, b = g0[step][k], g1[step][k]
                assert float((a - b).abs().max()) <= 1e-6 * (float(a.abs().max()) + 1e-12) + 1e-12, (step, k)
        # the generic post-backward exchange on HIP tensors (1-rank group: early return)
        net = UNet(1, 2).to(dev)
        assert all_reduce_gradients(net.parameters()) == 0
        # the end-to-end harness in data-parallel mode inside the (1-rank) RCCL group: replica broadcast at start, one
        # gradient exchange per problem step (the multi-rank arithmetic of that exchange is covered on CPU with gloo)
        from semantic_segmentation_amd.harness import EndToEndTrainer, SyntheticLungDataset
        from semantic_segmentation_amd.models_pix2pix import networks
        netU = UNet(1, 1).to(dev)
        netG = networks.define_G(1, 1, 64, "unet_256", "batch", True, "normal", 0.02, [0])
        netD = networks.define_D(2, 64, "basic", 3, "batch", "normal", 0.02, [0])
        mk = lambda n, seed: torch.utils.data.DataLoader(SyntheticLungDataset(n, 256, seed), batch_size=2, drop_last=True)
        tr = EndToEndTrainer(netU, netG, netD, networks.GANLoss("vanilla").to(dev), mk(4, 1), mk(2, 2), dev, unroll_steps=1,
                             valid_every=0, data_parallel=True)
        tr.run(2, log_every=0)
        assert tr.global_step == 2 and all(torch.isfinite(v).all() for rec in tr.history for v in rec.values())
    finally:
        dist.destroy_process_group()
