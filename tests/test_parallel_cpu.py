"""Data-parallel logic on CPU: gloo, world size 2 (the N>1 path of bench.py / parallel.py without a GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from semantic_segmentation_amd.parallel import GradReducer, broadcast_module_state, shard_batch
        torch.manual_seed(100 + rank)                      # replicas start DIFFERENT ...
        net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
        broadcast_module_state(net)                        # ... and end identical to rank 0
        flat = torch.cat([p.detach().flatten() for p in net.parameters()] + [b.flatten().float() for b in net.buffers()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered)

        # the engine protocol: grads are allocated in the reducer's buckets, announced in backward order,
        # all-reduced bucket by bucket, and averaged
        red = GradReducer(net.named_parameters(), bucket_bytes=64)       # tiny buckets -> several collectives
        assert len(red.buckets) >= 2
        names = [n for n, _ in net.named_parameters()][::-1]
        params = dict(net.named_parameters())
        red.begin()
        local = {}
        for n in names:
            g = red.alloc(n, params[n])
            g.copy_(torch.full_like(params[n], float(rank + 1)) * (1 + names.index(n)))
            local[n] = g.clone()
            red.ready(n, g)
        red.finish()
        for n in names:
            want = (1 + names.index(n)) * (sum(range(1, world + 1)) / world)
            assert torch.allclose(red.views[n], torch.full_like(params[n], want)), n
        # a gradient produced outside the bucket is copied in
        red.begin()
        for n in names:
            red.ready(n, torch.ones_like(params[n]) * (rank + 1))
        red.finish()
        assert torch.allclose(red.flat, torch.full_like(red.flat, (world + 1) / 2))
        # an un-announced gradient leaves its bucket un-reduced: finish() must raise instead of handing out local gradients
        red.begin()
        for n in names[:-1]:
            red.ready(n, torch.ones_like(params[n]))
        try:
            red.finish()
            raise AssertionError("finish() accepted an incomplete bucket")
        except RuntimeError as e:
            assert "incomplete" in str(e)
        # parameters frozen after the reducer was built (the discriminator inside the Generator problem): announced, counted,
        # but a bucket made of frozen parameters only is not exchanged
        for p_ in net[2].parameters():
            p_.requires_grad = False
        red.begin()
        for n in names:
            red.ready(n, torch.ones_like(params[n]) * (rank + 1))
        n_live = len({red.bucket_of[n] for n in names if params[n].requires_grad})
        assert red.issued == n_live < len(red.buckets), (red.issued, n_live, len(red.buckets))
        red.finish()
        for p_ in net[2].parameters():
            p_.requires_grad = True
        lo, hi = shard_batch(8, rank, world)
        assert (lo, hi) == (rank * 4, rank * 4 + 4)
        # generic post-backward exchange (UNet3D / Pix2Pix / harness): rank-dependent gradients are averaged in
        # place, a parameter without gradient is skipped, small buckets force several collectives
        from semantic_segmentation_amd.parallel import all_reduce_gradients
        plist = list(net.parameters())
        for i, p in enumerate(plist):
            p.grad = None if i == 1 else torch.full_like(p, float((rank + 1) * (i + 1)))
        n = all_reduce_gradients(plist, bucket_bytes=48)
        assert n == sum(p.numel() for i, p in enumerate(plist) if i != 1)
        for i, p in enumerate(plist):
            if i == 1:
                assert p.grad is None
            else:
                assert torch.allclose(p.grad, torch.full_like(p, (i + 1) * (world + 1) / 2)), i
        # ---- the hooked-engine protocol of UNet3D / Pix2Pix G / D (parallel.GradEmitter + GradReducer.fetch): a parameter that
        # receives TWO contributions (the shared BatchNorm3d of the 3-D decoder blocks) is announced after the second one;
        # what autograd would receive (fetch) is the rank average; bf16 buckets exchange half the bytes
        from semantic_segmentation_amd.parallel import GradEmitter
        for bdt, tol in ((torch.float32, 0.0), (torch.bfloat16, 8e-3)):
            red2 = GradReducer(net.named_parameters(), bucket_bytes=64, dtype=bdt)
            announced = []
            def hook(n, g, red2=red2):
                announced.append(n)
                red2.ready(n, g)
            shared = names[1]                                # pretend this parameter is used by two stages
            em = GradEmitter(hook, expected={shared: 2})
            red2.begin()
            for n in names:
                em.emit(n, torch.full_like(params[n], float(rank + 1)))
                if n == shared:
                    assert shared not in announced           # one contribution still missing
            em.emit(shared, torch.full_like(params[shared], 10.0 * (rank + 1)))
            assert announced.count(shared) == 1 and sorted(announced) == sorted(names)
            red2.finish()
            for n in names:
                want = (11.0 if n == shared else 1.0) * (world + 1) / 2
                got = red2.fetch(n)
                assert got.dtype == torch.float32 and got.data_ptr() != red2.views[n].data_ptr()
                assert torch.allclose(got, torch.full_like(params[n], want), rtol=tol, atol=tol), (bdt, n)
            assert red2.flat.dtype == bdt
        q.put((rank, "ok"))
    except Exception as e:      # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_shard_batch_rejects_uneven():
    from semantic_segmentation_amd.parallel import shard_batch
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


def test_bench_self_launcher_two_gloo_ranks():
    """`python bench.py --gpus 2` with no launcher in front (how the driver may start it): the parent spawns the two ranks,
    relays rank 0's JSON line and propagates failure.  --backend gloo --dry skips the kernels (no GPU here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry",
                        "--steps", "3", "--warmup", "1", "--batch", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["dry"] is True and line["n_gpus"] == 2 and line["ranks"] == 2 and line["steps"] == 3
    # a failing rank (no GPU here, real mode) must give a non-zero exit, not a hang
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0
