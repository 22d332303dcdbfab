#!/usr/bin/env python3
"""What leaving CUs to RCCL costs on ONE GPU: the bs=32 256^2 U-Net step with the persistent grids (conv3x3, wgrad3x3, up-conv)
capped at 256 / 248 / 240 / 224 blocks (gs_set_persistent_grid), interleaved rounds in one process.  parallel.GradReducer
caps them at 256 - GSSEG_RCCL_CUS (default 8) when world > 1."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_segmentation_amd import ops
from semantic_segmentation_amd.harness import synthetic_batch
from semantic_segmentation_amd.losses import seg_loss
from semantic_segmentation_amd.unet import UNet
dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = UNet(1, 2).to(dev).train()
x, mask = synthetic_batch(32, 256, seed=1234)
x, mask = x.to(dev), mask.to(dev)
def step():
    for p in net.parameters():
        p.grad = None
    seg_loss(net(x), mask).backward()
def timeit(n=10):
    step(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
caps = [256, 248, 240, 224]
res = {c: [] for c in caps}
for rnd in range(4):
    for c in caps:
        ops.set_persistent_grid(c)
        res[c].append(timeit())
ops.set_persistent_grid(0)
base = statistics.median(res[256])
for c in caps:
    m = statistics.median(res[c])
    print(f"grid cap {c:4d} blocks: {m:7.3f} ms/step  ({32e3 / m:7.1f} img/s, {100 * (m / base - 1):+5.1f} %)")
