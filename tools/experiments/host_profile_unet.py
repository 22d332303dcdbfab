"""cProfile of the U-Net engine's forward + backward called directly (autograd runs backward on its own thread, invisible to
cProfile): where the host time of a small-batch step goes"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_segmentation_amd.harness import synthetic_batch
from semantic_segmentation_amd.unet import UNet
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
U = UNet(1, 1).to(dev).train()
U.engine.trust_versions = True
x, mask = synthetic_batch(B, 256, seed=3)
x = x.to(dev)
eng = U.engine
params = dict(eng.param_items())
def it():
    logits, ectx = eng.forward(x, params, True, True)
    dl = torch.ones_like(logits) * 1e-4
    eng.backward(ectx, params, dl, False)
for _ in range(3): it()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): it()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:7000])
