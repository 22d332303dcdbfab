"""cProfile of the eager config-3 steps at batch 2 (host side: where the ~16 us per launch go)"""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_segmentation_amd import steps
from semantic_segmentation_amd.harness import synthetic_batch
from semantic_segmentation_amd.models_pix2pix import networks
from semantic_segmentation_amd.unet import UNet
dev = torch.device("cuda:0")
B = 2
G = networks.define_G(1, 1, 64, "unet_256", "batch", True).to(dev).train()
D = networks.define_D(2, 64, "basic", 3, "batch").to(dev).train()
U = UNet(1, 1).to(dev).train()
networks.upconv_arch = (1e-3 * torch.randn(8, 3)).to(dev).requires_grad_(True)
crit = networks.GANLoss("vanilla").to(dev)
for m in (G, D, U):
    m.engine.trust_versions = True
x, mask = synthetic_batch(B, 256, seed=3)
x, mask = x.to(dev), mask.to(dev)
maskf, real = mask.float(), torch.rand(B, 1, 256, 256, device=dev)
def it():
    for n in (G, D, U):
        for p in n.parameters(): p.grad = None
    steps.generator_step_loss(G, D, crit, maskf, real).backward()
    steps.discriminator_step_loss(G, D, crit, maskf, real).backward()
    steps.unet_step_loss(U, G, x, mask, 1.0).backward()
for _ in range(3): it()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): it()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])
s2 = io.StringIO()
pstats.Stats(pr, stream=s2).sort_stats("tottime").print_callers("_named_members")
print(s2.getvalue()[:3000])
