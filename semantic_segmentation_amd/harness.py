"""Betty-free end-to-end training harness: the four problems of
running_files/train_end2end_jsrt.py (Generator :141-154, Discriminator :157-172, Unet :176-226, Arch :229-236),
their optimisers (:69-71, pix2pix_model.py:69-72, :318), `SSEngine.validation` (:239-255: Dice on the validation
loader, best-checkpoint save, ReduceLROnPlateau) and the final save (:345) -- without betty / wandb / imgaug.

What is reproduced is the PER-STEP MATH with first-order updates, in the order Betty's engine visits the problems
(lower problems every iteration, the upper `arch` problem every `unroll_steps` iterations).  Betty's scheduling
and its darts finite-difference hypergradient are NOT reproduced (betty-ml is not vendored by the reference and no
reference test pins it -- SURVEY section 8c): the architecture step here is the plain gradient of its own loss;
as written in the reference that loss (validation loss of the U-Net) does not depend on the architecture tensors
in first order, so the step is a no-op unless `arch_through_generator=True` routes the validation masks through
G first (the dependency the reference's `l2u` table intends, :329).

Everything heavy (U-Net, generator, discriminator, fused losses, fused RMSprop/Adam) runs on the HIP kernels.

    python -m semantic_segmentation_amd.harness --synthetic --iters 20 --batch-size 2
"""
from __future__ import annotations

import argparse
import os
import time
from typing import Callable, Dict, Iterable, Iterator, Optional

import torch

from . import optim as gs_optim
from .augment import MaskAugmenter
from . import steps
from .models_pix2pix import networks
from .unet import UNet
from .unet.evaluate import evaluate


def _cycle(loader: Iterable) -> Iterator:
    while True:
        for batch in loader:
            yield batch


class EndToEndTrainer:
    def __init__(self, net, netG, netD, criterionGAN, train_loader, val_loader, device, *, unet_lr=1e-5, gan_lr=2e-4,
                 beta1=0.5, arch_lr=3e-4, lambda_L1=100.0, loss_lambda=1.0, unroll_steps=1, valid_every=10,
                 mask_augment: Optional[Callable] = None, save_dir: Optional[str] = None,
                 arch_through_generator: bool = False, train_gan: bool = True, data_parallel: bool = False,
                 hip_graphs: bool = False, graph_warmup: int = 2):
        self.net, self.netG, self.netD, self.criterionGAN = net, netG, netD, criterionGAN
        self.device, self.train_loader, self.val_loader = device, train_loader, val_loader
        self.lambda_L1, self.loss_lambda, self.unroll_steps, self.valid_every = lambda_L1, loss_lambda, unroll_steps, valid_every
        self.mask_augment, self.save_dir, self.train_gan = mask_augment, save_dir, train_gan
        self.arch_through_generator = arch_through_generator
        # the architecture tensors live on the device (the reference creates them with .cuda(); a host-resident one costs a
        # host-to-device copy per generator forward and cannot be read inside a captured graph)
        for a in networks.arch_parameters():
            if a.device != torch.device(device):
                a.data = a.data.to(device)
                a.grad = None                        # (a gradient accumulated on the host belongs to the host copy)
        # optimisers of the reference, fused
        self.optimizer_unet = gs_optim.RMSprop(net.parameters(), lr=unet_lr, weight_decay=1e-8, momentum=0.9)
        self.scheduler_unet = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer_unet, "max", patience=5)
        self.optimizer_G = gs_optim.Adam(netG.parameters(), lr=gan_lr, betas=(beta1, 0.999))
        self.optimizer_D = gs_optim.Adam(netD.parameters(), lr=gan_lr, betas=(beta1, 0.999))
        self.optimizer_arch = torch.optim.Adam(networks.arch_parameters(), lr=arch_lr, betas=(0.5, 0.999), weight_decay=1e-5)
        self.global_step, self.val_best_score, self.history = 0, -1.0, []
        # this trainer owns every update of the three networks (fused optimisers and torch's Adam: all bump the parameters'
        # version counters; nothing writes through `.data`), so their engines may reuse version-keyed weight packs between
        # the training forwards of one iteration (the generator runs three times between two updates of its 1.09 GB of weights)
        for m in (net, netG, netD):
            eng = getattr(m, "engine", None)
            if eng is None and isinstance(m, (torch.nn.DataParallel,)):
                eng = getattr(m.module, "engine", None)
            if eng is not None and hasattr(eng, "trust_versions"):
                eng.trust_versions = True
        # the Unet problem sums two backward passes of the net: the second adds its gradients into `.grad` with ONE foreach launch
        # (unet_engine.accumulate_grads) -- switched on only around that problem's own forward + backward (_problem), because a
        # backward in that mode hands autograd `None` for gradients it has already accumulated (torch.autograd.grad, hooks and
        # retain_graph double use would miss them)
        self._accumulate_in_place = not data_parallel
        self._train_iter, self._val_iter = _cycle(train_loader), _cycle(val_loader)
        # HIP graphs (hip_graphs=True): at the script's batch size (2) an iteration is ~1,300 kernel launches of 2-100 us and the
        # Python / ctypes launch path (~16 us per launch) is what bounds it, not the GPU (tools/probe_graph_step.py: 105 -> 160
        # img/s).  After `graph_warmup` eager iterations the forward + backward of each of the three lower problems is captured
        # ONCE into a hipGraph (torch.cuda.CUDAGraph) and replayed; batches are copied into static input tensors, the mask
        # augmentation is drawn on the host outside the graph, optimiser steps / gradient exchange / arch step / validation stay
        # eager.  A graph reads nothing that eager code owns (and could free): inputs are the trainer's static tensors, weight packs
        # are rebuilt inside a graph after every update of their weights (the engines' pack caches hand a capture only entries
        # that were themselves built during a capture).
        self.hip_graphs, self.graph_warmup = bool(hip_graphs), int(graph_warmup)
        self._graphs: Dict[str, tuple] = {}
        self._static: Dict[str, torch.Tensor] = {}
        self._gpool, self._gkeep = None, []
        self._gstream = torch.cuda.Stream(device) if self.hip_graphs else None
        # data parallel (one process per GPU, torch.distributed initialised by the caller, every rank feeds its own shard
        # of the loaders): replicas start from rank 0's parameters / buffers; after every backward the gradients of the
        # problem that is about to step are averaged over the ranks in flat fp32 buckets (parallel.all_reduce_gradients: the
        # packing of bucket k+1 overlaps the collective of bucket k).  BatchNorm statistics and the Dice sums stay per rank
        # (SURVEY 8e).  The Unet problem accumulates two backward passes, so its exchange runs once, after both.
        self.data_parallel = bool(data_parallel)
        if self.data_parallel:
            from . import parallel
            for m in (net, netG, netD):
                parallel.broadcast_module_state(m)
            for a in networks.arch_parameters():
                if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                    torch.distributed.broadcast(a.data, src=0)

    # ---- one batch to the device in the layout the steps expect (JSRT_loader dict keys) ----
    def _batch(self, batch: Dict[str, torch.Tensor]):
        image = batch["image"].to(self.device, dtype=torch.float32, non_blocking=True)
        mask = batch["mask"].to(self.device, non_blocking=True)
        if mask.dim() == 3:
            mask = mask.unsqueeze(1)
        return image, mask

    def _exchange(self, params):
        if self.data_parallel:
            from . import parallel
            parallel.all_reduce_gradients(list(params))

    def _set_requires_grad(self, module, flag: bool):
        for p in module.parameters():
            p.requires_grad = flag

    # ---- HIP-graph plumbing ------------------------------------------------------------------------
    def _static_copy(self, name: str, t: torch.Tensor) -> torch.Tensor:
        """the graphs read their inputs from fixed tensors: copy the batch in (allocated at first use, shapes must not change)"""
        if not self.hip_graphs:
            return t
        st = self._static.get(name)
        if st is None:
            st = self._static[name] = t.detach().clone()
        else:
            if st.shape != t.shape or st.dtype != t.dtype:
                raise RuntimeError(f"hip_graphs: the shape of '{name}' changed ({tuple(st.shape)} -> {tuple(t.shape)}); use drop_last loaders")
            st.copy_(t, non_blocking=True)
        return st

    def _problem(self, name: str, loss_fn: Callable[[], torch.Tensor], optimizer, extra_none=()) -> torch.Tensor:
        """zero_grad + forward + backward of one problem: eager, or (hip_graphs, after the warm-up iterations) captured once into
        a hipGraph and replayed.  Returns the detached loss."""
        def run():
            """forward + backward; a tuple of losses is back-propagated term by term (see steps.unet_step_losses)"""
            loss = loss_fn()
            if isinstance(loss, tuple):
                for term in loss:
                    term.backward()
                loss = sum(t.detach() for t in loss)
            else:
                loss.backward()
            return loss.detach()

        inner_run = run
        u_eng = getattr(self.net, "engine", None) if (name == "U" and self._accumulate_in_place) else None

        def run():
            if u_eng is None or not hasattr(u_eng, "accumulate_grads"):
                return inner_run()
            u_eng.accumulate_grads = True
            try:
                return inner_run()
            finally:
                u_eng.accumulate_grads = False

        if not self.hip_graphs or self.global_step < self.graph_warmup:
            optimizer.zero_grad(set_to_none=True)
            return run()
        ent = self._graphs.get(name)
        if ent is None and name == "G" and self.train_gan and "D" not in self._graphs:
            # the Generator problem is captured one iteration after the other two: its graph then REUSES the generator's forward
            # packs that the Discriminator graph rebuilds after every update of G (captured entries of the engines' pack caches,
            # see _PackCache) instead of merging the 1.09 GB of fp32 kernels a second time per iteration
            optimizer.zero_grad(set_to_none=True)
            return run()
        if ent is None:
            from . import ops
            optimizer.zero_grad(set_to_none=True)              # the captured backward allocates the gradients in the graph's pool
            for a in extra_none:
                a.grad = None
            # nothing a graph reads may be owned by eager code (which could free it): the pack caches of the Pix2Pix engines hand
            # a capture only entries that were built during a capture; the U-Net's packs are rebuilt inside its graph
            eng = getattr(self.net, "engine", None)
            if eng is not None and hasattr(eng, "invalidate_packs"):
                eng.invalidate_packs()
            self._gkeep.append(ops.splitk_workspace_for_capture(self.device))      # exists before the capture and is never evicted
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=self._gpool, stream=self._gstream):
                loss = run()
            if self._gpool is None:
                self._gpool = graph.pool()
            grads = [(p, p.grad) for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
            ent = self._graphs[name] = (graph, loss, grads)
        graph, loss, grads = ent
        graph.replay()
        for p, g in grads:                                      # (eager code in between -- the arch step -- may have dropped them)
            p.grad = g
        return loss.clone()

    def train_iteration(self) -> Dict[str, "float | torch.Tensor"]:
        """One lower-level iteration.  Losses come back as 0-dim device tensors (convert with float() when they are needed:
        a conversion per step would make the host wait for the GPU four times per iteration)."""
        if not self.hip_graphs:
            return self._train_iteration()
        self._gstream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._gstream):
            out = self._train_iteration()
        torch.cuda.current_stream().wait_stream(self._gstream)
        return out

    def _train_iteration(self) -> Dict[str, "float | torch.Tensor"]:
        image, mask = self._batch(next(self._train_iter))
        aug = None
        if self.hip_graphs:                                     # static inputs; the host-side augmentation stays outside the graphs
            if self.mask_augment is not None:
                aug = self._static_copy("aug", self.mask_augment(mask.float()))
            image, mask = self._static_copy("image", image), self._static_copy("mask", mask)
        real_mask, real_image = (self._static_copy("real_mask", mask.float()) if self.hip_graphs else mask.float()), image
        out: Dict[str, float] = {}
        if self.train_gan:
            # Generator problem (:141-154): D frozen, GAN + lambda*L1
            self._set_requires_grad(self.netD, False)
            loss_g = self._problem("G", lambda: steps.generator_step_loss(self.netG, self.netD, self.criterionGAN, real_mask,
                                                                           real_image, self.lambda_L1),
                                   self.optimizer_G, extra_none=networks.arch_parameters())
            self._exchange(self.netG.parameters())
            self.optimizer_G.step()
            # Discriminator problem (:157-172)
            self._set_requires_grad(self.netD, True)
            loss_d = self._problem("D", lambda: steps.discriminator_step_loss(self.netG, self.netD, self.criterionGAN, real_mask,
                                                                               real_image), self.optimizer_D)
            self._exchange(self.netD.parameters())
            self.optimizer_D.step()
            out.update(loss_G=loss_g, loss_D=loss_d)            # 0-dim device tensors: no host sync per step
        # Unet problem (:176-226): real pair + generated pair from the (augmented) masks
        loss_u = self._problem("U", lambda: steps.unet_step_losses(self.net, self.netG, image, mask, self.loss_lambda,
                                                                    None if self.hip_graphs else self.mask_augment, aug),
                               self.optimizer_unet)
        self._exchange(self.net.parameters())
        self.optimizer_unet.step()
        out["loss_unet"] = loss_u
        self.global_step += 1
        # Arch problem (:229-236) on validation data, every unroll_steps lower iterations
        if self.global_step % self.unroll_steps == 0:
            v_image, v_mask = self._batch(next(self._val_iter))
            self.optimizer_arch.zero_grad(set_to_none=True)
            if self.arch_through_generator:
                fake = self.netG(v_mask.float())
                loss_a = steps.arch_step_loss(self.net, fake, v_mask)
            else:
                loss_a = steps.arch_step_loss(self.net, v_image, v_mask)
            for p in self.net.parameters():                    # the arch optimiser owns only the arch tensors
                p.grad = None
            loss_a.backward()
            self._exchange(networks.arch_parameters())
            if any(a.grad is not None for a in networks.arch_parameters()):
                self.optimizer_arch.step()
                if self.hip_graphs and self._graphs:
                    # The Generator graph READS the merged forward packs that the Discriminator graph writes (captured one
                    # iteration apart, see _problem); this eager update of the arch tensor falls between that writer's last replay
                    # and the Generator graph's next one, whose in-graph softmax / data-gradient packs / gradient split already use
                    # the NEW arch: bring the captured forward packs up to date in place (forward and backward of the
                    # G problem disagreed whenever (graph_warmup + 1) % unroll_steps != 0)
                    eng = getattr(self.netG, "engine", None)
                    if eng is not None and hasattr(eng, "refresh_arch_packs"):
                        eng.refresh_arch_packs(networks.upconv_arch)
            for p in list(self.net.parameters()) + list(self.netG.parameters()):
                p.grad = None
            out["loss_arch"] = loss_a.detach()
        if self.valid_every and self.global_step % self.valid_every == 0:
            out["val_score"] = self.validation()
        self.history.append(out)
        return out

    @torch.no_grad()
    def validation(self) -> float:
        """SSEngine.validation (:239-255)."""
        score = float(evaluate(self.net, self.val_loader, self.device, False))
        if score > self.val_best_score:
            self.val_best_score = score
            if self.save_dir:
                os.makedirs(self.save_dir, exist_ok=True)
                torch.save(self.net.state_dict(), os.path.join(self.save_dir, "unet.pkl"))
        n_train = max(len(self.train_loader), 1)
        if self.global_step % n_train == 0 and self.global_step:
            self.scheduler_unet.step(self.val_best_score)
        return score

    def run(self, iters: int, log_every: int = 10) -> None:
        t0 = time.perf_counter()
        for _ in range(iters):
            out = self.train_iteration()
            if log_every and self.global_step % log_every == 0:
                msg = " ".join(f"{k}={v:.4f}" for k, v in out.items())
                print(f"[{self.global_step:6d}] {msg}  ({(time.perf_counter() - t0) / self.global_step * 1e3:.1f} ms/iter)", flush=True)
        if self.save_dir:
            os.makedirs(self.save_dir, exist_ok=True)
            torch.save(self.net.state_dict(), os.path.join(self.save_dir, "final.pkl"))      # :345


class SyntheticLungDataset(torch.utils.data.Dataset):
    """Stand-in for util/JSRT_loader.py's BasicDataset (needs torchvision + private data): emits the same dict keys
    (`image` float [1,H,W] in [0,1], `mask` int64 [1,H,W]) with 1-3 filled ellipses per image."""

    def __init__(self, n: int, size: int = 256, seed: int = 0):
        self.n, self.size, self.seed = n, size, seed
        self._grid = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32),
                                    indexing="ij")
        self._items = {}                 # samples are deterministic in (seed, i): generated once (32 ms each on the host)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if i not in self._items:
            self._items[i] = self._make(i)
        return self._items[i]

    def _make(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        s = self.size
        yy, xx = self._grid
        mask = torch.zeros(s, s, dtype=torch.bool)
        for _ in range(int(torch.randint(1, 4, (1,), generator=g))):
            cy, cx = (torch.rand(2, generator=g) * 0.6 + 0.2) * s
            ry, rx = (torch.rand(2, generator=g) * 0.27 + 0.08) * s
            mask |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        m = mask.float()
        image = (0.25 + 0.5 * m + 0.1 * torch.randn(s, s, generator=g)).clamp(0, 1)
        return {"image": image.unsqueeze(0), "mask": mask.long().unsqueeze(0)}


def synthetic_batch(batch: int, size: int, seed: int = 1234, n_channels: int = 1):
    """SURVEY 8(d) synthetic inputs for the benchmarks: standard-normal images [B,C,S,S] and int64 masks [B,1,S,S]
    holding 1-3 random filled ellipses each (host tensors; the caller moves them to the GPU)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, n_channels, size, size, generator=g)
    yy, xx = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32), indexing="ij")
    mask = torch.zeros(batch, 1, size, size, dtype=torch.long)
    sc = size / 256.0
    for b in range(batch):
        for _ in range(int(torch.randint(1, 4, (1,), generator=g))):
            cy, cx = (torch.rand(2, generator=g) * size).tolist()
            ry, rx = ((torch.rand(2, generator=g) * 70 + 20) * sc).tolist()
            mask[b, 0] |= (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0).long()
    return x, mask


def main(argv=None):
    ap = argparse.ArgumentParser(description="Betty-free GenSeg end-to-end loop on the MI355X kernels")
    ap.add_argument("--synthetic", action="store_true", help="synthetic lung-like data (no dataset ships with the repo)")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch-size", type=int, default=2)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--classes", type=int, default=1)
    ap.add_argument("--ngf", type=int, default=64)
    ap.add_argument("--unet-lr", type=float, default=1e-5)
    ap.add_argument("--save-dir", default=None)
    ap.add_argument("--no-augment", action="store_true", help="do not augment the masks that feed the generator")
    ap.add_argument("--no-gan", action="store_true", help="freeze the Pix2Pix pair (only the Unet / Arch problems step)")
    ap.add_argument("--hip-graphs", action="store_true", help="capture forward + backward of the three lower problems into hipGraphs")
    args = ap.parse_args(argv)
    if not args.synthetic:
        raise SystemExit("only --synthetic data ships with this repository; build the loaders of your dataset and use EndToEndTrainer")
    dev = torch.device("cuda", 0)
    net = UNet(1, args.classes).to(dev)
    netG = networks.define_G(1, 1, args.ngf, "unet_256", "batch", True, "normal", 0.02, [0])
    netD = networks.define_D(2, 64, "basic", 3, "batch", "normal", 0.02, [0])
    crit = networks.GANLoss("vanilla").to(dev)
    mk = lambda n, seed: torch.utils.data.DataLoader(SyntheticLungDataset(n, args.size, seed), batch_size=args.batch_size,
                                                     shuffle=True, drop_last=True)
    trainer = EndToEndTrainer(net, netG, netD, crit, mk(8 * args.batch_size, 1), mk(2 * args.batch_size, 2), dev,
                              unet_lr=args.unet_lr, save_dir=args.save_dir, train_gan=not args.no_gan,
                              mask_augment=None if args.no_augment else MaskAugmenter(seed=0), hip_graphs=args.hip_graphs)
    trainer.run(args.iters, log_every=5)


if __name__ == "__main__":
    main()
