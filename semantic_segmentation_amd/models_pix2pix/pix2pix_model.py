"""`create_model(opt)` object of the reference (models_pix2pix/pix2pix_model.py:9-178 over base_model.py:7-232):
the stage-1 GAN pre-training model and the container train_end2end_jsrt.py:55-59 builds netG / netD from.

Same public surface -- `.netG .netD .optimizer_G .optimizer_D .optimizer_arch_upconv .optimizer_arch_conv
.setup(opt) .set_input .set_input_1 .forward .optimize_parameters .optimize_architect .save_model .load_model
.eval .test .update_learning_rate .get_current_losses .get_current_visuals .set_requires_grad` -- with the
networks, GAN loss and L1 running on the HIP kernels.  Differences, all deliberate:
  * no torchvision dependency (the reference builds an unused transforms.Normalize, pix2pix_model.py:52-54);
  * the architecture tensors stay the leaf tensors the optimisers own (the reference replaces the module globals
    by non-leaf `.cuda()` copies at :59-60, so its arch optimisers never see a gradient -- SURVEY section 3.3);
    `save_model` additionally writes them (`pix2pix_arch.pkl`) and `load_model` reads them when present;
  * one process per GPU: `gpu_ids` picks the device, nn.DataParallel is never used.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch
from torch.optim import lr_scheduler

from . import networks
from ..losses import l1_loss


def get_scheduler(optimizer, opt):
    """linear | step | plateau | cosine, as networks.py:44-70 of the reference."""
    policy = getattr(opt, "lr_policy", "linear")
    if policy == "linear":
        n_epochs, n_decay, start = opt.n_epochs, opt.n_epochs_decay, getattr(opt, "epoch_count", 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda e: 1.0 - max(0, e + start - n_epochs) / float(n_decay + 1))
    if policy == "step":
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if policy == "plateau":
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.2, threshold=0.01, patience=5)
    if policy == "cosine":
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.n_epochs, eta_min=0)
    raise NotImplementedError("learning rate policy [%s] is not implemented" % policy)


class Pix2PixModel:
    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        parser.set_defaults(norm="batch", netG="unet_256", dataset_mode="aligned")
        if is_train:
            parser.set_defaults(pool_size=0, gan_mode="vanilla")
            parser.add_argument("--lambda_L1", type=float, default=100.0, help="weight for L1 loss")
        return parser

    def __init__(self, opt):
        self.opt = opt
        self.gpu_ids = list(getattr(opt, "gpu_ids", [0]))
        self.isTrain = bool(getattr(opt, "isTrain", True))
        if not torch.cuda.is_available():
            raise RuntimeError("Pix2PixModel (semantic_segmentation_amd) runs on the MI355X only (no CPU fallback)")
        index = getattr(opt, "cuda_index", self.gpu_ids[0] if self.gpu_ids else 0)
        self.device = torch.device("cuda", index)
        torch.cuda.set_device(self.device)        # one process per GPU: kernels launch on the CURRENT device's stream
        self.save_dir = os.path.join(getattr(opt, "checkpoints_dir", "./checkpoints"), getattr(opt, "name", "pix2pix"))
        self.loss_names = ["G_GAN", "G_L1", "D_real", "D_fake"]
        self.visual_names = ["real_mask", "fake_image", "real_image"]
        self.model_names = ["G", "D"] if self.isTrain else ["G"]
        self.optimizers, self.schedulers, self.image_paths, self.metric = [], [], [], 0
        dev_ids = [index]
        self.netG = networks.define_G(opt.input_nc, opt.output_nc, opt.ngf, opt.netG, opt.norm, not opt.no_dropout,
                                      opt.init_type, opt.init_gain, dev_ids)
        if self.isTrain:
            self.netD = networks.define_D(opt.input_nc + opt.output_nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm,
                                          opt.init_type, opt.init_gain, dev_ids)
            self.criterionGAN = networks.GANLoss(opt.gan_mode).to(self.device)
            self.criterionL1 = l1_loss
            betas = (opt.beta1, 0.999)
            self.optimizer_G = torch.optim.Adam(self.netG.parameters(), lr=opt.lr, betas=betas)
            self.optimizer_D = torch.optim.Adam(self.netD.parameters(), lr=opt.lr, betas=betas)
            arch_lr = getattr(opt, "arch_lr", 3e-4)
            self.optimizer_arch_upconv = torch.optim.Adam(networks.upconv_arch_parameters(), lr=arch_lr,
                                                          betas=(0.5, 0.999), weight_decay=1e-3)
            self.optimizer_arch_conv = torch.optim.Adam(networks.conv_arch_parameters(), lr=arch_lr,
                                                        betas=(0.5, 0.999), weight_decay=1e-3)
            self.optimizers += [self.optimizer_G, self.optimizer_D, self.optimizer_arch_upconv, self.optimizer_arch_conv]

    # ---- base_model.py surface -------------------------------------------------------------------
    def setup(self, opt):
        if self.isTrain:
            self.schedulers = [get_scheduler(o, opt) for o in self.optimizers]
        if not self.isTrain or getattr(opt, "continue_train", False):
            suffix = "iter_%d" % opt.load_iter if getattr(opt, "load_iter", 0) > 0 else getattr(opt, "epoch", "latest")
            self.load_networks(suffix)
        self.print_networks(getattr(opt, "verbose", False))

    def eval(self):
        for name in self.model_names:
            getattr(self, "net" + name).eval()

    def train(self):
        for name in self.model_names:
            getattr(self, "net" + name).train()

    def test(self):
        with torch.no_grad():
            self.forward()

    def get_image_paths(self):
        return self.image_paths

    def update_learning_rate(self):
        old_lr = self.optimizers[0].param_groups[0]["lr"]
        for s in self.schedulers:
            if isinstance(s, lr_scheduler.ReduceLROnPlateau):
                s.step(self.metric)
            else:
                s.step()
        print("learning rate %.7f -> %.7f" % (old_lr, self.optimizers[0].param_groups[0]["lr"]))

    def get_current_visuals(self):
        return OrderedDict((n, getattr(self, n)) for n in self.visual_names if hasattr(self, n))

    def get_current_losses(self):
        return OrderedDict((n, float(getattr(self, "loss_" + n).detach())) for n in self.loss_names if hasattr(self, "loss_" + n))

    def save_networks(self, epoch):
        os.makedirs(self.save_dir, exist_ok=True)
        for name in self.model_names:
            torch.save(getattr(self, "net" + name).state_dict(), os.path.join(self.save_dir, "%s_net_%s.pth" % (epoch, name)))

    def load_networks(self, epoch):
        for name in self.model_names:
            path = os.path.join(self.save_dir, "%s_net_%s.pth" % (epoch, name))
            print("loading the model from %s" % path)
            getattr(self, "net" + name).load_state_dict(torch.load(path, map_location=str(self.device)))

    def print_networks(self, verbose):
        for name in self.model_names:
            net = getattr(self, "net" + name)
            if verbose:
                print(net)
            print("[Network %s] Total number of parameters : %.3f M" % (name, sum(p.numel() for p in net.parameters()) / 1e6))

    def set_requires_grad(self, nets, requires_grad=False):
        for net in nets if isinstance(nets, list) else [nets]:
            if net is not None:
                for p in net.parameters():
                    p.requires_grad = requires_grad

    # ---- pix2pix_model.py surface ----------------------------------------------------------------
    def set_input(self, image, mask):
        self.real_mask = mask.to(self.device, dtype=torch.float32)
        self.real_image = image.to(self.device, dtype=torch.float32)

    def set_input_1(self, input):
        self.set_input(input["image"], input["mask"])

    def forward(self):
        self.fake_image = self.netG(self.real_mask)

    def backward_D(self):
        pred_fake = self.netD(torch.cat((self.real_mask, self.fake_image), 1).detach())
        self.loss_D_fake = self.criterionGAN(pred_fake, False)
        pred_real = self.netD(torch.cat((self.real_mask, self.real_image), 1))
        self.loss_D_real = self.criterionGAN(pred_real, True)
        self.loss_D = (self.loss_D_fake + self.loss_D_real) * 0.5
        self.loss_D.backward()

    def _generator_loss(self, fake_image, real_mask, real_image):
        pred_fake = self.netD(torch.cat((real_mask, fake_image), 1))
        loss_gan = self.criterionGAN(pred_fake, True)
        loss_l1 = self.criterionL1(fake_image, real_image) * self.opt.lambda_L1
        return loss_gan, loss_l1

    def backward_G(self):
        self.loss_G_GAN, self.loss_G_L1 = self._generator_loss(self.fake_image, self.real_mask, self.real_image)
        self.loss_G = self.loss_G_GAN + self.loss_G_L1
        self.loss_G.backward()

    def optimize_parameters(self):
        self.forward()
        self.set_requires_grad(self.netD, True)
        self.optimizer_D.zero_grad()
        self.backward_D()
        self.optimizer_D.step()
        self.set_requires_grad(self.netD, False)
        self.optimizer_G.zero_grad()
        self.backward_G()
        self.optimizer_G.step()

    def optimize_architect(self, image, mask):
        real_mask = mask.to(self.device, dtype=torch.float32)
        real_image = image.to(self.device, dtype=torch.float32)
        self.set_requires_grad(self.netD, False)
        self.optimizer_arch_upconv.zero_grad()
        self.optimizer_arch_conv.zero_grad()
        fake_image = self.netG(real_mask)
        loss_gan, loss_l1 = self._generator_loss(fake_image, real_mask, real_image)
        (loss_gan + loss_l1).backward()
        self.optimizer_arch_upconv.step()
        if networks.conv_arch.grad is not None:       # the conv arch row is unused by unet_256 (networks.py:441-484)
            self.optimizer_arch_conv.step()

    def save_model(self, save_path):
        os.makedirs(save_path, exist_ok=True)
        torch.save(self.netD.state_dict(), os.path.join(save_path, "pix2pix_discriminator.pkl"))
        torch.save(self.netG.state_dict(), os.path.join(save_path, "pix2pix_generator.pkl"))
        torch.save({"upconv_arch": networks.upconv_arch.detach().cpu(), "conv_arch": networks.conv_arch.detach().cpu()},
                   os.path.join(save_path, "pix2pix_arch.pkl"))

    def load_model(self, D_model_filename, G_model_filename):
        d_path = os.path.join(os.getcwd(), D_model_filename)
        g_path = os.path.join(os.getcwd(), G_model_filename)
        self.netD.load_state_dict(torch.load(d_path, map_location=str(self.device)))
        self.netG.load_state_dict(torch.load(g_path, map_location=str(self.device)))
        arch_path = os.path.join(os.path.dirname(g_path), "pix2pix_arch.pkl")
        if os.path.exists(arch_path):
            arch = torch.load(arch_path, map_location="cpu")
            with torch.no_grad():
                networks.upconv_arch.copy_(arch["upconv_arch"])
                networks.conv_arch.copy_(arch["conv_arch"])
