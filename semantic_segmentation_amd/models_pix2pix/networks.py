"""Pix2Pix networks of GenSeg on the MI355X HIP engine (reference: models_pix2pix/networks.py).

Same public names, module tree and state-dict keys as the reference for the parts reached by
`--model pix2pix` defaults: `UnetGenerator` (+ `UnetSkipConnectionBlock`, `Cell_upconv`, `MixedOp_upconv`),
`NLayerDiscriminator`, `GANLoss`, `get_norm_layer`, `init_weights`, `init_net`, `define_G`, `define_D`, and the
architecture tensors `upconv_arch` / `conv_arch` / `arch_parameters()` (module globals, :443,:477-484).
The leaf torch modules only hold parameters; `UnetGenerator.forward` and `NLayerDiscriminator.forward`
run whole-network plans on the HIP kernels (pix2pix_engine.py).  No ATen / CPU fallback."""
import functools

import torch
import torch.nn as nn
from torch.nn import init

from ..architecture_pix2pix.genotypes import PRIMITIVES_conv, PRIMITIVES_upconv
from ..architecture_pix2pix.operations import OPS
from ..losses import MODE_BCE_CONST, MODE_MEAN, MODE_MSE_CONST, mean_loss


class Identity(nn.Module):
    def forward(self, x):
        return x


def get_norm_layer(norm_type='instance'):
    """networks.py:23-41.  Only 'batch' (the pix2pix default, pix2pix_model.py:34) runs on the HIP engine."""
    if norm_type == 'batch':
        return functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True)
    if norm_type == 'instance':
        return functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False)
    if norm_type == 'none':
        return lambda x: Identity()
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


def init_weights(net, init_type='normal', init_gain=0.02):
    """networks.py:73-104 (same distributions; applied to the parameter-holding leaf modules)."""
    def init_func(m):
        classname = m.__class__.__name__
        if hasattr(m, 'weight') and (classname.find('Conv') != -1 or classname.find('Linear') != -1):
            if init_type == 'normal':
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            if hasattr(m, 'bias') and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif classname.find('BatchNorm2d') != -1:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)
    net.apply(init_func)


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    """networks.py:107-122.  Multi-GPU is one process per GPU (parallel.py), never nn.DataParallel."""
    if len(gpu_ids) > 1:
        raise NotImplementedError("use one process per GPU (semantic_segmentation_amd.parallel), not nn.DataParallel")
    if len(gpu_ids) == 1:
        assert torch.cuda.is_available()
        net.to(gpu_ids[0])
    init_weights(net, init_type, init_gain=init_gain)
    return net


def define_G(input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, init_type='normal', init_gain=0.02,
             gpu_ids=[]):
    norm_layer = get_norm_layer(norm_type=norm)
    if netG == 'unet_128':
        net = UnetGenerator(input_nc, output_nc, 7, ngf, norm_layer=norm_layer, use_dropout=use_dropout)
    elif netG == 'unet_256':
        net = UnetGenerator(input_nc, output_nc, 8, ngf, norm_layer=norm_layer, use_dropout=use_dropout)
    else:
        raise NotImplementedError('Generator model name [%s] is not on the MI355X hot path' % netG)
    return init_net(net, init_type, init_gain, gpu_ids)


def define_D(input_nc, ndf, netD, n_layers_D=3, norm='batch', init_type='normal', init_gain=0.02, gpu_ids=[]):
    norm_layer = get_norm_layer(norm_type=norm)
    if netD == 'basic':
        net = NLayerDiscriminator(input_nc, ndf, n_layers=3, norm_layer=norm_layer)
    elif netD == 'n_layers':
        net = NLayerDiscriminator(input_nc, ndf, n_layers_D, norm_layer=norm_layer)
    else:
        raise NotImplementedError('Discriminator model name [%s] is not on the MI355X hot path' % netD)
    return init_net(net, init_type, init_gain, gpu_ids)


class GANLoss(nn.Module):
    """networks.py:215-281: vanilla = BCEWithLogits vs a constant label, lsgan = MSE, wgangp = -+mean; the
    label is never materialised (the reduction kernel takes it as a scalar)."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super(GANLoss, self).__init__()
        self.register_buffer('real_label', torch.tensor(target_real_label))
        self.register_buffer('fake_label', torch.tensor(target_fake_label))
        self.gan_mode = gan_mode
        if gan_mode not in ('lsgan', 'vanilla', 'wgangp'):
            raise NotImplementedError('gan mode %s not implemented' % gan_mode)
        self._labels = (float(target_real_label), float(target_fake_label))

    def get_target_tensor(self, prediction, target_is_real):
        return (self.real_label if target_is_real else self.fake_label).expand_as(prediction)

    def __call__(self, prediction, target_is_real):
        label = self._labels[0] if target_is_real else self._labels[1]
        if self.gan_mode == 'vanilla':
            return mean_loss(prediction, None, label, MODE_BCE_CONST)
        if self.gan_mode == 'lsgan':
            return mean_loss(prediction, None, label, MODE_MSE_CONST)
        return mean_loss(prediction, None, -1.0 if target_is_real else 1.0, MODE_MEAN)


# ---- architecture tensors (module globals like the reference: networks.py:441-484) -------------------
num_ops_conv = len(PRIMITIVES_conv)
conv_arch = (1e-3 * torch.randn(8, num_ops_conv)).requires_grad_(True)


def conv_arch_parameters():
    """The tensor the generator reads NOW: the reference returns a list captured at import time, which goes
    stale as soon as a script rebinds `networks.conv_arch` (pix2pix_model.py:59 does) -- its arch optimisers
    then own a tensor the forward pass never uses."""
    return [conv_arch, ]


num_ops_upconv = len(PRIMITIVES_upconv)
upconv_arch = (1e-3 * torch.randn(8, num_ops_upconv)).requires_grad_(True)


def upconv_arch_parameters():
    return [upconv_arch, ]


def arch_parameters():
    return [upconv_arch, conv_arch]


class MixedOp_upconv(nn.Module):
    """networks.py:486-496: holds the three transposed-conv primitives (`_ops.{0,1,2}.op.*`)."""

    def __init__(self, C_in, C_out, bias):
        super(MixedOp_upconv, self).__init__()
        self._ops = nn.ModuleList()
        for primitive in PRIMITIVES_upconv:
            self._ops.append(OPS[primitive](C_in, C_out, bias))

    def forward(self, x, weights):
        """sum(w * op(x)) (networks.py:495-496) as ONE merged transposed conv on the HIP kernels (cell_engine.py)."""
        from .cell_engine import mixed_upconv
        return mixed_upconv(x, weights, list(self._ops))


class Cell_upconv(nn.Module):
    """networks.py:499-511."""

    def __init__(self, C_in, C_out, bias, layer_index):
        super(Cell_upconv, self).__init__()
        self._layer_index = layer_index
        self._ops = MixedOp_upconv(C_in, C_out, bias)

    def forward(self, input):
        """networks.py:507-511: softmax of this cell's row of the architecture tensor, then the mixed op."""
        weights = torch.softmax(upconv_arch[self._layer_index].to(input.device), dim=-1)
        return self._ops(input, weights)


class UnetSkipConnectionBlock(nn.Module):
    """networks.py:553-617: same Sequential layout (hence the same state-dict keys)."""

    def __init__(self, outer_nc, inner_nc, input_nc=None, layer_index=None, submodule=None, outermost=False,
                 innermost=False, norm_layer=nn.BatchNorm2d, use_dropout=False):
        super(UnetSkipConnectionBlock, self).__init__()
        self.outermost = outermost
        self.innermost = innermost
        if type(norm_layer) == functools.partial:
            use_bias = norm_layer.func == nn.InstanceNorm2d
        else:
            use_bias = norm_layer == nn.InstanceNorm2d
        if use_bias:
            raise NotImplementedError("InstanceNorm generators are not on the MI355X hot path (pix2pix uses norm='batch')")
        if input_nc is None:
            input_nc = outer_nc
        downconv = nn.Conv2d(input_nc, inner_nc, kernel_size=4, stride=2, padding=1, bias=use_bias)
        downrelu = nn.LeakyReLU(0.2, True)
        downnorm = norm_layer(inner_nc)
        uprelu = nn.ReLU(True)
        upnorm = norm_layer(outer_nc)
        if outermost:
            upconv = Cell_upconv(inner_nc * 2, outer_nc, bias=True, layer_index=layer_index)
            model = [downconv] + [submodule] + [uprelu, upconv, nn.Tanh()]
        elif innermost:
            upconv = Cell_upconv(inner_nc, outer_nc, bias=use_bias, layer_index=layer_index)
            model = [downrelu, downconv] + [uprelu, upconv, upnorm]
        else:
            upconv = Cell_upconv(inner_nc * 2, outer_nc, bias=use_bias, layer_index=layer_index)
            model = [downrelu, downconv, downnorm] + [submodule] + [uprelu, upconv, upnorm]
            if use_dropout:
                model = model + [nn.Dropout(0.5)]
        self.model = nn.Sequential(*model)

    def forward(self, x):
        raise RuntimeError("UnetSkipConnectionBlock runs inside UnetGenerator.forward (HIP engine)")


class UnetGenerator(nn.Module):
    """networks.py:514-550."""

    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False,
                 compute_dtype=None):
        super(UnetGenerator, self).__init__()
        self.layer_index = 0
        unet_block = UnetSkipConnectionBlock(ngf * 8, ngf * 8, input_nc=None, layer_index=self.layer_index,
                                             submodule=None, norm_layer=norm_layer, innermost=True)
        self.layer_index += 1
        for i in range(num_downs - 5):
            unet_block = UnetSkipConnectionBlock(ngf * 8, ngf * 8, input_nc=None, layer_index=self.layer_index,
                                                 submodule=unet_block, norm_layer=norm_layer, use_dropout=use_dropout)
            self.layer_index += 1
        for mult in (4, 2, 1):
            unet_block = UnetSkipConnectionBlock(ngf * mult, ngf * mult * 2, input_nc=None,
                                                 layer_index=self.layer_index, submodule=unet_block,
                                                 norm_layer=norm_layer)
            self.layer_index += 1
        self.model = UnetSkipConnectionBlock(output_nc, ngf, input_nc=input_nc, layer_index=self.layer_index,
                                             submodule=unet_block, outermost=True, norm_layer=norm_layer)
        self.layer_index += 1
        from .pix2pix_engine import GeneratorEngine
        object.__setattr__(self, "_engine", GeneratorEngine(self, compute_dtype))

    @property
    def engine(self):
        return self._engine

    def forward(self, input, dropout_masks=None):
        """`dropout_masks`: optional list of uint8 keep-masks (NHWC) for the dropout blocks, innermost first;
        default draws them with torch.rand on the device (train mode) -- not the ATen Philox stream."""
        return self._engine.run(input, dropout_masks)


class NLayerDiscriminator(nn.Module):
    """networks.py:620-665 PatchGAN (same Sequential indices: convs at 0,2,5,8,11; norms at 3,6,9)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, compute_dtype=None):
        super(NLayerDiscriminator, self).__init__()
        if type(norm_layer) == functools.partial:
            use_bias = norm_layer.func == nn.InstanceNorm2d
        else:
            use_bias = norm_layer == nn.InstanceNorm2d
        if use_bias:
            raise NotImplementedError("InstanceNorm discriminators are not on the MI355X hot path")
        kw, padw = 4, 1
        sequence = [nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_mult_prev, nf_mult = nf_mult, min(2 ** n, 8)
            sequence += [nn.Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=2, padding=padw,
                                   bias=use_bias), norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        nf_mult_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        sequence += [nn.Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=1, padding=padw,
                               bias=use_bias), norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        sequence += [nn.Conv2d(ndf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.model = nn.Sequential(*sequence)
        from .pix2pix_engine import DiscriminatorEngine
        object.__setattr__(self, "_engine", DiscriminatorEngine(self, compute_dtype))

    @property
    def engine(self):
        return self._engine

    def forward(self, input):
        return self._engine.run(input)
