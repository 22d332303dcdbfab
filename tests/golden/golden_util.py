"""Helpers shared by make_golden.py (runs in the build container, imports the
reference) and by the tests (run anywhere, never import the reference)."""
import math
from typing import Dict

import numpy as np
import torch

N_SAMPLES = 32


def sample_index(n: int) -> np.ndarray:
    """Deterministic spread of N_SAMPLES flat indices over a tensor of n elements."""
    return np.unique(np.linspace(0, n - 1, N_SAMPLES).astype(np.int64))


def grad_summary(g: torch.Tensor) -> np.ndarray:
    """[float64 sum, float64 L2 norm, sampled elements...] of a (large) gradient tensor."""
    f = g.detach().double().flatten()
    idx = torch.from_numpy(sample_index(f.numel()))
    return np.concatenate([[f.sum().item(), f.norm().item()], f[idx].numpy()])


def tensor_checksum(t: torch.Tensor) -> np.ndarray:
    f = t.detach().double().flatten()
    return np.array([f.sum().item(), f.abs().sum().item(), f[0].item(), f[-1].item()])


def _normal(g, shape, std, mean=0.0):
    return torch.randn(shape, generator=g) * std + mean


def seeded_generator_state_dict(seed: int, input_nc=1, output_nc=1, num_downs=8, ngf=64) -> Dict[str, torch.Tensor]:
    """State dict with the 100 key names/shapes of the reference
    UnetGenerator(1,1,8,64,BatchNorm,use_dropout) (models_pix2pix/networks.py:514-617);
    values ~ init_net 'normal' (conv N(0,0.02), BN gamma N(1,0.02), biases 0; networks.py:84-104)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def bn(name, c):
        sd[name + ".weight"] = _normal(g, (c,), 0.02, 1.0)
        sd[name + ".bias"] = torch.zeros(c)
        sd[name + ".running_mean"] = torch.zeros(c)
        sd[name + ".running_var"] = torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def cell(name, cin, cout, bias):
        for j, k in enumerate((4, 6, 8)):
            sd[f"{name}._ops._ops.{j}.op.weight"] = _normal(g, (cin, cout, k, k), 0.02)
            if bias:
                sd[f"{name}._ops._ops.{j}.op.bias"] = torch.zeros(cout)

    # (outer_nc, inner_nc) from outermost to innermost
    chans = [(output_nc, ngf), (ngf, ngf * 2), (ngf * 2, ngf * 4), (ngf * 4, ngf * 8)]
    chans += [(ngf * 8, ngf * 8)] * (num_downs - 4)
    prefix = "model"
    for depth, (outer, inner) in enumerate(chans):
        p = prefix + ".model"
        outermost, innermost = depth == 0, depth == num_downs - 1
        cin = input_nc if outermost else outer
        if outermost:
            sd[p + ".0.weight"] = _normal(g, (inner, cin, 4, 4), 0.02)
            cell(p + ".3", inner * 2, outer, True)
            prefix = p + ".1"
        elif innermost:
            sd[p + ".1.weight"] = _normal(g, (inner, cin, 4, 4), 0.02)
            cell(p + ".3", inner, outer, False)
            bn(p + ".4", outer)
        else:
            sd[p + ".1.weight"] = _normal(g, (inner, cin, 4, 4), 0.02)
            bn(p + ".2", inner)
            cell(p + ".5", inner * 2, outer, False)
            bn(p + ".6", outer)
            prefix = p + ".3"
    return sd


def seeded_discriminator_state_dict(seed: int, input_nc=2, ndf=64) -> Dict[str, torch.Tensor]:
    """22 keys of NLayerDiscriminator(2,64,3,BatchNorm) (networks.py:620-665)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    sd["model.0.weight"] = _normal(g, (ndf, input_nc, 4, 4), 0.02)
    sd["model.0.bias"] = torch.zeros(ndf)
    for conv_i, bn_i, ci, co in ((2, 3, ndf, ndf * 2), (5, 6, ndf * 2, ndf * 4), (8, 9, ndf * 4, ndf * 8)):
        sd[f"model.{conv_i}.weight"] = _normal(g, (co, ci, 4, 4), 0.02)
        sd[f"model.{bn_i}.weight"] = _normal(g, (co,), 0.02, 1.0)
        sd[f"model.{bn_i}.bias"] = torch.zeros(co)
        sd[f"model.{bn_i}.running_mean"] = torch.zeros(co)
        sd[f"model.{bn_i}.running_var"] = torch.ones(co)
        sd[f"model.{bn_i}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    sd["model.11.weight"] = _normal(g, (1, ndf * 8, 4, 4), 0.02)
    sd["model.11.bias"] = torch.zeros(1)
    return sd
