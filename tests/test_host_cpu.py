"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/gsseg.h declares,
the Python surface keeps the reference's names / state-dict keys, host-side geometry is right, and the
product package never imports the oracle."""
import ast
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from semantic_segmentation_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from semantic_segmentation_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gsseg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gsseg.h but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.gs_abi_version() == _lib.ABI_VERSION


def test_argument_validation_is_synchronous(lib):
    from semantic_segmentation_amd import _lib
    g = _lib.GsConvGeom()
    assert lib.gs_conv_igemm(g, None, None, None, None, None, 0, 0, None, 0, None) != 0
    assert b"gs_conv_igemm" in lib.gs_last_error()
    assert lib.gs_bn_finalize(None, 0, 0, 0.0, None, None, None, None, 0.1, 1e-5, None, None, None, None, None) != 0
    assert lib.gs_bn_partials_floats(10, 64) >= 10 * 2 * 64


def test_unet_api_surface_and_state_dict_keys():
    from oracle import oracle
    from semantic_segmentation_amd.unet import UNet
    for n_classes in (1, 2):
        net = UNet(1, n_classes)
        assert (net.n_channels, net.n_classes, net.bilinear) == (1, n_classes, False)
        assert [n for n, _ in net.named_children()] == ["inc", "down1", "down2", "down3", "down4", "up1", "up2",
                                                        "up3", "up4", "outc"]
        sd = oracle.unet_state_dict(1, n_classes, seed=0)
        assert list(net.state_dict().keys()) == list(sd.keys())
        assert len(sd) == 118
        for k, v in net.state_dict().items():
            assert tuple(v.shape) == tuple(sd[k].shape), k
        assert sum(p.numel() for p in net.parameters()) == (31036481 if n_classes == 1 else 31036546)
        net.load_state_dict(sd, strict=True)
    with pytest.raises(RuntimeError):
        UNet(1, 1)(torch.zeros(1, 1, 32, 32))       # CPU tensor: no fallback


def test_geometry_builders():
    from semantic_segmentation_amd import ops
    g = ops.geom_conv(2, 17, 9, 64, 128, 3, 1, 1)
    assert (g.OHg, g.OWg, g.OH, g.OW, g.ntaps) == (17, 9, 17, 9, 9)
    assert (g.tap_dy[0], g.tap_dx[0], g.tap_dy[8], g.tap_dx[8]) == (-1, -1, 1, 1)
    g = ops.geom_conv(1, 256, 256, 64, 128, 4, 2, 1)
    assert (g.OH, g.OW, g.isy, g.ntaps) == (128, 128, 2, 16)
    d = ops.geom_conv_dgrad_s1(2, 17, 9, 64, 128, 3, 1)
    assert (d.Cin, d.Cout, d.tap_dy[0], d.tap_dy[8]) == (128, 64, 1, -1)


def test_igemm_stat_rows_follow_the_kernel_form():
    """gs_conv_igemm_mtiles is the host-side contract for the BatchNorm partial rows of a gs_conv_igemm launch: one row per 128-row
    tile on the register-staged engine, one per 16 output rows for the geometries the weight-streaming form covers (<= 8192
    output pixels, Cin % 64 == 0, Cout % 32 == 0, 2-D, a K of at least four 64-channel chunks)."""
    from semantic_segmentation_amd import ops
    cases = [((2, 16, 16, 512, 512, 4, 2, 1), True),      # M = 128
             ((2, 2, 2, 512, 512, 4, 2, 1), True),        # M = 2: ceil(2 / 16) = 1 row
             ((2, 128, 128, 64, 128, 4, 2, 1), True),     # M = 8192
             ((32, 256, 256, 64, 64, 3, 1, 1), False),    # 2 M pixels
             ((2, 16, 16, 512, 72, 4, 2, 1), False),      # Cout % 32
             ((2, 16, 16, 40, 128, 4, 2, 1), False),      # Cin % 64
             ((2, 16, 16, 64, 128, 1, 1, 0), False)]      # one tap x one chunk: nothing to pipeline
    for args, streaming in cases:
        g = ops.geom_conv(*args)
        M = g.N * g.OHg * g.OWg
        assert ops.conv_igemm_mtiles(g) == ((M + 15) // 16 if streaming else (M + 127) // 128), args
    if os.environ.get("GSSEG_SKINNY", "1") != "0":
        g = ops.geom_convT_class(2, 1, 1, 512, 512, 8, 3, 0, 0)      # 1x1 input: one of the class's 16 taps is visible -> 8 chunks
        assert ops.conv_igemm_mtiles(g) == 1


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "semantic_segmentation_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dp, f)).read())
            for node in ast.walk(tree):
                mods = []
                if isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom) and node.module:
                    mods = [node.module]
                for m in mods:
                    assert not m.split(".")[0] == "oracle", f"{f} imports {m}"


def test_pix2pix_api_surface_and_state_dict_keys():
    from golden_util import seeded_discriminator_state_dict, seeded_generator_state_dict
    from semantic_segmentation_amd.models_pix2pix import networks
    norm = networks.get_norm_layer("batch")
    G = networks.define_G(1, 1, 64, "unet_256", "batch", True)
    D = networks.define_D(2, 64, "basic", 3, "batch")
    sdG, sdD = seeded_generator_state_dict(1), seeded_discriminator_state_dict(2)
    # golden_util's key SETS were validated against the reference with load_state_dict(strict=True)
    assert sorted(G.state_dict().keys()) == sorted(sdG.keys()) and len(sdG) == 100
    assert sorted(D.state_dict().keys()) == sorted(sdD.keys()) and len(sdD) == 22
    for k, v in G.state_dict().items():
        assert tuple(v.shape) == tuple(sdG[k].shape), k
    assert sum(p.numel() for p in G.parameters()) == 272327811
    assert sum(p.numel() for p in D.parameters()) == 2764609
    assert len(networks.arch_parameters()) == 2 and tuple(networks.upconv_arch.shape) == (8, 3)
    assert networks.GANLoss("vanilla").gan_mode == "vanilla"
    with pytest.raises(RuntimeError):
        D(torch.zeros(1, 2, 64, 64))                  # CPU tensor: no fallback


def test_unet3d_api_surface_and_state_dict_keys():
    from oracle import oracle
    from semantic_segmentation_amd.unet3d import UNet3D
    net = UNet3D(1, 2)
    sd = oracle.unet3d_state_dict(1, 2, seed=0)      # key set validated against the reference (strict load) in make_golden
    assert sorted(net.state_dict().keys()) == sorted(sd.keys())
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    assert sum(p.numel() for p in net.parameters()) == 19072834
    net.load_state_dict(sd, strict=True)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 1, 8, 8, 8))
