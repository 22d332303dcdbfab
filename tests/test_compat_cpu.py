"""The drop-in boundary on the host (no GPU): after `compat.install()` the reference's import block
(running_files/train_end2end_jsrt.py:22-31) resolves the hot-path names to this package while the caller's own
`util` / `options` Python keeps working (options/base_options.py:3-5,65-71).  Each case runs in a fresh interpreter so
that sys.modules of the test process stays clean."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


def run_py(code, cwd, extra_path=()):
    env = dict(os.environ)
    env["PYTHONDONTWRITEBYTECODE"] = "1"          # never write __pycache__ into the (read-only) reference tree
    env["PYTHONPATH"] = os.pathsep.join([ROOT, *extra_path, env.get("PYTHONPATH", "")])
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=cwd, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def make_standin_tree(tmp_path):
    """A caller's tree with its own `util` package and an `options` package shaped like the reference's two-pass parser
    (our own few lines, not the reference's files)."""
    (tmp_path / "util").mkdir()
    (tmp_path / "util" / "__init__.py").write_text('"""stand-in util package"""\n')
    (tmp_path / "util" / "util.py").write_text("import os\n\nMARK = 'standin-util'\n\n\ndef mkdirs(p):\n    os.makedirs(p, exist_ok=True)\n")
    (tmp_path / "util" / "JSRT_loader.py").write_text("class CarvanaDataset:\n    MARK = 'standin-loader'\n")
    (tmp_path / "util" / "dice_score.py").write_text("def dice_loss(*a, **k):\n    raise AssertionError('caller dice_score must be replaced')\n")
    (tmp_path / "options").mkdir()
    (tmp_path / "options" / "__init__.py").write_text("")
    (tmp_path / "options" / "train_options.py").write_text(textwrap.dedent('''
        import argparse
        from util import util
        import models_pix2pix


        class TrainOptions:
            isTrain = True

            def parse(self, argv):
                p = argparse.ArgumentParser()
                p.add_argument("--model", default="cycle_gan")
                p.add_argument("--norm", default="instance")
                p.add_argument("--netG", default="resnet_9blocks")
                p.add_argument("--dataset_mode", default="unaligned")
                p.add_argument("--pool_size", type=int, default=50)
                p.add_argument("--gan_mode", default="lsgan")
                opt, _ = p.parse_known_args(argv)
                p = models_pix2pix.get_option_setter(opt.model)(p, self.isTrain)
                return p.parse_args(argv)
    '''))


def test_install_keeps_callers_util_and_options(tmp_path):
    make_standin_tree(tmp_path)
    out = run_py('''
        import json, sys
        sys.path.insert(0, '.')
        from semantic_segmentation_amd import compat
        compat.install()
        from util import util
        from util.JSRT_loader import CarvanaDataset
        from util.dice_score import dice_loss
        import util.dice_score as ds
        from options.train_options import TrainOptions
        from models_pix2pix import create_model, networks
        from unet import UNet
        from unet.evaluate import evaluate
        opt = TrainOptions().parse(["--model", "pix2pix"])
        print(json.dumps({
            "util": util.MARK, "loader": CarvanaDataset.MARK, "dice": dice_loss.__module__, "dice_mod": ds.__name__,
            "unet": UNet.__module__, "evaluate": evaluate.__module__, "networks": networks.__name__,
            "create_model": create_model.__module__,
            "opt": [opt.netG, opt.norm, opt.gan_mode, opt.lambda_L1, opt.pool_size, opt.dataset_mode]}))
    ''', cwd=str(tmp_path))
    assert out["util"] == "standin-util" and out["loader"] == "standin-loader"
    assert out["dice"] == "semantic_segmentation_amd.util.dice_score" == out["dice_mod"]
    assert out["unet"].startswith("semantic_segmentation_amd.unet")
    assert out["evaluate"].startswith("semantic_segmentation_amd.unet")
    assert out["networks"] == "semantic_segmentation_amd.models_pix2pix.networks"
    assert out["create_model"] == "semantic_segmentation_amd.models_pix2pix"
    assert out["opt"] == ["unet_256", "batch", "vanilla", 100.0, 0, "aligned"]


def test_install_after_callers_util_was_imported(tmp_path):
    """force=True re-binds util.dice_score even when the caller imported its own util (and dice_score) first."""
    make_standin_tree(tmp_path)
    out = run_py('''
        import json, sys
        sys.path.insert(0, '.')
        import util.dice_score, util.util
        from semantic_segmentation_amd import compat
        compat.install()
        import util
        from util.dice_score import dice_loss
        compat.install()                        # idempotent
        from util import util as uu
        print(json.dumps({"dice": dice_loss.__module__, "attr": util.dice_score.__name__, "util": uu.MARK}))
    ''', cwd=str(tmp_path))
    assert out == {"dice": "semantic_segmentation_amd.util.dice_score", "attr": "semantic_segmentation_amd.util.dice_score",
                   "util": "standin-util"}


def test_install_without_any_util_package(tmp_path):
    out = run_py('''
        import json
        from semantic_segmentation_amd import compat
        compat.install()
        from util.dice_score import dice_loss, dice_coeff, multiclass_dice_coeff
        print(json.dumps({"dice": dice_loss.__module__}))
    ''', cwd=str(tmp_path))
    assert out["dice"] == "semantic_segmentation_amd.util.dice_score"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "options")), reason="reference checkout only exists in the build container")
def test_reference_import_block_and_trainoptions_parse(tmp_path):
    """Build container only: the reference's own `options.train_options.TrainOptions().parse()` and its `util.util`
    after install(), run from the reference root as scripts/train_end2end_jsrt.sh does (result recorded in INTEGRATION.md)."""
    out = run_py('''
        import json, sys
        sys.path.append('.')                    # train_end2end_jsrt.py:3
        sys.argv = ["train_end2end_jsrt.py", "--dataroot", "x", "--model", "pix2pix", "--gpu_ids", "-1",
                    "--checkpoints_dir", %r]
        from semantic_segmentation_amd import compat
        compat.install()
        from util import util
        from util.dice_score import dice_loss
        from options.train_options import TrainOptions
        from models_pix2pix import create_model, networks
        from unet import UNet
        from unet.evaluate import evaluate
        opt = TrainOptions().parse()
        print(json.dumps({"util": util.__file__, "dice": dice_loss.__module__, "unet": UNet.__module__,
                          "opt": [opt.netG, opt.norm, opt.gan_mode, opt.lambda_L1, opt.pool_size, opt.isTrain]}))
    ''' % str(tmp_path), cwd=REFERENCE)
    assert out["util"] == os.path.join(REFERENCE, "util", "util.py")
    assert out["dice"] == "semantic_segmentation_amd.util.dice_score"
    assert out["unet"].startswith("semantic_segmentation_amd.")
    assert out["opt"] == ["unet_256", "batch", "vanilla", 100.0, 0, True]
