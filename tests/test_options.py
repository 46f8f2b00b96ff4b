"""The option surface: options/*.yaml carry the reference's option trees,
the --a.b=v grammar and the merge rules of utils/options.py behave like the reference's."""
import os

import pytest

from zeroshape_amd.utils import options

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_command_line_grammar():
    o = options.parse_arguments(["--yaml=options/shape.yaml", "--eval.vox_res=128", "--eval.brute_force", "--optim.sched!",
                                 "--data.dataset_test=pix3d", "--eval.range=[-1,1]", "--optim.lr=1.e-4", "--name=a=b"])
    assert o.yaml == "options/shape.yaml" and o.eval.vox_res == 128 and o.eval.brute_force is True
    assert o.optim.sched is False and o.data.dataset_test == "pix3d" and o.eval.range == [-1, 1]
    assert o.optim.lr == 1e-4 and o.name == "a=b"
    with pytest.raises(ValueError):
        options.parse_arguments(["--eval.vox_res=1", "--eval.vox_res=2"])
    with pytest.raises(ValueError):
        options.parse_arguments(["eval.vox_res=1"])


def test_shape_yaml_defaults_and_overrides(tmp_path):
    cmd = options.parse_arguments(["--yaml=%s/options/shape.yaml" % ROOT, "--eval.vox_res=128", "--eval.brute_force",
                                   "--eval.batch_size=1", "--output_root=%s" % tmp_path])
    opt = options.set(cmd, need_gpu=False)
    assert (opt.eval.vox_res, opt.eval.brute_force, opt.eval.batch_size) == (128, True, 1)      # README.md:108
    assert opt.batch_size == 28 and opt.arch.impl.skip_in == [2, 4, 6] and opt.arch.depth.encoder == "resnet"
    assert opt.optim.lr == 3e-5 and opt.optim.lr_ft == 1e-5 and opt.optim.fix_dpt is False and opt.optim.amp is False
    assert opt.loss_weight.shape == 1 and opt.loss_weight.depth is None and opt.training.n_sdf_points == 4096
    assert opt.eval.range == [-1.5, 1.5] and opt.eval.f_thresholds[0] == 0.005
    assert (opt.H, opt.W) == (224, 224) and opt.device == "cuda:0"
    assert opt.output_path == "%s/shape/shape_recon" % tmp_path and os.path.isdir(opt.output_path)
    with pytest.raises(KeyError):       # the reference prompts for unknown keys; a batch job errors
        options.set(options.parse_arguments(["--yaml=%s/options/shape.yaml" % ROOT, "--eval.vox_rez=128"]), need_gpu=False)
    options.save_options_file(opt)
    assert options.load_options(opt.output_path + "/options.yaml").eval.vox_res == 128


def test_parent_inheritance(tmp_path):
    child = tmp_path / "child.yaml"
    child.write_text("_parent_: %s/options/depth.yaml\nname: mine\noptim: {lr: 1.e-3}\n" % ROOT)
    opt = options.load_options(str(child))
    assert opt.name == "mine" and opt.optim.lr == 1e-3 and opt.optim.weight_decay == 0.05
    assert opt.loss_weight.intr == 10 and opt.eval.d_thresholds == [1.02, 1.05, 1.1, 1.2]
