"""Option system with the reference's surface (utils/options.py + utils/util.py:378-412):

  * ``EasyDict``: attribute-and-item dict used for ``opt`` / ``var`` (nested dicts become
    EasyDicts; attributes and items are one storage).
  * ``parse_arguments(argv)``: the ``--key1.key2=value`` command-line grammar (utils/options.py:16-34):
    values are YAML scalars / lists, ``--flag`` means true, ``--flag!`` means false, a repeated key is
    an error.
  * ``load_options(fname)``: a YAML file, with ``_parent_`` inheritance (:56-69).
  * ``override_options`` (:71-89): recursive merge; with safe_check a key that does not exist in the
    file is an ERROR here (the reference asks interactively; a batch job cannot answer).
  * ``process_options`` (:91-112): seeds, ``output_path = output_root/group/name``, ``device``,
    ``H, W``, default ``freq.eval``.
  * ``set(opt_cmd)`` (:36-54): the three together, as train.py / evaluate.py / demo.py call it.

options/shape.yaml and options/depth.yaml in this repository carry the reference's option trees.
"""
import os
import random
import string

import numpy as np
import yaml


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {})
        d.update(kwargs)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, name, value):
        if isinstance(value, (list, tuple)):
            value = [self.__class__(x) if isinstance(x, dict) else x for x in value]
        elif isinstance(value, dict) and not isinstance(value, self.__class__):
            value = self.__class__(value)
        super().__setattr__(name, value)
        super().__setitem__(name, value)

    __setitem__ = __setattr__

    def update(self, e=None, **f):
        d = dict(e or {})
        d.update(f)
        for k in d:
            setattr(self, k, d[k])

    def pop(self, k, d=None):
        if hasattr(self, k):
            delattr(self, k)
        return super().pop(k, d)


edict = EasyDict


def parse_arguments(args):
    """['--eval.vox_res=128', '--eval.brute_force', '--optim.sched!'] -> nested EasyDict."""
    tree = {}
    for arg in args:
        if not arg.startswith("--"):
            raise ValueError("options are written --key.subkey=value, got %r" % arg)
        body = arg[2:]
        if "=" in body:
            path, text = body.split("=", 1)
        elif body.endswith("!"):
            path, text = body[:-1], "false"
        else:
            path, text = body, "true"
        *parents, leaf = path.split(".")
        node = tree
        for k in parents:
            node = node.setdefault(k, {})
        if leaf in node:
            raise ValueError("option %s given twice" % path)
        node[leaf] = yaml.safe_load(text)
    return EasyDict(tree)


def load_options(fname):
    with open(fname) as f:
        opt = EasyDict(yaml.safe_load(f))
    if "_parent_" in opt:
        parents = opt.pop("_parent_")
        for parent in ([parents] if isinstance(parents, str) else parents):
            opt = override_options(load_options(parent), opt, key_stack=[])
    return opt


def override_options(opt, opt_over, key_stack=None, safe_check=False):
    key_stack = key_stack or []
    for key, value in opt_over.items():
        if isinstance(value, dict):
            opt[key] = override_options(opt.get(key, EasyDict()), value, key_stack + [key], safe_check)
        else:
            if safe_check and key not in opt:
                raise KeyError("option %s is not in the yaml file (the reference prompts here; add it to the "
                               "file or pass safe_check=False)" % ".".join(key_stack + [key]))
            opt[key] = value
    return opt


def process_options(opt, need_gpu=True):
    import torch
    if opt.seed is not None:
        random.seed(opt.seed)
        np.random.seed(opt.seed)
        torch.manual_seed(opt.seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(opt.seed)
    else:
        opt.name += "_" + "".join(random.choice(string.ascii_uppercase) for _ in range(4))
    opt.output_path = "{0}/{1}/{2}".format(opt.output_root, opt.group, opt.name)
    os.makedirs(opt.output_path, exist_ok=True)
    if need_gpu and not torch.cuda.is_available():
        raise RuntimeError("zeroshape_amd needs a GPU (there is no CPU path)")
    opt.device = "cuda:{}".format(opt.gpu)
    opt.H, opt.W = opt.image_size
    if opt.freq.eval is None:
        opt.freq.eval = max(opt.max_epoch // 20, 1)
    if "loss_weight" in opt:
        opt.get_depth = False
        opt.get_normal = False
    return opt


def set(opt_cmd=None, verbose=False, safe_check=True, need_gpu=True):      # noqa: A001 (the reference's name)
    """opt_cmd.yaml names the file; everything else in opt_cmd overrides it."""
    opt_cmd = opt_cmd if opt_cmd is not None else EasyDict()
    opt = override_options(load_options(opt_cmd.yaml), opt_cmd, key_stack=[], safe_check=safe_check)
    process_options(opt, need_gpu=need_gpu)
    if verbose:
        def show(o, level=0):
            for key, value in sorted(o.items()):
                if isinstance(value, dict):
                    print("   " * level + "* " + key + ":")
                    show(value, level + 1)
                else:
                    print("   " * level + "* " + key + ":", value)
        show(opt)
    return opt


def save_options_file(opt):
    """:114-137 without the interactive diff: writes <output_path>/options.yaml."""
    def plain(o):
        return {k: plain(v) if isinstance(v, dict) else v for k, v in o.items()}
    with open("{}/options.yaml".format(opt.output_path), "w") as f:
        yaml.safe_dump(plain(opt), f, default_flow_style=False, indent=4)
