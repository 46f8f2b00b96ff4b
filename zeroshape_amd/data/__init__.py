"""Datasets with the reference's sample-dict keys (data/synthetic.py:126-176).

Every module here is an ANALYTIC STAND-IN registered under the name of one of the reference's loaders
(zeroshape_amd/compat.py): posed ellipsoids rendered in closed form, not Objaverse / Pix3D / OmniObject3D.  They
exist so the engines and entry scripts run on a box without the (Dropbox-hosted) data; their scores are not
benchmark numbers.  ``load_by_name`` - the engines' way in - therefore refuses to hand one out unless the caller
opted in (``ZS_SYNTHETIC_STANDIN=1`` in the environment or ``opt.data.synthetic_standin``), warns when it does,
and the Dataset objects carry ``synthetic_standin``, the line the engines put at the top of every result file."""
import importlib
import os
import sys

_REAL_DIRS = {"synthetic": "data/train_data", "pix3d": "data/Pix3D", "omniobj3d": "data/OmniObject3D"}


def standin_opted_in(opt):
    data = opt.get("data", {}) if hasattr(opt, "get") else {}
    flag = data.get("synthetic_standin", False) if hasattr(data, "get") else False
    return bool(flag) or os.environ.get("ZS_SYNTHETIC_STANDIN", "0") not in ("", "0")


def load_by_name(opt, name, **kwargs):
    """data.<name>.Dataset(opt, **kwargs) as the reference's engines build it (model/shape_engine.py:52-81), with
    the stand-in fence described above.  Unknown names fail with ModuleNotFoundError like the reference."""
    module = importlib.import_module(__name__ + "." + name)
    if getattr(module, "SYNTHETIC_STANDIN", False):
        real = _REAL_DIRS.get(name)
        found = " (a real %s directory exists and would be IGNORED)" % real if real and os.path.isdir(real) else ""
        if not standin_opted_in(opt):
            raise RuntimeError(
                "data.%s is an analytic stand-in (posed ellipsoids), not the reference's %s loader: reading the real "
                "files is not built%s.  Set ZS_SYNTHETIC_STANDIN=1 (or opt.data.synthetic_standin) to run the engine "
                "on the stand-in on purpose; result files are then tagged as synthetic." % (name, name, found))
        print("WARNING: data.%s is a SYNTHETIC STAND-IN (analytic ellipsoids), not %s%s; scores are not benchmark "
              "numbers" % (name, name, found), file=sys.stderr, flush=True)
    return module.Dataset(opt, **kwargs)
