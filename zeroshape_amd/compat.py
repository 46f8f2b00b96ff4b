"""Run the reference's entry scripts (train.py / evaluate.py, which `import utils.options`,
`from utils.util import is_port_in_use` and `importlib.import_module('model.shape_engine')`)
against this package without editing them: install() aliases the reference's top-level package
names (utils, model, data, external) to their zeroshape_amd mirrors in sys.modules.

    python -c "import zeroshape_amd.compat as c; c.install(); import runpy; runpy.run_path('train.py', run_name='__main__')" \\
        --yaml=options/shape.yaml
or simply `python train.py ...` with the train.py / evaluate.py of this repository, which do that.

Only names that exist here are aliased; importing anything else of the reference (data.ocrtoc,
...) fails loudly with ModuleNotFoundError instead of silently falling back.
"""
import importlib
import sys

_ALIASES = {
    "utils": "zeroshape_amd.utils",
    "utils.options": "zeroshape_amd.utils.options",
    "utils.util": "zeroshape_amd.utils.util",
    "utils.camera": "zeroshape_amd.utils.camera",
    "utils.eval_3D": "zeroshape_amd.utils.eval_3D",
    "utils.eval_depth": "zeroshape_amd.utils.eval_depth",
    "utils.loss": "zeroshape_amd.utils.loss",
    "utils.layers": "zeroshape_amd.utils.layers",
    "utils.pos_embed": "zeroshape_amd.utils.pos_embed",
    "utils.util_vis": "zeroshape_amd.utils.util_vis",
    "model": "zeroshape_amd.model",
    "model.shape_engine": "zeroshape_amd.model.shape_engine",
    "model.depth_engine": "zeroshape_amd.model.depth_engine",
    "model.compute_graph": "zeroshape_amd.model.compute_graph",
    "model.compute_graph.graph_shape": "zeroshape_amd.model.compute_graph.graph_shape",
    "model.compute_graph.graph_depth": "zeroshape_amd.model.compute_graph.graph_depth",
    "model.depth": "zeroshape_amd.model.depth",
    "model.depth.dpt_depth": "zeroshape_amd.model.depth.dpt_depth",
    "model.shape": "zeroshape_amd.model.shape",
    "model.shape.implicit": "zeroshape_amd.model.shape.implicit",
    "model.shape.seen_coord_enc": "zeroshape_amd.model.shape.seen_coord_enc",
    "data": "zeroshape_amd.data",
    "data.synthetic": "zeroshape_amd.data.synthetic",
    "data.pix3d": "zeroshape_amd.data.pix3d",
    "data.omniobj3d": "zeroshape_amd.data.omniobj3d",
    "external": "zeroshape_amd.external",
    "external.chamfer3D": "zeroshape_amd.external.chamfer3D",
    "external.chamfer3D.dist_chamfer_3D": "zeroshape_amd.external.chamfer3D.dist_chamfer_3D",
    "chamfer_3D": "zeroshape_amd.chamfer_3D",
}


def install():
    for alias, target in _ALIASES.items():
        if alias in sys.modules and sys.modules[alias].__name__ != target:
            raise ImportError("cannot alias %r to %s: a different module of that name is already imported (%s)"
                              % (alias, target, getattr(sys.modules[alias], "__file__", "?")))
        sys.modules[alias] = importlib.import_module(target)
    return sorted(_ALIASES)
