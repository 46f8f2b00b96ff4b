"""Mirror of the reference's model/depth_engine.py::Runner for evaluation: DPT depth + intrinsics
through graph_depth.Graph, DepthMetric (scale/shift aligned d>thr / rmse / l1 / abs_rel, one fused
launch per batch), per-sample metrics gathered over the ranks, best_val.txt in the reference's
format (:270-382).  Training this task needs the MiDaS loss, which is not on the HIP path: train()
raises."""
import os

import torch
import torch.distributed as dist

from ..utils import util
from ..utils.eval_depth import DepthMetric
from ..utils.options import EasyDict as edict
from .compute_graph import graph_depth


class Runner:
    def __init__(self, opt):
        self.opt = opt
        world = getattr(opt, "world_size", 1) or 1
        if world > 1:
            if "port" in opt and isinstance(opt.device, int):
                util.setup(opt.device, world, opt.port)                      # :32
            if "batch_size" in opt and not getattr(opt, "_batch_divided", False):
                opt.batch_size = opt.batch_size // world
                opt._batch_divided = True
        self.test_data = self.test_loader = None

    def load_dataset(self, opt, eval_split="test", dataset=None):
        """:46-68 (test side; datasets without 3-D annotations: load_3D=False)."""
        import importlib
        if dataset is None:
            pkg = __name__.rsplit(".", 2)[0] + ".data."
            dataset = importlib.import_module(pkg + opt.data.dataset_test).Dataset(opt, split=eval_split, load_3D=False)
        self.test_data = dataset
        sampler = None
        if getattr(opt, "world_size", 1) > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(self.test_data, shuffle=False, drop_last=False)
        self.test_loader = torch.utils.data.DataLoader(self.test_data, batch_size=opt.eval.batch_size, shuffle=False,
                                                       sampler=sampler, num_workers=0, drop_last=False)

    def build_networks(self, opt):
        self.graph = graph_depth.Graph(opt).to(opt.device).eval()
        self.depth_metric = DepthMetric(thresholds=opt.eval.d_thresholds, depth_cap=opt.eval.depth_cap)   # :75

    def setup_optimizer(self, opt):
        raise NotImplementedError("depth_engine: training the depth task needs the MiDaS loss (not on the HIP path)")

    def train(self, opt):
        raise NotImplementedError("depth_engine: training the depth task needs the MiDaS loss (not on the HIP path)")

    def restore_checkpoint(self, opt, best=False, evaluate=False):
        if getattr(opt, "load", None):
            util.restore_checkpoint(opt, self, load_name=opt.load)

    def setup_visualizer(self, opt, test=False):
        return None

    @torch.no_grad()
    def evaluate_batch(self, opt, var, ep=None, it=None, single_gpu=False):
        var = util.move_to_device(var, opt.device)
        return self.graph.forward(opt, var, training=False, get_loss=False)

    @torch.no_grad()
    def evaluate(self, opt, ep=0, training=False):
        """:270-382: returns the mean l1_err (the validation metric) and, on rank 0 outside training,
        writes <output_path>/best_val.txt with one `key: value` line per metric."""
        from .. import parallel
        self.graph.eval()
        keys = self.depth_metric.metric_keys
        rows, ids = [], []
        for it, batch in enumerate(self.test_loader):
            var = self.evaluate_batch(opt, edict(batch), ep, it)
            mask = var.mask_eroded if 'mask_eroded' in var else var.mask_input_map
            sample_metrics, var.depth_pred_aligned = self.depth_metric.compute_metrics(var.depth_pred, var.depth_input_map,
                                                                                       mask)
            rows.append(torch.stack([sample_metrics[k].float() for k in keys], 1))
            ids.append(torch.as_tensor(var.idx).view(-1).to(opt.device))
        ids, (rows,) = parallel.gather_sample_rows(torch.cat(ids).long(), [torch.cat(rows)])
        assert rows.shape[0] == len(self.test_data)
        metric_avg = {k: rows[:, i].mean().item() for i, k in enumerate(keys)}
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if rank == 0:
            util.print_eval(opt, depth_metrics=metric_avg)
            if not training and getattr(opt, "output_path", None):
                os.makedirs(opt.output_path, exist_ok=True)
                with open(os.path.join(opt.output_path, 'best_val.txt'), "w") as outfile:
                    for k in keys:
                        outfile.write('{}: {:.6f}\n'.format(k, metric_avg[k]))
            self.last_metrics = metric_avg
            return metric_avg['l1_err']
        return float('inf')
