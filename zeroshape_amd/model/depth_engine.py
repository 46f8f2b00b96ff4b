"""Mirror of the reference's model/depth_engine.py::Runner: DPT depth + intrinsics through
graph_depth.Graph.  Training (setup_optimizer :77-101: two AdamW groups; train / train_epoch /
train_iteration :124-236) shares the loop of the shape engine - HIP autograd forward / backward, MiDaS
depth loss + intrinsics loss, bucketed RCCL gradient averaging, fused AdamW.  Evaluation (:270-382):
DepthMetric (scale/shift aligned d>thr / rmse / l1 / abs_rel, one fused launch per batch), per-sample
metrics gathered over the ranks, best_val.txt in the reference's format."""
import os

import torch
import torch.distributed as dist

from ..utils import util
from ..utils.eval_depth import DepthMetric
from ..utils.options import EasyDict as edict
from . import shape_engine
from .compute_graph import graph_depth


class Runner(shape_engine.Runner):
    """Inherits the process setup, the train loop (train / train_epoch / train_iteration /
    summarize_loss / checkpoints) and restore_checkpoint of the shape engine; differs in the graph,
    the data sets (no 3-D annotations), the optimiser groups and the evaluation."""

    def load_dataset(self, opt, eval_split="test", dataset=None, train_dataset=None):
        """:46-68 (datasets without 3-D annotations: load_3D=False)."""
        from ..data import load_by_name
        if dataset is None:
            dataset = load_by_name(opt, opt.data.dataset_test, split=eval_split, load_3D=False)
        self.test_data = dataset
        sampler = None
        if getattr(opt, "world_size", 1) > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(self.test_data, shuffle=False, drop_last=False)
        self.test_loader = torch.utils.data.DataLoader(self.test_data, batch_size=opt.eval.batch_size, shuffle=False,
                                                       sampler=sampler, num_workers=0, drop_last=False)
        if train_dataset is not None or ("batch_size" in opt and "dataset_train" in opt.data and "optim" in opt):
            if train_dataset is None:
                train_dataset = load_by_name(opt, opt.data.dataset_train, split="train", load_3D=False)
            self.load_train_dataset(opt, dataset=train_dataset)

    def build_networks(self, opt):
        self.graph = graph_depth.Graph(opt).to(opt.device).eval()
        self.depth_metric = DepthMetric(thresholds=opt.eval.d_thresholds, depth_cap=opt.eval.depth_cap)   # :73

    def setup_optimizer(self, opt):
        """:77-101: biases and 1-d tensors without weight decay, everything at optim.lr."""
        from .. import parallel
        from ..optim import FusedAdamW
        nodecay, decay = [], []
        for name, param in self.graph.named_parameters():
            if not param.requires_grad:
                continue
            (nodecay if (param.ndim <= 1 or name.endswith(".bias")) else decay).append(param)
        self.optim = FusedAdamW([{'params': nodecay, 'lr': opt.optim.lr, 'weight_decay': 0.},
                                 {'params': decay, 'lr': opt.optim.lr, 'weight_decay': opt.optim.weight_decay}],
                                betas=(0.9, 0.95))
        if opt.optim.sched:
            self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.optim, opt.max_epoch)
        # optim.amp (model/shape_engine.py:135-136, :252-269: fp16 autocast + GradScaler): here the forward
        # AND data-gradient convolutions / linear layers move to the 16-bit matrix pipe with split-fp16 operands
        # (~2^-21 relative instead of fp16's 2^-11, fp32 accumulation); weight gradients and everything else stay
        # fp32.  The data gradients need the loss scaled into fp16's range: optim.LossScaler, GradScaler's rules
        # with the scalars on the device
        from ..nn import autograd as A
        from ..optim import LossScaler
        A.set_forward_precision("f16x3" if opt.optim.amp else os.environ.get("ZS_TRAIN_FWD_PRECISION", "f32"))
        A.set_backward_precision("f16x3" if opt.optim.amp else "f32")
        self.__dict__.pop("scaler", None)          # (checkpoints carry every attribute named scaler*, like the reference's)
        if opt.optim.amp:
            self.scaler = LossScaler(opt.device)
        if getattr(opt, "world_size", 1) > 1:
            self.reducer = parallel.GradReducer(self.graph.parameters(), module=self.graph,
                                                bucket_mb=getattr(opt.optim, "bucket_mb", 64.0))

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
            return dict(metric_avg, cd=metric_avg['l1_err']) if training else metric_avg['l1_err']
        return dict(cd=float('inf')) if training else float('inf')
