"""Mirror of the reference's model/shape_engine.py::Runner: build the graph, restore a checkpoint,
TRAIN it (setup_optimizer :85-136, train :186-199, train_epoch :201-246, train_iteration :248-297)
and evaluate it (:335-523: test loader -> Graph.forward + eval_metrics -> per-sample metrics
gathered over the ranks -> the reference's result files).

Training runs one process per GPU: Graph.forward(training=True) on the HIP autograd path, gradients
averaged over the ranks by parallel.GradReducer (bucketed RCCL all-reduce under the backward pass -
the role torch DDP has in the reference), one fused AdamW launch (zeroshape_amd/optim.py) with the
reference's four parameter groups.  Not rebuilt (control plane): tensorboard
scalars, visual dumps.  optim.amp = split-fp16 forward and data-gradient GEMMs under a dynamic loss scale
(see setup_optimizer)."""
import os

import numpy as np
import torch
import torch.distributed as dist

from ..nn import autograd as A
from ..utils import eval_3D, util
from ..utils.options import EasyDict as edict
from .compute_graph.graph_shape import Graph


class Runner:
    def __init__(self, opt):
        self.opt = opt
        self.test_data = self.test_loader = None
        self.train_data = self.train_loader = None
        self.reducer = None
        self.ep = self.it = 0
        self.best_val, self.best_ep = float("inf"), 0
        self.epoch_start = self.iter_start = 0
        world = getattr(opt, "world_size", 1) or 1
        if world > 1:
            if "port" in opt and isinstance(opt.device, int):             # launched like train.py:14-16 does
                util.setup(opt.device, world, opt.port)                    # :32
                opt.device = int(os.environ.get("ZS_DEVICE_OVERRIDE", opt.device))   # (one-GPU rehearsal of N ranks)
            if "batch_size" in opt and not getattr(opt, "_batch_divided", False):
                opt.batch_size = opt.batch_size // world                  # :33
                opt._batch_divided = True

    def load_dataset(self, opt, eval_split="test", dataset=None, train_dataset=None):
        """:52-81.  Datasets are the modules data.<opt.data.dataset_train|dataset_test> (importlib,
        like the reference); this repository ships analytic stand-ins under those names (the real files live on
        Dropbox), handed out only on an explicit opt-in and tagged in every result file (data/__init__.py).
        `dataset` / `train_dataset` inject Dataset objects directly.
        The train side is loaded when the options describe a training run (batch_size present)."""
        from ..data import load_by_name
        if dataset is None:
            dataset = load_by_name(opt, opt.data.dataset_test, split=eval_split)
        self.test_data = dataset
        sampler = None
        world = getattr(opt, "world_size", 1) or 1
        # eval.shard_image (not in the reference; SURVEY.md section 8e, BASELINE config 5): the ranks share every test
        # image - point ranges of its grid, rotations of its pose search - instead of taking different images.
        # Unset = auto: on when there are more ranks than test images (a DistributedSampler would pad the shards
        # with duplicates and most GPUs would repeat each other's work).
        shard = opt.eval.get("shard_image", None) if hasattr(opt.eval, "get") else getattr(opt.eval, "shard_image", None)
        opt.eval.shard_image = world > 1 and (world > len(self.test_data) if shard is None else bool(shard))
        if world > 1 and not opt.eval.shard_image:
            sampler = torch.utils.data.distributed.DistributedSampler(self.test_data, shuffle=False, drop_last=False)
        self.test_loader = torch.utils.data.DataLoader(self.test_data, batch_size=opt.eval.batch_size, shuffle=False,
                                                       sampler=sampler, num_workers=0, drop_last=False)
        if train_dataset is not None or ("batch_size" in opt and "dataset_train" in opt.data and "optim" in opt):
            if train_dataset is None:
                train_dataset = load_by_name(opt, opt.data.dataset_train, split="train")
            self.load_train_dataset(opt, dataset=train_dataset)

    def setup_visualizer(self, opt, test=False):
        """:153-165: tensorboard / html dumps are not rebuilt (control plane); kept so the reference's
        train.py / evaluate.py call sequence runs."""
        return None

    def load_train_dataset(self, opt, dataset=None):
        """:52-58 (train side): shuffled, drop_last, DistributedSampler when world_size > 1."""
        from ..data import synthetic
        self.train_data = dataset if dataset is not None else synthetic.Dataset(opt, split="train")
        if opt.batch_size < 1:
            raise ValueError("batch_size %d per process: the global batch must be >= the world size" % opt.batch_size)
        sampler = None
        if getattr(opt, "world_size", 1) > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(self.train_data, shuffle=True, drop_last=True)
        self.train_loader = torch.utils.data.DataLoader(self.train_data, batch_size=opt.batch_size,
                                                        shuffle=sampler is None, sampler=sampler, num_workers=0,
                                                        drop_last=True)
        self.num_batches = len(self.train_loader)

    def build_networks(self, opt):
        self.graph = Graph(opt).to(opt.device).eval()

    # =============================================== training ===============================================
    def setup_optimizer(self, opt):
        """:85-136: AdamW(betas 0.9/0.95); biases and 1-d tensors without weight decay; with a
        trainable depth model its parameters (dpt_depth.*, intr_*) train at optim.lr_ft."""
        from .. import parallel
        from ..optim import FusedAdamW
        ft_nd, ft_d, sc_nd, sc_d = [], [], [], []
        for name, param in self.graph.named_parameters():
            if not param.requires_grad:
                continue
            nodecay = param.ndim <= 1 or name.endswith(".bias")
            if 'dpt_depth' in name or 'intr_' in name:
                if opt.optim.fix_dpt:
                    continue
                (ft_nd if nodecay else ft_d).append(param)
            else:
                (sc_nd if nodecay else sc_d).append(param)
        wd = opt.optim.weight_decay
        if opt.optim.fix_dpt:
            groups = [{'params': sc_nd, 'lr': opt.optim.lr, 'weight_decay': 0.},
                      {'params': sc_d, 'lr': opt.optim.lr, 'weight_decay': wd}]
        else:
            groups = [{'params': ft_nd, 'lr': opt.optim.lr_ft, 'weight_decay': 0.},
                      {'params': ft_d, 'lr': opt.optim.lr_ft, 'weight_decay': wd},
                      {'params': sc_nd, 'lr': opt.optim.lr, 'weight_decay': 0.},
                      {'params': sc_d, 'lr': opt.optim.lr, 'weight_decay': wd}]
        self.optim = FusedAdamW(groups, betas=(0.9, 0.95))
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

    def train(self, opt):
        """:164-190: resume skips the part of the first epoch that was already trained, a fresh run
        validates once before the first step, and the final state goes to checkpoint/ep<N>.ckpt."""
        self.ep = self.epoch_start
        self.it = self.iter_start
        self.iter_skip = self.iter_start % len(self.train_loader)                       # :171
        if not getattr(opt, "resume", False):                                            # :174-176
            self.best_val, self.best_ep = np.inf, 1
        if self.iter_start == 0 and not getattr(opt, "debug", False) and self.test_loader is not None:   # :178
            self.evaluate(opt, ep=0, training=True)
        self.graph.train()
        for self.ep in range(self.epoch_start, opt.max_epoch):
            self.train_epoch(opt)
        if getattr(opt, "output_path", None) and self._rank() == 0:                      # :182
            self.save_checkpoint(opt, ep=self.ep, it=self.it, best_val=self.best_val, best_ep=self.best_ep)

    def train_epoch(self, opt):
        """:192-246."""
        if isinstance(self.train_loader.sampler, torch.utils.data.distributed.DistributedSampler):
            self.train_loader.sampler.set_epoch(self.ep)
        self.graph.train()
        loader = iter(self.train_loader)
        for _ in range(len(self.train_loader)):
            if getattr(self, "iter_skip", 0) > 0:      # :223-226: slots of this epoch trained before the restart
                self.iter_skip -= 1
                continue
            var = edict(next(loader))
            opt.H, opt.W = opt.image_size
            var = util.move_to_device(var, opt.device)
            self.train_iteration(opt, var)
        if opt.optim.sched:
            self.sched.step()
        if self.test_loader is not None and (self.ep + 1) % opt.freq.eval == 0:
            current_val = self.evaluate(opt, ep=self.ep + 1, training=True)["cd"]
            self.graph.train()
            if current_val < self.best_val and self._rank() == 0:
                self.best_val, self.best_ep = current_val, self.ep + 1
                if getattr(opt, "output_path", None):
                    self.save_checkpoint(opt, ep=self.ep, it=self.it, best_val=self.best_val, best_ep=self.best_ep,
                                         best=True, latest=True)

    def summarize_loss(self, opt, var, loss, non_act_loss_key=[]):
        """:323-333."""
        loss_all = 0.
        assert "all" not in loss
        for key in loss:
            assert key in opt.loss_weight
            if opt.loss_weight[key] is not None:
                loss_all = loss_all + float(opt.loss_weight[key]) * loss[key].mean()
        loss.update(all=loss_all)
        return loss

    def train_iteration(self, opt, var, batch_progress=None):
        """:248-297 without its per-iteration barrier (the all-reduce already orders the ranks) and
        without the tensorboard / visualiser calls."""
        if self._step_capture_enabled(opt):
            return self._train_iteration_captured(opt, var)
        return self._train_iteration_eager(opt, var)

    def _train_iteration_eager(self, opt, var):
        var, loss = self.graph.forward(opt, var, training=True, get_loss=True)
        loss = self.summarize_loss(opt, var, loss)
        loss_scaled = loss.all / opt.optim.accum
        scaler = getattr(self, "scaler", None)
        if scaler is not None:
            loss_scaled = scaler.scale_loss(loss_scaled)
        if self.reducer is not None:          # only the last micro-step of an accumulation window is reduced
            self.reducer.armed = (self.it + 1) % opt.optim.accum == 0
        # (weight gradients on a side stream only where nothing reads a gradient before the join: no bucket hooks, no accumulation)
        with A.side_wgrads(self._side_wgrads(opt) and self.reducer is None and opt.optim.accum == 1):
            loss_scaled.backward()
        if (self.it + 1) % opt.optim.accum == 0:
            if self.reducer is not None:
                self.reducer.finish()
            self._optimizer_step(opt)
            self.optim.zero_grad()
        # :283-284: latest.ckpt every freq.ckpt_latest iterations (rank 0), so a crash loses at most that
        # many (the reference also writes one at iteration 0, i.e. the initial weights: skipped)
        if self._rank() == 0 and getattr(opt, "output_path", None) and not getattr(opt, "debug", False) \
                and self.it > 0 and self.it % opt.freq.ckpt_latest == 0:
            self.save_checkpoint(opt, ep=self.ep, it=self.it, best_val=self.best_val, best_ep=self.best_ep, latest=True)
        self.it += 1
        return loss

    # ---- the step as one hipGraph --------------------------------------------------------------------------- #
    # Forward + loss + backward are ~1500 launches whose Python / ctypes / allocator enqueue cost (31-37 ms at 4
    # images per GPU) is as long as their GPU time.  With `optim.hip_graph` (or ZS_TRAIN_HIP_GRAPH=1) the launch
    # sequence is stream-captured once per input signature and replayed: the batch is copied into the capture's
    # static inputs, the gradients land in the capture's static .grad tensors, and gradient clipping + the one-launch
    # AdamW run eagerly behind it (their scalars - step count, learning rate - change every step).  Everything the
    # sequence reads besides the inputs is addressed in place (parameters, buffers, packed operands, workspaces);
    # DropPath draws come from torch's graph-safe device generator.  Not captured: gradient accumulation windows (the eager
    # path handles them).  With a multi-process GradReducer the step is captured as one hipGraph per backward SEGMENT (decoder +
    # losses | coordinate encoder | DPT decoder + heads | ViT blocks 6-11 | blocks 3-5 | stem + blocks 0-2: nn/autograd.py "Segmented
    # backward"); between two replays the buckets whose gradients are final are packed and all-reduced on RCCL's stream
    # while the next segment's backward replays, and the averages are copied back into the graphs' static gradients at the
    # end (parallel.GradReducer.begin_in_place / launch_done / end_in_place).  optim.hip_graph_segments=false: one graph, every
    # bucket behind it (round 5's form: no overlap).
    def _optimizer_step(self, opt):
        """:270-277: clip, step; under optim.amp through the loss scaler (unscale, skip on overflow, update the scale)."""
        scaler = getattr(self, "scaler", None)
        if scaler is not None:
            scaler.step(self.optim, opt.optim.clip_norm)
            return
        if opt.optim.clip_norm:
            self.optim.clip_grad_norm_(opt.optim.clip_norm)
        self.optim.step()

    def _step_capture_enabled(self, opt):
        flag = os.environ.get("ZS_TRAIN_HIP_GRAPH")
        on = flag not in ("0", "") if flag is not None else bool(opt.optim.get("hip_graph", False))
        return on and opt.optim.accum == 1

    def _side_wgrads(self, opt):
        """Weight-gradient GEMMs on a side stream beside the data-gradient chain (nn/autograd.py SIDE_WGRAD)?
        optim.side_wgrads / ZS_TRAIN_SIDE_WGRADS (A/B switch)."""
        flag = os.environ.get("ZS_TRAIN_SIDE_WGRADS")
        return flag not in ("0", "") if flag is not None else bool(opt.optim.get("side_wgrads", False))

    def _segmented_capture(self, opt):
        """Capture the step as one hipGraph per backward segment?  Only worth it when buckets are really exchanged (a reducer
        that is not a no-op); optim.hip_graph_segments / ZS_TRAIN_GRAPH_SEGMENTS = 0 keeps the single graph (A/B switch)."""
        flag = os.environ.get("ZS_TRAIN_GRAPH_SEGMENTS")
        on = flag not in ("0", "") if flag is not None else bool(opt.optim.get("hip_graph_segments", True))
        r = self.reducer
        return on and r is not None and (r.world > 1 or r.always)

    def _train_iteration_captured(self, opt, var):
        """One step through the captured launch sequence; the first two steps of a signature run eagerly (they
        allocate workspaces, pack operands and build the launch tables).

        Warm-up steps and the capture share ONE side stream, and warm-up losses are returned detached: autograd
        binds a parameter's gradient accumulator to the stream it was first used on and keeps it for as long as any
        graph that reaches it is alive.  An accumulator born on the default stream (a caller still holding the loss
        of an eager step) would pull the default stream into the capture - hipStreamEndCapture does not survive
        that."""
        from ..nn import autograd as A
        dev = next(self.graph.parameters()).device
        if getattr(self, "_capture_stream", None) is None:
            self._capture_stream = torch.cuda.Stream(device=dev)
        cs = self._capture_stream
        tensors = {k: v for k, v in var.items() if torch.is_tensor(v)}
        trainable = tuple(id(p) for p in self.graph.parameters() if p.requires_grad)
        sig = (tuple((k, tuple(v.shape), v.dtype, str(v.device)) for k, v in sorted(tensors.items())), trainable)
        st = getattr(self, "_captured", None)
        if st is not None and (st["sig"] != sig or st["scratch"] != A.SCRATCH_GENERATION[0]):
            st = self._captured = None             # other shapes, or a workspace moved (an evaluation in between)
            self._capture_warm = 0
        if st is None:
            if getattr(self, "_capture_warm", 0) < 2:
                self._capture_warm = getattr(self, "_capture_warm", 0) + 1
                main = torch.cuda.current_stream(dev)
                cs.wait_stream(main)
                with torch.cuda.stream(cs):
                    loss = self._train_iteration_eager(opt, var)
                    loss = edict({k: (v.detach() if torch.is_tensor(v) else v) for k, v in loss.items()})
                main.wait_stream(cs)
                return loss
            static = {k: v.clone() for k, v in tensors.items()}
            static_var = edict({k: static.get(k, v) for k, v in var.items()})
            self.optim.zero_grad(set_to_none=True)
            import gc
            gc.collect()                           # (torch.cuda.graph collects too: operands of dead modules must
            A.refresh_packs(dev)                   # leave the re-pack table before it is cached, not inside)
            A.bump_generation()                    # the capture must contain the operand re-pack
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            graphs, seg_params = [graph], None
            scaler = getattr(self, "scaler", None)
            if self.reducer is not None:
                self.reducer.armed = False         # no collective inside a capture: the buckets go out BETWEEN the replays
            if self._segmented_capture(opt):
                # forward + the last segment's backward in the first graph, every earlier segment's backward in a graph of its
                # own (one memory pool: the replays run in capture order) - nn/autograd.py "Segmented backward"
                params = [p for p in self.graph.parameters() if p.requires_grad]
                A.begin_segments()
                try:
                    with torch.cuda.graph(graph, stream=cs):
                        out_var, loss = self.graph.forward(opt, static_var, training=True, get_loss=True)
                        loss = self.summarize_loss(opt, out_var, loss)
                        last = A.SEGMENTS["index"]
                        with A.side_wgrads(self._side_wgrads(opt)):
                            A.backward_segment(last, params, loss=loss.all if scaler is None else scaler.scale_loss(loss.all))
                    have = {id(p) for p in params if p.grad is not None}
                    seg_params = [[p for p in params if id(p) in have]]
                    for s in range(last - 1, -1, -1):
                        if not A.segment_has_work(s):
                            continue                # (a frozen or absent part of the network: nothing to capture)
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, pool=graph.pool(), stream=cs), A.side_wgrads(self._side_wgrads(opt)):
                            ran = A.backward_segment(s, [p for p in params if id(p) not in have])
                        assert ran
                        new = [p for p in params if p.grad is not None and id(p) not in have]
                        have.update(id(p) for p in new)
                        graphs.append(g)
                        seg_params.append(new)
                finally:
                    A.end_segments()
            else:
                with torch.cuda.graph(graph, stream=cs):
                    out_var, loss = self.graph.forward(opt, static_var, training=True, get_loss=True)
                    loss = self.summarize_loss(opt, out_var, loss)
                    with A.side_wgrads(self._side_wgrads(opt)):
                        (loss.all if scaler is None else scaler.scale_loss(loss.all)).backward()
            loss = edict({k: (v.detach() if torch.is_tensor(v) else v) for k, v in loss.items()})
            st = self._captured = dict(sig=sig, scratch=A.SCRATCH_GENERATION[0], graph=graph, graphs=graphs,
                                       seg_params=seg_params, static=static, loss=loss)
        for k, t in st["static"].items():
            t.copy_(tensors[k], non_blocking=True)
        if st["seg_params"] is not None and self.reducer is not None and self.reducer.begin_in_place():
            # bucket k's all-reduce (RCCL's own stream) runs while the replay of the next segment's backward does
            for g, done in zip(st["graphs"], st["seg_params"]):
                g.replay()
                self.reducer.launch_done(done)
            self.reducer.end_in_place()
        else:
            for g in st["graphs"]:
                g.replay()
            if self.reducer is not None:
                self.reducer.reduce_in_place()
        self._optimizer_step(opt)                  # no zero_grad: the replay overwrites the static gradients
        if self._rank() == 0 and getattr(opt, "output_path", None) and not getattr(opt, "debug", False) \
                and self.it > 0 and self.it % opt.freq.ckpt_latest == 0:
            self.save_checkpoint(opt, ep=self.ep, it=self.it, best_val=self.best_val, best_ep=self.best_ep, latest=True)
        self.it += 1
        return st["loss"]

    def save_checkpoint(self, opt, ep=0, it=0, best_val=np.inf, best_ep=1, latest=False, best=False):
        """:525-529."""
        util.save_checkpoint(opt, self, ep=ep, it=it, best_val=best_val, best_ep=best_ep, latest=latest, best=best)

    @staticmethod
    def _rank():
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    def restore_checkpoint(self, opt, best=False, evaluate=False):
        """:138-151: `opt.resume` continues from <output_path>/latest.ckpt (graph + optimiser state);
        `opt.load` names a checkpoint written by the reference or by save_checkpoint."""
        epoch_start, iter_start = None, None
        if getattr(opt, "resume", False):
            epoch_start, iter_start, self.best_val, self.best_ep = util.restore_checkpoint(
                opt, self, resume=opt.resume, best=best, evaluate=evaluate)
        elif getattr(opt, "load", None):
            util.restore_checkpoint(opt, self, load_name=opt.load)
        self.epoch_start = epoch_start or 0
        self.iter_start = iter_start or 0

    @torch.no_grad()
    def evaluate_batch(self, opt, var, ep=None, it=None, single_gpu=False):
        """:518-523."""
        var = util.move_to_device(var, opt.device)
        return self.graph.forward(opt, var, training=False, get_loss=False)

    @torch.no_grad()
    def evaluate(self, opt, ep=0, training=False):
        """:335-516 without the visual dumps: returns dict(cd, dist_acc, dist_cov, f_scores) and, on
        rank 0 with `opt.output_path`, writes <dataset>_full_results.txt / quantitative_<dataset>.txt /
        cd_cat.txt in the reference's formats."""
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if self.reducer is not None and training:      # every rank validates with rank 0's BatchNorm statistics
            self.reducer.sync_buffers()
        self.graph.eval()                                                    # :337
        cd_accs, cd_comps, f_scores, cats, ids = [], [], [], [], []
        for it, batch in enumerate(self.test_loader):
            var = self.evaluate_batch(opt, edict(batch), ep, it)
            eval_3D.eval_metrics(opt, var, self.graph.impl_network)
            cd_accs.append(var.cd_acc.float().view(-1))
            cd_comps.append(var.cd_comp.float().view(-1))
            f_scores.append(var.f_score.float().view(len(var.cd_acc.view(-1)), -1))
            cats.append(torch.as_tensor(var.category_label).view(-1).to(opt.device))
            ids.append(torch.as_tensor(var.idx).view(-1).to(opt.device))
        cd_accs, cd_comps, f_scores = torch.cat(cd_accs), torch.cat(cd_comps), torch.cat(f_scores)
        cats, ids = torch.cat(cats).long(), torch.cat(ids).long()
        from .. import parallel
        if not getattr(opt.eval, "shard_image", False):     # (image sharding: every rank already holds every row)
            ids, (cd_accs, cd_comps, f_scores, cats) = parallel.gather_sample_rows(   # :414-432
                ids, [cd_accs, cd_comps, f_scores, cats])
        assert cd_accs.shape[0] == len(self.test_data)
        out = dict(dist_acc=cd_accs.mean().item(), dist_cov=cd_comps.mean().item(),
                   f_scores=f_scores.mean(0).tolist())
        out["cd"] = (out["dist_acc"] + out["dist_cov"]) / 2
        path = getattr(opt, "output_path", None)
        if rank == 0 and path and not training:
            os.makedirs(path, exist_ok=True)
            name = opt.data.dataset_test
            tag = getattr(self.test_data, "synthetic_standin", None)       # data/__init__.py: stand-in data says so
            tag = tag + "\n" if tag else ""
            with open(os.path.join(path, "{}_full_results.txt".format(name)), "w") as f:
                f.write(tag + "IND, CD, ACC, COMP, ")
                f.write(", ".join("F-score@{:.2f}".format(t * 100) for t in opt.eval.f_thresholds))
                for i in range(len(ids)):
                    f.write("\n{:d}".format(int(ids[i])))
                    f.write("\t{:.4f}".format((cd_accs[i].item() + cd_comps[i].item()) / 2))
                    f.write("\t{:.4f}\t{:.4f}".format(cd_accs[i].item(), cd_comps[i].item()))
                    f.write("\t" + "\t".join("{:.4f}".format(v) for v in f_scores[i].tolist()))
            with open(os.path.join(path, "quantitative_{}.txt".format(name)), "w") as f:
                f.write(tag + "CD     Acc    Comp \n")
                f.write("%.4f %.4f %.4f\n" % (out["cd"], out["dist_acc"], out["dist_cov"]))
                for t, v in zip(opt.eval.f_thresholds, out["f_scores"]):
                    f.write("F-score @ %.2f: %.4f\n" % (t * 100, v))
            label2cat = getattr(self.test_data, "label2cat", None)
            with open(os.path.join(path, "cd_cat.txt"), "w") as f:
                f.write(tag + "CD     Acc    Comp   Count Cat\n")
                for i in torch.unique(cats).tolist():
                    sel = cats == i
                    a, c = cd_accs[sel].mean().item(), cd_comps[sel].mean().item()
                    f.write("%.4f %.4f %.4f %5d %s\n" % ((a + c) / 2, a, c, int(sel.sum()),
                                                        label2cat[i] if label2cat else str(i)))
        return out
