"""Mirror of the evaluation half of the reference's model/shape_engine.py::Runner
(:335-523): build the graph, restore a checkpoint, loop a test loader through
Graph.forward + eval_metrics, gather the per-sample metrics over the ranks, write the
reference's result files.  Training (:248-297) is not built."""
import os

import torch
import torch.distributed as dist

from ..utils import eval_3D, util
from ..utils.options import EasyDict as edict
from .compute_graph.graph_shape import Graph


class Runner:
    def __init__(self, opt):
        self.opt = opt
        self.test_data = self.test_loader = None

    def load_dataset(self, opt, eval_split="test", dataset=None):
        """:52-81 (test side): `dataset` defaults to the analytic stand-in, sharded over the ranks
        with a DistributedSampler when world_size > 1."""
        from ..data import synthetic
        self.test_data = dataset if dataset is not None else synthetic.Dataset(opt, split=eval_split)
        sampler = None
        if getattr(opt, "world_size", 1) > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(self.test_data, shuffle=False, drop_last=False)
        self.test_loader = torch.utils.data.DataLoader(self.test_data, batch_size=opt.eval.batch_size, shuffle=False,
                                                       sampler=sampler, num_workers=0, drop_last=False)

    def build_networks(self, opt):
        self.graph = Graph(opt).to(opt.device).eval()

    def restore_checkpoint(self, opt, best=False, evaluate=False):
        """:176-190: `opt.load` names a checkpoint written by the reference or by save_checkpoint."""
        if getattr(opt, "load", None):
            util.restore_checkpoint(opt, self, load_name=opt.load)

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
            with open(os.path.join(path, "{}_full_results.txt".format(name)), "w") as f:
                f.write("IND, CD, ACC, COMP, ")
                f.write(", ".join("F-score@{:.2f}".format(t * 100) for t in opt.eval.f_thresholds))
                for i in range(len(ids)):
                    f.write("\n{:d}".format(int(ids[i])))
                    f.write("\t{:.4f}".format((cd_accs[i].item() + cd_comps[i].item()) / 2))
                    f.write("\t{:.4f}\t{:.4f}".format(cd_accs[i].item(), cd_comps[i].item()))
                    f.write("\t" + "\t".join("{:.4f}".format(v) for v in f_scores[i].tolist()))
            with open(os.path.join(path, "quantitative_{}.txt".format(name)), "w") as f:
                f.write("CD     Acc    Comp \n")
                f.write("%.4f %.4f %.4f\n" % (out["cd"], out["dist_acc"], out["dist_cov"]))
                for t, v in zip(opt.eval.f_thresholds, out["f_scores"]):
                    f.write("F-score @ %.2f: %.4f\n" % (t * 100, v))
            label2cat = getattr(self.test_data, "label2cat", None)
            with open(os.path.join(path, "cd_cat.txt"), "w") as f:
                f.write("CD     Acc    Comp   Count Cat\n")
                for i in torch.unique(cats).tolist():
                    sel = cats == i
                    a, c = cd_accs[sel].mean().item(), cd_comps[sel].mean().item()
                    f.write("%.4f %.4f %.4f %5d %s\n" % ((a + c) / 2, a, c, int(sel.sum()),
                                                        label2cat[i] if label2cat else str(i)))
        return out
