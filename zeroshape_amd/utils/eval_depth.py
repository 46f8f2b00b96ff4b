"""Mirror of the reference's utils/eval_depth.py (DepthMetric): scale/shift-aligned depth
metrics, one fused launch per batch (zs_depth_metrics, csrc/depth_metrics.hip) instead of
the reference's ~60 masked-index kernels.  GPU only."""
import ctypes

import torch

PRED_IS_DISPARITY, SOLVE_ONLY = 1, 2     # ZS_DEPTH_* flags, include/zeroshape_hip.h


class DepthMetric:
    def __init__(self, thresholds=[1.25, 1.25**2, 1.25**3], depth_cap=None, prediction_type='depth'):
        self.thresholds = thresholds
        self.depth_cap = depth_cap
        self.metric_keys = self.get_metric_keys()
        self.prediction_type = prediction_type

    def get_metric_keys(self):
        """utils/eval_depth.py:36-44."""
        return ['d>{}'.format(t) for t in self.thresholds] + ['rmse', 'l1_err', 'abs_rel']

    def _run(self, prediction, target, mask, flags, want_depth):
        from .. import _lib
        lib = _lib.load()
        if not prediction.is_cuda:
            raise ValueError("DepthMetric: GPU tensors required (there is no CPU path)")
        p = prediction.detach().float().contiguous()
        t = target.detach().float().contiguous()
        m = mask.detach().float().contiguous()
        assert p.shape == t.shape == m.shape
        B = p.shape[0]
        n = p[0].numel()
        k = len(self.thresholds)
        thr = (ctypes.c_float * max(k, 1))(*[float(x) for x in self.thresholds])
        metrics = torch.empty(B, k + 3, dtype=torch.float32, device=p.device)
        depth = torch.empty_like(p) if want_depth else None
        ss = torch.empty(B, 2, dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(lib.zs_depth_metrics(_lib.ptr(p), _lib.ptr(t), _lib.ptr(m), B, n, flags,
                                            float(self.depth_cap) if self.depth_cap is not None else 0.0,
                                            thr, k, _lib.ptr(metrics), _lib.ptr(depth), _lib.ptr(ss),
                                            _lib.current_stream_ptr(p.device)), "zs_depth_metrics")
        return metrics, depth, ss

    def compute_scale_and_shift(self, prediction, target, mask):
        """utils/eval_depth.py:11-34: least-squares (scale, shift) per image [B,H,W] of
        ``scale * prediction + shift ~ target`` over the masked pixels; both returned as [B]."""
        _, _, ss = self._run(prediction, target, mask, SOLVE_ONLY, False)
        return ss[:, 0], ss[:, 1]

    def compute_metrics(self, prediction, target, mask):
        """utils/eval_depth.py:46-116: prediction/target/mask [B,1,H,W] -> (dict of [B] metrics,
        aligned depth [B,1,H,W])."""
        assert prediction.shape == target.shape == mask.shape
        assert len(prediction.shape) == 4
        assert prediction.shape[1] == 1
        if self.prediction_type == 'depth':
            flags = 0
        elif self.prediction_type == 'disparity':
            flags = PRED_IS_DISPARITY
        else:
            raise ValueError('Unknown prediction type: {}'.format(self.prediction_type))
        metrics, depth, _ = self._run(prediction, target, mask, flags, True)
        out = {key: metrics[:, i] for i, key in enumerate(self.metric_keys)}
        return out, depth
