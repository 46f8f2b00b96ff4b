"""Fused AdamW on the HIP library: torch.optim.AdamW semantics (the optimiser of the reference,
model/shape_engine.py:132, betas (0.9, 0.95)) with ONE launch for all parameters
(zs_adamw_multi over a device table of tensors) instead of ~10 small kernels per parameter.

It subclasses torch.optim.Optimizer only for the bookkeeping: param_groups with per-group lr /
weight_decay (the reference's four groups, model/shape_engine.py:80-131), and a state_dict in
torch.optim.AdamW's format ({step, exp_avg, exp_avg_sq} per parameter), so the `optim` entry of a
reference checkpoint (utils/util.py:261-270) loads and saves unchanged.
"""
import numpy as np
import torch

from . import _lib
from .nn import autograd as A

_ENTRY = np.dtype([("param", "<u8"), ("grad", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<u8"), ("lr", "<f4"),
                   ("wd", "<f4")])
assert _ENTRY.itemsize == 48


def build_table(entries, device):
    """entries: list of (param_ptr, grad_ptr, m_ptr, v_ptr, n, lr, wd) -> (table, chunk_tensor,
    chunk_start, n_chunks) device tensors for the zs_*_multi kernels."""
    chunk = _lib.load().zs_multi_tensor_chunk_elems()
    tab = np.zeros(len(entries), _ENTRY)
    ct, cs = [], []
    for i, e in enumerate(entries):
        tab[i] = e
        n = int(e[4])
        starts = np.arange(0, n, chunk, dtype=np.uint64)
        ct.append(np.full(len(starts), i, np.int32))
        cs.append(starts)
    ct = np.concatenate(ct) if ct else np.zeros(0, np.int32)
    cs = np.concatenate(cs) if cs else np.zeros(0, np.uint64)
    to = lambda a, dt: torch.from_numpy(a.view(dt) if a.dtype != dt else a).to(device, non_blocking=True)   # noqa: E731
    return (to(tab.view(np.uint8).reshape(-1), np.uint8), to(ct, np.int32), to(cs.view(np.int64), np.int64), len(ct))


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self._clip = None          # device scalar multiplying every gradient in the next step()

    def _live(self):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    yield group, p

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def grad_norm(self):
        """L2 norm of all gradients (device scalar), one launch (zs_sumsq_multi)."""
        lib = _lib.load()
        live = [p for _, p in self._live()]
        if not live:
            return None
        dev = live[0].device
        for p in live:
            self._check(p)
        entries = [(0, p.grad.data_ptr(), 0, 0, p.numel(), 0.0, 0.0) for p in live]
        tab, ct, cs, nchunks = build_table(entries, dev)
        partial = torch.empty(nchunks, dtype=torch.float32, device=dev)
        out = torch.empty((), dtype=torch.float32, device=dev)
        with _lib.on(dev):
            _lib.check(lib.zs_sumsq_multi(_lib.ptr(tab), _lib.ptr(ct), _lib.ptr(cs), nchunks, _lib.ptr(partial),
                                          _lib.ptr(out), _lib.current_stream_ptr(dev)), "zs_sumsq_multi")
        return out.sqrt_()

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ folded into the next step(): returns the total norm
        (device scalar) and arms the scale min(1, max_norm / (norm + 1e-6))."""
        norm = self.grad_norm()
        if norm is not None:
            self._clip = (max_norm / (norm + 1e-6)).clamp_(max=1.0)
        return norm

    @staticmethod
    def _check(p):
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
            raise ValueError("FusedAdamW: contiguous fp32 GPU parameters required (no CPU path)")
        if not (p.grad.is_cuda and p.grad.dtype == torch.float32):
            raise ValueError("FusedAdamW: fp32 GPU gradients required")
        if not p.grad.is_contiguous():
            p.grad = p.grad.contiguous()

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        lib = _lib.load()
        by_key = {}
        for group, p in self._live():
            self._check(p)
            st = self._init_state(p)
            for k in ("exp_avg", "exp_avg_sq"):
                if st[k].device != p.device or not st[k].is_contiguous():
                    st[k] = st[k].to(p.device).contiguous()
            st["step"] += 1
            key = (int(st["step"]), group["betas"], group["eps"], p.device)
            by_key.setdefault(key, []).append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                               st["exp_avg_sq"].data_ptr(), p.numel(), group["lr"],
                                               group["weight_decay"]))
        for (step, betas, eps, dev), entries in by_key.items():
            cache = self.__dict__.setdefault("_tables", {})
            hit = cache.get((betas, eps, dev))
            if hit is None or hit[0] != entries:   # same tensors, learning rates and decays as last step: same table
                hit = cache[(betas, eps, dev)] = (entries, build_table(entries, dev))
            tab, ct, cs, nchunks = hit[1]
            with _lib.on(dev):
                _lib.check(lib.zs_adamw_multi(_lib.ptr(tab), _lib.ptr(ct), _lib.ptr(cs), nchunks, betas[0], betas[1],
                                              eps, step, _lib.ptr(self._clip), _lib.current_stream_ptr(dev)),
                           "zs_adamw_multi")
        self._clip = None
        A.bump_generation()                   # parameters changed behind torch's version counters
        return None


class LossScaler:
    """Dynamic loss scale for split-fp16 data gradients (optim.amp), torch.cuda.amp.GradScaler's rules (the
    reference: model/shape_engine.py:135-136, :252-269) with every scalar on the device - no host read-back, so
    the step stays capturable: scale 2^16 at the start, x 0.5 and the step skipped when any gradient is inf / nan,
    x 2 after `growth_interval` clean steps in a row."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.scale = torch.full((), float(init_scale), dtype=torch.float32, device=device)
        self.tracker = torch.zeros((), dtype=torch.int32, device=device)
        self.found_inf = torch.zeros((), dtype=torch.float32, device=device)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval

    def scale_loss(self, loss):
        return loss * self.scale

    @torch.no_grad()
    def step(self, optim, clip_norm=None):
        """Unscale (folded into the optimiser's gradient multiplier), clip, step - or skip on overflow - and update
        the scale.  Returns the unscaled gradient norm (device scalar)."""
        norm = optim.grad_norm()                         # of the scaled gradients: inf / nan if any overflowed
        if norm is None:
            return None
        inv = 1.0 / self.scale
        true_norm = norm * inv
        mult = inv if not clip_norm else inv * (clip_norm / (true_norm + 1e-6)).clamp(max=1.0)
        finite = torch.isfinite(norm)
        self.found_inf.copy_((~finite).float())
        optim._clip = torch.where(finite, mult, torch.zeros_like(mult))      # 0: zs_adamw_multi leaves everything as is
        optim.step()
        torch._amp_update_scale_(self.scale, self.tracker, self.found_inf, self.growth_factor, self.backoff_factor,
                                 self.growth_interval)
        return true_norm

    def state_dict(self):
        """torch.cuda.amp.GradScaler.state_dict()'s keys: the `scaler` entry of a reference checkpoint loads unchanged."""
        return {"scale": float(self.scale), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(self.tracker)}

    def load_state_dict(self, sd):
        if not sd:                                   # a disabled GradScaler saves {}
            return
        self.scale.fill_(float(sd["scale"]))
        self.tracker.fill_(int(sd.get("_growth_tracker", 0)))
        self.growth_factor = sd.get("growth_factor", self.growth_factor)
        self.backoff_factor = sd.get("backoff_factor", self.backoff_factor)
        self.growth_interval = sd.get("growth_interval", self.growth_interval)
