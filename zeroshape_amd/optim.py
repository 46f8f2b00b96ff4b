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

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def grad_norm(self):
        """L2 norm of all gradients (device scalar), one launch per parameter class (zs_sumsq_multi over the
        optimiser's own table: it reads the gradient pointers and sizes only)."""
        lib = _lib.load()
        plan = self._prepare()
        if not plan["classes"]:
            return None
        total = None
        for cls in plan["classes"]:
            dev = cls["device"]
            partial = torch.empty(cls["nchunks"], dtype=torch.float32, device=dev)
            out = torch.empty((), dtype=torch.float32, device=dev)
            with _lib.on(dev):
                _lib.check(lib.zs_sumsq_multi(_lib.ptr(cls["tab_dev"]), _lib.ptr(cls["ct"]), _lib.ptr(cls["cs"]), cls["nchunks"],
                                              _lib.ptr(partial), _lib.ptr(out), _lib.current_stream_ptr(dev)), "zs_sumsq_multi")
            total = out if total is None else total + out.to(total.device)
        return total.sqrt_()

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ folded into the next step(): returns the total norm
        (device scalar) and arms the scale min(1, max_norm / (norm + 1e-6))."""
        norm = self.grad_norm()
        if norm is not None:
            self._clip = (max_norm / (norm + 1e-6)).clamp_(max=1.0)
        return norm

    @torch.no_grad()
    def step(self, closure=None):
        """One zs_adamw_multi launch per (step count, betas, eps, device) class - one in all for a model trained from
        the start.  The per-parameter bookkeeping is cached while the set of parameters with gradients stays the
        same: per step only the gradient pointers and the groups' lr / weight_decay are refreshed in the host table,
        and the table travels to the device only if a byte of it changed (eager steps get fresh gradient buffers,
        a captured step keeps them).  Step counts are Python integers, mirrored into the torch-format `step` tensors
        when the state is read (state_dict)."""
        assert closure is None
        lib = _lib.load()
        plan = self._prepare()
        counts = self.__dict__.setdefault("_counts", {})
        for cls in plan["classes"]:
            cls["step"] += 1
            for p in cls["params"]:
                counts[p] = cls["step"]
            betas, eps = cls["betas"], cls["eps"]
            with _lib.on(cls["device"]):
                _lib.check(lib.zs_adamw_multi(_lib.ptr(cls["tab_dev"]), _lib.ptr(cls["ct"]), _lib.ptr(cls["cs"]), cls["nchunks"],
                                              betas[0], betas[1], eps, cls["step"], _lib.ptr(self._clip),
                                              _lib.ptr(cls["skipped"]),
                                              _lib.current_stream_ptr(cls["device"])), "zs_adamw_multi")
        self._clip = None
        A.bump_generation()                   # parameters changed behind torch's version counters
        return None

    def _prepare(self):
        """The plan for the parameters that have gradients now, its device tables current."""
        groups = self.param_groups
        live = [(gi, p) for gi, g in enumerate(groups) for p in g["params"] if p.grad is not None]
        ids = tuple(id(p) for _, p in live)
        plan = self.__dict__.get("_plan")
        if plan is None or plan["ids"] != ids:
            plan = self._build_plan(live)
        for cls in plan["classes"]:
            params = cls["params"]
            for p in params:
                g = p.grad
                if not (g.is_cuda and g.dtype == torch.float32):
                    raise ValueError("FusedAdamW: fp32 GPU gradients required")
                if not g.is_contiguous():
                    p.grad = g.contiguous()
            tab = cls["tab"]
            tab["grad"] = [p.grad.data_ptr() for p in params]
            tab["lr"] = np.asarray([groups[gi]["lr"] for gi in cls["gidx"]], np.float32)
            tab["wd"] = np.asarray([groups[gi]["weight_decay"] for gi in cls["gidx"]], np.float32)
            raw = tab.tobytes()
            if raw != cls["sent"]:
                cls["tab_dev"] = torch.from_numpy(tab.view(np.uint8).reshape(-1).copy()).to(cls["device"], non_blocking=True)
                cls["sent"] = raw
        return plan

    def _build_plan(self, live):
        """Classes of parameters that share (step count, betas, eps, device), each with its host table, chunk tables
        on the device and the group index of every parameter."""
        counts = self.__dict__.setdefault("_counts", {})
        skip_of = self.__dict__.setdefault("_skip_of", {})
        by_key = {}
        for gi, p in live:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise ValueError("FusedAdamW: contiguous fp32 GPU parameters required (no CPU path)")
            st = self._init_state(p)
            for k in ("exp_avg", "exp_avg_sq"):
                if st[k].device != p.device or not st[k].is_contiguous():
                    st[k] = st[k].to(p.device).contiguous()
            if p not in counts:
                counts[p] = int(st["step"])
            g = self.param_groups[gi]
            # skipped (overflowed) step() calls are counted per CLASS, on the class's device (one optimiser-wide
            # counter charged parameters that joined later - unfrozen, restored - with skips from before they joined): a
            # parameter keeps the counter it first got; parameters with different counters never share a class
            by_key.setdefault((counts[p], g["betas"], g["eps"], p.device, id(skip_of.get(p))), []).append((gi, p))
        classes = []
        for (step, betas, eps, dev, _), members in by_key.items():
            params = [p for _, p in members]
            entries = [(p.data_ptr(), 0, self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(),
                        p.numel(), 0.0, 0.0) for p in params]
            _, ct, cs, nchunks = build_table(entries, dev)
            tab = np.zeros(len(entries), _ENTRY)
            for i, e in enumerate(entries):
                tab[i] = e
            skipped = skip_of.get(params[0])
            if skipped is None:
                skipped = torch.zeros((), dtype=torch.int32, device=dev)
            for p in params:
                skip_of[p] = skipped
            classes.append(dict(params=params, gidx=[gi for gi, _ in members], tab=tab, ct=ct, cs=cs, nchunks=nchunks,
                                betas=betas, eps=eps, device=dev, step=step, sent=None, tab_dev=None, skipped=skipped))
        plan = self._plan = dict(ids=tuple(id(p) for _, p in live), classes=classes)
        return plan

    def count_skipped_steps(self, found_inf):
        """A loss scaler's overflow flag (device scalar, 1.0 = this step() call was skipped): accumulated on the device in the
        counter of every parameter class that exists now, subtracted from the call count in the kernel's bias correction and
        in the `step` values of state_dict()."""
        if self.__dict__.get("_plan") is None:
            self._prepare()
        seen = set()
        for t in self.__dict__.get("_skip_of", {}).values():
            if id(t) not in seen:
                seen.add(id(t))
                t.add_(found_inf.to(device=t.device, dtype=torch.int32))

    def _sync_step_tensors(self):
        skip_of, host = self.__dict__.get("_skip_of", {}), {}
        for p, n in self.__dict__.get("_counts", {}).items():
            if p in self.state and "step" in self.state[p]:
                t = skip_of.get(p)
                if t is not None and id(t) not in host:
                    host[id(t)] = int(t)                                  # one host read per class, when the state is saved
                self.state[p]["step"] = torch.tensor(float(max(n - (host[id(t)] if t is not None else 0), 0)), dtype=torch.float32)

    def state_dict(self):
        self._sync_step_tensors()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.__dict__.pop("_plan", None)       # moments were replaced: new pointers, and the step counts of the file
        self.__dict__.pop("_skip_of", None)    # (the file's counts are applied steps)
        self._counts = {p: int(st["step"]) for p, st in self.state.items() if "step" in st}


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
        self.min_scale = 2.0 ** -24

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
        optim.count_skipped_steps(self.found_inf)      # before the launch: the kernel's bias correction reads the total
        optim.step()
        torch._amp_update_scale_(self.scale, self.tracker, self.found_inf, self.growth_factor, self.backoff_factor,
                                 self.growth_interval)
        # (not in GradScaler) gradients that are nan at ANY scale - a collapsed depth map makes the reference's
        # max-radius normalisation 0 / 0 - would halve the scale to 0 in 150 steps, and 1 / 0 then poisons every later
        # step; with a floor the run recovers when the gradients do
        self.scale.clamp_(min=self.min_scale)
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
