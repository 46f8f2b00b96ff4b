"""Base class of the encoder modules: a torch.nn.Module that only HOLDS parameters (so the
reference's checkpoints load by name) and runs its forward on the HIP library from a packed
copy of them."""
import torch
import torch.nn as nn


class HipModule(nn.Module):
    """Caches `self._pack(sd, device)` until a parameter or buffer changes."""

    def __init__(self):
        super().__init__()
        self._packed_cache = None

    def _tensors_key(self):
        from . import autograd
        return (autograd.GENERATION[0],) + tuple((t.data_ptr(), t._version)
                                                 for t in list(self.parameters()) + list(self.buffers()))

    def packed(self, device):
        key = (str(device),) + self._tensors_key()
        if self._packed_cache is None or self._packed_cache[0] != key:
            if self.training:
                raise NotImplementedError(
                    "%s: the packed (folded-BatchNorm) path is the inference path; in .train() mode call the "
                    "module under autograd (torch.enable_grad) or switch to .eval()" % type(self).__name__)
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            self._packed_cache = (key, self._pack(sd, torch.device(device)))
        return self._packed_cache[1]

    def _pack(self, sd, device):
        raise NotImplementedError

    @staticmethod
    def _need_gpu(t, what):
        if not (torch.is_tensor(t) and t.is_cuda):
            raise ValueError("%s must be a GPU tensor; zeroshape_amd has no CPU path" % what)
