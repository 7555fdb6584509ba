"""Adam on the device, hand-written: the optimiser step of the reference's training loop -- ``torch.optim.Adam(params, lr, betas=(0.9, 0.999),
eps=1e-8)``, /root/reference/train.py:91, no weight decay, no amsgrad -- as ONE multi-tensor HIP launch per parameter group
(``csbsr_adam_step``, csrc/elementwise.hip) instead of torch's foreach kernels (52 launches, 3 ms per config-2 step; the last vendor /
torch arithmetic on the timed path besides a few [B, 441] einsums).

Drop-in for ``torch.optim.Adam`` in that configuration: a ``torch.optim.Optimizer`` subclass (``LambdaLR`` and the reference's warm-up schedulers
work on it unchanged), the same per-parameter state keys (``step``, ``exp_avg``, ``exp_avg_sq``: ``state_dict()`` interchanges with
``torch.optim.Adam``), the same skip rule (a parameter whose ``.grad`` is None is not touched: no moment decay, no step count -- what the frozen
training phases and an overflowed backward rely on), the same arithmetic operation by operation (agreement to fp32 rounding:
tests/test_elementwise_gpu.py::test_adam_step_matches_torch).  fp32 parameters on the device only; anything else raises.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L

_CHUNK = 8192
_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("step_size", "<f4"), ("bc2_sqrt", "<f4")])
assert _DT.itemsize == 48


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        L.load()
        self._maps = {}          # tuple of tensor sizes -> (block_tensor, block_chunk) device int32 tensors
        self._host = None        # pinned staging for the per-step table (+ the event that says the last upload has been read)

    def _block_maps(self, sizes, device):
        key = (tuple(sizes), str(device))
        mp = self._maps.get(key)
        if mp is None:
            bt, bc = [], []
            for i, n in enumerate(sizes):
                k = (n + _CHUNK - 1) // _CHUNK
                bt.append(np.full(k, i, dtype=np.int32))
                bc.append(np.arange(k, dtype=np.int32))
            mp = (torch.from_numpy(np.concatenate(bt)).to(device), torch.from_numpy(np.concatenate(bc)).to(device))
            self._maps[key] = mp
        return mp

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            beta1, beta2 = group["betas"]
            lr = float(group["lr"])
            dev = ps[0].device
            tab = np.zeros(len(ps), dtype=_DT)
            for i, p in enumerate(ps):
                g = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.dtype == torch.float32 and g.is_contiguous()
                        and g.device == p.device == dev and not g.is_sparse):
                    raise L.CsbsrHipError("csbsr_amd.optim.Adam: contiguous fp32 parameters and gradients on one device only")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                t = float(st["step"])
                tab[i] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                          lr / (1.0 - beta1 ** t), math.sqrt(1.0 - beta2 ** t))
            bt, bc = self._block_maps([p.numel() for p in ps], dev)
            raw = torch.from_numpy(tab.view(np.uint8))
            if self._host is not None and self._host[0].numel() >= raw.numel() and self._host[1].device == dev:
                self._host[2].synchronize()          # (the previous step's upload was consumed long ago: returns at once)
            else:
                n = max(raw.numel(), 4096)
                self._host = [torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8, device=dev), torch.cuda.Event()]
            self._host[0][:raw.numel()].copy_(raw)
            with torch.cuda.device(dev):
                self._host[1][:raw.numel()].copy_(self._host[0][:raw.numel()], non_blocking=True)
                stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                L.call("csbsr_adam_step", C.c_void_p(self._host[1].data_ptr()), C.c_void_p(bt.data_ptr()), C.c_void_p(bc.data_ptr()),
                       int(bt.numel()), float(beta1), float(beta2), float(group["eps"]), stream)
                self._host[2].record(torch.cuda.current_stream(dev))
        return loss
