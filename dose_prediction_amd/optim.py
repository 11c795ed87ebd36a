"""Fused Adam for the HIP path: drop-in for the optimizer NetworkTrainer.set_optimizer builds
(network_trainer.py:120-125: optim.Adam(params, lr, weight_decay=3e-5, betas=(0.9, 0.999), eps=1e-8, amsgrad=True)).
One kernel launch per step over all parameters (dp_adam_multi); same update rule and state names as torch.optim.Adam."""
import numpy as np
import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        self._plans = {}

    def _plan(self, gi, plist):
        key = (gi, tuple(id(p) for p in plist))
        plan = self._plans.get(key)
        if plan is None:
            chunk = _lib.lib().dp_adam_chunk()
            ct, ci = [], []
            for t, p in enumerate(plist):
                n = (p.numel() + chunk - 1) // chunk
                ct += [t] * n
                ci += list(range(n))
            dev = plist[0].device
            plan = (torch.tensor(ct, dtype=torch.int32, device=dev), torch.tensor(ci, dtype=torch.int32, device=dev), len(ct),
                    torch.empty((len(plist), 6), dtype=torch.int64).pin_memory(), torch.empty((len(plist), 6), dtype=torch.int64, device=dev))
            self._plans[key] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise _lib.DoseHipError("FusedAdam needs fp32 CUDA/HIP parameters and gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format) if group["amsgrad"] else None
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise _lib.DoseHipError("FusedAdam needs contiguous parameters / gradients")
            steps = {self.state[p]["step"] for p in plist}
            if len(steps) != 1:
                raise _lib.DoseHipError("FusedAdam: parameters of one group must share the step count")
            step = steps.pop() + 1
            ct, ci, nchunks, host, devtab = self._plan(gi, plist)
            h = host.numpy()
            for t, p in enumerate(plist):
                st = self.state[p]
                vm = st["max_exp_avg_sq"]
                h[t] = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                        vm.data_ptr() if vm is not None else 0, p.numel())
                st["step"] = step
            devtab.copy_(host, non_blocking=True)
            b1, b2 = group["betas"]
            _lib.call("dp_adam_multi", devtab.data_ptr(), ct.data_ptr(), ci.data_ptr(), nchunks, float(group["lr"]), float(b1), float(b2),
                      float(group["eps"]), float(group["weight_decay"]), int(step), 1 if group["amsgrad"] else 0,
                      torch.cuda.current_stream().cuda_stream)
        return loss
