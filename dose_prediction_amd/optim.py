"""Fused Adam for the HIP path: drop-in for the optimizer NetworkTrainer.set_optimizer builds
(network_trainer.py:120-125: optim.Adam(params, lr, weight_decay=3e-5, betas=(0.9, 0.999), eps=1e-8, amsgrad=True)).
One kernel launch per step over all parameters (dp_adam_multi) plus ONE launch that rebuilds the kernel-layout copies of the
updated weights (dp_pack_multi); same update rule and state names as torch.optim.Adam, and its state_dict()s load either way
(network_trainer.py:340-363 saves / restores `optimizer_state_dict`)."""
import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    """capturable=True keeps the step count on the device (state['step'] is a 0-dim int32 device tensor shared by a group's
    parameters), so that a step captured in a HIP graph replays with the right bias corrections."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, capturable=False,
                 grad_scale=None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, capturable=capturable))
        self._plans = {}
        self.fuse_packs = True           # (A/B switch: False sends every packed copy through ops.refresh_packs)
        # 0-dim int32 device tensor raised by the kernel when a gradient element was not finite (that element's update is skipped);
        # it is never cleared by step(): read it with found_inf() / clear it with reset_found_inf()
        self._found_inf = None
        # gradients arrive multiplied by this factor (config.set_loss_scale: fp16 training); None = follow the global setting
        self.grad_scale = grad_scale

    def __setstate__(self, state):
        super().__setstate__(state)
        for g in self.param_groups:
            g.setdefault("amsgrad", False)
            g.setdefault("capturable", False)

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
                    torch.empty((len(plist), 10), dtype=torch.int64).pin_memory(), torch.empty((len(plist), 10), dtype=torch.int64, device=dev),
                    [None])
            self._plans[key] = plan
        return plan

    def found_inf(self):
        """True when any step() since the last reset_found_inf() met a non-finite gradient element (fp16 overflow under a static
        loss scale): those elements were skipped, parameter and Adam moments untouched.  Synchronises with the device."""
        return self._found_inf is not None and bool(self._found_inf.item())

    def reset_found_inf(self):
        if self._found_inf is not None:
            self._found_inf.zero_()

    def step_was_clean(self, reset=True):
        """GradScaler-style use under a static loss scale: call after step(); False means some gradient elements were not
        finite and THOSE elements were left untouched while the rest of the step was applied (torch's GradScaler would have skipped
        the whole step) -- lower the loss scale (config.set_loss_scale) and carry on, or restore a checkpoint.  Synchronises."""
        bad = self.found_inf()
        if reset and bad:
            self.reset_found_inf()
        return not bad

    @staticmethod
    def _step_value(s):
        """state['step'] as written by this class (int), by torch.optim.Adam (0-dim float tensor) or by capturable mode."""
        return int(s.item()) if torch.is_tensor(s) else int(s)

    @torch.no_grad()
    def step(self, closure=None):
        from . import ops
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ops.flush_deferred()          # (weight gradients recorded for the grouped launch: normally flushed at the end of backward)
        ops.join_wgrad_stream()       # (convolution weight gradients launched on their own stream: normally joined there too)
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            capt = bool(group.get("capturable", False))
            for p in plist:
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise _lib.DoseHipError("FusedAdam needs fp32 CUDA/HIP parameters and gradients")
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if group["amsgrad"] and st.get("max_exp_avg_sq") is None:
                    st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise _lib.DoseHipError("FusedAdam needs contiguous parameters / gradients")
            ct, ci, nchunks, host, devtab, copied = self._plan(gi, plist)
            if copied[0] is not None and not torch.cuda.is_current_stream_capturing():     # (torch.cuda.graph synchronises on entry)
                # the pinned table is rewritten below: the previous step's asynchronous copy out of it must have executed (the host
                # thread may be several steps ahead of the GPU)
                copied[0].synchronize()
            h = host.numpy()
            fused = {}
            for t, p in enumerate(plist):
                st = self.state[p]
                vm = st.get("max_exp_avg_sq") if group["amsgrad"] else None
                # one packed copy per parameter is written by the update kernel itself (plain 16-bit casts and fp32x3 Linear operands:
                # the layouts that follow the parameter's own element order); ops.refresh_packs rebuilds the others
                key, cdst, ckind, K, cpp = ops.adam_fusable_pack(p) if self.fuse_packs else (None, 0, 0, 0, 0)
                if key is not None:
                    fused[id(p)] = key
                h[t] = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                        vm.data_ptr() if vm is not None else 0, p.numel(), cdst, ckind, K, cpp)
            devtab.copy_(host, non_blocking=True)
            if not torch.cuda.is_current_stream_capturing():
                copied[0] = torch.cuda.Event()
                copied[0].record()
            b1, b2 = group["betas"]
            stream = torch.cuda.current_stream().cuda_stream
            from . import config
            gs = float(self.grad_scale if self.grad_scale is not None else config.effective_loss_scale())
            inv_gs = 1.0 / gs
            if self._found_inf is None or self._found_inf.device != plist[0].device:
                self._found_inf = torch.zeros((), dtype=torch.int32, device=plist[0].device)
            if capt:
                sdev = self.state[plist[0]]["step"]
                if not (torch.is_tensor(sdev) and sdev.is_cuda and sdev.dtype == torch.int32):
                    # (before any capture) one shared device counter per group, initialised from the host-side counts
                    steps = {self._step_value(self.state[p]["step"]) for p in plist}
                    if len(steps) != 1:
                        raise _lib.DoseHipError("FusedAdam: parameters of one group must share the step count")
                    sdev = torch.full((), steps.pop(), dtype=torch.int32, device=plist[0].device)
                for p in plist:
                    self.state[p]["step"] = sdev
                _lib.call("dp_adam_multi_dev", devtab.data_ptr(), ct.data_ptr(), ci.data_ptr(), nchunks, float(group["lr"]), float(b1),
                          float(b2), float(group["eps"]), float(group["weight_decay"]), inv_gs, sdev.data_ptr(), 1 if group["amsgrad"] else 0, self._found_inf.data_ptr(), stream)
            else:
                steps = {self._step_value(self.state[p]["step"]) for p in plist}
                if len(steps) != 1:
                    raise _lib.DoseHipError("FusedAdam: parameters of one group must share the step count")
                step = steps.pop() + 1
                for p in plist:
                    self.state[p]["step"] = step
                _lib.call("dp_adam_multi", devtab.data_ptr(), ct.data_ptr(), ci.data_ptr(), nchunks, float(group["lr"]), float(b1), float(b2),
                          float(group["eps"]), float(group["weight_decay"]), inv_gs, int(step), 1 if group["amsgrad"] else 0,
                          self._found_inf.data_ptr(), stream)
            # the kernel wrote the parameters through raw pointers (p._version did not move): rebuild their packed copies now
            ops.refresh_packs(plist, fused)
        return loss
