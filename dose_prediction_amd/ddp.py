"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI (torch.distributed 'nccl' IS RCCL on
ROCm), bucketed and overlapped with the backward pass.  The reference's only multi-GPU code is nn.DataParallel
(network_trainer.py:104: single process, replicate/scatter/gather through GPU 0); this replaces it.

The reducer is attached IN PLACE (hooks on the parameters, no wrapper module), so ``state_dict()`` keys and the
unchanged ``NetworkTrainer.forward/backward`` (network_trainer.py:185-213) keep working: gradients are averaged by the
time ``loss.backward()`` returns.

Design for 8 x MI355X: gradients are packed in REVERSE registration order (approximately the order autograd produces
them) into flat fp32 buckets; a bucket is all-reduced asynchronously as soon as its last gradient arrives, on RCCL's own
stream, while the remaining backward kernels keep the compute stream busy.  The patch-embedding weight (78.6 M of
162.6 M elements) is produced last and is given a bucket of its own, split in chunks, so its exchange starts
immediately and the tail is one chunk long.  Parameters that receive no gradient in a step (MainSubsetModel.out,
cls_token, UnetResBlock.conv3 when Cin == Cout, frozen net_A) are handled at the end-of-backward callback.
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, module, bucket_mb=32.0, process_group=None, broadcast=True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.backend = dist.get_backend(process_group)
        if broadcast:
            self._broadcast_state(module)
        cap = int(bucket_mb * (1 << 20) / 4)
        # buckets: lists of (param, offset, numel); very large tensors are split into `cap`-sized chunks of one bucket each
        self.buckets, cur, cur_n = [], [], 0
        for p in reversed(self.params):
            n = p.numel()
            if n >= cap:
                if cur:
                    self.buckets.append(cur)
                    cur, cur_n = [], 0
                self.buckets.append([(p, 0, n)])
                continue
            if cur_n + n > cap and cur:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur.append((p, cur_n, n))
            cur_n += n
        if cur:
            self.buckets.append(cur)
        self.flat, self.where, self._index = [], {}, {}
        for bi, b in enumerate(self.buckets):
            total = sum(n for _, _, n in b)
            dev, dt = b[0][0].device, torch.float32
            self.flat.append(torch.zeros(total, dtype=dt, device=dev))
            for k, (p, off, n) in enumerate(b):
                self.where[p] = (bi, off, n)
                self._index[p] = k
                p._dp_slice_zero = True
        self.views = [[self.flat[bi][off:off + n].view_as(p) for p, off, n in b] for bi, b in enumerate(self.buckets)]
        # the weight-gradient kernels write straight into the buckets (ops.GRAD_DEST): each backward pass hands out ONE fresh view
        # per parameter (autograd adopts it as .grad, so the pack copy below has nothing to do for it); a second request in the
        # same pass (a shared weight) gets a private tensor and is summed by autograd as usual
        self._taken = set()
        from . import ops
        for bi, b in enumerate(self.buckets):
            if self.inplace_candidate(b, cap):
                continue
            for p, off, n in b:
                if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous():
                    ops.GRAD_DEST[p.data_ptr()] = self._dest(p, bi, off, n)
        self.chunk = cap
        # a bucket that is ONE large tensor (the 78.6 M-element patch-embedding weight) is reduced in place on its gradient: no
        # pack / unpack copies on the tail of the backward pass
        self.inplace = [len(b) == 1 and b[0][2] >= cap for b in self.buckets]
        self.grad_ref = [None] * len(self.buckets)
        self._reset()
        self.handles = []
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)

    @staticmethod
    def inplace_candidate(b, cap):
        return len(b) == 1 and b[0][2] >= cap

    def _dest(self, p, bi, off, n):
        def take():
            # a retained .grad (zero_grad(set_to_none=False)) aliases the slot: autograd would then add the slot to itself
            if p in self._taken or p.grad is not None:
                return None
            self._taken.add(p)
            return self.flat[bi][off:off + n].view_as(p)
        return take

    def close(self):
        """Unregister the bucket slots from the weight-gradient kernels (the hooks stay: a closed reducer must not be used)."""
        from . import ops
        for p in self.params:
            ops.GRAD_DEST.pop(p.data_ptr(), None)

    def _broadcast_state(self, module):
        """Rank 0's parameters and buffers (BatchNorm running statistics) to every rank."""
        from . import ops
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0, group=self.pg)
        ops.invalidate_packs(module.parameters())     # written through .data: the version counters did not move

    def _reset(self):
        self.pending = [len(b) for b in self.buckets]
        self.ready = [False] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.callback_queued = False
        self.work = []
        self._taken = set()

    def _launch(self, bi):
        flat = self.flat[bi]
        if self.inplace[bi] and self.grad_ref[bi] is not None:
            flat = self.grad_ref[bi]
        else:
            # pack the whole bucket with one multi-tensor copy (a copy_ per parameter was ~230 launches and as many Python
            # round trips inside the backward pass: +3 ms per step before any byte moved)
            have = [(v, p.grad) for (p, _, _), v in zip(self.buckets[bi], self.views[bi])
                    if getattr(p, "_dp_has_grad", False) and p.grad.data_ptr() != v.data_ptr()]     # (already in place: written there)
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self.launched[bi] = True
        n = flat.numel()
        for c0 in range(0, n, self.chunk):
            piece = flat[c0:min(n, c0 + self.chunk)]
            if self.backend == "nccl":
                self.work.append(dist.all_reduce(piece, op=dist.ReduceOp.AVG, group=self.pg, async_op=True))
            else:
                self.work.append((dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), piece))

    def _hook(self, p):
        if not self.callback_queued:
            self.callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
        bi, off, n = self.where[p]
        if self.inplace[bi] and p.grad.is_contiguous():
            self.grad_ref[bi] = p.grad.view(-1)
        self.ready_mark(p)
        self.pending[bi] -= 1
        # launch in bucket order so every rank issues the same collective sequence
        while True:
            nxt = next((i for i in range(len(self.buckets)) if not self.launched[i]), None)
            if nxt is None or self.pending[nxt] > 0:
                break
            self._launch(nxt)

    def ready_mark(self, p):
        p._dp_has_grad = True
        p._dp_slice_zero = False

    def _finish(self):
        # parameters without a gradient this step contribute zeros (identical on every rank: same graph)
        for bi, b in enumerate(self.buckets):
            if not self.launched[bi]:
                for p, off, n in b:
                    # (the slice of a parameter that never receives a gradient stays zero from one step to the next)
                    if not getattr(p, "_dp_has_grad", False) and not getattr(p, "_dp_slice_zero", False):
                        self.flat[bi][off:off + n].zero_()
                        p._dp_slice_zero = True
                self._launch(bi)
        for w in self.work:
            if isinstance(w, tuple):
                w[0].wait()
                w[1].div_(self.world)
            else:
                w.wait()
        for bi, b in enumerate(self.buckets):
            for p, off, n in b:
                if getattr(p, "_dp_has_grad", False) and self.grad_ref[bi] is None:
                    # the averaged gradient stays in the bucket: .grad becomes a view of it (no unpack copy); the optimizer
                    # consumes it before the next backward pass refills the bucket
                    p.grad = self.views[bi][self._index[p]]
                p._dp_has_grad = False
            self.grad_ref[bi] = None
        self._reset()


def attach_gradient_allreduce(module, bucket_mb=32.0, process_group=None, broadcast=True):
    """Install the bucketed RCCL gradient exchange on ``module`` in place and return the reducer."""
    return GradAllReducer(module, bucket_mb, process_group, broadcast)
