"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI (torch.distributed 'nccl' IS RCCL on
ROCm), bucketed and overlapped with the backward pass.  The reference's only multi-GPU code is nn.DataParallel
(network_trainer.py:104: single process, replicate/scatter/gather through GPU 0); this replaces it.

The reducer is attached IN PLACE (hooks on the parameters, no wrapper module), so ``state_dict()`` keys and the
unchanged ``NetworkTrainer.forward/backward`` (network_trainer.py:185-213) keep working: gradients are averaged by the
time ``loss.backward()`` returns.

Design for 8 x MI355X: gradients are packed in REVERSE registration order (approximately the order autograd produces
them) into flat buckets; a bucket is all-reduced asynchronously as soon as its last gradient arrives, on RCCL's own
stream, while the remaining backward kernels keep the compute stream busy.  Every rank issues the collectives in the
same sequence: bucket index order in the first pass, afterwards the order in which the buckets completed in that pass
(rank 0's, broadcast once) -- a bucket whose gradients arrive late must not hold back the ones behind it.  The patch-embedding weight (78.6 M of
162.6 M elements) is given a bucket of its own, split in chunks, and is reduced in place on its gradient.  The graph is
static: parameters that received no gradient in the FIRST backward pass (MainSubsetModel.out, cls_token, UnetResBlock.conv3
when Cin == Cout) are taken out of the buckets' arrival counts from the second pass on, so that they cannot hold back the
in-order launches; should one of them receive a gradient later, or an expected gradient stay away, the end-of-backward
callback still reduces everything (correct, just without overlap for that bucket).

Exchange algorithm (``algo=`` / env ``DOSE_DDP_ALGO``): ``"allreduce"`` (default) hands every bucket chunk to RCCL as ONE all-reduce and
leaves the algorithm to the library; ``"rs_ag"`` issues it as reduce-scatter + all-gather (``reduce_scatter_tensor`` of the chunk into
this rank's 1/world share, ``all_gather_into_tensor`` of the shares back into the chunk: the two halves of a ring all-reduce as separate
collectives, the form SURVEY 8e asks to A/B over the seven xGMI links of a node).  Same bytes, same averaged result (tests/test_ddp_cpu.py
compares the two on two gloo ranks); behind the switch until an 8-GPU node can measure which one the xGMI mesh prefers.

``grad_dtype=torch.bfloat16`` exchanges bf16 copies of the gradients (half the xGMI bytes: 325 MB instead of 650 MB per step for
DOSE-PYFER); the fp32 ``.grad`` tensors are overwritten with the averaged values afterwards.
"""
import os

import torch
import torch.distributed as dist

_DRY = os.environ.get("DOSE_DDP_DRY", "0") == "1"
_REORDER = os.environ.get("DOSE_DDP_REORDER", "1") == "1"
_TIMING = os.environ.get("DOSE_DDP_TIMING", "0") == "1"       # host seconds spent in the reducer's hooks (self.host_s), diagnosis


def _cur_stream(dev=None):
    """torch.cuda.current_stream(dev) without its ~9 us of Python (ops._current_stream_object: cached by raw handle)."""
    from . import ops
    return ops._current_stream_object(dev)


def _timed(name):
    def deco(fn):
        if not _TIMING:
            return fn
        import functools
        import time

        @functools.wraps(fn)
        def wrapper(self, *a, **k):
            t0 = time.perf_counter()
            try:
                return fn(self, *a, **k)
            finally:
                self.host_s[name] = self.host_s.get(name, 0.0) + time.perf_counter() - t0
        return wrapper
    return deco


ALGOS = ("allreduce", "rs_ag")


class GradAllReducer:
    def __init__(self, module, bucket_mb=32.0, process_group=None, broadcast=True, grad_dtype=torch.float32, algo=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if grad_dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("grad_dtype must be float32, bfloat16 or float16")
        algo = algo or os.environ.get("DOSE_DDP_ALGO", "allreduce")
        if algo not in ALGOS:
            raise ValueError(f"gradient exchange algorithm {algo!r}: expected one of {ALGOS}")
        self.algo = algo
        self._shard = {}             # rs_ag: this rank's share of a chunk, one persistent buffer per (share size, dtype, device)
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.backend = dist.get_backend(process_group)
        self.grad_dtype = grad_dtype
        self.low = grad_dtype != torch.float32
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
            self.flat.append(torch.zeros(total, dtype=grad_dtype, device=b[0][0].device))
            for k, (p, off, n) in enumerate(b):
                self.where[p] = (bi, off, n)
                self._index[p] = k
                p._dp_slice_zero = True
        self.views = [[self.flat[bi][off:off + n].view_as(p) for p, off, n in b] for bi, b in enumerate(self.buckets)]
        self.chunk = cap
        # a bucket that is ONE large tensor (the 78.6 M-element patch-embedding weight) is reduced in place on its gradient: no
        # pack / unpack copies on the tail of the backward pass (fp32 exchange only)
        self.inplace = [(not self.low) and self.inplace_candidate(b, cap) for b in self.buckets]
        # fp32 exchange: the weight-gradient kernels write straight into the buckets (ops.GRAD_DEST): each backward pass hands out
        # ONE fresh view per parameter (autograd adopts it as .grad, so the pack copy has nothing to do for it); a second request in
        # the same pass (a shared weight) gets a private tensor and is summed by autograd as usual
        self._taken = set()
        from . import ops
        if not self.low:
            for bi, b in enumerate(self.buckets):
                if self.inplace[bi]:
                    continue
                for p, off, n in b:
                    if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous():
                        ops.GRAD_DEST[p.data_ptr()] = self._dest(p, bi, off, n)
        self.grad_ref = [None] * len(self.buckets)
        self.host_s = {}
        self.no_grad = None          # parameters without a gradient in the first backward pass (static graph), set by _finish
        self.stats = {"launched_in_backward": 0, "launched_at_end": 0}
        self.launch_order = None     # set after the first backward pass: the order in which its buckets completed
        self._reset()
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
        skip = self.no_grad or ()
        self.pending = [sum(1 for p, _, _ in b if p not in skip) for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self._arrivals, self._last_arrival = 0, [0] * len(self.buckets)
        self.callback_queued = False
        self._cursor = 0
        self._index_order = list(range(len(self.buckets)))
        self.work = []
        self._taken = set()
        self._streams = []
        self._home = None                # (the stream of THIS pass's bucket launches, not of an earlier one)

    @_timed("launch")
    def _launch(self, bi, at_end=False):
        from . import config, ops, streams
        flat = self.flat[bi]
        if not flat.is_cuda:
            self._flush(ops)
            if ops.deferred_pending():
                raise RuntimeError("gradient all-reduce: deferred weight gradients still pending after the flush")
            return self._launch_on_current(bi, at_end)
        cur = ops._current_stream_object(flat.device)
        home = streams.side_stream(flat.device, cur, streams.ROLE_WGRAD) if config.wgrad_stream() else cur
        # The bucket is packed and handed to RCCL from the WEIGHT-GRADIENT stream (config.set_wgrad_stream): that stream already
        # carries most of the bucket's producers, it is off the critical path of the backward pass, and RCCL orders the collective
        # behind the stream it is called from -- so the stream that walks the backward chain never waits for a bucket (it used to
        # wait for the transformer's and the weight-gradient stream at each of the 14 launches: 1-rank RCCL 28 -> 32 ms per step
        # once the weight gradients had moved off it).  Gradients of this bucket produced on other streams: `home` waits for them.
        for s_ in [cur] + self._streams:
            if s_ != home:
                home.wait_stream(s_)
        self._home = home
        if home != cur:
            torch.cuda.set_stream(home)
        try:
            # gradient tensors handed out unwritten (grouped Linear weight gradients) are filled first: flush_deferred() launches
            # EVERY recorded entry (it holds aliases of the handed-out tensors, so it does not depend on which .grad autograd has
            # accumulated so far -- ADVICE r2) and makes the current stream (`home`) wait for that launch
            self._flush(ops)
            if ops.deferred_pending():
                raise RuntimeError("gradient all-reduce: deferred weight gradients still pending after the flush")
            self._launch_on_current(bi, at_end)
        finally:
            if home != cur:
                torch.cuda.set_stream(cur)

    def _launch_on_current(self, bi, at_end):
        flat = self.flat[bi]
        if self.inplace[bi] and self.grad_ref[bi] is not None:
            flat = self.grad_ref[bi]
            if flat.is_cuda:
                flat.record_stream(_cur_stream(flat.device))      # (the in-place exchange runs on the launch stream)
        else:
            # pack the whole bucket with one multi-tensor copy (a copy_ per parameter was ~230 launches and as many Python
            # round trips inside the backward pass: +3 ms per step before any byte moved); converts to grad_dtype on the way
            have = [(v, p.grad) for (p, _, _), v in zip(self.buckets[bi], self.views[bi])
                    if getattr(p, "_dp_has_grad", False) and p.grad.data_ptr() != v.data_ptr()]     # (already in place: written there)
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
                if flat.is_cuda:
                    #  the gradients were allocated on the streams that produced them and are read here on the launch
                    # stream: the caching allocator must not hand their blocks out again before this copy has run
                    from . import ops as _ops
                    cs = _ops._current_stream_object(flat.device)
                    for _, g_ in have:
                        g_.record_stream(cs)
        self.launched[bi] = True
        self.stats["launched_at_end" if at_end else "launched_in_backward"] += 1
        n = flat.numel()
        if _DRY:
            return               # (diagnosis only, DOSE_DDP_DRY=1: everything but the collective itself)
        self._collectives(flat, n)

    @_timed("collectives")
    def _collectives(self, flat, n):
        for c0 in range(0, n, self.chunk):
            piece = flat[c0:min(n, c0 + self.chunk)]
            if self.algo == "rs_ag":
                piece = self._reduce_scatter_all_gather(piece)       # (returns the < world elements it could not split, or None)
                if piece is None:
                    continue
            if self.backend == "nccl":
                self.work.append(dist.all_reduce(piece, op=dist.ReduceOp.AVG, group=self.pg, async_op=True))
            else:
                self.work.append((dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), piece))

    def _reduce_scatter_all_gather(self, piece):
        """One chunk as reduce-scatter + all-gather: the first world * s elements (s = numel // world) are reduced into this rank's
        share [rank * s, (rank + 1) * s) -- a persistent side buffer, so no collective ever reads and writes overlapping memory -- and
        the averaged shares are gathered back into the chunk.  RCCL: both collectives are enqueued asynchronously; they execute in
        issue order on the process group's stream.  gloo (CPU tests): its worker threads may run asynchronous operations out of order,
        so the two halves run synchronously (SUM, then the share is divided: gloo has no AVG)."""
        w = self.world
        s_ = piece.numel() // w
        if s_ == 0:
            return piece
        body = piece[: w * s_]
        key = (s_, piece.dtype, piece.device)
        shard = self._shard.get(key)
        if shard is None:
            shard = self._shard[key] = torch.empty(s_, dtype=piece.dtype, device=piece.device)
        if self.backend == "nccl":
            self.work.append(dist.reduce_scatter_tensor(shard, body, op=dist.ReduceOp.AVG, group=self.pg, async_op=True))
            self.work.append(dist.all_gather_into_tensor(body, shard, group=self.pg, async_op=True))
        else:
            dist.reduce_scatter_tensor(shard, body, op=dist.ReduceOp.SUM, group=self.pg)
            shard.div_(w)
            dist.all_gather_into_tensor(body, shard, group=self.pg)
        return piece[w * s_:] if piece.numel() > w * s_ else None

    @_timed("flush")
    def _flush(self, ops):
        ops.flush_deferred()

    @_timed("hook")
    def _hook(self, p):
        if not self.callback_queued:
            self.callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
        bi, off, n = self.where[p]
        if self.inplace[bi] and p.grad.is_contiguous():
            self.grad_ref[bi] = p.grad.view(-1)
        if p.is_cuda:
            s = _cur_stream(p.device)
            if s not in self._streams:
                self._streams.append(s)
        p._dp_has_grad = True
        p._dp_slice_zero = False
        if self.no_grad is not None and p in self.no_grad:
            return               # unexpected gradient of a parameter that had none in the first pass: handled by _finish
        self.pending[bi] -= 1
        self._arrivals += 1
        self._last_arrival[bi] = self._arrivals          # (the bucket is complete when its last gradient has arrived)
        # every rank must issue the same collective sequence: bucket index order in the first pass, from then on the order in
        # which the buckets completed in that pass (rank 0's, broadcast once).  Index order = reverse registration order is only an
        # approximation of the backward order: the transformer's gradients arrive BEFORE those of skip1, which is registered after
        # it, so in index order every transformer bucket and the 314 MB patch-embedding exchange waited for the very last
        # gradients of the pass (kernel trace: 10 of 12 buckets launched after the backward pass had ended).
        order = self.launch_order if self.launch_order is not None else self._index_order
        c, n = self._cursor, len(order)      # (a cursor instead of a walk from the start: 182 hooks x 14 buckets per pass)
        while c < n:
            i = order[c]
            if not self.launched[i]:
                if self.pending[i] > 0:
                    break
                self._launch(i)
            c += 1
        self._cursor = c

    @_timed("finish")
    def _finish(self):
        first_pass = self.no_grad is None
        if first_pass:
            self.no_grad = {p for p in self.params if not getattr(p, "_dp_has_grad", False)}
        # parameters without a gradient this step contribute zeros (identical on every rank: same graph)
        for bi, b in enumerate(self.buckets):
            late = [p for p, _, _ in b if p in self.no_grad and getattr(p, "_dp_has_grad", False)]
            if self.launched[bi] and late:
                raise RuntimeError("gradient all-reduce: a parameter that had no gradient in the first backward pass received one after "
                                   "its bucket was exchanged; re-attach the reducer (the graph is assumed static)")
            if not self.launched[bi]:
                for p, off, n in b:
                    # (the slice of a parameter that never receives a gradient stays zero from one step to the next)
                    if not getattr(p, "_dp_has_grad", False) and not getattr(p, "_dp_slice_zero", False):
                        self.flat[bi][off:off + n].zero_()
                        p._dp_slice_zero = True
        for bi in (self.launch_order if self.launch_order is not None else range(len(self.buckets))):
            if not self.launched[bi]:
                self._launch(bi, at_end=True)
        if first_pass:
            # launch order of the following passes: completion order of this one (buckets that never completed last), agreed on
            # by all ranks (the graph is static and identical, so the orders should agree anyway; rank 0's is authoritative)
            nb = len(self.buckets)
            seq = sorted(range(nb), key=lambda i: (self._last_arrival[i] if self._last_arrival[i] else 1 << 60, i))
            t = torch.tensor(seq, dtype=torch.int64, device=self.flat[0].device if self.backend == "nccl" else "cpu")
            dist.broadcast(t, 0, group=self.pg)
            self.launch_order = [int(v) for v in t.tolist()] if _REORDER else None
        home = getattr(self, "_home", None)
        if home is not None:
            # (the collectives were issued from the weight-gradient stream: everything it carries is complete for the caller too)
            _cur_stream(home.device).wait_stream(home)
        for w in self.work:
            if isinstance(w, tuple):
                w[0].wait()
                w[1].div_(self.world)
            else:
                w.wait()
        for bi, b in enumerate(self.buckets):
            if self.low:
                # averaged bf16 / fp16 values back into the fp32 gradients (one multi-tensor pass per bucket)
                have = [(p.grad, v) for (p, _, _), v in zip(b, self.views[bi]) if getattr(p, "_dp_has_grad", False)]
                if have:
                    torch._foreach_copy_([g for g, _ in have], [v for _, v in have])
            for p, off, n in b:
                if not self.low and getattr(p, "_dp_has_grad", False) and self.grad_ref[bi] is None:
                    # the averaged gradient stays in the bucket: .grad becomes a view of it (no unpack copy); the optimizer
                    # consumes it before the next backward pass refills the bucket
                    p.grad = self.views[bi][self._index[p]]
                p._dp_has_grad = False
            self.grad_ref[bi] = None
        self._reset()


def attach_gradient_allreduce(module, bucket_mb=32.0, process_group=None, broadcast=True, grad_dtype=torch.float32, algo=None):
    """Install the bucketed RCCL gradient exchange on ``module`` in place and return the reducer.  algo: "allreduce" (default) or
    "rs_ag" (reduce-scatter + all-gather per bucket chunk); None reads env DOSE_DDP_ALGO."""
    return GradAllReducer(module, bucket_mb, process_group, broadcast, grad_dtype, algo)
