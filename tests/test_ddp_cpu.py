"""CPU (gloo, world_size 2): the bucketed gradient all-reduce averages gradients across ranks, tolerates parameters that
receive no gradient, broadcasts rank 0's parameters/buffers, and leaves state_dict keys untouched."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 300)
        self.bn = nn.BatchNorm1d(300)
        self.b = nn.Linear(300, 2000)      # larger than one (tiny) bucket -> chunked bucket of its own
        self.unused = nn.Linear(4, 4)      # never used in forward (like MainSubsetModel.out)
        self.frozen = nn.Linear(8, 8)
        for p in self.frozen.parameters():
            p.requires_grad = False

    def forward(self, x):
        return self.b(torch.relu(self.bn(self.a(self.frozen(x))))).sum()


def _worker(rank, world, port, out, grad_dtype=torch.float32, algo=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dose_prediction_amd.ddp import attach_gradient_allreduce
    torch.manual_seed(100 + rank)          # different initial weights per rank: the broadcast must fix that
    net = Net()
    keys_before = list(net.state_dict().keys())
    red = attach_gradient_allreduce(net, bucket_mb=0.002, grad_dtype=grad_dtype, algo=algo)
    assert red.algo == (algo or "allreduce")
    assert list(net.state_dict().keys()) == keys_before
    w0 = net.a.weight.detach().clone()
    res = {"w0": w0}
    for step in range(2):                  # two steps: bucket state must reset
        net.zero_grad(set_to_none=True)
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(7 + rank + 10 * step))
        before = dict(red.stats)
        net(x).backward()
        res[f"g{step}"] = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        res[f"x{step}"] = x
        res[f"launched{step}"] = (red.stats["launched_in_backward"] - before["launched_in_backward"],
                                  red.stats["launched_at_end"] - before["launched_at_end"], len(red.buckets))
    assert {k for k, p in net.named_parameters() if p in red.no_grad} == {"unused.weight", "unused.bias"}
    # a step in which parameters that HAD gradients get none: their bucket slices must be exchanged as zeros again
    net.zero_grad(set_to_none=True)
    net.a(net.frozen(x)).sum().backward()
    named = dict(net.named_parameters())
    for name in ("b.weight", "bn.weight", "unused.weight"):
        bi, off, n = red.where[named[name]]
        assert not red.flat[bi][off:off + n].any(), name
        assert named[name].grad is None, name
    assert named["a.weight"].grad is not None
    res["sd"] = {k: v.clone() for k, v in net.state_dict().items()}
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert torch.equal(r0["w0"], r1["w0"])                      # broadcast
    torch.manual_seed(100)
    ref = Net()
    for step in range(2):
        gs = []
        for r in (r0, r1):
            m = Net()
            m.load_state_dict(ref.state_dict())
            if step == 1:
                m.train()
                m(r["x0"])                                      # replay step 0 to advance the BN running stats identically
                m.zero_grad()
            m(r[f"x{step}"]).backward()
            gs.append({k: p.grad for k, p in m.named_parameters() if p.grad is not None})
        for k in gs[0]:
            avg = 0.5 * (gs[0][k] + gs[1][k])
            assert torch.allclose(r0[f"g{step}"][k], avg, rtol=1e-5, atol=1e-6), (step, k)
            assert torch.equal(r0[f"g{step}"][k], r1[f"g{step}"][k])
        assert "unused.weight" not in r0[f"g{step}"] and "frozen.weight" not in r0[f"g{step}"]
    # overlap: `unused` is registered last, i.e. it sits in bucket 0, and never receives a gradient.  In the first pass
    # the reducer cannot know that (bucket 0 waits until the end-of-backward callback, and every later bucket queues behind it);
    # from the second pass on the gradient-less parameters are known and EVERY bucket is exchanged from the hooks, inside backward
    in_bwd, at_end, nb = r0["launched1"]
    assert nb >= 3 and in_bwd == nb and at_end == 0, r0["launched1"]
    assert r0["launched0"][1] >= 1


class OutOfOrderNet(nn.Module):
    """Registration order first, second; the forward pass runs `second` BEFORE `first` (like DOSE-PYFER, whose transformer is
    registered before skip1 but enqueued after it), so the backward pass delivers first's gradients before second's -- the opposite
    of reverse registration order."""

    def __init__(self):
        super().__init__()
        self.first = nn.Linear(8, 600)
        self.second = nn.Linear(8, 600)
        self.head = nn.Linear(600, 4)

    def forward(self, x):
        s = torch.tanh(self.second(x))
        f = torch.tanh(self.first(x))
        return self.head(f * s).sum()


def _worker_order(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dose_prediction_amd.ddp import attach_gradient_allreduce
    torch.manual_seed(5)
    net = OutOfOrderNet()
    red = attach_gradient_allreduce(net, bucket_mb=0.01)
    res = {}
    for step in range(3):
        net.zero_grad(set_to_none=True)
        x = torch.randn(6, 8, generator=torch.Generator().manual_seed(20 + rank + 10 * step))
        before = dict(red.stats)
        net(x).backward()
        res[f"g{step}"] = {k: p.grad.clone() for k, p in net.named_parameters()}
        res[f"x{step}"] = x
        res[f"launched{step}"] = (red.stats["launched_in_backward"] - before["launched_in_backward"],
                                  red.stats["launched_at_end"] - before["launched_at_end"], len(red.buckets))
    res["order"] = list(red.launch_order)
    res["where"] = {k: red.where[p][0] for k, p in net.named_parameters()}
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_buckets_launch_in_the_order_the_backward_pass_completes_them():
    """From the second pass on the buckets are exchanged in the order in which they completed in the first (rank 0's order,
    broadcast): a bucket whose gradients arrive late no longer holds back the ones behind it in index order."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_order, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["order"] == r1["order"] and sorted(r0["order"]) == list(range(r0["launched0"][2]))
    # index order would exchange second's bucket(s) before first's; the recorded order has first's bucket(s) earlier
    pos = {b: i for i, b in enumerate(r0["order"])}
    assert pos[r0["where"]["first.weight"]] < pos[r0["where"]["second.weight"]]
    assert r0["where"]["first.weight"] > r0["where"]["second.weight"]          # (reverse registration order = index order)
    for step in (1, 2):
        in_bwd, at_end, nb = r0[f"launched{step}"]
        assert in_bwd == nb and at_end == 0, (step, r0[f"launched{step}"])
    ref = OutOfOrderNet()
    torch.manual_seed(5)
    ref = OutOfOrderNet()
    for step in range(3):
        gs = []
        for r in (r0, r1):
            ref.zero_grad()
            ref(r[f"x{step}"]).backward()
            gs.append({k: p.grad.clone() for k, p in ref.named_parameters()})
        for k in gs[0]:
            assert torch.allclose(r0[f"g{step}"][k], 0.5 * (gs[0][k] + gs[1][k]), rtol=1e-5, atol=1e-6), (step, k)
            assert torch.equal(r0[f"g{step}"][k], r1[f"g{step}"][k])


def test_bf16_gradient_buckets_gloo():
    """grad_dtype=torch.bfloat16 halves the exchanged bytes; the averaged gradients equal the fp32 exchange up to one bf16
    rounding of each rank's contribution (2^-9 relative), are identical on both ranks, and .grad stays fp32."""
    world = 2
    mgr = mp.Manager()
    out16, out32 = mgr.dict(), mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out16, torch.bfloat16), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), out32), nprocs=world, join=True)
    for step in range(2):
        for k, g32 in out32[0][f"g{step}"].items():
            g16 = out16[0][f"g{step}"][k]
            assert g16.dtype == torch.float32
            assert torch.equal(g16, out16[1][f"g{step}"][k])
            assert (g16 - g32).abs().max() <= 2 ** -7 * g32.abs().max() + 1e-12, (step, k)


def test_reduce_scatter_all_gather_exchange_equals_the_allreduce():
    """DOSE_DDP_ALGO=rs_ag ("direct reduce-scatter + all-gather across the 7 xGMI links"): every bucket chunk
    as reduce_scatter_tensor + all_gather_into_tensor instead of one all_reduce.  Two gloo ranks, the chunked bucket (b.weight, larger than
    a bucket) and odd-sized buckets included (a share that does not divide by the world size leaves a tail that is all-reduced): the
    averaged gradients are identical on both ranks and equal to the all-reduce path's (fp32: the same two addends in either form, so
    bit-identical; bf16 buckets: identical too -- one rounding of each rank's contribution, one sum), over two steps and the
    gradient-less-parameter step; the overlap bookkeeping (every bucket exchanged inside backward from the second pass) is unchanged."""
    world = 2
    mgr = mp.Manager()
    for dt in (torch.float32, torch.bfloat16):
        out_ar, out_rs = mgr.dict(), mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out_ar, dt, "allreduce"), nprocs=world, join=True)
        mp.spawn(_worker, args=(world, _free_port(), out_rs, dt, "rs_ag"), nprocs=world, join=True)
        for step in range(2):
            assert set(out_rs[0][f"g{step}"]) == set(out_ar[0][f"g{step}"])
            for k, g_ar in out_ar[0][f"g{step}"].items():
                g_rs = out_rs[0][f"g{step}"][k]
                assert torch.equal(g_rs, out_rs[1][f"g{step}"][k]), (dt, step, k)
                assert torch.equal(g_rs, g_ar), (dt, step, k, float((g_rs - g_ar).abs().max()))
        assert out_rs[0]["launched1"] == out_ar[0]["launched1"]


def test_unknown_exchange_algorithm_is_rejected():
    import pytest
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from dose_prediction_amd.ddp import attach_gradient_allreduce
        with pytest.raises(ValueError):
            attach_gradient_allreduce(nn.Linear(2, 2), algo="ring")
        red = attach_gradient_allreduce(nn.Linear(2, 2), algo="rs_ag")          # world size 1: share == chunk
        assert red.algo == "rs_ag"
    finally:
        dist.destroy_process_group()
