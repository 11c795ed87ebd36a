"""Worker of test_models_gpu.py::test_gradient_allreduce_two_ranks_matches_the_mean_of_local_gradients (started through
torch.distributed.run).  Backend: RCCL ("nccl") with one GPU per rank whenever the box has at least WORLD_SIZE GPUs -- so a 2- or 8-GPU
`pytest -m gpu` checks RCCL's gradients against the mean of the local gradients with no further work -- and gloo with
every rank on the box's one GPU otherwise (DDP_TEST_BACKEND overrides).  DDP_TEST_GRAD_DTYPE=bf16 exchanges bf16 buckets; with the tiny
DDP_TEST_BUCKET_MB every tensor above the cap (the patch-embedding weight, 983 k elements) takes the in-place chunked AVG path.  Every rank computes BOTH ranks' local gradients
without the reducer (expected = their mean), then attaches the bucketed reducer with bucket boundaries that separate Linear weights
from their biases and checks three backward passes (the launch order changes after the first) against the expectation."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    from helpers import load_golden, pcg_state_dict
    import dose_prediction_amd
    from dose_prediction_amd import ops
    from dose_prediction_amd.ddp import attach_gradient_allreduce
    from dose_prediction_amd.models.dose_pyfer import MainSubsetModel
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    backend = os.environ.get("DDP_TEST_BACKEND") or ("nccl" if torch.cuda.device_count() >= world else "gloo")
    idx = local if backend == "nccl" else 0
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    gdt = {"fp32": torch.float32, "bf16": torch.bfloat16}[os.environ.get("DDP_TEST_GRAD_DTYPE", "fp32")]
    tol = 5e-3 if gdt == torch.float32 else 2e-2
    dose_prediction_amd.set_compute_dtype(torch.float32)
    g = load_golden("g7_subset_multi")

    def build():
        net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96, num_heads=6,
                              num_layers=8, act="mish", mode_multi_dec=True, multiS_conv=True)
        sd = pcg_state_dict(g["keys"], g["shapes"], g["seed"])
        net.load_state_dict(sd)
        return net.to(dev).train()

    def local(net, r, step):
        net.zero_grad(set_to_none=True)
        x = g["x"].to(dev) * (1.0 + 0.25 * r + 0.1 * step)
        sum((o * o).mean() for o in net(x)).backward()
        torch.cuda.synchronize()
        return {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in net.named_parameters()}

    net = build()
    want = []
    for step in range(3):
        per = [local(net, r, step) for r in range(world)]
        want.append({k: (None if per[0][k] is None else sum(p[k] for p in per) / world) for k in per[0]})
    net = build()
    red = attach_gradient_allreduce(net, bucket_mb=float(os.environ.get("DDP_TEST_BUCKET_MB", "0.01")), grad_dtype=gdt)
    assert gdt != torch.float32 or any(red.inplace), "no bucket takes the in-place chunked exchange"
    # the bucket layout must separate at least one deferred Linear weight from its bias (the case ADVICE r2 describes)
    names = {p: k for k, p in net.named_parameters()}
    split = 0
    for k, p in net.named_parameters():
        if k.endswith(".weight") and p in red.where and k[:-6] + "bias" in dict(net.named_parameters()):
            b = dict(net.named_parameters())[k[:-6] + "bias"]
            if b in red.where and red.where[b][0] != red.where[p][0] and ("linear" in k or "out_proj" in k or "patch_embeddings" in k):
                split += 1
    assert split >= 1, "bucket size does not separate any Linear weight from its bias"
    worst = 0.0
    for step in range(3):
        got = local(net, rank, step)
        assert ops.deferred_pending() == 0
        norms = sorted(float(r.double().norm()) for r in want[step].values() if r is not None)
        floor = 5e-2 * norms[len(norms) // 2]
        for k, r in want[step].items():
            o = got[k]
            assert (r is None) == (o is None), (step, k)
            if r is not None:
                e = float((o.double() - r.double()).norm()) / max(float(r.double().norm()), floor)
                worst = max(worst, e)
                assert e < tol, (rank, step, k, e)
        if step >= 1:
            assert red.stats["launched_at_end"] <= len(red.buckets), red.stats
    assert red.stats["launched_in_backward"] >= 2 * (len(red.buckets) - 1), (red.stats, len(red.buckets))
    red.close()
    dist.barrier()
    if rank == 0:
        print(f"DDP_GPU_WORKER_OK backend={backend} grad_dtype={gdt} inplace_buckets={sum(red.inplace)} buckets={len(red.buckets)} split_pairs={split} worst_rel={worst:.2e} stats={red.stats}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
