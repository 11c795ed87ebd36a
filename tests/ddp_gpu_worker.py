"""Worker of test_models_gpu.py::test_gradient_allreduce_two_ranks_matches_the_mean_of_local_gradients (started through
torch.distributed.run, gloo backend, both ranks on the one GPU of the test box).  Every rank computes BOTH ranks' local gradients
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
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
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
    red = attach_gradient_allreduce(net, bucket_mb=float(os.environ.get("DDP_TEST_BUCKET_MB", "0.01")))
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
                assert e < 5e-3, (rank, step, k, e)
        if step >= 1:
            assert red.stats["launched_at_end"] <= len(red.buckets), red.stats
    assert red.stats["launched_in_backward"] >= 2 * (len(red.buckets) - 1), (red.stats, len(red.buckets))
    red.close()
    dist.barrier()
    if rank == 0:
        print(f"DDP_GPU_WORKER_OK buckets={len(red.buckets)} split_pairs={split} worst_rel={worst:.2e} stats={red.stats}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
