#!/usr/bin/env python3
"""Where does the DOSE-PYFER backward pass end?  HIP events (no profiler) at: loss.backward() entered (caller's stream), the first
transformer gradient accumulated (the transformer's LAST layer: its backward starts), the transformer's first layer done + grouped weight
gradients launched (blocks.PatchEmbeddingBlock hook, on the transformer's stream), backward() returned (caller's stream, after the joins),
optimizer.step() done.  Median over the timed steps, ms after the step's start.
    python tools/probes/backward_tail_probe.py [--dtype bf16|fp32x3]
Result (round 6, docs/experiments/r06_adam_overlap.md): the transformer's backward chain IS the tail of the backward pass."""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dose_prediction_amd  # noqa: E402
from dose_prediction_amd import losses, synth  # noqa: E402
from dose_prediction_amd.optim import FusedAdam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()
    args = argparse.Namespace(dtype=a.dtype, loss_scale=1.0, checkpoint=False, no_side_stream=False, no_branch_stream=False, no_wgrad_stream=False,
                              graph_one_stream=False, seg_mode=None, model="pyfer")
    dev = torch.device("cuda:0")
    dose_prediction_amd.config.set_backward_on_calling_thread(True)
    S = (128, 128, 128)
    net = bench.build_model(args, S, dev)
    params = [p for p in net.parameters() if p.requires_grad]
    opt = FusedAdam(params, lr=1e-4, weight_decay=3e-5, amsgrad=True)
    x = synth.dose_input(2, S, seed=1).to(dev)
    gt = synth.dose_target(2, S, seed=2).to(dev)
    vit = [m for m in net.modules() if type(m).__name__ == "ViT"][0]
    marks = {}

    def mark(name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks[name] = ev
    vit.norm.weight.register_post_accumulate_grad_hook(lambda q: mark("vit_bwd_first_grad"))
    blk = vit.blocks[len(vit.blocks) // 2]
    blk.norm1.weight.register_post_accumulate_grad_hook(lambda q: mark("vit_bwd_half"))
    pe = vit.patch_embedding
    lin = pe.patch_embeddings[1] if pe.pos_embed == "perceptron" else pe.patch_embeddings
    lin.weight.register_post_accumulate_grad_hook(lambda q_: mark("vit_bwd_done"))
    # the 128^3 block beside the transformer (ViTEncoder.skip1 = UnetResBlock: conv1 / conv3 are its first layers = last backward nodes)
    enc = [m for m in net.modules() if type(m).__name__ == "ViTEncoder"][0]
    names = dict(enc.skip1.named_parameters())
    for n, q in names.items():
        q.register_post_accumulate_grad_hook(lambda q_, n=n: mark("skip1." + n))
    rows = []
    for i in range(a.steps + 3):
        marks.clear()
        mark("start")
        opt.zero_grad(set_to_none=True)
        loss = losses.gen_loss(net(x), gt, 10.0, 1.0, casecade=True, freez=True)
        mark("forward_done")
        loss.backward()
        mark("backward_returned")
        opt.step()
        mark("step_done")
        torch.cuda.synchronize()
        if i >= 3:
            rows.append({k: marks["start"].elapsed_time(v) for k, v in marks.items()})
    for k in sorted(rows[0], key=lambda k: statistics.median(r[k] for r in rows)):
        print(f"{k:24s} {statistics.median(r[k] for r in rows):7.2f} ms")


if __name__ == "__main__":
    main()
