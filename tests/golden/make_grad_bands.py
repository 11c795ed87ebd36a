#!/usr/bin/env python3
"""Round-off bands of the golden networks' BACKWARD passes -> tests/golden/grad_bands.json.

Why: the golden gradient gates of tests/test_models_gpu.py / test_x3_gpu.py run in BOTH reduction modes since round 6.  With
fp32 atomics the order of additions differs from run to run, so a gate needs the tolerance a CORRECT evaluation of this network in this
arithmetic can be expected to need -- not a blanket number and not a retry.  That tolerance is computed here, once, from the CPU oracle
(this script never touches the reference and never runs on the GPU box): the float64 oracle's backward pass with one rounding of the
mode's size injected wherever the HIP path stores a tensor (oracle.grad_noise), over SEEDS seeds, measured with the very metrics the GPU
tests apply to the HIP gradients (worst parameter gradient against max(|gold|, 5 % of the median gradient norm); input gradient relative L2):

  mode "fp32"   exact-fp32 arithmetic: relative 2^-24 on every stored forward value and gradient
  mode "x3"     the fp32x3 mode with its one-product backward: forward values 2^-17 (a [hi | lo] bf16 pair), stored gradients rounded to
                8 significand bits after a 2^-24 perturbation (bf16 operands of the data / weight gradient products)

Every entry holds the LARGEST value over the seeds; a gate uses max(floor, 2 x band).  Re-run after changing a golden fixture:
    python tests/golden/make_grad_bands.py"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from helpers import load_golden, sub, pcg_state_dict, cmp_prefix  # noqa: E402

torch.set_num_threads(8)
SEEDS = 8
MODES = {"fp32": dict(eps=2.0 ** -24, bits=None, fwd_eps=2.0 ** -24), "x3": dict(eps=2.0 ** -24, bits=8, fwd_eps=2.0 ** -17)}


def leafify(sd):
    out = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            v = v.double()
            out[k] = v.clone().requires_grad_(True) if "running" not in k else v
        else:
            out[k] = v
    return out


def worst_param(sd, gold):
    """tests/test_models_gpu.py::_check_grads' metric."""
    norms = sorted(float(g.double().norm()) for g in gold.values())
    floor = 5e-2 * norms[len(norms) // 2]
    worst = 0.0
    for k, g in gold.items():
        ours = sd[k].grad.detach().reshape(-1)[: g.numel()].double()
        worst = max(worst, float((ours - g.reshape(-1).double()).norm()) / max(float(g.double().norm()), floor))
    return worst


def cases():
    g = load_golden("g1_base_unet")
    yield "g1_base_unet", sub(g, "sd"), "", g["x"], lambda sd, x: [oracle.base_unet(sd, "", x)], [g["r"]], g["gx"], sub(g, "grad")
    g = load_golden("g2_conv_3_1")
    for act in ("relu", "mish"):
        yield (f"g2_conv_3_1/{act}", sub(g, f"{act}/sd"), "b.", g["x"], lambda sd, x, act=act: [oracle.conv_3_1(sd, "b", x, act, training=True)],
               [g[f"{act}/r"]], g[f"{act}/gx"], sub(g, f"{act}/grad"))
    g = load_golden("g2_conv_3_1_old")
    yield "g2_conv_3_1_old", sub(g, "sd"), "b.", g["x"], lambda sd, x: [oracle.conv_3_1_old(sd, "b", x, training=True)], [g["r"]], g["gx"], sub(g, "grad")
    g = load_golden("g3_dual_dilated")
    for act in ("relu", "mish"):
        yield (f"g3_dual_dilated/{act}", sub(g, f"{act}/sd"), "b.", g["x"], lambda sd, x, act=act: [oracle.dual_dilated_block(sd, "b", x, act)],
               [g[f"{act}/r"]], g[f"{act}/gx"], sub(g, f"{act}/grad"))
    for tag, kw in (("multi", dict(mode_multi_dec=True, multiS_conv=True)), ("dual", dict(mode_multi_dec=True, multiS_conv=False)),
                    ("plain", dict(mode_multi_dec=False))):
        g = load_golden(f"g7_subset_{tag}")
        yield (f"g7_subset_{tag}", pcg_state_dict(g["keys"], g["shapes"], g["seed"]), "", g["x"],
               lambda sd, x, kw=kw: oracle.main_subset_model(sd, "", x, 8, 6, "mish", True, **kw), [g[f"r{i}"] for i in range(4)], g["gx"], sub(g, "grad"))
    for tag in ("new", "old"):
        g = load_golden(f"g7_transeg_{tag}")
        yield (f"g7_transeg_{tag}", pcg_state_dict(g["keys"], g["shapes"], g["seed"]), "", g["x"],
               lambda sd, x, tag=tag: [oracle.oar_transeg(sd, x, num_heads=12, training=True, old=(tag == "old"))], [g["r"]], g["gx"], sub(g, "grad"))


def main():
    out = {"_doc": "largest value over %d seeds of (worst parameter gradient, input gradient rel-L2) of the float64 oracle under oracle.grad_noise; "
                   "see tests/golden/make_grad_bands.py" % SEEDS, "_seeds": SEEDS}
    for name, sd0, pre, x0, fn, rs, gx, gold in cases():
        gold = {pre + k: v for k, v in gold.items()}
        ent = {}
        for mode, cfg in MODES.items():
            wp = wx = 0.0
            for seed in range(SEEDS):
                sd = leafify({pre + k: v for k, v in sd0.items()})
                x = x0.double().requires_grad_(True)
                with oracle.grad_noise(cfg["eps"], seed, bits=cfg["bits"], fwd_eps=cfg["fwd_eps"]):
                    outs = fn(sd, x)
                    torch.autograd.backward(list(outs), [r.double() for r in rs])
                wp = max(wp, worst_param(sd, gold))
                wx = max(wx, cmp_prefix(x.grad, gx))
            ent[mode] = {"param": wp, "gx": wx}
            print(f"{name:24s} {mode:5s} worst parameter gradient {wp:.3e}   input gradient {wx:.3e}", flush=True)
        out[name] = ent
    with open(os.path.join(HERE, "grad_bands.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
