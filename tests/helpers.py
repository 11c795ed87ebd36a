"""Shared test helpers: golden-fixture loading, the PCG64 weight recipe, error metrics."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" else z[k]) for k in z.files}


def sub(d, prefix):
    """Entries of ``d`` under ``prefix/`` with the prefix stripped."""
    n = len(prefix) + 1
    return {k[n:]: v for k, v in d.items() if k.startswith(prefix + "/")}


def pcg_state_dict(keys, shapes, seed):
    """Rebuild the weights of a PCG64-filled golden model (mirrors make_golden.pcg_fill: numpy PCG64,
    keys consumed in sorted order, scaling rules by key suffix / rank)."""
    shape_of = {k: tuple(int(t) for t in s.split(",") if t != "") for k, s in zip(keys, shapes)}
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    sd = {}
    for k in sorted(shape_of):
        shp = shape_of[k]
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
            continue
        n = int(np.prod(shp)) if len(shp) else 1
        a = rng.standard_normal(n, dtype=np.float32).reshape(shp)
        if k.endswith("running_var"):
            a = np.abs(a) + 0.5
        elif len(shp) == 1 and k.endswith("weight"):
            a = 1 + 0.2 * a
        elif len(shp) == 1 or k.endswith("position_embeddings") or k.endswith("cls_token"):
            a = 0.1 * a
        else:
            a = a * np.float32((1.0 / (n // shp[0])) ** 0.5)
        sd[k] = torch.from_numpy(a.astype(np.float32))
    return {k: sd[k] for k in keys}


def rel_err(a, b):
    """max|a-b| / max|b|  (the north-star's 'rel' on the dose map)."""
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def cmp_prefix(ours, gold):
    """Compare against a possibly trimmed golden gradient (first n flattened elements)."""
    if ours.numel() != gold.numel():
        ours = ours.reshape(-1)[: gold.numel()]
    ours, gold = ours.reshape(-1).double(), gold.reshape(-1).double()
    if gold.norm() < 1e-7 * max(1.0, gold.numel() ** 0.5):
        # analytically-zero gradient (e.g. a conv bias in front of a normalisation): absolute check
        return ours.norm().item() / max(1.0, gold.numel() ** 0.5)
    return rel_l2(ours, gold)


def pcg_tensor(shape, seed, kind="normal"):
    """Inputs that are not stored in fixtures: numpy PCG64 streams (mirrors make_golden.pcg_tensor)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(np.prod(shape))
    a = rng.standard_normal(n, dtype=np.float32) if kind == "normal" else rng.random(n, dtype=np.float32)
    return torch.from_numpy(a.reshape(shape))


def g6_data(shape=(64, 32, 32)):
    """Training batches and validation sample of the G6 trainer fixture (mirrors make_golden.g6_data)."""
    batches = []
    for i in range(4):
        B = 2 if i < 3 else 1
        x = pcg_tensor((B, 3) + shape, 620 + i)
        gt = torch.cat((pcg_tensor((B, 1) + shape, 630 + i, "uniform"), (pcg_tensor((B, 1) + shape, 640 + i, "uniform") > 0.5).float()), 1)
        batches.append({"Input": x, "GT": gt})
    return batches[:3], batches[3]
