"""Generate the golden vectors under tests/golden/*.npz by running the REFERENCE's own modules.

Runs only in the authoring container (needs /root/reference; never on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
The fixtures are data only (inputs, weights, expected outputs / gradients / buffers).

G1  c3d.BaseUNet                     (reference import, no MONAI)         c3d.py:118-149
G2  blocks_MDUNet.conv_3_1           relu/mish x train/eval               blocks_MDUNet.py:132-157
G2o OldModels conv_3_1                                                     OldModels/Nets/blocks_MDUNet.py:132-148
G3  blocks_MDUNet.DualDilatedBlock                                        blocks_MDUNet.py:194-215
G4  c3d.Model (two-U-Net cascade)                                         c3d.py:152-169
G5  Train/loss.py Loss + GenLoss values and gradients                     loss.py:7-41,50-119
G7  dose_pyfer.MainSubsetModel / dose_pyfer.Model / oar_transeg.Model wiring, through the
    test-only MONAI stand-in (tests/golden/monai_shim.py): pins the reference's WIRING only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import monai_shim  # noqa: E402

monai_shim.install()
torch.set_num_threads(8)


def npify(d):
    """The reference modules are run in float64 (``module.double()``) so that the stored vectors carry the
    reference ALGORITHM without fp32 round-off noise; they are stored rounded to float32."""
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            v = v.detach().cpu()
            v = v.float() if v.dtype == torch.float64 else v
            out[k] = v.numpy()
        else:
            out[k] = np.asarray(v)
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **npify(arrs))
    print(f"wrote {name}.npz  {os.path.getsize(path) / 1e6:.2f} MB")


def randomize(module, seed, bn=True):
    """Give every parameter/buffer a non-trivial deterministic value (so affine params, biases and BN
    buffers are all exercised)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if k.endswith("num_batches_tracked"):
                continue
            if k.endswith("running_var"):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif k.endswith("running_mean"):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
            elif v.dim() == 1 and (k.endswith("weight")):
                v.copy_(1 + 0.2 * torch.randn(v.shape, generator=g))
            elif v.dim() == 1 or k.endswith("position_embeddings") or k.endswith("cls_token"):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            else:
                fan_in = v[0].numel()
                v.copy_(torch.randn(v.shape, generator=g) * (1.0 / fan_in) ** 0.5)


def pcg_fill(module, seed):
    """Deterministic, torch-RNG-independent fill for models too large to store (numpy PCG64, keys in
    sorted order).  tests/helpers.py re-implements exactly this to rebuild the weights."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = module.state_dict()
    with torch.no_grad():
        for k in sorted(sd.keys()):
            v = sd[k]
            if k.endswith("num_batches_tracked"):
                continue
            a = rng.standard_normal(v.numel(), dtype=np.float32).reshape(tuple(v.shape))
            if k.endswith("running_var"):
                a = np.abs(a) + 0.5
            elif v.dim() == 1 and k.endswith("weight"):
                a = 1 + 0.2 * a
            elif v.dim() == 1 or k.endswith("position_embeddings") or k.endswith("cls_token"):
                a = 0.1 * a
            else:
                a = a * np.float32((1.0 / v[0].numel()) ** 0.5)
            v.copy_(torch.from_numpy(a.astype(np.float32)))


def grads_of(module, outputs, seed):
    """Back-propagate sum_i <out_i, R_i> with fixed random R_i; return (R list, {name: grad})."""
    g = torch.Generator().manual_seed(seed)
    rs = [torch.randn(o.shape, generator=g).double() for o in outputs]
    loss = sum((o * r).sum() for o, r in zip(outputs, rs))
    module.zero_grad()
    loss.backward()
    return rs, {k: p.grad.clone() for k, p in module.named_parameters() if p.grad is not None}


def keyinfo(net):
    """state_dict key order (as the reference module registers them) and shapes."""
    sd = net.state_dict()
    keys = list(sd.keys())
    return dict(keys=np.array(keys), shapes=np.array([",".join(str(d) for d in sd[k].shape) for k in keys]))


def trim(v, n=4096):
    """Large gradients are stored as their first n elements (flattened); tests compare that prefix."""
    return v.reshape(-1)[:n].clone() if v.numel() > 20000 else v


def pack(prefix, d):
    return {f"{prefix}/{k}": (v.detach().clone() if torch.is_tensor(v) else v) for k, v in d.items()}


def synth_input(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).double()


# --------------------------------------------------------------------------------------------- G1
def g1():
    from DosePrediction.Models.Networks.c3d import BaseUNet
    net = BaseUNet(3, [-1, 4, 8, 8, 16, 16])
    randomize(net, 11)
    net.double()
    net.train()
    # 64x32x32 keeps >= 16 voxels per InstanceNorm group at the deepest level (a 2-voxel group makes the
    # gradient ill-conditioned: fp32 runs of the reference itself then differ from fp64 by > 10 %)
    x = synth_input((1, 3, 64, 32, 32), 12).requires_grad_(True)
    y = net(x)
    rs, grads = grads_of(net, [y], 13)
    save("g1_base_unet", x=x, y=y, r=rs[0], gx=x.grad, **pack("sd", net.state_dict()), **pack("grad", grads))


# --------------------------------------------------------------------------------------------- G2/G3
def g2():
    from OARSegmentation.Models.Nets.blocks_MDUNet import conv_3_1, DualDilatedBlock
    from OARSegmentation.OldModels.Nets.blocks_MDUNet import conv_3_1 as conv_3_1_old
    out = {}
    x0 = synth_input((2, 8, 12, 12, 12), 21)
    for act in ("relu", "mish"):
        blk = conv_3_1(8, 4, act)
        randomize(blk, 22)
        blk.double()
        out.update(pack(f"{act}/sd", blk.state_dict()))
        blk.eval()
        out[f"{act}/y_eval"] = blk(x0)
        blk.train()
        x = x0.clone().requires_grad_(True)
        y = blk(x)
        rs, grads = grads_of(blk, [y], 23)
        out[f"{act}/y_train"], out[f"{act}/r"], out[f"{act}/gx"] = y, rs[0], x.grad
        out.update(pack(f"{act}/grad", grads))
        out.update(pack(f"{act}/sd_after", blk.state_dict()))
    save("g2_conv_3_1", x=x0, **out)

    out = {}
    blk = conv_3_1_old(8, 4)
    randomize(blk, 24)
    blk.double()
    out.update(pack("sd", blk.state_dict()))
    blk.eval()
    out["y_eval"] = blk(x0)
    blk.train()
    x = x0.clone().requires_grad_(True)
    y = blk(x)
    rs, grads = grads_of(blk, [y], 25)
    out["y_train"], out["r"], out["gx"] = y, rs[0], x.grad
    out.update(pack("grad", grads))
    out.update(pack("sd_after", blk.state_dict()))
    save("g2_conv_3_1_old", x=x0, **out)

    out = {}
    for act in ("relu", "mish"):
        blk = DualDilatedBlock(8, 4, act)
        randomize(blk, 26)
        blk.double()
        blk.train()
        x = x0.clone().requires_grad_(True)
        y = blk(x)
        rs, grads = grads_of(blk, [y], 27)
        out[f"{act}/y"], out[f"{act}/r"], out[f"{act}/gx"] = y, rs[0], x.grad
        out.update(pack(f"{act}/sd", blk.state_dict()))
        out.update(pack(f"{act}/grad", grads))
    save("g3_dual_dilated", x=x0, **out)


# --------------------------------------------------------------------------------------------- G4
def g4():
    from DosePrediction.Models.Networks.c3d import Model
    net = Model(3, 1, [-1, 4, 4, 8, 8, 8], [-1, 4, 4, 8, 8, 8])
    randomize(net, 41)
    net.double()
    net.train()
    x = synth_input((1, 3, 32, 32, 16), 42)
    ya, yb = net(x)
    save("g4_c3d_cascade", x=x, ya=ya, yb=yb, **pack("sd", net.state_dict()))


# --------------------------------------------------------------------------------------------- G5
def g5():
    from DosePrediction.Train.loss import Loss, GenLoss
    g = torch.Generator().manual_seed(51)
    S = 32
    gt = torch.cat((torch.rand((2, 1, S, S, S), generator=g), (torch.rand((2, 1, S, S, S), generator=g) > 0.6).float()), 1).double()
    pa = torch.rand((2, 1, S, S, S), generator=g).double().requires_grad_(True)
    pbs = [torch.rand((2, 1, S >> i, S >> i, S >> i), generator=g).double().requires_grad_(True) for i in range(4)]
    out = dict(gt=gt, pa=pa, **{f"pb{i}": p for i, p in enumerate(pbs)})
    for freez in (True, False):
        for p in [pa] + pbs:
            p.grad = None
        l = GenLoss(im_size=S)([pa, pbs], gt, delta1=10, delta2=1, mode="train", casecade=True, freez=freez)
        l.backward()
        tag = f"gen_freez{int(freez)}"
        out[tag] = l
        out[tag + "/gpa"] = pa.grad if pa.grad is not None else torch.zeros_like(pa)
        for i, p in enumerate(pbs):
            out[f"{tag}/gpb{i}"] = p.grad
        for p in [pa] + pbs:
            p.grad = None
        l = Loss(casecade=True)([pa, pbs[0]], gt, freez=freez)
        l.backward()
        tag = f"l1_freez{int(freez)}"
        out[tag] = l
        out[tag + "/gpb0"] = pbs[0].grad
    out["gen_val"] = GenLoss(im_size=S)(pbs[0], gt, mode="val")
    save("g5_loss", **out)


# --------------------------------------------------------------------------------------------- G7
def g7():
    from DosePrediction.Models.Networks.dose_pyfer import MainSubsetModel, Model
    from OARSegmentation.Models.Networks.oar_transeg import Model as Transeg
    from OARSegmentation.OldModels.Networks.oar_transeg import TRANSEG as TransegOld

    # (a) MainSubsetModel, tiny, three decoder variants; weights stored
    for tag, kw in (("multi", dict(mode_multi_dec=True, multiS_conv=True)),
                    ("dual", dict(mode_multi_dec=True, multiS_conv=False)),
                    ("plain", dict(mode_multi_dec=False))):
        net = MainSubsetModel(in_ch=5, out_ch=1, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96,
                              num_heads=6, num_layers=8, act="mish", **kw)
        pcg_fill(net, 71)
        net.double()
        net.train()
        x = synth_input((2, 5, 32, 16, 16), 72).requires_grad_(True)
        outs = net(x)
        rs, grads = grads_of(net, outs, 73)
        grads = {k: trim(v) for k, v in grads.items()}
        save(f"g7_subset_{tag}", x=x, gx=x.grad, seed=71, **keyinfo(net), **{f"y{i}": o for i, o in enumerate(outs)},
             **{f"r{i}": r for i, r in enumerate(rs)}, **pack("grad", grads),
             **pack("sd_after", {k: v for k, v in net.state_dict().items() if "running" in k}))

    # (b) full dose_pyfer.Model (hidden 768 is not a ctor argument): weights from the PCG64 recipe
    net = Model(in_ch=9, out_ch=1, list_ch_A=[-1, 4, 8, 8, 16, 16], feature_size=4, img_size=(32, 16, 16),
                num_layers=4, num_heads=6, act="mish", mode_multi_dec=True, multiS_conv=True)
    pcg_fill(net, 74)
    net.double()
    x = synth_input((1, 9, 32, 16, 16), 75)
    net.eval()          # eval first: the training-mode forward below updates the BN running buffers
    ya_e, ybs_e = net(x)
    net.train()
    ya, ybs = net(x)
    save("g7_pyfer_model", x=x, ya=ya, ya_eval=ya_e, **{f"yb{i}": o for i, o in enumerate(ybs)},
         **{f"yb{i}_eval": o for i, o in enumerate(ybs_e)}, seed=74, **keyinfo(net))

    # (c) OAR-TRANSEG (Models and OldModels), tiny
    for tag, cls in (("new", Transeg), ("old", TransegOld)):
        net = cls(in_channels=1, out_channels=8, img_size=(32, 16, 16), feature_size=4, hidden_size=48, mlp_dim=96,
                  num_heads=12, pos_embed="perceptron", norm_name="instance", res_block=True, conv_block=True,
                  dropout_rate=0.0)
        pcg_fill(net, 76)
        net.double()
        net.train()
        x = synth_input((2, 1, 32, 16, 16), 77).requires_grad_(True)
        y = net(x)
        rs, grads = grads_of(net, [y], 78)
        grads = {k: trim(v) for k, v in grads.items()}
        save(f"g7_transeg_{tag}", x=x, y=y, r=rs[0], gx=x.grad, seed=76, **keyinfo(net), **pack("grad", grads),
             **pack("sd_after", {k: v for k, v in net.state_dict().items() if "running" in k}))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g4", "g5", "g7"]
    for w in which:
        globals()[w]()
