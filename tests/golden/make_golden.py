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
G6  NetworkTrainer.run() of the reference's own network_trainer.py: one epoch x 3 iterations of c3d.Model with
    Train/loss.py Loss on CPU fp32 -- per-iteration losses, learning rates, validation index, final weights, checkpoint /
    optimizer-state / log structure                                       network_trainer.py:92-125,185-363
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
    # the remaining branches: Huber in train / val mode (loss.py:100-103, 112-115) and Loss(casecade=False) (29-39)
    for p in [pa] + pbs:
        p.grad = None
    l = GenLoss(im_size=S)([pa, pbs], gt, delta1=10, delta2=1, mode="train", casecade=True, freez=True, huber=True)
    l.backward()
    out["gen_huber"] = l
    for i, p in enumerate(pbs):
        out[f"gen_huber/gpb{i}"] = p.grad
        p.grad = None
    l = GenLoss(im_size=S)(pbs[0], gt, mode="val", huber=True)
    l.backward()
    out["gen_val_huber"], out["gen_val_huber/gpb0"] = l, pbs[0].grad
    pbs[0].grad = None
    l = Loss(casecade=False)(pbs[0], gt)
    l.backward()
    out["l1_plain"], out["l1_plain/gpb0"] = l, pbs[0].grad
    save("g5_loss", **out)


# --------------------------------------------------------------------------------------------- G6
def pcg_tensor(shape, seed, kind="normal"):
    """Inputs that are NOT stored: numpy PCG64 streams, regenerated identically by tests/helpers.pcg_tensor."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(np.prod(shape))
    a = rng.standard_normal(n, dtype=np.float32) if kind == "normal" else rng.random(n, dtype=np.float32)
    return torch.from_numpy(a.reshape(shape))


def g6_data(shape=(64, 32, 32)):
    """3 training batches (B=2) + 1 validation sample: Input [B,3,...] normal; GT = (dose uniform[0,1], mask = uniform > 0.5)."""
    batches = []
    for i in range(4):
        B = 2 if i < 3 else 1
        x = pcg_tensor((B, 3) + shape, 620 + i)
        gt = torch.cat((pcg_tensor((B, 1) + shape, 630 + i, "uniform"), (pcg_tensor((B, 1) + shape, 640 + i, "uniform") > 0.5).float()), 1)
        batches.append({"Input": x, "GT": gt})
    return batches[:3], batches[3]


def g6():
    """The reference trainer itself (NetworkTrainer.run) on a tiny C3D cascade, CPU fp32 (the trainer casts inputs with
    .float()).  Everything a drop-in network must reproduce when driven by that trainer is recorded.  Two runs: A = one
    iteration (max_iter=1; after ONE Adam step from identical weights the update is lr*sign(g) wherever |g| >> eps, so the
    weights are sharply defined), B = one epoch of three iterations (loss trajectory, learning rates, validation, bookkeeping)."""
    import shutil
    import tempfile
    from NetworkTrainer.network_trainer import NetworkTrainer
    from DosePrediction.Models.Networks.c3d import Model
    from DosePrediction.Train.loss import Loss
    chans = [-1, 4, 4, 8, 8, 8]
    torch.manual_seed(0)
    net0 = Model(3, 1, chans, chans)
    randomize(net0, 61)
    sd0 = {k: v.clone() for k, v in net0.state_dict().items()}
    batches, val = g6_data()
    xv, gtv = val["Input"], val["GT"]
    base_loss = Loss(casecade=True)

    def run_trainer(max_iter):
        net = Model(3, 1, chans, chans)
        net.load_state_dict(sd0)
        out_dir = tempfile.mkdtemp(prefix="g6_")
        losses = []

        def loss_fn(output, target):
            l = base_loss(output, target)
            losses.append(float(l.detach()))
            return l

        def val_fn(tr):
            with torch.no_grad():
                pred = tr.setting.network(xv.to(tr.setting.device))
            m = gtv[:, 1:2] > 0
            return -float((pred[1][m] - gtv[:, 0:1][m]).abs().mean())

        tr = NetworkTrainer()
        tr.setting.project_name = "g6"
        tr.setting.output_dir = out_dir
        tr.setting.max_epoch = 1
        if max_iter is not None:
            tr.setting.max_iter = max_iter
        tr.setting.train_loader = [{k: v.clone() for k, v in b.items()} for b in batches]
        tr.setting.network = net
        tr.setting.loss_function = loss_fn
        tr.setting.online_evaluation_function_val = val_fn
        tr.setting.lr_scheduler_update_on_iter = True
        tr.set_GPU_device([-1])
        tr.set_optimizer("Adam", {"lr": 1e-3})
        tr.set_lr_scheduler("cosine", {"T_max": 10, "eta_min": 1e-7, "last_epoch": -1})
        tr.run()
        files = sorted(os.listdir(out_dir))
        ck = torch.load(os.path.join(out_dir, "latest.pkl"), map_location="cpu", weights_only=False)
        log_txt = open(os.path.join(out_dir, "log.txt")).read()
        shutil.rmtree(out_dir)
        return ck, losses, files, log_txt, net

    def shadow64(n_iter):
        """float64 shadow of the same sequence (the reference modules in double, plain loop): Adam normalises every gradient by
        its own magnitude, so round-off-level gradients take lr-sized steps of random sign and the fp32 trajectory is only
        defined up to |fp32 - fp64|; the tests take their tolerance from this pair."""
        net64 = Model(3, 1, chans, chans)
        net64.load_state_dict(sd0)
        net64.double().train()
        opt64 = torch.optim.Adam(net64.parameters(), lr=1e-3, weight_decay=3e-5, betas=(0.9, 0.999), eps=1e-08, amsgrad=True)
        sch64 = torch.optim.lr_scheduler.CosineAnnealingLR(opt64, T_max=10, eta_min=1e-7, last_epoch=-1)
        l64 = []
        for b in batches[:n_iter]:
            opt64.zero_grad()
            l = Loss(casecade=True)(net64(b["Input"].double()), b["GT"].double())
            l.backward()
            opt64.step()
            sch64.step()
            l64.append(float(l))
        net64.eval()
        with torch.no_grad():
            pv = net64(xv.double())
        mv = gtv[:, 1:2] > 0
        return net64.state_dict(), l64, -float((pv[1][mv] - gtv[:, 0:1].double()[mv]).abs().mean())

    ckA, lossesA, _, _, _ = run_trainer(1)
    sdA64, _, valA64 = shadow64(1)
    ck, losses, files, log_txt, net = run_trainer(None)
    sd64, l64, val64 = shadow64(3)
    log = ck["log"]
    opt_sd = ck["optimizer_state_dict"]
    log_lines = [l.strip().split("  ")[0][:24] for l in log_txt.splitlines()]
    st0 = opt_sd["state"][0]
    out = dict(
        losses=np.array(losses), val_index=np.array(log.average_val_index), moving_train_loss=np.array(log.moving_train_loss),
        average_train_loss=np.array(log.average_train_loss), log_iter=np.array(log.iter), log_epoch=np.array(log.epoch),
        list_lr=np.array(log.list_lr_associate_iter, dtype=np.float64), end_lr=np.array(opt_sd["param_groups"][0]["lr"]),
        list_train=np.array(log.list_average_train_loss_associate_iter, dtype=np.float64),
        list_val=np.array(log.list_average_val_index_associate_iter, dtype=np.float64),
        files=np.array(files), ckpt_keys=np.array(list(ck.keys())), log_attrs=np.array(sorted(vars(log).keys())),
        opt_group_keys=np.array(sorted(k for k in opt_sd["param_groups"][0] if k != "params")),
        opt_n_params=np.array(len(opt_sd["param_groups"][0]["params"])), opt_state_ids=np.array(sorted(opt_sd["state"].keys())),
        opt_state_keys=np.array(sorted(st0.keys())), opt_step=np.array(float(st0["step"])),
        sched_keys=np.array(sorted(ck["lr_scheduler_state_dict"].keys())), log_lines=np.array(log_lines),
        losses_f64=np.array(l64), val_index_f64=np.array(val64),
        lossA=np.array(lossesA), valA=np.array(ckA["log"].average_val_index), valA_f64=np.array(valA64),
        iterA=np.array(ckA["log"].iter), stepA=np.array(float(ckA["optimizer_state_dict"]["state"][0]["step"])))
    names = [n for n, _ in net.named_parameters()]
    for pid in (sorted(opt_sd["state"].keys())[0], sorted(opt_sd["state"].keys())[-1]):
        for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            out[f"optA/{names[pid]}/{k}"] = ckA["optimizer_state_dict"]["state"][pid][k]
    save("g6_trainer", **out, **pack("sd0", sd0), **pack("sdA", ckA["network_state_dict"]), **pack("sdA_f64", sdA64),
         **pack("sd1", ck["network_state_dict"]), **pack("sd1_f64", sd64))


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
    which = sys.argv[1:] or ["g1", "g2", "g4", "g5", "g6", "g7"]
    for w in which:
        globals()[w]()
