"""Drop-in boundary, trainer level: a network driven through NetworkTrainer's call sequence
(tests/trainer_replay.py) must reproduce what the REFERENCE trainer + reference c3d.Model + reference Loss produced
(tests/golden/g6_trainer.npz, from the real network_trainer.py): per-iteration losses, learning rates, validation index,
final weights, Adam moments, checkpoint / optimizer-state / log structure, and the checkpoint must round-trip.

CPU: the oracle (wrapped as an nn.Module) pins harness + oracle against the fixture.
GPU: the HIP-backed c3d.Model with torch.optim.Adam (what the unchanged trainer builds) and with FusedAdam."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle
from helpers import load_golden, sub, rel_err, g6_data
from trainer_replay import TrainerReplay


class OracleC3D(nn.Module):
    """The functional oracle behind an nn.Module so that the harness can drive it (parameters named as in the reference)."""

    def __init__(self, sd):
        super().__init__()
        self.keys = list(sd.keys())
        self.params = nn.ParameterList([nn.Parameter(sd[k].clone().float()) for k in self.keys])

    def forward(self, x):
        return oracle.c3d_model(dict(zip(self.keys, self.params)), x)

    def state_dict(self, *a, **k):
        return {k_: p.detach().clone() for k_, p in zip(self.keys, self.params)}


def _batches():
    return [{k: v.clone() for k, v in b.items()} for b in g6_data()[0]]


def _val_fn(to_dev):
    val = g6_data()[1]
    xv, gtv = val["Input"], val["GT"]

    def f(tr):
        with torch.no_grad():
            pred = tr.network(to_dev(xv))
        m = gtv[:, 1:2] > 0
        return -float((pred[1].float().cpu()[m] - gtv[:, 0:1][m]).abs().mean())
    return f


# conv biases in front of InstanceNorm have an analytically zero gradient: Adam turns their round-off gradients into a +-lr random
# walk that no two runs share (and they cannot influence the output), so they are left out of the weight comparison
_NOISE_ONLY = ("single_conv.0.bias", "conv.0.bias")


def _update_error(net_sd, g, after, slack):
    """Distance of our weights from the reference trainer's (fp32), as a fraction of the update the reference made, against the
    same distance for the reference's own float64 run."""
    sd0, ref, ref64 = sub(g, "sd0"), sub(g, after), sub(g, after + "_f64")
    assert list(net_sd.keys()) == list(ref.keys())
    keys = [k for k in ref if ref[k].dtype.is_floating_point and not k.endswith(_NOISE_ONLY)]
    cat = lambda d: torch.cat([d[k].double().reshape(-1).cpu() for k in keys])  # noqa: E731
    ours, r32, r64, w0 = cat(net_sd), cat(ref), cat(ref64), cat(sd0)
    upd = (r32 - w0).norm()
    e_ours, e_64 = ((ours - r32).norm() / upd).item(), ((r64 - r32).norm() / upd).item()
    assert e_ours <= slack * e_64 + 1e-3, (after, e_ours, e_64)
    return e_ours, e_64


def _check_one_step(tr, g, out_dir, floor, slack=2.5, moment_tol=2e-2):
    """Run A of the fixture (max_iter = 1): after ONE Adam step from identical weights the update is lr*sign(g) wherever
    |g| >> eps, so the weights are sharply defined (the reference's own fp32 / fp64 runs differ by 9 % of the update norm: sign
    flips of round-off-level gradients); weights that did not move at all would be 100 % off."""
    assert len(tr.losses) == 1 and abs(tr.losses[0] - float(g["lossA"][0])) <= floor * 10 * abs(float(g["lossA"][0]))
    ck = torch.load(os.path.join(out_dir, "latest.pkl"), map_location="cpu", weights_only=False)
    assert ck["log"].iter == int(g["iterA"])
    e_ours, e_64 = _update_error(ck["network_state_dict"], g, "sdA", slack)
    assert e_64 < 0.15 and e_ours < 0.3
    band = slack * abs(float(g["valA_f64"]) - float(g["valA"])) + 2e-4
    assert abs(tr.log.average_val_index - float(g["valA"])) <= band, (tr.log.average_val_index, float(g["valA"]), band)
    osd = ck["optimizer_state_dict"]
    assert float(osd["state"][sorted(osd["state"].keys())[0]]["step"]) == float(g["stepA"])
    names = list(sub(g, "sd0").keys())
    for key in [k for k in g if k.startswith("optA/")]:       # Adam moments after one step: (1-beta) g and (1-beta2) g^2
        _, pname, which = key.split("/")
        got = osd["state"][names.index(pname)][which]
        assert rel_err(got.float().cpu(), g[key]) < moment_tol, key       # (fp32 gradients of the 16-voxel InstanceNorm levels carry ~3e-3 noise)


def _check_against_g6(tr, g, out_dir, floor, slack=2.5):
    """Run B (one epoch, three iterations).  Tolerances come from the fixture itself: Adam divides every gradient by its own
    magnitude, so round-off-level gradients take lr-sized steps of random sign and the reference's fp32 trajectory is only defined
    up to its distance from the reference's own float64 run of the same sequence (losses_f64, sd1_f64).  A path is accepted when
    it is within `slack` x that distance (+ a round-off floor) of the reference trainer's fp32 numbers; the loss decrease over
    the three iterations (0.042) is ~10x larger than the widest of these bands."""
    ref, ref64 = g["losses"].numpy(), g["losses_f64"].numpy()
    for i, l in enumerate(tr.losses):
        assert abs(l - ref[i]) <= slack * abs(ref64[i] - ref[i]) + floor * abs(ref[i]), (i, l, ref[i], ref64[i])
    assert ref[0] - tr.losses[2] > 0.03                    # it trains: the reference goes 0.572 -> 0.531
    band = slack * max(abs(float(g["val_index_f64"]) - float(g["val_index"])), np.abs(ref64 - ref).max()) + floor
    assert abs(tr.log.average_val_index - float(g["val_index"])) <= band
    assert abs(tr.log.moving_train_loss - float(g["moving_train_loss"])) <= slack * np.abs(ref64 - ref).max() + floor
    assert tr.log.iter == int(g["log_iter"]) and tr.log.epoch == int(g["log_epoch"])
    assert np.allclose(np.array(tr.log.list_lr_associate_iter, dtype=np.float64), g["list_lr"].numpy())
    assert abs(tr.optimizer.param_groups[0]["lr"] - float(g["end_lr"])) < 1e-12
    assert np.allclose(np.array(tr.log.list_average_train_loss_associate_iter)[:, 1], g["list_train"].numpy()[:, 1])
    # files, checkpoint keys, log attributes: exactly the reference trainer's
    assert sorted(os.listdir(out_dir)) == list(g["files"])
    ck = torch.load(os.path.join(out_dir, "latest.pkl"), map_location="cpu", weights_only=False)
    assert list(ck.keys()) == list(g["ckpt_keys"])
    assert sorted(vars(ck["log"]).keys()) == list(g["log_attrs"])
    assert sorted(ck["lr_scheduler_state_dict"].keys()) == list(g["sched_keys"])
    osd = ck["optimizer_state_dict"]
    assert set(g["opt_group_keys"]) - {"capturable", "decoupled_weight_decay", "differentiable", "foreach", "fused", "maximize"} \
        <= set(osd["param_groups"][0])
    assert len(osd["param_groups"][0]["params"]) == int(g["opt_n_params"])
    assert sorted(osd["state"].keys()) == [int(i) for i in g["opt_state_ids"]]
    st0 = osd["state"][sorted(osd["state"].keys())[0]]
    assert set(g["opt_state_keys"]) == set(st0.keys())
    assert float(st0["step"]) == float(g["opt_step"])
    _update_error(ck["network_state_dict"], g, "sd1", slack)
    # log.txt: same line structure (digits masked: times and timings differ)
    mask = lambda s: "".join("#" if c.isdigit() else c for c in s)  # noqa: E731
    skip = ("Train", "Val", "Total", "time", "End lr", "Local")
    lines = [mask(l.strip().split("  ")[0][:24]) for l in open(os.path.join(out_dir, "log.txt")).read().splitlines()]
    ref_lines = [mask(str(l)) for l in g["log_lines"]]
    assert [l for l in lines if not l.startswith(skip)] == [l for l in ref_lines if not l.startswith(skip)]
    return ck


def test_g6_oracle_through_the_trainer_sequence(tmp_path):
    g = load_golden("g6_trainer")
    sd0 = sub(g, "sd0")
    torch.set_num_threads(8)
    mk = lambda d, **kw: TrainerReplay(OracleC3D(sd0), torch.device("cpu"), lambda o, t: oracle.loss_l1_masked(o, t),  # noqa: E731
                                       _val_fn(lambda t: t), _batches(), d, **kw)
    os.makedirs(tmp_path / "a")
    os.makedirs(tmp_path / "b")
    tr = mk(str(tmp_path / "a"), max_iter=1)
    tr.run()
    _check_one_step(tr, g, str(tmp_path / "a"), 1e-6)
    tr = mk(str(tmp_path / "b"))
    tr.run()
    _check_against_g6(tr, g, str(tmp_path / "b"), 1e-6)


@pytest.fixture
def _restore_mode():
    yield
    import dose_prediction_amd
    dose_prediction_amd.config.set_x3_wgrad_terms(1)
    dose_prediction_amd.set_compute_dtype(torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("opt,mode", [("torch_adam", "fp32"), ("fused_adam", "fp32"), ("fused_adam", "fp32x3"), ("fused_adam", "fp32x3w3")])
def test_g6_hip_network_through_the_trainer_sequence(tmp_path, opt, mode, _restore_mode):
    """The HIP-backed c3d.Model in the fp32 parity mode -- and in the fast fp32x3 mode, with its default single-product weight gradients and
    with three-product ones (w3) -- driven exactly like NetworkTrainer drives the reference module: .to(device), train(), Adam(amsgrad) steps (each
    one must reach the packed weights), eval() forward, state_dict() -> torch.save -> load_state_dict into a fresh module -> identical
    eval forward.  Every mode has to land inside the SAME bands around the reference trainer's own numbers (2.5 x the distance between
    the reference's fp32 and float64 runs)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dose_prediction_amd
    from dose_prediction_amd import losses
    from dose_prediction_amd.models.c3d import Model
    from dose_prediction_amd.optim import FusedAdam
    dose_prediction_amd.config.set_x3_wgrad_terms(3 if mode == "fp32x3w3" else 1)
    dose_prediction_amd.set_compute_dtype("fp32x3" if mode.startswith("fp32x3") else torch.float32)
    dev = torch.device("cuda")
    g = load_golden("g6_trainer")
    sd0 = sub(g, "sd0")
    chans = [-1, 4, 4, 8, 8, 8]
    cls = torch.optim.Adam if opt == "torch_adam" else FusedAdam

    def mk(d, **kw):
        net = Model(3, 1, chans, chans)
        assert list(net.state_dict().keys()) == list(sd0.keys())
        net.load_state_dict(sd0)
        return TrainerReplay(net, dev, lambda o, t: losses.l1_loss(o, t), _val_fn(lambda t: t.to(dev)), _batches(), d,
                             optimizer_cls=cls, **kw)
    os.makedirs(tmp_path / "a")
    os.makedirs(tmp_path / "b")
    tr = mk(str(tmp_path / "a"), max_iter=1)
    tr.run()
    # Adam moments: the fp32x3 mode's backward pass is one-product (bf16-rounded operands, its stated gradient tolerance is 1e-2 -- section 3
    # of DESIGN.md), so (1 - beta2) g^2 may be 2 x that off; since round 5 the levels with rows shorter than 16 voxels take that path too
    # (they ran the exact-fp32 kernels before): this tiny network's first-layer second moment measures 2.2e-2
    _check_one_step(tr, g, str(tmp_path / "a"), 1e-5, moment_tol=3e-2 if mode.startswith("fp32x3") else 2e-2)
    tr = mk(str(tmp_path / "b"))
    tr.run()
    ck = _check_against_g6(tr, g, str(tmp_path / "b"), 1e-5)
    net = tr.network
    # checkpoint round trip (network_trainer.py:340-363): fresh module + load_state_dict == the trained module
    net2 = Model(3, 1, chans, chans)
    net2.load_state_dict(ck["network_state_dict"])
    net2.to(dev).eval()
    net.eval()
    xv = g6_data()[1]["Input"].to(dev)
    with torch.no_grad():
        a, b = net(xv), net2(xv)
    assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])
    # optimizer state resumes in either optimizer class
    for c2 in (torch.optim.Adam, FusedAdam):
        o2 = c2(net2.parameters(), lr=1e-3, weight_decay=3e-5, betas=(0.9, 0.999), eps=1e-08, amsgrad=True)
        o2.load_state_dict(ck["optimizer_state_dict"])
        assert len(o2.state_dict()["state"]) == len(ck["optimizer_state_dict"]["state"])
