"""Host-side logic that needs no GPU: the sliding-window layout of the cascade glue (MONAI sliding_window_inference as used at
train_light_linked_model.py:152-153) and its oracle restatement."""
import torch

import oracle


def test_window_layout_matches_monai_rules():
    from dose_prediction_amd import cascade
    # roi 96, overlap 0.25 -> scan interval 72; the last window is pulled back inside the volume
    assert cascade.window_starts((128, 128, 128), (96, 96, 96)) == [[0, 32]] * 3
    assert cascade.window_starts((192, 192, 128), (96, 96, 96)) == [[0, 72, 96], [0, 72, 96], [0, 32]]
    assert cascade.window_starts((96, 96, 96), (96, 96, 96)) == [[0]] * 3                 # one window: interval = roi
    assert cascade.window_starts((100, 96, 97), (96, 96, 96), overlap=0.5) == [[0, 4], [0], [0, 1]]
    try:
        cascade.window_starts((64, 96, 96), (96, 96, 96))
    except ValueError:
        pass
    else:
        raise AssertionError("window_starts takes padded volumes only (sliding_window_logits pads)")


def test_oracle_sliding_window_is_a_partition_of_unity():
    """With an identity predictor the stitched output equals the input whatever the overlap pattern (every voxel is divided by its
    own visit count), and the window batching (sw_batch_size) does not change the result."""
    x = torch.randn(2, 3, 20, 16, 24, generator=torch.Generator().manual_seed(3))
    for sw in (1, 3, 8):
        y = oracle.sliding_window_inference(x, (16, 16, 16), sw, lambda w: w, overlap=0.25)
        assert torch.allclose(y, x, atol=1e-6)
    calls = []
    oracle.sliding_window_inference(x, (16, 16, 16), 4, lambda w: (calls.append(w.shape[0]), w)[1], overlap=0.25)
    assert calls == [4, 4]                                                                # 2 images x (2 x 1 x 2) windows, 4 at a time


def test_oracle_sliding_window_pads_small_volumes():
    """An axis shorter than the roi is zero-padded diff // 2 in front (MONAI padding_mode='constant') and cropped back: an identity
    predictor returns the input, and a predictor that reports the window it was given sees the padded geometry."""
    x = torch.randn(1, 2, 10, 16, 13, generator=torch.Generator().manual_seed(4))
    seen = []
    y = oracle.sliding_window_inference(x, (16, 16, 16), 2, lambda w: (seen.append(tuple(w.shape[2:])), w)[1], overlap=0.25)
    assert torch.allclose(y, x, atol=1e-6) and seen == [(16, 16, 16)]
    # the zero padding is visible to the predictor: window = [3 zeros | 10 voxels | 3 zeros] along the first axis, 1 | 13 | 2 along the last
    y = oracle.sliding_window_inference(x, (16, 16, 16), 2, lambda w: w.abs().sum(dim=(2, 3, 4), keepdim=True).expand_as(w) * 0 +
                                        (w[:, :, :3].abs().sum() + w[:, :, 13:].abs().sum() + w[..., :1].abs().sum() + w[..., 14:].abs().sum()),
                                        overlap=0.25)
    assert float(y.abs().max()) == 0.0


def test_tools_and_package_sources_compile_and_losses_reject_unsupported_arguments():
    """The measurement tools (tools/*.py) are not run by any test (they need a GPU and minutes): at least every one of them, bench.py and
    the package must byte-compile; and the host-side argument checks of the round-5 additions work without a GPU."""
    import glob
    import os
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py"))) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")] + \
        sorted(glob.glob(os.path.join(root, "dose_prediction_amd", "**", "*.py"), recursive=True))
    assert len(files) > 40
    for f in files:
        compile(open(f).read(), f, "exec")          # (SyntaxError names the file; nothing is written)
    from dose_prediction_amd.losses import DiceCELoss, _weighted_sum
    DiceCELoss(to_onehot_y=True, softmax=True)
    for kw in (dict(), dict(to_onehot_y=True, softmax=True, squared_pred=True), dict(to_onehot_y=True, softmax=True, reduction="sum")):
        with pytest.raises(ValueError):
            DiceCELoss(**kw)
    ts = [torch.tensor(float(i + 1), requires_grad=True) for i in range(4)]
    out = _weighted_sum(ts, [1 / 3, 1 / 3, 1 / 3, 10.0])
    out.backward()
    assert abs(float(out.detach()) - 42.0) < 1e-5 and abs(float(ts[3].grad) - 10.0) < 1e-6 and abs(float(ts[0].grad) - 1 / 3) < 1e-6
    # the deterministic split-K helper declines shapes whose K shares would not start on 16-byte boundaries
    from dose_prediction_amd import ops
    assert ops._gemm_nt_splitk_det(torch.zeros(2, 30), torch.zeros(4, 30), None, 2, 4, 30, 30, 30, 3) is None


def test_folded_finalize_falls_back_to_the_two_launch_form_when_declined(monkeypatch):
    """ops._norm_forward / _norm_backward ask for the one-launch form first (dp_*_partial_finalize); on return value 3 (switched off, or an
    affine instance normalisation over several samples) they must make the two calls of the old form, in order, with the same scratch;
    on 0 nothing else may be launched.  Checked on the call sequence with the library mocked (no GPU)."""
    from dose_prediction_amd import _lib, ops

    class FakeLib:
        def dp_stats_nblk(self, V):
            return 3

    for decline in (True, False):
        calls = []

        def fake_call(name, *a, _d=decline):
            calls.append(name)
            return 3 if (_d and name.endswith("_partial_finalize")) else 0

        monkeypatch.setattr(_lib, "lib", lambda: FakeLib())
        monkeypatch.setattr(_lib, "call", fake_call)
        monkeypatch.setattr(ops, "_stream", lambda: 0)
        x = torch.zeros((2, 4, 4, 4, 16))
        y = torch.empty_like(x)
        mean, rstd, use_stats, ssn = ops._norm_forward(x, "instance", None, None, None, None, True, None, "relu", 1e-5, 0.1, y.data_ptr(), 16)
        assert tuple(mean.shape) == (2, 16) and use_stats and ssn == 16
        want = ["dp_stats_partial_finalize"] + (["dp_stats_partial", "dp_stats_finalize"] if decline else []) + ["dp_norm_act_fwd"]
        assert calls == want, calls
        # rows that came out of a convolution epilogue: only the finalize, never the folded form
        calls.clear()
        part = torch.zeros((2, 5, 2, 16))
        ops._norm_forward(x, "batch", None, None, None, None, True, None, "relu", 1e-5, 0.1, y.data_ptr(), 16, part=part)
        assert calls == ["dp_stats_finalize", "dp_norm_act_fwd"], calls
        calls.clear()
        gy = torch.zeros_like(x)
        ops._norm_backward(x, mean, rstd, None, None, None, "instance", "relu", True, ssn, gy.data_ptr(), 16, True, False, False)
        want = ["dp_norm_act_bwd_partial_finalize"] + (["dp_norm_act_bwd_partial", "dp_norm_bwd_finalize"] if decline else []) + ["dp_norm_act_bwd_apply"]
        assert calls == want, calls
