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
        raise AssertionError("a volume smaller than the crop must be rejected (padding is not implemented)")


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
