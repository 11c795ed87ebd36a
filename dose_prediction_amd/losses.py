"""The reference's training losses (DosePrediction/Train/loss.py) on device for the benchmark / trainer harness
(SURVEY.md section 8f row 1).  The masked means sum(|p-g|*m)/sum(m) are one fused HIP reduction each (ops.masked_l1:
dp_masked_l1_fwd / _bwd) instead of boolean-mask indexing: same value, no dynamic shapes, no host synchronisation; the
ground-truth pyramid (trilinear align_corners dose, nearest-exact mask) is dp_resample_gt."""
import torch

from . import _lib


def _resample_gt(dose, mask, dims):
    """(dose, mask) [B,1,D,H,W] fp32 -> the same at spatial size dims (loss.py:56-66)."""
    B, _, D, H, W = dose.shape
    dose, mask = dose.contiguous().float(), mask.contiguous().float()
    od = torch.empty((B, 1) + tuple(dims), dtype=torch.float32, device=dose.device)
    om = torch.empty_like(od)
    _lib.call("dp_resample_gt", dose.data_ptr(), mask.data_ptr(), od.data_ptr(), om.data_ptr(), B, D, H, W, dims[0], dims[1], dims[2],
              torch.cuda.current_stream().cuda_stream)
    return od, om


def _masked_l1(pred, gt, mask):
    from . import ops
    return ops.masked_l1(pred, gt, mask)


def gen_loss(predictions, gt, delta1=10.0, delta2=1.0, casecade=True, freez=True):
    """GenLoss.forward(mode='train', huber=False), loss.py:69-107."""
    dose, mask = gt[:, 0:1], gt[:, 1:2]
    pred_a = None
    if casecade:
        pred_a, predictions = predictions[0], predictions[1]
    size = dose.shape[2:]
    l_ds = 0
    inter = predictions[1:]
    for i, pr in enumerate(inter, start=1):
        dims = tuple(s // (2 ** i) for s in size)
        g, m = _resample_gt(dose, mask, dims)
        l_ds = l_ds + _masked_l1(pr, g, m)
    l_ds = l_ds / max(1, len(inter))
    loss = delta1 * _masked_l1(predictions[0], dose, mask) + delta2 * l_ds
    if casecade and not freez:
        loss = loss + 0.5 * _masked_l1(pred_a, dose, mask)
    return loss


def l1_loss(pred, gt, freez=True):
    """Loss.forward (cascade), loss.py:13-28."""
    dose, mask = gt[:, 0:1], gt[:, 1:2]
    lb = _masked_l1(pred[1] if not isinstance(pred[1], (list, tuple)) else pred[1][0], dose, mask)
    return lb if freez else 0.5 * _masked_l1(pred[0], dose, mask) + lb
