"""The reference's training losses (DosePrediction/Train/loss.py) on device for the benchmark / trainer harness
(SURVEY.md section 8f row 1).  The masked means sum(|p-g|*m)/sum(m) are one fused HIP reduction each (ops.masked_l1:
dp_masked_l1_fwd / _bwd) instead of boolean-mask indexing: same value, no dynamic shapes, no host synchronisation.  Only the
ground-truth down-sampling (no gradient, a few MB) stays in torch."""
import torch
import torch.nn.functional as F


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
        g = F.interpolate(dose, size=dims, mode="trilinear", align_corners=True)
        m = F.interpolate(mask, size=dims, mode="nearest-exact")
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
