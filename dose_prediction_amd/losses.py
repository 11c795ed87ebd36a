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


def _masked_huber(pred, gt, mask, delta=0.5):
    from . import ops
    return ops.masked_l1(pred, gt, mask, huber_delta=delta)


def gen_loss_val(prediction, gt, huber=False):
    """GenLoss.forward(mode != 'train'), loss.py:109-117: masked L1 (+ Huber(delta 0.5) when huber) of one prediction."""
    dose, mask = gt[:, 0:1], gt[:, 1:2]
    l1 = _masked_l1(prediction, dose, mask)
    return _masked_huber(prediction, dose, mask) + l1 if huber else l1


def gen_loss(predictions, gt, delta1=10.0, delta2=1.0, casecade=True, freez=True, huber=False, mode="train"):
    """GenLoss.forward, loss.py:69-117 (mode='train': 69-107; any other mode: the validation branch on a single prediction)."""
    if mode != "train":
        return gen_loss_val(predictions, gt, huber)
    # (the two channel slices of gt as contiguous fp32 volumes ONCE: every masked mean and the resampling below read them -- round 4 let
    # each of those eight uses make its own copy)
    dose, mask = gt[:, 0:1].contiguous().float(), gt[:, 1:2].contiguous().float()
    pred_a = None
    if casecade:
        pred_a, predictions = predictions[0], predictions[1]
    size = dose.shape[2:]
    inter = predictions[1:]
    terms, weights = [], []
    for i, pr in enumerate(inter, start=1):
        dims = tuple(s // (2 ** i) for s in size)
        g, m = _resample_gt(dose, mask, dims)
        terms.append(_masked_l1(pr, g, m))
        weights.append(float(delta2) / max(1, len(inter)))
    terms.append(_masked_huber(predictions[0], dose, mask) if huber else _masked_l1(predictions[0], dose, mask))
    weights.append(float(delta1))
    if casecade and not freez:
        terms.append(_masked_l1(pred_a, dose, mask))
        weights.append(0.5)
    # loss = delta1 * full + delta2 * mean(deep-supervision terms) (+ 0.5 * L1(pred_A)): one weighted sum instead of a chain of 0-dim
    # adds / multiplies / divides (19 ATen launches per step, forward + backward)
    return _weighted_sum(terms, weights)


class _WeightedSum(torch.autograd.Function):
    """sum_i w_i t_i of 0-dim fp32 tensors: forward stack + dot (3 launches), backward one scaled copy of the weights."""

    @staticmethod
    def forward(ctx, w, *ts):
        ctx.save_for_backward(w)
        return (torch.stack(ts) * w).sum()

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        gw = g * w
        return (None,) + tuple(gw.unbind(0))


_WS_W = {}


def _weighted_sum(terms, weights):
    key = (terms[0].device, tuple(weights))
    w = _WS_W.get(key)
    if w is None:
        if len(_WS_W) > 16:
            _WS_W.clear()
        w = _WS_W[key] = torch.tensor(weights, dtype=torch.float32, device=terms[0].device)
    return _WeightedSum.apply(w, *terms)


def l1_loss(pred, gt, freez=True, casecade=True):
    """Loss.forward, loss.py:13-39 (casecade=True: [pred_A, pred_B]; casecade=False: one prediction, 29-39)."""
    dose, mask = gt[:, 0:1], gt[:, 1:2]
    if not casecade:
        return _masked_l1(pred, dose, mask)
    lb = _masked_l1(pred[1] if not isinstance(pred[1], (list, tuple)) else pred[1][0], dose, mask)
    return lb if freez else 0.5 * _masked_l1(pred[0], dose, mask) + lb


class DiceCELoss(torch.nn.Module):
    """Drop-in for ``monai.losses.DiceCELoss`` as OAR-TRANSEG uses it (OARSegmentation/train_light_transeg.py:148:
    ``DiceCELoss(to_onehot_y=True, softmax=True)``; called on (logits [B, 8, D, H, W], labels [B, 1, D, H, W]) at :196 / :212), on the
    fused HIP kernels (ops.dice_ce).  Constructor arguments outside that use are rejected rather than silently ignored."""

    def __init__(self, include_background=True, to_onehot_y=False, sigmoid=False, softmax=False, other_act=None, squared_pred=False,
                 jaccard=False, reduction="mean", smooth_nr=1e-5, smooth_dr=1e-5, batch=False, ce_weight=None, lambda_dice=1.0,
                 lambda_ce=1.0):
        super().__init__()
        if not (to_onehot_y and softmax) or sigmoid or other_act is not None or squared_pred or jaccard or batch or ce_weight is not None \
                or not include_background or reduction != "mean":
            raise ValueError("HIP path implements DiceCELoss(to_onehot_y=True, softmax=True) with MONAI's other defaults (the reference's use)")
        self.smooth_nr, self.smooth_dr, self.lambda_dice, self.lambda_ce = float(smooth_nr), float(smooth_dr), float(lambda_dice), float(lambda_ce)

    def forward(self, input, target):
        from . import ops
        return ops.dice_ce(input, target, self.smooth_nr, self.smooth_dr, self.lambda_dice, self.lambda_ce)
