"""Cascade glue OAR-TRANSEG -> DOSE-PYFER (inference), the reference's LinkedNet.test_step
(DosePrediction/Train/train_light_linked_model.py:143-173):

    ct = input[:, :1]; ptv = input[:, 1:]                                  (143-144)
    oars = post_pred(seg(ct))            arg-max -> one-hot(8)             (152-157; one window when the volume == roi)
    oars = permute(oars, (0, 3, 2, 1)); ct = permute(ct, (0, 1, 4, 3, 2))  (158, 163: seg and dose loaders use opposite axis orders)
    structures = cat(ptv, oars[1:], ct)  -> 9 channels                     (165-167)
    dose = PYFER(structures)[1][0]; dose[mask < 1 or dose < 0] = 0; dose *= 70   (168-173)

The arg-max / one-hot / channel packing is one HIP kernel (dp_argmax_onehot) writing straight into the NDHWC 9-channel
PYFER input; the axis reversal is a strided copy.  Sliding-window stitching for volumes larger than the segmentation crop
is not implemented yet: the segmentation model must be built for the full volume size."""
import torch

from . import config, ops
from .models.c3d import to_ndhwc, from_ndhwc


@torch.no_grad()
def oar_masks(seg_model, ct):
    """ct [B,1,D,H,W] fp32 -> (labels int32 [B,D,H,W], logits NDHWC) through the HIP segmentation network."""
    logits = seg_model.forward_ndhwc(to_ndhwc(ct))
    labels = ops.argmax_onehot(logits, None, 0, labels=True)
    return labels, logits


@torch.no_grad()
def cascade_forward(seg_model, dose_model, ct, ptv, possible_dose_mask=None, reverse_axes=True):
    """Returns (dose_gy [B,1,...] fp32 in Gy, labels).  ct, ptv: [B,1,D,H,W] fp32 on the GPU.  With reverse_axes the OAR masks
    and the CT are flipped to the dose loader's axis order (W,H,D) exactly as lines 158/163 do; ptv is taken as given."""
    B = ct.shape[0]
    logits = seg_model.forward_ndhwc(to_ndhwc(ct))                       # [B,D,H,W,8]
    dt = config.compute_dtype()
    D, H, W = logits.shape[1:4]
    staged = torch.zeros((B, D, H, W, 16), dtype=dt, device=ct.device)   # channels: 0 PTV | 1..7 OARs | 8 CT | pad
    labels = ops.argmax_onehot(logits, staged, choff=1, labels=True)
    staged[..., 8:9] = to_ndhwc(ct)[..., :1]
    if reverse_axes:
        staged = staged.permute(0, 3, 2, 1, 4).contiguous()              # (D,H,W) -> (W,H,D)
    staged[..., 0:1] = to_ndhwc(ptv)[..., :1]
    out_a = dose_model.net_A.forward_ndhwc(staged)
    outs = dose_model.net_B.forward_ndhwc(ops.cat((out_a, staged)), (out_a, staged))
    dose = from_ndhwc(outs[0])
    if possible_dose_mask is not None:
        dose = torch.where(possible_dose_mask < 1, torch.zeros_like(dose), dose)
    dose = torch.where(dose < 0, torch.zeros_like(dose), dose)
    return 70.0 * dose, labels
