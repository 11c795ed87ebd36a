"""Cascade glue OAR-TRANSEG -> DOSE-PYFER (inference), the reference's LinkedNet.test_step
(DosePrediction/Train/train_light_linked_model.py:143-173):

    ct = input[:, :1]; ptv = input[:, 1:]                                  (143-144)
    oars = post_pred(seg(ct))            arg-max -> one-hot(8)             (152-157; one window when the volume == roi)
    oars = permute(oars, (0, 3, 2, 1)); ct = permute(ct, (0, 1, 4, 3, 2))  (158, 163: seg and dose loaders use opposite axis orders)
    structures = cat(ptv, oars[1:], ct)  -> 9 channels                     (165-167)
    dose = PYFER(structures)[1][0]; dose[mask < 1 or dose < 0] = 0; dose *= 70   (168-173)

The arg-max / one-hot / channel packing is one HIP kernel (dp_argmax_onehot) writing straight into the NDHWC 9-channel
PYFER input; the axis reversal is a strided copy.  Volumes larger than the segmentation crop go through
sliding_window_logits() (MONAI sliding_window_inference with constant blending, the call at 152-153: roi = IMAGE_SIZE^3,
overlap 0.25, sw_batch_size 4), stitched on device by dp_window_accumulate / dp_window_normalize."""
import contextlib
import math

import torch

from . import _lib, config, ops
from .models.c3d import to_ndhwc, from_ndhwc


def _seg_mode():
    """The segmentation network of the cascade is inference-only (train_light_linked_model.py:152-160 runs it under no_grad) and its
    arg-max masks are INPUTS of the dose network: it runs in config.cascade_seg_mode() (default 'fp32x3') so that the masks are the
    reference's, independently of the storage type the dose network trains in."""
    m = config.cascade_seg_mode()
    return config.compute_mode_as(m) if m is not None and m != config.compute_mode() else contextlib.nullcontext()


def _seg_logits(seg_model, ct, roi_size, sw_batch_size, overlap):
    with _seg_mode():
        if roi_size is not None and tuple(roi_size) != tuple(ct.shape[2:]):
            return sliding_window_logits(seg_model, ct, tuple(roi_size), sw_batch_size, overlap)
        return seg_model.forward_ndhwc(to_ndhwc(ct))


def window_starts(image_size, roi_size, overlap=0.25):
    """Window origins per axis exactly as MONAI 0.7 lays them out (_get_scan_interval + dense_patch_slices): interval =
    int(roi * (1 - overlap)) (the whole roi when the axis fits in one window), windows i * interval until one reaches the
    end of the axis, the last one pulled back inside the volume."""
    starts = []
    for size, roi in zip(image_size, roi_size):
        if roi > size:
            raise ValueError(f"volume axis {size} is smaller than the segmentation crop {roi}: pad it first (sliding_window_logits does)")
        interval = roi if roi == size else max(1, int(roi * (1 - overlap)))
        num = int(math.ceil(float(size) / interval))
        first = next((d for d in range(num) if d * interval + roi >= size), None)
        count = first + 1 if first is not None else 1
        starts.append([i * interval - max(i * interval + roi - size, 0) for i in range(count)])
    return starts


@torch.no_grad()
def sliding_window_logits(seg_model, ct, roi_size, sw_batch_size=4, overlap=0.25):
    """ct [B,1,D,H,W] fp32 -> NDHWC logits over the whole volume, averaged over overlapping roi windows (constant blend).
    seg_model is built for img_size == roi_size; windows run sw_batch_size at a time in image-major order like MONAI."""
    B, _, D, H, W = ct.shape
    rz, ry, rx = roi_size
    pads = [(max(r - s, 0) // 2, max(r - s, 0) - max(r - s, 0) // 2) for s, r in zip((D, H, W), roi_size)]
    if any(a or b for a, b in pads):
        # an axis shorter than the crop: MONAI zero-pads diff // 2 in front, the rest behind (padding_mode="constant"), runs the
        # windows on the padded volume and crops the stitched result back
        (z0, z1), (y0, y1), (x0, x1) = pads
        padded = torch.zeros((B, 1, D + z0 + z1, H + y0 + y1, W + x0 + x1), dtype=ct.dtype, device=ct.device)
        padded[:, :, z0:z0 + D, y0:y0 + H, x0:x0 + W] = ct
        full = sliding_window_logits(seg_model, padded, roi_size, sw_batch_size, overlap)
        return full[:, z0:z0 + D, y0:y0 + H, x0:x0 + W].contiguous()
    sz, sy, sx = window_starts((D, H, W), roi_size, overlap)
    wins = [(n, z, y, x) for n in range(B) for z in sz for y in sy for x in sx]
    x_all = to_ndhwc(ct)
    dt, acc, cnt, C = x_all.dtype, None, None, None
    for g0 in range(0, len(wins), sw_batch_size):
        grp = wins[g0:g0 + sw_batch_size]
        batch = torch.stack([x_all[n, z:z + rz, y:y + ry, x:x + rx] for n, z, y, x in grp]).contiguous()
        logits = ops.as_rows(seg_model.forward_ndhwc(batch))
        if acc is None:
            C = logits.shape[-1]
            acc = torch.zeros((B, D, H, W, C), dtype=torch.float32, device=ct.device)
            cnt = torch.zeros((B, D, H, W), dtype=torch.float32, device=ct.device)
        for i, (n, z, y, x) in enumerate(grp):
            _lib.call("dp_window_accumulate", logits[i].data_ptr(), logits.stride(-2), acc.data_ptr(), cnt.data_ptr(), n, D, H, W,
                      rz, ry, rx, z, y, x, C, ops._DT[dt], torch.cuda.current_stream().cuda_stream)
    out = torch.empty((B, D, H, W, C), dtype=dt, device=ct.device)
    _lib.call("dp_window_normalize", acc.data_ptr(), cnt.data_ptr(), out.data_ptr(), C, B * D * H * W, C,
              ops._DT[dt], torch.cuda.current_stream().cuda_stream)
    return out


@torch.no_grad()
def oar_masks(seg_model, ct):
    """ct [B,1,D,H,W] fp32 -> (labels int32 [B,D,H,W], logits NDHWC) through the HIP segmentation network."""
    logits = _seg_logits(seg_model, ct, None, 4, 0.25)
    labels = ops.argmax_onehot(logits, None, 0, labels=True)
    return labels, logits


def _put_channel(staged, ch, vol):
    """Write a [B,1,D,H,W] fp32 NCDHW volume into channel `ch` of an NDHWC staging buffer (one conversion pass; replaces a padded
    to_ndhwc() of the volume followed by a strided single-channel copy)."""
    B, _, D, H, W = vol.shape
    if tuple(staged.shape[:4]) != (B, D, H, W):
        raise ValueError(f"volume {tuple(vol.shape)} does not match the staging grid {tuple(staged.shape)}")
    vol = vol.contiguous().float()
    _lib.call("dp_ncdhw_to_ndhwc", vol.data_ptr(), staged.data_ptr() + ch * staged.element_size(), B, 1, D * H * W, staged.shape[-1], 1,
              ops._dt(staged), ops._stream())


@torch.no_grad()
def cascade_structures(seg_model, ct, ptv, reverse_axes=True, roi_size=None, sw_batch_size=4, overlap=0.25, staged=False):
    """The 9-channel PYFER input [B,9,D,H,W] fp32 of the linked model (train_light_linked_model.py:143-167) from a CT and a PTV
    channel: OAR-TRANSEG (no grad) -> arg-max -> one-hot classes 1..7 -> axis reversal -> cat(ptv, oars, ct).  Used when the dose
    network TRAINS on the segmentation network's masks (BASELINE.json configs[3], [4]); returns (structures, labels).
    staged=True returns the NDHWC staging buffer itself (compute dtype, 16 channels) for dose_model.forward_staged()."""
    B = ct.shape[0]
    logits = _seg_logits(seg_model, ct, roi_size, sw_batch_size, overlap)
    D, H, W = logits.shape[1:4]
    staged_out = staged
    staged = torch.zeros((B, D, H, W, 16), dtype=config.compute_dtype(), device=ct.device)
    labels = ops.argmax_onehot(logits, staged, choff=1, labels=True)
    _put_channel(staged, 8, ct)
    if reverse_axes:
        staged = staged.permute(0, 3, 2, 1, 4).contiguous()
    _put_channel(staged, 0, ptv)
    if staged_out:
        return staged, labels
    return from_ndhwc(staged)[:, :9].float().contiguous(), labels


@torch.no_grad()
def cascade_forward(seg_model, dose_model, ct, ptv, possible_dose_mask=None, reverse_axes=True, roi_size=None,
                    sw_batch_size=4, overlap=0.25):
    """Returns (dose_gy [B,1,...] fp32 in Gy, labels).  ct, ptv: [B,1,D,H,W] fp32 on the GPU.  With reverse_axes the OAR masks
    and the CT are flipped to the dose loader's axis order (W,H,D) exactly as lines 158/163 do; ptv is taken as given.
    roi_size: crop the segmentation network was built for (sliding-window inference when it is smaller than the volume)."""
    B = ct.shape[0]
    logits = _seg_logits(seg_model, ct, roi_size, sw_batch_size, overlap)   # [B,D,H,W,8]
    dt = config.compute_dtype()
    D, H, W = logits.shape[1:4]
    staged = torch.zeros((B, D, H, W, 16), dtype=dt, device=ct.device)   # channels: 0 PTV | 1..7 OARs | 8 CT | pad
    labels = ops.argmax_onehot(logits, staged, choff=1, labels=True)
    _put_channel(staged, 8, ct)
    if reverse_axes:
        staged = staged.permute(0, 3, 2, 1, 4).contiguous()              # (D,H,W) -> (W,H,D)
    _put_channel(staged, 0, ptv)
    out_a = dose_model.net_A.forward_ndhwc(staged)
    outs = dose_model.net_B.forward_ndhwc(ops.cat((out_a, staged)), (out_a, staged))
    dose = from_ndhwc(outs[0])
    if possible_dose_mask is not None:
        dose = torch.where(possible_dose_mask < 1, torch.zeros_like(dose), dose)
    dose = torch.where(dose < 0, torch.zeros_like(dose), dose)
    return 70.0 * dose, labels
