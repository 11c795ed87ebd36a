"""Functional (state-dict driven) restatement of the reference forward graphs.
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Every function takes ``sd`` (a ``state_dict``-like mapping name -> fp32 tensor with the
reference's key names) and a key ``prefix``; gradients flow through
``sd`` tensors that require grad, so the same code is the backward oracle via autograd.
BatchNorm running buffers are never modified in place: updated values are written into the
optional ``bn_out`` dict.
"""
import torch

from . import ref_ops as R

__all__ = [
    "single_conv", "base_unet", "c3d_model", "conv_block_3", "conv_block_7", "conv_3_1", "conv_3_1_old",
    "dilated_conv_block", "dual_dilated_block", "unet_res_block", "unet_basic_block", "unetr_pr_up_block",
    "modified_unetr_up_block", "unetr_up_block", "vit", "transformer_block", "vit_encoder", "main_subset_model", "dose_pyfer",
    "oar_transeg", "loss_l1_masked", "loss_l1_plain", "gen_loss", "gen_loss_val", "dose_postprocess", "dose_mae", "sliding_window_inference",
]


# ----------------------------------------------------------------------------- C3D (net_A)
def single_conv(sd, p, x, stride=1):
    """c3d.SingleConv, c3d.py:11-22: Conv3d(k3,p1,bias) -> InstanceNorm3d(affine) -> ReLU."""
    y = R.conv3d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride=stride, padding=1)
    y = R.instance_norm(y, sd[p + ".1.weight"], sd[p + ".1.bias"])
    return R.activation(y, "relu")


def base_unet(sd, p, x):
    """c3d.BaseUNet, c3d.py:41-149 (Encoder 65-72, UpConv 35-38, Decoder 99-115)."""
    enc = []
    h = x
    for lvl in range(1, 6):
        h = single_conv(sd, f"{p}encoder.encoder_{lvl}.0.single_conv", h, stride=1 if lvl == 1 else 2)
        h = single_conv(sd, f"{p}encoder.encoder_{lvl}.1.single_conv", h)
        enc.append(h)
    d = enc[4]
    for lvl in (4, 3, 2, 1):
        up = single_conv(sd, f"{p}decoder.upconv_{lvl}.conv", R.trilinear_up2(d))
        d = torch.cat((up, enc[lvl - 1]), dim=1)
        d = single_conv(sd, f"{p}decoder.decoder_conv_{lvl}.0.single_conv", d)
        if lvl > 1:
            d = single_conv(sd, f"{p}decoder.decoder_conv_{lvl}.1.single_conv", d)
    return d


def c3d_model(sd, x):
    """c3d.Model cascade of two BaseUNets, c3d.py:152-169."""
    x = R.store(x)
    a = base_unet(sd, "net_A.", x)
    b = base_unet(sd, "net_B.", torch.cat((a, x), dim=1))
    return [R.conv3d(a, sd["conv_out_A.weight"], sd["conv_out_A.bias"]),
            R.conv3d(b, sd["conv_out_B.weight"], sd["conv_out_B.bias"])]


# ----------------------------------------------------------------------------- multi-scale blocks
def conv_block_3(sd, p, x, act="relu", dilation=1):
    """blocks_MDUNet.conv_block_3, 64-78 (and dilated_conv_block_5/7, 160-191, with dilation 2/3):
    2 x [Conv3(bias) -> InstanceNorm(non-affine) -> act]."""
    for i in (0, 3):
        x = R.conv3d(x, sd[f"{p}.conv.{i}.weight"], sd[f"{p}.conv.{i}.bias"], padding=dilation, dilation=dilation)
        x = R.activation(R.instance_norm(x), act)
    return x


dilated_conv_block = conv_block_3


def conv_block_7(sd, p, x, training, bn_out=None):
    """blocks_MDUNet.conv_block_7, 98-112: 2 x [Conv7(bias) -> BatchNorm3d -> ReLU]."""
    for i in (0, 3):
        x = R.conv3d(x, sd[f"{p}.conv.{i}.weight"], sd[f"{p}.conv.{i}.bias"], padding=3)
        b = f"{p}.conv.{i + 1}"
        x, rm, rv = R.batch_norm(x, sd[b + ".weight"], sd[b + ".bias"], sd[b + ".running_mean"],
                                 sd[b + ".running_var"], training)
        if bn_out is not None and training:
            bn_out[b + ".running_mean"], bn_out[b + ".running_var"] = rm, rv
            bn_out[b + ".num_batches_tracked"] = sd[b + ".num_batches_tracked"] + 1
        x = R.activation(x, "relu")
    return x


def conv_3_1(sd, p, x, act, training, bn_out=None):
    """blocks_MDUNet.conv_3_1, 132-157.  NB line 137: conv_block_3 is built WITHOUT ``act`` so its
    inner activations are ReLU even for act='mish'; the trailing IN + act use ``act``."""
    x3 = R.activation(R.instance_norm(conv_block_3(sd, p + ".conv_3.0", x, "relu")), act)
    x7 = R.activation(R.instance_norm(conv_block_7(sd, p + ".conv_7.0", x, training, bn_out)), act)
    y = R.conv3d(torch.cat((x3, x7), dim=1), sd[p + ".conv.0.weight"], sd[p + ".conv.0.bias"])
    return R.activation(R.instance_norm(y), act)


def conv_3_1_old(sd, p, x, training, bn_out=None):
    """OldModels/Nets/blocks_MDUNet.conv_3_1, 132-148 (+ its conv_block_3, 64-78): BatchNorm+ReLU in
    BOTH branches, no trailing norm/act, bare 1x1x1 conv (key ``conv.weight``)."""
    def bn_branch(q, k):
        h = x
        for i in (0, 3):
            h = R.conv3d(h, sd[f"{q}.conv.{i}.weight"], sd[f"{q}.conv.{i}.bias"], padding=k // 2)
            b = f"{q}.conv.{i + 1}"
            h, rm, rv = R.batch_norm(h, sd[b + ".weight"], sd[b + ".bias"], sd[b + ".running_mean"],
                                     sd[b + ".running_var"], training)
            if bn_out is not None and training:
                bn_out[b + ".running_mean"], bn_out[b + ".running_var"] = rm, rv
            h = R.activation(h, "relu")
        return h
    x3, x7 = bn_branch(p + ".conv_3", 3), bn_branch(p + ".conv_7", 7)
    return R.conv3d(torch.cat((x3, x7), dim=1), sd[p + ".conv.weight"], sd[p + ".conv.bias"])


def dual_dilated_block(sd, p, x, act):
    """blocks_MDUNet.DualDilatedBlock, 194-215 (used iff multiS_conv=False)."""
    x3 = conv_block_3(sd, p + ".conv_3", x, act, 1)
    x5 = conv_block_3(sd, p + ".conv_5", x, act, 2)
    x7 = conv_block_3(sd, p + ".conv_7", x, act, 3)
    y = R.conv3d(torch.cat((x3, x5, x7), dim=1), sd[p + ".conv.0.weight"], sd[p + ".conv.0.bias"])
    return R.activation(R.instance_norm(y), act)


# ----------------------------------------------------------------------------- MONAI leaves [unpinned]
def unet_res_block(sd, p, x):
    """MONAI 0.7.0 dynunet_block.UnetResBlock (k3, stride 1, norm 'instance' = non-affine):
    conv1 -> IN -> LeakyReLU(.01) -> conv2 -> IN; residual through conv3(1x1x1) -> IN when
    Cin != Cout; add; LeakyReLU.  Convs have no bias.  Call sites dose_pyfer.py:69-77."""
    w1 = sd[p + ".conv1.conv.weight"]
    out = R.activation(R.instance_norm(R.conv3d(x, w1, padding=1)), "lrelu")
    out = R.instance_norm(R.conv3d(out, sd[p + ".conv2.conv.weight"], padding=1))
    res = x
    if w1.shape[0] != w1.shape[1]:
        res = R.store(R.instance_norm(R.conv3d(x, sd[p + ".conv3.conv.weight"])))     # (a stored tensor in the HIP path)
    return R.activation(out + res, "lrelu")


def unet_basic_block(sd, p, x):
    """MONAI UnetBasicBlock: conv1 -> IN -> LeakyReLU -> conv2 -> IN -> LeakyReLU."""
    out = R.activation(R.instance_norm(R.conv3d(x, sd[p + ".conv1.conv.weight"], padding=1)), "lrelu")
    return R.activation(R.instance_norm(R.conv3d(out, sd[p + ".conv2.conv.weight"], padding=1)), "lrelu")


def unetr_pr_up_block(sd, p, x, num_layer):
    """MONAI UnetrPrUpBlock (conv_block & res_block True): transp_conv_init then num_layer x
    [ConvTranspose k2s2 -> UnetResBlock(C->C)].  Call sites dose_pyfer.py:78-113."""
    x = R.conv_transpose3d_k2s2(x, sd[p + ".transp_conv_init.conv.weight"])
    for j in range(num_layer):
        x = R.conv_transpose3d_k2s2(x, sd[f"{p}.blocks.{j}.0.conv.weight"])
        x = unet_res_block(sd, f"{p}.blocks.{j}.1", x)
    return x


def transformer_block(sd, b, t, num_heads):
    """MONAI 0.7.0 TransformerBlock (pre-norm): t + SABlock(LayerNorm(t)); t + MLPBlock(LayerNorm(t)), MLP = Linear -> GELU(erf) ->
    Linear, dropout 0.  Cross-checked against nn.TransformerEncoderLayer(norm_first=True) in tests/test_oracle_leaves_cpu.py."""
    t = R.store(t + R.attention(R.layer_norm(t, sd[b + "norm1.weight"], sd[b + "norm1.bias"]),
                                sd[b + "attn.qkv.weight"], sd[b + "attn.out_proj.weight"],
                                sd[b + "attn.out_proj.bias"], num_heads))
    h = R.layer_norm(t, sd[b + "norm2.weight"], sd[b + "norm2.bias"])
    h = R.gelu(R.linear(h, sd[b + "mlp.linear1.weight"], sd[b + "mlp.linear1.bias"]))
    return R.store(t + R.linear(h, sd[b + "mlp.linear2.weight"], sd[b + "mlp.linear2.bias"]))


def vit(sd, p, x, num_layers, num_heads):
    """MONAI 0.7.0 ViT(pos_embed='perceptron', classification=False): patchify -> Linear ->
    + position_embeddings -> num_layers pre-norm TransformerBlocks -> LayerNorm.
    Returns (normed last state, [output of every block])."""
    if p + "patch_embedding.patch_embeddings.weight" in sd:
        # pos_embed='conv' (the constructor default of oar_transeg.Model, oar_transeg.py:28): Conv3d(kernel = stride = patch), then
        # flatten(2).transpose(-1, -2)  [MONAI 0.7.0 PatchEmbeddingBlock; parity unpinned like every MONAI leaf]
        w = sd[p + "patch_embedding.patch_embeddings.weight"]
        t = R.conv3d(x, w, sd[p + "patch_embedding.patch_embeddings.bias"], w.shape[2], 0, 1)
        t = t.flatten(2).transpose(-1, -2)
    else:
        t = R.linear(R.patchify(x), sd[p + "patch_embedding.patch_embeddings.1.weight"],
                     sd[p + "patch_embedding.patch_embeddings.1.bias"])
    t = R.store(t + R.store_weight(sd[p + "patch_embedding.position_embeddings"]))
    hidden = []
    for i in range(num_layers):
        t = transformer_block(sd, f"{p}blocks.{i}.", t, num_heads)
        hidden.append(t)
    return R.layer_norm(t, sd[p + "norm.weight"], sd[p + "norm.bias"]), hidden


def _proj_feat(t, feat):
    """dose_pyfer.py:118-122 / oar_transeg.py:165-169: [B,N,H] -> [B,H,f0,f1,f2]."""
    return t.view(t.shape[0], *feat, t.shape[-1]).permute(0, 4, 1, 2, 3).contiguous()


# ----------------------------------------------------------------------------- decoder
def modified_unetr_up_block(sd, p, x, skip, act, training, multiS_conv=True, bn_out=None, old=False):
    """base_blocks.ModifiedUnetrUpBlock.forward, 136-141: tconv k2s2 -> cat(out, skip) -> conv_3_1
    (MultiUnetBasicBlock 12-28) or DualDilatedBlock."""
    out = R.conv_transpose3d_k2s2(x, sd[p + ".transp_conv.conv.weight"])
    out = torch.cat((out, skip), dim=1)
    q = p + ".conv_block.cov_"
    if old:
        return conv_3_1_old(sd, q, out, training, bn_out)
    if multiS_conv:
        return conv_3_1(sd, q, out, act, training, bn_out)
    return dual_dilated_block(sd, q, out, act)


def unetr_up_block(sd, p, x, skip):
    """MONAI UnetrUpBlock(res_block=False) used when mode_multi_dec=False (dose_pyfer.py:163-170)."""
    out = R.conv_transpose3d_k2s2(x, sd[p + ".transp_conv.conv.weight"])
    return unet_basic_block(sd, p + ".conv_block", torch.cat((out, skip), dim=1))


# ----------------------------------------------------------------------------- DOSE-PYFER
def vit_encoder(sd, p, x, num_layers, num_heads):
    """dose_pyfer.ViTEncoder.forward, 124-144."""
    feat = tuple(s // 16 for s in x.shape[2:])
    i = num_layers // 4
    z12, hidden = vit(sd, p + "vit.", x, num_layers, num_heads)
    e1 = unet_res_block(sd, p + "skip1.layer", x)
    e2 = unetr_pr_up_block(sd, p + "skip2", _proj_feat(hidden[i], feat), 2)
    e3 = unetr_pr_up_block(sd, p + "skip3", _proj_feat(hidden[2 * i], feat), 1)
    e4 = unetr_pr_up_block(sd, p + "skip4", _proj_feat(hidden[3 * i], feat), 0)
    return [e1, e2, e3, e4, _proj_feat(z12, feat)]


def main_subset_model(sd, p, x, num_layers, num_heads, act, training, mode_multi_dec=True,
                      multiS_conv=True, bn_out=None):
    """dose_pyfer.MainSubsetModel.forward, 311-319 (+ PyMSCDecoder.forward 232-239)."""
    e1, e2, e3, e4, e5 = vit_encoder(sd, p + "encoder.", x, num_layers, num_heads)
    skips = {4: e4, 3: e3, 2: e2, 1: e1}
    d = e5
    decs = {}
    for lvl in (4, 3, 2, 1):
        q = f"{p}decoder.decoder{lvl}"
        if mode_multi_dec:
            d = modified_unetr_up_block(sd, q, d, skips[lvl], act, training, multiS_conv, bn_out)
        else:
            d = unetr_up_block(sd, q, d, skips[lvl])
        decs[lvl] = d
    return [R.conv3d(decs[l], sd[f"{p}dose_convertors.{l - 1}.0.weight"], sd[f"{p}dose_convertors.{l - 1}.0.bias"])
            for l in (1, 2, 3, 4)]


def dose_pyfer(sd, x, num_layers=8, num_heads=6, act="mish", training=True, mode_multi_dec=True,
               multiS_conv=True, bn_out=None):
    """dose_pyfer.Model.forward, 355-360: net_A -> cat(out_A, x) -> net_B; conv_out_A."""
    x = R.store(x)          # (entry conversion NCDHW fp32 -> NDHWC storage type)
    a = base_unet(sd, "net_A.", x)
    outs = main_subset_model(sd, "net_B.", torch.cat((a, x), dim=1), num_layers, num_heads, act, training,
                             mode_multi_dec, multiS_conv, bn_out)
    return [R.conv3d(a, sd["conv_out_A.weight"], sd["conv_out_A.bias"]), outs]


# ----------------------------------------------------------------------------- OAR-TRANSEG
def oar_transeg(sd, x, num_heads=12, training=True, bn_out=None, old=False):
    """oar_transeg.Model.forward, 171-185 (num_layers hard-coded 12, line 73; decoder act defaults to
    'relu', base_blocks.py:103).  ``old=True`` selects the OldModels TRANSEG decoder variant."""
    x = R.store(x)
    feat = tuple(s // 16 for s in x.shape[2:])
    z, hidden = vit(sd, "vit.", x, 12, num_heads)
    e1 = unet_res_block(sd, "encoder1.layer", x)
    e2 = unetr_pr_up_block(sd, "encoder2", _proj_feat(hidden[3], feat), 2)
    e3 = unetr_pr_up_block(sd, "encoder3", _proj_feat(hidden[6], feat), 1)
    e4 = unetr_pr_up_block(sd, "encoder4", _proj_feat(hidden[9], feat), 0)
    d = _proj_feat(z, feat)
    for name, skip in (("decoder5", e4), ("decoder4", e3), ("decoder3", e2), ("decoder2", e1)):
        d = modified_unetr_up_block(sd, name, d, skip, "relu", training, True, bn_out, old=old)
    return R.conv3d(d, sd["out.conv.conv.weight"], sd["out.conv.conv.bias"])


# ----------------------------------------------------------------------------- loss / metric (callers)
def loss_l1_masked(pred, gt, freez=True):
    """Train/loss.py Loss.forward (cascade), 13-28."""
    dose, mask = gt[:, 0:1], gt[:, 1:2] > 0
    lb = (pred[1][mask] - dose[mask]).abs().mean()
    return lb if freez else 0.5 * (pred[0][mask] - dose[mask]).abs().mean() + lb


def _huber(p, g, delta=0.5):
    """nn.HuberLoss(reduction='mean', delta=0.5) (loss.py:53)."""
    d = (p - g).abs()
    return torch.where(d < delta, 0.5 * d * d, delta * (d - 0.5 * delta)).mean()


def loss_l1_plain(pred, gt):
    """Train/loss.py Loss.forward with casecade=False, 29-39: one prediction, masked L1."""
    dose, mask = gt[:, 0:1], gt[:, 1:2] > 0
    return (pred[mask] - dose[mask]).abs().mean()


def gen_loss_val(prediction, gt, huber=False):
    """Train/loss.py GenLoss.forward mode != 'train', 109-117: masked L1 (+ Huber when huber=True) of ONE prediction."""
    dose, mask = gt[:, 0:1], gt[:, 1:2] > 0
    l1 = (prediction[mask] - dose[mask]).abs().mean()
    return _huber(prediction[mask], dose[mask]) + l1 if huber else l1


def gen_loss(predictions, gt, delta1=10, delta2=1, casecade=True, freez=True, huber=False):
    """Train/loss.py GenLoss.forward mode='train', 69-107: delta1 * masked L1 (Huber delta 0.5 when huber=True) at full
    resolution + delta2 * mean of masked L1 at 1/2,1/4,1/8 resolution against trilinear
    (align_corners) down-sampled dose and nearest-exact down-sampled mask."""
    dose, mask = gt[:, 0:1], gt[:, 1:2]
    pred_a = None
    if casecade:
        pred_a, predictions = predictions[0], predictions[1]
    size = dose.shape[-1]
    l_ds = 0
    for i, pr in enumerate(predictions[1:], start=1):
        dim = size // (2 ** i)
        g = torch.nn.functional.interpolate(dose, size=(dim,) * 3, mode="trilinear", align_corners=True)
        m = torch.nn.functional.interpolate(mask, size=(dim,) * 3, mode="nearest-exact") > 0
        l_ds = l_ds + (pr[m] - g[m]).abs().mean()
    l_ds = l_ds / len(predictions[1:])
    m0 = mask > 0
    full = _huber(predictions[0][m0], dose[m0]) if huber else (predictions[0][m0] - dose[m0]).abs().mean()
    loss = delta1 * full + delta2 * l_ds
    if casecade and not freez:
        loss = loss + 0.5 * (pred_a[m0] - dose[m0]).abs().mean()
    return loss


def dose_postprocess(pred, mask):
    """train_light_pyfer.py:166-172: zero where mask<1 or pred<0, then x70 Gy."""
    pred = pred.clone()
    pred[(mask < 1) | (pred < 0)] = 0
    return 70.0 * pred


def dose_mae(pred, gt, mask):
    """evaluate_openKBP.get_3D_Dose_dif, 42-48: mean |pred-gt| over possible_dose_mask>0."""
    m = mask > 0
    return (pred[m] - gt[m]).abs().mean()


def sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap=0.25):
    """MONAI 0.7 monai.inferers.sliding_window_inference with mode="constant" (call site
    train_light_linked_model.py:152-153), restated from the published algorithm: scan interval int(roi*(1-overlap)) (the roi
    itself on an axis that fits one window), dense_patch_slices origins with the last window pulled back inside the volume,
    predictions summed with unit importance and divided by the visit count.  An axis shorter than the roi is zero-padded
    (padding_mode="constant", cval 0) by diff // 2 in front and diff - diff // 2 behind, and the result is cropped back.
    PARITY UNPINNED: MONAI is absent from the authoring container."""
    import math
    orig = tuple(inputs.shape[2:])
    pads = [(max(r - s_, 0) // 2, max(r - s_, 0) - max(r - s_, 0) // 2) for s_, r in zip(orig, roi_size)]
    if any(a or b for a, b in pads):
        inputs = torch.nn.functional.pad(inputs, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
        full = sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap)
        return full[:, :, pads[0][0]:pads[0][0] + orig[0], pads[1][0]:pads[1][0] + orig[1], pads[2][0]:pads[2][0] + orig[2]]
    image_size = tuple(inputs.shape[2:])
    starts = []
    for size, roi in zip(image_size, roi_size):
        assert roi <= size
        interval = roi if roi == size else max(1, int(roi * (1 - overlap)))
        num = int(math.ceil(float(size) / interval))
        first = None
        for d in range(num):
            if d * interval + roi >= size:
                first = d
                break
        count = first + 1 if first is not None else 1
        axis = []
        for i in range(count):
            st = i * interval
            st -= max(st + roi - size, 0)
            axis.append(st)
        starts.append(axis)
    slices = [(z, y, x) for z in starts[0] for y in starts[1] for x in starts[2]]
    num_win, total = len(slices), len(slices) * inputs.shape[0]
    out = cnt = None
    for g0 in range(0, total, sw_batch_size):
        idxs = list(range(g0, min(g0 + sw_batch_size, total)))
        data = torch.cat([inputs[i // num_win:i // num_win + 1, :, slices[i % num_win][0]:slices[i % num_win][0] + roi_size[0],
                                 slices[i % num_win][1]:slices[i % num_win][1] + roi_size[1],
                                 slices[i % num_win][2]:slices[i % num_win][2] + roi_size[2]] for i in idxs])
        prob = predictor(data)
        if out is None:
            out = torch.zeros((inputs.shape[0], prob.shape[1]) + image_size, dtype=prob.dtype)
            cnt = torch.zeros_like(out)
        for k, i in enumerate(idxs):
            z, y, x = slices[i % num_win]
            out[i // num_win, :, z:z + roi_size[0], y:y + roi_size[1], x:x + roi_size[2]] += prob[k]
            cnt[i // num_win, :, z:z + roi_size[0], y:y + roi_size[1], x:x + roi_size[2]] += 1
    return out / cnt
