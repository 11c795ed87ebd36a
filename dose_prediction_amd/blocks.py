"""Leaf nn.Modules of the HIP path.

Every class keeps the SAME parameter / buffer names and shapes as the reference module it replaces (the
``state_dict`` is the drop-in contract): standard torch modules are used as parameter
holders (so initialisation and key names are torch's / the reference's own), while ``forward`` runs the
hand-written HIP kernels through ``dose_prediction_amd.ops`` on NDHWC tensors.
"""
import torch
import torch.nn as nn

from . import config, ops




def _branch_side_stream(device, main):
    """The stream of the 3x3x3 branches / small up-sampling blocks for the caller's stream `main` (streams.side_stream: chosen so that it
    does not share a hardware queue with the caller's stream)."""
    from . import streams
    return streams.side_stream(device, main, streams.ROLE_BRANCH)


def _act_name(act):
    return "relu" if act == "relu" else "mish"


# ------------------------------------------------------------------------------------------------ c3d.py leaves
class SingleConv(nn.Module):
    """c3d.SingleConv (c3d.py:11-22): Conv3d(bias) -> InstanceNorm3d(affine) -> ReLU.
    Keys: single_conv.0.{weight,bias}, single_conv.1.{weight,bias}."""

    def __init__(self, in_ch, out_ch, kernel_size, stride, padding):
        super().__init__()
        self.single_conv = nn.Sequential(
            nn.Conv3d(in_ch, out_ch, kernel_size=kernel_size, padding=padding, stride=stride, bias=True),
            nn.InstanceNorm3d(out_ch, affine=True),
            nn.ReLU(inplace=True))

    def forward(self, x, next_conv=None):
        """x: an NDHWC tensor, or a pair (a, b) standing for torch.cat((a, b), channels) (virtual concat).
        next_conv: the nn.Conv3d that is the ONLY consumer of the result (fp32x3: the normalisation writes its split operand)."""
        conv, norm = self.single_conv[0], self.single_conv[1]
        y, st = ops.conv3d(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], conv.dilation[0], stats=True, bias_grad_zero=True)
        return ops.norm_act(y, "instance", norm.weight, norm.bias, act="relu", eps=norm.eps, stats=st, x3_split_for=next_conv)


class UpConv(nn.Module):
    """c3d.UpConv (c3d.py:25-38): trilinear x2 (align_corners) -> Conv3d -> InstanceNorm3d(affine) -> ReLU."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv3d(in_ch, out_ch, kernel_size=3, padding=1, stride=1, bias=True),
            nn.InstanceNorm3d(out_ch, affine=True),
            nn.ReLU(inplace=True))

    def forward(self, x):
        conv, norm = self.conv[0], self.conv[1]
        y, st = ops.conv3d(ops.trilinear_up2(x, x3_split_for=conv), conv.weight, conv.bias, 1, 1, 1, stats=True, bias_grad_zero=True)
        return ops.norm_act(y, "instance", norm.weight, norm.bias, act="relu", eps=norm.eps, stats=st)


# ------------------------------------------------------------------------------------------------ blocks_MDUNet.py leaves
class _ConvNormActPair(nn.Module):
    """Two x [Conv3d(bias) -> norm -> act] with the reference's nn.Sequential key layout ``conv.{0,1,3,4}``."""

    def __init__(self, ch_in, ch_out, k, norm, act, dilation=1):
        super().__init__()
        pad = dilation * (k // 2)
        mk_norm = (lambda: nn.BatchNorm3d(ch_out)) if norm == "batch" else (lambda: nn.InstanceNorm3d(ch_out))
        mk_act = (lambda: nn.ReLU(inplace=True)) if act == "relu" else (lambda: nn.Mish(inplace=True))
        self.conv = nn.Sequential(
            nn.Conv3d(ch_in, ch_out, kernel_size=k, stride=1, padding=pad, dilation=dilation, bias=True), mk_norm(), mk_act(),
            nn.Conv3d(ch_out, ch_out, kernel_size=k, stride=1, padding=pad, dilation=dilation, bias=True), mk_norm(), mk_act())
        self._norm, self._act = norm, act

    def forward(self, x, final_norm_act=None):
        """final_norm_act: optional (act) of a trailing non-affine InstanceNorm + activation that the caller
        (conv_3_1) applies right after this block; kept separate (it is a second normalisation)."""
        for i in (0, 3):
            conv, norm = self.conv[i], self.conv[i + 1]
            if self._norm == "batch" and not self.training:
                x, st = ops.conv3d(x, conv.weight, conv.bias, 1, conv.padding[0], conv.dilation[0]), None
            else:
                x, st = ops.conv3d(x, conv.weight, conv.bias, 1, conv.padding[0], conv.dilation[0], stats=True, bias_grad_zero=True)
            nxt = self.conv[3] if i == 0 else None      # (fp32x3: the first normalisation writes the second convolution's split operand)
            if self._norm == "batch":
                upd = self.training and config.bn_updates_enabled()
                x = ops.norm_act(x, "batch", norm.weight, norm.bias, norm.running_mean if (upd or not self.training) else None,
                                 norm.running_var if (upd or not self.training) else None,
                                 training=self.training, act=self._act, eps=norm.eps, momentum=norm.momentum, stats=st, x3_split_for=nxt)
                if upd:
                    norm.num_batches_tracked += 1
            else:
                x = ops.norm_act(x, "instance", act=self._act, eps=norm.eps, stats=st, x3_split_for=nxt)
        return x


class conv_block_3(_ConvNormActPair):
    """blocks_MDUNet.conv_block_3 (64-78): 2 x [Conv3 -> InstanceNorm3d -> act]."""

    def __init__(self, ch_in, ch_out, act="relu"):
        super().__init__(ch_in, ch_out, 3, "instance", act)


class conv_block_7(_ConvNormActPair):
    """blocks_MDUNet.conv_block_7 (98-112): 2 x [Conv7 -> BatchNorm3d -> ReLU]."""

    def __init__(self, ch_in, ch_out):
        super().__init__(ch_in, ch_out, 7, "batch", "relu")


class dilated_conv_block_5(_ConvNormActPair):
    """blocks_MDUNet.dilated_conv_block_5 (160-174): 3^3 convs, dilation 2."""

    def __init__(self, ch_in, ch_out, act="relu"):
        super().__init__(ch_in, ch_out, 3, "instance", act, dilation=2)


class dilated_conv_block_7(_ConvNormActPair):
    """blocks_MDUNet.dilated_conv_block_7 (177-191): 3^3 convs, dilation 3."""

    def __init__(self, ch_in, ch_out, act="relu"):
        super().__init__(ch_in, ch_out, 3, "instance", act, dilation=3)


class conv_3_1(nn.Module):
    """blocks_MDUNet.conv_3_1 (132-157): 3^3 branch || 7^3 branch -> cat -> 1^3 conv -> IN -> act.
    NB (line 137) the inner conv_block_3 is built without ``act`` => inner activations are ReLU."""

    def __init__(self, ch_in, ch_out, act):
        super().__init__()
        mk_act = (lambda: nn.ReLU(inplace=True)) if act == "relu" else (lambda: nn.Mish(inplace=True))
        self.conv_3 = nn.Sequential(conv_block_3(ch_in, ch_out), nn.InstanceNorm3d(ch_out), mk_act())
        self.conv_7 = nn.Sequential(conv_block_7(ch_in, ch_out), nn.InstanceNorm3d(ch_out), mk_act())
        self.conv = nn.Sequential(nn.Conv3d(ch_out * 2, ch_out, kernel_size=1, stride=1, padding=0, bias=True),
                                  nn.InstanceNorm3d(ch_out), mk_act())
        self._act = _act_name(act)

    def forward(self, x):
        """x: tensor or (a, b) pair = virtual torch.cat (the two first convolutions then read the operands directly)."""
        first = x[0] if isinstance(x, (tuple, list)) else x
        if config.branch_stream() and first.is_cuda and config.branch_stream_allowed():
            # the latency- / fabric-bound 3x3x3 branch on a second stream beside the MFMA-bound 7x7x7 branch
            main = ops._current_stream_object(first.device)
            side = _branch_side_stream(first.device, main)
        else:
            main = side = None
        if side is not None and side != main:
            side.wait_stream(main)
            for t in (x if isinstance(x, (tuple, list)) else (x,)):
                t.record_stream(side)
            with torch.cuda.stream(side):
                r3 = self.conv_3[0](x)
            r7 = self.conv_7[0](x)
            main.wait_stream(side)
            r3.record_stream(main)
        else:
            r3, r7 = self.conv_3[0](x), self.conv_7[0](x)
        if r3.shape[-1] % 8 == 0:       # both branches normalised straight into the halves of the mixer's input (no cat copy)
            x37 = ops.norm_act_cat(r3, r7, act=self._act)
        else:
            x37 = ops.cat((ops.norm_act(r3, "instance", act=self._act), ops.norm_act(r7, "instance", act=self._act)))
        y = ops.conv3d(x37, self.conv[0].weight, self.conv[0].bias, bias_grad_zero=True)
        return ops.norm_act(y, "instance", act=self._act)


class _BnPair(_ConvNormActPair):
    def __init__(self, ch_in, ch_out, k):
        super().__init__(ch_in, ch_out, k, "batch", "relu")


class conv_3_1_old(nn.Module):
    """OldModels/Nets/blocks_MDUNet.conv_3_1 (132-148): BatchNorm+ReLU in both branches, bare 1^3 conv."""

    def __init__(self, ch_in, ch_out):
        super().__init__()
        self.conv_3 = _BnPair(ch_in, ch_out, 3)
        self.conv_7 = _BnPair(ch_in, ch_out, 7)
        self.conv = nn.Conv3d(ch_out * 2, ch_out, kernel_size=1, stride=1, padding=0, bias=True)

    def forward(self, x):
        return ops.conv3d(ops.cat((self.conv_3(x), self.conv_7(x))), self.conv.weight, self.conv.bias)


class DualDilatedBlock(nn.Module):
    """blocks_MDUNet.DualDilatedBlock (194-215)."""

    def __init__(self, ch_in, ch_out, act="relu"):
        super().__init__()
        mk_act = (lambda: nn.ReLU(inplace=True)) if act == "relu" else (lambda: nn.Mish(inplace=True))
        self.conv_3 = conv_block_3(ch_in, ch_out, act)
        self.conv_5 = dilated_conv_block_5(ch_in, ch_out, act)
        self.conv_7 = dilated_conv_block_7(ch_in, ch_out, act)
        self.conv = nn.Sequential(nn.Conv3d(ch_out * 3, ch_out, kernel_size=1, stride=1, padding=0, bias=True),
                                  nn.InstanceNorm3d(ch_out), mk_act())
        self._act = _act_name(act)

    def forward(self, x):
        if isinstance(x, (tuple, list)):          # dilated convolutions take the generic kernel: materialise the concat once
            x = ops.cat(x)
        y = ops.cat((self.conv_3(x), self.conv_5(x), self.conv_7(x)))
        y = ops.conv3d(y, self.conv[0].weight, self.conv[0].bias, bias_grad_zero=True)
        return ops.norm_act(y, "instance", act=self._act)


# ------------------------------------------------------------------------------------------------ MONAI-compatible leaves
class _Conv(nn.Sequential):
    """MONAI ``Convolution(conv_only=True)``: an nn.Sequential with a single child named ``conv``."""

    def __init__(self, conv):
        super().__init__()
        self.add_module("conv", conv)


def get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=3, stride=1, act=None, norm=None, dropout=None,
                   bias=False, conv_only=True, is_transposed=False):
    """MONAI dynunet_block.get_conv_layer for the cases the reference uses (3-D; padding (k-s+1)//2)."""
    if spatial_dims != 3:
        raise ValueError("only spatial_dims=3 is supported")
    pad = (kernel_size - stride + 1) // 2
    if is_transposed:
        if kernel_size != 2 or stride != 2:
            raise ValueError("HIP path implements ConvTranspose3d for kernel 2 / stride 2 only")
        return _Conv(nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride, pad, 0, bias=bias))
    return _Conv(nn.Conv3d(in_channels, out_channels, kernel_size, stride, pad, bias=bias))


def _run_conv(layer, x, stats=False):
    c = layer.conv
    if isinstance(c, nn.ConvTranspose3d):
        return ops.conv_transpose2x(x, c.weight)
    return ops.conv3d(x, c.weight, c.bias, c.stride[0], c.padding[0], c.dilation[0], stats=stats)


class UnetResBlock(nn.Module):
    """MONAI 0.7.0 UnetResBlock, norm 'instance' (non-affine): conv1-IN-LeakyReLU-conv2-IN (+ conv3(1x1)-IN on the
    residual when in != out or stride != 1) - add - LeakyReLU.  conv3/norm3 always exist as in 0.7.0."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, act_name=None, dropout=None):
        super().__init__()
        self.conv1 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size, stride)
        self.conv2 = get_conv_layer(spatial_dims, out_channels, out_channels, kernel_size, 1)
        self.conv3 = get_conv_layer(spatial_dims, in_channels, out_channels, 1, stride)
        self.lrelu = nn.LeakyReLU(0.01, inplace=True)
        self.norm1, self.norm2, self.norm3 = (nn.InstanceNorm3d(out_channels) for _ in range(3))
        self.downsample = in_channels != out_channels or stride != 1

    def forward(self, inp, inp_cat=None):
        """inp_cat: optional (a, b) pair with cat((a, b)) == inp, used by the 3x3x3 convolution (virtual concat)."""
        out, st = _run_conv(self.conv1, inp_cat if inp_cat is not None else inp, stats=True)
        out = ops.norm_act(out, "instance", act="lrelu", stats=st, x3_split_for=self.conv2.conv)
        out, st = _run_conv(self.conv2, out, stats=True)
        res = inp
        if self.downsample:
            res = ops.norm_act(_run_conv(self.conv3, inp), "instance")
        return ops.norm_act(out, "instance", res=res, act="lrelu", stats=st)        # IN(out) + res -> LeakyReLU, fused


class UnetBasicBlock(nn.Module):
    """MONAI UnetBasicBlock: conv1-IN-LeakyReLU-conv2-IN-LeakyReLU."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, act_name=None, dropout=None):
        super().__init__()
        self.conv1 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size, stride)
        self.conv2 = get_conv_layer(spatial_dims, out_channels, out_channels, kernel_size, 1)
        self.lrelu = nn.LeakyReLU(0.01, inplace=True)
        self.norm1, self.norm2 = nn.InstanceNorm3d(out_channels), nn.InstanceNorm3d(out_channels)

    def forward(self, inp):
        """inp: tensor or (a, b) pair = virtual torch.cat."""
        out, st = _run_conv(self.conv1, inp, stats=True)
        out = ops.norm_act(out, "instance", act="lrelu", stats=st, x3_split_for=self.conv2.conv)
        out, st = _run_conv(self.conv2, out, stats=True)
        return ops.norm_act(out, "instance", act="lrelu", stats=st)


class UnetrBasicBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, res_block=False):
        super().__init__()
        cls = UnetResBlock if res_block else UnetBasicBlock
        self.layer = cls(spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name)

    def forward(self, inp, inp_cat=None):
        if inp_cat is not None and isinstance(self.layer, UnetResBlock):
            return self.layer(inp, inp_cat)
        return self.layer(inp)


class UnetrPrUpBlock(nn.Module):
    """MONAI UnetrPrUpBlock (conv_block=True, res_block=True): tconv then num_layer x [tconv -> UnetResBlock]."""

    def __init__(self, spatial_dims, in_channels, out_channels, num_layer, kernel_size, stride, upsample_kernel_size,
                 norm_name, conv_block=False, res_block=False):
        super().__init__()
        if not (conv_block and res_block):
            raise ValueError("HIP path implements UnetrPrUpBlock for conv_block=True, res_block=True (the reference's use)")
        up = upsample_kernel_size
        self.transp_conv_init = get_conv_layer(spatial_dims, in_channels, out_channels, up, up, is_transposed=True)
        self.blocks = nn.ModuleList([
            nn.Sequential(get_conv_layer(spatial_dims, out_channels, out_channels, up, up, is_transposed=True),
                          UnetResBlock(spatial_dims, out_channels, out_channels, kernel_size, stride, norm_name))
            for _ in range(num_layer)])

    def forward(self, x):
        x = _run_conv(self.transp_conv_init, x)
        for blk in self.blocks:
            x = blk[1](_run_conv(blk[0], x))
        return x


class UnetrUpBlock(nn.Module):
    """MONAI UnetrUpBlock (res_block=False default): tconv -> cat(out, skip) -> UnetBasicBlock."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, upsample_kernel_size, norm_name, res_block=False):
        super().__init__()
        up = upsample_kernel_size
        self.transp_conv = get_conv_layer(spatial_dims, in_channels, out_channels, up, up, is_transposed=True)
        cls = UnetResBlock if res_block else UnetBasicBlock
        self.conv_block = cls(spatial_dims, out_channels + out_channels, out_channels, kernel_size, 1, norm_name)

    def forward(self, inp, skip):
        return self.conv_block((_run_conv(self.transp_conv, inp), skip))


# ------------------------------------------------------------------------------------------------ ViT (MONAI 0.7.0 semantics)
class PatchEmbeddingBlock(nn.Module):
    """MONAI PatchEmbeddingBlock: pos_embed='perceptron' (keys patch_embeddings.1.{weight,bias}; every reference call site) or
    'conv' (keys patch_embeddings.{weight,bias}; the constructor default of oar_transeg.Model); position_embeddings, cls_token."""

    def __init__(self, in_channels, img_size, patch_size, hidden_size, num_heads, pos_embed, dropout_rate=0.0, spatial_dims=3):
        super().__init__()
        if pos_embed not in ("perceptron", "conv"):
            raise ValueError(f"pos_embed should be 'conv' or 'perceptron', got {pos_embed!r}")
        n_patches = 1
        for i, p in zip(img_size, patch_size):
            if i % p:
                raise ValueError("img_size must be divisible by patch_size")
            n_patches *= i // p
        self.in_channels, self.patch, self.pos_embed = in_channels, patch_size[0], pos_embed
        patch_dim = in_channels * patch_size[0] * patch_size[1] * patch_size[2]
        if pos_embed == "conv":
            # MONAI: Conv3d(kernel = stride = patch) then flatten(2).transpose(-1, -2); keys patch_embeddings.{weight,bias}; default
            # Conv3d initialisation (MONAI's _init_weights only touches Linear / LayerNorm).  It is the SAME contraction as the
            # perceptron variant with the weight's input axes in (c, p1, p2, p3) instead of (p1, p2, p3, c) order.
            self.patch_embeddings = nn.Conv3d(in_channels, hidden_size, kernel_size=patch_size[0], stride=patch_size[0])
            lin = self.patch_embeddings
        else:
            self.patch_embeddings = nn.Sequential(nn.Identity(), nn.Linear(patch_dim, hidden_size))
            lin = self.patch_embeddings[1]
        self.position_embeddings = nn.Parameter(torch.zeros(1, n_patches, hidden_size))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))
        nn.init.trunc_normal_(self.position_embeddings, mean=0.0, std=0.02, a=-2.0, b=2.0)
        if pos_embed == "perceptron":
            nn.init.trunc_normal_(lin.weight, mean=0.0, std=0.02, a=-2.0, b=2.0)
            nn.init.zeros_(lin.bias)
        # the patch embedding is the transformer's first layer = its LAST backward node: once its gradient is in place, the grouped
        # launch fills every transformer weight gradient recorded during the backward pass (ops.flush_deferred), still on the ViT stream
        lin.bias.register_post_accumulate_grad_hook(ops.flush_deferred)
        lin.weight.register_post_accumulate_grad_hook(ops.flush_deferred)

    def _forward_conv(self, x):
        """pos_embed='conv': the Linear of the perceptron variant with the convolution weight viewed in the patchify order
        (p1, p2, p3, c); the permuted copy is a torch op (autograd returns the gradient in the parameter's own layout).  The
        transformer of OAR-TRANSEG has one input channel (3.1 M weight elements), so the copy is negligible there."""
        conv = self.patch_embeddings
        w2 = conv.weight.permute(0, 2, 3, 4, 1).reshape(conv.out_channels, -1)
        tok = ops.patchify(x, self.in_channels, self.patch)
        rows = tok.shape[0] * tok.shape[1]
        splitk = max(1, min(32, 480 // max(1, -(-rows // 128) * -(-conv.out_channels // 128))))
        t = ops.linear(tok, w2, conv.bias, splitk=splitk if tok.shape[-1] >= 4096 else 1)
        return ops.add_broadcast(t, self.position_embeddings)

    def forward(self, x):
        if self.pos_embed == "conv":
            return self._forward_conv(x)
        lin = self.patch_embeddings[1]
        if config.x3() and x.dtype == torch.float32 and not x.requires_grad:
            # fp32x3: split the voxels once, patchify the bf16 halves straight into the GEMM's operand blocks (no fp32 token matrix)
            n_tok_rows = x.shape[0] * (x.shape[1] // self.patch) * (x.shape[2] // self.patch) * (x.shape[3] // self.patch)
            sk = max(1, min(32, 480 // max(1, -(-n_tok_rows // 128) * -(-lin.out_features // 128))))
            t = ops.patch_embed_x3(x, self.in_channels, self.patch, lin.weight, lin.bias, sk if lin.in_features >= 4096 else 1)
            if t is not None:
                return ops.add_broadcast(t, self.position_embeddings)
        tok = ops.patchify(x, self.in_channels, self.patch)
        rows = tok.shape[0] * tok.shape[1]
        # K = p^3*C is huge while M x N is small: split K so the GEMM fills the 256 CUs
        # ... with 128x128 tiles in two full rounds (<= 512 blocks): 425 -> 283 us at 1024 x 768 x 102400
        splitk = max(1, min(32, 480 // max(1, -(-rows // 128) * -(-lin.out_features // 128))))
        t = ops.linear(tok, lin.weight, lin.bias, splitk=splitk if tok.shape[-1] >= 4096 else 1, defer_wgrad=True)
        return ops.add_broadcast(t, self.position_embeddings)


class SABlock(nn.Module):
    def __init__(self, hidden_size, num_heads, dropout_rate=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=False)

    def forward(self, x):
        o = ops.attention(ops.linear(x, self.qkv.weight, defer_wgrad=True), self.num_heads)
        return ops.linear(o, self.out_proj.weight, self.out_proj.bias, defer_wgrad=True)


class MLPBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0):
        super().__init__()
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)
        self.fn = nn.GELU()

    def forward(self, x):
        return ops.mlp(x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias)


class TransformerBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, num_heads, dropout_rate=0.0):
        super().__init__()
        self.mlp = MLPBlock(hidden_size, mlp_dim, dropout_rate)
        self.norm1 = nn.LayerNorm(hidden_size)
        self.attn = SABlock(hidden_size, num_heads, dropout_rate)
        self.norm2 = nn.LayerNorm(hidden_size)

    def forward(self, x):
        x = ops.add(x, self.attn(ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)))
        return ops.add(x, self.mlp(ops.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)))


class ViT(nn.Module):
    """MONAI 0.7.0 ViT(classification=False): returns (LayerNorm(last), [every block's output])."""

    def __init__(self, in_channels, img_size, patch_size, hidden_size=768, mlp_dim=3072, num_layers=12, num_heads=12,
                 pos_embed="conv", classification=False, num_classes=2, dropout_rate=0.0, spatial_dims=3):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        if dropout_rate != 0.0:
            raise ValueError("HIP path implements dropout_rate=0 (the reference's setting)")
        self.patch_embedding = PatchEmbeddingBlock(in_channels, img_size, patch_size, hidden_size, num_heads, pos_embed,
                                                   dropout_rate, spatial_dims)
        self.blocks = nn.ModuleList([TransformerBlock(hidden_size, mlp_dim, num_heads, dropout_rate) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(hidden_size)
        for m in self.blocks.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, mean=0.0, std=0.02, a=-2.0, b=2.0)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x, events=None):
        """events: optional list that receives one recorded HIP event per block output (the caller runs this method on a side
        stream and lets consumers of hidden[i] wait for events[i] only: models.dose_pyfer.run_vit_beside)."""
        x = self.patch_embedding(x)
        hidden = []
        if not self.blocks:
            return ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps), hidden
        # pre-norm blocks x = x + attn(norm1(x)); x = x + mlp(norm2(x)), with every residual add fused into the LayerNorm that reads
        # its result (ops.add_layer_norm): norm2 of the same block, then norm1 of the next block (the final ViT.norm after the last)
        first = self.blocks[0]
        n = ops.layer_norm(x, first.norm1.weight, first.norm1.bias, first.norm1.eps)
        for i, blk in enumerate(self.blocks):
            x, n = ops.add_layer_norm(x, blk.attn(n), blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            nxt = self.blocks[i + 1].norm1 if i + 1 < len(self.blocks) else self.norm
            x, n = ops.add_layer_norm(x, blk.mlp(n), nxt.weight, nxt.bias, nxt.eps)
            hidden.append(x)
            if events is not None:
                ev = torch.cuda.Event()
                ev.record()
                events.append(ev)
        return n, hidden
