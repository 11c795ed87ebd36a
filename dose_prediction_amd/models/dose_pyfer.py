"""DOSE-PYFER on the HIP path.  Mirrors DosePrediction/Models/Networks/dose_pyfer.py: class names, constructor
signatures, ValueErrors, state_dict keys and forward I/O ([output_A, [out128, out64, out32, out16]], NCDHW fp32)."""
from typing import Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..blocks import ViT, UnetrBasicBlock, UnetrPrUpBlock, UnetrUpBlock
from .base_blocks import ModifiedUnetrUpBlock
from .c3d import BaseUNet, to_ndhwc, from_ndhwc


def ensure_tuple_rep(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


_SIDE_STREAMS = {}


class _SideRun:
    """Handle of a ViT forward running on the side stream: hidden(i) / final() make the CURRENT stream wait for exactly the
    producer they need (a per-block event), and register the tensor with the allocator for that stream.  With a CU partition
    (`part` is not None) the consumers of the hidden states are meant to run inside `with run.beside():`, i.e. on the stream that
    owns the other compute units, until final(*outs) joins everything back into the caller's stream."""

    def __init__(self, z, hidden, events, side, main, part=None):
        self.z, self.hid, self.events, self.side, self.main, self.part = z, hidden, events, side, main, part

    def beside(self):
        import contextlib
        return torch.cuda.stream(self.part) if self.part is not None else contextlib.nullcontext()

    def hidden(self, i):
        t = self.hid[i]
        if self.side is not None:
            cur = ops._current_stream_object(t.device)
            cur.wait_event(self.events[i])
            t.record_stream(cur)
        return t

    def final(self, *outs):
        """The transformer's output on the caller's stream; `outs`: tensors produced inside beside() that the caller goes on to use."""
        if self.side is not None:
            self.main.wait_stream(self.side)
            self.z.record_stream(self.main)
            if self.part is not None:
                self.main.wait_stream(self.part)
                _record_all(outs, self.main)
        return self.z


def _record_all(out, stream):
    if torch.is_tensor(out):
        out.record_stream(stream)
    elif isinstance(out, (list, tuple)):
        for o in out:
            _record_all(o, stream)


def run_vit_beside(vit, x_in, first=None, first_inputs=()):
    """Start vit(x_in) on a second HIP stream (config.vit_side_stream) and return (_SideRun, first()).  `first` -- the 128^3 block
    on the same input -- is ENQUEUED BEFORE the transformer although both start from the same point of the caller's stream: its
    autograd nodes are then older than the transformer's, so the backward pass issues the whole transformer / patch-embedding
    backward (48 % of the gradient bytes of the data-parallel exchange) before the 128^3 block's.  The transformer is ~150 small-grid,
    latency-bound launches (1024 token rows), while the skip blocks that run meanwhile on the caller's stream are full-chip
    kernels on 128^3 .. 16^3 volumes; skip block k only waits for the transformer layer it reads.  Autograd replays each node on
    the stream of its forward pass, so the backward branches overlap the same way.

    (Round 3 also built the two branches on DISJOINT compute units -- hipExtStreamCreateWithCUMask -- and measured it slower for every
    split, 28-51 ms per step against 26.7: DESIGN section 5; removed in round 4.)  `first_inputs` is kept for source compatibility."""
    from .. import config
    if not (config.vit_side_stream() and x_in.is_cuda):
        out = first() if first is not None else None
        z, hidden = vit(x_in)
        return _SideRun(z, hidden, None, None, None), out
    main = ops._current_stream_object(x_in.device)
    key = (x_in.device.index, main.cuda_stream)
    side = _SIDE_STREAMS.get(key)
    if side is None:
        from .. import streams
        side = _SIDE_STREAMS[key] = streams.side_stream(x_in.device, main, streams.ROLE_VIT)
    part = None
    fork = torch.cuda.Event()
    fork.record(main)
    out = first() if first is not None else None
    side.wait_event(fork)
    x_in.record_stream(side)
    events = []
    with torch.cuda.stream(side):
        z, hidden = vit(x_in, events)
    return _SideRun(z, hidden, events, side, main, part if first is not None else None), out


class small_blocks_beside:
    """Context for the 64^3 .. 16^3 up-sampling blocks that hang off the transformer's hidden states (UnetrPrUpBlock skip2..4 /
    encoder2..4): with config.branch_stream() they run on a third HIP stream beside the 128^3 block of the main stream (their small
    grids leave most of the chip idle); `join(*outs)` inside the context hands their results back to the caller's stream.  Without the
    switch (or under a CU partition) it is run.beside()."""

    def __init__(self, x_in, run):
        from .. import config
        self.run = run
        self.on = bool(config.branch_stream() and x_in.is_cuda and run.side is not None and run.part is None
                       and config.branch_stream_allowed())
        self.dev = x_in.device

    def __enter__(self):
        if not self.on:
            self.ctx = self.run.beside()
            self.ctx.__enter__()
            return lambda *outs: None
        from ..blocks import _branch_side_stream
        self.main = ops._current_stream_object(self.dev)
        self.side = _branch_side_stream(self.dev, self.main)
        self.side.wait_stream(self.main)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()

        def join(*outs):
            self.outs = outs
        return join

    def __exit__(self, *a):
        r = self.ctx.__exit__(*a)
        if self.on and a[0] is None:
            self.main.wait_stream(self.side)
            for o in getattr(self, "outs", ()):
                o.record_stream(self.main)
        return r


class ViTEncoder(nn.Module):
    """dose_pyfer.ViTEncoder (22-144)."""

    def __init__(self, in_channels: int, img_size: Union[Sequence[int], int], feature_size: int = 16, hidden_size: int = 768,
                 mlp_dim: int = 3072, num_heads: int = 12, num_layers: int = 12, pos_embed: str = "conv",
                 norm_name: Union[Tuple, str] = "instance", conv_block: bool = True, res_block: bool = True,
                 dropout_rate: float = 0.0, spatial_dims: int = 3) -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        self.num_layers = num_layers
        img_size = ensure_tuple_rep(img_size, spatial_dims)
        self.patch_size = ensure_tuple_rep(16, spatial_dims)
        self.feat_size = tuple(img_d // p_d for img_d, p_d in zip(img_size, self.patch_size))
        self.hidden_size = hidden_size
        self.classification = False
        self.vit = ViT(in_channels=in_channels, img_size=img_size, patch_size=self.patch_size, hidden_size=hidden_size,
                       mlp_dim=mlp_dim, num_layers=self.num_layers, num_heads=num_heads, pos_embed=pos_embed,
                       classification=self.classification, dropout_rate=dropout_rate, spatial_dims=spatial_dims)
        self.skip1 = UnetrBasicBlock(spatial_dims, in_channels, feature_size, kernel_size=3, stride=1, norm_name=norm_name,
                                     res_block=res_block)
        self.skip2 = UnetrPrUpBlock(spatial_dims, hidden_size, feature_size * 2, num_layer=2, kernel_size=3, stride=1,
                                    upsample_kernel_size=2, norm_name=norm_name, conv_block=conv_block, res_block=res_block)
        self.skip3 = UnetrPrUpBlock(spatial_dims, hidden_size, feature_size * 4, num_layer=1, kernel_size=3, stride=1,
                                    upsample_kernel_size=2, norm_name=norm_name, conv_block=conv_block, res_block=res_block)
        self.skip4 = UnetrPrUpBlock(spatial_dims, hidden_size, feature_size * 8, num_layer=0, kernel_size=3, stride=1,
                                    upsample_kernel_size=2, norm_name=norm_name, conv_block=conv_block, res_block=res_block)
        self.proj_axes = (0, spatial_dims + 1) + tuple(d + 1 for d in range(spatial_dims))
        self.proj_view_shape = list(self.feat_size) + [self.hidden_size]

    def proj_feat(self, x):
        """[B, N, hidden] -> feature map.  In NDHWC the reference's view+permute+contiguous (118-122) is a free view."""
        return x.view([x.size(0)] + self.proj_view_shape)

    def forward(self, x_in, x_cat=None):
        """x_cat: optional (a, b) pair with cat((a, b)) == x_in (virtual concat for skip1's 3x3x3 convolution)."""
        i = self.num_layers // 4
        run, out_encoder_1 = run_vit_beside(self.vit, x_in, lambda: self.skip1(x_in, x_cat), x_cat or ())
        with small_blocks_beside(x_in, run) as join:
            out_encoder_2 = self.skip2(self.proj_feat(run.hidden(i)))
            out_encoder_3 = self.skip3(self.proj_feat(run.hidden(i * 2)))
            out_encoder_4 = self.skip4(self.proj_feat(run.hidden(i * 3)))
            join(out_encoder_2, out_encoder_3, out_encoder_4)
        out_encoder_5 = self.proj_feat(run.final(out_encoder_1, out_encoder_2, out_encoder_3, out_encoder_4))
        return [out_encoder_1, out_encoder_2, out_encoder_3, out_encoder_4, out_encoder_5]


class PyMSCDecoder(nn.Module):
    """dose_pyfer.PyMSCDecoder (150-239)."""

    def __init__(self, feature_size: int = 16, hidden_size: int = 768, norm_name: Union[Tuple, str] = "instance",
                 spatial_dims: int = 3, mode_multi: bool = False, act="relu", multiS_conv=True) -> None:
        super().__init__()
        chans = [(hidden_size, feature_size * 8), (feature_size * 8, feature_size * 4), (feature_size * 4, feature_size * 2),
                 (feature_size * 2, feature_size)]
        for lvl, (cin, cout) in zip((4, 3, 2, 1), chans):
            if mode_multi:
                blk = ModifiedUnetrUpBlock(spatial_dims=spatial_dims, in_channels=cin, out_channels=cout, upsample_kernel_size=2,
                                           act=act, multiS_conv=multiS_conv)
            else:
                blk = UnetrUpBlock(spatial_dims=spatial_dims, in_channels=cin, out_channels=cout, upsample_kernel_size=2,
                                   kernel_size=3, norm_name=norm_name)
            setattr(self, f"decoder{lvl}", blk)

    def forward(self, out_encoder):
        e1, e2, e3, e4, e5 = out_encoder
        from .. import config
        if config.activation_checkpointing() and torch.is_grad_enabled():
            from torch.utils.checkpoint import checkpoint

            def run(blk, a, b):
                calls = []

                def once(a_, b_):
                    # the recomputation pass (second call) must not update the BatchNorm running statistics a second time
                    calls.append(1)
                    with config.bn_buffer_updates(len(calls) == 1):
                        return blk(a_, b_)
                return checkpoint(once, a, b, use_reentrant=False)
        else:
            run = lambda blk, a, b: blk(a, b)  # noqa: E731
        dec4 = run(self.decoder4, e5, e4)
        dec3 = run(self.decoder3, dec4, e3)
        dec2 = run(self.decoder2, dec3, e2)
        dec1 = run(self.decoder1, dec2, e1)
        return [dec1, dec2, dec3, dec4]


class MainSubsetModel(nn.Module):
    """dose_pyfer.MainSubsetModel (245-319): ViT encoder + pyramid multi-scale decoder + 4 deep-supervision heads.
    ``self.out`` exists (state_dict) but is never used in forward, exactly as in the reference (301-305, 311-319)."""

    def __init__(self, in_ch, out_ch, img_size, feature_size: int = 16, hidden_size: int = 768, mlp_dim: int = 3072,
                 num_heads: int = 12, num_layers: int = 12, conv_block: bool = True, res_block: bool = True,
                 dropout_rate: float = 0.0, mode_multi_dec=False, act="relu", multiS_conv=True):
        super().__init__()
        self.encoder = ViTEncoder(in_channels=in_ch, img_size=img_size, feature_size=feature_size, hidden_size=hidden_size,
                                  mlp_dim=mlp_dim, num_heads=num_heads, num_layers=num_layers, pos_embed="perceptron",
                                  norm_name="instance", res_block=res_block, conv_block=conv_block, dropout_rate=dropout_rate)
        self.decoder = PyMSCDecoder(feature_size=feature_size, hidden_size=hidden_size, mode_multi=mode_multi_dec, act=act,
                                    multiS_conv=multiS_conv)

        def to_out(in_feature):
            return nn.Sequential(nn.Conv3d(in_feature, out_ch, kernel_size=1, padding=0, bias=True))

        self.dose_convertors = nn.ModuleList([to_out(feature_size)])
        for i in range(1, 4):
            self.dose_convertors.append(to_out(int(feature_size * np.power(2, i))))
        self.out = nn.Sequential(nn.Conv3d(feature_size, out_ch, kernel_size=1, padding=0, bias=True))

    def update_config(self, config_hparam):
        self.encoder.hidden_size = config_hparam["hidden_size"]
        self.encoder.num_layers = config_hparam["hidden_size"]

    def forward_ndhwc(self, x, x_cat=None):
        out_decoders = self.decoder(self.encoder(x, x_cat))
        return [ops.conv3d(d, conv[0].weight, conv[0].bias) for d, conv in zip(out_decoders, self.dose_convertors)]

    def forward(self, x):
        return [from_ndhwc(o) for o in self.forward_ndhwc(to_ndhwc(x))]


class Model(nn.Module):
    """dose_pyfer.Model (325-360): net_A (C3D U-Net) -> cat(out_A, x) -> net_B; returns [output_A, out_net_B]."""

    def __init__(self, in_ch, out_ch, list_ch_A, feature_size=16, img_size=(128, 128, 128), num_layers=8, num_heads=6,
                 act="mish", mode_multi_dec=True, multiS_conv=True):
        super().__init__()
        self.net_A = BaseUNet(in_ch, list_ch_A)
        self.net_B = MainSubsetModel(in_ch=in_ch + list_ch_A[1], out_ch=out_ch, feature_size=feature_size, img_size=img_size,
                                     num_layers=num_layers, num_heads=num_heads, act=act, mode_multi_dec=mode_multi_dec,
                                     multiS_conv=multiS_conv)
        self.conv_out_A = nn.Conv3d(list_ch_A[1], out_ch, kernel_size=1, padding=0, bias=True)

    def forward(self, x):
        return self.forward_staged(to_ndhwc(x))

    def forward_staged(self, xh):
        """forward() on an input that already is NDHWC in the compute dtype with its 9 channels zero-padded to 16 (what
        cascade.cascade_structures(..., staged=True) builds in place): saves the NDHWC -> NCDHW fp32 -> NDHWC round trip of the
        cascade glue (train_light_linked_model.py:165-168 hands the concatenated structures straight to the dose network)."""
        out_net_A = self.net_A.forward_ndhwc(xh)
        out_net_B = self.net_B.forward_ndhwc(ops.cat((out_net_A, xh)), (out_net_A, xh))
        output_A = ops.conv3d(out_net_A, self.conv_out_A.weight, self.conv_out_A.bias)
        return [from_ndhwc(output_A), [from_ndhwc(o) for o in out_net_B]]


def create_pretrained_unet(ckpt_file, in_ch, out_ch, list_ch_A, feature_size, img_size, num_layers=8, num_heads=6, act="mish",
                           mode_multi_dec=True, multiS_conv=True):
    """dose_pyfer.create_pretrained_unet (363-407): load by key intersection, strict=False."""
    pretrain = torch.load(ckpt_file, map_location="cpu")
    net = Model(in_ch, out_ch, list_ch_A, feature_size=feature_size, img_size=img_size, num_layers=num_layers,
                num_heads=num_heads, act=act, mode_multi_dec=mode_multi_dec, multiS_conv=multiS_conv)
    net_dict = net.state_dict()
    inside = tuple({k for k in pretrain["network_state_dict"] if k in net_dict.keys()})
    pretrain["network_state_dict"] = {k: v for k, v in pretrain["network_state_dict"].items() if k in net_dict.keys()}
    net.load_state_dict(pretrain["network_state_dict"], strict=False)
    return net, inside
