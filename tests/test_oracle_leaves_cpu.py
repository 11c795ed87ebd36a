"""Independent cross-checks of the oracle's MONAI-leaf restatements ("parity unpinned" for a9-a11 because MONAI is
absent).  torch's OWN modules implement the same published algorithms -- nn.MultiheadAttention (SABlock),
nn.TransformerEncoderLayer(norm_first=True, activation='gelu') (MONAI TransformerBlock: pre-norm, GELU MLP), and a plain
nn.Conv3d / nn.InstanceNorm3d / nn.LeakyReLU composition (UnetResBlock, UnetrPrUpBlock) -- so the oracle is checked against
code that neither this repository nor the reference wrote.  This does not pin MONAI 0.7.0 itself (still absent); it removes the
possibility that the oracle's leaves mis-state the standard algorithms they name."""
import torch
import torch.nn as nn

import oracle
from helpers import rel_err


def _rnd(shape, seed, s=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64) * s


def test_sablock_restatement_matches_nn_multihead_attention():
    """MONAI SABlock: qkv = Linear(h, 3h, bias=False) split "(qkv l d)", softmax(q k^T d^-1/2) v, out_proj == nn.MultiheadAttention
    with in_proj_weight = qkv.weight (rows [q | k | v], each "(l d)"), zero in_proj_bias, the same out_proj."""
    for (B, N, H, heads) in ((2, 24, 48, 6), (1, 64, 96, 12), (3, 7, 32, 4)):
        x = _rnd((B, N, H), 1)
        qkv_w, out_w, out_b = _rnd((3 * H, H), 2, H ** -0.5), _rnd((H, H), 3, H ** -0.5), _rnd((H,), 4, 0.1)
        mha = nn.MultiheadAttention(H, heads, dropout=0.0, bias=True, batch_first=True).double()
        with torch.no_grad():
            mha.in_proj_weight.copy_(qkv_w)
            mha.in_proj_bias.zero_()
            mha.out_proj.weight.copy_(out_w)
            mha.out_proj.bias.copy_(out_b)
        xr = x.clone().requires_grad_(True)
        ref = mha(xr, xr, xr, need_weights=False)[0]
        xo = x.clone().requires_grad_(True)
        wq = qkv_w.clone().requires_grad_(True)
        got = oracle.attention(xo, wq, out_w, out_b, heads)
        assert rel_err(got, ref) < 1e-12
        r = _rnd(ref.shape, 5)
        ref.backward(r)
        got.backward(r)
        assert rel_err(xo.grad, xr.grad) < 1e-11 and rel_err(wq.grad, mha.in_proj_weight.grad) < 1e-11


def test_vit_block_restatement_matches_nn_transformer_encoder_layer():
    """MONAI TransformerBlock: x + attn(norm1(x)); x + mlp(norm2(x)) with mlp = Linear -> GELU(erf) -> Linear ==
    nn.TransformerEncoderLayer(norm_first=True, activation='gelu', dropout=0).  Runs the oracle's ViT with an identity patch
    embedding stub removed: blocks only, via oracle.vit's block body."""
    B, N, H, heads, mlp = 2, 16, 48, 6, 96
    layer = nn.TransformerEncoderLayer(H, heads, dim_feedforward=mlp, dropout=0.0, activation="gelu", batch_first=True,
                                       norm_first=True).double()
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(_rnd(p.shape, 10 + p.numel() % 97, 0.2))
        layer.self_attn.in_proj_bias.zero_()          # SABlock's qkv has no bias
        layer.norm1.weight.add_(1.0)
        layer.norm2.weight.add_(1.0)
    sd = {"b.norm1.weight": layer.norm1.weight, "b.norm1.bias": layer.norm1.bias, "b.norm2.weight": layer.norm2.weight,
          "b.norm2.bias": layer.norm2.bias, "b.attn.qkv.weight": layer.self_attn.in_proj_weight,
          "b.attn.out_proj.weight": layer.self_attn.out_proj.weight, "b.attn.out_proj.bias": layer.self_attn.out_proj.bias,
          "b.mlp.linear1.weight": layer.linear1.weight, "b.mlp.linear1.bias": layer.linear1.bias,
          "b.mlp.linear2.weight": layer.linear2.weight, "b.mlp.linear2.bias": layer.linear2.bias}
    sd = {k: v.detach() for k, v in sd.items()}
    x = _rnd((B, N, H), 7)
    ref = layer(x)
    got = oracle.transformer_block(sd, "b.", x, heads)
    assert rel_err(got, ref.detach()) < 1e-12


def test_unet_res_block_restatement_matches_torch_modules():
    """MONAI UnetResBlock (norm 'instance' non-affine, LeakyReLU 0.01, bias-free convs, 1x1x1 + IN on the residual when
    Cin != Cout) and UnetrPrUpBlock's ConvTranspose3d(k2,s2) against torch's own layers."""
    for cin, cout in ((5, 4), (4, 4)):
        c1, c2, c3 = nn.Conv3d(cin, cout, 3, 1, 1, bias=False).double(), nn.Conv3d(cout, cout, 3, 1, 1, bias=False).double(), \
            nn.Conv3d(cin, cout, 1, 1, 0, bias=False).double()
        n1, n2, n3, act = nn.InstanceNorm3d(cout), nn.InstanceNorm3d(cout), nn.InstanceNorm3d(cout), nn.LeakyReLU(0.01)
        x = _rnd((2, cin, 6, 5, 7), 3)
        out = n2(c2(act(n1(c1(x)))))
        res = n3(c3(x)) if cin != cout else x
        ref = act(out + res)
        sd = {"p.conv1.conv.weight": c1.weight.detach(), "p.conv2.conv.weight": c2.weight.detach(), "p.conv3.conv.weight": c3.weight.detach()}
        assert rel_err(oracle.unet_res_block(sd, "p", x), ref.detach()) < 1e-12
    t = nn.ConvTranspose3d(6, 3, 2, 2, bias=False).double()
    x = _rnd((1, 6, 3, 4, 2), 4)
    assert rel_err(oracle.conv_transpose3d_k2s2(x, t.weight.detach()), t(x).detach()) < 1e-12


def test_storage_emulation_is_identity_without_context_and_rounds_within():
    x = _rnd((2, 8), 1).float()
    assert torch.equal(oracle.store(x), x)
    with oracle.storage(torch.bfloat16):
        assert torch.equal(oracle.store(x), x.bfloat16().float())
        with oracle.storage(None):
            assert torch.equal(oracle.store(x), x)
    assert torch.equal(oracle.store(x), x)


def test_dice_ce_restatement_matches_a_per_class_loop_and_torch_cross_entropy():
    """oracle.dice_ce_loss (MONAI 0.7.0 DiceCELoss(to_onehot_y=True, softmax=True), train_light_transeg.py:148): the cross-entropy half
    is torch's own F.cross_entropy; the Dice half is re-derived here as an explicit loop over (sample, class) from the published
    formula 1 - (2 sum(p y) + 1e-5) / (sum(y) + sum(p) + 1e-5), mean over samples and classes (background included)."""
    B, C, S = 2, 8, (5, 6, 7)
    z = _rnd((B, C) + S, 21)
    lab = torch.randint(0, C, (B, 1) + S, generator=torch.Generator().manual_seed(22)).double()
    zr = z.clone().requires_grad_(True)
    p = torch.softmax(zr, 1)
    dice = 0.0
    for b in range(B):
        for c in range(C):
            y = (lab[b, 0] == c).double()
            dice = dice + (1.0 - (2.0 * (p[b, c] * y).sum() + 1e-5) / (y.sum() + p[b, c].sum() + 1e-5))
    ref = dice / (B * C) + torch.nn.functional.cross_entropy(zr, lab[:, 0].long())
    zo = z.clone().requires_grad_(True)
    got = oracle.dice_ce_loss(zo, lab)
    assert abs(float(got) - float(ref)) < 1e-12
    ref.backward()
    got.backward()
    assert rel_err(zo.grad, zr.grad) < 1e-11
    # labels without the channel axis and integer labels are the same thing
    assert abs(float(oracle.dice_ce_loss(z, lab[:, 0].long()[:, None])) - float(ref)) < 1e-12
