"""Seeded synthetic OpenKBP-like volumes (no dataset exists on any box): value ranges follow the reference's loader
(DosePrediction/DataLoader/dataloader_OpenKBP_monai.py:116-146,195-197,258); see SURVEY.md section 8d."""
import torch


def _grid(shape):
    axes = [torch.linspace(-1, 1, s) for s in shape]
    return torch.meshgrid(*axes, indexing="ij")


def dose_input(batch, shape, seed=1234):
    """Input [B,9,D,H,W]: ch0 PTV in {0,.8,.9,1}; ch1-7 binary OAR blobs; ch8 CT/1000 in [-1.024,1.5]."""
    g = torch.Generator().manual_seed(seed)
    z, y, x = _grid(shape)
    out = torch.zeros((batch, 9) + tuple(shape))
    for b in range(batch):
        c = (torch.rand(3, generator=g) - 0.5) * 0.4
        r2 = ((z - c[0]) / 0.9) ** 2 + ((y - c[1]) / 0.8) ** 2 + ((x - c[2]) / 0.7) ** 2
        body = r2 < 1.0
        ptv = torch.zeros(shape)
        for lvl, rad in ((0.8, 0.35), (0.9, 0.25), (1.0, 0.15)):
            ptv[r2 < rad ** 2] = lvl
        out[b, 0] = ptv
        for o in range(7):
            cc = (torch.rand(3, generator=g) - 0.5) * 1.2
            rr = 0.08 + 0.1 * torch.rand(1, generator=g).item()
            blob = ((z - cc[0]) ** 2 + (y - cc[1]) ** 2 + (x - cc[2]) ** 2) < rr ** 2
            out[b, 1 + o] = (blob & body & (out[b, 1:1 + o].sum(0) == 0)).float()
        ct = torch.full(shape, -1.024)
        tissue = 0.03 + 0.03 * torch.randn(shape, generator=g)
        ct[body] = tissue[body]
        bone = (torch.rand(shape, generator=g) > 0.985) & body
        ct[bone] = 0.3 + 1.2 * torch.rand(shape, generator=g)[bone]
        out[b, 8] = ct.clamp_(-1.024, 1.5)
    return out


def dose_target(batch, shape, seed=5678):
    """GT [B,2,D,H,W]: ch0 dose/70 in [0,1.1] with a smooth fall-off; ch1 possible-dose mask (~35 % ones)."""
    g = torch.Generator().manual_seed(seed)
    z, y, x = _grid(shape)
    out = torch.zeros((batch, 2) + tuple(shape))
    for b in range(batch):
        c = (torch.rand(3, generator=g) - 0.5) * 0.4
        r = torch.sqrt((z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2)
        out[b, 0] = (1.1 * torch.exp(-(r / 0.45) ** 2)).clamp_(0, 1.1)
        out[b, 1] = (r < 0.78).float()
    return out


def ct_input(batch, shape, seed=1234):
    return dose_input(batch, shape, seed)[:, 8:9].contiguous()
