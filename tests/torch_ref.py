"""Independent pure-PyTorch formulations used as a second opinion on the oracle (tests only).

They share no code with oracle/: deformable sampling is written with F.grid_sample
(align_corners=True, zero padding == the reference's zero-padded bilinear with its (-1, H) range
guard), contraction with einsum, gradients by autograd.
"""
import torch
import torch.nn.functional as F


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def deform_sample(x, offset, kh, kw, stride=1, padding=0, dilation=1, deformable_groups=1, mask=None):
    """Returns samples [N, C, K, Ho, Wo]."""
    N, C, H, W = x.shape
    sh, sw = _pair(stride); ph, pw = _pair(padding); dh, dw = _pair(dilation)
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    K = kh * kw
    DG = deformable_groups
    oy = torch.arange(Ho, dtype=x.dtype, device=x.device).view(1, 1, Ho, 1) * sh - ph
    ox = torch.arange(Wo, dtype=x.dtype, device=x.device).view(1, 1, 1, Wo) * sw - pw
    ki = (torch.arange(kh, dtype=x.dtype, device=x.device) * dh).repeat_interleave(kw).view(1, K, 1, 1)
    kj = (torch.arange(kw, dtype=x.dtype, device=x.device) * dw).repeat(kh).view(1, K, 1, 1)
    off = offset.view(N, DG, K, 2, Ho, Wo)
    out = []
    cpg = C // DG
    for g in range(DG):
        py = oy + ki + off[:, g, :, 0]          # [N, K, Ho, Wo]
        px = ox + kj + off[:, g, :, 1]
        gx = 2 * px / max(W - 1, 1) - 1 if W > 1 else px * 0
        gy = 2 * py / max(H - 1, 1) - 1 if H > 1 else py * 0
        grid = torch.stack([gx, gy], -1).view(N, K * Ho, Wo, 2)
        s = F.grid_sample(x[:, g * cpg:(g + 1) * cpg], grid, mode='bilinear', padding_mode='zeros',
                          align_corners=True).view(N, cpg, K, Ho, Wo)
        if mask is not None:
            s = s * mask.view(N, DG, K, Ho, Wo)[:, g].unsqueeze(1)
        out.append(s)
    return torch.cat(out, 1)


def deform_conv(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                mask=None, bias=None):
    O, Cg, kh, kw = weight.shape
    s = deform_sample(x, offset, kh, kw, stride, padding, dilation, deformable_groups, mask)
    N, C, K, Ho, Wo = s.shape
    s = s.view(N, groups, Cg, K, Ho, Wo)
    w = weight.view(groups, O // groups, Cg, K)
    out = torch.einsum('ngckhw,gock->ngohw', s, w).reshape(N, O, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def py_sigmoid_focal_loss(pred, target, gamma=2.0, alpha=0.25):
    """Element-wise [N, C] focal loss from labels in {0..C} (0 = background): the formula of the
    reference's debugging helper (mmdet/models/losses/focal_loss.py:10-25) on one-hot targets."""
    N, C = pred.shape
    t = torch.zeros_like(pred)
    pos = target > 0
    t[pos, target[pos] - 1] = 1
    p = pred.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    fw = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    return F.binary_cross_entropy_with_logits(pred, t, reduction='none') * fw
