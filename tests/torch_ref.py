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


def deform_psroi_pool(data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                      sample_per_part=4, trans_std=0.0):
    """Deformable PS-RoI pooling written as dense tensor algebra + ``grid_sample`` (fp64-capable, differentiable):
    an independent formulation of deform_pool_cuda_kernel.cu:53-140 for pinning the C oracle and, through autograd,
    its backward (:143-263).  Clamped bilinear sampling == grid_sample(border padding) on clamped coordinates; samples
    outside [-0.5, dim - 0.5] are masked out of the average."""
    import torch
    import torch.nn.functional as F
    P, S, G = out_size, sample_per_part, group_size
    part_size = P if part_size is None else part_size
    B, C, H, W = data.shape
    R = rois.shape[0]
    dt = data.dtype
    b = rois[:, 0].long()
    # C's round() (half away from zero) of the RoI corners
    rnd = lambda v: torch.sign(v) * torch.floor(torch.abs(v) + 0.5)
    rsw, rsh = rnd(rois[:, 1]) * spatial_scale - 0.5, rnd(rois[:, 2]) * spatial_scale - 0.5
    rew, reh = (rnd(rois[:, 3]) + 1) * spatial_scale - 0.5, (rnd(rois[:, 4]) + 1) * spatial_scale - 0.5
    rw, rh = (rew - rsw).clamp(min=0.1), (reh - rsh).clamp(min=0.1)
    bw, bh = rw / P, rh / P
    ar = torch.arange(P)
    part = torch.floor(ar.to(dt) / P * part_size).long()                       # bin -> offset cell
    grp = torch.clamp(torch.floor(ar.to(dt) * G / P).long(), 0, G - 1)         # bin -> position-sensitive group
    ctop = torch.arange(out_channels)
    if no_trans:
        tx = ty = data.new_zeros(R, out_channels, P, P)
    else:
        ncls = offset.shape[1] // 2
        each = out_channels // ncls
        cls = ctop // each
        o = offset[:, :, part][:, :, :, part]                                  # [R, 2*ncls, P(ph), P(pw)]
        tx, ty = o[:, 2 * cls] * trans_std, o[:, 2 * cls + 1] * trans_std       # [R, out_c, P, P]
    ws = ar.view(1, 1, 1, P).to(dt) * bw.view(R, 1, 1, 1) + rsw.view(R, 1, 1, 1) + tx * rw.view(R, 1, 1, 1)
    hs = ar.view(1, 1, P, 1).to(dt) * bh.view(R, 1, 1, 1) + rsh.view(R, 1, 1, 1) + ty * rh.view(R, 1, 1, 1)
    s = torch.arange(S).to(dt)
    w = ws.unsqueeze(-1).unsqueeze(-1) + (s.view(1, S) * (bw / S).view(R, 1, 1, 1, 1, 1))      # [R,oc,P,P,1->S(ih),S(iw)]
    h = hs.unsqueeze(-1).unsqueeze(-1) + (s.view(S, 1) * (bh / S).view(R, 1, 1, 1, 1, 1))
    w, h = w.expand(R, out_channels, P, P, S, S), h.expand(R, out_channels, P, P, S, S)
    ok = ((w >= -0.5) & (w <= W - 0.5) & (h >= -0.5) & (h <= H - 0.5)).to(dt)
    wc, hc = w.clamp(0, W - 1), h.clamp(0, H - 1)
    # input channel of (ctop, ph, pw)
    chan = (ctop.view(-1, 1, 1) * G + grp.view(1, P, 1)) * G + grp.view(1, 1, P)                # [oc, P, P]
    out = data.new_zeros(R, out_channels, P, P)
    cnt = ok.sum((-1, -2))
    for r in range(R):
        planes = data[b[r]][chan]                                               # [oc, P, P, H, W]
        planes = planes.reshape(1, out_channels * P * P, H, W)
        gx = 2 * wc[r] / max(W - 1, 1) - 1
        gy = 2 * hc[r] / max(H - 1, 1) - 1
        grid = torch.stack([gx, gy], -1).reshape(out_channels * P * P, S * S, 1, 2)
        # one grid per plane: treat planes as the batch
        val = F.grid_sample(planes.permute(1, 0, 2, 3), grid, mode='bilinear', padding_mode='border', align_corners=True)
        val = val.reshape(out_channels, P, P, S, S) * ok[r]
        out[r] = torch.where(cnt[r] > 0, val.sum((-1, -2)) / cnt[r].clamp(min=1), val.new_zeros(()))
    return out, cnt
