"""Do the split convolution kernels take maps with an odd pixel count directly (no zero column)?  forward / grad_input / grad_weight of
1x1 and 3x3 convolutions on 13x21 and 7x11 maps against torch's fp32 convolution:  python tools/check_odd_maps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from kgdet_amd import conv1x1
torch.backends.cudnn.allow_tf32 = False
for (H, W) in ((13, 21), (7, 11), (25, 42), (5, 5)):
    for k in (1, 3):
        torch.manual_seed(H * 10 + k)
        x = torch.randn(2, 256, H, W, device='cuda', requires_grad=True)
        w = (torch.randn(256, 256, k, k, device='cuda') * 0.05).requires_grad_()
        g = torch.randn(2, 256, H, W, device='cuda')
        ref = F.conv2d(x.double(), w.double(), padding=k // 2)
        gx_r, gw_r = torch.autograd.grad(ref, (x, w), g.double())
        y = conv1x1.conv_split(x, w)
        gx, gw = torch.autograd.grad(y, (x, w), g)
        e = [float((a.double() - b).abs().max() / b.abs().max()) for a, b in ((y, ref), (gx, gx_r), (gw, gw_r))]
        print('%dx%d k=%d: forward %.1e  grad_input %.1e  grad_weight %.1e' % (H, W, k, *e), flush=True)
    # the biased convolution + ReLU of config 5's FPN / head (conv1x1._ConvBiasAct)
    conv = torch.nn.Conv2d(256, 256, 3, padding=1).cuda()
    x = torch.randn(2, 256, H, W, device='cuda', requires_grad=True)
    g = torch.randn(2, 256, H, W, device='cuda')
    ref = F.relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1))
    r = torch.autograd.grad(ref, (x, conv.weight, conv.bias), g.double())
    y = conv1x1.conv_bias_act(conv, x, relu=True)
    o = torch.autograd.grad(y, (x, conv.weight, conv.bias), g)
    e = [float((a.double() - b).abs().max() / b.abs().max()) for a, b in ((y, ref), (o[0], r[0]), (o[1], r[1]), (o[2], r[2]))]
    print('%dx%d conv + bias + relu: forward %.1e  grad_input %.1e  grad_weight %.1e  grad_bias %.1e' % (H, W, *e), flush=True)
