"""Which memory format do the feature maps have on their way through config 5's inference chain?  python tools/dbg_serial_layout.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kgdet_amd import build_detector, configs, synthetic
def fmt(t):
    return 'NHWC' if (t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)) else ('NCHW' if t.is_contiguous() else 'other')
for name, cfg in (('kgdet', configs.kgdet_r50_fpn()), ('serial', configs.reppoints_kp_r50_fpn())):
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().eval()
    batch = synthetic.make_batch(2, 'cuda', seed=0)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        c = model.backbone(batch['img'])
        print(name, 'backbone:', [(tuple(f.shape), str(f.dtype)[6:], fmt(f)) for f in c])
        n = model.neck(c)
        print(name, 'neck    :', [(tuple(f.shape), str(f.dtype)[6:], fmt(f)) for f in n])
        neck = model.neck
        lat = neck.lateral_convs[0](c[neck.start_level])
        print(name, 'lateral0:', fmt(lat), ' interpolate:', fmt(torch.nn.functional.interpolate(lat, scale_factor=2, mode='nearest')),
              ' fpn_conv0:', fmt(neck.fpn_convs[0](lat)), type(neck.fpn_convs[0].conv).__name__, neck.fpn_convs[0].conv.bias is not None)
