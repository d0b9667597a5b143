import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kgdet_amd import build_detector, configs, synthetic
cfg = configs.reppoints_kp_r50_fpn(soft_nms=True)
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().eval()
a = synthetic.make_batch(2, torch.device('cuda'), seed=0, img_shape=(384, 500, 3), pad_shape=(384, 512, 3))
synthetic.calibrate_scores_serial(model, a, cfg.test_cfg.score_thr, 0.004)
metas = [dict(m) for m in a['img_meta']]
metas[1].update(img_shape=(300, 480, 3), scale_factor=1.5)
head = model.bbox_head
with torch.no_grad():
    outs = head(model.extract_feat(a['img']), metas)
    want = head.get_bboxes(*(outs + (metas, cfg.test_cfg, True)))
    got = head.get_bboxes_numpy(*(outs + (metas, cfg.test_cfg, True)))
for i, ((gd, gl, gk), (wd, wl, wk)) in enumerate(zip(got, want)):
    wd, wl, wk = wd.cpu().numpy(), wl.cpu().numpy(), wk.reshape(wk.shape[0], -1).cpu().numpy()
    print('image', i, gd.shape, wd.shape, 'labels equal', np.array_equal(gl, wl) if gl.shape == wl.shape else 'shape')
    if gd.shape == wd.shape:
        d = np.abs(gd - wd)
        print(' max box diff', d[:, :4].max(), 'max score diff', d[:, 4].max(), 'rows differing', (d.max(1) > 0).sum(), 'kp diff', np.abs(gk - wk).max())
        bad = np.nonzero(d.max(1) > 0)[0][:5]
        for r in bad:
            print('  row', r, gd[r], wd[r], gl[r], wl[r])
    else:
        print(' got labels', np.bincount(gl, minlength=13), 'want', np.bincount(wl, minlength=13))
        print(gd[:5], wd[:5])
