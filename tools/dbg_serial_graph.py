import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kgdet_amd import build_detector, configs, synthetic
cfg = configs.reppoints_kp_r50_fpn(soft_nms=True)
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().eval()
a = synthetic.make_batch(2, torch.device('cuda'), seed=0, img_shape=(384, 500, 3), pad_shape=(384, 512, 3))
b = synthetic.make_batch(2, torch.device('cuda'), seed=5, img_shape=(384, 500, 3), pad_shape=(384, 512, 3))
synthetic.calibrate_scores_serial(model, a, cfg.test_cfg.score_thr, 0.004)
run = model.graphed_test_batch(a['img'], a['img_meta'], rescale=True)
for name, batch in (('a', a), ('b', b), ('a', a)):
    got = run(batch['img'])
    with torch.no_grad():
        want = model.simple_test_batch(batch['img'], batch['img_meta'], rescale=True)
        want2 = model.simple_test_batch(batch['img'], batch['img_meta'], rescale=True)
    for i, (g, w, w2) in enumerate(zip(got, want, want2)):
        if len(w) != 3:
            print(name, i, 'empty'); continue
        for c in range(13):
            if len(w[0][c]) == 0: continue
            db = np.abs(g[0][c] - w[0][c]).max() if g[0][c].shape == w[0][c].shape else 'shape'
            dk = np.abs(g[2][c] - w[2][c]).max() if g[2][c].shape == w[2][c].shape else 'shape'
            dk2 = np.abs(w2[2][c] - w[2][c]).max()
            if (isinstance(dk, str) or dk > 0 or isinstance(db, str) or db > 0):
                nz = np.argwhere(np.abs(g[2][c] - w[2][c]) > 0)[:6] if not isinstance(dk, str) else None
                print(name, 'img', i, 'class', c, 'n', len(w[0][c]), 'box diff', db, 'kp diff', dk, 'eager-vs-eager', dk2, 'where', None if nz is None else nz.tolist(), 'scale', np.abs(w[2][c]).max())
