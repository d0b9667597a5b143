"""Generate tests/golden/demo_dets_golden.npz (build container only; needs /root/reference):

BASELINE config 1 -- the demo config on the 32-image demo set -- run through the REFERENCE'S OWN detector
(mmdet/models/backbones/resnet.py + necks/fpn2.py + anchor_heads/reppoints_head_kp3rep_cas_1_assign_once.py +
detectors/reppoints_detector_kp.py, imported in place by ref_loader.load_detector(); deformable conv = the test-side
CPU formulation, NMS = the compiled reference nms_cpu.cpp) on the CPU, the path README.md:67-74 /
tools/test.py:61-100 describe, then results2json + COCOeval exactly as coco_utils.py:121-216 does, with the
reference's own pycocotools.  Weights: this repo's seeded detector state_dict loaded strictly into the reference
detector.  Stored per image: detections (boxes, scores, labels, the landmarks of the detection's category) and the
bbox / keypoint AP statistics.

    python tests/golden/make_demo_golden.py [float64]
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from kgdet_amd import evaluation  # noqa: E402
from tests.golden import demo_cases, ref_loader  # noqa: E402


def main():
    dt = torch.float64 if 'float64' in sys.argv[1:] else torch.float32
    ns = ref_loader.load_detector()
    cfg, ours = demo_cases.demo_detector()
    mcfg = {k: v for k, v in cfg.model.items()}
    mcfg['pretrained'] = None
    ref = ns.build_detector(dict(mcfg), train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    res = ref.load_state_dict(ours.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    ref = ref.to(dt).eval()
    data = demo_cases.demo_dataset(test_mode=True)
    out, results = {}, []
    t0 = time.time()
    with torch.no_grad():
        for idx in range(len(data)):
            d = data[idx]
            r = ref(return_loss=False, rescale=True, img=[d['img'][0][None].to(dt)], img_meta=[[d['img_meta'][0]]])
            results.append(r)
            if len(r) == 3:
                boxes = np.concatenate(r[0]).astype(np.float32)
                labels = np.concatenate([np.full(len(b), c, np.int64) for c, b in enumerate(r[0])])
                kpts = np.concatenate(r[2]).astype(np.float32)
            else:
                boxes, labels, kpts = np.zeros((0, 5), np.float32), np.zeros(0, np.int64), np.zeros((0, 882), np.float32)
            out['img%d:bboxes' % idx], out['img%d:labels' % idx] = boxes, labels
            sl = demo_cases.category_slices(data)
            out['img%d:kpts' % idx] = np.stack([np.pad(k[3 * sl[c + 1][0]:3 * sl[c + 1][1]], (0, 117))[:117]
                                                for k, c in zip(kpts, labels)]) if len(labels) else np.zeros((0, 117), np.float32)
            print('image %d %s: %d detections (%.0f s)' % (idx, tuple(d['img'][0].shape), len(labels), time.time() - t0),
                  flush=True)
    # the reference's own writers + evaluator (coco_utils.py results2json / coco_eval flow, pycocotools in place)
    from oracle import build_ref
    COCO, COCOeval = build_ref.load_reference_evaluator()
    cu = ns.coco_utils
    with tempfile.TemporaryDirectory() as tmp:
        class _DS(object):                          # what coco_utils.kpt2json reads of a dataset
            img_ids, cat_ids = data.img_ids, data.cat_ids
            def __len__(self):
                return len(data)
        files = cu.results2json(_DS(), results, os.path.join(tmp, 'ref'))
        coco = COCO(demo_cases.ANN)
        for typ in ('bbox', 'keypoints'):
            ev = COCOeval(coco, coco.loadRes(files[typ]), typ)
            ev.params.imgIds = coco.getImgIds()
            ev.evaluate(), ev.accumulate(), ev.summarize()
            out['stats:' + typ] = np.asarray(ev.stats, np.float64)
        # the same through this repo's writers + evaluator (already pinned to the reference evaluator): must agree
        mine = evaluation.coco_eval(evaluation.results2json(data, results, os.path.join(tmp, 'mine')),
                                    ['bbox', 'keypoints'], data.coco, verbose=False)
        for typ in ('bbox', 'keypoints'):
            assert np.allclose(mine[typ], out['stats:' + typ], atol=1e-9, equal_nan=True), (typ, mine[typ], out['stats:' + typ])
    out['dtype'] = np.array(str(dt))
    path = os.path.join(HERE, 'demo_dets_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes; detections per image',
          [int(out['img%d:labels' % i].shape[0]) for i in range(len(data))])


if __name__ == '__main__':
    main()
