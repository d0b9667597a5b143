"""Extract the DeepFashion2 landmark definitions (per-category landmark ranges, left/right swap pairs, landmark
groups, OKS sigmas) from the reference checkout into kgdet_amd/data/deepfashion2_landmarks.json.

They are dataset constants (DeepFashion2's 294-landmark scheme), stored here in one normalised form: everything is
a 0-based GLOBAL landmark index.  Sources: mmdetection/mmdet/datasets/deepfashion2.py:9-99 (names, ranges, pairs,
groups) and deepfashion2_api/PythonAPI/pycocotools/cocoeval.py:206-243 (sigmas).  Run once, in the dev container."""
import ast
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
src = open(os.path.join(REF, 'mmdetection/mmdet/datasets/deepfashion2.py')).read()
tree = ast.parse(src)
found = {}
for node in ast.walk(tree):
    if isinstance(node, ast.Assign):
        t = node.targets[0]
        name = t.attr if isinstance(t, ast.Attribute) else getattr(t, 'id', None)
        if name in ('CLASSES', 'gt_class_keypoints_dict', 'keypoint_groups', '_flip_pairs') and name not in found:
            found[name] = ast.literal_eval(node.value)
ranges = [list(found['gt_class_keypoints_dict'][c + 1]) for c in range(13)]
swap = []   # per category: the left/right landmark pairs exchanged by a horizontal flip
for c, pairs in enumerate(found['_flip_pairs']):
    base = ranges[c][0] - 1
    swap.append([[a + base, b + base] for a, b in pairs])
groups = [[k - 1 for k in g] for g in found['keypoint_groups']]
ev = open(os.path.join(REF, 'deepfashion2_api/PythonAPI/pycocotools/cocoeval.py')).read()
m = re.search(r'sigmas = np\.array\(\[(.*?)\]\)', ev, re.S)
sigmas = [float(v) for v in m.group(1).replace('\n', ' ').split(',')]
assert len(sigmas) == 294 and ranges[-1][1] == 294
out = dict(classes=list(found['CLASSES']), landmark_ranges=ranges, swap_pairs=swap, groups=groups,
           oks_sigmas_e4=[int(round(s * 1e4)) for s in sigmas])
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'kgdet_amd', 'data',
                   'deepfashion2_landmarks.json')
json.dump(out, open(dst, 'w'), separators=(',', ':'))
print(dst, sum(len(s) for s in swap), 'pairs', len(groups), 'groups')
