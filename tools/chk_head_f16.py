import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os
from tests import ref_checks
from tests.golden import ref_cases
for f16 in ('1', '0'):
    import kgdet_amd.conv1x1 as c1
    c1.FORWARD_F16 = f16 == '1'
    head = ref_cases.kgdet_head().cuda()
    try:
        w = ref_checks.check_kgdet_head(head, 'cuda', tol_grad=1e-2)
        print('F16', f16, {k: '%.1e' % v for k, v in w.items() if k.startswith(('grad', 'out:cls_3', 'out:kpt_3'))})
    except AssertionError as e:
        print('F16', f16, 'assert', str(e)[:300])
