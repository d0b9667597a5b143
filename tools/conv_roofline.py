"""Aggregate roofline of the backbone / tower convolution kernels in one training step: enumerates the convolutions that
run on csrc/conv1x1.hip (by shape rule of kgdet_amd/conv1x1.applicable), sums their algorithmic flops and bytes, and divides
by the per-step kernel times of a committed step profile:  python tools/conv_roofline.py <train_steady.md>"""
import re
import sys

B, H0, W0 = 2, 800, 1344


def resnet50_convs():
    convs = []   # (cin, cout, k, stride, Hin, Win, trainable)
    h, w = H0 // 2, W0 // 2
    convs.append((3, 64, 7, 2, H0, W0, False))
    h, w = h // 2, w // 2          # maxpool
    inpl = 64
    for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
        for b in range(blocks):
            s = stride if b == 0 else 1
            train = li >= 1          # frozen_stages = 1
            convs.append((inpl, planes, 1, 1, h, w, train))
            convs.append((planes, planes, 3, s, h, w, train))
            ho, wo = (h + s - 1) // s, (w + s - 1) // s
            convs.append((planes, planes * 4, 1, 1, ho, wo, train))
            if b == 0:
                convs.append((inpl, planes * 4, 1, s, h, w, train))
            inpl = planes * 4
            h, w = ho, wo
    return convs


def main():
    times = {}
    for line in open(sys.argv[1]):
        m = re.match(r"\| `(?:void )?kgdet::(conv_n[nt]8?<\d>|conv3x3_patch4?)", line)
        if m:
            cols = [c.strip() for c in line.split('|')]
            name = 'conv_nn<9>' if m.group(1).startswith('conv3x3_patch') else m.group(1)   # the stride-1 3x3 kernels
            times[name] = times.get(name, 0.0) + float(cols[-3])
    convs = resnet50_convs()
    # FPN2 lateral 1x1 (2048 -> 256 at 25x42 etc.) and the head towers (6 x 3x3 256 -> 256 at 25x42) are small next to these
    agg = {}
    for cin, cout, k, s, h, w, train in convs:
        if k == 7:
            continue
        ho, wo = (h + s - 1) // s, (w + s - 1) // s
        flops = 2.0 * B * cin * cout * k * k * ho * wo
        byts = 4.0 * B * (cin * h * w + cout * ho * wo) + 4.0 * cin * cout * k * k
        if s == 1 or k == 1:
            key_f = 'conv_nn<%d>' % (k * k)
            a = agg.setdefault(key_f, [0.0, 0.0, 0]); a[0] += flops; a[1] += byts; a[2] += 1          # forward
            if train or True:                       # grad_input flows through frozen layers too (not into the stem)
                a[0] += flops; a[1] += byts; a[2] += 1
            if train and s == 1:
                key_w = 'conv_nt8<%d>' % (k * k)
                a = agg.setdefault(key_w, [0.0, 0.0, 0]); a[0] += flops; a[1] += byts; a[2] += 1
    print('| kernel family | convolutions/step (approx.) | GFLOP | algorithmic MB | us/step (profile) | TFLOP/s | of 833 TF (bf16/3) | GB/s | of 8 TB/s |')
    print('|---|---|---|---|---|---|---|---|---|')
    for k, (f, b, n) in sorted(agg.items()):
        t = times.get(k)
        if not t:
            continue
        print('| `%s` | %d | %.1f | %.0f | %.0f | %.1f | %.2f | %.0f | %.2f |' % (
            k, n, f / 1e9, b / 1e6, t, f / t / 1e6, f / t / 1e6 / 833.3, b / t / 1e3, b / t / 1e3 / 8000))


if __name__ == '__main__':
    main()
