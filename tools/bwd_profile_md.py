"""Markdown section of one regime of profiles/r06_dcn_bwd_plane_kernels.md from the files tools/profile_bwd_group.sh leaves in
gpurun_out/<run>/ (kernel stats + FETCH_SIZE / WRITE_SIZE passes):  python tools/bwd_profile_md.py gpurun_out/<run> <regime>"""
import collections, csv, sys

out, regime = sys.argv[1], sys.argv[2]
stats = {}
for r in csv.DictReader(open(out + '/bwd_kernel_stats.csv')):
    n = r['Name']
    if 'dcn_' not in n:
        continue
    short = n.split('(')[0].replace('void ', '').replace('kgdet::', '')
    stats[short] = (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3)


def pmc(f, counter):
    acc = collections.defaultdict(list)
    try:
        for r in csv.DictReader(open(out + '/' + f + '.csv')):
            if r['Counter_Name'] == counter and 'dcn_' in r['Kernel_Name']:
                acc[r['Kernel_Name'].split('(')[0].replace('void ', '').replace('kgdet::', '')].append(float(r['Counter_Value']))
    except FileNotFoundError:
        pass
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = pmc('pmcf', 'FETCH_SIZE'), pmc('pmcw', 'WRITE_SIZE')
print('## %s offsets\n' % regime)
print('| kernel | calls | avg us | min | max | FETCH KiB | WRITE KiB |\n|---|---|---|---|---|---|---|')
for k, (c, a, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
    print('| `%s` | %d | %.1f | %.1f | %.1f | %.0f | %.0f |' % (k, c, a, mn, mx, fetch.get(k, 0), write.get(k, 0)))
products = {
    'grad_weight': ['dcn_bwd_weight_os<2>', 'dcn_pack_grad_out', 'dcn_build_taps'],
    'grad_input': ['dcn_bwd_input_plane<2>', 'dcn_bwd_input_prepare', 'dcn_hot_gemm', 'dcn_inv_medium_sums', 'dcn_inv_overflow_sums', 'dcn_fwd_fixup_static'],
    'grad_offset': ['dcn_bwd_offset_pair', 'dcn_build_grad_taps', 'dcn_bwd_offset_plane_fixup'],
}
print('\n| product | kernels (avg us) | us | fraction of 833 TFLOP/s | fabric traffic (2 x FETCH + WRITE) | vs 72.1 MB |\n|---|---|---|---|---|---|')
flops = 45.69e9
for p, ks in products.items():
    ks = [k for k in ks if k in stats]
    us = sum(stats[k][1] for k in ks)
    traffic = sum((2 * fetch.get(k, 0) + write.get(k, 0)) * 1024 for k in ks)
    main = stats[ks[0]][1]
    print('| %s | %s | %.1f | %.3f (kernel alone %.3f) | %.1f MB | %.2fx |' % (
        p, ' + '.join('`%s` %.1f' % (k, stats[k][1]) for k in ks), us, flops / (us * 1e-6) / 833e12, flops / (main * 1e-6) / 833e12,
        traffic / 1e6, traffic / 72.1e6))
    print('traffic_bytes_%s%s: %d' % ('' if regime == 'random' else regime + '_', p, traffic), file=sys.stderr)
