"""Kernel time per family of a steady-state training-step table (tools/trace_steady_stats.py output, all rows): prints the
`family_us <name>: <us per step>` lines that profiles/r06_train_step_fp32_steady.md carries and bench.py copies into
`roofline_step.kernel_us_by_family`.   python tools/step_families.py gpurun_out/<run>/train_steady.md"""
import re, sys

FAMILIES = [
    ('deformable (forward, three gradients, builders, fix-ups, packs)', r'kgdet::dcn_|kgdet::\(anonymous namespace\)::large_'),
    ('dense grad_weight', r'conv_nt8|conv_ntp|conv_s2_gather9|igemm_wrw|batched_transpose'),
    ('partial sums / BatchNorm fold / packs of the dense kernels', r'conv1x1_sum|conv_wsum_fold|conv3x3_wsum|conv1x1_pack|bn_partial_sum|pad_rows2'),
    ('dense forward / grad_input (split MFMA)', r'conv_nn<|conv3x3_patch|conv3x3_s2_grad_input|stem_conv7x7'),
    ('library GEMMs (vendor)', r'Cijk_|igemm_(?!wrw)|naive_conv|gridwise'),
    ('optimizer (clip + Adam)', r'multi_clip_adam|multi_sqnorm|multi_clip_sgd'),
    ('GroupNorm / losses / glue (HIP)', r'kgdet::|bn_relu_maxpool'),
    ('ATen element-wise / reduce / copy', r'.'),
]
rows = []
total_line = ''
for line in open(sys.argv[1]):
    if line.startswith('steady-state'):
        total_line = line.strip()
    m = re.match(r'\| `(.*)` \| ([\d.]+) \| ([\d.]+) \|', line)
    if m:
        rows.append((m.group(1), float(m.group(2)), float(m.group(3))))
fam = {n: [0.0, 0.0] for n, _ in FAMILIES}
for name, launches, us in rows:
    for n, pat in FAMILIES:
        if re.search(pat, name):
            fam[n][0] += us
            fam[n][1] += launches
            break
print(total_line)
for n, _ in FAMILIES:
    print('family_us %s: %.1f   (%d launches)' % (n, fam[n][0], fam[n][1]))
print('sum of the table rows: %.1f us, %d launches' % (sum(r[2] for r in rows), sum(r[1] for r in rows)))
