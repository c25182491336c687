"""Per-kernel-family roofline table of the cfg2 forward from the three rocprofv3 summaries (1-stream kernel times, FETCH / WRITE traffic, SQ MFMA busy):
time, HBM bytes, achieved TB/s against the ~5.3 TB/s a streaming kernel reaches here, MFMA busy share, and which of the two is the nearer bound.
    python3 scripts/roofline_by_kernel.py profiles/r04_bench_cfg2_kernels_1stream.md profiles/r04_bench_cfg2_kernels.md profiles/r04_bench_cfg2_mfma_util.md > profiles/r04_bench_cfg2_roofline_by_kernel.md"""
import re, sys, collections

def fam(name):
    name = name.strip('`')
    m = re.match(r'(conv_bneck_tail_kernel)<[^,]+, (\w+), (\w+)>', name)          # <T, DUAL, POOLT> (round 5: the STAGED parameter is gone)
    if m:
        return 'layer1 tail ' + {'true, false': '(first block: conv2 + conv3 + downsample)', 'false, false': '(plain block)', 'false, true': '(last block, maxpool2 inside)'}.get(', '.join(m.groups()[1:]), name)
    for k, v in (('conv_stem_pt_kernel', 'stem + maxpool1 (persistent, reads the fp32 clip)'), ('bneck_frame_kernel', 'layer3 whole bottleneck per frame (x5)'),
                 ('conv_tflat_kernel', 'layer1 conv1 3x1x1 (x3)'), ('conv_bneck_tail128_kernel', 'layer2 tails (x3)'), ('conv_p8_kernel', 'ping-pong 256 x 256 tile (layer2-4 pointwise / temporal / strided)'),
                 ('conv_igemm_kernel', 'generic / split-K tiles (layer4, downsample)'), ('conv_patch', 'patch / flat halo tiles (layer4 3x3)'), ('stem_pool_fix', 'maxpool1 seam fix'), ('avgpool', 'global average pool')):
        if k in name:
            return v
    return name

def rows(path, start_after=None):
    out, on = [], start_after is None
    for line in open(path):
        if start_after and start_after in line:
            on = True
        if on and line.startswith('| `'):
            out.append([c.strip() for c in line.strip().strip('|').split('|')])
    return out

t1, tr, sq = sys.argv[1:4]
T = collections.Counter(); B = collections.Counter(); MB = collections.Counter(); MC = collections.Counter()
for r in rows(t1):
    T[fam(r[0])] += float(r[3])
for r in rows(tr, 'HBM traffic of one forward'):
    B[fam(r[0])] += float(r[4]) * 1e6
for r in rows(sq):
    MB[fam(r[0])] += float(r[2]) * float(r[3].rstrip(' %')) / 100.0
    MC[fam(r[0])] += float(r[2])
tot = sum(T.values())
print('# r04: cfg2 forward (largei3d, 375 clips, f16), per kernel family: time, HBM traffic, MFMA busy, nearer bound\n')
print('Sources: `%s` (one stream: undisturbed durations), `%s` (FETCH_SIZE x 2 + WRITE_SIZE), `%s` (SQ_VALU_MFMA_BUSY_CYCLES / GPU-active cycles x 1024 SIMDs);' % (t1, tr, sq))
print('made by `scripts/roofline_by_kernel.py`. The three passes tune their tiles independently, so the ping-pong / generic families are summed. "HBM share" = achieved TB/s over the')
print('5.3 TB/s the layer1 kernels reach (streaming kernels here: 4.8-5.6); "MFMA share" = busy cycles at the clock the chip holds (nominal-peak share = x 0.72).\n')
print('| kernel family | us / forward | share | GB / forward | TB/s | HBM share | MFMA busy | nearer bound |')
print('|---|---|---|---|---|---|---|---|')
for k, us in sorted(T.items(), key=lambda kv: -kv[1]):
    gb = B.get(k, 0.0) / 1e9
    tbs = gb / us * 1e3 if us else 0
    mf = 100 * MB[k] / MC[k] if MC.get(k) else float('nan')
    hs = 100 * tbs / 5.3
    bound = 'HBM' if hs > mf else 'MFMA / LDS fill'
    print('| %s | %.0f | %.1f %% | %.2f | %.2f | %.0f %% | %s | %s |' % (k, us, 100 * us / tot, gb, tbs, hs, ('%.0f %%' % mf) if mf == mf else '-', bound))
print('\n* all kernels %.2f ms per forward of 375 clips; %.2f GB of HBM traffic (%.1f MB per clip).' % (tot / 1e3, sum(B.values()) / 1e9, sum(B.values()) / 375e6))
