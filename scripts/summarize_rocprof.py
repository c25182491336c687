"""Summarise a rocprofv3 run of bench.py into profiles/: per-kernel time of the LAST timed step
(kernel-trace CSV) and, if given, HBM traffic from the FETCH_SIZE / WRITE_SIZE counter passes.

    python scripts/summarize_rocprof.py <kernel_trace.csv> [--fetch <counter_collection.csv>] [--write <...csv>] --batch 50 --out profiles/r01_x.md
"""
import argparse, csv, collections, json, re

ap = argparse.ArgumentParser()
ap.add_argument('trace')
ap.add_argument('--fetch'); ap.add_argument('--write')
ap.add_argument('--batch', type=int, default=50)
ap.add_argument('--forwards', type=int, default=45, help='forwards in the last step')
ap.add_argument('--arch', default='largei3d', choices=['largei3d', 'i3d'], help='network of the run: sets the algorithmic GFLOP per clip (BASELINE.md §2)')
ap.add_argument('--gflop-per-clip', type=float, default=None, help='override (default: by --arch)')
ap.add_argument('--out', required=True)
ap.add_argument('--title', default='bench.py cfg2, last timed step')
ap.add_argument('--streams', type=int, default=1, help='HIP streams the bench used (kernels of different forwards overlap when > 1)')
a = ap.parse_args()
if a.gflop_per_clip is None:
    a.gflop_per_clip = {'largei3d': 32.829145088, 'i3d': 55.575138304}[a.arch]      # conv MACs x 2 per 16 x 224^2 clip

def short(n):
    n = re.sub(r'tedspad::\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(tedspad.*$', '', n).replace('void ', '')
    return n[:90]

rows = [r for r in csv.DictReader(open(a.trace)) if 'tedspad' in r['Kernel_Name'] and 'clock_probe' not in r['Kernel_Name']]      # (bench.py's clock probe runs after the timed region)
FIRST = ('clip_to_channels_last', 'to_channels_last', 'clip_to_tp')      # the first kernel of a forward: the layout pass ...
if not any(any(f in r['Kernel_Name'] for f in FIRST) for r in rows):
    FIRST = ('conv_stem_pt_kernel',)                                    # ... or, since round 4 (the stem reads the fp32 clip itself), the persistent stem
starts = [i for i, r in enumerate(rows) if any(f in r['Kernel_Name'] for f in FIRST)]
first = starts[-a.forwards] if len(starts) >= a.forwards else starts[0]
sel = rows[first:]
nf = len([1 for r in sel if any(f in r['Kernel_Name'] for f in FIRST)])
agg = collections.OrderedDict()
for r in sel:
    k = short(r['Kernel_Name'])
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    c = agg.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += d
tot = sum(v[1] for v in agg.values())
span = (max(int(r['End_Timestamp']) for r in sel) - min(int(r['Start_Timestamp']) for r in sel)) / 1e3   # us, wall clock of the step
conv = sum(v[1] for k, v in agg.items() if k.startswith(('conv_', 'bneck_')))
lines = ['# %s' % a.title, '',
         'Source: `rocprofv3 --kernel-trace --stats -- python3 bench.py ...` on MI355X; %d forwards of %d clips.' % (nf, a.batch), '',
         '| kernel | launches / forward | avg µs | µs / forward | share |', '|---|---|---|---|---|']
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append('| `%s` | %.1f | %.1f | %.1f | %.1f %% |' % (k, c / nf, t / c, t / nf, 100 * t / tot))
lines += ['', '* all kernels: **%.3f ms / forward** (%.1f µs / clip); conv kernels (implicit-GEMM + halo stem): **%.3f ms / forward**' % (tot / nf / 1e3, tot / nf / a.batch, conv / nf / 1e3),
          '* algorithmic %.3f GFLOP/clip x %d clips / conv time = **%.1f TFLOP/s** (%.1f %% of 2500 dense f16/bf16 MFMA); / all-kernel time = %.1f TFLOP/s'
          % (a.gflop_per_clip, a.batch, a.gflop_per_clip * a.batch / (conv / nf) * 1e3, a.gflop_per_clip * a.batch / (conv / nf) * 1e3 / 25.0,
             a.gflop_per_clip * a.batch / (tot / nf) * 1e3)]
lines += ['* wall-clock span of these %d forwards (first kernel start -> last kernel end): **%.2f ms = %.3f ms / forward** -> %.0f clips/s, '
          '%.1f TFLOP/s algorithmic (%.1f %% of 2500) -- the quantity `bench.py` reports as `roofline.achieved` (HIP events around the same region)'
          % (nf, span / 1e3, span / nf / 1e3, a.batch * nf / (span * 1e-6), a.gflop_per_clip * a.batch * nf / span * 1e3,
             a.gflop_per_clip * a.batch * nf / span * 1e3 / 25.0)]
if a.streams > 1:
    lines += ['* the forwards alternate over %d HIP streams, so kernels of different forwards run concurrently: the per-kernel durations above are '
              'measured while sharing the CUs (their sum exceeds the wall-clock span); see the single-stream profile for undisturbed durations.' % a.streams]
summary = {'ms_per_forward_all': tot / nf / 1e3, 'ms_per_forward_conv': conv / nf / 1e3, 'ms_per_forward_wall': span / nf / 1e3}

def counter(path, name):
    """per-kernel sums over the LAST forward of the pass (earlier launches include the tile autotuner)."""
    rows = [r for r in csv.DictReader(open(path)) if r.get('Counter_Name') == name and 'tedspad' in r['Kernel_Name'] and 'clock_probe' not in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    st = [i for i, r in enumerate(rows) if any(f in r['Kernel_Name'] for f in FIRST)]
    rows = rows[st[-1]:]
    per = collections.OrderedDict(); n = collections.OrderedDict()
    for r in rows:
        k = short(r['Kernel_Name'])
        per[k] = per.get(k, 0.0) + float(r['Counter_Value']); n[k] = n.get(k, 0) + 1
    return per, n

if a.fetch and a.write:
    f, fn = counter(a.fetch, 'FETCH_SIZE'); w, wn = counter(a.write, 'WRITE_SIZE')
    lines += ['', '## HBM traffic of one forward (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, last forward of each pass)', '',
              'FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, so the read',
              'side is doubled before comparing with byte counts (MI355X_MICROARCH.md §HBM).', '',
              '| kernel | launches | fetch MB (x2 corrected) | write MB | total MB / forward |', '|---|---|---|---|---|']
    for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0))):
        fm = 2 * f[k] * 1024 / 1e6; wm = w.get(k, 0) * 1024 / 1e6
        lines.append('| `%s` | %d | %.1f | %.1f | %.1f |' % (k, fn[k], fm, wm, fm + wm))
    conv_bytes = sum((2 * f[k] + w.get(k, 0)) * 1024 for k in f if k.startswith(('conv_', 'bneck_')))
    all_bytes = sum((2 * f[k] + w.get(k, 0)) * 1024 for k in f)
    summary['conv_traffic_bytes_per_forward'] = conv_bytes
    summary['all_traffic_bytes_per_forward'] = all_bytes
    lines += ['', '* conv kernels: **%.2f GB / forward** of %d clips (%.1f MB / clip); all kernels %.2f GB / forward. Minimum (each layer reads its input and writes' % (conv_bytes / 1e9, a.batch, conv_bytes / 1e6 / a.batch, all_bytes / 1e9),
              '  its output once, f16): ~129 MB / clip (SURVEY.md §8d).',
              '* over the wall-clock time of a forward (%.3f ms) this is %.2f TB/s of HBM traffic for the conv kernels, %.2f TB/s for all kernels (peak ~8 TB/s).'
              % (span / nf / 1e3, conv_bytes / (span / nf * 1e-6) / 1e12, all_bytes / (span / nf * 1e-6) / 1e12)]
open(a.out, 'w').write('\n'.join(lines) + '\n')
print(json.dumps(summary))
