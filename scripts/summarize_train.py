"""Per-kernel table of the cfg3 training iteration from rocprofv3 passes over scripts/train_prof_run.py (the dispatches between its two marker launches of
clock_probe_kernel): launches, time (kernel trace), HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md) and MFMA utilisation
(SQ counters) per iteration.
    python3 scripts/summarize_train.py --trace T.csv --fetch F.csv --write W.csv --sq S.csv --iters 3 --out profiles/r03_train_cfg3_kernels.md"""
import argparse, collections, csv, re

ap = argparse.ArgumentParser()
ap.add_argument('--trace'); ap.add_argument('--fetch'); ap.add_argument('--write'); ap.add_argument('--sq'); ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--out', required=True); ap.add_argument('--title', default='cfg3 training iteration')
a = ap.parse_args()


def short(n):
    n = re.sub(r'tedspad::\(anonymous namespace\)::', '', n)
    n = re.sub(r'\((tedspad|float|unsigned|void|int|long|_Float16|__bf16|at::|c10::|char).*$', '', n).replace('void ', '')
    return n[:86]


def window(path):
    """rows (dicts) of the dispatches strictly between the two marker launches, grouped by dispatch"""
    rows = list(csv.DictReader(open(path)))
    key = 'Dispatch_Id'
    disp = collections.OrderedDict()
    for r in rows:
        disp.setdefault(int(r[key]), []).append(r)
    ids = sorted(disp)
    marks = [i for i in ids if 'clock_probe' in disp[i][0]['Kernel_Name']]
    assert len(marks) >= 2, 'markers not found in %s' % path
    return [disp[i] for i in ids if marks[-2] < i < marks[-1]]


agg = collections.OrderedDict()
def G(name):
    return agg.setdefault(name, collections.Counter())

if a.trace:
    for d in window(a.trace):
        r = d[0]
        g = G(short(r['Kernel_Name'])); g['n'] += 1; g['ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for path, cname, field in ((a.fetch, 'FETCH_SIZE', 'fetch'), (a.write, 'WRITE_SIZE', 'write')):
    if path:
        for d in window(path):
            for r in d:
                if r['Counter_Name'] == cname:
                    G(short(r['Kernel_Name']))[field] += float(r['Counter_Value']) * 1024.0      # KiB
if a.sq:
    for d in window(a.sq):
        g = G(short(d[0]['Kernel_Name']))
        for r in d:
            g[r['Counter_Name']] += float(r['Counter_Value'])

K = a.iters
tot = collections.Counter()
lines = ['# %s' % a.title, '',
         'Source: `rocprofv3 {--kernel-trace | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...} -- python3 scripts/train_prof_run.py %d` (or the script named in the title)' % K,
         'on MI355X; the dispatches between the script\'s two marker launches, divided by its %d iterations (phase 1 + phase 2 each). Kernels are serialised under' % K,
         'counter collection. Traffic = FETCH_SIZE x 2 + WRITE_SIZE (gfx950 correction); MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).', '',
         '| kernel | launches / iteration | ms / iteration | share | HBM read MB | HBM written MB | TB/s | MFMA utilisation |', '|---|---|---|---|---|---|---|---|']
tns = sum(g['ns'] for g in agg.values()) or 1
for k, g in sorted(agg.items(), key=lambda kv: -kv[1]['ns']):
    rd, wr = 2 * g['fetch'] / K, g['write'] / K
    ms = g['ns'] / K / 1e6
    util = 100 * g['SQ_VALU_MFMA_BUSY_CYCLES'] / (g['GRBM_GUI_ACTIVE'] / 8 * 1024) if g['GRBM_GUI_ACTIVE'] else float('nan')
    for f in ('ns', 'fetch', 'write', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'n'):
        tot[f] += g[f]
    if ms < 0.02 and rd + wr < 5e6:
        tot['small_n'] += g['n']; tot['small_ns'] += g['ns']
        continue
    lines.append('| `%s` | %.1f | %.3f | %.1f %% | %.0f | %.0f | %s | %s |' % (k, g['n'] / K, ms, 100 * g['ns'] / tns, rd / 1e6, wr / 1e6,
                 '%.2f' % ((rd + wr) / (ms * 1e9)) if ms > 0 else '-', '%.1f %%' % util if util == util else '-'))
lines.append('| (kernels under 0.02 ms and 5 MB per iteration) | %.1f | %.3f | %.1f %% | | | | |' % (tot['small_n'] / K, tot['small_ns'] / K / 1e6, 100 * tot['small_ns'] / tns))
ms = tot['ns'] / K / 1e6
rd, wr = 2 * tot['fetch'] / K, tot['write'] / K
lines += ['', '* all kernels: **%.0f launches, %.2f ms of kernel time per iteration**; HBM traffic **%.2f GB read + %.2f GB written = %.2f GB per iteration** (%.2f TB/s over the kernel time)' % (
              tot['n'] / K, ms, rd / 1e9, wr / 1e9, (rd + wr) / 1e9, (rd + wr) / (ms * 1e9) if ms else 0),
          '* MFMA utilisation over all kernels: **%.1f %%** of the GPU-active cycles' % (100 * tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (tot['GRBM_GUI_ACTIVE'] / 8 * 1024) if tot['GRBM_GUI_ACTIVE'] else float('nan'))]
open(a.out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[-3:]))
import json
print(json.dumps({'ms_per_iteration_kernels': ms, 'traffic_bytes_per_iteration': rd + wr, 'launches_per_iteration': tot['n'] / K}))
