"""Per-kernel MFMA utilisation and wave-state fractions of the LAST forward of a `rocprofv3 --pmc SQ_... GRBM_GUI_ACTIVE`
pass over bench.py (kernels are serialised under counter collection, so GRBM_GUI_ACTIVE is the kernel's own busy time).

  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs)   (cycles; rocprofv3 reports
  GRBM_GUI_ACTIVE summed over the 8 XCDs: a 5.8 ms forward shows ~100 Mcycles = 8 x 12.5 M)
  wait / stall / issue = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (quad-cycles, disjoint)

    python scripts/summarize_sq.py <counter_collection.csv> --out profiles/r01_bench_cfg2_mfma_util.md
"""
import argparse, collections, csv, re

ap = argparse.ArgumentParser()
ap.add_argument('csv'); ap.add_argument('--out', required=True); ap.add_argument('--batch', type=int, default=75)
a = ap.parse_args()

def short(n):
    n = re.sub(r'tedspad::\(anonymous namespace\)::', '', n)
    return re.sub(r'\(tedspad.*$', '', n).replace('void ', '')[:80]

rows = [r for r in csv.DictReader(open(a.csv)) if 'tedspad' in r['Kernel_Name'] and 'clock_probe' not in r['Kernel_Name']]
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r['Dispatch_Id']), {'name': short(r['Kernel_Name']), 'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
    d[r['Counter_Name']] = float(r['Counter_Value'])
ids = sorted(disp)
starts = [i for i in ids if 'to_channels_last' in disp[i]['name'] or 'clip_to_tp' in disp[i]['name']]
if not starts:       # round 4: the stem reads the fp32 clip itself -- it is the first kernel of a forward
    starts = [i for i in ids if 'conv_stem_pt_kernel' in disp[i]['name']]
sel = [disp[i] for i in ids if i >= starts[-1]]
agg = collections.OrderedDict()
for d in sel:
    g = agg.setdefault(d['name'], collections.Counter())
    g['n'] += 1
    for k, v in d.items():
        if k != 'name':
            g[k] += v
lines = ['# MFMA utilisation and wave states per kernel, last forward of %d clips (bench.py cfg2 under rocprofv3 --pmc)' % a.batch, '',
         'Counters: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY (one pass;',
         'kernels are serialised under counter collection). MFMA utilisation = MFMA busy cycles / (GPU-active cycles x 1024 SIMDs),',
         'GPU-active cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs).', '',
         '| kernel | launches | GPU-active Mcycles | MFMA utilisation | waves waiting (s_waitcnt/barrier) | issue-stalled | issuing |', '|---|---|---|---|---|---|---|']
tot = collections.Counter()
for k, g in sorted(agg.items(), key=lambda kv: -kv[1]['GRBM_GUI_ACTIVE']):
    wc = max(g['SQ_WAVE_CYCLES'], 1.0)
    lines.append('| `%s` | %d | %.2f | %.1f %% | %.0f %% | %.0f %% | %.0f %% |' % (
        k, g['n'], g['GRBM_GUI_ACTIVE'] / 8e6, 100 * g['SQ_VALU_MFMA_BUSY_CYCLES'] / (g['GRBM_GUI_ACTIVE'] / 8 * 1024),
        100 * g['SQ_WAIT_ANY'] / wc, 100 * g['SQ_WAIT_INST_ANY'] / wc, 100 * g['SQ_ACTIVE_INST_ANY'] / wc))
    if k.startswith(('conv_', 'bneck_')):
        tot.update(g)
lines += ['', '* conv kernels together: MFMA utilisation **%.1f %%** of the GPU-active cycles (%.1f Mcycles per forward)'
          % (100 * tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (tot['GRBM_GUI_ACTIVE'] / 8 * 1024), tot['GRBM_GUI_ACTIVE'] / 8e6),
          '* MFMA busy cycles per forward / (32 cycles per 32x32x16 MFMA x 32768 FLOP) = %.1f GFLOP executed per clip on the matrix cores (algorithmic: 32.83; the rest is K / Cout / tile padding)'
          % (tot['SQ_VALU_MFMA_BUSY_CYCLES'] / 32 * 32768 / 1e9 / a.batch)]
open(a.out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
