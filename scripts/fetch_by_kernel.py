"""HBM-side read traffic per kernel from a rocprofv3 --pmc FETCH_SIZE pass over scripts/train_prof_run.py (between its marker launches): GB per iteration
(FETCH_SIZE x 2, the gfx950 correction).   python3 scripts/fetch_by_kernel.py COUNTERS.csv [iterations] [name substring ...]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
subs = sys.argv[3:]
disp = collections.OrderedDict()
for r in rows:
    disp.setdefault(int(r['Dispatch_Id']), []).append(r)
ids = sorted(disp)
marks = [i for i in ids if 'clock_probe' in disp[i][0]['Kernel_Name']]
agg = collections.Counter(); n = collections.Counter()
for i in ids:
    if not (marks[-2] < i < marks[-1]):
        continue
    name = disp[i][0]['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('tedspad::', '').split('(')[0][:60]
    if subs and not any(s in name for s in subs):
        continue
    n[name] += 1
    for r in disp[i]:
        if r['Counter_Name'] == 'FETCH_SIZE':
            agg[name] += float(r['Counter_Value']) * 1024.0 * 2
for k, v in agg.most_common(20):
    print('%-62s x%6.1f  %8.2f GB read per iteration' % (k, n[k] / K, v / K / 1e9))
