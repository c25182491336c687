"""Timeline of the cfg3 training iteration from a rocprofv3 --kernel-trace pass over scripts/train_prof_run.py (the dispatches between its two marker launches): wall span,
time with no kernel running (launch gaps), time with kernels of both HIP queues running, busy time per queue, and the kernels that run ALONE the longest (the
candidates for the critical path).    python3 scripts/train_timeline.py TRACE.csv [iterations]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
disp = {}
for r in rows:
    disp[int(r['Dispatch_Id'])] = r
ids = sorted(disp)
marks = [i for i in ids if 'clock_probe' in disp[i]['Kernel_Name']]
win = [disp[i] for i in ids if marks[-2] < i < marks[-1]]


def short(n):
    n = re.sub(r'tedspad::\(anonymous namespace\)::', '', n)
    n = re.sub(r'\((tedspad|float|unsigned|void|int|long|_Float16|__bf16|at::|c10::|char).*$', '', n).replace('void ', '')
    return n[:70]


ev = []
for r in win:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    ev.append((s, 1, r)); ev.append((e, -1, r))
ev.sort(key=lambda t: (t[0], t[1]))
t0, t1 = ev[0][0], ev[-1][0]
live = {}
idle = multi = 0
gaps = []
big = []
prev_end = None
alone = collections.Counter()
last = t0
for t, k, r in ev:
    dt = t - last
    if dt > 0:
        if not live:
            idle += dt; gaps.append(dt)
            if dt > 10000:
                big.append((dt, prev_end, r))
        elif len(live) == 1:
            alone[short(next(iter(live.values()))['Kernel_Name'])] += dt
        else:
            multi += dt
    last = t
    if k == 1:
        live[id(r)] = r
    else:
        live.pop(id(r), None)
        prev_end = r
q = collections.Counter(); qn = collections.Counter()
for r in win:
    q[r['Queue_Id']] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); qn[r['Queue_Id']] += 1
ms = lambda ns: ns / 1e6 / K
print('per iteration (%d iterations, %d dispatches): wall %.2f ms | no kernel running %.2f ms in %d gaps (median %.1f us, > 10 us: %d worth %.2f ms) | >= 2 kernels running %.2f ms'
      % (K, len(win), ms(t1 - t0), ms(idle), len(gaps) // K, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, sum(1 for g in gaps if g > 10000) // K,
         ms(sum(g for g in gaps if g > 10000)), ms(multi)))
for k in q:
    print('  queue %s: %d launches, %.2f ms of kernel time per iteration' % (k, qn[k] // K, ms(q[k])))
print('running alone (ms per iteration):')
for n, v in alone.most_common(16):
    print('  %7.2f  %s' % (ms(v), n))
big.sort(key=lambda t: -t[0])
print('largest gaps with no kernel running (all %d iterations): us | kernel that ended before -> kernel that starts after' % K)
for dt, a, b in big[:40]:
    print('  %7.1f  %s  ->  %s' % (dt / 1e3, short(a['Kernel_Name']) if a else '-', short(b['Kernel_Name'])))
