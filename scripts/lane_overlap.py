"""usage: lane_overlap.py <kernel_trace.csv>  -- how many likelihood launches run side by side over the trace (rocprofv3
--kernel-trace of a bench command), and what the set-up launches cost between them."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
dur = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0]
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    short = 'lnl' if 'lnl_kernel' in name else 'setup' if 'setup_kernel' in name else 'sum' if 'lnl_sum' in name else None
    if short:
        dur[short].append(b - a)
        if short == 'lnl':
            ev.append((a, 1)); ev.append((b, -1))
ev.sort()
# the busy half of the trace (the timed blocks): from the first to the last likelihood launch
t_prev, n = ev[0][0], 0
hist = collections.defaultdict(int)
for t, d in ev:
    hist[n] += t - t_prev
    t_prev, n = t, n + d
tot = sum(hist.values())
print('likelihood launches running side by side (share of the time between the first and the last of them):')
print('  ' + '  '.join(f'{k}: {100 * v / tot:.1f} %' for k, v in sorted(hist.items())))
for k, v in dur.items():
    v = sorted(v)
    print(f'{k:6s}: {len(v):6d} launches, median {v[len(v) // 2] / 1e3:8.1f} us, mean {sum(v) / len(v) / 1e3:8.1f} us, p90 {v[int(len(v) * 0.9)] / 1e3:8.1f} us')
