#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV: per kernel name count / mean duration, and the
busy fraction (union of all kernel intervals) over the window that holds the lnl_kernel launches.
usage: trace_timeline.py <kernel_trace.csv> [first_n_rows_to_print]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:48], r.get('Queue_Id', '?')))
rows.sort()
lnl = [r for r in rows if 'lnl_kernel' in r[2]]
t0 = lnl[len(lnl) // 4][0]; t1 = lnl[-len(lnl) // 4][1]
win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
by = {}
for a, b, n, q in win:
    by.setdefault(n, []).append(b - a)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f'{n:50s} n={len(v):5d} mean={sum(v)/len(v)/1e3:8.2f} us  sum={sum(v)/1e3:10.1f} us')
cur_a, cur_b, busy = None, None, 0
for a, b, n, q in win:
    if cur_b is None or a > cur_b:
        if cur_b is not None: busy += cur_b - cur_a
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
nl = sum(1 for r in win if 'lnl_kernel' in r[2])
print(f'window {1e-3*(t1-t0):.1f} us, {nl} lnl launches -> {1e-3*(t1-t0)/nl:.2f} us per step; some kernel running {100*busy/(t1-t0):.1f} % of the time')
lb = [(a, b) for a, b, n, q in win if 'lnl_kernel' in n]
ov = 0
for k in range(1, len(lb)):
    ov += max(0, min(lb[k-1][1], lb[k][1]) - lb[k][0])
print(f'mean overlap of consecutive lnl launches: {1e-3*ov/max(1,len(lb)-1):.2f} us')
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
base = win[0][0]
for a, b, nm, q in win[:n]:
    print(f'{1e-3*(a-base):9.2f} {1e-3*(b-base):9.2f} {1e-3*(b-a):8.2f}  q{q} {nm}')
