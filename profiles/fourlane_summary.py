#!/usr/bin/env python3
"""usage: fourlane_summary.py <kernel_trace.csv> <bench line json>

The headline's own command (four stream lanes, one mode) under `rocprofv3 --kernel-trace`: what the kernels of
the timed blocks add up to against the wall time the bench line reports -- the isolated launch (one lane) against the
pipelined rate, shown instead of argued."""
import csv, json, sys, collections
trace, line = sys.argv[1], json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(trace)))
dur = collections.defaultdict(list)
t_min, t_max = None, None
for r in rows:
    name = r['Kernel_Name'].split('(')[0]
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'lnl_kernel' in name or 'setup_kernel' in name or 'lnl_sum_kernel' in name:
        dur[name].append((a, b, int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size'])))
m = line['modes'][line['config']['exp_mode']]
spl = line['config']['steps_per_launch']
B = 4096
# the launches of `spl` coalesced steps: one wave per unit (grid = units x 64), or the table mode's queue form (resident workgroups)
lnl = [d for n, v in dur.items() if 'lnl_kernel' in n and 'lnl_sum' not in n for d in v if d[2] >= spl * B * 2 * 64 or 'lnl_kernel_queue' in n]
tot = sum(b - a for a, b, g in lnl)
ivs = sorted((a, b) for a, b, g in lnl)
union, ca, cb = 0, None, None
for a, b in ivs:
    if cb is None or a > cb:
        if cb is not None:
            union += cb - ca
        ca, cb = a, b
    else:
        cb = max(cb, b)
if cb is not None:
    union += cb - ca
n = len(lnl)
print(f'launches of {spl} coalesced steps (lnl_kernel, {spl * B} evaluations each): {n}')
print(f'sum of their durations: {tot} ns = {tot / max(n, 1) / 1e3:.1f} us per launch = {tot / max(n * spl, 1) / 1e3:.2f} us per step (launches overlap on the stream lanes: each is stretched by its neighbours)')
print(f'time during which at least one of them ran (union of the intervals): {union} ns = {union / max(n * spl, 1) / 1e3:.2f} us per step')
print(f'bench line of the same run (under the tracer): {m["ms_per_step"] * 1e3:.2f} us per step over blocks of {m["repeats_per_block"]} x {line["steps"]} steps, {line["value"] / 1e6:.1f} M evaluations/s')
print(f'one-lane launch of the same shape (roofline.avg_launch_us of that line, HIP events): {line["roofline"].get("avg_launch_us")} us = {line["roofline"].get("avg_launch_us", 0) / spl:.2f} us per step')
