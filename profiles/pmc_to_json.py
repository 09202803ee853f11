#!/usr/bin/env python3
"""usage: pmc_to_json.py <kernel substring> <evals per launch> <out.json> <counter_collection.csv> [...]

Per-launch means of the rocprofv3 --pmc counters of ONE kernel (only dispatches whose name holds the
substring AND whose grid is the full batch are averaged: no dilution by other launches), instructions
per evaluation, and the busy fraction of the vector ALUs:
    valu_busy = (SQ_ACTIVE_INST_VALU * 4 / SIMDs) / (GRBM_GUI_ACTIVE / XCDs)
(SQ_ACTIVE_INST_* count quad-cycles summed over the SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs:
MI355X_MICROARCH.md, rocprofv3 PMC slots / cycle constants)."""
import collections
import csv
import json
import sys

kern, evals, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = collections.defaultdict(list)
grid = collections.Counter()
for path in sys.argv[4:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            if kern in row['Kernel_Name']:
                grid[int(row['Grid_Size'])] += 1
full = max(grid) if grid else 0
for path in sys.argv[4:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            if kern in row['Kernel_Name'] and int(row['Grid_Size']) == full:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
mean = {k: sum(v) / len(v) for k, v in acc.items()}
res = {'kernel': kern, 'grid_size': full, 'evals_per_launch': evals, 'dispatches_averaged': {k: len(v) for k, v in acc.items()},
       'per_launch': mean}
per_eval = {k.replace('SQ_INSTS_', '').lower(): mean[k] / evals for k in mean if k.startswith('SQ_INSTS_')}
res['instructions_per_eval'] = per_eval
if 'SQ_ACTIVE_INST_VALU' in mean and 'GRBM_GUI_ACTIVE' in mean:
    simds, xcds = 1024, 8
    busy = mean['SQ_ACTIVE_INST_VALU'] * 4 / simds
    elapsed = mean['GRBM_GUI_ACTIVE'] / xcds
    res['valu_busy_cycles_per_simd'] = busy
    res['elapsed_cycles'] = elapsed
    res['valu_busy_frac'] = busy / elapsed
    if busy > elapsed:
        res['valu_busy_note'] = ('numerator and denominator come from different passes (separate runs of the same command): '
                                 'a ratio a few per cent above 1 says the vector ALUs issue in every cycle of the launch')
    if 'SQ_ACTIVE_INST_SCA' in mean:
        res['scalar_busy_frac'] = mean['SQ_ACTIVE_INST_SCA'] * 4 / simds / elapsed      # per wave slot, like VALU
if 'SQ_WAVE_CYCLES' in mean and 'GRBM_GUI_ACTIVE' in mean:
    # waves resident per SIMD, averaged over the launch (SQ_WAVE_CYCLES: quad-cycles summed over the waves)
    res['avg_waves_per_simd'] = mean['SQ_WAVE_CYCLES'] * 4 / 1024 / (mean['GRBM_GUI_ACTIVE'] / 8)
if 'SQ_LDS_IDX_ACTIVE' in mean and 'GRBM_GUI_ACTIVE' in mean:
    res['lds_idx_active_frac'] = mean['SQ_LDS_IDX_ACTIVE'] / 256 / (mean['GRBM_GUI_ACTIVE'] / 8)     # per CU
if 'SQ_ACTIVE_INST_LDS' in mean and 'GRBM_GUI_ACTIVE' in mean:
    res['lds_inst_busy_frac'] = mean['SQ_ACTIVE_INST_LDS'] * 4 / 1024 / (mean['GRBM_GUI_ACTIVE'] / 8)      # like valu_busy_frac
if 'SQ_LDS_BANK_CONFLICT' in mean and mean.get('SQ_LDS_IDX_ACTIVE'):
    # cycles the LDS spent replaying bank conflicts / cycles it was serving indexed accesses
    res['lds_bank_conflict_frac'] = mean['SQ_LDS_BANK_CONFLICT'] / mean['SQ_LDS_IDX_ACTIVE']
print(json.dumps(res, indent=1))
json.dump(res, open(out, 'w'), indent=1)
