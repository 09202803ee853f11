#!/usr/bin/env python3
"""usage: traffic_summary.py <mode> <FETCH_SIZE counter csv> <WRITE_SIZE counter csv> [out.json]

Per-launch HBM traffic of the likelihood kernel from rocprofv3 PMC passes, corrected as
/opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE / WRITE_SIZE are in KiB;
FETCH_SIZE under-counts reads on gfx950 by a factor that depends on the access width, so it
is calibrated on a stream of known size issued with the engine's own pattern
(null_lnz_kernel over a 256 MiB cube, see traffic_probe.py)."""
import csv, json, sys
mode, fcsv, wcsv = sys.argv[1:4]
out = sys.argv[4] if len(sys.argv) > 4 else None
out_spec = sys.argv[5] if len(sys.argv) > 5 else None

def rows(path, counter):
    r = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] == counter:
                r.append((row['Kernel_Name'], int(row['Grid_Size']), float(row['Counter_Value'])))
    return r

fetch, write = rows(fcsv, 'FETCH_SIZE'), rows(wcsv, 'WRITE_SIZE')
cal = [v for k, g, v in fetch if 'null_lnz_kernel' in k and g >= 16384 * 2 * 64]
cal_bytes = 16384 * 2048 * 8
factor = cal_bytes / (max(cal) * 1024.0)
import re
def spectra_out(k):
    # lnl_kernel<MODE, WRITE_SPEC, ...>, lnl_kernel_w8<MODE, WRITE_SPEC, ...>, lnl_kernel_queue<WRITE_SPEC, NCOMP>
    m = re.search(r'lnl_kernel_queue<(true|false)', k) or re.search(r'lnl_kernel(?:_w8)?<\d+, (true|false)', k)
    return bool(m) and m.group(1) == 'true'
lnl_f = [v for k, g, v in fetch if 'lnl_kernel' in k and g >= 4096 * 64 and not spectra_out(k)]
lnl_w = [v for k, g, v in write if 'lnl_kernel' in k and g >= 4096 * 64 and not spectra_out(k)]
spec_f = [v for k, g, v in fetch if 'lnl_kernel' in k and g >= 4096 * 64 and spectra_out(k)]
spec_w = [v for k, g, v in write if 'lnl_kernel' in k and g >= 4096 * 64 and spectra_out(k)]
# lnl_sum_kernel (adds the per-spectrum terms) belongs to the same launch: 4096 threads
sum_f = [v for k, g, v in fetch if 'lnl_sum_kernel' in k and g == 4096] or [0.0]
sum_w = [v for k, g, v in write if 'lnl_sum_kernel' in k and g == 4096] or [0.0]
sum_bytes = sum(sum_f) / len(sum_f) * 1024 * factor + sum(sum_w) / len(sum_w) * 1024
half = len(lnl_f) // 2
res = {
    'fetch_calibration_factor': factor,
    'one_pixel': {'fetch_bytes': sum(lnl_f[:half]) / half * 1024 * factor, 'write_bytes': sum(lnl_w[:half]) / half * 1024},
    'pixel_per_item': {'fetch_bytes': sum(lnl_f[half:]) / (len(lnl_f) - half) * 1024 * factor,
                       'write_bytes': sum(lnl_w[half:]) / (len(lnl_w) - half) * 1024},
}
for k in ('one_pixel', 'pixel_per_item'):
    res[k]['lnl_sum_bytes'] = sum_bytes
    res[k]['total_bytes'] = res[k]['fetch_bytes'] + res[k]['write_bytes'] + sum_bytes
if spec_f and spec_w:
    # the spectra-out launch of the probe (4096 rows, one pixel): the last launches are the timed-shape ones
    res['spectra_out'] = {'fetch_bytes': sum(spec_f[-5:]) / len(spec_f[-5:]) * 1024 * factor, 'write_bytes': sum(spec_w[-5:]) / len(spec_w[-5:]) * 1024}
    res['spectra_out']['total_bytes'] = res['spectra_out']['fetch_bytes'] + res['spectra_out']['write_bytes']
print(json.dumps(res, indent=1))
if out_spec and 'spectra_out' in res and mode == 'fast':
    json.dump({'mode': mode, 'rows_per_launch': 4096, 'bytes_per_launch': res['spectra_out']['total_bytes'], **res['spectra_out'],
               'algorithmic_bytes_per_launch': 4096 * (16584 + 16384), 'fetch_calibration_factor': factor,
               'source': 'profiles/traffic_probe.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), lnl_kernel<2, true, ...>'},
              open(out_spec, 'w'), indent=1)
if out:
    try:
        allres = json.load(open(out))
    except Exception:
        allres = {}
    allres.setdefault('C2', {})[mode] = res['one_pixel']['total_bytes']
    allres.setdefault('detail', {})[mode] = res
    json.dump(allres, open(out, 'w'), indent=1)
