"""usage: python scripts/kernel_registers.py [filter ...]  -- vector / scalar registers and scratch of every kernel of the engine, read
from the gfx950 assembly hipcc makes of nestfit_amd/csrc/nfa_engine.hip (the flags of nestfit_amd/build.py)."""
import re, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from nestfit_amd import build as nb
flt = sys.argv[1:]
with tempfile.TemporaryDirectory() as tmp:
    out = Path(tmp) / 'k.s'
    flags = [f for f in nb.FLAGS if f not in ('-shared', '-fPIC')]
    subprocess.run(['/opt/rocm/bin/hipcc'] + flags + ['--offload-device-only', '-S', '-o', str(out), str(nb.SRC)], check=True, stderr=subprocess.DEVNULL)
    s = out.read_text()
rows = []
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'\(.*', '', dem).replace('void ', '')
    if flt and not any(f in dem for f in flt):
        continue
    g = lambda k: int(re.search(k + r' (\d+)', body).group(1))
    rows.append((dem, g('next_free_vgpr'), g('next_free_sgpr'), g('private_segment_fixed_size')))
for dem, v, sg, sc in sorted(rows):
    alloc = -(-v // 8) * 8
    print(f'{dem:64s} vgpr {v:4d} (waves/SIMD {min(8, 512 // alloc)})  sgpr {sg:4d}  scratch {sc:5d} B')
