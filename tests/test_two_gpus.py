"""The N > 1 path on real devices: one process per GPU, RCCL between them through the engine's C ABI,
the cube driver fitting one longitude stripe per rank (nestfit/main.py:516-523, 565-571).  Skipped where fewer than
two GPUs are visible (the builder's boxes have one); on a one-GPU box the same driver is rehearsed with both ranks on
the one device and the socket communicator."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _n_devices():
    import nestfit_amd as na
    return na.device_count()


def _ranks(cmd, world, timeout=300):
    """`cmd` as `world` processes with a launcher's environment; returns their outputs (rank order)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable] + cmd, env=env, cwd=str(ROOT), stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out)
            assert p.returncode == 0, out[-3000:]
    finally:
        for p in procs:                       # a rank that failed leaves the others at the barrier: end them
            if p.poll() is None:
                p.kill()
                p.wait()
    return outs


def test_rccl_between_two_devices():
    if _n_devices() < 2:
        pytest.skip('needs two visible GPUs')
    outs = _ranks([str(ROOT / 'tests' / 'two_gpu_worker.py')], 2)
    line = [ln for ln in outs[0].splitlines() if ln.startswith('TWO_GPU')]
    assert line and 'kind=rccl' in line[0] and line[0].endswith('ok=True'), outs[0][-2000:]


def _check_store(store_dir, n_pix):
    from nestfit_amd.store import HdfStore
    with HdfStore(str(store_dir)) as store:
        groups = list(store.iter_pix_groups())
        assert len(groups) == n_pix and store.nchunks == 2
        for g in groups:
            assert 0 <= g.attrs['nbest'] <= 2 and '1' in g
        lon = sorted({g.attrs['i_lon'] for g in groups})
        assert lon == list(range(16))


def test_cube_driver_on_two_devices(tmp_path):
    if _n_devices() < 2:
        pytest.skip('needs two visible GPUs')
    outs = _ranks([str(ROOT / 'scripts' / 'fit_cube_distributed.py'), str(tmp_path / 'two'), 'crop=16x4'], 2)
    assert 'comm rccl' in outs[0] and 'linked 2 chunk files: 64 pixels' in outs[0], outs[0][-2000:]
    devices = [ln for ln in outs[0].splitlines() if ln.startswith('comm ')][0].split('devices ')[1].split()
    assert len(set(devices)) == 2
    _check_store(tmp_path / 'two', 64)


def test_cube_driver_with_two_ranks_on_one_device(tmp_path):
    """The same two-rank driver on whatever one GPU is there (ranks share it: sockets instead of RCCL)."""
    outs = _ranks([str(ROOT / 'scripts' / 'fit_cube_distributed.py'), str(tmp_path / 'one'), 'same_gpu', 'crop=16x4'], 2)
    assert 'comm tcp' in outs[0] and 'linked 2 chunk files: 64 pixels' in outs[0], outs[0][-2000:]
    _check_store(tmp_path / 'one', 64)
