#!/usr/bin/env python3
"""One process per GPU over the cube driver (the reference's fit_cube(nproc), nestfit/main.py:476-526,
with processes = GPUs): rank r fits the longitude stripe i_lon % world == r into its own chunk file,
rank 0 links the chunks when everybody is done.  No data-path collective: the only synchronisation
is the barrier before linking.

    RANK=r LOCAL_RANK=r WORLD_SIZE=N MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 \\
        python scripts/fit_cube_distributed.py STORE [same_gpu] [crop=LONxLAT]        (one such process per rank;
    `python -m torch.distributed.run --nproc-per-node N ...` sets the same variables)

`same_gpu` makes every rank use GPU 0 (a rehearsal of the N > 1 path on a one-GPU box); `crop=16x4` fits the corner of
that size only.  Prints `comm rccl|tcp` and every rank's device UUID, so that a silent socket fallback or two ranks on
one GPU show."""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    store_name = sys.argv[1] if len(sys.argv) > 1 else '/tmp/nestfit_amd_dist'
    same_gpu = 'same_gpu' in sys.argv[2:]
    crop = [a for a in sys.argv[2:] if a.startswith('crop=')]
    import nestfit_amd as na
    from nestfit_amd.comm import TcpComm, comm_from_env
    from nestfit_amd.cubeio import CubeStack, DataCube, SimpleCube
    from nestfit_amd.fitter import CubeFitter
    from nestfit_amd.store import HdfStore
    na.set_device(0 if same_gpu else local)
    # only a barrier is needed: RCCL between the ranks' GPUs (sockets if RCCL cannot start), plain sockets when they share one
    comm, kind = (TcpComm.from_env(), 'tcp') if same_gpu else comm_from_env()
    import ctypes as C
    import numpy as np
    from nestfit_amd import _ffi
    buf = C.create_string_buffer(40)
    _ffi.check(_ffi.load().nfa_device_uuid(buf, 40))
    mine = np.frombuffer(bytes.fromhex(buf.value.decode()), dtype=np.uint8).astype(np.float64)
    uuids = [bytes(row.astype(np.uint8)).decode('ascii', 'replace') for row in comm.allgather(mine).reshape(world, 16)]
    if rank == 0:
        print(f'comm {kind}; devices {" ".join(uuids)}', flush=True)
    cubes = [DataCube(SimpleCube.read(ROOT / 'tests' / 'golden' / f'ammonia_{t}{t}_cutout.fits')[:-1], 0.35, trans_id=t)
             for t in (1, 2)]
    if crop:
        n_lon, n_lat = (int(v) for v in crop[0][5:].split('x'))
        for dc in cubes:
            dc.data = dc.data[:n_lon, :n_lat, :].copy()
            dc.shape, dc.spatial_shape = dc.data.shape, (n_lon, n_lat)
    stack = CubeStack(cubes)
    ut = na.get_irdc_priors(size=500, vsys=63.7)
    fitter = CubeFitter(stack, ut, na.AmmoniaRunner, lnZ_thresh=11, ncomp_max=2,
                        mn_kwargs={'nlive': 100, 'tol': 1.0, 'efr': 0.3, 'seed': 1}, nlive_snr_fact=5,
                        nlive_quantum=20)        # few, large lock-step groups (the store records the quantum)
    t0 = time.perf_counter()
    fitter.fit_cube(store_name, nproc=world, rank=rank)
    dt = time.perf_counter() - t0
    print(f'rank {rank}/{world}: stripe fitted in {dt:.1f} s', flush=True)
    comm.barrier()
    if rank == 0:
        with HdfStore(store_name) as store:
            store.link_files()
            n = len(list(store.iter_pix_groups()))
        print(f'linked {world} chunk files: {n} pixels in {store_name}.store', flush=True)
    comm.barrier()
    comm.close()


if __name__ == '__main__':
    main()
