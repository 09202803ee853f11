#!/usr/bin/env python3
"""One process per GPU over the cube driver (the reference's fit_cube(nproc), nestfit/main.py:476-526,
with processes = GPUs): rank r fits the longitude stripe i_lon % world == r into its own chunk file,
rank 0 links the chunks when everybody is done.  No data-path collective: the only synchronisation
is the barrier before linking.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port 29511 scripts/fit_cube_distributed.py STORE [same_gpu]

`same_gpu` makes every rank use GPU 0 (a rehearsal of the N > 1 path on a one-GPU box)."""
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
    same_gpu = len(sys.argv) > 2 and sys.argv[2] == 'same_gpu'
    import nestfit_amd as na
    from nestfit_amd.comm import TcpComm, comm_from_env
    from nestfit_amd.cubeio import CubeStack, DataCube, SimpleCube
    from nestfit_amd.fitter import CubeFitter
    from nestfit_amd.store import HdfStore
    na.set_device(0 if same_gpu else local)
    # only a barrier is needed: RCCL between the ranks' GPUs (sockets if RCCL cannot start), plain sockets when they share one
    comm = TcpComm.from_env() if same_gpu else comm_from_env()[0]
    stack = CubeStack([
        DataCube(SimpleCube.read(ROOT / 'tests' / 'golden' / f'ammonia_{t}{t}_cutout.fits')[:-1], 0.35, trans_id=t)
        for t in (1, 2)])
    ut = na.get_irdc_priors(size=500, vsys=63.7)
    fitter = CubeFitter(stack, ut, na.AmmoniaRunner, lnZ_thresh=11, ncomp_max=2,
                        mn_kwargs={'nlive': 100, 'tol': 1.0, 'efr': 0.3, 'seed': 1}, nlive_snr_fact=5)
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
