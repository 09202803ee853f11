"""One rank of tests/test_two_gpus.py: RCCL between two devices through the engine's C ABI (nfa_comm_*)."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    import nestfit_amd as na
    from nestfit_amd import _ffi, comm as nfcomm
    from nestfit_amd.cube import shard_pixels
    na.set_device(int(os.environ['LOCAL_RANK']))
    comm, kind = nfcomm.comm_from_env(rccl_timeout=120.0)
    buf = C.create_string_buffer(40)
    _ffi.check(_ffi.load().nfa_device_uuid(buf, 40))
    mine = np.frombuffer(bytes.fromhex(buf.value.decode()), dtype=np.uint8).astype(np.float64)
    uuids = [bytes(r.astype(np.uint8)).hex() for r in comm.allgather(mine).reshape(world, 16)]
    lon, lat = shard_pixels((7, 5), rank, world)                # uneven stripes
    rec = np.stack([lon, lat, np.full(lon.size, rank), 1000.0 * lon + lat], axis=1).astype(float)
    allrec = nfcomm.gather_pixel_records(rec, comm)
    top = comm.allreduce(np.array([float(rank)]), 'max')[0]
    tot = comm.allreduce(np.array([1.0 + rank]), 'sum')[0]
    comm.barrier()
    if rank == 0:
        ok = (kind == 'rccl' and len(set(uuids)) == world and allrec.shape == (35, 4)
              and (allrec[:, 0] % world == allrec[:, 2]).all() and top == world - 1 and tot == world * (world + 1) / 2)
        print(f'TWO_GPU kind={kind} devices={",".join(uuids)} records={allrec.shape[0]} ok={ok}', flush=True)
    comm.close()


if __name__ == '__main__':
    main()
