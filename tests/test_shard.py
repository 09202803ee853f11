"""N > 1 path on CPU: pixel striping identical to the reference's
get_multiproc_indices (nestfit/main.py:565-571) and the end-of-run gather of
per-pixel records (nestfit_amd.comm.gather_pixel_records) over a gloo-backed stand-in
for the RCCL communicator (world_size 2 and 3)."""
import os
import socket

import numpy as np
import pytest

from nestfit_amd.cube import get_multiproc_indices, shard_pixels


def test_striping_matches_reference_rule():
    shape = (13, 7)
    for world in (1, 2, 3, 8):
        seen = np.zeros(shape, dtype=int)
        for rank in range(world):
            lon, lat = shard_pixels(shape, rank, world)
            assert (lon % world == rank).all()                 # i_lon mod nproc == rank
            assert lon.size == lat.size
            seen[lon, lat] += 1
        assert (seen == 1).all()                               # every pixel exactly once
    lon, lat = get_multiproc_indices((4, 3), 2)[1]
    assert lon.tolist() == [1, 1, 1, 3, 3, 3] and lat.tolist() == [0, 1, 2, 0, 1, 2]
    with pytest.raises(ValueError):
        shard_pixels((4, 4), 4, 4)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, out_dir):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from comm_gloo import GlooComm
    from nestfit_amd.comm import gather_pixel_records
    from nestfit_amd.cube import shard_pixels
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    comm = GlooComm()
    lon, lat = shard_pixels(shape, rank, world)
    # a record per pixel: (lon, lat, rank, stand-in for lnZ) -- fixed width, variable count
    rec = np.stack([lon, lat, np.full(lon.size, rank), 1000.0 * lon + lat], axis=1).astype(float)
    allrec = gather_pixel_records(rec, comm)
    assert comm.allreduce(np.array([float(rank)]), 'max')[0] == world - 1
    comm.barrier()
    np.save(os.path.join(out_dir, f'r{rank}.npy'), allrec)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_gather_of_pixel_records_gloo(tmp_path, world):
    import torch.multiprocessing as mp
    shape = (7, 5)                                             # uneven stripes on purpose
    port = _free_port()
    mp.spawn(_worker, args=(world, port, shape, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f'r{r}.npy') for r in range(world)]
    for g in got[1:]:
        assert np.array_equal(g, got[0])                       # every rank sees the same table
    g = got[0]
    assert g.shape == (35, 4)
    assert len({(int(a), int(b)) for a, b in g[:, :2]}) == 35  # all pixels, once
    assert (g[:, 0] % world == g[:, 2]).all()                  # owner = i_lon mod world
    assert np.array_equal(g[:, 3], 1000.0 * g[:, 0] + g[:, 1])


def test_solo_comm_and_id_exchange():
    """world = 1 is the identity; the unique-id hand-off (rank 0 serves, the others fetch) over loopback."""
    import threading
    from nestfit_amd.comm import SoloComm, _exchange_id, gather_pixel_records
    rec = np.arange(12.0).reshape(4, 3)
    assert np.array_equal(gather_pixel_records(rec, SoloComm()), rec)
    port = _free_port()
    got = {}

    def client(rank):
        got[rank] = _exchange_id(rank, 3, None, '127.0.0.1', port, timeout=30)

    threads = [threading.Thread(target=client, args=(r,)) for r in (1, 2)]
    for t in threads:
        t.start()
    uid = _exchange_id(0, 3, lambda: bytes(range(128)), '127.0.0.1', port, timeout=30)
    for t in threads:
        t.join()
    assert uid == bytes(range(128)) and got[1] == uid and got[2] == uid


def _tcp_worker(rank, world, port, q):
    from nestfit_amd.comm import TcpComm, gather_pixel_records
    from nestfit_amd.cube import shard_pixels
    comm = TcpComm(rank, world, '127.0.0.1', port, timeout=60)
    lon, lat = shard_pixels((7, 5), rank, world)
    rec = np.stack([lon, lat, np.full(lon.size, rank)], axis=1).astype(float)
    allrec = gather_pixel_records(rec, comm)
    tmax = comm.allreduce(np.array([1.0 + rank, 10.0 - rank]), 'max')
    comm.barrier()
    comm.close()
    q.put((rank, allrec, tmax))


def test_tcp_comm_world_3():
    """The socket communicator used when ranks share a GPU: the same gather, same results on every rank."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_tcp_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(3)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for rank, allrec, tmax in got:
        assert np.array_equal(allrec, got[0][1]) and allrec.shape == (35, 3)
        assert (allrec[:, 0] % 3 == allrec[:, 2]).all()
        assert tmax.tolist() == [3.0, 10.0]


def _from_env_worker(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from nestfit_amd.comm import comm_from_env
    comm, kind = comm_from_env(rccl_timeout=20)
    tot = comm.allreduce(np.array([float(rank + 1)]), 'sum')
    comm.barrier()
    comm.close()
    q.put((rank, kind, float(tot[0])))


def test_comm_from_env_falls_back_to_sockets_together():
    """Where RCCL cannot start (here: no GPU, so rank 0 gets no unique id) every rank takes the socket
    communicator -- agreed over the sockets themselves, nobody is left waiting in an RCCL rendezvous."""
    import multiprocessing as mp
    import nestfit_amd as na
    if na.device_count() > 0:
        pytest.skip('a GPU is visible: RCCL would start (two ranks on one device are refused by RCCL itself)')
    ctx = mp.get_context('spawn')
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_from_env_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    assert [g[1] for g in got] == ['tcp', 'tcp'] and [g[2] for g in got] == [3.0, 3.0]


@pytest.mark.gpu
def test_comm_from_env_two_ranks_on_one_gpu_agree_on_sockets():
    """Two ranks on the one GPU of the box: RCCL itself refuses the second rank on a device (or its rendezvous
    times out); both ranks must come out with the socket communicator and a working collective."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_from_env_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    # (should an RCCL build accept two ranks on one device, both ranks report 'rccl': what matters is that they agree)
    assert got[0][1] == got[1][1] and got[0][1] in ('tcp', 'rccl') and [g[2] for g in got] == [3.0, 3.0]
