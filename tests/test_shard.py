"""N > 1 path on CPU: pixel striping identical to the reference's
get_multiproc_indices (nestfit/main.py:565-571) and the end-of-run gather of
per-pixel records over torch.distributed (gloo, world_size 2 and 3)."""
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
    import torch.distributed as dist
    from nestfit_amd.cube import gather_pixel_records, shard_pixels
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    lon, lat = shard_pixels(shape, rank, world)
    # a record per pixel: (lon, lat, rank, stand-in for lnZ) -- fixed width, variable count
    rec = np.stack([lon, lat, np.full(lon.size, rank), 1000.0 * lon + lat], axis=1).astype(float)
    allrec = gather_pixel_records(rec)
    dist.barrier()
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
