"""The shared-memory ring between sampler processes and the one serving process (csrc/nfa_ring.h,
nestfit_amd/ring.py; reference pattern: one process per stripe, nestfit/main.py:516-523, each calling
LogLike point by point, cmultinest.pxd:27-28).  CPU: the transport with a numpy evaluator between poll and
complete, real processes on the client side.  GPU: the engine's native serving loop against direct calls."""
import ctypes as C
import multiprocessing as mp
import os
import threading

import numpy as np
import pytest

from nestfit_amd import _ffi
from nestfit_amd.ring import RingClient, RingServer, ring_library

NDIM = 5


def _evaluate(pix, U):
    """Stand-in for the engine: theta and lnL as exact functions of the request."""
    theta = 2.0 * U + pix[:, None]
    return theta, -np.sum(U * U, axis=1) + pix


def _client(name, rank, n_points, out, start):
    # a sampler process: only the ring library, no engine, no GPU
    client = RingClient(name, wait_ms=20000)
    start.wait(60)                                             # all processes attached: they run side by side
    rng = np.random.default_rng(100 + rank)
    rows = []
    for i in range(n_points):
        u = rng.random(NDIM)
        theta = u.copy()
        lnl = client.loglikelihood(theta, pix=rank if i % 2 else -1)
        rows.append((u, theta, lnl, rank if i % 2 else -1))
    client.close()
    out.put((rank, rows))


def _serve_with(server, evaluate, n_total):
    served = 0
    while served < n_total:
        slots, pix, U, stopped = server.poll(max_wait_us=5000, idle_ms=20000)
        assert not stopped and slots.size > 0, 'clients went silent'
        theta, lnl = evaluate(pix.copy(), U.copy())
        server.complete(slots, theta, lnl)
        served += slots.size
    return served


def test_ring_between_processes():
    name = f'nfa_test_ring_{os.getpid()}'
    n_clients, n_points = 3, 200
    ctx = mp.get_context('spawn')
    out, start = ctx.Queue(), ctx.Barrier(n_clients)
    with RingServer(name, n_slots=4, ndim=NDIM) as server:
        procs = [ctx.Process(target=_client, args=(name, r, n_points, out, start)) for r in range(n_clients)]
        for p in procs:
            p.start()
        _serve_with(server, _evaluate, n_clients * n_points)
        got = dict(out.get(timeout=60) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        stats = server.stats
    assert stats['evals'] == n_clients * n_points and stats['largest_batch'] >= 2 and stats['clients'] == 0
    assert stats['batches'] < stats['evals']                   # points of different processes shared batches
    for rank, rows in got.items():
        for u, theta, lnl, pix in rows:
            t, l = _evaluate(np.array([pix]), u[None, :])
            assert np.array_equal(theta, t[0]) and lnl == l[0]
    assert not os.path.exists(f'/dev/shm/{name}')              # the server's close removes the object


def test_ring_callback_signature_stop_and_errors():
    lib = ring_library()
    name = f'nfa_test_ring_cb_{os.getpid()}'
    with pytest.raises(_ffi.EngineError, match='no such ring'):
        RingClient(name + '_absent', wait_ms=30)
    server = RingServer(name, n_slots=1, ndim=NDIM)
    with pytest.raises(_ffi.EngineError, match='being served'):
        RingServer(name, n_slots=1, ndim=NDIM)             # the name belongs to a live server
    client = RingClient(name)
    assert client.ndim == NDIM and client.slot == 0
    with pytest.raises(_ffi.EngineError, match='no free slot'):
        RingClient(name, wait_ms=30)
    with pytest.raises(ValueError, match='Invalid shape'):
        client.loglikelihood(np.zeros(NDIM + 1))

    def serve_two():
        for k in range(2):
            slots, pix, U, _stopped = server.poll(idle_ms=20000)
            theta, lnl = _evaluate(pix.copy(), U.copy())
            server.complete(slots, theta, lnl, rc=0 if k == 0 else 3)
    t = threading.Thread(target=serve_two)
    t.start()
    # MultiNest's LogLike: void (*)(double *Cube, int *ndim, int *npars, double *lnew, void *context)
    proto = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p)
    fn, ctx = client.callback(pix=7)
    loglike = C.cast(fn, proto)
    cube = np.linspace(0.1, 0.5, NDIM)
    u0 = cube.copy()
    nd, lnew = C.c_int(NDIM), C.c_double()
    loglike(cube.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nd), C.byref(nd), C.byref(lnew), C.cast(ctx, C.c_void_p))
    assert np.array_equal(cube, 2 * u0 + 7) and lnew.value == -np.sum(u0 * u0) + 7
    # a failed batch: NaN through the callback (no error channel), the cube left alone
    cube = u0.copy()
    loglike(cube.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nd), C.byref(nd), C.byref(lnew), C.cast(ctx, C.c_void_p))
    assert np.isnan(lnew.value) and np.array_equal(cube, u0)
    t.join()
    # a wrong ndim never reaches the ring
    bad = C.c_int(NDIM - 1)
    loglike(cube.ctypes.data_as(C.POINTER(C.c_double)), C.byref(bad), C.byref(bad), C.byref(lnew), C.cast(ctx, C.c_void_p))
    assert np.isnan(lnew.value)
    # a client blocked in a call leaves when the ring is stopped
    result = {}

    def blocked():
        try:
            client.loglikelihood(u0.copy())
        except _ffi.EngineError as e:
            result['err'] = str(e)
    t = threading.Thread(target=blocked)
    t.start()
    slots, _, _, stopped = server.poll(idle_ms=20000)         # the point has arrived ...
    assert slots.size == 1 and not stopped
    server.stop()                                              # ... and is never answered
    t.join(timeout=30)
    assert 'stopped' in result['err']
    assert server.poll(idle_ms=10)[3] is True
    client.close()
    server.close()
    assert lib.nfa_ring_close(None) == 0


def test_two_servers_share_one_ring():
    """Two serving loops on one ring (each would have a runner of its own): every point is claimed by exactly
    one of them."""
    name = f'nfa_test_ring_two_{os.getpid()}'
    n_clients, n_points = 4, 150
    ctx = mp.get_context('spawn')
    out, start = ctx.Queue(), ctx.Barrier(n_clients)
    served = [0, 0]
    with RingServer(name, n_slots=n_clients, ndim=NDIM) as server:
        def loop(which):
            srv = server if which == 0 else other        # the second loop: same ring, buffers of its own
            while True:
                slots, pix, U, stopped = srv.poll(max_wait_us=200, idle_ms=200)
                if stopped:
                    return
                if slots.size == 0:
                    continue
                theta, lnl = _evaluate(pix.copy(), U.copy())
                srv.complete(slots, theta, lnl)
                served[which] += int(slots.size)
        other = RingServer.__new__(RingServer)
        other.__dict__.update(server.__dict__)
        other._slots, other._pix, other._U = np.zeros(128, np.int32), np.zeros(128, np.int32), np.zeros((128, NDIM))
        threads = [threading.Thread(target=loop, args=(w,)) for w in (0, 1)]
        for t in threads:
            t.start()
        procs = [ctx.Process(target=_client, args=(name, r, n_points, out, start)) for r in range(n_clients)]
        for p in procs:
            p.start()
        got = dict(out.get(timeout=60) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        server.stop()
        for t in threads:
            t.join(timeout=30)
        assert sum(served) == n_clients * n_points == server.stats['evals'] and min(served) > 0
    for rank, rows in got.items():
        for u, theta, lnl, pix in rows:
            t, l = _evaluate(np.array([pix]), u[None, :])
            assert np.array_equal(theta, t[0]) and lnl == l[0]


_C_SAMPLER = r"""
/* a serial sampler in plain C: links libnestfit_amd_ring.so only and hands the ring's callback to its
   sampling loop exactly where MultiNest takes LogLike (cmultinest.pxd:27-28) */
#include <stdio.h>
#include <stdlib.h>
#include "nestfit_amd.h"
typedef void (*loglike_fn)(double *Cube, int *ndim, int *npars, double *lnew, void *context);
static void sample(loglike_fn loglike, void *context, int ndim, int n_points) {
    double cube[64], lnew;
    for (int k = 0; k < n_points; ++k) {
        for (int j = 0; j < ndim; ++j) cube[j] = (double)((k * 7 + j * 3) % 101) / 101.0;
        loglike(cube, &ndim, &ndim, &lnew, context);
        printf("%d %a", k, lnew);
        for (int j = 0; j < ndim; ++j) printf(" %a", cube[j]);
        printf("\n");
    }
}
int main(int argc, char **argv) {
    nfa_ring *ring = NULL;
    if (argc < 3 || nfa_ring_attach(&ring, argv[1], 20000) != NFA_OK) return 2;
    nfa_ring_client ctx = { ring, 5 };
    sample(nfa_ring_callback, &ctx, nfa_ring_ndim(ring), atoi(argv[2]));
    return nfa_ring_close(ring);
}
"""


def test_a_c_sampler_process_through_the_ring(tmp_path):
    """The boundary as a compiled host program sees it: C99, the public header, the ring library alone."""
    import shutil
    import subprocess
    from pathlib import Path
    from nestfit_amd.build import OUT_RING
    cc = shutil.which('gcc') or shutil.which('cc')
    if cc is None:
        pytest.skip('no C compiler')
    root = Path(__file__).resolve().parents[1]
    (tmp_path / 'sampler.c').write_text(_C_SAMPLER)
    exe = tmp_path / 'sampler'
    subprocess.run([cc, '-std=c99', '-pedantic', '-Wall', '-Werror', '-I', str(root / 'include'), '-o', str(exe),
                    str(tmp_path / 'sampler.c'), str(OUT_RING), f'-Wl,-rpath,{OUT_RING.parent}'], check=True)
    ldd = subprocess.run(['ldd', str(exe)], capture_output=True, text=True).stdout
    assert 'libnestfit_amd_ring' in ldd and 'amdhip' not in ldd and 'libnestfit_amd.so' not in ldd
    name, n_points = f'nfa_test_ring_c_{os.getpid()}', 120
    with RingServer(name, n_slots=1, ndim=NDIM) as server:
        proc = subprocess.Popen([str(exe), name, str(n_points)], stdout=subprocess.PIPE, text=True)
        _serve_with(server, _evaluate, n_points)
        out, _ = proc.communicate(timeout=60)
        assert proc.returncode == 0
    rows = [line.split() for line in out.strip().split('\n')]
    assert len(rows) == n_points
    for k, row in enumerate(rows):
        u = np.array([((k * 7 + j * 3) % 101) / 101.0 for j in range(NDIM)])
        theta, lnl = _evaluate(np.array([5]), u[None, :])
        assert int(row[0]) == k and float.fromhex(row[1]) == lnl[0]
        assert np.array_equal([float.fromhex(v) for v in row[2:]], theta[0])


def _dies_holding_a_slot(name):
    RingClient(name, wait_ms=20000)
    os._exit(0)                                                # no close: the slot stays marked with a dead pid


def test_slot_of_a_dead_client_is_inherited():
    name = f'nfa_test_ring_dead_{os.getpid()}'
    with RingServer(name, n_slots=1, ndim=NDIM) as server:
        p = mp.get_context('spawn').Process(target=_dies_holding_a_slot, args=(name,))
        p.start()
        p.join(timeout=60)
        assert p.exitcode == 0 and server.stats['clients'] == 1
        with RingClient(name, wait_ms=100) as client:          # the only slot: taken over from the dead process
            assert client.slot == 0 and server.stats['clients'] == 1
        assert server.stats['clients'] == 0


def _many_client(name, rank, n_calls, k, out):
    client = RingClient(name, wait_ms=20000)
    rng = np.random.default_rng(300 + rank)
    rows = []
    for _ in range(n_calls):
        U = rng.random((k, NDIM))
        theta = U.copy()
        lnl = client.loglikelihood_many(theta, pix=rank)
        rows.append((U, theta, lnl))
    assert client.max_points == 8
    client.close()
    out.put((rank, rows))


def test_several_points_per_call():
    """nfa_ring_loglike_many: a client posts k points in one call (a sampler whose next proposals are independent
    draws); the points of a request stay together in a batch and come back in order."""
    name = f'nfa_test_ring_many_{os.getpid()}'
    n_clients, n_calls, k = 3, 40, 5
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    with RingServer(name, n_slots=n_clients, ndim=NDIM, max_points=8) as server:
        procs = [ctx.Process(target=_many_client, args=(name, r, n_calls, k, out)) for r in range(n_clients)]
        for p in procs:
            p.start()
        served = 0
        while served < n_clients * n_calls * k:
            slots, pix, U, stopped = server.poll(max_batch=12, max_wait_us=5000, idle_ms=20000)    # room for two requests
            assert not stopped and slots.size in (k, 2 * k)
            assert all(len(set(slots[a:a + k])) == 1 for a in range(0, slots.size, k))           # whole requests, row by row
            theta, lnl = _evaluate(pix.copy(), U.copy())
            server.complete(slots, theta, lnl)
            served += slots.size
        got = dict(out.get(timeout=60) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert server.stats['evals'] == n_clients * n_calls * k
    for rank, rows in got.items():
        for U, theta, lnl in rows:
            t, l = _evaluate(np.full(k, rank), U)
            assert np.array_equal(theta, t) and np.array_equal(lnl, l)
    with RingServer(name + 'b', n_slots=1, ndim=NDIM) as server:      # one point per slot unless asked otherwise
        with RingClient(name + 'b') as client:
            with pytest.raises(_ffi.EngineError, match='more points'):
                client.loglikelihood_many(np.zeros((2, NDIM)))


def _server_that_dies(name, ready):
    server = RingServer(name, n_slots=2, ndim=NDIM)
    ready.set()
    slots, _pix, _U, _stopped = server.poll(idle_ms=30000)     # takes the request ...
    assert slots.size == 1
    os._exit(0)                                                # ... and dies with it


def test_a_client_does_not_wait_for_a_dead_server():
    """The process that created the ring dies while a client waits for its answer: the call returns an error within
    its wake-up period instead of blocking for ever."""
    import time
    name = f'nfa_test_ring_orphan_{os.getpid()}'
    ctx = mp.get_context('spawn')
    ready = ctx.Event()
    p = ctx.Process(target=_server_that_dies, args=(name, ready))
    p.start()
    assert ready.wait(60)
    client = RingClient(name, wait_ms=20000)
    t0 = time.time()
    with pytest.raises(_ffi.EngineError, match='server process is gone|no serving loop'):
        client.loglikelihood(np.full(NDIM, 0.25))
    assert time.time() - t0 < 20
    p.join(timeout=30)
    client.close()
    try:
        os.unlink(f'/dev/shm/{name}')
    except OSError:
        pass


def _posts_and_dies(name, posted):
    import threading as th
    client = RingClient(name, wait_ms=20000)
    th.Thread(target=lambda: client.loglikelihood(np.full(NDIM, 0.5), pix=3), daemon=True).start()
    while RingServerStatsProbe.posts(name) == 0:
        pass
    posted.set()
    import time
    time.sleep(0.5)                                            # the server claims the request meanwhile
    os._exit(0)


class RingServerStatsProbe:
    """Reads the post counter of a ring's header straight from the shared-memory object (test helper)."""
    @staticmethod
    def posts(name):
        import struct
        with open(f'/dev/shm/{name}', 'rb') as f:
            head = f.read(64)
        return struct.unpack_from('<I', head, 44)[0]           # RingHeader.posts (csrc/nfa_ring.h)


def test_a_late_result_is_not_delivered_to_the_slots_next_owner():
    """A client dies after a server has claimed its request; another process inherits the slot and posts a request of
    its own.  The first request's result, completed late, must be dropped (slot generations), the second served."""
    name = f'nfa_test_ring_gen_{os.getpid()}'
    ctx = mp.get_context('spawn')
    posted = ctx.Event()
    with RingServer(name, n_slots=1, ndim=NDIM) as server:
        p = ctx.Process(target=_posts_and_dies, args=(name, posted))
        p.start()
        assert posted.wait(60)
        slots, pix, U, _ = server.poll(idle_ms=20000)          # claimed under the dying client's generation
        old = (slots.copy(), *_evaluate(pix.copy(), U.copy()))
        p.join(timeout=30)
        result = {}
        client = RingClient(name, wait_ms=1000)                # inherits the only slot
        u = np.full(NDIM, 0.125)

        def call():
            theta = u.copy()
            result['lnl'] = client.loglikelihood(theta, pix=1)
            result['theta'] = theta
        t = threading.Thread(target=call)
        t.start()
        import time
        while RingServerStatsProbe.posts(name) < 2:
            time.sleep(0.001)
        server.complete(*old)                                   # late: must not reach the new owner
        time.sleep(0.05)
        assert 'lnl' not in result
        slots, pix, U, _ = server.poll(idle_ms=20000)
        assert slots.size == 1 and pix[0] == 1
        server.complete(slots, *_evaluate(pix.copy(), U.copy()))
        t.join(timeout=30)
        want_theta, want_lnl = _evaluate(np.array([1]), u[None, :])
        assert np.array_equal(result['theta'], want_theta[0]) and result['lnl'] == want_lnl[0]
        client.close()


@pytest.mark.gpu
def test_ring_serves_processes_from_the_engine(engine, nfo):
    """Four sampler processes (no GPU context) against the native serving loop: bitwise what the runner gives
    directly, several processes' points per launch."""
    from nestfit_amd.synth import TRUTH_2COMP, freq_axis
    rng = np.random.default_rng(3)
    axes = [freq_axis(t, 512) for t in (1, 2)]
    data = []
    for t, x in zip((1, 2), axes):
        s = nfo.AmmoniaSpectrum(x, np.zeros(512), 0.2, t)
        nfo.amm_predict(s, TRUTH_2COMP)
        data.append(s.get_spec() + rng.normal(0, 0.2, 512))
    ut = engine.get_irdc_priors(size=500)
    runner = engine.AmmoniaRunner.from_data([[x, d, 0.2, t] for x, d, t in zip(axes, data, (1, 2))], ut, ncomp=2)
    name = f'nfa_test_ring_gpu_{os.getpid()}'
    n_clients, n_points = 4, 150
    ctx = mp.get_context('spawn')
    out, start = ctx.Queue(), ctx.Barrier(n_clients)
    # two serving threads, each with a runner of its own over the same spectra: they share the clients
    second = engine.AmmoniaRunner.from_data([[x, d, 0.2, t] for x, d, t in zip(axes, data, (1, 2))], ut, ncomp=2)
    with RingServer(name, n_slots=n_clients, runner=runner) as server:
        assert server.ndim == 12
        procs = [ctx.Process(target=_gpu_client, args=(name, r, n_points, out, start)) for r in range(n_clients)]
        for p in procs:
            p.start()
        threads = server.serve_in_threads([runner, second], max_wait_us=200, idle_ms=60000)
        got = dict(out.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        server.stop()
        for t in threads:
            t.join(timeout=30)
        stats = server.stats
    assert stats['evals'] == n_clients * n_points and stats['largest_batch'] >= 2
    for rank, rows in got.items():
        U = np.array([r[0] for r in rows])
        direct_theta = U.copy()
        direct = runner.loglikelihood_batch(direct_theta)
        assert np.array_equal(np.array([r[1] for r in rows]), direct_theta)
        assert np.array_equal(np.array([r[2] for r in rows]), direct)


def _gpu_client(name, rank, n_points, out, start):
    client = RingClient(name, wait_ms=30000)
    start.wait(60)
    rng = np.random.default_rng(200 + rank)
    rows = []
    for _ in range(n_points):
        u = rng.random(client.ndim)
        theta = u.copy()
        rows.append((u, theta, client.loglikelihood(theta)))
    client.close()
    out.put((rank, rows))


def _patient_client(name, out):
    client = RingClient(name, wait_ms=20000)
    rows = []
    for i in range(2):
        u = np.full(NDIM, 0.125 * (i + 1))
        theta = u.copy()
        try:
            rows.append((u, theta, client.loglikelihood(theta)))
        except _ffi.EngineError as e:
            rows.append(('error', str(e), i))
    client.close()
    out.put(rows)


def test_a_slow_poll_complete_server_is_not_taken_for_dead(monkeypatch):
    """A server built on nfa_ring_poll / nfa_ring_complete alone (never inside nfa_ring_serve, which is what the
    serving-loop counter counts) that starts polling later than the clients' grace period, and then sits on a claimed
    batch for longer than that period: the clients wait -- the heartbeat of poll / complete and the CLAIMED state tell
    them a server is there (advisor finding of round 3).  The grace period is shortened to 0.4 s for the test."""
    import time
    monkeypatch.setenv('NFA_RING_GRACE_MS', '400')             # read by the client process at its first wait
    name = f'nfa_test_ring_slow_{os.getpid()}'
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    with RingServer(name, n_slots=2, ndim=NDIM) as server:
        p = ctx.Process(target=_patient_client, args=(name, out))
        p.start()
        time.sleep(1.5)                                        # the server's start-up: several grace periods, no poll yet
        for held in (1.2, 0.0):                                # first batch: held (claimed) for three grace periods
            slots, pix, U, stopped = server.poll(max_wait_us=2000, idle_ms=20000)
            assert not stopped and slots.size == 1
            time.sleep(held)
            theta, lnl = _evaluate(pix.copy(), U.copy())
            server.complete(slots, theta, lnl)
        rows = out.get(timeout=60)
        p.join(timeout=60)
        assert p.exitcode == 0
    for u, theta, lnl in rows:
        assert not isinstance(u, str), (theta, lnl)
        t, l = _evaluate(np.array([-1]), u[None, :])
        assert np.array_equal(theta, t[0]) and lnl == l[0]


def test_a_client_gives_up_on_a_server_that_went_quiet(monkeypatch):
    """... and a server that HAS served and then stops polling (no claimed request in its hands) is given up on after
    the grace period: the post comes back as an error instead of blocking for ever."""
    import time
    monkeypatch.setenv('NFA_RING_GRACE_MS', '300')
    name = f'nfa_test_ring_quiet_{os.getpid()}'
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    with RingServer(name, n_slots=2, ndim=NDIM) as server:
        p = ctx.Process(target=_patient_client, args=(name, out))
        p.start()
        slots, pix, U, stopped = server.poll(max_wait_us=2000, idle_ms=20000)
        theta, lnl = _evaluate(pix.copy(), U.copy())
        server.complete(slots, theta, lnl)                     # the first point is served, the second never polled
        t0 = time.time()
        rows = out.get(timeout=60)
        assert time.time() - t0 < 20
        p.join(timeout=60)
    assert not isinstance(rows[0][0], str)
    assert rows[1][0] == 'error' and 'no serving loop' in rows[1][1]


def _gpu_client_pix(name, rank, n_points, n_pix, out, start):
    client = RingClient(name, wait_ms=30000)
    start.wait(60)
    rng = np.random.default_rng(300 + rank)
    rows = []
    for i in range(n_points):
        u = rng.random(client.ndim)
        theta = u.copy()
        pix = int(rng.integers(0, n_pix))
        rows.append((u, theta, client.loglikelihood(theta, pix=pix), pix))
    # a pixel the cube does not have: this request fails alone
    try:
        client.loglikelihood(rng.random(client.ndim), pix=n_pix + 5)
        rows.append('no error')
    except _ffi.EngineError:
        rows.append('refused')
    u = rng.random(client.ndim)
    theta = u.copy()
    rows.append((u, theta, client.loglikelihood(theta, pix=0), 0))      # ... and the service goes on
    client.close()
    out.put((rank, rows))


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['fast', 'table'])
def test_ring_served_by_the_resident_kernel(engine, nfo, mode):
    """nfa_ring_serve_device: the resident kernel polls the slots itself.  Sampler processes without a GPU context post
    point by point (MultiNest's pattern); every answer is bitwise what the runner gives directly -- a single-pixel
    runner and a cube runner with pixel indices --, a request with a pixel beyond the cube fails alone, kernel
    instances of 3 ms follow each other without a request being lost, and the loop ends by itself when the ring
    goes idle or is stopped."""
    import time
    from nestfit_amd.cube import CubeRunner
    from nestfit_amd.synth import TRUTH_2COMP, freq_axis
    engine.set_exp_mode(mode)
    try:
        rng = np.random.default_rng(5)
        n = 256
        axes = [freq_axis(t, n) for t in (1, 2)]
        clean = []
        for t, x in zip((1, 2), axes):
            s = nfo.AmmoniaSpectrum(x, np.zeros(n), 0.2, t)
            nfo.amm_predict(s, TRUTH_2COMP)
            clean.append(s.get_spec())
        ut = engine.get_irdc_priors(size=500)
        n_pix = 5
        data = np.concatenate(clean)[None, :] + rng.normal(0, 0.2, (n_pix, 2 * n))
        single = engine.AmmoniaRunner.from_data([[x, data[0, k * n:(k + 1) * n], 0.2, t] for k, (x, t) in enumerate(zip(axes, (1, 2)))], ut, ncomp=2)
        cube = CubeRunner(axes, (1, 2), data, np.full((n_pix, 2), 0.2), ut, ncomp=2)
        ctx = mp.get_context('spawn')
        # (a) single-pixel runner, three processes, short-lived kernel instances
        name = f'nfa_test_ring_dev_{os.getpid()}_{mode}'
        n_clients, n_points = 3, 120
        out, start = ctx.Queue(), ctx.Barrier(n_clients)
        with RingServer(name, n_slots=n_clients, runner=single) as server:
            procs = [ctx.Process(target=_gpu_client, args=(name, r, n_points, out, start)) for r in range(n_clients)]
            for p in procs:
                p.start()
            th = threading.Thread(target=server.serve_device, kwargs=dict(lifetime_ms=3, idle_ms=30000))
            th.start()
            got = dict(out.get(timeout=120) for _ in procs)
            for p in procs:
                p.join(timeout=60)
                assert p.exitcode == 0
            server.stop()
            th.join(timeout=30)
            assert not th.is_alive()
            assert server.stats['evals'] == n_clients * n_points
        for rank, rows in got.items():
            U = np.array([r[0] for r in rows])
            want_theta = U.copy()
            want = single.loglikelihood_batch(want_theta)
            assert np.array_equal(np.array([r[1] for r in rows]), want_theta)
            assert np.array_equal(np.array([r[2] for r in rows]), want)
        # (b) cube runner: pixel indices, a bad one, and the loop ending by itself after 5 s without requests
        name = f'nfa_test_ring_devc_{os.getpid()}_{mode}'
        out, start = ctx.Queue(), ctx.Barrier(2)
        with RingServer(name, n_slots=2, runner=cube) as server:
            procs = [ctx.Process(target=_gpu_client_pix, args=(name, r, 40, n_pix, out, start)) for r in range(2)]
            for p in procs:
                p.start()
            th = threading.Thread(target=server.serve_device, kwargs=dict(lifetime_ms=5, idle_ms=5000))      # (the clients are fresh interpreters: seconds to start)
            th.start()
            got = dict(out.get(timeout=120) for _ in procs)
            for p in procs:
                p.join(timeout=60)
                assert p.exitcode == 0
            t0 = time.time()
            th.join(timeout=30)                                 # nobody posts any more: idle_ms ends the loop
            assert not th.is_alive() and time.time() - t0 < 20
        for rank, rows in got.items():
            assert rows[-2] == 'refused'
            good = [r for r in rows if not isinstance(r, str)]
            U = np.array([r[0] for r in good])
            pix = np.array([r[3] for r in good], dtype=np.int32)
            want_theta = U.copy()
            want = cube.loglikelihood_batch(pix, want_theta)
            assert np.array_equal(np.array([r[1] for r in good]), want_theta)
            assert np.array_equal(np.array([r[2] for r in good]), want)
    finally:
        engine.set_exp_mode('fast')
