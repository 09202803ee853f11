#!/usr/bin/env python3
"""Sampler processes behind the shared-memory ring (nestfit_amd/ring.py, csrc/nfa_ring.h): N processes
that never touch the GPU post their LogLike points (one at a time each, like MultiNest:
nestfit/core/cmultinest.pxd:27-28, one process per stripe: nestfit/main.py:516-523 -- or k at a time through
nfa_ring_loglike_many: a sampler whose next k proposals are independent draws), ONE process with the
runner serves them, every round of posted points one launch.  Prints the aggregate rate, the
time a client waits per call and the mean batch; beside it profiles/r02/multiproc_points.txt has the same
processes with a runner each (no ring)."""
import multiprocessing as mp
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def client(name, rank, n_calls, start, out, k_pts=1):
    from nestfit_amd.ring import RingClient
    c = RingClient(name, wait_ms=60000)
    rng = np.random.default_rng(rank)
    U = rng.uniform(size=(n_calls + 200, k_pts, c.ndim))
    call = (lambda u: c.loglikelihood(u[0])) if k_pts == 1 else c.loglikelihood_many
    for k in range(200):
        call(U[k])
    start.wait(120)
    t0 = time.perf_counter()
    for k in range(200, n_calls + 200):
        call(U[k])
    t1 = time.perf_counter()
    c.close()
    out.put((rank, t0, t1))


C_CLIENT = r"""
/* a compiled serial sampler (what MultiNest is to the ring): attach, warm up, wait for the start time, call */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "nestfit_amd.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv) {
    nfa_ring *ring = NULL;
    if (argc < 6 || nfa_ring_attach(&ring, argv[1], 60000) != NFA_OK) return 2;
    const int rank = atoi(argv[2]), n_calls = atoi(argv[3]), ndim = nfa_ring_ndim(ring), k_pts = atoi(argv[5]);
    const double t_start = atof(argv[4]);
    nfa_ring_client ctx = { ring, -1 };
    static double cube[64 * 64], lnew[64];
    unsigned long long state = 88172645463325252ull + (unsigned long long)rank * 7919ull;
    for (int k = -200; k < n_calls; ++k) {
        if (k == 0) while (now() < t_start) ;
        for (int j = 0; j < ndim * k_pts; ++j) {             /* xorshift: fresh unit-cube points per call */
            state ^= state << 13; state ^= state >> 7; state ^= state << 17;
            cube[j] = (double)(state >> 11) / 9007199254740992.0;
        }
        if (k_pts == 1) nfa_ring_callback(cube, (int *)&ndim, (int *)&ndim, lnew, &ctx);      /* MultiNest's own call */
        else if (nfa_ring_loglike_many(ring, -1, cube, lnew, k_pts) != NFA_OK) return 3;
    }
    printf("%.9f %.9f\n", t_start, now());
    return nfa_ring_close(ring);
}
"""


def native_clients(name, n_proc, n_calls, k_pts=1):
    """N compiled clients as subprocesses; returns [(t_start, t_end)] on the monotonic clock."""
    import subprocess
    import tempfile
    from nestfit_amd.build import OUT_RING
    tmp = Path(tempfile.mkdtemp())
    (tmp / 'client.c').write_text(C_CLIENT)
    exe = tmp / 'client'
    subprocess.run(['gcc', '-O2', '-std=gnu99', '-I', str(ROOT / 'include'), '-o', str(exe), str(tmp / 'client.c'),
                    str(OUT_RING), f'-Wl,-rpath,{OUT_RING.parent}'], check=True)
    t_start = time.clock_gettime(time.CLOCK_MONOTONIC) + 1.0 + 0.02 * n_proc
    procs = [subprocess.Popen([str(exe), name, str(k), str(n_calls), f'{t_start:.9f}', str(k_pts)], stdout=subprocess.PIPE, text=True)
             for k in range(n_proc)]
    out = []
    for p in procs:
        text, _ = p.communicate(timeout=900)
        assert p.returncode == 0, p.returncode
        out.append(tuple(float(v) for v in text.split()))
    return out


def main():
    """measure_ring.py [native] [processes[:serving threads[:points per call]] ...]
    (native: compiled C clients instead of Python ones)"""
    import nestfit_amd as na
    from nestfit_amd.ring import RingServer
    from nestfit_amd.synth import TRUTH_2COMP, freq_axis
    rng = np.random.default_rng(0)
    args = []
    for t in (1, 2):
        x = freq_axis(t, 1024)
        s = na.AmmoniaSpectrum(x, np.zeros(1024), 0.2, t)
        na.amm_predict(s, TRUTH_2COMP)
        args.append([x, s.get_spec() + rng.normal(0, 0.2, 1024), 0.2, t])
    priors = na.get_irdc_priors(size=500, vsys=0.0)
    runners = []
    ctx = mp.get_context('spawn')
    argv = [a for a in sys.argv[1:] if a != 'native']
    native = 'native' in sys.argv[1:]
    # 'N:dev' = N processes served by the resident kernel (nfa_ring_serve_device); 'N:S:K' = S serving threads, K points per call
    cases = [tuple((0 if v == 'dev' else int(v)) for v in (a + ':1:1').split(':')[:3]) for a in argv] or [(1, 1, 1), (2, 1, 1), (4, 1, 1), (8, 1, 1), (14, 1, 1)]
    for n_proc, n_serv, k_pts in cases:
        while len(runners) < max(1, n_serv):                   # one runner (its own streams) per serving thread
            runners.append(na.AmmoniaRunner.from_data(args, priors, ncomp=2))
        n_calls = (20000 if n_proc <= 16 else 8000) // max(1, k_pts // 4)
        name = f'nfa_measure_ring_{n_proc}_{n_serv}_{k_pts}'
        with RingServer(name, n_slots=n_proc, runner=runners[0], max_points=k_pts) as server:
            if n_serv == 0:
                threads = [threading.Thread(target=server.serve_device, kwargs=dict(lifetime_ms=20, idle_ms=120000))]
                threads[0].start()
            else:
                threads = server.serve_in_threads(runners[:n_serv], max_wait_us=30, idle_ms=120000)
            if native:
                res = [(k, a, b) for k, (a, b) in enumerate(native_clients(name, n_proc, n_calls, k_pts))]
            else:
                start, out = ctx.Barrier(n_proc), ctx.Queue()
                procs = [ctx.Process(target=client, args=(name, k, n_calls, start, out, k_pts)) for k in range(n_proc)]
                for p in procs:
                    p.start()
                res = [out.get(timeout=900) for _ in procs]
                for p in procs:
                    p.join()
            server.stop()
            for t in threads:
                t.join()
            st = server.stats
        wall = max(r[2] for r in res) - min(r[1] for r in res)
        per_call = np.mean([r[2] - r[1] for r in res]) / n_calls
        print(f'{n_proc:2d} {"compiled" if native else "Python"} processes x {k_pts} point(s) per call, {"the resident kernel" if n_serv == 0 else str(n_serv) + " serving thread(s)"}: '
              f'{n_proc * n_calls * k_pts / wall / 1e3:7.1f} k evals/s in all, '
              f'{per_call * 1e6:6.1f} us per call in each, {st["evals"] / st["batches"]:5.2f} points per launch '
              f'(largest {st["largest_batch"]})', flush=True)


if __name__ == '__main__':
    main()
