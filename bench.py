#!/usr/bin/env python3
"""Benchmark of the hot path: log-likelihood evaluations per second at the metric shape of
BASELINE.json (config C2: B = 4096 live-point draws against one pixel, NH3 (1,1)+(2,2), 1024
channels each, 2 velocity components, get_irdc_priors).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (prior transform -> model spectra -> chi^2) over one batch of
B unit-cube rows against one map pixel.  The pixels are those of BASELINE config C3 (a synthetic
128 x 128 cube of 2-component NH3 (1,1)+(2,2) spectra): rank r of N owns the longitude stripe
i_lon % N == r exactly like the reference's get_multiproc_indices (nestfit/main.py:565-571), holds
it in HBM and walks it, one pixel per step.  Per-GPU work is fixed (weak scaling); there is no
data-path collective: the ranks only meet in the barrier / max-time reduction around a timed block
and in the end-of-run gather of per-pixel records (RCCL through the engine's C ABI, no torch).

Every input of a timed block (the unit-cube rows of all its steps, the stripe, the prior tables) is
resident in HBM before the clock starts.  A timed block is the K-step sequence of `--steps`, repeated back to
back `repeats_per_block` times (chosen so that a block lasts >= 20 ms whatever K is: K = 20 steps alone are
0.6 ms, of which the fill and drain of the launch pipeline are 12 %; every repeat works on unit-cube rows of its own)
between barrier + device synchronisation on both sides; its time is the maximum over ranks and `ms_per_step` =
block time / (K x repeats).  `--blocks` blocks are timed; `value` is the median block (whole-job evaluations /
block time), `spread` holds min / max; `k_steps_alone` is the same for blocks of exactly K steps.
Both numerical modes are timed in the same run (`modes`); `value`, `ms_per_step`, `dtype` and `roofline` are the
TABLE mode's -- the reference's own arithmetic (core/fastexp.c:234-283: float-narrowed argument, f64 product of three
table entries; models/hyperfine.pyx:93-96: tau summed in f64) --, the fast mode (f32 exponentials, <= 1e-6 on Tb) is
the `fast_*` keys of `roofline` and `modes.fast`.  `roofline`: algorithmic bytes per launch / the average duration of
the likelihood kernel, measured live with HIP events on a one-lane runner (launches do not overlap there, so
the interval is what `rocprofv3 --kernel-trace --stats -- python bench.py --streams 1` reports for
the kernel: profiles/r05/); the pipelined rate of the timed blocks is given beside it.

Rank 0 prints ONE JSON line (DESIGN.md "Measurement" explains every field).
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MODES = ('table', 'fast')
# what each mode computes in.  `table` is the reference's own arithmetic (core/fastexp.c:234-283: the argument narrowed to
# float, an f64 product of three f64 table entries; models/hyperfine.pyx:93-96: tau summed in f64): the like-for-like number
DTYPES = {'table': 'f64 (reference FastExp: float-narrowed argument, f64 three-table product, f64 optical depth)',
          'fast': 'f64 indices and FastExp arguments, f32 exp and optical depth (mode fast, <= 1e-6 on Tb)'}
REFERENCE_PRECISION_MODE = 'table'
WORKLOADS = {
    # name: (trans ids, channels, vhalf, ncomp, truth key, B)
    'C2': ((1, 2), 1024, 30.0, 2, 'TRUTH_2COMP', 4096),
    'C4': ((1, 2, 3), 2048, 40.0, 3, 'TRUTH_3COMP', 4096),
    'C1': ((1,), 256, 30.0, 1, 'TRUTH_1COMP', 4096),
}
PROFILE_DIRS = [ROOT / 'profiles' / 'r05', ROOT / 'profiles' / 'r04', ROOT / 'profiles' / 'r03', ROOT / 'profiles' / 'r02']
MIN_BLOCK_S = 0.025            # length a timed block is sized for from a lone K-step probe (>= 20 ms in effect)
MAX_BLOCK_STEPS = 4096
# (table mode: launches of two and more units per wave slot -- the engine's coalesced steps -- run as lnl_kernel_queue)
LNL_KERNEL_NAME = {'fast': 'void lnl_kernel<2, false, false, 2>', 'table': 'void lnl_kernel_queue<false, 2>'}
OTHER_MODE = {'table': 'fast', 'fast': 'table'}


def profile_file(name):
    """The committed profiler summary `name` of the latest round that has one (or None)."""
    for d in PROFILE_DIRS:
        if (d / name).exists():
            return d / name
    return None


def algorithmic_bytes(trans, n_chan, ncomp):
    """sum_s N_s*8 (data) + ndim*8 (u in) + ndim*8 (theta out) + 8 (lnL)."""
    ndim = 6 * ncomp
    return len(trans) * n_chan * 8 + 2 * ndim * 8 + 8


def stripe_truths(workload, side, lon, lat):
    """Per-pixel truth parameters of the pixels (lon, lat) of the synthetic cube.  C2: ParamSampler
    ranges (reference: nestfit/synth_spectra.py:165-192), seeded per pixel by its global index so that
    a pixel is the same whatever the number of ranks; pixel (0, 0) carries get_test_spectra(kind=0)'s
    truth; the other workloads repeat their one truth."""
    from nestfit_amd import synth
    trans, n_chan, vhalf, ncomp, truth_key, B = WORKLOADS[workload]
    if workload != 'C2':
        return np.tile(getattr(synth, truth_key), (lon.size, 1))
    out = np.empty((lon.size, 6 * ncomp))
    for k, (i, j) in enumerate(zip(lon, lat)):
        out[k] = synth.param_sampler_draw(np.random.default_rng(11 + int(i) * side + int(j)))
    if lon.size and (lon[0], lat[0]) == (0, 0):
        out[0] = synth.TRUTH_2COMP
    return out


def make_stripe(na, workload, side, rank, world, noise):
    """This rank's stripe of the cube as one device-resident spectra set: engine model spectra of the
    per-pixel truths + noise seeded by the pixel's global index."""
    from nestfit_amd.cube import CubeRunner, shard_pixels
    from nestfit_amd.synth import freq_axis
    trans, n_chan, vhalf, ncomp, truth_key, B = WORKLOADS[workload]
    lon, lat = shard_pixels((side, side), rank, world)
    axes = [freq_axis(t, n_chan, vhalf) for t in trans]
    ut = na.get_irdc_priors(size=500, vsys=0.0)
    truths = stripe_truths(workload, side, lon, lat)
    n_pix = lon.size
    n_tot = len(trans) * n_chan
    probe = CubeRunner(axes, trans, np.zeros((1, n_tot)), np.full((1, len(trans)), noise), ut, ncomp=ncomp)
    data = np.empty((n_pix, n_tot))
    for a in range(0, n_pix, 2048):
        b = min(n_pix, a + 2048)
        data[a:b], _ = probe.predict_batch(np.zeros(b - a, dtype=np.int32), truths[a:b])
    del probe
    for k in range(n_pix):
        data[k] += np.random.default_rng(5 + int(lon[k]) * side + int(lat[k])).normal(0, noise, n_tot)
    runner = CubeRunner(axes, trans, data, np.full((n_pix, len(trans)), noise), ut, ncomp=ncomp)
    spec0 = [[axes[s], data[0, s * n_chan:(s + 1) * n_chan].copy(), noise, trans[s]] for s in range(len(trans))]
    return runner, ut, lon, lat, spec0


def _cpu_worker(args):
    """cpu_baseline worker: the oracle (reference-flag build) on one core."""
    spec_data, program, ncomp, U, reps = args
    from oracle import nfo
    spectra = [nfo.AmmoniaSpectrum(x, d, n, t, native=True) for x, d, n, t in spec_data]
    run = nfo.AmmoniaRunner(spectra, nfo.PriorSet(program), ncomp=ncomp, native=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        Uc = U.copy()
        run.loglikelihood_batch(Uc)
    return time.perf_counter() - t0


def cpu_baseline(spec_data, program, ncomp, U, budget_s=12.0):
    """Times the CPU oracle (same algorithm, reference compile flags) on a bounded sample of the
    same workload: 1 core, then all cores (one process per core, like the reference's
    fit_cube(nproc))."""
    from oracle import nfo
    nfo.build(native=True)
    sample = U[:1024].copy()
    t = _cpu_worker((spec_data, program, ncomp, sample, 1))
    per_eval = t / sample.shape[0]
    one_core = 1.0 / per_eval
    cores = min(len(os.sched_getaffinity(0)), 16)     # the GPU box gives 16 cores per GPU
    reps = max(1, int(budget_s / (per_eval * sample.shape[0])))
    with mp.get_context('spawn').Pool(cores) as pool:   # fresh interpreters: never touch the GPU
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(spec_data, program, ncomp, sample, reps)] * cores)
        wall = time.perf_counter() - t0
    all_cores = cores * reps * sample.shape[0] / wall
    return {
        'value': all_cores, 'unit': 'evals/s', 'cores': cores, 'kind': 'port',
        'sample': f'{reps} x 1024 rows of the same U per core against pixel (0, 0), {cores} processes '
                  f'(oracle/nf_oracle.c, -O3 -march=native -ffast-math)',
        'one_core': one_core,
    }


def c5_specified(na, mode, side=32, n=1024, seed=5):
    """BASELINE config 5 as SURVEY.md 8d specifies it, through the cube driver: side x side pixels of config 3's generator
    (two-component truths of the ParamSampler ranges, NH3 (1,1)+(2,2), n channels each, 0.2 K noise), 400 live points,
    tol 0.5, efr 0.3, fixed seed, ncomp_max = 2 with the lnZ_thresh = 11 loop of nestfit/main.py:452-469
    (CubeFitter.fit_cube into a store in a temporary directory), on the built-in device sampler (libmultinest is not
    available here).  {seconds, evaluations, nbest histogram, mean lnZ_err ...} of the run in numerical mode `mode`."""
    import tempfile
    from nestfit_amd import _ffi
    from nestfit_amd.fitter import CubeFitter
    from nestfit_amd.store import HdfStore
    from nestfit_amd.synth import c5_stack
    stack, truths, model, data, axes, ut = c5_stack(side, n, 0.2)
    n_pix = side * side
    na.set_exp_mode(mode)
    fitter = CubeFitter(stack, ut, na.AmmoniaRunner, lnZ_thresh=11, ncomp_max=2, nlive_snr_fact=0,
                        mn_kwargs={'nlive': 400, 'tol': 0.5, 'efr': 0.3, 'seed': seed})
    stages = {}
    inner = fitter._fit_on_device

    def counted(lon, lat, ncomp, nlive, kw):
        t0 = time.perf_counter()
        res, null_lnZ, n_chan_tot = inner(lon, lat, ncomp, nlive, kw)
        st = stages.setdefault(int(ncomp), {'pixels': 0, 'seconds': 0.0, 'likelihood_evals': 0, 'iterations': 0, 'lnZ_err': []})
        st['pixels'] += len(res); st['seconds'] += time.perf_counter() - t0
        st['likelihood_evals'] += int(sum(r.n_evals for r in res)); st['iterations'] += int(sum(r.n_iter for r in res))
        st['lnZ_err'] += [float(r.lnZ_err) for r in res]
        return res, null_lnZ, n_chan_tot
    fitter._fit_on_device = counted
    with tempfile.TemporaryDirectory() as tmp:
        _ffi.check(_ffi.load().nfa_device_synchronize())
        t0 = time.perf_counter()
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):          # (the driver's progress lines)
            fitter.fit_cube(os.path.join(tmp, 'c5'), nproc=1)
        seconds = time.perf_counter() - t0
        with HdfStore(os.path.join(tmp, 'c5')) as store:
            nbest = np.array([int(g.attrs['nbest']) for g in store.iter_pix_groups()])
    evals = sum(st['likelihood_evals'] for st in stages.values())
    out = {'mode': mode, 'dtype': DTYPES[mode], 'seconds': seconds, 'pixels': n_pix, 'pixels_per_s': n_pix / seconds,
           'likelihood_evals': evals, 'evals_per_pixel': evals / n_pix, 'evals_per_s': evals / seconds,
           'nbest_histogram': {str(k): int((nbest == k).sum()) for k in (0, 1, 2)},
           'mean_lnZ_err': float(np.mean(sum((st['lnZ_err'] for st in stages.values()), [])))}
    for ncomp, st in sorted(stages.items()):
        out[f'ncomp_{ncomp}'] = {'pixels': st['pixels'], 'seconds': st['seconds'], 'evals_per_pixel': st['likelihood_evals'] / max(st['pixels'], 1),
                                 'iterations_per_pixel': st['iterations'] / max(st['pixels'], 1), 'mean_lnZ_err': float(np.mean(st['lnZ_err']))}
    return out


C5_WORKLOAD = ('C5: 32x32 pixels of the C3 generator (two-component truths, NH3 (1,1)+(2,2), 1024 channels each, 0.2 K), nlive 400, '
               'tol 0.5, efr 0.3, seed 5, ncomp_max 2 / lnZ_thresh 11 through CubeFitter.fit_cube (nestfit/main.py:452-469); built-in '
               'device sampler (libmultinest is not available), store written')


def bench_c5(args):
    """BASELINE config 5 as specified on one GPU, both numerical modes (or --modes one): one JSON line of
    its own shape -- not the headline metric."""
    import nestfit_amd as na
    from nestfit_amd import _ffi
    for key, val in (('setup_ti', args.setup_ti), ('setup_threads', args.setup_threads), ('wpb', args.wpb), ('streams', args.streams),
                     ('sampler_parts', args.sampler_parts), ('sampler_refit_every', args.sampler_refit_every), ('sampler_walk_factor', args.sampler_walk_factor), ('sampler_ellipsoids', args.sampler_ellipsoids), ('sampler_walkers', args.sampler_walkers), ('lnl_cap', max(args.lnl_cap, 0))):
        if val:
            _ffi.set_option(key, val)
    modes = list(MODES) if args.modes == 'all' else [args.exp_mode]
    c5_specified(na, 'fast', side=8, n=256)                     # (code objects, streams and buffers come up outside the clock)
    out = {m: c5_specified(na, m) for m in modes}
    head = out[modes[0]]
    print(json.dumps({
        'metric': 'seconds, nested-sampling fit of the 32x32 NH3(1,1)+(2,2) cube of BASELINE config 5 (400 live points, ncomp_max 2)',
        'value': head['seconds'], 'unit': 's', 'n_gpus': 1, 'higher_is_better': False,
        'vs_baseline': None, 'dtype': DTYPES[modes[0]], 'data': 'synthetic',
        'config': {'workload': C5_WORKLOAD, 'exp_mode': modes[0]}, 'modes': out}))


def bench_c5_round4_shape(args):
    """The cube rounds 2-4 timed as "C5" (--workload C5r4; kept so that the sampler's numbers stay comparable across
    rounds): nested sampling with 400 live points of every pixel of a 32x32 synthetic NH3 (1,1)+(2,2) cube of 512 channels
    (nestfit_amd.synth.c5r4_cube: smooth maps of as many components as are fitted, not the C3 generator), with one and
    with two velocity components in two lock-step runs (no component loop, no store), margins of rounds 2-4
    (precision='speed').  BASELINE config 5 as SURVEY specifies it is --workload C5 (c5_specified)."""
    import nestfit_amd as na
    from nestfit_amd import _ffi, sampler
    from nestfit_amd.cube import CubeRunner
    from nestfit_amd.synth import freq_axis
    na.set_exp_mode(args.exp_mode)
    for key, val in (('setup_ti', args.setup_ti), ('setup_threads', args.setup_threads), ('wpb', args.wpb), ('streams', args.streams),
                     ('sampler_parts', args.sampler_parts), ('sampler_refit_every', args.sampler_refit_every), ('sampler_walk_factor', args.sampler_walk_factor), ('sampler_ellipsoids', args.sampler_ellipsoids), ('sampler_walkers', args.sampler_walkers), ('lnl_cap', max(args.lnl_cap, 0))):
        if val:
            _ffi.set_option(key, val)
    if args.prior_stage >= 0:
        _ffi.set_option('prior_stage', args.prior_stage)
    from nestfit_amd.synth import c5r4_cube
    side, n, noise, nlive = 32, 512, 0.1, 400
    n_pix = side * side
    out = {}
    for ncomp in (1, 2):
        if args.c5_ncomp and ncomp != args.c5_ncomp:
            out[ncomp] = {'seconds': float('nan'), 'pixels_per_s': float('nan'), 'evals_per_pixel': float('nan'), 'mean_lnZ_err': float('nan')}
            continue
        axes, data, noise, ut = c5r4_cube(ncomp, side, n, noise)
        cube = CubeRunner(axes, (1, 2), data, np.full((n_pix, 2), noise), ut, ncomp=ncomp)
        _ffi.check(_ffi.load().nfa_device_synchronize())
        t0 = time.perf_counter()
        knobs = {'precision': 'speed'}                    # rounds 2-4's margins: what their numbers were measured with
        if args.sampler_batch_target:
            knobs['batch_target'] = args.sampler_batch_target
        if args.sampler_upd_frac:
            knobs['upd_frac'] = args.sampler_upd_frac
        res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=nlive, tol=0.5, efr=0.3, seed=1, **knobs)
        dt = time.perf_counter() - t0
        gain = np.array([x.lnZ for x in res]) - cube.null_lnZ
        out[ncomp] = {'seconds': dt, 'pixels_per_s': n_pix / dt, 'likelihood_evals': int(sum(x.n_evals for x in res)),
                      'evals_per_pixel': float(np.mean([x.n_evals for x in res])), 'iterations_per_pixel': float(np.mean([x.n_iter for x in res])),
                      'evals_per_s': sum(x.n_evals for x in res) / dt, 'mean_lnZ_err': float(np.mean([x.lnZ_err for x in res])),
                      'detections': int((gain > 11).sum())}
    print(json.dumps({
        'metric': 'pixels/sec, nested sampling (400 live points) of a 32x32 NH3(1,1)+(2,2) cube, 1 component',
        'value': out[1]['pixels_per_s'], 'unit': 'pixels/s', 'n_gpus': 1, 'higher_is_better': True,
        'vs_baseline': None, 'dtype': DTYPES[args.exp_mode], 'data': 'synthetic',
        'config': {'workload': 'C5r4 (rounds 2-4 shape, NOT the specified config 5): 32x32 pixels x 2 spectra x 512 channels, smooth truth maps '
                               '(synth.c5r4_cube), precision speed, built-in device sampler (libmultinest is not available), tol 0.5, efr 0.3', 'exp_mode': args.exp_mode},
        'one_component': out[1], 'two_components': out[2]}))


def relaunch_one_rank_per_gpu(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU as child processes with the
    launcher's environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT).  This process has not
    touched the GPU and never will; the children are plain `python bench.py ...` processes (no exec from a
    process with a GPU context, no torch)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:                # one rank failed: the others would wait for it in a barrier forever
                    q.terminate()
    raise SystemExit(rc)


class Measure:
    """One workload on this rank's stripe: its device-resident inputs and the timed blocks over them."""

    def __init__(self, args, na, lib, comm, rank, world, workload, side, batch=0):
        from nestfit_amd import _ffi
        self._ffi, self.lib, self.comm, self.args, self.rank, self.world = _ffi, lib, comm, args, rank, world
        self.workload = workload
        self.trans, self.n_chan, self.vhalf, self.ncomp, _, self.B = WORKLOADS[workload]
        if batch:
            self.B = batch
        self.ndim = 6 * self.ncomp
        self.per_row = args.pixels_per_step == 'B'
        if self.per_row and side * side // world < self.B:
            raise SystemExit('--pixels-per-step B needs a stripe of at least B pixels')
        na.set_exp_mode('fast')                        # the synthetic data are made in one mode, whatever is timed
        self.cube, self.ut, self.lon, self.lat, self.spec0 = make_stripe(na, workload, side, rank, world, 0.2)
        self.n_pix = self.cube.n_pix
        self.rh = self.cube._run.handle
        # inputs of all steps of a block resident in HBM before the clock starts: unit-cube rows of its own for every
        # step (a pass overwrites them with theta in place), the pixel index of every row, one result vector per step
        self.U_host = np.ascontiguousarray(np.random.default_rng(7 + rank).uniform(size=(self.B, self.ndim)))
        self.step_bytes = self.U_host.nbytes
        self.buf = {'cap': 0, 'U': C.c_void_p(), 'lnL': C.c_void_p(), 'pix': C.c_void_p()}
        self.bytes_eval = algorithmic_bytes(self.trans, self.n_chan, self.ncomp)
        # What the engine makes of the steps: device-pointer batches of one shape that arrive back to back are launched
        # together (option coalesce: up to eight, a group below eight waves per wave slot), and a sequence of launches
        # rotates over four stream lanes (six for launches of about one wave per wave slot).
        units, slots = self.B * len(self.trans), 256 * 32
        self.group = args.coalesce or 8
        gmax = int(os.environ.get('NFA_GROUP_MAX', 8))           # (experiment builds: -DNFA_GROUP_MAX=...)
        self.steps_per_launch = 1
        if self.group > 1 and self.B % 64 == 0 and 2 * units <= gmax * slots:
            self.steps_per_launch = int(max(1, min(self.group, (gmax * slots) // units)))
        launch_units = units * self.steps_per_launch
        self.lanes_used = args.streams or (6 if 4 * launch_units >= 3 * slots and 2 * launch_units <= 3 * slots else 4)

    def close(self):
        for key in ('U', 'lnL', 'pix'):
            if self.buf[key].value:
                self.lib.nfa_free(self.buf[key])
                self.buf[key] = C.c_void_p()
        self.cube = None

    def alloc_inputs(self, n_cap):
        _ffi, lib, buf, B = self._ffi, self.lib, self.buf, self.B
        for key in ('U', 'lnL', 'pix'):
            if buf[key].value:
                _ffi.check(lib.nfa_free(buf[key]))
                buf[key] = C.c_void_p()
        if self.per_row:
            pix_all = (np.arange(n_cap)[:, None] * 977 + np.arange(B)[None, :]) % self.n_pix
        else:
            pix_all = np.repeat((np.arange(n_cap) % self.n_pix)[:, None], B, axis=1)
        pix_all = np.ascontiguousarray(pix_all.astype(np.int32))
        _ffi.check(lib.nfa_malloc(C.byref(buf['U']), n_cap * self.step_bytes))
        _ffi.check(lib.nfa_malloc(C.byref(buf['lnL']), n_cap * B * 8))
        _ffi.check(lib.nfa_malloc(C.byref(buf['pix']), pix_all.nbytes))
        _ffi.check(lib.nfa_memcpy_h2d(buf['pix'], pix_all.ctypes.data_as(C.c_void_p), pix_all.nbytes))
        buf['cap'] = n_cap

    def reset_inputs(self, n):
        """Fresh unit-cube rows for steps 0 .. n-1 (every step gets the same B rows): one upload, then copies on the
        device that double the filled part."""
        _ffi, lib, buf, step_bytes = self._ffi, self.lib, self.buf, self.step_bytes
        assert n <= buf['cap']
        _ffi.check(lib.nfa_memcpy_h2d(buf['U'], self.U_host.ctypes.data_as(C.c_void_p), step_bytes))
        have = 1
        while have < n:
            m = min(have, n - have)
            _ffi.check(lib.nfa_memcpy_d2d(C.c_void_p(buf['U'].value + have * step_bytes), buf['U'], m * step_bytes))
            have += m
        # the copies run on the default stream, the runner's lanes do not wait for it: nothing may start (and transform
        # rows in place) before every row holds its unit-cube values
        _ffi.check(lib.nfa_device_synchronize())

    def step(self, handle, k):
        # consecutive steps may overlap on the device (stream lanes): no buffer is shared
        buf, B = self.buf, self.B
        self._ffi.check(self.lib.nfa_runner_loglike_batch_dev(handle, C.c_void_p(buf['pix'].value + k * B * 4),
                                                              C.c_void_p(buf['U'].value + k * self.step_bytes),
                                                              C.c_void_p(buf['lnL'].value + k * B * 8), B))

    def sync(self, handle):
        self._ffi.check(self.lib.nfa_runner_synchronize(handle))
        self._ffi.check(self.lib.nfa_device_synchronize())

    def timed_blocks(self, handle, n_blocks, repeats, step=None):
        """[seconds of each block]: `repeats` x --steps steps between barrier + synchronise, max (and min) over ranks."""
        args, comm = self.args, self.comm
        step = step or self.step
        out = []
        n_timed = args.steps * repeats
        for _ in range(n_blocks):
            self.reset_inputs(args.warmup + n_timed)
            for k in range(args.warmup):
                step(handle, k)
            self.sync(handle)
            comm.barrier()
            t0 = time.perf_counter()
            for k in range(args.warmup, args.warmup + n_timed):
                step(handle, k)
            t_enq = time.perf_counter() - t0         # host time to enqueue the steps (diagnostic, NFA_BENCH_HOST=1)
            self.sync(handle)
            out.append(time.perf_counter() - t0)     # this rank's steps; the block is the slowest rank's
            if os.environ.get('NFA_BENCH_HOST') and self.rank == 0:
                print(f'host enqueue {t_enq / n_timed * 1e6:.1f} us/step of {out[-1] / n_timed * 1e6:.1f}', file=sys.stderr)
            comm.barrier()
        mine = np.array(out)
        return comm.allreduce(mine, 'max'), comm.allreduce(mine, 'min')

    def repeats_for(self, handle, step=None):
        """How often the K-step sequence is repeated inside a timed block so that the block lasts MIN_BLOCK_S: from
        two probe blocks of K steps (the slowest rank's time, so that every rank repeats alike)."""
        args = self.args
        t, _ = self.timed_blocks(handle, 2, 1, step)
        r = int(np.ceil(MIN_BLOCK_S / float(t.min())))       # (a lone K-step block pays fill and drain: the repeated block runs faster per step, hence the margin in MIN_BLOCK_S)
        return max(1, min(r, MAX_BLOCK_STEPS // args.steps if args.steps <= MAX_BLOCK_STEPS else 1))

    def one_lane_kernel_times(self, spl=1, launch=None):
        """(lnl_kernel us, set-up kernel us, launches) on a one-lane runner: launches do not overlap, a
        HIP-event interval is the execution time (what rocprofv3 shows for such a launch).  `spl` steps per
        launch: the launch the engine makes of `spl` coalesced steps, here as one call over their (contiguous) rows."""
        from nestfit_amd._model import _RunnerHandle
        _ffi, lib, buf, B, args = self._ffi, self.lib, self.buf, self.B, self.args
        _ffi.set_option('streams', 1)
        solo = _RunnerHandle(self.cube._ss, self.ut, self.ncomp, False, False)
        _ffi.set_option('streams', args.streams if args.streams else 0)
        # launches behind an idle gap (the copies and the synchronisation below) run at the clocks the chip idled at: the
        # first two thirds of the sequence only bring it back to the state of a timed block and are left out of the average
        n_have = min(240, buf['cap'])
        n = max(1, n_have // spl)
        n_skip = (2 * n) // 3 if n >= 12 else 0

        def plain(handle, k):
            _ffi.check(lib.nfa_runner_loglike_batch_dev(handle, C.c_void_p(buf['pix'].value + k * spl * B * 4),
                                                        C.c_void_p(buf['U'].value + k * spl * self.step_bytes),
                                                        C.c_void_p(buf['lnL'].value + k * spl * B * 8), spl * B))
        launch = launch or plain
        self.reset_inputs(n_have)
        for k in range(min(5, n)):
            launch(solo.handle, k)
        self.sync(solo.handle)
        self.reset_inputs(n_have)
        _ffi.check(lib.nfa_runner_set_profiling(solo.handle, 1))
        _ffi.set_option('profile_skip', n_skip)
        for k in range(n):
            launch(solo.handle, k)
        self.sync(solo.handle)
        sp = (C.c_double * 4)(0, 0, 0, 0)
        sc = C.c_int64(0)
        _ffi.check(lib.nfa_runner_get_profile(solo.handle, sp, C.byref(sc)))
        _ffi.set_option('profile_skip', 0)
        _ffi.check(lib.nfa_runner_set_profiling(solo.handle, 0))
        return sp[1] / sc.value * 1e3, sp[0] / sc.value * 1e3, int(sc.value)

    def run_modes(self, na, modes, head_mode, n_blocks_head, n_blocks_other, extras=True):
        """{mode: entry}: the timed blocks of every mode, the one-lane kernel times behind them."""
        args, world, B = self.args, self.world, self.B
        _ffi = self._ffi
        per_mode = {}
        # buffers for the longest block any mode will run: sized from a probe in the fastest mode of the run
        self.alloc_inputs(args.warmup + args.steps)
        na.set_exp_mode('fast' if 'fast' in modes else modes[0])
        r_cap = self.repeats_for(self.rh) + 1
        self.alloc_inputs(args.warmup + max(args.steps * r_cap, min(60, args.warmup + args.steps)))
        for mode in modes:
            na.set_exp_mode(mode)
            n_blocks = n_blocks_head if mode == head_mode else n_blocks_other
            repeats = min(r_cap, self.repeats_for(self.rh))
            steps_block = args.steps * repeats
            evals_per_block = steps_block * B * world
            t, t_fastest = self.timed_blocks(self.rh, n_blocks, repeats)
            med = float(np.median(t))
            entry = {'value': evals_per_block / med, 'ms_per_step': med / steps_block * 1e3, 'blocks': int(n_blocks),
                     'repeats_per_block': repeats, 'block_ms': med * 1e3,
                     'min': evals_per_block / float(t.max()), 'max': evals_per_block / float(t.min()), 'dtype': DTYPES[mode]}
            if world > 1:
                # the slowest and the fastest rank of the median block: their ratio - 1 is the imbalance of the stripes
                k_med = int(np.argsort(t)[len(t) // 2])
                entry['rank_ms_per_step'] = {'slowest': float(t[k_med]) / steps_block * 1e3,
                                             'fastest': float(t_fastest[k_med]) / steps_block * 1e3}
            if extras and mode == head_mode and repeats > 1:
                # blocks of exactly K steps, nothing repeated: what the fill and drain of the launch pipeline cost a short block
                tk, _ = self.timed_blocks(self.rh, max(3, n_blocks // 3), 1)
                entry['k_steps_alone'] = {'value': args.steps * B * world / float(np.median(tk)),
                                          'ms_per_step': float(np.median(tk)) / args.steps * 1e3, 'block_ms': float(np.median(tk)) * 1e3}
            if extras and mode == head_mode and self.steps_per_launch > 1 and not args.skip_single_step:
                # the same blocks with every step launched on its own
                _ffi.set_option('coalesce', 1)
                ta, _ = self.timed_blocks(self.rh, max(3, n_blocks // 3), repeats)
                _ffi.set_option('coalesce', self.group)
                entry['one_step_per_launch'] = {'value': evals_per_block / float(np.median(ta)),
                                                'ms_per_step': float(np.median(ta)) / steps_block * 1e3}
            if self.rank == 0 and world == 1 and not self.per_row:
                # the launch as the engine makes it (steps_per_launch steps together), alone on one lane ...
                spl = self.steps_per_launch
                lnl_us, setup_us, n_l = self.one_lane_kernel_times(spl)
                entry.update({'lnl_kernel_us': lnl_us, 'setup_kernel_us': setup_us, 'one_lane_launches': n_l,
                              'evals_per_launch': B * spl,
                              'roofline_frac': self.bytes_eval * B * spl / (lnl_us * 1e-6) / 1e9 / HBM_PEAK_GBS})
                if extras and spl > 1 and not args.skip_single_step and mode == head_mode:   # ... and the launch of a single step
                    l1, s1, n1 = self.one_lane_kernel_times(1)
                    entry['single_step_launch'] = {'lnl_kernel_us': l1, 'setup_kernel_us': s1, 'one_lane_launches': n1,
                                                   'roofline_frac': self.bytes_eval * B / (l1 * 1e-6) / 1e9 / HBM_PEAK_GBS}
            per_mode[mode] = entry
        return per_mode

    def run_spectra_out(self, na, mode, n_blocks):
        """The spectra-out mode (the one genuinely HBM-bound form of the path): B rows of physical parameters
        -> B x chan_tot model spectra written to HBM through nfa_runner_predict_batch_dev -- what
        deblend_hf_intensity / generate_predicted_profiles ask of `runner.predict` once per (pixel, component)
        (nestfit/main.py:1106-1113, 1182-1188).  Algorithmic bytes per evaluation = the likelihood's + sum_s N_s * 8."""
        args, B, lib, _ffi = self.args, self.B, self.lib, self._ffi
        chan_tot = len(self.trans) * self.n_chan
        spl = self.steps_per_launch                           # consecutive predict calls of one shape travel as one launch, like the likelihood's
        n_out = 4 * max(spl, 2)                                # output buffers in rotation: every batch of the launches in flight its own
        out = C.c_void_p()
        _ffi.check(lib.nfa_malloc(C.byref(out), n_out * B * chan_tot * 8))
        theta = self.U_host.copy()
        self.ut.transform_batch(theta, self.ncomp)             # physical parameters of the same draws
        theta = np.ascontiguousarray(np.tile(theta, (spl, 1)))  # (spl copies: the one-lane measurement launches spl steps as one call)
        d_theta = C.c_void_p()
        _ffi.check(lib.nfa_malloc(C.byref(d_theta), theta.nbytes))
        _ffi.check(lib.nfa_memcpy_h2d(d_theta, theta.ctypes.data_as(C.c_void_p), theta.nbytes))
        buf = self.buf

        def step(handle, k):
            _ffi.check(lib.nfa_runner_predict_batch_dev(handle, C.c_void_p(buf['pix'].value + k * B * 4), d_theta, B,
                                                        C.c_void_p(out.value + (k % n_out) * B * chan_tot * 8),
                                                        C.c_void_p(buf['lnL'].value + k * B * 8)))
        na.set_exp_mode(mode)
        repeats = self.repeats_for(self.rh, step)
        repeats = max(1, min(repeats, (buf['cap'] - args.warmup) // args.steps))
        steps_block = args.steps * repeats
        t, _ = self.timed_blocks(self.rh, n_blocks, repeats, step)
        med = float(np.median(t))
        bytes_eval = self.bytes_eval + chan_tot * 8
        entry = {'mode': mode, 'dtype': DTYPES[mode], 'value': steps_block * B * self.world / med, 'unit': 'evals/s',
                 'ms_per_step': med / steps_block * 1e3, 'blocks': int(n_blocks), 'repeats_per_block': repeats,
                 'algorithmic_bytes_per_eval': bytes_eval, 'bytes_written_per_step': B * chan_tot * 8,
                 'entry_point': 'nfa_runner_predict_batch_dev (theta, spectra and lnL in HBM)'}
        pipe = bytes_eval * B * self.world / (entry['ms_per_step'] * 1e-3) / 1e9
        entry['pipeline_achieved_GBs'] = pipe
        entry['pipeline_frac'] = pipe / (HBM_PEAK_GBS * self.world)
        if self.rank == 0 and self.world == 1:
            def launch(handle, k):                              # the launch the engine makes of spl steps, as one call over their rows
                _ffi.check(lib.nfa_runner_predict_batch_dev(handle, C.c_void_p(buf['pix'].value + k * spl * B * 4), d_theta, spl * B,
                                                            C.c_void_p(out.value + (k % (n_out // spl)) * spl * B * chan_tot * 8),
                                                            C.c_void_p(buf['lnL'].value + k * spl * B * 8)))
            lnl_us, setup_us, n_l = self.one_lane_kernel_times(spl, launch)
            ach = bytes_eval * B * spl / (lnl_us * 1e-6) / 1e9
            entry.update({'lnl_kernel_us': lnl_us, 'setup_kernel_us': setup_us, 'one_lane_launches': n_l, 'evals_per_launch': B * spl,
                          'steps_per_launch': spl, 'achieved_GBs': ach, 'frac': ach / HBM_PEAK_GBS,
                          'kernel': f'lnl_kernel<{0 if mode == "table" else 2}, true, false, {self.ncomp}>'})
            if spl > 1:
                l1, s1, n1 = self.one_lane_kernel_times(1, lambda h, k: step(h, k))
                entry['single_step_launch'] = {'lnl_kernel_us': l1, 'setup_kernel_us': s1, 'one_lane_launches': n1,
                                               'frac': bytes_eval * B / (l1 * 1e-6) / 1e9 / HBM_PEAK_GBS}
        lib.nfa_free(out)
        lib.nfa_free(d_theta)
        return entry


def rocprof_kernel_us(csv_name, kernel_prefix):
    """(average us, path) of the kernel whose name starts with `kernel_prefix` in a committed rocprofv3 --stats summary."""
    import csv
    f = profile_file(csv_name)
    if f is None:
        return None, None
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row['Name'].startswith(kernel_prefix):
                return float(row['AverageNs']) * 1e-3, f
    return None, f


def main():
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: what RCCL needs across processes on this driver
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--blocks', type=int, default=121, help='timed blocks of --steps steps (value = the median block)')
    ap.add_argument('--workload', default='C2', choices=sorted(WORKLOADS) + ['C5', 'C5r4'],
                    help='C2 (default) is the headline metric; C5 = nested sampling of a 32x32 cube (not a "step" bench)')
    ap.add_argument('--side', type=int, default=128, help='pixels per side of the synthetic cube (C3: 128)')
    ap.add_argument('--batch', type=int, default=0, help='rows per step (default: workload B)')
    ap.add_argument('--pixels-per-step', default='1', choices=['1', 'B'],
                    help="1: the B rows of a step share one pixel (C2, the metric); B: every row has its own pixel "
                         "(one evaluation per pixel: the shape whose data really stream from HBM)")
    ap.add_argument('--exp-mode', default=os.environ.get('NFA_EXP_MODE', REFERENCE_PRECISION_MODE), choices=list(MODES),
                    help='the mode `value` is quoted in (both are timed)')
    ap.add_argument('--modes', default='all', choices=['all', 'one'], help='time both numerical modes or only --exp-mode')
    ap.add_argument('--spectra-out', default='auto', choices=['auto', 'on', 'off', 'only'],
                    help='also time the spectra-out mode (nfa_runner_predict_batch_dev; auto: with the default C2 run on one GPU); '
                         'only: nothing else (profiler runs)')
    ap.add_argument('--configs', default='auto', choices=['auto', 'on', 'off'],
                    help='also time a short C4 block in both modes (auto: with the default C2 run on one GPU)')
    ap.add_argument('--wpb', type=int, default=0, help='engine A/B knob: waves per workgroup (0 = default)')
    ap.add_argument('--wpb-table', type=int, default=0, help='engine A/B knob: waves per workgroup in table mode (0 = chosen per spectra set)')
    ap.add_argument('--lnl-cap', type=int, default=-1, help='engine A/B knob: likelihood workgroups per CU (0 = no cap)')
    ap.add_argument('--lnl-split', type=int, default=-1, help='engine A/B knob: waves per (item, spectrum) unit (0 = by launch size)')
    ap.add_argument('--lnl-queue-wg', type=int, default=0, help='engine A/B knob (table mode): workgroups per CU of a queue launch')
    ap.add_argument('--lnl-queue', type=int, default=-1, help='engine A/B knob (table mode): 0 = one unit per wave always, 1 = large launches draw their units from a queue')
    ap.add_argument('--streams', type=int, default=0, help='engine A/B knob: stream lanes (0 = default)')
    ap.add_argument('--coalesce', type=int, default=0, help='engine A/B knob: device-pointer batches launched together at most (1 = none; 0 = default 8)')
    ap.add_argument('--c5-ncomp', type=int, default=0, help='C5: only this number of components (A/B runs)')
    ap.add_argument('--sampler-walkers', type=int, default=0, help='engine A/B knob (C5): walkers per pixel of a walk cycle (64, 128, 192, 256; 0 = by the live points)')
    ap.add_argument('--sampler-ellipsoids', type=int, default=0, help='engine A/B knob (C5): 1 = one bounding ellipsoid per pixel whatever the dimension')
    ap.add_argument('--sampler-walk-factor', type=int, default=0, help='engine A/B knob (C5): to walks below an acceptance of 1 / (factor n_steps)')
    ap.add_argument('--sampler-refit-every', type=int, default=0, help='engine A/B knob (C5): rounds between the refits of rejection-mode pixels')
    ap.add_argument('--sampler-batch-target', type=int, default=0, help='A/B knob (C5): candidates per round of all pixels together')
    ap.add_argument('--sampler-upd-frac', type=float, default=0.0, help='A/B knob (C5): replacements (fraction of nlive) between refits')
    ap.add_argument('--sampler-parts', type=int, default=0, help='engine A/B knob (C5): groups of pixels pipelined over the lanes')
    ap.add_argument('--prior-stage', type=int, default=-1, help='engine A/B knob: prior tables staged in LDS (1) or left in global memory (0)')
    ap.add_argument('--setup-ti', type=int, default=0, help='engine A/B knob: items per set-up workgroup (0 = default)')
    ap.add_argument('--setup-sub', type=int, default=0, help='engine A/B knob: 1 = one group of items per set-up workgroup in the table mode (default: two)')
    ap.add_argument('--setup-threads', type=int, default=0, help='engine A/B knob: threads per set-up workgroup (0 = default)')
    ap.add_argument('--ablate', type=int, default=0, help='timing experiment with the -DNFA_ABLATE build (INVALID results)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--skip-single-step', action='store_true',
                    help='do not also time the launch of a single step (profiler runs: one launch shape per kernel name)')
    args = ap.parse_args()

    if args.workload == 'C5':
        return bench_c5(args)
    if args.workload == 'C5r4':
        return bench_c5_round4_shape(args)

    if args.gpus > 1 and 'RANK' not in os.environ:
        relaunch_one_rank_per_gpu(args)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU '
                         f'(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)')
    same_gpu = bool(os.environ.get('NFA_BENCH_SAME_GPU'))     # rehearsal of the N > 1 path on a one-GPU box

    import nestfit_amd as na
    from nestfit_amd import _ffi, comm as nfcomm
    na.set_device(0 if same_gpu else local_rank)      # one process per GPU, before any other call
    if args.ablate and 'NFA_ENGINE_LIB' not in os.environ:
        raise SystemExit('--ablate needs the test library: NFA_ENGINE_LIB=nestfit_amd/lib/libnestfit_amd_test.so')
    for key, val in (('wpb', args.wpb), ('streams', args.streams), ('ablate', args.ablate), ('setup_ti', args.setup_ti), ('coalesce', args.coalesce),
                     ('setup_threads', args.setup_threads), ('wpb_table', args.wpb_table), ('setup_sub', args.setup_sub)):
        if val:
            _ffi.set_option(key, val)
    if args.lnl_cap >= 0:
        _ffi.set_option('lnl_cap', args.lnl_cap)
    if args.lnl_split >= 0:
        _ffi.set_option('lnl_split', args.lnl_split)
    if args.lnl_queue >= 0:
        _ffi.set_option('lnl_queue', args.lnl_queue)
    if args.lnl_queue_wg > 0:
        _ffi.set_option('lnl_queue_wg', args.lnl_queue_wg)
    if args.prior_stage >= 0:
        _ffi.set_option('prior_stage', args.prior_stage)
    lib = _ffi.engine()
    if world == 1:
        comm, comm_kind = nfcomm.SoloComm(), 'solo'
    elif same_gpu:
        comm, comm_kind = nfcomm.TcpComm.from_env(), 'tcp'
    else:
        comm, comm_kind = nfcomm.comm_from_env()      # RCCL over xGMI through the C ABI (sockets if RCCL cannot start)

    # ---- everything that runs on the GPU first (the driver samples the device's activity while the command runs);
    #      the CPU baseline, a quarter of a minute of host work, last
    M = Measure(args, na, lib, comm, rank, world, args.workload, args.side, args.batch)
    B, ncomp, ndim, trans, n_chan = M.B, M.ncomp, M.ndim, M.trans, M.n_chan
    per_row, steps_per_launch, bytes_eval = M.per_row, M.steps_per_launch, M.bytes_eval
    default_run = args.workload == 'C2' and world == 1 and not per_row and B == 4096 and not args.ablate
    modes = list(MODES) if args.modes == 'all' else [args.exp_mode]
    n_other = max(3, args.blocks // 2)
    spectra = None
    if args.spectra_out == 'only':
        M.alloc_inputs(args.warmup + max(args.steps * 8, 60))
        spectra = M.run_spectra_out(na, args.exp_mode, args.blocks)
        if rank == 0:
            print(json.dumps({'metric': 'spectra-out evals/sec (predict_batch), 1024-ch 2-comp NH3(1,1)+(2,2)', **spectra}), flush=True)
        M.close()
        comm.barrier()
        comm.close()
        return
    per_mode = M.run_modes(na, modes, args.exp_mode, args.blocks, n_other)

    # results of the last step, for the end-of-run gather and a sanity check
    buf, step_bytes, lon, lat, n_pix = M.buf, M.step_bytes, M.lon, M.lat, M.n_pix
    lnL = np.empty(B)
    n_steps = args.warmup + args.steps
    _ffi.check(lib.nfa_memcpy_d2h(lnL.ctypes.data_as(C.c_void_p),
                                  C.c_void_p(buf['lnL'].value + (n_steps - 1) * B * 8), B * 8))
    if not np.isfinite(lnL).all() and not args.ablate:
        raise SystemExit('non-finite log-likelihood in the benchmark batch')
    # ... and the timed steps were fed what they are said to be fed: theta of the last step of the first and of the
    # last repeat of the last block is the prior transform of the unit-cube rows (a step that found rows an earlier pass
    # had already turned into theta would do different, cheaper work)
    want = M.U_host.copy()
    M.ut.transform_batch(want, ncomp)
    theta = np.empty((B, ndim))
    last_repeats = per_mode[modes[-1]]['repeats_per_block']
    for k in sorted({n_steps - 1, args.warmup + args.steps * last_repeats - 1}):
        _ffi.check(lib.nfa_memcpy_d2h(theta.ctypes.data_as(C.c_void_p), C.c_void_p(buf['U'].value + k * step_bytes), step_bytes))
        if not np.allclose(theta, want, rtol=1e-9, atol=1e-9) and not args.ablate:
            raise SystemExit(f'theta of timed step {k} is not the prior transform of its unit-cube rows: stale inputs')
    # end-of-run gather of fixed-size per-pixel records: here (i_lon, i_lat, rank, best lnL
    # of the last pixel walked, evaluations made per block by this rank)
    k_last = (n_steps - 1) % n_pix
    rec = np.array([[lon[k_last], lat[k_last], rank, float(lnL.max()), float(args.steps * B)]])
    allrec = nfcomm.gather_pixel_records(rec, comm)
    assert allrec.shape == (world, 5) and (allrec[:, 0] % world == allrec[:, 2]).all()

    # the spectra-out mode (the HBM-bound form of the path) on the same stripe, in the fast mode (the form whose store
    # counts: in the table mode the same rows are arithmetic for twice as long)
    if args.spectra_out == 'on' or (args.spectra_out == 'auto' and default_run):
        spectra = M.run_spectra_out(na, 'fast', max(5, args.blocks // 8))
    spec0, ut, U_host = M.spec0, M.ut, M.U_host
    M.close()
    # BASELINE config 4 (three transitions x 2048 channels x 3 components: the LDS hyperfine-table stress), a short block
    # in both modes on a small cube of its own
    configs = {}
    if args.configs == 'on' or (args.configs == 'auto' and default_run):
        M4 = Measure(args, na, lib, comm, rank, world, 'C4', 32)
        pm4 = M4.run_modes(na, list(MODES), None, 0, max(5, args.blocks // 6), extras=False)
        configs['C4'] = {
            'workload': f'C4: B={M4.B} draws per step against one pixel, NH3 (1,1)+(2,2)+(3,3), {M4.n_chan} ch, {M4.ncomp} comp; 32x32 cube',
            'algorithmic_bytes_per_eval': M4.bytes_eval, 'steps_per_launch': M4.steps_per_launch,
            **{m: {k: e.get(k) for k in ('value', 'ms_per_step', 'blocks', 'repeats_per_block', 'lnl_kernel_us', 'evals_per_launch',
                                          'roofline_frac', 'min', 'max', 'dtype')} for m, e in pm4.items()}}
        M4.close()
        # BASELINE config 5 as specified (the cube driver's component loop on the device sampler, a store written), in the
        # fast mode (`python bench.py --workload C5` runs it in both)
        if rank == 0 and world == 1:
            c5_specified(na, 'fast', side=8, n=256)             # (code objects, streams and buffers come up outside the clock)
            configs['C5'] = {'workload': C5_WORKLOAD, **c5_specified(na, 'fast')}

    # which device every rank computes on (two ranks on one GPU, or a silent fallback, show on the line)
    ubuf = C.create_string_buffer(40)
    _ffi.check(lib.nfa_device_uuid(ubuf, 40))
    mine = np.frombuffer(bytes.fromhex(ubuf.value.decode()), dtype=np.uint8).astype(np.float64)
    uuids = []
    for row in comm.allgather(mine).reshape(world, 16):
        raw = bytes(row.astype(np.uint8))
        # HIP reports the UUID as 16 ASCII hex digits on this stack: show those; raw bytes otherwise
        uuids.append(raw.decode('ascii') if all(32 < b < 127 for b in raw) else raw.hex())

    if rank == 0:
        head = per_mode[args.exp_mode]
        value = head['value']
        step_s = head['ms_per_step'] * 1e-3
        # (key order: what reads this line keeps the first twenty scalars of `roofline` -- the headline mode's figures
        # first, then the other mode's as other_*)
        other = OTHER_MODE[args.exp_mode]
        po = per_mode.get(other)
        roof = {'bound': 'valu', 'achieved': None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': None, 'traffic': None,
                'kernel': LNL_KERNEL_NAME[args.exp_mode] if ncomp == 2 else 'lnl_kernel', 'mode': args.exp_mode,
                'algorithmic_bytes_per_eval': bytes_eval, 'evals_per_launch': B * steps_per_launch,
                'steps_per_launch': steps_per_launch, 'avg_launch_us': None, 'rocprof_avg_launch_us': None, 'rocprof_frac': None,
                'pipeline_frac': None}
        if po:
            roof.update({f'{other}_value': po['value'], f'{other}_frac': po.get('roofline_frac'), f'{other}_ms_per_step': po['ms_per_step'],
                         f'{other}_avg_launch_us': po.get('lnl_kernel_us')})
        if 'lnl_kernel_us' in head:
            ach = bytes_eval * B * steps_per_launch / (head['lnl_kernel_us'] * 1e-6) / 1e9
            roof.update({'achieved': ach, 'frac': ach / HBM_PEAK_GBS, 'avg_launch_us': head['lnl_kernel_us'],
                         'setup_kernel_us': head['setup_kernel_us']})
            if 'single_step_launch' in head:
                roof['single_step_launch'] = head['single_step_launch']
        pipe = bytes_eval * B * world / step_s / 1e9
        roof['pipeline'] = {'us_per_step': step_s * 1e6, 'achieved': pipe, 'frac': pipe / (HBM_PEAK_GBS * world),
                            'note': 'the same bytes / the time per step of the timed blocks (the engine launches '
                                    f'{steps_per_launch} step(s) together and overlaps launches on its stream lanes): '
                                    'the rate the job sustains, a hard bound on the kernel'}
        roof['pipeline_frac'] = roof['pipeline']['frac']
        # figures that need their own profiler passes come from the committed summaries of the latest round that has them
        def rel(path):
            return str(path.relative_to(ROOT))
        try:
            f = profile_file(f'pmc_lnl_{args.exp_mode}.json')
            pmc = json.loads(f.read_text())
            roof['valu'] = {'busy_frac_one_lane': pmc.get('valu_busy_frac'),
                            'valu_instructions_per_eval': pmc['instructions_per_eval'].get('valu'),
                            'salu_instructions_per_eval': pmc['instructions_per_eval'].get('salu'),
                            'source': f'{rel(f)} (rocprofv3 --pmc passes of the one-lane command, profiles/collect_round.sh pmc)'}
            roof['valu_busy_frac'] = pmc.get('valu_busy_frac')
            roof['valu_instructions_per_eval'] = pmc['instructions_per_eval'].get('valu')
        except Exception:
            pass
        try:
            f = profile_file('pmc_traffic.json')
            tr = json.loads(f.read_text())
            key = 'pixel_per_item' if per_row else 'one_pixel'
            # measured per 4096-row batch; a launch of coalesced steps moves that once per step
            roof['traffic'] = tr['detail'][args.exp_mode][key]['total_bytes'] * steps_per_launch if args.workload == 'C2' and B == 4096 else None
            roof['traffic_source'] = f'{rel(f)} (FETCH_SIZE x calibration + WRITE_SIZE, separate passes, per 4096-row batch) x steps_per_launch'
        except Exception:
            pass
        if args.workload == 'C2' and B == 4096:
            csv_name = 'onelane_kernel_stats.csv' if args.exp_mode == 'fast' else f'onelane_kernel_stats_{args.exp_mode}.csv'
            us, f = rocprof_kernel_us(csv_name, LNL_KERNEL_NAME[args.exp_mode])
            if us:
                roof['rocprof_avg_launch_us'] = us
                roof['rocprof_frac'] = bytes_eval * B * steps_per_launch / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
                roof['rocprof'] = (f'{rel(f)} (rocprofv3 --kernel-trace --stats -- python bench.py --streams 1 --modes one '
                                   f'--exp-mode {args.exp_mode} --no-cpu-baseline --skip-single-step --spectra-out off --configs off)')
        # the like-for-like number: the same workload in the reference's own arithmetic (f64 table product, f64 tau).
        # Flat keys beside the block: what reads this line keeps scalars of `roofline` and drops nested objects.
        rp = per_mode.get(REFERENCE_PRECISION_MODE)
        if rp:
            block = {'mode': REFERENCE_PRECISION_MODE, 'dtype': DTYPES[REFERENCE_PRECISION_MODE], 'value': rp['value'],
                     'ms_per_step': rp['ms_per_step'], 'kernel': LNL_KERNEL_NAME[REFERENCE_PRECISION_MODE] if ncomp == 2 else 'lnl_kernel<0, ...>',
                     'blocks': rp['blocks'], 'min': rp['min'], 'max': rp['max'],
                     'avg_launch_us': rp.get('lnl_kernel_us'), 'frac': rp.get('roofline_frac'),
                     'pipeline_frac': bytes_eval * B * world / (rp['ms_per_step'] * 1e-3) / 1e9 / (HBM_PEAK_GBS * world)}
            if args.workload == 'C2' and B == 4096:
                us, f = rocprof_kernel_us('onelane_kernel_stats_table.csv', LNL_KERNEL_NAME['table'])
                if us:
                    block['rocprof_avg_launch_us'] = us
                    block['rocprof_frac'] = bytes_eval * B * steps_per_launch / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
                    block['rocprof'] = rel(f)
            try:
                f = profile_file('pmc_lnl_table.json')
                pmc = json.loads(f.read_text())
                block['valu_busy_frac'] = pmc.get('valu_busy_frac')
                block['valu_instructions_per_eval'] = pmc['instructions_per_eval'].get('valu')
                block['lds_bank_conflict_frac'] = pmc.get('lds_bank_conflict_frac')
                block['pmc'] = rel(f)
            except Exception:
                pass
            roof['reference_precision'] = block
            for k, v in block.items():
                if not isinstance(v, (dict, list)):
                    roof[f'reference_precision_{k}'] = v
        if spectra:
            roof['spectra_out'] = spectra
            for k in ('value', 'ms_per_step', 'frac', 'achieved_GBs', 'lnl_kernel_us', 'algorithmic_bytes_per_eval', 'pipeline_frac'):
                if k in spectra:
                    roof[f'spectra_out_{k}'] = spectra[k]
            try:
                f = profile_file('pmc_traffic_spectra_out.json')
                tr = json.loads(f.read_text())
                spectra['traffic'] = tr['bytes_per_launch'] * spectra.get('steps_per_launch', 1)      # (measured per 4096-row batch)
                spectra['traffic_frac_of_peak'] = spectra['traffic'] / (spectra['lnl_kernel_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS
                spectra['traffic_source'] = rel(f)
                roof['spectra_out_traffic'] = spectra['traffic']
                roof['spectra_out_traffic_frac_of_peak'] = spectra['traffic_frac_of_peak']
            except Exception:
                pass
        if 'C5' in configs:
            roof['C5_seconds'] = configs['C5']['seconds']
            roof['C5_evals_per_pixel'] = configs['C5']['evals_per_pixel']
        for cname, cfg in configs.items():
            for m in MODES:
                if m in cfg and isinstance(cfg[m], dict):
                    roof[f'{cname}_{m}_value'] = cfg[m]['value']
                    roof[f'{cname}_{m}_frac'] = cfg[m].get('roofline_frac')
        roof['note'] = ('achieved / frac: algorithmic bytes (SURVEY 8d) per launch / lnl_kernel time per launch -- the launch the '
                        'engine makes of `steps_per_launch` steps that arrive back to back -- HIP events on a one-lane runner after '
                        'the timed blocks; `single_step_launch` = the same for the launch of one step; `rocprof` names the '
                        'committed summary of the one-lane command whose average agrees. The kernel is VALU bound: `valu` = busy fraction of the vector ALUs from '
                        'SQ_ACTIVE_INST_VALU and instructions per evaluation (separate --pmc passes); with --pixels-per-step 1 '
                        'the pixel stays in L2 (traffic << algorithmic bytes), see DESIGN.md.  `reference_precision` (and the flat '
                        'reference_precision_* keys): the same workload and timing in the table mode, the reference\'s own f64 arithmetic')
        cpu = None
        if args.gpus == 1 and not args.no_cpu_baseline and not per_row:
            cpu = cpu_baseline(spec0, ut.lower(), ncomp, U_host)
            # the port beside the real reference.  The reference cannot travel; its rate at this shape
            # was measured in the build container and the port was timed on the same CPU there
            # (scripts/measure_port_vs_reference.py): ratio_to_reference = port / reference, per core.
            f = profile_file('port_vs_reference.json')
            if f is not None and args.workload == 'C2':
                pv = json.loads(f.read_text())
                ratio = pv['ratio_to_reference']
                cpu['ratio_to_reference'] = ratio
                cpu['reference_equivalent'] = cpu['value'] / ratio
                cpu['ratio_source'] = f'{f.relative_to(ROOT)}: port {pv["port_evals_per_s_one_core"]:.0f} evals/s against the ' \
                                      f'reference\'s {pv["reference_evals_per_s_one_core"]:.0f} on one core of the same CPU'
        name = C.create_string_buffer(128)
        lib.nfa_device_name(name, 128)
        metric = 'loglikelihood evals/sec, 1024-ch 2-comp NH3(1,1)+(2,2); HBM GB/s vs peak'
        if per_row:
            metric += ' [one evaluation per pixel]'
        line = {
            'metric': metric, 'value': value, 'unit': 'evals/s', 'n_gpus': args.gpus, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': head['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': DTYPES[args.exp_mode], 'data': 'synthetic',
            'config': {
                'workload': f'{args.workload}: B={B} live-point draws per step per GPU against one pixel, '
                            f'NH3 {"+".join(f"({t},{t})" for t in trans)}, {n_chan} ch, {ncomp} comp, '
                            f'get_irdc_priors(size=500); pixels = rank stripe (i_lon % {world}) of the C3 cube '
                            f'{args.side}x{args.side}, one pixel per '
                            + ('ROW (one evaluation per pixel)' if per_row else 'step'),
                'exp_mode': args.exp_mode, 'stream_lanes': M.lanes_used, 'steps_per_launch': steps_per_launch, 'pixels_per_gpu': int(n_pix),
                'repeats_per_block': head['repeats_per_block'], 'block_ms': head['block_ms'],
                **({'reference_precision_mode': REFERENCE_PRECISION_MODE, 'reference_precision_value': per_mode[REFERENCE_PRECISION_MODE]['value'],
                    'reference_precision_ms_per_step': per_mode[REFERENCE_PRECISION_MODE]['ms_per_step']}
                   if REFERENCE_PRECISION_MODE in per_mode else {}),
                'comm': comm_kind, 'devices': uuids,
                **({'rccl_fallback': 'RCCL did not come up on every rank: barrier / max-time / record gather went over TCP sockets'}
                   if comm_kind == 'tcp' and not same_gpu else {}),
                **({'ranks_share_a_device': True} if len(set(uuids)) < world else {}),
                'sharding': 'pixel stripes i_lon % world (nestfit/main.py:565-571), no data-path collective; '
                            'barrier / max-time / record gather over '
                            + {'rccl': 'RCCL (nfa_comm_*)', 'tcp': 'TCP sockets (ranks share a GPU, or RCCL did not start)',
                               'solo': 'RCCL (nfa_comm_*) when N > 1'}[comm_kind],
                'device': name.value.decode(),
            },
            'spread': {'blocks': head['blocks'], 'statistic': 'median block', 'min': head['min'], 'max': head['max'],
                       **({'rank_ms_per_step': head['rank_ms_per_step']} if 'rank_ms_per_step' in head else {})},
            'modes': per_mode, 'roofline': roof, 'cpu_baseline': cpu,
        }
        if configs:
            line['configs'] = configs
        if spectra:
            line['spectra_out'] = spectra
        if cpu:
            line['speedup_vs_cpu_all_cores'] = value / cpu['value']
            if 'reference_equivalent' in cpu:       # against what the reference itself would deliver on these cores
                line['speedup_vs_reference_all_cores'] = value / cpu['reference_equivalent']
                if REFERENCE_PRECISION_MODE in per_mode:
                    line['reference_precision_speedup_vs_reference_all_cores'] = per_mode[REFERENCE_PRECISION_MODE]['value'] / cpu['reference_equivalent']
        print(json.dumps(line), flush=True)

    comm.barrier()
    stuck = getattr(comm, 'stuck_thread', None)
    comm.close()
    if stuck is not None and stuck.is_alive():     # an RCCL rendezvous that never returned: do not wait for its teardown
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
