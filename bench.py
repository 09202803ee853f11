#!/usr/bin/env python3
"""Benchmark of the hot path: log-likelihood evaluations per second at the
metric shape of BASELINE.json (config C2: B = 4096 live-point draws per pixel,
NH3 (1,1)+(2,2), 1024 channels each, 2 velocity components, get_irdc_priors).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (prior transform -> model spectra -> chi^2)
over one batch of B unit-cube rows for this rank's pixel.  Every input of the
timed region (unit-cube rows for all steps, spectra, prior tables) is resident
in HBM before the clock starts.  For N > 1 the driver launches one rank per GPU
(torch.distributed.run); pixels are striped over ranks like the reference's
get_multiproc_indices (nestfit/main.py:565-571), per-GPU work is fixed (weak
scaling) and there is no data-path collective: the only communication is the
barrier / max-time reduction around the timed region and the end-of-run gather
of per-pixel summaries.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import contextlib
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
WORKLOADS = {
    # name: (trans ids, channels, vhalf, ncomp, truth key, B)
    'C2': ((1, 2), 1024, 30.0, 2, 'TRUTH_2COMP', 4096),
    'C4': ((1, 2, 3), 2048, 40.0, 3, 'TRUTH_3COMP', 4096),
    'C1': ((1,), 256, 30.0, 1, 'TRUTH_1COMP', 4096),
}


def algorithmic_bytes(trans, n_chan, ncomp):
    """sum_s N_s*8 (data) + ndim*8 (u in) + ndim*8 (theta out) + 8 (lnL)."""
    ndim = 6 * ncomp
    return len(trans) * n_chan * 8 + 2 * ndim * 8 + 8


def make_pixel(na, trans, n_chan, vhalf, truth, noise, seed):
    """Synthetic pixel: engine model spectrum of `truth` + default_rng(seed) normal noise."""
    from nestfit_amd.synth import freq_axis
    rng = np.random.default_rng(seed)
    axes = [freq_axis(t, n_chan, vhalf) for t in trans]
    spec_data = []
    for t, x in zip(trans, axes):
        s = na.AmmoniaSpectrum(x, np.zeros(n_chan), noise, t)
        na.amm_predict(s, truth)
        spec_data.append([x, s.get_spec() + rng.normal(0, noise, n_chan), noise, t])
    return spec_data


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
    """Times the CPU oracle (same algorithm, reference compile flags) on a bounded
    sample of the same workload: 1 core, then all cores (one process per core,
    like the reference's fit_cube(nproc))."""
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
        'sample': f'{reps} x 1024 rows of the same U per core, {cores} processes '
                  f'(oracle/nf_oracle.c, -O3 -march=native -ffast-math)',
        'one_core': one_core,
    }


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on stdout when the first communicator is created; the
    contract is ONE JSON line on stdout, so native stdout goes to stderr meanwhile."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def bench_c5(args):
    """BASELINE config 5 on the built-in device sampler (one GPU): nested sampling with 400 live
    points of every pixel of a 32x32 synthetic NH3 (1,1)+(2,2) cube, with one and with two velocity
    components.  Not the headline metric: one JSON line of its own shape."""
    import nestfit_amd as na
    from nestfit_amd import _ffi, sampler
    from nestfit_amd.cube import CubeRunner
    from nestfit_amd.synth import freq_axis
    na.set_exp_mode(args.exp_mode)
    side, n, noise, nlive = 32, 512, 0.1, 400
    n_pix = side * side
    rng = np.random.default_rng(0)
    axes = [freq_axis(1, n), freq_axis(2, n)]
    ut = na.get_irdc_priors(size=500, vsys=0.0)
    lon, lat = np.indices((side, side))
    r = np.hypot(lon - side / 2, lat - side / 2) / (side / 2)
    out = {}
    for ncomp in (1, 2):
        truths = np.zeros((n_pix, 6 * ncomp))
        for c in range(ncomp):
            truths[:, c] = (-1.0 + 2.0 * lon.ravel() / side) + 1.5 * c
            truths[:, ncomp + c], truths[:, 2 * ncomp + c] = 12.0 + 3 * c, 5.0 + c
            truths[:, 3 * ncomp + c], truths[:, 4 * ncomp + c] = 14.6 - 0.6 * r.ravel(), 0.4
        probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=ncomp)
        model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
        cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut,
                          ncomp=ncomp)
        _ffi.check(_ffi.load().nfa_device_synchronize())
        t0 = time.perf_counter()
        res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=nlive, tol=0.5, efr=0.3, seed=1)
        dt = time.perf_counter() - t0
        gain = np.array([x.lnZ for x in res]) - cube.null_lnZ
        out[ncomp] = {'seconds': dt, 'pixels_per_s': n_pix / dt, 'likelihood_evals': int(sum(x.n_evals for x in res)),
                      'evals_per_s': sum(x.n_evals for x in res) / dt, 'mean_lnZ_err': float(np.mean([x.lnZ_err for x in res])),
                      'detections': int((gain > 11).sum())}
    print(json.dumps({
        'metric': 'pixels/sec, nested sampling (400 live points) of a 32x32 NH3(1,1)+(2,2) cube, 1 component',
        'value': out[1]['pixels_per_s'], 'unit': 'pixels/s', 'n_gpus': 1, 'higher_is_better': True,
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'C5: 32x32 pixels x 2 spectra x 512 channels, built-in device sampler '
                               '(libmultinest is not available), tol 0.5, efr 0.3', 'exp_mode': args.exp_mode},
        'one_component': out[1], 'two_components': out[2]}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='C2', choices=sorted(WORKLOADS) + ['C5'],
                    help='C2 (default) is the headline metric; C5 = nested sampling of a 32x32 cube (not a "step" bench)')
    ap.add_argument('--batch', type=int, default=0, help='rows per step (default: workload B)')
    ap.add_argument('--exp-mode', default=os.environ.get('NFA_EXP_MODE', 'fast'),
                    choices=['table', 'poly', 'fast'])
    ap.add_argument('--wpb', type=int, default=0, help='engine A/B knob: waves per workgroup (0 = default)')
    ap.add_argument('--lnl-cap', type=int, default=-1, help='engine A/B knob: likelihood workgroups per CU (0 = no cap)')
    ap.add_argument('--streams', type=int, default=0, help='engine A/B knob: stream lanes (0 = default)')
    ap.add_argument('--ablate', type=int, default=0, help='timing experiment: 1 skip Tb, 2 skip lines, 3 both (INVALID results)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile-events', action='store_true',
                    help='do not record per-kernel HIP events inside the timed region')
    args = ap.parse_args()

    if args.workload == 'C5':
        return bench_c5(args)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('NFA_BENCH_SAME_GPU'):      # rehearsal of the N > 1 path on a one-GPU box (gloo barrier)
        local_rank = 0
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    use_dist = world > 1 or os.environ.get('NFA_BENCH_FORCE_DIST') == '1'
    if use_dist:
        # torch ships its own HIP runtime (same soname as /opt/rocm's): it has to be the one
        # that is loaded first, otherwise torch finds "no HIP GPUs" after the engine's init
        import torch
        torch.cuda.set_device(local_rank)
        torch.cuda.init()
    import nestfit_amd as na
    from nestfit_amd import _ffi, synth
    na.set_device(local_rank)                 # one process per GPU, before any other call
    na.set_exp_mode(args.exp_mode)
    if args.wpb:
        _ffi.set_option('wpb', args.wpb)
    if args.lnl_cap >= 0:
        _ffi.set_option('lnl_cap', args.lnl_cap)
    if args.ablate:
        _ffi.set_option('ablate', args.ablate)
    if args.streams:
        _ffi.set_option('streams', args.streams)
    lib = _ffi.engine()

    dist = None
    if use_dist:                              # one rank per GPU over RCCL
        import torch.distributed as dist
        with stdout_to_stderr():
            if os.environ.get('NFA_BENCH_SAME_GPU'):
                dist.init_process_group('gloo')
            else:
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            dist.barrier()                    # creates the communicator (and prints the banner)
            torch.cuda.synchronize()

    trans, n_chan, vhalf, ncomp, truth_key, B = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    truth = getattr(synth, truth_key)
    ndim = 6 * ncomp
    noise = 0.2
    # pixel striping: rank r owns pixels i_lon with i_lon % world == r (main.py:565-571);
    # here one pixel per rank, seeded by its global index
    spec_data = make_pixel(na, trans, n_chan, vhalf, truth, noise, seed=5 + rank)
    ut = na.get_irdc_priors(size=500, vsys=0.0)
    runner = na.AmmoniaRunner.from_data(spec_data, ut, ncomp=ncomp)
    rh = runner._run.handle

    # inputs of all steps resident in HBM before the clock starts
    n_total = args.warmup + args.steps
    U_host = np.random.default_rng(7 + rank).uniform(size=(B, ndim))
    d_U = C.c_void_p()
    d_lnL = C.c_void_p()
    _ffi.check(lib.nfa_malloc(C.byref(d_U), n_total * B * ndim * 8))
    _ffi.check(lib.nfa_malloc(C.byref(d_lnL), n_total * B * 8))     # one result vector per step
    for k in range(n_total):
        _ffi.check(lib.nfa_memcpy_h2d(C.c_void_p(d_U.value + k * B * ndim * 8),
                                      U_host.ctypes.data_as(C.c_void_p), B * ndim * 8))

    def step(k):
        # consecutive steps may overlap on the device (stream lanes): no buffer is shared
        _ffi.check(lib.nfa_runner_loglike_batch_dev(rh, None, C.c_void_p(d_U.value + k * B * ndim * 8),
                                                    C.c_void_p(d_lnL.value + k * B * 8), B))

    def sync():
        _ffi.check(lib.nfa_runner_synchronize(rh))
        _ffi.check(lib.nfa_device_synchronize())
        if dist is not None:
            import torch
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    profile = not args.no_profile_events
    if profile:
        _ffi.check(lib.nfa_runner_set_profiling(rh, 1))
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(args.warmup, n_total):
        step(k)
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    prof = (C.c_double * 4)(0, 0, 0, 0)
    calls = C.c_int64(0)
    if profile:
        _ffi.check(lib.nfa_runner_get_profile(rh, prof, C.byref(calls)))
        _ffi.check(lib.nfa_runner_set_profiling(rh, 0))

    # The same kernel with the GPU to itself: a second runner with ONE stream lane, so that launches
    # do not overlap and a HIP-event interval is the kernel's execution time (what rocprofv3 reports
    # for a launch).  Outside the timed region.
    alone_us = None
    if profile and rank == 0:
        _ffi.set_option('streams', 1)
        solo = na.AmmoniaRunner.from_data(spec_data, ut, ncomp=ncomp)
        _ffi.set_option('streams', args.streams if args.streams else 4)
        sh = solo._run.handle
        n_solo = min(40, n_total)
        for k in range(5):
            _ffi.check(lib.nfa_runner_loglike_batch_dev(sh, None, C.c_void_p(d_U.value + k * B * ndim * 8),
                                                        C.c_void_p(d_lnL.value + k * B * 8), B))
        _ffi.check(lib.nfa_runner_synchronize(sh))
        # the warm-up overwrote the first U slices with theta: restore unit-cube inputs
        for k in range(n_solo):
            _ffi.check(lib.nfa_memcpy_h2d(C.c_void_p(d_U.value + k * B * ndim * 8),
                                          U_host.ctypes.data_as(C.c_void_p), B * ndim * 8))
        _ffi.check(lib.nfa_runner_set_profiling(sh, 1))
        for k in range(n_solo):
            _ffi.check(lib.nfa_runner_loglike_batch_dev(sh, None, C.c_void_p(d_U.value + k * B * ndim * 8),
                                                        C.c_void_p(d_lnL.value + (n_total - 1) * B * 8), B))
        _ffi.check(lib.nfa_runner_synchronize(sh))
        sp = (C.c_double * 4)(0, 0, 0, 0)
        sc = C.c_int64(0)
        _ffi.check(lib.nfa_runner_get_profile(sh, sp, C.byref(sc)))
        if sc.value > 0:
            alone_us = (sp[1] / sc.value * 1e3, sp[0] / sc.value * 1e3)

    # results of the last step, for the end-of-run gather and a sanity check
    lnL = np.empty(B)
    _ffi.check(lib.nfa_memcpy_d2h(lnL.ctypes.data_as(C.c_void_p),
                                  C.c_void_p(d_lnL.value + (n_total - 1) * B * 8), B * 8))
    if not np.isfinite(lnL).all():
        raise SystemExit('non-finite log-likelihood in the benchmark batch')

    t_max = elapsed
    summary = [(rank, float(lnL.max()), int(args.steps * B))]
    if dist is not None:
        import torch
        tdev = 'cpu' if os.environ.get('NFA_BENCH_SAME_GPU') else 'cuda'
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_max = float(t.item())
        # end-of-run gather of fixed-size per-pixel records
        rec = torch.tensor([float(rank), float(lnL.max()), float(args.steps * B)],
                           dtype=torch.float64, device=tdev)
        out = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(out, rec)
        summary = [(int(o[0].item()), float(o[1].item()), int(o[2].item())) for o in out]

    if rank == 0:
        total_evals = sum(s[2] for s in summary)
        value = total_evals / t_max
        bytes_eval = algorithmic_bytes(trans, n_chan, ncomp)
        roof = None
        if profile and calls.value > 0:
            n = calls.value
            raw_s = prof[1] / n / 1e3             # mean event interval (includes queueing behind other lanes)
            eff_s = prof[3] / n / 1e3             # time with >= 1 likelihood kernel running, per launch
            achieved = bytes_eval * B / eff_s / 1e9
            roof = {
                'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                'kernel': 'lnl_kernel', 'avg_launch_us': eff_s * 1e6,
                'avg_launch_us_raw_mean': raw_s * 1e6,
                'achieved_from_raw_mean': bytes_eval * B / raw_s / 1e9,
                'algorithmic_bytes_per_eval': bytes_eval, 'evals_per_launch': B,
                'setup_kernel_avg_us': prof[2] / n * 1e3, 'setup_kernel_avg_us_raw_mean': prof[0] / n * 1e3,
                'alone_launch_us': None if alone_us is None else alone_us[0],
                'alone_setup_us': None if alone_us is None else alone_us[1],
                'achieved_alone': None if alone_us is None else bytes_eval * B / (alone_us[0] * 1e-6) / 1e9,
                'note': 'algorithmic bytes (SURVEY 8d) per launch / likelihood-kernel time per launch '
                        '(lnl_kernel + lnl_sum_kernel). Consecutive steps run on different HIP streams and '
                        'overlap, so over the timed region the time per launch is the union of the launch '
                        'intervals / launches (HIP events on the launch streams); *_raw_mean is the plain '
                        'mean of those intervals, which also contains the time a launch waits behind the '
                        'other lanes. alone_launch_us is the same kernel on a one-lane runner after the '
                        'timed region (no overlap: the execution time rocprofv3 shows for such a launch). '
                        'The kernel is VALU bound, the pixel data stay in L2 (traffic << algorithmic '
                        'bytes), see DESIGN.md',
            }
            tfile = ROOT / 'profiles' / 'pmc_traffic.json'
            if tfile.exists():
                try:
                    roof['traffic'] = json.loads(tfile.read_text()).get(args.workload, {}).get(
                        args.exp_mode)
                except Exception:
                    pass
        cpu = None
        if args.gpus == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(spec_data, ut.lower(), ncomp, U_host)
        name = C.create_string_buffer(128)
        lib.nfa_device_name(name, 128)
        line = {
            'metric': 'loglikelihood evals/sec, 1024-ch 2-comp NH3(1,1)+(2,2); HBM GB/s vs peak',
            'value': value, 'unit': 'evals/s', 'n_gpus': args.gpus, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': t_max / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': f'{args.workload}: B={B} live-point draws per step per GPU, '
                            f'NH3 {"+".join(f"({t},{t})" for t in trans)}, {n_chan} ch, '
                            f'{ncomp} comp, get_irdc_priors(size=500)',
                'exp_mode': args.exp_mode, 'stream_lanes': args.streams or 4, 'pixels_per_gpu': 1, 'sharding': 'pixel stripes, no collective',
                'device': name.value.decode(),
            },
            'roofline': roof, 'cpu_baseline': cpu,
        }
        if cpu:
            line['speedup_vs_cpu_all_cores'] = value / cpu['value']
        print(json.dumps(line), flush=True)

    lib.nfa_free(d_U)
    lib.nfa_free(d_lnL)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
