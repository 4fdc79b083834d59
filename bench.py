#!/usr/bin/env python3
"""Headline benchmark: grids/s of one full training step (forward + loss + backward + Adam [+ RCCL gradient all-reduce]) of
Homogeneous_Poisson_NN_Legacy(hpnn.json) on synthetic data resident in HBM.

  python bench.py --gpus N --steps K --warmup W [--workload c4|c3|small]
    c4 (default): 8 x 1024^2 Dirichlet grids per GPU, data-parallel weak scaling (BASELINE.json configs[3])
    c3          : 32 x 512^2 grids per GPU (BASELINE.json configs[2])

Launching.  N > 1 works both ways: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK / WORLD_SIZE
in the environment: this process IS one rank), or bare (`python bench.py --gpus N`): the parent then starts N fresh child processes - one
rank per GPU, rendezvous on 127.0.0.1 - BEFORE anything touches the GPU (no exec of a GPU-initialised process), relays rank 0's JSON line
and exits with the worst child status.

What is timed.  W warm-up + K timed steps, the timed region bracketed by barrier + synchronize, MAX over ranks:
  * `value` / `ms_per_step` / `dtype` / `roofline`: math mode "fp32" - exact fp32 products and sums, the library default and the
    reference's precision (experiments/hpnn.json:78).  This is ALL the default run times;
  * `--modes split_f16,fp32` adds a `split_f16` block: the opt-in 3 x fp16 split mode, timed the same way in the same
    run, with `accuracy_vs_fp32` (forward output and flat gradient of THIS benchmark batch in split mode against fp32 mode).  Since round 4 the
    mode is no faster than fp32 on the shipped model, so the default run no longer spends a third of its GPU time on it.
`roofline` (bound "hbm") describes ALL convolution launches of the timed steps - forward, data gradient, weight gradient.  `achieved` = the
ALGORITHMIC bytes of SURVEY 8(d) (every layer's input + output + filter, once per pass) / the summed duration of those launches in THIS run (HIP
events on the launch stream); `frac` = achieved / 8 TB/s.  Beside it the EXECUTED view: `traffic` = HBM bytes the convolution kernels of one step
really move (2 x FETCH_SIZE + WRITE_SIZE from separate `rocprofv3 --pmc` passes, committed as profiles/<round>_<workload>_pmc_summary_<mode>.json by
tools/collect_pmc.sh, used only when the summary's source stamp matches the kernel sources of this tree), `executed_frac` = traffic / the same time
/ 8 TB/s, `traffic_over_algorithmic` the waste factor (tile spectra written and re-read); `mfma_busy` the matrix-pipe busy fraction of the same
kernels; `kernels` (detail file) the per-kernel table of the stamped summary.  `direct_conv_equivalent` is the direct convolution's FLOP over the
same time - a speed-up measure, not a roofline fraction.  `roofline.hbm_bound` is the north star's "conv forward vs HBM roofline" figure: the conv
launches whose arithmetic intensity is below the fp32 ridge (3x3 tail, <= 8 channels), algorithmic bytes / time, with the transposed-convolution
and resize forward kernels beside it.
The default c4 run appends a `c3` block (32 x 512^2, fp32: ms/step, grids/s, the same roofline figures) so that both halves of BASELINE.json's
metric are timed by one command (--no-c3 skips it; at --gpus N > 1 only with --c3: one number per lease); `rank_ms_per_step` lists every rank's own step time, `collective_ms` the gradient
all-reduce timed on its own (N > 1).
`cpu_baseline` times the oracle's torch-CPU twin of the same graph (fp32, host cores) on one 512^2 grid - a stand-in for the reference's
TF-CPU path, which cannot run here.  `dataset` is the on-device 512^2 FD reference-solution generator next to its scipy stand-in.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_SPLIT_TFLOPS = 2500.0 / 3  # 3 fp16 MFMA FLOP per algorithmic fp32 FLOP at the ~2.5 PFLOP/s dense fp16 peak
PEAK_FP64_MFMA_TFLOPS = 78.6    # MI355X data-sheet fp64 matrix rate (v_mfma_f64_16x16x4_f64: 256 FLOP/clk/CU at 2.4 GHz, 256 CUs, half the MI300X rate)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s is what a float4 copy achieves)
WORKLOADS = {'c4': (8, 1024), 'c3': (32, 512), 'small': (2, 256), 'launch-check': (0, 0)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='c4', choices=sorted(WORKLOADS),
                    help="launch-check: no model - rendezvous, barrier, all-reduce of the 22.2 MB gradient bucket and the JSON line only "
                         "(runs on CPU/gloo; exercises the N > 1 launch path without a GPU)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-dataset', action='store_true')
    ap.add_argument('--no-inference', action='store_true', help='skip the forward-only block (a few seconds)')
    ap.add_argument('--no-c3', action='store_true', help='skip the 32 x 512^2 block that the default c4 run appends')
    ap.add_argument('--modes', default='fp32', help='math modes to time, in order; the LAST one is the headline.  Default: fp32 only (the library default); '
                                                    '"split_f16,fp32" also times the opt-in split mode with its accuracy gate (a third more GPU time)')
    ap.add_argument('--c3', action='store_true', help='append the 32 x 512^2 block also at --gpus N > 1 (default there: headline workload only - one number per lease)')
    ap.add_argument('--math', default=None, choices=['split_f16', 'fp32'], help='time only this mode (profiling runs)')
    ap.add_argument('--cpu-baseline-hw', type=int, default=512)
    ap.add_argument('--no-cpu-baseline-1024', action='store_true', help='skip the 1024^2 leg of the CPU baseline (1 warm-up + 2 reps, about a minute)')
    ap.add_argument('--launch-timeout', type=float, default=3000.0, help='bare N > 1 launch: seconds after which the parent stops every rank and exits 124')
    ap.add_argument('--timeout', type=float, default=600.0, help='seconds a rank waits in a rendezvous / barrier / collective before it gives up (process-group timeout)')
    ap.add_argument('--detail', default=os.path.join(ROOT, 'bench_detail.json'),
                    help='where rank 0 writes the FULL result (per-kernel tables, provenance strings); stdout carries only the compact line')
    ap.add_argument('--fail-rank', type=int, default=-1, help=argparse.SUPPRESS)       # tests: this rank exits 3 before the first barrier
    ap.add_argument('--overlap-wgrad', type=int, default=1, choices=[0, 1],
                    help='1 (the library default): the weight gradients of the direct / narrow routes run on a second HIP stream beside the '
                         'data-gradient convolutions - that is what the TIMED region runs.  With two kernels sharing the chip a launch\'s duration is no '
                         'longer attributable to it, so the per-kernel HIP-event times of the roofline block come from a SECOND region of the same K '
                         'steps on one stream (its own ms/step is reported as roofline.single_stream_ms_per_step).  0: everything on one stream, one region')
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------- self-launch
def self_launch(args):
    """Bare `python bench.py --gpus N`: one child process per rank, started before this process has imported torch or touched a GPU.

    The parent POLLS its children: when one exits non-zero (or the whole run exceeds --launch-timeout) the others - which would otherwise sit
    in a barrier / RCCL collective until the driver's limit - are terminated (SIGTERM, then SIGKILL; fresh processes only, nothing is
    re-exec'ed), the failing rank's stderr tail is printed and the parent exits non-zero within seconds.  Ranks != 0 keep their stderr (a
    temporary file each, relayed on failure); only their stdout is dropped, so that rank 0's JSON line is the only line on stdout."""
    import tempfile
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs, errs = [], []
    tmp = tempfile.mkdtemp(prefix='pcnn_bench_')
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        err = None if r == 0 else open(os.path.join(tmp, 'rank%d.stderr' % r), 'w+')
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=err))

    def tail(r, n=30):
        if errs[r] is None:
            return '(rank 0 writes to this stderr directly)'
        errs[r].flush()
        errs[r].seek(0)
        return ''.join(errs[r].readlines()[-n:])

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    deadline = time.monotonic() + args.launch_timeout
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                print('[bench] rank %d exited with status %d; stopping the other ranks.  Its stderr tail:\n%s' % (r, c, tail(r)), file=sys.stderr, flush=True)
                stop_all()
                rc = abs(c) or 1
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                live = [r for r, c in enumerate(codes) if c is None]
                print('[bench] --launch-timeout %.0f s exceeded with ranks %s still running; stopping them' % (args.launch_timeout, live), file=sys.stderr, flush=True)
                for r in live[:2]:
                    print('[bench] rank %d stderr tail:\n%s' % (r, tail(r)), file=sys.stderr, flush=True)
                stop_all()
                rc = 124
                break
            time.sleep(0.1)
    finally:
        for e in errs:
            if e is not None:
                e.close()
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return rc


# ----------------------------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline(sample_hw=512, seed=0, with_1024=True):
    """fwd+bwd of the identical layer graph on the host cores (oracle twin, fp32) for ONE sample_hw^2 grid: 2 warm-ups, then the MEDIAN of
    5 timed reps; optionally one 1024^2 grid beside it (1 warm-up + 2 reps: a bounded sample of the c4 workload's grid size)."""
    import numpy as np
    import torch
    from oracle import hpnn as ohpnn, torch_twin, loss as oloss
    from poisson_cnn_amd import configs
    full = configs.hpnn()
    cfg = full['model']
    torch_twin.set_dtype(torch.float32)
    try:
        # the GPU box gives one GPU's job a 16-core CPU share; using every visible core of the host oversubscribes the cgroup
        cores = max(1, min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16))
        torch.set_num_threads(cores)
        p = ohpnn.init_params(cfg, seed=seed)
        pt = {k: torch.tensor(v, dtype=torch.float32, requires_grad=not k.endswith(('moving_mean', 'moving_variance'))) for k, v in p.items()}
        L = oloss.loss_wrapper(global_batch_size=1, **full['training']['loss_parameters'])

        def measure(hw, warm, reps):
            rng = np.random.default_rng(seed)
            rhs = torch.tensor(rng.uniform(-1, 1, (1, 1, hw, hw)), dtype=torch.float32)
            dx = torch.tensor(rng.uniform(5e-3, 5e-2, (1, 1)), dtype=torch.float32)
            tgt = torch.tensor(rng.standard_normal((1, 1, hw, hw)) * 0.1, dtype=torch.float32)
            times = []
            for i in range(warm + reps):
                for v in pt.values():
                    v.grad = None
                t0 = time.perf_counter()
                pred = ohpnn.forward(torch_twin, cfg, pt, rhs, dx)
                loss = L(tgt, pred, rhs, np.concatenate([dx.numpy(), dx.numpy()], 1))
                loss.backward()
                if i >= warm:
                    times.append(time.perf_counter() - t0)
            return sorted(times)[len(times) // 2] if len(times) % 2 else 0.5 * (sorted(times)[len(times) // 2 - 1] + sorted(times)[len(times) // 2]), times
        H = sample_hw
        dt, times = measure(H, 2, 5)
        out = {'value': 1.0 / dt, 'unit': 'grids/s (%dx%d grids, fwd+bwd)' % (H, H), 'cores': cores, 'kind': 'port',
               'grids_per_s_1024_equivalent': H * H / dt / (1024.0 * 1024.0), 'seconds_per_rep': times,
               'sample': '2 warm-ups + median of %d timed reps of fwd+bwd on one %dx%d grid (%.2f s each), oracle torch-CPU twin in fp32 on %d threads - stand-in for TF-CPU'
                         % (len(times), H, H, dt, cores)}
        if with_1024 and H != 1024:
            d2, t2 = measure(1024, 1, 2)
            out['at_1024'] = {'value': 1.0 / d2, 'unit': 'grids/s (1024x1024 grids, fwd+bwd)', 'seconds_per_rep': t2,
                              'sample': '1 warm-up + %d timed reps on one 1024x1024 grid (%.1f s each)' % (len(t2), d2)}
    finally:
        torch_twin.set_dtype(torch.float64)
    return out


def dataset_block():
    """On-device FD reference-solution generator at 512^2 (BASELINE configs[4]) next to the scipy sparse-direct stand-in for pyamg."""
    import numpy as np
    import torch
    from oracle import dataset as ods
    from poisson_cnn_amd.dataset import numerical_dataset_generator
    N, H = 32, 512
    gen = numerical_dataset_generator(batch_size=N, batches_per_epoch=1, randomize_rhs_smoothness=True, rhs_random_smoothness_range=[3, 8], seed=0,
                                      output_shape=[H, H], return_rhs=True, return_boundaries=True, return_dx=True, boundary_smoothness=5)
    for _ in range(3):
        gen[0]
    torch.cuda.synchronize()
    rounds = []
    for _ in range(4):           # four rounds of ten batches: the first is a warm-up (allocator, table uploads), the MEDIAN of the other three counts -
        t0 = time.perf_counter()  # the generator's host side (numpy control points, launch glue) shares the box's cores with whatever else runs there
        for _ in range(10):
            inp, soln = gen[0]
        torch.cuda.synchronize()
        rounds.append((time.perf_counter() - t0) / 10)
    t = sorted(rounds[1:])[1]
    rhs, left, top, right, bottom, dx = [x.cpu().numpy().astype(np.float64) for x in inp]
    t0 = time.perf_counter()
    ref = ods.multigrid_poisson_solve(rhs[:2, 0], {'left': left[:2, 0], 'right': right[:2, 0], 'top': top[:2, 0], 'bottom': bottom[:2, 0]}, dx[:2, 0])
    tc = (time.perf_counter() - t0) / 2
    err = float(np.linalg.norm(soln[:2, 0].cpu().numpy() - ref) / np.linalg.norm(ref))
    n = H - 2
    return {'metric': 'reference-solution samples/s at 512^2 (control points -> legacy bicubic -> fp64 DST-I solve, on device)', 'value': N / t,
            'seconds_per_batch_rounds': rounds, 'value_is': 'median of rounds 2-4 (round 1 is a warm-up)', 'fp64_mfma_tflops': 8.0 * n ** 3 * N / t / 1e12, 'fp64_mfma_peak_tflops': PEAK_FP64_MFMA_TFLOPS,
            'fp64_mfma_frac': 8.0 * n ** 3 * N / t / 1e12 / PEAK_FP64_MFMA_TFLOPS, 'algorithm': 'GEMM DST-I on v_mfma_f64_16x16x4_f64, 8 n^3 FLOP per sample',
            'cpu_baseline': {'value': 1.0 / tc, 'unit': 'samples/s', 'kind': 'port',
                             'sample': '2 samples, scipy.sparse.linalg.splu of the same 5-point system (stand-in for pyamg), 1 thread'},
            'rel_l2_gpu_vs_cpu': err}


# ----------------------------------------------------------------------------------------------------------------- the benchmark
def launch_check(args, dp):
    """No model: times K all-reduces of a 22.2 MB fp32 bucket (the per-step gradient exchange) through the same rendezvous / barrier /
    max-over-ranks path as the real benchmark.  Runs on CPU (gloo); `dry_run` marks the line as not a benchmark result."""
    import torch
    dev = 'cuda' if dp.backend == 'nccl' else 'cpu'
    bucket = torch.full((5556956,), float(dp.rank + 1), dtype=torch.float32, device=dev)
    for _ in range(args.warmup):
        dp.all_reduce_sum(bucket)
    dp.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dp.all_reduce_sum(bucket)
    if dev == 'cuda':
        torch.cuda.synchronize()
    local = time.perf_counter() - t0
    dp.barrier()
    elapsed = dp.max_over_ranks(time.perf_counter() - t0)
    per_rank = [1e3 * v / args.steps for v in dp.gather_over_ranks(local)]
    if dp.rank == 0:
        print(json.dumps({'metric': 'launch-check (no model)', 'dry_run': True, 'value': args.steps / elapsed, 'unit': 'all-reduces/s (22.2 MB fp32 bucket)',
                          'n_gpus': dp.world_size, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'config': {'workload': 'launch-check', 'collective': dp.collective_name(), 'parallelism': 'dp%d' % dp.world_size,
                                     'ranks_seen': dp.ranks_seen(), 'rccl': dp.rccl_version()},
                          'rank_ms_per_step': {'max': max(per_rank), 'min': min(per_rank), 'ranks': per_rank, 'conv_ms_per_step': [0.0] * dp.world_size},
                          'affinity': AFFINITY}), flush=True)


PROFILE_ROUND = 'r06'
COMPACT_LIMIT = 3072      # bytes: the driver keeps ~8.6 KB of stdout; round 3's 23 KB line arrived cut and unparseable


def _r(x, nd=4):
    """round floats for the compact line (None and non-floats pass through)"""
    return round(x, nd) if isinstance(x, float) else x


def compact_line(d):
    """The ONE line rank 0 prints: every field of the driver's contract plus the headline figures of each block, <= COMPACT_LIMIT bytes.
    Everything else (per-kernel tables, provenance strings, per-round timings) is in the detail file (--detail, default bench_detail.json)."""
    if d.get('dry_run'):
        return d
    rf = d.get('roofline') or {}
    hbm = rf.get('hbm_bound') or {}
    dom = max((k for k in (rf.get('kernels') or []) if k.get('ms_per_step')), key=lambda k: k['ms_per_step'], default=None)
    cfg = d.get('config') or {}
    out = {k: d.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')}
    out['value'], out['ms_per_step'] = _r(out['value'], 3), _r(out['ms_per_step'], 3)
    out['config'] = {'workload': cfg.get('workload'), 'global_batch': cfg.get('global_batch'), 'grid': cfg.get('grid'), 'parallelism': cfg.get('parallelism'),
                     'math': cfg.get('math'), 'collective': cfg.get('collective'), 'ranks_seen': cfg.get('ranks_seen'), 'rccl': cfg.get('rccl')}
    out['rank_ms_per_step'] = {k: ([_r(x, 2) for x in v] if isinstance(v, list) else _r(v, 3)) for k, v in (d.get('rank_ms_per_step') or {}).items()}
    if d.get('affinity'):
        out['affinity'] = d['affinity']
    out['collective_ms'] = _r(d.get('collective_ms'), 4)
    out['roofline'] = {'bound': rf.get('bound'), 'kernel': 'all conv launches of a step (spectral + direct + narrow)', 'achieved': _r(rf.get('achieved'), 1),
                       'peak': rf.get('peak'), 'unit': rf.get('unit'), 'frac': _r(rf.get('frac')), 'traffic': rf.get('traffic'),
                       'achieved_is': 'algorithmic bytes (SURVEY 8d) / HIP-event time of all conv launches',
                       'algorithmic_bytes': rf.get('algorithmic_bytes_per_step'),
                       'executed_frac': _r(rf.get('executed_frac')), 'executed_is': 'PMC bytes (traffic) / the same time' if rf.get('traffic') else 'no stamped PMC summary',
                       'traffic_over_algorithmic': _r(rf.get('traffic_over_algorithmic'), 3), 'mfma_busy': _r(rf.get('mfma_busy')),
                       'conv_ms_per_step': _r(rf.get('conv_kernel_ms_per_step'), 2), 'single_stream_ms_per_step': _r(rf.get('single_stream_ms_per_step'), 2), 'hbm_bound_frac': _r(hbm.get('frac')),
                       'fused_stage_frac': _r((rf.get('fused_stage') or {}).get('frac')),
                       'dominant_kernel': None if dom is None else {'name': str(dom['kernel'])[:40], 'ms_per_step': _r(dom['ms_per_step'], 2),
                                                                    'avg_launch_ms': _r(dom.get('avg_launch_ms')), 'hbm_frac': _r(dom.get('hbm_frac'), 3),
                                                                    'mfma_busy': _r(dom.get('mfma_busy_frac'), 3)}}
    cb = d.get('cpu_baseline')
    if cb:
        out['cpu_baseline'] = {'value': _r(cb.get('value')), 'unit': cb.get('unit'), 'cores': cb.get('cores'), 'kind': cb.get('kind'), 'sample': str(cb.get('sample'))[:160]}
        if cb.get('at_1024'):
            out['cpu_baseline']['value_1024'] = _r(cb['at_1024'].get('value'), 5)
    c3 = d.get('c3')
    if c3:
        out['c3'] = {'value': _r(c3.get('value'), 3), 'unit': c3.get('unit'), 'ms_per_step': _r(c3.get('ms_per_step'), 3), 'roofline_frac': _r((c3.get('roofline') or {}).get('frac')), 'executed_frac': _r((c3.get('roofline') or {}).get('executed_frac')),
                     'hbm_bound_frac': _r(((c3.get('roofline') or {}).get('hbm_bound') or {}).get('frac'))}
    inf = d.get('inference')
    if inf:
        out['inference'] = {'error': str(inf['error'])[:100]} if 'error' in inf else {'ms_per_batch': _r(inf.get('ms_per_batch'), 2), 'grids_per_s': _r(inf.get('grids_per_s'), 1),
                                                                                     'hbm_bound_frac_inference': _r((inf.get('hbm_bound') or {}).get('frac'))}
    sp = d.get('split_f16')
    if sp:
        acc = sp.get('accuracy_vs_fp32') or {}
        out['split_f16'] = {'value': _r(sp.get('value'), 3), 'ms_per_step': _r(sp.get('ms_per_step'), 3), 'fwd_rel_l2': (acc.get('forward_output') or {}).get('rel_l2'),
                            'grad_rel_l2': (acc.get('flat_gradient') or {}).get('rel_l2')}
    ds = d.get('dataset')
    if ds:
        out['dataset'] = {'error': str(ds['error'])[:120]} if 'error' in ds else {'value': _r(ds.get('value'), 1), 'unit': 'samples/s at 512^2', 'frac': _r(ds.get('fp64_mfma_frac')),
                                                                                  'bound': 'fp64 mfma'}
    out['detail'] = d.get('detail_file')
    return out


def emit(detail, path):
    """rank 0: the full result to `path` (best effort) and to stderr as one '[bench-detail]' line; the compact line - and nothing else - to stdout."""
    try:
        with open(path, 'w') as f:
            json.dump(detail, f, indent=1)
        detail['detail_file'] = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError as e:
        detail['detail_file'] = 'not written: %r' % (e,)
    line = json.dumps(compact_line(detail))
    if len(line) > COMPACT_LIMIT:          # cannot happen with the fixed field list above; never let it pass silently
        raise RuntimeError('compact bench line is %d bytes (limit %d)' % (len(line), COMPACT_LIMIT))
    print(line, flush=True)




def pmc_summary(math, workload):
    """(summary dict, provenance string) of the committed PMC summary of this (workload, mode) - only if it was measured on these kernel sources."""
    from poisson_cnn_amd import _lib
    name = '%s_%s_pmc_summary_%s.json' % (PROFILE_ROUND, workload, math)
    path = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(path):
        return None, 'profiles/%s not found (tools/collect_pmc.sh writes it)' % name
    with open(path) as f:
        summ = json.load(f)
    if summ.get('source_hash') != _lib.source_hash():
        return None, 'profiles/%s was measured on other kernel sources (stamp %s, this tree %s): re-run tools/collect_pmc.sh' % (name, summ.get('source_hash'), _lib.source_hash())
    return summ, ('profiles/%s (2*FETCH_SIZE + WRITE_SIZE per launch, separate rocprofv3 --pmc passes; SQ_VALU_MFMA_BUSY_CYCLES / (128 GRBM_GUI_ACTIVE); '
                  'source stamp %s matches this tree)' % (name, summ['source_hash']))


def run(args):
    import numpy as np
    import torch
    from poisson_cnn_amd import configs, ops, parallel
    os.environ['PCNN_WGRAD_STREAM'] = str(args.overlap_wgrad)
    if args.fail_rank >= 0 and int(os.environ.get('RANK', '0')) == args.fail_rank:       # tests: a rank that dies before its first barrier
        print('[bench] rank %d: --fail-rank asked for this exit' % args.fail_rank, file=sys.stderr, flush=True)
        sys.exit(3)
    dp = parallel.DataParallel.from_env(timeout_s=args.timeout)
    if dp.world_size != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, dp.world_size))
    if args.workload == 'launch-check':
        return launch_check(args, dp)
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam

    full = configs.hpnn()
    model = Homogeneous_Poisson_NN_Legacy(**full['model'])
    dp.attach(model)

    def note(msg):
        if dp.rank == 0:
            print('[bench] ' + msg, file=sys.stderr, flush=True)

    modes = [args.math] if args.math else [m for m in args.modes.split(',') if m]
    note('model built, %d params; modes %s' % (model.count_params(), modes))
    DTYPE = {'fp32': 'f32', 'split_f16': 'f32 tensors and accumulate; products as 3 x fp16 MFMA on 2 x fp16 splits of the fp32 operands (opt-in mode)'}

    def make_batch(workload):
        per_gpu, H = WORKLOADS[workload]
        g = torch.Generator(device='cpu').manual_seed(4 + dp.rank)
        rhs = (torch.rand((per_gpu, 1, H, H), generator=g) * 2 - 1).cuda()
        rhs = rhs / rhs.abs().amax(dim=(1, 2, 3), keepdim=True)
        dx = (torch.rand((per_gpu, 1), generator=g) * 4.5e-2 + 5e-3).cuda()
        target = (torch.randn((per_gpu, 1, H, H), generator=g) * 0.1).cuda()
        return per_gpu, H, ((rhs, dx), target)

    def accuracy_gate(per_gpu, batch):
        """same weights, same inputs, the two math modes side by side (no optimizer step)"""
        (rhs, dx), target = batch
        res = {}
        for mode in ('fp32', 'split_f16'):
            ops.set_math_mode(mode)
            _, pred = model._loss_and_grads(rhs, dx.reshape(per_gpu, -1)[:, :1].contiguous(), target)
            torch.cuda.synchronize()
            res[mode] = (pred.double().clone(), model.store.flat_g.double().clone())

        def cmp(a, b):
            d = (a - b).abs()
            sig = b.abs() >= 1e-3 * b.abs().max()
            return {'rel_l2': float((a - b).norm() / b.norm()), 'max_abs_err_over_max_abs': float(d.max() / b.abs().max()),
                    'max_rel_err_where_ref_above_1e-3_of_max': float((d[sig] / b.abs()[sig]).max())}
        acc = {'what': 'split_f16 mode vs fp32 mode on this benchmark batch (rank 0 shard), identical weights and inputs',
               'forward_output': cmp(res['split_f16'][0], res['fp32'][0]), 'flat_gradient': cmp(res['split_f16'][1], res['fp32'][1])}
        note('accuracy gate: forward rel-L2 %.3g, gradient rel-L2 %.3g' % (acc['forward_output']['rel_l2'], acc['flat_gradient']['rel_l2']))
        return acc

    def timed(mode, batch, steps, warmup):
        ops.set_math_mode(mode)
        for _ in range(warmup):
            model.train_step(batch)
        torch.cuda.synchronize()
        note('%s: warm-up done; timing %d steps' % (mode, steps))
        overlap = bool(args.overlap_wgrad) and model.ctx.use_side

        def region(prof):
            dp.barrier()
            torch.cuda.synchronize()
            ops.set_kernel_timer(prof)
            t0 = time.perf_counter()
            for _ in range(steps):
                logs = model.train_step(batch)
            torch.cuda.synchronize()
            local = time.perf_counter() - t0
            dp.barrier()
            elapsed = time.perf_counter() - t0
            ops.set_kernel_timer(None)
            return dp.max_over_ranks(elapsed), local, float(logs['loss'])
        # THE timed region: exactly `steps` steps of the library as shipped.  With the side stream on, no per-launch events are recorded in it.
        prof = None if overlap else ops.KernelTimer()
        elapsed, local, loss = region(prof)
        # per-rank view of the same region (before the closing barrier): load imbalance between ranks shows as max > min
        per_rank = [1e3 * v / steps for v in dp.gather_over_ranks(local)]
        rank_ms = {'max': max(per_rank), 'min': min(per_rank), 'ranks': per_rank}
        note('%s: timed region %.3f s' % (mode, elapsed))
        single = None
        if overlap:                    # roofline region: the same steps on ONE stream, HIP events around every convolution launch
            model.ctx.use_side = False
            try:
                model.train_step(batch)
                prof = ops.KernelTimer()
                e2, _, _ = region(prof)
                single = 1e3 * e2 / steps
                note('%s: single-stream roofline region %.3f s' % (mode, e2))
            finally:
                model.ctx.use_side = True
        # every rank's own convolution-kernel time per step (HIP events of the single-stream region): with rank_ms_per_step.ranks this tells a slow
        # rank (clock, NUMA placement) from collective cost on a scaling curve
        conv_s = sum(prof.totals(k)[1] for k in ('conv_fwd', 'conv_wgrad', 'conv_bwd_fused', 'conv_stage', 'conv_fwd_post')) if prof is not None else 0.0
        rank_ms['conv_ms_per_step'] = [1e3 * v / steps for v in dp.gather_over_ranks(conv_s)]
        return elapsed, prof, loss, rank_ms, single

    def collective_ms(reps=10):
        """the gradient exchange alone: `reps` all-reduces of the flat gradient bucket, HIP-event time on this stream"""
        if dp.world_size == 1:
            return None
        flat = model.store.flat_g
        dp.all_reduce_sum(flat); torch.cuda.synchronize(); dp.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            dp.all_reduce_sum(flat)
        torch.cuda.synchronize()
        return 1e3 * dp.max_over_ranks(time.perf_counter() - t0) / reps

    def roofline(mode, prof, workload, steps):
        """The convolution kernels of one train step against the HBM roofline (they are bound by the tile spectra they stream, not by the
        matrix pipes): counter traffic over their HIP-event time, per kernel where the committed rocprofv3 summary carries the table."""
        flops, secs, calls = prof.totals('conv_fwd')
        wf, ws_, wc = prof.totals('conv_wgrad')
        bf, bs, bc = prof.totals('conv_bwd_fused')
        gf, gs, gc = prof.totals('conv_stage')                                # narrow resnet stages as one launch (three convolutions each)
        pf, ps, pc = prof.totals('conv_fwd_post')                             # narrow data gradients that also run the producer's activation backward (round 6)
        af, as_, ac = flops + wf + bf + gf + pf, secs + ws_ + bs + gs + ps, calls + wc + bc + gc + pc
        conv_s = as_ / max(steps, 1)
        alg_bytes = (prof.total_bytes('conv_fwd') + prof.total_bytes('conv_bwd_fused') + prof.total_bytes('conv_wgrad') + prof.total_bytes('conv_stage')
                     + prof.total_bytes('conv_fwd_post')) / max(steps, 1)
        summ, src = pmc_summary(mode, workload)
        traffic = busy = table = None
        if summ is not None:
            for k, v in summ['kernels'].items():
                if k.startswith('conv ('):
                    traffic, busy = v['traffic_bytes_per_launch'], v.get('mfma_busy_frac')
            table = [{'kernel': k, 'launches_per_step': v.get('launches_per_step', v['launches']), 'ms_per_step': v.get('ms_per_step'),
                      'avg_launch_ms': v.get('avg_launch_ms'), 'traffic_MB_per_launch': v['traffic_bytes_per_launch'] / 1e6, 'hbm_TBs': v.get('hbm_tbs'),
                      'hbm_frac': v.get('hbm_frac'), 'mfma_busy_frac': v.get('mfma_busy_frac')}
                     for k, v in sorted(summ['kernels'].items(), key=lambda kv: -(kv[1].get('ms_per_step') or 0.0)) if not k.startswith('conv (')]
        achieved = alg_bytes / conv_s / 1e9 if conv_s else None                # ALGORITHMIC bytes over the kernels' measured time
        executed = traffic / conv_s / 1e9 if (traffic and conv_s) else None      # what the kernels really move (PMC), over the same time
        ridge = PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)           # FLOP per byte below which the fp32 conv is HBM-bound
        # launches that do a convolution and nothing else (the north star's "conv forward vs HBM roofline").  Since round 6 the narrow DATA-GRADIENT launches also run
        # their producer's activation backward (kind conv_fwd_post) and left this set: what remains is the forward launches (compare inference.hbm_bound_frac_inference)
        # and the few plain data gradients; hbm_bound.incl_post_launches puts them back with the convolution's own bytes over their whole time
        hf, hb, hs, hc = prof.select(lambda k, f, b: k in ('conv_fwd', 'conv_stage') and b > 0 and f / b < ridge)
        nf, nb_, ns, nc = prof.select(lambda k, f, b: k in ('conv_fwd', 'conv_stage', 'conv_fwd_post') and b > 0 and f / b < ridge)
        sf, sb, ss, sc = prof.select(lambda k, f, b: k == 'conv_stage')   # narrow resnet stages as one launch: algorithmic bytes of the UNFUSED layers / time
        qf, qb, qs, qc = prof.select(lambda k, f, b: k == 'conv_fwd_post')
        df, db, ds, dc = prof.select(lambda k, f, b: k == 'deconv_fwd')
        rf, rb, rs, rc = prof.select(lambda k, f, b: k == 'resize_fwd')
        direct_tflops = af / as_ / 1e12 if as_ else None
        return {'bound': 'hbm',
                'kernel': 'all convolution launches of the timed steps (forward, data gradient, weight gradient): tiled spectral route (fft32 / fft64_fwd, '
                          'spec_mix / spec_mixw on the matrix cores, fft32 / fft64_inv with the fused epilogues: in-register FFTs, 64-point tiles for 13..15 taps) for the wide filters, '
                          'fp32-MFMA implicit GEMM, vector-ALU forward and 16x16x4-MFMA weight-gradient kernels for the 3x3 layers of <= 16 channels',
                'achieved': achieved, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': achieved / PEAK_HBM_GBS if achieved else None,
                'achieved_source': 'ALGORITHMIC bytes (every layer\'s input + output + filter, forward + data gradient + weight gradient: SURVEY 8d) / the summed HIP-event time of all convolution launches in this run',
                'executed_achieved': executed, 'executed_frac': executed / PEAK_HBM_GBS if executed else None,
                'executed_source': ('EXECUTED traffic (PMC bytes of all convolution kernels of a step, stamped summary) / the same HIP-event time' if traffic
                                    else 'no stamped PMC summary for this tree: executed traffic unknown'),
                'traffic': traffic, 'traffic_unit': 'bytes per training step over all convolution kernels (2*FETCH_SIZE + WRITE_SIZE)', 'traffic_source': src,
                'mfma_busy': busy, 'conv_kernel_ms_per_step': 1e3 * conv_s, 'launches': ac, 'avg_launch_ms': 1e3 * as_ / ac if ac else None,
                'algorithmic_bytes_per_step': alg_bytes, 'algorithmic_GBs': alg_bytes / conv_s / 1e9 if conv_s else None,
                'traffic_over_algorithmic': traffic / alg_bytes if (traffic and alg_bytes) else None,
                'direct_conv_equivalent': {'what': 'the direct convolution\'s 2 N Ho Wo kh kw Cin Cout FLOP over the same time - a speed-up measure, NOT a roofline '
                                                   'fraction: the spectral route does not execute these FLOP',
                                           'tflops': direct_tflops, 'speedup_vs_direct_fp32_peak': direct_tflops / PEAK_FP32_MFMA_TFLOPS if direct_tflops else None},
                'kernels': table,
                'fused_stage': {'what': 'narrow resnet stages (3x3, 4 / 8 channels) as ONE launch (pcnn_resnet3_fwd): the algorithmic bytes of the three unfused '
                                        'layers (8 tensor passes in training) over the launch time; the launch itself moves 5 passes',
                                'achieved': sb / ss / 1e9 if ss else None, 'frac': sb / ss / 1e9 / PEAK_HBM_GBS if ss else None, 'launches': sc,
                                'avg_launch_ms': 1e3 * ss / sc if sc else None},
                'hbm_bound': {'what': 'conv forward / data-gradient launches below the fp32 ridge (%.1f FLOP/B: the 3x3 tail with <= 8 channels and the Scaling convs)' % ridge,
                              'bound': 'hbm', 'achieved': hb / hs / 1e9 if hs else None, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s (algorithmic bytes)',
                              'frac': hb / hs / 1e9 / PEAK_HBM_GBS if hs else None, 'launches': hc, 'avg_launch_ms': 1e3 * hs / hc if hc else None,
                              'incl_post_launches': {'frac': nb_ / ns / 1e9 / PEAK_HBM_GBS if ns else None, 'launches': nc},
                              'narrow_dgrad_with_post': {'what': 'narrow data-gradient launches that also apply the producer\'s activation backward (input, output, the activation read, '
                                                                 'optional skip gradient and raw copy): the convolution\'s own bytes over the whole launch time',
                                                         'achieved': qb / qs / 1e9 if qs else None, 'frac': qb / qs / 1e9 / PEAK_HBM_GBS if qs else None, 'launches': qc},
                              'deconv_fwd': {'achieved': db / ds / 1e9 if ds else None, 'frac': db / ds / 1e9 / PEAK_HBM_GBS if ds else None, 'launches': dc},
                              'resize_fwd': {'achieved': rb / rs / 1e9 if rs else None, 'frac': rb / rs / 1e9 / PEAK_HBM_GBS if rs else None, 'launches': rc}}}

    def bench_workload(workload, wl_modes, steps, warmup, gate):
        per_gpu, H, batch = make_batch(workload)
        gbs = per_gpu * dp.world_size
        model.compile(loss=loss_wrapper(global_batch_size=gbs, **full['training']['loss_parameters']), optimizer=Adam(**full['training']['optimizer_parameters']))
        accuracy = accuracy_gate(per_gpu, batch) if (gate and 'split_f16' in wl_modes and 'fp32' in wl_modes) else None
        blocks = {}
        for mode in wl_modes:
            elapsed, prof, loss, rank_ms, single = timed(mode, batch, steps, warmup)
            blocks[mode] = {'value': gbs * steps / elapsed, 'unit': 'grids/s', 'ms_per_step': 1e3 * elapsed / steps, 'dtype': DTYPE[mode], 'final_loss': loss,
                            'rank_ms_per_step': rank_ms, 'roofline': roofline(mode, prof, workload, steps) if dp.rank == 0 else None}
            if blocks[mode]['roofline'] is not None:
                blocks[mode]['roofline']['single_stream_ms_per_step'] = single
                blocks[mode]['roofline']['measured_in'] = ('a second region of the same %d steps with every launch on one stream (the timed region overlaps the direct / narrow '
                                                           'weight gradients on a side stream, where a launch\'s duration is not attributable to it)' % steps) if single else 'the timed region'

            del prof
        if accuracy is not None:
            blocks['split_f16']['accuracy_vs_fp32'] = accuracy
        return per_gpu, H, gbs, blocks

    per_gpu, H, gbs, blocks = bench_workload(args.workload, modes, args.steps, args.warmup, True)
    coll_ms = collective_ms()

    def inference_block(workload, reps=5):
        """Forward only (model([rhs, dx]), the reference's inference path) on the headline batch: ms per batch and the HBM-bound view of its narrow
        convolution launches - algorithmic bytes of the UNFUSED layers over the launch time, the definition of roofline.hbm_bound."""
        _, _, ((rhs, dx), _) = make_batch(workload)
        ops.set_math_mode('fp32')
        for _ in range(2):
            model([rhs, dx])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            model([rhs, dx])
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        prof = ops.KernelTimer()
        ops.set_kernel_timer(prof)
        try:
            for _ in range(reps):
                model([rhs, dx])
            torch.cuda.synchronize()
        finally:
            ops.set_kernel_timer(None)
        ridge = PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
        hf, hb, hs, hc = prof.select(lambda k, f, b: k in ('conv_fwd', 'conv_stage') and b > 0 and f / b < ridge)
        af, ab, as_, ac = prof.select(lambda k, f, b: k in ('conv_fwd', 'conv_stage') and b > 0)
        return {'what': 'inference forward of the same batch (no saved activations): ms per batch; hbm_bound = its convolution launches below the fp32 ridge '
                        '(%.1f FLOP/B), algorithmic bytes of the unfused layers / HIP-event time' % ridge,
                'ms_per_batch': ms, 'grids_per_s': per_gpu / (ms * 1e-3),
                'hbm_bound': {'achieved': hb / hs / 1e9 if hs else None, 'frac': hb / hs / 1e9 / PEAK_HBM_GBS if hs else None, 'launches': hc // reps,
                              'ms': 1e3 * hs / reps},
                'all_conv_fwd': {'algorithmic_GBs': ab / as_ / 1e9 if as_ else None, 'frac': ab / as_ / 1e9 / PEAK_HBM_GBS if as_ else None, 'launches': ac // reps,
                                 'ms': 1e3 * as_ / reps}}
    inference = None
    if dp.world_size == 1 and not args.no_inference:
        note('inference block (forward only) ...')
        try:
            inference = inference_block(args.workload)
        except Exception as e:   # noqa: BLE001 - a side measurement must not kill the headline
            inference = {'error': repr(e)}
    # the other half of BASELINE.json's metric ("at 512^2 & 1024^2"): the default c4 run appends the 32 x 512^2 workload, fp32 mode, same timing rules
    c3 = None
    if args.workload == 'c4' and not args.no_c3 and modes[-1] == 'fp32' and (dp.world_size == 1 or args.c3):
        note('c3 block (32 x 512^2, fp32) ...')
        p3, H3, g3, b3 = bench_workload('c3', ['fp32'], args.steps, args.warmup, False)
        c3 = dict(b3['fp32'], metric='grids/sec (fwd+bwd) at %d^2' % H3, steps=args.steps, warmup=args.warmup,
                  config={'workload': 'c3: the same full train step on %d x %dx%d grids per GPU (BASELINE.json configs[2] size)' % (p3, H3, H3), 'global_batch': g3})
    if dp.rank != 0:
        return
    head = modes[-1]
    hb = blocks[head]
    W = H
    out = {
        'metric': 'grids/sec (fwd+bwd) at %d^2' % H, 'value': hb['value'], 'unit': 'grids/s',
        'n_gpus': dp.world_size, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': hb['ms_per_step'],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': hb['dtype'], 'data': 'synthetic',
        'config': {'workload': '%s: Homogeneous_Poisson_NN_Legacy(hpnn.json) full train step (fwd+bwd+loss+Adam%s), %d x %dx%d Dirichlet grids per GPU'
                               % (args.workload, ('+' + dp.collective_name()) if dp.world_size > 1 else '', per_gpu, H, W),
                   'global_batch': gbs, 'grid': [H, W], 'parallelism': 'dp%d' % dp.world_size, 'math': head, 'library_default_math': 'fp32',
                   'overlap_wgrad': bool(args.overlap_wgrad), 'final_loss': hb['final_loss'], 'collective': dp.collective_name() if dp.world_size > 1 else None,
                   'ranks_seen': dp.ranks_seen(), 'rccl': dp.rccl_version()},
        'rank_ms_per_step': hb['rank_ms_per_step'], 'collective_ms': coll_ms, 'affinity': AFFINITY,
        'roofline': hb['roofline'],
    }
    for mode in modes[:-1]:
        out[mode] = blocks[mode]
    if inference is not None:
        out['inference'] = inference
    if c3 is not None:
        out['c3'] = c3
    if dp.world_size == 1 and not args.no_cpu_baseline:
        note('cpu baseline (bounded sample: one %d^2 grid) ...' % args.cpu_baseline_hw)
        out['cpu_baseline'] = cpu_baseline(args.cpu_baseline_hw, with_1024=(args.workload == 'c4' and not args.no_cpu_baseline_1024))
    if dp.world_size == 1 and not args.no_dataset:
        note('dataset generator block ...')
        try:
            out['dataset'] = dataset_block()
        except Exception as e:   # the headline must not die on the side measurement
            out['dataset'] = {'error': repr(e)}
    try:                                       # detail file only: what the kept filter spectra cost (pcnn_set_filter_version)
        out['filter_cache'] = dict(ops.filter_cache_stats(), enabled=bool(ops.filter_version()))
    except Exception as e:
        out['filter_cache'] = {'error': repr(e)}
    emit(out, args.detail)


def bind_to_gpu_numa_node(local_rank):
    """Pins this rank to the CPUs of its GPU's NUMA node - os.sched_setaffinity in the rank's own process, BEFORE torch is
    imported or any HIP call is made (no numactl / taskset wrapper, nothing re-exec'ed).  The GPU's node comes from sysfs alone: the KFD topology
    lists the GPU nodes in HIP's enumeration order (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES applied on top), each with the PCI address whose
    sysfs entry names the NUMA node.  Anything missing or inconsistent - no topology, node -1, an empty intersection with the CPUs this process
    may use (a cgroup share) - leaves the affinity untouched.  PCNN_BENCH_AFFINITY=0 switches it off.  Returns a report dict."""
    rep = {'bound': False}
    if os.environ.get('PCNN_BENCH_AFFINITY', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        rep['why'] = 'disabled'
        return rep
    try:
        base = '/sys/class/kfd/kfd/topology/nodes'
        gpus = []
        for n in sorted(os.listdir(base), key=int):
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, n, 'properties')) if len(ln.split()) >= 2)
            if int(props.get('simd_count', '0')) > 0:
                loc, dom = int(props.get('location_id', '0')), int(props.get('domain', '0'))
                gpus.append('%04x:%02x:%02x.%x' % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 0x7))
        for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):       # each filters the list the previous one left
            v = os.environ.get(var)
            if v:
                idx = [int(t) for t in v.split(',') if t.strip().isdigit()]
                gpus = [gpus[i] for i in idx if i < len(gpus)]
        if not gpus:
            rep['why'] = 'no GPU nodes in the KFD topology'
            return rep
        bdf = gpus[local_rank % len(gpus)]
        node = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read())
        rep.update(pci=bdf, numa_node=node)
        if node < 0:
            rep['why'] = 'the device reports no NUMA node'
            return rep
        cpus = set()
        for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
            a, _, b = part.partition('-')
            cpus.update(range(int(a), int(b or a) + 1))
        mine = cpus & os.sched_getaffinity(0)
        if not mine:
            rep['why'] = 'the node shares no CPU with this process\'s allowed set'
            return rep
        os.sched_setaffinity(0, mine)
        rep.update(bound=True, cpus=len(mine))
    except (OSError, ValueError, KeyError, IndexError) as e:
        rep['why'] = repr(e)
    return rep


AFFINITY = None


def main():
    global AFFINITY
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    if args.gpus > 1:                       # a rank process (self-launched or the driver's torch.distributed.run child): torch is not imported yet
        AFFINITY = bind_to_gpu_numa_node(int(os.environ.get('LOCAL_RANK', '0')))
    run(args)


if __name__ == '__main__':
    main()
