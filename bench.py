#!/usr/bin/env python3
"""Benchmarks of the MI355X light-curve likelihood engine.  One JSON line on stdout (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload mcmc|companion|population|sed]

Headline (`--workload mcmc`, the default): walker-steps/s of the ensemble sampler on BASELINE.json configs[1]
(ShockCooling, 1024 walkers per GPU, 500 synthetic epochs x {U,B,V,g,r,i} = 3000 points, float64).  A "step" is one
ensemble step: every walker gets one stretch-move proposal, i.e. one log-posterior evaluation of all 3000 points (two
half-steps of n_walkers/2 proposals each).

`--gpus N` with N > 1 and no launcher environment (no WORLD_SIZE): this process starts N fresh worker processes -- one
rank per GPU, before it makes any GPU call itself (it never imports torch) -- and returns their worst exit code.  Under
a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) the ranks are taken from the
environment; a WORLD_SIZE that differs from --gpus is an error, never a silently smaller run.  With N GPUs the walkers
of one ensemble are sharded: proposals and accept/reject are replicated, each rank evaluates its shard of the active
half, and one all-gather per half-step (RCCL) makes the ranks agree (`scaling`: weak = 1024 walkers per GPU for the
headline, strong = the 4096 walkers of configs[2] over the N GPUs for `--workload companion`).

Extra objects of the line:
  roofline      the dominant kernel against the FP64 vector-ALU issue rate.  `achieved` = vector-ALU lane-instructions
                of the band-sum loop THE SHIPPED ALGORITHM executes (samples it walks x the loop's instruction count
                from the ISA, tools/isa_count.py) / live kernel time (HIP events on the kernel's stream): a lower bound
                of what the kernel issues, so `frac` <= 1 by construction.  `executed_pmc` (all vector-ALU
                instructions, SQ_INSTS_VALU) and `traffic` come from rocprofv3 --pmc passes and are included only when
                the committed summary was collected from the kernel sources of this tree (hash recorded in the file).
                `algorithmic_speedup` = SURVEY 8d's instruction count of the reference's algorithm / the time.
  cpu_baseline  the CPU oracle in reference-shaped mode (one call per walker, Python loop over points) on 1 core
  collective    (N > 1) which driver ran the all-gather, the rank count the RCCL communicator itself reports, and the
                measured time of one per-half-step all-gather
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WALKERS_PER_GPU = 1024
N_EPOCHS = 500
BANDS = ['U', 'B', 'V', 'g', 'r', 'i']
TRUTH = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
SEED = 20241024
COMPANION_BANDS = ['U', 'B', 'V', 'g', 'r', 'i', 'DLT40', 'unfilt.']
COMPANION_TRUTH = np.array([57001., 0.5, 1.2, 57018., 1.05, 0.95, 0.9, 0.6])
COMPANION_WALKERS = 4096          # BASELINE configs[2]

# FP64 VALU instructions per Planck sample / per point of the REFERENCE's algorithm
ALG_INSTR_PER_SAMPLE, ALG_INSTR_PER_POINT = 34, 68
ALG_SAMPLES = N_EPOCHS * (13 + 11 + 15 + 89 + 75 + 89)        # Planck samples the reference evaluates: 146000
ALG_POINTS = N_EPOCHS * len(BANDS)
ALG_INSTR = ALG_INSTR_PER_SAMPLE * ALG_SAMPLES + ALG_INSTR_PER_POINT * ALG_POINTS
ALG_BYTES = 8 * (5 + 1)                                        # HBM bytes per walker-step: parameters in, lnL out
PEAK_FP64_TINSTR = 256 * 64 * 2.4e9 / 1e12                     # 39.3 T FP64 lane-instructions/s (= 78.6 TFLOP/s FMA)
PEAK_FP32_TINSTR = 2 * PEAK_FP64_TINSTR                        # 157.3 TFLOP/s FMA
PEAK_HBM_GBS = 8000.
# The float32 SED loop (k_sed<1>, hardware exponential and reciprocal): vector-ALU instructions per quad of samples
VALU_PER_QUAD_F32 = 21


def isa_counts():
    """Vector-ALU instructions per unit of work of the shipped loops, counted by the BUILD in the ISA of the library that
    runs (csrc/liblcf_hip.isa.json, written by the Makefile through tools/isa_count.py):
      quad_main   one trip of the band-sum loop over four Planck samples (exp via table + degree-4 polynomial, four
                  samples sharing one division)
      point_lean  one interpolated data point (interval, Horner on 8 coefficients, one table exponential, residual) of
                  the model-specialised half-step kernel; every other model's points are priced with the same number (what
                  they execute at least)
      state_lean  one log-space thermal state of a power-law model (short logarithm + table exponential)
      log_lean    its logarithm alone: the price of the state of a model that needs nothing else (companion shocking)
    Nothing here is typed in: without the report there is no roofline basis, and the line says so."""
    path = os.path.join(ROOT, 'lightcurve_fitting_amd', 'csrc', 'liblcf_hip.isa.json')
    try:
        doc = json.load(open(path))
        return {k: float(doc[k]) for k in ('quad_main', 'quad_safe', 'point_lean', 'state_lean', 'log_lean')}, None
    except Exception as exc:  # noqa: BLE001
        return None, f'no ISA report next to the library ({type(exc).__name__}): build with make -C lightcurve_fitting_amd/csrc'


# =====================================================================================================================
# launching
# =====================================================================================================================
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='mcmc', choices=['mcmc', 'sed', 'population', 'companion'],
                    help="'mcmc' (default) = the headline configs[1] line; the others print their own line")
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000,
                    help='ensemble steps in the timed region (BASELINE configs[1] runs 2000)')
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--scaling', choices=['weak', 'strong'], default=None,
                    help='mcmc: weak (1024 walkers per GPU, default) or strong (1024 in all); companion: strong (the '
                         '4096 walkers of configs[2] over the GPUs, default) or weak (512 per GPU)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--variant', type=int, default=3,
                    help='band sum: 3 = interpolated ln S(ln T) per filter where proved, Gauss-compressed tables elsewhere '
                         '(default), 2 = Gauss-compressed tables, 1 = the full tables, 0 = libm')
    ap.add_argument('--collective', choices=['auto', 'allgather', 'peers', 'rows'], default='auto',
                    help='N > 1: how the ranks work together.  rows: every rank moves its share of the walkers itself '
                         "(k_solo) and stores their new rows into every rank's board over IPC-mapped memory; peers: every "
                         'rank evaluates its share (k_fused), stores partial sums into every mailbox and replicates the '
                         'bookkeeping; allgather: the same with one RCCL all-gather per half-step.  auto (default): all '
                         'three are tried for a few untimed steps -- a driver that fails, or whose ranks end in different '
                         'states, is out -- and the fastest one runs the timed steps; the line says which and why')
    ap.add_argument('--end-to-end-child', action='store_true',
                    help='(internal) a fresh process times lightcurve_mcmc() calls from call to chain and prints them')
    ap.add_argument('--launch-check', action='store_true',
                    help='only start the ranks, let them find each other (gloo) and print what they saw')
    return ap.parse_args(argv)


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """Start n worker processes of this script (fresh interpreters: this parent has not touched the GPU and never
    does) and return the worst exit code.  Rank 0 prints the JSON line."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), LCF_BENCH_SPAWNED='1')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # The ranks are polled, not waited for one by one: as soon as one exits non-zero -- or the deadline passes -- the
    # others (which may sit in a collective that will never complete) are ended and the bench fails, instead of hanging.
    deadline = time.monotonic() + float(os.environ.get('LCF_BENCH_DEADLINE_S', '1500'))
    worst = 0
    while True:
        codes = [p.poll() for p in procs]
        failed = [c for c in codes if c not in (None, 0)]
        if all(c is not None for c in codes):
            return max([worst] + [abs(c) for c in codes])
        if failed or time.monotonic() > deadline:
            worst = max([abs(c) for c in failed] + [124])
            time.sleep(2.)                        # (a rank that is about to fail the same way gets to say so)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10.)
                except subprocess.TimeoutExpired:
                    p.kill()
            print(f'bench.py: rank exit codes {[p.poll() for p in procs]}'
                  + (' (deadline passed)' if not failed else ''), file=sys.stderr)
            return worst
        time.sleep(0.05)


def init_distributed(args):
    """(torch.distributed or None, rank, world, local_rank).  Dry runs on a single-GPU box: LCF_BENCH_ONE_DEVICE=1 maps
    every rank to cuda:0 and uses gloo (RCCL refuses two ranks on one device); the numbers of such a run are
    meaningless, the code path is the real one."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}')
    import torch
    if world == 1:
        torch.cuda.set_device(0)
        return None, 0, 1, 0
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if os.environ.get('LCF_BENCH_ONE_DEVICE') == '1':
        local_rank = 0
        # (ranks that share a device: their resident launches must fit the device side by side -- the dry runs use
        # ensembles of 128 proposals per rank and half-step; a launch that finds its workgroups not all started gives up
        # after 50 ms and the steps are repeated launch by launch)
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    return dist, rank, world, local_rank


def launch_check(args):
    """No GPU work: every rank joins a gloo group and reports; rank 0 prints what the group looks like."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}')
    seen = [rank]
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group('gloo', rank=rank, world_size=world)
        got = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(got, torch.tensor([rank], dtype=torch.int64))
        seen = sorted(int(g.item()) for g in got)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit({'launch_check': True, 'n_gpus': world, 'ranks_seen': seen,
              'spawned_by_bench': os.environ.get('LCF_BENCH_SPAWNED') == '1'})


def max_over_ranks(dist, elapsed):
    if dist is None:
        return elapsed
    import torch
    tmax = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return float(tmax.item())


def timed_run(sampler, dist, warmup, steps, x0, store=True, reps=None):
    """W untimed warm-up steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; the maximum
    over the ranks.  `store`: the timed run keeps its chain (every walker's position and log-posterior after every step,
    in HBM), as the reference's sampling run does (fitting.py:144-148: emcee's default).

    The timed region is REPEATED (`reps`, default LCF_BENCH_REPS or 5: each repetition is a run of exactly `steps` steps
    that continues the previous one, bracketed the same way) and the MEDIAN repetition is what the line reports: at the
    driver's 20 steps one repetition is a single 0.3 ms sample.  Returns (seconds of the median repetition, all
    repetitions' seconds, device milliseconds of the median repetition on this rank)."""
    import torch

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    import gc
    reps = reps or max(1, int(os.environ.get('LCF_BENCH_REPS', '5')))
    was_enabled = gc.isenabled()
    if was_enabled:      # As timeit does: a generation-2 collection of the interpreter (40-80 ms when it strikes, seen
        gc.collect()     # once in eight 0.6 ms runs, tools/debug/short_run_breakdown.py) is not the sampler's time.
        gc.disable()     # Collected BEFORE the warm-up (the caller may have done it even earlier, see quiet_interpreter):
    times, device = [], []
    try:                 # the timed steps must follow busy work without a pause in which the device clocks down.
        if store:
            sampler.reserve_chain(steps)              # (the chain's device memory: allocated with everything else)
        sampler.run_mcmc(x0, warmup, store=False)     # untimed warm-up (also allocates everything)
        for rep in range(reps):
            if rep:
                sampler.reset()                       # (untimed: the previous repetition's chain is dropped, not downloaded)
            barrier()
            t0 = time.perf_counter()
            sampler.run_mcmc(None, steps, store=store)    # returns after the device has finished
            barrier()
            times.append(max_over_ranks(dist, time.perf_counter() - t0))
            device.append(sampler.last_run_ms)
    finally:
        if was_enabled:
            gc.enable()
    mid = int(np.argsort(times)[(len(times) - 1) // 2])    # (an actual repetition: the lower median of an even count)
    return times[mid], times, device[mid]


def timed_region_block(times, steps, ran):
    """What the timed region consisted of: its repetitions (value = the median one) and what executed the half-steps."""
    return {'repetitions': len(times), 'steps_per_repetition': steps, 'seconds': [float(t) for t in times],
            'reported': 'median repetition', 'spread': float((max(times) - min(times)) / np.median(times)),
            'half_step_kernel': ran['kernel'], 'launches_per_repetition': ran['launches']}


def quiet_interpreter():
    """Collect the interpreter's garbage now and switch its collector off until the line is printed: the 50 ms a
    collection takes would otherwise fall between the device's busy phases (kernel timing -> warm-up -> timed steps) and
    let it clock down -- launches then take 13.2 instead of 12.5 us for the next milliseconds (rocprofv3 trace of a
    20-step run, tools/debug/trace_short_run.py)."""
    import gc
    gc.collect()
    gc.disable()


def pick_collective(make_sampler, dist, x0, args):
    """N > 1: the sampler that runs the timed steps.  `--collective auto` = the library's own probe
    (lightcurve_fitting_amd.sampler.probe_collectives: what EnsembleSampler(collective='auto') runs on its first multi-rank
    run), here with the bench's sampler factory so that the selected sampler itself runs the timed steps and the report
    goes into the line."""
    if dist is None or args.collective != 'auto':
        return make_sampler(None if dist is None else args.collective), None
    from lightcurve_fitting_amd.sampler import probe_collectives
    return probe_collectives(make_sampler, dist, x0, probe_steps=max(10, min(50, args.warmup * 4)))


def collective_info(sampler, dist, world):
    """Which driver a multi-rank run used, the communicator's own rank count, one all-gather's time."""
    if dist is None:
        return None
    n_half = (sampler.nwalkers + 1) // 2
    rows = sampler._native.rows_ptr()[1]
    if sampler.collective == 'rows' and sampler._boards:
        resident = sampler._native.last_run_kernel() == 'run'
        return {'driver': 'row boards: every rank moves its share of the walkers '
                          + ('with RESIDENT workgroups (k_solo_run<..., RANKS>: one launch per block of up to 32 steps, the '
                             "ranks' drift bounded by one progress word per rank and launch) " if resident else
                             'with a k_solo launch per half-step ')
                          + "and stores their new rows (position, log-posterior, acceptance count) straight into all ranks' "
                            'boards (IPC-mapped device memory); no collective, nothing replicated',
                'resident_workgroups': resident,
                'rccl_comm_ranks': None, 'group_ranks': dist.get_world_size(), 'allgather_us': None,
                'payload_bytes_per_rank': 16 * (sampler.ndim + 2) * n_half // world}
    if sampler.collective == 'peers' and sampler._peers:
        return {'driver': "peer mailboxes: every rank stores its shard's rows straight into all ranks' mailboxes "
                          '(IPC-mapped device memory) and polls its own; no collective, no launch between half-steps',
                'rccl_comm_ranks': None, 'group_ranks': dist.get_world_size(), 'allgather_us': None,
                'payload_bytes_per_rank': 16 * rows * n_half // world}
    comm = sampler._native_comm()
    if comm is not None:
        n, r = comm.count()
        return {'driver': 'native: lcf_sampler_run_sharded, ncclAllGather enqueued per half-step',
                'rccl_comm_ranks': n, 'allgather_us': 1e3 * comm.time_allgather(sampler._native, 200),
                'payload_bytes_per_rank': 8 * rows * n_half // world}
    import torch
    backend = dist.get_backend()
    dev = 'cuda' if backend == 'nccl' else 'cpu'
    w = (n_half + world - 1) // world
    buf = torch.zeros(w * world, rows, dtype=torch.float64, device=dev)
    r = dist.get_rank()
    for _ in range(5):
        dist.all_gather_into_tensor(buf, buf[r * w:(r + 1) * w].clone())
    if dev == 'cuda':
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        dist.all_gather_into_tensor(buf, buf[r * w:(r + 1) * w].clone())
    if dev == 'cuda':
        torch.cuda.synchronize()
    return {'driver': f'torch.distributed ({backend}) collectives driven from Python per half-step',
            'rccl_comm_ranks': None, 'group_ranks': dist.get_world_size(),
            'allgather_us': 1e4 * (time.perf_counter() - t0), 'payload_bytes_per_rank': 8 * rows * w}


# =====================================================================================================================
# roofline
# =====================================================================================================================
def kernel_source_sha():
    """Hash of everything the device code is compiled from: a PMC summary applies to this tree iff it carries it."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'lightcurve_fitting_amd', 'csrc')
    # (the Makefile too: its flags -- machine LICM off -- change the ISA as much as a source line does)
    for path in sorted([os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(('.hip', '.h')) or f == 'Makefile'] +
                       [os.path.join(ROOT, 'include', 'lcf.h')]):
        h.update(open(path, 'rb').read())
    return h.hexdigest()


def committed_pmc(tag):
    """Counters of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/r04_pmc_<tag>.json, written
    by tools/collect_profiles.sh; counters cannot be read from inside this process) -- or the reason they are not used."""
    path = os.path.join(ROOT, 'profiles', f'r04_pmc_{tag}.json')
    try:
        doc = json.load(open(path))
    except Exception as exc:  # noqa: BLE001
        return None, f'no PMC summary ({type(exc).__name__})'
    if doc.get('kernel_source_sha256') != kernel_source_sha():
        return None, ('profiles/' + os.path.basename(path) + ' was collected from other kernel sources (commit '
                      + str(doc.get('collected_at_commit')) + '): not used')
    return doc, None


def roofline_block(kernel, kern_ms, evals_per_launch, quads_per_eval, valu_per_quad, peak, alg_instr_per_eval,
                   alg_bytes_per_eval, pmc_tag, waves_per_launch=None, interp=None, hs_per_launch=None):
    """`achieved` / `frac`: lane-instructions of the band-sum loop the shipped algorithm executes per second, against
    the vector-ALU issue peak -- a lower bound of what the kernel issues.  `interp` = (points, epochs) per evaluation on
    the interpolated path (variant 3), counted at their own instruction counts; `quads_per_eval` then covers only the
    points that walk sample tables."""
    sec = kern_ms * 1e-3
    isa, isa_note = isa_counts()
    if valu_per_quad is None:      # float64 band sums: from the build's report
        valu_per_quad = isa['quad_main'] if isa else float('nan')
    per_eval = quads_per_eval * valu_per_quad
    basis = (f'{quads_per_eval:.0f} quads of samples per evaluation (shortest valid table per point) x {valu_per_quad:g} '
             'vector-ALU instructions per quad')
    if interp is not None:
        per_point = isa['point_lean'] if isa else float('nan')
        per_state = (isa[interp[2] if len(interp) > 2 else 'state_lean']) if isa else float('nan')
        per_eval += interp[0] * per_point + interp[1] * per_state
        basis += (f' + {interp[0]:.0f} interpolated points x {per_point:.1f} + {interp[1]:.0f} log-space thermal '
                  f'states x {per_state:g}')
    shipped = evals_per_launch * per_eval
    achieved = shipped / sec / 1e12
    out = {'bound': 'valu-issue', 'achieved': achieved, 'peak': peak, 'unit': 'Tinstr/s', 'frac': achieved / peak,
           'kernel': kernel, 'kernel_ms': kern_ms, 'evaluations_per_launch': evals_per_launch,
           'basis': basis + (' (counted by the build in the ISA of the library that runs: csrc/liblcf_hip.isa.json, '
                             'tools/isa_count.py -- in k_solo<5,1,true,2,false,ShockCooling> and its generic twin, the '
                             'launch-per-half-step kernels: the resident launch runs the same inlined half-step body, '
                             'plus spill code of its own that is not counted): the likelihood loops only, a lower bound '
                             'of the instructions issued'
                             if isa_note is None else f' -- {isa_note}'),
           'algorithmic_speedup': evals_per_launch * alg_instr_per_eval / sec / 1e12 / peak,
           'algorithmic_speedup_note': "SURVEY 8d's instruction count of the reference's algorithm per second / the "
                                       'issue peak: exceeds 1 where the shipped algorithm needs fewer instructions '
                                       '(not a utilisation)',
           'hbm': {'achieved': evals_per_launch * alg_bytes_per_eval / sec / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                   'frac': evals_per_launch * alg_bytes_per_eval / sec / 1e9 / PEAK_HBM_GBS,
                   'algorithmic_bytes_per_evaluation': alg_bytes_per_eval}}
    doc, why = committed_pmc(pmc_tag)
    if doc is None:
        out['traffic'] = None
        out['executed_pmc'] = None
        out['pmc_note'] = why
        return out
    # (resident launches: the profiled launches may cover another number of half-steps than this run's -- scaled per half-step)
    scale = hs_per_launch / doc['half_steps_per_launch'] if hs_per_launch and doc.get('half_steps_per_launch') else 1.
    c = {k: v['mean_per_launch'] * (1. if k == 'SQ_WAVES' else scale) for k, v in doc['counters'].items()}
    # FETCH_SIZE is doubled as the gfx950 correction for 16-B-per-lane coalesced reads prescribes (MI355X_MICROARCH.md,
    # HBM section); both counters are in KiB
    out['traffic'] = (2. * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024. if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c else None
    out['traffic_unit'] = 'bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)'
    out['executed_pmc'] = None
    if 'SQ_INSTS_VALU' in c and 'SQ_WAVES' in c:
        waves = waves_per_launch or c['SQ_WAVES']
        valu_launch = c['SQ_INSTS_VALU'] * waves / c['SQ_WAVES']   # (the SQ counters may cover a subset of the waves)
        real = 64. * valu_launch / sec / 1e12
        ex = {'achieved': real, 'peak': peak, 'unit': 'Tinstr/s', 'frac': real / peak,
              'valu_instr_per_wave': c['SQ_INSTS_VALU'] / c['SQ_WAVES'],
              'note': 'ALL vector-ALU lane-instructions issued (SQ_INSTS_VALU x 64 per launch) / live kernel time'}
        if 'SQ_ACTIVE_INST_VALU' in c and 'GRBM_GUI_ACTIVE' in c:
            # SQ_ACTIVE_INST_VALU counts quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs
            simd_cycles = 1024 * c['GRBM_GUI_ACTIVE'] / 8. * (c['SQ_WAVES'] / waves)
            ex['valu_busy'] = 4. * c['SQ_ACTIVE_INST_VALU'] / simd_cycles
        out['executed_pmc'] = ex
    out['pmc_provenance'] = {'file': f'profiles/r04_pmc_{pmc_tag}.json', 'collected_at_commit': doc.get('collected_at_commit'),
                             'kernel_source_sha256': doc.get('kernel_source_sha256')[:16], 'kernel': doc.get('kernel')}
    return out


def quads_per_evaluation(engine, truth):
    """(quads of samples one likelihood evaluation walks, points on the interpolated path) at the parameters `truth`:
    the table level is chosen per point from its temperature."""
    T, _ = engine.temperature_radius(np.asarray(truth, dtype=float))
    variant = getattr(engine, '_variant', 3)
    tabs, f = engine.tables, engine.filt_idx
    interp = (variant == 3) & (T[0] >= tabs.itmin[f]) & (T[0] <= tabs.itmax)
    n = np.where(interp, 0, tabs.samples_at(f, T[0], variant >= 2))
    return float(np.sum(n) / 4.), int(np.sum(interp))


def half_step_kernel_ms(engine, nwalkers, x0, seed, reps=992):
    """(average duration of ONE LAUNCH of the half-step kernel of a single-GPU run in ms, which kernel, half-steps per
    launch), from HIP events on the engine's stream around `reps` steps of back-to-back launches (the draw-record kernels
    in between: 30 us per 256 steps).  k_solo / k_fused: one launch = one half-step = nwalkers/2 proposals (proposal +
    thermal states + likelihood + accept test).  k_solo_run ('run'): the workgroups stay for a block of up to 128 steps
    (the first block of a run's draw records has 32 steps, the later ones 256: launches of 64 and 256 half-steps)."""
    from lightcurve_fitting_amd.engine import NativeSampler
    s = NativeSampler(engine, nwalkers, seed)
    used = s.set_half_step_kernel('auto')
    s.set_state(x0[:nwalkers])
    s.run(0, 50, 'random', False)
    s.run(50, reps, 'random', False)
    launches = s.last_run_launches()
    ms = s.last_run_ms() / launches
    s.close()
    return ms, used, 2. * reps / launches


# =====================================================================================================================
# workloads
# =====================================================================================================================
def build_problem(device):
    from lightcurve_fitting_amd import models as M
    rng = np.random.default_rng(SEED)
    epochs = np.sort(rng.uniform(0.5, 10., N_EPOCHS))
    t = np.repeat(epochs, len(BANDS))
    names = list(np.tile(BANDS, N_EPOCHS))
    model = M.ShockCooling(redshift=0., n=1.5)
    model.device = device
    ytrue = model(t, names, *TRUTH)                      # synthetic truth from the engine itself
    y = ytrue * (1. + 0.05 * rng.standard_normal(len(t)))
    dy = 0.05 * ytrue
    lc = {'MJD': t, 'filter': names, 'lum': y, 'dlum': dy}
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]
    return model, lc, priors


def initial_walkers(n):
    rng = np.random.default_rng(SEED + 1)
    lo, hi = TRUTH * 0.8, TRUTH * 1.2
    lo[4], hi[4] = 0., 0.2
    return rng.uniform(lo, hi, (n, 5))


def cpu_baseline(lc, budget_s=12., max_evals=128):
    """Reference-shaped CPU evaluation (per-walker call, per-point Python loop, K-sample trapezoid) on one core."""
    from oracle import lcf_oracle as O   # checker only: never on the product path
    bands = [O.band(n) for n in lc['filter']]
    model = ('ShockCooling', O.ShockCoolingOracle(0., 1.5))
    P = initial_walkers(max_evals)
    n = 0
    t0 = time.perf_counter()
    while n < max_evals and (time.perf_counter() - t0 < budget_s or n < 8):
        O.log_likelihood(model, lc['MJD'], bands, lc['lum'], lc['dlum'], P[n], reference_shaped=True)
        n += 1
    dt = time.perf_counter() - t0
    # second figure: the NumPy form vectorised over walkers -- the reference's own dense branch
    # (models.py:1163-1164), one process
    Pv = initial_walkers(64)
    tv = time.perf_counter()
    O.log_likelihood(model, lc['MJD'], bands, lc['lum'], lc['dlum'], Pv.T)
    dtv = time.perf_counter() - tv
    return {'value': n / dt, 'unit': 'walker-steps/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n} per-walker log-likelihood evaluations of the same {len(bands)}-point light curve '
                      f'(oracle in reference-shaped mode: Python loop over points), '
                      f'{dt:.1f} s on 1 of {os.cpu_count()} host cores',
            'ideal_pool_value': n / dt * (os.cpu_count() or 1), 'host_cores': os.cpu_count(),
            'vectorised_numpy_value': len(Pv) / dtv}


def cpu_baseline_c(lc, budget_s=8.):
    """The same evaluation through the plain-C oracle (oracle/lcf_oracle_c.c, gcc -O2, OpenMP over walkers): what an
    optimised CPU port reaches, on the host cores a 1-GPU slot owns.  Reported next to the reference-shaped figure."""
    from oracle import lcf_oracle as O
    so = os.path.join(ROOT, 'oracle', 'liblcf_oracle.so')
    if not os.path.exists(so):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True, capture_output=True)
    threads = max(1, min(16, os.cpu_count() or 1))
    bands = [O.band(n) for n in lc['filter']]
    orc = O.ShockCoolingOracle(0., 1.5)
    P = initial_walkers(64 * threads)
    O.c_shock_cooling_loglike(orc, lc['MJD'], bands, lc['lum'], lc['dlum'], P[:threads], threads)  # warm-up
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        O.c_shock_cooling_loglike(orc, lc['MJD'], bands, lc['lum'], lc['dlum'], P, threads)
        n += len(P)
    dt = time.perf_counter() - t0
    return {'value': n / dt, 'unit': 'walker-steps/s', 'cores': threads, 'kind': 'port-c',
            'sample': f'{n} log-likelihood evaluations of the same light curve in plain C (exact per-sample sums), '
                      f'{dt:.1f} s on {threads} threads'}


def run_mcmc(args):
    """BASELINE configs[1], the headline line."""
    dist, rank, world, local_rank = init_distributed(args)
    from lightcurve_fitting_amd.sampler import EnsembleSampler
    model, lc, priors = build_problem(local_rank)
    engine = model.engine_for(lc, priors=priors)
    engine.set_variant(args.variant)
    engine._variant = args.variant
    scaling = args.scaling or 'weak'
    # (a dry run with all ranks on ONE device: ensembles small enough that the ranks' launches fit the device side by
    # side -- workgroups of one rank that wait for rows of another must not depend on the driver's time-slicing)
    per_gpu = 256 if os.environ.get('LCF_BENCH_ONE_DEVICE') == '1' else WALKERS_PER_GPU
    n_walkers = per_gpu * (world if scaling == 'weak' else 1)
    x0 = initial_walkers(n_walkers)
    # The dominant kernel alone (HIP events over 2000 back-to-back launches), at this rank's share of a half-step when
    # the run is sharded.  Measured BEFORE the timed steps: a device that has just been handed over idles at low clocks,
    # and a 20-step run (0.6 ms) would be over before they have ramped (the driver's default is that short).
    quiet_interpreter()
    sampler, probe = pick_collective(lambda mode: EnsembleSampler(n_walkers, 5, engine, seed=SEED, collective=mode), dist,
                                     x0, args)
    kern_ms, used, hs_per_launch = (half_step_kernel_ms(engine, n_walkers // world, x0, SEED + 7)
                                    if rank == 0 or world > 1 else (None, None, None))
    elapsed, elapsed_all, device_ms = timed_run(sampler, dist, args.warmup, args.steps, x0)   # the chain is stored, as the reference's run does
    value = n_walkers * args.steps / elapsed
    ran = {'kernel': sampler._native.last_run_kernel(), 'launches': sampler._native.last_run_launches()}
    measured_in_the_run = False
    if world > 1 and getattr(sampler, 'collective', None) == 'rows' and ran['kernel'] == 'run' and ran['launches'] > 0:
        # The kernel the selected multi-rank driver actually launched: this rank's resident launches of the timed run
        # itself (HIP events around the whole run on the rank's stream / its launches: the row collection behind every
        # launch and the two small kernels around the run are inside -- an upper bound of the launch alone).
        kern_ms, used, hs_per_launch = device_ms / ran['launches'], 'run', 2. * args.steps / ran['launches']
        measured_in_the_run = True
    t0 = time.perf_counter()
    chain_shape = sampler.get_chain().shape                                # 48 bytes per walker and step over PCIe
    download_s = time.perf_counter() - t0
    elapsed_nochain, _, _ = timed_run(sampler, dist, args.warmup, args.steps, None, store=False)
    coll = collective_info(sampler, dist, world)
    if coll is not None:
        coll['probe'] = probe
    if rank == 0:
        # dominant kernel alone, at this rank's share of a half-step when the run is sharded over the GPUs
        per_rank = n_walkers // world
        quads, n_interp = quads_per_evaluation(engine, TRUTH)
        name = {'run': 'k_solo_run<5,1,true,2,ShockCooling> (resident workgroups, one per proposal: %d half-steps per '
                       'launch, each of them proposal + thermal states + likelihood + accept test; the rows travel from '
                       'workgroup to workgroup through a board of tagged rows in HBM)' % round(hs_per_launch),
                'solo': 'k_solo<5,1,true,2> (a whole half-step, one workgroup per proposal: proposal + thermal states '
                        '+ likelihood + accept test)',
                'fused': 'k_fused<5,1,true>', 'phases': 'k_step + k_points'}[used]
        # the committed counters are those of the default configuration (k_solo_run, interpolated level); any other
        # kernel or table level has no PMC pass of its own and reports null
        pmc_tag = 'k_solo_run_mcmc' if (used, args.variant) == ('run', 3) else f'k_{used}_v{args.variant}_mcmc'
        roof = roofline_block(name, kern_ms, round((per_rank // 2) * hs_per_launch), quads, None, PEAK_FP64_TINSTR, ALG_INSTR,
                              ALG_BYTES, pmc_tag, waves_per_launch=(per_rank // 2) * 8,
                              interp=(n_interp, N_EPOCHS) if n_interp else None, hs_per_launch=hs_per_launch)
        roof['half_steps_per_launch'] = hs_per_launch
        roof['kernel_ms_per_half_step'] = kern_ms / hs_per_launch
        if world > 1 and measured_in_the_run:
            roof['kernel'] = ('k_solo_run<5,1,true,2,ShockCooling,RANKS> (this rank\'s resident workgroups, one per proposal of '
                              'its share: %d half-steps per launch; rows from its own board, commits posted on every rank\'s '
                              'board) -- timed in the run itself, row collection included' % round(hs_per_launch))
        elif world > 1:
            roof['kernel'] += (" [one rank's share as a single-GPU run: the reference point of the multi-rank drivers that "
                               'launch k_solo<BOARD> / k_fused per half-step]')
        out = {
            'metric': 'walker-steps/sec (emcee ensemble)', 'value': value, 'unit': 'walker-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: ShockCooling (Sapir-Waxman n=1.5), 1024 walkers per GPU, '
                                   '500 synthetic epochs x 6 filters (UBVgri) = 3000 points, float64, '
                                   'device-resident stretch-move ensemble',
                       'walkers': n_walkers, 'points': ALG_POINTS, 'planck_samples_per_eval': ALG_SAMPLES,
                       'planck_samples_executed_per_eval': 4 * quads, 'points_interpolated_per_eval': n_interp,
                       'band_sum_variant': args.variant,
                       'parallelism': f'walker-sharded x{world}' if world > 1 else 'single GPU'},
            'roofline': roof, 'collective': coll,
            'timed_region': timed_region_block(elapsed_all, args.steps, ran),
            'device_ms_per_step': device_ms / args.steps if world == 1 else None,
            'chain': {'stored': True, 'shape': list(chain_shape),
                      'note': 'value is the run that keeps its chain in HBM (the reference: emcee stores every step, '
                              'fitting.py:144-148); below: the same steps without a chain, and with the chain also '
                              'copied to the host afterwards (PCIe-inclusive, never the headline)',
                      'value_without_chain': n_walkers * args.steps / elapsed_nochain,
                      'value_with_chain_on_host': n_walkers * args.steps / (elapsed + download_s) if world == 1 else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(lc)
            out['speedup_vs_cpu_baseline'] = value / out['cpu_baseline']['value']
            try:
                out['cpu_baseline_c'] = cpu_baseline_c(lc)
            except Exception as exc:  # noqa: BLE001 - the extra reference point must never break the bench line
                out['cpu_baseline_c'] = {'error': repr(exc)}
        if world == 1 and os.environ.get('LCF_BENCH_NO_E2E') != '1':
            out['end_to_end'] = end_to_end_block(lc)
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def end_to_end_child(args):
    """In a FRESH process: what a user's `lightcurve_mcmc(lc, model, priors, ..., nwalkers=1024, nsteps=2000,
    nsteps_burnin=1000)` costs from the call to the chain complete in HBM (fitting.py:16-168 is the unit a user runs; the
    timed region of the headline is its two run_mcmc calls only).  Three calls: the first one of the process (cold: HIP
    and the library's code objects are loaded inside it), a second one on another light curve (warm process, everything
    of the fit itself new: tables, engine, sampler), a third one on the second light curve again (its engine is still
    the model's).  Prints one JSON object."""
    t_start = time.perf_counter()
    import torch  # noqa: F401  (what `import lightcurve_fitting_amd` of a user's script costs most of; no GPU call yet)
    from lightcurve_fitting_amd import fitting
    from lightcurve_fitting_amd import models as M
    import_s = time.perf_counter() - t_start
    data = np.load(os.environ['LCF_E2E_LC'])     # the headline's light curve, made by the parent: nothing here has
    lc0 = {'MJD': data['MJD'], 'filter': [str(f) for f in data['filter']], 'lum': data['lum'], 'dlum': data['dlum']}
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]   # touched the GPU before the first call
    out = {'import_s': import_s, 'walkers': WALKERS_PER_GPU, 'steps': 2000, 'burn_in_steps': 1000, 'calls': []}
    lo, hi = TRUTH * 0.8, TRUTH * 1.2
    lo[4], hi[4] = 0., 0.2
    model, lc, sampler = None, None, None
    for k, label in enumerate(['cold process: first call (HIP, the library and its code objects are loaded inside it)',
                               'warm process: new model, new light curve, new engine',
                               'warm process: same light curve again (engine kept by the model)']):
        if k < 2:
            model = M.ShockCooling(redshift=0., n=1.5)
            lc = dict(lc0, lum=lc0['lum'] * (1. + 0.001 * k))    # (other photometry: nothing of the first call is reused)
        sampler = None                                           # (the previous call's sampler and its chain go now)
        np.random.seed(SEED + k)
        t0 = time.perf_counter()
        sampler = fitting.lightcurve_mcmc(lc, model, priors=priors, p_lo=lo, p_up=hi, nwalkers=WALKERS_PER_GPU,
                                          nsteps=2000, nsteps_burnin=1000, seed=SEED + k)
        wall = time.perf_counter() - t0
        t1 = time.perf_counter()
        shape = sampler.flatchain.shape
        out['calls'].append({'what': label, 'call_to_chain_in_hbm_s': wall, 'phases_s': sampler.timings,
                             'chain_to_host_s': time.perf_counter() - t1, 'flatchain_shape': list(shape),
                             'half_step_kernel': sampler._native.last_run_kernel(),
                             'walker_steps_per_s_of_the_whole_call': WALKERS_PER_GPU * 3000 / wall})
    print(json.dumps(out), flush=True)


def end_to_end_block(lc):
    """Runs end_to_end_child in a fresh interpreter (the GPU is shared with this idle process) and returns its object."""
    import tempfile
    try:
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
        tmp = tempfile.NamedTemporaryFile(suffix='.npz', delete=False)
        tmp.close()
        np.savez(tmp.name, MJD=lc['MJD'], filter=np.array(lc['filter']), lum=lc['lum'], dlum=lc['dlum'])
        env['LCF_E2E_LC'] = tmp.name
        res = subprocess.run([sys.executable, os.path.abspath(__file__), '--end-to-end-child'], env=env, capture_output=True,
                             text=True, timeout=300)
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
        if res.returncode != 0 or not lines:
            return {'error': (res.stderr or res.stdout)[-400:]}
        os.unlink(tmp.name)
        doc = json.loads(lines[-1])
        doc['note'] = ('lightcurve_mcmc(lc, model, priors, nwalkers=1024, nsteps=2000, nsteps_burnin=1000) in a fresh process, '
                       'wall time from the call to the chain complete in HBM; phases: checks / engine (band tables on the '
                       'host + device engine) / sampler / burn-in / run')
        return doc
    except Exception as exc:  # noqa: BLE001 - an extra block must never break the bench line
        return {'error': repr(exc)}


def build_companion(device):
    from lightcurve_fitting_amd import models as M
    rng = np.random.default_rng(SEED + 1)
    epochs = np.sort(rng.uniform(57001., 57060., 1000))
    t, names = np.repeat(epochs, 8), list(np.tile(COMPANION_BANDS, 1000))
    peak = {'U': 2.1e20, 'B': 2.6e20, 'V': 2.4e20, 'g': 2.5e20, 'r': 2.2e20, 'i': 1.7e20, 'DLT40': 2.2e20,
            'unfilt.': 2.2e20}
    lum0 = np.array([peak[n] for n in names]) * np.exp(-0.5 * ((t - 57018.) / 12.) ** 2)
    model = M.CompanionShocking({'MJD': t, 'filter': names, 'lum': lum0, 'dlum': 0.05 * lum0}, redshift=0.003)
    model.device = device
    ytrue = model(t, names, *COMPANION_TRUTH)
    lc = {'MJD': t, 'filter': names, 'lum': ytrue * (1 + 0.05 * rng.standard_normal(len(t))),
          'dlum': 0.05 * np.maximum(ytrue, 1e17)}
    priors = [M.UniformPrior(56990., 57010.), M.UniformPrior(0., 10.), M.UniformPrior(0., 10.),
              M.UniformPrior(57005., 57030.), M.UniformPrior(0.5, 2.)] + [M.UniformPrior(0., 3.)] * 3
    return model, lc, priors, lum0


def companion_walkers(nw):
    q = COMPANION_TRUTH
    x0 = q * (1 + 0.01 * np.random.default_rng(SEED + 2).standard_normal((nw, 8)))
    x0[:, [0, 3]] = q[[0, 3]] + 0.3 * np.random.default_rng(SEED + 3).standard_normal((nw, 2))
    return x0


def cpu_baseline_companion(lc, lum0, budget_s=15.):
    from oracle import lcf_oracle as O
    bands = [O.band(n) for n in lc['filter']]
    orc = O.CompanionShockingOracle(bands, lum0, z=0.003, variant=1)
    P = companion_walkers(2048)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        O.log_likelihood(('CompanionShocking', orc), lc['MJD'], bands, lc['lum'], lc["dlum"], P[n % len(P)])
        n += 1
    dt = time.perf_counter() - t0
    return {'value': n / dt, 'unit': 'walker-steps/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n} per-walker log-likelihood evaluations of the same 8000-point light curve (oracle: NumPy over '
                      f'the points of a band, SciPy cubic spline per band), {dt:.1f} s on 1 of {os.cpu_count()} host cores'}


def run_companion(args):
    """BASELINE configs[2]: CompanionShocking (Kasen shock + SiFTO template), 8 filters x 1000 epochs = 8000 points,
    4096 walkers in one ensemble sharded over the GPUs with one all-gather per half-step (strong scaling, the
    default: on one GPU all 4096 walkers), or 512 walkers per GPU (weak)."""
    dist, rank, world, local_rank = init_distributed(args)
    from lightcurve_fitting_amd.sampler import EnsembleSampler
    model, lc, priors, lum0 = build_companion(local_rank)
    scaling = args.scaling or 'strong'
    one_device = os.environ.get('LCF_BENCH_ONE_DEVICE') == '1'   # (dry run: launches that fit the device side by side)
    nw = (512 if one_device else COMPANION_WALKERS) if scaling == 'strong' else (128 if one_device else 512) * world
    engine = model.engine_for(lc, priors=priors)
    engine.set_variant(args.variant)
    engine._variant = args.variant
    x0 = companion_walkers(nw)
    quiet_interpreter()
    sampler, probe = pick_collective(lambda mode: EnsembleSampler(nw, 8, engine, seed=SEED, collective=mode), dist, x0,
                                     args)
    kern_ms, used, hs_per_launch = (half_step_kernel_ms(engine, nw // world, x0, SEED + 7, reps=192)
                                    if rank == 0 or world > 1 else (None, None, None))
    elapsed, elapsed_all, device_ms = timed_run(sampler, dist, args.warmup, args.steps, x0)
    value = nw * args.steps / elapsed
    ran = {'kernel': sampler._native.last_run_kernel(), 'launches': sampler._native.last_run_launches()}
    coll = collective_info(sampler, dist, world)
    if coll is not None:
        coll['probe'] = probe
    if rank == 0:
        per_rank = nw // world
        quads, n_interp = quads_per_evaluation(engine, COMPANION_TRUTH)
        full = int(engine.samples_per_eval)
        alg_instr = ALG_INSTR_PER_SAMPLE * full + (ALG_INSTR_PER_POINT + 20) * 8000   # + one cubic per point
        name = {'run': 'k_solo_run<8,1,true,4,generic> (resident 512-thread workgroups, %d half-steps per launch; a workgroup '
                       'takes its proposals of a half-step one after the other, its two halves two of the four parts each)'
                       % round(hs_per_launch),
                'solo': 'k_solo<8,1,true,4> (one 512-thread workgroup per proposal, its two halves take two of the four parts each)', 'fused': 'k_fused<8,1,true>',
                'phases': 'k_step + k_points'}[used]
        # (the committed counters are those of the default configuration -- k_solo, interpolated level, one GPU; anything
        # else has no PMC pass of its own and reports null; on N > 1 GPUs kernel_ms is the single-GPU k_solo launch of one
        # rank's share, whichever driver runs the timed steps)
        pmc_tag = ({'run': 'k_solo_run_companion', 'solo': 'k_solo_companion'}.get(used) if (args.variant, world) == (3, 1)
                   else None) or f'k_{used}_v{args.variant}_companion_x{world}'
        roof = roofline_block(name, kern_ms, round((per_rank // 2) * hs_per_launch), quads, None, PEAK_FP64_TINSTR, alg_instr,
                              8 * (8 + 1), pmc_tag, waves_per_launch=None if used == 'run' else (per_rank // 2) * 8,
                              interp=(n_interp, 1000, 'log_lean') if n_interp else None)
        roof['half_steps_per_launch'] = hs_per_launch
        roof['kernel_ms_per_half_step'] = kern_ms / hs_per_launch
        if world > 1:
            roof['kernel'] += " [one rank's share as a single-GPU launch: the reference point of the multi-rank drivers]"
        out = {'metric': 'walker-steps/sec (emcee ensemble)', 'value': value, 'unit': 'walker-steps/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
               'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'acceptance_fraction': float(sampler.acceptance_fraction.mean()),
               'config': {'workload': 'BASELINE configs[2]: CompanionShocking + SiFTO template, 4096 walkers, 8 filters '
                                      'x 1000 epochs = 8000 points, float64, walkers sharded over the GPUs',
                          'walkers': nw, 'points': 8000, 'planck_samples_per_eval': full,
                          'planck_samples_executed_per_eval': 4 * quads, 'points_interpolated_per_eval': n_interp},
               'roofline': roof, 'collective': coll,
               'timed_region': timed_region_block(elapsed_all, args.steps, ran),
               'device_ms_per_step': device_ms / args.steps if world == 1 else None}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_companion(lc, lum0)
            out['speedup_vs_cpu_baseline'] = value / out['cpu_baseline']['value']
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_population(args):
    """BASELINE configs[4]: population mode -- independent synthetic transients (config-2-like, 100 epochs x 6 filters
    = 600 points, own truth drawn +-20 %), 512 walkers each, transients partitioned over the GPUs (no communication).
    32 transients per GPU."""
    import torch
    dist, rank, world, local_rank = init_distributed(args)
    from lightcurve_fitting_amd import models as M
    from lightcurve_fitting_amd.sampler import PopulationSampler
    n_tr, nw = 32 * world, 512
    rng = np.random.default_rng(SEED + 5)
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]
    problems, x0, first_lc, first_truth = [], {}, None, None
    for k in range(n_tr):
        truth = TRUTH * rng.uniform(0.8, 1.2, 5)
        epochs = np.sort(rng.uniform(0.5, 10., 100))
        t, names = np.repeat(epochs, 6), list(np.tile(BANDS, 100))
        model = M.ShockCooling(redshift=0.)
        model.device = local_rank
        noise = rng.standard_normal(600)
        walkers = truth * rng.uniform(0.9, 1.1, (nw, 5))
        if k * world // n_tr == rank:  # only this rank's share is built on the device
            ytrue = model(t, names, *truth)
            lc = {'MJD': t, 'filter': names, 'lum': ytrue * (1 + 0.05 * noise), 'dlum': 0.05 * ytrue}
            problems.append((model, lc, priors))
            if first_lc is None:
                first_lc, first_truth = lc, truth
        else:
            problems.append((model, None, priors))
        x0[k] = walkers
    pop = PopulationSampler(problems, nw, seed=SEED, device=local_rank)
    for k in pop.indices:
        pop[k].engine.set_variant(args.variant)
    for k in pop.indices:
        pop[k].reserve_chain(args.steps)          # (the chains' device memory: allocated ahead of the timed steps)
    pop.run_mcmc(x0, args.warmup, store=False)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    pop.run_mcmc(None, args.steps, store=True)    # every ensemble keeps its chain in HBM, as the reference's run does
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(dist, time.perf_counter() - t0)
    if rank == 0:
        eng = pop[pop.indices[0]].engine
        eng._variant = args.variant
        quads, n_interp = quads_per_evaluation(eng, first_truth)
        pair_ms = pop.last_run_ms / (2 * args.steps)    # the launch(es) of one half-step of all transients
        used = pop[pop.indices[0]]._native.last_run_kernel()
        name = {'population': 'k_pop<5,1,4> (ONE launch per half-step for all 32 transients of this GPU: a workgroup per '
                              'four proposals, serial heads side by side, accept test included)',
                'population-run': 'k_pop_run<5,1,4> (the workgroups of all 32 transients of this GPU stay for blocks of up to 64 '
                                  'half-steps: 4 proposals at a time each, rows handed over through the transients\' boards of '
                                  'tagged rows, tables and interpolants staged in LDS once per launch; kernel_ms = one launch, '
                                  'kernel_ms_per_half_step = one half-step of all transients)',
                'population-phases': 'k_step_multi + k_points_multi (the two launches of a half-step of all 32 '
                                     'transients of this GPU; the likelihood launch dominates)'}[used]
        alg_instr = ALG_INSTR_PER_SAMPLE * int(eng.samples_per_eval) + ALG_INSTR_PER_POINT * 600
        launches = pop[pop.indices[0]]._native.last_run_launches() if used == 'population-run' else 2 * args.steps
        hs_per_launch = 2 * args.steps / launches       # (resident launches: blocks of up to 256 half-steps)
        roof = roofline_block(name, pair_ms * hs_per_launch, int(32 * nw // 2 * hs_per_launch), quads, None, PEAK_FP64_TINSTR,
                              alg_instr, ALG_BYTES,
                              'population' if (used, args.variant) == ('population-run', 3) else f'{used}_v{args.variant}',
                              interp=(n_interp, 100) if n_interp else None, hs_per_launch=hs_per_launch)
        out = {'metric': 'walker-steps/sec (population of independent ensembles)',
               'value': n_tr * nw * args.steps / elapsed, 'unit': 'walker-steps/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': 'BASELINE configs[4]: population mode, 32 transients per GPU x 512 walkers, 600 '
                                      'points each (100 epochs x UBVgri)',
                          'transients': n_tr, 'walkers_per_transient': nw, 'points': 600,
                          'planck_samples_executed_per_eval': 4 * quads},
               'roofline': roof, 'device_ms_per_step': pop.last_run_ms / args.steps}
        roof['half_steps_per_launch'] = hs_per_launch
        roof['kernel_ms_per_half_step'] = pair_ms
        if world == 1 and not args.no_cpu_baseline:
            from oracle import lcf_oracle as O
            bands = [O.band(n) for n in first_lc['filter']]
            mod = ('ShockCooling', O.ShockCoolingOracle(0., 1.5))
            n, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < 10.:
                O.log_likelihood(mod, first_lc['MJD'], bands, first_lc['lum'], first_lc['dlum'],
                                 x0[pop.indices[0]][n % nw], reference_shaped=True)
                n += 1
            dt = time.perf_counter() - t1
            out['cpu_baseline'] = {'value': n / dt, 'unit': 'walker-steps/s', 'cores': 1, 'kind': 'port',
                                   'sample': f'{n} per-walker log-likelihood evaluations of one transient (600 points, '
                                             f'oracle in reference-shaped mode), {dt:.1f} s on 1 of {os.cpu_count()} cores'}
            out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


def run_sed(args):
    """BASELINE configs[3]: per-epoch blackbody SED grid, 10 000 epochs x 6 filters (UBVgri) x 128 (T, R) candidates,
    float32 arithmetic with the float64 kernel as the error reference.  Epochs are independent: N GPUs = N replicas of
    an epoch shard (no communication); this line is the single-GPU figure."""
    if args.gpus != 1 or int(os.environ.get('WORLD_SIZE', '1')) != 1:
        raise SystemExit('bench.py --workload sed: replicas only, run it with --gpus 1')
    import torch  # noqa: F401  (one HIP runtime per process: see engine.load_library)
    from lightcurve_fitting_amd import bolometric as B
    from lightcurve_fitting_amd import models as M
    from lightcurve_fitting_amd.filters import PackedTables
    rng = np.random.default_rng(SEED + 4)
    n_ep, n_c = 10000, 128
    Tt, Rt = rng.uniform(5., 50., n_ep), 10 ** rng.uniform(-1., 2., n_ep)
    ytrue = M.blackbody_to_filters(BANDS, Tt, Rt)                     # (6, n_ep) on the GPU
    y = ytrue.T * (1. + 0.03 * rng.standard_normal((n_ep, 6)))
    epochs = [(BANDS, y[e], 0.03 * ytrue[:, e]) for e in range(n_ep)]
    cand = np.stack([rng.uniform(1., 100., (n_ep, n_c)), 10 ** rng.uniform(-2., 3., (n_ep, n_c))], axis=-1)
    like = B.SpectrumLikelihood(epochs, z=0.)
    out = {}
    reps = max(3, args.steps // 20)
    for prec in ('f64', 'f64-tables', 'f32'):
        res = like(cand, precision=prec)                              # warm-up
        ms = []
        for _ in range(reps):
            res = like(cand, precision=prec)
            ms.append(like.engine.last_kernel_ms)
        out[prec] = (res, float(np.median(ms)))
    alg_samples = n_ep * n_c * (13 + 11 + 15 + 89 + 75 + 89)
    # quads the sample-table kernels walk per candidate: the compressed ("cool") table of a filter where the candidate is
    # hot enough
    tabs = PackedTables(BANDS, z=0.)
    per_cand = np.zeros(cand.shape[:2])
    for f in range(6):
        nfull, nc = tabs.off[f + 1] - tabs.off[f], tabs.coff[f + 1] - tabs.coff[f]
        per_cand += (np.where((nc > 0) & (cand[..., 0] >= tabs.ctmin[f]), nc, nfull) + 3) // 4
    quads = float(per_cand.mean())
    exact = out['f64-tables'][0]
    err_interp = np.abs(out['f64'][0] - exact) / np.abs(exact)
    err_f32 = np.abs(out['f32'][0] - exact) / np.abs(exact)
    # The headline: float64 through the interpolants of ln S(ln T) -- the precision north_star's 1e-6 is stated at, and
    # faster than the float32 sample-table kernel configs[3] names.  Per candidate: one logarithm + 6 interpolated points;
    # the candidates outside the interpolants' range (T < 2 kK: ~1 %) are finished over the sample tables by k_sed_rest,
    # inside the timed launch pair.
    from lightcurve_fitting_amd.filters import INTERP_TMAX
    inside = (cand[..., 0] >= like.itab_tmin.max()) & (cand[..., 0] <= INTERP_TMAX)
    isa, isa_note = isa_counts()
    ms = out['f64'][1]
    shipped = float(inside.sum()) * (6 * isa['point_lean'] + isa['log_lean']) if isa else float('nan')
    rest_quads = float(per_cand[~inside].sum())
    shipped += rest_quads * (isa['quad_main'] if isa else float('nan'))
    roof = {'bound': 'valu-issue', 'achieved': shipped / (ms * 1e-3) / 1e12, 'peak': PEAK_FP64_TINSTR, 'unit': 'Tinstr/s',
            'frac': shipped / (ms * 1e-3) / 1e12 / PEAK_FP64_TINSTR,
            'kernel': 'k_sed_interp + k_sed_rest (float64: one lane per candidate, one logarithm, per observation one '
                      'coefficient row from LDS + Horner + one table exponential; candidates outside the interpolants '
                      'over the sample tables)', 'kernel_ms': ms, 'evaluations_per_launch': n_ep * n_c,
            'basis': (f'{int(inside.sum())} candidates x (6 interpolated points x {isa["point_lean"]:.1f} + one logarithm x '
                      f'{isa["log_lean"]:g}) + {rest_quads:.0f} quads of samples x {isa["quad_main"]:g} (counted by the '
                      'build in the ISA of the light-curve kernels that share these loops: csrc/liblcf_hip.isa.json)')
                     if isa else isa_note,
            'algorithmic_speedup': n_ep * n_c * (ALG_INSTR_PER_SAMPLE * 292 + 8 * 6) / (ms * 1e-3) / 1e12 / PEAK_FP64_TINSTR,
            'hbm': {'achieved': n_ep * n_c * 24 / (ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': n_ep * n_c * 24 / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 'algorithmic_bytes_per_evaluation': 24},
            'traffic': None}
    doc, why = committed_pmc('k_sed')
    roof['pmc_note'] = why if doc is None else None
    if doc is not None:
        c = {k: v['mean_per_launch'] for k, v in doc['counters'].items()}
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            roof['traffic'] = (2. * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024.
            roof['traffic_unit'] = 'bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)'
        roof['pmc_provenance'] = {'file': 'profiles/r04_pmc_k_sed.json', 'collected_at_commit': doc.get('collected_at_commit'),
                                  'kernel': doc.get('kernel')}
    line = {'metric': 'SED candidate evaluations/sec', 'value': n_ep * n_c / (ms * 1e-3),
            'unit': 'candidates/s', 'n_gpus': 1, 'steps': reps, 'warmup': 1, 'ms_per_step': ms,
            'dtype': 'f64', 'data': 'synthetic', 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'config': {'workload': 'BASELINE configs[3]: 10000 epochs x 6 filters (UBVgri) x 128 (T,R) candidates',
                       'planck_samples_per_launch': alg_samples, 'candidates_inside_the_interpolants': float(inside.mean()),
                       'quads_executed_per_candidate_by_the_sample_table_kernels': quads},
            'lnL_relative_error_vs_exact_sums': {'max': float(err_interp.max()), 'median': float(np.median(err_interp))},
            'kernel_ms_f64_interpolated': ms,
            'side_notes': {
                'float32_sample_tables (configs[3] as specified)': {
                    'kernel_ms': out['f32'][1], 'candidates_per_s': n_ep * n_c / (out['f32'][1] * 1e-3),
                    'lnL_relative_error': {'max': float(err_f32.max()), 'median': float(np.median(err_f32))},
                    'valu_per_quad': VALU_PER_QUAD_F32},
                'float64_sample_tables': {'kernel_ms': out['f64-tables'][1],
                                          'candidates_per_s': n_ep * n_c / (out['f64-tables'][1] * 1e-3)}},
            'roofline': roof}
    if not args.no_cpu_baseline:
        from oracle import lcf_oracle as O
        bands = [O.band(n) for n in BANDS]
        mod = ('Blackbody', O.ShockCoolingOracle(0., 1.5))
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.:
            e, c = n % n_ep, (n // n_ep) % n_c
            O.log_likelihood(mod, np.zeros(6), bands, y[e], 0.03 * ytrue[:, e], cand[e, c], reference_shaped=True)
            n += 1
        dt = time.perf_counter() - t0
        line['cpu_baseline'] = {'value': n / dt, 'unit': 'candidates/s', 'cores': 1, 'kind': 'port',
                                'sample': f'{n} candidate evaluations (6 filters each, oracle in reference-shaped mode), '
                                          f'{dt:.1f} s on 1 of {os.cpu_count()} host cores'}
        line['speedup_vs_cpu_baseline'] = line['value'] / line['cpu_baseline']['value']
    emit(line)


_RESULT_FD = None


def keep_stdout_for_the_result():
    """The contract is ONE JSON line on stdout.  Libraries write there too (gloo's "[Gloo] Rank 0 is connected ...",
    RCCL's version banner, both from C++): from here on descriptor 1 is the process's stderr, and the result line goes
    to the saved descriptor."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    data = (json.dumps(line) + '\n').encode()
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, data)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher: start the ranks here, BEFORE anything in this process initialises the GPU (torch is not even
        # imported in the parent)
        sys.exit(spawn_ranks(args.gpus))
    if args.end_to_end_child:
        return end_to_end_child(args)
    keep_stdout_for_the_result()
    if args.launch_check:
        return launch_check(args)
    {'mcmc': run_mcmc, 'companion': run_companion, 'population': run_population, 'sed': run_sed}[args.workload](args)


if __name__ == '__main__':
    main()
