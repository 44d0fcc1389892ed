"""Ensemble sampler driver with the surface of ``emcee.EnsembleSampler`` that ``lightcurve_mcmc`` and its
consumers use (reference fitting.py:130-148, 171-277): ``run_mcmc``, ``reset``, ``chain``, ``flatchain``,
``lnprobability``/``get_log_prob``, ``acceptance_fraction``.

The stretch move itself (Goodman & Weare 2010, red/blue halves, a = 2) runs on the GPU
(``lcf_sampler_*`` in ``include/lcf.h``):

* single GPU: the whole run is enqueued by one native call, no host round-trip per step;
* several GPUs (``torch.distributed`` initialised, one process per GPU): every rank holds the full ensemble,
  proposals and accept/reject are replicated from the same counter-based RNG, each rank evaluates the likelihood of
  its contiguous shard of the active half, and ONE all-gather of ``n_walkers/2`` float64 log-probabilities per
  half-step (RCCL over xGMI; latency-bound, <= 16 KiB) makes the ranks agree.  Chains are therefore identical for
  any number of GPUs.

The per-half-step protocol is factored into :class:`ShardedStretchDriver` over a small backend interface so that
the multi-rank logic can be exercised on CPU with ``gloo`` (tests inject a checker backend there).
"""
import numpy as np

from . import rng as _rng


class State(tuple):
    """``(coords, log_prob, random_state)`` -- unpacks like emcee's ``State`` (fitting.py:133)."""

    def __new__(cls, coords, log_prob, random_state=None):
        return super().__new__(cls, (coords, log_prob, random_state))

    coords = property(lambda self: self[0])
    log_prob = property(lambda self: self[1])
    random_state = property(lambda self: self[2])


def shard_bounds(n_items, world_size, rank):
    """Contiguous, balanced partition of ``range(n_items)``: returns ``(lo, hi, width)`` with ``width`` the padded
    per-rank slot size used by the all-gather."""
    width = -(-n_items // world_size)
    lo = min(rank * width, n_items)
    hi = min(lo + width, n_items)
    return lo, hi, width


class NativeBackend:
    """Backend over ``lcf_sampler``: device buffers live in the native library; the collective sees them as torch
    tensors through ``__cuda_array_interface__`` (no copy)."""

    def __init__(self, native_sampler, rows=False):
        self.ns = native_sampler
        self.n_half = (native_sampler.nwalkers + 1) // 2   # slots per half-step (an odd ensemble's larger colour)
        #: what the collective carries per proposal: its row of partial chi^2 sums + log-prior (no finalize launch;
        #: the protocol of the native ``lcf_sampler_run_sharded``), or its finished log-posterior
        self.rows = bool(rows)
        self._newlp = {}
        self._side = None

    def begin(self, first_step, nsteps, split, store):
        self.ns.begin(first_step, nsteps, split, store)

    def stream(self):
        """Raw handle of torch's current stream.  The native ABI reads handle 0 as "the engine's own stream", which
        is NOT ordered with torch's default stream: callers run under :meth:`stream_context` (a real side stream)."""
        import torch
        return torch.cuda.current_stream().cuda_stream

    def stream_context(self):
        """Context under which kernels enqueued through the ABI and torch collectives share one (non-default) stream."""
        import torch
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.ns.engine.device)
        self._side.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self._side)

    def stream_sync(self):
        if self._side is not None:
            self._side.synchronize()

    def propose(self, step, half):
        self.ns.propose(step, half, self.stream())

    def evaluate(self, lo, hi):
        self.ns.evaluate(lo, hi, self.stream())

    def half_step(self, step, half, lo, hi):
        if self.rows:
            self.ns.half_step_rows(step, half, lo, hi, self.stream())
        else:
            self.ns.half_step(step, half, lo, hi, self.stream())

    def accept(self, step, half):
        self.ns.accept(step, half, self.stream())

    def newlp(self):
        """float64 torch tensor aliasing the native buffer the collective works on, for the half-step drawn last:
        ``rows[n_half][row]`` or ``newlp[n_half]`` (first axis = proposal slot either way; the native side
        double-buffers it, so there are two aliases)."""
        if self.rows:
            ptr, row = self.ns.rows_ptr()
            shape = (self.n_half, row)
        else:
            ptr, shape = self.ns.newlp_ptr(), (self.n_half,)
        if ptr not in self._newlp:
            import torch

            class _Alias:
                __cuda_array_interface__ = {'shape': shape, 'typestr': '<f8', 'data': (ptr, False), 'version': 2,
                                            'strides': None}
            self._newlp[ptr] = torch.as_tensor(_Alias(), device=f'cuda:{self.ns.engine.device}')
        return self._newlp[ptr]

    def empty(self, n):
        """Staging buffer for ``n`` proposal slots, shaped like :meth:`newlp`."""
        import torch
        shape = (n, self.ns.rows_ptr()[1]) if self.rows else (n,)
        return torch.empty(shape, dtype=torch.float64, device=f'cuda:{self.ns.engine.device}')

    def finish(self):
        self.ns.check()


class ShardedStretchDriver:
    """Runs half-steps over a backend, sharding the likelihood evaluation over the ranks of a process group."""

    def __init__(self, backend, group=None, force_collective=False):
        import torch.distributed as dist
        self.backend = backend
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi, self.width = shard_bounds(backend.n_half, self.world, self.rank)
        self.collective = self.world > 1 or force_collective
        # equal shards: the all-gather runs IN PLACE on the native newlp buffer (rank r's input is the slice
        # [r*width, (r+1)*width) of the output) -- no staging copies, one collective per half-step
        self.in_place = backend.n_half == self.width * self.world
        if self.collective and not self.in_place:
            self._send = backend.empty(self.width)
            self._recv = backend.empty(self.width * self.world)

    def run(self, first_step, nsteps, split, store):
        """``split``: 'random' | 'identity' | int32 array (nsteps, nwalkers), see ``NativeSampler._split``."""
        b = self.backend
        b.begin(first_step, nsteps, split, store)
        if hasattr(b, 'stream_context'):
            with b.stream_context():
                self._loop(first_step, nsteps)
            b.stream_sync()
        else:
            self._loop(first_step, nsteps)
        b.finish()

    def _loop(self, first_step, nsteps):
        b = self.backend
        for k in range(nsteps):
            step = first_step + k
            for half in (0, 1):
                if hasattr(b, 'half_step'):
                    b.half_step(step, half, self.lo, self.hi)
                else:
                    b.propose(step, half)
                    b.evaluate(self.lo, self.hi)
                if self.collective and self.in_place:
                    newlp = b.newlp()
                    self.dist.all_gather_into_tensor(newlp, newlp[self.lo:self.hi], group=self.group)
                elif self.collective:
                    newlp = b.newlp()
                    n = self.hi - self.lo
                    if n:
                        self._send[:n].copy_(newlp[self.lo:self.hi])
                    self.dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
                    for r in range(self.world):  # unpad: rank r owns [r*width, min((r+1)*width, n_half))
                        lo, hi, _ = shard_bounds(b.n_half, self.world, r)
                        if hi > lo and r != self.rank:
                            newlp[lo:hi].copy_(self._recv[r * self.width:r * self.width + (hi - lo)])
                b.accept(step, half)


def partition(n_items, world_size, rank):
    """Indices of the items (transients, epochs) owned by ``rank``: contiguous, balanced blocks."""
    lo, hi, _ = shard_bounds(n_items, world_size, rank)
    return range(lo, hi)


class PopulationSampler:
    """Population mode (BASELINE configs[4]): many independent transients, each with its own light curve, engine and
    walker ensemble.  Embarrassingly parallel: transients are partitioned over the ranks of a process group (no
    communication) and, on each GPU, all of a rank's ensembles are enqueued on their own HIP streams before any is
    waited for, so that small per-transient kernels overlap on the device.

    ``problems``: sequence of ``(model, lc, priors)``; every rank passes the full list and keeps
    ``self.indices`` (its share).  ``seed + index`` keys each transient's RNG, so results do not depend on the
    number of GPUs."""

    def __init__(self, problems, nwalkers, seed=0, a=2.0, group=None, device=None):
        world, rank = 1, 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                world, rank = dist.get_world_size(group), dist.get_rank(group)
        except ImportError:
            pass
        self.indices = list(partition(len(problems), world, rank))
        self.samplers = {}
        for k in self.indices:
            model, lc, priors, *extra = problems[k]  # optional 4th entry: engine keywords (use_sigma, sigma_type)
            if device is not None:
                model.device = device
            eng = model.engine_for(lc, priors=priors, **(extra[0] if extra else {}))
            self.samplers[k] = EnsembleSampler(nwalkers, eng.ndim, eng, seed=seed + k, a=a)
        self.nwalkers = nwalkers

    def run_mcmc(self, initial_states, nsteps, store=True, batched=True):
        """``initial_states``: mapping/sequence index -> (nwalkers, ndim) coordinates, or None to continue.

        ``batched`` (default): one native call runs all transients in lock step with one proposal launch and one
        likelihood launch per half-step for the whole population; if the transients cannot share launches (mixed
        table placement etc.) or ``batched`` is false, every ensemble is enqueued on its own stream instead."""
        from .engine import LcfError, population_run
        samplers = list(self.samplers.values())
        for k, s in self.samplers.items():
            s._prepare(None if initial_states is None else initial_states[k])
        done = False
        if batched and samplers and len({s._steps_done for s in samplers}) == 1 and \
                len({s.randomize_split for s in samplers}) == 1:
            try:
                self.last_run_ms = population_run([s._native for s in samplers], samplers[0]._steps_done, nsteps,
                                                  'random' if samplers[0].randomize_split else 'identity', store)
                done = True
            except LcfError as exc:
                if exc.status == 6:
                    raise ValueError('Probability function returned NaN') from None
                if exc.status != 5:  # LCF_ERR_UNSUPPORTED -> per-transient streams
                    raise
        for s in samplers:
            if not done:
                s._launch(nsteps, store, asynchronous=True)
        for s in samplers:
            s._finish(nsteps, store)
        return {k: s._state for k, s in self.samplers.items()}

    def __getitem__(self, k):
        return self.samplers[k]


#: multi-rank drivers in the order they are probed (fastest expected first)
COLLECTIVES = ('rows', 'peers', 'allgather')
_probe_cache = {}   # (process group, world size) -> the driver the first probe in that group selected


def probe_collectives(make_sampler, dist, x0, probe_steps=20, group=None, modes=COLLECTIVES, device=None):
    """Which multi-rank driver runs an ensemble fastest on THIS node: every driver in ``modes`` gets a few untimed steps
    from ``x0`` on a sampler of its own (``make_sampler(mode)``); one that raises on any rank (its waits are bounded),
    that could not connect its peer memory, or that leaves the ranks with different replicas of the ensemble is out; of
    the rest the fastest (the slowest rank's time counts) wins.  Returns ``(sampler, report)``.

    Every rank takes the same path: success is AGREED (all-reduce of a flag) after the sampler is made, after its first
    short run and after the probe run, and a stage is skipped on every rank as soon as one rank failed the one before
    -- a rank that raised never sits in a different collective than the ranks that did not.

    ``device``: where the flags of those agreements live (``'cuda:<n>'`` of this rank's engine under RCCL; default: the
    current CUDA device under RCCL, the CPU otherwise)."""
    import time
    import torch
    dev = device if device is not None else ('cuda' if dist.get_backend(group) == 'nccl' else 'cpu')

    def agreed(ok):   # logical AND over the ranks
        flag = torch.tensor([1. if ok else 0.], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return float(flag[0]) == 1.

    weights = np.cos(np.arange(x0.size, dtype=np.float64)).reshape(x0.shape)
    report, best = {}, None
    for mode in modes:
        seconds, checksum, why, s = float('inf'), 0., None, None
        try:
            s = make_sampler(mode)
        except Exception as exc:  # noqa: BLE001
            why = f'{type(exc).__name__}: {exc}'[:200]
        ok = agreed(why is None)
        if ok:
            try:
                s.run_mcmc(x0, 5, store=False)
                if (mode == 'peers' and not s._peers) or (mode == 'rows' and not s._boards):
                    why = 'the peer memory could not be connected'
            except Exception as exc:  # noqa: BLE001
                why = f'{type(exc).__name__}: {exc}'[:200]
            ok = agreed(why is None)
        if ok:
            try:
                t0 = time.perf_counter()
                state = s.run_mcmc(None, probe_steps, store=False)   # (returns after the device has finished)
                seconds = time.perf_counter() - t0
                checksum = float(np.sum(np.asarray(state[0]) * weights))
            except Exception as exc:  # noqa: BLE001
                seconds, why = float('inf'), f'{type(exc).__name__}: {exc}'[:200]
            ok = agreed(why is None)
        agg = torch.tensor([seconds if np.isfinite(seconds) else 1e30, checksum, -checksum], dtype=torch.float64, device=dev)
        dist.all_reduce(agg, op=dist.ReduceOp.MAX, group=group)
        worst, hi, lo = float(agg[0]), float(agg[1]), -float(agg[2])
        ok = ok and worst < 1e29 and hi == lo
        report[mode] = {'ok': ok, 'ms_per_step': 1e3 * worst / probe_steps if worst < 1e29 else None,
                        'replicas_agree': hi == lo, 'note': why}
        if ok and (best is None or worst < best[0]):
            best = (worst, mode, s)
    if best is None:
        raise RuntimeError(f'no multi-GPU driver completed its probe: {report}')
    report['selected'] = best[1]
    report['probe_steps'] = probe_steps
    return best[2], report


class EnsembleSampler:
    """Drop-in for the subset of ``emcee.EnsembleSampler`` the reference uses, bound to one engine.

    Parameters
    ----------
    nwalkers, ndim : int
    engine : lightcurve_fitting_amd.engine.Engine
        Device engine whose log-posterior is sampled (priors baked in at engine creation).
    seed : int
        Key of the counter-based RNG (emcee uses NumPy's global state instead; sampler parity is statistical).
    a : float
        Stretch scale (emcee default 2.0).
    randomize_split : bool
        Fresh random red/blue colouring every step, as emcee's ``RedBlueMove`` does.
    """

    def __init__(self, nwalkers, ndim, engine, seed=0, a=2.0, randomize_split=True, group=None,
                 force_sharded=False, native_collectives=True, collective=None):
        from .engine import NativeSampler
        if nwalkers < 2 * ndim:  # emcee's check; odd ensembles are fine (the red-blue split is then ceil / floor)
            raise ValueError('It is unadvisable to use a red-blue move with fewer walkers than twice the number of '
                             'dimensions.')
        if ndim != engine.ndim:
            raise ValueError(f'ndim = {ndim} but the engine has {engine.ndim} parameters')
        self.nwalkers, self.ndim = nwalkers, ndim
        self.engine = engine
        self.seed = int(seed)
        self.randomize_split = randomize_split
        self._native = NativeSampler(engine, nwalkers, seed, a)
        self._group = group
        self._force_sharded = force_sharded  # run the phase-by-phase collective path even with a single rank
        self.native_collectives = native_collectives
        #: how the ranks of a multi-GPU run work together: 'rows' (every rank moves its share of the walkers with k_solo
        #: and stores their new rows into every rank's board over IPC-mapped memory), 'peers' (every rank evaluates its
        #: share, stores partial sums into every mailbox, replicates the bookkeeping), 'allgather' (the same with one RCCL
        #: all-gather per half-step; the default -- the only driver whose wire protocol is RCCL's own), or 'auto'
        #: (``collective='auto'`` / LCF_COLLECTIVE=auto): the first run of the first sampler of a process group probes
        #: all of them for a few steps (probe_collectives) and the group keeps the fastest that works.  The peer-memory
        #: drivers stay opt-in until an 8-GPU run has timed them (bench.py --gpus N probes by default and says what it saw).
        import os
        self.collective = collective or os.environ.get('LCF_COLLECTIVE', 'allgather')
        if self.collective not in COLLECTIVES + ('auto',):
            raise ValueError("collective must be 'auto', 'allgather', 'peers' or 'rows'")
        self.collective_probe = None   # the report of the probe this sampler ran (None: named, cached or single rank)
        self._boards = None
        self._peers = None  # True once the mailboxes are connected, False if unavailable
        self._comm = None  # NativeComm once created, False if unavailable
        self._steps_done = 0   # RNG step counter: never reset, so burn-in and sampling use disjoint streams
        self._chain_host = np.empty((0, nwalkers, ndim))
        self._lp_host = np.empty((0, nwalkers))
        self._chain_on_device = 0   # steps of the last stored run whose chain is still in HBM only
        self._naccepted = np.zeros(nwalkers, dtype=np.int64)
        self._acc_seen = None   # the native acceptance counts as of the last finished run
        self._nsteps_counted = 0
        self._state = None

    def _native_comm(self):
        """RCCL communicator for the native sharded loop, or None (then the Python-driven loop over
        torch.distributed collectives is used): needs equal shards and a loadable RCCL.  Every rank takes the same
        decision: the outcome of the (collective) creation is agreed with an all-reduce."""
        if self._comm is False:
            return None
        if self._comm is None:
            import torch
            import torch.distributed as dist
            from .engine import NativeComm
            world = dist.get_world_size(self._group)
            dev = f'cuda:{self.engine.device}' if dist.get_backend(self._group) == 'nccl' else 'cpu'

            def agreed(ok):  # logical AND over the ranks
                flag = torch.tensor([1 if ok else 0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self._group)
                return int(flag.item()) == 1

            # 1. everything that can fail locally (options, shard shape, binding RCCL) BEFORE any collective RCCL call,
            #    so that no rank is left waiting in ncclCommInitRank for one that bailed out
            comm = None
            if agreed(self.native_collectives and ((self.nwalkers + 1) // 2) % world == 0 and NativeComm.probe()):
                try:
                    comm = NativeComm(self.engine.device, self._group)
                except Exception:  # noqa: BLE001 - then every rank falls back to the torch.distributed path
                    comm = None
                if not agreed(comm is not None):
                    if comm is not None:
                        comm.close()
                    comm = None
            self._comm = comm if comm is not None else False
        return self._comm or None

    def _peer_mailboxes(self):
        """Connect the ranks' mailboxes (once): every rank exports an IPC handle, the handles travel over
        torch.distributed, every rank maps them all.  All ranks take the same decision (all-reduce of the outcome)."""
        if self._peers is None:
            import torch
            import torch.distributed as dist
            world, rank = dist.get_world_size(self._group), dist.get_rank(self._group)
            dev = f'cuda:{self.engine.device}' if dist.get_backend(self._group) == 'nccl' else 'cpu'

            def agreed(ok):
                flag = torch.tensor([1 if ok else 0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self._group)
                return int(flag.item()) == 1

            handle = None
            try:
                if world <= 8 and ((self.nwalkers + 1) // 2) % world == 0 and self._native.one_launch:
                    handle, _ = self._native.mailbox_export()
            except Exception:  # noqa: BLE001 - then every rank falls back to the all-gather
                handle = None
            ok = agreed(handle is not None)
            if ok:
                handles = [None] * world
                dist.all_gather_object(handles, handle, group=self._group)
                try:
                    self._native.mailbox_connect(world, rank, handles=handles)
                except Exception:  # noqa: BLE001
                    ok = False
                ok = agreed(ok)
            self._peers = ok
        return self._peers

    def _peer_boards(self):
        """Connect the ranks' row boards (once), as :meth:`_peer_mailboxes` connects the mailboxes: for the sharded run
        in which every rank moves its share of the walkers itself (``lcf_sampler_run_rows``)."""
        if self._boards is None:
            import torch
            import torch.distributed as dist
            world, rank = dist.get_world_size(self._group), dist.get_rank(self._group)
            dev = f'cuda:{self.engine.device}' if dist.get_backend(self._group) == 'nccl' else 'cpu'

            def agreed(ok):
                flag = torch.tensor([1 if ok else 0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self._group)
                return int(flag.item()) == 1

            handle = None
            try:
                # (whether the engine can run one workgroup per proposal is checked by board_connect on every rank)
                if world <= 8 and ((self.nwalkers + 1) // 2) % world == 0:
                    handle, _ = self._native.board_export()
            except Exception:  # noqa: BLE001 - then every rank falls back to the all-gather
                handle = None
            ok = agreed(handle is not None)
            if ok:
                handles = [None] * world
                dist.all_gather_object(handles, handle, group=self._group)
                try:
                    self._native.board_connect(world, rank, handles=handles)
                except Exception:  # noqa: BLE001
                    ok = False
                ok = agreed(ok)
            self._boards = ok
        return self._boards

    def _resolve_collective(self):
        """'auto' -> the driver this process group runs fastest: probed once per group (on samplers of their own, a few
        steps from this sampler's present state), then remembered.  Every rank gets here at the same run."""
        import torch.distributed as dist
        # (the fastest driver depends on the shape of a half-step -- walkers, parameters, parts of the light curve --, not
        # only on the node: one probe per group AND shape)
        key = (id(self._group), dist.get_world_size(self._group), self.nwalkers, self.ndim, self._native.rows_ptr()[1])
        if key not in _probe_cache:
            x0 = self._native.get_state()[0]
            made = []

            def make(mode):
                made.append(EnsembleSampler(self.nwalkers, self.ndim, self.engine, seed=self.seed + 7919,
                                            randomize_split=self.randomize_split, group=self._group, collective=mode,
                                            native_collectives=self.native_collectives))
                return made[-1]
            dev = f'cuda:{self.engine.device}' if dist.get_backend(self._group) == 'nccl' else 'cpu'
            _, self.collective_probe = probe_collectives(make, dist, x0, group=self._group, device=dev)
            _probe_cache[key] = self.collective_probe['selected']
            # The probe samplers own peer-mapped memory (boards, mailboxes) that other ranks may still be writing into:
            # every rank has finished its probe runs before any rank unmaps anything, and the unmapping is explicit
            # (not left to the garbage collector's order).
            dist.barrier(group=self._group)
            for s in made:
                s.close()
            made.clear()
            dist.barrier(group=self._group)
        self.collective = _probe_cache[key]

    def close(self):
        """Release the native sampler (device buffers, peer mappings) now instead of at garbage collection."""
        if self._comm:
            self._comm.close()
        self._comm = False
        self._native.close()

    def _distributed(self):
        try:
            import torch.distributed as dist
        except ImportError:
            return False
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return self._force_sharded or dist.get_world_size(self._group) > 1

    # --- emcee surface -------------------------------------------------------------------------------------------
    def reset(self):
        """Forget the stored chain (not the RNG position), like ``emcee.EnsembleSampler.reset``."""
        self._chain_host = np.empty((0, self.nwalkers, self.ndim))
        self._lp_host = np.empty((0, self.nwalkers))
        self._chain_on_device = 0
        self._naccepted[:] = 0
        self._nsteps_counted = 0

    def reserve_chain(self, nsteps):
        """Allocate the device memory of a stored run of ``nsteps`` steps ahead of it (a timed run then allocates
        nothing)."""
        self._collect_chain()
        self._native.reserve_chain(nsteps)

    def _collect_chain(self):
        """Bring the last stored run's chain to the host (once)."""
        if self._chain_on_device:
            chain, lp = self._native.get_chain()
            self._chain_host = np.concatenate([self._chain_host, chain])
            self._lp_host = np.concatenate([self._lp_host, lp])
            self._chain_on_device = 0

    @property
    def _chain(self):
        self._collect_chain()
        return self._chain_host

    @property
    def _lp(self):
        self._collect_chain()
        return self._lp_host

    def _prepare(self, initial_state, skip_initial_state_check=False):
        """Validate and upload the starting positions (or continue from the stored state)."""
        self._collect_chain()   # (the next run reuses the device's chain buffer)
        if initial_state is not None:
            coords = np.array(initial_state[0] if isinstance(initial_state, tuple) else initial_state,
                              dtype=np.float64)
            if coords.shape != (self.nwalkers, self.ndim):
                raise ValueError('incompatible input dimensions')
            if not np.all(np.isfinite(coords)):
                raise ValueError('At least one parameter value was infinite or NaN')
            if not skip_initial_state_check and np.linalg.matrix_rank(coords - coords.mean(0)) < self.ndim:
                raise ValueError('Initial state has a large condition number. '
                                 'Make sure that your walkers are linearly independent for the best performance')
            self._native.set_state(coords)
            self._acc0 = np.zeros(self.nwalkers, dtype=np.int64)
            self._acc_seen = None
        elif self._state is None:
            raise ValueError('Cannot have `initial_state=None` if run_mcmc has never been called.')
        else:
            # (the counts the last run left: remembered by _finish -- nothing but a run changes them)
            self._acc0 = self._acc_seen if self._acc_seen is not None else self._native.naccepted()
        if initial_state is not None and np.any(np.isnan(self._native.get_state()[1])):
            raise ValueError('Probability function returned NaN')
        self._in_flight = False

    def _launch(self, nsteps, store=True, asynchronous=True):
        split = 'random' if self.randomize_split else 'identity'
        if self.randomize_split and self.nwalkers > 16384:  # beyond the device sort: host-generated colouring
            split = _rng.split_permutations(self.seed, self._steps_done, nsteps, self.nwalkers)
        self._in_flight = False
        distributed = self._distributed()      # (asked once: every question below used to ask torch.distributed again)
        if self.collective == 'auto' and distributed:
            self._resolve_collective()
        try:
            if distributed and self.collective == 'rows' and self._peer_boards():
                import torch.distributed as dist
                dist.barrier(group=self._group)  # every rank has returned from its previous run (see lcf_sampler_run_rows)
                self._native.run_rows(self._steps_done, nsteps, split, store)
            elif distributed and self.collective == 'peers' and self._peer_mailboxes():
                import torch.distributed as dist
                dist.barrier(group=self._group)  # every rank has returned from its previous run (see lcf_sampler_run_peers)
                self._native.run_peers(self._steps_done, nsteps, split, store)
            elif distributed and self._native_comm() is not None:
                self._native.run_sharded(self._comm, self._steps_done, nsteps, split, store)
            elif distributed:
                ShardedStretchDriver(NativeBackend(self._native, rows=True), self._group,
                                     force_collective=self._force_sharded).run(self._steps_done, nsteps, split, store)
            elif asynchronous:
                self._native.run_async(self._steps_done, nsteps, split, store)
                self._in_flight = True
            else:
                self._native.run(self._steps_done, nsteps, split, store)
        except Exception as exc:
            self._acc_seen = None   # (see _finish)
            if getattr(exc, 'status', None) == 6:
                raise ValueError('Probability function returned NaN') from None
            raise

    def _finish(self, nsteps, store=True):
        try:
            if self._in_flight:
                self._native.wait()
        except Exception as exc:
            self._acc_seen = None   # (a run that ended midway: the counts on the device are not what the last run left)
            if getattr(exc, 'status', None) == 6:
                raise ValueError('Probability function returned NaN') from None
            raise
        finally:
            self._in_flight = False
        self._steps_done += nsteps
        if store and nsteps:
            # The chain of a run stays in HBM until somebody reads it (or the next run needs the buffer): a run returns
            # when the device has finished, not when 80 bytes per walker and step have crossed PCIe.
            self._chain_on_device = nsteps
        x, lp, self._acc_seen = self._native.snapshot()     # (state and counts: one trip through the binding)
        self._naccepted += self._acc_seen - self._acc0
        self._nsteps_counted += nsteps
        self._state = State(x, lp, None)
        return self._state

    def run_mcmc(self, initial_state, nsteps, progress=False, progress_kwargs=None, skip_initial_state_check=False,
                 store=True, **kwargs):
        self._prepare(initial_state, skip_initial_state_check)
        self._launch(nsteps, store, asynchronous=False)
        return self._finish(nsteps, store)

    def get_chain(self, flat=False, thin=1, discard=0):
        c = self._chain[discard::thin]
        return c.reshape(-1, self.ndim) if flat else c

    def get_log_prob(self, flat=False, thin=1, discard=0):
        lp = self._lp[discard::thin]
        return lp.reshape(-1) if flat else lp

    @property
    def chain(self):
        """(nwalkers, nsteps, ndim), as emcee's deprecated-but-used ``.chain`` (fitting.py:139, 152)."""
        return np.swapaxes(self._chain, 0, 1)

    @property
    def flatchain(self):
        """(nwalkers * nsteps, ndim), walker-major like emcee's ``.flatchain`` (fitting.py:147, 203)."""
        s = self.chain.shape
        return self.chain.reshape(s[0] * s[1], s[2])

    @property
    def lnprobability(self):
        return self._lp.T

    @property
    def flatlnprobability(self):
        return self.lnprobability.reshape(-1)

    @property
    def iteration(self):
        return len(self._chain_host) + self._chain_on_device

    @property
    def acceptance_fraction(self):
        """Accepted proposals per walker and step since the last reset (steps run with ``store=False`` count)."""
        return self._naccepted / max(1, self._nsteps_counted)

    @property
    def last_run_ms(self):
        return self._native.last_run_ms()
