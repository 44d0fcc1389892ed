"""Shared test helpers: golden light curves and a CPU checker backend for the sharded sampler protocol."""
import numpy as np

from conftest import golden
from oracle import lcf_oracle as O


def lc_dict(t, names, y, dy):
    return {'MJD': np.asarray(t), 'filter': [str(n) for n in names], 'lum': np.asarray(y), 'dlum': np.asarray(dy)}


def shockcooling_case():
    s = golden('shockcooling')
    return s, lc_dict(s['scb/t'], s['scb/names'], s['scb/y'], s['scb/dy'])


def config2_case():
    g = golden('config2')
    return g, lc_dict(g['cfg2/t'], g['cfg2/names'], g['cfg2/y'], g['cfg2/dy'])


def small_problem(npts=60, seed=5):
    """A small ShockCooling problem with a well-defined posterior, for sampler tests."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0.5, 8., npts))
    names = rng.choice(['U', 'B', 'V', 'g', 'r', 'i'], npts)
    bands = [O.band(n) for n in names]
    orc = O.ShockCoolingOracle(z=0., n=1.5)
    truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    ytrue = O.evaluate(('ShockCooling', orc), t, bands, truth)
    y = ytrue * (1 + 0.05 * rng.standard_normal(npts))
    dy = 0.05 * ytrue
    priors = [(0, 0., 10., 0., 1.)] * 4 + [(0, -1., 0.5, 0., 1.)]
    return dict(t=t, names=names, bands=bands, orc=orc, truth=truth, y=y, dy=dy, priors=priors)


def oracle_log_posterior(pb):
    model = pb.get('model') or ('ShockCooling', pb['orc'])

    def fn(block):
        block = np.atleast_2d(block)
        out = np.full(len(block), -np.inf)
        lp = np.array([O.log_prior(pb['priors'], p) for p in block])
        ok = np.isfinite(lp)
        if ok.sum() == 1:  # a single column would be squeezed away (np.squeeze in temperature_radius)
            out[ok] = lp[ok] + O.log_likelihood(model, pb['t'], pb['bands'], pb['y'], pb['dy'], block[ok][0])
        elif ok.any():
            out[ok] = lp[ok] + O.log_likelihood(model, pb['t'], pb['bands'], pb['y'], pb['dy'], block[ok].T)
        return out
    return fn


class OracleBackend:
    """CPU stand-in for ``NativeBackend`` (same protocol), used to test the multi-rank logic under gloo."""

    def __init__(self, log_prob_fn, coords, seed, a=2.):
        import torch
        self.torch = torch
        self.fn = log_prob_fn
        self.X = np.array(coords, dtype=np.float64)
        self.nw, self.ndim = self.X.shape
        self.n_half = (self.nw + 1) // 2   # slots per half-step: the larger colour of an odd ensemble
        self.LP = np.asarray(log_prob_fn(self.X), dtype=np.float64)
        self.seed, self.a = seed, a
        self._newlp = torch.zeros(self.n_half, dtype=torch.float64)
        self.chain = None

    def begin(self, first_step, nsteps, split, store):
        if isinstance(split, str):
            split = None if split == 'identity' else np.array(
                [O.split_permutation(self.seed, first_step + k, self.nw) for k in range(nsteps)])
        self.first, self.perm = first_step, split
        self.chain = np.empty((nsteps, self.nw, self.ndim))

    def propose(self, step, half):
        perm = self.perm[step - self.first] if self.perm is not None else np.arange(self.nw)
        sets = (perm[:self.n_half], perm[self.n_half:])
        self.act, oth = sets[half], sets[1 - half]
        z, j, self.lnu = O.stretch_draws(self.seed, step, half, self.act, len(oth), self.a)
        partner = self.X[oth[j]]
        self.Q = partner - (partner - self.X[self.act]) * z[:, None]
        self.zl = (self.ndim - 1.) * np.log(z)
        self._newlp.fill_(float('nan'))  # a rank that forgets to fill its shard is caught by the NaN check
        self.n_act = len(self.act)       # (one less than n_half in the second half-step of an odd ensemble)
        self._newlp[self.n_act:] = -np.inf

    def evaluate(self, lo, hi):
        hi = min(hi, self.n_act)
        if hi > lo:
            self._newlp[lo:hi] = self.torch.from_numpy(np.asarray(self.fn(self.Q[lo:hi]), dtype=np.float64))

    def newlp(self):
        return self._newlp

    def empty(self, n):
        return self.torch.empty(n, dtype=self.torch.float64)

    def accept(self, step, half):
        new = self._newlp.numpy()[:self.n_act]
        assert not np.any(np.isnan(new)), 'a shard of newlp was never filled'
        ok = self.zl + new - self.LP[self.act] > self.lnu
        self.X[self.act[ok]] = self.Q[ok]
        self.LP[self.act[ok]] = new[ok]
        self.chain[step - self.first] = self.X

    def finish(self):
        pass
