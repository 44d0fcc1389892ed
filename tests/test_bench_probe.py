"""bench.py's choice of the multi-rank driver (`--collective auto`), without a GPU: every driver gets a few untimed
steps; one that raises, that could not connect its peer memory, or that leaves the ranks with different replicas of the
ensemble is out; the fastest of the rest runs the timed steps."""
import argparse
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class FakeDist:
    """One process standing in for the group: a reduction returns what this rank contributed -- or, for the rank that
    disagrees, a different replica checksum; `other_rank_fails_at` = the agreement point (0: sampler made, 1: first run,
    2: probe run) at which ANOTHER rank reports a failure."""
    class ReduceOp:
        MAX = 'max'
        MIN = 'min'

    def __init__(self, other_rank_checksum_offset=0., other_rank_fails_at=None):
        self.offset = other_rank_checksum_offset
        self.fails_at = other_rank_fails_at
        self.agreements = 0

    @staticmethod
    def get_backend(group=None):
        return 'gloo'

    def all_reduce(self, t, op=None, group=None):
        if op == 'min':        # an agreement point
            if self.fails_at is not None and self.agreements % 3 == self.fails_at:
                t[0] = 0.
            self.agreements += 1 if float(t[0]) == 1. else 3 - self.agreements % 3   # (a failed stage skips the later ones)
            return
        if self.offset:        # another rank holds a different state: max(c, c') and max(-c, -c') no longer mirror
            t[1] = max(float(t[1]), float(t[1]) + self.offset)


class FakeSampler:
    def __init__(self, mode, seconds, fails=False, connected=True):
        self.collective, self.seconds, self.fails = mode, seconds, fails
        self._peers = self._boards = connected
        self.calls = 0

    def run_mcmc(self, x0, steps, store=False):
        self.calls += 1
        if self.fails:
            raise RuntimeError('rows did not arrive within 0.5 s')
        if x0 is None:
            time.sleep(self.seconds)
        return (np.ones((4, 2)), np.zeros(4), None)


def _args(collective='auto'):
    return argparse.Namespace(collective=collective, warmup=5)


def test_fastest_working_driver_is_selected():
    made = {}

    def make(mode):
        made[mode] = FakeSampler(mode, {'rows': 0.08, 'peers': 0.004, 'allgather': 0.2}[mode])
        return made[mode]
    s, report = bench.pick_collective(make, FakeDist(), np.ones((4, 2)), _args())
    assert report['selected'] == 'peers' and s is made['peers']
    assert all(report[m]['ok'] for m in ('rows', 'peers', 'allgather'))


def test_failing_and_unconnected_drivers_drop_out():
    def make(mode):
        return FakeSampler(mode, 0.001, fails=(mode == 'rows'), connected=(mode != 'peers'))
    s, report = bench.pick_collective(make, FakeDist(), np.ones((4, 2)), _args())
    assert report['selected'] == 'allgather' and s.collective == 'allgather'
    assert not report['rows']['ok'] and 'did not arrive' in report['rows']['note']
    assert not report['peers']['ok'] and 'could not be connected' in report['peers']['note']


def test_ranks_with_different_replicas_disqualify_a_driver():
    with pytest.raises(RuntimeError, match='no multi-GPU driver completed'):
        bench.pick_collective(lambda mode: FakeSampler(mode, 0.001), FakeDist(other_rank_checksum_offset=1.), np.ones((4, 2)),
                              _args())


def test_a_failure_on_another_rank_skips_the_later_stages_here_too():
    """The ranks agree on success after every stage: when another rank could not finish its first run, THIS rank does not
    start the probe run of that driver either (it would sit in the run's collectives alone)."""
    made = {}

    def make(mode):
        made[mode] = FakeSampler(mode, 0.001)
        return made[mode]
    with pytest.raises(RuntimeError, match='no multi-GPU driver completed'):
        bench.pick_collective(make, FakeDist(other_rank_fails_at=1), np.ones((4, 2)), _args())
    assert all(s.calls == 1 for s in made.values())      # the 5-step run only, never the probe run


def test_a_named_driver_is_not_probed():
    s, report = bench.pick_collective(lambda mode: FakeSampler(mode, 0.), FakeDist(), np.ones((4, 2)), _args('peers'))
    assert report is None and s.collective == 'peers' and s.calls == 0
    s, report = bench.pick_collective(lambda mode: FakeSampler(mode or 'single', 0.), None, np.ones((4, 2)), _args())
    assert report is None and s.collective == 'single'
