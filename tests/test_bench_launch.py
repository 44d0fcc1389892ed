"""bench.py --gpus N must run N ranks or fail: it starts them itself when no launcher did (fresh processes, before
anything touches a GPU), and refuses a launcher environment of another size.  CPU: the ranks only find each other
(gloo); GPU: a real two-rank bench line on one device."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    env.update(extra)
    return env


def test_gpus_flag_starts_that_many_ranks():
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--launch-check'], env=_clean_env(), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['ranks_seen'] == [0, 1] and line['spawned_by_bench']


def test_gpus_flag_must_match_the_launcher():
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--launch-check'],
                         env=_clean_env(WORLD_SIZE='3', RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999'),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and 'WORLD_SIZE=3' in out.stderr
    out = subprocess.run([sys.executable, BENCH, '--gpus', '0'], env=_clean_env(), capture_output=True, text=True,
                         timeout=120)
    assert out.returncode != 0


@pytest.mark.gpu
@pytest.mark.parametrize('workload', ['mcmc', 'companion'])
def test_two_rank_bench_line_on_one_device(workload):
    """Two ranks sharing the one GPU of the box (gloo collectives on device buffers; RCCL refuses two ranks on one
    device): the sharded code path end to end, and a line that says n_gpus == 2."""
    extra = ['--scaling', 'weak'] if workload == 'companion' else []
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--workload', workload, '--steps', '6', '--warmup', '2',
                          '--no-cpu-baseline'] + extra, env=_clean_env(LCF_BENCH_ONE_DEVICE='1'),   # (default wait bound: a cold box loads the ranks' code at different speeds)
                         capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(out.stdout.splitlines()) == 1, out.stdout[:500]   # ONE line: what libraries print goes to stderr
    line = json.loads(out.stdout)
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['steps'] == 6
    # all multi-rank drivers are probed (row boards and peer mailboxes over IPC, collectives -- gloo here), the fastest
    # one is timed
    coll = line['collective']
    probe = coll['probe']
    # (two ranks that SHARE a device can starve each other in the row-board run -- a launch that fills the device polls
    # for rows of a launch that waits for room -- so that driver may drop out here, with its bounded wait; on a node every
    # rank has its own GPU)
    assert all(probe[m]['ok'] and probe[m]['replicas_agree'] for m in ('peers', 'allgather'))
    assert probe['rows']['ok'] or 'not posted within' in probe['rows']['note'] or 'waited 0.5 s' in probe['rows']['note'] \
        or 'did not arrive' in probe['rows']['note']
    assert probe[probe['selected']]['ok'] and coll.get('group_ranks', 2) == 2
    assert ('row boards' in coll['driver']) == (probe['selected'] == 'rows')
    assert ('mailboxes' in coll['driver']) == (probe['selected'] == 'peers')
    assert 0 < line['roofline']['frac'] <= 1
