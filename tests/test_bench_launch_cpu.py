"""`python bench.py --gpus N` is a complete command: without a launcher around it the script re-executes itself
under torch.distributed.run (one process per rank, 127.0.0.1, free port) and rank 0 prints the one JSON line.  Runs here with `--dry`
(gloo, host stand-in for the device work): what is covered is the launcher, RANK / WORLD_SIZE plumbing, dist.GradSync with weight masses,
dist.all_gather_logits over round-robin chunk shards and the JSON contract -- not a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config')


def _run(cmd, env=None):
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout  # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize('n', [1, 2, 3])
def test_bench_launches_itself(n):
    line = _run([sys.executable, 'bench.py', '--gpus', str(n), '--steps', '3', '--warmup', '1', '--dry'])
    for k in KEYS:
        assert k in line, k
    assert line['n_gpus'] == n and line['steps'] == 3 and line['warmup'] == 1 and line['scaling'] == 'weak' and line['dry'] is True
    assert line['params_untouched'] is True  # broadcast made the ranks equal to rank 0, nothing else wrote the parameters
    # both scaling modes and the evidence of the collective layer in the line itself
    st = line['strong']
    assert st['scaling'] == 'strong' and st['chunks_per_gpu'] == max(1, 32 // n) and st['global_batch'] == st['chunks_per_gpu'] * n and st['value'] > 0
    co = line['collective']
    if n == 1:
        assert co is None
    else:
        assert co['backend'] == 'gloo' and co['world_size'] == n and co['allreduce_of_ones'] == n and co['allreduce_ok'] is True
        assert sorted(r['rank'] for r in co['ranks']) == list(range(n)) and len({r['pid'] for r in co['ranks']}) == n


def test_bench_under_an_external_launcher():
    """The driver's command shape: torch.distributed.run around bench.py --gpus N."""
    line = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                 '--master-port', '29713', 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--dry'])
    assert line['n_gpus'] == 2


@pytest.mark.parametrize('graph', [False, True])
def test_host_only_peer_ranks(graph):
    """MVP_REAL_RANKS = r: the ranks >= r of a bench job are host-only peers (bench.peer_run: the model on the CPU, the parameter broadcast,
    one dist.GradSync all-reduce per step, the real rank's barriers and closing MAX all-reduce) -- how a one-GPU box runs `--gpus 8` as one
    real rank + seven peers (tools/multi_rank_host.sh).  With r = 0 EVERY rank is a peer: the collective sequence must be consistent with
    itself (no deadlock, clean exit), eager and --graph (whose real rank adds five eager steps behind the timed region)."""
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e['MVP_REAL_RANKS'] = '0'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
           '29731' if graph else '29729', 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--train-only', '--extras', 'none'] + (['--graph'] if graph else [])
    out = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith('{')]  # peers print no result line


def test_launch_mode_defaults():
    """--launch: eager for one rank, the probing mode for N > 1 (eight ranks share a host), --graph / --launch graph the replay; the peer ranks
    of tools/multi_rank_host.sh resolve it the same way (they mirror the probe's collectives)."""
    import argparse
    import bench
    ns = lambda **kw: argparse.Namespace(**dict(dict(launch='', graph=False), **kw))
    assert bench.resolve_launch(ns(), 1) == 'eager'
    assert bench.resolve_launch(ns(), 8) == 'auto'
    assert bench.resolve_launch(ns(graph=True), 8) == 'graph'
    assert bench.resolve_launch(ns(launch='eager'), 8) == 'eager'
    assert bench.resolve_launch(ns(launch='auto'), 1) == 'auto'
    assert 0.0 <= bench.AUTO_GRAPH_RATIO <= 1.0 and bench.AUTO_PROBE_STEPS > 0

