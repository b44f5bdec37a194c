"""GPU tier: the multi-GPU control flow of bench.py on RCCL, on the ONE GPU the test box has.

`SM3_BENCH_FORCE_DIST=1 python bench.py --gpus 1` initialises a 1-rank `nccl` (= RCCL) process group and runs exactly
what `--gpus N` runs: gradients packed into 64 MiB flat buckets inside the captured hipGraphs (thread-local capture
mode next to RCCL's watchdog thread), the backward replayed in two segments with the late-stage buckets all-reduced
(`ReduceOp.AVG`, asynchronously on RCCL's stream) while the early-stage backward replays, `finalize()` waiting on the
handles, the optimizer graph reading the reduced buckets, and the barrier / MAX / checksum collectives around the
timed region.  With one rank every collective is an identity, so the run must reproduce the plain single-process run:
same loss after the same number of steps.  What this cannot show is xGMI bandwidth or scaling -- no 2/4/8-GPU number
exists for this code yet."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', RANK='0',
               LOCAL_RANK='0', WORLD_SIZE='1', SM3_BENCH_RES='512', **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
                        '--no-ops', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    return json.loads(line), r.stderr


def test_split_backward_graph_replay_with_rccl_collectives_world_size_1():
    dist_run, err = _bench(dict(SM3_BENCH_FORCE_DIST='1', SM3_BENCH_SPLIT='1'))
    cfg = dist_run['config']
    assert 'capture failed' not in err, err[-1500:]
    assert cfg['dist_backend'] == 'nccl' and cfg['collective_avg'] is True
    assert cfg['hip_graph'] is True and cfg['split_backward'] is True
    assert cfg['grad_buckets'] >= 2 and cfg['replica_checksum_spread'] == 0.0
    plain, _ = _bench({})
    assert plain['config']['dist_backend'] is None and plain['config']['split_backward'] is False
    # identical data, seeds and step count: the averaged (1-rank) gradients are the gradients
    assert abs(dist_run['loss'] - plain['loss']) <= 1e-4 * max(1.0, abs(plain['loss'])), (dist_run['loss'], plain['loss'])
