"""bench.py's multi-rank launcher up to the first GPU call, on a box without (enough) GPUs: a run that cannot be measured
must end with ONE JSON line carrying "error" and a non-zero exit code -- never a hang, never a traceback as the last line
(the first contact with an 8-GPU node will be the driver's)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _devices():
    import torch
    return torch.cuda.device_count()


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    e.pop('WORLD_SIZE', None); e.pop('RANK', None); e.pop('LOCAL_RANK', None)
    e.update(env or {})
    r = subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    return r, (json.loads(lines[-1]) if lines else None)


def test_more_ranks_than_devices_is_refused_by_the_parent():
    n = _devices() + 2
    r, last = _run(['--gpus', str(n), '--steps', '2', '--warmup', '1'])
    assert r.returncode == 2
    assert last is not None and last['value'] is None and 'error' in last and last['n_gpus'] == n
    assert last['visible_devices'] == n - 2
    assert 'Traceback' not in r.stderr


def test_a_rank_started_by_the_driver_checks_for_itself():
    """The driver starts `python -m torch.distributed.run ... bench.py --gpus N` itself: no parent of ours looks first."""
    n = _devices() + 1
    r, last = _run(['--gpus', str(n), '--steps', '2', '--warmup', '1'],
                   env={'WORLD_SIZE': str(n), 'RANK': '0', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29571'})
    assert r.returncode == 2
    assert last is not None and 'error' in last and last['visible_devices'] == n - 1


@pytest.mark.skipif(_devices() > 0, reason='with a GPU present the ranks would run the bench')
def test_failed_ranks_end_in_one_json_error_line():
    """The whole launcher path (torch.distributed.run --standalone, two ranks) with the sums routed through the host, on a
    box without any GPU: the ranks get as far as the device check and leave; the parent reports it."""
    r, last = _run(['--gpus', '2', '--steps', '2', '--warmup', '1'], env={'DBAT_BENCH_HOST_ALLREDUCE': '1'})
    assert r.returncode != 0
    assert last is not None and last['value'] is None and 'the ranks exited with code' in last['error']
    assert 'HIP device(s) visible' in r.stdout          # rank 0's own line is above the parent's


def test_watchdog_leaves_with_a_message():
    code = ("import sys, time; sys.path.insert(0, %r); sys.argv=['bench.py']; import bench\n"
            "with bench.Watchdog(0.2, 3, 'a phase that hangs'):\n    time.sleep(5)\n" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and 'rank 3: no progress within' in r.stderr and 'a phase that hangs' in r.stderr
