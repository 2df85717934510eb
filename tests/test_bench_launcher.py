"""bench.py --gpus N starts N ranks itself (no GPU touched by the parent); rehearsed here on CPU:
gloo instead of RCCL and a stand-in forward (--stub), everything else -- sharding by city, the
variable-length gather to a rotating collector, the metric all-reduce, the timing contract and the
JSON line -- is the code the GPU run uses."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(*flags, env=None):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--stub', '--steps', '3', '--warmup', '1'] + list(flags)
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    e.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=e)


def _json_line(proc):
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, proc.stdout          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_six_city_mode_world2_gathers_in_sharding_order():
    res = _json_line(_run('--gpus', '2', '--config', '4', '--samples', '15'))
    assert res['n_gpus'] == 2 and [r[0] for r in res['ranks_seen']] == [0, 1]
    cities = [i % 6 for i in range(15)]
    want = sorted(range(15), key=lambda i: cities[i])               # grouped by city, stable
    assert res['expected_sample_order'] == want
    assert res['gathered_sample_order'] == want                     # rows arrive rank after rank, in that order
    assert res['samples_per_rank'] == [8, 7]
    # contiguous-by-city blocks: the two ranks share at most one city
    a, b = (set(c) for c in res['cities_per_rank'])
    assert len(a & b) <= 1 and a | b == set(range(6))
    nodes = (11, 13, 17, 19, 23, 29)
    assert res['gathered_rows'] == sum(nodes[c] for c in cities)
    assert res['metric_count'] == 15.0 and res['scaling'] == 'strong' and res['steps'] == 3
    _check_phase_fields(res)


def _check_phase_fields(res):
    """A multi-rank line says where its steps went: forward and exchange time per step, what a rank
    ships per step, and the step time of the same run without the exchange."""
    for key in ('forward_ms', 'exchange_wait_ms', 'payload_MB_per_rank_and_step', 'ms_per_step_without_exchange'):
        assert key in res and res[key] >= 0.0, key
    assert res['payload_MB_per_rank_and_step'] > 0.0
    # the parts of a step cannot take longer than the step (measured on the same steps, per rank)
    assert res['forward_ms'] + res['exchange_wait_ms'] <= 1.5 * res['ms_per_step'] + 1.0


def test_default_mode_world2_and_world1():
    res = _json_line(_run('--gpus', '2'))
    assert res['n_gpus'] == 2 and len(res['ranks_seen']) == 2 and res['scaling'] == 'weak'
    _check_phase_fields(res)
    # step 2 (the last of 3) is collected on rank 2 % 2 = 0: both ranks' batches, rank order
    assert res['gathered_sample_order'] == list(range(16))
    one = _json_line(_run('--gpus', '1', '--config', '4', '--samples', '7'))
    assert one['n_gpus'] == 1 and one['gathered_sample_order'] == one['expected_sample_order']
    assert 'forward_ms' not in one                                  # a single rank exchanges nothing


def test_more_gpus_than_the_node_has_is_refused_before_anything_starts():
    import torch
    have = torch.cuda.device_count()
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(have + 2), '--steps', '1', '--warmup', '0']
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=e)
    assert proc.returncode == 2 and 'nothing was started' in proc.stderr


def test_rank_count_must_match_gpus_flag():
    # started by a launcher with a different world size than --gpus: refuse instead of mislabelling
    proc = _run('--gpus', '2', env={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert proc.returncode != 0 and 'WORLD_SIZE=1' in proc.stderr
