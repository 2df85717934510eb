"""Multi-process (gloo, world size 2, CPU) test of the batch-sharding helpers used at N > 1."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gunet_amd import sharding
    cities = ['B', 'A', 'B', 'A', 'C', 'A', 'B']                     # 7 samples, 3 cities
    mine = sharding.shard_by_city(cities, world, rank)
    # every rank "predicts" rows that encode the sample index; cities have different node counts
    n_nodes = {'A': 3, 'B': 5, 'C': 2}
    rows = [torch.full((n_nodes[cities[i]], 4), float(i)) for i in mine]
    local = torch.cat(rows) if rows else torch.zeros((0, 4))
    gathered = sharding.gather_rows(local, dst=0)
    eq = sharding.gather_equal(torch.full((2, 3), float(rank)), dst=0)
    metrics = sharding.sum_metrics(torch.tensor([float(len(mine)), float(local.sum())], dtype=torch.float64))
    if rank == 0:
        np.savez(os.path.join(out_dir, 'r0.npz'), gathered=gathered.numpy(), eq=eq.numpy(), metrics=metrics.numpy())
    else:
        assert gathered is None and eq is None
    np.save(os.path.join(out_dir, f'mine{rank}.npy'), np.array(mine))
    # the per-step exchange of the benchmark: 5 steps, double-buffered, checked after draining
    ex = sharding.StepExchange(torch.zeros((3, 4)))
    for i in range(5):
        y = torch.full((3, 4), float(10 * i + rank))
        ex.post(i, y, torch.tensor([float(i), float(rank)], dtype=torch.float64))
    ex.drain()
    if rank == 0:
        for i in (3, 4):                                               # the last two steps are still in the slots
            want = torch.stack([torch.full((3, 4), float(10 * i + r)) for r in range(world)])
            assert torch.equal(ex.gathered[i & 1], want), i
    for i in (3, 4):
        assert ex.metrics[i & 1].tolist() == [float(i * world), float(sum(range(world)))]
    # rotating collector: step i is gathered on rank i % world
    ex = sharding.StepExchange(torch.zeros((3, 4)), rotate=True)
    for i in range(6):
        ex.post(i, torch.full((3, 4), float(10 * i + rank)), torch.tensor([1.0], dtype=torch.float64))
    ex.drain()
    for i in (4, 5):
        assert ex.collector[i & 1] == i % world
        if rank == i % world:
            want = torch.stack([torch.full((3, 4), float(10 * i + r)) for r in range(world)])
            assert torch.equal(ex.gathered[i & 1], want), i
        assert ex.metrics[i & 1].tolist() == [float(world)]
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    from gunet_amd import sharding
    for n in (0, 1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_gather_and_metrics_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    mine = [np.load(tmp_path / f'mine{r}.npy').tolist() for r in range(world)]
    assert sorted(mine[0] + mine[1]) == list(range(7))
    cities = ['B', 'A', 'B', 'A', 'C', 'A', 'B']
    order = mine[0] + mine[1]
    assert [cities[i] for i in order] == sorted(cities)              # grouped by city, contiguous blocks
    z = np.load(tmp_path / 'r0.npz')
    n_nodes = {'A': 3, 'B': 5, 'C': 2}
    want = np.concatenate([np.full((n_nodes[cities[i]], 4), float(i)) for i in order])
    assert np.array_equal(z['gathered'], want)
    assert np.array_equal(z['eq'], np.stack([np.zeros((2, 3)), np.ones((2, 3))]))
    assert z['metrics'][0] == 7 and z['metrics'][1] == want.sum()
