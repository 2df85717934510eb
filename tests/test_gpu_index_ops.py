"""Bit-exact parity of the HIP index ("plan") kernels with the oracle, through the C ABI."""
import numpy as np
import pytest
import torch

from gpu_helpers import make_case, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import gunet_amd
    return gunet_amd


def _graphs(oracle):
    yield 'tiny', oracle.synth_road_graph(9, 14, 0.6, 3)
    yield 'ragged', oracle.synth_road_graph(37, 23, 0.35, 4)
    yield 'road', oracle.synth_road_graph(495, 436, 0.45, 0)


def test_quadrant_partition_micro(amd):
    # directions [0,1,2,8,4] -> q0=[e0,e1,e2] q1=[e2,e4] q2=[e4] q3=[e0]; e3 (self-loop) dropped
    ei = torch.arange(10, dtype=torch.int32).reshape(5, 2).cuda()
    d = torch.tensor([0, 1, 2, 8, 4], dtype=torch.int64).cuda()
    q = amd.data_utils.sample_quadrant_subgraphs(ei, d)
    assert [t.cpu().tolist() for t in q] == [[[0, 1], [2, 3], [4, 5]], [[4, 5], [8, 9]], [[8, 9]], [[0, 1]]]


def test_quadrant_partition_matches_oracle(amd, oracle):
    for name, (node_loc, ei, d, _) in _graphs(oracle):
        want = oracle.sample_quadrant_subgraphs(ei, d)
        got = amd.data_utils.sample_quadrant_subgraphs(to_dev(ei, torch.int32), to_dev(d, torch.int64))
        for q in range(4):
            assert got[q].dtype == torch.int32
            assert np.array_equal(got[q].cpu().numpy(), want[q]), (name, q)


def test_quadrant_partition_edge_cases(amd):
    empty = amd.data_utils.sample_quadrant_subgraphs(torch.zeros((0, 2), dtype=torch.int32).cuda(),
                                                     torch.zeros((0,), dtype=torch.int64).cuda())
    assert all(t.shape == (0, 2) for t in empty)
    only_self = amd.data_utils.sample_quadrant_subgraphs(torch.zeros((3, 2), dtype=torch.int32).cuda(),
                                                         torch.full((3,), 8, dtype=torch.int64).cuda())
    assert all(t.shape == (0, 2) for t in only_self)
    with pytest.raises(amd.GunetError):
        amd.data_utils.sample_quadrant_subgraphs(torch.zeros((2, 2), dtype=torch.int32).cuda(),
                                                 torch.tensor([1, 9], dtype=torch.int64).cuda())


def test_csr_is_stable_receiver_sort(amd, oracle):
    for name, (node_loc, ei, d, _) in _graphs(oracle):
        n = node_loc.shape[0]
        for q, e in enumerate(oracle.sample_quadrant_subgraphs(ei, d)):
            qg_edges = [to_dev(e, torch.int32)] * 4
            qg = amd.plan.QuadGraph(qg_edges, n)
            order = np.argsort(e[:, 1], kind='stable')
            counts = np.bincount(e[:, 1], minlength=n)
            rowptr = np.concatenate(([0], np.cumsum(counts)))
            assert np.array_equal(qg.rowptr[0].cpu().numpy(), rowptr), (name, q)
            assert np.array_equal(qg.perm[0].cpu().numpy()[:e.shape[0]], order), (name, q)
            assert counts.max() <= 3          # quadrant in-degree of a pixel grid


def test_csr_long_rows(amd):
    # a star: every edge ends in node 0 (rows longer than the insertion-sort cutoff)
    rng = np.random.Generator(np.random.PCG64(5))
    keys = np.where(rng.random(5000) < 0.7, 0, rng.integers(0, 7, 5000)).astype(np.int32)
    keys[::11] = -1
    rowptr, perm = amd.plan.csr_from_keys(to_dev(keys), 7)
    valid = np.nonzero(keys >= 0)[0]
    order = valid[np.argsort(keys[valid], kind='stable')]
    assert np.array_equal(perm.cpu().numpy()[:order.size], order)
    assert np.array_equal(rowptr.cpu().numpy(), np.concatenate(([0], np.cumsum(np.bincount(keys[valid], minlength=7)))))
    with pytest.raises(amd.GunetError):
        amd.plan.csr_from_keys(to_dev(np.array([0, 9], np.int32)), 4)


def test_cluster_ids_micro(amd):
    def run(loc):
        pp = amd.plan.PoolPlan(to_dev(np.array(loc), torch.int64),
                               [torch.zeros((0, 2), dtype=torch.int32).cuda()] * 4)
        return pp.cluster.cpu().tolist(), pp.n_pooled, pp.pooled_loc.cpu().tolist()
    full_2x4 = [[r, c] for r in range(2) for c in range(4)]
    assert run(full_2x4) == ([0, 0, 1, 1, 0, 0, 1, 1], 2, [[0, 0], [0, 1]])
    full_4x4 = [[r, c] for r in range(4) for c in range(4)]
    assert run(full_4x4)[0] == [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3]
    # first-occurrence labelling is not sorted by cell id
    assert run([[0, 2], [1, 0], [1, 3]]) == ([0, 1, 0], 2, [[0, 1], [0, 0]])


def _pool_chain_oracle(oracle, node_loc, quads, levels=3):
    out = []
    graph = {'node_loc': node_loc, 'edge_index': list(quads),
             'edges': [np.zeros((q.shape[0], 1), np.float32) for q in quads]}
    x = np.zeros((node_loc.shape[0], 1), np.float32)
    for _ in range(levels):
        ids, p = oracle.cluster_ids_2x2(graph['node_loc'])
        x, graph = oracle.graph_pooling_2x2(x, graph)
        out.append((ids, p, graph['node_loc'], list(graph['edge_index'])))
    return out


@pytest.mark.parametrize('case', ['tiny', 'ragged', 'road', 'dense'])
def test_pool_and_upsample_plans_bit_exact(amd, oracle, case):
    if case == 'dense':
        node_loc, ei, d, _ = oracle.dense_grid_graph(495, 436)
    else:
        node_loc, ei, d, _ = dict(_graphs(oracle))[case]
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    want = _pool_chain_oracle(oracle, node_loc, quads)
    loc = to_dev(node_loc, torch.int64)
    edges = [to_dev(q, torch.int32) for q in quads]
    locs = [loc]
    for lvl, (ids, p, new_loc, new_edges) in enumerate(want):
        pp = amd.plan.get_pool_plan(loc, edges)
        assert pp.n_pooled == p
        assert np.array_equal(pp.cluster.cpu().numpy(), ids), (case, lvl)
        assert np.array_equal(pp.pooled_loc.cpu().numpy(), new_loc), (case, lvl)
        for q in range(4):
            assert np.array_equal(pp.edge_index[q].cpu().numpy(), new_edges[q]), (case, lvl, q)
            # dup_map sends every fine edge to its pooled edge (or -1 for a self-loop)
            dm = pp.dup_map[q].cpu().numpy()
            src = edges[q].cpu().numpy()
            rel = ids[src]
            keep = dm >= 0
            assert np.array_equal(rel[~keep][:, 0], rel[~keep][:, 1])
            assert np.array_equal(new_edges[q][dm[keep]], rel[keep])
            assert np.bincount(new_edges[q][:, 1], minlength=p).max(initial=0) <= 3
        loc, edges = pp.pooled_loc, pp.edge_index
        locs.append(loc)
    if case == 'dense':
        assert [int(l.shape[0]) for l in locs] == [215820, 54064, 13516, 3410]
        assert [int(e.shape[0]) for e in edges] == [9997] * 4
    # upsampling plans back up: parent ids equal the pooling cluster ids of that level
    for lvl in (2, 1, 0):
        up = amd.plan.get_upsample_plan(locs[lvl + 1], locs[lvl])
        coarse, fine = locs[lvl + 1].cpu().numpy(), locs[lvl].cpu().numpy()
        parent, lists = oracle.upsample_edges(coarse, fine)
        assert np.array_equal(up.parent.cpu().numpy(), parent)
        assert np.array_equal(parent, want[lvl][0])
        assert sum(int(e.shape[0]) for e in up.edge_index) == fine.shape[0]
        for q in range(4):
            assert np.array_equal(up.edge_index[q].cpu().numpy(), lists[q]), (case, lvl, q)


def test_remove_duplicate_edges(amd, oracle):
    rng = np.random.Generator(np.random.PCG64(11))
    e = rng.integers(0, 12, (400, 2)).astype(np.int32)
    f = rng.standard_normal((400, 8)).astype(np.float32)
    want_e, want_f = oracle.remove_duplicate_edges(e, 50, f)
    got_e, got_f = amd.hybrid_gnn.remove_duplicate_edges(to_dev(e), 50, to_dev(f))
    assert np.array_equal(got_e.cpu().numpy(), want_e)
    assert np.array_equal(got_f.cpu().numpy(), want_f)      # max is exact in floating point


def test_plan_cache_reuses_structures(amd, oracle):
    node_loc, ei, d, _ = oracle.synth_road_graph(20, 20, 0.5, 1)
    quads = [to_dev(q, torch.int32) for q in oracle.sample_quadrant_subgraphs(ei, d)]
    a = amd.plan.get_quadgraph(quads, node_loc.shape[0])
    b = amd.plan.get_quadgraph(list(quads), node_loc.shape[0])
    assert a is b


def test_pooling_and_collision_hand_derived(amd):
    """The two hand-worked cases of tests/test_oracle.py through the library (no oracle involved):
    dedup-then-self-loop removal with the pre-pooling node count in the key, and a cell-id
    collision on a graph much wider than high, where no fine node gets an upsampling edge."""
    from gunet_amd.layers.hybrid_gnn import graph_pooling_2x2
    e0 = torch.tensor([[0, 1], [1, 2], [0, 2], [2, 3], [3, 1], [1, 2]], dtype=torch.int32).cuda()
    none = torch.zeros((0, 2), dtype=torch.int32).cuda()
    g = {'node_loc': torch.tensor([[0, 0], [0, 1], [0, 2], [1, 2]], dtype=torch.int64).cuda(),
         'edge_index': [e0, none, none.clone(), none.clone()],
         'edges': [torch.tensor([[1.], [5.], [3.], [7.], [2.], [4.]]).cuda()] + [torch.zeros((0, 1)).cuda() for _ in range(3)]}
    x = torch.tensor([[1., -1.], [0., 3.], [2., 2.], [5., -4.]]).cuda()
    xp, g = graph_pooling_2x2(x, g)
    assert xp.cpu().tolist() == [[1., 3.], [5., 2.]]
    assert g['node_loc'].cpu().tolist() == [[0, 0], [0, 1]]
    assert g['edge_index'][0].cpu().tolist() == [[0, 1], [1, 0]] and g['edges'][0][:, 0].cpu().tolist() == [5.0, 2.0]
    fine = torch.tensor([[0, 8], [2, 0], [3, 1]], dtype=torch.int64).cuda()
    pp = amd.plan.PoolPlan(fine, [none] * 4)
    assert pp.cluster.cpu().tolist() == [0, 0, 0] and pp.pooled_loc.cpu().tolist() == [[0, 0]]
    up = amd.plan.UpsamplePlan(pp.pooled_loc, fine)
    assert up.parent.cpu().tolist() == [0, 0, 0] and all(e.shape[0] == 0 for e in up.edge_index)
    fine2 = torch.tensor([[0, 0], [0, 1], [1, 0], [1, 1], [2, 9]], dtype=torch.int64).cuda()
    pp2 = amd.plan.PoolPlan(fine2, [none] * 4)
    assert pp2.cluster.cpu().tolist() == [0, 0, 0, 0, 1]
    up2 = amd.plan.UpsamplePlan(pp2.pooled_loc, fine2)
    assert [e.cpu().tolist() for e in up2.edge_index] == [[[0, 3], [1, 6]], [[0, 5]], [[0, 4]], [[0, 2]]]
