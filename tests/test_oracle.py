"""CPU tests of the oracle: hand-derived known answers, checkpoint shapes, naive cross-checks,
structural invariants and the committed golden vectors."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err


def test_shapes_match_shipped_checkpoint(oracle):
    """Every variable of the reference's checkpoint index has the shape the oracle derives."""
    with open(os.path.join(GOLDEN, 'ckpt_shapes.json')) as f:
        ckpt = {k: tuple(v) for k, v in json.load(f)['shapes'].items()}
    assert ckpt.pop('cnn1/kernel') == (3, 3, 1, 32) and ckpt.pop('cnn2/kernel') == (3, 3, 32, 8)
    mine = oracle.graph_unet_shapes()
    assert mine == ckpt
    assert sum(int(np.prod(s)) for s in ckpt.values()) + 3 * 3 * 32 + 3 * 3 * 32 * 8 == 1948282
    last = 'upsampling_layers/2/gcns/1'
    assert f'{last}/glob_transform/kernel' not in mine and f'{last}/node_transform/kernel' in mine


def test_quadrant_partition_known_answer(oracle):
    ei = np.arange(10, dtype=np.int32).reshape(5, 2)
    q = oracle.sample_quadrant_subgraphs(ei, np.array([0, 1, 2, 8, 4]))
    assert [t.tolist() for t in q] == [[[0, 1], [2, 3], [4, 5]], [[4, 5], [8, 9]], [[8, 9]], [[0, 1]]]


def test_quadrant_partition_naive(oracle):
    rng = np.random.Generator(np.random.PCG64(0))
    d = rng.integers(0, 9, 300)
    ei = rng.integers(0, 50, (300, 2)).astype(np.int32)
    got = oracle.sample_quadrant_subgraphs(ei, d)
    for q in range(4):
        want = [ei[i].tolist() for i in range(300) if d[i] in oracle.QUADRANT_IDS[q]]
        assert got[q].tolist() == want


def test_tf_unique_first_occurrence(oracle):
    y, idx = oracle.tf_unique(np.array([7, 3, 7, 9, 3, 1]))
    assert y.tolist() == [7, 3, 9, 1] and idx.tolist() == [0, 1, 0, 2, 1, 3] and idx.dtype == np.int32
    rng = np.random.Generator(np.random.PCG64(1))
    x = rng.integers(0, 40, 500)
    y, idx = oracle.tf_unique(x)
    seen = {}
    for v in x.tolist():
        seen.setdefault(v, len(seen))
    assert y.tolist() == list(seen) and idx.tolist() == [seen[v] for v in x.tolist()]


def test_segment_ops_naive(oracle):
    rng = np.random.Generator(np.random.PCG64(2))
    data = rng.standard_normal((200, 5)).astype(np.float32)
    ids = rng.integers(0, 17, 200)
    ids[ids == 4] = 5                                  # leave segment 4 empty
    s = oracle.unsorted_segment_sum(data, ids, 17)
    mx = oracle.unsorted_segment_max(data, ids, 17)
    mn = oracle.unsorted_segment_min(data, ids, 17)
    for k in range(17):
        rows = data[ids == k]
        if rows.shape[0] == 0:
            assert np.all(s[k] == 0) and np.all(mx[k] == np.finfo(np.float32).min)
            assert np.all(mn[k] == np.finfo(np.float32).max)
            continue
        acc = np.zeros(5, np.float32)
        for r in rows:
            acc = acc + r                              # in input order
        assert np.array_equal(s[k], acc)
        assert np.array_equal(mx[k], rows.max(0)) and np.array_equal(mn[k], rows.min(0))


def test_cluster_ids_known_answers(oracle):
    full_2x4 = np.array([[r, c] for r in range(2) for c in range(4)])
    ids, p = oracle.cluster_ids_2x2(full_2x4)
    assert ids.tolist() == [0, 0, 1, 1, 0, 0, 1, 1] and p == 2
    full_4x4 = np.array([[r, c] for r in range(4) for c in range(4)])
    assert oracle.cluster_ids_2x2(full_4x4)[0].tolist() == [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3]
    ids, p = oracle.cluster_ids_2x2(np.array([[0, 2], [1, 0], [1, 3]]))
    assert ids.tolist() == [0, 1, 0] and p == 2
    x = np.arange(3, dtype=np.float32)[:, None]
    g = {'node_loc': np.array([[0, 2], [1, 0], [1, 3]]), 'edge_index': [np.zeros((0, 2), np.int32)] * 4,
         'edges': [np.zeros((0, 1), np.float32)] * 4}
    xp, g = oracle.graph_pooling_2x2(x, g)
    assert g['node_loc'].tolist() == [[0, 1], [0, 0]] and xp[:, 0].tolist() == [2.0, 1.0]


def test_remove_duplicate_edges_known_answer(oracle):
    e = np.array([[0, 1], [2, 2], [0, 1], [1, 0], [2, 2], [0, 1]], dtype=np.int32)
    f = np.array([[1.], [9.], [3.], [4.], [5.], [2.]], dtype=np.float32)
    ne, nf = oracle.remove_duplicate_edges(e, 3, f)
    assert ne.tolist() == [[0, 1], [1, 0]] and nf[:, 0].tolist() == [3.0, 4.0] and ne.dtype == np.int32


def test_pooling_dedup_then_selfloop_removal_known_answer(oracle):
    """graph_pooling_2x2 on 4 nodes in two 2x2 cells, worked by hand (hybrid_gnn.py:391-447):
    nodes (0,0) (0,1) (0,2) (1,2): max_row + 1 = 2, pos_id = (r//2)*2 + c//2 = [0, 0, 1, 1] -> clusters [0, 0, 1, 1];
    edges (0,1) (1,2) (0,2) (2,3) (3,1) (1,2) relabel to (0,0) (0,1) (0,1) (1,1) (1,0) (0,1);
    keys s*4 + r (4 = node count BEFORE pooling) = 0 1 1 5 4 1 -> first occurrences 0 1 5 4 ->
    (0,0) (0,1) (1,1) (1,0) with feature maxima 1, max(5,3,4) = 5, 7, 2; only then the self-loops go:
    (0,1) -> 5 and (1,0) -> 2, in that order."""
    g = {'node_loc': np.array([[0, 0], [0, 1], [0, 2], [1, 2]]),
         'edge_index': [np.array([[0, 1], [1, 2], [0, 2], [2, 3], [3, 1], [1, 2]], np.int32)] + [np.zeros((0, 2), np.int32)] * 3,
         'edges': [np.array([[1.], [5.], [3.], [7.], [2.], [4.]], np.float32)] + [np.zeros((0, 1), np.float32)] * 3}
    x = np.array([[1., -1.], [0., 3.], [2., 2.], [5., -4.]], np.float32)
    xp, g = oracle.graph_pooling_2x2(x, g)
    assert xp.tolist() == [[1., 3.], [5., 2.]]
    assert g['node_loc'].tolist() == [[0, 0], [0, 1]]
    assert g['edge_index'][0].tolist() == [[0, 1], [1, 0]] and g['edges'][0][:, 0].tolist() == [5.0, 2.0]
    assert all(g['edge_index'][q].shape == (0, 2) for q in (1, 2, 3))


def test_collided_upsampling_cell_known_answer(oracle):
    """The reference's cell id (row//2) * (max_row + 1) + col//2 merges distinct cells on graphs much
    wider than high (hybrid_gnn.py:417-423).  Nodes (0,8) (2,0) (3,1): max_row + 1 = 4, cells (0,4),
    (1,0), (1,0) all get id 4 -> one cluster; its position is the column-wise minimum (0,0).  On the
    way up (:336-357) the parent sits at 2*(0,0) + 0.5 and the children at offsets (-0.5, 7.5),
    (1.5, -0.5), (2.5, 0.5): none is one of the four (+-0.5, +-0.5) places, so no child gets an edge."""
    fine = np.array([[0, 8], [2, 0], [3, 1]])
    ids, p = oracle.cluster_ids_2x2(fine)
    assert ids.tolist() == [0, 0, 0] and p == 1
    coarse = oracle.unsorted_segment_min(fine // 2, ids, p)
    assert coarse.tolist() == [[0, 0]]
    parent, lists = oracle.upsample_edges(coarse, fine)
    assert parent.tolist() == [0, 0, 0] and all(l.shape[0] == 0 for l in lists)
    # a 4x4 block far enough left keeps its structure next to the collision: (2,0) alone in its cell
    fine2 = np.array([[0, 0], [0, 1], [1, 0], [1, 1], [2, 9]])
    ids2, p2 = oracle.cluster_ids_2x2(fine2)      # max_row + 1 = 3: cells 0, 0, 0, 0 and 1*3 + 4 = 7
    assert ids2.tolist() == [0, 0, 0, 0, 1] and p2 == 2
    coarse2 = oracle.unsorted_segment_min(fine2 // 2, ids2, p2)
    parent2, lists2 = oracle.upsample_edges(coarse2, fine2)
    # children of cell 0: (0,0) -> q3, (0,1) -> q0, (1,1) -> q1, (1,0) -> q2; (2,9) is (even, odd) -> q0
    assert [l.tolist() for l in lists2] == [[[0, 3], [1, 6]], [[0, 5]], [[0, 4]], [[0, 2]]]


def test_column_sum_choice_bracketed_by_float32_orders(oracle):
    """The oracle's reduce_sum over nodes is the exact sum rounded once.  TensorFlow adds in float32 in
    an order the reference does not fix; the two extremes -- strictly sequential and pairwise -- on a
    tensor of the full graph's size both land within 1e-5 (relative) of the oracle's value, so the
    order-free choice cannot move a feature by more than the parity tolerance."""
    rng = np.random.Generator(np.random.PCG64(9))
    x = np.maximum(rng.standard_normal((96493, 128)).astype(np.float32) * np.float32(0.5) + np.float32(0.2), 0)
    want = oracle.column_sum(x)[0].astype(np.float64)
    seq = np.cumsum(x, axis=0, dtype=np.float32)[-1].astype(np.float64)      # ((x0 + x1) + x2) + ...
    t = x
    while t.shape[0] > 1:                                                     # balanced tree, float32 at every node
        if t.shape[0] % 2:
            t = np.concatenate((t, np.zeros((1, t.shape[1]), np.float32)))
        t = t[0::2] + t[1::2]
    pair = t[0].astype(np.float64)
    err_seq = np.max(np.abs(seq - want) / np.abs(want))
    err_pair = np.max(np.abs(pair - want) / np.abs(want))
    assert err_pair <= 2e-7, err_pair
    assert err_seq <= 1e-5, err_seq
    # and the float32 value of the exact sum is within half an ulp of it
    assert np.max(np.abs(want - np.sum(x.astype(np.float64), axis=0)) / np.abs(want)) <= 2.0 ** -24


def test_upsample_child_quadrants(oracle):
    # coarse node (k, j): children (2k,2j)->q3, (2k,2j+1)->q0, (2k+1,2j+1)->q1, (2k+1,2j)->q2
    fine = np.array([[r, c] for r in range(4) for c in range(4)])
    ids, p = oracle.cluster_ids_2x2(fine)
    coarse = oracle.unsorted_segment_min(fine // 2, ids, p)
    parent, lists = oracle.upsample_edges(coarse, fine)
    assert np.array_equal(parent, ids)
    for q, (dr, dc) in enumerate([(0, 1), (1, 1), (1, 0), (0, 0)]):
        kids = lists[q][:, 1] - p
        assert np.all(fine[kids, 0] % 2 == dr) and np.all(fine[kids, 1] % 2 == dc)
        assert np.array_equal(lists[q][:, 0], ids[kids])
    assert sum(l.shape[0] for l in lists) == 16


def test_dense_grid_counts(oracle):
    """495x436 all-pixel grid: level sizes and per-quadrant edge counts derived by hand."""
    node_loc, ei, d, mask = oracle.dense_grid_graph(495, 436)
    assert node_loc.shape[0] == 215820 and ei.shape[0] == 1936798 and mask.all()
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    assert [q.shape[0] for q in quads] == [645599] * 4
    both = np.concatenate(quads)
    assert np.unique(both[:, 0].astype(np.int64) * 215820 + both[:, 1]).size == 1720978
    g = {'node_loc': node_loc, 'edge_index': list(quads), 'edges': [np.zeros((q.shape[0], 1), np.float32) for q in quads]}
    x = np.zeros((215820, 1), np.float32)
    sizes, ecounts = [], []
    for _ in range(3):
        x, g = oracle.graph_pooling_2x2(x, g)
        sizes.append(x.shape[0])
        ecounts.append([e.shape[0] for e in g['edge_index']])
        for q in range(4):                             # quadrant in-degree stays <= 3 at every level
            assert np.bincount(g['edge_index'][q][:, 1]).max() <= 3
    assert sizes == [54064, 13516, 3410]
    assert ecounts == [[161261] * 4, [40083] * 4, [9997] * 4]


def test_image_to_graph_semantics(oracle):
    mask = np.zeros((4, 5), bool)
    mask[1, 1] = mask[1, 2] = mask[2, 2] = mask[0, 4] = True    # (0,4) is isolated: no edges, no node
    node_loc, ei, d, node_mask = oracle.image_to_graph(oracle.directional_maps_from_mask(mask))
    assert node_loc.tolist() == [[1, 1], [1, 2], [2, 2]] and not node_mask[0, 4]
    # sorted by (source node, direction), self-loops carry direction 8
    assert list(zip(ei[:, 0].tolist(), ei[:, 1].tolist(), d.tolist())) == [
        (0, 1, 2), (0, 2, 3), (0, 0, 8), (1, 2, 4), (1, 0, 6), (1, 1, 8), (2, 1, 0), (2, 0, 7), (2, 2, 8)]


def test_gn_layer_widths_and_chained_global(oracle):
    """The layer consumes exactly the checkpoint's kernel shapes; the per-direction global
    transforms are chained (the second one sees the first one's 32-wide output)."""
    node_loc, ei, d, _ = oracle.synth_road_graph(8, 9, 0.7, 1)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    rng = np.random.Generator(np.random.PCG64(3))
    n = node_loc.shape[0]
    W = oracle.make_weights(oracle.gn_layer_shapes('L', 128, 16, 113), 4)
    graph = {'edge_index': list(quads), 'node_loc': node_loc,
             'edges': [rng.random((q.shape[0], 16)).astype(np.float32) for q in quads],
             'global': rng.random((1, 113)).astype(np.float32)}
    x = rng.random((n, 128)).astype(np.float32)
    keep = {}
    g0 = graph['global']
    edges_in = list(graph['edges'])
    y, g = oracle.geo_quadrant_gcn(x, graph, W, 'L', keep=keep)
    assert y.shape == (n, 128) and g['global'].shape == (1, 128)
    assert [e.shape for e in g['edges']] == [(q.shape[0], 32) for q in quads]
    glob = g0
    for q in range(4):
        glob = np.maximum(glob @ W[f'L/glob_transforms/{q}/kernel'] + W[f'L/glob_transforms/{q}/bias'], 0)
        assert rel_err(keep[f'glob{q}'], glob) < 1e-6
    # one message and its aggregation, restated in float64
    q, e = 2, 5
    s, r = quads[q][e]
    xq = keep[f'x_quad{q}'].astype(np.float64)
    row = np.concatenate([edges_in[q][e], xq[s], xq[r], keep[f'glob{q}'][0]])
    msg = np.maximum(row @ W[f'L/edge_transforms/{q}/kernel'] + W[f'L/edge_transforms/{q}/bias'], 0)
    assert rel_err(g['edges'][q][e], msg) < 1e-5
    into_r = [i for i in range(quads[q].shape[0]) if quads[q][i, 1] == r]
    want = 0.25 * np.sum([g['edges'][q][i] for i in into_r], axis=0)
    assert rel_err(keep['agg'][r, 32 * q:32 * q + 32], want) < 1e-6


@pytest.mark.parametrize('name', ['unet_9x14', 'unet_24x30'])
def test_golden_unet(oracle, name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    node_loc, ei, d, _ = oracle.synth_road_graph(int(z['height']), int(z['width']), float(z['density']),
                                                 int(z['graph_seed']))
    assert np.array_equal(node_loc, z['node_loc']) and np.array_equal(ei, z['edge_index'])
    quads = [z[f'quad{q}'] for q in range(4)]
    for q, got in enumerate(oracle.sample_quadrant_subgraphs(ei, d)):
        assert np.array_equal(got, quads[q])
    W = oracle.make_weights(oracle.graph_unet_shapes(), int(z['weights_seed']))
    graph = {'edge_index': quads, 'edges': [z[f'edges_in{q}'] for q in range(4)], 'node_loc': z['node_loc'],
             'global': z['global']}
    y, g = oracle.graph_unet_forward(z['x'], graph, W)
    assert rel_err(y, z['y']) < 1e-6 and rel_err(g['global'], z['global_out']) < 1e-6
    gi = {'node_loc': z['node_loc'], 'edge_index': list(quads),
          'edges': [np.zeros((q.shape[0], 1), np.float32) for q in quads]}
    xx = np.zeros((z['node_loc'].shape[0], 1), np.float32)
    for lvl in range(3):
        ids, _ = oracle.cluster_ids_2x2(gi['node_loc'])
        xx, gi = oracle.graph_pooling_2x2(xx, gi)
        assert np.array_equal(ids, z[f'cluster{lvl}']) and np.array_equal(gi['node_loc'], z[f'pooled_loc{lvl}'])
        for q in range(4):
            assert np.array_equal(gi['edge_index'][q], z[f'pooled_edges{lvl}_{q}'])


def test_fp32_oracle_close_to_float64_restatement(oracle):
    """The fp32 oracle agrees with the same computation in float64 far inside the 1e-5 gate."""
    z = np.load(os.path.join(GOLDEN, 'unet_24x30.npz'))
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    W64 = {k: np.asarray(v, np.float64) for k, v in W.items()}
    quads = [z[f'quad{q}'] for q in range(4)]
    graph = {'edge_index': quads, 'edges': [z[f'edges_in{q}'].astype(np.float64) for q in range(4)],
             'node_loc': z['node_loc'], 'global': z['global'].astype(np.float64)}
    y64, _ = oracle.graph_unet_forward(z['x'].astype(np.float64), graph, W64)
    assert y64.dtype == np.float64
    assert rel_err(z['y'], y64) < 2e-6


def test_synthetic_generators_match_oracle(oracle):
    """The package's workload generators (used by bench.py) produce exactly the oracle's data."""
    from gunet_amd import synthetic
    for h, w, dens, seed in [(9, 14, 0.6, 3), (60, 45, 0.45, 0), (30, 30, 1.1, 1)]:
        node_loc, ei, d, _ = oracle.synth_road_graph(h, w, dens, seed)
        nl2, ei2, d2 = synthetic.road_graph(h, w, dens, seed)
        assert np.array_equal(node_loc, nl2) and np.array_equal(ei, ei2) and np.array_equal(d, d2)
        assert ei2.dtype == np.int32 and nl2.dtype == np.int64 and d2.dtype == np.int64
    assert synthetic.graph_unet_shapes() == oracle.graph_unet_shapes()
    W1, W2 = oracle.make_weights(oracle.graph_unet_shapes(), 1), synthetic.random_weights(synthetic.graph_unet_shapes(), 1)
    assert all(np.array_equal(W1[k], W2[k]) for k in W1)
    node_loc, ei, d, _ = oracle.synth_road_graph(20, 20, 0.5, 2)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    enc = oracle.synth_street_encoding(node_loc.shape[0], 5)
    assert np.array_equal(enc, synthetic.street_encoding(node_loc.shape[0], 5))
    sn, te, wd = oracle.synth_sample(node_loc, quads, 7, time_idx=24, weekday=3)
    sn2, te2, wd2 = synthetic.sample_inputs(node_loc.shape[0], 7, time_idx=24, weekday=3)
    assert np.array_equal(sn, sn2) and np.array_equal(te, te2) and np.array_equal(wd, wd2)
    x, graph = oracle.assemble_inputs(sn, enc, quads, node_loc, te, wd)
    x2, edges2, glob2 = synthetic.assemble_model_inputs(sn2, enc, quads, te2, wd2)
    assert np.array_equal(x, x2) and np.array_equal(graph['global'], glob2)
    assert all(np.array_equal(a, b) for a, b in zip(graph['edges'], edges2))


def test_street_cnn_naive(oracle):
    """conv2d_same / forward_street_encoding against element-by-element loops (tf.keras Conv2D:
    cross-correlation, 'SAME' zero padding, kernel [kh, kw, in, out])."""
    rng = np.random.Generator(np.random.PCG64(5))
    h, w = 6, 7
    smap = rng.integers(0, 256, (h, w)).astype(np.uint8)
    W = oracle.make_weights(oracle.cnn_shapes(), 2)
    k1, k2 = W['cnn1/kernel'].astype(np.float64), W['cnn2/kernel'].astype(np.float64)
    img = smap.astype(np.float64) / 255.0

    def conv(x, k):
        out = np.zeros((h, w, k.shape[3]))
        for y in range(h):
            for xx in range(w):
                for ky in range(3):
                    for kx in range(3):
                        yy, xc = y + ky - 1, xx + kx - 1
                        if 0 <= yy < h and 0 <= xc < w:
                            out[y, xx] += x[yy, xc] @ k[ky, kx]
        return out

    want = np.maximum(conv(np.maximum(conv(img[..., None], k1), 0), k2), 0)
    loc = np.array([[0, 0], [5, 6], [2, 3], [0, 6]])
    got = oracle.forward_street_encoding(smap, W, loc)
    assert got.shape == (4, 8) and got.dtype == np.float32
    assert np.abs(got - want[loc[:, 0], loc[:, 1]]).max() <= 1e-6 * max(1.0, np.abs(want).max())
    enc = rng.random((9, 8)).astype(np.float32)
    e = np.array([[0, 3], [8, 8], [2, 1]], dtype=np.int32)
    feats = oracle.compute_edge_features(enc, e)
    assert feats.shape == (3, 16) and np.array_equal(feats[0], np.concatenate((enc[0], enc[3])))


def test_prediction_image_and_metrics_known_answers(oracle):
    """Hand-derived values for the post-processing restatements (submission.py:69-94,
    evaluation.py:173-244)."""
    mask = np.array([[True, False], [False, True]])
    pred = np.zeros((2, 96), dtype=np.float32)
    pred[0, 0] = 0.5 / 255      # tie -> even -> 0
    pred[0, 1] = 1.5 / 255      # tie -> even -> 2
    pred[0, 8 * 5 + 3] = 0.4    # step 5 is kept as index 3 -> 102
    pred[1, 8 * 3] = 1.0        # step 3 is dropped
    pred[1, 8 * 11 + 7] = 2.0   # clipped to 255
    img = oracle.predict_sample_image(pred, mask)
    assert img.shape == (6, 2, 2, 8) and img.dtype == np.uint8
    assert img[0, 0, 0, 0] == 0 and img[0, 0, 0, 1] == 2 and img[3, 0, 0, 3] == 102
    assert img[5, 1, 1, 7] == 255 and img[:, 0, 1].max() == 0 and img[:, 1, 0].max() == 0
    assert int(img.astype(np.int64).sum()) == 2 + 102 + 255
    t = np.zeros((2, 96), dtype=np.float32)
    p = np.zeros((2, 96), dtype=np.float32)
    p[0, 1] = 1.0               # speed column of the first step
    p[1, 8 * 11] = 2.0          # volume column of the last step
    m = oracle.compute_metrics(t, p)
    assert abs(m['mse'] - 5.0 / 192) < 1e-12 and abs(m['mae'] - 3.0 / 192) < 1e-12
    assert abs(m['mse_scaled'] - 255.0 ** 2 * 5.0 / 192) < 1e-6
    assert abs(m['mae_scaled_speed'] - 255.0 / 8) < 1e-9        # columns 1,3,5,7 of 2 rows
    assert m['mae_scaled_volume'] == 0.0                         # columns 0,2,4,6: the last step is not in them
    assert abs(m['mse_scaled_5_min'] - 255.0 ** 2 / 16) < 1e-6 and abs(m['mse_scaled_60_min'] - 255.0 ** 2 * 4 / 16) < 1e-6
    assert m['mse_scaled_30_min'] == 0.0 and abs(m['rmse_scaled'] - np.sqrt(m['mse_scaled'])) < 1e-12
    full = oracle.model_compute_metrics(np.ones_like(t), t, p, 10, 10)
    assert abs(full['score'] - m['mse_scaled'] * 2 / 100) < 1e-9
    assert abs(full['rel_score'] - (255.0 ** 2) / m['mse_scaled']) < 1e-9
