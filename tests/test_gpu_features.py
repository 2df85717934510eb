"""Feature parity of the HIP kernels with the oracle, through the C ABI.

Tolerance (fp32 path): normwise relative error max|got - want| / max|want| <= 1e-5 for every
feature tensor; index tensors and max-pooled values must be bit-exact."""
import numpy as np
import pytest
import torch

from conftest import assert_close, rel_err
from gpu_helpers import copy_graph, graph_to_dev, make_case, to_dev

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope='module')
def amd():
    import gunet_amd
    return gunet_amd


def _layer_weights(oracle, prefix, c_in, f_e, g, seed=1, with_gt=True):
    return oracle.make_weights(oracle.gn_layer_shapes(prefix, c_in, f_e, g, 128, with_gt), seed)


def _run_layer(amd, oracle, x, graph, W, prefix, with_gt=True, batch=None, activation='relu', use_bias=True):
    layer = amd.hybrid_gnn.GeoQuadrantGCN(128, activation=activation, use_bias=use_bias, use_global=True,
                                          learn_global_scale=with_gt)
    layer.load_weights(W, prefix, 'cuda')
    g = graph_to_dev(graph)
    xt = to_dev(x)
    if batch:
        xt = xt.unsqueeze(0).repeat(batch, 1, 1)
        g['edges'] = [e.unsqueeze(0).repeat(batch, 1, 1) for e in g['edges']]
        g['global'] = g['global'].repeat(batch, 1)
    y, g = layer(xt, g)
    torch.cuda.synchronize()
    return y, g


def test_dense_forward(amd):
    rng = np.random.Generator(np.random.PCG64(2))
    for rows, k, n, bias, act in [(1000, 104, 128, True, None), (333, 128, 96, False, None),
                                  (257, 64, 256, True, 'relu'), (50, 10, 7, True, 'relu'), (0, 8, 32, True, None)]:
        x = rng.standard_normal((rows, k)).astype(np.float32)
        w = rng.standard_normal((k, n)).astype(np.float32) * 0.1
        b = rng.standard_normal(n).astype(np.float32)
        layer = amd.hybrid_gnn.Dense(n, activation=act, use_bias=bias)
        layer.load_weights({'d/kernel': w, 'd/bias': b}, 'd', 'cuda')
        got = layer(to_dev(x)).cpu().numpy()
        want = x.astype(np.float64) @ w.astype(np.float64) + (b if bias else 0)
        if act == 'relu':
            want = np.maximum(want, 0)
        assert got.shape == (rows, n)
        if rows:
            assert_close(got, want, TOL, str((rows, k, n)))


def test_dense_forward_operand_beyond_4_gib(amd):
    """An operand larger than the 2 GiB the tiled kernel addresses with 32-bit offsets is worked off in row
    chunks; checked on rows on both sides of the chunk boundary and at the end (float64 products of
    those rows as the reference)."""
    rows, k, n = 600_000, 2048, 128   # 4.9 GB of float32 input
    gen = torch.Generator(device='cuda').manual_seed(5)
    x = torch.empty((rows, k), device='cuda', dtype=torch.float32)
    for r0 in range(0, rows, 100_000):   # filled piecewise: no second 4.9 GB temporary
        x[r0:r0 + 100_000].normal_(generator=gen)
    w = torch.randn((k, n), device='cuda', generator=gen) * 0.05
    b = torch.randn((n,), device='cuda', generator=gen)
    layer = amd.hybrid_gnn.Dense(n, activation='relu', use_bias=True)
    layer.load_weights({'d/kernel': w.cpu().numpy(), 'd/bias': b.cpu().numpy()}, 'd', 'cuda')
    got = layer(x)
    torch.cuda.synchronize()
    assert got.shape == (rows, n)
    chunk = (1 << 30) // (k * 4) // 128 * 128   # rows per chunk of the library's split
    picks = torch.tensor(sorted({0, 1, 127, 128, chunk - 1, chunk, chunk + 1, 2 * chunk - 1, 2 * chunk, rows - 129, rows - 1}),
                         device='cuda')
    want = torch.relu(x[picks].double() @ w.double() + b.double()).cpu().numpy()
    assert_close(got[picks].cpu().numpy(), want, TOL, 'rows around the chunk boundaries')
    del x, got


def test_get_sender_receiver(amd, oracle):
    rng = np.random.Generator(np.random.PCG64(3))
    for width in (32, 7):
        x = rng.standard_normal((200, width)).astype(np.float32)
        e = rng.integers(0, 200, (999, 2)).astype(np.int32)
        s, r = amd.graph_ops.get_sender_receiver(to_dev(x), to_dev(e))
        ws, wr = oracle.get_sender_receiver(x, e)
        assert np.array_equal(s.cpu().numpy(), ws) and np.array_equal(r.cpu().numpy(), wr)


@pytest.mark.parametrize('shape', [(128, 16, 113), (128, 32, 128), (256, 16, 128), (128, 2, 256), (104, 5, 20)])
def test_gn_layer_small(amd, oracle, shape):
    c_in, f_e, g = shape
    node_loc, ei, d, _ = oracle.synth_road_graph(23, 31, 0.55, 2)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    rng = np.random.Generator(np.random.PCG64(9))
    n = node_loc.shape[0]
    x = rng.random((n, c_in)).astype(np.float32)
    graph = {'edge_index': list(quads), 'node_loc': node_loc,
             'edges': [rng.random((q.shape[0], f_e)).astype(np.float32) for q in quads],
             'global': rng.random((1, g)).astype(np.float32)}
    W = _layer_weights(oracle, 'L', c_in, f_e, g)
    keep = {}
    want_x, want_g = oracle.geo_quadrant_gcn(x, copy_graph(graph), W, 'L', keep=keep)
    got_x, got_g = _run_layer(amd, oracle, x, graph, W, 'L')
    assert_close(got_x.cpu().numpy(), want_x, TOL)
    for q in range(4):
        assert_close(got_g['edges'][q].cpu().numpy(), want_g['edges'][q], TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)


def test_gn_layer_variants(amd, oracle):
    """identity global transform (last layer), linear activation, no bias, batched call."""
    node_loc, ei, d, _ = oracle.synth_road_graph(17, 19, 0.6, 6)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    rng = np.random.Generator(np.random.PCG64(10))
    n = node_loc.shape[0]
    x = (rng.random((n, 128)) - 0.3).astype(np.float32)
    graph = {'edge_index': list(quads), 'node_loc': node_loc,
             'edges': [(rng.random((q.shape[0], 32)) - 0.5).astype(np.float32) for q in quads],
             'global': rng.random((1, 128)).astype(np.float32)}
    W = _layer_weights(oracle, 'L', 128, 32, 128, with_gt=False)
    want_x, want_g = oracle.geo_quadrant_gcn(x, copy_graph(graph), W, 'L')
    got_x, got_g = _run_layer(amd, oracle, x, graph, W, 'L', with_gt=False)
    assert want_g['global'].shape == (1, 288) and tuple(got_g['global'].shape) == (1, 288)
    assert_close(got_x.cpu().numpy(), want_x, TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)

    W = _layer_weights(oracle, 'L', 128, 32, 128)
    for act, bias in ((None, True), ('relu', False)):
        want_x, want_g = oracle.geo_quadrant_gcn(x, copy_graph(graph), W, 'L', activation=act, use_bias=bias)
        got_x, got_g = _run_layer(amd, oracle, x, graph, W, 'L', activation=act, use_bias=bias)
        assert_close(got_x.cpu().numpy(), want_x, TOL, str((act, bias)))
        assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL, str((act, bias)))
        for q in range(4):
            assert_close(got_g['edges'][q].cpu().numpy(), want_g['edges'][q], TOL)

    got_b, got_gb = _run_layer(amd, oracle, x, graph, W, 'L', batch=3, use_bias=False)
    for b in range(3):                      # every sample of a batch equals the single-sample call, bit for bit
        assert torch.equal(got_b[b], got_x)
        assert torch.equal(got_gb['global'][b], got_g['global'][0])


def test_gn_layer_errors(amd, oracle):
    with pytest.raises(AttributeError):
        amd.hybrid_gnn.GeoQuadrantGCN(128, combination='additive')
    with pytest.raises(NotImplementedError):
        amd.layers.get('no_such_layer')
    layer = amd.hybrid_gnn.GeoQuadrantGCN(128, activation='relu', use_bias=True, use_global=True)
    x = torch.zeros((4, 128), device='cuda')
    e = [torch.zeros((0, 2), dtype=torch.int32, device='cuda')] * 4
    with pytest.raises(KeyError):
        layer(x, {'edge_index': e, 'edges': [torch.zeros((0, 32), device='cuda')] * 4})
    up = amd.hybrid_gnn.GraphUpsampling2x2(128, method='nearest')
    with pytest.raises(NotImplementedError):
        up(x, {}, {})


def test_gn_layer_config1_road_graph(amd, oracle):
    """Single HybridGNN block, 4 quadrant subgraphs, 495x436 road graph, batch 1, D0 shapes."""
    node_loc, ei, d, _ = oracle.synth_road_graph(495, 436, 0.45, 0)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    rng = np.random.Generator(np.random.PCG64(1))
    n = node_loc.shape[0]
    x = rng.random((n, 128)).astype(np.float32)
    graph = {'edge_index': list(quads), 'node_loc': node_loc,
             'edges': [rng.random((q.shape[0], 16)).astype(np.float32) for q in quads],
             'global': rng.random((1, 113)).astype(np.float32)}
    W = _layer_weights(oracle, 'L', 128, 16, 113)
    keep = {}
    want_x, want_g = oracle.geo_quadrant_gcn(x, copy_graph(graph), W, 'L', keep=keep)
    got_x, got_g = _run_layer(amd, oracle, x, graph, W, 'L')
    assert_close(got_x.cpu().numpy(), want_x, TOL)
    for q in range(4):
        assert_close(got_g['edges'][q].cpu().numpy(), want_g['edges'][q], TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)
    # run twice: all reductions are ordered, so the result is bitwise reproducible
    again_x, again_g = _run_layer(amd, oracle, x, graph, W, 'L')
    assert torch.equal(again_x, got_x) and torch.equal(again_g['global'], got_g['global'])


def test_pooling_features_bit_exact(amd, oracle):
    """Pool / upsample chain (levels 0->1->2->3): max-pooled features are exact."""
    node_loc, ei, d, _ = oracle.synth_road_graph(61, 47, 0.5, 8)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    rng = np.random.Generator(np.random.PCG64(12))
    batch = 4
    x = rng.standard_normal((batch, node_loc.shape[0], 128)).astype(np.float32)
    feats = [rng.standard_normal((batch, q.shape[0], 32)).astype(np.float32) for q in quads]
    g = {'edge_index': [to_dev(q, torch.int32) for q in quads], 'node_loc': to_dev(node_loc, torch.int64),
         'edges': [to_dev(f) for f in feats]}
    xt = to_dev(x)
    want = []
    for b in range(batch):
        gb = {'edge_index': list(quads), 'node_loc': node_loc, 'edges': [f[b] for f in feats]}
        xb = x[b]
        per = []
        for _ in range(3):
            xb, gb = oracle.graph_pooling_2x2(xb, gb)
            per.append((xb, [e for e in gb['edges']]))
        want.append(per)
    for lvl in range(3):
        xt, g = amd.hybrid_gnn.graph_pooling_2x2(xt, g)
        for b in range(batch):
            assert np.array_equal(xt[b].cpu().numpy(), want[b][lvl][0])
            for q in range(4):
                assert np.array_equal(g['edges'][q][b].cpu().numpy(), want[b][lvl][1][q])


def _unet(amd, oracle, W):
    model = amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                          use_bias=True, use_global=True, output_activation=None)
    return model.load_weights(W, 'cuda')


@pytest.mark.parametrize('size', [(12, 16, 0.6), (40, 52, 0.45)])
def test_graph_unet_small(amd, oracle, size):
    x, graph, _ = make_case(oracle, *size)
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    keep = {}
    want_y, want_g = oracle.graph_unet_forward(x, copy_graph(graph), W, keep=keep)
    model = _unet(amd, oracle, W)
    got_y, got_g = model(to_dev(x), graph_to_dev(graph))
    assert tuple(got_y.shape) == want_y.shape
    assert_close(got_y.cpu().numpy(), want_y, TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)
    for q in range(4):
        assert np.array_equal(got_g['edge_index'][q].cpu().numpy(), want_g['edge_index'][q])
        assert_close(got_g['edges'][q].cpu().numpy(), want_g['edges'][q], TOL)
    assert np.array_equal(got_g['node_loc'].cpu().numpy(), want_g['node_loc'])


def test_graph_unet_config2_road_graph(amd, oracle):
    """Full GraphUNet at checkpoint shapes on the 495x436 road graph; batch of 2 samples checked
    against two single-sample oracle runs."""
    node_loc, ei, d, _ = oracle.synth_road_graph(495, 436, 0.45, 0)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    enc = oracle.synth_street_encoding(node_loc.shape[0], 5)
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    xs, graphs = [], []
    for b in range(2):
        sn, te, wd = oracle.synth_sample(node_loc, quads, 100 + b, time_idx=12 * b, weekday=b)
        x, graph = oracle.assemble_inputs(sn, enc, quads, node_loc, te, wd)
        xs.append(x)
        graphs.append(graph)
    model = _unet(amd, oracle, W)
    g = graph_to_dev(graphs[0])
    g['edges'] = [torch.stack([to_dev(gr['edges'][q]) for gr in graphs]) for q in range(4)]
    g['global'] = torch.cat([to_dev(gr['global']) for gr in graphs], dim=0)
    got_y, got_g = model(torch.stack([to_dev(x) for x in xs]), g)
    got_y = torch.clamp_min(got_y, 0).cpu().numpy()
    for b in range(2):
        want_y, want_g = oracle.predict(xs[b], copy_graph(graphs[b]), W)
        assert_close(got_y[b], want_y, TOL, str(b))
        assert_close(got_g['global'][b:b + 1].cpu().numpy(), want_g['global'], TOL)


def test_graph_upsampling_layer(amd, oracle):
    """GraphUpsampling2x2 (method 'graph') on its own: the zero-row-free kernel path against the
    oracle's GN layer on [x ; zeros], single sample and batched."""
    node_loc, ei, d, _ = oracle.synth_road_graph(33, 41, 0.5, 13)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    rng = np.random.Generator(np.random.PCG64(14))
    fine = {'node_loc': node_loc, 'edge_index': list(quads),
            'edges': [rng.random((q.shape[0], 16)).astype(np.float32) for q in quads],
            'global': rng.random((1, 113)).astype(np.float32)}
    ids, p = oracle.cluster_ids_2x2(node_loc)
    _, coarse = oracle.graph_pooling_2x2(np.zeros((node_loc.shape[0], 1), np.float32), copy_graph(fine))
    coarse['edges'] = [rng.random((e.shape[0], 32)).astype(np.float32) for e in coarse['edge_index']]
    coarse['global'] = rng.random((1, 128)).astype(np.float32)
    x = (rng.random((p, 128)) - 0.2).astype(np.float32)
    W = oracle.make_weights(oracle.gn_layer_shapes('U/gcn', 128, 2, 128 + 113), 3)
    want_x, want_g = oracle.graph_upsampling_graph(x, copy_graph(coarse), fine, W, 'U')
    up = amd.hybrid_gnn.GraphUpsampling2x2(128, method='graph', layer_type='geo_quadrant_gcn', use_global=True,
                                           use_bias=True, activation='relu')
    up.gcn.load_weights(W, 'U/gcn', 'cuda')
    got_x, got_g = up(to_dev(x), graph_to_dev(coarse), graph_to_dev(fine))
    assert tuple(got_x.shape) == want_x.shape
    assert_close(got_x.cpu().numpy(), want_x, TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)
    for q in range(4):                                  # the target graph's lists and features come back
        assert np.array_equal(got_g['edge_index'][q].cpu().numpy(), fine['edge_index'][q])
        assert np.array_equal(got_g['edges'][q].cpu().numpy(), fine['edges'][q])
    # generic path (the layer on [x ; zeros] with explicit zero edge features) gives the same numbers
    upg = amd.plan.get_upsample_plan(to_dev(coarse['node_loc'], torch.int64), to_dev(node_loc, torch.int64))
    xz = torch.cat((to_dev(x), torch.zeros((node_loc.shape[0], 128), device='cuda')))
    gg = {'edge_index': upg.edge_index, 'edges': [torch.zeros((e.shape[0], 2), device='cuda') for e in upg.edge_index],
          'global': torch.cat((to_dev(coarse['global']), to_dev(fine['global'])), dim=-1)}
    gen_x, gen_g = up.gcn(xz, gg)
    assert rel_err(got_x.cpu().numpy(), gen_x[p:].cpu().numpy()) <= 2e-6
    assert rel_err(got_g['global'].cpu().numpy(), gen_g['global'].cpu().numpy()) <= 2e-6
    # batched
    cb, fb = graph_to_dev(coarse), graph_to_dev(fine)
    cb['global'] = cb['global'].repeat(3, 1)
    fb['global'] = fb['global'].repeat(3, 1)
    bx, bg = up(to_dev(x).unsqueeze(0).repeat(3, 1, 1), cb, fb)
    for b in range(3):
        assert torch.equal(bx[b], got_x) and torch.equal(bg['global'][b], got_g['global'][0])


@pytest.mark.parametrize('mode,tol', [('split3', 1e-5), ('bf16', 2e-2)])
def test_matmul_modes(amd, oracle, mode, tol):
    """The bf16-matrix-core modes: 'split3' (three bf16 pieces per operand, six products) holds the
    float32 gate of 1e-5, 'bf16' (one product; on this checkpoint-shaped model the intermediates of the
    whole-model call are kept in bf16 as well) the bf16 gate of 2e-2."""
    x, graph, _ = make_case(oracle, 40, 52, 0.45)
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    want_y, want_g = oracle.graph_unet_forward(x, copy_graph(graph), W)
    assert amd.get_matmul_mode() == 'f32'
    amd.set_matmul_mode(mode)
    try:
        assert amd.get_matmul_mode() == mode
        model = _unet(amd, oracle, W)
        got_y, got_g = model(to_dev(x), graph_to_dev(graph))
        got_y, got_glob = got_y.cpu().numpy(), got_g['global'].cpu().numpy()
        rng = np.random.Generator(np.random.PCG64(4))
        xd = rng.standard_normal((3000, 104)).astype(np.float32)
        wd = (rng.standard_normal((104, 96)) * 0.1).astype(np.float32)
        layer = amd.hybrid_gnn.Dense(96, use_bias=False)
        layer.load_weights({'d/kernel': wd}, 'd', 'cuda')
        got_d = layer(to_dev(xd)).cpu().numpy()
    finally:
        amd.set_matmul_mode('f32')
    assert rel_err(got_y, want_y) <= tol
    assert rel_err(got_glob, want_g['global']) <= tol
    assert rel_err(got_d, xd.astype(np.float64) @ wd.astype(np.float64)) <= tol
    if mode == 'bf16':
        assert rel_err(got_y, want_y) > 1e-5           # the mode really is on
    with pytest.raises(amd.GunetError):
        amd._lib.call('gunet_set_matmul_mode', 17)


@pytest.mark.parametrize('size', [(40, 52, 0.45), (23, 31, 0.6)])
def test_bf16_mode_resident_kernels_against_oracle(amd, oracle, size):
    """The bf16 mode with every GEMM forced onto the weights-resident kernels -- the ones a full-size
    graph takes -- on small graphs, against the float32 oracle: 2e-2 normwise, the bf16 gate.  The
    default choice would give a graph this small the tiled kernels."""
    x, graph, _ = make_case(oracle, *size)
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    want_y, want_g = oracle.graph_unet_forward(x, copy_graph(graph), W)
    amd.set_matmul_mode('bf16')
    amd.set_kernel_choice('resident')
    try:
        got_y, got_g = _unet(amd, oracle, W)(to_dev(x), graph_to_dev(graph))
        got_y, got_glob = got_y.cpu().numpy(), got_g['global'].cpu().numpy()
        got_e = [e.cpu().numpy() for e in got_g['edges']]
    finally:
        amd.set_kernel_choice('auto')
        amd.set_matmul_mode('f32')
    assert np.isfinite(got_y).all()
    assert rel_err(got_y, want_y) <= 2e-2, rel_err(got_y, want_y)
    assert rel_err(got_glob, want_g['global']) <= 2e-2
    for q in range(4):
        assert got_e[q].dtype == np.float32 and rel_err(got_e[q], want_g['edges'][q]) <= 2e-2
    assert rel_err(got_y, want_y) > 1e-5               # the mode really is on


def test_bf16_mode_single_layer_with_skip_segment(amd, oracle, resident_kernels):
    """One GN layer in bf16 mode on the resident kernels, with a skip segment (three operand segments
    of the node MLP) and 16 edge features: node features, messages and global against the oracle at
    the bf16 gate."""
    node_loc, ei, d, _ = oracle.synth_road_graph(29, 37, 0.55, 17)
    quads = list(oracle.sample_quadrant_subgraphs(ei, d))
    rng = np.random.Generator(np.random.PCG64(19))
    n = node_loc.shape[0]
    x = (rng.random((n, 256)) - 0.3).astype(np.float32)
    graph = {'edge_index': quads, 'node_loc': node_loc,
             'edges': [rng.random((q.shape[0], 16)).astype(np.float32) for q in quads],
             'global': rng.random((1, 128)).astype(np.float32)}
    W = _layer_weights(oracle, 'L', 256, 16, 128)
    want_x, want_g = oracle.geo_quadrant_gcn(x, copy_graph(graph), W, 'L')
    layer = amd.hybrid_gnn.GeoQuadrantGCN(128, activation='relu', use_bias=True, use_global=True)
    layer.load_weights(W, 'L', 'cuda')
    xt = to_dev(x)
    amd.set_matmul_mode('bf16')
    try:
        got_x, got_g = layer._forward(xt[:, :128].contiguous(), graph_to_dev(graph), x_skip=xt[:, 128:].contiguous())
        torch.cuda.synchronize()
    finally:
        amd.set_matmul_mode('f32')
    assert rel_err(got_x.cpu().numpy(), want_x) <= 2e-2
    assert rel_err(got_g['global'].cpu().numpy(), want_g['global']) <= 2e-2
    for q in range(4):
        assert rel_err(got_g['edges'][q].cpu().numpy(), want_g['edges'][q]) <= 2e-2


def test_tiled_gemm_k_tail_reads_zeros_not_memory(amd):
    """A K segment that is wide (beyond 1024 columns) and no whole number of 32-wide K tiles, on the
    tiled kernel: the columns of the last tile past k must enter as zeros whatever lies behind the
    operand row and behind the weight matrix in memory -- here NaNs."""
    rng = np.random.Generator(np.random.PCG64(77))
    rows, k, n = 777, 1044, 128
    xbuf = torch.full((rows * k + 4096,), float('nan'), device='cuda')
    wbuf = torch.full(((k + 64) * n,), float('nan'), device='cuda')
    x = rng.standard_normal((rows, k)).astype(np.float32)
    w = (rng.standard_normal((k, n)) * 0.05).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    xbuf[:rows * k] = to_dev(x).reshape(-1)
    wbuf[:k * n] = to_dev(w).reshape(-1)
    layer = amd.hybrid_gnn.Dense(n, activation='relu', use_bias=True)
    layer.load_weights({'d/kernel': w, 'd/bias': b}, 'd', 'cuda')
    layer.kernel = wbuf[:k * n].view(k, n)          # the same values with NaNs right behind the last row
    amd.set_kernel_choice('tiled')
    try:
        got = layer(xbuf[:rows * k].view(rows, k)).cpu().numpy()
    finally:
        amd.set_kernel_choice('auto')
    want = np.maximum(x.astype(np.float64) @ w.astype(np.float64) + b, 0)
    assert np.isfinite(got).all()
    assert_close(got, want, TOL)


def test_segment_max_drops_nans(amd):
    """gunet_segment_max_f32 takes fmaxf's maximum starting from -FLT_MAX: a NaN among the members of a
    segment is dropped, a segment of NaNs only gives -FLT_MAX like an empty one (TensorFlow's
    unsorted_segment_max would hand the NaN on; the deviation is documented in the header)."""
    from gunet_amd import _lib, plan
    nan = float('nan')
    src = torch.tensor([[1.0, nan, 3.0, -2.0], [nan, nan, 0.5, -7.0], [2.0, nan, nan, nan], [5.0, 6.0, 7.0, 8.0]], device='cuda')
    keys = torch.tensor([0, 0, 0, 2], dtype=torch.int32, device='cuda')      # segment 1 is empty
    rowptr, perm = plan.csr_from_keys(keys, 3)
    out = torch.empty((3, 4), device='cuda')
    _lib.call('gunet_segment_max_f32', _lib.ptr(src), 4, 4, _lib.ptr(rowptr), _lib.ptr(perm), 3, 1, _lib.ptr(out),
              _lib.current_stream(src.device))
    got = out.cpu().numpy()
    lowest = -np.finfo(np.float32).max
    assert np.array_equal(got[0], np.array([2.0, lowest, 3.0, -2.0], np.float32))
    assert (got[1] == lowest).all()
    assert np.array_equal(got[2], [5.0, 6.0, 7.0, 8.0])


def test_dense_linear_keeps_nan(amd, resident_kernels):
    """A linear Dense layer hands a NaN on (as TensorFlow's does) on both GEMM families; ReLU of a NaN
    is 0 here (fmaxf), NaN in TensorFlow -- the documented deviation."""
    rng = np.random.Generator(np.random.PCG64(5))
    x = rng.standard_normal((500, 128)).astype(np.float32)
    x[7, 3] = np.nan
    w = (rng.standard_normal((128, 128)) * 0.1).astype(np.float32)
    for choice in ('resident', 'tiled'):
        amd.set_kernel_choice(choice)
        lin = amd.hybrid_gnn.Dense(128, use_bias=False)
        lin.load_weights({'d/kernel': w}, 'd', 'cuda')
        got = lin(to_dev(x)).cpu().numpy()
        assert np.isnan(got[7]).all() and np.isfinite(np.delete(got, 7, axis=0)).all(), choice
        relu = amd.hybrid_gnn.Dense(128, activation='relu', use_bias=False)
        relu.load_weights({'d/kernel': w}, 'd', 'cuda')
        got = relu(to_dev(x)).cpu().numpy()
        assert (got[7] == 0).all() and np.isfinite(got).all(), choice


def test_gn_layer_arbitrary_graph(amd, oracle):
    """Edge lists that are no pixel grid (random senders and receivers, in-degree far above 3, some
    nodes without incoming edges, one empty quadrant): the layer must not rely on grid structure."""
    rng = np.random.Generator(np.random.PCG64(21))
    n = 300
    quads = [rng.integers(0, n - 40, (m, 2)).astype(np.int32) for m in (1500, 37, 0, 2600)]
    x = (rng.random((n, 128)) - 0.4).astype(np.float32)
    graph = {'edge_index': quads, 'node_loc': np.zeros((n, 2), np.int64),
             'edges': [rng.random((q.shape[0], 32)).astype(np.float32) for q in quads],
             'global': rng.random((1, 128)).astype(np.float32)}
    W = _layer_weights(oracle, 'L', 128, 32, 128)
    want_x, want_g = oracle.geo_quadrant_gcn(x, copy_graph(graph), W, 'L')
    got_x, got_g = _run_layer(amd, oracle, x, graph, W, 'L')
    qg = amd.plan.get_quadgraph(graph_to_dev(graph)['edge_index'], n)
    assert qg.max_in_degree > 3 and qg.in_table is None
    assert_close(got_x.cpu().numpy(), want_x, TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)
    for q in range(4):
        assert_close(got_g['edges'][q].cpu().numpy(), want_g['edges'][q], TOL)


def test_in_edge_table(amd, oracle):
    """The per-node table of incoming edges lists, per quadrant, the edge positions whose receiver is
    the node, ascending, -1 padded; road graphs and their poolings never exceed 3."""
    node_loc, ei, d, _ = oracle.synth_road_graph(30, 26, 0.7, 5)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    qg = amd.plan.QuadGraph([to_dev(q, torch.int32) for q in quads], node_loc.shape[0])
    assert 1 <= qg.max_in_degree <= 3
    table = qg.in_table.cpu().numpy()
    for q in range(4):
        recv = quads[q][:, 1]
        for v in range(node_loc.shape[0]):
            want = np.flatnonzero(recv == v)
            assert list(table[v, q, :want.size]) == list(want) and (table[v, q, want.size:] == -1).all()


@pytest.fixture
def resident_kernels(amd):
    """Take the weights-resident / message-summing GEMMs whatever the size (small test graphs would
    otherwise get the tiled kernel)."""
    amd.set_kernel_choice('resident')
    yield
    amd.set_kernel_choice('auto')


@pytest.mark.parametrize('choice', ['resident', 'tiled'])
def test_graph_unet_kernel_choices(amd, oracle, choice):
    """The whole model with every GEMM forced onto one kernel family."""
    x, graph, _ = make_case(oracle, 40, 52, 0.45)
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    want_y, want_g = oracle.graph_unet_forward(x, copy_graph(graph), W)
    amd.set_kernel_choice(choice)
    try:
        got_y, got_g = _unet(amd, oracle, W)(to_dev(x), graph_to_dev(graph))
        got_y, got_glob = got_y.cpu().numpy(), got_g['global'].cpu().numpy()
    finally:
        amd.set_kernel_choice('auto')
    assert_close(got_y, want_y, TOL)
    assert_close(got_glob, want_g['global'], TOL)


@pytest.mark.parametrize('k', [104, 100, 124, 116])
def test_dense_ragged_k_resident(amd, resident_kernels, k):
    """Input widths that are not a whole number of 32-wide K tiles (the model's 104-column in_layer) on
    the weights-resident GEMM: the last tile's missing columns are masked, rows are not a multiple of
    the 32-row groups, and NaNs planted behind the last row's end must not leak in through the
    redirected loads.  Same result as the tiled kernel to rounding, and as float64 to 1e-5."""
    rng = np.random.Generator(np.random.PCG64(40 + k))
    rows, n = 1000 + k, 128
    buf = torch.full((rows * k + 64,), float('nan'), device='cuda')
    x = rng.standard_normal((rows, k)).astype(np.float32)
    buf[:rows * k] = to_dev(x).reshape(-1)
    xd = buf[:rows * k].view(rows, k)
    w = rng.standard_normal((k, n)).astype(np.float32) * 0.1
    b = rng.standard_normal(n).astype(np.float32)
    layer = amd.hybrid_gnn.Dense(n, activation='relu', use_bias=True)
    layer.load_weights({'d/kernel': w, 'd/bias': b}, 'd', 'cuda')
    got = layer(xd).cpu().numpy()
    want = np.maximum(x.astype(np.float64) @ w.astype(np.float64) + b, 0)
    assert np.isfinite(got).all()
    assert_close(got, want, TOL)
    amd.set_kernel_choice('tiled')
    tiled = layer(xd).cpu().numpy()
    assert rel_err(got, tiled) <= 1e-6


def test_gn_layer_fused_aggregation_edge_cases(amd, oracle, resident_kernels):
    """The message-summing node MLP on a grid graph with one quadrant emptied and a skip segment
    (three operand segments), single sample and batched."""
    node_loc, ei, d, _ = oracle.synth_road_graph(29, 37, 0.55, 17)
    quads = list(oracle.sample_quadrant_subgraphs(ei, d))
    quads[2] = quads[2][:0]
    rng = np.random.Generator(np.random.PCG64(18))
    n = node_loc.shape[0]
    x = (rng.random((n, 256)) - 0.3).astype(np.float32)
    graph = {'edge_index': quads, 'node_loc': node_loc,
             'edges': [rng.random((q.shape[0], 16)).astype(np.float32) for q in quads],
             'global': rng.random((1, 128)).astype(np.float32)}
    W = _layer_weights(oracle, 'L', 256, 16, 128)
    want_x, want_g = oracle.geo_quadrant_gcn(x, copy_graph(graph), W, 'L')
    layer = amd.hybrid_gnn.GeoQuadrantGCN(128, activation='relu', use_bias=True, use_global=True)
    layer.load_weights(W, 'L', 'cuda')
    g = graph_to_dev(graph)
    assert amd.plan.get_quadgraph(g['edge_index'], n).in_table is not None
    xt = to_dev(x)
    got_x, got_g = layer._forward(xt[:, :128].contiguous(), g, x_skip=xt[:, 128:].contiguous())
    assert_close(got_x.cpu().numpy(), want_x, TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)
    g2 = graph_to_dev(graph)
    g2['edge_index'] = g['edge_index']
    g2['edges'] = [e.unsqueeze(0).repeat(3, 1, 1) for e in g2['edges']]
    g2['global'] = g2['global'].repeat(3, 1)
    bx, bg = layer._forward(xt[:, :128].contiguous().unsqueeze(0).repeat(3, 1, 1), g2,
                            x_skip=xt[:, 128:].contiguous().unsqueeze(0).repeat(3, 1, 1))
    for b in range(3):
        assert torch.equal(bx[b], got_x) and torch.equal(bg['global'][b], got_g['global'][0])


def test_captured_forward(amd, oracle):
    """The forward replayed as one HIP graph gives the eager results bit for bit, for new inputs too."""
    x, graph, _ = make_case(oracle, 40, 52, 0.45)
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    model = _unet(amd, oracle, W)
    g = graph_to_dev(graph)
    xt = to_dev(x)
    eager = model.predict(xt, dict(g)).clone()
    cap = amd.CapturedForward(model, xt, g)
    assert torch.equal(cap(xt), eager)
    x2 = to_dev(np.ascontiguousarray(x[::-1] * np.float32(0.5)))
    glob2 = g['global'] * 1.5
    g2 = dict(g)
    g2['global'] = glob2
    eager2 = model.predict(x2, g2).clone()
    assert torch.equal(cap(x2, glob=glob2), eager2)
    assert torch.equal(cap(xt, glob=g['global']), eager)
    # a later, larger forward replaces the shared scratch buffers; the graph has its own
    big = dict(g)
    big['edges'] = [e.unsqueeze(0).repeat(6, 1, 1) for e in g['edges']]
    big['global'] = g['global'].repeat(6, 1)
    yb = model.predict(xt.unsqueeze(0).repeat(6, 1, 1), big)
    junk = torch.full((1 << 22,), float('nan'), device='cuda')
    assert torch.equal(yb[3], eager)
    assert torch.equal(cap(xt, glob=g['global']), eager)
    del junk
    # the per-layer mirrors captured the same way
    model.native_forward = False
    cap2 = amd.CapturedForward(model, xt, g)
    model.native_forward = True
    assert torch.equal(cap2(x2, glob=glob2), eager2)
