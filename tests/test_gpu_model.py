"""The whole-model entry points: gunet_plan_* (every level's index structures behind one handle)
and gunet_unet_forward_f32 (GraphUNet.call as one library call).

  * the plan's levels, pooling maps and upsampling graphs against the oracle -- bit-exact;
  * the one-call forward against the oracle (fp32: max|got - want| / max|want| <= 1e-5) and
    against the layer-by-layer path through the per-layer entry points -- bit-identical;
  * argument checking: weights that do not chain, workspace too small."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import assert_bracketed, assert_close, exact_inputs, rel_err
from gpu_helpers import copy_graph, graph_to_dev, make_case, to_dev

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope='module')
def amd():
    import gunet_amd
    return gunet_amd


def _model(amd, W, depth=3, units=32, **kw):
    model = amd.GraphUNet(units, out_units=96, depth=depth, activation='relu', layer_type='geo_quadrant_gcn',
                          use_bias=True, use_global=True, output_activation=None, **kw)
    return model.load_weights(W, 'cuda')


@pytest.mark.parametrize('size', [(12, 16, 0.6), (40, 52, 0.45), (97, 83, 0.4)])
def test_plan_levels_bit_exact(amd, oracle, size):
    from gunet_amd import plan
    node_loc, ei, d, _ = oracle.synth_road_graph(*size, 3)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    depth = 3
    up = plan.UNetPlan(to_dev(node_loc, torch.int64), [to_dev(q, torch.int32) for q in quads], depth)
    loc, lists = node_loc, list(quads)
    for level in range(depth + 1):
        got = up.level(level)
        assert got['n_nodes'] == loc.shape[0] == up.n_nodes[level]
        assert np.array_equal(got['node_loc'].cpu().numpy(), loc)
        for q in range(4):
            assert np.array_equal(got['edge_index'][q].cpu().numpy(), lists[q]), (level, q)
            # receiver CSR: rows ascending in the reference's edge order
            rowptr, perm = got['rowptr'][q].cpu().numpy(), got['perm'][q].cpu().numpy()
            order = np.argsort(lists[q][:, 1], kind='stable') if lists[q].shape[0] else np.zeros(0, np.int64)
            assert np.array_equal(perm[:lists[q].shape[0]], order)
            assert np.array_equal(rowptr, np.concatenate(([0], np.cumsum(np.bincount(lists[q][:, 1], minlength=loc.shape[0])))))
        if level == depth:
            break
        # pooling level -> level + 1 (hybrid_gnn.py:415-434)
        ids, n_pooled = oracle.cluster_ids_2x2(loc)
        cluster, dup_map = up.pooling(level)
        assert np.array_equal(cluster.cpu().numpy(), ids)
        coarse_loc = oracle.unsorted_segment_min(loc // 2, ids, n_pooled)
        new_lists = []
        for q in range(4):
            feats = np.arange(lists[q].shape[0], dtype=np.float32)[:, None]
            new_e, _ = oracle.remove_duplicate_edges(ids[lists[q]], np.int64(loc.shape[0]), feats)
            new_lists.append(new_e)
            dm = dup_map[q].cpu().numpy()
            relabelled = ids[lists[q]]
            alive = dm >= 0
            assert np.array_equal(alive, relabelled[:, 0] != relabelled[:, 1])
            assert np.array_equal(new_e[dm[alive]], relabelled[alive])
        # upsampling level + 1 -> level (hybrid_gnn.py:336-357)
        parent, upg = up.upsampling(level)
        want_parent, want_lists = oracle.upsample_edges(coarse_loc, loc)
        assert np.array_equal(parent.cpu().numpy(), want_parent)
        assert upg['n_nodes'] == n_pooled + loc.shape[0]
        for q in range(4):
            assert np.array_equal(upg['edge_index'][q].cpu().numpy(), want_lists[q])
        loc, lists = coarse_loc, new_lists


# (18, 50): more than twice as wide as high, where the reference's cell ids collide and some fine
# nodes get no upsampling edge -- the forward then runs that layer on [x ; zeros] as written
@pytest.mark.parametrize('size,batch', [((12, 16, 0.6), 1), ((40, 52, 0.45), 3), ((18, 50, 0.55), 2)])
def test_unet_forward_one_call(amd, oracle, size, batch):
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    model = _model(amd, W)
    assert model._native_ok()
    x, graph, _ = make_case(oracle, *size)
    node_loc, quads = graph['node_loc'], graph['edge_index']
    enc = oracle.synth_street_encoding(node_loc.shape[0], 5)
    xs, graphs = [], []
    for b in range(batch):
        sn, te, wd = oracle.synth_sample(node_loc, quads, 300 + b, time_idx=5 * b, weekday=b)
        xb, gb = oracle.assemble_inputs(sn, enc, quads, node_loc, te, wd)
        xs.append(xb)
        graphs.append(gb)

    def dev_graph():
        g = graph_to_dev(graphs[0])
        g['edge_index'], g['node_loc'] = shared['edge_index'], shared['node_loc']
        g['edges'] = [torch.stack([to_dev(gr['edges'][q]) for gr in graphs]) for q in range(4)]
        g['global'] = torch.cat([to_dev(gr['global']) for gr in graphs], dim=0)
        return g

    shared = graph_to_dev(graphs[0])
    xd = torch.stack([to_dev(v) for v in xs])
    got_y, got_g = model(xd, dev_graph())
    for b in range(batch):
        want_y, want_g = oracle.graph_unet_forward(xs[b], copy_graph(graphs[b]), W)
        assert_close(got_y[b].cpu().numpy(), want_y, TOL)
        assert_close(got_g['global'][b:b + 1].cpu().numpy(), want_g['global'], TOL)
        # no further from the float64 forward than twice the float32 oracle is
        exact_y, exact_g = oracle.graph_unet_forward(*exact_inputs(xs[b], graphs[b], W))
        assert_bracketed(got_y[b].cpu().numpy(), want_y, exact_y, TOL, f'sample {b}')
        for q in range(4):
            assert_close(got_g['edges'][q][b].cpu().numpy(), want_g['edges'][q], TOL)
            assert_bracketed(got_g['edges'][q][b].cpu().numpy(), want_g['edges'][q], exact_g['edges'][q], TOL,
                             f'sample {b}, messages of direction {q}')
            assert np.array_equal(got_g['edge_index'][q].cpu().numpy(), want_g['edge_index'][q])
    # the layer objects, one library call per layer: same kernels, same order
    model.native_forward = False
    assert not model._native_ok()
    ref_y, ref_g = model(xd, dev_graph())
    assert torch.equal(got_y, ref_y)
    assert torch.equal(got_g['global'], ref_g['global'])
    for q in range(4):
        assert torch.equal(got_g['edges'][q], ref_g['edges'][q])
    # predict() = forward + max(pred, 0) fused into the last GEMM
    model.native_forward = True
    assert torch.equal(model.predict(xd, dev_graph()), torch.clamp_min(got_y, 0))
    # 2-D single-sample form
    y1, g1 = model(xd[0], {**dev_graph(), 'edges': [to_dev(graphs[0]['edges'][q]) for q in range(4)],
                           'global': to_dev(graphs[0]['global'])})
    assert y1.dim() == 2 and g1['edges'][0].dim() == 2
    assert_close(y1.cpu().numpy(), oracle.graph_unet_forward(xs[0], copy_graph(graphs[0]), W)[0], TOL)


@pytest.mark.parametrize('size', [(1, 2), (2, 2), (3, 5), (1, 7), (5, 4)])
def test_unet_forward_tiny_graphs(amd, oracle, size):
    """Graphs that run out of edges (and nearly of nodes) on the way down: the deeper levels have a
    single node or no edges at all in some or all quadrants."""
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    model = _model(amd, W)
    x, graph, _ = make_case(oracle, size[0], size[1], 1.0, dense=True)
    want_y, want_g = oracle.graph_unet_forward(x, copy_graph(graph), W)
    got_y, got_g = model(to_dev(x), graph_to_dev(graph))
    assert_close(got_y.cpu().numpy(), want_y, TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)
    model.native_forward = False
    ref_y, ref_g = model(to_dev(x), graph_to_dev(graph))
    assert torch.equal(got_y, ref_y) and torch.equal(got_g['global'], ref_g['global'])


@pytest.mark.parametrize('seed', range(6))
def test_unet_forward_random_shapes(amd, oracle, seed):
    """Random image sizes (tall, wide, odd), densities and batch sizes; depth 2 or 3."""
    rng = np.random.Generator(np.random.PCG64(900 + seed))
    h, w = int(rng.integers(5, 70)), int(rng.integers(5, 70))
    density = float(rng.uniform(0.3, 0.9))
    depth = int(rng.integers(2, 4))
    batch = int(rng.integers(1, 4))
    shapes = oracle.graph_unet_shapes(depth=depth)
    W = oracle.make_weights(shapes, 10 + seed)
    model = _model(amd, W, depth=depth)
    x, graph, _ = make_case(oracle, h, w, density, graph_seed=seed)
    node_loc, quads = graph['node_loc'], graph['edge_index']
    enc = oracle.synth_street_encoding(node_loc.shape[0], 5)
    xs, graphs = [], []
    for b in range(batch):
        sn, te, wd = oracle.synth_sample(node_loc, quads, 500 + 10 * seed + b, time_idx=b, weekday=b)
        xb, gb = oracle.assemble_inputs(sn, enc, quads, node_loc, te, wd)
        xs.append(xb)
        graphs.append(gb)
    g = graph_to_dev(graphs[0])
    g['edges'] = [torch.stack([to_dev(gr['edges'][q]) for gr in graphs]) for q in range(4)]
    g['global'] = torch.cat([to_dev(gr['global']) for gr in graphs], dim=0)
    xd = torch.stack([to_dev(v) for v in xs])
    got_y, got_g = model(xd, dict(g))
    for b in range(batch):
        want_y, want_g = oracle.graph_unet_forward(xs[b], copy_graph(graphs[b]), W, depth=depth)
        assert_close(got_y[b].cpu().numpy(), want_y, TOL, str((h, w, depth, b)))
        assert_close(got_g['global'][b:b + 1].cpu().numpy(), want_g['global'], TOL)
    model.native_forward = False
    ref_y, ref_g = model(xd, dict(g))
    assert torch.equal(got_y, ref_y) and torch.equal(got_g['global'], ref_g['global'])


def test_unet_forward_mixed_widths(amd, oracle):
    """units=64, depth=3: block widths 128, 128, 256 on the way down (hybrid widths in one model)."""
    shapes = oracle.graph_unet_shapes_for(64, depth=3)
    W = oracle.make_weights(shapes, 2)
    model = _model(amd, W, units=64)
    x, graph, _ = make_case(oracle, 30, 26, 0.55)
    want_y, _ = oracle.graph_unet_forward(x, copy_graph(graph), W)
    got_y, _ = model(to_dev(x), graph_to_dev(graph))
    assert_close(got_y.cpu().numpy(), want_y, TOL)
    model.native_forward = False
    ref_y, _ = model(to_dev(x), graph_to_dev(graph))
    assert torch.equal(got_y, ref_y)


def test_unet_forward_repeatable_full_size(amd, oracle):
    """495x436 road graph, batch 3, thirty forwards: every run and every batch row of identical
    inputs gives the same bits (no atomics or launch-order dependence anywhere on the path)."""
    node_loc, ei, d, _ = oracle.synth_road_graph(495, 436, 0.45, 0)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    enc = oracle.synth_street_encoding(node_loc.shape[0], 5)
    sn, te, wd = oracle.synth_sample(node_loc, quads, 77, time_idx=3, weekday=2)
    x, graph = oracle.assemble_inputs(sn, enc, quads, node_loc, te, wd)
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    model = _model(amd, W)
    g = graph_to_dev(graph)
    g['edges'] = [e.unsqueeze(0).repeat(3, 1, 1) for e in g['edges']]
    g['global'] = g['global'].repeat(3, 1)
    xd = to_dev(x).unsqueeze(0).repeat(3, 1, 1)
    first = model.predict(xd, dict(g)).clone()
    assert torch.equal(first[0], first[1]) and torch.equal(first[0], first[2])
    for _ in range(30):
        assert torch.equal(model.predict(xd, dict(g)), first)


def test_unet_reload_weights(amd, oracle):
    """load_weights after a forward: the next forward uses the new tensors."""
    x, graph, _ = make_case(oracle, 12, 16, 0.6)
    model = None
    for seed in (1, 2):
        W = oracle.make_weights(oracle.graph_unet_shapes(), seed)
        model = _model(amd, W) if model is None else model.load_weights(W, 'cuda')
        want_y, _ = oracle.graph_unet_forward(x, copy_graph(graph), W)
        got_y, _ = model(to_dev(x), graph_to_dev(graph))
        assert_close(got_y.cpu().numpy(), want_y, TOL, str(seed))


def test_unet_forward_errors(amd, oracle):
    from gunet_amd import _lib, plan
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    model = _model(amd, W)
    x, graph, _ = make_case(oracle, 12, 16, 0.6)
    g = graph_to_dev(graph)
    xd = to_dev(x)
    model(xd, dict(g))
    up = plan.get_unet_plan(g['node_loc'], g['edge_index'], 3)
    ws = model._native_struct
    need = _lib.lib.gunet_unet_workspace_bytes(up.handle, C.byref(ws), 1)
    assert need > 0
    out = torch.empty((1, x.shape[0], 96), device='cuda')
    buf = torch.empty(need + 512, dtype=torch.uint8, device='cuda')
    base = (buf.data_ptr() + 255) // 256 * 256
    feats = _lib.ptr4([e.unsqueeze(0) for e in g['edges']])
    args = lambda nbytes: (up.handle, C.byref(ws), 1, _lib.ptr(xd), C.byref(feats), 0, _lib.ptr(g['global']), 1, 0,
                           _lib.ptr(out), None, None, C.c_void_p(base), nbytes, None)
    assert _lib.lib.gunet_unet_forward_f32(*args(need)) == 0
    torch.cuda.synchronize()
    assert _lib.lib.gunet_unet_forward_f32(*args(need - 1024)) == -3          # GUNET_ERR_WORKSPACE
    assert 'workspace' in _lib.last_error()
    # a plan of another depth, and weights whose widths do not chain
    up2 = plan.UNetPlan(g['node_loc'], g['edge_index'], 2)
    assert _lib.lib.gunet_unet_workspace_bytes(up2.handle, C.byref(ws), 1) == 0
    assert 'depth' in _lib.last_error()
    saved = ws.down[1].c_in
    ws.down[1].c_in = 96
    try:
        assert _lib.lib.gunet_unet_forward_f32(*args(need)) == -1             # GUNET_ERR_INVALID
        assert 'down[1]' in _lib.last_error()
    finally:
        ws.down[1].c_in = saved
    with pytest.raises(ValueError):
        model(xd[:-1], dict(g))


def test_shared_edge_features_equal_expanded(amd, oracle):
    """Input edge features given once for the whole batch ([E_q, F], edges_shared in the C ABI) give
    the same bits as B per-sample copies, in the one-call forward, the layer-by-layer path and a
    single layer; so does the 256-wide generic-free path's neighbour, the wide generic edge GEMM."""
    x, graph, _ = make_case(oracle, 33, 29, 0.55)
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    model = _model(amd, W)
    B = 3
    rng = np.random.default_rng(5)
    xb = to_dev(np.stack([x * rng.random(x.shape, dtype=np.float32) for _ in range(B)]))
    results = []
    for native in (True, False):
        model.native_forward = native
        for shared in (True, False):
            g = graph_to_dev(graph)
            g['global'] = g['global'].repeat(B, 1) * torch.arange(1, B + 1, device='cuda').reshape(B, 1)
            if not shared:
                g['edges'] = [e.unsqueeze(0).repeat(B, 1, 1).contiguous() for e in g['edges']]
            y, go = model(xb, g)
            results.append((y, go['global'], go['edges']))
    for y, gl, ed in results[1:]:
        assert torch.equal(y, results[0][0]) and torch.equal(gl, results[0][1])
        assert all(torch.equal(a, b) for a, b in zip(ed, results[0][2]))
    # a wide layer (generic edge GEMM): shared vs expanded
    from gunet_amd.layers.hybrid_gnn import GeoQuadrantGCN
    n = x.shape[0]
    Wl = oracle.make_weights(oracle.gn_layer_shapes('L', 128, 16, 113, 512), 2)
    layer = GeoQuadrantGCN(512, activation='relu', use_bias=True, use_global=True)
    layer.load_weights(Wl, 'L', 'cuda')
    xw = torch.rand((B, n, 128), device='cuda')
    outs = []
    for shared in (True, False):
        g = graph_to_dev(graph)
        g['global'] = g['global'].repeat(B, 1)
        if not shared:
            g['edges'] = [e.unsqueeze(0).repeat(B, 1, 1).contiguous() for e in g['edges']]
        outs.append(layer(xw, g))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1]['global'], outs[1][1]['global'])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs in one process')
def test_tensors_on_a_device_that_is_not_current(amd, oracle):
    """Every entry point runs on the device of the stream it is given, whatever device is current:
    model inputs, one-call forward and layer-by-layer path with all tensors on cuda:1 while cuda:0
    is current give the bits of the same calls made with cuda:1 current."""
    x, graph, _ = make_case(oracle, 24, 30, 0.6)
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    outs = []
    for current in (1, 0):
        with torch.cuda.device(current):
            model = amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                                  use_bias=True, use_global=True).load_weights(W, 'cuda:1')
            for native in (True, False):
                model.native_forward = native
                g = {k: ([t.to('cuda:1') for t in v] if isinstance(v, list) else v.to('cuda:1'))
                     for k, v in graph_to_dev(graph).items()}
                y, go = model(to_dev(x).to('cuda:1'), g)
                torch.cuda.synchronize('cuda:1')
                assert y.device == torch.device('cuda:1')
                outs.append(y.cpu())
    for y in outs[1:]:
        assert torch.equal(y, outs[0])


def test_plan_pool_features_and_predict_out(amd, oracle):
    """gunet_plan_pool_features_f32 on a small graph against the oracle's graph_pooling_2x2 (bit-exact,
    either half alone too, argument errors), and predict(out=...) writing into a caller's slice."""
    from gunet_amd import _lib, plan
    x, graph, _ = make_case(oracle, 31, 27, 0.6)
    g = graph_to_dev(graph)
    up = plan.UNetPlan(g['node_loc'], g['edge_index'], 2)
    rng = np.random.default_rng(3)
    B, U, u = 2, 24, 8
    xs = rng.standard_normal((B, up.n_nodes[0], U)).astype(np.float32)
    es = [rng.standard_normal((B, up.n_edges[0][q], u)).astype(np.float32) for q in range(4)]
    xd, ed = to_dev(xs), [to_dev(e) for e in es]
    xo = torch.full((B, up.n_nodes[1], U), float('nan'), device='cuda')
    eo = [torch.full((B, up.n_edges[1][q], u), float('nan'), device='cuda') for q in range(4)]
    src, dst = _lib.ptr4(ed), _lib.ptr4(eo)
    stream = _lib.current_stream(xd.device)
    _lib.call('gunet_plan_pool_features_f32', up.handle, 0, B, _lib.ptr(xd), U, C.byref(src), u, _lib.ptr(xo), C.byref(dst), stream)
    for b in range(B):
        gq = {'edge_index': list(graph['edge_index']), 'edges': [e[b] for e in es], 'node_loc': graph['node_loc'], 'global': None}
        want_x, want_g = oracle.graph_pooling_2x2(xs[b], gq)
        assert np.array_equal(xo[b].cpu().numpy(), want_x)
        for q in range(4):
            assert np.array_equal(eo[q][b].cpu().numpy(), want_g['edges'][q])
    xo2 = torch.empty_like(xo)
    _lib.call('gunet_plan_pool_features_f32', up.handle, 0, B, _lib.ptr(xd), U, None, 0, _lib.ptr(xo2), None, stream)
    assert torch.equal(xo2, xo)
    assert _lib.lib.gunet_plan_pool_features_f32(up.handle, 2, B, _lib.ptr(xd), U, None, 0, _lib.ptr(xo2), None, None) == -1
    assert 'level' in _lib.last_error()
    assert _lib.lib.gunet_plan_pool_features_f32(up.handle, 0, B, _lib.ptr(xd), U, None, 0, None, None, None) == -1
    # predict(out=): the result lands in the caller's buffer, bit-identical to the allocating call
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    model = _model(amd, W)
    xb = to_dev(np.stack([x, x * 0.5]))
    def batch_graph():
        gg = graph_to_dev(graph)
        gg['global'] = gg['global'].repeat(2, 1)
        return gg
    want = model.predict(xb, batch_graph())
    buf = torch.zeros((5, x.shape[0], 96), device='cuda')
    got = model.predict(xb, batch_graph(), out=buf[2:4])
    assert got.data_ptr() == buf[2:4].data_ptr() and torch.equal(buf[2:4], want) and not buf[:2].any() and not buf[4:].any()
    with pytest.raises(ValueError):
        model.predict(xb, batch_graph(), out=buf[:3])


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_forward_repeats_bit_for_bit_over_a_poisoned_workspace(amd, oracle, mode):
    """The same forward 60 times with the shared workspace overwritten in between: every run must
    give the bits of the first.  Nothing in the workspace may be read before it is written, and no
    kernel may depend on timing (this is the check that caught an LDS pick-up the compiler had split
    into a 12-byte and two narrower reads, which now and then returned zeros in bf16 mode)."""
    from gunet_amd import plan
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    model = _model(amd, W)
    x, graph, _ = make_case(oracle, 40, 52, 0.45)
    g0 = graph_to_dev(graph)
    xd = to_dev(x)
    amd.set_matmul_mode(mode)
    try:
        first, _ = model(xd, dict(g0))
        first = first.clone()
        assert torch.isfinite(first).all()
        for it in range(60):
            for buf in plan._model_workspaces.values():
                buf.fill_(0xFF)          # bytes 0xFF: NaN as float32 and as bf16
            y, _ = model(xd, dict(g0))
            assert torch.equal(y, first), f'run {it} differs from the first ({mode})'
    finally:
        amd.set_matmul_mode('f32')


def test_release_temporaries_frees_the_block_cache_and_plans_still_build(amd, oracle):
    """The plan functions keep their temporary blocks in a cache inside the library;
    gunet_release_temporaries() (plan.release_workspaces() calls it) hands them back, and the next
    plan build -- from an empty cache -- gives the same index structures as one from a warm cache."""
    from gunet_amd import plan
    x, graph, _ = make_case(oracle, 97, 83, 0.4)
    g = graph_to_dev(graph)
    plan.clear_caches()
    warm = plan.get_unet_plan(g['node_loc'], g['edge_index'], 3)
    want = [[t.clone() for t in warm.level(l)['edge_index']] for l in range(4)]
    want_cluster = [warm.pooling(l)[0].clone() for l in range(3)]
    del warm
    plan.clear_caches()
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info()[0]
    plan.release_workspaces()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= before          # idle blocks went back to the device
    cold = plan.get_unet_plan(g['node_loc'], g['edge_index'], 3)
    for l in range(4):
        for q in range(4):
            assert torch.equal(cold.level(l)['edge_index'][q], want[l][q])
    for l in range(3):
        assert torch.equal(cold.pooling(l)[0], want_cluster[l])


@pytest.mark.parametrize('size', [(1, 2), (3, 5), (5, 4), (9, 33), (31, 17)])
def test_bf16_mode_tiny_and_ragged_graphs(amd, oracle, size):
    """bf16 mode (intermediates stored in bf16 inside the whole-model call) on graphs whose levels
    run out of edges or nodes and whose row counts are no multiple of a 32-row group: 2e-2 against
    the float32 oracle for the prediction, the global and the returned messages."""
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    model = _model(amd, W)
    x, graph, _ = make_case(oracle, size[0], size[1], 1.0, dense=True)
    want_y, want_g = oracle.graph_unet_forward(x, copy_graph(graph), W)
    amd.set_matmul_mode('bf16')
    try:
        got_y, got_g = model(to_dev(x), graph_to_dev(graph))
        assert rel_err(got_y.cpu().numpy(), want_y) <= 2e-2
        assert rel_err(got_g['global'].cpu().numpy(), want_g['global']) <= 2e-2
        for q in range(4):
            assert got_g['edges'][q].dtype == torch.float32
            if want_g['edges'][q].size:
                assert rel_err(got_g['edges'][q].cpu().numpy(), want_g['edges'][q]) <= 2e-2, q
    finally:
        amd.set_matmul_mode('f32')


@pytest.mark.parametrize('seed', range(4))
def test_bf16_mode_random_shapes_and_batches(amd, oracle, seed):
    rng = np.random.Generator(np.random.PCG64(1900 + seed))
    h, w = int(rng.integers(5, 70)), int(rng.integers(5, 70))
    density = float(rng.uniform(0.3, 0.9))
    batch = int(rng.integers(1, 5))
    W = oracle.make_weights(oracle.graph_unet_shapes(), 20 + seed)
    model = _model(amd, W)
    x, graph, _ = make_case(oracle, h, w, density, graph_seed=seed)
    node_loc, quads = graph['node_loc'], graph['edge_index']
    enc = oracle.synth_street_encoding(node_loc.shape[0], 5)
    xs, graphs = [], []
    for b in range(batch):
        sn, te, wd = oracle.synth_sample(node_loc, quads, 700 + 10 * seed + b, time_idx=b, weekday=b)
        xb, gb = oracle.assemble_inputs(sn, enc, quads, node_loc, te, wd)
        xs.append(xb)
        graphs.append(gb)
    g = graph_to_dev(graphs[0])
    g['edges'] = [torch.stack([to_dev(gr['edges'][q]) for gr in graphs]) for q in range(4)]
    g['global'] = torch.cat([to_dev(gr['global']) for gr in graphs], dim=0)
    xd = torch.stack([to_dev(v) for v in xs])
    amd.set_matmul_mode('bf16')
    try:
        got_y, got_g = model(xd, dict(g))
        # a sample's result does not depend on the batch it is computed in, in this mode either
        alone, _ = model(xd[:1].contiguous(), {**g, 'edges': [e[:1].contiguous() for e in g['edges']], 'global': g['global'][:1].contiguous()})
        assert torch.equal(alone[0], got_y[0])
    finally:
        amd.set_matmul_mode('f32')
    for b in range(batch):
        want_y, want_g = oracle.graph_unet_forward(xs[b], copy_graph(graphs[b]), W)
        assert rel_err(got_y[b].cpu().numpy(), want_y) <= 2e-2, (h, w, b)
        assert rel_err(got_g['global'][b:b + 1].cpu().numpy(), want_g['global']) <= 2e-2
