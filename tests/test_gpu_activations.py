"""Keras activations other than relu / linear (layers/hybrid_gnn.py:25 and models/graph_unet.py:21
take any name): the GN layer, the graph upsampling layer, Dense and the whole model, each against
the oracle at the float32 gate.  These run on the library's generic kernels, layer by layer."""
import numpy as np
import pytest
import torch

from conftest import assert_close
from gpu_helpers import copy_graph, graph_to_dev, make_case, to_dev

pytestmark = pytest.mark.gpu
TOL = 1e-5
NAMES = ['tanh', 'sigmoid', 'elu', 'softplus', 'selu', 'swish', 'gelu', 'softsign', 'exponential', 'hard_sigmoid']


@pytest.fixture
def amd():
    import gunet_amd
    return gunet_amd


def _layer_weights(oracle, prefix, c_in, f_e, g, seed=1):
    shapes = oracle.gn_layer_shapes(prefix, c_in, f_e, g, 128)
    return oracle.make_weights(shapes, seed)


@pytest.mark.parametrize('name', NAMES)
def test_dense_activations(amd, oracle, name):
    rng = np.random.Generator(np.random.PCG64(4))
    for rows, k, n in [(700, 104, 128), (50, 10, 7)]:           # the tiled GEMM and the plain kernel
        x = rng.standard_normal((rows, k)).astype(np.float32)
        w = rng.standard_normal((k, n)).astype(np.float32) * 0.04   # pre-activations of order 0.4: little saturation
        b = rng.standard_normal(n).astype(np.float32)
        layer = amd.hybrid_gnn.Dense(n, activation=name, use_bias=True)
        layer.load_weights({'d/kernel': w, 'd/bias': b}, 'd', 'cuda')
        got = layer(to_dev(x)).cpu().numpy()
        want = oracle.ACTIVATIONS[name](x.astype(np.float64) @ w.astype(np.float64) + b)
        assert_close(got, want, TOL, f'{name} {(rows, k, n)}')


# ('exponential' is left to the Dense test: on this layer's weights exp overflows float32 in the oracle too)
@pytest.mark.parametrize('name', [n for n in NAMES if n != 'exponential'])
def test_gn_layer_activations(amd, oracle, name):
    c_in, f_e, g = 128, 16, 113
    node_loc, ei, d, _ = oracle.synth_road_graph(23, 31, 0.55, 2)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    rng = np.random.Generator(np.random.PCG64(9))
    n = node_loc.shape[0]
    x = rng.random((n, c_in)).astype(np.float32)
    graph = {'edge_index': list(quads), 'node_loc': node_loc,
             'edges': [rng.random((q.shape[0], f_e)).astype(np.float32) for q in quads],
             'global': rng.random((1, g)).astype(np.float32)}
    W = _layer_weights(oracle, 'L', c_in, f_e, g)
    want_x, want_g = oracle.geo_quadrant_gcn(x, copy_graph(graph), W, 'L', activation=name)
    layer = amd.hybrid_gnn.GeoQuadrantGCN(128, activation=name, use_bias=True, use_global=True)
    layer.load_weights(W, 'L', 'cuda')
    for batch in (None, 3):
        gd = graph_to_dev(graph)
        xt = to_dev(x)
        if batch:
            xt = xt.unsqueeze(0).repeat(batch, 1, 1)
            gd['edges'] = [e.unsqueeze(0).repeat(batch, 1, 1) for e in gd['edges']]
            gd['global'] = gd['global'].repeat(batch, 1)
        got_x, got_g = layer(xt, gd)
        pick = (lambda t: t[batch - 1]) if batch else (lambda t: t)
        assert_close(pick(got_x).cpu().numpy(), want_x, TOL, name)
        for q in range(4):
            assert_close(pick(got_g['edges'][q]).cpu().numpy(), want_g['edges'][q], TOL, name)
        got_glob = got_g['global'][batch - 1:batch] if batch else got_g['global']
        assert_close(got_glob.cpu().numpy(), want_g['global'], TOL, name)


@pytest.mark.parametrize('name', ['tanh', 'swish'])
def test_graph_unet_with_other_activations(amd, oracle, name):
    """The whole model with another activation in every GN layer and at the output: the one-call
    forward declines (linear / relu only), the layer objects take over."""
    x, graph, _ = make_case(oracle, 40, 52, 0.45)
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    want_y, want_g = oracle.graph_unet_forward(x, copy_graph(graph), W, activation=name)
    want_y = oracle.ACTIVATIONS['sigmoid'](want_y)
    model = amd.GraphUNet(32, out_units=96, depth=3, activation=name, layer_type='geo_quadrant_gcn', use_bias=True,
                          use_global=True, output_activation='sigmoid').load_weights(W, 'cuda')
    assert not model._native_ok()
    got_y, got_g = model(to_dev(x), graph_to_dev(graph))
    assert_close(got_y.cpu().numpy(), want_y, TOL, name)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL, name)
    for q in range(4):
        assert_close(got_g['edges'][q].cpu().numpy(), want_g['edges'][q], TOL, name)


def test_one_call_forward_declines_other_activations(amd, oracle):
    import ctypes as C
    from gunet_amd import _lib, plan
    from gunet_amd._lib import GunetError, call, current_stream, ptr, ptr4
    x, graph, _ = make_case(oracle, 12, 16, 0.6)
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    model = amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn', use_bias=True,
                          use_global=True).load_weights(W, 'cuda')
    g = graph_to_dev(graph)
    xb = to_dev(x[None])
    feats = [e.unsqueeze(0).contiguous() for e in g['edges']]
    up = plan.get_unet_plan(g['node_loc'], g['edge_index'], 3)
    ws = model._native_weights(xb.shape[2], feats[0].shape[2], g['global'].shape[1], xb.device)
    nbytes = int(_lib.lib.gunet_unet_workspace_bytes(up.handle, C.byref(ws), 1))
    scratch = torch.empty(nbytes + 256, dtype=torch.uint8, device='cuda')
    base = (scratch.data_ptr() + 255) // 256 * 256
    out = torch.empty((1, xb.shape[1], 96), device='cuda')
    in_ptrs = ptr4(feats)
    with pytest.raises(GunetError) as err:
        call('gunet_unet_forward_f32', up.handle, C.byref(ws), 1, ptr(xb), C.byref(in_ptrs), 0, ptr(g['global']),
             _lib.ACT_TANH, _lib.ACT_LINEAR, ptr(out), None, None, C.c_void_p(base), nbytes, current_stream(xb.device))
    assert err.value.code == -4 and 'layer by layer' in str(err.value)
