"""The 256-wide GN layer (u = 64 per direction) and models that mix 128- and 256-wide blocks, as
GraphUNet(units >= 64) builds them (reference models/graph_unet.py:37-55), against the oracle.

Tolerance (fp32 path): max|got - want| / max|want| <= 1e-5; index tensors bit-exact."""
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


def _layer(amd, W, prefix, with_gt=True, activation='relu', use_bias=True):
    layer = amd.hybrid_gnn.GeoQuadrantGCN(256, activation=activation, use_bias=use_bias, use_global=True,
                                          learn_global_scale=with_gt)
    layer.load_weights(W, prefix, 'cuda')
    return layer


@pytest.mark.parametrize('shape', [(256, 64, 256, True), (128, 32, 128, True), (512, 16, 256, False),
                                   (104, 5, 20, True), (256, 0, 64, True)])
def test_gn_layer_256(amd, oracle, shape):
    c_in, f_e, g, with_gt = shape
    node_loc, ei, d, _ = oracle.synth_road_graph(37, 45, 0.55, 21)       # > 1024 edges per quadrant
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    rng = np.random.Generator(np.random.PCG64(22))
    n = node_loc.shape[0]
    x = (rng.random((n, c_in)) - 0.2).astype(np.float32)
    graph = {'edge_index': list(quads), 'node_loc': node_loc,
             'edges': [(rng.random((q.shape[0], f_e)) - 0.3).astype(np.float32) for q in quads],
             'global': rng.random((1, g)).astype(np.float32)}
    W = oracle.make_weights(oracle.gn_layer_shapes('L', c_in, f_e, g, 256, with_gt), 4)
    want_x, want_g = oracle.geo_quadrant_gcn(x, copy_graph(graph), W, 'L')
    got_x, got_g = _layer(amd, W, 'L', with_gt)(to_dev(x), graph_to_dev(graph))
    assert tuple(got_x.shape) == (n, 256)
    assert_close(got_x.cpu().numpy(), want_x, TOL)
    for q in range(4):
        assert tuple(got_g['edges'][q].shape) == (quads[q].shape[0], 64)
        assert_close(got_g['edges'][q].cpu().numpy(), want_g['edges'][q], TOL)
    assert got_g['global'].shape[1] == (256 if with_gt else g + 256 + 64)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)
    # batched call: every sample equals the single-sample call bit for bit
    gb = graph_to_dev(graph)
    gb['edges'] = [e.unsqueeze(0).repeat(2, 1, 1) for e in gb['edges']]
    gb['global'] = gb['global'].repeat(2, 1)
    bx, bg = _layer(amd, W, 'L', with_gt)(to_dev(x).unsqueeze(0).repeat(2, 1, 1), gb)
    for b in range(2):
        assert torch.equal(bx[b], got_x) and torch.equal(bg['global'][b], got_g['global'][0])


def test_graph_upsampling_layer_256(amd, oracle):
    node_loc, ei, d, _ = oracle.synth_road_graph(33, 41, 0.5, 13)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    rng = np.random.Generator(np.random.PCG64(15))
    fine = {'node_loc': node_loc, 'edge_index': list(quads),
            'edges': [rng.random((q.shape[0], 16)).astype(np.float32) for q in quads],
            'global': rng.random((1, 113)).astype(np.float32)}
    _, p = oracle.cluster_ids_2x2(node_loc)
    _, coarse = oracle.graph_pooling_2x2(np.zeros((node_loc.shape[0], 1), np.float32), copy_graph(fine))
    coarse['edges'] = [rng.random((e.shape[0], 32)).astype(np.float32) for e in coarse['edge_index']]
    coarse['global'] = rng.random((1, 128)).astype(np.float32)
    x = (rng.random((p, 128)) - 0.2).astype(np.float32)
    W = oracle.make_weights(oracle.gn_layer_shapes('U/gcn', 128, 2, 128 + 113, 256), 3)
    want_x, want_g = oracle.graph_upsampling_graph(x, copy_graph(coarse), fine, W, 'U')
    up = amd.hybrid_gnn.GraphUpsampling2x2(256, method='graph', layer_type='geo_quadrant_gcn', use_global=True,
                                           use_bias=True, activation='relu')
    up.gcn.load_weights(W, 'U/gcn', 'cuda')
    got_x, got_g = up(to_dev(x), graph_to_dev(coarse), graph_to_dev(fine))
    assert tuple(got_x.shape) == want_x.shape == (node_loc.shape[0], 256)
    assert_close(got_x.cpu().numpy(), want_x, TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)


@pytest.mark.parametrize('units,depth', [(64, 3), (128, 2)])
def test_graph_unet_wide(amd, oracle, units, depth):
    """units=64, depth 3: blocks of width 128,128,256 / 128,128 / 256,128,128; units=128, depth 2:
    128,256 / 128,128 / 256,128."""
    x, graph, _ = make_case(oracle, 40, 52, 0.45)
    W = oracle.make_weights(oracle.graph_unet_shapes_for(units, depth=depth), 7)
    want_y, want_g = oracle.graph_unet_forward(x, copy_graph(graph), W, depth=depth)
    model = amd.GraphUNet(units, out_units=96, depth=depth, activation='relu', layer_type='geo_quadrant_gcn',
                          use_bias=True, use_global=True, output_activation=None).load_weights(W, 'cuda')
    got_y, got_g = model(to_dev(x), graph_to_dev(graph))
    assert tuple(got_y.shape) == want_y.shape
    assert_close(got_y.cpu().numpy(), want_y, TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)
    for q in range(4):
        assert_close(got_g['edges'][q].cpu().numpy(), want_g['edges'][q], TOL)


def test_graph_unet_uniform_256(amd, oracle):
    """GraphUNet(64, min_units=256): every block 256 wide (the wide stress configuration)."""
    x, graph, _ = make_case(oracle, 36, 44, 1.0, dense=True)
    W = oracle.make_weights(oracle.graph_unet_shapes_for(64, min_units=256), 9)
    assert W['bottleneck_gcn_1/node_transform/kernel'].shape == (256 + 256 + 256, 256)
    want_y, want_g = oracle.graph_unet_forward(x, copy_graph(graph), W)
    model = amd.GraphUNet(64, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                          use_bias=True, use_global=True, min_units=256).load_weights(W, 'cuda')
    got_y, got_g = model(to_dev(x), graph_to_dev(graph))
    assert_close(got_y.cpu().numpy(), want_y, TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)


@pytest.mark.parametrize('units,c_in,f_e,g', [(512, 256, 64, 256), (1024, 512, 128, 512), (128, 512, 128, 512)])
def test_gn_layer_wide(amd, oracle, units, c_in, f_e, g):
    """512- and 1024-wide layers (generic kernels), and the 128-wide bottleneck layer that follows a
    512-wide block (512 input columns, 128 edge features: its edge MLP takes the generic form too)."""
    node_loc, ei, d, _ = oracle.synth_road_graph(29, 35, 0.55, 23)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    rng = np.random.Generator(np.random.PCG64(24))
    n = node_loc.shape[0]
    x = (rng.random((n, c_in)) - 0.2).astype(np.float32)
    graph = {'edge_index': list(quads), 'node_loc': node_loc,
             'edges': [(rng.random((q.shape[0], f_e)) - 0.3).astype(np.float32) for q in quads],
             'global': rng.random((1, g)).astype(np.float32)}
    W = oracle.make_weights(oracle.gn_layer_shapes('L', c_in, f_e, g, units), 4)
    want_x, want_g = oracle.geo_quadrant_gcn(x, copy_graph(graph), W, 'L')
    layer = amd.hybrid_gnn.GeoQuadrantGCN(units, activation='relu', use_bias=True, use_global=True)
    layer.load_weights(W, 'L', 'cuda')
    got_x, got_g = layer(to_dev(x), graph_to_dev(graph))
    assert tuple(got_x.shape) == (n, units)
    assert_close(got_x.cpu().numpy(), want_x, TOL)
    for q in range(4):
        assert_close(got_g['edges'][q].cpu().numpy(), want_g['edges'][q], TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)
    gb = graph_to_dev(graph)                                   # batch of 3: rows equal the single call
    gb['edges'] = [e.unsqueeze(0).repeat(3, 1, 1) for e in gb['edges']]
    gb['global'] = gb['global'].repeat(3, 1)
    bx, bg = layer(to_dev(x).unsqueeze(0).repeat(3, 1, 1), gb)
    for b in range(3):
        assert torch.equal(bx[b], got_x) and torch.equal(bg['global'][b], got_g['global'][0])


@pytest.mark.parametrize('units', [128, 256])
def test_graph_unet_reference_width_schedule(amd, oracle, units):
    """The reference constructor's widths for units >= 128 at depth 3 (models/graph_unet.py:37-55):
    units=128 -> blocks 128, 256, 512 / 128, 128 / 512, 256, 128; units=256 -> 256, 512, 1024 / 128, 128 /
    1024, 512, 256.  One-call forward against the oracle, and against the layer-by-layer path."""
    x, graph, _ = make_case(oracle, 40, 52, 0.45)
    shapes = oracle.graph_unet_shapes_for(units)
    assert shapes['downsampling_layers/2/gcns/0/node_transform/kernel'][1] == 4 * units
    W = oracle.make_weights(shapes, 7)
    want_y, want_g = oracle.graph_unet_forward(x, copy_graph(graph), W)
    model = amd.GraphUNet(units, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                          use_bias=True, use_global=True, output_activation=None).load_weights(W, 'cuda')
    got_y, got_g = model(to_dev(x), graph_to_dev(graph))
    assert_close(got_y.cpu().numpy(), want_y, TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)
    for q in range(4):
        assert_close(got_g['edges'][q].cpu().numpy(), want_g['edges'][q], TOL)
    model.native_forward = False
    ly, lg = model(to_dev(x), graph_to_dev(graph))
    assert torch.equal(ly, got_y) and torch.equal(lg['global'], got_g['global'])


def test_unsupported_width(amd):
    layer = amd.hybrid_gnn.GeoQuadrantGCN(384, activation='relu', use_bias=False, use_global=True)
    x = torch.zeros((4, 128), device='cuda')
    e = [torch.zeros((0, 2), dtype=torch.int32, device='cuda')] * 4
    with pytest.raises(NotImplementedError):
        layer(x, {'edge_index': e, 'edges': [torch.zeros((0, 32), device='cuda')] * 4,
                  'global': torch.zeros((1, 16), device='cuda')})
