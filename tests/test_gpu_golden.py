"""HIP path against the committed golden vectors (tests/golden/*.npz, made by make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_close, rel_err
from gpu_helpers import to_dev

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.mark.parametrize('name', ['unet_9x14', 'unet_24x30'])
def test_unet_golden(oracle, name):
    import gunet_amd
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    quads = gunet_amd.data_utils.sample_quadrant_subgraphs(to_dev(z['edge_index'], torch.int32),
                                                           to_dev(z['direction'], torch.int64))
    for q in range(4):
        assert np.array_equal(quads[q].cpu().numpy(), z[f'quad{q}'])
    W = oracle.make_weights(oracle.graph_unet_shapes(), int(z['weights_seed']))
    model = gunet_amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                                use_bias=True, use_global=True).load_weights(W, 'cuda')
    graph = {'edge_index': list(quads), 'edges': [to_dev(z[f'edges_in{q}']) for q in range(4)],
             'node_loc': to_dev(z['node_loc'], torch.int64), 'global': to_dev(z['global'])}
    y, g = model(to_dev(z['x']), graph)
    assert_close(y.cpu().numpy(), z['y'], TOL)
    assert_close(g['global'].cpu().numpy(), z['global_out'], TOL)
    for q in range(4):
        assert_close(g['edges'][q].cpu().numpy(), z[f'edges_out{q}'], TOL)
    loc, edges = to_dev(z['node_loc'], torch.int64), list(quads)
    for lvl in range(3):
        pp = gunet_amd.plan.get_pool_plan(loc, edges)
        assert np.array_equal(pp.cluster.cpu().numpy(), z[f'cluster{lvl}'])
        assert np.array_equal(pp.pooled_loc.cpu().numpy(), z[f'pooled_loc{lvl}'])
        for q in range(4):
            assert np.array_equal(pp.edge_index[q].cpu().numpy(), z[f'pooled_edges{lvl}_{q}'])
        loc, edges = pp.pooled_loc, pp.edge_index


def test_gn_layer_golden(oracle):
    import gunet_amd
    z = np.load(os.path.join(GOLDEN, 'gn_layer_14x18.npz'))
    W = oracle.make_weights(oracle.gn_layer_shapes('L', 128, 16, 113), 1)
    layer = gunet_amd.hybrid_gnn.GeoQuadrantGCN(128, activation='relu', use_bias=True, use_global=True)
    layer.load_weights(W, 'L', 'cuda')
    graph = {'edge_index': [to_dev(z[f'quad{q}'], torch.int32) for q in range(4)],
             'edges': [to_dev(z[f'edges_in{q}']) for q in range(4)],
             'node_loc': to_dev(z['node_loc'], torch.int64), 'global': to_dev(z['global'])}
    y, g = layer(to_dev(z['x']), graph)
    assert_close(y.cpu().numpy(), z['y'], TOL)
    assert_close(g['global'].cpu().numpy(), z['global_out'], TOL)
    for q in range(4):
        assert_close(g['edges'][q].cpu().numpy(), z[f'edges_out{q}'], TOL)
