"""Workload configurations beyond the headline one, at their full sizes.

  * dense full-grid graph (every pixel of the 495x436 grid a node, ~1.7M edges) through one
    256-wide GN layer: direct comparison with the oracle plus size-independent properties
    (batch rows identical bit for bit, which is also run-to-run reproducibility);
  * several cities through one model in one process, sharded by city as the multi-GPU launcher
    shards them: per-sample results must not depend on which other cities share the process.

Tolerance (fp32 path): max|got - want| / max|want| <= 1e-5."""
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


def test_dense_grid_width256_layer(amd, oracle):
    node_loc, ei, d, _ = oracle.dense_grid_graph(495, 436)
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    n = node_loc.shape[0]
    assert n == 495 * 436 and sum(q.shape[0] for q in quads) > 2_000_000
    rng = np.random.Generator(np.random.PCG64(31))
    x = (rng.random((n, 256), dtype=np.float32) - np.float32(0.2))
    graph = {'edge_index': list(quads), 'node_loc': node_loc,
             'edges': [rng.random((q.shape[0], 64), dtype=np.float32) for q in quads],
             'global': rng.random((1, 256), dtype=np.float32)}
    W = oracle.make_weights(oracle.gn_layer_shapes('L', 256, 64, 256, 256), 5)
    want_x, want_g = oracle.geo_quadrant_gcn(x, copy_graph(graph), W, 'L')
    layer = amd.hybrid_gnn.GeoQuadrantGCN(256, activation='relu', use_bias=True, use_global=True)
    layer.load_weights(W, 'L', 'cuda')
    g = graph_to_dev(graph)
    ei_dev = g['edge_index']
    got_x, got_g = layer(to_dev(x), g)
    assert_close(got_x.cpu().numpy(), want_x, TOL)
    assert_close(got_g['global'].cpu().numpy(), want_g['global'], TOL)
    for q in range(4):
        assert_close(got_g['edges'][q].cpu().numpy(), want_g['edges'][q], TOL)
    # batch of 2 = the single call twice, bit for bit (also a second run of the same kernels)
    gb = graph_to_dev(graph)
    gb['edge_index'] = ei_dev
    gb['edges'] = [e.unsqueeze(0).repeat(2, 1, 1) for e in gb['edges']]
    gb['global'] = gb['global'].repeat(2, 1)
    bx, bg = layer(to_dev(x).unsqueeze(0).repeat(2, 1, 1), gb)
    for b in range(2):
        assert torch.equal(bx[b], got_x)
        assert torch.equal(bg['global'][b], got_g['global'][0])
        for q in range(4):
            assert torch.equal(bg['edges'][q][b], got_g['edges'][q])


def test_six_cities_one_process(amd, oracle):
    """Six small 'cities' (different graphs), 2 samples each, run in the order the sharding gives
    a rank; each prediction equals the oracle's and does not depend on its neighbours."""
    from gunet_amd import sharding
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    model = amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                          use_bias=True, use_global=True).load_weights(W, 'cuda')
    sizes = [(24, 30, 0.6), (31, 22, 0.5), (40, 36, 0.45), (18, 50, 0.55), (27, 27, 0.7), (36, 20, 0.5)]
    cities = []
    for c, (h, w, dens) in enumerate(sizes):
        node_loc, ei, d, _ = oracle.synth_road_graph(h, w, dens, 40 + c)
        quads = oracle.sample_quadrant_subgraphs(ei, d)
        enc = oracle.synth_street_encoding(node_loc.shape[0], 50 + c)
        dev = {'edge_index': [to_dev(q, torch.int32) for q in quads], 'node_loc': to_dev(node_loc, torch.int64)}
        cities.append((node_loc, quads, enc, dev))
    sample_city = [0, 3, 1, 5, 2, 4, 4, 2, 5, 1, 3, 0]
    results = {}
    for world in (1, 3):
        for rank in range(world):
            for i in sharding.shard_by_city(sample_city, world, rank):
                node_loc, quads, enc, dev = cities[sample_city[i]]
                sn, te, wd = oracle.synth_sample(node_loc, quads, 200 + i, time_idx=7 * i, weekday=i)
                x, graph = oracle.assemble_inputs(sn, enc, quads, node_loc, te, wd)
                g = graph_to_dev(graph)
                g['edge_index'], g['node_loc'] = dev['edge_index'], dev['node_loc']   # one plan per city
                y = model.predict(to_dev(x), g)
                if world == 1:
                    want, _ = oracle.predict(x, copy_graph(graph), W)
                    assert_close(y.cpu().numpy(), want, TOL, str(i))
                    results[i] = y
                else:
                    assert torch.equal(y, results[i]), (rank, i)
