"""Generate the golden vectors under tests/golden/ with the CPU oracle.

The reference itself cannot run here (TensorFlow 2.6 is not installable), so these vectors pin the
oracle against silent drift and give the GPU tests fixed inputs/outputs; they are NOT outputs of
the reference.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..', 'oracle'))
import gunet_oracle as O  # noqa: E402


def unet_case(name, height, width, density, graph_seed, sample_seed):
    node_loc, ei, d, _ = O.synth_road_graph(height, width, density, graph_seed)
    quads = O.sample_quadrant_subgraphs(ei, d)
    enc = O.synth_street_encoding(node_loc.shape[0], 5)
    sn, te, wd = O.synth_sample(node_loc, quads, sample_seed, time_idx=3, weekday=2)
    x, graph = O.assemble_inputs(sn, enc, quads, node_loc, te, wd)
    W = O.make_weights(O.graph_unet_shapes(), 1)
    keep = {}
    g_in = {k: (list(v) if isinstance(v, list) else v) for k, v in graph.items()}
    y, g_out = O.graph_unet_forward(x, g_in, W, keep=keep)
    out = {'height': height, 'width': width, 'density': density, 'graph_seed': graph_seed,
           'sample_seed': sample_seed, 'weights_seed': 1,
           'node_loc': node_loc, 'edge_index': ei, 'direction': d, 'x': x, 'global': graph['global'],
           'y': y, 'global_out': g_out['global']}
    for q in range(4):
        out[f'quad{q}'] = quads[q]
        out[f'edges_in{q}'] = graph['edges'][q]
        out[f'edges_out{q}'] = g_out['edges'][q]
    for k, v in keep.items():
        out['keep_' + k] = v
    # the index chain
    gi = {'node_loc': node_loc, 'edge_index': list(quads),
          'edges': [np.zeros((q.shape[0], 1), np.float32) for q in quads]}
    xx = np.zeros((node_loc.shape[0], 1), np.float32)
    for lvl in range(3):
        ids, p = O.cluster_ids_2x2(gi['node_loc'])
        xx, gi = O.graph_pooling_2x2(xx, gi)
        out[f'cluster{lvl}'] = ids
        out[f'pooled_loc{lvl}'] = gi['node_loc']
        for q in range(4):
            out[f'pooled_edges{lvl}_{q}'] = gi['edge_index'][q]
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'nodes', node_loc.shape[0], 'edges', [q.shape[0] for q in quads])


def layer_case(name):
    node_loc, ei, d, _ = O.synth_road_graph(14, 18, 0.55, 21)
    quads = O.sample_quadrant_subgraphs(ei, d)
    rng = np.random.Generator(np.random.PCG64(22))
    n = node_loc.shape[0]
    x = rng.random((n, 128)).astype(np.float32)
    graph = {'edge_index': list(quads), 'node_loc': node_loc,
             'edges': [rng.random((q.shape[0], 16)).astype(np.float32) for q in quads],
             'global': rng.random((1, 113)).astype(np.float32)}
    W = O.make_weights(O.gn_layer_shapes('L', 128, 16, 113), 1)
    keep = {}
    g_in = {k: (list(v) if isinstance(v, list) else v) for k, v in graph.items()}
    y, g_out = O.geo_quadrant_gcn(x, g_in, W, 'L', keep=keep)
    out = {'node_loc': node_loc, 'x': x, 'global': graph['global'], 'y': y, 'global_out': g_out['global']}
    for q in range(4):
        out[f'quad{q}'] = quads[q]
        out[f'edges_in{q}'] = graph['edges'][q]
        out[f'edges_out{q}'] = g_out['edges'][q]
    for k, v in keep.items():
        out['keep_' + k] = v
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'nodes', n)


if __name__ == '__main__':
    unet_case('unet_9x14', 9, 14, 0.6, 3, 7)
    unet_case('unet_24x30', 24, 30, 0.5, 4, 8)
    layer_case('gn_layer_14x18')
