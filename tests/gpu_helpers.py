"""Shared helpers of the GPU parity tests: oracle <-> torch conversion."""
import numpy as np
import torch


def to_dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def graph_to_dev(graph):
    return {'edge_index': [to_dev(e, torch.int32) for e in graph['edge_index']],
            'edges': [to_dev(e, torch.float32) for e in graph['edges']],
            'node_loc': to_dev(graph['node_loc'], torch.int64),
            'global': to_dev(graph['global'], torch.float32)}


def copy_graph(graph):
    """Fresh containers, same arrays: the oracle mutates the dict it is given."""
    return {'edge_index': list(graph['edge_index']), 'edges': list(graph['edges']),
            'node_loc': graph['node_loc'], 'global': graph['global']}


def make_case(oracle, height, width, density, graph_seed=0, sample_seed=7, dense=False):
    """Synthetic (graph, sample) in the layout GraphBaseModel._call hands to GraphUNet."""
    if dense:
        node_loc, edge_index, direction, _ = oracle.dense_grid_graph(height, width)
    else:
        node_loc, edge_index, direction, _ = oracle.synth_road_graph(height, width, density, graph_seed)
    quads = oracle.sample_quadrant_subgraphs(edge_index, direction)
    enc = oracle.synth_street_encoding(node_loc.shape[0], 5)
    seed_nodes, time_enc, weekday = oracle.synth_sample(node_loc, quads, sample_seed)
    x, graph = oracle.assemble_inputs(seed_nodes, enc, quads, node_loc, time_enc, weekday)
    return x, graph, (edge_index, direction)
