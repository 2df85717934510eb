"""Quadrant subgraph construction (reference data/data_utils.py:10-58)."""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from .._lib import call, current_stream, ptr, ptr4

# (dx, dy) of the directions N, NE, E, SE, S, SW, W, NW, self-loop
DIR_TO_EDGE = np.array([[0, -1], [1, -1], [1, 0], [1, 1], [0, 1], [-1, 1], [-1, 0], [-1, -1], [0, 0]])

# directions filed under the NE, SE, SW, NW quadrant; the fifth row is the self-loop bucket
QUADRANT_IDS = [[0, 1, 2], [2, 3, 4], [4, 5, 6], [6, 7, 0], [8, 8, 8]]


def sample_quadrant_subgraphs(edge_index, direction):
    """Split a directed pixel-grid edge list into the 4 quadrant subgraphs.

    edge_index int32 [E, 2], direction int [E] in 0..8.  Returns a tuple of 4 int32 [E_q, 2]
    tensors; edges keep their relative order, cardinal directions (0, 2, 4, 6) appear in both
    neighbouring quadrants and self-loops (8) are dropped."""
    if not (isinstance(edge_index, torch.Tensor) and edge_index.is_cuda):
        raise TypeError('edge_index must be a CUDA tensor')
    e = edge_index.to(torch.int32).contiguous()
    d = direction.to(device=e.device, dtype=torch.int64).contiguous()
    if e.dim() != 2 or e.shape[1] != 2 or d.shape != (e.shape[0],):
        raise ValueError(f'expected edge_index [E, 2] and direction [E], got {tuple(e.shape)} and {tuple(d.shape)}')
    n = int(e.shape[0])
    bufs = [torch.empty((max(n, 1), 2), dtype=torch.int32, device=e.device) for _ in range(4)]
    counts = _lib.I64x4()
    out = ptr4(bufs)
    call('gunet_quadrant_partition', ptr(e), ptr(d), n, C.byref(out), C.byref(counts),
         current_stream(e.device))
    return tuple(bufs[q][:int(counts[q])].clone() for q in range(4))


def image_to_graph(graph_data, return_img_mask=False, add_selfloops=True):
    """Directional maps -> graph (reference data/data_utils.py:95-150, mask_invalid_ids :84-93).

    graph_data: uint8 (or bool) CUDA tensor [8, H, W], non-zero where the pixel has an edge towards
    N, NE, E, SE, S, SW, W, NW.  Returns node_position int64 [N, 2] (row, col; raster order),
    edge_index int32 [E, 2] ordered by (source pixel, direction), direction int64 [E] (8 = the
    self-loop added per node), and the node mask bool [H, W] if return_img_mask."""
    if not (isinstance(graph_data, torch.Tensor) and graph_data.is_cuda):
        raise TypeError('graph_data must be a CUDA tensor')
    if graph_data.dim() != 3 or graph_data.shape[0] != 8:
        raise ValueError(f'graph_data must be [8, H, W], got {tuple(graph_data.shape)}')
    maps = (graph_data != 0).to(torch.uint8).contiguous() if graph_data.dtype != torch.uint8 else graph_data.contiguous()
    h, w = int(maps.shape[1]), int(maps.shape[2])
    dev = maps.device
    node_loc = torch.empty((h * w, 2), dtype=torch.int64, device=dev)
    edge_index = torch.empty((9 * h * w, 2), dtype=torch.int32, device=dev)
    direction = torch.empty((9 * h * w,), dtype=torch.int64, device=dev)
    mask = torch.empty((h, w), dtype=torch.uint8, device=dev)
    counts = (C.c_int64 * 2)()
    call('gunet_image_to_graph', ptr(maps), h, w, 1 if add_selfloops else 0, ptr(node_loc), ptr(edge_index),
         ptr(direction), ptr(mask), C.byref(counts), current_stream(dev))
    n, e = int(counts[0]), int(counts[1])
    out = (node_loc[:n].clone(), edge_index[:e].clone(), direction[:e].clone())
    return out + (mask.bool(),) if return_img_mask else out
