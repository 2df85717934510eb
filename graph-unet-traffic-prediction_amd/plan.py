"""Device-resident index structures ("plans") of the Graph U-Net forward path.

Everything here depends only on the static city graph, so it is built once (on
the GPU, through the C ABI) and cached by the identity of the index tensors it
was built from.  The reference recomputes all of it inside every forward
(layers/hybrid_gnn.py:410-447, :267-280, :336-357; tf.unique and boolean masks).

  QuadGraph     four reference-ordered edge lists + receiver-sorted CSR per list
  PoolPlan      2x2 cluster ids, pooled node positions, pooled edge lists and
                the member lists used by the segment-max kernels
  UpsamplePlan  parent ids and the QuadGraph of the 'graph' upsampling method
"""
import ctypes as C
from collections import OrderedDict

import torch

from . import _lib
from ._lib import QuadGraphStruct, call, current_stream, ptr, ptr4

_CACHE_LIMIT = 256
_quad_cache = OrderedDict()
_pool_cache = OrderedDict()
_up_cache = OrderedDict()


def _remember(cache, key, value):
    cache[key] = value
    while len(cache) > _CACHE_LIMIT:
        cache.popitem(last=False)
    return value


def clear_caches():
    _quad_cache.clear()
    _pool_cache.clear()
    _up_cache.clear()


def _check_edge_list(e):
    if not (isinstance(e, torch.Tensor) and e.is_cuda):
        raise TypeError('edge_index entries must be CUDA tensors')
    if e.dtype != torch.int32 or e.dim() != 2 or e.shape[1] != 2:
        raise ValueError(f'edge_index entries must be int32 [E, 2], got {e.dtype} {tuple(e.shape)}')
    return e.contiguous()


def _check_node_loc(node_loc):
    if not (isinstance(node_loc, torch.Tensor) and node_loc.is_cuda):
        raise TypeError('node_loc must be a CUDA tensor')
    if node_loc.dtype != torch.int64 or node_loc.dim() != 2 or node_loc.shape[1] != 2:
        raise ValueError(f'node_loc must be int64 [N, 2], got {node_loc.dtype} {tuple(node_loc.shape)}')
    return node_loc.contiguous()


def csr_from_keys(keys, n_segments):
    """(rowptr int32[n_segments+1], perm int32[#valid]) for int32 keys; key < 0 = no segment."""
    keys = keys.contiguous()
    n = keys.numel()
    rowptr = torch.empty(n_segments + 1, dtype=torch.int32, device=keys.device)
    perm = torch.empty(max(n, 1), dtype=torch.int32, device=keys.device)
    call('gunet_csr_from_keys', ptr(keys), n, n_segments, ptr(rowptr), ptr(perm),
         current_stream(keys.device))
    return rowptr, perm


class QuadGraph:
    """Four quadrant edge lists (sender, receiver) over n_nodes nodes plus, per list, the CSR
    sorted by receiver that the aggregation kernel walks."""

    def __init__(self, edge_index, n_nodes):
        self.edge_index = [_check_edge_list(e) for e in edge_index]
        if len(self.edge_index) != 4:
            raise ValueError('a quadrant graph needs exactly 4 edge lists')
        self.n_nodes = int(n_nodes)
        self.n_edges = [int(e.shape[0]) for e in self.edge_index]
        self.device = self.edge_index[0].device
        self.rowptr, self.perm = [], []
        stream = current_stream(self.device)
        for e in self.edge_index:
            rowptr = torch.empty(self.n_nodes + 1, dtype=torch.int32, device=self.device)
            perm = torch.empty(max(e.shape[0], 1), dtype=torch.int32, device=self.device)
            call('gunet_csr_build', ptr(e), e.shape[0], self.n_nodes, ptr(rowptr), ptr(perm), stream)
            self.rowptr.append(rowptr)
            self.perm.append(perm)
        s = QuadGraphStruct()
        s.n_nodes = self.n_nodes
        for q in range(4):
            s.n_edges[q] = self.n_edges[q]
            s.edge_index[q] = self.edge_index[q].data_ptr()
            s.rowptr[q] = self.rowptr[q].data_ptr()
            s.perm[q] = self.perm[q].data_ptr()
        # per-node table of the (at most 3) incoming edges per quadrant, when the graph allows it
        self.in_table = None
        if self.n_nodes > 0:
            table = torch.empty((self.n_nodes, 4, 3), dtype=torch.int32, device=self.device)
            max_deg = C.c_int32(0)
            call('gunet_in_edge_table', C.byref(s), ptr(table), C.byref(max_deg), stream)
            self.max_in_degree = int(max_deg.value)
            if self.max_in_degree <= 3:
                self.in_table = table
                s.in_table = table.data_ptr()
        self.struct = s
        self._ws_bytes = {}

    def workspace_bytes(self, batch, units):
        key = (batch, units)
        if key not in self._ws_bytes:
            self._ws_bytes[key] = int(_lib.lib.gunet_gn_layer_workspace_bytes(
                C.byref(self.struct), batch, units))
        return self._ws_bytes[key]


def _edge_key(edge_index):
    return tuple((e.data_ptr(), int(e.shape[0])) for e in edge_index)


def get_quadgraph(edge_index, n_nodes):
    """Cached QuadGraph for a list of 4 edge tensors (keyed by their storage)."""
    key = (_edge_key(edge_index), int(n_nodes))
    hit = _quad_cache.get(key)
    if hit is not None:
        return hit
    plan = QuadGraph(edge_index, n_nodes)
    plan._keepalive = list(edge_index)   # the key is made of their addresses
    return _remember(_quad_cache, key, plan)


class PoolPlan:
    """Index part of graph_pooling_2x2 (layers/hybrid_gnn.py:410-447)."""

    def __init__(self, node_loc, edge_index):
        node_loc = _check_node_loc(node_loc)
        edge_index = [_check_edge_list(e) for e in edge_index]
        dev = node_loc.device
        stream = current_stream(dev)
        n = int(node_loc.shape[0])
        self.n_nodes = n
        self.node_loc = node_loc
        self.src_edge_index = edge_index
        self.cluster = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
        pooled = torch.empty((max(n, 1), 2), dtype=torch.int64, device=dev)
        count = C.c_int64(0)
        call('gunet_cluster_ids', ptr(node_loc), n, ptr(self.cluster), ptr(pooled), C.byref(count),
             stream)
        self.n_pooled = int(count.value)
        self.pooled_loc = pooled[:self.n_pooled].clone()
        self.cluster_rowptr, self.cluster_perm = csr_from_keys(self.cluster, self.n_pooled)
        self.edge_index, self.dup_map, self.dup_rowptr, self.dup_perm = [], [], [], []
        for e in edge_index:
            ne = int(e.shape[0])
            new_edges = torch.empty((max(ne, 1), 2), dtype=torch.int32, device=dev)
            dup_map = torch.empty(max(ne, 1), dtype=torch.int32, device=dev)[:ne]
            n_new = C.c_int64(0)
            if ne > 0:
                call('gunet_pool_edges', ptr(e), ne, ptr(self.cluster), max(n, 1), ptr(new_edges),
                     ptr(dup_map), C.byref(n_new), stream)
            n_new = int(n_new.value)
            rowptr, perm = csr_from_keys(dup_map, n_new)
            self.edge_index.append(new_edges[:n_new].clone())
            self.dup_map.append(dup_map)
            self.dup_rowptr.append(rowptr)
            self.dup_perm.append(perm)
        self.n_pooled_edges = [int(e.shape[0]) for e in self.edge_index]


def get_pool_plan(node_loc, edge_index):
    key = (node_loc.data_ptr(), int(node_loc.shape[0]), _edge_key(edge_index))
    hit = _pool_cache.get(key)
    if hit is not None:
        return hit
    plan = PoolPlan(node_loc, edge_index)
    plan._keepalive = [node_loc] + list(edge_index)
    return _remember(_pool_cache, key, plan)


class UpsamplePlan:
    """Index part of GraphUpsampling2x2.call, method 'graph' (layers/hybrid_gnn.py:267-280,
    336-357): every fine node i receives one edge (parent[i] -> n_coarse + i), filed under the
    quadrant given by its place in the parent's 2x2 cell."""

    def __init__(self, coarse_loc, fine_loc):
        coarse_loc = _check_node_loc(coarse_loc)
        fine_loc = _check_node_loc(fine_loc)
        dev = fine_loc.device
        self.coarse_loc, self.fine_loc = coarse_loc, fine_loc
        self.n_coarse = int(coarse_loc.shape[0])
        self.n_fine = int(fine_loc.shape[0])
        nf = max(self.n_fine, 1)
        self.parent = torch.empty(nf, dtype=torch.int32, device=dev)[:self.n_fine]
        bufs = [torch.empty((nf, 2), dtype=torch.int32, device=dev) for _ in range(4)]
        counts = _lib.I64x4()
        out = ptr4(bufs)
        call('gunet_upsample_edges', ptr(coarse_loc), self.n_coarse, ptr(fine_loc), self.n_fine,
             ptr(self.parent), C.byref(out), C.byref(counts), current_stream(dev))
        self.edge_index = [bufs[q][:int(counts[q])].clone() for q in range(4)]
        self.quadgraph = QuadGraph(self.edge_index, self.n_coarse + self.n_fine)


def get_upsample_plan(coarse_loc, fine_loc):
    key = (coarse_loc.data_ptr(), int(coarse_loc.shape[0]), fine_loc.data_ptr(),
           int(fine_loc.shape[0]))
    hit = _up_cache.get(key)
    if hit is not None:
        return hit
    plan = UpsamplePlan(coarse_loc, fine_loc)
    plan._keepalive = [coarse_loc, fine_loc]
    return _remember(_up_cache, key, plan)


_workspaces = {}


def workspace(nbytes, device):
    """A per-device scratch buffer that only grows; kernels on one stream reuse it in order."""
    key = (device.type, device.index)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf
