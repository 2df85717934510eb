"""Device-resident index structures ("plans") of the Graph U-Net forward path.

Everything here depends only on the static city graph, so it is built once (on
the GPU, through the C ABI) and cached by the identity of the index tensors it
was built from (storage address, row count and torch's in-place version
counter: editing an index tensor in place gives a new key and a fresh plan).
The reference recomputes all of it inside every forward
(layers/hybrid_gnn.py:410-447, :267-280, :336-357; tf.unique and boolean masks).

  QuadGraph     four reference-ordered edge lists + receiver-sorted CSR per list
  PoolPlan      2x2 cluster ids, pooled node positions, pooled edge lists and
                the member lists used by the segment-max kernels
  UpsamplePlan  parent ids and the QuadGraph of the 'graph' upsampling method
  UNetPlan      all of the above for every level of a U-Net, owned by the library
                (gunet_plan_create) and consumed by the whole-model entry point
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


def _lookup(cache, key):
    """Cache hit, marked as the most recently used entry (eviction takes the least recently used)."""
    hit = cache.get(key)
    if hit is not None:
        cache.move_to_end(key)
    return hit


def _tensor_key(t):
    """Identity of an index tensor's contents as far as torch can tell: its storage address, row
    count and in-place version counter (bumped by every in-place edit through torch).  Tensors
    made under torch.inference_mode() keep no counter (reading it raises); they get -1."""
    try:
        version = int(t._version)
    except RuntimeError:
        version = -1
    return (t.data_ptr(), int(t.shape[0]), version)


def clear_caches():
    _quad_cache.clear()
    _pool_cache.clear()
    _up_cache.clear()
    _unet_cache.clear()


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
    return tuple(_tensor_key(e) for e in edge_index)


def get_quadgraph(edge_index, n_nodes):
    """Cached QuadGraph for a list of 4 edge tensors (keyed by their storage)."""
    key = (_edge_key(edge_index), int(n_nodes))
    hit = _lookup(_quad_cache, key)
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
    key = (_tensor_key(node_loc), _edge_key(edge_index))
    hit = _lookup(_pool_cache, key)
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
    key = (_tensor_key(coarse_loc), _tensor_key(fine_loc))
    hit = _lookup(_up_cache, key)
    if hit is not None:
        return hit
    plan = UpsamplePlan(coarse_loc, fine_loc)
    plan._keepalive = [coarse_loc, fine_loc]
    return _remember(_up_cache, key, plan)


class _DeviceView:
    """A typed window on device memory the library owns, importable with torch.as_tensor."""

    def __init__(self, address, shape, typestr, owner):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(address), False),
                                         'version': 2}
        self._owner = owner


_TYPESTR = {torch.int32: '<i4', torch.int64: '<i8'}


class UNetPlan:
    """Handle of a library-owned plan: quadrant graphs, pooling maps and upsampling graphs of all
    depth + 1 levels of one city graph (what QuadGraph / PoolPlan / UpsamplePlan hold level by
    level), built by one call and passed to gunet_unet_forward_f32."""

    def __init__(self, node_loc, edge_index, depth):
        node_loc = _check_node_loc(node_loc)
        edge_index = [_check_edge_list(e) for e in edge_index]
        if len(edge_index) != 4:
            raise ValueError('a quadrant graph needs exactly 4 edge lists')
        self.device = node_loc.device
        self.depth = int(depth)
        self.handle = C.c_void_p()
        lists = ptr4(edge_index)
        counts = _lib.I64x4(*[int(e.shape[0]) for e in edge_index])
        with torch.cuda.device(self.device):
            call('gunet_plan_create', ptr(node_loc), int(node_loc.shape[0]), C.byref(lists), C.byref(counts),
                 self.depth, C.byref(self.handle), current_stream(self.device))
        self.n_nodes, self.n_edges = [], []
        for level in range(self.depth + 1):
            s = self._level_struct(level)[0]
            self.n_nodes.append(int(s.n_nodes))
            self.n_edges.append([int(s.n_edges[q]) for q in range(4)])

    def __del__(self):
        handle, self.handle = getattr(self, 'handle', None), None
        if handle and _lib is not None and getattr(_lib, 'lib', None) is not None:
            _lib.lib.gunet_plan_destroy(handle)

    def _view(self, address, shape, dtype):
        if any(d == 0 for d in shape):
            return torch.empty(shape, dtype=dtype, device=self.device)
        return torch.as_tensor(_DeviceView(address, shape, _TYPESTR[dtype], self), device=self.device)

    def _level_struct(self, level):
        s, loc = QuadGraphStruct(), C.c_void_p()
        call('gunet_plan_level', self.handle, level, C.byref(s), C.byref(loc))
        return s, loc

    def _graph_views(self, s):
        n = int(s.n_nodes)
        return {'n_nodes': n,
                'edge_index': [self._view(s.edge_index[q], (int(s.n_edges[q]), 2), torch.int32) for q in range(4)],
                'rowptr': [self._view(s.rowptr[q], (n + 1,), torch.int32) for q in range(4)],
                'perm': [self._view(s.perm[q], (int(s.n_edges[q]),), torch.int32) for q in range(4)],
                'in_table': self._view(s.in_table, (n, 4, 3), torch.int32) if s.in_table else None}

    def level(self, level):
        """Views of level `level` (after that many poolings): edge lists, CSRs, in-edge table, node_loc."""
        s, loc = self._level_struct(level)
        out = self._graph_views(s)
        out['node_loc'] = self._view(loc.value, (out['n_nodes'], 2), torch.int64)
        return out

    def pooling(self, level):
        """(cluster int32 [n_level], [dup_map int32 [E_q]] * 4) of the pooling level -> level + 1."""
        cluster, dup = C.c_void_p(), _lib.VoidP4()
        call('gunet_plan_pooling', self.handle, level, C.byref(cluster), C.byref(dup))
        return (self._view(cluster.value, (self.n_nodes[level],), torch.int32),
                [self._view(dup[q], (self.n_edges[level][q],), torch.int32) for q in range(4)])

    def upsampling(self, level):
        """(parent int32 [n_level], graph views) of the upsampling level + 1 -> level."""
        s, parent = QuadGraphStruct(), C.c_void_p()
        call('gunet_plan_upsampling', self.handle, level, C.byref(s), C.byref(parent))
        return self._view(parent.value, (self.n_nodes[level],), torch.int32), self._graph_views(s)


_unet_cache = OrderedDict()


def get_unet_plan(node_loc, edge_index, depth):
    key = (_tensor_key(node_loc), _edge_key(edge_index), int(depth))
    hit = _lookup(_unet_cache, key)
    if hit is not None:
        return hit
    plan = UNetPlan(node_loc, edge_index, depth)
    plan._keepalive = [node_loc] + list(edge_index)
    return _remember(_unet_cache, key, plan)


_workspaces = {}
_model_workspaces = {}
_scopes = []


class WorkspaceScope:
    """Scratch buffers of their own for everything run inside `with scope:`.  The shared buffers
    below are replaced by a larger allocation when a later call needs more, which would pull the
    memory from under a captured HIP graph; a capture runs inside a scope and keeps it."""

    def __init__(self):
        self.layer, self.model = {}, {}

    def __enter__(self):
        _scopes.append(self)
        return self

    def __exit__(self, *exc):
        _scopes.remove(self)
        return False


_WORKSPACE_STREAMS = 4   # buffers kept per table: the streams used most recently


def _grown(table, nbytes, device, slack):
    # one buffer per (device, stream): kernels of one stream use it in order, two streams never
    # share one, so forwards issued on different streams of a device cannot race on scratch.
    # Only the most recently used streams keep theirs: a server that runs forwards on many or
    # short-lived streams would otherwise hold one whole-model buffer per stream handle ever seen.
    # An evicted buffer goes back to torch's allocator, which hands it out again only after the
    # work queued on its stream has finished with it.
    key = (device.type, device.index, int(torch.cuda.current_stream(device).cuda_stream))
    buf = table.pop(key, None)
    if buf is None or buf.numel() < nbytes:
        buf = None                   # let go of the old buffer before asking for the new one
        buf = torch.empty(int(nbytes * slack) + 256, dtype=torch.uint8, device=device)
    table[key] = buf                 # (re-)inserted last = most recently used
    while len(table) > _WORKSPACE_STREAMS * max(1, torch.cuda.device_count()):
        del table[next(iter(table))]
    return buf


def workspace(nbytes, device):
    """Scratch buffer of the current stream of `device`; it only grows and is reused in stream order."""
    return _grown(_scopes[-1].layer if _scopes else _workspaces, nbytes, device, 1.25)


def model_workspace(nbytes, device):
    """Buffer of the whole-model entry point (all intermediates of one forward, skip tensors
    included) for the current stream of `device`; grows to the largest request."""
    return _grown(_scopes[-1].model if _scopes else _model_workspaces, nbytes, device, 1.0)


def model_workspace_of(device):
    """The largest whole-model buffer held for `device` (any stream), or None."""
    bufs = [b for k, b in _model_workspaces.items() if k[:2] == (device.type, device.index) and b is not None]
    return max(bufs, key=lambda b: b.numel()) if bufs else None


def release_workspaces():
    """Drop every shared scratch buffer (they are re-allocated on demand) and the library's cache of
    temporary blocks of the plan functions."""
    _workspaces.clear()
    _model_workspaces.clear()
    call('gunet_release_temporaries')
