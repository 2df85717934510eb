"""Host-side mirror of the reference's layers/hybrid_gnn.py on top of libgunet_hip.so.

Same class names, constructor arguments, call signatures and graph-dict
convention as the reference, so a caller of the reference layers can switch
modules.  Tensors are torch CUDA tensors; all arithmetic runs in the HIP library.

graph dict (models/base_model.py:302-312 in the reference):
    'edge_index': list of 4 int32 [E_q, 2] (sender, receiver), quadrants NE, SE, SW, NW
    'edges':      list of 4 float32 [E_q, F_e]
    'node_loc':   int64 [N, 2] (row, col)
    'global':     float32 [1, G]
Batched extension (several samples on one graph): x [B, N, C], edges [B, E_q, F_e],
global [B, G]; 2-D x is the reference's single-sample form.  With a batched x, 2-D edge tensors
[E_q, F_e] are taken as shared by all samples (the street-map features depend on the city only).
"""
import ctypes as C
import math

import torch

from .. import _lib, plan as _plan
from .._lib import GnWeightsStruct, call, current_stream, ptr, ptr4

DEBUGGING = False

# Keras activation names (the reference passes them to tf.keras.layers.Dense, hybrid_gnn.py:25).  relu
# and linear run on the tuned kernels; the others on the library's generic ones, layer by layer.
_ACTIVATIONS = {None: _lib.ACT_LINEAR, 'linear': _lib.ACT_LINEAR, 'relu': _lib.ACT_RELU, 'tanh': _lib.ACT_TANH,
                'sigmoid': _lib.ACT_SIGMOID, 'elu': _lib.ACT_ELU, 'softplus': _lib.ACT_SOFTPLUS,
                'selu': _lib.ACT_SELU, 'swish': _lib.ACT_SWISH, 'gelu': _lib.ACT_GELU, 'softsign': _lib.ACT_SOFTSIGN,
                'exponential': _lib.ACT_EXPONENTIAL, 'hard_sigmoid': _lib.ACT_HARD_SIGMOID}
_TUNED = (_lib.ACT_LINEAR, _lib.ACT_RELU)


def _activation_code(name):
    if name not in _ACTIVATIONS:
        raise NotImplementedError(f'Activation "{name}" is not implemented ({", ".join(str(k) for k in _ACTIVATIONS)}).')
    return _ACTIVATIONS[name]


def _as_f32(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError(f'{what} must be a CUDA tensor')
    if t.dtype != torch.float32:
        raise TypeError(f'{what} must be float32, got {t.dtype}')
    return t.contiguous()


def _glorot(shape, device, generator=None):
    limit = math.sqrt(6.0 / (shape[0] + shape[1]))
    w = torch.rand(shape, dtype=torch.float32, generator=generator) * (2 * limit) - limit
    return w.to(device)


class Dense:
    """tf.keras.layers.Dense: y = activation(x @ kernel + bias), kernel [in, out]
    (used as in_layer / out_model, models/graph_unet.py:37,58)."""

    def __init__(self, units, activation=None, use_bias=True):
        self.units = units
        self.activation = activation
        self._act = _activation_code(activation)
        self.use_bias = use_bias
        self.kernel = None
        self.bias = None

    def build(self, in_features, device):
        if self.kernel is None:
            self.kernel = _glorot((in_features, self.units), device)
            self.bias = torch.zeros(self.units, dtype=torch.float32, device=device) if self.use_bias else None

    def load_weights(self, flat, prefix, device):
        self.kernel = torch.as_tensor(flat[prefix + '/kernel'], dtype=torch.float32).to(device).contiguous()
        self.bias = (torch.as_tensor(flat[prefix + '/bias'], dtype=torch.float32).to(device).contiguous()
                     if self.use_bias else None)

    def named_weights(self, prefix):
        out = {prefix + '/kernel': self.kernel}
        if self.use_bias:
            out[prefix + '/bias'] = self.bias
        return out

    def __call__(self, x, activation=Ellipsis):
        """activation overrides the layer's own for this call (used to fuse a following ReLU)."""
        act = self._act if activation is Ellipsis else _activation_code(activation)
        x = _as_f32(x, 'x')
        self.build(x.shape[-1], x.device)
        if x.shape[-1] != self.kernel.shape[0]:
            raise ValueError(f'Dense expects {self.kernel.shape[0]} input features, got {x.shape[-1]}')
        rows = x.numel() // x.shape[-1]
        out = torch.empty(x.shape[:-1] + (self.units,), dtype=torch.float32, device=x.device)
        scratch = _plan.workspace(_lib.lib.gunet_dense_workspace_bytes(x.shape[-1], self.units), x.device)
        call('gunet_dense_forward_f32', ptr(x), rows, x.shape[-1], ptr(self.kernel), ptr(self.bias),
             self.units, act, ptr(out), ptr(scratch), scratch.numel(), current_stream(x.device))
        return out


class GeoQuadrantGCN:
    """Quadrant-directional full GN block (reference layers/hybrid_gnn.py:10-137).

    Per direction q: node embedding Dense(units/4), edge MLP over
    [edge feature | sender | receiver | chained global], sum-aggregation at the
    receiver scaled by 1/4; then a node MLP over [x | 4 aggregations | global] and a
    global update from the node and edge sums."""

    def __init__(self, units, activation=None, use_bias=False, combination='concat',
                 learn_global_scale=True, use_global=False, **kwargs):
        self.units = units
        self.activation_fn = activation
        self._act = _activation_code(activation)
        self.use_bias = use_bias
        self.global_scale = 1.
        self.global_scale_nodes = 1.
        self.global_scale_edges = 1.
        self.use_global = use_global
        self.learn_global_scale = learn_global_scale
        if combination == 'additive':
            # the reference points self.call at a method that does not exist (:37-38)
            raise AttributeError("'GeoQuadrantGCN' object has no attribute 'call_additive'")
        if combination != 'concat':
            raise NotImplementedError(f'combination "{combination}" is not implemented.')
        self.max_neighbors = 3
        self.scale = 1. / (self.max_neighbors + 1)
        self.weights = None       # name -> tensor, names as in the reference checkpoint
        self._struct = None
        self._shape = None

    # -- parameters -------------------------------------------------------
    @property
    def has_glob_transform(self):
        return self.use_global and self.learn_global_scale            # :65-70

    def _shapes(self, c_in, f_e, g):
        u = self.units // 4
        shapes = {}
        for q in range(4):
            shapes[f'node_embeddings/{q}'] = (c_in, u)
            shapes[f'edge_transforms/{q}'] = (f_e + 3 * u, u)
            shapes[f'glob_transforms/{q}'] = (g if q == 0 else u, u)
        shapes['node_transform'] = (c_in + self.units + g, self.units)
        if self.has_glob_transform:
            shapes['glob_transform'] = (g + self.units + u, self.units)
        return shapes

    def build(self, c_in, f_e, g, device):
        """Create Glorot-uniform kernels / zero biases (the Keras defaults) if none are loaded."""
        if self.weights is not None:
            return
        w = {}
        for name, shp in self._shapes(c_in, f_e, g).items():
            w[name + '/kernel'] = _glorot(shp, device)
            if self.use_bias:
                w[name + '/bias'] = torch.zeros(shp[1], dtype=torch.float32, device=device)
        self.weights = w
        self._struct = None

    def load_weights(self, flat, prefix, device):
        """Take this layer's variables from a flat {checkpoint name: array} mapping."""
        w = {}
        names = [f'{kind}/{q}' for kind in ('node_embeddings', 'edge_transforms', 'glob_transforms')
                 for q in range(4)] + ['node_transform']
        if self.has_glob_transform:
            names.append('glob_transform')
        for name in names:
            w[name + '/kernel'] = torch.as_tensor(flat[f'{prefix}/{name}/kernel'],
                                                  dtype=torch.float32).to(device).contiguous()
            if self.use_bias:
                w[name + '/bias'] = torch.as_tensor(flat[f'{prefix}/{name}/bias'],
                                                    dtype=torch.float32).to(device).contiguous()
        if self.has_glob_transform:
            self.global_scale_nodes = float(flat.get(f'{prefix}/global_scale_nodes', 1.0))
            self.global_scale_edges = float(flat.get(f'{prefix}/global_scale_edges', 1.0))
        self.weights = w
        self._struct = None

    def named_weights(self, prefix):
        out = {f'{prefix}/{k}': v for k, v in self.weights.items()}
        if self.has_glob_transform:
            out[f'{prefix}/global_scale_nodes'] = self.global_scale_nodes
            out[f'{prefix}/global_scale_edges'] = self.global_scale_edges
        return out

    def _weights_struct(self, c_in, f_e, g):
        if self._struct is not None and self._shape == (c_in, f_e, g):
            return self._struct
        w = self.weights
        for name, shp in self._shapes(c_in, f_e, g).items():
            got = tuple(w[name + '/kernel'].shape)
            if got != shp:
                raise ValueError(f'{name}/kernel has shape {got}, the inputs need {shp}')
        s = GnWeightsStruct()
        s.c_in, s.f_e, s.g, s.units = c_in, f_e, g, self.units
        for q in range(4):
            s.node_emb_w[q] = w[f'node_embeddings/{q}/kernel'].data_ptr()
            s.edge_w[q] = w[f'edge_transforms/{q}/kernel'].data_ptr()
            s.glob_w[q] = w[f'glob_transforms/{q}/kernel'].data_ptr()
            if self.use_bias:
                s.node_emb_b[q] = w[f'node_embeddings/{q}/bias'].data_ptr()
                s.edge_b[q] = w[f'edge_transforms/{q}/bias'].data_ptr()
                s.glob_b[q] = w[f'glob_transforms/{q}/bias'].data_ptr()
        s.node_w = w['node_transform/kernel'].data_ptr()
        if self.use_bias:
            s.node_b = w['node_transform/bias'].data_ptr()
        if self.has_glob_transform:
            s.globt_w = w['glob_transform/kernel'].data_ptr()
            if self.use_bias:
                s.globt_b = w['glob_transform/bias'].data_ptr()
        s.global_scale_nodes = self.global_scale_nodes
        s.global_scale_edges = self.global_scale_edges
        self._struct, self._shape = s, (c_in, f_e, g)
        return s

    # -- forward ----------------------------------------------------------
    def call_concat(self, x, graph, training=False):
        return self._forward(x, graph)

    call = call_concat
    __call__ = call_concat

    def _prepare(self, x, graph):
        glob = graph['global']                      # KeyError when absent, like the reference (:83)
        if not self.use_global:
            raise NotImplementedError('GeoQuadrantGCN with use_global=False is not implemented; the '
                                      "reference's GraphUNet path always sets use_global=True.")
        x = _as_f32(x, 'x')
        single = x.dim() == 2
        xb = x.unsqueeze(0) if single else x
        if xb.dim() != 3:
            raise ValueError(f'x must be [N, C] or [B, N, C], got {tuple(x.shape)}')
        glob = _as_f32(glob, "graph['global']").reshape(-1, glob.shape[-1])
        if glob.shape[0] != xb.shape[0]:
            raise ValueError(f"graph['global'] has {glob.shape[0]} rows for a batch of {xb.shape[0]}")
        return xb, glob, single

    def _outputs(self, batch, n_rows, g, device):
        u = self.units // 4
        x_out = torch.empty((batch, n_rows, self.units), dtype=torch.float32, device=device)
        g_out = self.units if self.has_glob_transform else g + self.units + u
        global_out = torch.empty((batch, g_out), dtype=torch.float32, device=device)
        return x_out, global_out

    def _scratch(self, qg, batch, device):
        nbytes = qg.workspace_bytes(batch, self.units)
        if nbytes == 0:
            raise NotImplementedError(f'GeoQuadrantGCN with units={self.units} is not implemented.')
        return _plan.workspace(nbytes, device)

    def _forward(self, x, graph, x_skip=None):
        """x_skip: optional trailing input columns, so that a caller need not materialise
        concat(x, x_skip) (the up path, reference :235)."""
        edge_index = graph['edge_index']
        edges = graph['edges']
        xb, glob, single = self._prepare(x, graph)
        batch, n, c_a = xb.shape
        c_skip = 0
        if x_skip is not None:
            x_skip = _as_f32(x_skip, 'x_skip')
            x_skip = x_skip.unsqueeze(0) if x_skip.dim() == 2 else x_skip
            if x_skip.shape[:2] != xb.shape[:2]:
                raise ValueError(f'x_skip has shape {tuple(x_skip.shape)}, expected [{batch}, {n}, C]')
            c_skip = x_skip.shape[2]
        c_in = c_a + c_skip
        g = glob.shape[1]
        if len(edge_index) != 4 or len(edges) != 4:
            raise ValueError('GeoQuadrantGCN needs 4 edge lists and 4 edge-feature tensors')
        qg = _plan.get_quadgraph(edge_index, n)
        feats = []
        f_e = None
        shared = all(e.dim() == 2 for e in edges) and batch > 1
        for q in range(4):
            e = _as_f32(edges[q], f"graph['edges'][{q}]")
            eb = e.unsqueeze(0) if e.dim() == 2 else e
            if eb.shape[0] != (1 if shared else batch) or eb.shape[1] != qg.n_edges[q]:
                raise ValueError(f"graph['edges'][{q}] has shape {tuple(e.shape)}, expected "
                                 f'[{batch}, {qg.n_edges[q]}, F] (or [{qg.n_edges[q]}, F] shared by the batch)')
            f_e = eb.shape[2] if f_e is None else f_e
            if eb.shape[2] != f_e:
                raise ValueError('all four edge-feature tensors must have the same width')
            feats.append(eb)
        self.build(c_in, f_e, g, x.device)
        ws = self._weights_struct(c_in, f_e, g)
        u = self.units // 4
        dev = x.device
        x_out, global_out = self._outputs(batch, n, g, dev)
        edges_out = [torch.empty((batch, qg.n_edges[q], u), dtype=torch.float32, device=dev)
                     for q in range(4)]
        scratch = self._scratch(qg, batch, dev)
        in_ptrs = ptr4(feats)
        out_ptrs = ptr4(edges_out)
        call('gunet_gn_layer_forward_f32', C.byref(qg.struct), C.byref(ws), batch, ptr(xb), ptr(x_skip),
             c_skip, C.byref(in_ptrs), 1 if shared else 0, ptr(glob), self._act, ptr(x_out), C.byref(out_ptrs),
             ptr(global_out), ptr(scratch), scratch.numel(), current_stream(dev))
        if single:
            x_out = x_out[0]
            edges_out = [e[0] for e in edges_out]
        graph['edges'] = edges_out                                          # :128
        graph['global'] = global_out                                        # :135
        return x_out, graph

    def _upsample_forward(self, x, glob, up):
        """The layer applied to [x ; zeros(n_fine)] on the upsampling graph `up` with zero edge
        features, returning only the fine rows and the new global (reference :342-377)."""
        xb, glob, single = self._prepare(x, {'global': glob})
        batch, n_coarse, c_in = xb.shape
        g = glob.shape[1]
        self.build(c_in, 2, g, x.device)
        ws = self._weights_struct(c_in, 2, g)
        qg = up.quadgraph
        x_out, global_out = self._outputs(batch, up.n_fine, g, x.device)
        scratch = self._scratch(qg, batch, x.device)
        call('gunet_gn_upsample_forward_f32', C.byref(qg.struct), n_coarse, C.byref(ws), batch, ptr(xb),
             ptr(glob), self._act, ptr(x_out), ptr(global_out), ptr(scratch), scratch.numel(),
             current_stream(x.device))
        return (x_out[0] if single else x_out), global_out


def _layer_class(layer_type, default):
    from . import get
    return get(layer_type, default)


class GraphDownsamplingBlock:
    """n_layers GN layers followed by 2x2 graph pooling (reference :140-191)."""

    def __init__(self, units, activation=None, use_bias=False, n_layers=1, diffuse_graph=True,
                 residual=False, layer_type=None, use_global=False):
        self.units = units
        self.activation = activation
        self.use_bias = use_bias
        self.diffuse_graph = diffuse_graph
        self.residual = residual
        self.n_layers = n_layers
        self.layer_type = layer_type
        self.use_global = use_global
        self.gcns = None
        self.build(None)

    def build(self, input_shape):
        if self.gcns is None:
            GraphLayer = _layer_class(self.layer_type, GeoQuadrantGCN)
            self.gcns = [GraphLayer(self.units, activation=self.activation, use_bias=self.use_bias,
                                    use_global=self.use_global) for _ in range(self.n_layers)]

    def call(self, x, graph):
        if self.diffuse_graph and self.n_layers > 1:
            raise NotImplementedError('diffuse_graph=True is not implemented (GraphUNet passes False).')
        res = x
        for gcn in self.gcns:
            x, graph = gcn(x, graph)
        if self.residual:
            x = torch.cat((x, res), dim=-1)
        return graph_pooling_2x2(x, graph)

    __call__ = call


class GraphUpsamplingBlock:
    """Graph upsampling, skip concatenation, n_layers GN layers (reference :193-252)."""

    def __init__(self, units, activation=None, use_bias=False, n_layers=2, diffuse_graph=True,
                 residual=False, layer_type=None, last_block=False, use_global=False):
        self.units = units
        self.activation = activation
        self.use_bias = use_bias
        self.diffuse_graph = diffuse_graph
        self.residual = residual
        self.n_layers = n_layers
        self.last_block = last_block
        self.layer_type = layer_type
        self.use_global = use_global
        self.gcns = None
        self.build(None)

    def build(self, input_shape):
        if self.gcns is not None:
            return
        self.upsampling = GraphUpsampling2x2(self.units, method='graph', layer_type=self.layer_type,
                                             use_bias=self.use_bias, activation=self.activation,
                                             use_global=self.use_global)
        GraphLayer = _layer_class(self.layer_type, GeoQuadrantGCN)
        self.gcns = [GraphLayer(self.units, activation=self.activation, use_bias=self.use_bias,
                                learn_global_scale=not (self.last_block and l == self.n_layers - 1),
                                use_global=self.use_global) for l in range(self.n_layers)]

    def call(self, x, graph, x_skip, target_graph):
        if self.diffuse_graph and self.n_layers > 1:
            raise NotImplementedError('diffuse_graph=True is not implemented (GraphUNet passes False).')
        x, graph = self.upsampling(x, graph, target_graph)
        x_res = x
        if isinstance(self.gcns[0], GeoQuadrantGCN):
            x, graph = self.gcns[0]._forward(x, graph, x_skip=x_skip)       # concat(x, x_skip) :235-237
        else:
            x, graph = self.gcns[0](torch.cat((x, x_skip), dim=-1), graph)
        for gcn in self.gcns[1:]:
            x, graph = gcn(x, graph)
        if self.residual:
            x = x + x_res
        return x, graph

    __call__ = call


class GraphUpsampling2x2:
    """Unparameterised-topology upsampling (reference :254-389).  Only method 'graph', the one
    GraphUNet uses (:209), is implemented: every fine node gets one edge from its coarse parent,
    the GN layer runs on [coarse nodes; zero-initialised fine nodes] and the fine rows are kept."""

    def __init__(self, units=None, method='constant', col_idx=0, layer_type=None, use_global=False,
                 use_bias=False, activation=None):
        self.col_idx = col_idx
        self.method = method
        self.units = units
        self.use_bias = use_bias
        if self.method == 'graph':
            GraphLayer = _layer_class(layer_type, GeoQuadrantGCN)
            self.gcn = GraphLayer(self.units, activation=activation, use_bias=use_bias,
                                  use_global=use_global)

    def call(self, x, graph, target_graph):
        if self.method != 'graph' or self.col_idx != 0:
            raise NotImplementedError(f'Upsampling method "{self.method}" is not implemented.')
        x = _as_f32(x, 'x')
        up = _plan.get_upsample_plan(graph['node_loc'], target_graph['node_loc'])
        n_in = x.shape[-2]
        if n_in != up.n_coarse:
            raise ValueError(f'x has {n_in} rows but the coarse graph has {up.n_coarse} nodes')
        glob = torch.cat((graph['global'].reshape(-1, graph['global'].shape[-1]),
                          target_graph['global'].reshape(-1, target_graph['global'].shape[-1])),
                         dim=-1)                                            # :375
        if (isinstance(self.gcn, GeoQuadrantGCN) and self.gcn.units <= 256 and self.gcn._act in _TUNED
                and sum(int(e.shape[0]) for e in up.edge_index) == up.n_fine):
            # same result as the layer on [x ; zeros] with zero edge features (:342-377), without
            # computing the zero rows
            x, graph['global'] = self.gcn._upsample_forward(x, glob, up)
        else:
            zeros = torch.zeros(x.shape[:-2] + (up.n_fine, x.shape[-1]), dtype=x.dtype, device=x.device)
            x = torch.cat((x, zeros), dim=-2)                               # :342-344
            lead = x.shape[:-2]
            graph['edge_index'] = up.edge_index                             # :371
            graph['edges'] = [torch.zeros(lead + tuple(e.shape), dtype=torch.float32, device=x.device)
                              for e in up.edge_index]                       # :373-374
            graph['global'] = glob
            x, graph = self.gcn(x, graph)                                   # :376
            x = x[..., n_in:, :]                                            # :377
        graph['edge_index'] = target_graph['edge_index']                    # :382-384
        graph['edges'] = target_graph['edges']
        graph['node_loc'] = target_graph['node_loc']
        return x, graph

    __call__ = call


def _segment_max(src, rowptr, perm, n_segments):
    """src [n, w] or [B, n, w] -> [n_segments, w] / [B, n_segments, w]."""
    src = _as_f32(src, 'features')
    single = src.dim() == 2
    sb = src.unsqueeze(0) if single else src
    batch, n_src, width = sb.shape
    out = torch.empty((batch, n_segments, width), dtype=torch.float32, device=src.device)
    call('gunet_segment_max_f32', ptr(sb), n_src, width, ptr(rowptr), ptr(perm), n_segments, batch,
         ptr(out), current_stream(src.device))
    return out[0] if single else out


def remove_duplicate_edges(edge_index, n_nodes, edges, remove_selfloops=True):
    """Reference :391-408 for an already relabelled edge list: first-occurrence unique on
    sender * n_nodes + receiver, max of the features over duplicates, then self-loop removal."""
    if not remove_selfloops:
        raise NotImplementedError('remove_selfloops=False is not implemented.')
    dev = edge_index.device
    e = edge_index.to(torch.int32).contiguous()
    ne = int(e.shape[0])
    identity = torch.arange(int(n_nodes), dtype=torch.int32, device=dev)
    new_edges = torch.empty((max(ne, 1), 2), dtype=torch.int32, device=dev)
    dup_map = torch.empty(max(ne, 1), dtype=torch.int32, device=dev)[:ne]
    n_new = C.c_int64(0)
    if ne > 0:
        call('gunet_pool_edges', ptr(e), ne, ptr(identity), int(n_nodes), ptr(new_edges), ptr(dup_map),
             C.byref(n_new), current_stream(dev))
    n_new = int(n_new.value)
    rowptr, perm = _plan.csr_from_keys(dup_map, n_new)
    return new_edges[:n_new].to(edge_index.dtype), _segment_max(edges, rowptr, perm, n_new)


def graph_pooling_2x2(x, graph, col_idx=0):
    """2x2 max pooling of node features, node positions, edge lists and edge features
    (reference :410-447)."""
    if col_idx != 0:
        raise NotImplementedError('graph_pooling_2x2 is implemented for col_idx=0 only.')
    pp = _plan.get_pool_plan(graph['node_loc'], graph['edge_index'])
    if x.shape[-2] != pp.n_nodes:
        raise ValueError(f"x has {x.shape[-2]} rows but graph['node_loc'] has {pp.n_nodes}")
    x_pooled = _segment_max(x, pp.cluster_rowptr, pp.cluster_perm, pp.n_pooled)        # :427
    new_feats = [_segment_max(graph['edges'][q], pp.dup_rowptr[q], pp.dup_perm[q],
                              pp.n_pooled_edges[q]) for q in range(4)]                # :401
    graph['edge_index'] = pp.edge_index
    graph['edges'] = new_feats
    graph['node_loc'] = pp.pooled_loc
    return x_pooled, graph
