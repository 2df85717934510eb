"""CPU restatement (numpy) of the Graph U-Net forward hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke
check and the benchmark's cpu_baseline leg may import it.  The product path
(the HIP library) never calls into it.

PARITY UNPINNED: the reference is TensorFlow 2.6 code whose arithmetic lives in
tensorflow-gpu==2.6.0 (requirements.txt:8), which is not installable here, and
the reference ships no tests or golden vectors for this path.  The pins that do
exist are (1) the variable names/shapes of the shipped checkpoint index
(tests/golden/ckpt_shapes.json), (2) hand-derived micro known-answers
(tests/test_oracle.py) and (3) structural invariants of grid graphs.

Every function follows the reference operation order (unfused: four separate
dense layers, explicit gathers, concat, tile, segment ops) and cites the lines
it restates (paths relative to the reference checkout).

TensorFlow op semantics restated here:
  tf.unique             -> y in first-occurrence order, idx int32
  unsorted_segment_sum  -> sum per segment id (rows added in input order)
  unsorted_segment_max  -> max per segment id (empty segment: dtype lowest)
  unsorted_segment_min  -> min per segment id (empty segment: dtype highest)
  boolean_mask / dynamic_partition / where -> stable, original order kept
  keras Dense           -> x @ kernel + bias, kernel stored [in, out]
"""
import numpy as np

F32 = np.float32

# data/data_utils.py:10-20 -- (dx, dy) per direction N,NE,E,SE,S,SW,W,NW,self
DIR_TO_EDGE = np.array(
    [[0, -1], [1, -1], [1, 0], [1, 1], [0, 1], [-1, 1], [-1, 0], [-1, -1], [0, 0]],
    dtype=np.int64)

# data/data_utils.py:24-30 -- directions belonging to quadrant NE, SE, SW, NW, self
QUADRANT_IDS = np.array(
    [[0, 1, 2], [2, 3, 4], [4, 5, 6], [6, 7, 0], [8, 8, 8]], dtype=np.int64)


# --------------------------------------------------------------------------
# TensorFlow primitive restatements
# --------------------------------------------------------------------------

def tf_unique(x):
    """tf.unique on a 1-D array: (y, idx); y in first-occurrence order, idx int32."""
    x = np.asarray(x)
    if x.size == 0:
        return x.copy(), np.zeros((0,), np.int32)
    y_sorted, first, inv = np.unique(x, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')          # sorted-unique slot -> appearance rank
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    return y_sorted[order], rank[inv.reshape(-1)].astype(np.int32)


def _segments(ids, num_segments):
    ids = np.asarray(ids).astype(np.int64)
    order = np.argsort(ids, kind='stable')
    counts = np.bincount(ids, minlength=num_segments)
    starts = np.concatenate(([0], np.cumsum(counts)[:-1]))
    return order, counts, starts


def _segment_reduce(ufunc, data, ids, num_segments, empty):
    data = np.asarray(data)
    out = np.full((num_segments,) + data.shape[1:], empty, dtype=data.dtype)
    if data.shape[0] == 0 or num_segments == 0:
        return out
    order, counts, starts = _segments(ids, num_segments)
    nz = counts > 0
    # reduceat walks each slice front to back, i.e. rows are combined in input order
    red = ufunc.reduceat(data[order], starts[nz].astype(np.int64), axis=0)
    out[nz] = red
    return out


def unsorted_segment_sum(data, ids, num_segments):
    """Rows are added in input order, one rank-within-segment at a time (each pass touches every
    segment at most once, so the vectorised add is exactly the sequential sum)."""
    data = np.asarray(data)
    out = np.zeros((num_segments,) + data.shape[1:], dtype=data.dtype)
    if data.shape[0] == 0 or num_segments == 0:
        return out
    order, counts, starts = _segments(ids, num_segments)
    for r in range(int(counts.max())):
        seg = np.nonzero(counts > r)[0]
        out[seg] = out[seg] + data[order[starts[seg] + r]]
    return out


def unsorted_segment_max(data, ids, num_segments):
    d = np.asarray(data)
    lowest = np.finfo(d.dtype).min if d.dtype.kind == 'f' else np.iinfo(d.dtype).min
    return _segment_reduce(np.maximum, d, ids, num_segments, lowest)


def unsorted_segment_min(data, ids, num_segments):
    d = np.asarray(data)
    highest = np.finfo(d.dtype).max if d.dtype.kind == 'f' else np.iinfo(d.dtype).max
    return _segment_reduce(np.minimum, d, ids, num_segments, highest)


def dense(x, kernel, bias=None):
    y = np.matmul(x, kernel, dtype=x.dtype)
    if bias is not None:
        y = y + bias
    return y


def relu(x):
    return np.maximum(x, F32(0))


def identity(x):
    return x


def _erf(x):
    from scipy.special import erf
    return erf(x)


def _f32_like(x, y):
    return y.astype(x.dtype, copy=False)


def sigmoid(x):
    return _f32_like(x, 1 / (1 + np.exp(-x)))


def softplus(x):                      # tf.keras.activations.softplus: log(exp(x) + 1)
    return _f32_like(x, np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x))))


def elu(x):                           # alpha = 1
    return _f32_like(x, np.where(x > 0, x, np.expm1(np.minimum(x, 0))))


def selu(x):                          # tf.keras.activations.selu: scale * elu(x, alpha)
    return _f32_like(x, 1.0507009873554805 * np.where(x > 0, x, 1.6732632423543772 * np.expm1(np.minimum(x, 0))))


# layers/hybrid_gnn.py:25 and models/graph_unet.py:21 pass the name to tf.keras.layers.Dense
ACTIVATIONS = {'relu': relu, None: identity, 'linear': identity, 'tanh': np.tanh, 'sigmoid': sigmoid,
               'elu': elu, 'softplus': softplus, 'selu': selu, 'swish': lambda x: _f32_like(x, x * sigmoid(x)),
               'gelu': lambda x: _f32_like(x, 0.5 * x * (1 + _erf(x / np.sqrt(2.0)))),       # approximate=False
               'softsign': lambda x: _f32_like(x, x / (1 + np.abs(x))),
               'exponential': lambda x: _f32_like(x, np.exp(x)),
               'hard_sigmoid': lambda x: _f32_like(x, np.clip(0.2 * x + 0.5, 0, 1))}


def column_sum(x):
    """tf.reduce_sum(x, axis=0, keepdims=True).  The exact sum rounded once to
    fp32 (accumulated in fp64): every fp32 summation order lies within
    O(sqrt(rows)*eps) of it, which for 1e5 rows is the same size as the parity
    tolerance, so the oracle takes the order-free value."""
    return np.sum(x, axis=0, keepdims=True, dtype=np.float64).astype(x.dtype)


# --------------------------------------------------------------------------
# graph synthesis  (data/data_utils.py:84-150)
# --------------------------------------------------------------------------

def mask_invalid_ids(graph_data):
    """data/data_utils.py:84-93 -- zero the directional maps that leave the image."""
    g = graph_data
    g[[0, 1, 7], 0] = 0        # N, NE, NW in the first row
    g[[1, 2, 3], :, -1] = 0    # NE, E, SE in the last column
    g[[3, 4, 5], -1] = 0       # SE, S, SW in the last row
    g[[5, 6, 7], :, 0] = 0     # SW, W, NW in the first column
    return g


def image_to_graph(graph_data, add_selfloops=True):
    """data/data_utils.py:95-150.  graph_data [8,H,W] binary directional maps.
    Returns node_loc (N,2) int64 [row,col], edge_index (E,2) int32 [src,dst],
    direction (E,) int64, node_mask (H,W) bool.  Edges are ordered by
    (source pixel, direction); self-loops carry direction 8."""
    graph_data = np.array(graph_data, dtype=np.float64)
    h, w = graph_data.shape[1:]
    graph_data = mask_invalid_ids(graph_data)
    graph_data = graph_data / np.max(graph_data)
    flat = graph_data.reshape(8, -1)
    node_map = np.max(flat, axis=0).astype(np.int32)
    node_mask = node_map == 1
    if add_selfloops:
        flat = np.concatenate((flat, node_map[None].astype(flat.dtype)), axis=0)
    pixels = np.nonzero(node_mask)[0]
    node_ids = np.zeros(h * w, dtype=np.int32)
    node_ids[pixels] = np.arange(pixels.size, dtype=np.int32)
    src_pixel, direction = np.nonzero(flat.T)
    offset = DIR_TO_EDGE[direction, 0] + DIR_TO_EDGE[direction, 1] * w
    src = node_ids[src_pixel]
    dst = node_ids[src_pixel + offset]   # a non-node target silently maps to node 0
    node_loc = np.stack((pixels // w, pixels % w), axis=-1).astype(np.int64)
    edge_index = np.stack((src, dst), axis=-1).astype(np.int32)
    return node_loc, edge_index, direction.astype(np.int64), node_mask.reshape(h, w)


def directional_maps_from_mask(mask):
    """Synthetic road map: a directed edge p -> p+delta for each of the 8
    neighbour offsets whenever both pixels are set in `mask` (symmetric, so every
    edge target is itself a node)."""
    mask = np.asarray(mask, dtype=bool)
    h, w = mask.shape
    maps = np.zeros((8, h, w), dtype=np.float64)
    padded = np.zeros((h + 2, w + 2), dtype=bool)
    padded[1:-1, 1:-1] = mask
    for d in range(8):
        dx, dy = DIR_TO_EDGE[d]
        neigh = padded[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
        maps[d] = mask & neigh
    return maps


def synth_road_graph(height, width, density, seed):
    """A Traffic4Cast-shaped road graph: mask = rng.random((H,W)) < density."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mask = rng.random((height, width)) < density
    return image_to_graph(directional_maps_from_mask(mask))


def dense_grid_graph(height, width):
    return image_to_graph(directional_maps_from_mask(np.ones((height, width), bool)))


# --------------------------------------------------------------------------
# a1: quadrant partition  (data/data_utils.py:47-58)
# --------------------------------------------------------------------------

def sample_quadrant_subgraphs(edge_index, direction):
    """Stable 5-way partition (5th = self-loops, dropped).  Cardinal directions
    0,2,4,6 belong to two quadrants; original relative order is kept."""
    edge_index = np.asarray(edge_index)
    direction = np.asarray(direction).astype(np.int32)
    rep_ids = np.array([0, 2, 4, 6], dtype=np.int32)
    repeats = (1 - np.minimum(1, np.min(np.abs(direction[:, None] - rep_ids), axis=-1))) + 1
    rep_edges = np.repeat(edge_index, repeats, axis=0)
    member = np.min(np.abs(direction[:, None, None] - QUADRANT_IDS[None]), axis=-1) == 0
    subgraph_ids = np.argwhere(member)[:, 1].astype(np.int32)   # row-major = (edge, quadrant)
    return tuple(rep_edges[subgraph_ids == q] for q in range(4))


# --------------------------------------------------------------------------
# a2: gathers  (layers/graph_ops.py:5-17)
# --------------------------------------------------------------------------

def get_sender_receiver(node_features, edge_index):
    return node_features[edge_index[:, 0]], node_features[edge_index[:, 1]]


# --------------------------------------------------------------------------
# a3/a4: GeoQuadrantGCN  (layers/hybrid_gnn.py:20-137)
# --------------------------------------------------------------------------

def _kb(W, name, use_bias=True):
    k = W[name + '/kernel']
    b = W.get(name + '/bias') if use_bias else None
    return k, b


def geo_quadrant_gcn(x, graph, W, prefix, activation='relu', use_bias=True, keep=None):
    """layers/hybrid_gnn.py:72-137 (call_concat, use_global=True).  Mutates and
    returns `graph` exactly as the reference does.  `keep`, if a dict, receives
    the intermediate tensors."""
    act = ACTIVATIONS[activation]
    n_nodes = x.shape[0]
    edge_index = graph['edge_index']
    edges = graph['edges']
    edge_global = F32(0)
    quad_neighbors, quad_edges = [], []
    glob = graph['global']
    scale = F32(1.0 / (3 + 1))                                       # :60-61
    for q in range(4):
        x_quad = act(dense(x, *_kb(W, f'{prefix}/node_embeddings/{q}', use_bias)))   # :91-92
        sender, receiver = get_sender_receiver(x_quad, edge_index[q])               # :93
        edge = np.concatenate((edges[q], sender, receiver), axis=-1)                # :95
        glob = act(dense(glob, *_kb(W, f'{prefix}/glob_transforms/{q}', use_bias)))  # :98-99 (chained)
        edge = np.concatenate((edge, np.tile(glob, (edge.shape[0], 1))), axis=-1)   # :100-101
        edge = act(dense(edge, *_kb(W, f'{prefix}/edge_transforms/{q}', use_bias)))  # :103-104
        edge_global = edge_global + column_sum(edge)                                # :108
        agg = unsorted_segment_sum(edge, edge_index[q][:, 1], n_nodes)              # :110-111
        agg = agg * scale                                                           # :113
        quad_neighbors.append(agg)
        quad_edges.append(edge)
        if keep is not None:
            keep[f'x_quad{q}'] = x_quad
            keep[f'glob{q}'] = glob
    xin = np.concatenate([x] + quad_neighbors, axis=-1)                             # :118-119
    xin = np.concatenate((xin, np.tile(graph['global'], (n_nodes, 1))), axis=-1)    # :122-123
    x_out = act(dense(xin, *_kb(W, f'{prefix}/node_transform', use_bias)))          # :125-126
    graph['edges'] = quad_edges                                                     # :128
    has_gt = (prefix + '/glob_transform/kernel') in W
    gse = F32(W.get(prefix + '/global_scale_edges', 1.0))
    gsn = F32(W.get(prefix + '/global_scale_nodes', 1.0))
    edge_global = (edge_global * F32(1.0)) / F32(gse * F32(300.) * F32(300.) * F32(8.))   # :131
    node_global = column_sum(x_out)                                                       # :132
    node_global = (node_global * F32(1.0)) / F32(gsn * F32(300.) * F32(300.))             # :133
    global_in = np.concatenate((graph['global'], node_global, edge_global), axis=-1)      # :134
    if has_gt:
        graph['global'] = act(dense(global_in, *_kb(W, f'{prefix}/glob_transform', use_bias)))  # :66,135
    else:
        graph['global'] = global_in                                                       # :70
    if keep is not None:
        keep['agg'] = np.concatenate(quad_neighbors, axis=-1)
        keep['node_global'] = node_global
        keep['edge_global'] = edge_global
    return x_out, graph


# --------------------------------------------------------------------------
# a5/a6: pooling  (layers/hybrid_gnn.py:391-447)
# --------------------------------------------------------------------------

def cluster_ids_2x2(node_loc, col_idx=0):
    """layers/hybrid_gnn.py:417-425 (and :270-280).  Returns (ids int32, P).
    The multiplier is max(node_loc[:, col_idx]) + 1 with col_idx = 0, i.e. the
    largest ROW index + 1, applied to the half row; the half column is added."""
    node_loc = np.asarray(node_loc, dtype=np.int64)
    m = np.max(node_loc[:, col_idx]) + 1
    padding = np.array([m, 1], dtype=np.int64) if col_idx == 0 else np.array([1, m], dtype=np.int64)
    pos_id = np.sum(node_loc // 2 * padding, axis=-1)
    y, idx = tf_unique(pos_id)
    return idx, int(y.shape[0])


def remove_duplicate_edges(edge_index, n_nodes, edges, remove_selfloops=True):
    """layers/hybrid_gnn.py:391-408.  n_nodes is the PRE-pooling node count."""
    dtype = edge_index.dtype
    e64 = edge_index.astype(np.int64)
    lin = e64[:, 0] * np.int64(n_nodes) + e64[:, 1]
    y, idx = tf_unique(lin)
    new_edges = np.stack((y // n_nodes, y % n_nodes), axis=-1)
    new_feat = unsorted_segment_max(edges, idx, new_edges.shape[0])
    if remove_selfloops:
        keep = new_edges[:, 0] != new_edges[:, 1]
        new_edges = new_edges[keep]
        new_feat = new_feat[keep]
    return new_edges.astype(dtype), new_feat


def graph_pooling_2x2(x, graph, col_idx=0):
    """layers/hybrid_gnn.py:410-447."""
    node_loc = np.asarray(graph['node_loc'], dtype=np.int64)
    n_nodes = np.int64(node_loc.shape[0])
    new_ids, n_pooled = cluster_ids_2x2(node_loc, col_idx)
    x_pooled = unsorted_segment_max(x, new_ids, n_pooled)                     # :427
    new_pos = unsorted_segment_min(node_loc // 2, new_ids, n_pooled)          # :428-429
    new_index, new_feats = [], []
    for e_index, e_feat in zip(graph['edge_index'], graph['edges']):
        relabelled = new_ids[e_index]                                         # :432
        ei, ef = remove_duplicate_edges(relabelled, n_nodes, e_feat)          # :433
        new_index.append(ei)
        new_feats.append(ef)
    graph['edge_index'] = new_index
    graph['edges'] = new_feats
    graph['node_loc'] = new_pos
    return x_pooled, graph


# --------------------------------------------------------------------------
# a7: graph upsampling, method 'graph'  (layers/hybrid_gnn.py:267-280, 331-384)
# --------------------------------------------------------------------------

def upsample_edges(coarse_loc, fine_loc, col_idx=0):
    """Index part of the 'graph' upsampling: parent ids and the 4 edge lists
    (parent -> N_in + fine index) split by the child's position in its 2x2 cell."""
    n_in = coarse_loc.shape[0]
    n_t = fine_loc.shape[0]
    parent, _ = cluster_ids_2x2(fine_loc, col_idx)
    receivers = np.arange(n_in, n_in + n_t, dtype=np.int32)
    new_edges = np.stack((parent, receivers), axis=-1)                         # :338-340
    old_locs = coarse_loc.astype(F32) * F32(2) + F32(0.5)                      # :349
    pos = np.concatenate((old_locs, fine_loc.astype(F32)), axis=0)             # :350
    rel = pos[new_edges]                                                       # :351
    rel = rel[:, 1] - rel[:, 0]                                                # :352
    cond = np.array([[-.5, .5], [.5, .5], [.5, -.5], [-.5, -.5]], dtype=F32)   # :353
    lists = [new_edges[np.all(rel == cond[i], axis=-1)] for i in range(4)]     # :355-357
    return parent, lists


def graph_upsampling_graph(x, graph, target_graph, W, prefix, activation='relu', use_bias=True):
    """GraphUpsampling2x2.call with method='graph' (layers/hybrid_gnn.py:331-384)."""
    n_in = x.shape[0]
    fine_loc = np.asarray(target_graph['node_loc'], dtype=np.int64)
    n_t = fine_loc.shape[0]
    _, lists = upsample_edges(np.asarray(graph['node_loc'], dtype=np.int64), fine_loc)
    x = np.concatenate((x, np.zeros((n_t, x.shape[1]), dtype=x.dtype)), axis=0)   # :342-344
    graph['edge_index'] = lists                                                   # :371
    graph['edges'] = [np.zeros(e.shape, dtype=F32) for e in lists]                # :373-374
    graph['global'] = np.concatenate((graph['global'], target_graph['global']), axis=-1)  # :375
    x, graph = geo_quadrant_gcn(x, graph, W, prefix + '/gcn', activation, use_bias)        # :376
    x = x[n_in:]                                                                  # :377
    graph['edge_index'] = target_graph['edge_index']                              # :382-384
    graph['edges'] = target_graph['edges']
    graph['node_loc'] = target_graph['node_loc']
    return x, graph


# --------------------------------------------------------------------------
# a8/a9/a10: blocks and the model
# --------------------------------------------------------------------------

def graph_downsampling_block(x, graph, W, prefix, activation='relu', use_bias=True, n_layers=1):
    """layers/hybrid_gnn.py:160-191 (diffuse_graph=False, residual=False)."""
    for l in range(n_layers):
        x, graph = geo_quadrant_gcn(x, graph, W, f'{prefix}/gcns/{l}', activation, use_bias)
    return graph_pooling_2x2(x, graph)


def graph_upsampling_block(x, graph, x_skip, target_graph, W, prefix, activation='relu',
                           use_bias=True, n_layers=2):
    """layers/hybrid_gnn.py:221-252 (diffuse_graph=False, residual=False)."""
    x, graph = graph_upsampling_graph(x, graph, target_graph, W, prefix + '/upsampling',
                                      activation, use_bias)
    x = np.concatenate((x, x_skip), axis=-1)                                   # :235
    for l in range(n_layers):
        x, graph = geo_quadrant_gcn(x, graph, W, f'{prefix}/gcns/{l}', activation, use_bias)
    return x, graph


def graph_unet_forward(x, graph, W, depth=3, activation='relu', use_bias=True, keep=None):
    """models/graph_unet.py:60-84."""
    skip_graphs, skips = [], []
    x = dense(x, W['in_layer/kernel'], W['in_layer/bias'])                     # :66 (always biased, no activation)
    for d in range(depth):
        skip_graphs.append(dict(graph))                                        # :69 shallow copy
        skips.append(x)
        x, graph = graph_downsampling_block(x, graph, W, f'downsampling_layers/{d}',
                                            activation, use_bias)
        if keep is not None:
            keep[f'down{d}_x'] = x
            keep[f'down{d}_global'] = graph['global']
    x, graph = geo_quadrant_gcn(x, graph, W, 'bottleneck_gcn_1', activation, use_bias)
    x, graph = geo_quadrant_gcn(x, graph, W, 'bottleneck_gcn_2', activation, use_bias)
    if keep is not None:
        keep['bottleneck_x'] = x
    for d in range(depth):
        x, graph = graph_upsampling_block(x, graph, skips[-(d + 1)], skip_graphs[-(d + 1)], W,
                                          f'upsampling_layers/{d}', activation, use_bias)
        if keep is not None:
            keep[f'up{d}_x'] = x
    x = dense(x, W['out_model/kernel'])                                        # :82 (no bias)
    return x, graph


# --------------------------------------------------------------------------
# caller-side input assembly  (models/base_model.py:282-325)
# --------------------------------------------------------------------------

def assemble_inputs(seed_nodes, street_encoding, edge_index, node_loc, time_enc, weekday):
    """GraphBaseModel._call minus the street-map CNN: seed_nodes (N,96),
    street_encoding (N,8), time_enc (1,T,2), weekday (1,) ->
    x (N,104), graph {'edge_index','edges' 4x(E_q,16),'node_loc','global' (1,113)}."""
    edges = []
    for es in edge_index:
        enc = street_encoding[es]                                              # :283
        edges.append(np.concatenate((enc[:, 0], enc[:, 1]), axis=-1))          # :284
    traffic_input = np.concatenate((seed_nodes, street_encoding), axis=-1)     # :300
    glob = column_sum(traffic_input) / F32(300. * 300.)                        # :308
    t_enc = time_enc[:, -1]                                                    # :309
    weekday_enc = np.eye(7, dtype=F32)[np.asarray(weekday).reshape(-1)]        # :310
    glob = np.concatenate((glob, t_enc.astype(F32), weekday_enc), axis=-1)     # :311
    graph = {'edge_index': list(edge_index), 'edges': edges, 'node_loc': node_loc,
             'global': glob.astype(F32)}
    return traffic_input.astype(F32), graph


def conv2d_same(x, kernel):
    """tf.keras.layers.Conv2D(kernel_size=3, strides=1, padding='SAME', use_bias=False) on one image
    x [H, W, C_in] with kernel [3, 3, C_in, C_out]: cross-correlation over a zero-padded input."""
    h, w, _ = x.shape
    xp = np.pad(x.astype(F32), ((1, 1), (1, 1), (0, 0)))
    out = np.zeros((h, w, kernel.shape[3]), dtype=F32)
    for ky in range(3):
        for kx in range(3):
            out += xp[ky:ky + h, kx:kx + w, :] @ kernel[ky, kx].astype(F32)
    return out


def forward_street_encoding(street_map, W, node_loc):
    """models/base_model.py:274-280: uint8 street map [H, W] -> /255 -> cnn1 (3x3, 32, relu) ->
    cnn2 (3x3, 8, relu) -> gather_nd at node_loc (row, col) -> [N, 8]."""
    img = street_map.astype(F32)[..., None] / F32(255)                         # :276-277
    enc = relu(conv2d_same(relu(conv2d_same(img, W['cnn1/kernel'])), W['cnn2/kernel']))   # :263-264,278
    loc = np.asarray(node_loc, dtype=np.int64)
    return enc[loc[:, 0], loc[:, 1]]                                           # :279


def compute_edge_features(street_encoding, edge_index):
    """models/base_model.py:282-285."""
    enc = street_encoding[edge_index]
    return np.concatenate((enc[:, 0], enc[:, 1]), axis=-1)


def model_call(x, W, **kw):
    """GraphBaseModel._call (models/base_model.py:287-325), inference: x is the reference's nested
    input {'graph': {'seed_nodes', 'edge_index' (4 lists), 'node_loc'}, 'image': {'street_map'},
    'time_enc' [1, T, 2], 'weekday' [1]} -> max(pred, 0) [N, out_units]."""
    g = x['graph']
    enc = forward_street_encoding(x['image']['street_map'], W, g['node_loc'])  # :293
    x_in, graph = assemble_inputs(g['seed_nodes'], enc, g['edge_index'], g['node_loc'],
                                  x['time_enc'], x['weekday'])                 # :294-312
    return predict(x_in, graph, W, **kw)[0]                                    # :314,325


def synth_street_map(mask, seed):
    """uint8 street map: non-zero (1..255) on the mask, zero elsewhere."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return (np.asarray(mask, dtype=bool) * rng.integers(1, 256, size=np.shape(mask))).astype(np.uint8)


def cnn_shapes():
    return {'cnn1/kernel': (3, 3, 1, 32), 'cnn2/kernel': (3, 3, 32, 8)}


def predict_sample_image(pred, img_mask, data_scale=255):
    """submission.py:69-94: scatter the node predictions [N, 96] to the image, split into 12 time
    steps, keep steps [0, 1, 2, 5, 8, 11], round(x * scale) -> uint8 [6, H, W, 8].  (tf.round is
    round-half-even; out-of-range values, which tf.cast leaves undefined, are clipped here.)"""
    h, w = img_mask.shape
    img = np.zeros((h, w, pred.shape[1]), dtype=F32)
    img[img_mask] = pred                                                       # :75-76 (raster order)
    steps = np.stack(np.split(img, 12, axis=-1))                               # :87
    steps = steps[[0, 1, 2, 5, 8, 11]]                                         # :91
    return np.clip(np.rint(steps * F32(data_scale)), 0, 255).astype(np.uint8)  # :92-93


def compute_metrics(y_true, y_pred, data_scale=255, score_only=False):
    """evaluation.py:173-244 for graph-format tensors [N, features * time] (mask=None):
    tf.keras.losses.MSE / MAE = mean over the last axis, then reduce_mean over the nodes."""
    t, p = y_true.astype(np.float64), y_pred.astype(np.float64)
    ts, ps = t * data_scale, p * data_scale                                    # :193-194
    mse = np.mean(np.mean((t - p) ** 2, axis=-1))                              # :203
    mse_scaled = np.mean(np.mean((ts - ps) ** 2, axis=-1))                     # :204
    if score_only:
        return {'mse_scaled': mse_scaled}                                      # :206-207
    speed, vol = slice(1, 8, 2), slice(0, 8, 2)                                # data/__init__.py:6-7
    out = {
        'mse': mse,
        'mae': np.mean(np.mean(np.abs(t - p), axis=-1)),                       # :209
        'mse_scaled': mse_scaled,
        'mae_scaled': np.mean(np.mean(np.abs(ts - ps), axis=-1)),              # :210-211
        'mae_scaled_speed': np.mean(np.mean(np.abs(ts[..., speed] - ps[..., speed]), axis=-1)),   # :212-215
        'mae_scaled_volume': np.mean(np.mean(np.abs(ts[..., vol] - ps[..., vol]), axis=-1)),      # :216-219
        'rmse_scaled': np.sqrt(mse_scaled),                                    # :220
    }
    ts12, ps12 = np.split(ts, 12, axis=-1), np.split(ps, 12, axis=-1)          # :232-233
    for i in (0, 1, 2, 5, 8, 11):                                              # :235-242
        out[f'mse_scaled_{(i + 1) * 5}_min'] = np.mean(np.mean((ts12[i] - ps12[i]) ** 2, axis=-1))
    return {k: float(v) for k, v in out.items()}


def model_compute_metrics(seed, target, pred, height, width):
    """GraphBaseModel.compute_metrics (models/base_model.py:333-351)."""
    metrics = compute_metrics(target, pred)
    seed_metrics = compute_metrics(target, seed, score_only=True)
    n_nodes = float(pred.shape[0])
    n_zeros = float(height * width) - n_nodes
    metrics['score'] = metrics['mse_scaled'] * n_nodes / (n_nodes + n_zeros)
    metrics['rel_score'] = float(seed_metrics['mse_scaled']) * n_nodes / (n_nodes + n_zeros) / metrics['score']
    return metrics


def predict(x, graph, W, **kw):
    """pred = max(GraphUNet(x, graph), 0)  (models/base_model.py:314,325)."""
    y, graph = graph_unet_forward(x, graph, W, **kw)
    return np.maximum(y, F32(0)), graph


# --------------------------------------------------------------------------
# checkpoint-shaped weights
# --------------------------------------------------------------------------

def gn_layer_shapes(prefix, c_in, f_e, g, units=128, with_glob_transform=True):
    u = units // 4
    shapes = {}
    for q in range(4):
        shapes[f'{prefix}/node_embeddings/{q}/kernel'] = (c_in, u)
        shapes[f'{prefix}/node_embeddings/{q}/bias'] = (u,)
        shapes[f'{prefix}/edge_transforms/{q}/kernel'] = (f_e + 3 * u, u)
        shapes[f'{prefix}/edge_transforms/{q}/bias'] = (u,)
        shapes[f'{prefix}/glob_transforms/{q}/kernel'] = (g if q == 0 else u, u)
        shapes[f'{prefix}/glob_transforms/{q}/bias'] = (u,)
    shapes[f'{prefix}/node_transform/kernel'] = (c_in + units + g, units)
    shapes[f'{prefix}/node_transform/bias'] = (units,)
    if with_glob_transform:
        shapes[f'{prefix}/glob_transform/kernel'] = (g + units + u, units)
        shapes[f'{prefix}/glob_transform/bias'] = (units,)
        shapes[f'{prefix}/global_scale_edges'] = ()
        shapes[f'{prefix}/global_scale_nodes'] = ()
    return shapes


def graph_unet_shapes(units=128, in_features=104, out_units=96, depth=3, f_e0=16, g0=113):
    """Variable name -> shape for GraphUNet.  units is the effective width
    (max(config units * 2**i, 128) = 128 for the shipped checkpoint,
    models/graph_unet.py:37-55)."""
    u = units // 4
    shapes = {'in_layer/kernel': (in_features, units), 'in_layer/bias': (units,),
              'out_model/kernel': (units, out_units)}
    f_e, g = f_e0, g0
    skip_fe, skip_g = [], []
    for d in range(depth):
        skip_fe.append(f_e)
        skip_g.append(g)
        shapes.update(gn_layer_shapes(f'downsampling_layers/{d}/gcns/0', units, f_e, g, units))
        f_e, g = u, units
    shapes.update(gn_layer_shapes('bottleneck_gcn_1', units, f_e, g, units))
    shapes.update(gn_layer_shapes('bottleneck_gcn_2', units, u, units, units))
    for d in range(depth):
        lvl = depth - 1 - d
        shapes.update(gn_layer_shapes(f'upsampling_layers/{d}/upsampling/gcn', units, 2,
                                      units + skip_g[lvl], units))
        shapes.update(gn_layer_shapes(f'upsampling_layers/{d}/gcns/0', 2 * units, skip_fe[lvl],
                                      units, units))
        last = d == depth - 1
        shapes.update(gn_layer_shapes(f'upsampling_layers/{d}/gcns/1', units, u, units, units,
                                      with_glob_transform=not last))
    return shapes


def graph_unet_block_units(units, depth=3, min_units=128):
    """Widths the constructor gives the blocks (models/graph_unet.py:37-55): in_layer and the
    down blocks max(units * 2**i, 128), both bottleneck layers 128, the up blocks
    max(units * 2**(depth - i - 1), 128).  min_units replaces the 128 (a uniform-width variant,
    e.g. units=64, min_units=256, that the reference constructor itself cannot build)."""
    down = [max(units * 2 ** i, min_units) for i in range(depth)]
    up = [max(units * 2 ** (depth - i - 1), min_units) for i in range(depth)]
    return max(units, min_units), down, min_units, up


def graph_unet_shapes_for(units, in_features=104, out_units=96, depth=3, f_e0=16, g0=113, min_units=128):
    """Variable name -> shape of GraphUNet(units, depth=depth) for any constructor `units`,
    following the widths through models/graph_unet.py:60-84 (units <= 32 gives the shipped
    checkpoint's uniform 128)."""
    w_in, down, w_mid, up = graph_unet_block_units(units, depth, min_units)
    shapes = {'in_layer/kernel': (in_features, w_in), 'in_layer/bias': (w_in,)}
    c, f_e, g = w_in, f_e0, g0
    skip_c, skip_fe, skip_g = [], [], []
    for d in range(depth):
        skip_c.append(c)
        skip_fe.append(f_e)
        skip_g.append(g)
        shapes.update(gn_layer_shapes(f'downsampling_layers/{d}/gcns/0', c, f_e, g, down[d]))
        c, f_e, g = down[d], down[d] // 4, down[d]
    shapes.update(gn_layer_shapes('bottleneck_gcn_1', c, f_e, g, w_mid))
    shapes.update(gn_layer_shapes('bottleneck_gcn_2', w_mid, w_mid // 4, w_mid, w_mid))
    c, g = w_mid, w_mid
    for d in range(depth):
        lvl = depth - 1 - d
        shapes.update(gn_layer_shapes(f'upsampling_layers/{d}/upsampling/gcn', c, 2, g + skip_g[lvl], up[d]))
        shapes.update(gn_layer_shapes(f'upsampling_layers/{d}/gcns/0', up[d] + skip_c[lvl], skip_fe[lvl],
                                      up[d], up[d]))
        shapes.update(gn_layer_shapes(f'upsampling_layers/{d}/gcns/1', up[d], up[d] // 4, up[d], up[d],
                                      with_glob_transform=d != depth - 1))
        c, g = up[d], up[d]
    # the last layer keeps the untransformed global (hybrid_gnn.py:70); nothing reads it afterwards
    shapes['out_model/kernel'] = (c, out_units)
    return shapes


def make_weights(shapes, seed=1):
    """Glorot-uniform kernels, U(-0.1, 0.1) biases (non-zero so every bias path
    is exercised), global scales 1.0.  Deterministic in (shapes order, seed)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    W = {}
    for name, shp in shapes.items():
        if len(shp) == 0:
            W[name] = F32(1.0)
        elif len(shp) == 1:
            W[name] = rng.uniform(-0.1, 0.1, shp).astype(F32)
        elif len(shp) == 4:                                     # convolution kernel [kh, kw, in, out]
            lim = np.sqrt(6.0 / (shp[0] * shp[1] * (shp[2] + shp[3])))
            W[name] = rng.uniform(-lim, lim, shp).astype(F32)
        else:
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            W[name] = rng.uniform(-lim, lim, shp).astype(F32)
    return W


def synth_sample(node_loc, edge_index_q, seed, time_idx=0, weekday=0):
    """One synthetic (city, time-window) sample on a given graph."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = node_loc.shape[0]
    seed_nodes = (rng.random((n, 96)) * (rng.random((n, 96)) < 0.3)).astype(F32)
    ts = 2 * np.pi * (np.arange(12) + time_idx) / 288.0
    time_enc = np.stack((np.sin(ts + 12), np.cos(ts)), axis=-1)[None].astype(F32)
    return seed_nodes, time_enc, np.array([weekday % 7])


def synth_street_encoding(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random((n, 8)).astype(F32)
