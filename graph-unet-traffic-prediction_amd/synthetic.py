"""Synthetic Traffic4Cast-shaped workloads (host side, numpy): road graphs, checkpoint-shaped
random weights and model inputs.  No dataset or pretrained weights ship with the reference
(README.md:43; the checkpoint's data shard is absent), so benchmarks and tests run on these.

The graph follows the reference's image_to_graph conventions (data/data_utils.py:95-150):
nodes are the pixels with at least one outgoing edge, numbered in raster order; edges are sorted
by (source pixel, direction) with directions N, NE, E, SE, S, SW, W, NW = 0..7 and a self-loop
per node as direction 8; node_loc holds (row, col).
"""
import numpy as np

_DX = np.array([0, 1, 1, 1, 0, -1, -1, -1])
_DY = np.array([-1, -1, 0, 1, 1, 1, 0, -1])


def graph_from_mask(mask):
    """mask [H, W] bool -> node_loc int64 [N, 2], edge_index int32 [E, 2], direction int64 [E].
    Pixel p is linked to each of its 8 neighbours that is also set in the mask."""
    mask = np.asarray(mask, dtype=bool)
    h, w = mask.shape
    padded = np.zeros((h + 2, w + 2), dtype=bool)
    padded[1:-1, 1:-1] = mask
    has = np.zeros((9, h, w), dtype=bool)
    for d in range(8):
        has[d] = mask & padded[1 + _DY[d]:1 + _DY[d] + h, 1 + _DX[d]:1 + _DX[d] + w]
    is_node = has[:8].any(axis=0)
    has[8] = is_node
    node_id = np.full(h * w, -1, dtype=np.int64)
    pixels = np.flatnonzero(is_node)
    node_id[pixels] = np.arange(pixels.size)
    # (pixel, direction) pairs in pixel-major order
    pix, direction = np.nonzero(has.reshape(9, -1).T)
    offset = np.concatenate((_DX + _DY * w, [0]))[direction]
    edge_index = np.stack((node_id[pix], node_id[pix + offset]), axis=-1).astype(np.int32)
    node_loc = np.stack((pixels // w, pixels % w), axis=-1).astype(np.int64)
    return node_loc, edge_index, direction.astype(np.int64)


def road_graph(height=495, width=436, density=0.45, seed=0):
    """mask = PCG64(seed).random((H, W)) < density."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return graph_from_mask(rng.random((height, width)) < density)


def dense_graph(height=495, width=436):
    return graph_from_mask(np.ones((height, width), dtype=bool))


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
    """Variable name -> shape of a GraphUNet with the shipped checkpoint's configuration
    (width 128 at every level, depth 3, 12x8 input + 8 street channels, 12x8 output)."""
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
        shapes.update(gn_layer_shapes(f'upsampling_layers/{d}/gcns/0', 2 * units, skip_fe[lvl], units, units))
        shapes.update(gn_layer_shapes(f'upsampling_layers/{d}/gcns/1', units, u, units, units,
                                      with_glob_transform=d != depth - 1))
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


def cnn_shapes():
    """The street-map CNN of the model's caller (models/base_model.py:263-264)."""
    return {'cnn1/kernel': (3, 3, 1, 32), 'cnn2/kernel': (3, 3, 32, 8)}


def street_map(mask, seed=3):
    """uint8 street map: non-zero (1..255) on the mask, zero elsewhere."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return (np.asarray(mask, dtype=bool) * rng.integers(1, 256, size=np.shape(mask))).astype(np.uint8)


def random_weights(shapes, seed=1):
    """Glorot-uniform kernels, U(-0.1, 0.1) biases, global scales 1."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shp in shapes.items():
        if len(shp) == 0:
            out[name] = np.float32(1.0)
        elif len(shp) == 1:
            out[name] = rng.uniform(-0.1, 0.1, shp).astype(np.float32)
        elif len(shp) == 4:                                     # convolution kernel [kh, kw, in, out]
            lim = np.sqrt(6.0 / (shp[0] * shp[1] * (shp[2] + shp[3])))
            out[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
        else:
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            out[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
    return out


def street_encoding(n_nodes, seed=5):
    """Stand-in for the street-map CNN output gathered at the nodes: [N, 8] in [0, 1)."""
    return np.random.Generator(np.random.PCG64(seed)).random((n_nodes, 8)).astype(np.float32)


def sample_inputs(n_nodes, seed, time_idx=0, weekday=0):
    """One (city, time-window) sample: seed_nodes [N, 96] (sparse traffic in [0, 1)),
    time_enc [1, 12, 2] (train_utils.py:49-63 in the reference), weekday [1]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    seed_nodes = (rng.random((n_nodes, 96)) * (rng.random((n_nodes, 96)) < 0.3)).astype(np.float32)
    ts = 2 * np.pi * (np.arange(12) + time_idx) / 288.0
    time_enc = np.stack((np.sin(ts + 12), np.cos(ts)), axis=-1)[None].astype(np.float32)
    return seed_nodes, time_enc, np.array([weekday % 7])


def assemble_model_inputs(seed_nodes, street_enc, quads, time_enc, weekday):
    """What GraphBaseModel._call (models/base_model.py:287-312) hands to GraphUNet, on the host:
    x [N, 104], edge features 4 x [E_q, 16], global [1, 113]."""
    edges = [np.concatenate((street_enc[q[:, 0]], street_enc[q[:, 1]]), axis=-1) for q in quads]
    x = np.concatenate((seed_nodes, street_enc), axis=-1).astype(np.float32)
    glob = np.sum(x, axis=0, keepdims=True, dtype=np.float64).astype(np.float32) / np.float32(300. * 300.)
    week = np.eye(7, dtype=np.float32)[np.asarray(weekday).reshape(-1)]
    glob = np.concatenate((glob, time_enc[:, -1].astype(np.float32), week), axis=-1)
    return x, edges, glob
