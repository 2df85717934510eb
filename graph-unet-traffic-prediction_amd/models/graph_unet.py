"""Graph U-Net model (reference models/graph_unet.py:9-84) on libgunet_hip.so."""
import ctypes as C

import torch

from .. import _lib, layers as L, plan as _plan
from .._lib import UNetWeightsStruct, call, current_stream, ptr, ptr4
from ..layers.hybrid_gnn import (Dense, GeoQuadrantGCN, GraphDownsamplingBlock, GraphUpsamplingBlock,
                                 _activation_code, _as_f32)
from .base_model import GraphBaseModel


class GraphUNet(GraphBaseModel):
    """Dense in -> depth x (GN layer + 2x2 pool) -> 2 bottleneck GN layers ->
    depth x (graph upsample + skip concat + 2 GN layers) -> Dense out (no bias).

    call(x, graph) takes x [N, F] (or [B, N, F]) and the graph dict and returns
    (y [N, out_units], graph), as the reference does.

    The reference floors every block width at 128 and builds both bottleneck layers 128 wide
    (models/graph_unet.py:37,40,46-49,53); `min_units` (keyword-only, not in the reference) replaces
    that 128, so GraphUNet(64, min_units=256) is 256 wide throughout.  Block widths of 128, 256, 512
    and 1024 are implemented (GraphUNet(128, depth=3) builds 128 / 256 / 512, GraphUNet(256, depth=3)
    256 / 512 / 1024); the tuned kernels serve 128 and 256, wider blocks take generic ones."""

    def __init__(self, units, out_units=8, depth=5, activation=None, layer_type=None, use_bias=False,
                 use_global=False, output_activation=None, *, min_units=128, **layer_args):
        super().__init__()
        self.depth = depth
        self.downsampling_layers = []
        self.upsampling_layers = []
        self.use_global = use_global
        self.activation = activation
        self.layer_type = layer_type
        self.use_bias = use_bias
        layer_units = units
        GraphLayer = L.get(layer_type)
        self.in_layer = Dense(max(layer_units, min_units))
        for i in range(self.depth):
            ds = GraphDownsamplingBlock(max(layer_units, min_units), activation=activation, use_bias=use_bias,
                                        diffuse_graph=False, layer_type=layer_type, use_global=use_global)
            layer_units = units * 2 ** (i + 1)
            self.downsampling_layers.append(ds)
        self.bottleneck_gcn_1 = GraphLayer(min_units, activation=self.activation, use_bias=self.use_bias,
                                           learn_global_scale=self.use_global, use_global=self.use_global)
        self.bottleneck_gcn_2 = GraphLayer(min_units, activation=self.activation, use_bias=self.use_bias,
                                           learn_global_scale=self.use_global, use_global=self.use_global)
        for i in range(self.depth):
            layer_units = units * 2 ** (self.depth - i - 1)
            us = GraphUpsamplingBlock(max(layer_units, min_units), activation=activation, use_bias=use_bias,
                                      diffuse_graph=False, layer_type=layer_type, use_global=use_global,
                                      last_block=i == (self.depth - 1))
            self.upsampling_layers.append(us)
        self.out_model = Dense(out_units, use_bias=False, activation=output_activation)
        # call() runs the whole forward through one library call (gunet_unet_forward_f32) when the
        # model is the plain stack of GeoQuadrantGCN layers; False walks the layer objects instead
        # (same kernels in the same order, identical results).
        self.native_forward = True
        # True: the one-call forward keeps its intermediates in the library's static scratch buffer
        # (gunet_unet_forward_static_f32) instead of a torch-allocated workspace; graphs whose
        # intermediates exceed _lib.STATIC_SCRATCH_BYTES raise GunetError.
        self.static_scratch = False
        self._native_struct = None
        self._native_key = None

    def _gn_layers(self):
        """(checkpoint prefix, layer) for every GeoQuadrantGCN, in the checkpoint's naming."""
        out = []
        for d, blk in enumerate(self.downsampling_layers):
            out += [(f'downsampling_layers/{d}/gcns/{l}', g) for l, g in enumerate(blk.gcns)]
        out += [('bottleneck_gcn_1', self.bottleneck_gcn_1), ('bottleneck_gcn_2', self.bottleneck_gcn_2)]
        for d, blk in enumerate(self.upsampling_layers):
            out.append((f'upsampling_layers/{d}/upsampling/gcn', blk.upsampling.gcn))
            out += [(f'upsampling_layers/{d}/gcns/{l}', g) for l, g in enumerate(blk.gcns)]
        return out

    def load_weights(self, flat, device='cuda'):
        """flat: {checkpoint variable name (without the 'net/' prefix): array}."""
        device = torch.device(device)
        self.in_layer.load_weights(flat, 'in_layer', device)
        self.out_model.load_weights(flat, 'out_model', device)
        for prefix, layer in self._gn_layers():
            layer.load_weights(flat, prefix, device)
        if 'cnn1/kernel' in flat and 'cnn2/kernel' in flat:
            self.load_frontend_weights(flat, device)
        self._native_struct = None      # the library's weight table points at the old tensors
        return self

    def named_weights(self):
        out = {}
        out.update(self.in_layer.named_weights('in_layer'))
        out.update(self.out_model.named_weights('out_model'))
        for prefix, layer in self._gn_layers():
            if layer.weights is not None:
                out.update(layer.named_weights(prefix))
        return out

    def predict(self, x, graph, training=False, out=None):
        """max(GraphUNet(x, graph), 0): what GraphBaseModel._call returns for the forward
        (reference models/base_model.py:314,325); the clamp is fused into the output GEMM.
        out (one-call forward only): a contiguous float32 tensor of the result's shape to write
        into, e.g. this batch's slice of a buffer that is gathered afterwards."""
        if self.out_model.activation not in (None, 'linear', 'relu'):
            raise NotImplementedError('predict() needs a linear or relu output activation')
        return self.call(x, graph, training=training, _out_activation='relu', _out=out)[0]

    # -- whole forward as one library call ---------------------------------------
    def _native_ok(self):
        gcns = [g for _, g in self._gn_layers()]
        blocks = self.downsampling_layers + self.upsampling_layers
        return (self.native_forward and 1 <= self.depth <= _lib.MAX_DEPTH
                and all(type(g) is GeoQuadrantGCN and g.use_global and g._act in (_lib.ACT_LINEAR, _lib.ACT_RELU)
                        for g in gcns)
                and all(not b.residual for b in blocks)
                and all(b.n_layers == 1 for b in self.downsampling_layers)
                and all(b.n_layers == 2 and b.upsampling.method == 'graph' and b.upsampling.col_idx == 0
                        for b in self.upsampling_layers))

    def _native_weights(self, c_in, f_e, g, device):
        """The library's weight table for inputs of these widths; layers without loaded weights are
        initialised on the way (Keras defaults), each with the widths the layers before it give."""
        # rebuilt when a layer's own table was dropped (its load_weights) or the widths change
        layers = [layer for _, layer in self._gn_layers()]
        def key():
            return (c_in, f_e, g, str(device), self.in_layer.kernel is None or self.in_layer.kernel.data_ptr(),
                    self.out_model.kernel is None or self.out_model.kernel.data_ptr())

        if (self._native_struct is not None and self._native_key == key()
                and all(layer._struct is not None for layer in layers)):
            return self._native_struct
        s = UNetWeightsStruct()
        keep = []

        def put(dst, layer, c, fe, gw):
            layer.build(c, fe, gw, device)
            ws = layer._weights_struct(c, fe, gw)
            keep.append(ws)
            C.memmove(C.addressof(dst), C.addressof(ws), C.sizeof(ws))
            gout = layer.units if layer.has_glob_transform else gw + layer.units + layer.units // 4
            return layer.units, layer.units // 4, gout

        self.in_layer.build(c_in, device)
        if tuple(self.in_layer.kernel.shape) != (c_in, self.in_layer.units):
            raise ValueError(f'in_layer/kernel has shape {tuple(self.in_layer.kernel.shape)}, '
                             f'the inputs need {(c_in, self.in_layer.units)}')
        s.depth, s.c_in, s.out_units = self.depth, c_in, self.out_model.units
        s.in_kernel = self.in_layer.kernel.data_ptr()
        s.in_bias = self.in_layer.bias.data_ptr() if self.in_layer.bias is not None else None
        wx, we, wg = [self.in_layer.units], [f_e], [g]
        for l, blk in enumerate(self.downsampling_layers):
            c, fe, gw = put(s.down[l], blk.gcns[0], wx[l], we[l], wg[l])
            wx.append(c); we.append(fe); wg.append(gw)
        c, fe, gw = wx[-1], we[-1], wg[-1]
        for i, layer in enumerate((self.bottleneck_gcn_1, self.bottleneck_gcn_2)):
            c, fe, gw = put(s.bottleneck[i], layer, c, fe, gw)
        for i, blk in enumerate(self.upsampling_layers):
            l = self.depth - 1 - i
            c, _, gw = put(s.up_sampling[i], blk.upsampling.gcn, c, 2, gw + wg[l])
            c, fe, gw = put(s.up_gcn[i][0], blk.gcns[0], c + wx[l], we[l], gw)
            c, fe, gw = put(s.up_gcn[i][1], blk.gcns[1], c, fe, gw)
        self.out_model.build(c, device)
        if tuple(self.out_model.kernel.shape) != (c, self.out_model.units):
            raise ValueError(f'out_model/kernel has shape {tuple(self.out_model.kernel.shape)}, '
                             f'the last layer gives {(c, self.out_model.units)}')
        s.out_kernel = self.out_model.kernel.data_ptr()
        s.out_bias = self.out_model.bias.data_ptr() if self.out_model.bias is not None else None
        s._keep = keep
        self._native_struct, self._native_key = s, key()
        self._native_widths = (fe, gw)
        return s

    def _call_native(self, x, graph, out_activation, out=None):
        x = _as_f32(x, 'x')
        single = x.dim() == 2
        xb = x.unsqueeze(0) if single else x
        if xb.dim() != 3:
            raise ValueError(f'x must be [N, C] or [B, N, C], got {tuple(x.shape)}')
        batch, n, c_in = xb.shape
        dev = x.device
        glob = _as_f32(graph['global'], "graph['global']")
        glob = glob.reshape(-1, glob.shape[-1])
        if glob.shape[0] != batch:
            raise ValueError(f"graph['global'] has {glob.shape[0]} rows for a batch of {batch}")
        edge_index, edges = graph['edge_index'], graph['edges']
        if len(edge_index) != 4 or len(edges) != 4:
            raise ValueError('GraphUNet needs 4 edge lists and 4 edge-feature tensors')
        up = _plan.get_unet_plan(graph['node_loc'], edge_index, self.depth)
        if up.n_nodes[0] != n:
            raise ValueError(f"x has {n} rows but graph['node_loc'] has {up.n_nodes[0]}")
        feats = []
        shared = all(e.dim() == 2 for e in edges) and batch > 1      # one [E_q, F] tensor for the whole batch
        for q in range(4):
            e = _as_f32(edges[q], f"graph['edges'][{q}]")
            eb = e.unsqueeze(0) if e.dim() == 2 else e
            if (eb.shape[0] != (1 if shared else batch) or eb.shape[1] != up.n_edges[0][q]
                    or eb.shape[2] != edges[0].shape[-1]):
                raise ValueError(f"graph['edges'][{q}] has shape {tuple(e.shape)}, expected "
                                 f'[{batch}, {up.n_edges[0][q]}, {edges[0].shape[-1]}] (or 2-D, shared by the batch)')
            feats.append(eb)
        ws = self._native_weights(c_in, feats[0].shape[2], glob.shape[1], dev)
        u_last, g_last = self._native_widths
        shape = (batch, n, self.out_model.units)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=dev)
        else:
            if (out.dtype != torch.float32 or out.device != dev or not out.is_contiguous()
                    or tuple(out.shape) != (shape[1:] if single else shape)):
                raise ValueError(f'out must be a contiguous float32 tensor of shape {shape[1:] if single else shape} on {dev}')
            out = out.unsqueeze(0) if single else out
        edges_out = [torch.empty((batch, up.n_edges[0][q], u_last), dtype=torch.float32, device=dev) for q in range(4)]
        global_out = torch.empty((batch, g_last), dtype=torch.float32, device=dev)
        nbytes = int(_lib.lib.gunet_unet_workspace_bytes(up.handle, C.byref(ws), batch))
        if nbytes == 0:
            raise NotImplementedError(f'GraphUNet: {_lib.last_error()}')
        in_ptrs, out_ptrs = ptr4(feats), ptr4(edges_out)
        with torch.cuda.device(dev):     # kernels go to the current device; the plan checks that it is its own
            if self.static_scratch:
                call('gunet_unet_forward_static_f32', up.handle, C.byref(ws), batch, ptr(xb), C.byref(in_ptrs),
                     1 if shared else 0, ptr(glob), _activation_code(self.activation), out_activation, ptr(out), C.byref(out_ptrs),
                     ptr(global_out), current_stream(dev))
            else:
                scratch = _plan.model_workspace(nbytes, dev)
                base = (scratch.data_ptr() + 255) // 256 * 256
                call('gunet_unet_forward_f32', up.handle, C.byref(ws), batch, ptr(xb), C.byref(in_ptrs),
                     1 if shared else 0, ptr(glob),
                     _activation_code(self.activation), out_activation, ptr(out), C.byref(out_ptrs), ptr(global_out),
                     C.c_void_p(base), scratch.numel() - (base - scratch.data_ptr()), current_stream(dev))
        if single:
            out = out[0]
            edges_out = [e[0] for e in edges_out]
        graph['edges'] = edges_out
        graph['global'] = global_out
        return out, graph

    def call(self, x, graph, training=False, _out_activation=Ellipsis, _out=None):
        if self._native_ok():
            act = self.out_model._act if _out_activation is Ellipsis else _activation_code(_out_activation)
            return self._call_native(x, graph, act, _out)
        if _out is not None:
            raise NotImplementedError('out= needs the one-call forward (native_forward)')
        skip_graphs = []
        skips = []
        x = self.in_layer(x)
        for layer in self.downsampling_layers:
            skip_graphs.append(graph.copy())
            skips.append(x)
            x, graph = layer(x, graph)
        x, graph = self.bottleneck_gcn_1(x, graph)
        x, graph = self.bottleneck_gcn_2(x, graph)
        for di, layer in enumerate(self.upsampling_layers):
            x, graph = layer(x, graph, skips[-(di + 1)], skip_graphs[-(di + 1)])
        x = self.out_model(x, activation=_out_activation)
        return x, graph

    __call__ = call
