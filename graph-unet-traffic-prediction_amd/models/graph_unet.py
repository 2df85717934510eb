"""Graph U-Net model (reference models/graph_unet.py:9-84) on libgunet_hip.so."""
import torch

from .. import layers as L
from ..layers.hybrid_gnn import Dense, GraphDownsamplingBlock, GraphUpsamplingBlock
from .base_model import GraphBaseModel


class GraphUNet(GraphBaseModel):
    """Dense in -> depth x (GN layer + 2x2 pool) -> 2 bottleneck GN layers ->
    depth x (graph upsample + skip concat + 2 GN layers) -> Dense out (no bias).

    call(x, graph) takes x [N, F] (or [B, N, F]) and the graph dict and returns
    (y [N, out_units], graph), as the reference does.

    The reference floors every block width at 128 and builds both bottleneck layers 128 wide
    (models/graph_unet.py:37,40,46-49,53); `min_units` (keyword-only, not in the reference) replaces
    that 128, so GraphUNet(64, min_units=256) is 256 wide throughout.  Block widths of 128 and 256
    are implemented."""

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
        return self

    def named_weights(self):
        out = {}
        out.update(self.in_layer.named_weights('in_layer'))
        out.update(self.out_model.named_weights('out_model'))
        for prefix, layer in self._gn_layers():
            if layer.weights is not None:
                out.update(layer.named_weights(prefix))
        return out

    def predict(self, x, graph, training=False):
        """max(GraphUNet(x, graph), 0): what GraphBaseModel._call returns for the forward
        (reference models/base_model.py:314,325); the clamp is fused into the output GEMM."""
        if self.out_model.activation not in (None, 'linear', 'relu'):
            raise NotImplementedError('predict() needs a linear or relu output activation')
        return self.call(x, graph, training=training, _out_activation='relu')[0]

    def call(self, x, graph, training=False, _out_activation=Ellipsis):
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
