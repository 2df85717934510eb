"""Eager vs HIP-graph replay of the full forward at small batch sizes (development tool)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gunet_amd
from gunet_amd import synthetic

node_loc, edge_index, direction = synthetic.road_graph(495, 436, 0.45, 0)
dev = torch.device('cuda')
model = gunet_amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                            use_bias=True, use_global=True)
model.load_weights(synthetic.random_weights(synthetic.graph_unet_shapes(), 1), dev)
quads = gunet_amd.data_utils.sample_quadrant_subgraphs(torch.as_tensor(edge_index).to(dev), torch.as_tensor(direction).to(dev))
quads_np = [q.cpu().numpy() for q in quads]
enc = synthetic.street_encoding(node_loc.shape[0], 5)
for B in (1, 2, 8):
    sn, te, wd = synthetic.sample_inputs(node_loc.shape[0], 1000)
    x, edges_np, glob = synthetic.assemble_model_inputs(sn, enc, quads_np, te, wd)
    xt = torch.as_tensor(x).to(dev).unsqueeze(0).repeat(B, 1, 1)
    g = {'edge_index': list(quads), 'node_loc': torch.as_tensor(node_loc).to(dev),
         'edges': [torch.as_tensor(e).to(dev).unsqueeze(0).repeat(B, 1, 1).contiguous() for e in edges_np],
         'global': torch.as_tensor(glob).to(dev).repeat(B, 1)}

    def timeit(fn, n=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    model.native_forward = False
    layerwise = timeit(lambda: model.predict(xt, dict(g)))
    model.native_forward = True
    eager = timeit(lambda: model.predict(xt, dict(g)))
    cap = gunet_amd.CapturedForward(model, xt, g)
    replay = timeit(lambda: cap(xt))
    print(f'B={B}: layer objects {layerwise:.3f} ms, one call {eager:.3f} ms ({B / eager * 1e3:.0f} samples/s), graph replay {replay:.3f} ms ({B / replay * 1e3:.0f} samples/s)', flush=True)
