import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import gunet_amd as amd, gunet_oracle as O
from gpu_helpers import copy_graph, graph_to_dev, make_case, to_dev
size = (40, 52, 0.45)
x, graph, _ = make_case(O, *size)
W = O.make_weights(O.graph_unet_shapes(), 1)
want_y, want_g = O.graph_unet_forward(x, copy_graph(graph), W)
model = amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn', use_bias=True, use_global=True, output_activation=None).load_weights(W, 'cuda')
amd.set_matmul_mode('bf16')
for choice in ('auto',):
    got_y, got_g = model(to_dev(x), graph_to_dev(graph))
    got_y = got_y.cpu().numpy()
    err = np.abs(got_y - want_y).max(axis=1) / np.abs(want_y).max()
    bad = np.flatnonzero(err > 2e-2)
    print('n', got_y.shape[0], 'max err', err.max(), 'bad rows', len(bad), bad[:40])
    print('global err', np.abs(got_g['global'].cpu().numpy() - want_g['global']).max() / np.abs(want_g['global']).max())
    for q in range(4):
        e = got_g['edges'][q].cpu().numpy(); w = want_g['edges'][q]
        er = np.abs(e - w).max(axis=1) / np.abs(w).max()
        print('edges', q, e.shape, er.max(), np.flatnonzero(er > 2e-2)[:20])
    model.native_forward = False
    y2, g2 = model(to_dev(x), graph_to_dev(graph))
    model.native_forward = True
    print('per-layer bf16 path err', np.abs(y2.cpu().numpy() - want_y).max() / np.abs(want_y).max())
