import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import gunet_amd as amd, gunet_oracle as O
from gunet_amd import plan
from gpu_helpers import copy_graph, graph_to_dev, make_case, to_dev
mode = sys.argv[1] if len(sys.argv) > 1 else 'f32'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
x, graph, _ = make_case(O, 40, 52, 0.45)
W = O.make_weights(O.graph_unet_shapes(), 1)
model = amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn', use_bias=True, use_global=True, output_activation=None).load_weights(W, 'cuda')
amd.set_matmul_mode(mode)
g0 = graph_to_dev(graph)
xd = to_dev(x)
ref, _ = model(xd, dict(g0))
ref = ref.clone()
bad = 0
for it in range(iters):
    for b in plan._model_workspaces.values():
        b.fill_(0xFF)
    y, _ = model(xd, dict(g0))
    if not torch.equal(y, ref):
        bad += 1
        d = (y - ref).abs().max(dim=1).values
        rows = torch.nonzero(d > 0).flatten()[:8].tolist()
        print('iteration', it, 'differs in rows', rows, 'max diff', float(d.max()))
print(mode, 'iterations', iters, 'mismatching', bad)
