import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import gunet_amd as amd, gunet_oracle as O
from gunet_amd import plan
from gpu_helpers import copy_graph, graph_to_dev, make_case, to_dev
size = (40, 52, 0.45)
x, graph, _ = make_case(O, *size)
W = O.make_weights(O.graph_unet_shapes(), 1)
want_y, want_g = O.graph_unet_forward(x, copy_graph(graph), W)
model = amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn', use_bias=True, use_global=True, output_activation=None).load_weights(W, 'cuda')
amd.set_matmul_mode('bf16')
g0 = graph_to_dev(graph)
xd = to_dev(x)
def run(tag, newplan):
    g = graph_to_dev(graph) if newplan else dict(g0)
    got_y, got_g = model(xd, g)
    torch.cuda.synchronize()
    got_y = got_y.cpu().numpy()
    err = np.abs(got_y - want_y).max(axis=1) / np.abs(want_y).max()
    bad = np.flatnonzero(~(err <= 2e-2))
    print(tag, 'newplan', newplan, 'max err', err.max(), 'bad rows', len(bad), bad[:20])
run('first', False)
for fill in (0xFF, 0xFF, 0xFF, 0xFF):
    for newplan in (False, True):
        for b in plan._model_workspaces.values():
            b.fill_(fill)
        torch.cuda.synchronize()
        run(f'fill {fill}', newplan)
junk = [torch.full((1 << 20,), float('nan'), device='cuda') for _ in range(8)]
del junk
for newplan in (False, True, False, True):
    run('after nan junk', newplan)
