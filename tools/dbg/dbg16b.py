import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import gunet_amd as amd, gunet_oracle as O
from gunet_amd import _lib, plan
from gunet_amd._lib import call, current_stream, ptr, ptr4
from gpu_helpers import copy_graph, graph_to_dev, make_case, to_dev
size = tuple(float(v) if '.' in v else int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (40, 52, 0.45)
x, graph, _ = make_case(O, *size)
W = O.make_weights(O.graph_unet_shapes(), 1)
keep = {}
want_y, want_g = O.graph_unet_forward(x, copy_graph(graph), W, keep=keep)
keep['in_x'] = O.dense(x, W['in_layer/kernel'], W['in_layer/bias'])
model = amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn', use_bias=True, use_global=True, output_activation=None).load_weights(W, 'cuda')
amd.set_matmul_mode('bf16')
g = graph_to_dev(graph)
xb = to_dev(x[None])
feats = [e.unsqueeze(0).contiguous() for e in g['edges']]
glob = g['global'].contiguous()
up = plan.get_unet_plan(g['node_loc'], g['edge_index'], 3)
ws = model._native_weights(104, 16, 113, xb.device)
nbytes = int(_lib.lib.gunet_unet_workspace_bytes(up.handle, C.byref(ws), 1))
scratch = torch.zeros(nbytes + 512, dtype=torch.uint8, device='cuda')
base = (scratch.data_ptr() + 255) // 256 * 256
n = xb.shape[1]
out = torch.zeros((1, n, 96), device='cuda')
eo = [torch.zeros((1, up.n_edges[0][q], 32), device='cuda') for q in range(4)]
go = torch.zeros((1, 288), device='cuda')
ip, op = ptr4(feats), ptr4(eo)

def run():
    call('gunet_unet_forward_f32', up.handle, C.byref(ws), 1, ptr(xb), C.byref(ip), 0, ptr(glob), 1, 0, ptr(out), C.byref(op), ptr(go),
         C.c_void_p(base), nbytes, current_stream(xb.device))
    torch.cuda.synchronize()
    y = out[0].cpu().numpy()
    err = np.abs(y - want_y).max(axis=1) / np.abs(want_y).max()
    return float(np.nanmax(err)) if not np.isnan(err).all() else float('nan'), np.flatnonzero(~(err <= 2e-2))
b0 = base - scratch.data_ptr()
off = [b0]
def take(floats):
    o = (off[0] - b0 + 255) // 256 * 256 + b0
    off[0] = o + 4 * max(floats, 0)
    return o
B = 1
wx, we, wg = [128] * 4, [16, 32, 32, 32], [113, 128, 128, 128]
regions = []
for l in range(4):
    o = take(B * up.n_nodes[l] * wx[l]); regions.append((f'skip_x{l}', o, off[0]))
    if l > 0:
        o = take(B * wg[l]); regions.append((f'skip_g{l}', o, off[0]))
        for q in range(4):
            o = take(B * up.n_edges[l][q] * we[l]); regions.append((f'skip_e{l}{q}', o, off[0]))
take(0); take(0)
max_rows = max(up.n_nodes); max_edges = [max(up.n_edges[l][q] for l in range(4)) for q in range(4)]
for i in range(2):
    o = take(B * max_rows * 128); regions.append((f'tmp_x{i}', o, off[0]))
    for q in range(4):
        o = take(B * max_edges[q] * 32); regions.append((f'tmp_e{i}{q}', o, off[0]))
regions.append(('rest(globals+layer_ws)', off[0], b0 + nbytes))

def bf16_at(o, rows, cols):
    raw = scratch[o:o + rows * cols * 2].view(torch.bfloat16).reshape(rows, cols)
    return raw.float().cpu().numpy()
reg = {name: lo for name, lo, hi in regions}
def report(name, got, want):
    err = np.abs(got - want).max(axis=1) / np.abs(want).max()
    bad = np.flatnonzero(~(err <= 2e-2))
    print(f'   {name:12s} rows {got.shape[0]:6d} max err {np.nanmax(err):.4f} bad {len(bad)} {bad[:12]}')
for it in range(300):
    scratch.fill_(0xFF)
    torch.cuda.synchronize()
    e, bad = run()
    if len(bad):
        print('iteration', it, 'err', e, 'bad', bad)
        report('in_layer', bf16_at(reg['skip_x0'], up.n_nodes[0], 128), keep['in_x'])
        got = bf16_at(reg['skip_x0'], up.n_nodes[0], 128); want = keep['in_x']
        errm = np.abs(got - want) / np.abs(want).max()
        rows = np.flatnonzero(~(errm.max(axis=1) <= 2e-2))
        for r in rows[:4]:
            cols = np.flatnonzero(~(errm[r] <= 2e-2))
            print('   row', r, 'bad cols', len(cols), cols[:40])
            c = cols[0]
            print('     got', got[r, c:c+4], 'want', want[r, c:c+4])
            # does the bad row equal some other row of the reference?
            d = np.abs(want - got[r][None, :]).max(axis=1)
            print('     closest reference row', int(np.argmin(d)), float(d.min()))
        break
        for d in range(3):
            report(f'down{d}_x', bf16_at(reg[f'skip_x{d+1}'], up.n_nodes[d + 1], 128), keep[f'down{d}_x'])
        for i in range(2):
            report(f'tmp_x{i}~up2', bf16_at(reg[f'tmp_x{i}'], up.n_nodes[0], 128), keep['up2_x'])
        for q in range(4):
            eo_q = eo[q][0].cpu().numpy()
            er = np.abs(eo_q - want_g['edges'][q]).max(axis=1) / np.abs(want_g['edges'][q]).max()
            print('   edges_out', q, 'bad', np.flatnonzero(~(er <= 2e-2))[:16])
print('done')
