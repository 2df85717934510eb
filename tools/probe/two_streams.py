"""Probe: one batch of 8 as one forward on one stream, against two forwards of 4 on two streams that
run concurrently (development experiment, not part of the library)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import gunet_amd
from gunet_amd import synthetic

args = bench.parse_args(['--no-cpu-baseline'])
mode = sys.argv[1] if len(sys.argv) > 1 else 'f32'
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda', 0)
gunet_amd.set_matmul_mode(mode)
node_loc, edge_index, direction = synthetic.road_graph(495, 436, 0.45, 0)
n = node_loc.shape[0]
model, weights = bench.make_model(args, dev, gunet_amd, synthetic)
quads = gunet_amd.data_utils.sample_quadrant_subgraphs(torch.as_tensor(edge_index).to(dev), torch.as_tensor(direction).to(dev))
quads_np = [q.cpu().numpy() for q in quads]
enc = synthetic.street_encoding(n, 5)
B = 8
xs, globs = [], []
for b in range(B):
    sn, te, wd = synthetic.sample_inputs(n, 1000 + b, time_idx=12 * b, weekday=b)
    x, edges_np, glob = synthetic.assemble_model_inputs(sn, enc, quads_np, te, wd)
    xs.append(torch.as_tensor(x))
    globs.append(torch.as_tensor(glob))
x_dev = torch.stack(xs).to(dev)
glob_dev = torch.cat(globs, dim=0).to(dev)
edges_dev = [torch.as_tensor(e).to(dev) for e in edges_np]
node_loc_dev = torch.as_tensor(node_loc).to(dev)


def forward(lo, hi):
    graph = {'edge_index': list(quads), 'edges': list(edges_dev), 'node_loc': node_loc_dev, 'global': glob_dev[lo:hi]}
    return model.predict(x_dev[lo:hi], graph)


def timed(fn, steps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


whole = forward(0, B).clone()
t_one = timed(lambda: forward(0, B))
streams = [torch.cuda.Stream() for _ in range(parts)]
cut = [B * i // parts for i in range(parts + 1)]
outs = [None] * parts


def split():
    cur = torch.cuda.current_stream()
    for i, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs[i] = forward(cut[i], cut[i + 1])
    for s in streams:
        cur.wait_stream(s)


split()
torch.cuda.synchronize()
same = all(torch.equal(outs[i], whole[cut[i]:cut[i + 1]]) for i in range(parts))
t_split = timed(split)
t_half = timed(lambda: forward(0, B // parts))
print(f'{mode}: one forward of {B}: {t_one:.3f} ms; {parts} concurrent forwards of {B // parts}: {t_split:.3f} ms '
      f'(bit-identical: {same}); one forward of {B // parts} alone: {t_half:.3f} ms')
