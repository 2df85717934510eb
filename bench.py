#!/usr/bin/env python3
"""Benchmark of the Graph U-Net forward hot path on MI355X.

Workloads (`--config`, numbered as the five benchmark configurations of this path):
  2 (default)  full GraphUNet forward at the shipped checkpoint's shapes (12x8-channel input, 12x8-channel
               output, width 128, depth 3, 14 GeoQuadrantGCN layers) on one synthetic 495x436 road graph
               (mask density 0.45, seed 0), a batch of 8 independent (city, time-window) samples per GPU,
               float32.  One step = one forward of the batch, inputs resident in HBM, per-city index plan
               cached (the plan build is timed separately and reported in `config`).
  1            one GeoQuadrantGCN layer (first down block's shapes) on that graph, batch 1.
  3            the 2x2 pooling chain of all levels alone (node + 4 edge-feature sets, batch 32): HBM-bound.
  4            six cities (seeds 10..15, densities 0.25 .. 0.60), 64 samples round-robin over the cities,
               sharded contiguous-by-city over the GPUs; one step = all 64 samples (strong scaling).
  5            dense 495x436 grid, every block 256 wide (= --dense --min-units 256): matrix-core stress.

`--gpus N` (N > 1) runs one process per GPU over RCCL: started by torchrun (WORLD_SIZE in the
environment) this process is one of the ranks; started plainly it launches the N ranks itself
(`python -m torch.distributed.run`) before touching any GPU and relays their output.  Per step the
predictions are gathered on one rank (the collector rotates) and a small metrics vector is
all-reduced, overlapped with the next step.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CITY_SEEDS = (10, 11, 12, 13, 14, 15)
CITY_DENSITIES = (0.25, 0.30, 0.35, 0.40, 0.50, 0.60)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=2, choices=[1, 2, 3, 4, 5])
    ap.add_argument('--batch', type=int, default=None, help='samples per GPU per step (config 2/5: 8, 1: 1, 3: 32)')
    ap.add_argument('--samples', type=int, default=64, help='config 4: samples of the whole job per step')
    ap.add_argument('--height', type=int, default=495)
    ap.add_argument('--width', type=int, default=436)
    ap.add_argument('--density', type=float, default=0.45)
    ap.add_argument('--dense', action='store_true', help='all pixels are nodes (MFMA stress graph)')
    ap.add_argument('--min-units', type=int, default=128, choices=[128, 256, 512, 1024],
                    help='width of every block (128 = the shipped checkpoint; 256 = the wide stress variant)')
    ap.add_argument('--units', type=int, default=32,
                    help="the reference constructor's `units` (block i is max(units * 2**i, min-units) wide)")
    ap.add_argument('--gather', default='f32', choices=['f32', 'u8', 'none'],
                    help='N > 1: what is gathered per step -- float32 predictions, the uint8 prediction image '
                         '(submission format, 6 horizons), or nothing')
    ap.add_argument('--expanded-edges', action='store_true',
                    help='config 2/5: give every sample its own copy of the (city-static) input edge features')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-runs', type=int, default=7, help='timed CPU forwards (about 1.5 s each)')
    ap.add_argument('--no-profile', action='store_true', help='skip the per-kernel event pass')
    ap.add_argument('--no-alt-modes', action='store_true', help='skip the short bf16-operand run reported beside the result')
    ap.add_argument('--matmul', default=os.environ.get('GUNET_BENCH_MATMUL', 'f32'), choices=['f32', 'split3', 'split2', 'bf16'],
                    help='matrix-product arithmetic (see gunet_set_matmul_mode)')
    ap.add_argument('--master-port', type=int, default=0, help='rendezvous port when this process launches the ranks')
    ap.add_argument('--stub', action='store_true',
                    help='CPU rehearsal of the multi-rank plumbing: gloo instead of RCCL, and a stand-in forward that '
                         'writes the sample number into every row (no GPU, no library call)')
    args = ap.parse_args(argv)
    if args.config == 5:
        args.dense, args.min_units = True, max(args.min_units, 256)
    if args.batch is None:
        args.batch = {1: 1, 3: 32}.get(args.config, 8)
    return args


# ------------------------------------------------------------------------------------------------
# ranks

def launch_ranks(args):
    """Start `--gpus` ranks of this script with torch.distributed.run and relay their output.  The
    caller has not touched a GPU: a process that has initialised HIP must not be replaced or forked."""
    if not args.stub:
        import torch
        have = torch.cuda.device_count()          # counting devices does not initialise the GPU runtime
        if have < args.gpus:
            sys.stderr.write(f'bench.py: --gpus {args.gpus} but this node shows {have} GPU(s) '
                             f'(torch.cuda.device_count()); nothing was started\n')
            return 2
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    # The image this runs in exports HSA_ENABLE_IPC_MODE_LEGACY=0 and documents why: the pool's host
    # driver only supports dmabuf IPC, and with the legacy mode RCCL's buffer sharing fails with
    # `hipIpcGetMemHandle: invalid argument`.  That is the image's statement, not a measurement made
    # here; the default only fills the variable in for an environment that lost it.
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class Ranks:
    """world / rank / device of this process, and the process group when there is more than one rank."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        if self.world != args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={self.world} ranks')
        if args.stub:
            self.dev = torch.device('cpu')
        else:
            torch.cuda.set_device(self.local_rank if self.world > 1 else 0)
            self.dev = torch.device('cuda', torch.cuda.current_device())
        if self.world > 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # see launch_ranks
            if args.stub:
                dist.init_process_group('gloo')
            else:
                dist.init_process_group('nccl', device_id=self.dev)
        self.dist = dist

    def sync(self):
        import torch
        if self.dev.type == 'cuda':
            torch.cuda.synchronize()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        import torch
        if self.world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def ranks_seen(self):
        """[(rank, device index)] of every rank that took part (gathered on all ranks)."""
        mine = (self.rank, self.dev.index if self.dev.type == 'cuda' else -1)
        if self.world == 1:
            return [list(mine)]
        got = [None] * self.world
        self.dist.all_gather_object(got, mine)
        return [list(g) for g in got]

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def timed_steps(step, drain, args, ranks):
    """`warmup` untimed steps, then exactly `steps` timed ones between barrier + device sync on both
    sides; the maximum over the ranks, in seconds."""
    for i in range(args.warmup):
        step(i)
    drain()
    ranks.barrier()
    ranks.sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    ranks.sync()
    ranks.barrier()
    return ranks.max_over_ranks(time.perf_counter() - t0)


class PhaseClock:
    """Where a multi-rank step spends its time, per rank: marks on the compute stream (device
    events on a GPU, host clock on the CPU rehearsal) at the start of a step, after the forward and
    after the exchange was posted.  `forward` is the device time of the forward; `exchange` is what
    the step spends on the exchange on top of it: waiting for the transfers of two steps ago that
    still hold the slot, staging this step's payload and queueing the collectives.  Steps before
    `reset()` (the warm-up) are dropped."""

    def __init__(self, dev):
        self.cuda = dev.type == 'cuda'
        self.marks = []

    def _now(self):
        if not self.cuda:
            return time.perf_counter()
        import torch
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def reset(self):
        self.marks = []

    def start(self):
        self.marks.append([self._now()])

    def mark(self):
        self.marks[-1].append(self._now())

    def summary(self):
        """(forward_ms, exchange_ms) averaged over the recorded steps; call after a device sync."""
        fw = ex = 0.0
        done = [m for m in self.marks if len(m) == 3]
        for a, b, c in done:
            if self.cuda:
                fw += a.elapsed_time(b)
                ex += b.elapsed_time(c)
            else:
                fw += (b - a) * 1e3
                ex += (c - b) * 1e3
        n = max(1, len(done))
        return fw / n, ex / n


def multi_rank_report(args, ranks, clock, exchange, forward_only, payload_bytes):
    """For more than one rank: where the timed steps spent their time (maximum over the ranks of each
    rank's per-step averages), what a rank ships per step, and the same number of steps timed once more
    without the exchange -- so a curve that bends says whether the forward or the exchange bent it.
    Every rank must call this (it times steps between barriers).  {} for a single rank."""
    if ranks.world == 1:
        return {}
    ranks.sync()
    forward_ms, exchange_ms = clock.summary()
    out = {'forward_ms': ranks.max_over_ranks(forward_ms), 'exchange_wait_ms': ranks.max_over_ranks(exchange_ms),
           'payload_MB_per_rank_and_step': payload_bytes / 1e6}
    if exchange is not None:
        bare = timed_steps(forward_only, lambda: None, args, ranks)
        out['ms_per_step_without_exchange'] = bare / args.steps * 1e3
    return out


# ------------------------------------------------------------------------------------------------
# roofline accounting

GEMM_CLASSES = ('gemm_tiled', 'gemm_resident_k128', 'gemm_resident_k256', 'gemm_resident_k384', 'gemm_resident_bf16',
                'gemm_split')
MFMA_CLASSES = GEMM_CLASSES + ('edge_mlp',)
BYTE_CLASSES = ('aggregate', 'segment_reduce')
KERNEL_OF_CLASS = {'gemm_tiled': 'k_gemm_f32', 'gemm_resident_k128': 'k_gemm_resident<*, 4>',
                   'gemm_resident_k256': 'k_gemm_resident<*, 8>', 'gemm_resident_k384': 'k_gemm_resident<*, 12>',
                   'gemm_resident_bf16': 'k_gemm_resident<.., bf16>', 'gemm_split': 'k_gemm_split', 'edge_mlp': 'k_edge_mlp',
                   'aggregate': 'k_aggregate', 'segment_reduce': 'k_pool_max', 'layer_prologue': 'k_layer_prologue',
                   'global_update': 'k_global_update', 'global_bridge': 'k_global_bridge', 'gather_rows': 'k_gather_rows',
                   'dense_naive': 'k_dense_naive'}


def source_stamp():
    """Hash of the library's sources: identifies the kernels a counter table was measured on."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, 'graph-unet-traffic-prediction_amd', 'csrc')
    for name in sorted(os.listdir(src)):
        if name.endswith(('.hip', '.h')):
            with open(os.path.join(src, name), 'rb') as f:
                h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def workload_key(args):
    """Key of the committed counter tables: only the default shape of each configuration has one."""
    default_shape = (args.height, args.width, args.units) == (495, 436, 32)
    if not default_shape or args.matmul not in ('f32', 'bf16'):
        return None
    if args.matmul == 'bf16':
        ok = args.config == 2 and (args.density, args.dense, args.min_units, args.batch) == (0.45, False, 128, 8)
        return 'cfg2_bf16' if ok else None
    if args.config in (1, 2) and (args.density, args.dense, args.min_units) == (0.45, False, 128) and \
            args.batch == {1: 1, 2: 8}[args.config]:
        return f'cfg{args.config}'
    if args.config == 3 and (args.density, args.dense, args.batch) == (0.45, False, 32):
        return 'cfg3'
    if args.config == 5 and args.min_units == 256 and args.batch == 8:
        return 'cfg5'
    return None


def measured_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes of this same
    command (profiles/hbm_traffic_per_launch.json: FETCH_SIZE doubled as gfx950 needs, plus
    WRITE_SIZE), with `stale` = the library sources have changed since those passes; None when the
    workload or the kernel has no entry."""
    key = workload_key(args)
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic_per_launch.json')
    if key is None or not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    wl = table.get('workloads', {}).get(key)
    entry = (wl or {}).get('kernels', {}).get(kernel)
    if entry is None:
        return None
    return {'bytes_per_launch': entry['hbm_MB_per_launch'] * 1e6, 'source': wl.get('source', path),
            'stale': wl.get('source_stamp') != source_stamp()}


def measured_step_traffic(args):
    """HBM bytes one step (forward) moves, summed over every feature kernel of the committed counter
    passes of this workload (profiles/hbm_traffic_per_launch.json: `feature_MB_per_forward`), and whether
    the library sources have changed since; None when the workload has no table."""
    key = workload_key(args)
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic_per_launch.json')
    if key is None or not os.path.exists(path):
        return None
    with open(path) as f:
        wl = json.load(f).get('workloads', {}).get(key)
    if not wl or 'feature_MB_per_forward' not in wl:
        return None
    return {'bytes': wl['feature_MB_per_forward'] * 1e6, 'stale': wl.get('source_stamp') != source_stamp()}


def roofline_from_profile(prof, args, workload):
    """The `roofline` object: the kernel class with the most device time, its algorithmic work as the
    library counted it for exactly those launches, over its summed HIP-event durations."""
    dom = max(prof, key=lambda k: prof[k][0])
    ms, n, work_done = prof[dom]
    if n == 0:
        return None
    products = {'f32': 1, 'split3': 6, 'split2': 3, 'bf16': 1}[args.matmul] if dom == 'gemm_split' else 1
    if dom in MFMA_CLASSES:
        peak = workload.PEAK_BF16_MFMA_TFLOPS if dom in ('gemm_split', 'gemm_resident_bf16') else workload.PEAK_F32_MFMA_TFLOPS
        achieved = products * work_done / (ms * 1e-3) / 1e12
        roof = {'kernel': KERNEL_OF_CLASS[dom], 'bound': 'mfma', 'achieved': achieved, 'peak': peak,
                'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': None, 'avg_launch_ms': ms / n,
                'gflop_per_launch': work_done / n / 1e9}
        if products > 1:
            roof['note'] = f'{products} bf16 MFMA products per float32-class multiply-add; peak = dense bf16'
    else:
        achieved = work_done / (ms * 1e-3) / 1e9
        roof = {'kernel': KERNEL_OF_CLASS[dom], 'bound': 'hbm', 'achieved': achieved, 'peak': workload.PEAK_HBM_GBPS,
                'unit': 'GB/s', 'frac': achieved / workload.PEAK_HBM_GBPS, 'traffic': None,
                'avg_launch_ms': ms / n, 'MB_per_launch': work_done / n / 1e6}
    traffic = measured_traffic(KERNEL_OF_CLASS[dom], args)
    if traffic is not None:
        roof['traffic'] = traffic['bytes_per_launch']
        roof['traffic_source'] = traffic['source']
        roof['traffic_stale'] = traffic['stale']
        if dom == 'gemm_resident_bf16':
            # K = 128..384 against 128 output columns in bf16 is 43..85 FLOP per byte, far left of the bf16
            # ridge (312): these launches are priced by the bytes they move, not by the matrix peak
            rate = traffic['bytes_per_launch'] / (ms / n * 1e-3) / 1e9
            roof['measured_hbm_GBps'] = rate
            roof['frac_of_hbm_peak'] = rate / workload.PEAK_HBM_GBPS
            roof['note'] = 'memory-side kernel at bf16 operand widths: see measured_hbm_GBps; frac is against the dense bf16 matrix peak'
    return roof


def kernel_table(prof, steps):
    return {k: {'ms_per_step': v[0] / steps, 'launches_per_step': v[1] / steps,
                ('gflop_per_step' if k in MFMA_CLASSES else 'MB_per_step'):
                    v[2] / steps / (1e9 if k in MFMA_CLASSES else 1e6)}
            for k, v in prof.items() if v[1]}


def profile_pass(fn, steps, ranks):
    from gunet_amd import _lib
    _lib.profile_enable(True)
    for i in range(steps):
        fn(i)
    ranks.sync()
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    return prof


def parity(got, want):
    """(normwise, elementwise) relative error: max|d| / max|want| and max(|d| / max(|want|, 1e-3 max|want|))."""
    import numpy as np
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    top = max(float(np.abs(want).max()), 1e-30)
    d = np.abs(got - want)
    return float(d.max() / top), float((d / np.maximum(np.abs(want), 1e-3 * top)).max())


def cpu_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get('num_threads', 1) for p in threadpool_info()] or [1]))
    except Exception:
        return int(os.environ.get('OMP_NUM_THREADS', '1'))


def oracle_module():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import gunet_oracle
    return gunet_oracle


def base_result(args, ranks, value, unit, elapsed, scaling='weak'):
    return {'metric': 'GraphUNet fwd samples/sec (495x436 grid)', 'value': value, 'unit': unit,
            'n_gpus': ranks.world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
            'dtype': {'f32': 'f32', 'split3': 'f32 (3xbf16-split operands, 6 bf16 MFMA products, f32 accumulate)',
                      'split2': 'f32 storage, 2xbf16-split matmul', 'bf16': 'bf16 matmul and intermediate storage, f32 accumulate'}[args.matmul],
            'data': 'synthetic'}


# ------------------------------------------------------------------------------------------------
# config 2 / 5: the whole model on one city

def make_model(args, dev, gunet_amd, synthetic):
    weights = synthetic.random_weights(synthetic.graph_unet_shapes_for(args.units, min_units=args.min_units), 1)
    model = gunet_amd.GraphUNet(args.units, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                                use_bias=True, use_global=True, output_activation=None, min_units=args.min_units)
    model.load_weights(weights, dev)
    return model, weights


def level_sizes(gunet_amd, node_loc_dev, quads):
    loc, edges = node_loc_dev, list(quads)
    n_levels, e_levels = [int(loc.shape[0])], [sum(int(q.shape[0]) for q in quads)]
    for _ in range(3):
        pp = gunet_amd.plan.get_pool_plan(loc, edges)
        loc, edges = pp.pooled_loc, pp.edge_index
        n_levels.append(pp.n_pooled)
        e_levels.append(sum(pp.n_pooled_edges))
    return n_levels, e_levels


def run_model(args, ranks):
    import numpy as np
    import torch
    world, rank, dev = ranks.world, ranks.rank, ranks.dev
    B = args.batch
    if args.stub:
        from gunet_amd import sharding
        ids = torch.arange(rank * B, (rank + 1) * B, dtype=torch.float32).reshape(B, 1, 1)

        def forward():
            return ids.expand(B, 7, 96).contiguous()
        y = forward()
        exchange = sharding.StepExchange(y, rotate=True)
        clock = PhaseClock(dev)

        def step(i):
            clock.start()
            y = forward()
            clock.mark()
            if world > 1:
                exchange.post(i, y, torch.tensor([float(y.sum()), float(B)], dtype=torch.float64))
            clock.mark()
            return y
        elapsed = timed_steps(step, exchange.drain, args, ranks)
        clock.marks = clock.marks[args.warmup:]
        phases = multi_rank_report(args, ranks, clock, exchange, lambda i: forward(), int(y.numel() * y.element_size()))
        seen = ranks.ranks_seen()
        if rank == 0:
            res = base_result(args, ranks, world * B * args.steps / elapsed, 'samples/s', elapsed)
            res.update({'stub': True, 'ranks_seen': seen, 'config': {'workload': 'stub forward (CPU rehearsal)'}})
            res.update(phases)
            last = args.steps - 1
            if world > 1 and exchange.collector[last & 1] == 0:
                res['gathered_sample_order'] = [int(v) for v in exchange.gathered[last & 1][:, :, 0, 0].reshape(-1).tolist()]
            print(json.dumps(res))
        return

    import gunet_amd
    from gunet_amd import synthetic, workload, sharding, _lib
    gunet_amd.set_matmul_mode(args.matmul)
    if args.dense:
        node_loc, edge_index, direction = synthetic.dense_graph(args.height, args.width)
    else:
        node_loc, edge_index, direction = synthetic.road_graph(args.height, args.width, args.density, 0)
    n_nodes = node_loc.shape[0]
    model, weights = make_model(args, dev, gunet_amd, synthetic)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    quads = gunet_amd.data_utils.sample_quadrant_subgraphs(torch.as_tensor(edge_index).to(dev),
                                                           torch.as_tensor(direction).to(dev))
    torch.cuda.synchronize()
    partition_ms = (time.perf_counter() - t0) * 1e3
    quads_np = [q.cpu().numpy() for q in quads]

    enc = synthetic.street_encoding(n_nodes, 5)
    xs, globs, firsts = [], [], []
    for b in range(B):
        sid = rank * B + b
        sn, te, wd = synthetic.sample_inputs(n_nodes, 1000 + sid, time_idx=12 * sid, weekday=sid)
        x, edges_np, glob = synthetic.assemble_model_inputs(sn, enc, quads_np, te, wd)
        if b in (0, B // 2, B - 1) and len(firsts) < 3:
            firsts.append((b, x, glob))
        xs.append(torch.as_tensor(x))
        globs.append(torch.as_tensor(glob))
    x_dev = torch.stack(xs).to(dev)                                   # [B, N, 104]
    glob_dev = torch.cat(globs, dim=0).to(dev)                        # [B, 113]
    # the street-map edge features depend on the city only: one [E_q, 16] tensor per direction serves the
    # whole batch (--expanded-edges hands the library B copies instead, [B, E_q, 16])
    edges_dev = [torch.as_tensor(e).to(dev) for e in edges_np]
    if args.expanded_edges:
        edges_dev = [e.unsqueeze(0).repeat(B, 1, 1).contiguous() for e in edges_dev]
    node_loc_dev = torch.as_tensor(node_loc).to(dev)
    del xs, globs

    def forward():
        graph = {'edge_index': list(quads), 'edges': list(edges_dev), 'node_loc': node_loc_dev,
                 'global': glob_dev}
        return model.predict(x_dev, graph)

    # the index plan of this city (all levels), built once and cached; timed on its own, twice: the very
    # first build also pays the one-off set-up of the library's kernels and of its temporary-block cache
    plan_ms = []
    for _ in range(2):
        gunet_amd.plan.clear_caches()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gunet_amd.plan.get_unet_plan(node_loc_dev, list(quads), 3)
        torch.cuda.synchronize()
        plan_ms.append((time.perf_counter() - t0) * 1e3)
    # first forward: weight table, workspace allocation, first launch of every kernel
    t0 = time.perf_counter()
    y = forward()
    torch.cuda.synchronize()
    first_call_ms = (time.perf_counter() - t0) * 1e3

    # what is gathered per step: the float32 predictions, or the uint8 prediction image
    horizons = torch.tensor([0, 1, 2, 5, 8, 11], dtype=torch.int32, device=dev)

    def payload(y):
        if args.gather == 'u8':
            img = torch.empty((B, 6, args.height, args.width, 8), dtype=torch.uint8, device=dev)
            _lib.call('gunet_prediction_image_u8', _lib.ptr(y), n_nodes, 96, B, _lib.ptr(node_loc_dev), args.height,
                      args.width, _lib.ptr(horizons), 6, 8, 255.0, _lib.ptr(img), _lib.current_stream(dev))
            return img
        return y

    # every step's payload is collected on one rank; the collector rotates (step i -> rank
    # i % world) so that no single GPU's links and memory take all of them
    exchange = sharding.StepExchange(payload(y), rotate=True) if args.gather != 'none' else None
    if world > 1 and exchange is not None:   # first use of every sender -> collector route, outside the timed steps
        for i in range(world):
            exchange.post(i, payload(y), torch.zeros(2, dtype=torch.float64, device=dev))
        exchange.drain()

    clock = PhaseClock(dev)

    def step(i):
        clock.start()
        y = forward()
        clock.mark()
        if world > 1 and exchange is not None:
            m = torch.stack((y.sum(dtype=torch.float64), torch.tensor(float(y.shape[0]), dtype=torch.float64, device=dev)))
            exchange.post(i, payload(y), m)
        clock.mark()
        return y

    drain = exchange.drain if exchange is not None else (lambda: None)
    elapsed = timed_steps(step, drain, args, ranks)
    clock.marks = clock.marks[args.warmup:]
    samples_per_s = world * B * args.steps / elapsed
    y = forward()
    phases = multi_rank_report(args, ranks, clock, exchange, lambda i: forward(),
                               0 if exchange is None else int(payload(y).numel() * payload(y).element_size()))
    seen = ranks.ranks_seen()

    prof = None if args.no_profile else profile_pass(lambda i: forward(), args.steps, ranks)

    # ---- the bf16-operand mode (float32 storage and accumulation), reported beside the result --------
    alt, alt_samples = None, {}
    if world == 1 and args.matmul == 'f32' and not args.no_alt_modes and args.min_units <= 256:
        gunet_amd.set_matmul_mode('bf16')
        try:
            for _ in range(2):
                y_alt = forward()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                y_alt = forward()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            alt = {'bf16': {'value': B * 10 / dt, 'unit': 'samples/s', 'ms_per_step': dt / 10 * 1e3, 'steps': 10,
                            'rel_diff_vs_f32_run': float((y_alt - y).abs().max() / y.abs().max()),
                            'note': 'same workload with gunet_set_matmul_mode(GUNET_MATMUL_BF16): every product on '
                                    'v_mfma_f32_32x32x16_bf16 (float32 accumulate), node features, node embeddings and '
                                    'messages stored in bf16 between the kernels, inputs and outputs float32; tolerance 2e-2 '
                                    'against the float32 CPU oracle'}}
            alt_samples = {b: y_alt[b].cpu().numpy() for b, _, _ in firsts}
        finally:
            gunet_amd.set_matmul_mode('f32')

    if rank != 0:
        return

    n_levels, e_levels = level_sizes(gunet_amd, node_loc_dev, quads)
    work = workload.work_per_sample(workload.layer_table(n_levels, e_levels, units=args.min_units)) \
        if args.units * 4 <= args.min_units else None
    result = base_result(args, ranks, samples_per_s, 'samples/s', elapsed)
    payload_bytes = 0 if world == 1 or exchange is None else int(payload(y).numel() * payload(y).element_size())
    result['ranks_seen'] = seen
    result.update(phases)
    result['config'] = {
        'workload': (f'full GraphUNet fwd, checkpoint shapes (12x8-ch in, 12x8-ch out, width {args.min_units}, depth 3), '
                     f'one {args.height}x{args.width} ' + ('dense grid' if args.dense else f'road graph (density {args.density}, seed 0)')
                     + f', batch {B} per GPU'),
        'nodes_per_level': n_levels, 'quadrant_edges_per_level': e_levels, 'batch_per_gpu': B,
        'parallelism': f'replicas x{world}, batch-sharded' + (
            f', RCCL gather of {args.gather} outputs ({payload_bytes / 1e6:.1f} MB per rank and step) to a rotating '
            'collector rank + metrics all-reduce per step' if world > 1 and exchange is not None else ''),
        'forward': 'one gunet_unet_forward_f32 call per step over a cached gunet_plan',
        'workspace_GB': (lambda w: None if w is None else w.numel() / 1e9)(
            gunet_amd.plan.model_workspace_of(dev)),
        'plan': 'cached', 'plan_build_ms': plan_ms[1], 'plan_build_ms_very_first': plan_ms[0],
        'first_forward_ms': first_call_ms, 'quadrant_partition_ms': partition_ms,
        # the reference redoes the index work inside every forward: the same step with the plan built each time
        'samples_per_s_with_plan_build_every_step': world * B / ((elapsed / args.steps) + plan_ms[1] * 1e-3),
    }
    if work is not None:
        result['config'].update({'gflop_per_sample': work['total_flops'] / 1e9,
                                 'executed_gflop_per_sample': work['total_flops_executed'] / 1e9,
                                 'compulsory_GB_per_sample': work['compulsory_bytes'] / 1e9,
                                 'staged_GB_per_sample': work['staged_bytes'] / 1e9})
        step_traffic = measured_step_traffic(args)
        if step_traffic is not None:
            # every feature kernel's measured HBM bytes per launch x its launches per forward, against the bytes a
            # fully fused layer would have to move (inputs read once, outputs written once)
            # (compulsory), and against the same layers as the passes their data dependencies force: embeddings and
            # messages through HBM once each (staged; gunet_amd/workload.py); both are float32 byte counts
            result['hbm_bytes_per_step_measured'] = step_traffic['bytes']
            if args.matmul == 'f32':
                result['traffic_over_compulsory'] = step_traffic['bytes'] / (B * work['compulsory_bytes'])
                result['traffic_over_staged'] = step_traffic['bytes'] / (B * work['staged_bytes'])
            result['hbm_bytes_per_step_stale'] = step_traffic['stale']
    if prof is not None:
        steps = args.steps
        result['kernels'] = kernel_table(prof, steps)
        result['launches_per_forward'] = sum(v[1] for v in prof.values()) / steps
        roof = roofline_from_profile(prof, args, workload)
        if roof is not None:
            result['roofline'] = roof
        gemm_ms = sum(prof[k][0] for k in GEMM_CLASSES)
        gemm_flops = sum(prof[k][2] for k in GEMM_CLASSES)
        edge_ms, edge_n, edge_flops = prof['edge_mlp']
        agg_ms, agg_n, agg_bytes = prof['aggregate']
        if gemm_ms > 0:
            result['gemm_all'] = {'ms_per_step': gemm_ms / steps, 'launches_per_step': sum(prof[k][1] for k in GEMM_CLASSES) / steps,
                                  'executed_tflops': gemm_flops / (gemm_ms * 1e-3) / 1e12}
            if work is not None:
                result['gemm_all']['as_written_tflops'] = B * work['gemm_flops'] * steps / (gemm_ms * 1e-3) / 1e12
        if edge_n:
            gs = {'edge_mlp_avg_launch_ms': edge_ms / edge_n, 'aggregate_launches_per_step': agg_n / steps}
            if work is not None:
                # NOT an HBM rate: the bytes a gather -> concat -> segment-sum pipeline would move, over the
                # time of the kernels that replace it (the messages are never re-read by a pass of their own)
                logical = B * (work['aggregate_bytes'] + work['gather_bytes']) * steps
                gs['logical_gather_scatter_GB_per_s_not_an_hbm_rate'] = logical / ((agg_ms + edge_ms) * 1e-3) / 1e9
            if agg_n:
                gs['aggregate_GBps'] = agg_bytes / (agg_ms * 1e-3) / 1e9
                gs['aggregate_frac_of_hbm_peak'] = gs['aggregate_GBps'] / workload.PEAK_HBM_GBPS
            edge_traffic = measured_traffic('k_edge_mlp', args)
            if edge_traffic is not None:
                gs['edge_mlp_measured_hbm_GBps'] = edge_traffic['bytes_per_launch'] / (edge_ms / edge_n * 1e-3) / 1e9
                gs['edge_mlp_frac_of_hbm_peak'] = gs['edge_mlp_measured_hbm_GBps'] / workload.PEAK_HBM_GBPS
                gs['edge_mlp_traffic_stale'] = edge_traffic['stale']
            pool_ms, pool_n, pool_bytes = prof['segment_reduce']
            if pool_n:
                gs['pooling_GBps'] = pool_bytes / (pool_ms * 1e-3) / 1e9
            result['gather_scatter'] = gs
        mfma_ms = gemm_ms + edge_ms
        if mfma_ms > 0:
            tf = (gemm_flops + edge_flops) / (mfma_ms * 1e-3) / 1e12
            result['mfma_utilisation'] = {'executed_tflops_over_gemm_and_edge_kernels': tf,
                                          'frac_of_f32_peak': tf / workload.PEAK_F32_MFMA_TFLOPS}
            if work is not None:
                result['mfma_utilisation']['as_written_tflops'] = B * work['total_flops'] * steps / (mfma_ms * 1e-3) / 1e12
    if alt is not None:
        result['alt_matmul_modes'] = alt
    if world == 1 and not args.no_cpu_baseline:
        O = oracle_module()
        times, errs, alt_errs = [], [], []
        for i in range(args.cpu_runs + 1):
            # every run times one single-sample forward (equal work); the samples rotate so that the
            # runs double as parity checks of the first, middle and last sample of the batch
            b, x, glob = firsts[i % len(firsts)]
            graph = {'edge_index': list(quads_np), 'edges': list(edges_np), 'node_loc': node_loc, 'global': glob}
            t0 = time.perf_counter()
            y_cpu, _ = O.predict(x, graph, weights)
            dt = time.perf_counter() - t0
            if i > 0:
                times.append(dt)
            if i < len(firsts):
                errs.append((b,) + parity(y[b].cpu().numpy(), y_cpu))
                if b in alt_samples:
                    alt_errs.append(parity(alt_samples[b], y_cpu)[0])
        med = float(np.median(times))
        result['cpu_baseline'] = {
            'value': 1.0 / med, 'unit': 'samples/s', 'cores': cpu_threads(), 'kind': 'port',
            'sample': f'{args.cpu_runs} timed single-sample forwards (median {med:.3f} s) after 1 warm-up, same graph and '
                      'weights as the GPU run; numpy oracle, unfused reference op order'}
        result['parity_rel_err_vs_cpu'] = max(e[1] for e in errs)
        result['parity_elementwise_rel_err_vs_cpu'] = max(e[2] for e in errs)
        result['parity_samples_checked'] = sorted({e[0] for e in errs})
        if alt_errs:
            result['alt_matmul_modes']['bf16']['parity_rel_err_vs_cpu'] = max(alt_errs)
            result['alt_matmul_modes']['bf16']['parity_samples_checked'] = sorted(alt_samples)
    print(json.dumps(result))


# ------------------------------------------------------------------------------------------------
# config 1: one GeoQuadrantGCN layer

def run_layer(args, ranks):
    import numpy as np
    import torch
    import gunet_amd
    from gunet_amd import synthetic, workload
    from gunet_amd.layers.hybrid_gnn import GeoQuadrantGCN
    dev = ranks.dev
    gunet_amd.set_matmul_mode(args.matmul)
    node_loc, edge_index, direction = synthetic.road_graph(args.height, args.width, args.density, 0)
    n = node_loc.shape[0]
    quads = gunet_amd.data_utils.sample_quadrant_subgraphs(torch.as_tensor(edge_index).to(dev),
                                                           torch.as_tensor(direction).to(dev))
    quads_np = [q.cpu().numpy() for q in quads]
    U, C, F, G, B = args.min_units, args.min_units, 16, 113, args.batch
    prefix = 'downsampling_layers/0/gcns/0'
    W = synthetic.random_weights(synthetic.gn_layer_shapes(prefix, C, F, G, U), 1)
    layer = GeoQuadrantGCN(U, activation='relu', use_bias=True, use_global=True)
    layer.load_weights(W, prefix, dev)
    rng = np.random.Generator(np.random.PCG64(2))
    x = rng.random((B, n, C), dtype=np.float32)
    ef = [rng.random((B, q.shape[0], F), dtype=np.float32) for q in quads_np]
    glob = rng.random((B, G), dtype=np.float32)
    x_d, ef_d, glob_d = torch.as_tensor(x).to(dev), [torch.as_tensor(e).to(dev) for e in ef], torch.as_tensor(glob).to(dev)
    node_loc_d = torch.as_tensor(node_loc).to(dev)
    out = {}

    def forward(i=0):
        graph = {'edge_index': list(quads), 'edges': list(ef_d), 'node_loc': node_loc_d, 'global': glob_d}
        out['y'], out['g'] = layer(x_d, graph)

    forward()
    elapsed = timed_steps(forward, lambda: None, args, ranks)
    prof = None if args.no_profile else profile_pass(forward, args.steps, ranks)
    e_total = sum(q.shape[0] for q in quads_np)
    u = U // 4
    flops = 2.0 * n * C * U + 2.0 * e_total * (F + 3 * u) * u + 2.0 * n * (C + U + G) * U
    compulsory = 4.0 * (n * C + e_total * F + 2 * e_total + e_total * u + n * U)
    result = base_result(args, ranks, ranks.world * B * args.steps / elapsed, 'samples/s', elapsed)
    result['metric'] = 'GeoQuadrantGCN layer fwd samples/sec (495x436 grid)'
    result['ranks_seen'] = ranks.ranks_seen()
    result['config'] = {'workload': f'one GeoQuadrantGCN layer (C_in {C}, F_e {F}, G {G}, U {U}), 4 quadrant subgraphs, one '
                                    f'{args.height}x{args.width} road graph (density {args.density}, seed 0), batch {B}',
                        'nodes': n, 'quadrant_edges': e_total, 'gflop_per_sample_as_written': flops / 1e9,
                        'compulsory_MB_per_sample': compulsory / 1e6,
                        'as_written_tflops': B * flops * args.steps / elapsed / 1e12,
                        'compulsory_GBps': B * compulsory * args.steps / elapsed / 1e9}
    if prof is not None:
        result['kernels'] = kernel_table(prof, args.steps)
        roof = roofline_from_profile(prof, args, workload)
        if roof is not None:
            result['roofline'] = roof
    if ranks.rank == 0 and not args.no_cpu_baseline:
        O = oracle_module()
        times = []
        for i in range(args.cpu_runs + 1):
            graph = {'edge_index': list(quads_np), 'edges': [e[0] for e in ef], 'node_loc': node_loc, 'global': glob[:1]}
            t0 = time.perf_counter()
            y_cpu, g_cpu = O.geo_quadrant_gcn(x[0], graph, W, prefix)
            if i > 0:
                times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        result['cpu_baseline'] = {'value': 1.0 / med, 'unit': 'samples/s', 'cores': cpu_threads(), 'kind': 'port',
                                  'sample': f'{args.cpu_runs} timed single-layer forwards (median {med:.3f} s) after 1 warm-up, '
                                            'same graph, weights and sample; numpy oracle, unfused reference op order'}
        yg = out['y'][0].cpu().numpy()
        errs = [parity(yg, y_cpu), parity(out['g']['global'][0].cpu().numpy(), g_cpu['global'][0])] + \
               [parity(out['g']['edges'][q][0].cpu().numpy(), g_cpu['edges'][q]) for q in range(4)]
        result['parity_rel_err_vs_cpu'] = max(e[0] for e in errs)
        result['parity_elementwise_rel_err_vs_cpu'] = max(e[1] for e in errs)
    if ranks.rank == 0:
        print(json.dumps(result))


# ------------------------------------------------------------------------------------------------
# config 3: the pooling chain alone

def run_pool_chain(args, ranks):
    import ctypes as C
    import numpy as np
    import torch
    import gunet_amd
    from gunet_amd import synthetic, workload, plan as _plan, _lib
    dev = ranks.dev
    B, U = args.batch, args.min_units
    u = U // 4
    if args.dense:
        node_loc, edge_index, direction = synthetic.dense_graph(args.height, args.width)
    else:
        node_loc, edge_index, direction = synthetic.road_graph(args.height, args.width, args.density, 0)
    ei_d, dir_d = torch.as_tensor(edge_index).to(dev), torch.as_tensor(direction).to(dev)
    node_loc_d = torch.as_tensor(node_loc).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    quads = gunet_amd.data_utils.sample_quadrant_subgraphs(ei_d, dir_d)
    up = _plan.UNetPlan(node_loc_d, quads, 3)
    torch.cuda.synchronize()
    plan_ms = (time.perf_counter() - t0) * 1e3
    gen = torch.Generator(device=dev).manual_seed(3)
    x = [torch.rand((B, up.n_nodes[0], U), generator=gen, device=dev)]
    e = [[torch.rand((B, up.n_edges[0][q], u), generator=gen, device=dev) for q in range(4)]]
    for l in range(1, 4):
        x.append(torch.empty((B, up.n_nodes[l], U), device=dev))
        e.append([torch.empty((B, up.n_edges[l][q], u), device=dev) for q in range(4)])

    def chain(i=0):
        for l in range(3):
            src, dst = _lib.ptr4(e[l]), _lib.ptr4(e[l + 1])
            _lib.call('gunet_plan_pool_features_f32', up.handle, l, B, _lib.ptr(x[l]), U, C.byref(src), u,
                      _lib.ptr(x[l + 1]), C.byref(dst), _lib.current_stream(dev))

    chain()
    elapsed = timed_steps(chain, lambda: None, args, ranks)
    prof = None if args.no_profile else profile_pass(chain, args.steps, ranks)
    bytes_per_sample = 0.0
    for l in range(3):
        n, p = up.n_nodes[l], up.n_nodes[l + 1]
        bytes_per_sample += 4.0 * (n * U + p * U) + 4.0 * (n + p)
        for q in range(4):
            bytes_per_sample += 4.0 * (up.n_edges[l][q] * u + up.n_edges[l + 1][q] * u) + 4.0 * (up.n_edges[l][q] + up.n_edges[l + 1][q])
    result = base_result(args, ranks, ranks.world * B * args.steps / elapsed, 'samples/s', elapsed)
    result['metric'] = 'graph 2x2 pooling chain samples/sec (495x436 grid) + gather-scatter GB/s'
    result['ranks_seen'] = ranks.ranks_seen()
    result['config'] = {'workload': f'2x2 pooling chain levels 0->1->2->3 alone (segment max of node features [B,N,{U}] and 4 '
                                    f'edge-feature sets [B,E_q,{u}] per level, one launch per level), one {args.height}x{args.width} '
                                    + ('dense grid' if args.dense else f'road graph (density {args.density}, seed 0)') + f', batch {B}',
                        'nodes_per_level': up.n_nodes, 'quadrant_edges_per_level': [sum(v) for v in up.n_edges],
                        'index_chain_ms': plan_ms,
                        'index_chain': 'quadrant partition + gunet_plan_create: cluster ids, pooled positions, pooled edge lists, '
                                       'dup maps, CSRs and the upsampling edge partition of levels 3->2->1->0 (all bit-exact, '
                                       'built once per city); the way up has no unparameterised feature kernel -- the zero rows '
                                       'and zero edge features of the reference are never formed',
                        'algorithmic_MB_per_sample': bytes_per_sample / 1e6,
                        'algorithmic_GBps': B * bytes_per_sample * args.steps / elapsed / 1e9}
    result['config']['frac_of_hbm_peak'] = result['config']['algorithmic_GBps'] / workload.PEAK_HBM_GBPS
    if prof is not None:
        result['kernels'] = kernel_table(prof, args.steps)
        roof = roofline_from_profile(prof, args, workload)
        if roof is not None:
            result['roofline'] = roof
    if ranks.rank == 0 and not args.no_cpu_baseline:
        O = oracle_module()
        quads_np = [q.cpu().numpy() for q in quads]
        xs = x[0][0].cpu().numpy()
        es = [t[0].cpu().numpy() for t in e[0]]
        times, ok = [], True
        for i in range(3):
            graph = {'edge_index': list(quads_np), 'edges': list(es), 'node_loc': node_loc, 'global': None}
            xc = xs
            t0 = time.perf_counter()
            for l in range(3):
                xc, graph = O.graph_pooling_2x2(xc, graph)
                if i == 0:   # maxima are exact: bit-identical per level
                    ok = ok and np.array_equal(x[l + 1][0].cpu().numpy(), xc)
                    ok = ok and all(np.array_equal(e[l + 1][q][0].cpu().numpy(), graph['edges'][q]) for q in range(4))
            if i > 0:
                times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        result['cpu_baseline'] = {'value': 1.0 / med, 'unit': 'samples/s', 'cores': cpu_threads(), 'kind': 'port',
                                  'sample': f'2 timed single-sample chains (median {med:.3f} s) after 1 warm-up, index work '
                                            '(tf.unique, masks) included as the reference recomputes it per call; numpy oracle'}
        result['parity_bit_exact_vs_cpu'] = bool(ok)
    if ranks.rank == 0:
        print(json.dumps(result))


# ------------------------------------------------------------------------------------------------
# config 4: six cities, the sample list sharded over the ranks

def run_cities(args, ranks):
    import numpy as np
    import torch
    from gunet_amd import sharding
    world, rank, dev = ranks.world, ranks.rank, ranks.dev
    n_cities = len(CITY_SEEDS)
    sample_cities = [i % n_cities for i in range(args.samples)]
    mine = sharding.shard_by_city(sample_cities, world, rank)
    groups = sharding.city_groups(mine, sample_cities)
    stub_nodes = (11, 13, 17, 19, 23, 29)

    city = {}
    if not args.stub:
        import gunet_amd
        from gunet_amd import synthetic, workload, evaluation
        gunet_amd.set_matmul_mode(args.matmul)
        model, weights = make_model(args, dev, gunet_amd, synthetic)
    t0 = time.perf_counter()
    for c, ids in groups:
        if args.stub:
            city[c] = {'n': stub_nodes[c], 'ids': ids}
            continue
        node_loc, edge_index, direction = synthetic.road_graph(args.height, args.width, CITY_DENSITIES[c], CITY_SEEDS[c])
        n = node_loc.shape[0]
        quads = gunet_amd.data_utils.sample_quadrant_subgraphs(torch.as_tensor(edge_index).to(dev),
                                                               torch.as_tensor(direction).to(dev))
        quads_np = [q.cpu().numpy() for q in quads]
        enc = synthetic.street_encoding(n, 5 + c)
        xs, globs, first = [], [], None
        for sid in ids:
            sn, te, wd = synthetic.sample_inputs(n, 1000 + sid, time_idx=12 * sid, weekday=sid)
            x, edges_np, glob = synthetic.assemble_model_inputs(sn, enc, quads_np, te, wd)
            if first is None:
                first = (x, edges_np, glob)
            xs.append(torch.as_tensor(x))
            globs.append(torch.as_tensor(glob))
        b = len(ids)
        tg = torch.Generator(device=dev).manual_seed(77 + c)
        city[c] = {'n': n, 'ids': ids, 'quads': quads, 'quads_np': quads_np, 'node_loc_np': node_loc,
                   'node_loc': torch.as_tensor(node_loc).to(dev), 'x': torch.stack(xs).to(dev),
                   'glob': torch.cat(globs, dim=0).to(dev),
                   'edges': [torch.as_tensor(e).to(dev) for e in edges_np],      # [E_q, 16], shared by the city's samples
                   'target': torch.rand((b, n, 96), generator=tg, device=dev), 'first': first}
    setup_s = time.perf_counter() - t0

    # this rank's predictions: one [rows, 96] buffer, the samples in sharding order
    rows = sum(city[c]['n'] * len(ids) for c, ids in groups)
    flat = torch.empty((rows, 96), dtype=torch.float32, device=dev)
    off = 0
    for c, ids in groups:
        k = city[c]['n'] * len(ids)
        city[c]['out'] = flat[off:off + k].view(len(ids), city[c]['n'], 96)
        off += k
    counts = sharding.row_counts(rows, dev)
    gathered = [torch.empty((sum(counts), 96), dtype=torch.float32, device=dev) if world > 1 else None]
    metrics = torch.zeros(3, dtype=torch.float64, device=dev)

    def forward_all():
        metrics.zero_()
        for c, ids in groups:
            d = city[c]
            if args.stub:
                d['out'].copy_(torch.tensor(ids, dtype=torch.float32).reshape(-1, 1, 1).expand_as(d['out']))
                metrics[2] += len(ids)
                continue
            graph = {'edge_index': list(d['quads']), 'edges': list(d['edges']), 'node_loc': d['node_loc'], 'global': d['glob']}
            model.predict(d['x'], graph, out=d['out'])
            sums = evaluation.error_sums(d['target'], d['out'])           # float64 [b, 2, 96]
            metrics[:2] += sums.sum(dim=(0, 2))
            metrics[2] += float(len(ids) * d['n'] * 96)

    clock = PhaseClock(dev)

    def step(i):
        clock.start()
        forward_all()
        clock.mark()
        if world > 1:
            if args.gather != 'none':
                sharding.gather_rows(flat, dst=i % world, counts=counts, out=gathered[0])
            ranks.dist.all_reduce(metrics)
        clock.mark()

    t0 = time.perf_counter()
    forward_all()
    ranks.sync()
    first_pass_s = time.perf_counter() - t0
    if world > 1 and args.gather != 'none':   # first use of every sender -> collector route, outside the timed steps
        for dst in range(world):
            sharding.gather_rows(flat, dst=dst, counts=counts, out=gathered[0])
    elapsed = timed_steps(step, lambda: None, args, ranks)
    clock.marks = clock.marks[args.warmup:]
    phases = multi_rank_report(args, ranks, clock, True if args.gather != 'none' else None, lambda i: forward_all(),
                               int(flat.numel() * flat.element_size()))

    # one untimed gather to rank 0, read back: the order in which the samples arrive
    forward_all()
    final = sharding.gather_rows(flat, dst=0, counts=counts, out=gathered[0]) if world > 1 else flat
    all_mine = [None] * world
    if world > 1:
        ranks.dist.all_gather_object(all_mine, mine)
        ranks.dist.all_reduce(metrics)
    else:
        all_mine = [mine]
    seen = ranks.ranks_seen()
    prof = None
    if not args.stub and not args.no_profile:
        prof = profile_pass(lambda i: forward_all(), max(1, args.steps // 4), ranks)
    if rank != 0:
        return
    order = [i for part in all_mine for i in part]
    n_of = (lambda c: stub_nodes[c]) if args.stub else None
    result = base_result(args, ranks, args.samples * args.steps / elapsed, 'samples/s', elapsed, scaling='strong')
    result['ranks_seen'] = seen
    result.update(phases)
    result['samples_per_rank'] = [len(part) for part in all_mine]
    result['cities_per_rank'] = [sorted({sample_cities[i] for i in part}) for part in all_mine]
    result['gathered_rows'] = int(final.shape[0])
    if args.stub:
        got, off = [], 0
        for i in order:
            got.append(int(final[off, 0]))
            off += n_of(sample_cities[i])
        result.update({'stub': True, 'gathered_sample_order': got, 'expected_sample_order': order,
                       'metric_count': float(metrics[2]), 'config': {'workload': 'stub forward (CPU rehearsal)'}})
        print(json.dumps(result))
        return
    result['config'] = {
        'workload': (f'six-city GraphUNet fwd: {args.height}x{args.width} road graphs, seeds {list(CITY_SEEDS)}, densities '
                     f'{list(CITY_DENSITIES)}; {args.samples} (city, time-window) samples round-robin over the cities, sharded '
                     f'contiguous-by-city over {world} GPU(s); checkpoint shapes, width {args.min_units}'),
        'parallelism': f'replicas x{world}, sample list sharded by city' + (
            f'; per step RCCL point-to-point gather of the [rows, 96] float32 predictions ({sum(counts) * 384 / 1e6:.0f} MB in '
            'all) to a rotating collector + all-reduce of the error sums' if world > 1 else ''),
        'forward': 'one gunet_unet_forward_f32 call per city group and step, writing into the gather buffer',
        'setup_s_rank0': setup_s, 'first_pass_s_rank0': first_pass_s,
        'mse_all_samples': float(metrics[0] / metrics[2]), 'mae_all_samples': float(metrics[1] / metrics[2])}
    if prof is not None:
        steps = max(1, args.steps // 4)
        result['kernels'] = kernel_table(prof, steps)
        roof = roofline_from_profile(prof, args, workload)
        if roof is not None:
            result['roofline'] = roof
    if world == 1 and not args.no_cpu_baseline:
        O = oracle_module()
        c0, ids0 = groups[0]
        d = city[c0]
        x, edges_np, glob = d['first']
        times = []
        for i in range(3):
            graph = {'edge_index': list(d['quads_np']), 'edges': list(edges_np), 'node_loc': d['node_loc_np'], 'global': glob}
            t0 = time.perf_counter()
            y_cpu, _ = O.predict(x, graph, weights)
            if i > 0:
                times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        result['cpu_baseline'] = {'value': 1.0 / med, 'unit': 'samples/s', 'cores': cpu_threads(), 'kind': 'port',
                                  'sample': f'2 timed forwards of sample {ids0[0]} (city {c0}, {d["n"]} nodes; median {med:.3f} s) '
                                            'after 1 warm-up; numpy oracle, unfused reference op order'}
        e = parity(d['out'][0].cpu().numpy(), y_cpu)
        result['parity_rel_err_vs_cpu'], result['parity_elementwise_rel_err_vs_cpu'] = e
    print(json.dumps(result))


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))
    ranks = Ranks(args)
    try:
        if args.config in (2, 5):
            run_model(args, ranks)
        elif args.config == 1:
            run_layer(args, ranks)
        elif args.config == 3:
            run_pool_chain(args, ranks)
        else:
            run_cities(args, ranks)
    finally:
        ranks.close()


if __name__ == '__main__':
    main()
