#!/usr/bin/env python3
"""Benchmark of the Graph U-Net forward hot path on MI355X.

Workload (N = 1): full GraphUNet forward at the shipped checkpoint's shapes (12x8-channel input,
12x8-channel output, width 128, depth 3, 14 GeoQuadrantGCN layers) on one synthetic 495x436 road
graph (mask density 0.45, seed 0), a batch of 8 independent (city, time-window) samples, float32.
One "step" = one forward of the batch with inputs resident in HBM and the per-city index plan
cached (the plan build is timed separately and reported in `config`).

With --gpus N > 1 every rank runs its own batch (weak scaling, no collective inside the forward);
per step the predictions are gathered on one rank (the collector rotates) and a small metrics vector
is all-reduced over RCCL, overlapped with the next step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8, help='samples per GPU per step')
    ap.add_argument('--height', type=int, default=495)
    ap.add_argument('--width', type=int, default=436)
    ap.add_argument('--density', type=float, default=0.45)
    ap.add_argument('--dense', action='store_true', help='all pixels are nodes (MFMA stress graph)')
    ap.add_argument('--min-units', type=int, default=128, choices=[128, 256],
                    help='width of every block (128 = the shipped checkpoint; 256 = the wide stress variant)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-runs', type=int, default=7, help='timed CPU forwards (about 1.5 s each)')
    ap.add_argument('--no-profile', action='store_true', help='skip the per-kernel event pass')
    ap.add_argument('--no-alt-modes', action='store_true', help='skip the short bf16-operand run reported beside the result')
    ap.add_argument('--matmul', default=os.environ.get('GUNET_BENCH_MATMUL', 'f32'), choices=['f32', 'split3', 'split2', 'bf16'],
                    help='matrix-product arithmetic (see gunet_set_matmul_mode)')
    return ap.parse_args()


GEMM_CLASSES = ('gemm_tiled', 'gemm_resident_k128', 'gemm_resident_k256', 'gemm_resident_k384', 'gemm_resident_bf16',
                'gemm_split')
MFMA_CLASSES = GEMM_CLASSES + ('edge_mlp',)
KERNEL_OF_CLASS = {'gemm_tiled': 'k_gemm_f32', 'gemm_resident_k128': 'k_gemm_resident<*, 4>',
                   'gemm_resident_k256': 'k_gemm_resident<*, 8>', 'gemm_resident_k384': 'k_gemm_resident<*, 12>',
                   'gemm_resident_bf16': 'k_gemm_resident<.., bf16>', 'gemm_split': 'k_gemm_split', 'edge_mlp': 'k_edge_mlp',
                   'aggregate': 'k_aggregate', 'segment_reduce': 'k_segment_reduce', 'layer_prologue': 'k_layer_prologue',
                   'global_update': 'k_global_update', 'global_bridge': 'k_global_bridge', 'gather_rows': 'k_gather_rows', 'dense_naive': 'k_dense_naive'}


def measured_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes of this same
    command (profiles/hbm_traffic_per_launch.json: FETCH_SIZE doubled as gfx950 needs, plus
    WRITE_SIZE); None for a workload other than the default one or a kernel without an entry."""
    default = (args.height, args.width, args.density, args.dense, args.min_units, args.batch, args.matmul) == \
              (495, 436, 0.45, False, 128, 8, 'f32')
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic_per_launch.json')
    if not default or not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    entry = table.get('kernels', {}).get(kernel)
    if entry is None:
        return None
    return {'bytes_per_launch': entry['hbm_MB_per_launch'] * 1e6, 'source': table.get('source', path)}


def cpu_baseline(args, graph_np, weights, inputs_np):
    """Time the CPU restatement (oracle/, numpy) of the same forward on one sample."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import numpy as np
    import gunet_oracle as O
    node_loc, quads = graph_np
    x, edges, glob = inputs_np
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = int(os.environ.get('OMP_NUM_THREADS', '1'))
    times = []
    for i in range(args.cpu_runs + 1):
        graph = {'edge_index': list(quads), 'edges': list(edges), 'node_loc': node_loc, 'global': glob}
        t0 = time.perf_counter()
        y, _ = O.predict(x, graph, weights)
        dt = time.perf_counter() - t0
        if i > 0:
            times.append(dt)
    med = float(np.median(times))
    return {'value': 1.0 / med, 'unit': 'samples/s', 'cores': int(cores), 'kind': 'port',
            'sample': f'{args.cpu_runs} timed single-sample forwards (median {med:.3f} s) after 1 warm-up, '
                      'same graph, weights and first sample as the GPU run; numpy oracle, unfused reference op order'}, y


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', torch.cuda.current_device())

    import gunet_amd
    from gunet_amd import synthetic, workload, sharding, _lib
    gunet_amd.set_matmul_mode(args.matmul)

    # ---- workload -----------------------------------------------------------------------
    if args.dense:
        node_loc, edge_index, direction = synthetic.dense_graph(args.height, args.width)
    else:
        node_loc, edge_index, direction = synthetic.road_graph(args.height, args.width, args.density, 0)
    n_nodes = node_loc.shape[0]
    weights = synthetic.random_weights(synthetic.graph_unet_shapes_for(32, min_units=args.min_units), 1)
    model = gunet_amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                                use_bias=True, use_global=True, output_activation=None, min_units=args.min_units)
    model.load_weights(weights, dev)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    quads = gunet_amd.data_utils.sample_quadrant_subgraphs(torch.as_tensor(edge_index).to(dev),
                                                           torch.as_tensor(direction).to(dev))
    torch.cuda.synchronize()
    partition_ms = (time.perf_counter() - t0) * 1e3
    quads_np = [q.cpu().numpy() for q in quads]

    enc = synthetic.street_encoding(n_nodes, 5)
    B = args.batch
    xs, globs, first = [], [], None
    for b in range(B):
        sid = rank * B + b
        sn, te, wd = synthetic.sample_inputs(n_nodes, 1000 + sid, time_idx=12 * sid, weekday=sid)
        x, edges_np, glob = synthetic.assemble_model_inputs(sn, enc, quads_np, te, wd)
        if first is None:
            first = (x, edges_np, glob)
        xs.append(torch.as_tensor(x))
        globs.append(torch.as_tensor(glob))
    x_dev = torch.stack(xs).to(dev)                                   # [B, N, 104]
    glob_dev = torch.cat(globs, dim=0).to(dev)                        # [B, 113]
    edges_dev = [torch.as_tensor(e).to(dev).unsqueeze(0).repeat(B, 1, 1).contiguous() for e in first[1]]
    node_loc_dev = torch.as_tensor(node_loc).to(dev)
    del xs, globs

    def forward():
        graph = {'edge_index': list(quads), 'edges': list(edges_dev), 'node_loc': node_loc_dev,
                 'global': glob_dev}
        return model.predict(x_dev, graph)

    # first call builds (and caches) every plan of this city
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = forward()
    torch.cuda.synchronize()
    first_call_ms = (time.perf_counter() - t0) * 1e3

    # every step's predictions are collected on one rank; the collector rotates (step i -> rank
    # i % world) so that no single GPU's links and memory take all of them
    exchange = sharding.StepExchange(y, rotate=True)
    if world > 1:   # first use of every sender -> collector route, outside the warm-up and the timed steps
        for i in range(world):
            exchange.post(i, y, torch.zeros(2, dtype=torch.float64, device=dev))
        exchange.drain()

    def step(i):
        y = forward()
        if world > 1:
            m = torch.stack((y.sum(dtype=torch.float64), torch.tensor(float(y.shape[0]), dtype=torch.float64, device=dev)))
            exchange.post(i, y, m)
        return y

    drain = exchange.drain

    for i in range(args.warmup):
        y = step(i)
    drain()

    # ---- timed region -------------------------------------------------------------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        y = step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    samples_per_s = world * B * args.steps / elapsed

    # ---- per-kernel-class pass: HIP events around every launch, same steps again ------------
    prof = None
    if not args.no_profile:
        _lib.profile_enable(True)
        for i in range(args.steps):
            forward()
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(False)

    # ---- the bf16-operand mode (float32 storage and accumulation), reported beside the result --------
    alt = None
    if world == 1 and args.matmul == 'f32' and not args.no_alt_modes:
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
                            'note': 'same workload with gunet_set_matmul_mode(GUNET_MATMUL_BF16): GEMM operands rounded to '
                                    'bf16 for v_mfma_f32_32x32x16_bf16, float32 accumulate and storage; tolerance 2e-2'}}
        finally:
            gunet_amd.set_matmul_mode('f32')

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- algorithmic work and roofline ----------------------------------------------------
    loc, edges = node_loc_dev, list(quads)
    n_levels, e_levels = [n_nodes], [sum(int(q.shape[0]) for q in quads)]
    for _ in range(3):
        pp = gunet_amd.plan.get_pool_plan(loc, edges)
        loc, edges = pp.pooled_loc, pp.edge_index
        n_levels.append(pp.n_pooled)
        e_levels.append(sum(pp.n_pooled_edges))
    work = workload.work_per_sample(workload.layer_table(n_levels, e_levels, units=args.min_units))
    result = {
        'metric': 'GraphUNet fwd samples/sec (495x436 grid)', 'value': samples_per_s, 'unit': 'samples/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'f32': 'f32', 'split3': 'f32 (3xbf16-split operands, 6 bf16 MFMA products, f32 accumulate)',
                  'split2': 'f32 storage, 2xbf16-split matmul', 'bf16': 'bf16 matmul, f32 storage/accumulate'}[args.matmul],
        'data': 'synthetic',
        'config': {
            'workload': (f'full GraphUNet fwd, checkpoint shapes (12x8-ch in, 12x8-ch out, width {args.min_units}, depth 3), '
                         f'one {args.height}x{args.width} ' + ('dense grid' if args.dense else f'road graph (density {args.density}, seed 0)')
                         + f', batch {B} per GPU'),
            'nodes_per_level': n_levels, 'quadrant_edges_per_level': e_levels, 'batch_per_gpu': B,
            'parallelism': f'replicas x{world}, batch-sharded' + (', RCCL gather of outputs to a rotating collector rank + metrics all-reduce per step' if world > 1 else ''),
            'forward': 'one gunet_unet_forward_f32 call per step over a cached gunet_plan',
            'workspace_GB': (lambda w: None if w is None else w.numel() / 1e9)(
                gunet_amd.plan._model_workspaces.get((dev.type, dev.index))),
            'plan': 'cached', 'plan_build_ms_first_call': first_call_ms, 'quadrant_partition_ms': partition_ms,
            'gflop_per_sample': work['total_flops'] / 1e9, 'executed_gflop_per_sample': work['total_flops_executed'] / 1e9,
            'compulsory_GB_per_sample': work['compulsory_bytes'] / 1e9,
        },
    }
    if prof is not None:
        steps = args.steps
        result['kernels'] = {k: {'ms_per_step': v[0] / steps, 'launches_per_step': v[1] / steps,
                                 ('gflop_per_step' if k in MFMA_CLASSES else 'MB_per_step'):
                                     v[2] / steps / (1e9 if k in MFMA_CLASSES else 1e6)}
                             for k, v in prof.items() if v[1]}
        # dominant kernel = the class (one kernel template each) with the most device time; its work
        # is what the library counted for exactly those launches
        dom = max(prof, key=lambda k: prof[k][0])
        ms, n, work_done = prof[dom]
        products = {'f32': 1, 'split3': 6, 'split2': 3, 'bf16': 1}[args.matmul] if dom == 'gemm_split' else 1
        if dom in MFMA_CLASSES:
            peak = workload.PEAK_BF16_MFMA_TFLOPS if dom == 'gemm_split' else workload.PEAK_F32_MFMA_TFLOPS
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
        result['roofline'] = roof
        gemm_ms = sum(prof[k][0] for k in GEMM_CLASSES)
        gemm_flops = sum(prof[k][2] for k in GEMM_CLASSES)
        edge_ms, edge_n, edge_flops = prof['edge_mlp']
        agg_ms, agg_n, agg_bytes = prof['aggregate']
        if gemm_ms > 0:
            result['gemm_all'] = {'ms_per_step': gemm_ms / steps, 'launches_per_step': sum(prof[k][1] for k in GEMM_CLASSES) / steps,
                                  'executed_tflops': gemm_flops / (gemm_ms * 1e-3) / 1e12,
                                  'as_written_tflops': B * work['gemm_flops'] * steps / (gemm_ms * 1e-3) / 1e12}
        if edge_n:
            # gathers 4(2Eu+2E) happen inside the edge-MLP kernel; the segment sum 4(Eu+E+NU) is either
            # k_aggregate or, for layers whose node MLP sums the messages while loading them
            # (k_gemm_resident<.., true>), costs no pass of its own
            logical = B * (work['aggregate_bytes'] + work['gather_bytes']) * steps
            gs = {'gather_plus_segment_sum_logical_GBps': logical / ((agg_ms + edge_ms) * 1e-3) / 1e9,
                  'edge_mlp_avg_launch_ms': edge_ms / edge_n, 'aggregate_launches_per_step': agg_n / steps,
                  'note': 'logical bytes of all 14 layers (gathers 4(2Eu+2E) + segment sum 4(Eu+E+NU)) over the time of '
                          'the kernels that do nothing else (edge MLP incl. its matrix work, k_aggregate where it still '
                          'runs); the segment sum of the other layers is part of the node-MLP GEMM'}
            if agg_n:
                gs['aggregate_GBps'] = agg_bytes / (agg_ms * 1e-3) / 1e9
            edge_traffic = measured_traffic('k_edge_mlp', args)
            if edge_traffic is not None:
                gs['edge_mlp_measured_hbm_GBps'] = edge_traffic['bytes_per_launch'] / (edge_ms / edge_n * 1e-3) / 1e9
                gs['edge_mlp_frac_of_hbm_peak'] = gs['edge_mlp_measured_hbm_GBps'] / workload.PEAK_HBM_GBPS
            result['gather_scatter'] = gs
        mfma_ms = gemm_ms + edge_ms
        if mfma_ms > 0:
            tf = (gemm_flops + edge_flops) / (mfma_ms * 1e-3) / 1e12
            result['mfma_utilisation'] = {'executed_tflops_over_gemm_and_edge_kernels': tf,
                                          'frac_of_f32_peak': tf / workload.PEAK_F32_MFMA_TFLOPS,
                                          'as_written_tflops': B * work['total_flops'] * steps / (mfma_ms * 1e-3) / 1e12}
    if alt is not None:
        result['alt_matmul_modes'] = alt
    if world == 1 and not args.no_cpu_baseline:
        base, y_cpu = cpu_baseline(args, (node_loc, quads_np), weights, first)
        result['cpu_baseline'] = base
        got = y[0].cpu().numpy()
        result['parity_rel_err_vs_cpu'] = float(np.abs(got - y_cpu).max() / np.abs(y_cpu).max())
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
