"""Algorithmic work of one Graph U-Net forward: FLOPs and bytes per kernel class, from the graph's
level sizes.  These are the figures the benchmark's roofline fractions are computed from.

Per GeoQuadrantGCN layer on N nodes, E = sum of the four quadrant edge counts, input width C,
edge-feature width F, global width G, U units (128 or 256), u = U / 4 (float32 = 4 bytes):
  GEMM FLOPs      2*N*C*U (4 node embeddings)  +  2*N*(C + U + G)*U (node MLP, as the reference
                  writes it: the tiled global columns are counted although the kernels fold them
                  into a per-sample bias)
  edge FLOPs      2*E*(F + 3u)*u   (as written; the kernel folds the u global columns into a bias)
  aggregate bytes 4*(E*u + E + N*U)   read messages + read slot->edge map + write aggregates
  gather bytes    4*(2*E*u + 2*E)     the two row gathers feeding the edge MLP
  compulsory      4*(N*C + E*F + 2*E + E*u + N*U)   a layer as one pass: inputs read once, outputs written once
  staged          4*(2*N*C + 3*N*U + E*F + 2*E + 2*E*u + 12*N)   the same layer as the three passes its
                  data dependencies force (every message needs the embeddings of two arbitrary nodes, every
                  node update all of its incoming messages): embeddings written and read once, x read by
                  both GEMMs, messages written and read once, the in-edge table read once
The "*_executed" totals count what the kernels issue after the algebraic savings (global columns
folded into biases; the upsampling layer computed without its zero rows); roofline fractions of
the MFMA kernels use these, so they can never exceed 1, and the as-written totals are reported
beside them.
"""

def layer_table(n_nodes, n_edges, f_e0=16, g0=113, in_features=104, out_units=96, units=128):
    """n_nodes[l], n_edges[l] (sum over quadrants) for levels 0..depth -> list of layer dicts in
    execution order (GN layers and the two dense layers) of a U-Net that is `units` wide
    throughout (128 = the shipped checkpoint)."""
    U, SMALL_U = units, units // 4
    depth = len(n_nodes) - 1
    rows = [dict(name='in_layer', kind='dense', n=n_nodes[0], k=in_features, out=U)]
    f_e, g = f_e0, g0
    skip_fe, skip_g = [], []
    for d in range(depth):
        skip_fe.append(f_e)
        skip_g.append(g)
        rows.append(dict(name=f'down{d}.gcn', kind='gn', n=n_nodes[d], e=n_edges[d], c=U, f=f_e, g=g))
        rows.append(dict(name=f'down{d}.pool', kind='pool', n=n_nodes[d], p=n_nodes[d + 1], e=n_edges[d],
                         e2=n_edges[d + 1]))
        f_e, g = SMALL_U, U
    rows.append(dict(name='bottleneck1', kind='gn', n=n_nodes[depth], e=n_edges[depth], c=U, f=f_e, g=g))
    rows.append(dict(name='bottleneck2', kind='gn', n=n_nodes[depth], e=n_edges[depth], c=U, f=SMALL_U, g=U))
    for d in range(depth):
        lvl = depth - 1 - d
        rows.append(dict(name=f'up{d}.upsample', kind='gn', n=n_nodes[lvl + 1] + n_nodes[lvl], e=n_nodes[lvl],
                         c=U, f=2, g=U + skip_g[lvl], n_coarse=n_nodes[lvl + 1], n_fine=n_nodes[lvl]))
        rows.append(dict(name=f'up{d}.gcn0', kind='gn', n=n_nodes[lvl], e=n_edges[lvl], c=2 * U, f=skip_fe[lvl], g=U))
        rows.append(dict(name=f'up{d}.gcn1', kind='gn', n=n_nodes[lvl], e=n_edges[lvl], c=U, f=SMALL_U, g=U))
    rows.append(dict(name='out_model', kind='dense', n=n_nodes[0], k=U, out=out_units))
    for r in rows:
        r['U'] = U
    return rows


def work_per_sample(rows):
    """Totals for one sample: FLOPs per MFMA kernel class and bytes per HBM-bound class."""
    w = dict(gemm_flops=0.0, edge_flops=0.0, aggregate_bytes=0.0, gather_bytes=0.0, compulsory_bytes=0.0,
             pool_bytes=0.0, staged_bytes=0.0, gemm_launches=0, edge_launches=0, aggregate_launches=0,
             gemm_flops_executed=0.0, edge_flops_executed=0.0, aggregate_bytes_executed=0.0)
    for r in rows:
        U, SMALL_U = r['U'], r['U'] // 4
        if r['kind'] == 'dense':
            w['gemm_flops'] += 2.0 * r['n'] * r['k'] * r['out']
            w['gemm_flops_executed'] += 2.0 * r['n'] * r['k'] * r['out']
            w['gemm_launches'] += 1
            w['compulsory_bytes'] += 4.0 * r['n'] * (r['k'] + r['out'])
            w['staged_bytes'] += 4.0 * r['n'] * (r['k'] + r['out'])
        elif r['kind'] == 'gn':
            n, e, c, f, g = r['n'], r['e'], r['c'], r['f'], r['g']
            w['gemm_flops'] += 2.0 * n * c * U + 2.0 * n * (c + U + g) * U
            w['gemm_launches'] += 2
            w['edge_flops'] += 2.0 * e * (f + 3 * SMALL_U) * SMALL_U
            w['edge_launches'] += 1
            w['aggregate_bytes'] += 4.0 * (e * SMALL_U + e + n * U)
            w['aggregate_launches'] += 1
            w['gather_bytes'] += 4.0 * (2 * e * SMALL_U + 2 * e)
            w['compulsory_bytes'] += 4.0 * (n * c + e * f + 2 * e + e * SMALL_U + n * U)
            if 'n_coarse' in r:
                # embeddings of the coarse rows written and read, x read by the embedding and the column-sum
                # GEMM, one message per fine row written and read, fine rows written
                nc, nf = r['n_coarse'], r['n_fine']
                w['staged_bytes'] += 4.0 * (2 * nc * c + 2 * nc * U + 2 * e * SMALL_U + 2 * e + nf * U)
            else:
                w['staged_bytes'] += 4.0 * (2 * n * c + 3 * n * U + e * f + 2 * e + 2 * e * SMALL_U + 12 * n)
            # what the kernels execute: the global columns are folded into biases, edge features
            # are padded to a multiple of 8, and the upsampling layer skips its zero rows
            if 'n_coarse' in r:
                nc, nf = r['n_coarse'], r['n_fine']
                w['gemm_flops_executed'] += 2.0 * nc * c * U * 2 + 2.0 * nf * SMALL_U * U
                w['edge_flops_executed'] += 2.0 * e * SMALL_U * SMALL_U
            else:
                w['gemm_flops_executed'] += 2.0 * n * c * U + 2.0 * n * (c + U) * U
                w['edge_flops_executed'] += 2.0 * e * (-(-f // 8) * 8 + 2 * SMALL_U) * SMALL_U
                w['aggregate_bytes_executed'] += 4.0 * (e * SMALL_U + e + n * U)
        elif r['kind'] == 'pool':
            w['pool_bytes'] += 4.0 * (r['n'] * U + r['p'] * U) + 4.0 * (r['e'] * SMALL_U + r['e2'] * SMALL_U)
            w['staged_bytes'] += 4.0 * (r['n'] * U + r['p'] * U) + 4.0 * (r['e'] * SMALL_U + r['e2'] * SMALL_U)
    w['total_flops'] = w['gemm_flops'] + w['edge_flops']
    w['total_flops_executed'] = w['gemm_flops_executed'] + w['edge_flops_executed']
    return w


# Peaks used for the roofline fractions (dense, MI355X): float32 matrix cores and HBM3E.
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBPS = 8000.0
