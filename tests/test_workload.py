"""Algorithmic-work accounting used for the roofline fractions, against hand-derived totals."""


def test_dense_grid_totals():
    from gunet_amd import workload
    n = [215820, 54064, 13516, 3410]
    e = [2582396, 645044, 160332, 39988]
    rows = workload.layer_table(n, e)
    assert sum(r['kind'] == 'gn' for r in rows) == 14
    w = workload.work_per_sample(rows)
    # 278.6 GFLOP per sample on the dense 495x436 grid: in 5.75 + out 5.30 + 14 GN layers 267.5
    assert abs(w['total_flops'] / 1e9 - 278.6) < 0.1
    dense = [r for r in rows if r['kind'] == 'dense']
    assert abs(2.0 * dense[0]['n'] * dense[0]['k'] * dense[0]['out'] / 1e9 - 5.75) < 0.01
    assert abs(2.0 * dense[1]['n'] * dense[1]['k'] * dense[1]['out'] / 1e9 - 5.30) < 0.01
    dense_bytes = sum(4.0 * r['n'] * (r['k'] + r['out']) for r in dense)
    assert abs((w['compulsory_bytes'] - dense_bytes) / 1e9 - 3.80) < 0.01      # the 14 GN layers
    assert abs((w['aggregate_bytes'] + w['gather_bytes']) / 1e9 - 4.79) < 0.01
    assert w['gemm_launches'] == 30 and w['edge_launches'] == 14


def test_layer_shapes_follow_checkpoint():
    from gunet_amd import workload
    rows = {r['name']: r for r in workload.layer_table([100, 40, 12, 4], [300, 100, 30, 8])}
    assert (rows['down0.gcn']['c'], rows['down0.gcn']['f'], rows['down0.gcn']['g']) == (128, 16, 113)
    assert (rows['up0.upsample']['f'], rows['up0.upsample']['g']) == (2, 256)
    assert rows['up2.upsample']['g'] == 241 and rows['up2.upsample']['n'] == 140 and rows['up2.upsample']['e'] == 100
    assert (rows['up2.gcn0']['c'], rows['up2.gcn0']['f']) == (256, 16)
