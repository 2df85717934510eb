"""Add one workload's counter table to profiles/hbm_traffic_per_launch.json:
    python profiles/make_traffic_table.py <workload key> <hbm_traffic csv of summarize_pmc.py> [note] [forwards]
e.g. python profiles/make_traffic_table.py cfg2 profiles/r02_cfg2_hbm_traffic.csv

The file holds, per benchmark workload (bench.py's workload_key: cfg1, cfg2, cfg3, cfg5), the HBM
bytes per launch of every kernel class and `source_stamp`, the hash of the library sources the
passes were measured on (bench.py's source_stamp()); bench.py marks a table whose stamp differs from
the sources it runs as stale.

Kernel templates are merged into the classes bench.py's roofline speaks of (launch-weighted means):
k_gemm_resident<NT, NKT, GATHER, BF16, ...> -> 'k_gemm_resident<*, NKT>' for the float32
instantiations and 'k_gemm_resident<.., bf16>' for the bf16 ones; every other kernel by its name
without template arguments."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def class_of(name):
    name = name.replace('gunet::', '')
    m = re.match(r'k_gemm_resident<(\d+), (\d+), (\w+)(?:, (\w+))?', name)
    if m:
        return 'k_gemm_resident<.., bf16>' if m.group(4) == 'true' else f'k_gemm_resident<*, {m.group(2)}>'
    return name.split('<')[0]


# index kernels (the per-city plan build): not part of a forward
PLAN_KERNELS = ('k_scan_', 'k_fill_', 'k_count_keys', 'k_sort_rows', 'k_hash_insert', 'k_mark_first', 'k_assign_labels',
                'k_edge_keys', 'k_keep_flags', 'k_emit_edges', 'k_dup_map', 'k_pooled_loc', 'k_partition_scatter', 'k_pos_id',
                'k_mask_flags', 'k_max_row', 'k_child_mask', 'k_direction_mask', 'k_in_table', 'k_plan_')


def main(key, path, note='', forwards='6'):
    import bench
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if not row['kernel'].replace('gunet::', '').startswith('k_'):
                continue
            n = int(row['launches'])
            a = acc[class_of(row['kernel'])]
            a[0] += n
            a[1] += n * float(row['fetch_MB_per_launch_corrected_x2'])
            a[2] += n * float(row['write_MB_per_launch'])
    kernels = {k: {'launches': n, 'fetch_MB_per_launch': round(fe / n, 3), 'write_MB_per_launch': round(wr / n, 3),
                   'hbm_MB_per_launch': round((fe + wr) / n, 3)} for k, (n, fe, wr) in acc.items()}
    # k_global_update runs exactly once per forward: its launch count is the number of forwards profiled
    if 'k_global_update' in kernels:
        forwards = kernels['k_global_update']['launches']
    out = os.path.join(ROOT, 'profiles', 'hbm_traffic_per_launch.json')
    table = {'workloads': {}}
    if os.path.exists(out):
        with open(out) as f:
            old = json.load(f)
        if 'workloads' in old:
            table = old
    table['workloads'][key] = {
        'source': (f'{os.path.relpath(path, ROOT)}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (one counter per '
                   f'run) of the bench.py command of this workload with --steps 3 --warmup 1, FETCH_SIZE doubled (gfx950), '
                   f"averaged over each kernel class's launches" + (f'; {note}' if note else '')),
        'source_stamp': bench.source_stamp(), 'kernels': kernels,
        # bytes of one forward over all feature kernels: the profiled command ran `forwards` forwards
        # (first call + warm-up + steps + the one after the timed steps; counted, see above)
        'forwards': int(forwards),
        'feature_MB_per_forward': round(sum(v['hbm_MB_per_launch'] * v['launches'] for k, v in kernels.items()
                                            if not k.startswith(PLAN_KERNELS)) / int(forwards), 3)}
    with open(out, 'w') as f:
        json.dump(table, f, indent=1)
    print(key, 'stamp', table['workloads'][key]['source_stamp'], len(kernels), 'kernel classes')


if __name__ == '__main__':
    main(*sys.argv[1:5])
