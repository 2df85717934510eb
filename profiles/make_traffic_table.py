"""hbm_traffic_per_launch.json from a summarize_pmc.py table:
    python profiles/make_traffic_table.py profiles/r01_bench_n1_hbm_traffic_v4.csv > profiles/hbm_traffic_per_launch.json

Kernel templates are merged into the classes bench.py's roofline speaks of (launch-weighted means):
k_gemm_resident<NT, NKT, GATHER, BF16, RAGGED> -> 'k_gemm_resident<*, NKT>' for the float32
instantiations and 'k_gemm_resident<.., bf16>' for the bf16 ones; every other kernel by its name
without template arguments."""
import csv
import json
import re
import sys
from collections import defaultdict


def class_of(name):
    name = name.replace('gunet::', '')
    m = re.match(r'k_gemm_resident<(\d+), (\d+), (\w+)(?:, (\w+))?(?:, (\w+))?>', name)
    if m:
        return 'k_gemm_resident<.., bf16>' if m.group(4) == 'true' else f'k_gemm_resident<*, {m.group(2)}>'
    return name.split('<')[0]


def main(path):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            n = int(row['launches'])
            a = acc[class_of(row['kernel'])]
            a[0] += n
            a[1] += n * float(row['fetch_MB_per_launch_corrected_x2'])
            a[2] += n * float(row['write_MB_per_launch'])
    kernels = {k: {'launches': n, 'fetch_MB_per_launch': round(fe / n, 3), 'write_MB_per_launch': round(wr / n, 3),
                   'hbm_MB_per_launch': round((fe + wr) / n, 3)} for k, (n, fe, wr) in acc.items()}
    source = (f'{path}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `bench.py --steps 3 --warmup 1`, '
              "FETCH_SIZE doubled (gfx950), averaged over each kernel class's launches")
    json.dump({'source': source, 'kernels': kernels}, sys.stdout, indent=1)


if __name__ == '__main__':
    main(sys.argv[1])
