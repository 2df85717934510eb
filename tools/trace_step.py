"""Print the kernels of the last forward of a tools/trace_step.sh run, in launch order."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('gunet::', '')) for r in rows]
# a forward starts with the input dense layer: the GEMM launch that precedes the first k_layer_prologue
starts = [i for i, n in enumerate(names) if n.startswith('k_layer_prologue') and not any(m.startswith('k_layer_prologue') for m in names[max(0, i - 20):i])]
first = starts[-1] - 1
t0 = int(rows[first]['Start_Timestamp'])
prev_end = t0
total = 0.0
for r, n in zip(rows[first:], names[first:]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f'{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {n[:60]}  grid {r.get("Grid_Size_X", r.get("Grid_Size", ""))}')
    prev_end = e
    total += (e - s) / 1e3
print('sum of kernel durations', total, 'us; span', (prev_end - t0) / 1e3, 'us')
