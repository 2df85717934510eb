"""Sum rocprofv3 counter rows per kernel: python tools/pmc_table.py <counter_collection.csv>"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
        rows[name][r['Counter_Name']] += float(r['Counter_Value'])
        calls[name].add(r['Dispatch_Id'])
for name, c in rows.items():
    n = len(calls[name])
    print(name, 'dispatches', n)
    for k, v in sorted(c.items()):
        print(f'    {k:28s} {v / n:16.0f} per dispatch')
