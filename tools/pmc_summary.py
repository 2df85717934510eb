"""Per-dispatch SQ counter summary of the GEMM micro-benchmark: python tools/pmc_summary.py <dir>"""
import csv
import sys
from collections import OrderedDict

d = OrderedDict()
trace = {}
base = sys.argv[1]
for r in csv.DictReader(open(f'{base}/pmc_kernel_trace.csv')):
    trace[int(r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for r in csv.DictReader(open(f'{base}/pmc_counter_collection.csv')):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
    if 'k_gemm' not in name:
        continue
    d.setdefault((int(r['Dispatch_Id']), name), {})[r['Counter_Name']] = float(r['Counter_Value'])
keys = list(d.keys())
for i in range(5, len(keys), 14):
    c = d[keys[i]]
    gui = c['GRBM_GUI_ACTIVE'] / 8
    us = trace.get(keys[i][0], 0)
    print(f"{keys[i][1][-28:]:28s} {us:8.1f} us  clock {gui / us / 1e3 if us else 0:5.2f} GHz  mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024):.3f}"
          f"  wait_any {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f}  wait_inst {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}"
          f"  active {c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}  waves {int(c['SQ_WAVES'])}")
