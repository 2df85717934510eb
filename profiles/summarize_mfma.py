"""Per-kernel matrix-core occupancy and effective clock from one rocprofv3 counter pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace ... -- python3 bench.py ...
    python profiles/summarize_mfma.py <counter_collection.csv> <kernel_trace.csv> > profiles/r01_bench_n1_mfma_busy_vN.csv

GRBM_GUI_ACTIVE is summed over the 8 XCDs (so / 8 = busy GPU cycles of the dispatch; divided by its
duration = the clock it ran at); SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, so
MFMA_BUSY / (GUI_ACTIVE / 8 * 1024) = fraction of SIMD-cycles with the matrix pipe busy."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '').replace('gunet::', '')
    return re.sub(r'\(.*', '', name)


def main(counters, trace):
    dur = {}
    for r in csv.DictReader(open(trace)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    per = defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(counters)):
        per[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
        names[r['Dispatch_Id']] = short(r['Kernel_Name'])
    acc = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for d, c in per.items():
        if 'GRBM_GUI_ACTIVE' not in c or d not in dur:
            continue
        a = acc[names[d]]
        a[0] += 1
        a[1] += dur[d]
        a[2] += c['GRBM_GUI_ACTIVE'] / 8
        a[3] += c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'launches', 'avg_us', 'clock_GHz', 'mfma_busy_frac'])
    for k, (n, us, gui, mfma) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        if us / n < 5 or not k.startswith('k_'):
            continue
        w.writerow([k, n, round(us / n, 1), round(gui / us / 1e3, 3), round(mfma / (gui * 1024), 3)])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
