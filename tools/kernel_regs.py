#!/usr/bin/env python3
"""Register / LDS / spill table of the kernels of one HIP source (hipcc remarks, no GPU needed).

usage: tools/kernel_regs.py graph-unet-traffic-prediction_amd/csrc/gemm_ops.hip [name filter]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', f'-I{ROOT}/include',
           '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null']
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r'remark: +(.*?) \[-Rpass', line)
        if not m:
            continue
        text = m.group(1).strip()
        if text.startswith('Function Name:'):
            cur = {'name': text.split(':', 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ':' in text:
            k, v = text.split(':', 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows),
                           capture_output=True, text=True).stdout.splitlines()
    print(f'{"VGPR":>5} {"AGPR":>5} {"spill":>6} {"SGPR":>5} {"occ":>4} {"LDS":>7}  kernel')
    for r, n in zip(rows, names):
        n = re.sub(r'gunet::\(anonymous namespace\)::', '', n)
        n = re.sub(r'\(.*$', '', n)
        n = re.sub(r'^void ', '', n)
        if flt and flt not in n:
            continue
        print(f'{r.get("VGPRs", "?"):>5} {r.get("AGPRs", "?"):>5} {r.get("VGPR Spill", r.get("ScratchSize [bytes/lane]", "?")):>6} '
              f'{r.get("SGPRs", "?"):>5} {r.get("Occupancy [waves/SIMD]", "?"):>4} {r.get("LDS Size [bytes/block]", "?"):>7}  {n}')


if __name__ == '__main__':
    main()
