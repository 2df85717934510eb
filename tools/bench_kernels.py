"""Micro-benchmarks of single library kernels on the GPU (development tool).
usage: python tools/bench_kernels.py gemm"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gunet_amd
from gunet_amd._lib import call, ptr, current_stream


def time_ms(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def gemm():
    dev = torch.device('cuda')
    if os.environ.get('GUNET_MODE'):
        gunet_amd.set_matmul_mode(os.environ['GUNET_MODE'])
    rows = int(os.environ.get('GUNET_ROWS', 96493 * 8))
    for k, n in [(128, 128), (256, 128), (384, 128), (104, 128), (128, 96)]:
        data = os.environ.get('GUNET_DATA', 'rand')
        x = {'rand': lambda: torch.rand((rows, k), device=dev), 'zeros': lambda: torch.zeros((rows, k), device=dev),
             'randn': lambda: torch.randn((rows, k), device=dev),
             'relu': lambda: torch.relu(torch.randn((rows, k), device=dev))}[data]()
        w = torch.rand((k, n), device=dev) - 0.5
        b = torch.rand((n,), device=dev)
        out = torch.empty((rows, n), device=dev)
        st = current_stream(dev)
        wsb = torch.empty(gunet_amd._lib.lib.gunet_dense_workspace_bytes(k, n), dtype=torch.uint8, device=dev)
        fn = lambda: call('gunet_dense_forward_f32', ptr(x), rows, k, ptr(w), ptr(b), n, 1, ptr(out), ptr(wsb), wsb.numel(), st)
        ms = time_ms(fn, iters=int(os.environ.get('GUNET_ITERS', 10)))
        ref = torch.relu(x[:4096].double() @ w.double() + b.double())
        fn()
        err = float((out[:4096].double() - ref).abs().max() / ref.abs().max())
        print(f'gemm rows={rows} k={k} n={n}: {ms:.3f} ms  {2.0 * rows * k * n / ms / 1e9:.1f} TFLOP/s  err {err:.1e}', flush=True)




if __name__ == '__main__':
    {'gemm': gemm}[sys.argv[1]]()
