"""Micro-benchmark (not a test): latency of psvae_gemm_f32 on the small products of a batch-32 step, measured as
the recorded step runs them: 40 dependent calls in one HIP graph, replayed.
    python tests/bench_small_gemm.py            (PSVAE_HIP_LIB=... for an A/B build)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd import hip  # noqa: E402

dev = torch.device('cuda:0')
CALLS = 40


def time_graph(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(CALLS):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 / CALLS * 1e3


def case(M, K, N, mode):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    gy = torch.randn(M, N, device=dev)
    gx = torch.empty(M, K, device=dev)
    gw = torch.empty(N, K, device=dev)
    gb = torch.empty(N, device=dev)
    fn = {'fwd': lambda: hip.gemm(M, N, K, x, K, 1, w, 1, K, y, N, bias=b, relu=True),
          'dx': lambda: hip.gemm(M, K, N, gy, N, 1, w, K, 1, gx, K),
          'dw': lambda: hip.gemm(N, K, M, gy, 1, N, x, K, 1, gw, K, split_k=0, rowsum=gb)}[mode]
    us = time_graph(fn)
    print(f'{mode:3s} rows={M:5d} in={K:4d} out={N:4d}: {us:6.2f} us per call', flush=True)


for rows in (290, 750):
    for mode in ('fwd', 'dx', 'dw'):
        case(rows, 300, 300, mode)
    case(4 * rows, 300, 300, 'dw')
case(32, 400, 1200, 'fwd')
case(32, 56, 400, 'fwd')
case(4000, 328, 328, 'fwd')
case(4000, 328, 328, 'dw')
case(4096, 300, 300, 'dw')
