"""Micro-benchmark (not a test): one psvae_gemm_f32 shape, for profiler counter passes.
    python tests/bench_gemm_one.py M K N fwd|dx|dw [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd import hip  # noqa: E402

M, K, N = (int(a) for a in sys.argv[1:4])
mode = sys.argv[4]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device('cuda:0')
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
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fn()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
print(f'{mode} M={M} K={K} N={N}: {us:.1f} us  {2.0 * M * N * K / us / 1e6:.1f} TFLOP/s')
