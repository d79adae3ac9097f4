"""Micro-benchmark (not a test) of psvae_gemm_f32 on the shapes of one train step at batch
4096: forward X W^T, input gradient dY W, weight gradient dY^T X (+ bias gradient).
    python tests/bench_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd import hip  # noqa: E402

dev = torch.device('cuda:0')
V, P, B = 95167, 89201, 4096
# (rows, in, out) of the Linear layers, and how many times each runs per step
LAYERS = [('node_trans', V, 200, 300, 2), ('conv.0/2', V, 300, 300, 16), ('pair.0', P, 656, 328, 1),
          ('pair.2', P, 328, 328, 1), ('pair.4', P, 328, 656, 1), ('graph.linear', B, 1200, 400, 1),
          ('to_vocab', B * 7, 200, 302, 1), ('W_mean', B, 400, 56, 2)]


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


total = 0.0
print(f'{"layer":14s} {"M":>7s} {"K":>5s} {"N":>5s} | {"fwd us":>8s} {"TF":>6s} | {"dX us":>8s} {"TF":>6s} | {"dW us":>8s} {"TF":>6s}')
for name, M, K, N, reps in LAYERS:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    gy = torch.randn(M, N, device=dev)
    gx = torch.empty(M, K, device=dev)
    gw = torch.empty(N, K, device=dev)
    gb = torch.empty(N, device=dev)
    flop = 2.0 * M * N * K
    t_f = timeit(lambda: hip.gemm(M, N, K, x, K, 1, w, 1, K, y, N, bias=b, relu=True))
    t_x = timeit(lambda: hip.gemm(M, K, N, gy, N, 1, w, K, 1, gx, K, amask=y))
    t_w = timeit(lambda: hip.gemm(N, K, M, gy, 1, N, x, K, 1, gw, K, amask=y, split_k=0, rowsum=gb))
    total += reps * (t_f + t_x + t_w)
    print(f'{name:14s} {M:7d} {K:5d} {N:5d} | {t_f:8.1f} {flop / t_f / 1e6:6.1f} | {t_x:8.1f} {flop / t_x / 1e6:6.1f} | '
          f'{t_w:8.1f} {flop / t_w / 1e6:6.1f}')
print(f'sum over one step (with repeats): {total / 1e3:.2f} ms')
