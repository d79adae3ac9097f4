"""Micro-benchmark (not a test): host time per call of an entry point through its torch.library
operator and through a raw ctypes call, on tiny operands (the GPU work is negligible)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd import hip  # noqa: E402

d = torch.device('cuda:0')
x = torch.randn(64, 300, device=d)
w = torch.randn(300, 300, device=d)
out = torch.empty(64, 300, device=d)
beg = torch.tensor([0, 32], dtype=torch.int32, device=d)
end = torch.tensor([32, 64], dtype=torch.int32, device=d)
pooled = torch.empty(2, 300, device=d)
L = hip.lib()
N = 5000


def timeit(fn):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / N * 1e6


def gemm_ctypes():
    hip._check(L.psvae_gemm_f32(64, 300, 300, hip.ptr(x, torch.float32), None, 300, 1, hip.ptr(w, torch.float32), 1, 300,
                                hip.ptr(out, torch.float32), 300, None, None, None, 0, 0, 1, None, 0, hip._stream()), 'gemm')


def seg_ctypes():
    px, ldx = hip._rows(x)
    po, ldo = hip._rows(pooled)
    hip._check(L.psvae_segment_sum_fwd(2, 300, px, ldx, hip.ptr(beg, torch.int32), hip.ptr(end, torch.int32), po, ldo,
                                       hip._stream()), 'seg')


print('gemm          operator %.2f us   ctypes %.2f us' % (
    timeit(lambda: hip.gemm(64, 300, 300, x, 300, 1, w, 1, 300, out, 300, split_k=1)), timeit(gemm_ctypes)))
print('segment_sum   operator %.2f us   ctypes %.2f us' % (
    timeit(lambda: hip.segment_sum_fwd(x, beg, end, pooled)), timeit(seg_ctypes)))
print('torch.empty   %.2f us' % timeit(lambda: torch.empty(64, 300, device=d)))
