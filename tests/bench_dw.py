"""Sweep (not a test): weight-gradient GEMM (dY^T X, K = rows of the batch) vs split_k."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd import hip  # noqa: E402
from bench_gemm import timeit  # noqa: E402

dev = torch.device('cuda:0')
M, K, N = 95167, 300, 300
x = torch.randn(M, K, device=dev)
gy = torch.randn(M, N, device=dev)
y = torch.randn(M, N, device=dev)
gw = torch.empty(N, K, device=dev)
gb = torch.empty(N, device=dev)
only = os.environ.get('ONLY')
for mask, ones in ((False, False), (True, True)):
    for split in (8, 16, 32, 64, 128, 256, 512):
        if only and int(only) != split:
            continue
        t = timeit(lambda: hip.gemm(N, K, M, gy, 1, N, x, K, 1, gw, K, amask=y if mask else None, split_k=split,
                                    rowsum=gb if ones else None), iters=5)
        print(f'mask={mask} ones={ones} split={split:4d}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF')
