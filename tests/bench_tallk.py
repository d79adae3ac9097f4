"""Micro-benchmark (not a test): the stacked weight-gradient product of the encoder
(300 x 300 (+ row sums) over K = 380 140 rows) through psvae_gemm_f32.
    python tests/bench_tallk.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd import hip  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda:0')
M, N, K = 300, 300, 380140
dy = torch.randn(K, M, device=dev)
x = torch.randn(K, N, device=dev)
gw = torch.empty(M, N, device=dev)
gb = torch.empty(M, device=dev)


def run():
    hip.gemm(M, N, K, dy, 1, M, x, N, 1, gw, N, split_k=0, rowsum=gb)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
print(f'dW {M}x{N}x{K}: {us:.1f} us  {2.0 * M * N * K / us / 1e6:.1f} TFLOP/s  (lib {os.path.basename(hip.LIB_PATH)})')
