"""Diagnostic (needs a library built with -DPSVAE_EXP_CLOCK): where the epilogue of the conv
product spends its time, from stamps of wave 0 of every workgroup of the last launch."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd import hip  # noqa: E402

M, K, N = 95167, 300, 300
dev = torch.device('cuda:0')
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.05
b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
t0 = time.time()
while time.time() - t0 < 1.0:
    for _ in range(50):
        hip.gemm(M, N, K, x, K, 1, w, 1, K, y, N, bias=b, relu=True)
    torch.cuda.synchronize()
L = hip.lib()
n = 8192
main = np.zeros(8 * n, dtype=np.uint64)
epi = np.zeros(12 * n, dtype=np.uint64)
for fn, buf in ((L.psvae_exp_read_stamps, main), (L.psvae_exp_read_epi, epi)):
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn(buf.ctypes.data, buf.size)
main, epi = main.reshape(-1, 8), epi.reshape(-1, 12)
ok = (main[:, 0] > 0) & (main[:, 0] + np.uint64(100000) > main[:, 0].max()) & (epi[:, 0] > 0)
main, epi = main[ok].astype(np.float64), epi[ok].astype(np.float64)
loop_end = main[:, 3]
names = ['loop end -> bias rows ready']
for j in range(5):
    names += [f'tile {j}: -> patch written', f'tile {j}: -> 4 stores issued']
prev = loop_end
print(f'{len(main)} workgroups')
for q, name in enumerate(names):
    cur = epi[:, q]
    live = cur > 0
    d = (cur[live] - prev[live]) / 100.0
    print(f'{name:32s} median {np.median(d):6.2f} us  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}')
    prev = np.where(live, cur, prev)
