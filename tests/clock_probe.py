"""Diagnostic (needs a library built with -DPSVAE_EXP_CLOCK): in-kernel clock and k-loop
cycles of one GEMM shape after sustained back-to-back launches."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd import hip  # noqa: E402

M, K, N = (int(a) for a in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else 'fwd'
secs = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
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
      'dxm': lambda: hip.gemm(M, K, N, gy, N, 1, w, K, 1, gx, K, amask=y),
      'dw': lambda: hip.gemm(N, K, M, gy, 1, N, x, K, 1, gw, K, split_k=0, rowsum=gb)}[mode]
fn()
torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    n += 50
el = time.time() - t0
print(f'{n} launches, {el / n * 1e6:.1f} us each, {2.0 * M * N * K * n / el / 1e12:.1f} TFLOP/s')
L = hip.lib()
L.psvae_exp_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(8 * 8192, dtype=np.uint64)
rc = L.psvae_exp_read_stamps(buf.ctypes.data, 8 * 8192)
s = buf.reshape(-1, 8)
s = s[s[:, 0] > 0]
s = s[s[:, 0] + np.uint64(100000) > s[:, 0].max()]   # the last launch only
nb = len(s)
t = s[:, :6].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0   # us since the first block entered
cyc = s[:, 6].astype(np.float64)
clk = cyc / ((t[:, 3] - t[:, 1]) * 1e-6)
print(f'rc={rc} blocks={nb} clock median {np.median(clk) / 1e9:.3f} GHz')
hw = s[:, 7]
xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
hwid = hw & np.uint64(0xffffffff)
cu = (hwid >> np.uint64(8)) & np.uint64(0xf)
sh = (hwid >> np.uint64(12)) & np.uint64(0x1)
se = (hwid >> np.uint64(13)) & np.uint64(0x7)
key = (xcc.astype(np.int64) << 8) | (se.astype(np.int64) << 5) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
print('distinct CUs seen', len(np.unique(key)))
names = ['entry', 'loads issued', 'first tile in LDS', 'loop end', 'stores issued', 'stores done']
for i in range(1, 6):
    d = t[:, i] - t[:, i - 1]
    print(f'{names[i - 1]:>18s} -> {names[i]:<18s}: median {np.median(d):7.2f} us  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f}  max {d.max():7.2f}')
print('kernel span', t[:, 5].max().round(1), 'us')
for k in np.unique(key)[:3]:
    idx = np.nonzero(key == k)[0]
    idx = idx[np.argsort(t[idx, 0])]
    print(f'CU key {k:#x}:')
    for i in idx:
        print('   block', i, ' '.join(f'{v:7.1f}' for v in t[i]))
