"""Micro-benchmark (not a test): every psvae_gemm_f32 call of ONE train step at batch 4096
(shapes, strides and flags recorded from a real step), each replayed in a loop of its own on
random operands.  Prints the per-call table (slowest first) and the sum over the step, which is
the figure to compare between builds of the library:
    PSVAE_HIP_LIB=exp_libs/libpsvae_x.so python tests/bench_step_gemms.py [batch] [top]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd import hip  # noqa: E402
from ps_vae_amd.data.bpe_dataset import BPEMolDataset  # noqa: E402
from ps_vae_amd.data.packed import PackedMolecules  # noqa: E402
from ps_vae_amd.data.synthetic import synthetic_step1  # noqa: E402
from ps_vae_amd.pl_models import PSVAEModel  # noqa: E402
from ps_vae_amd.utils.nn_utils import build_config, default_args  # noqa: E402
from ps_vae_amd.utils.tokenizer import Tokenizer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = torch.device('cuda:0')
tok = Tokenizer()
torch.manual_seed(2021)
model = PSVAEModel(build_config(default_args(), tok), tok).to(dev)
model.two_streams = False
cache = f'/tmp/psvae_synth_zinc_{B}_2021.npz'
packed = PackedMolecules.load(cache) if os.path.exists(cache) else PackedMolecules.from_step1(
    synthetic_step1(B, seed=2021, shape='zinc'))
data = BPEMolDataset(None, tok, device=dev, data=packed)

calls = []
real_gemm = hip.gemm


def recording_gemm(M, N, K, A, sa_m, sa_k, Bm, sb_k, sb_n, C, ldc, bias=None, relu=False, accumulate=False,
                   amask=None, split_k=0, rowsum=None, cmask=None):
    calls.append((M, N, K, sa_m, sa_k, sb_k, sb_n, ldc, bias is not None, relu, accumulate, amask is not None,
                  split_k, rowsum is not None, cmask is not None))
    return real_gemm(M, N, K, A, sa_m, sa_k, Bm, sb_k, sb_n, C, ldc, bias=bias, relu=relu, accumulate=accumulate,
                     amask=amask, split_k=split_k, rowsum=rowsum, cmask=cmask)


real_bits = hip.gemm_bits


def recording_gemm_bits(M, N, K, A, sa_m, sa_k, Bm, sb_k, sb_n, C, ldc, bias=None, relu=False, relu_bits=None,
                        gate_bits=None):
    # recorded with the flag slots of a plain call: 'B' = leaves sign bits behind, 'G' = gated by sign bits
    calls.append((M, N, K, sa_m, sa_k, sb_k, sb_n, ldc, bias is not None, relu, False, False,
                  'B' if relu_bits is not None else 'G', False, False))
    return real_bits(M, N, K, A, sa_m, sa_k, Bm, sb_k, sb_n, C, ldc, bias=bias, relu=relu, relu_bits=relu_bits,
                     gate_bits=gate_bits)


batch = data.collate_ids(np.arange(B))
for _ in range(2):      # the second step is the recorded one (first-call set-up is over)
    calls.clear()
    hip.gemm, hip.gemm_bits = recording_gemm, recording_gemm_bits
    for mod in list(sys.modules.values()):
        if getattr(mod, 'hip', None) is hip:
            pass
    model.zero_grad()
    model.training_step(batch, 0).backward()
    hip.gemm, hip.gemm_bits = real_gemm, real_bits
torch.cuda.synchronize()

uniq = {}
for c in calls:
    uniq[c] = uniq.get(c, 0) + 1


def span(n0, s0, n1, s1):
    return (n0 - 1) * s0 + (n1 - 1) * s1 + 1


rows = []
for c, count in uniq.items():
    M, N, K, sa_m, sa_k, sb_k, sb_n, ldc, has_bias, relu, acc, has_amask, split_k, has_rowsum, has_cmask = c
    A = torch.randn(span(M, sa_m, K, sa_k) + 3, device=dev)
    Bm = torch.randn(span(K, sb_k, N, sb_n) + 3, device=dev) * 0.05
    C = torch.zeros(M * ldc, device=dev)
    bias = torch.randn(N, device=dev) if has_bias else None
    amask = torch.randn(A.numel(), device=dev) if has_amask else None
    cmask = torch.randn(C.numel(), device=dev) if has_cmask else None
    rowsum = torch.zeros(M, device=dev) if has_rowsum else None

    bits = torch.zeros(hip.gemm_bits_words(M, N), dtype=torch.int64, device=dev) if split_k in ('B', 'G') else None

    def run():
        if split_k == 'B':
            real_bits(M, N, K, A, sa_m, sa_k, Bm, sb_k, sb_n, C, ldc, bias=bias, relu=relu, relu_bits=bits)
        elif split_k == 'G':
            real_bits(M, N, K, A, sa_m, sa_k, Bm, sb_k, sb_n, C, ldc, bias=bias, relu=relu, gate_bits=bits)
        else:
            real_gemm(M, N, K, A, sa_m, sa_k, Bm, sb_k, sb_n, C, ldc, bias=bias, relu=relu, accumulate=acc, amask=amask,
                      split_k=split_k, rowsum=rowsum, cmask=cmask)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    iters = 6
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    rows.append((us * count, us, count, c))
    del A, Bm, C

rows.sort(reverse=True)
total = sum(r[0] for r in rows)
flops = sum(2.0 * c[0] * c[1] * c[2] * n for c, n in uniq.items())
print(f'{len(calls)} GEMM calls per step, {len(uniq)} distinct; sum {total / 1e3:.3f} ms; '
      f'{flops / 1e12:.3f} TFLOP -> {flops / total / 1e6:.1f} TFLOP/s')
print('  total_us   each_us  n |      M     N      K  A    B   flags')
for tot, us, count, c in rows[:TOP]:
    M, N, K, sa_m, sa_k, sb_k, sb_n, ldc, has_bias, relu, acc, has_amask, split_k, has_rowsum, has_cmask = c
    flags = ''.join(f for f, on in (('b', has_bias), ('r', relu), ('+', acc), ('m', has_amask), ('s', has_rowsum),
                                    ('g', has_cmask), ('B', split_k == 'B'), ('G', split_k == 'G')) if on)
    print(f'{tot:10.1f} {us:9.1f} {count:2d} | {M:6d} {N:5d} {K:6d}  {"k" if sa_k == 1 else "m"}    '
          f'{"k" if sb_k == 1 else "n"}   {flags:6s} {2.0 * M * N * K / us / 1e6:6.1f} TF')
