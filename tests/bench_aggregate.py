"""Micro-benchmark (not a test): the GINE aggregation kernels alone on a ZINC-shaped batch of 4096
molecules: forward (plain / leaving message bits), backward (from x / from bits, with and without the
per-graph add rows).  PSVAE_HIP_LIB selects the library build."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd import hip  # noqa: E402
from ps_vae_amd.data.bpe_dataset import BPEMolDataset  # noqa: E402
from ps_vae_amd.data.packed import PackedMolecules  # noqa: E402
from ps_vae_amd.data.synthetic import synthetic_step1  # noqa: E402
from ps_vae_amd.utils.tokenizer import Tokenizer  # noqa: E402

B, H, C = 4096, 300, 4
dev = torch.device('cuda:0')
tok = Tokenizer()
cache = f'/tmp/psvae_synth_zinc_{B}_2021.npz'
packed = PackedMolecules.load(cache) if os.path.exists(cache) else PackedMolecules.from_step1(
    synthetic_step1(B, seed=2021, shape='zinc'))
data = BPEMolDataset(None, tok, device=dev, data=packed)
c = data.collate_ids(np.arange(B)).compact
V = c.num_nodes
csr = c.csr
x = torch.randn(V, H, device=dev)
g = torch.randn(V, H, device=dev)
add = torch.randn(V, H, device=dev)
table = torch.randn(C, H, device=dev) * 0.5
out, gx, gt = torch.empty(V, H, device=dev), torch.empty(V, H, device=dev), torch.empty(C, H, device=dev)
words, wtype = hip.gine_mask_words(V, H, C)
mask = torch.zeros(words, dtype=wtype, device=dev)
seg = torch.randn(B, 4 * H, device=dev)
node_graph = c.segments.node_graph(V)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


flush_src = torch.randn(160 << 20, device=dev)     # 640 MB: more than the last-level cache holds
flush_dst = torch.empty_like(flush_src)


def timed_cold(fn, iters=12):
    """Each call after 1.3 GB of unrelated traffic (what the products between two aggregations of a
    training step do to the caches), timed by its own pair of events."""
    total = 0.0
    for _ in range(iters + 2):
        flush_dst.copy_(flush_src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if _ >= 2:
            total += e0.elapsed_time(e1)
    return total / iters * 1e3


E = csr.col.numel()
fb, bb = hip.aggregate_bytes(V, E, H), hip.aggregate_bytes(V, E, H, backward=True)
rows = [
    ('plain copy out = x (yardstick)', lambda: out.copy_(x), 2 * V * H * 4),
    ('fwd plain', lambda: hip.gine_aggregate_fwd(x, table, csr.row_ptr, csr.col, csr.cls, out), fb),
    ('fwd + bits', lambda: hip.gine_aggregate_fwd_bits(x, table, csr.row_ptr, csr.col, csr.cls, out, mask), fb),
    ('bwd from x, add', lambda: hip.gine_aggregate_bwd(x, table, g, csr.t_row_ptr, csr.t_col, csr.t_cls, gx, gt, add=add), bb),
    ('bwd from bits, add', lambda: hip.gine_aggregate_bwd_bits(mask, C, g, csr.t_row_ptr, csr.t_col, csr.t_cls, gx, gt, add=add), bb),
    ('bwd from bits, graph rows', lambda: hip.gine_aggregate_bwd_bits(mask, C, g, csr.t_row_ptr, csr.t_col, csr.t_cls, gx, gt,
                                                                     seg_add=seg[:, H:2 * H], node_graph=node_graph), bb),
    ('bwd from bits', lambda: hip.gine_aggregate_bwd_bits(mask, C, g, csr.t_row_ptr, csr.t_col, csr.t_cls, gx, gt), bb),
]
print(f'V={V} E={E} H={H}')
for name, fn, nbytes in rows:
    us, cold = timed(fn), timed_cold(fn)
    print(f'{name:32s} {us:8.1f} us (cold caches {cold:6.1f} us)   {nbytes / us / 1e3:7.0f} GB/s of algorithmic bytes ({nbytes / us / 8e6:.3f} of 8 TB/s)')
