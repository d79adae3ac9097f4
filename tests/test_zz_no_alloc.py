"""psvae_gine_aggregate_bwd_noalloc (the entry point that allocates nothing: its table-gradient
partials live in a fixed device array inside the library) against psvae_gine_aggregate_bwd on
the same inputs, through the C-ABI, bit for bit: two calls in a row, the second on a smaller
input (stale partials of the first call must not leak into it), then the size limit."""
import ctypes
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(V, H, C, seed, dev):
    from ps_vae_amd import ops
    rng = np.random.default_rng(seed)
    src, dst, cls = [], [], []
    for i in range(V):
        for _ in range(rng.integers(0, 5)):
            j = int(rng.integers(max(0, i - 20), min(V, i + 20)))
            src.append(j), dst.append(i), cls.append(int(rng.integers(0, C)))
    ei = torch.tensor([src, dst], dtype=torch.long).reshape(2, -1).to(dev)
    attr = torch.nn.functional.one_hot(torch.tensor(cls, dtype=torch.long), C).to(dev)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(V, H, generator=g).to(dev)
    table = (torch.randn(C, H, generator=g) * 0.5).to(dev)
    gy = torch.randn(V, H, generator=g).to(dev)
    add = torch.randn(V, H, generator=g).to(dev)
    return ops.Csr.from_edge_index(ei, attr, V), x, table, gy, add


def test_noalloc_backward_equals_workspace_backward_twice_then_smaller():
    from ps_vae_amd import hip
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    dev = torch.device('cuda:0')
    # the large case fills all 1024 partial slabs, the small one only a few of them
    for V, H, C, accumulate in ((40000, 300, 4, False), (700, 300, 4, True), (333, 128, 7, False)):
        csr, x, table, gy, add = _graph(V, H, C, seed=V + H, dev=dev)
        want_gx, got_gx = torch.empty_like(x), torch.empty_like(x)
        want_gt = torch.full_like(table, 0.25)
        got_gt = torch.full_like(table, 0.25)
        hip.gine_aggregate_bwd(x, table, gy, csr.t_row_ptr, csr.t_col, csr.t_cls, want_gx, want_gt, add=add,
                               accumulate_table=accumulate)
        # (garbage of earlier tests - autograd nodes in reference cycles that still hold device tensors - must not
        # be collected between the two readings: collect it now and keep the collector off in between)
        gc.collect()
        torch.cuda.synchronize()
        gc.disable()
        try:
            before = torch.cuda.memory_allocated(dev)
            hip.gine_aggregate_bwd_noalloc(x, table, gy, csr.t_row_ptr, csr.t_col, csr.t_cls, got_gx, got_gt, add=add,
                                           accumulate_table=accumulate)
            assert torch.cuda.memory_allocated(dev) == before
        finally:
            gc.enable()
        torch.cuda.synchronize()
        assert torch.equal(got_gx, want_gx), (V, H, C)
        assert torch.equal(got_gt, want_gt), (V, H, C)


def test_noalloc_rejects_sizes_beyond_its_scratch():
    from ps_vae_amd import hip
    L = hip.lib()
    P = ctypes.c_void_p(4096)
    # C * H = 8 * 300 = 2400 > PSVAE_NOALLOC_MAX_CH (2048): refused before anything is launched
    rc = L.psvae_gine_aggregate_bwd_noalloc(10, 300, 8, P, 300, P, P, 300, P, P, P, None, 300, P, 300, P, 0, None)
    assert rc == -2
    assert b'PSVAE_NOALLOC_MAX_CH' in L.psvae_last_error()
