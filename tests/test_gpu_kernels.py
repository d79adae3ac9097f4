"""Per-kernel parity on the GPU: every HIP entry point, called through the C-ABI binding,
against the CPU oracle / a plain fp32-or-better PyTorch restatement on the same inputs.

Tolerances: integer outputs bit-exact; fp32 within 1e-5 relative to the magnitude of the
reference tensor (the tolerance BASELINE.json's north_star states)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz, sub
from oracle import collate_ref, model_ref

pytestmark = pytest.mark.gpu

REL = 1e-5


def dev():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU (run with -m "not gpu" on CPU-only machines)')
    return torch.device('cuda:0')


ELEM_REL = 1e-3   # element-wise bound on the entries that are not small (see close)


def close(got, want, rel=REL, what='', elem_rel=ELEM_REL):
    """Two bounds.  (1) The largest absolute error is at most `rel` of the tensor's largest entry
    (the 1e-5 relative tolerance of the path, taken over the tensor: an entry that is small because
    its terms cancel has no relative accuracy of its own in fp32).  (2) Element by element, every
    entry of at least 1e-3 of the largest one is within `elem_rel` of its own value: a wrong entry
    among the small ones is not hidden behind the tensor's scale."""
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    assert got.shape == want.shape, f'{what}: shape {tuple(got.shape)} vs {tuple(want.shape)}'
    if want.numel() == 0:
        return
    scale = want.abs().max().item()
    diff = (got - want).abs()
    err = diff.max().item()
    assert err <= rel * max(scale, 1e-30) + 1e-30, f'{what}: max abs err {err:.3e} vs scale {scale:.3e}'
    big = want.abs() >= 1e-3 * scale
    if scale > 0 and elem_rel is not None and bool(big.any()):
        worst = (diff[big] / want.abs()[big]).max().item()
        assert worst <= elem_rel, f'{what}: an entry above 1e-3 of the largest is off by {worst:.3e} of its own value'


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('M,N,K', [(1, 1, 1), (7, 5, 3), (64, 64, 32), (130, 300, 200), (1000, 300, 300),
                                   (333, 656, 328), (257, 4, 656), (40, 56, 400), (129, 65, 33),
                                   (4096, 400, 1200), (2052, 328, 656), (1028, 96, 100), (512, 224, 64),
                                   (900, 352, 44), (384, 708, 16)])
def test_gemm_forward_shapes(M, N, K):
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    xd, wd, bd = x.to(d), w.to(d), b.to(d)
    out = torch.empty(M, N, device=d)
    hip.gemm(M, N, K, xd, K, 1, wd, 1, K, out, N, bias=bd, relu=True)
    close(out, torch.relu(x.double() @ w.double().t() + b.double()), what='relu(xW^T+b)')
    # input-gradient form dY W with the ReLU mask applied to dY while loading
    gy = torch.randn(M, N, generator=g)
    act = torch.relu(torch.randn(M, N, generator=g))
    gx = torch.empty(M, K, device=d)
    hip.gemm(M, K, N, gy.to(d), N, 1, wd, K, 1, gx, K, amask=act.to(d))
    close(gx, (gy.double() * (act > 0)) @ w.double(), what='dX')
    # the gate on the result instead: (dY W) * [mask > 0], also through the split-K reduction
    gate = torch.relu(torch.randn(M, K, generator=g))
    for split in (1, 3):
        gx2 = torch.empty(M, K, device=d)
        hip.gemm(M, K, N, gy.to(d), N, 1, wd, K, 1, gx2, K, cmask=gate.to(d), split_k=split)
        close(gx2, (gy.double() @ w.double()) * (gate > 0), what=f'gated dX split {split}')
    # weight-gradient form dY^T X, split-K chosen by the library and forced
    for split in (0, 1, 5):
        gw = torch.full((N, K), 3.0, device=d)
        gb = torch.full((N,), -2.0, device=d)
        hip.gemm(N, K, M, gy.to(d), 1, N, xd, K, 1, gw, K, amask=act.to(d), split_k=split, accumulate=True,
                 rowsum=gb)
        close(gw, 3.0 + (gy.double() * (act > 0)).t() @ x.double(), what=f'dW split {split}')
        close(gb, -2.0 + (gy.double() * (act > 0)).sum(0), what=f'db split {split}')


def test_gemm_strided_output_and_operands():
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator().manual_seed(5)
    big = torch.randn(200, 1200, generator=g)
    w = torch.randn(300, 300, generator=g)
    bigd = big.to(d)
    out = torch.zeros(200, 1200, device=d)
    xs = bigd[:, 300:600]                      # row stride 1200
    hip.gemm(200, 300, 300, xs, 1200, 1, w.to(d), 1, 300, out[:, 600:900], 1200)
    want = torch.zeros(200, 1200, dtype=torch.float64)
    want[:, 600:900] = big[:, 300:600].double() @ w.double().t()
    close(out, want, what='strided slot')


@pytest.mark.parametrize('M,N,K', [(1000, 300, 328), (64, 96, 40), (31, 8, 12)])
def test_gemm_rowsum_with_row_major_operands(M, N, K):
    """rowsum = sum_k A(m,k) with A [M,K] row-major (k-contiguous) and B [K,N] row-major
    (n-contiguous): the input-gradient form of the panel kernel, and the small generic one."""
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(K, N, generator=g)
    ad, bd = a.to(d), b.to(d)
    for accumulate in (False, True):
        c = torch.full((M, N), 0.5, device=d)
        rs = torch.full((M,), -1.0, device=d)
        hip.gemm(M, N, K, ad, K, 1, bd, N, 1, c, N, rowsum=rs, accumulate=accumulate, split_k=1)
        base = 1.0 if accumulate else 0.0
        close(c, base * 0.5 + a.double() @ b.double(), what='C')
        close(rs, base * -1.0 + a.double().sum(1), what='rowsum')


@pytest.mark.parametrize('M,N,K,lda,ldb', [(300, 300, 50003, 300, 300), (304, 100, 20000, 304, 100),
                                           (16, 8, 16400, 16, 8), (300, 300, 33000, 1200, 1200),
                                           (120, 304, 70000, 120, 304), (300, 300, 1000, 300, 300)])
def test_gemm_tall_k_weight_gradient(M, N, K, lda, ldb):
    """dW = dY^T X (+ db from the same pass) with a small output and a very long contraction:
    the tall-K kernel (whole output per workgroup, operands straight to LDS) where it applies,
    the panel kernel for the short case; row-strided operands (slots of a wider matrix), a K
    that is no multiple of the k-tile, accumulate and a gate on the result."""
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(K, lda, generator=g)
    x = torch.randn(K, ldb, generator=g)
    want = dy[:, :M].double().t() @ x[:, :N].double()
    dyd, xd = dy.to(d), x.to(d)
    gw = torch.full((M, N), 2.0, device=d)
    gb = torch.full((M,), -1.0, device=d)
    hip.gemm(M, N, K, dyd, 1, lda, xd, ldb, 1, gw, N, split_k=0, rowsum=gb, accumulate=True)
    close(gw, 2.0 + want, what='dW (+=)')
    close(gb, -1.0 + dy[:, :M].double().sum(0), what='db (+=)')
    gw2 = torch.empty(M, N, device=d)
    bias = torch.randn(N, generator=g)
    gate = torch.randn(M, N, generator=g)
    hip.gemm(M, N, K, dyd, 1, lda, xd, ldb, 1, gw2, N, split_k=0, bias=bias.to(d), relu=True, cmask=gate.to(d))
    close(gw2, torch.relu(want + bias.double()) * (gate > 0), what='relu(dW + b) gated')


@pytest.mark.parametrize('M,N,K,lda,ldc', [(95148, 300, 300, 300, 300), (8192, 320, 16, 16, 320),
                                           (8269, 164, 44, 44, 164), (20001, 300, 300, 1200, 1200),
                                           (33000, 200, 1200, 1200, 200), (70000, 300, 20, 20, 300)])
def test_gemm_tall_narrow_products(M, N, K, lda, ldc):
    """Tall products with k-contiguous operands and a narrow output (the per-node Linear layers of
    the benchmark batch): forward with bias and ReLU, the gated and the accumulating form, ragged
    last panel / last k-tile, row-strided A and C, repeats bit for bit."""
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator(device=d).manual_seed(M + N + K)
    a = torch.randn(M, lda, device=d, generator=g)
    w = torch.randn(N, K, device=d, generator=g)
    b = torch.randn(N, device=d, generator=g)
    want = a[:, :K].double() @ w.double().t()
    out = torch.full((M, ldc), 7.0, device=d)
    hip.gemm(M, N, K, a, lda, 1, w, 1, K, out, ldc, bias=b, relu=True)
    close(out[:, :N], torch.relu(want + b.double()), what='relu(xW^T+b)')
    if ldc > N:
        assert bool((out[:, N:] == 7.0).all()), 'columns beyond N were written'
    gate = torch.randn(M, ldc, device=d, generator=g)
    out2 = torch.full((M, ldc), 0.5, device=d)
    hip.gemm(M, N, K, a, lda, 1, w, 1, K, out2, ldc, cmask=gate, accumulate=True)
    close(out2[:, :N], (0.5 + want) * (gate[:, :N] > 0), what='gated +=')
    # repeats are bit-identical (tiles that reach LDS late would show here)
    again = torch.full((M, ldc), 7.0, device=d)
    for _ in range(5):
        hip.gemm(M, N, K, a, lda, 1, w, 1, K, again, ldc, bias=b, relu=True)
        assert torch.equal(again, out)


@pytest.mark.parametrize('M,N,K,ldc', [(716, 300, 300, 1200), (749, 656, 300, 656), (32, 200, 56, 200), (4096, 200, 200, 200),
                                       (4096, 56, 400, 56), (288, 600, 100, 600), (100, 44, 68, 48), (64, 64, 64, 64),
                                       (364, 300, 748, 300)])
@pytest.mark.parametrize('a_kc,b_kc', [(True, True), (True, False), (False, True), (False, False)])
def test_gemm_small_products_on_64x64_tiles(M, N, K, ldc, a_kc, b_kc):
    """Products too small to fill the chip with 128-row panels (every layer of a batch of 32, the
    per-molecule layers of a batch of 4096) run on 64 x 64 tiles: all four operand layouts, plain /
    bias + ReLU / masked input with row sums / gated accumulate, into a row-strided output that
    sits between guard zones - nothing outside the [M, N] view may change - and repeats are
    bit-identical."""
    from ps_vae_amd import hip
    if (not a_kc and M % 4) or (not b_kc and N % 4):
        pytest.skip('a 16-byte loader needs the contiguous extent to be a multiple of 4')
    d = dev()
    g = torch.Generator(device=d).manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((M, K) if a_kc else (K, M), device=d, generator=g)
    sa = (K, 1) if a_kc else (1, M)
    Bt = torch.randn((N, K) if b_kc else (K, N), device=d, generator=g)
    sb = (1, K) if b_kc else (N, 1)       # (stride along k, stride along n)
    a64, b64 = A.as_strided((M, K), sa).double(), Bt.as_strided((K, N), sb).double()
    bias = torch.randn(N, device=d, generator=g)
    amask = torch.randn(A.shape, device=d, generator=g)
    gate = torch.randn(M * ldc, device=d, generator=g)
    pad = 1024

    def run(**kw):
        buf = torch.full((pad + M * ldc + pad,), 12345.0, device=d)
        C = buf[pad:pad + M * ldc]
        hip.gemm(M, N, K, A, sa[0], sa[1], Bt, sb[0], sb[1], C, ldc, **kw)
        outside = torch.ones_like(buf, dtype=torch.bool)
        outside[pad:pad + M * ldc].as_strided((M, N), (ldc, 1)).fill_(False)
        assert bool((buf[outside] == 12345.0).all()), f'wrote outside the [M, N] view ({kw.keys()})'
        return C.as_strided((M, N), (ldc, 1)).clone()
    scale = (a64.abs() @ b64.abs())       # bounds the rounding of any summation order
    want = a64 @ b64

    def near(got, ref, what):
        err = ((got.double() - ref).abs() / scale.clamp_min(1e-30)).max().item()
        assert err < 2e-6, f'{what}: {err:.2e} of sum |a||b|'
    plain = run()
    near(plain, want, 'plain')
    assert torch.equal(run(), plain) and torch.equal(run(), plain), 'repeats differ'
    near(run(bias=bias, relu=True), torch.relu(want + bias.double()), 'bias + relu')
    if not b_kc:                          # the row sums ride on a virtual ones column of an n-contiguous B
        rs = torch.full((M,), 3.0, device=d)
        am64 = a64 * (amask.as_strided((M, K), sa) > 0)
        got = run(amask=amask, rowsum=rs)
        err = ((got.double() - am64 @ b64).abs() / scale.clamp_min(1e-30)).max().item()
        assert err < 2e-6, f'masked: {err:.2e}'
        close(rs, am64.sum(1), rel=1e-5, what='row sums of the masked operand')
    buf = torch.full((M * ldc,), 0.25, device=d)
    hip.gemm(M, N, K, A, sa[0], sa[1], Bt, sb[0], sb[1], buf, ldc, cmask=gate, accumulate=True)
    gated = (0.25 + want) * (gate.as_strided((M, N), (ldc, 1)) > 0)
    near(buf.as_strided((M, N), (ldc, 1)), gated, 'gated +=')
    if ldc > N:
        rest = buf.as_strided((M, ldc - N), (ldc, 1), N)
        assert bool((rest == 0.25).all()), 'columns beyond N were written'


@pytest.mark.parametrize('M,N,K,ldc', [(749, 300, 300, 300), (20000, 300, 300, 300), (100, 44, 68, 44), (64, 64, 64, 64),
                                       (4096, 200, 200, 200), (33, 300, 100, 1200), (9000, 328, 328, 328), (650, 300, 300, 300),
                                       (95009, 300, 300, 300)])
def test_gemm_relu_bits_and_bit_gate(M, N, K, ldc):
    """psvae_gemm_f32_bits: the forward Linear + ReLU leaves (result > 0) behind as bits in the
    documented layout, with a result bit-identical to the plain call; the input-gradient product
    gated by those bits is bit-identical to the one gated by the full activations (cmask)."""
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator(device=d).manual_seed(M + N)
    x = torch.randn(M, K, device=d, generator=g)
    w = torch.randn(N, K, device=d, generator=g) * 0.1
    b = torch.randn(N, device=d, generator=g)
    words = hip.gemm_bits_words(M, N)
    guard = torch.full((words + 4096,), -1, dtype=torch.int64, device=d)     # nothing behind the array may be written
    bits = guard[:words]
    hbuf = torch.full((M, ldc), 9.0, device=d)
    hip.gemm_bits(M, N, K, x, K, 1, w, 1, K, hbuf, ldc, bias=b, relu=True, relu_bits=bits)
    plain = torch.full((M, ldc), 9.0, device=d)
    hip.gemm(M, N, K, x, K, 1, w, 1, K, plain, ldc, bias=b, relu=True, split_k=1)
    assert torch.equal(hbuf, plain)
    # decode: word ((r/32) * NG + c/32) * 16 + 4 i + comp, bit l -> row 32 (r/32) + (l >> 3) + 8 i, col 32 (c/32) + 4 (l & 7) + comp
    ng = (N + 31) // 32
    rows = torch.arange(M, device=d)[:, None].expand(M, N)
    cols = torch.arange(N, device=d)[None, :].expand(M, N)
    i, lrow = (rows % 32) // 8, (rows % 32) % 8
    comp, lcol = (cols % 32) % 4, (cols % 32) // 4
    word = ((rows // 32) * ng + cols // 32) * 16 + 4 * i + comp
    lane = lrow * 8 + lcol
    got = (bits[word] >> lane) & 1
    assert torch.equal(got.bool(), hbuf[:, :N] > 0), 'bit layout'
    assert bool((guard[words:] == -1).all()), 'wrote behind the bit array'
    # gradient through the ReLU: gate by bits == gate by the activations
    gy = torch.randn(M, N, device=d, generator=g)
    w2 = torch.randn(N, N, device=d, generator=g) * 0.1
    by_bits = torch.full((M, ldc), 3.0, device=d)
    by_mask = torch.full((M, ldc), 3.0, device=d)
    hip.gemm_bits(M, N, N, gy, N, 1, w2, N, 1, by_bits, ldc, gate_bits=bits)
    hip.gemm(M, N, N, gy, N, 1, w2, N, 1, by_mask, ldc, cmask=hbuf, split_k=1)
    assert torch.equal(by_bits, by_mask)
    close(by_bits[:, :N], (gy.double() @ w2.double()) * (hbuf[:, :N] > 0), what='gated input gradient')


def test_gemm_random_shapes_and_forms():
    """Eighty seeded random products across the dispatch boundaries (64 x 64 tiles / 128-row panels /
    2 x 2 wave grid / scalar-load kernel, automatic and explicit K splits): random operand layouts,
    row-strided outputs, bias / ReLU / masked operand / row sums / gate / accumulate in random
    combination, every result against float64 relative to sum |a||b| and the memory around the
    [M, N] view untouched."""
    from ps_vae_amd import hip
    d = dev()
    rng = np.random.default_rng(2024)
    g = torch.Generator(device=d).manual_seed(11)
    for case in range(80):
        M = int(rng.choice([1, 5, 31, 32, 33, 64, 100, 128, 300, 364, 513, 749, 1500, 4096, 9000]))
        N = int(rng.choice([1, 2, 4, 44, 56, 64, 100, 200, 300, 302, 328, 400, 656]))
        K = int(rng.choice([1, 2, 11, 32, 50, 64, 100, 204, 300, 328, 656, 1200, 3000]))
        a_kc, b_kc = bool(rng.integers(2)), bool(rng.integers(2))
        ldc = N + int(rng.choice([0, 0, 4, 900]))
        lda = (K if a_kc else M) + int(rng.choice([0, 0, 4, 12]))
        ldb = (K if b_kc else N) + int(rng.choice([0, 0, 8]))
        A = torch.randn((M, lda) if a_kc else (K, lda), device=d, generator=g)
        Bt = torch.randn((N, ldb) if b_kc else (K, ldb), device=d, generator=g)
        sa = (lda, 1) if a_kc else (1, lda)
        sb = (1, ldb) if b_kc else (ldb, 1)
        kw = {}
        if rng.integers(2):
            kw['bias'] = torch.randn(N, device=d, generator=g)
        if rng.integers(3) == 0:
            kw['relu'] = True
        if rng.integers(3) == 0:
            kw['amask'] = torch.randn(A.shape, device=d, generator=g)
        if rng.integers(3) == 0 and 'bias' not in kw and 'relu' not in kw:
            kw['rowsum'] = torch.full((M,), 2.0, device=d)
        if rng.integers(3) == 0:
            kw['cmask'] = torch.randn(M * ldc, device=d, generator=g)
        acc = bool(rng.integers(4) == 0)
        kw['split_k'] = int(rng.choice([0, 0, 0, 1, 2, 3, 7]))
        pad = 512
        buf = torch.full((pad + M * ldc + pad,), 0.75, device=d)
        C = buf[pad:pad + M * ldc]
        a64 = A.as_strided((M, K), sa).double()
        if 'amask' in kw:
            a64 = a64 * (kw['amask'].as_strided((M, K), sa) > 0)
        b64 = Bt.as_strided((K, N), sb).double()
        want = a64 @ b64
        if 'bias' in kw:
            want = want + kw['bias'].double()
        if kw.get('relu'):
            want = want.clamp_min(0)
        if acc:
            want = want + 0.75
        if 'cmask' in kw:
            want = want * (kw['cmask'].as_strided((M, N), (ldc, 1)) > 0)
        hip.gemm(M, N, K, A, sa[0], sa[1], Bt, sb[0], sb[1], C, ldc, accumulate=acc, **kw)
        what = f'case {case}: M={M} N={N} K={K} a_kc={a_kc} b_kc={b_kc} lda={lda} ldb={ldb} ldc={ldc} acc={acc} ' \
               f'{ {k: (v if not torch.is_tensor(v) else "yes") for k, v in kw.items()} }'
        scale = (a64.abs() @ b64.abs()) + 1.0
        err = ((C.as_strided((M, N), (ldc, 1)).double() - want).abs() / scale).max().item()
        assert err < 3e-6, f'{what}: {err:.2e} of sum |a||b| + 1'
        outside = torch.ones_like(buf, dtype=torch.bool)
        outside[pad:pad + M * ldc].as_strided((M, N), (ldc, 1)).fill_(False)
        assert bool((buf[outside] == 0.75).all()), f'{what}: wrote outside the [M, N] view'
        if 'rowsum' in kw:
            want_rs = a64.sum(1) + (2.0 if acc else 0.0)
            rs_err = ((kw['rowsum'].double() - want_rs).abs() / (a64.abs().sum(1) + 1.0)).max().item()
            assert rs_err < 3e-6, f'{what}: row sums {rs_err:.2e}'


def test_gemm_tall_k_repeats_bit_for_bit():
    """The stacked conv-weight gradient of the benchmark (300 x 300 over 380 592 rows) forty times
    in a row: every result equals the first bit for bit.  The kernel's k-tiles reach LDS through
    loads that no barrier waits for by itself; a tile read before it has landed shows up as a
    result that differs by one k-step's products, about 1e-5 of an entry - below what a
    comparison with float64 notices, so the repeats are compared exactly."""
    from ps_vae_amd import hip
    d = dev()
    K = 380592
    g = torch.Generator(device=d).manual_seed(77)
    dy = torch.randn(K, 300, device=d, generator=g)
    x = torch.randn(K, 300, device=d, generator=g)
    first = torch.empty(300, 300, device=d)
    fb = torch.empty(300, device=d)
    hip.gemm(300, 300, K, dy, 1, 300, x, 300, 1, first, 300, split_k=0, rowsum=fb)
    close(first, (dy.double().t() @ x.double()), what='dW')
    outs = [(torch.empty(300, 300, device=d), torch.empty(300, device=d)) for _ in range(40)]
    for gw, gb in outs:
        hip.gemm(300, 300, K, dy, 1, 300, x, 300, 1, gw, 300, split_k=0, rowsum=gb)
    torch.cuda.synchronize()
    bad = [i for i, (gw, gb) in enumerate(outs) if not (torch.equal(gw, first) and torch.equal(gb, fb))]
    assert not bad, f'repeats {bad} differ from the first result'


@pytest.mark.parametrize('rows,K', [(290, 300), (749, 300), (290, 1160), (33, 56)])
def test_gemm_small_batch_products_repeat_bit_for_bit(rows, K):
    """The three products of a Linear at the sizes of a batch of 32 (the latency form of the kernel: 32 x 32 tiles
    whose four waves split K, k-tiles requested five ahead through two LDS stages with one barrier per tile,
    partial tiles summed through LDS) against float64 and thirty times in a row: a tile read before it was
    written, or partial sums added in a varying order, would change low bits that the float64 comparison cannot see."""
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator(device=d).manual_seed(rows + K)
    N = 300
    x = torch.randn(rows, K, device=d, generator=g)
    w = torch.randn(N, K, device=d, generator=g) * 0.05
    b = torch.randn(N, device=d, generator=g)
    gy = torch.randn(rows, N, device=d, generator=g)

    def run():
        y, gx, gw, gb = (torch.empty(rows, N, device=d), torch.empty(rows, K, device=d), torch.empty(N, K, device=d),
                         torch.empty(N, device=d))
        hip.gemm(rows, N, K, x, K, 1, w, 1, K, y, N, bias=b, relu=True)
        hip.gemm(rows, K, N, gy, N, 1, w, K, 1, gx, K)
        hip.gemm(N, K, rows, gy, 1, N, x, K, 1, gw, K, split_k=0, rowsum=gb)
        return y, gx, gw, gb

    first = run()
    close(first[0], torch.relu(x.double() @ w.double().t() + b.double()), what='y')
    close(first[1], gy.double() @ w.double(), what='dx')
    close(first[2], gy.double().t() @ x.double(), what='dw')
    close(first[3], gy.double().sum(0), what='db')
    for i in range(30):
        again = run()
        assert all(torch.equal(a, f) for a, f in zip(again, first)), f'repeat {i} differs from the first result'


@pytest.mark.parametrize('rows', [40, 4000, 50000])
def test_gemm_with_an_empty_contraction(rows):
    """K = 0 (operands without storage): the result is act(bias) in every kernel the row count selects."""
    from ps_vae_amd import hip
    d = dev()
    N = 300
    x, w = torch.empty(rows, 0, device=d), torch.empty(N, 0, device=d)
    b = torch.randn(N, device=d)
    y = torch.full((rows, N), float('nan'), device=d)
    hip.gemm(rows, N, 0, x, 0, 1, w, 1, 0, y, N, bias=b, relu=True)
    assert torch.equal(y, torch.relu(b).expand(rows, N))


def test_gemm_large_split_k():
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator().manual_seed(9)
    M = 20000
    gy = torch.randn(M, 300, generator=g)
    x = torch.randn(M, 200, generator=g)
    gw = torch.empty(300, 200, device=d)
    hip.gemm(300, 200, M, gy.to(d), 1, 300, x.to(d), 200, 1, gw, 200, split_k=0)
    close(gw, gy.double().t() @ x.double(), what='dW K=20000')


def test_colsum():
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(5000, 328, generator=g)
    m = torch.randn(5000, 328, generator=g)
    out = torch.ones(328, device=d)
    hip.colsum(x.to(d), out, mask=m.to(d), accumulate=True)
    close(out, 1.0 + (x.double() * (m > 0)).sum(0), what='masked colsum')
    out2 = torch.empty(328, device=d)
    hip.colsum(x.to(d), out2)
    close(out2, x.double().sum(0), what='colsum')


# ------------------------------------------------------------------ Linear autograd
@pytest.mark.parametrize('relu', [False, True])
def test_linear_autograd(relu):
    from ps_vae_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(777, 200, generator=g, requires_grad=True)
    w = (torch.randn(300, 200, generator=g) * 0.1).requires_grad_()
    b = torch.randn(300, generator=g, requires_grad=True)
    gy = torch.randn(777, 300, generator=g)
    y = torch.nn.functional.linear(x.double(), w.double(), b.double())
    y = torch.relu(y) if relu else y
    y.backward(gy.double())
    xd, wd, bd = (t.detach().to(d).requires_grad_() for t in (x, w, b))
    yd = ops.linear(xd, wd, bd, relu=relu)
    yd.backward(gy.to(d))
    close(yd, y, what='y')
    close(xd.grad, x.grad, what='dx')
    close(wd.grad, w.grad, what='dw')
    close(bd.grad, b.grad, what='db')


# ------------------------------------------------------------------ graphs for the GNN kernels
def random_graph(V, H, C, seed, max_deg=4):
    rng = np.random.default_rng(seed)
    src, dst, cls = [], [], []
    for i in range(V):
        for _ in range(rng.integers(0, max_deg + 1)):
            j = int(rng.integers(max(0, i - 20), min(V, i + 20)))
            src.append(j), dst.append(i), cls.append(int(rng.integers(0, C)))
    ei = torch.tensor([src, dst], dtype=torch.long).reshape(2, -1)
    attr = torch.nn.functional.one_hot(torch.tensor(cls, dtype=torch.long), C)
    g = torch.Generator().manual_seed(seed)
    return ei, attr, torch.randn(V, H, generator=g), torch.randn(C, H, generator=g) * 0.5


@pytest.mark.parametrize('V,H,C', [(1, 4, 1), (50, 36, 4), (700, 300, 4), (333, 128, 7)])
def test_gine_aggregate_fwd_bwd(V, H, C):
    from ps_vae_amd import ops
    d = dev()
    ei, attr, x, table = random_graph(V, H, C, seed=V + H)
    x.requires_grad_(), table.requires_grad_()
    e = attr.float() @ table
    out = model_ref.gine_aggregate(x, ei[0], ei[1], e)
    gy = torch.randn(V, H, generator=torch.Generator().manual_seed(1))
    out.backward(gy)
    csr = ops.Csr.from_edge_index(ei.to(d), attr.to(d), V)
    xd, td = x.detach().to(d).requires_grad_(), table.detach().to(d).requires_grad_()
    outd = ops.gine_aggregate(xd, td, csr)
    outd.backward(gy.to(d))
    close(outd, out, what='aggregate')
    close(xd.grad, x.grad, what='d aggregate / dx')
    close(td.grad, table.grad, what='d aggregate / dtable')


@pytest.mark.parametrize('V,H,C', [(1, 4, 1), (50, 36, 4), (5000, 300, 4), (333, 128, 7)])
def test_gine_aggregate_with_message_bits(V, H, C):
    """psvae_gine_aggregate_fwd_bits / _bwd_bits: the same results as the plain pair, bit for bit (the
    bits ARE the comparisons the plain backward recomputes from x and the table), and the per-graph
    add rows equal the same rows spread to the nodes."""
    from ps_vae_amd import hip, ops
    d = dev()
    ei, attr, x, table = random_graph(V, H, C, seed=V)
    g = torch.Generator().manual_seed(V + 1)
    gy, add = torch.randn(V, H, generator=g).to(d), torch.randn(V, H, generator=g).to(d)
    xd, td = x.to(d), table.to(d)
    csr = ops.Csr.from_edge_index(ei.to(d), attr.to(d), V)
    want_out, got_out = torch.empty(V, H, device=d), torch.empty(V, H, device=d)
    words, wtype = hip.gine_mask_words(V, H, C)
    mask = torch.zeros(words, dtype=wtype, device=d)
    hip.gine_aggregate_fwd(xd, td, csr.row_ptr, csr.col, csr.cls, want_out)
    hip.gine_aggregate_fwd_bits(xd, td, csr.row_ptr, csr.col, csr.cls, got_out, mask)
    assert torch.equal(got_out, want_out)
    # the bits themselves: element e of column block c of node j under class k
    pre = (x[:, None, :] + table[None, :, :]) > 0                                     # [V, C, H]
    e = torch.arange(H)
    weights = (1 << (4 * torch.arange(C)[:, None] + (e % 4)[None, :])).to(torch.int64)   # [C, H]
    want_bits = (pre.to(torch.int64) * weights[None]).sum(1).view(V, H // 4, 4).sum(-1)  # [V, H/4]
    got_bits = mask.cpu().to(torch.int64) & ((1 << (4 * C)) - 1)      # (bits of classes beyond C are not defined)
    assert torch.equal(got_bits.view(V, H // 4), want_bits)
    want_gx, got_gx = torch.empty(V, H, device=d), torch.empty(V, H, device=d)
    want_gt, got_gt = torch.full((C, H), 0.5, device=d), torch.full((C, H), 0.5, device=d)
    hip.gine_aggregate_bwd(xd, td, gy, csr.t_row_ptr, csr.t_col, csr.t_cls, want_gx, want_gt, add=add, accumulate_table=True)
    hip.gine_aggregate_bwd_bits(mask, C, gy, csr.t_row_ptr, csr.t_col, csr.t_cls, got_gx, got_gt, add=add,
                                accumulate_table=True)
    assert torch.equal(got_gx, want_gx) and torch.equal(got_gt, want_gt)
    # per-graph rows: graphs of 1..7 nodes, a column block of a wider matrix as the row source
    sizes = []
    while sum(sizes) < V:
        sizes.append(min(int(torch.randint(1, 8, (1,), generator=g)), V - sum(sizes)))
    node_graph = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)).to(torch.int32).to(d)
    wide = torch.randn(len(sizes), 2 * H + 4, generator=g).to(d)
    seg = wide[:, 4:4 + H]
    hip.gine_aggregate_bwd(xd, td, gy, csr.t_row_ptr, csr.t_col, csr.t_cls, want_gx, want_gt,
                           add=seg[node_graph.long()].contiguous())
    hip.gine_aggregate_bwd_bits(mask, C, gy, csr.t_row_ptr, csr.t_col, csr.t_cls, got_gx, got_gt, seg_add=seg,
                                node_graph=node_graph)
    assert torch.equal(got_gx, want_gx) and torch.equal(got_gt, want_gt)


@pytest.mark.parametrize('V,N,B', [(1, 4, 1), (95, 300, 7), (4100, 328, 300), (64, 32, 64)])
def test_graph_rows_gated(V, N, B):
    """out[v] = q[graph(v)] where the product's sign bit is set: the bits come from a real
    psvae_gemm_f32_bits call, the expectation from its full-width result."""
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator().manual_seed(V + N)
    node_graph = torch.sort(torch.randint(0, B, (V,), generator=g)).values.to(torch.int32).to(d)
    q = torch.randn(B, N + 4, generator=g).to(d)[:, :N]
    if V >= 32:
        a, w = torch.randn(V, 16, generator=g).to(d), torch.randn(N, 16, generator=g).to(d)
        act = torch.empty(V, N, device=d)
        bits = torch.zeros(hip.gemm_bits_words(V, N), dtype=torch.int64, device=d)
        hip.gemm_bits(V, N, 16, a, 16, 1, w, 1, 16, act, N, relu=True, relu_bits=bits)
        keep = act > 0
    else:   # below the bit product's row minimum: bits set by hand in its layout
        keep = torch.rand(V, N, generator=g).to(d) > 0.5
        words = torch.zeros(hip.gemm_bits_words(V, N), dtype=torch.int64)
        kc = keep.cpu()
        for r in range(V):
            for c in range(N):
                if kc[r, c]:
                    idx = ((r // 32) * ((N + 31) // 32) + c // 32) * 16 + 4 * ((r % 32) // 8) + (c % 4)
                    lane = 8 * (r % 8) + (c % 32) // 4
                    words[idx] |= (1 << lane) if lane < 63 else -(1 << 63)
        bits = words.to(d)
    out = torch.full((V, N + 4), 9.0, device=d)
    hip.graph_rows_gated(q, node_graph, bits, out[:, :N])
    want = torch.where(keep, q[node_graph.long()], torch.zeros((), device=d))
    assert torch.equal(out[:, :N], want) and bool((out[:, N:] == 9.0).all())


@pytest.mark.parametrize('B,H,t', [(300, 300, 4), (5, 36, 2), (40, 300, 1)])
def test_gine_encoder_pooled_backward(B, H, t):
    """ops.gine_encoder(..., pool=segments) - the sum over each graph's nodes taken inside the node,
    its backward working from one gradient row per graph - against the same encoder followed by
    ops.segment_sum: equal outputs, gradients of every input within 1e-5."""
    from ps_vae_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(B + H)
    sizes = torch.randint(1, 30, (B,), generator=g)
    V = int(sizes.sum())
    off = torch.zeros(B + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(sizes, 0)
    ei, attr, x, table = random_graph(V, H, 4, seed=B)
    csr = ops.Csr.from_edge_index(ei.to(d), attr.to(d), V)
    segs = ops.Segments.from_offsets(off.to(d))
    w1, w2 = torch.randn(H, H, generator=g) / H ** 0.5, torch.randn(H, H, generator=g) / H ** 0.5
    b1, b2 = torch.randn(H, generator=g) * 0.1, torch.randn(H, generator=g) * 0.1
    gout = torch.randn(B, t * H, generator=g).to(d)

    def run(pool):
        leaves = [v.clone().to(d).requires_grad_(True) for v in (x, table, w1, b1, w2, b2)]
        xs, tb, a1, c1, a2, c2 = leaves
        if pool:
            out = ops.gine_encoder(xs, None, None, tb, a1, c1, a2, c2, csr, t, False, segs)
        else:
            out = ops.segment_sum(ops.gine_encoder(xs, None, None, tb, a1, c1, a2, c2, csr, t), segs)
        out.backward(gout)
        return out.detach(), [v.grad for v in leaves]
    out_a, grads_a = run(False)
    out_b, grads_b = run(True)
    assert torch.equal(out_a, out_b)
    for ga, gb, name in zip(grads_a, grads_b, ('x', 'table', 'w1', 'b1', 'w2', 'b2')):
        close(gb, ga, what=f'pooled encoder: d/d{name}')


def test_gine_aggregate_strided_rows():
    from ps_vae_amd import hip, ops
    d = dev()
    V, H, C = 200, 300, 4
    ei, attr, x, table = random_graph(V, H, C, seed=3)
    want = model_ref.gine_aggregate(x, ei[0], ei[1], attr.float() @ table)
    csr = ops.Csr.from_edge_index(ei.to(d), attr.to(d), V)
    wide = torch.zeros(V, 1200, device=d)
    wide[:, 300:600] = x.to(d)
    hip.gine_aggregate_fwd(wide[:, 300:600], table.to(d), csr.row_ptr, csr.col, csr.cls, wide[:, 600:900])
    close(wide[:, 600:900], want, what='aggregate into a slot')


# ------------------------------------------------------------------ small fused ops
def test_embed_atom_fwd_bwd():
    from ps_vae_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(4)
    V = 3000
    ea, ep, eq = torch.randn(11, 50, generator=g), torch.randn(302, 100, generator=g), torch.randn(50, 50, generator=g)
    x = torch.randint(0, 11, (V,), generator=g)
    xp = torch.randint(0, 302, (V,), generator=g)
    xq = torch.randint(0, 50, (V,), generator=g)
    for t in (ea, ep, eq):
        t.requires_grad_()
    want = torch.cat([ea[x], ep[xp], eq[xq]], dim=-1)
    gy = torch.randn(V, 200, generator=g)
    want.backward(gy)
    tabs = [t.detach().to(d).requires_grad_() for t in (ea, ep, eq)]
    got = ops.embed_atom(x.to(d).int(), xp.to(d).int(), xq.to(d).int(), *tabs)
    got.backward(gy.to(d))
    assert torch.equal(got.cpu(), want.detach())
    for t, r, name in zip(tabs, (ea, ep, eq), ('atom', 'piece', 'pos')):
        close(t.grad, r.grad, what=f'd embed / d {name} table')


@pytest.mark.parametrize('V,H,sizes', [(1, 4, (2, 3, 1)), (700, 300, (11, 302, 50)), (5001, 36, (11, 302, 50)),
                                        (40000, 300, (11, 302, 50))])
def test_embed_trans_fwd(V, H, sizes):
    """Gather-sum of three table rows per atom against float64 index arithmetic."""
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator().manual_seed(V + H)
    na, npc, nq = sizes
    x = torch.randint(0, na, (V,), generator=g, dtype=torch.int32)
    xp = torch.randint(0, npc, (V,), generator=g, dtype=torch.int32)
    xq = torch.randint(0, nq, (V,), generator=g, dtype=torch.int32)
    T = torch.randn(na + npc + nq, H, generator=g)
    out = torch.empty(V, H, device=d)
    hip.embed_trans_fwd(x.to(d), xp.to(d), xq.to(d), sizes, T.to(d), out)
    T64 = T.double()
    want = T64[x.long()] + T64[na + xp.long()] + T64[na + npc + xq.long()]
    close(out, want, what='embed_trans forward')


def test_embed_trans_equals_node_trans_of_embedded_rows():
    """ops.embed_trans(ids, W, b) == Linear(W, b)(embed_atom(ids)) and so are the gradients of
    the three tables, W and b (float64 yardstick)."""
    from ps_vae_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(77)
    V, H, dims, sizes = 3000, 300, (50, 100, 50), (11, 302, 50)
    ids = [torch.randint(0, n, (V,), generator=g, dtype=torch.int32) for n in sizes]
    tabs = [torch.randn(n, dd, generator=g) for n, dd in zip(sizes, dims)]
    W = torch.randn(H, sum(dims), generator=g) * 0.1
    b = torch.randn(H, generator=g)
    gout = torch.randn(V, H, generator=g)
    leaves = [t.clone().to(d).requires_grad_(True) for t in tabs + [W, b]]
    out = ops.embed_trans(ops.EmbeddedIds(*[i.to(d) for i in ids], *leaves[:3]), leaves[3], leaves[4])
    out.backward(gout.to(d))
    ref = [t.clone().double().requires_grad_(True) for t in tabs + [W, b]]
    rows = torch.cat([ref[k][ids[k].long()] for k in range(3)], dim=1)
    want = rows @ ref[3].t() + ref[4]
    want.backward(gout.double())
    close(out, want, what='embed_trans vs Linear(embed_atom)')
    for got, w, name in zip(leaves, ref, ['E_atom', 'E_piece', 'E_pos', 'W', 'b']):
        close(got.grad, w.grad, what=f'd/d{name}')


def test_segment_sum_fwd_bwd():
    from ps_vae_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(6)
    lens = [3, 0, 17, 38, 1, 9]
    off = np.concatenate([[0], np.cumsum(lens)])
    V, W = int(off[-1]), 1200
    x = torch.randn(V, W, generator=g, requires_grad=True)
    gid = torch.repeat_interleave(torch.arange(len(lens)), torch.tensor(lens))
    want = torch.zeros(len(lens), W).index_add_(0, gid, x)
    gy = torch.randn(len(lens), W, generator=g)
    want.backward(gy)
    segs = ops.Segments.from_offsets(torch.tensor(off, dtype=torch.int32, device=d))
    xd = x.detach().to(d).requires_grad_()
    got = ops.segment_sum(xd, segs)
    got.backward(gy.to(d))
    close(got, want, what='segment sum')
    close(xd.grad, x.grad, what='segment sum backward')


def test_rsample_kl_fwd_bwd():
    from ps_vae_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(8)
    B, L = 257, 56
    mean = torch.randn(B, L, generator=g, requires_grad=True)
    pre = torch.randn(B, L, generator=g, requires_grad=True)
    eps = torch.randn(B, L, generator=g)
    lv = -torch.abs(pre)
    kl = -0.5 * torch.sum(1.0 + lv - mean * mean - torch.exp(lv)) / B
    z = mean + torch.exp(lv / 2) * eps
    gz = torch.randn(B, L, generator=g)
    (z * gz).sum().add(kl * 0.37).backward()
    md, pd = mean.detach().to(d).requires_grad_(), pre.detach().to(d).requires_grad_()
    zd, kld = ops.rsample_kl(md, pd, eps.to(d))
    ((zd * gz.to(d)).sum() + kld * 0.37).backward()
    close(zd, z, what='z')
    close(kld, kl, what='kl')
    close(md.grad, mean.grad, what='d/dmean')
    close(pd.grad, pre.grad, what='d/dlogvar')


@pytest.mark.parametrize('F,max_nodes', [(328, 40), (328, 12), (64, 40), (1024, 9)])
def test_pair_combine_per_graph_equals_the_flat_kernel(F, max_nodes):
    """psvae_pair_combine_fwd_graphs (a graph's node rows staged in LDS) against psvae_pair_combine_fwd on all atom
    pairs m < k of molecules of 0 .. 30 atoms, bit for bit - also for graphs larger than the staging area
    (max_nodes 12 and 9), which the kernel serves from memory, and molecules without pairs."""
    from ps_vae_amd import hip
    d = dev()
    rng = np.random.default_rng(F + max_nodes)
    sizes = [0, 1, 2, 30, 17, 1, 25, 8, 0, 13] + [int(v) for v in rng.integers(2, 31, size=40)]
    node_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    src, dst, graph, ptr = [], [], [], [0]
    for g, n in enumerate(sizes):
        for m in range(n):
            for k in range(m + 1, n):
                src.append(node_off[g] + m), dst.append(node_off[g] + k), graph.append(g)
        ptr.append(len(src))
    V, B, P = int(node_off[-1]), len(sizes), len(src)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.int32), device=d)   # noqa: E731
    gen = torch.Generator(device=d).manual_seed(F)
    ab = torch.randn(V, 2 * F, device=d, generator=gen)
    c = torch.randn(B, F, device=d, generator=gen)
    want = torch.empty(P, F, device=d)
    hip.pair_combine_fwd(ab[:, :F], ab[:, F:], c, t(src), t(dst), t(graph), want, relu=True)
    got = torch.full((P, F), float('nan'), device=d)
    hip.pair_combine_fwd_graphs(ab[:, :F], ab[:, F:], c, t(src), t(dst), t(ptr), t(node_off[:-1]), t(node_off[1:]), max_nodes,
                                got, relu=True)
    assert torch.equal(got, want)


def test_block_table_against_the_slice_assignments():
    """_BlockTable (one launch) against the block matrix written slice by slice, and its gradient views."""
    from ps_vae_amd import ops
    d = dev()
    torch.manual_seed(5)
    ea = torch.randn(11, 50, device=d, requires_grad=True)
    ep = torch.randn(302, 100, device=d, requires_grad=True)
    eq = torch.randn(50, 50, device=d, requires_grad=True)
    Z = ops._BlockTable.apply(ea, ep, eq)
    want = torch.zeros(363, 204, device=d)
    want[:11, :50], want[11:313, 50:150], want[313:, 150:200] = ea.detach(), ep.detach(), eq.detach()
    want[:11, 200] = 1.0
    assert torch.equal(Z, want)
    g = torch.randn_like(Z)
    Z.backward(g)
    assert torch.equal(ea.grad, g[:11, :50]) and torch.equal(ep.grad, g[11:313, 50:150]) and torch.equal(eq.grad, g[313:, 150:200])


@pytest.mark.parametrize('B,sel', [(32, [0, 2]), (4096, [0, 1, 2]), (1, [1])])
def test_mse_selected_against_mse_loss(B, sel):
    """ops.mse_selected against nn.MSELoss on the selected property columns (src/pl_models/ps_vae_model.py:55-57)."""
    from ps_vae_amd import ops
    d = dev()
    torch.manual_seed(B)
    pred = torch.randn(B, len(sel), device=d, requires_grad=True)
    props = torch.randn(B, 3, device=d)
    idx = torch.tensor(sel, dtype=torch.int64, device=d)
    loss = ops.mse_selected(pred, props, idx)
    (loss * 0.7).backward()
    ref = pred.detach().double().clone().requires_grad_(True)
    want = torch.nn.functional.mse_loss(ref, props.double().index_select(1, idx))
    (want * 0.7).backward()
    close(loss, want.float(), what='mse')
    close(pred.grad, ref.grad.float(), what='d/dpred')


@pytest.mark.parametrize('beta_on_device', [False, True])
def test_loss_mix_rounds_as_the_expression(beta_on_device):
    """ops.loss_mix against alpha * rec + (1 - alpha) * pred + beta * kl (src/pl_models/ps_vae_model.py:71-74):
    the same bits (every product and sum is rounded on its own in both) and the same three gradients."""
    from ps_vae_amd import ops
    d = dev()
    torch.manual_seed(9)
    alpha, beta = 0.1, 0.0003
    for _ in range(20):
        leaves = [torch.rand((), device=d).mul_(5).requires_grad_(True) for _ in range(3)]
        twins = [v.detach().clone().requires_grad_(True) for v in leaves]
        b = torch.tensor(beta, device=d) if beta_on_device else beta
        got = ops.loss_mix(leaves[0], leaves[1], leaves[2], alpha, b)
        want = alpha * twins[0] + (1 - alpha) * twins[1] + b * twins[2]
        assert torch.equal(got, want.detach())
        (got * 3.0).backward()
        (want * 3.0).backward()
        for v, w in zip(leaves, twins):
            assert torch.equal(v.grad, w.grad)


@pytest.mark.parametrize('C,J,K,bias', [(4, 656, 328, True), (3, 70, 33, True), (8, 40, 100, False)])
def test_affine_merge(C, J, K, bias):
    """W_last (W_out x + b_out) + b_last as one affine map, and the gradients of both layers."""
    from ps_vae_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(C * 100 + J)
    wl = torch.randn(C, J, generator=g, dtype=torch.float64, requires_grad=True)
    wo = torch.randn(J, K, generator=g, dtype=torch.float64, requires_grad=True)
    bl = torch.randn(C, generator=g, dtype=torch.float64, requires_grad=True) if bias else None
    bo = torch.randn(J, generator=g, dtype=torch.float64, requires_grad=True) if bias else None
    gw = torch.randn(C, K, generator=g, dtype=torch.float64)
    gb = torch.randn(C, generator=g, dtype=torch.float64)
    w_ref = wl @ wo
    b_ref = (wl @ bo + bl) if bias else torch.zeros(C, dtype=torch.float64)
    ((w_ref * gw).sum() + ((b_ref * gb).sum() if bias else 0.0)).backward()

    leaves = [t.detach().float().to(d).requires_grad_(True) if t is not None else None for t in (wl, bl, wo, bo)]
    w_eff, b_eff = ops.affine_merge(*leaves)
    close(w_eff, w_ref, what='W_eff')
    close(b_eff, b_ref, what='b_eff')
    ((w_eff * gw.float().to(d)).sum() + (b_eff * gb.float().to(d)).sum()).backward()
    for got, want, name in zip(leaves, (wl, bl, wo, bo), ('W_last', 'b_last', 'W_out', 'b_out')):
        if want is not None:
            close(got.grad, want.grad, what=f'grad {name}')


@pytest.mark.parametrize('T,B,E,H', [(1, 3, 100, 200), (7, 300, 100, 200), (9, 33, 100, 200), (5, 17, 12, 20),
                                     (4, 64, 8, 4), (3, 40, 16, 256), (3, 9, 16, 260)])
def test_gru_fwd_bwd(T, B, E, H):
    """The GRU of the piece language model against float64 nn.GRU: the one-launch recurrence
    (H a multiple of 4 up to 256: full and partial blocks of 16 hidden units, batch sizes that
    are no multiple of 16) and the per-step kernels (H = 260)."""
    from ps_vae_amd import ops
    d = dev()
    torch.manual_seed(T * 100 + B)
    ref = torch.nn.GRU(E, H, batch_first=True).double()
    x = torch.randn(B, T, E, dtype=torch.float64, requires_grad=True)
    h0 = torch.randn(B, H, dtype=torch.float64, requires_grad=True)
    out, _ = ref(x, h0.unsqueeze(0))
    gy = torch.randn(B, T, H, dtype=torch.float64)
    out.backward(gy)
    params = [p.detach().float().to(d).requires_grad_() for p in
              (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)]
    xd = x.detach().float().transpose(0, 1).contiguous().to(d).requires_grad_()
    hd = h0.detach().float().to(d).requires_grad_()
    hs = ops.gru(xd, hd, *params)                     # [T,B,H]
    hs.backward(gy.float().transpose(0, 1).contiguous().to(d))
    close(hs.transpose(0, 1), out, what='GRU hidden states')
    close(xd.grad.transpose(0, 1), x.grad, what='d GRU / dx')
    close(hd.grad, h0.grad, what='d GRU / dh0')
    for p, r, name in zip(params, (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0),
                          ('w_ih', 'w_hh', 'b_ih', 'b_hh')):
        close(p.grad, r.grad, what=f'd GRU / d {name}')


def test_embedding_fwd_bwd():
    from ps_vae_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(4)
    table = torch.randn(302, 100, generator=g, dtype=torch.float64, requires_grad=True)
    ids = torch.randint(0, 302, (11, 700), generator=g)
    gy = torch.randn(11, 700, 100, generator=g, dtype=torch.float64)
    out = torch.nn.functional.embedding(ids, table)
    out.backward(gy)
    td = table.detach().float().to(d).requires_grad_()
    outd = ops.embedding(ids.to(torch.int32).to(d), td)
    outd.backward(gy.float().to(d))
    assert torch.equal(outd.cpu(), out.detach().float())
    close(td.grad, table.grad, what='embedding table gradient')


@pytest.mark.parametrize('V,sizes,H', [(0, (11, 302, 50), 8), (1, (3,), 4), (1024, (11, 302, 50), 300),
                                        (1025, (11, 302), 100), (95398, (11, 302, 50), 300), (45056, (302,), 100),
                                        (5000, (4000, 4000), 12)])
def test_key_sort_and_keyed_row_sum(V, sizes, H):
    """psvae_key_sort: the permutation is the stable sort of the entries by (table, id) - bit-exact
    against numpy, ids outside a table clamped - and the offsets are the prefix counts;
    psvae_keyed_row_sum: every stacked-table row is the float64 sum of its entries' rows within
    1e-5 of the largest entry, and two calls agree bit for bit."""
    from ps_vae_amd import hip
    d = dev()
    rng = np.random.default_rng(V + H)
    keys = []
    for t, n in enumerate(sizes):
        k = rng.integers(-2, n + 2, size=V) if t == 0 else rng.integers(0, n, size=V)
        if t == 0 and V > 10:
            k[rng.random(V) < 0.7] = min(1, n - 1)          # one dominant id, like carbon among the atom types
        keys.append(k.astype(np.int32))
    chunk = hip.key_chunk(V)
    perm, row_off, unit_off = hip.key_sort([torch.from_numpy(k).to(d) for k in keys], sizes)
    R = sum(sizes)
    want_perm, counts = [], []
    for k, n in zip(keys, sizes):
        kc = np.clip(k, 0, n - 1)
        want_perm.append(np.argsort(kc, kind='stable').astype(np.int32))
        counts.append(np.bincount(kc, minlength=n))
    want_perm = np.concatenate(want_perm) if V else np.zeros(0, np.int32)
    counts = np.concatenate(counts)
    assert np.array_equal(perm.cpu().numpy(), want_perm)
    assert np.array_equal(row_off.cpu().numpy(), np.concatenate([[0], np.cumsum(counts)]))
    assert np.array_equal(unit_off.cpu().numpy(), np.concatenate([[0], np.cumsum(-(-counts // chunk))]))
    g = torch.randn(max(V, 1), H, generator=torch.Generator().manual_seed(V))[:V]
    gd = g.to(d)
    out = torch.full((R, H), 7.0, device=d)
    hip.keyed_row_sum(gd, perm, row_off, unit_off, out)
    want = torch.zeros(R, H, dtype=torch.float64)
    base = 0
    for k, n in zip(keys, sizes):
        want.index_add_(0, torch.from_numpy(np.clip(k, 0, n - 1).astype(np.int64)) + base, g.double())
        base += n
    close(out, want, what='keyed row sum')
    again = torch.empty(R, H, device=d)
    hip.keyed_row_sum(gd, perm, row_off, unit_off, again)
    assert torch.equal(out, again)
    # a strided gradient (a column block of a wider matrix) and a strided output
    if V and H >= 8:
        wide = torch.randn(V, H + 8, generator=torch.Generator().manual_seed(1)).to(d)
        out2 = torch.zeros(R, H + 4, device=d)
        hip.keyed_row_sum(wide[:, 4:4 + H], perm, row_off, unit_off, out2[:, :H])
        want2 = torch.zeros(R, H, dtype=torch.float64)
        base = 0
        for k, n in zip(keys, sizes):
            want2.index_add_(0, torch.from_numpy(np.clip(k, 0, n - 1).astype(np.int64)) + base,
                             wide[:, 4:4 + H].cpu().double())
            base += n
        close(out2[:, :H], want2, what='keyed row sum, strided')
        assert float(out2[:, H:].abs().max()) == 0.0


def make_pairs(B, N, seed, p=0.15):
    g = torch.Generator().manual_seed(seed)
    sel = torch.rand(B, N, N, generator=g) < p
    sel[1] = False                                  # a graph without any pair
    return sel


def test_pair_gather_fwd_bwd():
    from ps_vae_amd import ops
    d = dev()
    B, N, H, L = 5, 13, 300, 56
    g = torch.Generator().manual_seed(10)
    sel = make_pairs(B, N, 10)
    h = torch.randn(B * N, H, generator=g, requires_grad=True)
    z = torch.randn(B, L, generator=g, requires_grad=True)
    b, m, n = sel.nonzero(as_tuple=True)
    want = torch.cat([h[b * N + m], h[b * N + n], z[b]], dim=-1)
    gy = torch.randn(want.shape, generator=g)
    want.backward(gy)
    pairs = ops.PairList.from_edge_select(sel.to(d))
    hd, zd = h.detach().to(d).requires_grad_(), z.detach().to(d).requires_grad_()
    got = ops.pair_gather(hd, zd, pairs)
    got.backward(gy.to(d))
    assert torch.equal(got.cpu(), want.detach())
    close(hd.grad, h.grad, what='d pair / dh')
    close(zd.grad, z.grad, what='d pair / dz')


def test_pair_gather_bwd_many_pairs_per_graph():
    """More pairs in one graph than one LDS chunk holds (inference-sized candidate sets)."""
    from ps_vae_amd import ops
    d = dev()
    B, N, H, L = 2, 60, 36, 8
    g = torch.Generator().manual_seed(12)
    sel = torch.rand(B, N, N, generator=g) < 0.6
    h = torch.randn(B * N, H, generator=g, requires_grad=True)
    z = torch.randn(B, L, generator=g, requires_grad=True)
    b, m, n = sel.nonzero(as_tuple=True)
    assert (b == 0).sum() > 1024
    want = torch.cat([h[b * N + m], h[b * N + n], z[b]], dim=-1)
    gy = torch.randn(want.shape, generator=g)
    want.backward(gy)
    pairs = ops.PairList.from_edge_select(sel.to(d))
    hd, zd = h.detach().to(d).requires_grad_(), z.detach().to(d).requires_grad_()
    ops.pair_gather(hd, zd, pairs).backward(gy.to(d))
    close(hd.grad, h.grad, what='d pair / dh')
    close(zd.grad, z.grad, what='d pair / dz')


@pytest.mark.parametrize('B,N,F,dense', [(5, 13, 328, 0.1), (2, 60, 36, 0.6), (3, 7, 8, 0.0), (2, 9, 2048, 0.5)])
def test_pair_combine_fwd_bwd(B, N, F, dense):
    """relu(ab[src][:F] + ab[dst][F:] + c[graph]) per pair and its backward (gated sums into the node
    and graph rows, in pair order) against autograd in float64: a training-sized case, one with more
    pairs per graph than an LDS chunk, and one without any pair (all gradients zero)."""
    from ps_vae_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(B * 100 + N)
    sel = torch.rand(B, N, N, generator=g) < dense
    ab = torch.randn(B * N, 2 * F, generator=g, dtype=torch.float64, requires_grad=True)
    c = torch.randn(B, F, generator=g, dtype=torch.float64, requires_grad=True)
    b, m, n = sel.nonzero(as_tuple=True)
    if dense > 0.5:
        assert (b == 0).sum() > 1024
    want = torch.relu(ab[b * N + m][:, :F] + ab[b * N + n][:, F:] + c[b])
    gy = torch.randn(want.shape, generator=g, dtype=torch.float64)
    want.backward(gy)
    pairs = ops.PairList.from_edge_select(sel.to(d))
    abd, cd = ab.detach().float().to(d).requires_grad_(), c.detach().float().to(d).requires_grad_()
    got = ops.pair_combine(abd, cd, pairs, relu=True)
    got.backward(gy.float().to(d))
    if want.numel():
        close(got, want.detach(), what='pair_combine')
    else:
        assert got.shape == (0, F)
    # (a pre-activation within rounding of zero could take the other side in float32: none does here)
    close(abd.grad, ab.grad, what='d / d ab')
    close(cd.grad, c.grad, what='d / d c')


@pytest.mark.parametrize('F,C', [(656, 4), (328, 4), (64, 3), (512, 7), (330, 4), (100, 8)])
@pytest.mark.parametrize('P', [0, 1, 5, 4097])
def test_link_ce_fwd_bwd(P, F, C):
    """The class layer + cross-entropy kernels in every form: half a wave per pair (rows up to 512 floats, two to four
    float4 per lane, up to 4 or 8 classes), one wave per pair (wider rows), scalar loads (a width that is no
    multiple of 4); the input gradient as float4 columns and as scalars."""
    from ps_vae_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(13 + P + F)
    hid = torch.randn(P, F, generator=g, requires_grad=True)
    w = (torch.randn(C, F, generator=g) * 0.05).requires_grad_()
    b = torch.randn(C, generator=g, requires_grad=True)
    gold = torch.randint(0, C, (P,), generator=g)
    logits = torch.nn.functional.linear(hid, w, b)
    hd, wd, bd = (t.detach().to(d).requires_grad_() for t in (hid, w, b))
    lossd, logitsd, okd = ops.link_ce(hd, wd, bd, gold.to(d).int())
    if P == 0:
        assert torch.isnan(lossd).item()      # torch: mean over an empty batch
        return
    loss = torch.nn.functional.cross_entropy(logits, gold)
    (loss * 0.1).backward()
    (lossd * 0.1).backward()
    close(logitsd, logits, what='logits')
    close(lossd, loss, what='CE')
    assert okd.item() == (logits.argmax(-1) == gold).sum().item()
    close(hd.grad, hid.grad, what='d CE / d hidden')
    close(wd.grad, w.grad, what='d CE / dW')
    close(bd.grad, b.grad, what='d CE / db')


def test_link_infer():
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator().manual_seed(14)
    logits = torch.randn(1000, 4, generator=g) * 3
    prob = torch.softmax(logits, -1)
    conf, et = prob.max(-1)
    c = torch.empty(1000, device=d)
    t = torch.empty(1000, dtype=torch.int32, device=d)
    k = torch.empty(1000, dtype=torch.uint8, device=d)
    hip.link_infer(logits.to(d), 0.5, 0, c, t, k)
    close(c, conf, what='confidence')
    assert torch.equal(t.cpu().long(), et)
    # identical to the host rule except where conf sits within rounding of the threshold
    want = (conf >= 0.5) & (et != 0)
    near = (conf - 0.5).abs() < 1e-6
    assert torch.equal(k.cpu().bool()[~near], want[~near])


def test_exclusive_scan():
    from ps_vae_amd import hip
    d = dev()
    for n in (0, 1, 63, 1024, 1025, 5000):
        x = torch.randint(0, 100, (n,), dtype=torch.int32)
        out = torch.empty(n + 1, dtype=torch.int32, device=d)
        hip.exclusive_scan(x.to(d), out)
        want = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(x.long(), 0)])
        assert torch.equal(out.cpu().long(), want)


# ------------------------------------------------------------------ collation
REF_KEYS = ['x', 'x_pieces', 'x_pos', 'atom_mask', 'pieces', 'edge_index', 'edge_attr', 'edge_select',
            'golden_edge', 'in_piece_edge_idx']
COMPACT_KEYS = ['cx', 'cxp', 'cxpos', 'row_ptr', 'col', 'cls', 'in_row_ptr', 'in_col', 'in_cls',
                'pair_src', 'pair_dst', 'pair_graph', 'golden']


def compact_arrays(c):
    return {'node_off': c.node_off, 'cx': c.cx, 'cxp': c.cxp, 'cxpos': c.cxpos,
            'row_ptr': c.csr.row_ptr, 'col': c.csr.col, 'cls': c.csr.cls,
            'in_row_ptr': c.csr_in.row_ptr, 'in_col': c.csr_in.col, 'in_cls': c.csr_in.cls,
            'pair_src': c.pairs.src, 'pair_dst': c.pairs.dst, 'pair_graph': c.pairs.graph,
            'golden': c.golden, 'pair_ptr': c.pairs.ptr}


def check_collate(step1, uniforms, want, tokenizer):
    from ps_vae_amd.data.collate import collate_packed
    from ps_vae_amd.data.packed import PackedMolecules
    d = dev()
    packed = PackedMolecules.from_step1(step1)
    u = torch.from_numpy(np.ascontiguousarray(uniforms if len(uniforms) else np.zeros(1))).to(d)
    got = collate_packed(packed.to_device(d), np.arange(len(step1)), uniforms=u,
                         atom_pad=tokenizer.atom_pad_idx(), piece_pad=tokenizer.pad_idx())
    assert got.num_uniforms == len(uniforms)
    for k in REF_KEYS:
        g = got[k].cpu().numpy()
        w = want[k]
        if k == 'edge_index' and w.shape[1] == 0:
            assert g.shape == (2, 0)
            continue
        assert g.shape == w.shape and g.dtype == w.dtype, (k, g.shape, w.shape, g.dtype, w.dtype)
        assert np.array_equal(g, w), k
    assert np.array_equal(got['props'].cpu().numpy(), want['props'].astype(np.float32).reshape(len(step1), -1))
    wc = collate_ref.compact_from_reference({k: want[k] for k in REF_KEYS})
    gc = compact_arrays(got.compact)
    for k, w in wc.items():
        g = gc[k].cpu().numpy()
        assert g.shape == w.shape and np.array_equal(g, w), k


@pytest.mark.parametrize('name', ['single_atom', 'one_piece_only', 'no_bond_only', 'mixed_small',
                                  'zinc_ragged', 'qm9_ragged', 'max_size_38'])
def test_collate_golden_bit_exact(name, tokenizer):
    with open(os.path.join(GOLDEN, 'collate_cases.json')) as f:
        step1 = json.load(f)[name]
    gold = sub(load_npz('collate_cases.npz'), name)
    check_collate(step1, gold['uniforms'], gold, tokenizer)


def test_collate_batch_512_vs_oracle(tokenizer):
    from ps_vae_amd.data.synthetic import synthetic_step1
    step1 = synthetic_step1(512, seed=77, shape='zinc')
    rng = np.random.default_rng(5)
    uniforms = rng.random(700 * 512)
    want, used = collate_ref.collate(step1, uniforms, piece_pad=tokenizer.pad_idx())
    check_collate(step1, uniforms[:used], want, tokenizer)


def test_collate_subset_order_and_large_molecule(tokenizer):
    """Batches are arbitrary id subsets of a packed dataset; a 120-atom molecule exercises
    many 256-cell chunks and the LDS sizing."""
    from ps_vae_amd.data.collate import collate_packed
    from ps_vae_amd.data.packed import PackedMolecules
    from ps_vae_amd.data.synthetic import random_molecule, synthetic_step1
    d = dev()
    step1 = synthetic_step1(20, seed=3, shape='zinc')
    step1.append(random_molecule(np.random.default_rng(1), 120))
    packed = PackedMolecules.from_step1(step1)
    ids = [20, 3, 3, 17, 0]
    uniforms = np.random.default_rng(9).random(40000)
    want, used = collate_ref.collate([step1[i] for i in ids], uniforms, piece_pad=tokenizer.pad_idx())
    got = collate_packed(packed.to_device(d), ids, uniforms=torch.from_numpy(uniforms).to(d),
                         atom_pad=tokenizer.atom_pad_idx(), piece_pad=tokenizer.pad_idx())
    assert got.num_uniforms == used
    for k in REF_KEYS:
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k


def test_collate_largest_molecule_and_limits(tokenizer):
    """250 atoms is the largest molecule the collation kernels take (its n x n byte matrix fills
    the 64 KB of LDS): bit-exact against the oracle beside small molecules; 251 atoms and an
    insufficient uniform stream are refused on the host before anything is launched."""
    from ps_vae_amd.data.collate import collate_packed
    from ps_vae_amd.data.packed import PackedMolecules
    from ps_vae_amd.data.synthetic import random_molecule, synthetic_step1
    d = dev()
    rng = np.random.default_rng(4)
    step1 = synthetic_step1(3, seed=8, shape='qm9') + [random_molecule(rng, 250), random_molecule(rng, 251)]
    packed = PackedMolecules.from_step1(step1)
    dp = packed.to_device(d)
    uniforms = np.random.default_rng(10).random(int(packed.n_rng.sum()) + 4)
    ids = [3, 0, 1, 2]
    want, used = collate_ref.collate([step1[i] for i in ids], uniforms, piece_pad=tokenizer.pad_idx())
    got = collate_packed(dp, ids, uniforms=torch.from_numpy(uniforms).to(d), atom_pad=tokenizer.atom_pad_idx(),
                         piece_pad=tokenizer.pad_idx())
    assert got.num_uniforms == used and got['x'].shape[1] == 250
    for k in REF_KEYS:
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    with pytest.raises(ValueError, match='251 atoms'):
        collate_packed(dp, [4], uniforms=torch.from_numpy(uniforms).to(d))
    with pytest.raises(ValueError, match='uniforms needed'):
        collate_packed(dp, ids, uniforms=torch.from_numpy(uniforms[:10]).to(d))


def test_collate_empty_batch(tokenizer):
    """No molecules: every tensor of both layouts comes back with zero rows."""
    from ps_vae_amd.data.collate import collate_packed
    from ps_vae_amd.data.packed import PackedMolecules
    from ps_vae_amd.data.synthetic import synthetic_step1
    packed = PackedMolecules.from_step1(synthetic_step1(2, seed=8, shape='qm9'))
    got = collate_packed(packed.to_device(dev()), [], atom_pad=tokenizer.atom_pad_idx(), piece_pad=tokenizer.pad_idx())
    torch.cuda.synchronize()
    assert got.batch_size == 0 and got['x'].shape == (0, 0) and got['edge_index'].shape == (2, 0)
    assert got['golden_edge'].numel() == 0 and got['edge_select'].numel() == 0
    c = got.compact
    assert c.num_nodes == 0 and len(c.pairs) == 0 and c.csr.row_ptr.cpu().tolist() == [0]


# ------------------------------------------------------------------ greedy bond assembly
def _assembly_case(rng, n_lo, n_hi):
    """A random fragment set: atoms, bonds inside the fragments, groups, candidates between
    fragments in a random visiting order with random bond classes (0 = None appears too)."""
    n = int(rng.integers(n_lo, n_hi + 1))
    atoms = rng.choice(10, size=n, p=[.02, .02, .6, .02, .03, .01, .12, .1, .02, .06]).tolist()
    k = int(rng.integers(1, max(2, n // 3) + 1))
    frag = np.sort(rng.integers(0, k, size=n))
    _, frag = np.unique(frag, return_inverse=True)
    groups = [np.nonzero(frag == g)[0].tolist() for g in range(int(frag.max()) + 1)]
    bonds = []
    for grp in groups:                        # a chain through each fragment, sometimes closed
        for a, b in zip(grp[:-1], grp[1:]):
            bonds.append((a, b, int(rng.choice([1, 1, 1, 2]))))
        if len(grp) in (5, 6) and rng.random() < 0.5:
            bonds.append((grp[0], grp[-1], 1))
    cand = [(i, j) for i in range(n) for j in range(i + 1, n) if frag[i] != frag[j]]
    ctype = rng.choice([0, 1, 1, 1, 2, 3], size=len(cand)).tolist()
    take = rng.random(len(cand)) < 0.6
    order = rng.permutation(np.nonzero(take)[0]).tolist()
    return atoms, bonds, groups, cand, ctype, order


@pytest.mark.gpu
@pytest.mark.parametrize('B,n_lo,n_hi', [(1, 1, 1), (7, 2, 6), (300, 1, 40), (65, 30, 70)])
def test_assemble_links_vs_oracle(B, n_lo, n_hi, tokenizer):
    """One lane per molecule against the per-molecule Python loop: accepted flags, final
    blocks and kept atoms are identical."""
    from oracle import assemble_ref
    from ps_vae_amd import hip
    from ps_vae_amd.utils.chem_vocab import MAX_VALENCE
    d = dev()
    rng = np.random.default_rng(1000 + B)
    vocab = tokenizer.chem_vocab
    cases = [_assembly_case(rng, n_lo, n_hi) for _ in range(B)]
    node_off = np.concatenate([[0], np.cumsum([len(c[0]) for c in cases])]).astype(np.int32)
    V = int(node_off[-1])
    atom_type = np.concatenate([c[0] for c in cases]).astype(np.int32)
    block0 = np.zeros(V, dtype=np.int32)
    n_blocks = np.zeros(B, dtype=np.int32)
    rows = [[] for _ in range(V)]
    ord_m, ord_n, ord_t, order_ptr = [], [], [], [0]
    want_acc, want_block, want_keep = [], [], []
    for b, (atoms, bonds, groups, cand, ctype, order) in enumerate(cases):
        o = int(node_off[b])
        for gi, grp in enumerate(groups):
            block0[o + np.asarray(grp, dtype=np.int64)] = gi
        n_blocks[b] = len(groups)
        for a, e, t in bonds:
            rows[o + a].append((o + e, t)), rows[o + e].append((o + a, t))
        for i in order:
            ord_m.append(cand[i][0]), ord_n.append(cand[i][1]), ord_t.append(ctype[i])
        order_ptr.append(len(ord_m))
        acc, blk, keep = assemble_ref.assemble(atoms, bonds, groups, cand, ctype, order, vocab)
        want_acc += acc
        want_block += blk
        want_keep += keep
    row_ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    col = np.asarray([c for r in rows for c, _ in r], dtype=np.int32)
    cls = np.asarray([t for r in rows for _, t in r], dtype=np.uint8)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)   # noqa: E731
    K = len(ord_m)
    accepted = torch.full((K,), 7, dtype=torch.uint8, device=d)
    block = torch.full((V,), -1, dtype=torch.int32, device=d)
    keep_atom = torch.full((V,), 7, dtype=torch.uint8, device=d)
    status = torch.full((B,), -1, dtype=torch.int32, device=d)
    max_val = torch.tensor([MAX_VALENCE[vocab.idx_to_atom(i)] for i in range(len(MAX_VALENCE))], dtype=torch.int32,
                           device=d)
    hip.assemble_links(t(node_off), t(atom_type), t(block0), t(n_blocks), t(row_ptr), t(col), t(cls),
                       t(np.asarray(order_ptr, dtype=np.int32)), t(np.asarray(ord_m, dtype=np.int32)),
                       t(np.asarray(ord_n, dtype=np.int32)), t(np.asarray(ord_t, dtype=np.int32)), max_val,
                       vocab.atom_to_idx('S'), accepted, block, keep_atom, status)
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0] * B
    assert accepted.cpu().tolist() == [int(v) for v in want_acc]
    assert block.cpu().tolist() == want_block
    assert keep_atom.cpu().tolist() == [int(v) for v in want_keep]


@pytest.mark.gpu
def test_assemble_links_rejects_bad_ids(tokenizer):
    """A block id outside 0..n_blocks-1 is reported in status, not followed."""
    from ps_vae_amd import hip
    d = dev()
    i32 = torch.int32
    z = lambda n, dt=i32: torch.zeros(n, dtype=dt, device=d)   # noqa: E731
    node_off = torch.tensor([0, 2], dtype=i32, device=d)
    block0 = torch.tensor([0, 5], dtype=i32, device=d)
    status = z(1)
    acc, blk, keep = z(1, torch.uint8), z(2), z(2, torch.uint8)
    hip.assemble_links(node_off, z(2) + 2, block0, z(1) + 2, z(3), z(0), z(0, torch.uint8),
                       torch.tensor([0, 1], dtype=i32, device=d), z(1), z(1) + 1, z(1) + 1,
                       torch.full((10,), 4, dtype=i32, device=d), 9, acc, blk, keep, status)
    assert status.item() == 2 and acc.item() == 0 and keep.cpu().tolist() == [0, 0]


# ------------------------------------------------------------------ piece loss (CE with an ignored label)
@pytest.mark.parametrize('R,V,ld', [(45056, 302, 302), (1000, 302, 320), (7, 1, 1), (130, 64, 64), (129, 65, 80),
                                    (50, 1000, 1000), (0, 302, 302)])
def test_token_ce_against_torch(R, V, ld):
    """psvae_token_ce_fwd + its autograd wrapper against nn.CrossEntropyLoss(ignore_index) in
    float64: loss, gradient of the logits, token count and argmax hits."""
    from ps_vae_amd import ops as hip_ops
    d = dev()
    g = torch.Generator().manual_seed(R + V)
    buf = torch.randn(R, ld, generator=g) * 3
    gold = torch.randint(0, V, (R,), generator=g)
    ignore = V - 1 if V > 1 else 5
    if R:
        gold[torch.rand(R, generator=g) < 0.3] = ignore
    logits = buf[:, :V].to(d).requires_grad_(True) if ld == V else buf.to(d)[:, :V].requires_grad_(True)
    loss, counts = hip_ops.token_ce(logits, gold.to(d).to(torch.int32), ignore)
    ref_x = buf[:, :V].double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_x, gold, ignore_index=ignore) if R else torch.tensor(float('nan'))
    live = gold != ignore
    if int(live.sum()) == 0:
        assert torch.isnan(loss).item()
        return
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    (loss * 0.37).backward()
    (ref * 0.37).backward()
    close(logits.grad, ref_x.grad, what='dlogits')
    hits = int(((ref_x.argmax(-1) == gold) & live).sum())
    assert counts.tolist() == [int(live.sum()), hits]


def test_token_ce_with_padded_vocabulary():
    """Logits whose last columns are padding (302 classes in rows of 304): loss, counts and the
    gradient of the class columns equal the unpadded call's, the padding columns get zero gradient
    whatever they hold."""
    from ps_vae_amd import ops as hip_ops
    d = dev()
    g = torch.Generator().manual_seed(5)
    R, V, W = 1000, 302, 304
    x = (torch.randn(R, W, generator=g) * 3).to(d)
    x[:, V:] = 50.0                      # would win every argmax and dominate every softmax if it were read
    gold = torch.randint(0, V, (R,), generator=g).to(d).to(torch.int32)
    gold[::7] = V - 1
    a = x.clone().requires_grad_(True)
    b = x[:, :V].clone().requires_grad_(True)
    la, ca = hip_ops.token_ce(a, gold, V - 1, n_classes=V)
    lb, cb = hip_ops.token_ce(b, gold, V - 1)
    assert la.item() == lb.item() and ca.tolist() == cb.tolist()
    la.backward()
    lb.backward()
    assert torch.equal(a.grad[:, :V], b.grad) and not bool(a.grad[:, V:].any())


def test_token_ce_all_rows_ignored_is_nan():
    from ps_vae_amd import ops as hip_ops
    d = dev()
    logits = torch.randn(9, 12, device=d)
    gold = torch.full((9,), 3, dtype=torch.int32, device=d)
    loss, counts = hip_ops.token_ce(logits, gold, 3)
    assert torch.isnan(loss).item() and counts.tolist() == [0, 0]


# ------------------------------------------------------------------ piece draw (generation)
@pytest.mark.parametrize('B,V,temperature,first', [(1, 7, 1.0, True), (1000, 302, 1.0, False), (513, 302, 0.7, True),
                                                    (200, 1500, 1.3, False)])
def test_piece_draw_against_the_elementwise_form(B, V, temperature, first):
    """psvae_piece_draw against the chain of torch operations it replaces (masks, softmax, running sum,
    inverse-CDF pick, counters): the same pick wherever the threshold is not within rounding of an entry
    of the running sum, and the same state for every molecule whose pick agrees."""
    from ps_vae_amd import hip
    d = dev()
    g = torch.Generator().manual_seed(B + V)
    logits = (torch.randn(B, V, generator=g) * 3).to(d)
    logits[::7, 3:40] = -1e30                                   # rows with classes of zero probability
    u = torch.rand(B, 5, generator=g).to(d)
    pad, end = V - 2, V - 1
    cnt = torch.randint(0, 9, (V,), generator=g, dtype=torch.int32).to(d)
    cnt[pad:] = 0
    active0 = (torch.rand(B, generator=g) < 0.8).to(d)
    atoms0 = torch.randint(0, 40, (B,), generator=g, dtype=torch.int32).to(d)
    length0 = torch.randint(0, 50, (B,), generator=g, dtype=torch.int32).to(d)
    max_atom, max_pos = 38, 50
    # the element-wise form
    lg = logits.clone()
    lg[:, pad] = float('-inf')
    if first:
        lg[:, end] = float('-inf')
    probs = torch.softmax(lg / temperature, dim=-1)
    cdf = torch.cumsum(probs, dim=-1)
    last = ((probs > 0) * torch.arange(V, device=d)[None, :]).max(dim=-1).values
    thr = u[:, 2:3] * cdf[:, -1:]
    idx = torch.minimum((cdf <= thr).sum(dim=-1), last)
    margin = ((cdf - thr).abs() / cdf[:, -1:]).min(dim=-1).values       # distance of the threshold from the nearest entry
    idx = torch.where(active0, idx, torch.full_like(idx, pad))
    want_len = length0 + active0.int()
    want_atoms = atoms0 + torch.where(active0, cnt[idx], torch.zeros_like(atoms0))
    stop = (idx == end) | (want_atoms > max_atom) | (want_len == max_pos)
    want_active = active0 & ~stop
    want_cur = torch.where(want_active, idx, torch.full_like(idx, end))
    # the kernel
    active, atoms, length = active0.to(torch.uint8).clone(), atoms0.clone(), length0.clone()
    out = torch.full((B, 5), -1, dtype=torch.int64, device=d)
    cur = torch.full((B,), -1, dtype=torch.int64, device=d)
    hip.piece_draw(logits, temperature, u[:, 2], pad, end, first, cnt, max_atom, max_pos, active, atoms, length, out[:, 2], cur)
    same = out[:, 2] == idx
    clear = margin > 1e-5
    assert bool(same[clear].all()), 'a pick differs although the threshold is clear of the running sum'
    assert int((~same).sum()) <= max(1, B // 200)
    assert bool((out[:, [0, 1, 3, 4]] == -1).all())                      # only its own column is written
    for got, want, what in ((active.bool(), want_active, 'active'), (atoms, want_atoms, 'atoms'), (length, want_len, 'length'),
                            (cur, want_cur, 'cur')):
        assert torch.equal(got[same], want[same]), what
    assert torch.equal(logits, logits)                                   # (the logits are not modified)


def test_pair_enumerate_against_the_masked_cells():
    """psvae_pair_enumerate_count / _fill against the dense form (all n x n cells of every molecule, filtered):
    same candidates in the same order, including empty molecules, single-fragment molecules and a molecule
    larger than one round of threads."""
    from ps_vae_amd import hip
    d = dev()
    rng = np.random.default_rng(2)
    sizes = [0, 1, 5, 23, 38, 0, 300, 2, 17] + [int(v) for v in rng.integers(1, 40, size=200)]
    node_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    frag = np.concatenate([rng.integers(0, max(1, s // 4 + 1), size=s) for s in sizes]).astype(np.int32)
    frag[node_off[3]:node_off[4]] = 7                      # a molecule that is one fragment: no candidate
    want = {k: [] for k in ('src', 'dst', 'graph', 'm', 'k')}
    ptr = [0]
    for b, s in enumerate(sizes):
        f = frag[node_off[b]:node_off[b + 1]]
        m, k = np.nonzero((np.arange(s)[:, None] < np.arange(s)[None, :]) & (f[:, None] != f[None, :]))
        want['src'] += list(node_off[b] + m), 
        want['dst'] += list(node_off[b] + k),
        want['graph'] += [np.full(len(m), b)]
        want['m'] += [m]
        want['k'] += [k]
        ptr.append(ptr[-1] + len(m))
    got = hip.pair_enumerate(torch.from_numpy(node_off).to(d), torch.from_numpy(frag).to(d), max(sizes))
    assert got[5].cpu().tolist() == ptr
    for arr, key in zip(got[:5], ('src', 'dst', 'graph', 'm', 'k')):
        flat = np.concatenate([np.asarray(v).reshape(-1) for v in want[key]]) if want[key] else np.zeros(0)
        assert np.array_equal(arr.cpu().numpy(), flat.astype(np.int32)), key


def test_gine_table_gradient_through_slabs():
    """psvae_gine_aggregate_bwd_bits leaving its table-gradient partials as slabs and psvae_gine_table_reduce over
    the slabs of two calls = the two calls' table gradients added; g_x is the same either way."""
    from ps_vae_amd import hip, ops
    d = dev()
    V, H, C = 777, 300, 4
    g0 = torch.Generator().manual_seed(9)
    ei, attr, x, table = random_graph(V, H, C, seed=21)
    csr = ops.Csr.from_edge_index(ei.to(d), attr.to(d), V)
    x, table = x.to(d), table.to(d)
    out = torch.empty(V, H, device=d)
    words, wtype = hip.gine_mask_words(V, H, C)
    mask = torch.zeros(words, dtype=wtype, device=d)
    hip.gine_aggregate_fwd_bits(x, table, csr.row_ptr, csr.col, csr.cls, out, mask)
    gs = [torch.randn(V, H, generator=g0).to(d) for _ in range(2)]
    want_t = torch.zeros(C, H, device=d)
    want_x = []
    for i, gy in enumerate(gs):
        gx = torch.empty(V, H, device=d)
        hip.gine_aggregate_bwd_bits(mask, C, gy, csr.t_row_ptr, csr.t_col, csr.t_cls, gx, want_t, accumulate_table=i > 0)
        want_x.append(gx)
    n = hip.gine_bwd_slabs(V, H, C)
    assert n >= 1
    slabs = torch.full((2, n, C, H), float('nan'), device=d)
    untouched = torch.full((C, H), 7.0, device=d)
    for i, gy in enumerate(gs):
        gx = torch.empty(V, H, device=d)
        hip.gine_aggregate_bwd_bits(mask, C, gy, csr.t_row_ptr, csr.t_col, csr.t_cls, gx, untouched, slabs=slabs[i])
        assert torch.equal(gx, want_x[i])
    assert bool((untouched == 7.0).all())
    got_t = torch.empty(C, H, device=d)
    hip.gine_table_reduce(slabs.view(-1, C, H), got_t)
    close(got_t, want_t, what='table gradient through slabs')
    again = torch.empty(C, H, device=d)
    hip.gine_table_reduce(slabs.view(-1, C, H), again)
    assert torch.equal(again, got_t)
    hip.gine_table_reduce(slabs.view(-1, C, H), again, accumulate=True)
    close(again, 2 * want_t, what='accumulated')


@pytest.mark.parametrize('M,N,K,C', [(1, 328, 328, 4), (1000, 328, 328, 4), (130, 160, 64, 3), (4099, 384, 300, 4), (257, 132, 8, 1)])
def test_gemm_relu_classes(M, N, K, C):
    """psvae_gemm_f32_relu_classes (hidden Linear + ReLU + class layer, the hidden rows never written) against
    the same arithmetic in float64."""
    from ps_vae_amd import hip, ops
    d = dev()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w, b = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    cw, cb = torch.randn(C, N, generator=g) / N ** 0.5, torch.randn(C, generator=g)
    want = torch.relu(x.double() @ w.double().T + b.double()) @ cw.double().T + cb.double()
    assert hip.relu_classes_tiles(N) in (1, 2) and hip.relu_classes_tiles(100) == 0 and hip.relu_classes_tiles(600) == 0
    got = ops.linear_relu_classes(x.to(d), w.to(d), b.to(d), cw.to(d), cb.to(d))
    close(got, want, what='logits')
    nobias = ops.linear_relu_classes(x.to(d), w.to(d), None, cw.to(d), None)
    close(nobias, torch.relu(x.double() @ w.double().T) @ cw.double().T, what='logits without biases')
    assert ops.linear_relu_classes(x.to(d), w[:100].contiguous().to(d), b[:100].to(d), cw[:, :100].contiguous().to(d), cb.to(d)) is None
