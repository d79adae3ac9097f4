"""Binding of csrc/libpsvae_hip.so (C ABI declared in include/psvae_hip.h).

Two routes, both thin.  The operators the training step spends its launches on - matrix product,
message-passing aggregation, pair combination, link loss - are PyTorch custom operators
(torch.ops.psvae.*, registered with torch.library by csrc/libpsvae_torch.so over the same C ABI).
Every other entry point is bound with ctypes: the wrapper checks that its tensors live on the GPU
with the expected dtype / contiguity, passes raw device pointers plus PyTorch's current HIP stream,
and turns a non-zero status into a RuntimeError carrying psvae_last_error().  There is no
fallback: without the libraries, or with a CPU tensor, the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PSVAE_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libpsvae_hip.so')
TORCH_OPS_PATH = os.path.join(_HERE, 'csrc', 'libpsvae_torch.so')

_c_int, _c_i64, _c_size = ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
_c_f32, _c_f64, _c_ll, _p = ctypes.c_float, ctypes.c_double, ctypes.c_longlong, ctypes.c_void_p

# name -> (restype, argtypes); argument order follows include/psvae_hip.h
_SIGNATURES = {
    'psvae_version': (_c_int, []),
    'psvae_last_error': (ctypes.c_char_p, []),
    'psvae_collate_count': (_c_int, [_c_int, _c_int] + [_p] * 8 + [_p, _c_f64, _p, _p, _p]),
    'psvae_collate_fill': (_c_int, [_c_int, _c_int, _c_int] + [_p] * 16 + [_c_ll, _p, _c_f64, _c_int, _c_int]
                           + [_p] * 7 + [_c_int] + [_p] * 4 + [_p] * 13 + [_p]),
    'psvae_exclusive_scan_i32': (_c_int, [_c_int, _p, _p, _p]),
    'psvae_embed_atom_fwd': (_c_int, [_c_int, _p, _p, _p, _p, _c_int, _c_int, _p, _c_int, _c_int, _p, _c_int,
                                      _c_int, _p, _p]),
    'psvae_embed_atom_bwd': (_c_int, [_c_int, _p, _p, _p, _p] + [_c_int] * 6 + [_p, _p, _p, _p]),
    'psvae_embed_atom_onehot': (_c_int, [_c_int, _p, _p, _p, _c_int, _c_int, _c_int, _p, _c_int, _p]),
    'psvae_embedding_fwd': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p]),
    'psvae_onehot_f32': (_c_int, [_c_int, _p, _p, _c_int, _p]),
    'psvae_gine_aggregate_fwd': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _p, _p, _p, _p, _c_int, _p]),
    'psvae_gine_aggregate_bwd_workspace_bytes': (_c_size, [_c_int, _c_int, _c_int]),
    'psvae_gine_aggregate_bwd': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _p, _c_int, _p, _p, _p, _p,
                                          _c_int, _p, _c_int, _p, _c_int, _p, _c_size, _p]),
    'psvae_gine_aggregate_bwd_noalloc': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _p, _c_int, _p, _p, _p, _p,
                                                  _c_int, _p, _c_int, _p, _c_int, _p]),
    'psvae_gemm_f32_workspace_bytes': (_c_size, [_c_int, _c_int, _c_int, _c_int]),
    'psvae_gemm_f32': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _c_i64, _c_i64, _p, _c_i64, _c_i64, _p, _c_i64,
                                _p, _p, _p, _c_int, _c_int, _c_int, _p, _c_size, _p]),
    'psvae_pair_combine_fwd': (_c_int, [_c_int, _c_int, _p, _c_int, _p, _c_int, _p, _c_int, _p, _p, _p, _c_int, _p,
                                        _c_int, _p]),
    'psvae_affine_merge_fwd': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_affine_merge_bwd': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_colsum_workspace_bytes': (_c_size, [_c_int, _c_int]),
    'psvae_colsum_f32': (_c_int, [_c_int, _c_int, _p, _p, _c_i64, _p, _c_int, _p, _c_size, _p]),
    'psvae_segment_sum_fwd': (_c_int, [_c_int, _c_int, _p, _c_int, _p, _p, _p, _c_int, _p]),
    'psvae_segment_sum_bwd': (_c_int, [_c_int, _c_int, _p, _c_int, _p, _p, _p, _c_int, _p]),
    'psvae_rsample_kl_fwd': (_c_int, [_c_int, _c_int, _p, _p, _p, _p, _p, _p]),
    'psvae_rsample_kl_bwd': (_c_int, [_c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_gru_cell_fwd': (_c_int, [_c_int, _c_int, _p, _p, _p, _p, _p, _p]),
    'psvae_gru_cell_bwd': (_c_int, [_c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_gru_seq_fwd': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_gru_seq_bwd': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_pair_gather_fwd': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _c_int, _p, _p, _p, _p, _p]),
    'psvae_pair_gather_bwd': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _c_int, _p, _c_int, _p]),
    'psvae_link_ce_workspace_bytes': (_c_size, [_c_int]),
    'psvae_link_ce_fwd': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _c_size, _p]),
    'psvae_link_ce_bwd_input': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p, _c_int, _p]),
    'psvae_pair_combine_bwd': (_c_int, [_c_int, _c_int, _p, _c_int, _p, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _c_int, _p,
                                        _c_int, _p]),
    'psvae_token_ce_workspace_bytes': (_c_size, [_c_int]),
    'psvae_token_ce_fwd': (_c_int, [_c_int, _c_int, _p, _c_i64, _p, _c_int, _p, _p, _p, _p, _c_size, _p]),
    'psvae_link_infer': (_c_int, [_c_int, _c_int, _p, _c_f32, _c_int, _p, _p, _p, _p]),
    'psvae_embed_trans_fwd': (_c_int, [_c_int, _c_int, _p, _p, _p, _c_int, _c_int, _c_int, _p, _p, _c_int, _p]),
    'psvae_clip_adam_workspace_bytes': (_c_size, [_c_ll]),
    'psvae_clip_adam_step': (_c_int, [_c_ll, _p, _p, _p, _p, _c_f32, _c_f32] + [_c_f64] * 5 + [_c_int, _c_int, _p, _p, _c_size, _p]),
    'psvae_assemble_links_workspace_bytes': (_c_size, [_c_int]),
    'psvae_assemble_links': (_c_int, [_c_int, _c_int] + [_p] * 12 + [_c_int, _c_int] + [_p] * 5 + [_c_size, _p]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` '
                               '(or __graft_entry__.build()); there is no CPU fallback')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


_ops = None


def ops():
    """torch.ops.psvae (the operators csrc/libpsvae_torch.so registers); raises if it is not built.
    With PSVAE_HIP_LIB set (an A/B build of the kernel library) the operator library, which is
    linked to the in-tree kernels, is bypassed: these calls then take the ctypes route too."""
    global _ops
    if _ops is None:
        if not os.path.exists(TORCH_OPS_PATH):
            raise RuntimeError(f'{TORCH_OPS_PATH} is missing: build it with `make -C {os.path.dirname(TORCH_OPS_PATH)}` '
                               '(or __graft_entry__.build()); there is no CPU fallback')
        lib()                                  # the kernel library first: the operator library links to it
        torch.ops.load_library(TORCH_OPS_PATH)
        _ops = _Overloads(torch.ops.psvae)
    return _ops


class _Overloads:
    """The `.default` overload of every operator, looked up once: calling an overload directly skips
    the per-call overload resolution of torch.ops.<ns>.<name>(...)."""

    def __init__(self, namespace):
        self._namespace = namespace

    def __getattr__(self, name):
        fn = getattr(self._namespace, name).default
        setattr(self, name, fn)
        return fn


_CTYPES_ONLY = bool(os.environ.get('PSVAE_HIP_LIB'))

CALLS = 0   # C-ABI calls made so far (bench.py reports calls per step)


def _check(rc, name):
    global CALLS
    CALLS += 1
    if rc != 0:
        msg = lib().psvae_last_error().decode(errors='replace')
        raise RuntimeError(f'{name} failed with status {rc}: {msg}')


try:   # raw handle of PyTorch's current stream without building a torch.cuda.Stream object per launch
    _raw_stream = torch._C._cuda_getCurrentRawStream
    _cur_device = torch._C._cuda_getDevice
except AttributeError:   # other PyTorch builds: the public (slower) route
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream

    _cur_device = torch.cuda.current_device


def _stream():
    return ctypes.c_void_p(_raw_stream(_cur_device()))


def ptr(t, dtype=None):
    """Device pointer of tensor t (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('psvae HIP ops need GPU tensors (there is no CPU fallback)')
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f'expected {dtype}, got {t.dtype}')
    return ctypes.c_void_p(t.data_ptr())


def _rows(t):
    """(pointer, row stride in elements) of a 2-D tensor whose last dim is contiguous."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f'need a 2-D tensor with contiguous rows, got shape {tuple(t.shape)} strides {t.stride()}')
    return ptr(t, torch.float32), (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


_workspaces = {}


def workspace(nbytes, device):
    """A scratch buffer per (device, stream), grown on demand.  Kernels that reuse it are
    ordered by the stream they are enqueued on; two streams never share one."""
    index = device.index if device.index is not None else _cur_device()
    key = (index, _raw_stream(index))
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


# --------------------------------------------------------------------------- profiling
# bench.py sets PROFILE = {} around its timed region: every wrapped launch is then bracketed
# by two events on the launch stream (PyTorch's current stream) and logged with its
# algorithmic work (flops for the GEMM, bytes for the memory-bound kernels).
PROFILE = None


class _Prof:
    __slots__ = ('name', 'work', 'e0', 'tag')

    def __init__(self, name, work, tag=None):
        self.name, self.work, self.e0, self.tag = name, work, None, tag

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.e0 is not None and PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE.setdefault(self.name, []).append((self.e0, e1, self.work, self.tag))


def aggregate_bytes(V, E, H, backward=False):
    """Algorithmic bytes of one GINE aggregation: per directed edge one neighbour row (4H),
    its int32 column index and 1-byte bond class; per node its own row in, the output row out
    and an int32 row pointer; the backward also reads the saved input row."""
    return E * (4 * H + 5) + V * (8 * H + 4) + (V * 4 * H if backward else 0)


def gemm_shape_table(prof):
    """Per GEMM shape: [M, N, K, A contiguous along, B contiguous along, masked, rowsum,
    launches, total ms, TFLOP/s], slowest first (call after a device synchronise)."""
    acc = {}
    for it in prof.get('gemm_f32_kernel', []):
        a = acc.setdefault(it[3], [0, 0.0, 0.0])
        a[0] += 1
        a[1] += it[0].elapsed_time(it[1])
        a[2] += it[2]
    rows = [list(k) + [v[0], round(v[1], 3), round(v[2] / (v[1] * 1e-3) / 1e12, 1) if v[1] > 0 else 0.0]
            for k, v in acc.items()]
    return sorted(rows, key=lambda r: -r[8])


def roofline_report(prof, hbm_peak_gbs, mfma_peak_tflops, traffic=None):
    """Summarise a PROFILE dict (call after a device synchronise).  `traffic`: optional
    {kernel group: {'hbm_bytes_per_launch': ...}} from a separate counter pass."""
    torch.cuda.synchronize()
    groups = {}
    for name, items in prof.items():
        ms = sum(it[0].elapsed_time(it[1]) for it in items)
        groups[name] = {'launches': len(items), 'ms': ms, 'work': sum(it[2] for it in items)}
    out = {}

    def entry(name, bound):
        g = groups.get(name)
        if not g or g['ms'] <= 0:
            return None
        if bound == 'mfma':
            ach, peak, unit = g['work'] / (g['ms'] * 1e-3) / 1e12, mfma_peak_tflops, 'TFLOP/s'
        else:
            ach, peak, unit = g['work'] / (g['ms'] * 1e-3) / 1e9, hbm_peak_gbs, 'GB/s'
        return {'kernel': name, 'bound': bound, 'achieved': round(ach, 2), 'peak': peak, 'unit': unit,
                'frac': round(ach / peak, 4),
                'traffic': (traffic or {}).get(name, {}).get('hbm_bytes_per_launch'), 'launches': g['launches'],
                'avg_launch_us': round(g['ms'] * 1e3 / g['launches'], 2),
                'work_per_launch': round(g['work'] / g['launches'], 1)}

    mf = entry('gemm_f32_kernel', 'mfma')
    hb = entry('gine_aggregate_fwd_kernel', 'hbm')
    hbb = entry('gine_aggregate_bwd_kernel', 'hbm')
    cands = [e for e in (mf, hb, hbb) if e]
    if cands:
        dominant = max(cands, key=lambda e: groups[e['kernel']]['ms'])
        out['roofline'] = dominant
    if mf:
        out['roofline_mfma'] = mf
    if hb:
        out['roofline_hbm'] = hb
    if hbb:
        out['roofline_hbm_bwd'] = hbb
    out['event_ms_total'] = {k: round(v['ms'], 3) for k, v in sorted(groups.items())}
    return out


# --------------------------------------------------------------------------- raw calls
def gemm(M, N, K, A, sa_m, sa_k, B, sb_k, sb_n, C, ldc, bias=None, relu=False, accumulate=False,
         amask=None, split_k=0, rowsum=None, cmask=None):
    tag = None
    if PROFILE is not None:
        tag = (M, N, K, 'k' if sa_k == 1 else 'm', 'k' if sb_k == 1 else 'n', amask is not None, rowsum is not None)
    if not _CTYPES_ONLY:
        global CALLS
        CALLS += 1
        with _Prof('gemm_f32_kernel', 2.0 * M * N * K, tag):
            ops().gemm(M, N, K, A, amask, sa_m, sa_k, B, sb_k, sb_n, C, ldc, bias, rowsum, cmask, relu, accumulate, split_k)
        return
    L = lib()
    nbytes = L.psvae_gemm_f32_workspace_bytes(M, N, K, split_k)
    ws = workspace(nbytes, C.device) if nbytes else None
    with _Prof('gemm_f32_kernel', 2.0 * M * N * K, tag):
        rc = L.psvae_gemm_f32(M, N, K, ptr(A, torch.float32), ptr(amask, torch.float32), sa_m, sa_k,
                              ptr(B, torch.float32), sb_k, sb_n, ptr(C, torch.float32), ldc,
                              ptr(bias, torch.float32), ptr(rowsum, torch.float32), ptr(cmask, torch.float32),
                              1 if relu else 0,
                              1 if accumulate else 0, split_k, ptr(ws), nbytes, _stream())
    _check(rc, 'psvae_gemm_f32')


def affine_merge_fwd(w_last, b_last, w_out, b_out, w_eff, b_eff):
    C, J = w_last.shape
    K = w_out.shape[1]
    _check(lib().psvae_affine_merge_fwd(C, J, K, ptr(w_last, torch.float32), ptr(b_last, torch.float32),
                                        ptr(w_out, torch.float32), ptr(b_out, torch.float32),
                                        ptr(w_eff, torch.float32), ptr(b_eff, torch.float32), _stream()),
           'psvae_affine_merge_fwd')


def affine_merge_bwd(w_last, w_out, b_out, g_w_eff, g_b_eff, g_w_last, g_w_out, g_b_out):
    C, J = w_last.shape
    K = w_out.shape[1]
    _check(lib().psvae_affine_merge_bwd(C, J, K, ptr(w_last, torch.float32), ptr(w_out, torch.float32),
                                        ptr(b_out, torch.float32), ptr(g_w_eff, torch.float32),
                                        ptr(g_b_eff, torch.float32), ptr(g_w_last, torch.float32),
                                        ptr(g_w_out, torch.float32), ptr(g_b_out, torch.float32), _stream()),
           'psvae_affine_merge_bwd')


def colsum(x, out, mask=None, accumulate=False):
    L = lib()
    M, N = x.shape
    px, ld = _rows(x)
    if mask is not None and _rows(mask)[1] != ld:
        raise ValueError('mask must have the row stride of x')
    nbytes = L.psvae_colsum_workspace_bytes(M, N)
    ws = workspace(nbytes, x.device) if nbytes else None
    _check(L.psvae_colsum_f32(M, N, px, ptr(mask, torch.float32), ld, ptr(out, torch.float32),
                              1 if accumulate else 0, ptr(ws), nbytes, _stream()), 'psvae_colsum_f32')


def gine_aggregate_fwd(x, table, row_ptr, col, cls, out):
    V, H = x.shape
    if not _CTYPES_ONLY:
        global CALLS
        CALLS += 1
        with _Prof('gine_aggregate_fwd_kernel', aggregate_bytes(V, col.numel(), H)):
            ops().gine_aggregate_fwd(x, table, row_ptr, col, cls, out)
        return
    px, ldx = _rows(x)
    po, ldo = _rows(out)
    with _Prof('gine_aggregate_fwd_kernel', aggregate_bytes(V, col.numel(), H)):
        rc = lib().psvae_gine_aggregate_fwd(V, H, table.shape[0], px, ldx, ptr(table, torch.float32),
                                            ptr(row_ptr, torch.int32), ptr(col, torch.int32),
                                            ptr(cls, torch.uint8), po, ldo, _stream())
    _check(rc, 'psvae_gine_aggregate_fwd')


def gine_aggregate_bwd(x, table, g, t_row_ptr, t_col, t_cls, g_x, g_table, add=None, accumulate_table=False):
    V, H = x.shape
    if not _CTYPES_ONLY:
        global CALLS
        CALLS += 1
        with _Prof('gine_aggregate_bwd_kernel', aggregate_bytes(V, t_col.numel(), H, backward=True)):
            ops().gine_aggregate_bwd(x, table, g, t_row_ptr, t_col, t_cls, add, g_x, g_table, accumulate_table)
        return
    L = lib()
    C = table.shape[0]
    px, ldx = _rows(x)
    pg, ldg = _rows(g)
    pgx, ldgx = _rows(g_x)
    padd, ldadd = _rows(add) if add is not None else (None, H)
    nbytes = L.psvae_gine_aggregate_bwd_workspace_bytes(V, H, C)
    ws = workspace(nbytes, x.device) if nbytes else None
    with _Prof('gine_aggregate_bwd_kernel', aggregate_bytes(V, t_col.numel(), H, backward=True)):
        rc = L.psvae_gine_aggregate_bwd(V, H, C, px, ldx, ptr(table, torch.float32), pg, ldg,
                                        ptr(t_row_ptr, torch.int32), ptr(t_col, torch.int32),
                                        ptr(t_cls, torch.uint8), padd, ldadd, pgx, ldgx,
                                        ptr(g_table, torch.float32), 1 if accumulate_table else 0,
                                        ptr(ws), nbytes, _stream())
    _check(rc, 'psvae_gine_aggregate_bwd')


def gine_aggregate_bwd_noalloc(x, table, g, t_row_ptr, t_col, t_cls, g_x, g_table, add=None, accumulate_table=False):
    """gine_aggregate_bwd through the entry point that allocates nothing (fixed scratch inside
    the library; calls on one GPU must not overlap)."""
    V, H = x.shape
    px, ldx = _rows(x)
    pg, ldg = _rows(g)
    pgx, ldgx = _rows(g_x)
    padd, ldadd = _rows(add) if add is not None else (None, H)
    rc = lib().psvae_gine_aggregate_bwd_noalloc(V, H, table.shape[0], px, ldx, ptr(table, torch.float32), pg, ldg,
                                                ptr(t_row_ptr, torch.int32), ptr(t_col, torch.int32),
                                                ptr(t_cls, torch.uint8), padd, ldadd, pgx, ldgx,
                                                ptr(g_table, torch.float32), 1 if accumulate_table else 0, _stream())
    _check(rc, 'psvae_gine_aggregate_bwd_noalloc')


def embed_atom_fwd(x, xp, xq, ea, ep, eq, out):
    _check(lib().psvae_embed_atom_fwd(x.numel(), ptr(x, torch.int32), ptr(xp, torch.int32), ptr(xq, torch.int32),
                                      ptr(ea, torch.float32), ea.shape[0], ea.shape[1],
                                      ptr(ep, torch.float32), ep.shape[0], ep.shape[1],
                                      ptr(eq, torch.float32), eq.shape[0], eq.shape[1],
                                      ptr(out, torch.float32), _stream()), 'psvae_embed_atom_fwd')


def embed_atom_onehot(x, xp, xq, n_atom, n_piece, n_pos, out):
    _check(lib().psvae_embed_atom_onehot(x.numel(), ptr(x, torch.int32), ptr(xp, torch.int32), ptr(xq, torch.int32),
                                         n_atom, n_piece, n_pos, ptr(out, torch.float32), out.stride(0), _stream()),
           'psvae_embed_atom_onehot')


def embed_atom_bwd(x, xp, xq, g, ga, gp, gq):
    _check(lib().psvae_embed_atom_bwd(x.numel(), ptr(x, torch.int32), ptr(xp, torch.int32), ptr(xq, torch.int32),
                                      ptr(g, torch.float32), ga.shape[0], ga.shape[1], gp.shape[0], gp.shape[1],
                                      gq.shape[0], gq.shape[1], ptr(ga, torch.float32), ptr(gp, torch.float32),
                                      ptr(gq, torch.float32), _stream()), 'psvae_embed_atom_bwd')


def embedding_fwd(ids, table, out):
    _check(lib().psvae_embedding_fwd(ids.numel(), table.shape[1], table.shape[0], ptr(ids, torch.int32),
                                     ptr(table, torch.float32), ptr(out, torch.float32), _stream()),
           'psvae_embedding_fwd')


def onehot(ids, out):
    _check(lib().psvae_onehot_f32(ids.numel(), ptr(ids, torch.int32), ptr(out, torch.float32), out.stride(0), _stream()),
           'psvae_onehot_f32')


def embed_trans_fwd(x, xp, xq, sizes, T, out):
    na, npc, nq = sizes
    po, ldo = _rows(out)
    if T.shape[0] != na + npc + nq or T.shape[1] != out.shape[1] or not T.is_contiguous():
        raise ValueError('embed_trans_fwd: T must be a contiguous [n_atom + n_piece + n_pos, H] table')
    if not (x.numel() == xp.numel() == xq.numel() == out.shape[0]):
        raise ValueError('embed_trans_fwd: id vectors and output rows differ in length')
    _check(lib().psvae_embed_trans_fwd(x.numel(), out.shape[1], ptr(x, torch.int32), ptr(xp, torch.int32),
                                       ptr(xq, torch.int32), na, npc, nq, ptr(T, torch.float32), po, ldo, _stream()),
           'psvae_embed_trans_fwd')


def segment_sum_fwd(x, seg_begin, seg_end, out):
    px, ldx = _rows(x)
    po, ldo = _rows(out)
    _check(lib().psvae_segment_sum_fwd(out.shape[0], x.shape[1], px, ldx, ptr(seg_begin, torch.int32),
                                       ptr(seg_end, torch.int32), po, ldo, _stream()), 'psvae_segment_sum_fwd')


def segment_sum_bwd(g_out, seg_begin, seg_end, g_x):
    pg, ldg = _rows(g_out)
    px, ldx = _rows(g_x)
    _check(lib().psvae_segment_sum_bwd(g_out.shape[0], g_out.shape[1], pg, ldg, ptr(seg_begin, torch.int32),
                                       ptr(seg_end, torch.int32), px, ldx, _stream()), 'psvae_segment_sum_bwd')


def rsample_kl_fwd(mean, lv_pre, eps, z, kl):
    B, Ld = mean.shape
    _check(lib().psvae_rsample_kl_fwd(B, Ld, ptr(mean, torch.float32), ptr(lv_pre, torch.float32),
                                      ptr(eps, torch.float32), ptr(z, torch.float32), ptr(kl, torch.float32),
                                      _stream()), 'psvae_rsample_kl_fwd')


def rsample_kl_bwd(mean, lv_pre, eps, g_z, g_kl, g_mean, g_lv_pre):
    B, Ld = mean.shape
    _check(lib().psvae_rsample_kl_bwd(B, Ld, ptr(mean, torch.float32), ptr(lv_pre, torch.float32),
                                      ptr(eps, torch.float32), ptr(g_z, torch.float32), ptr(g_kl, torch.float32),
                                      ptr(g_mean, torch.float32), ptr(g_lv_pre, torch.float32), _stream()),
           'psvae_rsample_kl_bwd')


def gru_cell_fwd(gi, gh, h_prev, h, gates):
    B, H = h_prev.shape
    f = torch.float32
    _check(lib().psvae_gru_cell_fwd(B, H, ptr(gi, f), ptr(gh, f), ptr(h_prev, f), ptr(h, f), ptr(gates, f),
                                    _stream()), 'psvae_gru_cell_fwd')


def gru_cell_bwd(gates, gh, h_prev, dh, dh_extra, dgi, dgh, dh_prev):
    B, H = h_prev.shape
    f = torch.float32
    _check(lib().psvae_gru_cell_bwd(B, H, ptr(gates, f), ptr(gh, f), ptr(h_prev, f), ptr(dh, f), ptr(dh_extra, f),
                                    ptr(dgi, f), ptr(dgh, f), ptr(dh_prev, f), _stream()), 'psvae_gru_cell_bwd')


GRU_SEQ_MAX_H = 256


def gru_seq_fwd(gi, h0, w_hh, b_hh, hall, gates, ghn):
    T, B, H3 = gi.shape
    f = torch.float32
    _check(lib().psvae_gru_seq_fwd(T, B, H3 // 3, ptr(gi, f), ptr(h0, f), ptr(w_hh, f), ptr(b_hh, f), ptr(hall, f),
                                   ptr(gates, f), ptr(ghn, f), _stream()), 'psvae_gru_seq_fwd')


def gru_seq_bwd(gates, ghn, hall, g_out, w_hh_t, dgi, dgh, dh0):
    T, B, H3 = gates.shape
    f = torch.float32
    _check(lib().psvae_gru_seq_bwd(T, B, H3 // 3, ptr(gates, f), ptr(ghn, f), ptr(hall, f), ptr(g_out, f),
                                   ptr(w_hh_t, f), ptr(dgi, f), ptr(dgh, f), ptr(dh0, f), _stream()),
           'psvae_gru_seq_bwd')


def pair_gather_fwd(h, z, src, dst, graph, feats):
    ph, ldh = _rows(h)
    pz, ldz = _rows(z)
    _check(lib().psvae_pair_gather_fwd(src.numel(), h.shape[1], z.shape[1], ph, ldh, pz, ldz,
                                       ptr(src, torch.int32), ptr(dst, torch.int32), ptr(graph, torch.int32),
                                       ptr(feats, torch.float32), _stream()), 'psvae_pair_gather_fwd')


def pair_gather_bwd(g_feats, src, dst, pair_ptr, node_begin, node_end, g_h, g_z):
    pgh, ldgh = _rows(g_h)
    pgz, ldgz = _rows(g_z)
    _check(lib().psvae_pair_gather_bwd(g_z.shape[0], g_h.shape[1], g_z.shape[1], ptr(g_feats, torch.float32),
                                       ptr(src, torch.int32), ptr(dst, torch.int32), ptr(pair_ptr, torch.int32),
                                       ptr(node_begin, torch.int32), ptr(node_end, torch.int32), pgh, ldgh,
                                       pgz, ldgz, _stream()), 'psvae_pair_gather_bwd')


def pair_combine_fwd(a, b, c, src, dst, graph, out, relu=True):
    if not _CTYPES_ONLY:
        global CALLS
        CALLS += 1
        return ops().pair_combine_fwd(a, b, c, src, dst, graph, relu, out)
    pa, lda = _rows(a)
    pb, ldb = _rows(b)
    pc, ldc = _rows(c)
    po, ldo = _rows(out)
    _check(lib().psvae_pair_combine_fwd(src.numel(), out.shape[1], pa, lda, pb, ldb, pc, ldc, ptr(src, torch.int32),
                                        ptr(dst, torch.int32), ptr(graph, torch.int32), 1 if relu else 0, po, ldo,
                                        _stream()), 'psvae_pair_combine_fwd')


def pair_combine_bwd(g_out, out, relu, src, dst, pair_ptr, node_begin, node_end, g_ab, g_c):
    if not _CTYPES_ONLY:
        global CALLS
        CALLS += 1
        return ops().pair_combine_bwd(g_out, out, relu, src, dst, pair_ptr, node_begin, node_end, g_ab, g_c)
    B, F = g_c.shape
    P = src.numel()
    pg, ldg = _rows(g_out) if P else (None, F)
    po, ldo = _rows(out) if P else (None, F)
    pab, ldab = _rows(g_ab)
    pc, ldc = _rows(g_c)
    _check(lib().psvae_pair_combine_bwd(B, F, pg, ldg, po, ldo, 1 if relu else 0, ptr(src, torch.int32),
                                        ptr(dst, torch.int32), ptr(pair_ptr, torch.int32), ptr(node_begin, torch.int32),
                                        ptr(node_end, torch.int32), pab, ldab, pc, ldc, _stream()), 'psvae_pair_combine_bwd')


def link_ce_fwd(hid, W, b, golden, logits, dlogits, loss, n_correct):
    if not _CTYPES_ONLY:
        global CALLS
        CALLS += 1
        return ops().link_ce_fwd(hid, W, b, golden, logits, dlogits, loss, n_correct)
    L = lib()
    P, F = hid.shape
    ph, ldh = _rows(hid) if P else (None, F)
    nbytes = L.psvae_link_ce_workspace_bytes(P)
    ws = workspace(nbytes, W.device)
    _check(L.psvae_link_ce_fwd(P, F, W.shape[0], ph, ldh, ptr(W, torch.float32), ptr(b, torch.float32),
                               ptr(golden, torch.int32), ptr(logits, torch.float32), ptr(dlogits, torch.float32),
                               ptr(loss, torch.float32), ptr(n_correct, torch.int32), ptr(ws), nbytes, _stream()),
           'psvae_link_ce_fwd')


def link_ce_bwd_input(dlogits, W, g_loss, g_hid):
    if not _CTYPES_ONLY:
        global CALLS
        CALLS += 1
        return ops().link_ce_bwd_input(dlogits, W, g_loss, g_hid)
    P, C = dlogits.shape
    pg, ldg = _rows(g_hid) if P else (None, W.shape[1])
    _check(lib().psvae_link_ce_bwd_input(P, W.shape[1], C, ptr(dlogits, torch.float32), ptr(W, torch.float32),
                                         ptr(g_loss, torch.float32), pg, ldg, _stream()), 'psvae_link_ce_bwd_input')


def token_ce_fwd(logits, gold, ignore_index, dlogits, loss, counts):
    """logits [R,V] (rows of any stride), gold int32 [R] -> dlogits [R,V], loss f32[2], counts i32[2]."""
    L = lib()
    R, V = logits.shape
    pl, ld = _rows(logits) if R else (None, V)
    nbytes = L.psvae_token_ce_workspace_bytes(R)
    ws = workspace(nbytes, loss.device)
    _check(L.psvae_token_ce_fwd(R, V, pl, ld, ptr(gold, torch.int32), int(ignore_index), ptr(dlogits, torch.float32),
                                ptr(loss, torch.float32), ptr(counts, torch.int32), ptr(ws), nbytes, _stream()),
           'psvae_token_ce_fwd')


def link_infer(logits, add_edge_th, none_bond, conf, etype, keep):
    P, C = logits.shape
    _check(lib().psvae_link_infer(P, C, ptr(logits, torch.float32), float(add_edge_th), int(none_bond),
                                  ptr(conf, torch.float32), ptr(etype, torch.int32), ptr(keep, torch.uint8),
                                  _stream()), 'psvae_link_infer')


def assemble_links(node_off, atom_type, block0, n_blocks, row_ptr, col, cls, order_ptr, ord_m, ord_n, ord_type,
                   max_valence, sulfur_idx, accepted, block, keep_atom, status):
    L = lib()
    i32, u8 = torch.int32, torch.uint8
    B, V = n_blocks.numel(), atom_type.numel()
    if node_off.numel() != B + 1 or order_ptr.numel() != B + 1 or row_ptr.numel() != V + 1:
        raise ValueError('assemble_links: offset arrays do not match B / V')
    if not (block0.numel() == block.numel() == keep_atom.numel() == V and status.numel() == B):
        raise ValueError('assemble_links: per-atom / per-molecule array sizes do not match')
    if not (ord_m.numel() == ord_n.numel() == ord_type.numel() == accepted.numel()):
        raise ValueError('assemble_links: candidate arrays differ in length')
    nbytes = L.psvae_assemble_links_workspace_bytes(V)
    ws = workspace(nbytes, n_blocks.device) if nbytes else None
    _check(L.psvae_assemble_links(B, V, ptr(node_off, i32), ptr(atom_type, i32), ptr(block0, i32), ptr(n_blocks, i32),
                                  ptr(row_ptr, i32), ptr(col, i32), ptr(cls, u8), ptr(order_ptr, i32),
                                  ptr(ord_m, i32), ptr(ord_n, i32), ptr(ord_type, i32), ptr(max_valence, i32),
                                  max_valence.numel(), int(sulfur_idx), ptr(accepted, u8), ptr(block, i32),
                                  ptr(keep_atom, u8), ptr(status, i32), ptr(ws), nbytes, _stream()),
           'psvae_assemble_links')


def exclusive_scan(inp, out):
    _check(lib().psvae_exclusive_scan_i32(inp.numel(), ptr(inp, torch.int32), ptr(out, torch.int32), _stream()),
           'psvae_exclusive_scan_i32')


def collate_count(B, N, mol_ids, packed, rng_off, uniforms, keep_prob, pair_cnt, pair_off):
    _check(lib().psvae_collate_count(B, N, ptr(mol_ids, torch.int32), ptr(packed.atom_off), ptr(packed.atom_pos),
                                     ptr(packed.bond_off), ptr(packed.bond_a), ptr(packed.bond_b),
                                     ptr(packed.bond_cls), ptr(rng_off, torch.int32), ptr(uniforms, torch.float64),
                                     float(keep_prob), ptr(pair_cnt, torch.int32), ptr(pair_off, torch.int32),
                                     _stream()), 'psvae_collate_count')


def collate_fill(B, N, S, mol_ids, packed, offs, E, uniforms, keep_prob, atom_pad, piece_pad, num_bond_type,
                 ref, compact):
    """offs: dict node_off/edge_off/inedge_off/rng_off/pair_off; ref / compact: dicts of output
    tensors (missing key -> NULL -> that tensor is not written)."""
    r = lambda k, dt: ptr(ref.get(k), dt)          # noqa: E731
    c = lambda k, dt: ptr(compact.get(k), dt)      # noqa: E731
    i32, i64, u8, f32 = torch.int32, torch.int64, torch.uint8, torch.float32
    rc = lib().psvae_collate_fill(
        B, N, S, ptr(mol_ids, i32), ptr(packed.atom_off), ptr(packed.atom_type), ptr(packed.atom_pos),
        ptr(packed.bond_off), ptr(packed.bond_a), ptr(packed.bond_b), ptr(packed.bond_cls),
        ptr(packed.piece_off), ptr(packed.piece_id), ptr(packed.props, f32),
        ptr(offs['node_off'], i32), ptr(offs['edge_off'], i32), ptr(offs['inedge_off'], i32),
        ptr(offs['rng_off'], i32), ptr(offs['pair_off'], i32), int(E), ptr(uniforms, torch.float64),
        float(keep_prob), int(atom_pad), int(piece_pad),
        r('x', i64), r('x_pieces', i64), r('x_pos', i64), r('atom_mask', u8), r('pieces', i64),
        r('edge_index', i64), r('edge_attr', i64), int(num_bond_type), r('edge_select', u8),
        r('golden_edge', i64), r('in_piece_edge_idx', i64), r('props', f32),
        c('cx', i32), c('cxp', i32), c('cxpos', i32), c('row_ptr', i32), c('col', i32), c('cls', u8),
        c('in_row_ptr', i32), c('in_col', i32), c('in_cls', u8),
        c('pair_src', i32), c('pair_dst', i32), c('pair_graph', i32), c('golden32', i32), _stream())
    _check(rc, 'psvae_collate_fill')


def clip_adam_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0,
                   max_norm=0.0, write_grad=True, norm_out=None):
    """Scale + global-norm clip + Adam on flat f32 buffers (psvae_clip_adam_step)."""
    L = lib()
    n = p.numel()
    if not (g.numel() == m.numel() == v.numel() == n):
        raise ValueError('clip_adam_step: buffers differ in length')
    nbytes = L.psvae_clip_adam_workspace_bytes(n)
    ws = workspace(nbytes, p.device) if nbytes else None
    f = torch.float32
    _check(L.psvae_clip_adam_step(n, ptr(p, f), ptr(g, f), ptr(m, f), ptr(v, f), float(grad_scale), float(max_norm or 0.0),
                                  float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                  int(step), 1 if write_grad else 0, ptr(norm_out, f), ptr(ws), nbytes, _stream()),
           'psvae_clip_adam_step')
