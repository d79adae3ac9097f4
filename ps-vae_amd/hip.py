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
    'psvae_key_sort_workspace_bytes': (_c_size, [_c_int, _c_int]),
    'psvae_key_sort': (_c_int, [_c_int, _c_int, _p, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p, _c_size, _p]),
    'psvae_keyed_row_sum_workspace_bytes': (_c_size, [_c_ll, _c_int, _c_int, _c_int]),
    'psvae_keyed_row_sum': (_c_int, [_c_int, _c_int, _c_ll, _c_int, _p, _c_int, _p, _p, _p, _p, _c_int, _p, _c_size, _p]),
    'psvae_gine_aggregate_fwd': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _p, _p, _p, _p, _c_int, _p]),
    'psvae_gine_mask_bytes': (_c_size, [_c_int, _c_int, _c_int]),
    'psvae_gine_aggregate_fwd_bits': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _p, _p, _p, _p, _c_int, _p, _p]),
    'psvae_gine_aggregate_bwd_bits': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _c_int, _p, _p, _p, _p, _c_int, _p, _c_int, _p,
                                               _p, _c_int, _p, _c_int, _p, _c_size, _p]),
    'psvae_gine_aggregate_bwd_graph_add': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _p, _c_int, _p, _p, _p, _p, _c_int,
                                                    _p, _c_int, _p, _p, _c_int, _p, _c_int, _p, _c_size, _p]),
    'psvae_graph_rows_gated': (_c_int, [_c_int, _c_int, _p, _c_int, _p, _p, _p, _c_int, _p]),
    'psvae_gine_aggregate_bwd_workspace_bytes': (_c_size, [_c_int, _c_int, _c_int]),
    'psvae_gine_aggregate_bwd_slabs': (_c_int, [_c_int, _c_int, _c_int]),
    'psvae_gine_table_reduce': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _c_int, _p]),
    'psvae_gine_aggregate_bwd': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _p, _c_int, _p, _p, _p, _p,
                                          _c_int, _p, _c_int, _p, _c_int, _p, _c_size, _p]),
    'psvae_gine_aggregate_bwd_noalloc': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _p, _c_int, _p, _p, _p, _p,
                                                  _c_int, _p, _c_int, _p, _c_int, _p]),
    'psvae_gemm_f32_workspace_bytes': (_c_size, [_c_int, _c_int, _c_int, _c_int]),
    'psvae_gemm_f32': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _c_i64, _c_i64, _p, _c_i64, _c_i64, _p, _c_i64,
                                _p, _p, _p, _c_int, _c_int, _c_int, _p, _c_size, _p]),
    'psvae_pair_combine_fwd': (_c_int, [_c_int, _c_int, _p, _c_int, _p, _c_int, _p, _c_int, _p, _p, _p, _c_int, _p,
                                        _c_int, _p]),
    'psvae_block_table': (_c_int, [_c_int] * 6 + [_p, _p, _p, _p, _p]),
    'psvae_mse_selected_fwd': (_c_int, [_c_int, _c_int, _p, _c_int, _p, _c_int, _p, _p, _p, _p]),
    'psvae_mse_selected_bwd': (_c_int, [_c_int, _p, _p, _p, _p]),
    'psvae_loss_mix_fwd': (_c_int, [_p, _p, _p, _c_f32, _c_f32, _c_f32, _p, _p, _p]),
    'psvae_loss_mix_bwd': (_c_int, [_p, _c_f32, _c_f32, _c_f32, _p, _p, _p]),
    'psvae_affine_merge_fwd': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_affine_merge_bwd': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_colsum_workspace_bytes': (_c_size, [_c_int, _c_int]),
    'psvae_colsum_f32': (_c_int, [_c_int, _c_int, _p, _p, _c_i64, _p, _c_int, _p, _c_size, _p]),
    'psvae_segment_sum_fwd': (_c_int, [_c_int, _c_int, _p, _c_int, _p, _p, _p, _c_int, _p]),
    'psvae_segment_sum_bwd': (_c_int, [_c_int, _c_int, _p, _c_int, _p, _p, _p, _c_int, _p]),
    'psvae_rsample_kl_workspace_bytes': (_c_size, [_c_int, _c_int]),
    'psvae_rsample_kl_fwd': (_c_int, [_c_int, _c_int, _p, _p, _p, _p, _p, _p, _c_size, _p]),
    'psvae_rsample_kl_bwd': (_c_int, [_c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_gru_cell_fwd': (_c_int, [_c_int, _c_int, _p, _p, _p, _p, _p, _p]),
    'psvae_gru_cell_bwd': (_c_int, [_c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_gru_seq_fwd': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_gru_seq_bwd': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_pair_gather_fwd': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _c_int, _p, _p, _p, _p, _p]),
    'psvae_pair_gather_bwd': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _c_int, _p, _c_int, _p]),
    'psvae_link_ce_workspace_bytes': (_c_size, [_c_int]),
    'psvae_link_ce_fwd': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _c_size, _p]),
    'psvae_link_ce_bwd_input': (_c_int, [_c_int, _c_int, _c_int, _p, _p, _p, _p, _c_int, _p]),
    'psvae_pair_combine_fwd_graphs': (_c_int, [_c_int, _c_int, _c_int, _p, _c_int, _p, _c_int, _p, _c_int, _p, _p, _p, _p, _p,
                                               _c_int, _p, _c_int, _p]),
    'psvae_pair_combine_bwd': (_c_int, [_c_int, _c_int, _p, _c_int, _p, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _c_int, _p,
                                        _c_int, _p]),
    'psvae_gemm_f32_bits_words': (_c_size, [_c_int, _c_int]),
    'psvae_gemm_f32_bits': (_c_int, [_c_int] * 3 + [_p, _c_i64, _c_i64, _p, _c_i64, _c_i64, _p, _c_i64, _p, _c_int, _p, _p, _p]),
    'psvae_token_ce_workspace_bytes': (_c_size, [_c_int]),
    'psvae_token_ce_fwd': (_c_int, [_c_int, _c_int, _p, _c_i64, _p, _c_int, _p, _c_i64, _p, _p, _p, _c_size, _p]),
    'psvae_link_infer': (_c_int, [_c_int, _c_int, _p, _c_f32, _c_int, _p, _p, _p, _p]),
    'psvae_gemm_f32_relu_classes_tiles': (_c_int, [_c_int]),
    'psvae_gemm_f32_relu_classes': (_c_int, [_c_int, _c_int, _c_int, _p, _c_i64, _p, _c_i64, _p, _p, _p, _p]),
    'psvae_pair_enumerate_count': (_c_int, [_c_int, _c_int, _p, _p, _p, _p]),
    'psvae_pair_enumerate_fill': (_c_int, [_c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'psvae_piece_draw': (_c_int, [_c_int, _c_int, _p, _c_i64, _c_f32, _p, _c_i64, _c_int, _c_int, _c_int, _p, _c_int, _c_int,
                                  _p, _p, _p, _p, _c_i64, _p, _p]),
    'psvae_embed_trans_fwd': (_c_int, [_c_int, _c_int, _p, _p, _p, _c_int, _c_int, _c_int, _p, _p, _c_int, _p]),
    'psvae_clip_adam_workspace_bytes': (_c_size, [_c_ll]),
    'psvae_clip_adam_step': (_c_int, [_c_ll, _p, _p, _p, _p, _c_f32, _c_f32] + [_c_f64] * 5 + [_c_int, _c_int, _p, _p, _c_size, _p]),
    'psvae_clip_adam_step_dev': (_c_int, [_c_ll, _p, _p, _p, _p, _c_f32, _c_f32] + [_c_f64] * 4 + [_p, _c_int, _p, _p, _c_size, _p]),
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
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


_ops = None


def ops():
    """torch.ops.psvae (the operators csrc/libpsvae_torch.so registers); raises if it is not built.
    The kernel library is loaded first, so the operator library binds to that one (also when
    PSVAE_HIP_LIB names an A/B build of it: the dependency is resolved by its soname)."""
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


def _count():
    global CALLS
    CALLS += 1


CALLS = 0   # C-ABI calls made so far (bench.py reports calls per step)


def _check(rc, name):
    _count()
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
    # the ctypes route launches on the CURRENT device and takes its current stream (_stream()):
    # a tensor that lives on another GPU would be touched from the wrong device
    if t.device.index != _cur_device():
        raise RuntimeError(f'tensor on cuda:{t.device.index} while cuda:{_cur_device()} is the current device: '
                           'make the tensor\'s device current (torch.cuda.device) around the call')
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


def aggregate_bytes(V, E, H, backward=False, mask_bytes=0):
    """Algorithmic bytes of one GINE aggregation: per directed edge one neighbour row (4H),
    its int32 column index and 1-byte bond class; per node its own row in, the output row out
    and an int32 row pointer.  The backward that recomputes the ReLU decisions also reads the saved
    input row (V * 4H); the one that takes them as bits reads `mask_bytes` per node instead."""
    extra = 0
    if backward:
        extra = V * mask_bytes if mask_bytes else V * 4 * H
    return E * (4 * H + 5) + V * (8 * H + 4) + extra


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
        hbb['work_note'] = ('bytes the kernel has to move: one gathered gradient row, index and class per edge; own '
                            'gradient row in, result row out, row pointer and (when the ReLU decisions arrive as bits) '
                            'the mask words per node - not the 4H bytes of an x row per node that a backward '
                            'recomputing the decisions would read')
        out['roofline_hbm_bwd'] = hbb
    out['event_ms_total'] = {k: round(v['ms'], 3) for k, v in sorted(groups.items())}
    return out


# --------------------------------------------------------------------------- raw calls
def gemm(M, N, K, A, sa_m, sa_k, B, sb_k, sb_n, C, ldc, bias=None, relu=False, accumulate=False,
         amask=None, split_k=0, rowsum=None, cmask=None):
    tag = None
    if PROFILE is not None:
        tag = (M, N, K, 'k' if sa_k == 1 else 'm', 'k' if sb_k == 1 else 'n', amask is not None, rowsum is not None)
    _count()
    with _Prof('gemm_f32_kernel', 2.0 * M * N * K, tag):
        ops().gemm(M, N, K, A, amask, sa_m, sa_k, B, sb_k, sb_n, C, ldc, bias, rowsum, cmask, relu, accumulate, split_k)


def gemm_bits_words(M, N):
    """64-bit words of the sign-bit array of an [M, N] result (psvae_gemm_f32_bits_words)."""
    return ((M + 31) // 32) * ((N + 31) // 32) * 16


def gemm_bits(M, N, K, A, sa_m, sa_k, B, sb_k, sb_n, C, ldc, bias=None, relu=False, relu_bits=None, gate_bits=None):
    """psvae_gemm_f32_bits: C = act(A B + bias) leaving (C > 0) behind as bits (relu_bits, int64 storage)
    and / or zeroed where the bits of gate_bits are clear."""
    tag = None
    if PROFILE is not None:
        tag = (M, N, K, 'k' if sa_k == 1 else 'm', 'k' if sb_k == 1 else 'n', False, False)
    _count()
    with _Prof('gemm_f32_kernel', 2.0 * M * N * K, tag):
        ops().gemm_bits(M, N, K, A, sa_m, sa_k, B, sb_k, sb_n, C, ldc, bias, relu, relu_bits, gate_bits)


def block_table(ea, ep, eq, Z):
    _count()
    ops().block_table(ea, ep, eq, Z)


def mse_selected_fwd(pred, props, sel, diff, loss):
    _count()
    ops().mse_selected_fwd(pred, props, sel, diff, loss)


def mse_selected_bwd(diff, g, g_pred):
    _count()
    ops().mse_selected_bwd(diff, g, g_pred)


def loss_mix_fwd(rec, pred, kl, alpha, beta, out):
    """beta: a Python number or a device scalar."""
    _count()
    dev = beta if torch.is_tensor(beta) else None
    ops().loss_mix_fwd(rec, pred, kl, float(alpha), float(1 - alpha), 0.0 if dev is not None else float(beta), dev, out)


def loss_mix_bwd(g, alpha, beta, g3):
    _count()
    dev = beta if torch.is_tensor(beta) else None
    ops().loss_mix_bwd(g, float(alpha), float(1 - alpha), 0.0 if dev is not None else float(beta), dev, g3)


def affine_merge_fwd(w_last, b_last, w_out, b_out, w_eff, b_eff):
    _count()
    ops().affine_merge_fwd(w_last, b_last, w_out, b_out, w_eff, b_eff)


def affine_merge_bwd(w_last, w_out, b_out, g_w_eff, g_b_eff, g_w_last, g_w_out, g_b_out):
    _count()
    ops().affine_merge_bwd(w_last, w_out, b_out, g_w_eff, g_b_eff, g_w_last, g_w_out, g_b_out)


def colsum(x, out, mask=None, accumulate=False):
    _count()
    ops().colsum(x, mask, out, accumulate)


def gine_aggregate_fwd(x, table, row_ptr, col, cls, out):
    V, H = x.shape
    _count()
    with _Prof('gine_aggregate_fwd_kernel', aggregate_bytes(V, col.numel(), H)):
        ops().gine_aggregate_fwd(x, table, row_ptr, col, cls, out)


def gine_aggregate_bwd(x, table, g, t_row_ptr, t_col, t_cls, g_x, g_table, add=None, accumulate_table=False, seg_add=None,
                       node_graph=None):
    """seg_add [graphs, H] / node_graph [V]: a per-graph row added to every node of the graph
    (psvae_gine_aggregate_bwd_graph_add)."""
    V, H = x.shape
    _count()
    with _Prof('gine_aggregate_bwd_kernel', aggregate_bytes(V, t_col.numel(), H, backward=True)):
        ops().gine_aggregate_bwd(x, table, g, t_row_ptr, t_col, t_cls, add, seg_add, node_graph, g_x, g_table,
                                 accumulate_table)


def gine_mask_words(V, H, C):
    """(number of words, dtype) of the mask-bit array of gine_aggregate_fwd_bits: one word per float4
    column of a node, of the width psvae_gine_mask_bytes reports (int16 or int32 storage)."""
    words = V * (H // 4)
    width = lib().psvae_gine_mask_bytes(max(V, 1), H, C) // (max(V, 1) * (H // 4))
    return words, (torch.int16 if width == 2 else torch.int32)


def gine_aggregate_fwd_bits(x, table, row_ptr, col, cls, out, mask):
    """psvae_gine_aggregate_fwd_bits: the aggregation, leaving the ReLU decisions of the messages behind
    as bits in `mask` (gine_mask_words)."""
    V, H = x.shape
    _count()
    with _Prof('gine_aggregate_fwd_kernel', aggregate_bytes(V, col.numel(), H)):
        ops().gine_aggregate_fwd_bits(x, table, row_ptr, col, cls, out, mask)


def gine_aggregate_bwd_bits(mask, C, g, t_row_ptr, t_col, t_cls, g_x, g_table, add=None, seg_add=None, node_graph=None,
                            accumulate_table=False, slabs=None):
    """psvae_gine_aggregate_bwd_bits: the backward from the mask bits (x and the table are not read);
    seg_add [graphs, H] / node_graph [V]: a per-graph row added to every node of the graph.
    slabs [gine_bwd_slabs(V, H, C), C, H]: the table-gradient partials stay there and g_table is not
    touched (gine_table_reduce sums the slabs of several calls at once)."""
    V, H = g.shape
    _count()
    with _Prof('gine_aggregate_bwd_kernel', aggregate_bytes(V, t_col.numel(), H, backward=True,
                                                            mask_bytes=mask.numel() * mask.element_size() // max(V, 1))):
        ops().gine_aggregate_bwd_bits(mask, C, g, t_row_ptr, t_col, t_cls, add, seg_add, node_graph, g_x, g_table,
                                      accumulate_table, slabs)


def gine_bwd_slabs(V, H, C):
    return int(lib().psvae_gine_aggregate_bwd_slabs(int(V), int(H), int(C)))


def gine_table_reduce(slabs, g_table, accumulate=False):
    """g_table (+)= sum of the [C,H] slabs in slab order (psvae_gine_table_reduce)."""
    _count()
    ops().gine_table_reduce(slabs, g_table, accumulate)


def graph_rows_gated(q, node_graph, gate_bits, out):
    """out[v] = q[node_graph[v]] where the sign bit of (v, n) is set, else 0 (psvae_graph_rows_gated)."""
    _count()
    ops().graph_rows_gated(q, node_graph, gate_bits, out)


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
    _count()
    ops().embed_atom_onehot(x, xp, xq, n_atom, n_piece, n_pos, out)


def embed_atom_bwd(x, xp, xq, g, ga, gp, gq):
    _check(lib().psvae_embed_atom_bwd(x.numel(), ptr(x, torch.int32), ptr(xp, torch.int32), ptr(xq, torch.int32),
                                      ptr(g, torch.float32), ga.shape[0], ga.shape[1], gp.shape[0], gp.shape[1],
                                      gq.shape[0], gq.shape[1], ptr(ga, torch.float32), ptr(gp, torch.float32),
                                      ptr(gq, torch.float32), _stream()), 'psvae_embed_atom_bwd')


def embedding_fwd(ids, table, out):
    _count()
    ops().embedding_fwd(ids, table, out)


def onehot(ids, out):
    _count()
    ops().onehot(ids, out)


KEY_CHUNK = 256   # entries per work unit of the keyed row sum
KEY_SORT_MAX_ROWS = 8192   # table rows psvae_key_sort accepts (all tables together)


def key_chunk(entries):
    """Entries per work unit for id vectors of `entries` rows: small units for a small batch, whose sums would
    otherwise sit on a handful of CUs (a CU receives a fixed share of the memory bandwidth, see gemm_panel.h)."""
    return 32 if entries <= 8192 else KEY_CHUNK


def key_sort(keys, sizes, chunk=None):
    """psvae_key_sort: keys = 1..3 int32 id vectors over the same V entries, sizes = rows of their
    tables.  Returns (perm [len(keys) * V], row_off [R + 1], unit_off [R + 1]) int32."""
    V, dev = keys[0].numel(), keys[0].device
    R = int(sum(sizes))
    perm = torch.empty(len(keys) * V, dtype=torch.int32, device=dev)
    row_off = torch.empty(R + 1, dtype=torch.int32, device=dev)
    unit_off = torch.empty(R + 1, dtype=torch.int32, device=dev)
    k = list(keys) + [None] * (3 - len(keys))
    n = [int(s) for s in sizes] + [0] * (3 - len(sizes))
    _count()
    ops().key_sort(k[0], k[1], k[2], n[0], n[1], n[2], key_chunk(V) if chunk is None else chunk, perm, row_off, unit_off)
    return perm, row_off, unit_off


def keyed_row_sum(g, perm, row_off, unit_off, out, chunk=None):
    """out[r] = sum of g[perm[i]] over i in [row_off[r], row_off[r+1]) (psvae_keyed_row_sum).  chunk: the one the
    sort was made with (None on both sides: key_chunk of the row count of g, which is the length of the id vectors)."""
    _count()
    ops().keyed_row_sum(g, perm, row_off, unit_off, key_chunk(g.shape[0]) if chunk is None else chunk, out)


def embed_trans_fwd(x, xp, xq, sizes, T, out):
    na, npc, nq = sizes
    _count()
    ops().embed_trans_fwd(x, xp, xq, na, npc, nq, T, out)


def segment_sum_fwd(x, seg_begin, seg_end, out):
    _count()
    ops().segment_sum_fwd(x, seg_begin, seg_end, out)


def segment_sum_bwd(g_out, seg_begin, seg_end, g_x):
    _count()
    ops().segment_sum_bwd(g_out, seg_begin, seg_end, g_x)


def rsample_kl_fwd(mean, lv_pre, eps, z, kl):
    _count()
    ops().rsample_kl_fwd(mean, lv_pre, eps, z, kl)


def rsample_kl_bwd(mean, lv_pre, eps, g_z, g_kl, g_mean, g_lv_pre):
    _count()
    ops().rsample_kl_bwd(mean, lv_pre, eps, g_z, g_kl, g_mean, g_lv_pre)


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
    _count()
    ops().gru_seq_fwd(gi, h0, w_hh, b_hh, hall, gates, ghn)


def gru_seq_bwd(gates, ghn, hall, g_out, w_hh_t, dgi, dgh, dh0):
    _count()
    ops().gru_seq_bwd(gates, ghn, hall, g_out, w_hh_t, dgi, dgh, dh0)


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
    _count()
    ops().pair_combine_fwd(a, b, c, src, dst, graph, relu, out)


def pair_combine_fwd_graphs(a, b, c, src, dst, pair_ptr, node_begin, node_end, max_nodes, out, relu=True):
    """psvae_pair_combine_fwd_graphs: one workgroup per graph, the graph's node rows staged in LDS."""
    _count()
    ops().pair_combine_fwd_graphs(a, b, c, src, dst, pair_ptr, node_begin, node_end, int(max_nodes), relu, out)


def pair_combine_bwd(g_out, out, relu, src, dst, pair_ptr, node_begin, node_end, g_ab, g_c):
    _count()
    ops().pair_combine_bwd(g_out, out, relu, src, dst, pair_ptr, node_begin, node_end, g_ab, g_c)


def link_ce_fwd(hid, W, b, golden, logits, dlogits, loss, n_correct, live=None):
    """live: device int32 [1]: only the first live[0] rows are pairs (the rest pads a fixed-size buffer)."""
    _count()
    ops().link_ce_fwd(hid, W, b, golden, live, logits, dlogits, loss, n_correct)


def link_ce_bwd_input(dlogits, W, g_loss, g_hid):
    _count()
    ops().link_ce_bwd_input(dlogits, W, g_loss, g_hid)


def token_ce_fwd(logits, gold, ignore_index, n_classes, dlogits, loss, counts):
    """logits [R, >= n_classes] (rows of any stride), gold int32 [R] -> dlogits (same width as its
    rows are wide: columns from n_classes on are zeroed), loss f32[2], counts i32[2]."""
    _count()
    ops().token_ce_fwd(logits, gold, int(ignore_index), int(n_classes), dlogits, loss, counts)


def relu_classes_tiles(N):
    return int(lib().psvae_gemm_f32_relu_classes_tiles(int(N)))


def gemm_relu_classes(x, weight, bias, cls_w4, partials):
    """psvae_gemm_f32_relu_classes: partials[m][t][c] = sum over column tile t of relu(x W^T + b) * cls_w4[c]."""
    M, K = x.shape
    N = weight.shape[0]
    _count()
    px, ldx = _rows(x)
    pw, ldw = _rows(weight)
    with _Prof('gemm_f32_kernel', 2.0 * M * N * (K + 4), (M, N, K, 'k', 'k', False, False)):
        _check(lib().psvae_gemm_f32_relu_classes(M, N, K, px, ldx, pw, ldw, ptr(bias, torch.float32) if bias is not None else None,
                                                 ptr(cls_w4, torch.float32), ptr(partials, torch.float32), _stream()),
               'psvae_gemm_f32_relu_classes')


def pair_enumerate(node_off, frag, max_atoms):
    """Candidate pairs of generated molecules (psvae_pair_enumerate_count / _fill): returns
    (src, dst, graph, m, k int32 [P], ptr int32 [B + 1]).  One host read (the total) sizes the outputs."""
    B = node_off.numel() - 1
    dev = node_off.device
    counts = torch.empty(max(B, 1), dtype=torch.int32, device=dev)
    ptr_t = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    _count()
    _check(lib().psvae_pair_enumerate_count(B, int(max_atoms), ptr(node_off, torch.int32), ptr(frag, torch.int32),
                                            ptr(counts, torch.int32), _stream()), 'psvae_pair_enumerate_count')
    if B > 0:
        exclusive_scan(counts[:B], ptr_t)
    P = int(ptr_t[-1].item()) if B > 0 else 0
    outs = [torch.empty(P, dtype=torch.int32, device=dev) for _ in range(5)]
    if P > 0:
        _check(lib().psvae_pair_enumerate_fill(B, int(max_atoms), ptr(node_off, torch.int32), ptr(frag, torch.int32),
                                               ptr(ptr_t, torch.int32), *[ptr(o, torch.int32) for o in outs], _stream()),
               'psvae_pair_enumerate_fill')
    return (*outs, ptr_t)


def piece_draw(logits, temperature, uniforms_col, pad, end, first_step, piece_atoms, max_atom_num, max_pos, active, atoms,
               length, out_col, cur):
    """psvae_piece_draw: one sampling step for all molecules (uniforms_col / out_col: column views [B] of the
    [B, max_pos] arrays)."""
    B, V = logits.shape
    _count()
    _check(lib().psvae_piece_draw(B, V, ptr(logits, torch.float32), logits.stride(0), float(temperature),
                                  ptr(uniforms_col, torch.float32), uniforms_col.stride(0) if B > 1 else 1, int(pad), int(end),
                                  int(bool(first_step)), ptr(piece_atoms, torch.int32), int(max_atom_num), int(max_pos),
                                  ptr(active, torch.uint8), ptr(atoms, torch.int32), ptr(length, torch.int32),
                                  ptr(out_col, torch.int64), out_col.stride(0) if B > 1 else 1, ptr(cur, torch.int64),
                                  _stream()), 'psvae_piece_draw')


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


def adam_step_scalars(step, lr, betas=(0.9, 0.999)):
    """(lr / (1 - beta1^t), sqrt(1 - beta2^t)) in double: the two step-dependent scalars of the update."""
    return lr / (1.0 - betas[0] ** step), (1.0 - betas[1] ** step) ** 0.5


def clip_adam_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0,
                   max_norm=0.0, write_grad=True, norm_out=None, step_scalars=None):
    """Scale + global-norm clip + Adam on flat f32 buffers (psvae_clip_adam_step).  step_scalars: a
    device float32 [2] holding adam_step_scalars(step, lr) - then `step` and `lr` are not read and
    nothing step-dependent is a kernel argument (psvae_clip_adam_step_dev: for HIP graph replay)."""
    L = lib()
    n = p.numel()
    if not (g.numel() == m.numel() == v.numel() == n):
        raise ValueError('clip_adam_step: buffers differ in length')
    nbytes = L.psvae_clip_adam_workspace_bytes(n)
    ws = workspace(nbytes, p.device) if nbytes else None
    f = torch.float32
    if step_scalars is not None:
        _check(L.psvae_clip_adam_step_dev(n, ptr(p, f), ptr(g, f), ptr(m, f), ptr(v, f), float(grad_scale),
                                          float(max_norm or 0.0), float(betas[0]), float(betas[1]), float(eps),
                                          float(weight_decay), ptr(step_scalars, f), 1 if write_grad else 0,
                                          ptr(norm_out, f), ptr(ws), nbytes, _stream()), 'psvae_clip_adam_step_dev')
        return
    _check(L.psvae_clip_adam_step(n, ptr(p, f), ptr(g, f), ptr(m, f), ptr(v, f), float(grad_scale), float(max_norm or 0.0),
                                  float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                  int(step), 1 if write_grad else 0, ptr(norm_out, f), ptr(ws), nbytes, _stream()),
           'psvae_clip_adam_step')
