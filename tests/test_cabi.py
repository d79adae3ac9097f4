"""The C-ABI library: it loads, exports every symbol include/psvae_hip.h declares, and the
ctypes signatures in ps_vae_amd.hip agree with the header's parameter lists.  No compute
call is made (CPU only)."""
import ctypes
import os
import re

import pytest

import ps_vae_amd  # noqa: F401
from ps_vae_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'psvae_hip.h')

_KIND = [
    (re.compile(r'\*'), ctypes.c_void_p),
    (re.compile(r'\bint64_t\b'), ctypes.c_int64),
    (re.compile(r'\blong long\b'), ctypes.c_longlong),
    (re.compile(r'\bsize_t\b'), ctypes.c_size_t),
    (re.compile(r'\bdouble\b'), ctypes.c_double),
    (re.compile(r'\bfloat\b'), ctypes.c_float),
    (re.compile(r'\bint\b'), ctypes.c_int),
]


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(int|size_t|const char \*)\s*(psvae_\w+)\s*\(([^;]*?)\)\s*;', text, flags=re.S):
        params = m.group(3).strip()
        plist = [] if params in ('', 'void') else [p.strip() for p in params.split(',')]
        kinds = []
        for p in plist:
            for rx, ct in _KIND:
                if rx.search(p):
                    kinds.append(ct)
                    break
            else:
                raise AssertionError(f'cannot classify parameter {p!r}')
        out[m.group(2)] = kinds
    return out


def test_header_and_binding_agree():
    decl = declared_functions()
    assert set(decl) == set(hip.exported_symbols())
    for name, kinds in decl.items():
        _, argtypes = hip._SIGNATURES[name]
        assert len(argtypes) == len(kinds), name
        for i, (a, k) in enumerate(zip(argtypes, kinds)):
            assert a is k, f'{name} parameter {i}: binding {a} vs header {k}'


def test_library_loads_and_exports_every_symbol():
    if not os.path.exists(hip.LIB_PATH):
        pytest.fail(f'{hip.LIB_PATH} not built (run __graft_entry__.build())')
    handle = ctypes.CDLL(hip.LIB_PATH)
    for name in declared_functions():
        assert hasattr(handle, name), name
    assert hip.lib().psvae_version() >= 1


def test_ops_refuse_cpu_tensors():
    import torch
    with pytest.raises(RuntimeError):
        hip.ptr(torch.zeros(4))


def test_rejected_arguments_return_codes_without_touching_the_gpu():
    """Argument checks run before any launch: bad sizes, null pointers, short workspaces and
    unsupported shapes come back as negative status codes with a message, and empty problems
    return 0.  (Pointers here are never dereferenced: every call returns from its checks.)"""
    if not os.path.exists(hip.LIB_PATH):
        pytest.fail(f'{hip.LIB_PATH} not built (run __graft_entry__.build())')
    L = hip.lib()
    EINVAL, ETOOLARGE, EWORKSPACE = -1, -2, -3
    P = ctypes.c_void_p(4096)          # a non-null, 16-byte aligned stand-in
    null = None

    def gemm(M, N, K, A=P, B=P, C=P, ldc=None, split_k=1, ws=null, ws_bytes=0, act=0):
        return L.psvae_gemm_f32(M, N, K, A, null, K, 1, B, 1, K, C, N if ldc is None else ldc, null, null, null, act,
                                0, split_k, ws, ws_bytes, null)

    assert gemm(0, 8, 8) == 0 and gemm(8, 0, 8) == 0                  # empty output: nothing to do
    assert gemm(-1, 8, 8) == EINVAL
    assert gemm(8, 8, 8, C=null) == EINVAL
    assert gemm(8, 8, 8, A=null) == EINVAL
    assert gemm(8, 8, 8, ldc=4) == EINVAL
    assert gemm(8, 8, 8, act=7) == EINVAL
    assert gemm(64, 64, 4096, split_k=4) == EWORKSPACE               # split-K without its workspace
    assert b'workspace' in L.psvae_last_error()
    assert L.psvae_gemm_f32_workspace_bytes(64, 64, 4096, 4) >= 4 * 64 * 64 * 4
    assert L.psvae_gemm_f32_workspace_bytes(64, 64, 4096, 1) == 0

    # first encoder layer from ids: widths must be multiples of 4, tables non-empty
    et = L.psvae_embed_trans_fwd
    assert et(0, 300, null, null, null, 11, 302, 50, P, P, 300, null) == 0
    assert et(5, 302, P, P, P, 11, 302, 50, P, P, 304, null) == EINVAL
    assert et(5, 300, P, P, P, 11, 302, 50, P, P, 296, null) == EINVAL
    assert et(5, 300, P, P, P, 0, 302, 50, P, P, 300, null) == EINVAL
    assert et(5, 300, null, P, P, 11, 302, 50, P, P, 300, null) == EINVAL
    assert et(5, 300, P, P, P, 11, 302, 50, ctypes.c_void_p(4100), P, 300, null) == EINVAL   # misaligned table

    # greedy assembly: sizes, null arrays, workspace
    need = L.psvae_assemble_links_workspace_bytes(100)
    assert need == 100 * (8 + 6) * 4 and L.psvae_assemble_links_workspace_bytes(0) == 0

    def assemble(B, V, ws=P, ws_bytes=need, node_off=P, n_atom_type=10):
        return L.psvae_assemble_links(B, V, node_off, P, P, P, P, P, P, P, P, P, P, P, n_atom_type, 9, P, P, P, P, ws,
                                      ws_bytes, null)

    assert assemble(0, 0) == 0
    assert assemble(-1, 0) == EINVAL
    assert assemble(4, 100, n_atom_type=0) == EINVAL
    assert assemble(4, 100, node_off=null) == EINVAL
    assert assemble(4, 100, ws_bytes=need - 1) == EWORKSPACE
    assert assemble(4, 100, ws=null) == EWORKSPACE

    # keyed row sums: table sizes, strides, workspace
    ks = L.psvae_key_sort
    assert ks(5, 0, P, null, null, 4, 0, 0, 256, P, P, P, null, 0, null) == EINVAL          # no table
    assert ks(5, 2, P, null, null, 4, 4, 0, 256, P, P, P, null, 0, null) == EINVAL          # second id array missing
    assert ks(5, 1, P, null, null, 9000, 0, 0, 256, P, P, P, null, 0, null) == ETOOLARGE     # more rows than fit the counters
    assert ks(5000, 1, P, null, null, 40, 0, 0, 256, P, P, P, null, 0, null) == EWORKSPACE   # several workgroups need histograms
    assert L.psvae_key_sort_workspace_bytes(5000, 40) == 5 * 40 * 4 and L.psvae_key_sort_workspace_bytes(1000, 40) == 0
    need = L.psvae_keyed_row_sum_workspace_bytes(1000, 40, 8, 256)
    assert need == (1000 // 256 + 40) * 8 * 4
    krs = L.psvae_keyed_row_sum
    assert krs(40, 8, 1000, 256, P, 8, P, P, P, P, 8, P, need - 1, null) == EWORKSPACE
    assert krs(40, 6, 1000, 256, P, 8, P, P, P, P, 8, P, need, null) == EINVAL               # width not a multiple of 4
    assert krs(40, 8, 1000, 256, P, 4, P, P, P, P, 8, P, need, null) == EINVAL               # row stride below the width

    # link loss / inference helpers
    assert L.psvae_link_infer(0, 4, null, 0.5, 0, null, null, null, null) == 0

    # generation: sampling step, candidate pairs, hidden + class layer in one product
    draw = L.psvae_piece_draw
    assert draw(0, 302, null, 302, 1.0, null, 1, 300, 301, 1, null, 38, 50, null, null, null, null, 1, null, null) == 0
    assert draw(4, 302, P, 300, 1.0, P, 1, 300, 301, 1, P, 38, 50, P, P, P, P, 1, P, null) == EINVAL      # rows shorter than V
    assert draw(4, 302, P, 302, 0.0, P, 1, 300, 301, 1, P, 38, 50, P, P, P, P, 1, P, null) == EINVAL      # temperature
    assert draw(4, 302, P, 302, 1.0, P, 1, 302, 301, 1, P, 38, 50, P, P, P, P, 1, P, null) == EINVAL      # pad outside the vocabulary
    assert draw(4, 302, P, 302, 1.0, null, 1, 300, 301, 1, P, 38, 50, P, P, P, P, 1, P, null) == EINVAL
    assert L.psvae_pair_enumerate_count(0, 0, null, null, null, null) == 0
    assert L.psvae_pair_enumerate_count(3, 40, null, P, P, null) == EINVAL
    assert L.psvae_pair_enumerate_count(3, 20000, P, P, P, null) == ETOOLARGE
    assert L.psvae_pair_enumerate_fill(3, 40, P, P, P, P, P, P, P, null, null) == EINVAL
    tiles = L.psvae_gemm_f32_relu_classes_tiles
    assert tiles(328) == 2 and tiles(160) == 1 and tiles(300) == 2 and tiles(100) == 0 and tiles(600) == 0 and tiles(330) == 0
    rc = L.psvae_gemm_f32_relu_classes
    assert rc(0, 328, 328, null, 328, null, 328, null, null, null, null) == 0
    assert rc(8, 100, 328, P, 328, P, 328, P, P, P, null) == EINVAL          # width not covered
    assert rc(8, 328, 326, P, 328, P, 328, P, P, P, null) == EINVAL          # K not a multiple of 4
    assert rc(8, 328, 328, P, 328, P, 328, P, null, P, null) == EINVAL       # no class rows
    assert L.psvae_gine_aggregate_bwd_slabs(0, 300, 4) == 0 and 1 <= L.psvae_gine_aggregate_bwd_slabs(95000, 300, 4) <= 1024
    assert L.psvae_gine_table_reduce(0, 4, 300, null, null, 0, null) == EINVAL                   # no output
    assert L.psvae_gine_table_reduce(3, 4, 300, null, P, 0, null) == EINVAL                      # slabs missing
    assert ETOOLARGE == -2


def test_torch_operator_library_registers_its_operators():
    """csrc/libpsvae_torch.so loads and registers the operators of the training step under
    torch.ops.psvae; on a CPU tensor they raise instead of computing anything."""
    import torch
    from ps_vae_amd import hip
    ops = hip.ops()
    names = ['gemm', 'gine_aggregate_fwd', 'gine_aggregate_bwd', 'pair_combine_fwd', 'pair_combine_bwd',
             'link_ce_fwd', 'link_ce_bwd_input', 'segment_sum_fwd', 'segment_sum_bwd', 'rsample_kl_fwd',
             'rsample_kl_bwd', 'token_ce_fwd', 'embed_trans_fwd', 'embed_atom_onehot', 'embedding_fwd', 'onehot',
             'key_sort', 'keyed_row_sum',
             'affine_merge_fwd', 'affine_merge_bwd', 'colsum', 'gru_seq_fwd', 'gru_seq_bwd', 'block_table', 'mse_selected_fwd',
             'mse_selected_bwd', 'loss_mix_fwd', 'loss_mix_bwd', 'pair_combine_fwd_graphs']
    for name in names:
        assert getattr(ops, name)._schema.name == f'psvae::{name}'
    x = torch.zeros(4, 4)
    with pytest.raises(RuntimeError, match='GPU tensors'):
        ops.gemm(4, 4, 4, x, None, 4, 1, x, 1, 4, torch.zeros(4, 4), 4, None, None, None, False, False, 1)
