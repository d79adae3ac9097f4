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
