"""The C-ABI library loads and exports every symbol include/mreserve_hip.h declares (no GPU, no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'mreserve_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mr_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from merlot_reserve_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f'{n} declared in the header but not exported'
    assert set(names) == set(_lib.PROTOTYPES), (set(names) ^ set(_lib.PROTOTYPES))
    assert _lib.load().mr_version() >= 1


def test_errors_are_reported_not_thrown():
    from merlot_reserve_amd import _lib
    lib = _lib.load()
    g = _lib.GemmArgs()          # all zero: must be rejected with MR_EINVAL and a message, no launch
    rc = lib.mr_gemm(ctypes.byref(g), None)
    assert rc == -1 and b'mr_gemm' in lib.mr_last_error()
    assert lib.mr_layernorm_fwd(None, 0, None, None, None, 0, None, None, 1, 8, 1e-5, None) == -1


def test_product_does_not_import_oracle():
    for root, _d, files in [w for pkg in ('merlot_reserve_amd', 'mreserve', 'pretrain', 'finetune') for w in os.walk(os.path.join(ROOT, pkg))]:
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert 'oracle' not in re.sub(r'#.*', '', src).replace('"""', ''), f'{f} mentions the oracle'
