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


def test_dynamic_symbol_table_is_the_header_and_nothing_else():
    """`nm -D`: the defined dynamic symbols are exactly the header's C names -- no mangled C++ helper, kernel handle or device stub leaks
    (csrc/exports.map, the link's version script)."""
    import subprocess
    import __graft_entry__
    __graft_entry__.build()
    from merlot_reserve_amd import _lib
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == declared_symbols(), sorted(set(exported) ^ set(declared_symbols()))


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


def test_options_are_named_and_unknown_names_rejected(monkeypatch):
    """mr_set_option: every knob the header documents is accepted, an unknown name is MR_EINVAL; ops.gemm_cus (the 240-workgroup plan of
    the data-parallel backward) sets and restores the knob only for world > 1."""
    from merlot_reserve_amd import _lib, ops
    lib = _lib.load()
    text = open(os.path.join(ROOT, 'include', 'mreserve_hip.h')).read()
    names = re.findall(r'^ \*   "([a-z0-9_]+)"', text, flags=re.M)
    assert {'gemm3', 'gemm3_phases', 'gemm4', 'gemm_cus', 'gemm_tile_n'} <= set(names)
    defaults = {'gemm3': 1, 'gemm4': -1, 'gemm5': -1}
    for n in names:
        assert lib.mr_set_option(n.encode(), defaults.get(n, 0)) == 0, n
    assert lib.mr_set_option(b'no_such_knob', 1) == -1 and b'no_such_knob' in lib.mr_last_error()
    calls = []
    monkeypatch.setattr(ops, 'set_option', lambda name, value: calls.append((name, value)))
    with ops.gemm_cus(1):
        pass
    assert calls == []
    with ops.gemm_cus(8):
        assert calls == [('gemm_cus', 240)]
    assert calls == [('gemm_cus', 240), ('gemm_cus', 0)]        # (0 = the value the option had on entry)
    monkeypatch.setenv('MR_COMM_GEMM_CUS', '0')
    with ops.gemm_cus(8):
        pass
    assert len(calls) == 2


def test_handles_carry_their_own_options():
    """mr_create / mr_destroy / mr_make_current: a handle's option set is its own; the calling thread launches under its
    CURRENT handle's options; mr_set_option / mr_get_option address the current handle, or the process defaults without one; another
    thread is not affected; ops.gemm_cus restores the value it found."""
    import threading
    from merlot_reserve_amd import _lib, ops
    lib = _lib.load()
    assert lib.mr_version() >= 4
    assert lib.mr_get_current() is None
    ops.set_option('gemm_cus', 0)
    h1, h2 = ops.Handle(0), ops.Handle(0)                # (no workspace: nothing touches a GPU here)
    h1.set_option('gemm_cus', 240)
    assert h1.get_option('gemm_cus') == 240 and h2.get_option('gemm_cus') == 0 and ops.get_option('gemm_cus') == 0
    with h1:
        assert ops.get_option('gemm_cus') == 240        # the shim reads the current handle
        ops.set_option('gemm3', 192)                    # ... and writes it
        seen = []
        t = threading.Thread(target=lambda: seen.append((lib.mr_get_current(), ops.get_option('gemm_cus'), ops.get_option('gemm3'))))
        t.start(); t.join()
        assert seen == [(None, 0, 1)]                   # another thread: no current handle, the process defaults
        with h2:
            assert ops.get_option('gemm_cus') == 0
            with ops.gemm_cus(8):
                assert ops.get_option('gemm_cus') == 240 and h2.get_option('gemm_cus') == 240
            assert h2.get_option('gemm_cus') == 0
        assert ops.get_option('gemm_cus') == 240        # h1 is current again
        with ops.gemm_cus(8):
            pass
        assert h1.get_option('gemm_cus') == 240         # restored to what it was, not to 0 (advisor finding, round 3)
    assert lib.mr_get_current() is None and ops.get_option('gemm3') == 1 and h1.get_option('gemm3') == 192
    bad = ctypes.c_int32(0)
    assert lib.mr_handle_get_option(None, b'no_such_knob', ctypes.byref(bad)) == -1
    assert lib.mr_create(-1, 0, ctypes.byref(ctypes.c_void_p())) == -1
    h1.close(); h2.close()
    assert lib.mr_last_gemm_kernel() == b''             # nothing launched, tracing off


def test_destroy_is_refused_while_another_thread_holds_the_handle():
    """A handle current on another thread must not be freed under that thread's next launch (advisor finding, round 4): mr_destroy answers MR_EINVAL
    until the other thread has let go."""
    import threading
    from merlot_reserve_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.mr_create(0, 0, ctypes.byref(h)) == 0
    holding, release = threading.Event(), threading.Event()

    def other():
        lib.mr_make_current(h)
        holding.set()
        release.wait(10)
        lib.mr_make_current(None)
    t = threading.Thread(target=other)
    t.start()
    assert holding.wait(10)
    assert lib.mr_destroy(h) == -1 and b'current on 1 other thread' in lib.mr_last_error()
    lib.mr_make_current(h)                               # current here too: still one OTHER holder
    assert lib.mr_destroy(h) == -1
    release.set(); t.join()
    assert lib.mr_destroy(h) == 0                        # only this thread holds it: destroyed, and no longer current
    assert lib.mr_get_current() is None


def test_grouped_gemm_validates_every_problem_before_launching():
    """mr_gemm_grouped runs mr_gemm's operand checks per problem BEFORE handing the list to any kernel (advisor finding, round 3)."""
    from merlot_reserve_amd import _lib
    lib = _lib.load()
    arr = (_lib.GemmArgs * 2)()
    for g in arr:
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.transA = 512, 512, 4096, 512, 512, 512, 1
        g.A = g.B = g.C = 1 << 20                        # aligned fake addresses: never dereferenced, the second problem is rejected first
    arr[1].B = (1 << 20) + 2                             # misaligned
    assert lib.mr_gemm_grouped(arr, 2, None) == -1 and b'problem 1' in lib.mr_last_error() and b'aligned' in lib.mr_last_error()
    arr[1].B = 1 << 20
    arr[1].K = 0
    assert lib.mr_gemm_grouped(arr, 2, None) == -1 and b'problem 1 is empty' in lib.mr_last_error()
