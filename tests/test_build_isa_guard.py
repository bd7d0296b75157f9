"""The build's scan for the packed-fp32 hazard (merlot_reserve_amd/build.py: v_pk_*_f32 with an `op_sel` source goes wrong in lanes 48-63 while another
kernel's MFMA waves share the SIMD -- scripts/pk_probe.py): the pattern flags the instruction forms measured wrong and not the forms
measured clean, and the device assembly of the files that keep the SLP vectoriser contains none."""
import os
import re

from merlot_reserve_amd import build as B


def test_pattern_flags_the_measured_forms():
    rx = re.compile(B.HAZARD_RE, re.M)
    wrong = ['\tv_pk_add_f32 v[32:33], v[40:41], v[50:51] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]',        # the packed LayerNorm's x - mean
             '\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]',
             '\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[2:3] op_sel:[0,1,0]']
    clean = ['\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] neg_lo:[0,1] neg_hi:[0,1]',
             '\tv_pk_mul_f32 v[40:41], v[48:49], v[40:41] op_sel_hi:[0,1]',
             '\tv_pk_add_f32 v[6:7], v[6:7], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]',
             '\tv_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]',
             '\tv_pk_add_f32 v[0:1], v[2:3], v[4:5]']
    assert all(rx.search(l) for l in wrong)
    assert not any(rx.search(l) for l in clean)


def _fresh_build():
    if B._asm_stale() or B._needs_build():
        B.build(force=True, verbose=False)          # (cross-compiles without a GPU: ~1 min)


def test_no_hip_file_carries_a_hazardous_instruction():
    """EVERY .hip file's device assembly (the by-product of the shipped object's own compile), not only the files that keep the SLP pass: the files
    built with -fno-slp-vectorize still contain packed fp32 arithmetic from explicit vector expressions."""
    _fresh_build()
    assert set(B.HIP_FILES) == {s for s in B.SOURCES if s.endswith('.hip')}
    assert all(os.path.exists(B._asm_path(f)) for f in B.HIP_FILES)
    assert B.scan_packed_op_sel() == []
    import re
    n_pk = {f: len(re.findall(r'^\s*v_pk_(?:add|mul|fma)_f32', open(B._asm_path(f)).read(), re.M)) for f in B.HIP_FILES}
    assert n_pk['attention.hip'] > 0, 'the scan must see the packed code the scalar-built files do contain'
    # the scalar files are built without the SLP pass at all
    for f in ('layernorm.hip', 'rowops.hip', 'adam.hip', 'attention.hip', 'f32path.hip', 'f32bwd.hip'):
        assert '-fno-slp-vectorize' in B.EXTRA_FLAGS[f], f
    assert set(B.SLP_FILES) | set(B.EXTRA_FLAGS) >= set(B.HIP_FILES)


def test_register_spills_are_listed_and_cold():
    """No kernel spills vector registers unless build.SPILL_ALLOWED lists it (with its count and reason); scratch traffic inside a basic block
    that also holds MFMAs (a k-loop body) only in the three listed HOT kernels -- none of them on the default dispatch path of the bench workloads
    except the masked dQ kernel, whose trade is measured (build.py)."""
    _fresh_build()
    assert B.scan_spills() == []
    res = B.kernel_resources()
    spilled = {k: e['vgpr_spill'] for ks in res.values() for k, e in ks.items() if e.get('vgpr_spill', 0) > 0}
    import re
    for k, n in spilled.items():
        caps = [c for rx, (c, _why) in B.SPILL_ALLOWED.items() if re.search(rx, k)]
        assert caps and n <= max(caps), (k, n)
    # round 4's offenders are gone
    names = [k for ks in res.values() for k in ks]
    assert not any('gemm4_kernelILi256ELi0E' in k for k in names), 'gemm4<256,0> (183 spilled, reloads in the k-loop) must not be built'
    assert not any('gemm3_kernelILi256ELi4ELi1E' in k for k in names)
    assert all(e.get('vgpr_spill', 0) == 0 for k, e in res['gemm.hip'].items()), 'the small-GEMM kernels carry no spills'
    hot = {k for _f, k, *_ in B.hot_spill_blocks()}
    allowed_hot = [rx for rx, (_c, why) in B.SPILL_ALLOWED.items() if why.startswith('HOT')]
    for k in hot:
        assert any(re.search(rx, k) for rx in allowed_hot), f'scratch traffic in an MFMA block of {k}'


def test_flag_stamp_triggers_rebuild(monkeypatch):
    """Toggling MR_DEBUG_ENV changes the stamp build() compares: a debug library cannot be left in place as the product (or the reverse)."""
    _fresh_build()
    assert not B._needs_build()
    monkeypatch.setenv('MR_DEBUG_ENV', '1')
    assert B._needs_build()
