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


def test_slp_files_carry_no_hazardous_instruction():
    asm = [os.path.join(B.HERE, 'build', f.rsplit('.', 1)[0] + '.s') for f in B.SLP_FILES]
    stale = any(not os.path.exists(a) or os.path.getmtime(a) < os.path.getmtime(os.path.join(B.CSRC, f)) for a, f in zip(asm, B.SLP_FILES))
    if stale:
        B.build(force=True, verbose=False)          # (cross-compiles without a GPU: ~1 min)
    assert B.scan_packed_op_sel() == []
    # the scalar files are built without the SLP pass at all
    for f in ('layernorm.hip', 'rowops.hip', 'adam.hip', 'attention.hip', 'f32path.hip', 'f32bwd.hip'):
        assert '-fno-slp-vectorize' in B.EXTRA_FLAGS[f], f
    assert set(B.SLP_FILES) | set(B.EXTRA_FLAGS) >= {s for s in B.SOURCES if s.endswith('.hip')}, 'every .hip file is either scanned or built scalar'
