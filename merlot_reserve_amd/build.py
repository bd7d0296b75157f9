"""Builds libmreserve_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the library is a plain
C-ABI shared object (include/mreserve_hip.h) loaded through ctypes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmreserve_hip.so')
# attention.hip: the SLP vectoriser turns adjacent scalar fp32 adds / multiplies of the softmax into v_pk_add_f32 / v_pk_mul_f32, which
# issue slower than the scalar pairs they replace on gfx950 (MI355X_MICROARCH.md): 961 -> 192 packed ops, backward kernels 2-4 % faster
# layernorm.hip / rowops.hip / adam.hip: -fno-slp-vectorize for CORRECTNESS.  With SLP the LayerNorm kernels' (x - mean) * (rstd * gamma) + beta
# became v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with op_sel operands, and in ~1e-4 of the rows element 0 of the 16-byte vectors of
# lanes 48-63 came out as if computed from other mean / rstd -- only while a second queue's kernels shared the chip (the audio tower on
# the side stream), never with one stream; run-to-run differences of a bf16 ulp or seven that made eager steps, hipGraph replays
# and repeated replays of the SAME step disagree (scripts/det_check.py).  Scalar fp32 code: bit-identical
# results in every mode.  These kernels are HBM-bound: no cost.  adam.hip also -ffp-contract=off: the optimizer state is byte data
# (pretrain/optimization.py:36-51), the oracle evaluates (1 - b) * g + b * m without fused multiply-adds.
# Round 4 reduced the trigger (scripts/slp_repro.py, tests/test_coresident_determinism_gpu.py): packed fp32 code with op_sel / neg operands goes wrong
# in lanes 48-63 while waves of an MFMA kernel (the small 128 x 128 GEMM) share the SIMD.  f32path.hip / f32bwd.hip (not performance paths) are built
# scalar as well; the bf16 GEMM files keep their packed epilogues (3.2 ms of the step) and are held to bit-stability under that neighbour by the test.
EXTRA_FLAGS = {'attention.hip': ['-fno-slp-vectorize'], 'layernorm.hip': ['-fno-slp-vectorize'], 'rowops.hip': ['-fno-slp-vectorize'],
               'adam.hip': ['-fno-slp-vectorize', '-ffp-contract=off'], 'f32bwd.hip': ['-fno-slp-vectorize'], 'f32path.hip': ['-fno-slp-vectorize']}
# The hazard at instruction level (scripts/micro/pk_probe.hip, scripts/pk_probe.py): v_pk_add_f32 / v_pk_fma_f32 (any packed fp32 arithmetic) with an
# `op_sel:[..1..]` source -- the LOW result element reading the HIGH register of a source pair -- returns wrong values in lanes 48-63 of a wave
# while another kernel's MFMA waves are resident on its SIMD (0 mismatches alone; op_sel_hi, neg_lo / neg_hi, v_pk_mov_b32 op_sel are unaffected).
# The files that keep the SLP pass are therefore SCANNED: build() also emits their device assembly and fails if such an instruction appears.
SLP_FILES = ['gemm.hip', 'gemm256.hip', 'gemm3.hip', 'gemm4.hip', 'gemm5.hip']
HAZARD_RE = r'^\s*v_pk_(add|mul|fma|min|max)_f32\b.*\bop_sel:\['
SOURCES = ['gemm.hip', 'gemm256.hip', 'gemm3.hip', 'gemm4.hip', 'gemm5.hip', 'attention.hip', 'layernorm.hip', 'rowops.hip', 'adam.hip', 'f32path.hip', 'f32bwd.hip', 'mr_error.cpp', 'comm.cpp']


def _needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'mreserve_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def scan_packed_op_sel():
    """[(file, count, example)] of hazardous packed-fp32 instructions in the device assembly build() left under build/ (see HAZARD_RE)."""
    import re
    out = []
    rx = re.compile(HAZARD_RE, re.M)
    for src in SLP_FILES:
        asm = os.path.join(HERE, 'build', src.rsplit('.', 1)[0] + '.s')
        if not os.path.exists(asm):
            raise FileNotFoundError(f'{asm}: run merlot_reserve_amd.build.build(force=True)')
        hits = [m.group(0).strip() for m in rx.finditer(open(asm).read())]
        if hits:
            out.append((src, len(hits), hits[0]))
    return out


def build(force=False, verbose=True):
    if not force and not _needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # MR_DEBUG_ENV=1: a DEBUG build whose kernels' dispatch reads the experiment scripts' environment knobs (MR_GEMM3, MR_G3_PH, ...:
    # csrc/mr_options.h mr_env_int); the product build ignores the environment
    debug = ['-DMR_DEBUG_ENV'] if os.environ.get('MR_DEBUG_ENV') == '1' else []
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, 'build', src.rsplit('.', 1)[0] + '.o')
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17'] + debug + EXTRA_FLAGS.get(src, []) + ['-x', 'hip', '-c', os.path.join(CSRC, src), '-o', obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    asm_procs = []
    for src in SLP_FILES:             # device assembly of the packed-code files, for the hazard scan (same flags as the object)
        asm = os.path.join(HERE, 'build', src.rsplit('.', 1)[0] + '.s')
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17'] + debug + EXTRA_FLAGS.get(src, []) + ['-x', 'hip', '--cuda-device-only', '-S', os.path.join(CSRC, src), '-o', asm]
        asm_procs.append((src, asm, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out.decode()}')
    for src, asm, p in asm_procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc -S failed on {src}:\n{out.decode()}')
    bad = scan_packed_op_sel()
    if bad:
        raise RuntimeError('packed fp32 instructions with an op_sel source (wrong in lanes 48-63 beside MFMA waves, see build.py) in:\n' +
                           '\n'.join(f'{f}: {n} e.g. {ex}' for f, n, ex in bad) + '\ncompile that file with -fno-slp-vectorize or restructure the code')
    # -z defs: an undefined kernel stub (a template the host pass silently failed to emit) fails the build instead of the first launch
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,-z,defs', '-o', LIB] + objs + ['-ldl']
    subprocess.check_call(cmd)
    if verbose:
        print(f'built {LIB}', file=sys.stderr)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
