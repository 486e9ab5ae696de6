#!/usr/bin/env python3
"""Generates build/ub_issue2.hip (not kept in the repository) and, with --build, compiles it to
tools/ub_issue2: issue-cost microbenchmarks for mixes of gfx950 VALU instructions.  Each variant is a
list of 8 instruction templates; {d} = k-th destination VGPR, {dp} = k-th destination VGPR pair,
{a},{b} VGPRs, {ap},{bp} VGPR pairs, {s} SGPR, {sp} SGPR pair.
usage: gen_ub_issue2.py [--build] [--only SUBSTRING[,SUBSTRING...]]"""
import os
import subprocess
import sys
V = {
 "mul vgpr": ["v_mul_f32_e32 {d}, {a}, {b}"] * 8,
 "add vgpr": ["v_add_f32_e32 {d}, {a}, {b}"] * 8,
 "sub vgpr": ["v_sub_f32_e32 {d}, {a}, {b}"] * 8,
 "max vgpr": ["v_max_f32_e32 {d}, {a}, {b}"] * 8,
 "min vgpr": ["v_min_f32_e32 {d}, {a}, {b}"] * 8,
 "max3 vgpr": ["v_max3_f32 {d}, {a}, {b}, {b}"] * 8,
 "med3 vgpr": ["v_med3_f32 {d}, {a}, {b}, {b}"] * 8,
 "med3 vgpr, own dest as src2": ["v_med3_f32 {d}, {a}, {b}, {d}"] * 8,
 "alt med3/pk_fma/med3/med3": ["v_med3_f32 {d}, {a}, {b}, {b}", "v_pk_fma_f32 {dp}, {ap}, {bp}, {bp}", "v_med3_f32 {d}, {a}, {b}, {b}", "v_med3_f32 {d}, {a}, {b}, {b}"] * 2,
 "alt max/min": ["v_max_f32_e32 {d}, {a}, {b}", "v_min_f32_e32 {d}, {a}, {b}"] * 4,
 "ashr vgpr": ["v_ashrrev_i32_e32 {d}, 31, {a}"] * 8,
 "bitop3 vgpr": ["v_bitop3_b32 {d}, {a}, {b}, {b} bitop3:0xfd"] * 8,
 "alt med3/sub": ["v_med3_f32 {d}, {a}, {b}, {b}", "v_sub_f32_e32 {d}, {a}, {b}"] * 4,
 "alt med3/ashr": ["v_med3_f32 {d}, {a}, {b}, {b}", "v_ashrrev_i32_e32 {d}, 31, {a}"] * 4,
 "alt med3(3 distinct)/sub": ["v_med3_f32 {d}, {a}, {b}, {d}", "v_sub_f32_e32 {d}, {a}, {b}"] * 4,
 "alt max/mul": ["v_max_f32_e32 {d}, {a}, {b}", "v_mul_f32_e32 {d}, {a}, {b}"] * 4,
 "alt pk_fma/ashr": ["v_pk_fma_f32 {dp}, {ap}, {bp}, {bp}", "v_ashrrev_i32_e32 {d}, 31, {a}"] * 4,
 "alt med3/med3/sub/and": ["v_med3_f32 {d}, {a}, {b}, {b}", "v_med3_f32 {d}, {a}, {b}, {b}", "v_sub_f32_e32 {d}, {a}, {b}", "v_and_b32_e32 {d}, {a}, {b}"] * 2,
 "cndmask vcc": ["v_cndmask_b32_e32 {d}, {a}, {b}, vcc"] * 8,
 "cndmask sgpr pair": ["v_cndmask_b32_e64 {d}, {a}, {b}, s[20:21]"] * 8,
 "cmp then cndmask vcc": ["v_cmp_gt_f32_e32 vcc, {a}, {b}", "v_cndmask_b32_e32 {d}, {a}, {b}, vcc"] * 4,
 "cmp e64 then cndmask sgpr": ["v_cmp_gt_f32_e64 s[22:23], {a}, {b}", "v_cndmask_b32_e64 {d}, {a}, {b}, s[22:23]"] * 4,
 "cmp, 3 mul, cndmask": ["v_cmp_gt_f32_e32 vcc, {a}, {b}", "v_mul_f32_e32 {d}, {a}, {b}", "v_mul_f32_e32 {d}, {a}, {b}", "v_cndmask_b32_e32 {d}, {a}, {b}, vcc"] * 2,
 "bfi": ["v_bfi_b32 {d}, {a}, {b}, {b}"] * 8,
 "sqrt": ["v_sqrt_f32_e32 {d}, {a}"] * 8,
 "rcp": ["v_rcp_f32_e32 {d}, {a}"] * 8,
 "div_scale": ["v_div_scale_f32 {d}, vcc, {a}, {b}, {a}"] * 8,
 "div_fmas": ["v_div_fmas_f32 {d}, {a}, {b}, {b}"] * 8,
 "div_fixup": ["v_div_fixup_f32 {d}, {a}, {b}, {b}"] * 8,
 "min_u32": ["v_min_u32_e32 {d}, {a}, {b}"] * 8,
 "and_or": ["v_and_or_b32 {d}, {a}, {b}, {b}"] * 8,
 "lshl_add": ["v_lshl_add_u32 {d}, {a}, 3, {b}"] * 8,
 "mov vgpr": ["v_mov_b32_e32 {d}, {a}"] * 8,
 "and vgpr": ["v_and_b32_e32 {d}, {a}, {b}"] * 8,
 "mul sgpr src0": ["v_mul_f32_e32 {d}, {s}, {b}"] * 8,
 "mul inline-const src0": ["v_mul_f32_e32 {d}, 2.0, {b}"] * 8,
 "fma vgpr": ["v_fma_f32 {d}, {a}, {b}, {b}"] * 8,
 "fmac vgpr": ["v_fmac_f32_e32 {d}, {a}, {b}"] * 8,
 "pk_fma vgpr": ["v_pk_fma_f32 {dp}, {ap}, {bp}, {bp}"] * 8,
 "pk_fma sgpr-pair src0": ["v_pk_fma_f32 {dp}, {sp}, {bp}, {bp}"] * 8,
 "pk_fma sgpr-pair src0, bcast src1": ["v_pk_fma_f32 {dp}, {sp}, {bp}, {dp} op_sel_hi:[1,0,1]"] * 8,
 "pk_mul vgpr": ["v_pk_mul_f32 {dp}, {ap}, {bp}"] * 8,
 "pk_add vgpr": ["v_pk_add_f32 {dp}, {ap}, {bp}"] * 8,
 "cmp vgpr -> vcc": ["v_cmp_gt_f32_e32 vcc, {a}, {d}"] * 8,
 "cmp sgpr -> vcc": ["v_cmp_gt_f32_e32 vcc, {s}, {d}"] * 8,
 "alt fma/mul": ["v_fma_f32 {d}, {a}, {b}, {b}", "v_mul_f32_e32 {d}, {a}, {b}"] * 4,
 "alt fma/mul/mul/mul": ["v_fma_f32 {d}, {a}, {b}, {b}", "v_mul_f32_e32 {d}, {a}, {b}", "v_mul_f32_e32 {d}, {a}, {b}", "v_mul_f32_e32 {d}, {a}, {b}"] * 2,
 "alt pk_fma/mul": ["v_pk_fma_f32 {dp}, {ap}, {bp}, {bp}", "v_mul_f32_e32 {d}, {a}, {b}"] * 4,
 "alt pk_fma/mul/mul/mul": ["v_pk_fma_f32 {dp}, {ap}, {bp}, {bp}", "v_mul_f32_e32 {d}, {a}, {b}", "v_mul_f32_e32 {d}, {a}, {b}", "v_mul_f32_e32 {d}, {a}, {b}"] * 2,
 "alt pk_fma(sgpr)/mul": ["v_pk_fma_f32 {dp}, {sp}, {bp}, {bp}", "v_mul_f32_e32 {d}, {a}, {b}"] * 4,
 "alt fma(sgpr)/mul": ["v_fma_f32 {d}, {s}, {b}, {b}", "v_mul_f32_e32 {d}, {a}, {b}"] * 4,
 "alt cmp/mul": ["v_cmp_gt_f32_e32 vcc, {a}, {d}", "v_mul_f32_e32 {d}, {a}, {b}"] * 4,
 "alt fma/fma(sgpr)": ["v_fma_f32 {d}, {a}, {b}, {b}", "v_fma_f32 {d}, {s}, {b}, {b}"] * 4,
 "alt mul/add": ["v_mul_f32_e32 {d}, {a}, {b}", "v_add_f32_e32 {d}, {a}, {b}"] * 4,
 "s_add only": ["s_add_u32 s24, s25, 1"] * 8,
 "s_and_b64 only": ["s_and_b64 s[26:27], s[28:29], exec"] * 8,
 "alt mul/s_add": ["v_mul_f32_e32 {d}, {a}, {b}", "s_add_u32 s24, s25, 1"] * 4,
 "alt fma/s_add": ["v_fma_f32 {d}, {a}, {b}, {b}", "s_add_u32 s24, s25, 1"] * 4,
 "alt fma/s_and_b64": ["v_fma_f32 {d}, {a}, {b}, {b}", "s_and_b64 s[26:27], s[28:29], exec"] * 4,
 "fma, 3 s_add": ["v_fma_f32 {d}, {a}, {b}, {b}", "s_add_u32 s24, s25, 1", "s_add_u32 s24, s25, 2", "s_add_u32 s24, s25, 3"] * 2,
 "mul, mul, s_add, s_add": ["v_mul_f32_e32 {d}, {a}, {b}", "v_mul_f32_e32 {d}, {a}, {b}", "s_add_u32 s24, s25, 1", "s_add_u32 s24, s25, 2"] * 2,
 "s_nop 0 only": ["s_nop 0"] * 8,
 "alt fma/s_nop 0": ["v_fma_f32 {d}, {a}, {b}, {b}", "s_nop 0"] * 4,
 "mul_lo_u32": ["v_mul_lo_u32 {d}, {a}, {b}"] * 8,
 "mad_u64_u32": ["v_mad_u64_u32 {dp}, s[26:27], {a}, {b}, {bp}"] * 8,
 "mad_u32_u24": ["v_mad_u32_u24 {d}, {a}, {b}, {b}"] * 8,
 "lshl_add_u32": ["v_lshl_add_u32 {d}, {a}, 3, {b}"] * 8,
 "mul dependent chain (1 reg)": ["v_mul_f32_e32 {d0}, {d0}, {b}"] * 8,
 "fma dependent chain (1 reg)": ["v_fma_f32 {d0}, {d0}, {b}, {b}"] * 8,
 # integer and bit forms a slab test could use instead of the f32 ones
 "int: max_i32": ["v_max_i32_e32 {d}, {a}, {b}"] * 8,
 "int: min_i32": ["v_min_i32_e32 {d}, {a}, {b}"] * 8,
 "int: max_u32": ["v_max_u32_e32 {d}, {a}, {b}"] * 8,
 "int: max3_i32": ["v_max3_i32 {d}, {a}, {b}, {b}"] * 8,
 "int: min3_i32": ["v_min3_i32 {d}, {a}, {b}, {b}"] * 8,
 "int: min3_u32": ["v_min3_u32 {d}, {a}, {b}, {b}"] * 8,
 "int: med3_i32": ["v_med3_i32 {d}, {a}, {b}, {b}"] * 8,
 "int: cmp_lt_i32 -> vcc": ["v_cmp_lt_i32_e32 vcc, {a}, {d}"] * 8,
 "int: cmp_lt_u32 -> vcc": ["v_cmp_lt_u32_e32 vcc, {a}, {d}"] * 8,
 "int: addc_co_u32 vcc": ["v_addc_co_u32_e32 {d}, vcc, {a}, {b}, vcc"] * 8,
 "int: add_u32": ["v_add_u32_e32 {d}, {a}, {b}"] * 8,
 "int: sub_u32": ["v_sub_u32_e32 {d}, {a}, {b}"] * 8,
 "int: sub_co_u32": ["v_sub_co_u32_e32 {d}, vcc, {a}, {b}"] * 8,
 "int: alignbit": ["v_alignbit_b32 {d}, {a}, {b}, 31"] * 8,
 "int: lshl_or": ["v_lshl_or_b32 {d}, {a}, 1, {b}"] * 8,
 "int: or3": ["v_or3_b32 {d}, {a}, {b}, {b}"] * 8,
 "int: xor": ["v_xor_b32_e32 {d}, {a}, {b}"] * 8,
 "int: or": ["v_or_b32_e32 {d}, {a}, {b}"] * 8,
 "int: lshlrev": ["v_lshlrev_b32_e32 {d}, 3, {a}"] * 8,
 "int: perm": ["v_perm_b32 {d}, {a}, {b}, {b}"] * 8,
 "int: cvt_i32_f32": ["v_cvt_i32_f32_e32 {d}, {a}"] * 8,
 "int: bfe_u32": ["v_bfe_u32 {d}, {a}, 3, 5"] * 8,
 "int: add3_u32": ["v_add3_u32 {d}, {a}, {b}, {b}"] * 8,
 "int: mov_b64": ["v_mov_b64 {dp}, {ap}"] * 8,
 "int: pk_mov_b32": ["v_pk_mov_b32 {dp}, {ap}, {bp}"] * 8,
 "int: cmp_gt_f32 e64 -> sgpr": ["v_cmp_gt_f32_e64 s[22:23], {a}, {d}"] * 8,
 "int: alt max_i32/min_i32": ["v_max_i32_e32 {d}, {a}, {b}", "v_min_i32_e32 {d}, {a}, {b}"] * 4,
 "int: alt pk_fma/max_i32/max_i32/min_i32": ["v_pk_fma_f32 {dp}, {ap}, {bp}, {bp}", "v_max_i32_e32 {d}, {a}, {b}", "v_max_i32_e32 {d}, {a}, {b}", "v_min_i32_e32 {d}, {a}, {b}"] * 2,
 "int: alt pk_fma/max3_f32/min3_f32/sub": ["v_pk_fma_f32 {dp}, {ap}, {bp}, {bp}", "v_max3_f32 {d}, {a}, {b}, {b}", "v_min3_f32 {d}, {a}, {b}, {b}", "v_sub_f32_e32 {d}, {a}, {b}"] * 2,
 "int: cmp f32 then addc": ["v_cmp_lt_f32_e32 vcc, {a}, {b}", "v_addc_co_u32_e32 {d}, vcc, {a}, {a}, vcc"] * 4,
 "int: mul_f32 dpp quad_perm": ["v_mul_f32_dpp {d}, {a}, {b} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"] * 8,
 "int: mov_b32 dpp row_shr:1": ["v_mov_b32_dpp {d}, {a} row_shr:1 row_mask:0xf bank_mask:0xf"] * 8,
}
out = ['#include <hip/hip_runtime.h>', '#include <cstdio>', 'typedef float f2 __attribute__((ext_vector_type(2)));',
       'template <int V> __global__ __launch_bounds__(256) void k(float *out, int iters, float sa, float sb) {',
       '  float a = threadIdx.x * 1e-3f, b = 1.0001f; f2 ap = {a, a}, bp = {b, b};',
       '  f2 r0 = {0,0}, r1 = {0,0}, r2 = {0,0}, r3 = {0,0}, r4 = {0,0}, r5 = {0,0}, r6 = {0,0}, r7 = {0,0};',
       '  float q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0, q7 = 0;',
       '  for (int it = 0; it < iters; ++it) {']
only = None
if "--only" in sys.argv:
    only = sys.argv[sys.argv.index("--only") + 1].split(",")
names = [n for n in V if only is None or any(o in n for o in only)]
for vi, name in enumerate(names):
    lines = []
    for rep in range(8):
        for k, t in enumerate(V[name]):
            lines.append(t.format(d="%%%d" % k, dp="%%%d" % (8 + k), d0="%0", a="%16", b="%17", ap="%18", bp="%19",
                                  s="s20", sp="s[20:21]"))
    body = "\\n ".join(lines)
    # destinations are 64-bit pairs; plain f32 ops write the low half
    out.append('    if (V == %d) asm volatile("%s" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "v"(ap), "v"(bp) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");' % (vi, body))
out += ['  }', '  out[blockIdx.x * 256 + threadIdx.x] = r0.x + r1.x + r2.x + r3.x + r4.y + r5.x + r6.x + r7.x + q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;', '}',
        'template <int V> void run(const char *name, float *out) {',
        '  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); const int iters = 10000;',
        '  printf("%-36s", name);',
        '  for (int w : {1, 2, 4, 8}) { int blocks = 256 * w;',
        '    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, 100, 0.5f, 0.25f); (void)hipDeviceSynchronize();',
        '    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);',
        '    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);',
        '    printf("  w=%d: %5.2f", w, ms * 1e-3 * 2.4e9 / ((double)w * iters * 64.0)); }',
        '  printf("\\n"); }',
        'int main() { float *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);']
for vi, name in enumerate(names):
    out.append('  run<%d>("%s", out);' % (vi, name))
out += ['  return 0; }']
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
SRC = os.path.join(ROOT, "build", "ub_issue2.hip")
open(SRC, "w").write("\n".join(out) + "\n")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", os.path.join(ROOT, "tools", "ub_issue2"), SRC])
