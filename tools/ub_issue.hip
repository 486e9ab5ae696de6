// Microbenchmark: issue cost of f32 VALU forms on gfx950 (cycles per wave-instruction
// per SIMD) for 1..8 waves per SIMD.  Streams of 64 independent instructions per loop trip.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int V>
__global__ __launch_bounds__(256) void k(float *out, int iters, float sa, float sb) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {   // VOP2 fmac, VGPR operands
            asm volatile(REP8("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                              "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (V == 1) {   // VOP3 fma, VGPR operands
            asm volatile(REP8("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                              "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (V == 2) {   // VOP2 fmac, SGPR src0
            asm volatile(REP8("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                              "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(sa), "v"(b));
        } else if (V == 3) {   // VOP3 fma, SGPR src0
            asm volatile(REP8("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                              "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(sa), "v"(b));
        } else if (V == 4) {   // VOP2 mul, VGPR
            asm volatile(REP8("v_mul_f32_e32 %0, %8, %9\n v_mul_f32_e32 %1, %8, %9\n v_mul_f32_e32 %2, %8, %9\n v_mul_f32_e32 %3, %8, %9\n"
                              "v_mul_f32_e32 %4, %8, %9\n v_mul_f32_e32 %5, %8, %9\n v_mul_f32_e32 %6, %8, %9\n v_mul_f32_e32 %7, %8, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (V == 5) {   // packed fma: two f32 per lane per instruction
            asm volatile(REP8("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                              "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n")
                         : "+v"(*(double *)&r0), "+v"(*(double *)&r2), "+v"(*(double *)&r4), "+v"(*(double *)&r6)
                         : "v"((double)a), "v"((double)b));
        } else if (V == 7) {   // VOP3 fma, independent (dest not a source)
            asm volatile(REP8("v_fma_f32 %0, %8, %9, %9\n v_fma_f32 %1, %8, %9, %9\n v_fma_f32 %2, %8, %9, %9\n v_fma_f32 %3, %8, %9, %9\n"
                              "v_fma_f32 %4, %8, %9, %9\n v_fma_f32 %5, %8, %9, %9\n v_fma_f32 %6, %8, %9, %9\n v_fma_f32 %7, %8, %9, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (V == 8) {   // VOP2 mul, dependent at distance 8 (r = r * b)
            asm volatile(REP8("v_mul_f32_e32 %0, %0, %9\n v_mul_f32_e32 %1, %1, %9\n v_mul_f32_e32 %2, %2, %9\n v_mul_f32_e32 %3, %3, %9\n"
                              "v_mul_f32_e32 %4, %4, %9\n v_mul_f32_e32 %5, %5, %9\n v_mul_f32_e32 %6, %6, %9\n v_mul_f32_e32 %7, %7, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (V == 9) {   // VOP3-encoded mul
            asm volatile(REP8("v_mul_f32_e64 %0, %8, %9\n v_mul_f32_e64 %1, %8, %9\n v_mul_f32_e64 %2, %8, %9\n v_mul_f32_e64 %3, %8, %9\n"
                              "v_mul_f32_e64 %4, %8, %9\n v_mul_f32_e64 %5, %8, %9\n v_mul_f32_e64 %6, %8, %9\n v_mul_f32_e64 %7, %8, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (V == 10) {   // VOPC compare to vcc
            asm volatile(REP8("v_cmp_gt_f32_e32 vcc, %8, %0\n v_cmp_gt_f32_e32 vcc, %8, %1\n v_cmp_gt_f32_e32 vcc, %8, %2\n v_cmp_gt_f32_e32 vcc, %8, %3\n"
                              "v_cmp_gt_f32_e32 vcc, %8, %4\n v_cmp_gt_f32_e32 vcc, %8, %5\n v_cmp_gt_f32_e32 vcc, %8, %6\n v_cmp_gt_f32_e32 vcc, %8, %7\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
        } else if (V == 11) {   // packed mul
            asm volatile(REP8("v_pk_mul_f32 %0, %4, %5\n v_pk_mul_f32 %1, %4, %5\n v_pk_mul_f32 %2, %4, %5\n v_pk_mul_f32 %3, %4, %5\n"
                              "v_pk_mul_f32 %0, %4, %5\n v_pk_mul_f32 %1, %4, %5\n v_pk_mul_f32 %2, %4, %5\n v_pk_mul_f32 %3, %4, %5\n")
                         : "+v"(*(double *)&r0), "+v"(*(double *)&r2), "+v"(*(double *)&r4), "+v"(*(double *)&r6)
                         : "v"((double)a), "v"((double)b));
        } else if (V == 12) {   // VOP2 add with SGPR src0
            asm volatile(REP8("v_add_f32_e32 %0, %8, %9\n v_add_f32_e32 %1, %8, %9\n v_add_f32_e32 %2, %8, %9\n v_add_f32_e32 %3, %8, %9\n"
                              "v_add_f32_e32 %4, %8, %9\n v_add_f32_e32 %5, %8, %9\n v_add_f32_e32 %6, %8, %9\n v_add_f32_e32 %7, %8, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(sa), "v"(b));
        } else if (V == 13) {   // VOP2 fmac, independent-ish: 64 distinct... use dest chain distance 8 but srcs vgpr (same as V0) with v_fmac reading different regs
            asm volatile(REP8("v_fmac_f32_e32 %0, %8, %9\n v_mul_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_mul_f32_e32 %3, %8, %9\n"
                              "v_fmac_f32_e32 %4, %8, %9\n v_mul_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_mul_f32_e32 %7, %8, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (V == 6) {   // v_cmp to an SGPR pair (VOP3) with SGPR operand
            asm volatile(REP8("v_cmp_gt_f32_e64 s[20:21], %8, %0\n v_cmp_gt_f32_e64 s[22:23], %8, %1\n v_cmp_gt_f32_e64 s[24:25], %8, %2\n v_cmp_gt_f32_e64 s[26:27], %8, %3\n"
                              "v_cmp_gt_f32_e64 s[20:21], %8, %4\n v_cmp_gt_f32_e64 s[22:23], %8, %5\n v_cmp_gt_f32_e64 s[24:25], %8, %6\n v_cmp_gt_f32_e64 s[26:27], %8, %7\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(sa), "v"(b)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

template <int V>
void run(const char *name, float *out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    printf("%-34s", name);
    for (int w : {1, 2, 4, 8}) {
        int blocks = 256 * w;
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, 100, 0.5f, 0.25f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  w=%d: %5.2f", w, ms * 1e-3 * 2.4e9 / ((double)w * iters * 64.0));
    }
    printf("   cycles/instr/SIMD @2.4GHz\n");
}

int main() {
    float *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_fmac_f32_e32 (VOP2) vgpr", out);
    run<1>("v_fma_f32 (VOP3) vgpr", out);
    run<2>("v_fmac_f32_e32 (VOP2) sgpr src0", out);
    run<3>("v_fma_f32 (VOP3) sgpr src0", out);
    run<4>("v_mul_f32_e32 (VOP2) vgpr", out);
    run<5>("v_pk_fma_f32 (2 x f32)", out);
    run<6>("v_cmp_gt_f32_e64 -> sgpr pair", out);
    run<7>("v_fma_f32 (VOP3) no dest dependency", out);
    run<8>("v_mul_f32_e32 dependent dist 8", out);
    run<9>("v_mul_f32_e64 (VOP3)", out);
    run<10>("v_cmp_gt_f32_e32 -> vcc", out);
    run<11>("v_pk_mul_f32 (2 x f32)", out);
    run<12>("v_add_f32_e32 sgpr src0", out);
    run<13>("alternating fmac / mul", out);
    return 0;
}
