// Calibration of the SQ counters the roofline figure is built from (run under rocprofv3 --pmc,
// see tools/calibrate_counters.sh).  Every kernel issues a KNOWN number of one VALU instruction
// with a KNOWN number of active lanes, so that the units of SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU,
// SQ_THREAD_CYCLES_VALU and of the per-type counters can be read off the CSV:
//   k<V, LANES>: 1024 blocks x 256 threads, ITERS x 64 instructions per wave inside `if (lane < LANES)`
//   V = 0 v_mul_f32 (VGPR operands), 1 v_fma_f32, 2 v_pk_fma_f32, 3 v_cmp_gt_f32, 4 v_mov_b32, 5 v_rcp_f32,
//       6 v_mul_f32 with an SGPR operand
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define R8(x) x x x x x x x x
template <int V, int LANES>
__global__ __launch_bounds__(256) void k(float *out, int iters, float sa) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f, q = 0.0f;
    f2 ap = {a, a}, bp = {b, b}, r = {0.0f, 0.0f};
    if ((threadIdx.x & 63) < LANES) {
        for (int it = 0; it < iters; ++it) {
            if (V == 0) asm volatile(R8(R8("v_mul_f32_e32 %0, %1, %2\n")) : "+v"(q) : "v"(a), "v"(b));
            if (V == 1) asm volatile(R8(R8("v_fma_f32 %0, %1, %2, %2\n")) : "+v"(q) : "v"(a), "v"(b));
            if (V == 2) asm volatile(R8(R8("v_pk_fma_f32 %0, %1, %2, %2\n")) : "+v"(r) : "v"(ap), "v"(bp));
            if (V == 3) asm volatile(R8(R8("v_cmp_gt_f32_e32 vcc, %1, %2\n")) : "+v"(q) : "v"(a), "v"(b) : "vcc");
            if (V == 4) asm volatile(R8(R8("v_mov_b32_e32 %0, %1\n")) : "+v"(q) : "v"(a));
            if (V == 5) asm volatile(R8(R8("v_rcp_f32_e32 %0, %1\n")) : "+v"(q) : "v"(a));
            if (V == 6) asm volatile(R8(R8("v_mul_f32_e32 %0, %1, %2\n")) : "+v"(q) : "s"(sa), "v"(b));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = q + r.x + r.y;
}
template <int V, int LANES> void run(float *out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000, blocks = 1024;
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, LANES>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // 4 waves per block, one block per CU at a time x 4 rounds: cycles per instruction per SIMD
    printf("V=%d lanes=%2d: %.3f ms, %.0f wave-instructions, %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", V, LANES, ms,
           (double)blocks * 4 * iters * 64, ms * 1e-3 * 2.4e9 / ((double)blocks * 4 / 1024 * iters * 64));
}
int main() {
    float *out; (void)hipMalloc(&out, 1024 * 256 * 4);
    run<0, 64>(out); run<0, 16>(out); run<1, 64>(out); run<1, 16>(out); run<2, 64>(out); run<2, 16>(out);
    run<3, 64>(out); run<4, 64>(out); run<5, 64>(out); run<6, 64>(out);
    (void)hipDeviceSynchronize();
    return 0;
}
