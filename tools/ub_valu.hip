// Microbenchmark: VALU issue rate of the sphere rejection body on gfx950.
// Each wave runs ITER iterations of an 8-sphere body (56 v_fma + 8 v_cmp, sphere
// records in SGPRs taken from kernel arguments, no memory traffic in the loop).
// Prints cycles per VALU instruction per SIMD for several occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Rec { float v[32]; };

template <int VARIANT>
__global__ __launch_bounds__(256) void body(Rec r, float *out, int iters) {
    float ux = threadIdx.x * 1e-3f, uy = 0.5f, uz = 0.25f, nou = -1.0f;
    float ax = 2.0f, ay = 1.0f + threadIdx.x * 1e-4f, az = 0.3f, oo = 7.0f;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float cx = r.v[4 * k], cy = r.v[4 * k + 1], cz = r.v[4 * k + 2], thr = r.v[4 * k + 3];
            float m = __builtin_fmaf(cx, ux, __builtin_fmaf(cy, uy, __builtin_fmaf(cz, uz, nou)));
            float w = __builtin_fmaf(cx, ax, __builtin_fmaf(cy, ay, __builtin_fmaf(cz, az, oo)));
            float g = __builtin_fmaf(-m, m, w);
            acc |= __builtin_amdgcn_ballot_w64(!(g > thr));
        }
        } else if (VARIANT == 2) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 vux = {ux, ux}, vuy = {uy, uy}, vuz = {uz, uz}, vnou = {nou, nou};
            const f2 vax = {ax, ax}, vay = {ay, ay}, vaz = {az, az}, voo = {oo, oo};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f2 cx = {r.v[8 * k], r.v[8 * k + 1]}, cy = {r.v[8 * k + 2], r.v[8 * k + 3]};
                const f2 cz = {r.v[8 * k + 4], r.v[8 * k + 5]};
                f2 m = __builtin_elementwise_fma(cx, vux, __builtin_elementwise_fma(cy, vuy, __builtin_elementwise_fma(cz, vuz, vnou)));
                f2 w = __builtin_elementwise_fma(cx, vax, __builtin_elementwise_fma(cy, vay, __builtin_elementwise_fma(cz, vaz, voo)));
                f2 g = __builtin_elementwise_fma(-m, m, w);
                acc |= __builtin_amdgcn_ballot_w64(!(g.x > r.v[8 * k + 6]));
                acc |= __builtin_amdgcn_ballot_w64(!(g.y > r.v[8 * k + 7]));
            }
        } else {
            float m[8], w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { m[k] = __builtin_fmaf(r.v[4 * k + 2], uz, nou); w[k] = __builtin_fmaf(r.v[4 * k + 2], az, oo); }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) { m[k] = __builtin_fmaf(r.v[4 * k + 1], uy, m[k]); w[k] = __builtin_fmaf(r.v[4 * k + 1], ay, w[k]); }
#pragma unroll
            for (int k = 0; k < 8; ++k) { m[k] = __builtin_fmaf(r.v[4 * k], ux, m[k]); w[k] = __builtin_fmaf(r.v[4 * k], ax, w[k]); }
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = __builtin_fmaf(-m[k], m[k], w[k]);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc |= __builtin_amdgcn_ballot_w64(!(w[k] > r.v[4 * k + 3]));
        }
        // keep the loop from being hoisted: perturb a ray component with the iteration
        oo += 1e-6f; ux += 1e-6f; uy -= 1e-6f; uz += 2e-6f; nou += 1e-6f; ax -= 1e-6f; ay += 1e-6f; az += 3e-6f;

    }
    if (acc == 0x123456789ull) out[threadIdx.x] = oo;
}

int main() {
    float *out; hipMalloc(&out, 4096);
    Rec r; for (int i = 0; i < 32; ++i) r.v[i] = 0.1f * (i + 1);
    for (int i = 0; i < 8; ++i) r.v[4 * i + 3] = -1e30f;   // never "pass"
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int variant = 0; variant < 3; ++variant)
    for (int waves_per_simd : {1, 2, 3, 4, 6, 8}) {
        int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves = 1 per SIMD) per block
        auto kern = variant == 2 ? body<2> : variant ? body<1> : body<0>;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, r, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, r, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double valu = (double)waves_per_simd * iters * 64.0;   // wave-instr per SIMD
        double cyc = ms * 1e-3 * 2.4e9;
        printf("variant %d waves/SIMD %d: %.3f ms, %.2f cycles per VALU wave-instr per SIMD (at 2.4 GHz)\n",
               variant, waves_per_simd, ms, cyc / valu);
    }
    return 0;
}
