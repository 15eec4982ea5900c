// Diagnostic for VERDICT r2 item 5a: could the radix-16 middle stage of the Fresnel line engine run on the matrix pipe
// (v_mfma_f32_16x16x4_f32, exact fp32) BESIDE the vector butterflies?  Measures, with the line kernel's occupancy
// (768-thread workgroups = 3 waves per SIMD, one per CU):
//   (a) the fp32 MFMA rate of a CU (all waves issue v_mfma_f32_16x16x4_f32),
//   (b) the packed-fp32 VALU rate (all waves issue v_pk_fma_f32),
//   (c) both at once: waves 0-7 on the VALU, waves 8-11 on the matrix pipe (one MFMA wave per SIMD) -- do the pipes overlap?
// A 16-point complex DFT as a dense product is a 32 x 32 real matrix per vector: 1024 MACs, against ~100 packed instructions
// (~400 flops) for the FFT butterfly in registers -- the matrix pipe would need 5x the vector pipe's flop rate to break even.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// kind: 0 = MFMA on every wave, 1 = pk_fma on every wave, 2 = waves 8..11 MFMA, the rest pk_fma
__global__ void __launch_bounds__(768) k(float *out, int iters, int kind) {
    const int wave = threadIdx.x >> 6;
    const bool mfma = kind == 0 || (kind == 2 && wave >= 8);
    v4f acc[4];
    v2f a[8];
    for (int i = 0; i < 4; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) a[i] = (v2f){threadIdx.x * 1e-3f + i, 1.f + i};
    const float x = 1.0001f + threadIdx.x * 1e-7f, y = 0.9999f;
    const v2f c = (v2f){1.0001f, 0.9999f};
    if (mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float run(int kind, int iters) {
    float *d;
    hipMalloc(&d, 256 * 768 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<<<256, 768>>>(d, 100, kind);
    hipEventRecord(e0);
    k<<<256, 768>>>(d, iters, kind);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms;
}

int main() {
    const int iters = 20000;
    const float m0 = run(0, iters), m1 = run(1, iters), m2 = run(2, iters);
    // per SIMD: 3 waves x iters x 32 instructions; an MFMA 16x16x4 is 16*16*4 = 1024 MACs per wave instruction
    const double inst = 3.0 * iters * 32;
    printf("(a) MFMA only      : %.3f ms -> %.2f ns per v_mfma_f32_16x16x4_f32 per SIMD = %.1f TFLOP/s fp32 on 256 CUs\n", m0,
           m0 * 1e6 / inst, 2.0 * 1024 * inst * 4 * 256 / (m0 * 1e-3) / 1e12);
    printf("(b) pk_fma only    : %.3f ms -> %.2f ns per v_pk_fma_f32 per SIMD = %.1f TFLOP/s fp32 on 256 CUs\n", m1, m1 * 1e6 / inst,
           2.0 * 128 * inst * 4 * 256 / (m1 * 1e-3) / 1e12);
    // (c): per SIMD 2 VALU waves + 1 MFMA wave, each iters x 32 instructions
    printf("(c) 2 VALU waves + 1 MFMA wave per SIMD: %.3f ms; alone they would take %.3f (VALU share) and %.3f ms (MFMA share): "
           "%s\n", m2, m1 * 2.0 / 3.0, m0 / 3.0, m2 < 0.9 * (m1 * 2.0 / 3.0 + m0 / 3.0) ? "the pipes overlap" : "no overlap");
    const double t_mfma_dft = 2.0 * 1152 * (32.0 * 32.0) / 1024.0 / 4.0 * (m0 * 1e6 / inst);   // ns per round: 1152 slabs, fwd + inv, 4 SIMDs
    printf("middle stage of one round (1152 slabs of 16 points, forward + inverse) as dense 32x32 real products on the matrix pipe: "
           "%.1f us per CU (the vector butterflies take 1.8-3.2 us)\n", t_mfma_dft / 1e3);
    return 0;
}
