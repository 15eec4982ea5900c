// Diagnostic: issue cost per wave64 instruction on one SIMD (gfx950) of the VALU forms the Fresnel butterflies use.
// 768-thread workgroups (3 waves per SIMD, as the line kernel), one per CU; cycles from s_memtime-equivalent wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ void __launch_bounds__(768) k(float* out, int iters) {
    v2f a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (v2f){threadIdx.x * 1e-3f + i, 1.f + i};
    v2f c = (v2f){1.0001f, 0.9999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i].x) : "v"(c.x));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
                if (KIND == 3) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].x) : "v"(c.x));
                if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name) {
    float* d; hipMalloc(&d, 256 * 768 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<KIND><<<256, 768>>>(d, 100);
    hipEventRecord(e0); k<KIND><<<256, 768>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = 3.0 * iters * 32;   // 3 waves per SIMD
    printf("%-34s %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / inst_per_simd,
           ms * 1e6 / inst_per_simd * 2.4);
    hipFree(d);
}
int main() {
    run<0>("v_pk_fma_f32"); run<1>("v_fma_f32"); run<2>("v_pk_add_f32 op_sel+neg"); run<3>("v_mov_b32"); run<4>("v_pk_mul_f32");
    run<5>("v_add_u32");
    return 0;
}
