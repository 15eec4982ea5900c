// Microbenchmark (diagnostic): cost of LDS float atomics and wave shuffles on gfx950, by number of active lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_atomic(float* out, int active, int iters, int spread) {
    __shared__ float acc[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    if (lane < active) {
        for (int it = 0; it < iters; ++it) atomicAdd(&acc[(threadIdx.x * spread + it * 64) & 4095], 1.0f);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0];
}
template <class TT>
__global__ void k_atomic_int(float* out, int active, int iters, int spread) {
    __shared__ TT acc[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    if (lane < active) {
        for (int it = 0; it < iters; ++it) atomicAdd(&acc[(threadIdx.x * spread + it * 64) & 4095], (TT)(it + 1));
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)acc[0];
}
__global__ void k_rmw(float* out, int active, int iters, int spread) {   // plain read-add-write (no atomicity)
    __shared__ float acc[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    if (lane < active) {
        for (int it = 0; it < iters; ++it) { volatile float* p = &acc[(threadIdx.x * spread + it * 64) & 4095]; *p = *p + 1.0f; }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0];
}
__global__ void k_shfl(float* out, int iters) {
    float v = threadIdx.x;
    for (int it = 0; it < iters; ++it) v = __shfl_up(v, 1) + 1.f;
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void k_dpp(float* out, int iters) {
    float v = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
        v = __int_as_float(x) + 1.f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, blocks = 256 * 2, threads = 512;
    for (int spread : {1, 17}) for (int active : {64, 32, 8, 1}) {
        k_atomic<<<blocks, threads>>>(d, active, iters, spread); hipDeviceSynchronize();
        hipEventRecord(a); k_atomic<<<blocks, threads>>>(d, active, iters, spread); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // per CU: 2 blocks x 8 waves x iters atomic instructions
        double cyc = ms * 1e-3 * 2.4e9 / (2.0 * 8 * iters);
        printf("ds_add_f32 spread=%d active=%2d: %.3f ms  -> %.1f cycles per wave-instruction per CU\n", spread, active, ms, cyc);
    }
    for (int active : {64, 8}) {
        float ms;
        k_atomic_int<unsigned><<<blocks, threads>>>(d, active, iters, 1); hipDeviceSynchronize();
        hipEventRecord(a); k_atomic_int<unsigned><<<blocks, threads>>>(d, active, iters, 1); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("ds_add_u32 active=%2d: %.3f ms -> %.1f cycles per wave-instruction per CU\n", active, ms, ms * 1e-3 * 2.4e9 / (2.0 * 8 * iters));
        k_atomic_int<unsigned long long><<<blocks, threads>>>(d, active, iters, 1); hipDeviceSynchronize();
        hipEventRecord(a); k_atomic_int<unsigned long long><<<blocks, threads>>>(d, active, iters, 1); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("ds_add_u64 active=%2d: %.3f ms -> %.1f cycles per wave-instruction per CU\n", active, ms, ms * 1e-3 * 2.4e9 / (2.0 * 8 * iters));
        k_rmw<<<blocks, threads>>>(d, active, iters, 1); hipDeviceSynchronize();
        hipEventRecord(a); k_rmw<<<blocks, threads>>>(d, active, iters, 1); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("plain LDS read-add-write active=%2d: %.3f ms -> %.1f cycles per iteration per CU\n", active, ms, ms * 1e-3 * 2.4e9 / (2.0 * 8 * iters));
    }
    k_shfl<<<blocks, threads>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(a); k_shfl<<<blocks, threads>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("__shfl_up: %.3f ms -> %.1f cycles per wave-instruction per CU\n", ms, ms * 1e-3 * 2.4e9 / (2.0 * 8 * iters));
    k_dpp<<<blocks, threads>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(a); k_dpp<<<blocks, threads>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("dpp wave_shr:1: %.3f ms -> %.1f cycles per wave-instruction per CU\n", ms, ms * 1e-3 * 2.4e9 / (2.0 * 8 * iters));
    return 0;
}
