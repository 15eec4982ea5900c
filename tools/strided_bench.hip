// Microbenchmark (diagnostic): what do 16-byte pieces at a 32 KiB stride cost as LOADS vs as STORES?  (The Fresnel line
// kernel must transpose somewhere: each workgroup touches 4096 rows x 16 B of a row-major 4096x4096 complex image.)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int N = 4096;
__device__ __forceinline__ int xcd_group(int b, int ng) { const int q = ng >> 3, r = ng & 7, x = b & 7; return x * q + (x < r ? x : r) + (b >> 3); }
__global__ __launch_bounds__(768) void k_load(const float4* __restrict__ in, float* out) {
    const int g = xcd_group(blockIdx.x, N / 2);
    float acc = 0.f;
    for (int i = threadIdx.x; i < N; i += 768) { const float4 v = in[(size_t)i * (N / 2) + g]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
template <int AUX>
__global__ __launch_bounds__(768) void k_load_buf(const float4* __restrict__ in, float* out) {
    const int g = xcd_group(blockIdx.x, N / 2);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 0x7fffffff, 0x00020000);
    float acc = 0.f;
    for (int i = threadIdx.x; i < N; i += 768) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        typedef int v4i __attribute__((ext_vector_type(4)));
        const unsigned off = (unsigned)(((size_t)i * (N / 2) + g) * 16);
        v4i r = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX);
        acc += __int_as_float(r.x) + __int_as_float(r.y) + __int_as_float(r.z) + __int_as_float(r.w);
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(768) void k_store(float4* __restrict__ outp) {
    const int g = xcd_group(blockIdx.x, N / 2);
    for (int i = threadIdx.x; i < N; i += 768) outp[(size_t)i * (N / 2) + g] = make_float4(i, g, 1.f, 2.f);
}
__global__ __launch_bounds__(768) void k_load_rows(const float4* __restrict__ in, float* out) {   // contiguous 64 KiB per workgroup
    float acc = 0.f;
    for (int i = threadIdx.x; i < N; i += 768) { const float4 v = in[(size_t)blockIdx.x * N + i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(768) void k_store_rows(float4* __restrict__ outp) {
    for (int i = threadIdx.x; i < N; i += 768) outp[(size_t)blockIdx.x * N + i] = make_float4(i, 1.f, 1.f, 2.f);
}
int main() {
    float4* a; float* o; hipMalloc(&a, sizeof(float4) * (size_t)N * N / 2); hipMalloc(&o, 1 << 20);
    hipMemset(a, 0, sizeof(float4) * (size_t)N * N / 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* nm, auto f) { f(); hipDeviceSynchronize(); hipEventRecord(e0); for (int r = 0; r < 10; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-34s %.1f us per pass over 134 MB  (%.2f TB/s)\n", nm, ms * 100, 0.134 / (ms / 10 * 1e-3) / 1e3); };
    run("strided 16-B loads", [&] { k_load<<<N / 2, 768>>>(a, o); });
    run("strided 16-B buffer loads aux=0", [&] { k_load_buf<0><<<N / 2, 768>>>(a, o); });
    run("strided 16-B buffer loads sc0", [&] { k_load_buf<1><<<N / 2, 768>>>(a, o); });
    run("strided 16-B buffer loads nt", [&] { k_load_buf<2><<<N / 2, 768>>>(a, o); });
    run("strided 16-B buffer loads sc1", [&] { k_load_buf<16><<<N / 2, 768>>>(a, o); });
    run("strided 16-B buffer loads sc0 sc1", [&] { k_load_buf<17><<<N / 2, 768>>>(a, o); });
    run("strided 16-B stores", [&] { k_store<<<N / 2, 768>>>(a); });
    run("contiguous loads (64 KiB / WG)", [&] { k_load_rows<<<N / 2, 768>>>(a, o); });
    run("contiguous stores (64 KiB / WG)", [&] { k_store_rows<<<N / 2, 768>>>(a); });
    return 0;
}
