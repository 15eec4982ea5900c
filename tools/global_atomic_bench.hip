// Microbenchmark (diagnostic): rate of scattered float atomic adds to a large image in HBM, by memory scope.  The far-ray replay of
// config 5 retires ~11 device-scope atomics per ns (counters: every one a 64-byte write request to the fabric); would atomics that
// stop at the issuing XCD's L2 (workgroup scope) be faster?  -- they would need the targets partitioned by XCD to be correct.
//   hipcc --offload-arch=gfx950 -O3 tools/global_atomic_bench.hip -o tools/global_atomic_bench && tools/global_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int SCOPE, bool OWN>
__global__ __launch_bounds__(256) void k_scatter(float *img, uint32_t npix, int per_thread, uint32_t spread, int *xcc_seen) {
    uint32_t xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    if (threadIdx.x == 0 && xcc_seen) atomicAdd(&xcc_seen[(blockIdx.x & 7) * 16 + xcc], 1);
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    // a thread's targets wander around a base (as the shares of a tile's far rays do), `spread` pixels wide
    const uint32_t base = (uint32_t)(((uint64_t)s * npix) >> 32);
    for (int k = 0; k < per_thread; ++k) {
        s = s * 1664525u + 1013904223u;
        uint32_t p = base + (s >> 8) % spread;
        if (p >= npix) p -= npix;
        if (OWN) {                     // targets owned by the XCD that issues: 4096-pixel blocks dealt round-robin
            p = (p & ~(8u * 4096u - 1u)) | (xcc << 12) | (p & 4095u);
            if (p >= npix) p = xcc << 12;
        }
        __hip_atomic_fetch_add(img + p, 1.0f, __ATOMIC_RELAXED, SCOPE);
    }
}

int main() {
    const uint32_t npix = 16384u * 16384u;
    float *img;
    int *seen;
    hipMalloc(&img, sizeof(float) * (size_t)npix);
    hipMalloc(&seen, sizeof(int) * 128);
    hipMemset(img, 0, sizeof(float) * (size_t)npix);
    hipMemset(seen, 0, sizeof(int) * 128);
    const int blocks = 16384, per_thread = 8;                         // 33.5 M atomics
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, uint32_t spread) {
        kern<<<blocks, 256>>>(img, npix, per_thread, spread, seen);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) kern<<<blocks, 256>>>(img, npix, per_thread, spread, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = 3.0 * blocks * 256.0 * per_thread;
        printf("%-44s spread %8u: %7.3f ms per launch, %6.2f atomics per ns\n", name, spread, ms / 3, n / (ms * 1e6));
    };
    for (uint32_t spread : {64u, 4096u, 1u << 20, npix}) {
        run("agent scope (what atomicAdd is)", k_scatter<__HIP_MEMORY_SCOPE_AGENT, false>, spread);
        run("workgroup scope", k_scatter<__HIP_MEMORY_SCOPE_WORKGROUP, false>, spread);
        run("wavefront scope", k_scatter<__HIP_MEMORY_SCOPE_WAVEFRONT, false>, spread);
        run("agent scope, targets owned by the issuing XCD", k_scatter<__HIP_MEMORY_SCOPE_AGENT, true>, spread);
        run("workgroup scope, targets owned by the XCD", k_scatter<__HIP_MEMORY_SCOPE_WORKGROUP, true>, spread);
    }
    std::vector<int> h(128);
    hipMemcpy(h.data(), seen, sizeof(int) * 128, hipMemcpyDeviceToHost);
    printf("workgroups by (blockIdx %% 8) x XCC_ID:\n");
    for (int b = 0; b < 8; ++b) {
        printf("  b%%8=%d:", b);
        for (int x = 0; x < 8; ++x) printf(" %6d", h[b * 16 + x]);
        printf("\n");
    }
    // check of the sum under workgroup scope with owned targets: every add must have arrived
    hipMemset(img, 0, sizeof(float) * (size_t)npix);
    k_scatter<__HIP_MEMORY_SCOPE_WORKGROUP, true><<<blocks, 256>>>(img, npix, per_thread, 1u << 20, nullptr);
    hipDeviceSynchronize();
    std::vector<float> hi(npix);
    hipMemcpy(hi.data(), img, sizeof(float) * (size_t)npix, hipMemcpyDeviceToHost);
    double tot = 0;
    for (uint32_t i = 0; i < npix; ++i) tot += hi[i];
    printf("workgroup scope, owned targets: sum of the image %.0f (expected %.0f)\n", tot, (double)blocks * 256 * per_thread);
    hipMemset(img, 0, sizeof(float) * (size_t)npix);
    k_scatter<__HIP_MEMORY_SCOPE_WORKGROUP, false><<<blocks, 256>>>(img, npix, per_thread, 1u << 20, nullptr);
    hipDeviceSynchronize();
    hipMemcpy(hi.data(), img, sizeof(float) * (size_t)npix, hipMemcpyDeviceToHost);
    tot = 0;
    for (uint32_t i = 0; i < npix; ++i) tot += hi[i];
    printf("workgroup scope, targets NOT partitioned: sum of the image %.0f (expected %.0f)\n", tot, (double)blocks * 256 * per_thread);
    return 0;
}
