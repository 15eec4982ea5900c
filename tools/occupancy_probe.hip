// Diagnostic: how many workgroups of T threads, V VGPRs and S bytes of dynamic LDS does the dispatcher really keep
// resident per CU, and how does it spread a workgroup's waves over the four SIMDs?  (The occupancy API said 2 for the
// 384-thread / 72 KiB / 164-VGPR Fresnel line kernel, the phase stamps said 1.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };
template <int VG, int LB>
__global__ void __launch_bounds__(LB) k_spin(Rec* r, int spin_ticks, int waves_per_block) {
    extern __shared__ float lds[];
    float keep[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) keep[i] = threadIdx.x * 0.5f + i;
    const unsigned long long t0 = wall_clock64();
    lds[threadIdx.x] = 1.f;
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
        for (int i = 0; i < VG; ++i) keep[i] = keep[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += keep[i];
    if (s == 123.456f) lds[0] = s;
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        Rec& o = r[blockIdx.x * waves_per_block + (threadIdx.x >> 6)];
        o.t0 = t0; o.t1 = wall_clock64(); o.hw = hw; o.xcc = xcc;
    }
}
template <int VG, int LB>
void probe(int T, size_t S) {
    const int nb = 2048, W = T / 64;
    Rec* d; hipMalloc(&d, sizeof(Rec) * nb * W);
    auto kern = k_spin<VG, LB>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S);
    int api = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, kern, T, S);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)kern);
    kern<<<nb, T, S>>>(d, 2000, W);   // 20 us at 100 MHz
    hipDeviceSynchronize();
    std::vector<Rec> h(nb * W);
    hipMemcpy(h.data(), d, sizeof(Rec) * nb * W, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0;
    std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;   // per CU: (+1 at start, -1 at end) of blocks
    std::map<std::vector<int>, int> spread;                                   // waves per SIMD pattern -> count
    for (int b = 0; b < nb; ++b) {
        const Rec& r0 = h[b * W];
        lo = std::min(lo, r0.t0); hi = std::max(hi, r0.t1);
        const unsigned cu = ((r0.xcc & 15) << 16) | (r0.hw & 0xff00);   // xcc | se, sh, cu
        ev[cu].push_back({r0.t0, +1}); ev[cu].push_back({r0.t1, -1});
        std::vector<int> s(4, 0);
        for (int w = 0; w < W; ++w) s[(h[b * W + w].hw >> 4) & 3]++;
        spread[s]++;
    }
    int peak_max = 0; double peak_avg = 0;
    for (auto& kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int cur = 0, pk = 0;
        for (auto& e : kv.second) { cur += e.second; pk = std::max(pk, cur); }
        peak_max = std::max(peak_max, pk); peak_avg += pk;
    }
    printf("T=%4d LDS=%6zu VGPRs=%3d: API %d/CU; %zu CUs seen, peak resident blocks per CU avg %.2f max %d; span %.0f us; waves/SIMD:",
           T, S, fa.numRegs, api, ev.size(), peak_avg / ev.size(), peak_max, (hi - lo) * 0.01);
    for (auto& kv : spread) printf(" [%d %d %d %d]x%d", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
    printf("\n");
    hipFree(d);
}
int main() {
    probe<16, 384>(384, 1024);  probe<60, 384>(384, 1024);  probe<90, 384>(384, 1024);  probe<120, 384>(384, 1024);
    probe<150, 384>(384, 1024); probe<16, 384>(384, 73728); probe<90, 384>(384, 73728); probe<120, 384>(384, 73728);
    probe<150, 384>(384, 73728);
    probe<90, 256>(256, 49152); probe<120, 256>(256, 49152); probe<150, 256>(256, 49152);
    probe<90, 512>(512, 73728); probe<120, 512>(512, 73728);
    probe<150, 768>(768, 1024);
    return 0;
}
