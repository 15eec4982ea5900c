// Microbenchmark (diagnostic): what does an LDS write / read cost per wave instruction on gfx950 by width and alignment?
// (The line kernels issue ~190 ds_write_b64 per engine thread and round; DESIGN.md section 7, open lead.)  One workgroup of 512
// threads per CU (8 waves, as the engine), every lane its own conflict-free slot; shader-clock cycles per wave instruction as
// the CU sees them (all 8 waves issuing).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int REP = 256;
// MODE 0: ds_write_b64, lane stride 8 B; 1: ds_write_b128 aligned, lane stride 16 B; 2: ds_write_b128 at 8-byte alignment (odd slot);
// 3: ds_read_b64; 4: ds_read_b128 aligned; 5: ds_read_b128 at 8-byte alignment; 6: ds_write_b64 x2 to adjacent slots (what one b128 replaces)
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long *out, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int w = t >> 6, ln = t & 63;
    // a wave's region: 64 lanes x 16 B = 1 KiB (+ 8 B when misaligned), 8 waves: 8 KiB + slack; REP instructions walk 8 regions
    char *base = smem + w * 9 * 1024 + ln * ((MODE == 0 || MODE == 3) ? 8 : 16) + ((MODE == 2 || MODE == 5) ? 8 : 0);
    v4f acc = {seed, seed, seed, seed};
    __syncthreads();
    const unsigned long long c0 = clock64();
#pragma unroll 16
    for (int i = 0; i < REP; ++i) {
        char *p = base + (i & 7) * 1024;
        if (MODE == 0) *(volatile __attribute__((address_space(3))) v2f *)(__attribute__((address_space(3))) char *)p = (v2f){acc.x, acc.y};
        if (MODE == 1 || MODE == 2) asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) char *)p), "v"(acc) : "memory");
        if (MODE == 6) {
            *(volatile __attribute__((address_space(3))) v2f *)(__attribute__((address_space(3))) char *)p = (v2f){acc.x, acc.y};
            *(volatile __attribute__((address_space(3))) v2f *)(__attribute__((address_space(3))) char *)(p + 8) = (v2f){acc.z, acc.w};
        }
        if (MODE == 7) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:64" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) char *)(smem + w * 9 * 1024 + ln * 8 + (i & 7) * 1024)), "v"((v2f){acc.x, acc.y}), "v"((v2f){acc.z, acc.w}) : "memory");      // two 8-byte stores 512 B apart: both conflict-free
        if (MODE == 8) asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:0 offset1:1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) char *)(smem + w * 9 * 1024 + ln * 8 + (i & 7) * 1024)), "v"((v2f){acc.x, acc.y}), "v"((v2f){acc.z, acc.w}) : "memory");
        if (MODE == 9) {
            v4f r;
            asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:64\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) char *)(smem + w * 9 * 1024 + ln * 8 + (i & 7) * 1024)) : "memory");
            acc.x += r.x;
        }
        if (MODE == 3) { const v2f r = *(volatile __attribute__((address_space(3))) v2f *)(__attribute__((address_space(3))) char *)p; acc.x += r.x; }
        if (MODE == 4 || MODE == 5) {
            v4f r;
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) char *)p) : "memory");
            acc.x += r.x;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long c1 = clock64();
    if (t == 0) out[blockIdx.x] = c1 - c0;
    if (acc.x == 12345.678f) out[1000] = 1;
}
int main() {
    unsigned long long *d, h[256];
    (void)hipMalloc(&d, 2048 * sizeof(unsigned long long));
    const char *names[] = {"ds_write_b64  (8 B / lane)", "ds_write_b128 (16 B / lane, aligned)", "ds_write_b128 at 8-byte alignment", "ds_read_b64", "ds_read_b128 aligned (waited one by one)",
                           "ds_read_b128 at 8-byte alignment (waited)", "2 x ds_write_b64 to adjacent slots", "ds_write2_b64 (two 8-byte stores 512 B apart)", "ds_write2st64_b64 (the same)", "ds_read2_b64 (waited)"};
    auto run = [&](int m, auto kern) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        kern<<<256, 512, 80 * 1024>>>(d, 1.f);
        kern<<<256, 512, 80 * 1024>>>(d, 1.f);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < 256; ++i) s += (double)h[i];
        s /= 256;
        printf("%-44s %8.1f shader cycles per wave instruction with 8 waves issuing  (%.2f per instruction and CU)\n", names[m], s / REP, s / REP / 8);
    };
    run(0, k<0>); run(1, k<1>); run(2, k<2>); run(6, k<6>); run(7, k<7>); run(8, k<8>); run(3, k<3>); run(4, k<4>); run(5, k<5>); run(9, k<9>);
    return 0;
}
