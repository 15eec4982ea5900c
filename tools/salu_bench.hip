// Diagnostic: scalar-ALU issue rate per CU (is the scalar unit shared by the four SIMDs?).  W waves per CU run a block of
// independent s_add_u32; cycles per instruction per CU tells.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out, int iters) {
    unsigned a0 = blockIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                         "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1"
                         : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3), "+s"(a4), "+s"(a5), "+s"(a6), "+s"(a7)
                         :
                         : "scc");
    }
    if (threadIdx.x == 0) out[blockIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    unsigned* d; hipMalloc(&d, 4 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves : {1, 4, 8, 16}) {
        k<<<256, 64 * waves>>>(d, 100);
        hipEventRecord(e0); k<<<256, 64 * waves>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess || h != 28u + 8u * 8u * (unsigned)iters) printf("  (check: %s, out[0] = %u)\n", hipGetErrorString(err), h);
        const double inst_per_cu = (double)waves * iters * 64;
        printf("%2d waves per CU: %.3f ms -> %.3f ns per scalar instruction per CU (%.2f per cycle at 2.4 GHz)\n", waves, ms,
               ms * 1e6 / inst_per_cu, inst_per_cu / (ms * 1e-3 * 2.4e9));
    }
    return 0;
}
