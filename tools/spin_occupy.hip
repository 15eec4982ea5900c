// Diagnostic: occupies `nblocks` CUs' worth of workgroup slots for `micros` microseconds on a side stream -- a stand-in for
// the copy kernels of an RCCL transfer (few workgroups, a little LDS each) running next to the library's kernels.
//   hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libspin_occupy.so tools/spin_occupy.hip
#include <hip/hip_runtime.h>
__global__ void k_spin_occupy(unsigned long long ticks, float *sink) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = (float)threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    float v = 1.f;
    while (wall_clock64() - t0 < ticks) v = v * 1.0001f + lds[(threadIdx.x + 1) & 63];
    if (v == 123.456f) sink[0] = v;
}
extern "C" int spin_occupy(int nblocks, int threads, int lds_bytes, double micros, float *sink, void *stream) {
    k_spin_occupy<<<nblocks, threads, lds_bytes, (hipStream_t)stream>>>((unsigned long long)(micros * 100.0), sink);   // 100 MHz wall clock
    return (int)hipGetLastError();
}
