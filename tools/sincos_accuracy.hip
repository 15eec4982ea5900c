// Micro-benchmark: accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) against float64 sin / cos on
// [-0.5, 0.5) revolutions, next to sincosf on the same angle in radians.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/sincos_accuracy tools/sincos_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(int n, double *err) {   // err[0..3]: max |hw sin - ref|, |hw cos - ref|, |sincosf sin - ref|, |sincosf cos - ref|
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double rev = ((double)i + 0.37) / n - 0.5;            // revolutions, float64
        const float xr = (float)rev;
        const double ref_s = sin(6.283185307179586476925 * (double)xr), ref_c = cos(6.283185307179586476925 * (double)xr);
        float hs, hc;
        asm volatile("v_sin_f32 %0, %1" : "=v"(hs) : "v"(xr));
        asm volatile("v_cos_f32 %0, %1" : "=v"(hc) : "v"(xr));
        float ss, sc;
        sincosf((float)(6.283185307179586476925 * (double)xr), &ss, &sc);
        // reference for sincosf: the angle it was GIVEN (its float32 argument)
        const double ang = (double)(float)(6.283185307179586476925 * (double)xr);
        e0 = fmax(e0, fabs((double)hs - ref_s));
        e1 = fmax(e1, fabs((double)hc - ref_c));
        e2 = fmax(e2, fabs((double)ss - sin(ang)));
        e3 = fmax(e3, fabs((double)sc - cos(ang)));
    }
    for (int o = 32; o > 0; o >>= 1) {
        e0 = fmax(e0, __shfl_xor(e0, o)); e1 = fmax(e1, __shfl_xor(e1, o));
        e2 = fmax(e2, __shfl_xor(e2, o)); e3 = fmax(e3, __shfl_xor(e3, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax((unsigned long long *)&err[0], (unsigned long long)__double_as_longlong(e0));
        atomicMax((unsigned long long *)&err[1], (unsigned long long)__double_as_longlong(e1));
        atomicMax((unsigned long long *)&err[2], (unsigned long long)__double_as_longlong(e2));
        atomicMax((unsigned long long *)&err[3], (unsigned long long)__double_as_longlong(e3));
    }
}
int main() {
    double *d, h[4];
    hipMalloc(&d, 32);
    hipMemset(d, 0, 32);
    k<<<1024, 256>>>(1 << 26, d);
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("max abs error over 2^26 points of [-0.5, 0.5) revolutions: v_sin_f32 %.3e  v_cos_f32 %.3e | sincosf (vs its own float32 angle) sin %.3e cos %.3e\n", h[0], h[1], h[2], h[3]);
    return 0;
}
