// fft_regs.hpp -- small DFTs held entirely in registers (one thread, R complex values), natural order in and out.
// Radix 2/3/4 kernels + one Cooley-Tukey split with compile-time twiddles (fft_consts.hpp).  Every loop has a
// compile-time trip count and every array index is a constant after unrolling, so the arrays live in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fft_consts.hpp"

namespace psx {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
// a * conj(b)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
// multiply by -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 rot90(float2 a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<2, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[2]) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <bool INV>
struct Dft<3, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[3]) {
        const float2 s = cadd(v[1], v[2]), d = csub(v[1], v[2]);
        const float2 m = make_float2(fmaf(-0.5f, s.x, v[0].x), fmaf(-0.5f, s.y, v[0].y));
        const float h = 0.86602540378443865f;
        const float2 n = rot90<INV>(make_float2(h * d.x, h * d.y));   // -i*h*d (forward)
        v[0] = cadd(v[0], s);
        v[1] = cadd(m, n);
        v[2] = csub(m, n);
    }
};

template <bool INV>
struct Dft<4, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[4]) {
        const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]);
        const float2 t3 = rot90<INV>(csub(v[1], v[3]));
        v[0] = cadd(t0, t2);
        v[1] = cadd(t1, t3);
        v[2] = csub(t0, t2);
        v[3] = csub(t1, t3);
    }
};

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// w_N^t (forward: exp(-2 pi i t/N); inverse: conjugate) as a constant expression
template <int N, int T, bool INV>
__device__ __forceinline__ float2 twiddle_const() {
    constexpr float c = TwConst<N>::c[T], s = TwConst<N>::s[T];
    return make_float2(c, INV ? s : -s);
}

// N = R1*R2:  n = R2*n1 + n2,  k = k1 + R1*k2
template <int R1, int R2, bool INV>
__device__ __forceinline__ void dft_split(float2 (&v)[R1 * R2]) {
    constexpr int N = R1 * R2;
    float2 y[N];
    static_for<0, R2>([&](auto n2c) __attribute__((always_inline)) {
        constexpr int n2 = decltype(n2c)::value;
        float2 a[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) a[n1] = v[R2 * n1 + n2];
        Dft<R1, INV>::run(a);
        static_for<0, R1>([&](auto k1c) __attribute__((always_inline)) {
            constexpr int k1 = decltype(k1c)::value;
            constexpr int t = (n2 * k1) % N;
            if constexpr (t == 0)
                y[n2 * R1 + k1] = a[k1];
            else
                y[n2 * R1 + k1] = cmul(a[k1], twiddle_const<N, t, INV>());
        });
    });
    static_for<0, R1>([&](auto k1c) __attribute__((always_inline)) {
        constexpr int k1 = decltype(k1c)::value;
        float2 b[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) b[n2] = y[n2 * R1 + k1];
        Dft<R2, INV>::run(b);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = b[k2];
    });
}

template <bool INV>
struct Dft<8, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[8]) { dft_split<2, 4, INV>(v); }
};
template <bool INV>
struct Dft<16, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[16]) { dft_split<4, 4, INV>(v); }
};
template <bool INV>
struct Dft<24, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[24]) { dft_split<3, 8, INV>(v); }
};
template <bool INV>
struct Dft<32, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[32]) { dft_split<4, 8, INV>(v); }
};

}  // namespace psx
