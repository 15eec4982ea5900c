// fft_pk.hpp -- small DFTs held in registers, written for the packed-fp32 VALU of gfx950 (v_pk_add/mul/fma_f32).
//
// One complex value = one aligned 64-bit VGPR pair (re, im).  A wave64 v_pk_* instruction issues in ~1.6x the time of a
// scalar fp32 instruction and does two lanes of work (tools/valu_bench.hip), so the butterflies are fastest when EVERY
// arithmetic instruction is packed and no v_mov is spent on pairing or swizzling.  hipcc forms packed ops from float2
// code by itself but cannot fold a swap or a one-sided negation into them (it emits v_mov/v_xor and falls back to scalar
// ops: 403 VALU instructions per radix-24 butterfly with twiddles).  The swizzled forms are therefore spelled out with
// the VOP3P modifiers op_sel / op_sel_hi / neg_lo / neg_hi:
//     a + (-/+ i) b        one v_pk_add_f32      (swap b's halves, negate one of them)
//     a * b (complex)      v_pk_mul_f32 + v_pk_fma_f32
// Compile-time twiddles sit in SGPR pairs (one constant-bus operand per instruction).  Natural order in and out; INV
// selects the conjugate transform (unnormalised).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fft_consts.hpp"

namespace psx {

typedef float v2f __attribute__((ext_vector_type(2)));

// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ v2f add_mi(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + (+i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2f add_pi(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + w b and a - w b with w = -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ v2f add_rot(v2f a, v2f b) {
    return INV ? add_pi(a, b) : add_mi(a, b);
}
template <bool INV>
__device__ __forceinline__ v2f sub_rot(v2f a, v2f b) {
    return INV ? add_mi(a, b) : add_pi(a, b);
}

// complex products; the second factor in VGPRs
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f b) {    // a * b
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));   // (a.y b.y, a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
__device__ __forceinline__ v2f pk_cmulc(v2f a, v2f b) {   // a * conj(b)
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// the same with a wave-uniform second factor in an SGPR pair
__device__ __forceinline__ v2f pk_cmul_s(v2f a, v2f b) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "s"(b), "v"(t));
    return r;
}
__device__ __forceinline__ v2f pk_cmulc_s(v2f a, v2f b) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "s"(b), "v"(t));
    return r;
}
// a * k + c with a real wave-uniform k (both halves)
__device__ __forceinline__ v2f pk_fma_k(v2f a, float k, v2f c) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "s"((v2f){k, k}), "v"(c));
    return r;
}
__device__ __forceinline__ v2f pk_mul_k(v2f a, float k) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "s"((v2f){k, k}));
    return r;
}
// (-i) a (forward) / (+i) a (inverse): swap, negate one half
template <bool INV>
__device__ __forceinline__ v2f pk_rot(v2f a) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "s"(INV ? (v2f){-1.f, 1.f} : (v2f){1.f, -1.f}));
    return r;
}

// a * w_N^t, w_N = exp(-2 pi i / N) (forward) or its conjugate (inverse); t a compile-time index
template <int N, int T, bool INV>
__device__ __forceinline__ v2f pk_twiddle(v2f a) {
    constexpr int t = ((T % N) + N) % N;
    if constexpr (t == 0) {
        return a;
    } else if constexpr (2 * t == N) {
        return -a;
    } else if constexpr (4 * t == N) {
        return pk_rot<INV>(a);
    } else if constexpr (4 * t == 3 * N) {
        return pk_rot<!INV>(a);
    } else {
        const v2f w = (v2f){TwConst<N>::c[t], TwConst<N>::s[t]};     // exp(+2 pi i t / N)
        return INV ? pk_cmul_s(a, w) : pk_cmulc_s(a, w);
    }
}

template <int R, bool INV>
struct DftPk;

template <bool INV>
struct DftPk<2, INV> {
    static __device__ __forceinline__ void run(v2f (&v)[2]) {
        const v2f a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};

template <bool INV>
struct DftPk<3, INV> {
    static __device__ __forceinline__ void run(v2f (&v)[3]) {
        const v2f s = v[1] + v[2], d = v[1] - v[2];
        const v2f m = pk_fma_k(s, -0.5f, v[0]);
        const v2f hd = pk_mul_k(d, 0.86602540378443865f);
        v[0] = v[0] + s;
        v[1] = add_rot<INV>(m, hd);      // m + (-i) h d  (forward)
        v[2] = sub_rot<INV>(m, hd);
    }
};

template <bool INV>
struct DftPk<4, INV> {
    static __device__ __forceinline__ void run(v2f (&v)[4]) {
        const v2f t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], t3 = v[1] - v[3];
        v[0] = t0 + t2;
        v[2] = t0 - t2;
        v[1] = add_rot<INV>(t1, t3);
        v[3] = sub_rot<INV>(t1, t3);
    }
};

// 8 = 2 x 4 with the twiddles w8, -i, w8^3 folded into the second layer: 27 packed instructions
template <bool INV>
struct DftPk<8, INV> {
    static __device__ __forceinline__ void run(v2f (&v)[8]) {
        constexpr float h = 0.70710678118654752f;
        // layer 1: radix 2 over n1 (stride 4)
        const v2f e0 = v[0] + v[4], e1 = v[1] + v[5], e2 = v[2] + v[6], e3 = v[3] + v[7];     // k1 = 0
        const v2f c0 = v[0] - v[4], c1 = v[1] - v[5], c2 = v[2] - v[6], c3 = v[3] - v[7];     // k1 = 1, twiddles pending
        // k1 = 0: plain radix 4 -> outputs k = 0, 2, 4, 6
        {
            const v2f t0 = e0 + e2, t1 = e0 - e2, t2 = e1 + e3, t3 = e1 - e3;
            v[0] = t0 + t2;
            v[4] = t0 - t2;
            v[2] = add_rot<INV>(t1, t3);
            v[6] = sub_rot<INV>(t1, t3);
        }
        // k1 = 1: inputs c0, w c1, w^2 c2, w^3 c3 with w = exp(-/+ i pi/4):  w x = h (x + r x),  w^3 x = -h (x - r x),  r = -/+ i
        {
            const v2f u1 = add_rot<INV>(c1, c1), u3 = sub_rot<INV>(c3, c3);   // w c1 = h u1,  w^3 c3 = -h u3
            const v2f d = u1 - u3, e = u1 + u3;                               // (w c1 + w^3 c3) = h d,  (w c1 - w^3 c3) = h e
            const v2f t0 = add_rot<INV>(c0, c2), t1 = sub_rot<INV>(c0, c2);   // c0 +/- w^2 c2
            const v2f t3 = pk_mul_k(e, h);
            v[1] = pk_fma_k(d, h, t0);
            v[5] = pk_fma_k(d, -h, t0);
            v[3] = add_rot<INV>(t1, t3);
            v[7] = sub_rot<INV>(t1, t3);
        }
    }
};

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void pk_static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        pk_static_for<B + 1, E>(f);
    }
}

// N = R1*R2:  n = R2*n1 + n2,  k = k1 + R1*k2
template <int R1, int R2, bool INV>
__device__ __forceinline__ void dft_pk_split(v2f (&v)[R1 * R2]) {
    constexpr int N = R1 * R2;
    v2f y[N];
    pk_static_for<0, R2>([&](auto n2c) __attribute__((always_inline)) {
        constexpr int n2 = decltype(n2c)::value;
        v2f a[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) a[n1] = v[R2 * n1 + n2];
        DftPk<R1, INV>::run(a);
        pk_static_for<0, R1>([&](auto k1c) __attribute__((always_inline)) {
            constexpr int k1 = decltype(k1c)::value;
            y[n2 * R1 + k1] = pk_twiddle<N, n2 * k1, INV>(a[k1]);
        });
    });
    pk_static_for<0, R1>([&](auto k1c) __attribute__((always_inline)) {
        constexpr int k1 = decltype(k1c)::value;
        v2f b[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) b[n2] = y[n2 * R1 + k1];
        DftPk<R2, INV>::run(b);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = b[k2];
    });
}

template <bool INV>
struct DftPk<16, INV> {
    static __device__ __forceinline__ void run(v2f (&v)[16]) { dft_pk_split<4, 4, INV>(v); }
};
template <bool INV>
struct DftPk<24, INV> {
    static __device__ __forceinline__ void run(v2f (&v)[24]) { dft_pk_split<3, 8, INV>(v); }
};
template <bool INV>
struct DftPk<32, INV> {
    static __device__ __forceinline__ void run(v2f (&v)[32]) { dft_pk_split<4, 8, INV>(v); }
};

}  // namespace psx
