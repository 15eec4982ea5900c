// fresnel_p2_dev.hpp -- device helpers shared by the power-of-two line kernels of the LDS Fresnel engine (fresnel_p2.hip: lines that
// fit ONE M-point transform; fresnel_p2x.hip: lines of 4 M points' worth, two coupled rounds): stage A's twiddles from binary
// powers, the stage-B butterflies on registers, the window stores of inverse stage A.
#pragma once
#include "fresnel_stages.hpp"

namespace psx {
namespace p2dev {

using namespace psx::lines;

constexpr int BSTR = 256 + 8;  // padded stride of a block of 256 points (one pad slot per 32)

// Stage A's twiddles w_M^{n q}, q < R, from the LB = log2(R) powers w^(n 2^b) of the thread's row: w^(n q) is the product of the
// powers of q's set bits, at most four multiplications deep -- 26 products for R = 32 where two factor tables cost 31 and 62 LDS
// reads (fresnel_lds.hip, twiddle_A); 5 reads here.  The lower half is applied as it is built, the upper half takes w^(16 n) first.
template <int R, bool CONJ>
__device__ __forceinline__ v2f tw_apply(v2f x, v2f w) {
    return CONJ ? pk_cmulc(x, w) : pk_cmul(x, w);
}
template <int R>
__device__ __forceinline__ void tw_powers(v2f (&pw)[5], const v2f *row) {
    pw[0] = lds_read(row);
    if constexpr (R >= 4) pw[1] = lds_read(row + 1);
    if constexpr (R >= 8) pw[2] = lds_read(row + 2);
    if constexpr (R >= 16) pw[3] = lds_read(row + 3);
    if constexpr (R == 32) pw[4] = lds_read(row + 4);
}
template <int R, bool CONJ>
__device__ __forceinline__ void twiddle_A2(v2f (&v)[R], const v2f (&pw)[5]) {
    constexpr int H = R >= 16 ? 16 : R;          // twiddles built explicitly: q < H
    v2f t[H];
    t[1] = pw[0];
    if constexpr (R >= 4) t[2] = pw[1];
    if constexpr (R >= 8) t[4] = pw[2];
    if constexpr (R >= 16) t[8] = pw[3];
    v2f t16 = (v2f){1.f, 0.f};
    if constexpr (R == 32) t16 = pw[4];
    if constexpr (R >= 4) t[3] = pk_cmul(t[1], t[2]);
    if constexpr (R >= 8) {
#pragma unroll
        for (int r = 1; r < 4; ++r) t[4 + r] = pk_cmul(t[4], t[r]);
    }
    if constexpr (R >= 16) {
#pragma unroll
        for (int r = 1; r < 8; ++r) t[8 + r] = pk_cmul(t[8], t[r]);
    }
#pragma unroll
    for (int q = 1; q < H; ++q) v[q] = tw_apply<R, CONJ>(v[q], t[q]);
    if constexpr (R == 32) {
#pragma unroll
        for (int q = 16; q < 32; ++q) v[q] = tw_apply<R, CONJ>(v[q], t16);
#pragma unroll
        for (int q = 17; q < 32; ++q) v[q] = tw_apply<R, CONJ>(v[q], t[q - 16]);
    }
}

// leg q of a stage-B butterfly / point q of a slab
__device__ __forceinline__ int offB(int q) { return 16 * q + (q >> 1); }

// forward stage B on registers: radix 16, then twiddle w_256^{n3 k2} on the outputs
__device__ __forceinline__ void fwdB_regs(v2f (&v)[16], const v2f (&w)[16]) {
    DftPk<16, false>::run(v);
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = pk_cmul(v[q], w[q]);
}
// inverse stage B: conjugate twiddle on the inputs, then the inverse butterfly
__device__ __forceinline__ void invB_regs(v2f (&v)[16], const v2f (&w)[16]) {
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = pk_cmulc(v[q], w[q]);
    DftPk<16, true>::run(v);
}

// Legs Q0 .. R-1 + the wrapped leg of one inverse stage-A butterfly through a buffer descriptor whose range is the window the
// line may touch (fresnel_stages.hpp, store_window: the hardware drops what falls outside).  Leg q goes to element e0 + q *
// estep, the wrapped leg vw (leg 0 + its fix-up) to e0 + R * estep.
// Q0 = R / 2: the caller knows that the lower half of the legs lies before sample 0 for EVERY butterfly of the line (leg q of
// butterfly n is sample n + 256 q - (P - 1); so whenever P - 1 >= M / 2, i.e. on every power-of-two grid) -- their stores, and
// the |.|^2 or global phase in front of them, are not issued at all.  (A per-leg uniform test instead of the two compiled forms
// cost pass 2 what it saved: gpurun_out/r6s9.)
template <int R, int Q0>
__device__ __forceinline__ void store_legs(const v2f (&v)[R], v2f vw, v2f *wo, float *io, int64_t wbase, int welems, int e0, int estep,
                                           v2f gp, float sc, int accumulate) {
    // byte offsets in UNSIGNED arithmetic: an element index below the window wraps to a huge offset (dropped), and the window of a
    // 16384^2 intermediate is 2^31 bytes long
    if (wo) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(wo + wbase, 0, (int)((unsigned)welems * 8u), 0x00020000);
        const bool plain = gp.x == 1.f && gp.y == 0.f;     // pass 1: no global phase
        unsigned off = ((unsigned)e0 + (unsigned)Q0 * (unsigned)estep) * 8u;
#pragma unroll
        for (int q = Q0; q <= R; ++q) {
            const v2f x = q < R ? v[q < R ? q : 0] : vw;
            const v2f r = plain ? x : pk_cmul_s(x, gp);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, r), rs, (int)off, 0, 0);
            off += (unsigned)estep * 8u;
        }
    }
    if (io) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(io + wbase, 0, (int)((unsigned)welems * 4u), 0x00020000);
        unsigned off = ((unsigned)e0 + (unsigned)Q0 * (unsigned)estep) * 4u;
#pragma unroll
        for (int q = Q0; q <= R; ++q) {
            const v2f x = q < R ? v[q < R ? q : 0] : vw;
            float I = sc * (x.x * x.x + x.y * x.y);
            if (accumulate) I += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, I), rs, (int)off, 0, 0);
            off += (unsigned)estep * 4u;
        }
    }
}

// CNT consecutive legs, the first of them leg q0, of a butterfly whose leg q goes to element e0 + q * estep of the window (the
// two-round kernel stores a butterfly in two halves, each as soon as its data is there)
template <int CNT>
__device__ __forceinline__ void store_legs_range(const v2f (&v)[CNT], int q0, v2f *wo, float *io, int64_t wbase, int welems, int e0, int estep,
                                                 v2f gp, float sc, int accumulate) {
    if (wo) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(wo + wbase, 0, (int)((unsigned)welems * 8u), 0x00020000);
        const bool plain = gp.x == 1.f && gp.y == 0.f;
        unsigned off = ((unsigned)e0 + (unsigned)q0 * (unsigned)estep) * 8u;
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            const v2f r = plain ? v[q] : pk_cmul_s(v[q], gp);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, r), rs, (int)off, 0, 0);
            off += (unsigned)estep * 8u;
        }
    }
    if (io) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(io + wbase, 0, (int)((unsigned)welems * 4u), 0x00020000);
        unsigned off = ((unsigned)e0 + (unsigned)q0 * (unsigned)estep) * 4u;
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            float I = sc * (v[q].x * v[q].x + v[q].y * v[q].y);
            if (accumulate) I += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, I), rs, (int)off, 0, 0);
            off += (unsigned)estep * 4u;
        }
    }
}

}  // namespace p2dev
}  // namespace psx
