// fresnel_stages.hpp -- what the two line kernels of the LDS Fresnel engine share (fresnel_lds.hip: k_fresnel_lines for lines
// that fit one LDS transform, k_fresnel_part for the partitioned / coupled / DIF rounds of longer lines): constants, the
// argument block, the LDS-only barriers, the geometry of a transform in LDS, the twiddle tables and the stages whose code is the
// same in every mode -- forward / inverse stage B, stage A's twiddle pass, the plain forward stage A and the plain middle
// stage.  Round 4: until then ONE kernel with seven boolean template parameters held every mode, and
// a variable introduced for one mode could spill registers in another.  The kernels keep only their own control flow,
// loaders, coupling stages and output paths.
#pragma once
#include <cstdint>
#include <type_traits>

#include "fft_pk.hpp"
#include "common.hpp"

namespace psx {
namespace lines {

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// One 16-wave workgroup per CU owns the whole LDS.  Waves 0..11 (TC threads) are the butterfly engine -- 3 per SIMD are
// needed to keep the vector pipes issuing -- and waves 12..15 (TL threads, one per SIMD) only move samples: they fetch
// the NEXT line group from HBM while the engine transforms the current one, and spread it into LDS in the shadow of the
// last butterfly + store.  Four waves per SIMD cap every wave at 128 VGPRs.  The workgroups are persistent (one per CU,
// a strided list of line groups each), so the exposed fetch latency is paid once per launch instead of once per group.
constexpr int TC = 768;       // engine threads = radix-24 butterflies per stage
constexpr int TL = 256;       // loader threads
constexpr int T = TC + TL;
constexpr int TOT = 18432;    // complex points resident in LDS per workgroup = LINES * M
constexpr int RAD = 24;       // radix of the two big stages
// The intermediate between the two passes is stored in blocks of IB samples of a pass-1 line: [Nx/IB][Ny][IB].  Pass 1
// still writes whole 128-byte lines (2 image rows x 8 samples), and the 16-byte pieces a pass-2 workgroup reads (two
// adjacent pass-2 lines) sit 64 bytes apart instead of a whole image row: half the cache lines per wave load.
#ifndef PSX_IB
#define PSX_IB 8              // build-time A/B of the block shape (tools/ab_ib.sh): 4, 8, 16
#endif
constexpr int IB = PSX_IB;
constexpr int IBS = IB == 4 ? 2 : (IB == 8 ? 3 : 4);      // log2(IB)
static_assert((1 << IBS) == IB, "intermediate block size");
// Byte stride of consecutive samples of ONE pass-1 line inside the blocked intermediate, as a shift: the ONLY place it is
// derived.  (Round 4, gpurun_out/r4s2: a loader's buffer-descriptor range was written `N << 6` -- right for IB = 8 only -- and
// the IB = 4 A/B build read twice past its intermediate until 5b1bf8b; every such stride now comes from these two constants.)
constexpr int SAMPLE_SHIFT = 3;                           // log2(sizeof(float2)): contiguous lines
constexpr int BLOCKED_SAMPLE_SHIFT = IBS + SAMPLE_SHIFT;  // log2(IB * sizeof(float2)): lines of the blocked intermediate
static_assert((1 << SAMPLE_SHIFT) == sizeof(float2) && (1 << BLOCKED_SAMPLE_SHIFT) == IB * sizeof(float2),
              "sample strides of the intermediate");
constexpr int QUEUE_WORDS = 16 * 257;     // work queues: a counter per workgroup (<= 256, 64 bytes apart) + the count of workgroups done
#ifndef PSX_DIF_KEEP
#define PSX_DIF_KEEP 48       // DIF rounds: positions (of 72 per loader thread) fetched once per line and kept in registers for its
#endif                        // later rounds; the other 24 rotate through a 12-position buffer every round (0: nothing kept)
#ifndef PSX_DIF_NHA
#define PSX_DIF_NHA 60        // DIF rounds: window positions (of 72 per loader thread) that travel during the transform; the rest is
#endif                        // fetched between barriers (3) and (4).  52 / 56 / 60: 16384^2 passes 13.17 + 10.99 / 13.02 + 11.05 / 13.00 + 10.88 ms

__host__ __device__ constexpr int phys(int p) { return p + (p >> 5); }   // one pad slot per 32: conflict-free slabs

// (distance, source) pairs one line launch can carry: the PSX_MAX_DIST distances of one source wave, or -- a batch of source
// waves, e.g. the energies of a detector bin -- up to MAX_LINE (source, distance) pairs, each with its own input and tables
constexpr int MAX_LINE = 32;

struct LineArgs {
    int n_dist;             // distances merged into this launch: work item w = d * ngroups + g  (d-th table / buffers, group g)
    int dist_inner;         // 1: every distance reads the SAME source (pass 1): a workgroup takes the n_dist work items of a
                            // line group in consecutive rounds and its loaders fetch the group once, spreading it n_dist times
    const float2 *src[MAX_LINE];        // input wave of each distance
    int N, nlines, margin, P, L;
    int64_t in_si, in_sl;   // sample i of line l is element i*in_si + l*in_sl of src ...
    int in_blocked;         // ... or, blocked: element ((l / IB)*N + i)*IB + l % IB  (the intermediate, see IB)
    int64_t out_ld;         // output sample i of line l goes to l*out_ld + i ...
    int out_blocked;        // ... or, blocked: element ((i / IB)*nlines + l)*IB + i % IB
    const float2 *twA, *twB;   // [n][24] stage twiddles
    const float2 *H[MAX_LINE];          // kernel spectrum FFT_M(h) of each distance, digit-reversed, 1/M folded in
    float2 *wave_out[MAX_LINE];         // complex result (pass 1: the blocked intermediate) or null
    float *inten_out[MAX_LINE];         // scale * |result|^2 or null
    float scale[MAX_LINE];
    float2 gph[MAX_LINE];               // global phase factor exp(i k z / M) of the complex result
    int accumulate;
    // Partitioned convolution (PART instantiations; lines too long for one M-point transform in LDS): the N outputs of a
    // line are cut into NB blocks of B, the P-tap kernel into S segments of Lh (B + Lh - 1 <= M); a work unit is
    // (distance, line group, block) and takes S consecutive rounds, one per segment, whose results add up in `part`
    // (complex, same layout as the complex output; it IS the complex output when that is wanted).  H[d] then holds S spectra.
    int B, Lh, S, NB;
    float2 *part[MAX_LINE];
    const float2 *w2;       // PAIR: w_2M^{k0} of each 16-point slab, k0 = q1 + 24 q2  (576 entries)
    // DIF (see k_fresnel_lines): one 4M-point convolution per line in two PAIR rounds
    const float2 *w4;       // exp(+2 pi i n0 / 4M), n0 < 2*S1 = 768: the thread-dependent factor of the radix-2 twiddle w_4M^{-n}
    float2 *wgpart;         // [workgroups][2M]: the even half-spectrum's result of a line, private to the workgroup, between its two rounds
    int wg_groups;          // line buffers in wgpart (host-side check against the grid)
    int dsh, thr;           // D = 2M - P: L[n + D] = e[n + 2M] (the extension is P-periodic); thr = N + P - 1 - 2M: positions that have one
    unsigned *queue;        // work queues of the one-transform passes (null: static shares): the counter of workgroup w at [16 w], workgroups done at [16 * 256]
    unsigned long long *stamps;   // optional diagnostics: 32 phase timestamps per workgroup (psx_debug_stamps)
    int stamp_j;                  // ... of this round of every workgroup (psx_debug_switch "stamp_round", default 1: a steady-state round)
};

// phase timestamp k of round a.stamp_j of each workgroup, taken by the thread for which `who` holds (diagnostic runs only)
#define PSX_STAMP_IF(k, who)                                                                       \
    do {                                                                                           \
        if (a.stamps && j == a.stamp_j && (who)) a.stamps[(size_t)blockIdx.x * 32 + (k)] = wall_clock64(); \
    } while (0)

// orders the LDS traffic of ONE wave (cross-lane exchange through LDS without a workgroup barrier): no instruction is
// emitted beyond the wait the fence implies; the compiler may not move LDS accesses across it
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One ds_read_b64 per value.  Left alone, the compiler pairs neighbouring reads into ds_read2_b64, which moves the same
// 16 bytes per lane in 8 LDS cycles instead of 2 x 2 (MI355X_MICROARCH.md, LDS table); a volatile access is not paired.
__device__ __forceinline__ v2f lds_read(const v2f *p) {
    typedef const volatile __attribute__((address_space(3))) v2f *lds_ptr;   // explicit: a volatile generic load is a flat load
    return *(lds_ptr)p;
}

// workgroup barrier that orders LDS traffic only: a loader wave passes it with its global loads still in flight
// (__syncthreads() would wait vmcnt(0) and stall the engine behind an HBM round trip)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}


// ---- geometry of one transform of M = 576 R3 points in LDS (one pad slot per 32 points; PAIRPAD: the second line starts 16
// points further, so that a point of line 0 and the same point of line 1 sit 32 banks apart)
template <int R3_, bool PAIRPAD>
struct LineGeom {
    static constexpr int R3 = R3_, M = 576 * R3, LINES = TOT / M, S1 = M / RAD, MP = M + M / 32 + (PAIRPAD ? 16 : 0);
    static constexpr int SLAB = 16, NSLABS = TOT / SLAB;       // 16 contiguous points per slab in the middle stage
    static constexpr int WSLABS = 64 * RAD / SLAB;              // slabs inside the 1536 points one wave owns between barriers
    static constexpr int NSLAB = (WSLABS + 63) / 64;            // slab rounds per lane (the last one is partly idle)
    static constexpr int TWB_LD = RAD + 1;                      // twiddle-table rows padded to 25: 16 rows on 16 bank pairs
    static_assert(R3 <= 16 && SLAB % R3 == 0 && TOT == RAD * TC && 64 % R3 == 0 && NSLABS == (TC / 64) * WSLABS, "unsupported geometry");
    // LDS index of butterfly element j: with S1 a multiple of 32 (and R3 | 32) the pad term of phys() is affine in j, so
    // every element is one base register + a compile-time offset (ds_read/ds_write immediate offsets)
    static constexpr bool AFF = (S1 % 32 == 0);
    static __device__ __forceinline__ int idxA(int n, int j) { return AFF ? phys(n) + j * (S1 + S1 / 32) : phys(n + j * S1); }
    static __device__ __forceinline__ int idxB(int p0, int j) {      // p0 = q1*S1 + n,  n < R3
        return AFF ? phys(p0) + j * R3 + ((j * R3) >> 5) : phys(p0 + j * R3);
    }
    static constexpr size_t lds_bytes = sizeof(float2) * ((size_t)LINES * MP + (2 * R3 + RAD) * (RAD + 1)) + 16;   // lines + tables + unit ring
};

// Stage B's twiddles w_S1^{n q} (n < R3, q < 24: 3 KiB) live in LDS behind the line buffers, rows padded to 25 so that
// the 16 rows start on 16 different bank pairs: a ds_read_b64 costs 2 LDS cycles where the 16-byte global loads of
// the same table (L1 hits) took the CU's 64 B/clk vector-memory return path that the loaders and the other tables need.
// Stage A's twiddles w_M^{n q}, n < S1 = 24 R3, would be 72 KiB; with n = R3 n1 + n0 they factor into
// w_576^{n1 q} * w_M^{n0 q}: a [24][24] and an [R3][24] table (rows R3*n1 and n0 of the global table), 7.8 KiB, at the
// price of one more complex multiply per point and stage (+46 packed instructions per butterfly).  With all three
// tables in LDS the engine's only global loads are the kernel spectrum's.
struct TwTables {
    v2f *twl;    // [R3][25]  stage B
    v2f *tw1;    // [24][25]  stage A, n1 part
    v2f *tw0;    // [R3][25]  stage A, n0 part
};
template <class GE>
__device__ __forceinline__ TwTables fill_tables(float2 *lds, const LineArgs &a, int tid) {
    constexpr int R3 = GE::R3, TWB_LD = GE::TWB_LD;
    TwTables t;
    t.twl = reinterpret_cast<v2f *>(lds) + GE::LINES * GE::MP;
    t.tw1 = t.twl + R3 * TWB_LD;
    t.tw0 = t.tw1 + RAD * TWB_LD;
    for (int idx = tid; idx < R3 * RAD; idx += T) {
        const float2 w = a.twB[idx], w0 = a.twA[idx];
        t.twl[(idx / RAD) * TWB_LD + idx % RAD] = (v2f){w.x, w.y};
        t.tw0[(idx / RAD) * TWB_LD + idx % RAD] = (v2f){w0.x, w0.y};
    }
    for (int idx = tid; idx < RAD * RAD; idx += T) {
        const float2 w = a.twA[(size_t)(idx / RAD) * R3 * RAD + idx % RAD];     // row n = R3 * n1
        t.tw1[(idx / RAD) * TWB_LD + idx % RAD] = (v2f){w.x, w.y};
    }
    return t;
}

// v[q] *= (or conj-*=) row1[q] * row0[q] for q in [Q0, Q1): half of the 23 twiddles at a time (128-VGPR budget)
template <int Q0, int Q1, bool CONJ>
__device__ __forceinline__ void twiddle_A(v2f (&v)[RAD], const v2f *rowA1, const v2f *rowA0) {
    v2f w1[Q1 - Q0], w0[Q1 - Q0];
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
        w1[q - Q0] = lds_read(rowA1 + q);
        w0[q - Q0] = lds_read(rowA0 + q);
    }
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
        const v2f w = pk_cmul(w1[q - Q0], w0[q - Q0]);
        v[q] = CONJ ? pk_cmulc(v[q], w) : pk_cmul(v[q], w);
    }
}

// forward stage A of a line that needs no partner arithmetic: radix 24 over stride S1, twiddle w_M^{n q}
template <class GE>
__device__ __forceinline__ void fwd_stage_A(v2f *baseA, int nA, const v2f *rowA1, const v2f *rowA0) {
    v2f v[RAD];
#pragma unroll
    for (int q = 0; q < RAD; ++q) v[q] = baseA[GE::idxA(nA, q)];
    DftPk<RAD, false>::run(v);
    __builtin_amdgcn_sched_barrier(0);
    twiddle_A<1, 12, false>(v, rowA1, rowA0);
    twiddle_A<12, 24, false>(v, rowA1, rowA0);
#pragma unroll
    for (int q = 0; q < RAD; ++q) baseA[GE::idxA(nA, q)] = v[q];
}

// forward stage B: radix 24 inside each block of S1, stride R3, twiddle w_S1^{n q} on the outputs (in two halves: the
// kernel spectrum of the first slab, 32 registers, is in flight here)
template <class GE>
__device__ __forceinline__ void fwd_stage_B(v2f *bB, int pB, const v2f *rowB) {
    v2f v[RAD];
#pragma unroll
    for (int q = 0; q < RAD; ++q) v[q] = lds_read(bB + GE::idxB(pB, q));
    DftPk<RAD, false>::run(v);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        v2f w[RAD / 2];
#pragma unroll
        for (int q = (h == 0 ? 1 : 0); q < RAD / 2; ++q) w[q] = lds_read(rowB + h * (RAD / 2) + q);
#pragma unroll
        for (int q = (h == 0 ? 1 : 0); q < RAD / 2; ++q) v[h * (RAD / 2) + q] = pk_cmul(v[h * (RAD / 2) + q], w[q]);
    }
#pragma unroll
    for (int q = 0; q < RAD; ++q) bB[GE::idxB(pB, q)] = v[q];
}

// inverse stage B: conjugate twiddle on the inputs, then the inverse radix-24 butterfly
template <class GE>
__device__ __forceinline__ void inv_stage_B(v2f *bB, int pB, const v2f *rowB) {
    v2f v[RAD], w[RAD];
#pragma unroll
    for (int q = 0; q < RAD; ++q) v[q] = lds_read(bB + GE::idxB(pB, q));
#pragma unroll
    for (int q = 1; q < RAD; ++q) w[q] = lds_read(rowB + q);
#pragma unroll
    for (int q = 1; q < RAD; ++q) v[q] = pk_cmulc(v[q], w[q]);
    __builtin_amdgcn_sched_barrier(0);
    DftPk<RAD, true>::run(v);
#pragma unroll
    for (int q = 0; q < RAD; ++q) bB[GE::idxB(pB, q)] = v[q];
}

// The middle stage of the uncoupled modes, slab by slab: forward radix R3 on contiguous chunks, x FFT_M(h_d), inverse radix
// R3, back to LDS.  Each thread rewrites exactly the slabs it read; hh holds the first slab's spectrum on entry (requested
// before barrier (1)), the next slab's is requested under the current one's inverse DFT.
template <class GE>
__device__ __forceinline__ void middle_plain(float2 *lds, const float2 *Hd, int slabw, int slab0, float4 (&hh)[GE::SLAB / 2]) {
    constexpr int R3 = GE::R3, M = GE::M, MP = GE::MP, SLAB = GE::SLAB, WSLABS = GE::WSLABS, NSLAB = GE::NSLAB;
#pragma unroll
    for (int r = 0; r < NSLAB; ++r) {
        if (slabw + 64 * r >= WSLABS) break;
        const int s = slab0 + 64 * r, line = s / (M / SLAB), p0 = (s % (M / SLAB)) * SLAB;
        v2f *base = reinterpret_cast<v2f *>(lds) + line * MP + phys(p0);   // p0 % 16 == 0: no pad slot inside a slab
        v2f f[SLAB];
#pragma unroll
        for (int q = 0; q < SLAB; ++q) f[q] = lds_read(base + q);
#pragma unroll
        for (int c = 0; c < SLAB / R3; ++c) {
            v2f w[R3];
#pragma unroll
            for (int q = 0; q < R3; ++q) w[q] = f[c * R3 + q];
            DftPk<R3, false>::run(w);
#pragma unroll
            for (int q = 0; q < R3; ++q) f[c * R3 + q] = w[q];
        }
#pragma unroll
        for (int q = 0; q < SLAB / 2; ++q) {
            f[2 * q] = pk_cmul(f[2 * q], (v2f){hh[q].x, hh[q].y});
            f[2 * q + 1] = pk_cmul(f[2 * q + 1], (v2f){hh[q].z, hh[q].w});
        }
        if (r + 1 < NSLAB && slabw + 64 * (r + 1) < WSLABS) {      // next slab's spectrum, under this one's inverse DFT
            __builtin_amdgcn_sched_barrier(0);
            const float4 *h4 = reinterpret_cast<const float4 *>(Hd + ((slab0 + 64 * (r + 1)) % (M / SLAB)) * SLAB);
#pragma unroll
            for (int q = 0; q < SLAB / 2; ++q) hh[q] = h4[q];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < SLAB / R3; ++c) {
            v2f w[R3];
#pragma unroll
            for (int q = 0; q < R3; ++q) w[q] = f[c * R3 + q];
            DftPk<R3, true>::run(w);
#pragma unroll
            for (int q = 0; q < R3; ++q) base[c * R3 + q] = w[q];
        }
    }
}

// The wanted outputs of 24 legs through a buffer descriptor whose range is exactly the window they may touch: the hardware
// drops the stores of the unwanted outputs (an index below the window wraps to a huge offset, one above lies past it).  No
// compare, no exec-mask bookkeeping per output -- the scalar unit is shared by the whole CU (0.9 instructions per cycle,
// tools/salu_bench.hip) and the masked form of this loop issued 500 scalar instructions per wave.
// Leg q goes to element e0 + q * estep of the window [wbase, wbase + welems) of the complex result wo (x the global phase gp) and /
// or of the intensity image io (sc |.|^2, stored or added).
__device__ __forceinline__ void store_window(const v2f (&v)[RAD], v2f *wo, float *io, int64_t wbase, int welems, int e0, int estep,
                                             v2f gp, float sc, int accumulate) {
    if (wo) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(wo + wbase, 0, welems * 8, 0x00020000);
        int off = e0 * 8;
        if (gp.x == 1.f && gp.y == 0.f) {     // pass 1 (and z-independent callers): no global phase to apply
#pragma unroll
            for (int q = 0; q < RAD; ++q) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v[q]), rs, off, 0, 0);
                off += estep * 8;
            }
        } else {
#pragma unroll
            for (int q = 0; q < RAD; ++q) {
                const v2f r = pk_cmul_s(v[q], gp);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, r), rs, off, 0, 0);
                off += estep * 8;
            }
        }
    }
    if (io) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(io + wbase, 0, welems * 4, 0x00020000);
        int off = e0 * 4;
        if (accumulate) {
#pragma unroll
            for (int q = 0; q < RAD; ++q) {
                const float I = sc * (v[q].x * v[q].x + v[q].y * v[q].y);
                const float old = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, old + I), rs, off, 0, 0);
                off += estep * 4;
            }
        } else {
#pragma unroll
            for (int q = 0; q < RAD; ++q) {
                const float I = sc * (v[q].x * v[q].x + v[q].y * v[q].y);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, I), rs, off, 0, 0);
                off += estep * 4;
            }
        }
    }
}

}  // namespace lines
}  // namespace psx
