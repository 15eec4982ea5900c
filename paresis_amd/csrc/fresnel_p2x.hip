// fresnel_p2x.hip -- power-of-two line kernel of the LDS Fresnel engine for lines of about 16384 samples (BASELINE config 5:
// the 16384^2 grid; EXP:219-252 of the reference).
//
// fresnel_p2.hip convolves a line of N <= 4096 samples through ONE M = 8192-point circular transform that lives in one LDS line,
// with the Lx = N + P - 1 - M outputs whose window wraps put right from the taps they missed.  A line of N ~ 16384 samples
// (P = N + 2 * margin taps, extension of N + P - 1 = 32797 points) takes ONE circular convolution of R = 4 M = 32768 points the
// same way -- 11 % fewer points than the 36864 of fresnel_lds.hip's k_fresnel_part, and no second copy of the line in LDS:
//   * the R-point transform is split by a radix-2 decimation-in-frequency step over TWO rounds (E: even bins, O: odd bins),
//         ye = IDFT_Q( FFT_Q( x[n] + x[n + Q] )           * H_R[2k]   ),      Q = 2 M = 16384,
//         yo = IDFT_Q( FFT_Q((x[n] - x[n + Q]) w_R^n )    * H_R[2k+1] ),      y[m] = ye[m mod Q] + w_R^-m yo[m mod Q]   (1/2 in H);
//   * each round's Q-point transform lives in BOTH LDS lines -- even samples in line 0, odd samples in line 1 -- coupled by the
//     radix-2 butterfly of the double-size transform in the middle stage, where a wave holds the same slab of both lines anyway
//     (fresnel_p2.hip: the wave-private range is 1024 points of two lines);
//   * the extension e is P-periodic and P - Q = 2 * margin is small: x[n + Q] = e[n + Q] is e[n - (P - Q)], i.e. the SAME LDS
//     line, margin points earlier (the first margin points find theirs in the slack behind the line).  The loaders write one copy,
//     e[t] for t < P, and forward stage A forms the sum / difference from LDS (one more barrier: every input and partner is read
//     before anything is written in place);
//   * a line is fetched ONCE for its two rounds -- in pass 1 once for the rounds of ALL distances -- and re-spread from the
//     loaders' registers; round E parks ye in a buffer of the workgroup's own (read back by the same thread one round later),
//     round O recombines and stores;
//   * the Lx = N + P - 1 - R wrapped outputs (29) are put right as in fresnel_p2.hip.
// Everything else -- the three in-place stages, the thread budget (8 engine waves x 32 points, 4 loader waves), the barriers, the
// digit order of a transformed line -- is fresnel_p2.hip's at R1 = 32.  tools/p2_model.py holds the same arithmetic in numpy
// (engine_line_dif: 6e-15 against the reference operator).
#include <atomic>

#include "fresnel_p2.hpp"
#include "fresnel_p2_dev.hpp"

using namespace psx;
using namespace psx::lines;
using namespace psx::p2dev;

namespace {

constexpr int TE = 512, TLD = 256, TT = TE + TLD;
constexpr int LXM = psx::p2::LXMAX;
constexpr int R1 = 32, M = 256 * R1, Q = 2 * M, RR = 4 * M;
constexpr int MP = M + M / 32 + LXM;         // a line buffer: M points, pad slots, and the slack that holds positions >= Q of the extension
constexpr int LDP = 5, LDB = 17;
constexpr int NCH = 4, CL = LXM / NCH;       // fix-up: 4 chunks of 8 taps per wrapped output (128 loader threads)
// LDS map, in float2 elements
constexpr int O_TP = 2 * MP, O_TB = O_TP + 256 * LDP, O_SA = O_TB + 16 * LDB, O_SB = O_SA + LXM, O_C = O_SB + LXM, O_HT = O_C + NCH * LXM;
constexpr size_t lds_bytes(int ntap) { return sizeof(float2) * (size_t)(O_HT + ntap * LXM); }
constexpr int MAX_TAPS = (160 * 1024 / (int)sizeof(float2) - O_HT) / LXM < MAX_LINE ? (160 * 1024 / (int)sizeof(float2) - O_HT) / LXM : MAX_LINE;
static_assert(MAX_TAPS >= PSX_MAX_DIST, "LDS budget");

#define PSX_STAMP(k) PSX_STAMP_IF(k, tid == 0)

// LDS position (line, index) of position t of the extension: even positions in line 0, odd ones in line 1
__device__ __forceinline__ int pos_lds(int t) { return (t & 1) * MP + phys(t >> 1); }

template <bool CONTIG>
__global__ __launch_bounds__(TT) void k_fresnel_p2x(LineArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    v2f *const Lb = reinterpret_cast<v2f *>(lds);
    v2f *const tP = Lb + O_TP, *const tB = Lb + O_TB, *const SA = Lb + O_SA, *const SB = Lb + O_SB, *const CF = Lb + O_C, *const HT = Lb + O_HT;
    const int tid = threadIdx.x;
    const int N = a.N, mg = a.margin, P = a.P;
    const int Lx = a.L - RR;               // outputs whose window wraps (<= 0: none)
    const int DH = (P - Q) >> 1;           // the partner x[n + Q] of LDS point (line, i) is point (line, i - DH)
    const int sh = P - 1 - Q;              // point m' of a round's result is output sample m' - sh (and m' + Q - sh when it wraps)

    // ---- work units: one image line (pass 1, dist_inner: all its distances, 2 n_dist rounds) or one (distance, image line) pair
    // (2 rounds); XCD-contiguous chunks, static shares
    const int nwork = a.dist_inner ? a.nlines : a.nlines * a.n_dist;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int cq = nwork >> 3, cr = nwork & 7;
    const int cstart = xcd * cq + (xcd < cr ? xcd : cr), clen = cq + (xcd < cr ? 1 : 0);
    const int nunits = slot < clen ? (clen - slot + nslot - 1) / nslot : 0;
    const int nsubr = 2 * (a.dist_inner ? a.n_dist : 1);
    const int nj = nunits * nsubr;
    // round j -> distance d, image line l, round of the pair ps (0: even bins, 1: odd bins), first round of a unit
    auto item = [&](int j, int &d, int &l, int &ps, bool &first) __attribute__((always_inline)) {
        const int u = j / nsubr, r = j - u * nsubr;
        const int wk = cstart + slot + u * nslot;
        if (a.dist_inner) {
            d = r >> 1;
            l = wk;
        } else {
            d = wk / a.nlines;
            l = wk - d * a.nlines;
        }
        ps = r & 1;
        first = r == 0;
    };

    // ---- stage twiddles and the first taps of every distance into LDS
    for (int idx = tid; idx < 256 * 5; idx += TT) {
        const float2 wv = a.twA[idx];
        tP[idx] = (v2f){wv.x, wv.y};
    }
    for (int idx = tid; idx < 256; idx += TT) {
        const float2 w = a.twB[idx];
        tB[(idx >> 4) * LDB + (idx & 15)] = (v2f){w.x, w.y};
    }
    for (int idx = tid; idx < a.n_dist * LXM; idx += TT) {
        const float2 h = a.H[idx / LXM][4 * M + idx % LXM];
        HT[idx] = (v2f){h.x, h.y};
    }

    if (tid >= TE) {
        // =============================== loader waves =====================================================================
        // Thread lt moves the samples 2j and 2j + 1, j = lt + 256 k: sample i sits at position i - 1 of the extension, so 2j + 1
        // goes to LDS line 0, point j, and 2j to line 1, point j - 1 (sample 0: position P - 1) -- consecutive lanes write
        // consecutive points of one line.  One period of e in all: [x1 .. x(N-1) | right mirror | left mirror | x0].
        const int lt = tid - TE;
        constexpr int NLV = M / TLD;                       // 32 sample pairs per thread
        constexpr int PST = TLD + TLD / 32;                // padded stride of 256 points
        float2 *const p0 = lds + lt + (lt >> 5);                         // line 0, point lt
        float2 *const p1 = lds + MP + (lt - 1) + ((lt - 1) >> 5);        // line 1, point lt - 1 (lt = 0: k >= 1 only)
        // mirror duty of the first 2 mg threads (np.pad 'reflect', EXP:237): sample im goes to position jm as well -- left mirror |
        // right mirror.  Re-derived from an opaque copy of the thread index where it is used (128 of the loaders' registers hold the line).
        auto mirror_of = [&](int &im, int &jm) __attribute__((always_inline)) {
            int t = lt;
            asm volatile("" : "+v"(t));
            im = t < 2 * mg ? (t < mg ? t + 1 : N - 1 - 2 * mg + t) : -1;
            jm = t < mg ? P - 1 - im : 2 * N - 3 - im;
        };
        const int rp = RR - P;                                           // SB[t - rp] = e[t], rp <= t < rp + LXM
        float4 xv[NLV];
        float2 xm = make_float2(0.f, 0.f);
        auto fetch = [&](int j) __attribute__((always_inline)) {
            int d, l, ps;
            bool first;
            item(j, d, l, ps, first);
            if (!first) return;                            // the line is in the registers already
            const float2 *src = a.src[d];
            const bool lok = l < a.nlines;
            int im, jm;
            mirror_of(im, jm);
            if constexpr (CONTIG) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float2 *>(src) + (int64_t)l * a.in_sl, 0, lok ? N * (int)sizeof(float2) : 0, 0x00020000);
#pragma unroll
                for (int k = 0; k < NLV; ++k)
                    xv[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (lt + TLD * k) * 16, 0, 0));
                xm = src[(im >= 0 && lok) ? (int64_t)l * a.in_sl + im : (int64_t)0];
            } else {
                // the blocked intermediate: sample i of line l is element ((l / IB) N + i) IB + l % IB -- one 8-byte piece per sample
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float2 *>(src) + ((int64_t)(l / IB) * N) * IB + l % IB, 0,
                    lok ? N * IB * (int)sizeof(float2) - (l % IB) * (int)sizeof(float2) : 0, 0x00020000);
#pragma unroll
                for (int k = 0; k < NLV; ++k) {
                    const int o = (lt + TLD * k) * (2 * IB * (int)sizeof(float2));
                    const float2 e0 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, o, 0, 0));
                    const float2 e1 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, o + IB * (int)sizeof(float2), 0, 0));
                    xv[k] = make_float4(e0.x, e0.y, e1.x, e1.y);
                }
                xm = src[(im >= 0 && lok) ? ((int64_t)(l / IB) * N + im) * IB + l % IB : (int64_t)0];
            }
        };
        auto spread = [&](int j) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < NLV; ++k) {
                const int jj = lt + TLD * k;
                const float2 xe = make_float2(xv[k].x, xv[k].y), xo = make_float2(xv[k].z, xv[k].w);     // samples 2 jj, 2 jj + 1
                if (2 * jj + 1 < N) p0[k * PST] = xo;                                      // position 2 jj
                if (2 * jj < N) {
                    if (k > 0 || lt > 0) p1[k * PST] = xe;                                 // position 2 jj - 1
                    else lds[pos_lds(P - 1)] = xe;                                         // sample 0: position P - 1
                }
                if (k == 0) {                                                              // e[t], t < LXM: saved for the fix-up
                    if (2 * jj < LXM && 2 * jj + 1 < N) SA[2 * jj] = (v2f){xo.x, xo.y};
                    if (jj >= 1 && 2 * jj - 1 < LXM && 2 * jj < N) SA[2 * jj - 1] = (v2f){xe.x, xe.y};
                }
                if (k == NLV - 1) {                                                        // e[t], rp <= t < rp + LXM
                    if ((unsigned)(2 * jj - rp) < (unsigned)LXM && 2 * jj + 1 < N) SB[2 * jj - rp] = (v2f){xo.x, xo.y};
                    if ((unsigned)(2 * jj - 1 - rp) < (unsigned)LXM && 2 * jj < N) SB[2 * jj - 1 - rp] = (v2f){xe.x, xe.y};
                }
            }
            int im, jm;
            mirror_of(im, jm);
            if (im >= 0) {
                lds[pos_lds(jm)] = xm;
                if ((unsigned)(jm - rp) < (unsigned)LXM) SB[jm - rp] = (v2f){xm.x, xm.y};
                if (jm < LXM) SA[jm] = (v2f){xm.x, xm.y};
            }
        };
        // fix-up of the wrapped outputs (round O): CF[c][m'] = sum over the taps t of chunk c, t <= m', of h_d[t] * (e[R + m' - t] - e[m' - t])
        auto fixup = [&](int j) __attribute__((always_inline)) {
            int d, l, ps;
            bool first;
            item(j, d, l, ps, first);
            if (Lx <= 0 || ps == 0 || lt >= NCH * LXM) return;
            const int c = lt / LXM, mp = lt % LXM;
            const v2f *ht = HT + d * LXM;
            v2f acc = (v2f){0.f, 0.f};
#pragma unroll
            for (int tt = 0; tt < CL; ++tt) {
                const int t = c * CL + tt;
                const int jx = mp >= t ? mp - t : 0;
                const v2f dl = lds_read(SB + jx) - lds_read(SA + jx);
                const v2f term = pk_cmul(dl, lds_read(ht + t));
                acc += mp >= t ? term : (v2f){0.f, 0.f};
            }
            CF[c * LXM + mp] = acc;
        };

        if (nj > 0) {
            fetch(0);
            spread(0);
        }
        lds_barrier();                                   // (0)
        for (int j = 0; j < nj; ++j) {
            const bool more = j + 1 < nj;
            lds_barrier();                               // (1a) engine: stage A has read its inputs and their partners
            lds_barrier();                               // (1)
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 16] = wall_clock64();
            if (more) fetch(j + 1);
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 17] = wall_clock64();
            fixup(j);
            lds_barrier();                               // (2)
            lds_barrier();                               // (3)
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 18] = wall_clock64();
            __builtin_amdgcn_s_setprio(3);
            if (more) spread(j + 1);
            __builtin_amdgcn_s_setprio(0);
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 19] = wall_clock64();
            lds_barrier();                               // (4)
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 20] = wall_clock64();
        }
        return;
    }

    // =================================== engine waves =========================================================================
    // Every per-thread index below is re-derived where it is used from an opaque copy of the thread index, not kept across the
    // round loop (the engine waves have no register to spare: kept, they spilled 8):
    auto ftid = [&]() __attribute__((always_inline)) {
        int t = tid;
        asm volatile("" : "+v"(t));
        return t;
    };
    // stage A: the lower half-wave takes butterflies of line 0 (even positions), the upper one the same butterflies of line 1: a
    // wave's outputs are then 64 consecutive samples of the image line, and each half reads / writes consecutive LDS points
    auto stageA_at = [&](int &lineA, int &nA) __attribute__((always_inline)) {
        const int t = ftid();
        lineA = (t >> 5) & 1;
        nA = 32 * (t >> 6) + (t & 31);
    };
    // stage B and middle stage: the wave's 4 blocks (64 slabs) of BOTH lines, as fresnel_p2.hip
    auto stageB_at = [&](v2f *&pb0, const v2f *&rowB) __attribute__((always_inline)) {
        const int t = ftid(), ln = t & 63;
        const int n3 = ln & 15, blk = ((ln >> 4) & 1) * 2 + (ln >> 5);
        pb0 = Lb + (4 * (t >> 6) + blk) * BSTR + n3;
        rowB = tB + n3 * LDB;
    };
    auto slab_of = [&]() __attribute__((always_inline)) {
        const int t = ftid(), ln = t & 63;
        return 64 * (t >> 6) + (ln & 32) + ((ln & 31) < 16 ? 2 * (ln & 31) : 2 * ((ln & 31) - 16) + 1);
    };
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 32 + 0] = wall_clock64();
    lds_barrier();                                       // (0)
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 32 + 1] = wall_clock64();
    for (int j = 0; j < nj; ++j) {
        int d, l, ps;
        bool first;
        item(j, d, l, ps, first);
        PSX_STAMP(2);

        // ---- forward stage A: x[n] +- x[n + Q] (the partner: margin points earlier in the same line), round O x w_R^n; radix 32
        {
            v2f v[R1], pw[5];
            int lineA, nA;
            stageA_at(lineA, nA);
            v2f *const pA = Lb + lineA * MP + nA + (nA >> 5);
            const v2f *pP = Lb + lineA * MP + (nA - DH) + ((nA - DH) >> 5);      // arithmetic shift: affine in the leg for q >= 1
            const v2f *pP0 = nA >= DH ? pP : Lb + lineA * MP + M + M / 32 + nA;   // leg 0 of the first DH butterflies: the slack
#pragma unroll
            for (int q = 0; q < R1; ++q) v[q] = pA[q * BSTR];
            const float sg = ps == 0 ? 1.f : -1.f;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                v2f b[R1 / 2];
#pragma unroll
                for (int q = 0; q < R1 / 2; ++q) {
                    const int qq = hf * (R1 / 2) + q;
                    b[q] = lds_read(qq == 0 ? pP0 : pP + qq * BSTR);
                }
#pragma unroll
                for (int q = 0; q < R1 / 2; ++q) v[hf * (R1 / 2) + q] = pk_fma_k(b[q], sg, v[hf * (R1 / 2) + q]);
            }
            lds_barrier();                           // (1a) every input and partner has been read: the in-place writes may start
            if (ps != 0) {
                const float2 wdf = a.w4[2 * nA + lineA];                      // w_R^(+n0): leg q is position n0 + 512 q of the round's sequence
                const v2f wd = (v2f){wdf.x, wdf.y};
                pk_static_for<0, R1>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = decltype(qc)::value;
                    v[q] = pk_cmulc(v[q], pk_twiddle<64, q, true>(wd));       // x conj(w_R^(+n0) exp(+2 pi i q / 64)) = w_R^n
                });
            }
            DftPk<R1, false>::run(v);
            __builtin_amdgcn_sched_barrier(0);
            tw_powers<R1>(pw, tP + nA * LDP);
            twiddle_A2<R1, false>(v, pw);
#pragma unroll
            for (int q = 0; q < R1; ++q) pA[q * BSTR] = v[q];
        }
        // the first half of the slab's kernel-spectrum pairs travels ahead of its use
        const int S0 = slab_of();
        const float4 *h4 = reinterpret_cast<const float4 *>(a.H[d] + (size_t)ps * 2 * M) + 16 * S0;
        float4 hh[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) hh[q] = h4[q];
        const float2 wsf = a.w2[S0];                     // w_Q^k0 of this lane's slab
        const v2f wsl = (v2f){wsf.x, wsf.y};
        PSX_STAMP(3);
        lds_barrier();                               // (1)
        PSX_STAMP(4);

        // ---- forward stage B
        {
            v2f *pb0;
            const v2f *rowB;
            stageB_at(pb0, rowB);
            v2f *const pb1 = pb0 + MP;
            v2f wt[16];
#pragma unroll
            for (int q = 1; q < 16; ++q) wt[q] = lds_read(rowB + q);
            wt[0] = (v2f){1.f, 0.f};
            v2f v0[16], v1[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v0[q] = lds_read(pb0 + offB(q));
#pragma unroll
            for (int q = 0; q < 16; ++q) v1[q] = lds_read(pb1 + offB(q));
            fwdB_regs(v0, wt);
#pragma unroll
            for (int q = 0; q < 16; ++q) pb0[offB(q)] = v0[q];
            fwdB_regs(v1, wt);
#pragma unroll
            for (int q = 0; q < 16; ++q) pb1[offB(q)] = v1[q];
        }
        PSX_STAMP(5);
        wave_sync();
        PSX_STAMP(6);

        // ---- middle stage: the slab of line 0 holds E[k], that of line 1 O[k] of the Q-point transform: X[k] = E + w_Q^k O,
        // X[k + M] = E - w_Q^k O, each x its bin of the kernel spectrum, then the inverse step E' = Y[k] + Y[k + M], O' = (Y[k] - Y[k + M]) w_Q^-k
        {
            const int S0m = slab_of();
            v2f *const ps0 = Lb + 16 * S0m + (S0m >> 1), *const ps1 = ps0 + MP;
            v2f x[16], y[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = lds_read(ps0 + q);
#pragma unroll
            for (int q = 0; q < 16; ++q) y[q] = lds_read(ps1 + q);
            DftPk<16, false>::run(x);
            DftPk<16, false>::run(y);
            pk_static_for<0, 2>([&](auto cc) __attribute__((always_inline)) {
                constexpr int c = decltype(cc)::value;
                pk_static_for<0, 8>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = 8 * c + decltype(qc)::value, hq = decltype(qc)::value;
                    const v2f wq = pk_twiddle<32, q, false>(wsl);            // w_Q^k = w_Q^k0 exp(-2 pi i q / 32)
                    const v2f t = pk_cmul(y[q], wq);
                    const v2f s0 = x[q] + t, s1 = x[q] - t;
                    const v2f y0 = pk_cmul(s0, (v2f){hh[hq].x, hh[hq].y}), y1 = pk_cmul(s1, (v2f){hh[hq].z, hh[hq].w});
                    x[q] = y0 + y1;
                    y[q] = pk_cmulc(y0 - y1, wq);
                });
                if constexpr (c == 0) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 8; ++q) hh[q] = h4[8 + q];
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            DftPk<16, true>::run(x);
#pragma unroll
            for (int q = 0; q < 16; ++q) ps0[q] = x[q];
            DftPk<16, true>::run(y);
#pragma unroll
            for (int q = 0; q < 16; ++q) ps1[q] = y[q];
        }
        PSX_STAMP(7);
        wave_sync();
        PSX_STAMP(8);

        // ---- inverse stage B
        {
            v2f *pb0;
            const v2f *rowB;
            stageB_at(pb0, rowB);
            v2f *const pb1 = pb0 + MP;
            v2f wt[16];
#pragma unroll
            for (int q = 1; q < 16; ++q) wt[q] = lds_read(rowB + q);
            wt[0] = (v2f){1.f, 0.f};
            v2f v0[16], v1[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v0[q] = lds_read(pb0 + offB(q));
#pragma unroll
            for (int q = 0; q < 16; ++q) v1[q] = lds_read(pb1 + offB(q));
            invB_regs(v0, wt);
#pragma unroll
            for (int q = 0; q < 16; ++q) pb0[offB(q)] = v0[q];
            invB_regs(v1, wt);
#pragma unroll
            for (int q = 0; q < 16; ++q) pb1[offB(q)] = v1[q];
        }
        PSX_STAMP(9);
        lds_barrier();                               // (2)
        PSX_STAMP(10);

        // ---- inverse stage A.  Round E parks its result (point n0 + 512 q in leg q) in the workgroup's own buffer; round O
        // fetches it back, forms y = ye -+ w_R^-m' yo and stores: point m' is output sample m' - sh, and sample m' + Q - sh as well
        // when m' < Lx (the wrapped outputs, with their fix-up).
        {
            v2f v[R1], pw[5];
            int lineA, nA;
            stageA_at(lineA, nA);
            const int n0 = 2 * nA + lineA;
            const v2f *const pA = Lb + lineA * MP + nA + (nA >> 5);
#pragma unroll
            for (int q = 0; q < R1; ++q) v[q] = pA[q * BSTR];
            v2f cf = (v2f){0.f, 0.f};
            if (ps != 0) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) cf += CF[c * LXM + (n0 & (LXM - 1))];
            }
            lds_barrier();                           // (3)
            PSX_STAMP(11);
            tw_powers<R1>(pw, tP + nA * LDP);
            twiddle_A2<R1, true>(v, pw);
            __builtin_amdgcn_sched_barrier(0);
            DftPk<R1, true>::run(v);
            int to = tid;
            asm volatile("" : "+v"(to));
            const __amdgpu_buffer_rsrc_t rpk = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<v2f *>(a.wgpart) + (size_t)blockIdx.x * Q, 0, Q * 8, 0x00020000);
            if (ps != 0) {
                const float2 wdf = a.w4[n0];
                const v2f wd = (v2f){wdf.x, wdf.y};
                v4u o[R1 / 2];
#pragma unroll
                for (int q = 0; q < R1 / 2; ++q) o[q] = __builtin_amdgcn_raw_buffer_load_b128(rpk, to * 16, q * TE * 16, 0);
                __builtin_amdgcn_sched_barrier(0);
                pk_static_for<0, R1>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = decltype(qc)::value;
                    v[q] = pk_cmul(v[q], pk_twiddle<64, q, true>(wd));        // z = w_R^-m' yo,  w_R^-m' = w_R^(+n0) exp(+2 pi i q / 64)
                });
                __builtin_amdgcn_sched_barrier(0);
                // the wrapped output first (it needs ye + z of leg 0), then every leg becomes ye - z
                const v2f ye0 = __builtin_bit_cast(v2f, (v2u){o[0].x, o[0].y});
                const v2f vw = ye0 + v[0] + cf;
#pragma unroll
                for (int q = 0; q < R1 / 2; ++q) {
                    v[2 * q] = __builtin_bit_cast(v2f, (v2u){o[q].x, o[q].y}) - v[2 * q];
                    v[2 * q + 1] = __builtin_bit_cast(v2f, (v2u){o[q].z, o[q].w}) - v[2 * q + 1];
                }
                const int lu = l;
                const bool lok = lu < a.nlines;
                const int ifirst = n0 - sh;                          // sample of leg 0; leg q: + 512 q; the wrapped one: + Q
                const int e0 = a.out_blocked ? ((ifirst >> IBS) * a.nlines + lu) * IB + (ifirst & (IB - 1)) : ifirst;
                const int estep = a.out_blocked ? (512 / IB) * a.nlines * IB : 512;
                const int64_t wbase = a.out_blocked ? 0 : (int64_t)lu * a.out_ld;
                const int welems = lok ? (a.out_blocked ? ((N + IB - 1) / IB) * IB * a.nlines : N) : 0;
                // a wrapped leg whose butterfly has none (n0 >= Lx) would land on sample n0 + Q - sh >= N: dropped by the window
                store_legs<R1, 0>(v, vw, reinterpret_cast<v2f *>(a.wave_out[d]), a.inten_out[d], wbase, welems, e0, estep,
                                  (v2f){a.gph[d].x, a.gph[d].y}, a.scale[d], a.accumulate);
            }
            int pe = ps;
            asm volatile("" : "+s"(pe));             // opaque: seen as the complement of the test above, the two branches are merged again
            if (pe == 0) {
#pragma unroll
                for (int q = 0; q < R1 / 2; ++q) {
                    const v2u lo = __builtin_bit_cast(v2u, v[2 * q]), hi = __builtin_bit_cast(v2u, v[2 * q + 1]);
                    __builtin_amdgcn_raw_buffer_store_b128((v4u){lo.x, lo.y, hi.x, hi.y}, rpk, to * 16, q * TE * 16, 0);
                }
            }
        }
        PSX_STAMP(12);
        lds_barrier();                               // (4)
        PSX_STAMP(13);
    }
}

// ---- kernel spectrum of the two rounds: taps h[d] = hP[d] / P (d < P <= 2Q) folded to Q points -- round E: h[d] + h[d + Q],
// round O: (h[d] - h[d + Q]) w_R^d -- ready for ONE batched forward transform of Q points
__global__ void k_p2x_pad(const double2 *__restrict__ hP, double2 *__restrict__ buf, int P) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * Q) return;
    const int sg = t / Q, d = t - sg * Q;
    double2 h0 = d < P ? hP[d] : make_double2(0.0, 0.0);
    const double2 h1 = d + Q < P ? hP[d + Q] : make_double2(0.0, 0.0);
    double2 v;
    if (sg == 0) {
        v = make_double2((h0.x + h1.x) / P, (h0.y + h1.y) / P);
    } else {
        const double dx = (h0.x - h1.x) / P, dy = (h0.y - h1.y) / P;
        double s, c;
        sincospi(-(double)d / (double)Q, &s, &c);            // w_R^d = exp(-2 pi i d / 2Q)
        v = make_double2(dx * c - dy * s, dx * s + dy * c);
    }
    buf[t] = v;
}

// out: [2 rounds][M] float4 = (G[k(p)], G[k(p) + M]) / (2 Q) for the M positions of an LDS line, then the first LXMAX taps (float2)
__global__ void k_p2x_perm(const double2 *__restrict__ Hh, const double2 *__restrict__ hP, float2 *__restrict__ out, int P) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 2 * M) {
        const int sg = t / M, p = t - sg * M;
        const int k = (p >> 8) + R1 * ((p >> 4) & 15) + 16 * R1 * (p & 15);
        const double2 g0 = Hh[(size_t)sg * Q + k], g1 = Hh[(size_t)sg * Q + k + M];
        const double sc = 1.0 / (2.0 * Q);
        reinterpret_cast<float4 *>(out)[t] = make_float4((float)(g0.x * sc), (float)(g0.y * sc), (float)(g1.x * sc), (float)(g1.y * sc));
    } else if (t < 2 * M + LXM) {
        const int d = t - 2 * M;
        const double2 v = d < P ? hP[d] : make_double2(0.0, 0.0);
        out[4 * M + d] = make_float2((float)(v.x / P), (float)(v.y / P));
    }
}

// w2[s] = w_Q^k0 of slab s (k0 = k1 + 32 k2, s = 16 k1 + k2); w4[n0] = exp(+2 pi i n0 / R), n0 < 512
__global__ void k_p2x_twiddles(float2 *w2, float2 *w4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 512) return;
    double s, c;
    const int k0 = (i >> 4) + R1 * (i & 15);
    sincospi(-2.0 * (double)k0 / (double)Q, &s, &c);
    w2[i] = make_float2((float)c, (float)s);
    sincospi(2.0 * (double)i / (double)RR, &s, &c);
    w4[i] = make_float2((float)c, (float)s);
}

template <bool CONTIG>
int launch_x(const LineArgs &la, hipStream_t st, const char *name) {
    static std::atomic<unsigned long long> attr_mask{0};
    if (first_on_device(attr_mask))
        PSX_HIP(hipFuncSetAttribute((const void *)k_fresnel_p2x<CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(MAX_TAPS)));
    if (la.n_dist < 1 || la.n_dist > MAX_TAPS || (!CONTIG && (la.dist_inner || !la.in_blocked)) || (CONTIG && la.in_si != 1))
        return fail(PSX_E_STATE, "LDS engine (power-of-two lines, two rounds): %d distances, dist_inner %d", la.n_dist, la.dist_inner);
    const int nwork = la.nlines * (la.dist_inner ? 1 : la.n_dist);
    int nslot = current_cu_count() / 8;
    if (nslot > (nwork + 7) / 8) nslot = (nwork + 7) / 8;
    if (!la.wgpart || !la.w4 || !la.w2 || 8 * nslot > la.wg_groups)
        return fail(PSX_E_STATE, "LDS engine: %d workgroups for %d private line buffers", 8 * nslot, la.wg_groups);
    PSX_TIMED(name, st, k_fresnel_p2x<CONTIG><<<8 * nslot, TT, lds_bytes(la.n_dist), st>>>(la));
    return launch_check(name);
}

}  // namespace

namespace psx {
namespace p2 {

bool x_serves(int N, int margin) {
    const int P = N + 2 * margin, L = N + P - 1;
    // the partner x[n + Q] sits (P - Q) / 2 points earlier in the same LDS line: P - Q even, positive, within the slack; the
    // wrapped outputs within LXMAX; the mirrors (2 margin threads of one loader wave... of the 256) and the fix-up's saved ranges
    return (N % 2 == 0) && P > Q && P - Q <= 2 * LXMAX - 2 && L - RR <= LXMAX && L - RR <= N - 1 && 2 * margin <= TLD && N <= Q && margin >= 1;
}
int x_points() { return Q; }
size_t x_spectrum_elems() { return (size_t)4 * M + LXMAX; }
size_t x_line_buffer_elems() { return (size_t)Q; }

int x_build_twiddles(float2 *w2, float2 *w4, hipStream_t st) {
    k_p2x_twiddles<<<2, 256, 0, st>>>(w2, w4);
    return launch_check("k_p2x_twiddles");
}

int x_pad_taps(const double2 *hP, double2 *buf, int P, hipStream_t st) {
    PSX_TIMED("k_kern_pad", st, k_p2x_pad<<<(2 * Q + 255) / 256, 256, 0, st>>>(hP, buf, P));
    return launch_check("k_p2x_pad");
}

int x_perm_spectrum(const double2 *Hh, const double2 *hP, float2 *out, int P, hipStream_t st) {
    PSX_TIMED("k_kern_perm", st, k_p2x_perm<<<(2 * M + LXM + 255) / 256, 256, 0, st>>>(Hh, hP, out, P));
    return launch_check("k_p2x_perm");
}

int x_launch(bool contig, const lines::LineArgs &la, hipStream_t st, const char *name) {
    return contig ? launch_x<true>(la, st, name) : launch_x<false>(la, st, name);
}

}  // namespace p2
}  // namespace psx
