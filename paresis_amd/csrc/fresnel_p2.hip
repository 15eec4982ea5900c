// fresnel_p2.hip -- power-of-two line kernels of the LDS Fresnel engine (K3..K8 of Experiment.wavePropagation, EXP:219-252).
//
// Same operator, same three launches and same persistent-workgroup structure as fresnel_lds.hip (read its header first); what
// differs is the transform a line goes through.  The reference pads a line of N samples to P = N + 2*margin and convolves it
// circularly with the P taps h = IDFT_P(chirp) (EXP:236-251).  k_fresnel_lines evaluates that as a LINEAR convolution of the
// periodic / mirrored extension e (N + P - 1 = 2N + 29 points at margin 15) through a transform of 576 R3 >= N + P - 1 points:
// 9216 for N = 4096.  Here the transform has M = 256 R1 points, the power of two just BELOW N + P - 1 (8192 for N = 4096):
//     y_c = IDFT_M( FFT_M(e[0 .. M)) * FFT_M(h) )          (circular, period M)
// is the wanted output y[m] = sum_d h[d] e[m - d] for every m in [P - 1, M): the window of P taps does not wrap.  The last
// Lx = N + P - 1 - M outputs (29) have m >= M; their circular slot m' = m - M holds the right sum except for the taps d <= m',
// which met e[m' - d] instead of e[M + m' - d]:
//     y[M + m'] = y_c[m'] + sum_{t <= m'} h[t] * (e[M + m' - t] - e[m' - t]),        m' < Lx,
// Lx (Lx + 1) / 2 = 435 complex multiply-adds per line and distance, done by the loader waves while the engine transforms
// (tools/p2_model.py: the same arithmetic in numpy, 1e-14 against the reference operator).  11 % fewer points than 9216, and
// radices 32 x 16 x 16 whose butterflies cost 1.3 packed instructions per point and bit of the index against 1.6 for 24 x 24 x 16.
//
// One persistent workgroup per CU, 12 waves: 8 engine waves (512 threads x 32 points = the 16384 points in LDS: 2 lines of
// 8192, 4 of 4096, 8 of 2048 or 16 of 1024) and 4 loader waves.  Three waves per SIMD leave a wave 168 VGPRs: an engine thread
// holds the 32 points of its stage in registers -- one radix-32 butterfly in the first stage, two radix-16 butterflies in the
// others, the second one's LDS reads in flight under the first one's arithmetic.
//     stage A: radix R1 over stride 256, twiddle w_M^{m k1}               (workgroup barriers either side)
//     stage B: radix 16 over stride 16 inside each block of 256, twiddle w_256^{n3 k2}
//     stage C: radix 16 on 16 contiguous points, x FFT_M(h), inverse radix 16  ("middle stage", slab by slab)
// Between the barriers after forward stage A and before inverse stage A a wave only touches the 2048 points it owns (8 blocks of
// 256 = 128 slabs of 16), as in fresnel_lds.hip.  Position p = 256 k1 + 16 k2 + k3 of a transformed line holds frequency
// k1 + R1 k2 + 16 R1 k3; the kernel-spectrum table is stored in that order (p2::perm_spectrum).
#include "fresnel_p2.hpp"

#include <atomic>

#include "fresnel_p2_dev.hpp"

using namespace psx;
using namespace psx::lines;
using namespace psx::p2dev;

namespace {

constexpr int TE = 512;        // engine threads
constexpr int TLD = 256;       // loader threads
constexpr int TT = TE + TLD;
constexpr int TOT2 = 16384;    // complex points resident in LDS per workgroup
constexpr int LXM = psx::p2::LXMAX;
// build-time A/B switches (tools/ab_p2.sh builds one library per setting; both arms of a comparison run on ONE box)
#ifndef PSX_P2_SKIP_LEGS
#define PSX_P2_SKIP_LEGS 1     // inverse stage A: the legs that lie before sample 0 for every butterfly are not stored
#endif
#ifndef PSX_P2_PRIO
#define PSX_P2_PRIO 0          // 1: static priority 1 for engine waves 4-7 (the younger wave of every SIMD)
#endif
#ifndef PSX_P2_TWEARLY
#define PSX_P2_TWEARLY 0       // 1: stage A's twiddle powers are read together with the butterfly's inputs
#endif
#ifndef PSX_P2_LDPRIO
#define PSX_P2_LDPRIO 0        // priority of the loader waves while they issue a round's fetch
#endif

template <int R1_, bool DUAL_>
struct G2 {
    static constexpr int R1 = R1_, M = 256 * R1, LINES = TOT2 / M, LH = DUAL_ ? LINES / 2 : LINES;
    // a line buffer: M points + pad slots + the LXM dropped samples e[M + j] behind them (the loaders write every sample at its
    // position of the extension; positions >= M are not part of the transform and keep what the fix-up needs)
    static constexpr int MP = M + M / 32 + LXM;
    static constexpr int NBA = 32 / R1;          // stage-A butterflies per engine thread
    static constexpr int SPL = M / 16;           // slabs per line
    static constexpr int LB = R1 == 32 ? 5 : (R1 == 16 ? 4 : (R1 == 8 ? 3 : 2));   // log2(R1): powers w^(n 2^b) per stage-A row
    static constexpr int LDP = LB | 1, LDB = 17; // padded rows of the twiddle tables (odd: consecutive rows on different banks)
    // fix-up: TPL loader threads per LDS line, the taps of an output in NCH chunks of CL (one partial sum per chunk)
    static constexpr int TPL = TLD / LINES, NCH = TPL >= 2 * LXM ? TPL / LXM : 1, CL = LXM / NCH;
    // LDS map, in float2 elements
    static constexpr int O_TP = LINES * MP, O_TB = O_TP + 256 * LDP, O_SA = O_TB + 16 * LDB, O_C = O_SA + LH * LXM,
                         O_UQ = O_C + NCH * LINES * LXM, O_HT = O_UQ + 2;
    static constexpr size_t lds_bytes(int ntap) { return sizeof(float2) * (size_t)(O_HT + ntap * LXM); }
    static constexpr int max_taps = (160 * 1024 / (int)sizeof(float2) - O_HT) / LXM < MAX_LINE ? (160 * 1024 / (int)sizeof(float2) - O_HT) / LXM : MAX_LINE;
    static_assert(LINES >= 2 && NBA >= 1 && max_taps >= 16, "LDS budget");
};

#define PSX_STAMP(k) PSX_STAMP_IF(k, tid == 0)

// CONTIG: the samples of a line are adjacent in memory (pass 1 reads rows of the transposed source); otherwise the lanes of a
// loader wave walk across the lines of the group (pass 2 reads columns of the blocked intermediate).
// DUAL (pass 1 of a call with several distances): a round is HALF the LDS lines' worth of image lines and TWO distances -- the
// lines are transformed forward once, the middle stage writes the product with the first distance's spectrum in place and the
// product with the second one's into the same slab of the other half of the LDS lines, the inverse stages run at full width.
// QUEUE: work units from per-workgroup queues with stealing inside the XCD (fresnel_lds.hip, k_fresnel_lines).
template <int R1, bool CONTIG, bool DUAL, bool QUEUE>
__global__ __launch_bounds__(TT) void k_fresnel_p2(LineArgs a) {
    using GE = G2<R1, DUAL>;
    constexpr int M = GE::M, LINES = GE::LINES, LH = GE::LH, MP = GE::MP, NBA = GE::NBA, SPL = GE::SPL, LDP = GE::LDP, LDB = GE::LDB;
    constexpr int NCH = GE::NCH, CL = GE::CL, TPL = GE::TPL;
    constexpr int LPG = LH;          // image lines per round
    constexpr int LL = LH;           // LDS lines the loaders fill
    static_assert(!DUAL || CONTIG, "DUAL is pass 1");
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    v2f *const Lb = reinterpret_cast<v2f *>(lds);
    v2f *const tP = Lb + GE::O_TP, *const tB = Lb + GE::O_TB;
    v2f *const SA = Lb + GE::O_SA;         // [LH][LXM]: e[j], j < LXM, of every image line of the round (the transform overwrites them)
    v2f *const CF = Lb + GE::O_C;          // [NCH][LINES][LXM]: fix-up of the wrapped outputs of every LDS line, one partial sum per tap chunk
    v2f *const HT = Lb + GE::O_HT;         // [distances][LXM]: the first taps of every distance of the launch
    int *const uq = reinterpret_cast<int *>(Lb + GE::O_UQ);
    const int tid = threadIdx.x;
    const int N = a.N, mg = a.margin;
    const int Lx = a.L - M;                // outputs whose window wraps (<= 0: none)
    const bool upper_half = PSX_P2_SKIP_LEGS && ((a.P - 1) >> 8) >= R1 / 2;   // the lower half of an inverse stage-A butterfly's legs lies before sample 0 (store_legs)

    // ---- work units: exactly k_fresnel_lines' order (XCD-contiguous chunks of line groups, static shares or queues)
    const int ngroups = (a.nlines + LPG - 1) / LPG;
    const int nwork = a.dist_inner ? ngroups : ngroups * a.n_dist;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int cq = nwork >> 3, cr = nwork & 7;
    const int cstart = xcd * cq + (xcd < cr ? xcd : cr), clen = cq + (xcd < cr ? 1 : 0);
    const int nunits = slot < clen ? (clen - slot + nslot - 1) / nslot : 0;
    const int nsub = DUAL ? (a.n_dist + 1) / 2 : a.n_dist;
    const int nj = a.dist_inner ? nunits * nsub : nunits;
    constexpr bool DYN = QUEUE;
    const int nsubr = a.dist_inner ? nsub : 1;
    auto item = [&](int j, int &d, int &g) __attribute__((always_inline)) {
        if (a.dist_inner) {
            const int u = j / nsub;
            d = j - u * nsub;                                // DUAL: index of the distance PAIR
            g = cstart + (DYN ? uq[u & 3] : slot + u * nslot);
        } else {
            const int wk = cstart + (DYN ? uq[j & 3] : slot + j * nslot);
            d = wk / ngroups;
            g = wk - d * ngroups;
        }
    };
    auto valid = [&](int j) __attribute__((always_inline)) { return DYN ? uq[(j / nsubr) & 3] >= 0 : j < nj; };
    auto qcount = [&](int v) __attribute__((always_inline)) { return &a.queue[16 * (xcd + 8 * v)]; };
    auto share = [&](int v) __attribute__((always_inline)) { return v < clen ? (clen - v + nslot - 1) / nslot : 0; };
    bool dry = false, own_dry = false;
    const int myshare = share(slot);
    auto steal = [&]() __attribute__((always_inline)) {
        const int ln = tid & 63;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const int v = ln;
            bool avail = false;
            if (v < nslot && v != slot)
                avail = __hip_atomic_load(qcount(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)share(v);
            const unsigned long long m = __ballot(avail);
            if (m == 0ull) {
                dry = true;
                return -1;
            }
            const unsigned long long hi = m >> ((slot + 1) & 63);
            const int pick = hi ? (slot + 1 + __builtin_ctzll(hi)) : __builtin_ctzll(m);
            unsigned k = 0u;
            if (ln == 0) k = atomicAdd(qcount(pick), 1u);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k < (unsigned)share(pick)) return pick + (int)k * nslot;
        }
        return -1;
    };

    unsigned first2 = 0u;
    if (DYN && tid == TE) first2 = atomicAdd(qcount(slot), 2u);
    // ---- stage twiddles and the first taps of every distance into LDS
    for (int idx = tid; idx < 256 * GE::LB; idx += TT) {
        const float2 wv = a.twA[idx];
        tP[(idx / GE::LB) * LDP + idx % GE::LB] = (v2f){wv.x, wv.y};
    }
    for (int idx = tid; idx < 256; idx += TT) {
        const float2 w = a.twB[idx];
        tB[(idx >> 4) * LDB + (idx & 15)] = (v2f){w.x, w.y};
    }
    {
        const int ntap = DUAL ? 2 * nsub : a.n_dist;
        for (int idx = tid; idx < ntap * LXM; idx += TT) {
            const float2 h = a.H[idx / LXM][M + idx % LXM];
            HT[idx] = (v2f){h.x, h.y};
        }
    }
    if constexpr (DYN) {
        if (tid == TE) {
            uq[0] = first2 < (unsigned)myshare ? slot + (int)first2 * nslot : -1;
            uq[1] = first2 + 1u < (unsigned)myshare ? slot + (int)(first2 + 1u) * nslot : -1;
        }
        lds_barrier();
    }

    if (tid >= TE) {
        // =============================== loader waves =====================================================================
        const int lt = tid - TE;
        constexpr int STEP = TLD / LL, NLD = M / (2 * STEP), PSTEP = STEP + STEP / 32;
        static_assert(TLD % LL == 0 && NLD * STEP == M / 2, "sample ownership covers M / 2 samples of every line");
        constexpr bool AFFL = (STEP % 32 == 0);
        const int line = CONTIG ? lt / STEP : lt % LL, i0 = CONTIG ? lt % STEP : lt / LL;
        const int nmir = LL * 2 * mg;
        const int lm = lt % LL;
        int im = -1, jm = 0;
        if (lt < nmir) {
            const int r = lt / LL;
            im = r < mg ? r + 1 : N - 1 - 2 * mg + r;
            jm = r < mg ? N + 2 * mg - 1 - im : 2 * N - 3 - im;
        }
        float2 *base = lds + line * MP;
        const int ja = i0 + N + 2 * mg - 1, jb = i0 - 1;
        const int oa = phys(ja), ob = phys(jb);
        float2 *sa_line = lds + GE::O_SA + line * LXM;

        // Pass 2 (!CONTIG) reads the blocked intermediate, where sample i of the lines 8b .. 8b+7 is one 64-byte piece: a thread
        // moves sample i of TWO adjacent lines (2 c2, 2 c2 + 1 of the group) with one 16-byte load -- half the lanes and half the
        // wave-instructions of a load per (line, sample).  (First form of this kernel, 8-byte loads: the loaders needed 5.35 us to
        // issue a round's fetch at 4096^2, the engine reached barrier (2) after 5.1: gpurun_out/r6s1.)
        constexpr int CH = LL / 2 > 0 ? LL / 2 : 1, STEPV = TLD / CH, NLV = M / (2 * STEPV), PSTEPV = STEPV + STEPV / 32;
        static_assert(CONTIG || (LL % 2 == 0 && IB % 2 == 0 && STEPV % 32 == 0 && NLV * STEPV == M / 2), "16-byte loads of the blocked intermediate");
        const int c2 = lt % CH, i0v = lt / CH;
        float2 *base_v = lds + 2 * c2 * MP;
        const int jav = i0v + N + 2 * mg - 1, jbv = i0v - 1;
        const int oav = phys(jav), obv = phys(jbv);
        float2 *sa_v = lds + GE::O_SA + 2 * c2 * LXM;
        // byte offset of (sample i0v, line pair c2) from sample 0 of the group's first line, in the blocked intermediate
        const int voff_v = ((((2 * c2) / IB) * N + i0v) * IB + (2 * c2) % IB) * (int)sizeof(float2);

        float2 xs[CONTIG ? NLD : 1], xm = make_float2(0.f, 0.f);
        float4 xv[CONTIG ? 1 : NLV];
        auto fetch = [&](int j) __attribute__((always_inline)) {
            int d, g;
            item(j, d, g);
            if (a.dist_inner && d != 0) return;
            const float2 *src = a.src[d];
            const int l0 = g * LPG;
            if constexpr (!CONTIG) {
                // The group's lines l0 .. l0 + LL - 1 lie in one block of IB lines (two at LL = 16): the descriptor starts at
                // sample 0 of line l0 and ends with the last block that holds a line of the image, so a sample index >= N and a
                // block past the image fall outside its range and read zeros -- one address add per load, no compare, no select
                // (the loader waves are the youngest of their SIMDs: with eight vector instructions per load a round's fetch took
                // 4.9 us to ISSUE, gpurun_out/r6s2).  A pair (l, l + 1) whose second line lies past the image reads the block's
                // padding; the spread drops it.
                constexpr int LLB = LL > IB ? LL / IB : 1;
                const int nblk = min(LLB, (a.nlines - l0 + IB - 1) / IB);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float2 *>(src) + ((int64_t)(l0 / IB) * N) * IB + l0 % IB, 0,
                    nblk * N * IB * (int)sizeof(float2) - (l0 % IB) * (int)sizeof(float2), 0x00020000);
#pragma unroll
                for (int k = 0; k < NLV; ++k) {
                    const v4u r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_v + k * (STEPV * IB * (int)sizeof(float2)), 0, 0);
                    xv[k] = __builtin_bit_cast(float4, r);
                }
                const int64_t pixm = ((int64_t)((l0 + lm) / IB) * N + im) * IB + (l0 + lm) % IB;
                xm = src[(im >= 0 && l0 + lm < a.nlines) ? pixm : (int64_t)0];
                return;
            } else {
                // rows of the transposed source, in_si == 1: the descriptor covers the group's lines inside the image
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float2 *>(src) + (int64_t)l0 * a.in_sl, 0, min(LL, a.nlines - l0) * (int)a.in_sl * (int)sizeof(float2), 0x00020000);
                const int voff = (line * (int)a.in_sl + i0) * (int)sizeof(float2);
#pragma unroll
                for (int k = 0; k < NLD; ++k) {
                    const v2u r = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + k * (STEP * (int)sizeof(float2)), 0, 0);
                    xs[k] = __builtin_bit_cast(float2, r);
                }
                const int64_t pixm = (int64_t)im + (int64_t)(l0 + lm) * a.in_sl;
                xm = src[(im >= 0 && l0 + lm < a.nlines) ? pixm : (int64_t)0];
                return;
            }
        };
        auto spread = [&](int j) __attribute__((always_inline)) {
            int d, g;
            item(j, d, g);
            const int l0 = g * LPG;
            const bool line_ok = l0 + line < a.nlines;
            for (int jz = a.L + lt; jz < M; jz += TLD) {             // zeros in [L, M) of every line (short lines only)
                const int pz = phys(jz);
#pragma unroll
                for (int ln = 0; ln < LL; ++ln) lds[ln * MP + pz] = make_float2(0.f, 0.f);
            }
            if constexpr (!CONTIG) {
                const bool ok0 = l0 + 2 * c2 < a.nlines, ok1 = l0 + 2 * c2 + 1 < a.nlines;
#pragma unroll
                for (int k = 0; k < NLV; ++k) {
                    if (i0v + STEPV * k < N) {
                        const float2 x0 = ok0 ? make_float2(xv[k].x, xv[k].y) : make_float2(0.f, 0.f);
                        const float2 x1 = ok1 ? make_float2(xv[k].z, xv[k].w) : make_float2(0.f, 0.f);
                        base_v[oav + k * PSTEPV] = x0;
                        base_v[MP + oav + k * PSTEPV] = x1;
                        if (k > 0 || jbv >= 0) {
                            base_v[obv + k * PSTEPV] = x0;
                            base_v[MP + obv + k * PSTEPV] = x1;
                        }
                        if (STEPV * k <= LXM) {
                            const int js = jbv + STEPV * k;
                            if ((unsigned)js < (unsigned)LXM) {
                                sa_v[js] = x0;
                                sa_v[LXM + js] = x1;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < (CONTIG ? NLD : 0); ++k) {
                if (i0 + STEP * k < N) {
                    const float2 x = line_ok ? xs[k] : make_float2(0.f, 0.f);
                    // first image: position i + P - 1 (the last Lx samples land behind the M points of the transform: they are
                    // the e[M + j] of the fix-up); second image one period earlier: position i - 1
                    if (AFFL) {
                        base[oa + k * PSTEP] = x;
                        if (k > 0 || jb >= 0) base[ob + k * PSTEP] = x;
                    } else {
                        base[phys(ja + STEP * k)] = x;
                        if (k > 0 || jb >= 0) base[phys(jb + STEP * k)] = x;
                    }
                    if (STEP * k <= LXM) {                           // compile-time: the few k that can hold a sample below LXM
                        const int js = jb + STEP * k;
                        if ((unsigned)js < (unsigned)LXM) sa_line[js] = x;
                    }
                }
            }
            if (im >= 0) lds[lm * MP + phys(jm)] = l0 + lm < a.nlines ? xm : make_float2(0.f, 0.f);
            for (int t = lt + TLD; t < nmir; t += TLD) {             // margins beyond TLD / (2 LL): rare
                const int ln = t % LL, r = t / LL;
                const int i2 = r < mg ? r + 1 : N - 1 - 2 * mg + r, j2 = r < mg ? N + 2 * mg - 1 - i2 : 2 * N - 3 - i2;
                float2 x2 = make_float2(0.f, 0.f);
                if (l0 + ln < a.nlines)
                    x2 = a.src[d][a.in_blocked ? ((int64_t)((l0 + ln) / IB) * N + i2) * IB + (l0 + ln) % IB
                                            : (int64_t)i2 * a.in_si + (int64_t)(l0 + ln) * a.in_sl];
                lds[ln * MP + phys(j2)] = x2;
            }
        };
        // fix-up of the wrapped outputs of round j: CF[c][lb][m'] = sum over the taps t of chunk c, t <= m', of
        // h_dd[t] * (e[M + m' - t] - e[m' - t]).  LDS traffic only (the loads of the next group are in flight meanwhile and their
        // counter is not waited on); every term of a thread is independent, so its reads travel together.
        auto fixup = [&](int j) __attribute__((always_inline)) {
            if (Lx <= 0) return;
            int d, g;
            item(j, d, g);
            int lo = lt;
            asm volatile("" : "+v"(lo));                            // opaque copy: what follows is re-derived every round, not kept across the loop
            const int lb = lo / TPL, r = lo % TPL;
            const int lbi = DUAL ? lb % LH : lb;                     // LDS line that holds the image line's samples
            const int dd = DUAL ? 2 * d + lb / LH : d;
            const v2f *ht = HT + dd * LXM;
            const v2f *sa = SA + lbi * LXM;
            const v2f *sb = Lb + lbi * MP + M + M / 32;              // phys(M + j) = M + M/32 + j for j < 32
            const int c = NCH > 1 ? r / LXM : 0;
            for (int mp = NCH > 1 ? r % LXM : r; mp < LXM; mp += (NCH > 1 ? LXM : TPL)) {
                v2f acc = (v2f){0.f, 0.f};
#pragma unroll
                for (int tt = 0; tt < CL; ++tt) {
                    const int t = c * CL + tt;
                    const int jx = mp >= t ? mp - t : 0;             // clamped: the reads stay unconditional
                    const v2f dl = lds_read(sb + jx) - lds_read(sa + jx);
                    const v2f term = pk_cmul(dl, lds_read(ht + t));
                    acc += mp >= t ? term : (v2f){0.f, 0.f};
                }
                CF[(c * LINES + lb) * LXM + mp] = acc;
            }
        };

        if (valid(0)) {
            fetch(0);
            spread(0);
        }
        lds_barrier();                                   // (0)
        int ucur = 0, usub = 0;
        auto ring_ok = [&](int u) __attribute__((always_inline)) { return uq[u & 3] >= 0; };
        for (int j = 0; DYN ? ring_ok(ucur) : j < nj; ++j) {
            const bool last_sub = usub + 1 == nsubr;
            const bool more = DYN ? ring_ok(last_sub ? ucur + 1 : ucur) : j + 1 < nj;
            lds_barrier();                               // (1) engine: forward stage A done
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 16] = wall_clock64();
            const bool qround = DYN && lt < 64 && usub == 0;
            const bool claiming = qround && ring_ok(ucur + 1);
            unsigned mine = 0xffffffffu;
            if (claiming && !own_dry && lt == 0) mine = atomicAdd(qcount(slot), 1u);
            if (PSX_P2_LDPRIO) __builtin_amdgcn_s_setprio(PSX_P2_LDPRIO);
            if (more) fetch(j + 1);
            if (PSX_P2_LDPRIO) __builtin_amdgcn_s_setprio(0);
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 17] = wall_clock64();
            fixup(j);
            lds_barrier();                               // (2) engine: wave-private stages done
            lds_barrier();                               // (3) engine: inverse stage A holds all of LDS in registers
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 18] = wall_clock64();
            __builtin_amdgcn_s_setprio(3);
            if (more) spread(j + 1);
            __builtin_amdgcn_s_setprio(0);
            if (qround) {
                int unit = -1;
                if (claiming) {
                    const unsigned k = __builtin_amdgcn_readfirstlane(mine);
                    if (k < (unsigned)myshare) unit = slot + (int)k * nslot;
                    else own_dry = true;
                    if (unit < 0 && !dry) unit = steal();
                }
                if (lt == 0) uq[(ucur + 2) & 3] = unit;
            }
            if (last_sub) {
                usub = 0;
                ++ucur;
            } else {
                ++usub;
            }
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 19] = wall_clock64();
            lds_barrier();                               // (4) next group is in LDS
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 20] = wall_clock64();
        }
        if (DYN && lt < 64) {
            unsigned prev = 0u;
            if (lt == 0) {
                __threadfence();
                prev = atomicAdd(&a.queue[16 * 256], 1u);
            }
            prev = __builtin_amdgcn_readfirstlane(prev);
            if (prev == gridDim.x - 1) {
                for (int wq = lt; wq < (int)gridDim.x; wq += 64) atomicExch(&a.queue[16 * wq], 0u);
                if (lt == 0) atomicExch(&a.queue[16 * 256], 0u);
            }
        }
        return;
    }

    // =================================== engine waves =========================================================================
    const int w = tid >> 6, ln = tid & 63;
    // stage B: lane -> (block of the wave's four of this iteration, element n3).  Lanes 0-15 / 16-31 of a read group take
    // blocks two apart: their points then sit 16 banks apart (a block is 264 = 8 mod 32 slots long)
    const int n3 = ln & 15, blk = ((ln >> 4) & 1) * 2 + (ln >> 5);
    // middle stage: within each half-wave the first 16 lanes take the even slabs and the last 16 the odd ones (one pad slot per
    // TWO slabs: 16 different bank pairs per ds_write_b64 group)
    const int slabw = (ln & 32) + ((ln & 31) < 16 ? 2 * (ln & 31) : 2 * ((ln & 31) - 16) + 1);
    const v2f *rowB = tB + n3 * LDB;
    // What a wave owns between the barriers after forward stage A and before inverse stage A: the same 1024 points (4 blocks of 256
    // = 64 slabs) of TWO LDS lines -- lines 2p and 2p + 1 of the round (DUAL: an image line's buffer and the buffer of its second
    // distance, LH lines further) -- so that one slab of the kernel spectrum, fetched once, meets both lines: half the spectrum
    // loads of a one-line range, and the engine's loads share the CU's memory pipeline with the loaders' fetch.
    constexpr int RPL = R1 / 4;                        // ranges of 1024 points per line
    const int G0 = DUAL ? 4 * w + blk : 2 * (w / RPL) * R1 + 4 * (w % RPL) + blk, G1 = G0 + (DUAL ? 32 : R1);      // this lane's two blocks
    const int S0 = DUAL ? 64 * w + slabw : 2 * (w / RPL) * SPL + 64 * (w % RPL) + slabw, S1 = S0 + (DUAL ? LH * SPL : SPL);   // ... and slabs
    auto block_ptr = [&](int G) __attribute__((always_inline)) { return Lb + (G / R1) * MP + (G % R1) * BSTR + n3; };
    auto slab_ptr = [&](int S) __attribute__((always_inline)) {
        const int sl = S % SPL;
        return Lb + (S / SPL) * MP + 16 * sl + (sl >> 1);
    };
    // does this thread take part in the forward stages?  DUAL transforms the first LH lines only: with R1 = 32 that is one
    // line = the butterflies of the first four waves; with smaller radices every thread has its share
    const bool fwdA_on = !(DUAL && R1 == 32) || tid < TE / 2;
    constexpr int NBAF = DUAL ? (R1 == 32 ? 1 : NBA / 2) : NBA;      // forward stage-A butterflies per thread
    if (PSX_P2_PRIO && tid >= TE / 2) __builtin_amdgcn_s_setprio(1);
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 32 + 0] = wall_clock64();
    lds_barrier();                                       // (0) first group is in LDS
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 32 + 1] = wall_clock64();
    int eu = 0, es = 0;
    for (int j = 0; DYN ? uq[eu & 3] >= 0 : j < nj; ++j) {
        int d, g;
        item(j, d, g);
        const int l0 = g * LPG;
        if (++es == nsubr) {
            es = 0;
            ++eu;
        }
        PSX_STAMP(2);

        // ---- forward stage A: radix R1 over stride 256, twiddle w_M^{n q} = w_M^{16 nh q} w_M^{nl q}
        if (fwdA_on) {
            v2f v[NBAF][R1], pw[5];
#pragma unroll
            for (int i = 0; i < NBAF; ++i) {
                const int b = tid + TE * i, n = b & 255;
                const v2f *p = Lb + (b >> 8) * MP + n + (n >> 5);
#pragma unroll
                for (int q = 0; q < R1; ++q) v[i][q] = p[q * BSTR];
            }
            if (PSX_P2_TWEARLY) tw_powers<R1>(pw, tP + (tid & 255) * LDP);      // n = b & 255 is the same for every butterfly of a thread
#pragma unroll
            for (int i = 0; i < NBAF; ++i) {
                const int b = tid + TE * i;
                v2f *p = Lb + (b >> 8) * MP + (b & 255) + ((b & 255) >> 5);
                DftPk<R1, false>::run(v[i]);
                __builtin_amdgcn_sched_barrier(0);
                if (!PSX_P2_TWEARLY && i == 0) tw_powers<R1>(pw, tP + (tid & 255) * LDP);
                twiddle_A2<R1, false>(v[i], pw);
#pragma unroll
                for (int q = 0; q < R1; ++q) p[q * BSTR] = v[i][q];
            }
        }
        // The kernel spectrum (the engine's only global loads) travels well ahead of its use: the slab's 128 bytes are requested
        // here, before barrier (1) and before the loaders' fetch (DUAL: the second distance's in forward stage B, behind its reads).
        float4 hh[8], hh1[DUAL ? 8 : 1];
        {
            const float4 *h4 = reinterpret_cast<const float4 *>(a.H[DUAL ? 2 * d : d] + 16 * (S0 % SPL));
#pragma unroll
            for (int q = 0; q < 8; ++q) hh[q] = h4[q];
        }
        PSX_STAMP(3);
        lds_barrier();                               // (1)
        PSX_STAMP(4);

        // ---- forward stage B: radix 16 inside each block of 256, stride 16 (DUAL: the first distance's buffer only)
        {
            v2f wt[16];
#pragma unroll
            for (int q = 1; q < 16; ++q) wt[q] = lds_read(rowB + q);
            wt[0] = (v2f){1.f, 0.f};
            v2f *p0 = block_ptr(G0), *p1 = block_ptr(G1);
            v2f v0[16], v1[DUAL ? 1 : 16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v0[q] = lds_read(p0 + offB(q));
            if constexpr (DUAL) {
                const float4 *h4b = reinterpret_cast<const float4 *>(a.H[2 * d + 1] + 16 * (S0 % SPL));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) hh1[q] = h4b[q];
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) v1[q] = lds_read(p1 + offB(q));
            }
            fwdB_regs(v0, wt);
#pragma unroll
            for (int q = 0; q < 16; ++q) p0[offB(q)] = v0[q];
            if constexpr (!DUAL) {
                fwdB_regs(v1, wt);
#pragma unroll
                for (int q = 0; q < 16; ++q) p1[offB(q)] = v1[q];
            }
        }
        PSX_STAMP(5);
        wave_sync();
        PSX_STAMP(6);

        // ---- middle stage, slab by slab: forward radix 16 on contiguous points, x FFT_M(h_d), inverse radix 16, back to LDS
        {
            v2f *b0 = slab_ptr(S0), *b1 = slab_ptr(S1);
            v2f x[16], y[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = lds_read(b0 + q);
            if constexpr (!DUAL) {
#pragma unroll
                for (int q = 0; q < 16; ++q) y[q] = lds_read(b1 + q);
            }
            DftPk<16, false>::run(x);
            if constexpr (DUAL) {
                // the spectrum slab is shared by the two distances: y = x * H_first, then x *= H_second
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    y[2 * q] = pk_cmul(x[2 * q], (v2f){hh[q].x, hh[q].y});
                    y[2 * q + 1] = pk_cmul(x[2 * q + 1], (v2f){hh[q].z, hh[q].w});
                }
                DftPk<16, true>::run(y);
#pragma unroll
                for (int q = 0; q < 16; ++q) b0[q] = y[q];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    x[2 * q] = pk_cmul(x[2 * q], (v2f){hh1[q].x, hh1[q].y});
                    x[2 * q + 1] = pk_cmul(x[2 * q + 1], (v2f){hh1[q].z, hh1[q].w});
                }
                DftPk<16, true>::run(x);
#pragma unroll
                for (int q = 0; q < 16; ++q) b1[q] = x[q];
            } else {
                // two lines, one spectrum slab
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    x[2 * q] = pk_cmul(x[2 * q], (v2f){hh[q].x, hh[q].y});
                    x[2 * q + 1] = pk_cmul(x[2 * q + 1], (v2f){hh[q].z, hh[q].w});
                }
                DftPk<16, true>::run(x);
#pragma unroll
                for (int q = 0; q < 16; ++q) b0[q] = x[q];
                DftPk<16, false>::run(y);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    y[2 * q] = pk_cmul(y[2 * q], (v2f){hh[q].x, hh[q].y});
                    y[2 * q + 1] = pk_cmul(y[2 * q + 1], (v2f){hh[q].z, hh[q].w});
                }
                DftPk<16, true>::run(y);
#pragma unroll
                for (int q = 0; q < 16; ++q) b1[q] = y[q];
            }
        }
        PSX_STAMP(7);
        wave_sync();
        PSX_STAMP(8);

        // ---- inverse stage B (DUAL: the wave's 4 blocks of EITHER half of the LDS lines -- what its middle stage wrote)
        {
            v2f wt[16];
#pragma unroll
            for (int q = 1; q < 16; ++q) wt[q] = lds_read(rowB + q);
            wt[0] = (v2f){1.f, 0.f};
            v2f *p0 = block_ptr(G0), *p1 = block_ptr(G1);
            v2f v0[16], v1[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v0[q] = lds_read(p0 + offB(q));
#pragma unroll
            for (int q = 0; q < 16; ++q) v1[q] = lds_read(p1 + offB(q));
            invB_regs(v0, wt);
#pragma unroll
            for (int q = 0; q < 16; ++q) p0[offB(q)] = v0[q];
            invB_regs(v1, wt);
#pragma unroll
            for (int q = 0; q < 16; ++q) p1[offB(q)] = v1[q];
        }
        PSX_STAMP(9);
        lds_barrier();                               // (2)
        PSX_STAMP(10);

        // ---- inverse stage A; the wanted outputs leave for HBM straight from the registers.  Once every engine thread holds its
        // 32 points LDS is free: the loaders fill it with the next group meanwhile.
        {
            v2f v[NBA][R1], cf[NBA], pw[5];
            if (PSX_P2_TWEARLY) tw_powers<R1>(pw, tP + (tid & 255) * LDP);
#pragma unroll
            for (int i = 0; i < NBA; ++i) {
                const int b = tid + TE * i, n = b & 255;
                const v2f *p = Lb + (b >> 8) * MP + n + (n >> 5);
#pragma unroll
                for (int q = 0; q < R1; ++q) v[i][q] = p[q * BSTR];
                cf[i] = CF[(b >> 8) * LXM + (n & (LXM - 1))];
#pragma unroll
                for (int c = 1; c < NCH; ++c) cf[i] += CF[(c * LINES + (b >> 8)) * LXM + (n & (LXM - 1))];
            }
            lds_barrier();                           // (3)
            PSX_STAMP(11);
#pragma unroll
            for (int i = 0; i < NBA; ++i) {
                int bo = tid + TE * i;
                asm volatile("" : "+v"(bo));         // opaque: the per-leg addresses are formed here, not hoisted out of the round loop
                const int n = bo & 255, lb = bo >> 8;
                if (!PSX_P2_TWEARLY && i == 0) tw_powers<R1>(pw, tP + n * LDP);
                twiddle_A2<R1, true>(v[i], pw);
                __builtin_amdgcn_sched_barrier(0);
                // (computing only the upper half of the legs when the lower half is not stored -- 23 instead of 27 instructions per
                // radix-8 block -- puts two butterflies behind the uniform branch and spills 28 registers: not done)
                DftPk<R1, true>::run(v[i]);
                // leg q holds y_c[n + 256 q] = output sample n + 256 q - (P - 1); leg 0 is also the wrapped output n + M - (P - 1)
                const int ifirst = n - (a.P - 1);
                const int lbu = __builtin_amdgcn_readfirstlane(lb);          // a wave's 64 butterflies belong to one LDS line
                const int dd = DUAL ? 2 * d + lbu / LH : d;
                const int l = l0 + (DUAL ? lbu % LH : lbu);
                v2f *wo = reinterpret_cast<v2f *>(a.wave_out[dd]);
                float *io = a.inten_out[dd];
                const bool lok = l < a.nlines;
                const int e0 = a.out_blocked ? ((ifirst >> IBS) * a.nlines + l) * IB + (ifirst & (IB - 1)) : ifirst;
                const int estep = a.out_blocked ? (256 / IB) * a.nlines * IB : 256;
                const int64_t wbase = a.out_blocked ? 0 : (int64_t)l * a.out_ld;
                const int welems = lok ? (a.out_blocked ? ((N + IB - 1) / IB) * IB * a.nlines : N) : 0;
                if (upper_half)
                    store_legs<R1, R1 / 2>(v[i], v[i][0] + cf[i], wo, io, wbase, welems, e0, estep, (v2f){a.gph[dd].x, a.gph[dd].y},
                                           a.scale[dd], a.accumulate);
                else
                    store_legs<R1, 0>(v[i], v[i][0] + cf[i], wo, io, wbase, welems, e0, estep, (v2f){a.gph[dd].x, a.gph[dd].y},
                                      a.scale[dd], a.accumulate);
            }
        }
        PSX_STAMP(12);
        lds_barrier();                               // (4)
        PSX_STAMP(13);
    }
}

__global__ void k_p2_twiddles(float2 *twA, float2 *twB, int R1, int LB) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = 256 * R1;
    if (idx < 256 * LB) {
        const int n = idx / LB, b = idx % LB;
        double s, c;
        sincospi(-2.0 * (double)((n << b) % M) / (double)M, &s, &c);          // w_M^(n 2^b)
        twA[idx] = make_float2((float)c, (float)s);
    }
    if (idx < 256) {
        const int n = idx >> 4, k = idx & 15;
        double s, c;
        sincospi(-2.0 * (double)((n * k) % 256) / 256.0, &s, &c);             // w_256^{n3 k2}
        twB[idx] = make_float2((float)c, (float)s);
    }
}

__global__ void k_p2_perm(const double2 *__restrict__ Hh, const double2 *__restrict__ hP, float2 *__restrict__ out, int R1, int P) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = 256 * R1;
    if (p < M) {
        const int k1 = p >> 8, k2 = (p >> 4) & 15, k3 = p & 15;
        const double2 v = Hh[k1 + R1 * k2 + 16 * R1 * k3];
        out[p] = make_float2((float)(v.x / M), (float)(v.y / M));
    } else if (p < M + LXM) {
        const int t = p - M;
        const double2 v = t < P ? hP[t] : make_double2(0.0, 0.0);
        out[p] = make_float2((float)(v.x / P), (float)(v.y / P));
    }
}

int line_grid_slots(int nwork, int cap) {
    int nslot = current_cu_count() / 8;
    if (nslot > (nwork + 7) / 8) nslot = (nwork + 7) / 8;
    if (cap && nslot > cap) nslot = cap;
    return nslot;
}

template <int R1, bool CONTIG, bool DUAL, bool QUEUE = false>
int launch_t(const LineArgs &la, hipStream_t st, const char *name) {
    if constexpr (!QUEUE) {
        if (la.queue) return launch_t<R1, CONTIG, DUAL, true>(la, st, name);
    }
    using GE = G2<R1, DUAL>;
    static std::atomic<unsigned long long> attr_mask{0};
    if (first_on_device(attr_mask))
        PSX_HIP(hipFuncSetAttribute((const void *)k_fresnel_p2<R1, CONTIG, DUAL, QUEUE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)GE::lds_bytes(GE::max_taps)));
    if (la.n_dist < 1 || la.n_dist > MAX_LINE || (DUAL && !la.dist_inner))
        return fail(PSX_E_STATE, "LDS engine (power-of-two lines): %d distances, dist_inner %d", la.n_dist, la.dist_inner);
    const int ntap = DUAL ? 2 * ((la.n_dist + 1) / 2) : la.n_dist;
    if (ntap > GE::max_taps) return fail(PSX_E_STATE, "LDS engine (power-of-two lines): %d distances in one launch, %d fit", ntap, GE::max_taps);
    const int nwork = ((la.nlines + GE::LH - 1) / GE::LH) * (la.dist_inner ? 1 : la.n_dist);
    const int nslot = line_grid_slots(nwork, QUEUE ? 32 : 0);
    PSX_TIMED(name, st, k_fresnel_p2<R1, CONTIG, DUAL, QUEUE><<<8 * nslot, TT, GE::lds_bytes(ntap), st>>>(la));
    const int rc = launch_check(name);
    if constexpr (QUEUE) {
        if (rc) (void)hipMemsetAsync(la.queue, 0, sizeof(unsigned) * QUEUE_WORDS, st);
    }
    return rc;
}

template <int R1>
int launch_r(bool contig, bool dual, const LineArgs &la, hipStream_t st, const char *name) {
    if (dual) return launch_t<R1, true, true>(la, st, name);
    return contig ? launch_t<R1, true, false>(la, st, name) : launch_t<R1, false, false>(la, st, name);
}

}  // namespace

namespace psx {
namespace p2 {

int pick_r1(int N, int margin) {
    const int P = N + 2 * margin, L = N + P - 1;
    for (int r1 : {4, 8, 16, 32}) {
        const int M = 256 * r1;
        // the loaders own M / 2 samples of a line; the taps fit the transform; at most LXMAX outputs wrap, and none of the
        // positions the fix-up reads (j < Lx) is a mirrored sample (those start at position N - 1)
        if (N <= M / 2 && P <= M && L - M <= LXMAX && L - M <= N - 1) return r1;
    }
    return 0;
}

static int log2_r1(int R1) { return R1 == 32 ? 5 : (R1 == 16 ? 4 : (R1 == 8 ? 3 : 2)); }
size_t twA_elems(int R1) { return (size_t)256 * log2_r1(R1); }
int max_distances(int R1) {
    switch (R1) {
        case 4: return G2<4, false>::max_taps;
        case 8: return G2<8, false>::max_taps;
        case 16: return G2<16, false>::max_taps;
    }
    return G2<32, false>::max_taps;
}
size_t twB_elems() { return 256; }
size_t spectrum_elems(int R1) { return (size_t)256 * R1 + LXMAX; }
int lines_per_round(int R1, bool dual) { return (TOT2 / (256 * R1)) / (dual ? 2 : 1); }

int build_tables(float2 *twA, float2 *twB, int R1, hipStream_t st) {
    k_p2_twiddles<<<(256 * log2_r1(R1) + 255) / 256, 256, 0, st>>>(twA, twB, R1, log2_r1(R1));
    return launch_check("k_p2_twiddles");
}

int perm_spectrum(const double2 *Hh, const double2 *hP, float2 *out, int R1, int P, hipStream_t st) {
    const int n = 256 * R1 + LXMAX;
    PSX_TIMED("k_kern_perm", st, k_p2_perm<<<(n + 255) / 256, 256, 0, st>>>(Hh, hP, out, R1, P));
    return launch_check("k_p2_perm");
}

int launch(int R1, bool contig, bool dual, const lines::LineArgs &la, hipStream_t st, const char *name) {
    switch (R1) {
        case 4: return launch_r<4>(contig, dual, la, st, name);
        case 8: return launch_r<8>(contig, dual, la, st, name);
        case 16: return launch_r<16>(contig, dual, la, st, name);
        case 32: return launch_r<32>(contig, dual, la, st, name);
    }
    return fail(PSX_E_UNSUPPORTED, "LDS engine (power-of-two lines): unsupported radix %d", R1);
}

}  // namespace p2
}  // namespace psx
