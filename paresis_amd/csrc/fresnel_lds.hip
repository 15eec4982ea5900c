// fresnel_lds.hip -- LDS-resident FFT-convolution engine for the Fresnel propagator (K1 + K3..K8 in three kernels).
//
// Replaces Experiment.wavePropagation (Experiment.py:219-252) without ever forming the padded 2-D spectrum in HBM.
//
// The reference operator  crop(IDFT2_P(H * DFT2_P(reflect_pad(psi))))  with H(u,v) = c(u)c(v) separable is a 1-D
// circular convolution of period P = N+2*margin along axis 0 followed by one along axis 1.  Each 1-D line convolution
//     out[n] = sum_{d<P} h[d] * x_per[n + margin - d],   h = IDFT_P(c),   x_per = periodic extension of the reflect pad,
// is evaluated exactly as a linear convolution through a power-friendly FFT of size M >= N+P-1 (M = 576*R3,
// R3 in {4,8,16}: 9216 for N = 4096) that lives entirely in the 160 KiB LDS of one CU:
//     line samples -> periodic/reflected images written to LDS ->
//     in-place DIF stages radix 24, 24, R3 -> multiply by FFT_M(h) (digit-reversed table, 1/M folded in) ->
//     in-place inverse stages R3, 24, 24 -> the N wanted outputs go straight from registers to HBM.
// P = 4126 = 2*2063 forces Bluestein in a library FFT (two length-8192+ transforms per 1-D DFT, forward AND inverse);
// here one forward + one inverse length-9216 transform per line does the whole forward-chirp-inverse of that axis, so a
// propagation costs 2 passes x (8 B read + 8 B written) per pixel instead of 4 padded FFT passes.
//
// A two-pass separable transform with contiguous stores needs two transposes.  The first is a tiled pre-pass
// (k_source_transposed) that also evaluates the transmitted source wave (K1) -- once per call, for all distances; pass 1
// (k_fresnel_cols: lines along axis 0) then reads whole lines and writes row y of the intermediate [Ny][Nx]; pass 2
// (k_fresnel_rows: lines along axis 1) reads the intermediate's columns (the second transpose, 16-byte pieces) and
// writes the final image.  The distances of a call share each launch (work item = distance x line group).  Pass 1 shares more:
// its rounds are ONE line x TWO distances (DUAL) -- forward stages once, the middle stage writes the product with either
// distance's kernel spectrum into one of the two LDS line buffers, inverse stages at full width.  Lines too long for one
// transform (N > 4593) run as a partitioned convolution whose rounds couple the two LDS lines into ONE 18432-point transform
// (PAIR: even samples in line 0, odd in line 1, the radix-2 butterfly of the double-size transform in the middle stage).
//
// Each CU runs ONE persistent 16-wave workgroup: 12 engine waves own the butterflies (packed-fp32 arithmetic,
// fft_pk.hpp), 4 loader waves fetch the next line group from HBM during the transform and spread it into LDS while the
// engine finishes the last butterfly and its stores.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <tuple>
#include <vector>

#include "fresnel_p2.hpp"
#include "fresnel_plan.hpp"
#include "fresnel_stages.hpp"

using namespace psx;
using namespace psx::lines;

namespace {

// ---- lines that fit one LDS transform (N <= 4593) ------------------------------------------------------------------------------------
// CONTIG: the samples of a line are adjacent in memory (in_si == 1) -- the lanes of a loader wave then walk along the line;
// otherwise they walk across the LINES lines of the group (adjacent columns of a row-major image).
// DUAL (pass 1 of a call with several distances): a round is HALF the LDS lines' worth of image lines (one at R3 = 16) and
// TWO distances.  The lines are transformed forward once (stages A and B on the first half of the LDS lines, by six of the
// twelve engine waves); the middle stage reads each spectrum slab once and writes its product with the first distance's
// kernel spectrum back in place and the product with the second one's into the same slab of the second half of the LDS
// lines; the inverse stages then run at full width on all line buffers, the first half for one distance, the second half
// for the other.
// That is the only way of sharing the forward transform between distances that fits LDS: 4.3 stage-units of work per two
// (line, distance) results instead of 5.3, and half the samples to fetch and spread per round.
// phase timestamp of the oldest engine wave in round a.stamp_j of each workgroup (diagnostic runs only: psx_debug_stamps)
#define PSX_STAMP(k) PSX_STAMP_IF(k, tid == 0)
template <int R3, bool CONTIG, bool DUAL = false, bool QUEUE = false>
__global__ __launch_bounds__(T) void k_fresnel_lines(LineArgs a) {
    using GE = LineGeom<R3, false>;
    constexpr int M = GE::M, LINES = GE::LINES, S1 = GE::S1, MP = GE::MP;
    constexpr int LH = DUAL ? LINES / 2 : LINES;        // DUAL: image lines of a round = half the LDS lines (the other half
                                                        // receives the second distance's products)
    constexpr int LPG = LH;                             // image lines per round
    constexpr int LL = LH;                              // LDS lines the loaders fill
    static_assert(!DUAL || (CONTIG && LINES % 2 == 0), "DUAL: LINES/2 image lines x two distances per round");
    constexpr int SLAB = GE::SLAB, WSLABS = GE::WSLABS, TWB_LD = GE::TWB_LD;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const int N = a.N, mg = a.margin;
    // A fresh copy of the thread index for values that are re-derived inside the round loop instead of being kept across it
    // (the engine waves have no register to spare)
    auto ftid = [&]() __attribute__((always_inline)) {
        int t = tid;
        asm volatile("" : "+v"(t));
        return t;
    };

    // ---- line groups of this workgroup: XCD x = blockIdx % 8 owns a contiguous chunk of groups (its 32 CUs then read
    // neighbouring columns at the same time: the 128-byte lines of the strided source are shared in that XCD's L2)
    const int ngroups = (a.nlines + LPG - 1) / LPG;
    const int nwork = a.dist_inner ? ngroups : ngroups * a.n_dist;   // all distances of a call in ONE launch
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int cq = nwork >> 3, cr = nwork & 7;
    const int cstart = xcd * cq + (xcd < cr ? xcd : cr), clen = cq + (xcd < cr ? 1 : 0);
    const int nunits = slot < clen ? (clen - slot + nslot - 1) / nslot : 0;   // units cstart + slot + u*nslot, u < nunits
    const int nsub = DUAL ? (a.n_dist + 1) / 2 : a.n_dist;       // rounds per line group when the distances are taken inside
    const int nj = a.dist_inner ? nunits * nsub : nunits;   // rounds of this workgroup (static order)
    // One-transform passes can take their units from QUEUES instead (DYN).  A static share per workgroup assumes that all 256
    // workgroups start together: one CU busy with anything else (the copy kernels of an RCCL transfer, another stream) makes
    // one workgroup start when the first of the others ends and DOUBLES the pass (tools/contention_probe.py: one foreign
    // workgroup, +48 % on a position).  Here the share of workgroup s of an XCD -- units s, s + 32, ... of the XCD's chunk,
    // as in the static order -- is a queue of its own (one atomic counter): the owner claims from it, two units ahead of the
    // engine, and a workgroup whose own queue has run dry STEALS from the queues of the others (loader wave 0 looks at all
    // of them at once).  On a quiet GPU everybody finishes its own share at the same moment and finds nothing to steal; a
    // workgroup that starts late finds its share taken and leaves.  uq: ring of the units claimed (index inside the chunk,
    // -1: nothing left).
    constexpr bool DYN = QUEUE;                          // its own instantiations: the engine waves have no register to spare
    const int nsubr = a.dist_inner ? nsub : 1;                   // rounds per unit
    int *const uq = reinterpret_cast<int *>(lds + LINES * MP + (2 * R3 + RAD) * (RAD + 1));   // behind the three twiddle tables
    // round j -> (distance, line group)
    auto item = [&](int j, int &d, int &g) __attribute__((always_inline)) {
        if (a.dist_inner) {
            const int u = j / nsub;
            d = j - u * nsub;                                // DUAL: index of the distance PAIR
            g = cstart + (DYN ? uq[u & 3] : slot + u * nslot);
        } else {
            const int w = cstart + (DYN ? uq[j & 3] : slot + j * nslot);
            d = w / ngroups;
            g = w - d * ngroups;
        }
    };
    // is round j a round at all?  (uniform: every thread reads the same ring entry, written at least one barrier earlier)
    auto valid = [&](int j) __attribute__((always_inline)) { return DYN ? uq[(j / nsubr) & 3] >= 0 : j < nj; };
    // queue of slot v of this XCD: its counter (64 bytes apart) and the length of its share
    auto qcount = [&](int v) __attribute__((always_inline)) { return &a.queue[16 * (xcd + 8 * v)]; };
    auto share = [&](int v) __attribute__((always_inline)) { return v < clen ? (clen - v + nslot - 1) / nslot : 0; };
    // loader wave 0 as a whole: one unit from somebody else's queue, or -1.  `dry` is set when every queue was seen empty.
    bool dry = false, own_dry = false;
    const int myshare = share(slot);
    auto steal = [&]() __attribute__((always_inline)) {
        const int ln = tid & 63;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const int v = ln;                                       // lane v looks at queue v (nslot <= 32)
            bool avail = false;
            if (v < nslot && v != slot)
                avail = __hip_atomic_load(qcount(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)share(v);
            const unsigned long long m = __ballot(avail);
            if (m == 0ull) {
                dry = true;
                return -1;
            }
            // the first queue with something left after my own, cyclically (thieves spread over the victims)
            const unsigned long long hi = m >> ((slot + 1) & 63);
            const int pick = hi ? (slot + 1 + __builtin_ctzll(hi)) : __builtin_ctzll(m);
            unsigned k = 0u;
            if (ln == 0) k = atomicAdd(qcount(pick), 1u);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k < (unsigned)share(pick)) return pick + (int)k * nslot;
        }
        return -1;
    };

    // the first two units of this workgroup: one atomic, in flight while the twiddle tables are copied
    unsigned first2 = 0u;
    if (DYN && tid == TC) first2 = atomicAdd(qcount(slot), 2u);
    const TwTables tw = fill_tables<GE>(lds, a, tid);          // stage twiddles into LDS, behind the line buffers

    if constexpr (DYN) {
        if (tid == TC) {                                 // the first loader thread runs the queue
            uq[0] = first2 < (unsigned)myshare ? slot + (int)first2 * nslot : -1;
            uq[1] = first2 + 1u < (unsigned)myshare ? slot + (int)(first2 + 1u) * nslot : -1;
        }
        lds_barrier();                                   // the first two units are known to every wave
    }


    if (tid >= TC) {
        // =============================== loader waves =====================================================================
        // Thread lt owns the samples i0 + STEP*k (k < NLD) of ONE line.  STEP is a multiple of 32 (R3 >= 4), so the padded
        // LDS index of sample k is the index of sample 0 plus a compile-time offset.
        const int lt = tid - TC;
        constexpr int STEP = TL / LL, NLD = (TOT / LINES) * LL / (2 * TL), PSTEP = STEP + STEP / 32;
        static_assert(TL % LL == 0 && NLD * STEP >= (576 * R3 + 1) / 2, "sample ownership does not cover the longest line");
        constexpr bool AFFL = (STEP % 32 == 0);
        const int line = CONTIG ? lt / STEP : lt % LL, i0 = CONTIG ? lt % STEP : lt / LL;
        // mirror duty (np.pad 'reflect', EXP:237): thread t < LINES*2*mg re-reads one of the 2*mg samples next to an edge
        const int nmir = LL * 2 * mg;
        const int lm = lt % LL;
        int im = -1, jm = 0;
        if (lt < nmir) {
            const int r = lt / LL;
            im = r < mg ? r + 1 : N - 1 - 2 * mg + r;                   // 1..mg   |   N-1-mg..N-2
            jm = r < mg ? N + 2 * mg - 1 - im : 2 * N - 3 - im;         // left mirror | right mirror one period earlier
        }
        float2 *base = lds + line * MP;
        const int ja = i0 + N + 2 * mg - 1, jb = i0 - 1;                 // first period | one period earlier (i >= 1)
        const int oa = phys(ja), ob = phys(jb);   // jb = -1 (sample 0 has no earlier image) -> -2: affine, unused at k = 0
        const int64_t pstep = (int64_t)STEP * (a.in_blocked ? (int64_t)IB : a.in_si);

        float2 xs[NLD], xm = make_float2(0.f, 0.f);
        auto fetch = [&](int j) __attribute__((always_inline)) {        // issue every load of round j, wait for none
            int d, g;
            item(j, d, g);
            if (a.dist_inner && d != 0) return;                          // same line group as the round before: registers keep it
            const float2 *src = a.src[d];
            const int l0 = g * LPG;
            const bool line_ok = l0 + line < a.nlines;
            const int64_t pix0 = a.in_blocked
                                     ? ((int64_t)((l0 + line) / IB) * N + i0) * IB + (l0 + line) % IB
                                     : (int64_t)i0 * a.in_si + (int64_t)(l0 + line) * a.in_sl;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                // out-of-range samples re-read element 0 (always valid): unconditional loads issue back to back
                const bool ok = line_ok && i0 + STEP * k < N;
                xs[k] = src[ok ? pix0 + pstep * k : (int64_t)0];
            }
            const int64_t pixm = a.in_blocked ? ((int64_t)((l0 + lm) / IB) * N + im) * IB + (l0 + lm) % IB
                                              : (int64_t)im * a.in_si + (int64_t)(l0 + lm) * a.in_sl;
            xm = src[(im >= 0 && l0 + lm < a.nlines) ? pixm : (int64_t)0];
        };
        auto spread = [&](int j) __attribute__((always_inline)) {       // periodic / mirrored images -> LDS
            int d, g;
            item(j, d, g);
            const int l0 = g * LPG;
            const bool line_ok = l0 + line < a.nlines;
            for (int jz = a.L + lt; jz < M; jz += TL) {                  // zeros in [L, M) of every line (no division:
                const int pz = phys(jz);                                 // this burst is issue-bound on ONE wave per SIMD)
#pragma unroll
                for (int ln = 0; ln < LL; ++ln) lds[ln * MP + pz] = make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                if (i0 + STEP * k < N) {
                    const float2 x = line_ok ? xs[k] : make_float2(0.f, 0.f);
                    if (AFFL) {
                        base[oa + k * PSTEP] = x;
                        if (k > 0 || jb >= 0) base[ob + k * PSTEP] = x;
                    } else {
                        base[phys(ja + STEP * k)] = x;
                        if (k > 0 || jb >= 0) base[phys(jb + STEP * k)] = x;
                    }
                }
            }
            if (im >= 0) lds[lm * MP + phys(jm)] = l0 + lm < a.nlines ? xm : make_float2(0.f, 0.f);
            for (int t = lt + TL; t < nmir; t += TL) {                   // margins beyond TL/(2*LINES): rare
                const int ln = t % LL, r = t / LL;
                const int i2 = r < mg ? r + 1 : N - 1 - 2 * mg + r, j2 = r < mg ? N + 2 * mg - 1 - i2 : 2 * N - 3 - i2;
                float2 x2 = make_float2(0.f, 0.f);
                if (l0 + ln < a.nlines)
                    x2 = a.src[d][a.in_blocked ? ((int64_t)((l0 + ln) / IB) * N + i2) * IB + (l0 + ln) % IB
                                            : (int64_t)i2 * a.in_si + (int64_t)(l0 + ln) * a.in_sl];
                lds[ln * MP + phys(j2)] = x2;
            }
        };

        if (valid(0)) {
            fetch(0);
            spread(0);
        }
        lds_barrier();                                   // (0) first group is in LDS
        // ucur, usub: unit of round j (as the workgroup's u-th unit) and the round inside it -- counted, not divided
        int ucur = 0, usub = 0;
        auto ring_ok = [&](int u) __attribute__((always_inline)) { return uq[u & 3] >= 0; };
        for (int j = 0; DYN ? ring_ok(ucur) : j < nj; ++j) {
            const bool last_sub = usub + 1 == nsubr;
            const bool more = DYN ? ring_ok(last_sub ? ucur + 1 : ucur) : j + 1 < nj;
            lds_barrier();                               // (1) engine: forward stage A done
            if constexpr (DUAL) lds_barrier();           // (1b) engine: forward stage B done (six waves), before the middle stage
            // Issued after barrier (1), not before: issuing strided loads stalls for ~5 us (the texture path hands out one
            // 128-byte line per lane pair) and forward stage A lasts only 3 us -- the engine would wait for the loaders.
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 16] = wall_clock64();
            // the unit after next: claimed during the first round of a unit, by loader wave 0.  The atomic on the workgroup's own
            // queue goes out AHEAD of the round's loads and nobody waits for it here (memory returns in order: behind the loads
            // it would come back last and hold up the spread; consumed at once it would hold up the loads by a round trip); its
            // result is looked at AFTER the spread below -- visible to every wave after barrier (4), first read, as "is there a
            // round after the next", at the top of the next iteration.
            const bool qround = DYN && lt < 64 && usub == 0;                               // wave-uniform
            const bool claiming = qround && ring_ok(ucur + 1);
            unsigned mine = 0xffffffffu;
            if (claiming && !own_dry && lt == 0) mine = atomicAdd(qcount(slot), 1u);
            if (more) fetch(j + 1);
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 17] = wall_clock64();
            lds_barrier();                               // (2) engine: wave-private stages done
            lds_barrier();                               // (3) engine: inverse stage A holds all of LDS in registers
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 18] = wall_clock64();
            // the spread sits between two barriers the engine waits at: it goes first on its SIMD (a loader wave would
            // otherwise get every fourth issue slot).  The fetch keeps normal priority: hurrying the strided loads only
            // queues the engine's twiddle loads behind them.
            __builtin_amdgcn_s_setprio(3);
            if (more) spread(j + 1);
            __builtin_amdgcn_s_setprio(0);
            if (qround) {
                int unit = -1;
                if (claiming) {
                    const unsigned k = __builtin_amdgcn_readfirstlane(mine);
                    if (k < (unsigned)myshare) unit = slot + (int)k * nslot;
                    else own_dry = true;
                    if (unit < 0 && !dry) unit = steal();             // two round trips, the whole wave: own queue empty only
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
            // the last workgroup to leave re-arms the queues for the next launch (every claim of this launch has been made)
            unsigned prev = 0u;
            if (lt == 0) {
                __threadfence();
                prev = atomicAdd(&a.queue[16 * 256], 1u);
            }
            prev = __builtin_amdgcn_readfirstlane(prev);
            if (prev == gridDim.x - 1) {
                for (int w = lt; w < (int)gridDim.x; w += 64) atomicExch(&a.queue[16 * w], 0u);
                if (lt == 0) atomicExch(&a.queue[16 * 256], 0u);
            }
        }
        return;
    }

    // =================================== engine waves =========================================================================
    // stage A and B thread mapping (one butterfly per thread per stage)
    const int lineA = tid / S1, nA = tid % S1;
    // stage B: thread -> (line, block q1 of S1 points, element n < R3).  DUAL: in the inverse stage a wave takes blocks
    // {2w, 2w+1} of BOTH halves of the LDS lines (lanes 0-31 first half, 32-63 second half), so that the points it owns between
    // the barriers are the same range of the two buffers the middle stage wrote
    const int remB = tid % S1, q1B = remB / R3, nB = DUAL ? (tid & 31) % R3 : remB % R3, p0B = q1B * S1 + nB;
    v2f *baseA = reinterpret_cast<v2f *>(lds) + lineA * MP;
    auto stageB_at = [&](v2f *&bB, int &pB, bool paired) __attribute__((always_inline)) {
        bB = baseA;
        pB = p0B;
        if (paired) {
            int to = ftid();
            // a half-wave (32 butterflies = 768 points) takes the wave's range of the first LH lines (lanes 0-31) or of the
            // second LH lines (lanes 32-63); blocks are counted through the concatenated half
            const int gb = (to >> 6) * (32 / R3) + (to & 31) / R3;          // block inside the half: 24 blocks per line
            bB = reinterpret_cast<v2f *>(lds) + (gb / RAD + ((to & 63) >> 5) * LH) * MP;
            pB = (gb % RAD) * S1 + (to & 31) % R3;
        }
    };
    // middle stage: slab of this lane inside the wave's own 96 (round 2: 32 lanes).  Within each half-wave the first 16
    // lanes take the even slabs and the last 16 the odd ones: the 16 lanes of a ds_write_b64 group then carry 16
    // different pad offsets (one pad slot per TWO slabs) and hit 16 different bank pairs instead of 8.
    const int slabw = ((tid & 63) & 32) + ((tid & 31) < 16 ? 2 * (tid & 31) : 2 * ((tid & 31) - 16) + 1);
    const int slab0 = (tid >> 6) * WSLABS + slabw;
    const v2f *rowA1 = tw.tw1 + (nA / R3) * TWB_LD, *rowA0 = tw.tw0 + (nA % R3) * TWB_LD;
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 32 + 0] = wall_clock64();
    lds_barrier();                                       // (0) first group is in LDS
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 32 + 1] = wall_clock64();
    int eu = 0, es = 0;                                  // unit of round j and the round inside it (counted, not divided)
    for (int j = 0; DYN ? uq[eu & 3] >= 0 : j < nj; ++j) {
        int d, g;
        item(j, d, g);
        const int l0 = g * LPG;
        if (++es == nsubr) {
            es = 0;
            ++eu;
        }
        PSX_STAMP(2);

        // ---- 2. forward stage A: radix 24 over stride S1, twiddle w_M^{n q}  (DUAL: LDS line 0 only = engine waves 0..5)
        if (!DUAL || tid < TC / 2) fwd_stage_A<GE>(baseA, nA, rowA1, rowA0);
        // The kernel spectrum (the engine's only global loads) travels one step ahead of its use: the first slab's 128
        // bytes are requested here, before barrier (1); the second slab's right after the first one's multiply.
        float4 hh[SLAB / 2];
        // DUAL: lane u < 48 of wave w takes slab 48 w + u of the first half of the LDS lines
        const int tp = DUAL ? ftid() : tid;
        const int pslab = 48 * (tp >> 6) + (tp & 63);
        const bool pact = (tp & 63) < 48;
        if constexpr (DUAL) {
            // the whole slab of the FIRST distance's spectrum (16 points = 8 float4); the second one's comes under the arithmetic
            const float4 *h4 = reinterpret_cast<const float4 *>(a.H[2 * d] + ((pact ? pslab : 0) % (M / SLAB)) * SLAB);
#pragma unroll
            for (int q = 0; q < SLAB / 2; ++q) hh[q] = h4[q];
        } else {
            int sl = slab0;
            if constexpr (R3 < 16) {                 // re-derived here (as above): the hoisted 64-bit offset was spilled
                int to = ftid();
                sl = (to >> 6) * WSLABS + ((to & 63) & 32) + ((to & 31) < 16 ? 2 * (to & 31) : 2 * ((to & 31) - 16) + 1);
            }
            const float4 *h4 = reinterpret_cast<const float4 *>(a.H[d] + (unsigned)((sl % (M / SLAB)) * SLAB));
#pragma unroll
            for (int q = 0; q < SLAB / 2; ++q) hh[q] = h4[q];
        }
        PSX_STAMP(3);
        lds_barrier();                               // (1)
        PSX_STAMP(4);
        // ---- 3. forward stage B: radix 24 inside each block of S1, stride R3, twiddle w_S1^{n q}
        if (!DUAL || tid < TC / 2) {
            // DUAL: plain thread map (line 0), n = remB % R3
            fwd_stage_B<GE>(baseA, DUAL ? (remB / R3) * S1 + remB % R3 : p0B, tw.twl + (DUAL ? remB % R3 : nB) * TWB_LD);
        }
        if constexpr (DUAL) lds_barrier();           // (1b) the middle stage's slabs were written by other waves
        PSX_STAMP(5);
        // From here to the end of inverse stage B every wave works on LDS points that only IT touches: its 64
        // radix-24 butterflies of stage B cover 64/R3 whole blocks of S1 points = the 1536 consecutive points
        // [1536 w, 1536 (w+1)), and the middle stage takes its slabs from the same range.  A wave's LDS operations
        // execute in order, so no workgroup barrier is needed -- the waves drift apart and overlap each other's LDS
        // and VALU phases.
        wave_sync();
        PSX_STAMP(6);

        // ---- 4+5. middle stage, slab by slab: forward radix R3 on contiguous chunks, x FFT_M(h_d), inverse radix R3,
        // back to LDS.  Each thread rewrites exactly the slabs it read.
        if constexpr (DUAL) {
            // the slab index once more from an opaque copy of the thread index: kept live across forward stage B it is spilled
            // at R3 < 16 (and comes back behind a full memory wait)
            int tq = tid;
            if constexpr (R3 < 16) asm volatile("" : "+v"(tq));
            const int mslab = R3 < 16 ? 48 * (tq >> 6) + (tq & 63) : pslab;
            const bool mact = R3 < 16 ? (tq & 63) < 48 : pact;
            if (mact) {
                // slab mslab of the concatenated first LH lines; its second-distance product goes LH lines further
                const int sline = mslab / (M / SLAB), sp0 = (mslab % (M / SLAB)) * SLAB;
                v2f *b0 = reinterpret_cast<v2f *>(lds) + sline * MP + phys(sp0), *b1 = b0 + LH * MP;
                v2f x[SLAB], y[SLAB];
#pragma unroll
                for (int q = 0; q < SLAB; ++q) x[q] = lds_read(b0 + q);
                auto dft_chunks = [&](v2f(&f)[SLAB], auto inv_tag) __attribute__((always_inline)) {
#pragma unroll
                    for (int c = 0; c < SLAB / R3; ++c) {
                        v2f w[R3];
#pragma unroll
                        for (int q = 0; q < R3; ++q) w[q] = f[c * R3 + q];
                        DftPk<R3, decltype(inv_tag)::value>::run(w);
#pragma unroll
                        for (int q = 0; q < R3; ++q) f[c * R3 + q] = w[q];
                    }
                };
                dft_chunks(x, std::false_type{});                        // the spectrum slab, shared by the two distances
#pragma unroll
                for (int q = 0; q < SLAB / 2; ++q) {
                    y[2 * q] = pk_cmul(x[2 * q], (v2f){hh[q].x, hh[q].y});
                    y[2 * q + 1] = pk_cmul(x[2 * q + 1], (v2f){hh[q].z, hh[q].w});
                }
                // second distance's spectrum: its first half travels under the first distance's inverse DFT
                const float4 *h4b = reinterpret_cast<const float4 *>(a.H[2 * d + 1] + sp0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < SLAB / 4; ++q) hh[q] = h4b[q];
                __builtin_amdgcn_sched_barrier(0);
                dft_chunks(y, std::true_type{});
#pragma unroll
                for (int q = 0; q < SLAB; ++q) b0[q] = y[q];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < SLAB / 4; ++q) hh[SLAB / 4 + q] = h4b[SLAB / 4 + q];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < SLAB / 2; ++q) {
                    x[2 * q] = pk_cmul(x[2 * q], (v2f){hh[q].x, hh[q].y});
                    x[2 * q + 1] = pk_cmul(x[2 * q + 1], (v2f){hh[q].z, hh[q].w});
                }
                dft_chunks(x, std::true_type{});
#pragma unroll
                for (int q = 0; q < SLAB; ++q) b1[q] = x[q];
            }
        } else {
            middle_plain<GE>(lds, a.H[d], slabw, slab0, hh);
        }
        PSX_STAMP(7);
        wave_sync();
        PSX_STAMP(8);

        // ---- 6. inverse stage B: conjugate twiddle on the inputs, then the inverse radix-24 butterfly
        {
            v2f *bB;
            int pB;
            stageB_at(bB, pB, DUAL);
            inv_stage_B<GE>(bB, pB, tw.twl + (DUAL ? pB % R3 : nB) * TWB_LD);     // DUAL: pB = block * S1 + n, n < R3 | S1
        }
        PSX_STAMP(9);
        lds_barrier();                               // (2)
        PSX_STAMP(10);

        // ---- 7. inverse stage A; the wanted outputs y[n + P - 1] leave for HBM straight from the registers.  Once
        // every engine thread holds its 24 inputs LDS is free: the loaders fill it with the next group meanwhile.
        {
            v2f v[RAD];
#pragma unroll
            for (int q = 0; q < RAD; ++q) v[q] = baseA[GE::idxA(nA, q)];
            lds_barrier();                           // (3)
            PSX_STAMP(11);
            twiddle_A<1, 12, true>(v, rowA1, rowA0);
            twiddle_A<12, 24, true>(v, rowA1, rowA0);
            __builtin_amdgcn_sched_barrier(0);
            DftPk<RAD, true>::run(v);
            // output sample i = nA + q*S1 - jout (jout = LDS position of output sample 0).  The first index is made opaque
            // so that the 48 per-q addresses are formed here from ONE pointer, not hoisted out of the group loop (they
            // would occupy 96 VGPRs there and spill).
            // (R3 < 16: formed from an opaque copy of nA as well -- hoisted out of the group loop it has no register left and
            // comes back from scratch memory once a round, behind a full memory wait)
            int nAo = nA, lineAo = lineA;
            if constexpr (R3 < 16) {
                int to = ftid();
                nAo = to % S1;
                lineAo = to / S1;
            }
            int ifirst = nAo - (N + 2 * mg - 1);
            asm volatile("" : "+v"(ifirst));
            // DUAL: LDS line 0 carries the first distance of the pair, line 1 the second (a wave belongs to one line)
            const int dd = DUAL ? 2 * d + __builtin_amdgcn_readfirstlane(tid >= TC / 2 ? 1 : 0) : d;
            v2f *wo = reinterpret_cast<v2f *>(a.wave_out[dd]);
            float *io = a.inten_out[dd];
            const float sc = a.scale[dd];
            const v2f gp = (v2f){a.gph[dd].x, a.gph[dd].y};
            static_assert(S1 % IB == 0, "blocked output stride; the block index below is i >> IBS");
            if constexpr (S1 % 64 == 0) {
                // A wave's 64 butterflies belong to ONE line, so its outputs go through a buffer descriptor whose range is
                // exactly the window they may touch: the hardware drops the stores of the unwanted outputs (i < 0 wraps to
                // a huge offset, i >= N lies past the window).  No compare, no exec-mask bookkeeping per output -- the
                // scalar unit is shared by the whole CU (0.9 instructions per cycle, tools/salu_bench.hip) and the masked
                // form of this loop issued 500 scalar instructions per wave.
                const int l = l0 + __builtin_amdgcn_readfirstlane(DUAL ? lineAo % LH : lineAo);
                const bool lok = l < a.nlines;
                // element index of output i inside the window: plain rows: i (window = row l); blocked: ((i>>3)*nlines+l)*8 + i%8
                const int e0 = a.out_blocked ? ((ifirst >> IBS) * a.nlines + l) * IB + (ifirst & (IB - 1)) : ifirst;
                const int estep = a.out_blocked ? (S1 / IB) * a.nlines * IB : S1;
                const int64_t wbase = a.out_blocked ? 0 : (int64_t)l * a.out_ld;
                const int welems = lok ? (a.out_blocked ? ((N + IB - 1) / IB) * IB * a.nlines : N) : 0;
                store_window(v, wo, io, wbase, welems, e0, estep, gp, sc, a.accumulate);
            } else if (l0 + (DUAL ? lineAo % LH : lineAo) < a.nlines) {
                // short lines (R3 <= 4): a wave straddles lines, per-lane pointers and masks
                const int lsh = l0 + (DUAL ? lineAo % LH : lineAo);
                const int64_t ob = a.out_blocked ? ((int64_t)(ifirst >> IBS) * a.nlines + lsh) * IB + (ifirst & (IB - 1))
                                                 : (int64_t)lsh * a.out_ld + ifirst;
                const int64_t oq = a.out_blocked ? (int64_t)(S1 / IB) * a.nlines * IB : (int64_t)S1;
                if (wo) wo += ob;
                if (io) io += ob;
#pragma unroll
                for (int q = 0; q < RAD; ++q) {
                    const int i = ifirst + q * S1;
                    if (i >= 0 && i < N) {
                        if (wo) wo[q * oq] = pk_cmul_s(v[q], gp);
                        if (io) {
                            const float I = sc * (v[q].x * v[q].x + v[q].y * v[q].y);
                            io[q * oq] = a.accumulate ? io[q * oq] + I : I;
                        }
                    }
                }
            }
        }
        PSX_STAMP(12);
        lds_barrier();                               // (4)
        PSX_STAMP(13);
    }
}


#undef PSX_STAMP
#define PSX_STAMP(k) PSX_STAMP_IF(k, (DIF ? ftid() == 0 : tid == 0))
// ---- lines too long for one LDS transform (N > 4593): partitioned convolution, coupled lines, DIF rounds ---------------------------
// PAIR: the two LDS lines of a round hold the EVEN and the ODD samples of ONE sequence of 2M = 18432 points, and the middle
// stage couples them with the radix-2 butterfly of a 2M-point transform
//     X[k] = E[k] + w^k O[k],  X[k+M] = E[k] - w^k O[k]   ...x H...   E'[k] = Y[k] + Y[k+M],  O'[k] = (Y[k] - Y[k+M]) w^-k
// (E, O = the two M-point spectra the engine computes anyway; w = exp(-2 pi i / 2M)).  One round is then ONE block x segment
// product of twice the size: a 16384-sample line needs 2 x 2 of them instead of 5 x 3 M-point products for two lines.
// DIF (lines of 12 280 <= N <= 18 402 samples, the 16384^2 grid of BASELINE config 5): the whole line is ONE circular
// convolution of 4M = 36864 >= N + P - 1 points, split by a decimation-in-frequency radix-2 step over TWO PAIR rounds:
//     round E:  ye = IDFT_2M( FFT_2M( x[n] + x[n + 2M] )          * H4[2k]   )
//     round O:  yo = IDFT_2M( FFT_2M((x[n] - x[n + 2M]) w_4M^n )  * H4[2k+1] )        y[m] = (ye[m'] + w_4M^-m yo[m']) / 2,  m' = m mod 2M
// (the block x segment partition needs 2 x 2 such rounds at N = 16384, each re-reading a 16398-sample window).  The
// extension e of the line is P-periodic, so x[n + 2M] = e[n + 2M - P]: the loaders write ONE copy L[n] = e[n], n < 2M, and
// forward stage A forms L[n] +- L[n + D] itself from LDS (one more workgroup barrier: everybody reads before anybody writes
// in place); the twiddle w_4M^n = (thread factor) x (compile-time 48th root per butterfly leg) rides on stage A's input and
// output multiplies.  Round E leaves ye in a line buffer private to the workgroup (it is read back 20 us later by the same
// CU: L2 / MALL traffic), round O combines and stores the N wanted samples, whose index wraps once along the 24 outputs of a
// butterfly.  Per line: 2 rounds and 2 x 18432 loads instead of 4 and 4 x 16398; no partial sums in the output image.
template <bool CONTIG, bool PAIR, bool DIF = false>
__global__ __launch_bounds__(T) void k_fresnel_part(LineArgs a) {
    static_assert(!DIF || PAIR, "DIF rounds are PAIR rounds");
    constexpr int R3 = 16;
    using GE = LineGeom<R3, PAIR>;
    constexpr int M = GE::M, LINES = GE::LINES, S1 = GE::S1, MP = GE::MP;
    static_assert(LINES == 2, "the partitioned engine works on the two LDS lines of the M = 9216 transform");
    constexpr int LPG = PAIR ? 1 : LINES;               // image lines per round
    constexpr int SLAB = GE::SLAB, WSLABS = GE::WSLABS, TWB_LD = GE::TWB_LD;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const int N = a.N, mg = a.margin;
    // A fresh copy of the thread index for values that are re-derived inside the round loop instead of being kept across it
    // (the engine waves have no register to spare).  DIF: not even the index itself stays in a VGPR -- the wave's base sits in
    // an SGPR and the lane number comes from v_mbcnt (it was the one value the DIF instance spilled: reloaded once a round
    // from scratch memory, behind a full memory wait).
    const int wave_base = DIF ? __builtin_amdgcn_readfirstlane(tid & ~63) : 0;
    auto ftid = [&]() __attribute__((always_inline)) {
        int t;
        if constexpr (DIF) {
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(t));
            t += wave_base;
        } else {
            t = tid;
            asm volatile("" : "+v"(t));
        }
        return t;
    };

    // ---- work units of this workgroup: (distance, output block, line group), S consecutive rounds each (one per kernel
    // segment).  XCD x = blockIdx % 8 owns a contiguous chunk of units (its 32 CUs then read neighbouring columns at the same
    // time: the 128-byte lines of the strided source are shared in that XCD's L2); static shares.
    // a.dist_inner (pass 1: every distance reads the SAME source): the chunks are cut over (block, line group) pairs and a
    // workgroup takes the n_dist distances of its pair in consecutive units -- the DIF loaders then keep most of the line in
    // registers for all of them (below) instead of fetching it 2 n_dist times.
    const int ngroups = (a.nlines + LPG - 1) / LPG;
    const int ndin = a.dist_inner ? a.n_dist : 1;                             // distances inside a chunk item
    const int nwork = ngroups * a.NB * (a.dist_inner ? 1 : a.n_dist);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int cq = nwork >> 3, cr = nwork & 7;
    const int cstart = xcd * cq + (xcd < cr ? xcd : cr), clen = cq + (xcd < cr ? 1 : 0);
    const int nunits = (slot < clen ? (clen - slot + nslot - 1) / nslot : 0) * ndin;   // chunk items cstart + slot + k*nslot, x distances inside
    const int nj = nunits * a.S;                                              // rounds of this workgroup
    // round j -> (distance, line group), output block pb, kernel segment ps
    int pb = 0, ps = 0;
    auto item = [&](int j, int &d, int &g) __attribute__((always_inline)) {
        const int u = j / a.S;
        ps = j - u * a.S;
        const int ui = u / ndin, di = u - ui * ndin;     // chunk item of this workgroup, distance inside it
        const int w = cstart + slot + ui * nslot;        // item: ((d * NB) + b) * ngroups + g, or (b * ngroups + g) with d = di
        const int db = w / ngroups;
        g = w - db * ngroups;
        if (a.dist_inner) {
            d = di;
            pb = db;
        } else {
            d = db / a.NB;
            pb = db - d * a.NB;
        }
    };

    const TwTables tw = fill_tables<GE>(lds, a, tid);          // stage twiddles into LDS, behind the line buffers

    if (tid >= TC) {
        // =========================== loader waves, partitioned convolution ============================================
        // LDS position t of a line holds e[a0 + t], e = the periodic / mirrored extension of the line the linear
        // convolution runs over (e[te] = x_per[te - (P-1) + margin]), a0 = b*B + P - (s+1)*Lh; zeros outside the window of
        // B + Lh - 1 positions.  A window is twice a regular line's share of registers, so it moves in two halves: the
        // first is fetched during the transform (as in the regular engine), the second between barriers (3) and (4), after
        // the first has been written out of the same registers.
        const int lt = tid - TC;
        constexpr int STEP = TL / LINES, NH = M / STEP / 2, PSTEP = STEP + STEP / 32;
        static_assert(STEP % 32 == 0 && (M / STEP) % 2 == 0, "affine LDS addressing of the loader halves");
        // (DIF: adjacent lanes take adjacent samples whatever the source layout -- 512 contiguous bytes per wave load in pass 1)
        const int line = (CONTIG && !DIF) ? lt / STEP : lt % LINES, i0 = (CONTIG && !DIF) ? lt % STEP : lt / LINES;
        float2 *base = lds + line * MP + phys(i0);
        const int P = a.P, Lw = a.B + a.Lh - 1, Etot = N + P - 1;
        // DIF: every one of the 2M window positions holds a sample (no validity masks), position t is sample
        // reflect(((t + 1 + mg) mod P) - mg) of the line: five vector instructions and a buffer load whose descriptor is the
        // line.  The 72 positions of a thread then move as NHA + NHB instead of 36 + 36: the more of them travel during the
        // transform, the shorter the fetch that is exposed between barriers (3) and (4).
        constexpr int NHA = DIF ? PSX_DIF_NHA : NH, NHB = 2 * NH - NHA;
        float2 xs[NHA];
        auto fetch_dif = [&](int j, auto k0_tag, auto cnt_tag) __attribute__((always_inline)) {
            constexpr int K0 = decltype(k0_tag)::value, CNT = decltype(cnt_tag)::value;
            int d, g;
            item(j, d, g);
            const int lc = min(g, a.nlines - 1);
            const float2 *srcl = a.src[d] + (a.in_blocked ? ((int64_t)(lc / IB) * N) * IB + lc % IB : (int64_t)lc * a.in_sl);
            const int sh = a.in_blocked ? BLOCKED_SAMPLE_SHIFT : SAMPLE_SHIFT;                     // byte stride of a sample: 8 IB (blocked) or 8
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float2 *>(srcl), 0, g < a.nlines ? (int)((unsigned)N << sh) : 0, 0x00020000);   // a line past the image reads zeros
            int jb = 2 * i0 + line + mg + 1;
            asm volatile("" : "+v"(jb));
#pragma unroll
            for (int k = 0; k < CNT; ++k) {
                const unsigned j0 = (unsigned)(jb + 2 * STEP * (K0 + k));
                const unsigned jp = min(j0, j0 - (unsigned)P);                   // mod P (j0 < 2P)
                unsigned i1, i2;       // |a - b| in one instruction (the compiler expands __sad into sub, neg, max)
                asm("v_sad_u32 %0, %1, %2, 0" : "=v"(i1) : "v"(jp), "s"(mg));           // np.pad 'reflect' (EXP:237) on the left ...
                asm("v_sad_u32 %0, %1, %2, 0" : "=v"(i2) : "v"(i1), "s"(N - 1));
                const unsigned i = (unsigned)(N - 1) - i2;                                 // ... and on the right
                xs[k] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(i << sh), 0, 0));
            }
        };
        auto spread_dif = [&](auto k0_tag, auto cnt_tag) __attribute__((always_inline)) {
            constexpr int K0 = decltype(k0_tag)::value, CNT = decltype(cnt_tag)::value;
#pragma unroll
            for (int k = 0; k < CNT; ++k) base[(K0 + k) * PSTEP] = xs[k];
        };
        using KA0 = std::integral_constant<int, 0>;
        using KAN = std::integral_constant<int, NHA>;
        using KBN = std::integral_constant<int, NHB>;
        unsigned vm0 = 0u, vm1 = 0u;                    // which of the NH positions hold a sample (the rest are zeros)
        static_assert(NH <= 64, "validity mask");
        auto fetch_half = [&](int j, int h) __attribute__((always_inline)) {
            int d, g;
            item(j, d, g);
            const int l = PAIR ? g : g * LINES + line;
            const int lc = min(l, a.nlines - 1);          // addresses stay inside the image for the idle lines of the last group
            // PART sources are the blocked intermediate or contiguous lines (in_si == 1): 32-bit element offsets from the line's base
            const float2 *srcl = a.src[d] + (a.in_blocked ? ((int64_t)(lc / IB) * N) * IB + lc % IB : (int64_t)lc * a.in_sl);
            const int istep = a.in_blocked ? IB : 1;
            // DIF: the window is the first 2M points of the extension, every position holds a sample
            const int a0 = DIF ? 0 : pb * a.B + P - (ps + 1) * a.Lh;
            const unsigned tlim = l < a.nlines ? (unsigned)(DIF ? 2 * M : Lw) : 0u;
            int tb = PAIR ? 2 * i0 + line : i0;          // window position of this thread's first sample; opaque, so that the
            asm volatile("" : "+v"(tb));                 // 72 positions are formed here and not kept across the rounds
            vm0 = 0u;
            vm1 = 0u;
#pragma unroll
            for (int k = 0; k < NH; ++k) {
                // PAIR: LDS line `line` holds the samples of parity `line`, LDS index = window position / 2
                const int t = tb + (PAIR ? 2 : 1) * STEP * (k + NH * h);
                const int te = a0 + t;
                const bool ok = (unsigned)t < tlim && (unsigned)te < (unsigned)Etot;
                int jp = te - (P - 1) + mg;                  // index into the padded line, one period either side
                jp += (jp >> 31) & P;
                int i = abs(jp - mg);                        // np.pad 'reflect' (EXP:237): -r on the left ...
                i = i >= N ? 2 * N - 2 - i : i;              // ... 2N-2-r on the right
                i = ok ? i : 0;                              // unconditional loads issue back to back
                xs[k] = srcl[(unsigned)(i * istep)];
                if (k < 32) vm0 |= (ok ? 1u : 0u) << (k & 31);
                else vm1 |= (ok ? 1u : 0u) << (k & 31);
            }
        };
        auto spread_half = [&](int h) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < NH; ++k) {
                const bool ok = ((k < 32 ? vm0 : vm1) >> (k & 31)) & 1u;
                base[(k + NH * h) * PSTEP] = ok ? xs[k] : make_float2(0.f, 0.f);
            }
        };
        if constexpr (DIF) {
            // DIF rounds, round 4: most of a line STAYS in the loaders' registers.  The two rounds of a line (and, in pass 1,
            // the rounds of all distances of a line: dist_inner order) spread the SAME 2M samples; only what is formed from them
            // differs (L[n] + L[n + D] or (L[n] - L[n + D]) w^n, by the engine).  A thread's 72 positions are therefore split
            // into NK that are fetched once per line and kept (2 NK registers) and two chunks of NR that rotate through one
            // small buffer every round: chunk A travels during the transform, chunk B between barriers (3) and (4).  Stamps of
            // the previous form (all 72 fetched every round, 60 + 12): the loaders needed 13-18 us to ISSUE a round's loads --
            // pass 2 reads one 64-byte sector per 8-byte sample and is bound by the L2 -> L1 path, pass 1 by HBM -- and the engine
            // waited for them at barrier (2) for 2.7-9.3 us of every round (gpurun_out/r4s10).
            constexpr int NK = PSX_DIF_KEEP, NR = (2 * NH - NK) / 2;
            static_assert(NK + 2 * NR == 2 * NH && NK >= 0 && NR >= 1, "kept + 2 rotating chunks = the 72 positions of a loader thread");
            float2 xk[NK > 0 ? NK : 1], xr[NR];
            auto fetch_pos = [&](int j, auto k0_tag, auto cnt_tag, auto &dst) __attribute__((always_inline)) {
                constexpr int K0 = decltype(k0_tag)::value, CNT = decltype(cnt_tag)::value;
                int d, g;
                item(j, d, g);
                const int lc = min(g, a.nlines - 1);
                const float2 *srcl = a.src[d] + (a.in_blocked ? ((int64_t)(lc / IB) * N) * IB + lc % IB : (int64_t)lc * a.in_sl);
                const int sh = a.in_blocked ? BLOCKED_SAMPLE_SHIFT : SAMPLE_SHIFT;                     // byte stride of a sample: 8 IB (blocked) or 8
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float2 *>(srcl), 0, g < a.nlines ? (int)((unsigned)N << sh) : 0, 0x00020000);   // a line past the image reads zeros
                int jb = 2 * i0 + line + mg + 1;
                asm volatile("" : "+v"(jb));
#pragma unroll
                for (int k = 0; k < CNT; ++k) {
                    const unsigned j0 = (unsigned)(jb + 2 * STEP * (K0 + k));
                    const unsigned jp = min(j0, j0 - (unsigned)P);                   // mod P (j0 < 2P)
                    unsigned i1, i2;       // |a - b| in one instruction (the compiler expands __sad into sub, neg, max)
                    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(i1) : "v"(jp), "s"(mg));           // np.pad 'reflect' (EXP:237) on the left ...
                    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(i2) : "v"(i1), "s"(N - 1));
                    const unsigned i = (unsigned)(N - 1) - i2;                                 // ... and on the right
                    dst[k] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(i << sh), 0, 0));
                }
            };
            auto spread_pos = [&](auto k0_tag, auto cnt_tag, const auto &srcv) __attribute__((always_inline)) {
                constexpr int K0 = decltype(k0_tag)::value, CNT = decltype(cnt_tag)::value;
#pragma unroll
                for (int k = 0; k < CNT; ++k) base[(K0 + k) * PSTEP] = srcv[k];
            };
            using K0 = std::integral_constant<int, 0>;
            using KK = std::integral_constant<int, NK>;
            using KR = std::integral_constant<int, NR>;
            using KB = std::integral_constant<int, NK + NR>;
            // which line the kept registers hold: (source plane, line); a round of another line re-fetches them
            const float2 *ksrc = nullptr;
            int kg = -1;
            auto line_of = [&](int j, const float2 *&sp, int &g) __attribute__((always_inline)) {
                int d;
                item(j, d, g);
                sp = a.src[d];
            };
            if (nj > 0) {
                if constexpr (NK > 0) {
                    fetch_pos(0, K0{}, KK{}, xk);
                    spread_pos(K0{}, KK{}, xk);
                    line_of(0, ksrc, kg);
                }
                fetch_pos(0, KK{}, KR{}, xr);
                spread_pos(KK{}, KR{}, xr);
                fetch_pos(0, KB{}, KR{}, xr);
                spread_pos(KB{}, KR{}, xr);
            }
            lds_barrier();                                   // (0)
            for (int j = 0; j < nj; ++j) {
                const bool more = j + 1 < nj;
                lds_barrier();                               // (1a) engine: stage A has read its inputs and their partners
                lds_barrier();                               // (1)
                if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 16] = wall_clock64();
                if (more) {
                    if constexpr (NK > 0) {
                        const float2 *sp;
                        int g;
                        line_of(j + 1, sp, g);
                        if (sp != ksrc || g != kg) {         // uniform: a new line
                            fetch_pos(j + 1, K0{}, KK{}, xk);
                            ksrc = sp;
                            kg = g;
                        }
                    }
                    fetch_pos(j + 1, KK{}, KR{}, xr);
                }
                if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 17] = wall_clock64();
                lds_barrier();                               // (2)
                lds_barrier();                               // (3)
                if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 18] = wall_clock64();
                __builtin_amdgcn_s_setprio(3);
                if (more) {
                    if constexpr (NK > 0) spread_pos(K0{}, KK{}, xk);
                    spread_pos(KK{}, KR{}, xr);
                    if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 21] = wall_clock64();
                    fetch_pos(j + 1, KB{}, KR{}, xr);
                    if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 22] = wall_clock64();
                    spread_pos(KB{}, KR{}, xr);
                }
                __builtin_amdgcn_s_setprio(0);
                if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 19] = wall_clock64();
                lds_barrier();                               // (4)
                if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 20] = wall_clock64();
            }
            return;
        }
        if (nj > 0) {
            if constexpr (DIF) {
                fetch_dif(0, KA0{}, KAN{});
                spread_dif(KA0{}, KAN{});
                fetch_dif(0, KAN{}, KBN{});
                spread_dif(KAN{}, KBN{});
            } else {
                fetch_half(0, 0);
                spread_half(0);
                fetch_half(0, 1);
                spread_half(1);
            }
        }
        lds_barrier();                                   // (0)
        for (int j = 0; j < nj; ++j) {
            const bool more = j + 1 < nj;
            if constexpr (DIF) lds_barrier();            // (1a) engine: stage A has read its inputs and their partners
            lds_barrier();                               // (1)
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 16] = wall_clock64();
            if (more) {
                if constexpr (DIF) fetch_dif(j + 1, KA0{}, KAN{});
                else fetch_half(j + 1, 0);
            }
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 17] = wall_clock64();
            lds_barrier();                               // (2)
            lds_barrier();                               // (3)
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 18] = wall_clock64();
            __builtin_amdgcn_s_setprio(3);
            if (more) {
                if constexpr (DIF) spread_dif(KA0{}, KAN{});
                else spread_half(0);
                if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 21] = wall_clock64();
                if constexpr (DIF) fetch_dif(j + 1, KAN{}, KBN{});
                else fetch_half(j + 1, 1);
                if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 22] = wall_clock64();
                if constexpr (DIF) spread_dif(KAN{}, KBN{});
                else spread_half(1);
            }
            __builtin_amdgcn_s_setprio(0);
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 19] = wall_clock64();
            lds_barrier();                               // (4)
            if (a.stamps && lt == 0 && j == a.stamp_j) a.stamps[(size_t)blockIdx.x * 32 + 20] = wall_clock64();
        }
        return;
    }

    // =================================== engine waves =========================================================================
    // stage A and B thread mapping (one butterfly per thread per stage)
    // PAIR: adjacent lanes take butterfly n of the even line and of the odd line -- their outputs are adjacent samples, so a
    // wave's stores (and its partial-sum loads) cover whole cache lines
    const int lineA = PAIR ? (tid & 1) : tid / S1, nA = PAIR ? (tid >> 1) : tid % S1;
    // stage B: thread -> (line, block q1 of S1 points, element n < R3).  PAIR: a wave takes blocks {2w, 2w+1} of BOTH lines
    // (lanes 0-31 line 0, 32-63 line 1), so that the points it owns between the barriers are the same range of the two lines
    // the middle stage couples
    // (the PAIR values are re-derived inside the round loop from an opaque copy of the thread index: hoisted out of it
    // they would stay live through inverse stage A, which has no register to spare)
    const int remB = tid % S1, q1B = remB / R3, nB = PAIR ? (tid & 31) % R3 : remB % R3, p0B = q1B * S1 + nB;
    v2f *baseA = reinterpret_cast<v2f *>(lds) + lineA * MP;
    auto stageB_at = [&](v2f *&bB, int &pB, bool paired) __attribute__((always_inline)) {
        bB = baseA;
        pB = p0B;
        if (paired) {
            int to = ftid();
            // a half-wave (32 butterflies = 768 points) takes the wave's range of line 0 (lanes 0-31) or of line 1 (lanes 32-63)
            const int gb = (to >> 6) * (32 / R3) + (to & 31) / R3;          // block inside the half: 24 blocks per line
            bB = reinterpret_cast<v2f *>(lds) + (gb / RAD + ((to & 63) >> 5)) * MP;
            pB = (gb % RAD) * S1 + (to & 31) % R3;
        }
    };
    // middle stage: slab of this lane inside the wave's own 96 (round 2: 32 lanes).  Within each half-wave the first 16
    // lanes take the even slabs and the last 16 the odd ones: the 16 lanes of a ds_write_b64 group then carry 16
    // different pad offsets (one pad slot per TWO slabs) and hit 16 different bank pairs instead of 8.
    const int slabw = ((tid & 63) & 32) + ((tid & 31) < 16 ? 2 * (tid & 31) : 2 * ((tid & 31) - 16) + 1);
    const int slab0 = (tid >> 6) * WSLABS + slabw;
    const v2f *rowA1 = tw.tw1 + (nA / R3) * TWB_LD, *rowA0 = tw.tw0 + (nA % R3) * TWB_LD;
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 32 + 0] = wall_clock64();
    lds_barrier();                                       // (0) first group is in LDS
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 32 + 1] = wall_clock64();
    for (int j = 0; j < nj; ++j) {
        int d, g;
        item(j, d, g);
        const int l0 = g * LPG;
        PSX_STAMP(2);

        // ---- 2. forward stage A: radix 24 over stride S1, twiddle w_M^{n q}
        if constexpr (DIF) {
            // leg q of this thread's butterfly is position n = n0 + 768 q of the 2M-point sequence, n0 = 2 nA + lineA = tid; its
            // partner x[n + 2M] = L[n + D] exists for n < thr.  Round E transforms L[n] + L[n + D], round O (L[n] - L[n + D]) w_4M^n
            // with w_4M^n = conj(w4[n0]) x exp(-2 pi i q / 48), the second factor a compile-time constant.
            v2f v[RAD];
            int to = ftid();
            const int np = to + a.dsh;
            const v2f *pp = reinterpret_cast<const v2f *>(lds) + (np & 1) * MP + phys(np >> 1);
            const int qb = (a.thr - to + 2 * S1 - 1) / (2 * S1);          // legs q < qb have a partner
            const float sg = ps == 0 ? 1.f : -1.f;                       // wave-uniform
#pragma unroll
            for (int q = 0; q < RAD; ++q) v[q] = baseA[GE::idxA(nA, q)];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v2f b[RAD / 2];
#pragma unroll
                for (int q = 0; q < RAD / 2; ++q) b[q] = lds_read(pp + (h * (RAD / 2) + q) * (S1 + S1 / 32));
#pragma unroll
                for (int q = 0; q < RAD / 2; ++q) {
                    // A leg without partner (q >= qb) reads whatever lies at its would-be partner's address and drops it.  That
                    // address can lie behind the twiddle tables, and for short DIF lines (large D = 2M - P) past the end of
                    // the workgroup's LDS allocation: furthest byte = 8 * (MP + phys((767 + D) / 2) + 23 * 396) <= 184 KB for
                    // P >= 12 310.  An out-of-range DS read is defined on gfx9: it returns 0 and touches nothing (the LDS
                    // aperture check; it does set MEM_VIOL in TRAPSTS, which only matters under a trap handler).  Clamping
                    // the 24 addresses instead costs 24 vector instructions in the stage that has none to spare.
                    // tests/test_host_cpu.py::test_dif_stage_a_partner_reads_stay_inside_the_lines restates this arithmetic for
                    // every line length the DIF rounds take and pins both facts: a leg whose value is used reads inside the line
                    // buffers; the furthest dropped read is 177 440 bytes (round 5).
                    const v2f bm = (h * (RAD / 2) + q) < qb ? b[q] : (v2f){0.f, 0.f};
                    v[h * (RAD / 2) + q] = pk_fma_k(bm, sg, v[h * (RAD / 2) + q]);
                }
            }
            lds_barrier();                           // (1a) every input and partner has been read: the in-place writes may start
            if (ps != 0) {
                // both factors on the inputs: the thread factor would otherwise stay live through the butterfly.  (Requesting
                // it ahead of the LDS reads above, so that its latency is not exposed behind barrier (1a), changes nothing:
                // 16384^2 passes 13.11 / 11.07 ms against 13.17 / 10.99, round 4.)
                const float2 wf = a.w4[to];
                const v2f wb = (v2f){wf.x, wf.y};
                pk_static_for<0, RAD>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = decltype(qc)::value;
                    v[q] = pk_cmulc(v[q], pk_twiddle<2 * RAD, q, true>(wb));      // x conj(w4[n0] exp(+2 pi i q / 48)) = w_4M^n
                });
            }
            DftPk<RAD, false>::run(v);
            __builtin_amdgcn_sched_barrier(0);
            twiddle_A<1, 12, false>(v, rowA1, rowA0);
            twiddle_A<12, 24, false>(v, rowA1, rowA0);
#pragma unroll
            for (int q = 0; q < RAD; ++q) baseA[GE::idxA(nA, q)] = v[q];
        } else {
            fwd_stage_A<GE>(baseA, nA, rowA1, rowA0);
        }
        // The kernel spectrum (the engine's only global loads) travels one step ahead of its use: the first slab's 128
        // bytes are requested here, before barrier (1); the second slab's right after the first one's multiply.
        float4 hh[SLAB / 2];
        // PAIR: lane u < 48 of wave w couples slab 48 w + u of line 0 with the same slab of line 1; its table is the interleaved
        // pair (H[k], H[k + M]) per point: 4 points = 4 float4 per chunk; the first chunk travels here, the others under the
        // arithmetic of the chunk before (a slab pair already holds 64 registers of data)
        const int tp = PAIR ? ftid() : tid;
        const int pslab = 48 * (tp >> 6) + (tp & 63);                         // slab index inside a line (PAIR)
        const bool pact = (tp & 63) < 48;
        const float4 *hp4 = reinterpret_cast<const float4 *>(a.H[d] + (size_t)ps * 2 * M) + (size_t)(pact ? pslab : 0) * SLAB;
        v2f w0p = (v2f){1.f, 0.f};
        if constexpr (PAIR) {
#pragma unroll
            for (int q = 0; q < 4; ++q) hh[q] = hp4[q];
            const float2 w = a.w2[pact ? pslab : 0];
            w0p = (v2f){w.x, w.y};
        } else {
            const int sl = slab0;
            const float4 *h4 = reinterpret_cast<const float4 *>(a.H[d] + ps * M + (unsigned)((sl % (M / SLAB)) * SLAB));
#pragma unroll
            for (int q = 0; q < SLAB / 2; ++q) hh[q] = h4[q];
        }
        PSX_STAMP(3);
        lds_barrier();                               // (1)
        PSX_STAMP(4);
        // ---- 3. forward stage B: radix 24 inside each block of S1, stride R3, twiddle w_S1^{n q}
        {
            v2f *bB;
            int pB;
            stageB_at(bB, pB, PAIR);
            fwd_stage_B<GE>(bB, pB, tw.twl + nB * TWB_LD);
        }
        PSX_STAMP(5);
        // From here to the end of inverse stage B every wave works on LDS points that only IT touches: its 64
        // radix-24 butterflies of stage B cover 64/R3 whole blocks of S1 points = the 1536 consecutive points
        // [1536 w, 1536 (w+1)), and the middle stage takes its slabs from the same range.  A wave's LDS operations
        // execute in order, so no workgroup barrier is needed -- the waves drift apart and overlap each other's LDS
        // and VALU phases.
        wave_sync();
        PSX_STAMP(6);

        // ---- 4+5. middle stage, slab by slab: forward radix R3 on contiguous chunks, x FFT_M(h_d), inverse radix R3,
        // back to LDS.  Each thread rewrites exactly the slabs it read.
        if constexpr (PAIR) {
            if (pact) {
                v2f *b0 = reinterpret_cast<v2f *>(lds) + phys(pslab * SLAB), *b1 = b0 + MP;
                v2f f0[SLAB], f1[SLAB];
#pragma unroll
                for (int q = 0; q < SLAB; ++q) f0[q] = lds_read(b0 + q);
#pragma unroll
                for (int q = 0; q < SLAB; ++q) f1[q] = lds_read(b1 + q);
                DftPk<SLAB, false>::run(f0);
                DftPk<SLAB, false>::run(f1);
                // radix-2 of the 2M-point transform around the product with the kernel spectrum, 4 points per table chunk: chunk
                // c + 1 is requested (into the other half of hh) before chunk c is used
                pk_static_for<0, 4>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr (c < 3) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) hh[4 * ((c + 1) & 1) + q] = hp4[4 * (c + 1) + q];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    pk_static_for<0, 4>([&](auto qc) __attribute__((always_inline)) {
                        constexpr int qq = decltype(qc)::value, q = 4 * c + qq, hq = 4 * (c & 1) + qq;
                        const v2f wq = pk_twiddle<32, q, false>(w0p);       // w^k = w^{k0} exp(-2 pi i q3 / 32), q3 = q
                        const v2f t = pk_cmul(f1[q], wq);
                        const v2f x0 = f0[q] + t, x1 = f0[q] - t;
                        const v2f y0 = pk_cmul(x0, (v2f){hh[hq].x, hh[hq].y}), y1 = pk_cmul(x1, (v2f){hh[hq].z, hh[hq].w});
                        f0[q] = y0 + y1;
                        f1[q] = pk_cmulc(y0 - y1, wq);
                    });
                });
                DftPk<SLAB, true>::run(f0);
#pragma unroll
                for (int q = 0; q < SLAB; ++q) b0[q] = f0[q];
                DftPk<SLAB, true>::run(f1);
#pragma unroll
                for (int q = 0; q < SLAB; ++q) b1[q] = f1[q];
            }
        } else {
            middle_plain<GE>(lds, a.H[d] + ps * M, slabw, slab0, hh);
        }
        PSX_STAMP(7);
        wave_sync();
        PSX_STAMP(8);

        // ---- 6. inverse stage B: conjugate twiddle on the inputs, then the inverse radix-24 butterfly
        {
            v2f *bB;
            int pB;
            stageB_at(bB, pB, PAIR);
            inv_stage_B<GE>(bB, pB, tw.twl + nB * TWB_LD);
        }
        PSX_STAMP(9);
        lds_barrier();                               // (2)
        PSX_STAMP(10);

        // ---- 7. inverse stage A; the wanted outputs y[n + P - 1] leave for HBM straight from the registers.  Once
        // every engine thread holds its 24 inputs LDS is free: the loaders fill it with the next group meanwhile.
        {
            v2f v[RAD];
#pragma unroll
            for (int q = 0; q < RAD; ++q) v[q] = baseA[GE::idxA(nA, q)];
            // DIF, round O: the thread factor of the recombination twiddle is requested here, ahead of the loaders' second
            // fetch (the CU's loads return in order: behind that fetch it would come back microseconds later)
            v2f wdif = (v2f){1.f, 0.f};
            if constexpr (DIF) {
                if (ps != 0) {
                    int tq = ftid();
                    const float2 wf = a.w4[tq];
                    wdif = (v2f){wf.x, wf.y};
                }
            }
            lds_barrier();                           // (3)
            PSX_STAMP(11);
            twiddle_A<1, 12, true>(v, rowA1, rowA0);
            twiddle_A<12, 24, true>(v, rowA1, rowA0);
            __builtin_amdgcn_sched_barrier(0);
            DftPk<RAD, true>::run(v);
            // output sample i = nA + q*S1 - jout (jout = LDS position of output sample 0).  The first index is made opaque
            // so that the 48 per-q addresses are formed here from ONE pointer, not hoisted out of the group loop (they
            // would occupy 96 VGPRs there and spill).
            // PART: index inside the output block; PAIR: LDS line = parity of the position in the 2M-point result
            int ifirst = (PAIR ? 2 * nA + lineA : nA) - (a.Lh - 1);

            asm volatile("" : "+v"(ifirst));
            const int dd = d;
            v2f *wo = reinterpret_cast<v2f *>(a.wave_out[dd]);
            float *io = a.inten_out[dd];
            const float sc = a.scale[dd];
            const v2f gp = (v2f){a.gph[dd].x, a.gph[dd].y};
            static_assert(S1 % IB == 0, "blocked output stride; the block index below is i >> IBS");
            if constexpr (DIF) {
                // leg q holds point n' = n0 + 768 q of this round's 2M-point result (n0 = tid).  Round E parks it in the
                // workgroup's own line buffer (legs 2k and 2k+1 of a thread side by side: the thread that writes is the thread
                // that reads, so the layout is free -- twelve 16-byte accesses per thread instead of 24 of 8; the window between
                // barriers (3) and (4) is bound by the CU's memory pipeline); round O fetches it back, forms y[m] = ye[n'] + w_4M^-m yo[n'] (the 1/2 sits in the
                // kernel-spectrum table) and stores output sample i = m - (P - 1), m = n' + 2M for the legs below qw (sign -:
                // w_4M^-2M = -1) and m = n' from leg qw on (the index wraps once along the butterfly).  Legs whose sample lies
                // outside [0, N) fall outside the descriptor's range and are dropped by the hardware, as everywhere.
                int to = ftid();
                constexpr int QS = 2 * S1;
                const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<v2f *>(a.wgpart) + (size_t)blockIdx.x * 2 * M, 0, 2 * M * 8, 0x00020000);
                // (two independent `if`s in this order, not if / else: the else-branch is laid out BEFORE the if-branch in the
                // structurized flow, and the 24 values the round-E stores read would then stay live -- spilled -- through the
                // whole round-O branch)
                if (ps != 0) {
                    PSX_STAMP(23);
                    const v2f wb = wdif, wbn = -wb;
                    const int qw = (a.P - 1 - to + QS - 1) / QS;              // first wrapped leg
                    // twiddles first, in place; then ye, eight loads at a time, fenced (as the partial sums of the partition:
                    // hoisted above the twiddles the 24 loads would not fit the register budget)
                    // ye is requested behind the butterfly (above it the 24 loads would not fit the register budget) and travels
                    // under the twiddle pass
                    __builtin_amdgcn_sched_barrier(0);
                    v4u o[RAD / 2];
#pragma unroll
                    for (int q = 0; q < RAD / 2; ++q) o[q] = __builtin_amdgcn_raw_buffer_load_b128(rp, to * 16, q * TC * 16, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    pk_static_for<0, RAD>([&](auto qc) __attribute__((always_inline)) {
                        constexpr int q = decltype(qc)::value;
                        const v2f wsel = q >= qw ? wb : wbn;
                        const v2f wq = pk_twiddle<2 * RAD, q, true>(wsel);            // +-exp(+2 pi i (n0 + 768 q) / 4M)
                        v[q] = pk_cmul(v[q], wq);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                    PSX_STAMP(14);
#pragma unroll
                    for (int q = 0; q < RAD / 2; ++q) {
                        v[2 * q] += __builtin_bit_cast(v2f, (v2u){o[q].x, o[q].y});
                        v[2 * q + 1] += __builtin_bit_cast(v2f, (v2u){o[q].z, o[q].w});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    PSX_STAMP(15);
                    const int l = l0;
                    const bool lok = l < a.nlines;
                    const int ifirst = to + 2 * M - (a.P - 1);               // sample index of leg 0, not wrapped
                    const unsigned e0 = a.out_blocked ? (unsigned)(((ifirst >> IBS) * a.nlines + l) * IB + (ifirst & (IB - 1))) : (unsigned)ifirst;
                    const unsigned estep = a.out_blocked ? (unsigned)((QS / IB) * a.nlines * IB) : (unsigned)QS;
                    const unsigned ewrap = e0 - (a.out_blocked ? (unsigned)((2 * M / IB) * a.nlines * IB) : (unsigned)(2 * M));
                    const int64_t wbase = a.out_blocked ? (int64_t)0 : (int64_t)l * a.out_ld;
                    const unsigned welems = lok ? (a.out_blocked ? (unsigned)(((N + IB - 1) / IB) * IB) * (unsigned)a.nlines : (unsigned)N) : 0u;
                    if (wo) {
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(wo + wbase, 0, (int)(welems * 8u), 0x00020000);
                        if (!(gp.x == 1.f && gp.y == 0.f)) {             // pass 2: the global phase exp(i k z / M)
#pragma unroll
                            for (int q = 0; q < RAD; ++q) v[q] = pk_cmul_s(v[q], gp);
                        }
#pragma unroll
                        for (int q = 0; q < RAD; ++q) {
                            const unsigned off = ((q >= qw ? ewrap : e0) + (unsigned)q * estep) * 8u;
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v[q]), rs, (int)off, 0, 0);
                        }
                    }
                    if (io) {
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(io + wbase, 0, (int)(welems * 4u), 0x00020000);
                        if (a.accumulate) {
#pragma unroll
                            for (int q = 0; q < RAD; ++q) {
                                const unsigned off = ((q >= qw ? ewrap : e0) + (unsigned)q * estep) * 4u;
                                const float I = sc * (v[q].x * v[q].x + v[q].y * v[q].y);
                                const float old = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, old + I), rs, (int)off, 0, 0);
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < RAD; ++q) {
                                const unsigned off = ((q >= qw ? ewrap : e0) + (unsigned)q * estep) * 4u;
                                const float I = sc * (v[q].x * v[q].x + v[q].y * v[q].y);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, I), rs, (int)off, 0, 0);
                            }
                        }
                    }
                }
                int pe = ps;
                asm volatile("" : "+s"(pe));         // opaque: seen as the complement of the test above, the two are merged again
                if (pe == 0) {
#pragma unroll
                    for (int q = 0; q < RAD / 2; ++q) {
                        const v2u lo = __builtin_bit_cast(v2u, v[2 * q]), hi = __builtin_bit_cast(v2u, v[2 * q + 1]);
                        __builtin_amdgcn_raw_buffer_store_b128((v4u){lo.x, lo.y, hi.x, hi.y}, rp, to * 16, q * TC * 16, 0);
                    }
                }
            } else {
                // A wave's 64 butterflies belong to ONE line: its outputs go through a buffer descriptor whose range is exactly
                // the window they may touch (store_window)
                const int l = l0 + (PAIR ? 0 : __builtin_amdgcn_readfirstlane(lineA));
                const bool lok = l < a.nlines;
                // element index of output i inside the window: plain rows: i (window = row l); blocked: ((i>>3)*nlines+l)*8 + i%8
                const int e0 = a.out_blocked ? ((ifirst >> IBS) * a.nlines + l) * IB + (ifirst & (IB - 1)) : ifirst;
                constexpr int QS = PAIR ? 2 * S1 : S1;                    // output samples between two outputs of a butterfly
                const int estep = a.out_blocked ? (QS / IB) * a.nlines * IB : QS;
                // the window is the output block only -- samples [b*B, b*B + Bv) of the line; in the blocked layout
                // they are the contiguous range of B/8 sample-blocks (B is a multiple of 8)
                const int nout = min(a.B, N - pb * a.B);
                const int64_t wbase = a.out_blocked ? (int64_t)(pb * a.B / IB) * a.nlines * IB : (int64_t)l * a.out_ld + pb * a.B;
                const int welems = lok ? (a.out_blocked ? ((nout + IB - 1) / IB) * IB * a.nlines : nout) : 0;
                bool emit = true;                // only the last kernel segment produces the outputs proper
                {
                    // segments 0..S-2 leave their sum in `part`, the last one adds it to its own result.  Every round of a
                    // unit maps output element -> (wave, lane, q) identically, so a lane re-reads what it wrote itself.
                    const bool first = ps == 0, last = ps == a.S - 1;
                    emit = last;
                    if (a.S > 1) {
                        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
                            reinterpret_cast<v2f *>(a.part[d]) + wbase, 0, welems * 8, 0x00020000);
                        int off = e0 * 8;
                        if (!first) {
                            // eight loads at a time, fenced: hoisted above the butterfly (they do not depend on it) the 24
                            // loads would not fit the 128-VGPR budget next to its temporaries
#pragma unroll
                            for (int q0 = 0; q0 < RAD; q0 += 8) {
                                __builtin_amdgcn_sched_barrier(0);
                                v2u o[8];
#pragma unroll
                                for (int q = 0; q < 8; ++q)
                                    o[q] = __builtin_amdgcn_raw_buffer_load_b64(rp, off + (q0 + q) * estep * 8, 0, 0);   // 0 outside the window
#pragma unroll
                                for (int q = 0; q < 8; ++q) v[q0 + q] += __builtin_bit_cast(v2f, o[q]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (!last) {
#pragma unroll
                            for (int q = 0; q < RAD; ++q)
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v[q]), rp, off + q * estep * 8, 0, 0);
                        }
                    }
                }
                if (emit) store_window(v, wo, io, wbase, welems, e0, estep, gp, sc, a.accumulate);
            }
        }
        PSX_STAMP(12);
        lds_barrier();                               // (4)
        PSX_STAMP(13);
    }
}


#undef PSX_STAMP

// z == 0 (EXP:233-234): out = psi, |psi|^2
template <int NM>
__global__ __launch_bounds__(256) void k_source_out(const float2 *__restrict__ src, float amp, Mats m,
                                                    float2 *__restrict__ wave_out, float *__restrict__ inten_out,
                                                    float scale, int accumulate, int64_t n) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const float2 v = source_wave<NM>(src, amp, m, p);
        if (wave_out) wave_out[p] = v;
        if (inten_out) {
            const float I = scale * (v.x * v.x + v.y * v.y);
            inten_out[p] = accumulate ? inten_out[p] + I : I;
        }
    }
}

// Transmitted source wave, written TRANSPOSED: out[y*Nx + x] = source_wave(x*Ny + y)  (K1 = SAM:248-282 fused with the
// first of the two transposes a two-pass separable transform needs).  One evaluation per propagate call serves all its
// distances, and pass 1 of the line engine then reads whole lines contiguously.  64x64 tiles through LDS: the thickness
// maps are read along y (their fast axis), the wave is written along x.
template <int NM>
__device__ __forceinline__ void source_transposed_tile(const float2 *__restrict__ src, float amp, const Mats &m,
                                                       float2 *__restrict__ out, int Nx, int Ny, int vec_ok, int tile_index) {
    __shared__ float2 tile[64][65];
    const int ntx = (Ny + 63) / 64;                       // tiles along y
    const int y0 = (tile_index % ntx) * 64, x0 = (tile_index / ntx) * 64;
    // read phase: a thread takes 4 consecutive y of one image row per pass (one 16-byte load per thickness map), a wave
    // 4 rows x 64 y; 4 passes cover the 64 rows of the tile.  Full tiles of maps whose rows are 16-byte aligned only.
    const bool vec = vec_ok && x0 + 64 <= Nx && y0 + 64 <= Ny;   // vec_ok: Ny % 4 == 0, 16-byte aligned maps (and input wave)
    if (vec) {
        const int yq = (threadIdx.x & 15) * 4, xr = threadIdx.x >> 4;
        float4 t[4][NM > 0 ? NM : 1], wv[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t p = (int64_t)(x0 + xr + 16 * r) * Ny + y0 + yq;
#pragma unroll
            for (int i = 0; i < NM; ++i) t[r][i] = *reinterpret_cast<const float4 *>(m.T[i] + p);
            if (src) {                                   // 4 complex samples of the input wave: two 16-byte loads
                wv[r][0] = *reinterpret_cast<const float4 *>(src + p);
                wv[r][1] = *reinterpret_cast<const float4 *>(src + p + 2);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float2 w = make_float2(amp, 0.f);
                if (src) {                               // same operation order as source_wave<NM>
                    const float2 wi = e == 0 ? make_float2(wv[r][0].x, wv[r][0].y)
                                             : (e == 1 ? make_float2(wv[r][0].z, wv[r][0].w)
                                                       : (e == 2 ? make_float2(wv[r][1].x, wv[r][1].y) : make_float2(wv[r][1].z, wv[r][1].w)));
                    float a = amp;
                    float2 ww = wi;
                    if (NM > 0) {
                        double ph = 0.0, la = 0.0;
#pragma unroll
                        for (int i = 0; i < NM; ++i) {
                            const float tv = e == 0 ? t[r][i].x : (e == 1 ? t[r][i].y : (e == 2 ? t[r][i].z : t[r][i].w));
                            ph = fma(m.cphase[i], (double)tv, ph);
                            la = fma(m.catt[i], (double)tv, la);
                        }
                        float c, sn;
                        cis_f64(ph, c, sn);
                        a *= exp_att(la);
                        ww = make_float2(wi.x * c - wi.y * sn, wi.x * sn + wi.y * c);
                    }
                    w = make_float2(a * ww.x, a * ww.y);
                } else if (NM > 0) {
                    double ph = 0.0, la = 0.0;
#pragma unroll
                    for (int i = 0; i < NM; ++i) {
                        const float tv = e == 0 ? t[r][i].x : (e == 1 ? t[r][i].y : (e == 2 ? t[r][i].z : t[r][i].w));
                        ph = fma(m.cphase[i], (double)tv, ph);
                        la = fma(m.catt[i], (double)tv, la);
                    }
                    float c, sn;
                    cis_f64(ph, c, sn);
                    const float av = amp * exp_att(la);
                    w = make_float2(av * c, av * sn);
                }
                tile[xr + 16 * r][yq + e] = w;
            }
        }
    } else {
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
        float2 v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int x = x0 + ty + 4 * r, y = y0 + tx;
            const bool ok = x < Nx && y < Ny;
            v[r] = source_wave<NM>(src, amp, m, ok ? (int64_t)x * Ny + y : (int64_t)0);   // clamped: loads stay unconditional
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[ty + 4 * r][tx] = v[r];
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int y = y0 + ty + 4 * r, x = x0 + tx;
        if (x < Nx && y < Ny) out[(int64_t)y * Nx + x] = tile[tx][ty + 4 * r];
    }
}

template <int NM>
__global__ __launch_bounds__(256) void k_source_transposed(const float2 *__restrict__ src, float amp, Mats m,
                                                           float2 *__restrict__ out, int Nx, int Ny, int vec_ok) {
    source_transposed_tile<NM>(src, amp, m, out, Nx, Ny, vec_ok, blockIdx.x);
}

// The same for a batch of source waves over the SAME thickness maps -- the energies of a detector bin (EXP:317-361): source
// e = blockIdx.y has its own input wave, amplitude and coefficients and writes plane e of `out`.
struct SrcBatch {
    const float2 *src[PSX_MAX_SRC];
    float amp[PSX_MAX_SRC];
    double cphase[PSX_MAX_SRC][PSX_MAX_MAT], catt[PSX_MAX_SRC][PSX_MAX_MAT];
};

struct MapPtrs {
    const float *T[PSX_MAX_MAT];
};

template <int NM>
__global__ __launch_bounds__(256) void k_source_transposed_batch(SrcBatch b, MapPtrs maps, float2 *__restrict__ out,
                                                                 size_t out_stride, int Nx, int Ny, int vec_ok) {
    const int e = blockIdx.y;
    Mats m;
    m.n = NM;
#pragma unroll
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) {
        m.T[i] = maps.T[i];
        m.cphase[i] = b.cphase[e][i];
        m.catt[i] = b.catt[e][i];
    }
    source_transposed_tile<NM>(b.src[e], b.amp[e], m, out + (size_t)e * out_stride, Nx, Ny, vec_ok, blockIdx.x);
}

// ---- float64 construction of the kernel spectrum FFT_M(IDFT_P(chirp)): chirp -> rocFFT inverse (length P, float64) ->
// zero-pad per kernel segment -> rocFFT forward (length M, float64, batched over the segments) -> digit-reversed float32
// table.  (Round 1 evaluated both transforms as direct O(P^2) / O(M P) float64 sums: 1.3 ms per spectrum, as much as a
// whole bench step, and a 25-energy spectrum visits 75 of them per position.)
__global__ void k_kern_H(double2 *H, int P, double a, double du) {   // EXP:243-250, FFT order
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int f = (i < (P + 1) / 2) ? i : i - P;
    const double u = (double)f * du;
    const double ph = -a * u * u;
    const double r = ph - PSX_TWO_PI * rint(ph * PSX_INV_TWO_PI);
    double s, c;
    sincos(r, &s, &c);
    H[i] = make_double2(c, s);
}

// The taps h = IDFT_P(H) come out of an unnormalised float64 inverse transform (rocFFT); segment sg of the kernel --
// taps [off, off + len) -- is laid out zero-padded to M points (and scaled by 1/P) for the forward transform of size M.
// DIF (part == 2; M here = the 18432 points of a coupled round): both "segments" hold ALL P taps, zero-padded to M points --
// segment 0 as they are (its transform is the even half H[2k] of the kernel spectrum of twice the size), segment 1 times w^d,
// w = exp(-2 pi i / 2M) (the odd half H[2k+1]): the decimation-in-frequency split of one double-size transform.
__global__ void k_kern_pad(const double2 *__restrict__ hP, double2 *__restrict__ buf, int P, int M, int Lh, int S, int part) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * S) return;
    const int sg = t / M, d = t - sg * M;
    const int off = part == 1 ? sg * Lh : 0, len = part == 1 ? min(Lh, P - off) : P;
    double2 v = make_double2(0.0, 0.0);
    if (d < len) {
        v = hP[off + d];
        v.x /= P;
        v.y /= P;
        if (part == 2 && sg == 1) {
            double s, c;
            sincospi(-(double)d / (double)M, &s, &c);
            v = make_double2(v.x * c - v.y * s, v.x * s + v.y * c);
        }
    }
    buf[t] = v;
}

// position p = q1*S1 + q2*R3 + q3 of the in-place DIF output holds frequency k = q1 + 24*q2 + 576*q3
__global__ void k_kern_perm(const double2 *__restrict__ Hh, float2 *__restrict__ out, int M, int R3, int S) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * S) return;
    const int sg = t / M, p = t - sg * M;
    const int S1 = M / RAD;
    const int q1 = p / S1, q2 = (p % S1) / R3, q3 = p % R3;
    const double2 v = Hh[(size_t)sg * M + q1 + RAD * q2 + RAD * RAD * q3];
    out[t] = make_float2((float)(v.x / M), (float)(v.y / M));
}

// PAIR: the spectrum of a segment has 2M points; position p of an LDS line meets the bins k(p) and k(p) + M
__global__ void k_kern_perm_pair(const double2 *__restrict__ Hh, float4 *__restrict__ out, int M, int R3, int S, double extra) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * S) return;
    const int sg = t / M, p = t - sg * M;
    const int S1 = M / RAD;
    const int q1 = p / S1, q2 = (p % S1) / R3, q3 = p % R3;
    const int k = q1 + RAD * q2 + RAD * RAD * q3;
    const double2 v0 = Hh[(size_t)sg * 2 * M + k], v1 = Hh[(size_t)sg * 2 * M + k + M];
    const double sc = extra / (2.0 * M);        // DIF: the 1/2 of the radix-2 recombination rides here
    out[t] = make_float4((float)(v0.x * sc), (float)(v0.y * sc), (float)(v1.x * sc), (float)(v1.y * sc));
}

// w2[slab] = exp(-2 pi i k0 / 2M) for the slab's first point, k0 = q1 + 24 q2, slab = q1 * (S1 / R3) + q2
__global__ void k_pair_twiddles(float2 *w2, int M, int R3) {
    const int sl = blockIdx.x * blockDim.x + threadIdx.x;
    const int S1 = M / RAD, per = S1 / R3;
    if (sl >= M / R3) return;
    const int q1 = sl / per, q2 = sl % per;
    double s, c;
    sincospi(-(double)(q1 + RAD * q2) / (double)M, &s, &c);       // -2 pi k0 / (2M)
    w2[sl] = make_float2((float)c, (float)s);
}

// DIF: w4[n0] = exp(+2 pi i n0 / 4M), n0 < 2 S1
__global__ void k_dif_twiddles(float2 *w4, int M) {
    const int n0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (n0 >= 2 * (M / RAD)) return;
    double s, c;
    sincospi((double)n0 / (2.0 * (double)M), &s, &c);
    w4[n0] = make_float2((float)c, (float)s);
}

__global__ void k_stage_twiddles(float2 *twA, float2 *twB, int M, int R3) {
    const int S1 = M / RAD;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    // layout [n][24]: the 24 twiddles of one butterfly are contiguous (192 B), so a thread fetches them with 16-byte
    // loads at immediate offsets from ONE address register
    if (idx < RAD * S1) {
        const int n = idx / RAD, q = idx % RAD;
        double s, c;
        sincospi(-2.0 * (double)(((long long)n * q) % M) / (double)M, &s, &c);
        twA[idx] = make_float2((float)c, (float)s);
    }
    if (idx < RAD * R3) {
        const int n = idx / RAD, q = idx % RAD;
        double s, c;
        sincospi(-2.0 * (double)((n * q) % S1) / (double)S1, &s, &c);
        twB[idx] = make_float2((float)c, (float)s);
    }
}

int pick_r3(int N, int margin) {
    const int need = 2 * N + 2 * margin - 1;   // L = N + P - 1
    // R3 = 2 (M = 1152, lines up to N = 561) exists in the kernel but is not used: its stage stride S1 = 48 is no multiple of
    // 32, so the LDS addressing is not affine and the instance spills 80 VGPRs; M = 2304 serves those grids as well -- they
    // are one round per CU either way (512^2: 64 groups x 4 distances on 256 CUs)
    for (int r3 : {4, 8, 16})
        if (576 * r3 >= need) return r3;
    return 0;
}

// Lines longer than that: partition outputs (blocks of B, a multiple of 8) and kernel taps (segments of Lh) so that one
// block x segment product is an M-point convolution, B + Lh - 1 <= M; fewest products, then fewest segments.
constexpr int PART_M = 576 * 16;
constexpr int BR = IB > 8 ? IB : 8;      // output blocks start on a block boundary of the intermediate
// mconv: points of one product -- PART_M, or 2 * PART_M when the two LDS lines are coupled into one transform (PAIR)
void part_geometry(int N, int margin, int &B, int &Lh, int &S, int &NB, int mconv = 2 * PART_M) {
    const int P = N + 2 * margin;
    long best = -1;
    for (int s = 1; s <= 64; ++s) {
        const int lh = (P + s - 1) / s;
        const int bmax = (mconv - lh + 1) / BR * BR;
        if (bmax < BR) continue;
        const int nb = (N + bmax - 1) / bmax;
        const long cost = (long)s * nb;
        if (best < 0 || cost < best) {
            best = cost;
            S = s; Lh = lh; NB = nb;
            B = ((N + nb - 1) / nb + BR - 1) / BR * BR;
        }
    }
}

// Does a line of N samples run as the two-round DIF convolution (k_fresnel_lines, DIF)?  It must be too long for one coupled
// product, fit one convolution of 4 x 9216 points, and the block x segment partition must need more than two rounds.
bool dif_geometry(int N, int margin) {
    // diagnostics (psx_debug_switch "no_dif" / "no_pair"): the round-2 / round-1 partition, read when a plan is created
    if (debug_switch(DBG_NO_DIF) || debug_switch(DBG_NO_PAIR) || pick_r3(N, margin)) return false;
    int B, Lh, S, NB;
    part_geometry(N, margin, B, Lh, S, NB);
    const int P = N + 2 * margin, Mc = 2 * PART_M;
    return S * NB > 2 && P <= Mc && N + P - 1 <= 2 * Mc && N + P - 1 >= Mc;
}

}  // namespace

namespace psx {

struct AxisTables {
    int N = 0, R3 = 0, M = 0;
    int p2 = 0;                                     // radix R1 of the power-of-two line kernels (fresnel_p2.hip; M = 256 R1), 0: the 576 R3-point ones
    int p2x = 0;                                    // lines of ~16384 samples on the two-round power-of-two kernel (fresnel_p2x.hip)
    int part = 0, B = 0, Lh = 0, S = 1, NB = 1;     // partitioned convolution (lines that do not fit one transform)
    int pair = 0, Mconv = 0;                        // part: the two LDS lines coupled into one transform of Mconv = 2M points
    int dif = 0;                                    // pair: one 2*Mconv-point convolution per line in two rounds (radix-2 DIF split)
    float2 *w2 = nullptr;                           // pair: slab twiddles
    float2 *w4 = nullptr;                           // dif: thread factors of w_{2 Mconv}
    float2 *twA = nullptr, *twB = nullptr;
    // float64 transforms of the kernel-spectrum build: taps = IDFT_P(chirp), spectrum = FFT_M(zero-padded taps) x S segments
    rocfft_plan planP = nullptr, planM = nullptr;
    rocfft_execution_info infoP = nullptr, infoM = nullptr;
    void *workP = nullptr, *workM = nullptr;
    double2 *bufP = nullptr, *bufM = nullptr;       // [P] chirp / taps, [S][M] padded taps / spectra
};

struct KernEntry {
    float2 *H;          // S spectra of M points (power-of-two lines: M points + the first taps, p2::spectrum_elems)
    int M, S;
    size_t elems;       // float2 elements allocated
    unsigned long long stamp;
};

// cache key of a kernel spectrum: it depends on these scalars only
typedef std::tuple<double, double, int, int> KernKey;   // (a, du, N, M)

struct LdsEngine {
    AxisTables ax[2];            // [0]: lines along axis 0 (length Nx), [1]: along axis 1 (length Ny)
    float2 *inter = nullptr;     // [max_dist][Nx/IB][Ny][IB] intermediates (pass-1 line y, sample x)
    size_t inter_elems = 0;
    float2 *pre = nullptr;       // [Ny][Nx] transmitted source wave, transposed (pass 0)
    float2 *part = nullptr;      // [max_dist][Nx][Ny] partial sums of pass 2 of the partitioned convolution when only |.|^2 is wanted
    float2 *wgpart = nullptr;    // [CUs][Mconv]: DIF rounds park the even half's result of a line here (one buffer per workgroup)
    int wgpart_groups = 0;
    unsigned *queue = nullptr;   // [2][QUEUE_WORDS]: work queues of pass 1 and pass 2 (zero between launches: the kernels re-arm them)
    bool use_queue = false;      // psx_fresnel_plan_work_queue
    float2 *pre_b = nullptr;     // [PSX_MAX_SRC][Ny][Nx], [MAX_LINE][inter_elems]: the same two for a batch of source waves,
    float2 *inter_b = nullptr;   // allocated by the first batched call (psx_fresnel_propagate_sources)
    // Kernel spectra, keyed by (a, du, N, M).  72 KiB each at 4096^2: the cache is sized for a polychromatic position
    // (energies x hops x axes: 25 x 3 x 2 = 150 keys visited cyclically -- an LRU smaller than that misses on EVERY lookup),
    // i.e. effectively unbounded; CACHE_CAP only bounds the memory of a plan fed with ever-changing scalars.
    std::map<KernKey, KernEntry> cache;
    size_t cache_bytes = 0;
    unsigned long long clock = 0;
};
constexpr size_t CACHE_CAP_BYTES = (size_t)1 << 30;

void lds_engine_work_queue(psx_fresnel_plan *p, int on) {
    if (p->lds) p->lds->use_queue = on != 0;
}

bool lds_engine_supported(int Nx, int Ny, int margin) {
    // any line length: one transform per line up to N = 4593, the partitioned convolution beyond.  The output windows are
    // buffer descriptors with 32-bit byte ranges: the whole blocked intermediate for a one-transform pass 1, one output
    // block of it (B/8 sample-blocks x Ny lines) for a partitioned one; a row of the result in pass 2.
    if (!(margin >= 0 && margin <= Nx - 1 && margin <= Ny - 1 && margin <= 2048 && (int64_t)Nx * Ny < (1ll << 31))) return false;
    int64_t span = (int64_t)cdiv(Nx, IB) * IB;                       // samples of a pass-1 line inside one window
    if (dif_geometry(Nx, margin)) {
        // the window is the whole blocked intermediate and its byte offsets are unsigned: the furthest leg of a butterfly
        // (dropped by the range check) must still be below 2^32
        const int64_t imax = 2 * (2 * PART_M) - (Nx + 2 * margin - 1) + 2 * (PART_M / RAD);
        return cdiv(imax, IB) * IB * (int64_t)Ny * (int64_t)sizeof(float2) < (1ll << 32);
    }
    if (!pick_r3(Nx, margin)) {
        int B, Lh, S, NB;
        part_geometry(Nx, margin, B, Lh, S, NB);      // the coupled-line partition (B is largest there)
        span = B;
    }
    return span * (int64_t)Ny * (int64_t)sizeof(float2) < (1ll << 31);
}

static int make_axis(AxisTables &t, int N, int margin, size_t &bytes) {
    t.N = N;
    t.R3 = pick_r3(N, margin);
    // a power of two just below N + P - 1 with the wrapped outputs put right (fresnel_p2.hip) wherever it is the shorter transform
    if (const int r1 = debug_switch(DBG_NO_P2) ? 0 : p2::pick_r1(N, margin); r1 && t.R3 && 256 * r1 < 576 * t.R3) t.p2 = r1;
    const bool no_pair = debug_switch(DBG_NO_PAIR) != 0;     // diagnostics: the round-1 partition (M-point products)
    if (!t.R3 && !debug_switch(DBG_NO_P2) && !debug_switch(DBG_NO_DIF) && !no_pair && p2::x_serves(N, margin)) {
        t.p2x = 1;                                           // one 32768-point convolution per line, two rounds of 16384 points
        t.R3 = 16;
        t.S = 2;
        t.NB = 1;
        t.B = (N + BR - 1) / BR * BR;
        t.Lh = N + 2 * margin;
    }
    if (!t.R3) {
        t.R3 = 16;
        t.part = 1;
        t.pair = no_pair ? 0 : 1;
        part_geometry(N, margin, t.B, t.Lh, t.S, t.NB, t.pair ? 2 * PART_M : PART_M);
        // Lines that fit ONE convolution of 4 x 9216 points take it in two coupled rounds (k_fresnel_lines, DIF) whenever the
        // block x segment partition needs more; the extension must have a partner for some positions and none beyond 2M
        if (t.pair && dif_geometry(N, margin)) {
            t.dif = 1;
            t.S = 2;
            t.NB = 1;
            t.B = (N + BR - 1) / BR * BR;
            t.Lh = N + 2 * margin;
        }
    }
    t.M = t.p2 ? 256 * t.p2 : (t.p2x ? p2::x_points() / 2 : 576 * t.R3);
    t.Mconv = (t.pair || t.p2x) ? 2 * t.M : t.M;
    const int S1 = t.M / RAD;
    if (t.p2x) {
        PSX_HIP(hipMalloc((void **)&t.twA, sizeof(float2) * p2::twA_elems(32)));
        PSX_HIP(hipMalloc((void **)&t.twB, sizeof(float2) * p2::twB_elems()));
        PSX_HIP(hipMalloc((void **)&t.w2, sizeof(float2) * 512));
        PSX_HIP(hipMalloc((void **)&t.w4, sizeof(float2) * 512));
        bytes += sizeof(float2) * (p2::twA_elems(32) + p2::twB_elems() + 1024);
        if (int rc = p2::build_tables(t.twA, t.twB, 32, nullptr)) return rc;
        if (int rc = p2::x_build_twiddles(t.w2, t.w4, nullptr)) return rc;
    } else if (t.p2) {
        PSX_HIP(hipMalloc((void **)&t.twA, sizeof(float2) * p2::twA_elems(t.p2)));
        PSX_HIP(hipMalloc((void **)&t.twB, sizeof(float2) * p2::twB_elems()));
        bytes += sizeof(float2) * (p2::twA_elems(t.p2) + p2::twB_elems());
        if (int rc = p2::build_tables(t.twA, t.twB, t.p2, nullptr)) return rc;
    } else {
        PSX_HIP(hipMalloc((void **)&t.twA, sizeof(float2) * RAD * S1));
        PSX_HIP(hipMalloc((void **)&t.twB, sizeof(float2) * RAD * t.R3));
        bytes += sizeof(float2) * RAD * (S1 + t.R3);
        k_stage_twiddles<<<(int)cdiv(RAD * S1, 256), 256>>>(t.twA, t.twB, t.M, t.R3);
        if (int rc = launch_check("k_stage_twiddles")) return rc;
    }
    // float64 transforms of the kernel-spectrum build
    if (int rc = rocfft_ensure_setup()) return rc;
    if (t.pair) {
        PSX_HIP(hipMalloc((void **)&t.w2, sizeof(float2) * (t.M / t.R3)));
        bytes += sizeof(float2) * (t.M / t.R3);
        k_pair_twiddles<<<(int)cdiv(t.M / t.R3, 256), 256>>>(t.w2, t.M, t.R3);
        if (int rc = launch_check("k_pair_twiddles")) return rc;
    }
    if (t.dif) {
        PSX_HIP(hipMalloc((void **)&t.w4, sizeof(float2) * 2 * S1));
        bytes += sizeof(float2) * 2 * S1;
        k_dif_twiddles<<<(int)cdiv(2 * S1, 256), 256>>>(t.w4, t.M);
        if (int rc = launch_check("k_dif_twiddles")) return rc;
    }
    const size_t P = (size_t)(N + 2 * margin), M = (size_t)t.Mconv;
    PSX_ROCFFT(rocfft_plan_create(&t.planP, rocfft_placement_inplace, rocfft_transform_type_complex_inverse,
                                  rocfft_precision_double, 1, &P, 1, nullptr));
    PSX_ROCFFT(rocfft_plan_create(&t.planM, rocfft_placement_inplace, rocfft_transform_type_complex_forward,
                                  rocfft_precision_double, 1, &M, (size_t)t.S, nullptr));
    size_t wP = 0, wM = 0;
    PSX_ROCFFT(rocfft_plan_get_work_buffer_size(t.planP, &wP));
    PSX_ROCFFT(rocfft_plan_get_work_buffer_size(t.planM, &wM));
    PSX_ROCFFT(rocfft_execution_info_create(&t.infoP));
    PSX_ROCFFT(rocfft_execution_info_create(&t.infoM));
    if (wP) {
        PSX_HIP(hipMalloc(&t.workP, wP));
        PSX_ROCFFT(rocfft_execution_info_set_work_buffer(t.infoP, t.workP, wP));
    }
    if (wM) {
        PSX_HIP(hipMalloc(&t.workM, wM));
        PSX_ROCFFT(rocfft_execution_info_set_work_buffer(t.infoM, t.workM, wM));
    }
    PSX_HIP(hipMalloc((void **)&t.bufP, sizeof(double2) * P));
    PSX_HIP(hipMalloc((void **)&t.bufM, sizeof(double2) * M * t.S));
    bytes += wP + wM + sizeof(double2) * (P + M * t.S);
    return 0;
}

int lds_engine_create(psx_fresnel_plan *p) {
    LdsEngine *e = new LdsEngine();
    p->lds = e;
    if (int rc = make_axis(e->ax[0], p->Nx, p->margin, p->bytes)) return rc;
    if (int rc = make_axis(e->ax[1], p->Ny, p->margin, p->bytes)) return rc;
    e->inter_elems = (size_t)cdiv(p->Nx, IB) * IB * (size_t)p->Ny;   // blocked layout [Nx/IB][Ny][IB]
    const size_t img = sizeof(float2) * e->inter_elems;
    PSX_HIP(hipMalloc((void **)&e->inter, img * p->max_dist));
    p->bytes += img * p->max_dist;
    // every buffer a propagate call needs exists from here on: the call itself allocates nothing but kernel spectra it
    // has not seen yet (and refuses to do that while its stream is being captured into a graph)
    const size_t npix = (size_t)p->Nx * p->Ny;
    PSX_HIP(hipMalloc((void **)&e->pre, sizeof(float2) * npix));
    p->bytes += sizeof(float2) * npix;
    PSX_HIP(hipMalloc((void **)&e->queue, sizeof(unsigned) * 2 * QUEUE_WORDS));
    PSX_HIP(hipMemset(e->queue, 0, sizeof(unsigned) * 2 * QUEUE_WORDS));
    if (e->ax[0].dif || e->ax[1].dif || e->ax[0].p2x || e->ax[1].p2x) {
        e->wgpart_groups = current_cu_count();
        // DIF rounds park 2 * PART_M points per workgroup, the two-round power-of-two kernel 2 * 16384 (ye and round O's input)
        const size_t per_wg = std::max((size_t)2 * PART_M, (e->ax[0].p2x || e->ax[1].p2x) ? p2::x_line_buffer_elems() : (size_t)0);
        PSX_HIP(hipMalloc((void **)&e->wgpart, sizeof(float2) * per_wg * (size_t)e->wgpart_groups));
        p->bytes += sizeof(float2) * per_wg * (size_t)e->wgpart_groups;
    }
    if (e->ax[1].part && e->ax[1].S > 1 && !e->ax[1].dif) {      // complex partial sums of pass 2 when only |.|^2 leaves the pass
        PSX_HIP(hipMalloc((void **)&e->part, sizeof(float2) * npix * p->max_dist));
        p->bytes += sizeof(float2) * npix * p->max_dist;
    }
    PSX_HIP(hipDeviceSynchronize());
    return 0;
}

void lds_engine_destroy(psx_fresnel_plan *p) {
    LdsEngine *e = p->lds;
    if (!e) return;
    for (auto &t : e->ax) {
        (void)hipFree(t.twA);
        (void)hipFree(t.twB);
        (void)hipFree(t.w2);
        (void)hipFree(t.w4);
        if (t.planP) rocfft_plan_destroy(t.planP);
        if (t.planM) rocfft_plan_destroy(t.planM);
        if (t.infoP) rocfft_execution_info_destroy(t.infoP);
        if (t.infoM) rocfft_execution_info_destroy(t.infoM);
        (void)hipFree(t.workP);
        (void)hipFree(t.workM);
        (void)hipFree(t.bufP);
        (void)hipFree(t.bufM);
    }
    for (auto &k : e->cache) (void)hipFree(k.second.H);
    (void)hipFree(e->inter);
    (void)hipFree(e->pre);
    (void)hipFree(e->part);
    (void)hipFree(e->wgpart);
    (void)hipFree(e->pre_b);
    (void)hipFree(e->inter_b);
    (void)hipFree(e->queue);
    delete e;
    p->lds = nullptr;
}

// Kernel spectrum of one (distance, axis): cached, because it depends on scalars only (the reference rebuilds its
// chirp on every call, EXP:243-248).  Built on `st` in float64.  A plan -- its cache, its scratch buffers, its
// intermediates -- serves ONE stream at a time (include/paresis_hip.h, thread model): stream order is then enough to keep a
// table alive until its readers are done; the one exception, evicting a table to reuse its memory, synchronises the stream.
static int kernel_spectrum(psx_fresnel_plan *p, AxisTables &t, double a, double du, hipStream_t st, const float2 **out) {
    LdsEngine *e = p->lds;
    const KernKey key(a, du, t.N, t.Mconv);
    auto it = e->cache.find(key);
    if (it != e->cache.end()) {
        it->second.stamp = ++e->clock;
        *out = it->second.H;
        return 0;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(PSX_E_STATE, "psx_fresnel_propagate: kernel spectrum (a=%g, du=%g) is not cached and cannot be built while "
                                 "the stream is being captured: run the call once outside the capture first", a, du);
    const size_t elems = t.p2x ? p2::x_spectrum_elems() : (t.p2 ? p2::spectrum_elems(t.p2) : (size_t)t.Mconv * t.S);
    KernEntry k{nullptr, t.Mconv, t.S, elems, ++e->clock};
    const size_t bytes = sizeof(float2) * elems;
    if (e->cache_bytes + bytes > CACHE_CAP_BYTES && !e->cache.empty()) {   // evict the least recently used table
        auto lru = e->cache.begin();
        for (auto jt = e->cache.begin(); jt != e->cache.end(); ++jt)
            if (jt->second.stamp < lru->second.stamp) lru = jt;
        PSX_HIP(hipStreamSynchronize(st));       // its last readers are done
        if (lru->second.elems == elems) {
            k.H = lru->second.H;
        } else {
            (void)hipFree(lru->second.H);
            e->cache_bytes -= sizeof(float2) * lru->second.elems;
        }
        e->cache.erase(lru);
    }
    if (!k.H) {
        PSX_HIP(hipMalloc((void **)&k.H, bytes));
        e->cache_bytes += bytes;
        p->bytes += bytes;
    }
    const int P = t.N + 2 * p->margin;
    PSX_ROCFFT(rocfft_execution_info_set_stream(t.infoP, st));
    PSX_ROCFFT(rocfft_execution_info_set_stream(t.infoM, st));
    PSX_TIMED("k_kern_H", st, k_kern_H<<<(int)cdiv(P, 256), 256, 0, st>>>(t.bufP, P, a, du));
    {
        ProfScope ps("kern_ifft_P", st);
        void *buf = t.bufP;
        PSX_ROCFFT(rocfft_execute(t.planP, &buf, nullptr, t.infoP));
    }
    if (t.p2x) {
        if (int rc = p2::x_pad_taps(t.bufP, t.bufM, P, st)) return rc;
    } else
        PSX_TIMED("k_kern_pad", st, k_kern_pad<<<(int)cdiv((int64_t)t.Mconv * t.S, 256), 256, 0, st>>>(t.bufP, t.bufM, P, t.Mconv, t.Lh, t.S, t.dif ? 2 : t.part));
    {
        ProfScope ps("kern_fft_M", st);
        void *buf = t.bufM;
        PSX_ROCFFT(rocfft_execute(t.planM, &buf, nullptr, t.infoM));
    }
    if (t.p2x) {
        if (int rc = p2::x_perm_spectrum(t.bufM, t.bufP, k.H, P, st)) {
            (void)hipFree(k.H);
            e->cache_bytes -= bytes;
            return rc;
        }
    } else if (t.p2) {
        if (int rc = p2::perm_spectrum(t.bufM, t.bufP, k.H, t.p2, P, st)) {
            (void)hipFree(k.H);
            e->cache_bytes -= bytes;
            return rc;
        }
    } else if (t.pair)
        PSX_TIMED("k_kern_perm", st, k_kern_perm_pair<<<(int)cdiv((int64_t)t.M * t.S, 256), 256, 0, st>>>(t.bufM, reinterpret_cast<float4 *>(k.H), t.M, t.R3, t.S, t.dif ? 0.5 : 1.0));
    else
        PSX_TIMED("k_kern_perm", st, k_kern_perm<<<(int)cdiv((int64_t)t.M * t.S, 256), 256, 0, st>>>(t.bufM, k.H, t.M, t.R3, t.S));
    if (int rc = launch_check("kernel spectrum")) {
        (void)hipFree(k.H);                       // not cached: give the table back
        e->cache_bytes -= bytes;
        return rc;
    }
    e->cache.emplace(key, k);
    *out = k.H;
    return 0;
}

// persistent workgroups: one per CU (the LDS footprint allows no more), a multiple of the 8 XCDs
static int line_grid_slots(int nwork, int cap) {
    int nslot = current_cu_count() / 8;
    if (nslot > (nwork + 7) / 8) nslot = (nwork + 7) / 8;
    if (cap && nslot > cap) nslot = cap;
    return nslot;
}

// lines that fit one LDS transform
template <int R3, bool CONTIG, bool DUAL = false, bool QUEUE = false>
static int launch_lines(const LineArgs &la, hipStream_t st, const char *name) {
    if constexpr (!QUEUE) {
        if (la.queue) return launch_lines<R3, CONTIG, DUAL, true>(la, st, name);
    }
    using GE = LineGeom<R3, false>;
    constexpr int LPG = DUAL ? GE::LINES / 2 : GE::LINES;        // image lines per round
    static std::atomic<unsigned long long> attr_mask{0};
    if (first_on_device(attr_mask))
        PSX_HIP(hipFuncSetAttribute((const void *)k_fresnel_lines<R3, CONTIG, DUAL, QUEUE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)GE::lds_bytes));
    const int nwork = ((la.nlines + LPG - 1) / LPG) * (la.dist_inner ? 1 : la.n_dist);
    // the queue buffer holds a counter per workgroup of at most 256 (QUEUE_WORDS) and a thief looks at the queues of its
    // XCD with one lane each, 32 at most: a device with more than 256 CUs runs the queued passes on 256 workgroups
    static_assert(QUEUE_WORDS == 16 * 257, "queue layout: 256 counters 64 bytes apart + the count of workgroups done");
    const int nslot = line_grid_slots(nwork, QUEUE ? 32 : 0);
    PSX_TIMED(name, st, k_fresnel_lines<R3, CONTIG, DUAL, QUEUE><<<8 * nslot, T, GE::lds_bytes, st>>>(la));
    const int rc = launch_check(name);
    if constexpr (QUEUE) {
        // a launch that did not start leaves nobody to re-arm the counters: zero them here, or every later launch would
        // silently skip the units they already count
        if (rc) (void)hipMemsetAsync(la.queue, 0, sizeof(unsigned) * QUEUE_WORDS, st);
    }
    return rc;
}

// longer lines: partitioned convolution, coupled lines (PAIR), DIF rounds
template <bool CONTIG, bool PAIR, bool DIF = false>
static int launch_part(const LineArgs &la, hipStream_t st, const char *name) {
    using GE = LineGeom<16, PAIR>;
    constexpr int LPG = PAIR ? 1 : GE::LINES;
    static std::atomic<unsigned long long> attr_mask{0};
    if (first_on_device(attr_mask))
        PSX_HIP(hipFuncSetAttribute((const void *)k_fresnel_part<CONTIG, PAIR, DIF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)GE::lds_bytes));
    const int nwork = ((la.nlines + LPG - 1) / LPG) * la.NB * (la.dist_inner ? 1 : la.n_dist);
    const int nslot = line_grid_slots(nwork, 0);
    if constexpr (DIF) {
        if (!la.wgpart || !la.w4 || 8 * nslot > la.wg_groups)
            return fail(PSX_E_STATE, "LDS engine: %d workgroups for %d private line buffers", 8 * nslot, la.wg_groups);
    }
    PSX_TIMED(name, st, k_fresnel_part<CONTIG, PAIR, DIF><<<8 * nslot, T, GE::lds_bytes, st>>>(la));
    return launch_check(name);
}

template <bool CONTIG>
static int launch_lines_r3(int R3, const LineArgs &la, hipStream_t st, const char *name, bool part = false, bool pair = false,
                           bool dif = false) {
    if (part && pair && dif) return launch_part<CONTIG, true, true>(la, st, name);
    if (part && pair) return launch_part<CONTIG, true>(la, st, name);
    if (part) return launch_part<CONTIG, false>(la, st, name);
    switch (R3) {
        case 4: return launch_lines<4, CONTIG>(la, st, name);
        case 8: return launch_lines<8, CONTIG>(la, st, name);
        case 16: return launch_lines<16, CONTIG>(la, st, name);
    }
    return fail(PSX_E_UNSUPPORTED, "LDS engine: unsupported line length");
}

int lds_engine_propagate(psx_fresnel_plan *p, const PropArgs &a) {
    LdsEngine *e = p->lds;
    hipStream_t st = a.stream;
    const int64_t npix = (int64_t)p->Nx * p->Ny;
    // z == 0 distances return the input field (EXP:233-234)
    int nz[PSX_MAX_DIST], nnz = 0;
    for (int d = 0; d < a.n_dist; ++d) {
        if (a.a[d] != 0.0) {
            nz[nnz++] = d;
            continue;
        }
        PSX_DISPATCH_NMAT(a.m.n, PSX_TIMED("k_source_out", st, k_source_out<NM><<<ew_grid(npix, 256), 256, 0, st>>>(
                                                                   a.wave_in, a.amp, a.m, a.wave_out ? a.wave_out[d] : nullptr,
                                                                   a.inten_out ? a.inten_out[d] : nullptr,
                                                                   a.inten_scale ? a.inten_scale[d] : 1.f, a.accumulate,
                                                                   npix)));
    }
    if (nnz == 0) return launch_check("k_source_out");

    // ---- pass 0: the transmitted source wave (K1), evaluated once for all distances and stored transposed [Ny][Nx]
    {
        const int ntiles = (int)(cdiv(p->Nx, 64) * cdiv(p->Ny, 64));
        int vec_ok = p->Ny % 4 == 0 && (uintptr_t)a.wave_in % 16 == 0;
        for (int i = 0; i < a.m.n && i < PSX_MAX_MAT; ++i) vec_ok = vec_ok && ((uintptr_t)a.m.T[i] % 16 == 0);
        PSX_DISPATCH_NMAT(a.m.n, PSX_TIMED("k_source_transposed", st, k_source_transposed<NM><<<ntiles, 256, 0, st>>>(
                                                                           a.wave_in, a.amp, a.m, e->pre, p->Nx, p->Ny, vec_ok)));
        if (int rc = launch_check("k_source_transposed")) return rc;
    }

    // ---- pass 1: lines along axis 0 of the image = rows of the transposed source (contiguous reads); line y writes row
    // y of the blocked intermediate of each distance.  ONE launch covers all distances (work item = (distance, line
    // group)): one prologue and one tail instead of n_dist.  The in-place middle stage consumes the forward spectrum, so the
    // forward stages are repeated per distance (keeping it in registers needs 64 VGPRs the engine waves do not have).
    const bool stamp_pass1 = debug_switch(DBG_STAMP_PASS1) != 0;   // diagnostics only (psx_debug_switch)
    const int stamp_round = debug_switch(DBG_STAMP_ROUND);
    {
        LineArgs la;
        la.N = p->Nx; la.nlines = p->Ny; la.margin = p->margin; la.P = p->Px; la.L = p->Nx + p->Px - 1;
        la.in_si = 1; la.in_sl = p->Nx; la.in_blocked = 0; la.out_ld = 0; la.out_blocked = 1;
        la.twA = e->ax[0].twA; la.twB = e->ax[0].twB;
        la.accumulate = 0; la.stamps = stamp_pass1 ? g_stamps : nullptr; la.stamp_j = stamp_round;
        la.queue = e->use_queue ? e->queue : nullptr;
        la.n_dist = nnz;
        const bool no_inner = debug_switch(DBG_NO_DIST_INNER) != 0;   // diagnostics: A/B of the work order
        // one source for all distances: a workgroup takes the distances of a line group in consecutive rounds and fetches the
        // group once -- but only when there are enough line groups to occupy every CU that way (small grids: 32 groups of 16
        // lines at 512^2 would leave 224 CUs idle; there every (distance, group) pair is its own work item)
        const int lds_lines = e->ax[0].p2 ? p2::lines_per_round(e->ax[0].p2, false) : TOT / (576 * e->ax[0].R3);
        const int lines_per_group = (nnz >= 2 && !e->ax[0].part) ? lds_lines / 2 : lds_lines;   // shared-forward rounds take half the LDS lines
        const int ngroups0 = (p->Ny + lines_per_group - 1) / lines_per_group;
        la.dist_inner = (no_inner || e->ax[0].part || ngroups0 < current_cu_count()) ? 0 : 1;
        // DIF rounds: the distances of a line in consecutive units of one workgroup, whose loaders keep the line in registers
        // (k_fresnel_part) -- when the lines alone fill the chip
        if (e->ax[0].dif && !no_inner && nnz >= 2 && p->Ny >= current_cu_count()) la.dist_inner = 1;
        if (e->ax[0].p2x) la.dist_inner = (!no_inner && p->Ny >= current_cu_count()) ? 1 : 0;     // a line fetched once for all its rounds
        la.B = e->ax[0].B; la.Lh = e->ax[0].Lh; la.S = e->ax[0].S; la.NB = e->ax[0].NB;
        for (int i = 0; i < PSX_MAX_DIST; ++i) {
            const int k = i < nnz ? i : 0;
            la.src[i] = e->pre;
            if (i < nnz) {
                if (int rc = kernel_spectrum(p, e->ax[0], a.a[nz[i]], a.du_x, st, &la.H[i])) return rc;
            } else {
                la.H[i] = la.H[0];
            }
            la.wave_out[i] = e->inter + (size_t)k * e->inter_elems;
            la.part[i] = la.wave_out[i];            // partial sums of a partitioned pass build up in the intermediate itself
            la.inten_out[i] = nullptr;
            la.scale[i] = 1.f;
            la.gph[i] = make_float2(1.f, 0.f);
        }
        la.w2 = e->ax[0].w2;
        la.w4 = e->ax[0].w4; la.wgpart = e->wgpart; la.wg_groups = e->wgpart_groups;
        la.dsh = 2 * PART_M - p->Px; la.thr = p->Nx + p->Px - 1 - 2 * PART_M;
        const bool no_dual = debug_switch(DBG_NO_DUAL) != 0;        // diagnostics: A/B of the shared forward transform
        if (e->ax[0].p2x) {
            if (int rc = p2::x_launch(true, la, st, "k_fresnel_cols")) return rc;
        } else if (!no_dual && la.dist_inner && nnz >= 2 && !e->ax[0].part) {
            // one line x two distances per round: the forward transform of a line is shared by the pair
            if (nnz & 1) {
                la.H[nnz] = la.H[nnz - 1];
                la.wave_out[nnz] = nullptr;       // odd count: the last pair's second result is computed and dropped
            }
            int rc = 0;
            if (e->ax[0].p2) rc = p2::launch(e->ax[0].p2, true, true, la, st, "k_fresnel_cols");
            else switch (e->ax[0].R3) {
                case 4: rc = launch_lines<4, true, true>(la, st, "k_fresnel_cols"); break;
                case 8: rc = launch_lines<8, true, true>(la, st, "k_fresnel_cols"); break;
                default: rc = launch_lines<16, true, true>(la, st, "k_fresnel_cols"); break;
            }
            if (rc) return rc;
        } else if (e->ax[0].p2) {
            if (int rc = p2::launch(e->ax[0].p2, true, false, la, st, "k_fresnel_cols")) return rc;
        } else if (int rc = launch_lines_r3<true>(e->ax[0].R3, la, st, "k_fresnel_cols", e->ax[0].part, e->ax[0].pair, e->ax[0].dif)) return rc;
    }

    // ---- pass 2: lines along axis 1 of the image = columns of the intermediate (strided reads: the second transpose);
    // line x writes row x of the result.  Again one launch for all distances.
    {
        LineArgs lb;
        lb.N = p->Ny; lb.nlines = p->Nx; lb.margin = p->margin; lb.P = p->Py; lb.L = p->Ny + p->Py - 1;
        lb.in_si = 0; lb.in_sl = 0; lb.in_blocked = 1; lb.out_ld = p->Ny; lb.out_blocked = 0;
        lb.twA = e->ax[1].twA; lb.twB = e->ax[1].twB;
        lb.accumulate = a.accumulate; lb.stamps = stamp_pass1 ? nullptr : g_stamps; lb.stamp_j = stamp_round;
        lb.queue = e->use_queue ? e->queue + QUEUE_WORDS : nullptr;
        lb.n_dist = nnz;
        lb.dist_inner = 0;
        lb.B = e->ax[1].B; lb.Lh = e->ax[1].Lh; lb.S = e->ax[1].S; lb.NB = e->ax[1].NB;
        for (int i = 0; i < PSX_MAX_DIST; ++i) {
            const int k = i < nnz ? i : 0, d = nz[k];
            lb.src[i] = e->inter + (size_t)k * e->inter_elems;
            if (i < nnz) {
                if (int rc2 = kernel_spectrum(p, e->ax[1], a.a[d], a.du_y, st, &lb.H[i])) return rc2;
            } else {
                lb.H[i] = lb.H[0];
            }
            lb.wave_out[i] = a.wave_out ? a.wave_out[d] : nullptr;
            lb.part[i] = lb.wave_out[i] ? lb.wave_out[i] : (e->part ? e->part + (size_t)k * npix : nullptr);
            lb.inten_out[i] = a.inten_out ? a.inten_out[d] : nullptr;
            lb.scale[i] = a.inten_scale ? a.inten_scale[d] : 1.f;
            const double g = a.gphase ? a.gphase[d] : 0.0;
            lb.gph[i] = make_float2((float)std::cos(g), (float)std::sin(g));   // exact reduction of ~1e11 rad (EXP:250)
        }
        lb.w2 = e->ax[1].w2;
        lb.w4 = e->ax[1].w4; lb.wgpart = e->wgpart; lb.wg_groups = e->wgpart_groups;
        lb.dsh = 2 * PART_M - p->Py; lb.thr = p->Ny + p->Py - 1 - 2 * PART_M;
        if (e->ax[1].p2x) {
            if (int rc2 = p2::x_launch(false, lb, st, "k_fresnel_rows")) return rc2;
        } else if (e->ax[1].p2) {
            if (int rc2 = p2::launch(e->ax[1].p2, false, false, lb, st, "k_fresnel_rows")) return rc2;
        } else if (int rc2 = launch_lines_r3<false>(e->ax[1].R3, lb, st, "k_fresnel_rows", e->ax[1].part, e->ax[1].pair, e->ax[1].dif)) return rc2;
    }
    return 0;
}

// ---- a batch of source waves in ONE launch per pass ------------------------------------------------------------------------
// The energies of a detector bin (EXP:317-361) differ in input wave, coefficients and chirp but share grid, maps and work
// buffers' shape.  On a grid that fills the chip a launch per energy costs nothing; on the grids the reference itself is run
// on (a few hundred pixels) a line kernel is ~25 us of start-up and drain whatever it computes, and a 25-energy position is
// 175 of them.  Here every (source, distance) pair is one more "distance" of the line kernels -- own source plane, own kernel
// spectrum, own outputs (the work item (pair, line group) of the small-grid order) -- and the pre-pass takes the source index
// as a grid axis: three launches for up to PSX_MAX_SRC sources.  Results are exactly those of one propagate call per source
// (same kernels, same arithmetic); nothing is accumulated here -- every pair has its own output.
int lds_engine_propagate_sources(psx_fresnel_plan *p, const SourcesArgs &a) {
    LdsEngine *e = p->lds;
    hipStream_t st = a.stream;
    const int V = a.n_src * a.n_dist;
    const size_t npix = (size_t)p->Nx * p->Ny;
    // one launch per pass only pays where a launch is mostly overhead and the generic (pair, group) work order applies
    const int lds_lines = e->ax[0].p2 ? p2::lines_per_round(e->ax[0].p2, false) : TOT / (576 * e->ax[0].R3);
    const int ngroups0 = (p->Ny + lds_lines - 1) / lds_lines;
    bool batched = a.n_src > 1 && !e->ax[0].part && !e->ax[1].part && ngroups0 < current_cu_count() && V <= MAX_LINE &&
                   (!e->ax[0].p2 || V <= p2::max_distances(e->ax[0].p2)) && (!e->ax[1].p2 || V <= p2::max_distances(e->ax[1].p2)) &&
                   a.n_src <= PSX_MAX_SRC;
    for (int v = 0; v < V && batched; ++v) batched = a.a[v] != 0.0;         // z == 0 pairs take the one-source path
    if (!batched) {
        for (int s = 0; s < a.n_src; ++s) {
            PropArgs pa;
            pa.wave_in = a.wave_in ? a.wave_in[s] : nullptr;
            pa.amp = a.amp[s];
            pa.m = a.maps;
            for (int i = 0; i < a.maps.n; ++i) {
                pa.m.cphase[i] = a.cphase ? a.cphase[(size_t)s * a.maps.n + i] : 0.0;
                pa.m.catt[i] = a.catt ? a.catt[(size_t)s * a.maps.n + i] : 0.0;
            }
            pa.n_dist = a.n_dist; pa.a = a.a + (size_t)s * a.n_dist; pa.gphase = a.gphase ? a.gphase + (size_t)s * a.n_dist : nullptr;
            pa.du_x = a.du_x; pa.du_y = a.du_y;
            pa.wave_out = a.wave_out ? a.wave_out + (size_t)s * a.n_dist : nullptr;
            pa.inten_out = a.inten_out ? a.inten_out + (size_t)s * a.n_dist : nullptr;
            pa.inten_scale = a.inten_scale ? a.inten_scale + (size_t)s * a.n_dist : nullptr;
            pa.accumulate = 0; pa.stream = st;
            if (int rc = lds_engine_propagate(p, pa)) return rc;
        }
        return 0;
    }
    if (!e->pre_b) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cs);
        if (cs != hipStreamCaptureStatusNone)
            return fail(PSX_E_STATE, "psx_fresnel_propagate_sources: the batch buffers are allocated by the first call; make it before capturing");
        PSX_HIP(hipMalloc((void **)&e->pre_b, sizeof(float2) * npix * PSX_MAX_SRC));
        PSX_HIP(hipMalloc((void **)&e->inter_b, sizeof(float2) * e->inter_elems * MAX_LINE));
        p->bytes += sizeof(float2) * (npix * PSX_MAX_SRC + e->inter_elems * MAX_LINE);
    }
    // ---- pass 0: the transmitted source wave of every source, transposed
    {
        SrcBatch b = {};
        MapPtrs mp = {};
        int vec_ok = p->Ny % 4 == 0;
        for (int i = 0; i < a.maps.n && i < PSX_MAX_MAT; ++i) {
            mp.T[i] = a.maps.T[i];
            vec_ok = vec_ok && ((uintptr_t)a.maps.T[i] % 16 == 0);
        }
        for (int s = 0; s < a.n_src; ++s) {
            b.src[s] = a.wave_in ? a.wave_in[s] : nullptr;
            vec_ok = vec_ok && ((uintptr_t)b.src[s] % 16 == 0);
            b.amp[s] = a.amp[s];
            for (int i = 0; i < a.maps.n; ++i) {
                b.cphase[s][i] = a.cphase ? a.cphase[(size_t)s * a.maps.n + i] : 0.0;
                b.catt[s][i] = a.catt ? a.catt[(size_t)s * a.maps.n + i] : 0.0;
            }
        }
        const dim3 grid((unsigned)(cdiv(p->Nx, 64) * cdiv(p->Ny, 64)), (unsigned)a.n_src);
        PSX_DISPATCH_NMAT(a.maps.n, PSX_TIMED("k_source_transposed", st, k_source_transposed_batch<NM><<<grid, 256, 0, st>>>(
                                                                              b, mp, e->pre_b, npix, p->Nx, p->Ny, vec_ok)));
        if (int rc = launch_check("k_source_transposed")) return rc;
    }
    // ---- pass 1 and pass 2: pair v = (source v / n_dist, distance v % n_dist) is "distance" v of the line kernels
    LineArgs la;
    la.N = p->Nx; la.nlines = p->Ny; la.margin = p->margin; la.P = p->Px; la.L = p->Nx + p->Px - 1;
    la.in_si = 1; la.in_sl = p->Nx; la.in_blocked = 0; la.out_ld = 0; la.out_blocked = 1;
    la.twA = e->ax[0].twA; la.twB = e->ax[0].twB;
    la.accumulate = 0; la.stamps = nullptr; la.stamp_j = 1; la.n_dist = V; la.dist_inner = 0; la.queue = e->use_queue ? e->queue : nullptr;
    la.B = e->ax[0].B; la.Lh = e->ax[0].Lh; la.S = e->ax[0].S; la.NB = e->ax[0].NB; la.w2 = e->ax[0].w2;
    la.w4 = nullptr; la.wgpart = nullptr; la.wg_groups = 0; la.dsh = 0; la.thr = 0;
    LineArgs lb;
    lb.N = p->Ny; lb.nlines = p->Nx; lb.margin = p->margin; lb.P = p->Py; lb.L = p->Ny + p->Py - 1;
    lb.in_si = 0; lb.in_sl = 0; lb.in_blocked = 1; lb.out_ld = p->Ny; lb.out_blocked = 0;
    lb.twA = e->ax[1].twA; lb.twB = e->ax[1].twB;
    lb.accumulate = 0; lb.stamps = nullptr; lb.stamp_j = 1; lb.n_dist = V; lb.dist_inner = 0; lb.queue = e->use_queue ? e->queue + QUEUE_WORDS : nullptr;
    lb.B = e->ax[1].B; lb.Lh = e->ax[1].Lh; lb.S = e->ax[1].S; lb.NB = e->ax[1].NB; lb.w2 = e->ax[1].w2;
    lb.w4 = nullptr; lb.wgpart = nullptr; lb.wg_groups = 0; lb.dsh = 0; lb.thr = 0;
    for (int i = 0; i < MAX_LINE; ++i) {
        const int v = i < V ? i : 0;
        la.src[i] = e->pre_b + (size_t)(v / a.n_dist) * npix;
        la.wave_out[i] = e->inter_b + (size_t)v * e->inter_elems;
        la.part[i] = la.wave_out[i];
        la.inten_out[i] = nullptr;
        la.scale[i] = 1.f;
        la.gph[i] = make_float2(1.f, 0.f);
        lb.src[i] = la.wave_out[i];
        lb.wave_out[i] = a.wave_out ? a.wave_out[v] : nullptr;
        lb.part[i] = lb.wave_out[i];
        lb.inten_out[i] = a.inten_out ? a.inten_out[v] : nullptr;
        lb.scale[i] = a.inten_scale ? a.inten_scale[v] : 1.f;
        const double g = a.gphase ? a.gphase[v] : 0.0;
        lb.gph[i] = make_float2((float)std::cos(g), (float)std::sin(g));
        if (i < V) {
            if (int rc = kernel_spectrum(p, e->ax[0], a.a[v], a.du_x, st, &la.H[i])) return rc;
            if (int rc = kernel_spectrum(p, e->ax[1], a.a[v], a.du_y, st, &lb.H[i])) return rc;
        } else {
            la.H[i] = la.H[0];
            lb.H[i] = lb.H[0];
        }
    }
    if (int rc = e->ax[0].p2 ? p2::launch(e->ax[0].p2, true, false, la, st, "k_fresnel_cols")
                             : launch_lines_r3<true>(e->ax[0].R3, la, st, "k_fresnel_cols")) return rc;
    return e->ax[1].p2 ? p2::launch(e->ax[1].p2, false, false, lb, st, "k_fresnel_rows")
                       : launch_lines_r3<false>(e->ax[1].R3, lb, st, "k_fresnel_rows");
}

}  // namespace psx
