// refract.hip -- ray-deflection step: phase gradient -> displacement -> bilinear intensity scatter (K9-K13).
//
// Replaces fastRefraction + fastloopNumba (refractionFileNumba2.py:25-86,198-263; v1 refractionFileNumba.py:11-135).
//
// The reference scatters in raster order on one CPU thread.  Here the scatter is turned into a per-tile GATHER:
// a workgroup owns one TH x TW output tile, stages the phase (float64) and the source intensity of the tile plus a halo
// of H+1 pixels in LDS, re-evaluates the displacement of every source pixel of tile+halo, and deposits only what lands
// in its own tile into an LDS accumulator.  The tile is then written once with plain coalesced stores: no zero-init
// pass, no global read-modify-write.
//
// The LDS accumulator is 64-bit FIXED POINT, not float: on gfx950 an LDS float atomic (ds_add_f32) retires one lane
// per ~3 clocks (194 clocks per wave instruction, tools/lds_atomic_bench.hip) and made this kernel LDS-bound, while
// ds_add_u64 takes ~8 clocks per wave instruction.  Each tile scales its deposits by a power of two chosen from the
// largest staged source intensity (2^-30 of it is one unit; 2^33 of headroom for sums), so the quantisation is far
// below float32 resolution and the tile sums do not depend on the order of the atomics (bitwise reproducible).
//
// Rays displaced by more than the halo ("far" rays) are written, already evaluated, to a per-tile list by the tile that
// owns the SOURCE pixel and replayed by a second kernel with global float atomics that applies the reference's border
// rules literally (RF2:235-262).
//
// HBM traffic per pixel: 4*nmat (thickness maps) + 4 (intensity, if given) read, 4 written  (BASELINE.md section 4
// prices the scatter at 12+4*nmat because the reference zero-initialises and read-modify-writes its output).
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "common.hpp"

using namespace psx;

namespace {

// Tile geometry.  H is the gather halo: rays with floor(D) in [-H, H-1] on both axes are "near".
template <int TH_, int TW_, int H_, int NT_>
struct Geo {
    static constexpr int TH = TH_, TW = TW_, H = H_, NT = NT_;
    static constexpr int SR = TH + 2 * H + 2, SC = TW + 2 * H + 2;   // staged phase (one more ring for the stencil)
    static constexpr int GR = TH + 2 * H, GC = TW + 2 * H;           // source pixels gathered by one tile
    // accumulator: the tile plus a one-pixel guard ring (a ray whose base pixel lies in [-1, TH-1] x [-1, TW-1] deposits
    // its four shares at base + {0, 1, AW, AW+1} with no per-share test; the ring is never written out), then a trash
    // area for rays that miss altogether.
    // Two layout choices, kept as build-time switches for the A/B of VERDICT r3 item 3 (tools/ab_refract.sh):
    //   PSX_ACC_PITCH  row pitch AW in entries: 58 -> TW + 2 (116 dwords = 52 mod 64 banks: lanes 6 or 26 apart share a bank as
    //                  soon as their floor(dx) differ), 64 -> 128 dwords (lanes collide only in the same column of different rows)
    //   PSX_MISS       where a ray that misses the tile adds its shares: 0 -> a slot of its own behind the accumulator
    //                  (ACC + 2*lane; pitch 58 only: with pitch 64 the area no longer fits two workgroups per CU), 1 -> the
    //                  bottom guard row (never read) at its lane's column, 2 -> the guard row at the column of its would-be
    //                  target modulo the pitch (pitch 64 only)
    // Measured (round 4, gpurun_out/r4s1; k_refract_near per 4-distance launch at 4096^2 / 16384^2): 58/0 267.5 us / 3.92 ms
    // (round 2's layout), 58/1 266.7 / 3.91, 64/1 268.6 / 3.89, 64/2 275.3 / 4.00 (round 3's: the regression): the default
    // is 58/1.  (Measured conflict ratios hardly differ: what is counted are same-address collisions of the scatter.)
#ifndef PSX_ACC_PITCH
#define PSX_ACC_PITCH 58
#endif
#ifndef PSX_MISS
#define PSX_MISS 1
#endif
    static constexpr int AW = PSX_ACC_PITCH == 64 ? 64 : TW + 2, ACC = (TH + 2) * AW;
    static constexpr int MISS = PSX_MISS;
    static constexpr int TRASH = MISS == 0 ? 2 * 64 + AW + 2 : 64 + 2;
    static_assert(AW >= TW + 2, "accumulator pitch");
    static_assert(MISS != 2 || (AW & (AW - 1)) == 0, "target-column misses need a power-of-two pitch");
    static constexpr size_t LDS = sizeof(double) * SR * SC + sizeof(float) * GR * GC + sizeof(long long) * (ACC + TRASH) + 16;
};
using GeoSmall = Geo<56, 56, 4, 1024>;   // 80 KiB of LDS: two workgroups per CU; 1.31 source evaluations per pixel
// 16 waves per workgroup, 32 per CU: the staging loads of one tile are hidden by more waves that are depositing (512
// threads: 134 us per 4096^2 launch, 1024: 123 us)
using GeoMid = Geo<52, 52, 6, 1024>;     // 76 KiB; 1.51 evaluations
using GeoWide = Geo<48, 48, 8, 1024>;    // 73 KiB: two workgroups per CU (one stages while the other deposits); 1.78 evaluations
// Round 5 (VERDICT r4 item 3 i): the 64 x 64 gather window with 12- and 16-pixel halos, for grids whose rays travel far in study
// pixels (oversampling >= 4): 2.56 and 4 evaluations per pixel against fewer replayed shares.  Measured in DESIGN.md section 4.3.
using GeoH12 = Geo<40, 40, 12, 1024>;    // 66 KiB
using GeoH16 = Geo<32, 32, 16, 1024>;    // 59 KiB
// one switch over the geometry of the calling thread (psx_refract_set_halo)
#define PSX_GEO_DISPATCH(G_, ...)                                             \
    [&]() {                                                                   \
        switch (g_refract_geometry) {                                         \
            case 1: { using G_ = GeoWide; return __VA_ARGS__; }               \
            case 2: { using G_ = GeoMid; return __VA_ARGS__; }                \
            case 3: { using G_ = GeoH12; return __VA_ARGS__; }                \
            case 4: { using G_ = GeoH16; return __VA_ARGS__; }                \
            default: { using G_ = GeoSmall; return __VA_ARGS__; }             \
        }                                                                     \
    }()
#define PSX_GEO_MAX(expr_of_G)                                                                                          \
    std::max({[&]() { using G_ = GeoSmall; return expr_of_G; }(), [&]() { using G_ = GeoMid; return expr_of_G; }(),     \
              [&]() { using G_ = GeoWide; return expr_of_G; }(), [&]() { using G_ = GeoH12; return expr_of_G; }(),      \
              [&]() { using G_ = GeoH16; return expr_of_G; }()})
constexpr int FAR_THREADS = 256;
// Lanes per far-ray list: a whole wave.  Fewer lanes per list (several lists per wave, on the idea that the replay is a chain
// of dependent latencies and most lists are short) was measured as a build-time A/B in round 4 (gpurun_out/r4s7) and is SLOWER
// the fewer lanes a list gets -- 4096^2 step, k_refract_far: 64 lanes 33.6 us, 32: 36.4, 16: 46.3, 8: 69.4, 4: 107.5; config 5
// (halo 8): 3.01 / 3.17 / 3.33 / 3.50 / 3.69 ms -- the records of a list are what runs in parallel.  The order-independent
// replay compacts with wave ballots, so the switch is gone.
constexpr int FAR_SUB = 64, FAR_LISTS = FAR_THREADS / FAR_SUB;     // lists per workgroup

// a far ray, already evaluated by the tile that owns its source pixel
struct FarRay {
    float dx, dy, I;
    int src;
};

struct RefractArgs {
    const float *I_in;
    // optional source split (psx_refract_split_f32): the call's TWO images are the refractions of the sources where mask == 0
    // (image 0) and where mask != 0 (image 1) -- fastRefractionDF's split by its width map (RF2:147-150).  The tile is staged
    // once; each source's side rides in the lowest mantissa bit of its staged float64 phase (1e-16 of the phase: the LDS
    // budget of two workgroups per CU has no room for a flag array) and each half runs the deposit loop only if the window
    // holds a source of its side.
    const float *mask;
    float I0;
    Mats m;
    const double *phi_in;
    float *I_out[PSX_MAX_DIST];   // one output image per distance of the call
    double dscale[PSX_MAX_DIST];  // displacement scale z / k / (h M) / h of each distance
    int ndist;
    float out_scale;
    int accumulate;
    float *Dx_out, *Dy_out;
    float *I_mut;
    int Nx, Ny, margin;
    float clamp_xf, clamp_yf;
    unsigned *status;
    unsigned *far_count;     // workspace: [ndist][ntiles] far rays found by each tile at each distance
    FarRay *far_list;        // then [ndist][ntiles][TH*TW] records (a tile can never overflow its slot)
    int tiles_x, tiles_y, tile_cap;
    unsigned far_stride;          // diagnostics (psx_debug_switch "far_stride"): the replay walks the lists in a strided order
    unsigned long long *stamps;   // diagnostics (psx_debug_stamps): 16 phase timestamps per workgroup
    // order-independent far-ray replay (psx_set_deterministic; null otherwise), all inside the caller's workspace:
    unsigned *det_gmax;           // largest finite |source intensity| any tile staged (float bits; cleared by a 16-byte memset node)
    unsigned det_scale_bits;      // != 0: the caller's intensity scale (psx_set_deterministic_scale) takes the place of det_gmax
    unsigned *det_fold_count;     // [ndist][ntiles] entries of each list's fold table (written by the ADD pass for EVERY list)
    long long *det_acc;           // [ndist][Nx*Ny] scratch words (only touched words are ever looked at: no initial state)
};

#define PSX_RSTAMP(k)                                                                     \
    do {                                                                                  \
        if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); \
    } while (0)

// One axis of the reference's split (RF2:228-233 + the sign cases of RF2:237-262), in padded coordinates.
// b = base index, nb = neighbour index, wb/wn their weights.
__device__ __forceinline__ void axis_split_ref(float d, int p, int &b, int &nb, float &wb, float &wn) {
    if (fabsf(d) > 1.f) {
        const float f = floorf(d);
        b = p + (int)f;
        const float w = d - f;          // exact in float32
        nb = b + 1;
        wn = w;
        wb = 1.f - w;
    } else {
        b = p;
        const float w = fabsf(d);
        nb = d >= 0.f ? p + 1 : p - 1;
        wn = w;
        wb = 1.f - w;
    }
}

// Displacement of one source pixel from the phase at its stencil neighbours (np.gradient edge_order=2, unit
// spacing; RF2:54-64).  get(i,j) returns phi at GLOBAL pixel (i,j).  Returns the (possibly zeroed) intensity.
// XCD-aware tile order: workgroups b and b+8 share an XCD (and its L2), so give each XCD a contiguous run of tiles;
// neighbouring tiles then re-read each other's halo rows from the same L2.  Bijective for any tile count.
__device__ __forceinline__ int xcd_tile(int b, int nt) {
    const int q = nt >> 3, r = nt & 7, x = b & 7;
    return x * q + (x < r ? x : r) + (b >> 3);
}

// the side of a split call's source in the lowest mantissa bit of its staged phase (see RefractArgs::mask)
__device__ __forceinline__ double with_side(double ph, unsigned side, bool split) {
    if (!split) return ph;
    return __longlong_as_double((__double_as_longlong(ph) & ~1ll) | (long long)side);
}

template <class G, int NM, bool HAS_I, bool HAS_PHI, bool SPLITC = false>       // SPLITC: a split call (RefractArgs::mask)
__device__ __forceinline__ void refract_near_body(const RefractArgs &a) {
    constexpr int TH = G::TH, TW = G::TW, H = G::H, SR = G::SR, SC = G::SC, GR = G::GR, GC = G::GC, NTHREADS = G::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *sphi = (double *)smem;                                        // [SR][SC]
    constexpr int AW = G::AW, ACC = G::ACC;
    long long *sacc = (long long *)(smem + sizeof(double) * SR * SC);     // [TH+2][AW] fixed point + trash
    float *sI = (float *)(sacc + ACC + G::TRASH);                         // [GR][GC]
    unsigned *sfar = (unsigned *)(sI + GR * GC);                          // far rays of this tile so far
    unsigned *smax = sfar + 1;                                            // largest |source intensity| (float bits)

    const int nt = a.tiles_x * a.tiles_y;
    const int tile = xcd_tile(blockIdx.x, nt);
    const int r0 = (tile / a.tiles_y) * TH, c0 = (tile % a.tiles_y) * TW;
    const int tid = threadIdx.x;

    PSX_RSTAMP(0);
    // ---- stage phi (float64) and source intensity for rows [r0-H-1, r0+TH+H+1) x cols [c0-H-1, c0+TW+H+1).
    // Addresses are clamped into the image so every load is unconditional (they can all be in flight together);
    // out-of-image entries are zeroed afterwards and never used as sources.
    constexpr int SITERS = (SR * SC + NTHREADS - 1) / NTHREADS;
    unsigned imax = 0u, imax1 = 0u;       // largest staged |intensity| (of side 0 / side 1 of a split call)
    if (tid == 0) {
        *sfar = 0u;
        smax[0] = 0u;
        smax[1] = 0u;
    }
    __syncthreads();
    // Tiles whose staged window (tile + halo + stencil ring) lies inside the image -- all but the outermost ring of
    // tiles -- take loops compiled without the clamps, the out-of-image tests and np.gradient's edge formulas: the
    // kernel is bound by instruction issue (vector AND scalar), and those tests are a quarter of its instructions.
    const bool window_inside = r0 - H - 1 >= 0 && r0 + TH + H + 1 <= a.Nx && c0 - H - 1 >= 0 && c0 + TW + H + 1 <= a.Ny;
    // staged pixels per thread whose loads are issued together: ALL of them (9 x nmat loads in flight, registers are
    // plentiful at 4 waves per SIMD) -- one memory latency per tile instead of one per batch of four
    // (more than four maps: in two batches -- 5 x 8 thickness values + intensities + phases do not fit the 64 VGPRs of a
    // kernel that keeps two 16-wave workgroups on a CU, and the nmat = 8 instances spilled 7-21 registers)
    // (and the one instantiation of the 16-pixel halo that would otherwise spill a register: four maps + image + phase)
    constexpr bool TIGHT = NM > 4 || (NM == 4 && HAS_I && HAS_PHI && H >= 16);
    constexpr int U = TIGHT ? (SITERS + 1) / 2 : SITERS;
    auto stage = [&](auto inside_tag) __attribute__((always_inline)) {
        constexpr bool IN = decltype(inside_tag)::value;
        for (int it0 = 0; it0 < SITERS; it0 += U) {
            float t[U][NM > 0 ? NM : 1], Iin[U];
            double phin[U];
            bool ok[U];
            unsigned sides = 0u;          // bit u: the side of staged pixel u (one register, not one per pixel)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = min((it0 + u) * NTHREADS + tid, SR * SC - 1);   // the last pass re-stages the last pixel
                const int sr = idx / SC, sc = idx - sr * SC;
                const int i = r0 - H - 1 + sr, j = c0 - H - 1 + sc;
                ok[u] = IN || (i >= 0 && i < a.Nx && j >= 0 && j < a.Ny);
                const int64_t p = IN ? (int64_t)i * a.Ny + j
                                     : (int64_t)min(max(i, 0), a.Nx - 1) * a.Ny + min(max(j, 0), a.Ny - 1);
#pragma unroll
                for (int m = 0; m < NM; ++m) t[u][m] = a.m.T[m][p];
                Iin[u] = HAS_I ? a.I_in[p] : 1.f;
                phin[u] = HAS_PHI ? a.phi_in[p] : 0.0;
                if constexpr (SPLITC) sides |= (a.mask[p] != 0.f ? 1u : 0u) << u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = min((it0 + u) * NTHREADS + tid, SR * SC - 1);
                const int sr = idx / SC, sc = idx - sr * SC;
                // the phase needs float64 (it is differenced); the attenuation exponent, at most a few units, does not:
                // exp(sum catt T) = exp2(sum (catt log2 e) T) in float32 is three instructions and good to 1e-7
                // (a transmission below 2^-126 comes out as zero)
                double ph = phin[u];
                float la2 = 0.f;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    ph = fma(a.m.cphase[m], (double)t[u][m], ph);
                    la2 = fmaf((float)(a.m.catt[m] * 1.4426950408889634), t[u][m], la2);
                }
                float I = a.I0 * Iin[u];
                if (NM > 0) I *= __builtin_amdgcn_exp2f(la2);   // v_exp_f32 itself: exp2f() wraps it in five instructions for results below 2^-126
                const unsigned side = (sides >> u) & 1u;
                sphi[idx] = with_side(ok[u] ? ph : 0.0, side, SPLITC);
                if (sr >= 1 && sr <= GR && sc >= 1 && sc <= GC) {
                    sI[(sr - 1) * GC + (sc - 1)] = ok[u] ? I : 0.f;
                    if (ok[u]) {                                              // NaN/inf sort above every finite value
                        const unsigned b = __float_as_uint(fabsf(I));
                        imax = max(imax, side ? 0u : b);
                        imax1 = max(imax1, side ? b : 0u);
                    }
                }
            }
        }
    };
    // Interior tiles, sources first: the GR x GC block of pixels that deposit is (GR*GC / NTHREADS) whole passes with
    // shift/mask indexing and no membership test; the one-pixel stencil ring around it only needs the phase.
    constexpr bool SPLIT = !TIGHT && (GC & (GC - 1)) == 0 && (GR * GC) % NTHREADS == 0 && 2 * (SR + SC) <= NTHREADS;
    auto stage_interior = [&]() __attribute__((always_inline)) {
        constexpr int NP = GR * GC / NTHREADS;
        float t[NP + 1][NM > 0 ? NM : 1], Iin[NP + 1];
        double phin[NP + 1];
        unsigned sides = 0u;
        // ring pixel of this thread (threads < 2*SC + 2*GR): top row, bottom row, left column, right column
        const int rt = tid;
        const bool ring = rt < 2 * SC + 2 * GR;
        const int rsr = rt < SC ? 0 : (rt < 2 * SC ? SR - 1 : 1 + ((rt - 2 * SC) >> 1));
        const int rsc = rt < SC ? rt : (rt < 2 * SC ? rt - SC : (((rt - 2 * SC) & 1) ? SC - 1 : 0));
#pragma unroll
        for (int u = 0; u <= NP; ++u) {
            const int idx = u * NTHREADS + tid;
            const int sr = u < NP ? 1 + idx / GC : (ring ? rsr : 0), sc = u < NP ? 1 + (idx & (GC - 1)) : (ring ? rsc : 0);
            const int64_t p = (int64_t)(r0 - H - 1 + sr) * a.Ny + (c0 - H - 1 + sc);
#pragma unroll
            for (int m = 0; m < NM; ++m) t[u][m] = a.m.T[m][p];
            Iin[u] = (HAS_I && u < NP) ? a.I_in[p] : 1.f;
            phin[u] = HAS_PHI ? a.phi_in[p] : 0.0;
            if constexpr (SPLITC) {
                if (u < NP) sides |= (a.mask[p] != 0.f ? 1u : 0u) << u;
            }
        }
#pragma unroll
        for (int u = 0; u <= NP; ++u) {
            const int idx = u * NTHREADS + tid;
            double ph = phin[u];
            float la2 = 0.f;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                ph = fma(a.m.cphase[m], (double)t[u][m], ph);
                if (u < NP) la2 = fmaf((float)(a.m.catt[m] * 1.4426950408889634), t[u][m], la2);
            }
            if (u < NP) {
                float I = a.I0 * Iin[u];
                if (NM > 0) I *= __builtin_amdgcn_exp2f(la2);   // v_exp_f32 itself: exp2f() wraps it in five instructions for results below 2^-126
                const unsigned side = (sides >> u) & 1u;
                sphi[(1 + idx / GC) * SC + 1 + (idx & (GC - 1))] = with_side(ph, side, SPLITC);
                sI[idx] = I;
                const unsigned b = __float_as_uint(fabsf(I));
                imax = max(imax, side ? 0u : b);
                imax1 = max(imax1, side ? b : 0u);
            } else if (ring) {
                sphi[rsr * SC + rsc] = ph;
            }
        }
    };
    if (window_inside) {
        if constexpr (SPLIT)
            stage_interior();
        else
            stage(std::true_type{});
    } else {
        stage(std::false_type{});
    }
    PSX_RSTAMP(1);
    for (int idx = tid; idx < ACC; idx += NTHREADS) sacc[idx] = 0ll;
    for (int o = 32; o > 0; o >>= 1) {
        imax = max(imax, (unsigned)__shfl_xor((int)imax, o));
        imax1 = max(imax1, (unsigned)__shfl_xor((int)imax1, o));
    }
    if ((tid & 63) == 0) {
        atomicMax(&smax[0], imax);
        if constexpr (SPLITC) atomicMax(&smax[1], imax1);
    }
    __syncthreads();
    PSX_RSTAMP(2);
    // fixed-point scale of this tile: one unit = 2^-30 of (the power of two above) the largest staged intensity
    const unsigned mside0 = smax[0], mside1 = smax[1];
    const unsigned mbits = max(mside0, mside1);
    const bool finite_in = mbits < 0x7f800000u;
    const int sexp = min(120, max(-120, 30 - (mbits ? ilogbf(__uint_as_float(mbits)) + 1 : 0)));
    const float fscale_f = finite_in ? ldexpf(1.f, sexp) : 0.f;     // power of two: scaling a float by it is exact
    const double finv = finite_in ? ldexp(1.0, -sexp) : 0.0;
    // order-independent far-ray replay: the far shares of the WHOLE call are summed in one fixed-point unit, 2^-30 of the
    // power of two above the largest finite intensity any tile staged.  A tile only sends its maximum when it beats what the
    // word already holds (a stale read costs one more atomic, nothing else): a few hundred atomics per call, not one per tile.
    if (a.det_gmax && !a.det_scale_bits && tid == 0 && finite_in && mbits &&
        mbits > __hip_atomic_load(a.det_gmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(a.det_gmax, mbits);
    // A window that holds no intensity at all deposits nothing, lists nothing and leaves zeros: skip the deposit loops (the
    // halves of the dark-field split, RF2:147-150, are zero over most of the image; so is any masked input).  Not when the
    // displacement maps are wanted (they do not depend on the intensity), nor when a foreign array is to be zeroed.
    if (mbits == 0u && !a.Dx_out && (a.I_mut == nullptr || a.I_mut == a.I_in)) {     // uniform: one LDS word
        for (int d = 0; d < a.ndist; ++d) {
            if (!a.accumulate) {
                float *const I_out = a.I_out[d];
                for (int idx = tid; idx < TH * TW; idx += NTHREADS) {
                    const int tr = idx / TW, tc = idx - tr * TW;
                    const int i = r0 + tr, j = c0 + tc;
                    if (i < a.Nx && j < a.Ny) I_out[(int64_t)i * a.Ny + j] = 0.f;    // out_scale * 0 (a NaN scale aside)
                }
            }
            if (tid == 0) a.far_count[(size_t)d * nt + tile] = 0u;
        }
        return;
    }


    // ---- every source pixel of tile+halo deposits what lands inside this tile.
    // Written with as few divergent branches as possible (each costs a save/restore of the exec mask): pixels outside
    // the image carry I = 0, deposits that miss the tile (or rays that are not "near") add 0 to a per-lane trash slot,
    // so the four LDS atomics are unconditional.
    // All distances of the call share the staged tile (the phase gradient is linear in z): the thickness maps are read
    // and the transmission evaluated ONCE per tile, then each distance runs its own deposit loop into the same
    // accumulator and writes its own image.
    bool any_bad = false;
    constexpr int ITERS = (GR * GC + NTHREADS - 1) / NTHREADS;   // uniform trip count: the loop holds wave ballots
    // Interior tiles of a distance batch: the float64 phase differences of a thread's ITERS sources are formed ONCE (16 registers)
    // and every distance scales them -- the same products bit for bit, (xp - xm) * hscale, without the four LDS reads, their
    // address and the two float64 subtractions per source and distance (PSX_NEAR_HOIST, tools/ab_src.sh)
#ifndef PSX_NEAR_HOIST
#define PSX_NEAR_HOIST 1
#endif
    constexpr bool HOIST = PSX_NEAR_HOIST && (GR * GC) % NTHREADS == 0;
    double hgx[HOIST ? ITERS : 1], hgy[HOIST ? ITERS : 1];
    const bool hoisted = HOIST && window_inside && a.ndist > 1;      // uniform
    if constexpr (HOIST) {
        if (hoisted) {
            const volatile __attribute__((address_space(3))) double *vphi = (const volatile __attribute__((address_space(3))) double *)sphi;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int idx = it * NTHREADS + tid;
                const int sidx = ((int)((unsigned)idx / (unsigned)GC) + 1) * SC + ((int)((unsigned)idx % (unsigned)GC) + 1);
                const double xp = vphi[sidx + SC], xm = vphi[sidx - SC], yp = vphi[sidx + 1], ym = vphi[sidx - 1];
                hgx[it] = xp - xm;
                hgy[it] = yp - ym;
            }
        }
    }
    // ... and with a wide halo such a tile takes its distances TWO at a time: once the differences sit in registers nobody reads the
    // staged phase again, and its 34 KiB hold a second accumulator -- one barrier and one write-out phase per PAIR of distances
    // (PSX_NEAR_PAIR; not the split call, whose sides ride in the staged phase).  A/B on one box (gpurun_out/r6s42), tiles +
    // replay per 4-distance launch: 16384^2 halo 12 7.79 / 7.85 -> 7.54 / 7.57 ms, halo 16 8.49 -> 7.83, halo 8 8.0 -> 7.9; the
    // narrow halos LOSE (4096^2 halo 6: tiles 0.2775 -> 0.2812 ms; 16384^2 halo 4: 10.05 -> 10.20): compiled in for halo >= 8.
#ifndef PSX_NEAR_PAIR
#define PSX_NEAR_PAIR 1
#endif
    static_assert(sizeof(double) * SR * SC >= sizeof(long long) * (ACC + G::TRASH), "the staged phase holds a second accumulator");
    long long *const sacc2 = (long long *)sphi;
    unsigned *const sfar2 = sfar + 3;                    // (sfar, smax[0], smax[1], sfar2: the 16 bytes behind the intensity window)
    const bool pairs = PSX_NEAR_PAIR && H >= 8 && HOIST && !SPLITC && hoisted;      // uniform
    if (pairs) {
        __syncthreads();                                 // every thread holds its differences
        for (int idx = tid; idx < ACC; idx += NTHREADS) sacc2[idx] = 0ll;
        if (tid == 0) *sfar2 = 0u;
        __syncthreads();
    }
    for (int d = 0; d < a.ndist; d += pairs ? 2 : 1) {
    const int nd = pairs && d + 1 < a.ndist ? 2 : 1;     // distances of this step (uniform)
    // the per-thread index arithmetic is recomputed per distance rather than kept live across the loop (opaque copy of
    // the thread index): hoisted, it costs 40 VGPRs and the second workgroup of the CU
    int tl = tid;
    asm volatile("" : "+v"(tl));
    tl &= NTHREADS - 1;                 // tells the compiler the range again (unsigned shifts for the index split)
    const int lane = tl & 63;
    const int d1 = nd == 2 ? d + 1 : d;
    const double dscale = a.dscale[d];
    const double hscale = 0.5 * dscale, hscale1 = 0.5 * a.dscale[d1];
    float *const I_out = a.I_out[d];
    FarRay *const far_list = a.far_list + ((size_t)d * nt + tile) * (TH * TW);
    FarRay *const far_list1 = a.far_list + ((size_t)d1 * nt + tile) * (TH * TW);
    if (SPLITC && (d == 0 ? mside0 : mside1) == 0u) {    // split call: no source of this side in the window (uniform)
        if (!a.accumulate)
            for (int idx = tl; idx < TH * TW; idx += NTHREADS) {
                const int tr = idx / TW, tc = idx - tr * TW;
                const int i = r0 + tr, j = c0 + tc;
                if (i < a.Nx && j < a.Ny) I_out[(int64_t)i * a.Ny + j] = 0.f;
            }
        if (tl == 0) a.far_count[(size_t)d * nt + tile] = 0u;
        continue;                                        // the accumulator and the list counter are untouched: no barrier owed
    }
    // D = gradient(phi) * dscale at staged pixel `sidx` (image pixel (i, j)), float64 differencing, float32 result
    auto displacement = [&](auto inside_tag, int it, int dd, int i, int j, int sidx, bool inside, float &dx, float &dy) __attribute__((always_inline)) {
        constexpr bool IN = decltype(inside_tag)::value && (GR * GC) % NTHREADS == 0;
        double gx, gy;
        if constexpr (IN && HOIST) {
            if (hoisted) {                                                // uniform
                const double hs = dd ? hscale1 : hscale;
                dx = (float)(hgx[it] * hs);
                dy = (float)(hgy[it] * hs);
                return;
            }
        }
        // gx, gy hold gradient * dscale.  (d * 0.5) * dscale == d * (0.5 * dscale) bit for bit: halving is exact.
        if (IN || (i > 0 && i < a.Nx - 1 && j > 0 && j < a.Ny - 1)) {   // interior: central differences (RF2:54)
            // four ds_read_b64 (2 LDS cycles per wave instruction), not the two ds_read2_b64 (8 each: MI355X_MICROARCH.md, LDS
            // table) the compiler makes of plain loads -- it leaves volatile ones alone.  0.271 -> 0.262 ms at 4096^2 (r4s21)
            const volatile __attribute__((address_space(3))) double *vphi =
                (const volatile __attribute__((address_space(3))) double *)sphi;
#if defined(PSX_NEAR_EXP) && (PSX_NEAR_EXP & 2)
            const double xp = 1e-3 * sidx, xm = 0.0, yp = -2e-3 * sidx, ym = 0.0;
#else
            const double xp = vphi[sidx + SC], xm = vphi[sidx - SC], yp = vphi[sidx + 1], ym = vphi[sidx - 1];
#endif
            gx = (xp - xm) * hscale;
            gy = (yp - ym) * hscale;
        } else if (inside) {                                         // image border: np.gradient(edge_order=2)
            if (i == 0)
                gx = -1.5 * sphi[sidx] + 2.0 * sphi[sidx + SC] - 0.5 * sphi[sidx + 2 * SC];
            else if (i == a.Nx - 1)
                gx = 0.5 * sphi[sidx - 2 * SC] - 2.0 * sphi[sidx - SC] + 1.5 * sphi[sidx];
            else
                gx = 0.5 * (sphi[sidx + SC] - sphi[sidx - SC]);
            if (j == 0)
                gy = -1.5 * sphi[sidx] + 2.0 * sphi[sidx + 1] - 0.5 * sphi[sidx + 2];
            else if (j == a.Ny - 1)
                gy = 0.5 * sphi[sidx - 2] - 2.0 * sphi[sidx - 1] + 1.5 * sphi[sidx];
            else
                gy = 0.5 * (sphi[sidx + 1] - sphi[sidx - 1]);
            gx *= dscale;
            gy *= dscale;
        } else {
            gx = 0.0;
            gy = 0.0;
        }
        // the displacement is a small number: everything after the float64 differencing runs in float32
        dx = (float)gx;
        dy = (float)gy;
    };
    auto gather = [&](auto inside_tag, auto nd_tag) __attribute__((always_inline)) {
    constexpr bool IN = decltype(inside_tag)::value && (GR * GC) % NTHREADS == 0;   // every slot is a pixel inside the image
    constexpr int ND = decltype(nd_tag)::value;          // distances deposited by this pass (2: into the two accumulators)
    // (distance outside, sources inside: one source feeding both distances in turn keeps a dozen more values alive -- 2 spilled
    // registers in the wide-halo instances; the pass is about the barrier and the write-out it saves, not about the sources)
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) {
    for (int it = 0; it < ITERS; ++it) {
        const int idx = IN ? it * NTHREADS + tl : min(it * NTHREADS + tl, GR * GC - 1);
        const bool live = IN || it * NTHREADS + tl < GR * GC;
        const int gr = (int)((unsigned)idx / (unsigned)GC), gc = (int)((unsigned)idx % (unsigned)GC);
        const int i = r0 - H + gr, j = c0 - H + gc;
        const bool inside = IN || (live && i >= 0 && i < a.Nx && j >= 0 && j < a.Ny);
        const bool core = gr >= H && gr < H + TH && gc >= H && gc < H + TW;
        float I0 = live ? sI[idx] : 0.f;                             // 0 outside the image
        const int sidx = (gr + 1) * SC + (gc + 1);                   // this pixel in the staged phase tile
        if constexpr (SPLITC)                                        // split call: only the sources of side d
            I0 = (reinterpret_cast<const unsigned *>(sphi)[2 * sidx] & 1u) == (unsigned)d ? I0 : 0.f;
        float I = I0;
        float dx, dy;
        displacement(inside_tag, it, dd, i, j, sidx, inside, dx, dy);
        // RF2:59-60 zeroes |D| < 1e-12.  For the deposit that is a no-op in float32 -- such a ray puts weight 1.0f on its
        // own pixel and less than 2^-30 of a unit elsewhere either way -- so only the displacement maps apply it.
        const bool clx = fabsf(dx) > a.clamp_xf, cly = fabsf(dy) > a.clamp_yf;   // RF2:61-64
        const bool clamped = clx || cly;
        I = clamped ? 0.f : I;
        dx = clx ? 0.f : dx;
        dy = cly ? 0.f : dy;
        const float fx = floorf(dx), fy = floorf(dy);
        const int ifx = (int)fx, ify = (int)fy;                      // |d| <= clamp: far inside int range (saturates beyond)
        // rays with floor(D) in [-H, H-1] on both axes are gathered whole by the tiles of their targets; a longer one
        // goes on this tile's list unless all four of its shares land in this tile's own core (the replay sorts out, share
        // by share, what the gathers already covered)
        const bool near = (unsigned)(ifx + H) < 2u * H && (unsigned)(ify + H) < 2u * H;
        const bool own = (unsigned)(gr - H + ifx) < (unsigned)(TH - 1) && (unsigned)(gc - H + ify) < (unsigned)(TW - 1);
        const bool far = core && inside && !near && !own && I != 0.f;
        const float Dxs = dx, Dys = dy, Is = I;
        // (moved into a pass of its own over the tile's core, out of this loop, the block below makes the kernel SLOWER: 0.269 ->
        // 0.276 ms at 4096^2, gpurun_out/r4s17 -- fewer instructions, another schedule)
        if (a.Dx_out || a.I_mut) {                                   // wave-uniform: only the class API asks for these
            if (core && inside) {
                if (a.Dx_out) {
                    const int64_t Py = a.Ny + 2 * a.margin;
                    const int64_t q = (int64_t)(i + a.margin) * Py + (j + a.margin);
                    a.Dx_out[q] = fabsf(dx) < 1e-12f ? 0.f : dx;
                    a.Dy_out[q] = fabsf(dy) < 1e-12f ? 0.f : dy;
                }
                if (clamped && a.I_mut) a.I_mut[(int64_t)i * a.Ny + j] = 0.f;
            }
        }
        {
            const float wx = dx - fx, wy = dy - fy;                  // exact in float32
            // base target in ring coordinates (+1): the four shares land inside the accumulator iff 0 <= ti <= TH, 0 <= tj <= TW
#ifndef PSX_NEAR_EXP
#define PSX_NEAR_EXP 0
#endif
            // PSX_NEAR_EXP (tools/ab_refract4.sh, wrong images, timing only): 1 -> every ray deposits at its own pixel (no two lanes of
            // a wave share an address or a bank), 2 -> no stencil reads (a constant displacement), 3 -> both, 4 -> two of the four atomics
            const int ti = gr - H + ((PSX_NEAR_EXP & 1) ? 0 : ifx) + 1, tj = gc - H + ((PSX_NEAR_EXP & 1) ? 0 : ify) + 1;
            // ANY ray of the window whose base pixel falls in the accumulator is deposited, however long it is: the
            // share of a ray at target pixel t is gathered exactly when the source lies in the window of t's tile, and
            // k_refract_far applies the same test to decide what is left for it
            const float Is_ = I * fscale_f;                          // 2^s scaling is exact
            const bool hit = (unsigned)ti <= (unsigned)TH && (unsigned)tj <= (unsigned)TW;
            // a miss adds its shares to the trash area; 24-bit multiply + select, not a divergent branch
            int aidx = (int)__umul24((unsigned)ti, (unsigned)AW) + tj;
            asm volatile("" : "+v"(aidx));               // computed for every lane, then selected
            // a miss goes to the column its target WOULD have had: the lanes of a wave then keep their distinct banks whether
            // they hit or miss (at column = lane, the eight halo-column lanes of every row collided with hits three or four
            // lanes away -- a two-way conflict in nearly every deposit instruction)
            long long *acc = (dd ? sacc2 : sacc) + (hit ? aidx
                                         : G::MISS == 0 ? ACC + 2 * lane
                                         : G::MISS == 1 ? (TH + 1) * AW + lane
                                                        : (TH + 1) * AW + (tj & (AW - 1)));
            // float -> fixed point with one native conversion: the unit is 2^-30 of (the power of two above) the
            // largest staged intensity, so |v|*2^s <= 2^30 fits int32; the 64-bit sum has 2^33 of headroom
            auto dep = [&](int off, float v) __attribute__((always_inline)) {
                int qi;                                  // floor(v + 0.5) in ONE instruction (rintf + cvt are two)
                asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(qi) : "v"(v));
                atomicAdd((unsigned long long *)(acc + off), (unsigned long long)(long long)qi);
            };
            // Wide halos: a wave is one row of the 64-wide window, and the rows deep in the halo often hold no ray that reaches the
            // tile -- such a wave skips its four LDS atomics and their weights (a uniform branch; build-time A/B PSX_SKIP_MISS,
            // tools/ab_skipmiss.sh, gpurun_out/r5s32, two rounds on one box at 16384^2: halo 8 5.17 -> 5.08 ms, halo 12 6.37 -> 6.20,
            // halo 16 8.09 -> 7.76).  Not for the narrow halos, whose few halo rows sit next to the tile and nearly always hold a hit.
#ifndef PSX_SKIP_MISS
#define PSX_SKIP_MISS 1
#endif
            if (!(PSX_SKIP_MISS && H >= 8) || __any(hit)) {
#ifndef PSX_PK_WEIGHTS
#define PSX_PK_WEIGHTS 1
#endif
            if (PSX_PK_WEIGHTS) {
                // the same eight products, two per instruction (v_pk_mul_f32): bit for bit the scalar form's
                typedef float v2f __attribute__((ext_vector_type(2)));
                const v2f ax = {1.f - wx, wx};
                const float omy = 1.f - wy;
                const v2f s0 = (ax * (v2f){omy, omy}) * (v2f){Is_, Is_};      // {(1-wx)(1-wy), wx(1-wy)} * I
                const v2f s1 = (ax * (v2f){wy, wy}) * (v2f){Is_, Is_};        // {(1-wx)wy, wx wy} * I
                dep(0, s0.x);
                dep(AW, s0.y);
                if (!(PSX_NEAR_EXP & 4)) {           // timing experiment: two atomics per ray (what a packed pair of 32-bit fields would issue)
                    dep(1, s1.x);
                    dep(AW + 1, s1.y);
                }
            } else {
            dep(0, Is_ * ((1.f - wx) * (1.f - wy)));
            dep(AW, Is_ * (wx * (1.f - wy)));
            dep(1, Is_ * ((1.f - wx) * wy));
            dep(AW + 1, Is_ * (wx * wy));
            }
            }
            // (kept as the reference's products I * (wx-part * wy-part), RF2:241-262: regrouping them as (I * wy-part) * wx-part
            // would save two multiplies but round differently)
        }
        // wave-aggregated append of far rays to this tile's own list: one LDS atomic per wave, no global atomics
        // (a single global counter saturates at ~90 returning atomics per microsecond on this chip)
        const unsigned long long mask = __ballot(far);
        if (mask) {
            const int leader = __ffsll((long long)mask) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(dd ? sfar2 : sfar, (unsigned)__popcll(mask));
            base = __shfl(base, leader);
            if (far) {
                // set bits of `mask` below this lane: two v_mbcnt, formed here (the shift-and-popcount form was hoisted out of
                // this rare branch into every iteration of the deposit loop: a 64-bit shift and two v_not per source pixel; -0.8 %)
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                FarRay fr;
                fr.dx = Dxs; fr.dy = Dys; fr.I = Is; fr.src = i * a.Ny + j;
                (dd ? far_list1 : far_list)[base + rank] = fr;
                // order-independent replay: the scratch words this ray's shares may be added to start at zero.  (Round 4 did
                // this in a pass of its own over the lists, FAR_PREP: 0.020 ms at 4096^2, 1.12 ms on config 5.)  A superset of
                // what the replay deposits is fine -- only words that receive a share are ever read back.
                if (a.det_acc) {                                     // wave-uniform
                    long long *const accd = a.det_acc + (size_t)(dd ? d1 : d) * a.Nx * a.Ny;
                    int bi, ni, bj, nj;
                    float w0, w1, w2, w3;
                    axis_split_ref(Dxs, i, bi, ni, w0, w1);
                    axis_split_ref(Dys, j, bj, nj, w2, w3);
                    const bool bio = (unsigned)bi < (unsigned)a.Nx, nio = (unsigned)ni < (unsigned)a.Nx;
                    const bool bjo = (unsigned)bj < (unsigned)a.Ny, njo = (unsigned)nj < (unsigned)a.Ny;
                    if (bio && bjo) accd[(int64_t)bi * a.Ny + bj] = 0ll;
                    if (nio && bjo) accd[(int64_t)ni * a.Ny + bj] = 0ll;
                    if (nio && njo) accd[(int64_t)ni * a.Ny + nj] = 0ll;
                    if (bio && njo) accd[(int64_t)bi * a.Ny + nj] = 0ll;
                }
            }
        }
    }
    }   // distances of the pass
    };
    if (nd == 2)                                         // pairs imply an interior tile
        gather(std::true_type{}, std::integral_constant<int, 2>{});
    else if (window_inside)
        gather(std::true_type{}, std::integral_constant<int, 1>{});
    else
        gather(std::false_type{}, std::integral_constant<int, 1>{});
    PSX_RSTAMP(3);
    __syncthreads();
    PSX_RSTAMP(4);

    // ---- write the tile once (coalesced rows of TW floats) and clear it for the next distance (the guard ring
    // is never read, so it may keep what it collected)
    const bool more = d + nd < a.ndist;
    for (int dd = 0; dd < nd; ++dd) {
        long long *const accw = dd ? sacc2 : sacc;
        float *const Io = dd ? a.I_out[d1] : I_out;
        for (int idx = tl; idx < TH * TW; idx += NTHREADS) {
            const int tr = idx / TW, tc = idx - tr * TW;
            const int i = r0 + tr, j = c0 + tc;
            const long long q = accw[(tr + 1) * AW + tc + 1];
            if (more) accw[(tr + 1) * AW + tc + 1] = 0ll;
            if (i < a.Nx && j < a.Ny) {
                const int64_t p = (int64_t)i * a.Ny + j;
                float v = finite_in ? a.out_scale * (float)((double)q * finv) : __uint_as_float(0x7fc00000u);
                if (a.accumulate) v += Io[p];
                any_bad |= !(fabsf(v) <= 3.0e38f);
                Io[p] = v;
            }
        }
    }
    PSX_RSTAMP(5);
    if (tl == 0) {
        a.far_count[(size_t)d * nt + tile] = *sfar;     // the barrier above ordered every append before this read
        *sfar = 0u;
        if (nd == 2) {
            a.far_count[(size_t)d1 * nt + tile] = *sfar2;
            *sfar2 = 0u;
        }
    }
    if (more) __syncthreads();
    }   // distances
    if (a.status && __any(any_bad) && (tid & 63) == 0) atomicOr(a.status, PSX_STATUS_NONFINITE);
}

template <class G, int NM, bool HAS_I, bool HAS_PHI>
__global__ __launch_bounds__(G::NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_refract_near(RefractArgs a) {
    refract_near_body<G, NM, HAS_I, HAS_PHI>(a);
}

// the split call's own instantiations (phase from the thickness maps only): the plain kernels carry none of its tests
template <class G, int NM, bool HAS_I>
__global__ __launch_bounds__(G::NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_refract_near_split(RefractArgs a) {
    refract_near_body<G, NM, HAS_I, false, true>(a);
}

// A batch of refractions in one launch -- the energies of a detector bin (EXP:448-486): refraction e = blockIdx.y has its own
// argument block (input intensity or uniform I0, coefficients, displacement scale, output image, far-ray lists) in the
// kernel-argument segment, which holds REFRACT_TAB of them; the kernels read theirs in place (scalar loads at a computed
// offset -- a copy of the block would live in scratch memory).
constexpr int REFRACT_TAB = 8;
struct RefractTab {
    RefractArgs e[REFRACT_TAB];
};
static_assert(sizeof(RefractTab) <= 4096, "kernel arguments are limited to 4 KiB");

// The block of refraction e as a LOCAL structure whose arrays are only ever touched at compile-time indices (one distance,
// NM materials): it is scalarised into registers.  Handing the bodies a reference into the argument segment at a run-time
// offset instead made the compiler re-load the arguments wherever they are used (six times the kernel time), and a plain
// copy of the block lives in scratch memory (its distance arrays are indexed by a loop variable).
template <int NM>
__device__ __forceinline__ RefractArgs one_distance_block(const RefractTab &t, int e) {
    const RefractArgs &s = t.e[e];
    RefractArgs a;
    a.I_in = s.I_in; a.mask = nullptr; a.I0 = s.I0; a.phi_in = nullptr;
    a.m.n = NM;
#pragma unroll
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) {
        a.m.T[i] = s.m.T[i];
        a.m.cphase[i] = s.m.cphase[i];
        a.m.catt[i] = s.m.catt[i];
    }
    a.I_out[0] = s.I_out[0]; a.dscale[0] = s.dscale[0]; a.ndist = 1;
    a.out_scale = s.out_scale; a.accumulate = s.accumulate;
    a.Dx_out = nullptr; a.Dy_out = nullptr; a.I_mut = nullptr;
    a.Nx = s.Nx; a.Ny = s.Ny; a.margin = s.margin; a.clamp_xf = s.clamp_xf; a.clamp_yf = s.clamp_yf;
    a.status = s.status; a.far_count = s.far_count; a.far_list = s.far_list;
    a.tiles_x = s.tiles_x; a.tiles_y = s.tiles_y; a.tile_cap = s.tile_cap; a.far_stride = s.far_stride; a.stamps = nullptr;
    a.det_gmax = s.det_gmax; a.det_scale_bits = s.det_scale_bits; a.det_fold_count = s.det_fold_count; a.det_acc = s.det_acc;
    return a;
}

template <class G, int NM, bool HAS_I>
__global__ __launch_bounds__(G::NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_refract_near_batch(RefractTab t) {
    const RefractArgs a = one_distance_block<NM>(t, blockIdx.y);
    refract_near_body<G, NM, HAS_I, false>(a);
}


// Replay of the far rays with the reference's literal border rules (RF2:235-262) in padded coordinates.
// One WAVE per list (list = distance * ntiles + tile): a list holds a few dozen records and the kernel is a chain of
// dependent latencies (count -> records -> atomics), so it wants as many lists in flight per CU as there are wave slots.
//
// Order-independent form (psx_set_deterministic; SURVEY.md section 5, "race detection"): the
// reference deposits in raster order, float atomics deposit in whatever order the waves arrive, and a far ray's last bit can
// flip a Poisson draw downstream.  With the mode on the replay is TWO passes over the lists (three in round 4), neither of
// which allocates, synchronises or relies on anything a previous call left behind:
//   (the tile kernel, when it appends a far ray, stores 0 into the 64-bit scratch words of the four pixels its shares may
//    reach -- scratch = [ndist][Nx*Ny] words inside the caller's workspace; only touched words are ever looked at, so the
//    region needs no initial state -- and folds its largest staged intensity into one word per call, det_gmax;)
//   FAR_ADD   a share goes into its pixel's word as a fixed-point integer with a RETURNING atomic add.  ONE unit per call:
//             2^-30 of the power of two above det_gmax -- every share is a product of a staged intensity and weights <= 1,
//             so |share| <= 2^30 units whatever tile it lands in: no tile without a unit, no share that does not fit (round
//             4 took the unit of the TARGET's tile and fell back to float atomics for dark targets and large shares: ADVICE
//             r4).  The word is [12-bit deposit counter | 52-bit two's-complement sum]: every deposit also adds 1 << 52, so
//             the word is zero only before the first deposit and the thread that reads back 0 knows it was first: it appends
//             the pixel to the list's FOLD TABLE (wave ballot, no atomic), written over the records the wave has consumed;
//   FAR_FOLD  walks the fold tables: the complete sum of a pixel (plain load: the kernel boundary ordered every add before it)
//             is added, ONCE, to the float image the tile kernel wrote.
// float(tile sum) + float(far sum) is then a function of the inputs alone: two runs are bitwise equal, on any number of GPUs.
// One atomic per share, as in the float form.  Headroom: 2^21 shares of the call's largest intensity per pixel.
struct DetAcc {
    long long *acc;          // psx_fastloop_f32's deterministic mode (see below)
    const unsigned *mx;
};
enum { FAR_FLOAT = 0, FAR_ADD = 2, FAR_FOLD = 3 };
constexpr unsigned long long DET_ONE = 1ull << 52, DET_MASK = DET_ONE - 1ull;

// exponent s of the call's fixed-point unit 2^-s: the largest staged intensity times 2^s lies in [2^29, 2^30)
__device__ __forceinline__ int det_unit_exp(unsigned gmax_bits) {
    return min(120, max(-120, 30 - (gmax_bits ? ilogbf(__uint_as_float(gmax_bits)) + 1 : 0)));
}

// psx_fastloop_f32's deterministic mode (no workspace there: scratch image + max word allocated per call)
__device__ __forceinline__ double det_scale(const unsigned *mx) {
    int ex;
    frexpf(__uint_as_float(*mx), &ex);
    return ldexp(1.0, 38 - ex);
}

__global__ __launch_bounds__(256) void k_det_apply(float *out, const long long *acc, const unsigned *mx, int64_t n) {
    const double inv = 1.0 / det_scale(mx);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const long long q = acc[p];
        if (q != 0) out[p] += (float)((double)q * inv);
    }
}

template <class G, int MODE = FAR_FLOAT, bool ONE = false>      // ONE: a single distance (its index is then a constant)
__device__ __forceinline__ void refract_far_body(const RefractArgs &a) {
    constexpr int TH = G::TH, TW = G::TW, H = G::H;
    const unsigned nlists = (unsigned)(a.tiles_x * a.tiles_y) * (unsigned)a.ndist;
    unsigned lst = blockIdx.x * FAR_LISTS + threadIdx.x / FAR_SUB;
    if (lst >= nlists) return;
    if (a.far_stride) lst = (unsigned)(((unsigned long long)lst * a.far_stride) % nlists);     // stride coprime to nlists (host)
    const unsigned lane = threadIdx.x % FAR_SUB;
    const unsigned dist = ONE ? 0u : lst / (unsigned)(a.tiles_x * a.tiles_y);
    float *const I_out = a.I_out[ONE ? 0 : dist];
    (void)dist;
    long long *const acc = MODE == FAR_FLOAT ? nullptr : a.det_acc + (size_t)dist * a.Nx * a.Ny;
    FarRay *const list = a.far_list + (size_t)lst * (TH * TW);
    if constexpr (MODE == FAR_FOLD) {
        const unsigned nf = a.det_fold_count[lst];
        if (nf == 0) return;
        const unsigned *fold = reinterpret_cast<const unsigned *>(list);
        const double inv = ldexp(1.0, -det_unit_exp(a.det_scale_bits ? a.det_scale_bits : *a.det_gmax));
        bool bad = false;
        for (unsigned e = lane; e < nf; e += FAR_SUB) {
            const unsigned p = fold[e];
            const long long w = acc[p];
            const long long sum = (long long)((unsigned long long)w << 12) >> 12;      // sign-extend the 52-bit sum
            const float add = a.out_scale * (float)((double)sum * inv);
            bad |= !(fabsf(add) <= 3.0e38f);
            I_out[p] += add;                                       // the only thread that owns this pixel's sum
        }
        if (a.status && bad) atomicOr(a.status, PSX_STATUS_NONFINITE);
        return;
    }
    const unsigned n = a.far_count[lst];
    if (n == 0) {
        if (MODE == FAR_ADD && lane == 0) a.det_fold_count[lst] = 0u;
        return;
    }
    const int Px = a.Nx + 2 * a.margin, Py = a.Ny + 2 * a.margin;
    const float unit = MODE == FAR_ADD ? ldexpf(1.f, det_unit_exp(a.det_scale_bits ? a.det_scale_bits : *a.det_gmax)) : 0.f;     // exact scaling
    unsigned *const fold = reinterpret_cast<unsigned *>(list);     // FAR_ADD: fold table, over the records already consumed --
    unsigned nf = 0;                                               // after t rounds it holds <= 4 * 64 t entries of 4 bytes = the
    (void)unit; (void)fold;                                        // 64 t records of 16 bytes every lane has read (wave-uniform nf)
    // uniform trip count: FAR_ADD's rounds hold wave ballots
    for (unsigned e0 = 0; e0 < n; e0 += FAR_SUB) {
        const unsigned e = e0 + lane;
        const bool live = e < n;
        FarRay fr = list[live ? e : n - 1];
        const int i = fr.src / a.Ny, j = fr.src - i * a.Ny;
        const float I = live ? fr.I : 0.f;                       // a share of 0 deposits nothing
        int bi, ni, bj, nj;
        float wbi, wni, wbj, wnj;
        axis_split_ref(fr.dx, i + a.margin, bi, ni, wbi, wni);
        axis_split_ref(fr.dy, j + a.margin, bj, nj, wbj, wnj);
        unsigned first[4] = {~0u, ~0u, ~0u, ~0u};                // pixels whose first deposit was this lane's share k
        (void)first;
        if (bi >= 0 && bi < Px && bj >= 0 && bj < Py) {                  // RF2:235-236
            auto deposit = [&](int k, int pi, int pj, float v) {
                const int ui = pi - a.margin, uj = pj - a.margin;           // crop (RF2:78)
                if (ui >= 0 && ui < a.Nx && uj >= 0 && uj < a.Ny && v != 0.f) {
                    // already deposited by the gather of the target's tile iff the source lies in that tile's window
                    const int tr = ui / TH, tc = uj / TW, r0 = tr * TH, c0 = tc * TW;
                    if (i >= r0 - H && i < r0 + TH + H && j >= c0 - H && j < c0 + TW + H) return;
                    const int64_t p = (int64_t)ui * a.Ny + uj;
                    if constexpr (MODE == FAR_FLOAT) {
                        const float add = a.out_scale * v;
                        if (a.status && !(fabsf(add) <= 3.0e38f)) atomicOr(a.status, PSX_STATUS_NONFINITE);
                        atomicAdd(&I_out[p], add);
                    } else {
                        if (!(fabsf(v) <= 3.0e38f)) {                          // a NaN / inf source: the tile kernel has raised the status
                            if (a.status) atomicOr(a.status, PSX_STATUS_NONFINITE);
                            return;
                        }
                        const float xq = v * unit;                              // |xq| <= 2^30 when the unit comes from the call's maximum
                        // a caller's scale 2^10 times too small is an "insane value": one share stays below 2^40 units, so the 2^11
                        // shares a pixel's 52-bit signed sum is then sure to hold outnumber what the 12-bit deposit counter admits of
                        // such extremes only by a factor of two -- and a share of the class's own chain (an intensity times weights
                        // <= 1, the scale being the incident intensity) is below 2^36 (the bound was 2^50, where two
                        // shares on one pixel wrapped into the counter unnoticed)
                        if (!(fabsf(xq) < 1.0995116e12f)) {
                            if (a.status) atomicOr(a.status, PSX_STATUS_NONFINITE);
                            return;
                        }
                        const long long q = llrintf(xq);
                        if (q == 0) return;                                     // below the unit: contributes nothing
                        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(acc + p),
                                                                 DET_ONE + ((unsigned long long)q & DET_MASK));
                        if (old == 0ull) first[k] = (unsigned)p;
                    }
                }
            };
            deposit(0, bi, bj, I * wbi * wbj);
            if (ni >= 0 && ni < Px && nj >= 0 && nj < Py) {                  // RF2:238-262: all three or none
                deposit(1, ni, bj, I * wni * wbj);
                deposit(2, ni, nj, I * wni * wnj);
                deposit(3, bi, nj, I * wbi * wnj);
            }
        }
        if constexpr (MODE == FAR_ADD) {
            // (the records of this round have been read by every lane: the ballots below wait for every lane's atomics, which
            // wait for its record)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool f = first[k] != ~0u;
                const unsigned long long mask = __ballot(f);
                if (f) fold[nf + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))] = first[k];
                nf += (unsigned)__popcll(mask);
            }
        }
    }
    if (MODE == FAR_ADD && lane == 0) a.det_fold_count[lst] = nf;
}

template <class G, int MODE = FAR_FLOAT>
__global__ __launch_bounds__(FAR_THREADS) void k_refract_far(RefractArgs a) {
    refract_far_body<G, MODE>(a);
}

template <class G, int MODE = FAR_FLOAT>
__global__ __launch_bounds__(FAR_THREADS) void k_refract_far_batch(RefractTab t) {
    const RefractArgs a = one_distance_block<0>(t, blockIdx.y);
    refract_far_body<G, MODE, true>(a);
}


__global__ __launch_bounds__(256) void k_absmax(const float *__restrict__ v, int64_t n, unsigned *mx) {
    float m = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const float x = fabsf(v[p]);
        if (x <= 3.0e38f) m = fmaxf(m, x);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(mx, __float_as_uint(m));
}

// fastloopNumba on explicit displacement maps: literal branch structure, global float atomics (DET: 64-bit fixed-point
// deposits into `det.acc`, see the deterministic mode above).
template <bool DET>
__global__ __launch_bounds__(256) void k_fastloop(const float *__restrict__ I, const float *__restrict__ Dx,
                                                       const float *__restrict__ Dy, float *I2, int Nx, int Ny, DetAcc det) {
    const int64_t n = (int64_t)Nx * Ny;
    const double fix = DET ? det_scale(det.mx) : 0.0;
    auto atomicAdd = [&](float *addr, float v) {       // shadows the builtin inside this kernel
        if constexpr (DET) {
            if (fabsf(v) <= 3.0e38f)
                ::atomicAdd(reinterpret_cast<unsigned long long *>(det.acc + (addr - I2)), (unsigned long long)llrint((double)v * fix));
        } else {
            ::atomicAdd(addr, v);
        }
    };
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(p / Ny), j = (int)(p - (int64_t)i * Ny);
        const float Iij = I[p];
        const double dx = Dx[p], dy = Dy[p];
        if (dx == 0.0 && dy == 0.0) {                                  // RF2:222-224
            atomicAdd(&I2[p], Iij);
            continue;
        }
        int bi, ni, bj, nj;
        float wbi, wni, wbj, wnj;
        axis_split_ref(dx, i, bi, ni, wbi, wni);
        axis_split_ref(dy, j, bj, nj, wbj, wnj);
        if (bi < 0 || bi >= Nx || bj < 0 || bj >= Ny) continue;
        atomicAdd(&I2[(int64_t)bi * Ny + bj], Iij * wbi * wbj);
        if (ni < 0 || ni >= Nx || nj < 0 || nj >= Ny) continue;
        atomicAdd(&I2[(int64_t)ni * Ny + bj], Iij * wni * wbj);
        atomicAdd(&I2[(int64_t)ni * Ny + nj], Iij * wni * wnj);
        atomicAdd(&I2[(int64_t)bi * Ny + nj], Iij * wbi * wnj);
    }
}

// Which geometry a call uses.  The wide halo costs ~20 % more source evaluations but keeps rays displaced by up to
// 8 pixels inside the LDS gather; the far replay (scattered global float atomics, ~0.1 TB/s) is what it avoids.
// Per HOST THREAD (the ABI's thread model is one host thread per GPU): a second thread driving another GPU keeps its own.
thread_local int g_refract_geometry = 0;   // 0: GeoSmall (H=4), 1: GeoWide (H=8), 2: GeoMid (H=6), 3: GeoH12, 4: GeoH16
thread_local int g_deterministic = 0;      // psx_set_deterministic
thread_local float g_det_scale = 0.f;      // psx_set_deterministic_scale (0: the unit comes from the call's measured maximum)
static unsigned det_scale_bits() {         // the scale with 2^6 of room for shares above it, as float bits (0: measure)
    if (!(g_det_scale > 0.f) || !(g_det_scale < 1e30f)) return 0u;
    const float s = g_det_scale * 64.f;
    unsigned b;
    std::memcpy(&b, &s, sizeof b);
    return b;
}

// scratch of psx_fastloop_f32's deterministic mode (that entry point has no workspace argument): accumulators + the max
// word, owned for the duration of one call -- a plain hipMalloc / synchronise / hipFree per call, a debugging aid.  The
// refraction entry points take theirs from the caller's workspace and allocate nothing (refract_far_body).
struct DetScratch {
    long long *acc = nullptr;
    unsigned *mx = nullptr;
    int alloc(size_t npix, hipStream_t st) {
        PSX_HIP(hipMalloc((void **)&acc, sizeof(long long) * npix + 16));
        mx = reinterpret_cast<unsigned *>(acc + npix);
        PSX_HIP(hipMemsetAsync(acc, 0, sizeof(long long) * npix + 16, st));
        return 0;
    }
    int release(hipStream_t st) {
        PSX_HIP(hipStreamSynchronize(st));
        PSX_HIP(hipFree(acc));
        acc = nullptr;
        return 0;
    }
};

// diagnostics: a stride for the replay's walk over the lists that is coprime to their number (0: the plain tile order)
static unsigned far_stride_for(unsigned nlists) {
    unsigned s = (unsigned)debug_switch(DBG_FAR_STRIDE);
    if (s == 0 || nlists < 2) return 0;
    auto gcd = [](unsigned x, unsigned y) { while (y) { const unsigned t = x % y; x = y; y = t; } return x; };
    s %= nlists;
    while (s < 2 || gcd(s, nlists) != 1) ++s;
    return s;
}

// far-ray counters and lists of a call: [ndist][ntiles] counts, then [ndist][ntiles][TH*TW] records
template <class G>
size_t lists_bytes(int Nx, int Ny, int ndist) {
    const size_t nt = (size_t)cdiv(Nx, G::TH) * (size_t)cdiv(Ny, G::TW) * (size_t)ndist;
    return 16 * ((sizeof(unsigned) * nt + 15) / 16) + sizeof(FarRay) * nt * G::TH * G::TW;
}
// + (order-independent replay) behind the lists: the call's maximum word (16 bytes), a fold-table count per list,
// [ndist][Nx*Ny] scratch words
inline size_t pad16(size_t b) { return (b + 15) / 16 * 16; }
template <class G>
size_t det_bytes(int Nx, int Ny, int ndist) {
    const size_t nt = (size_t)cdiv(Nx, G::TH) * (size_t)cdiv(Ny, G::TW);
    return 16 + pad16(sizeof(unsigned) * nt * ndist) + sizeof(long long) * (size_t)Nx * (size_t)Ny * (size_t)ndist;
}
template <class G>
void det_pointers(RefractArgs &a, char *gmax, char *rest, int ndist) {      // rest: det_bytes() - 16 bytes
    const size_t nt = (size_t)a.tiles_x * a.tiles_y;
    a.det_gmax = (unsigned *)gmax;
    a.det_fold_count = (unsigned *)rest;
    a.det_acc = (long long *)(rest + pad16(sizeof(unsigned) * nt * ndist));
}
template <class G>
size_t workspace_for(int Nx, int Ny, int ndist) {
    return lists_bytes<G>(Nx, Ny, ndist) + (g_deterministic ? det_bytes<G>(Nx, Ny, ndist) : 0);
}

// one chunk of a batch: the lists of REFRACT_TAB refractions, then (order-independent replay) REFRACT_TAB one-distance regions
template <class G>
size_t batch_chunk_bytes(int Nx, int Ny) {
    return lists_bytes<G>(Nx, Ny, REFRACT_TAB) + (g_deterministic ? (size_t)REFRACT_TAB * det_bytes<G>(Nx, Ny, 1) : 0);
}
static size_t batch_chunk_max(int Nx, int Ny) {
    return PSX_GEO_MAX(batch_chunk_bytes<G_>(Nx, Ny));
}

template <class G>
int launch_refract(RefractArgs &a, const float *I_in, const double *phi_in, int nmat, void *workspace, hipStream_t st) {
    a.tiles_x = (int)cdiv(a.Nx, G::TH);
    a.tiles_y = (int)cdiv(a.Ny, G::TW);
    a.tile_cap = G::TH * G::TW;
    a.far_stride = far_stride_for((unsigned)(a.tiles_x * a.tiles_y * a.ndist));
    a.far_count = (unsigned *)workspace;
    a.far_list = (FarRay *)((char *)workspace + 16 * ((sizeof(unsigned) * (size_t)a.tiles_x * a.tiles_y * a.ndist + 15) / 16));
    a.det_gmax = nullptr; a.det_fold_count = nullptr; a.det_acc = nullptr; a.det_scale_bits = 0u;
    if (g_deterministic) {
        char *const det = (char *)workspace + lists_bytes<G>(a.Nx, a.Ny, a.ndist);
        det_pointers<G>(a, det, det + 16, a.ndist);
        a.det_scale_bits = det_scale_bits();
        if (!a.det_scale_bits)
            PSX_HIP(hipMemsetAsync(a.det_gmax, 0, 16, st));      // the only word of the mode with an initial state: set per call
    }
    int rc_launch = 0;
    auto launch = [&](auto nm, auto hi, auto hp) -> int {
        constexpr int NM = decltype(nm)::value;
        constexpr bool HI = decltype(hi)::value, HP = decltype(hp)::value;
        if constexpr (!HP && NM > 0) {
            if (a.mask) {
                static std::atomic<unsigned long long> attr_mask_s{0};
                if (first_on_device(attr_mask_s))
                    PSX_HIP(hipFuncSetAttribute((const void *)k_refract_near_split<G, NM, HI>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS));
                PSX_TIMED("k_refract_near", st,
                          k_refract_near_split<G, NM, HI><<<a.tiles_x * a.tiles_y, G::NT, G::LDS, st>>>(a));
            }
        }
        if (!a.mask) {
            static std::atomic<unsigned long long> attr_mask{0};
            if (first_on_device(attr_mask))
                PSX_HIP(hipFuncSetAttribute((const void *)k_refract_near<G, NM, HI, HP>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            // diagnostics (psx_debug_switch "near_lds_pad"): more LDS than the tile needs, i.e. one workgroup per CU
            const size_t lds = std::min<size_t>(160 * 1024, G::LDS + 1024 * (size_t)std::max(0, debug_switch(DBG_NEAR_LDS_PAD)));
            PSX_TIMED("k_refract_near", st,
                      k_refract_near<G, NM, HI, HP><<<a.tiles_x * a.tiles_y, G::NT, lds, st>>>(a));
        }
        if (int rc = launch_check("k_refract_near")) return rc;
        const int nlists = a.tiles_x * a.tiles_y * a.ndist;
        const int fgrid = (nlists + FAR_LISTS - 1) / FAR_LISTS;
        if (a.det_acc) {      // order-independent replay: two passes over the lists, scratch from the workspace
            PSX_TIMED("k_refract_far_add", st, k_refract_far<G, FAR_ADD><<<fgrid, FAR_THREADS, 0, st>>>(a));
            PSX_TIMED("k_refract_far_fold", st, k_refract_far<G, FAR_FOLD><<<fgrid, FAR_THREADS, 0, st>>>(a));
            return 0;
        }
        PSX_TIMED("k_refract_far", st, k_refract_far<G><<<fgrid, FAR_THREADS, 0, st>>>(a));
        return 0;
    };
    PSX_DISPATCH_NMAT(nmat, {
        using N_ = std::integral_constant<int, NM>;
        if (I_in && phi_in) rc_launch = launch(N_{}, std::true_type{}, std::true_type{});
        else if (I_in) rc_launch = launch(N_{}, std::true_type{}, std::false_type{});
        else if (phi_in) rc_launch = launch(N_{}, std::false_type{}, std::true_type{});
        else rc_launch = launch(N_{}, std::false_type{}, std::false_type{});
    });
    if (rc_launch) return rc_launch;
    return launch_check("k_refract_far");
}

template <class G>
int launch_refract_batch(RefractTab &t, int n, bool has_I, int nmat, void *workspace, hipStream_t st) {
    const int tiles_x = (int)cdiv(t.e[0].Nx, G::TH), tiles_y = (int)cdiv(t.e[0].Ny, G::TW);
    const size_t nt = (size_t)tiles_x * tiles_y;
    for (int e = 0; e < REFRACT_TAB; ++e) {
        RefractArgs &a = t.e[e];
        a.tiles_x = tiles_x; a.tiles_y = tiles_y; a.tile_cap = G::TH * G::TW; a.far_stride = far_stride_for((unsigned)nt);
        const int k = e < n ? e : 0;
        a.far_count = (unsigned *)workspace + (size_t)k * nt;
        a.far_list = (FarRay *)((char *)workspace + 16 * ((sizeof(unsigned) * nt * REFRACT_TAB + 15) / 16)) + (size_t)k * nt * a.tile_cap;
        // order-independent replay: every refraction of the chunk its own exponents, marks and scratch words behind the chunk's lists
        a.det_gmax = nullptr; a.det_fold_count = nullptr; a.det_acc = nullptr; a.det_scale_bits = g_deterministic ? det_scale_bits() : 0u;
        if (g_deterministic) {    // the chunk's maximum words side by side (one memset node), then each refraction's counts and words
            char *const det = (char *)workspace + lists_bytes<G>(a.Nx, a.Ny, REFRACT_TAB);
            det_pointers<G>(a, det + 16 * k, det + 16 * REFRACT_TAB + (size_t)k * (det_bytes<G>(a.Nx, a.Ny, 1) - 16), 1);
        }
    }
    if (g_deterministic && !t.e[0].det_scale_bits)      // every refraction of the chunk its own maximum word (its own unit: an image of its own)
        PSX_HIP(hipMemsetAsync(t.e[0].det_gmax, 0, 16 * REFRACT_TAB, st));
    int rc_launch = 0;
    auto launch = [&](auto nm, auto hi) -> int {
        constexpr int NM = decltype(nm)::value;
        constexpr bool HI = decltype(hi)::value;
        static std::atomic<unsigned long long> attr_mask{0};
        if (first_on_device(attr_mask))
            PSX_HIP(hipFuncSetAttribute((const void *)k_refract_near_batch<G, NM, HI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)G::LDS));
        PSX_TIMED("k_refract_near", st, k_refract_near_batch<G, NM, HI><<<dim3((unsigned)nt, (unsigned)n), G::NT, G::LDS, st>>>(t));
        if (int rc = launch_check("k_refract_near")) return rc;
        const int fgrid = ((int)nt + FAR_LISTS - 1) / FAR_LISTS;
        const dim3 fg((unsigned)fgrid, (unsigned)n);
        if (t.e[0].det_acc) {
            PSX_TIMED("k_refract_far_add", st, k_refract_far_batch<G, FAR_ADD><<<fg, FAR_THREADS, 0, st>>>(t));
            PSX_TIMED("k_refract_far_fold", st, k_refract_far_batch<G, FAR_FOLD><<<fg, FAR_THREADS, 0, st>>>(t));
        } else {
            PSX_TIMED("k_refract_far", st, k_refract_far_batch<G><<<fg, FAR_THREADS, 0, st>>>(t));
        }
        return launch_check("k_refract_far");
    };
    PSX_DISPATCH_NMAT(nmat, {
        using N_ = std::integral_constant<int, NM>;
        if (has_I) rc_launch = launch(N_{}, std::true_type{});
        else rc_launch = launch(N_{}, std::false_type{});
    });
    return rc_launch;
}

}  // namespace

extern "C" {

size_t psx_refract_multi_workspace_bytes(int Nx, int Ny, int ndist) {
    if (Nx <= 0 || Ny <= 0 || ndist <= 0) return 16;
    return PSX_GEO_MAX(workspace_for<G_>(Nx, Ny, ndist));
}

size_t psx_refract_workspace_bytes(int Nx, int Ny) { return psx_refract_multi_workspace_bytes(Nx, Ny, 1); }

int psx_refract_set_halo(int halo) {
    PSX_REQUIRE(halo == 4 || halo == 6 || halo == 8 || halo == 12 || halo == 16,
                "psx_refract_set_halo: halo must be 4, 6, 8, 12 or 16, got %d", halo);
    g_refract_geometry = halo == 8 ? 1 : halo == 6 ? 2 : halo == 12 ? 3 : halo == 16 ? 4 : 0;
    return 0;
}

static int refract_multi_impl(const float *I_in, const float *mask, float I0, const float *const *T,
                              const double *cphase, const double *catt, int nmat, const double *phi_in, float *const *I_out,
                              float out_scale, int accumulate, float *Dx_out, float *Dy_out, float *I_mut, int Nx, int Ny,
                              int margin, const double *dscale, int ndist, double clamp_x, double clamp_y, unsigned *status,
                              void *workspace, void *stream) {
    PSX_REQUIRE(I_out != nullptr && dscale != nullptr && workspace != nullptr, "psx_refract_multi_f32: null outputs, distances or workspace");
    PSX_REQUIRE(ndist >= 1 && ndist <= PSX_MAX_DIST, "psx_refract_multi_f32: %d distances, 1..%d supported per call", ndist, PSX_MAX_DIST);
    PSX_REQUIRE(Nx >= 3 && Ny >= 3, "psx_refract_multi_f32: grid %dx%d too small for the edge_order=2 gradient", Nx, Ny);
    PSX_REQUIRE((int64_t)Nx * Ny < (1ll << 31), "psx_refract_multi_f32: grid %dx%d exceeds int32 pixel indices", Nx, Ny);
    PSX_REQUIRE(margin >= 8 && margin <= 4096, "psx_refract_multi_f32: margin %d must be >= 8 (the widest gather halo)", margin);
    PSX_REQUIRE((Dx_out == nullptr) == (Dy_out == nullptr), "psx_refract_multi_f32: Dx_out and Dy_out go together");
    PSX_REQUIRE(ndist == 1 || (Dx_out == nullptr && I_mut == nullptr),
                "psx_refract_multi_f32: displacement maps / input mutation belong to ONE distance (got %d)", ndist);
    PSX_REQUIRE(nmat > 0 || phi_in != nullptr, "psx_refract_multi_f32: no phase source (nmat=0 and phi_in=NULL)");
    RefractArgs a;
    if (int rc = pack_mats(a.m, T, cphase, catt, nmat)) return rc;
    hipStream_t st = (hipStream_t)stream;
    for (int d = 0; d < PSX_MAX_DIST; ++d) {
        const int e = d < ndist ? d : 0;
        PSX_REQUIRE(I_out[e] != nullptr, "psx_refract_multi_f32: null output image %d", e);
        for (int f = 0; f < e; ++f)
            PSX_REQUIRE(I_out[f] != I_out[e], "psx_refract_multi_f32: distances %d and %d share an output image", f, e);
        a.I_out[d] = I_out[e];
        a.dscale[d] = dscale[e];
    }
    a.ndist = ndist;
    a.I_in = I_in; a.mask = mask; a.I0 = I0; a.phi_in = phi_in; a.out_scale = out_scale; a.accumulate = accumulate;
    a.Dx_out = Dx_out; a.Dy_out = Dy_out; a.I_mut = I_mut; a.Nx = Nx; a.Ny = Ny; a.margin = margin;
    a.clamp_xf = (float)clamp_x; a.clamp_yf = (float)clamp_y; a.status = status; a.stamps = g_stamps;
    if (Dx_out) {
        const size_t padded = sizeof(float) * (size_t)(Nx + 2 * margin) * (size_t)(Ny + 2 * margin);
        PSX_HIP(hipMemsetAsync(Dx_out, 0, padded, st));     // zero margins (RF2:65-66)
        PSX_HIP(hipMemsetAsync(Dy_out, 0, padded, st));
    }
    return PSX_GEO_DISPATCH(G_, launch_refract<G_>(a, I_in, phi_in, nmat, workspace, st));
}

int psx_refract_multi_f32(const float *I_in, float I0, const float *const *T, const double *cphase, const double *catt,
                          int nmat, const double *phi_in, float *const *I_out, float out_scale, int accumulate,
                          float *Dx_out, float *Dy_out, float *I_mut, int Nx, int Ny, int margin, const double *dscale,
                          int ndist, double clamp_x, double clamp_y, unsigned *status, void *workspace, void *stream) {
    return refract_multi_impl(I_in, nullptr, I0, T, cphase, catt, nmat, phi_in, I_out, out_scale, accumulate, Dx_out, Dy_out,
                              I_mut, Nx, Ny, margin, dscale, ndist, clamp_x, clamp_y, status, workspace, stream);
}

int psx_refract_split_f32(const float *I_in, const float *mask, float I0, const float *const *T, const double *cphase,
                          const double *catt, int nmat, const double *phi_in, float *I_out_zero, float *I_out_nonzero,
                          float out_scale, int accumulate, int Nx, int Ny, int margin, double dscale, double clamp_x,
                          double clamp_y, unsigned *status, void *workspace, void *stream) {
    PSX_REQUIRE(I_out_zero != nullptr && I_out_nonzero != nullptr && mask != nullptr, "psx_refract_split_f32: null output or mask");
    PSX_REQUIRE(phi_in == nullptr && nmat > 0, "psx_refract_split_f32: the phase comes from the thickness maps (nmat > 0, phi_in = NULL)");
    float *const outs[2] = {I_out_zero, I_out_nonzero};
    const double ds[2] = {dscale, dscale};
    return refract_multi_impl(I_in, mask, I0, T, cphase, catt, nmat, phi_in, outs, out_scale, accumulate, nullptr, nullptr, nullptr,
                              Nx, Ny, margin, ds, 2, clamp_x, clamp_y, status, workspace, stream);
}

int psx_refract_f32(const float *I_in, float I0, const float *const *T, const double *cphase, const double *catt,
                    int nmat, const double *phi_in, float *I_out, float out_scale, int accumulate, float *Dx_out,
                    float *Dy_out, float *I_mut, int Nx, int Ny, int margin, double dscale, double clamp_x,
                    double clamp_y, unsigned *status, void *workspace, void *stream) {
    PSX_REQUIRE(I_out != nullptr, "psx_refract_f32: null output");
    return psx_refract_multi_f32(I_in, I0, T, cphase, catt, nmat, phi_in, &I_out, out_scale, accumulate, Dx_out, Dy_out,
                                 I_mut, Nx, Ny, margin, &dscale, 1, clamp_x, clamp_y, status, workspace, stream);
}

int psx_fastloop_f32(const float *I, const float *Dx, const float *Dy, float *I2, int Nx, int Ny, void *stream) {
    PSX_REQUIRE(I && Dx && Dy && I2 && Nx > 0 && Ny > 0, "psx_fastloop_f32: null pointer or empty grid");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)Nx * Ny;
    if (g_deterministic) {
        DetScratch ds;
        if (int rc = ds.alloc((size_t)n, st)) return rc;
        PSX_TIMED("k_absmax", st, k_absmax<<<ew_grid(n, 256), 256, 0, st>>>(I, n, ds.mx));
        PSX_TIMED("k_fastloop", st, k_fastloop<true><<<ew_grid(n, 256), 256, 0, st>>>(I, Dx, Dy, I2, Nx, Ny, DetAcc{ds.acc, ds.mx}));
        PSX_TIMED("k_det_apply", st, k_det_apply<<<ew_grid(n, 256), 256, 0, st>>>(I2, ds.acc, ds.mx, n));
        const int rc_det = launch_check("k_fastloop (deterministic)");
        const int rc_rel = ds.release(st);
        return rc_det ? rc_det : rc_rel;
    }
    PSX_TIMED("k_fastloop", st, k_fastloop<false><<<ew_grid(n, 256), 256, 0, st>>>(I, Dx, Dy, I2, Nx, Ny, DetAcc{nullptr, nullptr}));
    return launch_check("k_fastloop");
}

int psx_set_deterministic(int on) {
    g_deterministic = on ? 1 : 0;
    return 0;
}

int psx_get_deterministic(void) { return g_deterministic; }

int psx_set_deterministic_scale(float scale) {
    PSX_REQUIRE(scale >= 0.f && scale < 1e30f, "psx_set_deterministic_scale: scale %g outside [0, 1e30)", (double)scale);
    g_det_scale = scale;
    return 0;
}

float psx_get_deterministic_scale(void) { return g_det_scale; }

size_t psx_refract_batch_workspace_bytes(int Nx, int Ny, int n) {
    if (Nx <= 0 || Ny <= 0) return 16;
    const int chunks = (std::max(n, 1) + REFRACT_TAB - 1) / REFRACT_TAB;
    return (size_t)chunks * batch_chunk_max(Nx, Ny);
}

int psx_refract_batch_f32(int n, const float *const *I_in, const float *I0, const float *const *T, const double *cphase,
                          const double *catt, int nmat, float *const *I_out, float out_scale, int accumulate, int Nx, int Ny,
                          int margin, const double *dscale, double clamp_x, double clamp_y, unsigned *status, void *workspace,
                          void *stream) {
    PSX_REQUIRE(n >= 1 && n <= PSX_MAX_SRC, "psx_refract_batch_f32: %d refractions, 1..%d supported per call", n, PSX_MAX_SRC);
    PSX_REQUIRE(I_out != nullptr && dscale != nullptr && workspace != nullptr, "psx_refract_batch_f32: null outputs, scales or workspace");
    PSX_REQUIRE(nmat > 0, "psx_refract_batch_f32: the phase comes from the thickness maps (nmat > 0)");
    const bool has_I = I_in != nullptr && I_in[0] != nullptr;
    PSX_REQUIRE(has_I || I0 != nullptr, "psx_refract_batch_f32: neither input images nor uniform intensities");
    for (int e = 0; e < n; ++e) {
        PSX_REQUIRE(I_out[e] != nullptr, "psx_refract_batch_f32: null output image %d", e);
        PSX_REQUIRE(!has_I || I_in[e] != nullptr, "psx_refract_batch_f32: input images for all refractions or for none (%d)", e);
        for (int f = 0; f < e; ++f) PSX_REQUIRE(I_out[f] != I_out[e], "psx_refract_batch_f32: refractions %d and %d share an output image", f, e);
    }
    hipStream_t st = (hipStream_t)stream;
    PSX_REQUIRE(Nx >= 3 && Ny >= 3, "psx_refract_batch_f32: grid %dx%d too small for the edge_order=2 gradient", Nx, Ny);
    PSX_REQUIRE((int64_t)Nx * Ny < (1ll << 31), "psx_refract_batch_f32: grid %dx%d exceeds int32 pixel indices", Nx, Ny);
    PSX_REQUIRE(margin >= 8 && margin <= 4096, "psx_refract_batch_f32: margin %d must be >= 8 (the widest gather halo)", margin);
    const size_t chunk_ws = batch_chunk_max(Nx, Ny);
    for (int e0 = 0; e0 < n; e0 += REFRACT_TAB) {           // REFRACT_TAB argument blocks fit one launch
        const int m = std::min(REFRACT_TAB, n - e0);
        RefractTab t = {};
        for (int k = 0; k < REFRACT_TAB; ++k) {
            const int e = e0 + (k < m ? k : 0);
            RefractArgs &a = t.e[k];
            if (int rc = pack_mats(a.m, T, cphase ? cphase + (size_t)e * nmat : nullptr, catt ? catt + (size_t)e * nmat : nullptr, nmat))
                return rc;
            a.ndist = 1;
            a.I_in = has_I ? I_in[e] : nullptr; a.mask = nullptr; a.I0 = I0 ? I0[e] : 1.f;
            for (int d = 0; d < PSX_MAX_DIST; ++d) {
                a.I_out[d] = I_out[e];
                a.dscale[d] = dscale[e];
            }
            a.phi_in = nullptr; a.out_scale = out_scale; a.accumulate = accumulate;
            a.Dx_out = nullptr; a.Dy_out = nullptr; a.I_mut = nullptr; a.Nx = Nx; a.Ny = Ny; a.margin = margin;
            a.clamp_xf = (float)clamp_x; a.clamp_yf = (float)clamp_y; a.status = status; a.stamps = nullptr;
        }
        void *ws = (char *)workspace + (size_t)(e0 / REFRACT_TAB) * chunk_ws;     // every chunk its own far-ray lists
        const int rc = PSX_GEO_DISPATCH(G_, launch_refract_batch<G_>(t, m, has_I, nmat, ws, st));
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
