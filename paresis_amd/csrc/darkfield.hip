// darkfield.hip -- the variable-width Gaussian re-splat of fastRefractionDF (refractionFileNumba2.py:168-186).
//
// After the dark-field part of the intensity has been refracted, the reference spreads every pixel (i,j) of it with a
// normalised Gaussian patch gaussian_shape(DF[i,j]/2) (RF2:14-23: side round(3*sigma)*2+1, banker's rounding) in an
// interpreted double loop.  The scatter becomes a gather: each output pixel sums the patches of the sources within
// R = max patch half-size that reach it.  The per-source normalisation is separable, (sum_d exp(-d^2/2sigma^2))^2, and is
// precomputed once per source together with the patch half-size.
#include <algorithm>

#include "common.hpp"

using namespace psx;

namespace {

// half-size of gaussian_shape(sigma): round-half-even(3*sigma)
__device__ __forceinline__ int patch_half(float sigma) { return (int)rintf(3.f * sigma); }

// per source: half-size (0 = plain deposit) and 1/normalisation
__global__ __launch_bounds__(256) void k_df_prepare(const float *__restrict__ I2DF, const float *__restrict__ DF,
                                                    float2 *__restrict__ prep, int64_t n) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const float df = DF[p], v = I2DF[p];
        float half = -1.f, inv = 0.f;                 // -1: contributes nothing (RF2:170)
        if (v != 0.f) {
            if (df != 0.f) {                          // RF2:171-178
                const float sigma = 0.5f * df;
                const int h = patch_half(sigma);
                double s = 0.0;
                for (int d = -h; d <= h; ++d) s += exp(-(double)(d * d) / 2.0 / ((double)sigma * sigma));
                half = (float)h;
                inv = (float)(1.0 / (s * s));
            } else {                                  // RF2:183-184
                half = 0.f;
                inv = 1.f;
            }
        }
        prep[p] = make_float2(half, inv);
    }
}

// out[t] = I2[t] + sum over sources s with |t-s|_inf <= half(s) of I2DF[s] * exp(-|t-s|^2 / 2 sigma_s^2) * inv(s)
__global__ __launch_bounds__(256) void k_df_gather(const float *__restrict__ I2DF, const float *__restrict__ DF,
                                                   const float2 *__restrict__ prep, const float *__restrict__ I2,
                                                   float *__restrict__ out, int Nx, int Ny, int R) {
    const int64_t n = (int64_t)Nx * Ny;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(p / Ny), j = (int)(p - (int64_t)i * Ny);
        float acc = 0.f;
        for (int di = -R; di <= R; ++di) {
            const int si = i + di;
            if (si < 0 || si >= Nx) continue;
            for (int dj = -R; dj <= R; ++dj) {
                const int sj = j + dj;
                if (sj < 0 || sj >= Ny) continue;
                const int64_t s = (int64_t)si * Ny + sj;
                const float2 pr = prep[s];
                const int h = (int)pr.x;
                if (h < 0 || abs(di) > h || abs(dj) > h) continue;
                if (h == 0) {
                    acc += I2DF[s];
                } else {
                    const float sigma = 0.5f * DF[s];
                    acc += I2DF[s] * pr.y * expf(-(float)(di * di + dj * dj) / (2.f * sigma * sigma));
                }
            }
        }
        out[p] = acc + (I2 ? I2[p] : 0.f);
    }
}

// The same gather from LDS: a workgroup owns a 32x32 tile of outputs and stages the sources of tile + halo R (weight
// I2DF * inv, exponent coefficient -1/(2 sigma^2), patch half-size) once; the neighbourhood loop then runs on LDS, and only
// as far as the widest patch actually present in the staged window (most tiles of an image lie outside the scattering
// sample: half-size 0, one term).  Same terms in the same order as k_df_gather (the exponent is formed as d^2 * (-log2(e)/2 sigma^2)
// and goes through v_exp_f32, 1 ulp, instead of expf(-d^2 / (2 sigma^2)): 0.50 -> 0.39 ms at 4096^2).
// One source row of the gather with a compile-time reach: the 2 RE + 1 entries are requested together (one LDS latency per row, not
// one per term), offsets, squared distances and the ring test `half-size >= max(|di|, |dj|)` are constants.  Same terms in the same
// order as the run-time loop it replaces (round 4: that loop waited out a full LDS round trip per term and spent eight scalar
// instructions on its bookkeeping: 0.317 -> 0.235 ms at 4096^2, gpurun_out/r4s23).
template <int RE, bool UNI>
__device__ __forceinline__ float df_row_terms(const float4 *row, int di, float acc) {
    const float di2 = (float)(di * di);
    const int adi = abs(di);
    // h == 0: coefficient 0, exp2(0) = 1, selected only at di = dj = 0 (same value as the plain deposit)
    // e.w (always 0) rides in the multiply-add so that the entry is fetched with ONE ds_read_b128 (4 LDS cycles per wave
    // instruction): left unused, the compiler narrows the load to ds_read_b96 (8 cycles: MI355X_MICROARCH.md)
    auto term = [&](const float4 &e, int dj) __attribute__((always_inline)) {
        const float need = (float)max(adi, abs(dj));
        const float ex = __builtin_amdgcn_exp2f((di2 + (float)(dj * dj)) * e.y);
        // UNI: every source of the window that carries weight has the widest patch -- no ring test, and only (weight, coefficient)
        // are read: one ds_read_b64 (with e.w in the expression the compiler fetches x, y and w as three 4-byte pieces)
        if constexpr (UNI) return e.x * ex;
        const float t = fmaf(e.x, ex, e.w);
        return e.z >= need ? t : 0.f;
    };
    if constexpr (RE <= 4) {
        float4 e[2 * RE + 1];
#pragma unroll
        for (int dj = -RE; dj <= RE; ++dj) e[dj + RE] = row[dj];
#pragma unroll
        for (int dj = -RE; dj <= RE; ++dj) acc += term(e[dj + RE], dj);
    } else {
        // wider rows in pieces of CH entries, a real loop over the pieces: unrolled, every piece's loads are hoisted to the front
        // (29 entries = 116 registers at RE = 14) whatever separates them
        constexpr int CH = 2 * RE + 1 <= 15 ? (2 * RE + 1 + 1) / 2 : ((2 * RE + 1) % 3 == 0 ? (2 * RE + 1) / 3 : (2 * RE + 1 + 2) / 3);
        constexpr int NP = (2 * RE + 1 + CH - 1) / CH;
#pragma unroll 1
        for (int pc = 0; pc < NP; ++pc) {
            const int c0 = -RE + pc * CH;
            float4 e[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) e[u] = row[min(c0 + u, RE)];         // the last piece may repeat the last entry ...
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float t = term(e[u], c0 + u);
                acc += c0 + u <= RE ? t : 0.f;                                  // ... which then adds nothing (uniform test)
            }
        }
    }
    return acc;
}

template <int RE, bool UNI>
__device__ __forceinline__ float df_terms(const float4 *centre, int W) {
    float acc = 0.f;
    if constexpr (RE <= 4) {
#pragma unroll
        for (int di = -RE; di <= RE; ++di) acc = df_row_terms<RE, UNI>(centre + di * W, di, acc);
    } else {
#pragma unroll 1
        for (int di = -RE; di <= RE; ++di) acc = df_row_terms<RE, UNI>(centre + di * W, di, acc);
    }
    return acc;
}

constexpr int DT = 32;
// Round 4: one 16-byte LDS entry per staged source -- (weight, exponent coefficient, patch half-size, unused) read with ONE
// ds_read_b128 -- and a branch-free inner loop: the term of a source is formed for every (di, dj) of the window's widest
// patch and selected by `half-size >= max(|di|, |dj|)` (a scalar per loop trip), where the first version tested three
// conditions per term with the exec-mask bookkeeping of a divergent `continue` and read two LDS arrays.  Inside a scattering
// sample nearly every source has the window's widest patch, so almost nothing that is computed is thrown away.  The status
// scan of the result (RF2:190-193) rides on the store.  0.39 -> 0.33 ms at 4096^2 (2.4-pixel dark field inside a cylinder).
// Two further forms were built on it, measured and taken out again (same tests green): four adjacent outputs per thread
// sharing each LDS entry (a third of the reads: 0.36 ms -- not LDS-bound), and the separable form -- per staged source the
// tables A_k = w E(k), E_k = E(k), E(k) = exp2(k^2 c) for k up to the window's widest patch, so that a term is two 4-byte LDS
// reads and one fma with no exponential, compare or select: 0.51 ms with the tables built for every window, 0.48 with
// tables only where a patch reaches, 0.40 with the reads of a source row issued together (compile-time reach): the planes
// take 48 KB per 32 x 16 tile (three workgroups per CU) and every term waits on LDS latency.
__global__ __launch_bounds__(256) void k_df_gather_tiled(const float *__restrict__ I2DF, const float *__restrict__ DF,
                                                         const float2 *__restrict__ prep, const float *__restrict__ I2,
                                                         float *__restrict__ out, int Nx, int Ny, int R, int tiles_y,
                                                         unsigned *status, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) char sdf[];
    const int W = DT + 2 * R;
    float4 *swc = reinterpret_cast<float4 *>(sdf);                 // [W][W] (weight, coefficient, half-size, -)
    __shared__ int hmax, hlow;
    const int t0 = (blockIdx.x / tiles_y) * DT, c0 = (blockIdx.x % tiles_y) * DT;
    if (threadIdx.x == 0) {
        hmax = 0;
        hlow = 1 << 20;
    }
    __syncthreads();
    int hm = 0, hl = 1 << 20;                            // widest patch of the window; narrowest among the sources that carry weight
    for (int e = threadIdx.x; e < W * W; e += 256) {
        const int a = e / W, b = e - a * W;
        const int si = t0 - R + a, sj = c0 - R + b;
        float w = 0.f, c = 0.f;
        int h = -1;
        if (si >= 0 && si < Nx && sj >= 0 && sj < Ny) {
            const int64_t q = (int64_t)si * Ny + sj;
            const float2 pr = prep[q];
            h = (int)pr.x;
            if (h >= 0) {
                w = I2DF[q] * pr.y;
                if (h > 0) {
                    const float sigma = 0.5f * DF[q];
                    c = -1.4426950408889634f / (2.f * sigma * sigma);        // x log2(e): the gather uses the hardware's 2^x
                }
            }
        }
        swc[e] = make_float4(w, c, (float)h, 0.f);
        hm = max(hm, h);
        if (w != 0.f) hl = min(hl, h);
    }
    for (int o = 32; o > 0; o >>= 1) {
        hm = max(hm, __shfl_xor(hm, o));
        hl = min(hl, __shfl_xor(hl, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&hmax, hm);
        atomicMin(&hlow, hl);
    }
    __syncthreads();
    const int Re = __builtin_amdgcn_readfirstlane(min(R, hmax));      // LDS words: uniform, but the compiler must be told
    // every source with weight has the window's widest patch (the inside of a sample whose width map varies slowly): the ring test
    // of a term is then always true and is compiled out -- a third of a term's vector instructions (a source without weight adds
    // 0 * 2^x = 0 either way; NaN / inf weights are != 0 and so take part in the test)
    const bool uniform = __builtin_amdgcn_readfirstlane(hlow) >= Re;
    const int tj = threadIdx.x & 31, ti0 = threadIdx.x >> 5;
    bool bad = false;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const int ti = ti0 + 8 * k, i = t0 + ti, j = c0 + tj;
        if (i >= Nx || j >= Ny) continue;
        float acc = 0.f;
        const float4 *centre = swc + (ti + R) * W + tj + R;
        switch (Re) {                                    // uniform over the workgroup
            case 0: {                                    // no patch reaches this tile: every source deposits on itself
                const float4 e = *centre;
                acc = e.z >= 0.f ? e.x : 0.f;
                break;
            }
#define PSX_DF_CASE(n) case n: acc = uniform ? df_terms<n, true>(centre, W) : df_terms<n, false>(centre, W); break;
            PSX_DF_CASE(1) PSX_DF_CASE(2) PSX_DF_CASE(3) PSX_DF_CASE(4) PSX_DF_CASE(5) PSX_DF_CASE(6) PSX_DF_CASE(7)
            PSX_DF_CASE(8) PSX_DF_CASE(9) PSX_DF_CASE(10) PSX_DF_CASE(11) PSX_DF_CASE(12) PSX_DF_CASE(13) PSX_DF_CASE(14)
#undef PSX_DF_CASE
            default:                                     // (the launch admits R <= 14: 60 KiB of LDS)
                for (int di = -Re; di <= Re; ++di) {
                    const float4 *row = centre + di * W;
                    const float di2 = (float)(di * di);
                    for (int dj = -Re; dj <= Re; ++dj) {
                        const float4 e = row[dj];
                        const float need = (float)max(abs(di), abs(dj));
                        const float t = fmaf(e.x, __builtin_amdgcn_exp2f((di2 + (float)(dj * dj)) * e.y), e.w);
                        acc += e.z >= need ? t : 0.f;
                    }
                }
        }
        const int64_t p = (int64_t)i * Ny + j;
        const float v = acc + (I2 ? I2[p] : 0.f);
        bad |= !(fabsf(v) <= 3.0e38f);
        out[p] = accumulate ? out[p] + v : v;            // accumulate: the chain's sum over energies (EXP:478-483) rides on the store
    }
    if (status && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(status, PSX_STATUS_NONFINITE);
}

// Patches wider than the tiled kernel's LDS window (R > 14): the same gather with the sources staged in BANDS of rows.  A
// workgroup owns a 32 x 32 tile of outputs and walks the source rows [t0 - R, t0 + 32 + R) in bands of `band` rows x (32 + 2 R)
// columns of 16-byte entries (whatever fits the LDS budget); each thread keeps its four outputs in registers across the bands,
// so the terms of an output are added in the order of the plain gather (rows, then columns).  A band knows the widest patch it
// holds: rows and columns beyond it are skipped, so a tile outside the scattering sample costs its staging only.  Replaces the
// global-memory gather k_df_gather as the path of wide dark fields (round 5: a 13-pixel dark field, R = 21, took 27.7 ms per
// 4096^2 image there -- every term a dependent global load; its cost is now that of the terms, ~2 R^2 exponentials per pixel).
__global__ __launch_bounds__(256) void k_df_gather_banded(const float *__restrict__ I2DF, const float *__restrict__ DF,
                                                          const float2 *__restrict__ prep, const float *__restrict__ I2,
                                                          float *__restrict__ out, int Nx, int Ny, int R, int band, int tiles_y,
                                                          unsigned *status, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) char sdf[];
    const int W = DT + 2 * R;
    float4 *swc = reinterpret_cast<float4 *>(sdf);                 // [band][W] (weight, coefficient, half-size, 0)
    __shared__ int hband;
    const int t0 = (blockIdx.x / tiles_y) * DT, c0 = (blockIdx.x % tiles_y) * DT;
    const int tj = threadIdx.x & 31, ti0 = threadIdx.x >> 5;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = t0 - R; b0 < t0 + DT + R; b0 += band) {
        const int rows = min(band, t0 + DT + R - b0);
        __syncthreads();                                 // the previous band has been consumed
        if (threadIdx.x == 0) hband = -1;
        __syncthreads();
        int hm = -1;
        for (int e = threadIdx.x; e < rows * W; e += 256) {
            const int a = e / W, b = e - a * W;
            const int si = b0 + a, sj = c0 - R + b;
            float w = 0.f, c = 0.f;
            int h = -1;
            if (si >= 0 && si < Nx && sj >= 0 && sj < Ny) {
                const int64_t q = (int64_t)si * Ny + sj;
                const float2 pr = prep[q];
                h = (int)pr.x;
                if (h >= 0) {
                    w = I2DF[q] * pr.y;
                    if (h > 0) {
                        const float sigma = 0.5f * DF[q];
                        c = -1.4426950408889634f / (2.f * sigma * sigma);
                    }
                    if (w == 0.f) h = -1;                // carries nothing: must not widen the band's reach
                }
            }
            swc[e] = make_float4(w, c, (float)h, 0.f);
            hm = max(hm, h);
        }
        for (int o = 32; o > 0; o >>= 1) hm = max(hm, __shfl_xor(hm, o));
        if ((threadIdx.x & 63) == 0 && hm >= 0) atomicMax(&hband, hm);
        __syncthreads();
        const int hb = __builtin_amdgcn_readfirstlane(min(R, hband));
        if (hb < 0) continue;                            // nothing in this band carries weight (uniform)
        // A thread's four outputs sit in ONE column, eight rows apart: a staged source entry serves all four (its row distance
        // differs, its column distance does not), so the band is walked row by row and an entry is read once for the four --
        // a quarter of the LDS reads of the output-by-output walk (5.55 -> 5.30 ms at R = 21, 745 -> 676 ms for a 25-energy spectrum down to
        // R = 136, gpurun_out/r5s27: the exponentials, not the reads, are what a term costs).  Per output the terms
        // still arrive in the plain gather's order (rows, then columns).
        const int i0 = t0 + ti0;                         // rows i0, i0 + 8, i0 + 16, i0 + 24
        const int r_lo = max(b0, i0 - hb), r_hi = min(b0 + rows - 1, i0 + 24 + hb);
        for (int si = r_lo; si <= r_hi; ++si) {
            const float4 *row = swc + (si - b0) * W + tj + R;
            int adi[4];
            float di2[4];
            bool rowin = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int di = si - (i0 + 8 * k);
                adi[k] = abs(di);
                di2[k] = (float)(di * di);
                rowin = rowin || adi[k] <= hb;
            }
            if (!rowin) continue;                        // a gap between two outputs' reaches (hb < 4)
            int dj = -hb;
            for (; dj + 3 <= hb; dj += 4) {              // four entries requested together
                float4 e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) e[u] = row[dj + u];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (adi[k] > hb) continue;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float need = (float)max(adi[k], abs(dj + u));
                        const float t = fmaf(e[u].x, __builtin_amdgcn_exp2f((di2[k] + (float)((dj + u) * (dj + u))) * e[u].y), e[u].w);
                        acc[k] += e[u].z >= need ? t : 0.f;
                    }
                }
            }
            for (; dj <= hb; ++dj) {
                const float4 e = row[dj];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float need = (float)max(adi[k], abs(dj));
                    const float t = fmaf(e.x, __builtin_amdgcn_exp2f((di2[k] + (float)(dj * dj)) * e.y), e.w);
                    acc[k] += (adi[k] <= hb && e.z >= need) ? t : 0.f;
                }
            }
        }
    }
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = t0 + ti0 + 8 * k, j = c0 + tj;
        if (i < Nx && j < Ny) {
            const int64_t p = (int64_t)i * Ny + j;
            const float v = acc[k] + (I2 ? I2[p] : 0.f);
            bad |= !(fabsf(v) <= 3.0e38f);
            out[p] = accumulate ? out[p] + v : v;
        }
    }
    if (status && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(status, PSX_STATUS_NONFINITE);
}

// ---- the front of fastRefractionDF as ONE pass (RF2:114-150): width map in radians -> pixels (float64), its maximum (the
// margin of the returned displacement maps is ceil(6 max), RF2:117), the DF > Nx/4 -> 0 rule (RF2:135), the split of the
// intensity by DF != 0 (RF2:147-150), and -- the width map being all the patch normalisation depends on -- the per-source
// patch half-size and 1/normalisation that k_df_prepare would compute after the refraction.
// words[0] / words[1]: bit patterns of the largest width before / after the rule (non-negative doubles order like integers).
__global__ __launch_bounds__(256) void k_df_split(const float *__restrict__ I, const double *__restrict__ DFrad, double num,
                                                  double den, double limit, float *__restrict__ I_nodf, float *__restrict__ I_df,
                                                  float *__restrict__ DFpx, float2 *__restrict__ prep,
                                                  unsigned long long *__restrict__ words, int64_t n) {
    double m0 = 0.0, m1 = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        // RF2:114 in the reference's own order of operations, (DF * z) / (h M), in float64: the margin ceil(6 max), the
        // DF > Nx/4 rule and the patch sides round(3 DF/2) are step functions of this number, and a product with a
        // pre-divided z / (h M) differs from it in the last bit often enough to move a step (found by tests/test_gpu_fuzz.py)
        const double q = DFrad[p] * num;
        double d = 0.0;
        if (q != 0.0) d = q / den;            // (most of a width map is zero: outside the scattering sample)
        m0 = fmax(m0, d);
        if (d > limit) d = 0.0;
        m1 = fmax(m1, d);
        const float df = (float)d, v = I[p];
        DFpx[p] = df;
        I_nodf[p] = d != 0.0 ? 0.f : v;
        I_df[p] = d != 0.0 ? v : 0.f;
        float half = 0.f, inv = 1.f;                  // DF == 0: plain deposit (RF2:183-184)
        if (d != 0.0) {                               // RF2:171-178
            const double sigma = d / 2;               // RF2:174; the side of its patch from the float64 value (RF2:15)
            const int h = (int)rint(sigma * 3);
            // sum_{|k| <= h} exp(-k^2 a) with ONE exponential: the terms obey t_k = t_{k-1} q_k, q_k = q_{k-1} e^{-2a}, q_1 = e^{-a}
            // (float64 throughout: 1e-15 per step; the direct sum of 2h+1 float64 exponentials was 0.6 ms of this pass at 4096^2)
            const double e1 = exp(-1.0 / 2.0 / (sigma * sigma)), r = e1 * e1;
            double t = 1.0, q = e1, sum = 1.0;
            for (int k = 1; k <= h; ++k) {
                t *= q;
                q *= r;
                sum += 2.0 * t;
            }
            half = (float)h;
            inv = (float)(1.0 / (sum * sum));
        }
        prep[p] = make_float2(half, inv);
    }
    // one atomic pair per WORKGROUP of a grid of at most 1024 (atomics on one word retire at ~90 per microsecond: one pair
    // per wave of a 16384-block grid was 0.6 ms of a pass that streams in 0.1)
    __shared__ double sm[2][4];
    for (int o = 32; o > 0; o >>= 1) {
        m0 = fmax(m0, __shfl_xor(m0, o));
        m1 = fmax(m1, __shfl_xor(m1, o));
    }
    if ((threadIdx.x & 63) == 0) {
        sm[0][threadIdx.x >> 6] = m0;
        sm[1][threadIdx.x >> 6] = m1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        m0 = fmax(fmax(sm[0][0], sm[0][1]), fmax(sm[0][2], sm[0][3]));
        m1 = fmax(fmax(sm[1][0], sm[1][1]), fmax(sm[1][2], sm[1][3]));
        if (m0 > 0.0) atomicMax(&words[0], (unsigned long long)__double_as_longlong(m0));
        if (m1 > 0.0) atomicMax(&words[1], (unsigned long long)__double_as_longlong(m1));
    }
}

// I = a + b (the two halves of the split have disjoint supports; the refractions zeroed the clamped rays in them, RF2:128-129)
__global__ __launch_bounds__(256) void k_df_merge(float *__restrict__ I, const float *__restrict__ a, const float *__restrict__ b,
                                                  int64_t n) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) I[p] = a[p] + b[p];
}

// dst[Nx + 2 md][Ny + 2 md] = centre [Nx][Ny] of src[Nx + 2 ms][Ny + 2 ms], zero frame
__global__ __launch_bounds__(256) void k_repad(const float *__restrict__ src, int ms, float *__restrict__ dst, int md, int Nx,
                                               int Ny) {
    const int Wd = Ny + 2 * md, Ws = Ny + 2 * ms;
    const int64_t n = (int64_t)(Nx + 2 * md) * Wd;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(p / Wd) - md, j = (int)(p % Wd) - md;
        dst[p] = (i >= 0 && i < Nx && j >= 0 && j < Ny) ? src[(int64_t)(i + ms) * Ws + j + ms] : 0.f;
    }
}

// wide patches: bands of source rows through LDS (64 KiB per workgroup: two workgroups per CU)
int launch_banded(const float *I2DF, const float *DF, const float2 *prep, const float *I2, float *out, int Nx, int Ny, int R,
                  unsigned *status, int accumulate, hipStream_t st) {
    const int W = DT + 2 * R;
    const int band = (int)std::min<size_t>((size_t)(DT + 2 * R), (64 * 1024) / (sizeof(float4) * (size_t)W));
    if (band < 1) {      // a window row does not fit the budget (R > 2032): the plain gather from global memory
        const int64_t n = (int64_t)Nx * Ny;
        PSX_REQUIRE(!accumulate, "dark-field re-splat: patches of %d pixels cannot be accumulated in place", 2 * R + 1);
        PSX_TIMED("k_df_gather", st, k_df_gather<<<ew_grid(n, 256), 256, 0, st>>>(I2DF, DF, prep, I2, out, Nx, Ny, R));
        if (int rc = launch_check("k_df_gather")) return rc;
        return status ? psx_status_scan_f32(out, n, status, (void *)st) : 0;
    }
    const size_t lds = sizeof(float4) * (size_t)band * (size_t)W;
    static std::atomic<unsigned long long> attr_mask{0};
    if (first_on_device(attr_mask))
        PSX_HIP(hipFuncSetAttribute((const void *)k_df_gather_banded, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    const int tiles_x = (int)cdiv(Nx, DT), tiles_y = (int)cdiv(Ny, DT);
    PSX_TIMED("k_df_gather", st, k_df_gather_banded<<<tiles_x * tiles_y, 256, lds, st>>>(I2DF, DF, prep, I2, out, Nx, Ny, R, band,
                                                                                         tiles_y, status, accumulate));
    return launch_check("k_df_gather");
}

}  // namespace

extern "C" {

int psx_darkfield_split_f32(const float *I, const double *DF_rad, double num, double den, double limit, float *I_nodf,
                            float *I_df, float *DF_px, void *prep, unsigned long long *words, int Nx, int Ny, void *stream) {
    PSX_REQUIRE(I && DF_rad && I_nodf && I_df && DF_px && prep && words && Nx > 0 && Ny > 0 && den != 0.0,
                "psx_darkfield_split_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)Nx * Ny;
    PSX_HIP(hipMemsetAsync(words, 0, 2 * sizeof(unsigned long long), st));
    const int grid = (int)std::min<int64_t>(1024, cdiv(n, 256));
    PSX_TIMED("k_df_split", st, k_df_split<<<grid, 256, 0, st>>>(I, DF_rad, num, den, limit, I_nodf, I_df, DF_px,
                                                                            (float2 *)prep, words, n));
    return launch_check("k_df_split");
}

int psx_darkfield_merge_f32(float *I, const float *a, const float *b, int64_t n, void *stream) {
    PSX_REQUIRE(I && a && b && n > 0, "psx_darkfield_merge_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    PSX_TIMED("k_df_merge", st, k_df_merge<<<ew_grid(n, 256), 256, 0, st>>>(I, a, b, n));
    return launch_check("k_df_merge");
}

int psx_repad_f32(const float *src, int margin_src, float *dst, int margin_dst, int Nx, int Ny, void *stream) {
    PSX_REQUIRE(src && dst && Nx > 0 && Ny > 0 && margin_src >= 0 && margin_dst >= 0, "psx_repad_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)(Nx + 2 * margin_dst) * (Ny + 2 * margin_dst);
    PSX_TIMED("k_repad", st, k_repad<<<ew_grid(n, 256), 256, 0, st>>>(src, margin_src, dst, margin_dst, Nx, Ny));
    return launch_check("k_repad");
}

int psx_darkfield_blur_prepared_f32(const float *I2DF, const float *DF, const void *prep, const float *I2, float *out, int Nx,
                                    int Ny, int R, unsigned *status, int accumulate, void *stream) {
    PSX_REQUIRE(I2DF && DF && out && prep && Nx > 0 && Ny > 0 && R >= 0, "psx_darkfield_blur_prepared_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)Nx * Ny;
    const size_t lds = (size_t)(DT + 2 * R) * (DT + 2 * R) * sizeof(float4);
    if (lds <= 60 * 1024) {
        const int tiles_x = (int)cdiv(Nx, DT), tiles_y = (int)cdiv(Ny, DT);
        PSX_TIMED("k_df_gather", st, k_df_gather_tiled<<<tiles_x * tiles_y, 256, lds, st>>>(I2DF, DF, (const float2 *)prep, I2, out,
                                                                                            Nx, Ny, R, tiles_y, status, accumulate));
        return launch_check("k_df_gather");
    }
    (void)n;
    return launch_banded(I2DF, DF, (const float2 *)prep, I2, out, Nx, Ny, R, status, accumulate, st);
}

size_t psx_darkfield_workspace_bytes(int Nx, int Ny) { return sizeof(float2) * (size_t)(Nx > 0 ? Nx : 0) * (size_t)(Ny > 0 ? Ny : 0); }

int psx_darkfield_blur_f32(const float *I2DF, const float *DF, const float *I2, float *out, int Nx, int Ny, int R,
                           void *workspace, void *stream) {
    PSX_REQUIRE(I2DF && DF && out && workspace && Nx > 0 && Ny > 0 && R >= 0, "psx_darkfield_blur_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)Nx * Ny;
    PSX_TIMED("k_df_prepare", st, k_df_prepare<<<ew_grid(n, 256), 256, 0, st>>>(I2DF, DF, (float2 *)workspace, n));
    const size_t lds = (size_t)(DT + 2 * R) * (DT + 2 * R) * sizeof(float4);
    if (lds <= 60 * 1024) {          // patches of up to 2 R + 1 = 29 pixels; wider ones take the plain gather
        const int tiles_x = (int)cdiv(Nx, DT), tiles_y = (int)cdiv(Ny, DT);
        PSX_TIMED("k_df_gather", st, k_df_gather_tiled<<<tiles_x * tiles_y, 256, lds, st>>>(I2DF, DF, (const float2 *)workspace, I2,
                                                                                            out, Nx, Ny, R, tiles_y, nullptr, 0));
        return launch_check("k_df_gather");
    }
    return launch_banded(I2DF, DF, (const float2 *)workspace, I2, out, Nx, Ny, R, nullptr, 0, st);
}

}  // extern "C"
