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
// Round 4: the patch of a source is SEPARABLE -- exp(-(di^2 + dj^2) / 2 sigma_s^2) = E_s(|di|) E_s(|dj|), and so is its square
// support (max(|di|, |dj|) <= h_s  <=>  |di| <= h_s and |dj| <= h_s).  The staging pass therefore tabulates, per staged source,
//     A_k = w_s E_s(k) [k <= h_s],  k = 0..R      and      E_k = E_s(k) [k <= h_s],  k = 1..R
// (R + 1 exponentials per source, planes [2R + 1][window] of floats in LDS), and a term of the gather is
//     A_|di|[s] * E_|dj|[s]          -- two 4-byte LDS reads and one fma: no exponential, no compare, no select.
// The first tiled version evaluated w exp2((di^2 + dj^2) c) per term behind three tests (0.39 ms at 4096^2, 81 terms per
// output inside the sample); a branch-free form with one 16-byte LDS entry per term took 0.33 ms and was bound by the
// quarter-rate v_exp_f32; sharing an entry between four adjacent outputs changed nothing (0.36 ms: not LDS-bound).
// A workgroup owns DTX x DTY = 32 x 16 outputs, two per thread (rows ti and ti + 16 of one column).
// (16 x 16 for the widest patches: the planes of R = 12 then still fit the 160 KB of a CU)
constexpr int DTY = 16;
inline size_t df_lds_bytes(int R, int DTX) { return sizeof(float) * (size_t)(2 * R + 1) * (DTX + 2 * R) * (DTY + 2 * R); }

template <int DTX>
__global__ __launch_bounds__(256) void k_df_gather_tiled(const float *__restrict__ I2DF, const float *__restrict__ DF,
                                                         const float2 *__restrict__ prep, const float *__restrict__ I2,
                                                         float *__restrict__ out, int Nx, int Ny, int R, int tiles_y,
                                                         unsigned *status) {
    extern __shared__ __attribute__((aligned(16))) char sdf[];
    const int WX = DTX + 2 * R, WY = DTY + 2 * R, WN = WX * WY;
    float *pl = reinterpret_cast<float *>(sdf);          // planes A_0 .. A_R, then E_1 .. E_R, each [WX][WY]
    __shared__ int hmax;
    const int t0 = (blockIdx.x / tiles_y) * DTX, c0 = (blockIdx.x % tiles_y) * DTY;
    if (threadIdx.x == 0) hmax = 0;
    __syncthreads();
    int hm = 0;
    for (int e = threadIdx.x; e < WN; e += 256) {
        const int a = e / WY, b = e - a * WY;
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
                    c = -1.4426950408889634f / (2.f * sigma * sigma);        // x log2(e): the hardware's 2^x
                }
            }
        }
        pl[e] = h >= 0 ? w : 0.f;                                          // A_0
        for (int k = 1; k <= R; ++k) {
            const float E = k <= h ? __builtin_amdgcn_exp2f((float)(k * k) * c) : 0.f;
            pl[k * WN + e] = w * E;                                          // A_k
            pl[(R + k) * WN + e] = E;                                        // E_k
        }
        hm = max(hm, h);
    }
    for (int o = 32; o > 0; o >>= 1) hm = max(hm, __shfl_xor(hm, o));
    if ((threadIdx.x & 63) == 0) atomicMax(&hmax, hm);
    __syncthreads();
    const int Re = min(R, hmax);
    const int tj = threadIdx.x & (DTY - 1), ti0 = threadIdx.x / DTY;      // 16 columns x 16 rows of threads
    bool bad = false;
#pragma unroll
    for (int k = 0; k < DTX / 16; ++k) {
        const int ti = ti0 + 16 * k, i = t0 + ti, j = c0 + tj;
        if (i >= Nx || j >= Ny) continue;
        float acc = 0.f;
        for (int di = -Re; di <= Re; ++di) {
            const float *A = pl + abs(di) * WN + (ti + di + R) * WY + tj + R;   // plane A_|di|, this source row, own column
            const float *E = pl + R * WN + (ti + di + R) * WY + tj + R;           // planes E_1.. start at (R + 1) * WN
            float r = A[0];                                                       // dj = 0
            for (int dj = 1; dj <= Re; ++dj) {
                const float *Ed = E + dj * WN;
                r = fmaf(A[-dj], Ed[-dj], r);
                r = fmaf(A[dj], Ed[dj], r);
            }
            acc += r;
        }
        const int64_t p = (int64_t)i * Ny + j;
        const float v = acc + (I2 ? I2[p] : 0.f);
        bad |= !(fabsf(v) <= 3.0e38f);
        out[p] = v;
    }
    if (status && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(status, PSX_STATUS_NONFINITE);
}

// tiled gather when the planes fit LDS (patches of up to 2 R + 1 = 25 pixels), else the plain one; returns <0 when not launched
static int launch_df_gather_tiled(const float *I2DF, const float *DF, const float2 *prep, const float *I2, float *out, int Nx, int Ny,
                                  int R, unsigned *status, hipStream_t st) {
    constexpr size_t LDS_MAX = 160 * 1024 - 64;
    const int tiles_y = (int)cdiv(Ny, DTY);
    if (df_lds_bytes(R, 32) <= LDS_MAX) {
        static std::atomic<unsigned long long> m32{0};
        if (first_on_device(m32))
            PSX_HIP(hipFuncSetAttribute((const void *)k_df_gather_tiled<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
        PSX_TIMED("k_df_gather", st, k_df_gather_tiled<32><<<(int)cdiv(Nx, 32) * tiles_y, 256, df_lds_bytes(R, 32), st>>>(
                                         I2DF, DF, prep, I2, out, Nx, Ny, R, tiles_y, status));
        return launch_check("k_df_gather");
    }
    if (df_lds_bytes(R, 16) <= LDS_MAX) {
        static std::atomic<unsigned long long> m16{0};
        if (first_on_device(m16))
            PSX_HIP(hipFuncSetAttribute((const void *)k_df_gather_tiled<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
        PSX_TIMED("k_df_gather", st, k_df_gather_tiled<16><<<(int)cdiv(Nx, 16) * tiles_y, 256, df_lds_bytes(R, 16), st>>>(
                                         I2DF, DF, prep, I2, out, Nx, Ny, R, tiles_y, status));
        return launch_check("k_df_gather");
    }
    return -1000;
}

// ---- the front of fastRefractionDF as ONE pass (RF2:114-150): width map in radians -> pixels (float64), its maximum (the
// margin of the returned displacement maps is ceil(6 max), RF2:117), the DF > Nx/4 -> 0 rule (RF2:135), the split of the
// intensity by DF != 0 (RF2:147-150), and -- the width map being all the patch normalisation depends on -- the per-source
// patch half-size and 1/normalisation that k_df_prepare would compute after the refraction.
// words[0] / words[1]: bit patterns of the largest width before / after the rule (non-negative doubles order like integers).
__global__ __launch_bounds__(256) void k_df_split(const float *__restrict__ I, const double *__restrict__ DFrad, double scale,
                                                  double limit, float *__restrict__ I_nodf, float *__restrict__ I_df,
                                                  float *__restrict__ DFpx, float2 *__restrict__ prep,
                                                  unsigned long long *__restrict__ words, int64_t n) {
    double m0 = 0.0, m1 = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        double d = DFrad[p] * scale;
        m0 = fmax(m0, d);
        if (d > limit) d = 0.0;
        m1 = fmax(m1, d);
        const float df = (float)d, v = I[p];
        DFpx[p] = df;
        I_nodf[p] = df != 0.f ? 0.f : v;
        I_df[p] = df != 0.f ? v : 0.f;
        float half = 0.f, inv = 1.f;                  // DF == 0: plain deposit (RF2:183-184)
        if (df != 0.f) {                              // RF2:171-178
            const float sigma = 0.5f * df;
            const int h = patch_half(sigma);
            // sum_{|k| <= h} exp(-k^2 a) with ONE exponential: the terms obey t_k = t_{k-1} q_k, q_k = q_{k-1} e^{-2a}, q_1 = e^{-a}
            // (float64 throughout: 1e-15 per step; the direct sum of 2h+1 float64 exponentials was 0.6 ms of this pass at 4096^2)
            const double e1 = exp(-1.0 / 2.0 / ((double)sigma * sigma)), r = e1 * e1;
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

}  // namespace

extern "C" {

int psx_darkfield_split_f32(const float *I, const double *DF_rad, double scale, double limit, float *I_nodf, float *I_df,
                            float *DF_px, void *prep, unsigned long long *words, int Nx, int Ny, void *stream) {
    PSX_REQUIRE(I && DF_rad && I_nodf && I_df && DF_px && prep && words && Nx > 0 && Ny > 0, "psx_darkfield_split_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)Nx * Ny;
    PSX_HIP(hipMemsetAsync(words, 0, 2 * sizeof(unsigned long long), st));
    const int grid = (int)std::min<int64_t>(1024, cdiv(n, 256));
    PSX_TIMED("k_df_split", st, k_df_split<<<grid, 256, 0, st>>>(I, DF_rad, scale, limit, I_nodf, I_df, DF_px,
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
                                    int Ny, int R, unsigned *status, void *stream) {
    PSX_REQUIRE(I2DF && DF && out && prep && Nx > 0 && Ny > 0 && R >= 0, "psx_darkfield_blur_prepared_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)Nx * Ny;
    if (int rc = launch_df_gather_tiled(I2DF, DF, (const float2 *)prep, I2, out, Nx, Ny, R, status, st); rc != -1000) return rc;
    PSX_TIMED("k_df_gather", st, k_df_gather<<<ew_grid(n, 256), 256, 0, st>>>(I2DF, DF, (const float2 *)prep, I2, out, Nx, Ny, R));
    if (int rc = launch_check("k_df_gather")) return rc;
    return status ? psx_status_scan_f32(out, n, status, stream) : 0;
}

size_t psx_darkfield_workspace_bytes(int Nx, int Ny) { return sizeof(float2) * (size_t)(Nx > 0 ? Nx : 0) * (size_t)(Ny > 0 ? Ny : 0); }

int psx_darkfield_blur_f32(const float *I2DF, const float *DF, const float *I2, float *out, int Nx, int Ny, int R,
                           void *workspace, void *stream) {
    PSX_REQUIRE(I2DF && DF && out && workspace && Nx > 0 && Ny > 0 && R >= 0, "psx_darkfield_blur_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)Nx * Ny;
    PSX_TIMED("k_df_prepare", st, k_df_prepare<<<ew_grid(n, 256), 256, 0, st>>>(I2DF, DF, (float2 *)workspace, n));
    if (int rc = launch_df_gather_tiled(I2DF, DF, (const float2 *)workspace, I2, out, Nx, Ny, R, nullptr, st); rc != -1000) return rc;
    PSX_TIMED("k_df_gather", st, k_df_gather<<<ew_grid(n, 256), 256, 0, st>>>(I2DF, DF, (const float2 *)workspace, I2, out,
                                                                              Nx, Ny, R));
    return launch_check("k_df_gather");
}

}  // extern "C"
