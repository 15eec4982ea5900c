// darkfield.hip -- the variable-width Gaussian re-splat of fastRefractionDF (refractionFileNumba2.py:168-186).
//
// After the dark-field part of the intensity has been refracted, the reference spreads every pixel (i,j) of it with a
// normalised Gaussian patch gaussian_shape(DF[i,j]/2) (RF2:14-23: side round(3*sigma)*2+1, banker's rounding) in an
// interpreted double loop.  The scatter becomes a gather: each output pixel sums the patches of the sources within
// R = max patch half-size that reach it.  The per-source normalisation is separable, (sum_d exp(-d^2/2sigma^2))^2, and is
// precomputed once per source together with the patch half-size.
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
// sample: half-size 0, one term).  Same terms in the same order as k_df_gather (the exponent is formed as d^2 * (-1/2 sigma^2)
// instead of -d^2 / (2 sigma^2): one rounding apart).
constexpr int DT = 32;
__global__ __launch_bounds__(256) void k_df_gather_tiled(const float *__restrict__ I2DF, const float *__restrict__ DF,
                                                         const float2 *__restrict__ prep, const float *__restrict__ I2,
                                                         float *__restrict__ out, int Nx, int Ny, int R, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) char sdf[];
    const int W = DT + 2 * R;
    float2 *swc = reinterpret_cast<float2 *>(sdf);                 // [W][W] (weight, coefficient)
    int *sh = reinterpret_cast<int *>(swc + W * W);                // [W][W] half-size (-1: no contribution)
    __shared__ int hmax;
    const int t0 = (blockIdx.x / tiles_y) * DT, c0 = (blockIdx.x % tiles_y) * DT;
    if (threadIdx.x == 0) hmax = 0;
    __syncthreads();
    int hm = 0;
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
                    c = -1.f / (2.f * sigma * sigma);
                }
            }
        }
        swc[e] = make_float2(w, c);
        sh[e] = h;
        hm = max(hm, h);
    }
    for (int o = 32; o > 0; o >>= 1) hm = max(hm, __shfl_xor(hm, o));
    if ((threadIdx.x & 63) == 0) atomicMax(&hmax, hm);
    __syncthreads();
    const int Re = min(R, hmax);
    const int tj = threadIdx.x & 31, ti0 = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ti = ti0 + 8 * k, i = t0 + ti, j = c0 + tj;
        if (i >= Nx || j >= Ny) continue;
        float acc = 0.f;
        for (int di = -Re; di <= Re; ++di) {
            const int row = (ti + di + R) * W + tj + R;
            for (int dj = -Re; dj <= Re; ++dj) {
                const int h = sh[row + dj];
                if (h < 0 || abs(di) > h || abs(dj) > h) continue;
                const float2 wc = swc[row + dj];
                acc += h == 0 ? wc.x : wc.x * expf((float)(di * di + dj * dj) * wc.y);
            }
        }
        const int64_t p = (int64_t)i * Ny + j;
        out[p] = acc + (I2 ? I2[p] : 0.f);
    }
}

}  // namespace

extern "C" {

size_t psx_darkfield_workspace_bytes(int Nx, int Ny) { return sizeof(float2) * (size_t)(Nx > 0 ? Nx : 0) * (size_t)(Ny > 0 ? Ny : 0); }

int psx_darkfield_blur_f32(const float *I2DF, const float *DF, const float *I2, float *out, int Nx, int Ny, int R,
                           void *workspace, void *stream) {
    PSX_REQUIRE(I2DF && DF && out && workspace && Nx > 0 && Ny > 0 && R >= 0, "psx_darkfield_blur_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)Nx * Ny;
    PSX_TIMED("k_df_prepare", st, k_df_prepare<<<ew_grid(n, 256), 256, 0, st>>>(I2DF, DF, (float2 *)workspace, n));
    const size_t lds = (size_t)(DT + 2 * R) * (DT + 2 * R) * (sizeof(float2) + sizeof(int));
    if (lds <= 60 * 1024) {          // patches of up to 2 R + 1 = 51 pixels; wider ones take the plain gather
        const int tiles_x = (int)cdiv(Nx, DT), tiles_y = (int)cdiv(Ny, DT);
        PSX_TIMED("k_df_gather", st, k_df_gather_tiled<<<tiles_x * tiles_y, 256, lds, st>>>(I2DF, DF, (const float2 *)workspace, I2,
                                                                                            out, Nx, Ny, R, tiles_y));
        return launch_check("k_df_gather");
    }
    PSX_TIMED("k_df_gather", st, k_df_gather<<<ew_grid(n, 256), 256, 0, st>>>(I2DF, DF, (const float2 *)workspace, I2, out,
                                                                              Nx, Ny, R));
    return launch_check("k_df_gather");
}

}  // extern "C"
