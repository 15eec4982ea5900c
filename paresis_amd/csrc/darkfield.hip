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

}  // namespace

extern "C" {

size_t psx_darkfield_workspace_bytes(int Nx, int Ny) { return sizeof(float2) * (size_t)(Nx > 0 ? Nx : 0) * (size_t)(Ny > 0 ? Ny : 0); }

int psx_darkfield_blur_f32(const float *I2DF, const float *DF, const float *I2, float *out, int Nx, int Ny, int R,
                           void *workspace, void *stream) {
    PSX_REQUIRE(I2DF && DF && out && workspace && Nx > 0 && Ny > 0 && R >= 0, "psx_darkfield_blur_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)Nx * Ny;
    PSX_TIMED("k_df_prepare", st, k_df_prepare<<<ew_grid(n, 256), 256, 0, st>>>(I2DF, DF, (float2 *)workspace, n));
    PSX_TIMED("k_df_gather", st, k_df_gather<<<ew_grid(n, 256), 256, 0, st>>>(I2DF, DF, (const float2 *)workspace, I2, out,
                                                                              Nx, Ny, R));
    return launch_check("k_df_gather");
}

}  // extern "C"
