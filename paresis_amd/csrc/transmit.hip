// transmit.hip -- object transmission (K1, K2) and attenuated accumulation (K8 tail).
// Elementwise, HBM-bound: one coalesced pass, 16 B per lane where the layout allows.
#include "common.hpp"

using namespace psx;

// K1: Sample.py:279  wave <- exp((-i k delta - k beta) T) wave, all materials in one pass.
template <int NM>
__global__ __launch_bounds__(256) void k_transmit_wave(const float2 *__restrict__ win, float amp, Mats m,
                                                       float2 *__restrict__ wout, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        double ph, la;
        mats_eval<NM>(m, p, ph, la);
        float c, s;
        cis_f64(ph, c, s);
        const float a = amp * exp_att(la);
        float2 w = win ? win[p] : make_float2(1.f, 0.f);
        float2 o;
        o.x = a * (w.x * c - w.y * s);
        o.y = a * (w.x * s + w.y * c);
        wout[p] = o;
    }
}

// K2: Sample.py:347-348  I <- exp(-2 k beta T) I ; phi <- phi - k delta T
template <int NM>
__global__ __launch_bounds__(256) void k_transmit_rt(const float *__restrict__ Iin, float I0, Mats m,
                                                     float *__restrict__ Iout, const double *__restrict__ phin,
                                                     double *__restrict__ phout, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        double ph, la;
        mats_eval<NM>(m, p, ph, la);
        if (Iout) Iout[p] = I0 * (Iin ? Iin[p] : 1.f) * expf((float)la);
        if (phout) phout[p] = (phin ? phin[p] : 0.0) + ph;
    }
}

// EXP:351-358 / 478-483: acc (+)= scale * img * exp(sum catt T)
// acc and img may alias (in-place attenuation), so no __restrict__ here
template <int NM>
__global__ __launch_bounds__(256) void k_accumulate(float *acc, const float *img, float scale,
                                                    Mats m, int accumulate, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        float v = scale * img[p];
        if (NM > 0) {
            double ph, la;
            mats_eval<NM>(m, p, ph, la);
            v *= expf((float)la);
        }
        acc[p] = accumulate ? acc[p] + v : v;
    }
}

// The same with the sum of what was added reduced on the way (EXP:360-361, 485-486: the reference takes np.mean of the
// per-energy reference image for its intensity-weighted mean energy): S = sum of v goes to slot (blockIdx % PSX_SUM_SLOTS) of
// `sums` -- sums[slot*16 + 0] += S, sums[slot*16 + 1] += weight * S.  The slots are 128 bytes apart: float64 atomics on ONE
// address retire at ~90 per microsecond chip-wide (16384 blocks on one word took 0.37 ms), on 32 lines they overlap.
// 4 pixels per thread and trip (16-byte loads and stores) when the maps allow; wave shuffle reduction in float64.
template <int NM>
__global__ __launch_bounds__(256) void k_accumulate_sum(float *acc, const float *img, float scale, Mats m, int accumulate,
                                                        int64_t n, double *sums, double weight, int vec) {
#pragma clang fp contract(off)                 // every product and sum rounded on its own: k_accumulate_many repeats them bit for bit
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double s = 0.0;
    const int64_t nq = vec ? n >> 2 : 0;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        const float4 iv = reinterpret_cast<const float4 *>(img)[q];
        float v[4] = {scale * iv.x, scale * iv.y, scale * iv.z, scale * iv.w};
        if (NM > 0) {
            float4 t[NM > 0 ? NM : 1];
#pragma unroll
            for (int i = 0; i < NM; ++i) t[i] = reinterpret_cast<const float4 *>(m.T[i])[q];
            double la[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                la[0] = fma(m.catt[i], (double)t[i].x, la[0]);
                la[1] = fma(m.catt[i], (double)t[i].y, la[1]);
                la[2] = fma(m.catt[i], (double)t[i].z, la[2]);
                la[3] = fma(m.catt[i], (double)t[i].w, la[3]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * expf((float)la[e]);             // rounded products and sums (no fma
        }                                                                              // contraction): k_accumulate_many
        if (acc) {                                                                     // reproduces them bit for bit
            float4 *ap = reinterpret_cast<float4 *>(acc) + q;
            if (accumulate) {
                const float4 o = *ap;
                *ap = make_float4(o.x + v[0], o.y + v[1], o.z + v[2], o.w + v[3]);
            } else {
                *ap = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        s += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
    }
    for (int64_t p = nq * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        float v = scale * img[p];
        if (NM > 0) {
            double ph, la;
            mats_eval<NM>(m, p, ph, la);
            v = v * expf((float)la);
        }
        if (acc) acc[p] = accumulate ? acc[p] + v : v;
        s += (double)v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = part[0] + part[1] + part[2] + part[3];
        double *slot = sums + (size_t)(blockIdx.x % PSX_SUM_SLOTS) * PSX_SUM_STRIDE;
        atomicAdd(&slot[0], t);
        atomicAdd(&slot[1], weight * t);
    }
}

// The same for the images of several energies in ONE pass (the energies of a detector bin, EXP:351-361): acc (+)= sum over e
// of scale[e] * img_e * exp(sum_i catt[e][i] * T_i), added in the order of e exactly as one k_accumulate_sum launch per energy
// would (the float32 sums are bit-identical); sums += (sum of every term, sum of weight[e] * term).
struct ImgBatch {
    const float *img[PSX_MAX_SRC];
    float scale[PSX_MAX_SRC];
    double weight[PSX_MAX_SRC];
    double catt[PSX_MAX_SRC][PSX_MAX_MAT];
    const float *T[PSX_MAX_MAT];
    int n_img;
};

template <int NM>
__global__ __launch_bounds__(256) void k_accumulate_many(float *acc, ImgBatch b, int accumulate, int64_t n, double *sums, int vec) {
#pragma clang fp contract(off)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double s0 = 0.0, s1 = 0.0;
    const int64_t nq = vec ? n >> 2 : 0;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        float4 t[NM > 0 ? NM : 1];
#pragma unroll
        for (int i = 0; i < NM; ++i) t[i] = reinterpret_cast<const float4 *>(b.T[i])[q];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (acc && accumulate) o = reinterpret_cast<const float4 *>(acc)[q];
        for (int e = 0; e < b.n_img; ++e) {
            const float4 iv = reinterpret_cast<const float4 *>(b.img[e])[q];
            const float sc = b.scale[e];
            float v[4] = {sc * iv.x, sc * iv.y, sc * iv.z, sc * iv.w};
            if (NM > 0) {
                double la[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    const double c = b.catt[e][i];
                    la[0] = fma(c, (double)t[i].x, la[0]);
                    la[1] = fma(c, (double)t[i].y, la[1]);
                    la[2] = fma(c, (double)t[i].z, la[2]);
                    la[3] = fma(c, (double)t[i].w, la[3]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = v[k] * expf((float)la[k]);
            }
            if (e == 0 && !accumulate) o = make_float4(v[0], v[1], v[2], v[3]);
            else o = make_float4(o.x + v[0], o.y + v[1], o.z + v[2], o.w + v[3]);
            const double t4 = ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
            s0 += t4;
            s1 = fma(b.weight[e], t4, s1);
        }
        if (acc) reinterpret_cast<float4 *>(acc)[q] = o;
    }
    for (int64_t p = nq * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        float o = (acc && accumulate) ? acc[p] : 0.f;
        for (int e = 0; e < b.n_img; ++e) {
            float v = b.scale[e] * b.img[e][p];
            if (NM > 0) {
                double la = 0.0;
#pragma unroll
                for (int i = 0; i < NM; ++i) la = fma(b.catt[e][i], (double)b.T[i][p], la);
                v = v * expf((float)la);
            }
            o = (e == 0 && !accumulate) ? v : o + v;
            s0 += (double)v;
            s1 = fma(b.weight[e], (double)v, s1);
        }
        if (acc) acc[p] = o;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_down(s0, o, 64);
        s1 += __shfl_down(s1, o, 64);
    }
    __shared__ double part[4][2];
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6][0] = s0;
        part[threadIdx.x >> 6][1] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && sums) {
        double *slot = sums + (size_t)(blockIdx.x % PSX_SUM_SLOTS) * PSX_SUM_STRIDE;
        atomicAdd(&slot[0], part[0][0] + part[1][0] + part[2][0] + part[3][0]);
        atomicAdd(&slot[1], part[0][1] + part[1][1] + part[2][1] + part[3][1]);
    }
}

__global__ __launch_bounds__(256) void k_status_scan(const float *__restrict__ img, int64_t n, unsigned *status) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const float v = img[p];
        bad |= !(fabsf(v) <= 3.0e38f);   // NaN or inf; float32 cannot hold the reference's 1e50 bound
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(status, PSX_STATUS_NONFINITE);
}

extern "C" {

int psx_transmit_wave_c64(const psx_c64 *wave_in, float amp, const float *const *T, const double *cphase,
                          const double *catt, int nmat, psx_c64 *wave_out, int64_t n, void *stream) {
    PSX_REQUIRE(wave_out != nullptr && n >= 0, "psx_transmit_wave_c64: null output or negative n");
    Mats m;
    if (int rc = pack_mats(m, T, cphase, catt, nmat)) return rc;
    if (n == 0) return 0;
    PSX_DISPATCH_NMAT(nmat, PSX_TIMED("k_transmit_wave", (hipStream_t)stream, k_transmit_wave<NM><<<ew_grid(n, 256), 256, 0, (hipStream_t)stream>>>((const float2 *)wave_in, amp, m,
                                                                      (float2 *)wave_out, n)));
    return launch_check("k_transmit_wave");
}

int psx_transmit_rt_f32(const float *I_in, float I0, const float *const *T, const double *cphase, const double *catt,
                        int nmat, float *I_out, const double *phi_in, double *phi_out, int64_t n, void *stream) {
    PSX_REQUIRE(n >= 0 && (I_out || phi_out), "psx_transmit_rt_f32: nothing to write");
    Mats m;
    if (int rc = pack_mats(m, T, cphase, catt, nmat)) return rc;
    if (n == 0) return 0;
    PSX_DISPATCH_NMAT(nmat, PSX_TIMED("k_transmit_rt", (hipStream_t)stream, k_transmit_rt<NM><<<ew_grid(n, 256), 256, 0, (hipStream_t)stream>>>(I_in, I0, m, I_out, phi_in, phi_out, n)));
    return launch_check("k_transmit_rt");
}

int psx_accumulate_f32(float *acc, const float *img, float scale, const float *const *T, const double *catt, int nmat,
                       int accumulate, int64_t n, void *stream) {
    PSX_REQUIRE(acc && img && n >= 0, "psx_accumulate_f32: null pointer or negative n");
    Mats m;
    if (int rc = pack_mats(m, T, nullptr, catt, nmat)) return rc;
    if (n == 0) return 0;
    PSX_DISPATCH_NMAT(nmat, PSX_TIMED("k_accumulate", (hipStream_t)stream, k_accumulate<NM><<<ew_grid(n, 256), 256, 0, (hipStream_t)stream>>>(acc, img, scale, m, accumulate, n)));
    return launch_check("k_accumulate");
}

int psx_accumulate_sum_f32(float *acc, const float *img, float scale, const float *const *T, const double *catt, int nmat,
                           int accumulate, int64_t n, double *sums, double weight, void *stream) {
    PSX_REQUIRE(img && sums && n >= 0, "psx_accumulate_sum_f32: null pointer or negative n");
    Mats m;
    if (int rc = pack_mats(m, T, nullptr, catt, nmat)) return rc;
    if (n == 0) return 0;
    int vec = (uintptr_t)img % 16 == 0 && (uintptr_t)acc % 16 == 0;
    for (int i = 0; i < nmat && i < PSX_MAX_MAT; ++i) vec = vec && (uintptr_t)T[i] % 16 == 0;
    int grid = ew_grid(n, 256, 4);
    if (grid > 1024) grid = 1024;                     // 4 workgroups per CU; 2 x 1024 atomics over 32 lines
    PSX_DISPATCH_NMAT(nmat, PSX_TIMED("k_accumulate_sum", (hipStream_t)stream, k_accumulate_sum<NM><<<grid, 256, 0, (hipStream_t)stream>>>(acc, img, scale, m, accumulate, n, sums, weight, vec)));
    return launch_check("k_accumulate_sum");
}

int psx_accumulate_many_f32(float *acc, const float *const *imgs, const float *scale, int n_img, const float *const *T,
                            const double *catt, int nmat, int accumulate, int64_t n, double *sums, const double *weight,
                            void *stream) {
    PSX_REQUIRE(imgs && n_img >= 1 && n_img <= PSX_MAX_SRC && n >= 0, "psx_accumulate_many_f32: 1..%d images", PSX_MAX_SRC);
    PSX_REQUIRE(acc || sums, "psx_accumulate_many_f32: neither an accumulator nor sums requested");
    Mats m;
    if (int rc = pack_mats(m, T, nullptr, nullptr, nmat)) return rc;
    if (n == 0) return 0;
    ImgBatch b = {};
    b.n_img = n_img;
    int vec = (uintptr_t)acc % 16 == 0;
    for (int i = 0; i < nmat && i < PSX_MAX_MAT; ++i) {
        b.T[i] = T[i];
        vec = vec && (uintptr_t)T[i] % 16 == 0;
    }
    for (int e = 0; e < n_img; ++e) {
        PSX_REQUIRE(imgs[e] != nullptr, "psx_accumulate_many_f32: null image %d", e);
        b.img[e] = imgs[e];
        vec = vec && (uintptr_t)imgs[e] % 16 == 0;
        b.scale[e] = scale ? scale[e] : 1.f;
        b.weight[e] = weight ? weight[e] : 0.0;
        for (int i = 0; i < nmat; ++i) b.catt[e][i] = catt ? catt[(size_t)e * nmat + i] : 0.0;
    }
    int grid = ew_grid(n, 256, 4);
    if (grid > 1024) grid = 1024;
    PSX_DISPATCH_NMAT(nmat, PSX_TIMED("k_accumulate_sum", (hipStream_t)stream, k_accumulate_many<NM><<<grid, 256, 0, (hipStream_t)stream>>>(acc, b, accumulate, n, sums, vec)));
    return launch_check("k_accumulate_many");
}

int psx_status_scan_f32(const float *img, int64_t n, unsigned *status, void *stream) {
    PSX_REQUIRE(img && status && n >= 0, "psx_status_scan_f32: null pointer or negative n");
    if (n == 0) return 0;
    PSX_TIMED("k_status_scan", (hipStream_t)stream, k_status_scan<<<ew_grid(n, 256), 256, 0, (hipStream_t)stream>>>(img, n, status));
    return launch_check("k_status_scan");
}

}  // extern "C"
