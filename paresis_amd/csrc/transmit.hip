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
        const float a = amp * expf((float)la);
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
            for (int e = 0; e < 4; ++e) v[e] *= expf((float)la[e]);
        }
        if (acc) {
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
            v *= expf((float)la);
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

int psx_status_scan_f32(const float *img, int64_t n, unsigned *status, void *stream) {
    PSX_REQUIRE(img && status && n >= 0, "psx_status_scan_f32: null pointer or negative n");
    if (n == 0) return 0;
    PSX_TIMED("k_status_scan", (hipStream_t)stream, k_status_scan<<<ew_grid(n, 256), 256, 0, (hipStream_t)stream>>>(img, n, status));
    return launch_check("k_status_scan");
}

}  // extern "C"
