// fresnel_plan.hpp -- the plan object behind psx_fresnel_*: grid geometry + engine-specific state.
#pragma once
#include <rocfft/rocfft.h>

#include "common.hpp"

#define PSX_ROCFFT(expr)                                                                          \
    do {                                                                                          \
        rocfft_status s__ = (expr);                                                               \
        if (s__ != rocfft_status_success)                                                         \
            return psx::fail(1000 + (int)s__, "%s:%d: %s -> rocfft_status %d", __FILE__, __LINE__, #expr, (int)s__); \
    } while (0)

namespace psx {

int rocfft_ensure_setup();   // rocfft_setup() once per process (fresnel.hip)

struct LdsEngine;   // fresnel_lds.hip

struct RocfftEngine {
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info = nullptr;
    void *work = nullptr;
    size_t work_bytes = 0;
    float2 *spec = nullptr;   // [Px][Py] padded wave, transformed in place
    float2 *prod = nullptr;   // [Px][Py] spectrum x chirp, inverse-transformed in place (one distance at a time)
    float2 *cx = nullptr;     // [Px] chirp along axis 0 (FFT order), carries the global phase and 1/(Px*Py)
    float2 *cy = nullptr;     // [Py]
};

}  // namespace psx

struct psx_fresnel_plan {
    int Nx, Ny, margin, Px, Py, max_dist, engine;
    size_t bytes;
    psx::RocfftEngine *rf;
    psx::LdsEngine *lds;
};

namespace psx {

// engine entry points (same contract as psx_fresnel_propagate, arguments already validated and packed)
struct PropArgs {
    const float2 *wave_in;
    float amp;
    Mats m;
    int n_dist;
    const double *a, *gphase;
    double du_x, du_y;
    float2 *const *wave_out;
    float *const *inten_out;
    const float *inten_scale;
    int accumulate;
    hipStream_t stream;
};

// a batch of source waves over the same thickness maps (psx_fresnel_propagate_sources); arrays indexed [source] or
// [source * n_dist + distance], coefficients [source * maps.n + material]
struct SourcesArgs {
    int n_src, n_dist;
    const float2 *const *wave_in;   // may be NULL (unit waves), entries may be NULL
    const float *amp;
    Mats maps;                      // T and n; its coefficients are not used
    const double *cphase, *catt;
    const double *a, *gphase;
    double du_x, du_y;
    float2 *const *wave_out;
    float *const *inten_out;
    const float *inten_scale;
    hipStream_t stream;
};

int rocfft_engine_create(psx_fresnel_plan *p);
void rocfft_engine_destroy(psx_fresnel_plan *p);
int rocfft_engine_propagate(psx_fresnel_plan *p, const PropArgs &a);

bool lds_engine_supported(int Nx, int Ny, int margin);
int lds_engine_create(psx_fresnel_plan *p);
void lds_engine_destroy(psx_fresnel_plan *p);
int lds_engine_propagate(psx_fresnel_plan *p, const PropArgs &a);
void lds_engine_work_queue(psx_fresnel_plan *p, int on);
int lds_engine_propagate_sources(psx_fresnel_plan *p, const SourcesArgs &a);

// psi(p) = amp * wave_in * transmission at UN-padded pixel p
template <int NM>
__device__ __forceinline__ float2 source_wave(const float2 *__restrict__ wave_in, float amp, const Mats &m, int64_t p) {
    float2 w = wave_in ? wave_in[p] : make_float2(1.f, 0.f);
    float a = amp;
    if (NM > 0) {
        double ph, la;
        mats_eval<NM>(m, p, ph, la);
        float c, s;
        cis_f64(ph, c, s);
        a *= exp_att(la);
        w = make_float2(w.x * c - w.y * s, w.x * s + w.y * c);
    }
    return make_float2(a * w.x, a * w.y);
}

// np.pad(..., mode='reflect') index: mirror without repeating the edge sample (EXP:237, DET:93)
__host__ __device__ __forceinline__ int reflect_index(int q, int n) {
    if (q < 0) q = -q;
    if (q >= n) q = 2 * (n - 1) - q;
    return q;
}

}  // namespace psx
