// fresnel.hip -- Fresnel angular-spectrum propagator, plan management + the rocFFT engine (K3-K8).
//
// Replaces Experiment.wavePropagation (Experiment.py:219-252):  reflect-pad 15 -> FFT2 -> x chirp -> IFFT2 -> crop.
// rocFFT engine, per propagation of one wave to n_dist distances:
//   k_pad_transmit : psi = amp*wave_in*transmission evaluated straight into the reflect-padded [Px][Py] buffer (K1+K3)
//   rocFFT forward, in place (K4)                                        -- shared by all distances
//   k_chirp_table  : exp(-i a u^2) per axis in float64 on the device, FFT order (fftshift folded into the index), with
//                    the global phase exp(i k z/M) and the 1/(Px*Py) of the unnormalised inverse folded in (K5)
//   k_chirp_mul    : prod = spec * cx[i] * cy[j]  (separable chirp, two length-P tables)  (K6)
//   rocFFT inverse, in place (K7)
//   k_crop_out     : crop + optional |.|^2 accumulate (K8)
// P = N+30 is prime-laden for the headline sizes (4126 = 2*2063), so rocFFT runs Bluestein here; the LDS engine
// (fresnel_lds.hip) avoids that and is preferred whenever a padded row fits LDS.
#include "fresnel_plan.hpp"

using namespace psx;

namespace {

template <int NM>
__global__ __launch_bounds__(256) void k_pad_transmit(const float2 *__restrict__ wave_in, float amp, Mats m,
                                                      float2 *__restrict__ out, int Nx, int Ny, int margin, int Px,
                                                      int Py) {
    const int64_t n = (int64_t)Px * Py;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int pi = (int)(q / Py), pj = (int)(q - (int64_t)pi * Py);
        const int i = reflect_index(pi - margin, Nx), j = reflect_index(pj - margin, Ny);
        out[q] = source_wave<NM>(wave_in, amp, m, (int64_t)i * Ny + j);
    }
}

// table[i] = (sre + i*sim) * exp(-i * a * (f_i*du)^2),  f_i = i for i < ceil(P/2) else i-P   (EXP:243-250)
__global__ void k_chirp_table(float2 *table, int P, double a, double du, double sre, double sim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int f = (i < (P + 1) / 2) ? i : i - P;
    const double u = (double)f * du;
    const double ph = -a * u * u;
    const double r = ph - PSX_TWO_PI * rint(ph * PSX_INV_TWO_PI);
    double s, c;
    sincos(r, &s, &c);
    table[i] = make_float2((float)(sre * c - sim * s), (float)(sre * s + sim * c));
}

__global__ __launch_bounds__(256) void k_chirp_mul(const float2 *__restrict__ spec, const float2 *__restrict__ cx,
                                                   const float2 *__restrict__ cy, float2 *__restrict__ prod, int Px,
                                                   int Py) {
    const int64_t n = (int64_t)Px * Py;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int pi = (int)(q / Py), pj = (int)(q - (int64_t)pi * Py);
        const float2 a = cx[pi], b = cy[pj], s = spec[q];
        const float2 h = make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
        prod[q] = make_float2(s.x * h.x - s.y * h.y, s.x * h.y + s.y * h.x);
    }
}

__global__ __launch_bounds__(256) void k_crop_out(const float2 *__restrict__ padded, float2 *__restrict__ wave_out,
                                                  float *__restrict__ inten_out, float inten_scale, int accumulate,
                                                  int Nx, int Ny, int margin, int Py) {
    const int64_t n = (int64_t)Nx * Ny;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(p / Ny), j = (int)(p - (int64_t)i * Ny);
        const float2 v = padded[(int64_t)(i + margin) * Py + (j + margin)];
        if (wave_out) wave_out[p] = v;
        if (inten_out) {
            const float I = inten_scale * (v.x * v.x + v.y * v.y);
            inten_out[p] = accumulate ? inten_out[p] + I : I;
        }
    }
}

bool g_rocfft_ready = false;

}  // namespace

namespace psx {

int rocfft_ensure_setup() {
    if (!g_rocfft_ready) {
        PSX_ROCFFT(rocfft_setup());
        g_rocfft_ready = true;
    }
    return 0;
}

int rocfft_engine_create(psx_fresnel_plan *p) {
    if (int rc = rocfft_ensure_setup()) return rc;
    RocfftEngine *e = new RocfftEngine();
    p->rf = e;
    const size_t lengths[2] = {(size_t)p->Py, (size_t)p->Px};   // rocFFT lengths are fastest-first
    PSX_ROCFFT(rocfft_plan_create(&e->fwd, rocfft_placement_inplace, rocfft_transform_type_complex_forward,
                                  rocfft_precision_single, 2, lengths, 1, nullptr));
    PSX_ROCFFT(rocfft_plan_create(&e->inv, rocfft_placement_inplace, rocfft_transform_type_complex_inverse,
                                  rocfft_precision_single, 2, lengths, 1, nullptr));
    size_t wf = 0, wi = 0;
    PSX_ROCFFT(rocfft_plan_get_work_buffer_size(e->fwd, &wf));
    PSX_ROCFFT(rocfft_plan_get_work_buffer_size(e->inv, &wi));
    e->work_bytes = wf > wi ? wf : wi;
    PSX_ROCFFT(rocfft_execution_info_create(&e->info));
    if (e->work_bytes) {
        PSX_HIP(hipMalloc(&e->work, e->work_bytes));
        PSX_ROCFFT(rocfft_execution_info_set_work_buffer(e->info, e->work, e->work_bytes));
    }
    const size_t img = sizeof(float2) * (size_t)p->Px * (size_t)p->Py;
    PSX_HIP(hipMalloc((void **)&e->spec, img));
    PSX_HIP(hipMalloc((void **)&e->prod, img));
    PSX_HIP(hipMalloc((void **)&e->cx, sizeof(float2) * (size_t)p->Px));
    PSX_HIP(hipMalloc((void **)&e->cy, sizeof(float2) * (size_t)p->Py));
    p->bytes += 2 * img + e->work_bytes + sizeof(float2) * (size_t)(p->Px + p->Py);
    return 0;
}

void rocfft_engine_destroy(psx_fresnel_plan *p) {
    RocfftEngine *e = p->rf;
    if (!e) return;
    if (e->fwd) rocfft_plan_destroy(e->fwd);
    if (e->inv) rocfft_plan_destroy(e->inv);
    if (e->info) rocfft_execution_info_destroy(e->info);
    (void)hipFree(e->work);
    (void)hipFree(e->spec);
    (void)hipFree(e->prod);
    (void)hipFree(e->cx);
    (void)hipFree(e->cy);
    delete e;
    p->rf = nullptr;
}

int rocfft_engine_propagate(psx_fresnel_plan *p, const PropArgs &a) {
    RocfftEngine *e = p->rf;
    hipStream_t st = a.stream;
    const int64_t npad = (int64_t)p->Px * p->Py, n = (int64_t)p->Nx * p->Ny;
    PSX_DISPATCH_NMAT(a.m.n, PSX_TIMED("k_pad_transmit", st, k_pad_transmit<NM><<<ew_grid(npad, 256), 256, 0, st>>>(a.wave_in, a.amp, a.m, e->spec, p->Nx, p->Ny, p->margin, p->Px,
                                                      p->Py)));
    if (int rc = launch_check("k_pad_transmit")) return rc;
    bool need_fft = false;
    for (int d = 0; d < a.n_dist; ++d) {
        if (a.a[d] == 0.0) {   // z == 0: the reference returns its input untouched (EXP:233-234)
            PSX_TIMED("k_crop_out", st, k_crop_out<<<ew_grid(n, 256), 256, 0, st>>>(e->spec, a.wave_out ? a.wave_out[d] : nullptr,
                                                        a.inten_out ? a.inten_out[d] : nullptr,
                                                        a.inten_scale ? a.inten_scale[d] : 1.f, a.accumulate, p->Nx,
                                                        p->Ny, p->margin, p->Py));
            if (int rc = launch_check("k_crop_out")) return rc;
        } else {
            need_fft = true;
        }
    }
    if (!need_fft) return 0;
    PSX_ROCFFT(rocfft_execution_info_set_stream(e->info, st));
    void *buf[1] = {e->spec};
    {
        ProfScope ps("rocfft_forward", st);
        PSX_ROCFFT(rocfft_execute(e->fwd, buf, nullptr, e->info));
    }
    const double norm = 1.0 / ((double)p->Px * (double)p->Py);
    for (int d = 0; d < a.n_dist; ++d) {
        if (a.a[d] == 0.0) continue;
        const double g = a.gphase ? a.gphase[d] : 0.0;
        // k*z/M is ~1e11 rad: libm's cos/sin reduce the float64 argument exactly (as numpy does for EXP:250);
        // a remainder by a rounded 2*pi would already be off by 1e-5 rad
        PSX_TIMED("k_chirp_table", st, k_chirp_table<<<(int)cdiv(p->Px, 256), 256, 0, st>>>(e->cx, p->Px, a.a[d], a.du_x, norm * std::cos(g),
                                                             norm * std::sin(g)));
        PSX_TIMED("k_chirp_table", st, k_chirp_table<<<(int)cdiv(p->Py, 256), 256, 0, st>>>(e->cy, p->Py, a.a[d], a.du_y, 1.0, 0.0));
        PSX_TIMED("k_chirp_mul", st, k_chirp_mul<<<ew_grid(npad, 256), 256, 0, st>>>(e->spec, e->cx, e->cy, e->prod, p->Px, p->Py));
        if (int rc = launch_check("k_chirp_mul")) return rc;
        void *pb[1] = {e->prod};
        {
            ProfScope ps("rocfft_inverse", st);
            PSX_ROCFFT(rocfft_execute(e->inv, pb, nullptr, e->info));
        }
        PSX_TIMED("k_crop_out", st, k_crop_out<<<ew_grid(n, 256), 256, 0, st>>>(e->prod, a.wave_out ? a.wave_out[d] : nullptr,
                                                    a.inten_out ? a.inten_out[d] : nullptr,
                                                    a.inten_scale ? a.inten_scale[d] : 1.f, a.accumulate, p->Nx, p->Ny,
                                                    p->margin, p->Py));
        if (int rc = launch_check("k_crop_out")) return rc;
    }
    return 0;
}

}  // namespace psx

extern "C" {

int psx_fresnel_plan_create(int Nx, int Ny, int margin, int max_dist, int engine, psx_fresnel_plan **plan) {
    PSX_REQUIRE(plan != nullptr, "psx_fresnel_plan_create: null plan pointer");
    *plan = nullptr;
    PSX_REQUIRE(Nx >= 2 && Ny >= 2, "psx_fresnel_plan_create: grid %dx%d too small", Nx, Ny);
    PSX_REQUIRE(margin >= 0 && margin < Nx && margin < Ny,
                "psx_fresnel_plan_create: reflect margin %d needs a grid larger than it", margin);
    PSX_REQUIRE(max_dist >= 1 && max_dist <= PSX_MAX_DIST, "psx_fresnel_plan_create: max_dist=%d outside [1,%d]",
                max_dist, PSX_MAX_DIST);
    PSX_REQUIRE(engine >= PSX_ENGINE_AUTO && engine <= PSX_ENGINE_LDS, "psx_fresnel_plan_create: unknown engine %d",
                engine);
    if (engine == PSX_ENGINE_AUTO) engine = lds_engine_supported(Nx, Ny, margin) ? PSX_ENGINE_LDS : PSX_ENGINE_ROCFFT;
    if (engine == PSX_ENGINE_LDS && !lds_engine_supported(Nx, Ny, margin))
        return fail(PSX_E_UNSUPPORTED, "psx_fresnel_plan_create: LDS engine cannot hold a %dx%d grid row in LDS", Nx, Ny);
    psx_fresnel_plan *p = new psx_fresnel_plan();
    p->Nx = Nx; p->Ny = Ny; p->margin = margin; p->Px = Nx + 2 * margin; p->Py = Ny + 2 * margin;
    p->max_dist = max_dist; p->engine = engine; p->bytes = 0; p->rf = nullptr; p->lds = nullptr;
    int rc = engine == PSX_ENGINE_LDS ? lds_engine_create(p) : rocfft_engine_create(p);
    if (rc) {
        psx_fresnel_plan_destroy(p);
        return rc;
    }
    *plan = p;
    return 0;
}

int psx_fresnel_plan_destroy(psx_fresnel_plan *p) {
    if (!p) return 0;
    rocfft_engine_destroy(p);
    lds_engine_destroy(p);
    delete p;
    return 0;
}

int psx_fresnel_plan_engine(const psx_fresnel_plan *p) { return p ? p->engine : PSX_E_ARG; }

size_t psx_fresnel_plan_bytes(const psx_fresnel_plan *p) { return p ? p->bytes : 0; }

int psx_fresnel_plan_work_queue(psx_fresnel_plan *p, int on) {
    PSX_REQUIRE(p != nullptr, "psx_fresnel_plan_work_queue: null plan");
    if (p->engine == PSX_ENGINE_LDS) lds_engine_work_queue(p, on);
    return 0;
}

int psx_fresnel_propagate(psx_fresnel_plan *plan, const psx_c64 *wave_in, float amp, const float *const *T,
                          const double *cphase, const double *catt, int nmat, int n_dist, const double *a,
                          const double *gphase, double du_x, double du_y, psx_c64 *const *wave_out,
                          float *const *inten_out, const float *inten_scale, int accumulate, void *stream) {
    PSX_REQUIRE(plan != nullptr, "psx_fresnel_propagate: null plan");
    PSX_REQUIRE(n_dist >= 1 && n_dist <= plan->max_dist, "psx_fresnel_propagate: n_dist=%d outside [1,%d]", n_dist,
                plan->max_dist);
    PSX_REQUIRE(a != nullptr, "psx_fresnel_propagate: null distance table");
    PSX_REQUIRE(wave_out || inten_out, "psx_fresnel_propagate: no output requested");
    for (int d = 0; d < n_dist; ++d) {
        const bool w = wave_out && wave_out[d], i = inten_out && inten_out[d];
        PSX_REQUIRE(w || i, "psx_fresnel_propagate: distance %d has no output", d);
        PSX_REQUIRE(std::isfinite(a[d]), "psx_fresnel_propagate: a[%d] is not finite", d);
    }
    PropArgs pa;
    if (int rc = pack_mats(pa.m, T, cphase, catt, nmat)) return rc;
    pa.wave_in = (const float2 *)wave_in; pa.amp = amp; pa.n_dist = n_dist; pa.a = a; pa.gphase = gphase;
    pa.du_x = du_x; pa.du_y = du_y; pa.wave_out = (float2 *const *)wave_out; pa.inten_out = inten_out;
    pa.inten_scale = inten_scale; pa.accumulate = accumulate; pa.stream = (hipStream_t)stream;
    return plan->engine == PSX_ENGINE_LDS ? lds_engine_propagate(plan, pa) : rocfft_engine_propagate(plan, pa);
}

int psx_fresnel_propagate_sources(psx_fresnel_plan *plan, int n_src, int n_dist, const psx_c64 *const *wave_in, const float *amp,
                                  const float *const *T, const double *cphase, const double *catt, int nmat, const double *a,
                                  const double *gphase, double du_x, double du_y, psx_c64 *const *wave_out,
                                  float *const *inten_out, const float *inten_scale, void *stream) {
    PSX_REQUIRE(plan != nullptr, "psx_fresnel_propagate_sources: null plan");
    PSX_REQUIRE(n_src >= 1 && n_src <= PSX_MAX_SRC, "psx_fresnel_propagate_sources: n_src=%d outside [1,%d]", n_src, PSX_MAX_SRC);
    PSX_REQUIRE(n_dist >= 1 && n_dist <= plan->max_dist, "psx_fresnel_propagate_sources: n_dist=%d outside [1,%d]", n_dist,
                plan->max_dist);
    PSX_REQUIRE(a != nullptr && amp != nullptr, "psx_fresnel_propagate_sources: null distance or amplitude table");
    PSX_REQUIRE(wave_out || inten_out, "psx_fresnel_propagate_sources: no output requested");
    for (int v = 0; v < n_src * n_dist; ++v) {
        const bool w = wave_out && wave_out[v], i = inten_out && inten_out[v];
        PSX_REQUIRE(w || i, "psx_fresnel_propagate_sources: source %d, distance %d has no output", v / n_dist, v % n_dist);
        PSX_REQUIRE(std::isfinite(a[v]), "psx_fresnel_propagate_sources: a[%d] is not finite", v);
    }
    SourcesArgs sa;
    if (int rc = pack_mats(sa.maps, T, nullptr, nullptr, nmat)) return rc;
    sa.n_src = n_src; sa.n_dist = n_dist; sa.wave_in = (const float2 *const *)wave_in; sa.amp = amp;
    sa.cphase = cphase; sa.catt = catt; sa.a = a; sa.gphase = gphase; sa.du_x = du_x; sa.du_y = du_y;
    sa.wave_out = (float2 *const *)wave_out; sa.inten_out = inten_out; sa.inten_scale = inten_scale;
    sa.stream = (hipStream_t)stream;
    if (plan->engine == PSX_ENGINE_LDS) return lds_engine_propagate_sources(plan, sa);
    for (int s = 0; s < n_src; ++s) {                    // rocFFT engine: one source at a time
        PropArgs pa;
        if (int rc = pack_mats(pa.m, T, cphase ? cphase + (size_t)s * nmat : nullptr, catt ? catt + (size_t)s * nmat : nullptr, nmat))
            return rc;
        pa.wave_in = wave_in ? (const float2 *)wave_in[s] : nullptr; pa.amp = amp[s]; pa.n_dist = n_dist;
        pa.a = a + (size_t)s * n_dist; pa.gphase = gphase ? gphase + (size_t)s * n_dist : nullptr;
        pa.du_x = du_x; pa.du_y = du_y;
        pa.wave_out = wave_out ? (float2 *const *)wave_out + (size_t)s * n_dist : nullptr;
        pa.inten_out = inten_out ? inten_out + (size_t)s * n_dist : nullptr;
        pa.inten_scale = inten_scale ? inten_scale + (size_t)s * n_dist : nullptr;
        pa.accumulate = 0; pa.stream = (hipStream_t)stream;
        if (int rc = rocfft_engine_propagate(plan, pa)) return rc;
    }
    return 0;
}

}  // extern "C"
