// core.hip -- error reporting, argument packing, device probe.
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"

namespace psx {

char *err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int pack_mats(Mats &m, const float *const *T, const double *cphase, const double *catt, int nmat) {
    if (nmat < 0 || nmat > PSX_MAX_MAT) return fail(PSX_E_ARG, "nmat=%d outside [0,%d]", nmat, PSX_MAX_MAT);
    if (nmat > 0 && T == nullptr) return fail(PSX_E_ARG, "T is null with nmat=%d", nmat);
    m.n = nmat;
    for (int i = 0; i < PSX_MAX_MAT; ++i) {
        m.T[i] = nullptr;
        m.cphase[i] = 0.0;
        m.catt[i] = 0.0;
    }
    for (int i = 0; i < nmat; ++i) {
        if (T[i] == nullptr) return fail(PSX_E_ARG, "T[%d] is null", i);
        m.T[i] = T[i];
        m.cphase[i] = cphase ? cphase[i] : 0.0;
        m.catt[i] = catt ? catt[i] : 0.0;
    }
    // pad up to the instantiated variant with a valid map and zero coefficients (see mats_eval)
    for (int i = nmat; i < mats_variant(nmat); ++i) m.T[i] = T[0];
    return 0;
}

// ---- per-kernel timing ----------------------------------------------------------------------------------------
namespace {
struct ProfRec {
    hipEvent_t a, b;
    int name;
};
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
std::vector<std::string> g_names;
std::vector<hipEvent_t> g_pool;

hipEvent_t take_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

unsigned long long *g_stamps = nullptr;

namespace {
const char *const g_dbg_names[DBG_COUNT] = {"no_dif", "no_pair", "no_dual", "no_dist_inner", "stamp_pass1", "stamp_round",
                                            "detect_4pass", "far_stride", "near_lds_pad", "no_p2"};
const int g_dbg_default[DBG_COUNT] = {0, 0, 0, 0, 0, 1, 0, 0, 0, 0};
std::atomic<int> g_dbg[DBG_COUNT] = {{0}, {0}, {0}, {0}, {0}, {1}, {0}, {0}, {0}, {0}};
}  // namespace

int debug_switch(DebugSwitch s) { return g_dbg[s].load(std::memory_order_relaxed); }

bool prof_enabled() { return g_prof_on; }

int prof_begin(const char *name, hipStream_t st) {
    int id = -1;
    for (size_t i = 0; i < g_names.size(); ++i)
        if (g_names[i] == name) id = (int)i;
    if (id < 0) {
        g_names.push_back(name);
        id = (int)g_names.size() - 1;
    }
    ProfRec r{take_event(), take_event(), id};
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}

void prof_end(int handle, hipStream_t st) { (void)hipEventRecord(g_recs[handle].b, st); }

}  // namespace psx

__global__ void psx_probe_kernel(int *out) { *out = 950; }

// Shader clock under load: every workgroup spins on dependent FMAs for `ticks` of the constant 100 MHz clock; thread 0 of
// workgroup 0 reads the shader-clock counter (s_memtime) and the constant one (s_memrealtime) either side of its spin.
__global__ __launch_bounds__(256) void psx_clock_probe_kernel(unsigned long long *out, unsigned ticks, float seed) {
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    float acc = seed + threadIdx.x;
    while (wall_clock64() - r0 < ticks) {
#pragma unroll
        for (int i = 0; i < 64; ++i) acc = fmaf(acc, 1.0000001f, 0.5f);
    }
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
    if (acc == 12345.678f) out[2] = 1ull;
}

extern "C" {

int psx_abi_version(void) { return 10; }   // 3: psx_debug_switch(es_active), psx_set_deterministic(1) allocation-free; 4: psx_darkfield_split_f32(num, den); 5: psx_get_deterministic; 6: psx_refract_split_f32, psx_darkfield_blur_prepared_f32(accumulate); 7: psx_set_deterministic_scale; 8: psx_detect_multi_f32; 9: psx_get_deterministic_scale; 10: psx_clock_probe

const char *psx_last_error(void) { return psx::err_buf(); }

int psx_device_ok(void) {
    int *d = nullptr;
    int h = 0;
    if (hipMalloc(&d, sizeof(int)) != hipSuccess) {
        psx::fail(1, "hipMalloc failed: no usable HIP device");
        return 0;
    }
    (void)hipMemset(d, 0, sizeof(int));
    psx_probe_kernel<<<1, 1>>>(d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess || h != 950) {
        psx::fail((int)e, "gfx950 code object not runnable on this device: %s", hipGetErrorString(e));
        return 0;
    }
    return 1;
}

int psx_clock_probe(float *mhz, void *stream) {
    if (!mhz) return psx::fail(PSX_E_ARG, "psx_clock_probe: null pointer");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *d = nullptr, h[2] = {0ull, 0ull};
    PSX_HIP(hipMalloc((void **)&d, 3 * sizeof(unsigned long long)));
    psx_clock_probe_kernel<<<psx::current_cu_count(), 256, 0, st>>>(d, 3000u, 1.0f);       // 30 us on every CU
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess || h[1] == 0ull) return psx::fail((int)e, "psx_clock_probe: %s", hipGetErrorString(e));
    *mhz = (float)((double)h[0] / (double)h[1] * 100.0);
    return 0;
}

int psx_debug_stamps(void *buf) {
    psx::g_stamps = (unsigned long long *)buf;
    return 0;
}

int psx_debug_switch(const char *name, int value) {
    if (!name) return psx::fail(PSX_E_ARG, "psx_debug_switch: null name");
    for (int i = 0; i < psx::DBG_COUNT; ++i)
        if (!strcmp(name, psx::g_dbg_names[i])) {
            psx::g_dbg[i].store(value, std::memory_order_relaxed);
            return 0;
        }
    return psx::fail(PSX_E_ARG, "psx_debug_switch: unknown switch '%s'", name);
}

int psx_debug_switches_active(char *buf, size_t cap) {
    if (!buf || cap == 0) return psx::fail(PSX_E_ARG, "psx_debug_switches_active: null buffer");
    size_t off = 0;
    buf[0] = 0;
    for (int i = 0; i < psx::DBG_COUNT; ++i) {
        const int v = psx::g_dbg[i].load(std::memory_order_relaxed);
        if (v == psx::g_dbg_default[i]) continue;
        int n = snprintf(buf + off, cap - off, "%s%s=%d", off ? " " : "", psx::g_dbg_names[i], v);
        if (n < 0 || (size_t)n >= cap - off) return psx::fail(PSX_E_ARG, "psx_debug_switches_active: buffer too small");
        off += (size_t)n;
    }
    return 0;
}

int psx_profile_enable(int on) {
    for (auto &r : psx::g_recs) {
        psx::g_pool.push_back(r.a);
        psx::g_pool.push_back(r.b);
    }
    psx::g_recs.clear();
    psx::g_prof_on = on != 0;
    return 0;
}

int psx_profile_summary(char *buf, size_t cap) {
    if (!buf || cap == 0) return psx::fail(PSX_E_ARG, "psx_profile_summary: null buffer");
    std::vector<double> total(psx::g_names.size(), 0.0);
    std::vector<int> count(psx::g_names.size(), 0);
    for (auto &r : psx::g_recs) {
        hipError_t e = hipEventSynchronize(r.b);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return psx::fail((int)e, "psx_profile_summary: %s", hipGetErrorString(e));
        total[r.name] += ms;
        count[r.name] += 1;
    }
    size_t off = 0;
    buf[0] = 0;
    for (size_t i = 0; i < psx::g_names.size(); ++i) {
        if (!count[i]) continue;
        int n = snprintf(buf + off, cap - off, "%s %d %.6f\n", psx::g_names[i].c_str(), count[i], total[i]);
        if (n < 0 || (size_t)n >= cap - off) return psx::fail(PSX_E_ARG, "psx_profile_summary: buffer too small");
        off += (size_t)n;
    }
    return 0;
}

}  // extern "C"
