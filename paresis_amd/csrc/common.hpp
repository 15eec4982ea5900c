// common.hpp -- shared host/device helpers of libparesis_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "paresis_hip.h"

namespace psx {

// ---- thread-local error text behind psx_last_error() --------------------------------------------------------
char *err_buf();
int fail(int code, const char *fmt, ...);

#define PSX_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return psx::fail((int)e__, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
    } while (0)

#define PSX_REQUIRE(cond, ...)                         \
    do {                                               \
        if (!(cond)) return psx::fail(PSX_E_ARG, __VA_ARGS__); \
    } while (0)

inline int launch_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

// ---- a stack of thickness maps with per-map coefficients, passed to kernels by value ---------------------------
struct Mats {
    const float *T[PSX_MAX_MAT];
    double cphase[PSX_MAX_MAT];
    double catt[PSX_MAX_MAT];
    int n;
};

// Validates (T, cphase, catt, nmat) and packs them.  Null coefficient arrays mean zeros.
int pack_mats(Mats &m, const float *const *T, const double *cphase, const double *catt, int nmat);

// ---- device math ---------------------------------------------------------------------------------------------
#define PSX_TWO_PI 6.283185307179586476925286766559
#define PSX_INV_TWO_PI 0.15915494309189533576888376337251

// exp(i*ph) for a float64 phase of any magnitude: the phase in REVOLUTIONS is reduced to [-0.5, 0.5] in float64 and goes
// through the hardware's v_sin_f32 / v_cos_f32, which take revolutions: two quarter-rate instructions instead of the ~50 of
// sincosf (round 4; the transposing pre-pass of the Fresnel engine was bound by this arithmetic: 49 us of a 135 us step at
// 2048^2).  Measured on gfx950 over 2^26 points of [-0.5, 0.5) (tools/sincos_accuracy.hip): max |error| 1.25e-7 for either
// function, against 0.6-0.7e-7 for sincosf -- both below the 1.9e-7 rad that rounding the reduced angle to float32 costs.
__device__ __forceinline__ void cis_f64(double ph, float &c, float &s) {
    const double t = ph * PSX_INV_TWO_PI;
    const float rev = (float)(t - rint(t));
    s = __builtin_amdgcn_sinf(rev);
    c = __builtin_amdgcn_cosf(rev);
}

// exp(x) for a float64 log-attenuation of a few units at most: 2^(x log2 e) through v_exp_f32 (1 ulp); the product is formed
// in float64 and rounded once (half an ulp of a number of a few units: <= 2e-7 relative in the result)
__device__ __forceinline__ float exp_att(double la) { return __builtin_amdgcn_exp2f((float)(la * 1.4426950408889634)); }

// sum_m cphase[m]*T[m][p] and sum_m catt[m]*T[m][p] in float64.
// NM is the compile-time material count the kernel was instantiated for (pack_mats pads unused slots with T[0] and zero
// coefficients), so the NM loads are unconditional and independent: the compiler issues them together and waits once.
// (A run-time `if (i < n)` around each load makes hipcc wait vmcnt(0) per material -- guide, "register or load" trap.)
template <int NM>
__device__ __forceinline__ void mats_eval(const Mats &m, int64_t p, double &ph, double &la) {
    float t[NM > 0 ? NM : 1];
#pragma unroll
    for (int i = 0; i < NM; ++i) t[i] = m.T[i][p];
    ph = 0.0;
    la = 0.0;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        ph = fma(m.cphase[i], (double)t[i], ph);
        la = fma(m.catt[i], (double)t[i], la);
    }
}

// smallest instantiated count >= n
inline int mats_variant(int n) { return n <= 4 ? n : 8; }

// instantiate the statement for the variant matching n; NM is the compile-time count inside it
#define PSX_DISPATCH_NMAT(n, ...)                    \
    switch (psx::mats_variant(n)) {                  \
        case 0: { constexpr int NM = 0; __VA_ARGS__; } break; \
        case 1: { constexpr int NM = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int NM = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int NM = 3; __VA_ARGS__; } break; \
        case 4: { constexpr int NM = 4; __VA_ARGS__; } break; \
        default: { constexpr int NM = 8; __VA_ARGS__; } break; \
    }

// diagnostic phase-timestamp buffer (psx_debug_stamps); null in every timed run
extern unsigned long long *g_stamps;

// Diagnostic A/B switches (psx_debug_switch): process-wide, all 0 by default (stamp_round: 1), never read from the
// environment -- a stray variable must not change what a product run computes.  The host code reads them per call.
enum DebugSwitch {
    DBG_NO_DIF = 0,        // 16384^2-class lines through the block x segment partition instead of the two-round DIF convolution
    DBG_NO_PAIR,           // partitioned engine with M-point products (round-1 form)
    DBG_NO_DUAL,           // pass 1 without the forward transform shared by two distances
    DBG_NO_DIST_INNER,     // pass 1: every (distance, line group) pair its own work item
    DBG_STAMP_PASS1,       // psx_debug_stamps records pass 1 instead of pass 2
    DBG_STAMP_ROUND,       // which round of a unit the stamps are taken in
    DBG_DETECT_4PASS,      // detector stages as four passes instead of fused pairs
    DBG_FAR_STRIDE,        // far-ray replay: lists handed to the waves in a strided order (value = stride; 0: tile order)
    DBG_NEAR_LDS_PAD,      // refraction tile kernel: KiB of LDS added to the launch (occupancy probe: beyond ~2 KiB one workgroup per CU)
    DBG_NO_P2,             // Fresnel LDS engine: lines through the 576 R3-point transforms even where a power of two serves (read when a plan is created)
    DBG_COUNT
};
int debug_switch(DebugSwitch s);

// ---- optional per-kernel timing with HIP events on the launch stream (psx_profile_*) ---------------------------------
// Off by default: a ProfScope then costs one branch.  When on, every kernel launch of the library is bracketed by two
// events recorded on the stream it is launched on; psx_profile_summary() resolves them after the work has drained.
bool prof_enabled();
int prof_begin(const char *name, hipStream_t st);
void prof_end(int handle, hipStream_t st);

struct ProfScope {
    int h;
    hipStream_t st;
    ProfScope(const char *name, hipStream_t s) : h(prof_enabled() ? prof_begin(name, s) : -1), st(s) {}
    ~ProfScope() {
        if (h >= 0) prof_end(h, st);
    }
};

// launch statement(s) bracketed by a timing scope:  PSX_TIMED("k_name", stream, k_name<<<g, b, 0, stream>>>(args));
#define PSX_TIMED(name, st, ...)        \
    do {                                \
        psx::ProfScope ps__(name, st);  \
        __VA_ARGS__;                    \
    } while (0)

// Per-device one-time set-up (function attributes such as the dynamic LDS limit belong to the device's copy of the code
// object): true the first time it is called for `mask` on the CURRENT device.  Nothing in the library caches a device
// property in a process-wide static, so one process may drive several GPUs (one host thread per GPU).
inline bool first_on_device(std::atomic<unsigned long long> &mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    return !(mask.fetch_or(bit) & bit);
}

// compute units of the current device (queried per call: ~0.1 us, no cache to go stale when the device changes)
inline int current_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 8;
    return n < 8 ? 8 : n;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8, capped (guide G11)
inline int ew_grid(int64_t n, int block, int per_thread = 1) {
    int64_t g = cdiv(n, (int64_t)block * per_thread);
    if (g > 2048 * 8) g = 2048 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace psx
