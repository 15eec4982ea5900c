// fresnel_lds.hip -- LDS-resident FFT-convolution engine for the Fresnel propagator (K1 + K3..K8 in two kernels).
//
// Replaces Experiment.wavePropagation (Experiment.py:219-252) without ever forming the padded 2-D spectrum in HBM.
//
// The reference operator  crop(IDFT2_P(H * DFT2_P(reflect_pad(psi))))  with H(u,v) = c(u)c(v) separable is a 1-D
// circular convolution of period P = N+2*margin along axis 0 followed by one along axis 1.  Each 1-D line convolution
//     out[n] = sum_{d<P} h[d] * x_per[n + margin - d],   h = IDFT_P(c),   x_per = periodic extension of the reflect pad,
// is evaluated exactly as a linear convolution through a power-friendly FFT of size M >= N+P-1 (M = 576*R3,
// R3 in {2,4,8,16}: 9216 for N = 4096) that lives entirely in the 160 KiB LDS of one CU:
//     line samples (transmission evaluated while loading) -> periodic/reflected images written to LDS ->
//     in-place DIF stages radix 24, 24, R3 -> multiply by FFT_M(h) (digit-reversed table, 1/M folded in) ->
//     in-place inverse stages R3, 24, 24 -> the N wanted outputs go straight from registers to HBM.
// P = 4126 = 2*2063 forces Bluestein in a library FFT (two length-8192+ transforms per 1-D DFT, forward AND inverse);
// here one forward + one inverse length-9216 transform per line does the whole forward-chirp-inverse of that axis, so a
// propagation costs 2 passes x (8 B read + 8 B written) per pixel instead of 4 padded FFT passes.
//
// Pass 1 runs along axis 0 (lines = columns, strided reads of the thickness maps / input wave) and writes its result
// TRANSPOSED, pass 2 runs along the transposed axis 0 (= original rows) and writes the final image: every global store
// of both passes is a contiguous line.  Several distances share pass 1's forward transform (its spectrum stays in
// registers while each distance's kernel is applied), e.g. Experiment.py:341 and :349.
#include <cstdlib>
#include <vector>

#include "fft_regs.hpp"
#include "fresnel_plan.hpp"

using namespace psx;

namespace {

// One 12-wave workgroup per CU owns the whole LDS: 3 waves per SIMD are needed to keep the vector pipes issuing (a
// 6-wave workgroup with two butterflies per thread measured 334 us per 4096^2 pass: VALU busy only a third of it).
constexpr int T = 768;        // threads per workgroup = radix-24 butterflies per stage
constexpr int TOT = 18432;    // complex points resident in LDS per workgroup = LINES * M
constexpr int RAD = 24;       // radix of the two big stages

__host__ __device__ constexpr int phys(int p) { return p + (p >> 5); }   // one pad slot per 32: conflict-free slabs

struct LineArgs {
    const float2 *src;      // input wave (may be null: unit wave)
    float amp;
    Mats m;
    int N, nlines, margin, P, L;
    int64_t in_stride;      // sample i of line l is pixel i*in_stride + l
    int64_t out_ld;         // output sample i of line l goes to l*out_ld + i
    const float2 *twA, *twB;   // [n][24] stage twiddles
    int n_dist;
    const float2 *H[PSX_MAX_DIST];
    float2 *wave_out[PSX_MAX_DIST];
    float *inten_out[PSX_MAX_DIST];
    float scale[PSX_MAX_DIST];
    float2 gph[PSX_MAX_DIST];
    int accumulate;
    unsigned long long *stamps;   // optional diagnostics: 16 phase timestamps per workgroup (psx_debug_stamps)
};

// phase timestamp of wave 0 (diagnostic runs only; a null buffer costs one scalar branch)
#define PSX_STAMP(k)                                                             \
    do {                                                                         \
        if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); \
    } while (0)

// orders the LDS traffic of ONE wave (cross-lane exchange through LDS without a workgroup barrier): no instruction is
// emitted beyond the wait the fence implies; the compiler may not move LDS accesses across it
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int xcd_group(int b, int ng) {
    const int q = ng >> 3, r = ng & 7, x = b & 7;
    return x * q + (x < r ? x : r) + (b >> 3);
}

template <int R3, int NM>
__global__ __launch_bounds__(T) void k_fresnel_lines(LineArgs a) {
    constexpr int M = 576 * R3, LINES = TOT / M, S1 = M / RAD, MP = M + M / 32;
    constexpr int SLAB = 16, NSLABS = TOT / SLAB;       // 16 contiguous points per slab in the middle stage
    constexpr int WSLABS = 64 * RAD / SLAB;              // slabs inside the 1536 points one wave owns between barriers
    constexpr int NSLAB = (WSLABS + 63) / 64;            // slab rounds per lane (the last one is partly idle)
    constexpr int BPT = TOT / RAD / T;                   // radix-24 butterflies per thread per stage
    static_assert(R3 <= 16 && SLAB % R3 == 0 && TOT == RAD * T && 64 % R3 == 0 && NSLABS == (T / 64) * WSLABS,
                  "unsupported geometry");
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const int ngroups = (a.nlines + LINES - 1) / LINES;
    const int l0 = xcd_group(blockIdx.x, ngroups) * LINES;
    const int N = a.N, mg = a.margin;

    PSX_STAMP(0);
    // ---- 1. samples -> LDS.  s[j] = x_per[j - (N+mg-1)], j in [0, L);  zeros in [L, M).
    for (int idx = tid; idx < LINES * (M - a.L); idx += T) {
        const int line = idx / (M - a.L), j = a.L + idx % (M - a.L);
        lds[line * MP + phys(j)] = make_float2(0.f, 0.f);
    }
    // Every thread owns the samples i0 + STEP*k (k < 12) of ONE line: their global loads are issued back to back so that
    // one memory latency covers all of them.  STEP is a multiple of 32 (R3 >= 4), so the padded LDS index of sample k is
    // the index of sample 0 plus a compile-time offset: two address registers serve all 24 stores.
    constexpr int STEP = T / LINES, NLD = TOT / (2 * T), PSTEP = STEP + STEP / 32;
    static_assert(T % LINES == 0 && NLD * STEP >= (576 * R3 + 1) / 2, "sample ownership does not cover the longest line");
    constexpr bool AFFL = (STEP % 32 == 0);
    {
        const int line = tid % LINES, i0 = tid / LINES;
        const bool line_ok = l0 + line < a.nlines;
        const int64_t pix0 = (int64_t)i0 * a.in_stride + (l0 + line), pstep = (int64_t)STEP * a.in_stride;
        float2 xs[NLD], xm = make_float2(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            xs[k] = make_float2(0.f, 0.f);
            if (line_ok && i0 + STEP * k < N) xs[k] = source_wave<NM>(a.src, a.amp, a.m, pix0 + pstep * k);
        }
        // mirror duty (np.pad 'reflect', EXP:237): thread t < LINES*2*mg re-reads one of the 2*mg samples next to an edge
        const int nmir = LINES * 2 * mg;
        int im = -1, jm = 0;
        if (tid < nmir) {
            const int r = tid / LINES;
            im = r < mg ? r + 1 : N - 1 - 2 * mg + r;                   // 1..mg   |   N-1-mg..N-2
            jm = r < mg ? N + 2 * mg - 1 - im : 2 * N - 3 - im;         // left mirror | right mirror one period earlier
            if (line_ok) xm = source_wave<NM>(a.src, a.amp, a.m, (int64_t)im * a.in_stride + (l0 + line));
        }
        float2 *base = lds + line * MP;
        const int ja = i0 + N + 2 * mg - 1, jb = i0 - 1;                 // first period | one period earlier (i >= 1)
        const int oa = phys(ja), ob = phys(jb);   // jb = -1 (sample 0 has no earlier image) -> -2: affine like the rest, unused at k = 0
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            if (i0 + STEP * k < N) {
                if (AFFL) {
                    base[oa + k * PSTEP] = xs[k];
                    if (k > 0 || jb >= 0) base[ob + k * PSTEP] = xs[k];
                } else {
                    base[phys(ja + STEP * k)] = xs[k];
                    if (k > 0 || jb >= 0) base[phys(jb + STEP * k)] = xs[k];
                }
            }
        }
        if (im >= 0) base[phys(jm)] = xm;
        for (int t = tid + T; t < nmir; t += T) {                        // margins beyond T/(2*LINES): rare
            const int ln = t % LINES, r = t / LINES;
            const int i2 = r < mg ? r + 1 : N - 1 - 2 * mg + r, j2 = r < mg ? N + 2 * mg - 1 - i2 : 2 * N - 3 - i2;
            float2 x2 = make_float2(0.f, 0.f);
            if (l0 + ln < a.nlines) x2 = source_wave<NM>(a.src, a.amp, a.m, (int64_t)i2 * a.in_stride + (l0 + ln));
            lds[ln * MP + phys(j2)] = x2;
        }
    }
    PSX_STAMP(1);
    __syncthreads();
    PSX_STAMP(2);

    // LDS index of butterfly element j: with S1 a multiple of 32 (and R3 | 32) the pad term of phys() is affine in j, so
    // every element is one base register + a compile-time offset (ds_read/ds_write immediate offsets)
    constexpr bool AFF = (S1 % 32 == 0);
    auto idxA = [&](int n, int j) __attribute__((always_inline)) {
        return AFF ? phys(n) + j * (S1 + S1 / 32) : phys(n + j * S1);
    };
    auto idxB = [&](int p0, int j) __attribute__((always_inline)) {      // p0 = q1*S1 + n,  n < R3
        return AFF ? phys(p0) + j * R3 + ((j * R3) >> 5) : phys(p0 + j * R3);
    };
    // the 24 twiddles of butterfly n: 12 x 16-byte loads from one base
    auto load_tw = [&](const float2 *tw, int n, float2(&w)[RAD]) __attribute__((always_inline)) {
        const float4 *t4 = reinterpret_cast<const float4 *>(tw + (size_t)n * RAD);
#pragma unroll
        for (int q = 0; q < RAD / 2; ++q) {
            const float4 x = t4[q];
            w[2 * q] = make_float2(x.x, x.y);
            w[2 * q + 1] = make_float2(x.z, x.w);
        }
    };

    // v[q] *= conj(tw[n][q])
    auto mul_tw_conj = [&](const float2 *tw, int n, float2(&v)[RAD]) __attribute__((always_inline)) {
        const float4 *t4 = reinterpret_cast<const float4 *>(tw + (size_t)n * RAD);
        constexpr int NB = 1, PER = RAD / 2 / NB;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float4 x[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) x[q] = t4[b * PER + q];
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int k = 2 * (b * PER + q);
                if (k > 0) v[k] = cmulc(v[k], make_float2(x[q].x, x[q].y));
                v[k + 1] = cmulc(v[k + 1], make_float2(x[q].z, x[q].w));
            }
        }
    };

    // ---- 2. forward stage A: radix 24 over stride S1, twiddle w_M^{n q}
#pragma unroll 1
    for (int u = 0; u < BPT; ++u) {
        const int e = tid + T * u, line = e / S1, n = e % S1;
        float2 *base = lds + line * MP;
        float2 v[RAD], w[RAD];
#pragma unroll
        for (int j = 0; j < RAD; ++j) v[j] = base[idxA(n, j)];
        Dft<RAD, false>::run(v);
        __builtin_amdgcn_sched_barrier(0);   // keep the twiddle loads below the butterfly: 48 fewer live VGPRs
        load_tw(a.twA, n, w);
#pragma unroll
        for (int q = 1; q < RAD; ++q) v[q] = cmul(v[q], w[q]);
#pragma unroll
        for (int q = 0; q < RAD; ++q) base[idxA(n, q)] = v[q];
    }
    PSX_STAMP(3);
    __syncthreads();
    PSX_STAMP(4);

    // ---- 3. forward stage B: radix 24 inside each block of S1, stride R3, twiddle w_S1^{n q}
#pragma unroll 1
    for (int u = 0; u < BPT; ++u) {
        const int e = tid + T * u, line = e / S1, rem = e % S1, q1 = rem / R3, n = rem % R3;
        float2 *base = lds + line * MP;
        const int p0 = q1 * S1 + n;
        float2 v[RAD], w[RAD];
#pragma unroll
        for (int j = 0; j < RAD; ++j) v[j] = base[idxB(p0, j)];
        Dft<RAD, false>::run(v);
        __builtin_amdgcn_sched_barrier(0);
        load_tw(a.twB, n, w);
#pragma unroll
        for (int q = 1; q < RAD; ++q) v[q] = cmul(v[q], w[q]);
#pragma unroll
        for (int q = 0; q < RAD; ++q) base[idxB(p0, q)] = v[q];
    }
    PSX_STAMP(5);
    // From here to the end of inverse stage B every wave works on LDS points that only IT touches: its 64 radix-24
    // butterflies of stage B cover 64/R3 whole blocks of S1 points = the 1536 consecutive points [1536 w, 1536 (w+1)),
    // and the middle stage below takes its slabs from the same range.  A wave's LDS operations execute in order, so no
    // workgroup barrier is needed -- the waves drift apart and overlap each other's LDS and VALU phases.
    wave_sync();
    PSX_STAMP(6);

    // ---- 4+5. middle stage, slab by slab: forward radix R3 on contiguous chunks, x FFT_M(h_d), inverse radix R3, back to
    // LDS.  Each thread rewrites exactly the slabs it read, so no barrier separates the two halves.
    for (int d = 0; d < a.n_dist; ++d) {
        if (d > 0) __syncthreads();   // the previous distance's inverse stage A has finished reading LDS
        const float2 *Hd = a.H[d];
#pragma unroll
        for (int r = 0; r < NSLAB; ++r) {
            // slab r of this lane inside the wave's own 96 slabs (the second round keeps 32 lanes busy)
            const int sw = (tid & 63) + 64 * r;
            if (sw >= WSLABS) break;
            const int s = (tid >> 6) * WSLABS + sw, line = s / (M / SLAB), p0 = (s % (M / SLAB)) * SLAB;
            float2 *base = lds + line * MP + phys(p0);      // p0 % 16 == 0: no pad slot inside a slab
            float4 hh[SLAB / 2];                            // kernel spectrum of this slab: issued before the LDS reads
            const float4 *h4 = reinterpret_cast<const float4 *>(Hd + p0);
#pragma unroll
            for (int j = 0; j < SLAB / 2; ++j) hh[j] = h4[j];
            float2 f[SLAB], g[SLAB];
#pragma unroll
            for (int j = 0; j < SLAB; ++j) f[j] = base[j];
#pragma unroll
            for (int c = 0; c < SLAB / R3; ++c) {
                float2 w[R3];
#pragma unroll
                for (int j = 0; j < R3; ++j) w[j] = f[c * R3 + j];
                Dft<R3, false>::run(w);
#pragma unroll
                for (int j = 0; j < R3; ++j) f[c * R3 + j] = w[j];
            }
#pragma unroll
            for (int j = 0; j < SLAB / 2; ++j) {
                g[2 * j] = cmul(f[2 * j], make_float2(hh[j].x, hh[j].y));
                g[2 * j + 1] = cmul(f[2 * j + 1], make_float2(hh[j].z, hh[j].w));
            }
#pragma unroll
            for (int c = 0; c < SLAB / R3; ++c) {
                float2 w[R3];
#pragma unroll
                for (int j = 0; j < R3; ++j) w[j] = g[c * R3 + j];
                Dft<R3, true>::run(w);
#pragma unroll
                for (int j = 0; j < R3; ++j) base[c * R3 + j] = w[j];
            }
        }
        PSX_STAMP(7);
        wave_sync();
        PSX_STAMP(8);

        // The inverse stages use the same twiddles as the forward ones; launder the pointers so that the compiler
        // reloads them (L2-resident) instead of keeping 92 VGPRs alive across the whole kernel and spilling.
        const float2 *twA_i = a.twA, *twB_i = a.twB;
        asm volatile("" : "+s"(twA_i), "+s"(twB_i));
        // ---- 6. inverse stage B: conjugate twiddle on the inputs, then the inverse radix-24 butterfly
#pragma unroll 1
        for (int u = 0; u < BPT; ++u) {
            const int e = tid + T * u, line = e / S1, rem = e % S1, q1 = rem / R3, n = rem % R3;
            float2 *base = lds + line * MP;
            const int p0 = q1 * S1 + n;
            float2 v[RAD];
#pragma unroll
            for (int q = 0; q < RAD; ++q) v[q] = base[idxB(p0, q)];
            mul_tw_conj(twB_i, n, v);
            __builtin_amdgcn_sched_barrier(0);
            Dft<RAD, true>::run(v);
#pragma unroll
            for (int j = 0; j < RAD; ++j) base[idxB(p0, j)] = v[j];
        }
        PSX_STAMP(9);
        __syncthreads();
        PSX_STAMP(10);

        // ---- 7. inverse stage A; the wanted outputs y[n + P - 1] leave for HBM straight from the registers
        const int jout = N + 2 * mg - 1;      // LDS position of output sample 0
        float2 *wo = a.wave_out[d];
        float *io = a.inten_out[d];
        const float sc = a.scale[d];
        const float2 gp = a.gph[d];
#pragma unroll 1
        for (int u = 0; u < BPT; ++u) {
            const int e = tid + T * u, line = e / S1, n = e % S1;
            const float2 *base = lds + line * MP;
            float2 v[RAD];
#pragma unroll
            for (int q = 0; q < RAD; ++q) v[q] = base[idxA(n, q)];
            mul_tw_conj(twA_i, n, v);
            __builtin_amdgcn_sched_barrier(0);
            Dft<RAD, true>::run(v);
            if (l0 + line < a.nlines) {
                const int64_t ob = (int64_t)(l0 + line) * a.out_ld;
#pragma unroll
                for (int j = 0; j < RAD; ++j) {
                    const int i = n + j * S1 - jout;
                    if (i >= 0 && i < N) {
                        if (wo) wo[ob + i] = cmul(v[j], gp);
                        if (io) {
                            const float I = sc * (v[j].x * v[j].x + v[j].y * v[j].y);
                            io[ob + i] = a.accumulate ? io[ob + i] + I : I;
                        }
                    }
                }
            }
        }
    }
    PSX_STAMP(11);
}

// z == 0 (EXP:233-234): out = psi, |psi|^2
template <int NM>
__global__ __launch_bounds__(256) void k_source_out(const float2 *__restrict__ src, float amp, Mats m,
                                                    float2 *__restrict__ wave_out, float *__restrict__ inten_out,
                                                    float scale, int accumulate, int64_t n) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const float2 v = source_wave<NM>(src, amp, m, p);
        if (wave_out) wave_out[p] = v;
        if (inten_out) {
            const float I = scale * (v.x * v.x + v.y * v.y);
            inten_out[p] = accumulate ? inten_out[p] + I : I;
        }
    }
}

// ---- float64 construction of the kernel spectrum FFT_M(IDFT_P(chirp)) ------------------------------------------------
__global__ void k_cis_table(double2 *t, int n) {   // t[r] = exp(+2 pi i r / n)
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double s, c;
    sincospi(2.0 * (double)r / (double)n, &s, &c);
    t[r] = make_double2(c, s);
}

__global__ void k_kern_H(double2 *H, int P, double a, double du) {   // EXP:243-250, FFT order
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int f = (i < (P + 1) / 2) ? i : i - P;
    const double u = (double)f * du;
    const double ph = -a * u * u;
    const double r = ph - PSX_TWO_PI * rint(ph * PSX_INV_TWO_PI);
    double s, c;
    sincos(r, &s, &c);
    H[i] = make_double2(c, s);
}

__global__ void k_kern_h(const double2 *H, const double2 *twP, double2 *h, int P) {   // h = IDFT_P(H)
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= P) return;
    double re = 0.0, im = 0.0;
    int r = 0;
    for (int k = 0; k < P; ++k) {
        const double2 x = H[k], w = twP[r];
        re += x.x * w.x - x.y * w.y;
        im += x.x * w.y + x.y * w.x;
        r += d;
        if (r >= P) r -= P;
    }
    h[d] = make_double2(re / P, im / P);
}

__global__ void k_kern_Hhat(const double2 *h, const double2 *twM, double2 *Hh, int P, int M) {   // FFT_M(h zero-padded)
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    double re = 0.0, im = 0.0;
    int r = 0;
    for (int d = 0; d < P; ++d) {
        const double2 x = h[d], w = twM[r];       // conj(w): exp(-2 pi i k d / M)
        re += x.x * w.x + x.y * w.y;
        im += x.y * w.x - x.x * w.y;
        r += k;
        if (r >= M) r -= M;
    }
    Hh[k] = make_double2(re, im);
}

// position p = q1*S1 + q2*R3 + q3 of the in-place DIF output holds frequency k = q1 + 24*q2 + 576*q3
__global__ void k_kern_perm(const double2 *Hh, float2 *out, int M, int R3) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= M) return;
    const int S1 = M / RAD;
    const int q1 = p / S1, q2 = (p % S1) / R3, q3 = p % R3;
    const double2 v = Hh[q1 + RAD * q2 + RAD * RAD * q3];
    out[p] = make_float2((float)(v.x / M), (float)(v.y / M));
}

__global__ void k_stage_twiddles(float2 *twA, float2 *twB, int M, int R3) {
    const int S1 = M / RAD;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    // layout [n][24]: the 24 twiddles of one butterfly are contiguous (192 B), so a thread fetches them with 16-byte
    // loads at immediate offsets from ONE address register
    if (idx < RAD * S1) {
        const int n = idx / RAD, q = idx % RAD;
        double s, c;
        sincospi(-2.0 * (double)(((long long)n * q) % M) / (double)M, &s, &c);
        twA[idx] = make_float2((float)c, (float)s);
    }
    if (idx < RAD * R3) {
        const int n = idx / RAD, q = idx % RAD;
        double s, c;
        sincospi(-2.0 * (double)((n * q) % S1) / (double)S1, &s, &c);
        twB[idx] = make_float2((float)c, (float)s);
    }
}

int pick_r3(int N, int margin) {
    const int need = 2 * N + 2 * margin - 1;   // L = N + P - 1
    for (int r3 : {2, 4, 8, 16})
        if (576 * r3 >= need) return r3;
    return 0;
}

}  // namespace

namespace psx {

struct AxisTables {
    int N = 0, R3 = 0, M = 0;
    float2 *twA = nullptr, *twB = nullptr;
};

struct KernEntry {
    double a, du;
    int N, M;
    float2 *H;
    unsigned long long stamp;
};

struct LdsEngine {
    AxisTables ax[2];            // [0]: lines along axis 0 (length Nx), [1]: along axis 1 (length Ny)
    float2 *inter = nullptr;     // [max_dist][Ny][Nx] transposed intermediates
    float2 *pre = nullptr;       // [Nx][Ny] pre-transmitted wave when nmat exceeds the fused variants
    double2 *wH = nullptr, *wh = nullptr, *wHh = nullptr, *twP = nullptr, *twM = nullptr;
    int twP_n = 0, twM_n = 0;
    std::vector<KernEntry> cache;
    unsigned long long clock = 0;
};

bool lds_engine_supported(int Nx, int Ny, int margin) {
    return margin >= 0 && margin <= Nx - 1 && margin <= Ny - 1 && pick_r3(Nx, margin) && pick_r3(Ny, margin);
}

static int make_axis(AxisTables &t, int N, int margin, size_t &bytes) {
    t.N = N;
    t.R3 = pick_r3(N, margin);
    t.M = 576 * t.R3;
    const int S1 = t.M / RAD;
    PSX_HIP(hipMalloc((void **)&t.twA, sizeof(float2) * RAD * S1));
    PSX_HIP(hipMalloc((void **)&t.twB, sizeof(float2) * RAD * t.R3));
    bytes += sizeof(float2) * RAD * (S1 + t.R3);
    k_stage_twiddles<<<(int)cdiv(RAD * S1, 256), 256>>>(t.twA, t.twB, t.M, t.R3);
    return launch_check("k_stage_twiddles");
}

int lds_engine_create(psx_fresnel_plan *p) {
    LdsEngine *e = new LdsEngine();
    p->lds = e;
    if (int rc = make_axis(e->ax[0], p->Nx, p->margin, p->bytes)) return rc;
    if (int rc = make_axis(e->ax[1], p->Ny, p->margin, p->bytes)) return rc;
    const size_t img = sizeof(float2) * (size_t)p->Nx * (size_t)p->Ny;
    PSX_HIP(hipMalloc((void **)&e->inter, img * p->max_dist));
    p->bytes += img * p->max_dist;
    const int Pm = (p->Nx > p->Ny ? p->Nx : p->Ny) + 2 * p->margin;
    const int Mm = e->ax[0].M > e->ax[1].M ? e->ax[0].M : e->ax[1].M;
    PSX_HIP(hipMalloc((void **)&e->wH, sizeof(double2) * Pm));
    PSX_HIP(hipMalloc((void **)&e->wh, sizeof(double2) * Pm));
    PSX_HIP(hipMalloc((void **)&e->twP, sizeof(double2) * Pm));
    PSX_HIP(hipMalloc((void **)&e->wHh, sizeof(double2) * Mm));
    PSX_HIP(hipMalloc((void **)&e->twM, sizeof(double2) * Mm));
    p->bytes += sizeof(double2) * (3 * (size_t)Pm + 2 * (size_t)Mm);
    PSX_HIP(hipDeviceSynchronize());
    return 0;
}

void lds_engine_destroy(psx_fresnel_plan *p) {
    LdsEngine *e = p->lds;
    if (!e) return;
    for (auto &t : e->ax) {
        (void)hipFree(t.twA);
        (void)hipFree(t.twB);
    }
    for (auto &k : e->cache) (void)hipFree(k.H);
    (void)hipFree(e->inter);
    (void)hipFree(e->pre);
    (void)hipFree(e->wH);
    (void)hipFree(e->wh);
    (void)hipFree(e->wHh);
    (void)hipFree(e->twP);
    (void)hipFree(e->twM);
    delete e;
    p->lds = nullptr;
}

// Kernel spectrum of one (distance, axis): cached, because it depends on scalars only (the reference rebuilds its
// chirp on every call, EXP:243-248).  Built on `st` in float64; the cache is per plan, so stream order is enough.
static int kernel_spectrum(psx_fresnel_plan *p, const AxisTables &t, double a, double du, hipStream_t st,
                           const float2 **out) {
    LdsEngine *e = p->lds;
    for (auto &k : e->cache)
        if (k.a == a && k.du == du && k.N == t.N && k.M == t.M) {
            k.stamp = ++e->clock;
            *out = k.H;
            return 0;
        }
    KernEntry k{a, du, t.N, t.M, nullptr, ++e->clock};
    if (e->cache.size() >= 64) {   // evict the least recently used table
        size_t lru = 0;
        for (size_t i = 1; i < e->cache.size(); ++i)
            if (e->cache[i].stamp < e->cache[lru].stamp) lru = i;
        k.H = e->cache[lru].H;
        if (e->cache[lru].M != t.M) {
            PSX_HIP(hipStreamSynchronize(st));
            (void)hipFree(k.H);
            k.H = nullptr;
        }
        e->cache.erase(e->cache.begin() + lru);
    }
    if (!k.H) PSX_HIP(hipMalloc((void **)&k.H, sizeof(float2) * t.M));
    const int P = t.N + 2 * p->margin;
    if (e->twP_n != P) {
        PSX_TIMED("k_cis_table", st, k_cis_table<<<(int)cdiv(P, 256), 256, 0, st>>>(e->twP, P));
        e->twP_n = P;
    }
    if (e->twM_n != t.M) {
        PSX_TIMED("k_cis_table", st, k_cis_table<<<(int)cdiv(t.M, 256), 256, 0, st>>>(e->twM, t.M));
        e->twM_n = t.M;
    }
    PSX_TIMED("k_kern_H", st, k_kern_H<<<(int)cdiv(P, 256), 256, 0, st>>>(e->wH, P, a, du));
    PSX_TIMED("k_kern_h", st, k_kern_h<<<(int)cdiv(P, 64), 64, 0, st>>>(e->wH, e->twP, e->wh, P));
    PSX_TIMED("k_kern_Hhat", st, k_kern_Hhat<<<(int)cdiv(t.M, 64), 64, 0, st>>>(e->wh, e->twM, e->wHh, P, t.M));
    PSX_TIMED("k_kern_perm", st, k_kern_perm<<<(int)cdiv(t.M, 256), 256, 0, st>>>(e->wHh, k.H, t.M, t.R3));
    if (int rc = launch_check("kernel spectrum")) return rc;
    e->cache.push_back(k);
    *out = k.H;
    return 0;
}

template <int R3, int NM>
static int launch_lines(const LineArgs &la, hipStream_t st, const char *name) {
    constexpr int M = 576 * R3, LINES = TOT / M;
    constexpr size_t lds_bytes = sizeof(float2) * (size_t)LINES * (M + M / 32);
    static bool attr_set = false;
    if (!attr_set) {
        PSX_HIP(hipFuncSetAttribute((const void *)k_fresnel_lines<R3, NM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_bytes));
        attr_set = true;
    }
    const int ngroups = (la.nlines + LINES - 1) / LINES;
    PSX_TIMED(name, st, k_fresnel_lines<R3, NM><<<ngroups, T, lds_bytes, st>>>(la));
    return launch_check(name);
}

template <int NM>
static int launch_lines_r3(int R3, const LineArgs &la, hipStream_t st, const char *name) {
    switch (R3) {
        case 2: return launch_lines<2, NM>(la, st, name);
        case 4: return launch_lines<4, NM>(la, st, name);
        case 8: return launch_lines<8, NM>(la, st, name);
        case 16: return launch_lines<16, NM>(la, st, name);
    }
    return fail(PSX_E_UNSUPPORTED, "LDS engine: unsupported line length");
}

int lds_engine_propagate(psx_fresnel_plan *p, const PropArgs &a) {
    LdsEngine *e = p->lds;
    hipStream_t st = a.stream;
    const int64_t npix = (int64_t)p->Nx * p->Ny;
    const float2 *src = a.wave_in;
    float amp = a.amp;
    Mats m = a.m;
    if (m.n > 3) {   // only 0..3 materials are fused into the line kernel: pre-transmit the rest in one pass
        if (!e->pre) {
            PSX_HIP(hipMalloc((void **)&e->pre, sizeof(float2) * npix));
            p->bytes += sizeof(float2) * npix;
        }
        PSX_DISPATCH_NMAT(m.n, PSX_TIMED("k_source_out", st, k_source_out<NM><<<ew_grid(npix, 256), 256, 0, st>>>(
                                                                 src, amp, m, e->pre, nullptr, 1.f, 0, npix)));
        src = e->pre;
        amp = 1.f;
        Mats none;
        if (int rc = pack_mats(none, nullptr, nullptr, nullptr, 0)) return rc;
        m = none;
    }
    // z == 0 distances return the input field (EXP:233-234)
    int nz[PSX_MAX_DIST], nnz = 0;
    for (int d = 0; d < a.n_dist; ++d) {
        if (a.a[d] != 0.0) {
            nz[nnz++] = d;
            continue;
        }
        PSX_DISPATCH_NMAT(m.n, PSX_TIMED("k_source_out", st, k_source_out<NM><<<ew_grid(npix, 256), 256, 0, st>>>(
                                                                 src, amp, m, a.wave_out ? a.wave_out[d] : nullptr,
                                                                 a.inten_out ? a.inten_out[d] : nullptr,
                                                                 a.inten_scale ? a.inten_scale[d] : 1.f, a.accumulate,
                                                                 npix)));
    }
    if (nnz == 0) return launch_check("k_source_out");

    // ---- pass 1: lines along axis 0 (columns), output transposed.  One launch per distance: keeping the forward spectrum
    // in registers across distances needs 64 more VGPRs than the 168 a 12-wave workgroup has, and the spills of that
    // variant cost 3.9 GB of HBM traffic per 4-distance launch (rocprof) -- no faster than re-running the forward stages.
    static const bool stamp_pass1 = getenv("PSX_STAMP_PASS1") != nullptr;   // diagnostics only
    for (int i = 0; i < nnz; ++i) {
        LineArgs la;
        la.src = src; la.amp = amp; la.m = m;
        la.N = p->Nx; la.nlines = p->Ny; la.margin = p->margin; la.P = p->Px; la.L = p->Nx + p->Px - 1;
        la.in_stride = p->Ny; la.out_ld = p->Nx;
        la.twA = e->ax[0].twA; la.twB = e->ax[0].twB;
        la.n_dist = 1; la.accumulate = 0; la.stamps = stamp_pass1 ? g_stamps : nullptr;
        if (int rc = kernel_spectrum(p, e->ax[0], a.a[nz[i]], a.du_x, st, &la.H[0])) return rc;
        la.wave_out[0] = e->inter + (size_t)i * npix;
        la.inten_out[0] = nullptr;
        la.scale[0] = 1.f;
        la.gph[0] = make_float2(1.f, 0.f);
        int rc = 0;
        switch (m.n) {
            case 0: rc = launch_lines_r3<0>(e->ax[0].R3, la, st, "k_fresnel_cols"); break;
            case 1: rc = launch_lines_r3<1>(e->ax[0].R3, la, st, "k_fresnel_cols"); break;
            case 2: rc = launch_lines_r3<2>(e->ax[0].R3, la, st, "k_fresnel_cols"); break;
            default: rc = launch_lines_r3<3>(e->ax[0].R3, la, st, "k_fresnel_cols"); break;
        }
        if (rc) return rc;
    }

    // ---- pass 2: lines along axis 1 of the original image (= axis 0 of the transposed intermediate)
    Mats none;
    if (int rc2 = pack_mats(none, nullptr, nullptr, nullptr, 0)) return rc2;
    for (int i = 0; i < nnz; ++i) {
        const int d = nz[i];
        LineArgs lb;
        lb.src = e->inter + (size_t)i * npix; lb.amp = 1.f; lb.m = none;
        lb.N = p->Ny; lb.nlines = p->Nx; lb.margin = p->margin; lb.P = p->Py; lb.L = p->Ny + p->Py - 1;
        lb.in_stride = p->Nx; lb.out_ld = p->Ny;
        lb.twA = e->ax[1].twA; lb.twB = e->ax[1].twB;
        lb.n_dist = 1; lb.accumulate = a.accumulate; lb.stamps = stamp_pass1 ? nullptr : g_stamps;
        if (int rc2 = kernel_spectrum(p, e->ax[1], a.a[d], a.du_y, st, &lb.H[0])) return rc2;
        lb.wave_out[0] = a.wave_out ? a.wave_out[d] : nullptr;
        lb.inten_out[0] = a.inten_out ? a.inten_out[d] : nullptr;
        lb.scale[0] = a.inten_scale ? a.inten_scale[d] : 1.f;
        const double g = a.gphase ? a.gphase[d] : 0.0;
        lb.gph[0] = make_float2((float)std::cos(g), (float)std::sin(g));   // exact reduction of ~1e11 rad (EXP:250)
        if (int rc2 = launch_lines_r3<0>(e->ax[1].R3, lb, st, "k_fresnel_rows")) return rc2;
    }
    return 0;
}

}  // namespace psx

