// detect.hip -- detector model: blur, resampling, PSF, shot noise (K14-K19).
//
// Replaces Detector.detection (Detector.py:79-119), resize (:185-198) and create_gaussian_shape (:201-220).
// reflect-pad(15*ov) -> Gaussian source blur (zero-extended 'same' convolution) -> ov x ov block SUM -> Gaussian PSF
// -> crop(15) is a chain of linear, separable operators, so per axis it collapses to ONE banded matrix
// C [n x N] (n detector pixels, N study pixels).  The plan composes C_x and C_y on the host in float64 by pushing unit
// vectors back through the chain (so reflect padding, zero extension at the padded border, banker's rounding of the
// kernel support and the bin-SUM are reproduced exactly), and the image is formed as  out = C_x * img * C_y^T  in two
// passes: along the contiguous axis first (shrinks the data by ov), then along axis 0.
#include <algorithm>
#include <vector>

#include "common.hpp"

using namespace psx;

struct psx_detector_plan {
    int Nx, Ny, ov, nx, ny, margin;
    int Wx, Wy;          // band widths
    int *sx = nullptr;   // [nx] first study row of output row r
    int *sy = nullptr;   // [ny]
    float *wx = nullptr; // [nx][Wx]
    float *wy = nullptr; // [ny][Wy]
    float *tmp = nullptr;  // [Nx][ny]
    size_t bytes = 0;
};

namespace {

// Python 3 round(): half to even (Detector.py:212 `round(sigma*3)`)
int py_round(double x) { return (int)std::nearbyint(x); }

// normalised 1-D factor of create_gaussian_shape: g2d = outer(g,g)/sum == outer(g/sum g, g/sum g)
std::vector<double> gauss1d(double sigma, int &radius) {
    const int dim = py_round(sigma * 3.0) * 2 + 1;   // DET:212
    radius = dim / 2;
    std::vector<double> g(dim);
    double s = 0.0;
    for (int i = 0; i < dim; ++i) {
        const double q = (double)i - std::floor(dim / 2.0);
        g[i] = std::exp(-(q * q) / 2.0 / (sigma * sigma));   // DET:218
        s += g[i];
    }
    for (double &v : g) v /= s;
    return g;
}

// Composite operator of one axis: row r of C as a window [lo, hi] over the N study pixels, then a common band width.
void compose_axis(int N, int ov, int n, int margin, double sigma_src, double sigma_psf, std::vector<int> &start,
                  std::vector<float> &weights, int &W) {
    const int Npad = N + 2 * margin * ov;   // DET:93
    const int npad = n + 2 * margin;        // DET:103
    const int s = Npad / npad;              // DET:192 (the axis-0 factor is used on both axes; equal for ov grids)
    int r1 = 0, r2 = 0;
    std::vector<double> g1, g2;
    if (sigma_src != 0.0) g1 = gauss1d(sigma_src, r1);
    if (sigma_psf != 0.0) g2 = gauss1d(sigma_psf, r2);
    std::vector<std::vector<double>> rows(n);
    std::vector<int> lo(n, 0);
    std::vector<double> wb(npad, 0.0), wp(Npad, 0.0), wq(Npad, 0.0), acc(N, 0.0);
    for (int r = 0; r < n; ++r) {
        // crop^T: unit at binned index r+margin (DET:118); PSF^T (DET:106-108, zero-extended 'same')
        const int p = r + margin;
        const int qa = std::max(0, p - r2), qb = std::min(npad - 1, p + r2);
        for (int q = qa; q <= qb; ++q) wb[q] = g2.empty() ? 1.0 : g2[r2 + (p - q)];
        // bin^T: binned q gathers padded study pixels [q*s, q*s+s) (DET:194-196; numpy slices clip at the array end)
        const int ka = qa * s, kb = std::min(qb * s + s, Npad) - 1;
        for (int q = qa; q <= qb; ++q)
            for (int k = q * s; k < std::min(q * s + s, Npad); ++k) wp[k] = wb[q];
        // source blur^T (DET:96-99, zero-extended 'same')
        const int la = std::max(0, ka - r1), lb = std::min(Npad - 1, kb + r1);
        for (int l = la; l <= lb; ++l) {
            if (g1.empty()) {
                wq[l] = wp[l];
            } else {
                double v = 0.0;
                for (int t = -r1; t <= r1; ++t) {
                    const int k = l + t;
                    if (k >= ka && k <= kb) v += wp[k] * g1[r1 + t];
                }
                wq[l] = v;
            }
        }
        // reflect pad^T (DET:93)
        int ulo = N, uhi = -1;
        for (int l = la; l <= lb; ++l) {
            // np.pad 'reflect' (pads wider than N-1 keep mirroring: period 2(N-1))
            const int per = 2 * (N - 1);
            int u = (l - margin * ov) % per;
            if (u < 0) u += per;
            if (u >= N) u = per - u;
            acc[u] += wq[l];
            ulo = std::min(ulo, u);
            uhi = std::max(uhi, u);
        }
        lo[r] = ulo;
        rows[r].assign(acc.begin() + ulo, acc.begin() + uhi + 1);
        for (int u = ulo; u <= uhi; ++u) acc[u] = 0.0;
        for (int q = qa; q <= qb; ++q) wb[q] = 0.0;
        for (int k = ka; k <= kb; ++k) wp[k] = 0.0;
        for (int l = la; l <= lb; ++l) wq[l] = 0.0;
    }
    W = 1;
    for (int r = 0; r < n; ++r) W = std::max(W, (int)rows[r].size());
    start.assign(n, 0);
    weights.assign((size_t)n * W, 0.f);
    for (int r = 0; r < n; ++r) {
        const int st = std::max(0, std::min(lo[r], N - W));
        start[r] = st;
        for (size_t w = 0; w < rows[r].size(); ++w) weights[(size_t)r * W + (lo[r] - st) + w] = (float)rows[r][w];
    }
}

// pass 1: tmp[i][c] = sum_w wy[c][w] * img[i][sy[c]+w]      (contiguous axis; ny outputs per row)
__global__ __launch_bounds__(256) void k_detect_cols(const float *__restrict__ img, float *__restrict__ tmp,
                                                     const int *__restrict__ sy, const float *__restrict__ wy, int Nx,
                                                     int Ny, int ny, int Wy) {
    const int64_t n = (int64_t)Nx * ny;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(q / ny), c = (int)(q - (int64_t)i * ny);
        const float *row = img + (int64_t)i * Ny + sy[c];
        const float *w = wy + (int64_t)c * Wy;
        const int lim = min(Wy, Ny - sy[c]);
        float acc = 0.f;
        for (int k = 0; k < lim; ++k) acc = fmaf(w[k], row[k], acc);
        tmp[q] = acc;
    }
}

// pass 2: out[r][c] = sum_w wx[r][w] * tmp[sx[r]+w][c]
__global__ __launch_bounds__(256) void k_detect_rows(const float *__restrict__ tmp, float *__restrict__ out,
                                                     const int *__restrict__ sx, const float *__restrict__ wx, int Nx,
                                                     int nx, int ny, int Wx) {
    const int64_t n = (int64_t)nx * ny;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(q / ny), c = (int)(q - (int64_t)r * ny);
        const float *col = tmp + (int64_t)sx[r] * ny + c;
        const float *w = wx + (int64_t)r * Wx;
        const int lim = min(Wx, Nx - sx[r]);
        float acc = 0.f;
        for (int k = 0; k < lim; ++k) acc = fmaf(w[k], col[(int64_t)k * ny], acc);
        out[q] = acc;
    }
}

__global__ __launch_bounds__(256) void k_resize(const float *__restrict__ img, int Nx, int Ny, float *__restrict__ out,
                                                int sx, int sy, int s) {
    const int64_t n = (int64_t)sx * sy;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int x0 = (int)(q / sy), y0 = (int)(q - (int64_t)x0 * sy);
        const int xb = min(x0 * s + s, Nx), yb = min(y0 * s + s, Ny);
        float acc = 0.f;
        for (int x = x0 * s; x < xb; ++x)
            for (int y = y0 * s; y < yb; ++y) acc += img[(int64_t)x * Ny + y];
        out[q] = acc;
    }
}

// ---- Philox4x32-10 counter-based generator -----------------------------------------------------------------------
struct Philox {
    uint32_t c[4], k[2];
    __device__ void round_() {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
    }
    __device__ void next(uint64_t ctr, uint32_t sub, uint64_t seed, float u[4]) {
        c[0] = (uint32_t)ctr; c[1] = (uint32_t)(ctr >> 32); c[2] = sub; c[3] = 0x5058u;
        k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
        for (int r = 0; r < 10; ++r) {
            round_();
            k[0] += 0x9E3779B9u;
            k[1] += 0xBB67AE85u;
        }
        for (int i = 0; i < 4; ++i) u[i] = ((float)(c[i] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
    }
};

// Poisson draw: product-of-uniforms for lam < 10, Hormann's PTRS transformed rejection otherwise (both exact).
__global__ __launch_bounds__(256) void k_poisson(const float *__restrict__ lam, float *__restrict__ out, int64_t n,
                                                 uint64_t seed) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const float L = lam[p];
        Philox g;
        float u[4];
        float res = 0.f;
        if (!(L > 0.f)) {
            res = 0.f;
        } else if (L < 10.f) {
            const float lim = expf(-L);
            float prod = 1.f;
            int k = 0;
            uint32_t sub = 0;
            bool done = false;
            while (!done && sub < 64) {
                g.next((uint64_t)p, sub++, seed, u);
                for (int i = 0; i < 4 && !done; ++i) {
                    prod *= u[i];
                    if (prod <= lim) done = true; else ++k;
                }
            }
            res = (float)k;
        } else {
            const float slam = sqrtf(L);
            const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
            const float invalpha = 1.1239f + 1.1328f / (b - 3.4f), vr = 0.9277f - 3.6224f / (b - 2.f);
            uint32_t sub = 0;
            res = floorf(L);
            while (sub < 64) {
                g.next((uint64_t)p, sub++, seed, u);
                bool acc = false;
                for (int h = 0; h < 2 && !acc; ++h) {
                    const float U = u[2 * h] - 0.5f, V = u[2 * h + 1];
                    const float us = 0.5f - fabsf(U);
                    const float k = floorf((2.f * a / us + b) * U + L + 0.43f);
                    if (us >= 0.07f && V <= vr) { res = k; acc = true; break; }
                    if (k < 0.f || (us < 0.013f && V > us)) continue;
                    if (log((double)V) + log((double)invalpha) - log((double)a / ((double)us * us) + b) <=
                        -(double)L + (double)k * log((double)L) - lgamma((double)k + 1.0)) {
                        res = k;
                        acc = true;
                    }
                }
                if (acc) break;
            }
        }
        out[p] = res;
    }
}

}  // namespace

extern "C" {

int psx_detector_plan_create(int Nx, int Ny, int ov, int nx, int ny, int margin, double sigma_src, double sigma_psf,
                             psx_detector_plan **plan) {
    PSX_REQUIRE(plan != nullptr, "psx_detector_plan_create: null plan pointer");
    *plan = nullptr;
    PSX_REQUIRE(ov >= 1 && nx >= 1 && ny >= 1 && margin >= 0, "psx_detector_plan_create: bad ov/nx/ny/margin");
    PSX_REQUIRE(Nx >= 2 && Ny >= 2, "psx_detector_plan_create: study grid %dx%d too small", Nx, Ny);
    PSX_REQUIRE((Nx + 2 * margin * ov) / (nx + 2 * margin) >= 1, "psx_detector_plan_create: detector larger than the study grid");
    PSX_REQUIRE(sigma_src >= 0.0 && sigma_psf >= 0.0 && std::isfinite(sigma_src) && std::isfinite(sigma_psf),
                "psx_detector_plan_create: negative or non-finite sigma");
    // Detector.resize uses the axis-0 factor on both axes (DET:192); require the grids to agree with it
    PSX_REQUIRE((Nx + 2 * margin * ov) / (nx + 2 * margin) == (Ny + 2 * margin * ov) / (ny + 2 * margin),
                "psx_detector_plan_create: different resampling factors on the two axes");
    psx_detector_plan *p = new psx_detector_plan();
    p->Nx = Nx; p->Ny = Ny; p->ov = ov; p->nx = nx; p->ny = ny; p->margin = margin;
    std::vector<int> sx, sy;
    std::vector<float> wx, wy;
    compose_axis(Nx, ov, nx, margin, sigma_src, sigma_psf, sx, wx, p->Wx);
    compose_axis(Ny, ov, ny, margin, sigma_src, sigma_psf, sy, wy, p->Wy);
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        PSX_HIP(hipMalloc(dst, bytes));
        PSX_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        p->bytes += bytes;
        return 0;
    };
    int rc = 0;
    if (!rc) rc = up((void **)&p->sx, sx.data(), sizeof(int) * sx.size());
    if (!rc) rc = up((void **)&p->sy, sy.data(), sizeof(int) * sy.size());
    if (!rc) rc = up((void **)&p->wx, wx.data(), sizeof(float) * wx.size());
    if (!rc) rc = up((void **)&p->wy, wy.data(), sizeof(float) * wy.size());
    if (!rc) {
        hipError_t e = hipMalloc((void **)&p->tmp, sizeof(float) * (size_t)Nx * (size_t)ny);
        if (e != hipSuccess) rc = fail((int)e, "psx_detector_plan_create: hipMalloc(tmp) failed: %s", hipGetErrorString(e));
        p->bytes += sizeof(float) * (size_t)Nx * (size_t)ny;
    }
    if (rc) {
        psx_detector_plan_destroy(p);
        return rc;
    }
    *plan = p;
    return 0;
}

int psx_detector_operator_host(int N, int ov, int n, int margin, double sigma_src, double sigma_psf, int *start,
                               float *weights, int wcap, int *W_out) {
    PSX_REQUIRE(start && weights && W_out, "psx_detector_operator_host: null pointer");
    PSX_REQUIRE(ov >= 1 && n >= 1 && margin >= 0 && N >= 2, "psx_detector_operator_host: bad geometry");
    PSX_REQUIRE((N + 2 * margin * ov) / (n + 2 * margin) >= 1, "psx_detector_operator_host: detector larger than the study grid");
    std::vector<int> st;
    std::vector<float> w;
    int W = 0;
    compose_axis(N, ov, n, margin, sigma_src, sigma_psf, st, w, W);
    *W_out = W;
    PSX_REQUIRE(W <= wcap, "psx_detector_operator_host: band width %d exceeds the caller's capacity %d", W, wcap);
    std::copy(st.begin(), st.end(), start);
    for (int r = 0; r < n; ++r) std::copy(w.begin() + (size_t)r * W, w.begin() + (size_t)(r + 1) * W, weights + (size_t)r * wcap);
    return 0;
}

int psx_detector_plan_destroy(psx_detector_plan *p) {
    if (!p) return 0;
    (void)hipFree(p->sx);
    (void)hipFree(p->sy);
    (void)hipFree(p->wx);
    (void)hipFree(p->wy);
    (void)hipFree(p->tmp);
    delete p;
    return 0;
}

int psx_detect_f32(psx_detector_plan *p, const float *img, float *out, void *stream) {
    PSX_REQUIRE(p && img && out, "psx_detect_f32: null pointer");
    hipStream_t st = (hipStream_t)stream;
    PSX_TIMED("k_detect_cols", st, k_detect_cols<<<ew_grid((int64_t)p->Nx * p->ny, 256), 256, 0, st>>>(img, p->tmp, p->sy, p->wy, p->Nx, p->Ny, p->ny,
                                                                        p->Wy));
    if (int rc = launch_check("k_detect_cols")) return rc;
    PSX_TIMED("k_detect_rows", st, k_detect_rows<<<ew_grid((int64_t)p->nx * p->ny, 256), 256, 0, st>>>(p->tmp, out, p->sx, p->wx, p->Nx, p->nx, p->ny,
                                                                        p->Wx));
    return launch_check("k_detect_rows");
}

int psx_resize_f32(const float *img, int Nx, int Ny, float *out, int sx, int sy, void *stream) {
    PSX_REQUIRE(img && out && Nx > 0 && Ny > 0 && sx > 0 && sy > 0, "psx_resize_f32: null pointer or empty grid");
    PSX_REQUIRE(Nx / sx >= 1, "psx_resize_f32: target larger than source");
    PSX_TIMED("k_resize", (hipStream_t)stream, k_resize<<<ew_grid((int64_t)sx * sy, 256), 256, 0, (hipStream_t)stream>>>(img, Nx, Ny, out, sx, sy, Nx / sx));
    return launch_check("k_resize");
}

int psx_poisson_f32(const float *lam, float *out, int64_t n, uint64_t seed, void *stream) {
    PSX_REQUIRE(lam && out && n >= 0, "psx_poisson_f32: null pointer or negative n");
    if (n == 0) return 0;
    PSX_TIMED("k_poisson", (hipStream_t)stream, k_poisson<<<ew_grid(n, 256), 256, 0, (hipStream_t)stream>>>(lam, out, n, seed));
    return launch_check("k_poisson");
}

}  // extern "C"
