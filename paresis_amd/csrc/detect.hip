// detect.hip -- detector model: blur, resampling, PSF, shot noise (K14-K19).
//
// Replaces Detector.detection (Detector.py:79-119), resize (:185-198) and create_gaussian_shape (:201-220).
// reflect-pad(15*ov) -> Gaussian source blur (zero-extended 'same' convolution) -> ov x ov block SUM -> Gaussian PSF
// -> crop(15) is a chain of linear, separable operators, so per axis it is a banded matrix.  The operators are composed
// on the host in float64 by pushing unit vectors back through the chain (so reflect padding, zero extension at the padded
// border, banker's rounding of the kernel support and the bin-SUM are reproduced exactly) and applied in two STAGES:
//   front  F [npad x N]: pad + source blur + bin  -- band width ov + 2 r_src, applied at study resolution, where the bytes are;
//   back   B [n x npad]: PSF + crop               -- band width 2 r_psf + 1, applied at detector resolution (ov^2 fewer pixels).
// (One composite [n x N] matrix per axis, as first built, has width (2 r_psf + 1) ov + 2 r_src: 36 taps for the 16384^2 ->
// 4096^2 case instead of 4 + 9, and every tap was a strided, uncoalesced load: 6.3 ms where the image streams in 0.3.)
// out = B_x F_x img F_y^T B_y^T in four passes: F_y along the contiguous axis first (the only pass that touches the
// full-resolution image), F_x, then the two small PSF passes.  Without a PSF the front operator carries the crop and the
// back passes are skipped.
#include <algorithm>
#include <vector>

#include "common.hpp"

using namespace psx;

// one banded operator: output o = sum_{k<W} w[o][k] * in[start[o] + k]
struct BandOp {
    int n_out = 0, n_in = 0, W = 0;
    int span = 0;          // longest input span of 256 consecutive outputs (LDS staging of the contiguous pass)
    int *start = nullptr;  // [n_out] device
    int2 *blk = nullptr;   // [ceil(n_out/256)] device: input range [x, y) the outputs of a 256-block read (start[] is not
                           // monotonic where the reflect pad folds the axis back)
    float *w = nullptr;    // [n_out][W] device (row-major: the axis-0 pass reads a row with scalar loads)
    float *wT = nullptr;   // [W][n_out] device (transposed: the contiguous pass reads it coalesced)
    std::vector<int> h_start;   // host copy of start[] (row tiles of the fused pair are laid out from it)
    bool stencil = false;       // every output reads start[0] + o .. with the SAME weights (a plain convolution: k_psf_tile)
    std::vector<float> h_w0;    // those weights
};

// A contiguous-axis operator followed by an axis-0 operator in ONE pass (k_band_pair): per tile of RT output rows the
// rows [x, y) of the input image that the axis-0 operator reads for them.
struct BandPair {
    bool ok = false;
    bool cheap = false;         // the tiles' halos re-read less than 30 % of the input rows (else two passes move fewer bytes)
    int RT = 0, n_rtiles = 0, mid_rows = 0, span_ld = 0;
    int2 *r_tile = nullptr;     // [n_rtiles] device
    size_t lds = 0;
};

struct psx_detector_plan {
    int Nx, Ny, ov, nx, ny, margin;
    BandOp fx, fy, bx, by;   // front / back operator of each axis; back.n_out == 0 when there is no PSF
    float *t1 = nullptr;     // [Nx][fy.n_out]
    float *t2 = nullptr;     // [fx.n_out][fy.n_out]   (only with a PSF)
    float *t3 = nullptr;     // [fx.n_out][ny]         (only with a PSF)
    BandPair front, back;    // fused (contiguous axis, axis 0) pairs; front writes t2p, back reads it
    float *t2p = nullptr;    // [PSX_MAX_DETECT][fx.n_out][pitch2], pitch2 = fy.n_out rounded up to 4 (16-byte rows for the back
                             // pair's loads); one per image of a psx_detect_multi_f32 launch
    int pitch2 = 0;
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
// stage: STAGE_ALL = the whole chain [n x N]; STAGE_FRONT = pad + blur + bin [npad x N] (rows = binned, uncropped
// indices); STAGE_BACK = PSF + crop [n x npad].
enum { STAGE_ALL = 0, STAGE_FRONT = 1, STAGE_BACK = 2 };

void compose_axis(int N, int ov, int n, int margin, double sigma_src, double sigma_psf, std::vector<int> &start,
                  std::vector<float> &weights, int &W, int stage = STAGE_ALL) {
    const int Npad = N + 2 * margin * ov;   // DET:93
    const int npad = n + 2 * margin;        // DET:103
    const int s = Npad / npad;              // DET:192 (the axis-0 factor is used on both axes; equal for ov grids)
    int r1 = 0, r2 = 0;
    std::vector<double> g1, g2;
    if (sigma_src != 0.0) g1 = gauss1d(sigma_src, r1);
    if (sigma_psf != 0.0) g2 = gauss1d(sigma_psf, r2);
    if (stage == STAGE_BACK) {              // rows of PSF + crop over the binned, uncropped axis
        W = 2 * r2 + 1;
        start.assign(n, 0);
        weights.assign((size_t)n * W, 0.f);
        for (int r = 0; r < n; ++r) {
            const int p = r + margin;
            const int qa = std::max(0, p - r2), qb = std::min(npad - 1, p + r2);
            const int st = std::max(0, std::min(qa, npad - W));
            start[r] = st;
            for (int q = qa; q <= qb; ++q) weights[(size_t)r * W + (q - st)] = (float)(g2.empty() ? 1.0 : g2[r2 + (p - q)]);
        }
        return;
    }
    const int nrows = stage == STAGE_FRONT ? npad : n;
    std::vector<std::vector<double>> rows(nrows);
    std::vector<int> lo(nrows, 0);
    std::vector<double> wb(npad, 0.0), wp(Npad, 0.0), wq(Npad, 0.0), acc(N, 0.0);
    for (int r = 0; r < nrows; ++r) {
        // crop^T: unit at binned index r+margin (DET:118); PSF^T (DET:106-108, zero-extended 'same')
        const int p = stage == STAGE_FRONT ? r : r + margin;
        const int rr2 = stage == STAGE_FRONT ? 0 : r2;
        const int qa = std::max(0, p - rr2), qb = std::min(npad - 1, p + rr2);
        for (int q = qa; q <= qb; ++q) wb[q] = (g2.empty() || stage == STAGE_FRONT) ? 1.0 : g2[r2 + (p - q)];
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
    for (int r = 0; r < nrows; ++r) W = std::max(W, (int)rows[r].size());
    start.assign(nrows, 0);
    weights.assign((size_t)nrows * W, 0.f);
    for (int r = 0; r < nrows; ++r) {
        const int st = std::max(0, std::min(lo[r], N - W));
        start[r] = st;
        for (size_t w = 0; w < rows[r].size(); ++w) weights[(size_t)r * W + (lo[r] - st) + w] = (float)rows[r][w];
    }
}

// Banded operator along the CONTIGUOUS axis: out[i][c] = sum_k wT[k][c] * in[i][start[c] + k].
// A workgroup owns 256 consecutive outputs c and a strip of rows.  The weights of output c sit in registers (WMAX of
// them, the band is narrower than that); the input span the 256 outputs need -- start[c0] .. start[c0+255] + W, about
// 256 ov + W floats -- is staged row by row (RG rows per round) in LDS with coalesced loads, so HBM sees every input once.
template <int WMAX, int RG>
__global__ __launch_bounds__(256) void k_band_cols(const float *__restrict__ in, float *__restrict__ out,
                                                   const int *__restrict__ start, const int2 *__restrict__ blk,
                                                   const float *__restrict__ wT, int R, int Cin, int Cout, int W,
                                                   int span_ld, int rows_per_block) {
    extern __shared__ float sdet[];                       // [RG][span_ld]
    const int c0 = blockIdx.x * 256, c = c0 + threadIdx.x;
    const bool live = c < Cout;
    const int cl = live ? c : Cout - 1;
    const int s0 = blk[blockIdx.x].x, s1 = blk[blockIdx.x].y;             // input range of this block's outputs
    const int span = s1 - s0, valid = min(s1, Cin) - s0;                  // a band wider than the axis reads zeros beyond it
    const int off = start[cl] - s0;
    float w[WMAX];
#pragma unroll
    for (int k = 0; k < WMAX; ++k) w[k] = k < W ? wT[(int64_t)k * Cout + cl] : 0.f;
    const int r_begin = blockIdx.y * rows_per_block, r_end = min(R, r_begin + rows_per_block);
    for (int i0 = r_begin; i0 < r_end; i0 += RG) {
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < RG; ++rr) {
            const int i = min(i0 + rr, R - 1);
            const float *row = in + (int64_t)i * Cin + s0;
            for (int t = threadIdx.x; t < span; t += 256) sdet[rr * span_ld + t] = t < valid ? row[t] : 0.f;
        }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int rr = 0; rr < RG; ++rr) {
                if (i0 + rr < r_end) {
                    const float *x = sdet + rr * span_ld + off;
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < WMAX; ++k)
                        if (k < W) acc = fmaf(w[k], x[k], acc);     // off + k < span for every k < W by construction
                    out[(int64_t)(i0 + rr) * Cout + c] = acc;
                }
            }
        }
    }
}

// The same with 16-byte staging loads, for rows that are 16-byte aligned (Cin % 4 == 0, aligned base) and spans of at most
// 64 * MIT float4 per row: wave v of the workgroup stages rows v, v+4 of the round; all of a thread's loads (2 rows x MIT
// float4) are issued before the first LDS write, so ~10 KB per wave are in flight.
template <int WMAX, int MIT>
__global__ __launch_bounds__(256) void k_band_cols_v4(const float *__restrict__ in, float *__restrict__ out,
                                                      const int *__restrict__ start, const int2 *__restrict__ blk,
                                                      const float *__restrict__ wT, int R, int Cin, int Cout, int W,
                                                      int span_ld, int rows_per_block) {
    constexpr int RG = 8;
    extern __shared__ __attribute__((aligned(16))) float sdet[];     // [RG][span_ld], span_ld % 4 == 0
    const int c0 = blockIdx.x * 256, c = c0 + threadIdx.x;
    const bool live = c < Cout;
    const int cl = live ? c : Cout - 1;
    const int s0 = blk[blockIdx.x].x & ~3, s1 = blk[blockIdx.x].y;
    const int n4 = (s1 - s0 + 3) >> 2;                                 // float4 per row (<= 64 * MIT, checked by the host)
    const int off = start[cl] - s0;
    float w[WMAX];
#pragma unroll
    for (int k = 0; k < WMAX; ++k) w[k] = k < W ? wT[(int64_t)k * Cout + cl] : 0.f;
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int r_begin = blockIdx.y * rows_per_block, r_end = min(R, r_begin + rows_per_block);
    for (int i0 = r_begin; i0 < r_end; i0 += RG) {
        float4 v[2][MIT];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = min(i0 + wv + 4 * h, R - 1);
            const float4 *row = reinterpret_cast<const float4 *>(in + (int64_t)i * Cin + s0);
#pragma unroll
            for (int m = 0; m < MIT; ++m) {
                const int t4 = ln + 64 * m;
                // a float4 of an aligned row lies wholly inside or wholly outside it; beyond the row: zeros (a band wider than the axis)
                v[h][m] = (t4 < n4 && s0 + 4 * t4 < Cin) ? row[t4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();                                               // the previous round's reads are done
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int m = 0; m < MIT; ++m) {
                const int t4 = ln + 64 * m;
                if (t4 < n4) *reinterpret_cast<float4 *>(sdet + (wv + 4 * h) * span_ld + 4 * t4) = v[h][m];
            }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int rr = 0; rr < RG; ++rr) {
                if (i0 + rr < r_end) {
                    const float *x = sdet + rr * span_ld + off;
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < WMAX; ++k)
                        if (k < W) acc = fmaf(w[k], x[k], acc);
                    out[(int64_t)(i0 + rr) * Cout + c] = acc;
                }
            }
        }
    }
}

// Banded operator along axis 0: out[r][c] = sum_k w[r][k] * in[start[r] + k][c].  Lanes run along c (coalesced), a
// workgroup takes 256 columns and a strip of output rows; the weights of a row are wave-uniform (scalar loads).
// RB output rows are formed together: their RB x W loads are independent and issued back to back (the narrow front
// operator would otherwise have W = ov loads in flight per thread).  WMAX = 0: any band width, one row at a time.
template <int WMAX, int RB>
__global__ __launch_bounds__(256) void k_band_rows(const float *__restrict__ in, float *__restrict__ out,
                                                   const int *__restrict__ start, const float *__restrict__ w, int Rin,
                                                   int Rout, int C, int W, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r_begin = blockIdx.y * rows_per_block, r_end = min(Rout, r_begin + rows_per_block);
    if constexpr (WMAX == 0) {
        for (int r = r_begin; r < r_end; ++r) {
            const int st = start[r];
            const float *wr = w + (int64_t)r * W;
            const float *col = in + (int64_t)st * C + c;
            const int lim = min(W, Rin - st);
            float acc = 0.f;
            for (int k = 0; k < lim; ++k) acc = fmaf(wr[k], col[(int64_t)k * C], acc);
            out[(int64_t)r * C + c] = acc;
        }
    } else {
        for (int r0 = r_begin; r0 < r_end; r0 += RB) {
            float x[RB][WMAX];
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int st = start[min(r0 + b, Rout - 1)];
#pragma unroll
                for (int k = 0; k < WMAX; ++k)
                    if (k < W) x[b][k] = in[(int64_t)min(st + k, Rin - 1) * C + c];     // clamped: the weight there is not used
            }
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int r = min(r0 + b, Rout - 1);
                const float *wr = w + (int64_t)r * W;
                const int lim = min(W, Rin - start[r]);
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < WMAX; ++k)
                    if (k < lim) acc = fmaf(wr[k], x[b][k], acc);
                if (r0 + b < r_end) out[(int64_t)r * C + c] = acc;
            }
        }
    }
}

// ---- a contiguous-axis operator and an axis-0 operator in one pass ------------------------------------------------------
// out = R (C in^T)^T for a tile of RT output rows x 256 output columns: the rows [ilo, ihi) of `in` that R reads for the
// tile are staged through LDS eight at a time (16-byte loads, all of a thread's loads in flight before the first LDS
// write), C is applied from LDS with the column's weights in registers, and the results stay in LDS (mid[row][256]) until R
// combines them -- the intermediate image of the two-pass form (34 MB written and read again per 4096^2 image for the front
// pair, 17 MB for the PSF pair) never exists.  Tiles are walked by a grid-stride loop (column blocks of a row tile are
// neighbours in the walk: they share input rows in L2).
constexpr int PAIR_MIT = 5;      // float4 per lane and staged row: spans up to 1280 floats

struct PairArgs {
    const float *in[PSX_MAX_DETECT];   // blockIdx.z = image of the launch (psx_detect_multi_f32: the images of one energy bin)
    float *out[PSX_MAX_DETECT];
    int in_pitch, out_pitch;
    const int *c_start;
    const int2 *c_blk;
    const float *c_wT;
    int c_W, Cin, Cout;
    const int *r_start;
    const float *r_w;
    int r_W, Rin, Rout;
    const int2 *r_tile;
    int RT, n_rtiles, span_ld, mid_rows;
};

template <int WC, int MIT>
__global__ __launch_bounds__(256) void k_band_pair(PairArgs a) {
    constexpr int H = 2, RG = 4 * H;                                 // two rows per wave and round
    extern __shared__ __attribute__((aligned(16))) float sdet[];     // stage [RG][span_ld] | mid [mid_rows][256] | wsh [RT][r_W] | ssh [RT]
    float *mid = sdet + RG * a.span_ld;
    float *wsh = mid + a.mid_rows * 256;
    int *ssh = reinterpret_cast<int *>(wsh + a.RT * a.r_W);
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const float *const img_in = a.in[blockIdx.z];
    float *const img_out = a.out[blockIdx.z];
    // a workgroup keeps its column block: the column's weights and the staged span are set up once
    const int cbk = blockIdx.x;
    const int c = cbk * 256 + threadIdx.x;
    const bool live = c < a.Cout;
    const int cl = live ? c : a.Cout - 1;
    const int s0 = a.c_blk[cbk].x & ~3, s1 = a.c_blk[cbk].y;
    const int n4 = (s1 - s0 + 3) >> 2;                                 // float4 per row (<= 64 * MIT, checked by the host)
    const int off = a.c_start[cl] - s0;
    // The column's taps are read from LDS as whole, 16-byte aligned float4s starting at off & ~3 (consecutive outputs start
    // `ov` floats apart: at ov = 4 their 4-byte reads of one tap landed on 8 of the 32 banks, a four-way conflict on every
    // tap -- MI355X_MICROARCH.md, LDS table), and the weights are shifted by off & 3 to match: NQ float4s cover any shift of up
    // to WC taps, the shifted-in weights are zeros (a zero weight times a finite sample adds nothing: same sums as before).
    constexpr int NQ = (WC + 3 + 3) / 4;
    const int sh = off & 3, offa = off & ~3;
    float w[4 * NQ];
#pragma unroll
    for (int j = 0; j < 4 * NQ; ++j) {
        const int k = j - sh;
        w[j] = (k >= 0 && k < a.c_W) ? a.c_wT[(int64_t)k * a.Cout + cl] : 0.f;
    }
    // the tail of a staged row beyond the block's span is never written by the rounds below, but the aligned reads may touch
    // it (with zero weights): make it zeros once (stale LDS could hold NaN bit patterns)
    for (int rr = 0; rr < RG; ++rr)
        for (int t = 4 * n4 + (int)threadIdx.x; t < a.span_ld; t += 256) sdet[rr * a.span_ld + t] = 0.f;
    float4 v[H][MIT];
    auto fetch = [&](int i0, int ihi) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int i = min(i0 + wv + 4 * h, ihi - 1);
            const float4 *row = reinterpret_cast<const float4 *>(img_in + (int64_t)i * a.in_pitch + s0);
#pragma unroll
            for (int m = 0; m < MIT; ++m) {
                const int t4 = ln + 64 * m;
                // rows are padded to whole float4s (in_pitch % 4 == 0); beyond the axis: zeros (a band wider than the axis)
                v[h][m] = (t4 < n4 && s0 + 4 * t4 < a.in_pitch) ? row[t4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    for (int rt = blockIdx.y; rt < a.n_rtiles; rt += gridDim.y) {
        const int ilo = a.r_tile[rt].x, ihi = a.r_tile[rt].y;
        const int r0 = rt * a.RT, r1 = min(a.Rout, r0 + a.RT);
        // the axis-0 operator's rows of this tile: weights and first input row, read back from LDS as broadcasts
        float wpre = 0.f;
        int spre = 0;
        if ((int)threadIdx.x < (r1 - r0) * a.r_W) wpre = a.r_w[(int64_t)r0 * a.r_W + threadIdx.x];
        if ((int)threadIdx.x < r1 - r0) spre = a.r_start[r0 + threadIdx.x];
        fetch(ilo, ihi);
        for (int i0 = ilo; i0 < ihi; i0 += RG) {
            __syncthreads();                                           // the previous round's (and tile's) reads are done
            if (i0 == ilo) {
                if ((int)threadIdx.x < a.RT * a.r_W) wsh[threadIdx.x] = wpre;
                if ((int)threadIdx.x < a.RT) ssh[threadIdx.x] = spre;
            }
#pragma unroll
            for (int h = 0; h < H; ++h)
#pragma unroll
                for (int m = 0; m < MIT; ++m) {
                    const int t4 = ln + 64 * m;
                    if (t4 < n4) {
                        float4 x = v[h][m];
                        const int e = s0 + 4 * t4;                     // columns at and beyond Cin are pitch padding, not data
                        if (e + 3 >= a.Cin) {
                            x.x = e < a.Cin ? x.x : 0.f;
                            x.y = e + 1 < a.Cin ? x.y : 0.f;
                            x.z = e + 2 < a.Cin ? x.z : 0.f;
                            x.w = 0.f;
                        }
                        *reinterpret_cast<float4 *>(sdet + (wv + 4 * h) * a.span_ld + 4 * t4) = x;
                    }
                }
            __syncthreads();
            if (i0 + RG < ihi) fetch(i0 + RG, ihi);                    // the next round's rows fly during this round's arithmetic
#pragma unroll
            for (int rr = 0; rr < RG; ++rr) {
                if (i0 + rr < ihi) {
                    const float4 *x = reinterpret_cast<const float4 *>(sdet + rr * a.span_ld + offa);
                    float4 q[NQ];
#pragma unroll
                    for (int m = 0; m < NQ; ++m) q[m] = x[m];          // offa + 4 NQ <= span_ld (host: span + 24)
                    float acc = 0.f;
#pragma unroll
                    for (int m = 0; m < NQ; ++m) {
                        acc = fmaf(w[4 * m], q[m].x, acc);
                        acc = fmaf(w[4 * m + 1], q[m].y, acc);
                        acc = fmaf(w[4 * m + 2], q[m].z, acc);
                        acc = fmaf(w[4 * m + 3], q[m].w, acc);
                    }
                    mid[(i0 + rr - ilo) * 256 + threadIdx.x] = acc;
                }
            }
        }
        // each thread reads back only what it wrote (its own column of mid): no barrier needed before the axis-0 operator
        for (int r = r0; r < r1; ++r) {
            const int st = ssh[r - r0];
            const float *wr = wsh + (r - r0) * a.r_W;
            const int lim = min(a.r_W, a.Rin - st);
            const float *col = mid + (st - ilo) * 256 + threadIdx.x;
            float acc = 0.f;
#pragma unroll 4
            for (int k = 0; k < lim; ++k) acc = fmaf(wr[k], col[k * 256], acc);
            if (live) img_out[(int64_t)r * a.out_pitch + c] = acc;
        }
    }
}

// ---- the PSF stage as a register-tiled stencil -----------------------------------------------------------------------------
// When both operators of the back stage are plain convolutions (every output reads the same W taps at start[0] + its index:
// PSF radius <= the cropped margin, the usual case) the stage is out[r][c] = sum_k sum_l gx[k] gy[l] in[ox + r + k][oy + c + l].
// A workgroup owns 16 x 128 outputs: the (16 + WT - 1) x (128 + WT - 1) inputs are staged in LDS with all loads in flight at
// once; the contiguous pass gives every thread 4 adjacent outputs from 4 + WT - 1 consecutive samples (three 16-byte LDS reads
// for 36 multiply-adds at WT = 9), the axis-0 pass 4 x 4 outputs from 4 + WT - 1 rows of 4 (twelve reads for 144), the
// weights are scalars of the launch.  Same sums in the same order as k_band_pair's (whose aligned tap reads fetched 16 taps for
// 9, at 1.5x halo rows: timing experiments in gpurun_out/r5s68): bit-identical images.  WT = 4 WQ - 3 taps, zero-padded.
struct PsfArgs {
    const float *in[PSX_MAX_DETECT];
    float *out[PSX_MAX_DETECT];
    int in_pitch, out_pitch, Rin, Cin, Rout, Cout, ox, oy;
    float gx[20], gy[20];
};

#ifndef PSX_PSF_TR
#define PSX_PSF_TR 16        // output rows per workgroup (build-time A/B, gpurun_out/r5s71: 16 -> 25 KB of LDS, six workgroups per CU: 0.0270 against 0.0287 ms with 32)
#endif
template <int WQ>
__global__ __launch_bounds__(256) void k_psf_tile(PsfArgs a) {
    constexpr int WT = 4 * WQ - 3, TR = PSX_PSF_TR, TC = 128, SR = TR + WT - 1, SC = TC + 4 * WQ - 4;   // SC: TC + WT - 1, a multiple of 4
    __shared__ __attribute__((aligned(16))) float sin_[SR * SC];
    __shared__ __attribute__((aligned(16))) float mid[SR * TC];
    const float *const in = a.in[blockIdx.z];
    float *const out = a.out[blockIdx.z];
    const int r0 = blockIdx.y * TR, c0 = blockIdx.x * TC;
    const int tid = threadIdx.x;
    // stage: clamped addresses (only outputs beyond the image read clamped samples, and they are not stored)
    constexpr int NS = (SR * SC + 255) / 256;
    float x[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int e = min(tid + 256 * j, SR * SC - 1);
        const int i = e / SC, l = e - i * SC;
        x[j] = in[(int64_t)min(a.ox + r0 + i, a.Rin - 1) * a.in_pitch + min(a.oy + c0 + l, a.Cin - 1)];
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int e = tid + 256 * j;
        if (e < SR * SC) sin_[e] = x[j];
    }
    __syncthreads();
    // contiguous pass: SR rows x 32 quads of 4 outputs
    for (int q = tid; q < SR * (TC / 4); q += 256) {
        const int i = q >> 5, c4 = (q & 31) * 4;
        const float4 *src = reinterpret_cast<const float4 *>(sin_ + i * SC + c4);
        float v[4 * WQ];
#pragma unroll
        for (int m = 0; m < WQ; ++m) {
            const float4 t = src[m];
            v[4 * m] = t.x; v[4 * m + 1] = t.y; v[4 * m + 2] = t.z; v[4 * m + 3] = t.w;
        }
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < WT; ++l)
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = fmaf(a.gy[l], v[l + u], o[u]);
        *reinterpret_cast<float4 *>(mid + i * TC + c4) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    // axis-0 pass: thread = 4 rows x 4 columns
    const int c4 = (tid & 31) * 4, rq = (tid >> 5) * 4;
    if (rq >= TR) return;
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < WT + 3; ++k) {
        const float4 m = *reinterpret_cast<const float4 *>(mid + (rq + k) * TC + c4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = k - u;                                   // output row rq + u takes this row with tap t
            if (t >= 0 && t < WT) {
                acc[u].x = fmaf(a.gx[t], m.x, acc[u].x);
                acc[u].y = fmaf(a.gx[t], m.y, acc[u].y);
                acc[u].z = fmaf(a.gx[t], m.z, acc[u].z);
                acc[u].w = fmaf(a.gx[t], m.w, acc[u].w);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = r0 + rq + u, c = c0 + c4;
        if (r < a.Rout) {
            float *dst = out + (int64_t)r * a.out_pitch + c;
            if (c + 3 < a.Cout && (a.out_pitch & 3) == 0 && ((uintptr_t)out & 15) == 0) {
                *reinterpret_cast<float4 *>(dst) = acc[u];
            } else {
                if (c < a.Cout) dst[0] = acc[u].x;
                if (c + 1 < a.Cout) dst[1] = acc[u].y;
                if (c + 2 < a.Cout) dst[2] = acc[u].z;
                if (c + 3 < a.Cout) dst[3] = acc[u].w;
            }
        }
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

// ---- photon-count images as 16-bit integers (the gather of main.py's per-position stacks moves half the bytes) ----------
// dst[p] = src[p] for counts below 65535; a larger count leaves the escape code 65535 and an entry (index0 + p, count) in the
// exception table exc[cap][2] (rare: caustic peaks).  *overflow is raised when some src[p] is not an integer in [0, 2^24] or
// the table is full -- dst is then not a copy of src and the caller sends the float32 image instead.  Eight pixels per
// thread: two 16-byte loads, one 16-byte store.
__global__ __launch_bounds__(256) void k_pack_u16(const float *__restrict__ src, uint16_t *__restrict__ dst, int64_t n,
                                                  int64_t index0, int32_t *__restrict__ exc, int32_t *__restrict__ exc_count,
                                                  int cap, int *__restrict__ overflow) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    const int lane = threadIdx.x & 63;
    // the escapes of a wave's pixels take their table slots with ONE atomic (an image behind a sphere membrane has caustics over
    // 3 % of its pixels: an atomic each -- or one per wave and pixel slot, as the compiler aggregates them -- is 1e5 returning
    // atomics on one word, about a millisecond): exclusive scan of the lanes' counts, the last lane draws the wave's block
    auto slots = [&](int c) {
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            incl += lane >= o ? t : 0;
        }
        int base = 0;
        if (lane == 63) base = atomicAdd(exc_count, incl);
        return __shfl(base, 63) + incl - c;
    };
    auto cvt = [&](float x, unsigned &q) {           // q = the 16-bit code; returns the count itself
        const float c = fminf(fmaxf(x, 0.f), 16777216.f);
        const unsigned v = (unsigned)c;
        bad |= !((float)v == x);                     // NaN, inf, negatives, fractions and counts above 2^24 all land here
        q = min(v, 65535u);
        return v;
    };
    auto escape = [&](int &e, unsigned v, int64_t p) {
        if (v >= 65535u) {
            if (e < cap) {
                exc[2 * e] = (int32_t)(index0 + p);
                exc[2 * e + 1] = (int32_t)v;
            } else {
                bad = true;
            }
            ++e;
        }
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const int64_t n8 = vec ? n / 8 : 0;
    for (int64_t g0 = (int64_t)blockIdx.x * blockDim.x; g0 < n8; g0 += stride) {     // whole waves: the scan holds shuffles
        const int64_t g = g0 + threadIdx.x;
        const bool live = g < n8;
        float x[8];
        unsigned q[8], v[8];
        if (live) {
            const float4 a = reinterpret_cast<const float4 *>(src)[2 * g], b = reinterpret_cast<const float4 *>(src)[2 * g + 1];
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        }
        int c = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = live ? cvt(x[k], q[k]) : 0u;
            c += v[k] >= 65535u ? 1 : 0;
        }
        if (__any(c > 0)) {
            int e = slots(c);
#pragma unroll
            for (int k = 0; k < 8; ++k) escape(e, v[k], 8 * g + k);
        }
        if (live) {
            uint4 o;
            o.x = q[0] | (q[1] << 16);
            o.y = q[2] | (q[3] << 16);
            o.z = q[4] | (q[5] << 16);
            o.w = q[6] | (q[7] << 16);
            reinterpret_cast<uint4 *>(dst)[g] = o;
        }
    }
    for (int64_t p0 = n8 * 8 + (int64_t)blockIdx.x * blockDim.x; p0 < n; p0 += stride) {
        const int64_t p = p0 + threadIdx.x;
        unsigned q = 0u, v = 0u;
        if (p < n) v = cvt(src[p], q);
        const int c = v >= 65535u ? 1 : 0;
        if (__any(c > 0)) {
            int e = slots(c);
            escape(e, v, p);
        }
        if (p < n) dst[p] = (uint16_t)q;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(overflow, 1);
}

__global__ __launch_bounds__(256) void k_unpack_u16(const uint16_t *__restrict__ src, float *__restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const int64_t n8 = vec ? n / 8 : 0;
    for (int64_t g = t0; g < n8; g += stride) {
        const uint4 q = reinterpret_cast<const uint4 *>(src)[g];
        reinterpret_cast<float4 *>(dst)[2 * g] = make_float4((float)(q.x & 0xffffu), (float)(q.x >> 16), (float)(q.y & 0xffffu), (float)(q.y >> 16));
        reinterpret_cast<float4 *>(dst)[2 * g + 1] = make_float4((float)(q.z & 0xffffu), (float)(q.z >> 16), (float)(q.w & 0xffffu), (float)(q.w >> 16));
    }
    for (int64_t p = n8 * 8 + t0; p < n; p += stride) dst[p] = (float)src[p];
}

// the escaped counts go back in (after k_unpack_u16 on the same stream)
__global__ __launch_bounds__(256) void k_unpack_exceptions(float *__restrict__ dst, int64_t n, const int32_t *__restrict__ exc,
                                                           const int32_t *__restrict__ exc_count, int cap) {
    const int m = min(*exc_count, cap);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < m; e += gridDim.x * blockDim.x) {
        const int64_t p = (uint32_t)exc[2 * e];
        if (p < n) dst[p] = (float)(uint32_t)exc[2 * e + 1];
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
    __device__ void next(uint64_t ctr, uint32_t sub, uint64_t seed, float u[4], uint32_t tag) {
        c[0] = (uint32_t)ctr; c[1] = (uint32_t)(ctr >> 32); c[2] = sub; c[3] = tag;
        k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
        for (int r = 0; r < 10; ++r) {
            round_();
            k[0] += 0x9E3779B9u;
            k[1] += 0xBB67AE85u;
        }
        for (int i = 0; i < 4; ++i) u[i] = ((float)(c[i] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
    }
};

// The hardware's own reciprocal, square root and logarithm (one instruction each, about 1 ulp): the IEEE forms of x / y and
// sqrtf cost ten more instructions apiece, and k_poisson is bound by its instruction count (34 M vector instructions per launch
// of two 2048^2 images, the vector pipe busy throughout: gpurun_out/r5s58).  PTRS's constants are fits to four digits; an ulp in
// b or vr is far inside their margin, and the acceptance test's two sides (below) are compared at 1e-4.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718f; }     // x normal: v_log_f32 is log2

// log of the Poisson probability  -L + k log L - log k!  for PTRS's acceptance test.  The three terms are of order
// L log L and cancel to order 1, so a direct float32 evaluation is useless for large means.  With x = (k - L)/L and Stirling's series for log k!
//     log pmf = -L g(x) - log(2 pi k)/2 - 1/(12 k) + 1/(360 k^3),      g(x) = (1 + x) log(1 + x) - x = x^2/2 - x^3/6 + ...
// every term is small and float32 is enough (|error| < 1e-4 for k, L >= 64: an acceptance decision can only change when the
// two sides are that close); small k or L keep the float64 form.
__device__ __forceinline__ float log_pmf(float k, float L) {
    if (k >= 64.f && L >= 64.f) {
        const float x = (k - L) * fast_rcp(L);                 // k - L is exact (Sterbenz) or far in the tail
        float g;
        if (fabsf(x) < 0.125f) {                               // sum_{n>=2} (-1)^n x^n / (n (n-1)), 9 terms: < 1e-9 relative
            g = 1.f / 90.f;
            g = fmaf(g, -x, 1.f / 72.f);
            g = fmaf(g, -x, 1.f / 56.f);
            g = fmaf(g, -x, 1.f / 42.f);
            g = fmaf(g, -x, 1.f / 30.f);
            g = fmaf(g, -x, 1.f / 20.f);
            g = fmaf(g, -x, 1.f / 12.f);
            g = fmaf(g, -x, 1.f / 6.f);
            g = fmaf(g, -x, 0.5f);
            g *= x * x;
        } else {
            g = (1.f + x) * log1pf(x) - x;
        }
        const float ik = fast_rcp(k);
        return -L * g - 0.5f * fast_log(6.2831853f * k) - ik * (1.f / 12.f) + ik * ik * ik * (1.f / 360.f);
    }
    // small k or L (10 <= L < 64): the terms are at most a few hundred, float32 leaves an absolute error ~2e-5 in a quantity
    // that is compared with the log of a uniform -- a decision changes with probability ~1e-5
    return -L + k * logf(L) - lgammaf(k + 1.f);
}

// Poisson draw: product-of-uniforms for lam < 10, Hormann's PTRS transformed rejection otherwise (both exact samplers).
// The draw of pixel p is a pure function of (seed, p):
//   * its FIRST PTRS candidate takes two words of the Philox block (counter p >> 1, tag 0x5058): one block serves the
//     first candidates of two neighbouring pixels (78 % of the pixels accept it through the squeeze -- P(|U| <= 0.43) x vr = 0.86 x 0.91 at a mean of 7500 --, no logarithm);
//   * everything else -- later candidates, the product of uniforms of small means -- comes from the pixel's own stream
//     (counter p, sub-counter 1, 2, ..., tag 0x5059).
struct PoissonImgs {
    float *img[PSX_MAX_POISSON];
    uint64_t seed[PSX_MAX_POISSON];
};

// One thread draws 4 consecutive pixels (16-byte load and store), in place; blockIdx.y = image of the batch (the three or
// four detector images of one energy bin are drawn by ONE launch, each under its own key).  Pixels whose first candidate
// fails the squeeze go through ONE copy of the exact test + retry loop, one pending pixel per lane at a time (values picked
// with selects: a dynamically indexed register array would live in scratch memory).
__device__ __forceinline__ float pick4(const float v[4], int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3])); }

// everything after the squeeze test of a candidate: the exact acceptance test, then further candidates (PTRS, L >= 10), or
// the product of uniforms (L < 10)
__device__ __forceinline__ float poisson_finish(float L, float U0, float V0, uint64_t p, uint64_t seed) {
    Philox g;
    float u[4];
    if (L < 10.f) {
        const float lim = expf(-L);
        float prod = 1.f;
        int k = 0;
        for (uint32_t sub = 1; sub < 64; ++sub) {
            g.next(p, sub, seed, u, 0x5059u);
            bool done = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!done) {
                    prod *= u[i];
                    if (prod <= lim) done = true;
                    else ++k;
                }
            }
            if (done) break;
        }
        return (float)k;
    }
    const float slam = fast_sqrt(L);
    const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
    const float invalpha = 1.1239f + 1.1328f * fast_rcp(b - 3.4f), vr = 0.9277f - 3.6224f * fast_rcp(b - 2.f);
    float U = U0, V = V0, U2 = 0.f, V2 = 0.f;
    uint32_t sub = 1;
    bool have = false;
    float res = floorf(L);
    for (int it = 0; it < 128; ++it) {
        const float us = 0.5f - fabsf(U);
        const float ius = fast_rcp(us);
        const float k = floorf((2.f * a * ius + b) * U + L + 0.43f);
        if (us >= 0.07f && V <= vr) { res = k; break; }
        if (!(k < 0.f || (us < 0.013f && V > us))) {
            const float lhs = fast_log(V * invalpha * fast_rcp(a * ius * ius + b));
            if (lhs <= log_pmf(k, L)) { res = k; break; }
        }
        if (have) {
            U = U2; V = V2; have = false;
        } else {
            g.next(p, sub++, seed, u, 0x5059u);
            U = u[0] - 0.5f; V = u[1]; U2 = u[2] - 0.5f; V2 = u[3]; have = true;
        }
    }
    return res;
}

__global__ __launch_bounds__(256) void k_poisson(PoissonImgs im, const float *__restrict__ lam_single, int64_t n) {
    float *img = im.img[blockIdx.y];
    const float *lam = lam_single ? lam_single : img;
    const uint64_t seed = im.seed[blockIdx.y];
    const int64_t nq = n >> 2;
    const bool vec = ((uintptr_t)img % 16 == 0) && ((uintptr_t)lam % 16 == 0);
    __shared__ float4 wq_all[4][256];                  // per wave: the pending pixels of a round (mean, candidate, owner)
    __shared__ float wr_all[4][256];                   // their draws
    const int lane = threadIdx.x & 63;
    float4 *wq = wq_all[threadIdx.x >> 6];
    float *wr = wr_all[threadIdx.x >> 6];
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < (vec ? nq : 0); q += (int64_t)gridDim.x * blockDim.x) {
        const float4 L4 = reinterpret_cast<const float4 *>(lam)[q];
        const float L[4] = {L4.x, L4.y, L4.z, L4.w};
        Philox g;
        float ua[4], ub[4];
        g.next((uint64_t)(2 * q), 0, seed, ua, 0x5058u);
        g.next((uint64_t)(2 * q + 1), 0, seed, ub, 0x5058u);
        const float U[4] = {ua[0] - 0.5f, ua[2] - 0.5f, ub[0] - 0.5f, ub[2] - 0.5f}, V[4] = {ua[1], ua[3], ub[1], ub[3]};
        float res[4];
        unsigned pending = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float Li = L[i];
            const float slam = fast_sqrt(Li);
            const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
            const float vr = 0.9277f - 3.6224f * fast_rcp(b - 2.f);
            const float us = 0.5f - fabsf(U[i]);
            const float k = floorf((2.f * a * fast_rcp(us) + b) * U[i] + Li + 0.43f);
            const bool fast = Li >= 10.f && us >= 0.07f && V[i] <= vr;
            res[i] = fast ? k : 0.f;
            if (!fast && Li > 0.f) pending |= 1u << i;
        }
        // The pixels the squeeze did not settle (~22 %) are dealt out again over the wave: left with their owners, a wave
        // would run the exact test as often as its unluckiest lane has pending pixels (3-4 times, a sixth of the lanes
        // alive); packed through LDS it runs it once or twice with the lanes full.  A draw is a function of (pixel, key)
        // alone, so who computes it changes nothing.
        int slot[4];
        int total = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned long long m = __ballot((pending >> i) & 1u);
            slot[i] = total + __popcll(m & ((1ull << lane) - 1ull));
            total += __popcll(m);
            if ((pending >> i) & 1u) {
                wq[slot[i]] = make_float4(L[i], U[i], V[i], __uint_as_float((unsigned)(lane * 4 + i)));
            }
        }
        if (total) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // the lanes still in the loop (all of them except in an image's last round) share the entries
            const unsigned long long act = __ballot(true);
            const int nact = __popcll(act), rank = __popcll(act & ((1ull << lane) - 1ull));
            for (int e = rank; e < total; e += nact) {
                const float4 t = wq[e];
                const unsigned who = __float_as_uint(t.w);
                const int64_t qo = q + ((int)(who >> 2) - lane);                // the owner's quad: lanes hold consecutive quads
                wr[e] = poisson_finish(t.x, t.y, t.z, (uint64_t)(4 * qo + (who & 3u)), seed);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if ((pending >> i) & 1u) res[i] = wr[slot[i]];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // read before the next quad's entries land
        }
        reinterpret_cast<float4 *>(img)[q] = make_float4(res[0], res[1], res[2], res[3]);
    }
    // scalar path: the tail of an image whose size is not a multiple of 4, or a misaligned image as a whole
    const int64_t t0 = vec ? nq * 4 : 0;
    for (int64_t p = t0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const float Li = lam[p];
        float r = 0.f;
        if (Li > 0.f) {
            Philox g;
            float u[4];
            g.next((uint64_t)(p >> 1), 0, seed, u, 0x5058u);
            const bool odd = p & 1;
            r = poisson_finish(Li, (odd ? u[2] : u[0]) - 0.5f, odd ? u[3] : u[1], (uint64_t)p, seed);
        }
        img[p] = r;
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
    int rc = 0;
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        PSX_HIP(hipMalloc(dst, bytes));
        PSX_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        p->bytes += bytes;
        return 0;
    };
    auto build = [&](BandOp &op, int N, int n, int stage) -> int {
        std::vector<int> st;
        std::vector<float> w;
        compose_axis(N, ov, n, margin, sigma_src, sigma_psf, st, w, op.W, stage);
        op.n_out = (int)st.size();
        op.n_in = stage == STAGE_BACK ? n + 2 * margin : N;
        op.h_start = st;
        op.stencil = op.W >= 1;
        for (int o = 0; o < op.n_out && op.stencil; ++o) {
            op.stencil = st[o] == st[0] + o;
            for (int k = 0; k < op.W && op.stencil; ++k) op.stencil = w[(size_t)o * op.W + k] == w[k];
        }
        op.h_w0.assign(w.begin(), w.begin() + std::min<size_t>(w.size(), (size_t)op.W));
        std::vector<float> wT((size_t)op.W * op.n_out);
        for (int o = 0; o < op.n_out; ++o)
            for (int k = 0; k < op.W; ++k) wT[(size_t)k * op.n_out + o] = w[(size_t)o * op.W + k];
        op.span = 0;
        std::vector<int2> blk;
        for (int o0 = 0; o0 < op.n_out; o0 += 256) {
            int lo = st[o0], hi = st[o0];
            for (int o = o0; o < std::min(o0 + 256, op.n_out); ++o) {
                lo = std::min(lo, st[o]);
                hi = std::max(hi, st[o]);
            }
            blk.push_back(make_int2(lo, hi + op.W));
            op.span = std::max(op.span, hi + op.W - lo);
        }
        if (int e = up((void **)&op.blk, blk.data(), sizeof(int2) * blk.size())) return e;
        if (int e = up((void **)&op.start, st.data(), sizeof(int) * st.size())) return e;
        if (int e = up((void **)&op.w, w.data(), sizeof(float) * w.size())) return e;
        return up((void **)&op.wT, wT.data(), sizeof(float) * wT.size());
    };
    const bool psf = sigma_psf != 0.0;
    if (!rc) rc = build(p->fx, Nx, nx, psf ? STAGE_FRONT : STAGE_ALL);
    if (!rc) rc = build(p->fy, Ny, ny, psf ? STAGE_FRONT : STAGE_ALL);
    if (!rc && psf) rc = build(p->bx, Nx, nx, STAGE_BACK);
    if (!rc && psf) rc = build(p->by, Ny, ny, STAGE_BACK);
    auto scratch = [&](float **dst, size_t elems) -> int {
        hipError_t e = hipMalloc((void **)dst, sizeof(float) * elems);
        if (e != hipSuccess) return fail((int)e, "psx_detector_plan_create: hipMalloc(scratch) failed: %s", hipGetErrorString(e));
        p->bytes += sizeof(float) * elems;
        return 0;
    };
    if (!rc) rc = scratch(&p->t1, (size_t)Nx * (size_t)p->fy.n_out);
    if (!rc && psf) rc = scratch(&p->t2, (size_t)p->fx.n_out * (size_t)p->fy.n_out);
    if (!rc && psf) rc = scratch(&p->t3, (size_t)p->fx.n_out * (size_t)ny);
    // the fused (contiguous axis, axis 0) pairs: tiles of RT output rows, the largest RT whose LDS footprint leaves four
    // workgroups per CU; a geometry that does not fit (very wide bands) keeps the four-pass form
    auto pair = [&](BandPair &pr, const BandOp &C, const BandOp &R) -> int {
        pr.ok = false;
        if (C.W > 16 || C.span + 3 + 3 > 4 * 64 * PAIR_MIT) return 0;
        pr.span_ld = (C.span + 3 + 24) / 4 * 4;      // + alignment slack of the block's first input + the aligned tap reads' overshoot
        for (int RT : {16, 8, 4}) {
            std::vector<int2> tiles;
            int mid = 1;
            long long rows_read = 0;
            for (int r0 = 0; r0 < R.n_out; r0 += RT) {
                int lo = R.n_in, hi = 0;
                for (int r = r0; r < std::min(R.n_out, r0 + RT); ++r) {
                    lo = std::min(lo, R.h_start[r]);
                    hi = std::max(hi, std::min(R.h_start[r] + R.W, R.n_in));
                }
                tiles.push_back(make_int2(lo, hi));
                mid = std::max(mid, hi - lo);
                rows_read += hi - lo;
            }
            const size_t lds = sizeof(float) * (8 * (size_t)pr.span_ld + (size_t)mid * 256 + (size_t)RT * R.W + RT);
            if ((lds > 40 * 1024 || RT * R.W > 256) && RT > 4) continue;
            if (RT * R.W > 256) return 0;
            if (lds > 64 * 1024) return 0;
            pr.RT = RT; pr.n_rtiles = (int)tiles.size(); pr.mid_rows = mid; pr.lds = lds;
            pr.cheap = (double)rows_read <= 1.3 * (double)R.n_in;
            if (int e = up((void **)&pr.r_tile, tiles.data(), sizeof(int2) * tiles.size())) return e;
            pr.ok = true;
            return 0;
        }
        return 0;
    };
    if (!rc) rc = pair(p->front, p->fy, p->fx);
    if (!rc && psf) rc = pair(p->back, p->by, p->bx);
    p->pitch2 = (p->fy.n_out + 3) / 4 * 4;
    if (!rc && psf && p->back.ok) rc = scratch(&p->t2p, (size_t)PSX_MAX_DETECT * (size_t)p->fx.n_out * (size_t)p->pitch2);
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
    for (BandOp *op : {&p->fx, &p->fy, &p->bx, &p->by}) {
        (void)hipFree(op->start);
        (void)hipFree(op->blk);
        (void)hipFree(op->w);
        (void)hipFree(op->wT);
    }
    (void)hipFree(p->t1);
    (void)hipFree(p->t2);
    (void)hipFree(p->t3);
    (void)hipFree(p->t2p);
    (void)hipFree(p->front.r_tile);
    (void)hipFree(p->back.r_tile);
    delete p;
    return 0;
}

namespace {

// strips of rows per workgroup: enough workgroups to fill 256 CUs several times over, strips long enough to amortise
// the weight loads
int strip_rows(int rows, int col_blocks) {
    int strips = std::max(1, 4096 / std::max(1, col_blocks));
    int per = (rows + strips - 1) / strips;
    return std::max(per, 8);
}

int band_cols(const BandOp &op, const float *in, float *out, int R, hipStream_t st) {
    // 16-byte staging when the rows allow it (the full-resolution pass always does for even grids)
    constexpr int MIT = 5;
    if (op.n_in % 4 == 0 && (uintptr_t)in % 16 == 0 && op.span + 3 + 3 <= 4 * 64 * MIT && op.W <= 64) {
        constexpr int RG8 = 8;
        const int cb = (op.n_out + 255) / 256;
        const int per = (strip_rows(R, cb) + RG8 - 1) / RG8 * RG8;
        const dim3 grid(cb, (R + per - 1) / per);
        const int span_ld = (op.span + 3 + 3 + 4) / 4 * 4;      // + alignment slack of the block's first input, whole float4s
        const size_t lds = sizeof(float) * RG8 * (size_t)span_ld;
#define PSX_BAND_COLS_V4(WMAX)                                                                                          \
    PSX_TIMED("k_band_cols", st, k_band_cols_v4<WMAX, MIT><<<grid, 256, lds, st>>>(in, out, op.start, op.blk, op.wT, R, \
                                                                                     op.n_in, op.n_out, op.W, span_ld, per))
        if (op.W <= 8) PSX_BAND_COLS_V4(8);
        else if (op.W <= 16) PSX_BAND_COLS_V4(16);
        else if (op.W <= 32) PSX_BAND_COLS_V4(32);
        else PSX_BAND_COLS_V4(64);
#undef PSX_BAND_COLS_V4
        return launch_check("k_band_cols");
    }
    constexpr int RG = 4;
    const int cb = (op.n_out + 255) / 256;
    const int per = (strip_rows(R, cb) + RG - 1) / RG * RG;
    const dim3 grid(cb, (R + per - 1) / per);
    const int span_ld = op.span + 1;
    const size_t lds = sizeof(float) * RG * (size_t)span_ld;
    PSX_REQUIRE(lds <= 64 * 1024, "detector: oversampling %d needs %zu bytes of LDS per row strip", op.span / 256, lds);
#define PSX_BAND_COLS(WMAX)                                                                                            \
    PSX_TIMED("k_band_cols", st, k_band_cols<WMAX, RG><<<grid, 256, lds, st>>>(in, out, op.start, op.blk, op.wT, R, op.n_in, \
                                                                                 op.n_out, op.W, span_ld, per))
    if (op.W <= 8) PSX_BAND_COLS(8);
    else if (op.W <= 16) PSX_BAND_COLS(16);
    else if (op.W <= 32) PSX_BAND_COLS(32);
    else if (op.W <= 64) PSX_BAND_COLS(64);
    else if (op.W <= 128) PSX_BAND_COLS(128);
    else return fail(PSX_E_UNSUPPORTED, "detector: band of %d taps (source blur of more than 60 study pixels?)", op.W);
#undef PSX_BAND_COLS
    return launch_check("k_band_cols");
}

int band_rows(const BandOp &op, const float *in, float *out, int C, hipStream_t st) {
    const int cb = (C + 255) / 256;
    const int per = strip_rows(op.n_out, cb);
    const dim3 grid(cb, (op.n_out + per - 1) / per);
#define PSX_BAND_ROWS(WMAX, RB)                                                                                          \
    PSX_TIMED("k_band_rows", st, k_band_rows<WMAX, RB><<<grid, 256, 0, st>>>(in, out, op.start, op.w, op.n_in, op.n_out, \
                                                                             C, op.W, per))
    if (op.W <= 4) PSX_BAND_ROWS(4, 8);
    else if (op.W <= 8) PSX_BAND_ROWS(8, 4);
    else if (op.W <= 16) PSX_BAND_ROWS(16, 2);
    else PSX_BAND_ROWS(0, 1);
#undef PSX_BAND_ROWS
    return launch_check("k_band_rows");
}

}  // namespace

namespace {

int band_pair(const BandPair &pr, const BandOp &C, const BandOp &R, const float *const *in, int in_pitch, float *const *out,
              int out_pitch, int nimg, hipStream_t st) {
    PairArgs a;
    for (int k = 0; k < PSX_MAX_DETECT; ++k) {
        a.in[k] = in[std::min(k, nimg - 1)];
        a.out[k] = out[std::min(k, nimg - 1)];
    }
    a.in_pitch = in_pitch; a.out_pitch = out_pitch;
    a.c_start = C.start; a.c_blk = C.blk; a.c_wT = C.wT; a.c_W = C.W; a.Cin = C.n_in; a.Cout = C.n_out;
    a.r_start = R.start; a.r_w = R.w; a.r_W = R.W; a.Rin = R.n_in; a.Rout = R.n_out;
    a.r_tile = pr.r_tile; a.RT = pr.RT; a.n_rtiles = pr.n_rtiles; a.span_ld = pr.span_ld; a.mid_rows = pr.mid_rows;
    // column blocks x row strips x images; a workgroup walks the row tiles of its strip (its column set-up is done once), and
    // the launch as a whole fills the chip once whatever the number of images
    const int cb = (C.n_out + 255) / 256;
    const int per_cu = (int)std::min<size_t>(8, (160 * 1024) / pr.lds);
    const int strips = std::max(1, std::min(a.n_rtiles, current_cu_count() * std::max(1, per_cu) / (cb * nimg)));
    const dim3 grid(cb, strips, nimg);
    // the column operator's taps sit in registers, a float4 of them per aligned LDS read: as few as the band needs (a plain
    // ov = 2 binning has 2 taps, the usual PSF 9: two reads instead of three, four instead of five)
    if (C.W <= 4) PSX_TIMED("k_band_pair", st, k_band_pair<4, PAIR_MIT><<<grid, 256, pr.lds, st>>>(a));
    else if (C.W <= 8) PSX_TIMED("k_band_pair", st, k_band_pair<8, PAIR_MIT><<<grid, 256, pr.lds, st>>>(a));
    else if (C.W <= 12) PSX_TIMED("k_band_pair", st, k_band_pair<12, PAIR_MIT><<<grid, 256, pr.lds, st>>>(a));
    else PSX_TIMED("k_band_pair", st, k_band_pair<16, PAIR_MIT><<<grid, 256, pr.lds, st>>>(a));
    return launch_check("k_band_pair");
}

// the PSF stage as a stencil (k_psf_tile): both back operators plain convolutions of at most 17 taps
bool psf_stencil_ok(const BandOp &C, const BandOp &R) { return C.stencil && R.stencil && C.W <= 17 && R.W <= 17; }

int psf_tile(const BandOp &C, const BandOp &R, const float *const *in, int in_pitch, float *const *out, int out_pitch, int nimg,
             hipStream_t st) {
    PsfArgs a = {};
    for (int k = 0; k < PSX_MAX_DETECT; ++k) {
        a.in[k] = in[std::min(k, nimg - 1)];
        a.out[k] = out[std::min(k, nimg - 1)];
    }
    a.in_pitch = in_pitch; a.out_pitch = out_pitch;
    a.Rin = R.n_in; a.Cin = C.n_in; a.Rout = R.n_out; a.Cout = C.n_out; a.ox = R.h_start[0]; a.oy = C.h_start[0];
    for (int k = 0; k < R.W; ++k) a.gx[k] = R.h_w0[k];
    for (int l = 0; l < C.W; ++l) a.gy[l] = C.h_w0[l];
    const dim3 grid((C.n_out + 127) / 128, (R.n_out + PSX_PSF_TR - 1) / PSX_PSF_TR, nimg);
    const int W = std::max(C.W, R.W);
    if (W <= 9) PSX_TIMED("k_psf_tile", st, k_psf_tile<3><<<grid, 256, 0, st>>>(a));
    else if (W <= 13) PSX_TIMED("k_psf_tile", st, k_psf_tile<4><<<grid, 256, 0, st>>>(a));
    else PSX_TIMED("k_psf_tile", st, k_psf_tile<5><<<grid, 256, 0, st>>>(a));
    return launch_check("k_psf_tile");
}

bool four_pass_forced() { return debug_switch(DBG_DETECT_4PASS) != 0; }   // diagnostics (psx_debug_switch "detect_4pass")

}  // namespace

// The detector operator on nimg images.  With both stages fused (k_band_pair) the images of the call share each launch
// (grid.z): a 4096^2 -> 2048^2 detection is two launches of 20-25 us whose set-up, first fetch and tail are a good part of
// them, and an energy bin always detects two to four images.  Image k is computed exactly as a call of its own would.
static int detect_impl(psx_detector_plan *p, const float *const *imgs, float *const *outs, int nimg, hipStream_t st) {
    const bool psf = p->bx.n_out != 0;
    // Each stage as ONE fused pass (k_band_pair) where that moves fewer bytes: the front stage when the image rows are 16-byte
    // aligned, the bands fit its tiles and the tiles' halos are small (a wide source blur makes an 8-row tile re-read half
    // its rows: two passes are cheaper then); the PSF stage whenever its bands fit -- it reads the front stage's result at a
    // 16-byte row pitch, which the two-pass front can only write when the row length is a multiple of 4 anyway.
    const bool allowed = !four_pass_forced();
    bool aligned = true;
    for (int k = 0; k < nimg; ++k) aligned = aligned && (uintptr_t)imgs[k] % 16 == 0;
    const bool front_f = allowed && p->front.ok && p->front.cheap && p->Ny % 4 == 0 && aligned;
    const bool back_f = allowed && psf && p->back.ok && p->t2p && (front_f || p->pitch2 == p->fy.n_out);
    if (nimg > 1 && front_f && (back_f || !psf)) {
        float *mids[PSX_MAX_DETECT];
        for (int k = 0; k < nimg; ++k) mids[k] = p->t2p + (size_t)k * p->fx.n_out * p->pitch2;
        if (!psf) return band_pair(p->front, p->fy, p->fx, imgs, p->Ny, outs, p->fy.n_out, nimg, st);
        if (int rc = band_pair(p->front, p->fy, p->fx, imgs, p->Ny, mids, p->pitch2, nimg, st)) return rc;
        if (psf_stencil_ok(p->by, p->bx)) return psf_tile(p->by, p->bx, mids, p->pitch2, outs, p->ny, nimg, st);
        return band_pair(p->back, p->by, p->bx, mids, p->pitch2, outs, p->ny, nimg, st);
    }
    for (int k = 0; k < nimg; ++k) {
        const float *img = imgs[k];
        float *out = outs[k];
        float *mid_img = back_f ? p->t2p : p->t2;
        const int mid_pitch = back_f ? p->pitch2 : p->fy.n_out;
        float *front_out = psf ? mid_img : out;
        if (front_f) {
            if (int rc = band_pair(p->front, p->fy, p->fx, &img, p->Ny, &front_out, psf ? mid_pitch : p->fy.n_out, 1, st)) return rc;
        } else {
            // contiguous axis first (the only pass over the full-resolution image), then axis 0
            if (int rc = band_cols(p->fy, img, p->t1, p->Nx, st)) return rc;
            if (int rc = band_rows(p->fx, p->t1, front_out, p->fy.n_out, st)) return rc;
        }
        if (!psf) continue;
        // back operator (PSF + crop) at detector resolution
        if (back_f) {
            if (psf_stencil_ok(p->by, p->bx)) {
                if (int rc = psf_tile(p->by, p->bx, &mid_img, mid_pitch, &out, p->ny, 1, st)) return rc;
            } else if (int rc = band_pair(p->back, p->by, p->bx, &mid_img, mid_pitch, &out, p->ny, 1, st)) {
                return rc;
            }
            continue;
        }
        if (int rc = band_cols(p->by, p->t2, p->t3, p->fx.n_out, st)) return rc;
        if (int rc = band_rows(p->bx, p->t3, out, p->ny, st)) return rc;
    }
    return 0;
}

int psx_detect_f32(psx_detector_plan *p, const float *img, float *out, void *stream) {
    PSX_REQUIRE(p && img && out, "psx_detect_f32: null pointer");
    return detect_impl(p, &img, &out, 1, (hipStream_t)stream);
}

int psx_detect_multi_f32(psx_detector_plan *p, const float *const *imgs, float *const *outs, int nimg, void *stream) {
    PSX_REQUIRE(p && imgs && outs, "psx_detect_multi_f32: null pointer");
    PSX_REQUIRE(nimg >= 1 && nimg <= PSX_MAX_DETECT, "psx_detect_multi_f32: %d images (1..%d)", nimg, PSX_MAX_DETECT);
    for (int k = 0; k < nimg; ++k) {
        PSX_REQUIRE(imgs[k] && outs[k], "psx_detect_multi_f32: null image %d", k);
        for (int j = 0; j < k; ++j) PSX_REQUIRE(outs[j] != outs[k], "psx_detect_multi_f32: outputs %d and %d are the same image", j, k);
    }
    return detect_impl(p, imgs, outs, nimg, (hipStream_t)stream);
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
    PoissonImgs im = {};
    im.img[0] = out;
    im.seed[0] = seed;
    PSX_TIMED("k_poisson", (hipStream_t)stream, k_poisson<<<dim3(ew_grid(n, 256, 4), 1), 256, 0, (hipStream_t)stream>>>(im, lam == out ? nullptr : lam, n));
    return launch_check("k_poisson");
}

int psx_poisson_multi_f32(float *const *imgs, const uint64_t *seeds, int nimg, int64_t n, void *stream) {
    PSX_REQUIRE(imgs && seeds && nimg >= 1 && nimg <= PSX_MAX_POISSON && n >= 0, "psx_poisson_multi_f32: bad image list");
    PoissonImgs im = {};
    for (int i = 0; i < nimg; ++i) {
        PSX_REQUIRE(imgs[i] != nullptr, "psx_poisson_multi_f32: null image %d", i);
        im.img[i] = imgs[i];
        im.seed[i] = seeds[i];
    }
    if (n == 0) return 0;
    PSX_TIMED("k_poisson", (hipStream_t)stream, k_poisson<<<dim3(ew_grid(n, 256, 4), nimg), 256, 0, (hipStream_t)stream>>>(im, nullptr, n));
    return launch_check("k_poisson");
}

int psx_pack_counts_u16(const float *src, uint16_t *dst, int64_t n, int64_t index0, int32_t *exc, int32_t *exc_count, int cap,
                        int *overflow, void *stream) {
    PSX_REQUIRE(src && dst && exc && exc_count && overflow && n >= 0 && cap >= 0 && index0 >= 0,
                "psx_pack_counts_u16: null pointer or negative size");
    PSX_REQUIRE(index0 + n <= (int64_t)0xffffffffLL, "psx_pack_counts_u16: the exception table holds 32-bit pixel indices");
    if (n == 0) return 0;
    PSX_TIMED("k_pack_u16", (hipStream_t)stream, k_pack_u16<<<ew_grid(n, 256, 8), 256, 0, (hipStream_t)stream>>>(src, dst, n, index0, exc, exc_count, cap, overflow));
    return launch_check("k_pack_u16");
}

int psx_unpack_counts_u16(const uint16_t *src, float *dst, int64_t n, const int32_t *exc, const int32_t *exc_count, int cap,
                          void *stream) {
    PSX_REQUIRE(src && dst && n >= 0 && cap >= 0 && (cap == 0 || (exc && exc_count)), "psx_unpack_counts_u16: null pointer or negative size");
    if (n == 0) return 0;
    PSX_TIMED("k_unpack_u16", (hipStream_t)stream, k_unpack_u16<<<ew_grid(n, 256, 8), 256, 0, (hipStream_t)stream>>>(src, dst, n));
    if (cap > 0)
        PSX_TIMED("k_unpack_exceptions", (hipStream_t)stream, k_unpack_exceptions<<<ew_grid(cap, 256, 4), 256, 0, (hipStream_t)stream>>>(dst, n, exc, exc_count, cap));
    return launch_check("k_unpack_u16");
}

}  // extern "C"
