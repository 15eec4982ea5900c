// membrane.hip -- speckle-membrane thickness synthesis: sum of sphere chords.
//
// Replaces the interpreted triple loop of getMembraneSegmentedFromFile (Samples/getMembraneFromFile.py:143-159), which
// the reference re-runs for every membrane position (main.py:64-65) and which dominates its wall-clock.  The scatter
// over spheres becomes a gather per 32x32 tile: the sphere windows are binned by tile on the host (O(#spheres), the list
// is a host array anyway), each workgroup walks its own list with the sphere parameters staged in LDS, and every pixel
// accumulates its chords in float64 -- no atomics, one coalesced store per pixel.  The reference's window rule is kept
// literally: a sphere touches [x-radInt, x+radInt) x [y-radInt, y+radInt), radInt = floor(r)+1, x = round-half-even(xf).
#include <algorithm>
#include <vector>

#include "common.hpp"

using namespace psx;

namespace {

constexpr int MT = 32;   // tile side
// Cell side of the plan's sphere bins (a build-time A/B: tools/ab_membrane.sh).  A tile's candidates are the spheres of every
// cell its window (tile + the largest sphere window on either side) meets: with 32-pixel cells that is 96 x 96 pixels of cells
// for a 46-pixel window at the bench's sphere size -- ~126 candidates for ~26 hits.  Finer cells stage fewer candidates but make
// more (layer, cell row) jobs, i.e. more staging rounds with their barriers: k_membrane per 4096^2 position 0.084 ms (32-pixel
// cells) / 0.098 (16) / 0.111 (8) -- gpurun_out/r5s13, two rounds on one box.  The candidates are not what costs; 32 stays.
#ifndef PSX_MEMBRANE_CELL
#define PSX_MEMBRANE_CELL 32
#endif
constexpr int MC = PSX_MEMBRANE_CELL;

struct Sphere {
    double xf, yf, r;
    int xi, yi, radInt, pad;
};

__global__ __launch_bounds__(256) void k_membrane(const Sphere *__restrict__ spheres, const int *__restrict__ offsets,
                                                  const int *__restrict__ ids, float *__restrict__ out, int dimX,
                                                  int dimY, int margin, int tiles_y, double scale, int accumulate) {
    __shared__ Sphere sh[64];
    const int tile = blockIdx.x, t0 = (tile / tiles_y) * MT, c0 = (tile % tiles_y) * MT;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 columns x 8 rows; each thread owns 4 rows
    const int beg = offsets[tile], end = offsets[tile + 1];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int py = c0 + tx + margin;                           // coordinates on the margin-extended grid
    for (int base = beg; base < end; base += 64) {
        const int cnt = min(64, end - base);
        __syncthreads();
        if (threadIdx.x < cnt) sh[threadIdx.x] = spheres[ids[base + threadIdx.x]];
        __syncthreads();
        for (int s = 0; s < cnt; ++s) {
            const Sphere sp = sh[s];
            const int jj = py - sp.yi;
            if (jj < -sp.radInt || jj >= sp.radInt) continue;
            const double dy = (double)py - sp.yf;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int px = t0 + ty + 8 * k + margin;
                const int ii = px - sp.xi;
                if (ii >= -sp.radInt && ii < sp.radInt) {
                    const double dx = (double)px - sp.xf;
                    // getMembraneFromFile.py:157-159 takes dist = sqrt(dx^2 + dy^2), tests dist < r and adds 2 sqrt(r^2 - dist^2);
                    // comparing the squares saves one of the two float64 square roots and moves the chord by one rounding
                    // of dist^2 (below 1e-9 of the membrane thickness, also at a sphere's rim)
                    const double d2 = dx * dx + dy * dy, r2 = sp.r * sp.r;
                    if (d2 < r2) acc[k] += 2.0 * sqrt(r2 - d2);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = t0 + ty + 8 * k, j = c0 + tx;
        if (i < dimX && j < dimY) {
            const float v = (float)(acc[k] * scale);
            const int64_t p = (int64_t)i * dimY + j;
            out[p] = accumulate ? out[p] + v : v;
        }
    }
}

// ---- plan variant: the sphere list lives on the GPU, binned ONCE by MC-pixel cell of its own frame -------------------
// getMembraneSegmentedFromFile re-places the same (scaled, stitched) list for every membrane position and layer with a
// new integer offset (getMembraneFromFile.py:139-142).  Binning by tile on the host per call -- what psx_membrane_f32
// does -- then costs 6 ms per layer at 4096^2 against 0.1 ms for the kernel.  Here a tile finds its spheres itself: its
// pixels, shifted by the offset, overlap a few cells of the list frame, whose lists are contiguous per cell row.
struct CellSphere {
    double x, y, r;     // pixels of the list frame (par / pixSize)
};

// ---- all layers of a position in ONE launch, sphere-centric ------------------------------------------------------------
// A pixel-centric tile (k_membrane above, one or four pixels per lane walking the tile's spheres) spends its float64 chord
// sequences at ~10-20 % lane occupancy: a sphere is ~10 pixels wide, a wave's pixels are 64.  Here a workgroup owns a
// 32 x 32 tile of 64-bit fixed-point accumulators in LDS (2^-36 pixel; integer adds commute, so the map is bitwise
// reproducible whatever the order) and 16 lanes take ONE sphere: a lane per window column, a loop over the window rows
// clipped to the tile, ds_add_u64 per chord.  Staging is by "job" = (layer, cell row of the list frame): a half-wave
// compacts the spheres of one job whose window meets the tile, eight jobs per round -- both layers of the usual two-layer
// membrane in one round -- into one flat list.  The layers of a position (same list, one integer offset each,
// getMembraneFromFile.py:139-142) are summed before the single store, and the uniform support map is written by the same launch.
#ifndef PSX_ML_OFF
#define PSX_ML_OFF 0          // timing experiments: 1 no splat, 2 no stores, 4 no search for spheres at all
#endif
// Tile (rows x columns) and workgroup size of the layers kernel -- build-time A/B, tools/ab_membrane_tile.sh.  Timing experiments
// on the 32 x 32 x 256 form (PSX_ML_OFF; gpurun_out/r5s63, kernel by event pairs, 4096^2, two layers of 15 um spheres): whole kernel
// 0.0733 ms; without the splat 0.0310; without the stores 0.0685; launch + zeroing alone 0.0142 -- the splat is 58 % of it, and a
// sphere whose window straddles a tile border is splatted once per tile it touches (2.07 tiles per sphere at 32 x 32, 1.75 at
// 32 x 64).  Measured (gpurun_out/r5s64): 32x32x256 0.0733, 32x64x256 **0.0596**, 64x64x512 0.0610, 32x128x512 0.0656, 32x64x512
// 0.0777, 64x32x512 0.0796, 64x64x256 0.0738 (three workgroups per CU), 128x32x256 0.0912.
#ifndef PSX_ML_TX
#define PSX_ML_TX 32
#endif
#ifndef PSX_ML_TY
#define PSX_ML_TY 64
#endif
constexpr int MLX = PSX_ML_TX, MLY = PSX_ML_TY;   // rows (axis 0) x columns
constexpr int ML_MAX = 8;        // layers per launch
#ifndef PSX_ML_THREADS
#define PSX_ML_THREADS 256
#endif
constexpr int MLT = PSX_ML_THREADS;           // threads per workgroup
constexpr int ML_SEGS = MLT / 32;             // jobs staged per round: one per half-wave
constexpr int ML_CAP = 32;       // spheres per job and batch
constexpr int ML_FRAC = 36;      // fractional bits of the accumulators: chords below 2^14 pixels, sums below 2^27

struct LayerArgs {
    int offx[ML_MAX], offy[ML_MAX];
    int nlayers, rows;           // rows: upper bound of the cell rows one layer's tile window can meet
};

struct StagedSphere {
    double xf, yf, r2;
    int xi, yi, radInt, pad;
};

// sqrt of a positive float64 from the float32 reciprocal square root and one Newton step in float64 (relative error
// ~2e-14; the library's correctly rounded sqrt costs three times the instructions and the result is stored as float32)
template <bool CLAMP>
__device__ __forceinline__ double chord_sqrt(double t) {
    const float tf = CLAMP ? fmaxf((float)t, 1e-30f) : (float)t;
    const double y = (double)__builtin_amdgcn_rsqf(tf);
    const double s0 = t * y, h0 = 0.5 * y;
    const double r = fma(-s0, h0, 0.5);
    return fma(s0, r, s0);
}

__global__ __launch_bounds__(MLT) void k_membrane_layers(const CellSphere *__restrict__ spheres,
                                                         const int *__restrict__ cell_off, int ncx, int ncy, double x0,
                                                         double y0, int rmax_int, LayerArgs la, float *__restrict__ out,
                                                         float *__restrict__ support, float support_value, int dimX,
                                                         int dimY, int margin, int margin2, int tiles_y, double scale,
                                                         int accumulate) {
    __shared__ unsigned long long acc[MLX * MLY];
    __shared__ StagedSphere flat[ML_SEGS * ML_CAP];
    __shared__ int shcnt[ML_SEGS], shrem[ML_SEGS];
    const int tile = blockIdx.x, t0 = (tile / tiles_y) * MLX, c0 = (tile % tiles_y) * MLY;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, hl = lane & 31;
    const int seg = 2 * wave + half;                 // the job slot this half-wave stages
    const int grp = tid >> 4, col = tid & 15;        // splat: 16 lanes per sphere
    const int njobs = la.nlayers * la.rows;
    const int tx0 = t0 + margin, ty0 = c0 + margin;  // the tile on the margin-extended grid
    for (int k = tid; k < MLX * MLY; k += MLT) acc[k] = 0ull;
    for (int j0 = 0; j0 < ((PSX_ML_OFF & 4) ? 0 : njobs); j0 += ML_SEGS) {
        // this half-wave's job: layer l, cell row cx0(l) + g, the spheres of cells [cy0, cy1] of that row
        const int job = j0 + seg, l = min(job / la.rows, ML_MAX - 1), g = job % la.rows;
        const int offx = la.offx[l], offy = la.offy[l];
        int beg = 0, end = 0;
        if (job < njobs) {
            // cells whose spheres can reach this tile: |centre - pixel| <= radInt + 1/2 on each axis
            const double xlo = (double)(offx + tx0 - rmax_int - 1) - x0, xhi = (double)(offx + tx0 + MLX + rmax_int + 1) - x0;
            const double ylo = (double)(offy + ty0 - rmax_int - 1) - y0, yhi = (double)(offy + ty0 + MLY + rmax_int + 1) - y0;
            const int cx = max(0, (int)floor(xlo / MC)) + g, cx1 = min(ncx - 1, (int)floor(xhi / MC));
            const int cy0 = max(0, (int)floor(ylo / MC)), cy1 = min(ncy - 1, (int)floor(yhi / MC));
            if (cx <= cx1 && cy0 <= cy1) {
                beg = cell_off[cx * ncy + cy0];
                end = cell_off[cx * ncy + cy1 + 1];
            }
        }
        bool more = true;
        for (int b0 = 0; more; b0 += ML_CAP) {
            StagedSphere sp = {};
            bool hit = false;
            if (beg + b0 + hl < end) {
                const CellSphere cs = spheres[beg + b0 + hl];
                sp.xf = cs.x - (double)offx;                                       // getMembraneFromFile.py:141-142
                sp.yf = cs.y - (double)offy;
                sp.r2 = cs.r * cs.r;
                sp.xi = (int)rint(sp.xf);                                          // np.round: half to even
                sp.yi = (int)rint(sp.yf);
                const bool ok = cs.r > 0.0 && margin2 < sp.xi && sp.xi < dimX + margin + margin2 && margin2 < sp.yi &&
                                sp.yi < dimY + margin + margin2;                   // :152
                sp.radInt = (int)floor(cs.r) + 1;
                // window [xi - radInt, xi + radInt) against the tile's pixels [tx0, tx0 + MLX)
                hit = ok && sp.xi + sp.radInt > tx0 && sp.xi - sp.radInt < tx0 + MLX && sp.yi + sp.radInt > ty0 &&
                      sp.yi - sp.radInt < ty0 + MLY;
            }
            const unsigned mh = (unsigned)(__ballot(hit) >> (32 * half));
            __syncthreads();                         // the previous batch has been splatted (and acc is zeroed)
            if (hl == 0) {
                shcnt[seg] = __popc(mh);
                shrem[seg] = max(0, end - beg - b0 - ML_CAP);
            }
            __syncthreads();
            int base = 0, total = 0;
            more = false;
#pragma unroll
            for (int q = 0; q < ML_SEGS; ++q) {
                const int c = shcnt[q];
                base += q < seg ? c : 0;
                total += c;
                more = more || shrem[q] > 0;
            }
            if (hit) flat[base + __popc(mh & ((1u << hl) - 1u))] = sp;
            __syncthreads();
            for (int s = grp; s < ((PSX_ML_OFF & 1) ? 0 : total); s += MLT / 16) {     // (PSX_ML_OFF: timing experiments, wrong images)
                const StagedSphere c = flat[s];
                // window rows clipped to the tile; columns: one per lane of the group, 16 at a time
                const int i_lo = max(c.xi - c.radInt, tx0), i_hi = min(c.xi + c.radInt, tx0 + MLX);
                for (int pyc = c.yi - c.radInt + col; pyc < c.yi + c.radInt; pyc += 16) {
                    const int j = pyc - ty0;
                    if (j < 0 || j >= MLY) continue;
                    // getMembraneFromFile.py:157-159 takes dist = sqrt(dx^2 + dy^2), tests dist < r and adds
                    // 2 sqrt(r^2 - dist^2); comparing the squares saves one of the two square roots and moves the chord by
                    // one rounding of dist^2 (below 1e-9 of the membrane thickness, also at a sphere's rim)
                    // Build-time A/B (tools/ab_membrane_inc.sh): PSX_MEMBRANE_SPLAT 0 = the row offset converted per row, lengths in
                    // pixels; 1 = the row offset by repeated addition (exact: both operands sit on one grid); 2 = also every length
                    // in the accumulator's unit of 2^-ML_FRAC pixel (powers of two: the same bits), so that the chord leaves the
                    // square root already in fixed point: doubling and rounding are one fma with an inline constant.
#ifndef PSX_MEMBRANE_SPLAT
#define PSX_MEMBRANE_SPLAT 2
#endif
                    constexpr double U = PSX_MEMBRANE_SPLAT == 2 ? (double)(1ull << ML_FRAC) : 1.0;
                    const double dy = ((double)pyc - c.yf) * U;
                    const double dy2 = dy * dy;
                    const double r2 = c.r2 * (U * U);
                    double dxr = ((double)i_lo - c.xf) * U;
                    for (int pxr = i_lo; pxr < i_hi; ++pxr, dxr += U) {
                        const double dx = PSX_MEMBRANE_SPLAT ? dxr : (double)pxr - c.xf;
                        const double d2 = fma(dx, dx, dy2);
                        if (d2 < r2) {
                            // 2 sqrt(.) in units of 2^-ML_FRAC pixel, rounded to nearest through the 2^52 + 2^51 offset
                            // (scaled: r2 - d2 > 0 is at least an ulp of r2, far above the float32 range's floor: no clamp)
                            const double v = PSX_MEMBRANE_SPLAT == 2
                                                 ? fma(chord_sqrt<false>(r2 - d2), 2.0, 6755399441055744.0)
                                                 : fma(chord_sqrt<true>(r2 - d2), (double)(2ull << ML_FRAC), 6755399441055744.0);
                            atomicAdd(&acc[(pxr - tx0) * MLY + j], (unsigned long long)(__double_as_longlong(v) - 0x4338000000000000ll));
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    const double unit = scale / (double)(1ull << ML_FRAC);
    for (int k = tid; k < MLX * MLY; k += MLT) {
        const int i = t0 + k / MLY, j = c0 + k % MLY;
        if (i < dimX && j < dimY) {
            const float v = (float)((double)acc[k] * unit);
            const int64_t p = (int64_t)i * dimY + j;
            if ((PSX_ML_OFF & 2) && v >= 0.f) continue;
            out[p] = accumulate ? out[p] + v : v;
            if (support) support[p] = support_value;
        }
    }
}

}  // namespace

struct psx_membrane_plan {
    CellSphere *spheres = nullptr;   // sorted by cell (row-major cells), list order inside a cell
    int *cell_off = nullptr;         // [ncx*ncy + 1]
    int ncx = 0, ncy = 0, rmax_int = 1;
    double x0 = 0.0, y0 = 0.0;
    int64_t n = 0;
};

extern "C" {

int psx_membrane_plan_create(const double *x, const double *y, const double *r, int64_t n, psx_membrane_plan **plan) {
    PSX_REQUIRE(plan != nullptr, "psx_membrane_plan_create: null plan pointer");
    *plan = nullptr;
    PSX_REQUIRE(n >= 0 && (n == 0 || (x && y && r)), "psx_membrane_plan_create: null sphere arrays");
    psx_membrane_plan *p = new psx_membrane_plan();
    double xmin = 0, xmax = 0, ymin = 0, ymax = 0, rmax = 0;
    bool any = false;
    for (int64_t s = 0; s < n; ++s) {
        if (!(r[s] > 0.0) || !std::isfinite(x[s]) || !std::isfinite(y[s]) || !std::isfinite(r[s])) continue;
        if (!any) { xmin = xmax = x[s]; ymin = ymax = y[s]; any = true; }
        xmin = std::min(xmin, x[s]); xmax = std::max(xmax, x[s]);
        ymin = std::min(ymin, y[s]); ymax = std::max(ymax, y[s]);
        rmax = std::max(rmax, r[s]);
    }
    p->x0 = std::floor(xmin); p->y0 = std::floor(ymin);
    const double ex = xmax - p->x0, ey = ymax - p->y0;
    if (!(ex / MC < 60000.0 && ey / MC < 60000.0 && (ex / MC + 1) * (ey / MC + 1) < 4.0e8 && rmax < 1.0e6)) {
        delete p;
        return fail(PSX_E_ARG, "psx_membrane_plan_create: sphere list spans %.3g x %.3g pixels (radius up to %.3g)", ex, ey, rmax);
    }
    p->ncx = (int)(ex / MC) + 1; p->ncy = (int)(ey / MC) + 1;
    p->rmax_int = (int)std::floor(rmax) + 1;
    const size_t nc = (size_t)p->ncx * p->ncy;
    std::vector<int> off(nc + 1, 0);
    std::vector<int> cell(n > 0 ? n : 1, -1);
    for (int64_t s = 0; s < n; ++s) {
        if (!(r[s] > 0.0) || !std::isfinite(x[s]) || !std::isfinite(y[s]) || !std::isfinite(r[s])) continue;
        const int cx = (int)((x[s] - p->x0) / MC), cy = (int)((y[s] - p->y0) / MC);
        cell[s] = cx * p->ncy + cy;
        off[cell[s] + 1]++;
    }
    for (size_t c = 0; c < nc; ++c) off[c + 1] += off[c];
    std::vector<CellSphere> sorted(off[nc] ? off[nc] : 1);
    std::vector<int> cursor(off.begin(), off.end() - 1);
    for (int64_t s = 0; s < n; ++s)
        if (cell[s] >= 0) sorted[cursor[cell[s]]++] = CellSphere{x[s], y[s], r[s]};
    p->n = off[nc];
    hipError_t e = hipMalloc((void **)&p->spheres, sizeof(CellSphere) * sorted.size());
    if (e == hipSuccess) e = hipMalloc((void **)&p->cell_off, sizeof(int) * off.size());
    if (e == hipSuccess) e = hipMemcpy(p->spheres, sorted.data(), sizeof(CellSphere) * sorted.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->cell_off, off.data(), sizeof(int) * off.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        psx_membrane_plan_destroy(p);
        return fail((int)e, "psx_membrane_plan_create: %s", hipGetErrorString(e));
    }
    *plan = p;
    return 0;
}

int psx_membrane_plan_destroy(psx_membrane_plan *p) {
    if (!p) return 0;
    (void)hipFree(p->spheres);
    (void)hipFree(p->cell_off);
    delete p;
    return 0;
}

int psx_membrane_layers_f32(psx_membrane_plan *p, int nlayers, const int *offx, const int *offy, int dimX, int dimY, int margin,
                            int margin2, double scale, int accumulate, float *out, float *support, float support_value,
                            void *stream) {
    PSX_REQUIRE(p != nullptr && out != nullptr && dimX > 0 && dimY > 0 && margin >= 0, "psx_membrane_layers_f32: bad argument");
    PSX_REQUIRE(nlayers >= 0 && (nlayers == 0 || (offx && offy)), "psx_membrane_layers_f32: null offsets");
    hipStream_t st = (hipStream_t)stream;
    const int tiles_x = (int)cdiv(dimX, MLX), tiles_y = (int)cdiv(dimY, MLY);
    // ML_MAX layers per launch; later launches add to the map (and leave the support alone)
    for (int l0 = 0; l0 == 0 || l0 < nlayers; l0 += ML_MAX) {
        LayerArgs la = {};
        la.nlayers = std::min(ML_MAX, nlayers - l0);
        la.rows = (MLX + 2 * (p->rmax_int + 1)) / MC + 2;      // a window of that many pixels meets at most this many cells
        for (int l = 0; l < la.nlayers; ++l) {
            la.offx[l] = offx[l0 + l];
            la.offy[l] = offy[l0 + l];
        }
        PSX_TIMED("k_membrane", st, k_membrane_layers<<<tiles_x * tiles_y, MLT, 0, st>>>(
                      p->spheres, p->cell_off, p->ncx, p->ncy, p->x0, p->y0, p->rmax_int, la, out, l0 == 0 ? support : nullptr,
                      support_value, dimX, dimY, margin, margin2, tiles_y, scale, (accumulate || l0 > 0) ? 1 : 0));
        if (int rc = launch_check("k_membrane")) return rc;
    }
    return 0;
}

int psx_membrane_layer_f32(psx_membrane_plan *p, int offx, int offy, int dimX, int dimY, int margin, int margin2,
                           double scale, int accumulate, float *out, void *stream) {
    return psx_membrane_layers_f32(p, 1, &offx, &offy, dimX, dimY, margin, margin2, scale, accumulate, out, nullptr, 0.f, stream);
}

// xf, yf, rad: HOST arrays (pixels of the margin-extended grid).  out: DEVICE [dimX][dimY] float32.
int psx_membrane_f32(const double *xf, const double *yf, const double *rad, int64_t n, int dimX, int dimY, int margin,
                     int margin2, double scale, int accumulate, float *out, void *stream) {
    PSX_REQUIRE(out != nullptr && dimX > 0 && dimY > 0 && margin >= 0 && n >= 0, "psx_membrane_f32: bad argument");
    PSX_REQUIRE(n == 0 || (xf && yf && rad), "psx_membrane_f32: null sphere arrays");
    hipStream_t st = (hipStream_t)stream;
    const int tiles_x = (int)cdiv(dimX, MT), tiles_y = (int)cdiv(dimY, MT), nt = tiles_x * tiles_y;
    std::vector<Sphere> sph;
    std::vector<int> counts(nt + 1, 0);
    std::vector<std::pair<int, int>> pairs;   // (tile, sphere)
    for (int64_t s = 0; s < n; ++s) {
        const double r = rad[s];
        if (!(r > 0.0) || !std::isfinite(xf[s]) || !std::isfinite(yf[s])) continue;
        const int xi = (int)std::nearbyint(xf[s]), yi = (int)std::nearbyint(yf[s]);      // np.round: half to even
        if (!(margin2 < xi && xi < dimX + margin + margin2 && margin2 < yi && yi < dimY + margin + margin2))
            continue;                                                                      // getMembraneFromFile.py:152
        const int radInt = (int)std::floor(r) + 1;
        // window on the cropped grid
        const int x0 = std::max(0, xi - radInt - margin), x1 = std::min(dimX - 1, xi + radInt - 1 - margin);
        const int y0 = std::max(0, yi - radInt - margin), y1 = std::min(dimY - 1, yi + radInt - 1 - margin);
        if (x0 > x1 || y0 > y1) continue;
        const int id = (int)sph.size();
        sph.push_back(Sphere{xf[s], yf[s], r, xi, yi, radInt, 0});
        for (int tx = x0 / MT; tx <= x1 / MT; ++tx)
            for (int ty = y0 / MT; ty <= y1 / MT; ++ty) pairs.emplace_back(tx * tiles_y + ty, id);
    }
    for (auto &pr : pairs) counts[pr.first + 1]++;
    for (int t = 0; t < nt; ++t) counts[t + 1] += counts[t];
    std::vector<int> ids(pairs.size() ? pairs.size() : 1), cursor(counts.begin(), counts.end() - 1);
    for (auto &pr : pairs) ids[cursor[pr.first]++] = pr.second;      // sphere order inside a tile = list order
    if (sph.empty()) sph.push_back(Sphere{0, 0, 0, 0, 0, 0, 0});
    Sphere *d_sph = nullptr;
    int *d_off = nullptr, *d_ids = nullptr;
    PSX_HIP(hipMallocAsync((void **)&d_sph, sizeof(Sphere) * sph.size(), st));
    PSX_HIP(hipMallocAsync((void **)&d_off, sizeof(int) * counts.size(), st));
    PSX_HIP(hipMallocAsync((void **)&d_ids, sizeof(int) * ids.size(), st));
    PSX_HIP(hipMemcpyAsync(d_sph, sph.data(), sizeof(Sphere) * sph.size(), hipMemcpyHostToDevice, st));
    PSX_HIP(hipMemcpyAsync(d_off, counts.data(), sizeof(int) * counts.size(), hipMemcpyHostToDevice, st));
    PSX_HIP(hipMemcpyAsync(d_ids, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice, st));
    PSX_TIMED("k_membrane", st, k_membrane<<<nt, 256, 0, st>>>(d_sph, d_off, d_ids, out, dimX, dimY, margin, tiles_y,
                                                                scale, accumulate));
    const int rc = launch_check("k_membrane");
    PSX_HIP(hipStreamSynchronize(st));       // the host staging vectors go out of scope below
    (void)hipFreeAsync(d_sph, st);
    (void)hipFreeAsync(d_off, st);
    (void)hipFreeAsync(d_ids, st);
    return rc;
}

}  // extern "C"
