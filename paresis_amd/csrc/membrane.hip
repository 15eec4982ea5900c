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
                    const double dist = sqrt(dx * dx + dy * dy);                 // getMembraneFromFile.py:157
                    if (dist < sp.r) acc[k] += 2.0 * sqrt(sp.r * sp.r - dist * dist);   // :159
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

}  // namespace

extern "C" {

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
