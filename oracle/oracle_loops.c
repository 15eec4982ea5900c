/*
 * oracle_loops.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Scalar fp64 C restatement of the two loops the reference JIT-compiles with Numba:
 *   - the bilinear intensity scatter  fastloopNumba  (CodePython/refractionFileNumba2.py:198-263,
 *     identical to CodePython/refractionFileNumba.py:70-135), and
 *   - the block-sum detector resampling  resize  (CodePython/Detector.py:185-198).
 * Raster order, one thread, same branch structure as the reference so that border behaviour (a base cell must be
 * inside the grid; the three neighbour deposits are made only if BOTH neighbour indices are inside) is identical.
 * Pinned by tests/golden/refraction.npz and tests/golden/scalars.npz (tests/test_oracle_golden.py).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>

/* RF2:217-263.  I, Dx, Dy, I2 are [Nx][Ny] row-major; I2 is accumulated into (RF2:77 passes zeros). */
void oracle_fastloop(int64_t Nx, int64_t Ny, const double *I, double *I2, const double *Dx, const double *Dy)
{
    for (int64_t i = 0; i < Nx; ++i) {
        for (int64_t j = 0; j < Ny; ++j) {
            const double Iij = I[i * Ny + j];
            double dx = Dx[i * Ny + j];
            double dy = Dy[i * Ny + j];
            if (dx == 0.0 && dy == 0.0) {               /* RF2:222-224 */
                I2[i * Ny + j] += Iij;
                continue;
            }
            int64_t inew = i, jnew = j;
            if (fabs(dx) > 1.0) {                        /* RF2:228-230 */
                const double f = floor(dx);
                inew = i + (int64_t)f;
                dx -= f;
            }
            if (fabs(dy) > 1.0) {                        /* RF2:231-233 */
                const double f = floor(dy);
                jnew = j + (int64_t)f;
                dy -= f;
            }
            if (inew < 0 || inew >= Nx || jnew < 0 || jnew >= Ny)   /* RF2:235-236 */
                continue;
            const double ax = fabs(dx), ay = fabs(dy);
            I2[inew * Ny + jnew] += Iij * (1.0 - ax) * (1.0 - ay);  /* RF2:237 */
            /* RF2:238-262: the x-neighbour is inew+1 for dx>=0 (needs inew<Nx-1) else inew-1 (needs inew>0);
             * likewise in y; all three neighbour deposits happen only when both neighbours exist. */
            int64_t ix, jy;
            if (dx >= 0.0) { if (inew >= Nx - 1) continue; ix = inew + 1; }
            else           { if (inew <= 0)      continue; ix = inew - 1; }
            if (dy >= 0.0) { if (jnew >= Ny - 1) continue; jy = jnew + 1; }
            else           { if (jnew <= 0)      continue; jy = jnew - 1; }
            I2[ix * Ny + jnew] += Iij * ax * (1.0 - ay);
            I2[ix * Ny + jy]   += Iij * ax * ay;
            I2[inew * Ny + jy] += Iij * (1.0 - ax) * ay;
        }
    }
}

/* DET:185-198.  Caller handles the identity case (DET:188-189).  s = int(Nx/sizeX) on BOTH axes (DET:192);
 * numpy slicing clips at the array end, reproduced by the min() below. */
void oracle_resize(int64_t Nx, int64_t Ny, const double *img, int64_t sizeX, int64_t sizeY, double *out)
{
    const int64_t s = Nx / sizeX;
    for (int64_t x0 = 0; x0 < sizeX; ++x0) {
        for (int64_t y0 = 0; y0 < sizeY; ++y0) {
            int64_t xa = x0 * s, xb = xa + s, ya = y0 * s, yb = ya + s;
            if (xb > Nx) xb = Nx;
            if (yb > Ny) yb = Ny;
            double acc = 0.0;
            for (int64_t x = xa; x < xb; ++x)
                for (int64_t y = ya; y < yb; ++y)
                    acc += img[x * Ny + y];
            out[x0 * sizeY + y0] = acc;
        }
    }
}

/* Sphere splat of getMembraneSegmentedFromFile (Samples/getMembraneFromFile.py:143-159) for one layer.
 * membrane is [Mx][My] (study grid + margin on every side); xf, yf, rad are in pixels of that grid.
 * A sphere is drawn when margin2 < round(x) < dimX+margin+margin2 (same in y); its window is
 * [x-radInt, x+radInt) x [y-radInt, y+radInt) with radInt = floor(rad)+1 -- note the half-open upper end. */
void oracle_membrane_splat(int64_t n, const double *xf, const double *yf, const double *rad, int64_t dimX, int64_t dimY,
                           int64_t margin, int64_t margin2, double *membrane)
{
    const int64_t My = dimY + 2 * margin;
    for (int64_t i = 0; i < n; ++i) {
        const double r = rad[i];
        const int64_t radInt = (int64_t)floor(r) + 1;
        const int64_t x = (int64_t)rint(xf[i]), y = (int64_t)rint(yf[i]);    /* np.round: half to even */
        if (!(margin2 < x && x < dimX + margin + margin2 && margin2 < y && y < dimY + margin + margin2)) continue;
        for (int64_t ii = -radInt; ii < radInt; ++ii)
            for (int64_t jj = -radInt; jj < radInt; ++jj) {
                const double dx = (double)(ii + x) - xf[i], dy = (double)(jj + y) - yf[i];
                const double dist = sqrt(dx * dx + dy * dy);
                if (dist < r) membrane[(x + ii) * My + (y + jj)] += 2.0 * sqrt(r * r - dist * dist);
            }
    }
}
