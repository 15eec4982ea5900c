/*
 * cpu_baseline.cpp -- TEST / BENCH INFRASTRUCTURE, not product code.
 *
 * C++17 + OpenMP float64 restatement of the array work of the hot path, used as the CPU baseline that bench.py times next
 * to the GPU numbers and golden-checked like the Python oracle
 * (tests/test_oracle_golden.py).  Same algorithm and operation order as the reference; `nthreads` = 1 is the stand-in for
 * the reference itself (numpy.fft and a Numba @jit without parallel=True are single-threaded), `nthreads` = all cores the
 * fair CPU ceiling.  The 2-D FFTs are NOT here: the baseline calls pocketfft through scipy.fft (workers = threads) -- the
 * same library the reference's numpy.fft wraps -- between cb_wave_pad and cb_chirp / cb_crop_abs2.
 *
 *   cb_refraction   AnalyticalSample.setWaveRT (Sample.py:285-351, scalar dark field) + fastRefraction
 *                   (refractionFileNumba2.py:25-86) + fastloopNumba (:198-263)
 *   cb_wave_pad     AnalyticalSample.setWave (Sample.py:248-282) + np.pad(.., 15, 'reflect') (Experiment.py:237)
 *   cb_chirp        x exp(-i z uv^2 / (2 k M)) with fftshift/ifftshift folded into the index (Experiment.py:243-250)
 *   cb_crop_abs2    crop (Experiment.py:251) + abs()**2 (Experiment.py:351-354); the global phase drops out
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <omp.h>

#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <memory>
#include <utility>
#include <vector>

typedef std::complex<double> cd;

static inline int reflect(int q, int n) {   /* np.pad 'reflect': mirror without repeating the edge sample */
    if (q < 0) q = -q;
    if (q >= n) q = 2 * (n - 1) - q;
    return q;
}

extern "C" {

int cb_max_threads(void) { return omp_get_max_threads(); }

/* psi = amp * prod_m exp((-i k delta_m - k beta_m) T_m), reflect-padded by `margin` into out[Px][Py] (complex128). */
void cb_wave_pad(const float *const *T, const double *delta, const double *beta, int nmat, int Nx, int Ny, double amp,
                 double k, int margin, cd *out, int nthreads) {
    const int Py = Ny + 2 * margin, Px = Nx + 2 * margin;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = 0; i < Nx; ++i) {
        cd *row = out + (int64_t)(i + margin) * Py + margin;
        for (int j = 0; j < Ny; ++j) {
            cd w(amp, 0.0);
            for (int m = 0; m < nmat; ++m) {                       /* SAM:279, material by material */
                const double t = (double)T[m][(int64_t)i * Ny + j];
                w = std::exp(cd(-k * beta[m] * t, -k * delta[m] * t)) * w;
            }
            row[j] = w;
        }
        for (int q = 0; q < margin; ++q) {                         /* left / right mirror of this row */
            row[-1 - q] = row[1 + q];
            row[Ny + q] = row[Ny - 2 - q];
        }
    }
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int p = 0; p < Px; ++p) {                                 /* top / bottom mirror rows */
        const int i = p - margin;
        if (i >= 0 && i < Nx) continue;
        std::memcpy(out + (int64_t)p * Py, out + (int64_t)(reflect(i, Nx) + margin) * Py, sizeof(cd) * Py);
    }
}

/* spec[i][j] *= exp(-i a (u_i^2 + v_j^2)),  a = z/(2kM),  u = (fftshift index - Px//2) * du_x  (EXP:243-250): spectrum
 * index i in FFT order sits at shifted position (i + Px//2) % Px, i.e. frequency number ((i + Px//2) % Px) - Px//2. */
void cb_chirp(cd *spec, int Px, int Py, double a, double du_x, double du_y, int nthreads) {
    std::vector<double> vy(Py);
    for (int j = 0; j < Py; ++j) {
        const double v = (double)(((j + Py / 2) % Py) - Py / 2) * du_y;
        vy[j] = v * v;
    }
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = 0; i < Px; ++i) {
        const double u = (double)(((i + Px / 2) % Px) - Px / 2) * du_x;
        const double u2 = u * u;
        cd *row = spec + (int64_t)i * Py;
        for (int j = 0; j < Py; ++j) {
            const double ph = -a * (u2 + vy[j]);
            row[j] *= cd(std::cos(ph), std::sin(ph));
        }
    }
}

void cb_crop_abs2(const cd *field, int Nx, int Ny, int margin, double *out, int nthreads) {
    const int Py = Ny + 2 * margin;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = 0; i < Nx; ++i) {
        const cd *row = field + (int64_t)(i + margin) * Py + margin;
        for (int j = 0; j < Ny; ++j) out[(int64_t)i * Ny + j] = std::norm(row[j]);
    }
}

/* np.pad(wave, margin, 'reflect') of a complex128 [Nx][Ny] field (EXP:237) */
void cb_pad_reflect(const cd *wave, int Nx, int Ny, int margin, cd *out, int nthreads) {
    const int Py = Ny + 2 * margin, Px = Nx + 2 * margin;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int p = 0; p < Px; ++p) {
        const cd *src = wave + (int64_t)reflect(p - margin, Nx) * Ny;
        cd *row = out + (int64_t)p * Py;
        for (int q = 0; q < Py; ++q) row[q] = src[reflect(q - margin, Ny)];
    }
}

/* setWaveRT (Sample.py:347-348, scalar dark field): I = I0 prod exp(-2 k beta T), phi = -sum k delta T */
void cb_transmit_rt(const float *const *T, const double *delta, const double *beta, int nmat, int Nx, int Ny, double I0,
                    double k_sample, double *I, double *phi, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t p = 0; p < (int64_t)Nx * Ny; ++p) {
        double Iv = I0, ph = 0.0;
        for (int m = 0; m < nmat; ++m) {
            const double t = (double)T[m][p];
            Iv = std::exp(-2 * k_sample * beta[m] * t) * Iv;                 /* SAM:347 */
            ph = ph - k_sample * delta[m] * t;                              /* SAM:348 */
        }
        I[p] = Iv;
        phi[p] = ph;
    }
}

/* fastRefraction v2 (refractionFileNumba2.py:25-86) on (I, phi) [Nx][Ny] float64.  out[Nx][Ny].  Returns 1 when the result
 * holds NaN / >1e50 (RF2:81-82), else 0.  With nthreads > 1 the raster loop runs over row bands (see below): the sums then
 * differ from raster order in the last bits. */
int cb_fast_refraction(const double *I, const double *phi, int Nx, int Ny, double k_refr, double z, double M, double pix_um,
                       double *out, int nthreads) {
    const int mg = 15, Px = Nx + 2 * mg, Py = Ny + 2 * mg;
    const double h = pix_um * 1e-6;
    /* uninitialised buffers, zeroed by the threads that will use them (first touch places the pages near their cores) */
    std::unique_ptr<double[]> Dxb(new double[(size_t)Px * Py]), Dyb(new double[(size_t)Px * Py]), Ipb(new double[(size_t)Px * Py]),
        I2b(new double[(size_t)Px * Py]);
    double *Dx = Dxb.get(), *Dy = Dyb.get(), *Ip = Ipb.get(), *I2 = I2b.get();
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int p = 0; p < Px; ++p)
        for (int q = 0; q < Py; ++q) {
            const int64_t e = (int64_t)p * Py + q;
            Dx[e] = 0.0; Dy[e] = 0.0; Ip[e] = 0.0; I2[e] = 0.0;
        }
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = 0; i < Nx; ++i) {
        for (int j = 0; j < Ny; ++j) {
            auto f = [&](int a, int b) { return phi[(int64_t)a * Ny + b]; };
            double gx, gy;                                                  /* np.gradient(phi, h, edge_order=2), RF2:54 */
            /* numpy's own operation order: interior (f[i+1] - f[i-1]) / (2 h); edges a f0 + b f1 + c f2 with
             * a = -1.5/h, b = 2/h, c = -0.5/h (and mirrored signs at the far end) */
            if (i == 0) gx = (-1.5 / h) * f(0, j) + (2. / h) * f(1, j) + (-0.5 / h) * f(2, j);
            else if (i == Nx - 1) gx = (0.5 / h) * f(i - 2, j) + (-2. / h) * f(i - 1, j) + (1.5 / h) * f(i, j);
            else gx = (f(i + 1, j) - f(i - 1, j)) / (2. * h);
            if (j == 0) gy = (-1.5 / h) * f(i, 0) + (2. / h) * f(i, 1) + (-0.5 / h) * f(i, 2);
            else if (j == Ny - 1) gy = (0.5 / h) * f(i, j - 2) + (-2. / h) * f(i, j - 1) + (1.5 / h) * f(i, j);
            else gy = (f(i, j + 1) - f(i, j - 1)) / (2. * h);
            double dx = gx * z / k_refr / (h * M), dy = gy * z / k_refr / (h * M);   /* RF2:55-56 */
            if (std::fabs(dx) < 1e-12) dx = 0;                              /* RF2:59-60 */
            if (std::fabs(dy) < 1e-12) dy = 0;
            double Iv = I[(int64_t)i * Ny + j];
            if (std::fabs(dx) > Nx) { Iv = 0; }                             /* RF2:61-64 */
            if (std::fabs(dy) > Ny) { Iv = 0; }
            if (std::fabs(dx) > Nx) dx = 0;
            if (std::fabs(dy) > Ny) dy = 0;
            const int64_t q = (int64_t)(i + mg) * Py + (j + mg);            /* RF2:65-67 zero pad */
            Dx[q] = dx;
            Dy[q] = dy;
            Ip[q] = Iv;
        }
    }
    /* RF2:217-263, raster order inside a band of rows per thread.  A deposit whose target row lies in the thread's own band is
     * a plain add; the few that cross a band border are kept in the thread's list and applied, band by band, after a barrier
     * -- no atomics, and one thread gives exactly the reference's raster order. */
    const int nt = nthreads < 1 ? 1 : (nthreads > Px ? Px : nthreads);
    std::vector<std::vector<std::pair<int64_t, double>>> cross(nt);
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        const int r0 = (int)((int64_t)Px * t / nt), r1 = (int)((int64_t)Px * (t + 1) / nt);
        auto &mine = cross[t];
        auto add = [&](int64_t row, int64_t col, double v) {
            if (row >= r0 && row < r1) I2[row * Py + col] += v;
            else mine.emplace_back(row * Py + col, v);
        };
        for (int i = r0; i < r1; ++i) {
            for (int j = 0; j < Py; ++j) {
                const double Iij = Ip[(int64_t)i * Py + j];
                double dx = Dx[(int64_t)i * Py + j], dy = Dy[(int64_t)i * Py + j];
                if (dx == 0.0 && dy == 0.0) {
                    I2[(int64_t)i * Py + j] += Iij;
                    continue;
                }
                int64_t inew = i, jnew = j;
                if (std::fabs(dx) > 1.0) { const double fl = std::floor(dx); inew = i + (int64_t)fl; dx -= fl; }
                if (std::fabs(dy) > 1.0) { const double fl = std::floor(dy); jnew = j + (int64_t)fl; dy -= fl; }
                if (inew < 0 || inew >= Px || jnew < 0 || jnew >= Py) continue;
                const double ax = std::fabs(dx), ay = std::fabs(dy);
                add(inew, jnew, Iij * (1.0 - ax) * (1.0 - ay));
                int64_t ix, jy;
                if (dx >= 0.0) { if (inew >= Px - 1) continue; ix = inew + 1; } else { if (inew <= 0) continue; ix = inew - 1; }
                if (dy >= 0.0) { if (jnew >= Py - 1) continue; jy = jnew + 1; } else { if (jnew <= 0) continue; jy = jnew - 1; }
                add(ix, jnew, Iij * ax * (1.0 - ay));
                add(ix, jy, Iij * ax * ay);
                add(inew, jy, Iij * (1.0 - ax) * ay);
            }
        }
#pragma omp barrier
        for (int u = 0; u < nt; ++u)
            for (const auto &e : cross[u]) {
                const int64_t row = e.first / Py;
                if (row >= r0 && row < r1) I2[e.first] += e.second;
            }
    }
    int bad = 0;
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(| : bad)
    for (int i = 0; i < Nx; ++i)
        for (int j = 0; j < Ny; ++j) {
            const double v = I2[(int64_t)(i + mg) * Py + j + mg];           /* RF2:78 crop */
            out[(int64_t)i * Ny + j] = v;
            if (std::isnan(v) || std::fabs(v) > 1e50) bad |= 1;
        }
    return bad;
}

/* setWaveRT + fastRefraction from the thickness maps (what one unit of the bench step does) */
int cb_refraction(const float *const *T, const double *delta, const double *beta, int nmat, int Nx, int Ny, double I0,
                  double k_sample, double k_refr, double z, double M, double pix_um, double *out, int nthreads) {
    std::vector<double> phi((size_t)Nx * Ny), I((size_t)Nx * Ny);
    cb_transmit_rt(T, delta, beta, nmat, Nx, Ny, I0, k_sample, I.data(), phi.data(), nthreads);
    return cb_fast_refraction(I.data(), phi.data(), Nx, Ny, k_refr, z, M, pix_um, out, nthreads);
}

}  /* extern "C" */
