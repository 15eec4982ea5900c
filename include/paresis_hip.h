/*
 * paresis_hip.h -- C ABI of libparesis_hip.so: the MI355X (gfx950) image-formation hot path of an X-ray
 * speckle-propagation simulator, drop-in for the numerical kernels behind
 * Experiment.computeSampleAndReferenceImages_{Fresnel,RT} of quenotl/PARESIS.
 *
 * The reference has no FFI layer: its boundary is the Python call surface of CodePython/{Experiment,Sample,
 * Detector,refractionFileNumba2,refractionFileNumba,getk}.py.  Each entry point below names the reference
 * function (file:line, relative to CodePython/) whose array work it replaces; the Python modules under paresis_amd/ keep the reference's
 * names and signatures on top of this ABI (see INTEGRATION.md for the ctypes binding a maintainer would add).
 *
 * Conventions
 *   - every pointer named T/I/wave/phi/out/... is a DEVICE pointer (HBM); "host array" is said explicitly;
 *   - images are row-major [Nx][Ny] (numpy C order: axis 0 = "x" of the reference), float32 / interleaved complex64;
 *     the optional explicit phase is float64 (fp32 cannot hold kilo-radian phases to 1e-5);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is asynchronous on it, performs
 *     no host synchronisation and allocates nothing the caller must free (plans own their work buffers);
 *   - return value: 0 = ok, <0 = invalid argument (PSX_E_*), >0 = hipError_t or 1000+rocfft_status;
 *     psx_last_error() returns a thread-local description of the last failure;
 *   - thread model: one host thread per GPU/process; plans are not shared between threads.
 *
 * Materials: an object is a stack of thickness maps T[m] (metres, float32, [Nx][Ny]) with two per-map coefficients,
 *   cphase[m] (rad/m, added to the phase:  phi += cphase[m]*T[m];  the reference's -k*delta, Sample.py:279,348) and
 *   catt[m]   (1/m, log-attenuation:       log a += catt[m]*T[m]; -k*beta for a wave, -2*k*beta for an intensity,
 *   Sample.py:279,347).  Products and sums over m are formed in float64 on the device, then range-reduced.
 */
#ifndef PARESIS_HIP_H
#define PARESIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSX_MAX_MAT 8
#define PSX_MAX_DIST 8
#define PSX_MAX_SRC 16    /* source waves per psx_fresnel_propagate_sources / images per psx_accumulate_many_f32 call */

#define PSX_E_ARG (-1)      /* bad argument (null pointer, size, count) */
#define PSX_E_STATE (-2)    /* plan/shape mismatch */
#define PSX_E_UNSUPPORTED (-3)

/* bits of the device status word (psx_status_*) */
#define PSX_STATUS_NONFINITE 1u   /* NaN or |v| > 1e50 in a refracted image: refractionFileNumba2.py:81-82 raises */

typedef struct psx_fresnel_plan psx_fresnel_plan;
typedef struct psx_detector_plan psx_detector_plan;

typedef struct { float re, im; } psx_c64;

/* library / build identification; psx_abi_version() changes when a signature changes */
int psx_abi_version(void);
const char *psx_last_error(void);
/* 1 when the code object for the current device (gfx950) is loadable, else 0 with psx_last_error() set */
int psx_device_ok(void);
/* Diagnostics: the shader clock (MHz) the device runs at under load right now -- a 30 us spin on every CU, issued on `stream` behind
 * whatever is queued there, bracketed by the shader-clock and the constant 100 MHz counters; synchronises the stream.  bench.py
 * reports it next to its timings: the boxes of a pool differ by more than 10 % in what they sustain. */
int psx_clock_probe(float *mhz, void *stream);

/* ---- K1: AnalyticalSample.setWave (Sample.py:248-282) ------------------------------------------------------
 * wave_out[p] = amp * wave_in[p] * exp(sum_m catt[m]*T[m][p]) * exp(i * sum_m cphase[m]*T[m][p]),  p < n.
 * wave_in may be NULL (unit wave).  T is a HOST array of nmat device pointers.  In place (wave_out==wave_in) is ok. */
int psx_transmit_wave_c64(const psx_c64 *wave_in, float amp, const float *const *T, const double *cphase,
                          const double *catt, int nmat, psx_c64 *wave_out, int64_t n, void *stream);

/* ---- K2: AnalyticalSample.setWaveRT (Sample.py:285-351, scalar dark field) ------------------------------------
 * I_out[p] = I0 * I_in[p] * exp(sum_m catt[m]*T[m][p]);   phi_out[p] = phi_in[p] + sum_m cphase[m]*T[m][p].
 * I_in may be NULL (ones); phi_in may be NULL (zeros); phi_out may be NULL (phase not wanted); I_out may be NULL. */
int psx_transmit_rt_f32(const float *I_in, float I0, const float *const *T, const double *cphase, const double *catt,
                        int nmat, float *I_out, const double *phi_in, double *phi_out, int64_t n, void *stream);

/* acc[p] (+)= scale * img[p] * exp(sum_m catt[m]*T[m][p])  -- plate attenuation + energy accumulation
 * (Experiment.py:351-358, 478-483).  accumulate: 0 store, 1 add.  nmat may be 0. */
int psx_accumulate_f32(float *acc, const float *img, float scale, const float *const *T, const double *catt, int nmat,
                       int accumulate, int64_t n, void *stream);

/* The same, also reducing what it adds (Experiment.py:360-361 / 485-486: np.mean of the per-energy reference image feeds
 * the intensity-weighted mean energy): with v[p] = scale*img[p]*exp(sum catt*T[p]) and S = sum_p v[p], the kernel adds S and
 * weight*S (weight = the energy) into `sums`, a device float64 array of PSX_SUM_SLOTS slots of PSX_SUM_STRIDE doubles
 * (128 bytes apart, so that the atomics of different workgroups do not serialise on one line), caller-zeroed:
 *   sum_s sums[s*PSX_SUM_STRIDE + 0] = sum of S over the calls,   sum_s sums[s*PSX_SUM_STRIDE + 1] = sum of weight*S.
 * acc may be NULL (sums only). */
#define PSX_SUM_SLOTS 32
#define PSX_SUM_STRIDE 16
int psx_accumulate_sum_f32(float *acc, const float *img, float scale, const float *const *T, const double *catt, int nmat,
                           int accumulate, int64_t n, double *sums, double weight, void *stream);
/* The same for the images of n_img <= PSX_MAX_SRC energies in ONE pass (the energies of a detector bin): acc (+)= sum over e of
 * scale[e] * imgs[e] * exp(sum_i catt[e*nmat + i] * T[i]), added in the order of e exactly as one psx_accumulate_sum_f32 call
 * per energy would (bit-identical float32 result); sums (may be NULL) += (sum of all terms, sum of weight[e] * term).
 * acc may be NULL (sums only).  imgs, scale, catt, weight: host arrays. */
int psx_accumulate_many_f32(float *acc, const float *const *imgs, const float *scale, int n_img, const float *const *T,
                            const double *catt, int nmat, int accumulate, int64_t n, double *sums, const double *weight,
                            void *stream);

/* ---- K9-K13: fastRefraction (refractionFileNumba2.py:25-86; variant v1: refractionFileNumba.py:11-68) ----------
 * Source intensity  I_src = I0 * I_in * exp(sum catt*T)          (I_in may be NULL = ones; fused K2)
 * Phase             phi   = phi_in + sum cphase*T                 (phi_in float64, may be NULL)
 * Displacement      D     = grad(phi) * dscale, np.gradient(edge_order=2) with unit spacing, i.e. the caller passes
 *                   dscale = z / k / (h*M) / h  (RF2:54-56);  |D|<1e-12 -> 0;  |Dx|>clamp_x or |Dy|>clamp_y -> I=0,
 *                   that component = 0 (RF2:59-64: clamp = Nx,Ny for v2, 1e3 for v1).
 * Output            I_out[Nx][Ny] (+)= out_scale * bilinear scatter of I_src by D on a grid padded by `margin`
 *                   (15 for v2, 10 for v1), cropped back (RF2:65-78, loop RF2:198-263).
 * Dx_out/Dy_out     optional [Nx+2*margin][Ny+2*margin] float32 (zero margins), as the reference returns them.
 * I_mut             optional: the reference zeroes clamped entries of its INPUT intensity in place (RF2:61-62);
 *                   pass I_in here to reproduce that, NULL otherwise.
 * status            optional device word; PSX_STATUS_NONFINITE is OR-ed in when the output holds NaN/inf.
 * workspace         device scratch of psx_refract_workspace_bytes(Nx,Ny) bytes (far-ray lists; with psx_set_deterministic on,
 *                   also the scratch words of the order-independent replay), caller-owned, no initial state needed.
 */
size_t psx_refract_workspace_bytes(int Nx, int Ny);
/* Gather halo of the tile kernel: 4, 6, 8, 12 or 16 pixels (default 4; the two widest for grids whose rays travel far in study pixels); a setting of the CALLING HOST THREAD (one thread per GPU).  A tile gathers every ray of its window (tile + halo)
 * that lands in it, however long; what remains for the slower far-ray replay are the shares whose source lies outside
 * the window of the target's tile.  A pure speed knob: results are identical up to the float-atomics order of those. */
int psx_refract_set_halo(int halo);
int psx_refract_f32(const float *I_in, float I0, const float *const *T, const double *cphase, const double *catt,
                    int nmat, const double *phi_in, float *I_out, float out_scale, int accumulate, float *Dx_out,
                    float *Dy_out, float *I_mut, int Nx, int Ny, int margin, double dscale, double clamp_x,
                    double clamp_y, unsigned *status, void *workspace, void *stream);
/* fastRefractionDF's split by its width map (refractionFileNumba2.py:147-150: I_nodf = I where DF == 0, I_df = I where
 * DF != 0) and the refraction of both halves (RF2:153-154) in one call: I_out_zero receives the refraction of the sources
 * where mask == 0, I_out_nonzero of those where mask != 0.  A tile is staged ONCE for both; a half runs its deposit loop only
 * where the tile's window holds a source of its side.  With the thickness maps as the source of intensity and phase, the
 * dark-field branch of the chain (EXP:469-473) needs neither the transmitted (I, phi) pair nor the two halves of the split in
 * memory.  mask: [Nx][Ny] float32 (the width map in pixels).  The phase comes from the thickness maps (nmat > 0; phi_in must
 * be NULL: the argument keeps the call's shape that of psx_refract_f32).  No displacement maps, no input mutation; workspace:
 * psx_refract_multi_workspace_bytes(Nx, Ny, 2); everything else as psx_refract_f32. */
int psx_refract_split_f32(const float *I_in, const float *mask, float I0, const float *const *T, const double *cphase,
                          const double *catt, int nmat, const double *phi_in, float *I_out_zero, float *I_out_nonzero,
                          float out_scale, int accumulate, int Nx, int Ny, int margin, double dscale, double clamp_x,
                          double clamp_y, unsigned *status, void *workspace, void *stream);

/* Propagation-distance batch (BASELINE.json north_star: "propagation-distance batches"; the reference would call
 * Experiment.refraction, Experiment.py:255-277, once per distance on the same (I, phi)): ndist <= PSX_MAX_DIST
 * refractions of ONE source (same I_in/I0/T/phi_in) with displacement scales dscale[d] into the distinct images
 * I_out[d] (host arrays of ndist entries), in one launch per kernel -- the thickness maps are read and the
 * transmission evaluated once per tile instead of once per distance.  Each image is bit-identical to the one
 * psx_refract_f32 gives for that distance (up to the float-atomics order of far rays).  Dx_out/Dy_out/I_mut are
 * only accepted with ndist == 1.  workspace: psx_refract_multi_workspace_bytes(Nx, Ny, ndist) bytes. */
size_t psx_refract_multi_workspace_bytes(int Nx, int Ny, int ndist);
int psx_refract_multi_f32(const float *I_in, float I0, const float *const *T, const double *cphase, const double *catt,
                          int nmat, const double *phi_in, float *const *I_out, float out_scale, int accumulate,
                          float *Dx_out, float *Dy_out, float *I_mut, int Nx, int Ny, int margin, const double *dscale,
                          int ndist, double clamp_x, double clamp_y, unsigned *status, void *workspace, void *stream);
/* A batch of n <= PSX_MAX_SRC refractions over the SAME thickness maps in one launch per kernel -- the energies of a detector
 * bin (Experiment.py:448-486 loops over them): refraction e has its own input image I_in[e] (all or none; none = the uniform
 * I0[e]), coefficients cphase/catt[e*nmat + i], displacement scale dscale[e] and output image I_out[e]; one distance each, no
 * displacement maps.  Every image is what psx_refract_f32 gives for that refraction (far rays apart: float atomics).
 * workspace: psx_refract_batch_workspace_bytes(Nx, Ny, n) bytes. */
size_t psx_refract_batch_workspace_bytes(int Nx, int Ny, int n);
int psx_refract_batch_f32(int n, const float *const *I_in, const float *I0, const float *const *T, const double *cphase,
                          const double *catt, int nmat, float *const *I_out, float out_scale, int accumulate, int Nx, int Ny,
                          int margin, const double *dscale, double clamp_x, double clamp_y, unsigned *status, void *workspace,
                          void *stream);

/* Order-independent scatter (the reference's scatter is a single-threaded raster loop, RF2:217-263; the
 * tile gathers are fixed point and reproducible as they are, but far rays and psx_fastloop_f32 deposit with global float
 * atomics in arrival order, and a last-bit difference can flip a Poisson draw downstream).  With on != 0 the far-ray replay of
 * psx_refract_f32 / _multi_f32 / _batch_f32 sums in fixed point: the tile kernel clears the 64-bit scratch words a far ray may
 * reach when it lists the ray; the replay adds every share as an integer -- ONE unit per call, 2^-30 of the power of two
 * above the largest intensity the call stages, so every share fits whatever tile it lands in -- with a returning atomic (the
 * thread that reads back zero was first at that pixel and notes the pixel in its list's fold table); a second pass adds each
 * pixel's complete sum ONCE to the float image.  Two runs of the same call are then bitwise equal, whatever the GPU count.
 * One atomic per share, as in the float form.  Allocates nothing and synchronises nothing: the scratch ([ndist][Nx*Ny] words)
 * is part of the caller's workspace, needs no initial state, and psx_refract_*_workspace_bytes() includes it WHILE THE MODE
 * IS ON (set the mode, then size the workspace).  psx_fastloop_f32, which has no workspace argument, keeps the allocating
 * form (hipMalloc + a stream synchronisation per call).  A setting of the calling host thread, off by default in the library;
 * the Experiment class turns it on around its ray-tracing chain unless exp_dict['reproducible'] is False (measured cost:
 * DESIGN.md section 4.3).  psx_get_deterministic returns the calling thread's setting (callers that restore it). */
int psx_set_deterministic(int on);
int psx_get_deterministic(void);
/* Optional: the fixed-point unit of the order-independent replay from the CALLER's intensity scale instead of the call's measured
 * maximum -- scale > 0: one unit = 2^-30 of the power of two above 64 * scale (a share up to 2^10 times that is taken -- 2^11 of them
 * fit a pixel's sum --; beyond, PSX_STATUS_NONFINITE is raised); the tile kernels then send no maximum and the call needs no memset node (4.7 us per call at
 * 4096^2).  The sums are quantised to the unit whatever the image holds, so the scale should be the image's order of magnitude
 * (the Experiment class passes the incident intensity per study pixel).  0 (default): measure.  A setting of the calling thread. */
int psx_set_deterministic_scale(float scale);
/* the calling thread's setting (callers that restore it: a nested scope must not lose the outer scope's scale) */
float psx_get_deterministic_scale(void);

/* The raw scatter loop on explicit displacement fields: fastloopNumba (refractionFileNumba2.py:198-263).
 * I, Dx, Dy, I2 are [Nx][Ny]; I2 is accumulated into (float atomics; order-dependent in the last bits). */
int psx_fastloop_f32(const float *I, const float *Dx, const float *Dy, float *I2, int Nx, int Ny, void *stream);

/* ---- K3-K8: Experiment.wavePropagation (Experiment.py:219-252) ------------------------------------------------
 * A plan fixes the study grid [Nx][Ny] and the reflect margin (15, EXP:236) and owns the padded work buffers.
 * engine: 0 = auto, 1 = rocFFT on the padded grid (pad -> FFT2 -> chirp -> IFFT2 -> crop, any size),
 *         2 = LDS-resident FFT convolution (row pass + column pass, the padded spectrum never touches HBM; lines longer
 *             than one LDS transform, N > 4593, as a partitioned convolution).  Auto picks 2 for every supported grid. */
#define PSX_ENGINE_AUTO 0
#define PSX_ENGINE_ROCFFT 1
#define PSX_ENGINE_LDS 2
int psx_fresnel_plan_create(int Nx, int Ny, int margin, int max_dist, int engine, psx_fresnel_plan **plan);
int psx_fresnel_plan_destroy(psx_fresnel_plan *plan);
/* engine actually selected (PSX_ENGINE_ROCFFT / PSX_ENGINE_LDS) */
int psx_fresnel_plan_engine(const psx_fresnel_plan *plan);
/* bytes of device memory the plan owns */
size_t psx_fresnel_plan_bytes(const psx_fresnel_plan *plan);
/* on != 0: the LDS engine's one-transform passes (N <= 4593) hand their line groups to the workgroups through queues (the
 * share of a workgroup is an atomic counter it claims from; a workgroup whose share is done steals from the others of its
 * XCD) instead of equal static shares.  Static shares are ~4 % faster on a GPU the call has to itself, and
 * TWICE as slow as soon as one CU is busy with anything else when a pass starts (its 256 workgroups need a whole CU each:
 * the one that cannot start waits for another to finish its whole share) -- e.g. the copy kernels of an RCCL transfer that
 * overlaps the computation.  Same results bit for bit.  Default off; no effect on the other engine paths. */
int psx_fresnel_plan_work_queue(psx_fresnel_plan *plan, int on);

/* Propagate ONE input wave to n_dist distances, e.g. EXP:341 and EXP:349 (shared across the distances: the transmitted
 * source wave in the LDS engine, the forward 2-D transform as well in the rocFFT engine).
 *   input  psi = amp * wave_in * transmission(T, cphase, catt)          (wave_in may be NULL = unit wave; fused K1)
 *   for d < n_dist:  out_d = exp(i*gphase[d]) * IDFT2( exp(-i*a[d]*(u^2+v^2)) * DFT2(reflect_pad(psi)) ) cropped,
 *                    u_i = (i - Px/2)*du_x, v_j = (j - Py/2)*du_y  with du = 2*pi/(N*h) from the UN-padded N (EXP:246-247),
 *                    a[d] = z/(2*k*M), gphase[d] = k*z/M  (EXP:250);  a[d] == 0 means "z == 0": out_d = psi (EXP:233).
 *   wave_out[d]  (host array of device pointers, entries may be NULL): complex result [Nx][Ny]
 *   inten_out[d] (host array of device pointers, entries may be NULL): inten_out[d] (+)= inten_scale[d]*|out_d|^2 (K8)
 */
int psx_fresnel_propagate(psx_fresnel_plan *plan, const psx_c64 *wave_in, float amp, const float *const *T,
                          const double *cphase, const double *catt, int nmat, int n_dist, const double *a,
                          const double *gphase, double du_x, double du_y, psx_c64 *const *wave_out,
                          float *const *inten_out, const float *inten_scale, int accumulate, void *stream);
/* The same for n_src <= PSX_MAX_SRC source waves over the SAME thickness maps in one call -- the energies of a detector bin
 * (EXP:317-361 loops over them): source s has its own input wave wave_in[s] (the array or an entry may be NULL = unit wave),
 * amplitude amp[s] and coefficients cphase/catt[s*nmat + i]; pair (s, d) has its own a, gphase, outputs and scale at index
 * s*n_dist + d.  Every pair's result is exactly what psx_fresnel_propagate gives for that source (nothing is accumulated:
 * each pair owns its output).  On grids too small to fill the chip (fewer line groups than CUs) the LDS engine runs the
 * whole batch in three launches -- the pairs are one more axis of its work items -- when n_src*n_dist <= 32; otherwise the
 * sources are taken one after the other.  The first batched call of a plan allocates its batch buffers (not under capture). */
int psx_fresnel_propagate_sources(psx_fresnel_plan *plan, int n_src, int n_dist, const psx_c64 *const *wave_in, const float *amp,
                                  const float *const *T, const double *cphase, const double *catt, int nmat, const double *a,
                                  const double *gphase, double du_x, double du_y, psx_c64 *const *wave_out,
                                  float *const *inten_out, const float *inten_scale, void *stream);

/* ---- K14-K19: Detector.detection (Detector.py:79-119), resize (:185-198), create_gaussian_shape (:201-220) -----
 * reflect-pad 15*ov -> source blur (sigma_src study px, 0 = none) -> ov x ov block SUM -> PSF blur (sigma_psf detector
 * px, 0 = none) -> crop 15.  All stages are linear and separable; the plan holds the composed banded operators.
 * out [nx][ny] (+)= ... ; Poisson noise is a separate call (psx_poisson_f32). */
int psx_detector_plan_create(int Nx, int Ny, int ov, int nx, int ny, int margin, double sigma_src, double sigma_psf,
                             psx_detector_plan **plan);
int psx_detector_plan_destroy(psx_detector_plan *plan);
int psx_detect_f32(psx_detector_plan *plan, const float *img, float *out, void *stream);
/* The detector operator on nimg <= PSX_MAX_DETECT images of the plan's shape in one call (imgs, outs: host arrays of device
 * pointers; distinct outputs) -- the two to four images of an energy bin (Experiment.py:388-394 / 507-514).  When both stages run
 * fused the images share each launch; out[i] is bit for bit what psx_detect_f32(plan, imgs[i], outs[i]) would write. */
#define PSX_MAX_DETECT 4
int psx_detect_multi_f32(psx_detector_plan *plan, const float *const *imgs, float *const *outs, int nimg, void *stream);
/* Host-only view of one axis of the composed operator (no GPU needed): row r of the [n x N] banded matrix is
 * weights[r*wcap .. r*wcap+W) applied to study pixels start[r] .. start[r]+W.  *W_out receives the band width;
 * fails with PSX_E_ARG when it exceeds wcap. */
int psx_detector_operator_host(int N, int ov, int n, int margin, double sigma_src, double sigma_psf, int *start,
                               float *weights, int wcap, int *W_out);
/* out[x][y] = sum of the s x s block of img, s = Nx/sx (Detector.resize, Detector.py:185-198) */
int psx_resize_f32(const float *img, int Nx, int Ny, float *out, int sx, int sy, void *stream);
/* out[p] = Poisson(lam[p]) drawn from a counter-based generator keyed by (seed, p)  (Detector.py:113-115;
 * the reference seeds from the wall clock, so only the distribution is reproducible) */
int psx_poisson_f32(const float *lam, float *out, int64_t n, uint64_t seed, void *stream);
/* The same IN PLACE on nimg <= PSX_MAX_POISSON images of n pixels in ONE launch, image i under key seeds[i] (imgs, seeds:
 * host arrays) -- the three or four detector images of an energy bin (Experiment.py:388-394).  Image i comes out exactly as
 * psx_poisson_f32(imgs[i], imgs[i], n, seeds[i]) would leave it. */
#define PSX_MAX_POISSON 8
int psx_poisson_multi_f32(float *const *imgs, const uint64_t *seeds, int nimg, int64_t n, void *stream);
/* Photon-count images as 16-bit integers for the gather of the per-position stacks (main.py:63-110 keeps every position's
 * images on one host; here they cross xGMI once): dst[p] = src[p] for counts below 65535; a larger count leaves the escape
 * code 65535 in dst and the pair (index0 + p, count) in exc[cap][2], *exc_count counting them (both device memory, the count
 * zeroed by the caller; index0 + n < 2^32).  *overflow (a device int the caller zeroed) is raised when some src[p] is not an
 * integer in [0, 2^24] or the table is full -- the caller then moves the float32 image instead, so the round trip is lossless
 * or not taken.  psx_unpack_counts_u16 is the inverse over a whole buffer: widen n pixels, then write back the exceptions
 * whose index is below n (cap = 0: none). */
int psx_pack_counts_u16(const float *src, uint16_t *dst, int64_t n, int64_t index0, int32_t *exc, int32_t *exc_count, int cap,
                        int *overflow, void *stream);
int psx_unpack_counts_u16(const uint16_t *src, float *dst, int64_t n, const int32_t *exc, const int32_t *exc_count, int cap,
                          void *stream);

/* ---- dark-field refraction, second half: the per-pixel variable-width Gaussian re-splat of
 * fastRefractionDF (refractionFileNumba2.py:168-186).  I2DF: refracted dark-field intensity, DF: dark-field width in
 * pixels at each TARGET pixel (0 = no spreading), I2: refracted non-dark-field intensity added at the end (may be NULL),
 * all [Nx][Ny] on the cropped grid; R >= round(1.5*max DF) is the largest patch half-size.  workspace:
 * psx_darkfield_workspace_bytes(Nx,Ny) bytes. */
size_t psx_darkfield_workspace_bytes(int Nx, int Ny);
int psx_darkfield_blur_f32(const float *I2DF, const float *DF, const float *I2, float *out, int Nx, int Ny, int R,
                           void *workspace, void *stream);
/* The front of fastRefractionDF in one pass (refractionFileNumba2.py:114-150): DF_px = DF_rad * num / den in float64 and in
 * that order -- RF2:114 with num = propagationDistance, den = studyPixelSize*1e-6*magnification, so that the step functions
 * of it (margin, DF > Nx/4 rule, patch sides) fall where the reference's fall --, its largest
 * value before and after the rule DF_px > limit -> 0 (RF2:135; words[0], words[1]: the doubles' bit patterns, device memory,
 * read back by the caller only when it does not know the maximum), the split of I by DF_px != 0 into I_nodf / I_df
 * (RF2:147-150) and the per-source patch table `prep` (psx_darkfield_workspace_bytes) of the re-splat, which depends on
 * the width map alone.  psx_darkfield_blur_prepared_f32 is psx_darkfield_blur_f32 on that table;
 * psx_darkfield_merge_f32: I = a + b (the caller's array after the clamped rays were zeroed in both halves, RF2:128-129);
 * psx_repad_f32: the centre [Nx][Ny] of a map padded by margin_src, re-padded with zeros to margin_dst (the displacement
 * maps fastRefractionDF returns are padded by ceil(6 max DF), RF2:117,139-140). */
int psx_darkfield_split_f32(const float *I, const double *DF_rad, double num, double den, double limit, float *I_nodf,
                            float *I_df, float *DF_px, void *prep, unsigned long long *words, int Nx, int Ny, void *stream);
int psx_darkfield_blur_prepared_f32(const float *I2DF, const float *DF, const void *prep, const float *I2, float *out, int Nx,
                                    int Ny, int R, unsigned *status, int accumulate, void *stream);   /* status: optional word, PSX_STATUS_NONFINITE (RF2:190-193); accumulate: out += result (the chain's sum over energies, EXP:478-483) */
int psx_darkfield_merge_f32(float *I, const float *a, const float *b, int64_t n, void *stream);
int psx_repad_f32(const float *src, int margin_src, float *dst, int margin_dst, int Nx, int Ny, void *stream);

/* ---- membrane thickness synthesis (next row of the scope table) -----------------------------
 * getMembraneSegmentedFromFile's sphere splat (Samples/getMembraneFromFile.py:143-159) for ONE layer:
 * out[i][j] (+)= scale * sum over spheres of 2*sqrt(r^2 - dist^2) on the cropped grid, with the reference's window and
 * placement rules.  xf, yf, rad are HOST arrays (pixels of the margin-extended grid: they come from a host-side list);
 * out is a DEVICE image.  Unlike the rest of the ABI this call synchronises the stream once (host staging buffers). */
int psx_membrane_f32(const double *xf, const double *yf, const double *rad, int64_t n, int dimX, int dimY, int margin,
                     int margin2, double scale, int accumulate, float *out, void *stream);

/* The same with the sphere list resident on the GPU.  The reference re-places ONE scaled, stitched list for every
 * membrane position and layer with a new integer offset (getMembraneFromFile.py:139-142); a plan takes the list once
 * (HOST arrays x, y, r in pixels of the list frame, i.e. par/pixSize before the offset), bins it by 32-pixel cell, and
 * psx_membrane_layer_f32 renders one layer at offset (offx, offy) -- xfloat = x - offx, yfloat = y - offy -- without
 * touching the host: asynchronous on the stream like the rest of the ABI. */
typedef struct psx_membrane_plan psx_membrane_plan;
int psx_membrane_plan_create(const double *x, const double *y, const double *r, int64_t n, psx_membrane_plan **plan);
int psx_membrane_plan_destroy(psx_membrane_plan *plan);
int psx_membrane_layer_f32(psx_membrane_plan *plan, int offx, int offy, int dimX, int dimY, int margin, int margin2,
                           double scale, int accumulate, float *out, void *stream);
/* Every layer of a membrane position in ONE launch (getMembraneFromFile.py:139-159 loops over nbOfLayers offsets and adds
 * into one float64 map): the chords of all nlayers offsets (offx, offy: HOST arrays) are summed in float64 and stored
 * once; nlayers = 0 stores zeros.  support (may be NULL) is the position's second map, the uniform support thickness
 * (getMembraneFromFile.py:163), filled with support_value by the same launch. */
int psx_membrane_layers_f32(psx_membrane_plan *plan, int nlayers, const int *offx, const int *offy, int dimX, int dimY,
                            int margin, int margin2, double scale, int accumulate, float *out, float *support,
                            float support_value, void *stream);

/* ---- per-kernel timing (bench.py's roofline leg) ----------------------------------------------------------------------
 * psx_profile_enable(1) clears the log and makes every kernel launch of the library record a HIP event pair on the
 * stream it is launched on; psx_profile_summary() waits for the recorded events and writes one line per kernel,
 * "name count total_ms\n".  Off by default (no events, no overhead). */
int psx_profile_enable(int on);
int psx_profile_summary(char *buf, size_t cap);

/* Diagnostics: device buffer receiving phase timestamps (100 MHz wall clock): 32 x uint64 per workgroup from the line
 * kernels of the LDS Fresnel engine, 16 x uint64 per workgroup from the refraction kernel (tools/stamp_*.py).
 * NULL (default) switches it off.  Never enabled in timed runs. */
int psx_debug_stamps(void *buf);

/* Diagnostic A/B switches, process-wide, all off by default; the library reads NOTHING from the environment.  Names:
 *   "no_dif", "no_pair" (read when a Fresnel plan is created), "no_dual", "no_dist_inner", "stamp_pass1", "stamp_round"
 *   (default 1), "detect_4pass", "far_stride" (the far-ray replay walks its lists in that stride instead of tile order), "near_lds_pad" (KiB of LDS added to the refraction tile launch: an occupancy probe), "no_p2" (read when a Fresnel plan is created: the
 *   576*R3-point line transforms of round 5 instead of the power-of-two ones).  tools/ and tests/ set them; timed product runs never do.  psx_debug_switches_active writes
 *   "name=value ..." of every switch that is not at its default (empty string: none) -- bench.py echoes it in its JSON line. */
int psx_debug_switch(const char *name, int value);
int psx_debug_switches_active(char *buf, size_t cap);

/* ---- status word ------------------------------------------------------------------------------------------------ */
/* OR PSX_STATUS_NONFINITE into *status when img holds NaN or |v| > 1e50 (float32: inf) */
int psx_status_scan_f32(const float *img, int64_t n, unsigned *status, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PARESIS_HIP_H */
