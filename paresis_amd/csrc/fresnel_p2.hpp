// fresnel_p2.hpp -- host interface of the power-of-two line kernels of the LDS Fresnel engine (fresnel_p2.hip).
//
// A line of N samples whose extension (N + P - 1 = 2N + 2*margin - 1 points) exceeds a power of two M by Lx <= 32 points is
// convolved through ONE M-point circular transform instead of the 576*R3 >= N + P - 1 points of fresnel_lds.hip's
// k_fresnel_lines (4096 samples: 8192 points instead of 9216, radices 32 x 16 x 16 instead of 24 x 24 x 16); the Lx outputs
// whose window wraps are put right by the taps they miss, Lx (Lx + 1) / 2 complex multiply-adds per line and distance.
#pragma once
#include "fresnel_stages.hpp"

namespace psx {
namespace p2 {

constexpr int LXMAX = 32;     // wrapped outputs a line may have (margin 15 on a power-of-two grid: 29)

// radix of the first stage (M = 256 * R1, R1 in {4, 8, 16, 32}) if lines of N samples with this margin run on these kernels, else 0
int pick_r1(int N, int margin);

// stage twiddles in global memory: twA = [256][log2 R1] powers w_M^(n 2^b) of stage A's row n; twB = [16][16] w_256^{n3 k2}
size_t twA_elems(int R1);
size_t twB_elems();
int build_tables(float2 *twA, float2 *twB, int R1, hipStream_t st);

// kernel-spectrum table of one distance: out[p] = Hh[k(p)] / M for the M positions of the in-place transform (k(p) = k1 + R1 k2 +
// 16 R1 k3 for p = 256 k1 + 16 k2 + k3), followed by the first LXMAX taps h[t] = hP[t] / P (float32)
size_t spectrum_elems(int R1);
int perm_spectrum(const double2 *Hh, const double2 *hP, float2 *out, int R1, int P, hipStream_t st);

// one pass over the lines of an image: la as for k_fresnel_lines, with twA / twB from build_tables and H[d] from perm_spectrum.
// dual: pass 1 of a call with several distances (one line x two distances per round; la.dist_inner must be 1).
int launch(int R1, bool contig, bool dual, const lines::LineArgs &la, hipStream_t st, const char *name);

// (source, distance) pairs one launch can carry (their first taps live in LDS): MAX_LINE, or a few less at R1 = 4
int max_distances(int R1);

// image lines per round of a launch (the host's choice of the work order needs it)
int lines_per_round(int R1, bool dual);

// ---- lines of about 16384 samples: one 32768-point circular convolution in two coupled rounds (fresnel_p2x.hip)
bool x_serves(int N, int margin);                 // do lines of N samples with this margin run on that kernel?
int x_points();                                   // points of one round's transform (the kernel spectrum is built from two of them)
size_t x_spectrum_elems();                        // float2 elements of one distance's table (two rounds of pairs + the first taps)
size_t x_line_buffer_elems();                     // float2 elements of a workgroup's private line buffer (LineArgs::wgpart)
int x_build_twiddles(float2 *w2, float2 *w4, hipStream_t st);      // 512 entries each (LineArgs::w2, w4)
int x_pad_taps(const double2 *hP, double2 *buf, int P, hipStream_t st);     // taps -> the two folded sequences of x_points()
int x_perm_spectrum(const double2 *Hh, const double2 *hP, float2 *out, int P, hipStream_t st);
// la as for fresnel_p2's launch at R1 = 32 (twA / twB from build_tables(.., 32, ..)), H[d] from x_perm_spectrum, w2 / w4 / wgpart set
int x_launch(bool contig, const lines::LineArgs &la, hipStream_t st, const char *name);

}  // namespace p2
}  // namespace psx
