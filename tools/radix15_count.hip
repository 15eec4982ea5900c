// Diagnostic for VERDICT r2 item 5b: M = 8640 = 24 x 24 x 15 (>= 8221) is 6 % fewer points than 9216 = 24 x 24 x 16 -- what
// would the radix-15 middle stage cost?  Compile-only:  tools/radix15_count.sh  counts the packed instructions of a forward
// 15-point and 16-point register DFT (same fft_pk.hpp idiom; the 15-point one = 3 x 5 with a 5-point butterfly written here).
#include "../paresis_amd/csrc/fft_pk.hpp"
using namespace psx;

template <bool INV>
__device__ __forceinline__ void dft5(v2f (&v)[5]) {
    constexpr float c1 = 0.30901699437494745f, c2 = -0.80901699437494745f, s1 = 0.95105651629515353f, s2 = 0.58778525229247314f;
    const v2f t1 = v[1] + v[4], t2 = v[2] + v[3], t3 = v[1] - v[4], t4 = v[2] - v[3];
    const v2f m1 = pk_fma_k(t2, c2, pk_fma_k(t1, c1, v[0])), m2 = pk_fma_k(t2, c1, pk_fma_k(t1, c2, v[0]));
    const v2f n1 = pk_fma_k(t4, s2, pk_mul_k(t3, s1)), n2 = pk_fma_k(t4, -s1, pk_mul_k(t3, s2));
    v[0] = v[0] + t1 + t2;
    v[1] = add_rot<INV>(m1, n1);
    v[4] = sub_rot<INV>(m1, n1);
    v[2] = add_rot<INV>(m2, n2);
    v[3] = sub_rot<INV>(m2, n2);
}

// 15 = 3 x 5: n = 5 n1 + n2, k = k1 + 3 k2; twiddles w15^{n2 k1} from sincos constants
__global__ void k_dft15(v2f *p) {
    v2f v[15], y[15];
    for (int i = 0; i < 15; ++i) v[i] = p[threadIdx.x * 15 + i];
    static constexpr float C[15] = {1.f, 0.91354546f, 0.66913061f, 0.30901699f, -0.10452846f, -0.5f, -0.80901699f, -0.97814760f,
                                    -0.97814760f, -0.80901699f, -0.5f, -0.10452846f, 0.30901699f, 0.66913061f, 0.91354546f};
    static constexpr float S[15] = {0.f, 0.40673664f, 0.74314483f, 0.95105652f, 0.99452190f, 0.8660254f, 0.58778525f, 0.20791169f,
                                    -0.20791169f, -0.58778525f, -0.8660254f, -0.99452190f, -0.95105652f, -0.74314483f, -0.40673664f};
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        v2f a[3] = {v[n2], v[5 + n2], v[10 + n2]};
        DftPk<3, false>::run(a);
#pragma unroll
        for (int k1 = 0; k1 < 3; ++k1) {
            const int t = (n2 * k1) % 15;
            y[n2 * 3 + k1] = t == 0 ? a[k1] : pk_cmulc_s(a[k1], (v2f){C[t], S[t]});
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
        v2f b[5] = {y[k1], y[3 + k1], y[6 + k1], y[9 + k1], y[12 + k1]};
        dft5<false>(b);
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) v[k1 + 3 * k2] = b[k2];
    }
    for (int i = 0; i < 15; ++i) p[threadIdx.x * 15 + i] = v[i];
}

__global__ void k_dft16(v2f *p) {
    v2f v[16];
    for (int i = 0; i < 16; ++i) v[i] = p[threadIdx.x * 16 + i];
    DftPk<16, false>::run(v);
    for (int i = 0; i < 16; ++i) p[threadIdx.x * 16 + i] = v[i];
}
