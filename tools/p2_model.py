#!/usr/bin/env python3
"""numpy model of ONE line of the power-of-two LDS Fresnel engine (csrc/fresnel_p2.hip): the index conventions of the three
in-place stages (radix R1 x 16 x 16, M = 256 R1), the digit-reversed kernel-spectrum table, and the alias fix-up of the Lx = N +
P - 1 - M wrapped outputs -- checked against the operator it must equal, crop(IDFT_P(chirp * DFT_P(reflect_pad(x)))) (EXP:236-251
of the reference).  A development aid (float64, no GPU): `python tools/p2_model.py` prints the errors."""
import numpy as np


def taps(P, a, du):
    f = np.arange(P)
    f = np.where(f < (P + 1) // 2, f, f - P)
    c = np.exp(-1j * a * (f * du) ** 2)
    return np.fft.ifft(c)              # h[d] = (1/P) sum_k c_k exp(+2 pi i k d / P)


def reference_line(x, mg, h):
    N = len(x)
    P = N + 2 * mg
    xp = np.pad(x, mg, mode="reflect")
    return np.fft.ifft(np.fft.fft(h) * np.fft.fft(xp))[mg:mg + N]


def extension(x, mg, L):
    """e[j] = x_per[j - (P - 1) + mg], x_per the P-periodic extension of the reflect-padded line."""
    N = len(x)
    P = N + 2 * mg
    xp = np.pad(x, mg, mode="reflect")
    j = np.arange(L)
    return xp[(j - (P - 1) + mg) % P]


def dif3(v, R1, inverse=False):
    """in-place 3-stage transform as the kernel runs it.  forward: stage A (radix R1 over stride 256, then x w_M^{m k1}),
    stage B (radix 16 over stride 16 inside each block of 256, then x w_256^{n3 k2}), stage C (radix 16, contiguous).
    Output position p = k1*256 + k2*16 + k3 holds frequency k1 + R1*k2 + 16*R1*k3.  inverse: the mirror image, unnormalised."""
    M = 256 * R1
    v = v.reshape(R1, 16, 16).astype(complex)       # [n1 | k1][n2 | k2][n3 | k3]
    sgn = 1 if inverse else -1
    m = (np.arange(16)[:, None] * 16 + np.arange(16)[None, :])          # m = n2*16 + n3
    twA = np.exp(sgn * 2j * np.pi * np.arange(R1)[:, None, None] * m[None] / M)
    twB = np.exp(sgn * 2j * np.pi * np.arange(16)[:, None] * np.arange(16)[None, :] / 256)   # [k2][n3]
    F = lambda R: np.exp(sgn * 2j * np.pi * np.outer(np.arange(R), np.arange(R)) / R)
    if not inverse:
        v = np.einsum("kn,nab->kab", F(R1), v) * twA
        v = np.einsum("kn,anb->akb", F(16), v) * twB[None]
        v = np.einsum("kn,abn->abk", F(16), v)
    else:
        v = np.einsum("kn,abn->abk", F(16), v)
        v = np.einsum("kn,anb->akb", F(16), v * twB[None])
        v = np.einsum("kn,nab->kab", F(R1), v * twA)
    return v.reshape(M)


def perm_spectrum(h, R1):
    """H table of the middle stage: position p holds FFT_M(h zero-padded)[k(p)] / M."""
    M = 256 * R1
    Hf = np.fft.fft(np.concatenate([h, np.zeros(M - len(h))])) / M
    p = np.arange(M)
    k1, k2, k3 = p // 256, (p // 16) % 16, p % 16
    return Hf[k1 + R1 * k2 + 16 * R1 * k3]


def engine_line(x, mg, h, R1):
    N = len(x)
    P = N + 2 * mg
    M = 256 * R1
    L = N + P - 1
    Lx = L - M
    assert P <= M and N <= M // 2 and Lx <= 64 and N - 1 >= Lx
    e = extension(x, mg, L)
    buf = np.zeros(M + 64, complex)
    buf[:min(L, M + 64)] = e[:M + 64]             # positions >= M: the dropped samples e[M + j] (side region of the line buffer)
    sa = buf[:64].copy()                           # e[j], j < 64, saved before the in-place transform
    y = dif3(dif3(buf[:M], R1) * perm_spectrum(h, R1), R1, inverse=True)
    out = np.empty(N, complex)
    i = np.arange(N)
    mpos = i + P - 1
    direct = mpos < M
    out[direct] = y[mpos[direct]]
    # wrapped outputs: m' = m - M < Lx;  y[M + m'] = y_c[m'] + sum_{t <= m'} h[t] (e[M + m' - t] - e[m' - t])
    for mp in range(max(Lx, 0)):
        t = np.arange(mp + 1)
        corr = np.sum(h[t] * (buf[M + mp - t] - sa[mp - t]))
        out[mp + M - (P - 1)] = y[mp] + corr
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for N, mg, R1 in [(4096, 15, 32), (2048, 15, 16), (1024, 15, 8), (512, 15, 4), (4000, 15, 32), (4090, 10, 32), (500, 15, 4),
                      (4096, 32, 32), (257, 3, 4)]:
        x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        P = N + 2 * mg
        h = taps(P, 1.3e-11, 2 * np.pi / (N * 2.9e-6))
        ref = reference_line(x, mg, h)
        out = engine_line(x, mg, h, R1)
        print("N %5d margin %2d M %5d Lx %4d  max err %.2e" % (N, mg, 256 * R1, N + P - 1 - 256 * R1, np.abs(out - ref).max() / np.abs(ref).max()))
