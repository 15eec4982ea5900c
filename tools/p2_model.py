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


# ---- lines of about 4 x 8192 samples' worth of extension (N = 16384): ONE circular convolution of R = 32768 points, split by a
# radix-2 decimation-in-frequency step over two rounds, each a Q = 16384-point transform whose even / odd samples live in the two
# LDS lines (coupled by the radix-2 butterfly of the double-size transform in the middle stage) -- csrc/fresnel_p2x.hip
def pair_fft(a, b, R1=32, inverse=False):
    """The two LDS lines a (even samples) and b (odd samples) of one Q = 2M-point sequence -> the two lines of its spectrum in
    digit order: line 0 position p holds bin k(p), line 1 position p holds bin k(p) + M."""
    M = 256 * R1
    p = np.arange(M)
    k = p // 256 + R1 * ((p // 16) % 16) + 16 * R1 * (p % 16)
    if not inverse:
        A, B = dif3(a, R1), dif3(b, R1)
        t = B * np.exp(-2j * np.pi * k / (2 * M))
        return A + t, A - t
    y0, y1 = a, b
    return dif3(y0 + y1, R1, inverse=True), dif3((y0 - y1) * np.exp(+2j * np.pi * k / (2 * M)), R1, inverse=True)


def engine_line_dif(x, mg, h, R1=32):
    N = len(x)
    P = N + 2 * mg
    M = 256 * R1
    Q, R = 2 * M, 4 * M
    L = N + P - 1
    Lx = L - R
    assert Q < P <= Q + 64 and Lx <= 32 and P <= R
    e = extension(x, mg, L)
    # one copy in LDS: positions t < P (the slack behind the Q points of the transform holds t >= Q); the partner x[n + Q] of
    # position n is e[n + Q] = e[(n + Q) mod P]: position n - (P - Q) for n >= P - Q, the slack position Q + n below
    lds = e[:P].copy()
    sa, sb = lds[:32].copy(), lds[R - P:R - P + 32].copy()        # e[j] and e[R + j] = e[R - P + j], j < 32: saved for the fix-up
    n = np.arange(Q)
    partner = lds[(n + Q) % P]
    # kernel spectrum halves: H_R[2k] = FFT_Q(h folded), H_R[2k+1] = FFT_Q((h[d] - h[d + Q]) w_R^d); 1/Q (inverse) and 1/2 (recombination)
    hp = np.concatenate([h, np.zeros(R - P)])
    he = hp[:Q] + hp[Q:]
    ho = (hp[:Q] - hp[Q:]) * np.exp(-2j * np.pi * n / R)
    p = np.arange(M)
    k = p // 256 + R1 * ((p // 16) % 16) + 16 * R1 * (p % 16)
    outs = []
    for rnd, hk in ((0, he), (1, ho)):
        s = lds[:Q] + partner if rnd == 0 else (lds[:Q] - partner) * np.exp(-2j * np.pi * n / R)
        G = np.fft.fft(hk) / (2 * Q)
        S0, S1 = pair_fft(s[0::2], s[1::2], R1)
        a2, b2 = pair_fft(S0 * G[k], S1 * G[k + M], R1, inverse=True)
        y = np.empty(Q, complex)
        y[0::2], y[1::2] = a2, b2
        outs.append(y)
    ye, yo = outs
    z = yo * np.exp(+2j * np.pi * n / R)                      # w_R^-m' yo[m']
    out = np.empty(N, complex)
    i = n - (P - 1 - Q)                                       # y[m' + Q] = ye - z  ->  sample m' + Q - (P - 1)
    ok = (i >= 0) & (i < N)
    out[i[ok]] = (ye - z)[ok]
    for mp in range(max(Lx, 0)):                              # y[m' + 2Q] = ye + z + fix-up  ->  sample m' + 2Q - (P - 1)
        t = np.arange(mp + 1)
        out[mp + R - (P - 1)] = ye[mp] + z[mp] + np.sum(h[t] * (sb[mp - t] - sa[mp - t]))
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for N, mg in [(16384, 15), (16380, 15), (16384, 10), (16370, 15)]:
        x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        P = N + 2 * mg
        h = taps(P, 1.3e-11, 2 * np.pi / (N * 1.46e-6))
        ref = reference_line(x, mg, h)
        out = engine_line_dif(x, mg, h)
        print("DIF  N %5d margin %2d Lx %4d  max err %.2e" % (N, mg, N + P - 1 - 32768, np.abs(out - ref).max() / np.abs(ref).max()))
