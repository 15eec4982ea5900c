"""CPU baseline of the hot path -- TEST / BENCH INFRASTRUCTURE, not product code.

The build's own CPU restatement timed next to the GPU numbers: float64, the
reference's algorithm and operation order (every distance is a full `wavePropagation` call: pad -> FFT2 -> chirp -> IFFT2
-> crop, Experiment.py:219-252; every refraction a full `fastRefraction`, refractionFileNumba2.py:25-86), preceded by its
2-material transmission (Sample.py:248-351).

  threads = 1     stand-in for the reference: numpy.fft (pocketfft) and a Numba @jit without parallel=True are
                  single-threaded; the 2-D FFTs go through scipy.fft (pocketfft, workers=1), everything else is
                  oracle/cpu_baseline.cpp at one thread
  threads = all   the fair CPU ceiling: scipy.fft workers=threads, OpenMP over rows in cpu_baseline.cpp

Golden-checked against the reference's own vectors in tests/test_oracle_golden.py.  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MARGIN = 15


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcpu_baseline.so")
        if not os.path.exists(path):
            raise RuntimeError("CPU baseline not built: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        vp, dp, ci, cdbl = ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_double
        lib.cb_max_threads.restype = ci
        lib.cb_wave_pad.argtypes = [vp, dp, dp, ci, ci, ci, cdbl, cdbl, ci, vp, ci]
        lib.cb_wave_pad.restype = None
        lib.cb_chirp.argtypes = [vp, ci, ci, cdbl, cdbl, cdbl, ci]
        lib.cb_chirp.restype = None
        lib.cb_crop_abs2.argtypes = [vp, ci, ci, ci, vp, ci]
        lib.cb_crop_abs2.restype = None
        lib.cb_pad_reflect.argtypes = [vp, ci, ci, ci, vp, ci]
        lib.cb_pad_reflect.restype = None
        lib.cb_fast_refraction.argtypes = [vp, vp, ci, ci, cdbl, cdbl, cdbl, cdbl, vp, ci]
        lib.cb_fast_refraction.restype = ci
        lib.cb_refraction.argtypes = [vp, dp, dp, ci, ci, ci, cdbl, cdbl, cdbl, cdbl, cdbl, cdbl, vp, ci]
        lib.cb_refraction.restype = ci
        _LIB = lib
    return _LIB


def max_threads():
    return int(_lib().cb_max_threads())


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _maps(geometry):
    g = np.ascontiguousarray(geometry, dtype=np.float32)
    ptr = (ctypes.c_void_p * g.shape[0])(*[g[m].ctypes.data for m in range(g.shape[0])])
    return g, ptr


def _dv(v):
    a = np.ascontiguousarray(v, dtype=np.float64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def k_getk(energy_keV):          # getk.py:16-19, called as getk(Energy*1000) (EXP:239): the association order matters at k z ~ 1e11 rad
    energy_eV = energy_keV * 1000
    return 2 * np.pi * energy_eV * 1.6e-19 / (6.626e-34 * 2.998e8)


def k_sample(energy_keV):        # Sample.py:265
    return 2 * np.pi * energy_keV * 1000 * 1.6e-19 / (6.626e-34 * 2.998e8)


def k_refraction(energy_keV):    # refractionFileNumba2.py:47-48
    lam = 6.626 * 1e-34 * 2.998e8 / (energy_keV * 1000 * 1.6e-19)
    return 2 * np.pi / lam


def fresnel_intensity(geometry, delta, beta, amp, z, energy_keV, M, pix_um, threads=1):
    """|wavePropagation(setWave(amp), z)|^2 on the study grid (Experiment.py:334-351), float64 [Nx][Ny]."""
    from scipy import fft as sfft
    lib = _lib()
    g, ptr = _maps(geometry)
    nmat, Nx, Ny = g.shape
    Px, Py = Nx + 2 * MARGIN, Ny + 2 * MARGIN
    d, dptr = _dv(delta)
    b, bptr = _dv(beta)
    w = np.empty((Px, Py), dtype=np.complex128)
    lib.cb_wave_pad(ptr, dptr, bptr, nmat, Nx, Ny, float(amp), k_sample(energy_keV), MARGIN, w.ctypes.data, threads)
    w = sfft.fft2(w, workers=threads, overwrite_x=True)
    h = pix_um * 1e-6
    lib.cb_chirp(w.ctypes.data, Px, Py, z / (2 * k_getk(energy_keV) * M), 2 * np.pi / (Nx * h), 2 * np.pi / (Ny * h), threads)
    w = sfft.ifft2(w, workers=threads, overwrite_x=True)
    w = np.ascontiguousarray(w)
    out = np.empty((Nx, Ny), dtype=np.float64)
    lib.cb_crop_abs2(w.ctypes.data, Nx, Ny, MARGIN, out.ctypes.data, threads)
    return out


def wave_propagation(wave, z, energy_keV, M, pix_um, threads=1):
    """Experiment.wavePropagation (Experiment.py:219-252) on an explicit complex wave: the golden-vector entry."""
    from scipy import fft as sfft
    if z == 0:
        return wave
    lib = _lib()
    wave = np.ascontiguousarray(wave, dtype=np.complex128)
    Nx, Ny = wave.shape
    Px, Py = Nx + 2 * MARGIN, Ny + 2 * MARGIN
    w = np.empty((Px, Py), dtype=np.complex128)
    lib.cb_pad_reflect(wave.ctypes.data, Nx, Ny, MARGIN, w.ctypes.data, threads)
    w = sfft.fft2(w, workers=threads, overwrite_x=True)
    h = pix_um * 1e-6
    k = k_getk(energy_keV)
    lib.cb_chirp(w.ctypes.data, Px, Py, z / (2 * k * M), 2 * np.pi / (Nx * h), 2 * np.pi / (Ny * h), threads)
    w = sfft.ifft2(w, workers=threads, overwrite_x=True)
    return np.exp(1j * k * z / M) * w[MARGIN:Px - MARGIN, MARGIN:Py - MARGIN]


def fast_refraction(intensity, phi, z, energy_keV, M, pix_um, threads=1):
    """fastRefraction v2 on explicit (I, phi): the golden-vector entry.  Returns the cropped image only."""
    lib = _lib()
    I = np.ascontiguousarray(intensity, dtype=np.float64)
    ph = np.ascontiguousarray(phi, dtype=np.float64)
    out = np.empty(I.shape, dtype=np.float64)
    bad = lib.cb_fast_refraction(I.ctypes.data, ph.ctypes.data, I.shape[0], I.shape[1], k_refraction(energy_keV), float(z),
                                 float(M), float(pix_um), out.ctypes.data, threads)
    if bad:
        raise Exception("The calculated intensity refractive includes some nans or insane values")
    return out


def refraction_intensity(geometry, delta, beta, I0, z, energy_keV, M, pix_um, threads=1):
    """fastRefraction(setWaveRT(I0), z) (Experiment.py:463-466), float64 [Nx][Ny]."""
    lib = _lib()
    g, ptr = _maps(geometry)
    nmat, Nx, Ny = g.shape
    d, dptr = _dv(delta)
    b, bptr = _dv(beta)
    out = np.empty((Nx, Ny), dtype=np.float64)
    bad = lib.cb_refraction(ptr, dptr, bptr, nmat, Nx, Ny, float(I0), k_sample(energy_keV), k_refraction(energy_keV),
                            float(z), float(M), float(pix_um), out.ctypes.data, threads)
    if bad:
        raise Exception("The calculated intensity refractive includes some nans or insane values")
    return out


def time_units(geometry, delta, beta, I0, distances, energy_keV, M, pix_um, threads):
    """The step's units (one Fresnel propagation + one refraction per distance, each with its transmission) on `threads`
    threads.  Returns (seconds_fresnel, seconds_refraction, [fresnel images], [refraction images])."""
    amp = float(np.sqrt(I0))
    F, R = [], []
    t0 = time.perf_counter()
    for z in distances:
        F.append(fresnel_intensity(geometry, delta, beta, amp, z, energy_keV, M, pix_um, threads))
    t1 = time.perf_counter()
    for z in distances:
        R.append(refraction_intensity(geometry, delta, beta, I0, z, energy_keV, M, pix_um, threads))
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, F, R
