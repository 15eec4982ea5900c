"""Ray-tracing propagator, v2 constants (mirror of CodePython/refractionFileNumba2.py:14-86,198-263).

Same function names and argument order as the reference; arrays are torch tensors in HBM (numpy inputs are uploaded) and
the work runs in csrc/refract.hip.  The dark-field variant fastRefractionDF (RF2:88-196) reuses the same kernels plus a variable-width
Gaussian re-splat (csrc/darkfield.hip).
"""
import numpy as np
import torch

from . import ops
from ._lib import PsxError
from ._tensors import to_dev
from .getk import k_refraction

MARGIN = 15   # RF2:50


def gaussian_shape(sigma):
    """RF2:14-23 (same construction as Detector.create_gaussian_shape)."""
    from .Detector import create_gaussian_shape
    return create_gaussian_shape(sigma)


def _fast_refraction(intensityRefracted, phi, propagationDistance, Energy, magnification, studyPixelSize, margin, clamp,
                     check=True):
    # a contiguous float32 tensor already in HBM is used (and mutated) in place; anything else is uploaded
    I = to_dev(intensityRefracted, torch.float32)
    mutate_host = intensityRefracted if isinstance(intensityRefracted, np.ndarray) else None
    Nx, Ny = I.shape
    k = k_refraction(Energy)
    h = studyPixelSize * 1e-6
    dscale = propagationDistance / k / (h * magnification) / h          # RF2:54-56 with the 1/h of np.gradient
    lim = (Nx, Ny) if clamp is None else (clamp, clamp)
    out, Dx, Dy = ops.refract((Nx, Ny), None, dscale, lim, margin=margin, I_in=I, phi_in=to_dev(phi, torch.float64),
                              want_D=True, I_mut=I)
    if mutate_host is not None:
        mutate_host[...] = I.cpu().numpy()                              # in-place zeroing of clamped rays, RF2:61-62
    if check:
        ops.check_status(out.device, "fastRefraction")                   # RF2:81-82
    return out, Dx, Dy


def fastRefraction(intensityRefracted, phi, propagationDistance, Energy, magnification, studyPixelSize):
    """RF2:25-86.  Returns (intensity after propagation [Nx,Ny], Dx, Dy padded [Nx+30,Ny+30]); zeroes the clamped
    entries of `intensityRefracted` in place like the reference."""
    return _fast_refraction(intensityRefracted, phi, propagationDistance, Energy, magnification, studyPixelSize,
                            MARGIN, None)


def fastRefractionDF(intensityRefracted, phi, propagationDistance, Energy, magnification, studyPixelSize, darkField,
                     darkFieldMax=None, check=True, mutate=True, want_D=True):
    """RF2:88-196: refraction with a dark-field (small-angle scattering) width map `darkField` in radians.

    The intensity is split where the dark field is / is not zero (RF2:147-150), both parts are refracted with the same
    kernels as fastRefraction, and the dark-field part is spread by a per-pixel Gaussian of sigma = DF/2 pixels
    (csrc/darkfield.hip).  Returns (intensity [Nx,Ny], Dx, Dy) with Dx, Dy padded by ceil(6*max DF) like the reference.
    The shapes of Dx, Dy depend on the largest width: `darkFieldMax` (extension: the exact maximum of `darkField`, in
    radians -- Experiment knows it per sample and energy) makes the call fully asynchronous; without it the maximum is read
    back from the GPU (one synchronisation).  Every array operation is a kernel of the library: the conversion to pixels,
    the DF > Nx/4 rule, the split and the patch table in one pass (psx_darkfield_split_f32), two refractions, the re-splat.
    The matplotlib pop-ups of RF2:152-167 are not reproduced.  mutate=False (extension; the chain's caller passes a temporary it
    never reads again): the clamped rays are not zeroed in `intensityRefracted` (RF2:128-129), which saves one pass.
    want_D=False (extension; the chain's sample image, EXP:473, which drops the maps): Dx, Dy are not materialised (two padded
    maps cleared and stored per call) and come back as None."""
    I = to_dev(intensityRefracted, torch.float32)
    mutate_host = intensityRefracted if isinstance(intensityRefracted, np.ndarray) else None
    Nx, Ny = I.shape
    k = k_refraction(Energy)
    h = studyPixelSize * 1e-6
    dscale = propagationDistance / k / (h * magnification) / h
    den = studyPixelSize * 1e-6 * magnification           # RF2:114 rad -> pixels: DF * z / den, evaluated in that order
    limit = Nx / 4                                                                         # RF2:135
    I_nodf, I_df, DF, prep, words = ops.darkfield_split(I, to_dev(darkField, torch.float64), propagationDistance, den, limit)
    if darkFieldMax is not None and float(darkFieldMax) * propagationDistance / den <= limit:
        maxDF = maxDFc = float(darkFieldMax) * propagationDistance / den    # the rule removes nothing: both maxima are the known one
    else:
        maxDF, maxDFc = ops.darkfield_maxima(words)
    margin2 = int(np.ceil(maxDF * 6))                                                      # RF2:117
    phi64 = to_dev(phi, torch.float64)
    # the tile kernels need a margin >= their gather halo; a wider margin only changes deposits that the crop removes
    m = max(margin2, 8)
    need_D = want_D or margin2 < 1
    I2, Dxp, Dyp = ops.refract((Nx, Ny), None, dscale, (Nx, Ny), margin=m, I_in=I_nodf, phi_in=phi64, want_D=need_D,
                               I_mut=I_nodf)
    I2DF, _, _ = ops.refract((Nx, Ny), None, dscale, (Nx, Ny), margin=m, I_in=I_df, phi_in=phi64, I_mut=I_df)
    if not need_D:
        Dx = Dy = None
    elif margin2 == m:
        Dx, Dy = Dxp, Dyp
    else:
        Dx, Dy = ops.repad(Dxp, m, margin2, (Nx, Ny)), ops.repad(Dyp, m, margin2, (Nx, Ny))
    if margin2 < 1:
        # margin 0 (dark field identically zero): the scatter's border rules act on the image edge itself, so the literal
        # loop is used on the un-padded arrays (RF2:235-262)
        I2 = ops.fastloop(I_nodf, Dx, Dy, ops.fill(torch.empty_like(I), 0.0))
        I2DF = ops.fastloop(I_df, Dx, Dy, ops.fill(torch.empty_like(I), 0.0))
        if not want_D:
            Dx = Dy = None
    if mutate:
        ops.darkfield_merge(I, I_nodf, I_df)             # clamped rays zeroed in the caller's array (RF2:128-129)
        if mutate_host is not None:
            mutate_host[...] = I.cpu().numpy()
    R = int(round(1.5 * maxDFc)) + 1
    out = ops.darkfield_blur_prepared(I2DF, DF, prep, I2, R)        # the NaN / inf scan of RF2:190-193 rides on its stores
    if check:                                            # (a chain that defers the check reads the status word once, at its end)
        ops.check_status(out.device, "fastRefractionDF")     # RF2:190-193
    return out, Dx, Dy


def fastloopNumba(Nx, Ny, intensityRefracted, intensityRefracted2, Dy, Dx, DxFloor=None, DyFloor=None):
    """RF2:198-263, same argument order (note Dy before Dx); DxFloor/DyFloor are unused, as in the reference.
    Accumulates into and returns intensityRefracted2 (a float32 tensor in HBM)."""
    I = to_dev(intensityRefracted, torch.float32)
    I2 = to_dev(intensityRefracted2, torch.float32)
    if tuple(I.shape) != (Nx, Ny):
        raise PsxError("intensityRefracted has shape %s, expected (%d, %d)" % (tuple(I.shape), Nx, Ny))
    return ops.fastloop(I, to_dev(Dx, torch.float32), to_dev(Dy, torch.float32), I2)
