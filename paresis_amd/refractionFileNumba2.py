"""Ray-tracing propagator, v2 constants (mirror of CodePython/refractionFileNumba2.py:14-86,198-263).

Same function names and argument order as the reference; arrays are torch tensors in HBM (numpy inputs are uploaded) and
the work runs in csrc/refract.hip.  The dark-field variant (fastRefractionDF, RF2:88-196) is the next row of the scope
table and raises until it is built.
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


def fastRefractionDF(intensityRefracted, phi, propagationDistance, Energy, magnification, studyPixelSize, darkField):
    raise PsxError("fastRefractionDF (refractionFileNumba2.py:88-196) is not built yet: SURVEY.md section 8f-2")


def fastloopNumba(Nx, Ny, intensityRefracted, intensityRefracted2, Dy, Dx, DxFloor=None, DyFloor=None):
    """RF2:198-263, same argument order (note Dy before Dx); DxFloor/DyFloor are unused, as in the reference.
    Accumulates into and returns intensityRefracted2 (a float32 tensor in HBM)."""
    I = to_dev(intensityRefracted, torch.float32)
    I2 = to_dev(intensityRefracted2, torch.float32)
    if tuple(I.shape) != (Nx, Ny):
        raise PsxError("intensityRefracted has shape %s, expected (%d, %d)" % (tuple(I.shape), Nx, Ny))
    return ops.fastloop(I, to_dev(Dx, torch.float32), to_dev(Dy, torch.float32), I2)
