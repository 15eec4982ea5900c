"""Ray-tracing propagator, v1 constants (mirror of CodePython/refractionFileNumba.py:11-135): margin 10 (RF1:36) and a
fixed clamp |D| > 1e3 (RF1:46-49) instead of v2's margin 15 and |D| > Nx,Ny.  Same kernels as v2."""
from .refractionFileNumba2 import _fast_refraction, fastloopNumba  # noqa: F401

MARGIN = 10   # RF1:36


def fastRefraction(intensityRefracted, phi, propagationDistance, Energy, magnification, studyPixelSize):
    """RF1:11-68."""
    return _fast_refraction(intensityRefracted, phi, propagationDistance, Energy, magnification, studyPixelSize,
                            MARGIN, 1e3)
