"""Thickness-map generators used by AnalyticalSample.getMyGeometry (stand-ins for CodePython/Samples/*.py).

Input synthesis is outside the hot path: the reference's generators need cv2/imutils/skimage and a
sphere list that is not shipped.  These analytic versions keep the parameters of the XML files (radius, orientation,
mean sphere radius, number of layers, support thickness).  Membranes are synthesised on the GPU
(paresis_amd/Samples/getMembraneFromFile.py).
"""
import numpy as np


def sphere(dimX, dimY, pix_um, radius_um):
    """createSampGeom.py:15-53 (sphere centred on the image)."""
    r = radius_um / pix_um
    i = np.arange(dimX, dtype=np.float64)[:, None]
    j = np.arange(dimY, dtype=np.float64)[None, :]
    dist = (dimX / 2 - i) ** 2 + (dimY / 2 - j) ** 2
    t = np.where(dist < r * r, 2 * np.sqrt(np.clip(r * r - dist, 0, None)), 0.0)
    return (t * pix_um * 1e-6)[None].astype(np.float32), {"Sphere_radius": (radius_um, "um")}


def cylinder(dimX, dimY, pix_um, radius_um, orientation_deg):
    """createSampGeom.py:56-107: cylinder through the image centre; the reference rasterises then rotates with
    imutils.rotate, here the chord length is evaluated analytically at the rotated coordinate."""
    r = radius_um / pix_um
    th = np.deg2rad(orientation_deg)
    i = np.arange(dimX, dtype=np.float64)[:, None] - dimX / 2
    j = np.arange(dimY, dtype=np.float64)[None, :] - dimY / 2
    d = -i * np.sin(th) + j * np.cos(th)        # distance to the cylinder axis
    t = 2 * np.sqrt(np.clip(r * r - d * d, 0, None))
    return (t * pix_um * 1e-6)[None].astype(np.float32), {"Cylinder_radius": (radius_um, "um"),
                                                           "Cylinder_orientation": (orientation_deg, "degree")}
