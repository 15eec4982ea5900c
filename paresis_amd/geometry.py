"""Thickness-map generators used by AnalyticalSample.getMyGeometry (stand-ins for CodePython/Samples/*.py).

Input synthesis sits just before the hot path.  The reference's deterministic generators
(sphere, two spheres in a cylinder, the editable example, the image-folder loader) are restated here as vectorised
numpy and pinned by goldens the reference itself produced (tests/golden/geometry.npz).  The two generators that rotate a
raster with imutils/cv2 (absent here) -- cylinder and parallelepiped -- evaluate the rotated shape analytically / with
scipy's bilinear resampling instead and are NOT pinned.  Membranes are synthesised on the GPU
(paresis_amd/Samples/getMembraneFromFile.py).
"""
import glob
import os

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


def _sphere_patch(radius_px):
    """The square patch both two-sphere generators paste (createSampGeom.py:137-149, 210-222): side 2*ceil(r), chord
    length 2*sqrt(r^2 - d^2) about the patch centre."""
    half = int(np.ceil(radius_px))
    size = 2 * half
    i = np.arange(size, dtype=np.float64)[:, None]
    j = np.arange(size, dtype=np.float64)[None, :]
    di, dj = size / 2 - i, size / 2 - j
    dist = di ** 2 + dj ** 2
    r2 = radius_px ** 2
    # same operation order as the reference's expression r^2 - (size/2 - j)^2 - (size/2 - i)^2
    patch = np.where(dist < r2, 2 * np.sqrt(np.clip(r2 - dj ** 2 - di ** 2, 0, None)), 0.0)
    return patch, half


def spheres_in_cylinder(dimX, dimY, pix_um):
    """createSampGeom.py:108-171: a vertical cylinder (radius 1000 um) holding two spheres (radius 500 um) on its axis;
    three materials [sphere 1, sphere 2, cylinder minus spheres], metres."""
    r0_um = 500.0
    r2_um = 2 * r0_um
    posY = dimY // 2
    pos1 = int(np.round(r0_um * 3 / pix_um))
    pos2 = int(np.round(r0_um * 7 / pix_um))
    r = r0_um / pix_um
    if 2 * r > dimX or 2 * r > dimY:
        raise Exception("The sample is too big for the detector field of view (increase dimX, dimY)")
    patch, half = _sphere_patch(r)
    R = r2_um / pix_um
    if 2 * R > dimX or 2 * R > dimY:
        raise Exception("The sample is too big for the detector field of view (increase dimX, dimY)")
    j = np.arange(dimY, dtype=np.float64)
    col = np.where(np.abs(dimY / 2 - j) < R, 2 * np.sqrt(np.clip(R ** 2 - (dimY / 2 - j) ** 2, 0, None)), 0.0)
    sample = np.zeros((3, dimX, dimY))
    sample[0, pos1 - half:pos1 + half, posY - half:posY + half] = patch     # raises like the reference if it does not fit
    sample[1, pos2 - half:pos2 + half, posY - half:posY + half] = patch
    sample[2] = np.broadcast_to(col, (dimX, dimY)) - sample[0] - sample[1]
    params = {"Spheres_radius": (r0_um, "um"), "Cylinder_radius": (r2_um, "um"),
              "Position_Sphere_1": (pos1 * pix_um, "um"), "Position_Sphere_2": (pos2 * pix_um, "um")}
    return sample * pix_um * 1e-6, params


def spheres_in_parallelepiped(dimX0, dimY0, pix_um):
    """createSampGeom.py:173-250: a rounded-edge slab (1000 um) holding two spheres, tilted by 15 degrees.  The
    reference tilts the raster with imutils.rotate (cv2.warpAffine, bilinear); here scipy.ndimage does the same bilinear
    rotation about the image centre -- same shape, not bit-identical (cv2 interpolates with 5-bit fixed-point weights)."""
    from scipy import ndimage
    r0_um, tilt = 500.0, 15.0
    r2_um = 2 * r0_um
    margin = max(dimX0, dimY0) // 2
    dimX, dimY = dimX0 + 2 * margin, dimY0 + 2 * margin
    posY, pos1, pos2 = dimY // 2, dimX * 2 // 5, dimX * 3 // 5
    r = r0_um / pix_um
    if 2 * r > dimX or 2 * r > dimY:
        raise Exception("The sample is too big for the detector field of view (increase dimX, dimY)")
    patch, half = _sphere_patch(r)
    R = r2_um / pix_um
    if 2 * R > dimX or 2 * R > dimY:
        raise Exception("The sample is too big for the detector field of view (increase dimX, dimY)")
    tube = np.zeros((dimX, dimY))
    for j in range(min(dimX, dimY)):                     # the reference loops j over dimX while indexing columns
        o = j - dimY / 2
        if abs(o) < R * 3 / 4:
            tube[:, j] = R * 2
        if R > o >= R * 3 / 4:
            tube[:, j] = R / 2 * 3 + 2 * np.sqrt(max((R / 4) ** 2 - (j - (dimY / 2 + R * 3 / 4)) ** 2, 0.0))
        if -R < o <= -R * 3 / 4:
            tube[:, j] = R / 2 * 3 + 2 * np.sqrt(max((R / 4) ** 2 - (j - (dimY / 2 - R * 3 / 4)) ** 2, 0.0))
    sample = np.zeros((3, dimX, dimY))
    sample[0, pos1 - half:pos1 + half, posY - half:posY + half] = patch
    sample[1, pos2 - half:pos2 + half, posY - half:posY + half] = patch
    sample[2] = tube - sample[0] - sample[1]
    sample = np.stack([ndimage.rotate(m, tilt, reshape=False, order=1, mode="constant") for m in sample])
    sample = sample[:, margin:dimX - margin, margin:dimY - margin]
    params = {"Spheres_radius": (r0_um, "um"), "Parallelepipede_size": (r2_um, "um"),
              "Position_Sphere_1": (pos1 * pix_um, "um"), "Position_Sphere_2": (pos2 * pix_um, "um")}
    return sample * pix_um * 1e-6, params


def your_sample_geometry(dimX, dimY):
    """createSampGeom.py:289-318, the example users edit: one material, 5 um everywhere."""
    thickness = 5 * 1e-6
    params = {"geometry thickness": (thickness, "um"), "geometry other parameter": ("unitlessParameter", "")}
    return np.ones((1, dimX, dimY)) * thickness, params


def load_sample_geometry_from_images(folder, dimX, dimY, pix_um):
    """createSampGeom.py:253-286: every .tif / .tiff / .edf of the folder, sorted by path within each extension group
    as the reference concatenates them, is one material's thickness map in metres."""
    from .InputOutput.pagailleIO import openImage
    paths = glob.glob(folder + "/*.tif") + glob.glob(folder + "/*.tiff") + glob.glob(folder + "/*.edf")
    paths.sort()
    if not paths:
        raise Exception("The sample geometry you are trying to load does not exist or is incorrectly named:", folder)
    return [openImage(p) for p in paths], {"myGeometryFolder": (folder, "")}
