"""Membrane thickness maps (mirror of CodePython/Samples/getMembraneFromFile.py:18-171) with the sphere splat on the GPU.

`getMembraneSegmentedFromFile` keeps the reference's signature and host logic (scale the sphere list to the requested
mean radius, stitch copies until it covers the grid, one random offset per layer) and hands each layer's spheres to
`psx_membrane_f32` (csrc/membrane.hip), which replaces the interpreted triple loop of :143-159.

Two deliberate differences:
  * the layer offsets come from a generator seeded by the membrane position (`seed(pointNum) = 1000 + pointNum`, or
    `seed=`), because the reference draws them from numpy's unseeded global state (:139-140) and results would then
    depend on how positions are sharded over GPUs;
  * the sphere list Samples/Membranes/CuSn.txt is not distributed with the reference (.MISSING_LARGE_BLOBS); when the file
    is absent a seeded synthetic list with the same layout is used (paresis_amd.synth.sphere_list).
"""
import ctypes
import json
import os
from ctypes import c_double, c_void_p

import numpy as np
import torch

from .. import synth
from .._lib import check, lib
from .._tensors import device

SPHERE_FILE = "Samples/Membranes/CuSn.txt"
_DP = ctypes.POINTER(c_double)


def load_sphere_list(path=SPHERE_FILE):
    """[[y, x, r], ...] in file units: the JSON file of the reference when it exists, else the synthetic stand-in."""
    if path and os.path.exists(path):
        with open(path) as f:
            return np.asarray(json.load(f), dtype=np.float64)
    return synth.sphere_list()


def sphere_layers(sphere_list, dimX, dimY, pixSize, meanSphereRadius, nbOfLayers, rand_state):
    """getMembraneFromFile.py:79-142 (host part): returns (margin, margin2, [(xfloat, yfloat, radFloat)] per layer)."""
    margin = int(np.ceil(10 * meanSphereRadius / pixSize))
    margin2 = int(np.floor(margin / 2))
    corrFactor = meanSphereRadius / 12.8
    sizeX = int(np.floor(8102)) * corrFactor + meanSphereRadius
    sizeY = int(np.floor(9740)) * corrFactor + meanSphereRadius
    par = np.asarray(sphere_list, dtype=np.float64) * corrFactor
    par[:, 1] += sizeX / 2
    par[:, 0] += sizeY / 2
    first, sx0, sy0 = par.copy(), sizeX, sizeY
    while sizeX / pixSize - dimX < 0:
        print("segmented membrane too small: proceeding with stitching along x")
        extra = first.copy()
        extra[:, 1] += sizeX
        par = np.concatenate((par, extra), axis=0)
        sizeX += sx0
    first = par.copy()
    while sizeY / pixSize - dimY < 0:
        print("segmented membrane too small: proceeding with stitching along y")
        extra = first.copy()
        extra[:, 0] += sizeY
        par = np.concatenate((par, extra), axis=0)
        sizeY += sy0
    layers = []
    for _ in range(int(nbOfLayers)):
        Offsetx = rand_state.randint(margin2, sizeX / pixSize - dimX - margin2)
        Offsety = rand_state.randint(margin2, sizeY / pixSize - dimY - margin2)
        layers.append((par[:, 1] / pixSize - Offsetx, par[:, 0] / pixSize - Offsety, par[:, 2] / pixSize))
    return margin, margin2, layers


def getMembraneSegmentedFromFile(sample, dimX, dimY, pixSize, pointNum, supportThickness, seed=None, sphere_list=None):
    """getMembraneFromFile.py:60-171.  Returns ([membrane_m, support_m] float32 tensors in HBM, parameters_dic)."""
    dimX, dimY = int(dimX), int(dimY)
    lst = load_sphere_list() if sphere_list is None else sphere_list
    rs = np.random.RandomState(synth.position_seed(pointNum) if seed is None else int(seed))
    margin, margin2, layers = sphere_layers(lst, dimX, dimY, pixSize, sample.myMeanSphereRadius, sample.myNbOfLayers, rs)
    dev = device()
    membrane = torch.zeros((dimX, dimY), dtype=torch.float32, device=dev)
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    for li, (xf, yf, rad) in enumerate(layers):
        xf, yf, rad = (np.ascontiguousarray(v, dtype=np.float64) for v in (xf, yf, rad))
        check(lib().psx_membrane_f32(xf.ctypes.data_as(_DP), yf.ctypes.data_as(_DP), rad.ctypes.data_as(_DP), len(rad),
                                     dimX, dimY, margin, margin2, c_double(pixSize * 1e-6), 1 if li else 0,
                                     c_void_p(membrane.data_ptr()), st), "psx_membrane_f32")
    support = torch.full((dimX, dimY), float(supportThickness) * 1e-6, dtype=torch.float32, device=dev)
    parameters_dic = {'Average sphere radius': (sample.myMeanSphereRadius, 'um'),
                      'Number of layers': (sample.myNbOfLayers, ''),
                      'Support total thickness': (supportThickness, 'um')}
    return [membrane, support], parameters_dic


def getMembraneFromFile(myMembraneFile, studyDimensions, numPoint, supportThickness):
    """getMembraneFromFile.py:18-57: load pre-rendered thickness maps (sorted *.tif / *.tiff / *.edf of a folder)."""
    import glob
    from ..InputOutput.pagailleIO import openImage
    paths = sorted(glob.glob(myMembraneFile + '/*.tif') + glob.glob(myMembraneFile + '/*.tiff') + glob.glob(myMembraneFile + '/*.edf'))
    thickness = np.asarray(openImage(paths[numPoint]), dtype=np.float32)
    if studyDimensions[0] != thickness.shape[0] or studyDimensions[1] != thickness.shape[1]:
        raise ValueError("The membrane you are trying to load does not have the correct dimensions")
    geom = [thickness, np.ones(thickness.shape, dtype=np.float32) * supportThickness * 1e-6]
    return geom, {"Membrane geometry folder": (paths, ''), "Support thickness": (supportThickness, 'um')}
