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


_lists = {}


def load_sphere_list(path=SPHERE_FILE):
    """[[y, x, r], ...] in file units: the JSON file of the reference when it exists, else the synthetic stand-in.
    Read (or synthesised) once per process: main.py asks for it at every membrane position."""
    key = (os.path.abspath(path), os.path.getmtime(path)) if path and os.path.exists(path) else None
    lst = _lists.get(key)
    if lst is None:
        if key is not None:
            with open(path) as f:
                lst = np.asarray(json.load(f), dtype=np.float64)
        else:
            lst = synth.sphere_list()
        lst.setflags(write=False)
        _lists[key] = lst
    return lst


def stitched_list(sphere_list, dimX, dimY, pixSize, meanSphereRadius):
    """getMembraneFromFile.py:79-137 (host part, the same for every position): scale the list to the requested mean
    radius, centre it, stitch copies until it covers the grid.  Returns (margin, margin2, par [n,3] in um, sizeX, sizeY)."""
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
    return margin, margin2, par, sizeX, sizeY


def layer_offsets(rand_state, nbOfLayers, margin2, sizeX, sizeY, pixSize, dimX, dimY):
    """getMembraneFromFile.py:139-140: one integer offset pair per layer, drawn x then y."""
    offs = []
    for _ in range(int(nbOfLayers)):
        Offsetx = rand_state.randint(margin2, sizeX / pixSize - dimX - margin2)
        Offsety = rand_state.randint(margin2, sizeY / pixSize - dimY - margin2)
        offs.append((int(Offsetx), int(Offsety)))
    return offs


def sphere_layers(sphere_list, dimX, dimY, pixSize, meanSphereRadius, nbOfLayers, rand_state):
    """getMembraneFromFile.py:79-142 (host part): returns (margin, margin2, [(xfloat, yfloat, radFloat)] per layer)."""
    margin, margin2, par, sizeX, sizeY = stitched_list(sphere_list, dimX, dimY, pixSize, meanSphereRadius)
    layers = [(par[:, 1] / pixSize - ox, par[:, 0] / pixSize - oy, par[:, 2] / pixSize)
              for ox, oy in layer_offsets(rand_state, nbOfLayers, margin2, sizeX, sizeY, pixSize, dimX, dimY)]
    return margin, margin2, layers


class _MembranePlan:
    """The stitched list of one (sphere list, grid, pixel size, mean radius) resident on one GPU (psx_membrane_plan)."""

    def __init__(self, lst, dimX, dimY, pixSize, meanSphereRadius):
        self.margin, self.margin2, par, self.sizeX, self.sizeY = stitched_list(lst, dimX, dimY, pixSize, meanSphereRadius)
        x, y, r = (np.ascontiguousarray(v, dtype=np.float64) for v in (par[:, 1] / pixSize, par[:, 0] / pixSize, par[:, 2] / pixSize))
        self.h = c_void_p(None)
        check(lib().psx_membrane_plan_create(x.ctypes.data_as(_DP), y.ctypes.data_as(_DP), r.ctypes.data_as(_DP), len(r),
                                             ctypes.byref(self.h)), "psx_membrane_plan_create")

    def __del__(self):
        try:
            if self.h:
                lib().psx_membrane_plan_destroy(self.h)
        except Exception:
            pass


_plans = {}


def _plan_for(lst, dimX, dimY, pixSize, meanSphereRadius, dev):
    import zlib
    arr = np.asarray(lst)
    # a read-only array (what load_sphere_list hands out) is identified by its address; anything else by its content
    ident = (id(lst),) if isinstance(lst, np.ndarray) and not lst.flags.writeable else (
        zlib.crc32(np.ascontiguousarray(arr, dtype=np.float64).tobytes()),)
    key = (dev.index, arr.shape, ident, dimX, dimY, float(pixSize), float(meanSphereRadius))
    plan = _plans.get(key)
    if plan is None:
        if len(_plans) >= 8:
            _plans.pop(next(iter(_plans)))
        plan = _plans[key] = _MembranePlan(arr, dimX, dimY, pixSize, meanSphereRadius)
        plan.source = lst          # keeps the array alive, so its id stays unique
    return plan


def getMembraneSegmentedFromFile(sample, dimX, dimY, pixSize, pointNum, supportThickness, seed=None, sphere_list=None,
                                 stacked=False):
    """getMembraneFromFile.py:60-171.  Returns ([membrane_m, support_m] float32 tensors in HBM, parameters_dic).

    main.py:64-65 calls this for every membrane position: the scaled, stitched list is the same each time, only the layer
    offsets change, so the list is uploaded and binned once (psx_membrane_plan) and a position costs the draws of its
    offsets plus one kernel per layer."""
    dimX, dimY = int(dimX), int(dimY)
    lst = load_sphere_list() if sphere_list is None else sphere_list
    rs = np.random.RandomState(synth.position_seed(pointNum) if seed is None else int(seed))
    dev = device()
    plan = _plan_for(lst, dimX, dimY, pixSize, sample.myMeanSphereRadius, dev)
    offs = layer_offsets(rs, sample.myNbOfLayers, plan.margin2, plan.sizeX, plan.sizeY, pixSize, dimX, dimY)
    # `stacked`: both maps are the two planes of ONE [2, dimX, dimY] tensor (what the kernels take as a material stack), so
    # the caller needs no torch.stack copy; geom[0].base-style access: the stack is returned as the third list entry
    stack = torch.empty((2, dimX, dimY), dtype=torch.float32, device=dev) if stacked else None
    membrane = stack[0] if stacked else torch.empty((dimX, dimY), dtype=torch.float32, device=dev)
    support = stack[1] if stacked else torch.empty((dimX, dimY), dtype=torch.float32, device=dev)
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    # every layer and the uniform support map in ONE launch of the library (the position loop launches no PyTorch kernel)
    ox = (ctypes.c_int * max(1, len(offs)))(*[int(o[0]) for o in offs])
    oy = (ctypes.c_int * max(1, len(offs)))(*[int(o[1]) for o in offs])
    check(lib().psx_membrane_layers_f32(plan.h, len(offs), ox, oy, dimX, dimY, plan.margin, plan.margin2,
                                        c_double(pixSize * 1e-6), 0, c_void_p(membrane.data_ptr()), c_void_p(support.data_ptr()),
                                        ctypes.c_float(float(supportThickness) * 1e-6), st), "psx_membrane_layers_f32")
    parameters_dic = {'Average sphere radius': (sample.myMeanSphereRadius, 'um'),
                      'Number of layers': (sample.myNbOfLayers, ''),
                      'Support total thickness': (supportThickness, 'um')}
    return ([membrane, support, stack] if stacked else [membrane, support]), parameters_dic


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
