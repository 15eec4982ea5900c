"""Detector model on the MI355X (mirror of CodePython/Detector.py:19-220).

`Detector.detection` keeps the reference signature; the image work (reflect pad, source blur, bin-sum, PSF, crop) runs as
one composed separable operator in HIP (csrc/detect.hip).  Differences a user can see:
  * images are torch tensors in HBM (float32); numpy inputs are uploaded;
  * shot noise comes from a counter-based generator with an explicit seed (the reference seeds from the wall clock,
    Detector.py:113) and can be switched off (`exp_param['noise']=False`) -- the parity tests compare that image.
"""
import numpy as np
import torch

from . import _xml, ops
from ._tensors import to_dev
from .getk import getk


class Detector:
    def __init__(self, exp_dict, xml_directory=None):
        self.xmlDetectorFileName = "xmlFiles/Detectors.xml"
        self._xml_directory = xml_directory
        self.myName = ""
        self.det_param = {"myDimensions": (0, 0), "myPixelSize": 0., "myPSF": 0., "myBinsThersholds": [],
                          "myScintillatorMaterial": None, "myScintillatorThickness": 0., "photonCounting": True}
        self.mySpectralEfficiency = []
        self.beta = []
        # units (Detector.py:36-40)
        self.det_param["myDimensions_unit"] = "pixels"
        self.det_param["myPixelSize_unit"] = "um"
        self.det_param["myPSF_unit"] = "pixels"
        self.det_param["myBinsThersholds_unit"] = "keV"
        self.det_param["myScintillatorThickness_unit"] = "um"
        self._plans = {}
        self._draws = 0

    def defineCorrectValuesDetector(self):
        """Detector.py:44-76."""
        doc = _xml.parse(self._xml_directory or _xml.xml_dir(), "Detectors.xml")
        node = _xml.find_named(doc, "detector", self.myName)
        if node is None:
            raise ValueError("detector not found in xml file")
        dp = self.det_param
        dp["myDimensions"] = np.array([int(_xml.child_text(node, "dimX")), int(_xml.child_text(node, "dimY"))])
        dp["myPixelSize"] = float(_xml.child_text(node, "myPixelSize"))
        dp["myPSF"] = float(_xml.child_text(node, "myPSF"))
        if _xml.has_child(node, "myEnergyLimit"):
            self.myEnergyLimit = float(_xml.child_text(node, "myEnergyLimit"))
        if _xml.has_child(node, "photonCounting"):
            dp["photonCounting"] = bool(_xml.child_text(node, "photonCounting"))   # bool("False") is True: DET:66
        if _xml.has_child(node, "myBinsThersholds"):
            dp["myBinsThersholds"] = [float(v) for v in _xml.child_text(node, "myBinsThersholds").split(",")]
        if _xml.has_child(node, "myScintillatorMaterial"):
            dp["myScintillatorMaterial"] = _xml.child_text(node, "myScintillatorMaterial")
            dp["myScintillatorThickness"] = float(_xml.child_text(node, "myScintillatorThickness"))

    def _plan(self, shape, ov, sigma_src, device):
        key = (tuple(shape), int(ov), float(sigma_src), float(self.det_param["myPSF"]), device.index)
        plan = self._plans.get(key)
        if plan is None:
            dims = self.det_param["myDimensions"]
            plan = ops.DetectorPlan(shape[0], shape[1], ov, int(dims[0]), int(dims[1]), sigma_src,
                                    float(self.det_param["myPSF"]), device=device)
            self._plans[key] = plan
        return plan

    def detection(self, incidentWave, effectiveSourceSize, exp_param, key=None, out=None):
        """Detector.py:79-119: blur with the projected source, resample to detector pixels, PSF, shot noise.

        incidentWave: intensity on the study grid; effectiveSourceSize: projected source FWHM in study pixels;
        exp_param: needs 'overSampling'; optional 'noise' (default True) and 'seed' (default 0).
        key (not in the reference): (pointNum, ibin, kind) of the image -- the shot noise is then a function of WHAT is
        drawn (paresis_amd.ops.poisson_key) and does not depend on call order or on the number of GPUs; without it the
        draws of this detector are numbered in call order (the reference seeds from the wall clock, DET:113).
        out: optional float32 [n, n] tensor in HBM that receives the image."""
        return self.detect_many([incidentWave], effectiveSourceSize, exp_param, [out], [key])[0]

    def detect_many(self, images, effectiveSourceSize, exp_param, outs=None, keys=None):
        """detection() of several images of one energy bin (EXP:388-394): the detector operator once per image, then the
        shot noise of all of them in ONE launch."""
        sigma_src = effectiveSourceSize / 2.355 if effectiveSourceSize != 0 else 0.0     # DET:96-97
        outs = [None] * len(images) if outs is None else list(outs)
        keys = [None] * len(images) if keys is None else list(keys)
        imgs = [to_dev(im, torch.float32) for im in images]
        if imgs and all(im.shape == imgs[0].shape and im.device == imgs[0].device for im in imgs):
            res = self._plan(imgs[0].shape, exp_param["overSampling"], sigma_src, imgs[0].device).detect_many(imgs, outs)
        else:
            res = [self._plan(img.shape, exp_param["overSampling"], sigma_src, img.device).detect(img, out=o)
                   for img, o in zip(imgs, outs)]
        if exp_param.get("noise", True):
            seed = int(exp_param.get("seed", 0))
            seeds = []
            for k in keys:
                if k is None:
                    self._draws += 1
                    seeds.append((seed << 20) + self._draws)
                else:
                    seeds.append(ops.poisson_key(seed, *k))
            ops.poisson_multi(res, seeds)
        return res

    def getBeta(self, sourceSpectrum):
        """Detector.py:131-160 reads the scintillator's beta from an .xls table; here from a registered table (same walk and
        interpolation) or from the material registry."""
        from . import materials
        name = self.det_param["myScintillatorMaterial"]
        if materials.has_table(name):                    # Detector.py:139-158: the same table walk, beta column
            self.beta = materials.table_walk(name, sourceSpectrum)[1]
            return
        self.beta = [(e, materials.delta_beta(name, e)[1]) for e, _ in sourceSpectrum]

    def getSpectralEfficiency(self):
        """Detector.py:163-172: 1 - exp(-2 k beta t) per energy."""
        self.mySpectralEfficiency = []
        for energyData, betaEn in self.beta:
            k = getk(energyData * 1000)
            eff = 1 - np.exp(-2 * k * self.det_param["myScintillatorThickness"] * 1e-6 * betaEn)
            self.mySpectralEfficiency.append((energyData, float(eff)))


def resize(imageToResize, sizeX, sizeY):
    """Detector.py:185-198: identity when the sizes match, else block SUM with factor int(Nx/sizeX) on both axes."""
    return ops.resize(to_dev(imageToResize, torch.float32), int(sizeX), int(sizeY))


def create_gaussian_shape(sigma):
    """Detector.py:201-220: normalised [dim,dim] Gaussian, dim = round(3 sigma)*2+1 (banker's rounding).

    Tiny (a few taps): built on the host in float64 and returned as a float64 numpy array, like the reference."""
    dim = round(sigma * 3) * 2 + 1
    q = np.arange(0, dim) - np.floor(dim / 2)
    g1 = np.exp(-(q ** 2) / 2. / sigma ** 2)
    g = np.outer(g1, g1)
    return g / np.sum(g)
