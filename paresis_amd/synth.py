"""Seeded synthetic thickness maps for tests, goldens and bench.py.

These stand in for the reference's input-synthesis layer (CodePython/Samples/getMembraneFromFile.py:60-171
and CodePython/Samples/createSampGeom.py:56-107), which needs a sphere list that is not shipped
(.MISSING_LARGE_BLOBS) and is out of scope for the hot path.  Everything here is plain numpy on the host so
that the oracle and the HIP path see bit-identical float32 inputs; seed(pointNum) = 1000 + pointNum makes a
membrane position independent of how positions are sharded over GPUs.
"""
import numpy as np

# order-of-magnitude physical index decrements at 52 keV (not from the xls tables)
DELTA_BETA_52KEV = {
    "CuSn": (6.2e-7, 4.0e-9),
    "PMMA": (9.87e-8, 4.5e-11),
    "Nylon": (9.5e-8, 4.2e-11),
    "air": (1.0e-10, 1.0e-15),
    "C": (1.5e-7, 6.0e-11),
}


def position_seed(pointNum):
    return 1000 + int(pointNum)


def sphere_membrane(Nx, Ny, pix_m, pointNum=0, coverage=0.5, rmin=3.0, rmax=8.0, dtype=np.float32):
    """Thickness map (metres) of randomly placed spheres: sum over spheres of 2*sqrt(r^2-d^2)*pix.

    Mirrors the splat of getMembraneFromFile.py:143-159 (thickness of a sphere along the beam), with a
    seeded RNG instead of the reference's unseeded np.random.randint offsets (:139-140).
    """
    rng = np.random.Generator(np.random.PCG64(position_seed(pointNum)))
    rbar = 0.5 * (rmin + rmax)
    n_s = max(1, int(coverage * Nx * Ny / (np.pi * rbar * rbar)))
    cx = rng.uniform(0, Nx, n_s)
    cy = rng.uniform(0, Ny, n_s)
    rr = rng.uniform(rmin, rmax, n_s)
    R = int(np.ceil(rmax)) + 1
    off = np.arange(-R, R + 1)
    ox, oy = np.meshgrid(off, off, indexing="ij")
    ox = ox.ravel()[None, :]
    oy = oy.ravel()[None, :]
    out = np.zeros(Nx * Ny, dtype=np.float64)
    chunk = max(1, int(2e6 // ox.size))     # bound the scratch arrays to ~2M entries
    for s in range(0, n_s, chunk):
        ix = np.floor(cx[s:s + chunk]).astype(np.int64)[:, None] + ox
        iy = np.floor(cy[s:s + chunk]).astype(np.int64)[:, None] + oy
        d2 = (ix - cx[s:s + chunk, None]) ** 2 + (iy - cy[s:s + chunk, None]) ** 2
        t = rr[s:s + chunk, None] ** 2 - d2
        ok = (t > 0) & (ix >= 0) & (ix < Nx) & (iy >= 0) & (iy < Ny)
        np.add.at(out, (ix[ok] * Ny + iy[ok]), 2.0 * np.sqrt(t[ok]) * pix_m)
    return out.reshape(Nx, Ny).astype(dtype)


def cylinder_sample(Nx, Ny, pix_m, radius_frac=0.23, dtype=np.float32):
    """Cylinder lying along axis 0 (x), radius = radius_frac*Ny pixels: T = 2*sqrt(R^2 - y^2)*pix."""
    Rpx = radius_frac * Ny
    y = np.arange(Ny, dtype=np.float64) - (Ny - 1) / 2.0
    t = np.clip(Rpx * Rpx - y * y, 0.0, None)
    row = 2.0 * np.sqrt(t) * pix_m
    return np.broadcast_to(row[None, :], (Nx, Ny)).astype(dtype).copy()


def slab(Nx, Ny, thickness_m, dtype=np.float32):
    """Uniform slab (Sample.py:239-243, get_my_thickness)."""
    return np.full((Nx, Ny), thickness_m, dtype=dtype)


def bench_geometry(N, pointNum=0, ov=2, det_pix_um=6.0, dSM=140.0, dMO=1.6, dOD=3.6):
    """The synthetic workload of SURVEY.md section 8d on an N x N study grid."""
    M = (dSM + dMO + dOD) / (dSM + dMO)
    pix_um = det_pix_um / ov / M
    mem_pix_um = pix_um * dSM / (dSM + dMO)
    membrane = np.stack([
        sphere_membrane(N, N, mem_pix_um * 1e-6, pointNum),
        slab(N, N, 6e-3),
    ])
    sample = cylinder_sample(N, N, pix_um * 1e-6)[None]
    return dict(M=M, pix_um=pix_um, membrane=membrane, sample=sample,
                membrane_materials=["CuSn", "PMMA"], sample_materials=["Nylon"],
                dSM=dSM, dMO=dMO, dOD=dOD, energy_keV=52.0)


# ---------------------------------------------------------------------------------------------------------------------
# Stand-in for the reference's sphere list Samples/Membranes/CuSn.txt (.MISSING_LARGE_BLOBS): same JSON layout, a list of
# [y, x, r] in "file units" centred on the origin, for a membrane of 9740 x 8102 units with mean sphere radius 12.8
# (getMembraneFromFile.py:84-87).  Seeded, so every run and every rank sees the same file.
SPHERE_FILE_SIZE_X = 8102
SPHERE_FILE_SIZE_Y = 9740
SPHERE_FILE_MEAN_RADIUS = 12.8


def sphere_list(seed=20211011, coverage=0.45, n_max=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(coverage * SPHERE_FILE_SIZE_X * SPHERE_FILE_SIZE_Y / (np.pi * SPHERE_FILE_MEAN_RADIUS ** 2))
    if n_max is not None:
        n = min(n, int(n_max))
    y = rng.uniform(-SPHERE_FILE_SIZE_Y / 2, SPHERE_FILE_SIZE_Y / 2, n)
    x = rng.uniform(-SPHERE_FILE_SIZE_X / 2, SPHERE_FILE_SIZE_X / 2, n)
    r = np.clip(rng.normal(SPHERE_FILE_MEAN_RADIUS, 3.0, n), 4.0, 24.0)
    return np.stack([y, x, r], axis=1)


def bench_experiment(N, sim="Fresnel", noise=True, seed=0, ov=2, psf=1.2, source_size_um=10.0, spectrum=None):
    """The synthetic experiment of SURVEY.md section 8d on an N x N study grid, built from objects (no XML): 52 keV
    (or `spectrum` = [(E_keV, weight)]), dSM/dMO/dOD = 140/1.6/3.6 m, detector N/ov pixels of 6 um, PSF 1.2 px, 10 um source,
    CuSn-sphere membrane on a 6 mm PMMA support, Nylon cylinder.  Returns (experiment, place) where place(pointNum) renders
    that position's membrane on the GPU (seeded offsets, getMembraneSegmentedFromFile: what main.py:64-65 does per position).
    bench.py's 64-position batch (BASELINE.json config 4) and tools/time_positions.py run on it."""
    import types

    import torch

    from .Detector import Detector
    from .Experiment import Experiment
    from .Sample import AnalyticalSample
    from .Samples.getMembraneFromFile import getMembraneSegmentedFromFile
    from .Source import Source

    geo = bench_geometry(N, pointNum=0, ov=ov)
    spectrum = [(52.0, 1.0)] if spectrum is None else list(spectrum)
    energies = [e for e, _ in spectrum]
    src = Source()
    src.myName = "synthetic"
    src.mySpectrum = list(spectrum)
    src.source_dict.update(mySize=source_size_um, myEnergySampling=1.0 if len(energies) < 2 else energies[1] - energies[0],
                           myType="Monochromatic" if len(energies) == 1 else "Polychromatic")
    det = Detector({})
    det.myName = "synthetic"
    det.det_param.update(myDimensions=np.array((N // ov, N // ov)), myPixelSize=6.0, myPSF=psf, myBinsThersholds=[])

    def scaled(name, e):
        d0, b0 = DELTA_BETA_52KEV[name]
        return d0 * (52.0 / e) ** 2, b0 * (52.0 / e) ** 3

    def sample(geom, name, mtype, mats):
        s = AnalyticalSample()
        s.myName, s.myType, s.myMaterials = name, mtype, list(mats)
        s.myGeometry = geom
        s.delta = [[(e, scaled(m, e)[0]) for e in energies] for m in mats]
        s.beta = [[(e, scaled(m, e)[1]) for e in energies] for m in mats]
        return s

    exp_dict = {"experimentName": "synthetic", "overSampling": ov, "nbExpPoints": 64,
                "simulation_type": "RayT" if sim in ("RT", "RayT") else "Fresnel", "studyPixelSize": geo["pix_um"],
                "studyDimensions": [N, N], "inVacuum": True, "meanShotCount": 30000.0, "meanEnergy": 0,
                "distSourceToMembrane": 140.0, "distMembraneToObject": 1.6, "distObjectToDetector": 3.6,
                "magnification": geo["M"], "noise": noise, "seed": seed, "deferMeanEnergy": True, "deferStatus": True}
    membrane = sample(geo["membrane"], "membrane", "membrane", geo["membrane_materials"])
    exp = Experiment.from_objects(exp_dict, src, det, membrane,
                                  sample(geo["sample"], "sample", "sample_of_interest", ["Nylon"]))
    smp = types.SimpleNamespace(myMeanSphereRadius=15.0, myNbOfLayers=2)
    mpix = geo["pix_um"] * 140.0 / 141.6

    def place(pointNum):
        geom, _ = getMembraneSegmentedFromFile(smp, N, N, mpix, pointNum, 6000.0, stacked=True)
        exp.myMembrane.myGeometry = geom[2]

    return exp, place
