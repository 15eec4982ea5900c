"""Build a paresis_amd Experiment from the injected configuration of tests/golden/experiment.npz (no XML)."""
import numpy as np


def build_experiment(cfg, sim, noise=False, sample_materials=("Nylon",), sample_name="sample"):
    from paresis_amd.Detector import Detector
    from paresis_amd.Experiment import Experiment
    from paresis_amd.Sample import AnalyticalSample
    from paresis_amd.Source import Source

    energies = [e for e, _ in cfg["spectrum"]]
    src = Source()
    src.myName = "injected"
    src.mySpectrum = list(cfg["spectrum"])
    src.source_dict.update(mySize=cfg["source_size_um"], myEnergySampling=cfg["energy_sampling"],
                           myType="Monochromatic" if len(energies) == 1 else "Polychromatic")
    det = Detector({})
    det.myName = "injected"
    det.det_param.update(myDimensions=np.array(cfg["det_dims"]), myPixelSize=cfg["det_pix_um"], myPSF=cfg["psf"],
                         myBinsThersholds=list(cfg["bins"]))

    if cfg.get("scintillator") is not None:
        det.det_param.update(myScintillatorMaterial="injected", myScintillatorThickness=cfg["scintillator"][0])
        det.beta = list(cfg["scintillator"][1])
        det.getSpectralEfficiency()

    def sample(obj, name, mtype, mats):
        if obj is None:
            return None
        s = AnalyticalSample()
        s.myName, s.myType, s.myMaterials = name, mtype, list(mats)
        s.myGeometry = obj.geometry
        s.delta = [[(e, obj.delta[m][i]) for i, e in enumerate(energies)] for m in range(len(mats))]
        s.beta = [[(e, obj.beta[m][i]) for i, e in enumerate(energies)] for m in range(len(mats))]
        return s

    exp_dict = {"experimentName": "injected", "overSampling": cfg["ov"], "nbExpPoints": 2,
                "simulation_type": "RayT" if sim == "RT" else "Fresnel", "studyPixelSize": cfg["pix_um"],
                "studyDimensions": list(cfg["N"]), "inVacuum": cfg["inVacuum"], "meanShotCount": cfg["meanShotCount"],
                "meanEnergy": 0, "distSourceToMembrane": cfg["dSM"], "distMembraneToObject": cfg["dMO"],
                "distObjectToDetector": cfg["dOD"], "magnification": cfg["M"], "noise": noise}
    return Experiment.from_objects(exp_dict, src, det,
                                   sample(cfg["membrane"], "membrane", "membrane", ["CuSn", "PMMA"]),
                                   sample(cfg["sample"], sample_name, "sample_of_interest", list(sample_materials)),
                                   air=None if cfg["inVacuum"] else sample(cfg["air"], "air_volume", "volume", ["Air"]),
                                   plate=sample(cfg["plate"], "plate", "thin_film", ["C"]))


def cfg_from_experiment(exp, Obj):
    """Oracle configuration holding exactly what an XML-built paresis_amd Experiment holds (same XML -> both sides)."""
    import numpy as np
    ed = exp.exp_dict
    energies = [e for e, _ in exp.mySource.mySpectrum]

    def obj(s):
        if s is None:
            return None
        g = s.myGeometry
        geom = np.asarray(g.cpu().numpy() if hasattr(g, "cpu") else g, dtype=np.float64)
        return Obj(geom, [[v for _, v in l] for l in s.delta], [[v for _, v in l] for l in s.beta])

    return dict(dSM=ed["distSourceToMembrane"], dMO=ed["distMembraneToObject"], dOD=ed["distObjectToDetector"],
                meanShotCount=ed["meanShotCount"], ov=ed["overSampling"], pix_um=ed["studyPixelSize"],
                M=ed["magnification"], inVacuum=ed["inVacuum"], N=tuple(ed["studyDimensions"]),
                spectrum=list(exp.mySource.mySpectrum), source_size_um=exp.mySource.source_dict["mySize"],
                energy_sampling=exp.mySource.source_dict["myEnergySampling"],
                det_dims=tuple(int(v) for v in exp.myDetector.det_param["myDimensions"]),
                det_pix_um=exp.myDetector.det_param["myPixelSize"], psf=exp.myDetector.det_param["myPSF"],
                bins=list(exp.myDetector.det_param["myBinsThersholds"]),
                membrane=obj(exp.myMembrane), sample=obj(exp.mySampleofInterest),
                air=None if ed["inVacuum"] else obj(exp.myAirVolume), plate=obj(exp.myPlate))
