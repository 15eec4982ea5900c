"""GPU parity of the full image-formation chains (Experiment.computeSampleAndReferenceImages_{RT,Fresnel}) against the
golden outputs of the reference on identical injected configurations, and of the reference-named module functions."""
import numpy as np
import pytest
import torch

from oracle import paresis_oracle as orc
from tests._build import build_experiment
from tests._golden import experiment_cfg, load, relmax

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.mark.parametrize("tag", ["mono", "poly"])
def test_chain_rt(tag):
    g = load("experiment.npz")
    cfg = experiment_cfg(g, tag + "/RT", orc.Obj)
    exp = build_experiment(cfg, "RT")
    for point in (0, 1):
        exp.myMembrane.myGeometry = g["%s/RT/p%d/membrane" % (tag, point)]
        exp.exp_dict["meanEnergy"] = 0
        S, R, Pg, W, Dx, Dy, DF = exp.computeSampleAndReferenceImages_RT(point)
        t = "%s/RT/p%d/" % (tag, point)
        for nm, a in (("Sample", S), ("Reference", R), ("Propag", Pg), ("White", W)):
            err = relmax(a.cpu().numpy(), g[t + nm])
            assert err < TOL, (tag, point, nm, err)
        if point == 0:
            assert relmax(Dx.cpu().numpy(), g[t + "Dx"]) < 1e-6 and relmax(Dy.cpu().numpy(), g[t + "Dy"]) < 1e-6
            assert float(DF.abs().max()) == 0.0
        assert abs(exp.exp_dict["meanEnergy"] - float(g[t + "meanEnergy"])) < 1e-4
    assert list(g[tag + "/RT/bins_after"]) == exp.myDetector.det_param["myBinsThersholds"]


def test_chain_rt_with_measured_halo():
    """exp_dict['refractionHalo'] = 'auto': the gather halo is picked by timing the experiment's own longest hop with 4, 6
    and 8 pixels on the first call; the images are those of the golden run whatever wins."""
    from paresis_amd import ops
    g = load("experiment.npz")
    cfg = experiment_cfg(g, "mono/RT", orc.Obj)
    exp = build_experiment(cfg, "RT")
    exp.exp_dict["refractionHalo"] = "auto"
    exp.exp_dict["reproducible"] = False          # a timing may only decide where the last bit of an image is allowed to move
    try:
        exp.myMembrane.myGeometry = g["mono/RT/p0/membrane"]
        exp.exp_dict["meanEnergy"] = 0
        S, R, Pg, W, Dx, Dy, DF = exp.computeSampleAndReferenceImages_RT(0)
        assert exp._halo in (4, 6, 8) and sorted(exp._halo_times) == [4, 6, 8]      # oversampling < 4: the three narrow halos
        for nm, a in (("Sample", S), ("Reference", R), ("Propag", Pg), ("White", W)):
            assert relmax(a.cpu().numpy(), g["mono/RT/p0/" + nm]) < TOL, nm
        # reproducible (the default): 'auto' is a RULE of how far the longest hop's rays travel in study pixels, z / (h M) -- the
        # same on every rank and in every run
        for pix, want in ((2.9, 4), (1.46, 8), (0.7, 12)):
            exp2 = build_experiment(cfg, "RT")
            exp2.exp_dict["refractionHalo"] = "auto"
            exp2.exp_dict.update(studyPixelSize=pix, distMembraneToObject=1.6, distObjectToDetector=3.6, magnification=1.025)
            exp2._set_halo(None, None, None)
            assert exp2._halo == want and not hasattr(exp2, "_halo_times"), (pix, exp2._halo)
    finally:
        ops.set_refract_halo(4)


@pytest.mark.parametrize("engine", [1, 0])
@pytest.mark.parametrize("tag", ["mono", "poly"])
def test_chain_fresnel(tag, engine):
    g = load("experiment.npz")
    cfg = experiment_cfg(g, tag + "/Fresnel", orc.Obj)
    exp = build_experiment(cfg, "Fresnel")
    exp.exp_dict["fresnelEngine"] = engine
    for point in (0, 1):
        exp.myMembrane.myGeometry = g["%s/Fresnel/p%d/membrane" % (tag, point)]
        exp.exp_dict["meanEnergy"] = 0
        S, R, Pg, W = exp.computeSampleAndReferenceImages(point)
        t = "%s/Fresnel/p%d/" % (tag, point)
        for nm, a in (("Sample", S), ("Reference", R), ("Propag", Pg), ("White", W)):
            err = relmax(a.cpu().numpy(), g[t + nm])
            assert err < TOL, (tag, point, nm, err)
        assert abs(exp.exp_dict["meanEnergy"] - float(g[t + "meanEnergy"])) < 1e-4


def test_energy_batched_chain_matches_energy_loop():
    """VERDICT r1 missing 4: on small grids the energies of a detector bin go through the Fresnel chain together
    (psx_fresnel_propagate_sources + psx_accumulate_many_f32, 7 launches per bin instead of per energy).  Same images as the
    per-energy loop (float rounding of one fused multiply-add apart), same mean energy, same field left behind."""
    g = load("experiment.npz")
    cfg = experiment_cfg(g, "poly/Fresnel", orc.Obj)
    outs = {}
    for batched in (True, False):
        exp = build_experiment(cfg, "Fresnel")
        exp.exp_dict["batchEnergies"] = batched
        for point in (0, 1):
            exp.myMembrane.myGeometry = g["poly/Fresnel/p%d/membrane" % point]
            exp.exp_dict["meanEnergy"] = 0
            out = exp.computeSampleAndReferenceImages(point)
            outs[batched, point] = [a.clone() for a in out] + [exp.waveSampleBeforeSample.clone(), exp.exp_dict["meanEnergy"]]
    for point in (0, 1):
        b, l = outs[True, point], outs[False, point]
        for k in range(4 if point == 0 else 2):
            assert relmax(b[k].cpu().numpy(), l[k].cpu().numpy()) < 1e-6, (point, k)
        assert torch.equal(b[4], l[4])
        assert abs(b[5] - l[5]) < 1e-9 * abs(l[5])
    t = "poly/Fresnel/p0/"
    assert relmax(outs[True, 0][0].cpu().numpy(), g[t + "Sample"]) < TOL


def test_energy_batched_rt_chain_matches_energy_loop():
    """The ray-tracing chain with the energies of a bin taken together (psx_refract_batch_f32) against the per-energy loop:
    same images (far rays are summed by float atomics in both), same displacement maps, same mean energy."""
    g = load("experiment.npz")
    cfg = experiment_cfg(g, "poly/RT", orc.Obj)
    outs = {}
    for batched in (True, False):
        exp = build_experiment(cfg, "RT")
        exp.exp_dict["batchEnergies"] = batched
        for point in (0, 1):
            exp.myMembrane.myGeometry = g["poly/RT/p%d/membrane" % point]
            exp.exp_dict["meanEnergy"] = 0
            out = exp.computeSampleAndReferenceImages(point)
            outs[batched, point] = [a.clone() for a in out[:6]] + [exp.exp_dict["meanEnergy"]]
    for point in (0, 1):
        b, l = outs[True, point], outs[False, point]
        for k in range(4 if point == 0 else 2):
            assert relmax(b[k].cpu().numpy(), l[k].cpu().numpy()) < 2e-6, (point, k)
        assert abs(b[6] - l[6]) < 1e-9 * abs(l[6])
    assert torch.equal(outs[True, 0][4], outs[False, 0][4]) and torch.equal(outs[True, 0][5], outs[False, 0][5])
    t = "poly/RT/p0/"
    assert relmax(outs[True, 0][0].cpu().numpy(), g[t + "Sample"]) < TOL


def test_reference_named_functions():
    """The module-level functions keep the reference's names, argument order and return arity."""
    from paresis_amd import refractionFileNumba as RF1
    from paresis_amd import refractionFileNumba2 as RF2
    from paresis_amd.Detector import Detector, create_gaussian_shape, resize
    from paresis_amd.getk import getk
    g = load("refraction.npz")
    z, E, M, pix = g["1/params"]
    for mod, ver in ((RF2, "v2"), (RF1, "v1")):
        I = g["1/I"].astype(np.float32)            # numpy in: uploaded, and zeroed in place like the reference
        out, Dx, Dy = mod.fastRefraction(I, g["1/phi"], z, E, M, pix)
        assert relmax(out.cpu().numpy(), g["1/%s/out" % ver]) < TOL
        assert Dx.shape == g["1/%s/Dx" % ver].shape
    I2 = RF2.fastloopNumba(20, 17, g["loop/I"], np.zeros((20, 17)), g["loop/Dy"], g["loop/Dx"], None, None)
    assert relmax(I2.cpu().numpy(), g["loop/out"]) < TOL
    s = load("scalars.npz")
    assert getk(25000.0) == s["getk/k"][0]
    for i, sig in enumerate(s["gauss/sigma"]):
        assert relmax(create_gaussian_shape(sig), s["gauss/%d/det" % i]) < 1e-12
        assert relmax(RF2.gaussian_shape(sig), s["gauss/%d/rf2" % i]) < 1e-12
    assert relmax(resize(s["resize/1/in"], 6, 4).cpu().numpy(), s["resize/1/out"]) < 1e-6
    d = load("detector.npz")
    d0, d1, ov, fwhm, psf = d["4/params"]
    det = Detector({})
    det.det_param.update(myDimensions=np.array([int(d0), int(d1)]), myPSF=psf)
    out = det.detection(d["4/in"], fwhm, {"overSampling": int(ov), "noise": False})
    assert relmax(out.cpu().numpy(), d["4/out"]) < TOL
    noisy = det.detection(d["4/in"] * 50, fwhm, {"overSampling": int(ov), "seed": 3})
    assert torch.all(noisy == torch.floor(noisy))                       # Poisson counts (DET:113-115)


def test_sample_methods_golden():
    from paresis_amd.Sample import AnalyticalSample
    g = load("transmission.npz")
    s = AnalyticalSample()
    s.myMaterials = ["CuSn", "PMMA"]
    s.myType = "membrane"
    s.myGeometry = g["T"]
    E = list(g["energies"])
    s.delta = [[(e, g["delta"][m][i]) for i, e in enumerate(E)] for m in range(2)]
    s.beta = [[(e, g["beta"][m][i]) for i, e in enumerate(E)] for m in range(2)]
    for ie, e in enumerate(E):
        w = s.setWave(g["wave_in"], e)
        assert relmax(w.cpu().numpy(), g["setWave/%d" % ie]) < TOL
        I, phi, df = s.setWaveRT(g["I_in"], e, g["phi_in"])
        assert relmax(I.cpu().numpy(), g["setWaveRT/%d/I" % ie]) < TOL
        assert relmax(phi.cpu().numpy(), g["setWaveRT/%d/phi" % ie]) < 1e-12 and df == 0
        I, phi, df = s.setWaveRT(g["I_in"], e)
        assert relmax(phi.cpu().numpy(), g["setWaveRT0/%d/phi" % ie]) < 1e-12
    s.myGeometry = g["T"][0]
    with pytest.raises(Exception, match="wrong nb of dim"):
        s.setWave(g["wave_in"], E[0])


def test_experiment_wave_propagation_and_refraction_methods():
    g = load("fresnel.npz")
    cfgg = load("experiment.npz")
    cfg = experiment_cfg(cfgg, "mono/Fresnel", orc.Obj)
    exp = build_experiment(cfg, "Fresnel")
    k = 11   # 96x80 grid: same shape as the experiment's study grid
    z, E, M, pix = g["%d/params" % k]
    assert tuple(g["%d/wave" % k].shape) == tuple(exp.exp_dict["studyDimensions"])
    exp.exp_dict["studyPixelSize"] = pix
    out = exp.wavePropagation(g["%d/wave" % k], z, E, M)
    assert relmax(out.cpu().numpy(), g["%d/out" % k]) < TOL
    w = g["%d/wave" % k]
    assert exp.wavePropagation(w, 0, E, M) is w                          # EXP:233-234
    r = load("refraction.npz")
    z, E, M, pix = r["1/params"]
    exp.exp_dict["studyPixelSize"] = pix
    out, Dx, Dy = exp.refraction(r["1/I"].copy(), r["1/phi"], z, E, M)
    assert relmax(out.cpu().numpy(), r["1/v2/out"]) < TOL


def test_darkfield_refraction_and_sample_model():
    """Dark-field branch: setWaveRT's scattering model, fastRefractionDF, and the RT chain."""
    from paresis_amd import refractionFileNumba2 as RF2
    from paresis_amd.Sample import AnalyticalSample
    g = load("darkfield.npz")
    for k in range(int(g["rf/n"])):
        z, E, M, pix = g["rf/%d/params" % k]
        I = g["rf/I"].astype(np.float32)
        out, Dx, Dy = RF2.fastRefractionDF(I, g["rf/phi"], z, E, M, pix, g["rf/%d/df" % k])
        assert tuple(Dx.shape) == g["rf/%d/Dx" % k].shape                 # padded by ceil(6*max DF)
        assert relmax(out.cpu().numpy(), g["rf/%d/out" % k]) < TOL, k
        assert relmax(Dx.cpu().numpy(), g["rf/%d/Dx" % k]) < 1e-6
    s = AnalyticalSample()
    s.myName, s.myType, s.myMaterials, s.myGeometry = "lungs", "sample_of_interest", ["Lung", "PMMA"], g["lung/geometry"]
    s.delta, s.beta = [[(52.0, 3.1e-7)], [(52.0, 9.87e-8)]], [[(52.0, 1.6e-10)], [(52.0, 4.5e-11)]]
    I1, phi1, df1 = s.setWaveRT(g["rf/I"], 52.0, g["rf/phi"])
    assert relmax(I1.cpu().numpy(), g["lung/I"]) < TOL
    assert relmax(phi1.cpu().numpy(), g["lung/phi"]) < 1e-7       # thickness maps are float32 on the device
    assert relmax(df1.cpu().numpy(), g["lung/df"]) < 1e-6
    s2 = AnalyticalSample()
    s2.myName, s2.myType, s2.myMaterials, s2.myGeometry = "cylinder_beeds", "sample_of_interest", ["PMMA"], g["lung/geometry"][:1]
    s2.delta, s2.beta = [[(52.0, 9.87e-8)]], [[(52.0, 4.5e-11)]]
    I2, phi2, df2 = s2.setWaveRT(g["rf/I"], 52.0, g["rf/phi"])
    assert relmax(I2.cpu().numpy(), g["beeds/I"]) < TOL and relmax(df2.cpu().numpy(), g["beeds/df"]) < 1e-6


def test_darkfield_chain():
    g = load("darkfield.npz")
    cfg = experiment_cfg(g, "chain", orc.Obj)
    exp = build_experiment(cfg, "RT", sample_materials=("Lung",), sample_name="lungs")
    DF0 = None
    for point in (0, 1):
        exp.myMembrane.myGeometry = g["chain/p%d/membrane" % point]
        exp.exp_dict["meanEnergy"] = 0
        S, R, Pg, W, Dx, Dy, DF = exp.computeSampleAndReferenceImages_RT(point)
        t = "chain/p%d/" % point
        for nm, a in (("Sample", S), ("Reference", R), ("Propag", Pg), ("White", W)):
            err = relmax(a.cpu().numpy(), g[t + nm])
            assert err < TOL, (point, nm, err)
        if point == 0:
            assert tuple(Dx.shape) == g[t + "Dx"].shape
            assert relmax(Dx.cpu().numpy(), g[t + "Dx"]) < 1e-6
            assert relmax(DF.cpu().numpy(), g[t + "DF"]) < 1e-6
            DF0 = DF
        else:
            # EXP:444 allocates a new map per call: position 1 returns zeros AND leaves position 0's map as it was
            # (main.run keeps results[0] until the gather at the end of the run)
            assert float(DF.abs().max()) == 0.0
            assert DF.data_ptr() != DF0.data_ptr()
            assert relmax(DF0.cpu().numpy(), g["chain/p0/DF"]) < 1e-6


def test_darkfield_chain_is_bitwise_repeatable_and_restores_the_callers_mode():
    """the reproducibility tests had no scattering sample.  The halves of the dark-field split (RF2:147-150) are
    zero over half the image each -- tiles that stage nothing, which round 4's replay could only serve with float atomics.  The
    chain of a scattering sample, run twice by the class alone (no main.run): same bits; the calling thread's replay mode is
    what it was before the call, whichever it was; exp_dict['reproducible'] = False stays within float rounding."""
    from paresis_amd import ops
    g = load("darkfield.npz")
    cfg = experiment_cfg(g, "chain", orc.Obj)
    exp = build_experiment(cfg, "RT", sample_materials=("Lung",), sample_name="lungs")
    exp.myMembrane.myGeometry = g["chain/p0/membrane"]
    runs = []
    for mode in (False, True):
        ops.set_deterministic(mode)
        exp.exp_dict["meanEnergy"] = 0
        out = exp.computeSampleAndReferenceImages_RT(0)
        assert ops.get_deterministic() == mode
        runs.append([t.clone() for t in out[:4]])
    ops.set_deterministic(False)
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    exp.exp_dict["reproducible"] = False
    exp.exp_dict["meanEnergy"] = 0
    out = exp.computeSampleAndReferenceImages_RT(0)
    for a, b in zip(out[:4], runs[0]):
        assert float((a - b).abs().max() / b.abs().max()) < 2e-6


def test_replay_scopes_nest_and_keep_the_callers_scale():
    """ops.deterministic restores mode AND scale, so a nested scope (Experiment.refraction inside the chain's
    own scope, the zero-width dark-field branch) neither loses the chain's fixed unit nor clobbers a user's own setting."""
    from paresis_amd import ops
    assert ops.get_deterministic_scale() == 0.0
    ops.set_deterministic_scale(123.0)
    try:
        with ops.deterministic(True, scale=7500.0):
            assert ops.get_deterministic() and ops.get_deterministic_scale() == 7500.0
            with ops.deterministic(True, scale=None):                # Experiment.refraction's scope: the caller's unit stays
                assert ops.get_deterministic_scale() == 7500.0
            assert ops.get_deterministic_scale() == 7500.0
            with ops.deterministic(False):
                assert not ops.get_deterministic() and ops.get_deterministic_scale() == 0.0
            assert ops.get_deterministic() and ops.get_deterministic_scale() == 7500.0
        assert ops.get_deterministic_scale() == 123.0 and not ops.get_deterministic()
    finally:
        ops.set_deterministic_scale(0.0)


@pytest.mark.parametrize("sim", ["RT", "Fresnel"])
def test_polychromatic_frontend_chain(sim):
    """the energy loop as a reduced axis -- 5 energies from the re-binned tabulated spectrum, table-walk
    delta/beta, scintillator efficiency, air, plate, two detector bins -- against the reference's own run."""
    from paresis_amd import materials
    from paresis_amd.Source import Source
    g = load("frontend_chain.npz")
    f = load("frontend.npz")
    cfg = experiment_cfg(g, "chain/" + sim, orc.Obj)
    exp = build_experiment(cfg, sim, sample_materials=("SynthNylon",))
    # the same front-end on the package side: spectrum from the injected sheet rows, delta/beta from registered tables
    src = Source()
    src.spectrumFromXls = True
    src.source_dict.update({"myType": "Polychromatic", "myEnergySampling": float(g["xls/sampling"]), "energyUnit": "keV",
                            "xlsRows": list(zip(g["xls/E"], g["xls/fluence"])), "mySize": cfg["source_size_um"]})
    src.setMySpectrum()
    assert src.mySpectrum == cfg["spectrum"]
    exp.mySource.mySpectrum = src.mySpectrum
    for m, n in enumerate(f["tab/names"]):
        materials.register_table(str(n), f["tab/%d/E_eV" % m], f["tab/%d/delta" % m], f["tab/%d/beta" % m])
    smp = exp.mySampleofInterest
    smp.delta, smp.beta = [], []
    smp.getDeltaBeta(src.mySpectrum)
    exp.myDetector.det_param["myScintillatorMaterial"] = "SynthGadox"
    exp.myDetector.getBeta(src.mySpectrum)
    exp.myDetector.getSpectralEfficiency()
    assert np.array_equal(np.array(exp.myDetector.beta), g["chain/%s/scint_beta" % sim])
    for point in (0, 1):
        exp.myMembrane.myGeometry = g["chain/%s/p%d/membrane" % (sim, point)]
        exp.exp_dict["meanEnergy"] = 0
        out = exp.computeSampleAndReferenceImages(point)
        t = "chain/%s/p%d/" % (sim, point)
        for nm, a in zip(("Sample", "Reference", "Propag", "White"), out[:4]):
            err = relmax(a.cpu().numpy(), g[t + nm])
            assert err < TOL, (sim, point, nm, err)
