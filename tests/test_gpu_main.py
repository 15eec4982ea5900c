"""XML-driven entry (paresis_amd/main.py, mirror of CodePython/main.py) on the GPU, and parity of the XML-built chains
with the oracle fed from the SAME XML-derived configuration."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import paresis_oracle as orc
from tests._build import cfg_from_experiment
from tests._golden import relmax

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sim,name,ov", [("RayT", "Fil_Nylon_ID17", 2), ("Fresnel", "Fil_Nylon_ID17", 2),
                                         ("Fresnel", "Sphere_PMMA_plate", 1), ("RayT", "Sphere_PMMA_plate", 2),
                                         ("RayT", "Config1_512", 2), ("Fresnel", "Config1_512", 2)])
def test_xml_experiment_matches_oracle(sim, name, ov):
    """The XML-built chains against the oracle.  The oracle is fed the configuration the package derived from the XML
    (cfg_from_experiment), so an XML plumbing error that changes both sides alike would be invisible here: the derived
    scalars are therefore held, first, to the reference's own arithmetic (tests/golden/xml_scalars.npz, where the
    experiment is listed; Config1_512 = BASELINE.json config 1 at its stated 512^2 = detector 256 x oversampling 2)."""
    from paresis_amd.Experiment import Experiment
    from tests._golden import load
    ed = {"experimentName": name, "filepath": "/tmp/", "overSampling": ov, "nbExpPoints": 2, "simulation_type": sim,
          "noise": False}
    exp = Experiment(ed)
    g = load("xml_scalars.npz")
    if name in [str(n) for n in g["names"]] and int(g[name + "/overSampling"]) == ov:
        assert ed["magnification"] == float(g[name + "/magnification"])
        assert [int(v) for v in ed["studyDimensions"]] == [int(v) for v in g[name + "/studyDimensions"]]
        assert ed["studyPixelSize"] == float(g[name + "/studyPixelSize"])
        assert exp.myMembrane.membranePixelSize == float(g[name + "/membranePixelSize"])
    for point in (0, 1):
        exp.myMembrane.myGeometry = []
        exp.myMembrane.getMyGeometry(ed["studyDimensions"], exp.myMembrane.membranePixelSize, ov, point, 2)   # main.py:64-65
        cfg = cfg_from_experiment(exp, orc.Obj)
        ed["meanEnergy"] = 0
        out = exp.computeSampleAndReferenceImages(point)
        if sim == "RayT":
            ref = orc.compute_rt(cfg, point)
            refs, mE = ref[:4], ref[6]
        else:
            ref = orc.compute_fresnel(cfg, point)
            refs, mE = ref[:4], ref[4]
        for nm, a, r in zip(("Sample", "Reference", "Propag", "White"), out[:4], refs):
            err = relmax(a.cpu().numpy(), r)
            assert err < 1e-5, (sim, name, point, nm, err)
        assert abs(ed["meanEnergy"] - mE) < 1e-4


def test_main_writes_images(tmp_path):
    from paresis_amd import main
    from paresis_amd.InputOutput.pagailleIO import openImage
    for sim, fmt in (("RayT", ".tif"), ("Fresnel", ".edf")):
        ed = {"experimentName": "Fil_Nylon_ID17", "filepath": str(tmp_path) + "/" + sim + "/", "overSampling": 2,
              "nbExpPoints": 2, "simulation_type": sim, "noise": True, "seed": 5}
        os.makedirs(ed["filepath"])
        res = main.run(ed, save=True, saving_format=fmt)
        assert sorted(res) == [0, 1] and res[0][0].shape == (1, 200, 200)
        files = glob.glob(ed["filepath"] + "*/sample/*" + fmt)
        assert len(files) == 2
        img = openImage(files[0])
        assert img.shape == (200, 200) and np.all(img == np.floor(img)) and img.mean() > 1000    # Poisson counts
        assert glob.glob(ed["filepath"] + "*/ref/*" + fmt) and glob.glob(ed["filepath"] + "*/propag/*" + fmt)
        assert glob.glob(ed["filepath"] + "*.txt")                                                  # saveAllParameters
        assert len(glob.glob(ed["filepath"] + "*/membraneThickness/*" + fmt)) == 2                  # main.py:98
        assert bool(glob.glob(ed["filepath"] + "*/DF" + fmt)) == (sim == "RayT")                    # main.py:100-101
        txt = open(glob.glob(ed["filepath"] + "*.txt")[0]).read()
        assert "delta/beta source" in txt and "SYNTHETIC" in txt                                    # provenance in the dump


def test_membrane_synthesis_golden_and_seeding():
    """getMembraneSegmentedFromFile on the GPU against the reference's own output (tests/golden/membrane.npz)."""
    import types
    from paresis_amd import synth
    from paresis_amd.Samples.getMembraneFromFile import getMembraneSegmentedFromFile
    from tests._golden import load
    g = load("membrane.npz")
    for tag in ("plain", "stitch"):
        dimX, dimY, pix, meanR, layers, support, nmax, seed = g[tag + "/params"]
        lst = synth.sphere_list(n_max=None if nmax < 0 else int(nmax))
        smp = types.SimpleNamespace(myMeanSphereRadius=meanR, myNbOfLayers=int(layers))
        geom, par = getMembraneSegmentedFromFile(smp, int(dimX), int(dimY), pix, 0, support, seed=int(seed), sphere_list=lst)
        assert relmax(geom[0].cpu().numpy(), g[tag + "/membrane"]) < 2e-7, tag
        assert relmax(geom[1].cpu().numpy(), g[tag + "/support"]) < 2e-7
        assert par['Number of layers'][0] == int(layers)
    # a position always gets the same membrane, different positions differ (seed(pointNum) = 1000 + pointNum)
    smp = types.SimpleNamespace(myMeanSphereRadius=15.0, myNbOfLayers=2)
    a = getMembraneSegmentedFromFile(smp, 400, 400, 1.45, 3, 6000.0)[0][0]
    b = getMembraneSegmentedFromFile(smp, 400, 400, 1.45, 3, 6000.0)[0][0]
    c = getMembraneSegmentedFromFile(smp, 400, 400, 1.45, 4, 6000.0)[0][0]
    assert torch.equal(a, b) and not torch.equal(a, c)
    # against the oracle at a larger size
    ref = orc.membrane_segmented(synth.sphere_list(), 400, 400, 1.45, 15.0, 2, 6000.0, synth.position_seed(3))
    assert relmax(a.cpu().numpy(), ref[0]) < 2e-7


def test_membrane_entry_points_agree():
    """psx_membrane_f32 (host sphere arrays, tile binning on the host) and the plan path (list resident on the GPU, cells
    found by the kernel) render the same layers, ragged grids and far-off offsets included."""
    import ctypes
    from ctypes import c_double, c_void_p
    from paresis_amd import synth
    from paresis_amd._lib import check, lib
    from paresis_amd.Samples import getMembraneFromFile as GM
    lst = synth.sphere_list()
    DP = ctypes.POINTER(c_double)
    for dimX, dimY, pix, meanR in ((333, 517, 1.45, 15.0), (96, 40, 2.9, 50.0), (1000, 1000, 0.8, 15.0)):
        margin, margin2, par, sizeX, sizeY = GM.stitched_list(lst, dimX, dimY, pix, meanR)
        x, y, r = (np.ascontiguousarray(v) for v in (par[:, 1] / pix, par[:, 0] / pix, par[:, 2] / pix))
        plan = c_void_p(None)
        check(lib().psx_membrane_plan_create(x.ctypes.data_as(DP), y.ctypes.data_as(DP), r.ctypes.data_as(DP), len(r),
                                             ctypes.byref(plan)), "plan")
        try:
            st = c_void_p(torch.cuda.current_stream().cuda_stream)
            for ox, oy in ((margin2, margin2), (int(sizeX / pix) - dimX - margin2 - 1, 7), (-500, 100000)):
                a = torch.full((dimX, dimY), 1.0, dtype=torch.float32, device="cuda")
                b = torch.full((dimX, dimY), 1.0, dtype=torch.float32, device="cuda")
                xf, yf = np.ascontiguousarray(x - ox), np.ascontiguousarray(y - oy)
                check(lib().psx_membrane_f32(xf.ctypes.data_as(DP), yf.ctypes.data_as(DP), r.ctypes.data_as(DP), len(r), dimX,
                                             dimY, margin, margin2, c_double(pix * 1e-6), 1, c_void_p(a.data_ptr()), st), "host")
                check(lib().psx_membrane_layer_f32(plan, ox, oy, dimX, dimY, margin, margin2, c_double(pix * 1e-6), 1,
                                                   c_void_p(b.data_ptr()), st), "plan")
                torch.cuda.synchronize()
                assert float((a - b).abs().max()) <= 1e-12 + 2e-7 * float(a.abs().max()), (dimX, dimY, ox, oy)
                if (ox, oy) == (margin2, margin2):
                    assert float(a.max()) > 1.0            # something was rendered
            # every layer in one launch (psx_membrane_layers_f32): float64 sum over the layers against the host-binned
            # layers added one by one, for 0, 1, 3 and 11 layers (more than one launch holds); the support map is filled by
            # the same launch
            offs = [(margin2 + 13 * k, margin2 + 7 * k * k) for k in range(11)]
            for nl in (0, 1, 3, 11):
                a = torch.zeros((dimX, dimY), dtype=torch.float32, device="cuda")
                b = torch.full((dimX, dimY), 7.0, dtype=torch.float32, device="cuda")
                sup = torch.zeros((dimX, dimY), dtype=torch.float32, device="cuda")
                ref64 = np.zeros((dimX, dimY))
                for ox, oy in offs[:nl]:
                    a.zero_()
                    xf, yf = np.ascontiguousarray(x - ox), np.ascontiguousarray(y - oy)
                    check(lib().psx_membrane_f32(xf.ctypes.data_as(DP), yf.ctypes.data_as(DP), r.ctypes.data_as(DP), len(r), dimX,
                                                 dimY, margin, margin2, c_double(pix * 1e-6), 0, c_void_p(a.data_ptr()), st), "host")
                    ref64 += a.cpu().numpy().astype(np.float64)
                oxs = (ctypes.c_int * max(1, nl))(*[o[0] for o in offs[:nl]])
                oys = (ctypes.c_int * max(1, nl))(*[o[1] for o in offs[:nl]])
                check(lib().psx_membrane_layers_f32(plan, nl, oxs, oys, dimX, dimY, margin, margin2, c_double(pix * 1e-6), 0,
                                                    c_void_p(b.data_ptr()), c_void_p(sup.data_ptr()), ctypes.c_float(0.006), st),
                      "layers")
                torch.cuda.synchronize()
                assert np.abs(b.cpu().numpy() - ref64).max() <= 1e-12 + 3e-7 * max(ref64.max(), 1e-30), (dimX, dimY, nl)
                assert bool((sup == np.float32(0.006)).all())
                if nl == 0:
                    assert float(b.abs().max()) == 0.0
        finally:
            lib().psx_membrane_plan_destroy(plan)


def test_fresh_experiment_starts_at_any_position():
    """a rank of a sharded run builds its own Experiment and may never compute position 0; the bin
    thresholds must be closed whichever position comes first, and the images must not depend on what was computed before."""
    from paresis_amd.Experiment import Experiment
    outs = {}
    for order in ((1,), (0, 1)):
        ed = {"experimentName": "Fil_Nylon_ID17", "filepath": "/tmp/", "overSampling": 2, "nbExpPoints": 2,
              "simulation_type": "Fresnel", "noise": True, "seed": 3}
        exp = Experiment(ed)
        for point in order:
            exp.myMembrane.myGeometry = []
            exp.myMembrane.getMyGeometry(ed["studyDimensions"], exp.myMembrane.membranePixelSize, 2, point, 2)
            out = exp.computeSampleAndReferenceImages(point)
        assert len(exp.myDetector.det_param["myBinsThersholds"]) == 1          # closed exactly once
        outs[order] = [t.clone() for t in out[:2]]
        assert out[0].shape[0] == 1 and float(out[0].mean()) > 100
    # same position, same seed -> the same noisy image whether or not position 0 ran first (noise keyed by content)
    for a, b in zip(outs[(1,)], outs[(0, 1)]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("sim", ["Fresnel", "RayT"])
def test_zero_stacks_of_later_positions_belong_to_the_caller(sim):
    """the reference returns FRESH zero arrays for Propag and White away from position 0 (EXP:363-375,
    488-498); a caller doing flat-field housekeeping in place (`White[White == 0] = 1`) must not change Propag, nor any other
    position's stacks.  The one-stack-per-experiment saving is an opt-in for callers that only read (main.run, the bench)."""
    from paresis_amd.Experiment import Experiment
    ed = {"experimentName": "Fil_Nylon_ID17", "filepath": "/tmp/", "overSampling": 2, "nbExpPoints": 3,
          "simulation_type": sim, "noise": False}
    exp = Experiment(ed)
    outs = []
    for point in (1, 2):
        exp.myMembrane.myGeometry = []
        exp.myMembrane.getMyGeometry(ed["studyDimensions"], exp.myMembrane.membranePixelSize, 2, point, 3)
        out = exp.computeSampleAndReferenceImages(point)
        S, R, Pg, W = out[:4]
        assert float(Pg.abs().max()) == 0.0 and float(W.abs().max()) == 0.0
        assert Pg.data_ptr() != W.data_ptr()
        W[W == 0] = 1.0                                   # the caller's housekeeping
        assert float(Pg.abs().max()) == 0.0
        outs.append((Pg, W))
    assert float(outs[0][1].min()) == 1.0 and float(outs[0][0].abs().max()) == 0.0     # position 1's stacks untouched by position 2
    # the opt-in: one shared stack, read-only by contract
    ed2 = dict(ed, sharedZeroStacks=True)
    exp2 = Experiment(ed2)
    exp2.myMembrane.myGeometry = []
    exp2.myMembrane.getMyGeometry(ed2["studyDimensions"], exp2.myMembrane.membranePixelSize, 2, 1, 3)
    o2 = exp2.computeSampleAndReferenceImages(1)
    assert o2[2].data_ptr() == o2[3].data_ptr() and float(o2[2].abs().max()) == 0.0


def test_noise_keys_differ_between_positions_bins_and_kinds():
    from paresis_amd import ops
    keys = {ops.poisson_key(5, p, b, k) for p in range(8) for b in range(3) for k in range(4)}
    assert len(keys) == 8 * 3 * 4
    lam = torch.full((256, 256), 50.0, dtype=torch.float32, device="cuda")
    a, b = lam.clone(), lam.clone()
    ops.poisson_multi([a, b], [ops.poisson_key(5, 0, 0, 0), ops.poisson_key(5, 1, 0, 0)])
    c = ops.poisson(lam, seed=ops.poisson_key(5, 0, 0, 0))
    assert torch.equal(a, c) and not torch.equal(a, b)                          # multi == single; positions differ
    corr = float(torch.corrcoef(torch.stack([a.flatten() - 50, b.flatten() - 50]))[0, 1])
    assert abs(corr) < 0.02


def _rank_main(rank, world, port, outdir, q):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      PARESIS_ALLOW_SYNTHETIC_MATERIALS="1")
    import torch as th
    from paresis_amd import main
    ed = {"experimentName": "Fil_Nylon_ID17", "filepath": outdir + "/", "overSampling": 2, "nbExpPoints": 5,
          "simulation_type": "Fresnel", "noise": True, "seed": 9}
    res = main.run(ed, save=True, saving_format=".tif", backend="gloo")       # both ranks on the ONE GPU of the box
    q.put({p: [t.numpy() for t in v[:2]] for p, v in res.items()} if rank == 0 else (res == {}))
    th.distributed.destroy_process_group()


def test_main_two_ranks_match_one(tmp_path):
    """The XML entry point with 2 ranks (gloo control plane, both on this GPU) against the 1-process run: identical images,
    shot noise included (noise keyed by position, not by call order)."""
    import socket
    import torch.multiprocessing as mp
    from paresis_amd import main
    ed = {"experimentName": "Fil_Nylon_ID17", "filepath": str(tmp_path) + "/one/", "overSampling": 2, "nbExpPoints": 5,
          "simulation_type": "Fresnel", "noise": True, "seed": 9}
    os.makedirs(ed["filepath"])
    one = main.run(ed, save=False)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    out = str(tmp_path) + "/two"
    os.makedirs(out)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    two = next(r for r in res if isinstance(r, dict))
    assert sorted(two) == sorted(one) == list(range(5))
    for p in range(5):
        for a, b in zip(one[p][:2], two[p]):
            assert np.array_equal(a.numpy(), b), p
    assert len(glob.glob(out + "/Fresnel_*/membraneThickness/*.tif")) == 5


def _rank_config4(rank, world, port, outdir, sim, q):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      PARESIS_ALLOW_SYNTHETIC_MATERIALS="1")
    import torch as th
    from paresis_amd import dist, main
    ed = {"experimentName": "Bench_4096", "filepath": outdir + "/", "overSampling": 2, "nbExpPoints": 8,
          "simulation_type": sim, "noise": True, "seed": 21}
    res = main.run(ed, save=False, backend="gloo")                             # both ranks on the ONE GPU of the box
    if rank == 0:
        import zlib
        q.put({"crc": {p: [zlib.crc32(t.numpy().tobytes()) for t in v[:2]] for p, v in res.items()},
               "packed": bool(dist.last_gather.get("packed")), "overlapped": bool(dist.last_gather.get("overlapped")),
               "wire_bytes": int(dist.last_gather.get("wire_bytes", 0))})
    else:
        q.put(res == {})
    th.distributed.destroy_process_group()


@pytest.mark.parametrize("sim", ["Fresnel", "RayT"])
def test_config4_workload_two_ranks_on_one_gpu(tmp_path, sim):
    """BASELINE config 4's workload inside the GPU suite: the XML entry point over 8 membrane positions of the
    4096^2 experiment (detector 2048^2, oversampling 2) -- membrane synthesis with seed(pointNum), chain, detection, shot noise
    -- sharded over 2 ranks on this GPU (gloo control plane, the gather round by round behind the computation, 16-bit packed
    counts on the wire: 3.3 % of this experiment's pixels are caustics above 65534 counts and ride in the escape table),
    against the 1-process run: every Sample / Reference stack bit for bit, shot noise included; position 0
    carries its Propag / White.  (64 positions over 8 GPUs is the driver's SCALE run; tests/test_dist_gloo.py rehearses its
    indexing at world 8 x 64 on the CPU.)"""
    import socket
    import zlib
    import torch.multiprocessing as mp
    from paresis_amd import main
    ed = {"experimentName": "Bench_4096", "filepath": str(tmp_path) + "/one/", "overSampling": 2, "nbExpPoints": 8,
          "simulation_type": sim, "noise": True, "seed": 21}
    os.makedirs(ed["filepath"])
    one = main.run(ed, save=False)
    assert sorted(one) == list(range(8)) and tuple(one[0][0].shape) == (1, 2048, 2048) and len(one[0]) >= 4
    assert float(one[0][2].abs().max()) > 0 and float(one[3][0].mean()) > 1000          # Propag at position 0; photon counts
    crc_one = {p: [zlib.crc32(t.numpy().tobytes()) for t in v[:2]] for p, v in one.items()}
    assert len({c[0] for c in crc_one.values()}) == 8                                     # eight different membranes / noise keys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    out = str(tmp_path) + "/two"
    os.makedirs(out)
    procs = [ctx.Process(target=_rank_config4, args=(r, 2, port, out, sim, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    two = next(r for r in res if isinstance(r, dict))
    assert two["crc"] == crc_one
    print("gather:", {k: v for k, v in two.items() if k != "crc"})
    assert two["packed"], two                               # 16-bit photon counts crossed, not float32 ...
    assert two["overlapped"], two                           # ... round by round behind the computation


def test_reproducible_ray_tracing_run_is_bitwise_repeatable(tmp_path):
    """exp_dict['reproducible']: the ray-tracing chain with the order-independent far-ray replay -- two runs of the XML entry
    point give the same bits, shot noise included (with float atomics a last bit may flip a Poisson draw); the images stay
    within 1e-5 of the float-atomic run before the noise."""
    from paresis_amd import main
    runs = []
    for k, rep in enumerate((True, True, False)):
        ed = {"experimentName": "Fil_Nylon_ID17", "filepath": str(tmp_path) + "/r%d/" % k, "overSampling": 2, "nbExpPoints": 2,
              "simulation_type": "RayT", "noise": rep, "seed": 4, "reproducible": rep}
        os.makedirs(ed["filepath"])
        runs.append(main.run(ed, save=False))
    for p in range(2):
        for a, b in zip(runs[0][p][:2], runs[1][p][:2]):
            assert np.array_equal(a.numpy(), b.numpy()), p
    # noise-free: reproducible vs float atomics agree to float rounding
    ed = {"experimentName": "Fil_Nylon_ID17", "filepath": str(tmp_path) + "/r3/", "overSampling": 2, "nbExpPoints": 1,
          "simulation_type": "RayT", "noise": False, "seed": 4, "reproducible": True}
    os.makedirs(ed["filepath"])
    quiet = main.run(ed, save=False)
    for a, b in zip(quiet[0][:2], runs[2][0][:2]):
        assert float(np.abs(a.numpy() - b.numpy()).max() / np.abs(b.numpy()).max()) < 2e-6


_RCCL_ONE_RANK = r'''
import os, sys, torch
import torch.distributed as td
sys.path.insert(0, os.getcwd())
from paresis_amd import dist
torch.cuda.set_device(0)
td.init_process_group(backend="nccl", rank=0, world_size=1)
g = torch.Generator(device="cuda").manual_seed(3)
def images(p, frac=False):
    S = torch.randint(0, 60000, (2, 64, 96), generator=g, device="cuda").to(torch.float32)
    R = torch.randint(0, 60000, (2, 64, 96), generator=g, device="cuda").to(torch.float32)
    S[0, 1, 2], R[1, 3, 4] = 70000.0 + p, 65535.0
    if frac:
        R[0, 0, 0] = 0.5
    return S, R
for frac in (False, True):
    res = {p: images(p, frac and p == 2) for p in range(3)}
    res[0] = res[0] + (torch.full((2, 64, 96), 3.5, device="cuda"), torch.full((2, 64, 96), 4.0, device="cuda"))
    for to_host in (False, True):
        out = dist.gather_positions(res, 3, 0, 1, to_host=to_host, force_collectives=True)
        assert dist.last_gather["packed"] == (not frac), dist.last_gather
        assert sorted(out) == [0, 1, 2] and len(out[0]) == 4
        for p in range(3):
            for a, b in zip(out[p][:2], res[p][:2]):
                assert a.is_cuda != to_host and torch.equal(a.cpu(), b.cpu()), (frac, to_host, p)
    # the same round by round, each gather issued without waiting (async_op on RCCL's stream)
    for to_host in (False, True):
        gat = dist.PositionGatherer(3, 0, 1, to_host=to_host, shape=(2, 64, 96), force_collectives=True)
        for p in range(3):
            gat.add(p, res[p])
        out = gat.finish()
        assert dist.last_gather["packed"] == (not frac) and sorted(out) == [0, 1, 2] and len(out[0]) == 4
        for p in range(3):
            for a, b in zip(out[p][:2], res[p][:2]):
                assert a.is_cuda != to_host and torch.equal(a.cpu(), b.cpu()), (frac, to_host, p)
td.barrier()
td.destroy_process_group()
print("RCCL-ONE-RANK-OK")
'''


def test_gather_path_on_rccl_one_rank(tmp_path):
    """The collectives of dist.gather_positions on the backend the 8-GPU run uses (RCCL), in a one-rank group -- all a
    one-GPU box allows: the byte-typed gather of the packed counts, the int32 / int64 all_reduce(MAX), the float32
    fallback, images left in HBM or brought to the host."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_RCCL_ONE_RANK)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, str(script)], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
