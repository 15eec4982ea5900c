"""The N>1 path on CPU: world-size-2 gloo run of the position sharding + final image gather (paresis_amd/dist.py)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_positions, q, kind="counts", overlapped=False, dst=0):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from paresis_amd import dist
    r, w = dist.init(backend="gloo")
    assert (r, w) == (rank, world)
    mine = dist.my_positions(n_positions, r, w)
    # stand-in for computeSampleAndReferenceImages: images that encode (position, kind); position 0 has Propag/White too
    results = {}
    # "counts": integer images up to 65535 (cross as 16-bit); "fractions": noise-free images (float32 on the wire);
    # "one_rank_fractions": only rank 1 holds a non-integer pixel -- every rank must then fall back together
    def images(p):
        S = torch.full((2, 5, 7), 10.0 * p + 1.0)
        R = torch.full((2, 5, 7), 10.0 * p + 2.0)
        S[0, 0, 1], R[1, 0, 0], R[0, 1, 1], S[1, 4, 6] = 65535.0, 32768.0, 65536.0 + p, 16777216.0    # escapes included
        if kind == "fractions" or (kind == "one_rank_fractions" and p == 1):
            S[1, 2, 3] = 0.25
        if kind == "too_large" and p == 2:
            R[0, 1, 2] = 16777218.0                       # an integer in float32, beyond the packed range
        if kind == "many_bright" and p == 3:
            S[:] = 70000.0                                # more escapes than the table holds
        return S, R
    for p in mine:
        S, R = images(p)
        results[p] = (S, R, torch.full((2, 5, 7), 3.5), torch.full((2, 5, 7), 4.0)) if p == 0 else (S, R)
    if overlapped:                    # round by round, each gather issued as soon as the rank's position of the round exists
        gat = dist.PositionGatherer(n_positions, r, w, dst=dst, to_host=True, shape=(2, 5, 7))
        for p in mine:
            gat.add(p, results[p])
        out = gat.finish()
    else:
        out = dist.gather_positions(results, n_positions, r, w, dst=dst)
    packed = dist.last_gather.get("packed")
    dist.finish()
    if r == dst:
        ok = sorted(out) == list(range(n_positions)) and packed == (kind == "counts")
        per_round = 2 * 70 * 2 + 8 + 8 * 64
        rounds = (n_positions + world - 1) // world
        if world == 2 and n_positions == 5:        # the byte counts of the original two-rank case, spelled out
            ok = ok and dist.last_gather["wire_bytes"] == ((2 * per_round if overlapped else 3 * 2 * 70 * 2 + 8 + 8 * 64) if packed
                                                           else 3 * 2 * 70 * 4)
        elif packed and overlapped:                # every position that is not the sink's crosses once, as one packed round
            ok = ok and dist.last_gather["wire_bytes"] == (n_positions - len(mine)) * per_round
        elif not packed:
            ok = ok and dist.last_gather["wire_bytes"] == (world - 1) * rounds * 2 * 70 * 4
        for p in range(n_positions):
            S, R = images(p)
            ok = ok and torch.equal(out[p][0], S) and torch.equal(out[p][1], R) and out[p][0].dtype == torch.float32
        ok = ok and len(out[0]) == 4 and float(out[0][3][0, 0, 0]) == 4.0 and float(out[0][2][0, 0, 0]) == 3.5
        q.put(bool(ok))
    else:
        q.put(out == {})
    torch.distributed.destroy_process_group()


import pytest


@pytest.mark.parametrize("overlapped", [False, True])
@pytest.mark.parametrize("kind", ["counts", "fractions", "one_rank_fractions", "too_large", "many_bright"])
def test_gather_positions_world2(kind, overlapped):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, q, kind, overlapped)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res)


@pytest.mark.parametrize("world,n_positions,overlapped,kind", [(3, 7, True, "counts"), (3, 7, False, "one_rank_fractions"),
                                                                (4, 6, True, "counts"), (4, 3, True, "counts")])
def test_gather_positions_more_ranks_and_ragged_rounds(world, n_positions, overlapped, kind):
    """The gather with 3 and 4 ranks, a position count that does not divide (the last round is partly empty) and fewer
    positions than ranks (a rank that owns none): what 8 ranks with 64 positions never exercise, and the indexing an 8-rank
    run relies on."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_positions, q, kind, overlapped)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res)


def _run_ranks(world, n_positions, kind, overlapped, dst, timeout=240):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_positions, q, kind, overlapped, dst)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res)


def test_gather_world8_64_positions_like_the_scale_run():
    """The exact indexing of the driver's 8-GPU run (BASELINE config 4: 64 positions over 8 ranks, 8 full rounds, round by
    round through PositionGatherer with the packed wire format), on gloo with small images."""
    _run_ranks(8, 64, "counts", True, 0)


@pytest.mark.parametrize("world,n_positions,overlapped,dst", [(8, 64, True, 7), (4, 6, True, 2), (3, 7, False, 1), (2, 1, True, 1)])
def test_gather_sink_is_not_the_owner_of_position_zero(world, n_positions, overlapped, dst):
    """dst != 0: rank 0 computes position 0 with its extras (Propag / White: the straggler), another rank receives every
    position (the sink).  The extras reach the sink by one point-to-point transfer; everything else is indexed as before."""
    _run_ranks(world, n_positions, "counts", overlapped, dst)


def test_gather_sink_elsewhere_float_fallback():
    _run_ranks(3, 5, "one_rank_fractions", True, 2)


def _worker_missing_zero(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    from paresis_amd import dist
    r, w = dist.init(backend="gloo")
    results = {} if r == 0 else {1: (torch.ones((1, 3, 4)), torch.ones((1, 3, 4)))}      # rank 0 never computed position 0
    t0 = time.monotonic()
    try:
        dist.gather_positions(results, 2, r, w, dst=1, to_host=True, pack=False, timeout=20)
        q.put(("no error", r, 0.0))
    except dist.DistError as exc:
        q.put(("DistError", r, time.monotonic() - t0))
    q.close()
    q.join_thread()      # the queue's feeder thread has written the result
    os._exit(0)          # the contract after a DistError: leave without tearing the group down


def test_owner_without_position_zero_fails_both_ends_at_once():
    """with a foreign sink, position 0's extras cross point to point.  An owner that never computed position 0 used
    to raise KeyError on its side while the sink sat in recv until the process-group timeout; now the header says so and both
    ends raise DistError within the transfer's own deadline (here: at once, far below the 20 s given)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_missing_zero, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(r[1] for r in res) == [0, 1]
    assert all(r[0] == "DistError" and r[2] < 10.0 for r in res), res


def test_single_process_is_world_one(monkeypatch):
    monkeypatch.delenv("RANK", raising=False)
    from paresis_amd import dist
    assert dist.init() == (0, 1)
    out = dist.gather_positions({0: (torch.ones(1, 2, 2), torch.zeros(1, 2, 2))}, 1, 0, 1)
    assert list(out) == [0] and out[0][0].shape == (1, 2, 2)


# ---- main.run across two ranks (CPU): the REAL entry point -- XML load, position sharding, the once-per-Experiment bin
# thresholds, the single gather, the saves -- with the GPU compute replaced by the oracle (test infrastructure) and the
# membrane synthesis by the seeded numpy stand-in.  Rank 1 never computes position 0: the case ADVICE r1 flagged.
def _patch_cpu_compute():
    import numpy as np
    from oracle import paresis_oracle as orc
    from paresis_amd import synth
    from paresis_amd.Experiment import Experiment
    from paresis_amd.Sample import AnalyticalSample
    from tests._build import cfg_from_experiment

    real_geometry = AnalyticalSample.getMyGeometry

    def geometry(self, dims, pix, ov, pointNum=0, number_of_positions=0):
        if self.myType == "membrane":
            self.myGeometry = np.stack([synth.sphere_membrane(int(dims[0]), int(dims[1]), pix * 1e-6, pointNum),
                                        synth.slab(int(dims[0]), int(dims[1]), 6e-3)])
            return
        return real_geometry(self, dims, pix, ov, pointNum, number_of_positions)

    def compute(self, pointNum):
        nbins = self._close_bins()                       # the product's own threshold logic
        cfg = cfg_from_experiment(self, orc.Obj)
        # the oracle follows the serial reference literally: it closes the thresholds when it is given position 0 and
        # expects them closed already otherwise (EXP:296-301)
        thr = list(self.myDetector.det_param["myBinsThersholds"])
        cfg["bins"] = thr[:-1] if pointNum == 0 else thr
        ref = orc.compute_rt(cfg, pointNum) if self.exp_dict["simulation_type"] == "RayT" else orc.compute_fresnel(cfg, pointNum)
        out = tuple(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) for a in ref[:4])
        assert out[0].shape[0] == nbins
        return out if pointNum == 0 else out[:2]

    AnalyticalSample.getMyGeometry = geometry
    Experiment.computeSampleAndReferenceImages = compute


def _main_worker(rank, world, port, outdir, q):
    if world > 1:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
    else:
        os.environ.pop("RANK", None)
        os.environ["WORLD_SIZE"] = "1"
    os.environ["PARESIS_ALLOW_SYNTHETIC_MATERIALS"] = "1"
    _patch_cpu_compute()
    from paresis_amd import main
    ed = {"experimentName": "Fil_Nylon_ID17", "filepath": outdir + "/", "overSampling": 1, "nbExpPoints": 3,
          "simulation_type": "RayT", "noise": False, "saveMembrane": True}
    res = main.run(ed, save=True, saving_format=".edf", backend="gloo")
    if rank == 0:
        q.put({p: [t.numpy() for t in v[:2]] for p, v in res.items()})
    else:
        q.put(res == {})
    if world > 1:
        torch.distributed.destroy_process_group()


def test_main_run_world2_matches_world1(tmp_path):
    import glob
    import numpy as np
    ctx = mp.get_context("spawn")
    runs = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        out = str(tmp_path / ("w%d" % world))
        os.makedirs(out)
        procs = [ctx.Process(target=_main_worker, args=(r, world, port, out, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        runs[world] = next(r for r in res if isinstance(r, dict))
        assert all(r is True for r in res if not isinstance(r, dict))
        # rank 0 wrote every position's images, the ranks together every membrane map (main.py:98-110)
        assert len(glob.glob(out + "/RayTracing_*/sample/*.edf")) == 3
        assert len(glob.glob(out + "/RayTracing_*/ref/*.edf")) == 3
        assert len(glob.glob(out + "/RayTracing_*/membraneThickness/*.edf")) == 3
        assert len(glob.glob(out + "/RayTracing_*/DF.edf")) == 1
        assert len(glob.glob(out + "/RayTracing_*/propag/*.edf")) == 1
    assert sorted(runs[1]) == sorted(runs[2]) == [0, 1, 2]
    for p in range(3):
        for a, b in zip(runs[1][p], runs[2][p]):
            assert np.array_equal(a, b), p


# ---- round 3: the first contact with 8 GPUs must not hang -----------------------------
def test_more_ranks_than_gpus_is_refused_before_any_collective(monkeypatch):
    """`nccl` needs one GPU per rank: world 2 on a 1-GPU lease used to reach td.barrier() and die there with
    `ncclInvalidUsage: Duplicate GPU detected`; now the rank-to-device mapping refuses up front.  gloo may share a GPU."""
    from paresis_amd import dist
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        dist.local_device("nccl", 1, 2, n_devices=1)
    assert "one GPU per rank" in str(e.value)
    assert dist.local_device("gloo", 1, 2, n_devices=1) == 0
    assert dist.local_device("nccl", 1, 2, n_devices=8) == 1
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert dist.local_device("nccl", 3, 8, n_devices=8) == 3


def _fallback_worker(rank, world, port, q, mode):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), PSX_DIST_TIMEOUT_S="20")
    import time
    from paresis_amd import dist
    r, w = dist.init(backend="gloo")
    P = 4
    img = lambda p: (torch.full((1, 3, 4), float(p + 1)), torch.full((1, 3, 4), float(10 * p + 1)))
    gat = dist.PositionGatherer(P, r, w, to_host=True, shape=(1, 3, 4))
    if mode == "prepare":
        # rank 1 cannot allocate its buffers: BOTH ranks must fall back to the one-gather form, in the same collective order
        overlapped = dist.agree_on_overlap(gat, inject_failure=(r == 1))
        assert overlapped is False
        out = dist.gather_positions({p: img(p) for p in dist.my_positions(P, r, w)}, P, r, w)
        if r == 0:
            q.put(sorted(out) == list(range(P)) and all(torch.equal(out[p][0], img(p)[0]) for p in range(P)))
        else:
            q.put(out == {})
        dist.finish()
        torch.distributed.destroy_process_group()
        return
    # modes "warmup" / "silent": both agreed on the overlapped form; rank 1 then dies (or goes quiet) BEFORE issuing its
    # gathers.  Rank 0 must notice -- a transport error, or the bounded wait -- and leave non-zero, not hang until a
    # process-group timeout.
    assert dist.agree_on_overlap(gat) is True
    if r == 1:
        if mode == "silent":          # alive, but never issues its gathers: only rank 0's bounded wait can end this
            time.sleep(8.0)
        os._exit(5)
    t0 = time.monotonic()
    try:
        for p in dist.my_positions(P, r, w):
            gat.add(p, img(p))
        gat.finish(timeout=3.0)
    except dist.DistError:
        q.put(time.monotonic() - t0)
        q.close()
        q.join_thread()               # os._exit does not flush the queue's feeder thread
        os._exit(6)
    q.put(-1.0)


@pytest.mark.parametrize("mode", ["prepare", "warmup", "silent"])
def test_gather_fallback_is_a_collective_decision_and_a_dead_rank_does_not_hang(mode):
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    t0 = time.monotonic()
    for p in procs:
        p.start()
    if mode == "prepare":
        res = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert all(r is True for r in res)
        return
    waited = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
    assert [p.exitcode for p in procs] == [6, 5]          # both ranks left non-zero ...
    assert waited < 10.0                                   # ... rank 0 within seconds, not after a process-group timeout
    if mode == "silent":
        assert waited > 2.5                                # (here it was the bounded wait of finish(timeout=3) that ended it)
    assert time.monotonic() - t0 < 60
