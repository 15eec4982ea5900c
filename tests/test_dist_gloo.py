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


def _worker(rank, world, port, n_positions, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from paresis_amd import dist
    r, w = dist.init(backend="gloo")
    assert (r, w) == (rank, world)
    mine = dist.my_positions(n_positions, r, w)
    # stand-in for computeSampleAndReferenceImages: images that encode (position, kind); position 0 has Propag/White too
    results = {}
    for p in mine:
        S = torch.full((2, 5, 7), 10.0 * p + 1.0)
        R = torch.full((2, 5, 7), 10.0 * p + 2.0)
        results[p] = (S, R, torch.full((2, 5, 7), 3.0), torch.full((2, 5, 7), 4.0)) if p == 0 else (S, R)
    out = dist.gather_positions(results, n_positions, r, w)
    dist.finish()
    if r == 0:
        ok = sorted(out) == list(range(n_positions))
        for p in range(n_positions):
            ok = ok and float(out[p][0][0, 0, 0]) == 10.0 * p + 1.0 and float(out[p][1][1, 4, 6]) == 10.0 * p + 2.0
        ok = ok and len(out[0]) == 4 and float(out[0][3][0, 0, 0]) == 4.0
        q.put(bool(ok))
    else:
        q.put(out == {})
    torch.distributed.destroy_process_group()


def test_gather_positions_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res)


def test_single_process_is_world_one(monkeypatch):
    monkeypatch.delenv("RANK", raising=False)
    from paresis_amd import dist
    assert dist.init() == (0, 1)
    out = dist.gather_positions({0: (torch.ones(1, 2, 2), torch.zeros(1, 2, 2))}, 1, 0, 1)
    assert list(out) == [0] and out[0][0].shape == (1, 2, 2)
