"""Multi-GPU sharding of the membrane-position loop (main.py:63 of the reference is a plain serial loop).

One process per GPU (torchrun).  Positions are independent given seed(pointNum), so rank r takes positions
r, r+G, r+2G, ... with no data-path collective; the only exchange is the final gather of the detector images onto
rank 0 with torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the tests).
"""
import os

import torch
import torch.distributed as td


def init(backend=None):
    """Returns (rank, world).  A plain `python` launch (no RANK in the environment) is world size 1."""
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return 0, 1
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if not td.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(1, torch.cuda.device_count()))
        td.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def barrier():
    if td.is_available() and td.is_initialized():
        td.barrier()


def broadcast_object(obj, rank, world, src=0):
    """The same small Python object on every rank (rank `src`'s)."""
    if world == 1:
        return obj
    box = [obj if rank == src else None]
    td.broadcast_object_list(box, src=src, device=_dev())
    return box[0]


def finish():
    if td.is_available() and td.is_initialized():
        td.barrier()


def my_positions(n_positions, rank, world):
    """Static strided partition: rank r owns r, r+world, ..."""
    return list(range(rank, n_positions, world))


def owner(position, world):
    return position % world


def gather_positions(results, n_positions, rank, world, dst=0, to_host=True):
    """results: {position: tuple of tensors} computed on this rank.  Returns on `dst` a dict with every position
    (tensors on the host, or left in `dst`'s HBM with to_host=False), {} elsewhere.

    ONE fixed-shape gather for the whole run: every rank contributes the Sample/Reference stacks of its positions
    [rounds][2][nbins][n][n] (slot t = position t*world + rank, zeros when it has none) -- 7 point-to-point transfers into
    rank `dst` over xGMI, no ring."""
    host = lambda tup: tuple(t.detach().cpu() if isinstance(t, torch.Tensor) and to_host else t for t in tup)
    if world == 1:
        return {p: host(v) for p, v in results.items()}
    out = {}
    rounds = (n_positions + world - 1) // world
    proto = None
    for v in results.values():
        proto = v
        break
    shape_t = torch.zeros(4, dtype=torch.int64, device=_dev())
    if proto is not None:
        shape_t[:3] = torch.tensor(proto[0].shape, dtype=torch.int64)
        shape_t[3] = 1
    td.all_reduce(shape_t, op=td.ReduceOp.MAX)          # ranks without work learn the stack shape
    shape = tuple(int(v) for v in shape_t[:3])
    mine = torch.empty((rounds, 2) + shape, dtype=torch.float32, device=_dev())
    for t in range(rounds):
        p = t * world + rank
        if p in results:
            mine[t, 0].copy_(results[p][0])
            mine[t, 1].copy_(results[p][1])
        else:
            mine[t].zero_()
    bucket = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    td.gather(mine, bucket, dst=dst)
    if rank == dst:
        if to_host:
            bucket = [b.cpu() for b in bucket]
        for r in range(world):
            for t in range(rounds):
                q = t * world + r
                if q < n_positions:
                    out[q] = (bucket[r][t, 0], bucket[r][t, 1])
    if rank == dst and 0 in results:                     # Propag / White / Dx,Dy exist for position 0 only (owner(0) == 0)
        out[0] = host(results[0])
    return out if rank == dst else {}


def _dev():
    if td.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")
