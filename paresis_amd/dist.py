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


def finish():
    if td.is_available() and td.is_initialized():
        td.barrier()


def my_positions(n_positions, rank, world):
    """Static strided partition: rank r owns r, r+world, ..."""
    return list(range(rank, n_positions, world))


def owner(position, world):
    return position % world


def gather_positions(results, n_positions, rank, world, dst=0):
    """results: {position: tuple of tensors} computed on this rank.  Returns on `dst` a dict with every position
    (tensors on the host), {} elsewhere.  One fixed-shape gather per round of positions: every rank contributes its
    Sample/Reference stacks of round t (position t*world + rank), padded with zeros when it has none."""
    host = lambda tup: tuple(t.detach().cpu() if isinstance(t, torch.Tensor) else t for t in tup)
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
    for t in range(rounds):
        p = t * world + rank
        mine = torch.zeros((2,) + shape, dtype=torch.float32, device=_dev())
        if p in results:
            mine[0] = results[p][0].to(mine.device, torch.float32)
            mine[1] = results[p][1].to(mine.device, torch.float32)
        bucket = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
        td.gather(mine, bucket, dst=dst)
        if rank == dst:
            for r in range(world):
                q = t * world + r
                if q < n_positions:
                    out[q] = (bucket[r][0].cpu(), bucket[r][1].cpu())
    if rank == dst and 0 in results:                     # Propag / White / Dx,Dy exist for position 0 only
        out[0] = host(results[0])
    elif owner(0, world) != dst:
        pass
    return out if rank == dst else {}


def _dev():
    if td.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")
