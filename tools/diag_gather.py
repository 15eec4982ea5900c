#!/usr/bin/env python3
"""Development aid: why does a 2-rank gloo run of Bench_4096 on one GPU fall back from the packed gather?  Logs, per pack() call
of the host restatement, the image statistics that can raise the flag.  python tools/diag_gather.py <outdir>"""
import os
import socket
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def rank_main(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      PARESIS_ALLOW_SYNTHETIC_MATERIALS="1")
    import torch as th
    from paresis_amd import dist, main
    orig = dist._CountsWire.pack
    log = open(os.path.join(outdir, "pack_rank%d.log" % rank), "w")

    def pack(self, img, index0, flag):
        before = int(flag.item())
        x = img.detach().float().cpu().view(-1)
        orig(self, img, index0, flag)
        log.write("index0 %d n %d dev %s min %.1f max %.1f >=65535 %d nonint %d nan %d cap %d head %d flag %d->%d\n" % (
            index0, x.numel(), img.device, x.min().item(), x.max().item(), int((x >= 65535).sum()), int((x != x.round()).sum()),
            int(th.isnan(x).sum()), self.cap, int(self.head[0]), before, int(flag.item())))
        log.flush()
    dist._CountsWire.pack = pack
    ed = {"experimentName": "Bench_4096", "filepath": outdir + "/", "overSampling": 2, "nbExpPoints": 4,
          "simulation_type": "Fresnel", "noise": True, "seed": 21}
    main.run(ed, save=False, backend="gloo")
    log.write("last_gather %s\n" % dict(dist.last_gather))
    log.close()
    th.distributed.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=rank_main, args=(r, 2, port, out)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(600)
    print("exit codes", [p.exitcode for p in ps])
