#!/bin/bash
# Round 6, GPU session 36: rehearsal of bench.py multi-rank path on the final build: 2 and 4 ranks on the one GPU over gloo.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s36
mkdir -p $OUT
for n in 2 4; do
  timeout -k 10 500 python bench.py --gpus $n --backend gloo --positions 16 --no-configs --no-cpu-baseline --steps 5 --warmup 2 > $OUT/n$n.out 2> $OUT/n$n.err; echo "n=$n rc $?"
  python - <<PY
import json
d = json.loads(open("$OUT/n$n.out").read().strip().splitlines()[-1])
print(d.get("error") or {k: d[k] for k in ("value", "n_gpus", "ranks_seen")})
for k, v in d.get("positions_batch", {}).items():
    print(k, v["ms_total"], v["check"], v["gather_packed_u16"], v["gather_overlapped"], v["sink_rank"])
PY
done
timeout -k 10 500 python bench.py --gpus 3 --backend gloo --sink 2 --positions 7 --no-configs --no-cpu-baseline --steps 5 --warmup 2 > $OUT/n3.out 2> $OUT/n3.err; echo "n=3 sink 2 rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/n3.out").read().strip().splitlines()[-1])
for k, v in d.get("positions_batch", {}).items():
    print(k, v["ms_total"], v["check"], v["sink_rank"])
PY
# the driver's own form of the N > 1 launch (it starts torch.distributed.run itself), 2 ranks over gloo
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --positions 8 --no-configs --no-cpu-baseline --steps 5 --warmup 2 > $OUT/drv2.out 2> $OUT/drv2.err; echo "driver-style n=2 rc $?"
python - <<PY
import json
d = json.loads([l for l in open("$OUT/drv2.out").read().strip().splitlines() if l.startswith("{")][-1])
print(d.get("error") or {k: d[k] for k in ("value", "n_gpus", "ranks_seen", "far_rays")})
PY
