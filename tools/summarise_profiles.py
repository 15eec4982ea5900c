#!/usr/bin/env python3
"""Turns the rocprofv3 outputs under gpurun_out/fin_{stats,fetch,write,sq}/ into the committed summaries in profiles/."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
sfx = sys.argv[2] if len(sys.argv) > 2 else ""          # "_16384": the run collected with tools/collect_profiles.sh _16384 ...
os.makedirs("profiles", exist_ok=True)
newest = lambda pat: sorted(glob.glob(pat) + glob.glob(pat.replace("/runc/*", "/runc_")), key=os.path.getmtime)[-1]
shutil.copy(newest("gpurun_out/fin_stats%s/runc/*kernel_stats.csv" % sfx), "profiles/%s_kernel_stats%s.csv" % (tag, sfx))


def agg(d):
    rows = list(csv.DictReader(open(newest("gpurun_out/%s%s/runc/*counter_collection.csv" % (d, sfx)))))
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        name = r["Kernel_Name"]
        if "anonymous" not in name:
            continue
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]     # with its template arguments
        out[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}


fetch, write, sq = agg("fin_fetch"), agg("fin_write"), agg("fin_sq")
summary = {}
for k in sorted(set(fetch) | set(write) | set(sq)):
    if not (k.startswith("k_fresnel") or k.startswith("k_refract") or k.startswith("k_source") or k.startswith("k_band") or k.startswith("k_psf")):
        continue
    e = {}
    if k in fetch:
        e["FETCH_SIZE_KB"] = fetch[k]["FETCH_SIZE"]
    if k in write:
        e["WRITE_SIZE_KB"] = write[k]["WRITE_SIZE"]
    if k in fetch and k in write:
        e["hbm_bytes_per_launch"] = int((2 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"]) * 1024)
    e.update(sq.get(k, {}))
    summary[k] = e
cmd = ("`python3 bench.py --only-configs --configs 16384 --no-config-parity` (config 5 as the driver's line runs it: GPU-synthesised "
       "membrane, halo picked by ops.tune_refract_halo -- its three candidates appear as three k_refract instances --, detector "
       "inside the step)") if sfx == "_cfg5" else (
       "`python3 bench.py --no-cpu-baseline --positions 0%s` (warm-up + timed steps + the per-kernel event pass)" % (" (" + sfx.strip("_") + "^2 grid)" if sfx else ""))
def csrc_sha1(root="."):
    """One hash over the kernel sources (paresis_amd/csrc/*.hip, *.hpp, Makefile, sorted by name): stamped into the summary so that
    bench.py can say whether the profile it quotes was collected on the sources it runs (roofline.traffic_sources_match)."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(root, "paresis_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")) or f == "Makefile":
            h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


json.dump({"csrc_sha1": csrc_sha1(), "note": "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE and the SQ counters each in its own run) of " + cmd + " on one MI355X; averages per " +
                   "launch.  hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE reads half of a wide "
                   "coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact.",
           "kernels": summary}, open("profiles/%s_pmc_summary%s.json" % (tag, sfx), "w"), indent=1)
for k, e in summary.items():
    print(k, {a: ("%.4g" % b if isinstance(b, float) else b) for a, b in e.items()})
