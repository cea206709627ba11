#!/usr/bin/env python3
"""Copy the outputs of scripts/refresh_profiles.sh <tag> (gpurun_out/) into profiles/r01_* and rebuild r01_pmc_count_walk.json."""
import csv, glob, json, shutil, collections, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
tag = sys.argv[1]; rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
src = R + "gpurun_out/refresh_%s/" % tag
for a, b in (("bench_line", "bench_line"), ("bench_scans_line", "bench_scans_line"), ("bench_perm_line", "bench_perm_line"),
             ("bench_line_under_rocprof", "bench_line_under_rocprof"), ("bench_line_two_streams", "bench_line_two_streams")):
    shutil.copy(src + a + ".json", R + "profiles/%s_%s.json" % (rnd, b))
shutil.copy(glob.glob(src + "stats/*/*kernel_stats.csv")[0], R + "profiles/%s_bench_kernel_stats.csv" % rnd)
for w in ("scans", "perm"):
    g = glob.glob(src + "stats_%s/*/*kernel_stats.csv" % w)
    if g:
        shutil.copy(g[0], R + "profiles/%s_bench_%s_kernel_stats.csv" % (rnd, w))
lines, keep = [], False
for l in open(src + "pmc_summary.txt"):
    if l.startswith("=="):
        keep = ("pmc_%s_" % tag) in l
    if keep:
        lines.append(l)
open(R + "profiles/%s_pmc_summary.txt" % rnd, "w").write("".join(lines))
ctr = {}
for f in glob.glob(R + "gpurun_out/pmc_%s_*/*/*counter_collection.csv" % tag):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "count_walk_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        ctr[k] = sum(v) / len(v)
fetch = ctr["FETCH_SIZE"] * 1024 * 2; wr = ctr["WRITE_SIZE"] * 1024
out = {"kernel": "count_walk_kernel<false,4>",
       "command": "rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 (scripts/pmc.sh, separate passes; scripts/refresh_profiles.sh)",
       "FETCH_SIZE_KB": ctr["FETCH_SIZE"], "WRITE_SIZE_KB": ctr["WRITE_SIZE"], "fetch_bytes_corrected": fetch, "write_bytes": wr,
       "traffic_bytes_per_launch": fetch + wr,
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streaming reads -> x2 (MI355X_MICROARCH.md, HBM); calibrated here by the known 1.2 GB of triples; WRITE_SIZE as is",
       "algorithmic_bytes_per_launch": 1.2e9, "counters": ctr}
json.dump(out, open(R + "profiles/%s_pmc_count_walk.json" % rnd, "w"), indent=1)
d = json.load(open(R + "profiles/%s_bench_line.json" % rnd))
d2 = json.load(open(R + "profiles/%s_bench_line_two_streams.json" % rnd))
d["two_streams"] = d2.get("two_streams")
print("value %.4g  ms/step %.4f  kernel_ms %.4f  frac %.3f  traffic %.4g  two_streams %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], out["traffic_bytes_per_launch"], (d.get("two_streams") or {}).get("value")))
