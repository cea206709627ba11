#!/usr/bin/env python3
"""Copy the outputs of scripts/refresh.sh <tag> (gpurun_out/refresh_<tag>/) into profiles/<tag>_* and build the PMC summaries.
FETCH_SIZE is in KB and is doubled for gfx950 as MI355X_MICROARCH.md prescribes (128-B requests tallied at 64 B); WRITE_SIZE as is."""
import collections, csv, glob, hashlib, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = R + "gpurun_out/refresh_%s/" % rnd


def newest(pattern):
    g = sorted(glob.glob(pattern), key=os.path.getmtime)
    return g[-1] if g else None


def sha16(files=("gtx_kernels.hip", "gtx_kernels.h", "gtx_capi.hip")):
    h = hashlib.sha256()
    for f in files:
        h.update(open(R + "ibm-cbc-genomic-tools_amd/csrc/" + f, "rb").read())
    return h.hexdigest()[:16]


for a in ("bench_line", "bench_scans_line", "bench_perm_line", "bench_c5_line", "bench_line_under_rocprof", "bench_rehearse4_line", "bench_selftest_line", "bench_fallback2_line"):
    if os.path.exists(src + a + ".json") and os.path.getsize(src + a + ".json"):
        shutil.copy(src + a + ".json", R + "profiles/%s_%s.json" % (rnd, a))
for tag, name in (("stats", "bench"), ("stats_scans", "bench_scans"), ("stats_perm", "bench_perm"), ("stats_c5", "c5_bench"), ("stats_bucket", "bucket"), ("stats_covshuf", "cov_shuffled"),
                  ("stats_cov", "coverage"), ("stats_scanfine", "scan_geometries"), ("stats_share", "share_member"), ("stats_pairs", "pairs")):
    f = newest(src + tag + "/*/*kernel_stats.csv")
    if f:
        shutil.copy(f, R + "profiles/%s_%s_kernel_stats.csv" % (rnd, name))
for t in ("share_timing.txt", "share_timing_1g.txt", "share_streams.txt", "membench_100m.txt", "membench_1g.txt", "ab_sched.txt"):
    if os.path.exists(src + t):
        shutil.copy(src + t, R + "profiles/%s_%s" % (rnd, t))
if os.path.exists(src + "wave_trace.txt"):
    open(R + "profiles/%s_wave_trace.txt" % rnd, "w").write("".join(l for l in open(src + "wave_trace.txt") if l.startswith(("{", "alive", "streaming"))))
for log in ("bench_bucket.log", "bench_cov.log", "bench_scan.log", "bench_covshuf.log", "bench_scanshuf.log", "bench_pairs.log", "bench_weighted.log", "bench_covw.log", "bench_sort.log"):
    if os.path.exists(src + log):
        keep = [l for l in open(src + log) if ("bucket path" in l or "coverage:" in l or "coverage, " in l or "scan -w" in l or "bit-equal" in l or "per call" in l or "gtx_set_ref_blocks" in l or "gtx_count_add_regions" in l or "weighted" in l or "gtx_sort_device" in l or "sortbed" in l or "LC_ALL" in l or l.startswith("sorted:"))]
        open(R + "profiles/%s_%s.txt" % (rnd, log[:-4]), "w").write("".join(keep))


def counters(name, match):
    out = {}
    f = newest(src + "pmc_%s/*/*counter_collection.csv" % name)
    if not f:
        return out
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if match in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def pmc_json(path, kernel, match, fetch, write, extra, alg, note):
    c = {}
    n = {}
    for nm in (fetch, write) + tuple(extra):
        r = counters(nm, match)
        if r:
            c.update(r[0]); n.update(r[1])
    if "FETCH_SIZE" not in c:
        print("no counters for", kernel); return
    fb = c["FETCH_SIZE"] * 1024 * 2; wb = c.get("WRITE_SIZE", 0.0) * 1024
    d = {"kernel": kernel, "command": "scripts/refresh.sh %s pmc (rocprofv3 --kernel-trace --pmc <counter>, one counter set per run)" % rnd,
         "kernel_source_sha16": sha16(), "counters_mean_per_dispatch": c, "dispatches": n,
         "fetch_bytes_corrected": fb, "write_bytes": wb, "traffic_bytes_per_launch": fb + wb,
         "correction": "gfx950: FETCH_SIZE (KB) tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE (KB) as is",
         "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fb + wb) / alg if alg else None, "note": note}
    if "TCC_HIT_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    json.dump(d, open(path, "w"), indent=1)
    print(os.path.basename(path), "traffic %.4g B = %.2fx algorithmic" % (fb + wb, (fb + wb) / alg), "L2 hit %.3f" % d["l2_hit_rate"] if "l2_hit_rate" in d else "")


pmc_json(R + "profiles/%s_pmc_count_walk.json" % rnd, "count_walk_kernel<false,4>", "count_walk_kernel", "count_fetch", "count_write", (), 1.2e9,
         "100 M reads x 1 M regions, bench.py default workload")
pmc_json(R + "profiles/%s_pmc_coverage_walk.json" % rnd, "coverage_walk_kernel<false>", "coverage_walk_kernel", "cov_fetch", "cov_write", (), 1.2e9,
         "100 M reads x 1 M regions, tests/tools/bench_coverage.py")
for k, alg, note in (("bucket_scatter_kernel", 2.0e9, "reads once + (start, end) pairs written once"),
                     ("bucket_count_kernel", 0.8e9, "(start, end) pairs read once")):
    pmc_json(R + "profiles/%s_pmc_%s.json" % (rnd, k), k, k, "bucket_fetch", "bucket_write", (), alg, "100 M shuffled reads x 1 M regions, scripts/bench_bucket.py; " + note)
# perm: per batch TWO launches of perm_stat_kernel (row-range parts): per-launch means are doubled to give the traffic of a 10 k-shuffle batch
line = json.load(open(R + "profiles/%s_bench_perm_line.json" % rnd)) if os.path.exists(R + "profiles/%s_bench_perm_line.json" % rnd) else None
r = [counters(nm, "perm_stat_kernel") for nm in ("perm_fetch", "perm_write", "perm_l2")]
if all(r) and "FETCH_SIZE" in r[0][0]:
    c = {}; [c.update(x[0]) for x in r]
    parts = 2
    unique = line["roofline"]["algorithmic_bytes"] if line else 0.81e9
    fb = c["FETCH_SIZE"] * 1024 * 2 * parts; wb = c["WRITE_SIZE"] * 1024 * parts
    d = {"kernel": "perm_stat_kernel<GTX_STAT_SUM,false,false,MODE_GE>, 2 row-range launches per 10 k-shuffle batch", "kernel_source_sha16": sha16(), "perm_source_sha16": sha16(("gtx_perm.hip",)),
         "counters_mean_per_launch": c, "launches_per_batch": parts, "fetch_bytes_corrected_per_batch": fb, "write_bytes_per_batch": wb,
         "traffic_bytes_per_launch": fb + wb, "unique_bytes_per_batch": unique, "traffic_over_unique": (fb + wb) / unique,
         "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
         "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md); L2's memory-side traffic, includes Infinity-Cache hits; includes the accumulators carried between the two launches"}
    json.dump(d, open(R + "profiles/%s_pmc_perm_stat.json" % rnd, "w"), indent=1)
    print("perm: traffic %.4g B per batch = %.2fx unique, L2 hit %.3f" % (fb + wb, (fb + wb) / unique, d["l2_hit_rate"]))
