#!/usr/bin/env python3
"""Per-wave time line of the streaming count kernel (diagnostic build: make -C ibm-cbc-genomic-tools_amd/csrc trace).

GTX_LIB_PATH=.../libgtx_trace.so python scripts/wave_trace.py [--reads N] [--cpw A,B,...]
For every chunks-per-wave setting: kernel time by events, and from the wave stamps (100 MHz): how long the kernel
took from the first wave's start to the last wave's end, how long a wave spends placing its windows and streaming,
and how many waves are alive / streaming in every 4 us slice.  Writes gpurun_out/wave_trace.json.
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ibm-cbc-genomic-tools_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import gtx  # noqa: E402
from gtx import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--refs", type=int, default=1_000_000)
    ap.add_argument("--cpw", default="56")
    ap.add_argument("--sched", default="", help="';'-separated GTX_SCHED values tried for every --cpw ('-' = the default tail, 'none' = no tail)")
    ap.add_argument("--slice-us", type=float, default=4.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "wave_trace.json"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    refs = synth.genome_intervals(args.refs, 43, 50, 2000)
    per = synth.apportion(args.reads, synth.CHROM_LEN)
    reads = bench.make_reads_on_device(0, np.arange(24), 1000, dev, per=per)
    n = reads.shape[0]
    hits = torch.zeros(len(refs), dtype=torch.int64, device=dev)
    res = []
    combos = [(int(x), sc) for x in args.cpw.split(",") for sc in (args.sched.split(";") if args.sched else ["-"])]
    for cpw, sc in combos:
        os.environ["GTX_CHUNKS_PER_WAVE"] = str(cpw)
        os.environ.pop("GTX_SCHED", None)
        if sc == "none":
            os.environ["GTX_SCHED"] = "0x0"
        elif sc != "-":
            os.environ["GTX_SCHED"] = sc
        eng = gtx.Engine(0)
        eng.set_refs(refs, synth.n_classes())
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            eng.count_device(reads.data_ptr(), n, hits.data_ptr(), None, gtx.READS_SORTED)
        eng.profile(True)
        for _ in range(5):
            eng.count_device(reads.data_ptr(), n, hits.data_ptr(), None, gtx.READS_SORTED)
        eng.sync()
        kms = [eng.profile_last(b)[0] for b in range(5)]
        cpw_eff = (cpw + 3) // 4 * 4
        waves = ((n + 63) // 64 + 7) // 8 + 8                              # upper bound: no span is shorter than 8 chunks
        buf = np.zeros((waves, 4), dtype=np.uint64)
        rc = eng.lib.gtx_debug_trace_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(waves))
        assert rc == 0, rc
        widx = np.nonzero(buf[:, 2] != 0)[0]
        buf = buf[widx]; waves = len(buf)
        t0 = buf[:, 0].astype(np.int64); t1 = buf[:, 1].astype(np.int64); t2 = buf[:, 2].astype(np.int64)
        xcc = (buf[:, 3] & np.uint64(0xf)).astype(np.int64)
        base = t0.min()
        us = lambda t: (t - base) / 100.0
        s0, s1, s2 = us(t0), us(t1), us(t2)
        total = float(s2.max())
        nsl = int(total / args.slice_us) + 1
        alive = np.zeros(nsl); streaming = np.zeros(nsl)
        edges = np.arange(nsl + 1) * args.slice_us
        for k in range(nsl):
            lo, hi = edges[k], edges[k + 1]
            alive[k] = np.sum(np.clip(np.minimum(s2, hi) - np.maximum(s0, lo), 0, None)) / args.slice_us
            streaming[k] = np.sum(np.clip(np.minimum(s2, hi) - np.maximum(s1, lo), 0, None)) / args.slice_us
        r = {"cpw": cpw_eff, "sched": sc, "waves": int(waves), "kernel_ms_events": [round(float(x), 4) for x in kms],
             "span_us_first_start_to_last_end": total,
             "place_us": {"p10": float(np.percentile(s1 - s0, 10)), "p50": float(np.percentile(s1 - s0, 50)), "p90": float(np.percentile(s1 - s0, 90)), "max": float((s1 - s0).max())},
             "stream_us": {"p10": float(np.percentile(s2 - s1, 10)), "p50": float(np.percentile(s2 - s1, 50)), "p90": float(np.percentile(s2 - s1, 90)), "max": float((s2 - s1).max())},
             "last_start_us": float(s0.max()),
             "end_by_xcc_us": [float(s2[xcc == x].max()) if np.any(xcc == x) else None for x in range(8)],
             "waves_by_xcc": [int(np.sum(xcc == x)) for x in range(8)],
             "slice_us": args.slice_us, "alive": [int(x) for x in alive], "streaming": [int(x) for x in streaming]}
        res.append(r)
        np.savez_compressed(args.out.replace(".json", "_cpw%d_%d.npz" % (cpw_eff, len(res))), t0=t0 - base, t1=t1 - base, t2=t2 - base, hw=buf[:, 3],
                            wave=widx)
        print(json.dumps({k: v for k, v in r.items() if k not in ("alive", "streaming")}))
        print("alive    ", " ".join("%d" % (x // 100) for x in alive))
        print("streaming", " ".join("%d" % (x // 100) for x in streaming), flush=True)
        eng.close()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"))


if __name__ == "__main__":
    main()
