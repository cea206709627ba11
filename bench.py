#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: overlap-counted reads/sec, 100M reads x 1M ref intervals (config 3).

One step = one pass of the hot path (gtx_count_device: streaming count kernel + prefix / gather kernels) over the
synthetic 50 bp reads, sorted by (chromosome, start), already resident in HBM, against 1M reference intervals spread
over the 24 hg38 chromosomes, strand ignored (`genomic_overlaps count -S -i`).

N GPUs = one process per GPU, each holding ONE member of a gtx_group (gtx_group_create_rank of libgtx.so: the product's own
multi-GPU code; torch.distributed only hands the communicator id around and provides the barrier).  ONE global read set --
--reads in total (default for N > 1, "strong": the fixed 100M x 1M shape of BASELINE.json; 1 G x 2 M with --reads 1000000000
--refs 2000000) or N x --reads (--scaling weak) -- is apportioned to the chromosomes in proportion to their length; the
chromosomes are dealt to the members by the product's longest-processing-time packing (gtx_group_assign) of the per-chromosome
read counts, so the shares are NOT equal: the line carries `reads_per_rank` and `imbalance` (max / mean), and `value` = all
reads / the slowest rank's time.  Per step every member runs the streaming kernel over its reads, finalizes the histogram
tiles and regions of ITS classes only, and its piece of the compact result vector travels to member 0 over xGMI (grouped
ncclSend / ncclRecv, on a stream of its own under the next step's kernels); member 0 restores file order.

Prints ONE JSON line (rank 0).  `roofline` is for the streaming count kernel of rank 0: algorithmic bytes (12 B per read)
/ its mean duration measured with HIP events on the launch stream inside the timed loop.  `cpu_baseline` is the CPU
oracle (sorted-merge restatement of the reference) on a bounded sample of the same reads, single thread, on this box's
host cores; its counts are also compared with the GPU's (parity check of the sample).  At N = 1 the line also carries
`host_to_result` (packed host arrays -> counts on the host, PCIe included) and `text_to_stdout` (the product CLI from BED
text / a packed region file to its last output line) -- reported beside `value`, never part of it.
"""
import argparse
import json
import types
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "ibm-cbc-genomic-tools_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import gtx  # noqa: E402
from gtx import shard, synth  # noqa: E402

# HIP events around the dominant kernel: on every PROFILE_EVERY-th step of the timed loop (a hipEventRecord leaves ~5 us of
# bubble on the stream here; bracketing every launch stretched a 0.25 ms step by 16 us).  The mean over the sampled
# launches is roofline.kernel_ms; roofline.kernel_samples says how many there were.
PROFILE_EVERY = 4
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
READ_LEN = 50


def make_reads_on_device(n, chrom_ids, seed, device, per=None):
    """n reads of READ_LEN bp on the given chromosomes (proportional to length, or `per` reads on each), sorted by
    (class, start).  The reads of a chromosome depend on (seed, chromosome) only, not on which rank makes them."""
    if per is None:
        per = synth.apportion(n, synth.CHROM_LEN[chrom_ids])
    n = int(np.sum(per))
    out = torch.empty((n, 3), dtype=torch.int32, device=device)
    at = 0
    for ci, cnt in zip(chrom_ids, per):
        cnt = int(cnt)
        if cnt == 0:
            continue
        g = torch.Generator(device=device)
        g.manual_seed(int(seed) * 1000 + int(ci))
        s = torch.randint(1, int(synth.CHROM_LEN[ci]) - READ_LEN - 1, (cnt,), device=device, generator=g, dtype=torch.int32)
        s, _ = torch.sort(s)
        out[at:at + cnt, 0] = int(ci)
        out[at:at + cnt, 1] = s
        out[at:at + cnt, 2] = s + (READ_LEN - 1)
        at += cnt
    return out


# RCCL prints a version banner on the process's stdout when its first communicator comes up; the contract is ONE
# JSON line there.  Multi-rank runs therefore point fd 1 at stderr for their whole life and write the line to the
# saved descriptor.
_REAL_STDOUT = None
DIST_ON = False          # a process group exists (N > 1, or the single-rank self-test GTX_BENCH_FORCE_DIST=1)


def guard_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_REAL_STDOUT, line)


def kernel_source_sha16(files=("gtx_kernels.hip", "gtx_kernels.h", "gtx_capi.hip")):
    """identity of the kernel sources a committed PMC profile belongs to (the GPU box has no .git to ask)"""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(PKG, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def measure_host_to_result(eng, reads, n, hits, flags):
    """(ii) packed host arrays -> counts on the host through gtx_count (host->device copies, kernels, result copy): from
    ordinary pageable memory (staged through page-locked slots by host threads) and from gtx_host_alloc memory (the DMA
    engine reads it directly).  Counts are compared with the device-resident run."""
    eng.count_device(reads.data_ptr(), n, hits.data_ptr(), None, flags)
    eng.sync()
    want = hits.cpu().numpy().view(np.uint64).copy()
    eng.set_stream(0)
    host = reads.cpu().numpy()
    out = {}
    pin = eng.pinned_array(host.shape)
    pin[:] = host
    for name, src in (("pageable", host), ("page_locked", pin)):
        eng.count(src[:1_000_000], None, flags)                                # slots and streams are up
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            got, _ = eng.count(src, None, flags)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        if not np.array_equal(got, want):
            sys.exit("PARITY FAILURE: gtx_count from %s host memory differs from the device-resident run" % name)
        out[name] = {"seconds": best, "reads_per_s": n / best, "GB_per_s": 12.0 * n / best / 1e9}
    eng.free_pinned(pin)
    out["note"] = "gtx_count on %d packed reads in host memory, best of 3, PCIe included; never part of `value`" % n
    return out


def measure_text_to_stdout(n, m, cli_sample=None):
    """(iii) the product CLI `genomic_overlaps count -S -i REFS READS` from BED text and from a packed region file (.gtx) of the
    same reads, wall time of the whole process with stdout going to a file; workload files made by gtx_packtool synth.
    Second result (`text_cli`, the bench line's cpu_baseline.text_cli): the CPU restatement's CLI -- `gtx_oracle count -S -i`, the
    reference's own parser, tokenizer, region objects and sorted merge restated (oracle/gtx_oracle.c follows genomic_overlaps.cpp:408-431
    and what it calls) -- on the SAME BED file (all of it unless `cli_sample` says fewer lines), on one core and as one process per chromosome shard,
    its output byte-equal to the product CLI's on that sample: the nearest thing to "the reference's genomic_overlaps timed beside"
    that can travel to this box (the reference binary cannot: DESIGN.md section 6)."""
    import hashlib
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from oracle import orc
    cli_sample = n if cli_sample is None else cli_sample
    bins = os.path.join(PKG, "csrc")
    tmp = tempfile.mkdtemp(prefix="gtx_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    rp, qp, gp, op, sp = [os.path.join(tmp, f) for f in ("refs.bed", "reads.bed", "reads.gtx", "out.txt", "sample.bed")]
    made = [rp, qp, gp, op, sp]
    res, cli = {}, None
    try:
        subprocess.run([os.path.join(bins, "gtx_packtool"), "synth", str(n), "7", qp], check=True)
        subprocess.run([os.path.join(bins, "gtx_packtool"), "synthrefs", str(m), "8", rp], check=True)
        t0 = time.perf_counter()
        subprocess.run([os.path.join(bins, "gtx_packtool"), "pack", qp, gp], check=True)
        res["pack_seconds"] = time.perf_counter() - t0
        digest = {}
        for name, q in (("bed_text", qp), ("packed_gtx", gp)):
            best = None
            for _ in range(2):
                with open(op, "wb") as f:
                    t0 = time.perf_counter()
                    r = subprocess.run([os.path.join(bins, "genomic_overlaps"), "count", "-S", "-i", rp, q], stdout=f, stderr=subprocess.PIPE)
                    dt = time.perf_counter() - t0
                if r.returncode != 0:
                    sys.exit("text_to_stdout: genomic_overlaps failed: " + r.stderr.decode()[-300:])
                best = dt if best is None else min(best, dt)
            digest[name] = hashlib.md5(open(op, "rb").read()).hexdigest()
            res[name] = {"seconds": best, "reads_per_s": n / best, "input_bytes": os.path.getsize(q)}
        if digest["bed_text"] != digest["packed_gtx"]:
            sys.exit("PARITY FAILURE: the CLI's output differs between BED text and the packed region file")
        res["output_md5"] = digest["bed_text"]
        # the inputs the reference's own examples use (examples/example01.tcsh:15 pipes its reads in; every shipped data file is .gz):
        # the same text through a pipe, a gzip file of its first 20 M lines (inflate is one thread: that is its bound), and
        # genomic_scans counts on the whole file -- all three tokenised on the device like the plain file
        def timed(cmd, stdin_from=None, best_of=2):
            best = None
            for _ in range(best_of):
                with open(op, "wb") as f:
                    t0 = time.perf_counter()
                    if stdin_from:
                        cat = subprocess.Popen(["cat", stdin_from], stdout=subprocess.PIPE)
                        r = subprocess.run(cmd, stdin=cat.stdout, stdout=f, stderr=subprocess.PIPE)
                        cat.stdout.close(); cat.wait()
                    else:
                        r = subprocess.run(cmd, stdout=f, stderr=subprocess.PIPE)
                    dt = time.perf_counter() - t0
                if r.returncode != 0:
                    sys.exit("text_to_stdout: %s failed: %s" % (cmd[0], r.stderr.decode()[-300:]))
                best = dt if best is None else min(best, dt)
            return best, hashlib.md5(open(op, "rb").read()).hexdigest()
        ovl = [os.path.join(bins, "genomic_overlaps"), "count", "-S", "-i", rp]
        dt, md = timed(ovl, stdin_from=qp)
        if md != res["output_md5"]:
            sys.exit("PARITY FAILURE: the CLI's output differs between a file and the same text on stdin")
        t0 = time.perf_counter(); subprocess.run("cat %s | wc -l > /dev/null" % qp, shell=True, check=True); pipe_alone = time.perf_counter() - t0
        res["stdin_pipe"] = {"seconds": dt, "reads_per_s": n / dt, "cat_into_wc_l_alone_seconds": pipe_alone,
                             "note": "cat reads.bed | genomic_overlaps count -S -i refs.bed; bound: the pipe -- what it delivers to any reader that touches the bytes (`cat reads.bed | wc -l` beside it) plus the process's start-up"}
        gz_lines = min(n, 20_000_000)
        gzp, gzs = os.path.join(tmp, "part.bed.gz"), os.path.join(tmp, "part.bed")
        made.extend([gzp, gzs])
        with open(qp, "rb") as f, open(gzs, "wb") as g:
            left = gz_lines
            while left > 0:
                buf = f.read(32 << 20)
                if not buf:
                    break
                k = buf.count(b"\n")
                if k > left:
                    at = -1
                    for _ in range(left):
                        at = buf.index(b"\n", at + 1)
                    buf, k = buf[:at + 1], left
                g.write(buf); left -= k
        subprocess.run("gzip -1 -c %s > %s" % (gzs, gzp), shell=True, check=True)
        dt_plain, md_plain = timed(ovl + [gzs])
        dt, md = timed(ovl + [gzp])
        if md != md_plain:
            sys.exit("PARITY FAILURE: the CLI's output differs between a file and its gzip")
        t0 = time.perf_counter(); subprocess.run("gzip -dc %s > /dev/null" % gzp, shell=True, check=True); inflate = time.perf_counter() - t0
        res["gz"] = {"seconds": dt, "reads_per_s": gz_lines / dt, "lines": gz_lines, "same_lines_uncompressed_seconds": dt_plain,
                     "gzip_dc_alone_seconds": inflate, "gz_bytes": os.path.getsize(gzp),
                     "note": "the first %d lines as .gz (gzip -1); bound: one inflate thread (gzip -dc of the same file alone beside it)" % gz_lines}
        from gtx import synth as _synth
        gnp = os.path.join(tmp, "genome.bed"); made.append(gnp)
        with open(gnp, "w") as f:
            for nm, ln in zip(_synth.CHROM_NAMES, _synth.CHROM_LEN):
                f.write("%s\t0\t%d\n" % (nm, int(ln)))
        scn = [os.path.join(bins, "genomic_scans"), "counts", "-i", "-g", gnp, "-w", "1000", "-d", "1000", "-min", "1"]
        dt, md = timed(scn + [qp])
        dt_g, md_g = timed(scn + [gp])
        if md != md_g:
            sys.exit("PARITY FAILURE: genomic_scans' output differs between BED text and the packed region file")
        res["scans_counts"] = {"seconds": dt, "reads_per_s": n / dt, "packed_gtx_seconds": dt_g, "output_md5": md,
                               "note": "genomic_scans counts -i -w 1000 -d 1000 -min 1 on the same %d reads (BASELINE config 4's geometry), from BED text and from .gtx" % n}
        res["note"] = ("genomic_overlaps count -S -i, %d reads x %d regions, process wall time to exit with stdout to a file, best of 2; "
                       "host cores %d; never part of `value`" % (n, m, os.cpu_count()))

        # ---- the CPU restatement's CLI on the same text (all of it by default: the restatement parses ~13 M lines/s on one core) ----
        if cli_sample > 0 and os.path.exists(orc.CLI):
            import mmap
            ns = min(cli_sample, n)
            with open(qp, "rb") as f:
                mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
                size = len(mm)

                def line_start(off):                                          # first line that begins at or after byte `off`
                    if off <= 0:
                        return 0
                    k = mm.find(b"\n", off - 1)
                    return size if k < 0 else k + 1

                def name_at(off):
                    return mm[off:mm.find(b"\t", off)]

                end = size
                if ns < n:                                                    # a prefix of ns lines (the file is in (chromosome, start) order)
                    at, left = -1, ns
                    while left > 0:
                        at = mm.find(b"\n", at + 1); left -= 1
                    end = at + 1
                    with open(sp, "wb") as g:
                        g.write(mm[:end])
                    sample_path = sp
                else:
                    sample_path = qp
                with open(op, "wb") as g:
                    r = subprocess.run([os.path.join(bins, "genomic_overlaps"), "count", "-S", "-i", rp, sample_path], stdout=g, stderr=subprocess.PIPE)
                if r.returncode != 0:
                    sys.exit("text_cli: genomic_overlaps failed on the sample: " + r.stderr.decode()[-300:])
                want = open(op, "rb").read()
                t0 = time.perf_counter()
                r = subprocess.run([orc.CLI, "count", "-S", "-i", rp, sample_path], capture_output=True)
                one_s = time.perf_counter() - t0
                if r.returncode != 0 or r.stdout != want:
                    sys.exit("PARITY FAILURE: the CPU restatement's CLI and the product CLI disagree on the BED text (%d lines)" % ns)
                cli = {"value": ns / one_s, "unit": "reads/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port", "seconds": one_s,
                       "sample": "gtx_oracle count -S -i on %s of the BED text the text_to_stdout leg uses (%d regions), process wall time, "
                                 "text parsing included; output md5 %s = the product CLI's on the same files"
                                 % ("all %d lines" % ns if ns == n else "the first %d lines" % ns, m, hashlib.md5(want).hexdigest())}
                # one process per chromosome shard (the reference's own sharding axis): the lines and the regions of a chromosome to a
                # process each.  Both files are in strcmp order of the chromosome names, so a shard is a byte range (found by bisection
                # on the line starts) and the shards' outputs, in the regions file's order, are the whole output.
                names = []
                with open(rp, "rb") as rf:
                    rdata = rf.read()
                pos = 0
                while pos < len(rdata):
                    c = rdata[pos:rdata.index(b"\t", pos)]
                    names.append(c)
                    lo, hi = pos, len(rdata)                                  # first line of another chromosome behind pos
                    while lo < hi:
                        mid = (lo + hi) // 2
                        k = rdata.find(b"\n", mid)
                        ls = len(rdata) if k < 0 else k + 1
                        if ls >= len(rdata) or rdata[ls:rdata.index(b"\t", ls)] != c:
                            hi = mid
                        else:
                            lo = mid + 1
                    k = rdata.find(b"\n", lo)
                    nxt = len(rdata) if k < 0 else k + 1
                    fn = os.path.join(tmp, "r_%s.bed" % c.decode()); made.append(fn)
                    with open(fn, "wb") as g:
                        g.write(rdata[pos:nxt])
                    pos = nxt
                del rdata

                def first_not_before(c):                                      # offset of the first line (< end) whose chromosome is >= c
                    lo, hi = 0, end
                    while lo < hi:
                        mid = (lo + hi) // 2
                        ls = min(line_start(mid), end)
                        if ls >= end or name_at(ls) >= c:
                            hi = mid
                        else:
                            lo = mid + 1
                    return min(line_start(lo), end)

                cuts = [first_not_before(c) for c in names] + [end]
                for i, c in enumerate(names):
                    fn = os.path.join(tmp, "q_%s.bed" % c.decode()); made.append(fn)
                    lo, hi = cuts[i], max(cuts[i + 1], cuts[i])
                    with open(fn, "wb") as g:
                        if hi > lo and name_at(lo) == c:
                            g.write(mm[lo:hi])
                mm.close()

            def one(c):
                return subprocess.run([orc.CLI, "count", "-S", "-i", os.path.join(tmp, "r_%s.bed" % c.decode()), os.path.join(tmp, "q_%s.bed" % c.decode())],
                                      capture_output=True)
            workers = max(1, min(len(names), os.cpu_count() or 1))
            t0 = time.perf_counter()
            with ThreadPoolExecutor(workers) as ex:
                outs = list(ex.map(one, names))
            all_s = time.perf_counter() - t0
            if any(o.returncode != 0 for o in outs) or b"".join(o.stdout for o in outs) != want:
                sys.exit("PARITY FAILURE: the sharded run of the CPU restatement's CLI disagrees with the product CLI")
            cli["all_cores"] = {"value": ns / all_s, "unit": "reads/s", "cores": workers, "host_cores": os.cpu_count(), "seconds": all_s,
                                "note": "one gtx_oracle process per chromosome shard (%d shards: its lines of the text, its regions), outputs "
                                        "concatenated in the regions file's chromosome order = the same bytes" % len(names)}
    finally:
        for f in set(made):
            if os.path.exists(f):
                os.remove(f)
        os.rmdir(tmp)
    return res, cli


def bench_scans(args, eng, reads, n, rank, world, device, rehearse, total_reads, reads_per_rank, grp=None, locals_=(), per_chrom=None, owner=None):
    """BASELINE config 4: sliding-window read counts (1 kb windows).  N > 1: the product's gtx_group_scan_device -- every member
    scans the chromosomes it owns into a packed vector of its own and the per-chromosome pieces travel to member 0 (grouped
    ncclSend / ncclRecv); `grp` is this process's view of the group (None: a rehearsal rank that only keeps the barriers company)."""
    step_bp, size_bp = 1000, 1000
    off, tot = gtx.scan_layout(synth.CHROM_LEN, step_bp, size_bp)
    out = torch.zeros(tot, dtype=torch.int64, device=device)
    if DIST_ON:
        mreads = []
        if grp is not None:
            grp.assign(per_chrom)
            for m in locals_:
                if not rehearse:
                    grp.set_stream(m, torch.cuda.current_stream().cuda_stream)
                ch = np.nonzero(owner == m)[0].astype(np.int64)
                mreads.append(make_reads_on_device(0, ch, 1000, device, per=per_chrom[ch]))
            eng = types.SimpleNamespace(profile=lambda on: grp.profile(locals_[0], on), profile_last=lambda b: grp.profile_last(locals_[0], b),
                                        profiled_calls=lambda: grp.profiled_calls(locals_[0]), sync=grp.sync)
            n = mreads[0].shape[0]
        else:
            eng = types.SimpleNamespace(profile=lambda on: None, profile_last=lambda b: (float("nan"), 0.0), profiled_calls=lambda: 0, sync=lambda: None)
        ptrs, ns = [r.data_ptr() for r in mreads], [r.shape[0] for r in mreads]
    else:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)

    def step():
        if not DIST_ON:
            eng.scan_device(reads.data_ptr(), n, synth.CHROM_LEN, step_bp, size_bp, out.data_ptr(), flags=gtx.READS_SORTED)
        elif grp is not None:
            grp.scan_device(ptrs, ns, synth.CHROM_LEN, step_bp, size_bp, out.data_ptr(), flags=gtx.READS_SORTED)

    def fence():
        if DIST_ON:
            eng.sync()
        torch.cuda.synchronize()
        if DIST_ON:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.profile(PROFILE_EVERY if args.steps >= PROFILE_EVERY else 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    k_ms = float(np.mean([eng.profile_last(b)[0] for b in range(eng.profiled_calls())])) if eng.profiled_calls() else float("nan")
    if DIST_ON:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if os.environ.get("GTX_BENCH_VERIFY") == "1":
            from oracle import orc
            mine = np.zeros(tot, dtype=np.int64)
            for r in (mreads if grp is not None else []):
                mine += orc.scan(r.cpu().numpy(), synth.CHROM_LEN, step_bp, size_bp, algo=1)[0].view(np.int64)
            want = torch.from_numpy(mine) if rehearse else torch.from_numpy(mine).to(device)
            dist.all_reduce(want, op=dist.ReduceOp.SUM)
            if rank == 0 and not torch.equal(out.cpu(), want.cpu()):
                sys.exit("PARITY FAILURE: the group's windows on member 0 differ from the oracle's scan of all members' reads")
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import orc
        ns = min(args.cpu_sample, n)
        sample = reads[:ns].cpu().numpy()
        t1 = time.perf_counter()
        want, _ = orc.scan(sample, synth.CHROM_LEN, step_bp, size_bp, algo=1)
        cpu_s = time.perf_counter() - t1
        eng.scan_device(reads.data_ptr(), ns, synth.CHROM_LEN, step_bp, size_bp, out.data_ptr(), flags=gtx.READS_SORTED)
        eng.sync()
        if not np.array_equal(out.cpu().numpy().view(np.uint64), want):
            sys.exit("PARITY FAILURE: GPU window counts differ from the CPU oracle on the %d-read sample" % ns)
        cpu = {"value": ns / cpu_s, "unit": "reads/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
               "sample": "first %d reads, sorted-scanner restatement (oracle/gtx_oracle.c) on packed triples; windows bit-equal to the GPU's" % ns}
    if rank == 0:
        alg = 12.0 * n + 8.0 * tot                                          # the reads once + the 8-byte window sums once
        emit(({
            "metric": "window-counted reads/sec, genomic_scans counts 1 kb windows (BASELINE config 4)",
            "value": total_reads * args.steps / elapsed, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "BASELINE config 4: genomic_scans counts -i -w 1000 -d 1000 over %d 50bp reads in total, hg38 chromosome "
                                   "shards (LPT, gtx_group), the owners' per-chromosome pieces of the %d-window vector to member 0 (ncclSend/ncclRecv)" % (total_reads, tot), "total_reads": total_reads,
                       "reads_per_rank": reads_per_rank, "imbalance": max(reads_per_rank) / (sum(reads_per_rank) / world), "windows": tot},
            "roofline": {"bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "scan_hist_kernel", "kernel_ms": k_ms,
                         "algorithmic_bytes": alg},
            "cpu_baseline": cpu}))
    if DIST_ON:
        dist.destroy_process_group()


def bench_perm(args, rank, world, local, device, rehearse):
    """BASELINE config 5, second half: permutation_test with 10k shuffles of a GO-like table (20k rows, 5k
    categories, ~1M memberships) per GPU; ranks take disjoint permutation ranges, exceed-counts are all-reduced."""
    from gtx import perm
    n_rows, n_cols, mean_size, P = 20000, 5000, 200, args.shuffles
    t = perm.PermTable.synthetic(n_rows, n_cols, mean_size, seed=1, values="gamma")
    nnz = int(t.col_ptr[-1])
    e = perm.PermEngine(local)
    e.set_table(t)
    Y = e.statistic("sum")
    acc = torch.zeros(n_cols, dtype=torch.int64, device="cpu" if rehearse else device)
    ms = []

    def step():
        c = e.count_ge("sum", Y, 2024, rank * P, P)                      # this rank's shuffles: [rank*P, (rank+1)*P)
        ms.append(e.last_ms())
        if DIST_ON:
            acc.copy_(torch.from_numpy(c.view(np.int64)))
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        return c

    def fence():
        torch.cuda.synchronize()
        if DIST_ON:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    del ms[:]
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        counts = step()
    fence()
    elapsed = time.perf_counter() - t0
    if DIST_ON:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    apply_ms = float(np.mean([m[0] for m in ms])); stat_ms = float(np.mean([m[1] for m in ms]))
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import porc
        ps = 24                                                            # ~10-20 s of the scalar restatement
        t1 = time.perf_counter()
        want = porc.count_ge(t, "sum", Y, 2024, 0, ps)
        cpu_s = time.perf_counter() - t1
        if not np.array_equal(e.count_ge("sum", Y, 2024, 0, ps), want) or not np.array_equal(Y.view(np.uint64), porc.statistic(t, "sum").view(np.uint64)):
            sys.exit("PARITY FAILURE: GPU exceed-counts differ from the CPU oracle on the %d-shuffle sample" % ps)
        cpu = {"value": nnz * ps / cpu_s, "unit": "member-sums/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
               "sample": "first %d shuffles of the same table, oracle/perm_oracle.c (permute + per-category sums + compare); "
                         "counts bit-equal to the GPU's" % ps}
    traffic, l2_hit = None, None                                          # L2->fabric bytes per batch and L2 hit rate from the committed --pmc passes
    if P == 10000:                                                        # of this command -- only if they were taken from the sources that run here
        import glob
        cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_perm_stat.json")))
        if cand:
            prof = json.load(open(cand[-1]))
            if prof.get("perm_source_sha16") == kernel_source_sha16(("gtx_perm.hip",)):
                traffic, l2_hit = prof.get("traffic_bytes_per_launch"), prof.get("l2_hit_rate")
    if rank == 0:
        alg = 4.0 * nnz * P                                               # gathered: one 4-byte slab element per (membership, shuffle)
        unique = 4.0 * n_rows * P + 4.0 * nnz + 8.0 * n_cols              # slab read once + membership lists + offsets
        emit(({
            "metric": "category-member sums/sec, permutation_test -S sum, 10k shuffles (BASELINE config 5)",
            "value": world * nnz * P * args.steps / elapsed, "unit": "member-sums/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 5 (permutation_test part): %d shuffles/GPU of a %d-row x %d-category table with %d "
                                   "memberships, statistic sum, all-reduce(sum) of the exceed-counts" % (P, n_rows, n_cols, nnz),
                       "shuffles_per_gpu": P, "rows": n_rows, "categories": n_cols, "memberships": nnz},
            # The dominant kernel is a row gather with heavy reuse: every slab element is read ~50 times (once per category
            # that holds its row).  Against HBM the algorithmic bytes are the slab + the membership lists, read once; the
            # kernel is bound by the L2 -> CU gather rate instead, priced in "l2_gather" against MI355X_MICROARCH.md's
            # measured L2 row-gather rate (16.8-18.8 TB/s chip-wide).
            "roofline": {"bound": "hbm", "achieved": unique / (stat_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": unique / (stat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "kernel": "perm_stat_kernel",
                         "kernel_ms": stat_ms, "apply_kernel_ms": apply_ms, "algorithmic_bytes": unique,
                         "l2_gather": {"gathered_bytes": alg, "achieved": alg / (stat_ms * 1e-3) / 1e9, "peak": 18800.0, "unit": "GB/s",
                                       "frac": alg / (stat_ms * 1e-3) / 1e9 / 18800.0},
                         "l2_hit_rate": l2_hit,
                         "note": "not HBM-bound: 4 B x memberships x shuffles of 256-byte row gathers are served by the XCD's L2; "
                                 "traffic = L2->fabric bytes per 10 k-shuffle batch and l2_hit_rate from the committed PMC passes (null when "
                                 "they were not taken from these sources), includes Infinity-Cache hits"},
            "cpu_baseline": cpu}))
    e.close()
    if DIST_ON:
        dist.destroy_process_group()


def shape_name(total_reads, n_refs):
    if total_reads == 100_000_000 and n_refs == 1_000_000:
        return "BASELINE config 3"
    if total_reads == 1_000_000_000 and n_refs == 2_000_000:
        return "BASELINE config 5 shape (count part)"
    return "BASELINE config 3 workload at another size"


def make_group(rank, world, local, device, rehearse, force_dist):
    """this process's view of the gtx_group of a multi-GPU run: (group or None, the members it drives)"""
    if rehearse:                                                      # one GPU, gloo: rank 0 drives all members on device 0
        if rank != 0:
            return None, []
        os.environ["GTX_GROUP_REHEARSE"] = "1"
        return gtx.Group([0] * world), list(range(world))
    if force_dist:
        os.environ["GTX_GROUP_SELF_EXCHANGE"] = "1"                  # the piece of the only member goes out and comes back through RCCL
    uid = torch.zeros(gtx.GROUP_ID_BYTES, dtype=torch.uint8, device=device)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(gtx.Group.unique_id()), dtype=torch.uint8))
    dist.broadcast(uid, src=0)
    return gtx.Group(rank=rank, world=world, device=local, unique_id=bytes(uid.cpu().numpy().tobytes())), [rank]


def bench_count_group(args, rank, world, local, device, rehearse, force_dist):
    """N > 1 (and the single-rank self-test GTX_BENCH_FORCE_DIST=1): every process holds one member of a gtx_group and the product's
    own multi-GPU step runs -- gtx_group_count_device.  GTX_BENCH_REHEARSE=1 (one GPU, gloo): rank 0 drives a rehearsal group of
    `world` members on device 0 through the same calls, the other ranks only keep the barriers company."""
    refs = synth.genome_intervals(args.refs, 43, 50, 2000)
    n_members = world
    total_reads = args.reads * (world if args.scaling == "weak" else 1)
    per_chrom = synth.apportion(total_reads, synth.CHROM_LEN)
    flags = gtx.READS_SORTED
    owner = gtx.lpt_assign(per_chrom, n_members)
    reads_per_rank = [int(per_chrom[owner == r].sum()) for r in range(n_members)]
    kernel_ms, elapsed, hits_final, member_reads = [], 0.0, None, None
    grp, locals_, reads, problem = None, [], [], ""
    step_no = [0]

    def step():
        if grp is not None:
            b = step_no[0] & 1
            step_no[0] += 1
            grp.count_device(ptrs, ns, hits_pp[b].data_ptr(), flags=flags)

    def fence():
        if grp is not None:
            grp.sync()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    # Everything up to the first result is guarded: if the group cannot be made on this machine (librccl not found, the
    # communicator refused) or its first result is not the sum of the ranks' own counts, EVERY rank leaves this function and the
    # run goes on with the plain path below (one context per rank + torch.distributed reduce), which says so in its line --
    # a scaling run then still ends with a valid number, never with a silently different one.
    try:
        if os.environ.get("GTX_BENCH_BREAK_GROUP") == "1":          # (a test of the way out below)
            raise RuntimeError("GTX_BENCH_BREAK_GROUP=1")
        grp, locals_ = make_group(rank, world, local, device, rehearse, force_dist)
        if grp is not None:
            assert np.array_equal(grp.assign(per_chrom), owner)
            grp.set_refs(refs, synth.n_classes())
            stream = torch.cuda.current_stream()
            for m in locals_:
                grp.set_stream(m, stream.cuda_stream) if not rehearse else None
                ch = np.nonzero(owner == m)[0].astype(np.int64)
                reads.append(make_reads_on_device(0, ch, 1000, device, per=per_chrom[ch]))
                assert reads[-1].shape[0] == reads_per_rank[m]
            ptrs, ns = [r.data_ptr() for r in reads], [r.shape[0] for r in reads]
            hits_pp = [torch.zeros(len(refs), dtype=torch.int64, device=device) for _ in range(2)]
        for _ in range(max(args.warmup, 2)):
            step()
    except Exception as e:                                            # noqa: BLE001 -- whatever it is, the other ranks must hear of it
        problem = "%s: %s" % (type(e).__name__, e)
    if not agree(not problem, rehearse, device):
        if grp is not None:
            grp.close()
        return "group unavailable" + (": " + problem if problem else " on another rank")
    fence()
    # always on: the group's result on member 0 against the sum over the ranks of what ONE context counts for the rank's own reads
    # (gtx_count_device, the N = 1 path the parity tests pin) -- the exchange checked end to end at full size on the real machine
    own = torch.zeros(len(refs), dtype=torch.int64, device=device)
    if grp is not None:
        eng1 = gtx.Engine(local)
        eng1.set_refs(refs, synth.n_classes())
        one = torch.zeros_like(own)
        for r in reads:
            eng1.count_device(r.data_ptr(), r.shape[0], one.data_ptr(), None, flags)
            eng1.sync()
            own += one
        eng1.close()
    if rehearse:
        own_c = own.cpu(); dist.all_reduce(own_c, op=dist.ReduceOp.SUM); own = own_c.to(device)
    else:
        dist.all_reduce(own, op=dist.ReduceOp.SUM)
    same = True
    if rank == 0:
        same = bool(torch.equal(hits_pp[(step_no[0] - 1) & 1], own))
    if not agree(same, rehearse, device):
        if grp is not None:
            grp.close()
        return "the group's result on member 0 differed from the sum of the ranks' own counts"
    if grp is not None:
        grp.profile(locals_[0], PROFILE_EVERY if args.steps >= PROFILE_EVERY else 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if grp is not None:
        kernel_ms = [grp.profile_last(locals_[0], b)[0] for b in range(grp.profiled_calls(locals_[0]))]
        grp.profile(locals_[0], False)
        hits_final = hits_pp[(step_no[0] - 1) & 1]
        member_reads = [int(x) for x in grp.member_reads()]
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if os.environ.get("GTX_BENCH_VERIFY") == "1":
        # the result on member 0 must be the one-process count of all members' reads (CPU oracle, summed over the ranks)
        from oracle import orc
        mine = np.zeros(len(refs), dtype=np.int64)
        if grp is not None:
            for r in reads:
                mine += orc.count(refs, r.cpu().numpy(), algo=orc.SORTED_MERGE).view(np.int64)
        tot = torch.from_numpy(mine) if rehearse else torch.from_numpy(mine).to(device)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        if rank == 0 and not torch.equal(hits_final.cpu(), tot.cpu()):
            sys.exit("PARITY FAILURE: the group's result on member 0 differs from the oracle's count of all members' reads")
    if rank == 0:
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else None
        n0 = reads_per_rank[0]
        achieved = 12.0 * n0 / (k_ms * 1e-3) / 1e9 if k_ms else None
        emit({
            "metric": "overlap-counted reads/sec, 100M reads x 1M ref intervals",
            "value": total_reads * args.steps / elapsed, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: %d 50bp reads in total, sorted by (chrom,start), x %d ref intervals over 24 hg38 "
                                   "chromosomes, strand ignored (genomic_overlaps count -S -i); reads resident in HBM" % (shape_name(total_reads, len(refs)), total_reads, len(refs)),
                       "total_reads": total_reads, "refs": len(refs), "reads_per_rank": reads_per_rank,
                       "imbalance": max(reads_per_rank) / (sum(reads_per_rank) / n_members),
                       "parallelism": "%s scaling through the product's gtx_group (libgtx.so): one process per GPU = one member each "
                                      "(gtx_group_create_rank), chromosomes dealt to the %d members by LPT packing of their read counts, reference "
                                      "set replicated; per step gtx_group_count_device: streaming kernel + finalize of the member's own classes on one of "
                                      "three streams in turn (the kernels of successive steps are not ordered behind each other), the member's regions to member 0" % (args.scaling, n_members),
                       "reduce": ("one-GPU rehearsal of the group code (not a measurement)" if rehearse else
                                  "grouped ncclSend/ncclRecv of each member's runs of the uint64 count vector (a run per chromosome: the reference file is in "
                                  "chromosome order) straight to their places in member 0's result vector over xGMI, on a stream of its own under the "
                                  "next steps' kernels; member 0's own regions are written there by its finalize step") +
                                 (" [single-rank self-test: the piece goes out and back through RCCL]" if force_dist else ""),
                       "verified": "before the timed steps: result on member 0 == sum over the ranks of gtx_count_device on each rank's own reads"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": None,
                         "kernel": "count_walk_kernel (member 0's launch over its %d reads; timed with the kernels of the neighbouring steps running "
                                   "beside it on the other two streams, i.e. sharing the chip's bandwidth: the per-step time is what counts here, "
                                   "the kernel's own roofline is the N = 1 line's)" % n0, "kernel_ms": k_ms,
                         "kernel_samples": len(kernel_ms), "algorithmic_bytes": 12.0 * n0},
            "cpu_baseline": None,
        })
    if grp is not None:
        grp.close()
    dist.destroy_process_group()
    return None


def agree(ok, rehearse, device):
    """True when `ok` holds on every rank"""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if rehearse else device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads in total (strong scaling, the default of the count workload) or per GPU (--scaling weak)")
    ap.add_argument("--refs", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample", type=int, default=100_000_000, help="reads given to the CPU baseline (0 = skip)")
    ap.add_argument("--workload", choices=["count", "scans", "permutation_test"], default="count",
                    help="count = BASELINE config 3 (the headline metric); scans = config 4: genomic_scans counts -i -w 1000 -d 1000; "
                         "permutation_test = the shuffle part of config 5")
    ap.add_argument("--shuffles", type=int, default=10000, help="permutation_test: shuffles per GPU per step")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="N > 1: strong = --reads in total (default for the count workload: the fixed shape BASELINE.json names), weak = N x "
                         "--reads in total; either way ONE global read set, sharded by chromosome with the product's LPT packing")
    ap.add_argument("--no-e2e", action="store_true", help="N = 1: skip the host_to_result / text_to_stdout measurements")
    ap.add_argument("--two-streams", action="store_true",
                    help="count, N=1: after the timed region also time the same steps alternating two contexts on two HIP streams "
                         "(extra 'two_streams' object; off by default so that a kernel trace of the default command sees only the timed steps)")
    args = ap.parse_args()

    if args.scaling is None:
        args.scaling = "strong" if args.workload == "count" else "weak"
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    # GTX_BENCH_REHEARSE=1: all ranks share GPU 0 and reduce through gloo -- only to exercise the N>1 code
    # path on a one-GPU box (RCCL refuses two ranks on one device); never a reported configuration
    rehearse = os.environ.get("GTX_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # GTX_BENCH_FORCE_DIST=1: run the N>1 code (process group, pipelined RCCL all-reduce) with a single rank -- a
    # self-test of that path on a one-GPU box, never a reported configuration
    force_dist = world == 1 and os.environ.get("GTX_BENCH_FORCE_DIST") == "1"
    global DIST_ON
    DIST_ON = world > 1 or force_dist
    if DIST_ON:
        guard_stdout()
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    if args.workload == "permutation_test":
        return bench_perm(args, rank, world, local, device, rehearse)
    group_note = None
    if args.workload == "count" and DIST_ON:
        group_note = bench_count_group(args, rank, world, local, device, rehearse, force_dist)
        if group_note is None:
            return
        print("bench.py: %s -- falling back to one context per rank + torch.distributed reduce of the full vectors" % group_note, file=sys.stderr)

    # ---- workload -------------------------------------------------------------------------------
    refs = synth.genome_intervals(args.refs, 43, 50, 2000)                 # 1M refs, all chromosomes, every rank
    # one global read set, apportioned to the chromosomes by length; chromosomes -> ranks by the product's LPT packing of the
    # per-chromosome read counts (gtx_lpt_assign, the function genomic_overlaps --ngpu uses)
    total_reads = args.reads * (world if args.scaling == "weak" else 1)
    per_chrom = synth.apportion(total_reads, synth.CHROM_LEN)
    owner = gtx.lpt_assign(per_chrom, world)
    reads_per_rank = [int(per_chrom[owner == r].sum()) for r in range(world)]
    if args.workload == "scans" and DIST_ON:
        grp, locals_ = make_group(rank, world, local, device, rehearse, force_dist)
        bench_scans(args, None, None, 0, rank, world, device, rehearse, total_reads, reads_per_rank, grp, locals_, per_chrom, owner)
        if grp is not None:
            grp.close()
        return
    my_chroms = np.nonzero(owner == rank)[0].astype(np.int64)
    reads = make_reads_on_device(0, my_chroms, 1000, device, per=per_chrom[my_chroms])
    n = reads.shape[0]
    assert n == reads_per_rank[rank]
    hits = torch.zeros(len(refs), dtype=torch.int64, device=device)        # uint64 bit pattern; int64 for RCCL sum
    # two count vectors in ping-pong: the all-reduce of step i (RCCL's own stream) overlaps the kernels of step i+1
    hits_pp = [hits, torch.zeros_like(hits)]
    pending = [None, None]
    pipelined = (world > 1 or force_dist) and os.environ.get("GTX_BENCH_SYNC_REDUCE") != "1"
    use_allreduce = os.environ.get("GTX_BENCH_ALLREDUCE") == "1"
    step_no = [0]

    eng = gtx.Engine(local)
    if args.workload == "scans":
        return bench_scans(args, eng, reads, n, rank, world, device, rehearse, total_reads, reads_per_rank)
    eng.set_refs(refs, synth.n_classes())
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    flags = gtx.READS_SORTED

    def step():
        if pipelined and not rehearse:
            b = step_no[0] & 1
            step_no[0] += 1
            if pending[b] is not None:
                pending[b].wait()                                          # stream-side wait: buffer b is free again
            eng.count_device(reads.data_ptr(), n, hits_pp[b].data_ptr(), None, flags)
            # north_star: "an RCCL reduce over xGMI of the per-region count vector" -- rank 0 (the one that would print) gets the
            # sum; a reduce moves half the bytes of an all-reduce over the ring (GTX_BENCH_ALLREDUCE=1: every rank gets it)
            if use_allreduce:
                pending[b] = dist.all_reduce(hits_pp[b], op=dist.ReduceOp.SUM, async_op=True)
            else:
                pending[b] = dist.reduce(hits_pp[b], dst=0, op=dist.ReduceOp.SUM, async_op=True)
            return
        eng.count_device(reads.data_ptr(), n, hits.data_ptr(), None, flags)
        if world > 1:
            if rehearse:
                h = hits.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                hits.copy_(h)
            else:
                shard.reduce_counts(hits, dist, device_tensor=True)        # RCCL all-reduce over xGMI of the count vector

    def fence():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # (no fallback: if the pipelined reduce fails the run fails -- a silently different reduce path would still print a number)
    for _ in range(max(args.warmup, 2 if pipelined else 0)):
        step()
    fence()
    eng.profile(PROFILE_EVERY if args.steps >= PROFILE_EVERY else 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    k_ms = [eng.profile_last(b)[0] for b in range(eng.profiled_calls())]
    eng.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # multi-rank parity: the reduced vector must be the single-process count of all ranks' reads
        if os.environ.get("GTX_BENCH_VERIFY") == "1":
            from oracle import orc
            mine = orc.count(refs, reads.cpu().numpy(), algo=orc.SORTED_MERGE).view(np.int64)
            tot = torch.from_numpy(mine.copy()) if rehearse else torch.from_numpy(mine.copy()).to(device)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            pipelined = False
            step()
            torch.cuda.synchronize()
            if not torch.equal(hits.cpu(), tot.cpu()):
                sys.exit("PARITY FAILURE: reduced multi-rank counts differ from the oracle's")

    # HBM traffic of the dominant kernel: bench.py cannot collect PMC counters itself; the figure is the
    # committed rocprofv3 --pmc pass of this same command (scripts/pmc.sh -> profiles/rNN_pmc_count_walk.json),
    # FETCH_SIZE corrected x2 as MI355X_MICROARCH.md prescribes for gfx950, valid for the default workload only
    traffic = None
    if n == 100_000_000 and args.refs == 1_000_000:
        import glob
        cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_count_walk.json")))
        if cand:
            pmc = json.load(open(cand[-1]))
            if pmc.get("kernel_source_sha16") == kernel_source_sha16():    # counters of THIS kernel source, else unknown
                traffic = pmc.get("traffic_bytes_per_launch")

    kernel_ms = float(np.mean(k_ms))
    alg_bytes = 12.0 * n                                                    # the triples the kernel must read once
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    value = total_reads * args.steps / elapsed

    # ---- CPU baseline + parity on the sample (rank 0, N=1 only) -----------------------------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import orc
        ns = min(args.cpu_sample, n)
        sample = reads[:ns].cpu().numpy()
        t1 = time.perf_counter()
        want = orc.count(refs, sample, algo=orc.SORTED_MERGE)
        cpu_s = time.perf_counter() - t1
        eng.count_device(reads.data_ptr(), ns, hits.data_ptr(), None, flags)
        eng.sync()
        got = hits.cpu().numpy().view(np.uint64)
        if not np.array_equal(got, want):
            sys.exit("PARITY FAILURE: GPU counts differ from the CPU oracle on the %d-read sample" % ns)
        cpu = {"value": ns / cpu_s, "unit": "reads/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
               "sample": "first %d reads of the same workload vs all %d refs, packed triples in memory, "
                         "sorted-merge restatement (oracle/gtx_oracle.c); counts bit-equal to the GPU's" % (ns, len(refs))}

        # the same restatement on all host cores the work divides over: one thread per chromosome (its reads, its regions), the
        # reference's sharding axis; ctypes releases the GIL, the oracle keeps no shared state on this path
        if os.environ.get("GTX_BENCH_CPU_ALL", "1") != "0":
            from concurrent.futures import ThreadPoolExecutor
            shards = []
            for c in range(synth.n_classes()):
                rs = np.nonzero(refs[:, 0] == c)[0]
                lo, hi = np.searchsorted(sample[:, 0], [c, c + 1])
                if len(rs) and hi > lo:
                    shards.append((rs, np.ascontiguousarray(refs[rs]), sample[lo:hi]))
            workers = max(1, min(len(shards), os.cpu_count() or 1))
            t2 = time.perf_counter()
            with ThreadPoolExecutor(workers) as ex:
                parts = list(ex.map(lambda sh: orc.count(sh[1], sh[2], algo=orc.SORTED_MERGE), shards))
            all_s = time.perf_counter() - t2
            merged = np.zeros(len(refs), dtype=np.uint64)
            for (rs, _, _), h in zip(shards, parts):
                merged[rs] = h
            if not np.array_equal(merged, want):
                sys.exit("PARITY FAILURE: the sharded CPU baseline disagrees with the single-thread one")
            cpu["all_cores"] = {"value": ns / all_s, "unit": "reads/s", "cores": workers, "host_cores": os.cpu_count(),
                                "note": "same restatement, one thread per chromosome shard (%d shards)" % len(shards)}

    # ---- not part of `value`: the same steps issued alternately through two contexts on two HIP streams, so that the
    # finalize launches of one step run under the streaming kernel of the next (N=1 only; reported for information,
    # the per-kernel roofline above is measured without this overlap)
    two_streams = None
    if rank == 0 and world == 1 and args.two_streams:
        s2 = torch.cuda.Stream()
        eng2 = gtx.Engine(local)
        eng2.set_refs(refs, synth.n_classes())
        eng2.set_stream(s2.cuda_stream)
        hits2 = torch.zeros_like(hits)
        pair = ((eng, hits), (eng2, hits2))
        for i in range(4):
            pair[i & 1][0].count_device(reads.data_ptr(), n, pair[i & 1][1].data_ptr(), None, flags)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(args.steps):
            pair[i & 1][0].count_device(reads.data_ptr(), n, pair[i & 1][1].data_ptr(), None, flags)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t2
        if not torch.equal(hits, hits2):
            sys.exit("PARITY FAILURE: the two contexts disagree")
        two_streams = {"value": n * args.steps / e2, "unit": "reads/s", "ms_per_step": e2 / args.steps * 1e3,
                       "note": "two contexts alternating on two HIP streams; informational, not the reported value"}
        eng2.close()

    # ---- not part of `value`: the same job timed from the host side (SURVEY 8(d) ii, iii), N = 1 only ----------------------
    host_to_result = text_to_stdout = None
    if rank == 0 and world == 1 and not args.no_e2e:
        host_to_result = measure_host_to_result(eng, reads, n, hits, flags)
        text_to_stdout, text_cli = measure_text_to_stdout(n, len(refs))
        if cpu is not None and text_cli is not None:
            cpu["text_cli"] = text_cli

    if rank == 0:
        line = {
            "metric": "overlap-counted reads/sec, 100M reads x 1M ref intervals",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: %d 50bp reads in total, sorted by (chrom,start), x %d ref intervals over 24 hg38 "
                                   "chromosomes, strand ignored (genomic_overlaps count -S -i); reads resident in HBM" % (shape_name(total_reads, len(refs)), total_reads, len(refs)),
                       "total_reads": total_reads, "refs": len(refs), "reads_per_rank": reads_per_rank,
                       "imbalance": max(reads_per_rank) / (sum(reads_per_rank) / world),
                       "parallelism": "%s scaling: one global read set, chromosomes dealt to %d rank(s) by LPT packing of their read counts "
                                      "(gtx_lpt_assign), reference set replicated" % (args.scaling if world > 1 else "single GPU;", world) +
                                      (" [FALLBACK: %s; every rank finalizes and reduces the full vector]" % group_note if group_note else ""),
                       "reduce": ("none (one rank)" if not DIST_ON else "gloo rehearsal on one GPU (not a measurement)" if rehearse else
                                  ("RCCL reduce(sum) to rank 0" if pipelined and not use_allreduce else "RCCL all-reduce(sum)") +
                                  (" of the uint64 count vector per step, enqueued on RCCL's stream under the next step's kernels" if pipelined
                                   else " of the uint64 count vector per step, synchronous"))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "count_walk_kernel", "kernel_ms": kernel_ms, "kernel_samples": len(k_ms), "algorithmic_bytes": alg_bytes},
            "cpu_baseline": cpu,
        }
        if two_streams:
            line["two_streams"] = two_streams
        if host_to_result:
            line["host_to_result"] = host_to_result
        if text_to_stdout:
            line["text_to_stdout"] = text_to_stdout
        emit(line)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
