"""Seeded synthetic BED-shaped inputs of the shapes BASELINE.json names.

Everything is packed int32 (class, start, end), 1-based inclusive.  Class ids are the
strcmp ranks of the hg38 chromosome names (chr1 < chr10 < ... < chr2 < ... < chrX < chrY),
optionally folded with the strand as  class = strand * n_chrom + chrom_rank  so that a stream
sorted by (strand, chromosome, start) -- `sortbed` order, bin/sortbed: -k1,1 -k6,6 -k2,2n is
chromosome-major; the packer regroups it strand-major -- is sorted by (class, start).
"""
import numpy as np

HG38 = {
    "chr1": 248956422, "chr2": 242193529, "chr3": 198295559, "chr4": 190214555, "chr5": 181538259,
    "chr6": 170805979, "chr7": 159345973, "chr8": 145138636, "chr9": 138394717, "chr10": 133797422,
    "chr11": 135086622, "chr12": 133275309, "chr13": 114364328, "chr14": 107043718, "chr15": 101991189,
    "chr16": 90338345, "chr17": 83257441, "chr18": 80373285, "chr19": 58617616, "chr20": 64444167,
    "chr21": 46709983, "chr22": 50818468, "chrX": 156040895, "chrY": 57227415,
}
CHROM_NAMES = sorted(HG38)                       # strcmp order == class id order
CHROM_LEN = np.array([HG38[c] for c in CHROM_NAMES], dtype=np.int64)


def apportion(total, weights):
    """Largest-remainder split of `total` items proportional to `weights`."""
    w = np.asarray(weights, dtype=np.float64)
    raw = total * w / w.sum()
    base = np.floor(raw).astype(np.int64)
    rest = int(total - base.sum())
    if rest:
        base[np.argsort(-(raw - base))[:rest]] += 1
    return base


def reads_single_chrom(n, length=50, chrom_len=HG38["chr1"], seed=42, cls=0):
    """C2: n reads of fixed length, uniform starts on one chromosome, sorted by start."""
    rng = np.random.default_rng(seed)
    s = np.sort(rng.integers(1, chrom_len - length, size=n, dtype=np.int64)).astype(np.int32)
    out = np.empty((n, 3), dtype=np.int32)
    out[:, 0] = cls
    out[:, 1] = s
    out[:, 2] = s + (length - 1)
    return out


def refs_single_chrom(m, min_len=50, max_len=2000, chrom_len=HG38["chr1"], seed=42, cls=0):
    """C2 'exons': uniform starts, uniform lengths in [min_len, max_len), sorted, may overlap."""
    rng = np.random.default_rng(seed + 1000)
    s = np.sort(rng.integers(1, chrom_len - max_len, size=m, dtype=np.int64))
    ln = rng.integers(min_len, max_len, size=m, dtype=np.int64)
    out = np.empty((m, 3), dtype=np.int32)
    out[:, 0] = cls
    out[:, 1] = s
    out[:, 2] = s + ln - 1
    return out


def genome_intervals(n, seed, min_len, max_len, stranded=False, chrom_subset=None, sort=True):
    """n intervals spread over the hg38 chromosomes in proportion to their length.

    Returns (n,3) int32 sorted by (class, start) when sort=True.  With stranded=True the
    class id is strand * 24 + chrom_rank and strands are drawn uniformly.
    """
    rng = np.random.default_rng(seed)
    idx = np.arange(len(CHROM_NAMES)) if chrom_subset is None else np.asarray(chrom_subset)
    per = apportion(n, CHROM_LEN[idx])
    parts = []
    for ci, cnt in zip(idx, per):
        if cnt == 0:
            continue
        hi = int(CHROM_LEN[ci]) - max_len
        s = rng.integers(1, hi, size=int(cnt), dtype=np.int64)
        ln = np.full(int(cnt), min_len, dtype=np.int64) if max_len <= min_len + 1 else rng.integers(min_len, max_len, size=int(cnt), dtype=np.int64)
        cls = np.full(int(cnt), ci, dtype=np.int64)
        if stranded:
            cls = cls + len(CHROM_NAMES) * rng.integers(0, 2, size=int(cnt), dtype=np.int64)
        parts.append(np.stack([cls, s, s + ln - 1], axis=1))
    a = np.concatenate(parts, axis=0) if parts else np.zeros((0, 3), dtype=np.int64)
    if sort and len(a):
        a = a[np.lexsort((a[:, 1], a[:, 0]))]
    return a.astype(np.int32)


def n_classes(stranded=False):
    return len(CHROM_NAMES) * (2 if stranded else 1)


def lpt_shards(weights, n_shards):
    """Longest-processing-time assignment of items (chromosomes) to shards; returns list of index lists."""
    order = np.argsort(-np.asarray(weights, dtype=np.float64), kind="stable")
    loads = [0.0] * n_shards
    out = [[] for _ in range(n_shards)]
    for i in order:
        k = int(np.argmin(loads))
        out[k].append(int(i))
        loads[k] += float(weights[i])
    return [sorted(x) for x in out]
