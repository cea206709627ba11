"""Parity of the permutation-test path on the MI355X (include/gtx_perm.h, through the C ABI) against the CPU
restatement oracle/perm_oracle.c: permutations, statistics and exceed-counts bit for bit (same seed,
same permutation definition, same summation order); the CLI byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from gtx import perm
from oracle import porc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "ibm-cbc-genomic-tools_amd", "csrc", "permutation_test")
STATS = ["sum", "n", "sens", "spec", "ratio", "t", "corr"]


@pytest.fixture(scope="module")
def pe():
    e = perm.PermEngine(0)
    yield e
    e.close()


def same_bits(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64)) or (np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]))


def tables():
    """(name, table): the three shapes of StringSets -- one value per row (use_totals with all totals 1),
    two values (use_totals with real totals), two values normalised (-norm: no totals)."""
    yield "one-value", perm.PermTable.synthetic(3000, 150, 40, seed=1, values="gamma")
    yield "binary", perm.PermTable.synthetic(2500, 120, 30, seed=2, values="binary")
    yield "signed", perm.PermTable.synthetic(1000, 60, 25, seed=3, values="signed")
    yield "totals", perm.PermTable.synthetic(2000, 100, 35, seed=4, values="normal", totals=True)
    t = perm.PermTable.synthetic(2000, 100, 35, seed=5, values="gamma", totals=True)
    yield "normalised", perm.PermTable(t.n_rows, t.col_ptr, t.rows, (t.V / t.Vtotal).astype(np.float32), None, use_totals=False)


@pytest.mark.parametrize("n", [1, 2, 3, 16, 17, 31, 64, 100, 1000, 65537, 300001])
def test_permutation_matches_definition(pe, n):
    pe.set_table(perm.PermTable(n, [0, 1], [0], np.zeros(n)))
    for seed, q in ((0, 0), (12345, 7), (2**63 + 5, 10**9)):
        got = pe.permutation(seed, q)
        np.testing.assert_array_equal(got, porc.permutation(seed, q, n))
        assert np.array_equal(np.sort(got), np.arange(n))


@pytest.mark.parametrize("under", [False, True])
def test_observed_statistics_bit_exact(pe, under):
    for name, t in tables():
        pe.set_table(t)
        for stat in STATS:
            if stat == "corr" and not t.use_totals:
                continue
            got, want = pe.statistic(stat, under), porc.statistic(t, stat, under)
            assert same_bits(got, want), "%s %s under=%s: max diff %g" % (name, stat, under, np.nanmax(np.abs(got - want)))


@pytest.mark.parametrize("under", [False, True])
def test_exceed_counts_bit_exact(pe, under):
    for name, t in tables():
        pe.set_table(t)
        for stat in STATS:
            if stat == "corr" and not t.use_totals:
                continue
            Y = porc.statistic(t, stat, under)
            got = pe.count_ge(stat, Y, seed=99, first_perm=0, n_perm=150, under=under)
            want = porc.count_ge(t, stat, Y, 99, 0, 150, under)
            np.testing.assert_array_equal(got, want, err_msg="%s %s under=%s" % (name, stat, under))
            assert got.max() <= 150


def test_permutation_range_shards_add_up(pe):
    t = perm.PermTable.synthetic(5000, 200, 50, seed=8, values="normal")
    pe.set_table(t)
    Y = pe.statistic("sum")
    whole = pe.count_ge("sum", Y, 5, 0, 1000)
    parts = sum(pe.count_ge("sum", Y, 5, a, b - a).astype(np.int64) for a, b in ((0, 1), (1, 64), (64, 65), (65, 700), (700, 1000)))
    np.testing.assert_array_equal(whole.astype(np.int64), parts)
    np.testing.assert_array_equal(whole, porc.count_ge(t, "sum", Y, 5, 0, 1000))
    assert pe.count_ge("sum", Y, 5, 0, 0).sum() == 0


def test_config5_table_10k_shuffles(pe):
    """BASELINE config 5's permutation stage at its full size -- 20 k rows x 5 k categories (~1 M memberships), 10 k shuffles:
    the shuffle ranges a multi-GPU run deals to its ranks add up to the whole (8 uneven shards), a 24-shuffle range and the observed
    statistic against the CPU oracle bit for bit (the scalar restatement takes ~0.5 s per shuffle of this table), and every count
    within the number of shuffles."""
    t = perm.PermTable.synthetic(20000, 5000, 200, seed=1, values="gamma")
    pe.set_table(t)
    Y = pe.statistic("sum")
    assert same_bits(Y, porc.statistic(t, "sum"))
    whole = pe.count_ge("sum", Y, 2024, 0, 10000)
    cuts = [0, 1250, 2500, 2563, 5000, 6250, 7500, 9999, 10000]
    parts = sum(pe.count_ge("sum", Y, 2024, a, b - a).astype(np.int64) for a, b in zip(cuts[:-1], cuts[1:]))
    np.testing.assert_array_equal(whole.astype(np.int64), parts)
    assert whole.max() <= 10000 and whole.sum() > 0
    np.testing.assert_array_equal(pe.count_ge("sum", Y, 2024, 5000, 24), porc.count_ge(t, "sum", Y, 2024, 5000, 24))


@pytest.mark.parametrize("n_rows", [1500, 20000, 36000, 37000])
def test_tabulated_and_direct_slab_writers_agree(pe, monkeypatch, n_rows):
    """Tables with grid sides <= 192 (up to ~36 k rows) get their slab from perm_apply_tab_kernel (round functions tabulated in
    LDS); the others, and everything under GTX_PERM_NO_TABLE, from perm_apply_kernel: the same images bit for bit -- counts of a
    range that ends inside a tile of 64 permutations, one- and two-value tables, against each other and (a short range) the oracle."""
    for totals in (False, True):
        t = perm.PermTable.synthetic(n_rows, 300, 60, seed=11, values="gamma", totals=totals)
        pe.set_table(t)
        for stat in ("sum", "t"):
            Y = pe.statistic(stat)
            monkeypatch.delenv("GTX_PERM_NO_TABLE", raising=False)
            a = pe.count_ge(stat, Y, 31, 3, 333)                 # (value vector in LDS beside a 32-permutation table where both fit: 1500 and 20000 rows)
            monkeypatch.setenv("GTX_PERM_NO_LDS_VALUES", "1")
            c = pe.count_ge(stat, Y, 31, 3, 333)                 # (64-permutation table, values gathered from memory)
            monkeypatch.delenv("GTX_PERM_NO_LDS_VALUES")
            monkeypatch.setenv("GTX_PERM_NO_TABLE", "1")
            b = pe.count_ge(stat, Y, 31, 3, 333)
            monkeypatch.delenv("GTX_PERM_NO_TABLE")
            np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(c, b)
            np.testing.assert_array_equal(pe.count_ge(stat, Y, 31, 3, 20), porc.count_ge(t, stat, Y, 31, 3, 20))


def test_small_slab_budget_batches_give_the_same_counts(pe):
    t = perm.PermTable.synthetic(40000, 100, 60, seed=9, values="gamma", totals=True)
    pe.set_table(t)
    Y = pe.statistic("t")
    want = pe.count_ge("t", Y, 21, 0, 500)
    os.environ["GTX_PERM_SLAB_MB"] = "16"                  # 16 MiB / (40000 rows x 4 B) -> 64 permutations per batch
    try:
        e2 = perm.PermEngine(0)
    finally:
        del os.environ["GTX_PERM_SLAB_MB"]
    e2.set_table(t)
    np.testing.assert_array_equal(e2.count_ge("t", Y, 21, 0, 500), want)
    e2.close()


def test_tiny_tables(pe):
    # 1..17 rows, categories of every size incl. the whole table and an empty one
    for n in (1, 2, 5, 16, 17):
        rows, ptr = [], [0]
        for size in [0, 1, n, max(1, n // 2)]:
            rows += list(range(size)); ptr.append(len(rows))
        t = perm.PermTable(n, ptr, rows if rows else [0], np.arange(n) - 1.5)
        t.rows = np.ascontiguousarray(rows, dtype=np.int32)
        pe.set_table(t)
        for stat in ("sum", "n", "t"):
            Y = porc.statistic(t, stat)
            assert same_bits(pe.statistic(stat), Y)
            np.testing.assert_array_equal(pe.count_ge(stat, Y, 3, 0, 100), porc.count_ge(t, stat, Y, 3, 0, 100))


@pytest.mark.parametrize("under", [False, True])
def test_approx_rank_histogram(pe, under):
    t = perm.PermTable.synthetic(3000, 180, 30, seed=11, values="signed")
    pe.set_table(t)
    tab_ptr, tab = porc.hypergeom_table(t, under)
    k = porc.statistic(t, "n", under).astype(np.int64)
    sorted_y = np.sort(tab[tab_ptr[:-1] + k], kind="stable")
    got = pe.count_rank(tab_ptr, tab, sorted_y, 17, 0, 300, under)
    np.testing.assert_array_equal(got, porc.count_rank(t, tab_ptr, tab, sorted_y, 17, 0, 300, under))
    assert got.sum() <= 300 * t.n_cols


def without_small_categories(t, kmin=5):
    """the table without the categories of fewer than kmin rows (what -kmin does in the tool): t / corr have no degrees of freedom there"""
    keep = [c for c in range(t.n_cols) if t.col_ptr[c + 1] - t.col_ptr[c] >= kmin]
    col_ptr = np.zeros(len(keep) + 1, dtype=np.int64)
    rows = []
    for i, c in enumerate(keep):
        rows.append(t.rows[t.col_ptr[c]:t.col_ptr[c + 1]]); col_ptr[i + 1] = col_ptr[i] + len(rows[-1])
    return perm.PermTable(t.n_rows, col_ptr, np.concatenate(rows).astype(np.int32), t.V, t.Vtotal if t.has_totals else None, use_totals=t.use_totals)


APPROX_CASES = [("t", "one-value"), ("t", "totals"), ("t", "normalised"), ("ratio", "normalised"), ("corr", "totals"), ("t", "signed")]


@pytest.mark.parametrize("under", [False, True])
@pytest.mark.parametrize("stat,shape", APPROX_CASES, ids=["%s-%s" % c for c in APPROX_CASES])
def test_approximate_p_values_and_their_rank_histogram(pe, stat, shape, under):
    """-a for ratio / t / corr (Calc*Statistic(approx = true) + RunApproxPermutations, permutation_test.cpp:305-308, :336-339, :447-451,
    :542, :612-627): the observed p-values within 1e-12 of the oracle's (same arithmetic; the device's log / exp / lgamma / erfc differ
    from the host's in the last bits), the rank histogram of 200 permutations against the oracle's sort + merge.  The histogram is
    compared bin for bin: a p-value of a permutation would have to fall within those last bits of an observed one WITHOUT being the
    same statistic to move."""
    t = without_small_categories(dict(tables())[shape])
    pe.set_table(t)
    want = porc.statistic_approx(t, stat, under)
    got = pe.statistic_approx(stat, under)
    assert not np.isnan(want).any()
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-300)
    # each side ranks against ITS OWN observed values: a permutation that reproduces an observed statistic then gives exactly that
    # p-value on either side (the tool takes the observed ones from the device for this reason)
    sorted_y = np.sort(got, kind="stable")
    h = pe.count_rank_approx(stat, sorted_y, 29, 0, 200, under)
    np.testing.assert_array_equal(h, porc.count_rank_approx(t, stat, np.sort(want, kind="stable"), 29, 0, 200, under))
    assert 0 < h.sum() <= 200 * t.n_cols
    # shards of the permutation range add up
    np.testing.assert_array_equal(pe.count_rank_approx(stat, sorted_y, 29, 0, 64, under) + pe.count_rank_approx(stat, sorted_y, 29, 64, 136, under), h)


def test_approx_rank_histogram_at_the_config5_table(pe):
    """-S t -a at BASELINE config 5's table (20 k rows x 5 k categories): the rank histogram of 10 k permutations adds up over the
    ranges a multi-GPU run deals out, a 16-permutation range against the oracle's sort + merge bin for bin, and every (category,
    permutation) pair is counted once or falls beyond the largest observed value"""
    t = without_small_categories(perm.PermTable.synthetic(20000, 5000, 200, seed=1, values="gamma"))
    tn = perm.PermTable(t.n_rows, t.col_ptr, t.rows, t.V, None, use_totals=False)
    pe.set_table(tn)
    got = pe.statistic_approx("t")
    want = porc.statistic_approx(tn, "t")
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-300)
    sorted_y = np.sort(got, kind="stable")
    whole = pe.count_rank_approx("t", sorted_y, 77, 0, 10000).astype(np.int64)
    cuts = [0, 1250, 2563, 5000, 9999, 10000]
    parts = sum(pe.count_rank_approx("t", sorted_y, 77, a, b - a).astype(np.int64) for a, b in zip(cuts[:-1], cuts[1:]))
    np.testing.assert_array_equal(whole, parts)
    assert 0 < whole.sum() <= 10000 * tn.n_cols
    # p-values of random arrangements are close to uniform: about half of the pairs lie below the median observed p-value... of a
    # uniform sample; here the observed ones are themselves a sample of the null, so the histogram's total is a large share of all pairs
    assert whole.sum() > 0.5 * 10000 * tn.n_cols
    np.testing.assert_array_equal(pe.count_rank_approx("t", sorted_y, 77, 5000, 16), porc.count_rank_approx(tn, "t", np.sort(want, kind="stable"), 77, 5000, 16))


def test_approx_refused_where_the_reference_has_no_distribution(pe):
    t = dict(tables())["totals"]
    pe.set_table(t)
    for stat in ("sum", "sens", "spec", "n", "ratio"):
        with pytest.raises(Exception):
            pe.statistic_approx(stat)
        with pytest.raises(Exception):
            pe.count_rank_approx(stat, np.zeros(t.n_cols), 1, 0, 10)


def test_p_values_estimate_the_exact_hypergeometric_tail(pe):
    """-S n: P(k_perm >= k_obs) is a hypergeometric tail; 20000 permutations must land within 5 binomial
    standard errors of it for every category (statistical check of the permutation source on the device)."""
    from scipy import stats
    t = perm.PermTable.synthetic(4000, 120, 40, seed=13, values="binary")
    pe.set_table(t)
    k = pe.statistic("n")
    P = 20000
    p_hat = pe.count_ge("n", k, 2024, 0, P) / P
    n1 = np.diff(t.col_ptr); pos = int((t.V > 0).sum())
    exact = stats.hypergeom.sf(k - 1, t.n_rows, n1, pos)
    se = np.sqrt(np.maximum(exact * (1 - exact), 1e-9) / P)
    assert np.all(np.abs(p_hat - exact) < 5 * se + 2.0 / P), np.max(np.abs(p_hat - exact) / se)


def run_both(args, seed):
    env = dict(os.environ, GTX_PERM_SEED=str(seed))
    a = subprocess.run([CLI] + args, capture_output=True, env=env)
    b = porc.run_cli(args, env={"GTX_PERM_SEED": str(seed)})
    return a, b


@pytest.mark.parametrize("args", [
    ["-h", "-S", "n", "-p", "300", "-q", "0.5"],
    ["-S", "sum", "-p", "200"],
    ["-S", "sum", "-u", "-p", "100", "-f"],
    ["-h", "-S", "sens", "-p", "100", "-kmin", "20"],
    ["-S", "spec", "-p", "100", "-kmin", "5", "-kmax", "40"],
    ["-S", "ratio", "-p", "150", "-d", "-q", "0.3"],
    ["-S", "t", "-p", "150"],
    ["-h", "-S", "n", "-a", "-p", "200", "-q", "0.05"],
    ["-S", "n", "-a", "-u", "-p", "100"],
    ["-h", "-S", "t", "-a", "-p", "150"],
    ["-S", "t", "-a", "-u", "-p", "100", "-f"],
])
def test_cli_matches_oracle_cli_one_value_table(args):
    a, b = run_both(args + [os.path.join(GOLD, "perm_go.txt")], seed=42)
    assert a.returncode == 0 and b.returncode == 0, a.stderr.decode() + b.stderr.decode()
    assert a.stdout == b.stdout
    assert len(a.stdout.splitlines()) > 3


@pytest.mark.parametrize("args", [
    ["-S", "sum", "-p", "100"],
    ["-S", "ratio", "-p", "100", "-u"],
    ["-S", "t", "-p", "100"],
    ["-S", "corr", "-p", "100"],
    ["-S", "corr", "-u", "-p", "50"],
    ["-norm", "-S", "t", "-p", "100"],
    ["-norm", "-S", "ratio", "-p", "100"],
    ["-norm", "-S", "sum", "-v", "-p", "64"],
    ["-S", "t", "-a", "-p", "100"],
    ["-S", "corr", "-a", "-p", "100"],
    ["-S", "corr", "-a", "-u", "-p", "64", "-f"],
    ["-norm", "-S", "t", "-a", "-p", "100"],
    ["-norm", "-S", "ratio", "-a", "-p", "100"],
    ["-norm", "-S", "ratio", "-a", "-u", "-p", "100", "-q", "0.9"],
])
def test_cli_matches_oracle_cli_two_value_table(args):
    a, b = run_both(args + [os.path.join(GOLD, "perm_go2.txt")], seed=7)
    assert a.returncode == 0 and b.returncode == 0, a.stderr.decode() + b.stderr.decode()
    assert a.stdout == b.stdout
    assert a.stderr == b.stderr
    assert len(a.stdout.splitlines()) > 3


def test_cli_values_from_second_file(tmp_path):
    lines = open(os.path.join(GOLD, "perm_go.txt")).read().splitlines()
    cats, vals = tmp_path / "cats.txt", tmp_path / "vals.txt"
    cats.write_text("".join("%s\t%s\n" % (l.split("\t")[0], l.split("\t")[2]) for l in lines))
    vals.write_text("".join("%s\n" % l.split("\t")[1] for l in lines))
    a1, b1 = run_both(["-S", "n", "-p", "100", str(cats), str(vals)], seed=5)
    a2, _ = run_both(["-S", "n", "-p", "100", os.path.join(GOLD, "perm_go.txt")], seed=5)
    assert a1.returncode == 0, a1.stderr.decode()
    assert a1.stdout == b1.stdout == a2.stdout


def test_cli_errors_like_the_reference():
    f = os.path.join(GOLD, "perm_go.txt")
    a, b = run_both(["-S", "bogus", f], 1)
    assert a.returncode == 1 and a.stderr == b.stderr == b"Error: unknown statistic 'bogus'!\n"
    a, b = run_both(["-norm", "-S", "corr", os.path.join(GOLD, "perm_go2.txt")], 1)
    assert a.returncode == 1 and a.stderr == b.stderr == b"Error: this operation is not permitted!\n"
    for args in (["-S", "sum", "-a", f], ["-S", "sens", "-a", f], ["-S", "ratio", "-a", os.path.join(GOLD, "perm_go2.txt")]):
        a, b = run_both(args, 1)
        assert a.returncode == 1 and a.stderr == b.stderr == b"Error: not implemented yet!\n"


def test_cli_reader_rules_match_oracle(tmp_path):
    f = tmp_path / "t.txt"
    f.write_text("g1\t1\ta b \ng2\t0\ta  b  \ng3\t1\ta a\n   g5 \t 2 \t b   c\ng4\t5\tb")
    for args in (["-kmin", "1", "-S", "n", "-p", "50", "-h"], ["-kmin", "1", "-kmax", "10", "-S", "sum", "-p", "64", "-d"]):
        a, b = run_both(args + [str(f)], seed=4)
        assert a.returncode == 0 and a.stdout == b.stdout and len(a.stdout) > 10


def test_cli_no_category_passes_the_support_filter():
    a, b = run_both(["-kmin", "100000", "-S", "sum", "-p", "10", "-h", os.path.join(GOLD, "perm_go.txt")], seed=1)
    assert a.returncode == 0 and b.returncode == 0
    assert a.stdout == b.stdout == b"CATEGORY\tCATEGORY-SIZE\tQ-VALUE\tP-VALUE\tSTATISTIC\n"


@pytest.mark.parametrize("l2_mb", ["0.05", "0.3", "100"])
def test_row_range_parts_keep_every_bit(pe, monkeypatch, l2_mb):
    """perm_stat_kernel cuts the rows of a slab tile into L2-sized ranges, one launch per range, the accumulators travelling between
    the launches: the members of a category are still added in list order, so exceed-counts and the rank histogram are bit for
    bit what one launch gives (and what the oracle gives) -- for every statistic, with and without totals, for 2 to 16 ranges
    (GTX_PERM_L2_MB shrinks the budget so that small tables are cut too), and for tables whose lists are NOT in ascending row
    order (no ranges then)."""
    monkeypatch.setenv("GTX_PERM_L2_MB", l2_mb)
    for name, t in tables():
        pe.set_table(t)
        for stat in STATS:
            if stat == "corr" and not t.use_totals:
                continue
            Y = porc.statistic(t, stat, False)
            got = pe.count_ge(stat, Y, seed=7, first_perm=3, n_perm=200)
            np.testing.assert_array_equal(got, porc.count_ge(t, stat, Y, 7, 3, 200), err_msg="%s %s" % (name, stat))
    # lists in descending row order: the ranges do not apply, the result is the same
    t = perm.PermTable.synthetic(3000, 150, 40, seed=11, values="gamma")
    rows = t.rows.copy()
    for c in range(len(t.col_ptr) - 1):
        rows[t.col_ptr[c]:t.col_ptr[c + 1]] = rows[t.col_ptr[c]:t.col_ptr[c + 1]][::-1]
    t2 = perm.PermTable(t.n_rows, t.col_ptr, rows, t.V, None)
    pe.set_table(t2)
    Y = porc.statistic(t2, "t", False)
    np.testing.assert_array_equal(pe.count_ge("t", Y, 7, 0, 128), porc.count_ge(t2, "t", Y, 7, 0, 128))
    # the approximate (rank histogram) mode goes through the same kernel
    t3 = perm.PermTable.synthetic(3000, 180, 30, seed=11, values="signed")
    pe.set_table(t3)
    tab_ptr, tab = porc.hypergeom_table(t3, False)
    k = porc.statistic(t3, "n", False).astype(np.int64)
    sorted_y = np.sort(tab[tab_ptr[:-1] + k], kind="stable")
    np.testing.assert_array_equal(pe.count_rank(tab_ptr, tab, sorted_y, 17, 0, 300, False), porc.count_rank(t3, tab_ptr, tab, sorted_y, 17, 0, 300, False))


@pytest.mark.parametrize("rows32", ["0", "1"])
def test_packed_and_plain_membership_lists(pe, monkeypatch, rows32):
    """Tables of <= 65536 rows keep their membership lists as 16-bit row ids, two per word (half the bytes next to the slab rows in
    the L2); larger tables, or GTX_PERM_ROWS32=1, read the 32-bit lists.  Lists that start at odd members, lists of 0..9 members
    (head, the blocks of 8, the tail), with and without row ranges: the same bits either way, and the oracle's."""
    monkeypatch.setenv("GTX_PERM_ROWS32", rows32)
    rng = np.random.default_rng(5)
    n_rows = 700
    sizes = [0, 1, 2, 7, 8, 9, 15, 16, 17, 3, 64, 65] + list(rng.integers(0, 40, size=60))
    col_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    rows = np.concatenate([np.sort(rng.choice(n_rows, size=s, replace=False)) for s in sizes]).astype(np.int32)
    V = rng.gamma(2.0, 3.0, size=n_rows).astype(np.float32)
    t = perm.PermTable(n_rows, col_ptr, rows, V, None)
    for l2 in ("100", "0.05"):
        monkeypatch.setenv("GTX_PERM_L2_MB", l2)
        pe.set_table(t)
        for stat in ("sum", "n", "t"):
            Y = porc.statistic(t, stat, False)
            assert same_bits(pe.statistic(stat, False), Y)
            np.testing.assert_array_equal(pe.count_ge(stat, Y, seed=3, first_perm=1, n_perm=130), porc.count_ge(t, stat, Y, 3, 1, 130), err_msg=stat)
    if rows32 == "0":                                    # more rows than 16 bits take: the 32-bit lists by themselves
        big = perm.PermTable.synthetic(70000, 40, 30, seed=4, values="normal")
        pe.set_table(big)
        Y = porc.statistic(big, "sum", False)
        np.testing.assert_array_equal(pe.count_ge("sum", Y, 9, 0, 64), porc.count_ge(big, "sum", Y, 9, 0, 64))
