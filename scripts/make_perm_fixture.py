#!/usr/bin/env python3
"""Writes tests/golden/perm_go.txt and perm_go2.txt: input tables for permutation_test in the shape of the
reference's own example (examples/example06.tcsh: LABEL <tab> VALUE <tab> GO categories), built from the
first rows of the reference's data file examples/gene.go plus seeded synthetic values.  Run once, here;
the outputs are committed (the reference tree does not exist on the GPU box)."""
import os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = [l.rstrip("\n").split("\t") for l in open("/root/reference/examples/gene.go")][:600]
rng = np.random.default_rng(6)
with open(os.path.join(R, "tests/golden/perm_go.txt"), "w") as f:          # one value per row: peak count, mostly 0
    for g, cats in rows:
        hot = "transcription" in cats or "development" in cats
        v = int(rng.poisson(1.2 if hot else 0.25))
        f.write("%s\t%d\t%s\n" % (g, v, cats))
with open(os.path.join(R, "tests/golden/perm_go2.txt"), "w") as f:         # two values per row: signal and total, signed signal
    for g, cats in rows[:400]:
        hot = "transport" in cats
        tot = float(rng.integers(5, 200))
        v = rng.normal(0.3 if hot else 0.0, 1.0) * tot / 10
        f.write("%s\t%.3f %.1f\t%s\n" % (g, v, tot, cats))
