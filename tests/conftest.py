import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ibm-cbc-genomic-tools_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # a fresh checkout has no built artefacts (they are git-ignored): build once -- hipcc cross-compiles without a GPU
    need = [os.path.join(PKG, "csrc", f) for f in ("libgtx.so", "genomic_overlaps", "genomic_scans", "permutation_test", "gtx_packtool", "api_caller")]
    need += [os.path.join(ROOT, "oracle", f) for f in ("libgtx_oracle.so", "gtx_oracle", "libperm_oracle.so", "perm_oracle")]
    if not all(os.path.exists(f) for f in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def engine():
    """One gtx context on cuda:0 for the whole GPU session (fails loudly without libgtx.so / a GPU)."""
    import gtx
    e = gtx.Engine(0)
    yield e
    e.close()
