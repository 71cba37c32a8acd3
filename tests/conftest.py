import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """In-tree build products; build them if a fresh checkout has none."""
    import util
    need = [os.path.join(util.ROOT, "t1k_amd", "lib", "libt1k_gpu.so"), util.ORACLE_SO, util.ORACLE_CLI, util.ORACLE_EXTRACT, util.SYNTH,
            os.path.join(util.ROOT, "t1k_amd", "bin", "genotyper"), os.path.join(util.ROOT, "t1k_amd", "bin", "fastq-extractor")]
    if os.path.exists("/root/reference/Genotyper.cpp"):  # the reference-built checkers can be (re)built here
        need += [util.REF_BIN, util.REF_EXTRACT, util.REF_ANALYZER]
    if not all(os.path.exists(p) for p in need):
        sys.path.insert(0, util.ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return True
