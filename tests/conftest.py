import os
import sys

# torch's caching allocator must not carve small tensors out of the multi-GB blocks the workload generator's temporaries leave behind: a
# 10 MB tensor then pins a 3 GB segment, empty_cache() cannot return it, and the library (which allocates with hipMalloc) finds the
# device "full" (the 1 B-read test: 72 GB allocated, 245 GB reserved).  Blocks above 256 MB are kept whole and go back to the driver.
for _v in ("PYTORCH_HIP_ALLOC_CONF", "PYTORCH_CUDA_ALLOC_CONF"):
    os.environ.setdefault(_v, "max_split_size_mb:256")

import pytest
import torch  # noqa: F401  -- before the HIP library is loaded: torch ships its own HIP runtime, and whichever
# copy of libamdhip64 is loaded first serves both; torch only finds its GPUs through its own

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def spombe30k():
    """First 30 kb of the reference's tests/resources/spombe.III.fa (data fixture)."""
    from util_bam import read_fasta

    (name, seq), = read_fasta(os.path.join(GOLDEN, "spombe_III_30k.fa"))
    return name, seq
