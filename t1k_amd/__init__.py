"""t1k_amd -- MI355X-native (gfx950, hand-written HIP) implementation of T1K's genotyper hot path.

The product is the C-ABI shared library built from t1k_amd/csrc (include/t1k_gpu.h) plus the drop-in `genotyper`
executable; this package only holds the ctypes binding used by tests, bench.py and __graft_entry__.py.
"""
from .capi import (Context, Job, Reads, Params, JobParams, lib, lib_path, T1kError, load_reference_fasta,
                   read_fastx, OVERLAP_DTYPE, ROW_DTYPE, FRAG_ASG_DTYPE, VARIANT_DTYPE, Variants, fragment_details, Comm, CommGroup, comm_unique_id, pool_release)

__all__ = ["Context", "Job", "Reads", "Params", "JobParams", "lib", "lib_path", "T1kError", "load_reference_fasta",
           "read_fastx", "OVERLAP_DTYPE", "ROW_DTYPE", "FRAG_ASG_DTYPE", "VARIANT_DTYPE", "Variants", "fragment_details", "Comm", "CommGroup", "comm_unique_id", "pool_release"]
