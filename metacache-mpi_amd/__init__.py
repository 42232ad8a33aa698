"""metacache-mpi_amd -- MI355X-native query-path engine for MetaCache-MPI.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI of
include/mcq.h) and a thin ctypes mirror of that ABI (engine.py).  There is no CPU
fallback: importing engine without the built HIP library raises.
"""
from .build import build_hip, build_host, lib_path, host_lib_path, cli_path, mpi_cli_path, mpi_lib_dir, source_digest  # noqa: F401
