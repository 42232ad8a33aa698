"""Builds csrc/ into libmcq_hip.so (in-tree, so it travels with gpurun snapshots)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_HDR = os.path.join(os.path.dirname(_HERE), "include", "mcq.h")
# translation unit -> what it depends on besides itself
_UNITS = {
    os.path.join(_HERE, "csrc", "mcq_engine.hip"): [os.path.join(_HERE, "csrc", "mcq_device.hpp"), os.path.join(_HERE, "csrc", "mcq_shard.hpp"), _HDR],
    os.path.join(_HERE, "csrc", "mcq_build.hip"): [_HDR],       # table construction (rocPRIM sorts)
}
_OBJ = os.path.join(_HERE, "csrc", "_obj")


def lib_path():
    """MCQ_HIP_LIB (tuning knob): another build of the library, e.g. a variant for a same-box A/B (scripts/ab_libs.sh)"""
    return os.environ.get("MCQ_HIP_LIB") or os.path.join(_HERE, "libmcq_hip.so")


def host_lib_path():
    return os.path.join(_HERE, "libmcq_host.so")


def cli_path():
    return os.path.join(_HERE, "mcq_query_cli")


def mpi_cli_path():
    """mcq_query_mpi (one process per GPU under mpiexec); built only where an MPI is installed (/opt/conda: MPICH)"""
    return os.path.join(_HERE, "mcq_query_mpi")


_MPI_ROOT = os.environ.get("MCQ_MPI_ROOT", "/opt/conda")
_MPI_LIBS = ["libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0", "libgomp.so.1"]


def mpi_lib_dir():
    """private directory with links to libmpi and what it needs, so that conda's old libstdc++ is not picked up at run time"""
    return os.path.join(_HERE, "_mpilib")


def source_digest():
    """sha256 (first 16 hex digits) over the kernel sources: profiles taken from one state of the kernels (PMC traffic,
    profiles/pmc_traffic_*.json) carry it, and bench.py refuses to quote them for another"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(os.path.join(_HERE, "csrc"))):
        if f.endswith((".hip", ".hpp")):
            with open(os.path.join(_HERE, "csrc", f), "rb") as fh:
                h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def build_host(force=False, verbose=False):
    """libmcq_host.so (shard reader, taxonomy keys, classify; g++, no GPU) and the
    mcq_query_cli binary (links both libraries)."""
    src = os.path.join(_HERE, "csrc", "host", "mcq_host.cpp")
    hdr = os.path.join(os.path.dirname(_HERE), "include", "mcq_host.h")
    out = host_lib_path()
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        cmd = ["g++", "-std=c++14", "-O2", "-Wall", "-shared", "-fPIC", src, "-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    cli_src = os.path.join(_HERE, "csrc", "host", "mcq_query_cli.cpp")
    cli = cli_path()
    open_hpp = os.path.join(os.path.dirname(_HERE), "include", "mcq_open.hpp")
    if force or not os.path.exists(cli) or os.path.getmtime(cli) < max(os.path.getmtime(cli_src), os.path.getmtime(out), os.path.getmtime(lib_path()),
                                                                         os.path.getmtime(os.path.join(_HERE, "csrc", "host", "mcq_cli_common.hpp")),
                                                                         os.path.getmtime(open_hpp)):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "-std=c++14", "-O2", "-pthread", cli_src, "-o", cli, "-L" + _HERE, "-lmcq_hip", "-lmcq_host", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # the MPI program (multi-GPU host in C++): only where an MPI is installed
    mpi_src = os.path.join(_HERE, "csrc", "host", "mcq_query_mpi.cpp")
    common = os.path.join(_HERE, "csrc", "host", "mcq_cli_common.hpp")
    mpi_h = os.path.join(_MPI_ROOT, "include", "mpi.h")
    mpi_so = os.path.join(_MPI_ROOT, "lib", _MPI_LIBS[0])
    if os.path.exists(mpi_h) and os.path.exists(mpi_so):
        os.makedirs(mpi_lib_dir(), exist_ok=True)
        for l in _MPI_LIBS:
            dst = os.path.join(mpi_lib_dir(), l)
            if not os.path.lexists(dst) and os.path.exists(os.path.join(_MPI_ROOT, "lib", l)):
                os.symlink(os.path.join(_MPI_ROOT, "lib", l), dst)
        mpi_cli = mpi_cli_path()
        newest = max(os.path.getmtime(f) for f in (mpi_src, common, out, lib_path(), open_hpp))
        if force or not os.path.exists(mpi_cli) or os.path.getmtime(mpi_cli) < newest:
            rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
            # plain g++ (host code only; hipcc would take libmpi.so.12 for a source file)
            cmd = ["g++", "-std=c++14", "-O2", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), "-I" + os.path.join(_MPI_ROOT, "include"),
                   mpi_src, "-o", mpi_cli, "-L" + _HERE, "-lmcq_hip", "-lmcq_host", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", mpi_so,
                   "-Wl,-rpath-link," + os.path.join(_MPI_ROOT, "lib"), "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/_mpilib",
                   "-Wl,-rpath," + os.path.join(rocm, "lib")]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return out


def build_hip(force=False, verbose=False):
    """hipcc every unit of csrc/ for gfx950 into csrc/_obj/*.o, link libmcq_hip.so"""
    if os.environ.get("MCQ_HIP_LIB"):          # a prebuilt variant was asked for: nothing to build
        return lib_path()
    out = lib_path()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(_OBJ, exist_ok=True)
    extra = os.environ.get("MCQ_HIPCC_FLAGS", "").split()      # tuning experiments, e.g. -DMCQ_WAVE_OCC=7
    force = force or bool(extra)
    objs, relink = [], force or not os.path.exists(out)
    for src, deps in _UNITS.items():
        obj = os.path.join(_OBJ, os.path.basename(src) + ".o")
        objs.append(obj)
        newest = max(os.path.getmtime(f) for f in [src] + deps)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            relink = True
        elif os.path.exists(out) and os.path.getmtime(out) < os.path.getmtime(obj):
            relink = True
    if relink:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out, "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out
