"""Builds csrc/ into libmcq_hip.so (in-tree, so it travels with gpurun snapshots)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = [os.path.join(_HERE, "csrc", "mcq_engine.hip")]
_DEPS = _SRC + [os.path.join(_HERE, "csrc", "mcq_device.hpp"),
                os.path.join(os.path.dirname(_HERE), "include", "mcq.h")]


def lib_path():
    return os.path.join(_HERE, "libmcq_hip.so")


def build_hip(force=False, verbose=False):
    out = lib_path()
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in _DEPS):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"] + _SRC + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out
