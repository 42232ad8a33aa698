"""The in-register sorting networks (mcq_device.hpp: wave_sort64_1, wave_sort_blocks32_1, cx_chain6_1 and the two-register forms
wave_sort64_x2 / cx_chain6_x2 / wave_regsort) against std::sort, on the GPU: hand-written DPP sequences whose wait states the
assembler does not check."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_sort_networks_against_std_sort(tmp_path):
    exe = str(tmp_path / "sort_networks")
    src = os.path.join(ROOT, "tests", "native", "sort_networks.hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "metacache-mpi_amd", "csrc"),
                    "-I" + os.path.join(ROOT, "include"), src, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAILED" not in r.stdout
