"""Builds and runs the C++ host-mirror parity tests (tests/host/test_host_mirror.cpp) on the GPU box."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "host" / "test_host_mirror.cpp"
EXE = ROOT / "tests" / "host" / "test_host_mirror"


def build():
    from oramacore_amd import _build

    _build.build_native()
    subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    csrc = ROOT / "oramacore_amd" / "csrc"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", f"-I{ROOT / 'include'}", f"-I{ROOT / 'oracle'}", str(SRC), "-o", str(EXE),
           f"-L{csrc}", "-lorama_hip", f"-L{ROOT / 'oracle'}", "-lorama_oracle", "-pthread",
           f"-Wl,-rpath,{csrc}", f"-Wl,-rpath,{ROOT / 'oracle'}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_host_mirror_compiles():
    """CPU: the C++ mirror and its tests compile and link against the C ABI (no GPU needed)."""
    assert build().exists()


@pytest.mark.gpu
def test_host_mirror_parity():
    exe = build()
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout + r.stderr
