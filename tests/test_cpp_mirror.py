"""The C++ host mirror (include/perceive.hpp): compiles against the C ABI with plain g++ (CPU), and
its end-to-end program passes on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.bin")


def compile_mirror():
    subprocess.run(
        ["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), SRC, "-o", BIN,
         "-L" + os.path.join(ROOT, "perceive_amd"), "-lperceive_hip", "-Wl,-rpath," + os.path.join(ROOT, "perceive_amd")],
        check=True, capture_output=True, text=True)


def test_cpp_mirror_compiles_and_links():
    compile_mirror()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_cpp_mirror_runs_on_gpu():
    if not os.path.exists(BIN):
        compile_mirror()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_mirror_test: ok" in r.stdout
