"""The boundary is a real C ABI: a gcc-compiled C program (no Python, no torch in the
process) links libscaldpc.so, decodes, and checks its own results."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client(tmp_path):
    exe = tmp_path / "demo"
    libdir = os.path.join(ROOT, "sca-ldpc_amd")
    subprocess.check_call(
        ["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "demo.c"), "-o", str(exe),
         "-L", libdir, "-lscaldpc", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    )
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bp: 70/70 single errors corrected" in out.stdout
    assert "bad input kind -> 1" in out.stdout and "qary: all-zero decoding yes" in out.stdout
    assert "append: error at 4 found" in out.stdout and "into_llr: 1.9459101 inf" in out.stdout
