"""csrc/scaldpc_logf.h restates glibc's logf for the device (the q-ary decoders' probability -> LLR
conversion must give the host's values bit for bit).  Here the header is compiled for the HOST and
compared with the host libm's logf: a 2^24 sample over all positive floats plus the edge cases on
every run, every positive float (2^31 - 2^23 values, ~10 s on 8 cores) with SCALDPC_LOGF_FULL=1.
The GPU side of the same claim is tests/test_qary_gpu.py::test_into_llr_on_the_device_*."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "scaldpc_logf.h"
int main(int argc, char **argv)
{
    const bool full = argc > 1 && atoi(argv[1]) != 0;
    long bad = 0, n = 0;
    // full: every positive float up to +inf; sample: every 127th bit pattern, phase-shifted
    const long step = full ? 1 : 127;
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
    for (long u = 1; u <= 0x7f800000L; u += step) {
        uint32_t ix = (uint32_t)(full ? u : u + (u >> 7) % 127);
        if (ix > 0x7f800000u) ix = 0x7f800000u;
        float x;
        memcpy(&x, &ix, 4);
        volatile float xv = x;
        const float ref = logf(xv), got = scaldpc::glibc_logf(x);
        uint32_t a, b;
        memcpy(&a, &ref, 4);
        memcpy(&b, &got, 4);
        bad += a != b;
        n++;
    }
    const float edge[] = {0.0f, -0.0f, 1.0f, INFINITY, -1.0f, NAN, 1e-45f, 1.17549435e-38f, 3.4028235e38f};
    for (float x : edge) {
        volatile float xv = x;
        const float ref = logf(xv), got = scaldpc::glibc_logf(x);
        const bool same = (std::isnan(ref) && std::isnan(got)) || (ref == got && std::signbit(ref) == std::signbit(got));
        bad += !same;
        n++;
    }
    printf("%ld %ld\n", n, bad);
    return bad != 0;
}
"""


def test_logf_port_equals_host_libm(tmp_path):
    src = tmp_path / "chk.cpp"
    src.write_text(SRC)
    exe = tmp_path / "chk"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fopenmp", "-I", os.path.join(ROOT, "sca-ldpc_amd", "csrc"),
                           str(src), "-o", str(exe), "-lm"])
    full = os.environ.get("SCALDPC_LOGF_FULL", "0")
    out = subprocess.run([str(exe), full], capture_output=True, text=True)
    n, bad = map(int, out.stdout.split())
    assert out.returncode == 0 and bad == 0, out.stdout
    assert n >= (1 << 24)
