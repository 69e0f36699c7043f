// logf exactly as glibc (>= 2.27) computes it, for host and device code.
//
// Why: the q-ary decoders turn channel probabilities into LLRs with `ln(max_p / p)` in f32
// (simulate_rs/src/decoder.rs:668-692; Rust's f32::ln is the platform libm's logf), and hard
// decisions are compared bit for bit.  The device's own logf differs from glibc's in the last
// bit on a fraction of inputs, so the conversion used to run on host threads.  This is glibc's
// algorithm (sysdeps/ieee754/flt-32/e_logf.c: 16-entry table of 1/c and log c in double, a cubic
// in r = z/c - 1, one final rounding to float) restated for the device: table look-up, six
// double-precision operations, one conversion.  Checked against the host libm over ALL positive
// floats (tests/test_logf_port.py runs a sample on every CPU run and the full sweep on request):
// 0 mismatches, and the result is the same with and without fused multiply-adds, so it does not
// depend on which variant of logf the host's glibc dispatches to.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SCALDPC_HD __host__ __device__
#else
#define SCALDPC_HD
#endif

namespace scaldpc {

SCALDPC_HD inline float glibc_logf(float x)
{
    // tab[i] = {1/c_i, log(c_i)} for the 16 subintervals of [0x1.66p-1, 0x1.66p0)
    const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                             0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                             0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                             0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
    const double logc[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                             -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                             -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                             0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    const double LN2 = 0x1.62e42fefa39efp-1;
    uint32_t ix;
    memcpy(&ix, &x, 4);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {  // x < 0x1p-126, or inf, or nan
        if (ix * 2 == 0) return -__builtin_inff();        // log(+-0) = -inf
        if (ix == 0x7f800000u) return x;                  // log(inf) = inf
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __builtin_nanf("");
        const float xs = x * 0x1p23f;  // subnormal: normalise
        memcpy(&ix, &xs, 4);
        ix -= 23u << 23;
    }
    // x = 2^k z with z in [OFF, 2 OFF); i = subinterval of z
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    float zf;
    memcpy(&zf, &iz, 4);
    const double z = (double)zf;
    // log(x) = log1p(z/c - 1) + log(c) + k ln2
    const double r = z * invc[i] - 1.0;
    const double y0 = logc[i] + (double)k * LN2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

}  // namespace scaldpc
