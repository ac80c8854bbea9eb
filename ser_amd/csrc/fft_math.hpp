// Per-lane complex arithmetic of the wave64 FFT: radix-4 / radix-16 butterflies
// held entirely in registers.  Plain C++ so the same code is unit-tested on the
// host (tests/test_native_host.py builds it with g++) and inlined into the
// gfx950 kernels.
#pragma once

#if defined(__HIPCC__)
#define SERHIP_HD __host__ __device__ __forceinline__
#else
#define SERHIP_HD inline
#endif

namespace serhip {

// two packed floats; on gfx950 the compiler maps +,-,* on this type to v_pk_*_f32
typedef float v2f __attribute__((ext_vector_type(2)));

SERHIP_HD v2f cmul(v2f a, v2f b) {
    return v2f{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
SERHIP_HD v2f mul_neg_i(v2f a) { return v2f{a.y, -a.x}; }  // a * (-i)

// forward 4-point DFT in place: (a,b,c,d) <- (X0,X1,X2,X3), W4 = -i
SERHIP_HD void dft4(v2f& a, v2f& b, v2f& c, v2f& d) {
    const v2f s0 = a + c, d0 = a - c, s1 = b + d, d1 = mul_neg_i(b - d);
    a = s0 + s1;
    b = d0 + d1;
    c = s0 - s1;
    d = d0 - d1;
}

// register index that holds output bin k after fft16(): base-4 digit reversal
#define SERHIP_REV4(k) (4 * ((k) & 3) + ((k) >> 2))

// forward 16-point DFT in place as 4x4 Cooley-Tukey.  Input v[n], n = 4*n1 + n2 in
// natural order; output bin k = k1 + 4*k2 ends up in v[SERHIP_REV4(k)].
SERHIP_HD void fft16(v2f v[16]) {
    const float c1 = 0.92387953251128674f;  // cos(pi/8)
    const float s1 = 0.38268343236508977f;  // sin(pi/8)
    const float r2 = 0.70710678118654752f;  // sqrt(1/2)
    // inner DFTs over n1 for each n2: elements n2, 4+n2, 8+n2, 12+n2 -> T[n2][k1] at v[n2 + 4*k1]
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    // twiddles W16^(n2*k1)
    v[1 + 4 * 1] = cmul(v[1 + 4 * 1], v2f{c1, -s1});   // W16^1
    v[1 + 4 * 2] = cmul(v[1 + 4 * 2], v2f{r2, -r2});   // W16^2
    v[1 + 4 * 3] = cmul(v[1 + 4 * 3], v2f{s1, -c1});   // W16^3
    v[2 + 4 * 1] = cmul(v[2 + 4 * 1], v2f{r2, -r2});   // W16^2
    v[2 + 4 * 2] = mul_neg_i(v[2 + 4 * 2]);            // W16^4
    v[2 + 4 * 3] = cmul(v[2 + 4 * 3], v2f{-r2, -r2});  // W16^6
    v[3 + 4 * 1] = cmul(v[3 + 4 * 1], v2f{s1, -c1});   // W16^3
    v[3 + 4 * 2] = cmul(v[3 + 4 * 2], v2f{-r2, -r2});  // W16^6
    v[3 + 4 * 3] = cmul(v[3 + 4 * 3], v2f{-c1, s1});   // W16^9
    // outer DFTs over n2 for each k1 -> X[k1 + 4*k2] at v[4*k1 + k2]
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}

}  // namespace serhip
