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

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 forms: the packed-f32 source modifiers (op_sel / neg_lo / neg_hi) give a complex multiply in
// two instructions and a +-i rotation folded into an add, which the compiler does not find by itself
// (it builds the swizzled, half-negated operand with v_xor + v_mov first).  The two instructions of a
// complex multiply sit in ONE asm statement: around every asm statement the compiler assumes the worst
// hazard and pads with s_nop when a neighbour touches the same registers.
__device__ __forceinline__ v2f cmul(v2f a, v2f b) {
    v2f t;   // (-a.y*b.y, a.y*b.x), then a.x * b + t
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]"
        : "=&v"(t) : "v"(a), "v"(b));
    return t;
}
// b wave-uniform (a compile-time twiddle): kept in a scalar register pair, no VGPRs spent on constants
__device__ __forceinline__ v2f cmul_uniform(v2f a, v2f b) {
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]"
        : "=&v"(t) : "v"(a), "s"(b));
    return t;
}
__device__ __forceinline__ v2f add_neg_i(v2f a, v2f b) {   // a + (-i) b = (a.x + b.y, a.y - b.x)
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f sub_neg_i(v2f a, v2f b) {   // a - (-i) b = (a.x - b.y, a.y + b.x)
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// the two sums of the real-FFT unpack, one instruction each
__device__ __forceinline__ v2f add_conj(v2f a, v2f b) {    // a + conj(b) = (a.x + b.x, a.y - b.y)
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f sub_conj_neg_i(v2f a, v2f b) {   // -i (a - conj(b)) = (a.y + b.y, b.x - a.x)
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
SERHIP_HD v2f add_conj(v2f a, v2f b) { return v2f{a.x + b.x, a.y - b.y}; }
SERHIP_HD v2f sub_conj_neg_i(v2f a, v2f b) { return v2f{a.y + b.y, b.x - a.x}; }
SERHIP_HD v2f cmul(v2f a, v2f b) {
    return v2f{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
SERHIP_HD v2f cmul_uniform(v2f a, v2f b) { return cmul(a, b); }
SERHIP_HD v2f add_neg_i(v2f a, v2f b) { return v2f{a.x + b.y, a.y - b.x}; }
SERHIP_HD v2f sub_neg_i(v2f a, v2f b) { return v2f{a.x - b.y, a.y + b.x}; }
#endif
SERHIP_HD v2f mul_neg_i(v2f a) { return v2f{a.y, -a.x}; }  // a * (-i)

// forward 4-point DFT in place: (a,b,c,d) <- (X0,X1,X2,X3), W4 = -i
SERHIP_HD void dft4(v2f& a, v2f& b, v2f& c, v2f& d) {
    const v2f s0 = a + c, d0 = a - c, s1 = b + d, t = b - d;
    a = s0 + s1;
    b = add_neg_i(d0, t);
    c = s0 - s1;
    d = sub_neg_i(d0, t);
}
// the same with input c standing for (-i) * c (the W16^4 twiddle folded into the first two sums)
SERHIP_HD void dft4_crot(v2f& a, v2f& b, v2f& c, v2f& d) {
    const v2f s0 = add_neg_i(a, c), d0 = sub_neg_i(a, c), s1 = b + d, t = b - d;
    a = s0 + s1;
    b = add_neg_i(d0, t);
    c = s0 - s1;
    d = sub_neg_i(d0, t);
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
    v[1 + 4 * 1] = cmul_uniform(v[1 + 4 * 1], v2f{c1, -s1});   // W16^1
    v[1 + 4 * 2] = cmul_uniform(v[1 + 4 * 2], v2f{r2, -r2});   // W16^2
    v[1 + 4 * 3] = cmul_uniform(v[1 + 4 * 3], v2f{s1, -c1});   // W16^3
    v[2 + 4 * 1] = cmul_uniform(v[2 + 4 * 1], v2f{r2, -r2});   // W16^2
    // v[2 + 4*2] takes W16^4 = -i inside dft4_crot below
    v[2 + 4 * 3] = cmul_uniform(v[2 + 4 * 3], v2f{-r2, -r2});  // W16^6
    v[3 + 4 * 1] = cmul_uniform(v[3 + 4 * 1], v2f{s1, -c1});   // W16^3
    v[3 + 4 * 2] = cmul_uniform(v[3 + 4 * 2], v2f{-r2, -r2});  // W16^6
    v[3 + 4 * 3] = cmul_uniform(v[3 + 4 * 3], v2f{-c1, s1});   // W16^9
    // outer DFTs over n2 for each k1 -> X[k1 + 4*k2] at v[4*k1 + k2]
    dft4(v[0], v[1], v[2], v[3]);
    dft4(v[4], v[5], v[6], v[7]);
    dft4_crot(v[8], v[9], v[10], v[11]);
    dft4(v[12], v[13], v[14], v[15]);
}

}  // namespace serhip
