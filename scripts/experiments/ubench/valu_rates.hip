// Micro-benchmark: issue cost of packed-f32 vs plain f32 vector instructions on gfx950 at 1, 2 and 4 waves per
// SIMD.  Prints shader cycles per wave-instruction per SIMD (s_memtime deltas, median over waves).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int kKind>
__global__ __launch_bounds__(1024) void k_rate(float* out, unsigned long long* cyc, int iters) {
    v2f a[8];
    float s[16];
    for (int i = 0; i < 8; ++i) a[i] = v2f{1.0f + threadIdx.x * 1e-6f + i, 0.5f};
    for (int i = 0; i < 16; ++i) s[i] = 1.0f + threadIdx.x * 1e-6f + i;
    const v2f m = v2f{1.0000001f, 0.9999999f};
    const v2f c = v2f{1e-9f, -1e-9f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (kKind == 0) {   // 16 plain fma (independent)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(m.x), "v"(c.x));
        } else if (kKind == 1) {   // 8 packed fma = same flops as 16 plain
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (kKind == 2) {   // 8 packed add
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (kKind == 3) {   // 8 packed mul
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        } else if (kKind == 4) {   // 16 plain add
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c.x));
        } else if (kKind == 5) {   // 8 packed fma with op_sel / neg modifiers (the complex-multiply form)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]"
                             : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (kKind == 6) {   // 16 v_sqrt
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[i]));
        } else if (kKind == 7) {   // 16 DPP moves
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(s[i]));
        } else if (kKind == 8) {   // 16 integer adds
            int* q = reinterpret_cast<int*>(s);
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(q[i]) : "v"(it));
        } else if (kKind == 9) {   // 8 plain fma + 4 packed fma interleaved (mixed stream)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[2 * i]) : "v"(m.x), "v"(c.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[2 * i + 1]) : "v"(m.x), "v"(c.x));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.0f;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
    for (int i = 0; i < 16; ++i) acc += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int kKind>
double run(int waves_per_cu, int per_iter, const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL(k_rate<kKind>, dim3(blocks), dim3(64 * waves_per_cu), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k_rate<kKind>, dim3(blocks), dim3(64 * waves_per_cu), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * waves_per_cu);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = static_cast<double>(h[h.size() / 2]);
    const double waves_per_simd = waves_per_cu / 4.0;
    // cycles the SIMD spends per wave-instruction: wave time / (instructions per wave * waves sharing the SIMD)
    const double per_inst = med / (static_cast<double>(iters) * per_iter) / std::max(1.0, waves_per_simd);
    printf("%-34s waves/SIMD %.0f: %7.2f cycles per instruction per SIMD (wave sees %.2f)\n", name, waves_per_simd,
           per_inst, med / (static_cast<double>(iters) * per_iter));
    return per_inst;
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
    for (int w : {4, 8, 16}) {
        run<0>(w, 16, "v_fma_f32 x16", out, cyc);
        run<1>(w, 8, "v_pk_fma_f32 x8", out, cyc);
        run<2>(w, 8, "v_pk_add_f32 x8", out, cyc);
        run<3>(w, 8, "v_pk_mul_f32 x8", out, cyc);
        run<4>(w, 16, "v_add_f32 x16", out, cyc);
        run<5>(w, 8, "v_pk_fma_f32 op_sel/neg x8", out, cyc);
        run<6>(w, 16, "v_sqrt_f32 x16", out, cyc);
        run<7>(w, 16, "v_mov_b32_dpp x16", out, cyc);
        run<8>(w, 16, "v_add_u32 x16", out, cyc);
        run<9>(w, 12, "mixed 4 pk_fma + 8 fma", out, cyc);
        printf("\n");
    }
    return 0;
}
