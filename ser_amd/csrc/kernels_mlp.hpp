// The fast profile's classifier on the device: sklearn
// Pipeline(StandardScaler, MLPClassifier) forward pass as built by
// ser/_internal/models/training_support.py:87-106 and used by
// ser/_internal/models/fast_path.py:180-192 (model.predict / model.predict_proba).
// Everything is float64, like sklearn on the float64 feature matrix the reference hands it.
//
//   k_mlp_scale    (x - mean) / scale per column (StandardScaler.transform), f32 or f64 input -> f64
//   k_mlp_dense    out = act(in @ W + b); one workgroup per 8 rows, the rows in LDS, thread j walks
//                  column j of W (coalesced across the workgroup)
//   k_mlp_output   softmax (multiclass) or logistic (binary) of the last layer, arg-max label index
#pragma once

#include <hip/hip_runtime.h>

namespace serhip {

constexpr int kMlpRows = 8;   // rows per workgroup in k_mlp_dense

template <typename T>
__global__ __launch_bounds__(256) void k_mlp_scale(const T* __restrict__ x, long long n_rows, int dim,
                                                   const double* __restrict__ mean, const double* __restrict__ scale,
                                                   double* __restrict__ out) {
#pragma clang fp contract(off)
    const long long total = n_rows * dim;
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = static_cast<int>(i % dim);
        double v = static_cast<double>(x[i]);
        if (mean) v -= mean[c];
        if (scale) v /= scale[c];
        out[i] = v;
    }
}

// activation codes: 0 identity, 1 relu, 2 tanh, 3 logistic
__device__ __forceinline__ double mlp_activate(double v, int act) {
    if (act == 1) return v > 0.0 ? v : 0.0;
    if (act == 2) return tanh(v);
    if (act == 3) return 1.0 / (1.0 + exp(-v));
    return v;
}

__global__ __launch_bounds__(256) void k_mlp_dense(const double* __restrict__ in, long long n_rows, int n_in, int n_out,
                                                   const double* __restrict__ w, const double* __restrict__ b, int act,
                                                   double* __restrict__ out) {
    extern __shared__ double s_rows[];   // [kMlpRows][n_in]
    const long long r0 = static_cast<long long>(blockIdx.x) * kMlpRows;
    const int rows = static_cast<int>(min(static_cast<long long>(kMlpRows), n_rows - r0));
    for (int i = threadIdx.x; i < kMlpRows * n_in; i += blockDim.x) {
        const int r = i / n_in;
        s_rows[i] = r < rows ? in[(r0 + r) * n_in + (i - r * n_in)] : 0.0;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n_out; j += blockDim.x) {
        double acc[kMlpRows];
#pragma unroll
        for (int r = 0; r < kMlpRows; ++r) acc[r] = 0.0;
        for (int k = 0; k < n_in; ++k) {
            const double wk = w[static_cast<size_t>(k) * n_out + j];
#pragma unroll
            for (int r = 0; r < kMlpRows; ++r) acc[r] = fma(s_rows[r * n_in + k], wk, acc[r]);
        }
        const double bj = b[j];
#pragma unroll
        for (int r = 0; r < kMlpRows; ++r)
            if (r < rows) out[(r0 + r) * n_out + j] = mlp_activate(acc[r] + bj, act);
    }
}

// out_activation 0: softmax over n_out >= 2 logits -> proba [n_rows, n_out];
//                1: logistic of a single logit -> proba [n_rows, 2] = {1 - p, p}, label = p > 0.5
__global__ __launch_bounds__(256) void k_mlp_output(const double* __restrict__ logits, long long n_rows, int n_out,
                                                    int out_activation, double* __restrict__ proba,
                                                    int* __restrict__ label) {
#pragma clang fp contract(off)
    const long long r = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const double* z = logits + r * n_out;
    if (out_activation == 1) {
        const double p = 1.0 / (1.0 + exp(-z[0]));   // scipy.special.expit
        proba[r * 2] = 1.0 - p;
        proba[r * 2 + 1] = p;
        if (label) label[r] = p > 0.5 ? 1 : 0;
        return;
    }
    double m = z[0];
    for (int j = 1; j < n_out; ++j) m = fmax(m, z[j]);
    double s = 0.0;
    double* p = proba + r * n_out;
    for (int j = 0; j < n_out; ++j) {
        const double e = exp(z[j] - m);
        p[j] = e;
        s += e;
    }
    int best = 0;
    double pb = -1.0;
    for (int j = 0; j < n_out; ++j) {
        const double v = p[j] / s;
        p[j] = v;
        if (v > pb) {   // first maximum, like numpy argmax
            pb = v;
            best = j;
        }
    }
    if (label) label[r] = best;
}

}  // namespace serhip
