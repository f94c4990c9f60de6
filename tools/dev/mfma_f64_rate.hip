// Dev microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 vs v_fma_f64 on one CU (cycles per instruction per wave
// and per SIMD), with 1/2/4 independent accumulator chains and 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int CH> __global__ void k_mfma(double *out, long long *cyc, int iters) {
    v4f64 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = v4f64{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH> __global__ void k_fma(double *out, long long *cyc, int iters) {
    double acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = c;
    double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-4;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = fma(acc[c], a, b);
    }
    long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1024);
    const int iters = 2000;
    long long h;
#define RUN(K, CH, threads, label) do { hipLaunchKernelGGL((K<CH>), dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipDeviceSynchronize(); \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%-6s chains %d waves/CU %2d : %7.1f cycles per instr per wave, %7.1f cycles per instr per SIMD\n", label, CH, threads / 64, (double)h / (iters * CH), (double)h / (iters * CH) / ((threads / 64 + 3) / 4)); } while (0)
    for (int threads : {64, 256, 512, 1024}) {
        RUN(k_mfma, 1, threads, "mfma"); RUN(k_mfma, 2, threads, "mfma"); RUN(k_mfma, 4, threads, "mfma");
        RUN(k_fma, 1, threads, "fma"); RUN(k_fma, 4, threads, "fma"); RUN(k_fma, 8, threads, "fma");
    }
    return 0;
}
