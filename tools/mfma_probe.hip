// micro-probe: fp32 MFMA issue rate, LDS read cost and kernel boundary on this GPU
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS, int LDSR>
__global__ void __launch_bounds__(256) k_probe(float *out, int iters) {
    __shared__ float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = 1.0f;
    __syncthreads();
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    const float *p = sm + (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (LDSR) { const float4 t = *reinterpret_cast<const float4 *>(p + ((it * 8 + j) & 15) * 256); a = t.x; b = t.y; }
            acc[j % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j % CHAINS], 0, 0, 0);
        }
    }
    float s = 0; for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void k_empty(float *o) { if (threadIdx.x == 9999) o[0] = 1; }
hipStream_t g_s = 0;
template <typename F> float timeit(F f, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) f();
    hipEventRecord(e0, g_s); for (int i = 0; i < n; i++) f(); hipEventRecord(e1, g_s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n;
}
int main() {
    float *out; hipMalloc(&out, 1 << 22);
    const int iters = 64;   // 512 MFMAs per wave
    for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) { hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking); printf("--- non-blocking stream ---\n"); }
    printf("empty kernel boundary: %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, g_s, out); }, 200));
    for (int wg : {256, 512}) {
        printf("wg=%d 1 chain  noLDS: %.2f us\n", wg, timeit([&] { hipLaunchKernelGGL((k_probe<1, 0>), dim3(wg), dim3(256), 0, g_s, out, iters); }, 100));
        printf("wg=%d 2 chains noLDS: %.2f us\n", wg, timeit([&] { hipLaunchKernelGGL((k_probe<2, 0>), dim3(wg), dim3(256), 0, g_s, out, iters); }, 100));
        printf("wg=%d 4 chains noLDS: %.2f us\n", wg, timeit([&] { hipLaunchKernelGGL((k_probe<4, 0>), dim3(wg), dim3(256), 0, g_s, out, iters); }, 100));
        printf("wg=%d 1 chain  LDS:   %.2f us\n", wg, timeit([&] { hipLaunchKernelGGL((k_probe<1, 1>), dim3(wg), dim3(256), 0, g_s, out, iters); }, 100));
        printf("wg=%d 2 chains LDS:   %.2f us\n", wg, timeit([&] { hipLaunchKernelGGL((k_probe<2, 1>), dim3(wg), dim3(256), 0, g_s, out, iters); }, 100));
    }
    }
    // long run to read the sustained clock: 4 chains, 256 WGs, 64x more work
    float us = timeit([&] { hipLaunchKernelGGL((k_probe<4, 0>), dim3(256), dim3(256), 0, g_s, out, iters * 64); }, 10);
    printf("long run: %.1f us for %d MFMAs/wave -> %.1f cycles@2.4GHz per MFMA, %.1f TFLOP/s\n", us, iters * 64 * 8, us * 2400.0 / (iters * 64 * 8),
           2.0 * 32 * 32 * 2 * (double)iters * 64 * 8 * 256 * 4 / (us * 1e-6) / 1e12);
    return 0;
}
