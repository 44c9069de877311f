// v_mfma_f32_4x4x1_16B_f32: which A / B lane feeds D[lane][vgpr], and how many cycles it holds the matrix pipe next to v_mfma_f32_16x16x4_f32.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma4x4_probe tools/experiments/mfma4x4_probe.hip && /tmp/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k_layout(float *out, int mode) {
    const int l = threadIdx.x;
    const float a = mode == 0 ? (float)(l + 1) : 1.0f, b = mode == 1 ? (float)(l + 1) : 1.0f;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    for (int v = 0; v < 4; v++) out[l * 4 + v] = acc[v];
}
template <int KIND, int CH>
__global__ void __launch_bounds__(256) k_rate(float *out, int iters) {
    v4f acc[CH];
    for (int c = 0; c < CH; c++) acc[c] = v4f{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
            else           acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        }
    }
    float s = 0; for (int c = 0; c < CH; c++) for (int r = 0; r < 4; r++) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> float timeit(F f, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(e0, 0); for (int i = 0; i < n; i++) f(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n;
}
int main() {
    float *out; hipMalloc(&out, 1 << 22);
    float h[256];
    for (int mode = 0; mode < 2; mode++) {
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out, mode); hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("%s operand lane (0-based) that feeds D[lane][vgpr]:\n", mode == 0 ? "A" : "B");
        for (int l = 0; l < 64; l++) { printf("  lane %2d:", l); for (int v = 0; v < 4; v++) printf(" %2d", (int)h[l * 4 + v] - 1); if (l % 4 == 3) printf("\n"); }
    }
    const int iters = 4096;
    float t;
    t = timeit([&] { hipLaunchKernelGGL((k_rate<0, 4>), dim3(256), dim3(256), 0, 0, out, iters); }, 5);
    printf("4x4x1   4 chains, 1 wave/SIMD: %.1f us -> %.1f cycles@2.4GHz per MFMA\n", t, t * 2400.0 / (iters * 4));
    t = timeit([&] { hipLaunchKernelGGL((k_rate<0, 1>), dim3(256), dim3(256), 0, 0, out, iters); }, 5);
    printf("4x4x1   1 chain (dependent):   %.1f us -> %.1f cycles per MFMA\n", t, t * 2400.0 / iters);
    t = timeit([&] { hipLaunchKernelGGL((k_rate<0, 2>), dim3(256), dim3(256), 0, 0, out, iters); }, 5);
    printf("4x4x1   2 chains:              %.1f us -> %.1f cycles per MFMA\n", t, t * 2400.0 / (iters * 2));
    t = timeit([&] { hipLaunchKernelGGL((k_rate<1, 4>), dim3(256), dim3(256), 0, 0, out, iters); }, 5);
    printf("16x16x4 4 chains, 1 wave/SIMD: %.1f us -> %.1f cycles per MFMA\n", t, t * 2400.0 / (iters * 4));
    t = timeit([&] { hipLaunchKernelGGL((k_rate<0, 4>), dim3(256), dim3(512), 0, 0, out, iters); }, 5);
    printf("4x4x1   4 chains, 2 waves/SIMD: %.1f us -> %.1f cycles per MFMA per SIMD\n", t, t * 2400.0 / (iters * 4 * 2));
    return 0;
}
