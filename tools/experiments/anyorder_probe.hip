// Probe: what does the barrier bit between two dependent launches cost, and does a flag + write-through protocol carry data from a kernel to the
// next one launched with hipExtAnyOrderLaunch (same stream, no barrier bit: its workgroups are dispatched as soon as the producer's have all been
// dispatched)?   hipcc --offload-arch=gfx950 -O3 -o /tmp/anyorder_probe tools/experiments/anyorder_probe.hip && /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_prod(float *data, unsigned *flag, unsigned epoch, int spin, int wt) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    float v = (float)epoch;
    for (int i = 0; i < spin; i++) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);       // ~spin x 4 cycles of dependent work
    v = (float)(epoch * 4 + (wg & 3)) + (v > 1e30f ? 1.f : 0.f);
    float *dst = data + (long)wg * 256 + tid;
    if (wt) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(dst), "v"(v) : "memory");
    else    *dst = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag + wg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(256) k_cons(const float *data, const unsigned *flag, unsigned epoch, int *bad, float *out, int wt, int poll) {
    const int wg = blockIdx.x, tid = threadIdx.x, src = (wg + 37) & 255;
    if (poll) {
        if (tid == 0) {
            int it = 0;
            while (__hip_atomic_load(flag + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) { __builtin_amdgcn_s_sleep(1); if (++it > (1 << 22)) { atomicAdd(bad + 1, 1); break; } }
        }
        __syncthreads();
    }
    float v;
    const float *p = data + (long)src * 256 + tid;
    if (wt) { asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
    else v = *p;
    if (v != (float)(epoch * 4 + (src & 3))) atomicAdd(bad, 1);
    out[(long)wg * 256 + tid] = v;
}
__global__ void k_empty(int *p) { if (p && threadIdx.x == 999) *p = 1; }

int main() {
    float *data, *out; unsigned *flag; int *bad;
    CK(hipMalloc(&data, 256 * 256 * 4)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&flag, 1024)); CK(hipMalloc(&bad, 8));
    CK(hipMemset(flag, 0, 1024)); CK(hipMemset(bad, 0, 8)); CK(hipMemset(data, 0, 256 * 256 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    // 1. empty launches, normal vs any-order
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipStreamSynchronize(s));
            auto t0 = now();
            for (int i = 0; i < 2000; i++) hipExtLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, (int *)nullptr);
            CK(hipStreamSynchronize(s));
            printf("empty x2000 %s: %.2f us per launch\n", mode ? "any-order" : "normal   ", us(t0, now()) / 2000);
        }
    }
    // 2. producer -> consumer pairs
    unsigned epoch = 1;
    for (int spin = 0; spin <= 2000; spin += 1000)
    for (int mode = 0; mode < 3; mode++) {       // 0: both normal, plain stores / loads, no poll (the barrier bit carries the data); 1: normal launches + flag protocol; 2: consumer any-order + flag protocol
        const int wt = mode >= 1, poll = mode >= 1;
        CK(hipMemset(bad, 0, 8));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipStreamSynchronize(s));
            auto t0 = now();
            for (int i = 0; i < 1000; i++, epoch++) {
                hipExtLaunchKernelGGL(k_prod, dim3(256), dim3(256), 0, s, nullptr, nullptr, 0, data, flag, epoch, spin, wt);
                hipExtLaunchKernelGGL(k_cons, dim3(256), dim3(256), 0, s, nullptr, nullptr, mode == 2 ? hipExtAnyOrderLaunch : 0, (const float *)data, (const unsigned *)flag, epoch, bad, out, wt, poll);
            }
            CK(hipStreamSynchronize(s));
            int hb[2]; CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
            printf("spin %4d mode %d: %.2f us per pair, bad %d timeouts %d\n", spin, mode, us(t0, now()) / 1000, hb[0], hb[1]);
        }
    }
    return 0;
}
