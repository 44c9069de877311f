// Does cold instruction fetch bound short single-pass kernels?  Same number of executed s_nop per wave, as a loop (256 B of code) and as
// straight-line code (16 KiB / 64 KiB), 256 workgroups x 256 threads, launches alternating with a different kernel.
//   hipcc --offload-arch=gfx950 -O2 -o icache_probe icache_probe.hip && ./icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP_(n) ".rept " #n "\n s_nop 0\n .endr\n"
template <int MODE> __global__ void __launch_bounds__(256) k_nops(float *o, int iters) {
    if (MODE == 0) { for (int i = 0; i < iters; i++) asm volatile(REP_(64) ::: "memory"); }            // iters x 64 nops, 256 B of code
    if (MODE == 1) { asm volatile(REP_(4096) ::: "memory"); }                                           // 16 KiB straight
    if (MODE == 2) { asm volatile(REP_(16384) ::: "memory"); }                                          // 64 KiB straight
    if (MODE == 3) { for (int i = 0; i < iters; i++) asm volatile(REP_(1024) ::: "memory"); }           // 4 KiB body looped
    if (threadIdx.x == 0 && o) o[blockIdx.x] = 1.f;
}
__global__ void k_other(float *o) { if (threadIdx.x == 0 && o) o[blockIdx.x] = 2.f; }
template <int MODE> float run(float *o, int iters, int wgs, bool alternate) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) { k_nops<MODE><<<wgs, 256>>>(o, iters); if (alternate) k_other<<<wgs, 64>>>(o); }
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 200; i++) { k_nops<MODE><<<wgs, 256>>>(o, iters); if (alternate) k_other<<<wgs, 64>>>(o); }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 200 * 1000;
}
int main() {
    float *o; hipMalloc(&o, 4096 * 4);
    float base = 0;
    { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); for (int i = 0; i < 20; i++) k_other<<<256, 64>>>(o); hipDeviceSynchronize();
      hipEventRecord(a); for (int i = 0; i < 200; i++) k_other<<<256, 64>>>(o); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&base, a, b); base = base / 200 * 1000; }
    printf("k_other alone: %.2f us per launch\n", base);
    for (int wgs : {256, 1024}) for (int alt = 0; alt < 2; alt++) {
        printf("wgs %4d alternate %d (pair time incl. k_other when alternate): loop 4096 nops %.2f us | straight 4096 (16 KiB) %.2f us | loop 16384 %.2f us | straight 16384 (64 KiB) %.2f us | 4 KiB body x4 %.2f us x16 %.2f us\n",
               wgs, alt, run<0>(o, 64, wgs, alt), run<1>(o, 0, wgs, alt), run<0>(o, 256, wgs, alt), run<2>(o, 0, wgs, alt), run<3>(o, 4, wgs, alt), run<3>(o, 16, wgs, alt));
    }
    return 0;
}
