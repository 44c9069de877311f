// buf_lds_probe.hip - does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` store ZEROS into its 16 bytes of the LDS destination on gfx950?
// (the conv kernels want that: a tap outside the image = a lane offset past num_records, no zero page, no 64-bit select).  Also: soffset is added to the address
// but is not part of the range check.   hipcc --offload-arch=gfx950 -O3 buf_lds_probe.hip -o /tmp/buf_lds_probe && /tmp/buf_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *X, float *Y, int n, int so) {
    extern __shared__ float lds[];
    const unsigned long a = (unsigned long)X;
    i32x4 r; r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xFFFF); r[2] = 0x7FFFFFFF; r[3] = 0x00020000;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < 4; i++) lds[threadIdx.x * 4 + i] = 7.f;
    __syncthreads();
    const unsigned voff = (lane % 3 != 2 && lane < n) ? lane * 16 : 0x80000000u;
    const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (threadIdx.x >> 6) * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(voff), "s"(r), "s"(la), "s"(so) : "memory");
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = 0; i < 4; i++) Y[threadIdx.x * 4 + i] = lds[threadIdx.x * 4 + i];
}
int main() {
    float *X, *Y; hipMalloc(&X, 1 << 20); hipMalloc(&Y, 4096);
    float h[4096]; for (int i = 0; i < 4096; i++) h[i] = (float)(i + 1);
    hipMemcpy(X, h, sizeof h, hipMemcpyHostToDevice);
    for (int so : {0, 1024}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, X, Y, 60, so);
        float o[256]; hipMemcpy(o, Y, sizeof o, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) {
            const bool valid = l % 3 != 2 && l < 60;
            const float want = valid ? (float)(l * 4 + i + 1 + so / 4) : 0.f;
            if (o[l * 4 + i] != want) { if (bad < 6) printf("  lane %d[%d]: got %g want %g\n", l, i, o[l * 4 + i], want); bad++; }
        }
        printf("soffset %d: %s (%d mismatches)\n", so, bad ? "MISMATCH" : "ok: out-of-range lanes wrote zeros, soffset outside the range check", bad);
    }
    return 0;
}
