// gemm_lab.hip - stand-alone laboratory for the 1024^3 fp32 MFMA GEMM (BASELINE config #2).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o build/gemm_lab && build/gemm_lab
// Every variant is timed like bench.py times the product kernel (non-blocking stream, 1500 warm launches, 5 x 200 launches
// between HIP events, random U(0,1) operands) and carries in-kernel stamps: s_memtime (shader cycles) at kernel entry, after
// the prologue, after the main loop and at exit, plus s_memrealtime (100 MHz) at entry / exit, so that
//   effective clock = d(memtime) / d(memrealtime) * 100 MHz   and   prologue / main loop / epilogue are split in cycles.
// ABL bits switch parts of a kernel OFF (results are then wrong on purpose): 1 = no DMA after stage 0, 2 = no LDS operand
// reads in the main loop, 4 = no barriers in the main loop, 8 = no global stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <type_traits>
#include <dlfcn.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

struct P { const float *A, *B; float *O; int M, N, K; unsigned long long *st; };

__device__ __forceinline__ void tile_of(int b, int tiles_m, int tiles_n, int &tm, int &tn) {
    const int T = tiles_m * tiles_n, q8 = T >> 3, r8 = T & 7, x = b & 7, i = b >> 3;
    const int L = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i;
    constexpr int GROUP_M = 4;
    const int per_group = GROUP_M * tiles_n, grp = L / per_group, first_m = grp * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    tm = first_m + (L % per_group) % gsz; tn = (L % per_group) / gsz;
}
__device__ __forceinline__ void stamp(const P &p, int slot) {
    if (p.st && threadIdx.x == 0) {
        p.st[blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime();
        if (slot == 0) p.st[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();
        if (slot == 3) p.st[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    }
}

__device__ __forceinline__ void dma16p(const float *gsrc, float *lds_dst) {       // per-lane 64-bit source address form
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(LDS_AS void *)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(la) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// V0: the product kernel (csrc/gemm.hip k_gemm_glds8<128, true, false>): 64x64 tile, 8 waves = 2 k-groups x (2x2 waves of
// one 32x32 accumulator pair), 128-deep stages, 2 LDS buffers, LDS-DMA.
template <int ABL, int SPREAD>
__global__ void __launch_bounds__(512) k_v0(P p) {
    constexpr int BM = 64, BN = 64, BK = 128;
    constexpr int NC = BK / 8, CH = BK / 4;
    constexpr int STAGE = (BM + BN) * BK, NI = BK / 4, NJ = NI / 8, NCG = NC / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stamp(p, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = w >> 2, w4 = w & 3, wm = w4 >> 1, wn = w4 & 1, h = lane >> 5, l31 = lane & 31;
    const int c0 = kg * NCG;
    const int M = p.M, N = p.N, K = p.K;
    int tm, tn; tile_of(blockIdx.x, M / BM, N / BN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN, nst = K / BK;
    const float *srcA[NJ], *srcB[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int i = w * NJ + j;
        { const int r = i * (256 / BK) + lane / CH, ql = lane % CH, q = ql ^ (r & (CH - 1)); srcA[j] = p.A + (long)(m0 + r) * K + q * 4; }
        { const int kk = i * 4 + lane / 16, ch = lane % 16; srcB[j] = p.B + (long)kk * N + n0 + ch * 4; }
    }
    const long stepA = BK, stepB = (long)BK * N;
    auto issue1 = [&](int kt, int buf, int j) __attribute__((always_inline)) {
        float *base = lds + buf * STAGE + (w * NJ) * 256;
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(srcA[j] + kt * stepA), (LDS_AS void *)(base + j * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(srcB[j] + kt * stepB), (LDS_AS void *)(base + BM * BK + j * 256), 16, 0, 0);
    };
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJ; j++) issue1(kt, buf, j);
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int ra_ = wm * 32 + l31, rb_ = wn * 32 + l31;
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        if (ABL & 2) { if (ci != c0) return; }
        const v4f t = *reinterpret_cast<const v4f *>(a + ra_ * BK + (((ci * 2 + h) ^ (ra_ & (CH - 1))) << 2));
        av[0] = t[0]; av[1] = t[1]; av[2] = t[2]; av[3] = t[3];
#pragma unroll
        for (int j = 0; j < 4; j++) bv[j] = b[(ci * 8 + 4 * h + j) * BN + rb_];
    };
    auto mm = [&](float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], acc1, 0, 0, 0);
    };
    auto wait_all = [&]() __attribute__((always_inline)) {
        if (ABL & 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return; }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stamp(p, 1);
    float ca[4], cb[4];
    rd(lds, lds + BM * BK, c0, ca, cb);
    int buf = 0;
    for (int kt = 0; kt < nst; kt++) {
        const int b1 = buf ^ 1;
        const bool more = kt + 1 < nst && !(ABL & 1);
        if (!SPREAD && more) issue(kt + 1, b1);
        const float *a = lds + buf * STAGE, *b = a + BM * BK;
#pragma unroll
        for (int ci = 0; ci + 1 < NCG; ci++) {
            float na[4], nbv[4];
            rd(a, b, c0 + ci + 1, na, nbv);
            if (SPREAD && more && ci < NJ) issue1(kt + 1, b1, ci);
            __builtin_amdgcn_sched_barrier(0);
            mm(ca, cb);
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 2)) {
#pragma unroll
                for (int j = 0; j < 4; j++) { ca[j] = na[j]; cb[j] = nbv[j]; }
            }
        }
        wait_all();
        float na[4], nbv[4];
        if (kt + 1 < nst) rd(lds + b1 * STAGE, lds + b1 * STAGE + BM * BK, c0, na, nbv);
        __builtin_amdgcn_sched_barrier(0);
        mm(ca, cb);
        if (kt + 1 < nst && !(ABL & 2)) {
#pragma unroll
            for (int j = 0; j < 4; j++) { ca[j] = na[j]; cb[j] = nbv[j]; }
        }
        buf = b1;
    }
    stamp(p, 2);
    const int gn = n0 + wn * 32 + l31;
    float add[16];
    __syncthreads();
    if (kg == 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) lds[(w4 * 16 + r) * 64 + lane] = acc0[r] + acc1[r];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int r = 0; r < 16; r++) add[r] = lds[(w4 * 16 + r) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = (acc0[r] + acc1[r]) + add[r];
        if (!(ABL & 8) || v == 12345.678f) p.O[(long)gm * N + gn] = v;
    }
    stamp(p, 3);
}

// ---------------------------------------------------------------------------------------------------------------------
// V1: every wave owns the WHOLE 64x64 tile (4 accumulators = 4 independent MFMA chains, one LDS read feeds two MFMAs) over
// its share of each stage's 8-deep k chunks (chunk ci belongs to wave ci % NW); partial tiles meet in LDS at the end.
// GU > 0: the DMA pieces of a later stage are issued between the first GU MFMA groups (4 MFMAs each) of a barrier interval
//         instead of as one burst right after the barrier.
// SUB:    the first stage is consumed in four 32-deep sub-stages (first MFMA after 16 KB instead of 64 KB have landed); the A
//         stage is then stored k-quarter-major [kq][row][32 k] so that each sub-stage is a contiguous run of DMA pieces.
template <int N> __device__ __forceinline__ void wait_vm() {                 // s_waitcnt vmcnt(N) only (gfx9 encoding)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
template <int NW, int ABL, int GU, int SUB>
__global__ void __launch_bounds__(NW * 64) k_v1(P p) {
    constexpr int BM = 64, BN = 64, BK = 128;
    constexpr int NC = BK / 8, CH = BK / 4, NCW = NC / NW;
    constexpr int STAGE = (BM + BN) * BK, NI = BK / 4, NJ = NI / NW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stamp(p, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int M = p.M, N = p.N, K = p.K;
    int tm, tn; tile_of(blockIdx.x, M / BM, N / BN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN, nst = K / BK;
    // DMA piece i of a stage (1 KiB of A and 1 KiB of B).  SUB: wave w issues pieces i = j * NW + w (the pieces of one k quarter come
    // from all waves, in issue order); otherwise pieces w*NJ .. w*NJ+NJ-1.
    const float *srcA[NJ], *srcB[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int i = SUB ? (j * NW + w) : (w * NJ + j);
        if (SUB) { const int kq = i / 8, rb = i % 8, r = rb * 8 + lane / 8, pl = lane % 8, q = kq * 8 + (pl ^ ((r >> 1) & 7));
                   srcA[j] = p.A + (long)(m0 + r) * K + q * 4; }
        else     { const int r = i * 2 + lane / CH, ql = lane % CH, q = ql ^ (r & (CH - 1)); srcA[j] = p.A + (long)(m0 + r) * K + q * 4; }
        { const int kk = i * 4 + lane / 16, ch = lane % 16; srcB[j] = p.B + (long)kk * N + n0 + ch * 4; }
    }
    const long stepA = BK, stepB = (long)BK * N;
    auto issue1 = [&](int kt, int buf, int j) __attribute__((always_inline)) {
        const int i = SUB ? (j * NW + w) : (w * NJ + j);
        float *sa = lds + buf * STAGE + i * 256, *sb = lds + buf * STAGE + BM * BK + i * 256;
        dma16p(srcA[j] + kt * stepA, sa);
        dma16p(srcB[j] + kt * stepB, sb);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    auto a_off = [&](int row, int lc) __attribute__((always_inline)) -> int {          // lc = logical 16-byte chunk 0..31 of the row
        if (SUB) return ((lc >> 3) * BM + row) * 32 + ((((lc & 7) ^ ((row >> 1) & 7))) << 2);
        return row * BK + ((lc ^ (row & (CH - 1))) << 2);
    };
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[2][4], float (&bv)[2][4]) __attribute__((always_inline)) {
        if (ABL & 2) { if (ci >= NW) return; }
#pragma unroll
        for (int mi = 0; mi < 2; mi++) {
            const int row = mi * 32 + l31;
            const v4f t = *reinterpret_cast<const v4f *>(a + a_off(row, ci * 2 + h));
            av[mi][0] = t[0]; av[mi][1] = t[1]; av[mi][2] = t[2]; av[mi][3] = t[3];
        }
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int j = 0; j < 4; j++) bv[ni][j] = b[(ci * 8 + 4 * h + j) * BN + ni * 32 + l31];
    };
    // one chunk = 4 groups of 4 MFMAs; group g of the barrier interval (g = 4 * position of the chunk + j) carries the DMA pieces jj
    // of stage `kt1` with jj * GU / NJ == g
    auto mm = [&](float (&av)[2][4], float (&bv)[2][4], bool dma, int kt1, int b1, int pos) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][j], bv[0][j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][j], bv[1][j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][j], bv[0][j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][j], bv[1][j], acc[1][1], 0, 0, 0);
            if (GU > 0 && dma) {
#pragma unroll
                for (int jj = 0; jj < NJ; jj++) if (jj * GU / NJ == pos * 4 + j) issue1(kt1, b1, jj);
            }
        }
    };
    auto barrier_all = [&]() __attribute__((always_inline)) {
        if (ABL & 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return; }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    float ca[2][4], cb[2][4], na[2][4], nb[2][4];
    auto take = [&]() __attribute__((always_inline)) {
        if (ABL & 2) return;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) { ca[i][j] = na[i][j]; cb[i][j] = nb[i][j]; }
    };
#pragma unroll
    for (int j = 0; j < NJ; j++) issue1(0, 0, j);
    if (nst > 1) {                                          // stage 1 goes out in the prologue as well (nothing to hide it under yet)
#pragma unroll
        for (int j = 0; j < NJ; j++) issue1(1, 1, j);
    }
    int s0 = 0;
    if (SUB) {
        constexpr int PQ = 8 / NW;                          // pieces (A + B pairs) per wave per k quarter
        static_assert(NJ == 4 * PQ, "sub-stage bookkeeping");
        const float *a = lds, *b = lds + BM * BK;
        auto quarter = [&](int q) __attribute__((always_inline)) {
            if (NW == 4 || w < 4) { rd(a, b, 4 * q + (w & 3), ca, cb); mm(ca, cb, false, 0, 0, 0); }
        };
        // outstanding loads allowed once quarter q has landed: (3 - q) * 2 * PQ of stage 0 + 2 * NJ of stage 1 (K >= 256 assumed here)
        wait_vm<3 * 2 * PQ + 2 * NJ>(); asm volatile("s_barrier" ::: "memory"); stamp(p, 1); quarter(0);
        wait_vm<2 * 2 * PQ + 2 * NJ>(); asm volatile("s_barrier" ::: "memory"); quarter(1);
        wait_vm<1 * 2 * PQ + 2 * NJ>(); asm volatile("s_barrier" ::: "memory"); quarter(2);
        wait_vm<0 * 2 * PQ + 2 * NJ>(); asm volatile("s_barrier" ::: "memory"); quarter(3);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                       // B(1): stage 1 landed, everybody is done with buffer 0
        if (nst > 2 && !(ABL & 1)) {                        // stage 2 into buffer 0 (the loop only continues DMAs it finds started)
#pragma unroll
            for (int j = 0; j < NJ; j++) issue1(2, 0, j);
        }
        rd(lds + STAGE, lds + STAGE + BM * BK, w, ca, cb);
        s0 = 1;
    } else {
        if (nst > 1) wait_vm<2 * NJ>(); else wait_vm<0>();              // stage 0 landed, stage 1 may fly
        asm volatile("s_barrier" ::: "memory");
        stamp(p, 1);
        rd(lds, lds + BM * BK, w, ca, cb);
    }
    // interval s (after barrier B(s)): [rd(s,1) mm(s,0)] ... [rd(s,NCW-1) mm(s,NCW-2)]  B(s+1)  rd(s+1,0) mm(s,NCW-1)
    // DMA of stage s+2 (into stage s's buffer, free after B(s+1)) starts with that last mm and continues in interval s+1's loop
    for (int s = s0; s < nst; s++) {
        const int buf = s & 1;
        const float *a = lds + buf * STAGE, *b = a + BM * BK;
        const bool dma_in = GU > 0 && s >= 1 + s0 && s + 1 < nst && !(ABL & 1);     // stage s+1 is still being issued (started at the end of interval s-1)
#pragma unroll
        for (int t = 0; t + 1 < NCW; t++) {
            rd(a, b, w + NW * (t + 1), na, nb);
            __builtin_amdgcn_sched_barrier(0);
            mm(ca, cb, dma_in, s + 1, buf ^ 1, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            take();
        }
        if (s + 1 < nst) barrier_all();
        const bool dma_out = s + 2 < nst && !(ABL & 1);
        if (GU == 0 && dma_out) {
#pragma unroll
            for (int j = 0; j < NJ; j++) issue1(s + 2, buf, j);
        }
        if (s + 1 < nst) rd(lds + (buf ^ 1) * STAGE, lds + (buf ^ 1) * STAGE + BM * BK, w, na, nb);
        __builtin_amdgcn_sched_barrier(0);
        mm(ca, cb, dma_out, s + 2, buf, 0);
        if (s + 1 < nst) take();
    }
    stamp(p, 2);
    // partial tiles -> LDS [wave][reg][lane]; wave w then sums registers [w*64/NW, (w+1)*64/NW) over the waves in wave order
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) lds[((w * 64) + (i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
    constexpr int RPW = 64 / NW;
#pragma unroll
    for (int q = 0; q < RPW; q++) {
        const int R = w * RPW + q, ab = R >> 4, r = R & 15, mi = ab >> 1, ni = ab & 1;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ww++) v += lds[((ww * 64) + R) * 64 + lane];
        const int gm = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, gn = n0 + ni * 32 + l31;
        if (!(ABL & 8) || v == 12345.678f) p.O[(long)gm * N + gn] = v;
    }
    stamp(p, 3);
}


// ---------------------------------------------------------------------------------------------------------------------
// V2: V0's tiling (8 waves = 2 k-groups x 2x2 waves, one 32x32 accumulator pair per wave) with
//   * LDS-DMA issued from inline asm (saddr form: scalar base + fixed 32-bit lane offset, no VALU address math).  The builtin
//     makes hipcc treat the load as "flat, may touch LDS": while one is pending every LDS dependency becomes lgkmcnt(0), i.e.
//     the just-issued prefetch reads are waited for at once (seen in V0's ISA: a full LDS round trip exposed every 8 MFMAs).
//     With the DMA invisible to the compiler it emits counted lgkmcnt(N) ladders; the DMA's own completion is waited for by hand.
//   * MFMA order pinned (acc0, acc1, acc0, acc1): hipcc otherwise pairs the MFMAs of one accumulator back to back.
//   * ILV: a k-group owns chunk pairs (2 of every 4 chunks) instead of one half of the stage, which lets
//   * SUB: the first stage be consumed in four 32-deep sub-stages (first MFMA after 16 KB instead of 64 KB have landed).
__device__ __forceinline__ void dma16(unsigned voff, const float *base, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(base), "s"(lds_byte) : "memory");
}
template <int ABL, int ILV, int SUB, int PIN>
__global__ void __launch_bounds__(512) k_v2(P p) {
    constexpr int BM = 64, BN = 64, BK = 128;
    constexpr int NC = BK / 8, CH = BK / 4;
    constexpr int STAGE = (BM + BN) * BK, NI = BK / 4, NJ = NI / 8, NCG = NC / 2;
    static_assert(!SUB || ILV, "sub-stages need interleaved chunk ownership");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stamp(p, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = w >> 2, w4 = w & 3, wm = w4 >> 1, wn = w4 & 1, h = lane >> 5, l31 = lane & 31;
    const int M = p.M, N = p.N, K = p.K;
    int tm, tn; tile_of(blockIdx.x, M / BM, N / BN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN, nst = K / BK;
    // chunk t (0..7) of this k-group inside a stage
    auto chunk_of = [&](int t) __attribute__((always_inline)) -> int { return ILV ? ((t >> 1) * 4 + 2 * kg + (t & 1)) : (kg * NCG + t); };
    // DMA pieces: SUB -> wave w issues pieces i = j*8 + w (A stored k-quarter-major), else pieces w*NJ + j (A row-major)
    unsigned voffA[NJ], voffB[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int i = SUB ? (j * 8 + w) : (w * NJ + j);
        if (SUB) { const int kq = i / 8, rb = i % 8, r = rb * 8 + lane / 8, pl = lane % 8, q = kq * 8 + (pl ^ ((r >> 1) & 7));
                   voffA[j] = (unsigned)(((m0 + r) * K + q * 4) * 4); }
        else     { const int r = i * 2 + lane / CH, ql = lane % CH, q = ql ^ (r & (CH - 1)); voffA[j] = (unsigned)(((m0 + r) * K + q * 4) * 4); }
        { const int kk = i * 4 + lane / 16, ch = lane % 16; voffB[j] = (unsigned)((kk * N + n0 + ch * 4) * 4); }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS void *)lds;
    auto issue1 = [&](int kt, int buf, int j) __attribute__((always_inline)) {
        const int i = SUB ? (j * 8 + w) : (w * NJ + j);
        const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + i * 256) * 4));
        dma16(voffA[j], p.A + (long)kt * BK, la);
        dma16(voffB[j], p.B + (long)kt * BK * N, la + BM * BK * 4);
    };
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJ; j++) issue1(kt, buf, j);
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int ra_ = wm * 32 + l31, rb_ = wn * 32 + l31;
    auto a_off = [&](int row, int lc) __attribute__((always_inline)) -> int {
        if (SUB) return ((lc >> 3) * BM + row) * 32 + ((((lc & 7) ^ ((row >> 1) & 7))) << 2);
        return row * BK + ((lc ^ (row & (CH - 1))) << 2);
    };
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        const v4f t = *reinterpret_cast<const v4f *>(a + a_off(ra_, ci * 2 + h));
        av[0] = t[0]; av[1] = t[1]; av[2] = t[2]; av[3] = t[3];
#pragma unroll
        for (int j = 0; j < 4; j++) bv[j] = b[(ci * 8 + 4 * h + j) * BN + rb_];
    };
    auto mm = [&](float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc0, 0, 0, 0);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc1, 0, 0, 0);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], acc0, 0, 0, 0);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], acc1, 0, 0, 0);
    };
    unsigned long long bw = 0, ww = 0, tl0 = 0;
    auto wait_all = [&]() __attribute__((always_inline)) {
        if (ABL & 4) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); return; }
        if (ABL & 16) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_barrier" ::: "memory");
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            ww += t1 - t0; bw += t2 - t1;
            return;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    float ca[4], cb[4];
    int kt0 = 0, buf = 0;
    issue(0, 0);
    if (SUB) {
        if (nst > 1) issue(1, 1);
        const float *a = lds, *b = lds + BM * BK;
        // wave w's loads in issue order: stage 0 pieces j = 0..3 (quarter j), 2 loads each, then 8 loads of stage 1
        auto quarter = [&](int q) __attribute__((always_inline)) {
            float qa[4], qb[4];
            rd(a, b, 4 * q + 2 * kg, ca, cb); rd(a, b, 4 * q + 2 * kg + 1, qa, qb);
            __builtin_amdgcn_sched_barrier(0);
            mm(ca, cb); mm(qa, qb);
        };
        asm volatile("s_waitcnt vmcnt(14)\n\ts_barrier" ::: "memory"); stamp(p, 1); quarter(0);
        asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory"); quarter(1);
        asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory"); quarter(2);
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); quarter(3);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        kt0 = 1; buf = 1;
        rd(lds + STAGE, lds + STAGE + BM * BK, chunk_of(0), ca, cb);
    } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stamp(p, 1);
        rd(lds, lds + BM * BK, chunk_of(0), ca, cb);
    }
    if (ABL & 16) tl0 = __builtin_amdgcn_s_memtime();
    for (int kt = kt0; kt < nst; kt++) {
        const int b1 = buf ^ 1;
        const bool more = kt + 1 < nst && !(ABL & 1);
        if (more) issue(kt + 1, b1);
        const float *a = lds + buf * STAGE, *b = a + BM * BK;
#pragma unroll
        for (int ci = 0; ci + 1 < NCG; ci++) {
            float na[4], nbv[4];
            rd(a, b, chunk_of(ci + 1), na, nbv);
            __builtin_amdgcn_sched_barrier(0);
            mm(ca, cb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; j++) { ca[j] = na[j]; cb[j] = nbv[j]; }
        }
        wait_all();
        float na[4], nbv[4];
        if (kt + 1 < nst) rd(lds + b1 * STAGE, lds + b1 * STAGE + BM * BK, chunk_of(0), na, nbv);
        __builtin_amdgcn_sched_barrier(0);
        mm(ca, cb);
        if (kt + 1 < nst) {
#pragma unroll
            for (int j = 0; j < 4; j++) { ca[j] = na[j]; cb[j] = nbv[j]; }
        }
        buf = b1;
    }
    if ((ABL & 16) && p.st && lane == 0 && blockIdx.x < 8) {      // per-wave: loop cycles, barrier wait, counter wait
        unsigned long long *d = p.st + 2048 + (blockIdx.x * 8 + w) * 4;
        d[0] = __builtin_amdgcn_s_memtime() - tl0; d[1] = bw; d[2] = ww; d[3] = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));   // HW_ID
    }
    stamp(p, 2);
    const int gn = n0 + wn * 32 + l31;
    float add[16];
    __syncthreads();
    if (kg == 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) lds[(w4 * 16 + r) * 64 + lane] = acc0[r] + acc1[r];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int r = 0; r < 16; r++) add[r] = lds[(w4 * 16 + r) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = (acc0[r] + acc1[r]) + add[r];
        if (!(ABL & 8) || v == 12345.678f) p.O[(long)gm * N + gn] = v;
    }
    stamp(p, 3);
}


// ---------------------------------------------------------------------------------------------------------------------
// V3: V2 generalised: KG k-groups (4 * KG waves per workgroup, each k-group owns 16/KG chunks of every stage), operand reads
// PF chunks ahead of their MFMAs (register ring), the stage barrier placed where the first read of the NEXT stage is due.
template <int KG, int PF, int ABL>
__global__ void __launch_bounds__(256 * KG) k_v3(P p) {
    constexpr int BM = 64, BN = 64, BK = 128, NW = 4 * KG;
    constexpr int NC = BK / 8, CH = BK / 4;
    constexpr int STAGE = (BM + BN) * BK, NI = BK / 4, NJ = NI / NW, NCG = NC / KG;
    constexpr int R = (PF == 1) ? 2 : 4;                     // register ring (NCG % R == 0 keeps every index a compile-time constant)
    static_assert(NCG % R == 0 && PF < R && PF <= NCG, "ring");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stamp(p, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = w >> 2, w4 = w & 3, wm = w4 >> 1, wn = w4 & 1, h = lane >> 5, l31 = lane & 31;
    const int M = p.M, N = p.N, K = p.K;
    int tm, tn; tile_of(blockIdx.x, M / BM, N / BN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN, nst = K / BK;
    unsigned voffA[NJ], voffB[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int i = w * NJ + j;
        { const int r = i * 2 + lane / CH, ql = lane % CH, q = ql ^ (r & (CH - 1)); voffA[j] = (unsigned)(((m0 + r) * K + q * 4) * 4); }
        { const int kk = i * 4 + lane / 16, ch = lane % 16; voffB[j] = (unsigned)((kk * N + n0 + ch * 4) * 4); }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS void *)lds;
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + (w * NJ + j) * 256) * 4));
            dma16(voffA[j], p.A + (long)kt * BK, la);
            dma16(voffB[j], p.B + (long)kt * BK * N, la + BM * BK * 4);
        }
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int ra_ = wm * 32 + l31, rb_ = wn * 32 + l31;
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        const v4f t = *reinterpret_cast<const v4f *>(a + ra_ * BK + (((ci * 2 + h) ^ (ra_ & (CH - 1))) << 2));
        av[0] = t[0]; av[1] = t[1]; av[2] = t[2]; av[3] = t[3];
#pragma unroll
        for (int j = 0; j < 4; j++) bv[j] = b[(ci * 8 + 4 * h + j) * BN + rb_];
    };
    auto mm = [&](float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], acc1, 0, 0, 0);
    };
    float oa[R][4], ob[R][4];
    issue(0, 0);
    if (p.st && tid == 0) p.st[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stamp(p, 1);
    if (nst > 1) issue(1, 1);
#pragma unroll
    for (int c = 0; c < PF; c++) rd(lds, lds + BM * BK, kg * NCG + c, oa[c], ob[c]);
    for (int s = 0; s < nst; s++) {
        const float *a = lds + (s & 1) * STAGE, *b = a + BM * BK;
        const float *a1 = lds + ((s & 1) ^ 1) * STAGE, *b1 = a1 + BM * BK;
#pragma unroll
        for (int c = 0; c < NCG; c++) {
            if (c == NCG - PF) {
                if (s + 1 < nst) {
                    if (ABL & 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
                if (s + 2 < nst && !(ABL & 1)) issue(s + 2, s & 1);
            }
            const int nc = c + PF;
            if (nc < NCG) rd(a, b, kg * NCG + nc, oa[nc % R], ob[nc % R]);
            else if (s + 1 < nst) rd(a1, b1, kg * NCG + nc - NCG, oa[nc % R], ob[nc % R]);
            __builtin_amdgcn_sched_barrier(0);
            mm(oa[c % R], ob[c % R]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    stamp(p, 2);
    const int gn = n0 + wn * 32 + l31;
    __syncthreads();
    if (kg > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) lds[(((kg - 1) * 4 + w4) * 16 + r) * 64 + lane] = acc0[r] + acc1[r];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc0[r] + acc1[r];
#pragma unroll
        for (int g = 1; g < KG; g++) v += lds[(((g - 1) * 4 + w4) * 16 + r) * 64 + lane];
        if (!(ABL & 8) || v == 12345.678f) p.O[(long)gm * N + gn] = v;
    }
    stamp(p, 3);
}


// ---------------------------------------------------------------------------------------------------------------------
// Issue probe: ONE wave per SIMD (256 threads / workgroup), no global memory, no barriers.  A chunk = 4 MFMAs (32x32x2 f32) plus the
// fillers a GEMM chunk carries (1 ds_read_b128, 2 ds_read2st64_b32, 3 v_add) in a chosen placement.  Reports cycles per chunk
// (ideal 256).  PAT: 0 none, 1 fillers clumped before the MFMAs, 2 two fillers after each of M0..M2, 3 one filler after each MFMA
// and two before, 4 clump + s_waitcnt lgkmcnt(3) in front of M0 (the compiler's shape), 5 distributed + the wait in front of M0,
// 6 distributed, wait before M1.   NACC accumulators (2 or 4), used round-robin.
#define F_RD128(v, a) asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a))
#define F_RD2(v, a, o0, o1) asm volatile("ds_read2st64_b32 %0, %1 offset0:" #o0 " offset1:" #o1 : "=v"(v) : "v"(a))
#define F_ADD(x, y) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(y))
#define SB __builtin_amdgcn_sched_barrier(0)
typedef float v2f __attribute__((ext_vector_type(2)));
template <int PAT, int NACC>
__global__ void __launch_bounds__(256) k_issue_probe(P p, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) lds[i] = 1.0f;
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    float a = 1.0f + lane, b = 0.5f;
    unsigned ad0 = lane * 16, ad1 = lane * 4, x0 = 1, x1 = 2, x2 = 3, one = 0;
    v4f r128; v2f r2a, r2b;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define MF(q) acc[(q) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(q) % NACC], 0, 0, 0); SB
    for (int c = 0; c < chunks; c++) {
        SB;
        if (PAT == 0) { MF(0); MF(1); MF(2); MF(3); }
        if (PAT == 1) { F_ADD(x0, one); SB; F_RD128(r128, ad0); SB; F_ADD(x1, one); SB; F_RD2(r2a, ad1, 128, 129); SB; F_RD2(r2b, ad1, 130, 131); SB; F_ADD(x2, one); SB;
                        MF(0); MF(1); MF(2); MF(3); }
        if (PAT == 2) { MF(0); F_ADD(x0, one); SB; F_RD128(r128, ad0); SB; MF(1); F_ADD(x1, one); SB; F_RD2(r2a, ad1, 128, 129); SB; MF(2); F_RD2(r2b, ad1, 130, 131); SB; F_ADD(x2, one); SB; MF(3); }
        if (PAT == 3) { F_ADD(x0, one); SB; F_ADD(x1, one); SB; MF(0); F_RD128(r128, ad0); SB; MF(1); F_RD2(r2a, ad1, 128, 129); SB; MF(2); F_RD2(r2b, ad1, 130, 131); SB; MF(3); F_ADD(x2, one); SB; }
        if (PAT == 4) { F_ADD(x0, one); SB; F_RD128(r128, ad0); SB; F_ADD(x1, one); SB; F_RD2(r2a, ad1, 128, 129); SB; F_RD2(r2b, ad1, 130, 131); SB; F_ADD(x2, one); SB;
                        asm volatile("s_waitcnt lgkmcnt(3)"); SB; MF(0); MF(1); MF(2); MF(3); }
        if (PAT == 5) { asm volatile("s_waitcnt lgkmcnt(3)"); SB; MF(0); F_ADD(x0, one); SB; F_RD128(r128, ad0); SB; MF(1); F_ADD(x1, one); SB; F_RD2(r2a, ad1, 128, 129); SB; MF(2); F_RD2(r2b, ad1, 130, 131); SB; F_ADD(x2, one); SB; MF(3); }
        if (PAT == 6) { MF(0); F_ADD(x0, one); SB; F_RD128(r128, ad0); SB; asm volatile("s_waitcnt lgkmcnt(1)"); SB; MF(1); F_ADD(x1, one); SB; F_RD2(r2a, ad1, 128, 129); SB; MF(2); F_RD2(r2b, ad1, 130, 131); SB; F_ADD(x2, one); SB; MF(3); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float sum = r128[0] + r2a[0] + r2b[0] + (float)(x0 + x1 + x2);
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) sum += acc[q][r];
    if (sum == 12345.678f) p.O[tid] = sum;
    if (p.st && lane == 0 && blockIdx.x < 64) p.st[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}


// ---------------------------------------------------------------------------------------------------------------------
// V4: V2 with an UNEVEN k split.  The two waves of a SIMD do not alternate: the older one (k-group 0) wins every MFMA slot it is ready
// for and the younger one only gets the slots the older one leaves while it issues its reads / waits (measured: waves 0-3 spend
// 29 % of the loop waiting at the stage barrier, waves 4-7 none).  So k-group 0 takes NA of a stage's 16 chunks and k-group 1 the
// rest (about what it gets anyway), and both reach the barrier together.  DB = 1: k-group 1 (which has the time) issues all the DMA.
template <int NA, int DB, int ABL>
__global__ void __launch_bounds__(512) k_v4(P p) {
    constexpr int BM = 64, BN = 64, BK = 128;
    constexpr int NC = BK / 8, CH = BK / 4;
    constexpr int STAGE = (BM + BN) * BK, NI = BK / 4;
    constexpr int NJ = DB ? NI / 4 : NI / 8;               // DMA piece pairs per issuing wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stamp(p, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = w >> 2, w4 = w & 3, wm = w4 >> 1, wn = w4 & 1, h = lane >> 5, l31 = lane & 31;
    const int M = p.M, N = p.N, K = p.K;
    int tm, tn; tile_of(blockIdx.x, M / BM, N / BN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN, nst = K / BK;
    const bool loader = DB ? (kg == 1) : true;
    const int lw = DB ? w4 : w;                             // loader index
    unsigned voffA[NJ], voffB[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int i = lw * NJ + j;
        { const int r = i * 2 + lane / CH, ql = lane % CH, q = ql ^ (r & (CH - 1)); voffA[j] = (unsigned)(((m0 + r) * K + q * 4) * 4); }
        { const int kk = i * 4 + lane / 16, ch = lane % 16; voffB[j] = (unsigned)((kk * N + n0 + ch * 4) * 4); }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS void *)lds;
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        if (!loader) return;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + (lw * NJ + j) * 256) * 4));
            dma16(voffA[j], p.A + (long)kt * BK, la);
            dma16(voffB[j], p.B + (long)kt * BK * N, la + BM * BK * 4);
        }
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int ra_ = wm * 32 + l31, rb_ = wn * 32 + l31;
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        const v4f t = *reinterpret_cast<const v4f *>(a + ra_ * BK + (((ci * 2 + h) ^ (ra_ & (CH - 1))) << 2));
        av[0] = t[0]; av[1] = t[1]; av[2] = t[2]; av[3] = t[3];
#pragma unroll
        for (int j = 0; j < 4; j++) bv[j] = b[(ci * 8 + 4 * h + j) * BN + rb_];
    };
    auto mm = [&](float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], acc1, 0, 0, 0);
    };
    unsigned long long bw = 0, tl0 = 0;
    auto wait_all = [&]() __attribute__((always_inline)) {
        if (ABL & 16) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_barrier" ::: "memory");
            bw += __builtin_amdgcn_s_memtime() - t1;
            return;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // one stage for a k-group owning chunks [C0, C0 + NCH): same software pipeline as V2 (reads one chunk ahead, last chunk's MFMAs after the barrier)
    auto run = [&](auto c0_, auto nch_) __attribute__((always_inline)) {
        constexpr int C0 = decltype(c0_)::value, NCH = decltype(nch_)::value;
        float ca[4], cb[4];
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stamp(p, 1);
        rd(lds, lds + BM * BK, C0, ca, cb);
        if (ABL & 16) tl0 = __builtin_amdgcn_s_memtime();
        int buf = 0;
        for (int kt = 0; kt < nst; kt++) {
            const int b1 = buf ^ 1;
            if (kt + 1 < nst && !(ABL & 1)) issue(kt + 1, b1);
            const float *a = lds + buf * STAGE, *b = a + BM * BK;
#pragma unroll
            for (int ci = 0; ci + 1 < NCH; ci++) {
                float na[4], nbv[4];
                rd(a, b, C0 + ci + 1, na, nbv);
                __builtin_amdgcn_sched_barrier(0);
                mm(ca, cb);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; j++) { ca[j] = na[j]; cb[j] = nbv[j]; }
            }
            wait_all();
            float na[4], nbv[4];
            if (kt + 1 < nst) rd(lds + b1 * STAGE, lds + b1 * STAGE + BM * BK, C0, na, nbv);
            __builtin_amdgcn_sched_barrier(0);
            mm(ca, cb);
            if (kt + 1 < nst) {
#pragma unroll
                for (int j = 0; j < 4; j++) { ca[j] = na[j]; cb[j] = nbv[j]; }
            }
            buf = b1;
        }
    };
    if (kg == 0) run(std::integral_constant<int, 0>{}, std::integral_constant<int, NA>{});
    else         run(std::integral_constant<int, NA>{}, std::integral_constant<int, NC - NA>{});
    if ((ABL & 16) && p.st && lane == 0 && blockIdx.x < 8) {
        unsigned long long *d = p.st + 2048 + (blockIdx.x * 8 + w) * 4;
        d[0] = __builtin_amdgcn_s_memtime() - tl0; d[1] = bw; d[2] = 0; d[3] = 0;
    }
    stamp(p, 2);
    const int gn = n0 + wn * 32 + l31;
    float add[16];
    __syncthreads();
    if (kg == 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) lds[(w4 * 16 + r) * 64 + lane] = acc0[r] + acc1[r];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int r = 0; r < 16; r++) add[r] = lds[(w4 * 16 + r) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = (acc0[r] + acc1[r]) + add[r];
        if (!(ABL & 8) || v == 12345.678f) p.O[(long)gm * N + gn] = v;
    }
    stamp(p, 3);
}


// ---------------------------------------------------------------------------------------------------------------------
// V5: V2 (pinned) cleaned up, with: OPT&1 non-temporal output stores, OPT&2 short prologue (power-of-two tile grid: shifts instead of the
// integer divisions of tile_of; one multiply per operand for the DMA offsets), OPT&4 reads two chunks ahead; PRI: s_setprio scheme
// (1: younger k-group raised for the whole kernel, 2: raised around each MFMA burst, 3: raised around reads + wait, lowered for the burst).
template <int OPT, int PRI>
__global__ void __launch_bounds__(512) k_v5(P p) {
    constexpr int BM = 64, BN = 64, BK = 128;
    constexpr int NC = BK / 8, CH = BK / 4;
    constexpr int STAGE = (BM + BN) * BK, NI = BK / 4, NJ = NI / 8, NCG = NC / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, N = p.N, K = p.K;
    int tm, tn;
    if (OPT & 2) {                                          // tiles_m, tiles_n powers of two, T % 8 == 0 (1024^2: 16 x 16)
        const int tnb = 31 - __builtin_clz(N >> 6), b = blockIdx.x, T = (M >> 6) << tnb;
        const int L = (b & 7) * (T >> 3) + (b >> 3), pg = 2 + tnb, r = L & ((1 << pg) - 1);
        tm = ((L >> pg) << 2) + (r & 3); tn = r >> 2;
    } else tile_of(blockIdx.x, M / BM, N / BN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN, nst = K / BK;
    unsigned voffA[NJ], voffB[NJ];
    {
        const int r0 = w * 8 + (lane >> 5), ql = lane & 31, kk0 = w * 16 + (lane >> 4);
        const unsigned ba = (unsigned)((m0 + r0) * K) * 4u, bb = (unsigned)(kk0 * N + n0 + (lane & 15) * 4) * 4u;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            voffA[j] = ba + (unsigned)(2 * j * K) * 4u + (unsigned)((ql ^ ((r0 + 2 * j) & 31)) << 4);
            voffB[j] = bb + (unsigned)(4 * j * N) * 4u;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS void *)lds;
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + (w * NJ + j) * 256) * 4));
            dma16(voffA[j], p.A + (long)kt * BK, la);
            dma16(voffB[j], p.B + (long)kt * BK * N, la + BM * BK * 4);
        }
    };
    issue(0, 0);
    stamp(p, 0);
    const int kg = w >> 2, w4 = w & 3, wm = w4 >> 1, wn = w4 & 1, h = lane >> 5, l31 = lane & 31;
    if (PRI == 1 && kg == 1) __builtin_amdgcn_s_setprio(1);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int ra_ = wm * 32 + l31, rb_ = wn * 32 + l31;
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        const v4f t = *reinterpret_cast<const v4f *>(a + ra_ * BK + (((ci * 2 + h) ^ (ra_ & (CH - 1))) << 2));
        av[0] = t[0]; av[1] = t[1]; av[2] = t[2]; av[3] = t[3];
#pragma unroll
        for (int j = 0; j < 4; j++) bv[j] = b[(ci * 8 + 4 * h + j) * BN + rb_];
    };
    auto mm = [&](float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        if (PRI == 2) __builtin_amdgcn_s_setprio(1);
        if (PRI == 3) __builtin_amdgcn_s_setprio(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], acc1, 0, 0, 0);
        if (PRI == 2) __builtin_amdgcn_s_setprio(0);
        if (PRI == 3) __builtin_amdgcn_s_setprio(1);
    };
    const int c0 = kg * NCG;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stamp(p, 1);
    constexpr int PF = (OPT & 4) ? 2 : 1, R = (PF == 1) ? 2 : 4;
    float oa[R][4], ob[R][4];
#pragma unroll
    for (int c = 0; c < PF; c++) rd(lds, lds + BM * BK, c0 + c, oa[c], ob[c]);
    for (int s = 0; s < nst; s++) {
        const float *a = lds + (s & 1) * STAGE, *b = a + BM * BK;
        const float *a1 = lds + ((s & 1) ^ 1) * STAGE, *b1 = a1 + BM * BK;
        if (s + 1 < nst) issue(s + 1, (s & 1) ^ 1);          // the other buffer was released by the barrier that ended stage s-1
#pragma unroll
        for (int c = 0; c < NCG; c++) {
            if (c == NCG - PF && s + 1 < nst) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int nc = c + PF;
            if (nc < NCG) rd(a, b, c0 + nc, oa[nc % R], ob[nc % R]);
            else if (s + 1 < nst) rd(a1, b1, c0 + nc - NCG, oa[nc % R], ob[nc % R]);
            __builtin_amdgcn_sched_barrier(0);
            mm(oa[c % R], ob[c % R]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    stamp(p, 2);
    if (PRI) __builtin_amdgcn_s_setprio(0);
    const int gn = n0 + wn * 32 + l31;
    float add[16];
    __syncthreads();
    if (kg == 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) lds[(w4 * 16 + r) * 64 + lane] = acc0[r] + acc1[r];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int r = 0; r < 16; r++) add[r] = lds[(w4 * 16 + r) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = (acc0[r] + acc1[r]) + add[r];
        if (OPT & 1) __builtin_nontemporal_store(v, &p.O[(long)gm * N + gn]);
        else p.O[(long)gm * N + gn] = v;
    }
    stamp(p, 3);
}


// ---------------------------------------------------------------------------------------------------------------------
// Resource probes on V2's data path: MODE 0 = DMA only (issue a stage, wait, barrier), 1 = DMA + the operand LDS reads of the real kernel
// (no MFMA), 2 = LDS reads only (no DMA).  Reports cycles per 128-deep stage (the real loop: ~4650; pure MFMA: 4096).
template <int MODE>
__global__ void __launch_bounds__(512) k_path_probe(P p) {
    constexpr int BM = 64, BN = 64, BK = 128, CH = 32, STAGE = (BM + BN) * BK, NJ = 4, NCG = 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = w >> 2, w4 = w & 3, wm = w4 >> 1, wn = w4 & 1, h = lane >> 5, l31 = lane & 31;
    const int N = p.N, K = p.K;
    int tm, tn; tile_of(blockIdx.x, p.M / BM, N / BN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN, nst = K / BK;
    unsigned voffA[NJ], voffB[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int i = w * NJ + j;
        { const int r = i * 2 + lane / CH, ql = lane % CH, q = ql ^ (r & (CH - 1)); voffA[j] = (unsigned)(((m0 + r) * K + q * 4) * 4); }
        { const int kk = i * 4 + lane / 16, ch = lane % 16; voffB[j] = (unsigned)((kk * N + n0 + ch * 4) * 4); }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS void *)lds;
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + (w * NJ + j) * 256) * 4));
            dma16(voffA[j], p.A + (long)kt * BK, la);
            dma16(voffB[j], p.B + (long)kt * BK * N, la + BM * BK * 4);
        }
    };
    const int ra_ = wm * 32 + l31, rb_ = wn * 32 + l31, c0 = kg * NCG;
    float sink = 0.f;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 4; rep++)
    for (int kt = 0; kt < nst; kt++) {
        const int buf = kt & 1;
        if (MODE != 2) issue((kt + 1) % nst, buf ^ 1);
        if (MODE >= 1) {
            const float *a = lds + buf * STAGE, *b = a + BM * BK;
#pragma unroll
            for (int ci = 0; ci < NCG; ci++) {
                const v4f t = *reinterpret_cast<const v4f *>(a + ra_ * BK + ((((c0 + ci) * 2 + h) ^ (ra_ & (CH - 1))) << 2));
                sink += t[0] + t[1] + t[2] + t[3];
#pragma unroll
                for (int j = 0; j < 4; j++) sink += b[((c0 + ci) * 8 + 4 * h + j) * BN + rb_];
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (sink == 12345.678f) p.O[tid] = sink;
    if (p.st && tid == 0) p.st[blockIdx.x] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_ref(const float *A, const float *B, float *O, int M, int N, int K) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), m = blockIdx.y * 4 + (threadIdx.x >> 6);
    double acc = 0;
    for (int k = 0; k < K; k++) acc += (double)A[(long)m * K + k] * (double)B[(long)k * N + n];
    O[(long)m * N + n] = (float)acc;
}

static hipStream_t g_s;
struct Res { float us_avg, us_best; };
template <typename F> Res timeit(F f, int warm, int iters, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < warm; i++) f();
    float tot = 0, best = 1e9f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0, g_s); for (int i = 0; i < iters; i++) f(); hipEventRecord(e1, g_s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); const float us = ms * 1e3f / iters; tot += us; best = std::min(best, us);
    }
    return { tot / reps, best };
}

int main(int argc, char **argv) {
    const int M = 1024, N = 1024, K = 1024;
    hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking);
    float *A, *B, *O, *R; unsigned long long *st;
    hipMalloc(&A, 4L * M * K); hipMalloc(&B, 4L * K * N); hipMalloc(&O, 4L * M * N); hipMalloc(&R, 4L * M * N); hipMalloc(&st, 8 * 4096); hipMemset(st, 0, 8 * 4096);
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N), hO((size_t)M * N), hR((size_t)M * N);
    srand(1234);
    for (auto &x : hA) x = (float)rand() / (float)RAND_MAX;
    for (auto &x : hB) x = (float)rand() / (float)RAND_MAX;
    hipMemcpy(A, hA.data(), 4L * M * K, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), 4L * K * N, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_ref, dim3(N / 64, M / 4), dim3(256), 0, g_s, A, B, R, M, N, K);
    hipStreamSynchronize(g_s); hipMemcpy(hR.data(), R, 4L * M * N, hipMemcpyDeviceToHost);
    const char *only = argc > 1 ? argv[1] : nullptr;
    const int warm = argc > 2 ? atoi(argv[2]) : 1500;

    auto run = [&](const char *name, auto kern, int threads, size_t ldsb, bool check) {
        if (only && !strstr(name, only)) return;
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        P p{A, B, O, M, N, K, nullptr};
        hipMemsetAsync(O, 0, 4L * M * N, g_s);
        Res r = timeit([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), ldsb, g_s, p); }, warm, 200, 5);
        hipError_t e = hipStreamSynchronize(g_s);
        double err = -1;
        if (check) {
            hipMemcpy(hO.data(), O, 4L * M * N, hipMemcpyDeviceToHost);
            double mx = 0, mr = 0; for (size_t i = 0; i < hO.size(); i++) { mx = std::max(mx, (double)fabsf(hO[i] - hR[i])); mr = std::max(mr, (double)fabsf(hR[i])); }
            err = mx / mr;
        }
        // stamped run (after the timed ones, clocks warm): 50 launches, keep the last
        P ps = p; ps.st = st;
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), ldsb, g_s, ps);
        hipStreamSynchronize(g_s);
        std::vector<unsigned long long> hs(8 * 256); hipMemcpy(hs.data(), st, 8 * 8 * 256, hipMemcpyDeviceToHost);
        { std::vector<unsigned long long> ws(2048); hipMemcpy(ws.data(), st + 2048, 8 * 2048, hipMemcpyDeviceToHost);
          if (ws[0]) for (int b = 0; b < 2; b++) for (int w = 0; w < 8; w++) { const unsigned long long *d = &ws[(b * 8 + w) * 4];
              printf("    blk %d wave %d: loop %6llu barrier-wait %6llu counter-wait %6llu hw_id %08llx (simd %llu wave-slot %llu cu %llu)\n", b, w, d[0], d[1], d[2], d[3], (d[3] >> 4) & 3, d[3] & 15, (d[3] >> 8) & 15); } }
        hipMemset(st, 0, 8 * 4096);
        double pro = 0, loop = 0, epi = 0, tot = 0, clk = 0, iss = 0; unsigned long long t0min = ~0ull, t1max = 0, r0min = ~0ull, r1max = 0;
        for (int b = 0; b < 256; b++) {
            const unsigned long long *s = &hs[b * 8];
            pro += (double)(s[1] - s[0]); iss += s[6] > s[0] && s[6] < s[1] ? (double)(s[6] - s[0]) : 0; loop += (double)(s[2] - s[1]); epi += (double)(s[3] - s[2]); tot += (double)(s[3] - s[0]);
            clk += (double)(s[3] - s[0]) / (double)(s[5] - s[4]) * 100.0;
            t0min = std::min(t0min, s[0]); t1max = std::max(t1max, s[3]); r0min = std::min(r0min, s[4]); r1max = std::max(r1max, s[5]);
        }
        printf("%-28s avg %6.2f us best %6.2f | %5.1f TF %4.1f%% | err %.1e %s | cyc pro %6.0f loop %6.0f epi %5.0f tot %6.0f | clk %4.0f MHz | span %.2f us | issue@%.0f\n",
               name, r.us_avg, r.us_best, 2.0 * M * N * K / (r.us_avg * 1e-6) / 1e12, 100.0 * 2.0 * M * N * K / (r.us_avg * 1e-6) / 157.3e12, err,
               e == hipSuccess ? "" : hipGetErrorString(e), pro / 256, loop / 256, epi / 256, tot / 256, clk / 256, (double)(r1max - r0min) / 100.0, iss / 256);
        fflush(stdout);
    };
    auto probe = [&](const char *name, auto kern) {
        if (only && !strstr(name, only)) return;
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        P p{A, B, O, M, N, K, st};
        const int chunks = 512;
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 65536, g_s, p, chunks);
        hipStreamSynchronize(g_s);
        std::vector<unsigned long long> h(256); hipMemcpy(h.data(), st, 8 * 256, hipMemcpyDeviceToHost); hipMemset(st, 0, 8 * 4096);
        double sum = 0; for (int i = 0; i < 256; i++) sum += (double)h[i];
        printf("%-24s %.1f cycles per chunk (ideal 256)\n", name, sum / 256 / chunks); fflush(stdout);
    };
    probe("probe_p0_2acc", k_issue_probe<0, 2>); probe("probe_p0_4acc", k_issue_probe<0, 4>);
    probe("probe_p1_2acc", k_issue_probe<1, 2>); probe("probe_p1_4acc", k_issue_probe<1, 4>);
    probe("probe_p2_2acc", k_issue_probe<2, 2>); probe("probe_p2_4acc", k_issue_probe<2, 4>);
    probe("probe_p3_2acc", k_issue_probe<3, 2>); probe("probe_p3_4acc", k_issue_probe<3, 4>);
    probe("probe_p4_2acc", k_issue_probe<4, 2>); probe("probe_p4_4acc", k_issue_probe<4, 4>);
    probe("probe_p5_2acc", k_issue_probe<5, 2>); probe("probe_p5_4acc", k_issue_probe<5, 4>);
    probe("probe_p6_2acc", k_issue_probe<6, 2>); probe("probe_p6_4acc", k_issue_probe<6, 4>);
    if (!only || strstr("prod", only)) {                    // the product library's t4k_gemm on the same operands, timed the same way
        void *h = dlopen(argc > 3 ? argv[3] : "tensorforth_amd/libt4hip.so", RTLD_NOW);
        if (h) {
            auto init = (int (*)(int))dlsym(h, "t4k_init");
            auto gemm = (int (*)(const float *, const float *, float *, float, float, int, int, int, int, int, int, void *))dlsym(h, "t4k_gemm");
            init(0);
            Res r = timeit([&] { gemm(A, B, O, 1.0f, 0.0f, 0, 0, M, N, K, 1, (void *)g_s); }, warm, 200, 5);
            hipStreamSynchronize(g_s);
            hipMemcpy(hO.data(), O, 4L * M * N, hipMemcpyDeviceToHost);
            double mx = 0, mr = 0; for (size_t i = 0; i < hO.size(); i++) { mx = std::max(mx, (double)fabsf(hO[i] - hR[i])); mr = std::max(mr, (double)fabsf(hR[i])); }
            printf("%-28s avg %6.2f us best %6.2f | %5.1f TF %4.1f%% | err %.1e\n", "prod t4k_gemm", r.us_avg, r.us_best, 2.0 * M * N * K / (r.us_avg * 1e-6) / 1e12,
                   100.0 * 2.0 * M * N * K / (r.us_avg * 1e-6) / 157.3e12, mx / mr);
        } else printf("prod: %s\n", dlerror());
    }
    auto pathp = [&](const char *name, auto kern) {
        if (only && !strstr(name, only)) return;
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        P p{A, B, O, M, N, K, st};
        for (int i = 0; i < 300; i++) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, g_s, p);
        hipStreamSynchronize(g_s);
        std::vector<unsigned long long> h(256); hipMemcpy(h.data(), st, 8 * 256, hipMemcpyDeviceToHost); hipMemset(st, 0, 8 * 4096);
        double sum = 0, mx = 0; for (int i = 0; i < 256; i++) { sum += (double)h[i]; mx = std::max(mx, (double)h[i]); }
        printf("%-24s %.0f cycles per 128-deep stage (avg over workgroups; slowest %.0f)\n", name, sum / 256 / 32, mx / 32); fflush(stdout);
    };
    pathp("path_dma_only", k_path_probe<0>); pathp("path_dma_lds", k_path_probe<1>); pathp("path_lds_only", k_path_probe<2>);
    const size_t L2 = 2 * 128 * 128 * 4;
    run("v0", k_v0<0, 0>, 512, L2, true);
    run("v0_spread", k_v0<0, 1>, 512, L2, true);
    run("v0_abl_nodma", k_v0<1, 0>, 512, L2, false);
    run("v0_abl_nolds", k_v0<2, 0>, 512, L2, false);
    run("v0_abl_nobar", k_v0<4, 0>, 512, L2, false);
    run("v0_abl_nostore", k_v0<8, 0>, 512, L2, false);
    run("v0_abl_mfmaonly", k_v0<15, 0>, 512, L2, false);
    run("v2_base", k_v2<0, 0, 0, 0>, 512, L2, true);
    run("v2_pin", k_v2<0, 0, 0, 1>, 512, L2, true);
    run("v2_pin_stat", k_v2<16, 0, 0, 1>, 512, L2, true);
    run("v2_ilv_pin", k_v2<0, 1, 0, 1>, 512, L2, true);
    run("v2_ilv_sub_pin", k_v2<0, 1, 1, 1>, 512, L2, true);
    run("v2_ilv_sub", k_v2<0, 1, 1, 0>, 512, L2, true);
    run("v2_pin_abl_nodma", k_v2<1, 0, 0, 1>, 512, L2, false);
    run("v2_pin_abl_nobar", k_v2<4, 0, 0, 1>, 512, L2, false);
    run("v3_kg2_pf1", k_v3<2, 1, 0>, 512, L2, true);
    run("v3_kg2_pf2", k_v3<2, 2, 0>, 512, L2, true);
    run("v3_kg2_pf3", k_v3<2, 3, 0>, 512, L2, true);
    run("v3_kg4_pf1", k_v3<4, 1, 0>, 1024, L2, true);
    run("v3_kg4_pf2", k_v3<4, 2, 0>, 1024, L2, true);
    run("v3_kg1_pf2", k_v3<1, 2, 0>, 256, L2, true);
    run("v3_kg1_pf3", k_v3<1, 3, 0>, 256, L2, true);
    run("v3_kg2_pf2_abl_nodma", k_v3<2, 2, 1>, 512, L2, false);
    run("v3_kg2_pf2_abl_nobar", k_v3<2, 2, 4>, 512, L2, false);
    run("v3_kg4_pf2_abl_nobar", k_v3<4, 2, 4>, 1024, L2, false);
    run("v4_na10", k_v4<10, 0, 0>, 512, L2, true);
    run("v4_na11", k_v4<11, 0, 0>, 512, L2, true);
    run("v4_na12", k_v4<12, 0, 0>, 512, L2, true);
    run("v4_na13", k_v4<13, 0, 0>, 512, L2, true);
    run("v4_na14", k_v4<14, 0, 0>, 512, L2, true);
    run("v4_na11_db", k_v4<11, 1, 0>, 512, L2, true);
    run("v4_na12_db", k_v4<12, 1, 0>, 512, L2, true);
    run("v4_na13_db", k_v4<13, 1, 0>, 512, L2, true);
    run("v4_na14_db", k_v4<14, 1, 0>, 512, L2, true);
    run("v4_na12_stat", k_v4<12, 0, 16>, 512, L2, true);
    run("v4_na13_db_stat", k_v4<13, 1, 16>, 512, L2, true);
    run("v5_plain", k_v5<0, 0>, 512, L2, true);
    run("v5_nt", k_v5<1, 0>, 512, L2, true);
    run("v5_fastpro", k_v5<2, 0>, 512, L2, true);
    run("v5_fastpro_nt", k_v5<3, 0>, 512, L2, true);
    run("v5_pf2", k_v5<4, 0>, 512, L2, true);
    run("v5_fastpro_pri1", k_v5<2, 1>, 512, L2, true);
    run("v5_fastpro_pri2", k_v5<2, 2>, 512, L2, true);
    run("v5_fastpro_pri3", k_v5<2, 3>, 512, L2, true);
    run("v1_w8", k_v1<8, 0, 0, 0>, 512, L2, true);
    run("v1_w8_gu6", k_v1<8, 0, 6, 0>, 512, L2, true);
    run("v1_w4", k_v1<4, 0, 0, 0>, 256, L2, true);
    run("v1_w4_gu12", k_v1<4, 0, 12, 0>, 256, L2, true);
    run("v1_w4_gu8", k_v1<4, 0, 8, 0>, 256, L2, true);
    run("v1_w4_sub", k_v1<4, 0, 0, 1>, 256, L2, true);
    run("v1_w4_gu12_sub", k_v1<4, 0, 12, 1>, 256, L2, true);
    run("v1_w8_gu6_sub", k_v1<8, 0, 6, 1>, 512, L2, true);
    run("v1_w4_abl_nodma", k_v1<4, 1, 0, 0>, 256, L2, false);
    run("v1_w4_abl_nolds", k_v1<4, 2, 0, 0>, 256, L2, false);
    run("v1_w4_abl_nobar", k_v1<4, 4, 0, 0>, 256, L2, false);
    run("v1_w4_abl_mfmaonly", k_v1<4, 15, 0, 0>, 256, L2, false);
    run("v1_w8_abl_mfmaonly", k_v1<8, 15, 0, 0>, 512, L2, false);
    return 0;
}
