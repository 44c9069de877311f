// EXPERIMENT (not built): measured 13.0 us forward / 21.2 us backward for the 980->100 layer at batch 128 versus 11.0 / 20.8 us
// for the LDS-staged 64x64 GEMM path - per-lane row-strided 16 B loads from L2 cost more than the staging they avoid.
// skinny.hip - mini-batch sized GEMMs of the fully-connected layers (one dimension = batch <= 512), e.g. the LeNet
// 980 -> 100 layer at batch 128: 25 MFLOP, operands 0.4-0.5 MB, i.e. L2 resident and latency bound.
//
// One wave owns one 32x32 accumulator (v_mfma_f32_32x32x2_f32) and reads its operands STRAIGHT from L2 - no LDS
// staging, no workgroup barrier: with 16-124 output tiles there is nothing to share between waves, and a barrier per
// k-step is exactly the latency this shape cannot afford.  Per trip a lane issues all loads of 32 k (two float4 when
// the operand is k-contiguous, coalesced scalars otherwise), then 16 MFMAs.
//   forward  Y = X W^T + b          A = X  [M,K] k-contiguous,  B = W  [N,K] k-contiguous, split-K
//   dX       dX = dY W              A = dY [M,K] k-contiguous,  B = W  [K,N] n-contiguous
//   dW|dB    dW += dY^T X, dB += .. A = dY [K,M] m-contiguous,  B = X  [K,N] n-contiguous + a virtual all-ones column N
// Split-K partials meet in the workspace: every wave stores its 32x32 block with agent-scope (write-through) stores,
// takes a ticket, and the LAST arriver of a tile sums the S blocks in slice order (deterministic) and runs the
// epilogue - no second launch, no fence (a release fence per workgroup costs an L2 write-back, see gemm.hip).
// Reference: Tensor::linear / gemm3 tensor.cu:79-87,161-180; _flinear forward.cu:157-198; _blinear backprop.cu:193-254.
#include "t4k_common.h"

using namespace t4k;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct SkP {
    const float *A, *B, *bias;
    float *O, *DB, *part;
    int *cnt;
    int M, N, K, lda, ldb, ldo;
    int tiles_n, S, kchunk, ones;
    float beta;
};

template <bool AK, bool BK>
__global__ void __launch_bounds__(64) k_skinny(SkP p) {
    const int lane = threadIdx.x, h = lane >> 5, l = lane & 31;
    const int tile = blockIdx.x, s = blockIdx.y;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * 32, n0 = tn * 32;
    const int kb = s * p.kchunk, ke = min(p.K, kb + p.kchunk);
    const int ri = min(m0 + l, p.M - 1);                         // clamped operand row / column of this lane
    const int cj = n0 + l, cjc = min(cj, p.N - 1);
    const bool one_col = p.ones && cj == p.N;                    // virtual all-ones column (bias gradient)
    const float *Ar = AK ? p.A + (long)ri * p.lda : p.A + ri;
    const float *Br = BK ? p.B + (long)cjc * p.ldb : p.B + cjc;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;

    for (int k = kb; k < ke; k += 32) {                          // 4 chunks of 8 k per trip: all loads first, then 16 MFMAs
        float a[4][4], b[4][4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int kk = k + 8 * c + 4 * h;                    // this lane half's four k: kk .. kk+3 (K % 4 == 0)
            const bool ok = kk < ke;
            const int kc = ok ? kk : kb;
            if (AK) { const v4f t = *reinterpret_cast<const v4f *>(Ar + kc);
#pragma unroll
                      for (int u = 0; u < 4; u++) a[c][u] = ok ? t[u] : 0.f; }
            else {
#pragma unroll
                      for (int u = 0; u < 4; u++) { const float t = Ar[(long)(kc + u) * p.lda]; a[c][u] = ok ? t : 0.f; } }
            if (BK) { const v4f t = *reinterpret_cast<const v4f *>(Br + kc);
#pragma unroll
                      for (int u = 0; u < 4; u++) b[c][u] = ok ? t[u] : 0.f; }
            else {
#pragma unroll
                      for (int u = 0; u < 4; u++) { const float t = Br[(long)(kc + u) * p.ldb]; b[c][u] = ok ? (one_col ? 1.f : t) : 0.f; } }
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][u], b[c][u], acc, 0, 0, 0);
    }

    if (p.S > 1) {                                               // split-K: park the block, last arriver of the tile folds
        float *slot = p.part + ((long)tile * p.S) * 1024;
#pragma unroll
        for (int r = 0; r < 16; r++) __hip_atomic_store(&slot[s * 1024 + r * 64 + lane], acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my stores are acknowledged before I take a ticket
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&p.cnt[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t != p.S - 1) return;
        if (lane == 0) __hip_atomic_store(&p.cnt[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        for (int q = 0; q < p.S; q++)                             // slice order, whoever arrives last
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] += __hip_atomic_load(&slot[q * 1024 + r * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // epilogue: D[row][col], col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * h
    const int gn = n0 + l;
    const float bias = (p.bias && gn < p.N) ? p.bias[gn] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (gm >= p.M) continue;
        if (gn < p.N) {
            float *o = p.O + (long)gm * p.ldo + gn;
            float v = acc[r] + bias;
            if (p.beta != 0.f) v += p.beta * *o;
            *o = v;
        } else if (p.ones && gn == p.N) p.DB[gm] += acc[r];
    }
}

bool al16(const void *q) { return (((uintptr_t)q) & 15) == 0; }

} // namespace

namespace t4k {

// O[M,N] = A B (+bias) (+beta O).  a_kcontig: A is [M,K] (else [K,M]); b_kcontig: B is [N,K] (else [K,N]).
// ones: B gets a virtual column N of ones whose result is ADDED to DB[M].  Returns false if the shape does not qualify.
bool skinny_gemm(const float *A, const float *B, const float *bias, float *O, float *DB, float beta,
                 bool a_kcontig, bool b_kcontig, int M, int N, int K, int ones, t4k_stream_t s) {
    if (M < 1 || N < 1 || K < 4 || (K & 3)) return false;
    if ((long)M * (N + ones) > 262144 || (M > 512 && N > 512)) return false;        // "skinny": at most a few hundred tiles
    const int lda = a_kcontig ? K : M, ldb = b_kcontig ? K : N;
    if (a_kcontig && (!al16(A) || (lda & 3))) return false;
    if (b_kcontig && (!al16(B) || (ldb & 3))) return false;
    State &g = st();
    if (!g.d_sync) return false;
    SkP p;
    p.A = A; p.B = B; p.bias = bias; p.O = O; p.DB = DB; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldo = N;
    p.ones = ones; p.beta = beta;
    const int tiles_m = (M + 31) / 32;
    p.tiles_n = (N + ones + 31) / 32;
    const int tiles = tiles_m * p.tiles_n;
    int ns = 1;
    if (tiles * 2 <= g.cu_count && K >= 256) {                   // few tiles, long K: split so ~2 waves land on every CU
        ns = (2 * g.cu_count + tiles - 1) / tiles; if (ns > 16) ns = 16;
        const int maxs = K / 64; if (ns > maxs) ns = maxs; if (ns < 1) ns = 1;
    }
    p.kchunk = (((K + ns - 1) / ns) + 31) / 32 * 32;
    ns = (K + p.kchunk - 1) / p.kchunk;
    p.S = ns;
    if (tiles > 4096 || (ns > 1 && (size_t)tiles * ns * 1024 * sizeof(float) > g.ws_bytes / 2)) return false;
    p.part = ws_for(s);
    p.cnt = g.d_sync + 4096;                                     // tile tickets [4096, 8192) (pair-mode GEMM and linear_small use [0, 4096))
    const dim3 grid(tiles, ns), blk(64);
    hipStream_t hs = S(s);
    if (a_kcontig && b_kcontig)       hipLaunchKernelGGL((k_skinny<true, true>),   grid, blk, 0, hs, p);
    else if (a_kcontig)               hipLaunchKernelGGL((k_skinny<true, false>),  grid, blk, 0, hs, p);
    else if (b_kcontig)               hipLaunchKernelGGL((k_skinny<false, true>),  grid, blk, 0, hs, p);
    else                              hipLaunchKernelGGL((k_skinny<false, false>), grid, blk, 0, hs, p);
    return true;
}

} // namespace t4k
