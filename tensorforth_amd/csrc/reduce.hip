// reduce.hip - reductions on wave64 shuffles: full reductions (sum / n*var / max / min /
// BCE / NaN count), per-channel dot, row softmax, hit count, bias gradient, batchnorm.
// Every reduction is deterministic (fixed tree, no fp32 atomics), unlike the reference's
// atomicAdd epilogues.
// Reference: src/t4math.cu:23-131,248-365, src/nn/nmath.cu:74-414, src/nn/loss.cpp:75-107.
#include "t4k_common.h"
#include <float.h>

using namespace t4k;

namespace {

constexpr int RED_MAX_PARTS = 1024;

enum { R_SUM = 0, R_NVAR, R_MAX, R_MIN, R_BCE };

template <int OP> __device__ __forceinline__ float r_init() {
    return OP == R_MAX ? -FLT_MAX : (OP == R_MIN ? FLT_MAX : 0.0f);
}
template <int OP> __device__ __forceinline__ float r_comb(float a, float b) {
    return OP == R_MAX ? fmaxf(a, b) : (OP == R_MIN ? fminf(a, b) : a + b);
}
template <int OP> __device__ __forceinline__ float r_term(float x, float y, float avg) {
    if (OP == R_NVAR) { float d = x - avg; return d * d; }
    if (OP == R_BCE)  return y * __logf(x + DU_EPS) + (1.0f - y) * __logf(1.0f - x + DU_EPS);  // x = O, y = T
    return x;
}
template <int OP> __device__ __forceinline__ float wave_comb_all(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = r_comb<OP>(v, __shfl_xor(v, off, 64));
    return v;
}
template <int OP> __device__ __forceinline__ float block_comb(float v, float *sm) {
    v = wave_comb_all<OP>(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = r_comb<OP>(r_comb<OP>(sm[0], sm[1]), r_comb<OP>(sm[2], sm[3]));
    __syncthreads();
    return r;
}

// stage 1: each block reduces a grid-strided slice to one partial (or the result if gridDim==1)
template <int OP>
__global__ void __launch_bounds__(BLK) k_reduce1(const float *__restrict__ X, const float *__restrict__ Y,
                                                 long n, float avg, float *__restrict__ out, bool vec) {
    __shared__ float sm[4];
    const long tx = (long)blockIdx.x * BLK + threadIdx.x, step = (long)gridDim.x * BLK;
    float v = r_init<OP>();
    if (vec) {
        const long n4 = n >> 2;
        for (long q = tx; q < n4; q += step) {
            float4 x = reinterpret_cast<const float4 *>(X)[q];
            float4 y = (OP == R_BCE) ? reinterpret_cast<const float4 *>(Y)[q] : make_float4(0, 0, 0, 0);
            v = r_comb<OP>(v, r_term<OP>(x.x, y.x, avg)); v = r_comb<OP>(v, r_term<OP>(x.y, y.y, avg));
            v = r_comb<OP>(v, r_term<OP>(x.z, y.z, avg)); v = r_comb<OP>(v, r_term<OP>(x.w, y.w, avg));
        }
        for (long j = (n4 << 2) + tx; j < n; j += step) v = r_comb<OP>(v, r_term<OP>(X[j], OP == R_BCE ? Y[j] : 0.0f, avg));
    } else {
        for (long j = tx; j < n; j += step) v = r_comb<OP>(v, r_term<OP>(X[j], OP == R_BCE ? Y[j] : 0.0f, avg));
    }
    v = block_comb<OP>(v, sm);
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
// stage 2: one block folds the partials in index order
template <int OP>
__global__ void __launch_bounds__(BLK) k_reduce2(const float *__restrict__ part, int np, float *__restrict__ out) {
    __shared__ float sm[4];
    float v = r_init<OP>();
    for (int j = threadIdx.x; j < np; j += BLK) v = r_comb<OP>(v, part[j]);
    v = block_comb<OP>(v, sm);
    if (threadIdx.x == 0) *out = v;
}

template <int OP>
int launch_reduce(const float *X, const float *Y, long n, float avg, float *out, hipStream_t s) {
    const bool vec = aligned16(X) && (OP != R_BCE || aligned16(Y));
    long g = (n + (long)BLK * 16 - 1) / ((long)BLK * 16);
    if (g > RED_MAX_PARTS) g = RED_MAX_PARTS;
    if (g <= 1) {
        T4K_LAUNCH(k_reduce1<OP>, dim3(1), dim3(BLK), 0, s, X, Y, n, avg, out, vec);
    } else {
        float *part = ws_for(s);
        T4K_LAUNCH(k_reduce1<OP>, dim3((int)g), dim3(BLK), 0, s, X, Y, n, avg, part, vec);
        T4K_LAUNCH(k_reduce2<(OP == R_MAX || OP == R_MIN) ? OP : R_SUM>, dim3(1), dim3(BLK), 0, s, part, (int)g, out);
    }
    T4K_LAUNCH_CHECK();
    return T4K_OK;
}

// NaN/Inf count: integer atomics are order-independent
__global__ void __launch_bounds__(BLK) k_nan_inf(const float *__restrict__ X, long n, int *cnt) {
    int v = 0;
    for (long j = (long)blockIdx.x * BLK + threadIdx.x; j < n; j += (long)gridDim.x * BLK) {
        float x = X[j];
        if (isnan(x) || isinf(x)) v++;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(cnt, v);
}

// O[c] = alpha * <A[:,c], B[:,c]> + beta * O[c]; one block per channel
__global__ void __launch_bounds__(BLK) k_dot(const float *__restrict__ A, const float *__restrict__ B, float *O,
                                             float alpha, float beta, int K, int C) {
    __shared__ float sm[4];
    const int c = blockIdx.x;
    float acc = 0.0f;
    for (int k = threadIdx.x; k < K; k += BLK) { long i = (long)k * C + c; acc = fmaf(A[i], B[i], acc); }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) O[c] = acc * alpha + (beta == 0.0f ? 0.0f : O[c] * beta);
}

// row softmax: one wave per sample (lanes stride over C), 4 samples per block
__global__ void __launch_bounds__(BLK) k_softmax(const float *__restrict__ I, float *__restrict__ O, int N, int C) {
    const int lane = threadIdx.x & 63;
    const int row  = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float *s = I + (long)row * C; float *d = O + (long)row * C;
    float mx = -FLT_MAX;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, s[c]);
    mx = wave_max_all(mx);
    float sm = 0.0f;
    for (int c = lane; c < C; c += 64) { float e = __expf(s[c] - mx); d[c] = e; sm += e; }
    sm = wave_sum_all(sm);
    for (int c = lane; c < C; c += 64) d[c] = d[c] / sm;
}

// log-softmax layer as the reference computes it (_flogsoftmax forward.cu:245-259, quirk a-16 kept): O = exp(I) - log10(max(sum_c exp(I), 1e-6)).
// One thread per row, the sum in index order (the reference sums each row sequentially on one thread as well).
__global__ void __launch_bounds__(BLK) k_logsoftmax(const float *__restrict__ I, float *O, int N, int C) {
    const int n = blockIdx.x * BLK + threadIdx.x;
    if (n >= N) return;
    const float *x = I + (long)n * C; float *o = O + (long)n * C;
    float sum = 0.f;
    for (int c = 0; c < C; c++) { const float e = __expf(x[c]); o[c] = e; sum += e; }
    const float ls = log10f(fmaxf(sum, 1.0e-6f));
    for (int c = 0; c < C; c++) o[c] -= ls;
}

// hit count: per-sample FIRST arg-max, then sum of hot[n, argmax]; single block, exact.  G lanes share a sample (G = 1 for class-count
// sized rows: 256 samples per pass, every thread scans its own row; wider rows take 8 or 64 lanes) - the one-wave-per-sample form this
// replaces walked a 128 x 10 batch in 32 dependent passes (31 us, the longest kernel of a dataset-fed LeNet step).
template <int G>
__global__ void __launch_bounds__(BLK) k_hit(const float *__restrict__ out, const float *__restrict__ hot, int N, int E, int *cnt,
                                             const uint32_t *__restrict__ label = nullptr, float *hot_w = nullptr) {   // label != NULL: also WRITE the one-hot rows (Model::onehot(Dataset&)) and count against the label
    __shared__ int sm[4];
    const int l = threadIdx.x % G, g = threadIdx.x / G;
    int local = 0;
    for (int n0 = 0; n0 < N; n0 += BLK / G) {            // uniform trip count: the shuffles below need every lane of the wave
        const int n = n0 + g; const bool live = n < N;
        const float *o = out + (long)(live ? n : 0) * E;
        float m = -FLT_MAX; int idx = 0x7fffffff;
        if (live) for (int e = l; e < E; e += G) { float v = o[e]; if (v > m) { m = v; idx = e; } }   // first max within lane
#pragma unroll
        for (int off = G >> 1; off > 0; off >>= 1) {
            float m2 = __shfl_xor(m, off, 64); int i2 = __shfl_xor(idx, off, 64);
            if (m2 > m || (m2 == m && i2 < idx)) { m = m2; idx = i2; }
        }
        if (label) {
            uint32_t m = live ? label[n] : 0u; if (m >= (uint32_t)E) m = 0;
            if (live) for (int e = l; e < E; e += G) hot_w[(long)n * E + e] = (e == (int)m) ? 1.0f : 0.0f;
            if (live && l == 0) { if (idx >= E) idx = 0; local += (idx == (int)m) ? 1 : 0; }
        } else
        if (live && l == 0) { if (idx >= E) idx = 0; local += (int)hot[(long)n * E + idx]; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) *cnt = sm[0] + sm[1] + sm[2] + sm[3];
}

// DB[e] += sum_n DY[n,e]: 64 columns x 4 row-groups per block, coalesced along e
__global__ void __launch_bounds__(BLK) k_dlinear_db(const float *__restrict__ DY, float *DB, int N, int E0) {
    __shared__ float sm[4][64];
    const int ex = threadIdx.x & 63, ny = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + ex;
    float acc = 0.0f;
    if (e < E0) for (int n = ny; n < N; n += 4) acc += DY[(long)n * E0 + e];
    sm[ny][ex] = acc;
    __syncthreads();
    if (ny == 0 && e < E0) DB[e] += (sm[0][ex] + sm[1][ex]) + (sm[2][ex] + sm[3][ex]);
}

// batchnorm statistics = two column sums of the [N*H*W, C] matrix view (NHWC: a row is one pixel's channels).
// Stage 1: grid (row chunks, C/64); a workgroup = 64 adjacent channels x 4 row groups, so every load instruction of a wave
// is one contiguous 256-byte run (the reference - and the first version here - walked a single channel with stride C:
// 1/64 of every cache line used).  Stage 2: one wave per channel folds the chunk partials with a fixed xor tree and
// finalises.  Deterministic: no fp32 atomics.  MODE 0: sum x, sum x^2 (k_batchnorm_1/2 nmath.cu:177-242);
// MODE 1: sum dy, sum dy*xhat (k_dbatchnorm_1 nmath.cu:295-381).
template <int MODE>
__global__ void __launch_bounds__(BLK) k_bn_part(const float *__restrict__ X, const float *__restrict__ Y, float *__restrict__ part,
                                                 long rows, int C, int rows_per_chunk) {
    __shared__ float sm[2][4][64];
    const int ex = threadIdx.x & 63, ry = threadIdx.x >> 6, e = blockIdx.y * 64 + ex;
    const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float a = 0.f, b = 0.f;
    if (e < C) {
#pragma unroll 4
        for (long r = r0 + ry; r < r1; r += 4) {
            const float v = X[r * C + e];
            if (MODE == 0) { a += v; b = fmaf(v, v, b); }
            else           { a += v; b = fmaf(v, Y[r * C + e], b); }
        }
    }
    sm[0][ry][ex] = a; sm[1][ry][ex] = b;
    __syncthreads();
    if (ry == 0 && e < C) {
        part[((long)blockIdx.x * 2 + 0) * C + e] = (sm[0][0][ex] + sm[0][1][ex]) + (sm[0][2][ex] + sm[0][3][ex]);
        part[((long)blockIdx.x * 2 + 1) * C + e] = (sm[1][0][ex] + sm[1][1][ex]) + (sm[1][2][ex] + sm[1][3][ex]);
    }
}
// The same two column sums with 16-byte loads (C % 4 == 0, 16-byte aligned tensors): a lane owns four adjacent channels, 16 lanes cover one 256-byte row of the
// 64-channel tile, a wave four rows per instruction, the workgroup sixteen - four times the bytes in flight per lane (k_bn_part on the CIFAR net's
// 256 x 32 x 32 x 64 conv output: 22 us = 3 TB/s; this one: see profiles/LAB_NOTES.md).  Fixed order: a lane's rows, the wave's four row groups (xor 16, 32), the four waves.
template <int MODE>
__global__ void __launch_bounds__(BLK) k_bn_part4(const float *__restrict__ X, const float *__restrict__ Y, float *__restrict__ part,
                                                  long rows, int C, int rows_per_chunk) {
    __shared__ float sm[2][4][64];
    const int lane = threadIdx.x & 63, ry = threadIdx.x >> 6, c4 = lane & 15, rs = lane >> 4, e = blockIdx.y * 64 + c4 * 4;
    const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    if (e < C) {
#pragma unroll 4
        for (long r = r0 + ry * 4 + rs; r < r1; r += 16) {
            const float4 v = *reinterpret_cast<const float4 *>(X + r * C + e);
            const float vv[4] = { v.x, v.y, v.z, v.w };
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < 4; q++) { a[q] += vv[q]; b[q] = fmaf(vv[q], vv[q], b[q]); }
            } else {
                const float4 y = *reinterpret_cast<const float4 *>(Y + r * C + e);
                const float yy[4] = { y.x, y.y, y.z, y.w };
#pragma unroll
                for (int q = 0; q < 4; q++) { a[q] += vv[q]; b[q] = fmaf(vv[q], yy[q], b[q]); }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        a[q] += __shfl_xor(a[q], 16); a[q] += __shfl_xor(a[q], 32);
        b[q] += __shfl_xor(b[q], 16); b[q] += __shfl_xor(b[q], 32);
    }
    if (rs == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) { sm[0][ry][c4 * 4 + q] = a[q]; sm[1][ry][c4 * 4 + q] = b[q]; }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, ex = threadIdx.x & 63, ec = blockIdx.y * 64 + ex;
        if (ec < C) part[((long)blockIdx.x * 2 + which) * C + ec] = (sm[which][0][ex] + sm[which][1][ex]) + (sm[which][2][ex] + sm[which][3][ex]);
    }
}
template <int MODE>
__global__ void __launch_bounds__(BLK) k_bn_fin(const float *__restrict__ part, float *stat, float *DW, float *DB,
                                                long NHW, int C, int nchunk, int train) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    float a = 0.f, b = 0.f;
#pragma unroll 8
    for (int k = lane; k < nchunk; k += 64) { a += part[((long)k * 2 + 0) * C + c]; b += part[((long)k * 2 + 1) * C + c]; }   // (unrolled: the loads of eight trips in flight together, same order of additions)
    a = wave_sum_all(a); b = wave_sum_all(b);
    if (lane == 0) {
        if (MODE == 0) {
            const float avg = a / (float)NHW;
            const float var = b / (float)NHW - avg * avg;
            stat[C + c] = avg;
            stat[c]     = 1.0f / (sqrtf(fmaxf(var, 0.0f)) + DU_EPS);
        } else {
            const float s1 = a / (float)NHW, s2 = b / (float)NHW;
            stat[C + c] = s1; stat[2 * C + c] = s2;
            if (train) { DB[c] += s1; DW[c] += s2; }
        }
    }
}
// Data-parallel (synchronised) batch norm: the batch is sharded over ranks, so the per-channel sums must cover every shard.
// k_bn_sums folds the chunk partials into sums[0..2C) (all-reduced in place by the caller) and keeps the rank-local copy in
// sums[2C..4C); k_bn_fin_sync finalises from the global sums over NHW_global = world x NHW.  dgamma/dbeta accumulate the LOCAL
// sums over NHW_global, so the SUM all-reduce of the gradient slab that follows yields the same batch means one rank x the
// whole batch would have produced.
__global__ void __launch_bounds__(BLK) k_bn_sums(const float *__restrict__ part, float *__restrict__ sums, int C, int nchunk) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int k = lane; k < nchunk; k += 64) { a += part[((long)k * 2 + 0) * C + c]; b += part[((long)k * 2 + 1) * C + c]; }
    a = wave_sum_all(a); b = wave_sum_all(b);
    if (lane == 0) { sums[c] = a; sums[C + c] = b; sums[2 * C + c] = a; sums[3 * C + c] = b; }
}
template <int MODE>
__global__ void __launch_bounds__(BLK) k_bn_fin_sync(const float *__restrict__ sums, float *stat, float *DW, float *DB,
                                                     float NHWg, int C, int train) {
    const int c = blockIdx.x * BLK + threadIdx.x;
    if (c >= C) return;
    const float a = sums[c], b = sums[C + c];
    if (MODE == 0) {
        const float avg = a / NHWg, var = b / NHWg - avg * avg;
        stat[C + c] = avg;
        stat[c]     = 1.0f / (sqrtf(fmaxf(var, 0.0f)) + DU_EPS);
    } else {
        stat[C + c] = a / NHWg; stat[2 * C + c] = b / NHWg;
        if (train) { DB[c] += sums[2 * C + c] / NHWg; DW[c] += sums[3 * C + c] / NHWg; }
    }
}
// single-launch variant for small row counts: one block per channel, finalised in the same launch
__global__ void __launch_bounds__(BLK) k_bn_stats(const float *__restrict__ I, float *stat, long NHW, int C) {
    __shared__ float sm[4];
    const int c = blockIdx.x;
    float ts = 0.0f, tq = 0.0f;
    for (long k = threadIdx.x; k < NHW; k += BLK) { float v = I[k * C + c]; ts += v; tq = fmaf(v, v, tq); }
    ts = block_sum(ts, sm); tq = block_sum(tq, sm);
    if (threadIdx.x == 0) {
        float avg = ts / (float)NHW;
        float var = tq / (float)NHW - avg * avg;
        stat[C + c] = avg;
        stat[c]     = 1.0f / (sqrtf(fmaxf(var, 0.0f)) + DU_EPS);
    }
}
__global__ void __launch_bounds__(BLK) k_bn_apply(const float *__restrict__ I, float *O, float *XH,
                                                  const float *__restrict__ W, const float *__restrict__ B,
                                                  const float *__restrict__ stat, long total, int C) {
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < total; z += (long)gridDim.x * BLK) {
        int c = (int)(z % C);
        float xh = (I[z] - stat[C + c]) * stat[c];
        XH[z] = xh; O[z] = xh * W[c] + B[c];
    }
}
__global__ void __launch_bounds__(BLK) k_dbn_stats(const float *__restrict__ DY, const float *__restrict__ XH,
                                                   float *stat, float *DW, float *DB, long NHW, int C, int train) {
    __shared__ float sm[4];
    const int c = blockIdx.x;
    float a = 0.0f, b = 0.0f;
    for (long k = threadIdx.x; k < NHW; k += BLK) { long z = k * C + c; float d = DY[z]; a += d; b = fmaf(d, XH[z], b); }
    a = block_sum(a, sm); b = block_sum(b, sm);
    if (threadIdx.x == 0) {
        float s1 = a / (float)NHW, s2 = b / (float)NHW;
        stat[C + c] = s1; stat[2 * C + c] = s2;
        if (train) { DB[c] += s1; DW[c] += s2; }
    }
}
__global__ void __launch_bounds__(BLK) k_dbn_apply(const float *__restrict__ W, const float *__restrict__ DY,
                                                   const float *__restrict__ XH, float *DX,
                                                   const float *__restrict__ stat, long total, int C) {
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < total; z += (long)gridDim.x * BLK) {
        int c = (int)(z % C);
        DX[z] = (stat[c] * W[c]) * (DY[z] - stat[C + c] - XH[z] * stat[2 * C + c]);
    }
}

// stage 1 of the chunked statistics: 16-byte loads where the tensors allow (T4K_BN_PART4=0: the scalar kernel)
template <int MODE>
static void launch_bn_part(const float *X, const float *Y, float *part, long NHW, int C, int rpc, long nch, hipStream_t hs) {
    static const int v4 = T4K_LAB_ENV("T4K_BN_PART4", 1);
    const bool vec = v4 && (C % 4) == 0 && ((((uintptr_t)X) | ((uintptr_t)(Y ? Y : X))) & 15) == 0;
    if (vec) T4K_LAUNCH(k_bn_part4<MODE>, dim3((unsigned)nch, (C + 63) / 64), dim3(BLK), 0, hs, X, Y, part, NHW, C, rpc);
    else     T4K_LAUNCH(k_bn_part<MODE>, dim3((unsigned)nch, (C + 63) / 64), dim3(BLK), 0, hs, X, Y, part, NHW, C, rpc);
}

// synchronised statistics for data-parallel runs (a communicator exists): chunk partials -> sums -> all-reduce -> finalise
template <int MODE>
static int bn_stats_sync(const float *X, const float *Y, float *stat, float *DW, float *DB, long NHW, int C, int train, t4k_stream_t s) {
    long nch = (NHW + 255) / 256; if (nch > 2048) nch = 2048;
    const int rpc = (int)((NHW + nch - 1) / nch); nch = (NHW + rpc - 1) / rpc;
    float *part = ws_for(s), *sums = part + (size_t)nch * 2 * C;
    if (((size_t)nch * 2 + 4) * C * sizeof(float) > st().ws_bytes / 2) return fail(T4K_ERR_NOMEM, "batchnorm workspace");
    launch_bn_part<MODE>(X, Y, part, NHW, C, rpc, nch, S(s));
    T4K_LAUNCH(k_bn_sums, dim3((C + 3) / 4), dim3(BLK), 0, S(s), part, sums, C, (int)nch);
    int rc = t4k_allreduce_sum(sums, 2L * C, s); if (rc != T4K_OK) return rc;
    T4K_LAUNCH(k_bn_fin_sync<MODE>, dim3((C + BLK - 1) / BLK), dim3(BLK), 0, S(s), sums, stat, DW, DB,
                       (float)NHW * (float)t4k_comm_world(), C, train);
    return T4K_OK;
}

} // namespace

namespace t4k {
// the batch-norm forward from chunk partials [nchunk][sum x | sum x^2][C] a producer left (t4k_conv2d_bn_fwd): finalise + apply
int bn_fwd_from_parts(const float *I, float *O, float *XH, const float *W, const float *B, float *stat, long NHW, int C, const float *part, int nchunk, hipStream_t hs) {
    T4K_LAUNCH(k_bn_fin<0>, dim3((C + 3) / 4), dim3(BLK), 0, hs, part, stat, (float *)nullptr, (float *)nullptr, NHW, C, nchunk, 0);
    T4K_LAUNCH(k_bn_apply, dim3(grid_for(NHW * C)), dim3(BLK), 0, hs, I, O, XH, W, B, stat, NHW * C, C);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
// the statistics alone: from a producer's chunk partials (nchunk > 0, in the stream's workspace) or by the passes of t4k_batchnorm_fwd
int bn_stats_for(const float *I, float *stat, int N, int HW, int C, const float *part, int nchunk, t4k_stream_t s);
}

extern "C" {

int t4k_reduce(int red_op, const float *src, long n, float avg, float *out, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!src || !out || n < 0) return fail(T4K_ERR_ARG, "t4k_reduce: bad argument");
    switch (red_op) {
    case T4K_RED_SUM:  return launch_reduce<R_SUM>(src, nullptr, n, 0.0f, out, S(s));
    case T4K_RED_NVAR: return launch_reduce<R_NVAR>(src, nullptr, n, avg, out, S(s));
    case T4K_RED_MAX:  return launch_reduce<R_MAX>(src, nullptr, n, 0.0f, out, S(s));
    case T4K_RED_MIN:  return launch_reduce<R_MIN>(src, nullptr, n, 0.0f, out, S(s));
    }
    return fail(T4K_ERR_ARG, "t4k_reduce: op %d", red_op);
}
int t4k_bce(const float *T, const float *O, long n, float *out, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!T || !O || !out || n < 0) return fail(T4K_ERR_ARG, "t4k_bce: bad argument");
    return launch_reduce<R_BCE>(O, T, n, 0.0f, out, S(s));
}
int t4k_nan_inf(const float *src, long n, int *cnt, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!src || !cnt) return fail(T4K_ERR_ARG, "t4k_nan_inf: null");
    T4K_HIP(hipMemsetAsync(cnt, 0, sizeof(int), S(s)));
    if (n > 0) T4K_LAUNCH(k_nan_inf, dim3(grid_for(n, 8)), dim3(BLK), 0, S(s), src, n, cnt);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_dot(const float *A, const float *B, float *O, float alpha, float beta, int K, int C, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!A || !B || !O || K < 0 || C < 1) return fail(T4K_ERR_ARG, "t4k_dot: bad argument");
    T4K_LAUNCH(k_dot, dim3(C), dim3(BLK), 0, S(s), A, B, O, alpha, beta, K, C);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_softmax(const float *I, float *O, int N, int C, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (N <= 0 || C <= 0) return T4K_OK;
    T4K_LAUNCH(k_softmax, dim3((N + 3) / 4), dim3(BLK), 0, S(s), I, O, N, C);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_logsoftmax(const float *I, float *O, int N, int C, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (N <= 0 || C <= 0) return T4K_OK;
    if (!I || !O) return fail(T4K_ERR_ARG, "t4k_logsoftmax: null tensor");
    T4K_LAUNCH(k_logsoftmax, dim3((N + BLK - 1) / BLK), dim3(BLK), 0, S(s), I, O, N, C);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_hit(const float *out, const float *hot, int N, int E, int *cnt, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!out || !hot || !cnt || N < 0 || E < 1) return fail(T4K_ERR_ARG, "t4k_hit: bad argument");
    if (E <= 32)       T4K_LAUNCH(k_hit<1>,  dim3(1), dim3(BLK), 0, S(s), out, hot, N, E, cnt, nullptr, nullptr);
    else if (E <= 256) T4K_LAUNCH(k_hit<8>,  dim3(1), dim3(BLK), 0, S(s), out, hot, N, E, cnt, nullptr, nullptr);
    else               T4K_LAUNCH(k_hit<64>, dim3(1), dim3(BLK), 0, S(s), out, hot, N, E, cnt, nullptr, nullptr);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_onehot_hit(const uint32_t *label, float *hot, const float *out, int N, int E, int *cnt, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!label || !out || !hot || !cnt || N < 0 || E < 1) return fail(T4K_ERR_ARG, "t4k_onehot_hit: bad argument");
    if (E <= 32)       T4K_LAUNCH(k_hit<1>,  dim3(1), dim3(BLK), 0, S(s), out, hot, N, E, cnt, label, hot);
    else if (E <= 256) T4K_LAUNCH(k_hit<8>,  dim3(1), dim3(BLK), 0, S(s), out, hot, N, E, cnt, label, hot);
    else               T4K_LAUNCH(k_hit<64>, dim3(1), dim3(BLK), 0, S(s), out, hot, N, E, cnt, label, hot);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_dlinear_db(const float *DY, float *DB, int N, int E0, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (N <= 0 || E0 <= 0) return T4K_OK;
    T4K_LAUNCH(k_dlinear_db, dim3((E0 + 63) / 64), dim3(BLK), 0, S(s), DY, DB, N, E0);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
// the statistics half of the batch-norm forward: stat_dev [0, C) = 1 / (sigma + eps), [C, 2C) = mean
static int bn_fwd_stats(const float *I, float *stat, int N, int HW, int C, t4k_stream_t s) {
    const long NHW = (long)N * HW;
    if (st().bn_sync && t4k_comm_world() > 0) {          // data parallel (opt-in, t4k_comm_sync_batchnorm): statistics over every rank's shard
        int rc = bn_stats_sync<0>(I, nullptr, stat, nullptr, nullptr, NHW, C, 0, s); if (rc != T4K_OK) return rc;
    } else if (NHW >= 2048) {                            // image-sized: chunked coalesced column sums + per-channel fold
        long nch = (NHW + 255) / 256; if (nch > 2048) nch = 2048;
        const int rpc = (int)((NHW + nch - 1) / nch); nch = (NHW + rpc - 1) / rpc;
        float *part = ws_for(s);
        if ((size_t)nch * 2 * C * sizeof(float) > st().ws_bytes / 2) return fail(T4K_ERR_NOMEM, "batchnorm workspace");
        launch_bn_part<0>(I, nullptr, part, NHW, C, rpc, nch, S(s));
        T4K_LAUNCH(k_bn_fin<0>, dim3((C + 3) / 4), dim3(BLK), 0, S(s), part, stat, (float *)nullptr, (float *)nullptr, NHW, C, (int)nch, 0);
    } else T4K_LAUNCH(k_bn_stats, dim3(C), dim3(BLK), 0, S(s), I, stat, NHW, C);
    return T4K_OK;
}
int t4k_batchnorm_fwd(const float *I, float *O, float *XH, const float *W, const float *B,
                      float *stat, int N, int HW, int C, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (N <= 0 || HW <= 0 || C <= 0) return fail(T4K_ERR_ARG, "t4k_batchnorm_fwd: shape");
    const long total = (long)N * HW * C;
    int rc = bn_fwd_stats(I, stat, N, HW, C, s); if (rc != T4K_OK) return rc;
    T4K_LAUNCH(k_bn_apply, dim3(grid_for(total)), dim3(BLK), 0, S(s), I, O, XH, W, B, stat, total, C);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_batchnorm_bwd(const float *W, const float *DY, const float *XH, float *DX,
                      float *DW, float *DB, float *stat, int N, int HW, int C, int train, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (N <= 0 || HW <= 0 || C <= 0) return fail(T4K_ERR_ARG, "t4k_batchnorm_bwd: shape");
    const long NHW = (long)N * HW, total = NHW * C;
    if (st().bn_sync && t4k_comm_world() > 0) {
        int rc = bn_stats_sync<1>(DY, XH, stat, DW, DB, NHW, C, train, s); if (rc != T4K_OK) return rc;
    } else if (NHW >= 2048) {
        long nch = (NHW + 255) / 256; if (nch > 2048) nch = 2048;
        const int rpc = (int)((NHW + nch - 1) / nch); nch = (NHW + rpc - 1) / rpc;
        float *part = ws_for(s);
        if ((size_t)nch * 2 * C * sizeof(float) > st().ws_bytes / 2) return fail(T4K_ERR_NOMEM, "batchnorm workspace");
        launch_bn_part<1>(DY, XH, part, NHW, C, rpc, nch, S(s));
        T4K_LAUNCH(k_bn_fin<1>, dim3((C + 3) / 4), dim3(BLK), 0, S(s), part, stat, DW, DB, NHW, C, (int)nch, train);
    } else T4K_LAUNCH(k_dbn_stats, dim3(C), dim3(BLK), 0, S(s), DY, XH, stat, DW, DB, NHW, C, train);
    T4K_LAUNCH(k_dbn_apply, dim3(grid_for(total)), dim3(BLK), 0, S(s), W, DY, XH, DX, stat, total, C);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}

} // extern "C"

namespace t4k {
int bn_stats_for(const float *I, float *stat, int N, int HW, int C, const float *part, int nchunk, t4k_stream_t s) {
    if (nchunk > 0) { T4K_LAUNCH(k_bn_fin<0>, dim3((C + 3) / 4), dim3(BLK), 0, S(s), part, stat, (float *)nullptr, (float *)nullptr, (long)N * HW, C, nchunk, 0); T4K_LAUNCH_CHECK(); return T4K_OK; }
    return bn_fwd_stats(I, stat, N, HW, C, s);
}
}
