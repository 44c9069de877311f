// elementwise.hip - HBM-bound streaming kernels (copy / transpose / unary / binary /
// activation+mask / bias / u8 normalise / one-hot).  16 B per lane where the pointers
// allow it, grid-stride over at most 2048 workgroups (256 CUs x 8).
// Reference: src/t4math.cu:134-234, src/nn/nmath.cu:27-70, src/nn/loss.cpp:47-72,
// src/mu/dataset.cu:140-143.
#include "t4k_common.h"

using namespace t4k;

namespace {

// ---------------------------------------------------------------- unary / scalar ops
template <int OP>
__device__ __forceinline__ float math1(float a, float v, long j, long n) {
    switch (OP) {
    case T4K_ABS:   return fabsf(a);
    case T4K_NEG:   return -a;
    case T4K_EXP:   return __expf(a);
    case T4K_LN:    return __logf(fmaxf(a, 1.0e-12f));
    case T4K_LOG:   return __log10f(fmaxf(a, 1.0e-12f));
    case T4K_TANH:  return tanhf(a);
    case T4K_RELU:  return fmaxf(0.0f, a);
    case T4K_SIGM:  return 1.0f / (1.0f + expf(-a));
    case T4K_SQRT:  return sqrtf(fmaxf(a, 0.0f));
    case T4K_RCP:   return 1.0f / a;
    case T4K_SAT:   return fminf(1.0f, fmaxf(0.0f, a));
    case T4K_FILL:  return v;
    case T4K_GFILL: return v * (float)j / (float)n;
    case T4K_SCALE: return a * v;
    case T4K_POW:   return powf(a, v);
    case T4K_ADD:   return a + v;
    case T4K_SUB:   return a - v;
    case T4K_MUL:   return a * v;
    case T4K_DIV:   return a / v;
    case T4K_SIN:   return sinf(a);        // the reference switch has no SIN/COS case;
    case T4K_COS:   return cosf(a);        // here the words work
    }
    return a;
}

template <int OP>
__global__ void __launch_bounds__(BLK) k_math(float *__restrict__ A, float v, long n, bool vec) {
    const long tx = (long)blockIdx.x * BLK + threadIdx.x, step = (long)gridDim.x * BLK;
    if (vec) {
        const long n4 = n >> 2;
        float4 *A4 = reinterpret_cast<float4 *>(A);
        for (long q = tx; q < n4; q += step) {
            float4 a = A4[q];
            a.x = math1<OP>(a.x, v, 4 * q, n);     a.y = math1<OP>(a.y, v, 4 * q + 1, n);
            a.z = math1<OP>(a.z, v, 4 * q + 2, n); a.w = math1<OP>(a.w, v, 4 * q + 3, n);
            A4[q] = a;
        }
        for (long j = (n4 << 2) + tx; j < n; j += step) A[j] = math1<OP>(A[j], v, j, n);
    } else {
        for (long j = tx; j < n; j += step) A[j] = math1<OP>(A[j], v, j, n);
    }
}

template <int OP> __device__ __forceinline__ float bin(float a, float b) {
    switch (OP) {
    case T4K_ADD: return a + b;
    case T4K_SUB: return a - b;
    case T4K_MUL: return a * b;
    case T4K_DIV: return a / b;
    }
    return a;
}
template <int OP>
__global__ void __launch_bounds__(BLK) k_ts(const float *A, float v, float *O, long n, bool vec) {
    const long tx = (long)blockIdx.x * BLK + threadIdx.x, step = (long)gridDim.x * BLK;
    if (vec) {
        const long n4 = n >> 2;
        for (long q = tx; q < n4; q += step) {
            float4 a = reinterpret_cast<const float4 *>(A)[q];
            a.x = bin<OP>(a.x, v); a.y = bin<OP>(a.y, v); a.z = bin<OP>(a.z, v); a.w = bin<OP>(a.w, v);
            reinterpret_cast<float4 *>(O)[q] = a;
        }
        for (long j = (n4 << 2) + tx; j < n; j += step) O[j] = bin<OP>(A[j], v);
    } else for (long j = tx; j < n; j += step) O[j] = bin<OP>(A[j], v);
}
template <int OP>
__global__ void __launch_bounds__(BLK) k_tt(const float *A, const float *B, float *O, float *O2, long n, bool vec) {
    const long tx = (long)blockIdx.x * BLK + threadIdx.x, step = (long)gridDim.x * BLK;
    if (vec) {
        const long n4 = n >> 2;
        for (long q = tx; q < n4; q += step) {
            float4 a = reinterpret_cast<const float4 *>(A)[q];
            float4 b = reinterpret_cast<const float4 *>(B)[q];
            a.x = bin<OP>(a.x, b.x); a.y = bin<OP>(a.y, b.y); a.z = bin<OP>(a.z, b.z); a.w = bin<OP>(a.w, b.w);
            reinterpret_cast<float4 *>(O)[q] = a;
            if (O2) reinterpret_cast<float4 *>(O2)[q] = a;
        }
        for (long j = (n4 << 2) + tx; j < n; j += step) { const float v = bin<OP>(A[j], B[j]); O[j] = v; if (O2) O2[j] = v; }
    } else for (long j = tx; j < n; j += step) { const float v = bin<OP>(A[j], B[j]); O[j] = v; if (O2) O2[j] = v; }
}

__global__ void __launch_bounds__(BLK) k_copy(const float *__restrict__ src, float *__restrict__ dst, long n, bool vec) {
    const long tx = (long)blockIdx.x * BLK + threadIdx.x, step = (long)gridDim.x * BLK;
    if (vec) {
        const long n4 = n >> 2;
        for (long q = tx; q < n4; q += step) reinterpret_cast<float4 *>(dst)[q] = reinterpret_cast<const float4 *>(src)[q];
        for (long j = (n4 << 2) + tx; j < n; j += step) dst[j] = src[j];
    } else for (long j = tx; j < n; j += step) dst[j] = src[j];
}

// transpose through a 64x65 LDS tile: both the read and the write are coalesced along
// the fastest (w*C+c) axis.  One block per 64x64 (i,j) tile per channel group.
__global__ void __launch_bounds__(BLK) k_transpose(const float *__restrict__ src, float *__restrict__ dst, int H, int W, int C) {
    __shared__ float tile[64][65];
    const int c  = blockIdx.z;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;        // 64 x 4
    for (int r = ty; r < 64; r += 4) {
        int i = i0 + r, j = j0 + tx;
        if (i < H && j < W) tile[r][tx] = src[((long)W * i + j) * C + c];
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        int j = j0 + r, i = i0 + tx;
        if (i < H && j < W) dst[((long)H * j + i) * C + c] = tile[tx][r];
    }
}
__global__ void __launch_bounds__(BLK) k_identity(float *T, int H, int W, int C) {
    const long n = (long)H * W * C;
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < n; z += (long)gridDim.x * BLK) {
        long p = z / C; int i = (int)(p / W), j = (int)(p % W);
        T[z] = (i == j) ? 1.0f : 0.0f;
    }
}

// ---------------------------------------------------------------- activation + mask
template <int L>
__device__ __forceinline__ void act1(float i, float f_in, float alpha, float &o, float &f) {
    switch (L) {
    case T4K_L_RELU:    if (i > 0.0f) { f = 1.0f; o = i; } else { f = 0.0f; o = 0.0f; } break;
    case T4K_L_TANH:    o = tanhf(i); f = 1.0f - o * o; break;
    case T4K_L_SIGMOID: o = 1.0f / (1.0f + expf(-i)); f = o * (1.0f - o); break;
    case T4K_L_SELU:    if (i > 0.0f) { f = (float)1.0507; o = i; }
                        else { f = (float)(1.7581 * (double)__expf(i)); o = (float)((double)f - 1.7581); } break;
    case T4K_L_LEAKYRL: if (i > 0.0f) { f = 1.0f; o = i; } else { f = alpha; o = alpha * i; } break;
    case T4K_L_ELU:     if (i > 0.0f) { f = 1.0f; o = i; } else { f = alpha * __expf(i); o = f - alpha; } break;
    case T4K_L_DROPOUT: if (f_in > alpha) { f = 1.0f; o = i; } else { f = 0.0f; o = 0.0f; } break;
    }
}
template <int L>
__global__ void __launch_bounds__(BLK) k_activate(const float *I, float *O, float *F, float alpha, long n, bool vec) {
    const long tx = (long)blockIdx.x * BLK + threadIdx.x, step = (long)gridDim.x * BLK;
    if (vec) {
        const long n4 = n >> 2;
        for (long q = tx; q < n4; q += step) {
            float4 i = reinterpret_cast<const float4 *>(I)[q], o, f;
            float4 fi = (L == T4K_L_DROPOUT) ? reinterpret_cast<const float4 *>(F)[q] : make_float4(0, 0, 0, 0);
            act1<L>(i.x, fi.x, alpha, o.x, f.x); act1<L>(i.y, fi.y, alpha, o.y, f.y);
            act1<L>(i.z, fi.z, alpha, o.z, f.z); act1<L>(i.w, fi.w, alpha, o.w, f.w);
            reinterpret_cast<float4 *>(O)[q] = o; reinterpret_cast<float4 *>(F)[q] = f;
        }
        for (long j = (n4 << 2) + tx; j < n; j += step) { float o, f; act1<L>(I[j], F[j], alpha, o, f); O[j] = o; F[j] = f; }
    } else for (long j = tx; j < n; j += step) { float o, f; act1<L>(I[j], F[j], alpha, o, f); O[j] = o; F[j] = f; }
}

__global__ void __launch_bounds__(BLK) k_bias(const float *__restrict__ B, float *O, long total, int E0) {
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < total; z += (long)gridDim.x * BLK)
        O[z] += B[z % E0];
}
__global__ void __launch_bounds__(BLK) k_u8norm(const uint8_t *__restrict__ src, float *dst, long n, float mean, float scale) {
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < n; z += (long)gridDim.x * BLK)
        dst[z] = ((float)(int)src[z] - mean) * scale;
}
// one batch of a dataset off its pinned staging slot: pixels four to a lane ((x - mean) * scale, 16-byte stores) and the 4-byte labels
// in the same launch (dataset.cu:64-121 does two host-to-device copies and a normalise pass)
__global__ void __launch_bounds__(BLK) k_stage_batch(const uint8_t *__restrict__ src, float *dst, long n, float mean, float scale,
                                                     const uint32_t *__restrict__ lab_src, uint32_t *lab_dst, int nlab, int vec) {
    const long t0 = (long)blockIdx.x * BLK + threadIdx.x, step = (long)gridDim.x * BLK;
    if (t0 < nlab) lab_dst[t0] = lab_src[t0];
    if (vec) {
        const long n4 = n >> 2;
        for (long z = t0; z < n4; z += step) {
            const uint32_t w = reinterpret_cast<const uint32_t *>(src)[z];
            float4 o;
            o.x = ((float)(int)(w & 255u) - mean) * scale;         o.y = ((float)(int)((w >> 8) & 255u) - mean) * scale;
            o.z = ((float)(int)((w >> 16) & 255u) - mean) * scale; o.w = ((float)(int)(w >> 24) - mean) * scale;
            reinterpret_cast<float4 *>(dst)[z] = o;
        }
        for (long z = (n4 << 2) + t0; z < n; z += step) dst[z] = ((float)(int)src[z] - mean) * scale;
    } else
        for (long z = t0; z < n; z += step) dst[z] = ((float)(int)src[z] - mean) * scale;
}
// Model::broadcast backprop.cu:17-29: a [N,1] target spread over the output width, O[n,e] = T[n]
__global__ void __launch_bounds__(BLK) k_broadcast_rows(const float *__restrict__ T, float *O, int N, int E) {
    const long total = (long)N * E;
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < total; z += (long)gridDim.x * BLK) O[z] = T[z / E];
}
// backprop's start when the output layer is an activation with a derivative mask (Model::backprop: `out = target` copy, then
// _bactivate `in = out * mask`, backprop.cu:43-53,256-263): both tensors from one pass over the target
__global__ void __launch_bounds__(BLK) k_copy_mask(const float *__restrict__ T, const float *__restrict__ M, float *OUT, float *IN, long n) {
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < n; z += (long)gridDim.x * BLK) { const float t = T[z]; OUT[z] = t; IN[z] = t * M[z]; }
}
__global__ void __launch_bounds__(BLK) k_onehot(const uint32_t *__restrict__ label, float *hot, int N, int E) {
    const long total = (long)N * E;
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < total; z += (long)gridDim.x * BLK) {
        int n = (int)(z / E), e = (int)(z % E);
        uint32_t m = label[n]; if (m >= (uint32_t)E) m = 0;
        hot[z] = (e == (int)m) ? 1.0f : 0.0f;
    }
}

} // namespace

#define VEC3(a, b, c) (aligned16(a) && aligned16(b) && aligned16(c))

extern "C" {

int t4k_copy(const float *src, float *dst, long n, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0) return T4K_OK;
    if (!src || !dst) return fail(T4K_ERR_ARG, "t4k_copy: null");
    bool vec = aligned16(src) && aligned16(dst);
    T4K_LAUNCH(k_copy, dim3(grid_for(n, 4)), dim3(BLK), 0, S(s), src, dst, n, vec);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_transpose(const float *src, float *dst, int H, int W, int C, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (H <= 0 || W <= 0 || C <= 0) return fail(T4K_ERR_ARG, "t4k_transpose: shape");
    dim3 g((W + 63) / 64, (H + 63) / 64, C);
    T4K_LAUNCH(k_transpose, g, dim3(BLK), 0, S(s), src, dst, H, W, C);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_identity(float *dst, int H, int W, int C, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (H <= 0 || W <= 0 || C <= 0) return fail(T4K_ERR_ARG, "t4k_identity: shape");
    T4K_LAUNCH(k_identity, dim3(grid_for((long)H * W * C)), dim3(BLK), 0, S(s), dst, H, W, C);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}

#define MATH_CASE(OP) case OP: T4K_LAUNCH(k_math<OP>, dim3(g), dim3(BLK), 0, S(s), A, v, n, vec); break
int t4k_math(int op, float *A, float v, long n, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0) return T4K_OK;
    if (!A) return fail(T4K_ERR_ARG, "t4k_math: null");
    const bool vec = aligned16(A); const int g = grid_for(n, 4);
    switch (op) {
    MATH_CASE(T4K_ABS); MATH_CASE(T4K_NEG); MATH_CASE(T4K_EXP); MATH_CASE(T4K_LN); MATH_CASE(T4K_LOG);
    MATH_CASE(T4K_TANH); MATH_CASE(T4K_RELU); MATH_CASE(T4K_SIGM); MATH_CASE(T4K_SQRT); MATH_CASE(T4K_RCP);
    MATH_CASE(T4K_SAT); MATH_CASE(T4K_FILL); MATH_CASE(T4K_GFILL); MATH_CASE(T4K_SCALE); MATH_CASE(T4K_POW);
    MATH_CASE(T4K_ADD); MATH_CASE(T4K_SUB); MATH_CASE(T4K_MUL); MATH_CASE(T4K_DIV);
    MATH_CASE(T4K_SIN); MATH_CASE(T4K_COS);
    default: return fail(T4K_ERR_UNSUPPORTED, "k_math op=%d not supported", op);
    }
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
#define TS_CASE(OP) case OP: T4K_LAUNCH(k_ts<OP>, dim3(g), dim3(BLK), 0, S(s), A, v, O, n, vec); break
int t4k_ts_op(int op, const float *A, float v, float *O, long n, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0) return T4K_OK;
    if (!A || !O) return fail(T4K_ERR_ARG, "t4k_ts_op: null");
    const bool vec = aligned16(A) && aligned16(O); const int g = grid_for(n, 4);
    switch (op) { TS_CASE(T4K_ADD); TS_CASE(T4K_SUB); TS_CASE(T4K_MUL); TS_CASE(T4K_DIV);
    default: return fail(T4K_ERR_UNSUPPORTED, "k_ts_op op=%d not supported", op); }
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
#define TT_CASE(OP) case OP: T4K_LAUNCH(k_tt<OP>, dim3(g), dim3(BLK), 0, S(s), A, B, O, O2, n, vec); break
int t4k_tt_op(int op, const float *A, const float *B, float *O, long n, t4k_stream_t s) { return t4k_tt_op2(op, A, B, O, nullptr, n, s); }
int t4k_tt_op2(int op, const float *A, const float *B, float *O, float *O2, long n, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0) return T4K_OK;
    if (!A || !B || !O) return fail(T4K_ERR_ARG, "t4k_tt_op: null");
    const bool vec = VEC3(A, B, O) && (!O2 || aligned16(O2)); const int g = grid_for(n, 4);
    switch (op) { TT_CASE(T4K_ADD); TT_CASE(T4K_SUB); TT_CASE(T4K_MUL); TT_CASE(T4K_DIV);
    default: return fail(T4K_ERR_UNSUPPORTED, "k_tt_op op=%d not supported", op); }
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
#define ACT_CASE(L) case L: T4K_LAUNCH(k_activate<L>, dim3(g), dim3(BLK), 0, S(s), I, O, F, alpha, n, vec); break
int t4k_activate(int layer, const float *I, float *O, float *F, float alpha, long n, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0) return T4K_OK;
    if (!I || !O || !F) return fail(T4K_ERR_ARG, "t4k_activate: null");
    const bool vec = VEC3(I, O, F); const int g = grid_for(n, 4);
    switch (layer) {
    ACT_CASE(T4K_L_RELU); ACT_CASE(T4K_L_TANH); ACT_CASE(T4K_L_SIGMOID); ACT_CASE(T4K_L_SELU);
    ACT_CASE(T4K_L_LEAKYRL); ACT_CASE(T4K_L_ELU); ACT_CASE(T4K_L_DROPOUT);
    default: return fail(T4K_ERR_UNSUPPORTED, "k_activate layer=%d not supported", layer);
    }
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_bias(const float *B, float *O, int N, int E0, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (N <= 0 || E0 <= 0) return T4K_OK;
    const long total = (long)N * E0;
    T4K_LAUNCH(k_bias, dim3(grid_for(total)), dim3(BLK), 0, S(s), B, O, total, E0);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_u8_normalize(const uint8_t *src, float *dst, long n, float mean, float scale, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0) return T4K_OK;
    T4K_LAUNCH(k_u8norm, dim3(grid_for(n)), dim3(BLK), 0, S(s), src, dst, n, mean, scale);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_stage_batch(const uint8_t *src, float *dst, long n, float mean, float scale,
                    const uint32_t *lab_src, uint32_t *lab_dst, int nlab, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0 && nlab <= 0) return T4K_OK;
    if (n < 0 || nlab < 0 || (n > 0 && (!src || !dst)) || (nlab > 0 && (!lab_src || !lab_dst))) return fail(T4K_ERR_ARG, "t4k_stage_batch: bad argument");
    const int vec = ((uintptr_t)src & 3) == 0 && aligned16(dst);
    const long lanes = std::max<long>(vec ? (n + 3) >> 2 : n, nlab);
    T4K_LAUNCH(k_stage_batch, dim3(grid_for(lanes)), dim3(BLK), 0, S(s), src, dst, n, mean, scale, lab_src, lab_dst, nlab, vec);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_broadcast_rows(const float *T, float *O, int N, int E, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (N <= 0 || E <= 0) return T4K_OK;
    if (!T || !O) return fail(T4K_ERR_ARG, "t4k_broadcast_rows: null tensor");
    T4K_LAUNCH(k_broadcast_rows, dim3(grid_for((long)N * E)), dim3(BLK), 0, S(s), T, O, N, E);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_copy_mask(const float *T, const float *MASK, float *OUT, float *IN, long n, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0) return T4K_OK;
    if (!T || !MASK || !OUT || !IN) return fail(T4K_ERR_ARG, "t4k_copy_mask: null tensor");
    T4K_LAUNCH(k_copy_mask, dim3(grid_for(n)), dim3(BLK), 0, S(s), T, MASK, OUT, IN, n);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_onehot(const uint32_t *label, float *hot, int N, int E, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (N <= 0 || E <= 0) return T4K_OK;
    T4K_LAUNCH(k_onehot, dim3(grid_for((long)N * E)), dim3(BLK), 0, S(s), label, hot, N, E);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}

} // extern "C"
