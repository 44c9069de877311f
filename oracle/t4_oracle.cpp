/*
 * t4_oracle.cpp - CPU oracle: serial restatement of the reference's CUDA kernels.
 *
 * TEST INFRASTRUCTURE ONLY (see t4_oracle.h).  Every function cites the reference
 * file:line it follows (paths relative to the reference tree, src/...).
 * Where the reference sums with atomics in unspecified order, the oracle picks the
 * deterministic order "block index ascending"; fp32 accumulators are kept as fp32.
 */
#include "t4_oracle.h"
#include <cmath>
#include <cfloat>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>

namespace {

enum { OK = 0, ERR_ARG = -1, ERR_UNSUPPORTED = -4, ERR_SINGULAR = -6 };

/* math_op (src/t4math.h:25-56) */
enum { ABS = 0, NEG, EXP, LN, LOG, TANH, RELU, SIGM, SQRT, RCP, SAT, IDEN, FILL, GFILL,
       SCALE, POW, ADD, SUB, MUL, DIV };
/* t4_layer (src/nn/ntypes.h:16-36) */
enum { L_NONE = 0, L_CONV, L_LINEAR, L_FLATTEN, L_RELU, L_TANH, L_SIGMOID, L_SELU, L_LEAKYRL,
       L_ELU, L_DROPOUT, L_SOFTMAX, L_LOGSMAX, L_AVGPOOL, L_MAXPOOL, L_MINPOOL, L_BATCHNM,
       L_USAMPLE, L_DCONV };

const float DU_EPS = 1.0e-6f;          /* src/ten4_types.h:85 */
const int   DIM_SQ = 256;              /* T4_DIM_SQ, src/ten4_config.h:73 */
const int   BLOCKS_PER_DEV = 128;      /* src/t4base.h:129 */

inline int fork_grid(long n) {         /* GRID_BLKS, src/t4base.h:130-131 */
    long g = (n + DIM_SQ - 1) / DIM_SQ;
    if (g > BLOCKS_PER_DEV) g = BLOCKS_PER_DEV;
    if (g < 1) g = 1;
    return (int)g;
}
/* WARP_SUM (src/t4base.h:122-124): shfl_down tree over 32 lanes, lane 0 holds the result */
inline float warp_sum32(float *v) {
    for (int off = 16; off > 0; off >>= 1)
        for (int i = 0; i + off < 32; i++) v[i] += v[i + off];   /* lanes >= 32-off: don't care */
    return v[0];
}
/* the two-level block reduction shared by k_sum/k_nvar/k_bce (src/t4math.cu:23-46):
 * per-thread strided partials -> warp shuffle tree -> smem[32] -> warp 0 tree -> atomicAdd */
template <typename F>
float fork_reduce_sum(long n, F term) {
    const int  grid = fork_grid(n);
    const long step = (long)grid * DIM_SQ;
    float total = 0.0f;                                  /* *sum pre-zeroed by the host */
    std::vector<float> part(DIM_SQ);
    for (int b = 0; b < grid; b++) {                     /* atomic order: block ascending */
        for (int t = 0; t < DIM_SQ; t++) {
            float v = 0.0f;
            for (long j = (long)b * DIM_SQ + t; j < n; j += step) v += term(j);
            part[t] = v;
        }
        float wsum[32];
        for (int w = 0; w < 32; w++) wsum[w] = 0.0f;
        for (int w = 0; w < DIM_SQ / 32; w++) wsum[w] = warp_sum32(&part[w * 32]);
        total += warp_sum32(wsum);
    }
    return total;
}

/* ---- Philox4x32-10 (Salmon et al. SC'11), counter = element_index/4, key = seed ---- */
uint64_t g_seed = 0, g_off = 0;
inline void philox4x32_10(uint64_t ctr, uint64_t key, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
inline float u01(uint32_t x) {        /* (0,1], as curand_uniform: x*2^-32 + 2^-33 */
    return fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

bool conv_supported(int K, int S, int P) {   /* src/nn/forward.cu:142-151 */
    return (K == 1 && S == 1 && P == 0) || (K == 3 && S == 1 && P == 1) ||
           (K == 4 && S == 2 && P == 1) || (K == 5 && S == 1 && P == 2);
}

} // namespace

extern "C" {

/* k_sum src/t4math.cu:23-46, k_nvar :48-72, k_max/d__max :85-131
 * (host wrappers Tensor::sum/std/norm/max/min src/mu/tensor.cu:224-277) */
int t4o_reduce(int red_op, const float *src, long n, float avg, float *out) {
    if (!src || !out || n < 0) return ERR_ARG;
    switch (red_op) {
    case 0: *out = fork_reduce_sum(n, [&](long j) { return src[j]; }); break;
    case 1: *out = fork_reduce_sum(n, [&](long j) { float d = src[j] - avg; return d * d; }); break;
    case 2: { float m = -FLT_MAX; for (long j = 0; j < n; j++) m = fmaxf(m, src[j]); *out = m; } break;
    case 3: { float m =  FLT_MAX; for (long j = 0; j < n; j++) m = fminf(m, src[j]); *out = m; } break;
    default: return ERR_ARG;
    }
    return OK;
}
/* k_nan_inf src/t4math.cu:278-291 */
int t4o_nan_inf(const float *src, long n, int *cnt) {
    int c = 0;
    for (long j = 0; j < n; j++) if (std::isnan(src[j]) || std::isinf(src[j])) c++;
    *cnt = c;
    return OK;
}
/* k_copy src/t4math.cu:134-149 */
int t4o_copy(const float *src, float *dst, long n) {
    if (n > 0) memmove(dst, src, (size_t)n * sizeof(float));
    return OK;
}
/* k_transpose src/t4math.cu:150-159 */
int t4o_transpose(const float *src, float *dst, int H, int W, int C) {
    for (int i = 0; i < H; i++) for (int j = 0; j < W; j++) for (int c = 0; c < C; c++)
        dst[((long)H * j + i) * C + c] = src[((long)W * i + j) * C + c];
    return OK;
}
/* k_identity src/t4math.cu:160-170 */
int t4o_identity(float *T, int H, int W, int C) {
    for (int i = 0; i < H; i++) for (int j = 0; j < W; j++) for (int c = 0; c < C; c++)
        T[((long)W * i + j) * C + c] = (i == j) ? 1.0f : 0.0f;
    return OK;
}
/* k_math src/t4math.cu:173-202 (scalar macros src/t4math.h:60-104).  The CUDA fast
 * intrinsics (__expf, __logf, __powf) are restated with libm; tolerance covers the ulps. */
int t4o_math(int op, float *A, float v, long n) {
    const float LNX = 1.0e-12f;                       /* DU_LNX :172 */
    for (long j = 0; j < n; j++) {
        float a = A[j];
        switch (op) {
        case ABS:   A[j] = fabsf(a);                     break;
        case NEG:   A[j] = -a;                           break;
        case EXP:   A[j] = expf(a);                      break;
        case LN:    A[j] = logf(fmaxf(a, LNX));          break;
        case LOG:   A[j] = log10f(fmaxf(a, LNX));        break;
        case TANH:  A[j] = tanhf(a);                     break;
        case RELU:  A[j] = fmaxf(0.0f, a);               break;
        case SIGM:  A[j] = 1.0f / (1.0f + expf(-a));     break;
        case SQRT:  A[j] = sqrtf(fmaxf(a, 0.0f));        break;
        case RCP:   A[j] = 1.0f / a;                     break;
        case SAT:   A[j] = fminf(1.0f, fmaxf(0.0f, a));  break;
        case FILL:  A[j] = v;                            break;
        case GFILL: A[j] = v * (float)j / (float)n;      break;
        case SCALE: A[j] = a * v;                        break;
        case POW:   A[j] = powf(a, v);                   break;
        case ADD:   A[j] = a + v;                        break;
        case SUB:   A[j] = a - v;                        break;
        case MUL:   A[j] = a * v;                        break;
        case DIV:   A[j] = a / v;                        break;
        default:    return ERR_UNSUPPORTED;              /* "k_math op=%d not supported" :199 */
        }
    }
    return OK;
}
/* k_ts_op src/t4math.cu:206-218 */
int t4o_ts_op(int op, const float *A, float v, float *O, long n) {
    for (long j = 0; j < n; j++) {
        switch (op) {
        case ADD: O[j] = A[j] + v; break;
        case SUB: O[j] = A[j] - v; break;
        case MUL: O[j] = A[j] * v; break;
        case DIV: O[j] = A[j] / v; break;
        default:  return ERR_UNSUPPORTED;
        }
    }
    return OK;
}
/* k_tt_op src/t4math.cu:222-234 */
int t4o_tt_op(int op, const float *A, const float *B, float *O, long n) {
    for (long j = 0; j < n; j++) {
        switch (op) {
        case ADD: O[j] = A[j] + B[j]; break;
        case SUB: O[j] = A[j] - B[j]; break;
        case MUL: O[j] = A[j] * B[j]; break;
        case DIV: O[j] = A[j] / B[j]; break;
        default:  return ERR_UNSUPPORTED;
        }
    }
    return OK;
}
/* k_bce src/t4math.cu:248-274 */
int t4o_bce(const float *T, const float *O, long n, float *out) {
    *out = fork_reduce_sum(n, [&](long j) {
        float t = T[j], o = O[j];
        return t * logf(o + DU_EPS) + (1.0f - t) * logf(1.0f - o + DU_EPS);
    });
    return OK;
}
/* k_dot src/t4math.cu:309-365: 32 threads x VLEN 4 strided partials, smem tree */
int t4o_dot(const float *A, const float *B, float *O, float alpha, float beta, int K, int C) {
    for (int c = 0; c < C; c++) {
        float part[32];
        const int step = 32 * 4;
        for (int tx = 0; tx < 32; tx++) {
            float acc = 0.0f;
            int k = tx * 4;
            for (; k + step <= K; k += step)
                for (int v = 0; v < 4; v++) { long i = (long)(k + v) * C + c; acc += A[i] * B[i]; }
            for (int v = 0; v < 4; v++) if (k + v < K) { long i = (long)(k + v) * C + c; acc += A[i] * B[i]; }
            part[tx] = acc;
        }
        for (int half = 16; half > 0; half >>= 1)
            for (int tx = 0; tx < half; tx++) part[tx] += part[tx + half];
        O[c] = part[0] * alpha + O[c] * beta;
    }
    return OK;
}
/* k_gemm_tile_claude src/t4math.cu:478-583 (Tensor::gemm3 src/mu/tensor.cu:161-180):
 * fp32 accumulator, k ascending (nvcc contracts acc += a*b to FMA), epilogue :580 */
int t4o_gemm(const float *A, const float *B, float *O, float alpha, float beta,
             int tA, int tB, int M, int N, int K, int C) {
    if (M < 0 || N < 0 || K < 0 || C < 1) return ERR_ARG;
    for (int c = 0; c < C; c++)
        for (int m = 0; m < M; m++)
            for (int n = 0; n < N; n++) {
                float acc = 0.0f;
                for (int k = 0; k < K; k++) {
                    long ai = tA ? ((long)k * M + m) * C + c : ((long)m * K + k) * C + c;
                    long bi = tB ? ((long)n * K + k) * C + c : ((long)k * N + n) * C + c;
                    acc = fmaf(A[ai], B[bi], acc);
                }
                long z = ((long)m * N + n) * C + c;
                O[z] = acc * alpha + O[z] * beta;
            }
    return OK;
}
/* k_gemm src/t4math.cu:370-391 / k_gemm_claude :411-452: double accumulator, tA/tB ignored */
int t4o_gemm_f64acc(const float *A, const float *B, float *O, float alpha, float beta,
                    int M, int N, int K, int C) {
    for (int c = 0; c < C; c++)
        for (int m = 0; m < M; m++)
            for (int n = 0; n < N; n++) {
                double acc = 0.0;
                for (int k = 0; k < K; k++)
                    acc += A[((long)m * K + k) * C + c] * B[((long)k * N + n) * C + c];  /* float product */
                long z = ((long)m * N + n) * C + c;
                O[z] = (float)(alpha * acc + beta * O[z]);
            }
    return OK;
}
/* Tensor::gemm (word `gemm`, the reference's own host loop) src/mu/tensor.cu:97-123 */
int t4o_gemm_host_blocked(const float *A, const float *B, float *O, float alpha, float beta,
                          int H, int W, int Ka) {
    const int BLOCK = 32;
    for (long i = 0; i < (long)H * W; ++i) O[i] *= beta;
    for (int kk = 0; kk < Ka; kk += BLOCK)
        for (int mm = 0; mm < H; mm += BLOCK)
            for (int nn = 0; nn < W; nn += BLOCK)
                for (int k = kk; k < std::min(kk + BLOCK, Ka); ++k)
                    for (int i = mm; i < std::min(mm + BLOCK, H); ++i) {
                        float av = alpha * A[(long)i * Ka + k];
                        for (int j = nn; j < std::min(nn + BLOCK, W); ++j)
                            O[(long)i * W + j] += av * B[(long)k * W + j];
                    }
    return OK;
}

/* ---------------------------------------------------------------- linear algebra */
static int find_pivot(const float *A, int z, int K) {   /* k_find_pivot src/t4math.cu:742-773 */
    float val = -1.0f; int idx = z;
    for (int j = z; j < K; j++) { float v = fabsf(A[(long)j * K + z]); if (v > val) { val = v; idx = j; } }
    return (val < DU_EPS) ? -1 : idx;
}
static void swap_rows(float *A, float *I, int u, int z, int K) {  /* k_swap_rows :780-792 */
    for (int j = 0; j < K; j++) {
        std::swap(A[(long)z * K + j], A[(long)u * K + j]);
        if (I) std::swap(I[(long)z * K + j], I[(long)u * K + j]);
    }
}
/* Tensor::inverse src/mu/tensor.cu:344-369; k_diag :798-810, k_elim :819-836 */
int t4o_inverse(float *A, float *I, int K, int *status) {
    *status = 0;
    for (int z = 0; z < K; z++) {
        int u = find_pivot(A, z, K);
        if (u < 0) { *status = z + 1; return OK; }      /* "singular matrix at column z" */
        if (u != z) swap_rows(A, I, u, z, K);
        float r0 = A[(long)z * K + z];
        for (int j = 0; j < K; j++) { A[(long)z * K + j] /= r0; I[(long)z * K + j] /= r0; }
        for (int j = 0; j < K; j++) {
            if (j == z) continue;
            float r1 = A[(long)j * K + z];
            if (fabsf(r1) < DU_EPS) continue;
            for (int k = 0; k < K; k++) {
                A[(long)j * K + k] -= r1 * A[(long)z * K + k];
                I[(long)j * K + k] -= r1 * I[(long)z * K + k];
            }
        }
    }
    return OK;
}
/* Tensor::plu src/mu/tensor.cu:371-398; k_lu_col src/t4math.cu:854-869, k_pivot :886-902 */
int t4o_plu(float *A, float *I, int *piv, int K, int *status) {
    *status = 0;
    for (int z = 0; z < K; z++) {
        int u = find_pivot(A, z, K);
        piv[z] = u;
        if (u < 0) { *status = z + 1; return OK; }
        if (u != z) swap_rows(A, nullptr, u, z, K);
        float pivot = A[(long)z * K + z];
        for (int j = z + 1; j < K; j++) {
            float lik = A[(long)j * K + z] / pivot;
            A[(long)j * K + z] = lik;
            for (int k = z + 1; k < K; k++) A[(long)j * K + k] -= lik * A[(long)z * K + k];
        }
    }
    if (I && I != A) {
        for (int j = 0; j < K; j++)
            for (int k = 0; k < K; k++) {
                int pk = piv[k];
                if (pk != k) std::swap(I[(long)k * K + j], I[(long)pk * K + j]);
            }
    }
    return OK;
}
/* Tensor::lu_inverse src/mu/tensor.cu:400-417; k_fsub src/t4math.cu:903-917, k_bsub :919-933 */
int t4o_lu_inverse(float *A, float *I, int *piv, int K, int *status) {
    t4o_plu(A, I, piv, K, status);
    if (*status) return OK;
    for (int i = 0; i < K; i++) {
        for (int k = 1; k < K; k++) {
            float s = I[(long)k * K + i];
            for (int j = 0; j < k; j++) s -= A[(long)k * K + j] * I[(long)j * K + i];
            I[(long)k * K + i] = s;
        }
        for (int j = K - 1; j >= 0; j--) {
            float s = I[(long)j * K + i];
            for (int k = j + 1; k < K; k++) s -= A[(long)j * K + k] * I[(long)k * K + i];
            I[(long)j * K + i] = s / A[(long)j * K + j];
        }
    }
    return OK;
}
/* Tensor::lu src/mu/tensor.cu:419-429; k_lu src/t4math.cu:935-949 */
int t4o_lu_extract(float *LU, int get_u, int K) {
    for (int ty = 0; ty < K; ty++) for (int tx = 0; tx < K; tx++) {
        float *v = &LU[(long)ty * K + tx];
        if (get_u) { if (tx < ty) *v = 0.0f; }
        else { if (tx == ty) *v = 1.0f; else if (tx > ty) *v = 0.0f; }
    }
    return OK;
}
/* k_logdet src/t4math.cu:951-979 */
int t4o_logdet(const float *LU, int K, float *logdet, int *sign) {
    float acc = 0.0f; int sg = 1;
    for (int j = 0; j < K; j++) {
        float u = LU[(long)j * K + j];
        if (u < 0.0f) { sg = -sg; u = -u; }
        acc += logf(u);
    }
    *logdet = acc; *sign = sg;
    return OK;
}

/* ---------------------------------------------------------------------- RNG */
/* t4_rand_init / t4_rand src/util.cu:28-70: x = scale*(bias+u).  The cuRAND XORWOW
 * stream is not reproduced (the reference seeds from time(), src/sys.cpp:37); the
 * oracle and the HIP backend share the Philox4x32-10 definition instead. */
int      t4o_rand_init(uint64_t seed) { g_seed = seed; g_off = 0; return OK; }
uint64_t t4o_rand_offset(void)        { return g_off; }
int      t4o_rand_set_offset(uint64_t off) { g_off = off; return OK; }
int t4o_rand(float *d, long n, int opt, float bias, float scale) {
    const uint64_t base = g_off / 4;                   /* g_off is kept a multiple of 4 */
    for (long q = 0; q * 4 < n; q++) {
        uint32_t r[4];
        philox4x32_10(base + (uint64_t)q, g_seed, r);
        float v[4];
        if (opt == 1) {                                /* NORMAL: Box-Muller on (r0,r1), (r2,r3) */
            for (int p = 0; p < 2; p++) {
                float u1 = u01(r[2 * p]), u2 = u01(r[2 * p + 1]);
                float rad = sqrtf(-2.0f * logf(u1)), ang = 6.2831853071795865f * u2;
                v[2 * p] = rad * cosf(ang); v[2 * p + 1] = rad * sinf(ang);
            }
        } else for (int k = 0; k < 4; k++) v[k] = u01(r[k]);
        for (int k = 0; k < 4; k++) { long i = q * 4 + k; if (i < n) d[i] = scale * (bias + v[k]); }
    }
    g_off += (uint64_t)((n + 3) / 4) * 4;
    return OK;
}

/* data-parallel shard of the stream (include/t4k.h t4k_rand_set_shard): a dropout mask of n elements is elements
 * [rank*n, (rank+1)*n) of the whole batch's n*world-element draw */
static int g_shard_rank = 0, g_shard_world = 1;
int t4o_rand_set_shard(int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) return ERR_ARG;
    g_shard_rank = rank; g_shard_world = world; return OK;
}
uint64_t t4o_rand_seed(void) { return g_seed; }
int t4o_rand_shard_world(void) { return g_shard_world; }
int t4o_dropout_mask(float *mask, long n) {
    const uint64_t nq = (uint64_t)((n + 3) / 4), off0 = g_off;
    g_off = off0 + (uint64_t)g_shard_rank * nq * 4;
    int rc = t4o_rand(mask, n, 0, 0.0f, 1.0f);
    g_off = off0 + (uint64_t)g_shard_world * nq * 4;
    return rc;
}

/* ------------------------------------------------------------------------ nn */
/* k_bias src/nn/nmath.cu:27-35 */
int t4o_bias(const float *B, float *O, int N, int E0) {
    for (int n = 0; n < N; n++) for (int e = 0; e < E0; e++) O[(long)n * E0 + e] += B[e];
    return OK;
}
/* k_activate src/nn/nmath.cu:37-70 (SELU constants src/nn/nmath.h:32-33) */
int t4o_activate(int layer, const float *I, float *O, float *F, float alpha, long n) {
    const double SELU_L = 1.0507, SELU_LA = 1.7581;
    for (long j = 0; j < n; j++) {
        float i = I[j];
        switch (layer) {
        case L_RELU:    if (i > 0.0f) { F[j] = 1.0f; O[j] = i; } else { F[j] = 0.0f; O[j] = 0.0f; } break;
        case L_TANH:    i = tanhf(i); O[j] = i; F[j] = 1.0f - i * i; break;
        case L_SIGMOID: i = 1.0f / (1.0f + expf(-i)); O[j] = i; F[j] = i * (1.0f - i); break;
        case L_SELU:    /* positive branch yields O = i (not lambda*i): comma expression :56-58 */
            if (i > 0.0f) { F[j] = (float)SELU_L; O[j] = i; }
            else { F[j] = (float)(SELU_LA * (double)expf(i)); O[j] = (float)((double)F[j] - SELU_LA); }
            break;
        case L_LEAKYRL: if (i > 0.0f) { F[j] = 1.0f; O[j] = i; } else { F[j] = alpha; O[j] = alpha * i; } break;
        case L_ELU:     if (i > 0.0f) { F[j] = 1.0f; O[j] = i; }
                        else { F[j] = alpha * expf(i); O[j] = F[j] - alpha; } break;
        case L_DROPOUT: /* mask = rand > p, no 1/(1-p) rescale :65-67 */
            if (F[j] > alpha) { F[j] = 1.0f; O[j] = i; } else { F[j] = 0.0f; O[j] = 0.0f; } break;
        default: return ERR_UNSUPPORTED;
        }
    }
    return OK;
}
/* k_softmax_small / k_softmax src/nn/nmath.cu:74-169 (Model::_fsoftmax src/nn/forward.cu:230-243) */
int t4o_softmax(const float *I, float *O, int N, int C) {
    for (int n = 0; n < N; n++) {
        const float *s = I + (long)n * C; float *d = O + (long)n * C;
        float mx = -FLT_MAX;
        for (int c = 0; c < C; c++) mx = fmaxf(mx, s[c]);
        float sm = 0.0f;
        for (int c = 0; c < C; c++) { d[c] = expf(s[c] - mx); sm += d[c]; }
        for (int c = 0; c < C; c++) d[c] /= sm;
    }
    return OK;
}
/* k_batchnorm_1/2/3 src/nn/nmath.cu:177-264 (Model::_fbatchnorm src/nn/forward.cu:263-309) */
int t4o_batchnorm_fwd(const float *I, float *O, float *XH, const float *W, const float *B,
                      float *stat, int N, int HW, int C) {
    float *rvar = stat, *avg = stat + C;
    const long NHW = (long)N * HW;
    for (int c = 0; c < C; c++) {
        float sum = 0.0f, sq = 0.0f;
        for (int n = 0; n < N; n++) {                    /* per-(c,n) block partials, then atomics */
            float ts = 0.0f, tq = 0.0f;
            for (int j = 0; j < HW; j++) { float v = I[((long)n * HW + j) * C + c]; ts += v; tq += v * v; }
            sum += ts; sq += tq;
        }
        float b_avg = sum / (float)NHW;
        float b_var = sq / (float)NHW - b_avg * b_avg;
        avg[c]  = b_avg;
        rvar[c] = 1.0f / (sqrtf(fmaxf(b_var, 0.0f)) + DU_EPS);   /* eps outside sqrt :236 */
    }
    for (long k = 0; k < NHW; k++) for (int c = 0; c < C; c++) {
        long z = k * C + c;
        XH[z] = (I[z] - avg[c]) * rvar[c];
        O[z]  = XH[z] * W[c] + B[c];
    }
    return OK;
}
/* k_dbatchnorm_1/2/3 src/nn/nmath.cu:295-414 (Model::_bbatchnorm src/nn/backprop.cu:311-370) */
int t4o_batchnorm_bwd(const float *W, const float *DY, const float *XH, float *DX,
                      float *DW, float *DB, float *stat, int N, int HW, int C, int train) {
    float *rvar = stat, *s1 = stat + C, *s2 = stat + 2 * C;
    const long NHW = (long)N * HW;
    for (int c = 0; c < C; c++) {
        float a = 0.0f, b = 0.0f;
        for (long k = 0; k < NHW; k++) { long z = k * C + c; a += DY[z]; b += DY[z] * XH[z]; }
        s1[c] = a / (float)NHW; s2[c] = b / (float)NHW;
        if (train) { DB[c] += s1[c]; DW[c] += s2[c]; }   /* the MEANS are accumulated :378-381 */
    }
    for (long k = 0; k < NHW; k++) for (int c = 0; c < C; c++) {
        long z = k * C + c;
        DX[z] = (rvar[c] * W[c]) * (DY[z] - s1[c] - XH[z] * s2[c]);
    }
    return OK;
}
/* k_dlinear_db src/nn/nmath.cu:274-280 */
int t4o_dlinear_db(const float *DY, float *DB, int N, int E0) {
    for (int n = 0; n < N; n++) for (int e = 0; e < E0; e++) DB[e] += DY[(long)n * E0 + e];
    return OK;
}
/* k_conv2d<TS,KS,S,P> src/nn/nmath.tcu:34-104 (Model::_fconv src/nn/forward.cu:125-155):
 * output zeroed, then per (n,c1,c0) plane: sum = (c1==0 ? B : 0) + sum_{y,x} F*I, atomicAdd over c1 */
int t4o_conv2d_fwd(const float *I, float *O, const float *F, const float *B,
                   int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P) {
    if (!conv_supported(K, S, P)) return ERR_UNSUPPORTED;
    memset(O, 0, sizeof(float) * (size_t)N * H0 * W0 * C0);
    for (int n = 0; n < N; n++) {
        const float *nI = I + (long)n * H1 * W1 * C1; float *nO = O + (long)n * H0 * W0 * C0;
        for (int c1 = 0; c1 < C1; c1++) for (int c0 = 0; c0 < C0; c0++) {
            const long zf = ((long)c1 * K * K) * C0 + c0;
            for (int i0 = 0; i0 < H0; i0++) for (int j0 = 0; j0 < W0; j0++) {
                float sum = (c1 == 0) ? B[c0] : 0.0f;
                for (int y = 0; y < K; y++) for (int x = 0; x < K; x++) {
                    int gi = i0 * S + y - P, gj = j0 * S + x - P;
                    float v = (gi >= 0 && gi < H1 && gj >= 0 && gj < W1) ? nI[((long)W1 * gi + gj) * C1 + c1] : 0.0f;
                    sum += F[zf + (long)(y * K + x) * C0] * v;
                }
                nO[((long)W0 * i0 + j0) * C0 + c0] += sum;
            }
        }
    }
    return OK;
}
/* k_dconv2d<TS,KS,S,P> src/nn/nmath.tcu:211-338 (Model::_bconv src/nn/backprop.cu:152-191):
 * dX pre-zeroed; dX scatter uses the 180-degree flipped filter index (:304-305,321-324) */
int t4o_conv2d_bwd(const float *I, const float *DO, float *DX, const float *F,
                   float *DF, float *DB,
                   int N, int H1, int W1, int C1, int H0, int W0, int C0,
                   int K, int S, int P, int train) {
    if (!conv_supported(K, S, P)) return ERR_UNSUPPORTED;
    memset(DX, 0, sizeof(float) * (size_t)N * H1 * W1 * C1);
    for (int n = 0; n < N; n++) {
        const float *nI = I + (long)n * H1 * W1 * C1; const float *nO = DO + (long)n * H0 * W0 * C0;
        float *nDX = DX + (long)n * H1 * W1 * C1;
        for (int c1 = 0; c1 < C1; c1++) for (int c0 = 0; c0 < C0; c0++) {
            const long zf = (long)c1 * K * K * C0 + c0;
            for (int i0 = 0; i0 < H0; i0++) for (int j0 = 0; j0 < W0; j0++) {
                float dO = nO[((long)W0 * i0 + j0) * C0 + c0];
                if (train && c1 == 0) DB[c0] += dO;
                for (int ky = 0; ky < K; ky++) for (int kx = 0; kx < K; kx++) {
                    int gi = i0 * S + ky - P, gj = j0 * S + kx - P;
                    bool in = (gi >= 0 && gi < H1 && gj >= 0 && gj < W1);
                    if (in) nDX[((long)W1 * gi + gj) * C1 + c1] +=
                                F[zf + (long)((K - 1 - ky) * K + (K - 1 - kx)) * C0] * dO;
                    if (train && in) DF[zf + (long)(ky * K + kx) * C0] += dO * nI[((long)W1 * gi + gj) * C1 + c1];
                }
            }
        }
    }
    return OK;
}
/* Transposed convolution layer (word `dconv2d`).  The reference allocates it (Model::_iconv txn, src/nn/model.cpp:121-180) and routes its
 * forward to _bconv / its backward to _fconv (src/nn/forward.cu:110, backprop.cu:137) with the operands unswapped - code that never ran.
 * The oracle states the finished layer by its definition (the scatter form of that dispatch, no tap flip):
 *   O[n, i*S+ky-P, j*S+kx-P, co] = B[co] + sum_ci F[ci,ky,kx,co] * I[n,i,j,ci]          F = T4(C1,K,K,C0)
 * pinned against torch.nn.functional.conv_transpose2d in tests/test_oracle_vs_torch.py. */
int t4o_dconv2d_fwd(const float *I, float *O, const float *F, const float *B,
                    int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P) {
    if (!conv_supported(K, S, P)) return ERR_UNSUPPORTED;
    for (long z = 0; z < (long)N * H0 * W0; z++) for (int co = 0; co < C0; co++) O[z * C0 + co] = B[co];
    for (int n = 0; n < N; n++) for (int i = 0; i < H1; i++) for (int j = 0; j < W1; j++)
        for (int ky = 0; ky < K; ky++) for (int kx = 0; kx < K; kx++) {
            const int y = i * S + ky - P, x = j * S + kx - P;
            if (y < 0 || y >= H0 || x < 0 || x >= W0) continue;
            const float *in = I + (((long)n * H1 + i) * W1 + j) * C1;
            float *out = O + (((long)n * H0 + y) * W0 + x) * C0;
            for (int ci = 0; ci < C1; ci++) {
                const float v = in[ci]; const float *f = F + (((long)ci * K + ky) * K + kx) * C0;
                for (int co = 0; co < C0; co++) out[co] += f[co] * v;
            }
        }
    return OK;
}
int t4o_dconv2d_bwd(const float *I, const float *DO, float *DX, const float *F, float *DF, float *DB,
                    int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P, int train) {
    if (!conv_supported(K, S, P)) return ERR_UNSUPPORTED;
    if (DX) memset(DX, 0, sizeof(float) * (size_t)N * H1 * W1 * C1);
    for (int n = 0; n < N; n++) for (int i = 0; i < H1; i++) for (int j = 0; j < W1; j++)
        for (int ky = 0; ky < K; ky++) for (int kx = 0; kx < K; kx++) {
            const int y = i * S + ky - P, x = j * S + kx - P;
            if (y < 0 || y >= H0 || x < 0 || x >= W0) continue;
            const float *in = I + (((long)n * H1 + i) * W1 + j) * C1;
            const float *go = DO + (((long)n * H0 + y) * W0 + x) * C0;
            for (int ci = 0; ci < C1; ci++) {
                const long fo = (((long)ci * K + ky) * K + kx) * C0;
                float acc = 0.f;
                for (int co = 0; co < C0; co++) {
                    acc += F[fo + co] * go[co];
                    if (train && DF) DF[fo + co] += in[ci] * go[co];
                }
                if (DX) DX[(((long)n * H1 + i) * W1 + j) * C1 + ci] += acc;
            }
        }
    if (train && DB) for (long z = 0; z < (long)N * H0 * W0; z++) for (int co = 0; co < C0; co++) DB[co] += DO[z * C0 + co];
    return OK;
}
/* k_pool<KS> src/nn/nmath.tcu:122-186.  The reference reads full KSxKS tiles with no bounds
 * check (UB for odd H/W, SURVEY a-15); the oracle defines the edge: out-of-range cells are skipped. */
int t4o_pool(int layer, const float *I, float *O, int N, int H1, int W1, int H0, int W0, int C, int KS) {
    if (KS != 2 && KS != 3) return ERR_UNSUPPORTED;
    for (int n = 0; n < N; n++) for (int i0 = 0; i0 < H0; i0++) for (int j0 = 0; j0 < W0; j0++)
        for (int c = 0; c < C; c++) {
            float v = 0.0f; bool first = true;
            for (int y = 0; y < KS; y++) for (int x = 0; x < KS; x++) {
                int gi = i0 * KS + y, gj = j0 * KS + x;
                if (gi >= H1 || gj >= W1) continue;
                float t = I[(((long)n * H1 + gi) * W1 + gj) * C + c];
                switch (layer) {
                case L_USAMPLE: case L_AVGPOOL: v += t; break;
                case L_MAXPOOL: v = first ? t : fmaxf(t, v); break;
                case L_MINPOOL: v = first ? t : fminf(t, v); break;
                default: return ERR_UNSUPPORTED;
                }
                first = false;
            }
            if (layer == L_AVGPOOL || layer == L_USAMPLE) v /= (float)(KS * KS);
            O[(((long)n * H0 + i0) * W0 + j0) * C + c] = v;
        }
    return OK;
}
/* k_dpool<KS> src/nn/nmath.tcu:475-568: in place on the forward input buffer */
int t4o_dpool(int layer, float *I, const float *DY, int N, int H1, int W1, int H0, int W0, int C, int KS) {
    if (KS != 2 && KS != 3) return ERR_UNSUPPORTED;
    for (int n = 0; n < N; n++) for (int i0 = 0; i0 < H0; i0++) for (int j0 = 0; j0 < W0; j0++)
        for (int c = 0; c < C; c++) {
            float dy = DY[(((long)n * H0 + i0) * W0 + j0) * C + c];
            float best = 0.0f; float *argp = nullptr;
            for (int y = 0; y < KS; y++) for (int x = 0; x < KS; x++) {
                int gi = i0 * KS + y, gj = j0 * KS + x;
                if (gi >= H1 || gj >= W1) continue;
                float *px = &I[(((long)n * H1 + gi) * W1 + gj) * C + c];
                switch (layer) {
                case L_AVGPOOL: *px = dy / (float)(KS * KS); break;
                case L_USAMPLE: *px = dy; break;
                case L_MAXPOOL: { float dx = *px; *px = 0.0f;
                    if (!argp || dx > best) { best = dx; argp = px; } } break;   /* first max wins :545 */
                case L_MINPOOL: { float dx = *px; *px = 0.0f;
                    if (!argp || dx < best) { best = dx; argp = px; } } break;
                default: return ERR_UNSUPPORTED;
                }
            }
            if (argp) *argp = dy;
        }
    return OK;
}
/* k_sgd src/nn/nmath.cu:419-436 */
int t4o_sgd(float *G, float *DG, float *M, int Nw, float lr, float b, long n) {
    for (long j = 0; j < n; j++) {
        float dg = DG[j] / Nw;
        if (fabsf(b) < DU_EPS) G[j] -= lr * dg;
        else { float mi = M[j] = b * M[j] + (1.0f - b) * dg; G[j] -= lr * mi; }
        DG[j] = 0.0f;
    }
    return OK;
}
/* k_adam src/nn/nmath.cu:438-454: no bias correction, eps added after sqrt */
int t4o_adam(float *G, float *DG, float *M, float *V, float lr, float b1, float b2, long n) {
    for (long j = 0; j < n; j++) {
        const float dg = DG[j];
        const float mi = M[j] = b1 * M[j] + (1.0f - b1) * dg;
        const float vi = V[j] = b2 * V[j] + (1.0f - b2) * dg * dg;
        G[j] -= lr * mi / (sqrtf(vi) + DU_EPS);
        DG[j] = 0.0f;
    }
    return OK;
}
/* k_adamw src/nn/nmath.cu:456-472 */
int t4o_adamw(float *G, float *DG, float *M, float *V, float lr, float b1, float b2, float wd, long n) {
    for (long j = 0; j < n; j++) {
        const float dg = DG[j];
        const float mi = M[j] = b1 * M[j] + (1.0f - b1) * dg;
        const float vi = V[j] = b2 * V[j] + (1.0f - b2) * dg * dg;
        G[j] -= lr * (mi / (sqrtf(vi) + DU_EPS) - wd * dg);
        DG[j] = 0.0f;
    }
    return OK;
}
/* Model::onehot(Dataset&) src/nn/loss.cpp:47-72 */
int t4o_onehot(const uint32_t *label, float *hot, int N, int E) {
    memset(hot, 0, sizeof(float) * (size_t)N * E);
    for (int n = 0; n < N; n++) { uint32_t m = label[n]; hot[(long)n * E + (m < (uint32_t)E ? m : 0)] = 1.0f; }
    return OK;
}
/* Model::hit src/nn/loss.cpp:75-107 */
int t4o_hit(const float *out, const float *hot, int N, int E, int *cnt) {
    int c = 0;
    for (int n = 0; n < N; n++) {
        const float *o = out + (long)n * E;
        float m = o[0]; int i = 0;
        for (int e = 1; e < E; e++) if (o[e] > m) { m = o[e]; i = e; }
        c += (int)hot[(long)n * E + i];
    }
    *cnt = c;
    return OK;
}
/* Dataset::_load src/mu/dataset.cu:140-143 */
int t4o_u8_normalize(const uint8_t *src, float *dst, long n, float mean, float scale) {
    for (long i = 0; i < n; i++) dst[i] = ((float)(int)src[i] - mean) * scale;
    return OK;
}
/* Model::_flinear src/nn/forward.cu:157-198: Tensor::linear(tB=1) + k_bias */
int t4o_linear_fwd(const float *X, const float *W, const float *B, float *Y, int N, int E0, int E1) {
    t4o_gemm(X, W, Y, 1.0f, 0.0f, 0, 1, N, E0, E1, 1);
    return t4o_bias(B, Y, N, E0);
}
/* Model::_blinear src/nn/backprop.cu:193-254 */
int t4o_linear_bwd(const float *X, const float *W, const float *DY, float *DX,
                   float *DW, float *DB, int N, int E0, int E1, int train) {
    if (train) {
        t4o_dlinear_db(DY, DB, N, E0);
        t4o_gemm(DY, X, DW, 1.0f, 1.0f, 1, 0, E0, E1, N, 1);     /* dW[E0,E1] += dY^T[E0,N] @ X[N,E1] */
    }
    if (DX == X) {                                               /* in-place: X consumed above */
        std::vector<float> tmp((size_t)N * E1);
        t4o_gemm(DY, W, tmp.data(), 1.0f, 0.0f, 0, 0, N, E1, E0, 1);
        memcpy(DX, tmp.data(), tmp.size() * sizeof(float));
        return OK;
    }
    return t4o_gemm(DY, W, DX, 1.0f, 0.0f, 0, 0, N, E1, E0, 1);  /* dX[N,E1] = dY[N,E0] @ W[E0,E1] */
}

} // extern "C"
