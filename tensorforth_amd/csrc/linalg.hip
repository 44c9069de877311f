// linalg.hip - Gauss-Jordan inverse, PLU, LU inverse, L/U extraction, log-determinant.
// Reference: k_find_pivot .. k_logdet src/t4math.cu:742-979 and the host loops
// Tensor::inverse/plu/lu_inverse/lu/det src/mu/tensor.cu:344-456, which launch 3-4 kernels
// and one D2H copy per pivot column.  Here the whole column loop runs inside ONE
// workgroup-resident kernel (the example matrices are 3x3 .. 4x4; K up to a few hundred is
// fine), so the host issues one launch and reads one status word.
#include "t4k_common.h"

using namespace t4k;

namespace {

// argmax_j |A[j,z]| for j in [z,K): lowest index wins ties; result broadcast via LDS
__device__ int block_find_pivot(const float *A, int z, int K, float *s_val, int *s_idx) {
    const int tx = threadIdx.x;
    float val = -1.0f; int idx = z;
    for (int j = tx + z; j < K; j += BLK) { float v = fabsf(A[(long)j * K + z]); if (v > val) { val = v; idx = j; } }
    s_val[tx] = val; s_idx[tx] = idx;
    __syncthreads();
    for (int half = BLK >> 1; half > 0; half >>= 1) {
        if (tx < half) {
            float v2 = s_val[tx + half]; int i2 = s_idx[tx + half];
            if (v2 > s_val[tx] || (v2 == s_val[tx] && i2 < s_idx[tx])) { s_val[tx] = v2; s_idx[tx] = i2; }
        }
        __syncthreads();
    }
    const int r = (s_val[0] < DU_EPS) ? -1 : s_idx[0];
    __syncthreads();
    return r;
}
__device__ void block_swap_rows(float *A, float *I, int u, int z, int K) {
    for (int j = threadIdx.x; j < K; j += BLK) {
        float t = A[(long)z * K + j]; A[(long)z * K + j] = A[(long)u * K + j]; A[(long)u * K + j] = t;
        if (I) { float q = I[(long)z * K + j]; I[(long)z * K + j] = I[(long)u * K + j]; I[(long)u * K + j] = q; }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(BLK) k_inverse(float *A, float *I, int K, int *status) {
    __shared__ float s_val[BLK]; __shared__ int s_idx[BLK];
    __shared__ float s_r0;
    if (threadIdx.x == 0) *status = 0;
    for (int z = 0; z < K; z++) {
        const int u = block_find_pivot(A, z, K, s_val, s_idx);
        if (u < 0) { if (threadIdx.x == 0) *status = z + 1; return; }
        if (u != z) block_swap_rows(A, I, u, z, K);
        if (threadIdx.x == 0) s_r0 = A[(long)z * K + z];
        __syncthreads();
        const float r0 = s_r0;
        for (int j = threadIdx.x; j < K; j += BLK) { A[(long)z * K + j] /= r0; I[(long)z * K + j] /= r0; }
        __syncthreads();
        for (int j = threadIdx.x; j < K; j += BLK) {            // one thread per row, as k_elim
            if (j == z) continue;
            const float r1 = A[(long)j * K + z];
            if (fabsf(r1) < DU_EPS) continue;
            for (int k = 0; k < K; k++) {
                A[(long)j * K + k] -= r1 * A[(long)z * K + k];
                I[(long)j * K + k] -= r1 * I[(long)z * K + k];
            }
        }
        __syncthreads();
    }
}

__device__ bool block_plu(float *A, float *I, int *piv, int K, int *status, float *s_val, int *s_idx) {
    if (threadIdx.x == 0) *status = 0;
    for (int z = 0; z < K; z++) {
        const int u = block_find_pivot(A, z, K, s_val, s_idx);
        if (threadIdx.x == 0) piv[z] = u;
        if (u < 0) { if (threadIdx.x == 0) *status = z + 1; __syncthreads(); return false; }
        if (u != z) block_swap_rows(A, nullptr, u, z, K);
        const float pivot = A[(long)z * K + z];
        __syncthreads();
        for (int j = z + 1 + threadIdx.x; j < K; j += BLK) {    // k_lu_col: rows below the pivot
            const float lik = A[(long)j * K + z] / pivot;
            A[(long)j * K + z] = lik;
            for (int k = z + 1; k < K; k++) A[(long)j * K + k] -= lik * A[(long)z * K + k];
        }
        __syncthreads();
    }
    if (I && I != A) {                                          // k_pivot: permute rows of I, one thread per column
        for (int j = threadIdx.x; j < K; j += BLK)
            for (int k = 0; k < K; k++) {
                const int pk = piv[k];
                if (pk != k) { float t = I[(long)k * K + j]; I[(long)k * K + j] = I[(long)pk * K + j]; I[(long)pk * K + j] = t; }
            }
        __syncthreads();
    }
    return true;
}
__global__ void __launch_bounds__(BLK) k_plu(float *A, float *I, int *piv, int K, int *status) {
    __shared__ float s_val[BLK]; __shared__ int s_idx[BLK];
    block_plu(A, I, piv, K, status, s_val, s_idx);
}
__global__ void __launch_bounds__(BLK) k_lu_inverse(float *A, float *I, int *piv, int K, int *status) {
    __shared__ float s_val[BLK]; __shared__ int s_idx[BLK];
    if (!block_plu(A, I, piv, K, status, s_val, s_idx)) return;
    for (int i = threadIdx.x; i < K; i += BLK) {                // one thread per RHS column (k_fsub, k_bsub)
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
}
__global__ void __launch_bounds__(BLK) k_lu_extract(float *LU, int get_u, int K) {
    const long n = (long)K * K;
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < n; z += (long)gridDim.x * BLK) {
        const int ty = (int)(z / K), tx = (int)(z % K);
        if (get_u) { if (tx < ty) LU[z] = 0.f; }
        else { if (tx == ty) LU[z] = 1.f; else if (tx > ty) LU[z] = 0.f; }
    }
}
__global__ void __launch_bounds__(BLK) k_logdet(const float *LU, int K, float *logdet, int *sign) {
    __shared__ float s_acc[BLK]; __shared__ int s_sgn[BLK];
    float acc = 0.f; int sg = 1;
    for (int j = threadIdx.x; j < K; j += BLK) {
        float u = LU[(long)j * K + j];
        if (u < 0.f) { sg = -sg; u = -u; }
        acc += logf(u);
    }
    s_acc[threadIdx.x] = acc; s_sgn[threadIdx.x] = sg;
    __syncthreads();
    for (int half = BLK >> 1; half > 0; half >>= 1) {
        if (threadIdx.x < half) { s_acc[threadIdx.x] += s_acc[threadIdx.x + half]; s_sgn[threadIdx.x] *= s_sgn[threadIdx.x + half]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { *logdet = s_acc[0]; *sign = s_sgn[0]; }
}

} // namespace

extern "C" {

int t4k_inverse(float *A, float *I, int K, int *status_dev, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!A || !I || !status_dev || K <= 0) return fail(T4K_ERR_ARG, "t4k_inverse: bad argument");
    T4K_LAUNCH(k_inverse, dim3(1), dim3(BLK), 0, S(s), A, I, K, status_dev);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_plu(float *A, float *I, int *piv_dev, int K, int *status_dev, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!A || !piv_dev || !status_dev || K <= 0) return fail(T4K_ERR_ARG, "t4k_plu: bad argument");
    T4K_LAUNCH(k_plu, dim3(1), dim3(BLK), 0, S(s), A, I, piv_dev, K, status_dev);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_lu_inverse(float *A, float *I, int *piv_dev, int K, int *status_dev, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!A || !I || !piv_dev || !status_dev || K <= 0) return fail(T4K_ERR_ARG, "t4k_lu_inverse: bad argument");
    T4K_LAUNCH(k_lu_inverse, dim3(1), dim3(BLK), 0, S(s), A, I, piv_dev, K, status_dev);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_lu_extract(float *LU, int get_u, int K, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!LU || K <= 0) return fail(T4K_ERR_ARG, "t4k_lu_extract: bad argument");
    T4K_LAUNCH(k_lu_extract, dim3(grid_for((long)K * K)), dim3(BLK), 0, S(s), LU, get_u, K);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_logdet(const float *LU, int K, float *logdet_dev, int *sign_dev, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!LU || !logdet_dev || !sign_dev || K <= 0) return fail(T4K_ERR_ARG, "t4k_logdet: bad argument");
    T4K_LAUNCH(k_logdet, dim3(1), dim3(BLK), 0, S(s), LU, K, logdet_dev, sign_dev);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}

} // extern "C"
