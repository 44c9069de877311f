// dconv.hip - transposed convolution layer (word `dconv2d`, t4_layer L_DCONV).
//
// The reference allocates the layer (Model::_iconv with txn = true, src/nn/model.cpp:121-180: filter T4(C1,K,K,C0), bias T1(C0),
// output (H1-1)*S - 2P + K + P0) and dispatches its forward to the convolution BACKWARD routine and its backward to the
// convolution FORWARD routine (src/nn/forward.cu:110 `L_DCONV: _bconv(in, out)`, src/nn/backprop.cu:137 `L_DCONV: _fconv(in, out)`),
// but passes (in, out) unswapped, so that code reads the not-yet-written output and indexes k_dconv2d with H0 > H1 - it never ran.
// Here the layer is finished the way that dispatch intends, on the same (4,2,1) kernels:
//   forward   O[n, i*S+ky-P, j*S+kx-P, co] = B[co] + sum_{ci} F[ci,ky,kx,co] * I[n,i,j,ci]      (no tap flip; = torch ConvTranspose2d
//             with weight[ci][co][ky][kx] = F[ci,ky,kx,co])        -> the conv dX kernel on the "virtual" conv O -> I
//   backward  dX[n,i,j,ci] = sum_{co,taps} F[ci,ky,kx,co] * dO[n, i*S+ky-P, j*S+kx-P, co]        -> the conv forward kernel
//             dF[ci,ky,kx,co] += sum_n I[n,i,j,ci] * dO[n, i*S+ky-P, j*S+kx-P, co]               -> the conv dF kernel
//             dB[co] += sum dO[..., co]                                                           -> column sum
// The conv kernels want the virtual conv's filter as T4(C0,K,K,C1) (input channel major): a transposed copy of F is made per call
// (C1*K*K*C0 elements, one small launch), with the taps flipped where the kernel applies the reference's flip itself (quirk a-11).
#include "t4k_common.h"

using namespace t4k;

namespace t4k { int colsum_add(const float *X, float *OUT, long rows, int E, hipStream_t hs); }

namespace {

// dst[(co*K+ky)*K+kx][ci] (op)= src[(ci*K+ky')*K+kx'][co], (ky',kx') = flip ? (K-1-ky, K-1-kx) : (ky,kx);  ACC: dst += (gradient fold-back,
// where dst is the layer's T4(C1,K,K,C0) gradient and src the virtual conv's T4(C0,K,K,C1) one: same formula with the roles of C0 / C1 swapped)
template <bool ACC>
__global__ void __launch_bounds__(BLK) k_filter_xpose(const float *__restrict__ src, float *dst, int Cs, int Cd, int K, int flip) {
    const long n = (long)Cs * K * K * Cd;                // src is [Cs][K][K][Cd], dst is [Cd][K][K][Cs]
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < n; z += (long)gridDim.x * BLK) {
        const int cs = (int)(z % Cs); long r = z / Cs;   // z indexes dst: [cd][ky][kx][cs]
        const int kx = (int)(r % K); r /= K;
        const int ky = (int)(r % K); const int cd = (int)(r / K);
        const int sy = flip ? K - 1 - ky : ky, sx = flip ? K - 1 - kx : kx;
        const float v = src[(((long)cs * K + sy) * K + sx) * Cd + cd];
        if (ACC) dst[z] += v; else dst[z] = v;
    }
}

float *g_buf = nullptr; size_t g_cap = 0;               // two filter-sized scratch tensors (transposed filter, virtual-conv gradient)
int scratch(size_t floats, float **a, float **b) {
    if (2 * floats > g_cap) {
        if (st().capturing) return fail(T4K_ERR_UNSUPPORTED, "dconv2d: first call inside a graph capture");
        if (g_buf) { (void)hipDeviceSynchronize(); (void)hipFree(g_buf); g_buf = nullptr; g_cap = 0; }
        const size_t want = 2 * floats + 1024;
        if (hipMalloc((void **)&g_buf, want * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return fail(T4K_ERR_NOMEM, "dconv2d scratch"); }
        g_cap = want;
    }
    *a = g_buf; *b = g_buf + ((floats + 63) & ~(size_t)63);
    return T4K_OK;
}

} // namespace

extern "C" {

int t4k_dconv2d_fwd(const float *I, float *O, const float *F, const float *B,
                    int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!I || !O || !F || !B || N <= 0 || C0 <= 0 || C1 <= 0 || H1 <= 0 || W1 <= 0) return fail(T4K_ERR_ARG, "t4k_dconv2d_fwd: bad argument");
    if ((H0 - K + 2 * P) / S + 1 != H1 || (W0 - K + 2 * P) / S + 1 != W1) return fail(T4K_ERR_ARG, "t4k_dconv2d_fwd: output %dx%d does not map back to %dx%d", H0, W0, H1, W1);
    const size_t nf = (size_t)C1 * K * K * C0;
    float *ft, *unused; int rc = scratch(nf + C1 + 64, &ft, &unused); if (rc) return rc;
    hipStream_t hs = t4k::S(s);
    T4K_LAUNCH(k_filter_xpose<false>, dim3(grid_for((long)nf)), dim3(BLK), 0, hs, F, ft, C1, C0, K, 1);   // flipped: the dX kernel flips back
    T4K_LAUNCH_CHECK();
    // virtual conv: input O [N,H0,W0,C0] -> output I [N,H1,W1,C1]; its dX, given "dO" = I, is the transposed convolution
    rc = t4k_conv2d_bwd(O, I, O, ft, nullptr, nullptr, N, H0, W0, C0, H1, W1, C1, K, S, P, 0, s); if (rc) return rc;
    return t4k_bias(B, O, N * H0 * W0, C0, s);
}

int t4k_dconv2d_bwd(const float *I, const float *DO, float *DX, const float *F, float *DF, float *DB,
                    int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P, int train, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!I || !DO || !F || N <= 0 || C0 <= 0 || C1 <= 0) return fail(T4K_ERR_ARG, "t4k_dconv2d_bwd: bad argument");
    if ((DF == nullptr) != (DB == nullptr)) return fail(T4K_ERR_ARG, "t4k_dconv2d_bwd: DF and DB go together");
    const size_t nf = (size_t)C1 * K * K * C0;
    float *ft, *dfv; int rc = scratch(nf + C1 + 64, &ft, &dfv); if (rc) return rc;   // + room for a C1-long vector behind each
    hipStream_t hs = t4k::S(s);
    if (train && DF) {
        // dF of the virtual conv (input dO, output-gradient I) is the layer's dF with the channel roles swapped; dB is the column sum of dO
        T4K_HIP(hipMemsetAsync(dfv, 0, nf * sizeof(float), hs));
        float *dbv = dfv + nf;                           // C1 floats of the scratch tail (the virtual conv's bias gradient: discarded)
        if ((size_t)(dbv - g_buf) + (size_t)C1 > g_cap) return fail(T4K_ERR_NOMEM, "dconv2d scratch (C1 = %d)", C1);
        T4K_HIP(hipMemsetAsync(dbv, 0, (size_t)C1 * sizeof(float), hs));
        rc = t4k_conv2d_bwd(DO, I, nullptr, F /* unused: no dX */, dfv, dbv, N, H0, W0, C0, H1, W1, C1, K, S, P, 1, s); if (rc) return rc;
        T4K_LAUNCH(k_filter_xpose<true>, dim3(grid_for((long)nf)), dim3(BLK), 0, hs, dfv, DF, C0, C1, K, 0);
        T4K_LAUNCH_CHECK();
        rc = colsum_add(DO, DB, (long)N * H0 * W0, C0, hs); if (rc) return rc;
    }
    if (DX) {
        T4K_LAUNCH(k_filter_xpose<false>, dim3(grid_for((long)nf)), dim3(BLK), 0, hs, F, ft, C1, C0, K, 0);
        T4K_LAUNCH_CHECK();
        float *zb = dfv + nf;                            // zero bias for the plain convolution (re-zeroed: dF may have used the slot)
        if ((size_t)(zb - g_buf) + (size_t)C1 > g_cap) return fail(T4K_ERR_NOMEM, "dconv2d scratch (C1 = %d)", C1);
        T4K_HIP(hipMemsetAsync(zb, 0, (size_t)C1 * sizeof(float), hs));
        rc = t4k_conv2d_fwd(DO, DX, ft, zb, N, H0, W0, C0, H1, W1, C1, K, S, P, s); if (rc) return rc;
    }
    return T4K_OK;
}

} // extern "C"
