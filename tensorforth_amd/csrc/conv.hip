// conv.hip - conv2d forward / backward and pooling for NHWC fp32 tensors.
// Reference: k_conv2d / k_dconv2d / k_pool / k_dpool, src/nn/nmath.tcu:34-568;
// host wrappers Model::_fconv src/nn/forward.cu:125-155, Model::_bconv src/nn/backprop.cu:152-191.
//
// Round-1 kernels (direct, VALU): the filter index is wave-uniform so filter taps are read
// through the scalar cache; every output is written exactly once (no memset + fp32
// atomicAdd over c1 as in the reference), dF/dB are reduced deterministically through
// workspace partials.  An MFMA implicit-GEMM path replaces the hot variants later.
#include "t4k_common.h"
#include <float.h>

using namespace t4k;

namespace {

// ------------------------------------------------------------------ forward
// one thread = one output pixel x CT output channels
template <int K, int S, int P, int CT>
__global__ void __launch_bounds__(BLK) k_conv_fwd(const float *__restrict__ I, float *__restrict__ O,
                                                  const float *__restrict__ F, const float *__restrict__ B,
                                                  int N, int H1, int W1, int C1, int H0, int W0, int C0) {
    const long npix = (long)N * H0 * W0;
    const long pix = (long)blockIdx.x * BLK + threadIdx.x;
    if (pix >= npix) return;
    const int j0 = (int)(pix % W0), i0 = (int)((pix / W0) % H0), n = (int)(pix / ((long)W0 * H0));
    const float *nI = I + (long)n * H1 * W1 * C1;
    for (int c0b = blockIdx.y * CT; c0b < C0; c0b += gridDim.y * CT) {
        float acc[CT];
#pragma unroll
        for (int t = 0; t < CT; t++) acc[t] = (c0b + t < C0) ? B[c0b + t] : 0.f;
        for (int c1 = 0; c1 < C1; c1++) {
#pragma unroll
            for (int ky = 0; ky < K; ky++) {
                const int gi = i0 * S + ky - P;
#pragma unroll
                for (int kx = 0; kx < K; kx++) {
                    const int gj = j0 * S + kx - P;
                    const float v = (gi >= 0 && gi < H1 && gj >= 0 && gj < W1) ? nI[((long)W1 * gi + gj) * C1 + c1] : 0.f;
                    const float *f = F + ((long)(c1 * K + ky) * K + kx) * C0 + c0b;     // wave-uniform
#pragma unroll
                    for (int t = 0; t < CT; t++) if (c0b + t < C0) acc[t] = fmaf(f[t], v, acc[t]);
                }
            }
        }
        float *o = O + pix * C0 + c0b;
#pragma unroll
        for (int t = 0; t < CT; t++) if (c0b + t < C0) o[t] = acc[t];
    }
}

// ------------------------------------------------------------------ backward: dX
// gather form of the reference's scatter: dX[gi,gj,c1] = sum over (ky,kx,c0) with
// i0*S+ky-P == gi, j0*S+kx-P == gj of F[c1,K-1-ky,K-1-kx,c0] * dO[i0,j0,c0]   (flipped index,
// nmath.tcu:304-305)
template <int K, int S, int P, int CT>
__global__ void __launch_bounds__(BLK) k_conv_dx(const float *__restrict__ DO, float *__restrict__ DX,
                                                 const float *__restrict__ F,
                                                 int N, int H1, int W1, int C1, int H0, int W0, int C0) {
    const long npix = (long)N * H1 * W1;
    const long pix = (long)blockIdx.x * BLK + threadIdx.x;
    if (pix >= npix) return;
    const int gj = (int)(pix % W1), gi = (int)((pix / W1) % H1), n = (int)(pix / ((long)W1 * H1));
    const float *nO = DO + (long)n * H0 * W0 * C0;
    for (int c1b = blockIdx.y * CT; c1b < C1; c1b += gridDim.y * CT) {
        float acc[CT];
#pragma unroll
        for (int t = 0; t < CT; t++) acc[t] = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ky++) {
            const int ti = gi + P - ky;
            if (ti < 0 || (ti % S) != 0) continue;
            const int i0 = ti / S; if (i0 >= H0) continue;
#pragma unroll
            for (int kx = 0; kx < K; kx++) {
                const int tj = gj + P - kx;
                if (tj < 0 || (tj % S) != 0) continue;
                const int j0 = tj / S; if (j0 >= W0) continue;
                const float *d = nO + ((long)W0 * i0 + j0) * C0;
                const int fo = ((K - 1 - ky) * K + (K - 1 - kx)) * C0;
                for (int c0 = 0; c0 < C0; c0++) {
                    const float dv = d[c0];
#pragma unroll
                    for (int t = 0; t < CT; t++)
                        if (c1b + t < C1) acc[t] = fmaf(F[(long)(c1b + t) * K * K * C0 + fo + c0], dv, acc[t]);
                }
            }
        }
        float *o = DX + pix * C1 + c1b;
#pragma unroll
        for (int t = 0; t < CT; t++) if (c1b + t < C1) o[t] = acc[t];
    }
}

// ------------------------------------------------------------------ backward: dF partials
// thread = one filter element (c1,ky,kx,c0); block.x = chunk of output pixels
template <int K, int S, int P>
__global__ void __launch_bounds__(BLK) k_conv_df(const float *__restrict__ I, const float *__restrict__ DO,
                                                 float *__restrict__ part,
                                                 int N, int H1, int W1, int C1, int H0, int W0, int C0,
                                                 long npix, int pix_per_chunk) {
    const int nf = C1 * K * K * C0;
    const int fi = blockIdx.y * BLK + threadIdx.x;
    if (fi >= nf) return;
    const int c0 = fi % C0, kx = (fi / C0) % K, ky = (fi / (C0 * K)) % K, c1 = fi / (C0 * K * K);
    const long p0 = (long)blockIdx.x * pix_per_chunk;
    const long p1 = min(npix, p0 + pix_per_chunk);
    float acc = 0.f;
    for (long pix = p0; pix < p1; pix++) {
        const int j0 = (int)(pix % W0), i0 = (int)((pix / W0) % H0), n = (int)(pix / ((long)W0 * H0));
        const int gi = i0 * S + ky - P, gj = j0 * S + kx - P;
        if (gi >= 0 && gi < H1 && gj >= 0 && gj < W1)
            acc = fmaf(I[(((long)n * H1 + gi) * W1 + gj) * C1 + c1], DO[pix * C0 + c0], acc);
    }
    part[(long)blockIdx.x * nf + fi] = acc;
}
// OUT[i] += sum_chunk part[chunk][i]   (chunk ascending: deterministic)
__global__ void __launch_bounds__(BLK) k_fold_add(const float *__restrict__ part, float *OUT, int n, int nchunk) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nchunk; k++) s += part[(long)k * n + i];
    OUT[i] += s;
}
// column sums: part[chunk][e] = sum over rows of the chunk of X[row][e]
__global__ void __launch_bounds__(BLK) k_colsum_part(const float *__restrict__ X, float *__restrict__ part,
                                                     long rows, int E, int rows_per_chunk, float *direct) {
    __shared__ float sm[4][64];
    const int ex = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int e = blockIdx.y * 64 + ex;
    const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float acc = 0.f;
    if (e < E) for (long r = r0 + ry; r < r1; r += 4) acc += X[r * E + e];
    sm[ry][ex] = acc;
    __syncthreads();
    if (ry == 0 && e < E) {
        const float t = (sm[0][ex] + sm[1][ex]) + (sm[2][ex] + sm[3][ex]);
        if (direct) direct[e] += t;                       // single chunk: accumulate in place
        else part[(long)blockIdx.x * E + e] = t;
    }
}

// ------------------------------------------------------------------ pooling
template <int KS>
__global__ void __launch_bounds__(BLK) k_pool(int layer, const float *__restrict__ I, float *__restrict__ O,
                                              int N, int H1, int W1, int H0, int W0, int C) {
    const long total = (long)N * H0 * W0 * C;
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < total; z += (long)gridDim.x * BLK) {
        const int c = (int)(z % C); long t = z / C;
        const int j0 = (int)(t % W0); t /= W0;
        const int i0 = (int)(t % H0); const int n = (int)(t / H0);
        float v = 0.f; bool first = true;
#pragma unroll
        for (int y = 0; y < KS; y++)
#pragma unroll
            for (int x = 0; x < KS; x++) {
                const int gi = i0 * KS + y, gj = j0 * KS + x;
                if (gi >= H1 || gj >= W1) continue;                 // defined edge (reference: UB)
                const float e = I[(((long)n * H1 + gi) * W1 + gj) * C + c];
                if (layer == T4K_L_MAXPOOL)      v = first ? e : fmaxf(e, v);
                else if (layer == T4K_L_MINPOOL) v = first ? e : fminf(e, v);
                else                             v += e;
                first = false;
            }
        if (layer == T4K_L_AVGPOOL || layer == T4K_L_USAMPLE) v /= (float)(KS * KS);
        O[z] = v;
    }
}
template <int KS>
__global__ void __launch_bounds__(BLK) k_dpool(int layer, float *I, const float *__restrict__ DY,
                                               int N, int H1, int W1, int H0, int W0, int C) {
    const long total = (long)N * H0 * W0 * C;
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < total; z += (long)gridDim.x * BLK) {
        const int c = (int)(z % C); long t = z / C;
        const int j0 = (int)(t % W0); t /= W0;
        const int i0 = (int)(t % H0); const int n = (int)(t / H0);
        const float dy = DY[z];
        float best = 0.f; long arg = -1;
#pragma unroll
        for (int y = 0; y < KS; y++)
#pragma unroll
            for (int x = 0; x < KS; x++) {
                const int gi = i0 * KS + y, gj = j0 * KS + x;
                if (gi >= H1 || gj >= W1) continue;
                const long a = (((long)n * H1 + gi) * W1 + gj) * C + c;
                if (layer == T4K_L_AVGPOOL)      I[a] = dy / (float)(KS * KS);
                else if (layer == T4K_L_USAMPLE) I[a] = dy;
                else {
                    const float dx = I[a]; I[a] = 0.f;
                    const bool better = (layer == T4K_L_MAXPOOL) ? (dx > best) : (dx < best);
                    if (arg < 0 || better) { best = dx; arg = a; }       // first extreme wins
                }
            }
        if (arg >= 0) I[arg] = dy;
    }
}

bool conv_supported(int K, int S, int P) {
    return (K == 1 && S == 1 && P == 0) || (K == 3 && S == 1 && P == 1) ||
           (K == 4 && S == 2 && P == 1) || (K == 5 && S == 1 && P == 2);
}

#define CONV_DISPATCH(KERN, CT, ...)                                                             \
    switch ((K << 8) | (S << 4) | P) {                                                           \
    case 0x110: hipLaunchKernelGGL((KERN<1, 1, 0, CT>), g, dim3(BLK), 0, hs, __VA_ARGS__); break; \
    case 0x311: hipLaunchKernelGGL((KERN<3, 1, 1, CT>), g, dim3(BLK), 0, hs, __VA_ARGS__); break; \
    case 0x421: hipLaunchKernelGGL((KERN<4, 2, 1, CT>), g, dim3(BLK), 0, hs, __VA_ARGS__); break; \
    case 0x512: hipLaunchKernelGGL((KERN<5, 1, 2, CT>), g, dim3(BLK), 0, hs, __VA_ARGS__); break; \
    }

} // namespace

namespace t4k {
// shared with fused.hip / reduce users: OUT[e] += sum_rows X[row][e], deterministic two-stage
int colsum_add(const float *X, float *OUT, long rows, int E, hipStream_t hs) {
    if (rows <= 0 || E <= 0) return T4K_OK;
    long want = (rows + 1023) / 1024; if (want > 256) want = 256; if (want < 1) want = 1;
    const int rpc = (int)((rows + want - 1) / want);
    const int nchunk = (int)((rows + rpc - 1) / rpc);
    float *part = (float *)st().ws + (8 << 20);            // second 32 MiB half of the workspace
    if ((size_t)nchunk * E * sizeof(float) > st().ws_bytes / 2) return fail(T4K_ERR_NOMEM, "colsum workspace");
    if (nchunk == 1) {
        hipLaunchKernelGGL(k_colsum_part, dim3(1, (E + 63) / 64), dim3(BLK), 0, hs, X, part, rows, E, rpc, OUT);
        return T4K_OK;
    }
    hipLaunchKernelGGL(k_colsum_part, dim3(nchunk, (E + 63) / 64), dim3(BLK), 0, hs, X, part, rows, E, rpc, (float *)nullptr);
    hipLaunchKernelGGL(k_fold_add, dim3((E + BLK - 1) / BLK), dim3(BLK), 0, hs, part, OUT, E, nchunk);
    return T4K_OK;
}
}

extern "C" {

int t4k_conv2d_fwd(const float *I, float *O, const float *F, const float *B,
                   int N, int H1, int W1, int C1, int H0, int W0, int C0,
                   int K, int S, int P, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!conv_supported(K, S, P))
        return fail(T4K_ERR_UNSUPPORTED, "nn#fconv kernel_size=%d stride=%d padding=%d not supported", K, S, P);
    if (!I || !O || !F || !B || N <= 0 || C0 <= 0 || C1 <= 0) return fail(T4K_ERR_ARG, "t4k_conv2d_fwd: bad argument");
    hipStream_t hs = t4k::S(s);
    const long npix = (long)N * H0 * W0;
    dim3 g((unsigned)((npix + BLK - 1) / BLK), 1);
    if (C0 <= 16) { CONV_DISPATCH(k_conv_fwd, 16, I, O, F, B, N, H1, W1, C1, H0, W0, C0) }
    else          { CONV_DISPATCH(k_conv_fwd, 32, I, O, F, B, N, H1, W1, C1, H0, W0, C0) }
    T4K_LAUNCH_CHECK();
    return T4K_OK;
}

int t4k_conv2d_bwd(const float *I, const float *DO, float *DX, const float *F, float *DF, float *DB,
                   int N, int H1, int W1, int C1, int H0, int W0, int C0,
                   int K, int S, int P, int train, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!conv_supported(K, S, P))
        return fail(T4K_ERR_UNSUPPORTED, "nn#bconv kernel_size=%d stride=%d padding=%d not supported", K, S, P);
    if (!I || !DO || !DX || !F || N <= 0) return fail(T4K_ERR_ARG, "t4k_conv2d_bwd: bad argument");
    hipStream_t hs = t4k::S(s);
    const long npix0 = (long)N * H0 * W0, npix1 = (long)N * H1 * W1;
    if (train) {
        if (!DF || !DB) return fail(T4K_ERR_ARG, "t4k_conv2d_bwd: train needs DF/DB");
        // dB[c0] += sum dO ; dF += sum I*dO  (before dX is written: DX may alias I in the host layer)
        int rc = colsum_add(DO, DB, npix0, C0, hs); if (rc) return rc;
        const int nf = C1 * K * K * C0;
        long want = (npix0 + 255) / 256; if (want > 512) want = 512; if (want < 1) want = 1;
        const int ppc = (int)((npix0 + want - 1) / want);
        const int nchunk = (int)((npix0 + ppc - 1) / ppc);
        float *part = (float *)st().ws;
        if ((size_t)nchunk * nf * sizeof(float) > st().ws_bytes / 2) return fail(T4K_ERR_NOMEM, "conv dF workspace");
        dim3 g(nchunk, (nf + BLK - 1) / BLK);
        switch ((K << 8) | (S << 4) | P) {
        case 0x110: hipLaunchKernelGGL((k_conv_df<1, 1, 0>), g, dim3(BLK), 0, hs, I, DO, part, N, H1, W1, C1, H0, W0, C0, npix0, ppc); break;
        case 0x311: hipLaunchKernelGGL((k_conv_df<3, 1, 1>), g, dim3(BLK), 0, hs, I, DO, part, N, H1, W1, C1, H0, W0, C0, npix0, ppc); break;
        case 0x421: hipLaunchKernelGGL((k_conv_df<4, 2, 1>), g, dim3(BLK), 0, hs, I, DO, part, N, H1, W1, C1, H0, W0, C0, npix0, ppc); break;
        case 0x512: hipLaunchKernelGGL((k_conv_df<5, 1, 2>), g, dim3(BLK), 0, hs, I, DO, part, N, H1, W1, C1, H0, W0, C0, npix0, ppc); break;
        }
        hipLaunchKernelGGL(k_fold_add, dim3((nf + BLK - 1) / BLK), dim3(BLK), 0, hs, part, DF, nf, nchunk);
    }
    {
        dim3 g((unsigned)((npix1 + BLK - 1) / BLK), 1);
        if (C1 <= 4)       { CONV_DISPATCH(k_conv_dx, 4,  DO, DX, F, N, H1, W1, C1, H0, W0, C0) }
        else if (C1 <= 16) { CONV_DISPATCH(k_conv_dx, 16, DO, DX, F, N, H1, W1, C1, H0, W0, C0) }
        else               { CONV_DISPATCH(k_conv_dx, 32, DO, DX, F, N, H1, W1, C1, H0, W0, C0) }
    }
    T4K_LAUNCH_CHECK();
    return T4K_OK;
}

int t4k_pool(int layer, const float *I, float *O, int N, int H1, int W1, int H0, int W0, int C, int KS, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (KS != 2 && KS != 3) return fail(T4K_ERR_UNSUPPORTED, "nn#fpool kernel_size=%d not supported", KS);
    if (layer != T4K_L_AVGPOOL && layer != T4K_L_MAXPOOL && layer != T4K_L_MINPOOL && layer != T4K_L_USAMPLE)
        return fail(T4K_ERR_UNSUPPORTED, "t4k_pool: layer %d", layer);
    const long total = (long)N * H0 * W0 * C; if (total <= 0) return T4K_OK;
    if (KS == 2) hipLaunchKernelGGL(k_pool<2>, dim3(grid_for(total)), dim3(BLK), 0, S(s), layer, I, O, N, H1, W1, H0, W0, C);
    else         hipLaunchKernelGGL(k_pool<3>, dim3(grid_for(total)), dim3(BLK), 0, S(s), layer, I, O, N, H1, W1, H0, W0, C);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_dpool(int layer, float *I, const float *DY, int N, int H1, int W1, int H0, int W0, int C, int KS, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (KS != 2 && KS != 3) return fail(T4K_ERR_UNSUPPORTED, "nn#bpool kernel_size=%d not supported", KS);
    if (layer != T4K_L_AVGPOOL && layer != T4K_L_MAXPOOL && layer != T4K_L_MINPOOL && layer != T4K_L_USAMPLE)
        return fail(T4K_ERR_UNSUPPORTED, "t4k_dpool: layer %d", layer);
    const long total = (long)N * H0 * W0 * C; if (total <= 0) return T4K_OK;
    if (KS == 2) hipLaunchKernelGGL(k_dpool<2>, dim3(grid_for(total)), dim3(BLK), 0, S(s), layer, I, DY, N, H1, W1, H0, W0, C);
    else         hipLaunchKernelGGL(k_dpool<3>, dim3(grid_for(total)), dim3(BLK), 0, S(s), layer, I, DY, N, H1, W1, H0, W0, C);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}

} // extern "C"
