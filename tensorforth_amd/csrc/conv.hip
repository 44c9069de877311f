// conv.hip - conv2d forward / backward as implicit GEMMs on the gfx950 matrix cores
// (v_mfma_f32_32x32x2_f32), plus pooling, for NHWC fp32 tensors.
// Reference: k_conv2d / k_dconv2d / k_pool / k_dpool, src/nn/nmath.tcu:34-568;
// host wrappers Model::_fconv src/nn/forward.cu:125-155, Model::_bconv src/nn/backprop.cu:152-191.
//
// The reference launches one 16x16 block per (n, c1, c0) plane tile, re-stages the same input
// patch for every c0 and accumulates across c1 with fp32 atomicAdd into a pre-zeroed output.
// Here each of the three contractions is one dense MFMA GEMM whose A operand is gathered on the
// fly (implicit im2col):
//   forward  O [pix0 , c0]  = sum_{ky,kx,c1} I [pix0 shifted, c1] * F [c1,ky,kx,c0]      (+ bias)
//   dX       dX[pix1 , c1]  = sum_{ky,kx,c0} dO[pix1 shifted, c0] * F [c1,K-1-ky,K-1-kx,c0]
//   dF|dB    dF[tap  , c0] += sum_{pix0}     I [pix0 shifted by tap] * dO[pix0, c0]   (row `ntaps` = dB)
// One wave owns a 32 (pixels or taps) x 32 (channels) accumulator.  The filter slice is staged in
// LDS in MFMA-B order; the k-pair of one MFMA is two adjacent channels of the same tap, so the two
// lane halves read adjacent floats.  Every output element is written once (no memset, no atomics);
// dF/dB are reduced wave -> workgroup (LDS) -> workspace slabs -> one fold launch, in fixed order.
#include "t4k_common.h"
#include <float.h>

namespace t4k { bool conv_thin_df(const float *I, const float *DO, float *part, size_t part_bytes, int N, int H, int W, int C1, int C0, int *nslice, hipStream_t hs); }
namespace t4k { bool conv_thin_fwd(const float *I, float *ICOPY, float *O, const float *F, const float *B, int N, int H, int W, int C1, int C0, hipStream_t hs,
                                  float *bn_part = nullptr, size_t bn_part_floats = 0, int *bn_chunks = nullptr);
                int bn_stats_for(const float *I, float *stat, int N, int HW, int C, const float *part, int nchunk, t4k_stream_t s);
                int bn_fwd_from_parts(const float *I, float *O, float *XH, const float *W, const float *B, float *stat, long NHW, int C, const float *part, int nchunk, hipStream_t hs); }
namespace t4k { bool conv_img_block_fwd(const float *I, float *ICOPY, float *O, const float *F, const float *B, const t4k_poolblock *blk,
                                        int N, int H, int W, int C1, int C0, hipStream_t hs); }
using namespace t4k;

namespace t4k {                                   // conv_big.hip: LDS-staged MFMA GEMM tiling for many channels
bool conv_big_ok(int Cin, int Cout);
template <bool BWD>
void launch_conv_big(int K, int S, int P, hipStream_t hs, const float *X, float *Y, float *Y2, const float *F, const float *B,
                     int N, int Hx, int Wx, int Cin, int Hy, int Wy, int Cout, int C0f, float *bn_part = nullptr, size_t bn_part_floats = 0, int *bn_chunks = nullptr);
int launch_conv_big_df(int K, int S, int P, hipStream_t hs, const float *I, const float *DO, float *part, size_t part_floats,
                       int N, int H1, int W1, int C1, int H0, int W0, int C0);
int colsum_add(const float *X, float *OUT, long rows, int E, hipStream_t hs);
}

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LDS_FILTER_FLOATS = 8192;          // 32 KiB filter slice per workgroup

// element-wise run that follows the convolution (dropout/activation -> 2x2 pool -> activation -> flatten copy, see fused.hip),
// applied in the conv epilogue: with a window-major pixel order the four positions of a pool window are four consecutive
// accumulator registers of one lane, so the whole run is register-local
struct PoolEpi {
    float *P, *Q, *R, *R2, *Fpre, *Fpost;
    int pre, pool, post; float a_pre, a_post;
    RngArg rng;
};

// component j of the Philox block held by lane SRC of this lane's quad, as a uniform (0,1] draw
template <int SRC>
__device__ __forceinline__ float quad_pick(const uint32_t r[4], int j) {
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)r[0], SRC * 0x55, 0xf, 0xf, true);
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)r[1], SRC * 0x55, 0xf, 0xf, true);
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)r[2], SRC * 0x55, 0xf, 0xf, true);
    const uint32_t t3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)r[3], SRC * 0x55, 0xf, 0xf, true);
    return u01(j == 0 ? t0 : (j == 1 ? t1 : (j == 2 ? t2 : t3)));
}

// ------------------------------------------------------------------ forward / dX gather-GEMM
// BWD = false: forward (Cin = C1 of I, Cout = C0);  BWD = true: dX (Cin = C0 of dO, Cout = C1)
template <int K, int S, int P, bool BWD, bool POOL = false>
__device__ __forceinline__ void conv_gemm_body(const float *__restrict__ X, float *__restrict__ Y, float *__restrict__ Y2,
                                               const float *__restrict__ F, const float *__restrict__ B,
                                               int N, int Hx, int Wx, int Cin, int Hy, int Wy, int Cout,
                                               int C0 /* filter inner dim */, int pairs_per_chunk, int bx, int by, int ksplit = 1,
                                               const PoolEpi *pe = nullptr, float *__restrict__ XC = nullptr) {
    __shared__ float Bl[LDS_FILTER_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const long npix = (long)N * Hy * Wy;
    // ksplit == 2 (small layers that would leave SIMDs empty): two waves share one 32x32 tile, each takes alternate
    // blocks of channel pairs, the halves meet in LDS - twice the waves, half the dependent loads / MFMAs per wave
    const int kh = (ksplit == 2) ? (w & 1) : 0;
    const long tile = (ksplit == 2) ? (long)bx * 2 + (w >> 1) : (long)bx * 4 + w;
    const long pix  = tile * 32 + l31;                       // this lane's A-row pixel
    const int  co0  = by * 32;                               // output-channel tile
    const bool pok  = pix < npix;
    int jy = 0, iy = 0, n = 0;
    if (pok) {
        if (POOL) {                                          // window-major: row m = 4 * window + position (Hy, Wy even)
            const long wdx = pix >> 2; const int pos = (int)(pix & 3), W2 = Wy >> 1, H2 = Hy >> 1;
            int j0, i0; split3(wdx, W2, H2, j0, i0, n);
            iy = 2 * i0 + (pos >> 1); jy = 2 * j0 + (pos & 1);
        } else split3(pix, Wy, Hy, jy, iy, n);
    }
    const float *nX = X + (long)n * Hx * Wx * Cin;
    if (POOL && XC && pok && by == 0 && kh == 0 && h == 0) {     // layer 0 keeps a COPY of the batch (forward.cu:39); same-size conv: shared pixel grid
        const long o = (((long)n * Hy + iy) * Wy + jy) * Cin;
        for (int ci = 0; ci < Cin; ci++) XC[o + ci] = X[o + ci];
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;

    // per-tap element offsets of this lane's pixel (-1 = outside the image): computed once, so the loads below are
    // unconditional (clamped address + select) and the compiler can keep a whole chunk of them in flight
    int off[K * K];
#pragma unroll
    for (int ky = 0; ky < K; ky++) {
        int gi; bool iok;
        if (!BWD) { gi = iy * S + ky - P; iok = gi >= 0 && gi < Hx; }
        else { const int ti = iy + P - ky; gi = ti / S; iok = ti >= 0 && (ti % S) == 0 && gi < Hx; }
#pragma unroll
        for (int kx = 0; kx < K; kx++) {
            int gj; bool jok;
            if (!BWD) { gj = jy * S + kx - P; jok = gj >= 0 && gj < Wx; }
            else { const int tj = jy + P - kx; gj = tj / S; jok = tj >= 0 && (tj % S) == 0 && gj < Wx; }
            off[ky * K + kx] = (pok && iok && jok) ? (gi * Wx + gj) * Cin : -1;
        }
    }
    const int npairs = (Cin + 1) >> 1;
    // Small filters (the whole [C1][K][K][C0] tensor fits the LDS stage) are copied verbatim with coalesced 16 B loads -
    // one round trip - and indexed in place; the per-element gather into MFMA-B order below costs a dependent global
    // load per LDS entry and dominated the LeNet-size layers.
    const int nF = (BWD ? Cout : Cin) * K * K * C0;
    const bool raw = nF <= LDS_FILTER_FLOATS;
    if (raw) {
        pairs_per_chunk = npairs;
        if ((((uintptr_t)F) & 15) == 0) {
            const int n4 = nF >> 2;
            for (int e = tid; e < n4; e += 256) reinterpret_cast<float4 *>(Bl)[e] = reinterpret_cast<const float4 *>(F)[e];
            for (int e = (n4 << 2) + tid; e < nF; e += 256) Bl[e] = F[e];
        } else for (int e = tid; e < nF; e += 256) Bl[e] = F[e];
        __syncthreads();
    }
    for (int cp0 = 0; cp0 < npairs; cp0 += pairs_per_chunk) {
        const int cpn = min(pairs_per_chunk, npairs - cp0);
        if (!raw) {
        // ---- stage the filter slice: Bl[((cp*K+ky)*K+kx)*2+hh][col] ----
        __syncthreads();
        const int nent = cpn * K * K * 2 * 32;
        for (int e = tid; e < nent; e += 256) {
            const int col = e & 31; int t = e >> 5;
            const int hh = t & 1; t >>= 1;
            const int kx = t % K; t /= K; const int ky = t % K; const int cp = t / K;
            const int ci = 2 * (cp0 + cp) + hh, co = co0 + col;
            float v = 0.f;
            if (ci < Cin && co < Cout) {
                if (!BWD) v = F[((long)(ci * K + ky) * K + kx) * C0 + co];                          // F[c1=ci][ky][kx][c0=co]
                else      v = F[((long)(co * K + (K - 1 - ky)) * K + (K - 1 - kx)) * C0 + ci];      // F[c1=co][flip][c0=ci]
            }
            Bl[e] = v;
        }
        __syncthreads();
        }
        for (int cpb = kh * 2; cpb < cpn; cpb += 2 * ksplit) {   // two channel pairs per trip: 2*K*K loads in flight
            float a[2][K * K];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int ci = 2 * (cp0 + cpb + u) + h;
                const bool cok = (cpb + u) < cpn && ci < Cin;
#pragma unroll
                for (int t = 0; t < K * K; t++) {
                    const bool ok = cok && off[t] >= 0;
                    const float v = nX[ok ? off[t] + ci : 0];
                    a[u][t] = ok ? v : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (cpb + u < cpn) {
#pragma unroll
                    for (int t = 0; t < K * K; t++) {
                        float b;
                        if (raw) {
                            const int ci = 2 * (cpb + u) + h, co = co0 + l31;
                            const bool ok = ci < Cin && co < Cout;
                            const int idx = !BWD ? (ci * K * K + t) * C0 + co : (co * K * K + (K * K - 1 - t)) * C0 + ci;
                            const float v = Bl[ok ? idx : 0];
                            b = ok ? v : 0.f;
                        } else b = Bl[(((cpb + u) * K * K + t) * 2 + h) * 32 + l31];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b, acc, 0, 0, 0);
                    }
                }
            }
        }
    }
    if (ksplit == 2) {                                      // the two k-halves of a tile meet in LDS (the filter stage is free now)
        __syncthreads();
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; r++) Bl[((w >> 1) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (kh == 1) return;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] += Bl[((w >> 1) * 16 + r) * 64 + lane];
    }
    // ---- epilogue: D[row = pixel][col = channel]; col = lane&31, row = (r&3)+8*(r>>2)+4*h
    const int co = co0 + l31;
    if (POOL) {
        // rows 4q+{0..3} (+4h) of this lane are the four positions of pool window tile*8 + 2q + h
        uint64_t rbase = 0, rseed = 0;
        const bool draw = pe->pre == T4K_L_DROPOUT;
        if (draw) rng_begin(pe->rng, rbase, rseed);
        if (co < Cout) {
            const float bias = B ? B[co] : 0.f;
            const int W2 = Wy >> 1, H2 = Hy >> 1;
            const long nwin = npix >> 2;
            // window coordinates: one split for the lane's first window, the other three advance by two windows each
            int j0, i0, nn0; split3(tile * 8 + h, W2, H2, j0, i0, nn0);
            long nn = nn0;
            // Cout % 4 == 0: the four lanes of a quad (channels 4g..4g+3) sit in ONE Philox counter block per pixel, so lane j
            // generates only the block of window position j and the quad exchanges components (4 blocks per lane, not 16)
            const bool quad = draw && (Cout & 3) == 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const long wdx = tile * 8 + 2 * q + h;
                if (wdx < nwin) {
                    const long a0 = (((nn * Hy + 2 * i0) * Wy) + 2 * j0) * Cout + co;    // window position 0; +Cout, +Wy*Cout, +both
                    float uq[4] = { 0.f, 0.f, 0.f, 0.f };
                    if (quad) {
                        const int j = lane & 3;
                        const long aj = a0 + (long)((j >> 1) * Wy + (j & 1)) * Cout;
                        uint32_t r4[4];
                        philox4x32_10(rbase + (uint64_t)(aj >> 2), rseed, r4);
                        uq[0] = quad_pick<0>(r4, j); uq[1] = quad_pick<1>(r4, j); uq[2] = quad_pick<2>(r4, j); uq[3] = quad_pick<3>(r4, j);
                    }
                    float pv = 0.f; bool first = true;
#pragma unroll
                    for (int pos = 0; pos < 4; pos++) {
                        const long a = a0 + (long)((pos >> 1) * Wy + (pos & 1)) * Cout;
                        float e = acc[4 * q + pos] + bias;
                        Y[a] = e;
                        if (pe->pre) {
                            float o, f;
                            act_rt_lean(pe->pre, e, quad ? uq[pos] : (draw ? philox_u01_at(rbase, rseed, a) : 0.f), pe->a_pre, o, f);
                            pe->Fpre[a] = f; pe->P[a] = o; e = o;
                        }
                        if (pe->pool == T4K_L_MAXPOOL)      pv = first ? e : fmaxf(e, pv);
                        else if (pe->pool == T4K_L_MINPOOL) pv = first ? e : fminf(e, pv);
                        else                                pv += e;
                        first = false;
                    }
                    if (pe->pool == T4K_L_AVGPOOL) pv /= 4.0f;
                    const long z = wdx * Cout + co;
                    pe->Q[z] = pv;
                    if (pe->post) { float o, f; act_rt_lean(pe->post, pv, 0.f, pe->a_post, o, f); pe->Fpost[z] = f; pe->R[z] = o; pv = o; }
                    if (pe->R2) pe->R2[z] = pv;
                }
                j0 += 2;                                                          // next window of this lane: wdx + 2
                while (j0 >= W2) { j0 -= W2; if (++i0 >= H2) { i0 = 0; nn++; } }
            }
        }
        if (draw && pe->rng.state) rng_advance_n(pe->rng.state, rbase, (uint64_t)((npix * Cout + 3) >> 2), gridDim.x * gridDim.y);
        return;
    }
    if (co < Cout) {
        const float bias = (!BWD && B) ? B[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const long p2 = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (p2 < npix) { const float v = acc[r] + bias; Y[p2 * Cout + co] = v; if (Y2) Y2[p2 * Cout + co] = v; }
        }
    }
}
// forward convolution with the element-wise run behind it (t4k_conv2d_block_fwd)
template <int K, int S, int P>
__global__ void __launch_bounds__(256) k_conv_gemm_pool(const float *__restrict__ X, float *__restrict__ Y, const float *__restrict__ F, const float *__restrict__ B,
                                                        int N, int Hx, int Wx, int Cin, int Hy, int Wy, int Cout, int C0, int pairs_per_chunk, int ksplit, PoolEpi pe, float *__restrict__ XC) {
    conv_gemm_body<K, S, P, false, true>(X, Y, nullptr, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0, pairs_per_chunk, blockIdx.x, blockIdx.y, ksplit, &pe, XC);
}

template <int K, int S, int P, bool BWD>
__global__ void __launch_bounds__(256) k_conv_gemm(const float *__restrict__ X, float *__restrict__ Y, float *__restrict__ Y2,
                                                   const float *__restrict__ F, const float *__restrict__ B,
                                                   int N, int Hx, int Wx, int Cin, int Hy, int Wy, int Cout, int C0, int pairs_per_chunk, int ksplit) {
    conv_gemm_body<K, S, P, BWD>(X, Y, Y2, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0, pairs_per_chunk, blockIdx.x, blockIdx.y, ksplit);
}

// ------------------------------------------------------------------ dF | dB
// grid = (slices, m_tiles, c0_tiles); each wave accumulates D[tap][c0] over its output rows
template <int K, int S, int P>
__device__ __forceinline__ void conv_df_body(const float *__restrict__ I, const float *__restrict__ DO,
                                             float *__restrict__ part,
                                             int N, int H1, int W1, int C1, int H0, int W0, int C0,
                                             int rows_per_wave, int bx, int by, int bz) {
    __shared__ float red[4][32][33];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int ntaps = C1 * K * K;                           // row `ntaps` is the bias row (all ones)
    const int tap = by * 32 + l31;
    const int co  = bz * 32 + l31;
    int c1 = 0, ky = 0, kx = 0;
    const bool is_tap = tap < ntaps, is_bias = tap == ntaps;
    if (is_tap) { kx = tap % K; ky = (tap / K) % K; c1 = tap / (K * K); }
    const bool cok = co < C0;
    const int rows = N * H0;
    const int row_beg = (bx * 4 + w) * rows_per_wave;
    const int row_end = min(rows, row_beg + rows_per_wave);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;

    for (int row = row_beg; row < row_end; row++) {
        const int n = row / H0, i0 = row - n * H0;          // wave-uniform
        const int gi = i0 * S + ky - P;
        const bool iok = is_tap && gi >= 0 && gi < H1;
        const float *rI = iok ? I + (((long)n * H1 + gi) * W1) * C1 + c1 : I;
        const float *rO = DO + ((long)row * W0) * C0 + (cok ? co : 0);
        const int nit = (W0 + 1) / 2;                       // wave-uniform trip count; lane half h takes pixel 2*it + h
        for (int itb = 0; itb < nit; itb += 7) {            // 7 pixel pairs per trip (W0 = 14, 28: no tail): 14 loads in flight
            float av[7], bv[7]; bool aok[7], jv[7];
#pragma unroll
            for (int u = 0; u < 7; u++) {
                const int j0 = 2 * (itb + u) + h;
                jv[u] = (itb + u) < nit && j0 < W0;
                const int gj = j0 * S + kx - P;
                aok[u] = jv[u] && iok && gj >= 0 && gj < W1;
                av[u] = rI[aok[u] ? gj * C1 : 0];           // unconditional (clamped) loads
                bv[u] = rO[jv[u] ? j0 * C0 : 0];
            }
#pragma unroll
            for (int u = 0; u < 7; u++) {
                if (itb + u < nit) {
                    const float a = (jv[u] && is_bias) ? 1.f : (aok[u] ? av[u] : 0.f);
                    const float b = (jv[u] && cok) ? bv[u] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
        }
    }
    // wave -> workgroup reduction through LDS, then one slab entry per (slice, tap, c0)
#pragma unroll
    for (int r = 0; r < 16; r++) red[w][(r & 3) + 8 * (r >> 2) + 4 * h][l31] = acc[r];
    __syncthreads();
    const int nrow1 = ntaps + 1;
    for (int e = tid; e < 1024; e += 256) {
        const int tr = e >> 5, tc = e & 31;
        const int gt = by * 32 + tr, gc = bz * 32 + tc;
        if (gt < nrow1 && gc < C0)
            part[((long)bx * nrow1 + gt) * C0 + gc] = (red[0][tr][tc] + red[1][tr][tc]) + (red[2][tr][tc] + red[3][tr][tc]);
    }
}
template <int K, int S, int P>
__global__ void __launch_bounds__(256) k_conv_df_mfma(const float *__restrict__ I, const float *__restrict__ DO, float *__restrict__ part,
                                                      int N, int H1, int W1, int C1, int H0, int W0, int C0, int rows_per_wave) {
    conv_df_body<K, S, P>(I, DO, part, N, H1, W1, C1, H0, W0, C0, rows_per_wave, blockIdx.x, blockIdx.y, blockIdx.z);
}
// fold the slabs: DF[i] += sum_slice part[slice][i], DB likewise.  One wave per output: lane l adds slices l, l+64, ...
// (all loads of a lane are independent), then a fixed xor-tree across the wave => deterministic, and the
// ~1000 slices of a LeNet-size layer are summed in two load rounds instead of a 200-deep dependent chain.
__device__ __forceinline__ void conv_df_fold_body(const float *__restrict__ part, float *DF, float *DB, int nslice, int ndf, int ntot, int bx) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = bx * 4 + w;
    if (i >= ntot) return;
    float s = 0.f;
#pragma unroll 4
    for (int k = lane; k < nslice; k += 64) s += part[(long)k * ntot + i];
    s = wave_sum_all(s);
    if (lane == 0) { if (i < ndf) DF[i] += s; else DB[i - ndf] += s; }
}
__global__ void __launch_bounds__(256) k_conv_df_fold(const float *__restrict__ part, float *DF, float *DB,
                                                      int nslice, int ndf, int ntot) {
    conv_df_fold_body(part, DF, DB, nslice, ndf, ntot, blockIdx.x);
}
// The dF fold and the layer's dX are independent once the dF partials exist, so they share a launch: the first nfold
// workgroups fold, the rest run the dX implicit GEMM (hx x hy grid, linearised).  dX may now overwrite the layer input
// (DX2 = I, the reference's `in = dx`): the dF kernel that read I finished with the previous launch.
template <int K, int S, int P>
__global__ void __launch_bounds__(256) k_conv_dx_and_fold(const float *__restrict__ part, float *DF, float *DB, int nslice, int ndf, int ntot, int nfold,
                                                          const float *__restrict__ DO, float *__restrict__ DX, float *__restrict__ DX2, const float *__restrict__ F,
                                                          int N, int H1, int W1, int C1, int H0, int W0, int C0, int hx, int ppc, int ksplit) {
    const int b = blockIdx.x;
    if (b < nfold) conv_df_fold_body(part, DF, DB, nslice, ndf, ntot, b);
    else { const int b2 = b - nfold; conv_gemm_body<K, S, P, true>(DO, DX, DX2, F, nullptr, N, H0, W0, C0, H1, W1, C1, C0, ppc, b2 % hx, b2 / hx, ksplit); }
}

// ------------------------------------------------------------------ generic column sums (dlinear_db)
__global__ void __launch_bounds__(BLK) k_fold_add(const float *__restrict__ part, float *OUT, int n, int nchunk) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < nchunk; k++) s += part[(long)k * n + i];
    OUT[i] += s;
}
__global__ void __launch_bounds__(BLK) k_colsum_part(const float *__restrict__ X, float *__restrict__ part,
                                                     long rows, int E, int rows_per_chunk, float *direct) {
    __shared__ float sm[4][64];
    const int ex = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int e = blockIdx.y * 64 + ex;
    const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float acc = 0.f;
    if (e < E) {
#pragma unroll 4
        for (long r = r0 + ry; r < r1; r += 4) acc += X[r * E + e];
    }
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
        int c, j0, i0, n; long t; split2(z, C, c, t); split3(t, W0, H0, j0, i0, n);
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
        int c, j0, i0, n; long t; split2(z, C, c, t); split3(t, W0, H0, j0, i0, n);
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

// ------------------------------------------------------------------ forward / dX for few channels (Cin, Cout <= 32)
// LeNet-class layers (1->10, 10->20 channels) are HBM/latency bound; padding 10 channels to the 32-wide MFMA tile and
// gathering one float per MFMA wastes the matrix unit.  A thread owns one output pixel x G output channels (G = 4 or 12
// accumulators; the channel group is uniform per workgroup, so filter reads are 16 B LDS broadcasts).  With < 1 wave per
// SIMD there is nothing to hide a load behind, so per image row of taps the thread first issues ALL its input loads
// (K taps x CH channels, unconditional: clamped address + select) and only then the FMAs: a 3x3x10 layer makes 6 memory
// round trips per pixel instead of 45.  Outputs leave through an LDS transpose so every store instruction is contiguous.
// The filter is staged once per workgroup as Wl[tap][ci][co] (taps flipped for dX, nmath.tcu:304-324).
template <int K, int S, int P, bool BWD, int G, int CH, int VW>
__global__ void __launch_bounds__(256) k_conv_few(const float *__restrict__ X, float *__restrict__ Y, float *__restrict__ Y2, float *__restrict__ XC,
                                                  const float *__restrict__ F, const float *__restrict__ B,
                                                  int N, int Hx, int Wx, int Cin, int Hy, int Wy, int Cout, int C0f, int NG) {
    __shared__ __attribute__((aligned(16))) float Wl[LDS_FILTER_FLOATS];
    __shared__ float Os[256 * G];
    constexpr int KK = K * K;
    const int COPT = NG * G;
    {
        const int nF = (BWD ? Cout : Cin) * KK * C0f;
        for (int e = threadIdx.x; e < KK * Cin * COPT; e += 256) Wl[e] = 0.f;
        __syncthreads();
        for (int e = threadIdx.x; e < nF; e += 256) {
            const int c0 = e % C0f; const int r = e / C0f; const int t = r % KK; const int c1 = r / KK;   // F[c1][t][c0]
            if (!BWD) Wl[(t * Cin + c1) * COPT + c0] = F[e];                  // ci = c1, co = c0
            else      Wl[((KK - 1 - t) * Cin + c0) * COPT + c1] = F[e];       // ci = c0, co = c1, taps flipped
        }
        __syncthreads();
    }
    const int g = blockIdx.y, co0 = g * G;
    const int gv = min(G, Cout - co0);                           // valid channels of this group
    const long npix = (long)N * Hy * Wy;
    for (long pix0 = (long)blockIdx.x * 256; pix0 < npix; pix0 += (long)gridDim.x * 256) {
        const long pix = pix0 + threadIdx.x;
        const bool live = pix < npix;
        const long pc = live ? pix : 0;
        int x, y, n; split3(pc, Wy, Hy, x, y, n);
        float acc[G];
#pragma unroll
        for (int u = 0; u < G; u++) acc[u] = 0.f;
        const float *nX = X + (long)n * Hx * Wx * Cin;
#pragma unroll
        for (int ky = 0; ky < K; ky++) {
            int gi; bool iok;
            if (!BWD) { gi = y * S + ky - P; iok = gi >= 0 && gi < Hx; }
            else { const int ti = y + P - ky; gi = ti / S; iok = ti >= 0 && (ti % S) == 0 && gi < Hx; }
            const float *d[K]; bool ok[K];
#pragma unroll
            for (int kx = 0; kx < K; kx++) {
                int gj; bool jok;
                if (!BWD) { gj = x * S + kx - P; jok = gj >= 0 && gj < Wx; }
                else { const int tj = x + P - kx; gj = tj / S; jok = tj >= 0 && (tj % S) == 0 && gj < Wx; }
                ok[kx] = live && iok && jok;
                d[kx] = nX + (ok[kx] ? ((long)gi * Wx + gj) * Cin : 0);
            }
            for (int ci0 = 0; ci0 < Cin; ci0 += CH) {
                float v[K][CH];
#pragma unroll
                for (int kx = 0; kx < K; kx++)
#pragma unroll
                    for (int q = 0; q < CH; q += VW) {
                        const int ci = (ci0 + q < Cin) ? ci0 + q : 0;          // clamped: the load is unconditional
                        if (VW == 4)      { const float4 t4 = *reinterpret_cast<const float4 *>(d[kx] + ci); v[kx][q] = t4.x; v[kx][(q + 1) % CH] = t4.y; v[kx][(q + 2) % CH] = t4.z; v[kx][(q + 3) % CH] = t4.w; }
                        else if (VW == 2) { const float2 t2 = *reinterpret_cast<const float2 *>(d[kx] + ci); v[kx][q] = t2.x; v[kx][(q + 1) % CH] = t2.y; }
                        else              v[kx][q] = d[kx][ci];
                    }
#pragma unroll
                for (int kx = 0; kx < K; kx++)
#pragma unroll
                    for (int q = 0; q < CH; q++) {
                        const float xv = (ok[kx] && ci0 + q < Cin) ? v[kx][q] : 0.f;
                        const float *wq = Wl + (((ky * K + kx) * Cin) + min(ci0 + q, Cin - 1)) * COPT + co0;
#pragma unroll
                        for (int u4 = 0; u4 < G; u4 += 4) {
                            const float4 f4 = *reinterpret_cast<const float4 *>(wq + u4);
                            acc[u4] = fmaf(xv, f4.x, acc[u4]); acc[u4 + 1] = fmaf(xv, f4.y, acc[u4 + 1]);
                            acc[u4 + 2] = fmaf(xv, f4.z, acc[u4 + 2]); acc[u4 + 3] = fmaf(xv, f4.w, acc[u4 + 3]);
                        }
                    }
            }
        }
        if (XC && g == 0 && live)                               // layer 0 keeps a COPY of the batch (forward.cu:39): same-size conv, pixel index is shared
            for (int ci = 0; ci < Cin; ci++) XC[pix * Cin + ci] = X[pix * Cin + ci];
        // transpose through LDS: the workgroup's 256 x gv results leave as contiguous runs
        __syncthreads();
#pragma unroll
        for (int u = 0; u < G; u++) Os[threadIdx.x * G + u] = acc[u] + ((!BWD && B && co0 + u < Cout) ? B[co0 + u] : 0.f);
        __syncthreads();
        const int nval = (int)min((long)256, npix - pix0) * gv;
        for (int e = threadIdx.x; e < nval; e += 256) {
            const int pp = e / gv, u = e - pp * gv;
            const float r = Os[pp * G + u];
            const long o = (pix0 + pp) * Cout + co0 + u;
            Y[o] = r; if (Y2) Y2[o] = r;
        }
    }
}
bool conv_few_ok(int K, int Cin, int Cout, int *G_out, int *NG_out) {
    // measured on MI355X: wins for image-input layers (1->10: 6.1 vs 8.4 us); at 10<->20 channels the thread-per-pixel
    // kernel is FMA/LDS bound with < 1 wave per SIMD and loses to the MFMA implicit GEMM (13.9 vs 10.3 us)
    if (Cin > 4 || Cout > 32 || (K != 3 && K != 5)) return false;
    const int G = Cout <= 4 ? 4 : 12;
    const int NG = (Cout + G - 1) / G;
    if (K * K * Cin * NG * G > LDS_FILTER_FLOATS) return false;
    *G_out = G; *NG_out = NG;
    return true;
}
template <bool BWD, int G, int CH, int VW>
void launch_conv_few3(int K, dim3 g, hipStream_t hs, const float *X, float *Y, float *Y2, float *XC, const float *F, const float *B,
                      int N, int Hx, int Wx, int Cin, int Hy, int Wy, int Cout, int C0f, int NG) {
    const dim3 b(256);
    if (K == 3) T4K_LAUNCH((k_conv_few<3, 1, 1, BWD, G, CH, VW>), g, b, 0, hs, X, Y, Y2, XC, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0f, NG);
    else        T4K_LAUNCH((k_conv_few<5, 1, 2, BWD, G, CH, VW>), g, b, 0, hs, X, Y, Y2, XC, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0f, NG);
}
template <bool BWD>
void launch_conv_few(int K, hipStream_t hs, const float *X, float *Y, float *Y2, float *XC, const float *F, const float *B,
                     int N, int Hx, int Wx, int Cin, int Hy, int Wy, int Cout, int C0f, int G, int NG) {
    const long npix = (long)N * Hy * Wy;
    long gx = (npix + 255) / 256; if (gx > 8192) gx = 8192;
    const dim3 g((unsigned)gx, (unsigned)NG);
    const bool v2 = (Cin & 1) == 0 && (((uintptr_t)X) & 7) == 0;
#define FEW(GG) do { if (Cin == 1)      launch_conv_few3<BWD, GG, 1, 1>(K, g, hs, X, Y, Y2, XC, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0f, NG); \
                     else if (Cin <= 4) { if (v2) launch_conv_few3<BWD, GG, 4, 2>(K, g, hs, X, Y, Y2, XC, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0f, NG); \
                                          else    launch_conv_few3<BWD, GG, 4, 1>(K, g, hs, X, Y, Y2, XC, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0f, NG); } \
                     else               { if (v2) launch_conv_few3<BWD, GG, 8, 2>(K, g, hs, X, Y, Y2, XC, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0f, NG); \
                                          else    launch_conv_few3<BWD, GG, 8, 1>(K, g, hs, X, Y, Y2, XC, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0f, NG); } } while (0)
    if (G == 4) FEW(4); else FEW(12);
#undef FEW
}

// ------------------------------------------------------------------ dX for very few input channels (C1 <= 4)
// The first layer of an image net has 1 (MNIST) or 3 (CIFAR) input channels: as an implicit GEMM its dX would use 1/32
// of the matrix unit's N dimension and gather one float per MFMA.  Here a thread owns one pixel of the input grid and its
// CO accumulators, reads the C0 contiguous gradients of each tap's output pixel (adjacent lanes = adjacent pixels, so a
// wave streams a contiguous span of dO) and takes the flipped filter (nmath.tcu:304-324) from LDS at a wave-uniform address.
template <int K, int S, int P, int CO>
__device__ __forceinline__ void conv_dx_few_body(const float *__restrict__ DO, float *__restrict__ DX, float *__restrict__ DX2,
                                                 const float *__restrict__ F, int N, int H0, int W0, int C0, int H1, int W1, int bx, int gx) {
    __shared__ __attribute__((aligned(16))) float Fl[LDS_FILTER_FLOATS];
    const int nF = CO * K * K * C0;
    for (int e = threadIdx.x; e < nF; e += 256) Fl[e] = F[e];
    __syncthreads();
    const long npix = (long)N * H1 * W1;
    for (long pix = (long)bx * 256 + threadIdx.x; pix < npix; pix += (long)gx * 256) {
        int x, y, n; split3(pix, W1, H1, x, y, n);
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; c++) acc[c] = 0.f;
        const float *nD = DO + (long)n * H0 * W0 * C0;
#pragma unroll
        for (int ky = 0; ky < K; ky++) {
            const int ti = y + P - ky, gi = ti / S;
            const bool iok = ti >= 0 && (ti % S) == 0 && gi < H0;
#pragma unroll
            for (int kx = 0; kx < K; kx++) {
                const int tj = x + P - kx, gj = tj / S;
                const bool ok = iok && tj >= 0 && (tj % S) == 0 && gj < W0;
                const float *d = nD + (ok ? ((long)gi * W0 + gj) * C0 : 0);
                const float *f = Fl + ((K - 1 - ky) * K + (K - 1 - kx)) * C0;     // F[c1][K-1-ky][K-1-kx][c0]
                const float msk = ok ? 1.f : 0.f;                 // loads are unconditional (clamped pixel), masked by a multiply-free select
                if ((C0 & 3) == 0) {                              // 16 B loads of dO and of the weights (LDS rows are 16 B aligned: C0 % 4 == 0)
#pragma unroll 4
                    for (int c0 = 0; c0 < C0; c0 += 4) {
                        const float4 v4 = *reinterpret_cast<const float4 *>(d + c0);
                        const float v0 = ok ? v4.x : 0.f, v1 = ok ? v4.y : 0.f, v2 = ok ? v4.z : 0.f, v3 = ok ? v4.w : 0.f;
#pragma unroll
                        for (int c = 0; c < CO; c++) {
                            const float4 w4 = *reinterpret_cast<const float4 *>(f + c * K * K * C0 + c0);
                            acc[c] = fmaf(v0, w4.x, acc[c]); acc[c] = fmaf(v1, w4.y, acc[c]); acc[c] = fmaf(v2, w4.z, acc[c]); acc[c] = fmaf(v3, w4.w, acc[c]);
                        }
                    }
                } else if ((C0 & 1) == 0) {                       // even channel count: 8 B loads (pixel rows are 8 B aligned)
#pragma unroll 5
                    for (int c0 = 0; c0 < C0; c0 += 2) {
                        const float2 v2 = *reinterpret_cast<const float2 *>(d + c0);
                        const float v0 = ok ? v2.x : 0.f, v1 = ok ? v2.y : 0.f;
#pragma unroll
                        for (int c = 0; c < CO; c++) { acc[c] = fmaf(v0, f[c * K * K * C0 + c0], acc[c]); acc[c] = fmaf(v1, f[c * K * K * C0 + c0 + 1], acc[c]); }
                    }
                } else {
                    for (int c0 = 0; c0 < C0; c0++) {
                        const float v0 = d[c0], v = ok ? v0 : 0.f;
#pragma unroll
                        for (int c = 0; c < CO; c++) acc[c] = fmaf(v, f[c * K * K * C0 + c0], acc[c]);
                    }
                }
                (void)msk;
            }
        }
#pragma unroll
        for (int c = 0; c < CO; c++) { DX[pix * CO + c] = acc[c]; if (DX2) DX2[pix * CO + c] = acc[c]; }
    }
}
// optional fold of the same layer's dF partials in the first `nfold` workgroups (see k_conv_dx_and_fold)
struct FoldArgs { const float *part; float *DF, *DB; int nslice, ndf, ntot, nfold; };
template <int K, int S, int P, int CO>
__global__ void __launch_bounds__(256) k_conv_dx_few(const float *__restrict__ DO, float *__restrict__ DX, float *__restrict__ DX2,
                                                     const float *__restrict__ F, int N, int H0, int W0, int C0, int H1, int W1, FoldArgs fa) {
    const int b = blockIdx.x;
    if (b < fa.nfold) conv_df_fold_body(fa.part, fa.DF, fa.DB, fa.nslice, fa.ndf, fa.ntot, b);
    else conv_dx_few_body<K, S, P, CO>(DO, DX, DX2, F, N, H0, W0, C0, H1, W1, b - fa.nfold, (int)gridDim.x - fa.nfold);
}
// Same layer shape (C1 = CO <= 4 input channels) but MANY output channels (C0 = 32 / 64 / 128, e.g. the 3 -> 64 first layer of a
// CIFAR net): with a thread per pixel every lane walks its own 4*C0-byte run of dO, a wave touches 64 different runs per load and
// the L1 thrashes (90 us for N=256, 32x32, 3->64 = 6 % of the vector peak).  Here LPP = C0/4 lanes share a pixel, lane q owns
// channels 4q..4q+3: a load instruction reads whole pixels (fully coalesced 16 B per lane), the lane's 4 x 9 x CO weights live in
// registers for the whole grid-stride loop, and the LPP partial sums meet through an xor tree.  3x3, stride 1 only.
template <int CO, int LPP>
__device__ __forceinline__ void conv_dx_wide_body(const float *__restrict__ DO, float *__restrict__ DX, float *__restrict__ DX2,
                                                  const float *__restrict__ F, int N, int H0, int W0, int H1, int W1, int bx, int gx) {
    constexpr int K = 3, KK = 9, C0 = LPP * 4, PPW = 64 / LPP, PPB = 4 * PPW;    // pixels per wave / per workgroup
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, q = lane % LPP, sub = lane / LPP;
    float4 wt[CO][KK];                                                           // F[c1][K-1-ky][K-1-kx][4q..4q+3]
#pragma unroll
    for (int c = 0; c < CO; c++)
#pragma unroll
        for (int t = 0; t < KK; t++) wt[c][t] = *reinterpret_cast<const float4 *>(F + ((long)c * KK + (KK - 1 - t)) * C0 + 4 * q);
    const long npix = (long)N * H1 * W1;
    for (long p0 = (long)bx * PPB; p0 < npix; p0 += (long)gx * PPB) {
        const long pix = p0 + w * PPW + sub;
        const bool live = pix < npix;
        int x, y, n; split3(live ? pix : 0, W1, H1, x, y, n);
        const float *nD = DO + (long)n * H0 * W0 * C0 + 4 * q;
        float4 v[KK]; bool ok[KK];
#pragma unroll
        for (int ky = 0; ky < K; ky++)
#pragma unroll
            for (int kx = 0; kx < K; kx++) {
                const int gi = y + 1 - ky, gj = x + 1 - kx;                       // P = 1, S = 1
                ok[ky * K + kx] = live && gi >= 0 && gi < H0 && gj >= 0 && gj < W0;
                v[ky * K + kx] = *reinterpret_cast<const float4 *>(nD + (ok[ky * K + kx] ? ((long)gi * W0 + gj) * C0 : 0));   // unconditional
            }
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; c++) acc[c] = 0.f;
#pragma unroll
        for (int t = 0; t < KK; t++) {
            const float v0 = ok[t] ? v[t].x : 0.f, v1 = ok[t] ? v[t].y : 0.f, v2 = ok[t] ? v[t].z : 0.f, v3 = ok[t] ? v[t].w : 0.f;
#pragma unroll
            for (int c = 0; c < CO; c++) {
                acc[c] = fmaf(v0, wt[c][t].x, acc[c]); acc[c] = fmaf(v1, wt[c][t].y, acc[c]);
                acc[c] = fmaf(v2, wt[c][t].z, acc[c]); acc[c] = fmaf(v3, wt[c][t].w, acc[c]);
            }
        }
#pragma unroll
        for (int off = LPP / 2; off > 0; off >>= 1)
#pragma unroll
            for (int c = 0; c < CO; c++) acc[c] += __shfl_xor(acc[c], off, 64);
        if (live && q == 0) {
#pragma unroll
            for (int c = 0; c < CO; c++) { DX[pix * CO + c] = acc[c]; if (DX2) DX2[pix * CO + c] = acc[c]; }
        }
    }
}
template <int CO, int LPP>
__global__ void __launch_bounds__(256) k_conv_dx_wide(const float *__restrict__ DO, float *__restrict__ DX, float *__restrict__ DX2,
                                                      const float *__restrict__ F, int N, int H0, int W0, int H1, int W1, FoldArgs fa) {
    const int b = blockIdx.x;
    if (b < fa.nfold) conv_df_fold_body(fa.part, fa.DF, fa.DB, fa.nslice, fa.ndf, fa.ntot, b);
    else conv_dx_wide_body<CO, LPP>(DO, DX, DX2, F, N, H0, W0, H1, W1, b - fa.nfold, (int)gridDim.x - fa.nfold);
}
template <int CO>
void launch_conv_dx_few(int K, int S, int P, hipStream_t hs, const float *DO, float *DX, float *DX2, const float *F,
                        int N, int H0, int W0, int C0, int H1, int W1, FoldArgs fa) {
    const long npix = (long)N * H1 * W1;
    static const int wide = T4K_LAB_ENV("T4K_DX_WIDE", 1);
    if (wide && K == 3 && S == 1 && P == 1 && (C0 == 32 || C0 == 64 || C0 == 128) && aligned16(DO) && aligned16(F)) {
        const int ppb = 4 * (64 / (C0 / 4));
        static const int wpc = T4K_LAB_ENV("T4K_DX_WIDE_WPC", 8);
        long gw = (npix + ppb - 1) / ppb; if (gw > (long)st().cu_count * wpc) gw = (long)st().cu_count * wpc;   // the weights are loaded once per workgroup
        const dim3 gg((unsigned)gw + fa.nfold), bb(256);
        if (C0 == 32)      T4K_LAUNCH((k_conv_dx_wide<CO, 8>),  gg, bb, 0, hs, DO, DX, DX2, F, N, H0, W0, H1, W1, fa);
        else if (C0 == 64) T4K_LAUNCH((k_conv_dx_wide<CO, 16>), gg, bb, 0, hs, DO, DX, DX2, F, N, H0, W0, H1, W1, fa);
        else               T4K_LAUNCH((k_conv_dx_wide<CO, 32>), gg, bb, 0, hs, DO, DX, DX2, F, N, H0, W0, H1, W1, fa);
        return;
    }
    long gx = (npix + 255) / 256; if (gx > 8192) gx = 8192;
    const dim3 g((unsigned)gx + fa.nfold), b(256);
    switch ((K << 8) | (S << 4) | P) {
    case 0x110: T4K_LAUNCH((k_conv_dx_few<1, 1, 0, CO>), g, b, 0, hs, DO, DX, DX2, F, N, H0, W0, C0, H1, W1, fa); break;
    case 0x311: T4K_LAUNCH((k_conv_dx_few<3, 1, 1, CO>), g, b, 0, hs, DO, DX, DX2, F, N, H0, W0, C0, H1, W1, fa); break;
    case 0x421: T4K_LAUNCH((k_conv_dx_few<4, 2, 1, CO>), g, b, 0, hs, DO, DX, DX2, F, N, H0, W0, C0, H1, W1, fa); break;
    case 0x512: T4K_LAUNCH((k_conv_dx_few<5, 1, 2, CO>), g, b, 0, hs, DO, DX, DX2, F, N, H0, W0, C0, H1, W1, fa); break;
    }
}

bool conv_block_on() { static const int v = T4K_LAB_ENV("T4K_CONV_BLOCK", 1); return v != 0; }
bool conv_big_on() { static const int v = T4K_LAB_ENV("T4K_CONV_BIG", 1); return v != 0; }
bool conv_few_on() { static const int v = T4K_LAB_ENV("T4K_CONV_FEW", 1); return v != 0; }
// two waves per tile when the layer is small enough to leave SIMDs empty and has enough k-work to split
int conv_gemm_ksplit(long npix, int Cout, int Cin, int K) {
    static const int on = T4K_LAB_ENV("T4K_CONV_KSPLIT", 1);
    const long waves = ((npix + 31) / 32) * ((Cout + 31) / 32);
    return (on && waves < 1536 && ((Cin + 1) / 2) * K * K >= 18) ? 2 : 1;
}
bool conv_supported(int K, int S, int P) {
    return (K == 1 && S == 1 && P == 0) || (K == 3 && S == 1 && P == 1) ||
           (K == 4 && S == 2 && P == 1) || (K == 5 && S == 1 && P == 2);
}

template <bool BWD>
void launch_conv_gemm(int K, int S, int P, dim3 g, hipStream_t hs, const float *X, float *Y, float *Y2, const float *F, const float *B,
                      int N, int Hx, int Wx, int Cin, int Hy, int Wy, int Cout, int C0) {
    const int ppc = LDS_FILTER_FLOATS / (K * K * 2 * 32);            // channel pairs per LDS filter slice
    const long npix = (long)N * Hy * Wy;
    const int ksplit = conv_gemm_ksplit(npix, Cout, Cin, K);
    g.x = (unsigned)((npix + (128 / ksplit) - 1) / (128 / ksplit));
    switch ((K << 8) | (S << 4) | P) {
    case 0x110: T4K_LAUNCH((k_conv_gemm<1, 1, 0, BWD>), g, dim3(256), 0, hs, X, Y, Y2, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0, ppc, ksplit); break;
    case 0x311: T4K_LAUNCH((k_conv_gemm<3, 1, 1, BWD>), g, dim3(256), 0, hs, X, Y, Y2, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0, ppc, ksplit); break;
    case 0x421: T4K_LAUNCH((k_conv_gemm<4, 2, 1, BWD>), g, dim3(256), 0, hs, X, Y, Y2, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0, ppc, ksplit); break;
    case 0x512: T4K_LAUNCH((k_conv_gemm<5, 1, 2, BWD>), g, dim3(256), 0, hs, X, Y, Y2, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0, ppc, ksplit); break;
    }
}

} // namespace

namespace t4k {
// OUT[e] += sum_rows X[row][e], deterministic (used by t4k_linear_bwd / t4k_dlinear_db)
int colsum_add(const float *X, float *OUT, long rows, int E, hipStream_t hs) {
    if (rows <= 0 || E <= 0) return T4K_OK;
    // ~256 rows per chunk (each of the 4 row groups then sums 64 rows), up to 2048 chunks: a 262144 x 64 matrix
    // (the dO of a CIFAR-size conv layer) spreads over 1024 workgroups instead of 64
    long want = (rows + 255) / 256; if (want > 2048) want = 2048; if (want < 1) want = 1;
    if (rows <= 1024) want = 1;                       // small: single chunk accumulates in place (one launch)
    const int rpc = (int)((rows + want - 1) / want);
    const int nchunk = (int)((rows + rpc - 1) / rpc);
    float *part = ws_for(hs) + (8 << 20);            // second 32 MiB half of the workspace
    if ((size_t)nchunk * E * sizeof(float) > st().ws_bytes / 2) return fail(T4K_ERR_NOMEM, "colsum workspace");
    if (nchunk == 1) {
        T4K_LAUNCH(k_colsum_part, dim3(1, (E + 63) / 64), dim3(BLK), 0, hs, X, part, rows, E, rpc, OUT);
        return T4K_OK;
    }
    T4K_LAUNCH(k_colsum_part, dim3(nchunk, (E + 63) / 64), dim3(BLK), 0, hs, X, part, rows, E, rpc, (float *)nullptr);
    T4K_LAUNCH(k_conv_df_fold, dim3((E + 3) / 4), dim3(256), 0, hs, part, OUT, OUT, nchunk, E, E);   // one wave per output, fixed xor tree
    return T4K_OK;
}
}

extern "C" {

int t4k_conv2d_fwd(const float *I, float *O, const float *F, const float *B,
                   int N, int H1, int W1, int C1, int H0, int W0, int C0,
                   int K, int S, int P, t4k_stream_t s) {
    return t4k_conv2d_fwd2(I, nullptr, O, F, B, N, H1, W1, C1, H0, W0, C0, K, S, P, s);
}
// the forward of one layer; bn_part (may be NULL): where the layer's kernel may leave the per-channel sums of O as chunk partials (*bn_chunks > 0 says it did)
static int conv2d_fwd_impl(const float *I, float *ICOPY, float *O, const float *F, const float *B,
                           int N, int H1, int W1, int C1, int H0, int W0, int C0,
                           int K, int S, int P, float *bn_part, size_t bn_part_floats, int *bn_chunks, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (bn_chunks) *bn_chunks = 0;
    if (!conv_supported(K, S, P))
        return fail(T4K_ERR_UNSUPPORTED, "nn#fconv kernel_size=%d stride=%d padding=%d not supported", K, S, P);
    if (!I || !O || !F || !B || N <= 0 || C0 <= 0 || C1 <= 0 || H0 <= 0 || W0 <= 0) return fail(T4K_ERR_ARG, "t4k_conv2d_fwd: bad argument");
    int fG, fNG;
    if (conv_few_on() && conv_few_ok(K, C1, C0, &fG, &fNG)) {
        launch_conv_few<false>(K, t4k::S(s), I, O, nullptr, (H0 == H1 && W0 == W1) ? ICOPY : nullptr, F, B, N, H1, W1, C1, H0, W0, C0, C0, fG, fNG);
        if (ICOPY && !(H0 == H1 && W0 == W1)) T4K_HIP(hipMemcpyAsync(ICOPY, I, sizeof(float) * (size_t)N * H1 * W1 * C1, hipMemcpyDeviceToDevice, t4k::S(s)));
        T4K_LAUNCH_CHECK();
        return T4K_OK;
    }
    // image in, a full MFMA tile or two of channels out (3 -> 64): filter in registers, the layer-0 copy from the same launch (conv_img.hip)
    if (K == 3 && S == 1 && P == 1 && H0 == H1 && W0 == W1 && ICOPY != O && conv_thin_fwd(I, ICOPY, O, F, B, N, H0, W0, C1, C0, t4k::S(s), bn_part, bn_part_floats, bn_chunks)) { T4K_LAUNCH_CHECK(); return T4K_OK; }
    if (ICOPY) T4K_HIP(hipMemcpyAsync(ICOPY, I, sizeof(float) * (size_t)N * H1 * W1 * C1, hipMemcpyDeviceToDevice, t4k::S(s)));
    if (conv_big_on() && conv_big_ok(C1, C0) && aligned16(I) && aligned16(F)) {       // many channels: LDS-staged GEMM tiling
        launch_conv_big<false>(K, S, P, t4k::S(s), I, O, nullptr, F, B, N, H1, W1, C1, H0, W0, C0, C0, bn_part, bn_part_floats, bn_chunks);
        T4K_LAUNCH_CHECK();
        return T4K_OK;
    }
    const long npix = (long)N * H0 * W0;
    dim3 g((unsigned)((npix + 127) / 128), (unsigned)((C0 + 31) / 32));
    launch_conv_gemm<false>(K, S, P, g, t4k::S(s), I, O, nullptr, F, B, N, H1, W1, C1, H0, W0, C0, C0);
    T4K_LAUNCH_CHECK();
    return T4K_OK;
}
int t4k_conv2d_fwd2(const float *I, float *ICOPY, float *O, const float *F, const float *B,
                    int N, int H1, int W1, int C1, int H0, int W0, int C0,
                    int K, int S, int P, t4k_stream_t s) {
    return conv2d_fwd_impl(I, ICOPY, O, F, B, N, H1, W1, C1, H0, W0, C0, K, S, P, nullptr, 0, nullptr, s);
}
// conv forward + the batch-norm forward behind it: same tensors, same arithmetic per element as t4k_conv2d_fwd2 + t4k_batchnorm_fwd; where the layer's kernel
// can carry them (k_convbig8, k_conv_thin_fwd) the per-channel sums leave its epilogue as chunk partials in the stream's workspace and the statistics pass over
// the conv output (a full read of it) is not launched.  T4K_CONV_BN_RIDER=0: always the two calls.
int t4k_conv2d_bn_fwd(const float *I, float *ICOPY, float *Y, const float *F, const float *Bc,
                      int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P,
                      float *O, float *XH, const float *W, const float *B, float *stat_dev, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!O || !XH || !W || !B || !stat_dev) return fail(T4K_ERR_ARG, "t4k_conv2d_bn_fwd: null batch-norm tensor");
    static const int on = T4K_LAB_ENV("T4K_CONV_BN_RIDER", 1);
    const bool rider = on && !(st().bn_sync && t4k_comm_world() > 0);                // synchronised statistics go through the all-reduce path of t4k_batchnorm_fwd
    int chunks = 0;
    int rc = conv2d_fwd_impl(I, ICOPY, Y, F, Bc, N, H1, W1, C1, H0, W0, C0, K, S, P, rider ? ws_for(s) : nullptr, st().ws_bytes / 8, &chunks, s);
    if (rc != T4K_OK) return rc;
    if (chunks > 0) return bn_fwd_from_parts(Y, O, XH, W, B, stat_dev, (long)N * H0 * W0, C0, ws_for(s), chunks, t4k::S(s));
    return t4k_batchnorm_fwd(Y, O, XH, W, B, stat_dev, N, H0 * W0, C0, s);
}

// conv + batch-norm + the element-wise run behind them (the CIFAR-style block conv -> batchnorm -> relu -> maxpool -> dropout): the conv (its epilogue carrying the
// per-channel sums where it can), the finalise of the statistics, then ONE pass that reads the conv output once and writes x-hat, the batch-norm output and
// every tensor of the run (t4k_bn_poolblock_fwd) - the batch-norm output is not read back from memory.  Same tensors as the three calls.
int t4k_conv2d_bn_block_fwd(const float *I, float *ICOPY, float *Y, const float *F, const float *Bc,
                            int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P,
                            float *O, float *XH, const float *W, const float *B, float *stat_dev,
                            const t4k_poolblock *blk, int Hq, int Wq, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!O || !XH || !W || !B || !stat_dev || !blk) return fail(T4K_ERR_ARG, "t4k_conv2d_bn_block_fwd: null tensor");
    static const int on = T4K_LAB_ENV("T4K_CONV_BN_RIDER", 1);
    const bool rider = on && !(st().bn_sync && t4k_comm_world() > 0);
    int chunks = 0;
    int rc = conv2d_fwd_impl(I, ICOPY, Y, F, Bc, N, H1, W1, C1, H0, W0, C0, K, S, P, rider ? ws_for(s) : nullptr, st().ws_bytes / 8, &chunks, s);
    if (rc != T4K_OK) return rc;
    rc = bn_stats_for(Y, stat_dev, N, H0 * W0, C0, ws_for(s), chunks, s); if (rc != T4K_OK) return rc;
    return t4k_bn_poolblock_fwd(Y, O, XH, W, B, stat_dev, blk, N, H0, W0, Hq, Wq, C0, s);
}

// conv forward + the element-wise run behind it (dropout/activation -> 2x2 pool -> activation -> flatten copy) in ONE launch
// when the layer takes the gather-MFMA kernel; otherwise the two launches t4k_conv2d_fwd2 + t4k_poolblock_fwd.
int t4k_conv2d_block_fwd(const float *I, float *ICOPY, float *O, const float *F, const float *B, const t4k_poolblock *blk,
                         int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!blk) return fail(T4K_ERR_ARG, "t4k_conv2d_block_fwd: null block");
    const bool fusable = blk->pool_layer && blk->KS == 2 && (H0 % 2) == 0 && (W0 % 2) == 0 && S == 1 && (K == 3 || K == 5) &&
                         blk->pool_out && (!blk->pre_layer || (blk->pre_mask && blk->pre_out)) && (!blk->post_layer || (blk->post_mask && blk->post_out)) &&
                         blk->post_layer != T4K_L_DROPOUT &&           // a dropout BEHIND the pool draws in the element-wise run kernel (fused.hip), not in the conv epilogue
                         !(conv_big_on() && conv_big_ok(C1, C0)) &&
                         conv_supported(K, S, P) && I && O && F && B && conv_block_on();
    if (!fusable) {
        int rc = t4k_conv2d_fwd2(I, ICOPY, O, F, B, N, H1, W1, C1, H0, W0, C0, K, S, P, s); if (rc) return rc;
        return t4k_poolblock_fwd(O, blk, N, H0, W0, H0 / blk->KS, W0 / blk->KS, C0, s);
    }
    hipStream_t hs = t4k::S(s);
    // image-input layer (1 or 3 channels in, <= 16 out): the thread-per-pool-window vector kernel of conv_img.hip
    if (K == 3 && P == 1 && H1 == H0 && W1 == W0 && conv_img_block_fwd(I, ICOPY, O, F, B, blk, N, H0, W0, C1, C0, hs)) { T4K_LAUNCH_CHECK(); return T4K_OK; }
    // layer-0 copy: written by the conv launch itself when input and output share the pixel grid and the channels are few
    float *xc = (ICOPY && H1 == H0 && W1 == W0 && C1 <= 4) ? ICOPY : nullptr;
    if (ICOPY && !xc) T4K_HIP(hipMemcpyAsync(ICOPY, I, sizeof(float) * (size_t)N * H1 * W1 * C1, hipMemcpyDeviceToDevice, hs));
    PoolEpi pe;
    pe.P = blk->pre_out; pe.Q = blk->pool_out; pe.R = blk->post_out; pe.R2 = blk->copy_out; pe.Fpre = blk->pre_mask; pe.Fpost = blk->post_mask;
    pe.pre = blk->pre_layer; pe.pool = blk->pool_layer; pe.post = blk->post_layer; pe.a_pre = blk->pre_alpha; pe.a_post = blk->post_alpha;
    const long npix = (long)N * H0 * W0;
    pe.rng = RngArg{0, 0, nullptr};
    if (pe.pre == T4K_L_DROPOUT) pe.rng = rng_draw(hs, (uint64_t)((npix * C0 + 3) >> 2), true);
    const int ksplit = conv_gemm_ksplit(npix, C0, C1, K);
    const int ppc = LDS_FILTER_FLOATS / (K * K * 2 * 32);
    const dim3 g((unsigned)((npix + (128 / ksplit) - 1) / (128 / ksplit)), (unsigned)((C0 + 31) / 32));
    if (K == 3) T4K_LAUNCH((k_conv_gemm_pool<3, 1, 1>), g, dim3(256), 0, hs, I, O, F, B, N, H1, W1, C1, H0, W0, C0, C0, ppc, ksplit, pe, xc);
    else        T4K_LAUNCH((k_conv_gemm_pool<5, 1, 2>), g, dim3(256), 0, hs, I, O, F, B, N, H1, W1, C1, H0, W0, C0, C0, ppc, ksplit, pe, xc);
    T4K_LAUNCH_CHECK();
    return T4K_OK;
}

int t4k_conv2d_bwd(const float *I, const float *DO, float *DX, const float *F, float *DF, float *DB,
                   int N, int H1, int W1, int C1, int H0, int W0, int C0,
                   int K, int S, int P, int train, t4k_stream_t s) {
    return t4k_conv2d_bwd2(I, DO, DX, nullptr, F, DF, DB, N, H1, W1, C1, H0, W0, C0, K, S, P, train, s);
}

int t4k_conv2d_bwd2(const float *I, const float *DO, float *DX, float *DX2, const float *F, float *DF, float *DB,
                    int N, int H1, int W1, int C1, int H0, int W0, int C0,
                    int K, int S, int P, int train, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!conv_supported(K, S, P))
        return fail(T4K_ERR_UNSUPPORTED, "nn#bconv kernel_size=%d stride=%d padding=%d not supported", K, S, P);
    if (!I || !DO || !F || N <= 0 || H0 <= 0 || W0 <= 0) return fail(T4K_ERR_ARG, "t4k_conv2d_bwd: bad argument");
    if ((DF == nullptr) != (DB == nullptr)) return fail(T4K_ERR_ARG, "t4k_conv2d_bwd: DF and DB go together");
    hipStream_t hs = t4k::S(s);
    FoldArgs fa = { nullptr, nullptr, nullptr, 0, 0, 0, 0 };
    if (train && DF) {                                  // DF == NULL: dX only (the caller runs dF|dB on another stream)
        // dF | dB first: they read I, which the host layer may let DX overwrite
        const int ntaps = C1 * K * K, nrow1 = ntaps + 1;
        int nbig = 0;
        if (conv_big_on() && conv_big_ok(C1, C0) && aligned16(I) && aligned16(DO)) {     // many channels: split-K GEMM over pixel slices
            nbig = launch_conv_big_df(K, S, P, hs, I, DO, ws_for(s), st().ws_bytes / 8, N, H1, W1, C1, H0, W0, C0);
            if (nbig > 0) {
                const int ntot = ntaps * C0;                      // no bias row in these slabs: dB is a plain column sum of dO
                // few slices x many outputs: one thread per output walks the slices (coalesced); the wave-per-output fold is
                // for the opposite shape (hundreds of slices, few outputs) and would run 8 of 64 lanes here
                if (nbig <= 32) T4K_LAUNCH(k_fold_add, dim3((ntot + BLK - 1) / BLK), dim3(BLK), 0, hs, ws_for(s), DF, ntot, nbig);
                else T4K_LAUNCH(k_conv_df_fold, dim3((ntot + 3) / 4), dim3(256), 0, hs, ws_for(s), DF, DB, nbig, ntot, ntot);
                int rc = colsum_add(DO, DB, (long)N * H0 * W0, C0, hs); if (rc) return rc;
            }
        }
        int nthin = 0;
        if (nbig == 0 && K == 3 && S == 1 && P == 1 && H0 == H1 && W0 == W1 &&          // image in, a tile or two of channels out: 32 pixels per wave and trip (conv_img.hip)
            conv_thin_df(I, DO, ws_for(s), st().ws_bytes / 2, N, H0, W0, C1, C0, &nthin, hs)) {
            const int ntot = nrow1 * C0;
            fa.part = ws_for(s); fa.DF = DF; fa.DB = DB; fa.nslice = nthin; fa.ndf = ntaps * C0; fa.ntot = ntot; fa.nfold = (ntot + 3) / 4;
        }
        if (nbig == 0 && nthin == 0) {
        const int rows = N * H0;
        // enough slices that ~2000 waves are in flight (each wave then issues only a few batches of loads) without
        // inflating the partial slab the fold has to read: 512 workgroups in total across the (tap, c0) tiles
        const int tiles = ((nrow1 + 31) / 32) * ((C0 + 31) / 32);
        static const int dfwg = std::max(1, T4K_LAB_ENV("T4K_CONV_DF_WG", 512));
        int nslice = (dfwg + tiles - 1) / tiles; if (nslice > (rows + 3) / 4) nslice = (rows + 3) / 4; if (nslice < 1) nslice = 1;
        while (nslice > 1 && (size_t)nslice * nrow1 * C0 * sizeof(float) > st().ws_bytes / 8) nslice >>= 1;
        const int rpw = (rows + nslice * 4 - 1) / (nslice * 4);
        nslice = (rows + rpw * 4 - 1) / (rpw * 4);
        float *part = ws_for(s);
        if ((size_t)nslice * nrow1 * C0 * sizeof(float) > st().ws_bytes / 2) return fail(T4K_ERR_NOMEM, "conv dF workspace");
        dim3 g(nslice, (nrow1 + 31) / 32, (C0 + 31) / 32);
        switch ((K << 8) | (S << 4) | P) {
        case 0x110: T4K_LAUNCH((k_conv_df_mfma<1, 1, 0>), g, dim3(256), 0, hs, I, DO, part, N, H1, W1, C1, H0, W0, C0, rpw); break;
        case 0x311: T4K_LAUNCH((k_conv_df_mfma<3, 1, 1>), g, dim3(256), 0, hs, I, DO, part, N, H1, W1, C1, H0, W0, C0, rpw); break;
        case 0x421: T4K_LAUNCH((k_conv_df_mfma<4, 2, 1>), g, dim3(256), 0, hs, I, DO, part, N, H1, W1, C1, H0, W0, C0, rpw); break;
        case 0x512: T4K_LAUNCH((k_conv_df_mfma<5, 1, 2>), g, dim3(256), 0, hs, I, DO, part, N, H1, W1, C1, H0, W0, C0, rpw); break;
        }
        const int ntot = nrow1 * C0;
        fa.part = part; fa.DF = DF; fa.DB = DB; fa.nslice = nslice; fa.ndf = ntaps * C0; fa.ntot = ntot; fa.nfold = (ntot + 3) / 4;
        }
    }
    int fG = 0, fNG = 0;
    if (DX && conv_big_on() && conv_big_ok(C0, C1) && aligned16(DO) && aligned16(F)) {  // many channels: LDS-staged GEMM tiling
        if (fa.nfold) { T4K_LAUNCH(k_conv_df_fold, dim3(fa.nfold), dim3(256), 0, hs, fa.part, fa.DF, fa.DB, fa.nslice, fa.ndf, fa.ntot); fa.nfold = 0; }
        launch_conv_big<true>(K, S, P, hs, DO, DX, DX2, F, nullptr, N, H0, W0, C0, H1, W1, C1, C0);
        T4K_LAUNCH_CHECK();
        return T4K_OK;
    }
    const bool dx_few = DX && C1 <= 4 && C1 * K * K * C0 <= LDS_FILTER_FLOATS;
    const bool dx_fewch = DX && !dx_few && conv_few_on() && conv_few_ok(K, C0, C1, &fG, &fNG);
    if (fa.nfold && (!DX || dx_fewch)) {                 // no dX kernel to share a launch with: fold on its own
        T4K_LAUNCH(k_conv_df_fold, dim3(fa.nfold), dim3(256), 0, hs, fa.part, fa.DF, fa.DB, fa.nslice, fa.ndf, fa.ntot);
        fa.nfold = 0;
    }
    if (dx_few) {                                       // image-input layer: direct kernel, one thread per input pixel (+ the fold)
        switch (C1) {
        case 1: launch_conv_dx_few<1>(K, S, P, hs, DO, DX, DX2, F, N, H0, W0, C0, H1, W1, fa); break;
        case 2: launch_conv_dx_few<2>(K, S, P, hs, DO, DX, DX2, F, N, H0, W0, C0, H1, W1, fa); break;
        case 3: launch_conv_dx_few<3>(K, S, P, hs, DO, DX, DX2, F, N, H0, W0, C0, H1, W1, fa); break;
        default: launch_conv_dx_few<4>(K, S, P, hs, DO, DX, DX2, F, N, H0, W0, C0, H1, W1, fa); break;
        }
    } else if (dx_fewch) {
        launch_conv_few<true>(K, hs, DO, DX, DX2, nullptr, F, nullptr, N, H0, W0, C0, H1, W1, C1, C0, fG, fNG);
    } else if (DX) {                                    // DX == NULL: dF|dB only; DX2 = optional second copy from the same launch
        const long npix1 = (long)N * H1 * W1;
        const int ksplit = conv_gemm_ksplit(npix1, C1, C0, K);
        const int hx = (int)((npix1 + (128 / ksplit) - 1) / (128 / ksplit)), hy = (C1 + 31) / 32;
        // dX: gather over dO (Hx=H0,Wx=W0,Cin=C0), output the input grid (Hy=H1,Wy=W1,Cout=C1); the dF fold rides along
        const int ppc = LDS_FILTER_FLOATS / (K * K * 2 * 32);
        const dim3 g((unsigned)(fa.nfold + hx * hy));
#define DXF(k, s_, p_) T4K_LAUNCH((k_conv_dx_and_fold<k, s_, p_>), g, dim3(256), 0, hs, fa.part, fa.DF, fa.DB, fa.nslice, fa.ndf, fa.ntot, fa.nfold, \
                                          DO, DX, DX2, F, N, H1, W1, C1, H0, W0, C0, hx, ppc, ksplit)
        switch ((K << 8) | (S << 4) | P) {
        case 0x110: DXF(1, 1, 0); break;
        case 0x311: DXF(3, 1, 1); break;
        case 0x421: DXF(4, 2, 1); break;
        case 0x512: DXF(5, 1, 2); break;
        }
#undef DXF
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
    if (KS == 2) T4K_LAUNCH(k_pool<2>, dim3(grid_for(total)), dim3(BLK), 0, t4k::S(s), layer, I, O, N, H1, W1, H0, W0, C);
    else         T4K_LAUNCH(k_pool<3>, dim3(grid_for(total)), dim3(BLK), 0, t4k::S(s), layer, I, O, N, H1, W1, H0, W0, C);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_dpool(int layer, float *I, const float *DY, int N, int H1, int W1, int H0, int W0, int C, int KS, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (KS != 2 && KS != 3) return fail(T4K_ERR_UNSUPPORTED, "nn#bpool kernel_size=%d not supported", KS);
    if (layer != T4K_L_AVGPOOL && layer != T4K_L_MAXPOOL && layer != T4K_L_MINPOOL && layer != T4K_L_USAMPLE)
        return fail(T4K_ERR_UNSUPPORTED, "t4k_dpool: layer %d", layer);
    const long total = (long)N * H0 * W0 * C; if (total <= 0) return T4K_OK;
    if (KS == 2) T4K_LAUNCH(k_dpool<2>, dim3(grid_for(total)), dim3(BLK), 0, t4k::S(s), layer, I, DY, N, H1, W1, H0, W0, C);
    else         T4K_LAUNCH(k_dpool<3>, dim3(grid_for(total)), dim3(BLK), 0, t4k::S(s), layer, I, DY, N, H1, W1, H0, W0, C);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}

} // extern "C"
