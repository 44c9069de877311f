// conv_direct.hip - conv2d forward / dX / dF|dB for SMALL channel counts (C1*C0 <= 512) on the vector ALUs.
//
// The LeNet-class layers of the headline config (1->10 and 10->20 channels, 3x3) are HBM/latency bound:
// 18-90 MFLOP against 1-4 MB of traffic.  Feeding the 32x32 matrix cores would mean padding 10 channels to
// 32 and gathering one predicated global load per MFMA, so these shapes run as direct convolutions instead:
//   * the filter (<= 32 KiB) is staged once per workgroup in LDS, output channels innermost, so a thread
//     fetches the 4 weights of its channel group with one ds_read_b128 (lanes of a wave that share the
//     group read the same address: broadcast, no bank conflicts);
//   * a thread owns PX adjacent output pixels x 4 output channels in registers; adjacent lanes own adjacent
//     channel groups, then adjacent pixels => loads and stores of a wave are contiguous in NHWC memory;
//   * dF|dB: a lane owns one (tap, c1) row x 4 output channels and walks a strip of pixels, so there is no
//     cross-lane reduction at all; strips are folded in fixed order (deterministic, no fp32 atomics).
// Arithmetic restates k_conv2d / k_dconv2d (src/nn/nmath.tcu:34-104, 211-338) incl. the flipped-filter dX.
// Larger channel counts take the MFMA implicit-GEMM kernels in conv.hip.
#include "t4k_common.h"

using namespace t4k;

namespace t4k {

constexpr int CD_LDS_FLOATS = 8192;           // 32 KiB filter stage

// ------------------------------------------------------------------ forward / dX
// BWD = false: Y[n,y,x,co] = B[co] + sum X[n, y*S+ky-P, x*S+kx-P, ci] * F[ci,ky,kx,co]           (Cin = C1, Cout = C0)
// BWD = true : Y[n,y,x,co] =         sum X[n,(y+P-ky)/S,(x+P-kx)/S, ci] * F[co,K-1-ky,K-1-kx,ci]  (Cin = C0, Cout = C1)
template <int K, int S, int P, bool BWD, int PX>
__global__ void __launch_bounds__(256) k_conv_direct(const float *__restrict__ X, float *__restrict__ Y, float *__restrict__ Y2,
                                                     const float *__restrict__ F, const float *__restrict__ B,
                                                     int N, int Hx, int Wx, int Cin, int Hy, int Wy, int Cout, int C0f) {
    __shared__ __attribute__((aligned(16))) float Wl[CD_LDS_FLOATS];
    const int NG = (Cout + 3) >> 2, COP = NG * 4;
    // ---- stage the filter: Wl[((ky*K+kx)*Cin + ci)*COP + co]
    const int nent = K * K * Cin * COP;
    for (int e = threadIdx.x; e < nent; e += 256) {
        const int co = e % COP; int t = e / COP;
        const int ci = t % Cin; t /= Cin;
        const int kx = t % K, ky = t / K;
        float v = 0.f;
        if (co < Cout) {
            if (!BWD) v = F[((long)(ci * K + ky) * K + kx) * C0f + co];
            else      v = F[((long)(co * K + (K - 1 - ky)) * K + (K - 1 - kx)) * C0f + ci];
        }
        Wl[e] = v;
    }
    __syncthreads();
    const int WyP = (Wy + PX - 1) / PX;                          // pixel groups per row
    const long ngrp = (long)N * Hy * WyP * NG;
    for (long z = (long)blockIdx.x * 256 + threadIdx.x; z < ngrp; z += (long)gridDim.x * 256) {
        const int g = (int)(z % NG); long t = z / NG;
        const int xg = (int)(t % WyP); t /= WyP;
        const int y = (int)(t % Hy); const int n = (int)(t / Hy);
        const int x0 = xg * PX;
        float acc[PX][4];
#pragma unroll
        for (int q = 0; q < PX; q++)
#pragma unroll
            for (int u = 0; u < 4; u++) acc[q][u] = 0.f;
        const float *nX = X + (long)n * Hx * Wx * Cin;
#pragma unroll
        for (int ky = 0; ky < K; ky++) {
            int gi; bool iok;
            if (!BWD) { gi = y * S + ky - P; iok = gi >= 0 && gi < Hx; }
            else { const int ti = y + P - ky; gi = ti / S; iok = ti >= 0 && (ti % S) == 0 && gi < Hx; }
            if (!iok) continue;
#pragma unroll
            for (int kx = 0; kx < K; kx++) {
                const float *wl = Wl + ((ky * K + kx) * Cin) * COP + g * 4;
                const float *px[PX]; bool ok[PX];
#pragma unroll
                for (int q = 0; q < PX; q++) {
                    int gj; bool jok;
                    if (!BWD) { gj = (x0 + q) * S + kx - P; jok = gj >= 0 && gj < Wx; }
                    else { const int tj = x0 + q + P - kx; gj = tj / S; jok = tj >= 0 && (tj % S) == 0 && gj < Wx; }
                    ok[q] = jok && (x0 + q) < Wy;
                    px[q] = nX + ((long)gi * Wx + (ok[q] ? gj : 0)) * Cin;
                }
                for (int ci = 0; ci < Cin; ci++) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(wl + ci * COP);
#pragma unroll
                    for (int q = 0; q < PX; q++) {
                        const float xv = ok[q] ? px[q][ci] : 0.f;
                        acc[q][0] = fmaf(xv, w4.x, acc[q][0]); acc[q][1] = fmaf(xv, w4.y, acc[q][1]);
                        acc[q][2] = fmaf(xv, w4.z, acc[q][2]); acc[q][3] = fmaf(xv, w4.w, acc[q][3]);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PX; q++) {
            if (x0 + q >= Wy) continue;
            const long o = (((long)n * Hy + y) * Wy + x0 + q) * Cout + g * 4;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int co = g * 4 + u;
                if (co < Cout) {
                    const float v = acc[q][u] + ((!BWD && B) ? B[co] : 0.f);
                    Y[o + u] = v;
                    if (Y2) Y2[o + u] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ dF | dB
// virtual lane v = row * NG + g: row = (c1*K+ky)*K+kx (filter order) or the bias row `ntaps`; g = group of 4 c0.
// A workgroup = SUB sub-strips x LV lanes; sub-strips are reduced through LDS in fixed order, then one partial per
// (strip, row, c0) goes to the workspace slab laid out [strip][ntaps+1][C0] for k_conv_df_fold.
template <int K, int S, int P>
__global__ void __launch_bounds__(256) k_conv_df_direct(const float *__restrict__ I, const float *__restrict__ DO, float *__restrict__ part,
                                                        int N, int H1, int W1, int C1, int H0, int W0, int C0,
                                                        int LV, int pix_per_sub) {
    __shared__ float red[256 * 4];
    const int NG = (C0 + 3) >> 2;
    const int ntaps = C1 * K * K, V = (ntaps + 1) * NG;
    const int SUB = 256 / LV;
    const int sub = threadIdx.x / LV, lv = threadIdx.x - sub * LV;
    const int v = blockIdx.y * LV + lv;
    const bool live = v < V;
    const int row = live ? v / NG : 0, g = live ? v - row * NG : 0;
    const bool is_bias = row == ntaps;
    int c1 = 0, ky = 0, kx = 0;
    if (!is_bias) { kx = row % K; ky = (row / K) % K; c1 = row / (K * K); }
    const long npix = (long)N * H0 * W0;
    const long p0 = ((long)blockIdx.x * SUB + sub) * pix_per_sub;
    const long p1 = min(npix, p0 + pix_per_sub);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (live && p0 < p1) {
        int x = (int)(p0 % W0); long t = p0 / W0; int y = (int)(t % H0); int n = (int)(t / H0);
        const int cb = g * 4;
        const bool m1 = cb + 1 < C0, m2 = cb + 2 < C0, m3 = cb + 3 < C0;
        for (long p = p0; p < p1; p++) {
            const int gi = y * S + ky - P, gj = x * S + kx - P;
            float xv = 1.f;
            if (!is_bias) xv = (gi >= 0 && gi < H1 && gj >= 0 && gj < W1) ? I[(((long)n * H1 + gi) * W1 + gj) * C1 + c1] : 0.f;
            const float *d = DO + p * C0 + cb;
            a0 = fmaf(xv, d[0], a0);
            if (m1) a1 = fmaf(xv, d[1], a1);
            if (m2) a2 = fmaf(xv, d[2], a2);
            if (m3) a3 = fmaf(xv, d[3], a3);
            if (++x == W0) { x = 0; if (++y == H0) { y = 0; n++; } }
        }
    }
    if (SUB > 1) {                                               // fixed-order reduction over the sub-strips
        red[threadIdx.x * 4 + 0] = a0; red[threadIdx.x * 4 + 1] = a1; red[threadIdx.x * 4 + 2] = a2; red[threadIdx.x * 4 + 3] = a3;
        __syncthreads();
        if (sub == 0) {
            for (int s2 = 1; s2 < SUB; s2++) {
                const float *r = red + (s2 * LV + lv) * 4;
                a0 += r[0]; a1 += r[1]; a2 += r[2]; a3 += r[3];
            }
        }
    }
    if (sub == 0 && live) {
        float *o = part + ((long)blockIdx.x * (ntaps + 1) + row) * C0 + g * 4;
        o[0] = a0;
        if (g * 4 + 1 < C0) o[1] = a1;
        if (g * 4 + 2 < C0) o[2] = a2;
        if (g * 4 + 3 < C0) o[3] = a3;
    }
}

bool conv_direct_ok(int K, int C1, int C0) {
    const int cmax = C1 > C0 ? C1 : C0;
    return (long)C1 * C0 <= 512 && K * K * cmax * (((cmax + 3) >> 2) * 4) <= CD_LDS_FLOATS
        && K * K * C1 * (((C0 + 3) >> 2) * 4) <= CD_LDS_FLOATS && K * K * C0 * (((C1 + 3) >> 2) * 4) <= CD_LDS_FLOATS;
}

template <bool BWD>
void launch_conv_direct(int K, int S, int P, hipStream_t hs, const float *X, float *Y, float *Y2, const float *F, const float *B,
                        int N, int Hx, int Wx, int Cin, int Hy, int Wy, int Cout, int C0f) {
    const int NG = (Cout + 3) >> 2;
    const long ngrp = (long)N * Hy * ((Wy + 1) / 2) * NG;
    long gx = (ngrp + 255) / 256; if (gx > 4096) gx = 4096; if (gx < 1) gx = 1;
    const dim3 g((unsigned)gx), b(256);
#define CD_CASE(k, s, p) hipLaunchKernelGGL((k_conv_direct<k, s, p, BWD, 2>), g, b, 0, hs, X, Y, Y2, F, B, N, Hx, Wx, Cin, Hy, Wy, Cout, C0f)
    switch ((K << 8) | (S << 4) | P) {
    case 0x110: CD_CASE(1, 1, 0); break;
    case 0x311: CD_CASE(3, 1, 1); break;
    case 0x421: CD_CASE(4, 2, 1); break;
    case 0x512: CD_CASE(5, 1, 2); break;
    }
#undef CD_CASE
}
template void launch_conv_direct<false>(int, int, int, hipStream_t, const float *, float *, float *, const float *, const float *, int, int, int, int, int, int, int, int);
template void launch_conv_direct<true>(int, int, int, hipStream_t, const float *, float *, float *, const float *, const float *, int, int, int, int, int, int, int, int);

// returns the number of strips written to `part` ([strip][ntaps+1][C0]); 0 when the workspace is too small
int launch_conv_df_direct(int K, int S, int P, hipStream_t hs, const float *I, const float *DO, float *part, size_t part_floats,
                          int N, int H1, int W1, int C1, int H0, int W0, int C0) {
    const int NG = (C0 + 3) >> 2, ntaps = C1 * K * K, V = (ntaps + 1) * NG;
    int LV = 32; while (LV < V && LV < 256) LV <<= 1;
    const int SUB = 256 / LV;
    const int gy = (V + LV - 1) / LV;
    const long npix = (long)N * H0 * W0;
    // ~2 workgroups per CU along x when there is enough work; at least 32 pixels per sub-strip
    long want_sub = (long)st().cu_count * 2 * SUB / gy; if (want_sub < SUB) want_sub = SUB;
    long pps = (npix + want_sub - 1) / want_sub; if (pps < 32) pps = 32;
    const long nsub = (npix + pps - 1) / pps;
    const int gx = (int)((nsub + SUB - 1) / SUB);
    if ((size_t)gx * (ntaps + 1) * C0 > part_floats) return 0;
    const dim3 g(gx, gy), b(256);
#define DF_CASE(k, s, p) hipLaunchKernelGGL((k_conv_df_direct<k, s, p>), g, b, 0, hs, I, DO, part, N, H1, W1, C1, H0, W0, C0, LV, (int)pps)
    switch ((K << 8) | (S << 4) | P) {
    case 0x110: DF_CASE(1, 1, 0); break;
    case 0x311: DF_CASE(3, 1, 1); break;
    case 0x421: DF_CASE(4, 2, 1); break;
    case 0x512: DF_CASE(5, 1, 2); break;
    }
#undef DF_CASE
    return gx;
}

} // namespace t4k
