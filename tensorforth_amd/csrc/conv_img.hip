// conv_img.hip - image-input convolution block in one launch: 3x3 / stride 1 / padding 1 convolution of a FEW-channel input (the first
// layer of a CNN: 1 or 3 channels in, <= 16 out) + the element-wise run behind it (activation -> 2x2 pool -> activation -> flatten copy)
// + the model's copy of the batch (layer 0, forward.cu:39).  Same tensors, same values as t4k_conv2d_fwd2 + t4k_poolblock_fwd
// (k_conv2d nmath.tcu:34-104, k_activate nmath.cu:37-70, k_pool nmath.tcu:122-186).
//
// Why not the gather-MFMA kernel (conv.hip) here: with 1 input channel the contraction is 9 deep and 10 of the 32 MFMA columns are live,
// and its epilogue stores from 10 of 32 lanes (14.5 us for the LeNet layer, 8 % of HBM speed).  This kernel is plain vector code shaped
// by the memory system instead: a thread owns one 2x2 pool window = 4 output pixels x all output channels, reads its 4x4 input patch
// once (16 loads per input channel, issued together), takes the filter from LDS as wave-wide broadcasts, and every tensor leaves in runs
// that are contiguous across the wave (2*Cout floats per lane for the conv-sized tensors, Cout per lane for the pooled ones).
#include "t4k_common.h"

using namespace t4k;

namespace {

struct ImgBlk {
    const float *X, *F, *B; float *XC, *Y;
    float *P, *Fpre, *Q, *R, *Fpost, *R2;                // pre_out, pre_mask, pool_out, post_out, post_mask, copy_out (absent: nullptr)
    int pre, pool, post; float a_pre, a_post;
    int N, H, W;                                         // input = output grid of the convolution (even H, W)
};

template <int CIN, int COUT, bool NT>
__global__ void __launch_bounds__(64) k_conv_img_block(ImgBlk p) {
    constexpr int NW = 9 * CIN * COUT;
    __shared__ __attribute__((aligned(16))) float Wl[NW + COUT];
    const int H = p.H, W = p.W, H2 = H >> 1, W2 = W >> 1;
    const long nwin = (long)p.N * H2 * W2;
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    const bool live = t < nwin;
    int px, py, n; split3(live ? t : 0, W2, H2, px, py, n);
    const int y0 = 2 * py, x0 = 2 * px;
    // ---- the 4x4xCIN input patch: unconditional loads from clamped addresses, zeroed afterwards where the tap is outside the image
    float v[4][4][CIN];
    const float *nX = p.X + (long)n * H * W * CIN;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int yy = min(max(y0 + a - 1, 0), H - 1), xx = min(max(x0 + b - 1, 0), W - 1);
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) v[a][b][ci] = nX[((long)yy * W + xx) * CIN + ci];
        }
    // ---- filter F[ci][ky][kx][co] -> Wl[tap][ci][co], bias behind it (while the patch loads are in flight)
    for (int e = threadIdx.x; e < NW; e += 64) {
        const int co = e % COUT, r = e / COUT, tp = r % 9, ci = r / 9;
        Wl[(tp * CIN + ci) * COUT + co] = p.F[e];
    }
    for (int e = threadIdx.x; e < COUT; e += 64) Wl[NW + e] = p.B[e];
    __syncthreads();
    if (!live) return;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const bool in = (y0 + a - 1 >= 0) && (y0 + a - 1 < H) && (x0 + b - 1 >= 0) && (x0 + b - 1 < W);
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) v[a][b][ci] = in ? v[a][b][ci] : 0.f;
        }
    float acc[2][2][COUT];
#pragma unroll
    for (int co = 0; co < COUT; co++) { const float b = Wl[NW + co]; acc[0][0][co] = b; acc[0][1][co] = b; acc[1][0][co] = b; acc[1][1][co] = b; }
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++)
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) {
                const float *w = Wl + ((ky * 3 + kx) * CIN + ci) * COUT;
#pragma unroll
                for (int co = 0; co < COUT; co++) {
                    const float f = w[co];
                    acc[0][0][co] = fmaf(v[ky][kx][ci],         f, acc[0][0][co]);
                    acc[0][1][co] = fmaf(v[ky][kx + 1][ci],     f, acc[0][1][co]);
                    acc[1][0][co] = fmaf(v[ky + 1][kx][ci],     f, acc[1][0][co]);
                    acc[1][1][co] = fmaf(v[ky + 1][kx + 1][ci], f, acc[1][1][co]);
                }
            }
    // ---- layer-0 copy of the batch: this thread's 2x2 input pixels (the centre of its patch)
    if (p.XC) {
        float *nC = p.XC + (long)n * H * W * CIN;
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
            for (int dx = 0; dx < 2; dx++)
#pragma unroll
                for (int ci = 0; ci < CIN; ci++) nC[((long)(y0 + dy) * W + x0 + dx) * CIN + ci] = v[dy + 1][dx + 1][ci];
    }
    // ---- conv-sized tensors: two rows of 2*COUT contiguous floats per thread (adjacent lanes continue the run)
    auto store_rows = [&](float *T, float (&val)[2][2][COUT]) __attribute__((always_inline)) {
#pragma unroll
        for (int dy = 0; dy < 2; dy++) {
            float *row = T + (((long)n * H + y0 + dy) * W + x0) * COUT;
            if (COUT % 2 == 0) {
#pragma unroll
                for (int q = 0; q < COUT; q += 2) {
                    if (NT) { __builtin_nontemporal_store(val[dy][0][q], row + q); __builtin_nontemporal_store(val[dy][0][q + 1], row + q + 1);
                              __builtin_nontemporal_store(val[dy][1][q], row + COUT + q); __builtin_nontemporal_store(val[dy][1][q + 1], row + COUT + q + 1); }
                    else {
                    *reinterpret_cast<float2 *>(row + q)        = make_float2(val[dy][0][q], val[dy][0][q + 1]);
                    *reinterpret_cast<float2 *>(row + COUT + q) = make_float2(val[dy][1][q], val[dy][1][q + 1]); }
                }
            } else {
#pragma unroll
                for (int q = 0; q < COUT; q++) { row[q] = val[dy][0][q]; row[COUT + q] = val[dy][1][q]; }
            }
        }
    };
    store_rows(p.Y, acc);
    if (p.pre) {                                         // activation in front of the pool: output + derivative mask
        float m[2][2][COUT];
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
            for (int dx = 0; dx < 2; dx++)
#pragma unroll
                for (int co = 0; co < COUT; co++) { float o, f; act_rt_lean(p.pre, acc[dy][dx][co], 0.f, p.a_pre, o, f); acc[dy][dx][co] = o; m[dy][dx][co] = f; }
        store_rows(p.P, acc); store_rows(p.Fpre, m);
    }
    // ---- 2x2 pool (window order (0,0) (0,1) (1,0) (1,1) as k_pool), activation behind it, flatten copy
    float q[COUT], r[COUT], fm[COUT];
#pragma unroll
    for (int co = 0; co < COUT; co++) {
        const float e0 = acc[0][0][co], e1 = acc[0][1][co], e2 = acc[1][0][co], e3 = acc[1][1][co];
        float pv;
        if (p.pool == T4K_L_MAXPOOL)      pv = fmaxf(fmaxf(fmaxf(e0, e1), e2), e3);
        else if (p.pool == T4K_L_MINPOOL) pv = fminf(fminf(fminf(e0, e1), e2), e3);
        else                              pv = (((e0 + e1) + e2) + e3) / 4.0f;
        q[co] = pv; r[co] = pv; fm[co] = 0.f;
        if (p.post) { float o, f; act_rt_lean(p.post, pv, 0.f, p.a_post, o, f); r[co] = o; fm[co] = f; }
    }
    auto store_win = [&](float *T, const float (&val)[COUT]) __attribute__((always_inline)) {
        float *d = T + t * COUT;
        if (COUT % 2 == 0) {
#pragma unroll
            for (int c = 0; c < COUT; c += 2) *reinterpret_cast<float2 *>(d + c) = make_float2(val[c], val[c + 1]);
        } else {
#pragma unroll
            for (int c = 0; c < COUT; c++) d[c] = val[c];
        }
    };
    store_win(p.Q, q);
    if (p.post) { store_win(p.R, r); store_win(p.Fpost, fm); }
    if (p.R2) store_win(p.R2, r);
}

template <int CIN, bool NTV>
bool launch_cout(const ImgBlk &p, int Cout, unsigned grid, hipStream_t hs) {
    switch (Cout) {
    case 4:  T4K_LAUNCH((k_conv_img_block<CIN, 4, NTV>),  dim3(grid), dim3(64), 0, hs, p); return true;
    case 6:  T4K_LAUNCH((k_conv_img_block<CIN, 6, NTV>),  dim3(grid), dim3(64), 0, hs, p); return true;
    case 8:  T4K_LAUNCH((k_conv_img_block<CIN, 8, NTV>),  dim3(grid), dim3(64), 0, hs, p); return true;
    case 10: T4K_LAUNCH((k_conv_img_block<CIN, 10, NTV>), dim3(grid), dim3(64), 0, hs, p); return true;
    case 12: T4K_LAUNCH((k_conv_img_block<CIN, 12, NTV>), dim3(grid), dim3(64), 0, hs, p); return true;
    case 16: T4K_LAUNCH((k_conv_img_block<CIN, 16, NTV>), dim3(grid), dim3(64), 0, hs, p); return true;
    default: return false;
    }
}

} // namespace

namespace t4k {
// true when the block was launched here (t4k_conv2d_block_fwd falls through to its other kernels otherwise)
bool conv_img_block_fwd(const float *I, float *ICOPY, float *O, const float *F, const float *B, const t4k_poolblock *blk,
                        int N, int H, int W, int C1, int C0, hipStream_t hs) {
    static int on = -1; if (on < 0) { const char *e = getenv("T4K_CONV_IMG"); on = e ? atoi(e) : 1; }
    if (!on || (C1 != 1 && C1 != 3) || C0 > 16 || (H & 1) || (W & 1) || blk->KS != 2 || !blk->pool_layer) return false;
    if (blk->pre_layer == T4K_L_DROPOUT || blk->post_layer == T4K_L_DROPOUT) return false;      // mask draws stay with the Philox-carrying kernels
    if ((C0 & 1) == 0 && (!aligned16(O) || !aligned16(blk->pool_out))) return false;
    ImgBlk p;
    p.X = I; p.F = F; p.B = B; p.XC = ICOPY; p.Y = O;
    p.P = blk->pre_out; p.Fpre = blk->pre_mask; p.Q = blk->pool_out; p.R = blk->post_out; p.Fpost = blk->post_mask; p.R2 = blk->copy_out;
    p.pre = blk->pre_layer; p.pool = blk->pool_layer; p.post = blk->post_layer; p.a_pre = blk->pre_alpha; p.a_post = blk->post_alpha;
    p.N = N; p.H = H; p.W = W;
    const long nwin = (long)N * (H / 2) * (W / 2);
    const unsigned grid = (unsigned)((nwin + 63) / 64);
    static int nt = -1; if (nt < 0) { const char *e = getenv("T4K_CONV_IMG_NT"); nt = e ? atoi(e) : 1; }
    if (nt) return C1 == 1 ? launch_cout<1, true>(p, C0, grid, hs) : launch_cout<3, true>(p, C0, grid, hs);
    return C1 == 1 ? launch_cout<1, false>(p, C0, grid, hs) : launch_cout<3, false>(p, C0, grid, hs);
}
}
