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


// ---- image-input convolution with MANY output channels (3 -> 64, the first layer of a CIFAR-style net), 3x3 / stride 1 / padding 1.
// The generic gather kernel (conv.hip) restages the filter in LDS per workgroup and gives every wave ONE 32x32 tile behind two dependent
// memory round trips: 37 us for 256 x 32 x 32 x 3 -> 64, four times the 67 MB its output costs.  Here the whole contraction (9 * CIN <= 36
// deep) is NS <= 18 MFMA steps whose B operands - the filter - live in registers for the life of the wave; a wave walks 32-pixel tiles,
// gathers its NS input values per lane (unconditional loads from clamped addresses), feeds them to all NT column tiles, stores.  The
// layer-0 copy of the batch (forward.cu:39) leaves from the registers that hold the centre tap.
typedef float f32x16i __attribute__((ext_vector_type(16)));
typedef float f32x4i __attribute__((ext_vector_type(4)));
template <int CIN, int NT, bool COPY, bool NTS, bool STAT>
__global__ void __launch_bounds__(256) k_conv_thin_fwd(const float *__restrict__ X, float *__restrict__ Y, float *__restrict__ XC,
                                                       const float *__restrict__ F, const float *__restrict__ B, int N, int H, int W, long ntile, float *__restrict__ part) {
    constexpr int KK = 9 * CIN, NS = (KK + 1) / 2, COUT = NT * 32;
    __shared__ __attribute__((aligned(16))) float Os[4 * 32 * COUT];
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6), GW = (long)gridDim.x * 4;
    const long npix = (long)N * H * W, nfull = npix / 32;    // tiles with all 32 pixels inside
    float bf[NT][NS], bias[NT];
    int dko[NS]; unsigned m_top = 0, m_bot = 0, m_lft = 0, m_rgt = 0, m_pad = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int k = 2 * s + h, tap = k / CIN, ci = k - tap * CIN, ky = tap / 3, kx = tap - ky * 3;
        const bool live = k < KK;
        dko[s] = live ? ((ky - 1) * W + (kx - 1)) * CIN + ci : 0;
        if (!live) m_pad |= 1u << s;
        else { if (ky == 0) m_top |= 1u << s; if (ky == 2) m_bot |= 1u << s; if (kx == 0) m_lft |= 1u << s; if (kx == 2) m_rgt |= 1u << s; }
#pragma unroll
        for (int t = 0; t < NT; t++) bf[t][s] = live ? F[(long)(ci * 9 + tap) * COUT + t * 32 + l31] : 0.f;     // F[c1][ky][kx][c0]
    }
#pragma unroll
    for (int t = 0; t < NT; t++) bias[t] = B[t * 32 + l31];
    auto gather = [&](long T, float (&a)[NS], unsigned &dead) __attribute__((always_inline)) {
        const long p = T * 32 + l31;
        const bool pv = p < npix;
        const unsigned q = pv ? (unsigned)p : 0u, tq = q / (unsigned)W, x = q - tq * (unsigned)W, y = tq % (unsigned)H;      // 32-bit, branch-free (the launcher checks the size)
        dead = m_pad | (y == 0 ? m_top : 0u) | (y == (unsigned)H - 1 ? m_bot : 0u) | (x == 0 ? m_lft : 0u) | (x == (unsigned)W - 1 ? m_rgt : 0u) | (pv ? 0u : ~0u);
        const float *px = X + (pv ? p : 0) * CIN;
#pragma unroll
        for (int s = 0; s < NS; s++) a[s] = px[(dead >> s) & 1u ? 0 : dko[s]];     // unconditional loads from clamped addresses
    };
    // Tile T's products go to the wave's LDS tile and leave from there - 16 bytes per lane, every store instruction one contiguous KiB - between
    // the MFMAs of the next tile; the loop over whole tiles is ONE basic block (first tile peeled, the prefetch index clamped instead of guarded,
    // the ragged last tile on a path of its own) so that its waits are counted ones.  Measured at 256 x 32 x 32 x 3 -> 64 (67 MB of output, a
    // 10 us memset): 17 us without the layer-0 copy, 21 us with it, against 37 us + a 3.7 us copy on the generic gather kernel.
    float *os = Os + (threadIdx.x >> 6) * (32 * COUT);
    float cs[NT], cq[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) { cs[t] = 0.f; cq[t] = 0.f; }
    constexpr int RPI = 256 / COUT, NST = 32 / RPI;          // tile rows per store instruction (64 lanes x 4 channels), store instructions per tile
    const int rr = lane / (COUT / 4), c4 = (lane % (COUT / 4)) * 4;
    auto store_piece = [&](long Tp, int i) __attribute__((always_inline)) {
        const int row = i * RPI + rr;
        const f32x4i v = *reinterpret_cast<const f32x4i *>(os + row * COUT + c4);
        if (NTS) __builtin_nontemporal_store(v, reinterpret_cast<f32x4i *>(Y + (Tp * 32 + row) * COUT + c4));
        else *reinterpret_cast<f32x4i *>(Y + (Tp * 32 + row) * COUT + c4) = v;
    };
    auto to_lds = [&](const f32x16i (&acc)[NT]) __attribute__((always_inline)) {
        __builtin_amdgcn_wave_barrier();                    // LDS executes a wave's accesses in order: the reads of the previous tile are done with before these writes
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float v = acc[t][r] + bias[t];
                os[((r & 3) + 8 * (r >> 2) + 4 * h) * COUT + t * 32 + l31] = v;   // D[row = pixel][col = channel]: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h
                if (STAT) { cs[t] += v; cq[t] = fmaf(v, v, cq[t]); }             // STAT: the per-channel sums of a batch-norm layer behind this one ride along (a wave's tiles; npix % 32 == 0)
            }
        __builtin_amdgcn_wave_barrier();
    };
    constexpr int XL = 8 * CIN;                              // the tile's own 32 x CIN input floats = XL 16-byte pieces: the layer-0 copy (forward.cu:39);
    const int xl = lane % XL;                                // lanes past XL repeat a piece (same value to the same address) rather than diverge
    float an[NS]; unsigned dn = 0; f32x4i xn = { 0.f, 0.f, 0.f, 0.f };
    long T = gw;
    if (T < nfull) {
        gather(T, an, dn);
        if (COPY) xn = *reinterpret_cast<const f32x4i *>(X + T * 32 * CIN + xl * 4);
        long Tprev; float pad[NST];
#pragma unroll
        for (int i = 0; i < NST; i++) pad[i] = 0.f;
        {                                                   // first tile: nothing to store yet
            float a[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) a[s] = (dn >> s) & 1u ? 0.f : an[s];
            if (COPY) *reinterpret_cast<f32x4i *>(XC + T * 32 * CIN + xl * 4) = xn;
            f32x16i acc[NT];
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bf[t][s], acc[t], 0, 0, 0);
            to_lds(acc);
            Tprev = T; T += GW;
        }
        if (T < nfull) {
            if (COPY) xn = *reinterpret_cast<const f32x4i *>(X + T * 32 * CIN + xl * 4);
            gather(T, an, dn);
            // NST loads nobody waits for until the loop is over, behind the gathers: vmcnt is ONE in-order counter, and the compiler merges what is
            // pending on the two ways into the loop.  On the back edge NST stores are younger than the gathers; without as many younger operations
            // on this side the merged state says "the last gather is the youngest operation" - s_waitcnt vmcnt(0) at the loop head, i.e. every tile
            // waits for the previous tile's stores to retire, which is exactly what the pipeline is there to avoid.
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NST; i++) pad[i] = X[(l31 + i) & 31];
            __builtin_amdgcn_sched_barrier(0);
        }
        for (; T < nfull; T += GW) {
            float a[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) a[s] = (dn >> s) & 1u ? 0.f : an[s];
            if (COPY) *reinterpret_cast<f32x4i *>(XC + T * 32 * CIN + xl * 4) = xn;
            const long Tn = T + GW < nfull ? T + GW : T;
            if (COPY) xn = *reinterpret_cast<const f32x4i *>(X + Tn * 32 * CIN + xl * 4);
            gather(Tn, an, dn);
            f32x16i acc[NT];
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NS; s++) {
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bf[t][s], acc[t], 0, 0, 0);
                if (s < NST) { __builtin_amdgcn_sched_barrier(0); store_piece(Tprev, s); __builtin_amdgcn_sched_barrier(0); }
            }
#pragma unroll
            for (int i = NS; i < NST; i++) store_piece(Tprev, i);
            to_lds(acc);
            Tprev = T;
        }
#pragma unroll
        for (int i = 0; i < NST; i++) store_piece(Tprev, i);
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < NST; i++) ps += pad[i];
        if (ps == 1.2345e-37f) os[lane] = ps;               // keeps the padding loads' registers pending through the loop (never true in effect: LDS only)
    }
    if (STAT) {                                             // one partial row pair per workgroup: [workgroup][sum y | sum y^2][COUT], folded by k_bn_fin in fixed order
        __builtin_amdgcn_wave_barrier();                    // the wave's own LDS tile is free: its last stores have read it
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const float s1 = cs[t] + __shfl_xor(cs[t], 32), s2 = cq[t] + __shfl_xor(cq[t], 32);
            if (h == 0) { os[t * 32 + l31] = s1; os[COUT + t * 32 + l31] = s2; }
        }
        __syncthreads();
        if (threadIdx.x < 2 * COUT)
            part[(long)blockIdx.x * 2 * COUT + threadIdx.x] = (Os[threadIdx.x] + Os[32 * COUT + threadIdx.x]) + (Os[2 * 32 * COUT + threadIdx.x] + Os[3 * 32 * COUT + threadIdx.x]);
    }
    if (nfull < ntile && gw == nfull % GW) {                // the ragged last tile: guarded everything, stored straight from the accumulators
        float a[NS]; unsigned dd;
        gather(nfull, a, dd);
#pragma unroll
        for (int s = 0; s < NS; s++) a[s] = (dd >> s) & 1u ? 0.f : a[s];
        f32x16i acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bf[t][s], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long p2 = nfull * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (p2 < npix) Y[p2 * COUT + t * 32 + l31] = acc[t][r] + bias[t];
            }
        if (COPY) for (long e = nfull * 32 * CIN + lane; e < npix * CIN; e += 64) XC[e] = X[e];
    }
}

// ---- dF | dB of the same layer: dF[c1][tap][c0] += sum over pixels of x[pixel + tap][c1] * dO[pixel][c0], a 28 x 64 output reduced over N*H*W pixels.
// The generic kernel (conv.hip, k_conv_df_mfma) walks an image row per wave in trips of 7 pixel pairs, each trip a dependent memory round trip
// (38 us for 256 x 32 x 32 x 3 -> 64).  Here a wave takes 32 pixels at a time: all 16 gathers of its A row (lane = filter row c1 * 9 + tap, row
// 9 * CIN fed 1.0 = dB; the MFMA's k pair = two adjacent pixels) and the 16 * NT loads of dO go out together, then 16 * NT MFMAs; the accumulators
// live in registers across all of the wave's tiles, the four waves meet in LDS, one slab row per workgroup for k_conv_df_fold.
template <int CIN, int NT>
__global__ void __launch_bounds__(256) k_conv_thin_df(const float *__restrict__ X, const float *__restrict__ DO, float *__restrict__ part,
                                                      int N, int H, int W, long ntile) {
    constexpr int NTAP = 9 * CIN, COUT = NT * 32, NROW = NTAP + 1;
    static_assert(NROW <= 32, "one MFMA row tile");
    __shared__ float red[4][32][COUT];
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
    const long gw = (long)blockIdx.x * 4 + w, GW = (long)gridDim.x * 4;
    const long npix = (long)N * H * W;
    const bool is_tap = l31 < NTAP, is_bias = l31 == NTAP;
    const int ci = l31 / 9, tap = l31 - ci * 9, ky = tap / 3, kx = tap - ky * 3;
    const int dko = is_tap ? ((ky - 1) * W + (kx - 1)) * CIN + ci : 0;
    f32x16i acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    for (long T = gw; T < ntile; T += GW) {
        const long p0 = T * 32 + h;                          // this lane half's pixels: p0, p0 + 2, ...
        const unsigned q = (unsigned)(p0 < npix ? p0 : npix - 1), tq = q / (unsigned)W;
        unsigned x = q - tq * (unsigned)W, y = tq % (unsigned)H;
        float a[16], b[NT][16]; unsigned deadm = 0, okm = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const long pj = p0 + 2 * j;
            const bool ok = pj < npix;
            const bool dead = !ok || !is_tap || (ky == 0 && y == 0) || (ky == 2 && y == (unsigned)H - 1) || (kx == 0 && x == 0) || (kx == 2 && x == (unsigned)W - 1);
            a[j] = X[dead ? 0 : pj * CIN + dko];
            deadm |= (dead ? 1u : 0u) << j; okm |= (ok ? 1u : 0u) << j;
#pragma unroll
            for (int t = 0; t < NT; t++) b[t][j] = DO[(ok ? pj : 0) * COUT + t * 32 + l31];
            x += 2; if (x >= (unsigned)W) { x -= (unsigned)W; y = y + 1 == (unsigned)H ? 0u : y + 1; }     // W >= 2 (the launcher checks)
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const bool ok = (okm >> j) & 1u;
            const float av = is_bias ? (ok ? 1.f : 0.f) : ((deadm >> j) & 1u ? 0.f : a[j]);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, ok ? b[t][j] : 0.f, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) red[w][(r & 3) + 8 * (r >> 2) + 4 * h][t * 32 + l31] = acc[t][r];
    __syncthreads();
    for (int e = threadIdx.x; e < NROW * COUT; e += 256) {
        const int gt = e / COUT, gc = e - gt * COUT;
        part[((long)blockIdx.x * NROW + gt) * COUT + gc] = (red[0][gt][gc] + red[1][gt][gc]) + (red[2][gt][gc] + red[3][gt][gc]);
    }
}

} // namespace

namespace t4k {
// true when the layer was launched here (t4k_conv2d_fwd2 falls through to its other kernels otherwise); ICOPY may be null
bool conv_thin_fwd(const float *I, float *ICOPY, float *O, const float *F, const float *B, int N, int H, int W, int C1, int C0, hipStream_t hs,
                   float *bn_part, size_t bn_part_floats, int *bn_chunks) {
    if (bn_chunks) *bn_chunks = 0;
    static const int on = T4K_LAB_ENV("T4K_CONV_THIN", 1);
    if (!on || C1 < 1 || C1 > 4 || (C0 != 32 && C0 != 64) || (long)N * H * W >= 0x7fffff00L) return false;
    if (!aligned16(I) || !aligned16(O) || (ICOPY && !aligned16(ICOPY))) return false;       // 16-byte pieces of the batch copy and of the output rows
    const long ntile = ((long)N * H * W + 31) / 32;
    static const int cap = std::max(1, T4K_LAB_ENV("T4K_CONV_THIN_WG", 512));
    long wg = (ntile + 3) / 4; if (wg > cap) wg = cap;         // a wave walks ntile / (4 wg) tiles with its filter in registers
    const dim3 g((unsigned)wg), b(256);
    static const int nts = T4K_LAB_ENV("T4K_CONV_THIN_NT", 1);
    const bool stat = bn_part && bn_chunks && ((long)N * H * W) % 32 == 0 && (size_t)wg * 2 * C0 <= bn_part_floats;    // batch-norm sums from the epilogue: whole tiles only
    if (stat) *bn_chunks = (int)wg;
#define T4K_THIN3(C_, T_, N_, S_) do { if (ICOPY) T4K_LAUNCH((k_conv_thin_fwd<C_, T_, true, N_, S_>), g, b, 0, hs, I, O, ICOPY, F, B, N, H, W, ntile, bn_part); \
                                   else T4K_LAUNCH((k_conv_thin_fwd<C_, T_, false, N_, S_>), g, b, 0, hs, I, O, ICOPY, F, B, N, H, W, ntile, bn_part); } while (0)
#define T4K_THIN2(C_, T_, N_) do { if (stat) T4K_THIN3(C_, T_, N_, true); else T4K_THIN3(C_, T_, N_, false); } while (0)
#define T4K_THIN(C_, T_) do { if (nts) T4K_THIN2(C_, T_, true); else T4K_THIN2(C_, T_, false); } while (0)
    switch (C1 * 4 + C0 / 32) {
    case 5: T4K_THIN(1, 1); break; case 6: T4K_THIN(1, 2); break;
    case 9: T4K_THIN(2, 1); break; case 10: T4K_THIN(2, 2); break;
    case 13: T4K_THIN(3, 1); break; case 14: T4K_THIN(3, 2); break;
    case 17: T4K_THIN(4, 1); break; case 18: T4K_THIN(4, 2); break;
    default: return false;
    }
#undef T4K_THIN
#undef T4K_THIN2
#undef T4K_THIN3
    return true;
}
// dF | dB partial slabs of the same layer: true when launched here, *nslice = slab rows ((9 C1 + 1) x C0 floats each) for k_conv_df_fold
bool conv_thin_df(const float *I, const float *DO, float *part, size_t part_bytes, int N, int H, int W, int C1, int C0, int *nslice, hipStream_t hs) {
    static const int on = T4K_LAB_ENV("T4K_CONV_THIN_DF", 1);
    if (!on || C1 < 1 || C1 > 3 || (C0 != 32 && C0 != 64) || W < 2 || (long)N * H * W >= 0x7fffff00L) return false;
    const long ntile = ((long)N * H * W + 31) / 32;
    static const int cap = std::max(1, T4K_LAB_ENV("T4K_CONV_THIN_DF_WG", 512));
    long wg = (ntile + 3) / 4; if (wg > cap) wg = cap;
    while (wg > 1 && (size_t)wg * (9 * C1 + 1) * C0 * sizeof(float) > part_bytes) wg >>= 1;
    if ((size_t)wg * (9 * C1 + 1) * C0 * sizeof(float) > part_bytes) return false;
    const dim3 g((unsigned)wg), b(256);
#define T4K_TDF(C_, T_) T4K_LAUNCH((k_conv_thin_df<C_, T_>), g, b, 0, hs, I, DO, part, N, H, W, ntile)
    switch (C1 * 4 + C0 / 32) {
    case 5: T4K_TDF(1, 1); break; case 6: T4K_TDF(1, 2); break;
    case 9: T4K_TDF(2, 1); break; case 10: T4K_TDF(2, 2); break;
    case 13: T4K_TDF(3, 1); break; case 14: T4K_TDF(3, 2); break;
    default: return false;
    }
#undef T4K_TDF
    *nslice = (int)wg;
    return true;
}
// true when the block was launched here (t4k_conv2d_block_fwd falls through to its other kernels otherwise)
bool conv_img_block_fwd(const float *I, float *ICOPY, float *O, const float *F, const float *B, const t4k_poolblock *blk,
                        int N, int H, int W, int C1, int C0, hipStream_t hs) {
    static const int on = T4K_LAB_ENV("T4K_CONV_IMG", 1);
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
    static const int nt = T4K_LAB_ENV("T4K_CONV_IMG_NT", 1);
    if (nt) return C1 == 1 ? launch_cout<1, true>(p, C0, grid, hs) : launch_cout<3, true>(p, C0, grid, hs);
    return C1 == 1 ? launch_cout<1, false>(p, C0, grid, hs) : launch_cout<3, false>(p, C0, grid, hs);
}
}
