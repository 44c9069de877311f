// conv_big.hip - conv2d forward / dX / dF for MANY channels (Cin % 32 == 0, Cout % 4 == 0) as LDS-staged MFMA GEMMs.
//
// The gather kernels in conv.hip feed every MFMA with one predicated global float per lane; that is fine for the
// 1..20-channel LeNet layers (latency bound anyway) but reaches only ~22 % of the fp32 MFMA peak at 64-128 channels.
// Here the implicit GEMM is tiled exactly like the dense GEMM in gemm.hip (same LDS layouts, XOR swizzle, k-permuted
// ds_read_b128, 2x2 waves of 32x32 MFMA blocks, double-buffered stages):
//   forward  Y[pix, co]  = sum_{tap,ci} X[pix (+) tap, ci] * F[ci, tap, co]            A k-contiguous, B n-contiguous
//   dX       dX[pix, c1] = sum_{tap,c0} dO[pix (-) tap, c0] * F[c1, K*K-1-tap, c0]     A k-contiguous, B k-contiguous
//   dF       dF[ci, tap, co] += sum_pix X[pix (+) tap, ci] * dO[pix, co]               A m-contiguous, B n-contiguous,
//            split over pixel slices, partial slabs folded in fixed order (k_conv_df_fold) - no fp32 atomics
// One stage = 32 channels of ONE tap (Cin % 32 == 0), so the tap shift is uniform per stage and a thread's pixel rows
// are decomposed into (n, y, x) once.  NHWC makes every operand row a contiguous 128-byte run: all global loads are
// coalesced 16-byte loads, out-of-image taps load nothing and stage zeros.
// Arithmetic restates k_conv2d / k_dconv2d (src/nn/nmath.tcu:34-104, 211-338) incl. the flipped-filter dX.
#include "t4k_common.h"

using namespace t4k;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor over [p, p + bytes): offsets are bytes, a lane offset >= bytes is out of range (loads return / the LDS-DMA stores zeros)
__device__ __forceinline__ i32x4 buf_srd(const void *p, unsigned bytes) {
    const unsigned long a = (unsigned long)p;
    i32x4 r; r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xFFFFu); r[2] = (int)bytes; r[3] = 0x00020000;
    return r;
}
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int BK = 32, CH = BK / 4, SW = 64 / BK, NC = BK / 8;

struct CbP {
    const float *X, *F, *B;            // X: forward input / dX: dO;  F: filter [C1][K][K][C0];  B: bias (forward only)
    float *Y, *Y2;                     // output (+ optional second copy)
    int N, Hx, Wx, Cin, Hy, Wy, Cout;  // gather grid (Hx, Wx, Cin) -> output grid (Hy, Wy, Cout)
    int C0f;                           // filter inner dimension (reference C0)
    int tiles_n;
};

// ------------------------------------------------------------------ forward / dX
template <int K, int S, int P, bool BWD, int BN>
__global__ void __launch_bounds__(256) k_convbig(CbP p) {
    constexpr int BM = 128, MT = BM / 64, NT = BN / 64, KK = K * K;
    constexpr int PA = BM * BK / 1024, PB = BN * BK / 1024;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *sA = lds, *sB = lds + 2 * BM * BK;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1, h = lane >> 5, l31 = lane & 31;
    const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Cin = p.Cin, Cout = p.Cout;
    const long npix = (long)p.N * p.Hy * p.Wy;
    const int nst = KK * (Cin / BK);

    // this thread's A rows (pixels) and 16-byte chunk within the stage's 32 channels
    int py[PA], px[PA]; long pbase[PA]; bool pok[PA];
    const int aq = tid % CH;
#pragma unroll
    for (int pp = 0; pp < PA; pp++) {
        const long m = m0 + pp * (256 / CH) + tid / CH;
        pok[pp] = m < npix;
        const long mc = pok[pp] ? m : 0;
        px[pp] = (int)(mc % p.Wy); const long t = mc / p.Wy; py[pp] = (int)(t % p.Hy);
        pbase[pp] = (t / p.Hy) * (long)p.Hx * p.Wx;              // image base (pixels)
    }
    v4f ra[PA], rb[PB];
    auto load_tiles = [&](int kt) __attribute__((always_inline)) {
        const int tap = kt / (Cin / BK), c0 = (kt - tap * (Cin / BK)) * BK;
        const int ky = tap / K, kx = tap - ky * K;
#pragma unroll
        for (int pp = 0; pp < PA; pp++) {
            int gi, gj; bool ok;
            if (!BWD) { gi = py[pp] * S + ky - P; gj = px[pp] * S + kx - P; ok = gi >= 0 && gi < p.Hx && gj >= 0 && gj < p.Wx; }
            else { const int ti = py[pp] + P - ky, tj = px[pp] + P - kx; gi = ti / S; gj = tj / S;
                   ok = ti >= 0 && tj >= 0 && (ti % S) == 0 && (tj % S) == 0 && gi < p.Hx && gj < p.Wx; }
            ok = ok && pok[pp];
            const v4f z = {0.f, 0.f, 0.f, 0.f};
            ra[pp] = ok ? *reinterpret_cast<const v4f *>(p.X + (pbase[pp] + (long)gi * p.Wx + gj) * Cin + c0 + aq * 4) : z;
        }
#pragma unroll
        for (int pp = 0; pp < PB; pp++) {
            const int id = pp * 256 + tid;
            const v4f z = {0.f, 0.f, 0.f, 0.f};
            if (!BWD) {                                          // B[k = ci][n = co] = F[ci][tap][co]: n-contiguous
                const int kk = id / (BN / 4), rq = id % (BN / 4), n = n0 + rq * 4;
                rb[pp] = (n < Cout) ? *reinterpret_cast<const v4f *>(p.F + ((long)(c0 + kk) * KK + tap) * p.C0f + n) : z;
            } else {                                             // B[n = c1][k = c0] = F[c1][KK-1-tap][c0]: k-contiguous
                const int r = id / CH, q = id % CH, n = n0 + r;
                rb[pp] = (n < Cout) ? *reinterpret_cast<const v4f *>(p.F + ((long)n * KK + (KK - 1 - tap)) * p.C0f + c0 + q * 4) : z;
            }
        }
    };
    int soa[PA], sob[PB];
#pragma unroll
    for (int pp = 0; pp < PA; pp++) { const int id = pp * 256 + tid, r = id / CH, q = id % CH; soa[pp] = r * BK + ((q ^ ((r / SW) & (CH - 1))) << 2); }
#pragma unroll
    for (int pp = 0; pp < PB; pp++) {
        const int id = pp * 256 + tid;
        if (BWD) { const int r = id / CH, q = id % CH; sob[pp] = r * BK + ((q ^ ((r / SW) & (CH - 1))) << 2); }
        else     sob[pp] = (id / (BN / 4)) * BN + (id % (BN / 4)) * 4;
    }
    auto store_tiles = [&](int buf) __attribute__((always_inline)) {
        float *a = sA + buf * BM * BK, *b = sB + buf * BN * BK;
#pragma unroll
        for (int pp = 0; pp < PA; pp++) *reinterpret_cast<v4f *>(a + soa[pp]) = ra[pp];
#pragma unroll
        for (int pp = 0; pp < PB; pp++) *reinterpret_cast<v4f *>(b + sob[pp]) = rb[pp];
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    load_tiles(0); store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nst; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nst) load_tiles(kt + 1);                     // in flight during the MFMAs below
        const float *a = sA + buf * BM * BK, *b = sB + buf * BN * BK;
#pragma unroll
        for (int ci = 0; ci < NC; ci++) {                         // lane half h holds k = 8*ci + 4*h + {0..3}
            float av[MT][4], bv[NT][4];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int r = wm * (BM / 2) + mt * 32 + l31;
                const v4f t = *reinterpret_cast<const v4f *>(a + r * BK + (((ci * 2 + h) ^ ((r / SW) & (CH - 1))) << 2));
                av[mt][0] = t[0]; av[mt][1] = t[1]; av[mt][2] = t[2]; av[mt][3] = t[3];
            }
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const int r = wn * (BN / 2) + nt * 32 + l31;
                if (BWD) { const v4f t = *reinterpret_cast<const v4f *>(b + r * BK + (((ci * 2 + h) ^ ((r / SW) & (CH - 1))) << 2));
                           bv[nt][0] = t[0]; bv[nt][1] = t[1]; bv[nt][2] = t[2]; bv[nt][3] = t[3]; }
                else {
#pragma unroll
                    for (int j = 0; j < 4; j++) bv[nt][j] = b[(ci * 8 + 4 * h + j) * BN + r];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt][j], bv[nt][j], acc[mt][nt], 0, 0, 0);
        }
        if (kt + 1 < nst) store_tiles(buf ^ 1);
        __syncthreads();
    }
    // epilogue: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * h
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int gn = n0 + wn * (BN / 2) + nt * 32 + l31;
            if (gn >= Cout) continue;
            const float bias = (!BWD && p.B) ? p.B[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long gm = m0 + wm * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (gm < npix) { const float v = acc[mt][nt][r] + bias; p.Y[gm * Cout + gn] = v; if (p.Y2) p.Y2[gm * Cout + gn] = v; }
            }
        }
}

// ------------------------------------------------------------------ forward / dX on the dense GEMM's lean pipeline (round 4)
// The same implicit GEMM as k_convbig on the structure of gemm.hip k_gemm_plain128: 8 waves = 2 k-groups x 2x2 waves, a wave owns a
// 64 x (32 NTW) block (2 x NTW accumulators of 32x32: one A fragment feeds NTW MFMAs, one B fragment two), stages of 64 channels of ONE
// tap moved by global_load_lds_dwordx4 (no VGPR round trip, no ds_write; k_convbig stages through registers, 32 channels at a time, on 4
// waves: 58-64 % of the MFMA peak), operand reads of chunk c + 1 issued before the MFMAs of chunk c.  What the conv adds is a lane's source
// address: row r of the A stage is the gathered pixel under the stage's tap - each lane keeps (y, x) of its four rows - and rows outside the
// image / past the tensor, columns past Cout, read a page of zeros (State::d_zero): no zero fill, no predicates in the MFMA loop.
// Stride 1 only (gather and output grid are then the same); Cin % 64 == 0.
struct Cb8 {
    const float *X, *F, *B;            // gathered tensor (forward: input; dX: dO), filter [C1][K][K][C0], bias (forward)
    float *Y, *Y2;
    float *part;                       // forward, every tile interior: per-channel sums of the output (sum y, sum y^2) per 64-row block, [2 tiles_m][2][Cout] - the statistics of a batch-norm layer behind this one
    int N, H, W, Cin, Cout, C0f;
    int tiles_n;
};
template <int K, int P, bool BWD, int NTW, bool NT_ST, int BK_>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(BK_ == 32 ? 4 : 2, BK_ == 32 ? 4 : 2))) k_convbig8(Cb8 p) {
    constexpr int BM = 128, BN = 64 * NTW, BK = BK_, KK = K * K;
    constexpr int NC = BK / 8, CH = BK / 4, NCG = NC / 2, RPI = 64 / CH;   // 8-deep chunks, 16-byte quads per row, chunks per k-group, rows per 1-KiB DMA instruction
    constexpr int STAGE = (BM + BN) * BK;
    constexpr int NJA = BM * BK / 2048, NJB = BN * BK / 2048;              // 1-KiB DMA instructions per wave per stage
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = w >> 2, w4 = w & 3, wm = w4 >> 1, wn = w4 & 1, h = lane >> 5, l31 = lane & 31;
    const int c0 = kg * NCG;
    const int Cin = p.Cin, Cout = p.Cout, H = p.H, W = p.W;
    const long npix = (long)p.N * H * W;
    const int tiles_m = (int)((npix + BM - 1) / BM), tiles_n = p.tiles_n, T = tiles_m * tiles_n;
    int tm, tn;
    {
        int L;
        { const int b = blockIdx.x, q8 = T >> 3, r8 = T & 7, x = b & 7, i = b >> 3;
          L = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i; }           // XCD x owns a contiguous run of the tile order
        tm = L / tiles_n; tn = L - tm * tiles_n;                                       // the N tiles of one pixel block side by side: they share the gathered rows
    }
    const long m0 = (long)tm * BM; const int n0 = tn * BN;
    const int nst = KK * (Cin / BK);                        // stages: BK channels of one tap each
    // this lane's NJA rows (pixels) of the A stage and its 16-byte quad of the stage's channels (XOR-swizzled with the row: the DMA writes lane-linear).
    // Sources are BUFFER addresses (buffer_load_dwordx4 ... offen lds): a lane keeps the byte offset of (pixel, quad) and one validity bit per tap; a tap
    // outside the image is an offset past num_records, for which the DMA stores zeros (tools/experiments/buf_lds_probe.hip) - no zero page, no 64-bit
    // selects.  The filter rows' lane offsets never change: a stage moves the scalar offset.  Rows of one lane are RPI pixels apart: one division.
    unsigned offa[NJA], vma[NJA];
    {
        const int r0 = w * NJA * RPI + lane / CH;
        const unsigned mf = (unsigned)(m0 + r0 < npix ? m0 + r0 : 0);                  // the launcher keeps npix Cin below 2^29: 32-bit pixel arithmetic
        const unsigned t = mf / (unsigned)W;
        int px = (int)(mf - t * (unsigned)W), py = (int)(t % (unsigned)H);
#pragma unroll
        for (int j = 0; j < NJA; j++) {
            const int r = r0 + j * RPI;
            const int qa = ((lane % CH) ^ (r & (CH - 1))) * 4;
            const long m = m0 + r; const bool pok = m < npix;
            offa[j] = ((pok ? (unsigned)m : 0u) * (unsigned)Cin + (unsigned)qa) * 4u;
            unsigned xm = 0, v = 0;                                                     // taps inside the image: (rows inside) x (columns inside)
#pragma unroll
            for (int kx = 0; kx < K; kx++) if ((unsigned)(px + (BWD ? P - kx : kx - P)) < (unsigned)W) xm |= 1u << kx;
#pragma unroll
            for (int ky = 0; ky < K; ky++) if ((unsigned)(py + (BWD ? P - ky : ky - P)) < (unsigned)H) v |= xm << (ky * K);
            vma[j] = pok ? v : 0u;
            px += RPI; while (px >= W) { px -= W; py++; } while (py >= H) py -= H;
        }
    }
    // B: forward F[ci][tap][co] - n-contiguous rows of BN floats, 1024 / (4 BN) k rows per instruction; dX F[c1][KK-1-tap][c0] - k-contiguous rows
    unsigned offb[NJB];
#pragma unroll
    for (int j = 0; j < NJB; j++) {
        const int i = w * NJB + j;
        if (!BWD) { constexpr int RPB = 256 / BN, LPR = BN / 4; const int kk = i * RPB + lane / LPR, col = (lane % LPR) * 4;
                    offb[j] = n0 + col < Cout ? (unsigned)(((long)kk * KK * p.C0f + n0 + col) * 4) : 0x80000000u; }
        else      { const int r = i * RPI + lane / CH, q = ((lane % CH) ^ (r & (CH - 1))) * 4;
                    offb[j] = n0 + r < Cout ? (unsigned)(((long)(n0 + r) * KK * p.C0f + q) * 4) : 0x80000000u; }
    }
    const i32x4 srdX = buf_srd(p.X, (unsigned)(npix * Cin * 4)), srdF = buf_srd(p.F, (unsigned)((long)(BWD ? Cout : Cin) * KK * p.C0f * 4));
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    // a stage's sources are worked out (prep) one stage before its DMA is fired: the arithmetic sits under the MFMAs, and what follows the stage barrier is
    // as short as the dense GEMM's (s_mov m0 + one DMA instruction per KiB)
    int itap = 0, icb = 0;                                  // the stage the next prep() works out: its tap, its first channel
    unsigned vo[NJA], sb4 = 0;
    auto prep = [&]() __attribute__((always_inline)) {
        const int ky = itap / K, kx = itap - ky * K;
        const int dy = BWD ? P - ky : ky - P, dx = BWD ? P - kx : kx - P;
        const unsigned sh4 = (unsigned)((((dy * W + dx) * Cin) + icb) * 4), bit = 1u << itap;
#pragma unroll
        for (int j = 0; j < NJA; j++) vo[j] = (vma[j] & bit) ? offa[j] + sh4 : 0x80000000u;
        sb4 = (unsigned)((!BWD ? (icb * KK + itap) * p.C0f : (KK - 1 - itap) * p.C0f + icb) * 4);
        icb += BK; if (icb == Cin) { icb = 0; itap++; }
    };
    auto fire = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJA; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + (w * NJA + j) * 256) * 4));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(vo[j]), "s"(srdX), "s"(la) : "memory");
        }
        const unsigned sbu = __builtin_amdgcn_readfirstlane(sb4);       // wave-uniform by construction; the compiler cannot always prove it across the loop
#pragma unroll
        for (int j = 0; j < NJB; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + BM * BK + (w * NJB + j) * 256) * 4));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(offb[j]), "s"(srdF), "s"(la), "s"(sbu) : "memory");   // (s_nop 4: sbu may come fresh from a v_readfirstlane)
        }
    };
    f32x16 acc[2][NTW];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < NTW; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    const int ra_ = wm * 64 + l31, rb_ = wn * (32 * NTW) + l31;
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[2][4], float (&bv)[NTW][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int ra = ra_ + t * 32;
            const v4f v = *reinterpret_cast<const v4f *>(a + ra * BK + (((ci * 2 + h) ^ (ra & (CH - 1))) << 2));
            av[t][0] = v[0]; av[t][1] = v[1]; av[t][2] = v[2]; av[t][3] = v[3];
        }
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const int rb = rb_ + t * 32;
            if (BWD) { const v4f v = *reinterpret_cast<const v4f *>(b + rb * BK + (((ci * 2 + h) ^ (rb & (CH - 1))) << 2));
                       bv[t][0] = v[0]; bv[t][1] = v[1]; bv[t][2] = v[2]; bv[t][3] = v[3]; }
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) bv[t][j] = b[(ci * 8 + 4 * h + j) * BN + rb];
            }
        }
    };
    auto mm = [&](float (&av)[2][4], float (&bv)[NTW][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < NTW; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][j], bv[b][j], acc[a][b], 0, 0, 0);
    };
    float ca[2][4], cbv[NTW][4];
    prep(); fire(0); prep();
    float bias[NTW];
#pragma unroll
    for (int b = 0; b < NTW; b++) { const int gn = n0 + wn * (32 * NTW) + b * 32 + l31; bias[b] = (!BWD && p.B && gn < Cout) ? p.B[gn] : 0.f; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int b = 0; b < NTW; b++) asm volatile("" :: "v"(bias[b]));      // the bias has landed HERE as far as the compiler is concerned: no vmcnt(0) between the epilogue's stores
    if (nst > 1) fire(1);                                   // stage kt + 2 is requested right behind stage kt's closing barrier: its buffer is free from there on
    prep();
    if constexpr (BK == 32) {
        // two workgroups per CU = four waves per SIMD: a wave reads a chunk's fragments and multiplies them, the other three cover its LDS latency.  No second
        // fragment set: with one the 128-wide forms spill, and a compiler-counted scratch reload inside this loop waits for vmcnt(0) - for the DMA of the next stage
        int buf = 0;
        for (int kt = 0; kt < nst; kt++) {
            const float *a = lds + buf * STAGE, *b = a + BM * BK;
#pragma unroll
            for (int ci = 0; ci < NCG; ci++) {
                rd(a, b, c0 + ci, ca, cbv);
                mm(ca, cbv);
                if (ci == 0 && kt > 0) { __builtin_amdgcn_sched_barrier(0); prep(); }  // the sources of stage kt + 2 (fired behind this stage's barrier), under the MFMAs just issued;
            }                                                                          // stage 2's were worked out in front of the loop
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 2 < nst) fire(buf);
            buf ^= 1;
        }
    } else {
    rd(lds, lds + BM * BK, c0, ca, cbv);
    int buf = 0;
    for (int kt = 0; kt < nst; kt++) {
        const int b1 = buf ^ 1;
        const float *a = lds + buf * STAGE, *b = a + BM * BK;
#pragma unroll
        for (int ci = 0; ci + 1 < NCG; ci++) {
            float na[2][4], nbv[NTW][4];
            rd(a, b, c0 + ci + 1, na, nbv);
            __builtin_amdgcn_sched_barrier(0);
            mm(ca, cbv);
            __builtin_amdgcn_sched_barrier(0);

#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int t = 0; t < 2; t++) ca[t][j] = na[t][j];
#pragma unroll
                for (int t = 0; t < NTW; t++) cbv[t][j] = nbv[t][j];
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nst) fire(buf);
        float na[2][4], nbv[NTW][4];
        if (kt + 1 < nst) rd(lds + b1 * STAGE, lds + b1 * STAGE + BM * BK, c0, na, nbv);
        __builtin_amdgcn_sched_barrier(0);
        mm(ca, cbv);
        __builtin_amdgcn_sched_barrier(0);
        prep();                                             // stage kt + 3, under the MFMAs just issued
        if (kt + 1 < nst) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int t = 0; t < 2; t++) ca[t][j] = na[t][j];
#pragma unroll
                for (int t = 0; t < NTW; t++) cbv[t][j] = nbv[t][j];
            }
        }
        buf = b1;
    }
    }
    // the two k-groups meet in LDS, group 0 stores (bias: forward; fetched before the main loop).  Streaming stores: the tile is not read again by this launch
    // and a layer tensor of these sizes does not stay in the L2s for the next one (with plain stores the dirty lines of all tiles are written back at the END of
    // the kernel: ~5 us of tail on a 16 MB output, tools/experiments/conv_stage_fit.py)
    __syncthreads();
    if (kg == 1) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < NTW; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) lds[((w4 * 2 * NTW + a * NTW + b) * 16 + r) * 64 + lane] = acc[a][b][r];
    }
    __syncthreads();
    if (kg == 1) return;
    // interior tiles (all of them when the pixel and channel counts are whole tiles): no predicates, 32-bit offsets from the tile's corner, the sixteen values of
    // a block read from LDS together and stored back to back (a predicated store per element costs an LDS round trip and a branch each: ~4 us of tail)
    const bool inner = m0 + BM <= npix && n0 + BN <= Cout;
    if (inner) {
        float *y0 = p.Y + (m0 * Cout + n0), *y2 = p.Y2 ? p.Y2 + (m0 * Cout + n0) : nullptr;
        float cs[NTW], cq[NTW];
#pragma unroll
        for (int b = 0; b < NTW; b++) { cs[b] = 0.f; cq[b] = 0.f; }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < NTW; b++) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = (acc[a][b][r] + lds[((w4 * 2 * NTW + a * NTW + b) * 16 + r) * 64 + lane]) + bias[b];
                const int o0 = (wm * 64 + a * 32 + 4 * h) * Cout + wn * (32 * NTW) + b * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int o = o0 + ((r & 3) + 8 * (r >> 2)) * Cout;
                    if (NT_ST) __builtin_nontemporal_store(v[r], &y0[o]); else y0[o] = v[r];
                }
                if (y2) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int o = o0 + ((r & 3) + 8 * (r >> 2)) * Cout;
                        if (NT_ST) __builtin_nontemporal_store(v[r], &y2[o]); else y2[o] = v[r];
                    }
                }
                if (!BWD && p.part) {                       // the column sums ride along: a lane's 16 rows, then (below) the block's other 16 and the wave's second block
#pragma unroll
                    for (int r = 0; r < 16; r++) { cs[b] += v[r]; cq[b] = fmaf(v[r], v[r], cq[b]); }
                }
            }
        if (!BWD && p.part) {
#pragma unroll
            for (int b = 0; b < NTW; b++) {
                const float s1 = cs[b] + __shfl_xor(cs[b], 32), s2 = cq[b] + __shfl_xor(cq[b], 32);
                if (h == 0) {
                    float *pp = p.part + ((long)(tm * 2 + wm) * 2) * Cout + n0 + wn * (32 * NTW) + b * 32 + l31;
                    pp[0] = s1; pp[Cout] = s2;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < NTW; b++) {
            const int gn = n0 + wn * (32 * NTW) + b * 32 + l31;
            if (gn >= Cout) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long gm = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (gm < npix) {
                    const float v = (acc[a][b][r] + lds[((w4 * 2 * NTW + a * NTW + b) * 16 + r) * 64 + lane]) + bias[b];
                    p.Y[gm * Cout + gn] = v;
                    if (p.Y2) p.Y2[gm * Cout + gn] = v;
                }
            }
        }
}

// ------------------------------------------------------------------ dF partials
// grid = (slices, taps * ci_tiles, co_tiles); tile 64 (ci) x 64 (co); K = the slice's pixels, 32 per stage
struct CdP { const float *I, *DO; float *part; int N, H1, W1, C1, H0, W0, C0; int pix_per_slice, ci_tiles; };

template <int K, int S, int P>
__global__ void __launch_bounds__(256) k_convbig_df(CdP p) {
    constexpr int BM = 64, BN = 64, KK = K * K;
    constexpr int PA = BM * BK / 1024, PB = BN * BK / 1024;     // 2, 2
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * BK];
    float *sA = lds, *sB = lds + 2 * BM * BK;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1, h = lane >> 5, l31 = lane & 31;
    const int tap = blockIdx.y / p.ci_tiles, cit = blockIdx.y - tap * p.ci_tiles;
    const int ky = tap / K, kx = tap - ky * K;
    const int m0 = cit * BM, n0 = blockIdx.z * BN;               // ci0, co0
    const long npix = (long)p.N * p.H0 * p.W0;
    const long k_beg = (long)blockIdx.x * p.pix_per_slice, k_end = min(npix, k_beg + p.pix_per_slice);
    const int nst = (int)((k_end - k_beg + BK - 1) / BK);
    v4f ra[PA], rb[PB];
    // this thread's pixel rows of the CURRENT stage, kept as (n, y, x) and advanced by 32 pixels per stage - no 64-bit
    // divisions in the loop (they cost more than the MFMAs they feed)
    int cx[PA], cy[PA]; long cn[PA];
#pragma unroll
    for (int pp = 0; pp < PA; pp++) {
        const long pix = k_beg + (pp * 256 + tid) / (BM / 4);
        cx[pp] = (int)(pix % p.W0); const long t = pix / p.W0; cy[pp] = (int)(t % p.H0); cn[pp] = t / p.H0;
    }
    int kt_loaded = 0;                                            // stage the coordinates above belong to
    auto load_tiles = [&](int kt) __attribute__((always_inline)) {
        const long k0 = k_beg + (long)kt * BK;
        const v4f z = {0.f, 0.f, 0.f, 0.f};
        while (kt_loaded < kt) {                                  // advance by one stage (32 pixels)
#pragma unroll
            for (int pp = 0; pp < PA; pp++) {
                cx[pp] += BK;
                while (cx[pp] >= p.W0) { cx[pp] -= p.W0; if (++cy[pp] == p.H0) { cy[pp] = 0; cn[pp]++; } }
            }
            kt_loaded++;
        }
#pragma unroll
        for (int pp = 0; pp < PA; pp++) {                         // A[k = pixel][m = ci] = I[pixel (+) tap][ci]: m-contiguous
            const int id = pp * 256 + tid, kk = id / (BM / 4), rq = id % (BM / 4);
            const long pix = k0 + kk; const int m = m0 + rq * 4;
            const int gi = cy[pp] * S + ky - P, gj = cx[pp] * S + kx - P;
            const bool ok = pix < k_end && m < p.C1 && gi >= 0 && gi < p.H1 && gj >= 0 && gj < p.W1;
            ra[pp] = ok ? *reinterpret_cast<const v4f *>(p.I + ((cn[pp] * p.H1 + gi) * (long)p.W1 + gj) * p.C1 + m) : z;
        }
#pragma unroll
        for (int pp = 0; pp < PB; pp++) {                         // B[k = pixel][n = co] = dO[pixel][co]: n-contiguous
            const int id = pp * 256 + tid, kk = id / (BN / 4), rq = id % (BN / 4);
            const long pix = k0 + kk; const int n = n0 + rq * 4;
            rb[pp] = (pix < k_end && n < p.C0) ? *reinterpret_cast<const v4f *>(p.DO + pix * p.C0 + n) : z;
        }
    };
    int soa[PA], sob[PB];
#pragma unroll
    for (int pp = 0; pp < PA; pp++) { const int id = pp * 256 + tid; soa[pp] = (id / (BM / 4)) * BM + (id % (BM / 4)) * 4; }
#pragma unroll
    for (int pp = 0; pp < PB; pp++) { const int id = pp * 256 + tid; sob[pp] = (id / (BN / 4)) * BN + (id % (BN / 4)) * 4; }
    auto store_tiles = [&](int buf) __attribute__((always_inline)) {
        float *a = sA + buf * BM * BK, *b = sB + buf * BN * BK;
#pragma unroll
        for (int pp = 0; pp < PA; pp++) *reinterpret_cast<v4f *>(a + soa[pp]) = ra[pp];
#pragma unroll
        for (int pp = 0; pp < PB; pp++) *reinterpret_cast<v4f *>(b + sob[pp]) = rb[pp];
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if (nst > 0) { load_tiles(0); store_tiles(0); }
    __syncthreads();
    for (int kt = 0; kt < nst; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nst) load_tiles(kt + 1);
        const float *a = sA + buf * BM * BK, *b = sB + buf * BN * BK;
        const int ra_ = wm * 32 + l31, rb_ = wn * 32 + l31;
#pragma unroll
        for (int ci = 0; ci < NC; ci++) {
            float av[4], bv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { av[j] = a[(ci * 8 + 4 * h + j) * BM + ra_]; bv[j] = b[(ci * 8 + 4 * h + j) * BN + rb_]; }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], acc1, 0, 0, 0);
        }
        if (kt + 1 < nst) store_tiles(buf ^ 1);
        __syncthreads();
    }
    // partial slab [slice][row = (ci*K+ky)*K+kx][co]  (the fold's layout, without a bias row)
    const int co = n0 + wn * 32 + l31;
    if (co < p.C0) {
        const long nrow = (long)p.C1 * KK;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ci = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ci < p.C1) p.part[((long)blockIdx.x * nrow + ((long)ci * KK + tap)) * p.C0 + co] = acc0[r] + acc1[r];
        }
    }
}

// ------------------------------------------------------------------ dF partials, 8-wave LDS-DMA pipeline (round 3)
// The same split-K GEMM (one tap x 64 ci x 64 co per workgroup, K = the slice's pixels) on the dense GEMM's lean pipeline
// (gemm.hip k_gemm_nn_plain, both operands k-major - its fastest layout): 8 waves = 2 k-groups x 2x2 waves of 32x32 accumulator pairs,
// stages of BKP pixels moved by global_load_lds_dwordx4 (no VGPR round trip, no ds_write), operand reads of chunk c+1 issued before the
// MFMAs of chunk c, the next stage's first reads before the last MFMAs of this one.  What the conv adds is the address of a lane's
// 16 bytes: row k of the A stage is the input pixel under tap (ky, kx) of output pixel k - each lane keeps the (n, y, x) of its rows and
// advances them by BKP pixels per stage without divisions - and rows outside the image / past the slice / channel groups past C1, C0
// read from a 4 KiB page of zeros (State::d_zero), so the LDS stage needs no zero fill and the MFMA loop no predicates.
struct Cd8 { const float *I, *DO, *Z; float *part; int N, H1, W1, C1, H0, W0, C0; int pix_per_slice, ci_tiles; long npix; int dbg; int nslice, ctiles; int tp2; };

template <int K, int S, int P, int BKP, int NST = 2>     // NST stage buffers: the DMA runs NST - 1 stages ahead of the MFMAs
__global__ void __launch_bounds__(512) k_convbig_df8(Cd8 p) {
    constexpr int BM = 64, BN = 64, KK = K * K;
    constexpr int NCH = BKP / 8, NCG = NCH / 2;            // 8-deep chunks per stage, per k-group
    constexpr int STAGE = (BM + BN) * BKP;                 // floats per stage buffer
    constexpr int NJ = BKP / 32;                           // DMA instructions (4 k rows x 256 B) per operand per wave per stage
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = w >> 2, w4 = w & 3, wm = w4 >> 1, wn = w4 & 1, h = lane >> 5, l31 = lane & 31;
    const int c0 = kg * NCG;
    // 1-D grid, XCD-aware: workgroup id % 8 is the XCD it runs on.  The K*K taps of one (pixel slice, ci tile, co tile) group re-read the same
    // dO rows and overlapping input rows, so a group's workgroups are consecutive ON ONE XCD (its L2 serves 8 of the 9 reads) and the groups
    // are dealt round-robin to the XCDs - any slice count balances, not only multiples of 8.
    const int wg = blockIdx.x, xcd = wg & 7, jx = wg >> 3;
    // tp2 (C1 == 32): a 64-row tile would be half empty, so it carries TWO taps - rows 0..31 the channels under tap 2 ts, rows 32..63 under
    // tap 2 ts + 1 (the tap is then a per-lane value: DMA lanes 0..7 of a row fetch the first tap's channel groups, lanes 8..15 the second's)
    const int kks = p.tp2 ? (KK + 1) / 2 : KK;
    const int grp = xcd + 8 * (jx / kks), ts = jx % kks;
    const int slice = grp / p.ctiles, tl = grp - slice * p.ctiles;
    if (slice >= p.nslice) return;
    const int cit = tl % p.ci_tiles, cot = tl / p.ci_tiles;
    const int ch = lane & 15;
    const int tap = p.tp2 ? 2 * ts + (ch >> 3) : ts, ach = p.tp2 ? (ch & 7) : ch;     // this lane's tap and 16-byte channel group of the input row
    const int ky = tap / K, kx = tap - ky * K;
    const int m0 = cit * BM, n0 = cot * BN;                 // ci0, co0
    const long k_beg = (long)slice * p.pix_per_slice, k_end = min(p.npix, k_beg + p.pix_per_slice);
    const int nst = k_end > k_beg ? (int)((k_end - k_beg + BKP - 1) / BKP) : 0;

    // this lane's rows of a stage: kk = (w * NJ + j) * 4 + lane / 16, its 16-byte channel group ch = lane % 16.
    // The address work per stage is kept small (it competes with the MFMAs for the SIMD's issue slot: 100 VALU instructions per stage
    // cost ~10 %): both row pointers ADVANCE by a constant per stage - dO is linear in the pixel index, and so is the input under a
    // stride-1 same-size tap (LIN: address = (pixel + (ky-P) W + kx-P) C1, only its validity needs (y, x)) - a stride-2 tap recomputes.
    constexpr bool LIN = S == 1;
    const bool a_col = p.tp2 ? tap < KK : m0 + ch * 4 < p.C1, b_col = n0 + ch * 4 < p.C0;
    int cx[NJ], cy[NJ], cn[NJ], left[NJ];
    int oa[NJ], ob[NJ];                                     // row offsets from I / dO in 16-byte units (64-bit pointer arrays ended up in scratch)
    const v4f *zsrc = reinterpret_cast<const v4f *>(p.Z) + ch;
    const v4f *I4 = reinterpret_cast<const v4f *>(p.I), *DO4 = reinterpret_cast<const v4f *>(p.DO);
    const int C14 = p.C1 >> 2, C04 = p.C0 >> 2;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const long pix = k_beg + (w * NJ + j) * 4 + (lane >> 4);
        left[j] = (int)min((long)(1 << 30), k_end - pix);          // > 0: the row lies inside the slice
        cx[j] = (int)(pix % p.W0); const long t = pix / p.W0; cy[j] = (int)(t % p.H0); cn[j] = (int)(t / p.H0);
        ob[j] = (int)(pix * C04 + (n0 >> 2) + ch);
        oa[j] = LIN ? (int)((pix + (long)(ky - P) * p.W1 + (kx - P)) * C14 + (m0 >> 2) + ach) : (m0 >> 2) + ach;
    }
    const int dxs = BKP % p.W0, dys = BKP / p.W0, dyr = dys % p.H0, dns = dys / p.H0;
    const int stepA = (p.dbg & 1) ? 0 : BKP * C14, stepB = (p.dbg & 1) ? 0 : BKP * C04;   // dbg bit 0 (lab only, T4K_CONVBIG_DF8_DBG): re-read the first stage's rows - the kernel without its HBM stream
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    // issue the DMA of the stage the row state stands at, then advance it one stage
    auto issue = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int gi = cy[j] * S + ky - P, gj = cx[j] * S + kx - P;
            const bool in = left[j] > 0;
            const bool oka = in && a_col && (unsigned)gi < (unsigned)p.H1 && (unsigned)gj < (unsigned)p.W1;
            const v4f *qa = I4 + (LIN ? (long)oa[j] : (long)oa[j] + (((long)cn[j] * p.H1 + gi) * p.W1 + gj) * C14);
            const v4f *sa = oka ? qa : zsrc;
            const v4f *sb = (in && b_col) ? DO4 + (long)ob[j] : zsrc;
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + (w * NJ + j) * 256) * 4));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(sa), "s"(la) : "memory");
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(sb), "s"(la + BM * BKP * 4) : "memory");
            left[j] -= BKP; ob[j] += stepB; if (LIN) oa[j] += stepA;
            cx[j] += dxs; const int c1 = cx[j] >= p.W0 ? 1 : 0; cx[j] -= c1 ? p.W0 : 0;
            cy[j] += dyr + c1; const int c2 = cy[j] >= p.H0 ? 1 : 0; cy[j] -= c2 ? p.H0 : 0;
            if (!LIN) cn[j] += dns + c2;
        }
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int ra_ = wm * 32 + l31, rb_ = wn * 32 + l31;
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; j++) { av[j] = a[(ci * 8 + 4 * h + j) * BM + ra_]; bv[j] = b[(ci * 8 + 4 * h + j) * BN + rb_]; }
    };
    auto mm = [&](float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], acc1, 0, 0, 0);
    };
    // next stage landed (this wave's share; the barrier makes it everybody's), up to NST - 2 later ones may still fly: each stage is
    // exactly 2 * NJ DMA instructions of this wave
    auto wait_next = [&](int flying) __attribute__((always_inline)) {
        constexpr int NPW = 2 * NJ;
        if (NST >= 4 && flying >= 2) { if (NPW == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); else if (NPW == 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); }
        else if (NST >= 3 && flying >= 1) { if (NPW == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); else if (NPW == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    constexpr int AHEAD = NST >= 4 ? 3 : NST - 1;          // stages in flight beyond the one being multiplied (vmcnt immediates cover up to 2 flying after the wait)
    float ca[4], cb[4];
    if (nst > 0) {
#pragma unroll
        for (int a = 0; a < AHEAD; a++) if (a < nst) issue(a);
        wait_next(min(nst, AHEAD) - 1);
        rd(lds, lds + BM * BKP, c0, ca, cb);
    }
    int buf = 0;
    for (int kt = 0; kt < nst; kt++) {
        int b1 = buf + 1; if (b1 >= NST) b1 -= NST;
        int bi = buf + AHEAD; if (bi >= NST) bi -= NST;      // the buffer read in stage kt - 1 when AHEAD == NST - 1; a free one otherwise
        if (kt + AHEAD < nst) issue(bi);
        const float *a = lds + buf * STAGE, *b = a + BM * BKP;
#pragma unroll
        for (int ci = 0; ci + 1 < NCG; ci++) {
            float na[4], nbv[4];
            rd(a, b, c0 + ci + 1, na, nbv);
            __builtin_amdgcn_sched_barrier(0);
            mm(ca, cb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; j++) { ca[j] = na[j]; cb[j] = nbv[j]; }
        }
        wait_next(min(nst - kt - 1, AHEAD) - 1);              // stages issued and not yet awaited, minus the one needed now
        float na[4], nbv[4];
        if (kt + 1 < nst) rd(lds + b1 * STAGE, lds + b1 * STAGE + BM * BKP, c0, na, nbv);
        __builtin_amdgcn_sched_barrier(0);
        mm(ca, cb);
        if (kt + 1 < nst) {
#pragma unroll
            for (int j = 0; j < 4; j++) { ca[j] = na[j]; cb[j] = nbv[j]; }
        }
        buf = b1;
    }
    // the two k-groups meet in LDS (the stage buffers are free now), group 0 writes the slab rows
    __syncthreads();
    if (kg == 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) lds[(w4 * 16 + r) * 64 + lane] = acc0[r] + acc1[r];
    }
    __syncthreads();
    if (kg == 1) return;
    // partial slab [slice][row = (ci*K+ky)*K+kx][co]  (the fold's layout, without a bias row)
    const int co = n0 + wn * 32 + l31;
    const long nrow = (long)p.C1 * KK;
    // a wave's 32 rows are one tap (also with two taps per tile): whole blocks take a branch-free path - the 16 values read from LDS together, 32-bit offsets,
    // stores back to back (a predicated block per element costs an LDS round trip, a branch and 64-bit address arithmetic each)
    const int tpw = p.tp2 ? 2 * ts + wm : ts, ci0 = p.tp2 ? 0 : m0 + wm * 32;
    if (tpw < KK && ci0 + 32 <= p.C1 && n0 + wn * 32 + 32 <= p.C0) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = (acc0[r] + acc1[r]) + lds[(w4 * 16 + r) * 64 + lane];
        float *base = p.part + ((long)slice * nrow + ((long)ci0 * KK + tpw)) * p.C0 + co;
        const int rs = KK * p.C0;
#pragma unroll
        for (int r = 0; r < 16; r++) base[((r & 3) + 8 * (r >> 2) + 4 * h) * rs] = v[r];
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ci = p.tp2 ? (m & 31) : m0 + m, tp = p.tp2 ? 2 * ts + (m >> 5) : ts;
        const float v = (acc0[r] + acc1[r]) + lds[(w4 * 16 + r) * 64 + lane];
        if (co < p.C0 && ci < p.C1 && tp < KK) p.part[((long)slice * nrow + ((long)ci * KK + tp)) * p.C0 + co] = v;
    }
}


// ------------------------------------------------------------------ dF partials on a 128-row tile (round 5)
// k_convbig_df8's 64 x 64 tile moves 32 KB per 2 x 64 x 64 x 64 flop stage - 16 bytes per clock for the two workgroups of a CU, which is what a CU's fetch
// path delivers: the kernel sat at 66 % of the MFMA peak however its stages were scheduled.  Here a tile is 128 rows x (64 NTW) columns of dF: the rows are
// 128 / CIW taps x CIW input channels (CIW = 128: one tap), the columns output channels; both operands are k-major with k = the slice's pixels
// (A[k][m] = the input under the row's tap, B[k][n] = dO), the dense GEMM's TN layout, and a wave owns 64 x (32 NTW) of it (one A fragment feeds NTW MFMAs,
// one B fragment two): half the bytes per flop.  Stride 1, same-size layers (the gather is then linear in the pixel index: a row's offset advances by a
// constant per stage, only its validity needs (y, x)); sources are buffer offsets, rows outside the image / the slice and taps past K K are out-of-range
// offsets (the DMA stores zeros), worked out one stage ahead of their DMA.  Slab layout and fold as k_convbig_df8.
struct Cdw { const float *I, *DO; float *part; int H, W, C1, C0; int pix_per_slice, nslice, ci_tiles, ctiles, kks; long npix; };
template <int K, int P, int CIW, int NTW, int BKP>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(BKP == 32 ? 4 : 2, BKP == 32 ? 4 : 2))) k_convbig_dfw(Cdw p) {
    constexpr int BM = 128, BN = 64 * NTW, KK = K * K, TPT = BM / CIW;
    constexpr int NC = BKP / 8, NCG = NC / 2;
    constexpr int STAGE = (BM + BN) * BKP;
    constexpr int NJA = BKP / 16, NJB = BKP * BN / 2048;    // 1-KiB DMA instructions per wave per stage: A 2 pixels x 128 rows each, B 256 / BN pixels x BN columns
    constexpr int LPR = BN / 4, RPB = 64 / LPR;             // B: lanes per pixel row, pixel rows per instruction
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = w >> 2, w4 = w & 3, wm = w4 >> 1, wn = w4 & 1, h = lane >> 5, l31 = lane & 31;
    const int c0 = kg * NCG;
    // one workgroup per tile, at most one (BKP 64) or two (BKP 32) per CU: XCD x (= workgroup id % 8) owns a contiguous run of the tile order
    // (slice, co tile, ci tile, tap group), so the taps of one pixel slice - same dO rows, overlapping input rows - meet in one L2 and every XCD gets the
    // same number of tiles (dealing whole 9-tap groups to the XCDs left four of them with 36 tiles for 32 CUs: a second round, 171 instead of 85 us)
    const int T = p.nslice * p.ctiles * p.kks;
    int L;
    { const int b = blockIdx.x, q8 = T >> 3, r8 = T & 7, x = b & 7, i = b >> 3;
      if (i >= q8 + (x < r8 ? 1 : 0)) return;
      L = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i; }
    const int slice = L / (p.ctiles * p.kks), rem = L - slice * (p.ctiles * p.kks), tl = rem / p.kks, ts = rem - tl * p.kks;
    const int cit = tl % p.ci_tiles, cot = tl / p.ci_tiles;
    const int n0 = cot * BN, H = p.H, W = p.W, C1 = p.C1, C0 = p.C0;
    const long k_beg = (long)slice * p.pix_per_slice, k_end = min(p.npix, k_beg + p.pix_per_slice);
    const int nst = k_end > k_beg ? (int)((k_end - k_beg + BKP - 1) / BKP) : 0;
    // A: this lane's four rows m .. m + 3 of the tile = one tap, four adjacent input channels; its pixels (k rows) 2 i + lane / 32 of the stage
    const int ma = (lane & 31) * 4, tap = ts * TPT + ma / CIW, cia = cit * CIW + ma % CIW;
    const int ky = tap / K, kx = tap - ky * K, dy = ky - P, dx = kx - P;
    const bool a_ok = tap < KK && cia < C1;
    unsigned offa[NJA]; int cx[NJA], cy[NJA], left[NJA];
#pragma unroll
    for (int j = 0; j < NJA; j++) {
        const long pix = k_beg + (w * NJA + j) * 2 + (lane >> 5);
        left[j] = (int)min((long)(1 << 30), k_end - pix);            // > 0: the row lies inside the slice
        const unsigned pu = (unsigned)pix, t = pu / (unsigned)W;
        cx[j] = (int)(pu - t * (unsigned)W); cy[j] = (int)(t % (unsigned)H);
        offa[j] = (unsigned)(((pix + dy * W + dx) * C1 + cia) * 4);
    }
    const int nb = (lane % LPR) * 4;
    const bool b_ok = n0 + nb < C0;
    unsigned offb[NJB]; int leftb[NJB];
#pragma unroll
    for (int j = 0; j < NJB; j++) {
        const long pix = k_beg + (w * NJB + j) * RPB + lane / LPR;
        leftb[j] = (int)min((long)(1 << 30), k_end - pix);
        offb[j] = (unsigned)((pix * C0 + n0 + nb) * 4);
    }
    const int dxs = BKP % W, dyr = (BKP / W) % H;
    const unsigned stepA = (unsigned)(BKP * C1 * 4), stepB = (unsigned)(BKP * C0 * 4);
    const i32x4 srdI = buf_srd(p.I, (unsigned)(p.npix * C1 * 4)), srdO = buf_srd(p.DO, (unsigned)(p.npix * C0 * 4));
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    unsigned voa[NJA], vob[NJB];
    auto prep = [&]() __attribute__((always_inline)) {      // the sources of the stage the row state stands at; the state moves on one stage
#pragma unroll
        for (int j = 0; j < NJA; j++) {
            const bool ok = a_ok && left[j] > 0 && (unsigned)(cy[j] + dy) < (unsigned)H && (unsigned)(cx[j] + dx) < (unsigned)W;
            voa[j] = ok ? offa[j] : 0x80000000u;
            offa[j] += stepA; left[j] -= BKP;
            cx[j] += dxs; const int c1 = cx[j] >= W ? 1 : 0; cx[j] -= c1 ? W : 0;
            cy[j] += dyr + c1; cy[j] -= cy[j] >= H ? H : 0;
        }
#pragma unroll
        for (int j = 0; j < NJB; j++) { vob[j] = (b_ok && leftb[j] > 0) ? offb[j] : 0x80000000u; offb[j] += stepB; leftb[j] -= BKP; }
    };
    auto fire = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJA; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + (w * NJA + j) * 256) * 4));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voa[j]), "s"(srdI), "s"(la) : "memory");
        }
#pragma unroll
        for (int j = 0; j < NJB; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + BM * BKP + (w * NJB + j) * 256) * 4));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(vob[j]), "s"(srdO), "s"(la) : "memory");
        }
    };
    f32x16 acc[2][NTW];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < NTW; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    const int ra_ = wm * 64 + l31, rb_ = wn * (32 * NTW) + l31;
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[2][4], float (&bv)[NTW][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) av[t][j] = a[(ci * 8 + 4 * h + j) * BM + ra_ + t * 32];
#pragma unroll
        for (int t = 0; t < NTW; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) bv[t][j] = b[(ci * 8 + 4 * h + j) * BN + rb_ + t * 32];
    };
    auto mm = [&](float (&av)[2][4], float (&bv)[NTW][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < NTW; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][j], bv[b][j], acc[a][b], 0, 0, 0);
    };
    float ca[2][4], cbv[NTW][4];
    if constexpr (BKP == 32) {                              // two workgroups per CU = four waves per SIMD: read a chunk's fragments, multiply; no second fragment set (no scratch)
        if (nst > 0) {
            prep(); fire(0); prep();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (nst > 1) fire(1);
            prep();
        }
        int buf = 0;
        for (int kt = 0; kt < nst; kt++) {
            const float *a = lds + buf * STAGE, *b = a + BM * BKP;
#pragma unroll
            for (int ci = 0; ci < NCG; ci++) {
                rd(a, b, c0 + ci, ca, cbv);
                mm(ca, cbv);
                if (ci == 0 && kt > 0) { __builtin_amdgcn_sched_barrier(0); prep(); }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 2 < nst) fire(buf);
            buf ^= 1;
        }
    } else {
    if (nst > 0) {
        prep(); fire(0); prep();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (nst > 1) fire(1);
        prep();
        rd(lds, lds + BM * BKP, c0, ca, cbv);
    }
    int buf = 0;
    for (int kt = 0; kt < nst; kt++) {
        const int b1 = buf ^ 1;
        const float *a = lds + buf * STAGE, *b = a + BM * BKP;
#pragma unroll
        for (int ci = 0; ci + 1 < NCG; ci++) {
            float na[2][4], nbv[NTW][4];
            rd(a, b, c0 + ci + 1, na, nbv);
            __builtin_amdgcn_sched_barrier(0);
            mm(ca, cbv);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int t = 0; t < 2; t++) ca[t][j] = na[t][j];
#pragma unroll
                for (int t = 0; t < NTW; t++) cbv[t][j] = nbv[t][j];
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nst) fire(buf);
        float na[2][4], nbv[NTW][4];
        if (kt + 1 < nst) rd(lds + b1 * STAGE, lds + b1 * STAGE + BM * BKP, c0, na, nbv);
        __builtin_amdgcn_sched_barrier(0);
        mm(ca, cbv);
        __builtin_amdgcn_sched_barrier(0);
        prep();                                             // stage kt + 3, under the MFMAs just issued
        if (kt + 1 < nst) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int t = 0; t < 2; t++) ca[t][j] = na[t][j];
#pragma unroll
                for (int t = 0; t < NTW; t++) cbv[t][j] = nbv[t][j];
            }
        }
        buf = b1;
    }
    }
    // the two k-groups meet in LDS (the stage buffers are free now), group 0 writes the slab rows [slice][(ci K + ky) K + kx][co]
    __syncthreads();
    if (kg == 1) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < NTW; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) lds[((w4 * 2 * NTW + a * NTW + b) * 16 + r) * 64 + lane] = acc[a][b][r];
    }
    __syncthreads();
    if (kg == 1) return;
    float *slab = p.part + (long)slice * C1 * KK * C0;
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const int mb = wm * 64 + a * 32;                    // a block of 32 rows: one tap (CIW >= 32)
        const int tp = ts * TPT + mb / CIW, cib = cit * CIW + mb % CIW;
        if (tp >= KK || cib >= C1) continue;
#pragma unroll
        for (int b = 0; b < NTW; b++) {
            const int co = n0 + wn * (32 * NTW) + b * 32 + l31;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = acc[a][b][r] + lds[((w4 * 2 * NTW + a * NTW + b) * 16 + r) * 64 + lane];
            if (co < C0) {
#pragma unroll
                for (int r = 0; r < 16; r++) slab[((cib + (r & 3) + 8 * (r >> 2) + 4 * h) * KK + tp) * C0 + co] = v[r];
            }
        }
    }
}

} // namespace

namespace t4k {

bool conv_big_ok(int Cin, int Cout) { return Cin >= 32 && (Cin % 32) == 0 && Cout >= 16 && (Cout % 4) == 0; }

// forward (BWD = false) or dX (BWD = true); X/Cin are the gathered tensor, Y/Cout the produced one
template <bool BWD>
void launch_conv_big(int K, int S, int P, hipStream_t hs, const float *X, float *Y, float *Y2, const float *F, const float *B,
                     int N, int Hx, int Wx, int Cin, int Hy, int Wy, int Cout, int C0f, float *bn_part, size_t bn_part_floats, int *bn_chunks) {
    if (bn_chunks) *bn_chunks = 0;
    const long npix = (long)N * Hy * Wy;
    {   // stride 1, "same" padding, whole 64-channel stages: the 8-wave LDS-DMA kernel (k_convbig8), every such layer since round 5 (round 4 kept the 9-stage
        // layers on k_convbig: with buffer-addressed DMA, the branch-free epilogue and two workgroups per CU on 32-channel stages they gain most -
        // 64 -> 128 @ 16x16 forward 96.2 -> 83.9 us, 64 -> 64 @ 32x32 221 -> 194.5 us, dX 185 -> 168 us); T4K_CONVBIG8=0: off
        static const int on = T4K_LAB_ENV("T4K_CONVBIG8", 1);
        const bool shape = S == 1 && P == K / 2 && (K == 1 || K == 3 || K == 5) && Cin % 64 == 0 && Cout % 4 == 0 && Hx == Hy && Wx == Wy &&
                           aligned16(X) && aligned16(F) && npix >= 128 && npix * Cin < (1L << 29) && (long)(Cin > Cout ? Cin : Cout) * K * K * C0f < (1L << 29);   // byte offsets of the buffer loads are 32-bit
        if (on && shape) {
            const int tiles_m = (int)((npix + 127) / 128);
            // 128-wide tiles when they still give every CU a workgroup, 64-wide otherwise (CIFAR conv3 dX: 128 -> 256 workgroups) and for 64 output channels
            const bool wide = Cout > 64 && (long)tiles_m * ((Cout + 127) / 128) >= (long)st().cu_count;
            const int BN = wide ? 128 : 64;
            float *rider = nullptr;                         // batch-norm statistics from the epilogue: forward, every tile interior, the slab fits
            if (!BWD && bn_part && bn_chunks && npix % 128 == 0 && Cout % BN == 0 && (size_t)tiles_m * 2 * 2 * Cout <= bn_part_floats) { rider = bn_part; *bn_chunks = tiles_m * 2; }
            Cb8 q = { X, F, B, Y, Y2, rider, N, Hy, Wy, Cin, Cout, C0f, (Cout + BN - 1) / BN };
            const dim3 g8((unsigned)(tiles_m * q.tiles_n)), b8(512);
            // grids of two or more tiles per CU: 32-channel stages, half the LDS, at most 128 registers - two workgroups share a CU and one's stage barrier
            // (and prologue, and epilogue) runs under the other's MFMAs (as k_gemm_plain128<.., 32>)
            static const int bk32 = T4K_LAB_ENV("T4K_CONVBIG8_BK32", 1);
            const bool two = bk32 && (bk32 >= 2 || (long)tiles_m * q.tiles_n >= 2L * st().cu_count);
            const size_t lds8 = std::max(sizeof(float) * 2 * (128 + BN) * (two ? 32 : 64), sizeof(float) * 4 * 2 * (BN / 64) * 16 * 64);   // stages | the k-groups' meeting
#define CB8_(k, pd, ntw, nt, bk) do { static bool a1 = false; if (!a1) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_convbig8<k, pd, BWD, ntw, nt, bk>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8); a1 = true; } \
                                      T4K_LAUNCH((k_convbig8<k, pd, BWD, ntw, nt, bk>), g8, b8, lds8, hs, q); } while (0)
#define CB8n(k, pd, ntw) do { if (two) { if (nts) CB8_(k, pd, ntw, true, 32); else CB8_(k, pd, ntw, false, 32); } else { if (nts) CB8_(k, pd, ntw, true, 64); else CB8_(k, pd, ntw, false, 64); } } while (0)
#define CB8(k, pd) do { if (wide) CB8n(k, pd, 2); else CB8n(k, pd, 1); } while (0)
            static const int nts = T4K_LAB_ENV("T4K_CONVBIG8_NT", 0);
            if (K == 1) CB8(1, 0); else if (K == 3) CB8(3, 1); else CB8(5, 2);
#undef CB8n
#undef CB8_
#undef CB8
            return;
        }
    }
    CbP p = { X, F, B, Y, Y2, N, Hx, Wx, Cin, Hy, Wy, Cout, C0f, 0 };
    const int tiles_m = (int)((npix + 127) / 128);
    static const int wmul = T4K_LAB_ENV("T4K_CONVBIG_WIDE_MUL", 1);
    const bool wide = Cout > 64 && (long)tiles_m * ((Cout + 127) / 128) >= (long)st().cu_count * wmul;   // 128-wide tiles only when they still give every CU a workgroup (CIFAR conv3 dX: 128 -> 256 workgroups)
    const int BN = wide ? 128 : 64;
    p.tiles_n = (Cout + BN - 1) / BN;
    const dim3 g((unsigned)(tiles_m * p.tiles_n)), b(256);
    const size_t lds = sizeof(float) * 2 * (128 + BN) * BK;
#define CB(k, s, pd) do { if (wide) { static bool a1 = false; if (!a1) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_convbig<k, s, pd, BWD, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); a1 = true; } \
                                       T4K_LAUNCH((k_convbig<k, s, pd, BWD, 128>), g, b, lds, hs, p); } \
                          else T4K_LAUNCH((k_convbig<k, s, pd, BWD, 64>), g, b, lds, hs, p); } while (0)
    switch ((K << 8) | (S << 4) | P) {
    case 0x110: CB(1, 1, 0); break;
    case 0x311: CB(3, 1, 1); break;
    case 0x421: CB(4, 2, 1); break;
    case 0x512: CB(5, 1, 2); break;
    }
#undef CB
}
template void launch_conv_big<false>(int, int, int, hipStream_t, const float *, float *, float *, const float *, const float *, int, int, int, int, int, int, int, int, float *, size_t, int *);
template void launch_conv_big<true>(int, int, int, hipStream_t, const float *, float *, float *, const float *, const float *, int, int, int, int, int, int, int, int, float *, size_t, int *);

// dF partial slabs; returns the number of slices written (0: workspace too small).  Layout [slice][C1*K*K][C0].
int launch_conv_big_df(int K, int S, int P, hipStream_t hs, const float *I, const float *DO, float *part, size_t part_floats,
                       int N, int H1, int W1, int C1, int H0, int W0, int C0) {
    const long npix = (long)N * H0 * W0;
    const int ci_tiles = (C1 + 63) / 64, co_tiles = (C0 + 63) / 64, KK = K * K;
    const int tiles = KK * ci_tiles * co_tiles;
    static const int wpc = std::max(1, T4K_LAB_ENV("T4K_DF_WGS_PER_CU", 3));
    long nslice = ((long)wpc * st().cu_count + tiles - 1) / tiles; if (nslice < 1) nslice = 1;      // workgroups per CU in total
    long pps = (npix + nslice - 1) / nslice; pps = (pps + BK - 1) / BK * BK; if (pps < 8 * BK) pps = 8 * BK;
    nslice = (npix + pps - 1) / pps;
    // Workgroups go to XCD (linear block id % 8) and the grid is slice-major: with a slice count that is a multiple of 8 every
    // tap / channel tile of one pixel slice lands on the SAME XCD, so the K*K-fold re-read of I and dO is served by that XCD's
    // L2 instead of crossing the fabric once per tap (a trailing slice may be empty: it writes a zero slab)
    static const int x8 = T4K_LAB_ENV("T4K_DF_XCD", 1);
    if (x8 && nslice >= 8) { nslice = (nslice + 7) / 8 * 8; pps = (npix + nslice - 1) / nslice; pps = (pps + BK - 1) / BK * BK; }
    {   // stride 1, same size, whole 128s of input channels, whole 64s of output channels: 128-row tiles (k_convbig_dfw; 128 -> 256 @ 8x8, N = 256: 92.0 -> 88.4 us).
        // Two or four taps of 64 / 32 channels per tile work too (T4K_CONVBIG_DFW=3) but lose: 9 taps fill 10 / 12 tap slots and the fold reads 51 slices
        // (64 -> 128 @ 16x16: 96.3 + 24 us of fold against 89 + 12)
        static const int dfw = T4K_LAB_ENV("T4K_CONVBIG_DFW", 1);
        const bool shape = S == 1 && P == K / 2 && (K == 1 || K == 3 || K == 5) && H1 == H0 && W1 == W0 && (C1 % 128 == 0 || (dfw >= 3 && (C1 == 32 || C1 == 64))) && C0 % 64 == 0 &&
                           npix * C1 < (1L << 29) && npix * C0 < (1L << 29) && aligned16(I) && aligned16(DO);
        if (dfw && shape) {
            const int ciw = C1 >= 128 ? 128 : C1, tpt = 128 / ciw, kks = (KK + tpt - 1) / tpt;
            const int ntw = C0 % 128 == 0 ? 2 : 1, bn = 64 * ntw;
            const int cit = (C1 + ciw - 1) / ciw, cot = C0 / bn, ctl = cit * cot, tilesw = kks * ctl;
            // 1: 64-pixel stages, one workgroup per CU; 2: 32-pixel stages, two per CU (twice the slices, twice the slab bytes)
            const int bkp = (dfw == 2 || dfw == 4) ? 32 : 64;
            const long slots = (long)st().cu_count * (bkp == 32 ? 2 : 1);
            long ns = slots / tilesw; if (ns < 1) ns = 1;
            long pp = (npix + ns - 1) / ns; pp = (pp + bkp - 1) / bkp * bkp; if (pp < 4 * bkp) pp = 4 * bkp;
            ns = (npix + pp - 1) / pp;
            if ((size_t)ns * C1 * KK * C0 <= part_floats) {
                Cdw q = { I, DO, part, H0, W0, C1, C0, (int)pp, (int)ns, cit, ctl, kks, npix };
                const long T = ns * ctl * kks;
                const dim3 gw((unsigned)(8 * ((T + 7) / 8))), bw(512);
                const size_t ldsw = std::max(sizeof(float) * 2 * (128 + bn) * bkp, sizeof(float) * 4 * 2 * ntw * 16 * 64);
#define DFW_(k, pd, cw, nt, bk) do { static bool a1 = false; if (!a1) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_convbig_dfw<k, pd, cw, nt, bk>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw); a1 = true; } \
                                     T4K_LAUNCH((k_convbig_dfw<k, pd, cw, nt, bk>), gw, bw, ldsw, hs, q); } while (0)
#define DFWb(k, pd, cw, nt) do { if (bkp == 32) DFW_(k, pd, cw, nt, 32); else DFW_(k, pd, cw, nt, 64); } while (0)
#define DFWn(k, pd, cw) do { if (ntw == 2) DFWb(k, pd, cw, 2); else DFWb(k, pd, cw, 1); } while (0)
#define DFW(k, pd) do { if (ciw == 128) DFWn(k, pd, 128); else if (ciw == 64) DFWn(k, pd, 64); else DFWn(k, pd, 32); } while (0)
                if (K == 1) DFW(1, 0); else if (K == 3) DFW(3, 1); else DFW(5, 2);
#undef DFW
#undef DFWn
#undef DFWb
#undef DFW_
                return (int)ns;
            }
        }
    }
    static const int df8 = T4K_LAB_ENV("T4K_CONVBIG_DF8", 64);     // 0: the 4-wave register-staged kernel; 64 / 128: pixels per stage of the 8-wave LDS-DMA kernel
    if (df8 && npix * (C0 > C1 ? C0 : C1) / 4 + (long)4 * W1 * C1 < (1L << 31) && st().d_zero) {      // row offsets are ints in 16-byte units
        // 64-pixel stages: 64 KiB of LDS, two workgroups per CU (one's barrier under the other's MFMAs) -> up to 2 x CUs workgroups at once, all resident
        const int bkp = df8 >= 128 ? 128 : df8 >= 64 ? 64 : 32;
        static const int wpc8 = T4K_LAB_ENV("T4K_CONVBIG_DF8_WPC", 0);
        static const int nstb = T4K_LAB_ENV("T4K_CONVBIG_DF8_NST", (bkp == 32 ? 4 : 2));
        const int lds_kb = (bkp == 128 ? 2 : bkp == 64 ? (nstb == 3 ? 3 : 2) : (nstb >= 5 ? 5 : nstb == 4 ? 4 : 3)) * 128 * bkp * 4 / 1024;
        const long slots = (long)st().cu_count * (wpc8 > 0 ? wpc8 : std::max(1, std::min(160 / lds_kb, 3)));
        static const int tp2on = T4K_LAB_ENV("T4K_CONVBIG_DF8_TP2", 1);
        const int tp2 = (tp2on && C1 == 32) ? 1 : 0;         // two taps per 64-row tile
        const int kks = tp2 ? (KK + 1) / 2 : KK;
        const int tiles8 = kks * ci_tiles * co_tiles;
        long ns = slots / tiles8; if (ns < 1) ns = 1;
        long pp = (npix + ns - 1) / ns; pp = (pp + bkp - 1) / bkp * bkp; if (pp < 4 * bkp) pp = 4 * bkp;
        ns = (npix + pp - 1) / pp;
        if ((size_t)ns * C1 * KK * C0 > part_floats) return 0;
        static const int dbg8 = T4K_LAB_ENV("T4K_CONVBIG_DF8_DBG", 0);
        const int ctl = ci_tiles * co_tiles;
        Cd8 q = { I, DO, st().d_zero, part, N, H1, W1, C1, H0, W0, C0, (int)pp, ci_tiles, npix, dbg8, (int)ns, ctl, tp2 };
        const long groups = ns * ctl;
        const dim3 g8((unsigned)(8 * kks * ((groups + 7) / 8))), b8(512);
#define DF8_(k, s, pd, bk, ns_) do { static bool a1 = false; const int lb = ns_ * 128 * bk * 4; \
            if (!a1) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_convbig_df8<k, s, pd, bk, ns_>), hipFuncAttributeMaxDynamicSharedMemorySize, lb); a1 = true; } \
            T4K_LAUNCH((k_convbig_df8<k, s, pd, bk, ns_>), g8, b8, lb, hs, q); } while (0)
#define DF8(k, s, pd) do { if (bkp == 128) DF8_(k, s, pd, 128, 2); else if (bkp == 64 && nstb == 3) DF8_(k, s, pd, 64, 3); else if (bkp == 64) DF8_(k, s, pd, 64, 2); \
                           else if (nstb >= 5) DF8_(k, s, pd, 32, 5); else if (nstb == 4) DF8_(k, s, pd, 32, 4); else DF8_(k, s, pd, 32, 3); } while (0)
        switch ((K << 8) | (S << 4) | P) {
        case 0x110: DF8(1, 1, 0); break;
        case 0x311: DF8(3, 1, 1); break;
        case 0x421: DF8(4, 2, 1); break;
        case 0x512: DF8(5, 1, 2); break;
        }
#undef DF8
#undef DF8_
        return (int)ns;
    }
    if ((size_t)nslice * C1 * KK * C0 > part_floats) return 0;
    CdP p = { I, DO, part, N, H1, W1, C1, H0, W0, C0, (int)pps, ci_tiles };
    const dim3 g((unsigned)nslice, (unsigned)(KK * ci_tiles), (unsigned)co_tiles), b(256);
    switch ((K << 8) | (S << 4) | P) {
    case 0x110: T4K_LAUNCH((k_convbig_df<1, 1, 0>), g, b, 0, hs, p); break;
    case 0x311: T4K_LAUNCH((k_convbig_df<3, 1, 1>), g, b, 0, hs, p); break;
    case 0x421: T4K_LAUNCH((k_convbig_df<4, 2, 1>), g, b, 0, hs, p); break;
    case 0x512: T4K_LAUNCH((k_convbig_df<5, 1, 2>), g, b, 0, hs, p); break;
    }
    return (int)nslice;
}

} // namespace t4k
