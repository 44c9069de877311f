// gemm.hip - fp32 GEMM on gfx950 matrix cores (v_mfma_f32_32x32x2_f32, exact f32 fma chain).
//
//   O[M,N,C] = alpha * op(A) @ op(B) + beta * O      (reference k_gemm_tile_claude,
//   src/t4math.cu:478-583; host wrappers Tensor::gemm3 / mm / linear, src/mu/tensor.cu:73-87,161-180)
//
// Design (MI355X-first, not a translation of the reference's 64x64x16 VALU tiling):
//   * workgroup = 256 threads = 4 waves in a 2x2 grid; macro tile 64x64 (one 32x32 MFMA
//     accumulator per wave) when that yields <= ~1 tile per CU (the 1024^2 case: exactly 256
//     tiles on 256 CUs), 128x128 (2x2 accumulators per wave) for larger problems.
//   * K is consumed in stages of 32; LDS is double buffered; the global loads of stage t+1
//     are issued into registers before the MFMAs of stage t and written to LDS after them
//     (issue-early / write-late), one barrier per stage.
//   * an operand whose K axis is contiguous in memory (A normal, B transposed) is kept
//     [row][k] in LDS with a 16-byte XOR swizzle and read with ds_read_b128 (4 k-values per
//     lane, conflict free); the other kind is kept [k][row] and read with ds_read_b32.
//     Within each 8-deep k chunk, MFMA j consumes k = {j, 4+j} (lane halves), so one b128
//     read feeds four MFMAs.  Summation order inside a chunk is therefore 0,4,1,5,2,6,3,7.
//   * blockIdx -> tile mapping is XCD aware (block b runs on XCD b%8): each XCD gets a
//     compact 4-row band of tiles so its private 4 MiB L2 holds the A rows / B columns it re-reads.
//   * small outputs (few tiles, deep K - the CNN's linear layers) are split along K across
//     workgroups; partial slabs go to the library workspace and a second launch folds them in
//     slice order (deterministic; no fp32 atomics).
//   * arbitrary M/N/K tails, channel stride C > 1 and unaligned operands take the same
//     kernel with per-element predicated loads (VEC = false).
#include "t4k_common.h"

using namespace t4k;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmP {
    const float *A, *B;
    const float *bias;                 // optional per-column bias fused in the epilogue (k_bias nmath.cu:27)
    float *O, *part;
    int M, N, K, C;
    int tiles_m, tiles_n;
    int kchunk, nsplit;
    float alpha, beta;
};

constexpr int BK = 32;

template <int BM, int BN, bool AKC, bool BKC, bool VEC>
__global__ void __launch_bounds__(256) k_gemm_mfma(GemmP p) {
    constexpr int MT = BM / 64, NT = BN / 64;      // 32x32 fragments per wave (wave grid is 2x2)
    constexpr int PA = BM / 32, PB = BN / 32;      // 16-byte loads per thread per stage
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * BK];
    float *sA = lds, *sB = lds + 2 * BM * BK;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1, h = lane >> 5, l31 = lane & 31;
    const int c = blockIdx.z, C = p.C;
    const int M = p.M, N = p.N, K = p.K;

    // ---- XCD-aware, L2-friendly tile order ----
    const int T = p.tiles_m * p.tiles_n;
    int L;
    {
        const int b = blockIdx.x, q8 = T >> 3, r8 = T & 7, x = b & 7, i = b >> 3;
        L = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i;
    }
    constexpr int GROUP_M = 4;
    const int per_group = GROUP_M * p.tiles_n;
    const int grp = L / per_group, first_m = grp * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (L % per_group) % gsz, tn = (L % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int kbeg = blockIdx.y * p.kchunk;
    const int kend = min(K, kbeg + p.kchunk);
    const int nst  = (kend - kbeg + BK - 1) / BK;

    const float *__restrict__ A = p.A;
    const float *__restrict__ B = p.B;

    float4 ra[PA], rb[PB];

    auto ldg = [&](const float *X, bool ok, long idx) -> float4 {       // VEC: one 16-byte load
        return ok ? *reinterpret_cast<const float4 *>(X + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto load_tiles = [&](int kt) {
        const int k0 = kbeg + kt * BK;
#pragma unroll
        for (int pp = 0; pp < PA; pp++) {
            const int id = pp * 256 + tid;
            if (AKC) {                                      // A stored [M][K]
                const int r = id >> 3, q = id & 7, m = m0 + r, k = k0 + q * 4;
                if (VEC) ra[pp] = ldg(A, m < M && k < kend, (long)m * K + k);
                else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = (m < M && k + e < kend) ? A[((long)m * K + k + e) * C + c] : 0.f;
                    ra[pp] = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {                                        // A stored [K][M]
                const int kk = id / (BM / 4), rq = id % (BM / 4), k = k0 + kk, m = m0 + rq * 4;
                if (VEC) ra[pp] = ldg(A, k < kend && m < M, (long)k * M + m);
                else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = (k < kend && m + e < M) ? A[((long)k * M + m + e) * C + c] : 0.f;
                    ra[pp] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
#pragma unroll
        for (int pp = 0; pp < PB; pp++) {
            const int id = pp * 256 + tid;
            if (BKC) {                                      // B stored [N][K]
                const int r = id >> 3, q = id & 7, n = n0 + r, k = k0 + q * 4;
                if (VEC) rb[pp] = ldg(B, n < N && k < kend, (long)n * K + k);
                else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = (n < N && k + e < kend) ? B[((long)n * K + k + e) * C + c] : 0.f;
                    rb[pp] = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {                                        // B stored [K][N]
                const int kk = id / (BN / 4), rq = id % (BN / 4), k = k0 + kk, n = n0 + rq * 4;
                if (VEC) rb[pp] = ldg(B, k < kend && n < N, (long)k * N + n);
                else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = (k < kend && n + e < N) ? B[((long)k * N + n + e) * C + c] : 0.f;
                    rb[pp] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    };
    auto store_tiles = [&](int buf) {
        float *a = sA + buf * BM * BK, *b = sB + buf * BN * BK;
#pragma unroll
        for (int pp = 0; pp < PA; pp++) {
            const int id = pp * 256 + tid;
            if (AKC) { const int r = id >> 3, q = id & 7;
                *reinterpret_cast<float4 *>(a + r * BK + ((q ^ ((r >> 1) & 7)) << 2)) = ra[pp]; }
            else     { const int kk = id / (BM / 4), rq = id % (BM / 4);
                *reinterpret_cast<float4 *>(a + kk * BM + rq * 4) = ra[pp]; }
        }
#pragma unroll
        for (int pp = 0; pp < PB; pp++) {
            const int id = pp * 256 + tid;
            if (BKC) { const int r = id >> 3, q = id & 7;
                *reinterpret_cast<float4 *>(b + r * BK + ((q ^ ((r >> 1) & 7)) << 2)) = rb[pp]; }
            else     { const int kk = id / (BN / 4), rq = id % (BN / 4);
                *reinterpret_cast<float4 *>(b + kk * BN + rq * 4) = rb[pp]; }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    if (nst > 0) { load_tiles(0); store_tiles(0); }
    __syncthreads();

    for (int kt = 0; kt < nst; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nst) load_tiles(kt + 1);               // in flight during the MFMAs below
        const float *a = sA + buf * BM * BK, *b = sB + buf * BN * BK;
#pragma unroll
        for (int ci = 0; ci < 4; ci++) {
            float av[MT][4], bv[NT][4];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int r = wm * (BM / 2) + mt * 32 + l31;
                if (AKC) {
                    const float4 t = *reinterpret_cast<const float4 *>(a + r * BK + (((ci * 2 + h) ^ ((r >> 1) & 7)) << 2));
                    av[mt][0] = t.x; av[mt][1] = t.y; av[mt][2] = t.z; av[mt][3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) av[mt][j] = a[(ci * 8 + 4 * h + j) * BM + r];
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const int r = wn * (BN / 2) + nt * 32 + l31;
                if (BKC) {
                    const float4 t = *reinterpret_cast<const float4 *>(b + r * BK + (((ci * 2 + h) ^ ((r >> 1) & 7)) << 2));
                    bv[nt][0] = t.x; bv[nt][1] = t.y; bv[nt][2] = t.z; bv[nt][3] = t.w;
                } else {
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

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const float alpha = p.alpha, beta = p.beta;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int gn = n0 + wn * (BN / 2) + nt * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int gm = m0 + wm * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (gm < M && gn < N) {
                    if (p.nsplit > 1) {
                        p.part[((long)blockIdx.y * M + gm) * N + gn] = acc[mt][nt][r];
                    } else {
                        const long z = ((long)gm * N + gn) * C + c;
                        float o = acc[mt][nt][r] * alpha;
                        if (beta != 0.f) o += p.O[z] * beta;
                        if (p.bias) o += p.bias[gn];
                        p.O[z] = o;
                    }
                }
            }
        }
}

// fold split-K slabs in slice order, then the alpha/beta epilogue (reference t4math.cu:580)
__global__ void __launch_bounds__(BLK) k_splitk_fold(const float *__restrict__ part, float *O, long mn, int nsplit,
                                                     float alpha, float beta, const float *__restrict__ bias, int N) {
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < mn; z += (long)gridDim.x * BLK) {
        float s = 0.f;
        for (int k = 0; k < nsplit; k++) s += part[(long)k * mn + z];
        float o = s * alpha;
        if (beta != 0.f) o += O[z] * beta;
        if (bias) o += bias[z % N];
        O[z] = o;
    }
}

// words gemm1/gemm2 (k_gemm src/t4math.cu:370, k_gemm_claude :411): double accumulator
__global__ void __launch_bounds__(BLK) k_gemm_f64(const float *__restrict__ A, const float *__restrict__ B, float *O,
                                                  float alpha, float beta, int M, int N, int K, int C) {
    const long total = (long)M * N * C;
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < total; z += (long)gridDim.x * BLK) {
        const int c = (int)(z % C); const long mn = z / C; const int n = (int)(mn % N), m = (int)(mn / N);
        double acc = 0.0;
        for (int k = 0; k < K; k++) acc += (double)(A[((long)m * K + k) * C + c] * B[((long)k * N + n) * C + c]);
        O[z] = (float)(alpha * acc + (beta == 0.f ? 0.0 : (double)(beta * O[z])));
    }
}

template <int BM, int BN, bool VEC>
void launch_variant(const GemmP &p, dim3 grid, int tA, int tB, hipStream_t s) {
    if (!tA && !tB) hipLaunchKernelGGL((k_gemm_mfma<BM, BN, true,  false, VEC>), grid, dim3(256), 0, s, p);
    else if (!tA)   hipLaunchKernelGGL((k_gemm_mfma<BM, BN, true,  true,  VEC>), grid, dim3(256), 0, s, p);
    else if (!tB)   hipLaunchKernelGGL((k_gemm_mfma<BM, BN, false, false, VEC>), grid, dim3(256), 0, s, p);
    else            hipLaunchKernelGGL((k_gemm_mfma<BM, BN, false, true,  VEC>), grid, dim3(256), 0, s, p);
}

int gemm_launch(const float *A, const float *B, float *O, const float *bias, float alpha, float beta,
                int tA, int tB, int M, int N, int K, int C, t4k_stream_t s) {
    if (!A || !B || !O || M < 0 || N < 0 || K < 0 || C < 1) return fail(T4K_ERR_ARG, "t4k_gemm: bad argument");
    if (M == 0 || N == 0) return T4K_OK;
    GemmP p;
    p.A = A; p.B = B; p.O = O; p.bias = bias; p.part = (float *)st().ws;
    p.M = M; p.N = N; p.K = K; p.C = C; p.alpha = alpha; p.beta = beta;

    // 16-byte loads need: C == 1, aligned bases, contiguous extents divisible by 4
    const int a_contig = tA ? M : K, b_contig = tB ? K : N;
    const bool vec = (C == 1) && aligned16(A) && aligned16(B) && (a_contig % 4 == 0) && (b_contig % 4 == 0);

    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const bool big = t128 >= (long)st().cu_count * 3 / 4;
    const int BMv = big ? 128 : 64;
    p.tiles_m = (M + BMv - 1) / BMv; p.tiles_n = (N + BMv - 1) / BMv;
    const long tiles = (long)p.tiles_m * p.tiles_n;

    // split K when the output alone cannot fill the chip
    int nsplit = 1, kchunk = ((K + BK - 1) / BK) * BK; if (kchunk == 0) kchunk = BK;
    if (!big && C == 1 && tiles * 2 <= st().cu_count && K >= 4 * BK) {
        int want = (int)((st().cu_count + tiles - 1) / tiles);
        int maxs = K / (2 * BK); if (want > maxs) want = maxs; if (want > 64) want = 64;
        if (want > 1) {
            kchunk = (((K + want - 1) / want) + BK - 1) / BK * BK;
            nsplit = (K + kchunk - 1) / kchunk;
            if ((size_t)nsplit * M * N * sizeof(float) > st().ws_bytes) { nsplit = 1; kchunk = ((K + BK - 1) / BK) * BK; }
        }
    }
    p.kchunk = kchunk; p.nsplit = nsplit;

    dim3 grid((unsigned)tiles, (unsigned)nsplit, (unsigned)C);
    hipStream_t hs = S(s);
    if (big) { if (vec) launch_variant<128, 128, true>(p, grid, tA, tB, hs); else launch_variant<128, 128, false>(p, grid, tA, tB, hs); }
    else     { if (vec) launch_variant<64, 64, true>(p, grid, tA, tB, hs);   else launch_variant<64, 64, false>(p, grid, tA, tB, hs); }
    if (nsplit > 1) {
        const long mn = (long)M * N;
        hipLaunchKernelGGL(k_splitk_fold, dim3(grid_for(mn)), dim3(BLK), 0, hs, p.part, O, mn, nsplit, alpha, beta, bias, N);
    }
    T4K_LAUNCH_CHECK();
    return T4K_OK;
}

} // namespace

namespace t4k { int colsum_add(const float *X, float *OUT, long rows, int E, hipStream_t hs); }

extern "C" {

int t4k_gemm(const float *A, const float *B, float *O, float alpha, float beta,
             int tA, int tB, int M, int N, int K, int C, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    return gemm_launch(A, B, O, nullptr, alpha, beta, tA, tB, M, N, K, C, s);
}

// Model::_flinear src/nn/forward.cu:157-198: Y[N,E0] = X[N,E1] @ W[E0,E1]^T + B[E0], bias fused
int t4k_linear_fwd(const float *X, const float *W, const float *B, float *Y, int N, int E0, int E1, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    return gemm_launch(X, W, Y, B, 1.0f, 0.0f, 0, 1, N, E0, E1, 1, s);
}
// Model::_blinear src/nn/backprop.cu:193-254
int t4k_linear_bwd(const float *X, const float *W, const float *DY, float *DX, float *DW, float *DB,
                   int N, int E0, int E1, int train, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (train) {
        if (!DW || !DB) return fail(T4K_ERR_ARG, "t4k_linear_bwd: train needs DW/DB");
        int rc = colsum_add(DY, DB, N, E0, S(s)); if (rc) return rc;              // dB += sum_n dY
        rc = gemm_launch(DY, X, DW, nullptr, 1.0f, 1.0f, 1, 0, E0, E1, N, 1, s);  // dW += dY^T @ X
        if (rc) return rc;
    }
    return gemm_launch(DY, W, DX, nullptr, 1.0f, 0.0f, 0, 0, N, E1, E0, 1, s);   // dX = dY @ W (may overwrite X)
}

int t4k_gemm_f64acc(const float *A, const float *B, float *O, float alpha, float beta,
                    int M, int N, int K, int C, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!A || !B || !O || M < 0 || N < 0 || K < 0 || C < 1) return fail(T4K_ERR_ARG, "t4k_gemm_f64acc: bad argument");
    const long total = (long)M * N * C; if (total == 0) return T4K_OK;
    hipLaunchKernelGGL(k_gemm_f64, dim3(grid_for(total)), dim3(BLK), 0, S(s), A, B, O, alpha, beta, M, N, K, C);
    T4K_LAUNCH_CHECK();
    return T4K_OK;
}

} // extern "C"
