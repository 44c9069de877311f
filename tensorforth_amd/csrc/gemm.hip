// gemm.hip - fp32 GEMM on gfx950 matrix cores (v_mfma_f32_32x32x2_f32, exact f32 fma chain).
//
//   O[M,N,C] = alpha * op(A) @ op(B) + beta * O      (reference k_gemm_tile_claude,
//   src/t4math.cu:478-583; host wrappers Tensor::gemm3 / mm / linear, src/mu/tensor.cu:73-87,161-180)
//
// Design (MI355X-first, not a translation of the reference's 64x64x16 VALU tiling):
//   * workgroup = 256 threads = 4 waves in a 2x2 grid; macro tile 64x64 (one 32x32 MFMA
//     accumulator per wave) when that yields <= ~1 tile per CU (the 1024^2 case: exactly 256
//     tiles on 256 CUs), 128x128 (2x2 accumulators per wave) for larger problems.
//   * K is consumed in stages of BK (32 or 64); LDS is double buffered; the global loads of a
//     later stage are issued into registers before the MFMAs of the current one and written
//     to LDS after them (issue-early / write-late), one barrier per stage.
//   * SKEW: the MFMAs of a stage's last 8-deep k chunk are issued AFTER the barrier, so the
//     matrix pipe stays busy while the next stage's first operand reads are in flight.
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
#include <stdlib.h>

using namespace t4k;

namespace {
T4K_SPIN_DECL


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));       // first-class 16-byte value (HIP's float4 is a struct: its
                                                               // copies become memcpy and can pin staging arrays in scratch)

// mask-multiply backward of the element-wise run in front of a linear layer, applied to dX where it is produced:
// d1 = dX * m1 (the run's last stage), d2 = d1 * m2 (the stage in front of it); absent stages are nullptr
struct MaskChain { const float *m1; float *d1; const float *m2; float *d2; };
// riders of a GEMM's last launch (its split-K fold, or the epilogue of the small-tile kernel): see k_splitk_fold
struct FoldRider { ActEpi ep2; const float *cp_src; float *cp_dst; long cp_n; int cp_blocks, cp_vec; MaskChain mc; int mc_done; };

struct GemmP {
    const float *A, *B;
    const float *bias;                 // optional per-column bias fused in the epilogue (k_bias nmath.cu:27)
    float *O, *part;
    int M, N, K, C;
    int tiles_m, tiles_n;
    int kchunk, nsplit;
    float alpha, beta;
    int pair;                          // 1: two workgroups per tile (K halves) combine in the epilogue, no fold launch
    int *sync;                         // [0,2048) tickets, [2048,4096) flags (self-cleaning)
    // optional rider (generic 64x64 kernel only): workgroups beyond the tile grid add the column sums of a [rows, E] matrix
    // into cs_out - the bias gradient of a linear layer shares the launch of its weight-gradient GEMM
    const float *cs_X; float *cs_out; int cs_rows, cs_E;
    int xmap;                          // 32x32 kernels: 0 = tiles in launch order; 1 / 2 = XCD x (workgroup id % 8) owns a contiguous run of the row-major / column-major tile order
    const float *Z;                    // 4 KiB of zeros (State::d_zero): source of LDS-DMA lanes past the K range / the matrix edge (DMA variants of the 32x32 kernels)
};

// FULL: every K slice is whole stages (K-slice%BK == 0, VEC): no predicates, no branches in the K loop, so the compiler can sink the
// next stage's loads and address math under the MFMAs.  Ragged M / N edges are handled by clamped source rows and predicated stores.
// gate (dual launches, see k_gemm_dual): mode 1 = this GEMM READS a buffer the other one overwrites: signal once the K loop
// has consumed every load; mode 2 = this GEMM is the writer: hold the epilogue stores until gate_n readers have signalled.
template <int BM, int BN, int BK, bool AKC, bool BKC, bool VEC, bool SKEW, bool FULL>
__device__ __forceinline__ void gemm_mfma_body(const GemmP &p, const int bx, const int by, const int bz,
                                               int *gate = nullptr, const int gate_mode = 0, const int gate_n = 0, const int gate_m = 0,
                                               const MaskChain *mc = nullptr) {
    constexpr int MT = BM / 64, NT = BN / 64;      // 32x32 fragments per wave (wave grid is 2x2)
    constexpr int PA = BM * BK / 1024, PB = BN * BK / 1024;   // 16-byte loads per thread per stage
    constexpr int NC = BK / 8;                     // 8-deep k chunks per stage
    constexpr int CH = BK / 4;                     // 16-byte chunks per LDS row of a K-contiguous operand
    constexpr int SW = (64 / BK) > 0 ? (64 / BK) : 1;   // rows per 256-byte LDS bank row
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *sA = lds, *sB = lds + 2 * BM * BK;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1, h = lane >> 5, l31 = lane & 31;
    const int c = bz, C = p.C;
    const int M = p.M, N = p.N, K = p.K;

    // ---- XCD-aware, L2-friendly tile order ----
    const int T = p.tiles_m * p.tiles_n;
    if (bx >= T) {                                 // rider workgroups: cs_out[e] += sum_r cs_X[r, e] (k_dlinear_db nmath.cu:274-280)
        const int ex = tid & 63, ry = tid >> 6, e = (bx - T) * 64 + ex;
        float a = 0.f;
        if (e < p.cs_E) {
#pragma unroll 8
            for (int r = ry; r < p.cs_rows; r += 4) a += p.cs_X[(long)r * p.cs_E + e];
        }
        lds[ry * 64 + ex] = a;
        __syncthreads();
        if (ry == 0 && e < p.cs_E) p.cs_out[e] += (lds[ex] + lds[64 + ex]) + (lds[128 + ex] + lds[192 + ex]);
        return;
    }
    int L;
    {
        const int b = bx, q8 = T >> 3, r8 = T & 7, x = b & 7, i = b >> 3;
        L = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i;
    }
    constexpr int GROUP_M = 4;
    const int per_group = GROUP_M * p.tiles_n;
    const int grp = L / per_group, first_m = grp * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (L % per_group) % gsz, tn = (L % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int kbeg = by * p.kchunk;
    const int kend = min(K, kbeg + p.kchunk);
    const int nst  = (kend - kbeg + BK - 1) / BK;

    const float *__restrict__ A = p.A;
    const float *__restrict__ B = p.B;

    v4f ra[PA], rb[PB];                 // staging register set 0
    v4f ra2[PA], rb2[PB];               // set 1 (FULL path: loads run two stages ahead)

    auto ldg = [&](const float *X, bool ok, long idx) -> v4f {          // VEC: one 16-byte load
        v4f z = {0.f, 0.f, 0.f, 0.f};
        return ok ? *reinterpret_cast<const v4f *>(X + idx) : z;
    };
    // !VEC: 4 predicated scalar loads; v0 is the fixed coordinate, v1.. the contiguous one
    auto lds4 = [&](const float *X, long idx, int lim0, int lim1, int v0, int v1) -> v4f {
        v4f v;
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = (v0 < lim0 && v1 + e < lim1) ? X[(idx + e) * C + c] : 0.f;
        return v;
    };
    // FULL: per-thread source pointers of stage 0; stage kt is at + kt * step
    const v4f *ga[PA], *gb[PB];
    if (FULL) {
#pragma unroll
        for (int pp = 0; pp < PA; pp++) {
            const int id = pp * 256 + tid;
            // rows / 4-column groups beyond the matrix edge are CLAMPED to the last valid one: the loads stay unpredicated (valid memory,
            // finite or not - those values only reach accumulator rows / columns the epilogue never stores)
            if (AKC) ga[pp] = reinterpret_cast<const v4f *>(A + (long)min(m0 + id / CH, M - 1) * K + kbeg + (id % CH) * 4);
            else     ga[pp] = reinterpret_cast<const v4f *>(A + (long)(kbeg + id / (BM / 4)) * M + min(m0 + (id % (BM / 4)) * 4, M - 4));
        }
#pragma unroll
        for (int pp = 0; pp < PB; pp++) {
            const int id = pp * 256 + tid;
            if (BKC) gb[pp] = reinterpret_cast<const v4f *>(B + (long)min(n0 + id / CH, N - 1) * K + kbeg + (id % CH) * 4);
            else     gb[pp] = reinterpret_cast<const v4f *>(B + (long)(kbeg + id / (BN / 4)) * N + min(n0 + (id % (BN / 4)) * 4, N - 4));
        }
    }
    const long ga_step = AKC ? BK / 4 : (long)BK * M / 4, gb_step = BKC ? BK / 4 : (long)BK * N / 4;   // in float4
    auto load_into = [&](int kt, v4f (&ra)[PA], v4f (&rb)[PB]) __attribute__((always_inline)) {
        if (FULL) {
#pragma unroll
            for (int pp = 0; pp < PA; pp++) ra[pp] = ga[pp][(long)kt * ga_step];
#pragma unroll
            for (int pp = 0; pp < PB; pp++) rb[pp] = gb[pp][(long)kt * gb_step];
            return;
        }
        const int k0 = kbeg + kt * BK;
#pragma unroll
        for (int pp = 0; pp < PA; pp++) {
            const int id = pp * 256 + tid;
            if (AKC) {                                      // A stored [M][K]
                const int r = id / CH, q = id % CH, m = m0 + r, k = k0 + q * 4;
                ra[pp] = VEC ? ldg(A, m < M && k < kend, (long)m * K + k) : lds4(A, (long)m * K + k, M, kend, m, k);
            } else {                                        // A stored [K][M]
                const int kk = id / (BM / 4), rq = id % (BM / 4), k = k0 + kk, m = m0 + rq * 4;
                ra[pp] = VEC ? ldg(A, k < kend && m < M, (long)k * M + m) : lds4(A, (long)k * M + m, kend, M, k, m);
            }
        }
#pragma unroll
        for (int pp = 0; pp < PB; pp++) {
            const int id = pp * 256 + tid;
            if (BKC) {                                      // B stored [N][K]
                const int r = id / CH, q = id % CH, n = n0 + r, k = k0 + q * 4;
                rb[pp] = VEC ? ldg(B, n < N && k < kend, (long)n * K + k) : lds4(B, (long)n * K + k, N, kend, n, k);
            } else {                                        // B stored [K][N]
                const int kk = id / (BN / 4), rq = id % (BN / 4), k = k0 + kk, n = n0 + rq * 4;
                rb[pp] = VEC ? ldg(B, k < kend && n < N, (long)k * N + n) : lds4(B, (long)k * N + n, kend, N, k, n);
            }
        }
    };
    // LDS store offsets (floats) are loop invariant
    int soa[PA], sob[PB];
#pragma unroll
    for (int pp = 0; pp < PA; pp++) {
        const int id = pp * 256 + tid;
        if (AKC) { const int r = id / CH, q = id % CH; soa[pp] = r * BK + ((q ^ ((r / SW) & (CH - 1))) << 2); }
        else     { soa[pp] = (id / (BM / 4)) * BM + (id % (BM / 4)) * 4; }
    }
#pragma unroll
    for (int pp = 0; pp < PB; pp++) {
        const int id = pp * 256 + tid;
        if (BKC) { const int r = id / CH, q = id % CH; sob[pp] = r * BK + ((q ^ ((r / SW) & (CH - 1))) << 2); }
        else     { sob[pp] = (id / (BN / 4)) * BN + (id % (BN / 4)) * 4; }
    }
    auto store_from = [&](int buf, v4f (&ra)[PA], v4f (&rb)[PB]) __attribute__((always_inline)) {
        float *a = sA + buf * BM * BK, *b = sB + buf * BN * BK;
#pragma unroll
        for (int pp = 0; pp < PA; pp++) *reinterpret_cast<v4f *>(a + soa[pp]) = ra[pp];
#pragma unroll
        for (int pp = 0; pp < PB; pp++) *reinterpret_cast<v4f *>(b + sob[pp]) = rb[pp];
    };
    auto load_tiles  = [&](int kt)  __attribute__((always_inline)) { load_into(kt, ra, rb); };
    auto store_tiles = [&](int buf) __attribute__((always_inline)) { store_from(buf, ra, rb); };
    // operand fragments of one 8-deep k chunk: lane half h holds k = 8*ci + 4*h + {0..3}
    auto read_chunk = [&](const float *a, const float *b, int ci, float (&av)[MT][4], float (&bv)[NT][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int r = wm * (BM / 2) + mt * 32 + l31;
            if (AKC) {
                const v4f t = *reinterpret_cast<const v4f *>(a + r * BK + (((ci * 2 + h) ^ ((r / SW) & (CH - 1))) << 2));
                av[mt][0] = t[0]; av[mt][1] = t[1]; av[mt][2] = t[2]; av[mt][3] = t[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) av[mt][j] = a[(ci * 8 + 4 * h + j) * BM + r];
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int r = wn * (BN / 2) + nt * 32 + l31;
            if (BKC) {
                const v4f t = *reinterpret_cast<const v4f *>(b + r * BK + (((ci * 2 + h) ^ ((r / SW) & (CH - 1))) << 2));
                bv[nt][0] = t[0]; bv[nt][1] = t[1]; bv[nt][2] = t[2]; bv[nt][3] = t[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) bv[nt][j] = b[(ci * 8 + 4 * h + j) * BN + r];
            }
        }
    };

    // A wave with a single 32x32 fragment keeps NACC = 2 accumulator chains (even / odd k-pairs):
    // back-to-back MFMAs on ONE accumulator lose the forwarding path as soon as a ds_read or
    // s_waitcnt sits between them (+43 cycles per pair, MI355X_MICROARCH.md), two chains do not.
    constexpr int NACC = (MT * NT == 1) ? 2 : 1;
    f32x16 acc[MT][NT][NACC];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int q = 0; q < NACC; q++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][q][r] = 0.f;

    auto mma_chunk = [&](float (&av)[MT][4], float (&bv)[NT][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    acc[mt][nt][j % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt][j], bv[nt][j], acc[mt][nt][j % NACC], 0, 0, 0);
    };

    // beta != 0 (dW += ...): the old output values are fetched up front instead of as dependent loads after the K loop
    constexpr bool PRE = (MT * NT == 1);
    float oprev[16];
    if (PRE && p.beta != 0.f && p.nsplit == 1) {
        const int gn = n0 + wn * (BN / 2) + l31;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int gm = m0 + wm * (BM / 2) + (r & 3) + 8 * (r >> 2) + 4 * h;
            oprev[r] = (gm < M && gn < N) ? p.O[((long)gm * N + gn) * C + c] : 0.f;
        }
    }

    if (nst > 0) { load_tiles(0); store_tiles(0); }
    __syncthreads();

    if (FULL && !SKEW) {
        // loads run TWO stages ahead: stage t+1 sits in one register set while stage t+2 lands in the other
        if (nst > 1) load_into(1, ra, rb);
        if (nst > 2) load_into(2, ra2, rb2);
        auto stage = [&](int kt, v4f (&rx)[PA], v4f (&ry)[PB]) __attribute__((always_inline)) {
            const int buf = kt & 1;
            const float *a = sA + buf * BM * BK, *b = sB + buf * BN * BK;
#pragma unroll
            for (int ci = 0; ci < NC; ci++) {
                float av[MT][4], bv[NT][4];
                read_chunk(a, b, ci, av, bv);
                mma_chunk(av, bv);
            }
            if (kt + 1 < nst) store_from(buf ^ 1, rx, ry);      // stage kt+1 (loaded a full stage ago)
            __syncthreads();
            if (kt + 3 < nst) load_into(kt + 3, rx, ry);        // refill the freed set
        };
        for (int kt = 0; kt < nst; kt += 2) {
            stage(kt, ra, rb);
            if (kt + 1 < nst) stage(kt + 1, ra2, rb2);
        }
    } else if (!SKEW) {
        for (int kt = 0; kt < nst; kt++) {
            const int buf = kt & 1;
            if (kt + 1 < nst) load_tiles(kt + 1);           // in flight during the MFMAs below
            const float *a = sA + buf * BM * BK, *b = sB + buf * BN * BK;
#pragma unroll
            for (int ci = 0; ci < NC; ci++) {
                float av[MT][4], bv[NT][4];
                read_chunk(a, b, ci, av, bv);
                mma_chunk(av, bv);
            }
            if (kt + 1 < nst) store_tiles(buf ^ 1);
            __syncthreads();
        }
    } else if (nst > 0) {
        float cav[MT][4], cbv[NT][4];                       // chunk whose MFMAs are pending
        if (nst > 1) load_tiles(1);
        read_chunk(sA, sB, 0, cav, cbv);
        for (int kt = 0; kt < nst; kt++) {
            const int buf = kt & 1;
            const float *a = sA + buf * BM * BK, *b = sB + buf * BN * BK;
#pragma unroll
            for (int ci = 0; ci + 1 < NC; ci++) {
                float nav[MT][4], nbv[NT][4];
                read_chunk(a, b, ci + 1, nav, nbv);
                mma_chunk(cav, cbv);
#pragma unroll
                for (int j = 0; j < 4; j++) {
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) cav[mt][j] = nav[mt][j];
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) cbv[nt][j] = nbv[nt][j];
                }
            }
            if (kt + 1 < nst) store_tiles(buf ^ 1);         // stage kt+1: registers -> the other buffer
            __syncthreads();
            if (kt + 2 < nst) load_tiles(kt + 2);
            float nav[MT][4], nbv[NT][4];
            if (kt + 1 < nst) read_chunk(sA + (buf ^ 1) * BM * BK, sB + (buf ^ 1) * BN * BK, 0, nav, nbv);
            mma_chunk(cav, cbv);                            // last chunk of stage kt covers the reads above
            if (kt + 1 < nst) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) cav[mt][j] = nav[mt][j];
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) cbv[nt][j] = nbv[nt][j];
                }
            }
        }
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const float alpha = p.alpha, beta = p.beta;
    if (gate_mode == 1) {                                   // every load of the shared buffer has been consumed
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(gate, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (gate_mode == 2) {
        if (tid == 0) T4K_SPIN_WAIT(__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gate_n, 1);
        __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int gn = n0 + wn * (BN / 2) + nt * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int gm = m0 + wm * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (gm < M && gn < N) {
                    float v = acc[mt][nt][0][r];
                    if (NACC == 2) v += acc[mt][nt][NACC - 1][r];
                    if (p.nsplit > 1) {
                        p.part[((long)by * M + gm) * N + gn] = v;
                    } else {
                        const long z = ((long)gm * N + gn) * C + c;
                        float o = v * alpha;
                        if (beta != 0.f) o += (PRE ? oprev[r] : p.O[z]) * beta;
                        if (p.bias) o += p.bias[gn];
                        p.O[z] = o;
                        if (mc && mc->d1) { const float g1 = o * mc->m1[z]; mc->d1[z] = g1; if (mc->d2) mc->d2[z] = g1 * mc->m2[z]; }
                    }
                }
            }
        }
    if (gate_mode == 2) {                                   // last writer re-arms the gate for the next launch
        __syncthreads();
        if (tid == 0) {
            const int t = __hip_atomic_fetch_add(gate + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gate_m - 1) {
                __hip_atomic_store(gate, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gate + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
template <int BM, int BN, int BK, bool AKC, bool BKC, bool VEC, bool SKEW, bool FULL>
__global__ void __launch_bounds__(256) k_gemm_mfma(GemmP p) {
    gemm_mfma_body<BM, BN, BK, AKC, BKC, VEC, SKEW, FULL>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}
// ------------------------------------------------------------------------------------------
// Latency-shaped GEMM for the slivers of small-batch training (a 256-row batch gives 8..100 tiles of 64x64: most CUs idle, every
// workgroup a serial chain of memory round trips).  One workgroup = one 32x32 output tile; its four waves split K (k-groups) and
// fetch their operand fragments STRAIGHT INTO REGISTERS in the MFMA operand layout - no LDS staging, no barriers in the K loop,
// 64 k of loads in flight per wave before the first MFMA and the next 64 issued under it.  The four partial accumulators meet in
// LDS once; each wave then finishes a quarter of the tile (alpha / beta / bias, mask chain).  4x the workgroups of the 64x64
// kernels, each with 1/16 of the matrix work per wave.
// Operand layout of v_mfma_f32_32x32x2_f32: lane (l31, h) supplies A[row = l31][k] and B[k][col = l31] for one k per instruction;
// chunk c covers k = 8c + 4h + {0..3}, as in the LDS kernels above.
// One 32x32 output tile per workgroup, K split over its NW waves (k-groups) which meet once in LDS; the whole epilogue rides (bias, activation + dropout riders, mask
// chain, column-sum and copy riders, split-K slabs).  The operand fragments come through wave-private LDS blocks filled by global_load_lds_dwordx4 (see the K loop):
// `red` is the workgroup's dynamic LDS of NW x 16 KiB; k-group w's partial accumulators land at red + w RS.  (Round 6: the register-fetch form of this body -
// k_gemm_s32 / k_gemm_dual32, row gathers of 16 bytes - is gone; shapes whose operands the DMA cannot take go to the 64x64 kernels.)
template <bool AKC, bool BKC, int NW = 4, bool RST = false>   // NW waves = NW k-groups per 32x32 tile; RST: blocks 2, 3 of a wave's range wait in registers
__device__ __forceinline__ void gemm_s32_body(const GemmP &p, const int bx, float *red,
                                              const int gate_mode = 0, const int gate_n = 0,
                                              const MaskChain *mc = nullptr, unsigned *slots = nullptr, const unsigned epoch = 0,
                                              const int by = 0, const FoldRider *fe = nullptr, const ActEpi *ep1 = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int M = p.M, N = p.N, K = p.K;
    const int kbeg = by * p.kchunk, kend = min(K, kbeg + p.kchunk);      // this workgroup's k range (split-K: slab `by`)
    const int T = p.tiles_m * p.tiles_n;
    if (bx >= T) {                                 // rider workgroups: cs_out[e] += sum_r cs_X[r, e] (k_dlinear_db nmath.cu:274-280)
        const int ex = tid & 63, ry = tid >> 6, e = (bx - T) * 64 + ex;
        float a = 0.f;
        if (e < p.cs_E && ry < 4) {
#pragma unroll 8
            for (int r = ry; r < p.cs_rows; r += 4) a += p.cs_X[(long)r * p.cs_E + e];
        }
        if (ry < 4) red[ry * 64 + ex] = a;
        __syncthreads();
        if (ry == 0 && e < p.cs_E) p.cs_out[e] += (red[ex] + red[64 + ex]) + (red[128 + ex] + red[192 + ex]);
        return;
    }
    // XCD-aware tile order (xmap): workgroup id % 8 is the XCD a workgroup runs on (private L2s).  In launch order neighbouring tiles - which share
    // an operand - sit on eight different XCDs and every L2 pulls both operands whole; a contiguous run of the column-major order gives an XCD
    // its own slice of B (2: outputs wider than tall), of the row-major order its own slice of A (1).
    int tm, tn;
    {
        int Lt = bx;                                           // (bx may be offset by a constant from the physical id - second GEMM of a dual launch: the groups bx % 8 are still the XCDs)
        if (p.xmap) { const int q8 = T >> 3, r8 = T & 7, x = bx & 7, i = bx >> 3; Lt = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i; }
        if (p.xmap == 2) { tn = Lt / p.tiles_m; tm = Lt - tn * p.tiles_m; } else { tm = Lt / p.tiles_n; tn = Lt - tm * p.tiles_n; }
    }
    const int tile = tm * p.tiles_n + tn;                  // logical id: the arrival slots are indexed by it
    const int m0 = tm * 32, n0 = tn * 32;
    // the share of the tile this wave finishes: accumulator registers QN w .. QN w + QN - 1 (QN = 16 / NW), row of register r = (r & 3) + 8 (r >> 2) + 4 h
    constexpr int QN = 16 / NW;
    const int gn = n0 + l31;
    float oprev[QN];
#pragma unroll
    for (int q = 0; q < QN; q++) oprev[q] = 0.f;
    if (p.beta != 0.f && p.nsplit == 1) {
#pragma unroll
        for (int q = 0; q < QN; q++) { const int r = QN * w + q, gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h; if (gm < M && gn < N) oprev[q] = p.O[(long)gm * N + gn]; }
    }
    // every read-only operand of the epilogue is requested here, with the K loop's first loads: fetched behind the reduction barrier, the
    // bias and the masks of the chain would each add a memory round trip to a launch that is little else
    float bias_v = 0.f, mk1[QN], mk2[QN];
    if (p.bias && p.nsplit == 1 && gn < N) bias_v = p.bias[gn];
#pragma unroll
    for (int q = 0; q < QN; q++) {
        mk1[q] = 0.f; mk2[q] = 0.f;
        if (mc && mc->d1 && p.nsplit == 1) {
            const int r = QN * w + q, gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (gm < M && gn < N) { const long z = (long)gm * N + gn; mk1[q] = mc->m1[z]; if (mc->d2) mk2[q] = mc->m2[z]; }
        }
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // gate_mode 1 with per-wave slots (gate_n <= 128 reader workgroups): a wave reports as soon as its LAST operand fragments sit in
    // registers, in front of its last MFMA batch - the writers' wait then overlaps that batch, the LDS reduction and the epilogue
    const bool early = gate_mode == 1 && gate_n <= 128 && NW == 4;       // per-wave slots: 4 per reader workgroup
    auto arrive = [&]() __attribute__((always_inline)) {
        if (!early) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(slots + 4 * tile + w, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    constexpr int RS = 4096;                                      // floats between two k-groups' partial accumulators in `red`
    {
        // Coalesced operand fetch for slivers.  The register path above makes every wave load a gather (32 rows x 16 bytes: 64 cache-line
        // look-ups per instruction for 1 KiB, the address unit's queue stalls the wave's issue - SQ_WAIT_INST_ANY 42 % on the K-contiguous
        // forward layers).  Here a wave moves its k range in blocks of 32 k through two private LDS slots (A 4 KiB + B 4 KiB each):
        // a DMA instruction takes whole 128-byte runs (8 rows x 32 k of a K-contiguous operand, 8 k rows x 32 columns of the other kind),
        // no VGPR round trip, no barrier in the K loop (the blocks are the wave's own: s_waitcnt vmcnt only).
        //   K-contiguous block [32 rows][32 k]: the 16-byte quad q of row r sits at quad (q ^ ((r >> 1) & 7)) - the 16 lanes of a
        //   ds_read_b128 group ({0-3,12-15,20-27} ...) then cover all 64 banks once;  [K][M] block: [32 k][32 columns], ds_read_b32 rows.
        typedef __attribute__((address_space(3))) const float lds_f;
        typedef __attribute__((address_space(3))) const v4f lds_v4;
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)red;
        const int wu = __builtin_amdgcn_readfirstlane(w);
        const int nblk = (kend - kbeg + 31) >> 5;
        const int b0 = wu * nblk / NW, b1 = (wu + 1) * nblk / NW;
        const int i8 = lane >> 3, i7 = lane & 7;
        const float *pa[4], *pb[4]; int ka[4], kb[4];            // lane's source of DMA instruction j at k = 0, and the k (within a block) its validity hangs on
        const bool acol = AKC || m0 + 4 * i7 < M, bcol = BKC || n0 + 4 * i7 < N;
        const float *zsrc = p.Z + 4 * i7;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = 8 * j + i8, q = i7 ^ ((r >> 1) & 7);
            if (AKC) { pa[j] = p.A + (long)min(m0 + r, M - 1) * K + 4 * q; ka[j] = 4 * q; } else { pa[j] = p.A + (long)r * M + m0 + 4 * i7; ka[j] = r; }
            if (BKC) { pb[j] = p.B + (long)min(n0 + r, N - 1) * K + 4 * q; kb[j] = 4 * q; } else { pb[j] = p.B + (long)r * N + n0 + 4 * i7; kb[j] = r; }
        }
        auto issue = [&](int b, int slot) __attribute__((always_inline)) {
            const int k0 = kbeg + 32 * b;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float *sa = (k0 + ka[j] < kend && acol) ? (AKC ? pa[j] + k0 : pa[j] + (long)k0 * M) : zsrc;
                const float *sb = (k0 + kb[j] < kend && bcol) ? (BKC ? pb[j] + k0 : pb[j] + (long)k0 * N) : zsrc;
                const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((wu * 4096 + slot * 2048 + j * 256) * 4));
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(sa), "s"(la) : "memory");
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(sb), "s"(la + 4096u) : "memory");
            }
        };
        const int sw = (l31 >> 1) & 7;
        auto frag = [&](int slot, float (&fa)[4][4], float (&fb)[4][4]) __attribute__((always_inline)) {
            lds_f *a = (lds_f *)red + wu * 4096 + slot * 2048, *b = a + 1024;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (AKC) { const v4f t = *(lds_v4 *)(a + l31 * 32 + (((2 * c + h) ^ sw) << 2)); fa[c][0] = t[0]; fa[c][1] = t[1]; fa[c][2] = t[2]; fa[c][3] = t[3]; }
                else {
#pragma unroll
                    for (int j = 0; j < 4; j++) fa[c][j] = a[(8 * c + 4 * h + j) * 32 + l31];
                }
                if (BKC) { const v4f t = *(lds_v4 *)(b + l31 * 32 + (((2 * c + h) ^ sw) << 2)); fb[c][0] = t[0]; fb[c][1] = t[1]; fb[c][2] = t[2]; fb[c][3] = t[3]; }
                else {
#pragma unroll
                    for (int j = 0; j < 4; j++) fb[c][j] = b[(8 * c + 4 * h + j) * 32 + l31];
                }
            }
        };
        // Everything a wave needs is requested before its first wait: blocks 0 and 1 by DMA into the two slots, blocks 2 and 3 (RST) into
        // registers with the DMA's own lane -> address map (coalesced) - they are written to a slot (ds_write_b128, the DMA's image) once
        // its fragments have been read.  A second round trip would cost more than the copies: the fetch is what bounds these launches.
        const int nb = b1 - b0;
        v4f rs[RST ? 2 : 1][8];
        auto regload = [&](int b, int x) __attribute__((always_inline)) {
            const int k0 = kbeg + 32 * b;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float *sa = (k0 + ka[j] < kend && acol) ? (AKC ? pa[j] + k0 : pa[j] + (long)k0 * M) : zsrc;
                const float *sb = (k0 + kb[j] < kend && bcol) ? (BKC ? pb[j] + k0 : pb[j] + (long)k0 * N) : zsrc;
                rs[x][j] = *reinterpret_cast<const v4f *>(sa); rs[x][4 + j] = *reinterpret_cast<const v4f *>(sb);
            }
        };
        auto regstore = [&](int x, int slot) __attribute__((always_inline)) {
            typedef __attribute__((address_space(3))) v4f lds_w4;
            lds_w4 *d = (lds_w4 *)((__attribute__((address_space(3))) float *)red + wu * 4096 + slot * 2048) + lane;
#pragma unroll
            for (int j = 0; j < 4; j++) { d[j * 64] = rs[x][j]; d[256 + j * 64] = rs[x][4 + j]; }
        };
        auto wait_vm = [&](int blocks_after) __attribute__((always_inline)) {       // DMA and register loads return in issue order
            if (blocks_after >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (blocks_after == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (blocks_after == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        auto mm4 = [&](float (&fa)[4][4], float (&fb)[4][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][2], fb[c][2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][3], fb[c][3], acc1, 0, 0, 0);
            }
        };
        if (nb > 0) {
            issue(b0, 0);
            if (nb > 1) issue(b0 + 1, 1);
            if (RST) { if (nb > 2) regload(b0 + 2, 0); if (nb > 3) regload(b0 + 3, 1); }
            constexpr int AHEAD = RST ? 4 : 2;                   // blocks requested up front
            int slot = 0;
            for (int i = 0; i < nb; i++) {
                if (i < 2 || !RST) wait_vm(min(nb, i < AHEAD ? AHEAD : i + 2) - 1 - i);   // blocks 2, 3 of RST: the register copies' own waits (compiler-counted) cover them
                else if (i >= 4) wait_vm(0);                      // deeper ranges (not dispatched today): DMA again, one block at a time
                float fa[4][4], fb[4][4];
                frag(slot, fa, fb);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (RST) { if (i == 0 && nb > 2) regstore(0, 0); if (i == 1 && nb > 3) regstore(1, 1); if (i + 2 < nb && i >= 2) issue(b0 + i + 2, slot); }
                else if (i + 2 < nb) issue(b0 + i + 2, slot);    // the slot's fragments sit in registers
                if (i + 1 >= nb) arrive();
                mm4(fa, fb);
                slot ^= 1;
            }
        } else arrive();
    }
    // the four k-groups meet in LDS: red[w][r][lane]
#pragma unroll
    for (int r = 0; r < 16; r++) red[w * RS + r * 64 + lane] = acc0[r] + acc1[r];
    // In-place dX: a writer waits until the reader workgroups of ITS columns have consumed their loads of the shared buffer.  No
    // counters: a reader stores this launch's epoch into its slot (fire and forget), the writer's first wave polls those slots with
    // agent-scope loads until all of them carry the epoch - one store and one load round trip on the critical path, nothing to re-arm.
    if (gate_mode == 1) {
        __syncthreads();
        if (!early && tid == 0) __hip_atomic_store(slots + tile, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (gate_mode == 2) {
        if (w == 0) {
            // only the readers of the columns this tile overwrites matter: dW tiles (e0t, tn), e0t = 0 .. gate_n / tiles_n - 1 (both GEMMs
            // have the same column tiling), each with 4 per-wave slots when those fit the 512-int block (gate_n <= 128)
            const int per = (gate_n <= 128 && NW == 4) ? 4 : 1, rows = gate_n / p.tiles_n, nslot = rows * per;
            for (int spin_it = 0;; spin_it++) {
                if (spin_it > T4K_SPIN_MAX) { if (lane == 0 && g_spin_err_dev) __hip_atomic_store(g_spin_err_dev, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }   // bounded: see t4k_common.h
                bool ok = true;
                unsigned bad = 0;                                  // no short-circuit: the loads of one pass are independent and go out together
#pragma unroll 4
                for (int i = lane; i < nslot; i += 64) {
                    const int e0t = i / per, ww = i - e0t * per;
                    bad |= __hip_atomic_load(slots + per * (e0t * p.tiles_n + tn) + ww, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ epoch;
                }
                ok = bad == 0;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    } else __syncthreads();
    const float alpha = p.alpha, beta = p.beta;
#pragma unroll
    for (int q = 0; q < QN; q++) {
        const int r = QN * w + q, gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = red[r * 64 + lane];
#pragma unroll
        for (int g = 1; g < NW; g++) v += red[g * RS + r * 64 + lane];           // k-groups in order
        if (gm < M && gn < N) {
            const long z = (long)gm * N + gn;
            if (p.nsplit > 1) { p.part[(long)by * M * N + z] = v; continue; }       // split-K slab: the consumer folds (XFold / k_splitk_fold)
            float o = v * alpha;
            if (beta != 0.f) o += oprev[q] * beta;
            if (p.bias) o += bias_v;
            p.O[z] = o;
            if (mc && mc->d1) { const float g1 = o * mk1[q]; mc->d1[z] = g1; if (mc->d2) mc->d2[z] = g1 * mk2[q]; }
            if (ep1 && ep1->layer) {                                              // element-wise layer(s) behind a linear layer: as k_splitk_fold
                const bool d1 = ep1->layer == T4K_L_DROPOUT, d2 = fe && fe->ep2.layer == T4K_L_DROPOUT;
                float u = 0.f;
                if (d1 || d2) { uint64_t base, seed; rng_begin(d2 ? fe->ep2.rng : ep1->rng, base, seed); u = philox_u01_at(base, seed, z); }
                float a, f; act_rt(ep1->layer, o, d1 ? u : 0.f, ep1->alpha, a, f); ep1->F[z] = f; ep1->A[z] = a;
                if (fe && fe->ep2.layer) { float a2, f2; act_rt(fe->ep2.layer, a, d2 ? u : 0.f, fe->ep2.alpha, a2, f2); fe->ep2.F[z] = f2; fe->ep2.A[z] = a2; }
            }
        }
    }
}
// one GEMM on 32x32 tiles (see gemm_s32_body): grid = (tiles + column-sum riders + copy riders, k slabs); epilogue riders as the fold launch's; dynamic LDS = NW x 16 KiB
template <bool AKC, bool BKC, int NW, bool RST>
__global__ void __launch_bounds__(64 * NW) k_gemm_l32(GemmP p, ActEpi ep, FoldRider fr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nwork = (int)gridDim.x - fr.cp_blocks;
    if ((int)blockIdx.x >= nwork) {                              // the model's copy of the batch into its layer 0 rides along (forward.cu:39)
        if (blockIdx.y) return;
        const long t0 = (long)((int)blockIdx.x - nwork) * (64 * NW) + threadIdx.x, step = (long)fr.cp_blocks * (64 * NW);
        if (fr.cp_vec) {
            const long n4 = fr.cp_n >> 2;
            for (long z = t0; z < n4; z += step) reinterpret_cast<float4 *>(fr.cp_dst)[z] = reinterpret_cast<const float4 *>(fr.cp_src)[z];
            for (long z = (n4 << 2) + t0; z < fr.cp_n; z += step) fr.cp_dst[z] = fr.cp_src[z];
        } else
            for (long z = t0; z < fr.cp_n; z += step) fr.cp_dst[z] = fr.cp_src[z];
        return;
    }
    if ((int)blockIdx.x >= p.tiles_m * p.tiles_n && blockIdx.y) return;   // column-sum riders run once
    gemm_s32_body<AKC, BKC, NW, RST>(p, blockIdx.x, lds, 0, 0, fr.mc.d1 ? &fr.mc : nullptr, nullptr, 0, blockIdx.y, &fr, &ep);
}
template <bool AKC, bool BKC, int NW, bool RST>
void launch_l32(const GemmP &p, const ActEpi &ep, const FoldRider &fr, dim3 grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_l32<AKC, BKC, NW, RST>), hipFuncAttributeMaxDynamicSharedMemorySize, NW * 16384); attr_done = true; }
    T4K_LAUNCH((k_gemm_l32<AKC, BKC, NW, RST>), grid, dim3(64 * NW), (size_t)NW * 16384, s, p, ep, fr);
}
// dW += dY^T X (+ dB rider) and dX = dY W of one linear layer on 32x32 tiles (see k_gemm_dual for the gate): NW waves, NW x 16 KiB of dynamic LDS.  More workgroups than resident slots are
// fine here although the dX writers spin on the dW readers' slots: a writer's readers all have LOWER workgroup ids, each XCD dispatches its
// workgroups in id order and a reader never waits - so every reader is running or done before the first writer of its XCD takes a slot
// (the same dispatch-order argument as the conv stack's band exchange; the wait is bounded and reported anyway).
template <bool RST, int NW = 4>
__global__ void __launch_bounds__(64 * NW) k_gemm_dual_l32(GemmP p1, GemmP p2, int nb1, int t1, int t2, unsigned *slots, unsigned epoch, MaskChain mc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x < nb1) gemm_s32_body<false, false, NW, RST>(p1, blockIdx.x, lds, slots ? 1 : 0, t1, nullptr, slots, epoch);
    else                       gemm_s32_body<true, false, NW, RST>(p2, (int)blockIdx.x - nb1, lds, slots ? 2 : 0, t1, &mc, slots, epoch);
}

// ---- classifier-head backward + the backward of the linear layer in front of it, ONE launch (t4k_mlp_head_bwd).
// The head backward (loss preparation out -= target, dW2 | dB2, dX2 = dY2 W2 in place, mask multiply -> dY1) and the big layer's dW1 += dY1^T X1,
// dX1 = dY1 W1 are dependent: the second needs dY1 [N][EA].  But dY1 is CHEAP to recompute - dY1[n, e] = mask[n, e] sum_j (P - T)[n, j] W2[j, e], EB <= 16
// terms - so every GEMM tile prepares the 32 rows / columns of dY1 it multiplies with in LDS from P, T, W2 and the mask (all there before the launch) and
// nothing waits for the head: GEMM tiles and the column-sliced head workgroups (k_linsmall_bwd_cols's body as riders) run side by side.  The riders store
// dY1, dX2, dW2, dB2 and dB1 (the column sums of the dY1 slice they have in hand).  The only shared write is `out -= target` over P: a rider of its own
// stores it once EVERY workgroup has its P and T values in registers (counter; off everybody's critical path).
// Round 6 (k_head_bwd_l32): round 3's form of this launch (16.2 us) lost to two launches (7.0 + 6.0 us) because its tiles made three dependent memory round
// trips in front of the GEMM and gathered the B operand row by row.  Here a tile workgroup
//   1. issues the LDS-DMA of its B blocks (X1 or W1: 32 k x 32 columns per block, whole 128-byte runs, wave-private slots as gemm_s32_body<.., DMA>),
//   2. requests P, T, W2 and the mask values it needs STRAIGHT INTO REGISTERS in MFMA operand layout - the same round trip as the DMA -
//   3. multiplies (P - T) W2 on the matrix cores (ceil(EB / 2) v_mfma_f32_32x32x2_f32 per 32 x 32 block of dY1), applies the mask and leaves the block in LDS
//      k-major with a pitch of 33 floats (conflict-free for the writes of either tile kind and for the fragment reads),
//   4. one barrier, then the K loop proper: B fragments from the wave's DMA slots, A fragments from the dY1 tile, k-groups meet in LDS, epilogue as
//      gemm_s32_body's (in-place dX1 behind the epoch-tagged arrival slots of the dW1 readers).
// The tiles' dY1 comes from an MFMA sum, the riders' stored dY1 from the oracle's fmaf chain: the two differ by rounding only (1e-7 relative to the
// largest term), far inside the 1e-4 bar of dW1 / dX1; every tensor the reference materialises is the riders' (bit-equal to the two-launch path).
struct HeadBwd {
    const float *P, *T, *W2, *MASK;     // softmax output [N][EB], target, W2 [EB][EA], derivative mask of the layer between the linear layers [N][EA]
    float *X2, *DW2, *DB2, *Y1, *Y2;    // head input [N][EA] (receives dX2), gradients, dY1 tensor (= dX2 * mask), second copy of out - target
    float *DB1;
    int N, EA, EB, train, nwg; int *sync;
    const float *MSKB; float *Y1B;     // a second mask layer between the linear layers (`leakyrelu dropout`): dY1 = dX2 * MASK * MSKB, Y1 keeps the first product, Y1B the second
    MaskChain mc1;                      // mask multiplies behind the big layer's dX1 (the run in front of it), as linear_bwd_dual
    const float *Z;                     // 4 KiB of zeros: source of DMA lanes past the K range / the matrix edge
};
constexpr int HB_CW = 4;                // columns of the head's input per rider workgroup (k_linsmall_bwd_cols: LSC_CW)
constexpr int HB_AP = 33;               // pitch of the dY1 tile in LDS
// workgroups [0, a1): dW1 tiles (read X1, never wait) | [a1, a1 + nr): riders - the store rider, then the column riders | the rest: dX1 tiles (write X1 in place,
// wait for the dW1 tiles of their column only).  slots == nullptr: a frozen layer (a1 == 0) - nothing to wait for.
__global__ void __launch_bounds__(256) k_head_bwd_l32(GemmP p1, GemmP p2, int a1, int nr, unsigned *slots, unsigned *pslots, unsigned epoch, HeadBwd hb, int lab) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef __attribute__((address_space(3))) const float lds_f;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, l31 = lane & 31;
    const int N = hb.N, EA = hb.EA, EB = hb.EB, bx = blockIdx.x;
#ifdef T4K_LAB
#define HB_LAB(bit) (lab & (bit))
#else
#define HB_LAB(bit) 0
#endif
    if (bx < a1 || bx >= a1 + nr) {
        // ---------------------------------------------------------------- GEMM tile
        if (HB_LAB(2)) return;
        const bool first = bx < a1;
        const int tile = first ? bx : bx - a1 - nr, tiles_n = p1.tiles_n;        // both GEMMs have the same column tiling (E1)
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n, m0 = tm * 32, n0 = tn * 32;
        const int M = first ? EA : N, Nn = p2.N, K = first ? N : EA;   // dW1: M = EA, K = N;  dX1: M = N, K = EA;  both: B = [K][Nn] rows (X1 / W1)
        const float *pB = first ? p1.B : p2.B;
        float *pO = first ? p1.O : p2.O;
        const int nblk = (K + 31) >> 5;
        const int b0 = w * nblk / 4, nb = (w + 1) * nblk / 4 - b0;   // this wave's k blocks (nblk <= 8: at most two, both requested up front)
        float *Bs = lds + w * 2048, *Ad = lds + 8192;                // wave-private B slots (2 x 4 KiB; the wave's partial accumulators afterwards) | dY1 tile [nblk * 32][33]
        {   // 1. B blocks by LDS-DMA
            const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)Bs;
            const int i8 = lane >> 3, i7 = lane & 7;
            const bool bcol = n0 + 4 * i7 < Nn;
            const float *zsrc = hb.Z + 4 * i7;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (s < nb) {
                    const int k0 = 32 * (b0 + s);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int r = 8 * j + i8;
                        const float *sb = (k0 + r < K && bcol) ? pB + (long)(k0 + r) * Nn + n0 + 4 * i7 : zsrc;
                        const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((s * 1024 + j * 256) * 4));
                        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(sb), "s"(la) : "memory");
                    }
                }
            }
        }
        // epilogue operands, requested with everything else
        const int gn = n0 + l31;
        float oprev[4], mk1[4], mk2[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = 4 * w + q, gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool ok = gm < M && gn < Nn;
            const long z = (long)gm * Nn + gn;
            oprev[q] = (first && ok) ? pO[z] : 0.f;
            mk1[q] = (!first && hb.mc1.d1 && ok) ? hb.mc1.m1[z] : 0.f;
            mk2[q] = (!first && hb.mc1.d2 && ok) ? hb.mc1.m2[z] : 0.f;
        }
        // 2. operands of the dY1 blocks this wave prepares (blocks w and w + 4 of the tile's k range), in MFMA layout
        //    dW1 tile: block = 32 samples, the tile's 32 columns e of dY1;   dX1 tile: block = 32 columns e, the tile's 32 samples
        float av[2][8], bv[2][8], mk[2][16];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int blk = w + 4 * t;
            const int arow = first ? 32 * blk + l31 : m0 + l31;      // sample of this lane's A values (P - T)
            const int bcl = first ? m0 + l31 : 32 * blk + l31;       // column e of this lane's B values (W2)
            const bool on = blk < nblk, aok = on && arow < N, bok = on && bcl < EA;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const int j = 2 * s + h;
                const bool jk = j < EB;
                if (t == 1 && !first)  av[t][s] = av[0][s];          // the same 32 samples for every column block
                else av[t][s] = (aok && jk && !HB_LAB(8)) ? hb.P[(long)arow * EB + j] - hb.T[(long)arow * EB + j] : 0.f;
                if (t == 1 && first)   bv[t][s] = bv[0][s];
                else bv[t][s] = (bok && jk && !HB_LAB(8)) ? hb.W2[(long)j * EA + bcl] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ro = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int n = first ? 32 * blk + ro : m0 + ro, e = first ? m0 + l31 : 32 * blk + l31;
                const bool ok = on && n < N && e < EA;
                const long zo = (long)n * EA + e;
                mk[t][r] = (ok && !HB_LAB(64)) ? (hb.MSKB ? hb.MASK[zo] * hb.MSKB[zo] : hb.MASK[zo]) : 0.f;
            }
        }
        // 3. dY1 blocks on the matrix cores -> LDS
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int blk = w + 4 * t;
            if (blk < nblk) {
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; r++) d[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 8; s++) if (2 * s < EB) d = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][s], bv[t][s], d, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int ro = (r & 3) + 8 * (r >> 2) + 4 * h;
                    Ad[first ? (32 * blk + ro) * HB_AP + l31 : (32 * blk + l31) * HB_AP + ro] = d[r] * mk[t][r];
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the wave's DMA blocks have landed (they were requested first)
        const bool early = slots && a1 <= 128;                       // per-wave arrival slots: X1 is consumed as far as this wave is concerned
        if (first && early && lane == 0) __hip_atomic_store(slots + 4 * tile + w, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (tid == 0 && !HB_LAB(4)) __hip_atomic_store(pslots + bx, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // P and T are in registers everywhere: `out -= target` may land as far as this workgroup is concerned
        if (HB_LAB(16)) return;
        // in-place dX1: the first look at the arrival slots of this column's dW1 tiles goes out here - its round trip passes under the K loop
        const int gper = early ? 4 : 1, gslot = (a1 / tiles_n) * gper;
        unsigned gbad = 0;
        if (!first && slots && w == 0 && !HB_LAB(32)) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = lane + 64 * u;
                if (i < gslot) { const int e0t = i / gper, ww = i - e0t * gper; gbad |= __hip_atomic_load(slots + gper * (e0t * tiles_n + tn) + ww, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ epoch; }
            }
        }
        if (first && slots && !early && tid == 0) __hip_atomic_store(slots + tile, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // 4. K loop
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (s < nb) {
                lds_f *b = (lds_f *)Bs + s * 1024, *a = (lds_f *)Ad + 32 * (b0 + s) * HB_AP;
                float fa[4][4], fb[4][4];
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int j = 0; j < 4; j++) { fa[c][j] = a[(8 * c + 4 * h + j) * HB_AP + l31]; fb[c][j] = b[(8 * c + 4 * h + j) * 32 + l31]; }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][2], fb[c][2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][3], fb[c][3], acc1, 0, 0, 0);
                }
            }
        }
        // the four k-groups meet in LDS (each wave's partial sums over its own, consumed, B slots)
#pragma unroll
        for (int r = 0; r < 16; r++) Bs[r * 64 + lane] = acc0[r] + acc1[r];
        if (!first && slots && w == 0 && !HB_LAB(32) && !__all(gbad == 0)) {      // the dW1 tiles of this column have consumed X1 (see gemm_s32_body gate_mode 2); a1 <= 512 slots
            for (int spin_it = 0;; spin_it++) {
                if (spin_it > T4K_SPIN_MAX) { if (lane == 0 && g_spin_err_dev) __hip_atomic_store(g_spin_err_dev, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                unsigned bad = 0;
#pragma unroll 4
                for (int i = lane; i < gslot; i += 64) {
                    const int e0t = i / gper, ww = i - e0t * gper;
                    bad |= __hip_atomic_load(slots + gper * (e0t * tiles_n + tn) + ww, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ epoch;
                }
                if (__all(bad == 0)) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = 4 * w + q, gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = lds[r * 64 + lane];
#pragma unroll
            for (int g = 1; g < 4; g++) v += lds[g * 2048 + r * 64 + lane];       // k-groups in order
            if (gm < M && gn < Nn) {
                const long z = (long)gm * Nn + gn;
                if (first) pO[z] = v + oprev[q];                                  // dW1 accumulates (beta = 1)
                else {
                    pO[z] = v;
                    if (hb.mc1.d1) { const float g1 = v * mk1[q]; hb.mc1.d1[z] = g1; if (hb.mc1.d2) hb.mc1.d2[z] = g1 * mk2[q]; }
                }
            }
        }
        return;
    }
    if (bx == a1) {
        // ---------------------------------------------------------------- store rider: `out -= target` in place (+ its copy), once every other workgroup holds P and T
        const int tot = N * EB;
        for (int i = tid; i < tot; i += 256) lds[i] = hb.P[i] - hb.T[i];
        if (w == 0 && !HB_LAB(4 | 1 | 2 | 16)) {         // (ablations that drop arrivals must drop the wait too)
            // every other workgroup tags its slot once its P / T values sit in registers / LDS: plain stores to separate words (an arrival COUNTER serialised
            // 270 agent-scope atomics on one address: +1.2 us on the launch), polled here with the 64 lanes' loads in flight together
            for (int spin_it = 0;; spin_it++) {
                if (spin_it > T4K_SPIN_MAX) { if (lane == 0 && g_spin_err_dev) __hip_atomic_store(g_spin_err_dev, 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                unsigned bad = 0;
#pragma unroll 4
                for (int i = lane; i < hb.nwg; i += 64) if (i != bx) bad |= __hip_atomic_load(pslots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ epoch;
                if (__all(bad == 0)) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        float *Pw = const_cast<float *>(hb.P);
        for (int i = tid; i < tot; i += 256) { const float v = lds[i]; Pw[i] = v; if (hb.Y2) hb.Y2[i] = v; }
        return;
    }
    // -------------------------------------------------------------------- column riders: k_linsmall_bwd_cols's slice of HB_CW columns of the head's input
    if (HB_LAB(1)) return;
    constexpr int CW = HB_CW, ZI = 8;
    const int cb = bx - a1 - 1, c0 = cb * CW, cw = min(CW, EA - c0);
    float *dys = lds, *Ws = dys + N * EB, *Xs = Ws + EB * CW, *red = Xs + N * CW;       // dY2 [N][EB], W2 slice [EB][CW], X2 slice [N][CW] (then the dY1 slice), partial sums [4][EB * CW] / [32][CW]
    float mk[ZI], mkb[ZI];
#pragma unroll
    for (int k = 0; k < ZI; k++) {
        const int z = tid + k * 256, n = z / CW, c = z - n * CW;
        const bool ok = z < N * CW && c < cw;
        const long o = (long)n * EA + c0 + c;
        mk[k] = ok ? hb.MASK[o] : 0.f; mkb[k] = (ok && hb.MSKB) ? hb.MSKB[o] : 0.f;
    }
    {   // staging: every global load goes out before the first LDS store (one round trip)
        constexpr int PRE = 6;
        float pd[PRE], pt[PRE], px[PRE], pw = 0.f;
#pragma unroll
        for (int q = 0; q < PRE; q++) { const int i = tid + q * 256; const bool ok = i < N * EB; pd[q] = ok ? hb.P[i] : 0.f; pt[q] = ok ? hb.T[i] : 0.f; }
        if (tid < EB * CW) { const int j = tid / CW, c = tid - j * CW; pw = c < cw ? hb.W2[(long)j * EA + c0 + c] : 0.f; }
#pragma unroll
        for (int q = 0; q < PRE; q++) { const int i = tid + q * 256, n = i / CW, c = i - n * CW; px[q] = (hb.train && i < N * CW && c < cw) ? hb.X2[(long)n * EA + c0 + c] : 0.f; }
#pragma unroll
        for (int q = 0; q < PRE; q++) { const int i = tid + q * 256; if (i < N * EB) dys[i] = pd[q] - pt[q]; }
        if (tid < EB * CW) Ws[tid] = pw;
        if (hb.train) {
#pragma unroll
            for (int q = 0; q < PRE; q++) { const int i = tid + q * 256; if (i < N * CW) Xs[i] = px[q]; }
        }
        for (int i = tid + PRE * 256; i < N * EB; i += 256) dys[i] = hb.P[i] - hb.T[i];
        if (hb.train)
            for (int i = tid + PRE * 256; i < N * CW; i += 256) { const int n = i / CW, c = i - n * CW; Xs[i] = c < cw ? hb.X2[(long)n * EA + c0 + c] : 0.f; }
    }
    __syncthreads();
    if (tid == 0 && !HB_LAB(4)) __hip_atomic_store(pslots + bx, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // P and the target are staged here
    const int nout = EB * CW, G = min(4, 256 / nout);               // thread groups splitting the batch of one dW2 output (contiguous ranges, summed in order)
    float dwacc = 0.f;
    if (hb.train && tid < nout * G) {
        const int g = tid / nout, t = tid - g * nout, j = t / CW, c = t - j * CW;
        const int nbt = (N + G - 1) / G, n0 = g * nbt, n1 = min(N, n0 + nbt);
#pragma unroll 8
        for (int n = n0; n < n1; n++) dwacc = fmaf(dys[n * EB + j], Xs[n * CW + c], dwacc);
        if (G > 1) red[g * nout + t] = dwacc;
    }
    __syncthreads();                                                 // the X2 slice is consumed: its place takes the dY1 slice
    {
        float g1v[ZI];
#pragma unroll
        for (int k = 0; k < ZI; k++) {                               // dX2[n, c0 + c] over X2 in place: fmaf chain ascending j (the oracle's order); dY1 = dX2 * mask
            const int z = tid + k * 256, n = z / CW, c = z - n * CW;
            g1v[k] = 0.f;
            if (z >= N * CW || c >= cw) continue;
            float acc = 0.f;
            for (int j = 0; j < EB; j++) acc = fmaf(dys[n * EB + j], Ws[j * CW + c], acc);
            const long o = (long)n * EA + c0 + c;
            hb.X2[o] = acc;
            float g1 = acc * mk[k]; hb.Y1[o] = g1;
            if (hb.MSKB) { g1 *= mkb[k]; hb.Y1B[o] = g1; }
            g1v[k] = g1;
        }
#pragma unroll
        for (int k = 0; k < ZI; k++) { const int z = tid + k * 256; if (z < N * CW) Xs[z] = g1v[k]; }
    }
    __syncthreads();
    if (hb.train) {
        if (tid < nout) {
            const int j = tid / CW, c = tid - j * CW;
            float a = dwacc;
            for (int g = 1; g < G; g++) a += red[g * nout + tid];
            if (c < cw) hb.DW2[(long)j * EA + c0 + c] += a;
        }
        __syncthreads();                                             // red is free again
        {   // dB1[c0 + c] = sum_n dY1[n, c0 + c] (k_dlinear_db nmath.cu:274-280): 64 row groups per column, then the groups in order
            const int c = tid & (CW - 1), g = tid >> 2;
            float b = 0.f;
#pragma unroll 4
            for (int n = g; n < N; n += 64) b += Xs[n * CW + c];
            red[g * CW + c] = b;
        }
        __syncthreads();
        if (tid < CW) {
            float b = 0.f;
#pragma unroll 8
            for (int g = 0; g < 64; g++) b += red[g * CW + tid];
            if (tid < cw) hb.DB1[c0 + tid] += b;
        }
        if (cb == 0) {                                               // dB2[j] = sum_n dY2[n, j]: 16 row groups per output
            __syncthreads();
            const int j = tid & 15, g = tid >> 4;
            float b = 0.f;
            if (j < EB) {
#pragma unroll 4
                for (int n = g; n < N; n += 16) b += dys[n * EB + j];
            }
            red[g * 16 + j] = b;
            __syncthreads();
            if (tid < EB) {
                float t = 0.f;
#pragma unroll
                for (int g2 = 0; g2 < 16; g2++) t += red[g2 * 16 + tid];
                hb.DB2[tid] += t;
            }
        }
    }
}

// Two independent 64x64-tiled GEMMs in ONE launch (a linear layer's dW += dY^T X and dX = dY W): workgroups [0, nb1) run the
// first, the rest the second.  When the second overwrites an operand of the first (dX lands in X's buffer, backprop.cu:240)
// its stores wait on an arrival counter; every workgroup is resident (grid <= CU count), so the wait cannot deadlock.
// F1 / F2: that GEMM's K is whole 64-deep stages -> the predicate-free pipeline with loads two stages ahead (unskewed)
template <bool A1, bool B1, bool A2, bool B2, bool F1 = false, bool F2 = false>
__global__ void __launch_bounds__(256) k_gemm_dual(GemmP p1, GemmP p2, int nb1, int t1, int t2, int *gate, MaskChain mc) {
    // gate == nullptr: dX does not land in a buffer the dW half reads (no aliasing) - nothing to wait for
    if ((int)blockIdx.x < nb1) gemm_mfma_body<64, 64, 64, A1, B1, true, !F1, F1>(p1, blockIdx.x, 0, 0, gate, gate ? 1 : 0);
    else                       gemm_mfma_body<64, 64, 64, A2, B2, true, !F2, F2>(p2, (int)blockIdx.x - nb1, 0, 0, gate, gate ? 2 : 0, t1, t2, &mc);
}


// ------------------------------------------------------------------------------------------
// 64x64 interior tiles, direct-to-LDS staging: global_load_lds_dwordx4 (LDS-DMA, no VGPR round
// trip, no ds_write pass), 3 LDS stage buffers, loads two stages ahead, counted vmcnt waits and
// raw s_barrier so the DMA of later stages stays in flight across barriers.  The DMA writes LDS
// lane-linearly (wave-uniform base + lane*16 B), so the XOR swizzle of a K-contiguous operand is
// applied to the per-lane SOURCE address and undone on the ds_read_b128 side.
// 8-wave workgroups: two waves per SIMD.  Waves 4..7 take the upper half of every stage's k chunks
// into their own accumulators, so one wave's LDS reads and waits sit under the other's MFMAs; the halves are summed
// through LDS in the epilogue (fixed order).  (Round 6: the 4-wave form of this pipeline, k_gemm_glds, and the variant switch that selected it are gone.)
//
// RAGK: K need not be a whole number of stages.  The last, partial stage is one more DMA stage whose source addresses are clamped to the
// operand's last valid 16-byte group / row (so every lane still moves 16 bytes and the vmcnt bookkeeping is unchanged); the 8-deep chunks
// past the tail are never read, and in a partial last chunk the k positions past the tail are zeroed in registers on BOTH operands
// (0 x stale-LDS garbage could be NaN).  The tail's chunks alternate between the two k-groups.  K = 784 (a 28 x 28 image row, the GAN
// layer width) = 6 stages of 128 + 16: this kernel instead of the predicated register-staged one.
template <int BK, bool AKC, bool BKC, bool PRIO = false, bool RAGK = false>   // PRIO: s_setprio around the MFMA burst - measured +1.2 us at 1024^3, kept off
__global__ void __launch_bounds__(512) k_gemm_glds8(GemmP p) {
    constexpr int BM = 64, BN = 64;
    constexpr int NC = BK / 8, CH = BK / 4, SW = (64 / BK) > 0 ? (64 / BK) : 1;
    constexpr int NST = (BK >= 128) ? 2 : 3;         // stage buffers: a 128-deep stage is 64 KiB, two of them fit (half the barriers per K)
    constexpr int STAGE = (BM + BN) * BK;          // floats per stage buffer
    constexpr int NI = BK / 4;                     // 1-KiB DMA instructions per operand per stage
    constexpr int NJ = NI / 8;                     // ... per wave (8 waves)
    constexpr int NPW = 2 * NJ;                    // DMA instructions per wave per stage
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = w >> 2, w4 = w & 3;               // k-group (which half of every stage's chunks), wave within the 2x2 tile grid
    const int wm = w4 >> 1, wn = w4 & 1, h = lane >> 5, l31 = lane & 31;
    constexpr int NCG = NC / 2;                       // chunks per k-group per stage
    const int c0 = kg * NCG;
    const int M = p.M, N = p.N, K = p.K;

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
    const int nst  = (kend - kbeg) / BK;
    const int tail = RAGK ? (kend - kbeg) - nst * BK : 0;      // k positions of the partial last stage
    const int nstT = nst + (tail > 0 ? 1 : 0);                 // stages the DMA pipeline moves

    // per-lane DMA source offsets (bytes from the operand base) of stage 0; the launcher guarantees both operands span < 4 GiB
    unsigned voffA[NJ], voffB[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int i = w * NJ + j;
        // rows / 4-column groups past a ragged edge are clamped to the last valid one (valid memory; they only feed accumulator rows /
        // columns the epilogue never stores), so N = 1000 classes or a 784-wide image row run this kernel like interior tiles
        if (AKC) { const int r = i * (256 / BK) + lane / CH, ql = lane % CH, q = ql ^ ((r / SW) & (CH - 1));
                   voffA[j] = (unsigned)(((long)min(m0 + r, M - 1) * K + kbeg + q * 4) * 4); }
        else     { const int kk = i * 4 + lane / 16, ch = lane % 16;
                   voffA[j] = (unsigned)(((long)(kbeg + kk) * M + min(m0 + ch * 4, M - 4)) * 4); }
        if (BKC) { const int r = i * (256 / BK) + lane / CH, ql = lane % CH, q = ql ^ ((r / SW) & (CH - 1));
                   voffB[j] = (unsigned)(((long)min(n0 + r, N - 1) * K + kbeg + q * 4) * 4); }
        else     { const int kk = i * 4 + lane / 16, ch = lane % 16;
                   voffB[j] = (unsigned)(((long)(kbeg + kk) * N + min(n0 + ch * 4, N - 4)) * 4); }
    }
    const long stepA = AKC ? BK : (long)BK * M, stepB = BKC ? BK : (long)BK * N;

    // The DMA is issued from inline asm (saddr form: scalar base + each lane's fixed 32-bit byte offset).  Through the
    // __builtin_amdgcn_global_load_lds builtin hipcc books the load as "flat, may touch LDS": while one is pending it turns every LDS
    // dependency into s_waitcnt lgkmcnt(0), so the operand reads just issued for the NEXT chunk were waited for at once - one exposed LDS
    // round trip per 8 MFMAs (1024^3: 21.5 us).  Invisible to the compiler, it emits counted lgkmcnt(N) ladders instead (19.4 us); the
    // DMA's own completion is waited for by the explicit vmcnt waits in wait_next().
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        const float *ba = p.A + kt * stepA, *bb = p.B + kt * stepB;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + (w * NJ + j) * 256) * 4));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffA[j]), "s"(ba), "s"(la) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffB[j]), "s"(bb), "s"(la + BM * BK * 4) : "memory");
        }
    };
    // the partial last stage: same LDS placement (a lane's 16 bytes land at its own slot whatever the exec mask), but only the lanes /
    // instructions whose k lies inside the tail move anything - a clamped full stage would cost a full stage of L2 bandwidth for 16 columns.
    // Three-buffer pipelines count vmcnt per stage, so there every instruction is issued (sources clamped into [kb, kend)).
    auto issue_tail = [&](int buf) __attribute__((always_inline)) {
        const int kb = kbeg + nst * BK;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int i = w * NJ + j;
            unsigned va, vb; bool oka, okb;
            if (AKC) { const int r = i * (256 / BK) + lane / CH, ql = lane % CH, q = ql ^ ((r / SW) & (CH - 1));
                       oka = q * 4 < tail;
                       va = (unsigned)(((long)min(m0 + r, M - 1) * K + min(kb + q * 4, kend - 4)) * 4); }
            else     { const int kk = i * 4 + lane / 16, ch = lane % 16;
                       oka = i * 4 < tail;                                  // wave-uniform: whole instructions past the tail are skipped
                       va = (unsigned)(((long)min(kb + kk, kend - 1) * M + min(m0 + ch * 4, M - 4)) * 4); }
            if (BKC) { const int r = i * (256 / BK) + lane / CH, ql = lane % CH, q = ql ^ ((r / SW) & (CH - 1));
                       okb = q * 4 < tail;
                       vb = (unsigned)(((long)min(n0 + r, N - 1) * K + min(kb + q * 4, kend - 4)) * 4); }
            else     { const int kk = i * 4 + lane / 16, ch = lane % 16;
                       okb = i * 4 < tail;
                       vb = (unsigned)(((long)min(kb + kk, kend - 1) * N + min(n0 + ch * 4, N - 4)) * 4); }
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + i * 256) * 4));
            if (NST == 3 || oka) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(va), "s"(p.A), "s"(la) : "memory");
            if (NST == 3 || okb) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(vb), "s"(p.B), "s"(la + BM * BK * 4) : "memory");
        }
    };
    auto issue_any = [&](int kt, int buf) __attribute__((always_inline)) {
        if (!RAGK || kt < nst) issue(kt, buf); else issue_tail(buf);
    };

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }

    const int ra_ = wm * 32 + l31, rb_ = wn * 32 + l31;
    // operand fragments of one 8-deep k chunk (lane half h holds k = 8*ci + 4*h + {0..3})
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        if (AKC) { const v4f t = *reinterpret_cast<const v4f *>(a + ra_ * BK + (((ci * 2 + h) ^ ((ra_ / SW) & (CH - 1))) << 2));
                   av[0] = t[0]; av[1] = t[1]; av[2] = t[2]; av[3] = t[3]; }
        else {
#pragma unroll
            for (int j = 0; j < 4; j++) av[j] = a[(ci * 8 + 4 * h + j) * BM + ra_];
        }
        if (BKC) { const v4f t = *reinterpret_cast<const v4f *>(b + rb_ * BK + (((ci * 2 + h) ^ ((rb_ / SW) & (CH - 1))) << 2));
                   bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3]; }
        else {
#pragma unroll
            for (int j = 0; j < 4; j++) bv[j] = b[(ci * 8 + 4 * h + j) * BN + rb_];
        }
    };
    auto mm = [&](float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        if (PRIO) __builtin_amdgcn_s_setprio(2);                 // keep the matrix pipe fed: the MFMA burst outranks the other wave's loads
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                       // keep the two accumulators alternating (hipcc pairs them otherwise)
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], acc1, 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto wait_next = [&](bool more) __attribute__((always_inline)) {      // next stage landed; later one may fly
        if (more) { if (NPW == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                    else          asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); }
        else        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    if (nstT > 0) issue_any(0, 0);
    if (NST == 3 && nstT > 1) issue_any(1, 1);
    wait_next(NST == 3 && nstT > 1);

    // Software pipeline: the operands of chunk c+1 are read BEFORE the MFMAs of chunk c are issued,
    // and the MFMAs of a stage's last chunk are issued after the barrier, behind the first reads of
    // the next stage - the matrix pipe never waits on an LDS round trip.
    float ca[4], cb[4];
    if (nst > 0) rd(lds, lds + BM * BK, c0, ca, cb);
    int buf = 0;
    for (int kt = 0; kt < nst; kt++) {
        int nb = buf + 2; if (nb >= 3) nb -= 3;
        int b1 = buf + 1; if (b1 >= NST) b1 = 0;
        if (NST == 3) { if (kt + 2 < nstT) issue_any(kt + 2, nb); }      // overwrites the buffer read in stage kt-1
        else          { if (kt + 1 < nstT) issue_any(kt + 1, b1); }      // two buffers: the other one was read in stage kt-1
        const float *a = lds + buf * STAGE, *b = a + BM * BK;
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
        wait_next(NST == 3 && kt + 2 < nstT);               // all my reads of stage kt done; stage kt+1 visible
        float na[4], nbv[4];
        if (kt + 1 < nst) rd(lds + b1 * STAGE, lds + b1 * STAGE + BM * BK, c0, na, nbv);
        __builtin_amdgcn_sched_barrier(0);
        mm(ca, cb);
        if (kt + 1 < nst) {
#pragma unroll
            for (int j = 0; j < 4; j++) { ca[j] = na[j]; cb[j] = nbv[j]; }
        }
        buf = b1;
    }
    if (RAGK && tail > 0) {                                  // the partial stage sits in `buf`, visible since the last barrier
        const float *a = lds + buf * STAGE, *b = a + BM * BK;
        const int nct = (tail + 7) >> 3;
        // k positions past the tail are zeroed when the fragment is handed to the MFMAs (not at the read: that would wait for the LDS round trip)
        auto zf = [&](int c, float (&dv)[4], float (&ev)[4], const float (&av)[4], const float (&bv)[4]) __attribute__((always_inline)) {
            const int k0 = c * 8 + 4 * h;
#pragma unroll
            for (int j = 0; j < 4; j++) { const bool in = k0 + j < tail; dv[j] = in ? av[j] : 0.f; ev[j] = in ? bv[j] : 0.f; }
        };
        int c = kg;                                          // its chunks alternate between the k-groups; same read-ahead as the main loop
        float ra4[4], rb4[4];
        if (c < nct) { rd(a, b, c, ra4, rb4); zf(c, ca, cb, ra4, rb4); }
        for (; c < nct; c += 2) {
            if (c + 2 < nct) rd(a, b, c + 2, ra4, rb4);
            __builtin_amdgcn_sched_barrier(0);
            mm(ca, cb);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < nct) zf(c + 2, ca, cb, ra4, rb4);
        }
    }

    const float alpha = p.alpha, beta = p.beta;
    const int gn = n0 + wn * 32 + l31;
    float add[16];
    // the two k-groups meet in LDS (the stage buffers are free now): group 1 parks its 32x32 blocks, group 0 adds them
    __syncthreads();
    if (kg == 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) lds[(w4 * 16 + r) * 64 + lane] = acc0[r] + acc1[r];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int r = 0; r < 16; r++) add[r] = lds[(w4 * 16 + r) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = (acc0[r] + acc1[r]) + add[r];
        if (gm >= M || gn >= N) continue;                   // ragged edge tile
        if (p.nsplit > 1) p.part[((long)blockIdx.y * M + gm) * N + gn] = v;
        else {
            const long z = (long)gm * N + gn;
            float o = v * alpha;
            if (beta != 0.f) o += p.O[z] * beta;
            if (p.bias) o += p.bias[gn];
            __builtin_nontemporal_store(o, &p.O[z]);        // streaming store: no dirty L2 lines left for the kernel-end write-back (-0.27 us at 1024^3)
        }
    }
}


// The plain product O = A @ B (word `matmul`, tA = tB = 0, alpha = 1, beta = 0, no bias, interior 64x64 tiles, K % 128 == 0) has its own
// copy of the 8-wave kernel with nothing else in it: the same loop inside the general template above measures 20.1 us at 1024^3, this
// one 19.4 (tools/gemm_lab.hip: every variant with in-kernel cycle stamps; the loop is sensitive to the code around it).
#ifndef T4K_GEMM_EARLY_ISSUE
#define T4K_GEMM_EARLY_ISSUE 1      // k_gemm_plain128: stage kt + 2 requested right behind stage kt's closing barrier (0: at the top of stage kt + 1; 2048^2 x 1024: 67.7 -> 66.7 us).
#endif                              // Measured and dropped: the same in k_gemm_nn_plain (19.1 -> 19.9 us at 1024^3), and the DMA instructions spread between the MFMAs of either kernel (-2..3 %)
struct PlainP { const float *A, *B; float *O; int M, N, K; float alpha, beta; const float *bias; int *sync; float *part; int swap; };   // swap: tile order with the roles of M and N exchanged (lab: T4K_GEMM_TT_SWAP)
// The other operand layouts (word `matmul` on transposed views; Tensor::linear tensor.cu:79-87 = X @ W^T + b with alpha / beta) get the same
// lean kernel instead of the general template: AKC / BKC pick the operand layout (K-contiguous rows, read with ds_read_b128 through the
// XOR swizzle, or k-major rows read per k), EPI adds alpha / beta / bias to the store.  Interior 64x64 tiles, K % 128 == 0, unsplit.
// PAIR: an output of 65..128 tiles would leave half the CUs idle (512 x 1024 x 1024: 128 tiles).  gridDim.y = 2 workgroups share a tile,
// one K half each; the first to finish parks its 64x64 partial in the workspace and raises a flag, the second adds it and stores -
// a + b == b + a, so the result does not depend on who arrives first and no fold launch follows.  The parker never waits, so the
// waiter's (bounded) spin cannot deadlock whatever the residency.
// RAGK: K >= 128 with a partial last stage (784 = 6 x 128 + 16), see k_gemm_glds8: the tail is one more DMA stage in which only the lanes whose
// k lies inside the tail move anything, its 8-deep chunks alternate between the k-groups, positions past the tail are zeroed in registers.
template <bool POW2, bool AKC = true, bool BKC = false, bool EPI = false, bool PAIR = false, bool RAGK = false>
__global__ void __launch_bounds__(512) k_gemm_nn_plain(PlainP p) {
    constexpr int BM = 64, BN = 64, BK = 128;
    constexpr int NC = BK / 8, CH = BK / 4;
    constexpr int STAGE = (BM + BN) * BK, NI = BK / 4, NJ = NI / 8, NCG = NC / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = w >> 2, w4 = w & 3, wm = w4 >> 1, wn = w4 & 1, h = lane >> 5, l31 = lane & 31;
    const int c0 = kg * NCG;
    const int M = p.M, N = p.N, K = p.K;
    const int tiles_m = M / BM, tiles_n = N / BN, T = tiles_m * tiles_n;
    int tm, tn;
    if (POW2) {                                             // power-of-two tile grid with T % 32 == 0 (1024^2: 16 x 16): the XCD-aware tile order in shifts -
        const int tnb = 31 - __builtin_clz(p.swap ? tiles_m : tiles_n), b = blockIdx.x;     // the general form below costs three integer divisions (~400 cycles) before the first DMA
        const int L = (b & 7) * (T >> 3) + (b >> 3), pg = 2 + tnb, r = L & ((1 << pg) - 1);
        const int u = ((L >> pg) << 2) + (r & 3), v = r >> 2;
        tm = p.swap ? v : u; tn = p.swap ? u : v;
    } else {
        int L;
        {
            const int b = blockIdx.x, q8 = T >> 3, r8 = T & 7, x = b & 7, i = b >> 3;
            L = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i;
        }
        constexpr int GROUP_M = 4;
        const int per_group = GROUP_M * tiles_n;
        const int grp = L / per_group, first_m = grp * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        tm = first_m + (L % per_group) % gsz; tn = (L % per_group) / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN, nst = PAIR ? (K >> 1) / BK : K / BK;
    const int kt0 = PAIR ? (int)blockIdx.y * nst : 0;      // first stage of this workgroup's K half
    const int tail = RAGK ? K - nst * BK : 0;
    unsigned voffA[NJ], voffB[NJ];
    {
        const int r0 = w * 8 + (lane >> 5), ql = lane & 31, kk0 = w * 16 + (lane >> 4);
        // one multiply per operand: a K-contiguous operand's instruction j covers rows r0 + 2j (swizzled 16-byte groups), a k-major one's k rows kk0 + 4j
        const unsigned ba = AKC ? (unsigned)((m0 + r0) * K) * 4u : (unsigned)(kk0 * M + m0 + (lane & 15) * 4) * 4u;
        const unsigned bb = BKC ? (unsigned)((n0 + r0) * K) * 4u : (unsigned)(kk0 * N + n0 + (lane & 15) * 4) * 4u;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            voffA[j] = AKC ? ba + (unsigned)(2 * j * K) * 4u + (unsigned)((ql ^ ((r0 + 2 * j) & 31)) << 4) : ba + (unsigned)(4 * j * M) * 4u;
            voffB[j] = BKC ? bb + (unsigned)(2 * j * K) * 4u + (unsigned)((ql ^ ((r0 + 2 * j) & 31)) << 4) : bb + (unsigned)(4 * j * N) * 4u;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        const long ks = kt0 + kt;
        const float *ba = p.A + (AKC ? ks * BK : ks * BK * M), *bb = p.B + (BKC ? ks * BK : ks * BK * N);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + (w * NJ + j) * 256) * 4));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffA[j]), "s"(ba), "s"(la) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffB[j]), "s"(bb), "s"(la + BM * BK * 4) : "memory");
        }
    };
    auto issue_tail = [&](int buf) __attribute__((always_inline)) {        // stage nst: same lane offsets, lanes past the tail switched off (nothing read out of bounds)
        const float *ba = p.A + (AKC ? (long)nst * BK : (long)nst * BK * M), *bb = p.B + (BKC ? (long)nst * BK : (long)nst * BK * N);
        const int r0 = w * 8 + (lane >> 5), ql = lane & 31, kk0 = w * 16 + (lane >> 4);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + (w * NJ + j) * 256) * 4));
            const bool kin = ((ql ^ ((r0 + 2 * j) & 31)) << 2) < tail, rin = kk0 + 4 * j < tail;
            if (AKC ? kin : rin) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffA[j]), "s"(ba), "s"(la) : "memory");
            if (BKC ? kin : rin) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffB[j]), "s"(bb), "s"(la + BM * BK * 4) : "memory");
        }
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int ra_ = wm * 32 + l31, rb_ = wn * 32 + l31;
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[4], float (&bv)[4]) __attribute__((always_inline)) {
        if (AKC) { const v4f t = *reinterpret_cast<const v4f *>(a + ra_ * BK + (((ci * 2 + h) ^ (ra_ & (CH - 1))) << 2));
                   av[0] = t[0]; av[1] = t[1]; av[2] = t[2]; av[3] = t[3]; }
        else {
#pragma unroll
            for (int j = 0; j < 4; j++) av[j] = a[(ci * 8 + 4 * h + j) * BM + ra_];
        }
        if (BKC) { const v4f t = *reinterpret_cast<const v4f *>(b + rb_ * BK + (((ci * 2 + h) ^ (rb_ & (CH - 1))) << 2));
                   bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3]; }
        else {
#pragma unroll
            for (int j = 0; j < 4; j++) bv[j] = b[(ci * 8 + 4 * h + j) * BN + rb_];
        }
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
    float ca[4], cb[4];
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    rd(lds, lds + BM * BK, c0, ca, cb);
    int buf = 0;
    for (int kt = 0; kt < nst; kt++) {
        const int b1 = buf ^ 1;
        if (kt + 1 < nst) issue(kt + 1, b1);
        else if (RAGK && tail > 0) issue_tail(b1);
        const float *a = lds + buf * STAGE, *b = a + BM * BK;
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
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float na[4], nbv[4];
        if (kt + 1 < nst) rd(lds + b1 * STAGE, lds + b1 * STAGE + BM * BK, c0, na, nbv);
        __builtin_amdgcn_sched_barrier(0);
        mm(ca, cb);
        if (kt + 1 < nst) {
#pragma unroll
            for (int j = 0; j < 4; j++) { ca[j] = na[j]; cb[j] = nbv[j]; }
        }
        buf = b1;
    }
    if (RAGK && tail > 0) {                                  // the partial stage sits in `buf`, visible since the last barrier
        const float *a = lds + buf * STAGE, *b = a + BM * BK;
        const int nct = (tail + 7) >> 3;
        auto zf = [&](int c, float (&dv)[4], float (&ev)[4], const float (&av)[4], const float (&bv)[4]) __attribute__((always_inline)) {
            const int k0 = c * 8 + 4 * h;
#pragma unroll
            for (int j = 0; j < 4; j++) { const bool in = k0 + j < tail; dv[j] = in ? av[j] : 0.f; ev[j] = in ? bv[j] : 0.f; }
        };
        int c = kg;
        float ra4[4], rb4[4];
        if (c < nct) { rd(a, b, c, ra4, rb4); zf(c, ca, cb, ra4, rb4); }
        for (; c < nct; c += 2) {
            if (c + 2 < nct) rd(a, b, c + 2, ra4, rb4);
            __builtin_amdgcn_sched_barrier(0);
            mm(ca, cb);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < nct) zf(c + 2, ca, cb, ra4, rb4);
        }
    }
    const int gn = n0 + wn * 32 + l31;
    float add[16];
    // the epilogue's operands (bias, the old C of a beta != 0 product) are requested here, in front of the k-groups' meeting: behind it they
    // were a memory round trip of their own at the very end of a single-wave launch (alpha / beta at 1024^3: 69.5 % of the MFMA peak)
    float bv_ = 0.f, old[16];
    if (EPI && kg == 0) {
        bv_ = p.bias ? p.bias[gn] : 0.f;
        if (p.beta != 0.f) {
#pragma unroll
            for (int r = 0; r < 16; r++) old[r] = p.O[(long)(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * N + gn];
        }
    }
    __syncthreads();
    if (kg == 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) lds[(w4 * 16 + r) * 64 + lane] = acc0[r] + acc1[r];
    }
    __syncthreads();
    if (!PAIR && kg == 1) return;
    if (PAIR) {
        if (kg == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) add[r] = (acc0[r] + acc1[r]) + lds[(w4 * 16 + r) * 64 + lane];      // this workgroup's K half, complete
        }
        __shared__ int role_s;
        const int tile_id = tm * tiles_n + tn;
        int *ticket = p.sync + tile_id, *flag = p.sync + 2048 + tile_id;
        float *slot = p.part + (long)tile_id * (BM * BN);
        const int t4 = tid & 255;
        if (tid == 0) role_s = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        // agent-scope (cache-bypassing) stores and loads carry the payload: the two workgroups may sit on different XCDs (private L2s)
        if (role_s == 0) {
            if (kg == 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) __hip_atomic_store(&slot[r * 256 + t4], add[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid == 0) {
            T4K_SPIN_WAIT(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0, 3);
            __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc0[r] = add[r]; acc1[r] = 0.f; add[r] = __hip_atomic_load(&slot[r * 256 + t4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) add[r] = lds[(w4 * 16 + r) * 64 + lane];
    }
    if (EPI) {
        const float alpha = p.alpha, beta = p.beta;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float o = ((acc0[r] + acc1[r]) + add[r]) * alpha;
            if (beta != 0.f) o += old[r] * beta;
            if (p.bias) o += bv_;
            __builtin_nontemporal_store(o, &p.O[(long)gm * N + gn]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            __builtin_nontemporal_store((acc0[r] + acc1[r]) + add[r], &p.O[(long)gm * N + gn]);
        }
    }
}
template <bool AKC, bool BKC, bool EPI, bool PAIR = false, bool RAGK = false>
void launch_plain_(const PlainP &q, int tmq, int tnq, unsigned gx, hipStream_t s) {
    constexpr size_t lds_bytes = (size_t)2 * 128 * 128 * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nn_plain<false, AKC, BKC, EPI, PAIR, RAGK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nn_plain<true, AKC, BKC, EPI, PAIR, RAGK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_done = true;
    }
    static const int fastpro = T4K_LAB_ENV("T4K_GEMM_FASTPRO", 1);
    const bool pow2 = fastpro && (tmq & (tmq - 1)) == 0 && (tnq & (tnq - 1)) == 0 && tmq >= 4 && (tmq * tnq) % 32 == 0;
    const dim3 grid(gx, PAIR ? 2 : 1);
    if (pow2) T4K_LAUNCH((k_gemm_nn_plain<true, AKC, BKC, EPI, PAIR, RAGK>),  grid, dim3(512), lds_bytes, s, q);
    else      T4K_LAUNCH((k_gemm_nn_plain<false, AKC, BKC, EPI, PAIR, RAGK>), grid, dim3(512), lds_bytes, s, q);
}
void launch_nn_plain(const GemmP &p, dim3 grid, hipStream_t s) {
    PlainP q{ p.A, p.B, p.O, p.M, p.N, p.K, 1.0f, 0.0f, nullptr, nullptr, nullptr };
    launch_plain_<true, false, false>(q, p.M / 64, p.N / 64, grid.x, s);
}
// any layout, alpha / beta / bias: interior tiles, K % 128 == 0, unsplit - or, pair = true, K % 256 == 0 and two workgroups per tile (the caller checks)
void launch_plain_any(const GemmP &p, dim3 grid, int tA, int tB, hipStream_t s, bool pair = false, bool ragk = false) {
    PlainP q{ p.A, p.B, p.O, p.M, p.N, p.K, p.alpha, p.beta, p.bias, p.sync, p.part };
    { static const int sw = T4K_LAB_ENV("T4K_GEMM_TT_SWAP", 0); q.swap = (sw && tA && tB) ? 1 : 0; }
    const bool epi = p.alpha != 1.0f || p.beta != 0.0f || p.bias;
    const int tmq = p.M / 64, tnq = p.N / 64;
#define T4K_PL(A_, B_) do { if (ragk) { if (epi) launch_plain_<A_, B_, true, false, true>(q, tmq, tnq, grid.x, s); else launch_plain_<A_, B_, false, false, true>(q, tmq, tnq, grid.x, s); } \
                            else if (pair) { if (epi) launch_plain_<A_, B_, true, true>(q, tmq, tnq, grid.x, s); else launch_plain_<A_, B_, false, true>(q, tmq, tnq, grid.x, s); } \
                            else if (epi) launch_plain_<A_, B_, true>(q, tmq, tnq, grid.x, s); else launch_plain_<A_, B_, false>(q, tmq, tnq, grid.x, s); } while (0)
    if (!tA && !tB) T4K_PL(true, false); else if (!tA) T4K_PL(true, true); else if (!tB) T4K_PL(false, false); else T4K_PL(false, true);
#undef T4K_PL
}

// Large outputs (>= ~3/4 of the CUs in 128x128 tiles): the lean pipeline on a 128x128 tile.  Same 8 waves = 2 k-groups x 2x2 waves, but every
// wave owns a 64x64 block (2x2 accumulators of 32x32): one A fragment feeds two MFMAs and so does one B fragment - half the LDS reads and half
// the L2 -> LDS bytes per MFMA of the 64x64 tile, which is what limits that kernel once the fixed costs are amortised (2048^3: 80 %).
// 64-deep double-buffered stages (64 KiB each).  Interior tiles, K % 64 == 0, unsplit; layouts and epilogue as k_gemm_nn_plain.
// RAGK: K >= 256 with a partial last stage (784 = 12 x 64 + 16), as k_gemm_nn_plain<RAGK>: one more DMA stage in which only the lanes inside the
// tail move anything, its 8-deep chunks alternate between the k-groups, positions past the tail zeroed in registers (2048 x 2048 x 784: 58.5 -> measured below).
// BK_ = 32 (round 5): 32-deep stages, 64 KiB of LDS, at most 128 registers - TWO workgroups per CU, each with its own stage barrier, so one's MFMAs fill the pipe
// while the other's waves meet (grids of >= 2 tiles per CU; the 64-deep form keeps one workgroup per CU).
template <bool AKC, bool BKC, bool EPI, bool RAGK = false, int BK_ = 64>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(BK_ == 32 ? 4 : 2, BK_ == 32 ? 4 : 2))) k_gemm_plain128(PlainP p) {
    constexpr int BM = 128, BN = 128, BK = BK_;
    constexpr int NC = BK / 8, CH = BK / 4, NCG = NC / 2;
    constexpr int STAGE = (BM + BN) * BK;
    constexpr int NJ = (BM * BK / 256) / 8;                // 1-KiB DMA instructions per operand per wave per stage (4)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = w >> 2, w4 = w & 3, wm = w4 >> 1, wn = w4 & 1, h = lane >> 5, l31 = lane & 31;
    const int c0 = kg * NCG;
    const int M = p.M, N = p.N, K = p.K;
    const int tiles_m = M / BM, tiles_n = N / BN, T = tiles_m * tiles_n;
    int tm, tn;
    {
        int L;
        { const int b = blockIdx.x, q8 = T >> 3, r8 = T & 7, x = b & 7, i = b >> 3;
          L = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i; }           // XCD x owns a contiguous run of the tile order
        constexpr int GROUP_M = 4;
        const int per_group = GROUP_M * tiles_n;
        const int grp = L / per_group, first_m = grp * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        tm = first_m + (L % per_group) % gsz; tn = (L % per_group) / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN, nst = K / BK;
    const int tail = RAGK ? K - nst * BK : 0;
    unsigned voffA[NJ], voffB[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int i = w * NJ + j;
        constexpr int RPI = 64 / CH;                       // rows of a K-contiguous operand one 1-KiB DMA instruction covers (CH quads of 16 bytes per row)
        if (AKC) { const int r = i * RPI + lane / CH, q = (lane % CH) ^ (r & (CH - 1)); voffA[j] = (unsigned)((m0 + r) * K + q * 4) * 4u; }
        else     { const int kk = i * 2 + (lane >> 5);                                  voffA[j] = (unsigned)(kk * M + m0 + (lane & 31) * 4) * 4u; }
        if (BKC) { const int r = i * RPI + lane / CH, q = (lane % CH) ^ (r & (CH - 1)); voffB[j] = (unsigned)((n0 + r) * K + q * 4) * 4u; }
        else     { const int kk = i * 2 + (lane >> 5);                                  voffB[j] = (unsigned)(kk * N + n0 + (lane & 31) * 4) * 4u; }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        const float *ba = p.A + (AKC ? (long)kt * BK : (long)kt * BK * M), *bb = p.B + (BKC ? (long)kt * BK : (long)kt * BK * N);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + (w * NJ + j) * 256) * 4));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffA[j]), "s"(ba), "s"(la) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffB[j]), "s"(bb), "s"(la + BM * BK * 4) : "memory");
        }
    };
    auto issue_tail = [&](int buf) __attribute__((always_inline)) {        // stage nst: same lane offsets, lanes past the tail switched off (nothing read out of bounds)
        const float *ba = p.A + (AKC ? (long)nst * BK : (long)nst * BK * M), *bb = p.B + (BKC ? (long)nst * BK : (long)nst * BK * N);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int i = w * NJ + j;
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + i * 256) * 4));
            const bool kin = ((((lane % CH) ^ ((i * (64 / CH) + lane / CH) & (CH - 1)))) << 2) < tail, rin = i * 2 + (lane >> 5) < tail;
            if (AKC ? kin : rin) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffA[j]), "s"(ba), "s"(la) : "memory");
            if (BKC ? kin : rin) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffB[j]), "s"(bb), "s"(la + BM * BK * 4) : "memory");
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    const int ra_ = wm * 64 + l31, rb_ = wn * 64 + l31;
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[2][4], float (&bv)[2][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int ra = ra_ + t * 32, rb = rb_ + t * 32;
            if (AKC) { const v4f v = *reinterpret_cast<const v4f *>(a + ra * BK + (((ci * 2 + h) ^ (ra & (CH - 1))) << 2));
                       av[t][0] = v[0]; av[t][1] = v[1]; av[t][2] = v[2]; av[t][3] = v[3]; }
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) av[t][j] = a[(ci * 8 + 4 * h + j) * BM + ra];
            }
            if (BKC) { const v4f v = *reinterpret_cast<const v4f *>(b + rb * BK + (((ci * 2 + h) ^ (rb & (CH - 1))) << 2));
                       bv[t][0] = v[0]; bv[t][1] = v[1]; bv[t][2] = v[2]; bv[t][3] = v[3]; }
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) bv[t][j] = b[(ci * 8 + 4 * h + j) * BN + rb];
            }
        }
    };
    auto mm = [&](float (&av)[2][4], float (&bv)[2][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; j++) {                       // four independent accumulators per k step: no MFMA waits on its predecessor
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][j], bv[0][j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][j], bv[1][j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][j], bv[0][j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][j], bv[1][j], acc[1][1], 0, 0, 0);
        }
    };
    float ca[2][4], cb[2][4];
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#if T4K_GEMM_EARLY_ISSUE
    if (nst > 1) issue(1, 1);                              // stage kt + 2 is requested right behind stage kt's closing barrier (its buffer is free from there on)
    else if (RAGK && tail > 0) issue_tail(1);
#endif
    rd(lds, lds + BM * BK, c0, ca, cb);
    int buf = 0;
    for (int kt = 0; kt < nst; kt++) {
        const int b1 = buf ^ 1;
#if !T4K_GEMM_EARLY_ISSUE
        if (kt + 1 < nst) issue(kt + 1, b1);
        else if (RAGK && tail > 0) issue_tail(b1);
#endif
        const float *a = lds + buf * STAGE, *b = a + BM * BK;
#pragma unroll
        for (int ci = 0; ci + 1 < NCG; ci++) {
            float na[2][4], nbv[2][4];
            rd(a, b, c0 + ci + 1, na, nbv);
            __builtin_amdgcn_sched_barrier(0);
            mm(ca, cb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) { ca[t][j] = na[t][j]; cb[t][j] = nbv[t][j]; }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#if T4K_GEMM_EARLY_ISSUE
        if (kt + 2 < nst) issue(kt + 2, buf);
        else if (RAGK && tail > 0 && kt + 2 == nst) issue_tail(buf);
#endif
        float na[2][4], nbv[2][4];
        if (kt + 1 < nst) rd(lds + b1 * STAGE, lds + b1 * STAGE + BM * BK, c0, na, nbv);
        __builtin_amdgcn_sched_barrier(0);
        mm(ca, cb);
        if (kt + 1 < nst) {
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) { ca[t][j] = na[t][j]; cb[t][j] = nbv[t][j]; }
        }
        buf = b1;
    }
    if (RAGK && tail > 0) {                                  // the partial stage sits in `buf`, visible since the last barrier
        const float *a = lds + buf * STAGE, *b = a + BM * BK;
        const int nct = (tail + 7) >> 3;
        for (int c = kg; c < nct; c += 2) {
            float ta[2][4], tb[2][4];
            rd(a, b, c, ta, tb);
            const int k0 = c * 8 + 4 * h;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) { if (k0 + j >= tail) { ta[t][j] = 0.f; tb[t][j] = 0.f; } }
            mm(ta, tb);
        }
    }
    // the two k-groups meet in LDS (64 KiB: 4 waves x 64 accumulator registers x 64 lanes), group 0 stores
    __syncthreads();
    if (kg == 1) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) lds[((w4 * 4 + a * 2 + b) * 16 + r) * 64 + lane] = acc[a][b][r];
    }
    __syncthreads();
    if (kg == 1) return;
    const float alpha = EPI ? p.alpha : 1.f, beta = EPI ? p.beta : 0.f;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int gn = n0 + wn * 64 + b * 32 + l31;
            const float bsv = (EPI && p.bias) ? p.bias[gn] : 0.f;
            float old[16];
            if (EPI && beta != 0.f) {                        // all 16 loads of the block in flight before the first store (a store ahead of a may-alias load serialises them)
#pragma unroll
                for (int r = 0; r < 16; r++) old[r] = p.O[(long)(m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * N + gn];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int gm = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float o = acc[a][b][r] + lds[((w4 * 4 + a * 2 + b) * 16 + r) * 64 + lane];
                if (EPI) { o *= alpha; if (beta != 0.f) o += old[r] * beta; o += bsv; }
                __builtin_nontemporal_store(o, &p.O[(long)gm * N + gn]);
            }
        }
}
template <bool AKC, bool BKC, bool EPI, bool RAGK = false, int BK_ = 64>
void launch_plain128_(const PlainP &q, hipStream_t s) {
    constexpr size_t lds_bytes = (size_t)2 * 256 * BK_ * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_plain128<AKC, BKC, EPI, RAGK, BK_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); attr_done = true; }
    T4K_LAUNCH((k_gemm_plain128<AKC, BKC, EPI, RAGK, BK_>), dim3((unsigned)((q.M / 128) * (q.N / 128))), dim3(512), lds_bytes, s, q);
}

// 256 x 256 tile, 16 waves (round 5): for outputs of at least one such tile per CU (4096^2 and up).  The 128 x 128 kernel pays its stage barrier once per 8 192 MFMA
// cycles of a SIMD (two waves); here sixteen waves - a 4 x 4 grid of 64 x 64 blocks, no k-groups, no meeting in LDS at the end - put 16 384 cycles of MFMAs behind every
// barrier on 32-deep stages of the same 64 KiB, and a stage moves half the bytes per flop (512 rows for 256 x 256 outputs).  126 registers: four waves per SIMD.
// Interior tiles, K % 32 == 0; layouts and epilogue as k_gemm_plain128.
template <bool AKC, bool BKC, bool EPI>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) k_gemm_plain256(PlainP p) {
    constexpr int BM = 256, BN = 256, BK = 32;
    constexpr int NC = BK / 8, CH = BK / 4, RPI = 64 / CH;
    constexpr int STAGE = (BM + BN) * BK;
    constexpr int NJ = (BM * BK / 256) / 16;               // 1-KiB DMA instructions per operand per wave per stage (2)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3, h = lane >> 5, l31 = lane & 31;
    const int M = p.M, N = p.N, K = p.K;
    const int tiles_m = M / BM, tiles_n = N / BN, T = tiles_m * tiles_n;
    int tm, tn;
    {
        int L;
        { const int b = blockIdx.x, q8 = T >> 3, r8 = T & 7, x = b & 7, i = b >> 3;
          L = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i; }           // XCD x owns a contiguous run of the tile order
        constexpr int GROUP_M = 4;
        const int per_group = GROUP_M * tiles_n;
        const int grp = L / per_group, first_m = grp * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        tm = first_m + (L % per_group) % gsz; tn = (L % per_group) / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN, nst = K / BK;
    unsigned voffA[NJ], voffB[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int i = w * NJ + j;
        if (AKC) { const int r = i * RPI + lane / CH, q = (lane % CH) ^ (r & (CH - 1)); voffA[j] = (unsigned)((m0 + r) * K + q * 4) * 4u; }
        else     {                                                                      voffA[j] = (unsigned)(i * M + m0 + lane * 4) * 4u; }     // one k row of 256 floats per instruction
        if (BKC) { const int r = i * RPI + lane / CH, q = (lane % CH) ^ (r & (CH - 1)); voffB[j] = (unsigned)((n0 + r) * K + q * 4) * 4u; }
        else     {                                                                      voffB[j] = (unsigned)(i * N + n0 + lane * 4) * 4u; }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        const float *ba = p.A + (AKC ? (long)kt * BK : (long)kt * BK * M), *bb = p.B + (BKC ? (long)kt * BK : (long)kt * BK * N);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * STAGE + (w * NJ + j) * 256) * 4));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffA[j]), "s"(ba), "s"(la) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voffB[j]), "s"(bb), "s"(la + BM * BK * 4) : "memory");
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    const int ra_ = wm * 64 + l31, rb_ = wn * 64 + l31;
    auto rd = [&](const float *a, const float *b, int ci, float (&av)[2][4], float (&bv)[2][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int ra = ra_ + t * 32, rb = rb_ + t * 32;
            if (AKC) { const v4f v = *reinterpret_cast<const v4f *>(a + ra * BK + (((ci * 2 + h) ^ (ra & (CH - 1))) << 2));
                       av[t][0] = v[0]; av[t][1] = v[1]; av[t][2] = v[2]; av[t][3] = v[3]; }
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) av[t][j] = a[(ci * 8 + 4 * h + j) * BM + ra];
            }
            if (BKC) { const v4f v = *reinterpret_cast<const v4f *>(b + rb * BK + (((ci * 2 + h) ^ (rb & (CH - 1))) << 2));
                       bv[t][0] = v[0]; bv[t][1] = v[1]; bv[t][2] = v[2]; bv[t][3] = v[3]; }
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) bv[t][j] = b[(ci * 8 + 4 * h + j) * BN + rb];
            }
        }
    };
    auto mm = [&](float (&av)[2][4], float (&bv)[2][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][j], bv[0][j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][j], bv[1][j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][j], bv[0][j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][j], bv[1][j], acc[1][1], 0, 0, 0);
        }
    };
    // four waves per SIMD: a wave reads a chunk's fragments and multiplies them - while it waits for LDS the other three keep the MFMA pipe busy, so no
    // second fragment set is carried (with it the kernel spills, and a compiler-counted scratch reload waits for vmcnt(0), i.e. for the DMA of the NEXT stage)
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (nst > 1) issue(1, 1);
    int buf = 0;
    for (int kt = 0; kt < nst; kt++) {
        const float *a = lds + buf * STAGE, *b = a + BM * BK;
#pragma unroll
        for (int ci = 0; ci < NC; ci++) {
            float ca[2][4], cb[2][4];
            rd(a, b, ci, ca, cb);
            mm(ca, cb);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nst) issue(kt + 2, buf);
        buf ^= 1;
    }
    const float alpha = EPI ? p.alpha : 1.f, beta = EPI ? p.beta : 0.f;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int gn = n0 + wn * 64 + b * 32 + l31;
            const float bsv = (EPI && p.bias) ? p.bias[gn] : 0.f;
            float old[16];
            if (EPI && beta != 0.f) {
#pragma unroll
                for (int r = 0; r < 16; r++) old[r] = p.O[(long)(m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * N + gn];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int gm = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float o = acc[a][b][r];
                if (EPI) { o *= alpha; if (beta != 0.f) o += old[r] * beta; o += bsv; }
                __builtin_nontemporal_store(o, &p.O[(long)gm * N + gn]);
            }
        }
}
template <bool AKC, bool BKC, bool EPI>
void launch_plain256_(const PlainP &q, hipStream_t s) {
    constexpr size_t lds_bytes = (size_t)2 * 512 * 32 * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_plain256<AKC, BKC, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); attr_done = true; }
    T4K_LAUNCH((k_gemm_plain256<AKC, BKC, EPI>), dim3((unsigned)((q.M / 256) * (q.N / 256))), dim3(1024), lds_bytes, s, q);
}
void launch_plain128(const GemmP &p, int tA, int tB, hipStream_t s) {
    PlainP q{ p.A, p.B, p.O, p.M, p.N, p.K, p.alpha, p.beta, p.bias, nullptr, nullptr };
    const bool epi = p.alpha != 1.0f || p.beta != 0.0f || p.bias, ragk = p.K % 64 != 0;
    {   // one 256 x 256 tile or more per CU: the 16-wave kernel (T4K_GEMM_PLAIN256: 0 off, 1 default, 2 any grid of whole 256-tiles)
        static const int p256 = T4K_LAB_ENV("T4K_GEMM_PLAIN256", 1);
        const long t256 = (long)(p.M / 256) * (p.N / 256), t128 = 4 * t256, cu = st().cu_count;
        // wave quantisation: 4352^2 is 289 tiles of 256^2 - two rounds, the second 13 % full - but 1156 tiles of 128^2 at 90 % fill; the 256-tile pipeline is worth ~4 % per FLOP
        auto fill_of = [](long tiles_, long cu_) { return (double)tiles_ / (double)(((tiles_ + cu_ - 1) / cu_) * cu_); };
        if (p256 && p.M % 256 == 0 && p.N % 256 == 0 && p.K % 32 == 0 && p.K >= 64 && (p256 >= 2 || (t256 >= cu && fill_of(t256, cu) * 1.04 > fill_of(t128, cu))) &&
            (long)p.M * p.K < (1L << 30) && (long)p.N * p.K < (1L << 30)) {
#define T4K_P256(A_, B_) do { if (epi) launch_plain256_<A_, B_, true>(q, s); else launch_plain256_<A_, B_, false>(q, s); } while (0)
            if (!tA && !tB) T4K_P256(true, false); else if (!tA) T4K_P256(true, true); else if (!tB) T4K_P256(false, false); else T4K_P256(false, true);
#undef T4K_P256
            return;
        }
    }
    // Two co-resident workgroups per CU on 32-deep stages once every CU gets at least two tiles (4096^2 x 1024: 261 -> 254 us, 8192 x 4096 x 512: 282 -> 264 us;
    // with one tile per CU the doubled barrier count loses: 2048^3 129.5 -> 135.4 us).  K in whole 32s is unragged for this form.  T4K_GEMM_PLAIN128_BK32 = 0 / 1 / 2 (always).
    static const int bk32 = T4K_LAB_ENV("T4K_GEMM_PLAIN128_BK32", 1);
    const long tiles = (long)(p.M / 128) * (p.N / 128);
    const bool two = bk32 && p.K % 32 == 0 && (bk32 >= 2 || tiles >= 2L * st().cu_count);
#define T4K_PL(A_, B_) do { if (two) { if (epi) launch_plain128_<A_, B_, true, false, 32>(q, s); else launch_plain128_<A_, B_, false, false, 32>(q, s); } \
                            else if (ragk) { if (epi) launch_plain128_<A_, B_, true, true>(q, s); else launch_plain128_<A_, B_, false, true>(q, s); } \
                            else if (epi) launch_plain128_<A_, B_, true>(q, s); else launch_plain128_<A_, B_, false>(q, s); } while (0)
    if (!tA && !tB) T4K_PL(true, false); else if (!tA) T4K_PL(true, true); else if (!tB) T4K_PL(false, false); else T4K_PL(false, true);
#undef T4K_PL
}

template <int BK, bool RAGK>
void launch_glds8_(const GemmP &p, dim3 grid, int tA, int tB, hipStream_t s) {
    constexpr size_t lds_bytes = (size_t)((BK >= 128) ? 2 : 3) * 128 * BK * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_glds8<BK, true, false, false, RAGK>),  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_glds8<BK, true, true, false, RAGK>),   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_glds8<BK, false, false, false, RAGK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_glds8<BK, false, true, false, RAGK>),  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_done = true;
    }
    if (!tA && !tB) T4K_LAUNCH((k_gemm_glds8<BK, true,  false, false, RAGK>), grid, dim3(512), lds_bytes, s, p);
    else if (!tA)   T4K_LAUNCH((k_gemm_glds8<BK, true,  true,  false, RAGK>), grid, dim3(512), lds_bytes, s, p);
    else if (!tB)   T4K_LAUNCH((k_gemm_glds8<BK, false, false, false, RAGK>), grid, dim3(512), lds_bytes, s, p);
    else            T4K_LAUNCH((k_gemm_glds8<BK, false, true,  false, RAGK>), grid, dim3(512), lds_bytes, s, p);
}
template <int BK>
void launch_glds8(const GemmP &p, dim3 grid, int tA, int tB, hipStream_t s) { launch_glds8_<BK, false>(p, grid, tA, tB, s); }
template <int BK>
void launch_glds8_ragk(const GemmP &p, dim3 grid, int tA, int tB, hipStream_t s) { launch_glds8_<BK, true>(p, grid, tA, tB, s); }   // K with a partial last stage

// fold split-K slabs in slice order, then the alpha/beta epilogue (reference t4math.cu:580)
// Optional activation epilogue (the layer that follows a linear layer: relu / tanh / ... / dropout): O keeps the linear
// output, ACT_O / ACT_F receive the activation output and derivative mask (k_activate nmath.cu:37-70); dropout draws its
// Philox slice here, exactly the values t4k_rand would have stored in the mask tensor.
// Riders of the fold launch: a second element-wise layer behind the first (the run `leakyrelu dropout` of the GAN nets), and a plain
// copy done by cp_blocks extra workgroups (the model's copy of the batch into its layer 0, forward.cu:39, rides with the first
// linear layer's fold instead of taking a launch of its own).
__global__ void __launch_bounds__(BLK) k_splitk_fold(const float *__restrict__ part, float *O, long mn, int nsplit,
                                                     float alpha, float beta, const float *__restrict__ bias, int N, ActEpi ep, FoldRider fr) {
    uint64_t base = 0, seed = 0;
    const ActEpi &ep2 = fr.ep2;
    const bool draw1 = ep.layer == T4K_L_DROPOUT, draw2 = ep2.layer == T4K_L_DROPOUT;      // at most one of the two (one dropout per run)
    const RngArg &rg = draw2 ? ep2.rng : ep.rng;
    if (draw1 || draw2) rng_begin(rg, base, seed);
    const int nfold = (int)gridDim.x - fr.cp_blocks;
    if ((int)blockIdx.x >= nfold) {
        const long t0 = (long)((int)blockIdx.x - nfold) * BLK + threadIdx.x, step = (long)fr.cp_blocks * BLK;
        if (fr.cp_vec) {
            const long n4 = fr.cp_n >> 2;
            for (long z = t0; z < n4; z += step) reinterpret_cast<float4 *>(fr.cp_dst)[z] = reinterpret_cast<const float4 *>(fr.cp_src)[z];
            for (long z = (n4 << 2) + t0; z < fr.cp_n; z += step) fr.cp_dst[z] = fr.cp_src[z];
        } else
            for (long z = t0; z < fr.cp_n; z += step) fr.cp_dst[z] = fr.cp_src[z];
    } else
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < mn; z += (long)nfold * BLK) {
        float s = 0.f;
#pragma unroll 4
        for (int k = 0; k < nsplit; k++) s += part[(long)k * mn + z];
        float o = s * alpha;
        if (beta != 0.f) o += O[z] * beta;
        if (bias) o += bias[z % N];
        O[z] = o;
        if (fr.mc.d1) { const float g1 = o * fr.mc.m1[z]; fr.mc.d1[z] = g1; if (fr.mc.d2) fr.mc.d2[z] = g1 * fr.mc.m2[z]; }
        if (ep.layer) {
            float a, f; act_rt(ep.layer, o, draw1 ? philox_u01_at(base, seed, z) : 0.f, ep.alpha, a, f); ep.F[z] = f; ep.A[z] = a;
            if (ep2.layer) { float a2, f2; act_rt(ep2.layer, a, draw2 ? philox_u01_at(base, seed, z) : 0.f, ep2.alpha, a2, f2); ep2.F[z] = f2; ep2.A[z] = a2; }
        }
    }
    if ((draw1 || draw2) && rg.state) rng_advance_last_block(rg.state, base, (uint64_t)((mn + 3) >> 2));
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

template <int BM, int BN, int BK, bool AKC, bool BKC, bool VEC, bool SKEW, bool FULL>
void launch_one(const GemmP &p, dim3 grid, hipStream_t s) {
    constexpr size_t lds_bytes = (size_t)2 * (BM + BN) * BK * sizeof(float);
    static bool attr_done = false;
    auto kern = k_gemm_mfma<BM, BN, BK, AKC, BKC, VEC, SKEW, FULL>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_done = true;
    }
    T4K_LAUNCH(kern, grid, dim3(256), lds_bytes, s, p);
}
template <int BM, int BN, int BK, bool VEC, bool SKEW, bool FULL>
void launch_variant(const GemmP &p, dim3 grid, int tA, int tB, hipStream_t s) {
    if (!tA && !tB) launch_one<BM, BN, BK, true,  false, VEC, SKEW, FULL>(p, grid, s);
    else if (!tA)   launch_one<BM, BN, BK, true,  true,  VEC, SKEW, FULL>(p, grid, s);
    else if (!tB)   launch_one<BM, BN, BK, false, false, VEC, SKEW, FULL>(p, grid, s);
    else            launch_one<BM, BN, BK, false, true,  VEC, SKEW, FULL>(p, grid, s);
}

struct ColSum { const float *X; float *out; int rows, E; bool done; };
bool plain_any_big() { static const int v = T4K_LAB_ENV("T4K_GEMM_PLAIN_ANY", 2); return v >= 2; }   // 0 off, 1 one-tile-per-CU shapes only, 2 (default) large ones too
bool big_dma() { static const int v = T4K_LAB_ENV("T4K_GEMM_BIG_DMA", 1); return v != 0 && gates_ok(); }
bool capturing(hipStream_t hs) { hipStreamCaptureStatus st_ = hipStreamCaptureStatusNone; return hipStreamIsCapturing(hs, &st_) == hipSuccess && st_ != hipStreamCaptureStatusNone; }   // a replayed graph would repeat the epoch argument
bool dual_on() { static const int v = T4K_LAB_ENV("T4K_GEMM_DUAL", 1); return v != 0 && gates_ok(); }
// dW += dY^T X (+ dB += column sums of dY) and dX = dY W of one linear layer in a single launch (k_gemm_dual); false when the
// shapes belong to the other kernels (deep K -> split-K, large -> 128x128 tiles).  Interior-tile shapes take it too since round 2
// (T4K_GEMM_DUAL_FULL=0: the LDS-DMA kernels, 5 launches with their folds and the column sum: GAN round 0.356 instead of 0.318 ms of GPU time)
bool linear_bwd_dual(const float *X, const float *W, const float *DY, float *DX, float *DW, float *DB, int N, int E0, int E1, hipStream_t hs,
                     const MaskChain *mcp = nullptr) {
    State &g = st();
    int *gate = gate_for(hs, 1);                             // nullptr: unknown stream, the two GEMMs go out as separate launches
    if (!dual_on() || !g.d_sync || !gate || (E0 & 3) || (E1 & 3) || !aligned16(DY) || !aligned16(X) || !aligned16(W) || N < 1) return false;
    const int cu = g.cu_count;
    auto tiles = [](int m, int n) { return (long)((m + 63) / 64) * ((n + 63) / 64); };
    auto big = [&](int m, int n) { return (long)((m + 127) / 128) * ((n + 127) / 128) >= (long)cu * 3 / 4; };
    auto splits = [&](int m, int n, int k) { return tiles(m, n) * 2 <= cu && k >= 256; };
    auto full = [](int m, int n, int k) { return m % 64 == 0 && n % 64 == 0 && k % 64 == 0; };
    static const int deepk = T4K_LAB_ENV("T4K_GEMM_DUAL_MAXK", 1024);
    // deep-K shapes would go split-K + fold (2 launches per GEMM); up to K = 1024 the single dual launch, unsplit, is faster (GAN round 0.318 ms of GPU time; with T4K_GEMM_DUAL_MAXK=256: 0.341)
    const bool sp = splits(E0, E1, N) || splits(N, E1, E0);
    if (big(E0, E1) || big(N, E1) || (sp && (N > deepk || E0 > deepk))) return false;
    static const int dfull = T4K_LAB_ENV("T4K_GEMM_DUAL_FULL", 1);
    if (!dfull && (full(E0, E1, N) || full(N, E1, E0))) return false;
    const long t1 = tiles(E0, E1), t2 = tiles(N, E1), riders = (E0 + 63) / 64;
    if (t1 + riders + t2 > cu || N > 4096) return false;
    {   // small layers: 32x32 tiles, K split over the waves of a workgroup, operand blocks through LDS-DMA (k_gemm_dual_l32)
        static const int s32 = T4K_LAB_ENV("T4K_GEMM_DUAL32", 1);
        auto t32 = [](int m, int n) { return (long)((m + 31) / 32) * ((n + 31) / 32); };
        const long a1 = t32(E0, E1), a2 = t32(N, E1), ar = (E0 + 63) / 64;
        // the gated (in-place dX) launch may hold more workgroups than fit the chip at 1-2 per CU (64*W threads, W*16 KiB of LDS): progress then rests on the
        // dispatcher handing out workgroups in id order - the dW blocks [0, a1 + ar) never wait, the dX writers behind them wait only for those - so the
        // id -> tile map stays the identity; a device that dispatches out of order ends in the bounded-spin error
        if (s32 && g.d_zero && a1 + ar + a2 <= 4L * cu - 32 && a1 <= 512 && N <= 1024 && E0 <= 1024 && !capturing(hs)) {      // a wave walks at most four 32-deep blocks; deeper K stays with the staged kernels
            GemmP q1, q2;
            auto fill32 = [&](GemmP &p, const float *A, const float *B, float *O, int M, int Nn, int K, float beta) {
                p.A = A; p.B = B; p.bias = nullptr; p.O = O; p.part = nullptr; p.M = M; p.N = Nn; p.K = K; p.C = 1;
                p.tiles_m = (M + 31) / 32; p.tiles_n = (Nn + 31) / 32; p.kchunk = K; p.nsplit = 1;
                p.alpha = 1.0f; p.beta = beta; p.pair = 0; p.sync = g.d_sync; p.cs_X = nullptr; p.cs_out = nullptr; p.cs_rows = 0; p.cs_E = 0; p.xmap = 0; p.Z = g.d_zero;
            };
            fill32(q1, DY, X, DW, E0, E1, N, 1.0f);
            q1.cs_X = DY; q1.cs_out = DB; q1.cs_rows = N; q1.cs_E = E0;
            fill32(q2, DY, W, DX, N, E1, E0, 0.0f);
            const bool alias32 = (const float *)DX == X || (const float *)DX == DY || (const float *)DX == W;
            const MaskChain mc32 = mcp ? *mcp : MaskChain{nullptr, nullptr, nullptr, nullptr};
            // arrival slots: ints [512, 1024) of the stream's gate block; the epoch is this stream's launch count (never 0, slots cleared when it wraps)
            unsigned *slots = reinterpret_cast<unsigned *>(gate_for(hs, 0)) + 512;
            const unsigned epoch = next_slot_epoch(hs, slots);     // the lane's ONE counter, shared with k_head_bwd_l32 (t4k_common.h)
            const dim3 gd((unsigned)(a1 + ar + a2));
#define T4K_DL32(R_, W_) do { static bool attr_done = false; \
                if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_dual_l32<R_, W_>), hipFuncAttributeMaxDynamicSharedMemorySize, W_ * 16384); attr_done = true; } \
                T4K_LAUNCH((k_gemm_dual_l32<R_, W_>), gd, dim3(64 * W_), (size_t)W_ * 16384, hs, q1, q2, (int)(a1 + ar), (int)a1, (int)a2, alias32 ? slots : nullptr, epoch, mc32); } while (0)
            if (N > 512 || E0 > 512) T4K_DL32(true, 8);            // deep K: 8 k-groups (one arrival slot per workgroup)
            else if (N > 256 || E0 > 256) T4K_DL32(true, 4); else T4K_DL32(false, 4);
#undef T4K_DL32
            return true;
        }
    }
    GemmP p1, p2;
    auto fill = [&](GemmP &p, const float *A, const float *B, float *O, int M, int Nn, int K, float beta) {
        p.A = A; p.B = B; p.bias = nullptr; p.O = O; p.part = nullptr; p.M = M; p.N = Nn; p.K = K; p.C = 1;
        p.tiles_m = (M + 63) / 64; p.tiles_n = (Nn + 63) / 64; p.kchunk = ((K + 63) / 64) * 64; p.nsplit = 1;
        p.alpha = 1.0f; p.beta = beta; p.pair = 0; p.sync = g.d_sync; p.cs_X = nullptr; p.cs_out = nullptr; p.cs_rows = 0; p.cs_E = 0; p.xmap = 0; p.Z = nullptr;
    };
    fill(p1, DY, X, DW, E0, E1, N, 1.0f);                    // A = dY^T ([K][M]), B = X ([K][N])
    p1.cs_X = DY; p1.cs_out = DB; p1.cs_rows = N; p1.cs_E = E0;
    fill(p2, DY, W, DX, N, E1, E0, 0.0f);                    // A = dY ([M][K]), B = W ([K][N])
    constexpr size_t lds_bytes = (size_t)2 * (64 + 64) * 64 * sizeof(float);
    static const int dfk = T4K_LAB_ENV("T4K_GEMM_DUAL_FULLK", 1);
    const bool f1 = dfk && N % 64 == 0 && E0 >= 4 && E1 >= 4, f2 = dfk && E0 % 64 == 0 && E1 >= 4;
    const dim3 grid((unsigned)(t1 + riders + t2));
    const bool alias = (const float *)DX == X || (const float *)DX == DY || (const float *)DX == W;   // only an in-place dX needs the arrival gate
    const MaskChain mc = mcp ? *mcp : MaskChain{nullptr, nullptr, nullptr, nullptr};
#define T4K_DUAL(F1_, F2_) do { auto kern = k_gemm_dual<false, false, true, false, F1_, F2_>; static bool attr_done = false; \
        if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); attr_done = true; } \
        T4K_LAUNCH(kern, grid, dim3(256), lds_bytes, hs, p1, p2, (int)(t1 + riders), (int)t1, (int)t2, alias ? gate : nullptr, mc); } while (0)
    if (f1 && f2) T4K_DUAL(true, true); else if (f1) T4K_DUAL(true, false); else if (f2) T4K_DUAL(false, true); else T4K_DUAL(false, false);
#undef T4K_DUAL
    return true;
}
int gemm_launch(const float *A, const float *B, float *O, const float *bias, float alpha, float beta,
                int tA, int tB, int M, int N, int K, int C, t4k_stream_t s, const ActEpi *epi = nullptr, bool *epi_done = nullptr,
                ColSum *cs = nullptr, XFold *defer = nullptr, FoldRider *rider = nullptr) {   // rider: in ep2 / cp_*, out cp_blocks > 0 when the copy went with the fold
    if (epi_done) *epi_done = false;
    if (defer) defer->part = nullptr;
    if (!A || !B || !O || M < 0 || N < 0 || K < 0 || C < 1) return fail(T4K_ERR_ARG, "t4k_gemm: bad argument");
    if (M == 0 || N == 0) return T4K_OK;
    GemmP p;
    p.A = A; p.B = B; p.O = O; p.bias = bias; p.part = ws_for(s);
    p.M = M; p.N = N; p.K = K; p.C = C; p.alpha = alpha; p.beta = beta; p.xmap = 0; p.Z = nullptr;

    // 16-byte loads need: C == 1, aligned bases, contiguous extents divisible by 4
    const int a_contig = tA ? M : K, b_contig = tB ? K : N;
    const bool vec = (C == 1) && aligned16(A) && aligned16(B) && (a_contig % 4 == 0) && (b_contig % 4 == 0);

    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const bool big = t128 >= (long)st().cu_count * 3 / 4;
    const int BMv = big ? 128 : 64;
    p.tiles_m = (M + BMv - 1) / BMv; p.tiles_n = (N + BMv - 1) / BMv;
    const long tiles = (long)p.tiles_m * p.tiles_n;

    // Slivers (the output gives at most half the CUs a 64x64 tile, K moderate): 32x32 tiles, K split over the waves, operand blocks through
    // LDS-DMA, the whole epilogue (bias, activation riders, mask chain, layer-0 copy, column sums) in the same launch - no fold
    {
        static const int s32 = T4K_LAB_ENV("T4K_GEMM_S32", 1);
        static const int maxk = T4K_LAB_ENV("T4K_GEMM_S32_MAXK", 832);   // measured: 256 x 512 x K wins up to K = 784 (7.6 vs 10.6 us at 512), loses at 1024 (12.8 vs 10.7 us split-K + fold)
        const bool akc = !tA, bkc = tB != 0;
        const bool al = (!akc || (K % 4 == 0 && aligned16(A))) && (!bkc || (K % 4 == 0 && aligned16(B)));
        const long t32 = (long)((M + 31) / 32) * ((N + 31) / 32);
        int ns = 1, kc = ((K + 7) / 8) * 8;
        if (defer && alpha == 1.0f && beta == 0.0f && K >= 256) {               // a consumer folds the slabs anyway: spread K over idle CUs
            int want = (int)((st().cu_count + t32 - 1) / t32); if (want > K / 128) want = K / 128; if (want > 16) want = 16;
            if (want > 1) { kc = (((K + want - 1) / want) + 7) / 8 * 8; ns = (K + kc - 1) / kc; }
            if ((size_t)ns * M * N * sizeof(float) > st().ws_bytes / 2) { ns = 1; kc = ((K + 7) / 8) * 8; }
        }
        // operands through LDS-DMA blocks: every 16-byte DMA lane aligned and whole (other shapes: the 64x64 kernels below)
        const bool dma_ok = st().d_zero && aligned16(A) && aligned16(B) && (akc ? K % 4 == 0 : M % 4 == 0) && (bkc ? K % 4 == 0 : N % 4 == 0);
        if (s32 && C == 1 && !big && tiles * 2 <= st().cu_count && K >= 1 && kc <= maxk && al && dma_ok && !capturing(S(s))) {   // kc: depth one workgroup walks
            hipStream_t hs2 = S(s);
            p.tiles_m = (M + 31) / 32; p.tiles_n = (N + 31) / 32;
            p.kchunk = kc; p.nsplit = ns; p.pair = 0; p.sync = st().d_sync;
            p.cs_X = nullptr; p.cs_out = nullptr; p.cs_rows = 0; p.cs_E = 0;
            unsigned gx = (unsigned)t32;
            if (cs && ns == 1 && cs->rows > 0 && cs->rows <= 4096 && cs->E > 0) {
                p.cs_X = cs->X; p.cs_out = cs->out; p.cs_rows = cs->rows; p.cs_E = cs->E; cs->done = true; gx += (unsigned)((cs->E + 63) / 64);
            }
            const long mn = (long)M * N;
            ActEpi ep = {0, 0.f, nullptr, nullptr, RngArg{0, 0, nullptr}};
            FoldRider fr = {ep, nullptr, nullptr, 0, 0, 0, MaskChain{nullptr, nullptr, nullptr, nullptr}, 0};
            if (ns == 1) {
                if (epi && epi->layer) {
                    ep = *epi; if (epi_done) *epi_done = true;
                    if (ep.layer == T4K_L_DROPOUT) ep.rng = rng_draw(hs2, (uint64_t)((mn + 3) >> 2), true);
                    if (rider && rider->ep2.layer) {
                        fr.ep2 = rider->ep2;
                        if (fr.ep2.layer == T4K_L_DROPOUT) fr.ep2.rng = rng_draw(hs2, (uint64_t)((mn + 3) >> 2), true);
                    }
                }
                if (rider && rider->mc.d1) { fr.mc = rider->mc; rider->mc_done = 1; }
                if (rider && rider->cp_src && rider->cp_dst && rider->cp_n > 0) {
                    fr.cp_src = rider->cp_src; fr.cp_dst = rider->cp_dst; fr.cp_n = rider->cp_n; fr.cp_vec = aligned16(fr.cp_src) && aligned16(fr.cp_dst);
                    fr.cp_blocks = grid_for(rider->cp_n, 16); if (fr.cp_blocks > 128) fr.cp_blocks = 128;
                    rider->cp_blocks = fr.cp_blocks; gx += (unsigned)fr.cp_blocks;
                }
            } else { defer->part = p.part; defer->nsplit = ns; defer->mn = mn; }
            const dim3 g32(gx, (unsigned)ns);
            {
                p.Z = st().d_zero;
                static const int xm = T4K_LAB_ENV("T4K_GEMM_XMAP", 0);   // measured: no effect on the GAN layers (the Infinity Cache serves all eight L2s), off
                p.xmap = (xm && t32 >= 16) ? (N >= M ? 2 : 1) : 0;       // XCD-aware tile order: an XCD's L2 pulls its own slice of the wider operand only
                const int nblk = (kc + 31) / 32;
                // waves per workgroup (= k-groups) and whether a wave walks more than two blocks (RST: blocks 2, 3 wait in registers)
                const bool w8 = nblk > 16 || (nblk >= 6 && t32 * ns <= (long)st().cu_count);
                const bool rst = nblk > (w8 ? 16 : 8);
#define T4K_L32(A_, B_) do { if (w8) { if (rst) launch_l32<A_, B_, 8, true>(p, ep, fr, g32, hs2); else launch_l32<A_, B_, 8, false>(p, ep, fr, g32, hs2); } \
                             else    { if (rst) launch_l32<A_, B_, 4, true>(p, ep, fr, g32, hs2); else launch_l32<A_, B_, 4, false>(p, ep, fr, g32, hs2); } } while (0)
                if (akc && !bkc) T4K_L32(true, false); else if (akc) T4K_L32(true, true); else if (!bkc) T4K_L32(false, false); else T4K_L32(false, true);
#undef T4K_L32
                T4K_LAUNCH_CHECK();
                return T4K_OK;
            }
        }
    }
    {   // 65..128 interior tiles, K in whole 256s: two workgroups per tile on the lean kernel, combined in its epilogue (k_gemm_nn_plain<.., PAIR>) -
        // 512 x 1024 x 1024: one launch instead of split-K slabs + a fold launch.  Tickets are per tile: default stream only.
        static const int ppair = T4K_LAB_ENV("T4K_GEMM_PLAIN_PAIR", 1);
        if (ppair && gates_ok() && !big && vec && C == 1 && M % 64 == 0 && N % 64 == 0 && K % 256 == 0 && K >= 512 &&
            tiles * 2 <= st().cu_count && tiles * 3 > st().cu_count && tiles <= 2048 && !defer && !(epi && epi->layer) && !rider && !cs &&
            st().d_sync && lane_of(S(s)) == 0 && (size_t)tiles * 4096 * sizeof(float) <= st().ws_bytes / 2 &&
            (size_t)M * K * sizeof(float) < ((size_t)1 << 32) && (size_t)K * N * sizeof(float) < ((size_t)1 << 32)) {
            p.sync = st().d_sync; p.pair = 0; p.nsplit = 1; p.kchunk = K;
            launch_plain_any(p, dim3((unsigned)tiles), tA, tB, S(s), true);
            T4K_LAUNCH_CHECK();
            return T4K_OK;
        }
    }
    // split K when the output alone cannot fill the chip (granularity = the deepest stage, 64)
    constexpr int KG = 64;
    int nsplit = 1, kchunk = ((K + KG - 1) / KG) * KG; if (kchunk == 0) kchunk = KG;
    if (!big && C == 1 && tiles * 2 <= st().cu_count && K >= 4 * KG) {
        int want = (int)((st().cu_count + tiles - 1) / tiles);
        static const int sdiv = std::max(1, T4K_LAB_ENV("T4K_GEMM_SPLIT_DIV", 1));
        int maxs = K / (sdiv * KG); if (want > maxs) want = maxs; if (want > 64) want = 64;
        if (want > 1) {
            kchunk = (((K + want - 1) / want) + KG - 1) / KG * KG;
            nsplit = (K + kchunk - 1) / kchunk;
            if ((size_t)nsplit * M * N * sizeof(float) > st().ws_bytes / 2) { nsplit = 1; kchunk = ((K + KG - 1) / KG) * KG; }
        }
    }
    p.pair = 0; p.sync = st().d_sync;
    p.kchunk = kchunk; p.nsplit = nsplit;

    dim3 grid((unsigned)tiles, (unsigned)nsplit, (unsigned)C);
    hipStream_t hs = S(s);
    // ragged M / N with whole K stages on the 8-wave LDS-DMA kernel (clamped source rows, predicated stores) when the K loop is long enough
    // to matter: unsplit products (N = 1000 classes: 1024 x 1000 x 4096 122 -> measured below) - split slivers stay with the skewed kernel
    static const int rag = T4K_LAB_ENV("T4K_GEMM_RAGGED_DMA", 1);
    const bool ragged8 = rag && !big && vec && C == 1 && nsplit == 1 && (M % 64 != 0 || N % 64 != 0) &&
                         kchunk % 64 == 0 && K % kchunk == 0 && M >= 4 && N >= 4 &&
                         (size_t)M * K * sizeof(float) < ((size_t)1 << 32) && (size_t)K * N * sizeof(float) < ((size_t)1 << 32);
    // ragged K (784 = 28 x 28, the GAN layer width; any K when no operand is K-contiguous): the same kernel with a partial last stage
    // (k_gemm_glds8<.., RAGK>) instead of the predicated register-staged one (1024 x 1024 x 784: 28.5 us there).  T4K_GEMM_RAGGED_K: 0 off,
    // 1 unsplit products (default), 2 split-K slabs too
    static const int ragk_on = T4K_LAB_ENV("T4K_GEMM_RAGGED_K", 1);
    const bool ragk = ragk_on && !big && vec && C == 1 && !(kchunk % 64 == 0 && K % kchunk == 0) &&
                      (nsplit == 1 || ragk_on >= 2) && M >= 4 && N >= 4 && K >= 8 &&
                      (size_t)M * K * sizeof(float) < ((size_t)1 << 32) && (size_t)K * N * sizeof(float) < ((size_t)1 << 32);
    p.cs_X = nullptr; p.cs_out = nullptr; p.cs_rows = 0; p.cs_E = 0;
    {   // column-sum rider: only the generic kernel carries it, and only when one workgroup per tile writes the output
        const bool full64 = !big && vec && M % 64 == 0 && N % 64 == 0 && kchunk % 64 == 0 && K % kchunk == 0;
        const bool generic = big || !vec || !(full64 || ragged8 || ragk);
        if (cs && generic && nsplit == 1 && C == 1 && cs->rows > 0 && cs->rows <= 4096 && cs->E > 0) {
            p.cs_X = cs->X; p.cs_out = cs->out; p.cs_rows = cs->rows; p.cs_E = cs->E; cs->done = true;
            grid.x += (unsigned)((cs->E + 63) / 64);
        }
    }
    static const int plain_big = T4K_LAB_ENV("T4K_GEMM_PLAIN_BIG", 1);
    // interior tiles, unsplit, K >= 128 with a partial last stage: the lean kernel with a tail (any size of output)
    static const int pragk = T4K_LAB_ENV("T4K_GEMM_PLAIN_RAGK", 2);   // 0 off (the general kernel's tail), 1 only K % 64 != 0, 2 (default) every K % 128 != 0 (K = 960: 18.9 vs 19.6 us on the 64-deep general kernel)
    // T4K_GEMM_PLAIN128: 0 off, 1 (default) where it wins, 2 every eligible shape (tests).  A workgroup per CU at a time either way, so the
    // choice is wave quantisation: tiles / (rounds x CUs) of each tiling, the 128x128 pipeline being ~3.5 % faster per FLOP (2048^3: 136 -> 131 us)
    static const int p128 = T4K_LAB_ENV("T4K_GEMM_PLAIN128", 1);
    static const int p128rag = T4K_LAB_ENV("T4K_GEMM_PLAIN128_RAGK", 1);   // 0: a partial last K stage keeps the product on 64x64 tiles
    auto fill_of = [](long tiles_, long cu_) { return (double)tiles_ / (double)(((tiles_ + cu_ - 1) / cu_) * cu_); };
    const long t128i = (long)(M / 128) * (N / 128), t64i = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (p128 > 0 && vec && C == 1 && nsplit == 1 && !p.cs_X && M % 128 == 0 && N % 128 == 0 && (K % 64 == 0 || (p128rag && ragk_on && K % 4 == 0)) && K >= 256 &&
        (p128 >= 2 || (t128i >= st().cu_count && fill_of(t128i, st().cu_count) * 1.035 > fill_of(t64i, st().cu_count))) &&
        (size_t)M * K * sizeof(float) < ((size_t)1 << 32) && (size_t)K * N * sizeof(float) < ((size_t)1 << 32)) {
        p.kchunk = K;
        launch_plain128(p, tA, tB, hs);
    } else
    if (pragk && ragk_on && vec && C == 1 && nsplit == 1 && !p.cs_X && M % 64 == 0 && N % 64 == 0 &&
        K > 128 && K % 128 != 0 && (pragk >= 2 || K % 64 != 0) &&
        (size_t)M * K * sizeof(float) < ((size_t)1 << 32) && (size_t)K * N * sizeof(float) < ((size_t)1 << 32)) {
        p.kchunk = K;
        launch_plain_any(p, dim3((unsigned)((M / 64) * (N / 64))), tA, tB, hs, false, true);
    } else
    if (big && plain_big && vec && C == 1 && !tA && !tB && alpha == 1.0f && beta == 0.0f && !bias && !p.cs_X && M % 64 == 0 && N % 64 == 0 && K % 128 == 0 &&
        (size_t)M * K * sizeof(float) < ((size_t)1 << 32) && (size_t)K * N * sizeof(float) < ((size_t)1 << 32)) {
        launch_nn_plain(p, dim3((unsigned)((M / 64) * (N / 64))), hs);      // large plain products on the 64x64 LDS-DMA kernel, several tiles per CU
    } else if (big && plain_any_big() && vec && C == 1 && !p.cs_X && M % 64 == 0 && N % 64 == 0 && K % 128 == 0 && nsplit == 1 &&
               (size_t)M * K * sizeof(float) < ((size_t)1 << 32) && (size_t)K * N * sizeof(float) < ((size_t)1 << 32)) {
        launch_plain_any(p, dim3((unsigned)((M / 64) * (N / 64))), tA, tB, hs);      // large interior products of every layout on the lean kernel (2048^2 x 1024 linear + bias: 86 -> 72 us vs the 128x128 register-staged kernel)
    } else if (big && big_dma() && vec && C == 1 && !p.cs_X && K % 64 == 0 && K >= 2048 && M >= 4 && N >= 4 && nsplit == 1 &&   // deep K only: at K = 1024 the 128x128 kernel's fewer, fatter tiles win (292 vs 318 us at 4096 x 4096 x 1024)
               (size_t)M * K * sizeof(float) < ((size_t)1 << 32) && (size_t)K * N * sizeof(float) < ((size_t)1 << 32)) {
        // every other large product (transposed operands, alpha / beta / bias: the linear layers of an MLP) on the 8-wave LDS-DMA kernel too,
        // 64x64 tiles, several per CU, ragged edges clamped (the 128x128 register-staged kernel: 69-78 % of peak)
        p.tiles_m = (M + 63) / 64; p.tiles_n = (N + 63) / 64; p.kchunk = K; p.nsplit = 1;
        const dim3 g64((unsigned)(p.tiles_m * p.tiles_n), 1, 1);
        if (K % 128 == 0) launch_glds8<128>(p, g64, tA, tB, hs); else launch_glds8<64>(p, g64, tA, tB, hs);
    } else if (big) {
        static const int bigfk = T4K_LAB_ENV("T4K_GEMM_BIG_FULLK", 1);
        // whole K stages are enough for the predicate-free pipeline: ragged M / N edges are clamped source rows + predicated stores
        const bool full = vec && kchunk % 32 == 0 && K % kchunk == 0 && M >= 4 && N >= 4 && (bigfk || (M % 128 == 0 && N % 128 == 0));
        if (full)     launch_variant<128, 128, 32, true, true, true>(p, grid, tA, tB, hs);
        else if (vec) launch_variant<128, 128, 32, true, true, false>(p, grid, tA, tB, hs);
        else          launch_variant<128, 128, 32, false, false, false>(p, grid, tA, tB, hs);
    } else if (!vec) {
        launch_variant<64, 64, 32, false, false, false>(p, grid, tA, tB, hs);
    } else {
        const bool full = M % 64 == 0 && N % 64 == 0 && kchunk % 64 == 0 && K % kchunk == 0;
        if (ragk) {
            if (K >= 128 && (nsplit == 1 || kchunk % 128 == 0)) launch_glds8_ragk<128>(p, grid, tA, tB, hs); else launch_glds8_ragk<64>(p, grid, tA, tB, hs);
        } else if (ragged8) {
            if (kchunk % 128 == 0) launch_glds8<128>(p, grid, tA, tB, hs); else launch_glds8<64>(p, grid, tA, tB, hs);
        } else if (full) {
            const bool span32 = (size_t)M * K * sizeof(float) < ((size_t)1 << 32) && (size_t)K * N * sizeof(float) < ((size_t)1 << 32);   // 32-bit DMA lane offsets
            static const int plany = T4K_LAB_ENV("T4K_GEMM_PLAIN_ANY", 1);
            if (!span32) launch_variant<64, 64, 64, true, false, true>(p, grid, tA, tB, hs);          // operands of 4 GiB and more: the register-staged kernel (64-bit addresses)
            else if (kchunk % 128 == 0 && !tA && !tB && nsplit == 1 && alpha == 1.0f && beta == 0.0f && !bias) launch_nn_plain(p, grid, hs);   // `matmul`
            else if (plany && kchunk % 128 == 0 && nsplit == 1) launch_plain_any(p, grid, tA, tB, hs);   // the other layouts, alpha / beta / bias: the same lean kernel
            else if (kchunk % 128 == 0) launch_glds8<128>(p, grid, tA, tB, hs);   // split-K slabs of whole 128-deep stages
            else launch_glds8<64>(p, grid, tA, tB, hs);
        } else {
            static const int fk = T4K_LAB_ENV("T4K_GEMM_FULLK", 0);   // measured on the GAN nets: the skewed kernel is 1 % faster for ragged split-K shapes, off
            const int am = tA ? M : K, bn = tB ? K : N;     // the contiguous extents hold at least one 16-byte group (vec) - the clamp needs M, N >= 4 on the non-K-contiguous side
            if (fk && kchunk % 64 == 0 && K % kchunk == 0 && M >= 4 && N >= 4 && am >= 4 && bn >= 4) launch_variant<64, 64, 64, true, false, true>(p, grid, tA, tB, hs);   // ragged M / N, whole K stages
            else launch_variant<64, 64, 64, true, true, false>(p, grid, tA, tB, hs);
        }
    }
    if (nsplit > 1 && defer && alpha == 1.0f && beta == 0.0f) {      // the consumer folds the slabs (fused head)
        defer->part = p.part; defer->nsplit = nsplit; defer->mn = (long)M * N;
    } else if (nsplit > 1) {
        const long mn = (long)M * N;
        ActEpi ep = {0, 0.f, nullptr, nullptr, RngArg{0, 0, nullptr}};
        FoldRider fr = {ep, nullptr, nullptr, 0, 0, 0, MaskChain{nullptr, nullptr, nullptr, nullptr}, 0};
        if (rider && rider->mc.d1) { fr.mc = rider->mc; rider->mc_done = 1; }
        if (epi && epi->layer) {
            ep = *epi; if (epi_done) *epi_done = true;
            if (ep.layer == T4K_L_DROPOUT) ep.rng = rng_draw(hs, (uint64_t)((mn + 3) >> 2), true);
            if (rider && rider->ep2.layer) {
                fr.ep2 = rider->ep2;
                if (fr.ep2.layer == T4K_L_DROPOUT) fr.ep2.rng = rng_draw(hs, (uint64_t)((mn + 3) >> 2), true);
            }
        }
        int gfold = grid_for(mn);
        if (rider && rider->cp_src && rider->cp_dst && rider->cp_n > 0) {
            fr.cp_src = rider->cp_src; fr.cp_dst = rider->cp_dst; fr.cp_n = rider->cp_n; fr.cp_vec = aligned16(fr.cp_src) && aligned16(fr.cp_dst);
            fr.cp_blocks = grid_for(rider->cp_n, 4); if (fr.cp_blocks > 1024) fr.cp_blocks = 1024;
            rider->cp_blocks = fr.cp_blocks;
        }
        T4K_LAUNCH(k_splitk_fold, dim3(gfold + fr.cp_blocks), dim3(BLK), 0, hs, p.part, O, mn, nsplit, alpha, beta, bias, N, ep, fr);
    }
    T4K_LAUNCH_CHECK();
    return T4K_OK;
}

} // namespace

namespace t4k {
void gemm_set_spin_err(int *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_spin_err_dev), &p, sizeof(p)); }

int colsum_add(const float *X, float *OUT, long rows, int E, hipStream_t hs);
// linear_small.hip: classifier-head sized layers on the vector ALUs, one launch each way
bool linear_small_ok(int E0, int E1);
int  linear_small_fwd(const float *X, const float *W, const float *B, float *Y, float *P, int N, int E0, int E1, hipStream_t hs, const XFold *xf = nullptr,
                      const ActEpi *oep = nullptr);   // oep: element-wise layer behind the linear layer, applied in the same launch
bool linear_small_bwd(const float *X, const float *W, const float *DY, float *DX, float *DW, float *DB, int N, int E0, int E1, bool train, hipStream_t hs,
                      const float *MASK = nullptr, float *DXM = nullptr, const float *TGT = nullptr, float *DY2 = nullptr,
                      const float *MASKB = nullptr, float *DXMB = nullptr);
}

extern "C" {

int t4k_gemm(const float *A, const float *B, float *O, float alpha, float beta,
             int tA, int tB, int M, int N, int K, int C, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    return gemm_launch(A, B, O, nullptr, alpha, beta, tA, tB, M, N, K, C, s);
}

// Model::_flinear src/nn/forward.cu:157-198: Y[N,E0] = X[N,E1] @ W[E0,E1]^T + B[E0], bias fused
int t4k_linear_fwd(const float *X, const float *W, const float *B, float *Y, int N, int E0, int E1, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!X || !W || !Y || N < 0) return fail(T4K_ERR_ARG, "t4k_linear_fwd: bad argument");
    if (N > 0 && linear_small_ok(E0, E1)) { linear_small_fwd(X, W, B, Y, nullptr, N, E0, E1, S(s)); T4K_LAUNCH_CHECK(); return T4K_OK; }
    return gemm_launch(X, W, Y, B, 1.0f, 0.0f, 0, 1, N, E0, E1, 1, s);
}
// linear followed by an element-wise layer (relu, tanh, ..., dropout): Y = X W^T + b; ACT_O, ACT_F = activation(Y).
// When the GEMM is split-K the activation rides in the fold launch; otherwise it is a second launch.
int t4k_linear_act_fwd(const float *X, const float *W, const float *B, float *Y, int layer, float alpha, float *ACT_F, float *ACT_O,
                       int N, int E0, int E1, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!X || !W || !Y || !ACT_F || !ACT_O || N < 0) return fail(T4K_ERR_ARG, "t4k_linear_act_fwd: bad argument");
    if (N == 0) return T4K_OK;
    bool done = false;
    if (linear_small_ok(E0, E1)) {
        ActEpi ep = { layer, alpha, ACT_F, ACT_O, RngArg{0, 0, nullptr} };
        const bool fuse = !(layer == T4K_L_DROPOUT && st().capturing);   // a captured mask draw advances a device-resident counter: the separate launches below do that
        if (fuse && layer == T4K_L_DROPOUT) ep.rng = rng_draw(S(s), (uint64_t)(((long)N * E0 + 3) >> 2), true);
        linear_small_fwd(X, W, B, Y, nullptr, N, E0, E1, S(s), nullptr, fuse ? &ep : nullptr);
        done = fuse;
    } else {
        // the mask's Philox slice is reserved only if the fold launch will really apply the epilogue (decided inside gemm_launch)
        ActEpi ep = { layer, alpha, ACT_F, ACT_O, RngArg{0, 0, nullptr} };
        int rc = gemm_launch(X, W, Y, B, 1.0f, 0.0f, 0, 1, N, E0, E1, 1, s, &ep, &done); if (rc) return rc;
    }
    if (!done) {
        const long n = (long)N * E0;
        if (layer == T4K_L_DROPOUT) { int rc = t4k_dropout_mask(ACT_F, n, s); if (rc) return rc; }
        return t4k_activate(layer, Y, ACT_O, ACT_F, alpha, n, s);
    }
    T4K_LAUNCH_CHECK();
    return T4K_OK;
}
// linear layer + the element-wise run behind it (one or two layers, no pool) + optionally the model's copy of its input batch: when the
// GEMM is split along K all of it rides in the fold launch; otherwise the same tensors come from the separate launches.
int t4k_linear_block_fwd(const float *X, float *XCOPY, const float *W, const float *B, float *Y, const t4k_poolblock *blk,
                         int N, int E0, int E1, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!X || !W || !Y || N < 0 || E0 < 1 || E1 < 1) return fail(T4K_ERR_ARG, "t4k_linear_block_fwd: bad argument");
    int n1 = 0, n2 = 0; float a1 = 0.f, a2 = 0.f; float *F1 = nullptr, *A1 = nullptr, *F2 = nullptr, *A2 = nullptr;
    if (blk) {
        if (blk->pool_layer || blk->copy_out || blk->KS != 1) return fail(T4K_ERR_UNSUPPORTED, "t4k_linear_block_fwd: the run behind a linear layer has no pool / flatten stage");
        if (blk->pre_layer)  { n1 = blk->pre_layer; a1 = blk->pre_alpha; F1 = blk->pre_mask; A1 = blk->pre_out; }
        if (blk->post_layer) { if (n1) { n2 = blk->post_layer; a2 = blk->post_alpha; F2 = blk->post_mask; A2 = blk->post_out; }
                               else    { n1 = blk->post_layer; a1 = blk->post_alpha; F1 = blk->post_mask; A1 = blk->post_out; } }
        if ((n1 && (!F1 || !A1)) || (n2 && (!F2 || !A2))) return fail(T4K_ERR_ARG, "t4k_linear_block_fwd: stage tensors missing");
        if (n1 == T4K_L_DROPOUT && n2 == T4K_L_DROPOUT) return fail(T4K_ERR_UNSUPPORTED, "t4k_linear_block_fwd: one dropout per run");
    }
    if (N == 0) return T4K_OK;
    bool done = false;
    FoldRider fr = {ActEpi{n2, a2, F2, A2, RngArg{0, 0, nullptr}}, (XCOPY && XCOPY != X) ? X : nullptr, XCOPY, (long)N * E1, 0, 0, MaskChain{nullptr, nullptr, nullptr, nullptr}, 0};
    if (linear_small_ok(E0, E1)) {
        ActEpi ep = { n1, a1, F1, A1, RngArg{0, 0, nullptr} };
        const bool fuse = n1 && !n2 && !(n1 == T4K_L_DROPOUT && st().capturing);     // one stage rides in the head kernel
        if (fuse && n1 == T4K_L_DROPOUT) ep.rng = rng_draw(S(s), (uint64_t)(((long)N * E0 + 3) >> 2), true);
        linear_small_fwd(X, W, B, Y, nullptr, N, E0, E1, S(s), nullptr, fuse ? &ep : nullptr);
        done = fuse;
    } else {
        ActEpi ep = { n1, a1, F1, A1, RngArg{0, 0, nullptr} };
        int rc = gemm_launch(X, W, Y, B, 1.0f, 0.0f, 0, 1, N, E0, E1, 1, s, &ep, &done, nullptr, nullptr, &fr); if (rc) return rc;
    }
    if (XCOPY && !fr.cp_blocks && XCOPY != X) { int rc = t4k_copy(X, XCOPY, (long)N * E1, s); if (rc) return rc; }
    if (!done && n1) {
        const long n = (long)N * E0;
        if (n2) return t4k_poolblock_fwd(Y, blk, N, 1, 1, 1, 1, E0, s);
        if (n1 == T4K_L_DROPOUT) { int rc = t4k_dropout_mask(F1, n, s); if (rc) return rc; }
        return t4k_activate(n1, Y, A1, F1, a1, n, s);
    }
    T4K_LAUNCH_CHECK();
    return T4K_OK;
}
// classifier head: [linear + element-wise layer] + [linear (+ softmax)].  When the first GEMM is split along K and the second
// layer is head-sized, the second layer's kernel folds the slabs, applies bias + activation (writing Y1, F1, A1 as the fold
// launch would) while staging its input rows: two launches instead of three.  Otherwise the two fused entries in sequence.
int t4k_mlp_head_fwd(const float *X, const float *W1, const float *B1, float *Y1, int layer, float alpha, float *F1, float *A1,
                     const float *W2, const float *B2, float *Y2, float *P2, int N, int H, int E1, int E2, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!X || !W1 || !Y1 || !F1 || !A1 || !W2 || !Y2 || N < 0) return fail(T4K_ERR_ARG, "t4k_mlp_head_fwd: bad argument");
    if (N == 0) return T4K_OK;
    static const int head_fold = T4K_LAB_ENV("T4K_HEAD_FOLD", 1);
    if (head_fold && !linear_small_ok(H, E1) && linear_small_ok(E2, H)) {
        XFold xf; xf.part = nullptr;
        int rc = gemm_launch(X, W1, Y1, B1, 1.0f, 0.0f, 0, 1, N, H, E1, 1, s, nullptr, nullptr, nullptr, &xf); if (rc) return rc;
        if (xf.part) {
            xf.bias = B1; xf.Y = Y1; xf.ep = ActEpi{ layer, alpha, F1, A1, RngArg{0, 0, nullptr} };
            if (layer == T4K_L_DROPOUT) xf.ep.rng = rng_draw(S(s), (uint64_t)((xf.mn + 3) >> 2), true);
            linear_small_fwd(A1, W2, B2, Y2, P2, N, E2, H, S(s), &xf);
            T4K_LAUNCH_CHECK(); return T4K_OK;
        }
        const long n = (long)N * H;                         // the GEMM ran unsplit: Y1 is complete, continue layer by layer
        if (layer == T4K_L_DROPOUT) { rc = t4k_dropout_mask(F1, n, s); if (rc) return rc; }
        rc = t4k_activate(layer, Y1, A1, F1, alpha, n, s); if (rc) return rc;
    } else {
        int rc = t4k_linear_act_fwd(X, W1, B1, Y1, layer, alpha, F1, A1, N, H, E1, s); if (rc) return rc;
    }
    if (P2) return t4k_linear_softmax_fwd(A1, W2, B2, Y2, P2, N, E2, H, s);
    return t4k_linear_fwd(A1, W2, B2, Y2, N, E2, H, s);
}
// linear followed by a softmax layer: Y = X W^T + b, P = softmax(Y); one launch when the head is small
int t4k_linear_softmax_fwd(const float *X, const float *W, const float *B, float *Y, float *P, int N, int E0, int E1, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!X || !W || !Y || !P || N < 0) return fail(T4K_ERR_ARG, "t4k_linear_softmax_fwd: bad argument");
    if (N == 0) return T4K_OK;
    if (linear_small_ok(E0, E1)) { linear_small_fwd(X, W, B, Y, P, N, E0, E1, S(s)); T4K_LAUNCH_CHECK(); return T4K_OK; }
    int rc = gemm_launch(X, W, Y, B, 1.0f, 0.0f, 0, 1, N, E0, E1, 1, s); if (rc) return rc;
    return t4k_softmax(Y, P, N, E0, s);
}
// Model::_blinear src/nn/backprop.cu:193-254
int t4k_linear_bwd(const float *X, const float *W, const float *DY, float *DX, float *DW, float *DB,
                   int N, int E0, int E1, int train, t4k_stream_t s) {
    return t4k_linear_bwd2(X, W, DY, DX, nullptr, nullptr, DW, DB, N, E0, E1, train, s);
}
// same, plus the mask-multiply backward of the element-wise layer in front of this linear layer (dropout, relu, ...):
// DXM = DX (*) MASK (`in = out * mask`, _bactivate backprop.cu:256-263) from the same launch when the head is small
int t4k_linear_bwd2(const float *X, const float *W, const float *DY, float *DX, const float *MASK, float *DXM, float *DW, float *DB,
                    int N, int E0, int E1, int train, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if ((MASK == nullptr) != (DXM == nullptr) || (DXM && !DX)) return fail(T4K_ERR_ARG, "t4k_linear_bwd2: MASK and DXM go together (with DX)");
    if (DXM && !(N > 0 && linear_small_ok(E0, E1))) {    // large layer: the GEMM path, then the mask multiply on its own
        int rc = t4k_linear_bwd2(X, W, DY, DX, nullptr, nullptr, DW, DB, N, E0, E1, train, s); if (rc) return rc;
        return t4k_tt_op(T4K_MUL, DX, MASK, DXM, (long)N * E1, s);
    }
    if ((DW == nullptr) != (DB == nullptr)) return fail(T4K_ERR_ARG, "t4k_linear_bwd: DW and DB go together");
    if (N > 0 && linear_small_ok(E0, E1) && linear_small_bwd(X, W, DY, DX, DW, DB, N, E0, E1, train != 0, S(s), MASK, DXM)) { T4K_LAUNCH_CHECK(); return T4K_OK; }
    if (DXM) { int rc = t4k_linear_bwd2(X, W, DY, DX, nullptr, nullptr, DW, DB, N, E0, E1, train, s); if (rc) return rc; return t4k_tt_op(T4K_MUL, DX, MASK, DXM, (long)N * E1, s); }
    if (train && DW && DX && N > 0 && linear_bwd_dual(X, W, DY, DX, DW, DB, N, E0, E1, S(s))) { T4K_LAUNCH_CHECK(); return T4K_OK; }
    if (train && DW) {                                  // DW == NULL: dX only (the caller forks dW|dB to another stream)
        ColSum cs = { DY, DB, N, E0, false };           // dB += sum_n dY rides in the dW launch when the generic kernel runs it
        int rc = gemm_launch(DY, X, DW, nullptr, 1.0f, 1.0f, 1, 0, E0, E1, N, 1, s, nullptr, nullptr, &cs);   // dW += dY^T @ X
        if (rc) return rc;
        if (!cs.done) { rc = colsum_add(DY, DB, N, E0, S(s)); if (rc) return rc; }
    }
    if (!DX) return T4K_OK;                             // DX == NULL: dW|dB only
    return gemm_launch(DY, W, DX, nullptr, 1.0f, 0.0f, 0, 0, N, E1, E0, 1, s);   // dX = dY @ W (may overwrite X)
}

// backprop's `out -= target` (+ the output layer's pass-through copy) folded into the last linear layer's backward
int t4k_loss_linear_bwd(const float *X, const float *W, float *OUT, const float *TGT, float *OUT2, float *DX,
                        const float *MASK, float *DXM, float *DW, float *DB, int N, int E0, int E1, int train, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!X || !W || !OUT || !TGT || N < 0 || E0 < 1 || E1 < 1) return fail(T4K_ERR_ARG, "t4k_loss_linear_bwd: bad argument");
    if ((MASK == nullptr) != (DXM == nullptr) || (DXM && !DX)) return fail(T4K_ERR_ARG, "t4k_loss_linear_bwd: MASK and DXM go together (with DX)");
    if ((DW == nullptr) != (DB == nullptr)) return fail(T4K_ERR_ARG, "t4k_loss_linear_bwd: DW and DB go together");
    if (N == 0) return T4K_OK;
    if (linear_small_ok(E0, E1) && linear_small_bwd(X, W, OUT, DX, DW, DB, N, E0, E1, train != 0, S(s), MASK, DXM, TGT, OUT2)) { T4K_LAUNCH_CHECK(); return T4K_OK; }
    int rc = t4k_tt_op2(T4K_SUB, OUT, TGT, OUT, OUT2, (long)N * E0, s); if (rc) return rc;
    return t4k_linear_bwd2(X, W, OUT, DX, MASK, DXM, DW, DB, N, E0, E1, train, s);
}

// linear backward + the backward of the element-wise run IN FRONT of the layer (the run that produced X), + optionally backprop's
// `out -= target` start: one launch when the shapes allow (vector-ALU head kernel, or the dual dW || dX launch), else layer by layer
int t4k_linear_block_bwd(const float *X, const float *W, float *DY, const float *TGT, float *DY2, float *DX, const t4k_poolblock *blk, float *XRUN,
                         float *DW, float *DB, int N, int E0, int E1, int train, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!X || !W || !DY || !DX || !blk || !XRUN || N < 0 || E0 < 1 || E1 < 1) return fail(T4K_ERR_ARG, "t4k_linear_block_bwd: bad argument");
    if ((DW == nullptr) != (DB == nullptr)) return fail(T4K_ERR_ARG, "t4k_linear_block_bwd: DW and DB go together");
    if (blk->pool_layer || blk->copy_out || blk->KS != 1 || (!blk->pre_layer && !blk->post_layer)) return fail(T4K_ERR_UNSUPPORTED, "t4k_linear_block_bwd: the run in front of a linear layer has no pool / flatten stage");
    if ((blk->pre_layer && (!blk->pre_mask || (blk->post_layer && !blk->pre_out))) || (blk->post_layer && !blk->post_mask)) return fail(T4K_ERR_ARG, "t4k_linear_block_bwd: stage tensors missing");
    if (N == 0) return T4K_OK;
    MaskChain mc = {nullptr, nullptr, nullptr, nullptr};
    if (blk->post_layer) { mc.m1 = blk->post_mask; mc.d1 = blk->pre_layer ? blk->pre_out : XRUN; if (blk->pre_layer) { mc.m2 = blk->pre_mask; mc.d2 = XRUN; } }
    else                 { mc.m1 = blk->pre_mask; mc.d1 = XRUN; }
    hipStream_t hs = S(s);
    if (linear_small_ok(E0, E1) &&
        linear_small_bwd(X, W, DY, DX, DW, DB, N, E0, E1, train != 0, hs, mc.m1, mc.d1, TGT, DY2, mc.m2, mc.d2)) { T4K_LAUNCH_CHECK(); return T4K_OK; }
    if (TGT) { int rc = t4k_tt_op2(T4K_SUB, DY, TGT, DY, DY2, (long)N * E0, s); if (rc) return rc; }
    if (linear_small_ok(E0, E1) &&                              // the head kernel could not take the target (dX apart from X): masks still ride
        linear_small_bwd(X, W, DY, DX, DW, DB, N, E0, E1, train != 0, hs, mc.m1, mc.d1, nullptr, nullptr, mc.m2, mc.d2)) { T4K_LAUNCH_CHECK(); return T4K_OK; }
    if (train && DW && linear_bwd_dual(X, W, DY, DX, DW, DB, N, E0, E1, hs, &mc)) { T4K_LAUNCH_CHECK(); return T4K_OK; }
    if (!(train && DW) && !linear_small_ok(E0, E1)) {       // dX only (a frozen net in the middle of a chain): the mask chain rides in the GEMM's fold launch
        FoldRider fr = {ActEpi{0, 0.f, nullptr, nullptr, RngArg{0, 0, nullptr}}, nullptr, nullptr, 0, 0, 0, mc, 0};
        int rc = gemm_launch(DY, W, DX, nullptr, 1.0f, 0.0f, 0, 0, N, E1, E0, 1, s, nullptr, nullptr, nullptr, nullptr, &fr); if (rc) return rc;
        if (fr.mc_done) return T4K_OK;
        return t4k_poolblock_bwd(DX, XRUN, blk, N, 1, 1, 1, 1, E1, s);
    }
    int rc = t4k_linear_bwd2(X, W, DY, DX, nullptr, nullptr, DW, DB, N, E0, E1, train, s); if (rc) return rc;
    return t4k_poolblock_bwd(DX, XRUN, blk, N, 1, 1, 1, 1, E1, s);
}

// Classifier-head backward AND the backward of the linear layer in front of it in ONE launch (k_head_bwd_l32): what
// t4k_loss_linear_bwd(X2, W2, P, TGT, Y2, X2, MASK, Y1, DW2, DB2, N, EB, EA, 1) followed by t4k_linear_bwd(X1, W1, Y1, X1, DW1, DB1, N, EA, E1, 1)
// compute (backprop.cu:103-121, 226-254; gradients accumulate), for a training pass with the in-place convention.  T4K_ERR_UNSUPPORTED when the shapes
// do not qualify - the caller then makes the two calls.
static size_t head_bwd_lds(int N, int EA, int EB) {                // dynamic LDS of k_head_bwd_l32: the larger of a tile's and a rider's
    const size_t kd = (size_t)((std::max(N, EA) + 31) / 32) * 32;
    const size_t lt = 8192 + kd * HB_AP, lr = (size_t)N * EB + (size_t)EB * HB_CW + (size_t)N * HB_CW + 256;
    return sizeof(float) * std::max(std::max(lt, lr), (size_t)N * EB);
}
int t4k_mlp_head_bwd_ok(int N, int E1, int EA, int EB) {
    if (!st().ready) return 0;
    static const int on = T4K_LAB_ENV("T4K_HEAD_BWD", 1);
    if (!on || st().capturing || !st().d_sync || !st().d_zero || !dual_on()) return 0;
    if (N < 1 || N > 256 || EA < 4 || EA > 256 || (EA & 3) || EB < 1 || EB > 16 || E1 < 4 || (E1 & 3)) return 0;
    auto t32 = [](int m, int n) { return (long)((m + 31) / 32) * ((n + 31) / 32); };
    const long a1 = t32(EA, E1), a2 = t32(N, E1), nr = 1 + (EA + HB_CW - 1) / HB_CW;
    const size_t lds = head_bwd_lds(N, EA, EB);
    int nb = 0;                                                   // resident workgroups per CU at this launch's LDS request
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_head_bwd_l32, 256, lds) != hipSuccess || nb < 1) return 0;
    return (a1 + a2 + nr <= (long)nb * st().cu_count - 32 && a1 <= 512 && a1 + a2 + nr <= 768) ? 1 : 0;      // every workgroup resident (the in-place gate and the target store spin); 768 per-workgroup slots
}
static int head_bwd_launch(float *X2, const float *W2, float *P, const float *TGT, float *Y2, const MaskChain &mc2, float *DW2, float *DB2,
                           float *X1, const float *W1, const MaskChain &mc1, float *DW1, float *DB1, int N, int E1, int EA, int EB, bool train, hipStream_t hs, const char *who) {
    int *gate = gate_for(hs, 0);
    float *DY1 = mc2.d2 ? mc2.d2 : mc2.d1;                        // what the big layer differentiates against: the product of all masks
    if (!t4k_mlp_head_bwd_ok(N, E1, EA, EB) || !gate || capturing(hs) || !aligned16(X1) || !aligned16(W1) || !aligned16(DY1))
        return fail(T4K_ERR_UNSUPPORTED, "%s: shapes do not qualify (t4k_mlp_head_bwd_ok)", who);
    State &g = st();
    auto t32 = [](int m, int n) { return (long)((m + 31) / 32) * ((n + 31) / 32); };
    const long a1 = train ? t32(EA, E1) : 0, a2 = t32(N, E1), nr = 1 + (EA + HB_CW - 1) / HB_CW;
    GemmP q1, q2;
    auto fill32 = [&](GemmP &p, const float *A, const float *B, float *O, int M, int Nn, int K, float beta) {
        p.A = A; p.B = B; p.bias = nullptr; p.O = O; p.part = nullptr; p.M = M; p.N = Nn; p.K = K; p.C = 1;
        p.tiles_m = (M + 31) / 32; p.tiles_n = (Nn + 31) / 32; p.kchunk = K; p.nsplit = 1;
        p.alpha = 1.0f; p.beta = beta; p.pair = 0; p.sync = g.d_sync; p.cs_X = nullptr; p.cs_out = nullptr; p.cs_rows = 0; p.cs_E = 0; p.xmap = 0; p.Z = g.d_zero;
    };
    fill32(q1, DY1, X1, DW1, EA, E1, N, 1.0f);                    // dW1 += dY1^T X1   (A is prepared in LDS: the pointer is not read)
    fill32(q2, DY1, W1, X1, N, E1, EA, 0.0f);                     // dX1 = dY1 W1, over X1 (backprop.cu:240)
    // arrival slots, all tagged with the lane's ONE epoch counter (shared with k_gemm_dual32): ints [512, 1024) of the lane's gate block for the in-place dX
    // (as linear_bwd_dual; a frozen layer has no dW readers to wait for), ints [1280, 2048) one per workgroup for the `out -= target` store
    unsigned *slots = reinterpret_cast<unsigned *>(gate) + 512, *pslots = reinterpret_cast<unsigned *>(gate) + 1280;
    const unsigned epoch = next_slot_epoch(hs, slots);
    if (!train) slots = nullptr;
    HeadBwd hb = { P, TGT, W2, mc2.m1, X2, DW2, DB2, mc2.d1, Y2, DB1, N, EA, EB, train ? 1 : 0, (int)(a1 + a2 + nr), gate, mc2.m2, mc2.d2, mc1, g.d_zero };
    const size_t lds = head_bwd_lds(N, EA, EB);
    static size_t attr = 0;
    if (lds > attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_head_bwd_l32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = lds; }
    int lab = 0;
#ifdef T4K_LAB
    { const char *e = getenv("T4K_HB_LAB"); lab = e ? atoi(e) : 0; }       // timing ablations (wrong results): see the kernel's HB_LAB bits
#endif
    T4K_LAUNCH(k_head_bwd_l32, dim3((unsigned)(a1 + a2 + nr)), dim3(256), lds, hs, q1, q2, (int)a1, (int)nr, slots, pslots, epoch, hb, lab);
    T4K_LAUNCH_CHECK();
    return T4K_OK;
}
int t4k_mlp_head_bwd(float *X2, const float *W2, float *P, const float *TGT, float *Y2, const float *MASK, float *Y1, float *DW2, float *DB2,
                     float *X1, const float *W1, float *DW1, float *DB1, int N, int E1, int EA, int EB, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!X2 || !W2 || !P || !TGT || !MASK || !Y1 || !DW2 || !DB2 || !X1 || !W1 || !DW1 || !DB1) return fail(T4K_ERR_ARG, "t4k_mlp_head_bwd: null argument");
    const MaskChain mc2 = { MASK, Y1, nullptr, nullptr }, none = { nullptr, nullptr, nullptr, nullptr };
    return head_bwd_launch(X2, W2, P, TGT, Y2, mc2, DW2, DB2, X1, W1, none, DW1, DB1, N, E1, EA, EB, true, S(s), "t4k_mlp_head_bwd");
}
// The same with element-wise RUNS instead of a single mask layer (the GAN nets: `leakyrelu dropout` between the linear layers): run2 stands in front of the
// head layer (it produced X2; its backward lands in its own tensors, the run's input tensor XRUN2 = the big layer's output receives dY1), run1 in front of
// the big layer (NULL: none; its mask multiplies ride in the dX1 epilogue into run1's tensors / XRUN1).  train = 0: a frozen net (no dW / dB, DW* may be NULL).
int t4k_mlp_block_bwd(float *X2, const float *W2, float *P, const float *TGT, float *Y2, const t4k_poolblock *run2, float *XRUN2, float *DW2, float *DB2,
                      float *X1, const float *W1, const t4k_poolblock *run1, float *XRUN1, float *DW1, float *DB1, int N, int E1, int EA, int EB, int train, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!X2 || !W2 || !P || !TGT || !run2 || !XRUN2 || !X1 || !W1) return fail(T4K_ERR_ARG, "t4k_mlp_block_bwd: null argument");
    if (train && (!DW2 || !DB2 || !DW1 || !DB1)) return fail(T4K_ERR_ARG, "t4k_mlp_block_bwd: a training pass needs the gradient tensors");
    auto chain = [&](const t4k_poolblock *b, float *xrun, MaskChain &mc) -> bool {
        mc = MaskChain{nullptr, nullptr, nullptr, nullptr};
        if (!b) return true;
        if (b->pool_layer || b->copy_out || b->KS != 1 || (!b->pre_layer && !b->post_layer) || !xrun) return false;
        if ((b->pre_layer && (!b->pre_mask || (b->post_layer && !b->pre_out))) || (b->post_layer && !b->post_mask)) return false;
        if (b->post_layer) { mc.m1 = b->post_mask; mc.d1 = b->pre_layer ? b->pre_out : xrun; if (b->pre_layer) { mc.m2 = b->pre_mask; mc.d2 = xrun; } }
        else               { mc.m1 = b->pre_mask; mc.d1 = xrun; }
        return true;
    };
    MaskChain mc2, mc1;
    if (!chain(run2, XRUN2, mc2) || !mc2.d1 || !chain(run1, XRUN1, mc1)) return fail(T4K_ERR_UNSUPPORTED, "t4k_mlp_block_bwd: the runs must be one or two mask-multiply layers without pool / flatten");
    return head_bwd_launch(X2, W2, P, TGT, Y2, mc2, DW2, DB2, X1, W1, mc1, DW1, DB1, N, E1, EA, EB, train != 0, S(s), "t4k_mlp_block_bwd");
}

int t4k_gemm_f64acc(const float *A, const float *B, float *O, float alpha, float beta,
                    int M, int N, int K, int C, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!A || !B || !O || M < 0 || N < 0 || K < 0 || C < 1) return fail(T4K_ERR_ARG, "t4k_gemm_f64acc: bad argument");
    const long total = (long)M * N * C; if (total == 0) return T4K_OK;
    T4K_LAUNCH(k_gemm_f64, dim3(grid_for(total)), dim3(BLK), 0, S(s), A, B, O, alpha, beta, M, N, K, C);
    T4K_LAUNCH_CHECK();
    return T4K_OK;
}

} // extern "C"
