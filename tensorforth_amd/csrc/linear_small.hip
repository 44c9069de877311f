// linear_small.hip - classifier-head sized linear layers (E0 <= 64 outputs, E1 <= 512 inputs) on the vector ALUs.
//
// A 100->10 layer on a batch of 128 is 0.26 MFLOP: as an MFMA GEMM it costs three latency-bound launches backward
// (column sum, dW, dX) and two forward (GEMM, softmax), ~5 us each.  Here it is one launch each way:
//   forward  Y = X W^T + b, and - when the next layer is a softmax - P = softmax(Y) from the same registers;
//   backward dB += sum_n dY, dW += dY^T X, dX = dY W.  The reference writes dX over X (backprop.cu:240), so the
//            dX workgroups compute first, then wait on an arrival counter until every dW workgroup has finished
//            reading X, and only then store (all workgroups are co-resident: grid <= CU count).
// Every dot product is one fmaf chain in ascending k, i.e. bit-identical to the CPU oracle's GEMM; the softmax uses
// the same max-shift / __expf / xor-tree as k_softmax (reduce.hip).  Reference: _flinear forward.cu:157-198,
// _blinear backprop.cu:193-254, k_softmax nmath.cu:74-169.
#include "t4k_common.h"
#include <float.h>

using namespace t4k;

namespace {
T4K_SPIN_DECL


constexpr int LS_MAX_FLOATS = 12288;          // 48 KiB of dynamic LDS

// rows per wave = 64 / LG, lanes of a group = output index e0.  NARROW (fold mode): a workgroup owns only ONE wave's worth of rows -
// all 256 threads fold / stage them (4x the workgroups and threads on the slab reads), wave 0 alone computes the few dot products
template <int LG, bool NARROW = false>
__global__ void __launch_bounds__(256) k_linsmall_fwd(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ B,
                                                      float *__restrict__ Y, float *__restrict__ P, int N, int E0, int E1, XFold xf, ActEpi oep) {   // oep: the element-wise layer BEHIND this one (sigmoid of a discriminator head, ...)
    extern __shared__ float sm[];
    constexpr int RPW = 64 / LG, RPB = NARROW ? RPW : 4 * RPW;
    const int ldw = E1 + 1;                                      // odd-ish stride: lanes (= rows of W) hit different banks
    float *Ws = sm, *Xs = sm + E0 * ldw;
    const int tid = threadIdx.x;
    for (int r = tid / 64; r < E0; r += 4)                       // one wave per row of W: coalesced reads
        for (int k = tid & 63; k < E1; k += 64) Ws[r * ldw + k] = W[(long)r * E1 + k];
    const int row0 = blockIdx.x * RPB;
    if (xf.part) {
        // the input rows are still split-K slabs of the layer in front: fold, add its bias, apply its element-wise layer (same
        // arithmetic and Philox positions as k_splitk_fold, gemm.hip) and publish Y / mask / activation for these rows
        uint64_t base = 0, seed = 0;
        const bool draw = xf.ep.layer == T4K_L_DROPOUT;
        if (draw) rng_begin(xf.ep.rng, base, seed);
        for (int e = tid; e < RPB * E1; e += 256) {               // one element per thread and trip, its slab loads all independent
            const int r = e / E1, k = e - r * E1, n = row0 + r;
            {
                float a = 0.f;
                if (n < N) {
                    const long z = (long)n * E1 + k;
                    float v[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) v[j] = xf.part[(long)(j < xf.nsplit ? j : 0) * xf.mn + z];
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < 16; j++) s += (j < xf.nsplit) ? v[j] : 0.f;
                    for (int j = 16; j < xf.nsplit; j++) s += xf.part[(long)j * xf.mn + z];
                    float o = s;
                    if (xf.bias) o += xf.bias[k];
                    xf.Y[z] = o;
                    float f; act_rt(xf.ep.layer, o, draw ? philox_u01_at(base, seed, z) : 0.f, xf.ep.alpha, a, f);
                    xf.ep.F[z] = f; xf.ep.A[z] = a;
                }
                Xs[r * E1 + k] = a;
            }
        }
        if (draw && xf.ep.rng.state) rng_advance_last_block(xf.ep.rng.state, base, (uint64_t)((xf.mn + 3) >> 2));
    } else
    for (int r = tid / 64; r < RPB; r += 4) {
        const int n = row0 + r;
        for (int k = tid & 63; k < E1; k += 64) Xs[r * E1 + k] = n < N ? X[(long)n * E1 + k] : 0.f;
    }
    __syncthreads();
    const int lane = tid & 63, w = tid >> 6, e0 = lane % LG, rloc = (NARROW ? 0 : w * RPW) + lane / LG, n = row0 + rloc;
    const bool live = e0 < E0 && n < N && (!NARROW || w == 0);
    float acc = 0.f;
    if (live) {
        const float *xs = Xs + rloc * E1, *ws = Ws + e0 * ldw;
        for (int k = 0; k < E1; k++) acc = fmaf(xs[k], ws[k], acc);
        acc += B ? B[e0] : 0.f;
        const long z = (long)n * E0 + e0;
        Y[z] = acc;
        if (oep.layer) {
            float u = 0.f;
            if (oep.layer == T4K_L_DROPOUT) { uint64_t ob, os; rng_begin(oep.rng, ob, os); u = philox_u01_at(ob, os, z); }
            float a, f; act_rt(oep.layer, acc, u, oep.alpha, a, f); oep.F[z] = f; oep.A[z] = a;
        }
    }
    if (P) {                                                     // softmax over the LG lanes of this row
        float mx = live ? acc : -FLT_MAX;
#pragma unroll
        for (int off = LG / 2; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        const float e = live ? __expf(acc - mx) : 0.f;
        float s = e;
#pragma unroll
        for (int off = LG / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (live) P[(long)n * E0 + e0] = e / s;
    }
}

// dW row e0: acc[q] += sum over this lane group's rows of dY[n] * X[n, c0 + 256 q]; U independent row loads per trip
template <int QN, int U>
__device__ __forceinline__ void dw_rows(const float *X, const float *dys, float (&acc)[2], int N, int E1, int NG, int ng, int c0) {
#pragma unroll 1
    for (int nb = ng; nb < N; nb += NG * U) {
        float xv[U][QN]; float dv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int n = nb + u * NG;
            const bool ok = n < N;
            dv[u] = ok ? dys[n] : 0.f;
            const float *xr = X + (long)(ok ? n : 0) * E1;
#pragma unroll
            for (int q = 0; q < QN; q++) { const int c = c0 + q * 256; xv[u][q] = xr[c < E1 ? c : 0]; }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < QN; q++) acc[q] = fmaf(dv[u], xv[u][q], acc[q]);
    }
}

// blocks [0, nB): dW | dB for output row e0 = blockIdx.x; blocks [nB, nB+nA): dX for RA rows each
__global__ void __launch_bounds__(256) k_linsmall_bwd(const float *X, const float *__restrict__ W, const float *__restrict__ DY,
                                                      float *DX, float *DW, float *DB, int N, int E0, int E1,
                                                      int nB, int nA, int RA, int *sync, int alias,
                                                      const float *__restrict__ MASK, float *__restrict__ DXM,
                                                      const float *__restrict__ TGT, float *DYW, float *DY2,
                                                      const float *__restrict__ MASKB, float *__restrict__ DXMB, int CBK) {
    // TGT != NULL (alias mode only): dY = DY - TGT is formed while staging (the `out -= target` start of backprop); the dX
    // workgroups store it over DY (= DYW) and into DY2 once every workgroup has staged its share (counter sync[2])
    extern __shared__ float sm[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < nB) {
        // one output row e0 per workgroup: CL lanes cover the E1 columns, the 256 / CL lane groups split the batch; all loads of a
        // trip are independent (U rows in flight per thread - the loop is pure latency, one memory round trip per trip), partial
        // sums meet in LDS.  The bias gradient is the plain sum of the staged dY column (no pass over X needed).
        // CBK > 1 (one or a few output rows, e.g. a discriminator's 256 -> 1 head): a row's E1 columns are spread over CBK workgroups of
        // 64 columns x 4 batch groups, so the batch is walked in ONE trip instead of one workgroup taking four dependent trips
        const int e0 = blockIdx.x / CBK, cb = blockIdx.x - e0 * CBK;
        int CL = 32; while (CL < E1 && CL < 256) CL <<= 1;
        if (CBK > 1) CL = 64;
        const int NG = 256 / CL, lc = tid % CL, c0 = cb * 64 + lc, ng = tid / CL;
        float *dys = sm;                                         // dY[:, e0] for the whole batch
        for (int n = tid; n < N; n += 256) dys[n] = DY[(long)n * E0 + e0] - (TGT ? TGT[(long)n * E0 + e0] : 0.f);
        __syncthreads();
        if (alias && TGT && tid == 0) __hip_atomic_fetch_add(sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // dY column staged
        float acc[2] = {0.f, 0.f};                               // columns c0 and c0 + 256 (E1 > 256)
        if (E1 <= 256 || CBK > 1) dw_rows<1, 64>(X, dys, acc, N, E1, NG, ng, c0);
        else           dw_rows<2, 32>(X, dys, acc, N, E1, NG, ng, c0);
        float bsum = 0.f;                                        // dB: fixed-order sum of the column (thread t: rows t, t+256, ...)
        for (int n = tid; n < N; n += 256) bsum += dys[n];
        __syncthreads();
        if (alias && tid == 0) __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // X fully consumed
        float *red = sm + N;                                     // [NG][CL][2], then [256] for the bias sum
        if (NG > 1) {
#pragma unroll
            for (int q = 0; q < 2; q++) red[(ng * CL + lc) * 2 + q] = acc[q];
            __syncthreads();
            if (ng == 0)
                for (int g2 = 1; g2 < NG; g2++)
#pragma unroll
                    for (int q = 0; q < 2; q++) acc[q] += red[(g2 * CL + lc) * 2 + q];
            __syncthreads();
        }
        red[tid] = bsum;
        __syncthreads();
        if (tid < 64) {                                          // 256 -> 64 -> xor tree, fixed order
            float b = (red[tid] + red[tid + 64]) + (red[tid + 128] + red[tid + 192]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) b += __shfl_xor(b, off, 64);
            if (tid == 0 && cb == 0) DB[e0] += b;
        }
        if (ng == 0) {
            if (c0 < E1) DW[(long)e0 * E1 + c0] += acc[0];
            if (CBK == 1 && c0 + 256 < E1) DW[(long)e0 * E1 + c0 + 256] += acc[1];
        }
        return;
    }
    // ---- dX[n, e1] = sum_e0 dY[n, e0] * W[e0, e1]
    float *Ws = sm, *Ds = sm + E0 * E1;                          // W verbatim, then RA rows of dY
    for (int e = tid; e < E0 * E1; e += 256) Ws[e] = W[e];
    const int row0 = ((int)blockIdx.x - nB) * RA;
    for (int e = tid; e < RA * E0; e += 256) {
        const int n = row0 + e / E0; const long o = (long)n * E0 + e % E0;
        Ds[e] = n < N ? DY[o] - (TGT ? TGT[o] : 0.f) : 0.f;
    }
    __syncthreads();
    if (alias && TGT && tid == 0) __hip_atomic_fetch_add(sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // dY rows staged (counted only where the re-arm below runs: the gate stays zero between launches)
    const int total = RA * E1;
    float out[4];                                                // RA * E1 <= 1024 outputs per block
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int z = tid + q * 256;
        float acc = 0.f;
        if (z < total) {
            const int r = z / E1, c = z - r * E1;
            for (int e0 = 0; e0 < E0; e0++) acc = fmaf(Ds[r * E0 + e0], Ws[e0 * E1 + c], acc);
        }
        out[q] = acc;
    }
    if (alias) {                                                 // DX overwrites X: wait until every dW workgroup is done reading it
        if (tid == 0) {
            T4K_SPIN_WAIT(__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nB, 4);
            if (TGT) T4K_SPIN_WAIT(__hip_atomic_load(sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nA + nB, 5);
        }
        __syncthreads();
        if (TGT)                                                 // every reader of DY has staged: out -= target lands in place
            for (int e = tid; e < RA * E0; e += 256) {
                const int n = row0 + e / E0;
                if (n < N) { const long o = (long)n * E0 + e % E0; DYW[o] = Ds[e]; if (DY2) DY2[o] = Ds[e]; }
            }
    } else if (TGT) {                                            // no dW workgroups in this launch (frozen layer): these rows of DY are read by nobody else
        for (int e = tid; e < RA * E0; e += 256) {
            const int n = row0 + e / E0;
            if (n < N) { const long o = (long)n * E0 + e % E0; DYW[o] = Ds[e]; if (DY2) DY2[o] = Ds[e]; }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int z = tid + q * 256;
        if (z < total) {
            const int r = z / E1, n = row0 + r;
            if (n < N) {
                const long o = (long)n * E1 + (z - r * E1);
                DX[o] = out[q];
                if (DXM) {                                       // the mask-multiply backward of the layer(s) in front (dropout / relu ...)
                    const float g1 = out[q] * MASK[o]; DXM[o] = g1;
                    if (DXMB) DXMB[o] = g1 * MASKB[o];
                }
            }
        }
    }
    if (alias) {                                                 // last dX workgroup re-arms the counters for the next launch
        __syncthreads();
        if (tid == 0) {
            const int t = __hip_atomic_fetch_add(sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == nA - 1) {
                __hip_atomic_store(sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sync + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sync + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// Column-sliced head backward (round 3).  dW[e0, c] = sum_n dY[n, e0] X[n, c] and dX[n, c] = sum_e0 dY[n, e0] W[e0, c] are both separable
// in the column c of the layer input, so a workgroup that owns a slice of CW columns needs nothing from the others: it stages ALL of dY
// (N x E0, with `out -= target` applied on the fly), its slice of X and of W in LDS - one memory round trip - and after ONE local barrier
// computes and stores dW[:, slice] (fmaf chain ascending n), dX[:, slice] over X in place (it alone read those columns) and the mask
// multiplies behind it.  No arrival gate stands between the dW readers and the dX writers any more (k_linsmall_bwd above: two dependent
// agent-scope round trips on the critical path).  Only the in-place `out -= target` store of the loss preparation is shared: every workgroup
// reports once it has staged dY, workgroup 0 stores it when all have (off everybody else's critical path; bounded spin).
constexpr int LSC_CW = 4;
__global__ void __launch_bounds__(256) k_linsmall_bwd_cols(const float *X, const float *__restrict__ W, const float *DY, float *DX, float *DW, float *DB,
                                                           int N, int E0, int E1, int train, int *sync,
                                                           const float *__restrict__ MASK, float *__restrict__ DXM,
                                                           const float *__restrict__ TGT, float *DYW, float *DY2,
                                                           const float *__restrict__ MASKB, float *__restrict__ DXMB) {
    extern __shared__ float sm[];
    constexpr int CW = LSC_CW, ZI = 8;                         // ZI x 256 outputs of dX per pass
    float *dys = sm, *Ws = sm + N * E0, *Xs = Ws + E0 * CW;          // dY [N][E0], W slice [E0][CW], X slice [N][CW]
    float *red = Xs + N * CW;                                  // [G][E0 * CW] partial dW sums
    const int tid = threadIdx.x, c0 = blockIdx.x * CW, cw = min(CW, E1 - c0);
    // the masks of this thread's first ZI outputs are fetched NOW, with the operands (a load behind a may-alias store would cost a round trip each)
    float mk[ZI], mkb[ZI];
#pragma unroll
    for (int k = 0; k < ZI; k++) {
        const int z = tid + k * 256, n = z / CW, c = z - n * CW;
        const bool ok = DXM && z < N * CW && c < cw;
        const long o = (long)n * E1 + c0 + c;
        mk[k] = ok ? MASK[o] : 0.f; mkb[k] = (ok && DXMB) ? MASKB[o] : 0.f;
    }
    {   // staging: EVERY global load of the first PRE x 256 elements of each operand goes out before the first LDS store - a loop per operand
        // (load, store, next operand) made three dependent memory round trips out of what is one
        constexpr int PRE = 6;
        float pd[PRE], pt[PRE], px[PRE], pw = 0.f;
#pragma unroll
        for (int q = 0; q < PRE; q++) { const int i = tid + q * 256; const bool ok = i < N * E0; pd[q] = ok ? DY[i] : 0.f; pt[q] = (ok && TGT) ? TGT[i] : 0.f; }
        if (tid < E0 * CW) { const int e0 = tid / CW, c = tid - e0 * CW; pw = c < cw ? W[(long)e0 * E1 + c0 + c] : 0.f; }
#pragma unroll
        for (int q = 0; q < PRE; q++) { const int i = tid + q * 256, n = i / CW, c = i - n * CW; px[q] = (train && i < N * CW && c < cw) ? X[(long)n * E1 + c0 + c] : 0.f; }
#pragma unroll
        for (int q = 0; q < PRE; q++) { const int i = tid + q * 256; if (i < N * E0) dys[i] = pd[q] - pt[q]; }
        if (tid < E0 * CW) Ws[tid] = pw;
        if (train) {
#pragma unroll
            for (int q = 0; q < PRE; q++) { const int i = tid + q * 256; if (i < N * CW) Xs[i] = px[q]; }
        }
        for (int i = tid + PRE * 256; i < N * E0; i += 256) dys[i] = DY[i] - (TGT ? TGT[i] : 0.f);
        for (int i = tid + 256; i < E0 * CW; i += 256) { const int e0 = i / CW, c = i - e0 * CW; Ws[i] = c < cw ? W[(long)e0 * E1 + c0 + c] : 0.f; }
        if (train)
            for (int i = tid + PRE * 256; i < N * CW; i += 256) { const int n = i / CW, c = i - n * CW; Xs[i] = c < cw ? X[(long)n * E1 + c0 + c] : 0.f; }
    }
    __syncthreads();
    if (TGT && tid == 0) __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // dY (and the target) staged here
    const int nout = E0 * CW, G = min(4, 256 / nout);           // thread groups splitting the batch of one dW output (contiguous ranges, summed in order)
    float dwacc = 0.f;
    if (train && tid < nout * G) {
        const int g = tid / nout, t = tid - g * nout, e0 = t / CW, c = t - e0 * CW;
        const int nb = (N + G - 1) / G, n0 = g * nb, n1 = min(N, n0 + nb);
#pragma unroll 8
        for (int n = n0; n < n1; n++) dwacc = fmaf(dys[n * E0 + e0], Xs[n * CW + c], dwacc);
        if (G > 1) red[g * nout + t] = dwacc;
    }
    if (DX) {
#pragma unroll
        for (int k = 0; k < ZI; k++) {                           // dX[n, c0 + c]: fmaf chain ascending e0 (the oracle's order)
            const int z = tid + k * 256, n = z / CW, c = z - n * CW;
            if (z >= N * CW || c >= cw) continue;
            float acc = 0.f;
            for (int e0 = 0; e0 < E0; e0++) acc = fmaf(dys[n * E0 + e0], Ws[e0 * CW + c], acc);
            const long o = (long)n * E1 + c0 + c;
            DX[o] = acc;                                         // may be X itself: this workgroup staged these columns before the barrier, nobody else reads them
            if (DXM) { const float g1 = acc * mk[k]; DXM[o] = g1; if (DXMB) DXMB[o] = g1 * mkb[k]; }
        }
        for (int z = tid + ZI * 256; z < N * CW; z += 256) {     // batches beyond ZI x 256 / CW rows
            const int n = z / CW, c = z - n * CW;
            if (c >= cw) continue;
            float acc = 0.f;
            for (int e0 = 0; e0 < E0; e0++) acc = fmaf(dys[n * E0 + e0], Ws[e0 * CW + c], acc);
            const long o = (long)n * E1 + c0 + c;
            DX[o] = acc;
            if (DXM) { const float g1 = acc * MASK[o]; DXM[o] = g1; if (DXMB) DXMB[o] = g1 * MASKB[o]; }
        }
    }
    if (train) {
        if (G > 1) __syncthreads();
        if (tid < nout) {
            const int e0 = tid / CW, c = tid - e0 * CW;
            float a = dwacc;
            for (int g = 1; g < G; g++) a += red[g * nout + tid];
            if (c < cw) DW[(long)e0 * E1 + c0 + c] += a;
        } else if (blockIdx.x == 0 && tid - nout < E0) {         // dB[e0] = sum_n dY[n, e0] (k_dlinear_db nmath.cu:274-280), spare threads of workgroup 0
            const int e0 = tid - nout;
            float b = 0.f;
            for (int n = 0; n < N; n++) b += dys[n * E0 + e0];
            DB[e0] += b;
        }
    }
    if (TGT && blockIdx.x == 0) {                                // `out -= target` in place, once every workgroup has read out / target
        if (tid == 0) {
            T4K_SPIN_WAIT(__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x, 6);
            __hip_atomic_store(sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // re-armed for the next launch on this stream
        }
        __syncthreads();
        for (int i = tid; i < N * E0; i += 256) { DYW[i] = dys[i]; if (DY2) DY2[i] = dys[i]; }
    }
}

// ---- thin heads (E0 <= 4 outputs, e.g. a discriminator's 256 -> 1 layer).  The kernels above give an OUTPUT to a lane: with one output
// per row, 4 lanes of a wave work and each walks a 256-step dependent LDS chain (k_linsmall_fwd<16>: 10 us for 0.13 MFLOP).
// Forward: a wave per batch row, the lanes split k (16-byte loads when the rows allow), xor-tree over the wave, lane e finishes output e.
__global__ void __launch_bounds__(256) k_linthin_fwd(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ B,
                                                     float *__restrict__ Y, int N, int E0, int E1, int vec, ActEpi oep) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float *xr = X + (long)n * E1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    typedef float v4 __attribute__((ext_vector_type(4)));
    if (vec) {
        for (int k = lane * 4; k < E1; k += 256) {
            const v4 x = *reinterpret_cast<const v4 *>(xr + k);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (e < E0) { const v4 w = *reinterpret_cast<const v4 *>(W + (long)e * E1 + k);
                              acc[e] = fmaf(x[3], w[3], fmaf(x[2], w[2], fmaf(x[1], w[1], fmaf(x[0], w[0], acc[e])))); }
            }
        }
    } else {
        for (int k = lane; k < E1; k += 64) {
            const float x = xr[k];
#pragma unroll
            for (int e = 0; e < 4; e++) if (e < E0) acc[e] = fmaf(x, W[(long)e * E1 + k], acc[e]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[e] += __shfl_xor(acc[e], off, 64);
    if (lane < E0) {
        float v = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : (lane == 2 ? acc[2] : acc[3]));
        v += B ? B[lane] : 0.f;
        const long z = (long)n * E0 + lane;
        Y[z] = v;
        if (oep.layer) {
            float u = 0.f;
            if (oep.layer == T4K_L_DROPOUT) { uint64_t ob, os; rng_begin(oep.rng, ob, os); u = philox_u01_at(ob, os, z); }
            float a, f; act_rt(oep.layer, v, u, oep.alpha, a, f); oep.F[z] = f; oep.A[z] = a;
        }
    }
}
// Backward: a workgroup owns RA batch rows and every column (thread = column c, c + 256).  Nothing it reads is written by another
// workgroup - its rows of X become its rows of dX in place, its rows of `out` take `out - target` in place - so there is no arrival gate
// (k_linsmall_bwd: the dX writers wait two agent-scope round trips for the dW readers).  What crosses workgroups is the batch sum:
// each one leaves dW | dB partials of its rows in the workspace (agent-scope stores), takes a ticket, and the LAST one adds the partials
// in row order into dW | dB (deterministic; nobody waits for anybody).  All loads of a thread go out before its first store.
template <int RA>
__global__ void __launch_bounds__(256) k_linthin_bwd(const float *X, const float *__restrict__ W, const float *DY, float *DX, float *DW, float *DB,
                                                     int N, int E0, int E1, int train, int *ticket, float *part,
                                                     const float *__restrict__ MASK, float *__restrict__ DXM,
                                                     const float *__restrict__ TGT, float *DYW, float *DY2,
                                                     const float *__restrict__ MASKB, float *__restrict__ DXMB) {
    __shared__ float dys[RA * 4];
    __shared__ int last_s;
    const int tid = threadIdx.x, row0 = blockIdx.x * RA, nr = min(RA, N - row0);
    const int c0 = tid, c1 = tid + 256;
    const bool h0 = c0 < E1, h1 = c1 < E1;
    float w0[4], w1[4], x0[RA], x1[RA], m0[RA], m1[RA], mb0[RA], mb1[RA];
#pragma unroll
    for (int e = 0; e < 4; e++) { w0[e] = (e < E0 && h0) ? W[(long)e * E1 + c0] : 0.f; w1[e] = (e < E0 && h1) ? W[(long)e * E1 + c1] : 0.f; }
#pragma unroll
    for (int r = 0; r < RA; r++) {
        const long o = (long)(row0 + (r < nr ? r : 0)) * E1;
        x0[r] = (train && h0) ? X[o + c0] : 0.f; x1[r] = (train && h1) ? X[o + c1] : 0.f;
        m0[r] = (DXM && h0) ? MASK[o + c0] : 0.f; m1[r] = (DXM && h1) ? MASK[o + c1] : 0.f;
        mb0[r] = (DXMB && h0) ? MASKB[o + c0] : 0.f; mb1[r] = (DXMB && h1) ? MASKB[o + c1] : 0.f;
    }
    float dyv = 0.f;
    if (tid < RA * E0) { const int r = tid / E0; if (r < nr) { const long o = (long)row0 * E0 + tid; dyv = DY[o] - (TGT ? TGT[o] : 0.f); } dys[(tid / E0) * 4 + tid % E0] = dyv; }
    __syncthreads();
    if (TGT && tid < nr * E0) { const long o = (long)row0 * E0 + tid; DYW[o] = dyv; if (DY2) DY2[o] = dyv; }     // `out -= target` (backprop.cu:152-160), these rows are nobody else's
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < RA; r++) {
        if (r >= nr) break;
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) if (e < E0) { const float dy = dys[r * 4 + e]; d0 = fmaf(dy, w0[e], d0); d1 = fmaf(dy, w1[e], d1); a0[e] = fmaf(dy, x0[r], a0[e]); a1[e] = fmaf(dy, x1[r], a1[e]); }
        if (DX) {
            const long o = (long)(row0 + r) * E1;
            if (h0) { DX[o + c0] = d0; if (DXM) { const float g1 = d0 * m0[r]; DXM[o + c0] = g1; if (DXMB) DXMB[o + c0] = g1 * mb0[r]; } }
            if (h1) { DX[o + c1] = d1; if (DXM) { const float g1 = d1 * m1[r]; DXM[o + c1] = g1; if (DXMB) DXMB[o + c1] = g1 * mb1[r]; } }
        }
    }
    if (!train) return;
    // partials of this row group: [wg][E0][E1] then [wg][E0] bias sums behind all of them
    const int G = (int)gridDim.x;
    float *pw = part + (long)blockIdx.x * E0 * E1, *pb = part + (long)G * E0 * E1 + (long)blockIdx.x * E0;
#pragma unroll
    for (int e = 0; e < 4; e++) if (e < E0) {
        if (h0) __hip_atomic_store(pw + (long)e * E1 + c0, a0[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (h1) __hip_atomic_store(pw + (long)e * E1 + c1, a1[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < E0) { float b = 0.f; for (int r = 0; r < nr; r++) b += dys[r * 4 + tid]; __hip_atomic_store(pb + tid, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last_s = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1;
    __syncthreads();
    if (!last_s) return;
    if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // re-armed for the next launch on this stream
    for (int i = tid; i < E0 * E1; i += 256) {                  // every partial's load is independent: one round trip, then the sum in row order
        float s_ = 0.f;
        for (int g = 0; g < G; g += 32) {                       // 32 row groups per trip (N = 256: all of them)
            float v[32];
#pragma unroll
            for (int j = 0; j < 32; j++) v[j] = g + j < G ? __hip_atomic_load(part + (long)(g + j) * E0 * E1 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
            for (int j = 0; j < 32; j++) s_ += v[j];
        }
        DW[i] += s_;
    }
    if (tid < E0) { float b = 0.f; for (int g = 0; g < G; g++) b += __hip_atomic_load(part + (long)G * E0 * E1 + (long)g * E0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); DB[tid] += b; }
}

// Trained thin head: a workgroup owns CW columns of the layer input and ALL batch rows (thread = column x row group, rows interleaved so a
// wave's load covers whole CW-float runs).  dW[:, c] and dX[:, c] need nothing from other columns, the thread that loaded X[n, c] is the
// one that overwrites it with dX[n, c] - no arrival gate, no partials in memory, no ticket on the critical path (the row-group kernel
// above pays three dependent agent-scope round trips for its fold: 11 us; it stays for frozen layers, where nothing crosses workgroups).
// The one shared store is the in-place `out -= target`: every workgroup reads all of out / target, so the LAST one to have staged them
// (a ticket taken after the staging barrier, looked at when the rest of the work is done) stores the difference.
template <int CW>
__global__ void __launch_bounds__(256) k_linthin_bwd_cols(const float *X, const float *__restrict__ W, const float *DY, float *DX, float *DW, float *DB,
                                                          int N, int E0, int E1, int *ticket,
                                                          const float *__restrict__ MASK, float *__restrict__ DXM,
                                                          const float *__restrict__ TGT, float *DYW, float *DY2,
                                                          const float *__restrict__ MASKB, float *__restrict__ DXMB) {
    extern __shared__ float sm[];
    constexpr int NG = 256 / CW, RPT = 8;
    float *dys = sm, *red = sm + N * 4;                         // dY [N][4]; partial dW [NG][CW][4]
    __shared__ int tk_s;
    const int tid = threadIdx.x, c = tid % CW, g = tid / CW, col = blockIdx.x * CW + c;
    const bool hc = col < E1;
    float w[4];
#pragma unroll
    for (int e = 0; e < 4; e++) w[e] = (e < E0 && hc) ? W[(long)e * E1 + col] : 0.f;
    float xv[RPT], mv[RPT], mbv[RPT];
    auto loads = [&](int nb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < RPT; i++) {
            const int n = nb + g + NG * i; const bool ok = hc && n < N; const long o = (long)(ok ? n : 0) * E1 + (hc ? col : 0);
            xv[i] = ok ? X[o] : 0.f; mv[i] = (ok && DXM) ? MASK[o] : 0.f; mbv[i] = (ok && DXMB) ? MASKB[o] : 0.f;
        }
    };
    loads(0);                                                    // in flight while dY is staged
    for (int i = tid; i < N * E0; i += 256) { const float d = DY[i] - (TGT ? TGT[i] : 0.f); dys[(i / E0) * 4 + i % E0] = d; }
    __syncthreads();
    if (TGT && tid == 0) tk_s = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int nb = 0; nb < N; nb += NG * RPT) {
        if (nb) loads(nb);
#pragma unroll
        for (int i = 0; i < RPT; i++) {
            const int n = nb + g + NG * i;
            if (n >= N || !hc) continue;
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) if (e < E0) { const float dy = dys[n * 4 + e]; d = fmaf(dy, w[e], d); acc[e] = fmaf(dy, xv[i], acc[e]); }
            if (DX) {
                const long o = (long)n * E1 + col;
                DX[o] = d;
                if (DXM) { const float g1 = d * mv[i]; DXM[o] = g1; if (DXMB) DXMB[o] = g1 * mbv[i]; }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) red[(g * CW + c) * 4 + e] = acc[e];
    __syncthreads();
    if (tid < CW * E0) {                                         // row groups in order: deterministic
        const int cc = tid % CW, e = tid / CW, oc = blockIdx.x * CW + cc;
        float a = 0.f;
        for (int gg = 0; gg < NG; gg++) a += red[(gg * CW + cc) * 4 + e];
        if (oc < E1) DW[(long)e * E1 + oc] += a;
    } else if (blockIdx.x == 0 && tid >= 192 && tid - 192 < E0) {      // dB[e] = sum_n dY[n, e] (k_dlinear_db nmath.cu:274-280)
        const int e = tid - 192;
        float b = 0.f;
        for (int n = 0; n < N; n++) b += dys[n * 4 + e];
        DB[e] += b;
    }
    if (TGT) {
        __syncthreads();
        if (tk_s == (int)gridDim.x - 1) {                        // everybody has staged out / target: the difference lands in place
            if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = tid; i < N * E0; i += 256) { const float d = dys[(i / E0) * 4 + i % E0]; DYW[i] = d; if (DY2) DY2[i] = d; }
        }
    }
}

} // namespace

namespace t4k {
void linsmall_set_spin_err(int *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_spin_err_dev), &p, sizeof(p)); }


bool linear_small_ok(int E0, int E1) {
    return E0 >= 1 && E0 <= 64 && E1 >= 1 && E1 <= 512 && E0 * (E1 + 1) + 16 * E1 <= LS_MAX_FLOATS && E0 * E1 + 64 * E0 <= LS_MAX_FLOATS;
}

int linear_small_fwd(const float *X, const float *W, const float *B, float *Y, float *P, int N, int E0, int E1, hipStream_t hs, const XFold *xfp, const ActEpi *oepp) {
    const ActEpi oep = oepp ? *oepp : ActEpi{0, 0.f, nullptr, nullptr, RngArg{0, 0, nullptr}};
    XFold xf; if (xfp) xf = *xfp; else { xf.part = nullptr; xf.nsplit = 0; xf.mn = 0; xf.bias = nullptr; xf.Y = nullptr; xf.ep = ActEpi{0, 0.f, nullptr, nullptr, RngArg{0, 0, nullptr}}; }
    static const int thin = T4K_LAB_ENV("T4K_LINTHIN", 1);
    if (thin && !xfp && !P && E0 <= 4 && N > 0) {             // thin head: a wave per row
        const int vec = (E1 % 4 == 0) && aligned16(X) && aligned16(W);
        T4K_LAUNCH(k_linthin_fwd, dim3((N + 3) / 4), dim3(256), 0, hs, X, W, B, Y, N, E0, E1, vec, oep);
        return T4K_OK;
    }
    const int LG = E0 <= 16 ? 16 : (E0 <= 32 ? 32 : 64);
    const int RPB = (xfp ? 1 : 4) * (64 / LG);
    const size_t lds = sizeof(float) * (size_t)(E0 * (E1 + 1) + RPB * E1);
    const dim3 g((N + RPB - 1) / RPB), b(256);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_linsmall_fwd<16>), hipFuncAttributeMaxDynamicSharedMemorySize, LS_MAX_FLOATS * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_linsmall_fwd<32>), hipFuncAttributeMaxDynamicSharedMemorySize, LS_MAX_FLOATS * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_linsmall_fwd<64>), hipFuncAttributeMaxDynamicSharedMemorySize, LS_MAX_FLOATS * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_linsmall_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, LS_MAX_FLOATS * 4);
        attr = true;
    }
    if (xfp) {
        if (LG == 16)      T4K_LAUNCH((k_linsmall_fwd<16, true>), g, b, lds, hs, X, W, B, Y, P, N, E0, E1, xf, oep);
        else if (LG == 32) T4K_LAUNCH((k_linsmall_fwd<32, true>), g, b, lds, hs, X, W, B, Y, P, N, E0, E1, xf, oep);
        else               T4K_LAUNCH((k_linsmall_fwd<64, true>), g, b, lds, hs, X, W, B, Y, P, N, E0, E1, xf, oep);
        return T4K_OK;
    }
    if (LG == 16)      T4K_LAUNCH(k_linsmall_fwd<16>, g, b, lds, hs, X, W, B, Y, P, N, E0, E1, xf, oep);
    else if (LG == 32) T4K_LAUNCH(k_linsmall_fwd<32>, g, b, lds, hs, X, W, B, Y, P, N, E0, E1, xf, oep);
    else               T4K_LAUNCH(k_linsmall_fwd<64>, g, b, lds, hs, X, W, B, Y, P, N, E0, E1, xf, oep);
    return T4K_OK;
}

// returns false when the shape does not qualify (caller falls back to the GEMM path)
bool linear_small_bwd(const float *X, const float *W, const float *DY, float *DX, float *DW, float *DB,
                      int N, int E0, int E1, bool train, hipStream_t hs, const float *MASK, float *DXM, const float *TGT, float *DY2,
                      const float *MASKB, float *DXMB) {
    {   // thin head (E0 <= 4): row-group workgroups, no arrival gate, the last one folds the dW | dB partials
        static const int thin = T4K_LAB_ENV("T4K_LINTHIN", 1);
        constexpr int RA = 8;
        const int G = (N + RA - 1) / RA;
        const bool tr = train && DW;
        int *tk = gate_for(hs, 2);                               // ints 8.. of the stream's gate block: the ticket (zero between launches)
        float *part = ws_for(hs) ? ws_for(hs) + st().ws_bytes / 8 : nullptr;      // second half of the stream's workspace (transient column-sum partials; the first half may hold a conv stack's deferred dF partials)
        static const int tcw = T4K_LAB_ENV("T4K_LINTHIN_CW", 8);
        if (thin && tr && tcw && E0 <= 4 && N >= 1 && N <= 1024 && DB && (!TGT || tk) && E1 >= 8) {          // trained: column stripes, nothing crosses workgroups
            const size_t ldsb = sizeof(float) * ((size_t)N * 4 + 256 * 4);
            if (tcw == 16) T4K_LAUNCH(k_linthin_bwd_cols<16>, dim3((E1 + 15) / 16), dim3(256), ldsb, hs, X, W, DY, DX, DW, DB, N, E0, E1, tk, MASK, DXM, TGT, const_cast<float *>(DY), DY2, MASKB, DXMB);
            else           T4K_LAUNCH(k_linthin_bwd_cols<8>,  dim3((E1 + 7) / 8),   dim3(256), ldsb, hs, X, W, DY, DX, DW, DB, N, E0, E1, tk, MASK, DXM, TGT, const_cast<float *>(DY), DY2, MASKB, DXMB);
            return true;
        }
        if (thin && E0 <= 4 && E1 <= 512 && N >= 1 && (DX || tr) && (!tr || (DB && tk && part && (size_t)(G + 1) * E0 * E1 * sizeof(float) <= st().ws_bytes / 2))) {
            T4K_LAUNCH(k_linthin_bwd<RA>, dim3(G), dim3(256), 0, hs, X, W, DY, DX, DW, DB, N, E0, E1, tr ? 1 : 0, tk, part,
                       MASK, DXM, TGT, const_cast<float *>(DY), DY2, MASKB, DXMB);
            return true;
        }
    }
    {   // column-sliced kernel: batches that fit LDS whole (N x (E0 + 16) floats), every output of dW in one thread (E0 x 16 <= 256)
        static const int cols_on = T4K_LAB_ENV("T4K_LINSMALL_COLS", 1);
        const size_t ldsc = sizeof(float) * ((size_t)N * E0 + (size_t)E0 * LSC_CW + (size_t)N * LSC_CW + 256);
        int *gatec = TGT ? gate_for(hs, 0) : nullptr;            // the one shared counter (ints 0.. of the stream's gate block; zero between launches)
        const int nwg = (E1 + LSC_CW - 1) / LSC_CW;
        if (cols_on && (gates_ok() || !TGT) && E0 * LSC_CW + E0 <= 256 && N >= 1 && N <= 512 && ldsc <= (size_t)LS_MAX_FLOATS * 4 && (DX || (train && DW)) && (!train || (DW && DB)) &&
            (!TGT || (gatec && st().d_sync && nwg <= st().cu_count)) && (cols_on >= 2 || N * E1 <= 32768)) {
            static bool attrc = false;
            if (!attrc) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_linsmall_bwd_cols), hipFuncAttributeMaxDynamicSharedMemorySize, LS_MAX_FLOATS * 4); attrc = true; }
            T4K_LAUNCH(k_linsmall_bwd_cols, dim3(nwg), dim3(256), ldsc, hs, X, W, DY, DX, DW, DB, N, E0, E1, (train && DW) ? 1 : 0, gatec,
                               MASK, DXM, TGT, const_cast<float *>(DY), DY2, MASKB, DXMB);
            return true;
        }
    }
    // dW: one workgroup per output row walks the batch in trips of (256 / CL) x 64 rows; when that takes more than one trip, the row's
    // columns are split over workgroups of 64 columns x 4 batch groups instead (a 256 -> 1 head: 1 workgroup x 4 trips -> 4 x 1)
    int CLh = 32; while (CLh < E1 && CLh < 256) CLh <<= 1;
    const int trips = (N + (256 / CLh) * 64 - 1) / ((256 / CLh) * 64);
    const int CBK = (E1 > 64 && trips > 1 && E0 * ((E1 + 63) / 64) <= 128) ? (E1 + 63) / 64 : 1;
    const int nB = (train && DW) ? E0 * CBK : 0;
    int RA = 1024 / E1; if (RA > 64) RA = 64; if (RA < 1) RA = 1;        // rows of dX per workgroup (<= 1024 outputs, <= 64 rows of dY in LDS)
    const int nA = DX ? (N + RA - 1) / RA : 0;
    if (nA + nB == 0) return true;
    const bool alias = DX && nB > 0 && (const float *)DX == X;
    if (TGT && !alias && nB > 0) return false;                             // the in-place `out -= target` needs the arrival counters (unless no dW workgroup reads dY: frozen layer)
    State &g = st();
    static const int gate_on = T4K_LAB_ENV("T4K_LINSMALL_GATE", 1);
    int *gate = gate_for(hs, 0);                                           // nullptr: a stream the library does not know -> no private counters
    if (alias && (nA + nB > g.cu_count || !g.d_sync || !gate || !gate_on || !gates_ok())) return false;   // the arrival counter needs every workgroup resident
    size_t lds = sizeof(float) * (size_t)(E0 * E1 + RA * E0);
    if (nB > 0 && sizeof(float) * (size_t)(N + 768) > lds) lds = sizeof(float) * (size_t)(N + 768);
    if (lds > (size_t)LS_MAX_FLOATS * 4) return false;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_linsmall_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, LS_MAX_FLOATS * 4); attr = true; }
    T4K_LAUNCH(k_linsmall_bwd, dim3(nA + nB), dim3(256), lds, hs, X, W, DY, DX, DW, DB, N, E0, E1, nB, nA, RA, gate, alias ? 1 : 0, MASK, DXM, TGT, const_cast<float *>(DY), DY2, MASKB, DXMB, CBK);
    return true;
}

} // namespace t4k
