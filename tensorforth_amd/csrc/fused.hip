// fused.hip - layer runs fused into one launch (MI355X-native; the reference launches each layer separately).
//
// A dependent kernel launch costs ~4.5 us on MI355X (measured, profiles/), more than most of these
// layers' own HBM time, so the host fuses the element-wise runs around a pooling layer:
//
//   forward   X --[pre: dropout | activation]--> P --[pool KSxKS]--> Q --[post: activation]--> R --[flatten copy]--> R2
//   backward  the same run in reverse: copy, mask multiply, pool scatter (first extreme wins), mask multiply
//
// Every tensor the unfused layers would have written (P, Q, R, R2, both derivative masks) is still written,
// with bit-identical values: the arithmetic is k_activate (nmath.cu:37-70), k_pool / k_dpool
// (nmath.tcu:122-186, 475-568) and, for dropout, the same Philox slice t4k_rand would have drawn.
#include "t4k_common.h"
#include <algorithm>

using namespace t4k;

namespace {

struct PB {                                   // device copy of t4k_poolblock + geometry
    const float *X; float *P, *Q, *R, *R2, *Fpre, *Fpost;
    int pre, pool, post; float a_pre, a_post;
    int N, H1, W1, H0, W0, C;
    RngArg rng, rng2;                        // Philox slices of a dropout pre-stage / post-stage (one run holds at most one dropout)
    float *XH, *BO; const float *bnW, *bnB, *bnS;    // BN form (k_poolblock_fwd<.., true>): X is a conv output, the run starts with the batch-norm apply - x-hat, its output
};

// VW channels per thread (1, 2 or 4; C % VW == 0 and 4*VW-byte aligned tensors): vector loads / stores, and one
// Philox call serves all VW dropout draws of a pixel (they share the 4-element counter block).
template <int VW> struct Vec { float v[VW]; };
template <int VW> __device__ __forceinline__ Vec<VW> vload(const float *p) {
    Vec<VW> r;
    if (VW == 4)      { const float4 t = *reinterpret_cast<const float4 *>(p); r.v[0] = t.x; r.v[1 % VW] = t.y; r.v[2 % VW] = t.z; r.v[3 % VW] = t.w; }
    else if (VW == 2) { const float2 t = *reinterpret_cast<const float2 *>(p); r.v[0] = t.x; r.v[1 % VW] = t.y; }
    else              r.v[0] = *p;
    return r;
}
template <int VW> __device__ __forceinline__ void vstore(float *p, const Vec<VW> &r) {
    if (VW == 4)      *reinterpret_cast<float4 *>(p) = make_float4(r.v[0], r.v[1 % VW], r.v[2 % VW], r.v[3 % VW]);
    else if (VW == 2) *reinterpret_cast<float2 *>(p) = make_float2(r.v[0], r.v[1 % VW]);
    else              *p = r.v[0];
}

// BN: the run is preceded by the apply half of a batch-norm layer (statistics already finalised in bnS: [0, C) 1 / sigma, [C, 2C) mean): every element of X
// is read ONCE, x-hat and the batch-norm output are written as k_bn_apply writes them (same expressions), and the run goes on from the value in registers.
template <int KS, int VW, bool BN = false>
__global__ void __launch_bounds__(BLK) k_poolblock_fwd(PB p) {
    const int CV = p.C / VW;
    const long total = (long)p.N * p.H0 * p.W0 * CV;
    uint64_t base = 0, seed = 0;
    const bool draw = p.pre == T4K_L_DROPOUT, draw2 = p.post == T4K_L_DROPOUT;
    if (draw) rng_begin(p.rng, base, seed);
    if (draw2) rng_begin(p.rng2, base, seed);
    for (long z = (long)blockIdx.x * blockDim.x + threadIdx.x; z < total; z += (long)gridDim.x * blockDim.x) {
        int c, j0, i0, n; long t; split2(z, CV, c, t); c *= VW; split3(t, p.W0, p.H0, j0, i0, n);
        Vec<VW> acc; bool first = true;
#pragma unroll
        for (int q = 0; q < VW; q++) acc.v[q] = 0.f;
#pragma unroll
        for (int y = 0; y < KS; y++)
#pragma unroll
            for (int x = 0; x < KS; x++) {
                const int gi = i0 * KS + y, gj = j0 * KS + x;
                if (gi >= p.H1 || gj >= p.W1) continue;
                const long a = (((long)n * p.H1 + gi) * p.W1 + gj) * p.C + c;
                Vec<VW> e = vload<VW>(p.X + a);
                if (BN) {
                    Vec<VW> xh;
#pragma unroll
                    for (int q = 0; q < VW; q++) { xh.v[q] = (e.v[q] - p.bnS[p.C + c + q]) * p.bnS[c + q]; e.v[q] = xh.v[q] * p.bnW[c + q] + p.bnB[c + q]; }
                    vstore<VW>(p.XH + a, xh); vstore<VW>(p.BO + a, e);
                }
                if (p.pre) {
                    Vec<VW> o, f;
                    uint32_t r[4] = {0, 0, 0, 0};
                    if (draw) philox4x32_10(base + (uint64_t)(a >> 2), seed, r);   // a .. a+VW-1 sit in one counter block
#pragma unroll
                    for (int q = 0; q < VW; q++) act_rt(p.pre, e.v[q], draw ? u01(r[(a + q) & 3]) : 0.f, p.a_pre, o.v[q], f.v[q]);
                    vstore<VW>(p.Fpre + a, f); vstore<VW>(p.P + a, o); e = o;
                }
#pragma unroll
                for (int q = 0; q < VW; q++) {
                    if (p.pool == T4K_L_MAXPOOL)      acc.v[q] = first ? e.v[q] : fmaxf(e.v[q], acc.v[q]);
                    else if (p.pool == T4K_L_MINPOOL) acc.v[q] = first ? e.v[q] : fminf(e.v[q], acc.v[q]);
                    else                              acc.v[q] += e.v[q];
                }
                first = false;
            }
        const long zo = (((long)n * p.H0 + i0) * p.W0 + j0) * p.C + c;
        if (p.pool == T4K_L_AVGPOOL) {
#pragma unroll
            for (int q = 0; q < VW; q++) acc.v[q] /= (float)(KS * KS);
        }
        if (p.pool) vstore<VW>(p.Q + zo, acc);
        if (p.post) {
            Vec<VW> o, f;
            uint32_t r[4] = {0, 0, 0, 0};
            if (draw2) philox4x32_10(base + (uint64_t)(zo >> 2), seed, r);          // zo .. zo+VW-1 sit in one counter block
#pragma unroll
            for (int q = 0; q < VW; q++) act_rt(p.post, acc.v[q], draw2 ? u01(r[(zo + q) & 3]) : 0.f, p.a_post, o.v[q], f.v[q]);
            vstore<VW>(p.Fpost + zo, f); vstore<VW>(p.R + zo, o); acc = o;
        }
        if (p.R2) vstore<VW>(p.R2 + zo, acc);
    }
    if (draw && p.rng.state) rng_advance_last_block(p.rng.state, base, (uint64_t)(((long)p.N * p.H1 * p.W1 * p.C + 3) >> 2));
    if (draw2 && p.rng2.state) rng_advance_last_block(p.rng2.state, base, (uint64_t)(((long)p.N * p.H0 * p.W0 * p.C + 3) >> 2));
}

// backward: DY = gradient w.r.t. the run's last tensor.  Writes (reference in-place convention: each layer's
// input buffer receives its dX): R2-side copy -> R buffer, R*Fpost -> Q buffer, pool scatter -> P buffer (which
// still holds the forward values needed to find the extreme), P*Fpre -> X buffer.
struct PBB {
    const float *DY; float *Rb, *Qb, *Pb, *Xb; const float *Fpre, *Fpost;
    int pre, pool, post;
    int N, H1, W1, H0, W0, C;
};
template <int KS, int VW>
__global__ void __launch_bounds__(BLK) k_poolblock_bwd(PBB p) {
    const int CV = p.C / VW;
    const long total = (long)p.N * p.H0 * p.W0 * CV;
    for (long z = (long)blockIdx.x * blockDim.x + threadIdx.x; z < total; z += (long)gridDim.x * blockDim.x) {
        int c, j0, i0, n; long t; split2(z, CV, c, t); c *= VW; split3(t, p.W0, p.H0, j0, i0, n);
        const long zo = (((long)n * p.H0 + i0) * p.W0 + j0) * p.C + c;
        Vec<VW> g = vload<VW>(p.DY + zo);
        if (p.Rb) vstore<VW>(p.Rb + zo, g);                       // flatten: in = out
        if (p.post) {                                             // activation: in = out (*) mask
            const Vec<VW> f = vload<VW>(p.Fpost + zo);
#pragma unroll
            for (int q = 0; q < VW; q++) g.v[q] *= f.v[q];
            vstore<VW>(p.Qb + zo, g);
        }
        if (!p.pool) {                                            // no pooling in this run (KS == 1)
            if (p.pre) {
                const Vec<VW> f = vload<VW>(p.Fpre + zo);
                Vec<VW> o;
#pragma unroll
                for (int q = 0; q < VW; q++) o.v[q] = g.v[q] * f.v[q];
                vstore<VW>(p.Xb + zo, o);
            }
            continue;
        }
        float best[VW]; int arg[VW];
        Vec<VW> ev[KS * KS]; long av[KS * KS];
#pragma unroll
        for (int q = 0; q < VW; q++) { best[q] = 0.f; arg[q] = -1; }
#pragma unroll
        for (int y = 0; y < KS; y++)
#pragma unroll
            for (int x = 0; x < KS; x++) {
                const int w = y * KS + x;
                const int gi = i0 * KS + y, gj = j0 * KS + x;
                av[w] = -1;
                if (gi >= p.H1 || gj >= p.W1) continue;
                const long a = (((long)n * p.H1 + gi) * p.W1 + gj) * p.C + c;
                av[w] = a;
                if (p.pool != T4K_L_AVGPOOL) {
                    ev[w] = vload<VW>(p.Pb + a);
#pragma unroll
                    for (int q = 0; q < VW; q++) {
                        const float e = ev[w].v[q];
                        const bool better = (p.pool == T4K_L_MAXPOOL) ? (e > best[q]) : (e < best[q]);
                        if (arg[q] < 0 || better) { best[q] = e; arg[q] = w; }   // first extreme wins
                    }
                }
            }
#pragma unroll
        for (int w = 0; w < KS * KS; w++) {
            const long a = av[w]; if (a < 0) continue;
            Vec<VW> d;
#pragma unroll
            for (int q = 0; q < VW; q++) d.v[q] = (p.pool == T4K_L_AVGPOOL) ? g.v[q] / (float)(KS * KS) : (arg[q] == w ? g.v[q] : 0.f);
            vstore<VW>(p.Pb + a, d);
            if (p.pre) {
                const Vec<VW> f = vload<VW>(p.Fpre + a);
#pragma unroll
                for (int q = 0; q < VW; q++) d.v[q] *= f.v[q];
                vstore<VW>(p.Xb + a, d);
            }
        }
    }
}

bool is_act(int l)  { return l == T4K_L_RELU || l == T4K_L_TANH || l == T4K_L_SIGMOID || l == T4K_L_SELU || l == T4K_L_LEAKYRL || l == T4K_L_ELU || l == T4K_L_DROPOUT; }
bool is_pool(int l) { return l == T4K_L_AVGPOOL || l == T4K_L_MAXPOOL || l == T4K_L_MINPOOL; }

// widest channel vector every tensor of the run supports (C % VW == 0, all pointers 4*VW-byte aligned)
int vec_width(int C, const float *X, const t4k_poolblock *b) {
    for (int vw = 4; vw > 1; vw >>= 1) {
        if (C % vw) continue;
        const uintptr_t m = (uintptr_t)(4 * vw - 1);
        const void *ptrs[] = { X, b->pre_mask, b->pre_out, b->pool_out, b->post_mask, b->post_out, b->copy_out };
        bool ok = true;
        for (const void *q : ptrs) if (q && (((uintptr_t)q) & m)) ok = false;
        if (ok) return vw;
    }
    return 1;
}
int check_block(const t4k_poolblock *b, const char *who) {
    if (!b) return fail(T4K_ERR_ARG, "%s: null block", who);
    if (b->pre_layer && !is_act(b->pre_layer))   return fail(T4K_ERR_UNSUPPORTED, "%s: pre layer %d", who, b->pre_layer);
    if (b->post_layer && (!is_act(b->post_layer) || (b->post_layer == T4K_L_DROPOUT && b->pre_layer == T4K_L_DROPOUT))) return fail(T4K_ERR_UNSUPPORTED, "%s: post layer %d", who, b->post_layer);   // one dropout per run
    if (b->pool_layer && !is_pool(b->pool_layer)) return fail(T4K_ERR_UNSUPPORTED, "%s: pool layer %d", who, b->pool_layer);
    if (b->pool_layer ? (b->KS != 2 && b->KS != 3) : (b->KS != 1)) return fail(T4K_ERR_UNSUPPORTED, "%s: kernel_size=%d not supported", who, b->KS);
    if (b->pre_layer && (!b->pre_mask || !b->pre_out))    return fail(T4K_ERR_ARG, "%s: pre tensors missing", who);
    if (b->post_layer && (!b->post_mask || !b->post_out)) return fail(T4K_ERR_ARG, "%s: post tensors missing", who);
    if (b->pool_layer && !b->pool_out)                    return fail(T4K_ERR_ARG, "%s: pool output missing", who);
    return T4K_OK;
}

} // namespace

extern "C" {

static int poolblock_fwd_impl(const float *X, const t4k_poolblock *b, int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t s,
                             float *XH, float *BO, const float *bnW, const float *bnB, const float *bnS) {
    T4K_REQUIRE_INIT();
    int rc = check_block(b, "t4k_poolblock_fwd"); if (rc) return rc;
    if (!X) return fail(T4K_ERR_ARG, "t4k_poolblock_fwd: null input");
    const long total = (long)N * H0 * W0 * C; if (total <= 0) return T4K_OK;
    PB p;
    p.X = X; p.P = b->pre_out; p.Q = b->pool_out; p.R = b->post_out; p.R2 = b->copy_out; p.Fpre = b->pre_mask; p.Fpost = b->post_mask;
    p.pre = b->pre_layer; p.pool = b->pool_layer; p.post = b->post_layer; p.a_pre = b->pre_alpha; p.a_post = b->post_alpha;
    p.N = N; p.H1 = H1; p.W1 = W1; p.H0 = H0; p.W0 = W0; p.C = C;
    p.XH = XH; p.BO = BO; p.bnW = bnW; p.bnB = bnB; p.bnS = bnS;
    p.rng = RngArg{0, 0, nullptr};
    p.rng2 = RngArg{0, 0, nullptr};
    if (b->pre_layer == T4K_L_DROPOUT) p.rng = rng_draw(S(s), (uint64_t)(((long)N * H1 * W1 * C + 3) >> 2), true);
    if (b->post_layer == T4K_L_DROPOUT) p.rng2 = rng_draw(S(s), (uint64_t)((total + 3) >> 2), true);   // the slice t4k_dropout_mask would draw for the post tensor
    int VW = vec_width(C, X, b);
    if (XH) { const uintptr_t m = (uintptr_t)(4 * VW - 1); if ((((uintptr_t)XH) | ((uintptr_t)BO)) & m) VW = 1; }
    const long nthr = total / VW;
    const int bs = (nthr < (long)BLK * 2 * st().cu_count) ? 64 : BLK;      // small runs: one-wave workgroups reach every CU
    const dim3 grid((unsigned)std::min<long>((nthr + bs - 1) / bs, 8192)), blk(bs);
#define PBF_(KS_, BN_) do { if (VW == 4) T4K_LAUNCH((k_poolblock_fwd<KS_, 4, BN_>), grid, blk, 0, S(s), p); \
                            else if (VW == 2) T4K_LAUNCH((k_poolblock_fwd<KS_, 2, BN_>), grid, blk, 0, S(s), p); \
                            else T4K_LAUNCH((k_poolblock_fwd<KS_, 1, BN_>), grid, blk, 0, S(s), p); } while (0)
#define PBF(KS_) do { if (XH) PBF_(KS_, true); else PBF_(KS_, false); } while (0)
    switch (b->KS) { case 1: PBF(1); break; case 2: PBF(2); break; default: PBF(3); break; }
#undef PBF
#undef PBF_
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_poolblock_fwd(const float *X, const t4k_poolblock *b, int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t s) {
    return poolblock_fwd_impl(X, b, N, H1, W1, H0, W0, C, s, nullptr, nullptr, nullptr, nullptr, nullptr);
}
// the apply half of a batch-norm layer + the run behind it in one pass over the conv output (the statistics are final in stat_dev)
int t4k_bn_poolblock_fwd(const float *Y, float *O, float *XH, const float *W, const float *B, const float *stat_dev, const t4k_poolblock *b,
                         int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t s) {
    if (!O || !XH || !W || !B || !stat_dev) return fail(T4K_ERR_ARG, "t4k_bn_poolblock_fwd: null batch-norm tensor");
    return poolblock_fwd_impl(Y, b, N, H1, W1, H0, W0, C, s, XH, O, W, B, stat_dev);
}

int t4k_poolblock_bwd(const float *DY, float *X, const t4k_poolblock *b, int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    int rc = check_block(b, "t4k_poolblock_bwd"); if (rc) return rc;
    if (!DY || !X) return fail(T4K_ERR_ARG, "t4k_poolblock_bwd: null tensor");
    const long total = (long)N * H0 * W0 * C; if (total <= 0) return T4K_OK;
    PBB p;
    p.DY = DY; p.Rb = b->copy_out ? (b->post_layer ? b->post_out : (b->pool_layer ? b->pool_out : b->pre_out)) : nullptr;
    p.Qb = b->pool_layer ? b->pool_out : (b->pre_layer ? b->pre_out : X);   // buffer the post activation read its input from
    p.Pb = b->pre_layer ? b->pre_out : X;                         // pool input buffer (forward values -> dX in place)
    p.Xb = X; p.Fpre = b->pre_mask; p.Fpost = b->post_mask;
    p.pre = b->pre_layer; p.pool = b->pool_layer; p.post = b->post_layer;
    p.N = N; p.H1 = H1; p.W1 = W1; p.H0 = H0; p.W0 = W0; p.C = C;
    int VW = vec_width(C, X, b); if (VW > 1 && (((uintptr_t)DY) & (4 * VW - 1))) VW = 1;
    const long nthr = total / VW;
    const int bs = (nthr < (long)BLK * 2 * st().cu_count) ? 64 : BLK;
    const dim3 grid((unsigned)std::min<long>((nthr + bs - 1) / bs, 8192)), blk(bs);
#define PBB_(KS_) do { if (VW == 4) T4K_LAUNCH((k_poolblock_bwd<KS_, 4>), grid, blk, 0, S(s), p); \
                       else if (VW == 2) T4K_LAUNCH((k_poolblock_bwd<KS_, 2>), grid, blk, 0, S(s), p); \
                       else T4K_LAUNCH((k_poolblock_bwd<KS_, 1>), grid, blk, 0, S(s), p); } while (0)
    switch (b->KS) { case 1: PBB_(1); break; case 2: PBB_(2); break; default: PBB_(3); break; }
#undef PBB_
    T4K_LAUNCH_CHECK(); return T4K_OK;
}

} // extern "C"
