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

using namespace t4k;

namespace {

struct PB {                                   // device copy of t4k_poolblock + geometry
    const float *X; float *P, *Q, *R, *R2, *Fpre, *Fpost;
    int pre, pool, post; float a_pre, a_post;
    int N, H1, W1, H0, W0, C;
    uint64_t *rng;
};

template <int KS>
__global__ void __launch_bounds__(BLK) k_poolblock_fwd(PB p) {
    const long total = (long)p.N * p.H0 * p.W0 * p.C;
    uint64_t base = 0, seed = 0;
    const bool draw = p.pre == T4K_L_DROPOUT;
    if (draw) rng_state_read(p.rng, base, seed);
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < total; z += (long)gridDim.x * BLK) {
        const int c = (int)(z % p.C); long t = z / p.C;
        const int j0 = (int)(t % p.W0); t /= p.W0;
        const int i0 = (int)(t % p.H0); const int n = (int)(t / p.H0);
        float v = 0.f; bool first = true;
#pragma unroll
        for (int y = 0; y < KS; y++)
#pragma unroll
            for (int x = 0; x < KS; x++) {
                const int gi = i0 * KS + y, gj = j0 * KS + x;
                if (gi >= p.H1 || gj >= p.W1) continue;
                const long a = (((long)n * p.H1 + gi) * p.W1 + gj) * p.C + c;
                float e = p.X[a];
                if (p.pre) {
                    float o, f;
                    act_rt(p.pre, e, draw ? philox_u01_at(base, seed, a) : 0.f, p.a_pre, o, f);
                    p.Fpre[a] = f; p.P[a] = o; e = o;
                }
                if (p.pool == T4K_L_MAXPOOL)      v = first ? e : fmaxf(e, v);
                else if (p.pool == T4K_L_MINPOOL) v = first ? e : fminf(e, v);
                else                              v += e;
                first = false;
            }
        if (p.pool == T4K_L_AVGPOOL) v /= (float)(KS * KS);
        if (p.pool) p.Q[z] = v;
        if (p.post) { float o, f; act_rt(p.post, v, 0.f, p.a_post, o, f); p.Fpost[z] = f; p.R[z] = o; v = o; }
        if (p.R2) p.R2[z] = v;
    }
    if (draw) rng_advance_last_block(p.rng, base, (uint64_t)(((long)p.N * p.H1 * p.W1 * p.C + 3) >> 2));
}

// backward: DY = gradient w.r.t. the run's last tensor.  Writes (reference in-place convention: each layer's
// input buffer receives its dX): R2-side copy -> R buffer, R*Fpost -> Q buffer, pool scatter -> P buffer (which
// still holds the forward values needed to find the extreme), P*Fpre -> X buffer.
struct PBB {
    const float *DY; float *Rb, *Qb, *Pb, *Xb; const float *Fpre, *Fpost;
    int pre, pool, post;
    int N, H1, W1, H0, W0, C;
};
template <int KS>
__global__ void __launch_bounds__(BLK) k_poolblock_bwd(PBB p) {
    const long total = (long)p.N * p.H0 * p.W0 * p.C;
    for (long z = (long)blockIdx.x * BLK + threadIdx.x; z < total; z += (long)gridDim.x * BLK) {
        const int c = (int)(z % p.C); long t = z / p.C;
        const int j0 = (int)(t % p.W0); t /= p.W0;
        const int i0 = (int)(t % p.H0); const int n = (int)(t / p.H0);
        float g = p.DY[z];
        if (p.Rb) p.Rb[z] = g;                                    // flatten: in = out
        if (p.post) { g = g * p.Fpost[z]; p.Qb[z] = g; }           // activation: in = out (*) mask
        if (!p.pool) {                                            // no pooling in this run (KS == 1)
            if (p.pre) p.Xb[z] = g * p.Fpre[z];
            continue;
        }
        float best = 0.f; long arg = -1;
        float dv[KS * KS]; long av[KS * KS];
#pragma unroll
        for (int y = 0; y < KS; y++)
#pragma unroll
            for (int x = 0; x < KS; x++) {
                const int q = y * KS + x;
                const int gi = i0 * KS + y, gj = j0 * KS + x;
                av[q] = -1; dv[q] = 0.f;
                if (gi >= p.H1 || gj >= p.W1) continue;
                const long a = (((long)n * p.H1 + gi) * p.W1 + gj) * p.C + c;
                av[q] = a;
                if (p.pool == T4K_L_AVGPOOL) dv[q] = g / (float)(KS * KS);
                else {
                    const float e = p.Pb[a];
                    const bool better = (p.pool == T4K_L_MAXPOOL) ? (e > best) : (e < best);
                    if (arg < 0 || better) { best = e; arg = a; }  // first extreme wins
                }
            }
#pragma unroll
        for (int q = 0; q < KS * KS; q++) {
            const long a = av[q]; if (a < 0) continue;
            const float d = (p.pool == T4K_L_AVGPOOL) ? dv[q] : (a == arg ? g : 0.f);
            p.Pb[a] = d;
            if (p.pre) p.Xb[a] = d * p.Fpre[a];
        }
    }
}

bool is_act(int l)  { return l == T4K_L_RELU || l == T4K_L_TANH || l == T4K_L_SIGMOID || l == T4K_L_SELU || l == T4K_L_LEAKYRL || l == T4K_L_ELU || l == T4K_L_DROPOUT; }
bool is_pool(int l) { return l == T4K_L_AVGPOOL || l == T4K_L_MAXPOOL || l == T4K_L_MINPOOL; }

int check_block(const t4k_poolblock *b, const char *who) {
    if (!b) return fail(T4K_ERR_ARG, "%s: null block", who);
    if (b->pre_layer && !is_act(b->pre_layer))   return fail(T4K_ERR_UNSUPPORTED, "%s: pre layer %d", who, b->pre_layer);
    if (b->post_layer && (!is_act(b->post_layer) || b->post_layer == T4K_L_DROPOUT)) return fail(T4K_ERR_UNSUPPORTED, "%s: post layer %d", who, b->post_layer);
    if (b->pool_layer && !is_pool(b->pool_layer)) return fail(T4K_ERR_UNSUPPORTED, "%s: pool layer %d", who, b->pool_layer);
    if (b->pool_layer ? (b->KS != 2 && b->KS != 3) : (b->KS != 1)) return fail(T4K_ERR_UNSUPPORTED, "%s: kernel_size=%d not supported", who, b->KS);
    if (b->pre_layer && (!b->pre_mask || !b->pre_out))    return fail(T4K_ERR_ARG, "%s: pre tensors missing", who);
    if (b->post_layer && (!b->post_mask || !b->post_out)) return fail(T4K_ERR_ARG, "%s: post tensors missing", who);
    if (b->pool_layer && !b->pool_out)                    return fail(T4K_ERR_ARG, "%s: pool output missing", who);
    return T4K_OK;
}

} // namespace

extern "C" {

int t4k_poolblock_fwd(const float *X, const t4k_poolblock *b, int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    int rc = check_block(b, "t4k_poolblock_fwd"); if (rc) return rc;
    if (!X) return fail(T4K_ERR_ARG, "t4k_poolblock_fwd: null input");
    const long total = (long)N * H0 * W0 * C; if (total <= 0) return T4K_OK;
    State &g = st();
    if (b->pre_layer == T4K_L_DROPOUT && !g.d_rng) { rc = t4k_rand_init(0); if (rc) return rc; }
    PB p;
    p.X = X; p.P = b->pre_out; p.Q = b->pool_out; p.R = b->post_out; p.R2 = b->copy_out; p.Fpre = b->pre_mask; p.Fpost = b->post_mask;
    p.pre = b->pre_layer; p.pool = b->pool_layer; p.post = b->post_layer; p.a_pre = b->pre_alpha; p.a_post = b->post_alpha;
    p.N = N; p.H1 = H1; p.W1 = W1; p.H0 = H0; p.W0 = W0; p.C = C; p.rng = g.d_rng;
    const dim3 grid(grid_for(total)), blk(BLK);
    switch (b->KS) {
    case 1: hipLaunchKernelGGL(k_poolblock_fwd<1>, grid, blk, 0, S(s), p); break;
    case 2: hipLaunchKernelGGL(k_poolblock_fwd<2>, grid, blk, 0, S(s), p); break;
    default: hipLaunchKernelGGL(k_poolblock_fwd<3>, grid, blk, 0, S(s), p); break;
    }
    T4K_LAUNCH_CHECK(); return T4K_OK;
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
    const dim3 grid(grid_for(total)), blk(BLK);
    switch (b->KS) {
    case 1: hipLaunchKernelGGL(k_poolblock_bwd<1>, grid, blk, 0, S(s), p); break;
    case 2: hipLaunchKernelGGL(k_poolblock_bwd<2>, grid, blk, 0, S(s), p); break;
    default: hipLaunchKernelGGL(k_poolblock_bwd<3>, grid, blk, 0, S(s), p); break;
    }
    T4K_LAUNCH_CHECK(); return T4K_OK;
}

} // extern "C"
