// optim.hip - fused parameter update + gradient zeroing, and Philox RNG fill.
// Reference: k_sgd / k_adam / k_adamw src/nn/nmath.cu:419-472 (Model::sgd/adam/adamw
// src/nn/gradient.cu:132-169); k_rand src/util.cu:56-70.
#include "t4k_common.h"
#include <string.h>

using namespace t4k;

namespace {

// The update arithmetic is spelled out operation by operation (__fmul_rn / __fadd_rn ... : no fused multiply-add, IEEE division and root), in
// the order of the reference's expressions as the oracle evaluates them (k_sgd / k_adam / k_adamw, nmath.cu:419-472; oracle built with
// -ffp-contract=off).  Left to the compiler (-ffp-contract=fast) the same source contracted differently in different kernels - the fold-inside-
// optimizer launch and the chunked launch must agree bit for bit (tests/test_gpu_deferred.py) - and differently from the oracle; spelled out,
// every kernel that updates a parameter produces the oracle's bits.
__device__ __forceinline__ void sgd1(float &g, float &dg, float &m, int Nw, float lr, float b, bool mom) {
    const float d = __fdiv_rn(dg, (float)Nw);              // Nw = parameter tensor's N, not batch (quirk a-19)
    if (!mom) g = __fsub_rn(g, __fmul_rn(lr, d));
    else { m = __fadd_rn(__fmul_rn(b, m), __fmul_rn(__fsub_rn(1.0f, b), d)); g = __fsub_rn(g, __fmul_rn(lr, m)); }
    dg = 0.0f;
}
__device__ __forceinline__ void adam1(float &g, float &dg, float &m, float &v, float lr, float b1, float b2) {
    const float d = dg;
    m = __fadd_rn(__fmul_rn(b1, m), __fmul_rn(__fsub_rn(1.0f, b1), d));
    v = __fadd_rn(__fmul_rn(b2, v), __fmul_rn(__fmul_rn(__fsub_rn(1.0f, b2), d), d));
    g = __fsub_rn(g, __fdiv_rn(__fmul_rn(lr, m), __fadd_rn(__fsqrt_rn(v), DU_EPS)));   // no bias correction, eps outside sqrt
    dg = 0.0f;
}
__device__ __forceinline__ void adamw1(float &g, float &dg, float &m, float &v, float lr, float b1, float b2, float wd) {
    const float d = dg;
    m = __fadd_rn(__fmul_rn(b1, m), __fmul_rn(__fsub_rn(1.0f, b1), d));
    v = __fadd_rn(__fmul_rn(b2, v), __fmul_rn(__fmul_rn(__fsub_rn(1.0f, b2), d), d));
    g = __fsub_rn(g, __fmul_rn(lr, __fsub_rn(__fdiv_rn(m, __fadd_rn(__fsqrt_rn(v), DU_EPS)), __fmul_rn(wd, d))));
    dg = 0.0f;
}

__global__ void __launch_bounds__(BLK) k_sgd(float *G, float *DG, float *M, int Nw, float lr, float b, long n) {
    const bool mom = !(fabsf(b) < DU_EPS);
    for (long j = (long)blockIdx.x * BLK + threadIdx.x; j < n; j += (long)gridDim.x * BLK) {
        float g = G[j], dg = DG[j], m = mom ? M[j] : 0.f;
        sgd1(g, dg, m, Nw, lr, b, mom);
        G[j] = g; DG[j] = 0.f; if (mom) M[j] = m;
    }
}
__global__ void __launch_bounds__(BLK) k_adam(float *G, float *DG, float *M, float *V, float lr, float b1, float b2, long n) {
    for (long j = (long)blockIdx.x * BLK + threadIdx.x; j < n; j += (long)gridDim.x * BLK) {
        float g = G[j], dg = DG[j], m = M[j], v = V[j];
        adam1(g, dg, m, v, lr, b1, b2);
        G[j] = g; DG[j] = 0.f; M[j] = m; V[j] = v;
    }
}
__global__ void __launch_bounds__(BLK) k_adamw(float *G, float *DG, float *M, float *V, float lr, float b1, float b2, float wd, long n) {
    for (long j = (long)blockIdx.x * BLK + threadIdx.x; j < n; j += (long)gridDim.x * BLK) {
        float g = G[j], dg = DG[j], m = M[j], v = V[j];
        adamw1(g, dg, m, v, lr, b1, b2, wd);
        G[j] = g; DG[j] = 0.f; M[j] = m; V[j] = v;
    }
}
// one launch over every parameter tensor of a model: blockIdx.y = tensor, blockIdx.x strides its elements
__global__ void __launch_bounds__(BLK) k_opt_multi(int kind, const t4k_param_rec *__restrict__ tab,
                                                   float lr, float b1, float b2, float wd) {
    const t4k_param_rec r = tab[blockIdx.y];
    const bool mom = !(fabsf(b1) < DU_EPS);
    for (long j = (long)blockIdx.x * BLK + threadIdx.x; j < r.n; j += (long)gridDim.x * BLK) {
        float g = r.G[j], dg = r.DG[j];
        if (kind == 0) {
            float m = mom ? r.M[j] : 0.f;
            sgd1(g, dg, m, r.Nw, lr, b1, mom);
            if (mom) r.M[j] = m;
        } else {
            float m = r.M[j], v = r.V[j];
            if (kind == 1) adam1(g, dg, m, v, lr, b1, b2); else adamw1(g, dg, m, v, lr, b1, b2, wd);
            r.M[j] = m; r.V[j] = v;
        }
        r.G[j] = g; r.DG[j] = 0.f;
    }
}

// the same step with the grid sized to the parameters: workgroup b owns 1024-element chunk b of the concatenation of all tensors; the
// record's `pad` field holds the tensor's first chunk (host-filled prefix), found by a short scan of the table (uniform -> scalar loads)
__global__ void __launch_bounds__(BLK) k_opt_chunked(int kind, const t4k_param_rec *__restrict__ tab, int nt,
                                                     float lr, float b1, float b2, float wd, const float *keep_src, float *keep_dst) {   // keep_*: t4k_opt_snapshot
    int i = 0;
    while (i + 1 < nt && (int)blockIdx.x >= tab[i + 1].pad) i++;
    const t4k_param_rec r = tab[i];
    const bool mom = !(fabsf(b1) < DU_EPS);
    const long j0 = ((long)blockIdx.x - r.pad) * 1024;
    typedef float v4 __attribute__((ext_vector_type(4)));
    const long jv = j0 + 4 * threadIdx.x;
    const bool vec = ((((uintptr_t)r.G | (uintptr_t)r.DG | (uintptr_t)r.M | (uintptr_t)r.V | (r.G == keep_src ? (uintptr_t)keep_dst : 0)) & 15) == 0) && jv + 3 < r.n;   // the snapshot store is a 16-byte one too
    if (vec) {                                              // four consecutive elements per thread: 16-byte loads and stores, a quarter of the instructions
        v4 g = *reinterpret_cast<const v4 *>(r.G + jv), dg = *reinterpret_cast<const v4 *>(r.DG + jv);
        v4 m = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
        if (kind != 0 || mom) m = *reinterpret_cast<const v4 *>(r.M + jv);
        if (kind != 0) v = *reinterpret_cast<const v4 *>(r.V + jv);
        if (r.G == keep_src) *reinterpret_cast<v4 *>(keep_dst + jv) = g;        // the pre-update values of a snapshotted tensor
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float gq = g[q], dq = dg[q], mq = m[q], vq = v[q];
            if (kind == 0) sgd1(gq, dq, mq, r.Nw, lr, b1, mom);
            else if (kind == 1) adam1(gq, dq, mq, vq, lr, b1, b2);
            else adamw1(gq, dq, mq, vq, lr, b1, b2, wd);
            g[q] = gq; m[q] = mq; v[q] = vq;
        }
        *reinterpret_cast<v4 *>(r.G + jv) = g;
        const v4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<v4 *>(r.DG + jv) = z;
        if (kind != 0 || mom) *reinterpret_cast<v4 *>(r.M + jv) = m;
        if (kind != 0) *reinterpret_cast<v4 *>(r.V + jv) = v;
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {                           // ragged end of a tensor / unaligned tensors: element by element
        const long j = jv + q;
        if (j >= r.n) break;
        float g = r.G[j], dg = r.DG[j];
        if (r.G == keep_src) keep_dst[j] = g;
        if (kind == 0) {
            float m = mom ? r.M[j] : 0.f;
            sgd1(g, dg, m, r.Nw, lr, b1, mom);
            if (mom) r.M[j] = m;
        } else {
            float m = r.M[j], v = r.V[j];
            if (kind == 1) adam1(g, dg, m, v, lr, b1, b2); else adamw1(g, dg, m, v, lr, b1, b2, wd);
            r.M[j] = m; r.V[j] = v;
        }
        r.G[j] = g; r.DG[j] = 0.f;
    }
}

// ---- the optimizer step with a conv stack's pending partial fold inside (t4k_opt_step).  Workgroups [0, nfold) fold 16 gradient
// elements each out of the backward's per-workgroup partial rows (cs_fold16: the arithmetic and order of k_cs_fold) and update those
// parameters at once; workgroups behind them own the 1024-element chunks of k_opt_chunked, one element per thread, and leave the
// tensors the fold workgroups handle alone (`skip`: bit i = table record i).  One launch instead of k_cs_fold + k_opt_chunked; the
// result is bit-identical to the two launches (dg = DG + fold either way).
struct FoldRecs { t4k_param_rec r[6]; };                   // the table record behind each fold segment
__device__ __forceinline__ void opt1(int kind, const t4k_param_rec &r, long j, float dg, float lr, float b1, float b2, float wd, bool mom,
                                     const float *keep_src = nullptr, float *keep_dst = nullptr) {
    float g = r.G[j];
    if (keep_src && r.G == keep_src) keep_dst[j] = g;
    if (kind == 0) {
        float m = mom ? r.M[j] : 0.f;
        sgd1(g, dg, m, r.Nw, lr, b1, mom);
        if (mom) r.M[j] = m;
    } else {
        float m = r.M[j], v = r.V[j];
        if (kind == 1) adam1(g, dg, m, v, lr, b1, b2); else adamw1(g, dg, m, v, lr, b1, b2, wd);
        r.M[j] = m; r.V[j] = v;
    }
    r.G[j] = g; r.DG[j] = 0.f;
}
// XCHG: the gradient element is first summed over all ranks through the one-shot peer exchange (xchg.hip) - pushed into every peer's
// window, then added up in rank order out of this rank's window - so fold + all-reduce + update are ONE launch.  `slab` = the model's
// gradient slab: an element's place in the windows is its offset in the slab.
T4K_SPIN_DECL
__global__ void k_opt_set_err(int *p) { g_spin_err_dev = p; }
template <bool XCHG>
__global__ void __launch_bounds__(1024) k_opt_step(int kind, const t4k_param_rec *__restrict__ tab, int nt, float lr, float b1, float b2, float wd,
                                                   const CsFoldArgs fa, const FoldRecs fr, int nfold, unsigned long long skip, const float *slab, const XchgDev xd,
                                                   const float *keep_src, float *keep_dst) {
    __shared__ CsFoldSm sm;
    const bool mom = !(fabsf(b1) < DU_EPS);
    if ((int)blockIdx.x < nfold) {
        // the parameter, its gradient and moments are requested BEFORE the partial rows are summed: behind the fold they were a second, dependent memory
        // round trip on the launch's longest path (round 6)
        const int blk = cs_fold_block(blockIdx.x, nfold), e = blk * 16 + (threadIdx.x & 15);
        int q0 = 0, k0 = 0; bool has = false;
#pragma unroll
        for (int t = 0; t < 6; t++) if (t < fa.nseg && e >= fa.seg[t].start && e < fa.seg[t].start + fa.seg[t].n) { k0 = e - fa.seg[t].start; q0 = t; has = true; }
        float g0 = 0.f, d0 = 0.f, m0 = 0.f, v0 = 0.f;
        if (has && threadIdx.x < 16) {
            const t4k_param_rec &r = fr.r[q0];
            g0 = r.G[k0]; d0 = fa.seg[q0].dst[k0];
            if (kind != 0 || mom) m0 = r.M[k0];
            if (kind != 0) v0 = r.V[k0];
        }
        float v; int q, k;
        if (cs_fold16(fa, blk, sm, v, q, k)) {
            float dg = d0 + v;
            bool ok = true;
            if (XCHG) { const long z = (long)(fa.seg[q].dst - slab) + k; xchg_push(xd, z, dg); dg = xchg_sum(xd, z, dg, g_spin_err_dev, ok); }
            if (ok) {
                const t4k_param_rec &r = fr.r[q];
                if (keep_src && r.G == keep_src) keep_dst[k] = g0;
                if (kind == 0) { sgd1(g0, dg, m0, r.Nw, lr, b1, mom); if (mom) r.M[k] = m0; }
                else { if (kind == 1) adam1(g0, dg, m0, v0, lr, b1, b2); else adamw1(g0, dg, m0, v0, lr, b1, b2, wd); r.M[k] = m0; r.V[k] = v0; }
                r.G[k] = g0; r.DG[k] = 0.f;
            }
        }
        return;
    }
    const int b = (int)blockIdx.x - nfold;
    int i = 0;
    while (i + 1 < nt && b >= tab[i + 1].pad) i++;
    if ((skip >> i) & 1ull) return;
    const t4k_param_rec r = tab[i];
    const long j = ((long)b - r.pad) * 1024 + threadIdx.x;
    if (j < r.n) {
        float dg = r.DG[j];
        bool ok = true;
        if (XCHG) { const long z = (long)(r.DG - slab) + j; xchg_push(xd, z, dg); dg = xchg_sum(xd, z, dg, g_spin_err_dev, ok); }
        if (ok) opt1(kind, r, j, dg, lr, b1, b2, wd, mom, keep_src, keep_dst);
    }
}

// d[i] = scale * (bias + u_i): element i <- Philox(counter = (off+i)/4)[i%4].
// The stream state (counter, seed) is read from device memory and advanced by the last workgroup to
// finish, so the same launch captured in a hipGraph draws a fresh slice of the stream on every replay.
__global__ void __launch_bounds__(BLK) k_rand(float *d, long n, int opt, float bias, float scale, RngArg ra) {
    uint64_t base, seed; rng_begin(ra, base, seed);
    const long nq = (n + 3) >> 2;
    for (long q = (long)blockIdx.x * BLK + threadIdx.x; q < nq; q += (long)gridDim.x * BLK) {
        uint32_t r[4]; float v[4];
        philox4x32_10(base + (uint64_t)q, seed, r);
        if (opt == T4K_NORMAL) {
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const float u1 = u01(r[2 * p]), u2 = u01(r[2 * p + 1]);
                const float rad = sqrtf(-2.0f * logf(u1)), ang = 6.2831853071795865f * u2;
                v[2 * p] = rad * cosf(ang); v[2 * p + 1] = rad * sinf(ang);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = u01(r[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { const long i = q * 4 + k; if (i < n) d[i] = scale * (bias + v[k]); }
    }
    if (ra.state) rng_advance_last_block(ra.state, base, (uint64_t)nq);
}

} // namespace

namespace t4k {
__global__ void k_rng_seed_dev(uint64_t *state, uint64_t ctr, uint64_t seed) { state[0] = ctr; state[1] = 0; state[2] = seed; }
// Reserve nq counters for one launch.  Eager: by-value (base, seed), host counter advances.  While a graph is being
// captured: the launch reads the device copy instead and advances it itself; the host only totals the draw.
RngArg rng_draw(hipStream_t hs, uint64_t nq, bool sample_keyed) {
    State &g = st();
    RngArg a = { g.rng_ctr, g.seed, nullptr };
    if (sample_keyed && g.shard_world > 1) {              // this rank's rows of the whole batch's draw (eager only: DP does not replay graphs)
        a.base += (uint64_t)g.shard_rank * nq; g.rng_ctr += (uint64_t)g.shard_world * nq;
    } else if (g.capturing) {
        if (!g.d_rng) (void)hipMalloc((void **)&g.d_rng, 4 * sizeof(uint64_t));   // allocated by t4k_graph_begin normally
        a.state = g.d_rng; g.cap_adv += nq;
    } else g.rng_ctr += nq;
    (void)hs;
    return a;
}
// bring the device copy up to date on stream `hs` if it is stale (before a graph that draws is launched)
void rng_sync_device(hipStream_t hs) {
    State &g = st();
    if (!g.d_rng || g.d_rng_ctr == g.rng_ctr) return;
    T4K_LAUNCH(k_rng_seed_dev, dim3(1), dim3(1), 0, hs, g.d_rng, g.rng_ctr, g.seed);
    g.d_rng_ctr = g.rng_ctr;
}
}

extern "C" {

int t4k_sgd(float *G, float *DG, float *M, int Nw, float lr, float beta, long n, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0) return T4K_OK;
    if (!G || !DG || Nw == 0) return fail(T4K_ERR_ARG, "t4k_sgd: bad argument");
    if (!(fabsf(beta) < DU_EPS) && !M) return fail(T4K_ERR_ARG, "t4k_sgd: momentum tensor missing");
    T4K_LAUNCH(k_sgd, dim3(grid_for(n)), dim3(BLK), 0, S(s), G, DG, M, Nw, lr, beta, n);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_adam(float *G, float *DG, float *M, float *V, float lr, float b1, float b2, long n, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0) return T4K_OK;
    if (!G || !DG || !M || !V) return fail(T4K_ERR_ARG, "t4k_adam: null");
    T4K_LAUNCH(k_adam, dim3(grid_for(n)), dim3(BLK), 0, S(s), G, DG, M, V, lr, b1, b2, n);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_adamw(float *G, float *DG, float *M, float *V, float lr, float b1, float b2, float wd, long n, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0) return T4K_OK;
    if (!G || !DG || !M || !V) return fail(T4K_ERR_ARG, "t4k_adamw: null");
    T4K_LAUNCH(k_adamw, dim3(grid_for(n)), dim3(BLK), 0, S(s), G, DG, M, V, lr, b1, b2, wd, n);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_opt_multi(int kind, const t4k_param_rec *tab_dev, int n_tensors, long max_n,
                  float lr, float b1, float b2, float wd, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n_tensors <= 0) return T4K_OK;
    if (!tab_dev || kind < 0 || kind > 2) return fail(T4K_ERR_ARG, "t4k_opt_multi: bad argument");
    int gx = grid_for(max_n); if (gx > 256) gx = 256;
    T4K_LAUNCH(k_opt_multi, dim3(gx, n_tensors), dim3(BLK), 0, S(s), kind, tab_dev, lr, b1, b2, wd);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}

// one pending snapshot request (t4k_opt_snapshot): consumed by the next chunked / step launch
static const float *g_keep_src = nullptr; static float *g_keep_dst = nullptr;
int t4k_opt_snapshot(const float *G, float *G_PREV) {
    if (G && !G_PREV) return fail(T4K_ERR_ARG, "t4k_opt_snapshot: null destination");
    g_keep_src = G; g_keep_dst = G ? G_PREV : nullptr;
    return T4K_OK;
}
int t4k_opt_snapshot_pending(void) { return g_keep_src ? 1 : 0; }
int t4k_opt_chunked(int kind, const t4k_param_rec *tab_dev, int n_tensors, int n_chunks,
                    float lr, float b1, float b2, float wd, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n_tensors <= 0 || n_chunks <= 0) return T4K_OK;
    if (!tab_dev || kind < 0 || kind > 2) return fail(T4K_ERR_ARG, "t4k_opt_chunked: bad argument");
    const float *ks = g_keep_src; float *kd = g_keep_dst; g_keep_src = nullptr; g_keep_dst = nullptr;
    T4K_LAUNCH(k_opt_chunked, dim3((unsigned)n_chunks), dim3(BLK), 0, S(s), kind, tab_dev, n_tensors, lr, b1, b2, wd, ks, kd);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}

// the optimizer step of a model: t4k_opt_chunked, plus whatever the backward deferred to it (t4k_conv_stack_bwd with train | 4: the
// dF | dB partial fold) inside the same launch.  tab_host = the host's copy of the table tab_dev holds.
// _dp: with the one-shot peer exchange connected (t4k_xchg_connect) the gradients are SUMMED OVER ALL RANKS on the way - `slab` / `slab_n` =
// the model's gradient slab, which must hold every DG of the table.  Without an exchange it is t4k_opt_step.
static int opt_step_impl(int kind, const t4k_param_rec *tab_dev, const t4k_param_rec *tab_host, int n_tensors, int n_chunks,
                         float lr, float b1, float b2, float wd, const float *slab, long slab_n, t4k_stream_t s) {
    State &g = st();
    std::lock_guard<std::recursive_mutex> pending_lock(pending_mu());      // the deferred fold is consumed (or flushed) exactly once, whoever else reaches an entry point meanwhile
    const bool dp = slab && xchg().connected && (xchg().world > 1 || xchg().self);
    bool slab_ok = dp && tab_dev && tab_host && n_tensors > 0 && n_tensors <= 64 && n_chunks > 0 && slab_n <= xchg().n && !g.capturing && kind >= 0 && kind <= 2;
    for (int i = 0; slab_ok && i < n_tensors; i++) if (tab_host[i].DG < slab || tab_host[i].DG + tab_host[i].n > slab + slab_n) slab_ok = false;
    if (dp && !slab_ok) {                                     // cannot ride in the update: sum the slab first (same transport, its own launches), then the plain step
        if (g.capturing) return fail(T4K_ERR_UNSUPPORTED, "t4k_opt_step_dp: the one-shot exchange cannot be recorded into a graph (every call carries its own epoch)");
        if (g.pending) flush_pending();
        const int rc = xchg_allreduce(const_cast<float *>(slab), slab_n, S(s)); if (rc) return rc;
        return t4k_opt_chunked(kind, tab_dev, n_tensors, n_chunks, lr, b1, b2, wd, s);
    }
    if (!(g.pending & 1) && !dp) return t4k_opt_chunked(kind, tab_dev, n_tensors, n_chunks, lr, b1, b2, wd, s);
    const PendingFold &pf = pending_fold();
    FoldRecs fr; memset((void *)&fr, 0, sizeof(fr));
    CsFoldArgs fa; memset((void *)&fa, 0, sizeof(fa));
    unsigned long long skip = 0;
    int nfold = 0;
    if (g.pending & 1) {
        static const int on = T4K_LAB_ENV("T4K_OPT_FOLD", 1);
        bool ok = on && tab_dev && tab_host && kind >= 0 && kind <= 2 && n_tensors > 0 && n_tensors <= 64 && n_chunks > 0 && pf.hs == S(s) && !g.capturing;
        for (int q = 0; ok && q < pf.fa.nseg; q++) {              // every pending segment must be a whole gradient tensor of this table
            int hit = -1;
            for (int i = 0; i < n_tensors; i++) if (tab_host[i].DG == pf.fa.seg[q].dst && tab_host[i].n == (long)pf.fa.seg[q].n) { hit = i; break; }
            if (hit < 0) ok = false; else { fr.r[q] = tab_host[hit]; skip |= 1ull << hit; }
        }
        if (!ok) { flush_pending(); if (!dp) return t4k_opt_chunked(kind, tab_dev, n_tensors, n_chunks, lr, b1, b2, wd, s); skip = 0; }
        else { g.pending &= ~1; fa = pf.fa; nfold = pf.fa.total / 16; }
    }
    static bool err_set = false;
    if (!err_set && g.spin_err) { hipLaunchKernelGGL(k_opt_set_err, dim3(1), dim3(1), 0, S(s), g.spin_err); err_set = true; }   // once per process: set-up, not a launch of the step (not counted)
    const float *ks = g_keep_src; float *kd = g_keep_dst; g_keep_src = nullptr; g_keep_dst = nullptr;
    if (dp) {
        const XchgDev xd = xchg_begin(false);
        T4K_LAUNCH(k_opt_step<true>, dim3((unsigned)(nfold + n_chunks)), dim3(1024), 0, S(s), kind, tab_dev, n_tensors, lr, b1, b2, wd, fa, fr, nfold, skip, slab, xd, ks, kd);
    } else {
        XchgDev xd; memset((void *)&xd, 0, sizeof(xd));
        T4K_LAUNCH(k_opt_step<false>, dim3((unsigned)(nfold + n_chunks)), dim3(1024), 0, S(s), kind, tab_dev, n_tensors, lr, b1, b2, wd, fa, fr, nfold, skip, slab, xd, ks, kd);
    }
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_opt_step(int kind, const t4k_param_rec *tab_dev, const t4k_param_rec *tab_host, int n_tensors, int n_chunks,
                 float lr, float b1, float b2, float wd, t4k_stream_t s) {
    T4K_REQUIRE_INIT_NOFLUSH();
    return opt_step_impl(kind, tab_dev, tab_host, n_tensors, n_chunks, lr, b1, b2, wd, nullptr, 0, s);
}
int t4k_opt_step_dp(int kind, const t4k_param_rec *tab_dev, const t4k_param_rec *tab_host, int n_tensors, int n_chunks,
                    float lr, float b1, float b2, float wd, float *slab, long slab_n, t4k_stream_t s) {
    T4K_REQUIRE_INIT_NOFLUSH();
    return opt_step_impl(kind, tab_dev, tab_host, n_tensors, n_chunks, lr, b1, b2, wd, slab, slab_n, s);
}

int t4k_rand_init(uint64_t seed) { T4K_REQUIRE_INIT(); State &g = st(); g.seed = seed; g.rng_ctr = 0; g.d_rng_ctr = ~0ull; return T4K_OK; }
uint64_t t4k_rand_offset(void) { return st().rng_ctr * 4; }
uint64_t t4k_rand_seed(void) { return st().seed; }
int t4k_rand_shard_world(void) { return st().shard_world; }
int t4k_rand_set_offset(uint64_t off) { T4K_REQUIRE_INIT(); State &g = st(); g.rng_ctr = off / 4; g.d_rng_ctr = ~0ull; return T4K_OK; }
int t4k_rand_set_shard(int rank, int world) {
    T4K_REQUIRE_INIT();
    if (world < 1 || rank < 0 || rank >= world) return fail(T4K_ERR_ARG, "t4k_rand_set_shard: rank %d of %d", rank, world);
    State &g = st(); g.shard_rank = rank; g.shard_world = world;
    return T4K_OK;
}
int t4k_dropout_mask(float *mask, long n, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0) return T4K_OK;
    if (!mask) return fail(T4K_ERR_ARG, "t4k_dropout_mask: null");
    const RngArg ra = rng_draw(S(s), (uint64_t)((n + 3) / 4), true);
    T4K_LAUNCH(k_rand, dim3(grid_for((n + 3) / 4)), dim3(BLK), 0, S(s), mask, n, (int)T4K_UNIFORM, 0.0f, 1.0f, ra);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}
int t4k_rand(float *d, long n, int opt, float bias, float scale, t4k_stream_t s) {
    T4K_REQUIRE_INIT(); if (n <= 0) return T4K_OK;
    if (!d) return fail(T4K_ERR_ARG, "t4k_rand: null");
    const RngArg ra = rng_draw(S(s), (uint64_t)((n + 3) / 4));
    T4K_LAUNCH(k_rand, dim3(grid_for((n + 3) / 4)), dim3(BLK), 0, S(s), d, n, opt, bias, scale, ra);
    T4K_LAUNCH_CHECK(); return T4K_OK;
}

} // extern "C"
