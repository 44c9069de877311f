/*
 * t4k_on_oracle.cpp - TEST INFRASTRUCTURE ONLY.
 *
 * Implements the C-ABI of include/t4k.h on plain host memory by forwarding every compute entry
 * point to the CPU oracle (t4o_*).  Linking the host VM sources against THIS file instead of
 * libt4hip.so yields `ten4_oracle`, a CPU replica of the product that runs the reference's own
 * .4th known-answer scripts.  It is used (a) to pin the oracle + host orchestration against the
 * expected values in examples/t4_30a/b/c, t4_20a, t4_22a, and (b) as the checker that the GPU
 * `ten4` output is compared with.  The product binary never links or loads this file.
 */
#include <cmath>
#include "../include/t4k.h"
#include "t4_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char g_err[128] = "";

extern "C" {

int t4k_device_count(void) { return 1; }
int t4k_init(int) { return T4K_OK; }
void t4k_shutdown(void) {}
const char *t4k_last_error(void) { return g_err; }
const char *t4k_backend_name(void) { return "cpu-oracle (test infrastructure)"; }
int t4k_device_info(int *cu, int *khz, size_t *hbm) { if (cu) *cu = 1; if (khz) *khz = 0; if (hbm) *hbm = 0; return T4K_OK; }

int t4k_malloc(void **p, size_t bytes) { *p = aligned_alloc(256, (bytes + 255) & ~(size_t)255); return *p ? T4K_OK : T4K_ERR_NOMEM; }
int t4k_free(void *p) { free(p); return T4K_OK; }
int t4k_host_alloc(void **p, size_t bytes) { *p = malloc(bytes ? bytes : 4); return T4K_OK; }
int t4k_host_free(void *p) { free(p); return T4K_OK; }
int t4k_memcpy_h2d(void *d, const void *s, size_t n, t4k_stream_t) { memmove(d, s, n); return T4K_OK; }
int t4k_memcpy_d2h(void *d, const void *s, size_t n, t4k_stream_t) { memmove(d, s, n); return T4K_OK; }
int t4k_memcpy_d2d(void *d, const void *s, size_t n, t4k_stream_t) { memmove(d, s, n); return T4K_OK; }
int t4k_memset(void *d, int b, size_t n, t4k_stream_t) { memset(d, b, n); return T4K_OK; }
int t4k_sync(t4k_stream_t) { return T4K_OK; }
int t4k_stream_create(t4k_stream_t *s) { *s = nullptr; return T4K_OK; }
int t4k_stream_create_plain(t4k_stream_t *s) { *s = nullptr; return T4K_OK; }
int t4k_stream_destroy(t4k_stream_t) { return T4K_OK; }
int t4k_stream_wait_event(t4k_stream_t, t4k_event_t) { return T4K_OK; }
int t4k_set_default_stream(t4k_stream_t) { return T4K_OK; }
t4k_stream_t t4k_default_stream(void) { return nullptr; }
int t4k_event_create(t4k_event_t *e) { *e = nullptr; return T4K_OK; }
int t4k_event_record(t4k_event_t, t4k_stream_t) { return T4K_OK; }
int t4k_event_sync(t4k_event_t) { return T4K_OK; }
int t4k_event_wait(t4k_event_t) { return T4K_OK; }
int t4k_gates_enable(int) { return T4K_OK; }
int t4k_gates_enabled(void) { return 0; }
int t4k_event_elapsed_ms(t4k_event_t, t4k_event_t, float *ms) { *ms = 0; return T4K_OK; }
int t4k_event_destroy(t4k_event_t) { return T4K_OK; }
int t4k_comm_unique_id(void *) { return T4K_ERR_UNSUPPORTED; }
int t4k_comm_init(const void *, int, int) { return T4K_ERR_UNSUPPORTED; }
int t4k_comm_world(void) { return 0; }
int t4k_comm_rank(void) { return 0; }
int t4k_allreduce_sum(float *, long, t4k_stream_t) { return T4K_OK; }
int t4k_comm_destroy(void) { return T4K_OK; }
int t4k_comm_sync_batchnorm(int) { return T4K_OK; }
int t4k_graph_begin(t4k_stream_t) { return T4K_ERR_UNSUPPORTED; }
int t4k_graph_end(t4k_stream_t, t4k_graph_t *) { return T4K_ERR_UNSUPPORTED; }
int t4k_graph_launch(t4k_graph_t, t4k_stream_t) { return T4K_ERR_UNSUPPORTED; }
int t4k_graph_destroy(t4k_graph_t) { return T4K_OK; }

static int rc(int r, const char *w) { if (r) snprintf(g_err, sizeof(g_err), "%s (oracle rc=%d)", w, r); return r; }

int t4k_reduce(int op, const float *s, long n, float avg, float *out, t4k_stream_t) { return rc(t4o_reduce(op, s, n, avg, out), "reduce"); }
int t4k_nan_inf(const float *s, long n, int *c, t4k_stream_t) { return t4o_nan_inf(s, n, c); }
int t4k_copy(const float *s, float *d, long n, t4k_stream_t) { return t4o_copy(s, d, n); }
int t4k_transpose(const float *s, float *d, int H, int W, int C, t4k_stream_t) { return t4o_transpose(s, d, H, W, C); }
int t4k_identity(float *d, int H, int W, int C, t4k_stream_t) { return t4o_identity(d, H, W, C); }
int t4k_math(int op, float *A, float v, long n, t4k_stream_t) { return rc(t4o_math(op, A, v, n), "k_math op not supported"); }
int t4k_ts_op(int op, const float *A, float v, float *O, long n, t4k_stream_t) { return rc(t4o_ts_op(op, A, v, O, n), "k_ts_op"); }
int t4k_tt_op(int op, const float *A, const float *B, float *O, long n, t4k_stream_t) { return rc(t4o_tt_op(op, A, B, O, n), "k_tt_op"); }
int t4k_tt_op2(int op, const float *A, const float *B, float *O, float *O2, long n, t4k_stream_t) {
    int r = rc(t4o_tt_op(op, A, B, O, n), "k_tt_op"); if (!r && O2) memcpy(O2, O, sizeof(float) * (size_t)n); return r;
}
int t4k_bce(const float *T, const float *O, long n, float *out, t4k_stream_t) { return t4o_bce(T, O, n, out); }
int t4k_dot(const float *A, const float *B, float *O, float a, float b, int K, int C, t4k_stream_t) { return t4o_dot(A, B, O, a, b, K, C); }
int t4k_gemm(const float *A, const float *B, float *O, float a, float b, int tA, int tB, int M, int N, int K, int C, t4k_stream_t) {
    if (b == 0.0f) memset(O, 0, sizeof(float) * (size_t)M * N * C);      // product contract: beta == 0 never reads O
    return rc(t4o_gemm(A, B, O, a, b, tA, tB, M, N, K, C), "gemm");
}
int t4k_gemm_f64acc(const float *A, const float *B, float *O, float a, float b, int M, int N, int K, int C, t4k_stream_t) {
    if (b == 0.0f) memset(O, 0, sizeof(float) * (size_t)M * N * C);
    return t4o_gemm_f64acc(A, B, O, a, b, M, N, K, C);
}
int t4k_inverse(float *A, float *I, int K, int *st, t4k_stream_t) { return t4o_inverse(A, I, K, st); }
int t4k_plu(float *A, float *I, int *piv, int K, int *st, t4k_stream_t) { return t4o_plu(A, I, piv, K, st); }
int t4k_lu_inverse(float *A, float *I, int *piv, int K, int *st, t4k_stream_t) { return t4o_lu_inverse(A, I, piv, K, st); }
int t4k_lu_extract(float *LU, int u, int K, t4k_stream_t) { return t4o_lu_extract(LU, u, K); }
int t4k_logdet(const float *LU, int K, float *ld, int *sg, t4k_stream_t) { return t4o_logdet(LU, K, ld, sg); }
int t4k_rand_init(uint64_t seed) { return t4o_rand_init(seed); }
int t4k_rand(float *d, long n, int opt, float bias, float scale, t4k_stream_t) { return t4o_rand(d, n, opt, bias, scale); }
uint64_t t4k_rand_offset(void) { return t4o_rand_offset(); }
int t4k_rand_set_offset(uint64_t o) { return t4o_rand_set_offset(o); }
int t4k_rand_set_shard(int r, int w) { return t4o_rand_set_shard(r, w); }
uint64_t t4k_rand_seed(void) { return t4o_rand_seed(); }
int t4k_rand_shard_world(void) { return t4o_rand_shard_world(); }
int t4k_dropout_mask(float *m, long n, t4k_stream_t) { return t4o_dropout_mask(m, n); }
int t4k_bias(const float *B, float *O, int N, int E0, t4k_stream_t) { return t4o_bias(B, O, N, E0); }
int t4k_activate(int l, const float *I, float *O, float *F, float a, long n, t4k_stream_t) { return rc(t4o_activate(l, I, O, F, a, n), "k_activate"); }
int t4k_softmax(const float *I, float *O, int N, int C, t4k_stream_t) { return t4o_softmax(I, O, N, C); }
int t4k_logsoftmax(const float *I, float *O, int N, int C, t4k_stream_t) {      // forward.cu:245-259 as written (exp(x) - log10 of the row sum)
    for (int n = 0; n < N; n++) {
        const float *x = I + (size_t)n * C; float *o = O + (size_t)n * C; float sum = 0.f;
        for (int c = 0; c < C; c++) { o[c] = expf(x[c]); sum += o[c]; }
        const float ls = log10f(fmaxf(sum, 1.0e-6f));
        for (int c = 0; c < C; c++) o[c] -= ls;
    }
    return T4K_OK;
}
int t4k_batchnorm_fwd(const float *I, float *O, float *XH, const float *W, const float *B, float *st, int N, int HW, int C, t4k_stream_t) {
    return t4o_batchnorm_fwd(I, O, XH, W, B, st, N, HW, C);
}
int t4k_batchnorm_bwd(const float *W, const float *DY, const float *XH, float *DX, float *DW, float *DB, float *st, int N, int HW, int C, int tr, t4k_stream_t) {
    return t4o_batchnorm_bwd(W, DY, XH, DX, DW, DB, st, N, HW, C, tr);
}
int t4k_dlinear_db(const float *DY, float *DB, int N, int E0, t4k_stream_t) { return t4o_dlinear_db(DY, DB, N, E0); }
int t4k_conv2d_fwd(const float *I, float *O, const float *F, const float *B, int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P, t4k_stream_t) {
    int r = t4o_conv2d_fwd(I, O, F, B, N, H1, W1, C1, H0, W0, C0, K, S, P);
    if (r) snprintf(g_err, sizeof(g_err), "nn#fconv kernel_size=%d stride=%d padding=%d not supported", K, S, P);
    return r;
}
int t4k_conv2d_fwd2(const float *I, float *IC, float *O, const float *F, const float *B, int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P, t4k_stream_t st) {
    if (IC) memcpy(IC, I, sizeof(float) * (size_t)N * H1 * W1 * C1);
    return t4k_conv2d_fwd(I, O, F, B, N, H1, W1, C1, H0, W0, C0, K, S, P, st);
}
int t4k_bn_poolblock_fwd(const float *Y, float *O, float *XH, const float *W, const float *B, const float *st, const t4k_poolblock *blk, int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t s) {
    const long NHW = (long)N * H1 * W1;                       // the apply half of t4o_batchnorm_fwd (oracle/t4_oracle.cpp), then the run as its own layers
    for (long k = 0; k < NHW; k++) for (int c = 0; c < C; c++) { const long z = k * C + c; XH[z] = (Y[z] - st[C + c]) * st[c]; O[z] = XH[z] * W[c] + B[c]; }
    return t4k_poolblock_fwd(O, blk, N, H1, W1, H0, W0, C, s);
}
int t4k_conv2d_bn_block_fwd(const float *I, float *IC, float *Y, const float *F, const float *Bc, int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P,
                            float *O, float *XH, const float *W, const float *B, float *st, const t4k_poolblock *blk, int Hq, int Wq, t4k_stream_t s) {
    int r = t4k_conv2d_fwd2(I, IC, Y, F, Bc, N, H1, W1, C1, H0, W0, C0, K, S, P, s); if (r) return r;
    r = t4k_batchnorm_fwd(Y, O, XH, W, B, st, N, H0 * W0, C0, s); if (r) return r;
    return t4k_poolblock_fwd(O, blk, N, H0, W0, Hq, Wq, C0, s);
}
int t4k_conv2d_bn_fwd(const float *I, float *IC, float *Y, const float *F, const float *Bc, int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P,
                      float *O, float *XH, const float *W, const float *B, float *st, t4k_stream_t s) {      // the two layers, one after the other
    int r = t4k_conv2d_fwd2(I, IC, Y, F, Bc, N, H1, W1, C1, H0, W0, C0, K, S, P, s); if (r) return r;
    return t4k_batchnorm_fwd(Y, O, XH, W, B, st, N, H0 * W0, C0, s);
}
int t4k_conv2d_bwd(const float *I, const float *DO, float *DX, const float *F, float *DF, float *DB, int N, int H1, int W1, int C1, int H0, int W0, int C0,
                   int K, int S, int P, int tr, t4k_stream_t) {
    // product contract: DX == NULL -> dF|dB only, DF == NULL -> dX only (the oracle always computes both; use scratch for the skipped half)
    float *dx = DX, *df = DF, *db = DB;
    if (!dx) dx = (float *)calloc((size_t)N * H1 * W1 * C1, sizeof(float));
    if (!df) { df = (float *)calloc((size_t)C1 * K * K * C0, sizeof(float)); db = (float *)calloc((size_t)C0, sizeof(float)); }
    int r = t4o_conv2d_bwd(I, DO, dx, F, df, db, N, H1, W1, C1, H0, W0, C0, K, S, P, tr);
    if (!DX) free(dx);
    if (!DF) { free(df); free(db); }
    if (r) snprintf(g_err, sizeof(g_err), "nn#bconv kernel_size=%d stride=%d padding=%d not supported", K, S, P);
    return r;
}
int t4k_dconv2d_fwd(const float *I, float *O, const float *F, const float *B, int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P, t4k_stream_t) {
    return rc(t4o_dconv2d_fwd(I, O, F, B, N, H1, W1, C1, H0, W0, C0, K, S, P), "nn#fdconv");
}
int t4k_dconv2d_bwd(const float *I, const float *DO, float *DX, const float *F, float *DF, float *DB, int N, int H1, int W1, int C1, int H0, int W0, int C0,
                    int K, int S, int P, int tr, t4k_stream_t) {
    return rc(t4o_dconv2d_bwd(I, DO, DX, F, DF, DB, N, H1, W1, C1, H0, W0, C0, K, S, P, tr), "nn#bdconv");
}
int t4k_conv2d_bwd2(const float *I, const float *DO, float *DX, float *DX2, const float *F, float *DF, float *DB, int N, int H1, int W1, int C1,
                    int H0, int W0, int C0, int K, int S, int P, int tr, t4k_stream_t st) {
    int r = t4k_conv2d_bwd(I, DO, DX, F, DF, DB, N, H1, W1, C1, H0, W0, C0, K, S, P, tr, st);
    if (!r && DX && DX2) memcpy(DX2, DX, sizeof(float) * (size_t)N * H1 * W1 * C1);
    return r;
}
int t4k_pool(int l, const float *I, float *O, int N, int H1, int W1, int H0, int W0, int C, int KS, t4k_stream_t) { return rc(t4o_pool(l, I, O, N, H1, W1, H0, W0, C, KS), "k_pool"); }
int t4k_dpool(int l, float *I, const float *DY, int N, int H1, int W1, int H0, int W0, int C, int KS, t4k_stream_t) { return rc(t4o_dpool(l, I, DY, N, H1, W1, H0, W0, C, KS), "k_dpool"); }
int t4k_sgd(float *G, float *DG, float *M, int Nw, float lr, float b, long n, t4k_stream_t) { return t4o_sgd(G, DG, M, Nw, lr, b, n); }
int t4k_adam(float *G, float *DG, float *M, float *V, float lr, float b1, float b2, long n, t4k_stream_t) { return t4o_adam(G, DG, M, V, lr, b1, b2, n); }
int t4k_adamw(float *G, float *DG, float *M, float *V, float lr, float b1, float b2, float wd, long n, t4k_stream_t) { return t4o_adamw(G, DG, M, V, lr, b1, b2, wd, n); }
int t4k_copy_mask(const float *T, const float *M, float *OUT, float *IN, long n, t4k_stream_t) { for (long i = 0; i < n; i++) { const float t = T[i]; OUT[i] = t; IN[i] = t * M[i]; } return T4K_OK; }
int t4k_broadcast_rows(const float *T, float *O, int N, int E, t4k_stream_t) { for (int n = 0; n < N; n++) for (int e = 0; e < E; e++) O[(size_t)n * E + e] = T[n]; return T4K_OK; }
int t4k_onehot(const uint32_t *l, float *hot, int N, int E, t4k_stream_t) { return t4o_onehot(l, hot, N, E); }
int t4k_hit(const float *out, const float *hot, int N, int E, int *cnt, t4k_stream_t) { return t4o_hit(out, hot, N, E, cnt); }
int t4k_onehot_hit(const uint32_t *l, float *hot, const float *out, int N, int E, int *cnt, t4k_stream_t) { int r = t4o_onehot(l, hot, N, E); return r ? r : t4o_hit(out, hot, N, E, cnt); }
int t4k_u8_normalize(const uint8_t *s, float *d, long n, float mean, float scale, t4k_stream_t) { return t4o_u8_normalize(s, d, n, mean, scale); }
int t4k_stage_batch(const uint8_t *src, float *dst, long n, float mean, float scale, const uint32_t *ls, uint32_t *ld, int nlab, t4k_stream_t) {
    if (nlab > 0) memcpy(ld, ls, sizeof(uint32_t) * (size_t)nlab);
    return n > 0 ? t4o_u8_normalize(src, dst, n, mean, scale) : 0;
}
int t4k_linear_fwd(const float *X, const float *W, const float *B, float *Y, int N, int E0, int E1, t4k_stream_t) {
    memset(Y, 0, sizeof(float) * (size_t)N * E0);
    return t4o_linear_fwd(X, W, B, Y, N, E0, E1);
}
int t4k_linear_act_fwd(const float *X, const float *W, const float *B, float *Y, int layer, float alpha, float *F, float *A, int N, int E0, int E1, t4k_stream_t st) {
    int r = t4k_linear_fwd(X, W, B, Y, N, E0, E1, st); if (r) return r;
    if (layer == T4K_L_DROPOUT) t4o_dropout_mask(F, (long)N * E0);
    return rc(t4o_activate(layer, Y, A, F, alpha, (long)N * E0), "k_activate");
}
int t4k_poolblock_fwd(const float *X, const t4k_poolblock *b, int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t);
int t4k_linear_block_fwd(const float *X, float *XCOPY, const float *W, const float *B, float *Y, const t4k_poolblock *blk, int N, int E0, int E1, t4k_stream_t st) {
    if (blk && (blk->pool_layer || blk->copy_out || blk->KS != 1)) return rc(T4K_ERR_UNSUPPORTED, "t4k_linear_block_fwd: the run behind a linear layer has no pool / flatten stage");
    if (XCOPY && XCOPY != X) t4o_copy(X, XCOPY, (long)N * E1);
    int r = t4k_linear_fwd(X, W, B, Y, N, E0, E1, st); if (r) return r;
    if (blk && (blk->pre_layer || blk->post_layer)) return t4k_poolblock_fwd(Y, blk, N, 1, 1, 1, 1, E0, st);
    return T4K_OK;
}
int t4k_linear_bwd2(const float *X, const float *W, const float *DY, float *DX, const float *MASK, float *DXM, float *DW, float *DB, int N, int E0, int E1, int tr, t4k_stream_t st) {
    int r = t4k_linear_bwd(X, W, DY, DX, DW, DB, N, E0, E1, tr, st); if (r) return r;
    if (DXM) return rc(t4o_tt_op(T4K_MUL, DX, MASK, DXM, (long)N * E1), "k_tt_op");
    return T4K_OK;
}
int t4k_poolblock_bwd(const float *DY, float *X, const t4k_poolblock *b, int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t);
int t4k_linear_block_bwd(const float *X, const float *W, float *DY, const float *TGT, float *DY2, float *DX, const t4k_poolblock *blk, float *XRUN,
                         float *DW, float *DB, int N, int E0, int E1, int tr, t4k_stream_t st) {
    if (!blk || blk->pool_layer || blk->copy_out || blk->KS != 1) return rc(T4K_ERR_UNSUPPORTED, "t4k_linear_block_bwd: the run in front of a linear layer has no pool / flatten stage");
    if (TGT) { int r = t4o_tt_op(T4K_SUB, DY, TGT, DY, (long)N * E0); if (r) return rc(r, "k_tt_op"); if (DY2) t4o_copy(DY, DY2, (long)N * E0); }
    int r = t4k_linear_bwd(X, W, DY, DX, DW, DB, N, E0, E1, tr, st); if (r) return r;
    return t4k_poolblock_bwd(DX, XRUN, blk, N, 1, 1, 1, 1, E1, st);
}
int t4k_mlp_head_fwd(const float *X, const float *W1, const float *B1, float *Y1, int layer, float alpha, float *F1, float *A1,
                     const float *W2, const float *B2, float *Y2, float *P2, int N, int H, int E1, int E2, t4k_stream_t st) {
    int r = t4k_linear_act_fwd(X, W1, B1, Y1, layer, alpha, F1, A1, N, H, E1, st); if (r) return r;
    if (P2) return t4k_linear_softmax_fwd(A1, W2, B2, Y2, P2, N, E2, H, st);
    return t4k_linear_fwd(A1, W2, B2, Y2, N, E2, H, st);
}
int t4k_loss_linear_bwd(const float *X, const float *W, float *OUT, const float *TGT, float *OUT2, float *DX, const float *MASK, float *DXM,
                        float *DW, float *DB, int N, int E0, int E1, int tr, t4k_stream_t st) {
    int r = t4k_tt_op2(T4K_SUB, OUT, TGT, OUT, OUT2, (long)N * E0, st); if (r) return r;
    return t4k_linear_bwd2(X, W, OUT, DX, MASK, DXM, DW, DB, N, E0, E1, tr, st);
}
int t4k_linear_softmax_fwd(const float *X, const float *W, const float *B, float *Y, float *P, int N, int E0, int E1, t4k_stream_t st) {
    int r = t4k_linear_fwd(X, W, B, Y, N, E0, E1, st); if (r) return r;
    return t4o_softmax(Y, P, N, E0);
}
int t4k_linear_bwd(const float *X, const float *W, const float *DY, float *DX, float *DW, float *DB, int N, int E0, int E1, int tr, t4k_stream_t) {
    float *dx = DX;                                     // product contract: DX == NULL -> dW|dB only, DW == NULL -> dX only
    if (!dx) dx = (float *)calloc((size_t)N * E1, sizeof(float));
    int r = t4o_linear_bwd(X, W, DY, dx, DW, DB, N, E0, E1, DW ? tr : 0);
    if (!DX) free(dx);
    return r;
}
// fused runs: the oracle composes the unfused layer functions - that composition IS the parity statement
int t4k_poolblock_fwd(const float *X, const t4k_poolblock *b, int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t) {
    const long n1 = (long)N * H1 * W1 * C, n0 = (long)N * H0 * W0 * C;
    const float *x = X;
    if (b->pre_layer) {
        if (b->pre_layer == T4K_L_DROPOUT) t4o_dropout_mask(b->pre_mask, n1);
        int r = t4o_activate(b->pre_layer, x, b->pre_out, b->pre_mask, b->pre_alpha, n1); if (r) return rc(r, "poolblock pre");
        x = b->pre_out;
    }
    if (b->pool_layer) { int r = t4o_pool(b->pool_layer, x, b->pool_out, N, H1, W1, H0, W0, C, b->KS); if (r) return rc(r, "poolblock pool"); x = b->pool_out; }
    if (b->post_layer) {
        if (b->post_layer == T4K_L_DROPOUT) t4o_dropout_mask(b->post_mask, n0);
        int r = t4o_activate(b->post_layer, x, b->post_out, b->post_mask, b->post_alpha, n0); if (r) return rc(r, "poolblock post"); x = b->post_out;
    }
    if (b->copy_out) t4o_copy(x, b->copy_out, n0);
    return T4K_OK;
}
int t4k_poolblock_bwd(const float *DY, float *X, const t4k_poolblock *b, int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t) {
    const long n1 = (long)N * H1 * W1 * C, n0 = (long)N * H0 * W0 * C;
    const float *g = DY;
    float *last = b->post_layer ? b->post_out : (b->pool_layer ? b->pool_out : (b->pre_layer ? b->pre_out : X));
    if (b->copy_out) { t4o_copy(g, last, n0); g = last; }                       // flatten: in = out
    if (b->post_layer) {
        float *in = b->pool_layer ? b->pool_out : (b->pre_layer ? b->pre_out : X);
        t4o_tt_op(T4K_MUL, g, b->post_mask, in, n0); g = in;
    }
    if (b->pool_layer) {
        float *in = b->pre_layer ? b->pre_out : X;
        int r = t4o_dpool(b->pool_layer, in, g, N, H1, W1, H0, W0, C, b->KS); if (r) return rc(r, "poolblock dpool");
        g = in;
    }
    if (b->pre_layer) t4o_tt_op(T4K_MUL, g, b->pre_mask, X, n1);
    return T4K_OK;
}
int t4k_conv2d_block_fwd(const float *I, float *IC, float *O, const float *F, const float *B, const t4k_poolblock *blk, int N, int H1, int W1, int C1,
                         int H0, int W0, int C0, int K, int S, int P, t4k_stream_t st) {
    int r = t4k_conv2d_fwd2(I, IC, O, F, B, N, H1, W1, C1, H0, W0, C0, K, S, P, st); if (r) return r;
    return t4k_poolblock_fwd(O, blk, N, H0, W0, H0 / blk->KS, W0 / blk->KS, C0, st);
}
static const float *g_keep_src = nullptr; static float *g_keep_dst = nullptr;
int t4k_opt_snapshot(const float *G, float *G_PREV) { g_keep_src = G; g_keep_dst = G ? G_PREV : nullptr; return T4K_OK; }
int t4k_opt_snapshot_pending(void) { return g_keep_src ? 1 : 0; }
int t4k_opt_multi(int kind, const t4k_param_rec *tab, int nt, long, float lr, float b1, float b2, float wd, t4k_stream_t) {
    for (int i = 0; i < nt; i++) {
        const t4k_param_rec &r = tab[i];
        if (g_keep_src && r.G == g_keep_src) { memcpy(g_keep_dst, r.G, sizeof(float) * (size_t)r.n); g_keep_src = nullptr; g_keep_dst = nullptr; }
        if (kind == 0) t4o_sgd(r.G, r.DG, r.M, r.Nw, lr, b1, r.n);
        else if (kind == 1) t4o_adam(r.G, r.DG, r.M, r.V, lr, b1, b2, r.n);
        else t4o_adamw(r.G, r.DG, r.M, r.V, lr, b1, b2, wd, r.n);
    }
    return T4K_OK;
}
int t4k_opt_chunked(int kind, const t4k_param_rec *tab, int nt, int, float lr, float b1, float b2, float wd, t4k_stream_t st) {
    return t4k_opt_multi(kind, tab, nt, 0, lr, b1, b2, wd, st);
}

int t4k_opt_step(int kind, const t4k_param_rec *tab, const t4k_param_rec *, int nt, int, float lr, float b1, float b2, float wd, t4k_stream_t st) {
    return t4k_opt_multi(kind, tab, nt, 0, lr, b1, b2, wd, st);
}

int t4k_opt_step_dp(int kind, const t4k_param_rec *tab, const t4k_param_rec *, int nt, int, float lr, float b1, float b2, float wd, float *, long, t4k_stream_t st) {
    return t4k_opt_multi(kind, tab, nt, 0, lr, b1, b2, wd, st);
}
int t4k_xchg_world(void) { return 0; }
int t4k_xchg_trust(int) { return T4K_OK; }
int t4k_xchg_active(void) { return 0; }

// the sample-resident conv stack is a launch-count optimisation of the product: the oracle VM always runs the separate layers
int t4k_conv_stack_ok(const t4k_conv_stage *, int, int) { return 0; }
int t4k_conv_stack_release(const float *) { return T4K_OK; }
int t4k_conv_stack_dx0_pending(const float *) { return 0; }
int t4k_conv_stack_dx0(const t4k_conv_stage *, int, t4k_stream_t) { return T4K_OK; }
int t4k_conv_stack_fwd(const float *, float *, const t4k_conv_stage *, int, int, t4k_stream_t) { return T4K_ERR_UNSUPPORTED; }
int t4k_conv_stack_bwd(const float *, const t4k_conv_stage *, int, int, int, t4k_stream_t) { return T4K_ERR_UNSUPPORTED; }
int t4k_conv_stack_bwd_ok(const t4k_conv_stage *, int, int, int, t4k_stream_t) { return 0; }
int t4k_conv_stack_head_ok(const t4k_conv_stage *, int, int, const t4k_stack_head *) { return 0; }
int t4k_mlp_head_bwd_ok(int, int, int, int) { return 0; }
int t4k_mlp_block_bwd(float *, const float *, float *, const float *, float *, const t4k_poolblock *, float *, float *, float *, float *, const float *, const t4k_poolblock *, float *, float *, float *, int, int, int, int, int, t4k_stream_t) { return T4K_ERR_UNSUPPORTED; }
int t4k_mlp_head_bwd(float *, const float *, float *, const float *, float *, const float *, float *, float *, float *, float *, const float *, float *, float *, int, int, int, int, t4k_stream_t) { return T4K_ERR_UNSUPPORTED; }
int t4k_conv_stack_head_fwd(const float *, float *, const t4k_conv_stage *, int, int, const t4k_stack_head *, t4k_stream_t) { return T4K_ERR_UNSUPPORTED; }

} // extern "C"
