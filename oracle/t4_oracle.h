/*
 * t4_oracle.h - CPU oracle for the tensorForth hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain serial restatement of the reference's CUDA
 * kernels (file:line cited at each function in t4_oracle.cpp).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the product
 * library (tensorforth_amd/csrc -> libt4hip.so) never links or loads it.
 *
 * Parity pinning: the oracle is checked against the reference's own known-answer scripts
 * (examples/t4_30a/b/c, t4_20a, t4_22a expected values, committed under tests/golden/);
 * ops no reference test pins (conv, pool, softmax/CE, batchnorm, Adam) are cross-checked
 * against torch-CPU in tests/test_oracle_vs_torch.py with the reference quirks asserted.
 *
 * All pointers are HOST pointers; signatures mirror include/t4k.h minus the stream.
 */
#ifndef T4_ORACLE_H_
#define T4_ORACLE_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int t4o_reduce(int red_op, const float *src, long n, float avg, float *out);
int t4o_nan_inf(const float *src, long n, int *cnt);
int t4o_copy(const float *src, float *dst, long n);
int t4o_transpose(const float *src, float *dst, int H, int W, int C);
int t4o_identity(float *dst, int H, int W, int C);
int t4o_math(int op, float *A, float v, long n);
int t4o_ts_op(int op, const float *A, float v, float *O, long n);
int t4o_tt_op(int op, const float *A, const float *B, float *O, long n);
int t4o_bce(const float *T, const float *O, long n, float *out);
int t4o_dot(const float *A, const float *B, float *O, float alpha, float beta, int K, int C);
int t4o_gemm(const float *A, const float *B, float *O, float alpha, float beta,
             int tA, int tB, int M, int N, int K, int C);
int t4o_gemm_f64acc(const float *A, const float *B, float *O, float alpha, float beta,
                    int M, int N, int K, int C);
/* the reference's own host GEMM (word `gemm`), used as the CPU baseline */
int t4o_gemm_host_blocked(const float *A, const float *B, float *O, float alpha, float beta,
                          int H, int W, int Ka);

int t4o_inverse(float *A, float *I, int K, int *status);
int t4o_plu(float *A, float *I, int *piv, int K, int *status);
int t4o_lu_inverse(float *A, float *I, int *piv, int K, int *status);
int t4o_lu_extract(float *LU, int get_u, int K);
int t4o_logdet(const float *LU, int K, float *logdet, int *sign);

int      t4o_rand_init(uint64_t seed);
int      t4o_rand(float *d, long n, int opt, float bias, float scale);
uint64_t t4o_rand_offset(void);
int      t4o_rand_set_offset(uint64_t off);
int      t4o_rand_set_shard(int rank, int world);   /* include/t4k.h t4k_rand_set_shard */
uint64_t t4o_rand_seed(void);
int      t4o_rand_shard_world(void);
int      t4o_dropout_mask(float *mask, long n);      /* include/t4k.h t4k_dropout_mask   */

int t4o_bias(const float *B, float *O, int N, int E0);
int t4o_activate(int layer, const float *I, float *O, float *F, float alpha, long n);
int t4o_softmax(const float *I, float *O, int N, int C);
int t4o_batchnorm_fwd(const float *I, float *O, float *XH, const float *W, const float *B,
                      float *stat, int N, int HW, int C);
int t4o_batchnorm_bwd(const float *W, const float *DY, const float *XH, float *DX,
                      float *DW, float *DB, float *stat, int N, int HW, int C, int train);
int t4o_dlinear_db(const float *DY, float *DB, int N, int E0);
int t4o_conv2d_fwd(const float *I, float *O, const float *F, const float *B,
                   int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P);
int t4o_conv2d_bwd(const float *I, const float *DO, float *DX, const float *F,
                   float *DF, float *DB,
                   int N, int H1, int W1, int C1, int H0, int W0, int C0,
                   int K, int S, int P, int train);
/* transposed convolution layer (include/t4k.h t4k_dconv2d_fwd / _bwd): direct loops over the definition */
int t4o_dconv2d_fwd(const float *I, float *O, const float *F, const float *B,
                    int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P);
int t4o_dconv2d_bwd(const float *I, const float *DO, float *DX, const float *F, float *DF, float *DB,
                    int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P, int train);
int t4o_pool(int layer, const float *I, float *O, int N, int H1, int W1, int H0, int W0, int C, int KS);
int t4o_dpool(int layer, float *I, const float *DY, int N, int H1, int W1, int H0, int W0, int C, int KS);
int t4o_sgd(float *G, float *DG, float *M, int Nw, float lr, float beta, long n);
int t4o_adam(float *G, float *DG, float *M, float *V, float lr, float b1, float b2, long n);
int t4o_adamw(float *G, float *DG, float *M, float *V, float lr, float b1, float b2, float wd, long n);
int t4o_onehot(const uint32_t *label, float *hot, int N, int E);
int t4o_hit(const float *out, const float *hot, int N, int E, int *cnt);
int t4o_u8_normalize(const uint8_t *src, float *dst, long n, float mean, float scale);
int t4o_linear_fwd(const float *X, const float *W, const float *B, float *Y, int N, int E0, int E1);
int t4o_linear_bwd(const float *X, const float *W, const float *DY, float *DX,
                   float *DW, float *DB, int N, int E0, int E1, int train);

#ifdef __cplusplus
}
#endif
#endif
