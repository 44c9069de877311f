/*
 * t4k.h - C-ABI of the MI355X (gfx950) tensor/CNN kernel backend for tensorForth.
 *
 * This is the drop-in boundary: every entry point below replaces one CUDA kernel
 * (or one host wrapper + kernel pair) of the reference, cited as file:line
 * relative to the reference tree.  All pointers are plain device pointers
 * (fp32 unless noted), all sizes plain ints/longs; no C++ / torch types.
 *
 * Conventions (reference src/mu/tensor.h:51-115):
 *   - tensors are fp32, NHWC contiguous; a sample slice is data + n*H*W*C
 *   - every call returns an int status (T4K_OK == 0); nothing throws or aborts
 *     (reference convention is print-and-continue, src/ten4_types.h:25,186-191)
 *   - calls are asynchronous on `stream` (NULL = the library's default stream);
 *     the caller synchronises (t4k_sync) before touching results on the host.
 *     The reference syncs after every launch (GPU_CHK, src/ten4_types.h:192);
 *     here the host layer syncs only at the host-touch sites.
 *   - kernels never allocate; scratch comes from a library-owned workspace that
 *     replaces the reference's per-tensor `_tmp` slot (src/mu/tensor.cu:481).
 */
#ifndef T4K_H_
#define T4K_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *t4k_stream_t;            /* hipStream_t */
typedef void *t4k_event_t;             /* hipEvent_t  */
typedef void *t4k_graph_t;             /* hipGraphExec_t */

enum {
    T4K_OK              =  0,
    T4K_ERR_ARG         = -1,          /* bad shape / null pointer / unsupported parameter   */
    T4K_ERR_HIP         = -2,          /* HIP runtime error (see t4k_last_error)             */
    T4K_ERR_NOMEM       = -3,
    T4K_ERR_UNSUPPORTED = -4,          /* e.g. conv (K,S,P) outside the reference's set      */
    T4K_ERR_NODEVICE    = -5,          /* no gfx950 device: the product path fails loudly    */
    T4K_ERR_SINGULAR    = -6           /* singular matrix in inverse/plu                     */
};

/* math_op: values identical to reference src/t4math.h:25-56 */
enum {
    T4K_ABS = 0, T4K_NEG, T4K_EXP, T4K_LN, T4K_LOG, T4K_TANH, T4K_RELU, T4K_SIGM,
    T4K_SQRT, T4K_RCP, T4K_SAT, T4K_IDEN, T4K_FILL, T4K_GFILL, T4K_SCALE, T4K_POW,
    T4K_ADD, T4K_SUB, T4K_MUL, T4K_DIV, T4K_MOD, T4K_MAX, T4K_MIN, T4K_MUL2, T4K_MOD2,
    T4K_SIN, T4K_COS
};
/* t4_layer: values identical to reference src/nn/ntypes.h:16-36 */
enum {
    T4K_L_NONE = 0, T4K_L_CONV, T4K_L_LINEAR, T4K_L_FLATTEN, T4K_L_RELU, T4K_L_TANH,
    T4K_L_SIGMOID, T4K_L_SELU, T4K_L_LEAKYRL, T4K_L_ELU, T4K_L_DROPOUT, T4K_L_SOFTMAX,
    T4K_L_LOGSMAX, T4K_L_AVGPOOL, T4K_L_MAXPOOL, T4K_L_MINPOOL, T4K_L_BATCHNM,
    T4K_L_USAMPLE, T4K_L_DCONV
};
/* rand_opt: reference src/util.h:22-25 */
enum { T4K_UNIFORM = 0, T4K_NORMAL = 1 };
/* reduction selector for t4k_reduce */
enum { T4K_RED_SUM = 0, T4K_RED_NVAR, T4K_RED_MAX, T4K_RED_MIN };

/* ------------------------------------------------------------------ runtime */
/* Replaces cudaSetDevice / cudaMallocManaged arena / cudaMemcpy / cudaDeviceSynchronize
 * uses (src/ten4.cu:125-152, src/mu/mmu.cu:44-46, src/ten4_types.h:192-201). */
int         t4k_device_count(void);
int         t4k_init(int device);                  /* select device, create default stream + workspace */
void        t4k_shutdown(void);
/* Kernels whose workgroups wait for each other (arrival gates of the one-launch dW || dX products, pair-mode tickets, the band exchange of the conv-stack
 * head, column-stripe tickets) need every workgroup of the launch resident at once: true on an exclusively owned device, not on a partitioned or shared
 * one.  t4k_gates_enable(0) makes every launcher pick its ungated path (more launches, same results); a wait that times out does it by itself, after
 * reporting T4K_ERR_HIP once for the launch it spoiled (runtime.hip spin_check). */
int t4k_gates_enable(int on);
int t4k_gates_enabled(void);
const char *t4k_last_error(void);
const char *t4k_backend_name(void);                /* "hip-gfx950" for the product library */
int         t4k_device_info(int *cu_count, int *clock_khz, size_t *hbm_bytes);

int t4k_malloc(void **p, size_t bytes);            /* device (HBM) allocation */
int t4k_free(void *p);
int t4k_host_alloc(void **p, size_t bytes);        /* pinned host staging */
int t4k_host_free(void *p);
int t4k_memcpy_h2d(void *dst, const void *src, size_t bytes, t4k_stream_t s);
int t4k_memcpy_d2h(void *dst, const void *src, size_t bytes, t4k_stream_t s);
int t4k_memcpy_d2d(void *dst, const void *src, size_t bytes, t4k_stream_t s);
int t4k_memset(void *dst, int byte, size_t bytes, t4k_stream_t s);   /* Tensor::zeros tensor.cu:558-563 */
int t4k_sync(t4k_stream_t s);
/* Kernels launched by the library since t4k_init (every launch site counts itself): measurement hook - bench.py brackets its timed
 * loop with it and prints the MEASURED launches per step (the reference issues one launch + cudaDeviceSynchronize per layer,
 * src/nn/forward.cu:82-113, backprop.cu:111-140; memcpy / memset / collective commands are not kernels and are not counted). */
unsigned long long t4k_launch_count(void);

/* Library streams own a private workspace, so independent work (e.g. dW beside dX) may be forked
 * onto them and still be captured into one graph through event edges. */
int t4k_stream_create(t4k_stream_t *s);
/* A stream WITHOUT a library workspace: copies and element-wise launches only (the batch prefetch of the dataset feed,
 * src/mu/dataset.cu:112 "TODO: async prefetch"); not a lane - kernels that need split-K / partial slabs refuse or serialise on it. */
int t4k_stream_create_plain(t4k_stream_t *s);
int t4k_stream_destroy(t4k_stream_t s);
int t4k_stream_wait_event(t4k_stream_t s, t4k_event_t e);   /* fork / join edge (also inside a capture) */
int t4k_set_default_stream(t4k_stream_t s);        /* adopt an external stream (e.g. torch's current) */
t4k_stream_t t4k_default_stream(void);

int t4k_event_create(t4k_event_t *e);
int t4k_event_record(t4k_event_t e, t4k_stream_t s);
int t4k_event_sync(t4k_event_t e);
/* the wait of a helper thread (host/dataset.cpp's reader): only the wait - no deferred work of the model's thread is run, the wait-error word is left alone */
int t4k_event_wait(t4k_event_t e);
int t4k_event_elapsed_ms(t4k_event_t start, t4k_event_t stop, float *ms);
int t4k_event_destroy(t4k_event_t e);

/* ---------------------------------------------------------------- data-parallel exchange
 * One process per GPU.  Rank 0 makes a 128-byte id, the launcher hands it to every rank (any side channel), each
 * rank joins; afterwards t4k_allreduce_sum() sums a device buffer in place over all ranks on the given stream
 * (RCCL over xGMI).  The host VM calls it on the model's gradient slab between `backprop` and the optimizer. */
int t4k_comm_unique_id(void *id128);
int t4k_comm_init(const void *id128, int rank, int world);
int t4k_comm_world(void);                          /* 0 = no communicator */
int t4k_comm_rank(void);
int t4k_allreduce_sum(float *buf, long n, t4k_stream_t s);
int t4k_comm_destroy(void);
/* Batch-norm statistics over the WHOLE batch (all ranks) instead of the rank's shard: off by default.  When on, every
 * t4k_batchnorm_fwd / _bwd issues one [2C] all-reduce, so EVERY rank must make the same batch-norm calls in the same order - the host
 * switches it on for training passes only (an evaluation pass that only some ranks run would hang in the collective). */
int t4k_comm_sync_batchnorm(int on);

/* ONE-SHOT gradient exchange over peer-mapped windows (csrc/xchg.hip; the reference has no multi-GPU path - the seam is Model::sgd / adam,
 * src/nn/gradient.cu:63-142).  Every rank allocates a receive window (t4k_xchg_create -> a 64-byte IPC handle), the launcher hands all
 * handles to all ranks, every rank maps its peers (t4k_xchg_connect).  From then on t4k_opt_step_dp() sums the gradient slab over all ranks
 * INSIDE the optimizer launch: each rank writes {epoch, value} words straight into every peer's window (xGMI is point to point: all links
 * at once) and adds the `world` copies of each element in rank order - fold + all-reduce + update in one launch, no collective kernel.
 * t4k_comm_world / t4k_comm_rank report the exchange's job when no RCCL communicator exists, and t4k_allreduce_sum then sums over the same
 * windows, so the host's scalar reductions need no second transport.  Dropout masks are keyed by the sample's place in the whole batch
 * from t4k_xchg_connect on (as after t4k_comm_init).  Waits for a peer are bounded: a missing rank gives T4K_ERR_HIP at the next t4k_sync. */
int t4k_xchg_create(long slab_floats, int rank, int world, void *handle64);
int t4k_xchg_connect(const void *handles /* world x 64 bytes, rank order */);
int t4k_xchg_allreduce(float *buf, long n, t4k_stream_t s);   /* in-place SUM over the ranks through the windows (any n; rank order: deterministic) */
int t4k_xchg_trust(int on);                        /* the launcher's verdict on the known-sum probe (tensorforth_amd/dp.py): waits are bounded by 2 s until it is 1, by T4K_XCHG_TIMEOUT_MS (20 s) after */
int t4k_xchg_self(int on);                         /* measurement: a one-rank job takes the exchanging optimizer launch too (bench.py dp_overhead_us) */
int t4k_xchg_active(void);                         /* 1 when t4k_opt_step_dp will exchange (connected and world > 1, or self mode) */
int t4k_xchg_world(void);                          /* 0 = not connected */
int t4k_xchg_rank(void);
int t4k_xchg_destroy(void);

/* hipGraph capture of a launch sequence (replaces ~40 launch+sync pairs per training
 * step of the reference).  begin..end captures every t4k_* kernel call
 * issued on `s`; launch replays it. */
int t4k_graph_begin(t4k_stream_t s);
int t4k_graph_end(t4k_stream_t s, t4k_graph_t *g);
int t4k_graph_launch(t4k_graph_t g, t4k_stream_t s);
int t4k_graph_destroy(t4k_graph_t g);

/* ------------------------------------------------- tensor kernels (t4math.cu) */
/* k_sum :23, k_nvar :48, k_max/d__max :85-131 via Tensor::sum/std/norm/max/min
 * (tensor.cu:224-277).  Result is written (not accumulated) to *out_dev. */
int t4k_reduce(int red_op, const float *src, long n, float avg, float *out_dev, t4k_stream_t s);
/* k_nan_inf :278 / Tensor::has_nan tensor.cu:326-333: count of NaN/Inf -> *cnt_dev (int) */
int t4k_nan_inf(const float *src, long n, int *cnt_dev, t4k_stream_t s);
/* k_copy :134 */
int t4k_copy(const float *src, float *dst, long n, t4k_stream_t s);
/* k_transpose :150 (one sample): dst[(H*j+i)*C+c] = src[(W*i+j)*C+c]; bit-exact */
int t4k_transpose(const float *src, float *dst, int H, int W, int C, t4k_stream_t s);
/* k_identity :160 (one sample) */
int t4k_identity(float *dst, int H, int W, int C, t4k_stream_t s);
/* k_math :173 in-place unary / scalar op (LN/LOG clamp 1e-12, SQRT clamp 0, GFILL = v*j/n) */
int t4k_math(int op, float *A, float v, long n, t4k_stream_t s);
/* k_ts_op :206  O = A op v,  op in {ADD,SUB,MUL,DIV} */
int t4k_ts_op(int op, const float *A, float v, float *O, long n, t4k_stream_t s);
/* k_tt_op :222  O = A op B */
int t4k_tt_op(int op, const float *A, const float *B, float *O, long n, t4k_stream_t s);
/* same with a second destination O2 (may be NULL): `out -= target` plus the pass-through copy `in = out` of a
 * final softmax/sigmoid layer (backprop.cu:129-131) in one launch */
int t4k_tt_op2(int op, const float *A, const float *B, float *O, float *O2, long n, t4k_stream_t s);
/* k_bce :248  *out_dev = sum t*ln(o+eps) + (1-t)*ln(1-o+eps), eps = 1e-6 */
int t4k_bce(const float *T, const float *O, long n, float *out_dev, t4k_stream_t s);
/* k_dot :309  O[c] = alpha*sum_k A[k*C+c]*B[k*C+c] + beta*O[c]  for c in [0,C) */
int t4k_dot(const float *A, const float *B, float *O, float alpha, float beta,
            int K, int C, t4k_stream_t s);
/* k_gemm_tile_claude :478 (Tensor::gemm3 tensor.cu:161, Tensor::linear :79):
 *   O[M,N,C] = alpha * op(A) @ op(B) + beta * O, per channel c (element stride C)
 *   op(A) = tA ? A stored [K,M,C] : A stored [M,K,C];  op(B) = tB ? B[N,K,C] : B[K,N,C]
 * fp32 accumulate on MFMA.  beta*O is read even when beta==0 (reference quirk, SURVEY a-1)
 * only if T4K_GEMM_BETA0_READS (default off: beta==0 never reads O). */
int t4k_gemm(const float *A, const float *B, float *O, float alpha, float beta,
             int tA, int tB, int M, int N, int K, int C, t4k_stream_t s);
/* k_gemm :370 / k_gemm_claude :411 (words gemm1/gemm2): double accumulator, tA/tB ignored */
int t4k_gemm_f64acc(const float *A, const float *B, float *O, float alpha, float beta,
                    int M, int N, int K, int C, t4k_stream_t s);

/* ------------------------------------------------ linear algebra (t4math.cu) */
/* Tensor::inverse tensor.cu:344-369 (k_find_pivot/k_swap_rows/k_diag/k_elim :742-836):
 * Gauss-Jordan with partial pivoting on A[K,K] and I[K,K] in place; whole host loop runs
 * on the device.  *status_dev (int): 0 ok, z+1 = singular at column z. */
int t4k_inverse(float *A, float *I, int K, int *status_dev, t4k_stream_t s);
/* Tensor::plu tensor.cu:371-398 (k_lu_col :854, k_pivot :887): A -> packed L\U in place,
 * piv_dev[K] pivot rows; if I != NULL and I != A, apply the row swaps to I (=> P). */
int t4k_plu(float *A, float *I, int *piv_dev, int K, int *status_dev, t4k_stream_t s);
/* Tensor::lu_inverse tensor.cu:400-417 (k_fsub :904, k_bsub :920) */
int t4k_lu_inverse(float *A, float *I, int *piv_dev, int K, int *status_dev, t4k_stream_t s);
/* Tensor::lu tensor.cu:419-429 (k_lu :936): keep U (get_u) or unit-L of a packed L\U */
int t4k_lu_extract(float *LU, int get_u, int K, t4k_stream_t s);
/* k_logdet :952: *logdet_dev = sum ln|U[j,j]|, *sign_dev = prod sign(U[j,j]) */
int t4k_logdet(const float *LU, int K, float *logdet_dev, int *sign_dev, t4k_stream_t s);

/* ------------------------------------------------------- RNG (util.cu:28-70) */
/* Counter-based Philox4x32-10 replaces the reference's 1024 cuRAND XORWOW states
 * (distribution parity only; the reference seeds from time(), sys.cpp:37). */
/* The host keeps the stream position (seed, counter): an eager draw gets its slice as kernel arguments.  Draws recorded
 * between t4k_graph_begin/end read a device copy instead and advance it themselves, so every replay gets a fresh slice;
 * t4k_graph_launch keeps host and device positions in step.  t4k_rand_offset()/set_offset() are host-only and cheap. */
int t4k_rand_init(uint64_t seed);
/* d[i] = scale * (bias + u_i), u uniform (0,1] or N(0,1) (util.cu:58-70) */
int t4k_rand(float *d, long n, int opt, float bias, float scale, t4k_stream_t s);
uint64_t t4k_rand_offset(void);                    /* current stream offset (for checkpoint/tests) */
int t4k_rand_set_offset(uint64_t off);
uint64_t t4k_rand_seed(void);                      /* the seed of the stream (a host that embeds several VMs saves / restores (seed, offset) per VM) */
/* Data-parallel shard of the stream (SURVEY 8e "dropout: per-rank Philox offset = global sample index").  With a shard (rank, world)
 * set, a dropout-mask draw of n elements takes elements [rank*n, (rank+1)*n) of the n*world-element draw the whole batch would make
 * and moves the stream by n*world, so `world` ranks x batch B draw exactly the masks of one rank x batch B*world (n % 4 == 0; every
 * rank must make the same draws in the same order).  Every other draw (weight init, `rand` words) stays replicated.  t4k_comm_init
 * sets the shard to the communicator's (rank, world), t4k_comm_destroy resets it to (0, 1). */
int t4k_rand_set_shard(int rank, int world);
int t4k_rand_shard_world(void);                    /* 1 = no shard.  Sharded draws are keyed on the host per launch: a host must not replay captured graphs while world > 1 */
/* the mask of a dropout layer (Model::_fstep L_DROPOUT forward.cu:100-103 + t4_rand): uniform (0,1], keyed by sample as above */
int t4k_dropout_mask(float *mask, long n, t4k_stream_t s);

/* --------------------------------------------------- nn kernels (nn/nmath.*) */
/* k_bias nmath.cu:27  O[n,e] += B[e] */
int t4k_bias(const float *B, float *O, int N, int E0, t4k_stream_t s);
/* k_activate nmath.cu:37: layer in {RELU,TANH,SIGMOID,SELU,LEAKYRL,ELU,DROPOUT};
 * writes output O and derivative mask F (for DROPOUT, F holds uniform randoms on entry) */
int t4k_activate(int layer, const float *I, float *O, float *F, float alpha, long n, t4k_stream_t s);
/* k_softmax_small / k_softmax nmath.cu:74-169: row softmax over C per sample */
int t4k_softmax(const float *I, float *O, int N, int C, t4k_stream_t s);
/* Model::_flogsoftmax forward.cu:245-259 as written there: O[n,c] = exp(I[n,c]) - log10(max(sum_c exp(I[n,c]), 1e-6)) */
int t4k_logsoftmax(const float *I, float *O, int N, int C, t4k_stream_t s);
/* k_batchnorm_1/2/3 nmath.cu:177-264 (Model::_fbatchnorm forward.cu:263-309).
 * stat_dev[3C]: [0,C) rvar = 1/(sqrt(max(var,0))+1e-6), [C,2C) mean, [2C,3C) scratch */
int t4k_batchnorm_fwd(const float *I, float *O, float *XH, const float *W, const float *B,
                      float *stat_dev, int N, int HW, int C, t4k_stream_t s);
/* k_dbatchnorm_1/2/3 nmath.cu:295-414 (Model::_bbatchnorm backprop.cu:311-370):
 * dX = gamma*rvar*(dY - mean(dY) - xhat*mean(dY*xhat)); dB,dW += the MEANS (quirk a-17) */
int t4k_batchnorm_bwd(const float *W, const float *DY, const float *XH, float *DX,
                      float *DW, float *DB, float *stat_dev, int N, int HW, int C,
                      int train, t4k_stream_t s);
/* k_dlinear_db nmath.cu:274  DB[e] += sum_n DY[n,e] */
int t4k_dlinear_db(const float *DY, float *DB, int N, int E0, t4k_stream_t s);
/* k_conv2d<TS,KS,S,P> nmath.tcu:34 (Model::_fconv forward.cu:125-155):
 * O[n,i,j,c0] = B[c0] + sum F[((c1*K+ky)*K+kx)*C0+c0] * I[n,i*S+ky-P,j*S+kx-P,c1]
 * supported (K,S,P): (1,1,0) (3,1,1) (4,2,1) (5,1,2); output is written, not accumulated */
int t4k_conv2d_fwd(const float *I, float *O, const float *F, const float *B,
                   int N, int H1, int W1, int C1, int H0, int W0, int C0,
                   int K, int S, int P, t4k_stream_t s);
/* same, plus the copy of the batch the model's layer 0 keeps (`n0 = input`, forward.cu:39): ICOPY (may be NULL) receives
 * I, from the same launch when the layer takes the few-input-channel direct kernel */
int t4k_conv2d_fwd2(const float *I, float *ICOPY, float *O, const float *F, const float *B,
                    int N, int H1, int W1, int C1, int H0, int W0, int C0,
                    int K, int S, int P, t4k_stream_t s);
/* Model::_fconv followed by Model::_fbatchnorm (forward.cu:129-155, 263-309) - a conv layer with a batch-norm layer right behind it: Y = conv(I) (+ ICOPY as above),
 * then O / XH / stat_dev exactly as t4k_batchnorm_fwd(Y, ...) writes them.  One call so that the per-channel sums of Y can leave the conv kernel's epilogue
 * (where the layer's kernel carries them) instead of costing a separate pass over Y; otherwise it IS the two calls */
int t4k_conv2d_bn_fwd(const float *I, float *ICOPY, float *Y, const float *F, const float *Bc,
                      int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P,
                      float *O, float *XH, const float *W, const float *B, float *stat_dev, t4k_stream_t s);
/* k_dconv2d<TS,KS,S,P> nmath.tcu:211 (Model::_bconv backprop.cu:152-191):
 * DB[c0] += sum dO; DF += sum I*dO (both only if train);
 * DX (overwritten) scatter (i*S+ky-P, j*S+kx-P) += F[c1,K-1-ky,K-1-kx,c0]*dO  (flipped, quirk a-11).
 * DX == NULL computes dF|dB only, DF == DB == NULL computes dX only (lets the host fork them). */
int t4k_conv2d_bwd(const float *I, const float *DO, float *DX, const float *F,
                   float *DF, float *DB,
                   int N, int H1, int W1, int C1, int H0, int W0, int C0,
                   int K, int S, int P, int train, t4k_stream_t s);
/* same, with an optional second destination for dX: the reference keeps dX in the layer's scratch tensor and
 * then copies it over the layer input (`in = dx`, backprop.cu:185); DX2 receives that copy from the same launch */
int t4k_conv2d_bwd2(const float *I, const float *DO, float *DX, float *DX2, const float *F,
                    float *DF, float *DB,
                    int N, int H1, int W1, int C1, int H0, int W0, int C0,
                    int K, int S, int P, int train, t4k_stream_t s);
/* Transposed convolution layer (word `dconv2d`, L_DCONV; allocation Model::_iconv txn model.cpp:121-180, dispatch forward.cu:110 /
 * backprop.cu:137 - the reference routes the layer's forward through the conv backward routine and vice versa but never finished it, see
 * csrc/dconv.hip).  I[N,H1,W1,C1] -> O[N,H0,W0,C0], F = T4(C1,K,K,C0), (H0-K+2P)/S+1 == H1:
 *   O[n,i*S+ky-P,j*S+kx-P,co] = B[co] + sum_ci F[ci,ky,kx,co] * I[n,i,j,ci]        (= torch ConvTranspose2d, weight[ci][co][ky][kx]) */
int t4k_dconv2d_fwd(const float *I, float *O, const float *F, const float *B,
                    int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P, t4k_stream_t s);
/* its backward: DX (overwritten) = conv(DO, F) with the same taps, DF += I x DO, DB[co] += sum DO (DF, DB only if train);
 * DX == NULL or DF == DB == NULL as for t4k_conv2d_bwd */
int t4k_dconv2d_bwd(const float *I, const float *DO, float *DX, const float *F, float *DF, float *DB,
                    int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P, int train, t4k_stream_t s);
/* k_pool<KS> nmath.tcu:122 (layer in AVGPOOL/MAXPOOL/MINPOOL/USAMPLE), KS in {2,3} */
int t4k_pool(int layer, const float *I, float *O, int N, int H1, int W1, int H0, int W0, int C,
             int KS, t4k_stream_t s);
/* k_dpool<KS> nmath.tcu:475: in place on the forward-input buffer I (reads x, zeroes the
 * tile, writes dy at the first arg-max/min; avg: dy/KS^2; usample: broadcast) */
int t4k_dpool(int layer, float *I, const float *DY, int N, int H1, int W1, int H0, int W0, int C,
              int KS, t4k_stream_t s);
/* k_sgd nmath.cu:419: dg=DG/Nw; beta~0: G-=lr*dg else M=b*M+(1-b)*dg, G-=lr*M; DG=0 */
int t4k_sgd(float *G, float *DG, float *M, int Nw, float lr, float beta, long n, t4k_stream_t s);
/* k_adam nmath.cu:438 (no bias correction, eps outside sqrt): zeroes DG */
int t4k_adam(float *G, float *DG, float *M, float *V, float lr, float b1, float b2,
             long n, t4k_stream_t s);
/* k_adamw nmath.cu:456 */
int t4k_adamw(float *G, float *DG, float *M, float *V, float lr, float b1, float b2, float wd,
              long n, t4k_stream_t s);
/* Model::backprop's start for an output layer with a derivative mask (tanh / relu / ... as the last op: backprop.cu:43-53 copies
   the target into the output tensor, _bactivate :256-263 multiplies by the mask): OUT[i] = T[i], IN[i] = T[i] * MASK[i] */
int t4k_copy_mask(const float *T, const float *MASK, float *OUT, float *IN, long n, t4k_stream_t s);
/* Model::broadcast backprop.cu:17-29: O[N,E] with O[n,e] = T[n] (a per-sample target spread over the output width) */
int t4k_broadcast_rows(const float *T, float *O, int N, int E, t4k_stream_t s);
/* Model::onehot(Dataset&) loss.cpp:47-72: hot[N,E] = 0; hot[n, label<E ? label : 0] = 1 */
int t4k_onehot(const uint32_t *label_dev, float *hot, int N, int E, t4k_stream_t s);
/* Model::hit loss.cpp:75-107: *cnt_dev = sum_n (int)hot[n, argmax_e out[n,e]] (first max wins) */
int t4k_hit(const float *out, const float *hot, int N, int E, int *cnt_dev, t4k_stream_t s);
/* both of the above in one launch (Model::forward on a dataset, forward.cu:57-60: onehot(dset) then hit(true)): hot written from the
   labels, *cnt_dev = number of rows whose first arg-max is the label's class */
int t4k_onehot_hit(const uint32_t *label_dev, float *hot, const float *out, int N, int E, int *cnt_dev, t4k_stream_t s);
/* Dataset::_load dataset.cu:123-158: dst[i] = ((float)src_u8[i] - mean) * scale */
int t4k_u8_normalize(const uint8_t *src_dev, float *dst, long n, float mean, float scale, t4k_stream_t s);
/* Dataset::fetch dataset.cu:64-121 (two host-to-device copies + the _load pass) as ONE launch off device-visible staging memory:
   dst[i] = ((float)src_u8[i] - mean) * scale for i < n, and lab_dst[j] = lab_src[j] for j < nlab (4-byte labels) */
int t4k_stage_batch(const uint8_t *src, float *dst, long n, float mean, float scale,
                    const uint32_t *lab_src, uint32_t *lab_dst, int nlab, t4k_stream_t s);

/* ---------------------------------------------- fused MI355X-native launches */
/* Model::_flinear forward.cu:157-198 in one launch:  Y[N,E0] = X[N,E1] @ W[E0,E1]^T + B[E0] */
int t4k_linear_fwd(const float *X, const float *W, const float *B, float *Y,
                   int N, int E0, int E1, t4k_stream_t s);
/* A run of element-wise layers around one pooling layer, one launch each way (csrc/fused.hip):
 *   X --[pre: dropout | activation]--> pre_out --[pool KSxKS]--> pool_out --[post: activation]--> post_out --[flatten]--> copy_out
 * Absent stages have layer == T4K_L_NONE (KS must be 1 when there is no pooling stage).  The post stage may also be a dropout layer
 * (`leakyrelu dropout`, `maxpool dropout`) when the pre stage is not one: it draws the slice t4k_dropout_mask would draw for post_mask.  Every tensor the
 * separate layers write (_factivate forward.cu:200-209, _fpool :211-227, flatten copy :96) is written with
 * identical values; a dropout pre-stage draws the Philox slice t4k_rand would have drawn for its mask. */
typedef struct t4k_poolblock {
    int    pre_layer;  float pre_alpha;  float *pre_mask;  float *pre_out;
    int    pool_layer; int KS;           float *pool_out;
    int    post_layer; float post_alpha; float *post_mask; float *post_out;
    float *copy_out;
} t4k_poolblock;
int t4k_poolblock_fwd(const float *X, const t4k_poolblock *blk, int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t s);
/* the apply half of a batch-norm layer (statistics final in stat_dev, as t4k_batchnorm_fwd leaves them) + such a run right behind it, one pass over the
 * layer's input Y: XH and O (the batch-norm layer's x-hat and output) and every tensor of the run are written exactly as t4k_batchnorm_fwd's apply +
 * t4k_poolblock_fwd(O, ...) would */
int t4k_bn_poolblock_fwd(const float *Y, float *O, float *XH, const float *W, const float *B, const float *stat_dev, const t4k_poolblock *blk,
                         int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t s);
/* convolution + batch norm + such a run (conv -> batchnorm -> [dropout|activation] -> pool -> [activation]): t4k_conv2d_bn_fwd's statistics, then
 * t4k_bn_poolblock_fwd; Hq x Wq = the pooled grid (H0 x W0 when the run has no pool layer) */
int t4k_conv2d_bn_block_fwd(const float *I, float *ICOPY, float *Y, const float *F, const float *Bc,
                            int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P,
                            float *O, float *XH, const float *W, const float *B, float *stat_dev,
                            const t4k_poolblock *blk, int Hq, int Wq, t4k_stream_t s);
/* convolution forward with such a run right behind it (conv -> [dropout|activation] -> 2x2 pool -> [activation] -> [flatten]):
 * O and every tensor of the run are written exactly as t4k_conv2d_fwd2 + t4k_poolblock_fwd would; one launch when the
 * layer takes the gather-MFMA kernel (the pool window is four consecutive accumulator registers of a lane) */
int t4k_conv2d_block_fwd(const float *I, float *ICOPY, float *O, const float *F, const float *B, const t4k_poolblock *blk,
                         int N, int H1, int W1, int C1, int H0, int W0, int C0, int K, int S, int P, t4k_stream_t s);
/* SAMPLE-RESIDENT convolution stack (csrc/conv_stack.hip): 1..3 stages of [conv2d KxK, stride 1, padding K/2 (K = 3 | 5; _fconv
 * forward.cu:115-155, k_conv2d nmath.tcu:34-104) + the element-wise run behind it (t4k_poolblock; 2x2 pool or none)], one workgroup
 * per image with the activations in LDS, ONE launch for the whole stack each way.  Stage s+1 reads the result of stage s's run
 * (H, W of stage s+1 = the pooled grid, C1 = C0 of stage s); only the last run may carry a flatten copy.  Every tensor the separate
 * layers write is written, dropout draws the same Philox slices in layer order.
 * Backward (_bconv backprop.cu:152-191, k_dconv2d nmath.tcu:211-338 incl. its un-flipped dX; _bactivate, _bpool): DY = gradient
 * w.r.t. the last run's last tensor; every run stage's input buffer receives its dX, each conv's input tensor X (forward values on
 * entry) receives dX (`in = dx`) and so does DXS when not NULL; train != 0: DF += sum over the batch, DB likewise (per-image
 * partials in the library workspace, folded in image order by a second small launch - deterministic).
 * t4k_conv_stack_fwd also leaves, per stack, a copy of every conv input and the pool arg-max codes in library memory; a backward that
 * follows it (same N) runs BANDED - several workgroups per image - on those instead of on layer tensors a neighbouring band overwrites in
 * place.  Bit 2 of `train` (train | 4) defers the fold of the dF | dB partials to t4k_opt_step (see there); bit 3 (train | 8) lets the
 * banded kernel skip the dX of stage 0 (t4k_conv_stack_dx0 below).
 * A caller whose latest forward did NOT go through t4k_conv_stack_fwd must say so with bit 1 of `train` (train | 2): the
 * whole-image kernel then reads the layer tensors themselves. */
typedef struct t4k_conv_stage {
    const float *F, *B;       /* filter T4(C1,K,K,C0), bias [C0] */
    float *O;                 /* conv output [N,H,W,C0] */
    float *DF, *DB;           /* backward: parameter gradients (accumulated); unused by the forward */
    float *X, *DXS;           /* backward: the conv's input tensor [N,H,W,C1] and the optional second dX copy */
    int H, W, C1, C0, K;
    t4k_poolblock run;        /* all-zero layers + KS = 1: no run */
} t4k_conv_stage;
int t4k_conv_stack_ok(const t4k_conv_stage *st, int n_stage, int N);           /* 1 when the stack qualifies (shapes, layers, LDS) and its kernels are built */
int t4k_conv_stack_selftest(void);   /* the stack kernels are compiled at run time (hipRTC) for the model's shapes: this compiles two reference
                                      * shape sets for gfx950 - no device needed - and returns 0, or an error with the compiler log in t4k_last_error() */
/* The stack AND the classifier head behind its flatten in ONE launch: [conv + run] x n, flatten, linear E1 -> E0a, one element-wise layer
 * (dropout / activation, or 0), linear E0a -> E0b, softmax - the layers t4k_conv_stack_fwd + t4k_mlp_head_fwd run (forward.cu:28-113,
 * 157-198, 231-243), every layer tensor stored, Philox positions as the separate layers.  t4k_conv_stack_head_ok() tells whether the shapes
 * qualify; the backward is unchanged (t4k_conv_stack_bwd reads what this forward saved, like t4k_conv_stack_fwd's). */
typedef struct t4k_stack_head {
    const float *W1, *B1; float *Y1;            /* first linear layer: W1[E0a][E1], bias, output [N][E0a]                     */
    int mid_layer; float mid_alpha;             /* element-wise layer behind it (t4k_layer; 0: none), its parameter          */
    float *mid_mask, *mid_out;                  /* derivative mask and output [N][E0a]                                      */
    const float *W2, *B2; float *Y2, *P;        /* second linear layer W2[E0b][E0a], bias, output [N][E0b]; softmax output  */
    int E1, E0a, E0b;
    /* optional (all NULL / 0: off).  The batch came from a dataset: Model::forward then turns the labels into one-hot rows and counts the hits
     * (Model::onehot(Dataset&) loss.cpp:47-72 + hit forward.cu:57-60).  The last band of image n has P[n] in hand: it writes hot[n][0..E0b)
     * (label >= E0b counts as class 0, loss.cpp:66) and hit_flag[n] = (first arg-max of P[n] == label) for n < n_label, 0 behind it - one
     * byte per image in device-visible memory (pinned host memory: the host adds them up, no atomics, no extra launch). */
    const unsigned *label; float *hot; unsigned char *hit_flag; int n_label;
} t4k_stack_head;
/* t4k_conv_stack_release: frees what the forward left behind for the banded backward of the stack whose first conv output tensor is
 * `first_conv_out` (the owner of that tensor calls it when the model is freed - Model::~Model in the reference, src/nn/model.h; unknown
 * keys are ignored).  t4k_conv_stack_stats: how this process got its stack kernels - compiled by hipRTC (*jit), loaded from the
 * code-object cache on disk (*disk: $T4K_CACHE_DIR, <library dir>/kcache as filled by build(), ~/.cache/tensorforth_amd) or not at
 * all (*failed: those stacks run the per-layer kernels; the library says so once on stderr). */
int  t4k_conv_stack_release(const float *first_conv_out);
/* Lazy dX of the FIRST conv layer.  In a classifier's training loop nobody reads the gradient w.r.t. the input batch, yet `in = dx`
 * (backprop.cu:185) makes the reference compute and store it twice per step.  t4k_conv_stack_bwd(train | 8) skips it (banded kernel only)
 * and remembers the filter it belongs to; t4k_conv_stack_dx0_pending() tells the caller whether that happened, t4k_conv_stack_dx0()
 * produces it on demand - X and DXS of stage 0 then hold what the eager backward would have stored (k_dconv2d's dX, nmath.tcu:304-324).
 * The host VM keeps a "stale" mark on the two tensors and calls it before any word can read them (host/tensor.cpp du2obj). */
int  t4k_conv_stack_dx0_pending(const float *first_conv_out);
int  t4k_conv_stack_dx0(const t4k_conv_stage *st, int N, t4k_stream_t s);
void t4k_conv_stack_stats(int *jit, int *disk, int *failed);
int t4k_conv_stack_head_ok(const t4k_conv_stage *st, int n_stage, int N, const t4k_stack_head *h);
int t4k_conv_stack_head_fwd(const float *X, float *X0, const t4k_conv_stage *st, int n_stage, int N, const t4k_stack_head *h, t4k_stream_t s);
int t4k_conv_stack_fwd(const float *X, float *XCOPY, const t4k_conv_stage *st, int n_stage, int N, t4k_stream_t s);
int t4k_conv_stack_bwd(const float *DY, const t4k_conv_stage *st, int n_stage, int N, int train, t4k_stream_t s);
/* 1 when that call would be accepted (kernels built; the banded or the whole-image kernel it would launch fits the LDS and the workspace): the host's
 * quiet test before it picks between the stack's backward and its per-layer kernels */
int t4k_conv_stack_bwd_ok(const t4k_conv_stage *st, int n_stage, int N, int train, t4k_stream_t s);
/* backward of the same run (_bactivate backprop.cu:256-263, _bpool, flatten `in = out`): DY is the gradient
 * w.r.t. the run's last tensor; each stage's input buffer receives its dX (X receives the run's dX). */
int t4k_poolblock_bwd(const float *DY, float *X, const t4k_poolblock *blk, int N, int H1, int W1, int H0, int W0, int C, t4k_stream_t s);
/* linear layer followed by an element-wise layer (t4k_layer: relu ... dropout; _flinear + _factivate forward.cu:157-209):
 * Y as above, ACT_O / ACT_F = activation output / derivative mask of Y; dropout draws the Philox slice t4k_rand would */
int t4k_linear_act_fwd(const float *X, const float *W, const float *B, float *Y, int layer, float alpha,
                       float *ACT_F, float *ACT_O, int N, int E0, int E1, t4k_stream_t s);
/* linear layer + the element-wise run behind it (Model::forward's loop over _flinear, _factivate / dropout, forward.cu:82-209) and,
 * when XCOPY != NULL, the model's copy of the batch into its layer 0 (forward.cu:39: XCOPY[N,E1] = X).  blk (may be NULL) holds the
 * run: pre and/or post stage (any t4k_layer activation or dropout, at most one dropout), no pool, no flatten copy, KS = 1.  Tensors
 * written: Y, each stage's mask and output, XCOPY - the same values as t4k_copy + t4k_linear_fwd + t4k_poolblock_fwd. */
int t4k_linear_block_fwd(const float *X, float *XCOPY, const float *W, const float *B, float *Y, const t4k_poolblock *blk,
                         int N, int E0, int E1, t4k_stream_t s);
/* backward of the same pair, seen from the linear layer behind the run: t4k_linear_bwd(X, W, DY, DX, DW, DB) of a layer whose input X
 * was produced by the element-wise run blk (pre and/or post mask-multiply stage, no pool, KS = 1) followed by that run's backward
 * (t4k_poolblock_bwd(DX, XRUN, blk): the post stage's input buffer = DX * post_mask, XRUN = that * pre_mask; _bactivate
 * backprop.cu:256-263).  TGT != NULL: DY -= TGT first (backprop's start, backprop.cu:43-53), the difference also stored in DY2 when
 * that is not NULL.  DX may alias X (the reference's in-place convention). */
/* Classifier-head backward + the backward of the linear layer in front of it in ONE launch: the two calls
 *   t4k_loss_linear_bwd(X2, W2, P, TGT, Y2, X2, MASK, Y1, DW2, DB2, N, E0b, E0a, 1);   t4k_linear_bwd(X1, W1, Y1, X1, DW1, DB1, N, E0a, E1, 1)
 * (backprop.cu:103-121, 226-254: out -= target, dW2 | dB2, dX2 in place, the mask multiply of the layer between them -> Y1, dW1 | dB1, dX1 over X1),
 * training passes only.  t4k_mlp_head_bwd_ok() says whether the shapes qualify. */
int t4k_mlp_head_bwd_ok(int N, int E1, int E0a, int E0b);
int t4k_mlp_head_bwd(float *X2, const float *W2, float *P, const float *TGT, float *Y2, const float *MASK, float *Y1, float *DW2, float *DB2,
                     float *X1, const float *W1, float *DW1, float *DB1, int N, int E1, int E0a, int E0b, t4k_stream_t s);
/* t4k_mlp_head_bwd with element-wise RUNS of one or two mask-multiply layers (t4k_poolblock, no pool / flatten) instead of the single mask layer:
 * run2 in front of the head layer (XRUN2 = the run's input tensor = the big layer's output), run1 in front of the big layer (NULL: none).
 * train = 0: a frozen net, dX only (gradient tensors may be NULL). */
int t4k_mlp_block_bwd(float *X2, const float *W2, float *P, const float *TGT, float *Y2, const t4k_poolblock *run2, float *XRUN2, float *DW2, float *DB2,
                      float *X1, const float *W1, const t4k_poolblock *run1, float *XRUN1, float *DW1, float *DB1, int N, int E1, int E0a, int E0b, int train, t4k_stream_t s);
int t4k_linear_block_bwd(const float *X, const float *W, float *DY, const float *TGT, float *DY2, float *DX, const t4k_poolblock *blk, float *XRUN,
                         float *DW, float *DB, int N, int E0, int E1, int train, t4k_stream_t s);
/* classifier head in one call: [linear E1 -> H + element-wise layer] + [linear H -> E2 (+ softmax when P2 != NULL)] =
 * t4k_linear_act_fwd(X, W1, B1, Y1, layer, alpha, F1, A1) then t4k_linear_softmax_fwd / t4k_linear_fwd on A1, with every
 * tensor written; when the first GEMM is split along K the second layer's launch folds the slabs itself (one launch fewer) */
int t4k_mlp_head_fwd(const float *X, const float *W1, const float *B1, float *Y1, int layer, float alpha, float *F1, float *A1,
                     const float *W2, const float *B2, float *Y2, float *P2, int N, int H, int E1, int E2, t4k_stream_t s);
/* linear layer followed by a softmax layer (_flinear + _fsoftmax forward.cu:157-198, 229-243): Y as above,
 * P[N,E0] = row softmax of Y; both tensors are written */
int t4k_linear_softmax_fwd(const float *X, const float *W, const float *B, float *Y, float *P,
                           int N, int E0, int E1, t4k_stream_t s);
/* Model::_blinear backprop.cu:193-254: DB += sum dY; DW += dY^T X (if train); DX = dY @ W.
 * DX may alias X's buffer only when the caller guarantees X is no longer needed: the
 * kernels read X for DW before DX is written (stream order).
 * DX == NULL computes dW|dB only, DW == DB == NULL computes dX only. */
int t4k_linear_bwd(const float *X, const float *W, const float *DY, float *DX,
                   float *DW, float *DB, int N, int E0, int E1, int train, t4k_stream_t s);
/* same, plus the backward of a mask-multiply layer (dropout, relu, ...) that sits in front of this linear layer:
 * DXM = DX (*) MASK (`in = out * mask`, _bactivate backprop.cu:256-263); MASK == DXM == NULL behaves as t4k_linear_bwd */
int t4k_linear_bwd2(const float *X, const float *W, const float *DY, float *DX, const float *MASK, float *DXM,
                    float *DW, float *DB, int N, int E0, int E1, int train, t4k_stream_t s);
/* loss-side start of backprop folded into the last linear layer's backward (Model::backprop prep `out -= target`,
 * backprop.cu:60-75, + the pass-through `in = out` of a softmax / sigmoid / log-softmax output layer, :122-131, + _blinear):
 * OUT -= TGT in place, OUT2 = OUT (may be NULL), then exactly t4k_linear_bwd2 with DY = OUT.  One launch when the head is
 * small and DX aliases X; otherwise the separate launches. */
int t4k_loss_linear_bwd(const float *X, const float *W, float *OUT, const float *TGT, float *OUT2, float *DX,
                        const float *MASK, float *DXM, float *DW, float *DB, int N, int E0, int E1, int train, t4k_stream_t s);
/* multi-tensor optimizer step over a parameter table (one launch for all layers).
 * tab_dev: array of n_tensors records {G, DG, M, V, n, Nw} on the device. */
typedef struct { float *G, *DG, *M, *V; long n; int Nw; int pad; } t4k_param_rec;
int t4k_opt_multi(int kind /*0 sgd,1 adam,2 adamw*/, const t4k_param_rec *tab_dev, int n_tensors,
                  long max_n, float lr, float b1, float b2, float wd, t4k_stream_t s);

/* the same step with the launch sized to the parameters (k_opt_multi starts max_n/256 workgroups for EVERY tensor: 8192 waves for the
 * LeNet net's 101 030 parameters).  The caller fills each record's `pad` with the tensor's first 1024-element chunk - the running sum of
 * ceil(n / 1024) over the records before it - and passes the total; one workgroup per chunk. */
int t4k_opt_chunked(int kind, const t4k_param_rec *tab_dev, int n_tensors, int n_chunks,
                    float lr, float b1, float b2, float wd, t4k_stream_t s);
/* Model::sgd / adam / adamw as the host calls them (gradient.cu:63-169): t4k_opt_chunked, plus the work a backward DEFERRED to the
 * optimizer inside the same launch - t4k_conv_stack_bwd(train | 4) leaves its per-workgroup dF | dB partial rows unfolded, and when
 * t4k_opt_step is the next entry point the fold rides in the update (one launch instead of two, bit-identical).  Any OTHER entry point
 * called in between runs the stand-alone fold first, so a caller that reads the gradient tensors, accumulates a second backward or
 * all-reduces the slab before the optimizer sees what the undeferred path leaves.  tab_host: the caller's host copy of tab_dev. */
int t4k_opt_step(int kind, const t4k_param_rec *tab_dev, const t4k_param_rec *tab_host, int n_tensors, int n_chunks,
                 float lr, float b1, float b2, float wd, t4k_stream_t s);
/* the data-parallel form: with the one-shot exchange connected (t4k_xchg_connect, world > 1) every gradient element is summed over all
 * ranks inside the same launch (slab / slab_n: the model's gradient slab, holding every DG of the table); otherwise t4k_opt_step. */
/* One-shot request: the NEXT t4k_opt_chunked / t4k_opt_step / t4k_opt_step_dp on stream order also stores the PRE-update values of the
 * parameter tensor G (a record's G pointer) into G_PREV - the copy rides in the update launch.  Model::backprop defers dX of a first
 * linear layer (`in = dX`, backprop.cu:240, is read by no training loop) and needs the weights of that backward should a word ask for
 * it after the step.  G == NULL cancels a pending request. */
int t4k_opt_snapshot(const float *G, float *G_PREV);
int t4k_opt_snapshot_pending(void);              /* 1 while a request has not been consumed by an optimizer launch (a host checks it after ITS launch) */
int t4k_opt_step_dp(int kind, const t4k_param_rec *tab_dev, const t4k_param_rec *tab_host, int n_tensors, int n_chunks,
                    float lr, float b1, float b2, float wd, float *slab, long slab_n, t4k_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* T4K_H_ */
