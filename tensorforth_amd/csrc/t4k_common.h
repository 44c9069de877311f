// t4k_common.h - shared internals of libt4hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include "../../include/t4k.h"

namespace t4k {

constexpr int WAVE   = 64;
constexpr int BLK    = 256;            // default workgroup: 4 waves, one per SIMD
constexpr int MAX_WG = 2048;           // 256 CUs x 8 resident blocks: grid-stride above this
constexpr float DU_EPS = 1.0e-6f;      // reference src/ten4_types.h:85

struct State {
    bool        ready   = false;
    int         device  = -1;
    hipStream_t stream  = nullptr;     // library default stream
    bool        own_stream = false;
    void       *ws      = nullptr;     // device workspace (replaces per-tensor _tmp)
    size_t      ws_bytes = 0;
    // Philox stream.  The HOST counter is authoritative: an eager launch gets (seed, counter) as kernel arguments and the
    // host advances - no device state read, no arrival ticket at the end of the kernel.  Only launches recorded into a
    // hipGraph read / advance the device copy d_rng ([0] counter, [1] ticket, [2] seed), so a replay draws fresh numbers;
    // the host adds the graph's total advance at every t4k_graph_launch and re-seeds the device copy when it is stale.
    uint64_t    seed = 0, rng_ctr = 0; // counter = element offset / 4
    uint64_t   *d_rng   = nullptr;
    uint64_t    d_rng_ctr = ~0ull;     // what the device copy holds (after pending stream work); ~0 = unknown
    uint64_t    cap_adv = 0;           // counters drawn by the capture in progress
    // data-parallel shard (t4k_rand_set_shard): a SAMPLE-KEYED draw (dropout masks) of nq counters takes the slice
    // [ctr + rank*nq, ctr + (rank+1)*nq) and moves the stream by world*nq - the draw the rank's samples would get inside the whole batch
    int         shard_rank = 0, shard_world = 1;
    bool        bn_sync = false;       // batchnorm statistics over all ranks (t4k_comm_sync_batchnorm; needs a communicator)
    bool        capturing = false;
    struct Lane { hipStream_t s; void *ws; } lane[8] = {};   // per-stream workspaces for t4k_stream_create()d streams
    int         n_lane  = 0;
    float      *d_zero  = nullptr;     // 4 KiB of zeros, never written: the source of LDS-DMA lanes whose operand row lies outside the tensor (conv_big.hip)
    int        *d_sync  = nullptr;     // 32768 zeroed, self re-arming ints: [0,4096) pair-mode GEMM tickets/flags, [8192,32768) per-stream arrival gates (gate_for / flags_for; ints [512,1024) of a stream's block are the epoch slots of k_gemm_dual32, [1280,2048) the per-workgroup slots of k_head_bwd_l32)
    int        *spin_err = nullptr;   // pinned, device-visible error word of the inter-workgroup waits (spin_check)
    int         cu_count = 256;
    unsigned    slot_epoch[16] = {};  // per stream lane: launch count of the kernels that tag the arrival slots (next_slot_epoch)
    unsigned long long launches = 0;  // kernels launched by the library since t4k_init (t4k_launch_count)
    bool        gates_off = false;    // no kernel that makes workgroups wait for each other is chosen any more (t4k_gates_enable(0), or set by spin_check() after a timed-out wait)
    int         pending = 0;          // work a launch left for the NEXT entry point: bit 0 = a conv stack's dF | dB partial fold.  Set, flushed and consumed under pending_mu():
                                      // whichever thread reaches an entry point first flushes (a gradient read from a second thread must see the folded dF | dB);
                                      // the one entry point a helper thread of the host calls while the model thread trains - t4k_event_wait, the dataset reader - never flushes
    char        err[512] = {0};
};
State &st();
inline bool gates_ok() { return !st().gates_off; }
// Environment switches.  The RELEASE library reads a short, documented list through env_int / getenv; every engine-selection and
// tuning knob the rounds accumulated (T4K_GEMM_*, T4K_CONV*, T4K_STACK_SPLIT ...) is a LAB switch: read only by the `make LAB=1` build
// (libt4hip_lab.so, tools/experiments/*), a compile-time constant = the measured-best default in the release build.
inline int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
#ifdef T4K_LAB
#define T4K_LAB_ENV(name, dflt) t4k::env_int(name, dflt)
#else
#define T4K_LAB_ENV(name, dflt) (dflt)
#endif
// every kernel launch of the library goes through this macro and is counted: bench.py prints the MEASURED launches per step
#define T4K_LAUNCH(kernel, ...) do { ++t4k::st().launches; hipLaunchKernelGGLInternal((kernel), __VA_ARGS__); } while (0)
// what a drawing kernel receives: eager = (base, seed) by value and state == nullptr; inside a graph = the device copy
struct RngArg { uint64_t base, seed; uint64_t *state; };
RngArg rng_draw(hipStream_t hs, uint64_t nq, bool sample_keyed = false);    // host: reserve nq counters (4 elements each) of the stream for one launch (optim.hip)

int  fail(int code, const char *fmt, ...);
int  hip_fail(hipError_t e, const char *what);
inline hipStream_t S(t4k_stream_t s) { return s ? (hipStream_t)s : st().stream; }
// workspace of the stream a kernel is launched on: work forked to a side stream must not share partial slabs
// arrival gates of the one-launch producer/consumer kernels (self re-arming ints, zero between launches): one block of 64
// ints per stream (kernels on different streams may overlap), `slot` picks 4 ints inside it
// Returns nullptr for a stream the library does not know (neither its default stream nor one made by t4k_stream_create, e.g. a caller's
// own hipStream_t): such launches could run beside a gated launch of the default stream and must not share its counters - the callers
// then take their ungated path (separate launches).  Launches on ONE stream are ordered, so a lane's gates are never used concurrently.
// Co-residency: the gated kernels spin until every workgroup of the launch has arrived, which needs all of them resident at once; the
// launchers only take the gated path for grids <= the CU count of an exclusively owned device (one process per GPU) - on a
// partitioned or shared device set T4K_GEMM_DUAL=0 / T4K_LINSMALL_GATE=0.
inline int lane_of(const void *s) {                      // 0 = default stream, i+1 = library stream i, -1 = unknown
    State &g = st();
    if (!s || s == (const void *)g.stream) return 0;
    for (int i = 0; i < g.n_lane; i++) if ((const void *)g.lane[i].s == s) return i + 1;
    return -1;
}
inline int *gate_for(const void *s, int slot) {
    const int li = lane_of(s);
    if (li < 0) return nullptr;
    return st().d_sync + 8192 + li * 2048 + slot * 4;
}
// 16 broadcast flags of the same stream, one per 64-byte line (hundreds of waiting workgroups poll these instead of the counter)
inline int *flags_for(const void *s) { int *g0 = gate_for(s, 0); return g0 ? g0 + 1024 : nullptr; }
// The arrival slots (ints [512, 1024) of a lane's gate block) are tagged with a per-lane launch EPOCH by every kernel that uses them
// (k_gemm_dual32, k_head_bwd_dual32): ONE counter per lane, shared by all of them - with a counter per kernel a stale tag left by one
// could equal the other's current epoch and let an in-place dX writer pass before the dW readers have read X.
// Never 0; the slots are cleared when the counter wraps.  Call only with a stream lane_of() knows.
inline unsigned next_slot_epoch(hipStream_t hs, unsigned *slots) {
    const int li = lane_of(hs);
    unsigned &e = st().slot_epoch[(li >= 0 && li < 15) ? li : 15];
    if (++e == 0) { (void)hipMemsetAsync(slots, 0, 1536 * sizeof(unsigned), hs); e = 1; }   // ints [512, 2048) of the block: the dual launches' slots, the broadcast flags (zero between launches anyway), k_head_bwd_l32's per-workgroup slots
    return e;
}
inline float *ws_for(const void *s) {            // accepts a t4k_stream_t or an already resolved hipStream_t
    State &g = st();
    if (s) for (int i = 0; i < g.n_lane; i++) if ((const void *)g.lane[i].s == s) return (float *)g.lane[i].ws;
    return (float *)g.ws;
}

// ---- bounded inter-workgroup waits.  The one-launch producer/consumer kernels (dual GEMMs, pair-mode GEMM, in-place head backward)
// poll an arrival gate in device memory.  Progress rests on every workgroup of the launch being resident (launchers only take the
// gated path for grids <= the CU count of an exclusively owned device) - if that ever fails (a co-scheduled kernel holds CUs, a
// partitioned device) a poll gives up after T4K_SPIN_MAX rounds (~seconds), writes a code to a pinned error word and lets the
// workgroup run on (results of THAT launch are wrong, nothing hangs); the next synchronising entry point (t4k_sync, t4k_event_sync)
// reports T4K_ERR_HIP.  The word lives per translation unit in a __device__ pointer set by t4k_init.
#define T4K_SPIN_MAX (1 << 22)
#define T4K_SPIN_DECL __device__ int *g_spin_err_dev = nullptr;
#define T4K_SPIN_WAIT(cond_not_met, code) do { int _it = 0; while (cond_not_met) { __builtin_amdgcn_s_sleep(1); \
        if (++_it > T4K_SPIN_MAX) { if (g_spin_err_dev) __hip_atomic_store(g_spin_err_dev, (code), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; } } } while (0)
void gemm_set_spin_err(int *p);          // gemm.hip
void linsmall_set_spin_err(int *p);      // linear_small.hip
inline bool gates_ok();                  // may a launcher pick a kernel whose workgroups wait for each other (arrival gates, tickets, band exchange)?
int  spin_check();                       // runtime.hip: T4K_OK, or T4K_ERR_HIP when a wait timed out since the last check (clears the word)

// Deferred work.  t4k_conv_stack_bwd(train | 4) leaves its per-workgroup dF | dB partial rows UNFOLDED: t4k_opt_step, when it is the next
// entry point, folds them inside the optimizer launch (fold + update = one launch); EVERY other entry point (they all start with
// T4K_REQUIRE_INIT) first runs the stand-alone fold on the stream of the backward, so whatever the caller does next - read a gradient
// tensor, accumulate a second backward, all-reduce the slab - sees exactly what the undeferred path would have left.
void flush_pending();                    // conv_stack.hip (takes pending_mu())
std::recursive_mutex &pending_mu();      // conv_stack.hip
#define T4K_REQUIRE_INIT_NOFLUSH() do { if (!t4k::st().ready) return t4k::fail(T4K_ERR_NODEVICE, "t4k_init not called or no gfx950 device"); } while (0)
#define T4K_REQUIRE_INIT() do { T4K_REQUIRE_INIT_NOFLUSH(); if (t4k::st().pending) t4k::flush_pending(); } while (0)
#define T4K_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) return t4k::hip_fail(_e, #call); } while (0)
#define T4K_LAUNCH_CHECK() do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) return t4k::hip_fail(_e, "kernel launch"); } while (0)

inline int grid_for(long n, int per_thread = 1) {
    long g = (n + (long)BLK * per_thread - 1) / ((long)BLK * per_thread);
    if (g > MAX_WG) g = MAX_WG;
    if (g < 1) g = 1;
    return (int)g;
}
inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// ---- device helpers ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;                                   // lane 0 holds the sum
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;                                   // every lane holds the sum
}
__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_min_all(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
// block-wide sum for BLK=256 (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float *smem4) {
    v = wave_sum_all(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) smem4[w] = v;
    __syncthreads();
    float r = smem4[0] + smem4[1] + smem4[2] + smem4[3];
    __syncthreads();
    return r;
}

// flat index -> (n, i, j) of an [N][H][W] grid.  Indices that fit 32 bits (every LeNet/CIFAR-size tensor) take the 32-bit
// divide: a 64-bit one is ~10x the instructions, and the latency-bound conv / pool kernels do several per lane.
__device__ __forceinline__ void split3(long p, int W, int H, int &j, int &i, int &n) {
    if (p < 0x7fffffffL) {
        const unsigned q = (unsigned)p, t = q / (unsigned)W, m = t / (unsigned)H;
        j = (int)(q - t * (unsigned)W); i = (int)(t - m * (unsigned)H); n = (int)m;
    } else {
        const long t = p / W, m = t / H;
        j = (int)(p - t * W); i = (int)(t - m * H); n = (int)m;
    }
}
__device__ __forceinline__ void split2(long p, int C, int &c, long &rest) {
    if (p < 0x7fffffffL) { const unsigned q = (unsigned)p, t = q / (unsigned)C; c = (int)(q - t * (unsigned)C); rest = (long)t; }
    else { const long t = p / C; c = (int)(p - t * C); rest = t; }
}

// Philox4x32-10; counter = element_index/4, key = seed (same definition as the oracle)
__device__ __forceinline__ void philox4x32_10(uint64_t ctr, uint64_t key, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(uint32_t x) {   // (0,1]
    return __fmaf_rn((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}


// One activation element, layer kind chosen at run time (same expressions as elementwise.hip act1<L>, which
// restates k_activate nmath.cu:37-70): i = input, u = uniform draw (dropout only); o = output, f = derivative mask.
__device__ __forceinline__ void act_rt(int L, float i, float u, float alpha, float &o, float &f) {
    switch (L) {
    case T4K_L_RELU:    if (i > 0.0f) { f = 1.0f; o = i; } else { f = 0.0f; o = 0.0f; } break;
    case T4K_L_TANH:    o = tanhf(i); f = 1.0f - o * o; break;
    case T4K_L_SIGMOID: o = 1.0f / (1.0f + expf(-i)); f = o * (1.0f - o); break;
    case T4K_L_SELU:    if (i > 0.0f) { f = (float)1.0507; o = i; }
                        else { f = (float)(1.7581 * (double)__expf(i)); o = (float)((double)f - 1.7581); } break;
    case T4K_L_LEAKYRL: if (i > 0.0f) { f = 1.0f; o = i; } else { f = alpha; o = alpha * i; } break;
    case T4K_L_ELU:     if (i > 0.0f) { f = 1.0f; o = i; } else { f = alpha * __expf(i); o = f - alpha; } break;
    case T4K_L_DROPOUT: if (u > alpha) { f = 1.0f; o = i; } else { f = 0.0f; o = 0.0f; } break;
    default:            o = i; f = 1.0f; break;
    }
}
// same, for heavily unrolled epilogues: the cheap mask activations stay inline, the transcendental ones (libm-sized bodies,
// 20 inlined copies made the conv+pool kernel 46 KB of code) go through one out-of-line copy
__device__ __noinline__ static float2 act_rt_slow(int L, float i, float alpha) { float o, f; act_rt(L, i, 0.f, alpha, o, f); return make_float2(o, f); }
__device__ __forceinline__ void act_rt_lean(int L, float i, float u, float alpha, float &o, float &f) {
    if (L == T4K_L_RELU)         { if (i > 0.0f) { f = 1.0f; o = i; } else { f = 0.0f; o = 0.0f; } }
    else if (L == T4K_L_DROPOUT) { if (u > alpha) { f = 1.0f; o = i; } else { f = 0.0f; o = 0.0f; } }
    else if (L == T4K_L_LEAKYRL) { if (i > 0.0f) { f = 1.0f; o = i; } else { f = alpha; o = alpha * i; } }
    else { const float2 r = act_rt_slow(L, i, alpha); o = r.x; f = r.y; }
}
// activation epilogue of a linear layer (k_splitk_fold / the fused head): F = derivative mask, A = activation output
struct ActEpi { int layer; float alpha; float *F, *A; RngArg rng; };
// a linear layer's input that does not exist yet: X[z] = act(sum_k part[k][z] + bias[z % E]) - the split-K slabs of the layer in
// front; the consumer folds them while staging its rows and writes Y (pre-activation), F and A exactly as k_splitk_fold would
struct XFold { const float *part; int nsplit; long mn; const float *bias; float *Y; ActEpi ep; };
// element `a` of a tensor whose Philox slice starts at counter `base` (units of 4 elements): the same value
// t4k_rand(uniform, bias 0, scale 1) would have stored at index a
__device__ __forceinline__ float philox_u01_at(uint64_t base, uint64_t seed, long a) {
    uint32_t r[4];
    philox4x32_10(base + (uint64_t)(a >> 2), seed, r);
    return u01(r[a & 3]);
}
// advance the device-resident stream by nq counters once every workgroup of the launch has read it
__device__ __forceinline__ void rng_advance_last_block(uint64_t *state, uint64_t base, uint64_t nq) {
    // No fences: the only ordering needed is "every workgroup has READ state[0] before it is rewritten", and each
    // workgroup's read has returned (its value was used) before the barrier below.  Relaxed agent-scope atomics keep
    // the ticket coherent across the 8 XCD L2s without the L2 write-back a release fence would trigger per workgroup.
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add((unsigned *)&state[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x * gridDim.y * gridDim.z - 1) {
            __hip_atomic_store((unsigned *)&state[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&state[0], base + nq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ void rng_begin(const RngArg &a, uint64_t &base, uint64_t &seed) {
    if (a.state) {
        base = __hip_atomic_load(&a.state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seed = __hip_atomic_load(&a.state[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else { base = a.base; seed = a.seed; }
}
// same, when `participants` workgroups of the launch each call it once (every workgroup that drew from the stream)
__device__ __forceinline__ void rng_advance_n(uint64_t *state, uint64_t base, uint64_t nq, unsigned participants) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add((unsigned *)&state[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == participants - 1) {
            __hip_atomic_store((unsigned *)&state[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&state[0], base + nq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ void rng_state_read(const uint64_t *state, uint64_t &base, uint64_t &seed) {
    base = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    seed = __hip_atomic_load(&state[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- fold of a conv stack's per-workgroup dF | dB partial rows (conv_stack.hip): dst[e] += sum over rows, in row order (deterministic).
// A 1024-thread block handles 16 elements x 64 row groups: group g adds rows g, g+64, ... (independent loads, one round trip for up to
// 256 rows), the 64 group sums are then added in order through LDS.  Used by k_cs_fold and, with the update behind it, by k_opt_step.
struct CsFoldSeg { float *dst; int off, n, start; };                  // elements [off, off+n) of a partial row -> dst[0..n); start = first global element id
struct CsFoldArgs { const float *part; long row; int nparts, nseg, total, pad_; CsFoldSeg seg[6]; };
struct CsFoldSm { float a[64][17], b[4][17]; };
// Workgroup id -> block of 16 elements.  Two neighbouring blocks share every 128-byte line of every partial row; workgroup ids go round the 8 XCDs, so
// ids w and w + 8 sit behind the SAME L2: they get the two halves of a line (the second is an L2 hit instead of a second fetch over the fabric by
// another XCD's L2 - k_opt_step read 6.8 MB for 2 MB of partial rows).  A permutation of who does what: no sum changes.
__device__ __forceinline__ int cs_fold_block(int w, int nblocks) {
    const int base = w & ~15;
    return base + 16 <= nblocks ? base + ((w & 7) << 1) + ((w >> 3) & 1) : w;
}
// returns true for the ONE thread per element that holds the sum (`v`), with `q` the segment and `k` the element inside it
__device__ __forceinline__ bool cs_fold16(const CsFoldArgs &a, int blk, CsFoldSm &sm, float &v, int &q, int &k) {
    const int el = threadIdx.x & 15, g = threadIdx.x >> 4, e = blk * 16 + el;
    int off = -1; q = 0; k = 0;
#pragma unroll
    for (int t = 0; t < 6; t++) if (t < a.nseg && e >= a.seg[t].start && e < a.seg[t].start + a.seg[t].n) { k = e - a.seg[t].start; off = a.seg[t].off + k; q = t; }
    float s = 0.f;
    if (off >= 0) {
        const float *src = a.part + off;
#pragma unroll 4
        for (int i = g; i < a.nparts; i += 64) s += src[(long)i * a.row];
    }
    sm.a[g][el] = s;
    __syncthreads();
    if (g < 4 && off >= 0) {                                        // 4 lanes per element add 16 group sums each, in order ...
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) t += sm.a[g * 16 + i][el];
        sm.b[g][el] = t;
    }
    __syncthreads();
    v = ((sm.b[0][el] + sm.b[1][el]) + sm.b[2][el]) + sm.b[3][el];   // ... and the four are added in order
    return g == 0 && off >= 0;
}
// ---- one-shot peer exchange (xchg.hip): every rank owns a window of 64-bit words {epoch, value}; a rank PUSHES its element into slot
// `rank` of every peer's window and SUMS the slots of its own window in rank order once their tags carry the call's epoch
constexpr int T4K_XCHG_MAX = 8;
struct Xchg { bool connected = false, self = false, trusted = false; int rank = 0, world = 0; long n = 0; unsigned long long *win[T4K_XCHG_MAX] = {}; unsigned epoch_slab = 0, epoch_gen = 0; };
Xchg &xchg();
struct XchgDev { unsigned long long *win[T4K_XCHG_MAX]; long per; unsigned long long patience; unsigned epoch; int rank, world; };   // windows already offset to the call's region and parity; patience in 100 MHz ticks
XchgDev xchg_begin(bool generic);                    // host: the device view of the next call (advances the region's epoch)
int xchg_allreduce(float *buf, long n, hipStream_t hs);
__device__ __forceinline__ void xchg_push(const XchgDev &x, long j, float v) {
    const unsigned long long w = ((unsigned long long)x.epoch << 32) | (unsigned long long)__float_as_uint(v);
#pragma unroll
    for (int r = 0; r < T4K_XCHG_MAX; r++)
        if (r < x.world && r != x.rank) __hip_atomic_store(x.win[r] + (long)x.rank * x.per + j, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ok = false when a peer's element never arrived within the patience: the caller must NOT use the sum (an optimizer step leaves W and dW of that element
// untouched, so a rank whose peer died keeps its last consistent state instead of diverging on a stale slot - ADVICE r4 #2); the error word says why
__device__ __forceinline__ float xchg_sum(const XchgDev &x, long j, float mine, int *err, bool &ok) {
    unsigned long long w[T4K_XCHG_MAX];
    const unsigned long long *my = x.win[x.rank] + j;
#pragma unroll
    for (int r = 0; r < T4K_XCHG_MAX; r++)            // every slot's load in flight before the first tag is looked at
        w[r] = (r < x.world && r != x.rank) ? __hip_atomic_load(my + (long)r * x.per, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
    float s = 0.f;
    ok = true;
#pragma unroll
    for (int r = 0; r < T4K_XCHG_MAX; r++) {          // rank order on every rank: the same numbers added in the same order
        if (r >= x.world) break;
        if (r == x.rank) { s += mine; continue; }
        // a peer may be late by whole seconds (its first step loads / compiles kernels): the bound is wall time (s_memrealtime, 100 MHz), not polls
        unsigned long long t0 = 0;
        for (int it = 0; (unsigned)(w[r] >> 32) != x.epoch; it++) {
            if ((it & 63) == 63) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (!t0) t0 = now;
                else if (now - t0 > x.patience) { if (err) __hip_atomic_store(err, 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); ok = false; break; }
            }
            __builtin_amdgcn_s_sleep(2);
            w[r] = __hip_atomic_load(my + (long)r * x.per, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        s += __uint_as_float((unsigned)w[r]);
    }
    return s;
}
struct PendingFold { CsFoldArgs fa; hipStream_t hs; };
PendingFold &pending_fold();             // conv_stack.hip

} // namespace t4k
