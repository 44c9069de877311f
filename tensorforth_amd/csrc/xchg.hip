// xchg.hip - ONE-SHOT gradient exchange over peer-mapped windows (MI355X-native; the reference is single-GPU, Model::sgd / adam
// src/nn/gradient.cu:63-142 is where the exchange sits; SURVEY.md section 5 "hand-written one-shot all-reduce over peer-mapped buffers ...
// all 7 links concurrently").
//
// Why not RCCL for this: the data-parallel step exchanges ONE 404 KB slab per 50 us step.  A ring / tree collective is a kernel of its own
// (launch boundary + several dependent hops), serialised between `backprop` and the optimizer.  xGMI is point to point - 7 links per GPU -
// so every rank can WRITE its slab straight into all seven peers at once and each rank adds the eight copies itself:
//   * every rank owns a receive WINDOW (device memory, shared through hipIpcGetMemHandle): [2 parities][world slots][n] 64-bit words;
//   * the optimizer launch (k_opt_step<true>, optim.hip) computes its local gradient element (folding the conv stack's partial rows on the
//     way), stores the word {epoch, value} into slot `rank` of every peer's window - one 8-byte store per element per peer, the "LL" form:
//     the tag travels WITH the value, so the reader needs no flag, no fence and no clean-up, a stale or half-arrived slot simply carries
//     the wrong tag - then polls its own window until all `world - 1` tags of the element carry this call's epoch, adds the values in RANK
//     order (every rank adds the same numbers in the same order: replicas stay bit-identical) and applies the update;
//   fold + all-reduce + SGD/Adam = ONE launch, no collective kernel, no cross-stream event.
// Two parities: call k writes parity k & 1.  A rank cannot finish call k+1 before every peer has pushed call k+1, which a peer does only
// after it has finished reading call k - so a window half is never overwritten while somebody still reads it.
// Waits are bounded in wall time (20 s, T4K_XCHG_TIMEOUT_MS): a peer that never arrives yields T4K_ERR_HIP at the next synchronising call, not a hang.
// t4k_allreduce_sum() uses the same windows (k_xchg_allreduce, a scratch region of 64 Ki elements per slot) when no RCCL communicator
// exists, so the VM's scalar reductions (nn.hit, loss words, synchronised batch-norm statistics) work over this transport as well.
#include "t4k_common.h"
#include <string.h>
#include <vector>

using namespace t4k;

namespace t4k {
Xchg &xchg() { static Xchg x; return x; }
}

namespace {

constexpr long SCRATCH = 65536;                      // elements per slot of the generic all-reduce region

struct Local {
    void *win = nullptr; size_t bytes = 0;           // this rank's window
    void *peer_map[T4K_XCHG_MAX] = {};               // what hipIpcOpenMemHandle returned (to close)
    int *d_err = nullptr;
} L;

T4K_SPIN_DECL
__global__ void k_xchg_set_err(int *p) { g_spin_err_dev = p; }

// in-place SUM of buf[0..n) over all ranks through the scratch region (n <= SCRATCH per launch)
__global__ void __launch_bounds__(256) k_xchg_allreduce(float *buf, long n, XchgDev x) {
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < n; j += (long)gridDim.x * 256) {
        const float v = buf[j];
        xchg_push(x, j, v);
        bool ok; const float sum = xchg_sum(x, j, v, g_spin_err_dev, ok);
        if (ok) buf[j] = sum;                                     // a peer that never arrived: the element keeps the local value, the call reports T4K_ERR_HIP
    }
}

} // namespace

extern "C" {

// Step 1 (every rank): allocate this rank's receive window for slabs of up to `slab_floats` elements and a `world`-rank job; returns the
// 64-byte IPC handle the launcher hands to every peer (any side channel: torch.distributed all_gather in bench.py, files in the tests).
int t4k_xchg_create(long slab_floats, int rank, int world, void *handle64) {
    T4K_REQUIRE_INIT();
    if (!handle64 || slab_floats < 1 || world < 1 || world > T4K_XCHG_MAX || rank < 0 || rank >= world) return fail(T4K_ERR_ARG, "t4k_xchg_create: bad argument");
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle larger than the 64 bytes the API hands around");
    t4k_xchg_destroy();
    Xchg &x = xchg();
    x.n = ((slab_floats + 63) & ~63L); x.rank = rank; x.world = world;
    // layout (64-bit words): slab region [2][world][n], then scratch region [2][world][SCRATCH]
    L.bytes = (size_t)2 * world * (size_t)(x.n + SCRATCH) * 8;
    // uncached / fine-grained device memory: a remote GPU's stores must be visible to this GPU's polls without a cache flush (what RCCL's
    // LL buffers use); plain hipMalloc is the fallback (enough on ONE device, where every process goes through the same L2s)
    hipError_t e = hipExtMallocWithFlags(&L.win, L.bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&L.win, L.bytes, hipDeviceMallocFinegrained); }
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&L.win, L.bytes); }
    if (e != hipSuccess) { L.win = nullptr; return hip_fail(e, "t4k_xchg_create: window allocation"); }
    T4K_HIP(hipMemset(L.win, 0, L.bytes));           // tag 0 is never an epoch
    T4K_HIP(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    T4K_HIP(hipIpcGetMemHandle(&h, L.win));
    memset(handle64, 0, 64); memcpy(handle64, &h, sizeof(h));
    x.win[rank] = (unsigned long long *)L.win;
    static bool err_set = false;
    if (!err_set && st().spin_err) { T4K_LAUNCH(k_xchg_set_err, dim3(1), dim3(1), 0, st().stream, st().spin_err); err_set = true; }
    return T4K_OK;
}
// Step 2 (every rank, after ALL ranks have made step 1): map the peers' windows.  handles = world x 64 bytes in rank order.
int t4k_xchg_connect(const void *handles) {
    T4K_REQUIRE_INIT();
    Xchg &x = xchg();
    if (!handles || !L.win || x.world < 1) return fail(T4K_ERR_ARG, "t4k_xchg_connect: t4k_xchg_create first");
    if (x.connected) return fail(T4K_ERR_ARG, "t4k_xchg_connect: already connected - t4k_xchg_create again first (fresh windows: the epochs restart at 0, old tags must not survive)");
    int ndev = 0; (void)hipGetDeviceCount(&ndev);
    for (int d = 0; d < ndev; d++) if (d != st().device) { int can = 0; if (hipDeviceCanAccessPeer(&can, st().device, d) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(d, 0); }
    (void)hipGetLastError();                         // "already enabled" is not an error
    for (int r = 0; r < x.world; r++) {
        if (r == x.rank) continue;
        hipIpcMemHandle_t h; memcpy(&h, (const char *)handles + 64 * (size_t)r, sizeof(h));
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void)hipGetLastError(); t4k_xchg_destroy(); return hip_fail(e, "t4k_xchg_connect: hipIpcOpenMemHandle (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)"); }
        L.peer_map[r] = p; x.win[r] = (unsigned long long *)p;
    }
    x.connected = true; x.epoch_slab = 0; x.epoch_gen = 0;
    st().shard_rank = x.rank; st().shard_world = x.world;      // dropout masks keyed by the sample's place in the whole batch (as t4k_comm_init)
    return T4K_OK;
}
// measurement hook: a ONE-rank job takes the exchanging optimizer launch as well (pushes to nobody, adds its own element) - what the
// machinery itself costs on a single GPU (bench.py `dp_overhead_us`)
int t4k_xchg_self(int on) { xchg().self = on != 0; return T4K_OK; }
int t4k_xchg_active(void) { const Xchg &x = xchg(); return (x.connected && (x.world > 1 || x.self)) ? 1 : 0; }
// in-place SUM of n floats over all ranks through the windows' scratch region (what t4k_allreduce_sum does when no RCCL communicator exists)
int t4k_xchg_allreduce(float *buf, long n, t4k_stream_t s) {
    T4K_REQUIRE_INIT();
    if (!xchg().connected) return fail(T4K_ERR_UNSUPPORTED, "t4k_xchg_allreduce: not connected (t4k_xchg_connect)");
    if (n <= 0) return T4K_OK;
    if (!buf) return fail(T4K_ERR_ARG, "t4k_xchg_allreduce: null");
    if (st().capturing) return fail(T4K_ERR_UNSUPPORTED, "t4k_xchg_allreduce: cannot be recorded into a graph (every call carries its own epoch)");
    return xchg().world > 1 ? xchg_allreduce(buf, n, S(s)) : T4K_OK;
}
int t4k_xchg_world(void) { return xchg().connected ? xchg().world : 0; }
int t4k_xchg_rank(void)  { return xchg().connected ? xchg().rank : 0; }
int t4k_xchg_trust(int on) { xchg().trusted = on != 0; return T4K_OK; }
int t4k_xchg_destroy(void) {
    Xchg &x = xchg();
    if (!L.win && !x.connected) return T4K_OK;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < T4K_XCHG_MAX; r++) if (L.peer_map[r]) { (void)hipIpcCloseMemHandle(L.peer_map[r]); L.peer_map[r] = nullptr; }
    if (L.win) { (void)hipFree(L.win); L.win = nullptr; }
    (void)hipGetLastError();
    if (x.connected) { st().shard_rank = 0; st().shard_world = 1; }
    x = Xchg();
    return T4K_OK;
}

} // extern "C"

namespace t4k {
// the device view of one call on the slab region (generic = false) or the scratch region (true); advances that region's epoch
XchgDev xchg_begin(bool generic) {
    Xchg &x = xchg();
    XchgDev d; memset((void *)&d, 0, sizeof(d));
    unsigned &ep = generic ? x.epoch_gen : x.epoch_slab;
    if (++ep == 0) ep = 1;                            // (a wrap after 2^32 calls meets only tags of calls 2^32 - 2 and older: never equal)
    d.epoch = ep; d.rank = x.rank; d.world = x.world;
    const long per = generic ? SCRATCH : x.n;
    const long base = (generic ? (long)2 * x.world * x.n : 0) + (long)(ep & 1) * x.world * per;
    for (int r = 0; r < x.world; r++) d.win[r] = x.win[r] + base;
    d.per = per;
    static const long ms = t4k::env_int("T4K_XCHG_TIMEOUT_MS", 20000);   // how long a rank waits for a peer's element
    // until the launcher has seen the known-sum probe succeed on every rank (t4k_xchg_trust) a wait gives up after 2 s at most: a peer that mapped the windows
    // but never runs (a partitioned / shared device, a rank that died behind the rendezvous) must cost the ladder seconds, not the first training step 20 s
    d.patience = (unsigned long long)(x.trusted ? ms : std::min(ms, 2000L)) * 100000ull;
    return d;
}
// in-place SUM over ranks of n floats on stream hs, in pieces of the scratch region
int xchg_allreduce(float *buf, long n, hipStream_t hs) {
    for (long off = 0; off < n; off += SCRATCH) {
        const long m = n - off < SCRATCH ? n - off : SCRATCH;
        const XchgDev d = xchg_begin(true);
        T4K_LAUNCH(k_xchg_allreduce, dim3((unsigned)((m + 255) / 256 < 1024 ? (m + 255) / 256 : 1024)), dim3(256), 0, hs, buf + off, m, d);
    }
    hipError_t e = hipGetLastError(); if (e != hipSuccess) return hip_fail(e, "k_xchg_allreduce");
    return T4K_OK;
}
}
