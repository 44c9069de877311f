// tensor.cpp - backend bring-up, HBM arena, object store and the Tensor methods.
// Semantics follow the reference's Tensor (src/mu/tensor.cu) and MMU (src/mu/mmu.cu) including
// their quirks (std() = sqrt(sum)/numel, SCALAR() LSB clearing, small-sum host loop); every
// device operation goes through the t4k_* C-ABI.
#include "t4.h"
#include <stdarg.h>
#include <stdlib.h>
#include <time.h>
#include <algorithm>

namespace t4 {

const char *LAYER_NAME[] = { "output ", "conv2d ", "linear ", "flatten", "relu   ", "tanh   ", "sigmoid", "selu   ",
                             "leakyrl", "elu    ", "dropout", "softmax", "logsmax", "avgpool", "maxpool", "minpool",
                             "batchnm", "upsampl", "dconv2d" };

static bool g_ready = false;
static float *g_scalar = nullptr;       // device scratch for reductions (replaces Tensor::_tmp)
static int   *g_iscalar = nullptr;

void die_if_no_backend() {
    if (g_ready) return;
    const char *dev = getenv("T4_DEVICE");
    int rc = t4k_init(dev ? atoi(dev) : 0);
    if (rc != T4K_OK) {
        fprintf(stderr, "tensorForth: GPU backend unavailable (%s): %s\n", t4k_backend_name(), t4k_last_error());
        exit(2);                         // the product path has no CPU fallback
    }
    const char *seed = getenv("T4_SEED");
    t4k_rand_init(seed ? strtoull(seed, 0, 10) : (uint64_t)time(NULL));     // reference seeds from time(), sys.cpp:37
    void *p = nullptr;
    t4k_malloc(&p, 256); g_scalar = (float *)p; g_iscalar = (int *)(g_scalar + 16);
    g_ready = true;
}
static void (*g_sink)(const char *, void *) = nullptr;
static void *g_sink_user = nullptr;
void set_host_sink(void (*fn)(const char *, void *), void *user) { g_sink = fn; g_sink_user = user; }
void get_host_sink(void (**fn)(const char *, void *), void **user) { *fn = g_sink; *user = g_sink_user; }
void hprintf(const char *fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (g_sink) g_sink(buf, g_sink_user); else fputs(buf, stdout);
}
void hputs(const std::string &text) { if (g_sink) g_sink(text.c_str(), g_sink_user); else fputs(text.c_str(), stdout); }   // text of any length (hprintf formats into 1 KiB)
int chk(int rc, const char *what) {
    if (rc != T4K_OK) hprintf("%s failed: %s\n", what, t4k_last_error());     // print-and-continue (ten4_types.h:25)
    return rc;
}
t4k_stream_t stream() { return nullptr; }

// ---------------------------------------------------------------- Arena
Arena &Arena::get() { static Arena a; return a; }
void Arena::add_slab(size_t need) {
    die_if_no_backend();
    size_t sz = (size_t)256 << 20;                       // 256 MiB slabs; 288 GB of HBM leaves room to grow
    const char *e = getenv("T4_SLAB_MB"); if (e) sz = (size_t)atol(e) << 20;
    while (sz < need) sz <<= 1;
    void *p = nullptr;
    if (chk(t4k_malloc(&p, sz), "arena slab")) { fprintf(stderr, "out of HBM\n"); exit(3); }
    slabs_.push_back({(char *)p, sz});
    put_free((char *)p, sz);
}
// Free blocks are kept twice: by address (coalescing with the neighbours on free) and by size (best fit in O(log n) on alloc - the
// reference's TLSF gives O(1); a `matmul drop` loop or a training script allocates and frees a tensor per word).  No headers live in
// device memory: everything here is host-side bookkeeping over 256 MiB hipMalloc slabs.
void Arena::put_free(char *p, size_t sz) { free_[p] = sz; by_size_.insert({sz, p}); }
void Arena::take_free(std::map<char *, size_t>::iterator it) { by_size_.erase({it->second, it->first}); free_.erase(it); }
float *Arena::alloc(size_t nfloat) {
    const size_t need = ((nfloat ? nfloat : 1) * sizeof(float) + 255) & ~(size_t)255;
    for (int pass = 0; pass < 2; pass++) {
        auto fit = by_size_.lower_bound({need, nullptr});  // smallest block that fits (lowest address among equals)
        if (fit != by_size_.end()) {
            char *p = fit->second; const size_t rest = fit->first - need;
            take_free(free_.find(p));
            if (rest) put_free(p + need, rest);
            blocks_[p] = need; used_ += need; peak_ = std::max(peak_, used_);
            return (float *)p;
        }
        add_slab(need);
    }
    return nullptr;
}
void Arena::free(float *fp) {
    char *p = (char *)fp;
    auto b = blocks_.find(p);
    if (b == blocks_.end()) return;
    size_t sz = b->second; blocks_.erase(b); used_ -= sz;
    auto same_slab = [this](char *a, char *c) { for (auto &s : slabs_) if (a >= s.base && a < s.base + s.size) return c >= s.base && c < s.base + s.size; return false; };
    auto nx = free_.lower_bound(p);
    if (nx != free_.end() && p + sz == nx->first && same_slab(p, nx->first)) { sz += nx->second; auto dead = nx++; take_free(dead); }   // merge with next
    if (nx != free_.begin()) {
        auto pv = std::prev(nx);
        if (pv->first + pv->second == p && same_slab(pv->first, p)) { p = pv->first; sz += pv->second; take_free(pv); }                 // merge with previous
    }
    put_free(p, sz);
}

// ---------------------------------------------------------------- Store
Store &Store::get() { static Store s; return s; }
int Store::put(Obj *o) {
    int id;
    if (!free_ids_.empty()) { id = free_ids_.back(); free_ids_.pop_back(); objs_[id] = o; }
    else { id = (int)objs_.size(); objs_.push_back(o); }
    o->id = id; nlive_++;
    return id;
}
void Store::release(Obj *o) { objs_[o->id] = nullptr; free_ids_.push_back(o->id); nlive_--; delete o; }
Obj &Store::du2obj(DU v) {
    Obj *o = objs_[du_bits(v) >> 2];
    if (o && o->type == T_TENSOR && ((Tensor *)o)->stale_owner) ((Tensor *)o)->stale_owner->materialize_dx0();   // a lazily skipped dX: produce it before anybody looks
    return *o;
}
DU   Store::obj2du(Obj &o) { return bits_du(((uint32_t)o.id << 2) | 1u); }

Tensor &Store::tensor(uint64_t sz) {
    Tensor *t = new Tensor();
    t->type = T_TENSOR; t->numel = sz; t->rank = 1;
    t->shape[0] = (uint32_t)sz; t->shape[1] = t->shape[2] = t->shape[3] = 1;
    t->data = Arena::get().alloc(sz);
    put(t);
    return *t;
}
Tensor &Store::tensor(uint32_t h, uint32_t w) { Tensor &t = tensor((uint64_t)h * w); t.reshape(h, w); return t; }
Tensor &Store::tensor(uint32_t n, uint32_t h, uint32_t w, uint32_t c) {
    Tensor &t = tensor((uint64_t)n * h * w * c); t.reshape(n, h, w, c); return t;
}
Tensor &Store::copy(Tensor &t0) {                       // MMU::copy src/mu/mmu.cu:273-295
    Tensor &t1 = tensor(t0.numel);
    t1.rank = t0.rank; memcpy(t1.shape, t0.shape, sizeof(t0.shape)); memcpy(t1.stride, t0.stride, sizeof(t0.stride));
    t1.iparm = t0.iparm; t1.xparm = t0.xparm;
    t1 = t0;
    return t1;
}
Tensor &Store::dim(Tensor &t0) {                        // MMU::dim mmu.cu:296-302: HWCN -> NHWC
    const int map[] = {3, 0, 1, 2};
    Tensor &t = tensor(4);
    float v[4]; for (int i = 0; i < 4; i++) v[i] = (float)t0.shape[map[i]];
    t.from_host(v, 4);
    return t;
}
Tensor &Store::slice(Tensor &t0, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1) {   // mmu.cu:307-330
    if (t0.rank < 2) { hprintf("dim?"); return t0; }
    if (x1 == (uint32_t)-1) x1 = t0.W();
    if (y1 == (uint32_t)-1) y1 = t0.H();
    Tensor &t1 = t0.rank == 2 ? tensor(y1 - y0, x1 - x0) : tensor(t0.N(), y1 - y0, x1 - x0, t0.C());
    const uint32_t N = t1.N(), C = t1.C();
    const size_t bsz = sizeof(float) * C * t1.W();
    for (uint32_t n = 0; n < N; n++)
        for (uint32_t j = y0, j0 = 0; j < y1; j++, j0++)
            t4k_memcpy_d2d(t1.slice(n) + (size_t)C * j0 * t1.W(), t0.slice(n) + (size_t)C * (j * t0.W() + x0), bsz, stream());
    return t1;
}
void Store::free(Tensor &t) {                           // MMU::free mmu.cu:247-268
    if (t.owns && t.data) Arena::get().free(t.data);
    if (t.grad_fn != 0) {
        for (int i = 0; i < 4 && t.mtum[i]; i++) { if (t.mtum[i] == t.grad[i]) continue; free(*t.mtum[i]); }
        if (t.mtum[4]) free(*t.mtum[4]);
        for (int i = 0; i < 4; i++) if (t.grad[i]) free(*t.grad[i]);
        if (t.grad[4]) free(*t.grad[4]);
    }
    release(&t);
}
void Store::drop(Obj &o) {
    if (o.type == T_MODEL) { ((Model &)o).free_all(); release(&o); return; }
    if (o.type == T_DATASET) {
        Dataset &d = (Dataset &)o;
        if (d.cp) d.cp->idle();
        d.release_ring();
        d.data = nullptr; d.owns = false;
        release(&o); return;
    }
    free((Tensor &)o);
}
void Store::mark_free(DU v) { if (!IS_VIEW(v)) marked_.push_back(v); }
void Store::sweep() {
    for (DU v : marked_) { uint32_t id = du_bits(v) >> 2; if (id < objs_.size() && objs_[id]) drop(*objs_[id]); }
    marked_.clear();
}

// ---------------------------------------------------------------- Tensor
Tensor &Tensor::reshape(uint64_t sz) {
    if (sz == numel) { rank = 1; shape[0] = (uint32_t)sz; shape[1] = shape[2] = shape[3] = 1; stride[0] = stride[1] = stride[2] = stride[3] = 1; }
    else hprintf("  tensor#reshape sz != numel (%ld != %ld)\n", (long)sz, (long)numel);
    return *this;
}
Tensor &Tensor::reshape(uint32_t h, uint32_t w) {
    if ((uint64_t)h * w == numel) { rank = 2; shape[0] = h; shape[1] = w; shape[2] = shape[3] = 1; }
    else hprintf("  tensor#reshape sz != numel (%ld != %ld)\n", (long)((uint64_t)h * w), (long)numel);
    return *this;
}
Tensor &Tensor::reshape(uint32_t n, uint32_t h, uint32_t w, uint32_t c) {
    if ((uint64_t)n * h * w * c == numel) { rank = 4; shape[0] = h; shape[1] = w; shape[2] = c; shape[3] = n; }
    else hprintf("  tensor#reshape sz != numel (%ld != %ld)\n", (long)((uint64_t)n * h * w * c), (long)numel);
    return *this;
}
Tensor &Tensor::zeros() { chk(t4k_memset(data, 0, sizeof(float) * numel, stream()), "zeros"); return *this; }
Tensor &Tensor::map(int op, DU v) { chk(t4k_math(op, data, v, (long)numel, stream()), "map"); return *this; }
Tensor &Tensor::identity() { for (uint32_t n = 0; n < N(); n++) chk(t4k_identity(slice(n), H(), W(), C(), stream()), "identity"); return *this; }
Tensor &Tensor::normalize(DU avg, DU std) {             // tensor.cu:573-578
    t4k_ts_op(T4K_SUB, data, avg, data, (long)numel, stream());
    t4k_ts_op(T4K_DIV, data, std, data, (long)numel, stream());
    return *this;
}
Tensor &Tensor::operator=(Tensor &t) { chk(t4k_copy(t.data, data, (long)std::min(numel, t.numel), stream()), "copy"); return *this; }

static DU read_scalar() { DU v = 0; t4k_memcpy_d2h(&v, g_scalar, sizeof(DU), stream()); t4k_sync(stream()); return v; }

DU Tensor::sum() {                                      // tensor.cu:224-236
    chk(t4k_reduce(T4K_RED_SUM, data, (long)numel, 0, g_scalar, stream()), "sum");    // every size on the device (the reference sums short tensors on the host)
    DU v = read_scalar();
    return SCALAR(v);
}
DU Tensor::avg() { DU v = sum() / numel; return SCALAR(v); }
DU Tensor::std() {                                      // sqrt(sum (x-avg)^2) / numel   (tensor.cu:242-250)
    DU mx = avg();
    t4k_reduce(T4K_RED_NVAR, data, (long)numel, mx, g_scalar, stream());
    DU v = read_scalar(); v = numel ? sqrtf(v) / numel : 0.0f;
    return SCALAR(v);
}
DU Tensor::norm() { t4k_reduce(T4K_RED_NVAR, data, (long)numel, 0, g_scalar, stream()); DU v = sqrtf(read_scalar()); return SCALAR(v); }
DU Tensor::max()  { t4k_reduce(T4K_RED_MAX, data, (long)numel, 0, g_scalar, stream()); DU v = read_scalar(); return SCALAR(v); }
DU Tensor::min()  { t4k_reduce(T4K_RED_MIN, data, (long)numel, 0, g_scalar, stream()); DU v = read_scalar(); return SCALAR(v); }
DU Tensor::dot(Tensor &B) {
    if (rank == 1 && B.rank == 1 && numel == B.numel) t4k_dot(data, B.data, g_scalar, 1.0f, 0.0f, (int)numel, 1, stream());
    else hprintf("A.dot(B) dim? %ld != %ld)\n", (long)numel, (long)B.numel);
    DU v = read_scalar(); return SCALAR(v);
}
DU Tensor::loss(Loss op, Tensor &tgt) {                 // tensor.cu:288-325
    DU z = 0;
    switch (op) {
    case LOSS_MSE: ten_op(T4K_SUB, *this, tgt, *this); ten_op(T4K_MUL, *this, *this, *this); z = sum(); break;
    case LOSS_BCE: t4k_bce(tgt.data, data, (long)numel, g_scalar, stream()); z = -read_scalar(); break;
    case LOSS_CE:  map(T4K_LN);                         /* fall through */
    case LOSS_NLL: ten_op(T4K_MUL, *this, tgt, *this); z = -sum(); break;
    default: hprintf("Model#loss op=%d not supported!\n", op);
    }
    z /= N();
    return SCALAR(z);
}
uint32_t Tensor::has_nan() {
    t4k_nan_inf(data, (long)numel, g_iscalar, stream());
    int c = 0; t4k_memcpy_d2h(&c, g_iscalar, sizeof(int), stream()); t4k_sync(stream());
    return (uint32_t)c;
}
void Tensor::to_host(std::vector<float> &h, uint64_t n) {
    if (!n || n > numel) n = numel;
    h.resize(n);
    if (n) { t4k_memcpy_d2h(h.data(), data, n * sizeof(float), stream()); t4k_sync(stream()); }
}
void Tensor::from_host(const float *h, uint64_t n, uint64_t off) {
    if (off + n > numel) n = off < numel ? numel - off : 0;
    if (n) { t4k_memcpy_h2d(data + off, h, n * sizeof(float), stream()); t4k_sync(stream()); }
}
DU Tensor::get(uint64_t i) { DU v = 0; if (i < numel) { t4k_memcpy_d2h(&v, data + i, sizeof(DU), stream()); t4k_sync(stream()); } return v; }
void Tensor::set(uint64_t i, DU v) { if (i < numel) { t4k_memcpy_h2d(data + i, &v, sizeof(DU), stream()); t4k_sync(stream()); } }

Tensor &Tensor::ten_op(int op, Tensor &A, DU v, Tensor &O) {     // tensor.cu:16-23
    chk(t4k_ts_op(op, A.data, v, O.data, (long)A.numel, stream()), "ten_op");
    return O;
}
Tensor &Tensor::ten_op(int op, Tensor &A, Tensor &B, Tensor &O) {   // tensor.cu:28-53 (N broadcast)
    const uint32_t Na = A.N(), Nb = B.N(), N = std::max(Na, Nb);
    if (A.HWC() != B.HWC() || (Na == 1 ? B.numel : A.numel) != O.numel) {
        hprintf("  tensor#ten_op A.HWC(%ld)!=B.HWC(%ld) or N, C diff\n", (long)A.HWC(), (long)B.HWC());
        return O;
    }
    if ((Na == 1 || Nb == 1) && Na != Nb) {
        for (uint32_t n = 0; n < N; n++)
            t4k_tt_op(op, A.slice(Na == 1 ? 0 : n), B.slice(Nb == 1 ? 0 : n), O.slice(n), (long)A.HWC(), stream());
    } else chk(t4k_tt_op(op, A.data, B.data, O.data, (long)A.numel, stream()), "ten_op");
    return O;
}
Tensor &Tensor::mm(Tensor &A, Tensor &B, Tensor &O, bool inc, bool tA, bool tB) {   // Tensor::mm/gemm3 tensor.cu:73-77,161-180
    const uint32_t H = tA ? A.W() : A.H(), W = tB ? B.H() : B.W();
    const uint32_t Ka = tA ? A.H() : A.W(), Kb = tB ? B.W() : B.H();
    const uint32_t Na = A.N(), Nb = B.N(), C = B.C(), N = std::max(Na, Nb);
    if (Ka != Kb || N != O.N() || C != O.C()) { hprintf("  tensor#gemm3 ka(%d)!=kb(%d) or N, C diff\n", Ka, Kb); return O; }
    for (uint32_t n = 0; n < N; n++)
        chk(t4k_gemm(A.slice(Na == 1 ? 0 : n), B.slice(Nb == 1 ? 0 : n), O.slice(n), 1.0f, inc ? 1.0f : 0.0f, tA, tB, H, W, Ka, C, stream()), "gemm");
    return O;
}
Tensor &Tensor::gemm(int variant, Tensor &A, Tensor &B, Tensor &O, DU alpha, DU beta) {   // words gemm, gemm1..4 (tensor.cu:97-201)
    const uint32_t H = A.H(), W = B.W(), Ka = A.W(), Kb = B.H();
    const uint32_t Na = A.N(), Nb = B.N(), C = B.C(), N = std::max(Na, Nb);
    if (variant == 0) {                                  // word `gemm` (tensor.cu:97-123: a blocked loop on the HOST in the reference): the MFMA kernel, same alpha / beta
        chk(t4k_gemm(A.data, B.data, O.data, alpha, beta, 0, 0, H, W, Ka, 1, stream()), "gemm");
        return O;
    }
    if (Ka != Kb || N != O.N() || C != O.C()) { hprintf("  tensor#gemm%d ka(%d)!=kb(%d) or N, C diff\n", variant, Ka, Kb); return O; }
    for (uint32_t n = 0; n < N; n++) {
        float *da = A.slice(Na == 1 ? 0 : n), *db = B.slice(Nb == 1 ? 0 : n);
        if (variant <= 2) chk(t4k_gemm_f64acc(da, db, O.slice(n), alpha, beta, H, W, Ka, C, stream()), "gemm_f64acc");
        else              chk(t4k_gemm(da, db, O.slice(n), alpha, beta, 0, 0, H, W, Ka, C, stream()), "gemm");
    }
    return O;
}
Tensor &Tensor::transpose(Tensor &A, Tensor &T) {
    for (uint32_t n = 0; n < A.N(); n++) chk(t4k_transpose(A.slice(n), T.slice(n), A.H(), A.W(), A.C(), stream()), "transpose");
    return T;
}
static int read_status() { int s = 0; t4k_memcpy_d2h(&s, g_iscalar, sizeof(int), stream()); t4k_sync(stream()); return s; }
Tensor &Tensor::inverse(Tensor &A, Tensor &I) {          // tensor.cu:344-369
    if (A.H() != A.W() || I.H() != I.W()) { hprintf(" A: square matrix required (%d x %d)\n", A.H(), A.W()); return A; }
    const int K = A.W();
    hprintf("  tensor#inverse [%d,%d]\n", K, K);
    chk(t4k_inverse(A.data, I.data, K, g_iscalar, stream()), "inverse");
    int st = read_status();
    if (st) { hprintf("  tensor#inverse: singular matrix at column %d\n", st - 1); return A; }
    return I;
}
Tensor &Tensor::plu(Tensor &A, Tensor &I, int *piv_dev) {
    if (A.H() != A.W()) { hprintf(" A: square matrix required (%d x %d)\n", A.H(), A.W()); return A; }
    chk(t4k_plu(A.data, (&A == &I) ? nullptr : I.data, piv_dev, A.W(), g_iscalar, stream()), "plu");
    int st = read_status();
    if (st) { hprintf("  tensor#plu: singular at column %d\n", st - 1); return A; }
    return I;
}
Tensor &Tensor::lu_inverse(Tensor &A, Tensor &I) {
    if (A.H() != A.W() || I.H() != I.W()) return I;
    const int K = A.W();
    hprintf("  tensor#lu_inverse [%d,%d]\n", K, K);
    Tensor &piv = Store::get().tensor(K);
    chk(t4k_lu_inverse(A.data, I.data, (int *)piv.data, K, g_iscalar, stream()), "lu_inverse");
    int st = read_status();
    if (st) hprintf("  tensor#plu: singular at column %d\n", st - 1);
    Store::get().free(piv);
    return I;
}
Tensor &Tensor::lu(Tensor &LU, bool get_u) {
    if (LU.H() != LU.W()) return LU;
    chk(t4k_lu_extract(LU.data, get_u, LU.H(), stream()), "lu");
    return LU;
}
DU Tensor::det() {                                      // tensor.cu:431-456
    const int K = H();
    Tensor &piv = Store::get().tensor(K);
    plu(*this, *this, (int *)piv.data);
    std::vector<int> hp(K);
    t4k_memcpy_d2h(hp.data(), piv.data, sizeof(int) * K, stream()); t4k_sync(stream());
    int cnt = 0; for (int i = 0; i < K; i++) if (hp[i] != i) cnt++;
    const int sign = (cnt % 2 == 0) ? 1 : -1;
    t4k_logdet(data, K, g_scalar, g_iscalar, stream());
    DU ld = read_scalar(); int dsign = read_status();
    Store::get().free(piv);
    DU d = expf(ld) * sign * dsign;
    return SCALAR(d);
}

} // namespace t4
