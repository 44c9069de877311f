// t4k_bind.cpp - the file a tensorForth maintainer adds to run the reference's host VM on libt4hip.so (MI355X).
//
// The reference's host half (src/vm, src/io, src/ld, src/nn/model.cpp, src/nn/loss.cpp ... plain g++) reaches its device half
// through ordinary member functions: Tensor::*, Model::_f*/_b*/sgd/adam, MMU::tensor/free, Dataset::_load,
// t4_rand*.  This file re-implements that seam as thin forwards to the C-ABI of include/t4k.h; the .cu files that used to
// define these symbols (src/mu/tensor.cu, src/mu/mmu.cu, src/mu/dataset.cu, src/nn/forward.cu, backprop.cu, gradient.cu,
// src/util.cu, src/t4math.cu, src/nn/nmath.*) are dropped from the build.  Each function cites the definition it replaces.
//
// It is compiled against the reference's own, unmodified headers:
//     g++ -std=c++17 -fsyntax-only -I<reference>/src -I<this repo>/include integration/t4k_bind.cpp
// (tests/test_integration_bind.py does exactly that when the reference tree is present).  Nothing here is part of the
// product library; the shipped host (tensorforth_amd/host) is the same binding written out against its own object store.
//
// Memory: the reference allocates tensors from one cudaMallocManaged arena and lets host code dereference them.  Here tensor
// DATA lives in HBM (t4k_malloc) and every host read/write the reference makes goes through Tensor::d2h / t4k_memcpy_*; the
// 160-byte object headers stay in host memory (MMU's Mpool), exactly the split tensorforth_amd/host/tensor.cpp uses.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>
#include "t4k.h"
#include "ten4_config.h"
#include "sys.h"                    // t4base.h, util.h, io/aio.h, mu/mmu.h (tensor.h, dataset.h)
#include "nn/model.h"

#if !(T4_DO_OBJ && T4_DO_NN)
#error "the binding covers the tensor + nn build of the reference (T4_DO_OBJ && T4_DO_NN)"
#endif

namespace {
// one small device scratch replaces the per-tensor `_tmp` slot the reference keeps at data[numel] (tensor.cu:481)
float *scratch() {
    static float *p = nullptr;
    if (!p) { void *q = nullptr; if (t4k_malloc(&q, 256) == T4K_OK) p = (float *)q; }
    return p;
}
float read_scalar() {                                    // observably synchronous, like the reference's D2H after a reduction
    float v = 0.0f;
    t4k_memcpy_d2h(&v, scratch(), sizeof(float), nullptr); t4k_sync(nullptr);
    return v;
}
int read_int(const void *d) { int v = 0; t4k_memcpy_d2h(&v, d, sizeof(int), nullptr); t4k_sync(nullptr); return v; }
void warn(int rc, const char *what) { if (rc != T4K_OK) ERROR("%s failed: %s\n", what, t4k_last_error()); }   // print-and-continue (ten4_types.h:25)
}

// ===================================================================================================== util.cu:28-70
namespace t4 {
extern "C" {
void t4_rand_init(long seed) { warn(t4k_rand_init((uint64_t)seed), "t4_rand_init"); }
void t4_rand(float *d, long sz, rand_opt opt, float bias, float scale) { warn(t4k_rand(d, sz, (int)opt, bias, scale, nullptr), "t4_rand"); }
}
}

namespace t4::mu {
// ===================================================================================================== mu/tensor.cu
// :16-23 / :28-55  element-wise (math_op values are identical to t4k's, so they pass through uncast)
Tensor &Tensor::ten_op(math_op op, Tensor &A, DU v, Tensor &O) {
    warn(t4k_ts_op((int)op, A.data, v, O.data, (long)A.numel, nullptr), "tensor#ten_op");
    return O;
}
Tensor &Tensor::ten_op(math_op op, Tensor &A, Tensor &B, Tensor &O) {
    const U32 Na = A.N(), Nb = B.N(), N = std::max(Na, Nb);
    if (A.HWC() != B.HWC() || (Na == 1 ? B.numel : A.numel) != O.numel) {
        ERROR("  tensor#ten_op A.HWC(%ld)!=B.HWC(%ld) or N, C diff\n", (long)A.HWC(), (long)B.HWC());
        return O;
    }
    if ((Na == 1 || Nb == 1) && Na != Nb) {              // broadcast over the batch
        for (U32 n = 0; n < N; n++)
            warn(t4k_tt_op((int)op, A.slice(Na == 1 ? 0 : n), B.slice(Nb == 1 ? 0 : n), O.slice(n), (long)A.HWC(), nullptr), "tensor#ten_op");
    } else warn(t4k_tt_op((int)op, A.data, B.data, O.data, (long)A.numel, nullptr), "tensor#ten_op");
    return O;
}
// :62-72  batched dot
Tensor &Tensor::dot(Tensor &A, Tensor &B, Tensor &O, DU alpha, DU beta) {
    const U32 K = A.W(), C = A.C(), Na = A.N(), Nb = B.N(), N = std::max(Na, Nb);
    for (U32 n = 0; n < N; n++)
        warn(t4k_dot(A.slice(Na == 1 ? 0 : n), B.slice(Nb == 1 ? 0 : n), O.slice(n), alpha, beta, (int)K, (int)C, nullptr), "tensor#dot");
    return O;
}
// :73-77  words `@` / `matmul`
Tensor &Tensor::mm(Tensor &A, Tensor &B, Tensor &O, bool inc, bool tA, bool tB) { return gemm3(A, B, O, DU1, inc ? DU1 : DU0, tA, tB); }
// :79-87  the nn layers' GEMM (one sample, C = 1)
Tensor &Tensor::linear(Tensor &A, Tensor &B, Tensor &O, int H, int W, int K, DU alpha, DU beta, bool tA, bool tB) {
    warn(t4k_gemm(A.data, B.data, O.data, alpha, beta, tA, tB, H, W, K, 1, nullptr), "tensor#linear");
    return O;
}
// :97-123  word `gemm` was the host's own blocked loop over managed memory: here the same product on the GPU
Tensor &Tensor::gemm(Tensor &A, Tensor &B, Tensor &O, DU alpha, DU beta, bool tA, bool tB) { return gemm3(A, B, O, alpha, beta, tA, tB); }
namespace {
// :124-216  gemm1/2 (double accumulator, tA/tB ignored) and gemm3/4 (tiled) share the batching loop
Tensor &gemm_any(bool f64acc, Tensor &A, Tensor &B, Tensor &O, DU alpha, DU beta, bool tA, bool tB, const char *nm) {
    const U32 H = (tA && !f64acc) ? A.W() : A.H(), W = (tB && !f64acc) ? B.H() : B.W();
    const U32 Ka = (tA && !f64acc) ? A.H() : A.W(), Kb = (tB && !f64acc) ? B.W() : B.H();
    const U32 Na = A.N(), Nb = B.N(), C = B.C(), N = std::max(Na, Nb);
    if (Ka != Kb || N != O.N() || C != O.C()) { ERROR("  tensor#%s ka(%d)!=kb(%d) or N, C diff\n", nm, Ka, Kb); return O; }
    for (U32 n = 0; n < N; n++) {
        DU *da = A.slice(Na == 1 ? 0 : n), *db = B.slice(Nb == 1 ? 0 : n);
        int rc = f64acc ? t4k_gemm_f64acc(da, db, O.slice(n), alpha, beta, (int)H, (int)W, (int)Ka, (int)C, nullptr)
                        : t4k_gemm(da, db, O.slice(n), alpha, beta, tA, tB, (int)H, (int)W, (int)Ka, (int)C, nullptr);
        warn(rc, nm);
    }
    return O;
}
}
Tensor &Tensor::gemm1(Tensor &A, Tensor &B, Tensor &O, DU alpha, DU beta, bool tA, bool tB) { return gemm_any(true,  A, B, O, alpha, beta, tA, tB, "gemm1"); }
Tensor &Tensor::gemm2(Tensor &A, Tensor &B, Tensor &O, DU alpha, DU beta, bool tA, bool tB) { return gemm_any(true,  A, B, O, alpha, beta, tA, tB, "gemm2"); }
Tensor &Tensor::gemm3(Tensor &A, Tensor &B, Tensor &O, DU alpha, DU beta, bool tA, bool tB) { return gemm_any(false, A, B, O, alpha, beta, tA, tB, "gemm3"); }
Tensor &Tensor::gemm4(Tensor &A, Tensor &B, Tensor &O, DU alpha, DU beta, bool tA, bool tB) { return gemm_any(false, A, B, O, alpha, beta, tA, tB, "gemm4"); }
// :198-214
Tensor &Tensor::copy(Tensor &A, Tensor &O) { warn(t4k_copy(A.data, O.data, (long)A.numel, nullptr), "tensor#copy"); return O; }
Tensor &Tensor::transpose(Tensor &A, Tensor &T) {
    for (U32 n = 0; n < A.N(); n++) warn(t4k_transpose(A.slice(n), T.slice(n), (int)A.H(), (int)A.W(), (int)A.C(), nullptr), "tensor#transpose");
    return T;
}
// :224-277  reductions: the result comes back through host memory (the word stays observably synchronous)
DU Tensor::sum()  { warn(t4k_reduce(T4K_RED_SUM, data, (long)numel, 0.0f, scratch(), nullptr), "tensor#sum"); DU v = read_scalar(); return SCALAR(v); }
DU Tensor::avg()  { DU v = sum() / numel; return SCALAR(v); }
DU Tensor::std()  { const DU mx = avg(); warn(t4k_reduce(T4K_RED_NVAR, data, (long)numel, mx, scratch(), nullptr), "tensor#std");
                    DU v = read_scalar(); v = numel ? sqrtf(v) / numel : DU0; return SCALAR(v); }
DU Tensor::norm() { warn(t4k_reduce(T4K_RED_NVAR, data, (long)numel, 0.0f, scratch(), nullptr), "tensor#norm"); DU v = sqrtf(read_scalar()); return SCALAR(v); }
DU Tensor::max()  { warn(t4k_reduce(T4K_RED_MAX, data, (long)numel, 0.0f, scratch(), nullptr), "tensor#max"); DU v = read_scalar(); return SCALAR(v); }
DU Tensor::min()  { warn(t4k_reduce(T4K_RED_MIN, data, (long)numel, 0.0f, scratch(), nullptr), "tensor#min"); DU v = read_scalar(); return SCALAR(v); }
DU Tensor::dot(Tensor &B) {
    if (rank == 1 && B.rank == 1 && numel == B.numel) warn(t4k_dot(data, B.data, scratch(), DU1, DU0, (int)numel, 1, nullptr), "tensor#dot");
    else ERROR("A.dot(B) dim? %ld != %ld)\n", (long)numel, (long)B.numel);
    DU v = read_scalar(); return SCALAR(v);
}
// :288-325
DU Tensor::loss(t4_loss op, Tensor &tgt) {
    DU z = DU0;
    switch (op) {
    case LOSS_MSE: ten_op(SUB, *this, tgt, *this); ten_op(MUL, *this, *this, *this); z = sum(); break;
    case LOSS_BCE: warn(t4k_bce(tgt.data, data, (long)numel, scratch(), nullptr), "tensor#bce"); z = -read_scalar(); break;
    case LOSS_CE:  map(LN);                              /* fall through */
    case LOSS_NLL: ten_op(MUL, *this, tgt, *this); z = -sum(); break;
    default: ERROR("Model#loss op=%d not supported!\n", op);
    }
    z /= N();
    return SCALAR(z);
}
U32 Tensor::has_nan() { int *c = (int *)(scratch() + 8); warn(t4k_nan_inf(data, (long)numel, c, nullptr), "tensor#has_nan"); return (U32)read_int(c); }
// :344-429  linear algebra: the reference's host loops over pivots run inside one launch each
Tensor &Tensor::inverse(Tensor &A, Tensor &I) {          // :344-369 (status word = failing column + 1, 0 = fine)
    if (!A.is_square("A") || !I.is_square("I")) return A;
    const int K = A.W();
    INFO("  tensor#inverse [%d,%d]\n", K, K);
    int *st = (int *)(scratch() + 8);
    warn(t4k_inverse(A.data, I.data, K, st, nullptr), "tensor#inverse");
    if (int z = read_int(st)) { ERROR("  tensor#inverse: singular matrix at column %d\n", z - 1); return A; }
    return I;
}
Tensor &Tensor::plu(Tensor &A, Tensor &I, int *d_piv) {  // :371-398; A == I: no permutation matrix is produced (Tensor::det)
    if (!A.is_square("A")) return A;
    int *st = (int *)(scratch() + 8);
    warn(t4k_plu(A.data, (&A == &I) ? nullptr : I.data, d_piv, (int)A.W(), st, nullptr), "tensor#plu");
    if (int z = read_int(st)) { ERROR("  tensor#plu: singular at column %d\n", z - 1); return A; }
    return I;
}
Tensor &Tensor::lu_inverse(Tensor &A, Tensor &I, int *d_piv) {   // :400-417
    if (!A.is_square("A") || !I.is_square("I")) return I;
    const int K = A.W();
    INFO("  tensor#lu_inverse [%d,%d]\n", K, K);
    int *st = (int *)(scratch() + 8);
    warn(t4k_lu_inverse(A.data, I.data, d_piv, K, st, nullptr), "tensor#lu_inverse");
    if (int z = read_int(st)) ERROR("  tensor#plu: singular at column %d\n", z - 1);
    return I;
}
Tensor &Tensor::lu(Tensor &LU, bool get_u) { if (!LU.is_square("LU")) return LU; warn(t4k_lu_extract(LU.data, get_u, (int)LU.H(), nullptr), "tensor#lu"); return LU; }
DU Tensor::det() {                                       // :431-456: in-place P.L.U, sign of the permutation x sign of the pivots x exp(sum ln|u_jj|)
    const int K = (int)H();
    int *d_piv = nullptr; warn(t4k_malloc((void **)&d_piv, sizeof(int) * K), "tensor#det");
    plu(*this, *this, d_piv);
    std::vector<int> piv(K);
    t4k_memcpy_d2h(piv.data(), d_piv, sizeof(int) * K, nullptr); t4k_sync(nullptr);
    int cnt = 0; for (int i = 0; i < K; i++) if (piv[i] != i) cnt++;
    const int sign = (cnt % 2 == 0) ? 1 : -1;
    float *ld = scratch() + 1; int *sg = (int *)(scratch() + 9);
    warn(t4k_logdet(data, K, ld, sg, nullptr), "tensor#det");
    float l = 0.0f; t4k_memcpy_d2h(&l, ld, sizeof(float), nullptr); t4k_sync(nullptr);
    const int dsign = read_int(sg);
    t4k_free(d_piv);
    DU v = expf(l) * sign * dsign;
    return SCALAR(v);
}
Tensor &Tensor::triu() { warn(t4k_lu_extract(data, 1, (int)H(), nullptr), "tensor#triu"); return *this; }
Tensor &Tensor::tril() { warn(t4k_lu_extract(data, 0, (int)H(), nullptr), "tensor#tril"); return *this; }
// :540-575  fills / maps
Tensor &Tensor::identity() { for (U32 n = 0; n < N(); n++) warn(t4k_identity(slice(n), (int)H(), (int)W(), (int)C(), nullptr), "tensor#identity"); return *this; }
Tensor &Tensor::zeros()    { warn(t4k_memset(data, 0, sizeof(DU) * numel, nullptr), "tensor#zeros"); return *this; }
Tensor &Tensor::map(math_op op, DU v) { warn(t4k_math((int)op, data, v, (long)numel, nullptr), "tensor#map"); return *this; }
Tensor &Tensor::normalize(DU avg, DU std) { warn(t4k_ts_op(T4K_SUB, data, avg, data, (long)numel, nullptr), "normalize"); warn(t4k_ts_op(T4K_DIV, data, std, data, (long)numel, nullptr), "normalize"); return *this; }
void Tensor::d2h(DU *h, DU *d, int bsz) { t4k_memcpy_d2h(h, d, (size_t)bsz, nullptr); t4k_sync(nullptr); }
// ---- the tensor debugger (tensor.cu:587-684: _dump, _view, show) - what `1 trace` / `2 trace` print of the layer tensors.  The reference reads managed
// memory through the device pointer; here the page is copied to the host first (d2h), everything else is the reference's text.
void Tensor::_dump(DU *dv, U32 H, U32 W, U32 C) {
    std::vector<DU> hb((size_t)H * W * C); d2h(hb.data(), dv, (int)(sizeof(DU) * hb.size())); DU *v = hb.data();
    const DU  hw = I2D(H) * W, sr = sqrtf(hw);
    const U32 sh = UINT(hw / sr) + ((hw - sr*sr) > DU0 ? 1 : 0);
    const U32 h  = W > 1 ? H : (hw < 36.0 ? 1 : sh);
    const U32 w  = W > 1 ? W : (hw < 36.0 ? H : UINT(sr));
    std::vector<DU> csum(C, DU0);
    for (U32 i = 0; i < h; i++) {
        INFO("\n");
        DU sum = DU0;
        for (U32 k = 0; k < C; k++) {
            for (U32 j = 0; j < w; j++) {
                U64 n = j + i * w;
                if (n >= hw) { INFO(" ...."); continue; }
                DU  r = v[k + n * C];
                INFO("%5.2f", r);
                sum += r;
                csum[k] += r;
            }
            INFO("|");
        }
        INFO("Σ=%6.3f", sum);
    }
    if (h > 1) {
        INFO("\nΣΣ=");
        for (U32 k = 0; k < C; k++) INFO("%6.3f ", csum[k]);
    }
}
void Tensor::_view(DU *dv, U32 H, U32 W, U32 C, DU mean, DU scale) {
    std::vector<DU> hb((size_t)H * W * C); d2h(hb.data(), dv, (int)(sizeof(DU) * hb.size())); DU *v = hb.data();
    auto map = [](DU x) {
        static const char *lk = " `.-:;!+*ixekO#@";     /// 16 shades
        static const int   sz = 16;
        int i = (int)((x + 1.0) * sz/2);
        return lk[i < 0 ? 0 : (i < sz ? i : sz-1)];
    };
    const U64 hw = H * W, sr = (U64)sqrtf(hw);
    const U32 sh = (hw / sr) + ((hw - sr*sr) > 0L ? 1 : 0);
    const U32 w  = W > 1 ? W : (hw < 36L ? H : sr);
    const U32 h  = W > 1 ? H : (hw < 36L ? 1 : sh);
    std::vector<DU> csum(C, DU0);
    for (U32 i = 0; i < h; i++) {
        INFO("\n");
        for (U32 k = 0; k < C; k++) {
            for (U32 j = 0; j < w; j++) {
                U64 n = j + i * w;
                if (n >= hw) { INFO("  "); continue; }
                DU r0 = v[k + (j>0 ? n - 1 : n) * C];
                DU r1 = v[k + n * C];
                DU x0 = (r0 - mean) * scale;
                DU x1 = (((r0 + r1) * 0.5) - mean) * scale;
                INFO("%c%c", map(x0), map(x1));  /// double width
                csum[k] += r1;
            }
            INFO("|");
        }
    }
    if (h > 1) {
        INFO("\nΣΣ=");
        for (U32 k = 0; k < C; k++) INFO("%6.3f ", csum[k]);
    }
    INFO("\n");
}
int Tensor::show(bool dump) {
    const U32 N  = this->N(), H = this->H(), W = this->W(), C = this->C();
    const U64 hw = (U64)H * W;
    DU mean  = avg();
    DU scale = 0.5 / std();            /// P=95%
    for (U32 n = 0; n < N; n++) {
        DU *d = slice(n);
        if (dump || hw < 100) {
            INFO("\nn=%d", n);
            _dump(d, H, W, C);
        }
        if (hw > 36L) _view(d, H, W, C, mean, scale);
    }
    INFO("\n");
    return 0;
}

// ===================================================================================================== mu/mmu.cu:208-262
// headers from the host-side object pool, data from HBM.  (talloc / mark_free / sweep / obj2du keep their reference bodies.)
Tensor &MMU::tensor(U64 sz) {
    Tensor &t = talloc(sz);                              // header + bookkeeping as in mmu.cu:198-206 ...
    void *d = nullptr; warn(t4k_malloc(&d, sizeof(DU) * sz), "mmu#tensor");
    t.reset(d, sz);                                      // ... but the data block is device memory
    return t;
}
Tensor &MMU::tensor(U32 h, U32 w) { Tensor &t = tensor((U64)h * w); t.reshape(h, w); return t; }
Tensor &MMU::tensor(U32 n, U32 h, U32 w, U32 c) { Tensor &t = tensor((U64)n * h * w * c); t.reshape(n, h, w, c); return t; }
void MMU::free(Tensor &t) {                              // mmu.cu:246-268: data block, the layer's parameter / moment / mask tensors, then the header
    t4k_sync(nullptr);
    if (t.data) warn(t4k_free(t.data), "mmu#free");
    t.data = nullptr;
    if (t.grad_fn != L_NONE) {
        for (int i = 0; i < 4 && t.mtum[i]; i++) { if (t.mtum[i] == t.grad[i]) continue; free(*t.mtum[i]); }   // SGD's placeholders alias w, b
        if (t.mtum[4]) free(*t.mtum[4]);
        for (int i = 0; i < 4 && t.grad[i]; i++) free(*t.grad[i]);
        if (t.grad[4]) free(*t.grad[4]);
    }
    _mpool.free(&t);
}
Tensor &MMU::copy(Tensor &t0) {                          // mmu.cu:273-297: attributes copied, not a layer any more, own data block
    if (!t0.is_tensor()) return t0;
    Tensor *t1 = (Tensor *)_mpool.malloc();
    memcpy((void *)t1, (void *)&t0, sizeof(Tensor));
    for (int i = 0; i < 5; i++) t1->grad[i] = t1->mtum[i] = NULL;
    t1->grad_fn = L_NONE; t1->nref = 1;
    void *d = nullptr; warn(t4k_malloc(&d, sizeof(DU) * t0.numel), "mmu#copy");
    t1->data = (DU *)d;
    warn(t4k_memcpy_d2d(t1->data, t0.data, sizeof(DU) * t0.numel, nullptr), "mmu#copy");
    return *t1;
}

// ===================================================================================================== mu/dataset.cu:123-158
void Dataset::_load(U8 *cp_data, U8 *cp_label, int n) {  // the loader's host buffers -> one H2D copy + u8 -> f32 on the GPU
    const long bytes = (long)n * (long)HWC();
    if (!data) { void *d = nullptr; warn(t4k_malloc(&d, sizeof(DU) * (numel + 1)), "dataset#_load"); data = (DU *)d; }   // :133-136: sized once the corpus is known
    if (!label) H_ALLOC(&label, N() * sizeof(U32));
    static void *stage = nullptr; static long cap = 0;
    if (bytes > cap) { if (stage) t4k_free(stage); warn(t4k_malloc(&stage, (size_t)bytes), "dataset#_load"); cap = bytes; }
    warn(t4k_memcpy_h2d(stage, cp_data, (size_t)bytes, nullptr), "dataset#_load");
    warn(t4k_u8_normalize((const uint8_t *)stage, data, bytes, _mean, _scale, nullptr), "dataset#_load");
    for (int i = 0; i < n; i++) label[i] = cp_label[i];  // labels stay on the host (dataset.h: `label data on host`)
    batch_sz = n;
}
} // namespace t4::mu

namespace t4::nn {
using mu::Tensor;
// ===================================================================================================== nn/forward.cu
int Model::_fconv(Tensor &in, Tensor &out) {             // :125-155
    Tensor &f = *in.grad[0], &b = *in.grad[1];
    int rc = t4k_conv2d_fwd(in.data, out.data, f.data, b.data, (int)out.N(), (int)in.H(), (int)in.W(), (int)in.C(),
                            (int)out.H(), (int)out.W(), (int)out.C(), (int)f.H(), in.stride[0], in.stride[2], nullptr);
    if (rc) ERROR("nn#fconv %s\n", t4k_last_error());    // for an unsupported (K,S,P) this is the reference's own message
    return rc;
}
int Model::_flinear(Tensor &in, Tensor &out) {           // :157-198 (GEMM + k_bias in one launch)
    return t4k_linear_fwd(in.data, in.grad[0]->data, in.grad[1]->data, out.data, (int)out.N(), (int)out.HWC(), (int)in.HWC(), nullptr);
}
int Model::_factivate(Tensor &in, Tensor &out, t4_layer fn) {   // :200-209 (t4_layer values are identical to t4k's)
    if (fn == L_DROPOUT) { int rc = t4k_dropout_mask(in.grad[4]->data, (long)in.numel, nullptr); if (rc) return rc; }   // RAND(mask) of _fstep :100-103
    return t4k_activate((int)fn, in.data, out.data, in.grad[4]->data, in.xparm, (long)in.numel, nullptr);
}
int Model::_fpool(Tensor &in, Tensor &out, t4_layer fn) {       // :211-227
    return t4k_pool((int)fn, in.data, out.data, (int)out.N(), (int)in.H(), (int)in.W(), (int)out.H(), (int)out.W(), (int)out.C(), in.stride[0], nullptr);
}
int Model::_fsoftmax(Tensor &in, Tensor &out) { return t4k_softmax(in.data, out.data, (int)in.N(), (int)in.HWC(), nullptr); }   // :229-243
int Model::_flogsoftmax(Tensor &in, Tensor &out) {       // :245-259: exp(x) - log10(sum exp(x)) per sample, quirk kept (SURVEY a-16); one launch
    return t4k_logsoftmax(in.data, out.data, (int)in.N(), (int)in.HWC(), nullptr);
}
int Model::_fbatchnorm(Tensor &in, Tensor &out) {        // :263-309
    return t4k_batchnorm_fwd(in.data, out.data, in.grad[4]->data, in.grad[0]->data, in.grad[1]->data, in.mtum[4]->data,
                             (int)out.N(), (int)(out.H() * out.W()), (int)out.C(), nullptr);
}
int Model::_fupsample(Tensor &in, Tensor &out) {         // :311-329: nearest - every cell broadcast to a k x k tile
    return t4k_dpool(T4K_L_USAMPLE, out.data, in.data, (int)in.N(), (int)out.H(), (int)out.W(), (int)in.H(), (int)in.W(), (int)in.C(), in.stride[0], nullptr);
}
// ===================================================================================================== nn/backprop.cu
int Model::_bconv(Tensor &in, Tensor &out) {             // :152-191 (dF|dB accumulate, flipped-filter dX, then `in = dx`)
    Tensor &f = *in.grad[0], &df = *in.grad[2], &db = *in.grad[3], &dx = *in.grad[4];
    return t4k_conv2d_bwd2(in.data, out.data, dx.data, in.data, f.data, train ? df.data : nullptr, train ? db.data : nullptr,
                           (int)in.N(), (int)in.H(), (int)in.W(), (int)in.C(), (int)out.H(), (int)out.W(), (int)out.C(),
                           (int)f.H(), in.stride[0], in.stride[2], train, nullptr);
}
int Model::_blinear(Tensor &in, Tensor &out) {           // :193-254 (dB += sum dY, dW += dY^T X, dX = dY W lands in X's buffer)
    return t4k_linear_bwd(in.data, in.grad[0]->data, out.data, in.data, in.grad[2]->data, in.grad[3]->data,
                          (int)in.N(), (int)out.HWC(), (int)in.HWC(), train, nullptr);
}
int Model::_bactivate(Tensor &in, Tensor &out) {         // :256-263  in = out * mask
    return t4k_tt_op(T4K_MUL, out.data, in.grad[4]->data, in.data, (long)in.numel, nullptr);
}
int Model::_bpool(Tensor &in, Tensor &out, t4_layer fn) {       // :265-282
    return t4k_dpool((int)fn, in.data, out.data, (int)out.N(), (int)in.H(), (int)in.W(), (int)out.H(), (int)out.W(), (int)out.C(), in.stride[0], nullptr);
}
int Model::_bupsample(Tensor &in, Tensor &out, t4_layer fn) {   // :284-300: gradient of nearest upsampling = sum over the tile
    (void)fn;
    int rc = t4k_pool(T4K_L_AVGPOOL, out.data, in.data, (int)in.N(), (int)out.H(), (int)out.W(), (int)in.H(), (int)in.W(), (int)in.C(), in.stride[0], nullptr);
    if (rc) return rc;
    return t4k_math(T4K_SCALE, in.data, (float)(in.stride[0] * in.stride[0]), (long)in.numel, nullptr);
}
int Model::_bbatchnorm(Tensor &in, Tensor &out) {        // :311-370
    return t4k_batchnorm_bwd(in.grad[0]->data, out.data, in.grad[4]->data, in.data, in.grad[2]->data, in.grad[3]->data,
                             in.mtum[4]->data, (int)in.N(), (int)(in.H() * in.W()), (int)in.C(), train, nullptr);
}
int Model::_check_nan(Tensor &t) { return (int)t.has_nan(); }   // nn/debug.cu:17-19

// ===================================================================================================== nn/gradient.cu:63-169
// the reference walks the layers and hands (w, dw, m, v) to a GdFunc; the three optimizers become three GdFuncs over t4k
Model &Model::sgd(DU lr, DU b) {
    DU parm[3] = { lr, _iter ? b : DU0, DU0 };           // `_iter ? b : 0` :139
    auto fn = [](DU *p, Tensor &w, Tensor &dw, Tensor &m, Tensor &) {
        warn(t4k_sgd(w.data, dw.data, m.data, (int)w.N(), p[0], p[1], (long)w.numel, nullptr), "nn#sgd");
    };
    return gradient("sgd", fabsf(b) < DU_EPS ? OPTI_SGD : OPTI_SGDM, fn, parm);
}
Model &Model::adam(DU lr, DU b1, DU b2) {
    DU parm[3] = { lr, b1, b2 };
    auto fn = [](DU *p, Tensor &w, Tensor &dw, Tensor &m, Tensor &v) {
        warn(t4k_adam(w.data, dw.data, m.data, v.data, p[0], p[1], p[2], (long)w.numel, nullptr), "nn#adam");
    };
    return gradient("adam", OPTI_ADAM, fn, parm);
}
Model &Model::adamw(DU lr, DU wd, DU b1, DU b2) {
    DU parm[4] = { lr, b1, b2, wd };
    auto fn = [](DU *p, Tensor &w, Tensor &dw, Tensor &m, Tensor &v) {
        warn(t4k_adamw(w.data, dw.data, m.data, v.data, p[0], p[1], p[2], p[3], (long)w.numel, nullptr), "nn#adamw");
    };
    return gradient("adamw", OPTI_ADAMW, fn, parm);
}

// ===================================================================================================== nn/loss.cpp:47-107
// (loss.cpp is host code that walks managed memory; with tensor data in HBM the two loops become launches)
Tensor &Model::onehot(mu::Dataset &dset) {
    Tensor &out = (*this)[-1];
    const U32 N = out.N(), E = (U32)out.HWC();
    if (!_hot) _hot = &T4(N, 1, E, 1);
    static void *lab = nullptr; static U32 cap = 0;
    if (N > cap) { if (lab) t4k_free(lab); warn(t4k_malloc(&lab, sizeof(U32) * N), "nn#onehot"); cap = N; }
    warn(t4k_memcpy_h2d(lab, dset.label, sizeof(U32) * (size_t)dset.batch_sz, nullptr), "nn#onehot");
    if ((U32)dset.batch_sz < N) _hot->zeros();
    warn(t4k_onehot((const uint32_t *)lab, _hot->data, dset.batch_sz, (int)E, nullptr), "nn#onehot");
    NLOG("\n  Model::onehot(ds) {\n");                    // loss.cpp:62-69: the text of its host loop
    if (*_trace > 1) {
        std::vector<DU> h((size_t)N * E); Tensor::d2h(h.data(), _hot->data, (int)(sizeof(DU) * h.size()));
        for (U32 n = 0; n < (U32)dset.batch_sz; n++) {
            const U32 m = dset.label[n];
            INFO("    n=%d {", n);
            for (U32 e = 0; e < E; e++) INFO("%2.0f%c", h[(size_t)n * E + e], e == m ? '*' : ' ');
            INFO("}\n");
        }
    }
    NLOG("  } Model::onehot(ds)");
    return *_hot;
}
int Model::hit(bool recalc) {
    if (!recalc || !_hot) return _hit;
    Tensor &out = (*this)[-1];
    int *c = (int *)(scratch() + 8);
    warn(t4k_hit(out.data, _hot->data, (int)out.N(), (int)out.HWC(), c, nullptr), "nn#hit");
    _hit = read_int(c);
    NLOG("\n  Model::hit {\n");                           // loss.cpp:96-104
    if (*_trace > 1) {
        const U32 N = out.N(), E = (U32)out.HWC();
        std::vector<DU> o((size_t)N * E), h((size_t)N * E);
        Tensor::d2h(o.data(), out.data, (int)(sizeof(DU) * o.size())); Tensor::d2h(h.data(), _hot->data, (int)(sizeof(DU) * h.size()));
        U32 cnt = 0;
        for (U32 n = 0; n < N; n++) {
            const DU *on = o.data() + (size_t)n * E, *hn = h.data() + (size_t)n * E;
            U32 m = 0; for (U32 e = 1; e < E; e++) if (on[e] > on[m]) m = e;
            cnt += D2I(hn[m]);
            INFO("    ");
            for (U32 e = 0; e < E; e++) INFO("%4.2f%c", on[e], EQ(hn[e], DU1) ? (e == m ? '#' : '*') : (e == m ? '<' : ' '));
            INFO(" n=%d cnt=%d\n", n, cnt);
        }
    }
    NLOG("  } Model::hit=%d", _hit);
    return _hit;
}
} // namespace t4::nn
