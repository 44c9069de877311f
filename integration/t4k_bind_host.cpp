// t4k_bind_host.cpp - second half of the reference-side binding: the HOST orchestration the reference keeps in its .cu files.
//
// t4k_bind.cpp supplies the per-kernel seam (Tensor::*, Model::_f*/_b*, optimizers).  The reference also defines, in the same .cu
// files, plain host code that its g++-compiled half links against: the MMU object store (src/mu/mmu.cu), Tensor::reset/reshape
// (src/mu/tensor.cu:461-547), Dataset::fetch/normalize (src/mu/dataset.cu), the layer walks Model::forward / backprop / broadcast /
// gradient (src/nn/forward.cu:27-113, backprop.cu:17-140, gradient.cu:60-126) and the Code statics.  With the .cu files dropped from
// the build these must come from the binding too; this file re-states them on top of include/t4k.h (tensor DATA in HBM, object
// headers in the host Mpool).  tests/test_integration_bind.py compiles the reference's 17 host .cpp files, links them with both binding
// files against libt4hip.so with -Wl,--no-undefined, and fails when any `t4::` symbol is left undefined.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include "t4k.h"
#include "ten4_config.h"
#include "sys.h"
#include "nn/model.h"
#include "ld/loader.h"

#if !(T4_DO_OBJ && T4_DO_NN)
#error "the binding covers the tensor + nn build of the reference (T4_DO_OBJ && T4_DO_NN)"
#endif

namespace {
void warn(int rc, const char *what) { if (rc != T4K_OK) ERROR("%s failed: %s\n", what, t4k_last_error()); }
t4::mu::MMU *g_mmu = nullptr;                             // the singleton (mmu.cu:27)
}

namespace t4::mu {
// ===================================================================================================== mu/mmu.cu:21-27 statics
UFP Code::cap = 0;
UFP Code::XT0 = ~(UFP)0;
UFP Code::NM0 = ~(UFP)0;

// ===================================================================================================== mu/mmu.cu:33-88 life cycle
// Object headers come from the host Mpool exactly as in the reference; the managed-memory object store is NOT created - tensor data
// blocks are t4k_malloc'ed (HBM) by MMU::tensor / talloc below, so TLSF only keeps its (empty) bookkeeping.
MMU::MMU() : _mpool(Mpool::get_instance()), _ostore(TLSF::get_instance()) {
    _obj = (U8 *)_mpool.init(sizeof(Dataset), T4_MPOOL_SZ);
    H_ALLOC(&_mark, sizeof(DU) * T4_TFREE_SZ);
    H_ALLOC(&_dict, sizeof(Code) * T4_DICT_SZ);
    H_ALLOC(&_vmss, sizeof(DU) * T4_SS_SZ * T4_VM_COUNT);
    H_ALLOC(&_vmrs, sizeof(DU) * T4_RS_SZ * T4_VM_COUNT);
    H_ALLOC(&_pmem, T4_PMEM_SZ);
    _midx = T4_USER_AREA;
    if (t4k_init(-1) != T4K_OK && t4k_init(0) != T4K_OK) ERROR("MMU: %s\n", t4k_last_error());
}
MMU::~MMU() {
    if (_mark) std::free((void *)_mark);                  // (H_FREE would resolve to MMU::free inside a member)
    std::free(_pmem); std::free(_vmrs); std::free(_vmss);
    for (int i = 0; i < (int)_didx; i++) _dict[i].~Code();
    std::free(_dict);
    t4k_shutdown();
}
MMU *MMU::get_mmu() { if (!g_mmu) g_mmu = new MMU(); return g_mmu; }
void MMU::free_mmu() { delete g_mmu; g_mmu = nullptr; }

// ===================================================================================================== mu/mmu.cu:93-165 dictionary
void MMU::dict_validate() {
    UFP x0 = ~(UFP)0, n0 = ~(UFP)0;
    for (int i = 0; i < (int)_didx; i++) { x0 = std::min(x0, (UFP)_dict[i].xt); n0 = std::min(n0, (UFP)_dict[i].name); }
    Code::XT0 = x0; Code::NM0 = n0;
}
IU MMU::find(const char *s) {
    for (IU i = _didx; i-- > 1; ) if (STRCMP(_dict[i].name, s) == 0) return i;
    return 0;
}
void MMU::status(bool hdr) {
    if (hdr) INFO("\\ MMU.stat dict[%d/%d], pmem[%d]=%0.1f%%, tfree[%d/%d]\n", _didx, T4_DICT_SZ, _midx, 100.0 * _midx / T4_PMEM_SZ, _fidx, T4_TFREE_SZ);
    _mpool.status();
}
void MMU::dict_dump() {
    INFO("Built-in Dictionary [name0=0x%zx, xt0=0x%zx]\n", (size_t)Code::NM0, (size_t)Code::XT0);
    for (int i = 0; i < (int)_didx; i++) {
        Code &c = _dict[i];
        INFO("%4d|%03x> name=%6x, %s=%6x %s\n", i, i, c.udf ? (U32)(c.pfa - c.nlen) : (U32)((UFP)c.name - Code::NM0), c.udf ? "pf" : "xt", (U32)c.pfa_or_xtoff(), c.name);
    }
}
void MMU::colon(const char *name) {
    const int nsz = ALIGN(STRLENB(name) + 1);
    Code &c = _dict[_didx++];
    align();
    c.udf = 1; c.nlen = nsz; c.didx = _didx - 1;
    c.name = (const char *)&_pmem[_midx];
    add((U8 *)name, nsz);
    c.pfa = _midx;
}

// ===================================================================================================== mu/mmu.cu:170-262 tensors
void MMU::sweep() {
    std::unique_lock<std::mutex> lock(_mutex);
    for (int i = 0; i < (int)_fidx; i++) drop(du2obj(_mark[i]));
    _fidx = 0;
}
void MMU::drop(T4Base &t) { if (t.is_model()) free((nn::Model &)t); else free((Tensor &)t); }
void MMU::mark_free(DU v) {
    if (IS_VIEW(v)) return;
    T4Base &t = du2obj(v);
    std::unique_lock<std::mutex> lock(_mutex);
    if (_fidx < T4_TFREE_SZ) _mark[_fidx++] = obj2du(t);
    else ERROR("ERR: tfree store full, increase T4_TFREE_SZ!");
}
Tensor &MMU::talloc(U64 sz) {                             // header only: MMU::tensor (t4k_bind.cpp) attaches the HBM data block
    Tensor *t = (Tensor *)_mpool.malloc();
    t->reset(nullptr, sz);
    return *t;
}
void MMU::resize(Tensor &t, U64 sz) {
    void *d = nullptr; warn(t4k_malloc(&d, sizeof(DU) * sz), "mmu#resize");
    warn(t4k_memcpy_d2d(d, t.data, sizeof(DU) * (t.numel < sz ? t.numel : sz), nullptr), "mmu#resize");
    t4k_sync(nullptr); t4k_free(t.data);
    t.data = (DU *)d; t.H() = (U32)sz; t.numel = sz;
}
Tensor &MMU::dim(Tensor &t0) {                            // HWCN -> { N, H, W, C } as a 4-vector
    Tensor &t = tensor((U64)4);
    const DU v[4] = { (DU)t0.N(), (DU)t0.H(), (DU)t0.W(), (DU)t0.C() };
    warn(t4k_memcpy_h2d(t.data, v, sizeof(v), nullptr), "mmu#dim");
    return t;
}
Tensor &MMU::slice(Tensor &t0, U32 x0, U32 x1, U32 y0, U32 y1) {
    if (t0.rank < 2) { ERROR("dim?"); return t0; }
    if (x1 == (U32)-1) x1 = t0.W();
    if (y1 == (U32)-1) y1 = t0.H();
    Tensor &t1 = t0.rank == 2 ? tensor(y1 - y0, x1 - x0) : tensor(t0.N(), y1 - y0, x1 - x0, t0.C());
    const U32 C = t1.C(); const size_t row = sizeof(DU) * C * t1.W();
    for (U32 n = 0; n < t1.N(); n++)                      // one device-to-device copy per row of the window
        for (U32 j = y0, j0 = 0; j < y1; j++, j0++)
            warn(t4k_memcpy_d2d(t1.slice(n) + (size_t)C * j0 * t1.W(), t0.slice(n) + (size_t)C * (j * t0.W() + x0), row, nullptr), "mmu#slice");
    return t1;
}
Dataset &MMU::dataset(U32 batch_sz) {
    Dataset *ds = (Dataset *)_mpool.malloc();
    ds->init(0, T4_DATASET, 4);
    ds->N() = batch_sz; ds->batch_id = 0; ds->label = NULL;
    ds->normalize(0.0f, 256.0f);
    return *ds;
}
nn::Model &MMU::model(int &trace, U32 nsz) {
    nn::Model *m = (nn::Model *)_mpool.malloc();
    DU *t; H_ALLOC(&t, nsz * sizeof(DU));                 // the layer list lives in host memory
    m->init(this, nsz, t, &trace);
    return *m;
}
void MMU::free(nn::Model &m) {
    for (int i = (int)m.numel - 1; i >= 0; i--) free(m[i]);
    std::free(m.data);
    _mpool.free(&m);
}

// ===================================================================================================== mu/tensor.cu:461-547
Tensor &Tensor::reset(void *mem, U64 sz, t4_obj tt, t4_layer fn) {
    init(sz, tt, 1);
    const U64 GB = 1UL << 30;
    data = (DU *)mem; grad_fn = fn;
    for (int i = 0; i < 4; i++) stride[i] = 1;
    shape[0] = (U32)(sz > GB ? (sz >> 30) : sz); shape[1] = (U32)(sz > GB ? GB : 1); shape[2] = shape[3] = 1;
    for (int i = 0; i < 5; i++) grad[i] = mtum[i] = NULL;
    _tmp = NULL;                                          // the per-tensor scratch slot is gone: reductions use the backend workspace
    return *this;
}
Tensor &Tensor::reshape(U64 sz) {
    if (sz == numel) reset(data, numel, (t4_obj)ttype, grad_fn);
    else ERROR("  tensor#reshape sz != numel (%ld != %ld)\n", (long)sz, (long)numel);
    return *this;
}
Tensor &Tensor::reshape(U32 h, U32 w) {
    if ((U64)h * w == numel) { rank = 2; for (int i = 0; i < 4; i++) stride[i] = 1; shape[0] = h; shape[1] = w; shape[2] = shape[3] = 1; }
    else ERROR("  tensor#reshape sz != numel (%ld != %ld)\n", (long)((U64)h * w), (long)numel);
    return *this;
}
Tensor &Tensor::reshape(U32 n, U32 h, U32 w, U32 c) {
    if ((U64)n * h * w * c == numel) { rank = 4; for (int i = 0; i < 4; i++) stride[i] = 1; shape[0] = h; shape[1] = w; shape[2] = c; shape[3] = n; }
    else ERROR("  tensor#reshape sz != numel (%ld != %ld)\n", (long)((U64)n * h * w * c), (long)numel);
    return *this;
}
Tensor &Tensor::reshape(U32 c1, U32 n, U32 h, U32 w, U32 c) {
    if ((U64)c1 * n * h * w * c == numel) { rank = 5; iparm = c1; for (int i = 0; i < 4; i++) stride[i] = 1; shape[0] = h; shape[1] = w; shape[2] = c; shape[3] = n; }
    else ERROR("  tensor#reshape sz != numel (%ld != %ld)\n", (long)((U64)c1 * n * h * w * c), (long)numel);
    return *this;
}

// ===================================================================================================== mu/dataset.cu:20-121
Dataset::Dataset(U32 n, U32 h, U32 w, U32 c) : Tensor(n, h, w, c), label(NULL) { H_ALLOC(&label, n * sizeof(U32)); }
Dataset::~Dataset() { if (label) H_FREE((void *)label); }
void Dataset::normalize(DU mean, DU scale) {
    _mean = mean;
    if (ZEQ(scale)) { ERROR("scale == 0?\n"); _scale = 1.0f; } else _scale = 1.0f / scale;
}
int Dataset::fetch(char *ds_name, bool rewind, bool trace) {   // dataset.cu:64-121, trace text included
    static const char *fn = "dataset#fetch";
    if (trace) INFO("  %s %s batch[%d] {\n", fn, ds_name ? ds_name : (rewind ? "rewind" : ""), batch_id);
    ld::Corpus *cp = ld::Loader::get(*this, ds_name);
    if (!cp) { ERROR("  } %s => not found in Loader\n", fn); return -1; }
    if (ds_name) {                                        // first use: dimensions from the corpus
        if (cp->init(N(), trace) == NULL) { ERROR("  } %s => corpus init failed!\n", fn); return -2; }
        dataset_size = cp->corpus_sz;
        _reshape(cp->N, cp->H, cp->W, cp->C);
    }
    if (rewind) { cp->rewind(); batch_id = done = 0; }
    if (!cp->fetch(batch_id, trace)) { ERROR("  } %s => corpus fetch failed\n", fn); return -3; }
    int n = batch_sz = cp->batch_sz; done = cp->eof;
    if (trace) {
        INFO("  } %s => batch[%d] ", fn, batch_id);
        if (done) INFO("completed, no more data.\n");
        else      INFO("%d record(s) loaded\n", n);
    }
    _load(cp->data, cp->label, n);                        // t4k_bind.cpp: one H2D copy + u8 -> f32 on the GPU
    batch_id++;
    return 0;
}
} // namespace t4::mu

namespace t4::nn {
using mu::Tensor;
// ===================================================================================================== nn/forward.cu:27-113
Model &Model::forward(Tensor &input) {
    Tensor &n0 = (*this)[0];
    if (*_trace) input.show(true);                        // preview of the input (forward.cu:31)
    if (input.numel != n0.numel) {
        ERROR("nn#forward dataset wrong shape[%d,%d,%d,%d] != model input[%d,%d,%d,%d]\n", input.N(), input.H(), input.W(), input.C(), n0.N(), n0.H(), n0.W(), n0.C());
        return *this;
    }
    n0 = input;                                           // the batch is copied into layer 0
    auto info = [](DU t, int i, Tensor &in, Tensor &out) {  // forward.cu:44-50
        INFO("\n%6.2f:%3d> %s [%2d,%2d,%2d,%2d] Σ/n=%6.2f p=%6.3f => out[%2d,%2d,%2d,%2d]",
            t, i, nname(in.grad_fn), in.N(), in.H(), in.W(), in.C(),
            in.sum() / in.N() / in.C(), in.xparm,
            out.N(), out.H(), out.W(), out.C());
    };
    NLOG("\nModel::forward starts trace=%d {", *_trace);
    DU t0 = System::clock(), t1 = t0, tt;
    for (int i = 0; i < (int)numel - 1; i++) {
        Tensor &in = (*this)[i], &out = (*this)[i + 1];
        if (*_trace) { info((tt = System::clock()) - t1, i, in, out); t1 = tt; }
        _fstep(in, out);
        if (*_trace && _check_nan(out)) {
            ERROR("nn#forward Nan in %s\n", nname(in.grad_fn));
            INFO("in=");  in.show(true);
            INFO("out="); out.show(true);
            this->err = 1; break;
        }
        if (*_trace > 1) out.show(true);
    }
    if (input.is_dataset()) { onehot((mu::Dataset &)input); _hit = hit(true); }
    NLOG("\n} Model::forward %5.2f ms\n", System::clock() - t0);
    return *this;
}
void Model::_fstep(Tensor &in, Tensor &out) {
    const t4_layer fn = in.grad_fn;
    switch (fn) {
    case L_CONV:    _fconv(in, out); break;
    case L_LINEAR:  _flinear(in, out); break;
    case L_FLATTEN: out = in; break;
    case L_DROPOUT:                                       // the mask (forward.cu:100-103 RAND) is drawn by _factivate through t4k_dropout_mask:
                                                          // keyed by sample, so N ranks x B draw the masks of 1 x N.B (include/t4k.h)
    case L_RELU: case L_TANH: case L_SIGMOID: case L_SELU: case L_LEAKYRL: case L_ELU: _factivate(in, out, fn); break;
    case L_SOFTMAX: _fsoftmax(in, out); break;
    case L_LOGSMAX: _flogsoftmax(in, out); break;
    case L_AVGPOOL: case L_MAXPOOL: case L_MINPOOL: _fpool(in, out, fn); break;
    case L_BATCHNM: _fbatchnorm(in, out); break;
    case L_USAMPLE: _fupsample(in, out); break;
    case L_DCONV:   _bconv(in, out); break;               // the reference's dispatch (forward.cu:110)
    default: ERROR("nn#fstep layer=%d not supported\n", fn);
    }
}

// ===================================================================================================== nn/backprop.cu:17-140
Model &Model::broadcast(Tensor &tgt) {                    // [N,1] -> [N,HWC]: N row fills on the GPU instead of a host loop over managed memory
    Tensor &out = (*this)[-1];
    const U64 HWC = out.HWC(); const U32 N = out.N();
    if (!_hot) _hot = &T4(N, 1, (U32)HWC, 1);
    std::vector<DU> v(N);
    Tensor::d2h(v.data(), tgt.data, (int)(sizeof(DU) * N));
    for (U32 n = 0; n < N; n++) warn(t4k_math(T4K_FILL, _hot->slice(n), v[n], (long)HWC, nullptr), "nn#broadcast");
    return *this;
}
Model &Model::backprop() {
    if (_hot) return backprop(*_hot);
    ERROR("nn#backprop missing onehot vector?\n");
    return *this;
}
int Model::_bprep(Tensor &tgt) {
    Tensor &out = (*this)[-1];
    if (out.numel != tgt.numel) {
        ERROR("Model#bprep: Onehot wrong shape[%d,%d,%d,%d] != [%d,%d,%d,%d]\n", tgt.N(), tgt.H(), tgt.W(), tgt.C(), out.N(), out.H(), out.W(), out.C());
        return 1;
    }
    NLOG("Model::bprep input(onehot) numel=%ld OK {\n", (long)tgt.numel);
    switch ((*this)[-2].grad_fn) {
    case L_LINEAR: case L_SIGMOID: case L_SOFTMAX: case L_LOGSMAX: out -= tgt; break;   // dLoss = out - target
    default: out = tgt; break;                                                            // a pre-computed dLoss passes through
    }
    if (*_trace) out.show(true);                          // the loss derivative (backprop.cu:104)
    NLOG("}\n");
    return 0;
}
Model &Model::backprop(Tensor &tgt) {
    auto trace = [](DU t, int i, Tensor &in, Tensor &out) {   // backprop.cu:40-47
        INFO("\n%6.2f:%3d> %s [%2d,%2d,%2d,%2d] p=%6.3f <= out'Σ/n=%6.2f [%2d,%2d,%2d,%2d]",
            t, i, nname(in.grad_fn),
            in.N(), in.H(), in.W(), in.C(), in.xparm,
            out.sum() / out.N() / out.C(),
            out.N(), out.H(), out.W(), out.C());
    };
    if (_bprep(tgt)) return *this;
    NLOG("\nModel::backprop starts trace=%d train=%d {", *_trace, train);
    DU t0 = System::clock(), t1 = t0, tt;
    for (int i = (int)numel - 2, j = 0; i >= 0; i--, j++) {
        Tensor &in = (*this)[i], &out = (*this)[i + 1];
        if (*_trace) { trace((tt = System::clock()) - t1, i, in, out); t1 = tt; }
        _bstep(in, out, j == 0);
        if (*_trace && _check_nan(in)) { ERROR("nn#backprop Nan %s\n", nname(in.grad_fn)); in.show(); out.show(); this->err = 1; break; }
        if (*_trace > 1) in.show(true);
    }
    NLOG("\n} Model::backprop %5.2f ms\n", System::clock() - t0);
    return *this;
}
void Model::_bstep(Tensor &in, Tensor &out, bool last_layer) {
    const t4_layer fn = in.grad_fn;
    switch (fn) {
    case L_CONV:    _bconv(in, out); break;
    case L_LINEAR:  if (last_layer) in = out; else _blinear(in, out); break;
    case L_FLATTEN: case L_SIGMOID: case L_SOFTMAX: case L_LOGSMAX: in = out; break;      // pass-through
    case L_RELU: case L_TANH: case L_SELU: case L_LEAKYRL: case L_ELU: case L_DROPOUT: _bactivate(in, out); break;
    case L_MAXPOOL: case L_AVGPOOL: case L_MINPOOL: _bpool(in, out, fn); break;
    case L_BATCHNM: _bbatchnorm(in, out); break;
    case L_USAMPLE: _bupsample(in, out, fn); break;
    case L_DCONV:   _fconv(in, out); break;               // the reference's dispatch (backprop.cu:137)
    default: ERROR("nn#bstep layer=%d not supported\n", fn);
    }
}

// ===================================================================================================== nn/gradient.cu:20-126
#define M2X(i)     (in.mtum[i] ? _mmu->OBJ2X(*in.mtum[i]) : 0)
Model &Model::grad_alloc(t4_optimizer op) {                // :19-59 momentum / second-moment tensors, by optimizer (as written: a model that
    NLOG("  #grad_alloc {\n");                             // took its first step with nn.sgd keeps mtum[2] = NULL for a later nn.adam)
    for (int i = 0; i < (int)numel - 1; i++) {
        Tensor &in = (*this)[i];
        Tensor *w = in.grad[0], *b = in.grad[1];
        NLOG("    %3d> %8s w,b[%d,%d] ", i, nname(in.grad_fn), w ? 1 : 0, b ? 1 : 0);
        switch (op) {
        case OPTI_SGD:  in.mtum[0] = w; in.mtum[2] = NULL; in.mtum[1] = b; in.mtum[3] = NULL; break;
        case OPTI_SGDM:
            if (w && !in.mtum[0]) { in.mtum[0] = &T4(*w).zeros(); in.mtum[2] = NULL; }
            if (b && !in.mtum[1]) { in.mtum[1] = &T4(*b).zeros(); in.mtum[3] = NULL; }
            break;
        case OPTI_ADAM: case OPTI_ADAMW:
            if (w && !in.mtum[0]) { in.mtum[0] = &T4(*w).zeros(); in.mtum[2] = &T4(*w).zeros(); }
            if (b && !in.mtum[1]) { in.mtum[1] = &T4(*b).zeros(); in.mtum[3] = &T4(*b).zeros(); }
            break;
        }
        NLOG("mtum=%zx,%zx,%zx,%zx\n", (size_t)M2X(0), (size_t)M2X(1), (size_t)M2X(2), (size_t)M2X(3));
    }
    NLOG("  } #grad_alloc\n");
    return *this;
}
Model &Model::gradient(const char *nm, t4_optimizer op, GdFunc fn, DU *parm) {   // :63-126, with its trace text
    auto step = [this, fn, parm](const char k, Tensor &g, Tensor &dg, Tensor &m, Tensor &v) {
        NLOG("     %c[%2d,%2d,%2d,%2d] Σ=%6.3f - %6.3f", k, g.N(), g.H(), g.W(), g.C(), g.sum(), dg.sum());
        if (*_trace > 1 && g.numel < T4_DIM_SQ) {
            INFO("\nbefore %c =", k); Tensor::_dump(g.data, g.H(), g.W(), g.C());
            INFO("\nbefore d%c=", k); Tensor::_dump(dg.data, dg.H(), dg.W(), dg.C());
            fn(parm, g, dg, m, v);
            INFO("\nafter  %c =", k); Tensor::_dump(g.data, g.H(), g.W(), g.C());
            INFO("\nafter  d%c=", k); Tensor::_dump(dg.data, dg.H(), dg.W(), dg.C());
            INFO("\n");
        }
        else fn(parm, g, dg, m, v);
        NLOG(" => %cΣ=%6.3f\n", k, g.sum());
    };
    NLOG("\nModel::%s starts (%s) batch_sz=%d, lr=%7.4f, mtum/b1=%6.3f, b2=%6.3f {\n", nm, train ? "trainning" : "testing", (*this)[1].N(), parm[0], parm[1], parm[2]);
    if (_iter++ == 0 && epoch == 0) grad_alloc(op);
    if (!train) return *this;
    DU t0 = System::clock();
    for (int i = 0; i < (int)numel - 1; i++) {
        Tensor &in = (*this)[i];
        NLOG("  %d> %s\n", i, nname(in.grad_fn));
        if (in.mtum[0]) { step('w', *in.grad[0], *in.grad[2], *in.mtum[0], *in.mtum[2]); if (*_trace && _check_nan(*in.grad[0])) { ERROR("nn::grad.w Nan %s\n", nname(in.grad_fn)); in.grad[0]->show(); this->err = 1; break; } }
        if (in.mtum[1]) { step('b', *in.grad[1], *in.grad[3], *in.mtum[1], *in.mtum[3]); if (*_trace && _check_nan(*in.grad[1])) { ERROR("nn::grad.b Nan %s\n", nname(in.grad_fn)); in.grad[1]->show(); this->err = 1; break; } }
    }
    NLOG("} Model::%s %5.2f ms\n", nm, System::clock() - t0);
    return *this;
}
} // namespace t4::nn
