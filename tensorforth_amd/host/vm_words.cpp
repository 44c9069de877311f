// vm_words.cpp - tensor and neural-network vocabularies on top of the eForth core.
// Stack effects and multi-dispatch rules: reference src/vm/tenvm.cpp:44-636 (TensorVM) and
// src/vm/netvm.cpp:20-485 (NetVM), README.md:387-638.
#include "vm.h"

namespace t4 {

enum { B_DOT = 0, B_DIV, B_SOLV, B_INV, B_LUINV, B_PLU, B_TRIU, B_TRIL, B_XPOS, B_DET };

// ---------------------------------------------------------------- 1- and 2-operand math (tenvm.cpp:44-130)
void VM::xop1(int op, DU v) {
    if (!IS_OBJ(tos_)) { sxop1(op); return; }
    if (!TOS1T()) { pstr("tensor?"); return; }
    Tensor &A = TTOS();
    if (op == T4K_IDEN) A.identity(); else A.map(op, v);
}
void VM::xop2(int op, bool keep) {
    const int tt = (IS_OBJ(SS(-1)) ? 2 : 0) | (IS_OBJ(tos_) ? 1 : 0);
    switch (tt) {
    case 0: sxop2(op); break;
    case 1: {                                            // scalar-tensor ( n T -- T' )
        const DU v = SS(-1); Tensor &A = TTOS();
        Tensor &O = keep ? st().copy(A) : A;
        if (op == T4K_DIV || op == T4K_SUB) {            // op(scalar, tensor): materialise the scalar (tenvm.cpp:252-263)
            Tensor &B = st().tensor(A.numel); B.map(T4K_FILL, v);
            Tensor::ten_op(op, B, A, O); st().free(B);
        } else Tensor::ten_op(op, A, v, O);
        if (keep) PUSH(O); else ss_.pop_back();
    } break;
    case 2: {                                            // tensor-scalar ( T n -- T' )
        Tensor &A = TNOS();
        Tensor &O = keep ? st().copy(A) : A;
        Tensor::ten_op(op, A, tos_, O);
        if (keep) PUSH(O); else POP();
    } break;
    case 3: {                                            // tensor-tensor
        Tensor &A = TNOS(), &B = TTOS();
        if ((A.N() == 1 || B.N() == 1) && A.HWC() != B.HWC()) { pstr("} dim?\n"); break; }
        Tensor &O = st().copy(A.N() == 1 ? B : A);
        Tensor::ten_op(op, A, B, O);
        if (B.rank == 1) O.reshape(O.numel);
        if (!keep) { DROP(POP()); DROP(POP()); }
        PUSH(O);
    } break;
    }
}

// ---------------------------------------------------------------- linear algebra words (tenvm.cpp:134-384)
static Tensor &tinv(Tensor &A, bool use_lu) {
    const int K = A.H();
    Tensor &I = Store::get().tensor(K, K).identity();
    if (use_lu) Tensor::lu_inverse(A, I); else Tensor::inverse(A, I);
    return I;
}
void VM::blas1(int op) {
    Tensor &A = TTOS();
    if (!TOS1T() || A.rank != 2) { pstr("tensor2?"); return; }
    Tensor &T = st().copy(A);
    bool tx = true;
    switch (op) {
    case B_INV:   { Tensor &I = tinv(T, false); PUSH(I); st().free(T); tx = false; } break;
    case B_LUINV: { Tensor &I = tinv(T, true);  PUSH(I); st().free(T); tx = false; } break;
    case B_PLU: {
        Tensor &piv = st().tensor(A.H());
        Tensor &I = st().copy(T).identity();
        Tensor::plu(T, I, (int *)piv.data);
        PUSH(I); st().free(piv);
    } break;
    case B_TRIU: Tensor::lu(T, true); break;
    case B_TRIL: Tensor::lu(T, false); break;
    case B_XPOS: T.reshape(A.W(), A.H()); Tensor::transpose(A, T); break;
    case B_DET:  { DU v = T.det(); PUSH(v); st().free(T); tx = false; } break;
    default: st().free(T); tx = false;
    }
    if (tx) PUSH(T);
}
void VM::blas2(int op, bool keep) {
    if (!TOS2T()) { pstr("tenvm#blas2 TNOS TTOS required!\n"); return; }
    Tensor &A = TNOS(), &B = TTOS();
    switch (op) {
    case B_DOT: {                                        // _tdot tenvm.cpp:329-367
        Tensor *C = nullptr;
        if (A.rank == 1 && B.rank == 1 && A.numel == B.numel) { PUSH(A.dot(B)); return; }
        if (B.rank == 1 && A.W() == B.numel)            { C = &st().tensor(A.H()); Tensor::mm(A, B, *C); }
        else if (A.rank == 2 && B.rank == 2 && A.W() == B.H()) { C = &st().tensor(A.H(), B.W()); Tensor::mm(A, B, *C); }
        else if ((A.N() == 1 || B.N() == 1) && A.N() != B.N() && A.C() == B.C() && A.W() == B.H()) {
            C = &st().tensor(std::max(A.N(), B.N()), A.H(), B.W(), A.C()); Tensor::mm(A, B, *C);
        } else { pstr("A.W != B.H dim?"); return; }
        if (!keep) { DROP(POP()); DROP(POP()); }
        PUSH(*C);
    } break;
    case B_DIV: {                                        // C = A @ inverse(B)
        if (B.H() != B.W() || A.W() != B.H()) return;
        Tensor &I = tinv(B, true);
        Tensor &O = st().tensor(A.H(), B.W());
        Tensor::mm(A, I, O); st().free(I);
        PUSH(O);
    } break;
    case B_SOLV: {                                       // ( B A -- B A X ) solve B = A X   (_solv: note A,B flipped)
        Tensor &Am = B, &Bv = A;
        if (Bv.rank != 1 || Am.H() != Am.W() || Am.W() != Bv.H()) { PUSH(Bv); return; }
        Tensor &T = st().copy(Am);
        Tensor &I = tinv(T, true);
        Tensor &O = st().tensor(Am.W());
        Tensor::mm(I, Bv, O);
        st().free(I); st().free(T);
        PUSH(O);
    } break;
    }
}
void VM::gemm(int opt) {                                 // ( a b A B C -- a b A B C O )  tenvm.cpp:224-248
    if (!TOS3T()) { pstr("tensors?"); return; }
    Tensor &C = TTOS(), &B = TNOS(), &A = (Tensor &)st().du2obj(SS(-2));
    const DU b = SS(-3), a = SS(-4);
    if (A.W() == B.H() && A.H() == C.H() && B.W() == C.W()) {
        Tensor &O = st().copy(C);
        Tensor::gemm(opt, A, B, O, a, b);
        PUSH(O);
    } else pstr("dim?");
}

// ---------------------------------------------------------------- tensor vocabulary
void VM::init_tensor() {
    auto CODE = [this](const char *n, std::function<void()> f) { add(n, std::move(f), false); };
    CODE("\nTensor::", [] {});
    CODE("vector", [this] { uint32_t sz = (uint32_t)POPi(); PUSH(st().tensor((uint64_t)sz)); });
    CODE("matrix", [this] { uint32_t w = (uint32_t)POPi(), h = (uint32_t)POPi(); PUSH(st().tensor(h, w)); });
    CODE("tensor", [this] { uint32_t c = (uint32_t)POPi(), w = (uint32_t)POPi(), h = (uint32_t)POPi(), n = (uint32_t)POPi(); PUSH(st().tensor(n, h, w, c)); });
    auto lit_begin = [this](uint32_t off) { ten_off_ = ten_base_ = off; ten_lvl_ = 1; ten_stage_.clear(); };
    auto lit_flush = [this] { if (!ten_stage_.empty() && TOS1T()) TTOS().from_host(ten_stage_.data(), ten_stage_.size(), ten_base_); ten_stage_.clear(); ten_base_ = ten_off_; };
    CODE("vector{", [this, lit_begin] { uint32_t sz = (uint32_t)POPi(); PUSH(st().tensor((uint64_t)sz)); lit_begin(0); });
    CODE("matrix{", [this, lit_begin] { uint32_t w = (uint32_t)POPi(), h = (uint32_t)POPi(); PUSH(st().tensor(h, w)); lit_begin(0); });
    CODE("view",  [this] { PUSH(DUP(tos_)); });
    CODE("copy",  [this] { if (IS_OBJ(tos_)) PUSH(st().copy(TTOS())); else PUSH(tos_); });
    CODE("flatten",  [this] { Tensor &t = TTOS(); t.reshape(t.numel); });
    CODE("reshape2", [this] { uint32_t w = (uint32_t)POPi(), h = (uint32_t)POPi(); TTOS().reshape(h, w); });
    CODE("reshape4", [this] { uint32_t c = (uint32_t)POPi(), w = (uint32_t)POPi(), h = (uint32_t)POPi(), n = (uint32_t)POPi(); TTOS().reshape(n, h, w, c); });
    CODE("same_shape?", [this] { if (IS_OBJ(tos_) && IS_OBJ(SS(-1))) PUSH(BOOL(TTOS().same_shape(TNOS()))); else pstr("TOS,NOS tensors?"); });
    CODE("={", [this, lit_begin] { if (IS_OBJ(tos_)) lit_begin(0); else { uint32_t o = (uint32_t)POPi(); lit_begin(o); ten_lvl_ = IS_OBJ(tos_) ? 1 : 0; } });
    CODE("zeros",    [this] { xop1(T4K_FILL, 0.0f); });
    CODE("ones",     [this] { xop1(T4K_FILL, 1.0f); });
    CODE("fill",     [this] { DU v = POP(); xop1(T4K_FILL, v); });
    CODE("gradfill", [this] { xop1(T4K_GFILL, 1.0f); });
    CODE("eye",      [this] { xop1(T4K_IDEN); });
    auto rnd = [this](int opt) { if (TOS1T()) { Tensor &t = TTOS(); chk(t4k_rand(t.data, (long)t.numel, opt, 0.0f, 1.0f, stream()), "rand"); } };
    CODE("rand",  [rnd] { rnd(T4K_UNIFORM); });
    CODE("randn", [rnd] { rnd(T4K_NORMAL); });
    CODE("normalize", [this] { DU std = POP(), avg = POP(); if (TOS1T()) TTOS().normalize(std, avg); });   // args swapped as in tenvm.cpp:513-515
    CODE("sum",  [this] { if (TOS1T()) PUSH(TTOS().sum()); });
    CODE("avg",  [this] { if (TOS1T()) PUSH(TTOS().avg()); });
    CODE("std",  [this] { if (TOS1T()) PUSH(TTOS().std()); });
    CODE("norm", [this] { if (TOS1T()) PUSH(TTOS().norm()); });
    CODE("{", [this] { if (TOS1T() && ten_lvl_ > 0) ++ten_lvl_; });
    CODE("}", [this, lit_flush] { if (TOS1T() && ten_lvl_ > 0) { if (--ten_lvl_ == 0) lit_flush(); } });
    CODE("slice", [this] {
        uint32_t y1 = (uint32_t)POPi(), y0 = (uint32_t)POPi(), x1 = (uint32_t)POPi(), x0 = (uint32_t)POPi();
        if (TOS1T()) PUSH(st().slice(TTOS(), x0, x1, y0, y1));
    });
    CODE("dim", [this] { if (TOS1D()) PUSH(st().dim(TTOS())); else pstr("TOS tensor?"); });
    CODE("t@",  [this] { if (!IS_OBJ(tos_) && IS_OBJ(SS(-1))) { int i = POPi(); DU v = TTOS().get(i); PUSH(SCALAR(v)); } });
    CODE("t!",  [this] { int i = POPi(); DU v = POP(); if (IS_OBJ(tos_)) TTOS().set(i, v); });
    CODE("exp",     [this] { xop1(T4K_EXP); });
    CODE("ln",      [this] { xop1(T4K_LN); });
    CODE("log",     [this] { xop1(T4K_LOG); });
    CODE("tanh",    [this] { xop1(T4K_TANH); });
    CODE("relu",    [this] { xop1(T4K_RELU); });
    CODE("sigmoid", [this] { xop1(T4K_SIGM); });
    CODE("sqrt",    [this] { xop1(T4K_SQRT); });
    CODE("1/x",     [this] { xop1(T4K_RCP); });
    CODE("sat",     [this] { xop1(T4K_SAT); });
    CODE("pow",     [this] { sxop2(T4K_POW); });
    CODE("sin",     [this] { xop1(T4K_SIN); });
    CODE("cos",     [this] { xop1(T4K_COS); });
    CODE("PI",      [this] { PUSH(SCALAR(3.1415927f)); });
    CODE("inverse",   [this] { blas1(B_INV); });
    CODE("luinv",     [this] { blas1(B_LUINV); });
    CODE("plu",       [this] { blas1(B_PLU); });
    CODE("upper",     [this] { blas1(B_TRIU); });
    CODE("lower",     [this] { blas1(B_TRIL); });
    CODE("transpose", [this] { blas1(B_XPOS); });
    CODE("det",       [this] { blas1(B_DET); });
    CODE("+=", [this] { xop2(T4K_ADD, false); });
    CODE("-=", [this] { xop2(T4K_SUB, false); });
    CODE("*=", [this] { xop2(T4K_MUL, false); });
    CODE("/=", [this] { xop2(T4K_DIV, false); });
    CODE("@=",     [this] { blas2(B_DOT, false); });
    CODE("matmul", [this] { blas2(B_DOT, true); });
    CODE("matdiv", [this] { blas2(B_DIV, true); });
    CODE("solve",  [this] { blas2(B_SOLV, true); });
    CODE("gemm",  [this] { gemm(0); });
    CODE("gemm1", [this] { gemm(1); });
    CODE("gemm2", [this] { gemm(2); });
    CODE("gemm3", [this] { gemm(3); });
    CODE("gemm4", [this] { gemm(4); });
    // file access modes, io/ostream.h:43-47 (FAM_WO = 0, FAM_RO = 1, FAM_RW = 2, FAM_RAW = 3)
    CODE("bin", [this] { PUSH(3.0f); });
    CODE("w/o", [this] { PUSH(0.0f); });
    CODE("r/w", [this] { PUSH(2.0f); });
    // ( T adr len [mode] -- T ) tenvm.cpp:389-410 -> sys.cpp:153-163 -> AIO::tsave aio_tensor.cpp:75-93.
    // text (default): the printed form with the 1024-cell threshold (:232-238).  raw (`bin`): 'T','4', shape[4] as U32 {H,W,C,N}, then
    // one byte per element, (U8)(v * 256) slice by slice (:240-255) - the byte layout is the reference's.  Reference hazard decided:
    // tsave tests mode bits (FAM_RW = 2 is a bit of FAM_RAW = 3), so its `bin save` opens the file read-only and writes nothing; here
    // `bin` writes the raw format it defines.  `load` of a tensor has no handler in the reference (OP_TLOAD falls through the switch at
    // sys.cpp:145-215); here it fills the tensor on the stack from a raw file of the same element count (v = byte / 256).
    auto tsave = [this](bool load) {
        int mode = 0;
        if (SP() > 1 && IS_OBJ(SS(-2))) { /* ( T adr len ) */ }
        else if (SP() > 2 && IS_OBJ(SS(-3))) mode = POPi();   // the reference's order of the two tests (tenvm.cpp:393-395): an object deeper in the stack is no mode
        else { pstr("tensor adr len [mode]?\n"); return; }
        POPi(); uint32_t adr = (uint32_t)POPi();
        const char *fn = (const char *)&pmem_[adr];
        if (!TOS1T()) { pstr("tensor adr len [mode]?\n"); return; }
        hold_begin(); hold_ = true;                      // syscall(OP_TSAVE / OP_TLOAD), tenvm.cpp:408 (serviced at the flush: its messages follow the buffered text)
        Tensor &t = TTOS();
        if (load) {
            FILE *f = fopen(fn, "rb"); if (!f) { pstr(" failed to open for input\n"); return; }
            char hdr[2] = {0, 0}; uint32_t shp[4] = {0, 0, 0, 0};
            const bool ok = fread(hdr, 1, 2, f) == 2 && hdr[0] == 'T' && hdr[1] == '4' && fread(shp, 4, 4, f) == 4;
            const uint64_t n = (uint64_t)shp[0] * shp[1] * shp[2] * shp[3];
            if (!ok || n != t.numel) { fclose(f); pstr(ok ? " tensor load: element count differs\n" : " tensor load: not a raw T4 file\n"); return; }
            std::vector<uint8_t> b(n); std::vector<float> h(n);
            if (fread(b.data(), 1, n, f) != n) { fclose(f); pstr(" tensor load: short file\n"); return; }
            fclose(f);
            for (uint64_t i = 0; i < n; i++) h[i] = (float)b[i] / 256.0f;
            t.from_host(h.data(), n);
            return;
        }
        FILE *f = fopen(fn, mode == 3 ? "wb" : "w"); if (!f) { pstr(" failed to open for output\n"); return; }
        if (mode == 3) {
            std::vector<float> h; t.to_host(h, t.numel);
            const uint32_t shp[4] = { t.H(), t.W(), t.C(), t.N() };
            std::vector<uint8_t> b(t.numel);
            for (uint64_t i = 0; i < t.numel; i++) b[i] = static_cast<uint8_t>(h[i] * 256.0);   // [0,1) => [0,256); the double product of :250
            fwrite("T4", 1, 2, f); fwrite(shp, 4, 4, f); fwrite(b.data(), 1, b.size(), f);
        } else { std::string s = fmt_tensor(t, 1024); fwrite(s.data(), 1, s.size(), f); }
        fclose(f);
    };
    CODE("save", [tsave] { tsave(false); });
    CODE("load", [tsave] { tsave(true); });
    // TensorBoard words (tenvm.cpp:603-612 -> sys.cpp:230-273): written by host/tboard.cpp when a log directory is configured
    // (`ten4 -t <logdir> -r <run>` / T4_TB_LOGDIR), otherwise the reference's hint is printed, as the reference does without -t
    auto tb = [this](const char *nm, int npop, bool has_tag) {
        std::string tag, txt;
        if (has_tag) { POPi(); tag = (const char *)&pmem_[(uint32_t)POPi()]; }
        DU n = 0; int i = 0;
        if (npop == 3) { POPi(); txt = (const char *)&pmem_[(uint32_t)POPi()]; }   // .text: ( txt_addr len tag_addr len -- )
        if (npop == 2) { i = POPi(); n = POP(); }
        if (npop == 1) { n = POP(); }
        if (tb_active()) {
            Tensor *t = IS_OBJ(n) ? &(Tensor &)st().du2obj(n) : nullptr;
            if (!strcmp(nm, "init")) tb_init(tag.c_str());
            else if (!strcmp(nm, "scalar")) tb_scalar(tag.c_str(), n);
            else if (!strcmp(nm, "text")) tb_text(tag.c_str(), txt.c_str());
            else if (!strcmp(nm, "histo") && t) tb_histo(tag.c_str(), *t, i);
            else if (!strcmp(nm, "tile") && t) tb_tile(tag.c_str(), *t, i);
            else if (!strcmp(nm, "image") && t) tb_image(tag.c_str(), *t);
            else if (!strcmp(nm, "embed") && t) tb_embed(tag.c_str(), *t);
            else { char b[200]; snprintf(b, sizeof(b), "  sys#tbx(op=%s, tag=%s): not written by this sink\n", nm, tag.c_str()); pstr(b); }
        } else {                                         // System::_process_tb sys.cpp:234-253 without -t: op as the TB_OP number, n as the raw cell (an object prints its handle)
            static const char *ops[] = {"init", "step", "scalar", "text", "image", "tile", "histo", "graph", "embed"};
            int op = 0; while (op < 9 && strcmp(nm, ops[op])) op++;
            char b[200]; snprintf(b, sizeof(b), "  sys#tbx(op=%d, n=%g, i=%d, tag=%s)\n", op, n, i, tag.c_str());
            pstr(b);
        }
        if (IS_OBJ(n)) { st().mark_free(n); hold_end(); }   // tenvm.cpp:418-420
    };
    CODE(".tbinit", [tb] { tb("init", 0, true); });
    CODE(".tbstep", [this] { int i = POPi(); if (tb_active()) { tb_step(i); return; }
                             char b[96]; snprintf(b, sizeof(b), "  sys#tbx(op=1, n=0, i=%d), check TensorBoard param -tlogdir -rrun_id\n", i); pstr(b); });
    CODE(".scalar", [tb] { tb("scalar", 1, true); });
    CODE(".text",   [tb] { tb("text", 3, true); });
    CODE(".image",  [tb] { tb("image", 1, true); });
    CODE(".tile",   [tb] { tb("tile", 2, true); });
    CODE(".histo",  [tb] { tb("histo", 2, true); });
    CODE(".embed",  [tb] { tb("embed", 1, true); });
    CODE(".graph",  [this] {                              // ( N -- ) tenvm.cpp:611 -> sys.cpp:241: the model's layer list as a GraphDef event
        DU n = POP();
        if (!tb_active()) { char b[120]; snprintf(b, sizeof(b), "  sys#tbx(op=7, n=%g, i=0), check TensorBoard param -tlogdir -rrun_id\n", n); pstr(b); return; }
        if (is_m(n)) tb_graph((Model &)st().du2obj(n)); else pstr("summary#graph requires model\n");
    });
    CODE(".png",    [this] { POPi(); POPi(); pstr("  .png: n/a\n"); });
    // redefined words
    CODE("@", [this] { if (TOS2T()) blas2(B_DOT, true); else { uint32_t i = (uint32_t)POPi(); PUSH(DUP(mem_du(i))); } });
    CODE("max", [this] { if (IS_OBJ(tos_)) PUSH(TTOS().max()); else { DU n = ss_pop(); tos_ = SCALAR(fmaxf(n, tos_)); } });
    CODE("min", [this] { if (IS_OBJ(tos_)) PUSH(TTOS().min()); else { DU n = ss_pop(); tos_ = SCALAR(fminf(n, tos_)); } });
}

// ---------------------------------------------------------------- nn vocabulary helpers (netvm.cpp:20-286)
void VM::nnop(int op) {
    if (TOS1T()) {                                       // tensor ops (destructive)
        Tensor &t = TTOS();
        switch (op) {
        case T4K_L_FLATTEN: t.reshape(t.numel); return;
        case T4K_L_RELU:    t.map(T4K_RELU); return;
        case T4K_L_TANH:    t.map(T4K_TANH); return;
        case T4K_L_SIGMOID: t.map(T4K_SIGM); return;
        case T4K_L_SOFTMAX: { DU mx = t.max(); Tensor::ten_op(T4K_SUB, t, mx, t); t.map(T4K_EXP); t.map(T4K_MUL, 1.0f / t.sum()); } return;
        case T4K_L_LOGSMAX: { DU sum = t.sum(); if (sum > DU_EPS) Tensor::ten_op(T4K_SUB, t, log10f(sum), t); else pstr("logsoftmax tensor sum < 0!"); } return;
        default: break;
        }
    }
    if (is_m(tos_)) {                                    // zero-parameter layers
        Model &m = MTOS();
        switch (op) {
        case T4K_L_FLATTEN: case T4K_L_RELU: case T4K_L_TANH: case T4K_L_SIGMOID: case T4K_L_SELU:
        case T4K_L_SOFTMAX: case T4K_L_LOGSMAX: m.add(op); return;
        case T4K_L_LEAKYRL: m.add(op, 0, 0.01f); return;
        case T4K_L_ELU:     m.add(op, 0, 1.0f); return;
        case T4K_L_BATCHNM: m.add(op, 0, 0.1f); return;
        default: break;
        }
    }
    if (M1V()) {                                         // one-parameter layers
        DU a = POP(); Model &m = MTOS();
        switch (op) {
        case T4K_L_LINEAR:  m.add(op, (uint32_t)(int)a, 1.0f); return;        // bias = 1.0 (netvm.cpp:77)
        case T4K_L_LEAKYRL: case T4K_L_ELU: case T4K_L_DROPOUT: m.add(op, 0, a); return;
        case T4K_L_AVGPOOL: case T4K_L_MAXPOOL: case T4K_L_MINPOOL: m.add(op, (uint32_t)(int)a); return;
        case T4K_L_BATCHNM: m.add(op, 0, a); return;
        case T4K_L_USAMPLE: m.add(op, (uint32_t)(int)a, 0.0f); return;
        default: break;
        }
        PUSH(a);
    }
    switch (op) {
    case T4K_L_LINEAR:
        if (M2V()) { uint32_t c = (uint32_t)POPi(); DU bias = POP(); MTOS().add(op, c, bias); }
        else pstr("( N [bias] n -- ) for linear required!");
        break;
    case T4K_L_FLATTEN: case T4K_L_SELU: case T4K_L_SOFTMAX: case T4K_L_LOGSMAX: pstr("( N -- ) no param needed!"); break;
    case T4K_L_LEAKYRL: case T4K_L_ELU: case T4K_L_DROPOUT: case T4K_L_AVGPOOL: case T4K_L_MAXPOOL: case T4K_L_MINPOOL:
    case T4K_L_BATCHNM: pstr("( N n -- ) one param required!"); break;
    case T4K_L_USAMPLE:
        if (M2V()) { uint16_t n = (uint16_t)POPi(); DU m = POP(); MTOS().add(op, n, m); }
        else pstr("( N [mtum] n -- ) for upsample required?");
        break;
    default:
        if (!IS_OBJ(tos_)) {
            switch (op) {
            case T4K_L_RELU: sxop1(T4K_RELU); break;
            case T4K_L_TANH: sxop1(T4K_TANH); break;
            case T4K_L_SIGMOID: sxop1(T4K_SIGM); break;
            default: pstr("nnop !IS_OBJ: layer not supported\n");
            }
        } else pstr("layer not supported(2)\n");
    }
}
void VM::conv(uint16_t k, bool txn, uint16_t s, uint16_t p, uint16_t d) {   // netvm.cpp:203-226
    uint16_t opt[] = {k, s, p, d};
    if (TOS1T()) {
        Tensor &t = TTOS();
        if (t.rank == 1) {
            std::vector<float> vo; t.to_host(vo, std::min<uint64_t>(t.numel, 4));
            DU x = POP(); DROP(x);
            for (size_t i = 0; i < vo.size(); i++) opt[i] = (uint16_t)(int)vo[i];
        } else { pstr("vec?"); return; }
    }
    if (!M2V()) { pstr("Model#add bias c for conv2d/dconv2d required!"); return; }
    const uint32_t c = (uint32_t)POPi(); const DU bias = POP();
    MTOS().add(txn ? T4K_L_DCONV : T4K_L_CONV, c, bias, opt);
}
void VM::loss(Loss op) {                                  // netvm.cpp:268-286
    if (TOS2T()) { Tensor &tmp = st().copy(TNOS()); DU n = tmp.loss(op, TTOS()); st().free(tmp); PUSH(n); }
    else if (TOS1T() && is_m(SS(-1))) { DU n = MNOS().loss(op, TTOS()); POP(); PUSH(n); }
    else if (is_m(tos_)) PUSH(MTOS().loss(op));
    else pstr("model?\n");
}
void VM::get_parm(int n) {                                // netvm.cpp:157-169
    if (!M1V() || n > 4) { pstr("N n(<5) required?"); return; }
    int i = POPi();
    Tensor &t = MTOS().at(i);
    Tensor *p = n ? t.grad[n] : (t.grad[0] ? t.grad[0] : t.grad[4]);
    if (p) { DU v = st().obj2du(*p); PUSH(DUP(v)); } else PUSH(0.0f);
}
void VM::set_parm(int n) {                                // netvm.cpp:174-193
    if (!MTV()) { pstr("N T n required?"); return; }
    int i = POPi();
    Tensor &t = TTOS();
    Tensor &mt = MNOS().at(i);
    Tensor *p = n ? mt.grad[n] : (mt.grad[0] ? mt.grad[0] : mt.grad[4]);
    if (p && t.numel == p->numel) {
        if (p != &t) { *p = t; DU x = POP(); DROP(x); }
        else pstr("Updating the same param tensor");
    } else { PUSH((DU)i); pstr("Tensor and model parameter is not the same shape"); }
}

void VM::init_nn() {
    auto CODE = [this](const char *n, std::function<void()> f) { add(n, std::move(f), false); };
    CODE("\nNetwork::", [] {});
    CODE("nn.model", [this] {
        if (SP() < 4 || IS_OBJ(tos_) || IS_OBJ(SS(-1)) || IS_OBJ(SS(-2)) || IS_OBJ(SS(-3))) { pstr("n h w c?\n"); return; }
        uint32_t c = (uint32_t)POPi(), w = (uint32_t)POPi(), h = (uint32_t)POPi(), n = (uint32_t)POPi();
        Model &m = st().model(&trace_lvl);
        m.layer.push_back(&st().tensor(n, h, w, c));
        PUSH(m);
    });
    CODE("conv1x1", [this] { conv(1); });
    CODE("conv2d",  [this] { conv(3); });
    CODE("dconv2d", [this] { conv(4, true, 2); });
    CODE("linear",  [this] { nnop(T4K_L_LINEAR); });
    CODE("relu",    [this] { nnop(T4K_L_RELU); });
    CODE("tanh",    [this] { nnop(T4K_L_TANH); });
    CODE("sigmoid", [this] { nnop(T4K_L_SIGMOID); });
    CODE("selu",    [this] { nnop(T4K_L_SELU); });
    CODE("leakyrelu", [this] { nnop(T4K_L_LEAKYRL); });
    CODE("elu",     [this] { nnop(T4K_L_ELU); });
    CODE("softmax", [this] { nnop(T4K_L_SOFTMAX); });
    CODE("logsoftmax", [this] { nnop(T4K_L_LOGSMAX); });
    CODE("batchnorm", [this] { nnop(T4K_L_BATCHNM); });
    CODE("maxpool", [this] { nnop(T4K_L_MAXPOOL); });
    CODE("avgpool", [this] { nnop(T4K_L_AVGPOOL); });
    CODE("minpool", [this] { nnop(T4K_L_MINPOOL); });
    CODE("dropout", [this] { nnop(T4K_L_DROPOUT); });
    CODE("upsample", [this] { nnop(T4K_L_USAMPLE); });
    CODE("loss.mse", [this] { loss(LOSS_MSE); });
    CODE("loss.bce", [this] { loss(LOSS_BCE); });
    CODE("loss.ce",  [this] { loss(LOSS_CE); });
    CODE("loss.nll", [this] { loss(LOSS_NLL); });
    CODE("nn.loss", [this] {
        if (is_m(tos_) || (TOS1T() && is_m(SS(-1)))) {
            Model &m = is_m(tos_) ? MTOS() : MNOS();
            switch (m.at(-2).grad_fn) {
            case T4K_L_TANH: case T4K_L_SIGMOID: loss(LOSS_BCE); break;
            case T4K_L_SOFTMAX: loss(LOSS_CE); break;
            case T4K_L_LOGSMAX: loss(LOSS_NLL); break;
            default: loss(LOSS_MSE);
            }
        } else pstr("TOS is not a tensor or NOS is not a model!\n");
    });
    CODE("nn.onehot",  [this] { if (is_m(tos_)) { DU v = st().obj2du(MTOS().onehot()); PUSH(DUP(v)); } else pstr("TOS is not a model!\n"); });
    CODE("nn.onehot=", [this] { if (IS_OBJ(tos_) && is_m(SS(-1))) { Tensor &hot = (Tensor &)st().du2obj(POP()); MTOS().onehot(hot); } else pstr("model tensor?\n"); });
    CODE("nn.hit",  [this] { if (is_m(tos_)) PUSH((DU)MTOS().hit(false)); else pstr("TOS is not a model!\n"); });
    CODE("nn.zero", [this] { if (is_m(tos_)) { MTOS().iter = 0; MTOS().hit_ = 0; } else pstr("TOS is not a model!\n"); });
    CODE("nn.sgd", [this] {
        if (M2V()) { DU b = POP(), lr = POP(); MTOS().sgd(lr, b); }
        else if (M1V()) { DU lr = POP(); MTOS().sgd(lr); }
        else pstr("rate mtum nn.sgd?\n");
    });
    CODE("nn.adam", [this] {
        if (M2V()) { DU b1 = POP(), lr = POP(); MTOS().adam(lr, b1); }
        else if (M1V()) { DU lr = POP(); MTOS().adam(lr); }
        else pstr("rate [beta1] nn.adam?\n");
    });
    CODE("nn.adamw", [this] {                            // the reference's nn.adamw runs plain Adam (netvm.cpp:400-410)
        if (M2V()) { DU wd = POP(), lr = POP(); MTOS().adam(lr, wd); }
        else if (M1V()) { DU lr = POP(); MTOS().adam(lr); }
        else pstr("rate [wd] nn.adamw?\n");
    });
    CODE("nn.max_norm", [this] { if (M1V()) MTOS().max_norm = POP(); else pstr("norm model?\n"); });
    CODE("trainable", [this] { if (M1V()) { bool on = POPi() != 0; MTOS().train = on; } else pstr("N [1|0] required\n"); });
    CODE("batchsize", [this] { if (is_m(tos_)) PUSH((DU)MTOS().batch_size()); else pstr("TOS a model?\n"); });
    CODE("dataset", [this] {
        const char *dsn = fetch(); std::string name = dsn ? dsn : "";
        Dataset &ds = st().dataset((uint32_t)POPi());
        PUSH(ds);
        hold_begin();
        ds.fetch(name.c_str(), false);                   // loads batch 0 immediately (sys.cpp:166-174)
        hold_end();
    });
    CODE("normalize", [this] {                           // ( DS mean scale -- DS' ) on a dataset, else the tensor word
        if (SP() > 1 && is_d(SS(-2))) {
            DU scale = POP(); int mean = POPi();
            Dataset &ds = (Dataset &)st().du2obj(tos_);
            hold_begin();
            hprintf("  OP_NORM(mean=%d, scale=%g)\n", mean, scale);          // a plain printf in the reference (sys.cpp:181): in front of the fetch's trace text
            ds.set_norm((DU)mean, scale); ds.fetch(nullptr, true);
            hold_end();
        } else { DU std = POP(), avg = POP(); if (TOS1T()) TTOS().normalize(std, avg); }
    });
    CODE("fetch",  [this] { hold_begin(); if (is_d(tos_)) ((Dataset &)st().du2obj(tos_)).fetch(nullptr, false); hold_end(); });
    CODE("rewind", [this] { hold_begin(); if (is_d(tos_)) ((Dataset &)st().du2obj(tos_)).fetch(nullptr, true); hold_end(); });
    CODE("forward", [this] {                             // netvm.cpp:230-247
        if (is_m(SS(-1)) && TOS1D()) {
            DU x = POP();
            MTOS().forward((Tensor &)st().du2obj(x));
            if (MTOS().err) stop_ = true;
            DROP(x);
        } else if (is_m(tos_) && !rs_.empty() && IS_OBJ(RS(-1))) {
            Tensor &t = (Tensor &)st().du2obj(RS(-1));
            if (t.type == T_DATASET) { MTOS().forward(t); if (MTOS().err) { rs_pop(); stop_ = true; } }
            else pstr("rs[-1] is not a dataset?\n");
        } else pstr("no NN model nor a dataset?\n");
    });
    CODE("backprop", [this] {                            // netvm.cpp:251-264
        if (is_m(SS(-1)) && TOS1T()) { Tensor &t = TTOS(); MNOS().backprop(t); if (MNOS().err) stop_ = true; DU x = POP(); DROP(x); }
        else if (is_m(tos_)) { MTOS().backprop(); if (MTOS().err) stop_ = true; }
        else pstr("TOS not a NN model?\n");
    });
    CODE("broadcast", [this] {
        if (is_m(SS(-1)) && TOS1T()) { DU y = POP(); MTOS().broadcast((Tensor &)st().du2obj(y)); DROP(y); }
        else pstr("TOS not a tensor nor NOS a model?\n");
    });
    CODE("network", [this] { if (is_m(tos_)) { pstr(fmt_model(MTOS())); pstr(" "); } });
    CODE(">n", [this] { if (M1V()) { DU t = POP(); if (IS_OBJ(t)) { MTOS().layer.push_back(&(Tensor &)st().du2obj(t)); MTOS().invalidate(); } } });   // a new layer: fused-run plan and slab are stale
    CODE("n@", [this] { if (!M1V()) return; int i = POPi(); DU v = st().obj2du(MTOS().at(i)); PUSH(DUP(v)); });
    CODE("nn.len", [this] {
        if (IS_OBJ(tos_)) {
            Obj &o = st().du2obj(tos_);
            PUSH(o.type == T_MODEL ? (DU)((Model &)o).layer.size() : (o.type == T_TENSOR ? (DU)((Tensor &)o).N() : (DU)((Dataset &)o).dataset_size));
        } else pstr("TOS a tensor, dataset, or model?\n");
    });
    CODE("nn.w",  [this] { get_parm(0); });
    CODE("nn.b",  [this] { get_parm(1); });
    CODE("nn.dw", [this] { get_parm(2); });
    CODE("nn.db", [this] { get_parm(3); });
    CODE("nn.ex", [this] { get_parm(4); });
    CODE("nn.w=", [this] { set_parm(0); });
    CODE("nn.b=", [this] { set_parm(1); });
    CODE("flatten", [this] { nnop(T4K_L_FLATTEN); });
    auto pickle = [this](bool save) {                    // ( N adr len [mode] -- N )  model persistence is a "next" row
        if (SP() > 1 && IS_OBJ(SS(-2))) { /* ( N adr len ) */ }
        else if (SP() > 2 && IS_OBJ(SS(-3))) POPi();     // optional mode (raw formats: TODO in the reference too); netvm.cpp:139-141, in that order
        else { pstr("(model|tensor) adr len [mode]?\n"); return; }
        POPi(); const uint32_t adr = (uint32_t)POPi();
        const char *fn = (const char *)&pmem_[adr];
        if (!is_m(tos_)) return;
        hold_begin(); hold_ = true;                      // syscall(OP_NSAVE / OP_NLOAD), netvm.cpp:148-152
        if (save) model_save(MTOS(), fn); else model_load(MTOS(), fn);
    };
    CODE("save", [this, pickle] { if (is_m(SS(-2)) || (SP() > 2 && is_m(SS(-3)))) pickle(true); else { auto it = shadow_.find("save"); if (it != shadow_.end() && !it->second.empty()) it->second[0](); } });   // a tensor: the tensor vocabulary's word (the FIRST body this name had)
    CODE("load", [this, pickle] { if (is_m(SS(-2)) || (SP() > 2 && is_m(SS(-3)))) pickle(false); else { auto it = shadow_.find("load"); if (it != shadow_.end() && !it->second.empty()) it->second[0](); } });
    CODE("\nUser::", [] {});
    const int user0 = (int)dict_.size() - 1; user0_ = user0;
    CODE("boot", [this, user0] {   // mmu.clear(FIND("boot") + 1), eforth.cpp:420: the same pfa - strlen(name) as forget (mmu.h:95-98)
         if ((int)dict_.size() > user0 + 1) { if (dict_[user0 + 1].udf) here_ = dict_[user0 + 1].pfa - (uint32_t)dict_[user0 + 1].name.size(); dict_.resize(user0 + 1); } });
}

} // namespace t4
