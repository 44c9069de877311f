// vm.cpp - eForth core: outer interpreter, token-threaded inner interpreter, compiler words.
// Word set and stack effects: reference src/vm/eforth.cpp:155-431; number parsing :459-483;
// inner-interpreter opcodes :81-137; stack dump format src/debug.cpp:63-81.
#include "vm.h"
#include <chrono>
#include <sstream>
#include <thread>

namespace t4 {

static inline uint32_t ALIGN4(uint32_t v) { return (v + 3u) & ~3u; }
static double now_ms() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

VM::VM() : pmem_(PMEM_SZ, 0) { base() = 10; Dataset::trace = &trace_lvl; }        // (the dataset words print the reference's fetch text at trace level >= 1)
static void vm_sink(const char *t, void *u) { ((VM *)u)->host_msg(t); }
VM::~VM() {
    void (*fn)(const char *, void *); void *user; get_host_sink(&fn, &user);
    if (user == this) set_host_sink(nullptr, nullptr);
    if (Dataset::trace == &trace_lvl) Dataset::trace = nullptr;
}
namespace {
// host-layer messages (hprintf / chk) reach the VM that is EXECUTING: the sink is process-global, several VMs may be embedded
struct SinkScope {
    void (*fn)(const char *, void *); void *user;
    explicit SinkScope(VM *vm) { get_host_sink(&fn, &user); set_host_sink(vm_sink, vm); }
    ~SinkScope() { if (user) set_host_sink(fn, user); }      // an attached VM keeps the sink between lines; never restore "no sink" over a live VM
};
}

// ---------------------------------------------------------------- input
const char *VM::fetch() {
    while (pos_ < line_.size() && isspace((unsigned char)line_[pos_])) pos_++;
    if (pos_ >= line_.size()) return nullptr;
    size_t e = pos_;
    while (e < line_.size() && !isspace((unsigned char)line_[e])) e++;
    tok_ = line_.substr(pos_, e - pos_);
    pos_ = e;
    return tok_.c_str();
}
std::string VM::scan(char delim) {
    if (delim == '\n') { std::string s = pos_ < line_.size() ? line_.substr(pos_) : ""; pos_ = line_.size(); return s; }
    size_t e = line_.find(delim, pos_);
    std::string s = line_.substr(pos_, e == std::string::npos ? std::string::npos : e - pos_);
    pos_ = e == std::string::npos ? line_.size() : e + 1;
    return s;
}

// ---------------------------------------------------------------- dictionary / compiler
int VM::find(const char *name) {
    for (int i = (int)dict_.size() - 1; i > 0; --i) if (dict_[i].name == name) return i;
    return 0;
}
void VM::add(const char *name, std::function<void()> f, bool immd) {
    // a built-in defined again (the tensor / nn vocabularies redefine `@ max min relu tanh sigmoid normalize flatten save load boot`) REPLACES the entry in
    // place, as MMU::add_word does (mmu.h:68-93: "*** redefined"): dictionary indices - what `'` pushes, what mstat counts - stay the reference's.  The
    // replaced body stays callable by the new one (shadow_).
    if (const int w = find(name)) { shadow_[name].push_back(std::move(dict_[w].xt)); dict_[w].xt = std::move(f); dict_[w].immd = immd; return; }
    Word w; w.name = name; w.immd = immd; w.xt = std::move(f); dict_.push_back(std::move(w));
}
void VM::add_cell(uint32_t v) { if (here_ + 4 <= PMEM_SZ) { set_cell(here_, v); here_ += 4; } else pstr("pmem full\n"); }
void VM::add_du(DU d) { uint32_t u; memcpy(&u, &d, 4); add_cell(u); }
void VM::add_p(int op, uint32_t operand, bool udf, bool exit) {
    add_cell(((uint32_t)op << 24) | (udf ? 1u << 28 : 0) | (exit ? 1u << 31 : 0) | (operand & 0xFFFFFFu));   // struct Param vm/param.h:15-26: ioff 24 | op 4 | udf 1 | 2 | exit 1 - `dump` shows the reference's bytes
}
void VM::add_lit(DU v, bool exit) { add_p(P_LIT, 0, false, exit); add_du(v); }
void VM::add_w(int w) { Word &c = dict_[w]; add_p(P_WORD, c.udf ? c.pfa : (uint32_t)w, c.udf); }
int VM::add_str(const std::string &s) {
    uint32_t sz = ALIGN4((uint32_t)s.size() + 1);
    if (here_ + sz > PMEM_SZ) return 0;
    memset(&pmem_[here_], 0, sz); memcpy(&pmem_[here_], s.data(), s.size());
    here_ += sz;
    return (int)sz;
}
bool VM::new_word() {
    const char *name = fetch();
    if (!name || !*name) { pstr(" name?\n"); return false; }
    if (find(name)) { pstr(name); pstr(" reDef? \n"); }
    Word w; w.name = name; w.udf = true;
    here_ = ALIGN4(here_);
    add_str(w.name);                                     // the name lives in front of the parameter field (MMU::colon mmu.cu:148-159): `here`, `see` and `dump` show the reference's addresses
    w.pfa = here_;
    dict_.push_back(std::move(w));
    return true;
}

// ---------------------------------------------------------------- inner interpreter
void VM::ds_next(uint32_t target) {                      // ForthVM::_ds_next eforth.cpp:614-634
    Obj &m = st().du2obj(tos_);
    if (m.type != T_MODEL) { pstr("TOS is not a network model?\n"); return; }
    Obj &d = st().du2obj(RS(-1));
    if (d.type != T_DATASET) { pstr("RTOS is not a dataset?\n"); return; }
    Dataset &ds = (Dataset &)d;
    if (ds.done) { DU v = rs_pop(); DROP(v); ((Model &)m).tick(); }
    else { hold_begin(); ds.fetch(nullptr, false); ip_ = target; hold_end(); }   // serviced inline (reference: OP_FETCH + HOLD)
}
void VM::nest() {
    query_ = false;
    while (ip_ && !stop_) {
        const uint32_t ix = cell(ip_);
        const int op = (ix >> 24) & 15; const uint32_t ioff = ix & 0xFFFFFFu;
        const bool udf = (ix >> 28) & 1, exitf = (ix >> 31) & 1;
        ip_ += 4;
        switch (op) {
        case P_EXIT: ip_ = (uint32_t)rs_pop(); break;
        case P_NEXT:
            if (IS_OBJ(tos_) && !rs_.empty() && IS_OBJ(RS(-1))) ds_next(ioff);
            else if (((RS(-1) -= 1.0f) - (-1.0f)) > DU_EPS) ip_ = ioff;
            else rs_pop();
            break;
        case P_LOOP:
            if ((RS(-2) - (RS(-1) += 1.0f)) > DU_EPS) ip_ = ioff;
            else { rs_pop(); rs_pop(); }
            break;
        case P_LIT:
            ss_.push_back(tos_); tos_ = DUP(mem_du(ip_)); ip_ += 4;
            if (exitf) ip_ = (uint32_t)rs_pop();
            break;
        case P_VAR:
            PUSH((DU)ALIGN4(ip_));
            if (ioff) ip_ = ioff; else ip_ = (uint32_t)rs_pop();
            break;
        case P_STR:  PUSH((DU)ip_); PUSH((DU)ioff); ip_ += ioff; break;
        case P_DOTQ: pstr((const char *)&pmem_[ip_]); ip_ += ioff; break;
        case P_BRAN: ip_ = ioff; break;
        case P_ZBRAN: if (ZEQ(POP())) ip_ = ioff; break;
        case P_FOR:  rs_.push_back(POP()); break;
        case P_DO:   { DU lim = ss_pop(); rs_.push_back(lim); rs_.push_back(POP()); } break;
        case P_KEY:  PUSH((DU)getchar()); break;
        default:
            if (udf) { rs_.push_back((DU)ip_); ip_ = ioff; }
            else { const size_t h0 = holds_; dict_[ioff].xt(); if (holds_ != h0) msg_pos_ = out_.size(); }   // (a word that was SERVICED: the reference flushes the VM's text there - later diagnostics follow it; words behind it in the same
                                                                                                              //  colon word print into the buffer again, host messages go in front of that - the reference's printf / fout interleaving)
        }
    }
}
void VM::call(int w) {
    Word &c = dict_[w];
    if (c.udf) { rs_.push_back((DU)ip_); ip_ = c.pfa; nest(); }
    else { const size_t h0 = holds_; c.xt(); if (holds_ != h0) msg_pos_ = out_.size(); }
}

// ---------------------------------------------------------------- outer interpreter
DU VM::number(const char *idiom, bool &ok) {             // ForthVM::number eforth.cpp:459-483
    int b = base();
    switch (*idiom) {
    case '%': b = 2; idiom++; break;
    case '&': case '#': b = 10; idiom++; break;
    case '$': b = 16; idiom++; break;
    }
    char *p = nullptr;
    double d = (b == 10 && strchr(idiom, '.')) ? strtof(idiom, &p) : (double)strtol(idiom, &p, b);
    ok = (*idiom != '\0') && p && *p == '\0';
    return (DU)d;
}
int VM::process(const char *idiom) {                     // TensorVM::process tenvm.cpp:16-40
    query_ = true;
    int w = find(idiom);
    if (w) {
        if (compile_ && !dict_[w].immd) add_w(w);
        else { ip_ = 0; call(w); }
        return 1;
    }
    bool ok; DU n = number(idiom, ok);
    if (!ok) {                                           // ForthVM::number eforth.cpp:470-473 (ERROR = printf: in front of the line's buffered text, as there)
        const char *t = idiom; int b = base();
        switch (*t) { case '%': b = 2; t++; break; case '&': case '#': b = 10; t++; break; case '$': b = 16; t++; break; }
        hprintf(" number(%s) base=%d => error\n", t, b);
        return 0;
    }
    n = SCALAR(n);
    if (compile_) add_lit(n);
    else if (ten_lvl_ > 0) {                             // literal goes into the tensor on TOS
        if (ten_stage_.size() <= ten_off_ - ten_base_) ten_stage_.resize(ten_off_ - ten_base_ + 1);
        ten_stage_[ten_off_ - ten_base_] = n; ten_off_++;
    } else PUSH(n);
    return 1;
}
bool VM::eval(const std::string &line) {
    SinkScope sink(this);
    line_ = line; pos_ = 0;
    msg_pos_ = out_.size();                              // (the previous line's output has been flushed)
    const char *idiom;
    while (!stop_ && (idiom = fetch()) != nullptr) {
        std::string tk = idiom;
        if (!process(tk.c_str())) {
            pstr(tk); pstr("? \n");
            compile_ = false; pos_ = line_.size();
            break;
        }
        if (hold_) {                                     // VM::outer `if (state==HOLD) break;` + the cleared input buffer
            static const bool warn = getenv("T4_HOLD_WARN") != nullptr;   // script hygiene (tests): name what the reference would drop too
            if (warn) { size_t p = line_.find_first_not_of(" \t", pos_); if (p != std::string::npos && line_[p] != '\\') fprintf(stderr, "HOLD drops: %s\n", line_.c_str() + p); }
            hold_ = false; pos_ = line_.size(); break;
        }
    }
    hold_ = false;
    if (!compile_) ss_dump();                            // ForthVM::post eforth.cpp:67-71 (after `bye` too)
    st().sweep();                                        // objects marked by `.` are released once per line
    return !stop_;
}

// ---------------------------------------------------------------- output
void VM::dot_obj(DU v) {
    Obj &o = st().du2obj(v);
    if (o.type == T_MODEL) pstr(fmt_model((Model &)o)); else pstr(fmt_tensor((Tensor &)o));
}
void VM::dot(DU v) {
    if (IS_OBJ(v)) { hold_begin(); dot_obj(v); pstr(" "); st().mark_free(v); hold_end(); return; }   // ForthVM::_print eforth.cpp:559-567
    char buf[48]; snprintf(buf, sizeof(buf), "%g", v);   // ostream << float, default precision 6
    std::string s = buf;
    if (fmt_w_ > (int)s.size()) s = std::string(fmt_w_ - s.size(), ' ') + s;
    fmt_w_ = 0;
    pstr(s); pstr(" ");
}
void VM::ss_dump() {                                     // Debug::ss_dump debug.cpp:63-81
    auto show = [&](DU v) {
        if (IS_OBJ(v)) pstr(fmt_objname(st().du2obj(v), IS_VIEW(v))); else pstr(fmt_scalar(v, base()));
        pstr(" ");
    };
    for (DU v : ss_) show(v);
    show(tos_);
    pstr("-> ok\n");
}
void VM::words() {
    int sz = 0;
    for (auto &w : dict_) {
        pstr("  "); pstr(w.name);
        sz += w.name[0] == '\n' ? 72 : (int)w.name.size() + 2;
        if (sz >= 72) { pstr("\n"); sz = 0; }
    }
    pstr("\n");
}
void VM::see(int w) {                                    // Debug::see / _see debug.cpp:136-166,206-249
    Word &c = dict_[w];
    pstr(": "); pstr(c.name); pstr("\n");
    if (!c.udf) { pstr(" ( built-ins ) ;\n"); return; }
    static const char *pn[] = {";", "next ", "loop ", "lit", "var", "str", "dotq", "bran ", "0bran", "for  ", "do", "key"};
    auto nfa = [&](int i) { return dict_[i].pfa - ALIGN4((uint32_t)dict_[i].name.size() + 1); };
    for (uint32_t a = c.pfa; a + 4 <= (uint32_t)PMEM_SZ;) {
        const uint32_t ix = cell(a); const int op = (ix >> 24) & 15; const uint32_t ioff = ix & 0xFFFFFFu; const bool udf = (ix >> 28) & 1, exitf = (ix >> 31) & 1;
        int idx = op;
        if (op == P_WORD) {
            idx = -1;
            if (udf) { for (int i = (int)dict_.size() - 1; i > 0; --i) if (dict_[i].udf && dict_[i].pfa == ioff) { idx = i; break; } }
            else if (ioff < dict_.size()) idx = (int)ioff;
            if (idx < 0) break;
        }
        char buf[64]; snprintf(buf, sizeof(buf), "  ( %04x[%3x] ) ", a, idx); pstr(buf);
        a += 4;
        bool done = false;
        if (op == P_WORD) { pstr(dict_[idx].name); pstr("  "); }
        else {
            switch (op) {
            case P_LIT:  pstr(fmt_scalar(mem_du(a), base())); break;
            case P_STR:  pstr("s\" "); pstr((const char *)&pmem_[a]); pstr("\""); break;
            case P_DOTQ: pstr(".\" "); pstr((const char *)&pmem_[a]); pstr("\""); break;
            case P_VAR: {
                const uint32_t end = ioff ? ioff : ((w + 1 < (int)dict_.size()) ? nfa(w + 1) : here_);
                for (uint32_t v = a; v + 4 <= end; v += 4) { char t[32]; snprintf(t, sizeof(t), "%g ", mem_du(v)); pstr(t); }
            }   /* falls through: the primitive's name follows the values */
            default: pstr(op < 12 ? pn[op] : "?"); break;
            }
            if (op == P_NEXT || op == P_LOOP || op == P_BRAN || op == P_ZBRAN) { snprintf(buf, sizeof(buf), " \\ $%04x", ioff); pstr(buf); }
            done = op == P_EXIT || (op == P_LIT && exitf) || (op == P_VAR && !ioff);
        }
        if (done) break;
        pstr("\n");
        switch (op) { case P_LIT: a += 4; break; case P_VAR: a = ioff; break; case P_STR: case P_DOTQ: a += ioff; break; default: break; }
    }
    pstr("\n");
}

// ---------------------------------------------------------------- scalar ALU (VM::xop1/xop2 vm.cpp:66-105)
void VM::sxop1(int op) {
    DU t = tos_;
    switch (op) {
    case T4K_ABS: t = fabsf(t); break;
    case T4K_NEG: t = -t; break;
    case T4K_EXP: t = expf(t); break;
    case T4K_LN:  t = t > DU_EPS ? logf(t) : 0.0f; break;
    case T4K_LOG: t = t > DU_EPS ? log10f(t) : 0.0f; break;
    case T4K_TANH: t = tanhf(t); break;
    case T4K_RELU: t = fmaxf(t, 0.0f); break;
    case T4K_SIGM: t = 1.0f / (1.0f + expf(-t)); break;
    case T4K_SQRT: t = sqrtf(t); break;
    case T4K_RCP: t = 1.0f / t; break;
    case T4K_SAT: t = fminf(1.0f, fmaxf(0.0f, t)); break;
    case T4K_SIN: t = sinf(t); break;
    case T4K_COS: t = cosf(t); break;
    default: pstr("method not supported: op=%d?\n"); break;
    }
    tos_ = SCALAR(t);
}
void VM::sxop2(int op) {
    DU t = tos_, n = ss_pop();
    switch (op) {
    case T4K_ADD: t = n + t; break;
    case T4K_MUL: t = n * t; break;
    case T4K_SUB: t = n - t; break;
    case T4K_DIV: t = n / t; break;
    case T4K_MOD: t = fmodf(n, t); break;
    case T4K_MAX: t = fmaxf(n, t); break;
    case T4K_MIN: t = fminf(n, t); break;
    case T4K_POW: t = powf(t, n); break;
    default: pstr("method not supported: op=%d?\n"); break;
    }
    tos_ = SCALAR(t);
}

// ---------------------------------------------------------------- core vocabulary
void VM::init_core() {
    auto CODE = [this](const char *n, std::function<void()> f) { add(n, std::move(f), false); };
    auto IMMD = [this](const char *n, std::function<void()> f) { add(n, std::move(f), true); };
    CODE("\nForth::", [] {});
    CODE("nop", [] {});
    // stack
    CODE("dup",  [this] { PUSH(DUP(tos_)); });
    CODE("drop", [this] { DROP(tos_); tos_ = ss_pop(); });
    CODE("over", [this] { DU v = DUP(SS(-1)); PUSH(v); });
    CODE("swap", [this] { DU n = ss_pop(); PUSH(n); });
    CODE("rot",  [this] { DU n = ss_pop(); DU m = ss_pop(); ss_.push_back(n); PUSH(m); });
    CODE("-rot", [this] { DU n = ss_pop(); DU m = ss_pop(); PUSH(m); PUSH(n); });
    CODE("pick", [this] { int i = (int)tos_; tos_ = DUP(SS(-i)); });
    CODE("nip",  [this] { ss_.pop_back(); });
    CODE("?dup", [this] { if (tos_ != 0.0f) PUSH(tos_); });
    CODE("2dup", [this] { DU v = DUP(SS(-1)); PUSH(v); v = DUP(SS(-1)); PUSH(v); });
    CODE("2drop", [this] { DU s = ss_pop(); DROP(s); DROP(tos_); tos_ = ss_pop(); });
    CODE("2over", [this] { DU v = DUP(SS(-3)); PUSH(v); v = DUP(SS(-3)); PUSH(v); });
    CODE("2swap", [this] { DU n = ss_pop(); DU m = ss_pop(); DU l = ss_pop();
                           ss_.push_back(n); PUSH(l); PUSH(m); });
    // arithmetic
    CODE("+", [this] { xop2(T4K_ADD, true); });
    CODE("-", [this] { xop2(T4K_SUB, true); });
    CODE("*", [this] { xop2(T4K_MUL, true); });
    CODE("/", [this] { xop2(T4K_DIV, true); });
    CODE("mod",  [this] { DU n = ss_pop(); DU m = (DU)((int)n % (int)tos_); tos_ = SCALAR(m); });
    CODE("fmod", [this] { DU n = ss_pop(); DU m = fmodf(n, tos_); tos_ = SCALAR(m); });
    CODE("/mod", [this] { DU n = ss_pop(); DU m = fmodf(n, tos_); ss_.push_back(m); DU v = n / tos_; tos_ = SCALAR(v); });
    CODE("*/",   [this] { double a = ss_pop(); double b = ss_pop(); DU v = (DU)(a * b / tos_); tos_ = SCALAR(v); });
    CODE("*/mod", [this] { double a = ss_pop(); double b = ss_pop(); double n2 = a * b;
                           DU m = (DU)fmod(n2, (double)tos_); ss_.push_back(SCALAR(m)); DU v = floorf((DU)(n2 / tos_)); tos_ = SCALAR(v); });
    CODE("and", [this] { DU n = ss_pop(); tos_ = (DU)((int)tos_ & (int)n); });
    CODE("or",  [this] { DU n = ss_pop(); tos_ = (DU)((int)tos_ | (int)n); });
    CODE("xor", [this] { DU n = ss_pop(); tos_ = (DU)((int)tos_ ^ (int)n); });
    CODE("abs", [this] { xop1(T4K_ABS); });
    CODE("negate", [this] { xop1(T4K_NEG); });
    CODE("invert", [this] { tos_ = (DU)(~(int)tos_); });
    CODE("rshift", [this] { DU n = ss_pop(); tos_ = (DU)((int)n >> (int)tos_); });
    CODE("lshift", [this] { DU n = ss_pop(); tos_ = (DU)((int)n << (int)tos_); });
    CODE("max", [this] { DU n = ss_pop(); tos_ = (tos_ > n) ? tos_ : n; });
    CODE("min", [this] { DU n = ss_pop(); tos_ = (tos_ < n) ? tos_ : n; });
    CODE("2*", [this] { tos_ *= 2.0f; });
    CODE("2/", [this] { tos_ /= 2.0f; });
    CODE("1+", [this] { tos_ += 1.0f; });
    CODE("1-", [this] { tos_ -= 1.0f; });
    CODE("f>s",   [this] { tos_ = (DU)(int)tos_; });
    CODE("round", [this] { tos_ = roundf(tos_); });
    CODE("ceil",  [this] { tos_ = ceilf(tos_); });
    CODE("floor", [this] { tos_ = floorf(tos_); });
    // logic (epsilon compares, booleans 0 / -1: ten4_types.h:85-90)
    CODE("0=", [this] { tos_ = BOOL(ZEQ(tos_)); });
    CODE("0<", [this] { tos_ = BOOL(tos_ < -DU_EPS); });
    CODE("0>", [this] { tos_ = BOOL(tos_ > DU_EPS); });
    CODE("=",  [this] { DU n = ss_pop(); tos_ = BOOL(ZEQ(n - tos_)); });
    CODE(">",  [this] { DU n = ss_pop(); tos_ = BOOL((n - tos_) > DU_EPS); });
    CODE("<",  [this] { DU n = ss_pop(); tos_ = BOOL((n - tos_) < -DU_EPS); });
    CODE("<>", [this] { DU n = ss_pop(); tos_ = BOOL(!ZEQ(n - tos_)); });
    CODE(">=", [this] { DU n = ss_pop(); tos_ = BOOL(!((n - tos_) < -DU_EPS)); });
    CODE("<=", [this] { DU n = ss_pop(); tos_ = BOOL(!((n - tos_) > DU_EPS)); });
    CODE("u<", [this] { DU n = ss_pop(); tos_ = BOOL((uint32_t)(int)n < (uint32_t)(int)tos_); });
    CODE("u>", [this] { DU n = ss_pop(); tos_ = BOOL((uint32_t)(int)n > (uint32_t)(int)tos_); });
    // io
    CODE("base",    [this] { PUSH(0.0f); });
    CODE("decimal", [this] { base() = 10; });
    CODE("hex",     [this] { base() = 16; });
    CODE("bl",      [this] { PUSH(32.0f); });
    CODE("cr",      [this] { pstr("\n"); });
    CODE(".",       [this] { dot(POP()); });
    CODE("u.",      [this] { char b[24]; snprintf(b, sizeof(b), "%u ", (uint32_t)(int)POP()); pstr(b); });
    CODE(".r",      [this] { fmt_w_ = POPi(); DU v = POP(); char b[48]; snprintf(b, sizeof(b), "%g", v); std::string s = b;
                             if (fmt_w_ > (int)s.size()) { s = std::string(fmt_w_ - s.size(), ' ') + s; }
                             fmt_w_ = 0; pstr(s); });
    CODE("u.r",     [this] { int w = POPi(); DU v = POP(); char b[48]; snprintf(b, sizeof(b), "%u", (uint32_t)v); std::string s = b;
                             if (w > (int)s.size()) { s = std::string(w - s.size(), ' ') + s; }
                             pstr(s); });
    CODE("type",    [this] { POP(); pstr((const char *)&pmem_[(uint32_t)POPi()]); });
    IMMD("key",     [this] { if (compile_) add_p(P_KEY); else PUSH((DU)getchar()); });
    CODE("emit",    [this] { char c = (char)(int)POP(); out_.push_back(c); });
    CODE("space",   [this] { pstr(" "); });
    CODE("spaces",  [this] { int n = POPi(); if (n > 0) pstr(std::string(n, ' ')); });
    // literals
    IMMD("(",   [this] { scan(')'); });
    IMMD(".(",  [this] { pstr(scan(')')); });
    IMMD("\\",  [this] { scan('\n'); });
    auto quote = [this](int op) {
        std::string s = scan('"');
        if (!s.empty() && s[0] == ' ') s = s.substr(1);
        if (compile_) { add_p(op, ALIGN4((uint32_t)s.size() + 1)); add_str(s); }
        else {
            uint32_t h0 = here_; int len = add_str(s);
            if (op == P_STR) { PUSH((DU)h0); PUSH((DU)len); } else pstr((const char *)&pmem_[h0]);
            here_ = h0;
        }
    };
    IMMD("s\"", [quote] { quote(P_STR); });
    IMMD(".\"", [quote] { quote(P_DOTQ); });
    // branching / loops
    IMMD("if",    [this] { PUSH((DU)here_); add_p(P_ZBRAN); });
    IMMD("else",  [this] { uint32_t h = here_; add_p(P_BRAN); setjmp_at((uint32_t)POPi()); PUSH((DU)h); });
    IMMD("then",  [this] { setjmp_at((uint32_t)POPi()); });
    IMMD("begin", [this] { PUSH((DU)here_); });
    IMMD("again", [this] { add_p(P_BRAN, (uint32_t)POPi()); });
    IMMD("until", [this] { add_p(P_ZBRAN, (uint32_t)POPi()); });
    IMMD("while", [this] { PUSH((DU)here_); add_p(P_ZBRAN); });
    IMMD("repeat", [this] { uint32_t t = (uint32_t)POPi(); add_p(P_BRAN, (uint32_t)POPi()); setjmp_at(t); });
    IMMD("for",   [this] { add_p(P_FOR); PUSH((DU)here_); });
    IMMD("next",  [this] { add_p(P_NEXT, (uint32_t)POPi()); });
    IMMD("aft",   [this] { POP(); uint32_t h = here_; add_p(P_BRAN); PUSH((DU)here_); PUSH((DU)h); });
    IMMD("do",    [this] { add_p(P_DO); PUSH((DU)here_); });
    CODE("i",     [this] { PUSH(RS(-1)); });
    CODE("leave", [this] { rs_pop(); rs_pop(); ip_ = (uint32_t)rs_pop(); });
    IMMD("loop",  [this] { add_p(P_LOOP, (uint32_t)POPi()); });
    CODE(">r", [this] { rs_.push_back(POP()); });
    CODE("r>", [this] { PUSH(rs_pop()); });
    CODE("r@", [this] { PUSH(DUP(RS(-1))); });
    // compiler
    CODE("[", [this] { compile_ = false; });
    CODE("]", [this] { compile_ = true; });
    CODE(":", [this] { compile_ = new_word(); });
    IMMD(";", [this] { add_p(P_EXIT); compile_ = false; });
    CODE("variable", [this] { if (!new_word()) return; add_p(P_VAR, 0, true); add_du(0.0f); });
    CODE("constant", [this] { if (!new_word()) return; add_lit(POP(), true); });
    CODE("value",    [this] { if (!new_word()) return; add_p(P_LIT, 0, true, true); add_du(POP()); });
    IMMD("immediate", [this] { dict_.back().immd = true; });
    CODE("exit",   [this] { ip_ = (uint32_t)rs_pop(); });
    CODE("exec",   [this] { int w = (int)POP(); call(w); });
    CODE("create", [this] { if (!new_word()) return; add_p(P_VAR, 0, true); });
    CODE("does>",  [this] {
        uint32_t pfa = dict_.back().pfa;
        while (((cell(pfa) >> 24) & 15) != P_VAR && pfa < here_) pfa += 4;
        setjmp_at(pfa);
        add_p(P_BRAN, ip_); ip_ = (uint32_t)rs_pop();
    });
    IMMD("to", [this] {
        int w = query_ ? find(fetch() ? tok_.c_str() : "") : POPi();
        if (!w) return;
        if (compile_) { add_lit((DU)w); add_w(find("to")); }
        else { uint32_t pfa = dict_[w].pfa; if (((cell(pfa) >> 24) & 15) == P_LIT) set_du(pfa + 4, POP()); }
    });
    IMMD("is", [this] {
        int w = query_ ? find(fetch() ? tok_.c_str() : "") : POPi();
        if (!w) return;
        if (compile_) { add_lit((DU)w); add_w(find("is")); }
        else { int d = POPi(); dict_[d].xt = dict_[w].xt; dict_[d].udf = dict_[w].udf; dict_[d].pfa = dict_[w].pfa; }
    });
    CODE("[to]", [this] {
        uint32_t a = (cell(ip_) & 0xFFFFFFu) + 4; DU d = POP(); ip_ += 4;
        if (a + 4 <= (uint32_t)PMEM_SZ) set_du(a, d); else { pstr("is ?"); stop_ = true; }
    });
    // memory
    CODE("@",  [this] { uint32_t i = (uint32_t)POPi(); PUSH(mem_du(i)); });
    CODE("!",  [this] { uint32_t i = (uint32_t)POPi(); set_du(i, POP()); });
    CODE("c@", [this] { uint32_t i = (uint32_t)POPi(); PUSH((DU)pmem_[i]); });
    CODE("c!", [this] { uint32_t i = (uint32_t)POPi(); pmem_[i] = (uint8_t)POPi(); });
    CODE("+!", [this] { uint32_t i = (uint32_t)POPi(); DU v = mem_du(i) + POP(); set_du(i, SCALAR(v)); });
    CODE("?",  [this] { uint32_t i = (uint32_t)POPi(); dot(mem_du(i)); });
    CODE(",",  [this] { add_du(POP()); });
    CODE("cells", [this] { int i = POPi(); PUSH((DU)(i * 4)); });
    CODE("allot", [this] { int n = POPi(); for (int i = 0; i < n; i += 4) add_du(0.0f); });
    CODE("th",    [this] { int i = POPi(); tos_ += i * 4; });
    // debug / os
    CODE("abort", [this] { tos_ = -1.0f; ss_.clear(); rs_.clear(); });
    CODE("here",  [this] { PUSH((DU)here_); });
    CODE("'",     [this] { const char *n = fetch(); int w = n ? find(n) : 0; if (w) PUSH((DU)w); });
    CODE(".s",    [this] { ss_dump(); });
    CODE("depth", [this] { PUSH((DU)((int)SP() - 1)); });
    CODE("words", [this] { words(); });
    CODE("dict",  [this] { words(); });
    CODE("dict_dump", [this] { words(); });
    CODE("see",   [this] { const char *n = fetch(); int w = n ? find(n) : 0; if (w) see(w); });
    CODE("dump",  [this] {                                  // Debug::mem_dump debug.cpp:110-132: 16 bytes per line, hex in groups of four, then the 7-bit characters
        int n = POPi(); uint32_t a = (uint32_t)POP();
        auto al16 = [](uint32_t v) { return v + ((0u - v) & 0xFu); };
        for (uint32_t i = al16(a); i <= al16(a + n) && i + 16 <= PMEM_SZ; i += 16) {
            char b[96]; int x = snprintf(b, sizeof(b), "%04x: ", i);
            char asc[17];
            for (int j = 0; j < 16; j++) {
                const uint8_t c = pmem_[i + j], c7 = c & 0x7f;
                x += snprintf(b + x, sizeof(b) - x, "%02x %s", c, (j % 4 == 3) ? " " : "");
                asc[j] = (c7 == 0x7f || c7 < 0x20) ? '.' : (char)c7;
            }
            asc[16] = 0;
            pstr(b); pstr(asc); pstr("\n");
        }
    });
    CODE("forget", [this] {
        const char *n = fetch(); int w = n ? find(n) : 0; if (!w) return;
        int b = user0_ + 1; if (w < b) w = b;
        if (w >= (int)dict_.size()) return;                   // a built-in with no user word defined yet: nothing to clear                 // (the reference clears from its eForth `boot` entry on, vocabularies and all: eforth.cpp:504-506)
        if (dict_[w].udf) here_ = dict_[w].pfa - (uint32_t)dict_[w].name.size();   // MMU::clear mmu.h:95-98 (name length, unaligned)
        dict_.resize(w);
    });
    CODE("trace", [this] { trace_lvl = POPi(); });
    CODE("mstat", [this] {
        char b[240]; snprintf(b, sizeof(b), "\\ MMU.stat dict[%d/1024], pmem[%d]=%0.1f%%, obj#used[%d], HBM used=%zu KiB in %zu blocks (peak %zu KiB, %zu free block(s), %zu slab(s))\n",
                              (int)dict_.size(), here_, 100.0 * here_ / PMEM_SZ, st().live(), Arena::get().used() >> 10, Arena::get().live(),
                              Arena::get().peak() >> 10, Arena::get().free_blocks(), Arena::get().slabs());
        hprintf("%s", b);                                  // MMU::status is a plain printf (mmu.cu:124-128): in front of the line's buffered text
    });
    CODE("ms",    [this] { std::this_thread::sleep_for(std::chrono::milliseconds(POPi())); });
    CODE("flush", [this] { fflush(stdout); hold_end(); });
    CODE("sprintf", [this] {                             // ( n1 [n2 ..] addr u -- addr' u' )  eforth.cpp:576-611
        POPi(); std::string buf = (const char *)&pmem_[(uint32_t)POP()];
        auto t2s = [this](char c) {
            std::ostringstream n;
            switch (c) {
            case 'd': n << (uint32_t)POP(); break;
            case 'g': case 'f': n << (DU)POP(); break;
            case 'x': n << "0x" << std::hex << (uint32_t)POP(); break;
            case 's': POP(); n << (const char *)&pmem_[(uint32_t)POP()]; break;
            default: n << c << '?'; break;
            }
            return n.str();
        };
        for (size_t i = buf.find_last_of('%'); i != std::string::npos; i = buf.find_last_of('%', i ? i - 1 : 0)) {
            if (i && buf[i - 1] == '%') buf.replace(--i, 1, "");
            else buf.replace(i, 2, t2s(buf[i + 1]));
            if (i == 0) break;
        }
        uint32_t h0 = here_; int len = add_str(buf);
        PUSH((DU)h0); PUSH((DU)len);
        here_ = h0;
    });
    CODE("clock", [this] { DU t = (DU)now_ms(); PUSH(SCALAR(t)); });
    CODE("bye",   [this] { stop_ = true; });
    CODE("boot",  [this] { int b = find("boot") + 1; if ((int)dict_.size() > b) { if (dict_[b].udf) here_ = dict_[b].pfa; dict_.resize(b); } });
}

void VM::init() {
    set_host_sink(vm_sink, this);                        // host-layer messages join this VM's output, in order (re-pointed at every eval)
    dict_.clear();
    init_core();
    init_tensor();
    init_nn();
}

} // namespace t4
