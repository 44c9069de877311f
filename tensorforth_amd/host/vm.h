// vm.h - eForth virtual machine (scalar core + tensor words + nn words) for the MI355X backend.
// The word set and stack effects are those of the reference (src/vm/eforth.cpp:155-431,
// tenvm.cpp:450-636, netvm.cpp:291-485); the implementation is a fresh token-threaded VM that
// services dataset / print requests inline instead of the reference's HOLD + event-queue protocol.
#pragma once
#include "t4.h"
#include <map>

namespace t4 {

enum Prim { P_EXIT = 0, P_NEXT, P_LOOP, P_LIT, P_VAR, P_STR, P_DOTQ, P_BRAN, P_ZBRAN, P_FOR, P_DO, P_KEY, P_WORD = 15 };

struct Word {
    std::string name;
    bool immd = false, udf = false;
    std::function<void()> xt;
    uint32_t pfa = 0;
};

class VM {
public:
    VM();
    ~VM();                                    // detaches the host-message sink if it still points here
    void init();
    // feed one line of Forth source; returns false after `bye`
    bool eval(const std::string &line);
    std::string take_output() { std::string s; s.swap(out_); msg_pos_ = 0; return s; }
    bool done() const { return stop_; }
    Tensor *tos_tensor() { return TOS1T() ? &TTOS() : nullptr; }   // embedding API: the tensor on top of the data stack (ten4_fetch)
    // Host-layer diagnostics (hprintf: the reference's INFO / ERROR, plain printf).  The reference's VM text is BUFFERED (events rendered by System::flush at the
    // end of the line or at a HOLD, sys.cpp:110-120) while those printf calls write at once: on one input line every such message comes out IN FRONT of what the
    // VM words of that line printed since the last flush.  msg_pos_ = where the unflushed part of this VM's output begins.
    void host_msg(const char *t) { const size_t at = msg_pos_ <= out_.size() ? msg_pos_ : out_.size(); const size_t n = strlen(t); out_.insert(at, t); msg_pos_ = at + n; }
    int  trace_lvl = 1;                       // T4_VERBOSE default, `trace` word

private:
    static constexpr int PMEM_SZ = 48 * 1024;
    std::vector<Word> dict_;
    std::map<std::string, std::vector<std::function<void()>>> shadow_;   // bodies replaced by redefinitions of a built-in (VM::add), oldest first: a redefining body calls the one it replaced by its INDEX (a look-up of "the latest" would find itself after a third definition)
    int user0_ = 0;                           // index of the `User::` marker: user words start behind it
    std::vector<DU> ss_, rs_;
    DU tos_ = -1.0f;
    std::vector<uint8_t> pmem_;
    uint32_t here_ = 16;                      // user area: pmem[0] = base
    uint32_t ip_ = 0;
    bool compile_ = false, stop_ = false, query_ = true;
    size_t msg_pos_ = 0;
    void hold_begin() { msg_pos_ = out_.size(); }          // a host service starts: everything buffered so far is flushed first, the service's own messages follow it
    void hold_end() { hold_ = true; holds_++; msg_pos_ = out_.size(); }
    size_t holds_ = 0;                                    // host services so far (a word that ended in one is where the reference flushes)
    bool hold_ = false;                       // a word asked for host service (reference: state = HOLD, eforth.h:85-92): the outer interpreter
                                              // drops the rest of the input line (sys.cpp:101-108 clears the buffer after resume(), vm.cpp:59)
    std::string line_; size_t pos_ = 0;
    std::string out_;
    int fmt_w_ = 0;
    // tensor literal mode
    uint32_t ten_off_ = 0; int ten_lvl_ = 0;
    std::vector<float> ten_stage_; uint32_t ten_base_ = 0;

    // ---- helpers
    Store &st() { return Store::get(); }
    uint8_t &base() { return pmem_[0]; }
    DU POP() { DU n = tos_; if (ss_.empty()) tos_ = -1.0f; else { tos_ = ss_.back(); ss_.pop_back(); } return n; }
    DU PUSH(DU v) { ss_.push_back(tos_); return tos_ = v; }
    DU PUSH(Obj &o) { ss_.push_back(tos_); return tos_ = st().obj2du(o); }
    int POPi() { return (int)POP(); }
    DU &SS(int i) { long k = (long)ss_.size() + i; if (k < 0 || k >= (long)ss_.size()) { dummy_ = 0; return dummy_; } return ss_[k]; }   // SS(-1) = NOS
    DU &RS(int i) { long k = (long)rs_.size() + i; if (k < 0 || k >= (long)rs_.size()) { dummy_ = 0; return dummy_; } return rs_[k]; }
    DU dummy_ = 0;
    DU ss_pop() { if (ss_.empty()) return 0; DU v = ss_.back(); ss_.pop_back(); return v; }
    size_t SP() const { return ss_.size(); }
    DU DUP(DU d) { return IS_OBJ(d) ? AS_VIEW(d) : d; }
    void DROP(DU d) { if (IS_OBJ(d) && !IS_VIEW(d)) st().drop(st().du2obj(d)); }
    DU rs_pop() { if (rs_.empty()) return 0; DU v = rs_.back(); rs_.pop_back(); return v; }

    Tensor &TTOS() { return (Tensor &)st().du2obj(tos_); }
    Tensor &TNOS() { return (Tensor &)st().du2obj(SS(-1)); }
    Model  &MTOS() { return (Model &)st().du2obj(tos_); }
    Model  &MNOS() { return (Model &)st().du2obj(SS(-1)); }
    bool is_t(DU v) { return IS_OBJ(v) && st().du2obj(v).type == T_TENSOR; }
    bool is_m(DU v) { return IS_OBJ(v) && st().du2obj(v).type == T_MODEL; }
    bool is_d(DU v) { return IS_OBJ(v) && st().du2obj(v).type == T_DATASET; }
    bool TOS1T() { return is_t(tos_); }
    bool TOS2T() { return SP() > 0 && TOS1T() && is_t(SS(-1)); }
    bool TOS3T() { return SP() > 1 && TOS2T() && is_t(SS(-2)); }
    bool TOS1D() { return IS_OBJ(tos_) && (is_t(tos_) || is_d(tos_)); }
    bool M1V() { return SP() > 0 && !IS_OBJ(tos_) && is_m(SS(-1)); }
    bool M2V() { return SP() > 1 && !IS_OBJ(tos_) && !IS_OBJ(SS(-1)) && is_m(SS(-2)); }
    bool MTV() { return SP() > 1 && !IS_OBJ(tos_) && IS_OBJ(SS(-1)) && is_m(SS(-2)); }

    // ---- input
    const char *fetch();                      // next blank-delimited token or nullptr
    std::string scan(char delim);
    std::string tok_;
    // ---- dictionary / compiler
    int  find(const char *name);
    void add(const char *name, std::function<void()> f, bool immd = false);
    void add_cell(uint32_t v);
    void add_du(DU d);
    void add_p(int op, uint32_t operand = 0, bool udf = false, bool exit = false);
    void add_lit(DU v, bool exit = false);
    void add_w(int w);
    int  add_str(const std::string &s);
    uint32_t cell(uint32_t a) { uint32_t v; memcpy(&v, &pmem_[a], 4); return v; }
    void set_cell(uint32_t a, uint32_t v) { memcpy(&pmem_[a], &v, 4); }
    DU   mem_du(uint32_t a) { DU v = 0; if ((uint64_t)a + 4 <= pmem_.size()) memcpy(&v, &pmem_[a], 4); return v; }   // user addresses (`@ ? +!`): a 4-byte access must end inside pmem
    void set_du(uint32_t a, DU v) { if ((uint64_t)a + 4 <= pmem_.size()) memcpy(&pmem_[a], &v, 4); }
    void setjmp_at(uint32_t a) { set_cell(a, (cell(a) & 0xFF000000u) | (here_ & 0xFFFFFFu)); }
    bool new_word();
    // ---- execution
    void nest();
    void call(int w);
    int  process(const char *idiom);
    DU   number(const char *idiom, bool &ok);
    void ds_next(uint32_t target);
    // ---- output
    void pstr(const std::string &s) { out_ += s; }
    void dot(DU v);
    void dot_obj(DU v);
    void ss_dump();
    void see(int w);
    void words();
    // ---- op groups
    void xop1(int op, DU v = 0);
    void xop2(int op, bool keep);
    void sxop1(int op);
    void sxop2(int op);
    void blas1(int op);
    void blas2(int op, bool keep);
    void gemm(int opt);
    void nnop(int layer);
    void conv(uint16_t k, bool txn = false, uint16_t s = 1, uint16_t p = 1, uint16_t d = 1);
    void loss(Loss op);
    void get_parm(int n);
    void set_parm(int n);
    void tboard(int op);
    void init_core();
    void init_tensor();
    void init_nn();
};

} // namespace t4
