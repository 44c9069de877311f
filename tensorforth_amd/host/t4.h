// t4.h - host side of the MI355X tensorForth backend: object model, HBM arena, Tensor / Model /
// Dataset classes and the eForth VM, all written against the C-ABI of include/t4k.h only.
//
// This restates (does not copy) the reference's host layer so that existing .4th scripts run
// unchanged: word tables src/vm/eforth.cpp:155-431, tenvm.cpp:450-636, netvm.cpp:291-485;
// Tensor/Model semantics src/mu/tensor.cu, src/nn/{model.cpp,forward.cu,backprop.cu,
// gradient.cu,loss.cpp}; print formats src/io/aio_tensor.cpp, aio_model.cpp, src/debug.cpp.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>
#include "../../include/t4k.h"

namespace t4 {
// Environment switches of the host (the documented list: DESIGN.md section 9) are read through these two helpers
inline bool env_flag(const char *name, bool dflt) { const char *e = getenv(name); return e ? atoi(e) != 0 : dflt; }
inline long env_long(const char *name, long dflt) { const char *e = getenv(name); return e ? atol(e) : dflt; }

typedef float DU;
constexpr DU DU_EPS = 1.0e-6f;

// ---- stack cell tagging (reference src/t4base.h:16-35): bit0 = object, bits0-1 == 3 = view
inline uint32_t du_bits(DU v) { uint32_t u; memcpy(&u, &v, 4); return u; }
inline DU bits_du(uint32_t u) { DU v; memcpy(&v, &u, 4); return v; }
inline bool IS_OBJ(DU v)  { return (du_bits(v) & 1u) != 0; }
inline bool IS_VIEW(DU v) { return (du_bits(v) & 3u) == 3u; }
inline DU   SCALAR(DU v)  { return bits_du(du_bits(v) & ~1u); }        // clears the mantissa LSB
inline DU   AS_VIEW(DU v) { return bits_du(du_bits(v) | 3u); }
inline bool ZEQ(DU d) { return fabsf(d) < DU_EPS; }
inline DU   BOOL(bool f) { return f ? -1.0f : 0.0f; }

enum ObjType { T_TENSOR = 0, T_MODEL = 1, T_DATASET = 2 };
enum Loss { LOSS_MSE = 0, LOSS_BCE, LOSS_CE, LOSS_NLL };
enum Optim { OPTI_SGD = 0, OPTI_SGDM, OPTI_ADAM, OPTI_ADAMW };
extern const char *LAYER_NAME[];     // 7-char names, src/nn/ntypes.h:47-51

void die_if_no_backend();            // lazily t4k_init(); prints and exits when no gfx950 device
int  chk(int rc, const char *what);  // prints t4k_last_error() on failure (print-and-continue)
// Host-layer diagnostics (model / tensor / dataset messages): printed through the VM's output buffer when a VM is attached, so that
// embedded users (ten4_eval / ten4_output, vm.py) see them and they stay in order with the text the words print; stdout otherwise.
void hprintf(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void hputs(const std::string &text);                     // the same sink, unformatted, any length
void set_host_sink(void (*fn)(const char *text, void *user), void *user);
void get_host_sink(void (**fn)(const char *text, void *user), void **user);
// TensorBoard sink (host/tboard.cpp): inactive - the words only print the reference's hint - until a log directory is given
struct Tensor;
bool tb_configure(const char *logdir, const char *run_id);
bool tb_active();
void tb_init(const char *run_id);
void tb_step(int i);
struct Model;
void tb_graph(Model &m);
void tb_embed(const char *tag, Tensor &t);
void tb_scalar(const char *tag, float v);
void tb_text(const char *tag, const char *txt);
void tb_histo(const char *tag, Tensor &t, int n_bucket);
void tb_tile(const char *tag, Tensor &t, int per_row);
void tb_image(const char *tag, Tensor &t);
void tb_close();
t4k_stream_t stream();

// ---------------------------------------------------------------- HBM arena
// One (growable) slab of device memory, out-of-band metadata on the host, 256-byte aligned blocks.
// Replaces the reference's host-run TLSF over a 2 GiB cudaMallocManaged arena (src/mu/mmu.cu:44-46,
// tlsf.cpp), whose in-band headers made the host touch device pages on every alloc/free.
class Arena {
public:
    static Arena &get();
    float *alloc(size_t nfloat);
    void   free(float *p);
    size_t used() const { return used_; }
    size_t live() const { return blocks_.size(); }
    size_t peak() const { return peak_; }
    size_t slabs() const { return slabs_.size(); }
    size_t free_blocks() const { return free_.size(); }
private:
    struct Slab { char *base; size_t size; };
    std::vector<Slab> slabs_;
    std::map<char *, size_t> free_;       // address -> bytes (coalesced)
    std::map<char *, size_t> blocks_;     // live allocations
    std::set<std::pair<size_t, char *>> by_size_;   // the free blocks again, ordered by (bytes, address): best fit
    size_t used_ = 0, peak_ = 0;
    void put_free(char *p, size_t sz);
    void take_free(std::map<char *, size_t>::iterator it);
    void add_slab(size_t need);
};

struct Obj {
    ObjType type = T_TENSOR;
    int id = -1;
    virtual ~Obj() {}
};

// ---------------------------------------------------------------- Tensor
struct Tensor : Obj {
    uint64_t numel = 0;
    int rank = 1;
    uint32_t shape[4] = {1, 1, 1, 1};        // H, W, C, N  (reference src/mu/tensor.h:53)
    uint16_t stride[4] = {1, 1, 1, 1};       // conv: S,S,P,P ; pool: k
    int   grad_fn = 0;                       // t4k layer enum
    Tensor *grad[5] = {0, 0, 0, 0, 0};       // w, b, dw, db, aux(mask | dx | xhat)
    Tensor *mtum[5] = {0, 0, 0, 0, 0};       // m_w, m_b, v_w, v_b, batchnorm stats
    int   iparm = 0;
    DU    xparm = 0;
    float *data = nullptr;                   // device pointer (HBM)
    bool  owns = true;
    struct Model *stale_owner = nullptr;     // the data is a dX the owner's backward skipped: Store::du2obj has the owner produce it before any word reads it

    uint32_t &H() { return shape[0]; }
    uint32_t &W() { return shape[1]; }
    uint32_t &C() { return shape[2]; }
    uint32_t &N() { return shape[3]; }
    uint64_t HWC() const { return (uint64_t)shape[0] * shape[1] * shape[2]; }
    float *slice(int n) { return data + HWC() * n; }
    bool same_shape(const Tensor &t) const { return memcmp(shape, t.shape, sizeof(shape)) == 0; }

    Tensor &reshape(uint64_t sz);
    Tensor &reshape(uint32_t h, uint32_t w);
    Tensor &reshape(uint32_t n, uint32_t h, uint32_t w, uint32_t c);

    // device ops (all asynchronous on the library stream)
    Tensor &zeros();
    Tensor &map(int op, DU v = 0);
    Tensor &identity();
    Tensor &normalize(DU avg, DU std);
    Tensor &operator=(Tensor &t);            // copy elements (k_copy)
    DU sum(); DU avg(); DU std(); DU norm(); DU max(); DU min();
    DU dot(Tensor &B);
    DU loss(Loss op, Tensor &tgt);           // destructive, as the reference's Tensor::loss
    DU det();
    uint32_t has_nan();
    // host access (synchronises)
    void to_host(std::vector<float> &h, uint64_t n = 0);
    void from_host(const float *h, uint64_t n, uint64_t off = 0);
    DU   get(uint64_t i);
    void set(uint64_t i, DU v);

    static Tensor &ten_op(int op, Tensor &A, DU v, Tensor &O);
    static Tensor &ten_op(int op, Tensor &A, Tensor &B, Tensor &O);
    static Tensor &mm(Tensor &A, Tensor &B, Tensor &O, bool inc = false, bool tA = false, bool tB = false);
    static Tensor &gemm(int variant, Tensor &A, Tensor &B, Tensor &O, DU alpha, DU beta);
    static Tensor &transpose(Tensor &A, Tensor &T);
    static Tensor &inverse(Tensor &A, Tensor &I);
    static Tensor &lu_inverse(Tensor &A, Tensor &I);
    static Tensor &plu(Tensor &A, Tensor &I, int *piv_dev);
    static Tensor &lu(Tensor &LU, bool get_u);
};

// ---------------------------------------------------------------- object store
class Store {
public:
    static Store &get();
    Tensor &tensor(uint64_t sz);
    Tensor &tensor(uint32_t h, uint32_t w);
    Tensor &tensor(uint32_t n, uint32_t h, uint32_t w, uint32_t c);
    Tensor &copy(Tensor &t);                          // deep copy (grad pointers dropped)
    Tensor &slice(Tensor &t, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1);
    Tensor &dim(Tensor &t);
    struct Model   &model(int *trace);
    struct Dataset &dataset(uint32_t batch);
    void free(Tensor &t);
    void drop(Obj &o);
    void mark_free(DU v);
    void sweep();
    Obj &du2obj(DU v);
    DU   obj2du(Obj &o);
    int  live() const { return nlive_; }
private:
    std::vector<Obj *> objs_;
    std::vector<int> free_ids_;
    std::vector<DU> marked_;
    int nlive_ = 0;
    int put(Obj *o);
    void release(Obj *o);
};

// ---------------------------------------------------------------- Dataset / loaders
struct Corpus {
    std::string name, f_data, f_label;
    int N = 0, H = 0, W = 0, C = 0, corpus_sz = 0;
    FILE *fd = nullptr, *fl = nullptr;
    bool init(int batch, bool trace = false);  // src/ld/mnist.cpp:21-62 (IDX header, big endian), cifar10.cpp:21-50
    bool cifar = false;
    bool names_shown_latch = false;            // a batch of this corpus has been fetched before (cifar10.cpp: `first` = its label block did not exist yet)
    int  n_batches() const { return N > 0 ? (corpus_sz + N - 1) / N : 0; }
    // double-buffered pinned staging: slot s = batch & 1 holds u8 pixels + u32 labels of one batch; a persistent
    // reader thread fills a slot (after the event of the launch that last read it) while the GPU works on earlier batches
    uint8_t  *pix[2] = {nullptr, nullptr};
    uint32_t *lab[2] = {nullptr, nullptr};
    t4k_event_t pin_done[2] = {nullptr, nullptr};   // recorded behind the staging launch that reads the slot (in-stream path)
    t4k_event_t pin_ev[2] = {nullptr, nullptr};     // the event that guards the slot now: pin_done[s], or the prefetch ring's `staged` event of that launch
    bool pin_wait[2] = {false, false};              // the event has been recorded since the slot was last filled
    int  slot_bid[2] = {-1, -1}, slot_n[2] = {0, 0};   // batch a slot holds (or is being filled with) and its sample count
    void *worker = nullptr;                    // persistent reader thread (dataset.cpp)
    int  read_into(int bid, int slot);         // blocking file read (runs on the worker thread, or inline on a cold start)
    void ensure_slots();
    void request(int bid);                     // read batch bid into slot bid & 1, asynchronously
    int  wait_batch(int bid);                  // samples of batch bid once its slot is filled (reads inline when nobody was asked to)
    void idle(); void settle();                               // wait for the reader and forget what the slots hold
};
struct Dataset : Tensor {
    uint64_t dataset_size = 0;
    int batch_id = 0, batch_sz = 0, done = 1;
    uint32_t *label = nullptr;                 // device: labels of the current batch (one of lbuf)
    DU mean = 0.0f, scale = 1.0f / 256.0f;     // src/mu/dataset.h:36-37
    Corpus *cp = nullptr;
    // prefetch ring (the reference's TODO, src/mu/dataset.cu:112): batch b lives in dbuf[b % RING]; while the model works on batch b the
    // staging launch of batch b + 1 runs on a side stream, so `fetch` / `next` only swap `data` / `label`
    static constexpr int RING = 8, MARK_EVERY = 4;         // RING >= MARK_EVERY + 2: see Dataset::fetch
    float    *dbuf[RING] = {};
    uint32_t *lbuf[RING] = {};
    int       dev_bid[RING], dev_n[RING] = {};
    t4k_event_t staged[RING] = {};             // side stream: batch is in its buffer (also what the reader waits for before refilling the pinned slot)
    t4k_event_t mark = nullptr;                // main stream: recorded every MARK_EVERY fetches, behind everything that read earlier batches
    int  mark_bid = -1;                        // the fetch (seq) it was recorded at (-1: none)
    int  seq = 0;                              // fetches since the ring was built: the ring slot of a fetch is seq % RING, ACROSS rewinds (batch 0 of the next epoch is staged at the last batch of this one)
    bool norm_dirty = false;                   // `normalize` since the last fetch: what the ring holds was scaled with the old constants
    uint64_t ring_numel = 0;
    Dataset() { for (int i = 0; i < RING; i++) dev_bid[i] = -1; }
    void set_norm(DU m, DU s) { mean = m; scale = 1.0f / s; norm_dirty = true; }
    int  fetch(const char *ds_name, bool rewind);
    static int *trace;                         // the VM's trace level (null: off): Dataset::fetch prints the reference's text at level >= 1
    void release_ring();
};

// ---------------------------------------------------------------- Model
struct Model : Obj {
    std::vector<Tensor *> layer;               // layer[i] = input tensor of op i; layer.back() = output
    bool train = true;
    bool err = false;
    int  epoch = 0, iter = 0, hit_ = 0;
    DU   max_norm = 0;
    int *trace = nullptr;
    Tensor *hot = nullptr, *loss_t = nullptr;
    unsigned char *hit_flags_ = nullptr, *hit_flags_dev_ = nullptr; int hit_flags_n_ = 0; bool hit_flags_pending_ = false, hit_flags_on_dev_ = false, hit_read_ = true;   // per-image hit flags the conv stack's head forward wrote (pinned host bytes when the loop reads `nn.hit` every batch, else device bytes): `nn.hit` adds them up
    int  *hit_dev = nullptr, *hit_pin = nullptr;          // scalar all-reduce scratch (HBM); hit counter (pinned host, written by k_hit)

    Tensor &at(int i) { return *layer[i < 0 ? (int)layer.size() + i : i]; }
    int  batch_size() { return layer.empty() ? 1 : (int)layer[0]->N(); }
    static void slab_exported() { use_opt_fold = false; }   // a zero-copy view of the gradient slab left the VM (ten4_grad_slab): its readers are invisible to the
                                               // library, so a conv stack's partial fold is never deferred to the optimizer again in this process
    void materialize_dx0();                    // produce the skipped dX of layer 0 now (Store::du2obj calls it through Tensor::stale_owner)
    void tick() { epoch++; iter = 0; }

    Model &add(int fn, uint32_t n = 0, DU bias = 0, uint16_t *opt = nullptr);
    Model &forward(Tensor &input);
    Model &broadcast(Tensor &tgt);
    Model &backprop();
    Model &backprop(Tensor &tgt);
    Tensor &onehot();
    Tensor &onehot(Tensor &t);
    Tensor &onehot(Dataset &d);
    int  hit(bool recalc = true);
    DU   dp_sum(DU v);                         // SUM over the data-parallel ranks (identity without a communicator)
    void onehot_hit(Dataset &d);               // forward(dataset): one-hot rows + hit count in one launch
    void hit_lazy();
    void traced_onehot_hit(Dataset &d);                           // forward(dataset): enqueue the count, defer the read-back
    bool hit_pending_ = false;
    DU   loss(Loss op);
    DU   loss(Loss op, Tensor &tgt);
    Model &sgd(DU lr, DU b = 0.9f);
    Model &adam(DU lr, DU b1 = 0.9f, DU b2 = 0.999f);
    Model &adamw(DU lr, DU wd = 0.001f, DU b1 = 0.9f, DU b2 = 0.999f);
    void  free_all();
    // ---- data-parallel hook: all dW|dB of the model live back-to-back in one slab (one all-reduce)
    Tensor *gslab = nullptr;
    static Model *current;                     // model that ran forward/backprop last (embedding API)
    // called from backprop right after the kernels that complete a layer's dW|dB have been enqueued: (layer, offset and
    // length of that layer's segment in the gradient slab).  Lets a data-parallel launcher start reducing the tail of
    // the slab (the big linear layers finish first) while the convolution layers are still back-propagating.
    static void (*grad_hook)(int layer, long off, long n, void *user);
    static void *grad_hook_user;
    static bool set_lazy_dx0(bool on) { const bool was = use_lazy_dx0; use_lazy_dx0 = on; return was; }   // ten4_set_lazy_dx0 (include/ten4.h)
    void  finalize();                          // build the gradient slab + side stream (first forward / backprop)
    void  invalidate();                        // drop captured graphs (layers added, shapes changed)
    static bool use_graphs, use_side;          // T4_GRAPH=0 / T4_SIDE=0 switch them off (debugging)
private:
    Tensor &T4(uint32_t n, uint32_t h, uint32_t w, uint32_t c);
    Tensor &VEC(uint64_t n);
    void RAND(Tensor &t, DU scale);
    const float *fstep(Tensor &in, Tensor &out, const float *x);
    const float *bstep(int i, Tensor &in, Tensor &out, const float *dy, bool last);
    void run_forward(Tensor &input);
    void run_backward(Tensor &tgt);
    Model &gradient(const char *nm, Optim op, DU lr, DU b1, DU b2, DU wd);
    // one-launch optimizer over a device parameter table
    std::vector<t4k_param_rec> tab_host_;      // host copy of tab_dev (t4k_opt_step matches a deferred fold's tensors against it)
    void *tab_dev = nullptr; int tab_n = 0; long tab_max = 0; int tab_chunks = 0; Optim tab_kind = OPTI_SGD;
    void build_table(Optim op);
    // ---- execution engine: the critical path (activations fwd, dX chain bwd) runs on the main stream;
    // copies the reference makes for bookkeeping (n0 = input, flatten, in = dX), dropout-mask generation
    // and all parameter gradients (dW, dB, dF) are forked to a side stream and joined at the end.  Each
    // word (forward / backprop / optimizer) is captured into a hipGraph on its second call with the
    // same operands and replayed afterwards.
    struct GraphSlot { t4k_graph_t g = nullptr; const void *key = nullptr; float p[4] = {0, 0, 0, 0}; int flags = -1, seen = 0; };
    GraphSlot g_fwd_, g_bwd_, g_opt_;
    t4k_stream_t side_ = nullptr;
    std::vector<t4k_event_t> ev_; size_t ev_i_ = 0;
    std::vector<Tensor *> gx_;                 // per-layer dX scratch of linear layers
    struct Run { int first = 0, count = 1; t4k_poolblock blk; };   // fused element-wise run starting at layer `first`
    std::vector<int> run_of_;                  // layer index -> index into runs_ or -1
    std::vector<Run> runs_;
    void plan_runs();
    static bool use_fusion;                    // T4_FUSE=0 keeps one launch per layer
    std::vector<char> stack_fresh_;            // per first-op index: the latest forward took the stack kernel (so what it saved for the backward is current)
    std::vector<int> stack_end_;              // last op of a conv stack -> its first op (run_backward), rebuilt after finalize
    double tl_ = 0;                                      // trace levels: clock of the previous layer's line
    bool stack_single_ = true;    // single-stage stacks too (round 3: slower, off; round 4, with the lazy first-layer dX and the fold inside the optimizer: the t4_40a net nn_c 0.0637 -> 0.0459 ms per step at N = 128)
    static bool use_head_bwd;                  // T4_HEAD_BWD=0: head backward and the linear layer in front of it as separate launches
    int also_ready_ = -1;                      // a second layer whose gradients the last bstep launch produced (run_backward reports it)
    static bool use_stack_head;                // T4_STACK_HEAD=0: conv stack and classifier head as separate launches
    static bool use_stack;                     // T4_STACK=0: no sample-resident conv stacks (csrc/conv_stack.hip)
    static bool use_lazy_dx0;                  // T4_LAZY_DX0=0: a conv stack's backward always computes the first layer's dX (default: on demand, materialize_dx0)
    bool dx0_stale_ = false;
    // ... of a first LINEAR layer (dX0 = dY W, backprop.cu:240): the backward runs dW | dB only and remembers where dY lives; the weight
    // tensor and the tensor holding dY carry the mark too (a word that could change either produces dX0 first), and an optimizer step in
    // between leaves the pre-update weights in w0_save_ (t4k_opt_snapshot: the copy rides in the update launch)
    bool dx0_lin_ = false, dx0_conv_ = false, w0_saved_ = false;   // dx0_lin_: a per-layer first layer (LINEAR, or CONV with dx0_conv_) rather than a conv stack
    const float *dx0_dy_ = nullptr;
    Tensor *dx0_dy_t_ = nullptr, *w0_save_ = nullptr;
    bool dp_in_opt_ = false;                   // this optimizer call sums the gradient slab over the ranks itself (one-shot peer exchange, t4k_opt_step_dp)
    void clear_dx0_marks();
    static bool use_opt_fold;                  // T4_OPT_FOLD=0: the conv stack's dF | dB partial fold as a launch of its own (default: inside the optimizer launch, t4k_opt_step)
    // sample-resident conv stack starting at layer i: [conv + run] x ns (stages filled for the C-ABI); ops = layers it covers
    int  stack_at(int i, t4k_conv_stage *st, int &ops);
    bool finalized_ = false, side_dirty_ = false, capturable_ = true;
    bool concurrent() const { return side_ != nullptr && !(trace && *trace); }
    t4k_stream_t fork();                       // side stream, ordered after everything issued on main so far
    void join();                               // main waits for the side stream
    void lazy_copy(const float *src, Tensor &dst);
    bool replay(GraphSlot &slot, const void *key, int flags, const float *p);
    void end_capture(GraphSlot &slot, bool capturing);
    bool capturing_ = false;
    // ---- data parallel (library-owned communicator): slab ranges that are complete are SUM all-reduced on a communication
    // stream while the earlier layers' backward still runs; `gradient` reduces the rest and joins before the update
    static int dp_overlap; static long dp_bucket;
    t4k_stream_t comm_s_ = nullptr; t4k_event_t dp_ev_[2] = { nullptr, nullptr };
    long dp_done_lo_ = -1, dp_pend_lo_ = -1;   // [dp_done_lo_, numel) submitted; [dp_pend_lo_, dp_done_lo_) complete, not yet submitted
    bool dp_busy_ = false, dp_mixed_ = false;
    void grads_ready(int i, Tensor &in);
    void dp_flush();
    void dp_begin_backward();
    void dp_finish();
    Tensor *prep_tgt_ = nullptr;               // `out -= target` pending: the last linear layer's backward launch performs it
    int  skip_cnt_ = 1;                        // ... how many ops in front it covered
    bool skip_next_ = false;                   // set by bstep when it also ran the backward of the op in front
};

// ---------------------------------------------------------------- printing
std::string fmt_scalar(DU v, int base);                 // src/io/aio.cpp:38-57
std::string fmt_objname(Obj &o, bool view);             // "T2[2,3]" etc, src/io/aio_tensor.cpp:16-58
std::string fmt_tensor(Tensor &t, int thres = 0);                      // src/io/aio_tensor.cpp:141-226
std::string fmt_model(Model &m);                        // src/io/aio_model.cpp:65-141
std::string fmt_dump(Tensor &t);                        // Tensor::_dump of a parameter tensor (optimizer trace, gradient.cu:70-74)
std::string fmt_show(Tensor &t, bool dump);              // Tensor::show src/mu/tensor.cu:587-684 (trace levels 1 / 2)
std::string fmt_parm(Tensor &in, Tensor &out);          // src/io/aio_model.cpp:103-141 (layer parameter text)
int model_save(Model &m, const char *fname);            // src/io/aio_model.cpp:16-35,143-181 (.t4 model file)
int model_load(Model &m, const char *fname);            // src/io/aio_model.cpp:38-60,206-238 (parameter section)

} // namespace t4
