// model.cpp - NN model: layer factory, forward / backprop orchestration, loss, optimizers.
// Restates src/nn/model.cpp:82-310, forward.cu:28-113, backprop.cu:39-140, gradient.cu:19-169,
// loss.cpp:16-136 on top of the t4k_* C-ABI.  Everything is launched asynchronously on one
// stream; the host only synchronises when a scalar (loss, hit) is read back.
#include <chrono>
#include "t4.h"
#include <algorithm>

namespace t4 {

#define NLOG(...) do { if (trace && *trace) hprintf(__VA_ARGS__); } while (0)
#define TSHOW(t, dump) hputs(fmt_show((t), (dump)))                     /* Tensor::show(dump) */
static double trace_ms() { static const auto t0 = std::chrono::steady_clock::now(); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }   // System::clock()

Tensor &Model::T4(uint32_t n, uint32_t h, uint32_t w, uint32_t c) { return Store::get().tensor(n, h, w, c); }
Tensor &Model::VEC(uint64_t n) { return Store::get().tensor(n); }
void Model::RAND(Tensor &t, DU scale) {                 // Model::RAND model.cpp:73-78: uniform [-scale, scale)
    chk(t4k_rand(t.data, (long)t.numel, T4K_UNIFORM, -0.5f, scale * 2.0f, stream()), "rand");
}

// ---------------------------------------------------------------- layer factory (Model::add)
Model &Model::add(int fn, uint32_t n, DU bias, uint16_t *opt) {
    Tensor &in = at(-1);
    if (in.grad_fn != T4K_L_NONE) return *this;
    NLOG("  Model::add %s n=%d bias=%g {\n", LAYER_NAME[fn], n, bias);
    for (int i = 0; i < 5; i++) in.grad[i] = in.mtum[i] = nullptr;
    switch (fn) {
    case T4K_L_CONV: case T4K_L_DCONV: {                // _iconv model.cpp:121-180
        const bool txn = fn == T4K_L_DCONV;
        const uint32_t N1 = in.N(), H1 = in.H(), W1 = in.W(), C1 = in.C(), C0 = n;
        const uint16_t K = opt[0], S = opt[1];
        const uint16_t P = (K > 1 && opt[2]) ? opt[2] : (K - 1) / 2;
        uint16_t H0, W0;
        if (txn) { const uint16_t P0 = (H1 + P * 2 - K) % S; H0 = (H1 - 1) * S - P * 2 + K + P0; W0 = (W1 - 1) * S - P * 2 + K + P0; }
        else     { H0 = (H1 - K + P * 2) / S + 1; W0 = (H1 - K + P * 2) / S + 1; }      // W0 from H1: reference quirk :137
        if ((!txn && K != 1 && K != 3 && K != 5) || (txn && K != 4)) {
            hprintf("nn#iconv %s f=[%d,%d]? 1x1, 3x3, 4x4, and 5x5 supported only.\n", LAYER_NAME[fn], K, K);
            return *this;
        }
        in.stride[0] = in.stride[1] = S; in.stride[2] = in.stride[3] = P; in.xparm = bias;
        Tensor *f = in.grad[0] = &T4(C1, K, K, C0);
        Tensor *b = in.grad[1] = &VEC(C0);
        in.grad[2] = &T4(C1, K, K, C0).zeros();
        in.grad[3] = &VEC(C0).zeros();
        in.grad[4] = &T4(N1, H1, W1, C1).zeros();
        RAND(*f, sqrtf(6.0f / (K * K * C1))); RAND(*b, bias);
        layer.push_back(&T4(N1, H0, W0, C0));
    } break;
    case T4K_L_LINEAR: {                                // _ilinear model.cpp:182-226
        const uint32_t N1 = in.N(); const uint64_t E1 = in.HWC(); const uint32_t E0 = n;
        Tensor *w = in.grad[0] = &T4(1, E0, (uint32_t)E1, 1);
        Tensor *b = in.grad[1] = &VEC(E0);
        in.grad[2] = &T4(1, E0, (uint32_t)E1, 1).zeros();
        in.grad[3] = &VEC(E0).zeros();
        if (in.W() != E1) hprintf("    WARN linear: treats in[%d,%d,%d,%d] as [%d,1,%ld,1]\n", N1, in.H(), in.W(), in.C(), N1, (long)E1);
        in.xparm = bias;
        RAND(*w, sqrtf(1.0f / (E0 + E1))); RAND(*b, bias);
        layer.push_back(&T4(N1, 1, E0, 1));
    } break;
    case T4K_L_FLATTEN: layer.push_back(&T4(in.N(), 1, (uint32_t)in.HWC(), 1)); break;
    case T4K_L_RELU: case T4K_L_TANH: case T4K_L_SIGMOID: case T4K_L_SELU: case T4K_L_LEAKYRL: case T4K_L_ELU: case T4K_L_DROPOUT:
        in.grad[4] = &T4(in.N(), in.H(), in.W(), in.C()); in.xparm = bias;
        layer.push_back(&T4(in.N(), in.H(), in.W(), in.C())); break;
    case T4K_L_SOFTMAX: case T4K_L_LOGSMAX:
        in.grad[4] = &T4(1, in.H(), in.W(), in.C());
        layer.push_back(&T4(in.N(), in.H(), in.W(), in.C())); break;
    case T4K_L_AVGPOOL: case T4K_L_MAXPOOL: case T4K_L_MINPOOL: {    // _ipool model.cpp:260-274 (ceil dims)
        const uint16_t k = (uint16_t)n;
        if (k != 2 && k != 3) { hprintf("nn#ipool k=%dx%d? 2x2 and 3x3 supported only\n", k, k); return *this; }
        in.stride[0] = k; in.stride[1] = 1; in.stride[2] = 1; in.stride[3] = 0;
        layer.push_back(&T4(in.N(), (in.H() + k - 1) / k, (in.W() + k - 1) / k, in.C()));
    } break;
    case T4K_L_BATCHNM: {                               // _ibatchnorm model.cpp:276-292 (dW/dB zeroed here)
        const int C = in.C();
        in.grad[0] = &VEC(C).map(T4K_FILL, 1.0f); in.grad[2] = &VEC(C).zeros();
        in.grad[1] = &VEC(C).zeros();             in.grad[3] = &VEC(C).zeros();
        in.grad[4] = &T4(in.N(), in.H(), in.W(), in.C());
        in.mtum[4] = &VEC(C * 3).zeros();
        in.xparm = bias;
        layer.push_back(&T4(in.N(), in.H(), in.W(), in.C()));
    } break;
    case T4K_L_USAMPLE: {                               // _iup model.cpp:294-310
        const uint16_t k = (uint16_t)n;
        if (k != 2 && k != 3) { hprintf("nn#iup k=%dx%d? only 2x2 and 3x3 supported\n", k, k); return *this; }
        in.iparm = (int)bias;
        in.stride[0] = k; in.stride[1] = 1; in.stride[2] = 1; in.stride[3] = 1;
        layer.push_back(&T4(in.N(), in.H() * k, in.W() * k, in.C()));
    } break;
    default: hprintf("Model#add layer %d not supported\n", fn); return *this;
    }
    in.grad_fn = fn;
    invalidate();
    Tensor &out = at(-1);
    NLOG("  } Model::add[%ld] %s => out[%d,%d,%d,%d]\n", (long)layer.size(), LAYER_NAME[fn], out.N(), out.H(), out.W(), out.C());
    return *this;
}

// ---------------------------------------------------------------- execution engine
// Both are OFF by default: measured on MI355X (ROCm 7.2, LeNet step, 40 launches) the single in-order stream is the
// fastest schedule - 0.26 ms/step vs 0.28 (hipGraph replay) vs 0.31 (forked side stream; every cross-queue event edge
// costs more than the ~4.5 us in-order dispatch it hides).  What pays is fewer launches (fused kernels below).
Model *Model::current = nullptr;
void (*Model::grad_hook)(int, long, long, void *) = nullptr;
void *Model::grad_hook_user = nullptr;
bool Model::use_fusion = env_flag("T4_FUSE", true);
bool Model::use_stack  = env_flag("T4_STACK", true);
bool Model::use_head_bwd = env_flag("T4_HEAD_BWD", true);    // classifier-head backward and the linear layer in front of it in ONE launch (round 6: k_head_bwd_l32, 9.7 us; T4_HEAD_BWD=0: head on column stripes + dual GEMM, 6.9 + 5.6 us)
bool Model::use_stack_head = env_flag("T4_STACK_HEAD", true);   // T4_STACK_HEAD=0: the classifier head behind a conv stack keeps its own launches
bool Model::use_lazy_dx0 = env_flag("T4_LAZY_DX0", true);
bool Model::use_opt_fold = env_flag("T4_OPT_FOLD", true);   // T4_OPT_FOLD=0: the conv stack's partial fold keeps its own launch behind the backward
bool Model::use_graphs = env_flag("T4_GRAPH", false);
bool Model::use_side   = getenv("T4_SIDE")  ? atoi(getenv("T4_SIDE"))  != 0 : false;

// Lazy dX of the first layer.  `in = dx` (backprop.cu:185) leaves the gradient w.r.t. the input batch in layer 0 and in the first conv
// layer's scratch tensor; a training loop never reads either.  The conv-stack backward skips it (t4k_conv_stack_bwd, train | 8) and the
// two tensors carry a mark; the first word that resolves one of them (Store::du2obj: `0 n@`, `0 nn.ex`, ten4_fetch, a chained backprop ...)
// has it produced from the dO and the filter copy that backward left.  The next forward overwrites layer 0 and ends the offer (after it
// `0 nn.ex` would show an EARLIER backward's dX - the one observable difference; T4_LAZY_DX0=0 restores the eager store).
// A first LINEAR layer (an MLP: the GAN discriminator's 784 -> 512 layer is a third of its backward launch) takes the same offer: the
// backward computes dW | dB only; dX0 = dY W is produced on demand from the dY the backward left and the weights OF THAT BACKWARD - the
// weight tensor and the tensor holding dY carry the mark as well, and an optimizer step in between snapshots the old weights inside its
// own launch (t4k_opt_snapshot).
void Model::clear_dx0_marks() {
    if (!dx0_stale_) return;
    dx0_stale_ = false;
    if (!layer.empty()) { at(0).stale_owner = nullptr; if (at(0).grad[4]) at(0).grad[4]->stale_owner = nullptr; }
    if (dx0_lin_) {
        if (!layer.empty() && at(0).grad[0]) at(0).grad[0]->stale_owner = nullptr;
        if (dx0_dy_t_) dx0_dy_t_->stale_owner = nullptr;
        dx0_lin_ = false; dx0_conv_ = false; w0_saved_ = false; dx0_dy_ = nullptr; dx0_dy_t_ = nullptr;
    }
}
void Model::materialize_dx0() {
    if (!dx0_stale_) return;
    if (dx0_lin_) {
        Tensor &in = at(0), &o = at(1);
        const float *w = w0_saved_ ? w0_save_->data : in.grad[0]->data, *dy = dx0_dy_;
        const bool conv = dx0_conv_;
        clear_dx0_marks();
        if (conv) {                                     // dX into the scratch tensor and over x, as the eager launch leaves them (backprop.cu:185)
            chk(t4k_conv2d_bwd2(in.data, dy, in.grad[4]->data, in.data, w, nullptr, nullptr, in.N(), in.H(), in.W(), in.C(), o.H(), o.W(), o.C(),
                                in.grad[0]->H(), in.stride[0], in.stride[2], 0, stream()), "nn#bconv dX0");
            return;
        }
        chk(t4k_linear_bwd(in.data, w, dy, in.data, nullptr, nullptr, in.N(), (int)o.HWC(), (int)in.HWC(), 0, stream()), "nn#blinear dX0");
        return;
    }
    clear_dx0_marks();
    t4k_conv_stage stg[3]; int ops = 0;
    if (stack_at(0, stg, ops) > 0) chk(t4k_conv_stack_dx0(stg, at(0).N(), stream()), "nn#bstack dX0");
}
void Model::invalidate() {
    clear_dx0_marks();
    for (GraphSlot *g : { &g_fwd_, &g_bwd_, &g_opt_ }) { if (g->g) t4k_graph_destroy(g->g); *g = GraphSlot(); }
    finalized_ = false;
}
void Model::finalize() {                                // gradient slab: SURVEY 8(e), one contiguous all-reduce buffer
    if (finalized_) return;
    finalized_ = true;
    auto pad = [](uint64_t n) { return (n + 63) & ~(uint64_t)63; };          // keep every tensor 256 B aligned
    uint64_t total = 0;
    capturable_ = true;
    for (Tensor *t : gx_) if (t) Store::get().free(*t);
    gx_.assign(layer.size(), nullptr);
    for (int i = 0; i + 1 < (int)layer.size(); i++) {
        Tensor &in = at(i);
        for (int k = 2; k < 4; k++) if (in.grad[k]) total += pad(in.grad[k]->numel);
        if (use_side && in.grad_fn == T4K_L_LINEAR && i + 2 < (int)layer.size()) gx_[i] = &T4(in.N(), in.H(), in.W(), in.C());
    }
    if (total) {
        Tensor *slab = &VEC(total).zeros();
        uint64_t off = 0;
        for (int i = 0; i + 1 < (int)layer.size(); i++) {
            Tensor &in = at(i);
            for (int k = 2; k < 4; k++) {
                Tensor *g = in.grad[k]; if (!g) continue;
                chk(t4k_copy(g->data, slab->data + off, (long)g->numel, stream()), "slab");
                if (g->owns && g->data) { t4k_sync(stream()); Arena::get().free(g->data); }
                g->data = slab->data + off; g->owns = false;
                off += pad(g->numel);
            }
        }
        if (gslab) Store::get().free(*gslab);
        gslab = slab;
        if (dp_busy_) { t4k_sync(comm_s_); dp_busy_ = false; }
        dp_done_lo_ = dp_pend_lo_ = -1; dp_mixed_ = false;     // offsets of the old slab are meaningless now
    }
    if (tab_dev) { t4k_free(tab_dev); tab_dev = nullptr; }                  // parameter table holds the old pointers
    if (use_side && !side_) chk(t4k_stream_create(&side_), "side stream");
    plan_runs();
    stack_end_.clear();
    t4k_sync(stream());
}
// Fused element-wise runs: [dropout|activation] [pool] [activation] [flatten] -> one launch each way (csrc/fused.hip).
// Only mask-multiply activations qualify (sigmoid is pass-through in the reference's backprop, backprop.cu:129-131).
static bool is_eltwise(int f) { return f == T4K_L_RELU || f == T4K_L_TANH || f == T4K_L_SIGMOID || f == T4K_L_SELU || f == T4K_L_LEAKYRL || f == T4K_L_ELU || f == T4K_L_DROPOUT; }
void Model::plan_runs() {
    const int L = (int)layer.size() - 1;                // number of ops
    run_of_.assign(layer.size(), -1); runs_.clear();
    if (!use_fusion) return;
    auto is_mact = [](int f) { return f == T4K_L_RELU || f == T4K_L_TANH || f == T4K_L_SELU || f == T4K_L_LEAKYRL || f == T4K_L_ELU; };
    auto is_pool = [](int f) { return f == T4K_L_AVGPOOL || f == T4K_L_MAXPOOL || f == T4K_L_MINPOOL; };
    for (int i = 0; i < L; ) {
        int j = i, pre = -1, pool = -1, post = -1, flat = -1;
        if (j < L && (is_mact(at(j).grad_fn) || at(j).grad_fn == T4K_L_DROPOUT)) pre = j++;
        if (j < L && is_pool(at(j).grad_fn) && (at(j).H() % at(j).stride[0] == 0) && (at(j).W() % at(j).stride[0] == 0)) pool = j++;
        if (j < L && (pool >= 0 || pre >= 0) &&
            (is_mact(at(j).grad_fn) || (at(j).grad_fn == T4K_L_DROPOUT && !(pre >= 0 && at(pre).grad_fn == T4K_L_DROPOUT)))) post = j++;   // activation or dropout behind
                                                                // (`leakyrelu dropout` of the GAN nets, `maxpool dropout` of the CIFAR nets): one dropout per run
        if (j < L && at(j).grad_fn == T4K_L_FLATTEN && j > i) flat = j++;
        if (j - i < 2 && !(j - i == 1 && pre >= 0 && at(pre).grad_fn == T4K_L_DROPOUT)) { i++; continue; }   // a lone dropout still fuses its mask draw
        Run r; r.first = i; r.count = j - i;
        t4k_poolblock &b = r.blk; memset(&b, 0, sizeof(b)); b.KS = 1;
        if (pre >= 0)  { b.pre_layer = at(pre).grad_fn; b.pre_alpha = at(pre).xparm; b.pre_mask = at(pre).grad[4]->data; b.pre_out = at(pre + 1).data; }
        if (pool >= 0) { b.pool_layer = at(pool).grad_fn; b.KS = at(pool).stride[0]; b.pool_out = at(pool + 1).data; }
        if (post >= 0) { b.post_layer = at(post).grad_fn; b.post_alpha = at(post).xparm; b.post_mask = at(post).grad[4]->data; b.post_out = at(post + 1).data; }
        if (flat >= 0) b.copy_out = at(flat + 1).data;
        run_of_[i] = (int)runs_.size(); runs_.push_back(r);
        i = j;
    }
}
// A run of [conv KxK stride 1 "same" + element-wise run] blocks that one workgroup per image can carry through LDS (conv_stack.hip).
// Returns the number of stages (0: layer i does not start such a stack), fills the C-ABI stage records for both directions.
int Model::stack_at(int i, t4k_conv_stage *st, int &ops) {
    const int L = (int)layer.size() - 1;
    int ns = 0, j = i, cnt[3] = {0, 0, 0};
    while (ns < 3 && j < L && at(j).grad_fn == T4K_L_CONV) {
        Tensor &in = at(j), &out = at(j + 1);
        const int K = in.grad[0]->H();
        if (in.stride[0] != 1 || in.stride[2] != K / 2 || !(K & 1) || out.H() != in.H() || out.W() != in.W()) break;
        t4k_conv_stage &t = st[ns]; memset(&t, 0, sizeof(t));
        t.F = in.grad[0]->data; t.B = in.grad[1]->data; t.O = out.data;
        t.DF = in.grad[2] ? in.grad[2]->data : nullptr; t.DB = in.grad[3] ? in.grad[3]->data : nullptr;
        t.X = in.data; t.DXS = in.grad[4] ? in.grad[4]->data : nullptr;
        t.H = in.H(); t.W = in.W(); t.C1 = in.C(); t.C0 = out.C(); t.K = K;
        t.run.KS = 1;
        cnt[ns] = 1;
        if (j + 1 < L && run_of_[j + 1] >= 0) { const Run &r = runs_[run_of_[j + 1]]; t.run = r.blk; cnt[ns] += r.count; }
        j += cnt[ns]; ns++;
        if (t.run.copy_out) break;                          // a flatten closes the stack
    }
    while (ns > 0 && !t4k_conv_stack_ok(st, ns, at(i).N())) ns--;     // longest prefix the kernel can hold
    ops = 0;
    for (int s = 0; s < ns; s++) ops += cnt[s];
    return ns;
}
t4k_stream_t Model::fork() {
    if (!concurrent()) return stream();
    if (ev_.size() < 16) { ev_.resize(16, nullptr); for (auto &e : ev_) t4k_event_create(&e); }
    t4k_event_t e = ev_[ev_i_++ % ev_.size()];
    t4k_event_record(e, stream()); t4k_stream_wait_event(side_, e);
    side_dirty_ = true;
    return side_;
}
void Model::join() {
    if (!side_dirty_ || !side_) { side_dirty_ = false; return; }
    t4k_event_t e = ev_[ev_i_++ % ev_.size()];
    t4k_event_record(e, side_); t4k_stream_wait_event(stream(), e);
    side_dirty_ = false;
}
void Model::lazy_copy(const float *src, Tensor &dst) {   // bookkeeping copy, off the critical path
    if (src == dst.data) return;
    chk(t4k_copy(src, dst.data, (long)dst.numel, fork()), "copy");
}
bool Model::replay(GraphSlot &slot, const void *key, int flags, const float *p) {
    capturing_ = false;
    if (!use_graphs || !capturable_ || (trace && *trace) || t4k_comm_world() > 1 || t4k_rand_shard_world() > 1) return false;   // data parallel (a communicator, or a shard set for another transport): mask draws are keyed on the host per launch, a replay would repeat them
    const bool same = slot.key == key && slot.flags == flags && (!p || memcmp(slot.p, p, sizeof(slot.p)) == 0);
    if (same && slot.g) { chk(t4k_graph_launch(slot.g, stream()), "graph launch"); return true; }
    if (!same) {                                        // new operands: run eagerly once, capture next time
        if (slot.g) { t4k_graph_destroy(slot.g); slot.g = nullptr; }
        slot.key = key; slot.flags = flags; slot.seen = 0;
        if (p) memcpy(slot.p, p, sizeof(slot.p));
    }
    if (slot.seen++ >= 1) capturing_ = t4k_graph_begin(stream()) == T4K_OK;
    return false;
}
void Model::end_capture(GraphSlot &slot, bool capturing) {
    if (!capturing) return;
    capturing_ = false;
    if (chk(t4k_graph_end(stream(), &slot.g), "graph capture") != T4K_OK) { slot.g = nullptr; use_graphs = false; return; }
    chk(t4k_graph_launch(slot.g, stream()), "graph launch");
}

// ---------------------------------------------------------------- forward
Model &Model::forward(Tensor &input) {
    Tensor &n0 = at(0);
    if (trace && *trace) TSHOW(input, true);             // preview of the input (forward.cu:31)
    if (input.numel != n0.numel) {
        hprintf("nn#forward dataset wrong shape[%d,%d,%d,%d] != model input[%d,%d,%d,%d]\n",
               input.N(), input.H(), input.W(), input.C(), n0.N(), n0.H(), n0.W(), n0.C());
        return *this;
    }
    finalize(); current = this;
    hit_flags_pending_ = false;
    NLOG("\nModel::forward starts trace=%d {", *trace);
    const double tf0 = trace_ms();
    if (!replay(g_fwd_, input.data, (int)train, nullptr)) {
        const bool cap = capturing_;
        run_forward(input);
        end_capture(g_fwd_, cap);
    }
    if (input.type == T_DATASET && trace && *trace) traced_onehot_hit((Dataset &)input);
    else
    if (input.type == T_DATASET && !hit_flags_pending_) onehot_hit((Dataset &)input);   // labels -> one-hot rows and the hit count, one launch (or none: they rode in the conv stack's head forward)
    NLOG("\n} Model::forward %5.2f ms\n", trace_ms() - tf0);
    return *this;
}
// data parallel: batch-norm statistics span all ranks during TRAINING passes only (every rank runs those in lock step); an
// evaluation pass (`0 trainable`, possibly on one rank only) uses the rank's own statistics and issues no collective
static void dp_bn_mode(bool train) {
    static const bool want = env_flag("T4_DP_SYNC_BN", true);
    if (t4k_comm_world() > 0) t4k_comm_sync_batchnorm(train && want);
}
void Model::run_forward(Tensor &input) {
    hit_flags_pending_ = false;
    clear_dx0_marks();                                   // this pass overwrites layer 0: a skipped dX of the previous backward is gone for good
    dp_bn_mode(train);
    const int L = (int)layer.size();
    Tensor &n0 = at(0);
    const bool fused = use_fusion && !(trace && *trace) && !concurrent();
    // layer 0 holds a COPY of the batch (forward.cu:39); a first conv layer reads the batch itself and writes the copy from its own launch
    const bool copy_in_conv = fused && L > 1 && n0.grad_fn == T4K_L_CONV && input.data != n0.data;
    // what stands behind linear layer i: 0 nothing fusable, 1 classifier head, 2 lone activation / dropout, 3 element-wise run without pool / flatten
    auto lin_kind = [&](int i) -> int {
        if (!(i + 2 < L) || at(i).grad_fn != T4K_L_LINEAR) return 0;
        const int rn = run_of_[i + 1];
        if (rn >= 0 && runs_[rn].count >= 2) return (!runs_[rn].blk.pool_layer && !runs_[rn].blk.copy_out) ? 3 : 0;
        if (!is_eltwise(at(i + 1).grad_fn)) return 0;
        return (at(i + 2).grad_fn == T4K_L_LINEAR && i + 4 < L && at(i + 3).grad_fn == T4K_L_SOFTMAX) ? 1 : 2;
    };
    const bool copy_in_lin = fused && input.data != n0.data && lin_kind(0) >= 2;   // a first linear layer's fold launch carries the layer-0 copy
    if (!copy_in_conv && !copy_in_lin) lazy_copy(input.data, n0);
    bool masks = false;
    if (concurrent())                                   // side stream: draw every dropout mask up front, in layer order
        for (int i = 0; i + 1 < L; i++)
            if (at(i).grad_fn == T4K_L_DROPOUT) {
                Tensor &m = *at(i).grad[4];
                chk(t4k_dropout_mask(m.data, (long)m.numel, fork()), "rand"); masks = true;
            }
    tl_ = trace_ms();
    stack_fresh_.assign(layer.size(), 0);                // which conv stacks this forward ran through t4k_conv_stack_fwd (their saved state is current)
    const float *x = input.data;
    for (int i = 0; i + 1 < L; i++) {
        Tensor &in = at(i), &out = at(i + 1);
        if (trace && *trace) {                          // forward.cu:44-58: time since the previous layer's line, the layer, its input's sum per sample and channel
            const double tt = trace_ms();
            hprintf("\n%6.2f:%3d> %s [%2d,%2d,%2d,%2d] \xCE\xA3/n=%6.2f p=%6.3f => out[%2d,%2d,%2d,%2d]", tt - tl_, i, LAYER_NAME[in.grad_fn],
                   in.N(), in.H(), in.W(), in.C(), in.sum() / in.N() / in.C(), in.xparm, out.N(), out.H(), out.W(), out.C());
            tl_ = tt;
        }
        if (masks && in.grad_fn == T4K_L_DROPOUT) { join(); masks = false; }
        if (fused && run_of_[i] >= 0) {                 // one launch for the whole element-wise run
            const Run &r = runs_[run_of_[i]];
            Tensor &lastt = at(i + r.count), &pin = (r.blk.pool_layer ? at(i + (r.blk.pre_layer ? 1 : 0)) : in);
            chk(t4k_poolblock_fwd(x, &r.blk, in.N(), pin.H(), pin.W(), r.blk.pool_layer ? at(i + (r.blk.pre_layer ? 2 : 1)).H() : pin.H(),
                                  r.blk.pool_layer ? at(i + (r.blk.pre_layer ? 2 : 1)).W() : pin.W(), pin.C(), stream()), "nn#frun");
            x = lastt.data; i += r.count - 1;
            continue;
        }
        if (fused && in.grad_fn == T4K_L_LINEAR && lin_kind(i) == 3) {   // linear + an element-wise run of two (`leakyrelu dropout`): everything rides in the GEMM's fold launch
            const Run &r = runs_[run_of_[i + 1]];
            chk(t4k_linear_block_fwd(x, (i == 0 && copy_in_lin) ? n0.data : nullptr, in.grad[0]->data, in.grad[1]->data, out.data, &r.blk,
                                     out.N(), (int)out.HWC(), (int)in.HWC(), stream()), "nn#flinear+run");
            x = at(i + 1 + r.count).data; i += r.count;
            continue;
        }
        if (fused && in.grad_fn == T4K_L_LINEAR && lin_kind(i) >= 1) {   // linear + lone activation / dropout: the activation rides in the GEMM's fold launch
            Tensor &act = at(i + 2);
            if (lin_kind(i) == 1) {
                // classifier head: [linear + activation] + [linear + softmax] - the second launch folds the first GEMM's split-K slabs
                Tensor &y2 = at(i + 3), &prob = at(i + 4);
                chk(t4k_mlp_head_fwd(x, in.grad[0]->data, in.grad[1]->data, out.data, out.grad_fn, out.xparm, out.grad[4]->data, act.data,
                                     act.grad[0]->data, act.grad[1]->data, y2.data, prob.data, out.N(), (int)out.HWC(), (int)in.HWC(), (int)y2.HWC(), stream()),
                    "nn#fhead");
                x = prob.data; i += 3;
                continue;
            }
            if (i == 0 && copy_in_lin) {                // the copy of the batch into layer 0 goes with the same launch
                t4k_poolblock b1; memset(&b1, 0, sizeof(b1)); b1.KS = 1;
                b1.pre_layer = out.grad_fn; b1.pre_alpha = out.xparm; b1.pre_mask = out.grad[4]->data; b1.pre_out = act.data;
                chk(t4k_linear_block_fwd(x, n0.data, in.grad[0]->data, in.grad[1]->data, out.data, &b1, out.N(), (int)out.HWC(), (int)in.HWC(), stream()), "nn#flinear+act");
            } else
                chk(t4k_linear_act_fwd(x, in.grad[0]->data, in.grad[1]->data, out.data, out.grad_fn, out.xparm, out.grad[4]->data, act.data,
                                       out.N(), (int)out.HWC(), (int)in.HWC(), stream()), "nn#flinear+act");
            x = act.data; i++;
            continue;
        }
        if (fused && in.grad_fn == T4K_L_LINEAR && i + 2 < L && out.grad_fn == T4K_L_SOFTMAX) {     // classifier head: linear + softmax in one launch
            Tensor &prob = at(i + 2);
            chk(t4k_linear_softmax_fwd(x, in.grad[0]->data, in.grad[1]->data, out.data, prob.data, out.N(), (int)out.HWC(), (int)in.HWC(), stream()), "nn#flinear+softmax");
            x = prob.data; i++;
            continue;
        }
        if (fused && use_stack && in.grad_fn == T4K_L_CONV) {   // [conv + run] x n, one workgroup per image, activations in LDS: ONE launch
            t4k_conv_stage stg[3]; int ops = 0;
            const int ns = stack_at(i, stg, ops);
            if (ns >= 2 || (ns == 1 && stack_single_)) {
                const int j = i + ops;                  // the layer behind the stack: a classifier head [linear + activation] + [linear + softmax]?
                if (use_stack_head && j + 1 < L && lin_kind(j) == 1) {   // then the whole forward pass is ONE launch
                    Tensor &l1 = at(j), &y1 = at(j + 1), &act = at(j + 2), &y2 = at(j + 3), &prob = at(j + 4);
                    t4k_stack_head hd; memset(&hd, 0, sizeof(hd));
                    hd.W1 = l1.grad[0]->data; hd.B1 = l1.grad[1]->data; hd.Y1 = y1.data;
                    hd.mid_layer = y1.grad_fn; hd.mid_alpha = y1.xparm; hd.mid_mask = y1.grad[4]->data; hd.mid_out = act.data;
                    hd.W2 = act.grad[0]->data; hd.B2 = act.grad[1]->data; hd.Y2 = y2.data; hd.P = prob.data;
                    hd.E1 = (int)l1.HWC(); hd.E0a = (int)y1.HWC(); hd.E0b = (int)y2.HWC();
                    // a dataset batch: the one-hot rows and the hit flags of Model::onehot(Dataset&) + hit (forward.cu:57-60) ride in the same launch
                    bool rider = false;
                    if (input.type == T_DATASET && j + 4 == L - 1 && !capturing_ && !use_graphs) {
                        Dataset &ds = (Dataset &)input;
                        const uint32_t E = (uint32_t)prob.HWC();
                        if (ds.label && ds.batch_sz >= 1 && (uint32_t)ds.batch_sz <= prob.N()) {
                            if (!hot) hot = &T4(prob.N(), 1, E, 1);
                            if ((uint32_t)ds.batch_sz < prob.N()) hot->zeros();     // short last batch: the rows past it stay zero and count nothing
                            if (!hit_flags_ || hit_flags_n_ < (int)prob.N()) {
                                if (hit_flags_) { t4k_sync(stream()); t4k_host_free(hit_flags_); t4k_free(hit_flags_dev_); }
                                void *pp; chk(t4k_host_alloc(&pp, prob.N()), "nn#hit flags"); hit_flags_ = (unsigned char *)pp; hit_flags_n_ = (int)prob.N();
                                chk(t4k_malloc(&pp, prob.N()), "nn#hit flags"); hit_flags_dev_ = (unsigned char *)pp;
                            }
                            // where the flags go: a loop that reads `nn.hit` every batch (the reference's demos, t4_30e.4th:68-76) gets them in pinned host memory -
                            // its device sync makes them readable, no copy; a loop that does not keeps them on the device, where the store does not hold the
                            // kernel's end behind a PCIe round trip, and the rare `nn.hit` copies them out
                            hit_flags_on_dev_ = !hit_read_;
                            hit_read_ = false;
                            hd.label = ds.label; hd.hot = hot->data; hd.hit_flag = hit_flags_on_dev_ ? hit_flags_dev_ : hit_flags_; hd.n_label = ds.batch_sz; rider = true;
                        }
                    }
                    if (t4k_conv_stack_head_ok(stg, ns, in.N(), &hd)) {
                        chk(t4k_conv_stack_head_fwd(x, (i == 0 && copy_in_conv) ? n0.data : nullptr, stg, ns, in.N(), &hd, stream()), "nn#fstack+head");
                        if (rider) { hit_flags_pending_ = true; hit_pending_ = false; }
                        stack_fresh_[i] = 1;
                        x = prob.data; i = j + 3;
                        continue;
                    }
                }
                chk(t4k_conv_stack_fwd(x, (i == 0 && copy_in_conv) ? n0.data : nullptr, stg, ns, in.N(), stream()), "nn#fstack");
                stack_fresh_[i] = 1;
                x = at(i + ops).data; i += ops - 1;
                continue;
            }
        }
        if (fused && in.grad_fn == T4K_L_CONV && i + 2 < L && run_of_[i + 1] >= 0 && runs_[run_of_[i + 1]].blk.pool_layer &&
            runs_[run_of_[i + 1]].blk.KS == 2) {        // conv + the element-wise run behind it: the run rides in the conv epilogue
            const Run &r = runs_[run_of_[i + 1]];
            chk(t4k_conv2d_block_fwd(x, (i == 0 && copy_in_conv) ? n0.data : nullptr, out.data, in.grad[0]->data, in.grad[1]->data, &r.blk,
                                     out.N(), in.H(), in.W(), in.C(), out.H(), out.W(), out.C(), in.grad[0]->H(), in.stride[0], in.stride[2], stream()), "nn#fconv+run");
            x = at(i + 1 + r.count).data; i += r.count;
            continue;
        }
        if (fused && in.grad_fn == T4K_L_CONV && i + 3 < L && out.grad_fn == T4K_L_BATCHNM && run_of_[i + 2] >= 0) {   // conv + batch-norm + the element-wise run behind them: the conv
            const int i2 = i + 2;                          // output is read once (statistics from the conv's epilogue, batch-norm apply inside the run's launch)
            const Run &r = runs_[run_of_[i2]];
            Tensor &o2 = at(i2), &pin = (r.blk.pool_layer ? at(i2 + (r.blk.pre_layer ? 1 : 0)) : o2);
            const int Hq = r.blk.pool_layer ? at(i2 + (r.blk.pre_layer ? 2 : 1)).H() : pin.H(), Wq = r.blk.pool_layer ? at(i2 + (r.blk.pre_layer ? 2 : 1)).W() : pin.W();
            chk(t4k_conv2d_bn_block_fwd(x, (i == 0 && copy_in_conv) ? n0.data : nullptr, out.data, in.grad[0]->data, in.grad[1]->data, out.N(), in.H(), in.W(), in.C(),
                                        out.H(), out.W(), out.C(), in.grad[0]->H(), in.stride[0], in.stride[2],
                                        o2.data, out.grad[4]->data, out.grad[0]->data, out.grad[1]->data, out.mtum[4]->data, &r.blk, Hq, Wq, stream()), "nn#fconv+batchnorm+run");
            x = at(i2 + r.count).data; i = i2 + r.count - 1;
            continue;
        }
        if (fused && in.grad_fn == T4K_L_CONV && i + 2 < L && out.grad_fn == T4K_L_BATCHNM) {   // conv + batch-norm: the statistics ride in the conv's epilogue where its kernel carries them
            Tensor &o2 = at(i + 2);
            chk(t4k_conv2d_bn_fwd(x, (i == 0 && copy_in_conv) ? n0.data : nullptr, out.data, in.grad[0]->data, in.grad[1]->data, out.N(), in.H(), in.W(), in.C(),
                                  out.H(), out.W(), out.C(), in.grad[0]->H(), in.stride[0], in.stride[2],
                                  o2.data, out.grad[4]->data, out.grad[0]->data, out.grad[1]->data, out.mtum[4]->data, stream()), "nn#fconv+batchnorm");
            x = o2.data; i += 1;
            continue;
        }
        if (i == 0 && copy_in_conv) {
            chk(t4k_conv2d_fwd2(x, n0.data, out.data, in.grad[0]->data, in.grad[1]->data, out.N(), in.H(), in.W(), in.C(),
                                out.H(), out.W(), out.C(), in.grad[0]->H(), in.stride[0], in.stride[2], stream()), "nn#fconv");
            x = out.data;
            continue;
        }
        x = fstep(in, out, x);
        if (trace && *trace && out.has_nan()) {
            hprintf("nn#forward Nan in %s\n", LAYER_NAME[in.grad_fn]);
            hprintf("in=");  TSHOW(in, true);
            hprintf("out="); TSHOW(out, true);
            err = true; break;
        }
        if (trace && *trace > 1) TSHOW(out, true);      // `2 trace`: every layer's output (forward.cu:68)
    }
    join();
}
// one layer forward: reads the activations at `x` (normally in.data), writes out.data, returns where the
// next layer finds its input (_fstep forward.cu:82-113)
const float *Model::fstep(Tensor &in, Tensor &out, const float *x) {
    const int fn = in.grad_fn;
    t4k_stream_t s = stream();
    switch (fn) {
    case T4K_L_CONV:
        chk(t4k_conv2d_fwd(x, out.data, in.grad[0]->data, in.grad[1]->data, out.N(), in.H(), in.W(), in.C(),
                           out.H(), out.W(), out.C(), in.grad[0]->H(), in.stride[0], in.stride[2], s), "nn#fconv"); break;
    case T4K_L_DCONV:                                   // transposed convolution: the scatter form of the conv backward (forward.cu:110, see csrc/dconv.hip)
        chk(t4k_dconv2d_fwd(x, out.data, in.grad[0]->data, in.grad[1]->data, out.N(), in.H(), in.W(), in.C(),
                            out.H(), out.W(), out.C(), in.grad[0]->H(), in.stride[0], in.stride[2], s), "nn#fdconv"); break;
    case T4K_L_LINEAR:
        chk(t4k_linear_fwd(x, in.grad[0]->data, in.grad[1]->data, out.data, out.N(), (int)out.HWC(), (int)in.HWC(), s), "nn#flinear"); break;
    case T4K_L_FLATTEN: lazy_copy(x, out); return x;    // a copy in the reference (forward.cu:96); the next layer reads the source
    case T4K_L_DROPOUT:                                 // mask = fresh uniform draws (already drawn up front on the side-stream schedule)
        if (!concurrent()) chk(t4k_dropout_mask(in.grad[4]->data, (long)in.grad[4]->numel, s), "rand");
        /* fall through */
    case T4K_L_RELU: case T4K_L_TANH: case T4K_L_SIGMOID: case T4K_L_SELU: case T4K_L_LEAKYRL: case T4K_L_ELU:
        chk(t4k_activate(fn, x, out.data, in.grad[4]->data, in.xparm, (long)in.numel, s), "nn#factivate"); break;
    case T4K_L_SOFTMAX: chk(t4k_softmax(x, out.data, in.N(), (int)in.HWC(), s), "nn#fsoftmax"); break;
    case T4K_L_LOGSMAX:                                 // _flogsoftmax forward.cu:245-259 (log10 and exp(x) kept: reference bug a-16)
        chk(t4k_logsoftmax(x, out.data, out.N(), (int)out.HWC(), s), "nn#flogsoftmax"); break;
    case T4K_L_AVGPOOL: case T4K_L_MAXPOOL: case T4K_L_MINPOOL:
        chk(t4k_pool(fn, x, out.data, out.N(), in.H(), in.W(), out.H(), out.W(), out.C(), in.stride[0], s), "nn#fpool"); break;
    case T4K_L_BATCHNM:
        chk(t4k_batchnorm_fwd(x, out.data, in.grad[4]->data, in.grad[0]->data, in.grad[1]->data, in.mtum[4]->data,
                              out.N(), out.H() * out.W(), out.C(), s), "nn#fbatchnorm"); break;
    case T4K_L_USAMPLE:                                 // nearest: broadcast each cell to a kxk tile
        chk(t4k_dpool(T4K_L_USAMPLE, out.data, x, in.N(), out.H(), out.W(), in.H(), in.W(), in.C(), in.stride[0], s), "nn#fupsample"); break;
    default: hprintf("nn#fstep layer=%d not supported\n", fn);
    }
    return out.data;
}

// ---------------------------------------------------------------- one-hot / hit / loss
Tensor &Model::onehot() {
    if (hot) return *hot;
    hprintf("Model.onehot not provided by dataset, use nn.onehot= to setup!\n");
    return at(-1);
}
Tensor &Model::onehot(Tensor &t) {
    Tensor &out = at(-1);
    const uint32_t N = out.N(), E = (uint32_t)out.HWC();
    if (hot) { hprintf("WARN: Model.onehot exists, replaced\n"); Store::get().free(*hot); }
    else if (t.N() != N || (uint32_t)t.HWC() != E) { hprintf("Model.onehot dimension is not [%d,1,%d,1]\n", N, E); return t; }
    hot = &t; hit_ = hit(true);
    return *hot;
}
Tensor &Model::onehot(Dataset &d) {                     // loss.cpp:47-72
    Tensor &out = at(-1);
    const uint32_t E = (uint32_t)out.HWC();
    if (!hot) hot = &T4(out.N(), 1, E, 1);
    if ((uint32_t)d.batch_sz < out.N()) hot->zeros();
    chk(t4k_onehot(d.label, hot->data, d.batch_sz, E, stream()), "nn#onehot");
    return *hot;
}
void Model::onehot_hit(Dataset &d) {                    // Model::onehot(Dataset&) + hit_lazy() (forward.cu:57-60) in one launch
    Tensor &out = at(-1);
    const uint32_t E = (uint32_t)out.HWC();
    if (!d.label || d.batch_sz < 1 || (uint32_t)d.batch_sz > out.N()) { onehot(d); hit_lazy(); return; }
    if (!hot) hot = &T4(out.N(), 1, E, 1);
    if ((uint32_t)d.batch_sz < out.N()) hot->zeros();     // short last batch: the rows past it stay zero and count nothing
    if (!hit_pin) { void *p; chk(t4k_host_alloc(&p, 64), "nn#hit"); hit_pin = (int *)p; }
    chk(t4k_onehot_hit(d.label, hot->data, out.data, d.batch_sz, (int)E, hit_pin, stream()), "nn#onehot+hit");
    hit_pending_ = true;
}
void Model::traced_onehot_hit(Dataset &d) {             // the same two steps with the text `1 trace` / `2 trace` add (loss.cpp:47-107): synchronous, on host copies
    Tensor &out = at(-1);
    const uint32_t E = (uint32_t)out.HWC(), N = (uint32_t)d.batch_sz;
    onehot(d);
    std::vector<float> h, o; std::vector<uint32_t> lab(N);
    hot->to_host(h); out.to_host(o);
    if (N) { t4k_memcpy_d2h(lab.data(), d.label, sizeof(uint32_t) * N, stream()); t4k_sync(stream()); }
    NLOG("\n  Model::onehot(ds) {\n");
    if (*trace > 1)
        for (uint32_t n = 0; n < N; n++) {
            std::string l = "    n=" + std::to_string(n) + " {"; char b[16];
            for (uint32_t e = 0; e < E; e++) { snprintf(b, sizeof(b), "%2.0f%c", h[(size_t)n * E + e], e == lab[n] ? '*' : ' '); l += b; }
            hputs(l + "}\n");
        }
    NLOG("  } Model::onehot(ds)");
    NLOG("\n  Model::hit {\n");
    int cnt = 0;
    for (uint32_t n = 0; n < out.N(); n++) {
        const float *on = o.data() + (size_t)n * E, *hn = h.data() + (size_t)n * E;
        uint32_t m = 0; for (uint32_t e = 1; e < E; e++) if (on[e] > on[m]) m = e;       // first maximum
        cnt += (int)hn[m];
        if (*trace > 1) {
            std::string l = "    "; char b[24];
            for (uint32_t e = 0; e < E; e++) { snprintf(b, sizeof(b), "%4.2f%c", on[e], fabsf(hn[e] - 1.0f) < DU_EPS ? (e == m ? '#' : '*') : (e == m ? '<' : ' ')); l += b; }
            snprintf(b, sizeof(b), " n=%d cnt=%d\n", (int)n, cnt); hputs(l + b);
        }
    }
    NLOG("  } Model::hit=%d", cnt);
    hit_ = (int)dp_sum((DU)cnt); hit_pending_ = false; hit_flags_pending_ = false;
}
void Model::hit_lazy() {                                 // count on the GPU now, read it back only if somebody asks (`nn.hit`)
    if (!hot) { hit_ = 0; hit_pending_ = false; return; }
    Tensor &out = at(-1);
    if (!hit_pin) { void *p; chk(t4k_host_alloc(&p, 64), "nn#hit"); hit_pin = (int *)p; }   // device-visible host word: the kernel's store IS the read-back
    chk(t4k_hit(out.data, hot->data, out.N(), (int)out.HWC(), hit_pin, stream()), "nn#hit");
    hit_pending_ = true;
}
// data parallel: `nn.hit` and the loss words report the WHOLE batch when the library owns a communicator - one small
// all-reduce(SUM) of the rank-local value on the VM stream (hit counts are exact in fp32 below 2^24; every rank must run the word).
// Training-mode models only: after `0 trainable` the words report the rank-local value and issue NO collective, so an evaluation
// pass may run on one rank alone (the same rule dp_bn_mode applies to the batch-norm statistics).
DU Model::dp_sum(DU v) {
    if (t4k_comm_world() < 2 || !train) return v;
    if (!hit_dev) { void *p; t4k_malloc(&p, 64); hit_dev = (int *)p; }
    float *d = (float *)hit_dev + 8, r = v;               // second half of the 64-byte scratch
    t4k_memcpy_h2d(d, &r, sizeof(float), stream());
    chk(t4k_allreduce_sum(d, 1, stream()), "allreduce (scalar)");
    t4k_memcpy_d2h(&r, d, sizeof(float), stream()); t4k_sync(stream());
    return r;
}
int Model::hit(bool recalc) {                           // loss.cpp:75-107
    if (recalc) { hit_flags_pending_ = false; hit_lazy(); }
    hit_read_ = true;
    if (hit_flags_pending_) {                             // the forward left one byte per image, in pinned host memory or on the device (see forward): add them up here
        int c = 0; const int n = std::min((int)at(-1).N(), hit_flags_n_);
        std::vector<unsigned char> fl((size_t)std::max(n, 1));
        if (hit_flags_on_dev_) t4k_memcpy_d2h(fl.data(), hit_flags_dev_, (size_t)n, stream());
        t4k_sync(stream());
        if (!hit_flags_on_dev_) memcpy(fl.data(), hit_flags_, (size_t)n);
        for (int i = 0; i < n; i++) c += fl[i] ? 1 : 0;
        hit_ = (int)dp_sum((DU)c); hit_flags_pending_ = false; hit_pending_ = false;
        return hit_;
    }
    if (hit_pending_) {
        t4k_sync(stream());                                // kernel completion makes its store to the pinned word visible (no copy command)
        const int c = *(volatile int *)hit_pin;
        hit_ = (int)dp_sum((DU)c); hit_pending_ = false;
    }
    return hit_;
}
DU Model::loss(Loss op) { return hot ? loss(op, *hot) : 0.0f; }
DU Model::loss(Loss op, Tensor &tgt) {                  // loss.cpp:119-136: non-destructive (works on a copy)
    Tensor &out = at(-1);
    if (out.numel != tgt.numel) {
        hprintf("nn::loss model output shape[%d,%d,%d,%d] != tgt[%d,%d,%d,%d]\n", out.N(), out.H(), out.W(), out.C(), tgt.N(), tgt.H(), tgt.W(), tgt.C());
        return 0;
    }
    if (loss_t) *loss_t = out; else loss_t = &Store::get().copy(out);
    const DU z = loss_t->loss(op, tgt);                  // = sum over the local rows / N_local
    { static const char *opn[] = { "MSE", "BCE", "CE", "NLL" }; NLOG("  Model#loss: %s=%6.3f\n", opn[(int)op & 3], z); }
    const int world = t4k_comm_world();
    return (world < 2 || !train) ? z : SCALAR(dp_sum(z) / (DU)world);   // equal shards (the batch size is fixed by nn.model): mean of the rank means = whole-batch mean
}

// ---------------------------------------------------------------- backprop
Model &Model::broadcast(Tensor &tgt) {                  // backprop.cu:17-29: [N,1] -> [N,HWC]
    Tensor &out = at(-1);
    const uint64_t HWC = out.HWC(); const uint32_t N = out.N();
    if (!hot) hot = &T4(N, 1, (uint32_t)HWC, 1);
    if (tgt.numel < N) hot->zeros();                    // a short target fills its rows only
    chk(t4k_broadcast_rows(tgt.data, hot->data, (int)std::min<uint64_t>(N, tgt.numel), (int)HWC, stream()), "nn#broadcast");
    return *this;
}
Model &Model::backprop() {
    if (hot) return backprop(*hot);
    hprintf("nn#backprop missing onehot vector?\n");
    return *this;
}
Model &Model::backprop(Tensor &tgt) {
    Tensor &out = at(-1);
    if (out.numel != tgt.numel) {                       // _bprep backprop.cu:75-109
        hprintf("Model#bprep: Onehot wrong shape[%d,%d,%d,%d] != [%d,%d,%d,%d], numel=%ld,%ld ", tgt.N(), tgt.H(), tgt.W(), tgt.C(),
               out.N(), out.H(), out.W(), out.C(), (long)tgt.numel, (long)out.numel);
        return *this;
    }
    finalize();
    const double tb0 = trace_ms();
    if (!replay(g_bwd_, tgt.data, (int)train, nullptr)) {
        const bool cap = capturing_;
        run_backward(tgt);
        end_capture(g_bwd_, cap);
    }
    NLOG("\n} Model::backprop %5.2f ms\n", trace_ms() - tb0);
    return *this;
}
// ---- data parallel overlap.  The slab fills tail-first (last layer's dW|dB first, tests/test_gpu_embed.py pins the order).
// Ranges complete on the main stream are queued; once a bucket is full it is all-reduced on the communication stream behind
// an event, concurrently with the backward of the earlier layers.  RCCL serialises the collectives of one communicator in
// issue order whatever stream they are on, and every rank issues the same sequence (same model, same bucket size).
// Off by default: with a one-rank communicator the two cross-stream edges on the main stream (RCCL's own ordering of the final
// reduction behind the early one + the join) cost +16 us per step (0.129 -> 0.145 ms), about what hiding a 400 KB all-reduce
// can win back on 8 GPUs - to be decided with measurements on an 8-GPU box.
int  Model::dp_overlap = (int)env_long("T4_DP_OVERLAP", 0);       // 0 off, 1 on for world > 1, 2 also for a one-rank communicator (tests)
long Model::dp_bucket  = env_long("T4_DP_BUCKET", 16384);          // floats per early all-reduce (64 KiB)
void Model::grads_ready(int i, Tensor &in) {
    if (!(train && gslab && in.grad[2] && in.grad[3] && !in.grad[2]->owns)) return;
    const long off = (long)(in.grad[2]->data - gslab->data);
    const long end = (long)(in.grad[3]->data - gslab->data) + (long)((in.grad[3]->numel + 63) & ~(uint64_t)63);
    if (grad_hook) grad_hook(i, off, end - off, grad_hook_user);
    if (!dp_overlap || dp_mixed_ || dp_pend_lo_ < 0 || t4k_comm_world() < (dp_overlap >= 2 ? 1 : 2) || concurrent() || use_graphs || capturing_) return;
    if (end != dp_pend_lo_ || off < 0 || off >= end) { dp_mixed_ = true; return; }   // not the next range down: leave the rest to `gradient`
    dp_pend_lo_ = off;
    if (dp_done_lo_ - dp_pend_lo_ >= dp_bucket) dp_flush();
}
void Model::dp_flush() {
    if (dp_done_lo_ <= dp_pend_lo_) return;
    if (!comm_s_) { chk(t4k_stream_create(&comm_s_), "comm stream"); t4k_event_create(&dp_ev_[0]); t4k_event_create(&dp_ev_[1]); }
    t4k_event_record(dp_ev_[0], stream()); t4k_stream_wait_event(comm_s_, dp_ev_[0]);            // the range is complete on the main stream
    chk(t4k_allreduce_sum(gslab->data + dp_pend_lo_, dp_done_lo_ - dp_pend_lo_, comm_s_), "allreduce (overlapped)");
    static const bool tr = env_long("T4_DP_TRACE", 0) != 0;
    if (tr) fprintf(stderr, "dp: early all-reduce of slab [%ld, %ld)\n", dp_pend_lo_, dp_done_lo_);
    dp_done_lo_ = dp_pend_lo_; dp_busy_ = true;
}
void Model::dp_begin_backward() {
    if (!gslab) return;
    const long numel = (long)gslab->numel;
    if (dp_done_lo_ >= 0 && dp_done_lo_ < numel) {
        // a second backprop before the optimizer (gradient accumulation) finds ranges that already hold the SUM over ranks:
        // turn them back into a local share (x 1/world, exact for power-of-two worlds) so the final all-reduce of the whole
        // slab gives SUM(first) + SUM(second); no more early reductions in this accumulation window
        if (dp_busy_) { t4k_event_record(dp_ev_[1], comm_s_); t4k_stream_wait_event(stream(), dp_ev_[1]); dp_busy_ = false; }
        chk(t4k_math(T4K_SCALE, gslab->data + dp_done_lo_, 1.0f / (float)t4k_comm_world(), numel - dp_done_lo_, stream()), "dp rescale");
        dp_mixed_ = true;
    }
    dp_done_lo_ = dp_pend_lo_ = numel;
}
// T4_DP_XCHG=0: the slab is all-reduced by a collective of its own (RCCL, or the exchange's generic kernel) in front of the optimizer
static const bool use_xchg = env_flag("T4_DP_XCHG", true);
void Model::dp_finish() {                                // before the update: reduce what is left, join the communication stream
    dp_in_opt_ = false;
    if (!gslab || (t4k_comm_world() <= 0 && !t4k_xchg_active())) return;
    const long numel = (long)gslab->numel;
    const long rest = (dp_done_lo_ >= 0 && dp_done_lo_ <= numel) ? dp_done_lo_ : numel;
    // one-shot peer exchange connected and nothing reduced early: the optimizer launch sums the slab over the ranks itself (t4k_opt_step_dp)
    if (use_xchg && t4k_xchg_active() && rest == numel && !dp_busy_ && !dp_mixed_ && !use_graphs) {
        dp_in_opt_ = true; dp_done_lo_ = dp_pend_lo_ = -1;
        return;
    }
    if (rest > 0) chk(t4k_allreduce_sum(gslab->data, rest, stream()), "allreduce");
    static const bool tr = env_long("T4_DP_TRACE", 0) != 0;
    if (tr) fprintf(stderr, "dp: final all-reduce of slab [0, %ld) of %ld\n", rest, numel);
    if (dp_busy_) { t4k_event_record(dp_ev_[1], comm_s_); t4k_stream_wait_event(stream(), dp_ev_[1]); dp_busy_ = false; }
    dp_done_lo_ = dp_pend_lo_ = -1; dp_mixed_ = false;
}
void Model::run_backward(Tensor &tgt) {
    if (dx0_stale_ && dx0_lin_) materialize_dx0();      // backprop twice without a forward: the layer tensors must hold what the reference's would
    dp_bn_mode(train);
    dp_begin_backward();
    Tensor &out = at(-1);
    t4k_stream_t s = stream();
    const bool fused = use_fusion && !(trace && *trace) && !concurrent();
    int skip = 0;                                       // layers already handled by the prep launch
    const float *dy0 = nullptr;                         // ... and where they left the gradient (default: the output tensor)
    NLOG("Model::bprep input(onehot) numel=%ld OK {\n", (long)tgt.numel);     // _bprep backprop.cu:84-106 (under trace nothing is fused: the loss derivative is a launch of its own)
    switch (at(-2).grad_fn) {
    case T4K_L_SIGMOID: case T4K_L_SOFTMAX: case T4K_L_LOGSMAX:
        if (fused && layer.size() > 3 && at(-3).grad_fn == T4K_L_LINEAR) { prep_tgt_ = &tgt; skip = 1; break; }   // rides in the linear backward launch
        if (fused && layer.size() > 2) {                // out -= target, and the pass-through `in = out` of the last layer, in one launch
            chk(t4k_tt_op2(T4K_SUB, out.data, tgt.data, out.data, at(-2).data, (long)out.numel, s), "bprep"); skip = 1; break;
        }
        /* fall through */
    case T4K_L_LINEAR:
        chk(t4k_tt_op(T4K_SUB, out.data, tgt.data, out.data, (long)out.numel, s), "bprep"); break;
    default: {
        const int lf = at(-2).grad_fn;                  // last op
        const bool mact = lf == T4K_L_RELU || lf == T4K_L_TANH || lf == T4K_L_SELU || lf == T4K_L_LEAKYRL || lf == T4K_L_ELU;
        if (fused && mact && layer.size() > 2 && run_of_[(int)layer.size() - 2] < 0 && at(-2).numel == out.numel) {   // copy + the activation's mask multiply, one launch
            chk(t4k_copy_mask(tgt.data, at(-2).grad[4]->data, out.data, at(-2).data, (long)out.numel, s), "bprep+bactivate"); skip = 1; dy0 = at(-2).data; break;
        }
        chk(t4k_copy(tgt.data, out.data, (long)out.numel, s), "bprep"); break;
    }
    }
    if (trace && *trace) TSHOW(out, true);              // the loss derivative (backprop.cu:104)
    NLOG("}\n");
    NLOG("\nModel::backprop starts trace=%d train=%d {", *trace, (int)train);
    tl_ = trace_ms();
    const float *dy = dy0 ? dy0 : out.data;             // where the gradient w.r.t. the current layer's output lives
    for (int i = (int)layer.size() - 2 - skip, j = skip; i >= 0; i--, j++) {
        Tensor &in = at(i), &o = at(i + 1);
        if (trace && *trace) {                          // backprop.cu:40-58
            const double tt = trace_ms();
            hprintf("\n%6.2f:%3d> %s [%2d,%2d,%2d,%2d] p=%6.3f <= out'\xCE\xA3/n=%6.2f [%2d,%2d,%2d,%2d]", tt - tl_, i, LAYER_NAME[in.grad_fn],
                   in.N(), in.H(), in.W(), in.C(), in.xparm, o.sum() / o.N() / o.C(), o.N(), o.H(), o.W(), o.C());
            tl_ = tt;
        }
        if (fused && use_stack && j > 0) {              // does a sample-resident conv stack END at op i?  [conv + run] x n backward in ONE launch (+ the partial fold)
            if (stack_end_.empty()) {                   // op index of a stack's last op -> its first op (conv layer), built once per finalize
                stack_end_.assign(layer.size(), -1);
                for (int k = 0; k + 1 < (int)layer.size(); ) {
                    t4k_conv_stage stg[3]; int ops = 0;
                    const int ns = at(k).grad_fn == T4K_L_CONV ? stack_at(k, stg, ops) : 0;
                    if (ns >= 2 || (ns == 1 && stack_single_)) { stack_end_[k + ops - 1] = k; k += ops; } else k++;
                }
            }
            if (stack_end_[i] >= 0) {
                const int k0 = stack_end_[i];
                t4k_conv_stage stg[3]; int ops = 0;
                const int ns = stack_at(k0, stg, ops);
                const int bflags = (train ? 1 : 0) | ((k0 < (int)stack_fresh_.size() && stack_fresh_[k0]) ? 0 : 2) | ((use_opt_fold && !grad_hook && !capturing_) ? 4 : 0) | ((use_lazy_dx0 && k0 == 0 && !capturing_ && !use_graphs) ? 8 : 0);
                // (asked first, quietly: without the forward's saved state the whole-image kernel runs, and its windows may not fit the LDS - the per-layer kernels then)
                if (ns > 0 && k0 + ops - 1 == i && t4k_conv_stack_bwd_ok(stg, ns, at(k0).N(), bflags, s) && chk(t4k_conv_stack_bwd(dy, stg, ns, at(k0).N(), bflags, s), "nn#bstack") == T4K_OK) {
                    if (k0 == 0 && use_lazy_dx0 && t4k_conv_stack_dx0_pending(stg[0].O)) {
                        dx0_stale_ = true; at(0).stale_owner = this; if (at(0).grad[4]) at(0).grad[4]->stale_owner = this;
                    }
                    for (int k = i; k >= k0; k--) if (at(k).grad_fn == T4K_L_CONV) grads_ready(k, at(k));      // slab segments complete, last layer first
                    dy = at(k0).data; j += i - k0; i = k0;
                    continue;
                }
            }
        }
        if (fused && j > 0) {                           // does a fused run END at op i?
            int rf = -1;
            for (int k = i; k >= 0 && k > i - 4; k--) if (run_of_[k] >= 0 && k + runs_[run_of_[k]].count - 1 == i) { rf = k; break; }
            if (rf >= 0) {
                const Run &r = runs_[run_of_[rf]];
                Tensor &fin = at(rf), &pin = (r.blk.pool_layer ? at(rf + (r.blk.pre_layer ? 1 : 0)) : fin);
                const int ho = r.blk.pool_layer ? at(rf + (r.blk.pre_layer ? 2 : 1)).H() : pin.H(), wo = r.blk.pool_layer ? at(rf + (r.blk.pre_layer ? 2 : 1)).W() : pin.W();
                chk(t4k_poolblock_bwd(dy, fin.data, &r.blk, fin.N(), pin.H(), pin.W(), ho, wo, pin.C(), s), "nn#brun");
                dy = fin.data; j += i - rf; i = rf;
                continue;
            }
        }
        dy = bstep(i, in, o, dy, j == 0);
        if (skip_next_) {                               // bstep also ran the backward of op i-1 (a lone mask-multiply layer)
            skip_next_ = false;
            grads_ready(i, in);
            if (also_ready_ >= 0) { grads_ready(also_ready_, at(also_ready_)); also_ready_ = -1; }   // a second layer's gradients came out of the same launch
            const int k = skip_cnt_; skip_cnt_ = 1;
            dy = at(i - k).data; i -= k; j += k;
            continue;
        }
        grads_ready(i, in);
        if (trace && *trace && in.has_nan()) { hprintf("nn#backprop Nan %s\n", LAYER_NAME[in.grad_fn]); TSHOW(in, false); TSHOW(o, false); err = true; break; }
        if (trace && *trace > 1) TSHOW(in, true);       // `2 trace`: every layer's dX (backprop.cu:67)
    }
    join();
}
// one layer backward (_bstep backprop.cu:111-140): reads dY at `dy`, leaves dX in in.data (the reference's
// in-place convention) and returns where the previous layer finds it.  Parameter gradients and the
// `in = dX` copies run on the side stream.
const float *Model::bstep(int i, Tensor &in, Tensor &out, const float *dy, bool last) {
    const int fn = in.grad_fn;
    t4k_stream_t s = stream();
    switch (fn) {
    case T4K_L_CONV: {
        Tensor &dx = *in.grad[4];
        const int N = in.N(), H1 = in.H(), W1 = in.W(), C1 = in.C(), H0 = out.H(), W0 = out.W(), C0 = out.C();
        const int K = in.grad[0]->H(), S = in.stride[0], P = in.stride[2];
        if (i == 0 && train && use_lazy_dx0 && use_fusion && !(trace && *trace) && !concurrent() && !capturing_ && !use_graphs && in.grad[2] && in.grad[3]) {
            // the net's first layer: nobody reads dX0 in a training loop - dF | dB now, dX0 when a word asks (materialize_dx0)
            Tensor *holder = nullptr;
            for (int k = 1; k < (int)layer.size() && !holder; k++) if (at(k).data == dy) holder = &at(k);
            if (holder) {
                chk(t4k_conv2d_bwd(in.data, dy, nullptr, in.grad[0]->data, in.grad[2]->data, in.grad[3]->data,
                                   N, H1, W1, C1, H0, W0, C0, K, S, P, 1, s), "nn#bconv dF");
                dx0_stale_ = true; dx0_lin_ = true; dx0_conv_ = true; w0_saved_ = false; dx0_dy_ = dy; dx0_dy_t_ = holder;
                in.stale_owner = this; dx.stale_owner = this; in.grad[0]->stale_owner = this; holder->stale_owner = this;
                return in.data;
            }
        }
        if (!concurrent()) {                            // one stream: dF|dB read x first, then dX lands in the scratch tensor AND over x
            chk(t4k_conv2d_bwd2(in.data, dy, dx.data, in.data, in.grad[0]->data, train ? in.grad[2]->data : nullptr, train ? in.grad[3]->data : nullptr,
                                N, H1, W1, C1, H0, W0, C0, K, S, P, train, s), "nn#bconv");
            return in.data;
        }
        if (train) chk(t4k_conv2d_bwd(in.data, dy, nullptr, in.grad[0]->data, in.grad[2]->data, in.grad[3]->data,
                                      N, H1, W1, C1, H0, W0, C0, K, S, P, 1, fork()), "nn#bconv dF");
        chk(t4k_conv2d_bwd(in.data, dy, dx.data, in.grad[0]->data, nullptr, nullptr, N, H1, W1, C1, H0, W0, C0, K, S, P, 0, s), "nn#bconv dX");
        lazy_copy(dx.data, in);                         // x = dX (overwrite), backprop.cu:185 - after dF has consumed x
        return dx.data;
    }
    case T4K_L_DCONV: {                                 // backprop.cu:137: the conv forward routine gives dX; dF|dB read x first, then `in = dx`
        Tensor &dx = *in.grad[4];
        chk(t4k_dconv2d_bwd(in.data, dy, dx.data, in.grad[0]->data, train ? in.grad[2]->data : nullptr, train ? in.grad[3]->data : nullptr,
                            in.N(), in.H(), in.W(), in.C(), out.H(), out.W(), out.C(), in.grad[0]->H(), in.stride[0], in.stride[2], train, s), "nn#bdconv");
        chk(t4k_copy(dx.data, in.data, (long)in.numel, s), "nn#bdconv in = dx");
        return in.data;
    }
    case T4K_L_LINEAR: {
        if (last) { lazy_copy(dy, in); return dy; }     // linear as the last layer: pass dY (backprop.cu:119-121)
        const int N = in.N(), E0 = (int)out.HWC(), E1 = (int)in.HWC();
        Tensor *gx = concurrent() ? gx_[i] : nullptr;
        if (!gx) {                                      // single stream: reference order (dW reads X, then dX overwrites it)
            const float *tg = prep_tgt_ ? prep_tgt_->data : nullptr;   // backprop's `out -= target` still pending (run_backward)
            prep_tgt_ = nullptr;
            // a lone mask-multiply layer (dropout, relu, ...) right in front of this linear layer: its backward rides along
            const bool fused = use_fusion && !(trace && *trace);
            if (fused && i > 0 && run_of_[i - 1] >= 0 && runs_[run_of_[i - 1]].count == 1 && !runs_[run_of_[i - 1]].blk.pool_layer &&
                runs_[run_of_[i - 1]].blk.pre_layer) {
                Tensor &prev = at(i - 1);
                // ... and when the layer in front of THAT is the linear layer behind a conv stack's flatten (the t4_30a/30e classifier), its backward
                // joins the launch too: every GEMM tile recomputes the rows of dY1 it needs (10 terms each), so nothing waits for the head (t4k_mlp_head_bwd)
                if (tg && use_head_bwd && train && i >= 3 && at(i - 2).grad_fn == T4K_L_LINEAR && !stack_end_.empty() && stack_end_[i - 3] >= 0 &&
                    at(i - 2).grad[2] && at(i - 2).grad[3] &&
                    t4k_mlp_head_bwd_ok(N, (int)at(i - 2).HWC(), E1, E0)) {
                    Tensor &big = at(i - 2);
                    chk(t4k_mlp_head_bwd(in.data, in.grad[0]->data, (float *)dy, tg, at(-2).data, prev.grad[4]->data, prev.data, in.grad[2]->data, in.grad[3]->data,
                                         big.data, big.grad[0]->data, big.grad[2]->data, big.grad[3]->data, N, (int)big.HWC(), E1, E0, s), "nn#bhead+blinear");
                    skip_next_ = true; skip_cnt_ = 2; also_ready_ = i - 2;
                    return big.data;
                }
                if (tg) chk(t4k_loss_linear_bwd(in.data, in.grad[0]->data, (float *)dy, tg, at(-2).data, in.data, prev.grad[4]->data, prev.data,
                                            train ? in.grad[2]->data : nullptr, train ? in.grad[3]->data : nullptr, N, E0, E1, train, s), "nn#bprep+blinear+act");
                else
                chk(t4k_linear_bwd2(in.data, in.grad[0]->data, dy, in.data, prev.grad[4]->data, prev.data, train ? in.grad[2]->data : nullptr,
                                    train ? in.grad[3]->data : nullptr, N, E0, E1, train, s), "nn#blinear+act");
                skip_next_ = true;
                return in.data;
            }
            // a lone activation that multiplies by its derivative mask (relu, tanh, leakyrelu, ... - not a fused run, not the
            // pass-through sigmoid) in front of a layer too large for the head kernel: the multiply rides in the dX epilogue
            auto in_run = [&](int op) { for (int k = op; k >= 0 && k > op - 5; k--) if (run_of_[k] >= 0 && k + runs_[run_of_[k]].count - 1 >= op) return true; return false; };
            if (fused && i > 0 && !tg && !in_run(i - 1)) {
                const int pf = at(i - 1).grad_fn;
                if (pf == T4K_L_RELU || pf == T4K_L_TANH || pf == T4K_L_SELU || pf == T4K_L_LEAKYRL || pf == T4K_L_ELU) {
                    Tensor &prev = at(i - 1);
                    t4k_poolblock b1; memset(&b1, 0, sizeof(b1)); b1.KS = 1;
                    b1.pre_layer = pf; b1.pre_alpha = prev.xparm; b1.pre_mask = prev.grad[4]->data; b1.pre_out = in.data;
                    chk(t4k_linear_block_bwd(in.data, in.grad[0]->data, (float *)dy, nullptr, nullptr, in.data, &b1, prev.data,
                                             train ? in.grad[2]->data : nullptr, train ? in.grad[3]->data : nullptr, N, E0, E1, train, s), "nn#blinear+act");
                    skip_next_ = true; skip_cnt_ = 1;
                    return in.data;
                }
            }
            // a run of two mask-multiply layers (`leakyrelu dropout`) in front: its backward rides along as well
            if (fused && i > 1 && run_of_[i - 2] >= 0 && runs_[run_of_[i - 2]].count == 2 && !runs_[run_of_[i - 2]].blk.pool_layer &&
                !runs_[run_of_[i - 2]].blk.copy_out) {
                const Run &r = runs_[run_of_[i - 2]];
                // the linear layer in front of that run joins the launch when it is the net's first layer or has such a run of its own in front (the GAN
                // discriminator: linear, leakyrelu, dropout, linear, leakyrelu, dropout, linear, sigmoid) - t4k_mlp_block_bwd, see t4k_mlp_head_bwd
                // (measured on the GAN nets, N = 256, 256-wide runs: 28.5 us against 9 + 11.7 us apart - re-reading two 256 KB masks per tile costs more than
                // reading the finished dY1; opt-in with T4_HEAD_BWD=2)
                static const bool runs_too = env_long("T4_HEAD_BWD", 1) >= 2;
                if (tg && use_head_bwd && runs_too && i >= 3 && at(i - 3).grad_fn == T4K_L_LINEAR && (!train || (at(i - 3).grad[2] && at(i - 3).grad[3]))) {
                    Tensor &big = at(i - 3);
                    const Run *r1 = nullptr;
                    if (i >= 5 && run_of_[i - 5] >= 0 && runs_[run_of_[i - 5]].count == 2 && !runs_[run_of_[i - 5]].blk.pool_layer && !runs_[run_of_[i - 5]].blk.copy_out) r1 = &runs_[run_of_[i - 5]];
                    if ((r1 || i == 3) && t4k_mlp_head_bwd_ok(N, (int)big.HWC(), E1, E0)) {
                        chk(t4k_mlp_block_bwd(in.data, in.grad[0]->data, (float *)dy, tg, at(-2).data, &r.blk, at(i - 2).data,
                                              train ? in.grad[2]->data : nullptr, train ? in.grad[3]->data : nullptr,
                                              big.data, big.grad[0]->data, r1 ? &r1->blk : nullptr, r1 ? at(i - 5).data : nullptr,
                                              train ? big.grad[2]->data : nullptr, train ? big.grad[3]->data : nullptr, N, (int)big.HWC(), E1, E0, train, s), "nn#bhead+run+blinear");
                        skip_next_ = true; skip_cnt_ = r1 ? 5 : 3; also_ready_ = i - 3;
                        return r1 ? at(i - 5).data : big.data;
                    }
                }
                chk(t4k_linear_block_bwd(in.data, in.grad[0]->data, (float *)dy, tg, tg ? at(-2).data : nullptr, in.data, &r.blk, at(i - 2).data,
                                         train ? in.grad[2]->data : nullptr, train ? in.grad[3]->data : nullptr, N, E0, E1, train, s), "nn#blinear+run");
                skip_next_ = true; skip_cnt_ = 2;
                return in.data;
            }
            if (tg) chk(t4k_loss_linear_bwd(in.data, in.grad[0]->data, (float *)dy, tg, at(-2).data, in.data, nullptr, nullptr,
                                        in.grad[2]->data, in.grad[3]->data, N, E0, E1, train, s), "nn#bprep+blinear");
            else if (i == 0 && train && use_lazy_dx0 && fused && !capturing_ && !use_graphs && in.grad[2] && in.grad[3] && (long)E0 * E1 >= 16384) {
                // the net's first layer: nobody reads dX0 in a training loop - dW | dB now, dX0 = dY W when a word asks (materialize_dx0)
                Tensor *holder = nullptr;
                for (int k = 1; k < (int)layer.size() && !holder; k++) if (at(k).data == dy) holder = &at(k);
                if (holder) {
                    chk(t4k_linear_bwd(in.data, in.grad[0]->data, dy, nullptr, in.grad[2]->data, in.grad[3]->data, N, E0, E1, 1, s), "nn#blinear dW");
                    dx0_stale_ = true; dx0_lin_ = true; w0_saved_ = false; dx0_dy_ = dy; dx0_dy_t_ = holder;
                    in.stale_owner = this; in.grad[0]->stale_owner = this; holder->stale_owner = this;
                    return in.data;
                }
                chk(t4k_linear_bwd(in.data, in.grad[0]->data, dy, in.data, in.grad[2]->data, in.grad[3]->data, N, E0, E1, train, s), "nn#blinear");
            }
            else chk(t4k_linear_bwd(in.data, in.grad[0]->data, dy, in.data, in.grad[2]->data, in.grad[3]->data, N, E0, E1, train, s), "nn#blinear");
            return in.data;
        }
        if (train) chk(t4k_linear_bwd(in.data, in.grad[0]->data, dy, nullptr, in.grad[2]->data, in.grad[3]->data, N, E0, E1, 1, fork()), "nn#blinear dW");
        chk(t4k_linear_bwd(in.data, in.grad[0]->data, dy, gx->data, nullptr, nullptr, N, E0, E1, 0, s), "nn#blinear dX");
        lazy_copy(gx->data, in);                        // dX lands in X's buffer (backprop.cu:240) once dW has read X
        return gx->data;
    }
    case T4K_L_FLATTEN: case T4K_L_SIGMOID: case T4K_L_SOFTMAX: case T4K_L_LOGSMAX:
        lazy_copy(dy, in); return dy;                   // pass-through (:122,129-131)
    case T4K_L_RELU: case T4K_L_TANH: case T4K_L_SELU: case T4K_L_LEAKYRL: case T4K_L_ELU: case T4K_L_DROPOUT:
        chk(t4k_tt_op(T4K_MUL, dy, in.grad[4]->data, in.data, (long)in.numel, s), "nn#bactivate"); break;
    case T4K_L_AVGPOOL: case T4K_L_MAXPOOL: case T4K_L_MINPOOL:
        chk(t4k_dpool(fn, in.data, dy, out.N(), in.H(), in.W(), out.H(), out.W(), out.C(), in.stride[0], s), "nn#bpool"); break;
    case T4K_L_BATCHNM:
        chk(t4k_batchnorm_bwd(in.grad[0]->data, dy, in.grad[4]->data, in.data, in.grad[2]->data, in.grad[3]->data,
                              in.mtum[4]->data, in.N(), in.H() * in.W(), in.C(), train, s), "nn#bbatchnorm"); break;
    case T4K_L_USAMPLE:                                 // gradient of nearest upsampling = sum over the tile = k*k * avgpool
        chk(t4k_pool(T4K_L_AVGPOOL, dy, in.data, in.N(), out.H(), out.W(), in.H(), in.W(), in.C(), in.stride[0], s), "nn#bupsample");
        in.map(T4K_SCALE, (DU)(in.stride[0] * in.stride[0])); break;
    default: hprintf("nn#bstep layer=%d not supported\n", fn);
    }
    return in.data;
}

// ---------------------------------------------------------------- optimizers
void Model::build_table(Optim op) {                      // one multi-tensor launch replaces 6-8 launch+sync pairs
    std::vector<t4k_param_rec> recs;
    tab_max = 0; tab_chunks = 0;
    for (int i = 0; i + 1 < (int)layer.size(); i++) {
        Tensor &in = at(i);
        for (int k = 0; k < 2; k++) {
            if (!in.mtum[k] || !in.grad[k] || !in.grad[k + 2]) continue;
            t4k_param_rec r;
            r.G = in.grad[k]->data; r.DG = in.grad[k + 2]->data; r.M = in.mtum[k]->data;
            r.V = in.mtum[k + 2] ? in.mtum[k + 2]->data : in.mtum[k]->data;
            r.n = (long)in.grad[k]->numel; r.Nw = (int)in.grad[k]->N(); r.pad = tab_chunks;     // g.N(): C1 for conv filters (quirk a-19); pad = first 1024-element chunk
            tab_chunks += (int)((r.n + 1023) / 1024);
            recs.push_back(r); tab_max = std::max(tab_max, r.n);
        }
    }
    if (tab_dev) t4k_free(tab_dev);
    tab_n = (int)recs.size(); tab_kind = op; tab_dev = nullptr; tab_host_ = recs;
    if (tab_n) {
        t4k_malloc(&tab_dev, recs.size() * sizeof(t4k_param_rec));
        t4k_memcpy_h2d(tab_dev, recs.data(), recs.size() * sizeof(t4k_param_rec), stream()); t4k_sync(stream());
    }
}
Model &Model::gradient(const char *nm, Optim op, DU lr, DU b1, DU b2, DU wd) {   // gradient.cu:63-126
    NLOG("\nModel::%s starts (%s) batch_sz=%d, lr=%7.4f, mtum/b1=%6.3f, b2=%6.3f {\n", nm, train ? "trainning" : "testing", at(1).N(), lr, b1, b2);
    finalize();
    const bool first = (iter++ == 0 && epoch == 0);
    if (first || tab_kind != op) {                      // grad_alloc gradient.cu:19-59 (+ re-allocation when the optimizer is switched mid-run,
        if (first) NLOG("  #grad_alloc {\n");
        for (int i = 0; i + 1 < (int)layer.size(); i++) {   //   where the reference would read the SGD alias / a null V)
            Tensor &in = at(i); Tensor *w = in.grad[0], *b = in.grad[1];
            if (first) NLOG("    %3d> %8s w,b[%d,%d] mtum=<pool offsets>\n", i, LAYER_NAME[in.grad_fn], w ? 1 : 0, b ? 1 : 0);   // (the reference prints its pool offsets of m / v here)
            Tensor *g[2] = { w, b };
            for (int k = 0; k < 2; k++) {
                if (!g[k]) continue;
                if (op == OPTI_SGD) { if (!in.mtum[k]) in.mtum[k] = g[k]; continue; }
                if (!in.mtum[k] || in.mtum[k] == g[k]) in.mtum[k] = &T4(g[k]->N(), g[k]->H(), g[k]->W(), g[k]->C()).zeros();
                if (op != OPTI_SGDM && !in.mtum[k + 2]) in.mtum[k + 2] = &T4(g[k]->N(), g[k]->H(), g[k]->W(), g[k]->C()).zeros();
            }
        }
        if (first) NLOG("  } #grad_alloc\n");
        build_table(op);
    }
    if (!train) return *this;
    if (!tab_dev || tab_kind != op) build_table(op);
    // trace levels (gradient.cu:66-79,95-99): per layer, per parameter tensor the sums of the parameter and of its gradient before the update and the
    // parameter's sum after it; `2 trace`: small tensors (< 256 elements) dumped before and after.  The update itself is ONE launch over all tensors
    // here, so the "before" halves are collected first and the text is assembled behind the launch (the tensors are independent: same numbers).
    struct TrLine { int layer; char k; Tensor *g, *dg; std::string head, before; };
    std::vector<TrLine> trl;
    const double tg0 = trace_ms();
    if (trace && *trace)
        for (int i = 0; i + 1 < (int)layer.size(); i++) {
            Tensor &in = at(i);
            for (int k = 0; k < 2; k++) {
                if (!in.mtum[k] || !in.grad[k] || !in.grad[k + 2]) continue;
                Tensor &g = *in.grad[k], &dg = *in.grad[k + 2];
                char b[160]; const char c = k ? 'b' : 'w';
                snprintf(b, sizeof(b), "     %c[%2d,%2d,%2d,%2d] \xCE\xA3=%6.3f - %6.3f", c, g.N(), g.H(), g.W(), g.C(), g.sum(), dg.sum());
                TrLine t{ i, c, &g, &dg, b, "" };
                if (*trace > 1 && g.numel < 256) t.before = std::string("\nbefore ") + c + " =" + fmt_dump(g) + "\nbefore d" + c + "=" + fmt_dump(dg);
                trl.push_back(t);
            }
        }
    const int kind = (op == OPTI_ADAM) ? 1 : (op == OPTI_ADAMW ? 2 : 0);
    const float p[4] = { lr, b1, b2, wd };
    // data parallel: every rank holds a shard of the batch; SUM the gradient slab (raw batch sums, quirk a-19) in-order
    // on the VM stream right before the update, so N ranks x batch B reproduce one rank x batch N*B
    dp_finish();
    if (dx0_stale_ && dx0_lin_ && !w0_saved_) {          // a deferred dX0 = dY W needs the weights of its backward: the update launch leaves them in w0_save_
        Tensor &w0 = *at(0).grad[0];
        if (!w0_save_) w0_save_ = &T4(w0.N(), w0.H(), w0.W(), w0.C());
        // the snapshot rides in THIS model's next t4k_opt_step launch; an update served by a graph replay (arguments baked at capture) would never take it:
        // there the copy is a launch of its own (the request is a process-global one-shot)
        if (use_graphs || capturing_) chk(t4k_memcpy_d2d(w0_save_->data, w0.data, sizeof(float) * w0.numel, stream()), "nn#w0 copy");
        else t4k_opt_snapshot(w0.data, w0_save_->data);
        w0_saved_ = true;
    }
    if (!replay(g_opt_, tab_dev, (int)op, p)) {
        const bool cap = capturing_;
        if (dp_in_opt_) chk(t4k_opt_step_dp(kind, (const t4k_param_rec *)tab_dev, tab_host_.data(), tab_n, tab_chunks, lr, b1, b2, wd, gslab->data, (long)gslab->numel, stream()), nm);
        else chk(t4k_opt_step(kind, (const t4k_param_rec *)tab_dev, tab_host_.data(), tab_n, tab_chunks, lr, b1, b2, wd, stream()), nm);   // one workgroup per 1024 parameters (+ a conv stack's deferred partial fold)
        dp_in_opt_ = false;
        end_capture(g_opt_, cap);
    }
    if (t4k_opt_snapshot_pending()) {                    // must not happen (the launch above consumes it); never leave the request for another model's update
        t4k_opt_snapshot(nullptr, nullptr); w0_saved_ = false;
        hprintf("nn#%s: weight snapshot not taken - the first layer's deferred dX is no longer available\n", nm);
    }
    if (trace && *trace) {
        size_t q = 0;
        for (int i = 0; i + 1 < (int)layer.size(); i++) {
            hprintf("  %d> %s\n", i, LAYER_NAME[at(i).grad_fn]);
            for (; q < trl.size() && trl[q].layer == i; q++) {
                const TrLine &t = trl[q];
                hputs(t.head);
                if (!t.before.empty()) hputs(t.before + "\nafter  " + t.k + " =" + fmt_dump(*t.g) + "\nafter  d" + t.k + "=" + fmt_dump(*t.dg) + "\n");
                hprintf(" => %c\xCE\xA3=%6.3f\n", t.k, t.g->sum());
            }
        }
    }
    NLOG("} Model::%s %5.2f ms\n", nm, trace_ms() - tg0);
    return *this;
}
Model &Model::sgd(DU lr, DU b) { return gradient("sgd", ZEQ(b) ? OPTI_SGD : OPTI_SGDM, lr, iter ? b : 0.0f, 0, 0); }   // `_iter ? b : 0` gradient.cu:139
Model &Model::adam(DU lr, DU b1, DU b2) { return gradient("adam", OPTI_ADAM, lr, b1, b2, 0); }
Model &Model::adamw(DU lr, DU wd, DU b1, DU b2) { return gradient("adamw", OPTI_ADAMW, lr, b1, b2, wd); }

// ---------------------------------------------------------------- persistence (.t4 model file, aio_model.cpp)
// Layout written by the reference: a `\\ tensorForth v4.0 model` line, one `<params><layer name>` line per layer, a blank
// line, then for every conv / linear (w, b) and batchnorm (w) tensor a `--- w.<layer>` marker line followed by the raw
// fp32 values, and a closing `---`.  Loading into an already built model skips the layer section and reads the blobs in
// layer order (the reference's own reload of the layer section is unfinished: `_parm` text is not Forth source).
int model_save(Model &m, const char *fname) {
    FILE *f = fopen(fname, "wb");
    if (!f) { hprintf("} => failed to open for output\n"); return 1; }
    fprintf(f, "\\ tensorForth v4.0 model\n");
    const int L = (int)m.layer.size();
    for (int i = 0; i + 1 < L; i++) { Tensor &in = m.at(i), &out = m.at(i + 1); fprintf(f, "%s%s\n", fmt_parm(in, out).c_str(), LAYER_NAME[in.grad_fn]); }
    std::vector<float> h;
    auto dump = [&](char pn, const char *nm, Tensor &t) {
        fprintf(f, "\n--- %c.%s\n", pn, nm);
        t.to_host(h); fwrite(h.data(), sizeof(float), t.numel, f);
    };
    for (int i = 0; i + 1 < L; i++) {
        Tensor &in = m.at(i); const int fn = in.grad_fn;
        if ((fn == T4K_L_CONV || fn == T4K_L_LINEAR || fn == T4K_L_DCONV) && in.grad[0] && in.grad[1]) { dump('w', LAYER_NAME[fn], *in.grad[0]); dump('b', LAYER_NAME[fn], *in.grad[1]); }   // (the reference skips dconv2d blobs, aio_model.cpp:170: a net it cannot run)
        else if (fn == T4K_L_BATCHNM && in.grad[0]) dump('w', LAYER_NAME[fn], *in.grad[0]);
    }
    fprintf(f, "\n---\n");
    fclose(f);
    return 0;
}
int model_load(Model &m, const char *fname) {
    FILE *f = fopen(fname, "rb");
    if (!f) { hprintf("} => failed to open for input\n"); return 1; }
    auto getline_ = [&](std::string &line) { line.clear(); int c; bool any = false; while ((c = fgetc(f)) != EOF) { any = true; if (c == '\n') break; line.push_back((char)c); } return any; };
    std::string line;
    if (m.layer.size() <= 1) {                           // nothing to load into.  The reference copies the file's layer section + " nn.load <file>" into the
        fclose(f); return 2;                             // input buffer (aio_model.cpp:183-204), which System::readline clears before any VM reads it (sys.cpp:101-108;
    }                                                    // `nn.load` is no word either): nothing is printed, nothing is loaded - observed on the reference's own VM
    while (getline_(line) && line.length()) {}           // skip the layer section (model already built)
    std::vector<float> h;
    int err = 0;
    auto rd = [&](Tensor &t) {
        while (getline_(line) && !line.length()) {}      // skip blank lines
        if (line.size() < 3 || line[0] != '-' || line[1] != '-' || line[2] != '-') { hprintf(" model format error"); err = 1; return; }   // aio_model.cpp:211 (no newline there either)
        h.resize(t.numel);
        if (fread(h.data(), sizeof(float), t.numel, f) != t.numel) { hprintf(" model format error"); err = 1; return; }
        t.from_host(h.data(), t.numel);
    };
    const int L = (int)m.layer.size();
    for (int i = 0; i + 1 < L && !err; i++) {
        Tensor &in = m.at(i); const int fn = in.grad_fn;
        if ((fn == T4K_L_CONV || fn == T4K_L_LINEAR || fn == T4K_L_DCONV) && in.grad[0] && in.grad[1]) { rd(*in.grad[0]); if (!err) rd(*in.grad[1]); }
        else if (fn == T4K_L_BATCHNM && in.grad[0]) rd(*in.grad[0]);
    }
    if (!err) {                                          // the blob list must end here: a closing `---` line and nothing else
        while (getline_(line) && !line.length()) {}
        if (line != "---") { hprintf(" model format error (layers of the file and of the model differ)\n"); err = 1; }
    }
    fclose(f);
    return err;
}

void Model::free_all() {
    invalidate();
    if (current == this) current = nullptr;
    t4k_sync(stream());
    if (side_) { t4k_stream_destroy(side_); side_ = nullptr; }
    if (comm_s_) { t4k_stream_destroy(comm_s_); comm_s_ = nullptr; for (auto &e : dp_ev_) if (e) { t4k_event_destroy(e); e = nullptr; } }
    for (auto e : ev_) if (e) t4k_event_destroy(e);
    ev_.clear();
    for (Tensor *t : gx_) if (t) Store::get().free(*t);
    gx_.clear();
    if (gslab) { Store::get().free(*gslab); gslab = nullptr; }
    if (w0_save_) { Store::get().free(*w0_save_); w0_save_ = nullptr; }
    for (int i = 0; i + 1 < (int)layer.size(); i++) if (at(i).grad_fn == T4K_L_CONV) t4k_conv_stack_release(at(i + 1).data);   // what a stack forward saved for its banded backward
    for (int i = (int)layer.size() - 1; i >= 0; i--) Store::get().free(*layer[i]);
    layer.clear();
    if (hot) { Store::get().free(*hot); hot = nullptr; }
    if (loss_t) { Store::get().free(*loss_t); loss_t = nullptr; }
    if (tab_dev) { t4k_free(tab_dev); tab_dev = nullptr; }
    if (hit_dev) { t4k_free(hit_dev); hit_dev = nullptr; }
    if (hit_pin) { t4k_sync(stream()); t4k_host_free(hit_pin); hit_pin = nullptr; }
    if (hit_flags_) { t4k_sync(stream()); t4k_host_free(hit_flags_); t4k_free(hit_flags_dev_); hit_flags_ = hit_flags_dev_ = nullptr; hit_flags_n_ = 0; }
}

Model &Store::model(int *trace) { Model *m = new Model(); m->type = T_MODEL; m->trace = trace; put(m); return *m; }
Dataset &Store::dataset(uint32_t batch) {
    Dataset *d = new Dataset(); d->type = T_DATASET; d->rank = 4; d->shape[3] = batch; d->owns = false; d->data = nullptr;
    put(d); return *d;
}

} // namespace t4
