// dataset.cpp - corpus readers (MNIST IDX, CIFAR-10 binary) and the batch feed.
// Restates src/ld/{loader,mnist,cifar10}.cpp and Dataset::fetch/_load (src/mu/dataset.cu:64-158):
// the u8 batch is copied to HBM once and normalised ON the GPU (t4k_u8_normalize) instead of a
// host loop into a pageable std::vector followed by a blocking H2D.
#include "t4.h"
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>

namespace t4 {

static std::map<std::string, Corpus *> &corpora() {      // Loader::init src/ld/loader.cpp:31-46
    static std::map<std::string, Corpus *> m;
    if (m.empty()) {
        auto mk = [&](const char *nm, const char *d, const char *l, bool cifar) {
            Corpus *c = new Corpus(); c->name = nm; c->f_data = d; c->f_label = l ? l : ""; c->cifar = cifar; m[nm] = c;
        };
        mk("mnist_train", "./data/MNIST/raw/train-images-idx3-ubyte", "./data/MNIST/raw/train-labels-idx1-ubyte", false);
        mk("mnist_test",  "./data/MNIST/raw/t10k-images-idx3-ubyte",  "./data/MNIST/raw/t10k-labels-idx1-ubyte", false);
        mk("cifar10_train", "./data/CIFAR10/cifar-10-batches-bin/data_batch.bin", nullptr, true);
        mk("cifar10_test",  "./data/CIFAR10/cifar-10-batches-bin/test_batch.bin", nullptr, true);
    }
    return m;
}
static uint32_t be32(FILE *f) { uint8_t b[4] = {0, 0, 0, 0}; if (fread(b, 1, 4, f) != 4) return 0; return (b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3]; }

bool Corpus::init(int batch) {
    cancel_ahead();
    if (N != batch) for (int sl = 0; sl < 2; sl++) { if (pix[sl]) { t4k_host_free(pix[sl]); pix[sl] = nullptr; } if (lab[sl]) { t4k_host_free(lab[sl]); lab[sl] = nullptr; } }
    N = batch; eof = false; batch_sz = 0;
    if (fd) { fclose(fd); fd = nullptr; } if (fl) { fclose(fl); fl = nullptr; }
    fd = fopen(f_data.c_str(), "rb");
    if (!fd) { hprintf("failed to open file %s\n", f_data.c_str()); return false; }
    if (cifar) {                                         // 1 label byte + 3x32x32 planar bytes per sample
        H = W = 32; C = 3;
        fseek(fd, 0, SEEK_END); corpus_sz = (int)(ftell(fd) / 3073); fseek(fd, 0, SEEK_SET);
        return true;
    }
    fl = fopen(f_label.c_str(), "rb");
    if (!fl) { hprintf("failed to open file %s\n", f_label.c_str()); return false; }
    be32(fl); const uint32_t n1 = be32(fl);              // label magic 0x0801, count
    be32(fd); const uint32_t n = be32(fd); H = be32(fd); W = be32(fd); C = 1;   // image magic 0x0803
    if (n != n1) { hprintf("Mnist::init label count %d != image count %d\n", n1, n); return false; }
    corpus_sz = n;
    return true;
}
// One persistent reader thread per corpus (spawning a thread per batch costs more than the read): the main thread
// posts (batch, slot), the reader fills the pinned slot and signals.
struct Reader {
    std::thread th; std::mutex mu; std::condition_variable cv;
    int req_bid = -1, req_slot = 0, result = 0; bool busy = false, quit = false;
};
static void reader_main(Corpus *c, Reader *r) {
    std::unique_lock<std::mutex> lk(r->mu);
    for (;;) {
        r->cv.wait(lk, [r] { return r->quit || (r->busy && r->req_bid >= 0); });
        if (r->quit) return;
        const int bid = r->req_bid, slot = r->req_slot;
        r->req_bid = -1;
        lk.unlock();
        const int n = c->read_into(bid, slot);
        lk.lock();
        r->result = n; r->busy = false;
        r->cv.notify_all();
    }
}
void Corpus::cancel_ahead() {                            // wait for an outstanding read-ahead and forget it
    if (worker) { Reader *r = (Reader *)worker; std::unique_lock<std::mutex> lk(r->mu); r->cv.wait(lk, [r] { return !r->busy; }); }
    ahead_bid = -1; ahead_n = 0;
}
void Corpus::rewind() { eof = false; cancel_ahead(); }
int Corpus::read_into(int bid, int slot) {               // returns the number of samples read (0 at end of corpus)
    const long off = (long)N * bid;
    if (off >= corpus_sz) return 0;
    const size_t cell = (size_t)H * W * C;
    uint8_t *data = pix[slot]; uint32_t *label = lab[slot];
    size_t n = 0;
    if (cifar) {
        std::vector<uint8_t> buf((size_t)N * 3073);
        fseek(fd, off * 3073, SEEK_SET);
        n = fread(buf.data(), 1, buf.size(), fd) / 3073;
        const size_t HW = (size_t)H * W;
        for (size_t i = 0; i < n; i++) {                 // planar RGB -> HWC (cifar10.cpp:118-126)
            const uint8_t *bp = &buf[i * 3073]; label[i] = bp[0]; bp++;
            uint8_t *dp = &data[i * cell];
            for (size_t j = 0; j < HW; j++) { *dp++ = bp[j]; *dp++ = bp[HW + j]; *dp++ = bp[2 * HW + j]; }
        }
    } else {
        std::vector<uint8_t> l8(N);
        fseek(fl, 8 + off, SEEK_SET);
        const size_t nl = fread(l8.data(), 1, N, fl);
        fseek(fd, 16 + off * (long)cell, SEEK_SET);
        n = fread(data, 1, (size_t)N * cell, fd) / cell;
        if (nl != n) { hprintf("Mnist::fetch #label=%d != #image=%d\n", (int)nl, (int)n); return 0; }
        for (size_t i = 0; i < n; i++) label[i] = l8[i];
    }
    return (int)n;
}
bool Corpus::fetch(int bid) {
    const long off = (long)N * bid;
    if (eof || off >= corpus_sz) { hprintf("%s::fetch EOF reached (needs rewind)\n", cifar ? "Cifar10" : "Mnist"); eof = true; return false; }
    const size_t cell = (size_t)H * W * C;
    for (int s = 0; s < 2; s++) {
        if (!pix[s]) { void *p; t4k_host_alloc(&p, (size_t)N * cell); pix[s] = (uint8_t *)p; t4k_host_alloc(&p, sizeof(uint32_t) * N); lab[s] = (uint32_t *)p; t4k_event_create(&copied[s]); }
    }
    const int slot = bid & 1;
    int n;
    if (ahead_bid == bid && worker) {                    // read ahead by the reader thread
        Reader *r = (Reader *)worker; std::unique_lock<std::mutex> lk(r->mu);
        r->cv.wait(lk, [r] { return !r->busy; }); n = r->result;
    } else { cancel_ahead(); if (copied[slot]) t4k_event_sync(copied[slot]); n = read_into(bid, slot); }
    ahead_bid = -1;
    cur_slot = slot; batch_sz = n;
    if (off + (long)n >= corpus_sz) eof = true;
    else {                                               // read batch bid+1 into the other slot while the GPU works on this one
        const int nslot = slot ^ 1, nbid = bid + 1;
        if (copied[nslot]) t4k_event_sync(copied[nslot]);    // that slot's previous H2D copies (batch bid-1) have long finished
        ahead_bid = nbid;
        if (!worker) { Reader *r = new Reader(); worker = r; r->th = std::thread(reader_main, this, r); r->th.detach(); }
        Reader *r = (Reader *)worker;
        { std::lock_guard<std::mutex> lk(r->mu); r->req_bid = nbid; r->req_slot = nslot; r->busy = true; }
        r->cv.notify_all();
    }
    return n > 0;
}

int Dataset::fetch(const char *ds_name, bool rewind) {   // dataset.cu:64-121
    if (ds_name) {
        auto it = corpora().find(ds_name);
        if (it == corpora().end()) { hprintf("  } dataset#fetch => not found in Loader\n"); return -1; }
        cp = it->second;
        if (!cp->init(N())) { hprintf("  } dataset#fetch => corpus init failed!\n"); return -2; }
        dataset_size = cp->corpus_sz;
        numel = (uint64_t)cp->N * cp->H * cp->W * cp->C;
        rank = 4; shape[0] = cp->H; shape[1] = cp->W; shape[2] = cp->C; shape[3] = cp->N;
    }
    if (!cp) { hprintf("  } dataset#fetch => not found in Loader\n"); return -1; }
    if (rewind) { cp->rewind(); batch_id = done = 0; }
    die_if_no_backend();
    if (!cp->fetch(batch_id)) { hprintf("  } dataset#fetch => corpus fetch failed\n"); return -3; }
    const int n = batch_sz = cp->batch_sz;
    done = cp->eof;
    die_if_no_backend();
    if (!data)    { void *p; t4k_malloc(&p, sizeof(float) * numel); data = (float *)p; t4k_memset(data, 0, sizeof(float) * numel, stream()); }
    if (!label)   { void *p; t4k_malloc(&p, sizeof(uint32_t) * N()); label = (uint32_t *)p; }
    if (!raw_dev) { void *p; t4k_malloc(&p, numel); raw_dev = (uint8_t *)p; }
    const long NX = (long)n * HWC();
    // The kernels read the pinned staging slot directly over the fabric (hipHostMalloc memory is device visible): no
    // DMA-engine copy, so no cross-engine synchronisation bubble in front of `forward` (measured: 2 H2D copies ~ 40 us/step)
    chk(t4k_stage_batch(cp->cur_pix(), data, NX, mean, scale, cp->cur_lab(), label, n, stream()), "dataset#load");   // (x - mean) * scale and the labels, one launch
    t4k_event_record(cp->copied[cp->cur_slot], stream());    // the staging slot may be refilled once these kernels are done
    batch_id++;
    return 0;
}

} // namespace t4
