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

bool Corpus::init(int batch, bool trace) {
    idle();
    if (N != batch) for (int sl = 0; sl < 2; sl++) { if (pix[sl]) { t4k_host_free(pix[sl]); pix[sl] = nullptr; } if (lab[sl]) { t4k_host_free(lab[sl]); lab[sl] = nullptr; } }
    N = batch;
    if (fd) { fclose(fd); fd = nullptr; } if (fl) { fclose(fl); fl = nullptr; }
    fd = fopen(f_data.c_str(), "rb");
    if (!fd) { hprintf("failed to open file %s\n", f_data.c_str()); return false; }
    if (cifar) {                                         // 1 label byte + 3x32x32 planar bytes per sample
        H = W = 32; C = 3;
        fseek(fd, 0, SEEK_END); corpus_sz = (int)(ftell(fd) / 3073); fseek(fd, 0, SEEK_SET);
        if (trace) hprintf("\tCIFAR-10 samples: [%d][%d,%d,%d]\n", corpus_sz, H, W, C);      // cifar10.cpp:44
        return true;
    }
    fl = fopen(f_label.c_str(), "rb");
    if (!fl) { hprintf("failed to open file %s\n", f_label.c_str()); return false; }
    const uint32_t x1 = be32(fl), n1 = be32(fl);         // label magic 0x0801, count
    if (trace) hprintf("\tMNIST label: magic=%08x => [%d]\n", x1, n1);                             // mnist.cpp:43
    const uint32_t x0 = be32(fd), n = be32(fd); H = be32(fd); W = be32(fd); C = 1;   // image magic 0x0803
    if (trace) hprintf("\tMNIST image: magic=%08x => [%d][%d,%d,%d]\n", x0, n, H, W, C);            // mnist.cpp:51
    if (n != n1) { hprintf("Mnist::init label count %d != image count %d\n", n1, n); return false; }
    corpus_sz = n;
    return true;
}
// One persistent reader thread per corpus (spawning a thread per batch costs more than the read): the main thread queues
// batch numbers, the reader waits for the launch that last read the pinned slot, fills it and signals.
struct Reader {
    std::thread th; std::mutex mu; std::condition_variable cv;
    int queue[2] = {-1, -1};                             // at most one request per slot
    bool busy = false;
};
static void reader_main(Corpus *c, Reader *r) {
    std::unique_lock<std::mutex> lk(r->mu);
    for (;;) {
        r->cv.wait(lk, [r] { return r->queue[0] >= 0 || r->queue[1] >= 0; });
        const int q = (r->queue[0] >= 0 && (r->queue[1] < 0 || r->queue[0] < r->queue[1])) ? 0 : 1;   // lower batch first
        const int bid = r->queue[q], slot = bid & 1;
        r->busy = true;
        const bool wait_ev = c->pin_wait[slot]; c->pin_wait[slot] = false;
        lk.unlock();
        if (wait_ev) t4k_event_wait(c->pin_ev[slot]);    // the staging launch of the batch this slot held (reader thread: the plain wait, t4k.h)
        const int n = c->read_into(bid, slot);
        lk.lock();
        c->slot_n[slot] = n; r->queue[q] = -1; r->busy = false;
        r->cv.notify_all();
    }
}
void Corpus::ensure_slots() {
    const size_t cell = (size_t)H * W * C;
    for (int s = 0; s < 2; s++)
        if (!pix[s]) { void *p; t4k_host_alloc(&p, (size_t)N * cell); pix[s] = (uint8_t *)p; t4k_host_alloc(&p, sizeof(uint32_t) * N); lab[s] = (uint32_t *)p;
                       if (!pin_done[s]) t4k_event_create(&pin_done[s]); }
}
void Corpus::idle() {                                    // wait for outstanding reads and forget what the slots hold
    if (worker) { Reader *r = (Reader *)worker; std::unique_lock<std::mutex> lk(r->mu); r->cv.wait(lk, [r] { return !r->busy && r->queue[0] < 0 && r->queue[1] < 0; }); }
    for (int s = 0; s < 2; s++) { if (pin_wait[s] && pin_ev[s]) t4k_event_sync(pin_ev[s]); pin_wait[s] = false; slot_bid[s] = -1; slot_n[s] = 0; }
}
void Corpus::settle() {                                  // as idle(), but the slots keep what they hold: a `rewind` at the end of an epoch finds batches 0 and 1 already read
    if (worker) { Reader *r = (Reader *)worker; std::unique_lock<std::mutex> lk(r->mu); r->cv.wait(lk, [r] { return !r->busy && r->queue[0] < 0 && r->queue[1] < 0; }); }
    for (int s = 0; s < 2; s++) { if (pin_wait[s] && pin_ev[s]) t4k_event_sync(pin_ev[s]); pin_wait[s] = false; }
}
void Corpus::request(int bid) {                          // the staging launch of the batch the slot holds now must have been ISSUED
    if (bid < 0 || bid >= n_batches() || slot_bid[bid & 1] == bid) return;
    ensure_slots();
    if (!worker) { Reader *r = new Reader(); worker = r; r->th = std::thread(reader_main, this, r); r->th.detach(); }
    Reader *r = (Reader *)worker;
    { std::unique_lock<std::mutex> lk(r->mu);
      r->cv.wait(lk, [&] { return r->queue[bid & 1] < 0; });            // an earlier request for this slot (unusual word order) finishes first
      slot_bid[bid & 1] = bid; r->queue[bid & 1] = bid; }
    r->cv.notify_all();
}
int Corpus::wait_batch(int bid) {
    if (bid < 0 || bid >= n_batches()) return 0;
    const int slot = bid & 1;
    ensure_slots();
    Reader *r = (Reader *)worker;
    if (slot_bid[slot] != bid) {                         // nobody was asked to: cold start / after a rewind - read here
        if (r) { std::unique_lock<std::mutex> lk(r->mu); r->cv.wait(lk, [r, slot] { return r->queue[slot] < 0; }); }
        if (pin_wait[slot]) { t4k_event_sync(pin_ev[slot]); pin_wait[slot] = false; }
        slot_bid[slot] = bid; slot_n[slot] = read_into(bid, slot);
        return slot_n[slot];
    }
    if (r) { std::unique_lock<std::mutex> lk(r->mu); r->cv.wait(lk, [r, slot] { return r->queue[slot] < 0; }); }   // a request leaves the queue when its read is done
    return slot_n[slot];
}
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

// The feed.  Batch b of an epoch lives in dbuf[b % RING] / lbuf[b % RING].  fetch(b) makes batch b current (swaps `data` / `label`: a
// pointer swap once the pipeline runs), then issues the staging launch of batch b + 1 - (x - mean) * scale and the labels, read straight out
// of the pinned slot over the fabric - on a side stream, and asks the reader thread for batch b + 2.  Ordering, with as few runtime calls
// as it takes (the fed step is 5 launches of ~50 us: every event call on the host shows):
//   * the side stream overwrites the buffer of batch b + 1 - RING: every MARK_EVERY-th fetch records a mark on the main stream (it stands
//     behind every launch that read batches below that fetch's) and the side stream waits for it ONCE (stream order carries it to all later
//     staging launches).  The newest mark is at most MARK_EVERY fetches old, so it covers batches <= b - MARK_EVERY - 1 >= b + 1 - RING;
//   * the model must not read dbuf[b % RING] before its staging launch is done: the HOST waits for that launch's event - issued one whole
//     step earlier, so the wait is a query; no cross-stream edge on the step's stream (measured in round 3: +8 us per edge);
//   * the reader refills pinned slot b & 1 after the event of the launch that read it (waited for on the reader thread).
static t4k_stream_t feed_stream() {
    static t4k_stream_t s = nullptr; static bool tried = false;
    if (!tried) { tried = true; if (t4k_stream_create_plain(&s) != T4K_OK) s = nullptr; }
    return s;
}
static const bool g_prefetch = env_flag("T4_FEED_PREFETCH", true);
void Dataset::release_ring() {
    for (int i = 0; i < RING; i++) {
        if (staged[i] && dev_bid[i] >= 0) t4k_event_sync(staged[i]);
        if (dbuf[i]) t4k_free(dbuf[i]);
        if (lbuf[i]) t4k_free(lbuf[i]);
        dbuf[i] = nullptr; lbuf[i] = nullptr; dev_bid[i] = -1;
        if (staged[i]) { t4k_event_destroy(staged[i]); staged[i] = nullptr; }      // (Corpus::idle has dropped every reference to them)
    }
    if (mark) { t4k_event_destroy(mark); mark = nullptr; }
    mark_bid = -1;
    data = nullptr; label = nullptr; ring_numel = 0;
}
int *Dataset::trace = nullptr;
int Dataset::fetch(const char *ds_name, bool rewind) {   // dataset.cu:64-121
    const bool tr = trace && *trace;                     // the words' trace level (sys.cpp:171-189 hands it down): the reference's text of a fetch
    if (tr) hprintf("  dataset#fetch %s batch[%d] {\n", ds_name ? ds_name : (rewind ? "rewind" : ""), batch_id);
    if (ds_name) {
        auto it = corpora().find(ds_name);
        if (it == corpora().end()) { hprintf("  } dataset#fetch => not found in Loader\n"); return -1; }
        cp = it->second;
        if (!cp->init(N(), tr)) { hprintf("  } dataset#fetch => corpus init failed!\n"); return -2; }
        dataset_size = cp->corpus_sz;
        numel = (uint64_t)cp->N * cp->H * cp->W * cp->C;
        rank = 4; shape[0] = cp->H; shape[1] = cp->W; shape[2] = cp->C; shape[3] = cp->N;
        rewind = true;
    }
    if (!cp) { hprintf("  } dataset#fetch => not found in Loader\n"); return -1; }
    die_if_no_backend();
    if (rewind) {                                        // also after `normalize`: whatever was staged ahead on the DEVICE is void (the pinned raw bytes are not)
        // the end-of-epoch rewind of a training loop finds batch 0 staged ahead by the last fetch of the epoch: nothing to wait for, the ring goes on
        const bool keep = !ds_name && !norm_dirty && done && ring_numel == numel && dev_bid[seq % RING] == 0;
        if (!keep) {
            if (ds_name) cp->idle(); else cp->settle();
            for (int i = 0; i < RING; i++) { if (staged[i] && dev_bid[i] >= 0) t4k_event_sync(staged[i]); dev_bid[i] = -1; }
            mark_bid = -1;
        }
        norm_dirty = false;
        batch_id = done = 0;
    }
    const int b = batch_id, nb = cp->n_batches();
    if (done || b >= nb) { hprintf("%s::fetch EOF reached (needs rewind)\n", cp->cifar ? "Cifar10" : "Mnist"); done = 1; hprintf("  } dataset#fetch => corpus fetch failed\n"); return -3; }
    if (ring_numel != numel) {
        if (ring_numel) { t4k_sync(stream()); release_ring(); }
        for (int i = 0; i < RING; i++) {
            void *p; t4k_malloc(&p, sizeof(float) * numel); dbuf[i] = (float *)p; t4k_memset(p, 0, sizeof(float) * numel, stream());
            t4k_malloc(&p, sizeof(uint32_t) * N()); lbuf[i] = (uint32_t *)p;
            if (!staged[i]) t4k_event_create(&staged[i]);
        }
        if (!mark) t4k_event_create(&mark);
        ring_numel = numel;
    }
    const long cell = HWC();
    t4k_stream_t side = g_prefetch ? feed_stream() : nullptr;     // none on a backend without streams (the oracle's): every batch then takes the in-stream path
    // ---- batch b becomes current
    const int r = seq % RING;
    if (dev_bid[r] == b) t4k_event_sync(staged[r]);      // staged ahead (issued a step ago)
    else {
        const int n = cp->wait_batch(b);
        if (n <= 0) { hprintf("  } dataset#fetch => corpus fetch failed\n"); return -3; }
        chk(t4k_stage_batch(cp->pix[b & 1], dbuf[r], (long)n * cell, mean, scale, cp->lab[b & 1], lbuf[r], n, stream()), "dataset#load");   // (x - mean) * scale and the labels, one launch
        t4k_event_record(cp->pin_done[b & 1], stream()); cp->pin_ev[b & 1] = cp->pin_done[b & 1]; cp->pin_wait[b & 1] = true;
        // a short last batch leaves the previous batch's samples behind it (Dataset::_load copies batch_sz samples into ONE buffer, dataset.cu:142-158)
        if (n < cp->N && b > 0) t4k_memcpy_d2d(dbuf[r] + (long)n * cell, dbuf[(seq + RING - 1) % RING] + (long)n * cell, sizeof(float) * (size_t)(cp->N - n) * cell, stream());
        dev_bid[r] = b; dev_n[r] = n;
    }
    data = dbuf[r]; label = lbuf[r]; batch_sz = dev_n[r];
    done = ((long)b * cp->N + batch_sz >= cp->corpus_sz) ? 1 : 0;
    const bool first_batch_of_corpus = !cp->names_shown_latch;
    if (tr) {                                            // mnist.cpp:88-90 / cifar10.cpp:71-73,122-132, dataset.cu:101-106
        if (cp->cifar && first_batch_of_corpus) {             // the first batch a CIFAR corpus ever reads: its class names, 16 to a line (cifar10.cpp dump_1st; `first` = the label block did not exist yet)
            static const char *nm[] = { "plane", "car", "bird", "cat", "deer", "dog", "frog ", "horse", "ship ", "truck", "ERROR" };
            std::vector<uint32_t> lab((size_t)batch_sz);
            if (batch_sz) { t4k_memcpy_d2h(lab.data(), label, sizeof(uint32_t) * (size_t)batch_sz, stream()); t4k_sync(stream()); }
            std::string o; char b16[16];
            for (int i2 = 0; i2 < batch_sz; i2++) { snprintf(b16, sizeof(b16), "%-5s%c", nm[lab[i2] < 10 ? lab[i2] : 10], ((i2 + 1) % 16) ? ' ' : '\n'); o += b16; }
            hputs(o);
        }
        hprintf("\t%s batch[%d] loaded=%ld/%d done=%d\n", cp->cifar ? "CIFAR-10" : "Mnist", b, (long)b * cp->N + batch_sz, cp->corpus_sz, done);
        hprintf("  } dataset#fetch => batch[%d] ", b);
        if (done) hprintf("completed, no more data.\n"); else hprintf("%d record(s) loaded\n", batch_sz);
    }
    cp->names_shown_latch = true;
    // ---- the batch of the NEXT fetch goes to its buffer on the side stream if the reader already holds it (full batches only); a cold start just asks for
    // it.  Behind the last batch of an epoch that is batch 0 again (the training loops rewind next): with an even number of batches it sits in the pinned slot
    // batch nb - 2 left (asked for below, one fetch earlier), so the rewind finds it staged like any other batch and waits for nothing.
    const bool wrap = nb >= 4 && !(nb & 1);
    if (!done || wrap) {
        const int b1 = done ? 0 : b + 1, r1 = (seq + 1) % RING;
        if (side && dev_bid[r1] != b1 && cp->slot_bid[b1 & 1] == b1 && (long)(b1 + 1) * cp->N <= cp->corpus_sz) {
            const int n1 = cp->wait_batch(b1);
            if (n1 == cp->N) {
                if (mark_bid < 0 || seq - mark_bid >= MARK_EVERY) {        // a fresh mark: behind everything issued so far; the side stream waits for it once
                    t4k_event_record(mark, stream()); mark_bid = seq;
                    t4k_stream_wait_event(side, mark);
                }
                chk(t4k_stage_batch(cp->pix[b1 & 1], dbuf[r1], (long)n1 * cell, mean, scale, cp->lab[b1 & 1], lbuf[r1], n1, side), "dataset#prefetch");
                t4k_event_record(staged[r1], side);
                cp->pin_ev[b1 & 1] = staged[r1]; cp->pin_wait[b1 & 1] = true;
                dev_bid[r1] = b1; dev_n[r1] = n1;
            }
        }
    }
    if (!done) {
        cp->request(b + 1);                              // no-op when the slot holds (or is being filled with) that batch
        cp->request(b + 2 == nb && wrap ? 0 : b + 2);    // slot b & 1: batch b's staging launch has been issued, the reader waits for its event
    } else {                                             // last batch of the epoch: the training loops rewind next - the reader wraps around ahead of them (it waits for
        cp->request(0); cp->request(1);                  // the staging launches of the batches the slots held), so the rewind costs no synchronous read
    }
    seq++;
    batch_id++;
    return 0;
}

} // namespace t4
