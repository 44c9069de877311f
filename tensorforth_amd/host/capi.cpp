// capi.cpp - C embedding API of the host VM (include/ten4.h).
#include "vm.h"
#include "../../include/ten4.h"
#include <sstream>

// Each embedded VM owns its Philox stream position: the backend's (seed, offset) is process-global, so it is swapped in when a
// VM starts executing and saved when it stops - a second VM neither rewinds nor advances the first one's stream.
struct ten4_vm { t4::VM vm; std::string out; uint64_t seed = 0, off = 0; };
static void rng_enter(ten4_vm *h) {
    if (t4k_rand_seed() != h->seed) { t4k_rand_init(h->seed); t4k_rand_set_offset(h->off); }
    else if (t4k_rand_offset() != h->off) t4k_rand_set_offset(h->off);
}
static void rng_leave(ten4_vm *h) { h->off = t4k_rand_offset(); }

extern "C" {

ten4_vm *ten4_new(int device, unsigned long long seed, int trace_level) {
    if (device >= 0) setenv("T4_DEVICE", std::to_string(device).c_str(), 1);
    setenv("T4_SEED", std::to_string(seed).c_str(), 1);
    if (t4k_device_count() <= 0) return nullptr;        // no CPU fallback
    ten4_vm *h = new ten4_vm();
    h->vm.trace_lvl = trace_level;
    h->vm.init();
    t4::die_if_no_backend();                            // the backend seeds itself once per process from T4_SEED ...
    h->seed = seed; h->off = 0;                         // ... every embedded VM has its own stream (seed, position 0), swapped in by ten4_eval
    return h;
}
void ten4_free(ten4_vm *h) { delete h; }
int ten4_eval(ten4_vm *h, const char *source) {
    std::istringstream in(source ? source : "");
    std::string line;
    rng_enter(h);
    while (!h->vm.done() && std::getline(in, line)) h->vm.eval(line);
    rng_leave(h);
    return h->vm.done() ? 0 : 1;
}
const char *ten4_output(ten4_vm *h) { h->out = h->vm.take_output(); return h->out.c_str(); }
int ten4_grad_slab(ten4_vm *, float **p, long *n) {
    t4::Model *m = t4::Model::current;
    if (!m || !m->gslab) return -1;
    t4::Model::slab_exported();                         // the caller reads the slab behind the library's back (torch.distributed on a zero-copy view):
    t4k_sync(t4::stream());                             // nothing may stay deferred - now (any pending fold runs) or later
    *p = m->gslab->data; *n = (long)m->gslab->numel;
    return 0;
}
void *ten4_stream(ten4_vm *) { return (void *)t4k_default_stream(); }
unsigned long long ten4_rand_tell(ten4_vm *h) { return h->off; }
void ten4_rand_seek(ten4_vm *h, unsigned long long off) { h->off = off; }
void ten4_rand_reseed(ten4_vm *h, unsigned long long seed) { h->seed = seed; h->off = 0; }   // installed by the next ten4_eval (rng_enter)
long ten4_fetch(ten4_vm *h, float *dst, long cap, int shape[4]) {
    t4::Tensor *t = h->vm.tos_tensor();
    if (!t) return -1;
    if (shape) { shape[0] = (int)t->H(); shape[1] = (int)t->W(); shape[2] = (int)t->C(); shape[3] = (int)t->N(); }
    const long n = (long)t->numel;
    if (dst && cap >= n && n > 0) { t4k_memcpy_d2h(dst, t->data, (size_t)n * sizeof(float), t4::stream()); t4k_sync(t4::stream()); }
    return n;
}
long ten4_store(ten4_vm *h, const float *src, long n) {
    t4::Tensor *t = h->vm.tos_tensor();
    if (!t || !src || n != (long)t->numel) return -1;
    t->from_host(src, (uint64_t)n);
    return n;
}
void ten4_set_grad_hook(ten4_vm *, ten4_grad_hook_fn fn, void *user) { t4::Model::grad_hook = fn; t4::Model::grad_hook_user = user; }
int ten4_set_lazy_dx0(int on) { return t4::Model::set_lazy_dx0(on != 0) ? 1 : 0; }

} // extern "C"
