// capi.cpp - C embedding API of the host VM (include/ten4.h).
#include "vm.h"
#include "../../include/ten4.h"
#include <sstream>

struct ten4_vm { t4::VM vm; std::string out; };

extern "C" {

ten4_vm *ten4_new(int device, unsigned long long seed, int trace_level) {
    if (device >= 0) setenv("T4_DEVICE", std::to_string(device).c_str(), 1);
    setenv("T4_SEED", std::to_string(seed).c_str(), 1);
    if (t4k_device_count() <= 0) return nullptr;        // no CPU fallback
    ten4_vm *h = new ten4_vm();
    h->vm.trace_lvl = trace_level;
    h->vm.init();
    t4::die_if_no_backend();                            // the backend seeds itself once per process from T4_SEED ...
    t4k_rand_init(seed);                                // ... every embedded VM starts its own stream at `seed` (a second VM in one process used to inherit the first one's position)
    return h;
}
void ten4_free(ten4_vm *h) { delete h; }
int ten4_eval(ten4_vm *h, const char *source) {
    std::istringstream in(source ? source : "");
    std::string line;
    while (!h->vm.done() && std::getline(in, line)) h->vm.eval(line);
    return h->vm.done() ? 0 : 1;
}
const char *ten4_output(ten4_vm *h) { h->out = h->vm.take_output(); return h->out.c_str(); }
int ten4_grad_slab(ten4_vm *, float **p, long *n) {
    t4::Model *m = t4::Model::current;
    if (!m || !m->gslab) return -1;
    *p = m->gslab->data; *n = (long)m->gslab->numel;
    return 0;
}
void *ten4_stream(ten4_vm *) { return (void *)t4k_default_stream(); }
void ten4_set_grad_hook(ten4_vm *, ten4_grad_hook_fn fn, void *user) { t4::Model::grad_hook = fn; t4::Model::grad_hook_user = user; }

} // extern "C"
