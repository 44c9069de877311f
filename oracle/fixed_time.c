/* fixed_time.c - TEST INFRASTRUCTURE (build container only): LD_PRELOADed into oracle/_ref/ten4_refhost so that the reference's own
 * TensorBoard writer (src/tb/writer.h:40,173 stamps events with std::time(nullptr)) writes reproducible files: time() returns
 * T4_TB_FIXED_TIME when it is set - the clock the product's sink reads from the same variable (host/tboard.cpp). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdlib.h>
#include <time.h>
time_t time(time_t *out) {
    const char *f = getenv("T4_TB_FIXED_TIME");
    time_t t;
    if (f) t = (time_t)atof(f);
    else { time_t (*real)(time_t *) = (time_t (*)(time_t *))dlsym(RTLD_NEXT, "time"); t = real ? real(NULL) : 0; }
    if (out) *out = t;
    return t;
}
