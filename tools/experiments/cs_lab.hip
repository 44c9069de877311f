// cs_lab.hip - lab harness for the conv-stack kernels (csrc/conv_stack.hip): event timing of the LeNet front end through the C-ABI.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/experiments/cs_lab.hip -L tensorforth_amd -lt4hip -Wl,-rpath,'$ORIGIN/../../tensorforth_amd' -o tools/experiments/cs_lab.bin
//   (link against -lt4hip_lab, the LAB build `make -C tensorforth_amd/csrc LAB=1`, for the s_memtime stamps: CS_LAB_PROF=1)
//   usage: cs_lab.bin [N] [fwd|bwd|head]      (T4K_STACK_SPLIT=1|2|4 forces the number of bands per image; head = forward with the classifier head)
#include <hip/hip_runtime.h>
#include "t4k.h"
#include <cstdio>
#include <cstring>
#include <vector>
#include <cstdlib>
static float *dalloc(size_t n, bool rnd = false) {
    float *d; hipMalloc((void **)&d, n * 4);
    std::vector<float> h(n, 0.f); if (rnd) for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 128;
    if (t4k_init(0) != 0) { printf("init failed\n"); return 1; }
    unsigned long long *prof; hipMalloc((void **)&prof, (size_t)N * 4 * 32 * 8); hipMemset(prof, 0, (size_t)N * 4 * 32 * 8);
    { static char e[64]; snprintf(e, sizeof(e), "T4K_STACK_PROF_PTR=%llx", (unsigned long long)prof); if (getenv("CS_LAB_PROF")) putenv(e); }
    t4k_conv_stage st[2]; memset(st, 0, sizeof(st));
    st[0].H = 28; st[0].W = 28; st[0].C1 = 1; st[0].C0 = 10; st[0].K = 3;
    st[1].H = 14; st[1].W = 14; st[1].C1 = 10; st[1].C0 = 20; st[1].K = 3;
    for (int s = 0; s < 2; s++) {
        t4k_conv_stage &t = st[s];
        t.F = dalloc(t.C1 * 9 * t.C0, true); t.B = dalloc(t.C0, true); t.O = dalloc((size_t)N * t.H * t.W * t.C0);
        t.DF = dalloc(t.C1 * 9 * t.C0); t.DB = dalloc(t.C0); t.X = dalloc((size_t)N * t.H * t.W * t.C1, true); t.DXS = dalloc((size_t)N * t.H * t.W * t.C1);
        t4k_poolblock &b = t.run; b.KS = 2; b.pool_layer = T4K_L_MAXPOOL; b.pool_out = dalloc((size_t)N * t.H * t.W * t.C0 / 4);
        b.post_layer = T4K_L_RELU; b.post_mask = dalloc((size_t)N * t.H * t.W * t.C0 / 4); b.post_out = dalloc((size_t)N * t.H * t.W * t.C0 / 4);
        if (s == 1) { b.pre_layer = T4K_L_DROPOUT; b.pre_alpha = 0.5f; b.pre_mask = dalloc((size_t)N * t.H * t.W * t.C0); b.pre_out = dalloc((size_t)N * t.H * t.W * t.C0);
                      b.copy_out = dalloc((size_t)N * t.H * t.W * t.C0 / 4); }
    }
    float *X = dalloc((size_t)N * 784, true), *X0 = dalloc((size_t)N * 784), *DY = dalloc((size_t)N * 980, true);
    const char *mode = argc > 2 ? argv[2] : "fwd";
    const bool bwd = mode[0] == 'b', head = mode[0] == 'h';
    t4k_stack_head hd; memset(&hd, 0, sizeof(hd));
    hd.W1 = dalloc(100 * 980, true); hd.B1 = dalloc(100, true); hd.Y1 = dalloc((size_t)N * 100); hd.mid_layer = T4K_L_DROPOUT; hd.mid_alpha = 0.5f;
    hd.mid_mask = dalloc((size_t)N * 100); hd.mid_out = dalloc((size_t)N * 100); hd.W2 = dalloc(1000, true); hd.B2 = dalloc(10, true);
    hd.Y2 = dalloc((size_t)N * 10); hd.P = dalloc((size_t)N * 10); hd.E1 = 980; hd.E0a = 100; hd.E0b = 10;
    if (head) printf("head ok=%d\n", t4k_conv_stack_head_ok(st, 2, N, &hd));
    printf("N=%d mode=%s ok=%d\n", N, mode, t4k_conv_stack_ok(st, 2, N));
    if (bwd && !getenv("CS_LAB_NOFWD")) t4k_conv_stack_fwd(X, X0, st, 2, N, nullptr);      // the banded backward runs on what a forward saved
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t ls = (hipStream_t)t4k_default_stream();
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0, ls);
        int rc = 0;
        for (int i = 0; i < 200; i++) rc |= bwd ? t4k_conv_stack_bwd(DY, st, 2, N, getenv("CS_LAB_TRAIN") ? atoi(getenv("CS_LAB_TRAIN")) : 1, nullptr) : (head ? t4k_conv_stack_head_fwd(X, X0, st, 2, N, &hd, nullptr) : t4k_conv_stack_fwd(X, X0, st, 2, N, nullptr));
        hipEventRecord(e1, ls); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("%s: %.2f us per call (rc %d %s)\n", mode, ms * 1000 / 200, rc, rc ? t4k_last_error() : "");
    }
    if (getenv("CS_LAB_PROF")) {
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)N * 4 * 32); hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost);
        printf("  stamp: cycles since the workgroup's stamp 0, averaged over workgroups\n");
        for (int k = 1; k < 32; k++) {
            double sum = 0; int cnt = 0;
            double mx = 0;
            for (int b = 0; b < N * 4; b++) if (h[b * 32 + k] && h[b * 32]) { const double d = (double)(h[b * 32 + k] - h[b * 32]); sum += d; cnt++; if (d > mx) mx = d; }
            if (cnt) printf("  %2d: avg %8.0f  max %8.0f  (%d workgroups)\n", k, sum / cnt, mx, cnt);
        }
    }
    return 0;
}
