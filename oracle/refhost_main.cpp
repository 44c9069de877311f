// refhost_main.cpp - TEST INFRASTRUCTURE (build container only).  The outer loop of the reference's own binary, restated so that the
// reference's REAL host half (its 17 g++-compiled sources: VM, printer, layer factory, loader, saver, TensorBoard writer) can be RUN on the
// CPU: those objects + integration/t4k_bind*.cpp (the reference-side binding of include/t4k.h) + oracle/t4k_on_oracle.cpp (the same C-ABI
// on host memory).  Nothing of the reference is copied or stubbed: its sources are compiled where they lie (oracle/Makefile, target
// _ref/ten4_refhost); this file only restates what lives in the one .cu file of the host side that cannot be compiled here:
//   src/ten4.cu:40-52   _vm_init       (factory + dictionary init + first QUERY)
//   src/ten4.cu:67-78   _ten4_tally    (sweep, count VM states)
//   src/ten4.cu:80-91   _vm_exec0      (resume a HOLDing VM, else outer())
//   src/ten4.cu:224-235 main_loop      (more_job && readline -> run -> flush)
//   src/ten4.cu:237-251 teardown
// The seed is the one deviation: the reference seeds from time() (src/sys.cpp:37); T4_SEED pins it so outputs are reproducible.
#include <cstdlib>
#include <iostream>
#include "t4k.h"
#include "debug.h"
#include "vm/vm.h"
#include "sys.h"
namespace t4::vm { VM *vm_factory(vm::vm_level level, int id, System &sys); }   // src/ten4.h:11-13 (ten4.h itself names cudaStream_t / cudaEvent_t)

using namespace t4;

int main(int argc, char **argv) {
    const char *tb_logdir = nullptr, *tb_run_id = nullptr;
    int verbose = T4_VERBOSE;
    for (int i = 1; i < argc; i++) {                                     // -t<logdir> -r<run_id> -v<level> (src/opt.h:44-58)
        if (!strncmp(argv[i], "-t", 2) && argv[i][2]) tb_logdir = argv[i] + 2;
        if (!strncmp(argv[i], "-r", 2) && argv[i][2]) tb_run_id = argv[i] + 2;
        if (!strncmp(argv[i], "-v", 2)) verbose = atoi(argv[i][2] ? argv[i] + 2 : (i + 1 < argc ? argv[++i] : "1"));
    }
    std::cout << T4_APP_NAME << std::endl;                               // ten4.cu:296
    System *sys = System::get_sys(std::cin, std::cout, 0, verbose);     // ten4.cu:155
    if (const char *s = getenv("T4_SEED")) { t4k_rand_init(strtoull(s, nullptr, 10)); t4k_rand_set_offset(0); }
    vm::VM *pool[T4_VM_COUNT];
    for (int id = 0; id < T4_VM_COUNT; id++) {                           // _vm_init
        pool[id] = vm::vm_factory(vm::NET, id, *sys);
        pool[id]->init();
    }
    sys->mu->dict_validate();
    sys->mu->status(true);
    pool[0]->state = vm::QUERY;
    if (tb_logdir && tb_run_id) {                                        // setup(), ten4.cu:170-177
        std::cout << "\\ TensorBoard logdir=" << tb_logdir << ", run_id=" << tb_run_id << std::endl;
        sys->setup_tb(tb_logdir, tb_run_id);
    }
    sys->db->self_tests();
    int cnt[vm::VM_STATE_MAX];
    for (;;) {                                                           // main_loop
        sys->mu->sweep();                                                // _ten4_tally
        for (int i = 0; i < vm::VM_STATE_MAX; i++) cnt[i] = 0;
        for (int id = 0; id < T4_VM_COUNT; id++) cnt[pool[id]->state]++;
        if (!(cnt[vm::STOP] < T4_VM_COUNT && sys->readline(cnt[vm::HOLD]))) break;
        for (int id = 0; id < T4_VM_COUNT; id++) {                       // run() / _vm_exec0
            vm::VM *v = pool[id];
            if (v->state == vm::STOP) continue;
            if (v->state == vm::HOLD) v->resume(); else v->outer();
        }
        sys->flush();
    }
    std::cout << "\\ VM[] ";                                             // teardown
    for (int id = 0; id < T4_VM_COUNT; id++) delete pool[id];
    std::cout << "freed" << std::endl;
    System::free_sys();
    std::cout << T4_APP_NAME << " done." << std::endl;
    return 0;
}
