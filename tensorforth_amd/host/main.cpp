// main.cpp - `ten4`: the tensorForth REPL on the MI355X backend.  Reads Forth source from stdin
// one line at a time (reference src/ten4.cu:224-235, System::readline src/sys.cpp:101-108).
#include "vm.h"
#include <iostream>

int main(int argc, char **argv) {
    t4::VM vm;
    const char *tb_dir = nullptr, *tb_run = nullptr;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-h")) { printf("usage: ten4 [-v level] [-t logdir [-r run_id]] < script.4th   (env: T4_DEVICE, T4_SEED, T4_TB_LOGDIR, T4_TB_RUN)\n"); return 0; }
        if (!strcmp(argv[i], "-v") && i + 1 < argc) vm.trace_lvl = atoi(argv[++i]);
        if (!strncmp(argv[i], "-t", 2)) tb_dir = argv[i][2] ? argv[i] + 2 : (i + 1 < argc ? argv[++i] : "");     // -t<logdir> | -t <logdir> (reference: -tlogdir)
        if (!strncmp(argv[i], "-r", 2)) tb_run = argv[i][2] ? argv[i] + 2 : (i + 1 < argc ? argv[++i] : "");     // -r<run_id>
    }
    vm.init();
    if (tb_dir) t4::tb_configure(tb_dir, tb_run);
    printf("tensorForth v4.0 (MI355X backend: %s)\n", t4k_backend_name());
    std::string line;
    while (!vm.done() && std::getline(std::cin, line)) {
        vm.eval(line);
        std::string out = vm.take_output();
        fwrite(out.data(), 1, out.size(), stdout);
        fflush(stdout);
    }
    std::string out = vm.take_output();
    fwrite(out.data(), 1, out.size(), stdout);
    t4::tb_close();
    printf("\ntensorForth done.\n");
    return 0;
}
