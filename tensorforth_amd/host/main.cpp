// main.cpp - `ten4`: the tensorForth REPL on the MI355X backend.  Reads Forth source from stdin
// one line at a time (reference src/ten4.cu:224-235, System::readline src/sys.cpp:101-108).
#include "vm.h"
#include <iostream>

int main(int argc, char **argv) {
    t4::VM vm;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-h")) { printf("usage: ten4 [-v level] < script.4th   (env: T4_DEVICE, T4_SEED)\n"); return 0; }
        if (!strcmp(argv[i], "-v") && i + 1 < argc) vm.trace_lvl = atoi(argv[++i]);
    }
    vm.init();
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
    printf("\ntensorForth done.\n");
    return 0;
}
