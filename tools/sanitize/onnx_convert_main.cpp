#include <cstdio>
#include "sdhip.h"
int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i)
        for (int kind = 0; kind < 2; ++kind) {
            int rc = sd_convert_onnx(argv[i], kind, "asan_out.sdw");
            printf("%s kind %d -> %d %s\n", argv[i], kind, rc, rc ? sd_convert_error() : "");
        }
    return 0;
}
