// main.cpp -- `StereoMatch script <file>` / `StereoMatch name value ...` (reference main.cpp:29-52).
#include "Error.h"
#include "StereoIO.h"
#include <cstdio>

int main(int argc, const char *argv[])
{
    try {
        CStereoIO s;
        s.InterpretCommandLine(argc, argv);
    } catch (CError &err) {
        fprintf(stderr, "%s\n", err.message);
        return -1;
    }
    return 0;
}
