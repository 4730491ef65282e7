// Debug build of the library: per-iteration IPM trace of block 0 (development only; see tools/pair_probe.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#define EMP_QP_DEBUG(...)                                              \
    do {                                                               \
        if (gl == 0 && blockIdx.x == 0) {                              \
            printf("[g%d] ", (int)((threadIdx.x & 63) >> 5));          \
            printf(__VA_ARGS__);                                       \
        }                                                              \
    } while (0)
#include "../emplanner_carla_amd/csrc/emp_api.hip"
