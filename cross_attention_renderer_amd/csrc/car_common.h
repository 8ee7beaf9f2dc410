// car_common.h — error plumbing shared by the translation units of libcar_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/car_hip.h"

void car_set_error(const char* fmt, ...);

#define CAR_REQUIRE(cond, ...)                         \
    do {                                               \
        if (!(cond)) {                                 \
            car_set_error(__VA_ARGS__);                \
            return CAR_E_ARG;                          \
        }                                              \
    } while (0)

// Checks the launch itself (configuration / missing code object); never synchronises.
#define CAR_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            car_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return CAR_E_LAUNCH;                                                      \
        }                                                                             \
    } while (0)

// -DCAR_BOUNDS (tools/build_bounds.py; tests/oob_runner.py with CAR_OOB_LIB): a debug build in which every LDS-DMA / buffer-load / row-load
// helper of the kernels compares the range it is about to read with the extent its launcher passed in the args struct and TRAPS when it
// leaves it — the one way to see a read whose value nobody uses (the NaN-margin harness only sees reads that reach a result).  The
// product build compiles these to nothing.
#ifdef CAR_BOUNDS
#define CAR_BOUNDS_TRAP(cond) do { if (!(cond)) __builtin_trap(); } while (0)
#else
#define CAR_BOUNDS_TRAP(cond) do { } while (0)
#endif

static inline unsigned car_div_up(long a, long b) { return (unsigned)((a + b - 1) / b); }
