// smc_platform.h -- the one place that names the HIP runtime.
//
// Product builds (hipcc --offload-arch=gfx950) include <hip/hip_runtime.h>.
// The CPU test-suite additionally compiles the same sources with g++
// -DSMC_EMULATE against tests/emu/hip_emu.h (a fiber emulator of the handful
// of HIP constructs used here) so kernel logic can be checked without a GPU;
// that build is test infrastructure and is never loaded by the product.
#pragma once

#ifdef SMC_EMULATE
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define SMC_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), 0, (stream), __VA_ARGS__)
#endif

#include <cstdint>

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;
