// poselib_amd — the host / device function qualifier shared by the math headers.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PL_HD __host__ __device__ __forceinline__
#define PL_UNROLL _Pragma("unroll") // small constant-trip loops over register arrays: no dynamic indexing -> no scratch
#else
#define PL_HD inline
#define PL_UNROLL
#endif
