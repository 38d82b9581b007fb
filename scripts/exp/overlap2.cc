// Experiment (round 6): the tile loop of k_score_mfma in isolation - per iteration three independent v_mfma_f32_32x32x16_f16
// (C = 0) and 32 vector instructions that CONSUME their results (16 v_or3 + 16 v_alignbit) - at 2, 4, 6, 8 wavefronts per SIMD:
// does the matrix pipe's time (96 cycles per iteration and wavefront) run under the vector instructions (128 cycles) or next to them?
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form overlap2.cc -o overlap2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// MODE 0: products, then their consumers (the kernel before round 6's pipelining); 1: the products of iteration i + 1 between the
// v_alignbit of iteration i (the kernel's form); 2: vector instructions only (the same 32, on stale values); 3: products only
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(int iters, unsigned *out, const half8_t *ab) {
    const int lane = threadIdx.x & 63;
    half8_t A0 = ab[lane], A1 = ab[64 + lane], A2 = ab[128 + lane], B = ab[192 + lane];
    const float16_t Z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned o[16];
    for (int i = 0; i < 16; ++i) o[i] = threadIdx.x * 2654435761u + i;
    float16_t D0 = Z, D1 = Z, D2 = Z;
    if (MODE != 2) {
        D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B, Z, 0, 0, 0);
        D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, B, Z, 0, 0, 0);
        D2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, B, Z, 0, 0, 0);
    }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        half8_t Bn = B;
        Bn[0] = (_Float16)(float)(it & 3); // (a fresh operand per iteration: the products cannot be hoisted)
        if (MODE == 0) {
            D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, Bn, Z, 0, 0, 0);
            D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, Bn, Z, 0, 0, 0);
            D2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, Bn, Z, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 16; ++v)
                o[v] = __builtin_amdgcn_alignbit(o[v], __float_as_uint(D0[v]) | __float_as_uint(D1[v]) | __float_as_uint(D2[v]), 31);
        } else if (MODE == 1) {
            unsigned T[16];
#pragma unroll
            for (int v = 0; v < 16; ++v)
                T[v] = __float_as_uint(D0[v]) | __float_as_uint(D1[v]) | __float_as_uint(D2[v]);
            D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, Bn, Z, 0, 0, 0);
            D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, Bn, Z, 0, 0, 0);
            D2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, Bn, Z, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 16; ++v)
                o[v] = __builtin_amdgcn_alignbit(o[v], T[v], 31);
            __builtin_amdgcn_sched_group_barrier(0x002, 17, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        } else if (MODE == 2) {
#pragma unroll
            for (int v = 0; v < 16; ++v)
                o[v] = __builtin_amdgcn_alignbit(o[v], o[(v + 1) & 15] | o[(v + 5) & 15] | (unsigned)it, 31);
        } else {
            D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, Bn, Z, 0, 0, 0);
            D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, Bn, Z, 0, 0, 0);
            D2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, Bn, Z, 0, 0, 0);
            o[it & 15] ^= __float_as_uint(D0[0]) ^ __float_as_uint(D1[1]) ^ __float_as_uint(D2[2]);
        }
    }
    unsigned acc = 0;
    for (int i = 0; i < 16; ++i) acc ^= o[i];
    for (int i = 0; i < 16; ++i) acc ^= __float_as_uint(D0[i]) ^ __float_as_uint(D1[i]) ^ __float_as_uint(D2[i]);
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = acc;
}
template <int MODE, int WAVES> float run(int iters, unsigned *out, const half8_t *ab) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256; // one workgroup per CU: WAVES wavefronts on each of its 4 SIMDs... (WAVES * 4 wavefronts per workgroup)
    k<MODE, WAVES * 4><<<blocks, 64 * WAVES * 4>>>(iters, out, ab);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) k<MODE, WAVES * 4><<<blocks, 64 * WAVES * 4>>>(iters, out, ab);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
template <int WAVES> void row(unsigned *out, const half8_t *ab) {
    const int iters = 20000;
    const float a = run<0, WAVES>(iters, out, ab), b = run<1, WAVES>(iters, out, ab), v = run<2, WAVES>(iters, out, ab), m = run<3, WAVES>(iters, out, ab);
    const double c = 1e-3 * 2.4e9 / (iters * (double)WAVES); // ms -> cycles per iteration and wavefront of a SIMD
    printf("| %d | %.0f | %.0f | %.0f | %.0f | %.0f | %.0f |\n", WAVES, v * c, m * c, a * c, b * c, (v + m) * c, (v > m ? v : m) * c);
}
int main() {
    unsigned *out; CK(hipMalloc(&out, 4 * 1024 * 256)); half8_t *ab; CK(hipMalloc(&ab, 16 * 256)); CK(hipMemset(ab, 0, 16 * 256));
    printf("cycles per iteration and wavefront at 2.4 GHz (one iteration = 3 products + 16 v_or3 + 16 v_alignbit)\n");
    printf("| wavefronts per SIMD | vector only | products only | products, then consumers | pipelined by hand | sum | max |\n|---|---|---|---|---|---|---|\n");
    row<1>(out, ab); row<2>(out, ab); row<3>(out, ab); row<4>(out, ab);
    return 0;
}
