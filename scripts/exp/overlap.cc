// Experiment: do v_mfma_f32_32x32x16_f16 and ordinary vector-ALU instructions of DIFFERENT wavefronts (or of the same one)
// overlap on a gfx950 SIMD?  Kernels: VALU only, MFMA only, both in every wavefront, MFMA in half of the wavefronts and
// VALU in the other half.    hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form overlap.cc -o overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE> // 0 VALU, 1 MFMA, 2 both per wave, 3 split by wave parity
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 8))) void k(int iters, int nvalu, unsigned *out, const half8_t *ab) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8_t A = ab[lane], B = ab[64 + lane];
    float16_t D0 = {0}, D1 = {0};
    unsigned r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + i;
    const bool do_mfma = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1));
    const bool do_valu = MODE == 0 || MODE == 2 || (MODE == 3 && !(wave & 1));
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
            D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, D0, 0, 0, 0);
            D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(B, A, D1, 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int j = 0; j < 3; ++j) // 3 x 8 = 24 independent-ish vector instructions
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    r[i] = __builtin_amdgcn_alignbit(r[i], r[(i + 1) & 7] | (unsigned)it, 31);
        }
    }
    unsigned acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= r[i];
    for (int i = 0; i < 16; ++i) acc ^= __float_as_uint(D0[i]) ^ __float_as_uint(D1[i]);
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int MODE> float run(int iters, unsigned *out, const half8_t *ab) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 3;
    k<MODE><<<blocks, 512>>>(iters, 24, out, ab);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) k<MODE><<<blocks, 512>>>(iters, 24, out, ab);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
    unsigned *out; CK(hipMalloc(&out, 4 * 512 * 256 * 3)); half8_t *ab; CK(hipMalloc(&ab, 16 * 128)); CK(hipMemset(ab, 0, 16 * 128));
    const int iters = 20000;
    const float v = run<0>(iters, out, ab), m = run<1>(iters, out, ab), b = run<2>(iters, out, ab), s = run<3>(iters, out, ab);
    // per SIMD: 6 wavefronts; VALU mode issues 24 instr / iteration / wavefront
    printf("iterations %d per wavefront, 6 wavefronts per SIMD\n", iters);
    printf("VALU only (24 instr/it)            : %.3f ms  -> %.2f cycles per vector instruction per SIMD (2.4 GHz)\n", v, v * 1e-3 * 2.4e9 / (iters * 24.0 * 6));
    printf("MFMA only (2 x 32x32x16 f16 / it)  : %.3f ms  -> %.1f cycles per MFMA per SIMD\n", m, m * 1e-3 * 2.4e9 / (iters * 2.0 * 6));
    printf("both in every wavefront            : %.3f ms  (sum %.3f, max %.3f)\n", b, v + m, v > m ? v : m);
    printf("MFMA in odd waves, VALU in even    : %.3f ms  (half the work of each kind: sum/2 %.3f, max/2 %.3f)\n", s, (v + m) / 2, (v > m ? v : m) / 2);
    return 0;
}
