// Experiment: what does ONE step of a sequential fp64 sum cost a gfx950 wavefront?  (k_lm's consumer: lane a adds row after row of an
// LDS ring, the additions are a dependent chain.)  Variants: registers only (pure v_add_f64 latency), LDS rows with the reads of the
// next eight rows issued before the current eight are added, the same with other wavefronts of the workgroup busy on the SIMDs.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off chain_add.cc -o chain_add && ./chain_add
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int RS = 29, ROWS = 64;
#include "../../poselib_amd/csrc/pl_lm_chain.inc"

__global__ void k_reg(int n, double *out, unsigned long long *cyc) {
    double t = threadIdx.x * 1e-3, a = 1.0000001;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(t) : "v"(a));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = t;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_reg_fma(int n, double *out, unsigned long long *cyc) {
    double t = threadIdx.x * 1e-3, a = 1.0000001, b = 0.5;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(t) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = t;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_reg_f32(int n, double *out, unsigned long long *cyc) {
    float t = threadIdx.x * 1e-3f, a = 1.0000001f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(t) : "v"(a));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = t;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// busy != 0: wavefronts 1.. of the workgroup run independent fp64 work on all SIMDs while wavefront 0 runs the chain
template <int DEPTH> __global__ void k_lds(int slots, int busy, double *out, unsigned long long *cyc) {
    __shared__ double ring[ROWS][RS];
    for (int i = threadIdx.x; i < ROWS * RS; i += blockDim.x) (&ring[0][0])[i] = 1e-3 * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        const int col = lane < 28 ? lane : 0;
        double tot = 0.0;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int s = 0; s < slots; ++s) {
            double ba[DEPTH], bb[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) ba[u] = ring[u][col];
#pragma unroll 1
            for (int q = 0; q < ROWS; q += 2 * DEPTH) {
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) bb[u] = ring[q + DEPTH + u][col];
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) tot += ba[u];
                if (q + 2 * DEPTH < ROWS) {
#pragma unroll
                    for (int u = 0; u < DEPTH; ++u) ba[u] = ring[q + 2 * DEPTH + u][col];
                }
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) tot += bb[u];
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        out[lane] = tot;
        if (lane == 0) cyc[0] = t1 - t0;
    } else if (busy) {
        double r[8];
        for (int i = 0; i < 8; ++i) r[i] = lane * 1e-3 + i;
        for (int it = 0; it < busy; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = r[i] * 1.0000001 + 0.5;
        double acc = 0; for (int i = 0; i < 8; ++i) acc += r[i];
        out[64 + threadIdx.x] = acc;
    }
}
constexpr int CS = ROWS + 2; // column stride in doubles (column-major ring: a lane's rows are adjacent)
__global__ void k_lds_asm(int slots, int busy, double *out, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) double ring[RS][CS];
    for (int i = threadIdx.x; i < ROWS * RS; i += blockDim.x) ring[i % RS][i / RS] = 1e-3 * i; // the same numbers as k_lds: row i / RS, column i % RS
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        const int col = lane < 28 ? lane : 0;
        double tot = 0.0;
        const uint32_t a = (uint32_t)(uintptr_t)&ring[col][0];
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int s = 0; s < slots; ++s)
            PL_LM_CHAIN64(tot, a);
        const unsigned long long t1 = __builtin_readcyclecounter();
        out[lane] = tot;
        if (lane == 0) cyc[0] = t1 - t0;
    } else if (busy) {
        double r[8];
        for (int i = 0; i < 8; ++i) r[i] = lane * 1e-3 + i;
        for (int it = 0; it < busy; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = r[i] * 1.0000001 + 0.5;
            if ((it & 7) == 0) ring[lane % RS][(it >> 3) & 63] += 0.0; // some LDS traffic of the other wavefronts
        }
        double acc = 0; for (int i = 0; i < 8; ++i) acc += r[i];
        out[64 + threadIdx.x] = acc;
    }
}
int main() {
    double *out; unsigned long long *cyc, h;
    CK(hipMalloc(&out, 8 * 2048)); CK(hipMalloc(&cyc, 8));
    const int n = 20000;
    k_reg<<<1, 64>>>(n, out, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("dependent v_add_f64 chain, one wavefront alone      : %.2f cycles per add\n", (double)h / (16.0 * n));
    k_reg_fma<<<1, 64>>>(n, out, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("dependent v_fma_f64 chain                           : %.2f cycles per fma\n", (double)h / (16.0 * n));
    k_reg_f32<<<1, 64>>>(n, out, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("dependent v_add_f32 chain                           : %.2f cycles per add\n", (double)h / (16.0 * n));
    const int slots = 4000;
    for (int busy : {0, 400000}) {
        for (int threads : {64, 512}) {
            if (!busy && threads == 512) continue;
            k_lds<4><<<1, threads>>>(slots, busy, out, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("LDS rows, 4 reads ahead, %d threads, others %s : %.2f cycles per row\n", threads, busy ? "busy" : "idle", (double)h / (64.0 * slots));
            k_lds<8><<<1, threads>>>(slots, busy, out, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("LDS rows, 8 reads ahead, %d threads, others %s : %.2f cycles per row\n", threads, busy ? "busy" : "idle", (double)h / (64.0 * slots));
            k_lds<16><<<1, threads>>>(slots, busy, out, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("LDS rows, 16 reads ahead, %d threads, others %s: %.2f cycles per row\n", threads, busy ? "busy" : "idle", (double)h / (64.0 * slots));
        }
    }
    for (int busy : {0, 400000}) for (int threads : {64, 512}) {
        if (!busy && threads == 512) continue;
        k_lds_asm<<<1, threads>>>(slots, busy, out, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        double ho[64]; CK(hipMemcpy(ho, out, 8 * 64, hipMemcpyDeviceToHost));
        printf("LDS rows, asm: b128 reads, 6 pairs ahead, %d threads, others %s : %.2f cycles per row (sum lane 3 = %.6f)\n", threads, busy ? "busy" : "idle", (double)h / (64.0 * slots), ho[3]);
    }
    k_lds<8><<<1, 64>>>(slots, 0, out, cyc); { double ho[64]; CK(hipMemcpy(ho, out, 8 * 64, hipMemcpyDeviceToHost)); printf("reference sum lane 3 = %.6f\n", ho[3]); }
    return 0;
}
