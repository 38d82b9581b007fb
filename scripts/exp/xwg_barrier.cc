// Experiment (round 5): what does ONE exchange step between the W workgroups of a refinement task cost?
// A split LM iteration would end its sweep with: every workgroup publishes its partial normal equations (NV doubles) to global memory,
// release fence, atomic increment of the task's arrival counter, spin until all W have arrived, acquire, read the W partials.
// Here: T tasks x W workgroups (the parts of a task on ONE XCD: linear workgroup ids that differ by multiples of 8), R rounds, no other work.
//   hipcc --offload-arch=gfx950 -O3 xwg_barrier.cc -o xwg_barrier && ./xwg_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int NV = 48;

__global__ __launch_bounds__(512) void k_xwg(int W, int rounds, int same_xcd, double *xbuf, unsigned *xcnt, double *out, unsigned *fail) {
    // block -> (task, part)
    const unsigned b = blockIdx.x;
    unsigned task, part;
    if (same_xcd) {
        const unsigned col = b & 7u, row = b >> 3;
        task = (row / W) * 8u + col;
        part = row % W;
    } else {
        task = b / W;
        part = b % W;
    }
    double acc = 1.0 + part;
    double *mine = xbuf + (size_t)task * 2 * W * NV;
    for (int r = 1; r <= rounds; ++r) {
        double *slot = mine + (size_t)(r & 1) * W * NV;
        if (threadIdx.x < NV)
            __hip_atomic_store(&slot[part * NV + threadIdx.x], acc + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_fetch_add(&xcnt[task], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(&xcnt[task], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(W * r)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) { // watchdog: never hang the device
                    atomicAdd(fail, 1u);
                    break;
                }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        if (threadIdx.x < NV) {
            double s = 0;
            for (int p = 0; p < W; ++p)
                s += __hip_atomic_load(&slot[p * NV + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc = s * 0.25;
        }
        acc = __shfl(acc, 0, 64);
        __syncthreads();
    }
    if (threadIdx.x == 0 && part == 0)
        out[task] = acc;
}

int main() {
    double *xbuf, *out;
    unsigned *xcnt, *fail;
    const int Tmax = 512, Wmax = 4;
    CK(hipMalloc(&xbuf, sizeof(double) * Tmax * 2 * Wmax * NV));
    CK(hipMalloc(&out, sizeof(double) * Tmax));
    CK(hipMalloc(&xcnt, sizeof(unsigned) * Tmax));
    CK(hipMalloc(&fail, sizeof(unsigned)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("| tasks | W | parts of a task on one XCD | us per exchange round | watchdog trips |\n|---|---|---|---|---|\n");
    for (int same : {1, 0})
        for (int W : {2, 4})
            for (int T : {8, 64, 128}) {
                const int rounds = 2000;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipMemset(xcnt, 0, sizeof(unsigned) * Tmax));
                    CK(hipMemset(fail, 0, sizeof(unsigned)));
                    CK(hipEventRecord(e0, 0));
                    k_xwg<<<T * W, 512>>>(W, rounds, same, xbuf, xcnt, out, fail);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    unsigned f;
                    CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
                    if (rep == 1)
                        printf("| %d | %d | %s | %.2f | %u |\n", T, W, same ? "yes" : "no", ms * 1e3 / rounds, f);
                }
            }
    return 0;
}
