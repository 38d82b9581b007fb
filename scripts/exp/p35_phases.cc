// Where the time of one P3.5Pf sample goes (k_focal_generate's form: one lane = one sample, 16 samples per workgroup, matrices in LDS):
// cycle counter at the phase boundaries of p35pf.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I poselib_amd/csrc scripts/exp/p35_phases.cc -o scripts/exp/p35_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
__device__ unsigned long long *g_marks; // [6][samples]
#define PL_P35_MARK(i) marks_[i] = __builtin_readcyclecounter()
#define PL_EIG_MARK() marks_[6] = __builtin_readcyclecounter()
static __device__ __host__ unsigned long long marks_dummy[7];
#include "pl_device.h"
namespace pl { static thread_local unsigned long long *marks_host; }
#define marks_ marks_of()
__device__ __host__ inline unsigned long long *marks_of();
#include "pl_solver_p35pf.h"
__device__ unsigned long long d_marks[7 * 4096];
__device__ __host__ inline unsigned long long *marks_of() {
#if defined(__HIP_DEVICE_COMPILE__)
    return d_marks + 7 * (blockIdx.x * 16 + threadIdx.x);
#else
    return marks_dummy;
#endif
}
using namespace pl;
__global__ __launch_bounds__(64) void k(const double *in, uint32_t count, int *nsol) {
    extern __shared__ double s_work[];
    const uint32_t it = blockIdx.x * 16 + threadIdx.x;
    if (it >= count) return;
    const double *p = in + (size_t)it * 20;
    double xs[8]; Vec3 X[4];
    for (int k = 0; k < 8; ++k) xs[k] = p[k];
    for (int k = 0; k < 4; ++k) X[k] = v3(p[8 + 3 * k], p[9 + 3 * k], p[10 + 3 * k]);
    P35Solution sol[10];
    nsol[it] = p35pf(xs, X, P35Work{s_work + threadIdx.x, (size_t)16}, sol);
}
int main() {
    const uint32_t n = 1008;
    std::vector<double> in(n * 20);
    srand(5);
    auto rnd = [] { return rand() / (double)RAND_MAX; };
    for (uint32_t i = 0; i < n; ++i) { // a true pose: points in front of a camera with focal 1000
        double *p = &in[i * 20];
        for (int k = 0; k < 4; ++k) {
            const double X = 4 * rnd() - 2, Y = 4 * rnd() - 2, Z = 4 + 4 * rnd();
            p[8 + 3 * k] = X + 0.3, p[9 + 3 * k] = Y - 0.2, p[10 + 3 * k] = Z - 5.0; // (world = camera - t, R = I)
            p[2 * k] = 1000.0 * X / Z, p[2 * k + 1] = 1000.0 * Y / Z;
        }
    }
    double *d_in; int *d_n;
    hipMalloc(&d_in, in.size() * 8); hipMalloc(&d_n, n * 4);
    hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice);
    const size_t bytes = sizeof(double) * kP35WorkDoubles * 16;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<<<dim3(n / 16), dim3(16), bytes>>>(d_in, n, d_n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.3f ms for %u samples\n", rep, ms, n);
    }
    std::vector<unsigned long long> m(7 * n);
    hipMemcpyFromSymbol(m.data(), HIP_SYMBOL(d_marks), m.size() * 8);
    std::vector<int> ns(n); hipMemcpy(ns.data(), d_n, n * 4, hipMemcpyDeviceToHost);
    double ph[5] = {0, 0, 0, 0, 0}; double sols = 0, hess = 0;
    for (uint32_t i = 0; i < n; ++i) { for (int q = 0; q < 5; ++q) ph[q] += (double)(m[7 * i + q + 1] - m[7 * i + q]); hess += (double)(m[7 * i + 6] - m[7 * i + 3]); sols += ns[i]; }
    const char *name[5] = {"null space 12x7", "equations -> rows", "elimination 25 pivots", "action matrix + eigenvalues", "null vectors + poses"};
    double tot = 0; for (double v : ph) tot += v;
    for (int q = 0; q < 5; ++q) printf("%-28s %10.0f cycles (%.1f %%)\n", name[q], ph[q] / n, 100 * ph[q] / tot);
    printf("of the eigenvalue stage: action matrix + Hessenberg reduction %.0f cycles\n", hess / n);
    printf("mean solutions %.2f; total %.0f cycles (counter at 100 MHz: x clock/100MHz)\n", sols / n, tot / n);
}
