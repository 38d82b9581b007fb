// Experiment harness (not part of the product): times the staged 5-point generator on a synthetic two-view scene and
// prints checksums of everything it writes, so that a rewritten stage can be compared bit for bit with the old one.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I../../poselib_amd/csrc genbench.cc -L../../poselib_amd/lib -lposelib_amd
#include "pl_kernels.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace pl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static uint64_t sm(uint64_t &s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static double uni(uint64_t &s) { return (sm(s) >> 11) * (1.0 / 9007199254740992.0); }
static uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull) { const unsigned char *c = (const unsigned char *)p; for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; } return h; }
int main(int argc, char **argv) {
    const uint32_t N = 5000, B = argc > 1 ? atoi(argv[1]) : 100000, slots = argc > 2 ? atoi(argv[2]) : 16;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    uint64_t s = 12345;
    // scene: R = rotation about y by 0.2, t = (1, 0.1, 0.05) normalised; 50 % outliers
    std::vector<double> x1(N), y1(N), x2(N), y2(N);
    const double c = cos(0.2), sn = sin(0.2), tn = sqrt(1 + 0.01 + 0.0025), t[3] = {1 / tn, 0.1 / tn, 0.05 / tn};
    for (uint32_t i = 0; i < N; ++i) {
        const double u = (uni(s) - 0.5) * 1.4, v = (uni(s) - 0.5) * 1.4, d = 2 + 6 * uni(s);
        const double X[3] = {u * d, v * d, d};
        const double Y[3] = {c * X[0] + sn * X[2] + t[0], X[1] + t[1], -sn * X[0] + c * X[2] + t[2]};
        x1[i] = u + 5e-4 * (uni(s) - 0.5), y1[i] = v + 5e-4 * (uni(s) - 0.5);
        x2[i] = Y[0] / Y[2] + 5e-4 * (uni(s) - 0.5), y2[i] = Y[1] / Y[2] + 5e-4 * (uni(s) - 0.5);
        if (uni(s) < 0.5) x2[i] = (uni(s) - 0.5) * 1.4, y2[i] = (uni(s) - 0.5) * 1.4;
    }
    double *d_pts; CK(hipMalloc(&d_pts, sizeof(double) * 4 * N));
    CK(hipMemcpy(d_pts, x1.data(), 8 * N, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pts + N, y1.data(), 8 * N, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pts + 2 * N, x2.data(), 8 * N, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pts + 3 * N, y2.data(), 8 * N, hipMemcpyHostToDevice));
    std::vector<uint32_t> samp((size_t)B * 5);
    for (uint32_t i = 0; i < B; ++i) for (int k = 0; k < 5; ++k) { bool dup; do { samp[i * 5 + k] = (uint32_t)(sm(s) % N); dup = false; for (int j = 0; j < k; ++j) dup |= samp[i * 5 + j] == samp[i * 5 + k]; } while (dup); }
    uint32_t *d_samp; CK(hipMalloc(&d_samp, 4 * samp.size())); CK(hipMemcpy(d_samp, samp.data(), 4 * samp.size(), hipMemcpyHostToDevice));
    GenerateArgs g; memset(&g, 0, sizeof(g));
    g.pts.n = N; for (int d = 0; d < 4; ++d) g.pts.a[d] = d_pts + (size_t)d * N;
    g.samples = d_samp; g.num_iters = B; g.slots_per_iter = slots;
    BatchCtl *ctl; CK(hipMalloc(&ctl, sizeof(BatchCtl) + 4 * 8192)); CK(hipMemset(ctl, 0, sizeof(BatchCtl) + 4 * 8192));
    g.ctl = ctl; g.blk_tot = reinterpret_cast<uint32_t *>(ctl + 1); g.blk_nan = g.blk_tot + 4096;
    const size_t mbytes = sizeof(double) * kModelStride * (size_t)B * slots;
    CK(hipMalloc(&g.models, mbytes)); CK(hipMemset(g.models, 0, mbytes));
    CK(hipMalloc(&g.num_models, 4 * B));
    const size_t sb = generate_stage_bytes(EST_REL, B);
    CK(hipMalloc(&g.stage, sb + 64)); CK(hipMemset(g.stage, 0, sb + 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipMemset(ctl, 0, sizeof(BatchCtl) + 4 * 8192));
        CK(hipEventRecord(e0, 0));
        CK(launch_generate(EST_REL, g, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) best = ms < best ? ms : best;
    }
    std::vector<uint32_t> nm(B); CK(hipMemcpy(nm.data(), g.num_models, 4 * B, hipMemcpyDeviceToHost));
    std::vector<double> models((size_t)B * slots * kModelStride); CK(hipMemcpy(models.data(), g.models, mbytes, hipMemcpyDeviceToHost));
    BatchCtl hc; CK(hipMemcpy(&hc, ctl, sizeof(hc), hipMemcpyDeviceToHost));
    uint64_t tot = 0, h = fnv(nm.data(), 4 * B); uint32_t hist[41] = {0};
    for (uint32_t i = 0; i < B; ++i) { tot += nm[i]; hist[nm[i] > 40 ? 40 : nm[i]]++; h = fnv(models.data() + (size_t)i * slots * kModelStride, sizeof(double) * kModelStride * nm[i], h); }
    std::vector<uint32_t> bt(8192); CK(hipMemcpy(bt.data(), g.blk_tot, 4 * 8192, hipMemcpyDeviceToHost));
    uint64_t bsum = 0, nsum = 0; for (int i = 0; i < 4096; ++i) bsum += bt[i], nsum += bt[4096 + i];
    printf("B %u slots %u: generator best %.3f ms; models %llu (blk_tot %llu, nan %llu) overflow %u checksum %016llx\n", B, slots, best,
           (unsigned long long)tot, (unsigned long long)bsum, (unsigned long long)nsum, hc.gen_overflow, (unsigned long long)h);
    printf("models per iteration histogram:"); for (int i = 0; i <= 16; ++i) printf(" %u", hist[i]); printf("\n");
    return 0;
}
