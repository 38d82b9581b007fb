// Experiment: is v_mfma_f64_4x4x4_4b_f64 / v_mfma_f64_16x16x4_f64 a k-ORDERED chain of fused multiply-adds with one rounding per step,
// i.e. D = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C))))?  With b = 1 that is the sequential sum fl(fl(fl(fl(C + a0) + a1) + a2) + a3) -
// the reference's summation order (jacobian_accumulator.h:82-97) delivered by the matrix pipe, four terms per instruction.
//   1. operand layout by one-hot probing, 2. random operands with a wide exponent range against the candidate orders on the host.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_f64_order.cc -o mfma_f64_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k4(const double *a, const double *b, const double *c, double *d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}
__global__ void k16(const double *a, const double *b, const double *c, double *d) {
    const int l = threadIdx.x;
    double4_t cc = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
    double4_t r = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], cc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[4 * l + i] = r[i];
}
int main() {
    double *da, *db, *dc, *dd;
    CK(hipMalloc(&da, 8 * 64)); CK(hipMalloc(&db, 8 * 64)); CK(hipMalloc(&dc, 8 * 256)); CK(hipMalloc(&dd, 8 * 256));
    std::vector<double> a(64), b(64), c(256), d(256);
    auto run4 = [&]() { hipMemcpy(da, a.data(), 8 * 64, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 8 * 64, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 8 * 64, hipMemcpyHostToDevice);
                        k4<<<1, 64>>>(da, db, dc, dd); hipMemcpy(d.data(), dd, 8 * 64, hipMemcpyDeviceToHost); };
    auto run16 = [&]() { hipMemcpy(da, a.data(), 8 * 64, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 8 * 64, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 8 * 256, hipMemcpyHostToDevice);
                         k16<<<1, 64>>>(da, db, dc, dd); hipMemcpy(d.data(), dd, 8 * 256, hipMemcpyDeviceToHost); };
    // ---- layout of 4x4x4_4b: which (a lane, b lane) pairs feed output lane o?  a one-hot, b = lane-coded
    // contrib[o] = list of (la, lb)
    std::vector<std::vector<std::pair<int, int>>> con4(64);
    for (int la = 0; la < 64; ++la) {
        std::fill(a.begin(), a.end(), 0.0); a[la] = 1.0;
        for (int l = 0; l < 64; ++l) b[l] = (double)(l + 1);
        std::fill(c.begin(), c.end(), 0.0);
        run4();
        for (int o = 0; o < 64; ++o) if (d[o] != 0.0) con4[o].push_back({la, (int)d[o] - 1});
    }
    printf("4x4x4_4b: output lane o <- (a lane, b lane) pairs:\n");
    for (int o : {0, 1, 4, 5, 16, 17, 21, 63}) { printf("  o=%2d:", o); for (auto &p : con4[o]) printf(" (%d,%d)", p.first, p.second); printf("\n"); }
    std::vector<std::vector<std::pair<int, int>>> con16(256);
    for (int la = 0; la < 64; ++la) {
        std::fill(a.begin(), a.end(), 0.0); a[la] = 1.0;
        for (int l = 0; l < 64; ++l) b[l] = (double)(l + 1);
        std::fill(c.begin(), c.end(), 0.0);
        run16();
        for (int o = 0; o < 256; ++o) if (d[o] != 0.0) con16[o].push_back({la, (int)d[o] - 1});
    }
    printf("16x16x4: output (lane, reg) o = 4 lane + reg <- (a lane, b lane) pairs:\n");
    for (int o : {0, 1, 2, 3, 4, 64, 65, 255}) { printf("  o=%3d:", o); for (auto &p : con16[o]) printf(" (%d,%d)", p.first, p.second); printf("\n"); }
    // ---- numerics: random operands, wide exponents; candidates: k-ordered fma chain in the order of the contribution list (ascending a lane),
    // descending, products rounded then added (no fma), exact sum rounded once (long double as a stand-in)
    std::mt19937_64 rng(7);
    auto rnd = [&]() { std::uniform_real_distribution<double> m(-1, 1); std::uniform_int_distribution<int> e(-30, 30); return std::ldexp(m(rng), e(rng)); };
    for (int variant = 0; variant < 2; ++variant) {
        long n_tot = 0, m_fwd = 0, m_rev = 0, m_nofma = 0, m_once = 0, m_fwd_ones = 0;
        for (int ones = 0; ones < 2; ++ones) {
            for (int trial = 0; trial < 400; ++trial) {
                for (int l = 0; l < 64; ++l) { a[l] = rnd(); b[l] = ones ? 1.0 : rnd(); }
                const int nout = variant ? 256 : 64;
                for (int o = 0; o < nout; ++o) c[o] = rnd();
                if (variant) run16(); else run4();
                auto &con = variant ? con16 : con4;
                for (int o = 0; o < nout; ++o) {
                    double f = c[o], r = c[o], nf = c[o]; long double once = c[o];
                    for (size_t k = 0; k < con[o].size(); ++k) { f = std::fma(a[con[o][k].first], b[con[o][k].second], f); nf = nf + a[con[o][k].first] * b[con[o][k].second]; once += (long double)a[con[o][k].first] * b[con[o][k].second]; }
                    for (size_t k = con[o].size(); k-- > 0;) r = std::fma(a[con[o][k].first], b[con[o][k].second], r);
                    ++n_tot; m_fwd += f == d[o]; m_rev += r == d[o]; m_nofma += nf == d[o]; m_once += (double)once == d[o];
                    if (ones) m_fwd_ones += f == d[o];
                }
            }
        }
        printf("%s: %ld outputs: k-ascending fma chain %ld, k-descending %ld, rounded products then add %ld, one rounding of the exact sum %ld; with b = 1 (the sequential-sum use): %ld of %ld\n",
               variant ? "v_mfma_f64_16x16x4_f64" : "v_mfma_f64_4x4x4_4b_f64", n_tot, m_fwd, m_rev, m_nofma, m_once, m_fwd_ones, n_tot / 2);
    }
    return 0;
}
