// Experiment (round 5, 5-point front end): what does a per-lane conditional move of a DOUBLE cost on gfx950?
// k_rel_front spends 30 % of its vector instructions on v_cndmask_b32 pairs (register-resident pivoting: a select of one
// double = two v_cndmask_b32_e64).  Alternatives measured here, 16 independent registers per wavefront, 1 and 2 wavefronts
// per SIMD (the occupancies the generator kernels run at):
//   cndmask pair          2 x v_cndmask_b32_e64 dst, dst, src, sgpr-mask         (what the compiler emits)
//   masked v_mov_b64 x N  s_and_saveexec_b64 + N x v_mov_b64 + s_mov_b64 exec     (one move per double under an EXEC mask)
//   v_swap_b32            register exchange (a conditional swap of two doubles = 2 of them under EXEC, instead of 4 selects)
//   v_mov_b64, v_accvgpr_read/write: the moves the register allocator adds
//   hipcc --offload-arch=gfx950 -O3 select_bench.cc -o select_bench && ./select_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Out { unsigned long long cyc, rt; double chk; };
#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define REP64(OP) REP16(OP) REP16(OP) REP16(OP) REP16(OP)

#define HEAD(NAME)                                                                                                       \
    __global__ __launch_bounds__(512) void NAME(int iters, Out *out, const double *seed) {                                \
        const int lane = threadIdx.x & 63;                                                                               \
        double r[16];                                                                                                    \
        for (int i = 0; i < 16; ++i) r[i] = seed[lane] * (double)(i + 1);                                                \
        double a = seed[64 + lane];                                                                                      \
        unsigned q[32];                                                                                                  \
        for (int i = 0; i < 32; ++i) q[i] = (unsigned)(lane * 37 + i);                                                   \
        unsigned qa = (unsigned)lane;                                                                                    \
        unsigned long long mk = (lane & 1) ? 0x5555555555555555ull : 0x5555555555555555ull;                              \
        mk = __builtin_amdgcn_readfirstlane((unsigned)mk) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mk >> 32)) << 32); \
        unsigned long long tmp = 0;                                                                                      \
        (void)tmp;                                                                                                       \
        __syncthreads();                                                                                                 \
        const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();                                 \
        for (int it = 0; it < iters; ++it) {
#define TAIL                                                                                                             \
        }                                                                                                                \
        const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();                                 \
        double acc = a;                                                                                                  \
        for (int i = 0; i < 16; ++i) acc += r[i];                                                                        \
        for (int i = 0; i < 32; ++i) acc += (double)q[i];                                                                \
        if (lane == 0) { Out o; o.cyc = t1 - t0; o.rt = w1 - w0; o.chk = acc; out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = o; } \
    }

// one select of a double = 2 cndmask on the halves (64 double-selects per iteration = 128 instructions)
#define OP_CND(i) asm volatile("v_cndmask_b32_e64 %0, %0, %2, %3\n v_cndmask_b32_e64 %1, %1, %2, %3" : "+v"(q[2 * i]), "+v"(q[2 * i + 1]) : "v"(qa), "s"(mk));
HEAD(k_cndmask_pair) REP64(OP_CND) TAIL
// masked moves, one exec set per move
#define OP_MM1(i) asm volatile("s_and_saveexec_b64 %2, %3\n v_mov_b64 %0, %1\n s_mov_b64 exec, %2" : "+v"(r[i]), "+v"(a), "+s"(tmp) : "s"(mk) : "scc");
HEAD(k_masked_mov_1) REP64(OP_MM1) TAIL
// masked moves, 16 per exec set (64 per iteration = 4 sets)
#define MM16 asm volatile("s_and_saveexec_b64 %17, %18\n v_mov_b64 %0, %16\n v_mov_b64 %1, %16\n v_mov_b64 %2, %16\n v_mov_b64 %3, %16\n v_mov_b64 %4, %16\n v_mov_b64 %5, %16\n v_mov_b64 %6, %16\n v_mov_b64 %7, %16\n v_mov_b64 %8, %16\n v_mov_b64 %9, %16\n v_mov_b64 %10, %16\n v_mov_b64 %11, %16\n v_mov_b64 %12, %16\n v_mov_b64 %13, %16\n v_mov_b64 %14, %16\n v_mov_b64 %15, %16\n s_mov_b64 exec, %17" \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(a), "+s"(tmp) : "s"(mk) : "scc");
HEAD(k_masked_mov_16) MM16 MM16 MM16 MM16 TAIL
// masked moves, 4 per exec set
#define MM4(b) asm volatile("s_and_saveexec_b64 %5, %6\n v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4\n s_mov_b64 exec, %5" \
    : "+v"(r[b]), "+v"(r[b + 1]), "+v"(r[b + 2]), "+v"(r[b + 3]), "+v"(a), "+s"(tmp) : "s"(mk) : "scc");
#define MM4x4 MM4(0) MM4(4) MM4(8) MM4(12)
HEAD(k_masked_mov_4) MM4x4 MM4x4 MM4x4 MM4x4 TAIL
#define OP_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "+v"(r[i]) : "v"(a));
HEAD(k_mov_b64) REP64(OP_MOV64) TAIL
// swap of two doubles = 2 v_swap_b32 (64 double-swaps per iteration = 128 instructions)
#define OP_SWAP(i) asm volatile("v_swap_b32 %0, %2\n v_swap_b32 %1, %3" : "+v"(q[2 * i]), "+v"(q[2 * i + 1]), "+v"(q[(2 * i + 2) & 31]), "+v"(q[(2 * i + 3) & 31]));
HEAD(k_swap_pair) REP64(OP_SWAP) TAIL
#define OP_ACC(i) asm volatile("v_accvgpr_write_b32 a" #i ", %0\n v_accvgpr_read_b32 %0, a" #i : "+v"(q[i]) : : "a" #i);
HEAD(k_accvgpr_wr_rd) REP64(OP_ACC) TAIL
#define OP_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(a));
HEAD(k_add_f64) REP64(OP_ADD64) TAIL
// fp64 work with selects in between, as in the LU: 1 add + 1 select per element
#define OP_ADD_CND(i) asm volatile("v_add_f64 %0, %0, %3\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5" : "+v"(r[i]), "+v"(q[2 * i]), "+v"(q[2 * i + 1]) : "v"(a), "v"(qa), "s"(mk));
HEAD(k_add_then_cndpair) REP64(OP_ADD_CND) TAIL

typedef void (*Kern)(int, Out *, const double *);
struct Entry { const char *name; Kern k; int instr_per_iter; int iters; };

int main() {
    Out *out; CK(hipMalloc(&out, sizeof(Out) * 65536));
    double *seed; CK(hipMalloc(&seed, 8 * 128));
    std::vector<double> hs(128); for (int i = 0; i < 128; ++i) hs[i] = 1.0 + 1e-3 * i;
    CK(hipMemcpy(seed, hs.data(), 8 * 128, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const Entry tab[] = {
        {"select of a double: 2 x v_cndmask_b32_e64 (per double)", k_cndmask_pair, 64, 2000},
        {"masked v_mov_b64, 1 per exec set (per double)", k_masked_mov_1, 64, 2000},
        {"masked v_mov_b64, 4 per exec set (per double)", k_masked_mov_4, 64, 2000},
        {"masked v_mov_b64, 16 per exec set (per double)", k_masked_mov_16, 64, 2000},
        {"v_mov_b64 (per instruction)", k_mov_b64, 64, 2000},
        {"swap of two doubles: 2 x v_swap_b32 (per double pair)", k_swap_pair, 64, 2000},
        {"v_accvgpr_write + read (per pair)", k_accvgpr_wr_rd, 64, 2000},
        {"v_add_f64 (per instruction)", k_add_f64, 64, 2000},
        {"v_add_f64 + select of the result (per element)", k_add_then_cndpair, 64, 2000},
    };
    printf("device %s, %d CUs\n\ncycles per UNIT (see the row) per SIMD, from the HIP-event wall time at the clock measured in the launch\n\n", prop.name, cus);
    printf("| what | 1 wave/SIMD | 2 waves/SIMD | MHz |\n|---|---|---|---|\n");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (const Entry &e : tab) {
        printf("| %s |", e.name);
        double mhz_last = 0;
        for (int w : {1, 2}) {
            const int threads = 256 * w, blocks = cus, waves = blocks * threads / 64;
            e.k<<<blocks, threads>>>(e.iters / 8, out, seed);
            CK(hipEventRecord(e0, 0));
            e.k<<<blocks, threads>>>(e.iters, out, seed);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<Out> h(waves); CK(hipMemcpy(h.data(), out, sizeof(Out) * waves, hipMemcpyDeviceToHost));
            double cyc = 0, rt = 0; for (const Out &o : h) { cyc += (double)o.cyc; rt += (double)o.rt; }
            cyc /= waves; rt /= waves;
            const double mhz = cyc / (rt / 100.0);
            const double n = (double)e.iters * e.instr_per_iter;
            printf(" %.2f |", ms * 1e-3 * mhz * 1e6 / n / w);
            mhz_last = mhz;
        }
        printf(" %.0f |\n", mhz_last);
        fflush(stdout);
    }
    return 0;
}
