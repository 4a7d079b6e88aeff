// Write-pattern calibration for the partitioned group-by's scatter pass: every wave appends runs of RUN 16-byte records to
// NP append regions of its own (one per partition), 64 records per store instruction -- the scatter's store stream without
// its loads and LDS work.  RUN = 16 / 24 keep every run a whole number of 128-byte lines; RUN = 21 is what 100 000 keys give.
//   build:  hipcc -O2 --offload-arch=gfx950 -o tools/append_calib tools/append_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// region of (wave, partition): `steps * RUN` records, laid out partition-major like the real thing: base = (part * nwaves + wave) * steps * RUN
template <int RUN, int RECB>
__global__ void __launch_bounds__(256) append_kernel(char *dst, int np, int steps, int skew) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const int per_step = (1024 / RUN) * RUN;       // records a step appends (<= 1024)
    for (int st = 0; st < steps; ++st) {
        const int p0 = (st * (1024 / RUN)) % np;   // rotate through the partitions
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int j = k * 64 + lane;
            if (j >= per_step) continue;
            const int r = j / RUN, o = j % RUN;
            const int part = (p0 + r) % np;
            const long long visits = (long long)st * (1024 / RUN) / np;      // earlier appends to this region (approx. uniform)
            const long long rec = ((long long)part * nwaves + wave) * ((long long)steps * (1024 / RUN) / np + 2) * RUN + visits * RUN + o;
            char *q = dst + rec * RECB + skew;
            if (RECB == 16) { u64x2 w; w.x = (u64)rec; w.y = (u64)st; *(u64x2 *)q = w; }
            else if (RECB == 8) *(u64 *)q = (u64)rec;
            else *(unsigned *)q = (unsigned)rec;
        }
    }
}

template <int RUN, int RECB>
static void run(char *dst, int np, int steps, int wgs_per_cu, int skew) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wgs_per_cu;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((append_kernel<RUN, RECB>), dim3(grid), dim3(256), 0, 0, dst, np, steps, skew);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double wb = (double)grid * 4 * steps * ((1024 / RUN) * RUN) * RECB;
    std::printf("run %3d x %2d B  partitions %4d  wg/cu %d  skew %2d: %6.2f GB  %7.3f ms  %6.0f GB/s\n", RUN, RECB, np, wgs_per_cu, skew, wb / 1e9, best, wb / best / 1e6);
}

int main() {
    char *dst;
    const size_t cap = 24ull << 30;
    CK(hipMalloc(&dst, cap));
    CK(hipMemset(dst, 0, cap));
    for (int wg : {2, 4}) {
        const int steps = 1000000000 / (256 * wg * 4 * 1024);   // ~1 B records in all
        for (int np : {64, 512}) {
            run<16, 16>(dst, np, steps, wg, 0);
            run<16, 16>(dst, np, steps, wg, 16);
            run<21, 16>(dst, np, steps, wg, 0);
            run<24, 16>(dst, np, steps, wg, 0);
            run<64, 16>(dst, np, steps, wg, 0);
            run<64, 16>(dst, np, steps, wg, 48);
            run<2, 16>(dst, np, steps, wg, 0);
            run<8, 16>(dst, np, steps, wg, 0);
            run<4, 16>(dst, np, steps, wg, 0);    // 64-byte runs: half lines
            run<12, 16>(dst, np, steps, wg, 0);   // 192 bytes: whole 64-byte sectors, not whole lines
            run<20, 16>(dst, np, steps, wg, 0);
            run<21, 8>(dst, np, steps, wg, 0);
            run<21, 4>(dst, np, steps, wg, 0);
            run<64, 8>(dst, np, steps, wg, 0);
            run<64, 4>(dst, np, steps, wg, 0);
        }
    }
    return 0;
}
