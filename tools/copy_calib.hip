// Roofline calibration for a read + write mix (the partitioned group-by's scatter pass reads 12 B and writes 12-16 B per row):
// streams R bytes in and W bytes out with 16-byte lane accesses, grid-stride, for a few store flavours and grid sizes.
//   build:  hipcc -O2 --offload-arch=gfx950 -o tools/copy_calib tools/copy_calib.hip      run: tools/copy_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// every wave iteration: NR 16-byte loads per lane from `src`, NW 16-byte stores per lane to `dst` (1 KiB per wave instruction)
template <int NR, int NW, int KIND>
__global__ void __launch_bounds__(256) rw_kernel(const u64x2 *__restrict__ src, u64x2 *__restrict__ dst, long long iters, u64 *sink) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    u64 acc = 0;
    for (long long it = wave; it < iters; it += nwaves) {
        u64x2 v[NR > 0 ? NR : 1];
#pragma unroll
        for (int k = 0; k < NR; ++k) v[k] = __builtin_nontemporal_load(src + (it * NR + k) * 64 + lane);
        u64x2 o;
        o.x = (u64)it; o.y = (u64)lane;
#pragma unroll
        for (int k = 0; k < NR; ++k) { o.x ^= v[k].x; o.y += v[k].y; }
        acc += o.x;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            u64x2 w = o; w.x += (u64)k;
            u64x2 *q = dst + (it * NW + k) * 64 + lane;
            if (KIND == 0) *q = w; else __builtin_nontemporal_store(w, q);
        }
    }
    if (acc == 0x1234567ull) *sink = acc;
}

template <int NR, int NW, int KIND>
static void run(const char *name, const u64x2 *src, u64x2 *dst, u64 *sink, long long iters, int wgs_per_cu) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wgs_per_cu;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((rw_kernel<NR, NW, KIND>), dim3(grid), dim3(256), 0, 0, src, dst, iters, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double rb = (double)iters * NR * 1024, wb = (double)iters * NW * 1024;
    std::printf("%-34s wg/cu %2d: read %5.1f GB write %5.1f GB  %7.3f ms  %6.0f GB/s total\n", name, wgs_per_cu, rb / 1e9, wb / 1e9, best, (rb + wb) / best / 1e6);
}

int main() {
    const long long iters = 4ll << 20;   // x 3 KiB read = 12.9 GB, x 4 KiB written = 17.2 GB
    u64x2 *src, *dst; u64 *sink;
    CK(hipMalloc(&src, (size_t)iters * 4 * 1024)); CK(hipMalloc(&dst, (size_t)iters * 4 * 1024)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(src, 1, (size_t)iters * 4 * 1024)); CK(hipMemset(dst, 0, (size_t)iters * 4 * 1024));
    for (int wg : {2, 4, 8}) {
        run<3, 0, 0>("read only (12 B/row)", src, dst, sink, iters, wg);
        run<4, 0, 0>("read only (16 B/row)", src, dst, sink, iters, wg);
        run<0, 4, 0>("write only 16 B/row, plain", src, dst, sink, iters, wg);
        run<0, 4, 1>("write only 16 B/row, nt", src, dst, sink, iters, wg);
        run<0, 3, 1>("write only 12 B/row, nt", src, dst, sink, iters, wg);
        run<3, 4, 0>("read 12 + write 16 B/row, plain", src, dst, sink, iters, wg);
        run<3, 4, 1>("read 12 + write 16 B/row, nt", src, dst, sink, iters, wg);
        run<3, 3, 0>("read 12 + write 12 B/row, plain", src, dst, sink, iters, wg);
        run<3, 3, 1>("read 12 + write 12 B/row, nt", src, dst, sink, iters, wg);
    }
    return 0;
}
