// sparse_lines_calib.hip -- standalone calibration (hipcc -O2 --offload-arch=gfx950 -o tools/sparse_lines_calib tools/sparse_lines_calib.hip).
// What does it cost to fetch a FRACTION p of the 128-byte lines of a large column, by access shape?  The fused filter+project
// kernel loads the columns of later conjuncts only for rows still alive: cfg 3's l_quantity needs 47 % of its lines, its
// l_extendedprice 25 % -- and those stages run at 4.3 TB/s of fetched lines while the dense stages run at 7.5 TB/s.  Is that the
// memory system (DRAM pages opened for fewer bursts) or the access shape (wave instructions with a few active lanes)?
//   A   compacted lines, all lanes: a wave instruction fetches 8 SELECTED lines (lane l reads 16 B of line sel[l / 8])
//   B8  masked, in place: a wave instruction spans 8 CONSECUTIVE lines, the 8 lanes of every selected line are active
//   B1  masked, in place, ONE lane (16 B) active per selected line -- the fused kernel's shape when one row of a line is alive
//   C   compacted rows: a wave instruction fetches 16 B from 64 different selected lines (one lane per line)
// Same selected lines in all four (hash of the line index < p); time per pass and GB/s of LINES fetched (128 B each).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__host__ __device__ inline bool selected(u64 line, u32 thr) {
    u32 h = (u32)line * 2654435761u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    return (h >> 8) < thr;   // thr = p * 2^24
}

constexpr int kRegion = 512;   // lines per wave step region (64 KiB): a wave walks its region, waves stride over the buffer

// B8 / B1: in place.  MODE 8: all 8 lanes of a selected line; MODE 1: lane (line & 7) of the line's 8 lanes only
template <int MODE>
__global__ void __launch_bounds__(256) k_masked(const u64x2 *src, u64 nlines, u32 thr, u64 *sink) {
    const int lane = threadIdx.x & 63;
    const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (u64)gridDim.x * 4;
    u64 acc = 0;
    for (u64 r0 = wave * kRegion; r0 + kRegion <= nlines; r0 += nwaves * kRegion) {
        for (int g = 0; g < kRegion; g += 64) {   // 8 instructions of 8 lines in flight
            u64x2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u64 line = r0 + g + j * 8 + (lane >> 3);
                v[j].x = 0; v[j].y = 0;
                const bool on = selected(line, thr) && (MODE == 8 || (lane & 7) == (int)(line & 7));
                if (on) v[j] = __builtin_nontemporal_load(src + line * 8 + (lane & 7));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j].x ^ v[j].y;
        }
    }
    if (acc == 0x0123456789abcdefull) sink[0] = acc;
}

// B8' / B1': as B8 / B1, but the selection comes from a precomputed bitmap (one wave-uniform 8-byte load per 64 lines) instead
// of a hash per lane and load: the mask costs nothing, as in the fused kernel (where it is the result of a compare)
template <int MODE>
__global__ void __launch_bounds__(256) k_masked_bm(const u64x2 *src, u64 nlines, const u64 *bm, u64 *sink) {
    const int lane = threadIdx.x & 63;
    const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (u64)gridDim.x * 4;
    u64 acc = 0;
    for (u64 r0 = wave * kRegion; r0 + kRegion <= nlines; r0 += nwaves * kRegion) {
        for (int g = 0; g < kRegion; g += 64) {
            const u64 m = bm[(r0 + g) >> 6];
            u64x2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u64 line = r0 + g + j * 8 + (lane >> 3);
                v[j].x = 0; v[j].y = 0;
                const bool on = ((m >> (j * 8 + (lane >> 3))) & 1ull) && (MODE == 8 || (lane & 7) == (int)(line & 7));
                if (on) v[j] = __builtin_nontemporal_load(src + line * 8 + (lane & 7));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j].x ^ v[j].y;
        }
    }
    if (acc == 0x0123456789abcdefull) sink[0] = acc;
}

// A / C: through the list of selected lines.  MODE 8: 8 lines per instruction, all 16-byte pieces; MODE 1: 64 lines per
// instruction, one 16-byte piece each
template <int MODE>
__global__ void __launch_bounds__(256) k_compact(const u64x2 *src, const u32 *sel, u64 nsel, u64 *sink) {
    const int lane = threadIdx.x & 63;
    const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (u64)gridDim.x * 4;
    constexpr u64 per_instr = MODE == 8 ? 8 : 64;
    constexpr u64 step = per_instr * 8;   // selected lines per wave step (8 instructions in flight)
    u64 acc = 0;
    for (u64 s0 = wave * step; s0 + step <= nsel; s0 += nwaves * step) {
        u32 id[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) id[j] = sel[s0 + j * per_instr + (MODE == 8 ? (lane >> 3) : lane)];
        u64x2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(src + (u64)id[j] * 8 + (MODE == 8 ? (lane & 7) : (id[j] & 7)));
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j].x ^ v[j].y;
    }
    if (acc == 0x0123456789abcdefull) sink[0] = acc;
}

int main(int argc, char **argv) {
    const u64 nbytes = (argc > 1 ? std::atoll(argv[1]) : 8ll) << 30;
    const int wgs_per_cu = argc > 2 ? std::atoi(argv[2]) : 6;
    const u64 nlines = nbytes / 128;
    void *d = nullptr;
    u64 *sink = nullptr;
    u32 *d_sel = nullptr;
    CK(hipMalloc(&d, nbytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&d_sel, nlines * 4));
    CK(hipMemset(d, 1, nbytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const dim3 grid(256 * wgs_per_cu), block(256);
    std::vector<u32> h_sel(nlines);
    std::vector<u64> h_bm(nlines / 64);
    u64 *d_bm = nullptr;
    CK(hipMalloc(&d_bm, nlines / 8));
    std::printf("buffer %.1f GB, %llu lines, %d workgroups per CU\n", nbytes / 1e9, nlines, wgs_per_cu);
    for (double p : {1.0, 0.75, 0.5, 0.25, 0.1, 0.03}) {
        const u32 thr = p >= 1.0 ? (1u << 24) : (u32)(p * (1 << 24));
        u64 nsel = 0;
        for (u64 l = 0; l < nlines; ++l)
            if (selected(l, thr)) h_sel[nsel++] = (u32)l;
        CK(hipMemcpy(d_sel, h_sel.data(), nsel * 4, hipMemcpyHostToDevice));
        std::fill(h_bm.begin(), h_bm.end(), 0ull);
        for (u64 i = 0; i < nsel; ++i) h_bm[h_sel[i] >> 6] |= 1ull << (h_sel[i] & 63);
        CK(hipMemcpy(d_bm, h_bm.data(), nlines / 8, hipMemcpyHostToDevice));
        auto time = [&](auto launch) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0) best = std::min(best, ms);
            }
            return best;
        };
        const float tA = time([&] { hipLaunchKernelGGL(k_compact<8>, grid, block, 0, 0, (const u64x2 *)d, d_sel, nsel, sink); });
        const float tB8 = time([&] { hipLaunchKernelGGL(k_masked<8>, grid, block, 0, 0, (const u64x2 *)d, nlines, thr, sink); });
        const float tB1 = time([&] { hipLaunchKernelGGL(k_masked<1>, grid, block, 0, 0, (const u64x2 *)d, nlines, thr, sink); });
        const float tC = time([&] { hipLaunchKernelGGL(k_compact<1>, grid, block, 0, 0, (const u64x2 *)d, d_sel, nsel, sink); });
        const float tM8 = time([&] { hipLaunchKernelGGL(k_masked_bm<8>, grid, block, 0, 0, (const u64x2 *)d, nlines, d_bm, sink); });
        const float tM1 = time([&] { hipLaunchKernelGGL(k_masked_bm<1>, grid, block, 0, 0, (const u64x2 *)d, nlines, d_bm, sink); });
        const double gb = nsel * 128.0 / 1e9;
        std::printf("p %.2f lines %.2f GB | A compact x8 %.3f ms %.0f GB/s | B8 masked x8 %.3f ms %.0f GB/s | B1 masked 1 lane/line %.3f ms %.0f GB/s | C compact 1 lane/line %.3f ms %.0f GB/s\n",
                    p, gb, tA, gb / tA * 1e3, tB8, gb / tB8 * 1e3, tB1, gb / tB1 * 1e3, tC, gb / tC * 1e3);
        std::printf("       bitmap-masked: x8 %.3f ms %.0f GB/s | 1 lane/line %.3f ms %.0f GB/s\n", tM8, gb / tM8 * 1e3, tM1, gb / tM1 * 1e3);
        std::fflush(stdout);
    }
    return 0;
}
