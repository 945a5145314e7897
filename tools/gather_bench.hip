// gather_bench.hip -- measures the ceiling the walk kernel lives under: dependent
// random gathers of B bytes per lane from a table much larger than the 256 MiB
// Infinity Cache.  Each lane runs one chain: the next row index is a hash of the
// row just loaded (so the load cannot be hoisted), exactly like the backward search.
// Prints rows/s and GB/s of useful bytes for B in {16,32,64} and several occupancies.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);       \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

template <int B>
__global__ void __launch_bounds__(256) k_chase(const uint4* __restrict__ tab, uint64_t nrows, int iters,
                                               uint64_t* sink) {
    uint64_t idx = mix(blockIdx.x * 256ull + threadIdx.x + 1) % nrows;
    uint64_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint4* p = tab + idx * (B / 16);
        uint4 v = p[0];
        uint64_t h = v.x ^ ((uint64_t)v.y << 32);
        if (B >= 32) {
            uint4 w = p[1];
            h ^= w.z;
        }
        if (B >= 64) {
            uint4 w = p[2];
            uint4 z = p[3];
            h ^= w.x ^ z.w;
        }
        acc += h;
        idx = mix(h + idx + i) % nrows;
    }
    if (acc == 0x1234567) sink[0] = acc;
}

// 32-byte rows fetched by lane PAIRS: in each of the two load instructions lanes 2j and 2j + 1 take the two halves of ONE
// row (row of lane 2j, then row of lane 2j + 1), so that an instruction touches 32 distinct lines instead of 64; the halves
// go to their owners by DPP (quad_perm [1,0,3,2]).  Same bytes, same lines, half the per-instruction line lookups.
__device__ __forceinline__ uint32_t swap1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); }
__global__ void __launch_bounds__(256) k_chase_paired(const uint4* __restrict__ tab, uint64_t nrows, int iters, uint64_t* sink) {
    uint64_t idx = mix(blockIdx.x * 256ull + threadIdx.x + 1) % nrows;
    uint64_t acc = 0;
    const bool odd = threadIdx.x & 1;
    for (int i = 0; i < iters; ++i) {
        const uint64_t pidx = ((uint64_t)swap1((uint32_t)(idx >> 32)) << 32) | swap1((uint32_t)idx);  // the partner's row
        const uint64_t ra = odd ? pidx : idx, rb = odd ? idx : pidx;  // rows of the even lane, of the odd lane
        const uint4 a = tab[ra * 2 + (odd ? 1 : 0)];  // instruction 1: the even lane's row, its two halves side by side
        const uint4 b = tab[rb * 2 + (odd ? 1 : 0)];  // instruction 2: the odd lane's row
        // even lane: own half 0 = a, own half 1 = partner's a; odd lane: own half 0 = partner's b, own half 1 = b
        const uint32_t sax = swap1(a.x), say = swap1(a.y), saz = swap1(a.z), sbx = swap1(b.x), sby = swap1(b.y);
        const uint32_t h0x = odd ? sbx : a.x, h0y = odd ? sby : a.y, h1z = odd ? b.z : saz;
        (void)sax;
        (void)say;
        uint64_t h = h0x ^ ((uint64_t)h0y << 32);
        h ^= h1z;
        acc += h;
        idx = mix(h + idx + i) % nrows;
    }
    if (acc == 0x1234567) sink[0] = acc;
}

void run_paired(const uint4* tab, uint64_t bytes, uint64_t* sink, int blocks_per_cu, int ncu) {
    const uint64_t nrows = bytes / 32;
    const int iters = 2000;
    const int grid = blocks_per_cu * ncu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_chase_paired<<<grid, 256>>>(tab, nrows, 100, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_chase_paired<<<grid, 256>>>(tab, nrows, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double loads = (double)grid * 256 * iters;
    printf("B= 32 PAIRED lanes, blocks/CU=%d (waves/CU=%2d): %8.2f Grows/s  latency/iter %.0f ns\n", blocks_per_cu, blocks_per_cu * 4,
           loads / ms / 1e6, ms * 1e6 / iters);
}

// calibration: two 16-byte gathers per iteration, either in the two 64-byte halves of
// ONE random 128-byte-aligned line (SAME=1) or in two different random lines (SAME=0).
// If the L2 fills whole 128-byte lines, SAME=1 costs one miss per iteration.
template <int SAME>
__global__ void __launch_bounds__(256) k_pair(const uint4* __restrict__ tab, uint64_t nlines, int iters,
                                              uint64_t* sink) {
    uint64_t idx = mix(blockIdx.x * 256ull + threadIdx.x + 1) % nlines;
    uint64_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint64_t idx2 = SAME ? idx : mix(idx + 0x9e3779b97f4a7c15ull) % nlines;
        uint4 v = tab[idx * 8];
        uint4 w = tab[idx2 * 8 + 4];
        uint64_t h = v.x ^ ((uint64_t)v.y << 32) ^ w.z;
        acc += h;
        idx = mix(h + idx + i) % nlines;
    }
    if (acc == 0x1234567) sink[0] = acc;
}

template <int SAME>
void run_pair(const uint4* tab, uint64_t bytes, uint64_t* sink, int blocks_per_cu, int ncu) {
    const uint64_t nlines = bytes / 128;
    const int iters = 2000;
    const int grid = blocks_per_cu * ncu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    k_pair<SAME><<<grid, 256>>>(tab, nlines, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double it = (double)grid * 256 * iters;
    printf("PAIR %s 128B line, blocks/CU=%d: %8.2f G iterations/s (2 x 16 B gathers each), %llu iterations\n",
           SAME ? "two halves of ONE" : "two DIFFERENT", blocks_per_cu, it / ms / 1e6, (unsigned long long)it);
}

__global__ void k_fill(uint4* tab, uint64_t n16) {
    uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i < n16) {
        uint64_t h = mix(i + 7);
        tab[i] = make_uint4((uint32_t)h, (uint32_t)(h >> 32), (uint32_t)(h * 3), (uint32_t)(h * 7));
    }
}

template <int B>
void run(const uint4* tab, uint64_t bytes, uint64_t* sink, int blocks_per_cu, int ncu) {
    const uint64_t nrows = bytes / B;
    const int iters = 2000;
    const int grid = blocks_per_cu * ncu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_chase<B><<<grid, 256>>>(tab, nrows, 100, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_chase<B><<<grid, 256>>>(tab, nrows, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double loads = (double)grid * 256 * iters;
    printf("B=%3d blocks/CU=%d (waves/CU=%2d) lanes=%8.0f : %8.2f Grows/s  %8.1f GB/s useful  %8.1f GB/s @64B-sector  latency/iter %.0f ns\n",
           B, blocks_per_cu, blocks_per_cu * 4, (double)grid * 256, loads / ms / 1e6, loads * B / ms / 1e6,
           loads * (B < 64 ? 64 : B) / ms / 1e6, ms * 1e6 / iters);
}

int main(int argc, char** argv) {
    uint64_t gb = argc > 1 ? atoll(argv[1]) : 16;
    uint64_t bytes = gb << 30;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, %.1f GB, table %llu GB\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.totalGlobalMem / 1e9, (unsigned long long)gb);
    uint4* tab;
    uint64_t* sink;
    CK(hipMalloc(&tab, bytes));
    CK(hipMalloc(&sink, 8));
    k_fill<<<(unsigned)((bytes / 16 + 255) / 256), 256>>>(tab, bytes / 16);
    CK(hipDeviceSynchronize());
    const int ncu = prop.multiProcessorCount;
    if (argc > 2) {  // calibration mode: one kernel shape only (for rocprofv3 --pmc)
        const int mode = atoi(argv[2]);
        if (mode == 0) run<16>(tab, bytes, sink, 4, ncu);
        if (mode == 1) run_pair<1>(tab, bytes, sink, 4, ncu);
        if (mode == 2) run_pair<0>(tab, bytes, sink, 4, ncu);
        return 0;
    }
    run_pair<1>(tab, bytes, sink, 4, ncu);
    run_pair<0>(tab, bytes, sink, 4, ncu);
    for (int bpc : {1, 2, 4, 8}) {
        run<16>(tab, bytes, sink, bpc, ncu);
        run<32>(tab, bytes, sink, bpc, ncu);
        run_paired(tab, bytes, sink, bpc, ncu);
        run<64>(tab, bytes, sink, bpc, ncu);
    }
    return 0;
}
