// frag_bench.hip -- does it matter for random gathers WHERE (how contiguously) the driver placed a table in HBM?
// The walk at r = 2e9 ran 0-30 % slower depending on what the process had allocated and freed before the index was
// flattened (profiles/r04_c5_findings.txt).  Hypothesis: a table pieced together from small free holes is mapped with
// small page-table fragments, and dependent random gathers over 100+ GB then miss the TLBs more often.
//   1. dependent 16-byte gathers (the walk's access) over a table allocated from a fresh device: 16 / 64 / 160 GB
//   2. the device filled with PIECE-sized allocations, every other one freed, and the table allocated from the holes
//   usage: frag_bench.bin [piece MB (default 2)] [table GB for step 2 (default 16)]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                               \
    do {                                                                    \
        hipError_t e = (x);                                                 \
        if (e != hipSuccess) {                                              \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
            exit(1);                                                        \
        }                                                                   \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

__global__ void k_fill(uint4* t, uint64_t n) {
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t h = mix(i + 1);
        t[i] = uint4{(uint32_t)h, (uint32_t)(h >> 32), (uint32_t)i, 7u};
    }
}

__global__ void __launch_bounds__(256) k_chase(const uint4* __restrict__ tab, uint64_t nrows, int iters, uint64_t* sink) {
    uint64_t idx = mix(blockIdx.x * 256ull + threadIdx.x + 1) % nrows;
    uint64_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint4 v = tab[idx];
        const uint64_t h = v.x ^ ((uint64_t)v.y << 32);
        acc += h;
        idx = mix(h + idx + i) % nrows;
    }
    if (acc == 0x1234567) sink[0] = acc;
}

static double chase(const uint4* tab, uint64_t bytes, uint64_t* sink, int ncu, int blocks_per_cu = 4, double* ns = nullptr, unsigned tpb = 256) {
    const int iters = 400;
    const unsigned grid = (unsigned)ncu * blocks_per_cu;  // 4 x 256 threads: 16 wavefronts per CU, the walk's occupancy; 1 x 64: one per CU (latency)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_chase<<<grid, tpb>>>(tab, bytes / 16, 50, sink);
    double best = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k_chase<<<grid, tpb>>>(tab, bytes / 16, iters, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double rate = (double)grid * tpb * iters / ms / 1e6;
        if (rate > best) {
            best = rate;
            if (ns) *ns = ms * 1e6 / iters;
        }
    }
    return best;
}

static void both(const char* what, const uint4* t, uint64_t bytes, uint64_t* sink, int ncu) {
    double ns1 = 0, ns4 = 0;
    const double r4 = chase(t, bytes, sink, ncu, 4, &ns4), r1 = chase(t, bytes, sink, ncu, 1, &ns1, 64);
    printf("%-52s: %6.2f G gathers/s at 16 wavefronts per CU (%4.0f ns per dependent gather); one wavefront per CU: %5.2f G/s, %4.0f ns\n", what, r4, ns4, r1, ns1);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const uint64_t piece = (argc > 1 ? atoll(argv[1]) : 2) << 20;
    const uint64_t tab2 = (uint64_t)(argc > 2 ? atoll(argv[2]) : 16) << 30;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    uint64_t* sink;
    CK(hipMalloc(&sink, 8));
    size_t fr, tot;
    CK(hipMemGetInfo(&fr, &tot));
    printf("device %s, %d CUs, %.1f of %.1f GB free\n", prop.gcnArchName, ncu, fr / 1e9, tot / 1e9);
    for (uint64_t gb : {16ull, 64ull, 160ull}) {
        uint4* t;
        CK(hipMalloc(&t, gb << 30));
        k_fill<<<65536, 256>>>(t, (gb << 30) / 16);
        CK(hipDeviceSynchronize());
        char what[96];
        snprintf(what, sizeof what, "fresh device, one allocation of %3llu GB", (unsigned long long)gb);
        both(what, t, gb << 30, sink, ncu);
        CK(hipFree(t));
    }
    // fill the device: one big block, then pieces over 2 x the table's size (+ slack); free every other piece
    CK(hipMemGetInfo(&fr, &tot));
    const uint64_t pieces_bytes = 2 * tab2 + (2ull << 30);
    const uint64_t filler = fr > pieces_bytes + (3ull << 30) ? fr - pieces_bytes - (3ull << 30) : 0;
    void* fill = nullptr;
    if (filler) CK(hipMalloc(&fill, filler));
    std::vector<void*> ps;
    for (uint64_t got = 0; got < pieces_bytes; got += piece) {
        void* p;
        if (hipMalloc(&p, piece) != hipSuccess) break;
        ps.push_back(p);
    }
    for (size_t i = 0; i < ps.size(); i += 2) CK(hipFree(ps[i]));
    CK(hipMemGetInfo(&fr, &tot));
    printf("filler %.1f GB + %zu pieces of %llu MB, every other one freed: %.1f GB free\n", filler / 1e9, ps.size(),
           (unsigned long long)(piece >> 20), fr / 1e9);
    uint4* t;
    if (hipMalloc(&t, tab2) != hipSuccess) {
        printf("the table does not fit the holes\n");
        return 0;
    }
    k_fill<<<65536, 256>>>(t, tab2 / 16);
    CK(hipDeviceSynchronize());
    {
        char what[96];
        snprintf(what, sizeof what, "table of %llu GB from %llu MB holes", (unsigned long long)(tab2 >> 30), (unsigned long long)(piece >> 20));
        both(what, t, tab2, sink, ncu);
    }
    CK(hipFree(t));
    for (size_t i = 1; i < ps.size(); i += 2) CK(hipFree(ps[i]));
    if (fill) CK(hipFree(fill));
    CK(hipMalloc(&t, tab2));
    k_fill<<<65536, 256>>>(t, tab2 / 16);
    CK(hipDeviceSynchronize());
    both("everything freed, the same table again", t, tab2, sink, ncu);
    return 0;
}
