// gather_modes.hip -- does a dependent random 16-byte gather have to cost a whole 128-byte line of
// HBM traffic?  gather_bench.hip found 50 G gathers/s = 50 G line fills/s x 128 B = the achievable HBM
// bandwidth.  This bench repeats the chase with the table in differently typed memory (default
// coarse-grained, fine-grained, uncached) and with the cache-control bits a gfx950 load can carry
// (sc0, sc1, nt), to see whether any combination fetches less than a line per gather.
//   gather_modes.bin <GB> [alloc [flavour]]   (alloc / flavour given: one kernel only, for rocprofv3 --pmc)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int F>
__device__ __forceinline__ u32x4 load16(const u32x4* p) {
    u32x4 v;
    if (F == 0) return *p;
    if (F == 1) return __builtin_nontemporal_load(p);
    if (F == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
static const char* FNAME[] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt"};

template <int F>
__global__ void __launch_bounds__(256) k_chase(const u32x4* __restrict__ tab, uint64_t nrows, int iters, uint64_t* sink) {
    uint64_t idx = mix(blockIdx.x * 256ull + threadIdx.x + 1) % nrows;
    uint64_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        u32x4 v = load16<F>(tab + idx);
        uint64_t h = v.x ^ ((uint64_t)v.y << 32);
        acc += h;
        idx = mix(h + idx + i) % nrows;
    }
    if (acc == 0x1234567) sink[0] = acc;
}

__global__ void k_fill(u32x4* tab, uint64_t n16) {
    uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i < n16) {
        uint64_t h = mix(i + 7);
        u32x4 v = {(uint32_t)h, (uint32_t)(h >> 32), (uint32_t)(h * 3), (uint32_t)(h * 7)};
        tab[i] = v;
    }
}

template <int F>
void run(const char* aname, const u32x4* tab, uint64_t bytes, uint64_t* sink, int ncu) {
    const uint64_t nrows = bytes / 16;
    const int iters = 1000, grid = 4 * ncu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_chase<F><<<grid, 256>>>(tab, nrows, 50, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_chase<F><<<grid, 256>>>(tab, nrows, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double loads = (double)grid * 256 * iters;
    printf("alloc=%-12s load=%-11s : %7.2f G gathers/s  (%.0f ns per dependent gather)\n", aname, FNAME[F], loads / ms / 1e6,
           ms * 1e6 / iters);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const uint64_t gb = argc > 1 ? atoll(argv[1]) : 16;
    const int only_a = argc > 2 ? atoi(argv[2]) : -1, only_f = argc > 3 ? atoi(argv[3]) : -1;
    const uint64_t bytes = gb << 30;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, table %llu GB\n", prop.gcnArchName, prop.multiProcessorCount, (unsigned long long)gb);
    uint64_t* sink;
    CK(hipMalloc(&sink, 8));
    const char* ANAME[] = {"coarse", "fine-grained", "uncached"};
    for (int a = 0; a < 3; ++a) {
        if (only_a >= 0 && a != only_a) continue;
        u32x4* tab = nullptr;
        hipError_t e = a == 0   ? hipMalloc((void**)&tab, bytes)
                       : a == 1 ? hipExtMallocWithFlags((void**)&tab, bytes, hipDeviceMallocFinegrained)
                                : hipExtMallocWithFlags((void**)&tab, bytes, hipDeviceMallocUncached);
        if (e != hipSuccess) {
            printf("alloc=%s: %s\n", ANAME[a], hipGetErrorString(e));
            (void)hipGetLastError();
            continue;
        }
        k_fill<<<(unsigned)((bytes / 16 + 255) / 256), 256>>>(tab, bytes / 16);
        CK(hipDeviceSynchronize());
        const int ncu = prop.multiProcessorCount;
#define RUN(F) \
    if (only_f < 0 || only_f == F) run<F>(ANAME[a], tab, bytes, sink, ncu);
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
        CK(hipFree(tab));
    }
    return 0;
}
