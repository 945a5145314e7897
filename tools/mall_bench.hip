// mall_bench -- the rate of dependent random gathers as a function of the table's footprint: what an array gains by shrinking
// into the 256 MB Infinity Cache or towards the L2s (would a 2-bit text make k_ms_extend's probes cheaper?).
// One byte per gather (the extension probes one text character), every lane its own chain, 16 and 32 wavefronts per CU.
//   hipcc --offload-arch=gfx950 -O2 tools/mall_bench.hip -o tools/mall_bench.bin ; tools/mall_bench.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__global__ void __launch_bounds__(256) k_chase(const uint8_t* __restrict__ tab, uint64_t n, int iters, uint64_t* sink) {
    uint64_t idx = mix(blockIdx.x * 256ull + threadIdx.x + 1) % n;
    uint64_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint64_t v = tab[idx];
        acc += v;
        idx = mix(v + idx + i) % n;
    }
    if (acc == 0x1234567) sink[0] = acc;
}
__global__ void k_fill(uint8_t* tab, uint64_t n) {
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) tab[i] = (uint8_t)mix(i + 7);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    uint64_t* sink;
    CK(hipMalloc(&sink, 8));
    const uint64_t top = 16ull << 30;
    uint8_t* tab;
    CK(hipMalloc(&tab, top));
    k_fill<<<65536, 256>>>(tab, top);
    CK(hipDeviceSynchronize());
    printf("device %s, %d CUs: dependent random 1-byte gathers, G gathers/s\n", prop.gcnArchName, ncu);
    printf("%12s %14s %14s\n", "table", "16 waves/CU", "32 waves/CU");
    const double mbs[] = {2, 8, 11.5, 24, 46, 100, 200, 256, 400, 1024, 4096, 16384};
    for (double mb : mbs) {
        const uint64_t n = (uint64_t)(mb * 1048576.0);
        double rate[2];
        int k = 0;
        for (int bpc : {4, 8}) {
            const int grid = bpc * ncu, iters = 4000;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            k_chase<<<grid, 256>>>(tab, n, 400, sink);  // (warms the caches with this footprint)
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            k_chase<<<grid, 256>>>(tab, n, iters, sink);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            rate[k++] = (double)grid * 256 * iters / ms / 1e6;
        }
        printf("%9.1f MB %14.1f %14.1f\n", mb, rate[0], rate[1]);
        fflush(stdout);
    }
    return 0;
}
