// overlap_hip -- do the three things a `spumoni run` worker does to the device overlap when two workers do them at once?
// One worker's super-batch is: a host-to-device copy (32 MB of reads, page-locked buffer), kernels (~1 ms), a device-to-host copy
// (75 MB of text) into page-locked memory -- the output file's mapping (hipHostRegister) or a hipHostMalloc buffer.  Two workers
// on two non-blocking streams, each from a thread of its own, each synchronising its own stream after every step the way the
// C-ABI's host-buffer entry points do (spx_api.hip: hipMemcpyAsync + hipStreamSynchronize on the handle's ctx_stream).
// Prints the time per super-batch of ONE worker alone and of two at once: 2 x alone = the device serialises them,
// ~max(copy-out, copy-in + kernels) = it overlaps them.
//   hipcc --offload-arch=gfx950 -O2 -pthread tools/overlap_hip.hip -o tools/overlap_hip.bin ; tools/overlap_hip.bin /dev/shm/x
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// a kernel that takes about a millisecond whatever the chip is doing otherwise: dependent arithmetic, few blocks
__global__ void spin(unsigned* out, unsigned rounds) {
    unsigned v = threadIdx.x + blockIdx.x;
    for (unsigned i = 0; i < rounds; ++i) v = v * 1664525u + 1013904223u;
    if (v == 12345u) out[0] = v;
}
// ... and one that is bound by memory the way the walk is (random 32-byte reads over a large array)
__global__ void gather(const uint4* a, size_t n, unsigned* out, unsigned steps) {
    size_t p = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 2654435761ull % n;
    unsigned acc = 0;
    for (unsigned i = 0; i < steps; ++i) {
        const uint4 v = a[p];
        acc += v.x;
        p = (p * 6364136223846793005ull + v.y + 1442695040888963407ull) % n;
    }
    if (acc == 12345u) out[0] = acc;
}

struct Worker {
    hipStream_t st;
    char *h_in, *d_in, *d_out, *h_out;
    unsigned* d_flag;
};

int main(int argc, char** argv) {
    const std::string path = std::string(argc > 1 ? argv[1] : "/dev/shm/x") + ".overlap_hip";
    const size_t in_bytes = 32u << 20, out_bytes = 75u << 20;
    const int batches = 40;
    uint4* big = nullptr;
    const size_t big_n = (8ull << 30) / sizeof(uint4);
    CK(hipMalloc((void**)&big, big_n * sizeof(uint4)));
    CK(hipMemset(big, 1, big_n * sizeof(uint4)));
    // the "file": 2 workers x out_bytes of a tmpfs file, mapped, populated, registered
    ::unlink(path.c_str());
    const int fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    const size_t file_bytes = 2 * out_bytes;
    if (fd < 0 || ::fallocate(fd, 0, 0, (off_t)file_bytes) != 0) { perror("file"); return 1; }
    char* m = (char*)::mmap(nullptr, file_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
    if (m == MAP_FAILED) { perror("mmap"); return 1; }
    CK(hipHostRegister(m, file_bytes, hipHostRegisterPortable));
    Worker w[2];
    char* h_out_malloc[2];
    for (int i = 0; i < 2; ++i) {
        CK(hipStreamCreateWithFlags(&w[i].st, hipStreamNonBlocking));
        CK(hipHostMalloc((void**)&w[i].h_in, in_bytes, hipHostMallocDefault));
        CK(hipHostMalloc((void**)&h_out_malloc[i], out_bytes, hipHostMallocDefault));
        CK(hipMalloc((void**)&w[i].d_in, in_bytes));
        CK(hipMalloc((void**)&w[i].d_out, out_bytes));
        CK(hipMalloc((void**)&w[i].d_flag, 64));
        for (size_t k = 0; k < in_bytes; k += 4096) w[i].h_in[k] = 1;
    }
    CK(hipDeviceSynchronize());
    // what = bit 0: copy in, bit 1: kernel, bit 2: copy out; kind: 0 spin kernel, 1 gather kernel
    auto one = [&](Worker& x, int what, int kind, bool sync_each) {
        if (what & 1) {
            CK(hipMemcpyAsync(x.d_in, x.h_in, in_bytes, hipMemcpyHostToDevice, x.st));
            if (sync_each) CK(hipStreamSynchronize(x.st));
        }
        if (what & 2) {
            if (kind == 0)
                spin<<<64, 64, 0, x.st>>>(x.d_flag, 300000);
            else
                gather<<<4096, 256, 0, x.st>>>(big, big_n, x.d_flag, 48);
            if (sync_each) CK(hipStreamSynchronize(x.st));
        }
        if (what & 4) {
            CK(hipMemcpyAsync(x.h_out, x.d_out, out_bytes, hipMemcpyDeviceToHost, x.st));
        }
        CK(hipStreamSynchronize(x.st));
    };
    auto run = [&](int nworkers, int what0, int what1, int kind, bool sync_each) {
        std::atomic<int> go{0};
        std::vector<std::thread> th;
        double t[2] = {0, 0};
        for (int i = 0; i < nworkers; ++i)
            th.emplace_back([&, i] {
                CK(hipSetDevice(0));
                for (int b = 0; b < 3; ++b) one(w[i], i ? what1 : what0, kind, sync_each);
                go.fetch_add(1);
                while (go.load() < nworkers) {}
                const double t0 = now();
                for (int b = 0; b < batches; ++b) one(w[i], i ? what1 : what0, kind, sync_each);
                t[i] = now() - t0;
            });
        for (auto& x : th) x.join();
        return std::max(t[0], t[1]) / batches * 1e3;
    };
    for (int dest = 0; dest < 2; ++dest) {
        for (int i = 0; i < 2; ++i) w[i].h_out = dest ? m + (size_t)i * out_bytes : h_out_malloc[i];
        printf("== copy-out destination: %s\n", dest ? "the registered mapping of a tmpfs file" : "hipHostMalloc memory");
        for (int kind = 0; kind < 2; ++kind) {
            printf("-- kernel: %s\n", kind ? "random gathers over 8 GB (memory-bound)" : "dependent arithmetic on 64 wavefronts (leaves the chip free)");
            const double in1 = run(1, 1, 1, kind, true), k1 = run(1, 2, 2, kind, true), out1 = run(1, 4, 4, kind, true), all1 = run(1, 7, 7, kind, true);
            printf("   one worker alone, ms per super-batch: copy in %.2f, kernel %.2f, copy out %.2f, all three %.2f\n", in1, k1, out1, all1);
            printf("   two workers, each all three:          %.2f ms per PAIR of super-batches (serial: %.2f)\n", run(2, 7, 7, kind, true), 2 * all1);
            printf("   two workers, steps not synchronised one by one (one sync per super-batch): %.2f ms per pair\n", run(2, 7, 7, kind, false));
            printf("   copy in beside copy out:              %.2f ms (serial: %.2f, overlapped: %.2f)\n", run(2, 1, 4, kind, true), in1 + out1, std::max(in1, out1));
            printf("   kernel beside copy out:               %.2f ms (serial: %.2f, overlapped: %.2f)\n", run(2, 2, 4, kind, true), k1 + out1, std::max(k1, out1));
            printf("   kernel beside copy in:                %.2f ms (serial: %.2f, overlapped: %.2f)\n", run(2, 2, 1, kind, true), k1 + in1, std::max(k1, in1));
            printf("   copy out beside copy out:             %.2f ms (serial: %.2f)\n", run(2, 4, 4, kind, true), 2 * out1);
            fflush(stdout);
        }
    }
    CK(hipHostUnregister(m));
    ::munmap(m, file_bytes);
    ::close(fd);
    ::unlink(path.c_str());
    return 0;
}
