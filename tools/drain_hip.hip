// drain_hip -- can the GPU write an output file's pages itself?  The file's tail is fallocate()d and mapped (MAP_SHARED |
// MAP_POPULATE) ahead of time, the mapping page-locked with hipHostRegister, and device-to-host copies land in the page cache
// directly: no staging buffer, no CPU copy.  Measures what each step costs on the box it runs on (tmpfs by default).
//   hipcc --offload-arch=gfx950 -O2 tools/drain_hip.hip -o /tmp/drain_hip ; /tmp/drain_hip /dev/shm/x 2000 154
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill(char* p, size_t n, int salt) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (char)('0' + (i + salt) % 10);
}

int main(int argc, char** argv) {
    if (argc < 4) { printf("usage: %s <path-prefix> <total MB> <piece MB>\n", argv[0]); return 2; }
    const std::string path = std::string(argv[1]) + ".drain_hip";
    const size_t total = (size_t)atol(argv[2]) << 20, piece = (size_t)atol(argv[3]) << 20, np = total / piece;
    char* d = nullptr;
    CK(hipMalloc((void**)&d, piece));
    fill<<<1024, 256>>>(d, piece, 3);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        ::unlink(path.c_str());
        const int fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        if (::fallocate(fd, 0, 0, (off_t)total) != 0) { perror("fallocate"); return 1; }
        const double t_alloc = now() - t0;
        t0 = now();
        char* m = (char*)::mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
        if (m == MAP_FAILED) { perror("mmap"); return 1; }
        const double t_map = now() - t0;
        t0 = now();
        hipError_t e = hipHostRegister(m, total, hipHostRegisterDefault);
        const double t_reg = now() - t0;
        printf("-- repetition %d: %zu MB: fallocate %.3f s, mmap(MAP_POPULATE) %.3f s, hipHostRegister %.3f s (%s)\n", rep, total >> 20, t_alloc,
               t_map, t_reg, hipGetErrorString(e));
        if (e != hipSuccess) { (void)hipGetLastError(); ::munmap(m, total); ::close(fd); break; }
        t0 = now();
        for (size_t i = 0; i < np; ++i) CK(hipMemcpy(m + i * piece, d, piece, hipMemcpyDeviceToHost));
        double dt = now() - t0;
        printf("   device -> file pages, %zu pieces of %zu MB: %.3f s = %.1f GB/s\n", np, piece >> 20, dt, (double)(np * piece) / dt / 1e9);
        // at an odd offset (a super-batch's text starts wherever the one before it ended)
        t0 = now();
        for (size_t i = 0; i + 1 < np; ++i) CK(hipMemcpy(m + i * piece + 12345, d, piece - 777, hipMemcpyDeviceToHost));
        dt = now() - t0;
        printf("   the same at unaligned offsets and sizes: %.3f s = %.1f GB/s\n", dt, (double)((np - 1) * (piece - 777)) / dt / 1e9);
        t0 = now();
        CK(hipHostUnregister(m));
        const double t_unreg = now() - t0;
        t0 = now();
        ::munmap(m, total);
        const double t_unmap = now() - t0;
        t0 = now();
        if (::ftruncate(fd, (off_t)(total * 3 / 4)) != 0) perror("ftruncate");
        printf("   hipHostUnregister %.3f s, munmap %.3f s, ftruncate to 3/4 %.3f s\n", t_unreg, t_unmap, now() - t0);
        // what pread sees is what the device wrote
        std::vector<char> back(4096);
        if (::pread(fd, back.data(), back.size(), (off_t)(piece + 12345)) != (ssize_t)back.size()) perror("pread");
        bool ok = true;
        for (size_t i = 0; i < back.size(); ++i) ok = ok && back[i] == (char)('0' + (i + 3) % 10);
        printf("   read back through the file: %s\n", ok ? "identical" : "DIFFERENT");
        ::close(fd);
    }
    // for comparison: the same copies into hipHostMalloc memory
    {
        char* h = nullptr;
        double t0 = now();
        CK(hipHostMalloc((void**)&h, piece, hipHostMallocDefault));
        printf("-- hipHostMalloc of one %zu MB piece: %.3f s\n", piece >> 20, now() - t0);
        t0 = now();
        for (size_t i = 0; i < np; ++i) CK(hipMemcpy(h, d, piece, hipMemcpyDeviceToHost));
        const double dt = now() - t0;
        printf("   device -> page-locked buffer: %.3f s = %.1f GB/s\n", dt, (double)(np * piece) / dt / 1e9);
        (void)hipHostFree(h);
    }
    // ---- the other direction (round 5, index start-up): a FILE's pages as the source of host-to-device copies.  Which mappings
    // can be registered (read-only / writable, shared / private), what populating and registering cost, the copy rate.
    {
        const int fdw = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        std::vector<char> blk(64 << 20, 'x');
        for (size_t o = 0; o < total; o += blk.size())
            if (::pwrite(fdw, blk.data(), blk.size(), (off_t)o) != (ssize_t)blk.size()) perror("pwrite");
        ::close(fdw);
        struct Variant { const char* name; int oflag; int prot; int mflag; unsigned reg; } vs[] = {
            {"O_RDONLY PROT_READ MAP_SHARED, hipHostRegisterReadOnly", O_RDONLY, PROT_READ, MAP_SHARED, 0x08},
            {"O_RDONLY PROT_READ MAP_SHARED, hipHostRegisterDefault", O_RDONLY, PROT_READ, MAP_SHARED, 0},
            {"O_RDONLY PROT_READ MAP_PRIVATE, hipHostRegisterReadOnly", O_RDONLY, PROT_READ, MAP_PRIVATE, 0x08},
            {"O_RDWR PROT_READ|WRITE MAP_SHARED, hipHostRegisterDefault", O_RDWR, PROT_READ | PROT_WRITE, MAP_SHARED, 0},
        };
        for (const Variant& v : vs) {
            const int fd = ::open(path.c_str(), v.oflag);
            char* m = (char*)::mmap(nullptr, total, v.prot, v.mflag, fd, 0);
            if (m == MAP_FAILED) { perror("mmap"); ::close(fd); continue; }
            double t0 = now();
            const int pr = ::madvise(m, total, (v.prot & PROT_WRITE) ? 23 /* MADV_POPULATE_WRITE */ : 22 /* MADV_POPULATE_READ */);
            const double t_pop = now() - t0;
            t0 = now();
            hipError_t e = hipHostRegister(m, total, v.reg);
            const double t_reg = now() - t0;
            printf("-- file as the source: %s: populate (one thread) %.3f s (rc %d), hipHostRegister %.3f s (%s)\n", v.name, t_pop, pr, t_reg, hipGetErrorString(e));
            if (e == hipSuccess) {
                t0 = now();
                for (size_t i = 0; i < np; ++i) CK(hipMemcpy(d, m + i * piece, piece, hipMemcpyHostToDevice));
                const double dt = now() - t0;
                printf("   file pages -> device, %zu pieces of %zu MB: %.3f s = %.1f GB/s\n", np, piece >> 20, dt, (double)(np * piece) / dt / 1e9);
                CK(hipHostUnregister(m));
            } else {
                (void)hipGetLastError();
                t0 = now();
                for (size_t i = 0; i < np; ++i) CK(hipMemcpy(d, m + i * piece, piece, hipMemcpyHostToDevice));
                const double dt = now() - t0;
                printf("   unregistered mapping -> device (the runtime stages): %.3f s = %.1f GB/s\n", dt, (double)(np * piece) / dt / 1e9);
            }
            ::munmap(m, total);
            ::close(fd);
        }
    }
    ::unlink(path.c_str());
    return 0;
}
