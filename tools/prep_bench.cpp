// how fast can a tmpfs file's pages be had: fallocate + populate (round 5) against ftruncate + populate on N threads
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const uint64_t size = (uint64_t)(atof(argv[1]) * (1ull << 30));
    const int mode = atoi(argv[2]);  // 0: fallocate + 4 threads, 1: ftruncate + nt threads
    const unsigned nt = argc > 3 ? atoi(argv[3]) : 4;
    const char* path = "/dev/shm/prep_bench.bin";
    unlink(path);
    double t0 = now();
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (mode == 0) { if (fallocate(fd, 0, 0, size) != 0) perror("fallocate"); } else { if (ftruncate(fd, size) != 0) perror("ftruncate"); }
    char* m = (char*)mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    double t1 = now();
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([=] {
            const uint64_t lo = (size / 4096 * t / nt) * 4096, hi = t + 1 == nt ? size : (size / 4096 * (t + 1) / nt) * 4096;
            const uint64_t step = 64ull << 20;
            for (uint64_t a = lo; a < hi; a += step) {
                const uint64_t n = std::min(step, hi - a);
                if (madvise(m + a, n, MADV_POPULATE_WRITE) != 0) { perror("madvise"); return; }
            }
        });
    for (auto& x : th) x.join();
    double t2 = now();
    printf("mode %d threads %u: %.1f GB: allocate+map %.3f s, populate %.3f s, total %.3f s\n", mode, nt, size / 1e9, t1 - t0, t2 - t1, t2 - t0);
    double t3 = now();
    munmap(m, size); close(fd); unlink(path);
    printf("   munmap+unlink %.3f s\n", now() - t3);
}
