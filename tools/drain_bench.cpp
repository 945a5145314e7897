// drain_bench -- how fast can ONE output file in the page cache take a super-batch's text?
//
// `spumoni run` ends in pwrite()s of ~150 MB pieces into <reads>.pseudo_lengths (compute_ms_pml.cpp:1001-1010 writes the
// same bytes through an ofstream).  Round 4 found the run bound by that one stream (5.7 GB/s into tmpfs).  This tool
// measures the ways a file's new tail can be filled, on the box and file system it is run on:
//   pwrite1      one thread, one pwrite per piece                        (round 4's writer)
//   pwriteN      N threads, disjoint ranges of the piece, one file       (serialise on the inode lock?)
//   mmapN        ftruncate + mmap(MAP_SHARED) the piece's range, N threads memcpy their parts (page faults, no lock)
//   mmapN_pop    the same, but every thread first madvise(MADV_POPULATE_WRITE)s its part
//   fallocN      fallocate the range first (allocation serial, in the kernel), then N threads pwrite
//   rewriteN     N threads pwrite over pages that already exist           (the copy alone)
//   filesN       N threads, each its own file                            (what the lock costs)
//   pre+...      the whole file fallocate()d BEFORE the clock starts (what `spumoni run` can do while the index loads), then
//                pwrite1 / pwriteN / mmapN / mmapN_pop over pages that exist
//   pre+mapped   the whole file fallocate()d AND mapped with MAP_POPULATE before the clock starts: the timed part is N threads'
//                memcpy into memory that happens to be the file (no system call, no fault); then ftruncate to the real size
// g++ -O2 -pthread tools/drain_bench.cpp -o tools/drain_bench.bin ; tools/drain_bench.bin /dev/shm/x 2000 154 16
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par(int nt, const std::function<void(int)>& f) {
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(f, t);
    f(0);
    for (auto& x : th) x.join();
}

static void pwrite_all(int fd, const char* p, size_t n, size_t at) {
    while (n) {
        ssize_t w = ::pwrite(fd, p, n, (off_t)at);
        if (w <= 0) { perror("pwrite"); exit(1); }
        p += w; n -= (size_t)w; at += (size_t)w;
    }
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s <path-prefix> <total MB> <piece MB> <threads>\n", argv[0]); return 2; }
    const std::string prefix = argv[1];
    const size_t total = (size_t)atol(argv[2]) << 20, piece = (size_t)atol(argv[3]) << 20;
    const int nt = atoi(argv[4]);
    const size_t npieces = total / piece;
    char* src = (char*)aligned_alloc(4096, piece);
    for (size_t i = 0; i < piece; ++i) src[i] = (char)('0' + i % 10);
    const size_t align = 4096;
    auto part = [&](int t, size_t& lo, size_t& hi) {
        lo = (piece * (size_t)t / (size_t)nt) / align * align;
        hi = t + 1 == nt ? piece : (piece * (size_t)(t + 1) / (size_t)nt) / align * align;
    };
    auto report = [&](const char* name, double s) {
        printf("%-12s %7.3f s  %6.2f GB/s\n", name, s, (double)(npieces * piece) / s / 1e9);
        fflush(stdout);
    };
    const std::string path = prefix + ".drain";
    auto fresh = [&]() {
        ::unlink(path.c_str());
        int fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) { perror("open"); exit(1); }
        return fd;
    };
    for (int rep = 0; rep < 2; ++rep) {
        printf("-- repetition %d: %zu pieces of %zu MB, %d threads, %s\n", rep, npieces, piece >> 20, nt, path.c_str());
        {
            int fd = fresh();
            double t0 = now();
            for (size_t i = 0; i < npieces; ++i) pwrite_all(fd, src, piece, i * piece);
            report("pwrite1", now() - t0);
            // rewrite: the pages exist
            t0 = now();
            for (size_t i = 0; i < npieces; ++i)
                par(nt, [&](int t) { size_t lo, hi; part(t, lo, hi); pwrite_all(fd, src + lo, hi - lo, i * piece + lo); });
            report("rewriteN", now() - t0);
            t0 = now();
            for (size_t i = 0; i < npieces; ++i) pwrite_all(fd, src, piece, i * piece);
            report("rewrite1", now() - t0);
            ::close(fd);
        }
        {
            int fd = fresh();
            double t0 = now();
            for (size_t i = 0; i < npieces; ++i)
                par(nt, [&](int t) { size_t lo, hi; part(t, lo, hi); pwrite_all(fd, src + lo, hi - lo, i * piece + lo); });
            report("pwriteN", now() - t0);
            ::close(fd);
        }
        for (int pop = 0; pop < 2; ++pop) {
            int fd = fresh();
            double t0 = now();
            for (size_t i = 0; i < npieces; ++i) {
                if (::ftruncate(fd, (off_t)((i + 1) * piece)) != 0) { perror("ftruncate"); return 1; }
                char* m = (char*)::mmap(nullptr, piece, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)(i * piece));
                if (m == MAP_FAILED) { perror("mmap"); return 1; }
                par(nt, [&](int t) {
                    size_t lo, hi; part(t, lo, hi);
                    if (pop) ::madvise(m + lo, hi - lo, MADV_POPULATE_WRITE);
                    memcpy(m + lo, src + lo, hi - lo);
                });
                ::munmap(m, piece);
            }
            report(pop ? "mmapN_pop" : "mmapN", now() - t0);
            ::close(fd);
        }
        {
            int fd = fresh();
            double t0 = now(), ta = 0;
            for (size_t i = 0; i < npieces; ++i) {
                double a0 = now();
                if (::fallocate(fd, 0, (off_t)(i * piece), (off_t)piece) != 0) { perror("fallocate"); break; }
                ta += now() - a0;
                par(nt, [&](int t) { size_t lo, hi; part(t, lo, hi); pwrite_all(fd, src + lo, hi - lo, i * piece + lo); });
            }
            report("fallocN", now() - t0);
            printf("             (fallocate alone %.3f s)\n", ta);
            ::close(fd);
        }
        for (int how = 0; how < 4; ++how) {
            int fd = fresh();
            double a0 = now();
            if (::fallocate(fd, 0, 0, (off_t)(npieces * piece)) != 0) { perror("fallocate"); ::close(fd); break; }
            const double ta = now() - a0;
            double t0 = now();
            for (size_t i = 0; i < npieces; ++i) {
                if (how == 0) {
                    pwrite_all(fd, src, piece, i * piece);
                } else if (how == 1) {
                    par(nt, [&](int t) { size_t lo, hi; part(t, lo, hi); pwrite_all(fd, src + lo, hi - lo, i * piece + lo); });
                } else {
                    char* m = (char*)::mmap(nullptr, piece, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)(i * piece));
                    if (m == MAP_FAILED) { perror("mmap"); return 1; }
                    par(nt, [&](int t) {
                        size_t lo, hi; part(t, lo, hi);
                        if (how == 3) ::madvise(m + lo, hi - lo, MADV_POPULATE_WRITE);
                        memcpy(m + lo, src + lo, hi - lo);
                    });
                    ::munmap(m, piece);
                }
            }
            static const char* const nm[4] = {"pre+pwrite1", "pre+pwriteN", "pre+mmapN", "pre+mmapNpop"};
            report(nm[how], now() - t0);
            if (how == 0) printf("             (fallocate of the whole file, before the clock: %.3f s)\n", ta);
            // what giving back an over-estimate costs: cut a quarter off
            if (how == 0) {
                double c0 = now();
                if (::ftruncate(fd, (off_t)(npieces * piece * 3 / 4)) != 0) perror("ftruncate");
                printf("             (ftruncate to 3/4: %.3f s)\n", now() - c0);
            }
            ::close(fd);
        }
        for (int tn = 1; tn <= nt; tn *= 2) {
            int fd = fresh();
            double a0 = now();
            if (::fallocate(fd, 0, 0, (off_t)(npieces * piece)) != 0) { perror("fallocate"); ::close(fd); break; }
            char* m = (char*)::mmap(nullptr, npieces * piece, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
            if (m == MAP_FAILED) { perror("mmap"); return 1; }
            const double ta = now() - a0;
            double t0 = now();
            for (size_t i = 0; i < npieces; ++i)
                par(tn, [&](int t) {
                    size_t lo = (piece * (size_t)t / (size_t)tn) / align * align;
                    size_t hi = t + 1 == tn ? piece : (piece * (size_t)(t + 1) / (size_t)tn) / align * align;
                    memcpy(m + i * piece + lo, src + lo, hi - lo);
                });
            char nm[40];
            snprintf(nm, sizeof nm, "pre+mapped%d", tn);
            report(nm, now() - t0);
            double c0 = now();
            ::munmap(m, npieces * piece);
            const double tu = now() - c0;
            if (tn == 1) printf("             (fallocate + mmap(MAP_POPULATE) of the whole file, before the clock: %.3f s; munmap after it: %.3f s)\n", ta, tu);
            ::close(fd);
        }
        {
            std::vector<int> fds((size_t)nt);
            for (int t = 0; t < nt; ++t) {
                std::string p = path + "." + std::to_string(t);
                ::unlink(p.c_str());
                fds[(size_t)t] = ::open(p.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
            }
            double t0 = now();
            for (size_t i = 0; i < npieces; ++i)
                par(nt, [&](int t) {
                    size_t lo, hi; part(t, lo, hi);
                    pwrite_all(fds[(size_t)t], src + lo, hi - lo, i * (hi - lo));
                });
            report("filesN", now() - t0);
            for (int t = 0; t < nt; ++t) {
                ::close(fds[(size_t)t]);
                ::unlink((path + "." + std::to_string(t)).c_str());
            }
        }
    }
    ::unlink(path.c_str());
    return 0;
}
