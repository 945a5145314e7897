#include "reads.hpp"

#include <dirent.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <mutex>

#include <algorithm>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <thread>

namespace spumoni_host {

// The reference's FATAL_ERROR / FATAL_WARNING end in std::exit(1) on its one thread.  Here a fatal error is raised on
// whichever thread meets it (the report thread, for a read that is empty after digestion) while the parser and the
// device workers are still at work: std::exit would run the static destructors -- the page-locked pool, the library's
// own -- under their feet (found as a heap-use-after-free by the CPU fuzz, tools/cli_fuzz_cpu.py).  So: flush what is
// buffered (the output files are written with pwrite and need nothing) and leave without them.
// outputs of an earlier run that were moved aside to be removed in the background (classify.cpp, OutFile::open): whoever
// ends the process first makes sure none of them stays behind
// (round 6: a fixed table of C strings, so that a signal handler can walk it too -- SIGINT / SIGTERM / SIGHUP used to leave
// the prepared multi-GB files behind, on tmpfs that is RAM: ADVICE r5)
static std::mutex g_leftover_mu;
static constexpr int MAX_LEFTOVERS = 64;
static char* g_leftovers[MAX_LEFTOVERS];
static std::atomic<int> g_nleftovers{0};
extern "C" void remove_leftovers_on_signal(int sig) {
    const int n = g_nleftovers.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i)
        if (g_leftovers[i]) (void)::unlink(g_leftovers[i]);  // (async-signal-safe)
    ::_exit(128 + sig);
}
void register_leftover(const std::string& path) {
    std::lock_guard<std::mutex> g(g_leftover_mu);
    const int n = g_nleftovers.load(std::memory_order_relaxed);
    if (n >= MAX_LEFTOVERS) return;
    g_leftovers[n] = ::strdup(path.c_str());
    g_nleftovers.store(n + 1, std::memory_order_release);
    static bool handlers = false;
    if (!handlers) {  // the first file that must not stay behind: from now on an interrupted run takes them along
        handlers = true;
        struct sigaction sa;
        std::memset(&sa, 0, sizeof sa);
        sa.sa_handler = remove_leftovers_on_signal;
        sigemptyset(&sa.sa_mask);
        for (int sig : {SIGINT, SIGTERM, SIGHUP}) {
            struct sigaction old;
            if (::sigaction(sig, nullptr, &old) == 0 && old.sa_handler == SIG_DFL) (void)::sigaction(sig, &sa, nullptr);
        }
    }
}
void remove_leftovers() {
    std::lock_guard<std::mutex> g(g_leftover_mu);
    const int n = g_nleftovers.load(std::memory_order_relaxed);
    for (int i = 0; i < n; ++i)
        if (g_leftovers[i]) (void)::unlink(g_leftovers[i]);  // (already gone: fine)
}

// `<final>.partial.<pid>` / `<final>.old.<pid>` of a run that was killed outright (SIGKILL, out of memory): nothing of it could
// clean up.  The next run over the same pattern file does, for every such name whose process is gone.
void remove_stale_leftovers(const std::string& final_path) {
    const size_t slash = final_path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? "." : final_path.substr(0, slash);
    const std::string base = slash == std::string::npos ? final_path : final_path.substr(slash + 1);
    DIR* d = ::opendir(dir.c_str());
    if (!d) return;
    while (struct dirent* e = ::readdir(d)) {
        const std::string name = e->d_name;
        for (const char* mark : {".partial.", ".old."}) {
            const std::string pre = base + mark;
            if (name.size() <= pre.size() || name.compare(0, pre.size(), pre) != 0) continue;
            char* endp = nullptr;
            const long pid = std::strtol(name.c_str() + pre.size(), &endp, 10);
            if (pid <= 0 || *endp != 0 || pid == (long)::getpid()) continue;
            if (::kill((pid_t)pid, 0) != 0 && errno == ESRCH) (void)::unlink((dir + "/" + name).c_str());
        }
    }
    ::closedir(d);
}

static void (*g_exit_hook)() = nullptr;
void set_exit_hook(void (*hook)()) { g_exit_hook = hook; }

[[noreturn]] static void leave(int code) {
    // (the hook cuts the output files under the device's and the pool's feet -- classify.cpp, settle_outputs -- so it comes
    // last and nothing but _Exit follows it: ADVICE r5)
    std::cout.flush();
    std::fflush(nullptr);
    remove_leftovers();
    if (g_exit_hook) g_exit_hook();
    std::_Exit(code);
}

void fatal_error(const char* fmt, ...) {  // FATAL_ERROR, include/spumoni_main.hpp:32-33
    std::fprintf(stderr, "\n\033[31mError: \033[0m");
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
    std::fprintf(stderr, "\n\n");
    leave(1);
}

void fatal_warning(const char* fmt, ...) {  // FATAL_WARNING, include/spumoni_main.hpp:28-29
    std::fprintf(stderr, "Warning: ");
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
    std::fprintf(stderr, "\n\n");
    leave(1);
}

ReadFile::ReadFile(const std::string& path, unsigned threads) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + path);
    struct stat st;
    if (::fstat(fd, &st) != 0) {
        ::close(fd);
        throw std::runtime_error("cannot stat " + path);
    }
    size_ = (size_t)st.st_size;
    if (size_ > 0) {
        void* p = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        if (p == MAP_FAILED) {
            ::close(fd);
            throw std::runtime_error("cannot map " + path);
        }
        data_ = (const char*)p;
    }
    ::close(fd);
    // line table (std::getline semantics: a trailing '\n' does not start another line): every thread
    // finds the newlines of its part of the file, the parts are then laid end to end
    const unsigned nt = std::max(1u, std::min<unsigned>(threads, (unsigned)(size_ / (8u << 20)) + 1));
    std::vector<std::vector<size_t>> part(nt);
    auto scan = [&](unsigned t) {
        const size_t lo = size_ * t / nt, hi = size_ * (t + 1) / nt;
        std::vector<size_t>& v = part[t];
        v.reserve((hi - lo) / 64 + 16);
        const char* p = data_ + lo;
        const char* const e = data_ + hi;
        while (p < e) {
            const char* q = (const char*)std::memchr(p, '\n', (size_t)(e - p));
            if (!q) break;
            v.push_back((size_t)(q - data_) + 1);  // the next line starts behind the newline
            p = q + 1;
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(scan, t);
    scan(0);
    for (auto& x : th) x.join();
    size_t total = 1;
    for (auto& v : part) total += v.size();
    line_start_.reserve(total + 1);
    line_start_.push_back(0);
    for (auto& v : part) line_start_.insert(line_start_.end(), v.begin(), v.end());
    ends_with_newline_ = size_ > 0 && data_[size_ - 1] == '\n';
    // a file that does not end in a newline has one more line; either way the last entry is one past
    // the (virtual) newline that ends the last line
    if (size_ > 0 && !ends_with_newline_) line_start_.push_back(size_ + 1);
    if (size_ == 0) eof_ = true;
}

ReadFile::~ReadFile() {
    if (data_) ::munmap((void*)data_, size_);
}

bool ReadFile::next_batch(size_t num_bases, std::vector<ParsedRead>& out) {
    out.clear();
    Range r;
    if (!next_range(num_bases, r)) return false;
    ParseError err;
    parse_range(r, out, err);
    if (err.fatal) fatal_error("%s", err.message.c_str());
    return true;
}

void ReadFile::precompute_ranges(size_t num_bases) {
    ranges_.clear();
    Range r;
    while (next_range_scan(num_bases, r)) ranges_.push_back(r);
    ranges_bases_ = num_bases;
    ranges_next_ = 0;
    ranges_ready_ = true;
}

bool ReadFile::next_range(size_t num_bases, Range& out) {
    if (ranges_ready_ && num_bases == ranges_bases_) {
        if (ranges_next_ >= ranges_.size()) return false;
        out = ranges_[ranges_next_++];
        return true;
    }
    return next_range_scan(num_bases, out);
}

bool ReadFile::next_range_scan(size_t num_bases, Range& out) {
    // input type sniffing (batch_loader.cpp:30-38)
    if (format_ == ReadFormat::NotClear) {
        if (size_ == 0) return false;
        switch (data_[0]) {
            case '>': format_ = ReadFormat::Fasta; break;
            case '@': format_ = ReadFormat::Fastq; break;
            default: fatal_error("unrecognized input query file type - expects FASTA or FASTQ.");
        }
    }
    // batch_loader.cpp:49-73 replayed over the line table
    size_t covered = 0, nlines = 0, record = 0;
    bool valid = false;
    const size_t first = next_line_;
    while (!eof_ && covered < num_bases) {
        if (next_line_ >= line_count()) {
            // getline() fails: nothing left.  loadBatch returns false and the lines already
            // taken for this batch are dropped (FASTQ tail quirk, Appendix C16)
            eof_ = true;
            return false;
        }
        const size_t len = line_len(next_line_);
        const bool last_line = next_line_ + 1 == line_count();
        next_line_++;
        nlines++;
        record += len;
        valid = true;
        if (last_line && !ends_with_newline_) eof_ = true;  // getline hit EOF while reading the line
        if (format_ == ReadFormat::Fastq) {
            if (nlines % 4 == 0) {
                covered += record / 2;
                record = 0;
            }
        } else {
            // input.peek(): at end of data it returns EOF and the stream stops being good()
            if (last_line) {
                eof_ = true;
            } else {
                const std::string_view nxt = line(next_line_);  // peek = first char of the next line
                if (!nxt.empty() && nxt[0] == '>') {
                    covered += record;
                    record = 0;
                }
            }
        }
    }
    if (!valid) return false;
    out.first = first;
    out.last = next_line_;
    // characters in those lines: the span minus one newline per line (the last line of the file may lack its own)
    out.bytes = line_start_[next_line_] - line_start_[first] - (next_line_ - first);
    return true;
}

static inline void strip_trailing_space(std::string_view& s) {
    while (!s.empty() && std::isspace((unsigned char)s.back())) s.remove_suffix(1);
}

// grabNextRead (batch_loader.cpp:78-131) over the lines of one batch
void ReadFile::scan_range(const Range& r, std::vector<ReadRec>& out, ParseError& err) const {
    const size_t last = r.last;
    size_t i = r.first;
    auto fail = [&](const std::string& msg) {
        err.fatal = true;
        err.message = msg;
    };
    const bool fastq = format_ == ReadFormat::Fastq;
    while (i < last) {
        const std::string_view hdr = line(i++);
        if (hdr.empty()) return;  // an empty header line ends the batch (Appendix C15)
        if (fastq) {
            if (hdr[0] != '@')
                return fail(std::string("Incorrect FASTQ entry, it should start with '@' but found ") + hdr[0]);
        } else if (hdr[0] != '>') {
            return fail(std::string("Incorrect FASTA entry, it should start with '>' but found ") + hdr[0]);
        }
        if (hdr.size() <= 2) return fail("header line is missing an id. invalid query cannot be processed.");
        size_t ws = 1;  // first of " \t\r" at or after position 1
        while (ws < hdr.size() && hdr[ws] != ' ' && hdr[ws] != '\t' && hdr[ws] != '\r') ++ws;
        ReadRec rd;
        rd.id = hdr.data() + 1;
        // header.substr(1, ws): count = ws, i.e. the whitespace character itself is kept (C6), clipped to the line
        rd.id_len = (uint32_t)std::min(ws, hdr.size() - 1);
        if (fastq) {
            if (i >= last) return;
            std::string_view s = line(i);
            rd.first_line = i++;
            strip_trailing_space(s);
            rd.nlines = 1;
            rd.seq_len = s.size();
            if (i >= last) return;  // '+' line
            i++;
            if (i >= last) return;  // qualities
            i++;
        } else {
            rd.first_line = i;
            for (;;) {
                if (i >= last) {
                    // the reference peeks past the end, getline fails and it returns seq.size():
                    // a record with an empty sequence at the end of a batch is dropped
                    if (rd.seq_len == 0) return;
                    break;
                }
                std::string_view s = line(i);
                if (!s.empty() && s[0] == '>') break;
                i++;
                strip_trailing_space(s);
                rd.seq_len += s.size();
                rd.nlines++;
            }
        }
        out.push_back(rd);
    }
}

void ReadFile::copy_seq_upper(const ReadRec& rd, uint8_t* dst) const {
    for (uint32_t l = 0; l < rd.nlines; ++l) {
        std::string_view s = line(rd.first_line + l);
        strip_trailing_space(s);
        const unsigned char* src = (const unsigned char*)s.data();
        const size_t len = s.size();
        for (size_t i = 0; i < len; ++i) {
            const unsigned char ch = src[i];
            dst[i] = (uint8_t)((ch >= 'a' && ch <= 'z') ? ch - 32 : ch);
        }
        dst += len;
    }
}

// the same with the reads as objects (dump-reads, next_batch): views for single-line records, a joined copy otherwise
void ReadFile::parse_range(const Range& r, std::vector<ParsedRead>& out, ParseError& err) const {
    std::vector<ReadRec> recs;
    scan_range(r, recs, err);
    for (const ReadRec& rc : recs) {
        ParsedRead rd;
        rd.id = std::string_view(rc.id, rc.id_len);
        if (rc.nlines <= 1) {
            std::string_view s = rc.nlines ? line(rc.first_line) : std::string_view();
            strip_trailing_space(s);
            rd.one = s;
        } else {
            rd.multi = true;
            for (uint32_t l = 0; l < rc.nlines; ++l) {
                std::string_view s = line(rc.first_line + l);
                strip_trailing_space(s);
                rd.joined.append(s);
            }
        }
        out.push_back(std::move(rd));
    }
}

}  // namespace spumoni_host
