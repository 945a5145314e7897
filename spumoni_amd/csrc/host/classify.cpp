#include "classify.hpp"

#include <sys/stat.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include <fcntl.h>
#include <unistd.h>

#include "index_files.hpp"
#include "reads.hpp"

namespace spumoni_host {

IndexSet::~IndexSet() {
    for (spx_index* p : ix) spx_index_free(p);
}

static bool file_exists(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return f.good();
}

// What the flat index is built from, as a short string: size and modification time of every file `run` would read
// for this mode (raw run files and / or the serialised index, the document array, the text), folded into one
// 64-bit FNV-1a value.  pml_t / ms_t always deserialise the CURRENT files (compute_ms_pml.cpp:700-721, 755-786):
// a cache whose tag differs was written for other files and is not used.
static std::string source_fingerprint(const RunOptions& o) {
    std::vector<std::string> names = {".bwt.heads", ".bwt.len", ".thr_pos", o.ms ? ".thrbv.ms" : ".thrbv.spumoni"};
    if (o.ms) {
        names.push_back(".ssa");
        names.push_back(".esa");
    }
    if (o.use_doc) names.push_back(".doc");
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) {
        for (size_t i = 0; i < n; ++i) h = (h ^ ((const unsigned char*)p)[i]) * 1099511628211ull;
    };
    int found = 0;
    auto add = [&](const std::string& path, const std::string& label) {
        struct stat st;
        if (stat(path.c_str(), &st) != 0) return;
        ++found;
        const uint64_t v[3] = {(uint64_t)st.st_size, (uint64_t)st.st_mtim.tv_sec, (uint64_t)st.st_mtim.tv_nsec};
        mix(label.data(), label.size());
        mix(v, sizeof v);
    };
    for (const std::string& nm : names) add(o.ref_file + nm, nm);
    if (o.ms && !o.text_file.empty()) add(o.text_file, "text");
    char buf[96];
    std::snprintf(buf, sizeof buf, "%s%s:%d files:%016llx", o.ms ? "ms" : "pml", o.use_doc ? "+doc" : "", found,
                  (unsigned long long)h);
    return buf;
}

// One flatten (or one read of the flat-layout cache), then device-to-device copies: the reference
// deserialises its index once per run (pml_t::pml_t / ms_t::ms_t, compute_ms_pml.cpp:700-721, 755-786);
// N devices cost one load plus N-1 peer copies, not N loads.
//   SPUMONI_CACHE=use (default)  read <ref>.{pml,ms}[.doc].spx when it is there
//   SPUMONI_CACHE=write          also write it after flattening the raw / serialised files
//   SPUMONI_CACHE=off            never touch it
void IndexSet::load(const RunOptions& o) {
    const char* pol = std::getenv("SPUMONI_CACHE");
    const std::string policy = pol ? pol : "use";
    const std::string cache = o.ref_file + (o.ms ? ".ms" : ".pml") + (o.use_doc ? ".doc" : "") + ".spx";
    const int dev0 = o.devices.empty() ? 0 : o.devices[0];
    spx_index* first = nullptr;
    const std::string tag = source_fingerprint(o);
    if (policy != "off" && file_exists(cache)) {
        first = spx_index_load_flat(cache.c_str(), dev0);
        if (!first) {
            std::fprintf(stderr, "\n[spumoni-gpu] %s not used (%s): flattening the index files instead\n", cache.c_str(),
                         spx_last_error());
        } else if (tag != spx_index_source_tag(first)) {
            // written for other index files (the index was rebuilt under the same prefix, or another text was given)
            std::fprintf(stderr, "\n[spumoni-gpu] %s is stale (it was written for '%s', the index files are now '%s'): "
                                 "flattening the index files instead\n", cache.c_str(), spx_index_source_tag(first), tag.c_str());
            spx_index_free(first);
            first = nullptr;
        } else {
            from_cache = true;
        }
    }
    if (!first) {
        RawIndex raw;
        std::string err;
        if (!load_raw_index(o.ref_file, o.ms, raw, err)) {
            // no raw run files: fall back to the serialised index the reference's `run` loads
            std::string err2;
            RawIndex ser;
            if (!load_serialized_index(o.ref_file + (o.ms ? ".thrbv.ms" : ".thrbv.spumoni"), o.ms, ser, err2))
                fatal_error("%s\n       and %s", err.c_str(), err2.c_str());
            raw = std::move(ser);
        }
        if (o.use_doc && !load_doc_array(o.ref_file + ".doc", raw, err)) fatal_error("%s", err.c_str());
        std::vector<uint8_t> text;
        if (o.ms && !o.text_file.empty() && !read_whole_file(o.text_file, text))
            fatal_error("cannot read the text file %s (SPUMONI_TEXT)", o.text_file.c_str());
        first = spx_index_from_runs(raw.heads.data(), raw.lens.data(), raw.thr.data(), raw.heads.size(),
                                    o.ms ? raw.ssa.data() : nullptr, o.ms ? raw.esa.data() : nullptr,
                                    o.use_doc ? raw.doc_start.data() : nullptr,
                                    o.use_doc ? raw.doc_end.data() : nullptr, 0, dev0);
        if (!first) fatal_error("%s", spx_last_error());
        // the text is checked against the index (length, and text[samples_start[k]] == head of run k):
        // a text that is not the indexed one is refused instead of giving wrong .lengths
        // ms_t reads the text through the SLP (<ref>.slp, :769-774); here it is plain text in GPU memory: from
        // SPUMONI_TEXT when given (checked against the index), otherwise rebuilt from the index itself
        if (o.ms && !o.text_file.empty()) {
            if (spx_index_set_text(first, text.data(), text.size(), 0) != SPX_OK)
                fatal_error("%s (SPUMONI_TEXT must be the exact text the index was built from)", spx_last_error());
        } else if (o.ms) {
            if (spx_index_rebuild_text(first) != SPX_OK) fatal_error("%s", spx_last_error());
        }
        (void)spx_index_set_source_tag(first, tag.c_str());
        if (policy == "write" && spx_index_save(first, cache.c_str()) != SPX_OK)
            std::fprintf(stderr, "\n[spumoni-gpu] could not write %s: %s\n", cache.c_str(), spx_last_error());
    }
    ix.push_back(first);
    if (spx_index_stats(first, &n, &r) != SPX_OK) fatal_error("%s", spx_last_error());
    // Replicas: flatten once, copy N - 1 times -- as a doubling tree (device 0 -> 1; 0 -> 2, 1 -> 3; 0 -> 4 ... 3 -> 7), the
    // copies of a round on threads of their own: every copy has its own source and, on an xGMI node, its own link, so
    // seven replicas of a 200 GB index cost three copy times instead of seven
    const size_t ndev = o.devices.size();
    const auto t_all = std::chrono::steady_clock::now();
    ix.resize(std::max<size_t>(ndev, 1), nullptr);
    for (size_t have = 1; have < ndev;) {
        const size_t nnew = std::min(have, ndev - have);
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::string> errs(nnew);
        std::vector<std::thread> th;
        for (size_t i = 0; i < nnew; ++i)
            th.emplace_back([&, i] {
                spx_index* p = spx_index_clone(ix[i], o.devices[have + i]);
                if (!p) errs[i] = spx_last_error();
                ix[have + i] = p;
            });
        for (auto& t : th) t.join();
        for (size_t i = 0; i < nnew; ++i) {
            if (!ix[have + i]) fatal_error("%s", errs[i].c_str());
            std::fprintf(stderr, "[timing] index replica on device %d (copied from device %d)\n", o.devices[have + i], o.devices[i]);
        }
        std::fprintf(stderr, "[timing] %zu replica%s in parallel  %.3f s\n", nnew, nnew > 1 ? "s" : "",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        have += nnew;
    }
    if (ndev > 1)
        std::fprintf(stderr, "[timing] all %zu index replicas  %.3f s\n", ndev - 1,
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count());
}

size_t max_value_threshold(double percentile_value, bool is_pml, bool use_promotions, bool use_dna_letters) {
    size_t max_value_thr = (size_t)std::max(percentile_value, 3.0);
    if (use_dna_letters)
        max_value_thr++;
    else if (is_pml && !use_dna_letters && !use_promotions)
        max_value_thr += 4;
    return max_value_thr;
}

namespace {

// "<value> " for every value, the way std::ostream_iterator<size_t>(file, " ") writes them
// (compute_ms_pml.cpp:1003-1010, 1190-1195).  At GPU speed, turning numbers into text IS the job
// (4 * 10^6 reads x 200 values = 2 GB of text): digits come two at a time from a table and go
// straight into a raw buffer whose worst case was reserved per read.
static const char DIGIT_PAIRS[] =
    "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
    "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";

struct TextBuf {
    std::vector<char> buf;
    size_t len = 0;
    char* reserve(size_t more) {
        if (len + more > buf.size()) buf.resize(std::max(buf.size() * 2, len + more + (1u << 20)));
        return buf.data() + len;
    }
    void header(std::string_view id) {
        char* p = reserve(id.size() + 2);
        *p++ = '>';
        std::memcpy(p, id.data(), id.size());
        p[id.size()] = '\n';
        len += id.size() + 2;
    }
    static inline char* put(char* p, uint64_t v) {  // decimal digits of v and a blank
        if (v < 10) {
            *p++ = (char)('0' + v);
        } else if (v < 100) {
            std::memcpy(p, DIGIT_PAIRS + 2 * v, 2);
            p += 2;
        } else if (v < 10000) {
            const uint32_t hi = (uint32_t)v / 100, lo = (uint32_t)v % 100;
            if (hi >= 10) {
                std::memcpy(p, DIGIT_PAIRS + 2 * hi, 2);
                p += 2;
            } else {
                *p++ = (char)('0' + hi);
            }
            std::memcpy(p, DIGIT_PAIRS + 2 * lo, 2);
            p += 2;
        } else {
            char tmp[24];
            int q = 24;
            while (v >= 100) {
                q -= 2;
                std::memcpy(tmp + q, DIGIT_PAIRS + 2 * (v % 100), 2);
                v /= 100;
            }
            if (v >= 10) {
                q -= 2;
                std::memcpy(tmp + q, DIGIT_PAIRS + 2 * v, 2);
            } else {
                tmp[--q] = (char)('0' + v);
            }
            std::memcpy(p, tmp + q, 24 - q);
            p += 24 - q;
        }
        *p++ = ' ';
        return p;
    }
    template <class T>
    void values(const T* v, uint64_t count) {  // "<v0> <v1> ... \n"
        char* p0 = reserve(count * (sizeof(T) > 4 ? 21 : 11) + 1);
        char* p = p0;
        for (uint64_t i = 0; i < count; ++i) p = put(p, v[i]);
        *p++ = '\n';
        len += (size_t)(p - p0);
    }
    void clear() { len = 0; }
};

// Page-locked blocks made ahead of time.  Locking pages is slow (~4 GB/s: a super-batch's 64 MB of reads and
// 160 MB of text cost 50 ms per slot the first time a slot is used, a quarter of a second in a 1-second run),
// so the blocks the slots will want are allocated on a helper thread while the index loads (prepare_pinned_pool)
// and PinnedBuf takes them from here.
struct PinnedPool {
    std::mutex mu;
    std::vector<std::pair<void*, size_t>> blocks;
    void* take(size_t want, size_t& got) {
        std::lock_guard<std::mutex> g(mu);
        size_t best = blocks.size();
        for (size_t i = 0; i < blocks.size(); ++i)
            if (blocks[i].second >= want && (best == blocks.size() || blocks[i].second < blocks[best].second)) best = i;
        if (best == blocks.size()) return nullptr;
        void* p = blocks[best].first;
        got = blocks[best].second;
        blocks.erase(blocks.begin() + (long)best);
        return p;
    }
    void put(void* p, size_t bytes) {
        std::lock_guard<std::mutex> g(mu);
        blocks.emplace_back(p, bytes);
    }
    // (no destructor: blocks still here when the process ends stay page-locked until the system takes them back --
    // unlocking them one by one at exit is the cost the slots avoid as well)
};
PinnedPool& g_pinned_pool = *new PinnedPool;

// grow-only buffer in page-locked host memory (spx_host_alloc): copies to and from the GPU
// then run at PCIe DMA speed (171 vs 42 M reads/s through spx_query_batch on the bench shape)
template <class T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { spx_host_free(p); }
    void reserve(size_t want) {
        if (want <= cap) return;
        size_t nc = std::max(want, cap * 2);
        size_t got = 0;
        T* q = (T*)g_pinned_pool.take(nc * sizeof(T), got);
        if (q)
            nc = got / sizeof(T);
        else
            q = (T*)spx_host_alloc(nc * sizeof(T));
        if (!q) fatal_error("%s", spx_last_error());
        if (n) std::memcpy(q, p, n * sizeof(T));
        spx_host_free(p);
        p = q;
        cap = nc;
    }
    void assign(size_t count, T v) {
        reserve(count);
        n = count;
        for (size_t i = 0; i < count; ++i) p[i] = v;
    }
    void resize_uninit(size_t count) {
        reserve(count);
        n = count;
    }
    void append(const T* src, size_t count) {
        if (count == 0) return;  // (an empty read of a general-text file: nothing to copy, and p may still be null)
        reserve(n + count);
        std::memcpy(p + n, src, count * sizeof(T));
        n += count;
    }
    void push_back(T v) {
        reserve(n + 1);
        p[n++] = v;
    }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    T& back() { return p[n - 1]; }
    const T& back() const { return p[n - 1]; }
};

struct SuperBatch {
    std::vector<std::string_view> ids;  // views into the mapped reads file (or into own_ids)
    std::deque<std::string> own_ids;    // general-text reads are named here (a deque: addresses stay put)
    PinnedBuf<uint8_t> seqs;
    std::vector<uint64_t> offs{0};
    void clear() {
        ids.clear();
        own_ids.clear();
        seqs.n = 0;
        offs.assign(1, 0);
    }
    size_t nreads() const { return ids.size(); }
};

struct Results {
    PinnedBuf<uint32_t> lengths, docs;
    // reads shorter than 65536 characters: lengths and document ids come back as 16-bit values (half the
    // bytes over PCIe, half the store instructions in the walk); `narrow` says which pair of buffers holds them
    PinnedBuf<uint16_t> lengths16, docs16;
    bool narrow = false;
    PinnedBuf<uint64_t> pointers;
    PinnedBuf<spx_class> cls;
    // results of read q are entries [beg[q], end[q]) of the arrays above: the read's own
    // offsets, or -- with -m / -a -- the offsets of the digested read
    std::vector<uint64_t> beg, end;
    // device_text: the vectors came back as the text of the output files (spx_query_text_begin / _fetch): stream i
    // (0 lengths, 1 pointers, 2 document ids) holds, for read q, gap bytes for its ">id\n" line and its values line
    // at [line_start[i][q], line_start[i][q + 1]); beg / end then only say whether a read has values at all
    bool device_text = false;
    uint32_t streams = 0;  // SPX_TEXT_* present in text[]
    PinnedBuf<char> text[3];
    PinnedBuf<uint64_t> line_start[3];
    std::vector<uint32_t> gap;
};

// One super-batch on one device (the worker thread of that device calls this): the batch form of the
// reference's loop body -- [digestion +] matching_statistics + bin classification
// (compute_ms_pml.cpp:916-995).
void run_on_device(spx_index* ix, const RunOptions& o, const SuperBatch& sb, size_t max_value_thr, Results& res) {
    const size_t nreads = sb.nreads();
    const uint64_t total = sb.offs.back();
    const bool digest = o.use_promotions || o.use_dna_letters;
    const int kind = o.use_promotions ? SPX_DIGEST_PROMOTED : SPX_DIGEST_DNA;
    // with digestion the results are laid out at the digested offsets, inside a region as large as
    // the worst case (every k-mer reported: 1 byte each for -m, k letters for -a)
    const uint64_t grow = digest && o.use_dna_letters ? (uint64_t)o.k : 1;
    uint64_t longest = 0;
    for (size_t q = 0; q < nreads; ++q) longest = std::max<uint64_t>(longest, sb.offs[q + 1] - sb.offs[q]);
    res.narrow = !digest && longest < 65536;
    if (res.narrow) {
        res.lengths16.resize_uninit(total + 8);
        if (o.use_doc) res.docs16.resize_uninit(total + 8);
    } else {
        res.lengths.resize_uninit(total * grow);
        if (o.use_doc) res.docs.resize_uninit(total * grow);
    }
    if (o.ms) res.pointers.resize_uninit(total * grow);
    if (o.write_report) res.cls.resize_uninit(nreads);
    res.beg.resize(nreads);
    res.end.resize(nreads);
    int rc;
    // SPUMONI_HOST_FORMAT=1: values over PCIe, digits on the host cores (the round-2 path; A/B runs and tests)
    static const bool host_format = std::getenv("SPUMONI_HOST_FORMAT") != nullptr;
    const bool no_len_text = o.report_only && !o.ms && o.write_report;
    const uint32_t streams = (no_len_text ? 0u : SPX_TEXT_LENGTHS) | (o.ms ? SPX_TEXT_POINTERS : 0u) | (o.use_doc ? SPX_TEXT_DOCS : 0u);
    res.device_text = !host_format && streams != 0;
    res.streams = res.device_text ? streams : 0;
    if (res.device_text) {
        // the output files' text is written on the device (compute_ms_pml.cpp:1001-1010, 1182-1205): what comes back
        // over PCIe is the files' new tail, with room for every ">id\n"
        res.gap.resize(nreads);
        for (size_t q = 0; q < nreads; ++q) res.gap[q] = (uint32_t)sb.ids[q].size() + 2;
        uint64_t bytes[3] = {0, 0, 0};
        rc = spx_query_text_begin(ix, o.ms ? SPX_MODE_MS : SPX_MODE_PML, digest ? kind : 0, (uint32_t)o.k, (uint32_t)o.w,
                                  sb.seqs.data(), sb.offs.data(), nreads, res.gap.data(), streams,
                                  o.write_report ? res.cls.data() : nullptr, o.bin_size, max_value_thr, bytes);
        if (rc != SPX_OK) fatal_error("%s", spx_last_error());
        char* tp[3] = {nullptr, nullptr, nullptr};
        uint64_t* lp[3] = {nullptr, nullptr, nullptr};
        int first = -1;
        for (int i = 0; i < 3; ++i) {
            if (!(streams & (1u << i))) continue;
            if (first < 0) first = i;
            res.text[i].resize_uninit(bytes[i] + 1);
            res.line_start[i].resize_uninit(nreads + 1);
            tp[i] = res.text[i].data();
            lp[i] = res.line_start[i].data();
        }
        rc = spx_query_text_fetch(ix, tp, lp);
        if (rc != SPX_OK) fatal_error("%s", spx_last_error());
        const uint64_t* ls = res.line_start[first].data();
        for (size_t q = 0; q < nreads; ++q) {  // (a read without values: its record is the header and a newline)
            res.beg[q] = 0;
            res.end[q] = ls[q + 1] - ls[q] - res.gap[q] - 1;
        }
        return;
    }
    if (!digest) {
        // SPUMONI_REPORT_ONLY (PML): the per-character values are neither written nor copied back
        const bool no_len = o.report_only && !o.ms && o.write_report;
        if (res.narrow)
            rc = spx_query_batch16(ix, o.ms ? SPX_MODE_MS : SPX_MODE_PML, sb.seqs.data(), sb.offs.data(), nreads,
                                   no_len ? nullptr : res.lengths16.data(), o.ms ? res.pointers.data() : nullptr,
                                   o.use_doc ? res.docs16.data() : nullptr, o.write_report ? res.cls.data() : nullptr,
                                   o.bin_size, max_value_thr);
        else
            rc = spx_query_batch(ix, o.ms ? SPX_MODE_MS : SPX_MODE_PML, sb.seqs.data(), sb.offs.data(), nreads,
                                 no_len ? nullptr : res.lengths.data(), o.ms ? res.pointers.data() : nullptr,
                                 o.use_doc ? res.docs.data() : nullptr, o.write_report ? res.cls.data() : nullptr,
                                 o.bin_size, max_value_thr);
        for (size_t q = 0; q < nreads; ++q) {
            res.beg[q] = sb.offs[q];
            res.end[q] = sb.offs[q + 1];
        }
    } else {
        // perform_minimizer_digestion / perform_dna_minimizer_digestion + matching_statistics
        // (compute_ms_pml.cpp:919-938), the batch on the device in one call
        std::vector<uint64_t> doffs(nreads + 1);
        rc = spx_digest_query_batch(ix, o.ms ? SPX_MODE_MS : SPX_MODE_PML, kind, (uint32_t)o.k, (uint32_t)o.w,
                                    sb.seqs.data(), sb.offs.data(), nreads, doffs.data(), total * grow,
                                    res.lengths.data(), o.ms ? res.pointers.data() : nullptr,
                                    o.use_doc ? res.docs.data() : nullptr, o.write_report ? res.cls.data() : nullptr,
                                    o.bin_size, max_value_thr);
        if (rc == SPX_OK)
            for (size_t q = 0; q < nreads; ++q) {
                res.beg[q] = doffs[q];
                res.end[q] = doffs[q + 1];
            }
    }
    if (rc != SPX_OK) fatal_error("%s", spx_last_error());
}

// The four output files as plain descriptors with a running end offset each: a super-batch is
// formatted by several host threads into their own buffers, the buffers' sizes are prefix-summed,
// and every thread pwrite()s its part at its own offset -- no concatenation, no serial write.
// (Copying the parts into a shared mapping of the file's new tail instead was tried: on tmpfs 0.69 s against
// 0.54 s for 2 GB -- allocating the file's pages is what takes the time, whichever way they are touched.)
struct OutFile {
    int fd = -1;
    uint64_t end = 0;
    void open(const std::string& path) {
        // A large output of an earlier run under the same name: truncating it gives its pages back synchronously
        // (0.2 s for 2 GB on tmpfs, inside "processing the patterns").  It is moved aside and removed on a thread
        // of its own instead; the new file starts empty either way.
        struct stat st;
        // (lstat: a symbolic link is left alone and its target truncated as before; the old file is registered, so that an
        // early exit removes it too)
        if (::lstat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > (64 << 20)) {
            const std::string old = path + ".old." + std::to_string((long)::getpid());
            if (::rename(path.c_str(), old.c_str()) == 0) {
                register_leftover(old);
                std::thread([old] { ::unlink(old.c_str()); }).detach();
            }
        }
        fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) fatal_error("cannot create %s", path.c_str());
        end = 0;
    }
    bool is_open() const { return fd >= 0; }
    void write_at(const char* p, size_t n, uint64_t at) const {
        while (n > 0) {
            const ssize_t w = ::pwrite(fd, p, n, (off_t)at);
            if (w <= 0) fatal_error("write failed (disk full?)");
            p += w;
            n -= (size_t)w;
            at += (uint64_t)w;
        }
    }
    void append(const std::string& s) {
        write_at(s.data(), s.size(), end);
        end += s.size();
    }
    ~OutFile() {
        if (fd >= 0) ::close(fd);
    }
};

struct Outputs {
    OutFile lengths, pointers, docs, report;
};

// Formats reads [lo, hi) of a super-batch into text; run by several host threads at once
// (the text of 10^6 reads x 200 values is ~0.5 GB: at GPU speed, formatting is the job).
struct TextChunk {
    TextBuf tl, tp, td;
    std::string report;
};

void format_range(const RunOptions& o, const SuperBatch& sb, const Results& res, size_t lo, size_t hi,
                  TextChunk& out) {
    out.report.clear();
    out.tl.clear();
    out.tp.clear();
    out.td.clear();
    for (size_t q = lo; q < hi; ++q) {
        const uint64_t a = res.beg[q], b = res.end[q];
        if (o.use_doc) {  // compute_ms_pml.cpp:1003-1007
            out.td.header(sb.ids[q]);
            if (res.narrow)
                out.td.values(res.docs16.data() + a, b - a);
            else
                out.td.values(res.docs.data() + a, b - a);
        }
        if (!o.report_only) {
            out.tl.header(sb.ids[q]);  // :1008-1010
            if (res.narrow)
                out.tl.values(res.lengths16.data() + a, b - a);
            else
                out.tl.values(res.lengths.data() + a, b - a);
        }
        if (o.ms) {  // :1190-1195
            out.tp.header(sb.ids[q]);
            out.tp.values(res.pointers.data() + a, b - a);
        }
        if (o.write_report) {  // :1012-1020
            const spx_class& c = res.cls[q];
            const size_t nbins = (size_t)c.bins_above + c.bins_below;
            const bool read_found = (c.bins_above / (c.bins_above + c.bins_below + 0.0) > 0.50);
            // setw(30) << left << id << setw(15) << status << setw(26) << avg (precision 3, default float format
            // = %.3g) << setw(12) << above << setw(12) << below: the same bytes through snprintf (an ostringstream
            // per thread spent 0.25 s on 4*10^6 such lines)
            const std::string_view id = sb.ids[q];
            char line[160];
            const int k = std::snprintf(line, sizeof line, "%-15s%-26.3g%-12zu%-12zu\n", read_found ? "FOUND" : "NOT_PRESENT",
                                        (c.sum_max_bin_values + 0.0) / nbins, (size_t)c.bins_above, (size_t)c.bins_below);
            out.report.append(id.data(), id.size());
            if (id.size() < 30) out.report.append(30 - id.size(), ' ');
            out.report.append(line, (size_t)k);
        }
    }
}

double g_format_s = 0;  // formatting alone, thread 0's share of every super-batch (for [timing])

// The text came from the device (spx_text.hip).  Two stages, so that the one thing that cannot be done in parallel --
// writes to one file serialise on its inode lock: 2 GB go into tmpfs at 5.7 GB/s however many threads call pwrite -- is
// never waited for by anything else:
//   begin   helper threads drop the ">id\n" lines into their gaps, then go on to format the report lines of their
//           ranges; the calling (writer) thread waits for the headers only and writes each stream with ONE pwrite;
//   finish  (the report thread, one super-batch behind) joins the helpers and appends the report lines.
struct TextJob {
    RunOptions ro;
    size_t nt = 0, nreads = 0;
    std::mutex mu;
    std::condition_variable cv;
    size_t filled = 0;
    std::vector<std::thread> th;
    std::vector<TextChunk> chunks;
};

void write_results_text_begin(Outputs& out, const RunOptions& o, const SuperBatch& sb, const Results& res, TextJob& job) {
    const size_t nreads = sb.nreads();
    const size_t nt = std::max<size_t>(1, std::min<size_t>(o.format_threads, (nreads + 4095) / 4096));
    job.nt = nt;
    job.nreads = nreads;
    job.filled = 0;
    if (job.chunks.size() < nt) job.chunks.resize(nt);
    job.ro = o;  // only the report is left to format
    job.ro.use_doc = false;
    job.ro.ms = false;
    job.ro.report_only = true;
    OutFile* const files[3] = {&out.lengths, &out.pointers, &out.docs};
    bool open_[3];
    for (int i = 0; i < 3; ++i) open_[i] = (res.streams & (1u << i)) && files[i]->is_open();
    const SuperBatch* psb = &sb;
    const Results* pres = &res;
    TextJob* pj = &job;
    const bool o0 = open_[0], o1 = open_[1], o2 = open_[2];
    auto helper = [psb, pres, pj, o0, o1, o2](size_t t) {
        const size_t lo = pj->nreads * t / pj->nt, hi = pj->nreads * (t + 1) / pj->nt;
        const bool op[3] = {o0, o1, o2};
        for (int i = 0; i < 3; ++i) {
            if (!op[i]) continue;
            char* base = const_cast<char*>(pres->text[i].data());
            const uint64_t* ls = pres->line_start[i].data();
            for (size_t q = lo; q < hi; ++q) {
                char* p = base + ls[q];
                const std::string_view id = psb->ids[q];
                *p++ = '>';
                std::memcpy(p, id.data(), id.size());
                p[id.size()] = '\n';
            }
        }
        {
            std::lock_guard<std::mutex> g(pj->mu);
            if (++pj->filled == pj->nt) pj->cv.notify_all();
        }
        format_range(pj->ro, *psb, *pres, lo, hi, pj->chunks[t]);
    };
    job.th.clear();
    for (size_t t = 0; t < nt; ++t) job.th.emplace_back(helper, t);
    const auto tf0 = std::chrono::steady_clock::now();
    {
        std::unique_lock<std::mutex> g(job.mu);
        job.cv.wait(g, [&] { return job.filled == nt; });
    }
    for (int i = 0; i < 3; ++i) {
        if (!open_[i]) continue;
        const uint64_t bytes = res.line_start[i][nreads];
        files[i]->write_at(res.text[i].data(), bytes, files[i]->end);
        files[i]->end += bytes;
    }
    g_format_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf0).count();
}

void write_results_text_finish(Outputs& out, const RunOptions& o, TextJob& job) {
    for (auto& x : job.th) x.join();
    job.th.clear();
    if (o.write_report)
        for (size_t t = 0; t < job.nt; ++t) {
            const std::string& r = job.chunks[t].report;
            if (r.empty()) continue;
            out.report.write_at(r.data(), r.size(), out.report.end);
            out.report.end += r.size();
        }
}

void write_results(Outputs& out, const RunOptions& o, const SuperBatch& sb, const Results& res,
                   std::vector<TextChunk>& chunks) {
    if (res.device_text) {  // (callers without a report stage: both stages at once)
        TextJob job;
        write_results_text_begin(out, o, sb, res, job);
        write_results_text_finish(out, o, job);
        return;
    }
    const size_t nreads = sb.nreads();
    const size_t nt = std::max<size_t>(1, std::min<size_t>(o.format_threads, (nreads + 4095) / 4096));
    if (chunks.size() < nt) chunks.resize(nt);
    // where thread t's text goes in each file: known once every thread has formatted its part
    std::vector<uint64_t> at_l(nt + 1), at_p(nt + 1), at_d(nt + 1), at_r(nt + 1);
    std::mutex mu;
    std::condition_variable cv;
    size_t formatted = 0;
    bool placed = false;
    auto lo_of = [&](size_t t) { return nreads * t / nt; };
    auto work = [&](size_t t) {
        TextChunk& c = chunks[t];
        const auto tf0 = std::chrono::steady_clock::now();
        format_range(o, sb, res, lo_of(t), lo_of(t + 1), c);
        if (t == 0) g_format_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf0).count();
        {
            std::unique_lock<std::mutex> g(mu);
            if (++formatted == nt) {  // the last one to finish lays the chunks out, in input order
                at_l[0] = out.lengths.end;
                at_p[0] = out.pointers.end;
                at_d[0] = out.docs.end;
                at_r[0] = out.report.end;
                for (size_t i = 0; i < nt; ++i) {
                    at_l[i + 1] = at_l[i] + chunks[i].tl.len;
                    at_p[i + 1] = at_p[i] + chunks[i].tp.len;
                    at_d[i + 1] = at_d[i] + chunks[i].td.len;
                    at_r[i + 1] = at_r[i] + chunks[i].report.size();
                }
                placed = true;
                cv.notify_all();
            } else {
                cv.wait(g, [&] { return placed; });
            }
        }
        if (c.tl.len) out.lengths.write_at(c.tl.buf.data(), c.tl.len, at_l[t]);
        if (o.ms && c.tp.len) out.pointers.write_at(c.tp.buf.data(), c.tp.len, at_p[t]);
        if (o.use_doc && c.td.len) out.docs.write_at(c.td.buf.data(), c.td.len, at_d[t]);
        if (o.write_report && !c.report.empty()) out.report.write_at(c.report.data(), c.report.size(), at_r[t]);
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    out.lengths.end = at_l[nt];
    out.pointers.end = at_p[nt];
    out.docs.end = at_d[nt];
    out.report.end = at_r[nt];
}

size_t open_outputs_and_threshold(Outputs& out, const RunOptions& o) {
    out.lengths.open(o.pattern_file + (o.ms ? ".lengths" : ".pseudo_lengths"));
    if (o.ms) out.pointers.open(o.pattern_file + ".pointers");
    if (o.use_doc) out.docs.open(o.pattern_file + ".doc_numbers");
    if (o.write_report) out.report.open(o.pattern_file + ".report");
    double percentile = 0.0;
    std::string err;
    // the reference does not check the stream either (:867-869): a missing null database
    // leaves percentile_value at 0.0
    (void)load_null_db(o.ref_file + (o.ms ? ".msnulldb" : ".pmlnulldb"), percentile, err);
    const size_t max_value_thr = max_value_threshold(percentile, !o.ms, o.use_promotions, o.use_dna_letters);
    if (o.write_report) {  // :877-886
        std::ostringstream hd;
        hd.precision(4);
        hd << std::setw(30) << std::left << "read id:" << std::setw(15) << std::left << "status:"
           << std::setw(19) << std::left << "avg max-value (thr=" << std::setw(2) << std::left
           << max_value_thr << std::setw(5) << std::left << "):" << std::setw(12) << std::left
           << "above thr:" << std::setw(12) << std::left << "below thr:" << std::endl;
        out.report.append(hd.str());
    }
    return max_value_thr;
}

}  // namespace

namespace {

// A super-batch in flight plus what has to happen after it was written.
struct Slot {
    SuperBatch sb;
    Results res;
    uint64_t seq = 0;            // position of this super-batch in the input (results are written in this order)
    bool last = false;           // no more input after this one
    std::vector<std::vector<ParsedRead>> parsed;  // fill_slot's per-thread reads (kept: their capacity is reused)
    std::unique_ptr<TextJob> job;  // device text: the helpers of the write in flight (finished by the report thread)
    int deferred = 0;            // 0 none, 1 FATAL_ERROR, 2 "empty after digestion" FATAL_WARNING
    std::string deferred_msg;
};

// blocking hand-off of slot indices between the three stages (parse -> GPU -> write)
class SlotQueue {
public:
    void push(int v) {
        std::lock_guard<std::mutex> g(mu_);
        q_.push_back(v);
        cv_.notify_one();
    }
    int pop() {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return !q_.empty(); });
        int v = q_.front();
        q_.erase(q_.begin());
        return v;
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<int> q_;
};

// Fills `slot` from the input: segmentation replayed sequentially (reads.cpp), the batches of
// the super-batch parsed by several threads, reads upper-cased while they are copied into the
// page-locked buffer.  A malformed / empty read truncates the super-batch there and is reported
// after everything before it has been written, like the reference running read by read.
double g_parse_s[4] = {0, 0, 0, 0};  // fill_slot: segmentation (serial), parse, placement (serial), copy
void fill_slot(ReadFile& input, const RunOptions& o, Slot& slot, bool& input_done) {
    auto tick = [] { return std::chrono::steady_clock::now(); };
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    auto t_phase = tick();
    slot.sb.clear();
    slot.last = false;
    slot.deferred = 0;
    std::vector<ReadFile::Range> ranges;
    size_t bytes = 0;
    while (!input_done && bytes < o.super_batch_chars) {
        ReadFile::Range r;
        if (!input.next_range(1000, r)) {  // reader.loadBatch(input_file, 1000)   (:903)
            input_done = true;
            break;
        }
        bytes += r.bytes;
        ranges.push_back(r);
    }
    slot.last = input_done;
    g_parse_s[0] += since(t_phase);
    t_phase = tick();
    const size_t nt = std::max<size_t>(1, std::min<size_t>(o.format_threads, (ranges.size() + 63) / 64));
    std::vector<std::vector<ParsedRead>>& parsed = slot.parsed;
    if (parsed.size() < nt) parsed.resize(nt);
    for (auto& v : parsed) v.clear();
    std::vector<ReadFile::ParseError> errs(nt);
    std::vector<size_t> err_at(nt, 0);
    auto work = [&](size_t t) {
        const size_t lo = ranges.size() * t / nt, hi = ranges.size() * (t + 1) / nt;
        for (size_t i = lo; i < hi && !errs[t].fatal; ++i) input.parse_range(ranges[i], parsed[t], errs[t]);
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    g_parse_s[1] += since(t_phase);
    t_phase = tick();
    // where does every thread's part go, and where is the first problem (if any)?
    std::vector<size_t> take(nt, 0), chars(nt, 0);
    for (size_t t = 0; t < nt && !slot.deferred; ++t) {
        for (const ParsedRead& rd : parsed[t]) {
            if (rd.seq().empty()) {  // :926-931
                slot.deferred = 2;
                slot.deferred_msg = std::string(rd.id);
                break;
            }
            take[t]++;
            chars[t] += rd.seq().size();
        }
        if (!slot.deferred && errs[t].fatal) {
            slot.deferred = 1;
            slot.deferred_msg = errs[t].message;
        }
    }
    if (slot.deferred) slot.last = true;
    std::vector<size_t> r0(nt + 1, 0), c0(nt + 1, 0);
    for (size_t t = 0; t < nt; ++t) {
        r0[t + 1] = r0[t] + take[t];
        c0[t + 1] = c0[t] + chars[t];
    }
    const size_t nreads = r0[nt], nchars = c0[nt];
    slot.sb.ids.resize(nreads);
    slot.sb.offs.resize(nreads + 1);
    slot.sb.offs[0] = 0;
    slot.sb.seqs.resize_uninit(nchars);
    g_parse_s[2] += since(t_phase);
    t_phase = tick();
    auto assemble = [&](size_t t) {
        size_t rdx = r0[t], cpos = c0[t];
        for (size_t q = 0; q < take[t]; ++q) {
            ParsedRead& rd = parsed[t][q];
            uint8_t* dst = slot.sb.seqs.data() + cpos;
            const char* src = rd.seq().data();
            const size_t len = rd.seq().size();
            // make sure all characters are upper-case (:916-917; ::toupper in the "C" locale)
            for (size_t i = 0; i < len; ++i) {
                const unsigned char ch = (unsigned char)src[i];
                dst[i] = (uint8_t)((ch >= 'a' && ch <= 'z') ? ch - 32 : ch);
            }
            cpos += len;
            slot.sb.offs[rdx + 1] = cpos;
            slot.sb.ids[rdx] = rd.id;
            rdx++;
        }
    };
    th.clear();
    for (size_t t = 1; t < nt; ++t) th.emplace_back(assemble, t);
    assemble(0);
    for (auto& x : th) x.join();
    g_parse_s[3] += since(t_phase);
}

}  // namespace

namespace {
struct StageTimer {  // per-stage wall time on stderr (ours)
    const char* name;
    double total = 0;
    std::chrono::steady_clock::time_point t0;
    void start() { t0 = std::chrono::steady_clock::now(); }
    void stop() { total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

// Re-sequences the super-batches the device workers finish, in whatever order, into input order.
class OrderedDone {
public:
    void put(uint64_t seq, int slot) {
        std::lock_guard<std::mutex> g(mu_);
        done_.emplace_back(seq, slot);
        cv_.notify_all();
    }
    int take(uint64_t seq) {  // blocks until super-batch `seq` is done
        std::unique_lock<std::mutex> g(mu_);
        for (;;) {
            for (size_t i = 0; i < done_.size(); ++i)
                if (done_[i].first == seq) {
                    const int slot = done_[i].second;
                    done_.erase(done_.begin() + (long)i);
                    return slot;
                }
            cv_.wait(g);
        }
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::pair<uint64_t, int>> done_;
};

// Called on a helper thread while the index loads: the page-locked blocks the slots of classify_reads will ask for
// (per slot: the reads of a super-batch, and per output stream its text and its record offsets).
void prepare_pinned_pool(const RunOptions& o, size_t ndev) {
    if (spx_device_count() <= 0) return;
    // sized from the reads file, not from the super-batch limit alone: a small file needs small blocks and few
    // slots, and below a megabyte locking pages ahead of time buys nothing
    struct stat sb;
    if (::stat(o.pattern_file.c_str(), &sb) != 0 || sb.st_size < (1 << 20)) return;
    const size_t fsize = (size_t)sb.st_size;
    const size_t batches = fsize / std::max<size_t>(o.super_batch_chars, 1) + 1;
    const size_t nslots = std::min<size_t>(2 * std::max<size_t>(ndev, 1) + 3, batches + 1);
    const bool report_only = o.report_only && !o.ms && o.write_report;
    const size_t chars = std::min<size_t>(o.super_batch_chars, fsize) + std::min<size_t>(4u << 20, fsize / 8 + 4096);
    std::vector<size_t> sizes;
    for (size_t i = 0; i < nslots; ++i) {
        sizes.push_back(chars);  // reads
        if (std::getenv("SPUMONI_HOST_FORMAT")) continue;
        const size_t reads_guess = chars / 100 + 4096;
        if (!report_only) {  // lengths: "<value> " is 2-4 bytes for most values
            sizes.push_back(chars * 3 + std::min<size_t>(8u << 20, chars));
            sizes.push_back((reads_guess + 1) * 8);
        }
        if (o.ms) {  // pointers: up to 13 digits
            sizes.push_back(chars * 12);
            sizes.push_back((reads_guess + 1) * 8);
        }
        if (o.use_doc) {
            sizes.push_back(chars * 3);
            sizes.push_back((reads_guess + 1) * 8);
        }
    }
    for (size_t b : sizes) {
        void* p = spx_host_alloc(b);
        if (!p) return;  // (the slots then allocate what they need themselves)
        g_pinned_pool.put(p, b);
    }
}

size_t classify_reads(IndexSet& set, const RunOptions& o, ReadFile* preloaded) {
    Outputs out;
    const size_t max_value_thr = open_outputs_and_threshold(out, o);
    StageTimer t_load{"load+index lines", 0, {}}, t_parse{"segment+parse", 0, {}}, t_write{"format+write", 0, {}},
        t_report{"report (one behind)", 0, {}};
    t_load.start();
    // the reads file is mapped and its lines are found while the index loads (spumoni_main.cpp)
    std::unique_ptr<ReadFile> own_input;
    if (!preloaded) {
        own_input.reset(new ReadFile(o.pattern_file, (unsigned)o.format_threads));
        preloaded = own_input.get();
    }
    ReadFile& input = *preloaded;
    t_load.stop();
    // Parser -> ONE queue of parsed super-batches -> one worker thread per device -> ordered writer.
    // Every device pulls its next super-batch when it is free (reads are independent,
    // compute_ms_pml.cpp:907-938: nothing is carried from one read to the next), so a slower or busier
    // device simply takes fewer of them; the writer puts the results back into input order (the
    // reference's -t 1 order).  Slots: one being parsed, one per device, one being written, and one
    // more per device so that no device waits for the parser.
    const size_t ndev = set.ix.size();
    const int NSLOTS = (int)(2 * ndev + 3);  // (one more: the report thread holds a slot too)
    // (the slots outlive the call on purpose: unlocking their ~1 GB of page-locked buffers takes a few tenths of a
    // second that the run would spend after its last byte is written; the process ends right after and the
    // operating system takes the pages back)
    std::vector<Slot>& slots = *new std::vector<Slot>((size_t)NSLOTS);
    const auto t_stage0 = std::chrono::steady_clock::now();
    SlotQueue free_q, parsed_q;
    OrderedDone done;
    for (int i = 0; i < NSLOTS; ++i) free_q.push(i);
    size_t num_reads = 0;
    std::vector<double> dev_busy(ndev, 0.0);
    std::vector<size_t> dev_batches(ndev, 0);
    auto device_worker = [&](size_t d) {
        for (;;) {
            const int i = parsed_q.pop();
            if (i < 0) break;
            const auto t0 = std::chrono::steady_clock::now();
            Slot& s = slots[(size_t)i];
            if (s.sb.nreads() > 0) run_on_device(set.ix[d], o, s.sb, max_value_thr, s.res);
            if (o.use_promotions || o.use_dna_letters) {
                // a read that digests to nothing is fatal where the reference meets it (:926-931):
                // everything before it is still written
                for (size_t q = 0; q < s.sb.nreads(); ++q)
                    if (s.res.beg[q] == s.res.end[q]) {
                        s.deferred = 2;
                        s.deferred_msg = std::string(s.sb.ids[q]);
                        s.sb.ids.resize(q);
                        s.sb.offs.resize(q + 1);
                        break;
                    }
            }
            dev_busy[d] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            dev_batches[d]++;
            done.put(s.seq, i);
        }
    };
    std::vector<std::thread> workers;
    for (size_t d = 0; d < ndev; ++d) workers.emplace_back(device_worker, d);
    // writer: the streams, in input order; report thread, one super-batch behind: the report lines, the slot's release
    // and whatever the reference would have stopped at
    SlotQueue report_q;
    std::thread writer([&] {
        std::vector<TextChunk> chunks;  // formatting buffers of the host-formatted path, kept across super-batches
        for (uint64_t seq = 0;; ++seq) {
            const int i = done.take(seq);
            Slot& s = slots[(size_t)i];
            t_write.start();
            if (s.sb.nreads() > 0) {
                if (s.res.device_text) {
                    if (!s.job) s.job.reset(new TextJob());
                    write_results_text_begin(out, o, s.sb, s.res, *s.job);
                } else {
                    write_results(out, o, s.sb, s.res, chunks);
                }
            }
            t_write.stop();
            // a super-batch that ends in a deferred fatal (a read empty after digestion is only known once the batch is
            // back from its device) is the last one written: the report thread, one batch behind, stops the run there,
            // and nothing of a later super-batch may reach the files before it does (the reference stops AT the read)
            const bool last = s.last || s.deferred != 0;
            report_q.push(i);
            if (last) break;
        }
    });
    std::thread reporter([&] {
        for (;;) {
            const int i = report_q.pop();
            Slot& s = slots[(size_t)i];
            t_report.start();
            if (s.sb.nreads() > 0 && s.res.device_text && s.job) write_results_text_finish(out, o, *s.job);
            t_report.stop();
            num_reads += s.sb.nreads();
            if (s.deferred == 1) fatal_error("%s", s.deferred_msg.c_str());
            if (s.deferred == 2) {
                std::cout << "\n\n";
                fatal_warning("%s was empty after digestion, commonly due to reads "
                              "consisting of mostly non-ACGT characters. Please remove "
                              "read or run SPUMONI without minimizer digestion.", s.deferred_msg.data());
            }
            const bool last = s.last;
            free_q.push(i);
            if (last) break;
        }
    });
    bool input_done = false;
    for (uint64_t seq = 0;; ++seq) {
        const int i = free_q.pop();
        t_parse.start();
        fill_slot(input, o, slots[(size_t)i], input_done);
        t_parse.stop();
        slots[(size_t)i].seq = seq;
        const bool last = slots[(size_t)i].last;
        parsed_q.push(i);
        if (last) break;
    }
    for (size_t d = 0; d < ndev; ++d) parsed_q.push(-1);
    for (auto& w : workers) w.join();
    writer.join();
    reporter.join();
    std::fprintf(stderr, "[timing] %-22s %.3f s\n", "first read .. last byte",
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t_stage0).count());
    // per-stage wall times (ours, additive; stages overlap, so they do not add up to the total)
    for (StageTimer* t : {&t_load, &t_parse, &t_write, &t_report})
        std::fprintf(stderr, "[timing] %-22s %.3f s\n", t->name, t->total);
    std::fprintf(stderr, "[timing]   segmentation %.3f  parse %.3f  placement %.3f  copy %.3f s\n", g_parse_s[0], g_parse_s[1],
                 g_parse_s[2], g_parse_s[3]);
    std::fprintf(stderr, "[timing] %-22s %.3f s\n", "  of which formatting", g_format_s);
    for (size_t d = 0; d < ndev; ++d)
        std::fprintf(stderr, "[timing] gpu worker %zu          %.3f s  (%zu super-batches, copies included)\n", d,
                     dev_busy[d], dev_batches[d]);
    return num_reads;
}

size_t classify_general_reads(IndexSet& set, const RunOptions& o) {
    // :1219-1297: reads are separated by \x01; trailing text without a separator is ignored
    RunOptions oo = o;
    oo.use_doc = false;
    oo.write_report = false;
    Outputs out;
    out.lengths.open(o.pattern_file + (o.ms ? ".lengths" : ".pseudo_lengths"));
    if (o.ms) out.pointers.open(o.pattern_file + ".pointers");
    std::vector<uint8_t> data;
    if (!read_whole_file(o.pattern_file, data)) fatal_error("The following path is not valid: %s", o.pattern_file.data());
    SuperBatch sb;
    Results res;
    std::vector<TextChunk> chunks;
    size_t num_reads = 0, start = 0;
    auto flush = [&]() {
        if (sb.nreads() == 0) return;
        run_on_device(set.ix[0], oo, sb, 0, res);
        write_results(out, oo, sb, res, chunks);
        sb.clear();
    };
    for (size_t i = 0; i < data.size(); ++i) {
        if (data[i] == 0x01) {
            sb.seqs.append(data.data() + start, i - start);
            sb.offs.push_back(sb.seqs.size());
            sb.own_ids.push_back("read_" + std::to_string(num_reads));
            sb.ids.push_back(sb.own_ids.back());
            num_reads++;
            start = i + 1;
            if (sb.seqs.size() >= o.super_batch_chars) flush();
        }
    }
    flush();
    return num_reads;
}

}  // namespace spumoni_host
