#include "classify.hpp"

#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>

#include <algorithm>
#include <array>
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
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
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
    // Replicas: flatten once, copy to every OTHER device once -- as a doubling tree (device 0 -> 1; 0 -> 2, 1 -> 3; 0 -> 4 ...
    // 3 -> 7), the copies of a round on threads of their own: every copy has its own source and, on an xGMI node, its own
    // link, so seven replicas of a 200 GB index cost three copy times instead of seven.  An entry of the
    // device list that names a device a second time is a second WORKER on it: a query context of its own (scratch,
    // stream, counters) over the arrays that are already there (spx_index_clone onto the same device copies nothing).
    const std::vector<int> devs = o.devices.empty() ? std::vector<int>{0} : o.devices;
    const size_t nwork = devs.size();
    const auto t_all = std::chrono::steady_clock::now();
    ix.resize(nwork, nullptr);
    std::vector<size_t> uniq;  // workers that are the first on their device
    for (size_t i = 0; i < nwork; ++i) {
        bool seen = false;
        for (size_t j : uniq) seen = seen || devs[j] == devs[i];
        if (!seen) uniq.push_back(i);
    }
    for (size_t have = 1; have < uniq.size();) {
        const size_t nnew = std::min(have, uniq.size() - have);
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::string> errs(nnew);
        std::vector<std::thread> th;
        for (size_t i = 0; i < nnew; ++i)
            th.emplace_back([&, i] {
                spx_index* p = spx_index_clone(ix[uniq[i]], devs[uniq[have + i]]);
                if (!p) errs[i] = spx_last_error();
                ix[uniq[have + i]] = p;
            });
        for (auto& t : th) t.join();
        for (size_t i = 0; i < nnew; ++i) {
            if (!ix[uniq[have + i]]) fatal_error("%s", errs[i].c_str());
            std::fprintf(stderr, "[timing] index replica on device %d (copied from device %d)\n", devs[uniq[have + i]], devs[uniq[i]]);
        }
        std::fprintf(stderr, "[timing] %zu replica%s in parallel  %.3f s\n", nnew, nnew > 1 ? "s" : "",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        have += nnew;
    }
    if (uniq.size() > 1)
        std::fprintf(stderr, "[timing] all %zu index replicas  %.3f s\n", uniq.size() - 1,
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count());
    for (size_t i = 1; i < nwork; ++i) {
        if (ix[i]) continue;
        size_t src = 0;
        for (size_t j : uniq)
            if (devs[j] == devs[i]) src = j;
        ix[i] = spx_index_clone(ix[src], devs[i]);
        if (!ix[i]) fatal_error("%s", spx_last_error());
        std::fprintf(stderr, "[timing] worker %zu: a second query context on device %d (shares the arrays of worker %zu)\n", i, devs[i], src);
    }
    // (the workers sleep while they wait for the device: the cores are the pool's -- SPUMONI_SPIN=1: they spin, the form before)
    if (!std::getenv("SPUMONI_SPIN"))
        for (size_t i = 0; i < nwork; ++i) (void)spx_set_option(ix[i], "blocking_sync", 1);
    // every worker's device scratch for a full super-batch, now: the first super-batch of each worker otherwise allocates it
    // inside the timed run (10 ms alone, 17-20 ms with two or three workers of one device queueing for the allocator), and a
    // buffer that grows mid-run is freed first, which synchronises the whole device (profiles/r05_cli_overlap.txt)
    struct stat rs;
    if (!o.is_general_text && !std::getenv("SPUMONI_HOST_FORMAT") && ::stat(o.pattern_file.c_str(), &rs) == 0 && rs.st_size > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        const uint64_t fsize = (uint64_t)rs.st_size;
        const uint64_t chars = std::min<uint64_t>(o.super_batch_chars, fsize) + 65536;
        const uint64_t reads_guess = chars / 100 + 4096;
        const bool report_only = o.report_only && !o.ms && o.write_report;
        const bool digest = o.use_promotions || o.use_dna_letters;
        const uint32_t streams = (report_only ? 0u : SPX_TEXT_LENGTHS) | (o.ms ? SPX_TEXT_POINTERS : 0u) | (o.use_doc ? SPX_TEXT_DOCS : 0u);
        const uint64_t text_bytes[3] = {chars * 3 + reads_guess * 40, chars * 11 + reads_guess * 40, chars * 3 + reads_guess * 40};
        if (streams != 0) {
            for (size_t i = 0; i < nwork; ++i)
                if (spx_query_text_reserve(ix[i], o.ms ? SPX_MODE_MS : SPX_MODE_PML, digest ? (o.use_promotions ? SPX_DIGEST_PROMOTED : SPX_DIGEST_DNA) : 0,
                                           (uint32_t)o.k, chars, reads_guess, streams, o.write_report ? 1 : 0, digest ? nullptr : text_bytes) != SPX_OK)
                    std::fprintf(stderr, "[spumoni-gpu] device scratch not reserved ahead of the run (%s): it will be allocated by the first super-batches\n", spx_last_error());
            // ... and one tiny query per worker: the first launch of every kernel on the handle's stream, the scan's temporary
            // storage, the published words -- 6 of the first super-batch's 7.6 ms (profiles/r05_cli_overlap.txt)
            if (!digest) {
                static const uint8_t warm_seq[64] = {'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C',
                                                     'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T',
                                                     'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T'};
                const uint64_t warm_off[3] = {0, 40, 64};
                const uint32_t warm_gap[2] = {3, 3};
                std::vector<std::thread> th;
                for (size_t i = 0; i < nwork; ++i)
                    th.emplace_back([&, i] {
                        spx_class cls[2];
                        uint64_t bytes[3] = {0, 0, 0};
                        if (spx_query_text_begin(ix[i], o.ms ? SPX_MODE_MS : SPX_MODE_PML, 0, (uint32_t)o.k, (uint32_t)o.w, warm_seq, warm_off, 2, warm_gap, streams,
                                                 o.write_report ? cls : nullptr, o.bin_size ? o.bin_size : 1, 1, bytes) != SPX_OK)
                            return;  // (nothing depends on it)
                        std::vector<char> t0(bytes[0] + 1), t1(bytes[1] + 1), t2(bytes[2] + 1);
                        char* tp[3] = {t0.data(), t1.data(), t2.data()};
                        (void)spx_query_text_fetch(ix[i], tp, nullptr);
                    });
                for (auto& t : th) t.join();
            }
            std::fprintf(stderr, "[timing] device scratch of %zu worker%s reserved in %.3f s\n", nwork, nwork > 1 ? "s" : "",
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
    }
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
    PinnedBuf<uint64_t> offs;           // nreads + 1 entries (page-locked: they travel to the device with the reads)
    PinnedBuf<uint32_t> gap;            // per read: bytes of its ">id\n" line (what the device leaves free in front of its values)
    uint64_t longest = 0;               // longest read
    SuperBatch() { offs.assign(1, 0); }
    void clear() {
        ids.clear();
        own_ids.clear();
        seqs.n = 0;
        gap.n = 0;
        longest = 0;
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
    char* text_at[3] = {nullptr, nullptr, nullptr};  // where stream i's text is: text[i], or memory of the output file itself
    PinnedBuf<uint64_t> line_start[3];
};

// One super-batch on one device (a worker thread of that device calls this): the batch form of the
// reference's loop body -- [digestion +] matching_statistics + bin classification
// (compute_ms_pml.cpp:916-995).  Only what the chosen path hands back is allocated (round 4 reserved the 16-bit value
// buffers on the text path as well: 128 MB of pages locked per slot for nothing, 10 of its 16 ms per super-batch).
// place(bytes, dest): called once per super-batch, between the walk and the copy-out on the text path (the streams' sizes
// are known then) and right away on the other paths (bytes all zero): the harness reserves the super-batch's place in every
// output file there, in input order, and may name, per stream, memory of the file itself for the text to land in
// (dest[i] stays null: the slot's page-locked buffer is used and a writer thread copies it).
using PlaceFn = std::function<void(const uint64_t bytes[3], char* dest[3])>;
// SPUMONI_CALL_TRACE=1: when every worker entered and left the library (begin = copy in + walk + sizes, fetch = digits + copy
// out), microseconds on one clock -- who waited for whom on the device (profiles/r05_cli_overlap.txt)
struct CallTrace {
    std::mutex mu;
    struct Rec { const void* ix; const char* what; double t0, t1; };
    std::vector<Rec> recs;
    const bool on = std::getenv("SPUMONI_CALL_TRACE") != nullptr;
    const std::chrono::steady_clock::time_point origin = std::chrono::steady_clock::now();
    double now() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - origin).count() * 1e3; }
    void add(const void* ix, const char* what, double t0) {
        if (!on) return;
        const double t1 = now();
        std::lock_guard<std::mutex> g(mu);
        recs.push_back(Rec{ix, what, t0, t1});
    }
    void print() {
        if (!on || recs.empty()) return;
        std::sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.t0 < b.t0; });
        const double base = recs.front().t0;
        for (const Rec& r : recs) std::fprintf(stderr, "[calls] %p %-6s %8.3f .. %8.3f  (%.3f ms)\n", r.ix, r.what, r.t0 - base, r.t1 - base, r.t1 - r.t0);
        recs.clear();
    }
};
CallTrace g_calls;
void run_on_device(spx_index* ix, const RunOptions& o, SuperBatch& sb, size_t max_value_thr, Results& res, const PlaceFn& place) {
    const size_t nreads = sb.nreads();
    const uint64_t total = sb.offs.back();
    const bool digest = o.use_promotions || o.use_dna_letters;
    const int kind = o.use_promotions ? SPX_DIGEST_PROMOTED : SPX_DIGEST_DNA;
    // with digestion the results are laid out at the digested offsets, inside a region as large as
    // the worst case (every k-mer reported: 1 byte each for -m, k letters for -a)
    const uint64_t grow = digest && o.use_dna_letters ? (uint64_t)o.k : 1;
    res.narrow = !digest && sb.longest < 65536;
    if (o.write_report) res.cls.resize_uninit(nreads);
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
        uint64_t bytes[3] = {0, 0, 0};
        const double tc0 = g_calls.now();
        rc = spx_query_text_begin(ix, o.ms ? SPX_MODE_MS : SPX_MODE_PML, digest ? kind : 0, (uint32_t)o.k, (uint32_t)o.w,
                                  sb.seqs.data(), sb.offs.data(), nreads, sb.gap.data(), streams,
                                  o.write_report ? res.cls.data() : nullptr, o.bin_size, max_value_thr, bytes);
        if (rc != SPX_OK) fatal_error("%s", spx_last_error());
        g_calls.add(ix, "begin", tc0);
        char* tp[3] = {nullptr, nullptr, nullptr};
        uint64_t* lp[3] = {nullptr, nullptr, nullptr};
        place(bytes, tp);
        int first = -1;
        for (int i = 0; i < 3; ++i) {
            res.text_at[i] = nullptr;
            if (!(streams & (1u << i))) continue;
            if (first < 0) first = i;
            if (!tp[i]) {
                res.text[i].resize_uninit(bytes[i] + 1);
                tp[i] = res.text[i].data();
            }
            res.text_at[i] = tp[i];
            res.line_start[i].resize_uninit(nreads + 1);
            lp[i] = res.line_start[i].data();
        }
        const double tc1 = g_calls.now();
        rc = spx_query_text_fetch(ix, tp, lp);
        g_calls.add(ix, "fetch", tc1);
        if (rc != SPX_OK) fatal_error("%s", spx_last_error());
        if (digest) {  // (only a digested read can come back without values: the worker looks for the first one)
            res.beg.resize(nreads);
            res.end.resize(nreads);
            const uint64_t* ls = res.line_start[first].data();
            for (size_t q = 0; q < nreads; ++q) {  // (a read without values: its record is the header and a newline)
                res.beg[q] = 0;
                res.end[q] = ls[q + 1] - ls[q] - sb.gap[q] - 1;
            }
        }
        return;
    }
    {
        const uint64_t none[3] = {0, 0, 0};
        char* unused[3] = {nullptr, nullptr, nullptr};
        place(none, unused);
    }
    res.beg.resize(nreads);
    res.end.resize(nreads);
    // SPUMONI_REPORT_ONLY (PML): the per-character values are neither written nor copied back
    const bool no_len = o.report_only && !o.ms && o.write_report;
    if (res.narrow) {
        if (!no_len) res.lengths16.resize_uninit(total + 8);
        if (o.use_doc) res.docs16.resize_uninit(total + 8);
    } else {
        if (!no_len || digest) res.lengths.resize_uninit(total * grow);
        if (o.use_doc) res.docs.resize_uninit(total * grow);
    }
    if (o.ms) res.pointers.resize_uninit(total * grow);
    if (!digest) {
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

// The output files.  Writes to ONE file serialise on its inode lock whatever the number of writing threads
// (tools/drain_bench.cpp on the GPU box, profiles/r05_drain_bench.txt: one thread puts 6.5-7 GB/s of fresh pages into a
// tmpfs file, sixteen threads 3.4, sixteen threads into sixteen FILES 58) -- a 2 GB .pseudo_lengths cannot take more than
// 13 M reads/s that way.  So the file's tail is prepared as MEMORY while the index loads (prepare_outputs): allocated
// (fallocate), mapped and populated, and page-locked for the device (spx_host_register); the text of a super-batch is then
// copied by the device straight into the file's pages, at its place in input order (profiles/r05_drain_hip.txt: 57 GB/s
// -- the link), the ">id" lines and the report are written into the mapping by the pool, and nothing goes through write()
// at all.  What does not fit the prepared tail (the estimate was short), a file system that cannot be mapped, host-
// formatted runs: one writer thread per file with pwrite, as before, the files side by side.
struct OutFile {
    int fd = -1;
    uint64_t end = 0;          // logical size: everything below is final (batches complete in input order)
    uint64_t reserved = 0;     // device-text runs: where the next super-batch's share starts (OffsetOrder)
    char* map = nullptr;       // the file's first bytes as memory (MAP_SHARED), or null: mapped_bytes of them were mapped,
    uint64_t mapped_bytes = 0;  // and the first map_size may be written to (shrinks when the file's excess is trimmed away early)
    std::atomic<uint64_t> map_size{0};
    bool pinned = false;       // ... and page-locked: the device writes it -- its first pinned_size bytes (the rest of the mapping takes the
    uint64_t pinned_size = 0;  // text through a page-locked buffer and the pool's memcpy, and its excess can be cut off beside the run)
    std::mutex cut_mu;         // held while the file's size is cut (EarlyTrim, settle) and while a writer thread writes to it
    bool settled = false;      // (under cut_mu) cut to its logical size: no one cuts it again
    std::string temp_path;     // prepared ahead of time under this name, renamed when the run starts
    void open(const std::string& path) {
        // A large output of an earlier run under the same name: truncating it gives its pages back synchronously
        // (0.2 s for 2 GB on tmpfs, inside "processing the patterns").  It is moved aside and removed on a thread
        // of its own instead; the new file starts empty either way.
        struct stat st;
        // (lstat: a symbolic link is left alone and its target truncated as before; the old file is registered, so that an
        // early exit removes it too)
        const bool regular = ::lstat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
        if (regular && st.st_size > (64 << 20)) {
            const std::string old = path + ".old." + std::to_string((long)::getpid());
            if (::rename(path.c_str(), old.c_str()) == 0) {
                register_leftover(old);
                std::thread([old] { ::unlink(old.c_str()); }).detach();
            }
        }
        if (fd >= 0) {
            // prepared under another name: it takes the name now (where the name is a symbolic link or something else that
            // is not a plain file the prepared file is dropped and the target opened as always)
            const bool plain_target = ::lstat(path.c_str(), &st) != 0 || S_ISREG(st.st_mode);
            if (plain_target && ::rename(temp_path.c_str(), path.c_str()) == 0) return;
            drop_mapping();
            ::close(fd);
            ::unlink(temp_path.c_str());
            fd = -1;
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
    void append(const char* p, size_t n) {
        write_at(p, n, end);
        end += n;
    }
    void append(const std::string& s) { append(s.data(), s.size()); }
    void drop_mapping() {
        if (!map) return;
        if (pinned) (void)spx_host_unregister(map);
        ::munmap(map, mapped_bytes);
        map = nullptr;
        mapped_bytes = 0;
        map_size = 0;
        pinned = false;
        pinned_size = 0;
    }
    // the file ends where its last complete super-batch does (the prepared tail was an estimate; after a fatal read the
    // device may already have written later super-batches behind it)
    double settle_s[3] = {0, 0, 0};  // giving the excess back: un-registering, (unused), the cut itself
    void settle(bool run_is_over) {
        if (fd < 0 || !map) return;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        const auto t0 = now();
        if (run_is_over && pinned) {  // (on a fatal exit other threads may still be looking: nothing is touched but the file)
            (void)spx_host_unregister(map);
            pinned = false;
        }
        const auto t1 = now();
        // (cutting N mapped pages off a file is two jobs under the inode's lock, the page table entries and the pages; dropping
        // the entries ahead with madvise(DONTNEED) on several threads made it slower on the GPU box, 45 -> 70-160 ms:
        // profiles/r05_early_trim_experiment.txt)
        const auto t2 = now();
        std::lock_guard<std::mutex> g(cut_mu);
        settled = true;
        if (::ftruncate(fd, (off_t)end) != 0) std::fprintf(stderr, "[spumoni-gpu] could not cut the output file to its size\n");
        settle_s[0] = secs(t0, t1);
        settle_s[1] = secs(t1, t2);
        settle_s[2] = secs(t2, now());
        // (the mapping itself is left to the end of the process: unmapping 2 GB is 75 ms that no one waits for there)
    }
    ~OutFile() {
        if (fd >= 0) ::close(fd);
    }
};

enum { F_LENGTHS = 0, F_POINTERS = 1, F_DOCS = 2, F_REPORT = 3, NFILES = 4 };
}  // namespace
struct OutputFiles {
    OutFile f[NFILES];  // .pseudo_lengths | .lengths, .pointers, .doc_numbers, .report
    double prepare_s = 0, prep_s[3] = {0, 0, 0};  // preparing the tails: in all; fallocate + mmap, page table entries, page-locking
};
namespace {
using Outputs = OutputFiles;

// Super-batches take their place in the output files in input order: reserve() blocks until every earlier super-batch
// has its place, then hands out this one's offsets.  (The device workers take super-batches strictly in input order, so the
// one that is waited for is always in some worker's hands.)
class OffsetOrder {
public:
    // input: the bytes of the reads file this super-batch stands for
    void reserve(uint64_t seq, Outputs& out, const uint64_t bytes[NFILES], uint64_t off[NFILES], uint64_t input) {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return next_ == seq; });
        for (int f = 0; f < NFILES; ++f) {
            off[f] = out.f[f].reserved;
            out.f[f].reserved += bytes[f];
        }
        consumed_ += input;
        if (input >= 4096)  // (the densest super-batch so far, per file: how far a later one may lie above the mean)
            for (int f = 0; f < NFILES; ++f) densest_[f] = std::max(densest_[f], (double)bytes[f] / (double)input);
        next_++;
        cv_.notify_all();
    }
    // the places handed out so far and the input they stand for, as of one moment; densest[f]: the most output bytes per input
    // byte any one super-batch has asked for
    void snapshot(const Outputs& out, uint64_t used[NFILES], uint64_t& consumed, double densest[NFILES]) {
        std::lock_guard<std::mutex> g(mu_);
        for (int f = 0; f < NFILES; ++f) used[f] = out.f[f].reserved, densest[f] = densest_[f];
        consumed = consumed_;
    }
    // file f's writable mapping ends at `target` from now on -- unless a place beyond it was handed out already
    bool shrink(Outputs& out, int f, uint64_t target) {
        std::lock_guard<std::mutex> g(mu_);
        if (out.f[f].reserved > target || target >= out.f[f].map_size.load()) return false;
        out.f[f].map_size.store(target);
        return true;
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    uint64_t next_ = 0, consumed_ = 0;
    double densest_[NFILES] = {0, 0, 0, 0};
};

// The prepared tails are sized from an estimate on the generous side, and what is too much has to be given back: cutting
// 390 MB of allocated, mapped, page-locked pages off the 4e6-read run's .pseudo_lengths took 50 ms AFTER the last byte was in
// place -- of a 131 ms run.  A quarter of the way through the input the places handed out predict the final sizes to a few percent, so a
// helper cuts the excess off THEN, beside the run (ftruncate takes the file's inode lock, which nothing else wants: the text
// arrives through the mapping): the mapping's writable end moves down first, under the lock the places are handed out under
// and only if no place beyond it has been handed out, so that later super-batches which turn out to need more take the plain
// way (writer thread) behind the cut.  What is left for the end is the last few percent.  Only the part of a tail that is NOT
// registered with the device is ever cut this way (OutFile::pinned_size, pin_share()): cutting registered pages off under live
// queues hung the run (profiles/r05_early_trim_experiment.txt).
class EarlyTrim {
public:
    EarlyTrim(OutputFiles& out, OffsetOrder& order, uint64_t input_bytes) : out_(out), order_(order), input_bytes_(std::max<uint64_t>(input_bytes, 1)) {
        if (const char* e = std::getenv("SPUMONI_TRIM_MIN")) min_ = std::strtoull(e, nullptr, 10);  // (tests: 0 -- tiny files are trimmed too)
        th_ = std::thread([this] { loop(); });
    }
    ~EarlyTrim() { finish(); }
    void finish() {  // (joined: seconds() / trimmed_bytes() are final after this)
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    void poke() {
        {
            std::lock_guard<std::mutex> g(mu_);
            ++pokes_;
        }
        cv_.notify_all();
    }
    double seconds() const { return seconds_; }
    uint64_t trimmed_bytes() const { return trimmed_; }

private:
    void loop() {
        std::unique_lock<std::mutex> g(mu_);
        uint64_t seen = 0;
        int stage[NFILES] = {0, 0, 0, 0};  // 0: untouched, 1: cut at a sixth of the input, 2: cut again at 70 %
        while (!stop_) {
            cv_.wait(g, [&] { return stop_ || pokes_ != seen; });  // (no wait_for: gcc 11's TSan does not know pthread_cond_clockwait)
            if (stop_) break;
            seen = pokes_;
            g.unlock();
            uint64_t used[NFILES], consumed = 0;
            double densest[NFILES];
            order_.snapshot(out_, used, consumed, densest);
            const double share = std::min(1.0, (double)consumed / (double)input_bytes_);
            // (the files side by side: every inode has its own lock, and MS mode has two large ones to cut)
            std::vector<std::thread> cuts;
            std::mutex sum_mu;
            for (int f = 0; f < NFILES; ++f) {
                OutFile& of = out_.f[f];
                if (of.fd < 0 || !of.map) continue;
                // (a sixth of the input in: the cut of a few hundred MB takes 40-80 ms and should be over before the run is --
                // started at half the input it ended 10-20 ms AFTER the last super-batch, and the run waited for it; the second,
                // small cut at 70 %)
                const int want_stage = share >= 0.7 ? 2 : (share >= 0.16 ? 1 : 0);
                if (want_stage <= stage[f]) continue;
                stage[f] = want_stage;
                // the predicted final size + a margin + 8 MB, on a page boundary.  The margin follows what the super-batches so far
                // say about their spread: 1.5 % (0.8 % the second time) when every one asked for the same bytes per input byte,
                // more by how far the densest lay above the mean, 10 % (5 %) at most -- what it leaves is cut after the run
                const double predicted = (double)used[f] / share;
                const double mean = consumed ? (double)used[f] / (double)consumed : 0.0;
                const double spread = mean > 0 ? std::max(0.0, densest[f] / mean - 1.0) : 0.1;
                const double margin = want_stage == 2 ? std::min(1.05, 1.008 + spread) : std::min(1.10, 1.015 + 1.5 * spread);
                // (never into the part that is registered with the device: pinned_size)
                const uint64_t target = std::max<uint64_t>(((uint64_t)(predicted * margin) + min_ / 4 + 4095) & ~4095ull, (of.pinned_size + 4095) & ~4095ull);
                const uint64_t before = of.map_size.load();
                if (target + min_ >= before) continue;  // (nothing worth a system call)
                cuts.emplace_back([this, f, target, before, &sum_mu] {
                    OutFile& of = out_.f[f];
                    // (cut_mu: a later super-batch that needs more than the cut leaves goes through the file's writer thread, which
                    // must not write behind the new end before the cut has happened; and a fatal exit settles the file once, for good)
                    std::lock_guard<std::mutex> cut(of.cut_mu);
                    if (of.settled || !order_.shrink(out_, f, target)) return;
                    const auto t0 = std::chrono::steady_clock::now();
                    if (::ftruncate(of.fd, (off_t)target) != 0) { /* (the file keeps its excess until the end) */ }
                    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    std::lock_guard<std::mutex> sg(sum_mu);
                    seconds_ += dt;  // (summed over the files: they are cut side by side)
                    trimmed_ += before - target;
                });
            }
            for (auto& t : cuts) t.join();
            g.lock();
        }
    }
    OutputFiles& out_;
    OffsetOrder& order_;
    const uint64_t input_bytes_;
    std::thread th_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
    uint64_t pokes_ = 0, trimmed_ = 0, min_ = 32u << 20;
    double seconds_ = 0;
};

// Formats reads [lo, hi) of a super-batch into text; run by several host threads at once
// (the text of 10^6 reads x 200 values is ~0.5 GB: at GPU speed, formatting is the job).
struct TextChunk {
    TextBuf tl, tp, td;
    std::string report;
};

// "%-26.3g" of sum / nbins the way the reference's stream prints it (setw(26) << left, precision 3, default float
// format): the common case -- an average that is a whole number below 1000, e.g. every read of up to one bin -- needs
// no printf (4 * 10^6 snprintf("%.3g") calls were 0.1 s of sixteen cores); everything else goes through it.
static inline int format_avg(char* p, uint64_t sum, size_t nbins) {
    if (nbins != 0 && sum % nbins == 0 && sum / nbins < 1000) {
        uint32_t v = (uint32_t)(sum / nbins);
        int k = 0;
        if (v >= 100) p[k++] = (char)('0' + v / 100);
        if (v >= 10) p[k++] = (char)('0' + (v / 10) % 10);
        p[k++] = (char)('0' + v % 10);
        while (k < 26) p[k++] = ' ';
        return k;
    }
    return std::snprintf(p, 64, "%-26.3g", (sum + 0.0) / nbins);
}
static inline int format_count12(char* p, size_t v) {  // "%-12zu"
    char tmp[24];
    int q = 24;
    do {
        tmp[--q] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    int k = 24 - q;
    std::memcpy(p, tmp + q, (size_t)k);
    while (k < 12) p[k++] = ' ';
    return k;
}

// report_to: where the report lines of [lo, hi) go when their place in the report file is memory (every line is
// max(30, id) + 66 bytes, so the place is known before the line is); null: collected in out.report
void format_range(const RunOptions& o, const SuperBatch& sb, const Results& res, size_t lo, size_t hi,
                  TextChunk& out, char* report_to = nullptr) {
    out.report.clear();
    out.tl.clear();
    out.tp.clear();
    out.td.clear();
    const bool values = o.use_doc || !o.report_only || o.ms;
    for (size_t q = lo; q < hi; ++q) {
        const uint64_t a = values ? res.beg[q] : 0, b = values ? res.end[q] : 0;
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
            // = %.3g) << setw(12) << above << setw(12) << below: the same bytes without a stream (an ostringstream
            // per thread spent 0.25 s on 4*10^6 such lines, snprintf 0.1 s)
            const std::string_view id = sb.ids[q];
            char line[192];
            int k = 0;
            if (read_found) {
                std::memcpy(line, "FOUND          ", 15);
            } else {
                std::memcpy(line, "NOT_PRESENT    ", 15);
            }
            k = 15;
            k += format_avg(line + k, c.sum_max_bin_values, nbins);
            k += format_count12(line + k, (size_t)c.bins_above);
            k += format_count12(line + k, (size_t)c.bins_below);
            line[k++] = '\n';
            if (report_to) {
                std::memcpy(report_to, id.data(), id.size());
                report_to += id.size();
                if (id.size() < 30) {
                    std::memset(report_to, ' ', 30 - id.size());
                    report_to += 30 - id.size();
                }
                std::memcpy(report_to, line, (size_t)k);
                report_to += k;
            } else {
                out.report.append(id.data(), id.size());
                if (id.size() < 30) out.report.append(30 - id.size(), ' ');
                out.report.append(line, (size_t)k);
            }
        }
    }
}

size_t open_outputs_and_threshold(Outputs& out, const RunOptions& o) {
    out.f[F_LENGTHS].open(o.pattern_file + (o.ms ? ".lengths" : ".pseudo_lengths"));
    if (o.ms) out.f[F_POINTERS].open(o.pattern_file + ".pointers");
    if (o.use_doc) out.f[F_DOCS].open(o.pattern_file + ".doc_numbers");
    if (o.write_report) out.f[F_REPORT].open(o.pattern_file + ".report");
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
        out.f[F_REPORT].append(hd.str());
    }
    return max_value_thr;
}

// ---- a persistent pool -----------------------------------------------------------------------------------------------------
// Round 4 created and joined its helper threads per phase and super-batch (parse, assemble, headers, report: some
// 800 thread starts in a 0.4 s run) and ran one super-batch's phases strictly one after the other.  The pool's threads
// live as long as the run; run(n, f) hands f(0) .. f(n - 1) to them AND to the caller, returns when all are done, and may
// be called from several threads at once (the feeders and the device workers): their items simply share the threads.
class Pool {
public:
    explicit Pool(size_t threads) {
        for (size_t t = 0; t < threads; ++t) th_.emplace_back([this] { worker(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    size_t size() const { return th_.size() + 1; }
    template <class F>
    void run(size_t n, F&& f) {
        if (n == 0) return;
        if (n == 1 || th_.empty()) {
            for (size_t i = 0; i < n; ++i) f(i);
            return;
        }
        auto job = std::make_shared<Job>();
        job->n = n;
        job->f = [&f](size_t i) { f(i); };
        {
            std::lock_guard<std::mutex> g(mu_);
            jobs_.push_back(job);
        }
        cv_.notify_all();
        work(*job);
        std::unique_lock<std::mutex> g(job->mu);
        job->cv.wait(g, [&] { return job->done == job->n; });
    }

private:
    struct Job {
        std::function<void(size_t)> f;
        size_t n = 0;
        std::atomic<size_t> next{0};
        std::mutex mu;
        std::condition_variable cv;
        size_t done = 0;
    };
    static void work(Job& j) {
        size_t mine = 0;
        for (;;) {
            const size_t i = j.next.fetch_add(1);
            if (i >= j.n) break;
            j.f(i);
            ++mine;
        }
        if (mine) {
            std::lock_guard<std::mutex> g(j.mu);
            j.done += mine;
            if (j.done == j.n) j.cv.notify_all();
        }
    }
    void worker() {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> g(mu_);
                for (;;) {
                    while (!jobs_.empty() && jobs_.front()->next.load() >= jobs_.front()->n) jobs_.pop_front();
                    if (stop_ || !jobs_.empty()) break;
                    cv_.wait(g);
                }
                if (jobs_.empty()) return;  // (stop_)
                j = jobs_.front();
            }
            work(*j);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Job>> jobs_;
    bool stop_ = false;
    std::vector<std::thread> th_;
};

struct Piece {
    const char* p;
    size_t n;
};

// bytes of a read's report line (compute_ms_pml.cpp:1012-1020: setw(30) id, setw(15) status, setw(26) average, two
// setw(12) counts, endl): known from the id alone, which is what lets the lines be formatted at their place in the file
static inline uint64_t report_line_bytes(size_t id_len) { return std::max<size_t>(30, id_len) + 66; }

// After the device: what every output file gets from this super-batch.  Text from the device (spx_text.hip): the ">id\n"
// lines are dropped into their gaps and the report lines formatted, by the pool -- in the file itself where the text
// landed there, in the slot's buffers otherwise (a piece for the file's writer thread); values from the device
// (SPUMONI_HOST_FORMAT, report-only PML): everything is formatted here.  file_off: the super-batch's place in every file
// (device-text runs).
void finish_batch(Pool& pool, const RunOptions& o, Outputs& out, const SuperBatch& sb, const Results& res,
                  std::vector<TextChunk>& chunks, const uint64_t file_off[NFILES], std::vector<Piece> pieces[NFILES]) {
    for (int i = 0; i < NFILES; ++i) pieces[i].clear();
    const size_t nreads = sb.nreads();
    if (nreads == 0) return;
    const size_t nt = std::max<size_t>(1, std::min<size_t>(pool.size(), (nreads + 2047) / 2048));
    if (chunks.size() < nt) chunks.resize(nt);
    // the report's lines go straight to their place when that is memory
    char* report_base = nullptr;
    std::vector<uint64_t> rep_off;
    if (o.write_report && out.f[F_REPORT].map) {
        rep_off.assign(nt + 1, 0);
        pool.run(nt, [&](size_t t) {
            uint64_t b = 0;
            for (size_t q = nreads * t / nt; q < nreads * (t + 1) / nt; ++q) b += report_line_bytes(sb.ids[q].size());
            rep_off[t + 1] = b;
        });
        for (size_t t = 0; t < nt; ++t) rep_off[t + 1] += rep_off[t];
        if (file_off[F_REPORT] + rep_off[nt] <= out.f[F_REPORT].map_size) report_base = out.f[F_REPORT].map + file_off[F_REPORT];
    }
    if (res.device_text) {
        RunOptions ro = o;  // only the report is left to format
        ro.use_doc = false;
        ro.ms = false;
        ro.report_only = true;
        bool open_[3], direct[3];
        char* dest[3] = {nullptr, nullptr, nullptr};  // the stream's place in its file, when that is memory
        for (int i = 0; i < 3; ++i) {
            open_[i] = (res.streams & (1u << i)) && out.f[i].is_open();
            const uint64_t bytes = open_[i] ? res.line_start[i][nreads] : 0;
            const bool fits = open_[i] && out.f[i].map && file_off[i] + bytes <= out.f[i].map_size;
            if (open_[i] && out.f[i].map && !fits) {
                // (the tail was prepared from an estimate -- bytes per value by mode, prepare_outputs -- and the run has used it up:
                // said once per file, so that a long-read run that silently takes the slower way can be seen: ADVICE r5)
                static std::atomic<unsigned> told{0};
                if (!(told.fetch_or(1u << i) & (1u << i)))
                    std::fprintf(stderr, "[spumoni-gpu] note: the prepared tail of output stream %d (%.1f MB, an estimate) is used up after %.1f MB: "
                                         "the rest is written the ordinary way\n", i, (double)out.f[i].mapped_bytes / 1e6, (double)file_off[i] / 1e6);
            }
            dest[i] = fits ? out.f[i].map + file_off[i] : nullptr;
            direct[i] = fits && res.text_at[i] == dest[i];  // the device wrote it there
        }
        pool.run(nt, [&](size_t t) {
            const size_t lo = nreads * t / nt, hi = nreads * (t + 1) / nt;
            for (int i = 0; i < 3; ++i) {
                if (!open_[i]) continue;
                const uint64_t* ls = res.line_start[i].data();
                char* base = res.text_at[i];
                if (dest[i] && !direct[i]) {
                    // mapped but not page-locked for the device (or the place was not known in time): this part's lines
                    // are copied into the file here -- the parts are whole lines, so nobody waits for anybody
                    std::memcpy(dest[i] + ls[lo], base + ls[lo], (size_t)(ls[hi] - ls[lo]));
                    base = dest[i];
                }
                for (size_t q = lo; q < hi; ++q) {
                    char* p = base + ls[q];
                    const std::string_view id = sb.ids[q];
                    *p++ = '>';
                    std::memcpy(p, id.data(), id.size());
                    p[id.size()] = '\n';
                }
            }
            format_range(ro, sb, res, lo, hi, chunks[t], report_base ? report_base + rep_off[t] : nullptr);
        });
        for (int i = 0; i < 3; ++i)
            if (open_[i] && !dest[i]) pieces[i].push_back(Piece{res.text_at[i], (size_t)res.line_start[i][nreads]});
    } else {
        pool.run(nt, [&](size_t t) {
            format_range(o, sb, res, nreads * t / nt, nreads * (t + 1) / nt, chunks[t], report_base ? report_base + rep_off[t] : nullptr);
        });
        for (size_t t = 0; t < nt; ++t) {
            const TextChunk& c = chunks[t];
            if (c.tl.len) pieces[F_LENGTHS].push_back(Piece{c.tl.buf.data(), c.tl.len});
            if (o.ms && c.tp.len) pieces[F_POINTERS].push_back(Piece{c.tp.buf.data(), c.tp.len});
            if (o.use_doc && c.td.len) pieces[F_DOCS].push_back(Piece{c.td.buf.data(), c.td.len});
        }
    }
    if (o.write_report && !report_base)
        for (size_t t = 0; t < nt; ++t)
            if (!chunks[t].report.empty()) pieces[F_REPORT].push_back(Piece{chunks[t].report.data(), chunks[t].report.size()});
}

}  // namespace

namespace {

// A super-batch in flight.
struct Slot {
    SuperBatch sb;
    Results res;
    uint64_t seq = 0;            // position of this super-batch in the input (results are written in this order)
    bool last = false;           // no more input after this one
    std::vector<std::vector<ReadRec>> recs;  // the feeder's reads, per part (kept: their capacity is reused)
    std::vector<TextChunk> chunks;           // per part: the report lines (host-formatted: the values lines too)
    std::vector<Piece> pieces[NFILES];       // what each file's writer copies into the file for this super-batch
    uint64_t file_off[NFILES] = {};          // device-text runs: the super-batch's place in every file (OffsetOrder)
    uint64_t file_bytes[NFILES] = {};        // ... and its size there
    bool placed = false;
    uint64_t report_bytes = 0;               // bytes of its report lines (known from the ids)
    uint64_t input_bytes = 0;                // bytes of the reads file it stands for (lines and their newlines)
    std::atomic<int> writers_left{0};
    int deferred = 0;            // 0 none, 1 FATAL_ERROR, 2 "empty after digestion" FATAL_WARNING
    std::string deferred_msg;
};

// blocking hand-off of slot indices between the stages
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
        q_.pop_front();
        return v;
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<int> q_;
};

// Re-sequences the super-batches the device workers finish, in whatever order, into input order -- for several readers
// (one writer thread per output file): look() blocks until super-batch `seq` is done and leaves the entry where it is,
// drop() takes it out once every writer is through with it.
class OrderedDone {
public:
    void put(uint64_t seq, int slot) {
        std::lock_guard<std::mutex> g(mu_);
        done_.emplace_back(seq, slot);
        cv_.notify_all();
    }
    int look(uint64_t seq) {
        std::unique_lock<std::mutex> g(mu_);
        for (;;) {
            if (seq > last_) return -1;
            for (const auto& e : done_)
                if (e.first == seq) return e.second;
            cv_.wait(g);
        }
    }
    void set_last(uint64_t seq) {  // nothing after super-batch `seq`: look() for a later one returns -1
        std::lock_guard<std::mutex> g(mu_);
        last_ = std::min(last_, seq);
        cv_.notify_all();
    }
    void drop(uint64_t seq) {
        std::lock_guard<std::mutex> g(mu_);
        for (size_t i = 0; i < done_.size(); ++i)
            if (done_[i].first == seq) {
                done_.erase(done_.begin() + (long)i);
                return;
            }
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::pair<uint64_t, int>> done_;
    uint64_t last_ = ~0ull;
};

// Fills `slot` from the ranges of one super-batch (the reference's ~1000-base batches, cut sequentially by the caller):
// the pool scans them into reads (positions in the mapped file), every part counts its own reads and characters -- no
// serial pass over the reads --, and the same parts then copy the reads, upper-cased, into the page-locked buffer and
// fill in ids, offsets and header sizes.  A malformed / empty read truncates the super-batch there and is reported after
// everything before it has been written, like the reference running read by read.
void fill_slot(Pool& pool, const ReadFile& input, const std::vector<ReadFile::Range>& ranges, Slot& slot, double t_phase[2]) {
    const auto tick = [] { return std::chrono::steady_clock::now(); };
    const auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    auto t0 = tick();
    slot.sb.clear();
    slot.deferred = 0;
    slot.placed = false;
    slot.report_bytes = 0;
    const size_t nt = std::max<size_t>(1, std::min<size_t>(pool.size(), (ranges.size() + 63) / 64));
    if (slot.recs.size() < nt) slot.recs.resize(nt);
    struct Part {
        size_t take = 0, chars = 0, longest = 0, report = 0;
        int problem = 0;  // 2: read `take` is empty; 1: a malformed record follows the part's reads
        ReadFile::ParseError err;
    };
    std::vector<Part> part(nt);
    pool.run(nt, [&](size_t t) {
        std::vector<ReadRec>& v = slot.recs[t];
        v.clear();
        Part& pt = part[t];
        const size_t lo = ranges.size() * t / nt, hi = ranges.size() * (t + 1) / nt;
        for (size_t i = lo; i < hi && !pt.err.fatal; ++i) input.scan_range(ranges[i], v, pt.err);
        for (const ReadRec& rd : v) {
            if (rd.seq_len == 0) {  // :926-931
                pt.problem = 2;
                break;
            }
            pt.take++;
            pt.chars += rd.seq_len;
            pt.longest = std::max<size_t>(pt.longest, rd.seq_len);
            pt.report += report_line_bytes(rd.id_len);
        }
        if (!pt.problem && pt.err.fatal) pt.problem = 1;
    });
    // where does every part's share go, and where is the first problem (if any)?
    std::vector<size_t> r0(nt + 1, 0), c0(nt + 1, 0);
    for (size_t t = 0; t < nt; ++t) {
        if (slot.deferred) part[t].take = part[t].chars = part[t].longest = part[t].report = 0;  // (behind the problem: not part of the run)
        slot.report_bytes += part[t].report;
        r0[t + 1] = r0[t] + part[t].take;
        c0[t + 1] = c0[t] + part[t].chars;
        slot.sb.longest = std::max<uint64_t>(slot.sb.longest, part[t].longest);
        if (!slot.deferred && part[t].problem) {
            slot.deferred = part[t].problem;
            slot.deferred_msg = part[t].problem == 2 ? std::string(slot.recs[t][part[t].take].id, slot.recs[t][part[t].take].id_len)
                                                     : part[t].err.message;
        }
    }
    if (slot.deferred) slot.last = true;
    const size_t nreads = r0[nt], nchars = c0[nt];
    slot.sb.ids.resize(nreads);
    slot.sb.offs.resize_uninit(nreads + 1);
    slot.sb.offs[0] = 0;
    slot.sb.gap.resize_uninit(nreads);
    slot.sb.seqs.resize_uninit(nchars);
    t_phase[0] += since(t0);
    t0 = tick();
    pool.run(nt, [&](size_t t) {
        size_t rdx = r0[t], cpos = c0[t];
        const std::vector<ReadRec>& v = slot.recs[t];
        for (size_t q = 0; q < part[t].take; ++q) {
            const ReadRec& rd = v[q];
            input.copy_seq_upper(rd, slot.sb.seqs.data() + cpos);  // all characters upper-case (:916-917)
            cpos += rd.seq_len;
            slot.sb.offs[rdx + 1] = cpos;
            slot.sb.ids[rdx] = std::string_view(rd.id, rd.id_len);
            slot.sb.gap[rdx] = rd.id_len + 2;
            rdx++;
        }
    });
    t_phase[1] += since(t0);
}

}  // namespace

// Called on a helper thread while the index loads: the page-locked blocks the slots of classify_reads will ask for
// (per slot: the reads of a super-batch with their offsets and header sizes, the class records, and per output stream
// its text and its record offsets).
bool outputs_can_be_mapped(const RunOptions& o);
// feeders: two for a device's two workers, one more per further device (eight at most): SURVEY 8(e)'s "one feeder thread group
// per GPU" -- the groups share the pool, whose size follows the devices as well (spumoni_main.cpp)
static size_t feeders_for(size_t nworkers) {
    static const int forced = std::getenv("SPUMONI_FEEDERS") ? std::atoi(std::getenv("SPUMONI_FEEDERS")) : 0;  // (A/B runs)
    if (forced > 0) return (size_t)std::min(forced, 8);
    return std::max<size_t>(2, std::min<size_t>(8, nworkers / 2 + 1));
}
static size_t slots_for(size_t nworkers) { return nworkers + feeders_for(nworkers) + 2; }  // one per worker, one per feeder, two being written
static double pin_share();
static uint64_t split_min_bytes();
static uint64_t mem_available();
// ONE budget -- a third of what the machine has available when the run starts -- for everything that is prepared ahead of the
// run and mostly page-locked: the value streams' tails, the report's, the pinned pool.  (Until round 6 each of the three asked
// /proc/meminfo on its own, while the others were allocating beside it: their sum could exceed what was there.  ADVICE r5.)
static bool claim_prepared_memory(uint64_t bytes) {
    static std::atomic<int64_t> left{[] {
        const uint64_t a = mem_available();
        return a == ~0ull ? INT64_MAX : (int64_t)(a / 3);
    }()};
    if (bytes > (uint64_t)INT64_MAX / 2) return false;
    if (left.fetch_sub((int64_t)bytes) - (int64_t)bytes >= 0) return true;
    left.fetch_add((int64_t)bytes);
    return false;
}
void prepare_pinned_pool(const RunOptions& o, size_t nworkers) {
    if (spx_device_count() <= 0) return;
    // sized from the reads file, not from the super-batch limit alone: a small file needs small blocks and few
    // slots, and below a megabyte locking pages ahead of time buys nothing
    struct stat sb;
    if (::stat(o.pattern_file.c_str(), &sb) != 0 || sb.st_size < (1 << 20)) return;
    const size_t fsize = (size_t)sb.st_size;
    const size_t batches = fsize / std::max<size_t>(o.super_batch_chars, 1) + 1;
    const size_t nslots = std::min<size_t>(slots_for(std::max<size_t>(nworkers, 1)), batches + 1);
    const bool report_only = o.report_only && !o.ms && o.write_report;
    const size_t chars = std::min<size_t>(o.super_batch_chars, fsize) + std::min<size_t>(4u << 20, fsize / 8 + 4096);
    std::vector<size_t> sizes;
    for (size_t i = 0; i < nslots; ++i) {
        const size_t reads_guess = chars / 100 + 4096;
        sizes.push_back(chars);  // reads
        sizes.push_back((reads_guess + 1) * 8);  // their offsets
        sizes.push_back(reads_guess * 4);        // their header sizes
        if (o.write_report) sizes.push_back(reads_guess * sizeof(spx_class));
        if (std::getenv("SPUMONI_HOST_FORMAT")) continue;
        // (the streams' record offsets always; their text only when it will not land in the files themselves)
        const bool text_staged = !outputs_can_be_mapped(o);
        // (... or lands in the part of a file's tail that is not registered with the device: lengths and document ids of large runs)
        const bool upper_part = !text_staged && pin_share() < 1.0 && (double)fsize * 2.2 >= (double)split_min_bytes();
        if (!report_only) {  // lengths: "<value> " is 2-4 bytes for most values
            if (text_staged || upper_part) sizes.push_back(chars * 3 + std::min<size_t>(8u << 20, chars));
            sizes.push_back((reads_guess + 1) * 8);
        }
        if (o.ms) {  // pointers: up to 13 digits (MS runs take 8 MB super-batches by default: 96 MB per slot)
            if (text_staged || upper_part) sizes.push_back(chars * 12);
            sizes.push_back((reads_guess + 1) * 8);
        }
        if (o.use_doc) {
            if (text_staged || upper_part) sizes.push_back(chars * 3);
            sizes.push_back((reads_guess + 1) * 8);
        }
    }
    {
        uint64_t all = 0;
        for (size_t b : sizes) all += b;
        if (!claim_prepared_memory(all)) return;  // (the slots then allocate what they need themselves)
    }
    for (size_t b : sizes) {
        void* p = spx_host_alloc(b);
        if (!p) return;  // (the slots then allocate what they need themselves)
        g_pinned_pool.put(p, b);
    }
}

// ---- the output files' tails as memory, prepared while the index loads ----------------------------------------------------
namespace {
OutputFiles* g_live_outputs = nullptr;  // what a fatal exit has to settle (reads.cpp: leave -> the exit hook)
std::mutex g_settle_mu;
// A fatal read ends the run while the other threads -- the device's copies, the pool -- may still be writing later
// super-batches into the files' mapped tails: cutting the files to their logical ends turns those stores into SIGBUS.  The
// process is on its way out; a thread that gets there simply stays there.
extern "C" void park_on_sigbus(int) {
    for (;;) ::pause();
}
void settle_outputs(bool run_is_over) {
    std::lock_guard<std::mutex> g(g_settle_mu);
    if (!g_live_outputs) return;
    if (!run_is_over) {  // (the fatal way out only: at the ordinary end every writer has been joined and a SIGBUS is a real one)
        struct sigaction sa;
        std::memset(&sa, 0, sizeof sa);
        sa.sa_handler = park_on_sigbus;
        sigemptyset(&sa.sa_mask);
        (void)::sigaction(SIGBUS, &sa, nullptr);
    }
    if (run_is_over) {  // (the files side by side: every inode has its own lock)
        std::vector<std::thread> th;
        for (OutFile& f : g_live_outputs->f)
            if (f.fd >= 0 && f.map) th.emplace_back([&f] { f.settle(true); });
        for (auto& x : th) x.join();
    } else {
        for (OutFile& f : g_live_outputs->f) f.settle(false);
    }
    g_live_outputs = nullptr;
}
void settle_outputs_at_exit() { settle_outputs(false); }
bool env_is(const char* name, const char* value) {
    const char* e = std::getenv(name);
    return e && std::strcmp(e, value) == 0;
}
}  // namespace

bool outputs_can_be_mapped(const RunOptions& o) {
    // only the path that produces the files' text on the device lands it in the files (SPUMONI_MAP_OUTPUT=0: never)
    return !(o.is_general_text || std::getenv("SPUMONI_HOST_FORMAT") || env_is("SPUMONI_MAP_OUTPUT", "0"));
}

// One file's tail as memory: created under a temporary name, `size` bytes allocated (fallocate), mapped, its page table
// entries made by a few threads, and -- the value streams -- page-locked for the device.
static double pin_share() {
    // The share of a value stream's prepared tail that is registered with the device.  What is registered cannot be given
    // back while the device works (profiles/r05_early_trim_experiment.txt: the run hangs), and giving back the estimate's
    // excess AFTER the run was a third of it; so the upper part of the tail is mapped and populated but not registered --
    // super-batches that land there go through a page-locked buffer and the pool's memcpy (the `nopin` way), and EarlyTrim
    // cuts what the run will not need while it runs.  1: everything registered, nothing cut early (round 5's first form).
    if (const char* e = std::getenv("SPUMONI_PIN_SHARE")) return std::min(1.0, std::max(0.0, std::atof(e)));
    return 0.7;
}

static void prepare_one(OutputFiles* out, int f, const std::string& final_path, uint64_t est, bool pin, double share = 1.0) {
    const auto tick = [] { return std::chrono::steady_clock::now(); };
    const auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    OutFile& of = out->f[f];
    remove_stale_leftovers(final_path);  // (what a killed run of the same pattern file left behind)
    of.temp_path = final_path + ".partial." + std::to_string((long)::getpid());
    // (a name nobody else can have made: O_EXCL, and no symbolic link planted in a shared directory is followed)
    int fd = ::open(of.temp_path.c_str(), O_RDWR | O_CREAT | O_EXCL | O_NOFOLLOW, 0644);
    if (fd < 0 && errno == EEXIST && ::unlink(of.temp_path.c_str()) == 0)  // (our own pid's name from an earlier life)
        fd = ::open(of.temp_path.c_str(), O_RDWR | O_CREAT | O_EXCL | O_NOFOLLOW, 0644);
    if (fd < 0) return;
    register_leftover(of.temp_path);
    const uint64_t size = (est + 4095) & ~4095ull;
    void* m = MAP_FAILED;
    auto t0 = tick();
    // The file's pages: fallocate makes them (one thread, under the inode's lock: 0.47 s for 8.6 GB on the GPU box), then four
    // threads make the page table entries (0.23 s).  SPUMONI_PREP=populate (round 6; not the default): the
    // file is only given its size and the threads that make the entries make the pages with them (MADV_POPULATE_WRITE on a
    // hole of a tmpfs file allocates and clears the page) -- measured SLOWER the more threads take part, 1.6 / 2.1 / 2.6 / 4.5 s
    // at 4 / 16 / 32 / 64 threads for the same 8.6 GB: tmpfs page allocation does not scale over cores, and one thread inside
    // fallocate is the fastest way to 2 * 10^6 pages (profiles/r06_cli_output_preparation.txt).  Either way a file system
    // that cannot give the estimate shows up here -- fallocate / populate fail -- not as a SIGBUS inside the run, and the file
    // is written the ordinary way.
    static const bool by_populate = [] {
        const char* e = std::getenv("SPUMONI_PREP");
        return e && std::strcmp(e, "populate") == 0;
    }();
    static const unsigned prep_threads = [] {
        const char* e = std::getenv("SPUMONI_PREP_THREADS");
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        return e ? std::max(1, std::atoi(e)) : (int)std::min(16u, std::max(4u, hw / 2));
    }();
    // (Also measured and withdrawn, round 6: the two steps OVERLAPPED -- fallocate by 256 MB pieces on one thread, the four populate
    // threads behind it on the pieces that have their pages.  fallocate then takes 2.5-2.8 s instead of 0.67 s for 9.5 GB: the
    // threads contend for the file's page cache tree.  One after the other it is.)
    const bool sized = by_populate ? ::ftruncate(fd, (off_t)size) == 0 : ::fallocate(fd, 0, 0, (off_t)size) == 0;
    if (sized) m = ::mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    auto give_up = [&] {  // (a file system that cannot do it, or is full: the file is written the ordinary way)
        if (m != MAP_FAILED) ::munmap(m, size);
        if (::ftruncate(fd, 0) != 0) { /* (still empty then) */ }
        of.fd = fd;
    };
    if (m == MAP_FAILED) return give_up();
    const double s0 = since(t0);
    t0 = tick();
    {
        // the page table entries now, by a few threads, so that nothing faults inside the run
        const unsigned nt = by_populate ? prep_threads : 4;
        std::atomic<bool> failed{false};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([=, &failed] {
                const uint64_t lo = (size / 4096 * t / nt) * 4096, hi = t + 1 == nt ? size : (size / 4096 * (t + 1) / nt) * 4096;
#ifdef MADV_POPULATE_WRITE
                // (by pieces: a thread that finds the file system full stops, and the others with it)
                const uint64_t piece = 64ull << 20;
                uint64_t a = lo;
                for (; a < hi && !failed.load(std::memory_order_relaxed); a += piece)
                    if (::madvise((char*)m + a, std::min(piece, hi - a), MADV_POPULATE_WRITE) != 0) break;
                if (a >= hi) return;
                if (failed.load(std::memory_order_relaxed)) return;
                if (by_populate || errno != EINVAL) {  // (EINVAL: a kernel without MADV_POPULATE_WRITE -- touch the pages instead)
                    failed.store(true);
                    return;
                }
#endif
                if (by_populate) {  // (touching a hole of a full file system would be a SIGBUS)
                    failed.store(true);
                    return;
                }
                for (uint64_t a2 = lo; a2 < hi; a2 += 4096) {
                    volatile char* p = (volatile char*)m + a2;
                    *p = *p;
                }
            });
        for (auto& x : th) x.join();
        if (failed.load()) return give_up();
    }
    const double s1 = since(t0);
    t0 = tick();
    of.fd = fd;
    of.map = (char*)m;
    of.mapped_bytes = size;
    of.map_size = size;
    if (pin && !env_is("SPUMONI_MAP_OUTPUT", "nopin")) {
        const uint64_t psize = share >= 1.0 ? size : std::min<uint64_t>(size, ((uint64_t)((double)size * share) + 4095) & ~4095ull);
        if (psize > 0 && spx_host_register(m, psize) == SPX_OK) {
            of.pinned = true;
            of.pinned_size = psize;
        }
    }
    std::lock_guard<std::mutex> g(g_settle_mu);
    out->prep_s[0] += s0;
    out->prep_s[1] += s1;
    out->prep_s[2] += since(t0);
}

static uint64_t split_min_bytes() {  // (below this a tail is registered as a whole: there is little to give back)
    if (std::getenv("SPUMONI_MAP_MIN")) return 0;  // (tests: tiny files too)
    return 256u << 20;
}
static uint64_t map_min_bytes() {
    // (SPUMONI_MAP_MIN / SPUMONI_MAP_FACTOR: tests map the tails of tiny files, and size them short so that a run crosses from
    // the prepared tail into plain writes)
    if (const char* e = std::getenv("SPUMONI_MAP_MIN")) return std::strtoull(e, nullptr, 10);
    return 8u << 20;  // (small files: write() is fine)
}
static double map_factor() {
    if (const char* e = std::getenv("SPUMONI_MAP_FACTOR")) return std::max(0.0, std::atof(e));
    return 1.0;
}
static uint64_t mem_available() {
    uint64_t avail = ~0ull;
    std::ifstream mi("/proc/meminfo");
    std::string key, unit;
    uint64_t kb;
    while (mi >> key >> kb >> unit)
        if (key == "MemAvailable:") avail = kb * 1024;
    return avail;
}

// The value streams: sized from the reads file's size alone, so that this can start with the process (nothing of the device
// is needed before the last step).
OutputFiles* new_output_files() { return new OutputFiles; }

void prepare_outputs(OutputFiles* out, const RunOptions& o, uint64_t reads_file_bytes) {
    if (!out || !outputs_can_be_mapped(o)) return;
    const auto t0 = std::chrono::steady_clock::now();
    const bool digest = o.use_promotions || o.use_dna_letters;
    // values per input character: every character without digestion; with it about two minimizers per window of
    // w - k + 1 k-mers (k letters each with -a)
    double v = 1.0;
    if (digest) v = std::min(1.0, 2.2 / (double)(o.w - o.k + 2) * (o.use_dna_letters ? (double)o.k : 1.0));
    double fb = (double)reads_file_bytes;
    {
        // (a FASTQ file is half qualities: batch_loader.cpp:30-38 tells the formats apart by the first character, so does this)
        char first = 0;
        const int rfd = ::open(o.pattern_file.c_str(), O_RDONLY);
        if (rfd >= 0) {
            if (::read(rfd, &first, 1) == 1 && first == '@') fb *= 0.52;
            ::close(rfd);
        }
    }
    // (values grow with the reads: a PML / MS length can have as many digits as the read's length, a 10 kbp read's values are
    // 4-5 digits where a 200 bp read's are 1-3 -- the first read of the file stands for all: ADVICE r5)
    double extra_digits = 0;
    {
        const int rfd = ::open(o.pattern_file.c_str(), O_RDONLY);
        if (rfd >= 0) {
            char head[1 << 16];
            const ssize_t got = ::read(rfd, head, sizeof head);
            ::close(rfd);
            if (got > 0) {
                const char* nl = (const char*)std::memchr(head, '\n', (size_t)got);
                size_t len = 0;
                if (nl) {  // the first record's sequence: up to the next '>' / '+' line (FASTA lines are summed, a FASTQ record has one)
                    for (const char* p = nl + 1; p < head + got && *p != '>' && *p != '+'; ++p) len += *p != '\n' && *p != '\r';
                    if (nl + 1 + len >= head + got - 2) len = std::max<size_t>(len, 60000);  // (did not end inside the window: long)
                }
                for (size_t t = 1000; t <= len; t *= 10) extra_digits += 1.0;
            }
        }
    }
    const double factor = map_factor();
    // bytes per value: lengths "<1-3 digits> ", pointers "<up to 13 digits> ", document ids "<1-3 digits> "; + the ">id" lines
    uint64_t est[3] = {0, 0, 0};
    const bool report_only = o.report_only && !o.ms && o.write_report;
    if (!report_only) est[F_LENGTHS] = (uint64_t)(fb * (0.15 + ((o.ms ? 3.4 : 2.6) + 0.8 * extra_digits) * v) * factor);
    if (o.ms) est[F_POINTERS] = (uint64_t)(fb * (0.15 + 10.5 * v) * factor);
    if (o.use_doc) est[F_DOCS] = (uint64_t)(fb * (0.15 + 2.2 * v) * factor);
    // (never more than a third of what the machine has free -- the budget shared with the report and the pinned pool: the
    // estimate is an upper-ish bound, not a promise)
    if (est[0] + est[1] + est[2] == 0 || !claim_prepared_memory(est[0] + est[1] + est[2])) return;
    static const char* const ext[3] = {nullptr, ".pointers", ".doc_numbers"};
    {   // (the files side by side: every inode has its own lock, and an MS run has three of them)
        std::vector<std::thread> th;
        for (int f = 0; f < 3; ++f) {
            if (est[f] == 0 || est[f] < map_min_bytes()) continue;
            th.emplace_back([out, f, &o, est] {
                prepare_one(out, f, o.pattern_file + (f == F_LENGTHS ? (o.ms ? ".lengths" : ".pseudo_lengths") : ext[f]), est[f], true,
                            est[f] < split_min_bytes() ? 1.0 : pin_share());
            });
        }
        for (auto& x : th) x.join();
    }
    std::lock_guard<std::mutex> g(g_settle_mu);
    out->prepare_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// The report: its size follows from the number of reads, known once the reads file's lines are found.
void prepare_report(OutputFiles* out, const RunOptions& o, uint64_t reads_guess) {
    if (!out || !o.write_report || !outputs_can_be_mapped(o)) return;
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t est = (uint64_t)((double)(reads_guess + 16) * 100.0 * map_factor()) + 256;
    if (est >= map_min_bytes() && claim_prepared_memory(est)) prepare_one(out, F_REPORT, o.pattern_file + ".report", est, false);
    std::lock_guard<std::mutex> g(g_settle_mu);
    out->prepare_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

size_t classify_reads(IndexSet& set, const RunOptions& o, ReadFile* preloaded, OutputFiles* prepared) {
    Outputs& out = prepared ? *prepared : *new OutputFiles;
    const size_t max_value_thr = open_outputs_and_threshold(out, o);
    {
        std::lock_guard<std::mutex> g(g_settle_mu);
        g_live_outputs = &out;
    }
    set_exit_hook(&settle_outputs_at_exit);
    const auto tick = [] { return std::chrono::steady_clock::now(); };
    const auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    // the reads file is mapped and its lines are found while the index loads (spumoni_main.cpp)
    std::unique_ptr<ReadFile> own_input;
    if (!preloaded) {
        own_input.reset(new ReadFile(o.pattern_file, (unsigned)o.format_threads));
        preloaded = own_input.get();
    }
    ReadFile& input = *preloaded;
    // Segmentation (sequential: it defines which reads exist) -> feeders, each parsing a super-batch on the pool -> the
    // device workers, which take the parsed super-batches in input order (SPUMONI_GPUS: one worker per entry; two that name
    // the same device are two query contexts over one copy of the index, so that one's copies run under the other's
    // kernels): [digestion +] walk + text on the device, the super-batch's place in every output file reserved (in input
    // order) as soon as its sizes are known, the text copied by the device into the files' pages, headers and report
    // lines written there by the pool -> one writer thread per output FILE for whatever did not go straight into the file,
    // in input order (the reference's -t 1 order); the last one through with a super-batch makes it final.  Reads are
    // independent (compute_ms_pml.cpp:907-938: nothing is carried from one read to the next), so a slower or busier
    // device simply takes fewer super-batches.  The reference's loop this replaces: compute_ms_pml.cpp:890-1024.
    const size_t nworkers = set.ix.size();
    const int NFEED = (int)feeders_for(nworkers);
    const int NSLOTS = (int)slots_for(nworkers);
    Pool pool(std::max<size_t>(1, o.format_threads) - 1);
    // (the slots outlive the call on purpose: unlocking their page-locked buffers takes a few tenths of a
    // second that the run would spend after its last byte is written; the process ends right after and the
    // operating system takes the pages back)
    std::vector<Slot>& slots = *new std::vector<Slot>((size_t)NSLOTS);
    const auto t_stage0 = tick();
    SlotQueue free_q;
    OrderedDone parsed, done;
    OffsetOrder order;
    for (int i = 0; i < NSLOTS; ++i) free_q.push(i);
    int nfiles_open = 0;
    for (int i = 0; i < NFILES; ++i) {
        nfiles_open += out.f[i].is_open() ? 1 : 0;
        out.f[i].reserved = out.f[i].end;  // (the report's header line is there already)
    }
    // the run's regime (the same rule as run_on_device): the value files' text comes from the device, with its size known
    // before it is copied out -- their places are reserved; or it is formatted on the host and appended in input order.  The
    // report's lines have a size known from the ids: its places are reserved in every regime.
    std::unique_ptr<EarlyTrim> trimmer_owner(new EarlyTrim(out, order, input.file_bytes()));
    EarlyTrim& trimmer = *trimmer_owner;
    static const bool host_format = std::getenv("SPUMONI_HOST_FORMAT") != nullptr;
    const bool no_len_text = o.report_only && !o.ms && o.write_report;
    const uint32_t run_streams = (no_len_text ? 0u : SPX_TEXT_LENGTHS) | (o.ms ? SPX_TEXT_POINTERS : 0u) | (o.use_doc ? SPX_TEXT_DOCS : 0u);
    const bool text_run = !host_format && run_streams != 0;
    auto by_offsets = [&](int f) { return f == F_REPORT || text_run; };
    std::atomic<size_t> num_reads{0};
    std::atomic<uint64_t> direct_bytes{0}, staged_bytes{0};
    // ---- device workers ----
    std::vector<double> dev_busy(nworkers, 0.0), dev_finish(nworkers, 0.0), dev_wait(nworkers, 0.0);
    std::vector<size_t> dev_batches(nworkers, 0);
    std::atomic<uint64_t> next_seq_to_run{0};
    std::atomic<int64_t> first_on_device_us{-1}, last_off_device_us{0};
    auto us_since_start = [&] { return (int64_t)(since(t_stage0) * 1e6); };
    auto device_worker = [&](size_t d) {
        for (;;) {
            const uint64_t seq = next_seq_to_run.fetch_add(1);
            const auto tw = tick();
            const int i = parsed.look(seq);
            dev_wait[d] += since(tw);
            if (i < 0) break;
            parsed.drop(seq);
            const auto t0 = tick();
            {
                int64_t none = -1;
                (void)first_on_device_us.compare_exchange_strong(none, us_since_start());
            }
            Slot& s = slots[(size_t)i];
            // the super-batch's place in the files: reserved once, in input order, as soon as its sizes are known
            auto place = [&](const uint64_t bytes[3], char* dest[3]) {
                uint64_t want[NFILES] = {bytes[0], bytes[1], bytes[2], o.write_report ? s.report_bytes : 0};
                for (int f = 0; f < NFILES; ++f) {
                    if (!out.f[f].is_open()) want[f] = 0;
                    s.file_bytes[f] = want[f];
                }
                order.reserve(s.seq, out, want, s.file_off, s.input_bytes);
                trimmer.poke();
                s.placed = true;
                for (int f = 0; f < 3; ++f)
                    if (want[f] && out.f[f].pinned && s.file_off[f] + want[f] <= out.f[f].pinned_size) dest[f] = out.f[f].map + s.file_off[f];
            };
            if (s.sb.nreads() > 0) run_on_device(set.ix[d], o, s.sb, max_value_thr, s.res, place);
            if (!s.placed) {
                const uint64_t none[3] = {0, 0, 0};
                char* unused[3] = {nullptr, nullptr, nullptr};
                s.report_bytes = 0;
                place(none, unused);
            }
            if (o.use_promotions || o.use_dna_letters) {
                // a read that digests to nothing is fatal where the reference meets it (:926-931):
                // everything before it is still written
                for (size_t q = 0; q < s.sb.nreads(); ++q)
                    if (s.res.beg[q] == s.res.end[q]) {
                        s.deferred = 2;
                        s.deferred_msg = std::string(s.sb.ids[q]);
                        s.sb.ids.resize(q);
                        s.sb.offs.n = q + 1;
                        // (what the super-batch leaves in the files ends before that read)
                        for (int f = 0; f < 3; ++f)
                            if (text_run && (run_streams & (1u << f)) && out.f[f].is_open()) s.file_bytes[f] = s.res.line_start[f][q];
                        uint64_t rb = 0;
                        for (size_t j = 0; j < q; ++j) rb += report_line_bytes(s.sb.ids[j].size());
                        if (o.write_report) s.file_bytes[F_REPORT] = rb;
                        break;
                    }
            }
            dev_busy[d] += since(t0);
            const auto t1 = tick();
            finish_batch(pool, o, out, s.sb, s.res, s.chunks, s.file_off, s.pieces);
            for (int f = 0; f < NFILES; ++f) {
                uint64_t staged = 0;
                for (const Piece& pc : s.pieces[f]) staged += pc.n;
                staged_bytes += staged;
                if (by_offsets(f)) direct_bytes += s.file_bytes[f] - std::min(staged, s.file_bytes[f]);
            }
            dev_finish[d] += since(t1);
            dev_batches[d]++;
            last_off_device_us.store(us_since_start());
            s.writers_left.store(nfiles_open);
            done.put(s.seq, i);
        }
    };
    std::vector<std::thread> workers;
    for (size_t d = 0; d < nworkers; ++d) workers.emplace_back(device_worker, d);
    // ---- writers: one per file, in input order.  The last one through with a super-batch makes it final (the files' logical
    // ends move), raises what the reference would have stopped at -- after every file holds everything before it -- and
    // gives the slot back.
    std::vector<double> write_s(NFILES, 0.0);
    auto file_writer = [&](int f) {
        for (uint64_t seq = 0;; ++seq) {
            const int i = done.look(seq);
            Slot& s = slots[(size_t)i];
            const auto t0 = tick();
            if (by_offsets(f)) {
                uint64_t at = s.file_off[f];
                // (not while EarlyTrim cuts the file: see there -- and only when there is something to write: the cut takes tens of
                // milliseconds, and a writer that waits for it holds up its slot, the feeders behind it and every worker)
                std::unique_lock<std::mutex> cut(out.f[f].cut_mu, std::defer_lock);
                if (!s.pieces[f].empty()) cut.lock();
                for (const Piece& pc : s.pieces[f]) {
                    const uint64_t n = std::min<uint64_t>(pc.n, s.file_off[f] + s.file_bytes[f] - at);  // (a super-batch cut at a fatal read)
                    out.f[f].write_at(pc.p, n, at);
                    at += n;
                }
            } else {
                // values formatted on the host: the pieces' sizes were not known when the places were handed out (they were
                // handed out empty): appended in input order
                for (const Piece& pc : s.pieces[f]) out.f[f].append(pc.p, pc.n);
            }
            write_s[(size_t)f] += since(t0);
            // a super-batch that ends in a deferred fatal (a read empty after digestion is only known once the batch is
            // back from its device) is the last one written: nothing of a later super-batch may reach the files (the
            // reference stops AT the read)
            const bool last = s.last || s.deferred != 0;
            if (s.writers_left.fetch_sub(1) == 1) {
                for (int g = 0; g < NFILES; ++g)
                    if (out.f[g].is_open() && by_offsets(g)) out.f[g].end = s.file_off[g] + s.file_bytes[g];
                num_reads += s.sb.nreads();
                if (s.deferred == 1) fatal_error("%s", s.deferred_msg.c_str());
                if (s.deferred == 2) {
                    std::cout << "\n\n";
                    fatal_warning("%s was empty after digestion, commonly due to reads "
                                  "consisting of mostly non-ACGT characters. Please remove "
                                  "read or run SPUMONI without minimizer digestion.", s.deferred_msg.data());
                }
                done.drop(seq);
                if (!last) free_q.push(i);
            }
            if (last) break;
        }
    };
    std::vector<std::thread> writers;
    for (int f = 0; f < NFILES; ++f)
        if (out.f[f].is_open()) writers.emplace_back(file_writer, f);
    // ---- feeders ----
    std::mutex seg_mu;
    std::atomic<bool> input_done{false};  // (set without the lock by a feeder whose super-batch ends in a fatal read: the other
                                          // feeder may be waiting for a slot with the lock held)
    uint64_t next_seq = 0;
    double seg_s = 0;
    std::vector<std::array<double, 2>> parse_s((size_t)NFEED, std::array<double, 2>{0.0, 0.0});
    auto feeder = [&](int fi) {
        std::vector<ReadFile::Range> ranges;
        for (;;) {
            int i;
            {
                std::lock_guard<std::mutex> g(seg_mu);
                if (input_done) return;
                // (the slot is taken under the lock: super-batch k holds a slot before k + 1 asks for one, so the stages
                // behind, which want k first, can never be starved of it by later super-batches)
                i = free_q.pop();
                const auto t0 = tick();
                ranges.clear();
                size_t bytes = 0;
                while (bytes < o.super_batch_chars) {
                    ReadFile::Range r;
                    if (!input.next_range(1000, r)) {  // reader.loadBatch(input_file, 1000)   (:903)
                        input_done = true;
                        break;
                    }
                    bytes += r.bytes;
                    ranges.push_back(r);
                }
                slots[(size_t)i].seq = next_seq++;
                slots[(size_t)i].last = input_done;
                slots[(size_t)i].input_bytes = 0;
                for (const ReadFile::Range& r : ranges) slots[(size_t)i].input_bytes += r.bytes + (r.last - r.first);
                seg_s += since(t0);
            }
            Slot& s = slots[(size_t)i];
            fill_slot(pool, input, ranges, s, parse_s[(size_t)fi].data());
            if (s.deferred) input_done = true;  // nothing behind a malformed / empty read is part of the run
            const bool last = s.last;
            const uint64_t seq = s.seq;
            parsed.put(seq, i);
            if (last) {
                parsed.set_last(seq);
                return;
            }
        }
    };
    std::vector<std::thread> feeders;
    for (int fi = 1; fi < NFEED; ++fi) feeders.emplace_back(feeder, fi);
    feeder(0);
    for (auto& t : feeders) t.join();
    for (auto& w : workers) w.join();
    for (auto& w : writers) w.join();
    // (the files' tails were prepared from an estimate on the generous side: what is too much is given back here, ~13 ms per
    // 100 MB of allocated, mapped, page-locked pages -- profiles/r05_early_trim_experiment.txt on why not beside the run)
    trimmer.finish();  // (joined before the files are cut)
    const double t_before_settle = since(t_stage0);
    settle_outputs(true);  // the files end where their last super-batch does
    std::fprintf(stderr, "[timing] the first super-batch reached its device after %.3f s, the last left it after %.3f s, the files were complete after %.3f s "
                         "and cut to their sizes after %.3f s (un-registering %.3f, the cut %.3f s; excess cut off beside the run: %.3f MB in %.3f s)\n",
                 (double)first_on_device_us.load() / 1e6, (double)last_off_device_us.load() / 1e6, t_before_settle, since(t_stage0),
                 out.f[0].settle_s[0] + out.f[1].settle_s[0] + out.f[2].settle_s[0] + out.f[3].settle_s[0],
                 out.f[0].settle_s[2] + out.f[1].settle_s[2] + out.f[2].settle_s[2] + out.f[3].settle_s[2], (double)trimmer.trimmed_bytes() / 1e6, trimmer.seconds());
    std::fprintf(stderr, "[timing] %-22s %.3f s\n", "first read .. last byte", since(t_stage0));
    g_calls.print();
    // per-stage times (ours; the stages overlap and most are sums over threads, so they do not add up to the total)
    double p0 = 0, p1 = 0;
    for (int fi = 0; fi < NFEED; ++fi) p0 += parse_s[(size_t)fi][0], p1 += parse_s[(size_t)fi][1];
    std::fprintf(stderr, "[timing] %-22s %.3f s  (%d feeders on a pool of %zu: segmentation %.3f  scan %.3f  copy %.3f s)\n", "segment+parse",
                 seg_s + p0 + p1, NFEED, pool.size(), seg_s, p0, p1);
    static const char* const fname[NFILES] = {"lengths", "pointers", "doc_numbers", "report"};
    for (int f = 0; f < NFILES; ++f)
        if (out.f[f].is_open())
            std::fprintf(stderr, "[timing] writer %-15s %.3f s  (%.1f MB; %s)\n", fname[f], write_s[(size_t)f], (double)out.f[f].end / 1e6,
                         out.f[f].map_size ? "its tail was prepared as memory" : "plain writes");
    std::fprintf(stderr, "[timing] output bytes: %.3f MB went straight into the files' pages, %.3f MB through the writer threads (files prepared in %.3f s beside the index load: allocate + map %.3f, page table entries %.3f, page-locking %.3f s)\n",
                 (double)direct_bytes.load() / 1e6, (double)staged_bytes.load() / 1e6, out.prepare_s, out.prep_s[0], out.prep_s[1], out.prep_s[2]);
    for (size_t d = 0; d < nworkers; ++d)
        std::fprintf(stderr, "[timing] gpu worker %zu          %.3f s  (%zu super-batches, copies included)  + headers / report %.3f s, waiting for input %.3f s\n", d,
                     dev_busy[d], dev_batches[d], dev_finish[d], dev_wait[d]);
    return num_reads.load();
}

size_t classify_general_reads(IndexSet& set, const RunOptions& o) {
    // :1219-1297: reads are separated by \x01; trailing text without a separator is ignored
    RunOptions oo = o;
    oo.use_doc = false;
    oo.write_report = false;
    Outputs out;
    out.f[F_LENGTHS].open(o.pattern_file + (o.ms ? ".lengths" : ".pseudo_lengths"));
    if (o.ms) out.f[F_POINTERS].open(o.pattern_file + ".pointers");
    std::vector<uint8_t> data;
    if (!read_whole_file(o.pattern_file, data)) fatal_error("The following path is not valid: %s", o.pattern_file.data());
    Pool pool(std::max<size_t>(1, o.format_threads) - 1);
    SuperBatch sb;
    Results res;
    std::vector<TextChunk> chunks;
    std::vector<Piece> pieces[NFILES];
    size_t num_reads = 0, start = 0;
    auto flush = [&]() {
        if (sb.nreads() == 0) return;
        const uint64_t at0[NFILES] = {0, 0, 0, 0};
        run_on_device(set.ix[0], oo, sb, 0, res, [](const uint64_t*, char**) {});
        finish_batch(pool, oo, out, sb, res, chunks, at0, pieces);
        for (int f = 0; f < NFILES; ++f)
            for (const Piece& pc : pieces[f]) out.f[f].append(pc.p, pc.n);
        sb.clear();
    };
    for (size_t i = 0; i < data.size(); ++i) {
        if (data[i] == 0x01) {
            sb.seqs.append(data.data() + start, i - start);
            sb.offs.push_back(sb.seqs.size());
            sb.longest = std::max<uint64_t>(sb.longest, i - start);
            sb.own_ids.push_back("read_" + std::to_string(num_reads));
            sb.ids.push_back(sb.own_ids.back());
            sb.gap.push_back((uint32_t)sb.own_ids.back().size() + 2);
            num_reads++;
            start = i + 1;
            if (sb.seqs.size() >= o.super_batch_chars) flush();
        }
    }
    flush();
    return num_reads;
}

}  // namespace spumoni_host
