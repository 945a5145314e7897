#include "classify.hpp"

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "index_files.hpp"
#include "reads.hpp"

namespace spumoni_host {

IndexSet::~IndexSet() {
    for (spx_index* p : ix) spx_index_free(p);
}

void IndexSet::load(const RunOptions& o) {
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
    if (o.ms) {
        if (o.text_file.empty() || !read_whole_file(o.text_file, text))
            fatal_error("MS lengths need the indexed text as a plain file (set SPUMONI_TEXT): the SLP of the\n"
                        "       reference is replaced by plain text in GPU memory (see DESIGN.md)");
    }
    n = raw.n;
    r = raw.heads.size();
    for (int dev : o.devices) {
        spx_index* p = spx_index_from_runs(raw.heads.data(), raw.lens.data(), raw.thr.data(), r,
                                           o.ms ? raw.ssa.data() : nullptr, o.ms ? raw.esa.data() : nullptr,
                                           o.use_doc ? raw.doc_start.data() : nullptr,
                                           o.use_doc ? raw.doc_end.data() : nullptr, 0, dev);
        if (!p) fatal_error("%s", spx_last_error());
        if (o.ms && spx_index_set_text(p, text.data(), text.size(), 0) != SPX_OK)
            fatal_error("%s", spx_last_error());
        ix.push_back(p);
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
struct TextBuf {
    std::string s;
    void header(const std::string& id) {
        s.push_back('>');
        s.append(id);
        s.push_back('\n');
    }
    void u64(uint64_t v) {
        char tmp[24];
        int p = 24;
        do {
            tmp[--p] = (char)('0' + v % 10);
            v /= 10;
        } while (v);
        s.append(tmp + p, 24 - p);
        s.push_back(' ');
    }
    void newline() { s.push_back('\n'); }
    void flush(std::ofstream& f) {
        f.write(s.data(), (std::streamsize)s.size());
        s.clear();
    }
};

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
        T* q = (T*)spx_host_alloc(nc * sizeof(T));
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
    std::vector<std::string> ids;
    PinnedBuf<uint8_t> seqs;
    std::vector<uint64_t> offs{0};
    void clear() {
        ids.clear();
        seqs.n = 0;
        offs.assign(1, 0);
    }
    size_t nreads() const { return ids.size(); }
};

struct Results {
    PinnedBuf<uint32_t> lengths, docs;
    PinnedBuf<uint64_t> pointers;
    PinnedBuf<spx_class> cls;
    // results of read q are entries [beg[q], end[q]) of the arrays above: the read's own
    // offsets, or -- with -m / -a -- the offsets of the digested read
    std::vector<uint64_t> beg, end;
};

// contiguous, character-balanced shards: one per device, run concurrently
void run_on_devices(IndexSet& set, const RunOptions& o, const SuperBatch& sb, size_t max_value_thr, Results& res) {
    const size_t nreads = sb.nreads();
    const uint64_t total = sb.offs.back();
    const bool digest = o.use_promotions || o.use_dna_letters;
    const int kind = o.use_promotions ? SPX_DIGEST_PROMOTED : SPX_DIGEST_DNA;
    // with digestion a shard's results are laid out at the digested offsets, inside a region as
    // large as its worst case (every k-mer reported: 1 byte each for -m, k letters for -a)
    const uint64_t grow = digest && o.use_dna_letters ? (uint64_t)o.k : 1;
    res.lengths.resize_uninit(total * grow);
    if (o.ms) res.pointers.resize_uninit(total * grow);
    if (o.use_doc) res.docs.resize_uninit(total * grow);
    if (o.write_report) res.cls.resize_uninit(nreads);
    res.beg.resize(nreads);
    res.end.resize(nreads);
    const size_t ndev = set.ix.size();
    std::vector<size_t> cut(ndev + 1, nreads);
    cut[0] = 0;
    for (size_t d = 1; d < ndev; ++d) {
        const uint64_t target = total * d / ndev;
        size_t c = std::lower_bound(sb.offs.begin(), sb.offs.end() - 1, target) - sb.offs.begin();
        cut[d] = std::max(cut[d - 1], std::min(c, nreads));
    }
    std::vector<std::string> errors(ndev);
    auto work = [&](size_t d) {
        const size_t lo = cut[d], hi = cut[d + 1];
        if (hi <= lo) return;
        const uint64_t a = sb.offs[lo];
        std::vector<uint64_t> offs(hi - lo + 1);
        for (size_t q = lo; q <= hi; ++q) offs[q - lo] = sb.offs[q] - a;
        const uint64_t ra = a * grow;  // where this shard's results start
        int rc;
        if (!digest) {
            rc = spx_query_batch(set.ix[d], o.ms ? SPX_MODE_MS : SPX_MODE_PML, sb.seqs.data() + a, offs.data(),
                                 hi - lo, res.lengths.data() + ra, o.ms ? res.pointers.data() + ra : nullptr,
                                 o.use_doc ? res.docs.data() + ra : nullptr,
                                 o.write_report ? res.cls.data() + lo : nullptr, o.bin_size, max_value_thr);
            for (size_t q = lo; q < hi; ++q) {
                res.beg[q] = sb.offs[q];
                res.end[q] = sb.offs[q + 1];
            }
        } else {
            // perform_minimizer_digestion / perform_dna_minimizer_digestion + matching_statistics
            // (compute_ms_pml.cpp:919-938), the batch on the device in one call
            std::vector<uint64_t> doffs(hi - lo + 1);
            rc = spx_digest_query_batch(set.ix[d], o.ms ? SPX_MODE_MS : SPX_MODE_PML, kind, (uint32_t)o.k,
                                        (uint32_t)o.w, sb.seqs.data() + a, offs.data(), hi - lo, doffs.data(),
                                        (sb.offs[hi] - a) * grow, res.lengths.data() + ra,
                                        o.ms ? res.pointers.data() + ra : nullptr,
                                        o.use_doc ? res.docs.data() + ra : nullptr,
                                        o.write_report ? res.cls.data() + lo : nullptr, o.bin_size, max_value_thr);
            if (rc == SPX_OK)
                for (size_t q = lo; q < hi; ++q) {
                    res.beg[q] = ra + doffs[q - lo];
                    res.end[q] = ra + doffs[q - lo + 1];
                }
        }
        if (rc != SPX_OK) errors[d] = spx_last_error();
    };
    std::vector<std::thread> th;
    for (size_t d = 1; d < ndev; ++d) th.emplace_back(work, d);
    work(0);
    for (auto& t : th) t.join();
    for (auto& e : errors)
        if (!e.empty()) fatal_error("%s", e.c_str());
}

struct Outputs {
    std::ofstream lengths, pointers, docs, report;
};

// Formats reads [lo, hi) of a super-batch into text; run by several host threads at once
// (the text of 10^6 reads x 200 values is ~0.5 GB: at GPU speed, formatting is the job).
struct TextChunk {
    TextBuf tl, tp, td;
    std::string report;
};

void format_range(const RunOptions& o, const SuperBatch& sb, const Results& res, size_t lo, size_t hi,
                  TextChunk& out) {
    std::ostringstream rep;
    for (size_t q = lo; q < hi; ++q) {
        const uint64_t a = res.beg[q], b = res.end[q];
        if (o.use_doc) {  // compute_ms_pml.cpp:1003-1007
            out.td.header(sb.ids[q]);
            for (uint64_t i = a; i < b; ++i) out.td.u64(res.docs[i]);
            out.td.newline();
        }
        out.tl.header(sb.ids[q]);  // :1008-1010
        for (uint64_t i = a; i < b; ++i) out.tl.u64(res.lengths[i]);
        out.tl.newline();
        if (o.ms) {  // :1190-1195
            out.tp.header(sb.ids[q]);
            for (uint64_t i = a; i < b; ++i) out.tp.u64(res.pointers[i]);
            out.tp.newline();
        }
        if (o.write_report) {  // :1012-1020
            const spx_class& c = res.cls[q];
            const size_t nbins = (size_t)c.bins_above + c.bins_below;
            const bool read_found = (c.bins_above / (c.bins_above + c.bins_below + 0.0) > 0.50);
            rep.precision(3);
            rep << std::setw(30) << std::left << sb.ids[q] << std::setw(15) << std::left
                << (read_found ? "FOUND" : "NOT_PRESENT") << std::setw(26) << std::left
                << (c.sum_max_bin_values + 0.0) / nbins << std::setw(12) << std::left << (size_t)c.bins_above
                << std::setw(12) << std::left << (size_t)c.bins_below << '\n';
        }
    }
    out.report = rep.str();
}

void write_results(Outputs& out, const RunOptions& o, const SuperBatch& sb, const Results& res) {
    const size_t nreads = sb.nreads();
    size_t nt = std::max<size_t>(1, std::min<size_t>(o.format_threads, (nreads + 4095) / 4096));
    std::vector<TextChunk> chunks(nt);
    std::vector<std::thread> th;
    auto lo_of = [&](size_t t) { return nreads * t / nt; };
    for (size_t t = 1; t < nt; ++t)
        th.emplace_back([&, t]() { format_range(o, sb, res, lo_of(t), lo_of(t + 1), chunks[t]); });
    format_range(o, sb, res, lo_of(0), lo_of(1), chunks[0]);
    for (auto& x : th) x.join();
    for (TextChunk& c : chunks) {  // written in input order (the reference's -t 1 order)
        c.tl.flush(out.lengths);
        if (o.ms) c.tp.flush(out.pointers);
        if (o.use_doc) c.td.flush(out.docs);
        if (o.write_report) out.report.write(c.report.data(), (std::streamsize)c.report.size());
    }
    if (o.write_report) out.report.flush();
}

size_t open_outputs_and_threshold(Outputs& out, const RunOptions& o) {
    out.lengths.open(o.pattern_file + (o.ms ? ".lengths" : ".pseudo_lengths"));
    if (o.ms) out.pointers.open(o.pattern_file + ".pointers");
    if (o.use_doc) out.docs.open(o.pattern_file + ".doc_numbers");
    if (o.write_report) out.report.open(o.pattern_file + ".report", std::ofstream::out);
    double percentile = 0.0;
    std::string err;
    // the reference does not check the stream either (:867-869): a missing null database
    // leaves percentile_value at 0.0
    (void)load_null_db(o.ref_file + (o.ms ? ".msnulldb" : ".pmlnulldb"), percentile, err);
    const size_t max_value_thr = max_value_threshold(percentile, !o.ms, o.use_promotions, o.use_dna_letters);
    if (o.write_report) {  // :877-886
        out.report.precision(4);
        out.report << std::setw(30) << std::left << "read id:" << std::setw(15) << std::left << "status:"
                   << std::setw(19) << std::left << "avg max-value (thr=" << std::setw(2) << std::left
                   << max_value_thr << std::setw(5) << std::left << "):" << std::setw(12) << std::left
                   << "above thr:" << std::setw(12) << std::left << "below thr:" << std::endl;
    }
    return max_value_thr;
}

}  // namespace

namespace {

// A super-batch in flight plus what has to happen after it was written.
struct Slot {
    SuperBatch sb;
    Results res;
    bool last = false;           // no more input after this one
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
void fill_slot(ReadFile& input, const RunOptions& o, Slot& slot, bool& input_done) {
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
    const size_t nt = std::max<size_t>(1, std::min<size_t>(o.format_threads, (ranges.size() + 63) / 64));
    std::vector<std::vector<ParsedRead>> parsed(nt);
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
    // where does every thread's part go, and where is the first problem (if any)?
    std::vector<size_t> take(nt, 0), chars(nt, 0);
    for (size_t t = 0; t < nt && !slot.deferred; ++t) {
        for (const ParsedRead& rd : parsed[t]) {
            if (rd.seq.empty()) {  // :926-931
                slot.deferred = 2;
                slot.deferred_msg = rd.id;
                break;
            }
            take[t]++;
            chars[t] += rd.seq.size();
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
    auto assemble = [&](size_t t) {
        size_t rdx = r0[t], cpos = c0[t];
        for (size_t q = 0; q < take[t]; ++q) {
            ParsedRead& rd = parsed[t][q];
            uint8_t* dst = slot.sb.seqs.data() + cpos;
            const char* src = rd.seq.data();
            const size_t len = rd.seq.size();
            // make sure all characters are upper-case (:916-917; ::toupper in the "C" locale)
            for (size_t i = 0; i < len; ++i) {
                const unsigned char ch = (unsigned char)src[i];
                dst[i] = (uint8_t)((ch >= 'a' && ch <= 'z') ? ch - 32 : ch);
            }
            cpos += len;
            slot.sb.offs[rdx + 1] = cpos;
            slot.sb.ids[rdx] = std::move(rd.id);
            rdx++;
        }
    };
    th.clear();
    for (size_t t = 1; t < nt; ++t) th.emplace_back(assemble, t);
    assemble(0);
    for (auto& x : th) x.join();
}

}  // namespace

namespace {
struct StageTimer {  // SPUMONI_TIMING=1: per-stage wall time on stderr (ours)
    const char* name;
    double total = 0;
    std::chrono::steady_clock::time_point t0;
    void start() { t0 = std::chrono::steady_clock::now(); }
    void stop() { total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

size_t classify_reads(IndexSet& set, const RunOptions& o) {
    Outputs out;
    const size_t max_value_thr = open_outputs_and_threshold(out, o);
    const bool timing = std::getenv("SPUMONI_TIMING") != nullptr;
    StageTimer t_load{"load+index lines", 0, {}}, t_parse{"segment+parse", 0, {}}, t_gpu{"gpu (incl. copies)", 0, {}},
        t_write{"format+write", 0, {}};
    t_load.start();
    ReadFile input(o.pattern_file);
    t_load.stop();
    // three stages, three slots: the host parses super-batch i+1 and formats/writes i-1 while
    // the GPUs walk super-batch i; results are written in input order
    constexpr int NSLOTS = 3;
    std::vector<Slot> slots(NSLOTS);
    SlotQueue free_q, parsed_q, computed_q;
    for (int i = 0; i < NSLOTS; ++i) free_q.push(i);
    size_t num_reads = 0;
    std::thread gpu([&] {
        for (;;) {
            const int i = parsed_q.pop();
            if (i < 0) break;
            t_gpu.start();
            if (slots[i].sb.nreads() > 0) run_on_devices(set, o, slots[i].sb, max_value_thr, slots[i].res);
            if (o.use_promotions || o.use_dna_letters) {
                // a read that digests to nothing is fatal where the reference meets it (:926-931):
                // everything before it is still written
                Slot& s = slots[i];
                for (size_t q = 0; q < s.sb.nreads(); ++q)
                    if (s.res.beg[q] == s.res.end[q]) {
                        s.deferred = 2;
                        s.deferred_msg = s.sb.ids[q];
                        s.sb.ids.resize(q);
                        s.sb.offs.resize(q + 1);
                        break;
                    }
            }
            t_gpu.stop();
            computed_q.push(i);
        }
        computed_q.push(-1);
    });
    std::thread writer([&] {
        for (;;) {
            const int i = computed_q.pop();
            if (i < 0) break;
            Slot& s = slots[i];
            t_write.start();
            if (s.sb.nreads() > 0) write_results(out, o, s.sb, s.res);
            t_write.stop();
            num_reads += s.sb.nreads();
            if (s.deferred == 1) {
                out.lengths.flush();
                fatal_error("%s", s.deferred_msg.c_str());
            }
            if (s.deferred == 2) {
                out.lengths.flush();
                out.pointers.flush();
                out.docs.flush();
                out.report.flush();
                std::cout << "\n\n";
                fatal_warning("%s was empty after digestion, commonly due to reads "
                              "consisting of mostly non-ACGT characters. Please remove "
                              "read or run SPUMONI without minimizer digestion.", s.deferred_msg.data());
            }
            free_q.push(i);
        }
    });
    bool input_done = false;
    for (;;) {
        const int i = free_q.pop();
        t_parse.start();
        fill_slot(input, o, slots[i], input_done);
        t_parse.stop();
        const bool last = slots[i].last;
        parsed_q.push(i);
        if (last) break;
    }
    parsed_q.push(-1);
    gpu.join();
    writer.join();
    if (timing)
        for (StageTimer* t : {&t_load, &t_parse, &t_gpu, &t_write})
            std::fprintf(stderr, "[timing] %-20s %.3f s\n", t->name, t->total);
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
    size_t num_reads = 0, start = 0;
    auto flush = [&]() {
        if (sb.nreads() == 0) return;
        run_on_devices(set, oo, sb, 0, res);
        write_results(out, oo, sb, res);
        sb.clear();
    };
    for (size_t i = 0; i < data.size(); ++i) {
        if (data[i] == 0x01) {
            sb.seqs.append(data.data() + start, i - start);
            sb.offs.push_back(sb.seqs.size());
            sb.ids.push_back("read_" + std::to_string(num_reads));
            num_reads++;
            start = i + 1;
            if (sb.seqs.size() >= o.super_batch_chars) flush();
        }
    }
    flush();
    return num_reads;
}

}  // namespace spumoni_host
