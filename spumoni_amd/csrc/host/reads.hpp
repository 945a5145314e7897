// reads.hpp -- host-side read batching with the reference's BatchLoader semantics
// (/root/reference/src/batch_loader.cpp:26-131), re-designed for feeding GPUs.
//
// The reference pulls ~1000-base batches out of an ifstream into a stringstream and
// parses reads from it one at a time under an OpenMP critical section.  Here the
// whole reads file is mapped once, line boundaries are found in one pass, and the
// same batch segmentation is replayed over the line table (it decides which reads
// exist at all for malformed input -- SURVEY Appendix C15/C16) while the reads are
// appended to large super-batches for spx_query_batch.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <string_view>
#include <vector>

namespace spumoni_host {

enum class ReadFormat { NotClear, Fasta, Fastq };

// A read as views into the mapped reads file (no copies: 4 * 10^6 reads a second have to come out of
// the parser); only a multi-line FASTA record owns a joined copy of its sequence.
struct ParsedRead {
    std::string_view id;      // header.substr(1, index of first whitespace)  -- keeps that whitespace (C6)
    std::string_view one;     // single-line sequence, trailing whitespace stripped
    std::string joined;       // multi-line FASTA concatenated, trailing whitespace of each line stripped
    bool multi = false;
    std::string_view seq() const { return multi ? std::string_view(joined) : one; }
};

// The same read as positions in the mapped file and nothing else (32 bytes, no constructor): what the harness parses
// super-batches into -- 10^7 reads a second have to come out of the parser, and a ParsedRead carries a std::string.
struct ReadRec {
    const char* id = nullptr;  // as ParsedRead::id
    uint32_t id_len = 0;
    uint32_t nlines = 0;       // sequence lines (FASTQ: 1; FASTA: as many as the record has)
    uint64_t seq_len = 0;      // characters of the sequence, trailing whitespace of every line stripped
    size_t first_line = 0;     // index of the first sequence line
};

class ReadFile {
public:
    // Maps the file and finds its lines (several threads); throws std::runtime_error if it cannot be read.
    explicit ReadFile(const std::string& path, unsigned threads = 8);
    ~ReadFile();
    ReadFile(const ReadFile&) = delete;
    ReadFile& operator=(const ReadFile&) = delete;

    // Next batch of the reference's loadBatch(input, num_bases) segmentation, parsed with
    // grabNextRead semantics.  Returns false when loadBatch would return false (end of
    // input, or the FASTQ tail quirk C16).  `out` receives the reads of the batch in order.
    bool next_batch(size_t num_bases, std::vector<ParsedRead>& out);

    // The same in two steps, so that many batches can be parsed by several threads:
    // the segmentation is sequential and cheap, parsing a range is const and thread-safe.
    struct Range {
        size_t first = 0, last = 0;  // lines [first, last) of one loadBatch() batch
        size_t bytes = 0;            // characters in those lines
    };
    // A malformed record is fatal in the reference the moment grabNextRead reaches it, i.e.
    // after every earlier read has been processed and written; parse_range therefore reports
    // it instead of exiting, together with the reads that precede it.
    struct ParseError {
        bool fatal = false;
        std::string message;  // printed with the FATAL_ERROR shape
    };
    bool next_range(size_t num_bases, Range& out);
    // The whole file's segmentation ahead of time (it is sequential, ~25 ns per line: 0.1 s for 4 * 10^6 reads):
    // done on the thread that maps the file while the index loads; next_range(num_bases) then hands the ranges out.
    void precompute_ranges(size_t num_bases);
    void parse_range(const Range& r, std::vector<ParsedRead>& out, ParseError& err) const;
    // grabNextRead over the lines of one batch, reads appended to `out` as positions (parse_range is this plus copies)
    void scan_range(const Range& r, std::vector<ReadRec>& out, ParseError& err) const;
    // the read's sequence, upper-cased (::toupper in the "C" locale, compute_ms_pml.cpp:916-917), to dst[0 .. seq_len)
    void copy_seq_upper(const ReadRec& rd, uint8_t* dst) const;
    // every range of the file at once (precompute_ranges must have run): the feeders cut super-batches out of it
    bool ranges_ready(size_t num_bases) const { return ranges_ready_ && num_bases == ranges_bases_; }

    ReadFormat format() const { return format_; }
    size_t file_bytes() const { return size_; }
    char first_char() const { return size_ ? data_[0] : '\0'; }  // ('@': FASTQ, batch_loader.cpp:30-38)
    size_t lines() const { return line_start_.size() - 1; }

private:
    const char* data_ = nullptr;  // the mapped file
    size_t size_ = 0;
    std::vector<size_t> line_start_;  // line i = [line_start_[i], line_start_[i + 1] - 1): one more entry than lines
    bool ends_with_newline_ = false;
    size_t next_line_ = 0;  // first line not yet consumed by a batch
    bool eof_ = false;      // the reference stream would no longer be good()
    ReadFormat format_ = ReadFormat::NotClear;

    std::vector<Range> ranges_;  // precompute_ranges()
    size_t ranges_bases_ = 0, ranges_next_ = 0;
    bool ranges_ready_ = false;
    bool next_range_scan(size_t num_bases, Range& out);

    size_t line_count() const { return line_start_.size() - 1; }
    size_t line_len(size_t i) const { return line_start_[i + 1] - 1 - line_start_[i]; }
    std::string_view line(size_t i) const { return std::string_view(data_ + line_start_[i], line_len(i)); }
};

// error helpers with the reference's message shapes (include/spumoni_main.hpp:28-33)
void register_leftover(const std::string& path);  // a file to be gone when the process ends, however it ends
void remove_leftovers();
void remove_stale_leftovers(const std::string& final_path);  // <final>.partial.<pid> / .old.<pid> of processes that are gone
void set_exit_hook(void (*hook)());               // called once on every way out through fatal_error / fatal_warning
[[noreturn]] void fatal_error(const char* fmt, ...);
[[noreturn]] void fatal_warning(const char* fmt, ...);

}  // namespace spumoni_host
