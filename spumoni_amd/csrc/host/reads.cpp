#include "reads.hpp"

#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>

namespace spumoni_host {

void fatal_error(const char* fmt, ...) {  // FATAL_ERROR, include/spumoni_main.hpp:32-33
    std::fprintf(stderr, "\n\033[31mError: \033[0m");
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
    std::fprintf(stderr, "\n\n");
    std::exit(1);
}

void fatal_warning(const char* fmt, ...) {  // FATAL_WARNING, include/spumoni_main.hpp:28-29
    std::fprintf(stderr, "Warning: ");
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
    std::fprintf(stderr, "\n\n");
    std::exit(1);
}

ReadFile::ReadFile(const std::string& path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    in.seekg(0, std::ios::end);
    const std::streamoff sz = in.tellg();
    in.seekg(0, std::ios::beg);
    data_.resize((size_t)sz);
    if (sz > 0) in.read(&data_[0], sz);
    // line table (std::getline semantics: a trailing '\n' does not start another line)
    size_t b = 0;
    for (size_t i = 0; i < data_.size(); ++i) {
        if (data_[i] == '\n') {
            lines_.emplace_back(b, i);
            b = i + 1;
        }
    }
    if (b < data_.size()) lines_.emplace_back(b, data_.size());
    ends_with_newline_ = !data_.empty() && data_.back() == '\n';
    if (data_.empty()) eof_ = true;
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

bool ReadFile::next_range(size_t num_bases, Range& out) {
    // input type sniffing (batch_loader.cpp:30-38)
    if (format_ == ReadFormat::NotClear) {
        if (data_.empty()) return false;
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
        if (next_line_ >= lines_.size()) {
            // getline() fails: nothing left.  loadBatch returns false and the lines already
            // taken for this batch are dropped (FASTQ tail quirk, Appendix C16)
            eof_ = true;
            return false;
        }
        const size_t len = lines_[next_line_].second - lines_[next_line_].first;
        const bool last_line = next_line_ + 1 == lines_.size();
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
    out.bytes = 0;
    for (size_t i = first; i < next_line_; ++i) out.bytes += lines_[i].second - lines_[i].first;
    return true;
}

static inline void strip_trailing_space(std::string_view& s) {
    while (!s.empty() && std::isspace((unsigned char)s.back())) s.remove_suffix(1);
}

// grabNextRead (batch_loader.cpp:78-131) over the lines of one batch
void ReadFile::parse_range(const Range& r, std::vector<ParsedRead>& out, ParseError& err) const {
    const size_t last = r.last;
    size_t i = r.first;
    auto fail = [&](const std::string& msg) {
        err.fatal = true;
        err.message = msg;
    };
    while (i < last) {
        std::string_view hdr = line(i++);
        if (hdr.empty()) return;  // an empty header line ends the batch (Appendix C15)
        if (format_ == ReadFormat::Fastq) {
            if (hdr[0] != '@')
                return fail(std::string("Incorrect FASTQ entry, it should start with '@' but found ") + hdr[0]);
        } else if (hdr[0] != '>') {
            return fail(std::string("Incorrect FASTA entry, it should start with '>' but found ") + hdr[0]);
        }
        if (hdr.size() <= 2) return fail("header line is missing an id. invalid query cannot be processed.");
        size_t ws = hdr.find_first_of(" \t\r", 1);
        if (ws == std::string_view::npos) ws = hdr.size();
        ParsedRead rd;
        rd.id.assign(hdr.substr(1, ws));  // count = ws: includes the whitespace character itself
        if (format_ == ReadFormat::Fastq) {
            if (i >= last) return;
            std::string_view s = line(i++);
            strip_trailing_space(s);
            rd.seq.assign(s);
            if (i >= last) return;  // '+' line
            i++;
            if (i >= last) return;  // qualities
            i++;
        } else {
            bool dropped = false;
            for (;;) {
                if (i >= last) {
                    // the reference peeks past the end, getline fails and it returns seq.size():
                    // a record with an empty sequence at the end of a batch is dropped
                    if (rd.seq.empty()) dropped = true;
                    break;
                }
                std::string_view s = line(i);
                if (!s.empty() && s[0] == '>') break;
                i++;
                strip_trailing_space(s);
                rd.seq.append(s);
            }
            if (dropped) return;
        }
        out.push_back(std::move(rd));
    }
}

}  // namespace spumoni_host
