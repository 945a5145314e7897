#include "index_files.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace spumoni_host {

bool read_whole_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize((size_t)sz);
    const bool ok = sz == 0 || std::fread(out.data(), 1, (size_t)sz, f) == (size_t)sz;
    std::fclose(f);
    return ok;
}

static void unpack5(const std::vector<uint8_t>& raw, size_t stride, size_t pick, std::vector<uint64_t>& out) {
    const size_t n = raw.size() / (5 * stride);
    out.resize(n);
    for (size_t i = 0; i < n; ++i) {
        uint64_t v = 0;
        std::memcpy(&v, raw.data() + (i * stride + pick) * 5, 5);
        out[i] = v;
    }
}

bool load_raw_index(const std::string& prefix, bool want_samples, RawIndex& out, std::string& err) {
    std::vector<uint8_t> raw;
    if (!read_whole_file(prefix + ".bwt.heads", out.heads) || out.heads.empty()) {
        err = "cannot read " + prefix + ".bwt.heads";
        return false;
    }
    const size_t r = out.heads.size();
    if (!read_whole_file(prefix + ".bwt.len", raw) || raw.size() != r * 5) {
        err = prefix + ".bwt.len is missing or does not hold one 5-byte length per run";
        return false;
    }
    unpack5(raw, 1, 0, out.lens);
    if (!read_whole_file(prefix + ".thr_pos", raw) || raw.size() != r * 5) {
        err = prefix + ".thr_pos is missing or does not hold one 5-byte threshold per run";
        return false;
    }
    unpack5(raw, 1, 0, out.thr);
    out.n = 0;
    for (uint64_t v : out.lens) out.n += v;
    if (want_samples) {
        for (int which = 0; which < 2; ++which) {
            const std::string path = prefix + (which ? ".esa" : ".ssa");
            if (!read_whole_file(path, raw) || raw.size() != r * 10) {
                err = path + " is missing or does not hold one (left,right) 5-byte pair per run";
                return false;
            }
            std::vector<uint64_t>& dst = which ? out.esa : out.ssa;
            unpack5(raw, 2, 1, dst);
            for (auto& v : dst) v = v ? v - 1 : out.n - 1;
        }
    }
    return true;
}

namespace {
struct Cursor {
    const std::vector<uint8_t>& b;
    size_t p = 0;
    bool get(void* dst, size_t n) {
        if (p + n > b.size()) return false;
        std::memcpy(dst, b.data() + p, n);
        p += n;
        return true;
    }
};

bool read_int_vector(Cursor& c, std::vector<uint64_t>& out) {
    uint64_t bits = 0;
    uint8_t width = 0;
    if (!c.get(&bits, 8) || !c.get(&width, 1)) return false;
    const uint64_t words = (bits + 63) / 64;
    std::vector<uint64_t> w(words);
    if (words && !c.get(w.data(), words * 8)) return false;
    if (width == 0 || width > 64) {
        out.clear();
        return bits == 0;
    }
    const uint64_t cnt = bits / width;
    out.resize(cnt);
    for (uint64_t i = 0; i < cnt; ++i) {
        const uint64_t bit = i * width;
        const uint64_t wi = bit >> 6, sh = bit & 63;
        uint64_t v = w[wi] >> sh;
        if (sh + width > 64) v |= w[wi + 1] << (64 - sh);
        if (width < 64) v &= (1ull << width) - 1;
        out[i] = v;
    }
    return true;
}

void append_int_vector(std::vector<uint8_t>& buf, const std::vector<uint64_t>& vals, uint8_t width) {
    const uint64_t bits = (uint64_t)vals.size() * width;
    const uint64_t words = (bits + 63) / 64;
    std::vector<uint64_t> w(words, 0);
    for (size_t i = 0; i < vals.size(); ++i) {
        const uint64_t bit = (uint64_t)i * width;
        const uint64_t wi = bit >> 6, sh = bit & 63;
        const uint64_t v = width < 64 ? (vals[i] & ((1ull << width) - 1)) : vals[i];
        w[wi] |= v << sh;
        if (sh + width > 64) w[wi + 1] |= v >> (64 - sh);
    }
    const size_t at = buf.size();
    buf.resize(at + 9 + words * 8);
    std::memcpy(&buf[at], &bits, 8);
    buf[at + 8] = width;
    if (words) std::memcpy(&buf[at + 9], w.data(), words * 8);
}

bool write_file(const std::string& path, const std::vector<uint8_t>& buf) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = buf.empty() || std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    std::fclose(f);
    return ok;
}
}  // namespace

bool load_doc_array(const std::string& path, RawIndex& out, std::string& err) {
    std::vector<uint8_t> raw;
    if (!read_whole_file(path, raw)) {
        err = "cannot read " + path;
        return false;
    }
    Cursor c{raw};
    uint64_t num_entries = 0;
    if (!c.get(&num_entries, 8) || !read_int_vector(c, out.doc_start) || !read_int_vector(c, out.doc_end)) {
        err = path + " is truncated or not a document array";
        return false;
    }
    if (out.doc_start.size() != out.heads.size() || out.doc_end.size() != out.heads.size()) {
        err = path + " does not hold one entry per BWT run";
        return false;
    }
    return true;
}

bool load_null_db(const std::string& path, double& percentile_value, std::string& err) {
    std::vector<uint8_t> raw;
    if (!read_whole_file(path, raw)) {
        err = "cannot read " + path;
        return false;
    }
    Cursor c{raw};
    uint64_t num_values = 0;
    double ks = 0, mean = 0;
    if (!c.get(&num_values, 8) || !c.get(&ks, 8) || !c.get(&mean, 8) || !c.get(&percentile_value, 8)) {
        err = path + " is truncated";
        return false;
    }
    return true;
}

bool write_doc_array(const std::string& path, const std::vector<uint64_t>& doc_start,
                     const std::vector<uint64_t>& doc_end, std::string& err) {
    uint64_t mx = 1;
    for (auto v : doc_start) mx = v > mx ? v : mx;
    for (auto v : doc_end) mx = v > mx ? v : mx;
    uint8_t width = 1;
    while ((1ull << width) <= mx && width < 63) width++;
    std::vector<uint8_t> buf(8);
    const uint64_t num_entries = doc_start.size();
    std::memcpy(buf.data(), &num_entries, 8);
    append_int_vector(buf, doc_start, width);
    append_int_vector(buf, doc_end, width);
    if (!write_file(path, buf)) {
        err = "cannot write " + path;
        return false;
    }
    return true;
}

bool write_null_db(const std::string& path, double percentile_value, const std::vector<uint64_t>& stats,
                   std::string& err) {
    std::vector<uint8_t> buf(32);
    const uint64_t num_values = stats.size();
    double ks = 0.0, mean = 0.0;
    uint64_t mx = 1;
    for (auto v : stats) {
        mean += (double)v;
        mx = v > mx ? v : mx;
    }
    if (!stats.empty()) mean /= (double)stats.size();
    std::memcpy(&buf[0], &num_values, 8);
    std::memcpy(&buf[8], &ks, 8);
    std::memcpy(&buf[16], &mean, 8);
    std::memcpy(&buf[24], &percentile_value, 8);
    uint8_t width = 1;
    while ((1ull << width) <= mx && width < 63) width++;
    append_int_vector(buf, stats, width);
    if (!write_file(path, buf)) {
        err = "cannot write " + path;
        return false;
    }
    return true;
}

}  // namespace spumoni_host
