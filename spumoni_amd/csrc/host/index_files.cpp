#include "index_files.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

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
    // 10^9 records take seconds on one thread (and so does first touching 8 GB of output): eight threads
    const size_t n = raw.size() / (5 * stride);
    out.resize(n);
    const unsigned nt = n >= (1u << 22) ? 8 : 1;
    auto work = [&](unsigned t) {
        for (size_t i = n * t / nt, e = n * (t + 1) / nt; i < e; ++i) {
            uint64_t v = 0;
            std::memcpy(&v, raw.data() + (i * stride + pick) * 5, 5);
            out[i] = v;
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
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
    // (a size field is believed only as far as the file reaches: a damaged one must not size an allocation)
    if (bits > ~0ull - 63 || words > (c.b.size() - c.p) / 8) return false;
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

// ---------------------------------------------------------------------------------------------
// Serialised index streams (UNVERIFIED layouts, restated from the published sources of
// simongog/sdsl-lite v2.1.x and maxrossi91/r-index; see the header).
//
//  sdsl::int_vector<0>        u64 size_in_bits, u8 width, ceil(bits/64) x u64
//  sdsl::bit_vector           u64 size_in_bits, ceil(bits/64) x u64
//  sdsl::int_vector<64>       u64 size_in_bits, words               (rank_support_v = one of these)
//  sdsl::select_support_mcl   u64 arg_cnt; if > 0: int_vector<0> superblock, bit_vector mini_or_long
//                             (empty when there is no long superblock), then one int_vector<0> per
//                             superblock ((arg_cnt + 4095) >> 12 of them)
//  sdsl::sd_vector<>          u64 size, u8 wl, int_vector<0> low, bit_vector high,
//                             select_support_mcl<1>, select_support_mcl<0>;
//                             i-th one = ((select1(high, i) - i) << wl) | low[i]
//  ri::sparse_sd_vector       u64 u, u64 n; if u > 0: sd_vector
//  sdsl::wt_huff<>            u64 size, u64 sigma, bit_vector bv, rank_support_v, select_support_mcl<1>,
//                             select_support_mcl<0>, tree: u64 #nodes, nodes {u64 bv_pos, u64 bv_pos_rank,
//                             u16 parent, u16 child[2]} (22 bytes each), u16 c_to_leaf[256], u64 path[256]
//  ri::rle_string             u64 n, u64 R, u64 B; if n > 0: sparse_sd_vector runs,
//                             256 x sparse_sd_vector runs_per_letter, huff_string (= wt_huff) run_heads
// ---------------------------------------------------------------------------------------------
namespace {

struct BitVec {
    uint64_t bits = 0;
    std::vector<uint64_t> w;
    bool get(uint64_t i) const { return (w[i >> 6] >> (i & 63)) & 1; }
};

bool read_bit_vector(Cursor& c, BitVec& out) {
    if (!c.get(&out.bits, 8) || out.bits > ~0ull - 63) return false;
    const uint64_t words = (out.bits + 63) / 64;
    if (words > (c.b.size() - c.p) / 8) return false;
    out.w.resize(words);
    return words == 0 || c.get(out.w.data(), words * 8);
}

bool skip_int_vector0(Cursor& c) {
    uint64_t bits = 0;
    uint8_t width = 0;
    if (!c.get(&bits, 8) || !c.get(&width, 1)) return false;
    const uint64_t bytes = ((bits + 63) / 64) * 8;
    if (bytes > c.b.size() - c.p) return false;
    c.p += bytes;
    return true;
}

bool skip_select_mcl(Cursor& c) {
    uint64_t arg_cnt = 0;
    if (!c.get(&arg_cnt, 8)) return false;
    if (arg_cnt == 0) return true;
    if (!skip_int_vector0(c)) return false;
    BitVec mol;
    if (!read_bit_vector(c, mol)) return false;
    const uint64_t sb = (arg_cnt + 4095) >> 12;
    if (mol.bits != 0 && mol.bits != sb) return false;
    for (uint64_t i = 0; i < sb; ++i)
        if (!skip_int_vector0(c)) return false;
    return true;
}

// positions of the ones of an ri::sparse_sd_vector, ascending
bool read_sparse_sd(Cursor& c, std::vector<uint64_t>& ones, uint64_t& universe) {
    uint64_t u = 0, n = 0;
    ones.clear();
    if (!c.get(&u, 8) || !c.get(&n, 8)) return false;
    universe = u;
    if (u == 0) return true;
    uint64_t size = 0;
    uint8_t wl = 0;
    std::vector<uint64_t> low;
    BitVec high;
    if (!c.get(&size, 8) || !c.get(&wl, 1) || wl > 63) return false;
    if (!read_int_vector(c, low) || !read_bit_vector(c, high)) return false;
    if (!skip_select_mcl(c) || !skip_select_mcl(c)) return false;
    if (size != u) return false;
    if (n > high.bits) return false;  // (as many ones as the high part has bits, at most)
    ones.reserve(n);
    uint64_t i = 0;
    for (uint64_t p = 0; p < high.bits && i < n; ++p) {
        if (high.get(p)) {
            const uint64_t lo = wl ? (i < low.size() ? low[i] : 0) : 0;
            ones.push_back(((p - i) << wl) | lo);
            i++;
        }
    }
    if (ones.size() != n) return false;
    for (size_t t = 1; t < ones.size(); ++t)
        if (ones[t] <= ones[t - 1]) return false;
    return ones.empty() || ones.back() < u;
}

struct WtNode {
    uint64_t bv_pos, bv_pos_rank;
    uint16_t parent, child[2];
};

// decodes a whole sdsl::wt_huff<> back into the byte sequence it stores
bool read_wt_huff(Cursor& c, std::vector<uint8_t>& seq) {
    uint64_t size = 0, sigma = 0;
    BitVec bv, rank_words;
    if (!c.get(&size, 8) || !c.get(&sigma, 8) || !read_bit_vector(c, bv)) return false;
    if (!read_bit_vector(c, rank_words)) return false;  // rank_support_v: an int_vector<64>
    if (!skip_select_mcl(c) || !skip_select_mcl(c)) return false;
    uint64_t nnodes = 0;
    if (!c.get(&nnodes, 8) || nnodes == 0 || nnodes > 1024) return false;
    std::vector<WtNode> nodes(nnodes);
    for (auto& nd : nodes)
        if (!c.get(&nd.bv_pos, 8) || !c.get(&nd.bv_pos_rank, 8) || !c.get(&nd.parent, 2) || !c.get(nd.child, 4))
            return false;
    uint16_t c_to_leaf[256];
    uint64_t path[256];
    if (!c.get(c_to_leaf, sizeof c_to_leaf) || !c.get(path, sizeof path)) return false;
    std::vector<int> leaf_symbol(nnodes, -1);
    for (int ch = 0; ch < 256; ++ch)
        if (c_to_leaf[ch] != 0xffff && c_to_leaf[ch] < nnodes) leaf_symbol[c_to_leaf[ch]] = ch;
    if (size > bv.bits) return false;  // (the root level alone holds one bit per symbol)
    seq.assign(size, 0);
    if (size == 0) return true;
    // iterative expansion: every node owns a list of output positions
    struct Work {
        uint16_t node;
        std::vector<uint64_t> pos;
    };
    std::vector<Work> stack;
    Work root;
    root.node = 0;
    root.pos.resize(size);
    for (uint64_t i = 0; i < size; ++i) root.pos[i] = i;
    stack.push_back(std::move(root));
    while (!stack.empty()) {
        Work w = std::move(stack.back());
        stack.pop_back();
        const WtNode& nd = nodes[w.node];
        const bool leaf = nd.child[0] == 0xffff && nd.child[1] == 0xffff;
        if (leaf) {
            if (leaf_symbol[w.node] < 0) return false;
            for (uint64_t p : w.pos) seq[p] = (uint8_t)leaf_symbol[w.node];
            continue;
        }
        if (nd.bv_pos + w.pos.size() > bv.bits) return false;
        Work l, r;
        l.node = nd.child[0];
        r.node = nd.child[1];
        for (size_t i = 0; i < w.pos.size(); ++i) (bv.get(nd.bv_pos + i) ? r : l).pos.push_back(w.pos[i]);
        if ((!l.pos.empty() && l.node >= nnodes) || (!r.pos.empty() && r.node >= nnodes)) return false;
        if (!l.pos.empty()) stack.push_back(std::move(l));
        if (!r.pos.empty()) stack.push_back(std::move(r));
    }
    return true;
}

}  // namespace

// Decodes ONE third-party stream and prints what the readers above make of it (`spumoni dump-sdsl`):
// lets tests feed the readers byte strings derived by hand from the sdsl-lite / r-index sources'
// serialize() functions (tests/test_sdsl_golden.py) instead of round-tripping our own writer.
bool dump_sdsl_stream(const std::string& kind, const std::string& path, std::string& text, std::string& err) {
    std::vector<uint8_t> raw;
    if (!read_whole_file(path, raw)) {
        err = "cannot read " + path;
        return false;
    }
    Cursor c{raw};
    text.clear();
    auto done = [&](bool ok) {
        if (!ok) err = "unexpected layout in the " + kind + " stream";
        else if (c.p != raw.size()) {
            err = "trailing bytes after the " + kind + " stream";
            ok = false;
        }
        return ok;
    };
    if (kind == "int_vector") {
        std::vector<uint64_t> v;
        if (!read_int_vector(c, v)) return done(false);
        for (uint64_t x : v) text += std::to_string(x) + " ";
        return done(true);
    }
    if (kind == "bit_vector") {
        BitVec b;
        if (!read_bit_vector(c, b)) return done(false);
        for (uint64_t i = 0; i < b.bits; ++i) text += b.get(i) ? '1' : '0';
        return done(true);
    }
    if (kind == "sparse_sd") {
        std::vector<uint64_t> ones;
        uint64_t u = 0;
        if (!read_sparse_sd(c, ones, u)) return done(false);
        text = "u " + std::to_string(u) + " ones";
        for (uint64_t x : ones) text += " " + std::to_string(x);
        return done(true);
    }
    if (kind == "wt_huff") {
        std::vector<uint8_t> seq;
        if (!read_wt_huff(c, seq)) return done(false);
        for (uint8_t x : seq) text += std::to_string((unsigned)x) + " ";
        return done(true);
    }
    err = "unknown stream kind " + kind;
    return false;
}

bool load_serialized_index(const std::string& path, bool is_ms, RawIndex& out, std::string& err) {
    std::vector<uint8_t> raw;
    if (!read_whole_file(path, raw)) {
        err = "cannot read " + path;
        return false;
    }
    Cursor c{raw};
    auto bad = [&](const char* what) {
        err = path + ": unexpected layout while reading " + what + " (the serialised-index reader is unverified; "
              "keep the raw run files with `spumoni build -k` instead)";
        return false;
    };
    uint64_t terminator_position = 0, fcount = 0;
    if (!c.get(&terminator_position, 8) || !c.get(&fcount, 8) || fcount != 256) return bad("F");
    uint64_t F[256];
    if (!c.get(F, sizeof F)) return bad("F");
    uint64_t n = 0, R = 0, B = 0;
    if (!c.get(&n, 8) || !c.get(&R, 8) || !c.get(&B, 8) || n == 0 || R == 0 || R > n) return bad("rle_string header");
    std::vector<uint64_t> ones;
    uint64_t u = 0;
    if (!read_sparse_sd(c, ones, u) || u != n) return bad("rle_string.runs");
    std::vector<std::vector<uint64_t>> per_letter(256);
    uint64_t total = 0, total_runs = 0;
    for (int ch = 0; ch < 256; ++ch) {
        if (!read_sparse_sd(c, per_letter[ch], u)) return bad("rle_string.runs_per_letter");
        if (!per_letter[ch].empty() && per_letter[ch].back() + 1 != u) return bad("rle_string.runs_per_letter");
        total += u;
        total_runs += per_letter[ch].size();
    }
    if (total != n || total_runs != R) return bad("rle_string.runs_per_letter totals");
    if (!read_wt_huff(c, out.heads) || out.heads.size() != R) return bad("rle_string.run_heads");
    // run lengths: consecutive differences of the per-letter run-end positions, dealt out in head order
    out.lens.assign(R, 0);
    size_t next[256] = {0};
    for (uint64_t i = 0; i < R; ++i) {
        const uint8_t h = out.heads[i];
        size_t& j = next[h];
        if (j >= per_letter[h].size()) return bad("run heads vs runs_per_letter");
        const uint64_t endp = per_letter[h][j];
        const uint64_t prev = j ? per_letter[h][j - 1] + 1 : 0;
        out.lens[i] = endp + 1 - prev;
        j++;
    }
    out.n = n;
    std::vector<uint64_t> slast;
    if (is_ms && (!read_int_vector(c, slast) || slast.size() != R)) return bad("samples_last");
    // thresholds: the stored (non-zero) thresholds of every letter, in run order; re-expanded to
    // one value per run with 0 for the first run of a letter (thr_bv's own convention)
    out.thr.assign(R, 0);
    std::vector<std::vector<uint64_t>> thr_letter(256);
    for (int ch = 0; ch < 256; ++ch)
        if (!read_sparse_sd(c, thr_letter[ch], u)) return bad("thresholds");
    size_t seen[256] = {0};
    for (uint64_t i = 0; i < R; ++i) {
        const uint8_t h = out.heads[i];
        const size_t j = seen[h]++;
        if (j >= 1 && j - 1 < thr_letter[h].size()) out.thr[i] = thr_letter[h][j - 1];
    }
    if (is_ms) {
        if (!read_int_vector(c, out.ssa) || out.ssa.size() != R) return bad("samples_start");
        out.esa = std::move(slast);
    }
    if (c.p != raw.size()) return bad("end of file (trailing bytes)");
    (void)F;
    (void)terminator_position;
    (void)B;
    return true;
}

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
