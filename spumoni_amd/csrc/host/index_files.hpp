// index_files.hpp -- readers for the files `spumoni run` consumes.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace spumoni_host {

struct RawIndex {
    std::vector<uint8_t> heads;   // <prefix>.bwt.heads        (include/ms_rle_string.hpp:226-253)
    std::vector<uint64_t> lens;   // <prefix>.bwt.len  5-byte LE (ms_rle_string.hpp:246-247)
    std::vector<uint64_t> thr;    // <prefix>.thr_pos  5-byte LE (include/thresholds_ds.hpp:393-417)
    std::vector<uint64_t> ssa;    // <prefix>.ssa (left,right) pairs -> right ? right-1 : n-1
    std::vector<uint64_t> esa;    //                              (src/compute_ms_pml.cpp:404-436)
    std::vector<uint64_t> doc_start, doc_end;  // <prefix>.doc  (src/doc_array.cpp:184-201)
    uint64_t n = 0;
};

// Each returns false and fills `err` on failure.
bool load_raw_index(const std::string& prefix, bool want_samples, RawIndex& out, std::string& err);
// DocumentArray::load: u64 num_entries, then two sdsl::int_vector<> streams
// (u64 size in bits, u8 width, ceil(bits/64) little-endian 64-bit words).
bool load_doc_array(const std::string& path, RawIndex& out, std::string& err);
// EmpNullDatabase::load (src/emp_null_database.cpp:104-110): u64 num_values, f64
// ks_stat_threshold, f64 mean_null_stat, f64 percentile_value, int_vector<> (not needed).
bool load_null_db(const std::string& path, double& percentile_value, std::string& err);
// Writers of the two small sdsl-framed files (used by tests and tooling).
bool write_doc_array(const std::string& path, const std::vector<uint64_t>& doc_start,
                     const std::vector<uint64_t>& doc_end, std::string& err);
bool write_null_db(const std::string& path, double percentile_value, const std::vector<uint64_t>& stats,
                   std::string& err);
bool read_whole_file(const std::string& path, std::vector<uint8_t>& out);

// Reader of the SERIALISED index the reference's `run` loads: <ref>.thrbv.spumoni (PML) /
// <ref>.thrbv.ms (MS)  (pml_pointers::load, src/compute_ms_pml.cpp:222-229; ms_pointers::load,
// :551-562).  First-party framing: u64 terminator_position, F (common.hpp:489-495), then
// ri::rle_string, [samples_last], 256 x thresholds sparse_sd_vector, [samples_start].
// The third-party streams inside (ri::rle_string / ri::sparse_sd_vector / ri::huff_string over
// sdsl::sd_vector, select_support_mcl, wt_huff, int_vector) are parsed from the layouts
// documented in index_files.cpp.  UNVERIFIED against an upstream-built file: neither library
// nor such a file is available offline; a structural mismatch is reported as an
// error, never guessed around.  The result is the raw per-run arrays.
bool load_serialized_index(const std::string& path, bool is_ms, RawIndex& out, std::string& err);
// one third-party stream (kind: int_vector | bit_vector | sparse_sd | wt_huff) decoded to text: test hook
bool dump_sdsl_stream(const std::string& kind, const std::string& path, std::string& text, std::string& err);

}  // namespace spumoni_host
