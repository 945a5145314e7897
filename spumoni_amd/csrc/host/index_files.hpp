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

}  // namespace spumoni_host
