// spumoni_index.hpp -- drop-in C++ mirror of the reference's index classes pml_t / ms_t
// (/root/reference/src/compute_ms_pml.cpp:694-838) on top of the C-ABI: same constructor
// arguments, same overloaded matching_statistics(), same get_bwt_stats().  Every reference
// caller of these classes (classify_reads_*, the general-text drivers, the build-time null
// generators :1410-1663) compiles against this header unchanged.  A call per read costs one
// kernel launch; batch callers should use spx_query_batch directly.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/spumoni_gpu.h"
#include "index_files.hpp"

namespace spumoni_host {

class gpu_index_base {
protected:
    spx_index* ix = nullptr;
    [[noreturn]] static void die(const char* msg) {
        std::fprintf(stderr, "\n\033[31mError: \033[0m%s\n\n", msg);  // FATAL_ERROR shape
        std::exit(1);
    }
    void load(const std::string& filename, bool use_doc, bool want_samples, int device) {
        RawIndex raw;
        std::string err;
        if (!load_raw_index(filename, want_samples, raw, err)) die(err.c_str());
        if (use_doc && !load_doc_array(filename + ".doc", raw, err)) die(err.c_str());
        ix = spx_index_from_runs(raw.heads.data(), raw.lens.data(), raw.thr.data(), raw.heads.size(),
                                 want_samples ? raw.ssa.data() : nullptr, want_samples ? raw.esa.data() : nullptr,
                                 use_doc ? raw.doc_start.data() : nullptr, use_doc ? raw.doc_end.data() : nullptr,
                                 0, device);
        if (!ix) die(spx_last_error());
    }
    void query(int mode, const char* read, size_t len, std::vector<size_t>* lengths, std::vector<size_t>* pointers,
               std::vector<size_t>* doc_nums) {
        uint64_t offs[2] = {0, len};
        std::vector<uint32_t> l32(lengths ? len : 0), d32(doc_nums ? len : 0);
        std::vector<uint64_t> p64(pointers ? len : 0);
        if (spx_query_batch(ix, mode, reinterpret_cast<const uint8_t*>(read), offs, 1, lengths ? l32.data() : nullptr,
                            pointers ? p64.data() : nullptr, doc_nums ? d32.data() : nullptr, nullptr, 0, 0) != SPX_OK)
            die(spx_last_error());
        if (lengths) lengths->assign(l32.begin(), l32.end());
        if (pointers) pointers->assign(p64.begin(), p64.end());
        if (doc_nums) doc_nums->assign(d32.begin(), d32.end());
    }

public:
    ~gpu_index_base() { spx_index_free(ix); }
    std::pair<uint64_t, uint64_t> get_bwt_stats() {  // :739-741 / :830-832
        uint64_t n = 0, r = 0;
        spx_index_stats(ix, &n, &r);
        return std::make_pair(n, r);
    }
};

class pml_t : public gpu_index_base {  // :694-746
public:
    pml_t(std::string filename, bool use_doc, bool verbose = false, int device = 0) {
        (void)verbose;
        load(filename, use_doc, false, device);
    }
    void matching_statistics(const char* read, size_t read_length, std::vector<size_t>& lengths) {
        query(SPX_MODE_PML, read, read_length, &lengths, nullptr, nullptr);
    }
    void matching_statistics(const char* read, size_t read_length, std::vector<size_t>& lengths,
                             std::vector<size_t>& doc_nums) {
        query(SPX_MODE_PML, read, read_length, &lengths, nullptr, &doc_nums);
    }
};

class ms_t : public gpu_index_base {  // :748-838; `text_file` replaces <filename>.slp
public:
    ms_t(std::string filename, bool use_doc, bool verbose = false, const std::string& text_file = "", int device = 0) {
        (void)verbose;
        load(filename, use_doc, true, device);
        std::vector<uint8_t> text;
        if (!read_whole_file(text_file.empty() ? filename + ".rawtext" : text_file, text))
            die("the MS index needs the indexed text as a plain file (it replaces <ref>.slp)");
        if (spx_index_set_text(ix, text.data(), text.size(), 0) != SPX_OK) die(spx_last_error());
    }
    void matching_statistics(const char* read, size_t read_length, std::vector<size_t>& lengths,
                             std::vector<size_t>& pointers) {
        query(SPX_MODE_MS, read, read_length, &lengths, &pointers, nullptr);
    }
    void matching_statistics(const char* read, size_t read_length, std::vector<size_t>& lengths,
                             std::vector<size_t>& pointers, std::vector<size_t>& doc_nums) {
        query(SPX_MODE_MS, read, read_length, &lengths, &pointers, &doc_nums);
    }
};

}  // namespace spumoni_host
