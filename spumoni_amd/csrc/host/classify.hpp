// classify.hpp -- host harness above the C-ABI: our mirror of the reference's run drivers
// (classify_reads_pml / classify_reads_ms / classify_general_reads_*,
// /root/reference/src/compute_ms_pml.cpp:845-1297) with ordered writers.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/spumoni_gpu.h"

namespace spumoni_host {

struct RunOptions {  // SpumoniRunOptions, include/spumoni_main.hpp:233-250
    std::string ref_file;      // index prefix INCLUDING the .fa/.bin extension (spumoni.cpp:744-747)
    std::string pattern_file;
    bool ms = false;           // result_type == MS
    bool use_doc = false;
    bool write_report = false;
    bool min_digest = true;
    bool use_promotions = false;
    bool use_dna_letters = false;
    bool is_general_text = false;
    size_t k = 4, w = 11, bin_size = 150, threads = 1;
    // ours (additive): devices to use, plain text for the MS length extension, batch size
    std::vector<int> devices{0};
    std::string text_file;
    size_t super_batch_chars = 32u << 20;  // (round 5: 32 MB -- the pipeline fills sooner; 16 .. 64 MB measure the same to a few percent, profiles/r05_cli_e2e.txt)
    size_t format_threads = 1;  // host threads that turn results into text (-t, or the core count)
    bool report_only = false;   // SPUMONI_REPORT_ONLY=1 with -P -c: <pattern>.pseudo_lengths stays empty (PML only)
};

// One spx_index per device: flattened (or read from the flat-layout cache) once, then replicated
// device to device.
class IndexSet {
public:
    IndexSet() = default;
    ~IndexSet();
    IndexSet(const IndexSet&) = delete;
    IndexSet& operator=(const IndexSet&) = delete;
    // loads <ref_file>.bwt.heads/.bwt.len/.thr_pos[/.ssa/.esa][+ <ref_file>.doc][+ text]
    void load(const RunOptions& o);
    std::vector<spx_index*> ix;
    uint64_t n = 0, r = 0;
    bool from_cache = false;  // the flat arrays came from <ref>.{pml,ms}[.doc].spx
};

// compute_ms_pml.cpp:871-875 (PML) / :1061-1063 (MS)
size_t max_value_threshold(double percentile_value, bool is_pml, bool use_promotions, bool use_dna_letters);

// FASTA/FASTQ driver (classify_reads_pml :845-1034, classify_reads_ms :1036-1217).
// Writes <pattern>.pseudo_lengths | .lengths + .pointers, [.doc_numbers], [.report];
// returns the number of reads processed.  Output order = input order (the reference's -t 1).
// preloaded: the reads file, already mapped (while the index was loading); nullptr = map it here.
class ReadFile;
// The output files, opened under temporary names with their tails prepared as memory -- allocated, mapped, populated and
// (the value streams) page-locked for the device -- while the index loads: the run then lands its text in the files' pages
// without a write() (classify.cpp, OutFile).  reads_file_bytes / reads_guess size the estimate; what does not fit is written
// the ordinary way.  Returns an object without prepared files where that does not apply (general text, SPUMONI_HOST_FORMAT,
// SPUMONI_MAP_OUTPUT=0, small outputs, a file system that cannot map).
struct OutputFiles;
OutputFiles* new_output_files();
void prepare_outputs(OutputFiles* out, const RunOptions& o, uint64_t reads_file_bytes);    // the value streams (from the start)
void prepare_report(OutputFiles* out, const RunOptions& o, uint64_t reads_guess);          // the report (reads counted)
size_t classify_reads(IndexSet& set, const RunOptions& o, ReadFile* preloaded = nullptr, OutputFiles* prepared = nullptr);
// page-locked buffers for the slots of classify_reads, made ahead of time (a helper thread, while the index loads)
void prepare_pinned_pool(const RunOptions& o, size_t ndev);
// general-text driver (:1219-1297): reads separated by \x01, named read_<i>
size_t classify_general_reads(IndexSet& set, const RunOptions& o);

}  // namespace spumoni_host
