// spumoni_main.cpp -- `spumoni run` on MI355X: the reference's command line
// (/root/reference/src/spumoni.cpp:163-206, 733-779), option validation
// (include/spumoni_main.hpp:252-329) and stderr/stdout log lines
// (src/compute_ms_pml.cpp:1305-1382), on top of libspumoni_gpu.so.
//
// Differences that are ours, all additive:
//   SPUMONI_GPUS=0,1,..   devices to use (default 0); reads are sharded, index replicated
//   SPUMONI_TEXT=<file>   plain indexed text for the MS length extension (optional: without it the text
//                         <ref>.slp encodes is rebuilt from the MS index itself)
//   the index is read from the raw run files kept by `spumoni build -k`
//   (<ref>.bwt.heads/.bwt.len/.thr_pos[/.ssa/.esa]); -t is accepted and ignored
//   (output is always in input order, the reference's -t 1 order).
#include <getopt.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <thread>

#include "classify.hpp"
#include "index_files.hpp"
#include "reads.hpp"

#define SPUMONI_VERSION "2.0.9"  // include/spumoni_main.hpp:24 (the version this build mirrors)

using namespace spumoni_host;

#define FORCE_LOG(func, ...)                                      \
    do {                                                          \
        std::fprintf(stderr, "\033[32m[%s] \033[0m", func);       \
        std::fprintf(stderr, __VA_ARGS__);                        \
        std::fprintf(stderr, "\n");                               \
    } while (0)
#define STATUS_LOG(x, ...)                                        \
    do {                                                          \
        std::fprintf(stderr, "\033[32m[%s] \033[0m", x);          \
        std::fprintf(stderr, __VA_ARGS__);                        \
        std::fprintf(stderr, " ... ");                            \
    } while (0)
#define DONE_LOG(x)                                               \
    do {                                                          \
        auto sec = std::chrono::duration<double>(x);              \
        std::fprintf(stderr, "done.  (%.3f sec)\n", sec.count()); \
    } while (0)

static int is_file(const std::string& path) {
    struct stat st;
    return ::stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
static bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

static int spumoni_run_usage() {
    std::fprintf(stderr, "spumoni run - Uses a spumoni index to compute MS/PML of patterns w.r.t. a reference.\n");
    std::fprintf(stderr, "Usage: spumoni run [options]\n\n");
    std::fprintf(stderr, "Options:\n");
    std::fprintf(stderr, "\tGeneral options:\n");
    std::fprintf(stderr, "\t%-35sprints this usage message\n", "-h, --help");
    std::fprintf(stderr, "\t%-25s%-10snumber of helper threads (default: 1)\n\n", "-t, --threads", "[INT]");
    std::fprintf(stderr, "\tInput/output options:\n");
    std::fprintf(stderr, "\t%-25s%-10soutput prefix used for index\n", "-r, --ref", "[FILE]");
    std::fprintf(stderr, "\t%-25s%-10spath to patterns file that will be used.\n", "-p, --pattern", "[FILE]");
    std::fprintf(stderr, "\t%-25s%-10suse index to compute MSs\n", "-M, --MS", "");
    std::fprintf(stderr, "\t%-25s%-10suse index to compute PMLs\n", "-P, --PML", "");
    std::fprintf(stderr, "\t%-25s%-10spattern file is general text (default: FASTA)\n", "-g, --general", "");
    std::fprintf(stderr, "\t%-25s%-10suse document array to get assignments\n", "-d, --doc-array", "");
    std::fprintf(stderr, "\t%-25s%-10swrite out the classifications in a report file\n", "-c, --classify", "");
    std::fprintf(stderr, "\t%-25s%-10ssize of region in bp for classification (default: 150)\n\n", "-w, --window", "[INT]");
    std::fprintf(stderr, "\tMinimizer options:\n");
    std::fprintf(stderr, "\t%-25s%-10sturn off minimizer digestion of reads (default: on)\n", "-n, --no-digest", "");
    std::fprintf(stderr, "\t%-25s%-10suse alphabet-promoted minimizers\n", "-m, --minimizer-alphabet", "");
    std::fprintf(stderr, "\t%-25s%-10suse DNA-letter based minimizers\n", "-a, --dna-minimizer", "");
    std::fprintf(stderr, "\t%-25s%-10ssmall window size (k) for finding minimizers (default: 4)\n", "-K, --small-window", "[INT]");
    std::fprintf(stderr, "\t%-25s%-10slarge window size (w) for finding minimizers (default: 11)\n\n", "-W, --large-window", "[INT]");
    return 0;
}

struct CliOptions : RunOptions {
    bool ms_requested = false, pml_requested = false;
    int result_type = 2;  // 0 MS, 1 PML, 2 NOT_CHOSEN
    int ref_type = 2;     // 0 FASTA, 1 MINIMIZER, 2 NOT_SET
};

static void parse_run_options(int argc, char** argv, CliOptions* opts) {
    // option table as in the reference, including `--dna-minimizer` being wired to 't'
    // (src/spumoni.cpp:179)
    static struct option long_options[] = {{"help", no_argument, NULL, 'h'},
                                           {"threads", required_argument, NULL, 't'},
                                           {"ref", required_argument, NULL, 'r'},
                                           {"pattern", required_argument, NULL, 'p'},
                                           {"MS", no_argument, NULL, 'M'},
                                           {"PML", no_argument, NULL, 'P'},
                                           {"general-text", no_argument, NULL, 'g'},
                                           {"doc-array", no_argument, NULL, 'd'},
                                           {"classify", no_argument, NULL, 'c'},
                                           {"window", required_argument, NULL, 'w'},
                                           {"no-digest", no_argument, NULL, 'n'},
                                           {"minimizer-alphabet", no_argument, NULL, 'm'},
                                           {"dna-minimizer", no_argument, NULL, 't'},
                                           {"small-window", required_argument, NULL, 'K'},
                                           {"large-window", required_argument, NULL, 'W'},
                                           {0, 0, 0, 0}};
    int long_index = 0;
    for (int c; (c = getopt_long(argc, argv, "hr:p:MPt:dcnmaK:W:w:g", long_options, &long_index)) >= 0;) {
        switch (c) {
            case 'h': spumoni_run_usage(); std::exit(1);
            case 'r': opts->ref_file.assign(optarg); break;
            case 'p': opts->pattern_file.assign(optarg); break;
            case 'M': opts->ms_requested = true; break;
            case 'P': opts->pml_requested = true; break;
            case 'c': opts->write_report = true; break;
            case 'm': opts->use_promotions = true; break;
            case 'a': opts->use_dna_letters = true; break;
            case 'n': opts->min_digest = false; break;
            case 'K': opts->k = std::max(std::atoi(optarg), 1); break;
            case 'W': opts->w = std::max(std::atoi(optarg), 1); break;
            case 'w': opts->bin_size = std::max(std::atoi(optarg), 1); break;
            case 'g': opts->is_general_text = true; break;
            case 't': opts->threads = std::max(optarg ? std::atoi(optarg) : 1, 1); break;
            case 'd': opts->use_doc = true; break;
            default: spumoni_run_usage(); std::exit(1);
        }
    }
}

static void populate_types(CliOptions& o) {  // include/spumoni_main.hpp:252-265
    if (o.ms_requested && !o.pml_requested) o.result_type = 0;
    if (!o.ms_requested && o.pml_requested) o.result_type = 1;
    bool is_fasta = (is_file(o.ref_file + ".fa") || is_file(o.ref_file + ".fasta") || is_file(o.ref_file + ".fna"));
    bool is_min = is_file(o.ref_file + ".bin");
    if (is_fasta && !is_min) o.ref_type = 0;
    if (!is_fasta && is_min) o.ref_type = 1;
}

static void validate(const CliOptions& o) {  // include/spumoni_main.hpp:267-329
    if (o.ref_file == "" || o.pattern_file == "") fatal_warning("Both a reference file (-r) and pattern file (-p) must be provided.");
    if (o.result_type == 2) fatal_warning("An output type with -M or -P must be specified, only one can be used at a time.");
    std::string extension = o.use_promotions ? ".bin" : ".fa";
    if (!is_file(o.ref_file + extension))
        fatal_error("The following path is not valid: %s (remember to only specify output prefix)", (o.ref_file + extension).data());
    if (!is_file(o.pattern_file)) fatal_error("The following path is not valid: %s", o.pattern_file.data());
    if (!o.is_general_text && o.ref_type == 2)
        fatal_error("Reference file is an unrecognized type. It needs to be a\n"
                    "       FASTA file or binary file produced by spumoni build.");
    if (!o.is_general_text && !ends_with(o.pattern_file, ".fa") && !ends_with(o.pattern_file, ".fasta") &&
        !ends_with(o.pattern_file, ".fna"))
        fatal_error("The pattern file provided does not appear to be a FASTA\n"
                    "       file, please convert to FASTA and re-run.");
    if (o.is_general_text && o.min_digest) fatal_warning("For general-text querying, minimizer digestion must be turned off with -n.");
    if (o.is_general_text && o.threads > 1) fatal_warning("For general-text querying, multi-threading is not available.");
    if (o.is_general_text && o.write_report) fatal_warning("For general-text querying, classification is not available.");
    if (o.use_doc && !is_file(o.ref_file + extension + ".doc"))
        fatal_warning("document array file (%s) is not present, so it cannot be used.", (o.ref_file + extension + ".doc").data());
    // index: the reference checks <ref>.thrbv.ms / .thrbv.spumoni; this build reads the raw run files
    const std::string base = o.ref_file + extension;
    const bool have_raw = is_file(base + ".bwt.heads") && is_file(base + ".bwt.len") && is_file(base + ".thr_pos") &&
                          (o.result_type != 0 || (is_file(base + ".ssa") && is_file(base + ".esa")));
    if (!have_raw) {
        // the reference's own check (include/spumoni_main.hpp:304-313): the serialised index
        const std::string ser = base + (o.result_type == 0 ? ".thrbv.ms" : ".thrbv.spumoni");
        if (!is_file(ser))
            fatal_warning("The index required for this computation is not available, please use spumoni build.");
    }
    if (o.k > 4) fatal_warning("small window size (k) cannot be larger than 4 characters.");
    if (o.w < o.k) fatal_warning("large window size (w) should be larger than the small window size (k)");
    if (o.min_digest) {
        if (o.use_promotions && o.use_dna_letters) fatal_error("Only one type of minimizer can be specified from either -m or -a.");
        if (!o.use_promotions && !o.use_dna_letters) fatal_error("A minimizer type must be specified using -m or -a.");
    } else {
        if (o.use_promotions || o.use_dna_letters)
            fatal_error("A minimizer type should not be specified if intending not to use minimizer digestion.");
    }
    if (o.bin_size < 50 || o.bin_size > 400)
        fatal_warning("the bin size used is not optimal. Re-run using a value between 50 and 400.");
}

static std::vector<int> all_devices_twice() {
    std::vector<int> v;
    const int nd = std::max(1, spx_device_count());
    for (int d = 0; d < nd; ++d) {
        v.push_back(d);
        v.push_back(d);
    }
    return v;
}

static int run_spumoni(CliOptions& o) {  // run_spumoni_main / run_spumoni_ms_main (:1305-1382)
    const char* tag = o.ms ? "compute_ms" : "compute_pml";
    IndexSet set;
    STATUS_LOG(o.ms ? "ms_construct" : "pml_construct", o.ms ? "loading the MS index" : "loading the PML index");
    auto start_time = std::chrono::system_clock::now();
    // the reads file is mapped and its lines are indexed on another thread while the index loads
    std::unique_ptr<ReadFile> reads;
    OutputFiles* outputs = o.is_general_text ? nullptr : new_output_files();
    std::string reads_err;
    // ... and the output files' tails are prepared as memory (classify.cpp: prepare_outputs) on a third: the value streams,
    // sized from the reads file's size, from the very start; the report once the reads are counted
    std::thread outputs_loader;
    if (!o.is_general_text)
        outputs_loader = std::thread([&] {
            struct stat st;
            prepare_outputs(outputs, o, ::stat(o.pattern_file.c_str(), &st) == 0 ? (uint64_t)st.st_size : 0);
        });
    std::thread reads_loader;
    if (!o.is_general_text)
        reads_loader = std::thread([&] {
            try {
                reads.reset(new ReadFile(o.pattern_file, (unsigned)o.format_threads));
                std::thread rep([&] { prepare_report(outputs, o, reads->first_char() == '@' ? reads->lines() / 4 : reads->lines() / 2); });
                reads->precompute_ranges(1000);  // reader.loadBatch(input_file, 1000)   (compute_ms_pml.cpp:903)
                rep.join();
            } catch (const std::exception& e) {
                reads_err = e.what();
            }
        });
    size_t auto_devices = 0;
    if (o.devices.empty()) {  // no SPUMONI_GPUS: one device -> three workers on it, several -> all of them, two workers each
        o.devices = spx_device_count() > 1 ? all_devices_twice() : std::vector<int>{0, 0, 0};
        auto_devices = o.devices.size() / 2;
    }
    std::thread pinned_loader;
    if (!o.is_general_text) pinned_loader = std::thread([&] { prepare_pinned_pool(o, std::max<size_t>(o.devices.size(), 1)); });
    set.load(o);
    if (reads_loader.joinable()) reads_loader.join();
    if (outputs_loader.joinable()) outputs_loader.join();
    if (pinned_loader.joinable()) pinned_loader.join();
    // (the cores are shared out per distinct device: run_main sized them for one; the reads' loader is done with the count)
    if (auto_devices > 1 && o.threads <= 1)
        o.format_threads = std::max(1u, std::min(16u * (unsigned)auto_devices, std::thread::hardware_concurrency()));
    if (!reads_err.empty()) fatal_error("%s", reads_err.c_str());
    // (test-only: hold the run here, the files prepared and nothing processed -- tests/test_host_harness_cpu.py interrupts it)
    if (const char* e = std::getenv("SPUMONI_TEST_STALL_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(std::atoi(e)));
    DONE_LOG((std::chrono::system_clock::now() - start_time));
    std::cout << std::endl;
    if (o.use_promotions)
        FORCE_LOG(tag, "input reads will digested using promoted minimizer alphabet (k=%d, w=%d)", (int)o.k, (int)o.w);
    else if (o.use_dna_letters)
        FORCE_LOG(tag, "input reads will digested using DNA minimizer alphabet (k=%d, w=%d)", (int)o.k, (int)o.w);
    else
        FORCE_LOG(tag, "input reads will be used directly, no minimizer digestion");
    if (o.use_promotions || o.use_dna_letters) {
        // the digestion restates dnbaker/bonsai @5273b81a92 (window rule, Lex order, and for -m
        // the 8-bit character hashes) without bonsai's source or any of its output to check against.
        // Say so, loudly, every time -- against an index digested by upstream `spumoni build` the results
        // may be wrong without any other sign.
        const char* pin = std::getenv("SPUMONI_CHARHASH");
        std::fprintf(stderr,
                     "\033[1m\033[33m[spumoni-gpu] WARNING:\033[0m minimizer digestion (-m / -a) is a restatement of bonsai's\n"
                     "              RollingHasher / Encoder that has NOT been verified against bonsai output (DESIGN.md 4.4).\n"
                     "              PML/MS values are only guaranteed for an index whose reference was digested by THIS\n"
                     "              package (spumoni_amd/build_index.py -m/-a).%s\n",
                     o.use_promotions ? (pin ? "  Character hashes pinned by SPUMONI_CHARHASH."
                                             : "  Pin the -m character hashes with SPUMONI_CHARHASH=<A>,<C>,<G>,<T>.")
                                      : "");
        if (pin && o.use_promotions) {
            unsigned v[4] = {0, 0, 0, 0};
            if (std::sscanf(pin, "%u,%u,%u,%u", &v[0], &v[1], &v[2], &v[3]) != 4)
                fatal_error("SPUMONI_CHARHASH must be four comma-separated byte values (A,C,G,T)");
            const int64_t packed = (int64_t)((v[0] & 255) | ((v[1] & 255) << 8) | ((v[2] & 255) << 16) | ((uint64_t)(v[3] & 255) << 24));
            for (spx_index* p : set.ix)
                if (spx_set_option(p, "minimizer_charhash", packed) != SPX_OK) fatal_error("%s", spx_last_error());
        }
    }
    start_time = std::chrono::system_clock::now();
    STATUS_LOG(tag, o.ms ? "processing the reads" : "processing the patterns");
    size_t num_reads = o.is_general_text ? classify_general_reads(set, o) : classify_reads(set, o, reads.get(), outputs);
    DONE_LOG((std::chrono::system_clock::now() - start_time));
    FORCE_LOG(tag, "finished processing %d reads. results are saved in *.%s file.", (int)num_reads,
              o.ms ? "lengths" : "pseudo_lengths");
    std::cout << std::endl;
    remove_leftovers();  // (a large output of an earlier run, moved aside when its name was taken: gone before we are)
    if (std::getenv("SPX_FREE_TRACE")) std::fprintf(stderr, "[spumoni] run_spumoni returns\n");
    return 0;
}

static int run_main(int argc, char** argv) {
    if (argc == 1) return spumoni_run_usage();
    CliOptions o;
    parse_run_options(argc, argv, &o);
    populate_types(o);
    validate(o);
    o.ref_file += o.use_promotions ? ".bin" : ".fa";  // spumoni.cpp:744-747
    o.ms = (o.result_type == 0);
    // SPUMONI_GPUS: the device of every WORKER (a host thread that feeds a device from the one queue of parsed
    // super-batches).  Two entries that name the same device are two query contexts over one copy of the index: one's
    // copies over PCIe run under the other's kernels.  Default (resolved in run_spumoni, beside the threads that prepare the
    // files: counting the devices starts the HIP runtime): one visible device -> "0,0,0" (three keep the copy engines busy:
    // files complete after 0.062-0.072 s against 0.076-0.081 with two, profiles/r05_cli_overlap.txt); several -> every visible
    // device, twice (the index replicated, the reads dealt to the workers: SURVEY 8(e)); "all" asks for the latter by name.
    o.devices.clear();
    if (const char* g = std::getenv("SPUMONI_GPUS")) {
        o.devices.clear();
        if (std::strcmp(g, "all") == 0) {
            o.devices = all_devices_twice();
        } else {
            std::stringstream ss(g);
            std::string tok;
            while (std::getline(ss, tok, ','))
                if (!tok.empty()) o.devices.push_back(std::atoi(tok.c_str()));
        }
        if (o.devices.empty()) o.devices.push_back(0);
    }
    if (const char* t = std::getenv("SPUMONI_TEXT")) o.text_file = t;
    // (PML only: with -M the lengths are what the report is made from and every stream is written -- the host-formatting
    // path used to leave <pattern>.lengths empty there while the device-text path wrote it)
    if (const char* t = std::getenv("SPUMONI_REPORT_ONLY")) o.report_only = o.write_report && !o.ms && std::atoi(t) != 0;
    // characters of reads per super-batch (32 MB; tests: a few thousand, so that a small input runs as many super-batches
    // through the queue, the workers and the ordered writer)
    // (MS mode writes ~13 bytes of text per character where PML writes ~2.5: a quarter of the characters is the same bytes per
    // super-batch, and the page-locked buffers of the slots -- 12 bytes per character for the pointers -- stay small)
    if (o.ms) o.super_batch_chars = 8u << 20;
    if (const char* t = std::getenv("SPUMONI_SUPER_BATCH")) o.super_batch_chars = std::max<size_t>(1000, std::strtoull(t, nullptr, 10));
    // -t: the reference's helper threads walk the index; here the GPU does, and the threads
    // format the output text instead (default: up to 16 of the available cores)
    // (sixteen per distinct device: the feeders' parsing, the headers and the report are what the host does per read, and
    // every device wants its share of cores for them -- SURVEY 8(e): one feeder thread group per GPU)
    {
        std::vector<int> distinct = o.devices;
        std::sort(distinct.begin(), distinct.end());
        distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
        const unsigned want = 16u * (unsigned)std::max<size_t>(1, distinct.size());
        o.format_threads = o.threads > 1 ? o.threads : std::max(1u, std::min(want, std::thread::hardware_concurrency()));
    }
    return run_spumoni(o);
}

// debugging aid used by the CPU tests: dump the reads the batch segmentation yields
static int dump_reads_main(int argc, char** argv) {
    if (argc < 2) return 1;
    ReadFile in(argv[1]);
    std::vector<ParsedRead> batch;
    size_t nb = 0;
    while (in.next_batch(1000, batch)) {
        std::printf("#batch %zu\n", nb++);
        for (auto& rd : batch) {
            const std::string_view sq = rd.seq();
            std::printf("%.*s\t%.*s\n", (int)rd.id.size(), rd.id.data(), (int)sq.size(), sq.data());
        }
    }
    return 0;
}

// debugging aid used by the CPU tests: dump what the serialised-index reader recovers
static int dump_index_main(int argc, char** argv) {
    if (argc < 3) return 1;
    RawIndex raw;
    std::string err;
    if (!load_serialized_index(argv[1], argv[2][0] == 'M', raw, err)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    std::printf("n %llu r %zu\n", (unsigned long long)raw.n, raw.heads.size());
    auto dump = [](const char* tag, const std::vector<uint64_t>& v) {
        std::printf("%s", tag);
        for (uint64_t x : v) std::printf(" %llu", (unsigned long long)x);
        std::printf("\n");
    };
    std::printf("heads");
    for (uint8_t h : raw.heads) std::printf(" %u", (unsigned)h);
    std::printf("\n");
    dump("lens", raw.lens);
    dump("thr", raw.thr);
    if (!raw.ssa.empty()) {
        dump("ssa", raw.ssa);
        dump("esa", raw.esa);
    }
    return 0;
}

// debugging aid used by the CPU tests: decode one sdsl / r-index stream
static int dump_sdsl_main(int argc, char** argv) {
    if (argc < 3) return 1;
    std::string text, err;
    if (!dump_sdsl_stream(argv[1], argv[2], text, err)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    std::printf("%s\n", text.c_str());
    return 0;
}

static int spumoni_usage() {
    std::fprintf(stderr, "SPUMONI has different sub-commands to run which can used as follows:\n");
    std::fprintf(stderr, "Usage: spumoni <command> [options]\n\n");
    std::fprintf(stderr, "Commands:\n");
    std::fprintf(stderr, "\tbuild\tbuilds the index needed to compute MS or PMLs for a specified reference.\n");
    std::fprintf(stderr, "\trun\tcomputes MSs or PMLs for patterns against already built SPUMONI index.\n\n");
    return 1;
}

int main(int argc, char** argv) {
    if (argc > 2 && std::strcmp(argv[1], "dump-reads") == 0) return dump_reads_main(argc - 1, argv + 1);
    if (argc > 3 && std::strcmp(argv[1], "dump-index") == 0) return dump_index_main(argc - 1, argv + 1);
    if (argc > 3 && std::strcmp(argv[1], "dump-sdsl") == 0) return dump_sdsl_main(argc - 1, argv + 1);
    std::fprintf(stderr, "\n\033[1m\033[31mSPUMONI version: %s \033[0m\n\n", SPUMONI_VERSION);
    if (argc > 1) {
        if (std::strcmp(argv[1], "build") == 0)
            fatal_error("`spumoni build` is not part of the MI355X run-path package: build the index with the\n"
                        "       reference (keep the raw files with -k) and query it here.");
        if (std::strcmp(argv[1], "run") == 0) {
            const int rc = run_main(argc - 1, argv + 1);
            if (std::getenv("SPX_FREE_TRACE")) std::fprintf(stderr, "[spumoni] main returns %d\n", rc);
            return rc;
        }
    }
    return spumoni_usage();
}
