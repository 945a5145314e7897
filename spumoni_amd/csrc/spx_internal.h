// spx_internal.h -- private definitions shared by the library's translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <functional>
#include <mutex>
#include <string>

#include "../../include/spumoni_gpu.h"
#include "spx_layout.h"

namespace spx {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define SPX_HIP(call)                                                        \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return spx::hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

// counters written by the walk kernels (one instance per index, device memory)
struct WalkCounters {
    unsigned long long reserved0;
    unsigned long long steps;
    unsigned long long jumps;
    unsigned long long pred_jumps;
    unsigned long long row_loads;
    unsigned long long dir_loads;
    unsigned long long error;      // non-zero: a structural invariant was violated
    unsigned long long pad_;
};

// ---- long reads: exact speculative chunking (spx_walk.hip) -----------------------
// The walk of a read is one dependent chain; a batch with fewer reads than the chip has lanes is
// latency-bound.  Such a batch is cut into chunks -- a read's intersection with an aligned block of
// CHUNK characters of the concatenated input -- and
//   pass 1  every chunk is walked from the default state (pos = n - 1): right for a read's last
//           chunk, speculative for the others;
//   pass 2  for every other chunk, the walk that ended the chunk above it (its recorded end state)
//           is carried on into the chunk, overwriting the speculative results, until its position
//           equals the position the speculative walk recorded at the same character (checkpoints):
//           from there on the two walks are the same walk;
//   pass 3  per read, top down: `length` / `sample` are counters that the walk only adds to until
//           the next jump resets them, so what is left of a wrong start value is a constant
//           offset (a stale value, for the document id) up to the first reset below each seam:
//           patched from the recorded values;
//   What pass 2 wrote is only the truth if the state it entered the chunk with was: true for the
//   chunk below a read's last, and from there down as long as every seam closed (the recorded end
//   state of a chunk whose seam closed IS the true walk's).  A seam left open -- pass 2 reached the
//   chunk's first character still apart from the speculative walk -- breaks the chain: between
//   rounds a scan per read (k_chunk_scan) follows the chain down to the first open seam and has the
//   chunk below it entered again in the next round, now with the state the true walk ended in, down
//   to at least where the earlier, unfounded walk had stopped writing.  After the last round a read
//   whose chain is still broken is walked again the plain way.  Then the bin classifier runs over
//   the finished lengths.
struct ChunkDesc {     // 16 B
    uint64_t gend;     // index (in the concatenated input) one past the chunk's last character
    uint32_t len;      // characters
    uint32_t rd;       // read
};
struct WalkState {     // 32 B: what a walk carries from one character to the next
    uint32_t k0;       // landing target of the last step: run ...
    uint32_t length;   // PML length counter
    uint64_t offp;     // ... and offset (OFF_END: last position of the run)
    uint64_t sample;   // MS pointer counter
    uint32_t doc;      // current document id
    uint32_t flags;    // bit 0: a reset happened since the state's owner started
};
struct SeamRec {       // 64 B, one per chunk that is not its read's last: where and how pass 2 ended
    uint64_t t;        // pass 2 stopped with the results of [t, chunk end) written; t == chunk start: never met
    uint32_t met;      // bit 0: positions met at t (0: ran to the chunk's start); bits 8..: round of pass 2
    uint32_t reset_above;  // pass 2 saw a reset between the chunk's end and t
    WalkState ext;     // pass 2's counters at t
    uint32_t spec_length, spec_doc;
    uint64_t spec_sample;  // the speculative walk's counters at t
};
constexpr uint32_t CKPT_SHIFT = 4;  // a checkpoint every 16 characters

struct ChunkArgs {
    const ChunkDesc* desc;       // chunks, a read's chunks consecutive, last chunk of a read first... no: ascending
    const uint64_t* nchunks;     // device counter
    WalkState* ends;             // per chunk: state after the chunk's first (lowest) character
    WalkState* ckpt;             // per 16 characters of the input: state before character 16 * i - 1
    SeamRec* seams;              // per chunk
    uint8_t* flags;              // per character: bit 0 length / sample were reset, bit 1 doc was set
    uint32_t* read_fail;         // per read: the chain of seams is still broken after the last round
    const uint64_t* chunk_start; // per read: first chunk
    WalkState* reentry;          // per chunk: state to enter it with in a later round (flags: how far down to write)
    uint64_t* vtop;              // per read: chunks from here up hold the truth
    uint32_t* pending;           // per round of pass 2: chunks to enter (round 1: unused)
    uint32_t round;              // pass 2: 1 = every chunk but a read's last, later = the chunks below open seams
    uint32_t last_round;         // pass 2: an open seam now condemns the read to the plain walk
};
constexpr uint32_t CHUNK_TOP = 0x80000000u, CHUNK_BOTTOM = 0x40000000u, CHUNK_ACTIVE = 0x20000000u;  // in ChunkDesc::len
constexpr uint32_t CHUNK_LEN_MASK = 0x1fffffffu;
constexpr int CHUNK_EXTRA_ROUNDS = 3;

struct BatchArgs {
    const uint8_t* seqs;   // readable for round_up(total_chars, 4) + 32 bytes
    const uint64_t* offs;
    uint64_t nreads;
    uint64_t total_chars;
    uint32_t* out_lengths;
    uint64_t* out_pointers;
    uint32_t* out_docs;
    spx_class* out_class;
    uint64_t bin_width;
    uint64_t bin_magic;  // floor(2^64 / bin_width) + 1: division by multiplication in the kernel
    uint64_t max_value_thr;
    WalkCounters* counters;
    uint32_t lanes_per_wave;  // active lanes per wavefront (64 unless the batch is small)
    uint32_t narrow;          // out_lengths / out_docs point to uint16_t arrays (reads < 65536 characters)
    ChunkArgs ch;             // chunked walks only
    const uint32_t* only_flagged;  // plain walk: skip the reads whose flag is 0 (fallback after chunking)
    // reads that were digested on the device a moment ago (spx_digest.hip) and are still where the digestion parked
    // them: read q's characters are seqs[in_starts[q] ..), offs[q + 1] - offs[q] of them, while offs places its results
    // as always (k_walk_fast only; null: seqs[offs[q] ..))
    const uint64_t* in_starts;
    // PML, plain walk: the lengths leave the walk as ONE BIT per character (length == 0, i.e. "reset here":
    // a PML length is the distance to the next reset at or after it, compute_ms_pml.cpp:249-250, 266-276) and
    // k_expand_lengths writes out_lengths from the bits as a stream.  Read q's bits: 16-byte pairs of words
    // [((offs[q] - offs[0]) >> 7) + q, ...) (pairs of different reads never overlap), bit p & 63 of word p >> 6
    // for character p of the read.
    uint64_t* len_mask;
    uint64_t len_mask_pairs;  // 16-byte pairs len_mask holds (sized from total_chars: checked, the caller may be wrong)
};

}  // namespace spx

namespace spx {
// The device arrays of an index once more than one handle uses them: spx_index_clone onto the SAME device hands out a
// second query context -- its own counters, scratch, stream and locks -- over the same (read-only) arrays instead of a
// copy; the arrays go when the last handle does.
struct ArrayOwner {
    int device = 0;
    void* p[10] = {};
    ~ArrayOwner() {
        (void)hipSetDevice(device);
        for (void* a : p)
            if (a) (void)hipFree(a);
    }
};
}  // namespace spx

struct spx_index {
    int device = 0;
    uint64_t n = 0, r = 0;
    bool has_samples = false, has_docs = false;
    spx::Row* rows = nullptr;
    uint32_t* q_alloc = nullptr;  // Q = q_alloc + 1
    spx::JumpRow* dirrows = nullptr;
    char* fat = nullptr;  // slots of DevIndex::fat_stride bytes
    uint32_t* fat_js = nullptr;  // fatjs_count(nfat) entries
    spx::Aux* aux = nullptr;
    uint64_t* ss_by_run = nullptr;
    uint32_t* rundocs = nullptr;
    spx::LetterInfo* letters = nullptr;
    uint8_t* text = nullptr;
    uint64_t n_text = 0;
    // bytes of every device array above, in the order of spx::index_arrays() (saved / cloned as they are)
    static constexpr int NARR = 10;
    uint64_t arr_bytes[NARR] = {};
    std::shared_ptr<spx::ArrayOwner> owner;  // set once the arrays are shared with a same-device clone (else: this handle frees them)
    // host-buffer queries (spx_query_batch*, spx_digest_*batch, spx_query_text_*) run on a stream of the handle's own, so
    // that two handles on one device -- the CLI's two workers per device -- overlap one's copies with the other's kernels
    hipStream_t ctx_stream = nullptr;
    // the digestion's scratch (spx_digest.hip: grow-only, under mu), the stream its last call ran on and what that call enqueued
    static constexpr int NDIGSCR = 9;
    hipEvent_t ev_dig = nullptr;
    hipStream_t dig_stream = nullptr;
    bool dig_used = false;
    spx::DevIndex view{};
    spx::WalkCounters* counters = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr;
    bool have_timing = false;
    hipStream_t last_stream = nullptr;
    uint64_t device_bytes = 0;
    int waves_per_cu = 0; // 0 = default occupancy target
    int occ_blocks[8] = {};  // resident 256-thread blocks per CU, per kernel variant (4..7: k_walk_fast)
    int num_cus = 0;
    int force_lanes_per_wave = 0;  // experiment knob: 0 = automatic
    int force_digest_kernel = 0;   // test knob: 0 automatic, 1 lane-per-read, 2 wavefront-per-read, 3 lane-per-chunk
    int digest_parked = 0;         // digest + walk: 0 the digested reads stay parked when the batch fills the device, 1 never, 2 whenever the walk can take them
    uint8_t charhash[4] = {0, 0, 0, 0};  // -m digestion: 8-bit character hashes of A, C, G, T
    char source_tag[128] = {0};          // spx_index_set_source_tag(): the caller's fingerprint of the index files
    // host-buffer queries of large batches run as a pipeline over chunks of reads: copy in,
    // walk, copy out on three streams (created on first use)
    static constexpr int PIPE_CHUNKS = 10;  // pieces of a pipelined host batch (growing: run_pipelined)
    hipStream_t pipe_s[3] = {nullptr, nullptr, nullptr};
    hipEvent_t pipe_in[PIPE_CHUNKS] = {}, pipe_k[PIPE_CHUNKS] = {};
    std::mutex mu;       // device-buffer queries / options
    std::mutex host_mu;  // host-buffer queries (own the scratch below)
    struct Scratch {
        void* p = nullptr;
        size_t cap = 0;
    } scratch[20], chunk_scr[9], digest_scr[NDIGSCR];  // host-buffer queries (8..19: text output); chunked walks and the length bits (under mu)
    // spx_query_text_begin -> spx_query_text_fetch: the streams' sizes and where they wait on the device
    uint64_t text_bytes[3] = {0, 0, 0};
    uint64_t text_nreads = 0;
    bool text_ready = false;
    spx_class* text_cls_host = nullptr;  // the class records travel with the text (spx_query_text_fetch)
    // A few words the host waits for at the end of a step (stream sizes, the digested total, the walk's counters): written by
    // a kernel into page-locked host memory, NOT copied -- a device-to-host copy of 8 bytes queues on the copy engine behind
    // whatever another query context of the same device is copying out (1.4 ms per super-batch of text), which made
    // spx_query_text_begin of one worker wait for spx_query_text_fetch of the other (profiles/r05_cli_overlap.txt).
    int blocking_sync = 0;          // "blocking_sync" option: the host-buffer text queries wait for their stream on an event with
    hipEvent_t ev_wait = nullptr;   // hipEventBlockingSync (the thread sleeps) instead of hipStreamSynchronize (it spins on a core)
    uint64_t* h_pub = nullptr;      // host address
    uint64_t* h_pub_dev = nullptr;  // the same memory as the device sees it
    int chunk_mode = 0;   // "chunk_mode" option: 0 automatic, 1 never, 2 always
    int chunk_shift = 0;  // "chunk_shift" option: log2 of the chunk size (0 = automatic)
    int chunk_len = 0;    // "chunk_len" option: chunk size in characters (rounded up to 16; 0 = automatic)
    uint64_t last_chunk_len = 0, last_chunk_bound = 0;  // last query: chunk size (0 = plain walk), chunk bound
};

namespace spx {
// the device arrays of an index, in a fixed order (cache file / clone)
enum { A_ROWS, A_DIRROWS, A_FAT, A_FATJ, A_Q, A_AUX, A_SSRUN, A_RUNDOCS, A_LETTERS, A_TEXT };
inline void index_arrays(spx_index* ix, void*** out) {
    out[A_ROWS] = (void**)&ix->rows;
    out[A_DIRROWS] = (void**)&ix->dirrows;
    out[A_FAT] = (void**)&ix->fat;
    out[A_FATJ] = (void**)&ix->fat_js;
    out[A_Q] = (void**)&ix->q_alloc;
    out[A_AUX] = (void**)&ix->aux;
    out[A_SSRUN] = (void**)&ix->ss_by_run;
    out[A_RUNDOCS] = (void**)&ix->rundocs;
    out[A_LETTERS] = (void**)&ix->letters;
    out[A_TEXT] = (void**)&ix->text;
}
// points the kernel-visible view at the index's arrays (scalars of the view are kept)
inline void bind_view(spx_index* ix) {
    DevIndex& v = ix->view;
    v.rows = ix->rows;
    v.dirrows = ix->dirrows;
    v.fat = ix->fat;
    v.Q = ix->q_alloc ? ix->q_alloc + 1 : nullptr;
    v.aux = ix->aux;
    v.ss_by_run = ix->ss_by_run;
    v.rundocs = ix->rundocs;
    v.fat_js = ix->fat_js;
    v.letters = ix->letters;
    v.text = ix->text;
    v.n_text = ix->n_text;
}
// grow-only device scratch of the chunked walk (callers hold ix->mu)
inline int chunk_scratch(spx_index* ix, int slot, size_t bytes, void** out) {
    spx_index::Scratch& sc = ix->chunk_scr[slot];
    if (sc.cap < bytes) {
        if (sc.p) (void)hipFree(sc.p);
        sc.p = nullptr;
        sc.cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        SPX_HIP(hipMalloc(&sc.p, want));
        sc.cap = want;
    }
    *out = sc.p;
    return SPX_OK;
}
// spx_flatten.hip: (re)builds fat / fat_js from letters, Q, dirrows and aux (view.r / nfat / fat_stride set)
int build_fat(spx_index* ix);
// spx_walk.hip: MS text against the index: text[samples_start[k]] must be the head of run k
int launch_text_check(spx_index* ix, unsigned long long* d_bad, hipStream_t stream);
// spx_walk.hip: the indexed text from the MS index (LF chains from every run's first position)
int launch_text_from_index(spx_index* ix, uint8_t* d_text, uint64_t n_text, unsigned long long* d_stuck, hipStream_t stream);
// spx_flatten.hip: builds every device array of `ix` from raw per-run arrays
// that already live on the device.
int flatten_on_device(spx_index* ix, const uint8_t* d_heads, const uint64_t* d_lens,
                      const uint64_t* d_thr, const uint64_t* d_ssa, const uint64_t* d_esa,
                      const uint64_t* d_ds, const uint64_t* d_de, const std::function<void()>& release_inputs = {});
// spx_walk.hip
// *wrote_lengths: the walk wrote the PML lengths itself (k_walk_fast): launch_len_expand is not needed
int launch_walk(spx_index* ix, int mode, const BatchArgs& args, uint64_t total_chars,
                hipStream_t stream, bool* wrote_lengths = nullptr);
int launch_ms_extend(spx_index* ix, const BatchArgs& args, hipStream_t stream);
// PML lengths through one bit per character: prepare_len_mask before the walk (sets args.len_mask; nothing
// to do for MS or a classification-only query), launch_len_expand after it
int prepare_len_mask(spx_index* ix, int mode, BatchArgs& args);
int launch_len_expand(spx_index* ix, const BatchArgs& args, hipStream_t stream);
// long-read batches: the chunked walk (returns SPX_OK and sets *done = false when the batch does not
// qualify and the plain walk should run)
// geom_chars (0: total_chars): the characters the batch is expected to hold when total_chars is only an upper bound
int launch_walk_chunked(spx_index* ix, int mode, const BatchArgs& args, uint64_t total_chars, hipStream_t stream,
                        bool* done, uint64_t geom_chars = 0);
// spx_text.hip: the values lines of one vector as text (count + scan, then -- the stream's size known -- the digits)
int launch_text_count(const void* d_vals, int value_bytes, const uint64_t* d_offs, const uint32_t* d_gap, uint64_t nreads,
                      uint64_t* d_line_bytes, uint64_t* d_line_start, void* d_cub, size_t cub_bytes, hipStream_t st);
size_t text_scan_bytes(uint64_t nreads);
int launch_text_write(const void* d_vals, int value_bytes, const uint64_t* d_offs, const uint32_t* d_gap, uint64_t nreads,
                      const uint64_t* d_line_start, char* d_text, hipStream_t st);
// spx_digest.hip: d_out_offs gets nreads + 1 offsets, d_out the digested reads (capacity is the
// caller's business: spx_digest_capacity)
int launch_digest(spx_index* ix, int kind, uint32_t k, uint32_t w, const uint8_t* d_seqs, const uint64_t* d_offs,
                  uint64_t nreads, uint64_t total_chars, uint8_t* d_out, uint64_t* d_out_offs, hipStream_t stream,
                  bool* parked = nullptr);
}  // namespace spx
