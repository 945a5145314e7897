/*
 * spumoni_gpu.h -- C-ABI of libspumoni_gpu.so: the MI355X (gfx950) drop-in for
 * the `spumoni run` hot path of oma219/spumoni v2.0.9.
 *
 * The reference has no plugin/FFI layer; the narrowest seam on its hot path is
 *     pml_t::matching_statistics(read, len, lengths[, doc_nums])
 *                                   (/root/reference/src/compute_ms_pml.cpp:730-737)
 *     ms_t::matching_statistics(read, len, lengths, pointers[, doc_nums])
 *                                   (src/compute_ms_pml.cpp:795-828)
 * plus the constructors pml_t(prefix, use_doc, verbose) (:700) / ms_t(...) (:755)
 * and get_bwt_stats() (:739-741).  This header is the batch form of exactly
 * that contract: plain pointers and sizes, no C++ / torch / HIP types.  The
 * C++ mirror of pml_t / ms_t on top of it lives in spumoni_amd/csrc/host/.
 *
 * Conventions
 *   - every function returns 0 on success or a negative SPX_E* code; the
 *     message is available from spx_last_error() (thread-local).
 *   - there is NO CPU fallback: if no gfx950 device is usable every entry
 *     point that needs one fails with SPX_E_NODEVICE.
 *   - an spx_index lives on ONE device (one process / one host thread per GPU,
 *     index replicated per GPU -- SURVEY.md 8(e)); queries on one index are
 *     serialised internally, different indexes are independent.
 */
#ifndef SPUMONI_GPU_H
#define SPUMONI_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPX_OK 0
#define SPX_E_ARG (-1)      /* bad argument                                    */
#define SPX_E_IO (-2)       /* file missing / malformed                        */
#define SPX_E_NODEVICE (-3) /* no usable HIP device (no CPU fallback exists)   */
#define SPX_E_HIP (-4)      /* HIP runtime error                               */
#define SPX_E_FORMAT (-5)   /* index arrays violate a structural invariant     */
#define SPX_E_UNSUPPORTED (-6)

/* what a query computes: replaces the choice between pml_t and ms_t */
#define SPX_MODE_PML 0 /* pml_pointers::_query   (compute_ms_pml.cpp:238-340)  */
#define SPX_MODE_MS 1  /* ms_pointers::_query    (compute_ms_pml.cpp:571-682)  */

typedef struct spx_index spx_index;

/* per-read result of the bin-max classifier (compute_ms_pml.cpp:969-995,
 * 1150-1176): bins whose max >= max_value_thr, bins below, and the sum of the
 * bin maxima (the report prints sum/(above+below) and FOUND iff
 * above/(above+below) > 0.5).                                                */
typedef struct spx_class {
    uint64_t sum_max_bin_values;
    uint32_t bins_above;
    uint32_t bins_below;
} spx_class;

/* walk statistics of the last query on an index (ours, for the roofline
 * model of SURVEY.md 8(d)): characters searched, steps that took the
 * mismatch branch (:251), steps that took the predecessor branch (:270),
 * and index touches by kind.                                                 */
typedef struct spx_walk_stats {
    uint64_t steps;
    uint64_t jumps;
    uint64_t pred_jumps;
    uint64_t row_loads;   /* landing-row loads (>= steps)                      */
    uint64_t dir_loads;   /* per-letter directory loads (count table + window) */
    float kernel_ms;      /* HIP-event time of the walk of the last query (the
                             kernel(s) that search; not the pass that writes the
                             PML lengths out, nor the MS length extension)      */
} spx_walk_stats;

const char *spx_last_error(void);
/* number of usable gfx950 devices (0 if none); never fails                   */
int spx_device_count(void);

/* ---- index construction -------------------------------------------------
 * Replaces pml_t::pml_t / ms_t::ms_t (compute_ms_pml.cpp:700-721, 755-786).
 *
 * spx_index_from_runs: raw per-run arrays as `newscanNT.x` + `pfp_thresholds
 * -r` write them and as pml_pointers / ms_pointers
 * consume them (compute_ms_pml.cpp:44-82, 357-402):
 *   heads  r bytes (0 and 1 both mean terminator, ms_rle_string.hpp:250)
 *   lens   r run lengths (> 0)
 *   thr    r raw thresholds (.thr_pos values; zeros are skipped exactly like
 *          thr_bv's constructor does, thresholds_ds.hpp:421-423)
 *   ssa/esa  r stored SA samples (val = right ? right-1 : n-1,
 *          compute_ms_pml.cpp:433) or NULL (PML-only index)
 *   doc_start/doc_end  r document ids (DocumentArray, doc_array.hpp:22-23) or NULL
 * `where` says where the arrays live: 0 = host memory, 1 = memory of `device`
 * (e.g. produced by another GPU library; no host round trip).  The arrays are
 * not retained.  The flat HBM layout is built ON THE DEVICE.                  */
spx_index *spx_index_from_runs(const uint8_t *heads, const uint64_t *lens, const uint64_t *thr,
                               uint64_t r, const uint64_t *ssa, const uint64_t *esa,
                               const uint64_t *doc_start, const uint64_t *doc_end, int where,
                               int device);
/* Same, from the raw files <prefix>.bwt.heads/.bwt.len/.thr_pos (+ .ssa/.esa
 * when mode == SPX_MODE_MS).  5-byte little-endian records (common.hpp:59-60). */
spx_index *spx_index_load_raw(const char *prefix, int mode, int device);
void spx_index_free(spx_index *ix);
/* get_bwt_stats() (compute_ms_pml.cpp:171-173, 739-741): n = bwt size, r = runs -- the runs of the
 * BWT as the files hold them.  (The flat layout may keep a run as several rows: one of 2^16 positions
 * or more, or one whose LF image covers many runs, is laid out as consecutive pieces of the same head;
 * spx_index_describe reports those rows as "flat_runs".  No result depends on it.)                    */
int spx_index_stats(const spx_index *ix, uint64_t *n, uint64_t *r);
/* bytes of HBM the flat layout occupies                                       */
int spx_index_device_bytes(const spx_index *ix, uint64_t *bytes);
/* Text for the MS length extension (ms_t's `ra.charAt`, compute_ms_pml.cpp:805):
 * the plain text replaces the SLP random-access structure (ms_t loads the SLP of
 * the SAME text the BWT was built from, :769-774).  where: 0 host, 1 device.
 * The text is checked against the index before it is accepted: it must have
 * n - 1 characters (the BWT has one position per character plus the
 * terminator), and -- index with SA samples -- text[samples_start[k]] must be the
 * head of run k for every run (the BWT character is the text character in front
 * of the suffix).  A text that fails is refused with SPX_E_FORMAT: wrong MS
 * lengths would otherwise come out silently.  where | SPX_TEXT_UNCHECKED skips
 * the check (synthetic indexes that are not the BWT of any text).              */
#define SPX_TEXT_UNCHECKED 2
int spx_index_set_text(spx_index *ix, const uint8_t *text, uint64_t n_text, int where);
/* The same text from the MS index itself, when no copy of it is at hand (the reference's own
 * index files hold everything needed: run heads, run lengths, samples_start): the BWT character
 * at position p is text[SA[p] - 1], samples_start[k] names that text position for the first
 * position of run k, and every LF step moves one position to the left -- one lane per run walks
 * LF until it reaches the first position of a run.  n steps in all, a fraction of a second per
 * 10^9 characters.  Needs an index built with SA samples.                                   */
int spx_index_rebuild_text(spx_index *ix);
/* The text the index holds (set or rebuilt), e.g. to keep it as a file next to the index.  out NULL:
 * only *n_text is filled.  where: 0 host, 1 device.                                              */
int spx_index_copy_text(spx_index *ix, uint8_t *out, uint64_t capacity, int where, uint64_t *n_text);

/* ---- the vectors as text --------------------------------------------------
 * The reference writes every vector as text: under a ">id" line one line of "<value> " per character
 * (compute_ms_pml.cpp:1001-1010, 1182-1205; std::ostream_iterator<size_t>(file, " ")).  At GPU speed the
 * digits are the job, and the values are in HBM: spx_query_text_begin runs [digestion (digest_kind 0: none) +]
 * the query and formats the requested streams (SPX_TEXT_*: .pseudo_lengths / .lengths, .pointers,
 * .doc_numbers) on the device; read q's record is  gap[q] free bytes  +  its values line  +  '\n'  (gap NULL:
 * none) -- the caller drops ">id\n" into the gap, ids never travel.  out_bytes[i] = size of stream i (0: not
 * asked for).  spx_query_text_fetch then copies the streams (and, when line_start[i] is given, the nreads + 1
 * record offsets) into the caller's buffers (page-locked ones copy at DMA speed) -- the next call on the
 * same index after a successful begin, from the same thread.  out_class (when given) must stay valid until
 * spx_query_text_fetch returns: the class records travel with the text.                                  */
#define SPX_TEXT_LENGTHS 1u
#define SPX_TEXT_POINTERS 2u
#define SPX_TEXT_DOCS 4u
int spx_query_text_begin(spx_index *ix, int mode, int digest_kind, uint32_t k, uint32_t w, const uint8_t *seqs,
                         const uint64_t *offsets, uint64_t nreads, const uint32_t *gap, uint32_t streams,
                         spx_class *out_class, uint64_t bin_width, uint64_t max_value_thr, uint64_t out_bytes[3]);
int spx_query_text_fetch(spx_index *ix, char *text[3], uint64_t *line_start[3]);
/* Optional, before a run of such calls: allocates the device scratch a begin / fetch pair of up to max_chars characters in
 * max_reads reads will need (text_bytes[i]: the expected size of stream i, or NULL), so that the first batch does not pay for
 * it (~10 ms) and no buffer has to grow -- free + allocate, a device-wide synchronisation -- while other query contexts of
 * the same device are at work.  A hint, not a limit.  (The reference has no counterpart: its vectors are std::vectors per
 * read, compute_ms_pml.cpp:238-245.)                                                                                      */
int spx_query_text_reserve(spx_index *ix, int mode, int digest_kind, uint32_t k, uint64_t max_chars, uint64_t max_reads,
                           uint32_t streams, int with_class, const uint64_t text_bytes[3]);

/* ---- flat-layout cache and replication -------------------------------------
 * pml_t / ms_t deserialise their index on every run (compute_ms_pml.cpp:700-721,
 * 755-786, the timed "loading the index" step).  Here the raw run files are
 * flattened on the device (seconds at 10^9 runs); spx_index_save writes the
 * flat arrays as they are (<path>, by convention <ref>.<mode>.spx), and
 * spx_index_load_flat brings them back with nothing but file reads and
 * host-to-device copies.  A cache written by another layout version
 * (spx_version()) is refused.  spx_index_clone copies an index to another
 * device (peer-to-peer over xGMI where the devices reach each other): flatten
 * once, replicate N-1 times ("index replicated in each GPU's HBM"). */
const char *spx_version(void);
int spx_index_save(spx_index *ix, const char *path);
spx_index *spx_index_load_flat(const char *path, int device);
spx_index *spx_index_clone(spx_index *src, int device);
/* What the index was built from, as the caller names it (at most 127 characters; the host harness: names,
 * sizes and modification times of <ref>.bwt.heads / .bwt.len / .thr_pos / ...).  Saved with the cache and
 * handed back by an index loaded from it: pml_t / ms_t always deserialise the CURRENT files
 * (compute_ms_pml.cpp:700-721), so a cache whose tag differs from the files' is stale and must not be used. */
int spx_index_set_source_tag(spx_index *ix, const char *tag);
const char *spx_index_source_tag(const spx_index *ix);
/* one-line JSON description of the layout (sizes, fat-table density, version) */
int spx_index_describe(const spx_index *ix, char *buf, size_t cap);

/* ---- queries -------------------------------------------------------------
 * Batch form of matching_statistics.  seqs = concatenated reads, already
 * upper-cased / digested by the caller (compute_ms_pml.cpp:916-923), offsets =
 * nreads+1 offsets into seqs.  All outputs are laid out at the same offsets
 * (element i of read q at offsets[q]+i), and may be NULL when not wanted:
 *   out_lengths   PML lengths (PML mode) or MS lengths (MS mode, needs
 *                 spx_index_set_text; compute_ms_pml.cpp:800-810).  PML mode
 *                 with out_lengths NULL and out_class given classifies without
 *                 writing the per-character values (the report's columns only)
 *   out_pointers  MS pointers (MS mode only)
 *   out_docs      document ids (index must have been built with doc arrays)
 *   out_class     nreads entries, bin-max classifier over out_lengths' values
 *                 (bin_width in [1, ..]; ignored when out_class is NULL)
 * Host-buffer form: copies in, runs, copies out, returns when done (large
 * batches as a pipeline of chunks, so that the copies overlap the kernels; the
 * kernel time spx_last_walk_stats reports then spans that pipeline).           */
int spx_query_batch(spx_index *ix, int mode, const uint8_t *seqs, const uint64_t *offsets,
                    uint64_t nreads, uint32_t *out_lengths, uint64_t *out_pointers,
                    uint32_t *out_docs, spx_class *out_class, uint64_t bin_width,
                    uint64_t max_value_thr);
/* Device-buffer form: every pointer is memory of the index's device; the work
 * is enqueued on `stream` (a hipStream_t passed as void*, NULL = default
 * stream) and the call returns without synchronising.  d_seqs must be 16-byte
 * aligned and readable for round_up(total_chars, 4) + 32 bytes (the walk reads
 * characters in aligned 32-byte windows); output buffers must be 16-byte
 * aligned (results are written as 16-byte vectors).  total_chars is
 * d_offsets[nreads] - d_offsets[0] or an upper bound of it (reads digested on
 * the device: the bound of spx_digest_capacity does); it sizes internal
 * scratch (the chunked walk of long-read batches, the state-machine walk's
 * length bits), and a batch that holds more characters than it says is then
 * reported through spx_last_walk_stats (SPX_E_FORMAT), results undefined.  (The
 * plain walk over compact rows needs no such scratch and is right regardless.) */
int spx_query_batch_device(spx_index *ix, int mode, const uint8_t *d_seqs,
                           const uint64_t *d_offsets, uint64_t nreads, uint64_t total_chars,
                           uint32_t *d_out_lengths, uint64_t *d_out_pointers,
                           uint32_t *d_out_docs, spx_class *d_out_class, uint64_t bin_width,
                           uint64_t max_value_thr, void *stream);
/* 16-bit outputs: the same two calls with out_lengths / out_docs as uint16_t arrays -- half
 * the output bytes (the PCIe copy of the host form, the stores of the device form).  Every
 * read must be shorter than 65536 characters (lengths and document ids then fit, cf. the
 * widths row of the boundary table): the host form checks, the device form reports a longer
 * read through spx_last_walk_stats (SPX_E_FORMAT).                              */
int spx_query_batch16(spx_index *ix, int mode, const uint8_t *seqs, const uint64_t *offsets,
                      uint64_t nreads, uint16_t *out_lengths, uint64_t *out_pointers,
                      uint16_t *out_docs, spx_class *out_class, uint64_t bin_width,
                      uint64_t max_value_thr);
int spx_query_batch_device16(spx_index *ix, int mode, const uint8_t *d_seqs,
                             const uint64_t *d_offsets, uint64_t nreads, uint64_t total_chars,
                             uint16_t *d_out_lengths, uint64_t *d_out_pointers,
                             uint16_t *d_out_docs, spx_class *d_out_class, uint64_t bin_width,
                             uint64_t max_value_thr, void *stream);
/* Statistics + HIP-event kernel time of the most recent query on `ix`
 * (synchronises with that query).                                            */
int spx_last_walk_stats(spx_index *ix, spx_walk_stats *out);

/* Long-read batches (fewer reads than the GPU has lanes) are cut into chunks that are walked
 * concurrently and joined exactly; results are the plain walk's, bit for bit.
 * out[0] = chunk size of the last query in characters (0: it ran the plain walk), out[1] = upper
 * bound of its chunks, out[2] = characters walked a second time to join chunks, out[3] = reads
 * whose chunks did not join and that were walked again the plain way.
 * Knobs (spx_set_option): "chunk_mode" 0 automatic / 1 never / 2 always, "chunk_shift" log2 of the
 * chunk size (0 automatic).                                                                      */
int spx_last_chunk_stats(spx_index *ix, uint64_t out[4]);

/* Page-locked host memory for spx_query_batch's buffers: with buffers from
 * spx_host_alloc the copies run at PCIe DMA speed instead of through a pageable
 * staging copy (any host memory is accepted; this is only faster).  NULL on failure. */
void *spx_host_alloc(size_t bytes);
void spx_host_free(void *p);

/* Page-locks memory the CALLER owns, for the same purpose.  What it is for: the harness maps the tail of an output
 * file (the ofstream of compute_ms_pml.cpp:1001-1010 in the reference) and registers the mapping, so that
 * spx_query_text_fetch lands the text in the file's page-cache pages directly -- no staging buffer, no write() --
 * at the link's rate (profiles/r05_drain_hip.txt: 57 GB/s into a tmpfs file against 6.5 GB/s for pwrite).  Works for
 * anonymous memory and for shared mappings of tmpfs files; a file system whose pages cannot be pinned makes it fail
 * with SPX_E_HIP, and the caller copies instead.  spx_host_unregister before the memory is unmapped.          */
int spx_host_register(void *p, size_t bytes);
int spx_host_unregister(void *p);

/* ---- minimizer digestion (run -m / -a) -------------------------------------
 * Replaces perform_minimizer_digestion / perform_dna_minimizer_digestion
 * (src/spumoni.cpp:294-319, 321-342), which the harness applies to every read
 * before matching_statistics (src/compute_ms_pml.cpp:919-923).  Reads must be
 * upper-cased already (:916-917).  k in [1,4], w >= k (spumoni_main.hpp:316-317).
 * The minimizer streams restate dnbaker/bonsai @5273b81a92 (source absent
 * offline, parity unpinned -- DESIGN.md 4.4); the four 8-bit character hashes
 * the -m variant depends on can be pinned with
 * spx_set_option(ix, "minimizer_charhash", A | C<<8 | G<<16 | T<<24).          */
#define SPX_DIGEST_PROMOTED 1 /* -m: promoted-alphabet minimizers, one byte each   */
#define SPX_DIGEST_DNA 2      /* -a: minimizer k-mers spelled in DNA letters       */
/* bytes d_out_seqs must hold for any input of total_chars characters (worst
 * case + the read-ahead padding spx_query_batch_device wants of its d_seqs)    */
uint64_t spx_digest_capacity(int kind, uint32_t k, uint64_t total_chars);
/* Device form: digested reads, concatenated, into d_out_seqs (16-byte aligned if
 * it is to be queried), their nreads+1 offsets into d_out_offsets; enqueued on
 * `stream`, returns without synchronising.  d_seqs must be 16-byte aligned and
 * readable for round_up(total_chars, 16) + 16 bytes (the kernels stage the reads
 * with aligned 16-byte loads): SPX_E_ARG otherwise.  d_out_seqs / d_out_offsets can be
 * handed to spx_query_batch_device as they are (total = d_out_offsets[nreads]). */
int spx_digest_batch_device(spx_index *ix, int kind, uint32_t k, uint32_t w, const uint8_t *d_seqs,
                            const uint64_t *d_offsets, uint64_t nreads, uint64_t total_chars,
                            uint8_t *d_out_seqs, uint64_t out_capacity, uint64_t *d_out_offsets,
                            void *stream);
/* Host form.  out_seqs may be NULL with out_capacity 0 to learn the sizes only
 * (out_offsets is filled, SPX_E_ARG is returned when anything was digested).   */
int spx_digest_batch(spx_index *ix, int kind, uint32_t k, uint32_t w, const uint8_t *seqs,
                     const uint64_t *offsets, uint64_t nreads, uint8_t *out_seqs,
                     uint64_t out_capacity, uint64_t *out_offsets);
/* Host form of digest + query in one call: what the harness loop body does for
 * one read (compute_ms_pml.cpp:916-938), for a batch.  The digested reads stay
 * on the device; out_offsets (nreads+1) receives their offsets, and every
 * output is laid out at THOSE offsets; out_capacity = entries each output
 * buffer holds (>= out_offsets[nreads], e.g. spx_digest_capacity()).  A read
 * that digests to nothing (offsets equal) is the caller's fatal case (:926-931). */
int spx_digest_query_batch(spx_index *ix, int mode, int kind, uint32_t k, uint32_t w,
                           const uint8_t *seqs, const uint64_t *offsets, uint64_t nreads,
                           uint64_t *out_offsets, uint64_t out_capacity, uint32_t *out_lengths,
                           uint64_t *out_pointers, uint32_t *out_docs, spx_class *out_class,
                           uint64_t bin_width, uint64_t max_value_thr);
/* The same with everything resident in device memory, asynchronous on `stream`:
 * the batch form of "digest the read, then matching_statistics" (compute_ms_pml.cpp:919-938, the body of the harness
 * loop under `run -m` / `run -a`) for reads that are already in HBM.  d_seqs / d_offsets as for
 * spx_digest_batch_device; d_digested (digested_capacity >= spx_digest_capacity() bytes) is working memory that holds
 * the digested reads afterwards -- concatenated at d_out_offsets, or, when the library chose to skip that pass, still
 * where the digestion parked them (read q's minimizers at d_digested[d_offsets[q] ..]): only d_out_offsets and the
 * outputs are the contract.  d_out_offsets (nreads + 1) receives the digested reads' offsets and every output is laid
 * out at THOSE offsets; size the outputs for total_chars entries (a read digests to at most its length).  Outputs
 * as for spx_query_batch_device / spx_query_batch_device16.                                                       */
int spx_digest_query_batch_device(spx_index *ix, int mode, int kind, uint32_t k, uint32_t w,
                                  const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t nreads,
                                  uint64_t total_chars, uint8_t *d_digested, uint64_t digested_capacity,
                                  uint64_t *d_out_offsets, uint32_t *d_out_lengths, uint64_t *d_out_pointers,
                                  uint32_t *d_out_docs, spx_class *d_out_class, uint64_t bin_width,
                                  uint64_t max_value_thr, void *stream);
int spx_digest_query_batch_device16(spx_index *ix, int mode, int kind, uint32_t k, uint32_t w,
                                    const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t nreads,
                                    uint64_t total_chars, uint8_t *d_digested, uint64_t digested_capacity,
                                    uint64_t *d_out_offsets, uint16_t *d_out_lengths, uint64_t *d_out_pointers,
                                    uint16_t *d_out_docs, spx_class *d_out_class, uint64_t bin_width,
                                    uint64_t max_value_thr, void *stream);

/* ---- tuning knobs (optional) --------------------------------------------- */
/* keys: "waves_per_cu" (occupancy target; default 16 for the plain PML walk, 12 with document ids / MS pointers, 20 for the
 * state-machine walk), "digest_parked" (digest + query: 0 automatic, 1 always concatenate the digested reads, 2 leave them
 * parked whenever the walk can take them), "lanes_per_wave" (reads per
 * wavefront, 0 = automatic; 1 = the one-wavefront-per-read mapping of SURVEY 7.1,
 * kept as a measurable experiment -- see DESIGN.md 4.1)                        */
/* "minimizer_charhash": see the digestion section above                        */
/* "blocking_sync" 1: spx_query_text_begin / _fetch sleep on an event while they wait for the device instead of spinning on
 * their stream -- for processes that keep several query contexts busy from as many threads and need the cores (0: spin)     */
int spx_set_option(spx_index *ix, const char *key, int64_t value);

#ifdef __cplusplus
}
#endif
#endif
