/*
 * spumoni_oracle.h -- CPU ORACLE for the `spumoni run` hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may build, link or call it.
 * The shipped path (spumoni_amd/csrc) never includes or links this file.
 *
 * PARITY UNPINNED: the reference (oma219/spumoni v2.0.9) ships no tests or
 * golden vectors and cannot be compiled offline (its rank/select arithmetic
 * lives in network-fetched, un-vendored r-index / sdsl-lite).  This oracle is
 * a plain-C restatement of the reference's control flow
 * (src/compute_ms_pml.cpp:238-340, 571-682, 795-828, 180-187) over the
 * published semantics of ri::rle_string (rank / select / operator[] /
 * run_of_position) and thr_bv (include/thresholds_ds.hpp:478-491).  It is
 * pinned instead by brute-force known-answer tests (tests/test_oracle_kat.py):
 * primitives against an expanded BWT string, MS lengths against brute-force
 * matching statistics.
 */
#ifndef SPUMONI_ORACLE_H
#define SPUMONI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_TERMINATOR 1 /* include/ms_rle_string.hpp:21 */

typedef struct orc_index {
    uint64_t n;           /* bwt.size(), text length incl. terminator            */
    uint64_t r;           /* bwt.number_of_runs()                                */
    uint64_t *S;          /* r+1 run start positions, S[r] = n                   */
    uint8_t *H;           /* r run heads (0 rewritten to 1, ms_rle_string.hpp:250) */
    uint64_t F[256];      /* #chars smaller than c (compute_ms_pml.cpp:119-147)  */
    uint64_t n_c[256];    /* number_of_letter(c)  (ms_rle_string.hpp:99)         */
    uint64_t r_c[256];    /* number_of_runs_of_letter(c)                         */
    uint64_t *Q[256];     /* run indices with head c, ascending                  */
    uint64_t *P[256];     /* r_c+1 prefix counts of c at each c-run start        */
    uint64_t *T[256];     /* stored (non-zero) thresholds of letter c, in run order
                             (thresholds_ds.hpp:421-430)                         */
    uint64_t t_c[256];    /* how many are stored                                 */
    uint64_t *samples_start; /* r entries or NULL (compute_ms_pml.cpp:397,433)   */
    uint64_t *samples_last;  /* r entries or NULL                                */
    uint64_t *start_runs_doc; /* r entries or NULL (include/doc_array.hpp:22-23) */
    uint64_t *end_runs_doc;
    uint64_t terminator_position; /* run index of the terminator (:137)          */
} orc_index;

/* Build from the raw per-run arrays that `newscanNT.x` / `pfp_thresholds -r`
 * write.  heads: r bytes; lens, thr: r values.
 * ssa/esa are the *stored* samples (val = right ? right-1 : n-1 already
 * applied, compute_ms_pml.cpp:433) or NULL; docs may be NULL.               */
orc_index *orc_build(const uint8_t *heads, const uint64_t *lens, const uint64_t *thr,
                     uint64_t r, const uint64_t *samples_start, const uint64_t *samples_last,
                     const uint64_t *start_runs_doc, const uint64_t *end_runs_doc);
/* Same, reading <prefix>.bwt.heads/.bwt.len/.thr_pos[/.ssa/.esa] (5-byte LE). */
orc_index *orc_load_raw(const char *prefix, int want_samples);
void orc_free(orc_index *ix);

/* ---- ri::rle_string primitives (upstream semantics) ---- */
uint64_t orc_run_of_position(const orc_index *ix, uint64_t p);        /* p < n   */
uint8_t orc_at(const orc_index *ix, uint64_t p);                      /* bwt[p]  */
uint64_t orc_rank(const orc_index *ix, uint64_t p, uint8_t c);        /* p <= n  */
uint64_t orc_select(const orc_index *ix, uint64_t i, uint8_t c);      /* i < n_c */
uint64_t orc_run_head_rank(const orc_index *ix, uint64_t k, uint8_t c);
uint64_t orc_threshold(const orc_index *ix, uint64_t k);  /* thr_bv::operator[] */
uint64_t orc_LF(const orc_index *ix, uint64_t p, uint8_t c);
uint64_t orc_last_run_sample(const orc_index *ix);  /* r_index::get_last_run_sample */

/* ---- the four _query variants, one read ---------------------------------- */
void orc_pml_query(const orc_index *ix, const char *pattern, size_t m, uint64_t *lengths);
void orc_pml_query_doc(const orc_index *ix, const char *pattern, size_t m, uint64_t *lengths,
                       uint64_t *doc_nums);
void orc_ms_query(const orc_index *ix, const char *pattern, size_t m, uint64_t *pointers);
void orc_ms_query_doc(const orc_index *ix, const char *pattern, size_t m, uint64_t *pointers,
                      uint64_t *doc_nums);
/* ms_t::matching_statistics second loop (compute_ms_pml.cpp:800-810): text is
 * the plain text the SLP would give random access to, n_text = ra.getLen().  */
void orc_ms_lengths(const char *read, size_t m, const uint64_t *pointers, const uint8_t *text,
                    uint64_t n_text, uint64_t *lengths);

/* bin-max classifier (compute_ms_pml.cpp:969-995).  Returns 1 = FOUND.       */
int orc_classify(const uint64_t *lengths, size_t m, size_t bin_width, size_t max_value_thr,
                 uint64_t *bins_above, uint64_t *bins_below, uint64_t *sum_max_bin_values);
/* max_value_thr derivation (compute_ms_pml.cpp:871-875 PML, :1061-1063 MS).  */
size_t orc_max_value_thr(double percentile_value, int is_pml, int use_promotions,
                         int use_dna_letters);

/* ---- batch forms used by tests / cpu_baseline (OpenMP over reads) --------- */
/* seqs: concatenated reads, offs: nreads+1 offsets.  Outputs laid out at the
 * same offsets.  out_docs / out_pointers / out_ms_lengths may be NULL.       */
void orc_pml_batch(const orc_index *ix, const uint8_t *seqs, const uint64_t *offs,
                   uint64_t nreads, uint32_t *out_lengths, uint32_t *out_docs, int nthreads);
void orc_ms_batch(const orc_index *ix, const uint8_t *seqs, const uint64_t *offs, uint64_t nreads,
                  uint64_t *out_pointers, uint32_t *out_docs, const uint8_t *text,
                  uint64_t n_text, uint32_t *out_ms_lengths, int nthreads);
/* per-read {FOUND, above, below, sum_max} from u32 lengths                   */
void orc_classify_batch(const uint32_t *lengths, const uint64_t *offs, uint64_t nreads,
                        uint64_t bin_width, uint64_t max_value_thr, uint8_t *found,
                        uint32_t *above, uint32_t *below, uint64_t *sum_max);
/* walk statistics over a batch: steps, mismatch(jump) steps, predecessor jumps */
void orc_pml_stats(const orc_index *ix, const uint8_t *seqs, const uint64_t *offs,
                   uint64_t nreads, uint64_t *steps, uint64_t *jumps, uint64_t *pred_jumps);

/* ---- tier T1: the same queries over Elias-Fano + Huffman wavelet tree + B-run block walk
 * (spumoni_oracle_t1.c); must agree with the flat tier above bit for bit ----------------- */
typedef struct orc_t1_index orc_t1_index;
orc_t1_index *orc_t1_build(const uint8_t *heads, const uint64_t *lens, const uint64_t *thr, uint64_t r,
                           const uint64_t *samples_start, const uint64_t *samples_last,
                           const uint64_t *start_runs_doc, const uint64_t *end_runs_doc);
void orc_t1_free(orc_t1_index *ix);
uint64_t orc_t1_run_of_position(const orc_t1_index *ix, uint64_t p);
uint8_t orc_t1_at(const orc_t1_index *ix, uint64_t p);
uint64_t orc_t1_rank(const orc_t1_index *ix, uint64_t p, uint8_t c);
uint64_t orc_t1_select(const orc_t1_index *ix, uint64_t i, uint8_t c);
uint64_t orc_t1_threshold(const orc_t1_index *ix, uint64_t k);
uint64_t orc_t1_LF(const orc_t1_index *ix, uint64_t p, uint8_t c);
void orc_t1_pml_batch(const orc_t1_index *ix, const uint8_t *seqs, const uint64_t *offs, uint64_t nreads,
                      uint32_t *out_lengths, uint32_t *out_docs, int nthreads);
void orc_t1_ms_batch(const orc_t1_index *ix, const uint8_t *seqs, const uint64_t *offs, uint64_t nreads,
                     uint64_t *out_pointers, uint32_t *out_docs, int nthreads);

/* ---- minimizer digestion pre-step of `run -m` / `run -a` (orc_digest.c; src/spumoni.cpp:294-342
 * over a restatement of bonsai's minimizer streams, parity unpinned) ----------------------- */
#define ORC_DIGEST_PROMOTED 1 /* -m */
#define ORC_DIGEST_DNA 2      /* -a */
void orc_digest_default_charhash(uint8_t out[4]);
size_t orc_digest(int kind, unsigned k, unsigned w, const uint8_t *charhash, const uint8_t *seq, size_t len,
                  uint8_t *out, size_t cap);
void orc_digest_batch(int kind, unsigned k, unsigned w, const uint8_t *charhash, const uint8_t *seqs,
                      const uint64_t *offs, uint64_t nreads, uint8_t *out, uint64_t cap, uint64_t *out_offs);

#ifdef __cplusplus
}
#endif
#endif
