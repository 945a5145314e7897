/*
 * spumoni_oracle.c -- CPU ORACLE (test infrastructure only; see the header).
 *
 * PARITY UNPINNED against an upstream binary (none can be built offline);
 * pinned by brute-force KATs in tests/test_oracle_kat.py.
 *
 * Every query function below restates, statement for statement, the control
 * flow of the reference's pml_pointers::_query / ms_pointers::_query
 * (/root/reference/src/compute_ms_pml.cpp).  The rank/select primitives are
 * written the obvious way (binary searches over plain arrays) from the
 * published semantics of ri::rle_string -- deliberately NOT sharing any code
 * or layout with the HIP path in spumoni_amd/csrc.
 *
 * Build: see oracle/Makefile  (gcc -O2 -fsigned-char -fopenmp -shared).
 */
#include "spumoni_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void *xcalloc(size_t n, size_t sz) {
    void *p = calloc(n ? n : 1, sz);
    if (!p) {
        fprintf(stderr, "oracle: out of memory\n");
        abort();
    }
    return p;
}

/* --------------------------------------------------------------------------
 * Construction.  Mirrors ms_rle_string's RLE constructor
 * (include/ms_rle_string.hpp:217-288), build_F_ (compute_ms_pml.cpp:119-147)
 * and thr_bv's constructor (include/thresholds_ds.hpp:384-440).
 * -------------------------------------------------------------------------- */
orc_index *orc_build(const uint8_t *heads, const uint64_t *lens, const uint64_t *thr, uint64_t r,
                     const uint64_t *samples_start, const uint64_t *samples_last,
                     const uint64_t *start_runs_doc, const uint64_t *end_runs_doc) {
    orc_index *ix = (orc_index *)xcalloc(1, sizeof(orc_index));
    ix->r = r;
    ix->S = (uint64_t *)xcalloc(r + 1, sizeof(uint64_t));
    ix->H = (uint8_t *)xcalloc(r + 1, 1);

    /* pass 1: heads (0 -> TERMINATOR, ms_rle_string.hpp:249-253), n, per-letter counts */
    uint64_t n = 0;
    for (uint64_t i = 0; i < r; ++i) {
        uint8_t c = heads[i];
        if (c <= ORC_TERMINATOR) {
            c = ORC_TERMINATOR;
            ix->terminator_position = i; /* compute_ms_pml.cpp:137 (last one wins) */
        }
        ix->H[i] = c;
        ix->S[i] = n;
        n += lens[i];
        ix->n_c[c] += lens[i];
        ix->r_c[c] += 1;
    }
    ix->S[r] = n;
    ix->n = n;

    /* F (compute_ms_pml.cpp:141-145): F[c] = number of characters smaller than c */
    {
        uint64_t acc = 0;
        for (int c = 0; c < 256; ++c) {
            ix->F[c] = acc;
            acc += ix->n_c[c];
        }
    }

    /* per-letter run lists Q_c, prefix counts P_c (runs_per_letter[c] marks the
     * last position of each c-run in the c-subsequence, ms_rle_string.hpp:259-260) */
    uint64_t fill[256];
    for (int c = 0; c < 256; ++c) {
        ix->Q[c] = (uint64_t *)xcalloc(ix->r_c[c], sizeof(uint64_t));
        ix->P[c] = (uint64_t *)xcalloc(ix->r_c[c] + 1, sizeof(uint64_t));
        ix->T[c] = (uint64_t *)xcalloc(ix->r_c[c], sizeof(uint64_t));
        fill[c] = 0;
    }
    for (uint64_t i = 0; i < r; ++i) {
        uint8_t c = ix->H[i];
        uint64_t j = fill[c]++;
        ix->Q[c][j] = i;
        ix->P[c][j + 1] = ix->P[c][j] + lens[i];
        /* thresholds_ds.hpp:421-423: only non-zero thresholds are stored */
        if (thr && thr[i] > 0) ix->T[c][ix->t_c[c]++] = thr[i];
    }

    if (samples_start) {
        ix->samples_start = (uint64_t *)xcalloc(r, sizeof(uint64_t));
        memcpy(ix->samples_start, samples_start, r * sizeof(uint64_t));
    }
    if (samples_last) {
        ix->samples_last = (uint64_t *)xcalloc(r, sizeof(uint64_t));
        memcpy(ix->samples_last, samples_last, r * sizeof(uint64_t));
    }
    if (start_runs_doc) {
        ix->start_runs_doc = (uint64_t *)xcalloc(r, sizeof(uint64_t));
        memcpy(ix->start_runs_doc, start_runs_doc, r * sizeof(uint64_t));
    }
    if (end_runs_doc) {
        ix->end_runs_doc = (uint64_t *)xcalloc(r, sizeof(uint64_t));
        memcpy(ix->end_runs_doc, end_runs_doc, r * sizeof(uint64_t));
    }
    return ix;
}

static uint64_t *read_5byte_file(const char *path, uint64_t *count, int stride_vals, int pick) {
    /* file of 5-byte little-endian values; keep value `pick` of every `stride_vals` */
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint64_t nvals = (uint64_t)sz / 5 / (uint64_t)stride_vals;
    uint64_t *out = (uint64_t *)xcalloc(nvals, sizeof(uint64_t));
    for (uint64_t i = 0; i < nvals; ++i) {
        for (int s = 0; s < stride_vals; ++s) {
            uint64_t v = 0;
            if (fread(&v, 5, 1, f) != 1) {
                fclose(f);
                free(out);
                return NULL;
            }
            if (s == pick) out[i] = v;
        }
    }
    fclose(f);
    *count = nvals;
    return out;
}

orc_index *orc_load_raw(const char *prefix, int want_samples) {
    char path[4096];
    snprintf(path, sizeof path, "%s.bwt.heads", prefix);
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    uint64_t r = (uint64_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *heads = (uint8_t *)xcalloc(r, 1);
    if (fread(heads, 1, r, f) != r) {
        fclose(f);
        return NULL;
    }
    fclose(f);
    uint64_t cnt = 0;
    snprintf(path, sizeof path, "%s.bwt.len", prefix);
    uint64_t *lens = read_5byte_file(path, &cnt, 1, 0);
    if (!lens || cnt != r) return NULL;
    snprintf(path, sizeof path, "%s.thr_pos", prefix);
    uint64_t *thr = read_5byte_file(path, &cnt, 1, 0);
    if (!thr || cnt != r) return NULL;
    uint64_t *ssa = NULL, *esa = NULL;
    if (want_samples) {
        uint64_t n = 0;
        for (uint64_t i = 0; i < r; ++i) n += lens[i];
        snprintf(path, sizeof path, "%s.ssa", prefix);
        ssa = read_5byte_file(path, &cnt, 2, 1); /* (left,right) pairs: keep right */
        if (!ssa || cnt != r) return NULL;
        snprintf(path, sizeof path, "%s.esa", prefix);
        esa = read_5byte_file(path, &cnt, 2, 1);
        if (!esa || cnt != r) return NULL;
        for (uint64_t i = 0; i < r; ++i) { /* compute_ms_pml.cpp:433 */
            ssa[i] = ssa[i] ? ssa[i] - 1 : n - 1;
            esa[i] = esa[i] ? esa[i] - 1 : n - 1;
        }
    }
    orc_index *ix = orc_build(heads, lens, thr, r, ssa, esa, NULL, NULL);
    free(heads);
    free(lens);
    free(thr);
    free(ssa);
    free(esa);
    return ix;
}

void orc_free(orc_index *ix) {
    if (!ix) return;
    free(ix->S);
    free(ix->H);
    for (int c = 0; c < 256; ++c) {
        free(ix->Q[c]);
        free(ix->P[c]);
        free(ix->T[c]);
    }
    free(ix->samples_start);
    free(ix->samples_last);
    free(ix->start_runs_doc);
    free(ix->end_runs_doc);
    free(ix);
}

/* --------------------------------------------------------------------------
 * ri::rle_string primitives.
 * -------------------------------------------------------------------------- */

/* index of the run containing position p (p < n) */
uint64_t orc_run_of_position(const orc_index *ix, uint64_t p) {
    uint64_t lo = 0, hi = ix->r; /* invariant: S[lo] <= p < S[hi] */
    while (hi - lo > 1) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (ix->S[mid] <= p)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

uint8_t orc_at(const orc_index *ix, uint64_t p) { return ix->H[orc_run_of_position(ix, p)]; }

/* number of c-runs with run index < k  (= run_heads.rank(k, c)) */
uint64_t orc_run_head_rank(const orc_index *ix, uint64_t k, uint8_t c) {
    const uint64_t *Q = ix->Q[c];
    uint64_t lo = 0, hi = ix->r_c[c];
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (Q[mid] < k)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

/* number of c in bwt[0, p)   (0 if the letter is absent; n_c if p == n) */
uint64_t orc_rank(const orc_index *ix, uint64_t p, uint8_t c) {
    if (ix->n_c[c] == 0) return 0;
    if (p == ix->n) return ix->n_c[c];
    uint64_t k = orc_run_of_position(ix, p);
    uint64_t rk = orc_run_head_rank(ix, k, c);
    uint64_t tail = (ix->H[k] == c) ? (p - ix->S[k]) : 0;
    return ix->P[c][rk] + tail;
}

/* position of the i-th c (0-based), i < n_c */
uint64_t orc_select(const orc_index *ix, uint64_t i, uint8_t c) {
    const uint64_t *P = ix->P[c];
    uint64_t lo = 0, hi = ix->r_c[c]; /* invariant: P[lo] <= i < P[hi] */
    if (i >= ix->n_c[c]) {
        fprintf(stderr, "oracle: select(%llu,%u) out of range (n_c=%llu)\n",
                (unsigned long long)i, (unsigned)c, (unsigned long long)ix->n_c[c]);
        abort();
    }
    while (hi - lo > 1) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (P[mid] <= i)
            lo = mid;
        else
            hi = mid;
    }
    return ix->S[ix->Q[c][lo]] + (i - P[lo]);
}

/* thr_bv::operator[]  (include/thresholds_ds.hpp:478-491) */
uint64_t orc_threshold(const orc_index *ix, uint64_t k) {
    uint8_t c = ix->H[k];                          /* head_of(i)           */
    uint64_t rank = orc_run_head_rank(ix, k, c);   /* run_head_rank(i, c)  */
    if (rank == 0) return 0;
    if (rank - 1 >= ix->t_c[c]) {
        /* sdsl select past the last 1: undefined upstream; make it loud here */
        fprintf(stderr, "oracle: threshold select(%llu) past %llu stored for letter %u\n",
                (unsigned long long)(rank - 1), (unsigned long long)ix->t_c[c], (unsigned)c);
        abort();
    }
    return ix->T[c][rank - 1];                     /* thresholds_per_letter[c].select(rank-1) */
}

/* compute_ms_pml.cpp:180-187 */
uint64_t orc_LF(const orc_index *ix, uint64_t p, uint8_t c) {
    uint64_t c_before = orc_rank(ix, p, c);
    return ix->F[c] + c_before;
}

/* ri::r_index::get_last_run_sample(): (samples_last[r-1] + 1) % bwt.size() */
uint64_t orc_last_run_sample(const orc_index *ix) {
    return (ix->samples_last[ix->r - 1] + 1) % ix->n;
}

/* --------------------------------------------------------------------------
 * The four _query variants.  `char` is signed here (-fsigned-char, as on the
 * reference's x86-64 build): bwt[pos] (unsigned char) == c (char) compares
 * after integer promotion, so bytes >= 128 never match.
 * -------------------------------------------------------------------------- */

#define QFN(name) orc_##name
#define QIDX orc_index
#define Q_SIZE(ix) ((ix)->n)
#define Q_NUMBER_OF_RUNS(ix) ((ix)->r)
#define Q_NUMBER_OF_LETTER(ix, c) ((ix)->n_c[(c)])
#define Q_AT(ix, p) orc_at((ix), (p))
#define Q_RANK(ix, p, c) orc_rank((ix), (p), (c))
#define Q_SELECT(ix, i, c) orc_select((ix), (i), (c))
#define Q_RUN_OF_POSITION(ix, p) orc_run_of_position((ix), (p))
#define Q_THRESHOLD(ix, k) orc_threshold((ix), (k))
#define Q_LF(ix, p, c) orc_LF((ix), (p), (c))
#define Q_LAST_RUN_SAMPLE(ix) orc_last_run_sample((ix))
#define Q_START_DOC(ix, k) ((ix)->start_runs_doc[(k)])
#define Q_END_DOC(ix, k) ((ix)->end_runs_doc[(k)])
#define Q_SAMPLE_START(ix, k) ((ix)->samples_start[(k)])
#define Q_SAMPLE_LAST(ix, k) ((ix)->samples_last[(k)])
#include "orc_queries.inc"

/* ms_t::matching_statistics, second loop (compute_ms_pml.cpp:800-810) */
void orc_ms_lengths(const char *read, size_t m, const uint64_t *pointers, const uint8_t *text,
                    uint64_t n_text, uint64_t *lengths) {
    size_t l = 0;
    for (size_t i = 0; i < m; ++i) {
        size_t pos = pointers[i];
        /* `read[i+l] == ra.charAt(pos+l)`: char vs the SLP's char type (signed char on
         * the reference build) -- both sides promoted the same way, so equal bytes match */
        while ((i + l) < m && (pos + l) < n_text && (i < 1 || pos != (pointers[i - 1] + 1)) &&
               read[i + l] == (char)text[pos + l]) {
            ++l;
        }
        lengths[i] = l;
        l = (l == 0 ? 0 : (l - 1));
    }
}

/* --------------------------------------------------------------------------
 * Classification (compute_ms_pml.cpp:969-995) and threshold derivation.
 * -------------------------------------------------------------------------- */
int orc_classify(const uint64_t *lengths, size_t m, size_t bin_width, size_t max_value_thr,
                 uint64_t *bins_above_out, uint64_t *bins_below_out, uint64_t *sum_out) {
    size_t sum_max_bin_values = 0;
    size_t bins_above = 0, bins_below = 0;
    size_t start_pos = 0, end_pos = 0;
    size_t nbins = 0;

    while (start_pos < m) {
        end_pos = (start_pos + bin_width < m) ? start_pos + bin_width : m; /* :977 */
        if (m - end_pos < bin_width) end_pos = m;                          /* :980 */
        uint64_t max_val = lengths[start_pos];
        for (size_t x = start_pos; x < end_pos; ++x)
            if (lengths[x] > max_val) max_val = lengths[x];
        if (max_val >= max_value_thr)
            bins_above++;
        else
            bins_below++;
        sum_max_bin_values += max_val;
        nbins++;
        start_pos += (end_pos - start_pos);
    }
    if (bins_above_out) *bins_above_out = bins_above;
    if (bins_below_out) *bins_below_out = bins_below;
    if (sum_out) *sum_out = sum_max_bin_values;
    (void)nbins;
    return (bins_above / (bins_above + bins_below + 0.0) > 0.50) ? 1 : 0; /* :993 */
}

size_t orc_max_value_thr(double percentile_value, int is_pml, int use_promotions,
                         int use_dna_letters) {
    double mx = percentile_value > 3.0 ? percentile_value : 3.0; /* std::max(p, 3.0) */
    size_t max_value_thr = (size_t)mx;                          /* :871 / :1061     */
    if (use_dna_letters)
        max_value_thr++;
    else if (is_pml && !use_dna_letters && !use_promotions)
        max_value_thr += 4; /* :874-875, PML only */
    return max_value_thr;
}

/* --------------------------------------------------------------------------
 * Batch helpers (tests + cpu_baseline).
 * -------------------------------------------------------------------------- */
void orc_pml_batch(const orc_index *ix, const uint8_t *seqs, const uint64_t *offs,
                   uint64_t nreads, uint32_t *out_lengths, uint32_t *out_docs, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        uint64_t cap = 0;
        uint64_t *lens = NULL, *docs = NULL;
#pragma omp for schedule(dynamic, 64)
        for (uint64_t q = 0; q < nreads; ++q) {
            uint64_t m = offs[q + 1] - offs[q];
            if (m > cap) {
                cap = m * 2;
                lens = (uint64_t *)realloc(lens, cap * sizeof(uint64_t));
                docs = (uint64_t *)realloc(docs, cap * sizeof(uint64_t));
            }
            const char *pat = (const char *)(seqs + offs[q]);
            if (out_docs)
                orc_pml_query_doc(ix, pat, m, lens, docs);
            else
                orc_pml_query(ix, pat, m, lens);
            for (uint64_t x = 0; x < m; ++x) {
                out_lengths[offs[q] + x] = (uint32_t)lens[x];
                if (out_docs) out_docs[offs[q] + x] = (uint32_t)docs[x];
            }
        }
        free(lens);
        free(docs);
    }
}

void orc_ms_batch(const orc_index *ix, const uint8_t *seqs, const uint64_t *offs, uint64_t nreads,
                  uint64_t *out_pointers, uint32_t *out_docs, const uint8_t *text,
                  uint64_t n_text, uint32_t *out_ms_lengths, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        uint64_t cap = 0;
        uint64_t *docs = NULL, *lens = NULL;
#pragma omp for schedule(dynamic, 64)
        for (uint64_t q = 0; q < nreads; ++q) {
            uint64_t m = offs[q + 1] - offs[q];
            if (m > cap) {
                cap = m * 2;
                docs = (uint64_t *)realloc(docs, cap * sizeof(uint64_t));
                lens = (uint64_t *)realloc(lens, cap * sizeof(uint64_t));
            }
            const char *pat = (const char *)(seqs + offs[q]);
            uint64_t *ptrs = out_pointers + offs[q];
            if (out_docs) {
                orc_ms_query_doc(ix, pat, m, ptrs, docs);
                for (uint64_t x = 0; x < m; ++x) out_docs[offs[q] + x] = (uint32_t)docs[x];
            } else {
                orc_ms_query(ix, pat, m, ptrs);
            }
            if (text && out_ms_lengths) {
                orc_ms_lengths(pat, m, ptrs, text, n_text, lens);
                for (uint64_t x = 0; x < m; ++x) out_ms_lengths[offs[q] + x] = (uint32_t)lens[x];
            }
        }
        free(docs);
        free(lens);
    }
}

void orc_classify_batch(const uint32_t *lengths, const uint64_t *offs, uint64_t nreads,
                        uint64_t bin_width, uint64_t max_value_thr, uint8_t *found,
                        uint32_t *above, uint32_t *below, uint64_t *sum_max) {
    uint64_t cap = 0;
    uint64_t *tmp = NULL;
    for (uint64_t q = 0; q < nreads; ++q) {
        uint64_t m = offs[q + 1] - offs[q];
        if (m > cap) {
            cap = m * 2;
            tmp = (uint64_t *)realloc(tmp, cap * sizeof(uint64_t));
        }
        for (uint64_t x = 0; x < m; ++x) tmp[x] = lengths[offs[q] + x];
        uint64_t a = 0, b = 0, s = 0;
        int f = orc_classify(tmp, m, bin_width, max_value_thr, &a, &b, &s);
        if (found) found[q] = (uint8_t)f;
        if (above) above[q] = (uint32_t)a;
        if (below) below[q] = (uint32_t)b;
        if (sum_max) sum_max[q] = s;
    }
    free(tmp);
}

/* Walk statistics (not a reference function): counts how often the walk takes
 * the mismatch branch (:251) and the predecessor branch (:270); feeds the
 * algorithmic-bytes model of SURVEY 8(d).                                     */
void orc_pml_stats(const orc_index *ix, const uint8_t *seqs, const uint64_t *offs,
                   uint64_t nreads, uint64_t *steps_out, uint64_t *jumps_out,
                   uint64_t *pred_out) {
    uint64_t steps = 0, jumps = 0, preds = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : steps, jumps, preds)
    for (uint64_t q = 0; q < nreads; ++q) {
        uint64_t m = offs[q + 1] - offs[q];
        const char *pattern = (const char *)(seqs + offs[q]);
        uint64_t pos = ix->n - 1;
        for (size_t i = 0; i < m; ++i) {
            char c = pattern[m - i - 1];
            steps++;
            if (ix->n_c[(uint8_t)c] == 0) {
            } else if (pos < ix->n && (int)orc_at(ix, pos) == (int)c) {
            } else {
                jumps++;
                uint64_t rnk = orc_rank(ix, pos, (uint8_t)c);
                size_t thr = ix->n + 1;
                uint64_t next_pos = pos;
                if (rnk < ix->n_c[(uint8_t)c]) {
                    uint64_t j = orc_select(ix, rnk, (uint8_t)c);
                    thr = orc_threshold(ix, orc_run_of_position(ix, j));
                    next_pos = j;
                }
                if (pos < thr) {
                    rnk--;
                    next_pos = orc_select(ix, rnk, (uint8_t)c);
                    preds++;
                }
                pos = next_pos;
            }
            pos = orc_LF(ix, pos, (uint8_t)c);
        }
    }
    *steps_out = steps;
    *jumps_out = jumps;
    *pred_out = preds;
}
