/*
 * spumoni_oracle_t1.c -- CPU ORACLE, tier T1 ("reference-shaped"); TEST INFRASTRUCTURE ONLY.
 *
 * Same query control flow as the flat oracle (orc_queries.inc), but over the data structures
 * and operation sequences the reference really uses, restated from the published sources of
 * maxrossi91/r-index and simongog/sdsl-lite (absent offline -- PARITY UNPINNED):
 *
 *   ri::sparse_sd_vector / sdsl::sd_vector  Elias-Fano: low bits packed, high bits unary,
 *                                           select1/select0 over the high bit vector
 *   ri::huff_string / sdsl::wt_huff         Huffman-shaped wavelet tree with rank / select
 *   ri::rle_string (B = 2)                  `runs` marks the last position of every B-th run,
 *                                           `runs_per_letter[c]` the last position of every
 *                                           c-run inside the c-subsequence; positions are
 *                                           resolved by a block lookup + a walk over <= B runs
 *   thr_bv                                  per-letter Elias-Fano of the stored thresholds
 *                                           (include/thresholds_ds.hpp:384-440, 478-491)
 *
 * It exists (a) as a second, structurally independent restatement that must agree with the
 * flat oracle bit for bit (tests/test_oracle_t1.py) and (b) as a CPU baseline with the
 * reference's cost profile -- many dependent cache misses per searched character.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "spumoni_oracle.h"

static void *xc(size_t n, size_t sz) {
    void *p = calloc(n ? n : 1, sz);
    if (!p) abort();
    return p;
}

/* ---------------- bit vector with rank / select (sampled popcounts) ---------------- */
typedef struct {
    uint64_t nbits, nwords;
    uint64_t *w;
    uint64_t *cum1; /* ones before word i */
} bitv;

static void bv_init(bitv *b, uint64_t nbits) {
    b->nbits = nbits;
    b->nwords = (nbits + 63) / 64 + 1;
    b->w = (uint64_t *)xc(b->nwords, 8);
    b->cum1 = NULL;
}
static void bv_set(bitv *b, uint64_t i) { b->w[i >> 6] |= 1ull << (i & 63); }
static int bv_get(const bitv *b, uint64_t i) { return (b->w[i >> 6] >> (i & 63)) & 1; }
static void bv_finish(bitv *b) {
    b->cum1 = (uint64_t *)xc(b->nwords + 1, 8);
    for (uint64_t i = 0; i < b->nwords; ++i) b->cum1[i + 1] = b->cum1[i] + (uint64_t)__builtin_popcountll(b->w[i]);
}
static uint64_t bv_rank1(const bitv *b, uint64_t i) { /* ones in [0, i) */
    uint64_t wi = i >> 6, r = b->cum1[wi];
    if (i & 63) r += (uint64_t)__builtin_popcountll(b->w[wi] & ((1ull << (i & 63)) - 1));
    return r;
}
static uint64_t select_in_word(uint64_t x, uint64_t k) { /* position of the k-th (0-based) set bit */
    for (uint64_t i = 0; i < k; ++i) x &= x - 1;
    return (uint64_t)__builtin_ctzll(x);
}
static uint64_t bv_select1(const bitv *b, uint64_t k) { /* k-th one, 0-based */
    uint64_t lo = 0, hi = b->nwords; /* cum1[lo] <= k < cum1[hi] */
    while (hi - lo > 1) {
        uint64_t mid = (lo + hi) / 2;
        if (b->cum1[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    return lo * 64 + select_in_word(b->w[lo], k - b->cum1[lo]);
}
static uint64_t bv_select0(const bitv *b, uint64_t k) { /* k-th zero, 0-based */
    uint64_t lo = 0, hi = b->nwords; /* zeros before word: 64*i - cum1[i] */
    while (hi - lo > 1) {
        uint64_t mid = (lo + hi) / 2;
        if (mid * 64 - b->cum1[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    return lo * 64 + select_in_word(~b->w[lo], k - (lo * 64 - b->cum1[lo]));
}
static void bv_free(bitv *b) {
    free(b->w);
    free(b->cum1);
}

/* ---------------- Elias-Fano (sdsl::sd_vector construction rule) ---------------- */
typedef struct {
    uint64_t u, n; /* universe (size), number of ones */
    unsigned wl;
    uint64_t *low; /* n values of wl bits, packed */
    bitv high;
} efv;

static unsigned hi_bit(uint64_t x) { return x ? 63u - (unsigned)__builtin_clzll(x) : 0u; }

static uint64_t low_get(const efv *e, uint64_t i) {
    if (!e->wl) return 0;
    uint64_t bit = i * e->wl, wi = bit >> 6, sh = bit & 63;
    uint64_t v = e->low[wi] >> sh;
    if (sh + e->wl > 64) v |= e->low[wi + 1] << (64 - sh);
    return v & ((1ull << e->wl) - 1);
}

static void ef_build(efv *e, const uint64_t *ones, uint64_t n, uint64_t u) {
    memset(e, 0, sizeof *e);
    e->u = u;
    e->n = n;
    if (u == 0) return;
    unsigned logm = hi_bit(u) + 1, logn = n ? hi_bit(n) + 1 : 0;
    if (logm == logn && logn > 0) --logn;
    e->wl = logm - logn;
    e->low = (uint64_t *)xc((n * e->wl + 63) / 64 + 2, 8);
    bv_init(&e->high, n + (1ull << logn) + 1);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t p = ones[i];
        if (e->wl) {
            uint64_t v = p & ((1ull << e->wl) - 1), bit = i * e->wl, wi = bit >> 6, sh = bit & 63;
            e->low[wi] |= v << sh;
            if (sh + e->wl > 64) e->low[wi + 1] |= v >> (64 - sh);
        }
        bv_set(&e->high, (p >> e->wl) + i);
    }
    bv_finish(&e->high);
}
static uint64_t ef_select(const efv *e, uint64_t i) { /* position of the i-th one */
    return ((bv_select1(&e->high, i) - i) << e->wl) | low_get(e, i);
}
static uint64_t ef_rank(const efv *e, uint64_t x) { /* ones at positions < x */
    if (e->u == 0 || e->n == 0) return 0;
    if (x >= e->u) return e->n;
    const uint64_t hv = x >> e->wl, lv = x & ((e->wl ? (1ull << e->wl) : 1) - 1);
    /* the hv-th zero of `high` closes bucket hv: ones before it = elements with high part <= hv */
    uint64_t sel = bv_select0(&e->high, hv);
    uint64_t cnt = sel - hv; /* elements with high part <= hv */
    /* walk back over the elements of bucket hv whose low part is >= lv */
    while (cnt > 0 && sel > 0 && bv_get(&e->high, sel - 1) && low_get(e, cnt - 1) >= lv) {
        --sel;
        --cnt;
    }
    return cnt;
}
static void ef_free(efv *e) {
    free(e->low);
    if (e->u) bv_free(&e->high);
}

/* ---------------- Huffman-shaped wavelet tree (wt_huff / huff_string) ---------------- */
typedef struct {
    int child[2], parent, sym; /* sym >= 0: leaf */
    uint64_t bv_pos, size;
} wnode;
typedef struct {
    uint64_t size;
    int nnodes, root;
    wnode *nodes;
    bitv bv;
    int leaf_of[256];
    uint64_t count[256];
} wtree;

static void wt_build(wtree *t, const uint8_t *seq, uint64_t n) {
    memset(t, 0, sizeof *t);
    t->size = n;
    for (int c = 0; c < 256; ++c) t->leaf_of[c] = -1;
    for (uint64_t i = 0; i < n; ++i) t->count[seq[i]]++;
    t->nodes = (wnode *)xc(512, sizeof(wnode));
    int alive[512], na = 0;
    uint64_t weight[512];
    for (int c = 0; c < 256; ++c)
        if (t->count[c]) {
            int id = t->nnodes++;
            t->nodes[id].sym = c;
            t->nodes[id].child[0] = t->nodes[id].child[1] = -1;
            t->nodes[id].parent = -1;
            t->leaf_of[c] = id;
            weight[id] = t->count[c];
            alive[na++] = id;
        }
    if (na == 1) { /* one symbol: a root with a single leaf child keeps the code length at 1 */
        int id = t->nnodes++;
        t->nodes[id].sym = -1;
        t->nodes[id].child[0] = alive[0];
        t->nodes[id].child[1] = -1;
        t->nodes[id].parent = -1;
        t->nodes[alive[0]].parent = id;
        weight[id] = weight[alive[0]];
        alive[0] = id;
    }
    while (na > 1) { /* Huffman: merge the two lightest */
        int a = 0, b = 1;
        if (weight[alive[b]] < weight[alive[a]]) { int s = a; a = b; b = s; }
        for (int i = 2; i < na; ++i) {
            if (weight[alive[i]] < weight[alive[a]]) { b = a; a = i; }
            else if (weight[alive[i]] < weight[alive[b]]) b = i;
        }
        int id = t->nnodes++;
        t->nodes[id].sym = -1;
        t->nodes[id].child[0] = alive[a];
        t->nodes[id].child[1] = alive[b];
        t->nodes[id].parent = -1;
        t->nodes[alive[a]].parent = id;
        t->nodes[alive[b]].parent = id;
        weight[id] = weight[alive[a]] + weight[alive[b]];
        int lo = a < b ? a : b, hi = a < b ? b : a;
        alive[lo] = id;
        alive[hi] = alive[--na];
    }
    t->root = na ? alive[0] : -1;
    if (t->root < 0) return;
    /* sizes, bit offsets (inner nodes in id order), then the bits */
    uint64_t total = 0;
    for (int id = 0; id < t->nnodes; ++id) {
        wnode *nd = &t->nodes[id];
        if (nd->sym >= 0) nd->size = t->count[nd->sym];
    }
    for (int id = 0; id < t->nnodes; ++id) { /* children were created before parents */
        wnode *nd = &t->nodes[id];
        if (nd->sym < 0) {
            nd->size = t->nodes[nd->child[0]].size + (nd->child[1] >= 0 ? t->nodes[nd->child[1]].size : 0);
            nd->bv_pos = total;
            total += nd->size;
        }
    }
    bv_init(&t->bv, total + 1);
    /* path of every symbol from the root: bit b_d at depth d */
    uint64_t *fill = (uint64_t *)xc((size_t)t->nnodes, 8);
    int path_node[256][64], path_bit[256][64], path_len[256];
    for (int c = 0; c < 256; ++c) {
        path_len[c] = 0;
        if (t->leaf_of[c] < 0) continue;
        int stack_n[64], stack_b[64], d = 0, cur = t->leaf_of[c];
        while (t->nodes[cur].parent >= 0) {
            int p = t->nodes[cur].parent;
            stack_n[d] = p;
            stack_b[d] = (t->nodes[p].child[1] == cur);
            d++;
            cur = p;
        }
        for (int i = 0; i < d; ++i) {
            path_node[c][i] = stack_n[d - 1 - i];
            path_bit[c][i] = stack_b[d - 1 - i];
        }
        path_len[c] = d;
    }
    for (uint64_t i = 0; i < n; ++i) {
        int c = seq[i];
        for (int d = 0; d < path_len[c]; ++d) {
            int nd = path_node[c][d];
            if (path_bit[c][d]) bv_set(&t->bv, t->nodes[nd].bv_pos + fill[nd]);
            fill[nd]++;
        }
    }
    free(fill);
    bv_finish(&t->bv);
}
static uint8_t wt_access(const wtree *t, uint64_t i) {
    int nd = t->root;
    while (t->nodes[nd].sym < 0) {
        const wnode *x = &t->nodes[nd];
        uint64_t ones_before = bv_rank1(&t->bv, x->bv_pos + i) - bv_rank1(&t->bv, x->bv_pos);
        if (bv_get(&t->bv, x->bv_pos + i)) {
            i = ones_before;
            nd = x->child[1];
        } else {
            i = i - ones_before;
            nd = x->child[0];
        }
    }
    return (uint8_t)t->nodes[nd].sym;
}
static uint64_t wt_rank(const wtree *t, uint64_t i, uint8_t c) { /* occurrences of c in [0, i) */
    if (t->leaf_of[c] < 0) return 0;
    int path[64], bits[64], d = 0, cur = t->leaf_of[c];
    while (t->nodes[cur].parent >= 0) {
        int p = t->nodes[cur].parent;
        path[d] = p;
        bits[d] = (t->nodes[p].child[1] == cur);
        d++;
        cur = p;
    }
    for (int k = d - 1; k >= 0; --k) {
        const wnode *x = &t->nodes[path[k]];
        uint64_t ones_before = bv_rank1(&t->bv, x->bv_pos + i) - bv_rank1(&t->bv, x->bv_pos);
        i = bits[k] ? ones_before : i - ones_before;
    }
    return i;
}
static uint64_t wt_select(const wtree *t, uint64_t j, uint8_t c) { /* position of the j-th c (0-based) */
    int cur = t->leaf_of[c];
    uint64_t i = j;
    while (t->nodes[cur].parent >= 0) {
        int p = t->nodes[cur].parent;
        const wnode *x = &t->nodes[p];
        uint64_t base1 = bv_rank1(&t->bv, x->bv_pos);
        if (x->child[1] == cur)
            i = bv_select1(&t->bv, base1 + i) - x->bv_pos;
        else
            i = bv_select0(&t->bv, (x->bv_pos - base1) + i) - x->bv_pos;
        cur = p;
    }
    return i;
}
static void wt_free(wtree *t) {
    free(t->nodes);
    if (t->root >= 0) bv_free(&t->bv);
}

/* ---------------- ri::rle_string + thr_bv ---------------- */
struct orc_t1_index {
    uint64_t n, R, B;
    efv runs;
    efv runs_per_letter[256];
    wtree run_heads;
    efv thresholds_per_letter[256];
    uint64_t F[256];
    uint64_t *samples_start, *samples_last, *start_runs_doc, *end_runs_doc;
};

orc_t1_index *orc_t1_build(const uint8_t *heads_in, const uint64_t *lens, const uint64_t *thr, uint64_t r,
                           const uint64_t *samples_start, const uint64_t *samples_last,
                           const uint64_t *start_runs_doc, const uint64_t *end_runs_doc) {
    orc_t1_index *ix = (orc_t1_index *)xc(1, sizeof *ix);
    ix->R = r;
    ix->B = 2; /* include/ms_rle_string.hpp:37 */
    uint8_t *heads = (uint8_t *)xc(r, 1);
    uint64_t n = 0, nb = 0;
    uint64_t *runs_onset = (uint64_t *)xc(r / 2 + 2, 8);
    uint64_t *cnt = (uint64_t *)xc(256, 8), *rc = (uint64_t *)xc(256, 8);
    for (uint64_t i = 0; i < r; ++i) rc[heads_in[i] <= 1 ? 1 : heads_in[i]]++;
    uint64_t *per[256], *tper[256], fill[256], tfill[256];
    for (int c = 0; c < 256; ++c) {
        per[c] = (uint64_t *)xc(rc[c], 8);
        tper[c] = (uint64_t *)xc(rc[c], 8);
        fill[c] = tfill[c] = 0;
    }
    for (uint64_t i = 0; i < r; ++i) { /* ms_rle_string.hpp:245-262 */
        uint8_t c = heads_in[i] <= 1 ? 1 : heads_in[i];
        heads[i] = c;
        if (i % ix->B == ix->B - 1) runs_onset[nb++] = n + lens[i] - 1;
        cnt[c] += lens[i];
        per[c][fill[c]++] = cnt[c] - 1;
        n += lens[i];
        if (thr && thr[i] > 0) tper[c][tfill[c]++] = thr[i]; /* thresholds_ds.hpp:421-423 */
    }
    ix->n = n;
    ef_build(&ix->runs, runs_onset, nb, n);
    uint64_t acc = 0;
    for (int c = 0; c < 256; ++c) {
        ef_build(&ix->runs_per_letter[c], per[c], fill[c], cnt[c]);
        ef_build(&ix->thresholds_per_letter[c], tper[c], tfill[c], rc[c] ? n : 0); /* :424,430 */
        ix->F[c] = acc;
        acc += cnt[c];
        free(per[c]);
        free(tper[c]);
    }
    wt_build(&ix->run_heads, heads, r);
#define COPY(dst, src)                                   \
    if (src) {                                           \
        ix->dst = (uint64_t *)xc(r, 8);                  \
        memcpy(ix->dst, src, r * 8);                     \
    }
    COPY(samples_start, samples_start)
    COPY(samples_last, samples_last)
    COPY(start_runs_doc, start_runs_doc)
    COPY(end_runs_doc, end_runs_doc)
#undef COPY
    free(heads);
    free(runs_onset);
    free(cnt);
    free(rc);
    return ix;
}

void orc_t1_free(orc_t1_index *ix) {
    if (!ix) return;
    ef_free(&ix->runs);
    for (int c = 0; c < 256; ++c) {
        ef_free(&ix->runs_per_letter[c]);
        ef_free(&ix->thresholds_per_letter[c]);
    }
    wt_free(&ix->run_heads);
    free(ix->samples_start);
    free(ix->samples_last);
    free(ix->start_runs_doc);
    free(ix->end_runs_doc);
    free(ix);
}

/* length of run k: head c, its rank among the c-runs, two selects in the c-subsequence */
static uint64_t t1_run_len(const orc_t1_index *ix, uint64_t k, uint8_t *head_out) {
    uint8_t c = wt_access(&ix->run_heads, k);
    uint64_t j = wt_rank(&ix->run_heads, k, c);
    uint64_t endp = ef_select(&ix->runs_per_letter[c], j);
    uint64_t startp = j ? ef_select(&ix->runs_per_letter[c], j - 1) + 1 : 0;
    if (head_out) *head_out = c;
    return endp + 1 - startp;
}

/* run containing position p and the start of that run: block lookup in `runs`, then a walk
 * over at most B runs (ri::rle_string::run_of_position / operator[] / rank share this) */
static uint64_t t1_locate(const orc_t1_index *ix, uint64_t p, uint64_t *run_start) {
    uint64_t last_block = ef_rank(&ix->runs, p); /* B-blocks that end before p */
    uint64_t current_run = last_block * ix->B;
    uint64_t pos = last_block ? ef_select(&ix->runs, last_block - 1) + 1 : 0;
    for (;;) {
        uint64_t len = t1_run_len(ix, current_run, NULL);
        if (pos + len > p) break;
        pos += len;
        current_run++;
    }
    if (run_start) *run_start = pos;
    return current_run;
}

uint64_t orc_t1_run_of_position(const orc_t1_index *ix, uint64_t p) { return t1_locate(ix, p, NULL); }
uint8_t orc_t1_at(const orc_t1_index *ix, uint64_t p) { return wt_access(&ix->run_heads, t1_locate(ix, p, NULL)); }

uint64_t orc_t1_rank(const orc_t1_index *ix, uint64_t p, uint8_t c) {
    if (ix->runs_per_letter[c].u == 0) return 0;          /* letter does not exist */
    if (p == ix->n) return ix->runs_per_letter[c].u;
    uint64_t start = 0;
    uint64_t k = t1_locate(ix, p, &start);
    uint64_t rk = wt_rank(&ix->run_heads, k, c);          /* c-runs before run k */
    uint64_t tail = (wt_access(&ix->run_heads, k) == c) ? p - start : 0;
    if (rk == 0) return tail;
    return ef_select(&ix->runs_per_letter[c], rk - 1) + 1 + tail;
}

uint64_t orc_t1_select(const orc_t1_index *ix, uint64_t i, uint8_t c) {
    if (i >= ix->runs_per_letter[c].u) {
        fprintf(stderr, "oracle T1: select out of range\n");
        abort();
    }
    uint64_t j = ef_rank(&ix->runs_per_letter[c], i);     /* c-runs that end before the i-th c */
    uint64_t before = j ? ef_select(&ix->runs_per_letter[c], j - 1) + 1 : 0;
    uint64_t k = wt_select(&ix->run_heads, j, c);         /* the j-th c-run */
    /* start of run k in the text order: block start + walk */
    uint64_t blk = k / ix->B;
    uint64_t pos = blk ? ef_select(&ix->runs, blk - 1) + 1 : 0;
    for (uint64_t x = blk * ix->B; x < k; ++x) pos += t1_run_len(ix, x, NULL);
    return pos + (i - before);
}

uint64_t orc_t1_threshold(const orc_t1_index *ix, uint64_t k) { /* thresholds_ds.hpp:478-491 */
    uint8_t c = wt_access(&ix->run_heads, k);
    uint64_t rank = wt_rank(&ix->run_heads, k, c);
    if (rank == 0) return 0;
    if (rank - 1 >= ix->thresholds_per_letter[c].n) {
        fprintf(stderr, "oracle T1: threshold select past the stored thresholds\n");
        abort();
    }
    return ef_select(&ix->thresholds_per_letter[c], rank - 1);
}

uint64_t orc_t1_LF(const orc_t1_index *ix, uint64_t p, uint8_t c) { return ix->F[c] + orc_t1_rank(ix, p, c); }

static uint64_t t1_last_run_sample(const orc_t1_index *ix) { return (ix->samples_last[ix->R - 1] + 1) % ix->n; }

#define QFN(name) orc_t1_##name
#define QIDX orc_t1_index
#define Q_SIZE(ix) ((ix)->n)
#define Q_NUMBER_OF_RUNS(ix) ((ix)->R)
#define Q_NUMBER_OF_LETTER(ix, c) ((ix)->runs_per_letter[(c)].u)
#define Q_AT(ix, p) orc_t1_at((ix), (p))
#define Q_RANK(ix, p, c) orc_t1_rank((ix), (p), (c))
#define Q_SELECT(ix, i, c) orc_t1_select((ix), (i), (c))
#define Q_RUN_OF_POSITION(ix, p) orc_t1_run_of_position((ix), (p))
#define Q_THRESHOLD(ix, k) orc_t1_threshold((ix), (k))
#define Q_LF(ix, p, c) orc_t1_LF((ix), (p), (c))
#define Q_LAST_RUN_SAMPLE(ix) t1_last_run_sample((ix))
#define Q_START_DOC(ix, k) ((ix)->start_runs_doc[(k)])
#define Q_END_DOC(ix, k) ((ix)->end_runs_doc[(k)])
#define Q_SAMPLE_START(ix, k) ((ix)->samples_start[(k)])
#define Q_SAMPLE_LAST(ix, k) ((ix)->samples_last[(k)])
#include "orc_queries.inc"

void orc_t1_pml_batch(const orc_t1_index *ix, const uint8_t *seqs, const uint64_t *offs, uint64_t nreads,
                      uint32_t *out_lengths, uint32_t *out_docs, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        uint64_t cap = 0;
        uint64_t *lens = NULL, *docs = NULL;
#pragma omp for schedule(dynamic, 16)
        for (uint64_t q = 0; q < nreads; ++q) {
            uint64_t m = offs[q + 1] - offs[q];
            if (m > cap) {
                cap = 2 * m;
                lens = (uint64_t *)realloc(lens, cap * 8);
                docs = (uint64_t *)realloc(docs, cap * 8);
            }
            const char *pat = (const char *)(seqs + offs[q]);
            if (out_docs)
                orc_t1_pml_query_doc(ix, pat, m, lens, docs);
            else
                orc_t1_pml_query(ix, pat, m, lens);
            for (uint64_t x = 0; x < m; ++x) {
                out_lengths[offs[q] + x] = (uint32_t)lens[x];
                if (out_docs) out_docs[offs[q] + x] = (uint32_t)docs[x];
            }
        }
        free(lens);
        free(docs);
    }
}

void orc_t1_ms_batch(const orc_t1_index *ix, const uint8_t *seqs, const uint64_t *offs, uint64_t nreads,
                     uint64_t *out_pointers, uint32_t *out_docs, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        uint64_t cap = 0;
        uint64_t *docs = NULL;
#pragma omp for schedule(dynamic, 16)
        for (uint64_t q = 0; q < nreads; ++q) {
            uint64_t m = offs[q + 1] - offs[q];
            if (m > cap) {
                cap = 2 * m;
                docs = (uint64_t *)realloc(docs, cap * 8);
            }
            const char *pat = (const char *)(seqs + offs[q]);
            if (out_docs) {
                orc_t1_ms_query_doc(ix, pat, m, out_pointers + offs[q], docs);
                for (uint64_t x = 0; x < m; ++x) out_docs[offs[q] + x] = (uint32_t)docs[x];
            } else {
                orc_t1_ms_query(ix, pat, m, out_pointers + offs[q]);
            }
        }
        free(docs);
    }
}
