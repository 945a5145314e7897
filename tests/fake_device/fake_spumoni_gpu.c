/*
 * fake_spumoni_gpu.c -- TEST INFRASTRUCTURE ONLY.  NOT the product, never built by __graft_entry__.build(), never
 * loaded by anything under spumoni_amd/.
 *
 * The C-ABI of include/spumoni_gpu.h answered by the CPU oracle (oracle/), so that the C++ HOST above the boundary
 * -- `spumoni run`'s harness: parser, the queue, a worker thread per "device", the ordered writer, the report thread
 * (spumoni_amd/csrc/host/classify.cpp) -- can be exercised, also under ThreadSanitizer / AddressSanitizer, on a
 * machine without a GPU (the driver's CPU tier).  tests/test_host_harness_cpu.py builds it into a temporary directory
 * as libspumoni_gpu.so and puts that directory on LD_LIBRARY_PATH of the ordinary host binary.
 *
 * What is under test there is the host code.  Results computed here say nothing about the HIP path: that is held
 * against the oracle on the GPU (tests/test_gpu_*.py, -m gpu).  The product library keeps failing loudly without a
 * device (tests/test_abi.py::test_no_gpu_fails_loudly); nothing in the repository routes a product call here.
 *
 * Entry points without a meaning off the device (device-buffer forms, the flat-layout cache) return an error.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spumoni_gpu.h"
#include "../../oracle/spumoni_oracle.h"

static __thread char g_err[512] = "";
static void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *spx_last_error(void) { return g_err; }
const char *spx_version(void) { return "fake-device (the CPU oracle behind the C-ABI; tests only)"; }

int spx_device_count(void) {
    const char *e = getenv("FAKE_SPX_DEVICES");
    return e ? atoi(e) : 1;
}

/* what replicas share (spx_index_clone): the oracle index and the text, read-only once built */
typedef struct core {
    orc_index *orc;
    int has_samples, has_docs;
    uint8_t *text;
    uint64_t n_text;
    int refs;
    pthread_mutex_t mu;
} core;

struct spx_index {
    core *c;
    int device;
    char tag[128];
    uint8_t charhash[4];
    /* spx_query_text_begin -> _fetch */
    char *tx[3];
    uint64_t *ls[3];
    uint64_t tb[3];
    uint64_t tn;
    int ready;
    /* the class records travel with the text: the caller's buffer is POISONED at begin and filled at fetch, as the real
       library does (spx_query_text_fetch) -- a harness that reads them in between, or frees them, shows up on the CPU tier */
    spx_class *cls_stash, *cls_out;
};

static spx_index *new_handle(core *c, int device) {
    spx_index *ix = (spx_index *)calloc(1, sizeof *ix);
    ix->c = c;
    ix->device = device;
    orc_digest_default_charhash(ix->charhash);
    return ix;
}

static int check_device(int device) {
    if (device < 0 || device >= spx_device_count()) {
        set_error("fake device %d out of range (FAKE_SPX_DEVICES=%d)", device, spx_device_count());
        return 0;
    }
    return 1;
}

spx_index *spx_index_from_runs(const uint8_t *heads, const uint64_t *lens, const uint64_t *thr, uint64_t r,
                               const uint64_t *ssa, const uint64_t *esa, const uint64_t *doc_start,
                               const uint64_t *doc_end, int where, int device) {
    if (!heads || !lens || !thr || r == 0 || where != 0) {
        set_error("fake device: host arrays only (where = 0), r > 0");
        return NULL;
    }
    if (!check_device(device)) return NULL;
    orc_index *o = orc_build(heads, lens, thr, r, ssa, esa, doc_start, doc_end);
    if (!o) {
        set_error("fake device: the oracle refused the run arrays");
        return NULL;
    }
    core *c = (core *)calloc(1, sizeof *c);
    c->orc = o;
    c->has_samples = ssa && esa;
    c->has_docs = doc_start && doc_end;
    c->refs = 1;
    pthread_mutex_init(&c->mu, NULL);
    return new_handle(c, device);
}

spx_index *spx_index_load_raw(const char *prefix, int mode, int device) {
    if (!check_device(device)) return NULL;
    orc_index *o = orc_load_raw(prefix, mode == SPX_MODE_MS);
    if (!o) {
        set_error("fake device: cannot read the raw files of %s", prefix);
        return NULL;
    }
    core *c = (core *)calloc(1, sizeof *c);
    c->orc = o;
    c->has_samples = mode == SPX_MODE_MS;
    c->refs = 1;
    pthread_mutex_init(&c->mu, NULL);
    return new_handle(c, device);
}

static void free_text_state(spx_index *ix) {
    for (int i = 0; i < 3; ++i) {
        free(ix->tx[i]);
        free(ix->ls[i]);
        ix->tx[i] = NULL;
        ix->ls[i] = NULL;
        ix->tb[i] = 0;
    }
    ix->ready = 0;
}

void spx_index_free(spx_index *ix) {
    if (!ix) return;
    core *c = ix->c;
    pthread_mutex_lock(&c->mu);
    const int left = --c->refs;
    pthread_mutex_unlock(&c->mu);
    if (left == 0) {
        orc_free(c->orc);
        free(c->text);
        pthread_mutex_destroy(&c->mu);
        free(c);
    }
    free_text_state(ix);
    free(ix->cls_stash);
    free(ix);
}

spx_index *spx_index_clone(spx_index *src, int device) {
    if (!src || !check_device(device)) return NULL;
    pthread_mutex_lock(&src->c->mu);
    src->c->refs++;
    pthread_mutex_unlock(&src->c->mu);
    spx_index *ix = new_handle(src->c, device);
    memcpy(ix->tag, src->tag, sizeof ix->tag);
    memcpy(ix->charhash, src->charhash, sizeof ix->charhash);
    return ix;
}

int spx_index_stats(const spx_index *ix, uint64_t *n, uint64_t *r) {
    if (!ix) return SPX_E_ARG;
    if (n) *n = ix->c->orc->n;
    if (r) *r = ix->c->orc->r;
    return SPX_OK;
}
int spx_index_device_bytes(const spx_index *ix, uint64_t *bytes) {
    if (!ix || !bytes) return SPX_E_ARG;
    *bytes = 0;
    return SPX_OK;
}

int spx_index_set_text(spx_index *ix, const uint8_t *text, uint64_t n_text, int where) {
    if (!ix || !text || (where & 1)) {
        set_error("fake device: host text only");
        return SPX_E_ARG;
    }
    if (!(where & SPX_TEXT_UNCHECKED) && n_text + 1 != ix->c->orc->n) {
        set_error("the text has %llu characters, the index was built from %llu", (unsigned long long)n_text,
                  (unsigned long long)(ix->c->orc->n - 1));
        return SPX_E_FORMAT;
    }
    core *c = ix->c;
    pthread_mutex_lock(&c->mu);
    free(c->text);
    c->text = (uint8_t *)malloc(n_text ? n_text : 1);
    memcpy(c->text, text, n_text);
    c->n_text = n_text;
    pthread_mutex_unlock(&c->mu);
    return SPX_OK;
}
int spx_index_rebuild_text(spx_index *ix) {
    (void)ix;
    set_error("fake device: the text is not rebuilt from the index here (give SPUMONI_TEXT)");
    return SPX_E_UNSUPPORTED;
}
int spx_index_copy_text(spx_index *ix, uint8_t *out, uint64_t capacity, int where, uint64_t *n_text) {
    if (!ix || where) return SPX_E_ARG;
    if (n_text) *n_text = ix->c->n_text;
    if (out) {
        if (capacity < ix->c->n_text) return SPX_E_ARG;
        memcpy(out, ix->c->text, ix->c->n_text);
    }
    return SPX_OK;
}

int spx_index_save(spx_index *ix, const char *path) {
    (void)ix;
    (void)path;
    set_error("fake device: no flat layout to save");
    return SPX_E_UNSUPPORTED;
}
spx_index *spx_index_load_flat(const char *path, int device) {
    (void)path;
    (void)device;
    set_error("fake device: no flat layout to load");
    return NULL;
}
int spx_index_set_source_tag(spx_index *ix, const char *tag) {
    if (!ix || !tag || strlen(tag) > 127) return SPX_E_ARG;
    strcpy(ix->tag, tag);
    return SPX_OK;
}
const char *spx_index_source_tag(const spx_index *ix) { return ix ? ix->tag : ""; }
int spx_index_describe(const spx_index *ix, char *buf, size_t cap) {
    if (!ix || !buf || !cap) return SPX_E_ARG;
    snprintf(buf, cap, "{\"layout\": \"fake-device\", \"n\": %llu, \"r\": %llu}", (unsigned long long)ix->c->orc->n,
             (unsigned long long)ix->c->orc->r);
    return SPX_OK;
}

/* ---- queries ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t *len, *doc;
    uint64_t *ptr;
} vals;
static void free_vals(vals *v) {
    free(v->len);
    free(v->doc);
    free(v->ptr);
}

/* lengths (PML, or MS with a text), pointers (MS), document ids at the reads' offsets; classes from the lengths */
static int run_query(spx_index *ix, int mode, const uint8_t *seqs, const uint64_t *offs, uint64_t nreads, int want_len,
                     int want_doc, spx_class *out_class, uint64_t bin_width, uint64_t max_value_thr, vals *v) {
    core *c = ix->c;
    memset(v, 0, sizeof *v);
    if (mode != SPX_MODE_PML && mode != SPX_MODE_MS) {
        set_error("unknown mode %d", mode);
        return SPX_E_ARG;
    }
    if (mode == SPX_MODE_MS && !c->has_samples) {
        set_error("MS query on an index built without SA samples");
        return SPX_E_ARG;
    }
    if (want_doc && !c->has_docs) {
        set_error("document ids asked of an index built without a document array");
        return SPX_E_ARG;
    }
    if (out_class && bin_width == 0) {
        set_error("bin_width must be >= 1");
        return SPX_E_ARG;
    }
    const uint64_t total = nreads ? offs[nreads] : 0;
    const int need_len = want_len || out_class;
    if (mode == SPX_MODE_MS && need_len && !c->text) {
        set_error("MS lengths need the text (spx_index_set_text)");
        return SPX_E_ARG;
    }
    if (need_len) v->len = (uint32_t *)calloc(total + 1, 4);
    if (want_doc) v->doc = (uint32_t *)calloc(total + 1, 4);
    if (mode == SPX_MODE_PML) {
        uint32_t *len = v->len ? v->len : (uint32_t *)calloc(total + 1, 4);
        orc_pml_batch(c->orc, seqs, offs, nreads, len, v->doc, 1);
        if (!v->len) free(len);
    } else {
        v->ptr = (uint64_t *)calloc(total + 1, 8);
        orc_ms_batch(c->orc, seqs, offs, nreads, v->ptr, v->doc, need_len ? c->text : NULL, c->n_text, v->len, 1);
    }
    if (out_class && nreads) {
        uint8_t *found = (uint8_t *)malloc(nreads);
        uint32_t *above = (uint32_t *)malloc(nreads * 4), *below = (uint32_t *)malloc(nreads * 4);
        uint64_t *sum = (uint64_t *)malloc(nreads * 8);
        orc_classify_batch(v->len, offs, nreads, bin_width, max_value_thr, found, above, below, sum);
        for (uint64_t q = 0; q < nreads; ++q) {
            out_class[q].sum_max_bin_values = sum[q];
            out_class[q].bins_above = above[q];
            out_class[q].bins_below = below[q];
        }
        free(found);
        free(above);
        free(below);
        free(sum);
    }
    return SPX_OK;
}

static int query_host(spx_index *ix, int mode, const uint8_t *seqs, const uint64_t *offs, uint64_t nreads, void *out_lengths,
                      uint64_t *out_pointers, void *out_docs, spx_class *out_class, uint64_t bin_width,
                      uint64_t max_value_thr, int width) {
    if (!ix || !seqs || !offs) {
        set_error("index, seqs and offsets must be non-null");
        return SPX_E_ARG;
    }
    if (out_pointers && mode != SPX_MODE_MS) {
        set_error("pointers are an MS output");
        return SPX_E_ARG;
    }
    if (width == 2)
        for (uint64_t q = 0; q < nreads; ++q)
            if (offs[q + 1] - offs[q] >= 65536) {
                set_error("read %llu has 65536 characters or more: use the 32-bit entry point", (unsigned long long)q);
                return SPX_E_ARG;
            }
    vals v;
    const int rc = run_query(ix, mode, seqs, offs, nreads, out_lengths != NULL, out_docs != NULL, out_class, bin_width,
                             max_value_thr, &v);
    if (rc != SPX_OK) {
        free_vals(&v);
        return rc;
    }
    const uint64_t total = nreads ? offs[nreads] : 0;
    for (uint64_t i = 0; i < total; ++i) {
        if (out_lengths) {
            if (width == 2)
                ((uint16_t *)out_lengths)[i] = (uint16_t)v.len[i];
            else
                ((uint32_t *)out_lengths)[i] = v.len[i];
        }
        if (out_docs) {
            if (width == 2)
                ((uint16_t *)out_docs)[i] = (uint16_t)v.doc[i];
            else
                ((uint32_t *)out_docs)[i] = v.doc[i];
        }
        if (out_pointers) out_pointers[i] = v.ptr[i];
    }
    free_vals(&v);
    return SPX_OK;
}

int spx_query_batch(spx_index *ix, int mode, const uint8_t *seqs, const uint64_t *offsets, uint64_t nreads,
                    uint32_t *out_lengths, uint64_t *out_pointers, uint32_t *out_docs, spx_class *out_class,
                    uint64_t bin_width, uint64_t max_value_thr) {
    return query_host(ix, mode, seqs, offsets, nreads, out_lengths, out_pointers, out_docs, out_class, bin_width, max_value_thr, 4);
}
int spx_query_batch16(spx_index *ix, int mode, const uint8_t *seqs, const uint64_t *offsets, uint64_t nreads,
                      uint16_t *out_lengths, uint64_t *out_pointers, uint16_t *out_docs, spx_class *out_class,
                      uint64_t bin_width, uint64_t max_value_thr) {
    return query_host(ix, mode, seqs, offsets, nreads, out_lengths, out_pointers, out_docs, out_class, bin_width, max_value_thr, 2);
}

static int no_device_memory(void) {
    set_error("fake device: there is no device memory (host-buffer entry points only)");
    return SPX_E_NODEVICE;
}
int spx_query_batch_device(spx_index *ix, int mode, const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t nreads,
                           uint64_t total_chars, uint32_t *d_out_lengths, uint64_t *d_out_pointers, uint32_t *d_out_docs,
                           spx_class *d_out_class, uint64_t bin_width, uint64_t max_value_thr, void *stream) {
    (void)ix, (void)mode, (void)d_seqs, (void)d_offsets, (void)nreads, (void)total_chars, (void)d_out_lengths;
    (void)d_out_pointers, (void)d_out_docs, (void)d_out_class, (void)bin_width, (void)max_value_thr, (void)stream;
    return no_device_memory();
}
int spx_query_batch_device16(spx_index *ix, int mode, const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t nreads,
                             uint64_t total_chars, uint16_t *d_out_lengths, uint64_t *d_out_pointers,
                             uint16_t *d_out_docs, spx_class *d_out_class, uint64_t bin_width, uint64_t max_value_thr,
                             void *stream) {
    (void)ix, (void)mode, (void)d_seqs, (void)d_offsets, (void)nreads, (void)total_chars, (void)d_out_lengths;
    (void)d_out_pointers, (void)d_out_docs, (void)d_out_class, (void)bin_width, (void)max_value_thr, (void)stream;
    return no_device_memory();
}
int spx_last_walk_stats(spx_index *ix, spx_walk_stats *out) {
    if (!ix || !out) return SPX_E_ARG;
    memset(out, 0, sizeof *out);
    return SPX_OK;
}
int spx_last_chunk_stats(spx_index *ix, uint64_t out[4]) {
    if (!ix || !out) return SPX_E_ARG;
    out[0] = out[1] = out[2] = out[3] = 0;
    return SPX_OK;
}
void *spx_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void spx_host_free(void *p) { free(p); }
/* (nothing to lock here: "registered" memory is written by memcpy like any other, which is what lets the CPU tier run the
 * harness's text-straight-into-the-file path) */
int spx_host_register(void *p, size_t bytes) { (void)bytes; return p ? SPX_OK : SPX_E_ARG; }
int spx_host_unregister(void *p) { (void)p; return SPX_OK; }

/* ---- digestion -------------------------------------------------------------------------------------------- */
uint64_t spx_digest_capacity(int kind, uint32_t k, uint64_t total_chars) {
    return (kind == SPX_DIGEST_DNA ? total_chars * (k ? k : 1) : total_chars) + 64;
}
static int check_digest(int kind, uint32_t k, uint32_t w) {
    if ((kind != SPX_DIGEST_PROMOTED && kind != SPX_DIGEST_DNA) || k < 1 || k > 4 || w < k) {
        set_error("digestion: kind 1 / 2, k in [1, 4], w >= k");
        return SPX_E_ARG;
    }
    return SPX_OK;
}
int spx_digest_batch(spx_index *ix, int kind, uint32_t k, uint32_t w, const uint8_t *seqs, const uint64_t *offsets,
                     uint64_t nreads, uint8_t *out_seqs, uint64_t out_capacity, uint64_t *out_offsets) {
    if (!ix || !seqs || !offsets || !out_offsets) return SPX_E_ARG;
    if (check_digest(kind, k, w) != SPX_OK) return SPX_E_ARG;
    const uint64_t cap = spx_digest_capacity(kind, k, nreads ? offsets[nreads] : 0);
    uint8_t *tmp = (uint8_t *)malloc(cap);
    orc_digest_batch(kind, k, w, ix->charhash, seqs, offsets, nreads, tmp, cap, out_offsets);
    int rc = SPX_OK;
    if (out_offsets[nreads] > out_capacity || !out_seqs) {
        if (out_offsets[nreads]) {
            set_error("output holds %llu bytes, the digested reads have %llu", (unsigned long long)out_capacity,
                      (unsigned long long)out_offsets[nreads]);
            rc = SPX_E_ARG;
        }
    } else {
        memcpy(out_seqs, tmp, out_offsets[nreads]);
    }
    free(tmp);
    return rc;
}
int spx_digest_batch_device(spx_index *ix, int kind, uint32_t k, uint32_t w, const uint8_t *d_seqs, const uint64_t *d_offsets,
                            uint64_t nreads, uint64_t total_chars, uint8_t *d_out_seqs, uint64_t out_capacity,
                            uint64_t *d_out_offsets, void *stream) {
    (void)ix, (void)kind, (void)k, (void)w, (void)d_seqs, (void)d_offsets, (void)nreads, (void)total_chars;
    (void)d_out_seqs, (void)out_capacity, (void)d_out_offsets, (void)stream;
    return no_device_memory();
}

int spx_digest_query_batch_device(spx_index *ix, int mode, int kind, uint32_t k, uint32_t w, const uint8_t *d_seqs,
                                  const uint64_t *d_offsets, uint64_t nreads, uint64_t total_chars, uint8_t *d_digested,
                                  uint64_t digested_capacity, uint64_t *d_out_offsets, uint32_t *d_out_lengths,
                                  uint64_t *d_out_pointers, uint32_t *d_out_docs, spx_class *d_out_class, uint64_t bin_width,
                                  uint64_t max_value_thr, void *stream) {
    (void)ix, (void)mode, (void)kind, (void)k, (void)w, (void)d_seqs, (void)d_offsets, (void)nreads, (void)total_chars;
    (void)d_digested, (void)digested_capacity, (void)d_out_offsets, (void)d_out_lengths, (void)d_out_pointers;
    (void)d_out_docs, (void)d_out_class, (void)bin_width, (void)max_value_thr, (void)stream;
    return no_device_memory();
}
int spx_digest_query_batch_device16(spx_index *ix, int mode, int kind, uint32_t k, uint32_t w, const uint8_t *d_seqs,
                                    const uint64_t *d_offsets, uint64_t nreads, uint64_t total_chars, uint8_t *d_digested,
                                    uint64_t digested_capacity, uint64_t *d_out_offsets, uint16_t *d_out_lengths,
                                    uint64_t *d_out_pointers, uint16_t *d_out_docs, spx_class *d_out_class, uint64_t bin_width,
                                    uint64_t max_value_thr, void *stream) {
    (void)ix, (void)mode, (void)kind, (void)k, (void)w, (void)d_seqs, (void)d_offsets, (void)nreads, (void)total_chars;
    (void)d_digested, (void)digested_capacity, (void)d_out_offsets, (void)d_out_lengths, (void)d_out_pointers;
    (void)d_out_docs, (void)d_out_class, (void)bin_width, (void)max_value_thr, (void)stream;
    return no_device_memory();
}

/* digested reads + their offsets (malloc'ed) */
static int digest_reads(spx_index *ix, int kind, uint32_t k, uint32_t w, const uint8_t *seqs, const uint64_t *offs,
                        uint64_t nreads, uint8_t **dseq, uint64_t **doff) {
    if (check_digest(kind, k, w) != SPX_OK) return SPX_E_ARG;
    const uint64_t cap = spx_digest_capacity(kind, k, nreads ? offs[nreads] : 0);
    *dseq = (uint8_t *)malloc(cap);
    *doff = (uint64_t *)malloc((nreads + 1) * 8);
    orc_digest_batch(kind, k, w, ix->charhash, seqs, offs, nreads, *dseq, cap, *doff);
    return SPX_OK;
}

int spx_digest_query_batch(spx_index *ix, int mode, int kind, uint32_t k, uint32_t w, const uint8_t *seqs,
                           const uint64_t *offsets, uint64_t nreads, uint64_t *out_offsets, uint64_t out_capacity,
                           uint32_t *out_lengths, uint64_t *out_pointers, uint32_t *out_docs, spx_class *out_class,
                           uint64_t bin_width, uint64_t max_value_thr) {
    if (!ix || !seqs || !offsets || !out_offsets) {
        set_error("index, seqs, offsets and out_offsets must be non-null");
        return SPX_E_ARG;
    }
    uint8_t *dseq = NULL;
    uint64_t *doff = NULL;
    int rc = digest_reads(ix, kind, k, w, seqs, offsets, nreads, &dseq, &doff);
    if (rc == SPX_OK) {
        memcpy(out_offsets, doff, (nreads + 1) * 8);
        if (doff[nreads] > out_capacity) {
            set_error("output buffers hold %llu entries, the digested reads have %llu characters",
                      (unsigned long long)out_capacity, (unsigned long long)doff[nreads]);
            rc = SPX_E_ARG;
        } else {
            rc = query_host(ix, mode, dseq, doff, nreads, out_lengths, out_pointers, out_docs, out_class, bin_width,
                            max_value_thr, 4);
        }
    }
    free(dseq);
    free(doff);
    return rc;
}

int spx_set_option(spx_index *ix, const char *key, int64_t value) {
    if (!ix || !key) return SPX_E_ARG;
    if (!strcmp(key, "minimizer_charhash"))
        for (int c = 0; c < 4; ++c) ix->charhash[c] = (uint8_t)((uint64_t)value >> (8 * c));
    return SPX_OK; /* (the other knobs tune kernels that do not exist here) */
}

/* ---- the vectors as text (compute_ms_pml.cpp:1001-1010, 1182-1205) ------------------------------------------- */
static uint32_t dec_width(uint64_t v) {
    uint32_t d = 1;
    while (v >= 10) {
        v /= 10;
        ++d;
    }
    return d;
}

int spx_query_text_begin(spx_index *ix, int mode, int digest_kind, uint32_t k, uint32_t w, const uint8_t *seqs,
                         const uint64_t *offsets, uint64_t nreads, const uint32_t *gap, uint32_t streams,
                         spx_class *out_class, uint64_t bin_width, uint64_t max_value_thr, uint64_t out_bytes[3]) {
    if (!ix || !seqs || !offsets || !out_bytes) {
        set_error("index, seqs, offsets and out_bytes must be non-null");
        return SPX_E_ARG;
    }
    if ((streams & SPX_TEXT_POINTERS) && mode != SPX_MODE_MS) {
        set_error("the pointers stream is only produced in MS mode");
        return SPX_E_ARG;
    }
    free_text_state(ix);
    uint8_t *dseq = NULL;
    uint64_t *doff = NULL;
    const uint8_t *qs = seqs;
    const uint64_t *qo = offsets;
    if (digest_kind) {
        const int rc = digest_reads(ix, digest_kind, k, w, seqs, offsets, nreads, &dseq, &doff);
        if (rc != SPX_OK) return rc;
        qs = dseq;
        qo = doff;
    }
    vals v;
    const int want_len = (streams & SPX_TEXT_LENGTHS) != 0, want_doc = (streams & SPX_TEXT_DOCS) != 0;
    free(ix->cls_stash);
    ix->cls_stash = out_class && nreads ? (spx_class *)malloc(nreads * sizeof(spx_class)) : NULL;
    ix->cls_out = out_class;
    int rc = run_query(ix, mode, qs, qo, nreads, want_len, want_doc, out_class ? (ix->cls_stash ? ix->cls_stash : out_class) : NULL,
                       bin_width, max_value_thr, &v);
    if (out_class && nreads) memset(out_class, 0xA5, nreads * sizeof(spx_class)); /* not yours before the fetch */
    if (rc == SPX_OK) {
        for (int i = 0; i < 3; ++i) {
            out_bytes[i] = 0;
            if (!(streams & (1u << i))) continue;
            uint64_t *ls = (uint64_t *)malloc((nreads + 1) * 8);
            uint64_t at = 0;
            for (uint64_t q = 0; q < nreads; ++q) {
                ls[q] = at;
                at += gap ? gap[q] : 0;
                for (uint64_t j = qo[q]; j < qo[q + 1]; ++j)
                    at += dec_width(i == 0 ? v.len[j] : i == 1 ? v.ptr[j] : v.doc[j]) + 1;
                at += 1;
            }
            ls[nreads] = at;
            char *tx = (char *)malloc(at + 1);
            for (uint64_t q = 0; q < nreads; ++q) {
                char *p = tx + ls[q];
                if (gap) {
                    memset(p, '?', gap[q]); /* (the caller's: it drops ">id\n" here) */
                    p += gap[q];
                }
                for (uint64_t j = qo[q]; j < qo[q + 1]; ++j)
                    p += sprintf(p, "%llu ", (unsigned long long)(i == 0 ? v.len[j] : i == 1 ? v.ptr[j] : v.doc[j]));
                *p = '\n';
            }
            ix->tx[i] = tx;
            ix->ls[i] = ls;
            ix->tb[i] = out_bytes[i] = at;
        }
        ix->tn = nreads;
        ix->ready = 1;
    }
    free_vals(&v);
    free(dseq);
    free(doff);
    return rc;
}

int spx_query_text_reserve(spx_index *ix, int mode, int digest_kind, uint32_t k, uint64_t max_chars, uint64_t max_reads,
                           uint32_t streams, int with_class, const uint64_t text_bytes[3]) {
    (void)digest_kind; (void)k; (void)max_chars; (void)max_reads; (void)streams; (void)with_class; (void)text_bytes;
    if (!ix || (mode != SPX_MODE_PML && mode != SPX_MODE_MS)) {
        set_error("spx_query_text_reserve: index and a mode");
        return SPX_E_ARG;
    }
    return SPX_OK; /* (nothing to allocate ahead on the host) */
}

int spx_query_text_fetch(spx_index *ix, char *text[3], uint64_t *line_start[3]) {
    if (!ix || !text) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    if (!ix->ready) {
        set_error("spx_query_text_fetch without a successful spx_query_text_begin");
        return SPX_E_ARG;
    }
    for (int i = 0; i < 3; ++i) {
        if (!ix->tx[i]) continue;
        if (text[i]) memcpy(text[i], ix->tx[i], ix->tb[i]);
        if (line_start && line_start[i]) memcpy(line_start[i], ix->ls[i], (ix->tn + 1) * 8);
    }
    if (ix->cls_out && ix->cls_stash) memcpy(ix->cls_out, ix->cls_stash, ix->tn * sizeof(spx_class));
    free(ix->cls_stash);
    ix->cls_stash = NULL;
    ix->cls_out = NULL;
    free_text_state(ix);
    return SPX_OK;
}
