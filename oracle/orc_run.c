/*
 * orc_run.c -- CPU ORACLE of the `spumoni run` harness (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of classify_reads_pml / classify_reads_ms
 * (/root/reference/src/compute_ms_pml.cpp:845-1217), BatchLoader
 * (src/batch_loader.cpp:26-131) and the output writers, on top of the query
 * oracle in spumoni_oracle.c.  Used to check the byte format of the files the
 * HIP-backed `spumoni run` writes.  PARITY UNPINNED (see spumoni_oracle.h).
 *
 * usage: orc_run <ref_file incl. .fa/.bin> <reads file> <P|M> <doc 0|1> <report 0|1>
 *                <bin_width> <flags: n|m|a|g> [text file for MS] [--dump-reads]
 *        g: general text (classify_general_reads_pml / _ms, :1219-1297): the reads file is raw bytes, every read
 *           ends in \x01, reads are named read_<k>; no upper-casing, no documents, no report
 * Index input: <ref_file>.bwt.heads/.bwt.len/.thr_pos[/.ssa/.esa], <ref_file>.doc,
 * <ref_file>.pmlnulldb/.msnulldb.   Output: <reads>.pseudo_lengths etc.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spumoni_oracle.h"

/* ---------- tiny growable string ---------- */
typedef struct {
    char *p;
    size_t n, cap;
} str;
static void s_clear(str *s) { s->n = 0; }
static void s_push(str *s, const char *d, size_t k) {
    if (s->n + k + 1 > s->cap) {
        s->cap = (s->n + k + 1) * 2;
        s->p = (char *)realloc(s->p, s->cap);
    }
    memcpy(s->p + s->n, d, k);
    s->n += k;
    s->p[s->n] = 0;
}

/* ---------- the input stream, with iostream-like state ---------- */
typedef struct {
    char *d;
    size_t n, pos;
    int eofbit, failbit;
} in_t;
static int in_good(const in_t *in) { return !in->eofbit && !in->failbit; }
static int in_peek(in_t *in) { /* std::istream::peek */
    if (!in_good(in)) {
        in->failbit = 1;
        return EOF;
    }
    if (in->pos >= in->n) {
        in->eofbit = 1;
        return EOF;
    }
    return (unsigned char)in->d[in->pos];
}
static int in_getline(in_t *in, str *out) { /* std::getline(stream, string) */
    s_clear(out);
    if (!in_good(in)) {
        in->failbit = 1;
        return 0;
    }
    size_t got = 0;
    for (;;) {
        if (in->pos >= in->n) {
            in->eofbit = 1;
            if (got == 0) in->failbit = 1;
            break;
        }
        char c = in->d[in->pos++];
        got++;
        if (c == '\n') break;
        s_push(out, &c, 1);
    }
    return !in->failbit;
}

/* ---------- BatchLoader (src/batch_loader.cpp) ---------- */
enum { NOT_CLEAR, FA, FQ };
typedef struct {
    int input_format;
    in_t batch_stream;
    str batch_text;
    str batch_buffer;
} loader_t;

static void fatal_error(const char *msg) {
    fprintf(stderr, "\n\033[31mError: \033[0m%s\n\n", msg);
    exit(1);
}

static int loadBatch(loader_t *L, in_t *input, size_t num_bases) { /* :26-76 */
    if (L->input_format == NOT_CLEAR) {
        if (!in_good(input)) return 0;
        switch (in_peek(input)) {
            case '>': L->input_format = FA; break;
            case '@': L->input_format = FQ; break;
            case EOF: return 0;
            default: fatal_error("unrecognized input query file type - expects FASTA or FASTQ.");
        }
    }
    s_clear(&L->batch_text);
    size_t num_bases_covered = 0, lines_covered = 0, record_size = 0;
    int valid_batch = 0;
    while (in_good(input) && num_bases_covered < num_bases) {
        if (!in_getline(input, &L->batch_buffer)) return 0;
        lines_covered++;
        record_size += L->batch_buffer.n;
        valid_batch = 1;
        if (L->input_format == FQ) {
            if (lines_covered % 4 == 0) {
                num_bases_covered += record_size / 2;
                record_size = 0;
            }
        } else {
            if (in_peek(input) == '>') {
                num_bases_covered += record_size;
                record_size = 0;
            }
        }
        s_push(&L->batch_text, L->batch_buffer.p ? L->batch_buffer.p : "", L->batch_buffer.n);
        s_push(&L->batch_text, "\n", 1);
    }
    L->batch_stream.d = L->batch_text.p;
    L->batch_stream.n = L->batch_text.n;
    L->batch_stream.pos = 0;
    L->batch_stream.eofbit = L->batch_stream.failbit = 0;
    return valid_batch;
}

static void strip_ws(str *s) {
    while (s->n > 0 && isspace((unsigned char)s->p[s->n - 1])) s->p[--s->n] = 0;
}

/* :78-131; returns 0 when no read could be grabbed */
static int grabNextRead(loader_t *L, str *id, str *seq) {
    str *bb = &L->batch_buffer;
    in_t *bs = &L->batch_stream;
    if (!in_getline(bs, bb)) return 0;
    if (!bb->n) return 0; /* an empty line */
    if (L->input_format == FQ) {
        if (bb->p[0] != '@') fatal_error("Incorrect FASTQ entry, it should start with '@'");
    } else if (bb->p[0] != '>') {
        fatal_error("Incorrect FASTA entry, it should start with '>'");
    }
    if (bb->n <= 2) fatal_error("header line is missing an id. invalid query cannot be processed.");
    size_t id_length = bb->n; /* find_first_of(" \t\r", 1) */
    for (size_t i = 1; i < bb->n; ++i)
        if (bb->p[i] == ' ' || bb->p[i] == '\t' || bb->p[i] == '\r') {
            id_length = i;
            break;
        }
    s_clear(id); /* substr(1, id_length): id_length characters starting at 1 (clamped) */
    size_t take = id_length;
    if (1 + take > bb->n) take = bb->n - 1;
    s_push(id, bb->p + 1, take);
    s_clear(seq);
    if (L->input_format == FQ) {
        if (!in_getline(bs, bb)) return 0;
        strip_ws(bb);
        s_push(seq, bb->p ? bb->p : "", bb->n);
        if (!in_getline(bs, bb)) return 0;
        if (!in_getline(bs, bb)) return 0;
    } else {
        s_push(seq, "", 0);
        while (in_good(bs) && in_peek(bs) != '>') {
            if (!in_getline(bs, bb)) return seq->n != 0;
            strip_ws(bb);
            s_push(seq, bb->p ? bb->p : "", bb->n);
        }
    }
    return 1;
}

/* ---------- index + side files ---------- */
static char *read_file(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *p = (char *)malloc((size_t)sz + 1);
    if (sz && fread(p, 1, (size_t)sz, f) != (size_t)sz) {
        fclose(f);
        free(p);
        return NULL;
    }
    fclose(f);
    p[sz] = 0;
    *n = (size_t)sz;
    return p;
}

static uint64_t *read_int_vector(const unsigned char **pp, uint64_t *count) { /* sdsl::int_vector<> */
    const unsigned char *p = *pp;
    uint64_t bits;
    memcpy(&bits, p, 8);
    unsigned width = p[8];
    p += 9;
    uint64_t words = (bits + 63) / 64;
    uint64_t cnt = width ? bits / width : 0;
    uint64_t *out = (uint64_t *)calloc(cnt ? cnt : 1, 8);
    for (uint64_t i = 0; i < cnt; ++i) {
        uint64_t v = 0;
        for (unsigned b = 0; b < width; ++b) {
            uint64_t bit = i * width + b;
            uint64_t w;
            memcpy(&w, p + (bit >> 6) * 8, 8);
            v |= ((w >> (bit & 63)) & 1ull) << b;
        }
        out[i] = v;
    }
    *pp = p + words * 8;
    *count = cnt;
    return out;
}

int main(int argc, char **argv) {
    if (argc < 8) {
        fprintf(stderr, "usage: see header of orc_run.c\n");
        return 2;
    }
    const char *ref = argv[1], *reads = argv[2];
    const int is_ms = argv[3][0] == 'M';
    const int use_doc = atoi(argv[4]), write_report = atoi(argv[5]);
    const size_t bin_width = (size_t)atol(argv[6]);
    const int use_promotions = argv[7][0] == 'm', use_dna_letters = argv[7][0] == 'a';
    const char *text_path = (argc > 8 && argv[8][0] != '-') ? argv[8] : NULL;
    int dump_reads = 0;
    unsigned dig_k = 4, dig_w = 11; /* include/spumoni_main.hpp:247-248 */
    for (int i = 8; i < argc; ++i) {
        if (!strcmp(argv[i], "--dump-reads")) dump_reads = 1;
        if (!strcmp(argv[i], "--k") && i + 1 < argc) dig_k = (unsigned)atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--w") && i + 1 < argc) dig_w = (unsigned)atoi(argv[i + 1]);
    }
    char path[4096];

    size_t fsz = 0;
    in_t input;
    memset(&input, 0, sizeof input);
    input.d = read_file(reads, &fsz);
    if (!input.d) fatal_error("cannot read the pattern file");
    input.n = fsz;
    loader_t L;
    memset(&L, 0, sizeof L);

    if (dump_reads) {
        size_t nb = 0;
        str id = {0}, seq = {0};
        while (loadBatch(&L, &input, 1000)) {
            printf("#batch %zu\n", nb++);
            while (grabNextRead(&L, &id, &seq)) printf("%s\t%s\n", id.p ? id.p : "", seq.p ? seq.p : "");
        }
        return 0;
    }

    orc_index *ix = orc_load_raw(ref, is_ms);
    if (!ix) fatal_error("cannot load the raw index files");
    if (use_doc) {
        snprintf(path, sizeof path, "%s.doc", ref);
        size_t dsz;
        unsigned char *d = (unsigned char *)read_file(path, &dsz);
        if (!d) fatal_error("cannot read the document array");
        const unsigned char *p = d + 8;
        uint64_t c1, c2;
        ix->start_runs_doc = read_int_vector(&p, &c1);
        ix->end_runs_doc = read_int_vector(&p, &c2);
    }
    unsigned char *text = NULL;
    size_t n_text = 0;
    if (is_ms) {
        if (!text_path) fatal_error("MS needs the text file");
        text = (unsigned char *)read_file(text_path, &n_text);
        if (!text) fatal_error("cannot read the text file");
    }
    double percentile = 0.0;
    snprintf(path, sizeof path, "%s.%s", ref, is_ms ? "msnulldb" : "pmlnulldb");
    {
        size_t nsz;
        char *nd = read_file(path, &nsz);
        if (nd && nsz >= 32) memcpy(&percentile, nd + 24, 8);
    }
    size_t max_value_thr = orc_max_value_thr(percentile, !is_ms, use_promotions, use_dna_letters);

    snprintf(path, sizeof path, "%s.%s", reads, is_ms ? "lengths" : "pseudo_lengths");
    FILE *lengths_file = fopen(path, "w");
    FILE *pointers_file = NULL, *doc_file = NULL, *report_file = NULL;
    if (is_ms) {
        snprintf(path, sizeof path, "%s.pointers", reads);
        pointers_file = fopen(path, "w");
    }
    if (use_doc) {
        snprintf(path, sizeof path, "%s.doc_numbers", reads);
        doc_file = fopen(path, "w");
    }
    if (write_report) { /* :877-886 */
        snprintf(path, sizeof path, "%s.report", reads);
        report_file = fopen(path, "w");
        fprintf(report_file, "%-30s%-15s%-19s%-2zu%-5s%-12s%-12s\n", "read id:", "status:", "avg max-value (thr=",
                max_value_thr, "):", "above thr:", "below thr:");
    }

    if (argv[7][0] == 'g') { /* :1219-1297 */
        uint64_t *gl = NULL, *gp = NULL;
        size_t gcap = 0, start = 0, nr = 0;
        for (size_t i = 0; i < fsz; ++i) {
            if (input.d[i] != '\x01') continue; /* (what follows the last separator is never a read, :1236-1256) */
            const char *rd = input.d + start;
            const size_t m = i - start;
            if (m + 1 > gcap) {
                gcap = 2 * (m + 1);
                gl = (uint64_t *)realloc(gl, gcap * 8);
                gp = (uint64_t *)realloc(gp, gcap * 8);
            }
            if (m) {
                if (!is_ms) {
                    orc_pml_query(ix, rd, m, gl);
                } else {
                    orc_ms_query(ix, rd, m, gp);
                    orc_ms_lengths(rd, m, gp, text, n_text, gl);
                }
            }
            fprintf(lengths_file, ">read_%zu\n", nr);
            for (size_t j = 0; j < m; ++j) fprintf(lengths_file, "%llu ", (unsigned long long)gl[j]);
            fputc('\n', lengths_file);
            if (is_ms) {
                fprintf(pointers_file, ">read_%zu\n", nr);
                for (size_t j = 0; j < m; ++j) fprintf(pointers_file, "%llu ", (unsigned long long)gp[j]);
                fputc('\n', pointers_file);
            }
            start = i + 1;
            nr++;
        }
        fclose(lengths_file);
        if (pointers_file) fclose(pointers_file);
        fprintf(stderr, "orc_run: %zu reads\n", nr);
        return 0;
    }

    str id = {0}, seq = {0}, seq_view = {0};
    char *dig = NULL;
    size_t dig_cap = 0;
    uint64_t *lengths = NULL, *pointers = NULL, *docs = NULL;
    size_t cap = 0, num_reads = 0;
    while (loadBatch(&L, &input, 1000)) { /* :903 */
        while (grabNextRead(&L, &id, &seq)) {
            for (size_t i = 0; i < seq.n; ++i) seq.p[i] = (char)toupper((unsigned char)seq.p[i]); /* :917 */
            if (use_promotions || use_dna_letters) { /* :920-923 */
                const size_t dcap = (size_t)dig_k * seq.n + 1;
                if (dcap > dig_cap) {
                    dig_cap = 2 * dcap;
                    dig = (char *)realloc(dig, dig_cap);
                }
                const size_t dn = orc_digest(use_promotions ? ORC_DIGEST_PROMOTED : ORC_DIGEST_DNA, dig_k, dig_w, NULL,
                                             (const uint8_t *)seq.p, seq.n, (uint8_t *)dig, dcap);
                dig[dn] = 0;
                seq_view.p = dig;
                seq_view.n = dn;
            } else {
                seq_view = seq;
            }
            const str rd = seq_view; /* the read as searched */
            if (rd.n == 0) { /* :926-931 */
                printf("\n\n");
                fprintf(stderr, "Warning: %s was empty after digestion, commonly due to reads "
                                "consisting of mostly non-ACGT characters. Please remove "
                                "read or run SPUMONI without minimizer digestion.\n\n", id.p);
                exit(1);
            }
            const size_t m = rd.n;
            if (m > cap) {
                cap = 2 * m;
                lengths = (uint64_t *)realloc(lengths, cap * 8);
                pointers = (uint64_t *)realloc(pointers, cap * 8);
                docs = (uint64_t *)realloc(docs, cap * 8);
            }
            if (!is_ms) {
                if (use_doc)
                    orc_pml_query_doc(ix, rd.p, m, lengths, docs);
                else
                    orc_pml_query(ix, rd.p, m, lengths);
            } else {
                if (use_doc)
                    orc_ms_query_doc(ix, rd.p, m, pointers, docs);
                else
                    orc_ms_query(ix, rd.p, m, pointers);
                orc_ms_lengths(rd.p, m, pointers, text, n_text, lengths);
            }
            uint64_t above = 0, below = 0, sum = 0;
            int found = 0;
            if (write_report) found = orc_classify(lengths, m, bin_width, max_value_thr, &above, &below, &sum);
            num_reads++;
            if (use_doc) { /* :1003-1007 */
                fprintf(doc_file, ">%s\n", id.p);
                for (size_t i = 0; i < m; ++i) fprintf(doc_file, "%llu ", (unsigned long long)docs[i]);
                fputc('\n', doc_file);
            }
            fprintf(lengths_file, ">%s\n", id.p);
            if (is_ms) fprintf(pointers_file, ">%s\n", id.p);
            for (size_t i = 0; i < m; ++i) fprintf(lengths_file, "%llu ", (unsigned long long)lengths[i]);
            if (is_ms)
                for (size_t i = 0; i < m; ++i) fprintf(pointers_file, "%llu ", (unsigned long long)pointers[i]);
            fputc('\n', lengths_file);
            if (is_ms) fputc('\n', pointers_file);
            if (write_report) /* :1012-1020; iostream precision(3) general format == %.3g */
                fprintf(report_file, "%-30s%-15s%-26.3g%-12llu%-12llu\n", id.p, found ? "FOUND" : "NOT_PRESENT",
                        (sum + 0.0) / (double)(above + below), (unsigned long long)above,
                        (unsigned long long)below);
        }
    }
    fclose(lengths_file);
    if (pointers_file) fclose(pointers_file);
    if (doc_file) fclose(doc_file);
    if (report_file) fclose(report_file);
    fprintf(stderr, "orc_run: %zu reads\n", num_reads);
    return 0;
}
