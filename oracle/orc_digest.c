/*
 * orc_digest.c -- CPU ORACLE for the minimizer digestion pre-step of `spumoni run -m / -a`
 * (SURVEY section 8 rows a17 / f2).  TEST INFRASTRUCTURE, NOT PRODUCT CODE (see
 * spumoni_oracle.h).
 *
 * PARITY UNPINNED.  The first-party part is small and restated exactly:
 *   perform_minimizer_digestion      src/spumoni.cpp:294-319   (-m, "promoted" alphabet)
 *   perform_dna_minimizer_digestion  src/spumoni.cpp:321-342   (-a, DNA-letter minimizers)
 * i.e. consecutive-duplicate suppression against the last value PUSHED (a uint8_t vector in
 * both functions, :300/:329), the `x > 2 ? x : x + 3` remap (:311) and the spelling of the
 * k-mer back into letters (:336).  The minimizer streams themselves come from
 * dnbaker/bonsai @ 5273b81a92 (thirdparty/CMakeLists.txt:61-71), whose source is NOT in this
 * container.  What follows restates its published algorithm; every point that could not be
 * checked against the source is listed so that a maintainer with the source can pin it:
 *
 *  [B1] bns::Encoder<score::Lex>::for_each, canonicalize = false, unspaced Spacer(k, w):
 *       2-bit encoding A=0 C=1 G=2 T=3, first base in the most significant bits; a character
 *       outside ACGT restarts the k-mer fill and nothing else (the window queue keeps its
 *       contents); the queue is reset per sequence.
 *  [B2] score::Lex(kmer) = kmer ^ XOR_MASK with XOR_MASK = 0xe37e28c4271b5a2d (the constant
 *       bonsai inherits from Kraken); the minimizer of a window is the k-mer of least score.
 *  [B3] the window holds  wsz = w - k + 1  consecutive k-mers (w bases), at least 1; the
 *       first value is reported when the queue holds wsz k-mers and one is reported per
 *       k-mer from then on (QueueMap::next_value: push, report the minimum, drop the oldest).
 *  [B4] bns::RollingHasher<uint8_t>(k, false, DNA, w)::for_each_uncanon: the same windowing
 *       over the 8-bit cyclic-polynomial hash of each k-mer (Lemire's CyclicHash with word
 *       size 8: h = rotl8(h, 1) ^ T[c] per character, so h(k-mer) = XOR_j rotl8(T[c_j],
 *       k-1-j)); score = value = the hash; non-ACGT characters restart the fill as in [B1].
 *  [B5] the character table T: CharacterHash draws T[0..255] from a Mersenne twister
 *       seeded with RollingHasher's default seed1 = 1337, masked to 8 bits.  Only T['A'],
 *       T['C'], T['G'], T['T'] matter.  orc_digest_default_charhash() reproduces that
 *       reading (MT19937, init_genrand(1337), output & 0xff); callers can pass the four
 *       constants explicitly instead, which is how a maintainer pins them.
 */
#include "spumoni_oracle.h"

#include <stdlib.h>
#include <string.h>

#define ORC_XOR_MASK 0xe37e28c4271b5a2dULL

/* MT19937 (Matsumoto & Nishimura 1998), init_genrand seeding */
static void mt_outputs(uint32_t seed, uint32_t *out, int count) {
    static uint32_t mt[624];
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    /* one full regeneration is enough for count <= 624 */
    for (int i = 0; i < 624; ++i) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (int i = 0; i < count && i < 624; ++i) {
        uint32_t y = mt[i];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        out[i] = y;
    }
}

void orc_digest_default_charhash(uint8_t out[4]) { /* [B5] */
    uint32_t o[256];
    mt_outputs(1337u, o, 256);
    out[0] = (uint8_t)(o['A'] & 0xff);
    out[1] = (uint8_t)(o['C'] & 0xff);
    out[2] = (uint8_t)(o['G'] & 0xff);
    out[3] = (uint8_t)(o['T'] & 0xff);
}

static int base_code(uint8_t c) {
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
    }
}

static uint8_t rotl8(uint8_t x, unsigned s) {
    s &= 7;
    return (uint8_t)((x << s) | (x >> ((8 - s) & 7)));
}

/* kind: 1 = promoted (-m), 2 = DNA letters (-a).  charhash: T[A],T[C],T[G],T[T] or NULL for the
 * default.  Returns the digested length; writes at most cap bytes (the length is still exact). */
size_t orc_digest(int kind, unsigned k, unsigned w, const uint8_t *charhash, const uint8_t *seq, size_t len,
                  uint8_t *out, size_t cap) {
    uint8_t T[4];
    if (charhash)
        memcpy(T, charhash, 4);
    else
        orc_digest_default_charhash(T);
    const size_t wsz = w > k ? (size_t)w - k + 1 : 1; /* [B3] */
    /* the queue: the last wsz (element, score) pairs, oldest first */
    uint64_t *q_el = (uint64_t *)malloc(wsz * sizeof(uint64_t));
    uint64_t *q_sc = (uint64_t *)malloc(wsz * sizeof(uint64_t));
    size_t q_n = 0;
    size_t n_out = 0;
    int have_last = 0;
    uint8_t last_pushed = 0; /* mseq_vec.back(), a uint8_t (:300, :329) */
    unsigned filled = 0;
    for (size_t i = 0; i < len; ++i) {
        if (base_code(seq[i]) < 0) { /* [B1]/[B4] */
            filled = 0;
            continue;
        }
        if (filled < k) filled++;
        if (filled < k) continue;
        /* the k-mer ending at i */
        uint64_t el, score;
        if (kind == 2) {
            uint64_t km = 0;
            for (unsigned j = 0; j < k; ++j) km = (km << 2) | (uint64_t)base_code(seq[i - k + 1 + j]);
            el = km;
            score = km ^ ORC_XOR_MASK; /* [B2] */
        } else {
            uint8_t h = 0;
            for (unsigned j = 0; j < k; ++j) h = (uint8_t)(rotl8(h, 1) ^ T[base_code(seq[i - k + 1 + j])]);
            el = h;
            score = h;
        }
        /* QueueMap::next_value [B3]: push; if full report the least (score, element) and pop */
        q_el[q_n] = el;
        q_sc[q_n] = score;
        q_n++;
        if (q_n < wsz) continue;
        size_t best = 0;
        for (size_t j = 1; j < q_n; ++j)
            if (q_sc[j] < q_sc[best] || (q_sc[j] == q_sc[best] && q_el[j] < q_el[best])) best = j;
        const uint64_t x = q_el[best];
        memmove(q_el, q_el + 1, (q_n - 1) * sizeof(uint64_t));
        memmove(q_sc, q_sc + 1, (q_n - 1) * sizeof(uint64_t));
        q_n--;
        /* the caller's lambda (src/spumoni.cpp:305-313 / :332-337), hp_compress = true */
        if (kind == 2) {
            if (!have_last || (uint64_t)last_pushed != x) {
                last_pushed = (uint8_t)x;
                have_last = 1;
                for (unsigned j = 0; j < k; ++j) { /* sp.to_string(x) */
                    const char letter = "ACGT"[(x >> (2 * (k - 1 - j))) & 3];
                    if (n_out < cap) out[n_out] = (uint8_t)letter;
                    n_out++;
                }
            }
        } else {
            const uint8_t xv = (uint8_t)x;
            if (!have_last || last_pushed != xv) {
                last_pushed = xv;
                have_last = 1;
                const uint8_t y = (xv > 2) ? xv : (uint8_t)(xv + 3); /* :311 */
                if (n_out < cap) out[n_out] = y;
                n_out++;
            }
        }
    }
    free(q_el);
    free(q_sc);
    return n_out;
}

/* batch form: out_offs gets nreads+1 offsets; out may be NULL to size only */
void orc_digest_batch(int kind, unsigned k, unsigned w, const uint8_t *charhash, const uint8_t *seqs,
                      const uint64_t *offs, uint64_t nreads, uint8_t *out, uint64_t cap, uint64_t *out_offs) {
    uint64_t o = 0;
    out_offs[0] = 0;
    for (uint64_t q = 0; q < nreads; ++q) {
        const size_t room = (out && cap > o) ? (size_t)(cap - o) : 0;
        o += orc_digest(kind, k, w, charhash, seqs + offs[q], (size_t)(offs[q + 1] - offs[q]), out ? out + o : NULL,
                        room);
        out_offs[q + 1] = o;
    }
}
