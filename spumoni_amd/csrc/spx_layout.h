// spx_layout.h -- flat HBM layout of the SPUMONI index (ours; gfx950 only).
//
// The reference walks Elias-Fano bitvectors and a Huffman wavelet tree
// (ri::rle_string, include/ms_rle_string.hpp; thr_bv, include/thresholds_ds.hpp)
// -- tens of dependent cache misses per searched character.  Here
// the state of a walk is (k, off): run index and offset inside the run
// (pos = S[k] + off), and every step is a handful of fixed-size gathers:
//
//  rows[k]  (16 B, by run index)  -- what a step needs at the run it lands in
//      len, H (head: bwt[pos], ms_rle_string.hpp:104), and the move-structure
//      pointer LFrun/LFoff = where LF(S[k]) lands (replaces rank() +
//      run_of_position() of LF, compute_ms_pml.cpp:180-187).  A match step is
//      k' = LFrun, off' = LFoff + off, then skip rows while off' >= len.
//
//  JumpRow  (32 B)  -- everything a threshold jump (compute_ms_pml.cpp:251-278)
//      needs for BOTH outcomes, for one run q with head c:
//        q        the run (successor c-run of the walk's position)
//        THR      thresholds[q] as (run, offset)   (thr_bv::operator[],
//                 thresholds_ds.hpp:478-491, zero-skipping of :421-423 applied)
//        sLF      where LF(start of q) lands        (pos >= THR: successor)
//        pLF      where LF(start of q) - 1 lands    (pos <  THR: predecessor; this
//                 is LF of the LAST character of the previous c-run, because LF
//                 images of consecutive same-letter runs are adjacent): one bit
//        Hs       head of the run the successor landing is in
//        j        position of q in the (letter, run index) order
//      dirrows[j]            one JumpRow per run, in (letter, run index) order
//      fat[fbase_c + blk_c(k)]  a 16-byte digest (FatRow) of the JumpRow of the first c-run at
//                            or after block blk_c(k) = umulhi(k, bmul_c) of letter c: one 16-byte
//                            gather answers most jumps outright; fat_js[slot >> FJ_SHIFT] is the
//                            directory position the slot's group of 8 starts at.  Every letter has its own block size
//                            2^32 / bmul_c (any real number >= 1), chosen at flatten time from the
//                            letter's share of the runs: dense tables for the letters whose runs
//                            are dense, coarse ones for the tail.
//      Q[j]                  run indices in (letter, run) order (4 B), to locate
//                            the successor when the block holds c-runs before k.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SPX_HD __host__ __device__ __forceinline__
#else
#define SPX_HD inline
#endif

// Names the flat layout AND the walk kernels that read it: a .spx cache written by another layout is
// refused, and measured HBM traffic (profiles/traffic.json) is only quoted for the version it was
// taken with.  Bump on any change to a record in this file or to the walk's access pattern.
#define SPX_LAYOUT_VERSION "spx-flat-r06a"

namespace spx {

constexpr uint64_t MASK40 = (1ull << 40) - 1;
constexpr uint32_t NO_LETTER = 0xffffffffu;
constexpr int ROW_PAD = 8;  // sentinel + padding rows after row r-1
constexpr int Q_PAD = 16;   // padding entries after (and 1 before) the directory

struct alignas(16) Row {
    uint64_t q0;  // len[40] | H[8] << 40 | LFrun[0:16] << 48
    uint64_t q1;  // LFoff[40] | LFrun[16:32] << 40 | thr_ok << 56 | room[7] << 57
};
constexpr uint32_t ROOM_SAT = 127;

// thr_ok: thresholds[k] <= S[k] (true for every consistent index): lets a byte >= 128
// that equals the head of its run (signed-char quirk) stay put
// without looking the threshold up.
// room: len[LFrun] - LFoff, saturated at ROOM_SAT -- how far a match step may move inside the
// destination run.  Knowing it BEFORE the gather lets a step whose offset exceeds it go
// straight to row LFrun + 1 instead of loading row LFrun only to skip it.
SPX_HD Row pack_row(uint32_t H, uint64_t len, uint32_t LFrun, uint64_t LFoff, bool thr_ok, uint64_t room) {
    Row r;
    const uint64_t rm = room < ROOM_SAT ? room : ROOM_SAT;
    r.q0 = (len & MASK40) | ((uint64_t)(H & 0xff) << 40) | ((uint64_t)(LFrun & 0xffff) << 48);
    r.q1 = (LFoff & MASK40) | ((uint64_t)(LFrun >> 16) << 40) | ((uint64_t)(thr_ok ? 1 : 0) << 56) | (rm << 57);
    return r;
}
SPX_HD uint32_t row_room(const Row& r) { return (uint32_t)(r.q1 >> 57); }
SPX_HD uint64_t row_len(const Row& r) { return r.q0 & MASK40; }
SPX_HD uint32_t row_H(const Row& r) { return (uint32_t)(r.q0 >> 40) & 0xff; }
SPX_HD uint32_t row_LFrun(const Row& r) {
    return (uint32_t)(r.q0 >> 48) | ((uint32_t)((r.q1 >> 40) & 0xffff) << 16);
}
SPX_HD uint64_t row_LFoff(const Row& r) { return r.q1 & MASK40; }
SPX_HD bool row_thr_ok(const Row& r) { return (r.q1 >> 56) & 1; }

// Compact encoding of the same 16 bytes, used for the whole index when every run is shorter
// than 2^16 (DevIndex::compact):
//     q0: len[16] | LFoff[16] << 16 | LFrun[32] << 32
//     q1: H[8] | thr_ok << 8 | cum0..cum3 [7 bits in a byte each] << 32
// cum_i = (len[LFrun] - LFoff) + len[LFrun+1] + ... + len[LFrun+i], saturated at 127: the
// offsets at which a match step leaves run LFrun, LFrun+1, ...  A step with offset `off` lands
// in run LFrun + t at offset off - cum_{t-1}, t = #{i : cum_i <= off} -- known BEFORE the
// gather, so the walk fetches the row it really lands in instead of walking the chain
// LFrun, LFrun+1, ... one gather at a time (0.27 gathers per character on the bench index).
constexpr uint32_t CUM_SAT = 127;
SPX_HD Row pack_row_compact(uint32_t H, uint32_t len, uint32_t LFrun, uint32_t LFoff, bool thr_ok,
                            const uint32_t cum[4]) {
    Row r;
    r.q0 = (uint64_t)(len & 0xffff) | ((uint64_t)(LFoff & 0xffff) << 16) | ((uint64_t)LFrun << 32);
    uint64_t c = 0;
    for (int i = 0; i < 4; ++i) c |= (uint64_t)(cum[i] < CUM_SAT ? cum[i] : CUM_SAT) << (8 * i);
    r.q1 = (uint64_t)(H & 0xff) | ((uint64_t)(thr_ok ? 1 : 0) << 8) | (c << 32);
    return r;
}
// A compact index stores its rows as Row32 (32 bytes, 32-byte aligned: one 128-byte line, two 16-byte lane
// loads): the compact row of run k -- with the heads of the first two runs a step from k can land in -- and,
// embedded in the same format, the row of the likeliest destination, run D = LFrun:
//     q0: len[16] | LFoff[16] << 16 | LFrun[32] << 32
//     q1: H[8] | thr_ok << 8 | cont << 9 | hd0[8] << 16 | hd1[8] << 24 | cums << 32      hd_i = head of run LFrun + i
//     e0, e1: q0, q1 of run D (its own hd0 / hd1 included)
// (hd = 0: past the last run; a head is never 0 after the terminator rewrite.)  A step from (k, off) whose
// destination is exactly (LFrun, LFoff + off) -- cum0 > off: half of the match steps on the statistical bench
// index, more on a real BWT -- goes on from the embedded row at once: two LF steps per gather.  And when the
// head of the run a step lands in is known and the next character differs from it, the next step is a jump,
// which needs only (run, offset): the landing row is not fetched at all (what the jump rows' Hs does for
// landings after a jump).  tools/layout_sim.py: row gathers per character 0.91 -> 0.50 on match-heavy reads,
// 0.46 -> 0.25 on the bench mix.
struct alignas(32) Row32 {
    uint64_t q0, q1, e0, e1;
};
SPX_HD uint32_t crow_len(const Row& r) { return (uint32_t)r.q0 & 0xffff; }
SPX_HD uint32_t crow_LFoff(const Row& r) { return ((uint32_t)r.q0 >> 16) & 0xffff; }
SPX_HD uint32_t crow_LFrun(const Row& r) { return (uint32_t)(r.q0 >> 32); }
SPX_HD uint32_t crow_H(const Row& r) { return (uint32_t)r.q1 & 0xff; }
SPX_HD bool crow_thr_ok(const Row& r) { return (r.q1 >> 8) & 1; }
SPX_HD uint32_t crow_cums(const Row& r) { return (uint32_t)(r.q1 >> 32); }
// a later piece of a run of 2^16 positions or more (spx_flatten.hip: such runs are laid out as pieces of the same head):
// not the first position of a run of the index -- the text rebuild's LF chains neither start nor end there
constexpr uint64_t CROW_CONT = 1ull << 9;
SPX_HD bool crow_cont(const Row& r) { return (r.q1 >> 9) & 1; }
SPX_HD uint32_t crow_dheads(uint64_t q1) { return (uint32_t)(q1 >> 16) & 0xffffu; }  // hd0 | hd1 << 8
SPX_HD uint64_t crow_with_dheads(uint64_t q1, uint32_t hd0, uint32_t hd1) {
    return (q1 & ~0xffff0000ull) | ((uint64_t)(hd0 & 0xff) << 16) | ((uint64_t)(hd1 & 0xff) << 24);
}

struct alignas(32) JumpRow {
    uint64_t d0;  // q[32] | THRrun[32] << 32
    uint64_t d1;  // THRoff[40] | sLFrun[0:24] << 40
    uint64_t d2;  // sLFoff[40] | sLFrun[24:32] << 40 | psame << 48 | Hs[8] << 49
    uint64_t d3;  // j[32]
};

// psame: the predecessor landing is in the same run as the successor landing (then it is
// (sLFrun, sLFoff-1)); otherwise it is the LAST position of run sLFrun-1, which the landing
// gather resolves (offset sentinel OFF_END).
// Hs: head of run sLFrun.  With it the walk knows the head of the run a jump lands in before
// touching that run's row: if the next character differs from it, the next step is another jump
// and the landing row is never fetched (a mismatch-heavy read then costs ONE gather per character
// instead of two).  (For a predecessor landing in run sLFrun-1 the fat digest carries that run's head, Hp; a full
// JumpRow does not, and the walk fetches the row.)
constexpr uint64_t OFF_END = ~0ull;
SPX_HD JumpRow pack_jumprow(uint32_t q, uint32_t THRrun, uint64_t THRoff, uint32_t sLFrun,
                            uint64_t sLFoff, bool psame, uint32_t Hs, uint32_t j) {
    JumpRow d;
    d.d0 = (uint64_t)q | ((uint64_t)THRrun << 32);
    d.d1 = (THRoff & MASK40) | ((uint64_t)(sLFrun & 0xffffff) << 40);
    d.d2 = (sLFoff & MASK40) | ((uint64_t)(sLFrun >> 24) << 40) | ((uint64_t)(psame ? 1 : 0) << 48) |
           ((uint64_t)(Hs & 0xff) << 49);
    d.d3 = (uint64_t)j;
    return d;
}
SPX_HD uint32_t jr_q(const JumpRow& d) { return (uint32_t)d.d0; }
SPX_HD uint32_t jr_THRrun(const JumpRow& d) { return (uint32_t)(d.d0 >> 32); }
SPX_HD uint64_t jr_THRoff(const JumpRow& d) { return d.d1 & MASK40; }
SPX_HD uint32_t jr_sLFrun(const JumpRow& d) {
    return (uint32_t)(d.d1 >> 40) | ((uint32_t)((d.d2 >> 40) & 0xff) << 24);
}
SPX_HD uint64_t jr_sLFoff(const JumpRow& d) { return d.d2 & MASK40; }
SPX_HD bool jr_psame(const JumpRow& d) { return (d.d2 >> 48) & 1; }
SPX_HD uint32_t jr_Hs(const JumpRow& d) { return (uint32_t)(d.d2 >> 49) & 0xff; }
SPX_HD uint32_t jr_j(const JumpRow& d) { return (uint32_t)d.d3; }

// What a fat slot holds: the JumpRow of the first c-run at or after its block, squeezed into ONE
// 16-byte lane load (the walk runs at the chip's rate of 16-byte lane loads) --
// the threshold run as a 20-bit distance below q, the two offsets in 16 bits, no j.  Where the predecessor landing is
// the last position of run sLFrun - 1 (psame = 0, which means sLFoff = 0) the sLFoff field holds Hp, the head of THAT
// run: a predecessor jump followed by another jump then needs no landing gather either (5 % of the C3 walk's gathers).
// `esc` marks a slot whose row does not fit; it holds its run's directory position instead (w0: q | j << 32; of w1 only
// the flag bits and FAT_SINGLE) and the walk reads the full JumpRow there.  When the slot's run lies before the walk's run
// (and the next slot is not the answer, FAT_SINGLE) the directory is scanned: from fat_js[slot >> FJ_SHIFT], the directory
// position of the first run of the letter at or after the block of the first slot of the slot's group of 8.  (Round 6: until
// then every slot had its directory position beside it, 4 of 20 bytes a slot -- a fifth of the table for a path 2-3 % of
// the jumps take; the same memory now holds 8.2 instead of 6.8 slots per run, and on a real BWT, where a letter's runs
// cluster, that is what the walk is short of: tools/fat_geom_sim.py, profiles/r06_fat_table.txt.)
// nosucc: the slot points past the letter's last run (no c-run at or after the block);
// first: the slot's run is the letter's first run (a predecessor jump from it is undefined).
struct alignas(16) FatRow {
    uint64_t w0;  // q[32] | sLFrun[32] << 32
    uint64_t w1;  // (q - THRrun)[19] | single << 19 | THRoff[16] << 20 | (psame ? sLFoff : Hp)[16] << 36 | Hs[8] << 52 |
                  // psame << 60 | nosucc << 61 | esc << 62 | first << 63
};
// single: the slot's run lies in the slot's own block and the letter's NEXT run lies in a later
// block.  A walk that finds the slot's run before its own run (the slot cannot answer) then knows
// that the successor is the run of the NEXT slot -- one more 16-byte load, next to the first one,
// instead of fat_js -> Q -> dirrows.
constexpr uint64_t FAT_SINGLE = 1ull << 19;
constexpr int FJ_SHIFT = 3;  // one directory position per 2^FJ_SHIFT slots; every letter's first slot is a multiple of that
constexpr uint64_t FJ_GROUP = 1ull << FJ_SHIFT;
SPX_HD uint64_t fatjs_count(uint64_t nfat) { return (nfat >> FJ_SHIFT) + 2; }
SPX_HD FatRow pack_fatrow(const JumpRow& f, bool nosucc, bool first, bool force_esc, bool single, uint32_t Hp, uint32_t j) {
    const uint32_t q = (uint32_t)f.d0, trun = (uint32_t)(f.d0 >> 32);
    const uint64_t toff = f.d1 & MASK40, soff = f.d2 & MASK40;
    const uint32_t srun = (uint32_t)(f.d1 >> 40) | ((uint32_t)((f.d2 >> 40) & 0xff) << 24);
    const uint64_t dthr = (nosucc || trun > q) ? 0 : (uint64_t)q - trun;
    const bool esc = force_esc || soff >= (1u << 16) ||
                     (!nosucc && (trun > q || dthr >= (1u << 19) || toff >= (1u << 16)));
    FatRow h;
    if (esc) {  // the row does not fit: the slot names its directory position
        h.w0 = (uint64_t)q | ((uint64_t)j << 32);
        h.w1 = (single ? FAT_SINGLE : 0) | ((uint64_t)(nosucc ? 1 : 0) << 61) | (1ull << 62) | ((uint64_t)(first ? 1 : 0) << 63);
        return h;
    }
    h.w0 = (uint64_t)q | ((uint64_t)srun << 32);
    const bool psame = (f.d2 >> 48) & 1;  // (= soff > 0)
    h.w1 = (dthr & 0x7ffff) | (single ? FAT_SINGLE : 0) | ((toff & 0xffff) << 20) | (((psame ? soff : (uint64_t)(Hp & 0xff)) & 0xffff) << 36) |
           (((f.d2 >> 49) & 0xff) << 52) |
           (((f.d2 >> 48) & 1) << 60) | ((uint64_t)(nosucc ? 1 : 0) << 61) | ((uint64_t)(esc ? 1 : 0) << 62) |
           ((uint64_t)(first ? 1 : 0) << 63);
    return h;
}

// per byte value c: everything the walk needs that depends only on the letter
struct alignas(16) LetterInfo {
    uint32_t qbeg;   // directory range of the letter: Q[qbeg, qend); empty = number_of_letter(c) == 0
    uint32_t qend;
    uint32_t frun;   // run containing F[c] (r if F[c] == n): landing after an absent letter
    uint32_t bmul;   // fat block of run k = umulhi(k, bmul): blocks of 2^32 / bmul runs
    uint64_t foff;   // F[c] - S[frun]
    uint64_t fbase;  // first fat slot of the letter
};
SPX_HD uint32_t fat_block(uint32_t k, uint32_t bmul) { return (uint32_t)(((uint64_t)k * bmul) >> 32); }
// slots of a letter: a block for every run index up to r (the sentinel position), one more for the next-slot shortcut
// (FAT_SINGLE), rounded up to whole groups of 8 (FJ_SHIFT below); all slots past the letter's last run say `nosucc`
SPX_HD uint64_t letter_slots(uint32_t r, uint32_t bmul) { return ((uint64_t)fat_block(r, bmul) + 2 + 7) & ~7ull; }

// Side data of directory position j, everything a jump to run Q[j] (or to the end of run Q[j-1])
// hands out in MS / doc mode, in 16 bytes:
//     a0: samples_start[Q[j]] [40] | samples_last[Q[j-1]] [0:24] << 40
//     a1: samples_last[Q[j-1]] [24:40] | (start_runs_doc[Q[j]] | end_runs_doc[Q[j-1]] << 16) << 16
// (fields of an index without samples / without a document array are zero)
struct alignas(16) Aux {
    uint64_t a0, a1;
};
SPX_HD Aux pack_aux(uint64_t ss, uint64_t se, uint32_t docs) {
    Aux a;
    a.a0 = (ss & MASK40) | ((se & 0xffffff) << 40);
    a.a1 = ((se >> 24) & 0xffff) | ((uint64_t)docs << 16);
    return a;
}
SPX_HD uint64_t aux_ss(const Aux& a) { return a.a0 & MASK40; }
SPX_HD uint64_t aux_se(const Aux& a) { return (a.a0 >> 40) | ((a.a1 & 0xffff) << 24); }
SPX_HD uint32_t aux_docs(const Aux& a) { return (uint32_t)(a.a1 >> 16); }

struct SamplePair {  // flatten-time temporary: entry j = {samples_start[Q[j]], samples_last[Q[j-1]]}
    uint64_t ss;
    uint64_t se;
};

// kernel-visible view of an index (all pointers are device memory)
struct DevIndex {
    const Row* rows;            // r + ROW_PAD rows (Row32 when compact: stride 32 bytes); row r is the "pos == n" sentinel
    const JumpRow* dirrows;     // r + 1 (+ pad) jump rows, (letter, run) order
    const char* fat;            // slots of the first c-run at or after a block, letter by letter (see fat_stride)
    const uint32_t* Q;          // directory; Q[-1] and Q[r .. r + Q_PAD) are readable
    const Aux* aux;             // r + 1 (+ pad) entries in directory order, or nullptr (PML-only, no docs)
    const uint64_t* ss_by_run;  // samples_start by run index (+ pad) or nullptr
    const uint32_t* rundocs;    // by run index k: docS[k] | docE[k] << 16, or nullptr
    // a fat slot is fat_stride bytes: the FatRow, then (index with SA samples or documents) the Aux
    // of that directory position at +16 -- everything a jump needs in MS / doc mode sits in the
    // same 32 aligned bytes
    const uint32_t* fat_js; // per group of 8 slots: directory position of the first run of the letter at or after the group's first block
    uint32_t fat_stride;   // 16 or 32
    const LetterInfo* letters;  // 256 entries
    const uint8_t* text;        // MS extension text or nullptr
    uint64_t n_text;
    uint64_t n;
    uint32_t r;         // runs of the flat layout (pieces of long runs count: >= the file's r, spx_index::r)
    uint32_t compact;   // rows use the compact encoding (every run / piece shorter than 2^16)
    uint32_t nletters;  // byte values that occur in the BWT
    uint64_t nfat;      // fat slots in all (every letter: fat_block(r, bmul) + 2, rounded up to a whole group of 8)
    uint32_t init_k;    // run of position n-1  (= r-1)
    uint64_t init_off;  // (n-1) - S[r-1]
    Row32 init_row;     // rows[r-1]: every read starts on it, so the walk never gathers it (general rows: q0, q1)
    uint64_t init_sample;  // get_last_run_sample(): (samples_last[r-1] + 1) % n
    uint32_t init_doc;     // end_runs_doc[r-1]         (compute_ms_pml.cpp:298)
    uint32_t doc_at0;      // start_runs_doc[run_of_position(0)]   (:641-642)
};

// row k of an index in either encoding (the first 16 bytes of a Row32 are the compact Row)
SPX_HD const Row& row_at(const DevIndex& ix, uint64_t k) {
    return *reinterpret_cast<const Row*>(reinterpret_cast<const char*>(ix.rows) + k * (ix.compact ? sizeof(Row32) : sizeof(Row)));
}

// ---- pieces of a run (spx_flatten.hip) -----------------------------------------------------------------------------
// A run is laid out as consecutive PIECES of the same head when (a) it holds 2^16 positions or more (the compact row's
// 16-bit fields), or (b) its LF image covers many runs: a step out of offset `off` of run k lands LF(S[k]) + off, in run
// LFrun + t, and the compact row answers t < 4 outright; past that the walk moves on one row -- one dependent gather --
// at a time.  That is 0.02 gathers per character on the bench index and unbounded in principle (the move structure's
// known worst case: a long run whose image covers thousands of short ones; tools/ff_model.py: 141 gathers per step on
// Pareto-distributed run lengths).  So a piece also ends where its image has covered `span` runs (balancing in the
// sense of Nishimoto & Tabei; spx_flatten.hip repeats the pass on its own output): at most span - 4 rows are walked on
// from a landing.
constexpr uint64_t PIECE_MAX = 65535;

// run of position p: the largest k in [0, r) with S[k] <= p  (S[0] = 0, ascending, S[r] = n)
SPX_HD uint64_t run_of_position(const uint64_t* S, uint64_t r, uint64_t p) {
    uint64_t lo = 0, hi = r;  // S[lo] <= p < S[hi]
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (S[mid] <= p)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// f(offset, length) for every piece of a run of `len` positions, in order; returns how many.
// S == nullptr or span == 0: pieces of PIECE_MAX positions only.  Otherwise lf = LF(start of the run), a = the run that
// holds position lf, nb = how many run starts S[a + 1 .. a + nb] lie inside the image (lf, lf + len).
template <class F>
SPX_HD uint32_t for_each_piece(uint64_t len, uint64_t lf, const uint64_t* S, uint64_t a, uint64_t nb, uint32_t span, F&& f) {
    uint32_t np = 0;
    uint64_t pos = 0, i = 1;  // S[a + i] - lf: the first run start inside the image that lies after pos
    const bool balance = S != nullptr && span > 0;
    do {
        uint64_t next = len - pos > PIECE_MAX ? pos + PIECE_MAX : len;
        if (balance && i + span - 1 <= nb) {
            // [pos, cand) covers the runs a + i - 1 .. a + i + span - 2: `span` of them
            // (cand > pos on a run list whose lengths add up without overflow; the test keeps the loop finite on one
            // that does not -- flatten_core refuses it afterwards)
            const uint64_t cand = S[a + i + span - 1] - lf;
            if (cand > pos && cand < next) next = cand;
        }
        f(pos, next - pos);
        ++np;
        pos = next;
        if (balance)
            while (i <= nb && S[a + i] - lf <= pos) ++i;  // (at most `span` steps: the piece covers no more runs)
    } while (pos < len);
    return np;
}

}  // namespace spx
