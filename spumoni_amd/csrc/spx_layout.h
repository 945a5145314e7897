// spx_layout.h -- flat HBM layout of the SPUMONI index (ours; gfx950 only).
//
// The reference walks Elias-Fano bitvectors and a Huffman wavelet tree
// (ri::rle_string, include/ms_rle_string.hpp; thr_bv, include/thresholds_ds.hpp)
// -- tens of dependent cache misses per searched character.  Here
// every run of the BWT is ONE 32-byte row that answers, with a single 32 B
// gather, everything a backward step needs at the run it lands in:
//
//   S      run start position            (run_of_position / select, Appendix B)
//   H      run head                      (bwt[pos], ms_rle_string.hpp:104)
//   len    run length
//   LFrun  run containing LF(S)          (move-structure pointer: replaces the
//   LFoff  LF(S) - S[LFrun]               rank() + run_of_position() of LF, :180-187)
//   THR    thresholds[run]               (thr_bv::operator[], thresholds_ds.hpp:478-491,
//                                         zero-skipping of :421-423 already applied)
//   docS/docE  start_runs_doc / end_runs_doc (doc_array.hpp:22-23)
//
// State of a walk is (k, off) with pos = S[k] + off.  A match step is
//   k' = LFrun[k], off' = LFoff[k] + off, then skip rows while off' >= len.
// A mismatch step needs the successor / predecessor run with head c: a
// per-letter directory Q_c (run indices, ascending) addressed through a block
// count table cnt[letter][k >> bshift] (number of c-runs before the block).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SPX_HD __host__ __device__ __forceinline__
#else
#define SPX_HD inline
#endif

namespace spx {

constexpr uint64_t MASK40 = (1ull << 40) - 1;
constexpr uint32_t NO_LETTER = 0xffffffffu;
constexpr int ROW_PAD = 8;  // sentinel + padding rows after row r-1
constexpr int Q_PAD = 16;   // padding entries after (and 1 before) the directory

struct alignas(32) Row {
    uint64_t q0;  // S[40] | H[8] << 40 | docS[16] << 48
    uint64_t q1;  // len[40] | LFrun[0:24] << 40
    uint64_t q2;  // LFoff[40] | LFrun[24:32] << 40 | docE[16] << 48
    uint64_t q3;  // THR[40] | reserved
};

SPX_HD Row pack_row(uint64_t S, uint32_t H, uint64_t len, uint32_t LFrun, uint64_t LFoff,
                    uint64_t THR, uint32_t docS, uint32_t docE) {
    Row r;
    r.q0 = (S & MASK40) | ((uint64_t)(H & 0xff) << 40) | ((uint64_t)(docS & 0xffff) << 48);
    r.q1 = (len & MASK40) | ((uint64_t)(LFrun & 0xffffff) << 40);
    r.q2 = (LFoff & MASK40) | ((uint64_t)(LFrun >> 24) << 40) | ((uint64_t)(docE & 0xffff) << 48);
    r.q3 = (THR & MASK40);
    return r;
}
SPX_HD uint64_t row_S(const Row& r) { return r.q0 & MASK40; }
SPX_HD uint32_t row_H(const Row& r) { return (uint32_t)(r.q0 >> 40) & 0xff; }
SPX_HD uint32_t row_docS(const Row& r) { return (uint32_t)(r.q0 >> 48); }
SPX_HD uint64_t row_len(const Row& r) { return r.q1 & MASK40; }
SPX_HD uint32_t row_LFrun(const Row& r) {
    return (uint32_t)(r.q1 >> 40) | ((uint32_t)((r.q2 >> 40) & 0xff) << 24);
}
SPX_HD uint64_t row_LFoff(const Row& r) { return r.q2 & MASK40; }
SPX_HD uint32_t row_docE(const Row& r) { return (uint32_t)(r.q2 >> 48); }
SPX_HD uint64_t row_THR(const Row& r) { return r.q3 & MASK40; }

// Directory row i (i = position in the (letter, run index) order; Q[i] is the run):
// everything a threshold jump needs, for BOTH outcomes, in one 32-byte gather:
//   q        run index Q[i]                 (successor c-run of the walk's position)
//   THR      thresholds[q]
//   sLF*     where LF(start of q) lands     (taken when pos >= THR: jump to successor)
//   pLF*     where LF(start of q) - 1 lands (= LF of the LAST character of run Q[i-1]:
//            taken when pos < THR: jump to predecessor; LF images of consecutive
//            directory rows are adjacent, also across letter boundaries)
//   docS     start_runs_doc[q];  docEp = end_runs_doc[Q[i-1]]
// Row i = r is a sentinel: no successor, predecessor = last run in directory order.
struct alignas(32) DirRow {
    uint64_t d0;  // q[32] | docS[16] << 32 | docEp[16] << 48
    uint64_t d1;  // THR[40] | sLFrun[0:24] << 40
    uint64_t d2;  // sLFoff[40] | sLFrun[24:32] << 40 | pLFrun[0:16] << 48
    uint64_t d3;  // pLFoff[40] | pLFrun[16:32] << 40
};

SPX_HD DirRow pack_dirrow(uint32_t q, uint64_t THR, uint32_t sLFrun, uint64_t sLFoff,
                          uint32_t pLFrun, uint64_t pLFoff, uint32_t docS, uint32_t docEp) {
    DirRow d;
    d.d0 = (uint64_t)q | ((uint64_t)(docS & 0xffff) << 32) | ((uint64_t)(docEp & 0xffff) << 48);
    d.d1 = (THR & MASK40) | ((uint64_t)(sLFrun & 0xffffff) << 40);
    d.d2 = (sLFoff & MASK40) | ((uint64_t)(sLFrun >> 24) << 40) | ((uint64_t)(pLFrun & 0xffff) << 48);
    d.d3 = (pLFoff & MASK40) | ((uint64_t)(pLFrun >> 16) << 40);
    return d;
}
SPX_HD uint32_t dir_q(const DirRow& d) { return (uint32_t)d.d0; }
SPX_HD uint32_t dir_docS(const DirRow& d) { return (uint32_t)(d.d0 >> 32) & 0xffff; }
SPX_HD uint32_t dir_docEp(const DirRow& d) { return (uint32_t)(d.d0 >> 48); }
SPX_HD uint64_t dir_THR(const DirRow& d) { return d.d1 & MASK40; }
SPX_HD uint32_t dir_sLFrun(const DirRow& d) {
    return (uint32_t)(d.d1 >> 40) | ((uint32_t)((d.d2 >> 40) & 0xff) << 24);
}
SPX_HD uint64_t dir_sLFoff(const DirRow& d) { return d.d2 & MASK40; }
SPX_HD uint32_t dir_pLFrun(const DirRow& d) {
    return (uint32_t)(d.d2 >> 48) | ((uint32_t)((d.d3 >> 40) & 0xffff) << 16);
}
SPX_HD uint64_t dir_pLFoff(const DirRow& d) { return d.d3 & MASK40; }

// per byte value c: everything the walk needs that depends only on the letter
struct alignas(16) LetterInfo {
    uint32_t lid;    // dense letter id, NO_LETTER if number_of_letter(c) == 0
    uint32_t qbeg;   // directory range of the letter: Q[qbeg, qend)
    uint32_t qend;
    uint32_t frun;   // run containing F[c] (r if F[c] == n): landing after an absent letter
    uint64_t foff;   // F[c] - S[frun]
    uint64_t pad_;
};

struct SamplePair {  // MS mode, directory order: entry i = {samples_start[Q[i]], samples_last[Q[i-1]]}
    uint64_t ss;
    uint64_t se;
};

// kernel-visible view of an index (all pointers are device memory)
struct DevIndex {
    const Row* rows;            // r + ROW_PAD rows; row r is the "pos == n" sentinel
    const DirRow* dirrows;      // r + 1 (+ pad) directory rows, (letter, run) order
    const uint32_t* cnt;        // [nletters][nblk] directory offsets (absolute into Q)
    const uint32_t* Q;          // directory; Q[-1] and Q[qtotal .. +Q_PAD) are readable
    const SamplePair* samples;  // r + 1 entries in directory order, or nullptr
    const uint64_t* ss_by_run;  // samples_start by run index (+2 pad) or nullptr
    const LetterInfo* letters;  // 256 entries
    const uint8_t* text;        // MS extension text or nullptr
    uint64_t n_text;
    uint64_t n;
    uint32_t r;
    uint32_t nblk;      // blocks per letter in cnt (= (r >> bshift) + 2)
    uint32_t bshift;    // log2(runs per directory block)
    uint32_t init_k;    // run of position n-1  (= r-1)
    uint64_t init_off;  // (n-1) - S[r-1]
    uint64_t init_sample;  // get_last_run_sample(): (samples_last[r-1] + 1) % n
    uint32_t init_doc;     // end_runs_doc[r-1]         (compute_ms_pml.cpp:298)
    uint32_t doc_at0;      // start_runs_doc[run_of_position(0)]   (:641-642)
};

}  // namespace spx
