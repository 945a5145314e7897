// spx_walk.hip -- the backward-search kernels (hand-written HIP for gfx950).
//
// What is computed is exactly pml_pointers::_query / ms_pointers::_query of the
// reference (/root/reference/src/compute_ms_pml.cpp:238-340, 571-682) and the
// length extension of ms_t::matching_statistics (:800-810); how it is computed
// is ours (spx_layout.h).
//
// k_walk_lanes -- "lane-per-read state machine".
// The walk of one read is a chain of dependent small gathers (1 per matching
// character, 2-3 per threshold jump), and MI355X's random-gather ceiling is
// ~50 G 16-byte gathers/s (41 G/s for 32-byte ones; tools/gather_bench.hip),
// reached only with >= 64k chains in flight -- far more chains than the 8k
// wavefronts the chip holds.  So every LANE owns one read and runs a small
// state machine.  Each loop iteration issues exactly ONE gather per lane, from
// a single load site, at the address the lane's current phase asks for:
//
//   P_READ   offsets[rd], offsets[rd+1]   16 B  (next read of this lane)
//   P_CHARS  32 characters of the read    32 B  (refill, every <= 32 steps)
//   P_LAND   rows[k0]                     16 B  (run the LF step lands in)
//   P_FAT    fat[fbase_c + blk_c(k)]      16 B  (digest of the jump row of the first c-run
//                                                at or after the block: usually THE answer;
//                                                the next slot when this one says so)
//   P_FATJ   fat_js[slot >> 3]             4 B  (only if the block holds c-runs < k -- or a byte >= 128 sits on its
//                                                own run with a threshold to be asked: where the scan starts)
//   P_QS     Q[j .. j+8)                  32 B  (after P_FATJ: the first run of the letter at or after k)
//   P_DIR    dirrows[j]                   32 B  (after P_QS, or when the digest does not hold the row: it holds j)
//   P_AUX / P_SAMP                              (MS samples / document ids)
//
// then consumes it and moves to the next phase.  Because the load site is
// unique, all 64 lanes of a wave always have their gather in flight together
// no matter which phase each is in: no lane waits for another lane's longer
// step, and no phase serialises behind another phase's s_waitcnt.
#include <hipcub/hipcub.hpp>

#include <type_traits>

#include "spx_internal.h"

namespace spx {

namespace {

enum : uint32_t {
    P_READ = 0,
    P_CHARS = 1,
    P_LAND = 2,
    P_FAT = 3,
    P_QS = 4,
    P_DIR = 5,
    P_AUX = 6,
    P_SAMP = 7,
    P_FATJ = 8,
    P_DONE = 9,
    P_START = 10,  // chunked walk, pass 2: the end state of the chunk above
    P_CKPT = 11    // chunked walk, pass 2: the speculative walk's state at a checkpoint
};

constexpr int WALK_TPB = 256;
#ifndef SPX_LEN_G
#define SPX_LEN_G 32
#endif
constexpr int LEN_G = SPX_LEN_G;  // PML lengths / document ids staged per flush
#ifndef SPX_PTR_G
#define SPX_PTR_G 4
#endif
constexpr int PTR_G = SPX_PTR_G;  // MS pointers staged per flush
// (document ids 8 and pointers 4 per flush: with 32 / 8 the doc and MS variants need ~100 VGPRs, a wavefront per SIMD
// fewer fits, and they run 10-15 % slower -- C4 shape MS+doc 9.4 -> 10.8 ms, PML+doc 8.2 -> 9.5 ms)
constexpr int DOC_G = 8;

struct __attribute__((packed, aligned(4))) V16 {  // 16 bytes at 4-byte alignment
    uint32_t x, y, z, w;
};

__device__ __forceinline__ uint64_t u64of(uint32_t lo, uint32_t hi) {
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// LF of position (k, off) given row k's move pointer: the run LFrun at offset LFoff + off, or --
// when the row's `room` says the offset overshoots that run -- the next run directly.
template <class OFFS>
__device__ __forceinline__ void lf_target(uint32_t LFrun, OFFS LFoff, uint32_t room, OFFS off, uint32_t& k0,
                                          OFFS& offp) {
    const bool over = room < ROOM_SAT && off >= room;
    k0 = LFrun + (over ? 1u : 0u);
    offp = over ? off - room : LFoff + off;
}

// The same for compact rows (spx_layout.h): cums = the offsets at which the step leaves runs
// LFrun, LFrun+1, LFrun+2, LFrun+3 (saturated at 127), so the destination run is known exactly
// unless it lies further than that.
template <class OFFS>
__device__ __forceinline__ void lf_target_c(uint32_t LFrun, OFFS LFoff, uint32_t cums, OFFS off, uint32_t& k0,
                                            OFFS& offp) {
    const uint32_t o = off < 126 ? (uint32_t)off : 126u;
    // byte i gets its top bit iff cum_i <= o (no borrows: every byte of the minuend is >= 128)
    const uint32_t flags = (((o * 0x01010101u) | 0x80808080u) - cums) & 0x80808080u;
    const uint32_t t = __popc(flags);  // cums ascend: the bytes that qualify are a prefix
    const uint32_t prev = (cums >> (8 * ((t + 3) & 3))) & 0x7fu;  // cum_{t-1} (unused when t == 0)
    k0 = LFrun + t;
    offp = t ? off - prev : LFoff + off;
}

// Output staging.  A lane's stores go to its own read: 64 lanes, 64 different cache lines per store
// instruction, and small scattered writes are expensive in DRAM (round 2: the walk without its length
// stores ran 13.0 -> 10.0 ms, with all of them landing in one hot megabyte 11.1 ms).  So values are collected
// in registers -- a group of consecutive elements of the READ, as u16 -- and written back to back, with as few
// instructions as the group allows, when its lowest element has been produced.  Groups are aligned to the read, not to memory: only the read's
// top group can be incomplete (round 1 aligned them to memory, which left a ragged group at BOTH ends of
// every read and scalar stores for each of its elements: 16 store instructions per 44-character read,
// now 11; with 16-bit outputs 6).  The vector stores are therefore only element-aligned, which gfx950
// global stores take.  (It made no difference to the time; nor did 32 values per lane through LDS and
// 64 / 128 contiguous bytes per flush, which was slower: profiles/r02_store_experiments.txt.)
struct __attribute__((packed, aligned(4))) U32x4 {
    uint32_t x, y, z, w;
};
struct __attribute__((packed, aligned(4))) U32x2 {
    uint32_t x, y;
};
struct __attribute__((packed, aligned(2))) H16x8 {
    uint64_t lo, hi;
};
struct __attribute__((packed, aligned(2))) H16x4 {
    uint64_t v;
};
struct __attribute__((packed, aligned(2))) H16x2 {
    uint32_t v;
};
struct __attribute__((packed, aligned(8))) P64x2 {
    uint64_t x, y;
};
__device__ __forceinline__ U32x4 widen4(uint64_t w) {
    return U32x4{(uint32_t)w & 0xffff, (uint32_t)(w >> 16) & 0xffff, (uint32_t)(w >> 32) & 0xffff, (uint32_t)(w >> 48)};
}

// Lengths, G of them per flush (as u16 in G / 4 registers; the values of a read shorter than 65 536 characters
// fit): written as 16-bit values (one 16-byte store per eight) or widened to 32 bits (two), back to back.  More per
// flush means fewer, larger writes reaching DRAM: 12.8 / 12.25 / 12.0 ms at 8 / 16 / 32 values per flush (16-bit
// outputs, profiles/r02_store_experiments.txt).
template <int G>
struct StageN {
    uint64_t a[G / 4];
};
template <int G, bool NARROW_OUT>
__device__ __forceinline__ void stage_n(StageN<G>& st, uint32_t value, uint32_t* out32, uint64_t base, uint32_t xi,
                                        uint32_t m) {
    const uint32_t idx = xi & (G - 1), r = idx >> 2;
    const uint64_t v = (uint64_t)(value & 0xffffu) << ((idx & 3) * 16);
#pragma unroll
    for (int j = 0; j < G / 4; ++j) st.a[j] |= (r == (uint32_t)j) ? v : 0ull;
    if (idx == 0) {
        const uint32_t cnt = m - xi;  // >= G for every group but the read's top one
#pragma unroll
        for (int q = 0; q < G / 8; ++q) {
            const uint64_t lo = st.a[2 * q], hi = st.a[2 * q + 1];
            if (cnt > 8u * q) {
                const uint32_t c = cnt - 8 * q;
                if (NARROW_OUT) {
                    uint16_t* p = reinterpret_cast<uint16_t*>(out32) + base + xi + 8 * q;
                    if (c >= 8) {
                        *reinterpret_cast<H16x8*>(p) = H16x8{lo, hi};
                    } else {
                        uint64_t src = lo;
                        if (c & 4) {
                            *reinterpret_cast<H16x4*>(p) = H16x4{lo};
                            src = hi;
                            p += 4;
                        }
                        if (c & 2) {
                            *reinterpret_cast<H16x2*>(p) = H16x2{(uint32_t)src};
                            src >>= 32;
                            p += 2;
                        }
                        if (c & 1) *p = (uint16_t)src;
                    }
                } else {
                    uint32_t* p = out32 + base + xi + 8 * q;
                    if (c >= 8) {
                        *reinterpret_cast<U32x4*>(p) = widen4(lo);
                        *reinterpret_cast<U32x4*>(p + 4) = widen4(hi);
                    } else {
                        uint64_t src = lo;
                        if (c & 4) {
                            *reinterpret_cast<U32x4*>(p) = widen4(lo);
                            src = hi;
                            p += 4;
                        }
                        if (c & 2) {
                            *reinterpret_cast<U32x2*>(p) = U32x2{(uint32_t)src & 0xffff, (uint32_t)(src >> 16) & 0xffff};
                            src >>= 32;
                            p += 2;
                        }
                        if (c & 1) *p = (uint32_t)src & 0xffff;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < G / 4; ++j) st.a[j] = 0;
    }
}

// ---------------------------------------------------------------------------
// lane-per-read state machine
// ---------------------------------------------------------------------------
// CHUNK: 0 = reads; 1 = pass 1 of the chunked walk (every chunk from the default state, checkpoints
// and end states recorded); 2 = pass 2 (a chunk entered with the end state of the chunk above,
// until the walk meets the recorded one) -- spx_internal.h
template <int MODE, bool DOC, bool COMPACT, bool NARROW, int CHUNK = 0>
__global__ void __launch_bounds__(WALK_TPB) k_walk_lanes(const DevIndex ix, const BatchArgs b) {
    constexpr bool AUX = (MODE == SPX_MODE_MS) || DOC;  // per-jump side data (samples / doc ids)
    const uint64_t nitems = CHUNK ? *b.ch.nchunks : b.nreads;  // work items: reads, or chunks
    if (CHUNK == 2 && b.ch.round > 1 && b.ch.pending[b.ch.round] == 0) return;  // no seam was left open
    if (CHUNK == 0 && b.only_flagged != nullptr && b.counters->pad_ == 0) return;  // no read fell back
    __shared__ LetterInfo s_let[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) s_let[t] = ix.letters[t];
    __syncthreads();

    const char* const rows_b = reinterpret_cast<const char*>(ix.rows);
    const char* const dir_b = reinterpret_cast<const char*>(ix.dirrows);
    const char* const fat_b = reinterpret_cast<const char*>(ix.fat);
    const char* const fatj_b = reinterpret_cast<const char*>(ix.fat_js);
    uint32_t fadd = 0;       // 1: the jump is looking at the slot AFTER its block's (FAT_SINGLE)
    constexpr uint32_t FAT_ROW = sizeof(FatRow);
    const char* const q_b = reinterpret_cast<const char*>(ix.Q);
    const char* const seq_b = reinterpret_cast<const char*>(b.seqs);
    const char* const off_b = reinterpret_cast<const char*>(b.offs);
    const uint32_t R = ix.r;
    const bool want_class = (CHUNK == 0) && (MODE == SPX_MODE_PML) && b.out_class != nullptr;
    // lanes_per_wave < 64 spreads a small batch over more wavefronts (see launch_lanes)
    const uint32_t lpw = b.lanes_per_wave;
    const uint64_t nlanes = (((uint64_t)gridDim.x * blockDim.x) >> 6) * lpw;

    uint32_t ph = P_READ;
    uint64_t rd = ((((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * lpw) + (threadIdx.x & 63);
    uint64_t base = 0;
    uint32_t m = 0, x = 0;  // x = characters still to search; next one is index x-1
    // landed position: run k, offset off; fields of row k
    uint32_t k = 0, H_k = 0, LFrun_k = 0, room_k = 0;
    // offsets inside runs: 16 bits are enough with compact rows, 40 bits otherwise
    typedef typename std::conditional<COMPACT, uint32_t, uint64_t>::type offs_t;
    constexpr offs_t OFF_LAST = (offs_t)OFF_END;  // "last position of the run", resolved at landing
    offs_t off = 0, LFoff_k = 0;
    bool thr_ok_k = true;
    // landing target
    uint32_t k0 = 0;
    offs_t offp = 0;
    // per-step results
    uint32_t length = 0, doc = 0;
    uint64_t sample = 0;
    // jump bookkeeping
    uint32_t c = 0, jdir = 0, qbeg = 0, qend = 0, aux_take = 0, Hland = 0;
    bool quirk = false, peek = false;
    // character window: the 32 bytes of the read starting at byte offset wbase of seqs, kept in LDS
    // as [dword j of the window][thread] (conflict-free) -- eight registers and a select chain less
    // than holding it in VGPRs (90 -> 72 registers, DNA walks +7 %)
    __shared__ uint32_t s_win[8 * WALK_TPB];
    uint64_t wbase = 0;
#define WIN_CHAR(wi) ((s_win[((wi) >> 2) * WALK_TPB + threadIdx.x] >> (((wi)&3) * 8)) & 0xffu)
    // output staging (PML): u16 values of the current group of outputs
    StageN<LEN_G> obn{};  // PML lengths (pass 1 of the chunked walk)
    // PML lengths of the plain walk: one bit per character (BatchArgs::len_mask), one 16-byte store per 128
    // characters -- for a read of up to 128 characters a single store when the read is done
    uint64_t mbits = 0, mbits_hi = 0;
    const uint64_t base0 = (CHUNK == 0 && MODE == SPX_MODE_PML && b.len_mask != nullptr) ? b.offs[0] : 0;
    StageN<DOC_G> dbn{};  // document ids
    uint64_t pbs[PTR_G] = {};  // MS pointers of the current group
    // classifier
    uint32_t bin_lo = 0, bin_max = 0, above = 0, below = 0;
    uint64_t sum_max = 0;
    // statistics
    uint32_t n_steps = 0, n_jumps = 0, n_pred = 0, n_rows = 0, n_dir = 0, n_err = 0;

    // chunked walks
    uint32_t rflag = 0;      // what the current step reset: bit 0 length / sample, bit 1 document id
    uint64_t fb = 0;         // pass 1: flags of the aligned group of 8 characters, a byte each
    uint32_t item_rd = 0, item_flags = 0, seen = 0, ph_after = P_LAND, tlimit = 0xffffffffu;
    // the chunk scratch is indexed by character index relative to the batch's first character: offsets are
    // absolute (a piece of a larger batch, or a caller whose offsets[0] != 0), the scratch is sized by total_chars
    const uint64_t ch_org = CHUNK ? b.offs[0] : 0;
    uint8_t* const ch_flags = CHUNK ? b.ch.flags - (ch_org & ~7ull) : nullptr;
    WalkState* const ch_ckpt = CHUNK ? b.ch.ckpt - (ch_org >> CKPT_SHIFT) : nullptr;
    if (rd >= nitems || (threadIdx.x & 63) >= lpw) ph = P_DONE;
#define STAND_ON(row)                                \
    do {                                             \
        if (COMPACT) {                               \
            H_k = crow_H(row);                       \
            LFrun_k = crow_LFrun(row);               \
            LFoff_k = crow_LFoff(row);               \
            thr_ok_k = crow_thr_ok(row);             \
            room_k = crow_cums(row);                 \
        } else {                                     \
            H_k = row_H(row);                        \
            LFrun_k = row_LFrun(row);                \
            LFoff_k = (offs_t)row_LFoff(row);        \
            thr_ok_k = row_thr_ok(row);              \
            room_k = row_room(row);                  \
        }                                            \
    } while (0)
// next work item of this lane: static striding, no atomics.  (Chunks taken from a shared counter as
// lanes become free were tried: equal chunks end together, and a few hundred thousand atomics on one
// address at once cost more than the imbalance they remove -- pass 1 5.1 instead of 4.3 ms.  The chunk
// size is chosen so that every lane gets a whole number of chunks instead.)
#define NEXT_ITEM()                               \
    do {                                          \
        rd += nlanes;                             \
        ph = rd < nitems ? P_READ : P_DONE;       \
    } while (0)
#define LF_TARGET()                                                      \
    do {                                                                 \
        if (COMPACT)                                                     \
            lf_target_c(LFrun_k, LFoff_k, room_k, off, k0, offp);        \
        else                                                             \
            lf_target(LFrun_k, LFoff_k, room_k, off, k0, offp);          \
    } while (0)

    while (ph != P_DONE) {
        // ---- the one gather of this iteration -------------------------------
        const char* p0;
        uint64_t fidx = 0;  // fat-table slot (P_FAT only)
        if (ph == P_LAND) {
            // (a compact index keeps its rows 32 bytes apart, spx_layout.h Row32: this kernel reads the row itself,
            // k_walk_fast also what is embedded next to it)
            p0 = rows_b + (uint64_t)k0 * (COMPACT ? sizeof(Row32) : sizeof(Row));
        } else if (ph == P_FAT) {
            fidx = s_let[c].fbase + fat_block(k, s_let[c].bmul) + fadd;
            p0 = fat_b + fidx * ix.fat_stride;
        } else if (ph == P_FATJ) {
            fidx = (s_let[c].fbase + fat_block(k, s_let[c].bmul) + fadd) >> FJ_SHIFT;  // the slot's group
            p0 = fatj_b + ((fidx * 4) & ~15ull);  // the aligned 16 bytes that hold fat_js[group]
        } else if (ph == P_DIR) {
            p0 = dir_b + (uint64_t)jdir * sizeof(JumpRow);
        } else if (ph == P_QS) {
            p0 = q_b + (uint64_t)jdir * 4;
        } else if (ph == P_CHARS) {
            p0 = seq_b + wbase;
        } else if (ph == P_READ) {
            p0 = CHUNK ? reinterpret_cast<const char*>(b.ch.desc + rd) : off_b + rd * 8;
        } else if (CHUNK == 2 && ph == P_START) {
            // round 1: the recorded end state of the chunk above (the next one); later: what the scan left
            p0 = reinterpret_cast<const char*>(b.ch.round > 1 ? b.ch.reentry + rd : b.ch.ends + rd + 1);
        } else if (CHUNK == 2 && ph == P_CKPT) {
            p0 = reinterpret_cast<const char*>(ch_ckpt + ((base + x) >> CKPT_SHIFT));
        } else if (MODE == SPX_MODE_MS && ph == P_SAMP) {
            p0 = reinterpret_cast<const char*>(ix.ss_by_run + k);
        } else {
            p0 = rows_b;
        }
        const bool wide = (ph == P_DIR) | (ph == P_QS) | (ph == P_CHARS) | (CHUNK == 2 && (ph == P_START || ph == P_CKPT));
        const V16 ga = *reinterpret_cast<const V16*>(p0);
        V16 gb{0, 0, 0, 0};
        if (wide) gb = *reinterpret_cast<const V16*>(p0 + 16);
        // side data of the jump row being fetched travels in parallel with it: MS samples
        // {samples_start[q], samples_last[previous c-run]} and doc ids {start_runs_doc[q],
        // end_runs_doc[previous c-run]}, from the fat copies (P_FAT) or by directory position
        SamplePair sp{0, 0};
        uint32_t dd = 0;
        if (AUX) {  // one 16-byte record: from the fat slot itself (P_FAT) or by directory position
            const Aux* pa = (ph == P_FAT) ? reinterpret_cast<const Aux*>(p0 + FAT_ROW)
                                          : ix.aux + ((ph == P_DIR || ph == P_AUX) ? jdir : 0);
            const Aux a = *pa;
            if (MODE == SPX_MODE_MS) {
                sp.ss = aux_ss(a);
                sp.se = aux_se(a);
            }
            if (DOC) dd = aux_docs(a);
        }
        if (DOC && ph == P_SAMP) dd = ix.rundocs[k];
        uint64_t g0 = u64of(ga.x, ga.y), g1 = u64of(ga.z, ga.w);
        uint64_t g2 = u64of(gb.x, gb.y), g3 = u64of(gb.z, gb.w);

        // ---- consume ---------------------------------------------------------
        bool do_emit = false;    // a character's result is final -> write it and advance
        bool do_step = false;    // landed on a run -> look at the next character
        bool do_decide = false;  // g0..g3 hold the jump row that answers the jump
        if (ph == P_LAND) {
            n_rows++;
            Row ra;
            ra.q0 = g0;
            ra.q1 = g1;
            const offs_t len = COMPACT ? (offs_t)crow_len(ra) : (offs_t)row_len(ra);
            if (offp == OFF_LAST) offp = (offs_t)len - 1;  // predecessor landing: last position of this run
            if (offp >= len) {  // LF image lies in a later run: skip this row
                offp -= len;
                k0++;
            } else {
                k = k0;
                off = offp;
                STAND_ON(ra);
                do_step = true;
            }
        } else if (ph == P_FAT) {
            n_dir++;
            const uint32_t hq = (uint32_t)g0;
            const bool nosucc = (g1 >> 61) & 1, esc = (g1 >> 62) & 1;
            // the block's first c-run is the successor unless it lies before k (or is k
            // itself when the walk sits on a c-run: byte >= 128, Appendix C1)
            const bool fat_found = nosucc || hq > k || (quirk && hq == k);
            if (fat_found && !esc && !quirk) {
                // the 16 bytes are the whole answer: spread them into JumpRow form for the decision
                // below (j is not known and not needed: only its place in [qbeg, qend] matters)
                const uint64_t w0 = g0, w1 = g1;
                const uint32_t srun = (uint32_t)(w0 >> 32);
                g0 = (uint64_t)hq | ((uint64_t)(hq - (uint32_t)(w1 & 0x7ffff)) << 32);
                g1 = ((w1 >> 20) & 0xffff) | ((uint64_t)(srun & 0xffffff) << 40);
                // (the sLFoff field holds the offset only when psame; otherwise Hp, which this body does not use)
                g2 = (((w1 >> 60) & 1) ? ((w1 >> 36) & 0xffff) : 0ull) | ((uint64_t)(srun >> 24) << 40) | (((w1 >> 60) & 1) << 48) |
                     (((w1 >> 52) & 0xff) << 49);
                jdir = nosucc ? qend : (((w1 >> 63) & 1) ? qbeg : qbeg + 1);
                do_decide = true;
            } else if (!fat_found && (g1 & FAT_SINGLE) && fadd == 0) {
                fadd = 1;  // the letter's next run lies past this block: it is the next slot's run
            } else if (fat_found && esc && !quirk) {
                jdir = (uint32_t)(g0 >> 32);  // the row did not fit the digest: the slot names its directory position
                ph = P_DIR;
            } else {
                ph = P_FATJ;  // scan the directory for the first run of the letter at or after k
            }
        } else if (ph == P_FATJ) {
            n_dir++;
            const uint32_t sel = (uint32_t)fidx & 3;
            const uint64_t gsel = (sel & 2) ? g1 : g0;
            jdir = (sel & 1) ? (uint32_t)(gsel >> 32) : (uint32_t)gsel;  // where the slot's group starts in the directory
            ph = P_QS;
        } else if (ph == P_QS) {
            n_dir++;
            const uint32_t e[8] = {(uint32_t)g0, (uint32_t)(g0 >> 32), (uint32_t)g1, (uint32_t)(g1 >> 32),
                                   (uint32_t)g2, (uint32_t)(g2 >> 32), (uint32_t)g3, (uint32_t)(g3 >> 32)};
            // window holds Q[jdir .. jdir+8): skip the c-runs with index < k (<= k when sitting
            // on a c-run and looking for that very run)
            // Q is ascending and "still inside the letter's range" is a prefix property too, so
            // the entries to skip form a prefix of the window: count them
            const uint32_t room_q = qend - jdir;  // entries of this letter left from jdir on
            uint32_t skip = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) skip += ((uint32_t)t < room_q && e[t] < k) ? 1u : 0u;
            jdir += skip;
            if (skip < 8) ph = P_DIR;
        } else if (ph == P_DIR) {
            n_dir++;
            do_decide = true;
        } else if (ph == P_CHARS) {
            s_win[0 * WALK_TPB + threadIdx.x] = (uint32_t)g0;
            s_win[1 * WALK_TPB + threadIdx.x] = (uint32_t)(g0 >> 32);
            s_win[2 * WALK_TPB + threadIdx.x] = (uint32_t)g1;
            s_win[3 * WALK_TPB + threadIdx.x] = (uint32_t)(g1 >> 32);
            s_win[4 * WALK_TPB + threadIdx.x] = (uint32_t)g2;
            s_win[5 * WALK_TPB + threadIdx.x] = (uint32_t)(g2 >> 32);
            s_win[6 * WALK_TPB + threadIdx.x] = (uint32_t)g3;
            s_win[7 * WALK_TPB + threadIdx.x] = (uint32_t)(g3 >> 32);
            do_step = true;  // only entered from a landed state
        } else if (CHUNK == 2 && ph == P_START) {
            // the walk that ended the chunk above goes on into this one
            k0 = (uint32_t)g0;
            length = (uint32_t)(g0 >> 32);
            offp = (g1 == OFF_END) ? OFF_LAST : (offs_t)g1;
            sample = g2;
            doc = (uint32_t)g3;
            // entered again: an earlier, unfounded walk wrote down to here -- write at least as far
            tlimit = b.ch.round > 1 ? (uint32_t)(g3 >> 32) : 0xffffffffu;
            seen = 0;
            wbase = ~0ull;  // no characters yet: the first step fetches its window
            ph = P_LAND;
        } else if (CHUNK == 2 && ph == P_CKPT) {
            n_dir++;
            const uint64_t my_off = (offp == OFF_LAST) ? OFF_END : (uint64_t)offp;
            if ((uint32_t)g0 == k0 && g1 == my_off && x <= tlimit) {
                // same position before the same character: from here on the speculative walk IS this walk
                SeamRec sr;
                sr.t = base + x;
                sr.met = 1 | (b.ch.round << 8);
                sr.reset_above = seen;
                sr.ext = WalkState{k0, length, my_off, sample, doc, seen};
                sr.spec_length = (uint32_t)(g0 >> 32);
                sr.spec_doc = (uint32_t)g3;
                sr.spec_sample = g2;
                b.ch.seams[rd] = sr;
                NEXT_ITEM();
            } else {
                ph = ph_after;
            }
        } else if (ph == P_READ) {
            if (CHUNK) {  // ChunkDesc: gend, len | top << 31 | bottom << 30, read
                item_flags = (uint32_t)g1 >> 29;  // bit 2 last chunk of its read, bit 1 first, bit 0 to be entered
                item_rd = (uint32_t)(g1 >> 32);
                m = (uint32_t)g1 & CHUNK_LEN_MASK;
                base = g0 - m;
            } else {
                base = g0;
                m = (uint32_t)(g1 - g0);
            }
            if (CHUNK == 0 && b.only_flagged != nullptr && b.only_flagged[rd] == 0) m = 0;  // not this pass's read
            if (CHUNK == 2 && ((item_flags & 4) || (b.ch.round > 1 && !(item_flags & 1)))) {
                // a read's last chunk was right from the start; later rounds: only below open seams
                NEXT_ITEM();
            } else if (m == 0) {
                if (want_class && b.only_flagged == nullptr) b.out_class[rd] = spx_class{0, 0, 0};
                NEXT_ITEM();
            } else {
                if (NARROW && m >= 65536) n_err++;  // 16-bit outputs cannot hold this read's values
                x = m;
                length = 0;
                sample = ix.init_sample;  // compute_ms_pml.cpp:575
                doc = ix.init_doc;        // :298 / :634
                // pos = bwt_size() - 1 (:243 / :574): the same row for every read, kept in the
                // kernel arguments -- the walk stands on it at once and fetches its first characters
                k = ix.init_k;
                off = (offs_t)ix.init_off;
                {
                    Row ir;
                    ir.q0 = ix.init_row.q0;
                    ir.q1 = ix.init_row.q1;
                    STAND_ON(ir);
                }
                {
                    const uint64_t end4 = (base + m + 3) & ~3ull;  // window ends past the last character
                    wbase = end4 >= 32 ? end4 - 32 : 0;
                }
                fb = 0;
                seen = 0;
                if (CHUNK == 2) ph = P_START;
                if (want_class) {
                    const uint32_t w = (uint32_t)b.bin_width;
                    // m / w without a divide: bin_magic = floor(2^64 / w) + 1 (exact for m < 2^32)
                    const uint32_t q = w > 1 ? (uint32_t)__umul64hi((uint64_t)m, b.bin_magic) : m;
                    const uint32_t nb = q > 0 ? q : 1;
                    bin_lo = (nb - 1) * w;
                    bin_max = above = below = 0;
                    sum_max = 0;
                }
                if (CHUNK != 2) ph = P_CHARS;
            }
        } else if (ph == P_AUX) {
            // only the inconsistent-threshold case of Appendix C1 comes here: side data of the
            // directory position AFTER run k (samples_last[k] / end_runs_doc[k])
            if (MODE == SPX_MODE_MS) sample = sp.se;
            if (DOC) doc = dd >> 16;
            rflag = 3;
            do_emit = true;
        } else {  // P_SAMP: byte >= 128 sitting on its own run (Appendix C1): stays there
            if (MODE == SPX_MODE_MS) sample = g0;  // samples_start[run of pos]
            if (DOC) doc = dd & 0xffff;            // start_runs_doc[run of pos]
            LF_TARGET();
            rflag = 3;
            do_emit = true;
        }

        if (do_decide) {
            // compute_ms_pml.cpp:253-278 (PML) / :585-615 (MS) on the flat layout
            JumpRow e;
            e.d0 = g0;
            e.d1 = g1;
            e.d2 = g2;
            e.d3 = g3;
            const bool has_succ = jdir < qend;  // rnk < number_of_letter(c)   (:259)
            const uint32_t trun = jr_THRrun(e);
            // pos < thr, with thr = n + 1 when there is no successor (:254, :270)
            const bool below_thr = !has_succ || (k < trun) || (k == trun && off < (offs_t)jr_THRoff(e));
            const uint32_t srun = jr_sLFrun(e);
            const offs_t soff = (offs_t)jr_sLFoff(e);
            length = 0;
            aux_take = 0;
            peek = false;
            if (!quirk) {
                if (!below_thr) {  // next_pos = first position of the next c-run; LF of it
                    k0 = srun;
                    offp = soff;
                    Hland = jr_Hs(e);
                    peek = true;
                } else {  // select(rnk-1, c): last position of the previous c-run; LF of it
                    n_pred++;
                    if (jdir <= qbeg) n_err++;  // rnk-- below zero: undefined upstream
                    aux_take = 1;
                    const bool ps = jr_psame(e);
                    k0 = ps ? srun : srun - 1;
                    offp = ps ? soff - 1 : OFF_LAST;
                    Hland = jr_Hs(e);  // only looked at when the landing stays in run sLFrun (peek)
                    peek = ps;  // the exact offset is only known when it stays in run sLFrun
                }
            } else {
                // the walk sits on run k whose head equals c >= 128 (Appendix C1) and
                // thresholds[k] > S[k] (inconsistent thresholds): jdir is k's directory position
                if (!below_thr) {  // select(rank(pos,c),c) == pos: stay
                    k0 = LFrun_k;
                    offp = LFoff_k + off;
                } else if (off > 0) {  // select(rnk-1, c) = pos - 1, still inside run k
                    n_pred++;
                    aux_take = 1;
                    jdir++;  // directory entry after k carries samples_last[k] / end_runs_doc[k]
                    k0 = LFrun_k;
                    offp = LFoff_k + off - 1;
                } else {
                    n_pred++;
                    if (jdir <= qbeg) n_err++;
                    aux_take = 1;
                    const bool ps = jr_psame(e);
                    k0 = ps ? srun : srun - 1;
                    offp = ps ? soff - 1 : OFF_LAST;
                }
            }
            if (AUX && quirk && below_thr && off > 0) {
                ph = P_AUX;  // needs the NEXT directory position's side data (jdir was advanced)
            } else {
                if (MODE == SPX_MODE_MS) sample = aux_take ? sp.se : sp.ss;   // :601 / :611
                if (DOC) doc = aux_take ? (dd >> 16) : (dd & 0xffff);          // :317 / :327
                rflag = 3;
                do_emit = true;
            }
        }

        if (do_step) {
            // next character: auto c = pattern[m - i - 1]   (:247)
            const uint64_t g = base + x - 1;
            if (g < wbase || g - wbase >= 32) {  // not in the window (wbase == ~0: none yet): refill
                const uint64_t end4 = (g + 4) & ~3ull;  // window ends at the next 4-byte boundary
                wbase = end4 >= 32 ? end4 - 32 : 0;
                ph = P_CHARS;
            } else {
                const uint32_t wi = (uint32_t)(g - wbase);
                c = WIN_CHAR(wi);
                const LetterInfo li = s_let[c];
                if (li.qbeg == li.qend) {  // number_of_letter(c) == 0   (:249)
                    length = 0;
                    if (MODE == SPX_MODE_MS) {
                        sample = 0;                 // :581
                        if (DOC) doc = ix.doc_at0;  // :641-642
                    }
                    k0 = li.frun;  // LF(pos, c) = F[c] + 0
                    offp = (offs_t)li.foff;
                    rflag = MODE == SPX_MODE_MS ? 3 : 1;  // PML keeps its document id over an absent letter
                    do_emit = true;
                } else if (k < R && H_k == c && c < 128) {  // pos < n && bwt[pos] == c   (:250)
                    length++;
                    sample--;  // :582 (wraps, Appendix C3)
                    LF_TARGET();
                    rflag = 0;
                    do_emit = true;
                } else if (k < R && H_k == c && thr_ok_k) {
                    // byte >= 128 equal to the head (signed-char quirk, Appendix C1): the jump
                    // branch runs, select(rank(pos,c),c) == pos, and thresholds[k] <= S[k] <= pos
                    // keeps it there
                    n_jumps++;
                    length = 0;
                    if (AUX) {
                        ph = P_SAMP;  // sample = samples_start[k], doc = start_runs_doc[k]
                    } else {
                        LF_TARGET();
                        rflag = 3;
                        do_emit = true;
                    }
                } else {
                    n_jumps++;
                    quirk = (k < R && H_k == c);
                    qbeg = li.qbeg;
                    qend = li.qend;
                    fadd = 0;
                    ph = P_FAT;
                }
            }
        }

        if (do_emit) {
            const uint32_t xi = x - 1;
            const uint64_t gi = base + xi;
            if (CHUNK) {
                // which counters this step reset: what pass 3 needs to know where a wrong start value ends
                seen |= rflag;
                if (CHUNK == 2) {
                    ch_flags[gi] = (uint8_t)rflag;
                } else {
                    const uint32_t slot = (uint32_t)gi & 7;
                    fb |= (uint64_t)rflag << (slot * 8);
                    if (slot == 0 || xi == 0) {
                        const uint64_t g8 = gi & ~7ull;
                        if (g8 >= base && g8 + 7 < base + m) {
                            *reinterpret_cast<uint64_t*>(ch_flags + g8) = fb;
                        } else {
#pragma unroll
                            for (int t = 0; t < 8; ++t)
                                if (g8 + t >= gi && g8 + t < base + m) ch_flags[g8 + t] = (uint8_t)(fb >> (t * 8));
                        }
                        fb = 0;
                    }
                }
            }
            if (CHUNK == 2) {
                // pass 2 writes over the speculative results one value at a time (a few steps per chunk)
                if (MODE == SPX_MODE_PML) {
                    if (NARROW)
                        reinterpret_cast<uint16_t*>(b.out_lengths)[gi] = (uint16_t)length;
                    else
                        b.out_lengths[gi] = length;
                } else {
                    b.out_pointers[gi] = sample;
                }
                if (DOC) {
                    if (NARROW)
                        reinterpret_cast<uint16_t*>(b.out_docs)[gi] = (uint16_t)doc;
                    else
                        b.out_docs[gi] = doc;
                }
            } else if (MODE == SPX_MODE_PML) {
                // lengths[m-i-1] = length   (:281)
                // (out_lengths == NULL: classification only -- the walk then runs at the gather ceiling)
                if (CHUNK == 0) {
                    if (b.len_mask != nullptr) {
                        const uint64_t bit = (uint64_t)(length == 0) << (xi & 63);
                        mbits |= (xi & 64) ? 0 : bit;
                        mbits_hi |= (xi & 64) ? bit : 0;
                        if ((xi & 127) == 0) {
                            const uint64_t pair = ((base - base0) >> 7) + rd + (xi >> 7);
                            if (pair < b.len_mask_pairs)
                                *reinterpret_cast<P64x2*>(b.len_mask + 2 * pair) = P64x2{mbits, mbits_hi};
                            else
                                n_err++;  // the batch holds more characters than the caller's total_chars
                            mbits = 0;
                            mbits_hi = 0;
                        }
                    }
                } else if (NARROW || m < 65536)
                    stage_n<LEN_G, NARROW>(obn, length, b.out_lengths, base, xi, m);
                else
                    b.out_lengths[gi] = length;
            } else {
                // ms_pointers[m-i-1] = sample   (:618), staged PTR_G at a time: elements [xi & ~(PTR_G-1), ...] of the read
                const uint32_t slot = xi & (PTR_G - 1);
#pragma unroll
                for (int j = 0; j < PTR_G; ++j) pbs[j] = slot == (uint32_t)j ? sample : pbs[j];
                if (slot == 0) {
                    uint64_t* o = b.out_pointers + gi;
                    const uint32_t cnt = m - xi;  // >= PTR_G for every group but the read's top one
#pragma unroll
                    for (int q = 0; q < PTR_G / 2; ++q) {
                        if (cnt >= 2u * q + 2)
                            *reinterpret_cast<P64x2*>(o + 2 * q) = P64x2{pbs[2 * q], pbs[2 * q + 1]};
                        else if (cnt == 2u * q + 1)
                            o[2 * q] = pbs[2 * q];
                    }
                }
            }
            if (DOC && CHUNK != 2) {  // doc_nums[m-i-1] = curr_doc_id   (:336 / :677); ids < 65536
                if (NARROW || m < 65536)
                    stage_n<DOC_G, NARROW>(dbn, doc, b.out_docs, base, xi, m);
                else
                    b.out_docs[gi] = doc;
            }
            if (want_class) {
                if (xi < bin_lo) {  // crossed into the previous bin (descending index)
                    if (bin_max >= b.max_value_thr)
                        above++;
                    else
                        below++;
                    sum_max += bin_max;
                    bin_max = 0;
                    bin_lo -= (uint32_t)b.bin_width;
                }
                bin_max = length > bin_max ? length : bin_max;
            }
            n_steps++;
            x = xi;
            if (x == 0) {
                if (want_class) {
                    if (bin_max >= b.max_value_thr)
                        above++;
                    else
                        below++;
                    sum_max += bin_max;
                    b.out_class[rd] = spx_class{sum_max, above, below};
                }
                if (CHUNK == 1)  // what a walk entering the chunk below starts from
                    b.ch.ends[rd] = WalkState{k0, length, (offp == OFF_LAST) ? OFF_END : (uint64_t)offp, sample, doc, seen};
                if (CHUNK == 2) {
                    // reached the chunk's first character without meeting the speculative walk: fine for a
                    // read's first chunk (the read is done), otherwise the read is walked again plainly
                    SeamRec sr;
                    sr.t = base;
                    sr.met = b.ch.round << 8;
                    sr.reset_above = seen;
                    sr.ext = WalkState{k0, length, (offp == OFF_LAST) ? OFF_END : (uint64_t)offp, sample, doc, seen};
                    sr.spec_length = 0;
                    sr.spec_doc = 0;
                    sr.spec_sample = 0;
                    b.ch.seams[rd] = sr;
                    // (k_chunk_scan has the chunk below entered with this state in the next round)
                    (void)item_rd;
                }
                NEXT_ITEM();
            } else {
                ph = P_LAND;
                if (peek) {
                    // The jump row told us the head of the run we land in.  If the next character
                    // is present in the index and differs from it, the next step is a jump again
                    // -- from (k0, offp), which is all a jump needs -- and the landing row is
                    // never fetched.  (Match / byte >= 128 / absent letter: load the row as usual.)
                    const uint64_t g = base + x - 1;
                    if (g >= wbase && g - wbase < 32) {
                        const uint32_t wi = (uint32_t)(g - wbase);
                        const uint32_t cn = WIN_CHAR(wi);
                        const LetterInfo li = s_let[cn];
                        if (li.qbeg != li.qend && cn != Hland && k0 < R) {
                            k = k0;
                            off = offp;
                            H_k = Hland;
                            c = cn;
                            quirk = false;
                            qbeg = li.qbeg;
                            qend = li.qend;
                            fadd = 0;
                            n_jumps++;
                            ph = P_FAT;
                        }
                    }
                }
            }
            peek = false;
            if (CHUNK && x != 0 && ((base + x) & ((1u << CKPT_SHIFT) - 1)) == 0) {
                // checkpoint: the state before character base + x - 1
                if (CHUNK == 1) {
                    ch_ckpt[(base + x) >> CKPT_SHIFT] =
                        WalkState{k0, length, (offp == OFF_LAST) ? OFF_END : (uint64_t)offp, sample, doc, seen};
                } else {
                    ph_after = ph;
                    ph = P_CKPT;
                }
            }
        }
        // One flat loop, one back edge.  Hiding the phase from the optimiser here keeps it from
        // threading the "read finished" path into a back edge of its own and splitting the loop
        // into an outer per-read and an inner per-character loop -- in which every lane waits at
        // the end of its read for the slowest lane of the wavefront (seen once: 19.3 ms vs 11 ms).
        asm volatile("" : "+v"(ph));
    }

    if (CHUNK == 2) {  // characters walked a second time: not steps of the batch
        atomicAdd(&b.counters->reserved0, (unsigned long long)n_steps);
    } else {
        atomicAdd(&b.counters->steps, (unsigned long long)n_steps);
        atomicAdd(&b.counters->jumps, (unsigned long long)n_jumps);
        atomicAdd(&b.counters->pred_jumps, (unsigned long long)n_pred);
    }
    atomicAdd(&b.counters->row_loads, (unsigned long long)n_rows);
    atomicAdd(&b.counters->dir_loads, (unsigned long long)n_dir);
    if (n_err) atomicAdd(&b.counters->error, (unsigned long long)n_err);
}

#include "spx_walk_fast.inc"

// ---------------------------------------------------------------------------
// MS length extension: ms_t::matching_statistics second loop
// (compute_ms_pml.cpp:800-810) with plain text in HBM instead of the SLP.
// One lane per read; `l` is carried exactly like the reference; characters are
// compared eight at a time (two aligned 64-bit loads + funnel shift per side).
// ---------------------------------------------------------------------------
// ascending counterpart of stage_n (the extension walks a read left to right): elements [i & ~(G-1), ...] of the
// read as u16 in G / 4 registers, written as 32-bit values when the group's highest element -- or the read's
// last -- has been produced
template <int G, bool NARROW_OUT>
__device__ __forceinline__ void stage_up(StageN<G>& st, uint32_t value, uint32_t* out, uint64_t base, uint32_t i,
                                         bool last) {
    const uint32_t idx = i & (G - 1), r = idx >> 2;
    const uint64_t v = (uint64_t)(value & 0xffffu) << ((idx & 3) * 16);
#pragma unroll
    for (int j = 0; j < G / 4; ++j) st.a[j] |= (r == (uint32_t)j) ? v : 0ull;
    if (idx == G - 1 || last) {
        uint32_t* o = out + base + (i - idx);
        const uint32_t cnt = idx + 1;
#pragma unroll
        for (int q = 0; q < G / 8; ++q) {
            const uint64_t lo = st.a[2 * q], hi = st.a[2 * q + 1];
            if (cnt > 8u * q) {
                const uint32_t c = cnt - 8 * q;
                if (NARROW_OUT) {
                    uint16_t* p = reinterpret_cast<uint16_t*>(out) + base + (i - idx) + 8 * q;
                    if (c >= 8) {
                        *reinterpret_cast<H16x8*>(p) = H16x8{lo, hi};
                    } else {
                        uint64_t src = lo;
                        if (c & 4) {
                            *reinterpret_cast<H16x4*>(p) = H16x4{lo};
                            src = hi;
                            p += 4;
                        }
                        if (c & 2) {
                            *reinterpret_cast<H16x2*>(p) = H16x2{(uint32_t)src};
                            src >>= 32;
                            p += 2;
                        }
                        if (c & 1) *p = (uint16_t)src;
                    }
                    continue;
                }
                uint32_t* p = o + 8 * q;
                if (c >= 8) {
                    *reinterpret_cast<U32x4*>(p) = widen4(lo);
                    *reinterpret_cast<U32x4*>(p + 4) = widen4(hi);
                } else {
                    uint64_t src = lo;
                    if (c & 4) {
                        *reinterpret_cast<U32x4*>(p) = widen4(lo);
                        src = hi;
                        p += 4;
                    }
                    if (c & 2) {
                        *reinterpret_cast<U32x2*>(p) = U32x2{(uint32_t)src & 0xffff, (uint32_t)(src >> 16) & 0xffff};
                        src >>= 32;
                        p += 2;
                    }
                    if (c & 1) *p = (uint32_t)src & 0xffff;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < G / 4; ++j) st.a[j] = 0;
    }
}

// eight bytes at any byte address: ONE global_load_dwordx2 (gfx950 under HSA takes unaligned
// vector loads; two aligned loads + a funnel shift doubled the kernel's lane loads, which is
// what it is bound by)
__device__ __forceinline__ uint64_t load8_unaligned(const uint8_t* base, uint64_t idx) {
    uint64_t v;
    __builtin_memcpy(&v, base + idx, 8);
    return v;
}

// Pointers reach the lanes through LDS.  A lane that reads its own read's pointers straight from memory touches
// one 128-byte line per load and uses 16 bytes of it before the other wavefronts of the CU have pushed the line out
// of the caches again: the pointers alone moved 8 x their size.  Here a wavefront loads a tile of EXT_PT pointers
// for each of its 64 reads cooperatively -- 16 consecutive lanes take 16 consecutive pointers of one read, one
// line -- and every lane then takes its own column of the tile from LDS.
constexpr int EXT_TPB = 64;  // one wavefront per block: __syncthreads() is a wavefront barrier
#ifndef SPX_EXT_PT
#define SPX_EXT_PT 16
#endif
#ifndef SPX_EXT_WAVES
#define SPX_EXT_WAVES 1
#endif
constexpr int EXT_PT = SPX_EXT_PT;   // pointers per read and tile

__global__ void __launch_bounds__(EXT_TPB, SPX_EXT_WAVES) k_ms_extend(const DevIndex ix, const BatchArgs b) {
    __shared__ uint64_t s_ptr[EXT_PT][EXT_TPB + 1];
    __shared__ uint64_t s_at[EXT_TPB];   // where each lane's tile starts in out_pointers
    __shared__ uint32_t s_cnt[EXT_TPB];  // pointers of the tile that exist
    const uint32_t lane = threadIdx.x;
    const uint64_t rd = blockIdx.x * (uint64_t)EXT_TPB + lane;
    const bool live = rd < b.nreads;
    const uint64_t base = live ? b.offs[rd] : 0;
    const uint64_t m = live ? b.offs[rd + 1] - base : 0;
    const uint8_t* text = ix.text;
    const uint64_t n = ix.n_text;
    const bool want_class = b.out_class != nullptr;
    uint64_t l = 0, prev = 0;
    StageN<LEN_G> obu{};  // staged lengths
    const bool staged = m < 65536;
    // classifier over ascending indices
    const uint64_t w = b.bin_width ? b.bin_width : 1;
    const uint64_t nb = m / w > 0 ? m / w : 1;
    uint64_t bin_hi = nb > 1 ? w : m;  // exclusive end of the current bin
    uint64_t bin_idx = 0;
    uint32_t bin_max = 0, above = 0, below = 0;
    uint64_t sum_max = 0;
    uint64_t mmax = m;  // the wavefront's longest read
    for (int sft = 32; sft > 0; sft >>= 1) {
        const uint64_t o = __shfl_xor(mmax, sft);
        mmax = o > mmax ? o : mmax;
    }
    for (uint64_t i0 = 0; i0 < mmax; i0 += EXT_PT) {
        __syncthreads();  // the tile before this one has been used up
        s_at[lane] = base + i0;
        s_cnt[lane] = i0 < m ? (uint32_t)(m - i0 < EXT_PT ? m - i0 : EXT_PT) : 0u;
        __syncthreads();
#pragma unroll
        for (int p = 0; p < EXT_TPB * EXT_PT / EXT_TPB; ++p) {  // 16 passes of 4 reads x 16 pointers
            const uint32_t R = (uint32_t)p * (EXT_TPB / EXT_PT) + lane / EXT_PT, j = lane & (EXT_PT - 1);
            if (j < s_cnt[R]) s_ptr[j][R] = b.out_pointers[s_at[R] + j];
        }
        __syncthreads();
        const uint32_t cnt = s_cnt[lane];
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint64_t i = i0 + j, gi = base + i;
            const uint64_t pos = s_ptr[j][lane];
            const bool cont = (i >= 1) && (pos == prev + 1);  // (i < 1 || pos != pointers[i-1] + 1)
            if (!cont) {
                // while (i+l < m && pos+l < n && read[i+l] == text[pos+l]) ++l;   unsigned arithmetic
                // as upstream: pos + l may wrap for wrapped pointers (Appendix C3)
                for (;;) {
                    const uint64_t ti = pos + l;
                    if (i + l >= m || ti >= n) break;
                    uint64_t lim = m - (i + l);
                    if (n - ti < lim) lim = n - ti;
                    const uint64_t x = load8_unaligned(b.seqs, base + i + l) ^ load8_unaligned(text, ti);
                    const uint64_t eq = x ? (uint64_t)(__builtin_ctzll(x) >> 3) : 8;  // equal leading bytes
                    const uint64_t adv = eq < lim ? eq : lim;  // never past the end of the read / text
                    l += adv;
                    if (adv < 8) break;  // mismatch, or an end reached inside this word
                }
            }
            if (b.narrow)  // (m < 65536: the walk refused the batch otherwise)
                stage_up<LEN_G, true>(obu, (uint32_t)l, b.out_lengths, base, (uint32_t)i, i + 1 == m);
            else if (staged)
                stage_up<LEN_G, false>(obu, (uint32_t)l, b.out_lengths, base, (uint32_t)i, i + 1 == m);
            else
                b.out_lengths[gi] = (uint32_t)l;
            if (want_class) {
                if (i >= bin_hi) {
                    if (bin_max >= b.max_value_thr)
                        above++;
                    else
                        below++;
                    sum_max += bin_max;
                    bin_max = 0;
                    bin_idx++;
                    bin_hi = (bin_idx + 1 == nb) ? m : bin_hi + w;
                }
                bin_max = (uint32_t)l > bin_max ? (uint32_t)l : bin_max;
            }
            l = (l == 0 ? 0 : (l - 1));
            prev = pos;
        }
    }
    if (want_class && live) {
        if (m > 0) {
            if (bin_max >= b.max_value_thr)
                above++;
            else
                below++;
            sum_max += bin_max;
        }
        b.out_class[rd] = spx_class{sum_max, above, below};
    }
}

// spx_index_set_text: the text against the index.  The BWT character at the start of run k is the
// text character in front of that suffix, i.e. text[samples_start[k]] (the stored sample is
// SA - 1, compute_ms_pml.cpp:433; n - 1 marks the terminator, which the text does not hold).
__global__ void k_text_check(const DevIndex ix, unsigned long long* bad) {
    const uint64_t k = blockIdx.x * (uint64_t)WALK_TPB + threadIdx.x;
    if (k >= ix.r) return;
    const Row row = row_at(ix, k);
    const uint32_t H = ix.compact ? crow_H(row) : row_H(row);
    const uint64_t s = ix.ss_by_run[k];
    const bool ok = s < ix.n_text ? ix.text[s] == H : (s == ix.n - 1 && H <= 1);
    if (!ok) atomicAdd(bad, 1ull);
}

// items: reads (CHUNK == 0) or an upper bound of the chunks (the kernel reads the real count from
// device memory)
// The indexed text from the MS index itself ("by parallel BWT inversion from SA samples"): ms_t reads
// the text through an SLP (compute_ms_pml.cpp:769-774, 805); here it is plain text in HBM, and when no text file
// is at hand it is rebuilt from the index.  The BWT character at position p is text[SA[p] - 1]; samples_start[k]
// is that text position for the first position of run k, and every LF step moves one text position to the left.
// One lane per run walks LF from the run's first position, writing the head of the run it is in, until it lands
// on the first position of a run (whose own lane goes on from there).  The chains partition the BWT: n steps in
// all, n / r per lane on average.
__global__ void k_text_from_index(const DevIndex ix, uint8_t* text, uint64_t n_text, unsigned long long* stuck) {
    const uint64_t k_start = blockIdx.x * (uint64_t)WALK_TPB + threadIdx.x;
    if (k_start >= ix.r) return;
    if (ix.compact && crow_cont(row_at(ix, k_start))) return;  // a later piece of a long run: not a run's first position
    uint64_t t = ix.ss_by_run[k_start];
    uint32_t k = (uint32_t)k_start;
    uint64_t off = 0;
    for (uint64_t guard = 0;; ++guard) {
        const Row row = row_at(ix, k);
        const uint32_t H = ix.compact ? crow_H(row) : row_H(row);
        if (t < n_text) text[t] = (uint8_t)H;
        // LF of (k, off): run LFrun at offset LFoff + off, or a later run when that overshoots
        uint32_t k0 = ix.compact ? crow_LFrun(row) : row_LFrun(row);
        uint64_t offp = (ix.compact ? (uint64_t)crow_LFoff(row) : row_LFoff(row)) + off;
        for (;;) {
            const Row r0 = row_at(ix, k0);
            const uint64_t len = ix.compact ? (uint64_t)crow_len(r0) : row_len(r0);
            if (offp < len || k0 >= ix.r) break;
            offp -= len;
            k0++;
        }
        // the first position of a run: its own lane takes over (the first position of a later PIECE of a long run is
        // not one: no sample names its text position, the chain goes on through it)
        if ((offp == 0 && !(ix.compact && crow_cont(row_at(ix, k0)))) || k0 >= ix.r) break;
        if (guard > ix.n) {                  // not a permutation: corrupt run structure
            atomicAdd(stuck, 1ull);
            break;
        }
        k = k0;
        off = offp;
        t -= 1;
    }
}

template <int MODE, bool DOC, bool COMPACT, bool NARROW, int CHUNK = 0>
int launch_lanes(spx_index* ix, const BatchArgs& args, hipStream_t stream, uint64_t items = 0, bool* wrote_lengths = nullptr) {
    if (CHUNK == 0) items = args.nreads;
    if (CHUNK == 0 && MODE == SPX_MODE_PML && args.out_lengths != nullptr && args.len_mask == nullptr) {
        set_error("internal: PML walk without prepare_len_mask");
        return SPX_E_ARG;
    }
    // the plain walk over compact rows has a body of its own (spx_walk_fast.inc); SPX_OLD_WALK=1 keeps the state
    // machine for it too (A/B runs, and the tests that hold the two against each other)
    static const bool old_walk = getenv("SPX_OLD_WALK") != nullptr;
    // k_walk_fast also walks passes 1 and 2 of the chunked walk (SPX_PASS2_LANES=1: pass 2 on the state machine, for A/B runs)
    static const bool pass2_lanes = getenv("SPX_PASS2_LANES") != nullptr;
    constexpr int FCHUNK = CHUNK;
    const bool fast = COMPACT && !(CHUNK == 2 && pass2_lanes) && args.only_flagged == nullptr && !old_walk && items < (1ull << 31);
    if (args.in_starts != nullptr && !(fast && CHUNK == 0)) {
        set_error("internal: parked reads (BatchArgs::in_starts) are taken by the plain k_walk_fast only");
        return SPX_E_ARG;
    }
    // resident blocks per CU and CU count are looked up once per index and kernel variant
    const int slot = (fast ? 4 : 0) + MODE * 2 + (DOC ? 1 : 0);
    if (ix->occ_blocks[slot] == 0) {
        int occ = 0;
        if (fast)
            SPX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_walk_fast<MODE, DOC, NARROW, 0>, WALK_TPB, 0));
        else
            SPX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_walk_lanes<MODE, DOC, COMPACT, NARROW, 0>, WALK_TPB, 0));
        ix->occ_blocks[slot] = occ < 1 ? 1 : occ;
    }
    if (ix->num_cus == 0) {
        hipDeviceProp_t prop;
        SPX_HIP(hipGetDeviceProperties(&prop, ix->device));
        ix->num_cus = prop.multiProcessorCount;
    }
    // Occupancy target: 20 waves per CU (five per SIMD, what the kernel's registers allow).
    // Measured on C3 (tools/ab.sh): 911 / 1001 / 1026 M reads/s at 12 / 16 / 20 waves per CU.
    // (With 32-byte fat rows -- two 16-byte lane loads per jump -- 12 was the optimum, 971 / 935 /
    // 914: the kernel sits at the chip's rate of ~80 G 16-byte lane loads per second, and more
    // chains only lengthened the queue.)
    // Match-heavy walks over small alphabets do best at 16 (DNA: 38.7 / 41.0 / 39.7 G steps/s); the
    // variants that also fetch the aux record per jump like 20 too (MS+doc 27.6 / 28.0 / 28.4,
    // PML+doc 30.3 / 32.2 / 33.0).
    int occ = ix->occ_blocks[slot];
    // Round 4, k_walk_fast (tools/ab.sh --waves-per-cu, profiles/r04_occupancy.txt): the headline batch 924 / 977 / 946 M
    // reads/s at 12 / 16 / 20; the 1.25 M-read share of an 8-GPU strong cut 1.375 / 1.396 / 1.448 ms; the variants with
    // document ids or MS pointers (whose LDS staging allows 12 at most for MS) 7.5 / 8.1 / 7.9 ms at 12 / 16 / 20.
    constexpr bool SIDE = (MODE == SPX_MODE_MS) || DOC;
    const int auto_waves = ix->view.nletters > 16 ? (fast ? (SIDE ? 12 : 16) : 20) : (fast && SIDE ? 12 : 16);
    const int target_waves = ix->waves_per_cu > 0 ? ix->waves_per_cu : auto_waves;
    int want = target_waves / (WALK_TPB / 64);
    if (want < 1) want = 1;
    if (want < occ) occ = want;
    unsigned tpb = WALK_TPB;
    uint64_t grid = (uint64_t)occ * ix->num_cus;
    const uint64_t need = (items + WALK_TPB - 1) / WALK_TPB;
    BatchArgs a = args;
    a.lanes_per_wave = 64;
    bool small_batch = false;  // fewer items than resident lanes
    if (ix->force_lanes_per_wave > 0) {
        // experiment knob: 1 = the "one wavefront owns one read" mapping
        const uint64_t lpw = (uint64_t)(ix->force_lanes_per_wave > 64 ? 64 : ix->force_lanes_per_wave);
        a.lanes_per_wave = (uint32_t)lpw;
        const uint64_t waves_needed = (items + lpw - 1) / lpw;
        const uint64_t waves_resident = grid * (WALK_TPB / 64);
        const uint64_t waves = waves_needed < waves_resident ? waves_needed : waves_resident;
        grid = (waves + (WALK_TPB / 64) - 1) / (WALK_TPB / 64);
    } else if (need < grid) {
        // Fewer reads than lanes (long-read batches): the walk of a read is one dependent
        // chain, so the batch is latency-bound.  Spread it: 64-thread blocks, one or more per
        // SIMD, and only as many active lanes per wavefront as needed -- a wavefront whose few
        // lanes sit in the same phase issues a fraction of the instructions per iteration.
        tpb = 64;
        small_batch = true;
        // (two per SIMD: 1.12 -> 1.10 ms for the 6 250-read share at r = 2e9, 0.87 -> 0.85 at r = 2^27; three and four are
        // slower again -- profiles/r04_c5_passes.txt)
        static const int spread = getenv("SPX_SPREAD_WAVES") ? atoi(getenv("SPX_SPREAD_WAVES")) : 2;
        const uint64_t waves = (uint64_t)ix->num_cus * 4 * (uint64_t)(spread > 0 ? spread : 1);  // wavefronts per SIMD to fill
        uint64_t lpw = (items + waves - 1) / waves;
        if (lpw < 1) lpw = 1;
        if (lpw > 64) lpw = 64;
        a.lanes_per_wave = (uint32_t)lpw;
        grid = (items + lpw - 1) / lpw;
        if (spread > 1) {  // (blocks of four wavefronts: a 64-thread block costs the LDS of a 256-thread one, 13 fit a CU)
            tpb = WALK_TPB;
            grid = (items + lpw * 4 - 1) / (lpw * 4);
        }
    }
    // Reads are dealt by striding (rd += lanes): when the batch is a small multiple of the lanes, the last round leaves most of
    // them idle while a few finish (1.25 M reads on 262 144 lanes: 4.77 reads per lane, a fifth round for three lanes in four).
    // The grid is cut to the blocks that give every lane the same number of reads.
    if (ix->force_lanes_per_wave == 0 && need >= grid && getenv("SPX_NO_EVEN_GRID") == nullptr) {
        const uint64_t lanes_all = grid * WALK_TPB;
        const uint64_t rounds = (items + lanes_all - 1) / lanes_all;
        const uint64_t even = (items + rounds * WALK_TPB - 1) / (rounds * WALK_TPB);
        // (a few rounds only: with dozens of them the uneven last one is noise and every resident block counts --
        // the headline batch, 39 rounds, ran 2-3 % slower on 1002 blocks than on 1024)
        if (rounds <= 8 && even < grid && even * 8 >= grid * 7) grid = even;
    }
    if (grid == 0) grid = 1;
    if (fast) {
        // (a launch that does not fill the chip, and pass 2 -- a few characters per chunk, then a long tail -- are latency
        // chains: they take the body that issues its gather early; SPX_EARLY_GATHER=0 / 1 forces either, for A/B runs)
        static const int early_env = getenv("SPX_EARLY_GATHER") ? atoi(getenv("SPX_EARLY_GATHER")) : -1;
        const bool early = early_env >= 0 ? early_env != 0 : (FCHUNK == 2 || small_batch);
        if (early)
            k_walk_fast<MODE, DOC, NARROW, FCHUNK, true><<<(unsigned)grid, tpb, 0, stream>>>(ix->view, a);
        else
            k_walk_fast<MODE, DOC, NARROW, FCHUNK, false><<<(unsigned)grid, tpb, 0, stream>>>(ix->view, a);
        if (wrote_lengths) *wrote_lengths = MODE == SPX_MODE_PML;  // no bit mask, no expansion kernel
    } else
        k_walk_lanes<MODE, DOC, COMPACT, NARROW, CHUNK><<<(unsigned)grid, tpb, 0, stream>>>(ix->view, a);
    SPX_HIP(hipGetLastError());
    return SPX_OK;
}


// ---------------------------------------------------------------------------
// chunked walk of long-read batches (spx_internal.h): preparation, pass 3, classifier
// ---------------------------------------------------------------------------
__global__ void k_chunk_count(const uint64_t* offs, uint64_t nreads, uint32_t L, int narrow, uint64_t* cnt,
                              uint32_t* read_fail, WalkCounters* counters) {
    const uint64_t q = blockIdx.x * (uint64_t)WALK_TPB + threadIdx.x;
    if (q > nreads) return;
    if (q == nreads) {
        cnt[q] = 0;  // the scan then leaves the number of chunks at chunk_start[nreads]
        return;
    }
    const uint64_t base = offs[q], m = offs[q + 1] - base;
    cnt[q] = m ? (base + m - 1) / L - base / L + 1 : 0;
    read_fail[q] = 0;
    if (narrow && m >= 65536) atomicAdd(&counters->error, 1ull);  // 16-bit outputs cannot hold this read's values
}

__global__ void k_chunk_fill(const uint64_t* offs, uint64_t nreads, uint32_t L, const uint64_t* chunk_start,
                             ChunkDesc* desc, uint64_t* nchunks) {
    const uint64_t q = blockIdx.x * (uint64_t)WALK_TPB + threadIdx.x;
    if (q >= nreads) return;
    if (q == 0) *nchunks = chunk_start[nreads];
    const uint64_t base = offs[q], end = offs[q + 1];
    uint64_t c = chunk_start[q];
    for (uint64_t a = base; a < end; ++c) {
        uint64_t e = (a / L + 1) * L;
        if (e > end) e = end;
        ChunkDesc d;
        d.gend = e;
        d.len = (uint32_t)(e - a) | (e == end ? CHUNK_TOP : 0u) | (a == base ? CHUNK_BOTTOM : 0u);
        d.rd = (uint32_t)q;
        desc[c] = d;
        a = e;
    }
}

// Between rounds of pass 2, one lane per read: follow the chain of seams down from where the truth is
// known to reach (vtop).  The walk that entered chunk j - 1 in its latest run started from the true end
// state of chunk j (invariant), so chunk j - 1 holds the truth; if its seam closed, the recorded end
// state of chunk j - 1 is the true one and the chain goes on with the round-1 walk of chunk j - 2; if
// not, chunk j - 2 is entered again in the next round with the state the walk through chunk j - 1 ended
// in.  After the last round a read whose chain has not reached its first chunk is marked for the plain walk.
__global__ void k_chunk_scan(ChunkArgs ch, uint64_t nreads, uint32_t round, uint32_t last, WalkCounters* counters) {
    const uint64_t q = blockIdx.x * (uint64_t)WALK_TPB + threadIdx.x;
    if (q >= nreads) return;
    const uint64_t cs = ch.chunk_start[q], ce = ch.chunk_start[q + 1];
    if (ce - cs < 2) return;
    ChunkDesc* desc = const_cast<ChunkDesc*>(ch.desc);
    uint64_t j = round == 1 ? ce - 1 : ch.vtop[q];
    if (j == cs) return;  // finished in an earlier round
    if (round > 1) desc[j - 1].len &= ~CHUNK_ACTIVE;  // the chunk this round entered again
    while (j > cs) {
        const uint64_t nx = j - 1;
        const SeamRec sr = ch.seams[nx];
        j = nx;  // chunk nx holds the truth
        if (nx == cs || (sr.met & 1)) continue;
        if (!last) {  // open: the walk goes on into chunk nx - 1
            WalkState st = sr.ext;
            const ChunkDesc below = desc[nx - 1];
            const SeamRec old = ch.seams[nx - 1];
            st.flags = (uint32_t)(old.t - (below.gend - (below.len & CHUNK_LEN_MASK)));
            ch.reentry[nx - 1] = st;
            desc[nx - 1].len = below.len | CHUNK_ACTIVE;
            atomicAdd(&ch.pending[round + 1], 1u);
        }
        break;
    }
    ch.vtop[q] = j;
    if (last && j > cs) {
        ch.read_fail[q] = 1;
        atomicAdd(&counters->pad_, 1ull);  // reads walked again the plain way
    }
}

// Pass 3: per read, from its last chunk down, what is left of the wrong start values.
template <int MODE, bool DOC, bool NARROW>
__global__ void k_chunk_fix(const BatchArgs b) {
    const uint64_t q = blockIdx.x * (uint64_t)WALK_TPB + threadIdx.x;
    if (q >= b.nreads) return;
    const uint64_t cs = b.ch.chunk_start[q], ce = b.ch.chunk_start[q + 1];
    if (ce - cs < 2 || b.ch.read_fail[q]) return;
    uint16_t* const len16 = reinterpret_cast<uint16_t*>(b.out_lengths);
    uint16_t* const doc16 = reinterpret_cast<uint16_t*>(b.out_docs);
    const uint8_t* const ch_flags = b.ch.flags - (b.offs[0] & ~7ull);  // as in the walk: relative to the batch
    // corrections carried into the results of the walk that enters the next chunk down: its start
    // values were the recorded end values of the chunk above, which are off by this much
    bool c_on = false, cd_on = false;
    uint32_t c_len = 0, c_doc = 0;
    uint64_t c_smp = 0;
    auto patch = [&](uint64_t from, uint64_t to, bool do_cnt, uint32_t dl, uint64_t ds, bool do_doc, uint32_t dv,
                     bool& cnt_reset, bool& doc_reset) {
        // indices from-1 down to `to`: counters get their offset until the first step that reset them,
        // the document id its value until the first step that set it
        cnt_reset = doc_reset = false;
        uint64_t w8 = 0, have = ~0ull;  // flags eight at a time: the aligned group that holds index i
        for (uint64_t i = from; i-- > to;) {
            if ((i & ~7ull) != have) {
                have = i & ~7ull;
                w8 = *reinterpret_cast<const uint64_t*>(ch_flags + have);
            }
            const uint32_t f = (uint32_t)(w8 >> ((i & 7) * 8)) & 0xffu;
            if (!cnt_reset) {
                if (f & 1) {
                    cnt_reset = true;
                } else if (do_cnt) {
                    if (MODE == SPX_MODE_PML) {
                        if (NARROW)
                            len16[i] = (uint16_t)(len16[i] + dl);
                        else
                            b.out_lengths[i] += dl;
                    } else {
                        b.out_pointers[i] += ds;
                    }
                }
            }
            if (DOC && !doc_reset) {
                if (f & 2) {
                    doc_reset = true;
                } else if (do_doc) {
                    if (NARROW)
                        doc16[i] = (uint16_t)dv;
                    else
                        b.out_docs[i] = dv;
                }
            }
            if (cnt_reset && (doc_reset || !DOC)) break;
        }
    };
    for (uint64_t j = ce - 1; j-- > cs;) {
        const ChunkDesc d = b.ch.desc[j];
        const uint64_t B = d.gend, A = B - (d.len & CHUNK_LEN_MASK);
        const SeamRec sr = b.ch.seams[j];
        bool r_cnt, r_doc;
        // pass-2 results [t, B): started from the recorded end values of chunk j + 1
        if (c_on || cd_on) patch(B, sr.t, c_on, c_len, c_smp, cd_on, c_doc, r_cnt, r_doc);
        const bool e_cnt = sr.reset_above & 1, e_doc = (sr.reset_above & 2) != 0;
        const uint32_t L_true = sr.ext.length + ((c_on && !e_cnt) ? c_len : 0u);
        const uint64_t S_true = sr.ext.sample + ((c_on && !e_cnt) ? c_smp : 0ull);
        const uint32_t D_true = (cd_on && !e_doc) ? c_doc : sr.ext.doc;
        if (!(sr.met & 1)) {
            // the walk from above ran through the whole chunk: to the read's first character, or on into
            // the chunk below (a later round of pass 2 entered it with this walk's state)
            if (d.len & CHUNK_BOTTOM) break;
            c_on = c_on && !e_cnt;
            cd_on = cd_on && !e_doc;
            continue;
        }
        // speculative results [A, t): counters differ from the true ones by a constant up to the first reset
        const uint32_t dl = L_true - sr.spec_length;
        const uint64_t ds = S_true - sr.spec_sample;
        const bool fix_cnt = MODE == SPX_MODE_PML ? dl != 0 : ds != 0;
        const bool fix_doc = DOC && D_true != sr.spec_doc;
        patch(sr.t, A, fix_cnt, dl, ds, fix_doc, D_true, r_cnt, r_doc);
        c_on = !r_cnt && fix_cnt;
        c_len = dl;
        c_smp = ds;
        cd_on = DOC && !r_doc && fix_doc;
        c_doc = D_true;
    }
}

// Pass 3 by wavefronts: one wavefront per read, one lane per chunk (top down, 64 chunks at a time).  What a
// chunk hands to the chunk below it (k_chunk_fix's c_on / c_len / ...) is a function of what it was handed and of
// its own records; the hand-overs die at the first reset, so instead of following the chain chunk by chunk every
// lane computes its hand-over from its upper neighbour's, again and again until nothing changes (a chain that is
// k chunks long settles in k rounds; the usual one in two).  Then every lane patches its own chunk.
template <int MODE, bool DOC, bool NARROW>
__global__ void k_chunk_fix_wave(const BatchArgs b) {
    const uint64_t q = (blockIdx.x * (uint64_t)WALK_TPB + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (q >= b.nreads) return;
    const uint64_t cs = b.ch.chunk_start[q], ce = b.ch.chunk_start[q + 1];
    if (ce - cs < 2 || b.ch.read_fail[q]) return;
    uint16_t* const len16 = reinterpret_cast<uint16_t*>(b.out_lengths);
    uint16_t* const doc16 = reinterpret_cast<uint16_t*>(b.out_docs);
    const uint8_t* const ch_flags = b.ch.flags - (b.offs[0] & ~7ull);
    struct Carry {
        uint32_t on;  // bit 0: counters, bit 1: document id
        uint32_t len, doc;
        uint64_t smp;
    };
    auto same = [](const Carry& x, const Carry& y) { return x.on == y.on && x.len == y.len && x.doc == y.doc && x.smp == y.smp; };
    auto from_above = [&](const Carry& c, const Carry& first) {
        Carry u;
        u.on = __shfl_up(c.on, 1);
        u.len = __shfl_up(c.len, 1);
        u.doc = __shfl_up(c.doc, 1);
        u.smp = ((uint64_t)__shfl_up((uint32_t)(c.smp >> 32), 1) << 32) | __shfl_up((uint32_t)c.smp, 1);
        return lane == 0 ? first : u;
    };
    auto patch = [&](uint64_t from, uint64_t to, bool do_cnt, uint32_t dl, uint64_t ds, bool do_doc, uint32_t dv) {
        bool cnt_reset = false, doc_reset = !DOC;
        uint64_t w8 = 0, have = ~0ull;
        for (uint64_t i = from; i-- > to;) {
            if ((i & ~7ull) != have) {
                have = i & ~7ull;
                w8 = *reinterpret_cast<const uint64_t*>(ch_flags + have);
            }
            const uint32_t f = (uint32_t)(w8 >> ((i & 7) * 8)) & 0xffu;
            if (!cnt_reset) {
                if (f & 1) {
                    cnt_reset = true;
                } else if (do_cnt) {
                    if (MODE == SPX_MODE_PML) {
                        if (NARROW)
                            len16[i] = (uint16_t)(len16[i] + dl);
                        else
                            b.out_lengths[i] += dl;
                    } else {
                        b.out_pointers[i] += ds;
                    }
                }
            }
            if (DOC && !doc_reset) {
                if (f & 2) {
                    doc_reset = true;
                } else if (do_doc) {
                    if (NARROW)
                        doc16[i] = (uint16_t)dv;
                    else
                        b.out_docs[i] = dv;
                }
            }
            if ((cnt_reset || !do_cnt) && (doc_reset || !do_doc)) break;
        }
    };
    Carry first = {0, 0, 0, 0};  // handed to the tile's top chunk
    for (uint64_t top = ce - 1; top > cs;) {  // chunks top - 1 down to max(cs, top - 64)
        const uint64_t span = top - cs < 64 ? top - cs : 64;
        const bool mine = lane < span;
        const uint64_t j = top - 1 - (mine ? lane : 0);
        const ChunkDesc d = b.ch.desc[j];
        const SeamRec sr = b.ch.seams[j];
        const uint64_t B = d.gend, A = B - (d.len & CHUNK_LEN_MASK);
        const bool met = sr.met & 1, e_cnt = sr.reset_above & 1, e_doc = (sr.reset_above & 2) != 0;
        // is there a reset among the speculative results [A, t)?  (only asked of chunks whose seam closed)
        bool r_cnt = false, r_doc = !DOC;
        if (mine && met) {
            uint64_t w8 = 0, have = ~0ull;
            for (uint64_t i = sr.t; i-- > A;) {
                if ((i & ~7ull) != have) {
                    have = i & ~7ull;
                    w8 = *reinterpret_cast<const uint64_t*>(ch_flags + have);
                }
                const uint32_t f = (uint32_t)(w8 >> ((i & 7) * 8)) & 0xffu;
                r_cnt |= (f & 1) != 0;
                if (DOC) r_doc |= (f & 2) != 0;
                if (r_cnt && r_doc) break;
            }
        }
        uint32_t dl = 0, D_true = 0;
        uint64_t ds = 0;
        bool fix_cnt = false, fix_doc = false;
        auto hand_over = [&](const Carry& in) {
            const bool c_on = (in.on & 1) && !e_cnt, cd_on = (in.on & 2) && !e_doc;
            Carry out;
            if (!met) {
                out.on = (c_on ? 1u : 0u) | (cd_on ? 2u : 0u);
                out.len = in.len, out.smp = in.smp, out.doc = in.doc;
                fix_cnt = fix_doc = false;
                return out;
            }
            const uint32_t L_true = sr.ext.length + (c_on ? in.len : 0u);
            const uint64_t S_true = sr.ext.sample + (c_on ? in.smp : 0ull);
            D_true = cd_on ? in.doc : sr.ext.doc;
            dl = L_true - sr.spec_length;
            ds = S_true - sr.spec_sample;
            fix_cnt = MODE == SPX_MODE_PML ? dl != 0 : ds != 0;
            fix_doc = DOC && D_true != sr.spec_doc;
            out.on = ((!r_cnt && fix_cnt) ? 1u : 0u) | ((DOC && !r_doc && fix_doc) ? 2u : 0u);
            out.len = dl, out.smp = ds, out.doc = D_true;
            return out;
        };
        Carry in = first, out = hand_over(lane == 0 ? first : Carry{0, 0, 0, 0});
        for (int round = 0; round < 66; ++round) {
            in = from_above(out, first);
            const Carry nx = hand_over(in);
            const bool changed = mine && !same(nx, out);
            out = nx;
            if (!__any(changed)) break;
        }
        if (mine) {
            if (in.on) patch(B, sr.t, in.on & 1, in.len, in.smp, (in.on & 2) != 0, in.doc);  // pass-2 results [t, B)
            if (met && (fix_cnt || fix_doc)) patch(sr.t, A, fix_cnt, dl, ds, fix_doc, D_true);  // speculative results [A, t)
        }
        const int last = (int)span - 1;
        first.on = __shfl(out.on, last);
        first.len = __shfl(out.len, last);
        first.doc = __shfl(out.doc, last);
        first.smp = ((uint64_t)__shfl((uint32_t)(out.smp >> 32), last) << 32) | __shfl((uint32_t)out.smp, last);
        top -= span;
    }
}

// bin-max classifier over finished lengths (compute_ms_pml.cpp:969-995): one wavefront per read,
// one lane per bin
template <bool NARROW>
__global__ void k_classify_reads(const BatchArgs b) {
    const uint64_t q = (blockIdx.x * (uint64_t)WALK_TPB + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (q >= b.nreads) return;
    const uint64_t base = b.offs[q], m = b.offs[q + 1] - base;
    const uint64_t w = b.bin_width ? b.bin_width : 1;
    const uint64_t nb = m / w > 0 ? m / w : 1;
    const uint16_t* const len16 = reinterpret_cast<const uint16_t*>(b.out_lengths);
    uint32_t above = 0, below = 0;
    uint64_t sum = 0;
    if (m > 0)
        for (uint64_t bin = lane; bin < nb; bin += 64) {
            const uint64_t lo = bin * w, hi = (bin + 1 == nb) ? m : lo + w;
            uint32_t mx = 0;
#pragma unroll 8
            for (uint64_t i = lo; i < hi; ++i) {  // independent loads: eight in flight
                const uint32_t v = NARROW ? len16[base + i] : b.out_lengths[base + i];
                mx = v > mx ? v : mx;
            }
            if (mx >= b.max_value_thr)
                above++;
            else
                below++;
            sum += mx;
        }
    for (int s = 32; s > 0; s >>= 1) {
        above += __shfl_xor(above, s);
        below += __shfl_xor(below, s);
        sum += __shfl_xor(sum, s);
    }
    if (lane == 0) b.out_class[q] = spx_class{sum, above, below};
}

// The same classifier reading the lengths the way they lie: a wavefront per read goes over it in tiles of 64 x E
// consecutive values, a lane takes E of them in one 16-byte load (they span two bins at most: bin_width >= E), the
// bins' maxima are collected in LDS (ds_max), and the bins that end inside the tile are judged by one lane each;
// the tile's last bin goes on into the next tile.  (k_classify_reads gives a lane a bin: 300-byte strides, 14 of 64
// lanes busy for a 2 200-character read -- 1.4 TB/s; this one reads at the rate of a copy.)
template <bool NARROW>
__global__ void __launch_bounds__(WALK_TPB) k_classify_tiles(const BatchArgs b) {
    constexpr uint32_t E = NARROW ? 8 : 4;
    constexpr uint32_t TILE = 64 * E;
    __shared__ uint32_t s_bins[WALK_TPB / 64][TILE / E + 8];
    const uint64_t q = (blockIdx.x * (uint64_t)WALK_TPB + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (q >= b.nreads) return;
    uint32_t* const sb = s_bins[threadIdx.x >> 6];
    const uint64_t base = b.offs[q], m = b.offs[q + 1] - base, end_all = b.offs[b.nreads];
    const uint64_t w = b.bin_width ? b.bin_width : 1;
    const uint64_t nb = m / w > 0 ? m / w : 1;
    const uint8_t* const lens = reinterpret_cast<const uint8_t*>(b.out_lengths);
    auto bin_of = [&](uint64_t i) {  // min(i / w, nb - 1)
        const uint64_t d = w > 1 ? __umul64hi(i, b.bin_magic) : i;
        return d < nb - 1 ? d : nb - 1;
    };
    uint32_t above = 0, below = 0;
    uint64_t sum = 0;
    for (uint32_t t = lane; t < TILE / E + 8; t += 64) sb[t] = 0;
    for (uint64_t t0 = 0; t0 < m; t0 += TILE) {
        const uint64_t first_bin = bin_of(t0);
        const uint64_t t_end = t0 + TILE < m ? t0 + TILE : m;
        const uint64_t last_bin = bin_of(t_end - 1);
        const uint64_t i0 = t0 + (uint64_t)lane * E;
        if (i0 < m) {
            uint32_t v[E];
            const uint64_t gi = base + i0;
            if (gi + E <= end_all) {
                uint32_t raw[4];
                __builtin_memcpy(raw, lens + gi * (NARROW ? 2 : 4), 16);
#pragma unroll
                for (uint32_t e = 0; e < E; ++e) v[e] = NARROW ? (raw[e >> 1] >> ((e & 1) * 16)) & 0xffffu : raw[e];
            } else {
#pragma unroll
                for (uint32_t e = 0; e < E; ++e)
                    v[e] = gi + e < end_all ? (NARROW ? (uint32_t)reinterpret_cast<const uint16_t*>(lens)[gi + e]
                                                      : reinterpret_cast<const uint32_t*>(lens)[gi + e])
                                            : 0u;
            }
            const uint64_t ba = bin_of(i0);
            const uint64_t cut = ba + 1 < nb ? (ba + 1) * w : ~0ull;  // first index of the next bin (the last bin takes the rest)
            uint32_t ma = 0, mb = 0;
            bool any_b = false;
#pragma unroll
            for (uint32_t e = 0; e < E; ++e) {
                const uint64_t i = i0 + e;
                if (i < m) {
                    if (i < cut) {
                        ma = v[e] > ma ? v[e] : ma;
                    } else {
                        mb = v[e] > mb ? v[e] : mb;
                        any_b = true;
                    }
                }
            }
            atomicMax(&sb[ba - first_bin], ma);
            if (any_b) atomicMax(&sb[ba + 1 - first_bin], mb);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // the bins that ended inside the tile: all below the bin the next tile starts in (all of them when the read ends here)
        const bool read_ends = t_end == m;
        const uint64_t next_first = read_ends ? last_bin + 1 : bin_of(t_end);
        const uint64_t done = next_first - first_bin;
        const uint32_t carry = next_first == last_bin ? sb[last_bin - first_bin] : 0u;
        for (uint64_t t = lane; t < done; t += 64) {
            const uint32_t mx = sb[t];
            if (mx >= b.max_value_thr)
                above++;
            else
                below++;
            sum += mx;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint64_t t = lane; t <= last_bin - first_bin; t += 64) sb[t] = 0;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) sb[0] = carry;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    for (int sft = 32; sft > 0; sft >>= 1) {
        above += __shfl_xor(above, sft);
        below += __shfl_xor(below, sft);
        sum += __shfl_xor(sum, sft);
    }
    if (lane == 0) b.out_class[q] = spx_class{sum, above, below};
}

template <int MODE, bool DOC, bool NARROW>
int run_chunked(spx_index* ix, BatchArgs a, uint64_t bound, hipStream_t stream) {
    int rc;
    if ((rc = launch_lanes<MODE, DOC, true, NARROW, 1>(ix, a, stream, bound)) != SPX_OK) return rc;
    for (int round = 1; round <= 1 + CHUNK_EXTRA_ROUNDS; ++round) {
        a.ch.round = (uint32_t)round;
        a.ch.last_round = round == 1 + CHUNK_EXTRA_ROUNDS;
        if ((rc = launch_lanes<MODE, DOC, true, NARROW, 2>(ix, a, stream, bound)) != SPX_OK) return rc;
        k_chunk_scan<<<(unsigned)((a.nreads + WALK_TPB - 1) / WALK_TPB), WALK_TPB, 0, stream>>>(
            a.ch, a.nreads, (uint32_t)round, a.ch.last_round, a.counters);
        SPX_HIP(hipGetLastError());
    }
    static const bool fix_by_lanes = getenv("SPX_CHUNK_FIX_LANES") != nullptr;  // the round-3 kernel, for A/B
    if (fix_by_lanes) {
        const unsigned grid = (unsigned)((a.nreads + WALK_TPB - 1) / WALK_TPB);
        k_chunk_fix<MODE, DOC, NARROW><<<grid, WALK_TPB, 0, stream>>>(a);
    } else {
        const unsigned grid = (unsigned)((a.nreads * 64 + WALK_TPB - 1) / WALK_TPB);
        k_chunk_fix_wave<MODE, DOC, NARROW><<<grid, WALK_TPB, 0, stream>>>(a);
    }
    SPX_HIP(hipGetLastError());
    if (MODE == SPX_MODE_PML && a.out_class != nullptr) {
        const unsigned cgrid = (unsigned)((a.nreads * 64 + WALK_TPB - 1) / WALK_TPB);
        static const bool by_bins = getenv("SPX_CLASSIFY_BY_BINS") != nullptr;  // the round-2 kernel, for A/B
        if (a.bin_width >= 8 && !by_bins)
            k_classify_tiles<NARROW><<<cgrid, WALK_TPB, 0, stream>>>(a);
        else
            k_classify_reads<NARROW><<<cgrid, WALK_TPB, 0, stream>>>(a);
        SPX_HIP(hipGetLastError());
    }
    // reads in which a seam did not close: the plain walk (rare; results and class are overwritten)
    a.only_flagged = a.ch.read_fail;
    if ((rc = launch_lanes<MODE, DOC, true, NARROW, 0>(ix, a, stream)) != SPX_OK) return rc;
    return launch_len_expand(ix, a, stream);
}

}  // namespace

// Long-read batches (BASELINE config 5: 50 000 x 10 kbp, 6 250 per GPU): fewer reads than the chip has
// lanes.  Cut them into chunks and walk the chunks (spx_internal.h).  *done = false: not such a batch.
// Offsets are absolute (a piece of a larger host batch starts where the piece before it ended; a caller's
// offsets[0] need not be 0): the kernels index the per-character scratch relative to offs[0], which they read
// from device memory.
int launch_walk_chunked(spx_index* ix, int mode, const BatchArgs& args, uint64_t total_chars, hipStream_t stream,
                        bool* done, uint64_t geom_chars) {
    *done = false;
    const uint64_t gchars = geom_chars ? geom_chars : total_chars;  // shapes the chunks; total_chars bounds the scratch
    ix->last_chunk_len = ix->last_chunk_bound = 0;
    if (!ix->view.compact || ix->force_lanes_per_wave > 0 || args.nreads == 0) return SPX_OK;
    if (mode == SPX_MODE_PML && args.out_lengths == nullptr) return SPX_OK;  // classification only: the plain walk
    if (ix->num_cus == 0) {
        hipDeviceProp_t prop;
        SPX_HIP(hipGetDeviceProperties(&prop, ix->device));
        ix->num_cus = prop.multiProcessorCount;
    }
    // what the walk keeps resident: launch_lanes' occupancy target for pass 1 (k_walk_fast; SPX_CHUNK_GEOM_WAVES overrides:
    // the geometry below was laid out for 20 wavefronts per CU until round 4 lowered the plain walk's default to 16 / 12 --
    // 623 000 chunks on 262 144 lanes were 2.4 rounds)
    static const int geom_waves_env = getenv("SPX_CHUNK_GEOM_WAVES") ? atoi(getenv("SPX_CHUNK_GEOM_WAVES")) : 0;
    const bool side = mode == SPX_MODE_MS || args.out_docs != nullptr;
    const int geom_waves = geom_waves_env > 0 ? geom_waves_env : (ix->waves_per_cu > 0 ? ix->waves_per_cu : (side ? 12 : 16));
    const uint64_t lanes = (uint64_t)ix->num_cus * (uint64_t)geom_waves * 64;
    int mode_knob = ix->chunk_mode;                          // 0 automatic, 1 never, 2 always (tests)
    if (mode_knob == 1) return SPX_OK;
    // Chunk size (a multiple of the checkpoint spacing).  A seam closes within a few dozen characters
    // (more for small alphabets: two walks approach each other by a factor of the alphabet per
    // character), and every chunk costs that much again in pass 2, so chunks should not be short; but
    // each lane should get a whole number of them: with k chunks per lane the chunk is
    // total / (k * lanes) characters, k chosen for a chunk near the preferred size.
    const uint32_t CK = 1u << CKPT_SHIFT;
    // Every read end cuts one more chunk, so k chunks per lane means total / L + nreads <= k * lanes.
    const uint32_t pref = ix->view.nletters > 16 ? 176 : 640, lo = ix->view.nletters > 16 ? 128 : 512;
    uint64_t k = (gchars / lanes + pref / 2) / pref;
    if (k < 1) k = 1;
    while (k * lanes <= args.nreads + args.nreads / 8) k++;
    const uint64_t slots = k * lanes - args.nreads - args.nreads / 16;  // full chunks that fit
    uint64_t L64 = ((gchars + slots - 1) / slots + CK - 1) / CK * CK;
    if (L64 < lo) L64 = lo;
    if (L64 > 1024) L64 = 1024;
    if (ix->chunk_shift > 0) L64 = 1ull << (ix->chunk_shift > 20 ? 20 : ix->chunk_shift);
    if (ix->chunk_len > 0) L64 = ((uint64_t)ix->chunk_len + CK - 1) / CK * CK;
    if (L64 < 2 * CK) L64 = 2 * CK;
    const uint32_t L = (uint32_t)L64;
    if (mode_knob != 2) {
        // worth it when the reads alone leave most lanes idle and are long enough to cut
        if (args.nreads * 2 > lanes || gchars < args.nreads * (4ull * L)) return SPX_OK;
    }
    const uint64_t bound = total_chars / L + 2 * args.nreads + 1;
    // scratch (grow-only, owned by the index)
    const uint64_t nck = (total_chars >> CKPT_SHIFT) + 2;
    size_t cub_bytes = 0;
    SPX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr,
                                             (int)(args.nreads + 1), stream));
    void *p_desc, *p_ends, *p_seams, *p_ckpt, *p_flags, *p_fail, *p_cnt, *p_start, *p_cub;
    const size_t fail_words = args.nreads + 1 + 16;  // per read, then the per-round counters
    int rc;
    if ((rc = chunk_scratch(ix, 0, bound * sizeof(ChunkDesc), &p_desc)) != SPX_OK) return rc;
    if ((rc = chunk_scratch(ix, 1, 2 * (bound + 1) * sizeof(WalkState), &p_ends)) != SPX_OK) return rc;
    if ((rc = chunk_scratch(ix, 2, bound * sizeof(SeamRec), &p_seams)) != SPX_OK) return rc;
    if ((rc = chunk_scratch(ix, 3, nck * sizeof(WalkState), &p_ckpt)) != SPX_OK) return rc;
    if ((rc = chunk_scratch(ix, 4, total_chars + 32, &p_flags)) != SPX_OK) return rc;
    if ((rc = chunk_scratch(ix, 5, fail_words * 4, &p_fail)) != SPX_OK) return rc;
    if ((rc = chunk_scratch(ix, 6, (args.nreads + 2) * 8 * 3 + 16, &p_cnt)) != SPX_OK) return rc;
    if ((rc = chunk_scratch(ix, 7, cub_bytes + 256, &p_cub)) != SPX_OK) return rc;
    uint64_t* cnt = (uint64_t*)p_cnt;
    uint64_t* nchunks = cnt + (args.nreads + 1);      // one counter
    p_start = cnt + (args.nreads + 2);                // nreads + 1 entries... laid out after the counter
    // (p_cnt holds cnt[nreads + 1], the counter, and chunk_start[nreads + 1]: sized above as 2 (nreads + 2) words)
    uint64_t* chunk_start = (uint64_t*)p_start;
    const unsigned grid = (unsigned)((args.nreads + 1 + WALK_TPB - 1) / WALK_TPB);
    k_chunk_count<<<grid, WALK_TPB, 0, stream>>>(args.offs, args.nreads, L, (int)args.narrow, cnt, (uint32_t*)p_fail,
                                                  args.counters);
    SPX_HIP(hipGetLastError());
    SPX_HIP(hipcub::DeviceScan::ExclusiveSum(p_cub, cub_bytes, cnt, chunk_start, (int)(args.nreads + 1), stream));
    k_chunk_fill<<<grid, WALK_TPB, 0, stream>>>(args.offs, args.nreads, L, chunk_start, (ChunkDesc*)p_desc, nchunks);
    SPX_HIP(hipGetLastError());
    BatchArgs a = args;
    a.ch.desc = (const ChunkDesc*)p_desc;
    a.ch.nchunks = nchunks;
    a.ch.ends = (WalkState*)p_ends;
    a.ch.ckpt = (WalkState*)p_ckpt;  // indexed by (character index >> CKPT_SHIFT) - (offs[0] >> CKPT_SHIFT)
    a.ch.seams = (SeamRec*)p_seams;
    // indexed by character index; flags are read and written in aligned groups of 8, so the origin is moved by a
    // multiple of 8 and a group in front of the first character stays inside the buffer
    a.ch.flags = (uint8_t*)p_flags + 8;
    a.ch.read_fail = (uint32_t*)p_fail;
    a.ch.chunk_start = chunk_start;
    a.ch.reentry = (WalkState*)p_ends + (bound + 1);
    a.ch.vtop = chunk_start + (args.nreads + 2);
    a.ch.pending = (uint32_t*)p_fail + (args.nreads + 1);
    SPX_HIP(hipMemsetAsync(a.ch.pending, 0, 16 * 4, stream));  // [0, 8) rounds, [8, 16) the passes' chunk counters
    const bool doc = args.out_docs != nullptr;
    const int sel = (mode == SPX_MODE_MS ? 4 : 0) | (doc ? 2 : 0) | (args.narrow ? 1 : 0);
    switch (sel) {
#define SPX_CASE(n, M, D, N) \
    case n:                  \
        rc = run_chunked<M, D, N>(ix, a, bound, stream); \
        break;
        SPX_CASE(0, SPX_MODE_PML, false, false)
        SPX_CASE(1, SPX_MODE_PML, false, true)
        SPX_CASE(2, SPX_MODE_PML, true, false)
        SPX_CASE(3, SPX_MODE_PML, true, true)
        SPX_CASE(4, SPX_MODE_MS, false, false)
        SPX_CASE(5, SPX_MODE_MS, false, true)
        SPX_CASE(6, SPX_MODE_MS, true, false)
        SPX_CASE(7, SPX_MODE_MS, true, true)
#undef SPX_CASE
        default:
            rc = SPX_E_ARG;
    }
    if (rc == SPX_OK) {
        *done = true;
        ix->last_chunk_len = L;
        ix->last_chunk_bound = bound;
    }
    return rc;
}

// ---------------------------------------------------------------------------
// PML lengths from the walk's reset bits (BatchArgs::len_mask).  length[p] = q - p, q = the first character at
// or after p whose step reset the length (bit set), or the read's length when there is none: the reference's
// `length = 0` / `length++` (compute_ms_pml.cpp:249-250, 266-276) read from the other side.  The walk writes 16
// bytes per 128 characters instead of 2-4 bytes per character in scattered pieces (its stores cost a fifth of its
// time, profiles/r02_store_experiments.txt); this kernel writes the lengths as a stream: groups of 8 output
// elements aligned in memory, one 16-byte (16-bit outputs) or two (32-bit) stores each, `lpr` consecutive lanes
// per read.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t next_reset(const uint64_t* mw, uint32_t nw, uint64_t p, uint64_t m) {
    uint32_t w = (uint32_t)(p >> 6);
    if (w >= nw) return m;
    uint64_t v = mw[w] & (~0ull << (p & 63));
    while (v == 0) {
        if (++w >= nw) return m;
        v = mw[w];
    }
    return ((uint64_t)w << 6) + (uint64_t)__builtin_ctzll(v);
}

template <bool NARROW>
__global__ void __launch_bounds__(WALK_TPB) k_expand_lengths(const BatchArgs b, uint32_t lpr_shift) {
    if (b.only_flagged != nullptr && b.counters->pad_ == 0) return;  // no read fell back to the plain walk
    const uint64_t tid = blockIdx.x * (uint64_t)WALK_TPB + threadIdx.x;
    // lanes per read from the batch's real size: the caller's total_chars, which sized the grid, may be an upper
    // bound (reads digested on the device: DESIGN.md 4.4) -- the surplus threads leave at once
    {
        const uint64_t groups = (b.offs[b.nreads] - b.offs[0]) / b.nreads / 8 + 1;
        uint32_t sh = 0;
        while (sh < lpr_shift && (1ull << sh) < groups) ++sh;
        lpr_shift = sh;
    }
    const uint64_t rd = tid >> lpr_shift;
    if (rd >= b.nreads) return;
    if (b.only_flagged != nullptr && b.only_flagged[rd] == 0) return;
    const uint32_t lpr = 1u << lpr_shift, j = (uint32_t)tid & (lpr - 1);
    const uint64_t base = b.offs[rd], end = b.offs[rd + 1], m = end - base;
    if (m == 0) return;
    const uint64_t pair0 = ((base - b.offs[0]) >> 7) + rd;
    const uint32_t nw = (uint32_t)((m + 63) >> 6);  // (k_walk_fast writes the words that hold characters, no more)
    if (pair0 + (nw + 1) / 2 > b.len_mask_pairs) return;  // more characters than total_chars said: the walk reported it
    const uint64_t* const mw = b.len_mask + 2 * pair0;
    uint16_t* const out16 = reinterpret_cast<uint16_t*>(b.out_lengths);
    // the first reset at or after a position, remembered across the lane's groups: a long reset-free stretch is scanned
    // once, not once per group of 8 (a matching stretch of g characters cost g^2 / 512 word loads)
    uint64_t far_from = ~0ull, far_is = 0;
    for (uint64_t G = (base >> 3) + j; G <= ((end - 1) >> 3); G += lpr) {
        const uint64_t glo = G * 8 < base ? base : G * 8, ghi = G * 8 + 8 > end ? end : G * 8 + 8;
        const uint32_t cnt = (uint32_t)(ghi - glo);  // 8 but for the read's first and last group
        const uint64_t p = glo - base;
        const uint32_t w = (uint32_t)(p >> 6), s = (uint32_t)p & 63;
        uint64_t bits = mw[w] >> s;  // resets of characters p .. p + 63
        if (s != 0 && w + 1 < nw) bits |= mw[w + 1] << (64 - s);
        uint64_t far = 0;  // first reset at or after p + 64 (only when the window holds none for the last element)
        if ((bits >> (cnt - 1)) == 0) {
            // no reset in [far_from, far_is): the answer for any start in that range is far_is
            if (!(far_from != ~0ull && p + 64 >= far_from && p + 64 <= far_is)) {
                far_from = p + 64;
                far_is = next_reset(mw, nw, p + 64, m);
            }
            far = far_is;
        }
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint64_t t = bits >> i;
            v[i] = t ? (uint32_t)__builtin_ctzll(t) : (uint32_t)(far - (p + i));
        }
        if (cnt == 8) {
            if (NARROW) {
                // (plain stores: non-temporal and write-through ones were 4-12 % slower end to end)
                *reinterpret_cast<uint4*>(out16 + glo) =
                    make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
            } else {
                *reinterpret_cast<uint4*>(b.out_lengths + glo) = make_uint4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<uint4*>(b.out_lengths + glo + 4) = make_uint4(v[4], v[5], v[6], v[7]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if ((uint32_t)i < cnt) {
                    if (NARROW)
                        out16[glo + i] = (uint16_t)v[i];
                    else
                        b.out_lengths[glo + i] = v[i];
                }
        }
    }
}

int prepare_len_mask(spx_index* ix, int mode, BatchArgs& args) {
    args.len_mask = nullptr;
    if (mode != SPX_MODE_PML || args.out_lengths == nullptr || args.nreads == 0) return SPX_OK;
    void* p = nullptr;
    const uint64_t pairs = (args.total_chars >> 7) + args.nreads + 2;
    const int rc = chunk_scratch(ix, 8, pairs * 16, &p);
    if (rc != SPX_OK) return rc;
    args.len_mask = (uint64_t*)p;
    args.len_mask_pairs = pairs;
    return SPX_OK;
}

int launch_len_expand(spx_index* ix, const BatchArgs& args, hipStream_t stream) {
    (void)ix;
    if (args.len_mask == nullptr || args.nreads == 0) return SPX_OK;
    // lanes per read: the power of two that covers an average read's groups of 8 (longer reads loop)
    const uint64_t groups = args.total_chars / args.nreads / 8 + 1;
    uint32_t lpr_shift = 0;
    while (lpr_shift < 6 && (1ull << lpr_shift) < groups) ++lpr_shift;
    const uint64_t threads = args.nreads << lpr_shift;
    const unsigned grid = (unsigned)((threads + WALK_TPB - 1) / WALK_TPB);
    if (args.narrow)
        k_expand_lengths<true><<<grid, WALK_TPB, 0, stream>>>(args, lpr_shift);
    else
        k_expand_lengths<false><<<grid, WALK_TPB, 0, stream>>>(args, lpr_shift);
    SPX_HIP(hipGetLastError());
    return SPX_OK;
}

int launch_walk(spx_index* ix, int mode, const BatchArgs& args, uint64_t total_chars,
                hipStream_t stream, bool* wrote_lengths) {
    (void)total_chars;
    if (wrote_lengths) *wrote_lengths = false;
    const bool doc = args.out_docs != nullptr;
    // pick the instantiation: mode x doc x row encoding x output width
    const int sel = (mode == SPX_MODE_MS ? 8 : 0) | (doc ? 4 : 0) | (ix->view.compact ? 2 : 0) | (args.narrow ? 1 : 0);
    switch (sel) {
#define SPX_CASE(n, M, D, C, N) \
    case n:                     \
        return launch_lanes<M, D, C, N>(ix, args, stream, 0, wrote_lengths);
        SPX_CASE(0, SPX_MODE_PML, false, false, false)
        SPX_CASE(1, SPX_MODE_PML, false, false, true)
        SPX_CASE(2, SPX_MODE_PML, false, true, false)
        SPX_CASE(3, SPX_MODE_PML, false, true, true)
        SPX_CASE(4, SPX_MODE_PML, true, false, false)
        SPX_CASE(5, SPX_MODE_PML, true, false, true)
        SPX_CASE(6, SPX_MODE_PML, true, true, false)
        SPX_CASE(7, SPX_MODE_PML, true, true, true)
        SPX_CASE(8, SPX_MODE_MS, false, false, false)
        SPX_CASE(9, SPX_MODE_MS, false, false, true)
        SPX_CASE(10, SPX_MODE_MS, false, true, false)
        SPX_CASE(11, SPX_MODE_MS, false, true, true)
        SPX_CASE(12, SPX_MODE_MS, true, false, false)
        SPX_CASE(13, SPX_MODE_MS, true, false, true)
        SPX_CASE(14, SPX_MODE_MS, true, true, false)
        SPX_CASE(15, SPX_MODE_MS, true, true, true)
#undef SPX_CASE
    }
    return SPX_E_ARG;
}

int launch_text_check(spx_index* ix, unsigned long long* d_bad, hipStream_t stream) {
    const unsigned grid = (unsigned)((ix->view.r + WALK_TPB - 1) / WALK_TPB);
    k_text_check<<<grid ? grid : 1, WALK_TPB, 0, stream>>>(ix->view, d_bad);
    SPX_HIP(hipGetLastError());
    return SPX_OK;
}

int launch_text_from_index(spx_index* ix, uint8_t* d_text, uint64_t n_text, unsigned long long* d_stuck, hipStream_t stream) {
    const unsigned grid = (unsigned)((ix->view.r + WALK_TPB - 1) / WALK_TPB);
    k_text_from_index<<<grid ? grid : 1, WALK_TPB, 0, stream>>>(ix->view, d_text, n_text, d_stuck);
    SPX_HIP(hipGetLastError());
    return SPX_OK;
}

int launch_ms_extend(spx_index* ix, const BatchArgs& args, hipStream_t stream) {
    const unsigned grid = (unsigned)((args.nreads + EXT_TPB - 1) / EXT_TPB);
    k_ms_extend<<<grid ? grid : 1, EXT_TPB, 0, stream>>>(ix->view, args);
    SPX_HIP(hipGetLastError());
    return SPX_OK;
}

}  // namespace spx
