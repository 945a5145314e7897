// spx_walk.hip -- the backward-search kernels (hand-written HIP for gfx950).
//
// What is computed is exactly pml_pointers::_query / ms_pointers::_query of the
// reference (/root/reference/src/compute_ms_pml.cpp:238-340, 571-682) and the
// length extension of ms_t::matching_statistics (:800-810); how it is computed
// is ours (spx_layout.h).
//
// k_walk_lanes -- "lane-per-read state machine".
// The walk of one read is a chain of dependent 32-byte gathers (1 per matching
// character, 3-4 per threshold jump), and MI355X's random-gather ceiling is
// ~50 G L2-line fills/s (tools/gather_bench.hip) reached with >= 64k chains in
// flight -- far more chains than the 8k wavefronts the chip holds.  So every
// LANE owns one read and runs a small state machine.  Each loop iteration
// issues exactly ONE 32-byte gather per lane, from a single load site, at the
// address the lane's current phase asks for:
//
//   P_READ   offsets[rd], offsets[rd+1]         (next read of this lane)
//   P_CHARS  32 characters of the read          (refill, every <= 32 steps)
//   P_LAND   rows[k0]                           (run the LF step lands in)
//   P_CNT    cnt[letter][k >> bshift]           (directory block bounds)
//   P_QS     Q[lo-1 .. lo+6]                    (only if the block holds c-runs)
//   P_DIR    dirrows[j]                         (threshold + both jump landings)
//
// then consumes it and moves to the next phase.  Because the load site is
// unique, all 64 lanes of a wave always have their gather in flight together
// no matter which phase each is in: no lane waits for another lane's longer
// step, and no phase serialises behind another phase's s_waitcnt.
#include "spx_internal.h"

namespace spx {

namespace {

enum : uint32_t {
    P_READ = 0,
    P_CHARS = 1,
    P_LAND = 2,
    P_CNT = 3,
    P_QS = 4,
    P_DIR = 5,
    P_SAMP = 6,
    P_DONE = 7
};

constexpr int WALK_TPB = 256;

struct __attribute__((packed, aligned(4))) V16 {  // 16 bytes at 4-byte alignment
    uint32_t x, y, z, w;
};

__device__ __forceinline__ uint64_t u64of(uint32_t lo, uint32_t hi) {
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// ---------------------------------------------------------------------------
// lane-per-read state machine
// ---------------------------------------------------------------------------
template <int MODE, bool DOC>
__global__ void __launch_bounds__(WALK_TPB) k_walk_lanes(const DevIndex ix, const BatchArgs b) {
    __shared__ LetterInfo s_let[256];
    for (int t = threadIdx.x; t < 256; t += WALK_TPB) s_let[t] = ix.letters[t];
    __syncthreads();

    const char* const rows_b = reinterpret_cast<const char*>(ix.rows);
    const char* const dir_b = reinterpret_cast<const char*>(ix.dirrows);
    const char* const cnt_b = reinterpret_cast<const char*>(ix.cnt);
    const char* const q_b = reinterpret_cast<const char*>(ix.Q);
    const char* const seq_b = reinterpret_cast<const char*>(b.seqs);
    const char* const off_b = reinterpret_cast<const char*>(b.offs);
    const uint32_t R = ix.r;
    const bool want_class = (MODE == SPX_MODE_PML) && b.out_class != nullptr;
    const uint64_t nlanes = (uint64_t)gridDim.x * WALK_TPB;

    uint32_t ph = P_READ;
    uint64_t rd = (uint64_t)blockIdx.x * WALK_TPB + threadIdx.x;
    uint64_t base = 0;
    uint32_t m = 0, x = 0;  // x = characters still to search; next one is index x-1
    // landed position: run k, offset off; fields of row k
    uint32_t k = 0, H_k = 0, LFrun_k = 0, docs_k = 0;
    uint64_t off = 0, S_k = 0, LFoff_k = 0, THR_k = 0;
    // landing target
    uint32_t k0 = 0;
    uint64_t offp = 0;
    // per-step results
    uint32_t length = 0, doc = 0;
    uint64_t sample = 0;
    // jump bookkeeping
    uint32_t c = 0, lo = 0, hi = 0, jdir = 0, qbeg = 0, qend = 0;
    bool quirk = false;
    // character window: 32 bytes starting at byte offset wbase of seqs
    uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, wbase = 0;
    // output staging (PML): 8 u16 values of the aligned group [obase, obase+8)
    uint64_t ob_lo = 0, ob_hi = 0;
    // classifier
    uint32_t bin_lo = 0, bin_max = 0, above = 0, below = 0;
    uint64_t sum_max = 0;
    // statistics
    uint32_t n_steps = 0, n_jumps = 0, n_pred = 0, n_rows = 0, n_dir = 0, n_err = 0;

    if (rd >= b.nreads) ph = P_DONE;

    while (ph != P_DONE) {
        // ---- the one gather of this iteration -------------------------------
        const char* p0;
        const char* p1;
        if (ph == P_LAND) {
            p0 = rows_b + (uint64_t)k0 * sizeof(Row);
            p1 = p0 + 16;
        } else if (ph == P_DIR) {
            p0 = dir_b + (uint64_t)jdir * sizeof(DirRow);
            p1 = p0 + 16;
        } else if (ph == P_CNT) {
            p0 = cnt_b + ((uint64_t)s_let[c].lid * ix.nblk + (k >> ix.bshift)) * 4;
            p1 = p0;
        } else if (ph == P_QS) {
            const int64_t at = (int64_t)(hi - lo <= 6 ? lo : lo + ((hi - lo) >> 1)) - 1;
            p0 = q_b + at * 4;
            p1 = p0 + 16;
        } else if (ph == P_CHARS) {
            p0 = seq_b + wbase;
            p1 = p0 + 16;
        } else if (ph == P_READ) {
            p0 = off_b + rd * 8;
            p1 = p0;
        } else {  // P_SAMP
            p0 = reinterpret_cast<const char*>(ix.ss_by_run + k);
            p1 = p0;
        }
        const V16 ga = *reinterpret_cast<const V16*>(p0);
        const V16 gb = *reinterpret_cast<const V16*>(p1);
        SamplePair sp{0, 0};
        if (MODE == SPX_MODE_MS) {
            // MS: the samples of the directory position travel with the directory row
            const SamplePair* ps = ix.samples + (ph == P_DIR ? jdir : 0);
            sp = *ps;
        }
        const uint64_t g0 = u64of(ga.x, ga.y), g1 = u64of(ga.z, ga.w);
        const uint64_t g2 = u64of(gb.x, gb.y), g3 = u64of(gb.z, gb.w);

        // ---- consume ---------------------------------------------------------
        bool do_emit = false;  // a character's result is final -> write it and advance
        bool do_step = false;  // landed on a run -> look at the next character
        if (ph == P_LAND) {
            n_rows++;
            Row ra;
            ra.q0 = g0;
            ra.q1 = g1;
            ra.q2 = g2;
            ra.q3 = g3;
            const uint64_t len = row_len(ra);
            if (offp >= len) {  // LF image lies in a later run: skip this row
                offp -= len;
                k0++;
            } else {
                k = k0;
                off = offp;
                S_k = row_S(ra);
                H_k = row_H(ra);
                LFrun_k = row_LFrun(ra);
                LFoff_k = row_LFoff(ra);
                THR_k = row_THR(ra);
                docs_k = row_docS(ra) | (row_docE(ra) << 16);
                do_step = true;
            }
        } else if (ph == P_DIR) {
            // compute_ms_pml.cpp:253-278 (PML) / :585-615 (MS) on the flat layout
            n_dir++;
            DirRow dr;
            dr.d0 = g0;
            dr.d1 = g1;
            dr.d2 = g2;
            dr.d3 = g3;
            const uint64_t pos = S_k + off;  // sentinel row r has S = n, off = 0
            const bool has_succ = jdir < qend;  // rnk < number_of_letter(c)   (:259)
            uint64_t thr = ix.n + 1;            // :254
            if (!quirk) {
                if (has_succ) {
                    thr = dir_THR(dr);
                    length = 0;
                    sample = sp.ss;        // samples_start[run_of_j]  (:601)
                    doc = dir_docS(dr);    // start_runs_doc[run_of_j] (:317)
                    k0 = dir_sLFrun(dr);   // next_pos = j; LF(j, c)
                    offp = dir_sLFoff(dr);
                }
                if (pos < thr) {  // :270  -> select(rnk-1, c): last character of the previous c-run
                    n_pred++;
                    if (jdir <= qbeg) n_err++;  // rnk-- below zero: undefined upstream
                    length = 0;
                    sample = sp.se;        // samples_last[run_of_j]  (:611)
                    doc = dir_docEp(dr);   // end_runs_doc[run_of_j]  (:327)
                    k0 = dir_pLFrun(dr);
                    offp = dir_pLFoff(dr);
                }
            } else {
                // byte >= 128 equal to the head of its run with pos < thresholds[run]
                // (Appendix C1; only reachable with inconsistent thresholds): jdir is the
                // directory position of run k when off == 0, of the next c-run when off > 0
                n_pred++;
                length = 0;
                sample = sp.se;
                if (off > 0) {  // select(rnk-1, c) = pos - 1, still inside run k
                    doc = docs_k >> 16;
                    k0 = LFrun_k;
                    offp = LFoff_k + off - 1;
                } else {
                    if (jdir <= qbeg) n_err++;
                    doc = dir_docEp(dr);
                    k0 = dir_pLFrun(dr);
                    offp = dir_pLFoff(dr);
                }
            }
            do_emit = true;
        } else if (ph == P_CNT) {
            n_dir++;
            lo = (uint32_t)g0;
            hi = (uint32_t)(g0 >> 32);
            if (hi == lo) {  // no c-run inside the block: the successor is directory entry lo
                jdir = lo;
                ph = P_DIR;
            } else {
                ph = P_QS;
            }
        } else if (ph == P_QS) {
            n_dir++;
            const uint32_t e[8] = {(uint32_t)g0, (uint32_t)(g0 >> 32), (uint32_t)g1, (uint32_t)(g1 >> 32),
                                   (uint32_t)g2, (uint32_t)(g2 >> 32), (uint32_t)g3, (uint32_t)(g3 >> 32)};
            if (hi - lo <= 6) {
                // window holds Q[lo-1 .. lo+6]; j = lo + number of c-runs of the block with index < k
                uint32_t cntlt = 0;
#pragma unroll
                for (int t = 1; t <= 6; ++t) cntlt += ((uint32_t)t <= hi - lo && e[t] < k) ? 1u : 0u;
                jdir = lo + cntlt;
                if (quirk && off > 0) jdir++;  // see P_DIR: entry after run k carries samples_last[k]
                ph = P_DIR;
            } else {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (e[1] < k)
                    lo = mid + 1;
                else
                    hi = mid;
            }
        } else if (ph == P_CHARS) {
            w0 = g0;
            w1 = g1;
            w2 = g2;
            w3 = g3;
            do_step = true;  // only entered from a landed state
        } else if (ph == P_READ) {
            base = g0;
            m = (uint32_t)(g1 - g0);
            if (m == 0) {
                if (want_class) b.out_class[rd] = spx_class{0, 0, 0};
                rd += nlanes;
                if (rd >= b.nreads) ph = P_DONE;
            } else {
                x = m;
                length = 0;
                sample = ix.init_sample;  // compute_ms_pml.cpp:575
                doc = ix.init_doc;        // :298 / :634
                k0 = ix.init_k;           // pos = bwt_size() - 1   (:243 / :574)
                offp = ix.init_off;
                wbase = ~0ull;            // no characters loaded yet
                ob_lo = ob_hi = 0;
                if (want_class) {
                    const uint32_t w = (uint32_t)b.bin_width;
                    const uint32_t nb = m / w > 0 ? m / w : 1;
                    bin_lo = (nb - 1) * w;
                    bin_max = above = below = 0;
                    sum_max = 0;
                }
                ph = P_LAND;
            }
        } else {  // P_SAMP: byte >= 128 sitting on its own run (Appendix C1), MS mode
            sample = g0;  // samples_start[run of pos]
            k0 = LFrun_k;
            offp = LFoff_k + off;
            do_emit = true;
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
                const uint64_t wsel = (wi & 16) ? ((wi & 8) ? w3 : w2) : ((wi & 8) ? w1 : w0);
                c = (uint32_t)(wsel >> ((wi & 7) * 8)) & 0xffu;
                const LetterInfo li = s_let[c];
                if (li.lid == NO_LETTER) {  // number_of_letter(c) == 0   (:249)
                    length = 0;
                    if (MODE == SPX_MODE_MS) {
                        sample = 0;                 // :581
                        if (DOC) doc = ix.doc_at0;  // :641-642
                    }
                    k0 = li.frun;  // LF(pos, c) = F[c] + 0
                    offp = li.foff;
                    do_emit = true;
                } else if (k < R && H_k == c && c < 128) {  // pos < n && bwt[pos] == c   (:250)
                    length++;
                    sample--;  // :582 (wraps, Appendix C3)
                    k0 = LFrun_k;
                    offp = LFoff_k + off;
                    do_emit = true;
                } else if (k < R && H_k == c && S_k + off >= THR_k) {
                    // byte >= 128 equal to the head (signed-char quirk, Appendix C1): the jump
                    // branch runs, select(rank(pos,c),c) == pos, and pos >= thr keeps it there
                    n_jumps++;
                    length = 0;
                    doc = docs_k & 0xffff;  // start_runs_doc[run of pos]
                    if (MODE == SPX_MODE_MS) {
                        ph = P_SAMP;  // sample = samples_start[run of pos]
                    } else {
                        k0 = LFrun_k;
                        offp = LFoff_k + off;
                        do_emit = true;
                    }
                } else {
                    n_jumps++;
                    quirk = (k < R && H_k == c);
                    qbeg = li.qbeg;
                    qend = li.qend;
                    ph = P_CNT;
                }
            }
        }

        if (do_emit) {
            const uint32_t xi = x - 1;
            const uint64_t gi = base + xi;
            if (MODE == SPX_MODE_PML) {
                // lengths[m-i-1] = length (:281), staged 8 at a time (u16) when the read is short
                if (m < 65536) {
                    const uint32_t slot = (uint32_t)gi & 7;
                    const uint64_t v = (uint64_t)length << ((slot & 3) * 16);
                    if (slot & 4)
                        ob_hi |= v;
                    else
                        ob_lo |= v;
                    if (slot == 0 || xi == 0) {
                        // flush the group [g8, g8+8) restricted to this read's range
                        const uint64_t g8 = gi & ~7ull;
                        uint32_t* o = b.out_lengths + g8;
                        const bool full_lo = (g8 >= base) && (g8 + 3 < base + m);
                        const bool full_hi = (g8 + 4 >= base) && (g8 + 7 < base + m);
                        if (full_lo) {
                            uint4 v4 = make_uint4((uint32_t)ob_lo & 0xffff, (uint32_t)(ob_lo >> 16) & 0xffff,
                                                  (uint32_t)(ob_lo >> 32) & 0xffff, (uint32_t)(ob_lo >> 48));
                            *reinterpret_cast<uint4*>(o) = v4;
                        }
                        if (full_hi) {
                            uint4 v4 = make_uint4((uint32_t)ob_hi & 0xffff, (uint32_t)(ob_hi >> 16) & 0xffff,
                                                  (uint32_t)(ob_hi >> 32) & 0xffff, (uint32_t)(ob_hi >> 48));
                            *reinterpret_cast<uint4*>(o + 4) = v4;
                        }
                        if (!full_lo || !full_hi) {
#pragma unroll
                            for (int t = 0; t < 8; ++t) {
                                const bool covered = t < 4 ? full_lo : full_hi;
                                const uint64_t gt = g8 + t;
                                if (!covered && gt >= gi && gt < base + m) {
                                    const uint64_t wv = t < 4 ? ob_lo : ob_hi;
                                    o[t] = (uint32_t)(wv >> ((t & 3) * 16)) & 0xffff;
                                }
                            }
                        }
                        ob_lo = ob_hi = 0;
                    }
                } else {
                    b.out_lengths[gi] = length;
                }
            } else {
                b.out_pointers[gi] = sample;  // :618
            }
            if (DOC) b.out_docs[gi] = doc;  // :336 / :677
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
                rd += nlanes;
                ph = rd < b.nreads ? P_READ : P_DONE;
            } else {
                ph = P_LAND;
            }
        }
    }

    atomicAdd(&b.counters->steps, (unsigned long long)n_steps);
    atomicAdd(&b.counters->jumps, (unsigned long long)n_jumps);
    atomicAdd(&b.counters->pred_jumps, (unsigned long long)n_pred);
    atomicAdd(&b.counters->row_loads, (unsigned long long)n_rows);
    atomicAdd(&b.counters->dir_loads, (unsigned long long)n_dir);
    if (n_err) atomicAdd(&b.counters->error, (unsigned long long)n_err);
}

// ---------------------------------------------------------------------------
// MS length extension: ms_t::matching_statistics second loop
// (compute_ms_pml.cpp:800-810) with plain text in HBM instead of the SLP.
// One lane per read; `l` is carried exactly like the reference.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(WALK_TPB) k_ms_extend(const DevIndex ix, const BatchArgs b) {
    const uint64_t rd = blockIdx.x * (uint64_t)WALK_TPB + threadIdx.x;
    if (rd >= b.nreads) return;
    const uint64_t base = b.offs[rd];
    const uint64_t m = b.offs[rd + 1] - base;
    const uint8_t* read = b.seqs + base;
    const uint64_t* ptrs = b.out_pointers + base;
    uint32_t* out = b.out_lengths + base;
    const uint8_t* text = ix.text;
    const uint64_t n = ix.n_text;
    const bool want_class = b.out_class != nullptr;
    uint64_t l = 0, prev = 0;
    // classifier over ascending indices
    const uint64_t w = b.bin_width ? b.bin_width : 1;
    const uint64_t nb = m / w > 0 ? m / w : 1;
    uint64_t bin_hi = nb > 1 ? w : m;  // exclusive end of the current bin
    uint64_t bin_idx = 0;
    uint32_t bin_max = 0, above = 0, below = 0;
    uint64_t sum_max = 0;
    for (uint64_t i = 0; i < m; ++i) {
        const uint64_t pos = ptrs[i];
        const bool cont = (i >= 1) && (pos == prev + 1);
        if (!cont) {
            // unsigned arithmetic as upstream: pos + l may wrap for wrapped pointers (C3)
            while ((i + l) < m && (pos + l) < n && read[i + l] == text[pos + l]) ++l;
        }
        out[i] = (uint32_t)l;
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
    if (want_class) {
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

template <int MODE, bool DOC>
int launch_lanes(spx_index* ix, const BatchArgs& args, hipStream_t stream) {
    int occ = 0;
    SPX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_walk_lanes<MODE, DOC>, WALK_TPB, 0));
    if (occ < 1) occ = 1;
    if (ix->waves_per_cu > 0) {
        int want = ix->waves_per_cu / (WALK_TPB / 64);
        if (want >= 1 && want < occ) occ = want;
    }
    hipDeviceProp_t prop;
    SPX_HIP(hipGetDeviceProperties(&prop, ix->device));
    uint64_t grid = (uint64_t)occ * prop.multiProcessorCount;
    uint64_t need = (args.nreads + WALK_TPB - 1) / WALK_TPB;
    if (need < grid) grid = need;
    if (grid == 0) grid = 1;
    k_walk_lanes<MODE, DOC><<<(unsigned)grid, WALK_TPB, 0, stream>>>(ix->view, args);
    SPX_HIP(hipGetLastError());
    return SPX_OK;
}

}  // namespace

int launch_walk(spx_index* ix, int mode, const BatchArgs& args, uint64_t total_chars,
                hipStream_t stream) {
    (void)total_chars;
    const bool doc = args.out_docs != nullptr;
    if (mode == SPX_MODE_PML) return doc ? launch_lanes<SPX_MODE_PML, true>(ix, args, stream)
                                          : launch_lanes<SPX_MODE_PML, false>(ix, args, stream);
    return doc ? launch_lanes<SPX_MODE_MS, true>(ix, args, stream)
               : launch_lanes<SPX_MODE_MS, false>(ix, args, stream);
}

int launch_ms_extend(spx_index* ix, const BatchArgs& args, hipStream_t stream) {
    const unsigned grid = (unsigned)((args.nreads + WALK_TPB - 1) / WALK_TPB);
    k_ms_extend<<<grid ? grid : 1, WALK_TPB, 0, stream>>>(ix->view, args);
    SPX_HIP(hipGetLastError());
    return SPX_OK;
}

}  // namespace spx
