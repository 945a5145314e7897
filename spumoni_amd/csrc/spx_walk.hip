// spx_walk.hip -- the backward-search kernels (hand-written HIP for gfx950).
//
// What is computed is exactly pml_pointers::_query / ms_pointers::_query of the
// reference (/root/reference/src/compute_ms_pml.cpp:238-340, 571-682) and the
// length extension of ms_t::matching_statistics (:800-810); how it is computed
// is ours (spx_layout.h).
//
// k_walk_lanes -- "lane-per-read state machine".  The walk of one read is a
// chain of dependent 32-byte gathers (1 per matching character, ~4 per
// threshold jump), so throughput is purely a question of how many chains are in
// flight.  Every lane owns one read and runs a small state machine whose every
// iteration issues exactly ONE round of gathers (whatever its read needs next:
// a landing row, a directory count, a directory window, the successor /
// predecessor rows) and then consumes it.  All 64 lanes of a wave therefore
// always have a gather in flight, no lane ever waits for another lane's longer
// step, and lanes pull new reads from a device-wide queue as they finish.
// 2048 lanes per CU x 256 CUs = 524k independent chains keep HBM's random-access
// path saturated (Little: ~8 MB must be in flight).
//
// k_walk_wave -- "wavefront-per-read"  for batches with too few
// reads to fill the lanes (long-read workloads): see below.
#include "spx_internal.h"

namespace spx {

namespace {

enum : uint32_t { PH_NEW = 0, PH_LAND = 1, PH_CNT = 2, PH_QS = 3, PH_ROWS = 4, PH_SAMP = 5 };

constexpr int WALK_TPB = 256;

struct __attribute__((packed, aligned(4))) QWin {
    uint32_t e[8];
};

__device__ __forceinline__ Row load_row(const Row* p) {
    const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
    ulonglong2 a = q[0], b = q[1];
    Row r;
    r.q0 = a.x;
    r.q1 = a.y;
    r.q2 = b.x;
    r.q3 = b.y;
    return r;
}

// ---------------------------------------------------------------------------
// lane-per-read state machine
// ---------------------------------------------------------------------------
template <int MODE, bool DOC>
__global__ void __launch_bounds__(WALK_TPB) k_walk_lanes(const DevIndex ix, const BatchArgs b) {
    __shared__ LetterInfo s_let[256];
    for (int t = threadIdx.x; t < 256; t += WALK_TPB) s_let[t] = ix.letters[t];
    __syncthreads();

    const Row* __restrict__ rows = ix.rows;
    const uint32_t* __restrict__ Q = ix.Q;
    const uint64_t* __restrict__ seqs64 = reinterpret_cast<const uint64_t*>(b.seqs);
    const uint32_t R = ix.r;
    const bool want_class = (MODE == SPX_MODE_PML) && b.out_class != nullptr;

    uint32_t ph = PH_NEW;
    uint64_t rd = 0, base = 0;
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
    uint32_t c = 0, lo = 0, hi = 0, qs = 0, qp = 0, qbeg = 0, qend = 0;
    bool has_succ = false, has_pred = false, quirk = false;
    // character window
    uint64_t cw = 0, cwn = 0, cw_idx = 0;
    // classifier
    uint32_t bin_lo = 0, bin_max = 0, above = 0, below = 0;
    uint64_t sum_max = 0;
    // statistics
    uint32_t n_steps = 0, n_jumps = 0, n_pred = 0, n_rows = 0, n_dir = 0, n_err = 0;

    for (;;) {
        if (ph == PH_NEW) {
            rd = atomicAdd(&b.counters->next_read, 1ull);
            if (rd >= b.nreads) break;
            base = b.offs[rd];
            m = (uint32_t)(b.offs[rd + 1] - base);
            if (m == 0) {
                if (want_class) b.out_class[rd] = spx_class{0, 0, 0};
                continue;
            }
            x = m;
            length = 0;
            sample = ix.init_sample;  // compute_ms_pml.cpp:575
            doc = ix.init_doc;        // :298 / :634
            k0 = ix.init_k;           // pos = bwt_size() - 1   (:243 / :574)
            offp = ix.init_off;
            const uint64_t g = base + m - 1;
            cw_idx = g >> 3;
            cw = seqs64[cw_idx];
            cwn = seqs64[cw_idx ? cw_idx - 1 : 0];
            if (want_class) {
                const uint32_t w = (uint32_t)b.bin_width;
                const uint32_t nb = m / w > 0 ? m / w : 1;
                bin_lo = (nb - 1) * w;
                bin_max = above = below = 0;
                sum_max = 0;
            }
            ph = PH_LAND;
        }

        // ---- one round of gathers: whatever this lane's read needs next ----
        Row ra, rb;
        uint32_t e0 = 0, e1 = 0;
        QWin win;
        SamplePair sa{0, 0}, sb{0, 0};
        if (ph == PH_LAND) {
            ra = load_row(rows + k0);
        } else if (ph == PH_CNT) {
            const uint32_t* p = ix.cnt + (uint64_t)s_let[c].lid * ix.nblk + (k >> ix.bshift);
            e0 = p[0];
            e1 = p[1];
        } else if (ph == PH_QS) {
            const int64_t at = (int64_t)(hi - lo <= 6 ? lo : lo + ((hi - lo) >> 1)) - 1;
            win = *reinterpret_cast<const QWin*>(Q + at);
        } else if (ph == PH_ROWS) {
            const uint32_t ia = has_succ ? qs : qp;
            const uint32_t ib = has_pred ? qp : ia;
            ra = load_row(rows + ia);
            rb = load_row(rows + ib);
            if (MODE == SPX_MODE_MS) {
                sa = ix.samples[ia];
                sb = ix.samples[ib];
            }
        } else {  // PH_SAMP
            sa = ix.samples[k];
        }

        // ---- consume ----
        bool do_emit = false;  // a character's result is final -> write it and advance
        bool do_step = false;  // landed on a run -> look at the next character
        if (ph == PH_LAND) {
            n_rows++;
            const uint64_t len = row_len(ra);
            if (offp >= len) {  // LF image lies in a later run: skip this row
                offp -= len;
                k0++;
                continue;
            }
            k = k0;
            off = offp;
            S_k = row_S(ra);
            H_k = row_H(ra);
            LFrun_k = row_LFrun(ra);
            LFoff_k = row_LFoff(ra);
            THR_k = row_THR(ra);
            docs_k = row_docS(ra) | (row_docE(ra) << 16);
            do_step = true;
        } else if (ph == PH_CNT) {
            n_dir++;
            lo = e0;
            hi = e1;
            ph = PH_QS;
            continue;
        } else if (ph == PH_QS) {
            n_dir++;
            if (hi - lo <= 6) {
                // window holds Q[lo-1 .. lo+6]; j = number of c-runs with index < k
                uint32_t cntlt = 0;
#pragma unroll
                for (int t = 1; t <= 6; ++t) cntlt += ((uint32_t)t <= hi - lo && win.e[t] < k) ? 1u : 0u;
                const uint32_t j = lo + cntlt;
                qp = win.e[0];
                qs = win.e[1];
#pragma unroll
                for (int t = 1; t <= 6; ++t) {
                    if (cntlt == (uint32_t)t) {
                        qp = win.e[t];
                        qs = win.e[t + 1];
                    }
                }
                has_succ = j < qend;
                has_pred = j > qbeg;
                ph = PH_ROWS;
            } else {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (win.e[1] < k)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            continue;
        } else if (ph == PH_ROWS) {
            // compute_ms_pml.cpp:253-278 (PML) / :585-615 (MS) on the flat layout
            const uint64_t pos = S_k + off;  // sentinel row r has S = n, off = 0
            uint64_t thr = ix.n + 1;         // :254
            uint32_t nk = k;
            uint64_t noff = off;
            const Row* land = &ra;
            if (has_succ) {  // rnk < number_of_letter(c)  (:259)
                thr = row_THR(ra);
                length = 0;
                sample = sa.ss;           // samples_start[run_of_j]  (:601)
                doc = row_docS(ra);       // start_runs_doc[run_of_j] (:317)
                if (!quirk) {
                    nk = qs;              // next_pos = j = start of the next c-run
                    noff = 0;
                }
            }
            if (pos < thr) {  // :270
                n_pred++;
                length = 0;
                if (quirk && off > 0) {  // select(rnk-1, c) is the previous position of run k
                    noff = off - 1;
                    sample = sa.se;
                    doc = row_docE(ra);
                } else {
                    if (!has_pred) n_err++;  // rnk-- below zero: undefined upstream
                    nk = qp;
                    noff = row_len(rb) - 1;
                    sample = sb.se;        // samples_last[run_of_j]  (:611)
                    doc = row_docE(rb);    // end_runs_doc[run_of_j]  (:327)
                    land = &rb;
                }
            }
            // pos = next_pos; pos = LF(pos, c)   (:278, :284)
            k0 = row_LFrun(*land);
            offp = row_LFoff(*land) + noff;
            (void)nk;
            do_emit = true;
        } else {  // PH_SAMP: byte >= 128 sitting on its own run (Appendix C1), MS mode
            sample = sa.ss;
            k0 = LFrun_k;
            offp = LFoff_k + off;
            do_emit = true;
        }

        if (do_step) {
            // next character: auto c = pattern[m - i - 1]   (:247)
            const uint64_t g = base + x - 1;
            if ((g >> 3) != cw_idx) {
                cw = cwn;
                cw_idx = g >> 3;
                cwn = seqs64[cw_idx ? cw_idx - 1 : 0];
            }
            c = (uint32_t)(cw >> ((g & 7) * 8)) & 0xffu;
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
                // branch runs but select(rank(pos,c),c) == pos, and pos >= thr keeps it there
                n_jumps++;
                length = 0;
                doc = docs_k & 0xffff;  // start_runs_doc[run of pos]
                if (MODE == SPX_MODE_MS) {
                    ph = PH_SAMP;  // sample = samples_start[run of pos]
                    continue;
                }
                k0 = LFrun_k;
                offp = LFoff_k + off;
                do_emit = true;
            } else {
                n_jumps++;
                quirk = (k < R && H_k == c);
                qbeg = li.qbeg;
                qend = li.qend;
                ph = PH_CNT;
                continue;
            }
        }

        if (do_emit) {
            const uint32_t xi = x - 1;
            if (MODE == SPX_MODE_PML) {
                b.out_lengths[base + xi] = length;  // lengths[m-i-1] = length   (:281)
            } else {
                b.out_pointers[base + xi] = sample;  // :618
            }
            if (DOC) b.out_docs[base + xi] = doc;  // :336 / :677
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
                ph = PH_NEW;
            } else {
                ph = PH_LAND;
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
