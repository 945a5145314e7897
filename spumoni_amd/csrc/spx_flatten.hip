// spx_flatten.hip -- builds the flat HBM layout (spx_layout.h) ON THE DEVICE from
// the raw per-run arrays of the index (heads / lengths / thresholds [/ samples /
// docs]).  Replaces, for the GPU, what ms_rle_string's RLE constructor
// (include/ms_rle_string.hpp:217-288), build_F_ (src/compute_ms_pml.cpp:119-147)
// and thr_bv's constructor (include/thresholds_ds.hpp:384-440) build on the CPU.
//
// Everything is a scan, a stable 8-bit radix sort or a per-run gather, so even a
// 10^9-run index is laid out in seconds without touching the host.
#include <hipcub/hipcub.hpp>

#include <functional>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "spx_internal.h"

namespace spx {

namespace {

constexpr int TPB = 256;
// index arrays start out zeroed: their padding is part of the .spx cache, which has to be a function
// of the input alone
#define SPX_ALLOC0(ptr, bytes)                                   \
    do {                                                         \
        const size_t nb_ = (bytes);                              \
        SPX_HIP(hipMalloc((void**)&(ptr), nb_));                 \
        SPX_HIP(hipMemsetAsync((ptr), 0, nb_, nullptr));         \
    } while (0)
inline unsigned nblocks(uint64_t n) { return (unsigned)((n + TPB - 1) / TPB); }

struct DevBuf {  // RAII scratch buffer
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T>
    T* as() {
        return (T*)p;
    }
};

// heads: 0 -> TERMINATOR(1) (ms_rle_string.hpp:249-253); also validates lens > 0
__global__ void k_norm_heads(const uint8_t* heads, const uint64_t* lens, uint64_t r, uint8_t* H,
                             uint32_t* iota, unsigned long long* err) {
    uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (i >= r) return;
    uint8_t c = heads[i];
    H[i] = c <= 1 ? 1 : c;
    iota[i] = (uint32_t)i;
    if (lens[i] == 0 || lens[i] > MASK40) atomicAdd(err, 1ull);
}

__global__ void k_gather_u64(const uint64_t* src, const uint32_t* idx, uint64_t r, uint64_t* dst) {
    uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (i < r) dst[i] = src[idx[i]];
}

__global__ void k_nonzero_flag(const uint64_t* thr, const uint32_t* idx, uint64_t r, uint32_t* flag) {
    uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (i < r) flag[i] = thr[idx[i]] != 0 ? 1u : 0u;
}

// compact the non-zero thresholds in (letter, run) order: what thr_bv stores per letter
__global__ void k_compact_thr(const uint64_t* thr, const uint32_t* idx, const uint32_t* nzpos,
                              uint64_t r, uint64_t* T) {
    uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (i >= r) return;
    uint64_t t = thr[idx[i]];
    if (t != 0) T[nzpos[i]] = t;
}

__device__ __forceinline__ uint64_t upper_bound_u64(const uint64_t* a, uint64_t n, uint64_t x) {
    uint64_t lo = 0, hi = n;  // first index with a[idx] > x
    while (lo < hi) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] <= x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// one thread per byte value: directory range, F[c] and where F[c] lands
__global__ void k_letters(const uint8_t* Hs, const uint64_t* LFs, const uint64_t* S, uint64_t r,
                          uint64_t n, LetterInfo* out) {
    int c = threadIdx.x;
    // lower_bound / upper_bound of c in the sorted head array
    uint64_t lo = 0, hi = r;
    while (lo < hi) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        if (Hs[mid] < c)
            lo = mid + 1;
        else
            hi = mid;
    }
    uint64_t qbeg = lo;
    hi = r;
    while (lo < hi) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        if (Hs[mid] <= c)
            lo = mid + 1;
        else
            hi = mid;
    }
    uint64_t qend = lo;
    // F[c] = number of characters smaller than c = LF image of the first c-run start
    uint64_t F = qbeg < r ? LFs[qbeg] : n;
    uint64_t frun = F >= n ? r : upper_bound_u64(S, r, F) - 1;
    LetterInfo li;
    li.qbeg = (uint32_t)qbeg;
    li.qend = (uint32_t)qend;
    li.frun = (uint32_t)frun;
    li.bmul = 0;  // the fat table's geometry is decided on the host (flatten_on_device)
    li.foff = F >= n ? 0 : F - S[frun];
    li.fbase = 0;
    out[c] = li;
}

// one thread per run, in (letter, run index) order: the 16-byte row of the run and the
// jump row of its directory position
__global__ void k_build_rows(const uint32_t* Qall, const uint8_t* Hs, const uint64_t* LFs,
                             const uint64_t* S, const uint64_t* lens, const uint32_t* nzpos,
                             const uint64_t* T, uint64_t nz_total, const uint64_t* ds,
                             const uint64_t* de, const LetterInfo* letters, const uint8_t* Hrun,
                             uint64_t r, uint64_t n, int compact, Row* rows, JumpRow* dirrows, uint32_t* dirdocs,
                             uint32_t* rundocs, unsigned long long* err, const uint8_t* cont) {
    uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (i >= r) return;
    uint32_t k = Qall[i];
    uint32_t c = Hs[i];
    LetterInfo li = letters[c];
    // LF(S[k]) = F[c] + (number of c before run k) = exclusive scan in (letter, run) order
    uint64_t lf = LFs[i];
    uint64_t dst = upper_bound_u64(S, r, lf) - 1;
    const uint64_t soff = lf - S[dst];
    // thr_bv::operator[]: rank = number of c-runs before k; 0 -> 0, else the
    // (rank-1)-th STORED (non-zero) threshold of the letter (thresholds_ds.hpp:484-488)
    uint64_t rank = i - li.qbeg;
    uint64_t thr = 0;
    if (rank > 0) {
        uint64_t base = nzpos[li.qbeg];
        uint64_t cnt = (li.qend < r ? nzpos[li.qend] : nz_total) - base;
        if (rank - 1 >= cnt) {
            atomicAdd(err, 1ull);  // select past the last stored threshold: undefined upstream
        } else {
            thr = T[base + rank - 1];
            if (thr > n) {
                atomicAdd(err, 1ull);
                thr = n;
            }
        }
    }
    // threshold as (run, offset) so that the walk never needs absolute positions
    uint64_t trun = 0, toff = 0;
    if (thr > 0) {
        trun = thr >= n ? r : upper_bound_u64(S, r, thr) - 1;
        toff = thr - S[trun];
    }
    if (compact) {
        uint32_t cum[4];
        uint64_t acc = S[dst + 1] - S[dst] - soff;
        for (int t = 0; t < 4; ++t) {
            cum[t] = acc < CUM_SAT ? (uint32_t)acc : CUM_SAT;
            const uint64_t nx = dst + 1 + t;
            acc = (acc >= CUM_SAT || nx >= r) ? CUM_SAT : acc + (S[nx + 1] - S[nx]);
        }
        // a compact index keeps its rows 32 bytes apart (Row32); the second half is filled in by k_embed_rows
        const Row cr = pack_row_compact(c, (uint32_t)lens[k], (uint32_t)dst, (uint32_t)soff, thr <= S[k], cum);
        Row32* r32 = reinterpret_cast<Row32*>(rows) + k;
        r32->q0 = cr.q0;
        r32->q1 = cr.q1 | ((cont && cont[k]) ? CROW_CONT : 0ull);  // a later piece of a long run
    } else {
        rows[k] = pack_row(c, lens[k], (uint32_t)dst, soff, thr <= S[k], S[dst + 1] - S[dst] - soff);
    }
    // predecessor landing = LF(S[k]) - 1 = LF of the last character of the previous run in
    // directory order
    const bool psame = soff > 0;  // for i == 0 (lf == 0) there is no predecessor: never taken
    dirrows[i] = pack_jumprow(k, (uint32_t)trun, toff, (uint32_t)dst, soff, psame, Hrun[dst], (uint32_t)i);
    if (dirdocs) {
        uint64_t d0 = ds[k], d1 = de[k], dp = i > 0 ? de[Qall[i - 1]] : 0;
        if (d0 > 0xffff || d1 > 0xffff) atomicAdd(err, 1ull);
        dirdocs[i] = (uint32_t)d0 | ((uint32_t)dp << 16);
        rundocs[k] = (uint32_t)d0 | ((uint32_t)d1 << 16);
        if (i + 1 == r) {
            for (int t = 0; t < 4; ++t) dirdocs[r + t] = (uint32_t)d1 << 16;
            for (int t = 0; t < 4; ++t) rundocs[r + t] = 0;
        }
    }
    if (i + 1 == r) {  // sentinel jump row r: LF image n, predecessor = position n-1
        JumpRow sd = pack_jumprow((uint32_t)r, 0, 0, (uint32_t)r, 0, false, 0, (uint32_t)r);
        for (int t = 0; t < 4; ++t) dirrows[r + t] = sd;
    }
}

__global__ void k_sentinel_rows(Row* rows, uint64_t r, int compact) {
    int t = threadIdx.x;
    const uint32_t sat[4] = {CUM_SAT, CUM_SAT, CUM_SAT, CUM_SAT};
    if (t < ROW_PAD) {
        if (compact) {
            const Row cr = pack_row_compact(0, 0xffff, (uint32_t)r, 0, true, sat);
            reinterpret_cast<Row32*>(rows)[r + t] = Row32{cr.q0, cr.q1, 0, 0};
        } else {
            rows[r + t] = pack_row(0, MASK40, (uint32_t)r, 0, true, ROOM_SAT);
        }
    }
}

// Second pass over the compact rows (spx_layout.h, Row32).  k_heads_rows: every row gets the heads of the first
// two runs a step from it can land in; k_embed_rows: then the finished row of run D = LFrun is copied next to the
// row that points to it.  Two kernels: the second reads the q1 the first wrote.  Hrun: heads by run index.
__global__ void k_heads_rows(Row* rows, const uint8_t* Hrun, uint64_t r) {
    const uint64_t k = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (k >= r + ROW_PAD) return;
    Row32* r32 = reinterpret_cast<Row32*>(rows);
    const uint64_t D = r32[k].q0 >> 32;
    r32[k].q1 = crow_with_dheads(r32[k].q1, D < r ? Hrun[D] : 0u, D + 1 < r ? Hrun[D + 1] : 0u);
}
__global__ void k_embed_rows(Row* rows, uint64_t r) {
    const uint64_t k = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (k >= r + ROW_PAD) return;
    Row32* r32 = reinterpret_cast<Row32*>(rows);
    uint64_t D = r32[k].q0 >> 32;
    if (D > r) D = r;  // (padding rows)
    r32[k].e0 = r32[D].q0;
    r32[k].e1 = r32[D].q1;
}

__global__ void k_max_len(const uint64_t* lens, uint64_t r, unsigned long long* out) {
    uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    unsigned long long v = i < r ? lens[i] : 0;
    for (int s = 32; s > 0; s >>= 1) {
        const unsigned long long o = __shfl_xor(v, s);
        v = o > v ? o : v;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, v);
}

// The fat table of letter c, one lane per directory position i of the letter (and one more for what lies behind its last
// run): the slots of the blocks (block of the letter's previous run, block of run Q[i]] all hold the digest of the jump row
// of run Q[i] -- the first run of the letter at or after their block -- and only the last of them can be FAT_SINGLE.
// Samples / dirdocs of that directory position ride in the same slot (DevIndex::fat_stride).  Every first slot of a group
// of 8 also leaves its directory position in fat_js.  blockIdx.y walks the letters that occur (lets[]), blockIdx.x strides
// over the letter's directory positions.
// A lane fills short stretches itself; a stretch of more than 16 slots -- on a real BWT the runs of a letter cluster, and
// between the clusters lie hundreds of thousands of blocks without one -- is filled by the lane's whole wavefront, 64 slots an
// iteration (round 5 found a lane alone walking such gaps one store at a time: 0.16 s of the 0.6 s it takes `spumoni run`
// to load a 5-strain E. coli index from the cache).
// (Until round 6 a first kernel wrote every slot's directory position into fat_j -- 4 bytes a slot, kept for the walk -- and
// a second one, a lane per SLOT, packed the digests from it.)
struct SlotImage {
    uint64_t w[4];  // FatRow, then the slot's second half (fat_stride 32)
};
__global__ void k_fill_fat(const JumpRow* dirrows, const Aux* aux, const uint32_t* Qall, const LetterInfo* letters,
                           const uint8_t* lets, uint64_t r, char* fat, uint32_t stride, uint32_t* fat_js, int force_esc,
                           const Row* rows, int compact) {
    const LetterInfo li = letters[lets[blockIdx.y]];
    const int64_t nslots = (int64_t)letter_slots((uint32_t)r, li.bmul);
    const uint32_t lane = threadIdx.x & 63;
    auto put = [&](int64_t x, const SlotImage& v, uint32_t j) {
        const uint64_t i = li.fbase + (uint64_t)x;
        uint64_t* slot = reinterpret_cast<uint64_t*>(fat + i * stride);
        slot[0] = v.w[0];
        slot[1] = v.w[1];
        if (stride == 32) {
            slot[2] = v.w[2];
            slot[3] = v.w[3];
        }
        if ((i & (FJ_GROUP - 1)) == 0) fat_js[i >> FJ_SHIFT] = j;
    };
    // [lo, hi] := v, by the lane itself when the stretch is short, by the wavefront otherwise (every lane of the wavefront
    // calls this in every round: the trip count below is the same for all of them)
    auto fill = [&](bool have, int64_t lo, int64_t hi, const SlotImage& v, uint32_t j) {
        const bool big = have && hi - lo >= 16;
        if (have && !big)
            for (int64_t x = lo; x <= hi; ++x) put(x, v, j);
        uint64_t todo = __builtin_amdgcn_ballot_w64(big);
        while (todo != 0) {
            const int l = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const int64_t glo = __shfl((long long)lo, l), ghi = __shfl((long long)hi, l);
            SlotImage gv;
#pragma unroll
            for (int t = 0; t < 4; ++t) gv.w[t] = (uint64_t)__shfl((long long)v.w[t], l);
            const uint32_t gj = (uint32_t)__shfl((int)j, l);
            for (int64_t x = glo + lane; x <= ghi; x += 64) put(x, gv, gj);
        }
    };
    const uint64_t n_i = (uint64_t)li.qend - li.qbeg + 1, step = (uint64_t)gridDim.x * TPB;
    const uint64_t rounds = (n_i + step - 1) / step;
    for (uint64_t it = 0; it < rounds; ++it) {
        const uint64_t t = blockIdx.x * (uint64_t)TPB + threadIdx.x + it * step;
        const bool live = t < n_i;
        const uint32_t j = li.qbeg + (uint32_t)(live ? t : 0);
        const bool tail = j >= li.qend;  // behind the letter's last run: no successor
        int64_t b = nslots - 1, pb = -1;
        bool single = false;
        SlotImage plain{{0, 0, 0, 0}}, last{{0, 0, 0, 0}};
        if (live) {
            if (!tail) {
                b = fat_block(Qall[j], li.bmul);
                single = (j + 1 >= li.qend) || fat_block(Qall[j + 1], li.bmul) > b;
            }
            pb = j > li.qbeg ? (int64_t)fat_block(Qall[j - 1], li.bmul) : -1;
            // Hp: the head of the run a predecessor jump lands in when that is not the successor's landing run
            const JumpRow jr = dirrows[j];
            const uint32_t srun = jr_sLFrun(jr);
            uint32_t Hp = 0;
            if (!jr_psame(jr) && srun > 0 && srun <= r) {
                const Row& pr = *reinterpret_cast<const Row*>(reinterpret_cast<const char*>(rows) +
                                                               (uint64_t)(srun - 1) * (compact ? sizeof(Row32) : sizeof(Row)));
                Hp = compact ? crow_H(pr) : row_H(pr);
            }
            const FatRow f0 = pack_fatrow(jr, tail, j <= li.qbeg, force_esc != 0, false, Hp, j);
            const FatRow f1 = pack_fatrow(jr, tail, j <= li.qbeg, force_esc != 0, single, Hp, j);
            plain.w[0] = f0.w0, plain.w[1] = f0.w1;
            if (aux) {
                const Aux a = aux[j];
                plain.w[2] = a.a0, plain.w[3] = a.a1;
            } else if (stride == 32 && srun <= r) {  // (PML-only, compact rows) the row of the landing run rides along: spx_walk_fast.inc, lrow
                const Row lr = *reinterpret_cast<const Row*>(reinterpret_cast<const char*>(rows) + (uint64_t)srun * sizeof(Row32));
                plain.w[2] = lr.q0, plain.w[3] = lr.q1;
            }
            last = plain;
            last.w[0] = f1.w0, last.w[1] = f1.w1;
        }
        // the blocks before the run's own (or, behind the last run, all that are left), then the run's own block
        fill(live, pb + 1, tail ? b : b - 1, plain, j);
        if (live && !tail && pb < b) put(b, last, j);  // (pb == b: the block's slot belongs to an earlier run of the letter)
    }
}

__global__ void k_copy_q(const uint32_t* Qall, uint64_t r, uint32_t* q_alloc) {
    uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (i < r) q_alloc[i + 1] = Qall[i];
    if (i == 0) q_alloc[0] = 0;
    if (i < Q_PAD) q_alloc[r + 1 + i] = 0xffffffffu;
}

// MS samples in directory order: entry i = {samples_start[Q[i]], samples_last[Q[i-1]]};
// plus samples_start by run index for the "byte >= 128 sitting on its own run" case
// side data by directory position: samples pair and doc word packed into one 16-byte record
__global__ void k_pack_aux(const SamplePair* samples, const uint32_t* dirdocs, uint64_t count, Aux* aux) {
    uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (i >= count) return;
    const SamplePair sp = samples ? samples[i] : SamplePair{0, 0};
    aux[i] = pack_aux(sp.ss, sp.se, dirdocs ? dirdocs[i] : 0u);
}

__global__ void k_samples(const uint64_t* ssa, const uint64_t* esa, const uint32_t* Qall, uint64_t r,
                          SamplePair* out, uint64_t* ss_by_run) {
    uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (i > r) return;
    SamplePair sp;
    sp.ss = i < r ? ssa[Qall[i]] : 0;
    sp.se = i > 0 ? esa[Qall[i - 1]] : 0;
    out[i] = sp;
    if (i < r) ss_by_run[i] = ssa[i];
    if (i == r) {
        ss_by_run[r] = 0;
        ss_by_run[r + 1] = 0;
    }
}

}  // namespace

// The fat table and fat_js from what the index already holds on the device: letters (geometry), Q, dirrows, aux.
// Run by the flatten step and by spx_index_load_flat: the cache file does not carry the table (it is most of the
// index -- 160 of 230 GB at 10^9 runs -- and takes a fraction of a second to rebuild, seconds to read).
int build_fat(spx_index* ix) {
    hipStream_t st = nullptr;
    const uint64_t r = ix->view.r, nfat = ix->view.nfat;
    const uint32_t fat_stride = ix->view.fat_stride;
    std::vector<LetterInfo> hl(256);
    SPX_HIP(hipMemcpy(hl.data(), ix->letters, 256 * sizeof(LetterInfo), hipMemcpyDeviceToHost));
    std::vector<uint8_t> lets;
    uint64_t most_runs = 0;
    for (int c = 0; c < 256; ++c)
        if (hl[c].qend > hl[c].qbeg) {
            lets.push_back((uint8_t)c);
            most_runs = std::max<uint64_t>(most_runs, hl[c].qend - hl[c].qbeg);
            if ((hl[c].fbase & (FJ_GROUP - 1)) != 0 || hl[c].fbase + letter_slots((uint32_t)r, hl[c].bmul) > nfat) {
                set_error("index letters describe a fat table that does not fit its %llu slots", (unsigned long long)nfat);
                return SPX_E_FORMAT;
            }
        }
    if (lets.empty()) {
        set_error("index without letters");
        return SPX_E_FORMAT;
    }
    DevBuf dl;
    SPX_HIP(dl.alloc(256));
    SPX_HIP(hipMemcpyAsync(dl.p, lets.data(), lets.size(), hipMemcpyHostToDevice, st));
    if (ix->fat_js) (void)hipFree(ix->fat_js);
    if (ix->fat) (void)hipFree(ix->fat);
    ix->fat_js = nullptr;
    ix->fat = nullptr;
    const bool timing = getenv("SPX_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    SPX_ALLOC0(ix->fat_js, fatjs_count(nfat) * 4 + 64);
    SPX_ALLOC0(ix->fat, (nfat + 2) * (uint64_t)fat_stride);
    if (timing) {
        SPX_HIP(hipStreamSynchronize(st));
        fprintf(stderr, "[spx] build_fat: allocate + zero %.1f GB: %.3f s\n", (nfat * (fat_stride + 4.0 / FJ_GROUP)) / 1e9, now() - t0);
        t0 = now();
    }
    const uint32_t* Qall = ix->q_alloc + 1;
    const unsigned gc = (most_runs + 1) / TPB + 1 < (1u << 20) ? (unsigned)((most_runs + 1) / TPB + 1) : (1u << 20);
    k_fill_fat<<<dim3(gc, (unsigned)lets.size()), TPB, 0, st>>>(ix->dirrows, ix->aux, Qall, ix->letters, dl.as<uint8_t>(), r, ix->fat,
                                                                 fat_stride, ix->fat_js, getenv("SPX_FAT_ALL_ESC") ? 1 : 0, ix->rows,
                                                                 (int)ix->view.compact);
    SPX_HIP(hipGetLastError());
    SPX_HIP(hipStreamSynchronize(st));
    if (timing) fprintf(stderr, "[spx] build_fat: k_fill_fat %.3f s\n", now() - t0);
    ix->arr_bytes[A_FAT] = (nfat + 2) * (uint64_t)fat_stride;
    ix->arr_bytes[A_FATJ] = fatjs_count(nfat) * 4 + 64;
    bind_view(ix);
    return SPX_OK;
}

// ---- long runs: pieces (round 3) ---------------------------------------------------------------------------------
// The compact row encoding (16-bit lengths and offsets; Row32 and k_walk_fast need it) holds runs shorter than 2^16.
// One longer run used to switch the WHOLE index to the general encoding (939 -> 774 M reads/s on the bench index).
// Instead such a run is laid out as consecutive PIECES of at most 65 535 positions with the same head.  Nothing the
// walk computes changes: a step inside a piece is an LF step from that piece's start; a jump to the letter lands on
// the run's first piece (successor) or its last piece's last position (predecessor) -- the positions the reference
// computes -- and never consults a later piece's threshold, which is the run's own (at least 1: stored thresholds are
// non-zero, so the pieces keep thr_bv's rank -> stored-value order intact, thresholds_ds.hpp:484-488) and only matters
// for a byte >= 128 sitting on the run (Appendix C1), where it compares positions exactly as the reference does.
// Samples and document ids of every piece are the run's.  Pieces after the first carry the CONT bit in their row: the
// text rebuild (LF chains from run starts) neither starts nor stops there.  The API still reports the file's r.
// Not applied (the general encoding is kept) when a letter's non-first run has a zero threshold -- thr_bv then skips
// stored values and inserted pieces would shift which one a later run reads.
// Balanced pieces (round 3, late): a piece also ends where its LF image has covered `span` runs (spx_layout.h:
// for_each_piece), so that a step is never more than span - 4 rows away from the row the compact encoding sends it to
// (tools/ff_model.py; tests/piece_cuts_check.cpp holds the cut rule against its specification on the CPU).  The images
// are known after a pass of their own over the run list (S, the heads' order, LF of every run start): it is made for an
// index whose longest run has SPX_BALANCE_MIN_RUN (2048) positions or more -- an image covers no more runs than the run
// has positions -- and cuts where an image covers more than SPX_BALANCE_SPAN (8; 0: never) runs.  The new pieces are run
// boundaries themselves and may push another image past the bound again, so the pass is repeated on its own output until
// nothing is cut or SPX_BALANCE_PASSES (4) were made (the fourth cuts a few rows in 10^4: profiles/r03_balanced_pieces_passes.txt).

__global__ void k_scatter_u64(const uint64_t* src, const uint32_t* idx, uint64_t r, uint64_t* dst) {
    uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (i < r) dst[idx[i]] = src[i];
}

struct PieceCounter {
    SPX_HD void operator()(uint64_t, uint64_t) const {}
};

// image of run k: a = run that holds LF(S[k]), nb = run starts inside the image (0 when the pass was not made)
__device__ __forceinline__ void image_of(const uint64_t* S, const uint64_t* LFk, uint64_t r, uint64_t k, uint64_t len, uint64_t& lf,
                                         uint64_t& a, uint64_t& nb) {
    lf = 0;
    a = 0;
    nb = 0;
    if (!S || len == 0 || len > MASK40) return;
    lf = LFk[k];
    a = run_of_position(S, r, lf);
    nb = run_of_position(S, r, lf + len - 1) - a;
}

// zero_thr counts the zero thresholds of runs that are NOT their letter's first (a count over all runs held
// against the number of letters balances out when one letter's first run has a threshold and another's later run has none)
__global__ void k_piece_count(const uint8_t* heads, const uint64_t* lens, const uint64_t* thr, uint64_t r, const uint64_t* S,
                              const uint64_t* LFk, uint32_t span, const uint32_t* first_of_letter, uint32_t* pieces,
                              unsigned long long* zero_thr, unsigned long long* span_max) {
    const uint64_t k = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (k >= r) return;
    uint64_t lf, a, nb;
    image_of(S, LFk, r, k, lens[k], lf, a, nb);
    pieces[k] = for_each_piece(lens[k], lf, S, a, nb, span, PieceCounter{});
    const uint32_t h = heads[k] <= 1 ? 1 : heads[k];
    if (thr[k] == 0 && first_of_letter[h] != (uint32_t)k) atomicAdd(zero_thr, 1ull);
    if (S && nb + 1 > span) atomicMax(span_max, (unsigned long long)(nb + 1));
}

__global__ void k_first_of_letter(const uint8_t* heads, uint64_t r, uint32_t* first_of_letter) {
    const uint64_t k = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (k >= r) return;
    const uint32_t h = heads[k] <= 1 ? 1 : heads[k];
    // (the plain read first: once the early runs have been through, hardly any run still has to use the atomic)
    if ((uint32_t)k < __builtin_nontemporal_load(&first_of_letter[h])) atomicMin(&first_of_letter[h], (uint32_t)k);
}

struct PieceWriter {
    uint8_t head;
    uint64_t thr, ssa, esa, ds, de;
    uint64_t o;  // where the next piece goes
    bool first;
    uint8_t cont0;  // the run is itself a later piece of a run (a pass after the first)
    uint8_t* heads2;
    uint64_t *lens2, *thr2, *ssa2, *esa2, *ds2, *de2;
    uint8_t* cont;
    SPX_HD void operator()(uint64_t, uint64_t plen) {
        heads2[o] = head;
        lens2[o] = plen;
        thr2[o] = first ? thr : (thr ? thr : 1);
        if (ssa2) ssa2[o] = ssa;
        if (esa2) esa2[o] = esa;
        if (ds2) ds2[o] = ds;
        if (de2) de2[o] = de;
        cont[o] = first ? cont0 : 1;
        first = false;
        ++o;
    }
};

__global__ void k_piece_fill(const uint8_t* heads, const uint64_t* lens, const uint64_t* thr, const uint64_t* ssa,
                             const uint64_t* esa, const uint64_t* ds, const uint64_t* de, const uint8_t* cont_in,
                             const uint32_t* first_piece, uint64_t r, const uint64_t* S, const uint64_t* LFk, uint32_t span,
                             uint8_t* heads2, uint64_t* lens2, uint64_t* thr2, uint64_t* ssa2, uint64_t* esa2, uint64_t* ds2,
                             uint64_t* de2, uint8_t* cont) {
    const uint64_t k = blockIdx.x * (uint64_t)TPB + threadIdx.x;
    if (k >= r) return;
    uint64_t lf, a, nb;
    image_of(S, LFk, r, k, lens[k], lf, a, nb);
    PieceWriter w{heads[k], thr[k], ssa2 ? ssa[k] : 0, esa2 ? esa[k] : 0, ds2 ? ds[k] : 0, de2 ? de[k] : 0,
                  first_piece[k], true, (uint8_t)(cont_in ? cont_in[k] : 0), heads2, lens2, thr2, ssa2, esa2, ds2, de2, cont};
    for_each_piece(lens[k], lf, S, a, nb, span, w);
}

static int flatten_core(spx_index* ix, uint64_t r, const uint8_t* d_heads, const uint64_t* d_lens, const uint64_t* d_thr,
                        const uint64_t* d_ssa, const uint64_t* d_esa, const uint64_t* d_ds, const uint64_t* d_de,
                        const uint8_t* d_cont, const std::function<void()>& release_inputs);

namespace {

struct RunList {  // a run list on the device (the caller's arrays, or a generation of pieces)
    uint64_t r = 0;
    const uint8_t* heads = nullptr;
    const uint64_t *lens = nullptr, *thr = nullptr, *ssa = nullptr, *esa = nullptr, *ds = nullptr, *de = nullptr;
    const uint8_t* cont = nullptr;  // pieces after a run's first one (nullptr: none)
};

struct PieceGen {  // storage of one generation of pieces
    DevBuf h, l, t, s, e, ds, de, c;
    void release() {
        for (DevBuf* b : {&h, &l, &t, &s, &e, &ds, &de, &c}) {
            if (b->p) (void)hipFree(b->p);
            b->p = nullptr;
        }
    }
};

// S (run starts, S[r] = n) and LF of every run start by run, for the images of a pass; ok = false: the run list is
// not one flatten_core will accept (a run of length 0, a BWT too long) -- it says so itself
int images_of_runs(const RunList& in, DevBuf& Sb, DevBuf& LFk, bool& ok, hipStream_t st) {
    const uint64_t r = in.r;
    ok = false;
    DevBuf H, iota, Hs, Qall, ls, LFs, tmp, err;
    SPX_HIP(H.alloc(r));
    SPX_HIP(iota.alloc(r * 4));
    SPX_HIP(Hs.alloc(r));
    SPX_HIP(Qall.alloc(r * 4));
    SPX_HIP(Sb.alloc((r + 1) * 8));
    SPX_HIP(err.alloc(8));
    SPX_HIP(hipMemsetAsync(err.p, 0, 8, st));
    k_norm_heads<<<nblocks(r), TPB, 0, st>>>(in.heads, in.lens, r, H.as<uint8_t>(), iota.as<uint32_t>(), err.as<unsigned long long>());
    size_t tb = 0, tb2 = 0;
    SPX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in.lens, Sb.as<uint64_t>(), r, st));
    SPX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, H.as<uint8_t>(), Hs.as<uint8_t>(), iota.as<uint32_t>(),
                                               Qall.as<uint32_t>(), r, 0, 8, st));
    if (tb2 > tb) tb = tb2;
    SPX_HIP(tmp.alloc(tb + 256));
    size_t tbs = tb;
    SPX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tbs, in.lens, Sb.as<uint64_t>(), r, st));
    uint64_t last_s = 0, last_len = 0;
    unsigned long long bad = 0;
    SPX_HIP(hipMemcpyAsync(&last_s, Sb.as<uint64_t>() + (r - 1), 8, hipMemcpyDeviceToHost, st));
    SPX_HIP(hipMemcpyAsync(&last_len, in.lens + (r - 1), 8, hipMemcpyDeviceToHost, st));
    SPX_HIP(hipMemcpyAsync(&bad, err.p, 8, hipMemcpyDeviceToHost, st));
    SPX_HIP(hipStreamSynchronize(st));
    const uint64_t n = last_s + last_len;
    if (bad || n > MASK40 - 2) return SPX_OK;
    SPX_HIP(hipMemcpyAsync(Sb.as<uint64_t>() + r, &n, 8, hipMemcpyHostToDevice, st));
    tbs = tb;
    SPX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tbs, H.as<uint8_t>(), Hs.as<uint8_t>(), iota.as<uint32_t>(),
                                               Qall.as<uint32_t>(), r, 0, 8, st));
    SPX_HIP(ls.alloc(r * 8));
    SPX_HIP(LFs.alloc((r + 1) * 8));
    SPX_HIP(LFk.alloc(r * 8));
    k_gather_u64<<<nblocks(r), TPB, 0, st>>>(in.lens, Qall.as<uint32_t>(), r, ls.as<uint64_t>());
    tbs = tb;
    SPX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tbs, ls.as<uint64_t>(), LFs.as<uint64_t>(), r, st));
    k_scatter_u64<<<nblocks(r), TPB, 0, st>>>(LFs.as<uint64_t>(), Qall.as<uint32_t>(), r, LFk.as<uint64_t>());
    SPX_HIP(hipGetLastError());
    SPX_HIP(hipStreamSynchronize(st));
    ok = true;
    return SPX_OK;
}

}  // namespace

// release_inputs (may be empty): frees the caller's device copies of the raw arrays; called once, when nothing reads them any
// more and BEFORE the fat table is sized -- an index loaded from files gives the table the memory its raw arrays held
int flatten_on_device(spx_index* ix, const uint8_t* d_heads, const uint64_t* d_lens,
                      const uint64_t* d_thr, const uint64_t* d_ssa, const uint64_t* d_esa,
                      const uint64_t* d_ds, const uint64_t* d_de, const std::function<void()>& release_inputs) {
    const uint64_t r = ix->r;
    if (r == 0 || r > 0xfffffff0ull) {
        set_error("number of runs %llu out of range (1 .. 2^32-16)", (unsigned long long)r);
        return SPX_E_FORMAT;
    }
    hipStream_t st = nullptr;
    unsigned long long max_len = 0;
    {
        DevBuf ml;
        SPX_HIP(ml.alloc(8));
        SPX_HIP(hipMemsetAsync(ml.p, 0, 8, st));
        k_max_len<<<nblocks(r), TPB, 0, st>>>(d_lens, r, ml.as<unsigned long long>());
        SPX_HIP(hipMemcpyAsync(&max_len, ml.p, 8, hipMemcpyDeviceToHost, st));
        SPX_HIP(hipStreamSynchronize(st));
    }
    uint32_t span = 8;
    uint64_t min_run = 2048;
    int passes = 4;
    if (const char* e = getenv("SPX_BALANCE_SPAN")) span = (uint32_t)atoi(e);
    if (const char* e = getenv("SPX_BALANCE_MIN_RUN")) min_run = (uint64_t)atoll(e);
    if (const char* e = getenv("SPX_BALANCE_PASSES")) passes = atoi(e) < 1 ? 1 : atoi(e);
    const bool want_balance = span > 0 && max_len >= min_run && max_len <= MASK40;
    RunList orig;
    orig.r = r;
    orig.heads = d_heads;
    orig.lens = d_lens;
    orig.thr = d_thr;
    orig.ssa = d_ssa;
    orig.esa = d_esa;
    orig.ds = d_ds;
    orig.de = d_de;
    bool released = false;
    auto release_once = [&] {
        if (!released && release_inputs) release_inputs();
        released = true;
    };
    auto as_it_is = [&] { return flatten_core(ix, r, d_heads, d_lens, d_thr, d_ssa, d_esa, d_ds, d_de, nullptr, release_once); };
    if (getenv("SPX_ROWS_WIDE") || getenv("SPX_NO_PIECES") || (max_len <= PIECE_MAX && !want_balance)) return as_it_is();
    const bool timing = getenv("SPX_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

    // Passes over the run list: pieces for length (the first pass) and for their images (every pass: the pieces of
    // one pass are run boundaries in the next, which may push an image past the bound again; until nothing is cut
    // or SPX_BALANCE_PASSES passes were made).
    PieceGen gen[2];
    RunList cur = orig;
    for (int pass = 0; pass < passes; ++pass) {
        const double t_pass = now();
        DevBuf Sb, LFk;
        bool balance = want_balance;
        if (balance) {
            // the passes want 45 B per run of scratch and a second copy of the run arrays (up to 50 B per run): an index
            // that leaves no room for that is flattened without them rather than not at all
            size_t mem_free = 0, mem_total = 0;
            SPX_HIP(hipMemGetInfo(&mem_free, &mem_total));
            if ((double)mem_free < 110.0 * (double)cur.r + (double)(256u << 20)) {
                balance = false;
                if (timing) fprintf(stderr, "[spx] pieces, pass %d: %.1f GB free, too little for the image pass over %llu rows: not balanced\n",
                                    pass, (double)mem_free / 1e9, (unsigned long long)cur.r);
            }
        }
        if (balance) {
            const int rc = images_of_runs(cur, Sb, LFk, balance, st);
            if (rc != SPX_OK) return rc;
        }
        if (pass == 0 && !balance && max_len <= PIECE_MAX) return as_it_is();
        if (pass > 0 && !balance) break;
        const uint64_t* d_S = balance ? Sb.as<uint64_t>() : nullptr;
        const uint64_t* d_LFk = balance ? LFk.as<uint64_t>() : nullptr;
        const uint64_t rc_ = cur.r;
        // how many pieces, and may the run list be extended at all?
        DevBuf pieces, first_piece, cnt, present, tmp;
        SPX_HIP(pieces.alloc((rc_ + 1) * 4));
        SPX_HIP(first_piece.alloc((rc_ + 1) * 4));
        SPX_HIP(cnt.alloc(16));
        SPX_HIP(present.alloc(256 * 4));
        SPX_HIP(hipMemsetAsync(cnt.p, 0, 16, st));
        SPX_HIP(hipMemsetAsync(present.p, 0xff, 256 * 4, st));
        SPX_HIP(hipMemsetAsync(pieces.as<uint32_t>() + rc_, 0, 4, st));
        k_first_of_letter<<<nblocks(rc_), TPB, 0, st>>>(cur.heads, rc_, present.as<uint32_t>());
        k_piece_count<<<nblocks(rc_), TPB, 0, st>>>(cur.heads, cur.lens, cur.thr, rc_, d_S, d_LFk, balance ? span : 0,
                                                     present.as<uint32_t>(), pieces.as<uint32_t>(), cnt.as<unsigned long long>(),
                                                     cnt.as<unsigned long long>() + 1);
        size_t tb = 0;
        SPX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, pieces.as<uint32_t>(), first_piece.as<uint32_t>(), rc_ + 1, st));
        SPX_HIP(tmp.alloc(tb + 256));
        SPX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, pieces.as<uint32_t>(), first_piece.as<uint32_t>(), rc_ + 1, st));
        unsigned long long zero_thr = 0, span_max = 0;
        uint32_t r2_32 = 0;
        SPX_HIP(hipMemcpyAsync(&zero_thr, cnt.p, 8, hipMemcpyDeviceToHost, st));
        SPX_HIP(hipMemcpyAsync(&span_max, cnt.as<unsigned long long>() + 1, 8, hipMemcpyDeviceToHost, st));
        SPX_HIP(hipMemcpyAsync(&r2_32, first_piece.as<uint32_t>() + rc_, 4, hipMemcpyDeviceToHost, st));
        SPX_HIP(hipStreamSynchronize(st));
        const uint64_t r2 = r2_32;
        if (timing)
            fprintf(stderr, "[spx] pieces, pass %d: %llu rows -> %llu%s; longest image %llu runs (0: none over %u), longest run %llu; %.3f s\n",
                    pass, (unsigned long long)rc_, (unsigned long long)r2, balance ? " (balanced)" : "", span_max, span, max_len,
                    now() - t_pass);
        // a non-first run with a zero threshold (thr_bv skips stored values: inserted pieces would shift which one a
        // later run reads) / too many pieces: the run list stays as it is (the first pass: as the caller gave it)
        if (zero_thr > 0 || r2 > 0xfffffff0ull || r2 < rc_) {
            if (pass == 0) return as_it_is();
            break;
        }
        if (r2 == rc_) break;  // nothing to cut (any more)
        PieceGen& g = gen[pass & 1];
        SPX_HIP(g.h.alloc(r2));
        SPX_HIP(g.l.alloc(r2 * 8));
        SPX_HIP(g.t.alloc(r2 * 8));
        if (cur.ssa) SPX_HIP(g.s.alloc(r2 * 8));
        if (cur.esa) SPX_HIP(g.e.alloc(r2 * 8));
        if (cur.ds) SPX_HIP(g.ds.alloc(r2 * 8));
        if (cur.de) SPX_HIP(g.de.alloc(r2 * 8));
        SPX_HIP(g.c.alloc(r2));
        k_piece_fill<<<nblocks(rc_), TPB, 0, st>>>(cur.heads, cur.lens, cur.thr, cur.ssa, cur.esa, cur.ds, cur.de, cur.cont,
                                                    first_piece.as<uint32_t>(), rc_, d_S, d_LFk, balance ? span : 0, g.h.as<uint8_t>(),
                                                    g.l.as<uint64_t>(), g.t.as<uint64_t>(), cur.ssa ? g.s.as<uint64_t>() : nullptr,
                                                    cur.esa ? g.e.as<uint64_t>() : nullptr, cur.ds ? g.ds.as<uint64_t>() : nullptr,
                                                    cur.de ? g.de.as<uint64_t>() : nullptr, g.c.as<uint8_t>());
        SPX_HIP(hipGetLastError());
        SPX_HIP(hipStreamSynchronize(st));
        RunList next;
        next.r = r2;
        next.heads = g.h.as<uint8_t>();
        next.lens = g.l.as<uint64_t>();
        next.thr = g.t.as<uint64_t>();
        next.ssa = cur.ssa ? g.s.as<uint64_t>() : nullptr;
        next.esa = cur.esa ? g.e.as<uint64_t>() : nullptr;
        next.ds = cur.ds ? g.ds.as<uint64_t>() : nullptr;
        next.de = cur.de ? g.de.as<uint64_t>() : nullptr;
        next.cont = g.c.as<uint8_t>();
        cur = next;
        if (pass == 0) release_once();  // (the caller's arrays: every later pass reads a generation of pieces)
        gen[(pass + 1) & 1].release();  // the generation this one was cut from
        if (!balance) break;            // (pieces for length only: nothing a further pass would change)
        // (Sb, LFk, pieces, ... go out of scope here: the table flatten_core builds sizes itself against free memory)
    }
    if (cur.cont == nullptr) return as_it_is();
    // (the last generation of pieces is this function's own: gone, like the caller's arrays, before the table is sized)
    return flatten_core(ix, cur.r, cur.heads, cur.lens, cur.thr, cur.ssa, cur.esa, cur.ds, cur.de, cur.cont, [&] {
        gen[0].release();
        gen[1].release();
    });
}

static int flatten_core(spx_index* ix, const uint64_t r, const uint8_t* d_heads, const uint64_t* d_lens, const uint64_t* d_thr,
                        const uint64_t* d_ssa, const uint64_t* d_esa, const uint64_t* d_ds, const uint64_t* d_de,
                        const uint8_t* d_cont, const std::function<void()>& release_inputs) {
    if (r == 0 || r > 0xfffffff0ull) {
        set_error("number of runs %llu out of range (1 .. 2^32-16)", (unsigned long long)r);
        return SPX_E_FORMAT;
    }
    hipStream_t st = nullptr;
    DevBuf H, iota, Hs, Qall, S, ls, LFs, flag, nzpos, T, tmp, err;
    SPX_HIP(H.alloc(r));
    SPX_HIP(iota.alloc(r * 4));
    SPX_HIP(Hs.alloc(r));
    SPX_HIP(Qall.alloc(r * 4));
    SPX_HIP(S.alloc((r + 1) * 8));
    SPX_HIP(err.alloc(8));
    SPX_HIP(hipMemsetAsync(err.p, 0, 8, st));

    k_norm_heads<<<nblocks(r), TPB, 0, st>>>(d_heads, d_lens, r, H.as<uint8_t>(),
                                              iota.as<uint32_t>(), err.as<unsigned long long>());
    // S = exclusive scan of run lengths; n = S[r]
    size_t tb = 0, tb2 = 0;
    SPX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_lens, S.as<uint64_t>(), r, st));
    SPX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, H.as<uint8_t>(), Hs.as<uint8_t>(),
                                               iota.as<uint32_t>(), Qall.as<uint32_t>(), r, 0, 8, st));
    if (tb2 > tb) tb = tb2;
    SPX_HIP(tmp.alloc(tb + 256));
    size_t tbs = tb;
    SPX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tbs, d_lens, S.as<uint64_t>(), r, st));
    uint64_t last_s = 0, last_len = 0;
    SPX_HIP(hipMemcpyAsync(&last_s, S.as<uint64_t>() + (r - 1), 8, hipMemcpyDeviceToHost, st));
    SPX_HIP(hipMemcpyAsync(&last_len, d_lens + (r - 1), 8, hipMemcpyDeviceToHost, st));
    SPX_HIP(hipStreamSynchronize(st));
    const uint64_t n = last_s + last_len;
    if (n > MASK40 - 2) {
        set_error("BWT length %llu exceeds the 40-bit position field", (unsigned long long)n);
        return SPX_E_FORMAT;
    }
    ix->n = n;
    SPX_HIP(hipMemcpyAsync(S.as<uint64_t>() + r, &n, 8, hipMemcpyHostToDevice, st));

    // runs grouped by head letter, run order kept: the per-letter directories Q_c
    tbs = tb;
    SPX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tbs, H.as<uint8_t>(), Hs.as<uint8_t>(),
                                               iota.as<uint32_t>(), Qall.as<uint32_t>(), r, 0, 8, st));
    SPX_HIP(hipStreamSynchronize(st));
    (void)hipFree(iota.p);
    iota.p = nullptr;

    // LF image of every run start = exclusive scan of lengths in (letter, run) order
    SPX_HIP(ls.alloc(r * 8));
    SPX_HIP(LFs.alloc((r + 1) * 8));
    k_gather_u64<<<nblocks(r), TPB, 0, st>>>(d_lens, Qall.as<uint32_t>(), r, ls.as<uint64_t>());
    tbs = tb;
    SPX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tbs, ls.as<uint64_t>(), LFs.as<uint64_t>(), r, st));
    SPX_HIP(hipMemcpyAsync(LFs.as<uint64_t>() + r, &n, 8, hipMemcpyHostToDevice, st));
    SPX_HIP(hipStreamSynchronize(st));
    (void)hipFree(ls.p);
    ls.p = nullptr;

    // stored (non-zero) thresholds per letter
    SPX_HIP(flag.alloc(r * 4));
    SPX_HIP(nzpos.alloc((r + 1) * 4));
    k_nonzero_flag<<<nblocks(r), TPB, 0, st>>>(d_thr, Qall.as<uint32_t>(), r, flag.as<uint32_t>());
    size_t tb3 = 0;
    SPX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb3, flag.as<uint32_t>(), nzpos.as<uint32_t>(), r, st));
    if (tb3 > tb) {
        set_error("scan scratch mis-sized");
        return SPX_E_HIP;
    }
    tbs = tb;
    SPX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tbs, flag.as<uint32_t>(), nzpos.as<uint32_t>(), r, st));
    uint32_t last_pos = 0, last_flag = 0;
    SPX_HIP(hipMemcpyAsync(&last_pos, nzpos.as<uint32_t>() + (r - 1), 4, hipMemcpyDeviceToHost, st));
    SPX_HIP(hipMemcpyAsync(&last_flag, flag.as<uint32_t>() + (r - 1), 4, hipMemcpyDeviceToHost, st));
    SPX_HIP(hipStreamSynchronize(st));
    const uint64_t nz_total = (uint64_t)last_pos + last_flag;
    (void)hipFree(flag.p);
    flag.p = nullptr;
    SPX_HIP(T.alloc((nz_total + 1) * 8));
    k_compact_thr<<<nblocks(r), TPB, 0, st>>>(d_thr, Qall.as<uint32_t>(), nzpos.as<uint32_t>(), r,
                                               T.as<uint64_t>());

    // letters
    SPX_ALLOC0(ix->letters, 256 * sizeof(LetterInfo));
    k_letters<<<1, 256, 0, st>>>(Hs.as<uint8_t>(), LFs.as<uint64_t>(), S.as<uint64_t>(), r, n,
                                  ix->letters);
    std::vector<LetterInfo> hl(256);
    SPX_HIP(hipMemcpyAsync(hl.data(), ix->letters, 256 * sizeof(LetterInfo), hipMemcpyDeviceToHost, st));
    SPX_HIP(hipStreamSynchronize(st));
    uint32_t nletters = 0;
    std::vector<uint8_t> lets;  // the byte values that occur
    for (int c = 0; c < 256; ++c)
        if (hl[c].qend > hl[c].qbeg) {
            nletters++;
            lets.push_back((uint8_t)c);
        }

    // rows + jump rows
    const bool docs = d_ds && d_de;
    DevBuf dirdocs_tmp;  // by directory position j: docS[Q[j]] | docE[Q[j-1]] << 16; packed into aux below
    if (docs) {
        SPX_HIP(dirdocs_tmp.alloc((r + ROW_PAD) * 4));
        SPX_ALLOC0(ix->rundocs, (r + ROW_PAD) * 4);
    }
    unsigned long long max_len = 0;
    {
        DevBuf ml;
        SPX_HIP(ml.alloc(8));
        SPX_HIP(hipMemsetAsync(ml.p, 0, 8, st));
        k_max_len<<<nblocks(r), TPB, 0, st>>>(d_lens, r, ml.as<unsigned long long>());
        SPX_HIP(hipMemcpyAsync(&max_len, ml.p, 8, hipMemcpyDeviceToHost, st));
        SPX_HIP(hipStreamSynchronize(st));
    }
    const int compact = (max_len < 65536 && !getenv("SPX_ROWS_WIDE")) ? 1 : 0;
    const uint64_t row_bytes = compact ? sizeof(Row32) : sizeof(Row);
    SPX_ALLOC0(ix->rows, (r + ROW_PAD) * row_bytes);
    SPX_ALLOC0(ix->dirrows, (r + ROW_PAD) * sizeof(JumpRow));
    k_build_rows<<<nblocks(r), TPB, 0, st>>>(Qall.as<uint32_t>(), Hs.as<uint8_t>(), LFs.as<uint64_t>(),
                                              S.as<uint64_t>(), d_lens, nzpos.as<uint32_t>(),
                                              T.as<uint64_t>(), nz_total, d_ds, d_de, ix->letters,
                                              H.as<uint8_t>(), r, n, compact, ix->rows, ix->dirrows,
                                              docs ? dirdocs_tmp.as<uint32_t>() : nullptr,
                                              ix->rundocs, err.as<unsigned long long>(), compact ? d_cont : nullptr);
    k_sentinel_rows<<<1, 64, 0, st>>>(ix->rows, r, compact);
    if (compact) {
        k_heads_rows<<<nblocks(r + ROW_PAD), TPB, 0, st>>>(ix->rows, H.as<uint8_t>(), r);
        k_embed_rows<<<nblocks(r + ROW_PAD), TPB, 0, st>>>(ix->rows, r);
    }
    SPX_HIP(hipStreamSynchronize(st));
    (void)hipFree(T.p);
    T.p = nullptr;
    (void)hipFree(nzpos.p);
    nzpos.p = nullptr;
    (void)hipFree(LFs.p);
    LFs.p = nullptr;

    // Everything but the fat table first -- the directory, the samples and document ids by directory position, the scalars of
    // the initial state -- so that the table, which takes whatever memory is left within the budget, is sized when the
    // temporaries of this function (14 B per run, 38 with samples) and the caller's copies of the raw arrays are gone
    // (round 6: the declared C4, flattened beside 50 GB of raw arrays and 18 GB of temporaries, got 3.3 slots per run where
    // its budget allows 4.2).
    SPX_ALLOC0(ix->q_alloc, (r + 1 + Q_PAD) * 4);
    k_copy_q<<<nblocks(r > (uint64_t)Q_PAD ? r : Q_PAD), TPB, 0, st>>>(Qall.as<uint32_t>(), r,
                                                                         ix->q_alloc);
    uint64_t bytes = (r + ROW_PAD) * (row_bytes + sizeof(JumpRow)) + (r + 1 + Q_PAD) * 4 + 256 * sizeof(LetterInfo) +
                     (docs ? (r + ROW_PAD) * 4 : 0);
    uint64_t last_esa = 0, last_de = 0, first_ds = 0;
    DevBuf samples_tmp;
    if (d_ssa && d_esa) {
        SPX_HIP(samples_tmp.alloc((r + 2) * sizeof(SamplePair)));
        SPX_ALLOC0(ix->ss_by_run, (r + 4) * 8);
        k_samples<<<nblocks(r + 1), TPB, 0, st>>>(d_ssa, d_esa, Qall.as<uint32_t>(), r, samples_tmp.as<SamplePair>(),
                                                   ix->ss_by_run);
        SPX_HIP(hipMemcpyAsync(&last_esa, d_esa + (r - 1), 8, hipMemcpyDeviceToHost, st));
        bytes += (r + 4) * 8;
        ix->has_samples = true;
    }
    ix->has_docs = docs;
    if (ix->has_samples || docs) {  // samples + doc words of a directory position in one record
        SPX_ALLOC0(ix->aux, (r + 2) * sizeof(Aux));
        k_pack_aux<<<nblocks(r + 1), TPB, 0, st>>>(ix->has_samples ? samples_tmp.as<SamplePair>() : nullptr,
                                                    docs ? dirdocs_tmp.as<uint32_t>() : nullptr, r + 1, ix->aux);
        bytes += (r + 2) * sizeof(Aux);
    }
    if (docs) {
        SPX_HIP(hipMemcpyAsync(&last_de, d_de + (r - 1), 8, hipMemcpyDeviceToHost, st));
        SPX_HIP(hipMemcpyAsync(&first_ds, d_ds, 8, hipMemcpyDeviceToHost, st));
    }
    const bool has_ms = d_ssa && d_esa;
    SPX_HIP(hipStreamSynchronize(st));
    for (DevBuf* t : {&samples_tmp, &dirdocs_tmp, &Qall, &S, &Hs, &H, &tmp}) {
        if (t->p) (void)hipFree(t->p);
        t->p = nullptr;
    }
    if (release_inputs) release_inputs();  // (d_heads .. d_de are not read past this line)
    // Fat-table geometry = how much HBM is traded for speed.  A fat slot answers a jump outright
    // when no c-run lies between its block's start and the walk's run, so smaller blocks mean fewer
    // fat_js / Q / dirrow gathers (measured on C3, same box, uniform blocks: 884 / 975 / 1 051 /
    // 1 076 M reads/s at 64 / 32 / 16 / 8 runs per block).  Every letter gets its own block size
    // B_c = K * (r_c / r)^-alpha runs (r_c = runs of the letter): a slot fails with probability
    // ~ B_c * r_c / r / 2, so for a given number of slots the failures over all letters are fewest
    // with alpha = 1/2 when every letter is asked for equally often and with alpha = 1 when letters
    // are asked for as often as they head runs; 0.7 sits between (SPX_FAT_ALPHA overrides).  K is the
    // smallest value -- the densest tables -- that keeps the whole flat index within the budget:
    // 75 % of the device's memory (SPX_INDEX_BUDGET_GB overrides) and what is free right now.  (66 % until round 6: a real
    // BWT, whose letters' runs cluster, keeps gaining up to ~16 slots per run -- 36.1 / 37.5 / 38.8 / 39.9 G steps/s at 6.8 /
    // 8.4 / 11 / 16 on the 2.9 * 10^7-run index of profiles/r06_fat_table.txt -- and what a query context needs beside
    // the index is a few GB.)
    size_t mem_free = 0, mem_total = 0;
    SPX_HIP(hipMemGetInfo(&mem_free, &mem_total));
    const uint32_t fat_row_bytes = sizeof(FatRow);
    // SPX_FAT_LROW=1: a PML-only index with compact rows keeps the landing run's row in its fat slots (32-byte slots, half
    // as many of them for the same budget; profiles/r04_fat_lrow.txt)
    const bool want_lrow = getenv("SPX_FAT_LROW") != nullptr && atoi(getenv("SPX_FAT_LROW")) != 0;
    const bool lrow = want_lrow && compact && !(has_ms || docs);
    const uint32_t fat_stride = fat_row_bytes + ((has_ms || docs || lrow) ? (uint32_t)sizeof(Aux) : 0);
    const double per_slot = fat_stride + 4.0 / FJ_GROUP /* fat_js */;
    // per-run arrays: rows 16 / 32 + dirrows 32 + Q 4 (+ aux 16, ss_by_run 8, rundocs 4)
    const double fixed = (double)r * ((double)row_bytes + 32 + 4 + ((has_ms || docs) ? 16 : 0) + (has_ms ? 8 : 0) + (docs ? 4 : 0));
    // (everything else of the index is allocated by now: the table shares what is free with nothing but the caller)
    const double to_come = (double)(64 << 20);
    double budget = 0.75 * (double)mem_total;
    if (const char* e = getenv("SPX_INDEX_BUDGET_GB")) budget = atof(e) * 1e9;
    double fat_bytes = budget - fixed;
    const double fat_bytes_free = 0.92 * (double)mem_free - to_come;
    if (fat_bytes > fat_bytes_free) fat_bytes = fat_bytes_free;
    double max_slots = fat_bytes > 0 ? fat_bytes / per_slot : 0;
    // past ~16 slots per run nothing is left to gain (at 7.6 a slot fails to answer ~4 % of the jumps it is asked,
    // at 16 under 2 %): a small index does not take 75 % of the device for its table
    if (max_slots > 16.0 * (double)r) max_slots = 16.0 * (double)r;
    if (const char* e = getenv("SPX_FAT_SLOTS_PER_RUN")) max_slots = atof(e) * (double)r;  // test / experiment knob
    double alpha = 0.7;
    if (const char* e = getenv("SPX_FAT_ALPHA")) alpha = atof(e);
    int force_bshift = -1;
    if (const char* e = getenv("SPX_FAT_BSHIFT")) {  // test / experiment knob: 2^b runs per block, every letter
        force_bshift = atoi(e) & 31;
        if (force_bshift > 20) force_bshift = 20;
    }
    auto geometry = [&](double K, bool commit) -> double {  // slots in all for block sizes K * share^-alpha
        uint64_t slots = 0;
        for (uint8_t c : lets) {
            LetterInfo& li = hl[c];
            double B = force_bshift >= 0 ? (double)(1u << force_bshift)
                                         : K * pow((double)(li.qend - li.qbeg) / (double)r, -alpha);
            if (B < 1.0) B = 1.0;
            const double m = 4294967296.0 / B;
            const uint32_t bmul = m >= 4294967295.0 ? 0xffffffffu : (m < 1.0 ? 1u : (uint32_t)m);
            if (commit) {
                li.bmul = bmul;
                li.fbase = slots;
            }
            slots += letter_slots((uint32_t)r, bmul);
        }
        return (double)slots;
    };
    double K = 1.0;
    if (force_bshift < 0) {
        double lo = 1e-6, hi = 4e9;  // slots(K) falls as K grows: smallest K that fits
        if (geometry(lo, false) <= max_slots) {
            K = lo;
        } else {
            for (int it = 0; it < 80; ++it) {
                const double mid = sqrt(lo * hi);
                if (geometry(mid, false) <= max_slots)
                    hi = mid;
                else
                    lo = mid;
            }
            K = hi;
        }
    }
    const uint64_t nfat = (uint64_t)geometry(K, true);
    SPX_HIP(hipMemcpyAsync(ix->letters, hl.data(), 256 * sizeof(LetterInfo), hipMemcpyHostToDevice, st));
    bytes += (nfat + 2) * (uint64_t)fat_stride;
    // the fat table: built from the arrays above (also what spx_index_load_flat does instead of reading it)
    ix->view.nletters = nletters;
    ix->view.nfat = nfat;
    ix->view.fat_stride = fat_stride;
    ix->view.r = (uint32_t)r;
    ix->view.compact = (uint32_t)compact;  // (build_fat reads the heads of landing runs out of the rows)
    {
        const int rc_fat = build_fat(ix);
        if (rc_fat != SPX_OK) return rc_fat;
    }
    bytes += fatjs_count(nfat) * 4 + 64;

    SPX_HIP(hipGetLastError());  // a kernel of this function that failed to launch
    unsigned long long herr = 0;
    SPX_HIP(hipMemcpyAsync(&herr, err.p, 8, hipMemcpyDeviceToHost, st));
    SPX_HIP(hipStreamSynchronize(st));
    if (herr) {
        set_error("index arrays violate a structural invariant (%llu violations: zero/oversized run "
                  "length, threshold select past the stored thresholds, threshold > n, or doc id > 65535)",
                  herr);
        return SPX_E_FORMAT;
    }

    // scalars of the initial state (compute_ms_pml.cpp:243, 298, 575, 641-642)
    DevIndex& v = ix->view;
    v.fat_stride = fat_stride;
    ix->n_text = 0;
    bind_view(ix);
    ix->arr_bytes[A_ROWS] = (r + ROW_PAD) * row_bytes;
    ix->arr_bytes[A_DIRROWS] = (r + ROW_PAD) * sizeof(JumpRow);
    ix->arr_bytes[A_FAT] = (nfat + 2) * (uint64_t)fat_stride;
    ix->arr_bytes[A_FATJ] = fatjs_count(nfat) * 4 + 64;
    ix->arr_bytes[A_Q] = (r + 1 + Q_PAD) * 4;
    ix->arr_bytes[A_AUX] = ix->aux ? (r + 2) * sizeof(Aux) : 0;
    ix->arr_bytes[A_SSRUN] = ix->ss_by_run ? (r + 4) * 8 : 0;
    ix->arr_bytes[A_RUNDOCS] = ix->rundocs ? (r + ROW_PAD) * 4 : 0;
    ix->arr_bytes[A_LETTERS] = 256 * sizeof(LetterInfo);
    ix->arr_bytes[A_TEXT] = 0;
    v.n = n;
    v.r = (uint32_t)r;
    v.compact = (uint32_t)compact;
    v.nletters = nletters;
    v.nfat = nfat;
    v.init_k = (uint32_t)(r - 1);
    v.init_off = last_len - 1;
    memset(&v.init_row, 0, sizeof v.init_row);
    SPX_HIP(hipMemcpy(&v.init_row, reinterpret_cast<const char*>(ix->rows) + (r - 1) * row_bytes, row_bytes,
                      hipMemcpyDeviceToHost));
    v.init_sample = ix->has_samples ? (last_esa + 1) % n : 0;
    v.init_doc = (uint32_t)last_de;
    v.doc_at0 = (uint32_t)first_ds;
    ix->device_bytes = bytes;
    return SPX_OK;
}

}  // namespace spx
