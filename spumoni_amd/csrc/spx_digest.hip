// spx_digest.hip -- minimizer digestion of a read batch on the device: the pre-step of
// `spumoni run -m` (perform_minimizer_digestion, src/spumoni.cpp:294-319) and `run -a`
// (perform_dna_minimizer_digestion, :321-342), which the reference runs read by read on the
// host before matching_statistics (src/compute_ms_pml.cpp:919-923).
//
// What is computed (the minimizer streams are bonsai's, dnbaker/bonsai @5273b81a92, absent
// offline -- the assumptions are listed in DESIGN.md section 4.4):
//   * k-mers of the read over ACGT (k <= 4, include/spumoni_main.hpp:316); a character
//     outside ACGT restarts the k-mer and contributes nothing else;
//   * value of a k-mer: -m  its 8-bit cyclic-polynomial hash  XOR_j rotl8(T[c_j], k-1-j)
//                       -a  its 2-bit code (A0 C1 G2 T3, first base most significant),
//                           ordered by  code ^ XOR_MASK;
//   * the stream of k-mers goes through a window of wsz = w-k+1 entries: from the wsz-th
//     k-mer on, every k-mer reports the least value of the last wsz;
//   * the caller's lambda drops a report equal to the previous one (:305, :333) and writes
//     -m the byte `x > 2 ? x : x + 3` (:311) / -a the k letters of the k-mer (:336).
//
// Mapping: one wavefront per read, 64 positions per iteration, byte work only -- HBM-bound
// streaming (read the characters twice, write ~0.2 bytes per character).  The k-mer stream,
// the window minima and the emitted bytes of one iteration are positioned with ballots +
// popcounts; the last wsz-1 stream entries and the last minimum are carried in an LDS ring.
// Two launches: count -> exclusive scan (hipCUB) -> write, so that the digested reads come out
// concatenated with an offsets array, which is what the walk kernel takes.
#include <hipcub/hipcub.hpp>

#include <type_traits>

#include "spx_internal.h"

namespace spx {
namespace {

constexpr uint64_t LEX_XOR_MASK = 0xe37e28c4271b5a2dULL;  // bonsai's XOR_MASK (from Kraken)

struct DigestArgs {
    const uint8_t* seqs;
    const uint64_t* offs;
    uint64_t nreads;
    uint32_t kind;  // SPX_DIGEST_PROMOTED / SPX_DIGEST_DNA
    uint32_t k;
    uint32_t wsz;   // k-mers per window
    uint32_t ring;  // power of two >= wsz + 64
    uint32_t xm;    // LEX_XOR_MASK restricted to the 2k bits of a k-mer
    uint32_t tile;   // lane-per-read kernel: bytes of input staged in LDS at a time
    uint32_t tpack;  // -m: the four character hashes T[A] | T[C] << 8 | T[G] << 16 | T[T] << 24
    uint8_t key_of_kmer[256];  // sort key of every k-mer code: the hash (-m) or code ^ xm (-a)
    const uint32_t* only;      // k_digest_wave: digest only the reads whose flag is set (null: all of them)
    uint64_t* counts;          // pass 0: counts[q + 1] = bytes read q digests to
    const uint64_t* out_offs;  // pass 1
    uint8_t* out;
};

// "ACGT"[c] without a table: the four letters packed in one constant
__device__ __forceinline__ uint32_t letter_of(uint32_t c) { return (0x54474341u >> (8 * c)) & 0xffu; }

__device__ __forceinline__ int base_code(uint32_t c) {
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
}

template <int PASS>
__global__ void __launch_bounds__(64) k_digest_wave(const DigestArgs a) {
    extern __shared__ uint8_t lds[];
    uint8_t* const keys = lds;            // ring of stream keys
    uint8_t* const mins = lds + a.ring;   // ring of window minima, by stream index
    uint8_t* const lut = lds + 2 * a.ring;
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 256; i += 64) lut[i] = a.key_of_kmer[i];
    __syncthreads();
    const uint32_t rmask = a.ring - 1;
    const uint32_t k = a.k, wsz = a.wsz;
    const uint64_t lt = (1ull << lane) - 1;
    for (uint64_t rd = blockIdx.x; rd < a.nreads; rd += gridDim.x) {
        if (a.only != nullptr && a.only[rd] == 0) continue;
        const uint64_t base = a.offs[rd];
        const uint64_t len = a.offs[rd + 1] - base;
        uint64_t ob = 0;
        if (PASS == 1) ob = a.out_offs[rd];
        uint64_t t_base = 0;  // k-mers streamed so far
        uint64_t e_base = 0;  // values emitted so far
        uint64_t prev_valid = 0;
        uint32_t prev_code = 0;
        for (uint64_t p0 = 0; p0 < len; p0 += 64) {
            const uint64_t p = p0 + lane;
            const uint32_t code = p < len ? (uint32_t)base_code(a.seqs[base + p]) : 4u;
            const uint64_t valid = __ballot(code < 4);
            // k-mer ending at p: the k characters p-k+1 .. p must all be ACGT
            uint64_t kv = valid;
            uint32_t kmer = code & 3;
            for (uint32_t j = 1; j < k; ++j) {
                kv &= (valid << j) | (prev_valid >> (64 - j));
                const uint32_t up = __shfl_up(code, j), carry = __shfl(prev_code, (int)(64 + lane - j) & 63);
                kmer |= ((lane >= j ? up : carry) & 3) << (2 * j);
            }
            const bool has = (kv >> lane) & 1;
            const uint64_t t = t_base + __popcll(kv & lt);
            if (has) keys[t & rmask] = lut[kmer];
            __syncthreads();
            const bool reports = has && t + 1 >= wsz;
            uint32_t mn = 0xffffffffu;
            if (reports) {
                for (uint32_t j = 0; j < wsz; ++j) mn = min(mn, (uint32_t)keys[(t - j) & rmask]);
                mins[t & rmask] = (uint8_t)mn;
            }
            __syncthreads();
            // mseq_vec.empty() || mseq_vec.back() != x   (values are < 256: the uint8_t vector
            // compares exactly)
            const bool emit = reports && (t + 1 == wsz || mins[(t - 1) & rmask] != mn);
            const uint64_t em = __ballot(emit);
            if (PASS == 1 && emit) {
                const uint64_t e = e_base + __popcll(em & lt);
                if (a.kind == SPX_DIGEST_PROMOTED) {
                    a.out[ob + e] = (uint8_t)(mn > 2 ? mn : mn + 3);
                } else {
                    const uint32_t code_min = mn ^ a.xm;
                    for (uint32_t j = 0; j < k; ++j)
                        a.out[ob + e * k + j] = (uint8_t)letter_of((code_min >> (2 * (k - 1 - j))) & 3);
                }
            }
            t_base += __popcll(kv);
            e_base += __popcll(em);
            prev_valid = valid;
            prev_code = code;
            __syncthreads();
        }
        if (PASS == 0 && lane == 0) a.counts[rd + 1] = e_base * (a.kind == SPX_DIGEST_DNA ? k : 1);
    }
}


// ---- lane-per-read variant: short reads, windows of at most 8 k-mers ------------------------
// 64 consecutive reads are one contiguous stretch of `seqs`: the wavefront copies it into LDS
// with coalesced 16-byte loads (TILE bytes at a time) and every lane then walks ITS read
// sequentially, exactly like the reference's loop -- k-mer in a register, the window's keys in
// one 64-bit register (newest in the low byte), minimum by packed 16-bit mins.  No ballots, no
// stream compaction: a character outside ACGT just resets the lane's k-mer fill.
constexpr uint32_t TILE_MAX = 32768;  // the tile is a.tile bytes, a multiple of 1024: see launch_digest

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_min_u16(uint32_t x, uint32_t y) {
    us2 a = __builtin_bit_cast(us2, x), b = __builtin_bit_cast(us2, y);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(a, b));
}

// least byte of a 64-bit word
__device__ __forceinline__ uint32_t byte_min8(uint64_t w) {
    const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
    const uint32_t m = pk_min_u16(pk_min_u16(lo & 0x00ff00ffu, (lo >> 8) & 0x00ff00ffu),
                                  pk_min_u16(hi & 0x00ff00ffu, (hi >> 8) & 0x00ff00ffu));
    return min(m & 0xffffu, m >> 16);
}

// The default shape -- k = 4, a window of eight k-mers (w = 11) -- over a read that is all ACGT, FOUR CHARACTERS PER STEP
// with every quantity packed in one register (round 4; the generic loop below spends ~30 instructions per character
// and the kernel was bound by them: 80 % VALU utilisation, profiles/r04_digest_counters.txt):
//   codes   t  = ((x >> 1) ^ (x >> 2)) & 0x03030303          the 2-bit codes of the four characters, one per byte
//   valid      = v_perm(letters, t) == x                      the code spelled back is the character itself
//   keys    -m   T[c_p] ^ rotl(T[c_p-1], 1) ^ rotl(T[c_p-2], 2) ^ rotl(T[c_p-3], 3): four v_perm lookups of the (rotated)
//                character hashes by the codes and the codes of the one / two / three characters before (v_alignbyte)
//           -a   (t | t1 << 2 | t2 << 4 | t3 << 6) ^ xm
//   minima       the keys of even and odd positions in 16-bit halves (E, O); the minimum over the last 2, 4, 8 positions by
//                doubling with v_pk_min_u16, "one position back" being the other register or a 16-bit funnel shift
//   emits        minimum != minimum one position back, as a 4-bit mask; a 16-entry table of v_perm selectors packs the
//                emitted bytes, which collect in a 64-bit register and leave four at a time
// A lane that meets a character outside ACGT (or a shape other than k = 4, wsz = 8) takes the generic loop for its read.
struct SwarState {
    uint32_t tp, Op, AEp, AOp, BEp, BOp, COp;
};

__device__ __forceinline__ uint32_t promote4(uint32_t m) {  // x > 2 ? x : x + 3 on four bytes (src/spumoni.cpp:311)
    const uint32_t ge3 = (m | ((m | 0x80808080u) - 0x03030303u)) & 0x80808080u;
#ifdef SPX_EXP_NOPROMOTE
    return m;
#endif
    const uint32_t lt = (ge3 ^ 0x80808080u) >> 7;  // 1 in every byte below 3
    return m + lt + (lt << 1);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    for (int sft = 32; sft > 0; sft >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, sft));
    return v;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    for (int sft = 32; sft > 0; sft >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, sft));
    return v;
}

template <int KIND, bool FAST8>
__global__ void __launch_bounds__(64) k_digest_lanes(const DigestArgs a) {
    constexpr int PASS = 2;  // count + park the minimizer bytes at the read's input offset (see below)
    // the tile: a.tile bytes (dynamic shared memory: what 64 reads of the batch's mean length need, so that more
    // wavefronts fit a CU -- 12.8 KB and 12 of them for 200 bp reads against 9 with the full 16 KB) + slack for the last
    // 4-byte read of a read
    extern __shared__ uint4 tile16[];
    const uint32_t TILE = a.tile;
    __shared__ uint8_t lut[256];
    __shared__ uint32_t s_sel[16];  // v_perm selectors that pack the bytes named by a 4-bit mask (0x0c: a zero byte)
    const uint8_t* const tile = reinterpret_cast<const uint8_t*>(tile16);
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 256; i += 64) lut[i] = a.key_of_kmer[i];
    if (lane < 16) {
        uint32_t sel = 0x0c0c0c0cu, n = 0;
        for (uint32_t j = 0; j < 4; ++j)
            if ((lane >> j) & 1) {
                sel = (sel & ~(0xffu << (8 * n))) | (j << (8 * n));
                n++;
            }
        s_sel[lane] = sel;
    }
    const uint32_t k = a.k, wsz = a.wsz;
    const uint32_t kmask = (1u << (2 * k)) - 1;
    const uint64_t hm = wsz >= 8 ? 0ull : (~0ull << (8 * wsz));  // window bytes that do not exist
    const uint64_t ngroups = (a.nreads + 63) / 64;
    // -m: the character hashes rotated by 0 .. 3 (bytes A, C, G, T)
    auto rot4 = [](uint32_t v, uint32_t r) {
        return r ? (((v << r) & (0x01010101u * ((0xffu << r) & 0xffu))) | ((v >> (8 - r)) & (0x01010101u * (0xffu >> (8 - r))))) : v;
    };
#ifdef SPX_EXP_TABV
    uint32_t T0 = a.tpack, R1 = rot4(a.tpack, 1), R2 = rot4(a.tpack, 2), R3 = rot4(a.tpack, 3);
    asm volatile("v_mov_b32 %0, %0" : "+v"(T0));
    asm volatile("v_mov_b32 %0, %0" : "+v"(R1));
    asm volatile("v_mov_b32 %0, %0" : "+v"(R2));
    asm volatile("v_mov_b32 %0, %0" : "+v"(R3));
#else
    const uint32_t T0 = a.tpack, R1 = rot4(a.tpack, 1), R2 = rot4(a.tpack, 2), R3 = rot4(a.tpack, 3);
#endif
    const uint32_t xm4 = (a.xm & 0xffu) * 0x01010101u;
    constexpr uint32_t FIRST = 10;  // the first reporting position of a read: k - 1 + wsz - 1
    for (uint64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const uint64_t rd = grp * 64 + lane;
        const bool live = rd < a.nreads;
        const uint64_t rbeg = a.offs[live ? rd : a.nreads];
        const uint64_t rend = a.offs[live ? rd + 1 : a.nreads];
        const uint64_t g0 = __shfl(rbeg, 0), g1 = __shfl(rend, 63);
        // the minimizer bytes are parked at the read's INPUT offset in a scratch buffer (k_digest_unstash moves them
        // once the offsets are known) -- the reads are digested once
        const uint64_t ob = rbeg;
        uint64_t e = 0;
        bool generic = !FAST8;
        auto load_tile = [&](uint64_t t0) {
            __syncthreads();  // everybody is done with the previous tile (and the tables are in place)
            const uint64_t tend = min(t0 + TILE, (g1 + 15) & ~15ull);
            // all loads of the tile in flight before the first LDS write; slots past the
            // end of the stretch re-read its first chunk (always a valid address) and are never used
            // (eight loads in flight at a time: all sixteen cost 64 registers and, beside the packed walk's state, went
            // through scratch memory)
            constexpr uint32_t NB = 8;
            const uint32_t span = (uint32_t)(tend - t0);
            const uint8_t* const src = a.seqs + t0;
            const uint32_t nch = TILE / 1024;  // chunks of 64 lanes x 16 bytes (a.tile is a multiple of 1024)
#pragma unroll 1
            for (uint32_t j0 = 0; j0 < nch; j0 += NB) {
                // (named values, not an array: the array was left in scratch memory)
                auto ld = [&](uint32_t j) {
                    const uint32_t o = ((j0 + j) * 64 + lane) * 16;
                    return *reinterpret_cast<const uint4*>(src + (o < span ? o : 0u));
                };
                const uint4 v0 = ld(0), v1 = ld(1), v2 = ld(2), v3 = ld(3), v4 = ld(4), v5 = ld(5), v6 = ld(6), v7 = ld(7);
                uint4* const dst = tile16 + j0 * 64 + lane;
                const uint32_t left = nch - j0;  // (uniform)
                dst[0 * 64] = v0;
                if (left > 1) dst[1 * 64] = v1;
                if (left > 2) dst[2 * 64] = v2;
                if (left > 3) dst[3 * 64] = v3;
                if (left > 4) dst[4 * 64] = v4;
                if (left > 5) dst[5 * 64] = v5;
                if (left > 6) dst[6 * 64] = v6;
                if (left > 7) dst[7 * 64] = v7;
            }
            __syncthreads();
        };
        if (FAST8) {
            // ---- four characters per step, everything packed (all-ACGT reads) ----
            SwarState q{0, 0, 0, 0, 0, 0, 0};
            uint64_t acc = 0;  // emitted bytes not yet stored, low byte first
            uint32_t nacc = 0;
            uint32_t bad = 0;
            uint64_t cur = rbeg;  // the next character of this lane's read
            const uint32_t len = (uint32_t)min<uint64_t>(rend - rbeg, 0xffffffffull);
            // consecutive tiles overlap by 16 bytes: a lane takes whole steps only (but at its read's end), and the up
            // to three characters it leaves behind at the end of a tile are in the next one
            for (uint64_t t0 = g0 & ~15ull; t0 < g1; t0 += TILE - 16) {
                load_tile(t0);
                const uint64_t hi = min(rend, t0 + TILE);
                const uint32_t avail = hi > cur ? (uint32_t)(hi - cur) : 0;
                const uint32_t take = hi == rend ? avail : (avail & ~3u);
                const uint32_t off = take ? (uint32_t)(cur - t0) : 0;
                const uint32_t pos0 = (uint32_t)(cur - rbeg);
                const uint32_t nst = (take + 3) >> 2;
                // a lane with nothing to do in this tile goes through the common steps like the others, on whatever bytes,
                // with its results masked away (vm) and its state put back afterwards
                const uint32_t vm = nst ? 0xffffffffu : 0u;
                const SwarState q_before = q;
                auto step = [&](uint32_t st, auto masked_tag) {
                    constexpr bool MASKED = decltype(masked_tag)::value;
                    uint32_t x;
                    __builtin_memcpy(&x, tile + off + 4 * st, 4);
                    const uint32_t pos = pos0 + 4 * st;
                    const uint32_t t = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
                    const uint32_t inval = __builtin_amdgcn_perm(0x54474341u, 0x54474341u, t) ^ x;
                    const uint32_t t1 = __builtin_amdgcn_alignbyte(t, q.tp, 3), t2 = __builtin_amdgcn_alignbyte(t, q.tp, 2),
                                   t3 = __builtin_amdgcn_alignbyte(t, q.tp, 1);
                    q.tp = t;
                    uint32_t key;
#ifdef SPX_EXP_DNAKEY
                    if (false)
#else
                    if (KIND == SPX_DIGEST_PROMOTED)
#endif
                        key = __builtin_amdgcn_perm(T0, T0, t) ^ __builtin_amdgcn_perm(R1, R1, t1) ^
                              __builtin_amdgcn_perm(R2, R2, t2) ^ __builtin_amdgcn_perm(R3, R3, t3);
                    else
                        key = (t | (t1 << 2) | (t2 << 4) | (t3 << 6)) ^ xm4;
                    const uint32_t E = key & 0x00ff00ffu, O = (key >> 8) & 0x00ff00ffu;
                    const uint32_t AE = pk_min_u16(E, __builtin_amdgcn_alignbit(O, q.Op, 16)), AO = pk_min_u16(O, E);
                    const uint32_t BE = pk_min_u16(AE, __builtin_amdgcn_alignbit(AE, q.AEp, 16)),
                                   BO = pk_min_u16(AO, __builtin_amdgcn_alignbit(AO, q.AOp, 16));
                    const uint32_t CE = pk_min_u16(BE, q.BEp), CO = pk_min_u16(BO, q.BOp);
                    const uint32_t XE = CE ^ __builtin_amdgcn_alignbit(CO, q.COp, 16), XO = CO ^ CE;
                    q.Op = O;
                    q.AEp = AE;
                    q.AOp = AO;
                    q.BEp = BE;
                    q.BOp = BO;
                    q.COp = CO;
                    const uint32_t FE = ((XE + 0x00ff00ffu) >> 8) & 0x00010001u, FO = ((XO + 0x00ff00ffu) >> 8) & 0x00010001u;
                    const uint32_t f = FE | (FO << 1);
                    uint32_t idx = (f | (f >> 14)) & 0xfu;
                    // positions that report: FIRST .. len - 1; the first report is always kept (:300 / :329)
                    if (MASKED) {  // (the first three steps of a read and its last one)
                        const int32_t lo = (int32_t)FIRST - (int32_t)pos, hiq = (int32_t)len - (int32_t)pos;
                        const uint32_t l4 = lo < 0 ? 0u : (lo > 4 ? 4u : (uint32_t)lo), h4 = hiq < 0 ? 0u : (hiq > 4 ? 4u : (uint32_t)hiq);
                        const uint32_t pm = h4 > l4 ? ((1u << h4) - (1u << l4)) : 0u;
                        idx = (idx & pm) | ((lo >= 0 && lo < 4 && (uint32_t)lo < h4) ? (1u << lo) : 0u);
                        bad |= (h4 >= 4 ? inval : (inval & ((1u << (8 * h4)) - 1u))) & vm;
                    } else {
                        bad |= inval & vm;
                    }
                    idx &= vm;
                    const uint32_t vals = CE | (CO << 8);
                    const uint32_t sel = s_sel[idx];
                    const uint32_t comp = __builtin_amdgcn_perm(vals, vals, sel);
                    acc |= (uint64_t)comp << (8 * nacc);
                    nacc += (uint32_t)__popc(idx);
                    if (nacc >= 4) {
                        const uint32_t w4 = KIND == SPX_DIGEST_PROMOTED ? promote4((uint32_t)acc) : (uint32_t)acc;
#ifdef SPX_EXP_DIGEST_NOSTORE
                        if (w4 == 0x12345678u)
#endif
                        __builtin_memcpy(a.out + ob + e, &w4, 4);
                        acc >>= 32;
                        nacc -= 4;
                        e += 4;
                    }
                };
                // the steps every lane takes run without a predicate (reads of one length: all of them), and those of them at
                // which every lane is past its read's first reports and before its end without the position masks ...
                uint32_t nst_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(nst ? nst : 0xffffffffu));
                if (nst_all == 0xffffffffu) nst_all = 0;
                const uint32_t f_l = pos0 >= FIRST + 2 ? 0u : (FIRST + 2 - pos0 + 3) >> 2;  // first step with pos >= FIRST + 2
                const uint32_t e_l = len > pos0 ? (len - pos0) >> 2 : 0u;                    // steps with pos + 4 <= len
                uint32_t st_f = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(vm ? f_l : 0u));
                uint32_t st_e = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(vm ? e_l : 0xffffffffu));
                st_e = st_e < nst_all ? st_e : nst_all;
                st_f = st_f < st_e ? st_f : st_e;
                using Masked = std::integral_constant<bool, true>;
                using Plain = std::integral_constant<bool, false>;
                uint32_t st = 0;
                for (; st < st_f; ++st) step(st, Masked{});
                for (; st + 2 <= st_e; st += 2) {
                    step(st, Plain{});
                    step(st + 1, Plain{});
                }
                for (; st < st_e; ++st) step(st, Plain{});
                for (; st < nst_all; ++st) step(st, Masked{});
                if (!vm) q = q_before;
                // ... the rest lane by lane
                for (st = nst_all; __builtin_amdgcn_ballot_w64(st < nst) != 0; ++st)
                    if (st < nst) step(st, Masked{});
                cur += take;
            }
            const uint32_t wt = KIND == SPX_DIGEST_PROMOTED ? promote4((uint32_t)acc) : (uint32_t)acc;
            if (!bad)
                for (uint32_t j = 0; j < nacc; ++j) a.out[ob + e + j] = (uint8_t)(wt >> (8 * j));
            e += nacc;
            generic = bad != 0;
        }
        if (__builtin_amdgcn_ballot_w64(generic) != 0) {
            // ---- the generic loop: any k <= 4, any window of at most eight k-mers, any characters ----
            // the lane's walk state
            uint32_t filled = 0, kmer = 0, cnt = 0, last = 0, acc = 0, nacc = 0;
            bool have = false;
            uint64_t win = 0;
            if (generic) e = 0;
            auto push_byte = [&](uint32_t v) {  // byte e of this read's digest
                const uint64_t addr = ob + e;
                acc |= v << (8 * (uint32_t)(addr & 3));
                nacc++;
                e++;
                if (((addr + 1) & 3) == 0) {  // the aligned dword that holds addr is complete
                    if (nacc == 4) {
                        *reinterpret_cast<uint32_t*>(a.out + (addr - 3)) = acc;
                    } else {  // its first bytes belong to the previous read
                        for (uint32_t j = 4 - nacc; j < 4; ++j) a.out[addr - 3 + j] = (uint8_t)(acc >> (8 * j));
                    }
                    acc = 0;
                    nacc = 0;
                }
            };
            for (uint64_t t0 = g0 & ~15ull; t0 < g1; t0 += TILE) {
                load_tile(t0);
                const uint64_t lo = max(rbeg, t0), hi = min(rend, t0 + TILE);
                const uint32_t mine = (generic && hi > lo) ? (uint32_t)(hi - lo) : 0;
                const uint32_t off = (uint32_t)(lo - t0);  // only used when mine > 0
                for (uint32_t s = 0; __builtin_amdgcn_ballot_w64(s < mine) != 0; s += 4) {
                    // four characters of this lane's read (bytes past `mine` are ignored below; the
                    // tile array is over-allocated so that the read stays inside LDS)
                    uint32_t w4 = 0;
                    if (s < mine) __builtin_memcpy(&w4, tile + off + s, 4);
                    uint32_t kmers[4], keys[4];
                    bool act[4], has[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t ch = (w4 >> (8 * j)) & 0xffu;
                        act[j] = s + j < mine;
                        const uint32_t d = ch - 'A';
                        const bool valid = d < 20 && ((0x80045u >> d) & 1);  // A C G T
                        const uint32_t nf = valid ? min(filled + 1, k) : 0;
                        filled = act[j] ? nf : filled;
                        const uint32_t nk = ((kmer << 2) | (((ch >> 1) ^ (ch >> 2)) & 3)) & kmask;  // A0 C1 G2 T3
                        kmer = act[j] ? nk : kmer;
                        kmers[j] = kmer;
                        has[j] = act[j] && filled == k;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        keys[j] = KIND == SPX_DIGEST_PROMOTED ? (uint32_t)lut[kmers[j]] : (kmers[j] ^ a.xm);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        win = has[j] ? ((win << 8) | keys[j]) : win;
                        cnt = has[j] ? min(cnt + 1, wsz) : cnt;
                        const uint32_t mn = byte_min8(win | hm);
                        const bool em = has[j] && cnt == wsz && (!have || mn != last);
                        last = em ? mn : last;
                        have = have || em;
                        if (em) {
                            if (KIND == SPX_DIGEST_PROMOTED)
                                push_byte(mn > 2 ? mn : mn + 3);
                            else
                                push_byte(mn);  // one byte per minimizer; spelled out by k_digest_unstash
                        }
                    }
                }
            }
            if (generic) {
                for (uint32_t j = 0; j < nacc; ++j) {
                    const uint64_t addr = ob + e - nacc + j;
                    a.out[addr] = (uint8_t)(acc >> (8 * (uint32_t)(addr & 3)));
                }
            }
        }
        (void)PASS;
        if (live) a.counts[rd + 1] = e * (KIND == SPX_DIGEST_DNA ? k : 1);
    }
}

// ---- long reads: lane per CHUNK --------------------------------------------------------------------------------------
// A read of thousands of characters keeps one lane of the kernel above busy and sixty-three waiting, and the
// wavefront-per-read kernel pays ballots and LDS rings for every 64 characters (120 G characters/s against 1 300 for short
// reads).  But the packed walk depends on its past for eleven characters only: a k-mer on the three before it, a window on the
// seven k-mers before that, an emit on the minimum one position back.  So an all-ACGT read is cut into chunks of CHUNK
// characters, a LANE takes a chunk, starts HALO = 12 characters early with its emits masked, and produces exactly what the
// sequential loop produces for the chunk's positions.  The chunks of a batch are consecutive in the input (a read's chunks,
// then the next read's), so 64 of them are one stretch of at most 64 * 240 + 12 bytes: one tile.  Every chunk's bytes are
// parked at the chunk's input offset and counted; ONE scan over the chunks gives every chunk's place in the concatenated
// output -- a read's offset is its first chunk's -- and k_digest_unstash moves the pieces with the chunks in the role of the
// reads.  A read with a character outside ACGT is flagged and redone by the wavefront-per-read kernel afterwards.
constexpr uint32_t DCHUNK = 240, DHALO = 12;

__global__ void k_dchunk_count(const uint64_t* offs, uint64_t nreads, uint64_t* cnt, uint32_t* bad) {
    const uint64_t q = blockIdx.x * 256ull + threadIdx.x;
    if (q > nreads) return;
    cnt[q] = q < nreads ? (offs[q + 1] - offs[q] + DCHUNK - 1) / DCHUNK : 0;  // (exclusive scan: cnt[nreads] = chunks in all)
    if (q < nreads) bad[q] = 0;
}

// chunk g: input offset c_start[g], read c_rd[g]; entries from the batch's last chunk up to `bound` are empty chunks at the
// input's end (the launches are sized by the bound: nothing returns to the host)
__global__ void k_dchunk_fill(const uint64_t* offs, uint64_t nreads, const uint64_t* first_chunk, uint64_t bound, uint64_t* c_start,
                              uint32_t* c_rd) {
    const uint64_t q = blockIdx.x * 256ull + threadIdx.x;
    if (q < nreads) {
        const uint64_t b = offs[q], e = offs[q + 1];
        uint64_t g = first_chunk[q];
        for (uint64_t at = b; at < e; at += DCHUNK, ++g) {
            c_start[g] = at;
            c_rd[g] = (uint32_t)q;
        }
    }
    // (the entries between the count and the bound: every thread takes its share)
    for (uint64_t g = first_chunk[nreads] + q; g <= bound; g += (uint64_t)gridDim.x * 256) {
        c_start[g] = offs[nreads];
        c_rd[g] = 0xffffffffu;
    }
}

template <int KIND>
__global__ void __launch_bounds__(64) k_digest_chunks(const DigestArgs a, const uint64_t* c_start, const uint32_t* c_rd, uint64_t bound,
                                                      uint64_t* c_count, uint32_t* bad) {
    // (+ slack: a lane whose chunk is shorter than its neighbours' goes through their steps on whatever follows its own bytes)
    __shared__ uint4 tile16[(64 * DCHUNK + 64 + 4 * DCHUNK) / 16 + 2];
    __shared__ uint32_t s_sel[16];
    const uint8_t* const tile = reinterpret_cast<const uint8_t*>(tile16);
    const uint32_t lane = threadIdx.x;
    if (lane < 16) {
        uint32_t sel = 0x0c0c0c0cu, n = 0;
        for (uint32_t j = 0; j < 4; ++j)
            if ((lane >> j) & 1) {
                sel = (sel & ~(0xffu << (8 * n))) | (j << (8 * n));
                n++;
            }
        s_sel[lane] = sel;
    }
    auto rot4 = [](uint32_t v, uint32_t r) {
        return r ? (((v << r) & (0x01010101u * ((0xffu << r) & 0xffu))) | ((v >> (8 - r)) & (0x01010101u * (0xffu >> (8 - r))))) : v;
    };
    const uint32_t T0 = a.tpack, R1 = rot4(a.tpack, 1), R2 = rot4(a.tpack, 2), R3 = rot4(a.tpack, 3);
    const uint32_t xm4 = (a.xm & 0xffu) * 0x01010101u;
    constexpr uint32_t FIRST = 10;
    const uint64_t ngroups = (bound + 63) / 64;
    for (uint64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const uint64_t g = grp * 64 + lane;
        const bool in = g < bound;
        const uint32_t rd = in ? c_rd[g] : 0xffffffffu;
        const bool live = rd != 0xffffffffu;
        const uint64_t cbeg = c_start[in ? g : bound];
        const uint64_t rbeg = live ? a.offs[rd] : cbeg, rend = live ? a.offs[rd + 1] : cbeg;
        const uint32_t pos0 = (uint32_t)(cbeg - rbeg);                                     // the chunk's first position in its read
        const uint32_t len = live ? (uint32_t)min<uint64_t>(rend - cbeg, DCHUNK) : 0u;     // its positions
        const uint32_t halo = pos0 ? DHALO : 0u;
        // the tile: from the first chunk's halo to the last chunk's end (dead lanes sit at the input's end)
        const uint64_t lo_all = __shfl(cbeg - halo, 0), hi_all = __shfl(cbeg + len, 63);
        const uint64_t t0 = lo_all & ~15ull, tend = (hi_all + 15) & ~15ull;
        __syncthreads();
        for (uint64_t o = (uint64_t)lane * 16; t0 + o < tend; o += 64 * 16) tile16[o >> 4] = *reinterpret_cast<const uint4*>(a.seqs + t0 + o);
        __syncthreads();
        const uint32_t off = (uint32_t)(cbeg - halo - t0);
        const uint32_t nst = (halo + len + 3) >> 2;
        const uint32_t end = pos0 + len;
        const uint32_t first_ok = pos0 > FIRST ? pos0 : FIRST;  // the chunk's first position that may emit
        const uint32_t full = (halo + len) >> 2;  // steps with four positions of the chunk (or its halo)
        SwarState q{0, 0, 0, 0, 0, 0, 0};
        uint64_t acc = 0, e = 0;
        uint32_t nacc = 0, wrong = 0;
        auto step = [&](uint32_t st, auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            uint32_t x;
            __builtin_memcpy(&x, tile + off + 4 * st, 4);
            const uint32_t t = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
            const uint32_t inval = __builtin_amdgcn_perm(0x54474341u, 0x54474341u, t) ^ x;
            const uint32_t t1 = __builtin_amdgcn_alignbyte(t, q.tp, 3), t2 = __builtin_amdgcn_alignbyte(t, q.tp, 2),
                           t3 = __builtin_amdgcn_alignbyte(t, q.tp, 1);
            q.tp = t;
            uint32_t key;
            if (KIND == SPX_DIGEST_PROMOTED)
                key = __builtin_amdgcn_perm(T0, T0, t) ^ __builtin_amdgcn_perm(R1, R1, t1) ^ __builtin_amdgcn_perm(R2, R2, t2) ^
                      __builtin_amdgcn_perm(R3, R3, t3);
            else
                key = (t | (t1 << 2) | (t2 << 4) | (t3 << 6)) ^ xm4;
            const uint32_t E = key & 0x00ff00ffu, O = (key >> 8) & 0x00ff00ffu;
            const uint32_t AE = pk_min_u16(E, __builtin_amdgcn_alignbit(O, q.Op, 16)), AO = pk_min_u16(O, E);
            const uint32_t BE = pk_min_u16(AE, __builtin_amdgcn_alignbit(AE, q.AEp, 16)),
                           BO = pk_min_u16(AO, __builtin_amdgcn_alignbit(AO, q.AOp, 16));
            const uint32_t CE = pk_min_u16(BE, q.BEp), CO = pk_min_u16(BO, q.BOp);
            const uint32_t XE = CE ^ __builtin_amdgcn_alignbit(CO, q.COp, 16), XO = CO ^ CE;
            q.Op = O;
            q.AEp = AE;
            q.AOp = AO;
            q.BEp = BE;
            q.BOp = BO;
            q.COp = CO;
            const uint32_t FE = ((XE + 0x00ff00ffu) >> 8) & 0x00010001u, FO = ((XO + 0x00ff00ffu) >> 8) & 0x00010001u;
            const uint32_t f = FE | (FO << 1);
            uint32_t idx = (f | (f >> 14)) & 0xfu;
            if (MASKED) {  // the halo (or a read's first ten positions) and the chunk's last step
                const int32_t pos = (int32_t)(pos0 - halo + 4 * st);
                const int32_t lo = (int32_t)first_ok - pos, hiq = (int32_t)end - pos, own = (int32_t)pos0 - pos;
                const uint32_t l4 = lo < 0 ? 0u : (lo > 4 ? 4u : (uint32_t)lo), h4 = hiq < 0 ? 0u : (hiq > 4 ? 4u : (uint32_t)hiq);
                const uint32_t o4 = own < 0 ? 0u : (own > 4 ? 4u : (uint32_t)own);  // bytes of the step that are the chunk before's
                const uint32_t pm = h4 > l4 ? ((1u << h4) - (1u << l4)) : 0u;
                const int32_t fo = (int32_t)FIRST - pos;  // a read's first report is always kept (:300 / :329)
                idx = (idx & pm) | ((pos0 == 0 && fo >= 0 && fo < 4 && (uint32_t)fo < h4) ? (1u << fo) : 0u);
                const uint32_t keep = (h4 >= 4 ? 0xffffffffu : ((1u << (8 * h4)) - 1u)) & (o4 >= 4 ? 0u : ~((1u << (8 * o4)) - 1u));
                wrong |= inval & keep;
            } else {  // (a lane that is past its chunk's end goes along with nothing kept)
                const uint32_t lm = st < full ? 0xffffffffu : 0u;
                wrong |= inval & lm;
                idx &= lm;
            }
            const uint32_t vals = CE | (CO << 8);
            const uint32_t comp = __builtin_amdgcn_perm(vals, vals, s_sel[idx]);
            acc |= (uint64_t)comp << (8 * nacc);
            nacc += (uint32_t)__popc(idx);
            if (nacc >= 4) {
                const uint32_t w4 = KIND == SPX_DIGEST_PROMOTED ? promote4((uint32_t)acc) : (uint32_t)acc;
                __builtin_memcpy(a.out + cbeg + e, &w4, 4);
                acc >>= 32;
                nacc -= 4;
                e += 4;
            }
        };
        using Masked = std::integral_constant<bool, true>;
        using Plain = std::integral_constant<bool, false>;
        // Steps 0 .. 2 are the halo (or a read's first positions): masked.  From there on every lane takes every step of the
        // wavefront's longest chunk, plain ones -- but the step at which some lane's chunk ends inside the four characters, which
        // is a masked step for the whole wavefront (a chunk that has ended keeps nothing in either kind of step).
        const uint32_t nst_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(nst));
        uint32_t st = 0;
        for (; st < 3 && st < nst_max; ++st) step(st, Masked{});
        for (; st < nst_max; ++st) {
            if (__builtin_amdgcn_ballot_w64(st >= full && st < nst) != 0)
                step(st, Masked{});
            else
                step(st, Plain{});
        }
        const uint32_t wt = KIND == SPX_DIGEST_PROMOTED ? promote4((uint32_t)acc) : (uint32_t)acc;
        for (uint32_t j = 0; j < nacc; ++j) a.out[cbeg + e + j] = (uint8_t)(wt >> (8 * j));
        e += nacc;
        if (in) c_count[g + 1] = live ? e * (KIND == SPX_DIGEST_DNA ? a.k : 1) : 0;
        if (live && wrong) atomicOr(&bad[rd], 1u);
    }
}

// a flagged read (a character outside ACGT) is the wavefront-per-read kernel's: its bytes count in its first chunk
__global__ void k_dchunk_fix_counts(const uint32_t* c_rd, const uint64_t* first_chunk, const uint32_t* bad, const uint64_t* read_count,
                                    uint64_t bound, uint64_t* c_count) {
    const uint64_t g = blockIdx.x * 256ull + threadIdx.x;
    if (g >= bound) return;
    const uint32_t rd = c_rd[g];
    if (rd == 0xffffffffu || !bad[rd]) return;
    c_count[g + 1] = first_chunk[rd] == g ? read_count[rd + 1] : 0;
}

// read q's offset in the concatenated output is its first chunk's
__global__ void k_dchunk_read_offsets(const uint64_t* first_chunk, const uint64_t* c_out, uint64_t nreads, uint64_t* out_offs) {
    const uint64_t q = blockIdx.x * 256ull + threadIdx.x;
    if (q <= nreads) out_offs[q] = c_out[first_chunk[q]];
}

// Second half of the one-pass digestion: the minimizer bytes of read q wait at stash[offs[q] ..];
// 64 consecutive reads go to one contiguous stretch of the output, so the wavefront assembles that
// stretch in LDS (every lane fetches its read's bytes with 16-byte loads, -a spells the k-mers
// out) and writes it with aligned 16-byte stores.
constexpr uint32_t STAGE = 8192;

template <int KIND>
__global__ void __launch_bounds__(64) k_digest_unstash(const DigestArgs a, const uint8_t* stash) {
    __shared__ uint4 stage16[STAGE / 16 + 2];
    uint8_t* const stage = reinterpret_cast<uint8_t*>(stage16);
    const uint32_t lane = threadIdx.x;
    const uint32_t k = a.k;
    const uint32_t mult = KIND == SPX_DIGEST_DNA ? k : 1;
    const uint64_t ngroups = (a.nreads + 63) / 64;
    for (uint64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const uint64_t rd = grp * 64 + lane;
        const bool live = rd < a.nreads;
        const uint64_t src = a.offs[live ? rd : a.nreads];
        const uint64_t dbeg = a.out_offs[live ? rd : a.nreads];
        const uint64_t dend = a.out_offs[live ? rd + 1 : a.nreads];
        const uint32_t nmin = (uint32_t)((dend - dbeg) / mult);  // minimizers of this read
        const uint64_t d0 = __shfl(dbeg, 0), d1 = __shfl(dend, 63);
        const uint64_t total = d1 - d0;
        const uint32_t skew = (uint32_t)(d0 & 15);  // stretch index = skew + (address - d0): chunks align
        // the stretch goes through LDS in windows of STAGE bytes (one window for 200 bp reads)
        const uint64_t base = skew + (dbeg - d0);  // stretch index of this read's first byte
        for (uint64_t w0 = 0; w0 < skew + total; w0 += STAGE) {
            __syncthreads();
            const uint64_t w1 = w0 + STAGE;
            // minimizers of this read with a byte in the window
            const uint64_t lo_b = base > w0 ? 0 : w0 - base, hi_b = w1 > base ? w1 - base : 0;
            const uint32_t t_lo = (uint32_t)min<uint64_t>(lo_b / mult, nmin);
            const uint32_t t_hi = (uint32_t)min<uint64_t>((hi_b + mult - 1) / mult, nmin);
            if (t_lo < t_hi) {
                const uint64_t s0 = src + t_lo, s1 = src + t_hi;
                for (uint64_t c = s0 & ~15ull; c < s1; c += 16) {  // aligned 16-byte chunks of the stash
                    const uint4 v = *reinterpret_cast<const uint4*>(stash + c);
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const uint64_t g = c + j;
                        if (g >= s0 && g < s1) {
                            const uint32_t byte = (w[j >> 2] >> (8 * (j & 3))) & 0xffu;
                            const uint64_t r = base + (g - src) * mult;  // stretch index of its first byte
                            if (KIND == SPX_DIGEST_PROMOTED) {
                                stage[r - w0] = (uint8_t)byte;
                            } else {
                                const uint32_t code = byte ^ a.xm;
                                if (k == 4 && r >= w0 && r + 4 <= w1) {  // the usual case: four letters, one write
                                    const uint32_t four = letter_of((code >> 6) & 3) |
                                                          (letter_of((code >> 4) & 3) << 8) |
                                                          (letter_of((code >> 2) & 3) << 16) |
                                                          (letter_of(code & 3) << 24);
                                    __builtin_memcpy(stage + (r - w0), &four, 4);
                                } else {
                                    for (uint32_t i = 0; i < k; ++i)
                                        if (r + i >= w0 && r + i < w1)
                                            stage[r + i - w0] = (uint8_t)letter_of((code >> (2 * (k - 1 - i))) & 3);
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
            // write: aligned 16-byte chunks, the partial ones at the two ends of the stretch byte by byte
            uint8_t* const obase = a.out + (d0 - skew) + w0;
            const uint64_t end = min<uint64_t>(skew + total, w1) - w0;  // bytes of this window
            const uint64_t first = w0 == 0 ? skew : 0;                  // bytes before it that are not ours
            for (uint64_t c = (uint64_t)lane * 16; c < end; c += 64 * 16) {
                if (c >= first && c + 16 <= end) {
                    *reinterpret_cast<uint4*>(obase + c) = stage16[c >> 4];
                } else {
                    for (uint32_t j = 0; j < 16; ++j)
                        if (c + j >= first && c + j < end) obase[c + j] = stage[c + j];
                }
            }
        }
    }
}

// the walk reads its characters in aligned 32-byte windows past the last read: define the tail
__global__ void k_zero_tail(const uint64_t* out_offs, uint64_t nreads, uint8_t* out) {
    const uint64_t total = out_offs[nreads];
    const uint64_t end = ((total + 3) / 4) * 4 + 32;
    for (uint64_t i = total + threadIdx.x; i < end; i += blockDim.x) out[i] = 0;
}

}  // namespace

// The digestion's scratch: grow-only device buffers owned by the index (spx_index::digest_scr), like the chunked walk's.
// (Round 4 took them from the stream-ordered allocator, which leaked them on every early return -- ADVICE r4 -- and needed a
// release threshold on the device's default pool, a process-wide side effect; a pool of the handle's own was tried first and
// hipMemPoolDestroy hung at exit once page-locked file mappings were in play: profiles/r05_cli_e2e_m_hang.txt.  Plain buffers
// have neither problem.)  Calls on one handle are enqueued under its mutex; a call on another stream than the one before
// waits for it (ev_dig), so the buffers are never shared by two calls in flight.
namespace {
struct DigestScratch {
    spx_index* ix;
    int n = 0;
    explicit DigestScratch(spx_index* i) : ix(i) {}
    hipError_t get(void** out, size_t bytes) {
        if (n >= spx_index::NDIGSCR) return hipErrorOutOfMemory;
        spx_index::Scratch& sc = ix->digest_scr[n++];
        if (sc.cap < bytes) {
            if (sc.p) (void)hipFree(sc.p);
            sc.p = nullptr;
            sc.cap = 0;
            const size_t want = bytes + bytes / 4 + 256;
            const hipError_t e = hipMalloc(&sc.p, want);
            if (e != hipSuccess) return e;
            sc.cap = want;
        }
        *out = sc.p;
        return hipSuccess;
    }
};
}  // namespace

int launch_digest(spx_index* ix, int kind, uint32_t k, uint32_t w, const uint8_t* d_seqs, const uint64_t* d_offs,
                  uint64_t nreads, uint64_t total_chars, uint8_t* d_out, uint64_t* d_out_offs, hipStream_t st, bool* parked) {
    // *parked (in): the caller is the walk itself and can take the digested reads where the digestion parks them -- read
    // q's bytes at d_out[d_offs[q] ..], d_out_offs its offsets in the concatenation that is then never made (no second
    // pass over the bytes); (out): whether that is what happened (-m, lane-per-read kernel)
    const bool want_parked = parked && *parked;
    if (parked) *parked = false;
    if (kind != SPX_DIGEST_PROMOTED && kind != SPX_DIGEST_DNA) {
        set_error("digest kind must be SPX_DIGEST_PROMOTED (-m) or SPX_DIGEST_DNA (-a)");
        return SPX_E_ARG;
    }
    if (k < 1 || k > 4) {  // include/spumoni_main.hpp:316
        set_error("small window size (k) must be in [1, 4]");
        return SPX_E_ARG;
    }
    if (w < k) {  // :317
        set_error("large window size (w) should be at least the small window size (k)");
        return SPX_E_ARG;
    }
    DigestArgs a;
    a.only = nullptr;
    a.seqs = d_seqs;
    a.offs = d_offs;
    a.nreads = nreads;
    a.kind = (uint32_t)kind;
    a.k = k;
    a.wsz = w - k + 1;
    if (a.wsz > 16384u - 64) {
        set_error("large window size (w) beyond %u is not supported by the device digestion", 16384u - 64 + k - 1);
        return SPX_E_UNSUPPORTED;
    }
    uint32_t ring = 128;
    while (ring < a.wsz + 64) ring <<= 1;
    a.ring = ring;
    a.xm = (uint32_t)(LEX_XOR_MASK & ((1ull << (2 * k)) - 1));
    for (uint32_t code = 0; code < 256; ++code) {
        if (kind == SPX_DIGEST_DNA) {
            a.key_of_kmer[code] = (uint8_t)((code ^ a.xm) & 0xff);
        } else {
            // CyclicHash<uint8_t>, word size 8, over the k characters, first character first
            uint8_t h = 0;
            for (uint32_t j = 0; j < k; ++j) {
                const uint32_t c = (code >> (2 * (k - 1 - j))) & 3;
                h = (uint8_t)(((h << 1) | (h >> 7)) ^ ix->charhash[c]);
            }
            a.key_of_kmer[code] = h;
        }
    }
    a.tpack = (uint32_t)ix->charhash[0] | ((uint32_t)ix->charhash[1] << 8) | ((uint32_t)ix->charhash[2] << 16) |
              ((uint32_t)ix->charhash[3] << 24);
    a.counts = d_out_offs;
    a.out_offs = d_out_offs;
    a.out = d_out;
    DigestScratch scr(ix);
    if (ix->dig_used && ix->dig_stream != st) SPX_HIP(hipStreamWaitEvent(st, ix->ev_dig, 0));
    struct Done {  // whatever way out: the next call on another stream waits for what this one enqueued
        spx_index* ix;
        hipStream_t st;
        ~Done() {
            if (hipEventRecord(ix->ev_dig, st) == hipSuccess) {
                ix->dig_used = true;
                ix->dig_stream = st;
            }
        }
    } done{ix, st};
    SPX_HIP(hipMemsetAsync(d_out_offs, 0, 8, st));
    if (nreads == 0) {
        k_zero_tail<<<1, 64, 0, st>>>(d_out_offs, 0, d_out);
        return SPX_OK;
    }
    // short reads and a window that fits a register: one lane per read; otherwise one
    // wavefront per read (any length, any window)
    const uint64_t mean_len = total_chars / nreads;
    bool lanes = a.wsz <= 8 && mean_len <= 2048;  // (k = 4, w = 11 above 640 characters: by chunks, below)
    if (ix->force_digest_kernel == 1) lanes = a.wsz <= 8;
    if (ix->force_digest_kernel == 2) lanes = false;
    const uint64_t cus = (uint64_t)(ix->num_cus > 0 ? ix->num_cus : 256);
    // (the scans' workspace: one buffer of its own, the last slot -- the scans of a call run one after the other on the stream)
    auto scan_tmp = [&](void** out, size_t bytes) -> hipError_t {
        spx_index::Scratch& sc = ix->digest_scr[spx_index::NDIGSCR - 1];
        if (sc.cap < bytes) {
            if (sc.p) (void)hipFree(sc.p);
            sc.p = nullptr;
            sc.cap = 0;
            const hipError_t e = hipMalloc(&sc.p, bytes + 4096);
            if (e != hipSuccess) return e;
            sc.cap = bytes + 4096;
        }
        *out = sc.p;
        return hipSuccess;
    };
    auto scan_counts = [&]() -> int {  // counts -> offsets, in place
        size_t tmp_bytes = 0;
        SPX_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, d_out_offs, d_out_offs, nreads + 1, st));
        void* tmp = nullptr;
        SPX_HIP(scan_tmp(&tmp, tmp_bytes));
        SPX_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, d_out_offs, d_out_offs, nreads + 1, st));
        return SPX_OK;
    };
    // long reads of the default shape: a lane per chunk of 240 characters (k_digest_chunks)
    // (tools/digest_bench.py --kernel 1 / 3: the lane-per-read kernel wins up to ~500 characters -- 3.2 against 3.4 ms per
    // 1.8e9 characters --, at 1 000 the chunks are 2 x faster, at 2 000 4.5 x: a group of 64 reads no longer fits a tile)
    bool by_chunks = a.wsz == 8 && k == 4 && mean_len > 640;
    if (ix->force_digest_kernel == 3) by_chunks = a.wsz == 8 && k == 4;
    if (ix->force_digest_kernel == 1 || ix->force_digest_kernel == 2) by_chunks = false;
    if (by_chunks) {
        const uint64_t bound = total_chars / DCHUNK + nreads + 1;  // chunks: at most this many (the count stays on the device)
        uint64_t *first_chunk = nullptr, *c_start = nullptr, *c_count = nullptr, *rcount = nullptr;
        uint32_t *c_rd = nullptr, *bad = nullptr;
        uint8_t* stash = nullptr;
        SPX_HIP(scr.get((void**)&first_chunk, (nreads + 2) * 8));
        SPX_HIP(scr.get((void**)&rcount, (nreads + 2) * 8));
        SPX_HIP(scr.get((void**)&bad, (nreads + 1) * 4));
        SPX_HIP(scr.get((void**)&c_start, (bound + 2) * 8));
        SPX_HIP(scr.get((void**)&c_count, (bound + 2) * 8));
        SPX_HIP(scr.get((void**)&c_rd, (bound + 2) * 4));
        SPX_HIP(scr.get((void**)&stash, total_chars + 64));
        auto scan = [&](uint64_t* v, uint64_t count, bool inclusive) -> int {
            size_t tmp_bytes = 0;
            void* tmp = nullptr;
            if (inclusive)
                SPX_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, v, v, count, st));
            else
                SPX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, v, v, count, st));
            SPX_HIP(scan_tmp(&tmp, tmp_bytes));
            if (inclusive)
                SPX_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, v, v, count, st));
            else
                SPX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, v, v, count, st));
            return SPX_OK;
        };
        const unsigned gq = (unsigned)((nreads + 1 + 255) / 256), gc = (unsigned)((bound + 255) / 256);
        k_dchunk_count<<<gq, 256, 0, st>>>(d_offs, nreads, first_chunk, bad);
        int rc = scan(first_chunk, nreads + 1, false);
        if (rc != SPX_OK) return rc;
        k_dchunk_fill<<<gq, 256, 0, st>>>(d_offs, nreads, first_chunk, bound, c_start, c_rd);
        SPX_HIP(hipMemsetAsync(c_count, 0, 8, st));
        a.out = stash;
        const uint64_t groups = (bound + 63) / 64;
        const uint32_t gridc = (uint32_t)(groups < cus * 9 ? groups : cus * 9);
        if (kind == SPX_DIGEST_PROMOTED)
            k_digest_chunks<SPX_DIGEST_PROMOTED><<<gridc, 64, 0, st>>>(a, c_start, c_rd, bound, c_count, bad);
        else
            k_digest_chunks<SPX_DIGEST_DNA><<<gridc, 64, 0, st>>>(a, c_start, c_rd, bound, c_count, bad);
        SPX_HIP(hipGetLastError());
        // the reads with a character outside ACGT: counted by the wavefront-per-read kernel (it returns at once for the others)
        const size_t lds = 2 * (size_t)ring + 256;
        const uint32_t gridw = (uint32_t)(nreads < cus * 64 ? nreads : cus * 64);
        DigestArgs aw = a;
        aw.only = bad;
        aw.counts = rcount;
        k_digest_wave<0><<<gridw, 64, lds, st>>>(aw);
        k_dchunk_fix_counts<<<gc, 256, 0, st>>>(c_rd, first_chunk, bad, rcount, bound, c_count);
        rc = scan(c_count, bound + 1, true);  // c_count[g]: where chunk g's bytes go in the concatenated output
        if (rc != SPX_OK) return rc;
        k_dchunk_read_offsets<<<gq, 256, 0, st>>>(first_chunk, c_count, nreads, d_out_offs);
        DigestArgs au = a;
        au.offs = c_start;
        au.out_offs = c_count;
        au.nreads = bound;
        au.out = d_out;
        const uint32_t grid2 = (uint32_t)(groups < cus * 16 ? groups : cus * 16);
        if (kind == SPX_DIGEST_PROMOTED)
            k_digest_unstash<SPX_DIGEST_PROMOTED><<<grid2, 64, 0, st>>>(au, stash);
        else
            k_digest_unstash<SPX_DIGEST_DNA><<<grid2, 64, 0, st>>>(au, stash);
        aw.out_offs = d_out_offs;
        aw.out = d_out;
        k_digest_wave<1><<<gridw, 64, lds, st>>>(aw);
        SPX_HIP(hipGetLastError());
        k_zero_tail<<<1, 64, 0, st>>>(d_out_offs, nreads, d_out);
        SPX_HIP(hipGetLastError());
        return SPX_OK;
    }
    if (lanes) {
        // one pass over the reads: minimizer bytes parked in a scratch buffer of the input's size,
        // moved to their place once the offsets are known
        const bool park = want_parked && kind == SPX_DIGEST_PROMOTED;
        uint8_t* stash = park ? d_out : nullptr;
        if (!park) SPX_HIP(scr.get((void**)&stash, total_chars + 64));
        const uint64_t groups = (nreads + 63) / 64;
        // the tile: what 64 reads of mean length take (+ 3 %), in steps of 1 KB; a group that needs more goes through it in
        // several (overlapping) tiles
        uint64_t want = 64 * mean_len + 64 * mean_len / 32 + 64;
        want = (want + 1023) / 1024 * 1024;
        a.tile = (uint32_t)(want < 4096 ? 4096 : (want > TILE_MAX ? TILE_MAX : want));
        const size_t lds_dyn = a.tile + 16;
        const uint64_t per_cu = (160 * 1024) / (lds_dyn + 256 + 64 + 64);
        const uint64_t waves = cus * (per_cu > 16 ? 16 : per_cu);
        const uint32_t grid = (uint32_t)(groups < waves ? groups : waves);
        a.out = stash;
        if (kind == SPX_DIGEST_PROMOTED)
            (a.wsz == 8 && k == 4) ? k_digest_lanes<SPX_DIGEST_PROMOTED, true><<<grid, 64, lds_dyn, st>>>(a)
                       : k_digest_lanes<SPX_DIGEST_PROMOTED, false><<<grid, 64, lds_dyn, st>>>(a);
        else
            (a.wsz == 8 && k == 4) ? k_digest_lanes<SPX_DIGEST_DNA, true><<<grid, 64, lds_dyn, st>>>(a)
                       : k_digest_lanes<SPX_DIGEST_DNA, false><<<grid, 64, lds_dyn, st>>>(a);
        SPX_HIP(hipGetLastError());
        int rc = scan_counts();
        if (rc != SPX_OK) return rc;
        if (park) {
            *parked = true;
            return SPX_OK;
        }
        a.out = d_out;
        const uint32_t grid2 = (uint32_t)(groups < cus * 16 ? groups : cus * 16);
        if (kind == SPX_DIGEST_PROMOTED)
            k_digest_unstash<SPX_DIGEST_PROMOTED><<<grid2, 64, 0, st>>>(a, stash);
        else
            k_digest_unstash<SPX_DIGEST_DNA><<<grid2, 64, 0, st>>>(a, stash);
        SPX_HIP(hipGetLastError());
    } else {
        const size_t lds = 2 * (size_t)ring + 256;
        const uint32_t grid = (uint32_t)(nreads < cus * 64 ? nreads : cus * 64);
        k_digest_wave<0><<<grid, 64, lds, st>>>(a);
        SPX_HIP(hipGetLastError());
        int rc = scan_counts();
        if (rc != SPX_OK) return rc;
        k_digest_wave<1><<<grid, 64, lds, st>>>(a);
    }
    k_zero_tail<<<1, 64, 0, st>>>(d_out_offs, nreads, d_out);
    SPX_HIP(hipGetLastError());
    return SPX_OK;
}

}  // namespace spx
