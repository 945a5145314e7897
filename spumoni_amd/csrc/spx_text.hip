// spx_text.hip -- the output files' text, written on the device (hand-written HIP for gfx950).
//
// The reference writes every vector as text, one line of "<value> " per read under a ">id" line
// (/root/reference/src/compute_ms_pml.cpp:1001-1010, 1182-1205: std::ostream_iterator<size_t>(file, " ")).
// At GPU speed turning 8 * 10^8 numbers into digits IS the job of `spumoni run` (round 2: 0.75 s of a 1.1 s run on
// 16 host cores), and the numbers are already in HBM.  So the values lines are produced here:
//   k_text_count   per read, the bytes of its values line (digits + one blank per value, the newline) plus the gap
//                  the caller wants in front of it for the ">id\n" line (ids never travel to the device);
//   (hipCUB scan)  where every read's record starts in the file's new tail;
//   k_text_write   the digits, at their final place.
// One wavefront per read: a lane takes a value, a wavefront prefix sum of the widths places it.  The host copies
// text instead of values over PCIe (about as many bytes: "12 " against a 16-bit 12), drops the headers into the
// gaps and pwrite()s.
#include <hipcub/hipcub.hpp>

#include "spx_internal.h"

namespace spx {

namespace {

constexpr int TEXT_TPB = 256;

__device__ __forceinline__ uint32_t dec_width(uint64_t v) {  // decimal digits of v
    uint32_t d = 1;
    while (v >= 10) {
        v /= 10;
        ++d;
    }
    return d;
}

template <class T>
__device__ __forceinline__ uint64_t load_value(const void* vals, uint64_t i) {
    return (uint64_t) reinterpret_cast<const T*>(vals)[i];
}

// inclusive prefix sum over the 64 lanes of a wavefront
__device__ __forceinline__ uint32_t wave_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t o = __shfl_up(v, s);
        if (lane >= (uint32_t)s) v += o;
    }
    return v;
}

template <class T>
__global__ void __launch_bounds__(TEXT_TPB) k_text_count(const void* vals, const uint64_t* offs, const uint32_t* gap,
                                                         uint64_t nreads, uint64_t* line_bytes) {
    const uint64_t q = (blockIdx.x * (uint64_t)TEXT_TPB + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (q > nreads) return;
    if (q == nreads) {  // (the scan then leaves the total at line_start[nreads])
        if (lane == 0) line_bytes[q] = 0;
        return;
    }
    const uint64_t a = offs[q], b = offs[q + 1];
    uint64_t sum = 0;  // (a read of 2 * 10^8 characters and more has a line of over 2^32 bytes)
    for (uint64_t i = a + lane; i < b; i += 64) sum += dec_width(load_value<T>(vals, i)) + 1;
    for (int s = 32; s > 0; s >>= 1) sum += __shfl_xor(sum, s);
    if (lane == 0) line_bytes[q] = sum + 1 + (gap ? gap[q] : 0);
}

template <class T>
__global__ void __launch_bounds__(TEXT_TPB) k_text_write(const void* vals, const uint64_t* offs, const uint32_t* gap,
                                                         uint64_t nreads, const uint64_t* line_start, char* out) {
    const uint64_t q = (blockIdx.x * (uint64_t)TEXT_TPB + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (q >= nreads) return;
    const uint64_t a = offs[q], b = offs[q + 1];
    uint64_t pos = line_start[q] + (gap ? gap[q] : 0);
    for (uint64_t i0 = a; i0 < b; i0 += 64) {
        const uint64_t i = i0 + lane;
        uint64_t v = i < b ? load_value<T>(vals, i) : 0;
        const uint32_t w = i < b ? dec_width(v) + 1 : 0;
        const uint32_t incl = wave_scan(w, lane);
        if (w) {
            char* p = out + pos + (incl - w);
            p[w - 1] = ' ';
            for (int j = (int)w - 2; j >= 0; --j) {
                p[j] = (char)('0' + (uint32_t)(v % 10));
                v /= 10;
            }
        }
        pos += __shfl(incl, 63);
    }
    if (lane == 0) out[pos] = '\n';
}

}  // namespace

// text of one vector: line_start gets nreads + 1 offsets (the last one is the stream's size); *d_text is sized by
// the caller after reading that size back (launch_text_write)
int launch_text_count(const void* d_vals, int value_bytes, const uint64_t* d_offs, const uint32_t* d_gap, uint64_t nreads,
                      uint64_t* d_line_bytes, uint64_t* d_line_start, void* d_cub, size_t cub_bytes, hipStream_t st) {
    const unsigned grid = (unsigned)(((nreads + 1) * 64 + TEXT_TPB - 1) / TEXT_TPB);
    switch (value_bytes) {
        case 2:
            k_text_count<uint16_t><<<grid, TEXT_TPB, 0, st>>>(d_vals, d_offs, d_gap, nreads, d_line_bytes);
            break;
        case 4:
            k_text_count<uint32_t><<<grid, TEXT_TPB, 0, st>>>(d_vals, d_offs, d_gap, nreads, d_line_bytes);
            break;
        default:
            k_text_count<uint64_t><<<grid, TEXT_TPB, 0, st>>>(d_vals, d_offs, d_gap, nreads, d_line_bytes);
    }
    SPX_HIP(hipGetLastError());
    SPX_HIP(hipcub::DeviceScan::ExclusiveSum(d_cub, cub_bytes, d_line_bytes, d_line_start, (int)(nreads + 1), st));
    return SPX_OK;
}

size_t text_scan_bytes(uint64_t nreads) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (int)(nreads + 1), nullptr);
    return bytes;
}

int launch_text_write(const void* d_vals, int value_bytes, const uint64_t* d_offs, const uint32_t* d_gap, uint64_t nreads,
                      const uint64_t* d_line_start, char* d_text, hipStream_t st) {
    if (nreads == 0) return SPX_OK;
    const unsigned grid = (unsigned)((nreads * 64 + TEXT_TPB - 1) / TEXT_TPB);
    switch (value_bytes) {
        case 2:
            k_text_write<uint16_t><<<grid, TEXT_TPB, 0, st>>>(d_vals, d_offs, d_gap, nreads, d_line_start, d_text);
            break;
        case 4:
            k_text_write<uint32_t><<<grid, TEXT_TPB, 0, st>>>(d_vals, d_offs, d_gap, nreads, d_line_start, d_text);
            break;
        default:
            k_text_write<uint64_t><<<grid, TEXT_TPB, 0, st>>>(d_vals, d_offs, d_gap, nreads, d_line_start, d_text);
    }
    SPX_HIP(hipGetLastError());
    return SPX_OK;
}

}  // namespace spx
