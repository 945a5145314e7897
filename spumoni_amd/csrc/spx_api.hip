#include <sys/stat.h>
// spx_api.hip -- the C-ABI of libspumoni_gpu.so (include/spumoni_gpu.h).
// No CPU fallback exists anywhere in this library: without a gfx950 device every
// entry point that needs one returns SPX_E_NODEVICE.
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <memory>
#include <random>
#include <thread>
#include <vector>

#include "spx_internal.h"

namespace spx {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    return SPX_E_HIP;
}

static int usable_devices() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

static int select_device(int device) {
    const int n = usable_devices();
    if (n <= 0) {
        set_error("no HIP device visible: libspumoni_gpu has no CPU fallback");
        return SPX_E_NODEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (0..%d)", device, n - 1);
        return SPX_E_ARG;
    }
    SPX_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SPX_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device,
                  prop.gcnArchName);
        return SPX_E_NODEVICE;
    }
    return SPX_OK;
}

struct HostFree {
    void operator()(void* p) const { free(p); }
};

static bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize((size_t)sz);
    bool ok = sz == 0 || fread(out.data(), 1, (size_t)sz, f) == (size_t)sz;
    fclose(f);
    return ok;
}

// 5-byte little-endian records (THRBYTES / SSABYTES, include/common.hpp:59-60)
static void unpack5(const std::vector<uint8_t>& raw, size_t stride, size_t pick, std::vector<uint64_t>& out) {
    // 10^9 records take seconds on one thread (and so does first touching 8 GB of output): eight threads
    const size_t n = raw.size() / (5 * stride);
    out.resize(n);
    const unsigned nt = n >= (1u << 22) ? 8 : 1;
    auto work = [&](unsigned t) {
        for (size_t i = n * t / nt, e = n * (t + 1) / nt; i < e; ++i) {
            uint64_t v = 0;
            std::memcpy(&v, raw.data() + (i * stride + pick) * 5, 5);
            out[i] = v;
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
}

}  // namespace spx

using namespace spx;

extern "C" {

const char* spx_last_error(void) { return g_err.c_str(); }

int spx_device_count(void) { return usable_devices(); }

void spx_index_free(spx_index* ix) {
    if (!ix) return;
    static const bool trace = getenv("SPX_FREE_TRACE") != nullptr;
#define SPX_FT(what) do { if (trace) fprintf(stderr, "[spx] free %p: %s\n", (void*)ix, what); } while (0)
    SPX_FT("begin");
    (void)hipSetDevice(ix->device);
    if (ix->owner) {
        ix->owner.reset();  // shared with same-device clones: the last handle frees the arrays
    } else {
        void** arr[spx_index::NARR];
        index_arrays(ix, arr);
        for (void** a : arr)
            if (*a) (void)hipFree(*a);
    }
    SPX_FT("arrays released");
    if (ix->counters) (void)hipFree(ix->counters);
    if (ix->ctx_stream) (void)hipStreamDestroy(ix->ctx_stream);
    if (ix->h_pub) (void)hipHostFree(ix->h_pub);
    if (ix->ev_wait) (void)hipEventDestroy(ix->ev_wait);
    SPX_FT("stream destroyed");
    for (auto& st : ix->pipe_s)
        if (st) (void)hipStreamDestroy(st);
    for (int c = 0; c < spx_index::PIPE_CHUNKS; ++c) {
        if (ix->pipe_in[c]) (void)hipEventDestroy(ix->pipe_in[c]);
        if (ix->pipe_k[c]) (void)hipEventDestroy(ix->pipe_k[c]);
    }
    for (auto& sc : ix->scratch)
        if (sc.p) (void)hipFree(sc.p);
    for (auto& sc : ix->chunk_scr)
        if (sc.p) (void)hipFree(sc.p);
    for (auto& sc : ix->digest_scr)
        if (sc.p) (void)hipFree(sc.p);
    if (ix->ev_dig) (void)hipEventDestroy(ix->ev_dig);
    if (ix->ev0) (void)hipEventDestroy(ix->ev0);
    if (ix->ev1) (void)hipEventDestroy(ix->ev1);
    if (ix->ev_done) (void)hipEventDestroy(ix->ev_done);
    SPX_FT("done");
#undef SPX_FT
    delete ix;
}

// bonsai's RollingHasher draws its character table from a Mersenne twister seeded with 1337
// and keeps 8 bits (our reading of CharacterHash): entry c = c-th output
static void default_charhash(uint8_t out[4]) {
    std::mt19937 gen(1337u);
    uint8_t table[256];
    for (int c = 0; c < 256; ++c) table[c] = (uint8_t)(gen() & 0xffu);
    out[0] = table['A'];
    out[1] = table['C'];
    out[2] = table['G'];
    out[3] = table['T'];
}

// the stream of the handle's host-buffer queries (created on first use; callers hold host_mu)
static int ctx_stream_of(spx_index* ix, hipStream_t* out) {
    if (!ix->ctx_stream) SPX_HIP(hipStreamCreateWithFlags(&ix->ctx_stream, hipStreamNonBlocking));
    *out = ix->ctx_stream;
    return SPX_OK;
}

// Waiting for the handle's stream: spinning (hipStreamSynchronize: lowest latency) or, "blocking_sync", asleep on an event.
static int ctx_wait(spx_index* ix, hipStream_t st) {
    if (!ix->blocking_sync) {
        SPX_HIP(hipStreamSynchronize(st));
        return SPX_OK;
    }
    if (!ix->ev_wait) SPX_HIP(hipEventCreateWithFlags(&ix->ev_wait, hipEventBlockingSync | hipEventDisableTiming));
    SPX_HIP(hipEventRecord(ix->ev_wait, st));
    SPX_HIP(hipEventSynchronize(ix->ev_wait));
    return SPX_OK;
}

// Words the host is waiting for, written into page-locked host memory by a kernel (spx_internal.h: h_pub): dst[i] = *src[i]
// (0 for a null source) for i < 4, then nwords words of `more`.
struct PubSrc {
    const uint64_t* one[4];
    const uint64_t* more;
    int nmore;
};
__global__ void k_publish(uint64_t* dst, PubSrc src) {
    const int t = (int)threadIdx.x;
    if (t < 4) dst[t] = src.one[t] ? *src.one[t] : 0;
    if (t >= 4 && t - 4 < src.nmore) dst[t] = src.more[t - 4];
    __threadfence_system();
}
static int publish(spx_index* ix, const PubSrc& src, hipStream_t st) {
    if (!ix->h_pub) {
        SPX_HIP(hipHostMalloc((void**)&ix->h_pub, 64 * sizeof(uint64_t), hipHostMallocMapped | hipHostMallocCoherent));  // (coherent whatever HIP_HOST_COHERENT says)
        SPX_HIP(hipHostGetDevicePointer((void**)&ix->h_pub_dev, ix->h_pub, 0));
    }
    k_publish<<<1, 64, 0, st>>>(ix->h_pub_dev, src);
    SPX_HIP(hipGetLastError());
    return SPX_OK;
}

// The text is one of the shared arrays: it can only be replaced while no same-device clone reads it (callers hold mu).
static int own_arrays_alone(spx_index* ix) {
    if (!ix->owner) return SPX_OK;
    if (ix->owner.use_count() > 1) {
        set_error("the index shares its arrays with a clone on the same device: set or rebuild the text before cloning");
        return SPX_E_ARG;
    }
    for (void*& a : ix->owner->p) a = nullptr;  // back to plain ownership by this handle
    ix->owner.reset();
    return SPX_OK;
}

// counters, events and knobs every index carries, however its arrays came to be
static int init_runtime(spx_index* ix) {
    default_charhash(ix->charhash);
    if (const char* e = getenv("SPX_WAVES_PER_CU")) ix->waves_per_cu = atoi(e);  // experiment knob
    SPX_HIP(hipMalloc((void**)&ix->counters, sizeof(WalkCounters)));
    SPX_HIP(hipMemset(ix->counters, 0, sizeof(WalkCounters)));
    SPX_HIP(hipEventCreate(&ix->ev0));
    SPX_HIP(hipEventCreate(&ix->ev1));
    SPX_HIP(hipEventCreateWithFlags(&ix->ev_done, hipEventDisableTiming));
    SPX_HIP(hipEventCreateWithFlags(&ix->ev_dig, hipEventDisableTiming));
    SPX_HIP(hipDeviceSynchronize());
    return SPX_OK;
}

static int from_runs_impl(spx_index* ix, const uint8_t* heads, const uint64_t* lens,
                          const uint64_t* thr, uint64_t r, const uint64_t* ssa, const uint64_t* esa,
                          const uint64_t* ds, const uint64_t* de, int where) {
    if (!heads || !lens || !thr || r == 0) {
        set_error("heads, lens and thr must be non-null and r > 0");
        return SPX_E_ARG;
    }
    if ((ssa == nullptr) != (esa == nullptr) || (ds == nullptr) != (de == nullptr)) {
        set_error("ssa/esa and doc_start/doc_end must be given in pairs");
        return SPX_E_ARG;
    }
    ix->r = r;
    struct Tmp {
        void* p = nullptr;
        ~Tmp() {
            if (p) (void)hipFree(p);
        }
    } t[7];
    const void* src[7] = {heads, lens, thr, ssa, esa, ds, de};
    const void* dev[7];
    for (int i = 0; i < 7; ++i) {
        dev[i] = src[i];
        if (where == 0 && src[i]) {
            const size_t bytes = (i == 0 ? 1 : 8) * r;
            SPX_HIP(hipMalloc(&t[i].p, bytes));
            SPX_HIP(hipMemcpy(t[i].p, src[i], bytes, hipMemcpyHostToDevice));
            dev[i] = t[i].p;
        }
    }
    // (host arrays: the device copies are this function's own and are given back before the fat table is sized)
    int rc = flatten_on_device(ix, (const uint8_t*)dev[0], (const uint64_t*)dev[1],
                               (const uint64_t*)dev[2], (const uint64_t*)dev[3],
                               (const uint64_t*)dev[4], (const uint64_t*)dev[5],
                               (const uint64_t*)dev[6], [&] {
                                   for (Tmp& x : t) {
                                       if (x.p) (void)hipFree(x.p);
                                       x.p = nullptr;
                                   }
                               });
    if (rc != SPX_OK) return rc;
    return init_runtime(ix);
}

spx_index* spx_index_from_runs(const uint8_t* heads, const uint64_t* lens, const uint64_t* thr,
                               uint64_t r, const uint64_t* ssa, const uint64_t* esa,
                               const uint64_t* doc_start, const uint64_t* doc_end, int where,
                               int device) {
    if (select_device(device) != SPX_OK) return nullptr;
    if (where != 0 && where != 1) {
        set_error("where must be 0 (host) or 1 (device)");
        return nullptr;
    }
    spx_index* ix = new spx_index();
    ix->device = device;
    if (from_runs_impl(ix, heads, lens, thr, r, ssa, esa, doc_start, doc_end, where) != SPX_OK) {
        spx_index_free(ix);
        return nullptr;
    }
    return ix;
}

spx_index* spx_index_load_raw(const char* prefix, int mode, int device) {
    if (!prefix) {
        set_error("prefix is null");
        return nullptr;
    }
    const std::string p(prefix);
    std::vector<uint8_t> heads, raw;
    std::vector<uint64_t> lens, thr, ssa, esa;
    if (!read_file(p + ".bwt.heads", heads) || heads.empty()) {
        set_error("cannot read %s.bwt.heads", prefix);
        return nullptr;
    }
    const uint64_t r = heads.size();
    if (!read_file(p + ".bwt.len", raw) || raw.size() != r * 5) {
        set_error("%s.bwt.len missing or not %llu 5-byte records", prefix, (unsigned long long)r);
        return nullptr;
    }
    unpack5(raw, 1, 0, lens);
    if (!read_file(p + ".thr_pos", raw) || raw.size() != r * 5) {
        set_error("%s.thr_pos missing or not %llu 5-byte records", prefix, (unsigned long long)r);
        return nullptr;
    }
    unpack5(raw, 1, 0, thr);
    if (mode == SPX_MODE_MS) {
        uint64_t n = 0;
        for (uint64_t v : lens) n += v;
        for (int which = 0; which < 2; ++which) {
            const std::string path = p + (which ? ".esa" : ".ssa");
            if (!read_file(path, raw) || raw.size() != r * 10) {
                set_error("%s missing or not %llu (left,right) 5-byte pairs", path.c_str(),
                          (unsigned long long)r);
                return nullptr;
            }
            std::vector<uint64_t>& dst = which ? esa : ssa;
            unpack5(raw, 2, 1, dst);
            for (auto& v : dst) v = v ? v - 1 : n - 1;  // compute_ms_pml.cpp:433
        }
    }
    return spx_index_from_runs(heads.data(), lens.data(), thr.data(), r,
                               ssa.empty() ? nullptr : ssa.data(), esa.empty() ? nullptr : esa.data(),
                               nullptr, nullptr, 0, device);
}

int spx_index_stats(const spx_index* ix, uint64_t* n, uint64_t* r) {
    if (!ix) {
        set_error("index is null");
        return SPX_E_ARG;
    }
    if (n) *n = ix->n;
    if (r) *r = ix->r;
    return SPX_OK;
}

int spx_index_device_bytes(const spx_index* ix, uint64_t* bytes) {
    if (!ix || !bytes) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    *bytes = ix->device_bytes + ix->n_text;
    return SPX_OK;
}

int spx_index_set_text(spx_index* ix, const uint8_t* text, uint64_t n_text, int where) {
    if (!ix || !text) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    const bool unchecked = (where & SPX_TEXT_UNCHECKED) != 0;
    where &= ~SPX_TEXT_UNCHECKED;
    if (!unchecked && n_text + 1 != ix->n) {
        // the BWT of a text of n_text characters plus its terminator has n_text + 1 positions
        set_error("text has %llu characters but the index was built over %llu (+ terminator): not the indexed text",
                  (unsigned long long)n_text, (unsigned long long)(ix->n - 1));
        return SPX_E_FORMAT;
    }
    std::lock_guard<std::mutex> g(ix->mu);
    SPX_HIP(hipSetDevice(ix->device));
    {
        const int rc_own = own_arrays_alone(ix);
        if (rc_own != SPX_OK) return rc_own;
    }
    if (ix->text) (void)hipFree(ix->text);
    ix->text = nullptr;
    ix->n_text = 0;
    bind_view(ix);
    SPX_HIP(hipMalloc((void**)&ix->text, n_text + 16));
    SPX_HIP(hipMemcpy(ix->text, text, n_text, where ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    SPX_HIP(hipMemset(ix->text + n_text, 0, 16));
    ix->n_text = n_text;
    ix->arr_bytes[A_TEXT] = n_text + 16;
    bind_view(ix);
    if (!unchecked && ix->has_samples) {
        // every run's first BWT character is the text character in front of its suffix:
        // text[samples_start[k]] == head of run k, for all r runs (one pass over the samples)
        // (whatever fails from here on, an UNCHECKED text must not stay bound to the index)
        auto drop_text = [&] {
            (void)hipFree(ix->text);
            ix->text = nullptr;
            ix->n_text = 0;
            ix->arr_bytes[A_TEXT] = 0;
            bind_view(ix);
        };
        unsigned long long* d_bad = nullptr;
        int rc = SPX_OK;
        if (hipMalloc((void**)&d_bad, 8) != hipSuccess || hipMemset(d_bad, 0, 8) != hipSuccess) {
            set_error("hipMalloc / hipMemset failed while checking the text");
            rc = SPX_E_HIP;
        }
        if (rc == SPX_OK) rc = launch_text_check(ix, d_bad, nullptr);
        unsigned long long bad = 0;
        if (rc == SPX_OK && hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost) != hipSuccess) {
            set_error("hipMemcpy failed while checking the text");
            rc = SPX_E_HIP;
        }
        if (d_bad) (void)hipFree(d_bad);
        if (rc != SPX_OK) {
            drop_text();
            return rc;
        }
        if (bad) {
            drop_text();
            set_error("text disagrees with the index at %llu of %llu runs (text[samples_start[k]] must be the head "
                      "of run k): not the text this index was built from", bad, (unsigned long long)ix->r);
            return SPX_E_FORMAT;
        }
    }
    return SPX_OK;
}

// A caller's fingerprint of what the index was built from (the host harness: names, sizes and modification times of
// the index files): saved with the flat-layout cache and handed back after spx_index_load_flat, so that a cache left
// over from an index that has since been rebuilt under the same prefix is recognised and not used.
int spx_index_set_source_tag(spx_index* ix, const char* tag) {
    if (!ix || !tag) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> g(ix->mu);
    snprintf(ix->source_tag, sizeof ix->source_tag, "%s", tag);
    return SPX_OK;
}

const char* spx_index_source_tag(const spx_index* ix) { return ix ? ix->source_tag : ""; }

int spx_index_rebuild_text(spx_index* ix) {
    if (!ix) {
        set_error("index is null");
        return SPX_E_ARG;
    }
    if (!ix->has_samples) {
        set_error("the text can only be rebuilt from an index with SA samples (an MS index)");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> g(ix->mu);
    SPX_HIP(hipSetDevice(ix->device));
    const uint64_t n_text = ix->n - 1;
    {
        const int rc_own = own_arrays_alone(ix);
        if (rc_own != SPX_OK) return rc_own;
    }
    if (ix->text) (void)hipFree(ix->text);
    ix->text = nullptr;
    ix->n_text = 0;
    ix->arr_bytes[A_TEXT] = 0;
    bind_view(ix);
    uint8_t* d_text = nullptr;
    unsigned long long* d_cnt = nullptr;
    SPX_HIP(hipMalloc((void**)&d_text, n_text + 16));
    SPX_HIP(hipMemset(d_text, 0, n_text + 16));
    SPX_HIP(hipMalloc((void**)&d_cnt, 16));
    SPX_HIP(hipMemset(d_cnt, 0, 16));
    int rc = launch_text_from_index(ix, d_text, n_text, d_cnt, nullptr);
    unsigned long long stuck = 0;
    if (rc == SPX_OK && hipMemcpy(&stuck, d_cnt, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = SPX_E_HIP;
    if (rc == SPX_OK && stuck) {
        set_error("the run structure is not a permutation (%llu LF chains did not end): corrupt index", stuck);
        rc = SPX_E_FORMAT;
    }
    if (rc == SPX_OK) {
        ix->text = d_text;
        ix->n_text = n_text;
        ix->arr_bytes[A_TEXT] = n_text + 16;
        bind_view(ix);
        // (the chains partition the BWT positions of a consistent run structure; the check below confirms the
        // samples' positions, which is what spx_index_set_text checks of a text handed in)
        SPX_HIP(hipMemset(d_cnt, 0, 8));
        rc = launch_text_check(ix, d_cnt, nullptr);
        unsigned long long bad = 0;
        if (rc == SPX_OK && hipMemcpy(&bad, d_cnt, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = SPX_E_HIP;
        if (rc == SPX_OK && bad) {
            set_error("the rebuilt text disagrees with the index at %llu runs: SA samples and run structure do not belong together", bad);
            rc = SPX_E_FORMAT;
        }
    } else {
        (void)hipFree(d_text);
    }
    (void)hipFree(d_cnt);
    return rc;
}

int spx_index_copy_text(spx_index* ix, uint8_t* out, uint64_t capacity, int where, uint64_t* n_text) {
    if (!ix) {
        set_error("index is null");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> g(ix->mu);
    if (n_text) *n_text = ix->n_text;
    if (!out) return SPX_OK;  // size query
    if (!ix->text || capacity < ix->n_text) {
        set_error(ix->text ? "buffer too small for the text" : "the index has no text");
        return SPX_E_ARG;
    }
    SPX_HIP(hipSetDevice(ix->device));
    SPX_HIP(hipMemcpy(out, ix->text, ix->n_text, where ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return SPX_OK;
}

void* spx_host_alloc(size_t bytes) {
    if (usable_devices() <= 0) {
        set_error("no HIP device visible: libspumoni_gpu has no CPU fallback");
        return nullptr;
    }
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hip_fail(e, "hipHostMalloc", __FILE__, __LINE__);
        return nullptr;
    }
    return p;
}

void spx_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int spx_host_register(void* p, size_t bytes) {
    if (!p || bytes == 0) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    if (usable_devices() <= 0) {
        set_error("no HIP device visible: libspumoni_gpu has no CPU fallback");
        return SPX_E_NODEVICE;
    }
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return hip_fail(e, "hipHostRegister", __FILE__, __LINE__);
    }
    return SPX_OK;
}

int spx_host_unregister(void* p) {
    if (!p) return SPX_OK;
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return hip_fail(e, "hipHostUnregister", __FILE__, __LINE__);
    }
    return SPX_OK;
}

int spx_set_option(spx_index* ix, const char* key, int64_t value) {
    if (!ix || !key) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> g(ix->mu);
    if (!strcmp(key, "blocking_sync")) {  // 1: spx_query_text_begin / _fetch sleep while they wait for the device (several
        ix->blocking_sync = value != 0;    // query contexts of one process: a core per spinning waiter is a core less for the host's work)
        return SPX_OK;
    }
    if (!strcmp(key, "waves_per_cu")) {
        ix->waves_per_cu = (int)value;
        return SPX_OK;
    }
    if (!strcmp(key, "lanes_per_wave")) {
        ix->force_lanes_per_wave = (int)value;
        return SPX_OK;
    }
    if (!strcmp(key, "chunk_mode")) {  // long-read chunking: 0 automatic, 1 never, 2 always
        ix->chunk_mode = (int)value;
        return SPX_OK;
    }
    if (!strcmp(key, "chunk_len")) {  // chunk size in characters (0 = automatic)
        ix->chunk_len = (int)value;
        return SPX_OK;
    }
    if (!strcmp(key, "chunk_shift")) {  // log2 of the chunk size (0 = automatic)
        ix->chunk_shift = (int)value;
        return SPX_OK;
    }
    if (!strcmp(key, "digest_parked")) {  // 0 automatic, 1 never, 2 whenever the walk can (tests, A/B)
        ix->digest_parked = (int)value;
        return SPX_OK;
    }
    if (!strcmp(key, "digest_kernel")) {
        ix->force_digest_kernel = (int)value;
        return SPX_OK;
    }
    if (!strcmp(key, "minimizer_charhash")) {
        for (int c = 0; c < 4; ++c) ix->charhash[c] = (uint8_t)((uint64_t)value >> (8 * c));
        return SPX_OK;
    }
    set_error("unknown option '%s'", key);
    return SPX_E_ARG;
}

uint64_t spx_digest_capacity(int kind, uint32_t k, uint64_t total_chars) {
    // every k-mer can be reported once: <= total_chars values of 1 byte (-m) or k bytes (-a);
    // + the padding spx_query_batch_device asks of its d_seqs
    const uint64_t body = (kind == SPX_DIGEST_DNA ? (uint64_t)(k ? k : 1) : 1ull) * total_chars;
    return ((body + 3) / 4) * 4 + 32;
}

int spx_digest_batch_device(spx_index* ix, int kind, uint32_t k, uint32_t w, const uint8_t* d_seqs,
                            const uint64_t* d_offsets, uint64_t nreads, uint64_t total_chars,
                            uint8_t* d_out_seqs, uint64_t out_capacity, uint64_t* d_out_offsets, void* stream) {
    if (!ix || !d_seqs || !d_offsets || !d_out_seqs || !d_out_offsets) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    if (((uintptr_t)d_seqs & 15) != 0) {
        set_error("d_seqs must be 16-byte aligned (and readable for round_up(total_chars, 16) + 16 bytes)");
        return SPX_E_ARG;
    }
    if (out_capacity < spx_digest_capacity(kind, k, total_chars)) {
        set_error("d_out_seqs must hold spx_digest_capacity() = %llu bytes",
                  (unsigned long long)spx_digest_capacity(kind, k, total_chars));
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> g(ix->mu);
    SPX_HIP(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;
    return launch_digest(ix, kind, k, w, d_seqs, d_offsets, nreads, total_chars, d_out_seqs, d_out_offsets, st);
}

// Digestion in front of a walk on the same device (run -m / -a, compute_ms_pml.cpp:919-923).  When the walk that follows
// is the plain one over compact rows (k_walk_fast) and needs the reads only as characters, the digested reads stay where the
// digestion parked them -- read q's at d_dig[d_offs[q] ..] -- and *in_starts = d_offs tells the walk so: the pass that would
// concatenate them (0.5 of 2.1 ms per 10^7 x 200 bp) is not made.  d_dig_offs are the offsets of the concatenation either
// way: that is where the results go.
static int digest_for_walk(spx_index* ix, int mode, bool ms_lengths, int kind, uint32_t k, uint32_t w, const uint8_t* d_raw,
                           const uint64_t* d_offs, uint64_t nreads, uint64_t total_in, uint8_t* d_dig, uint64_t cap,
                           uint64_t* d_dig_offs, void* stream, const uint64_t** in_starts) {
    *in_starts = nullptr;
    if (!ix || !d_raw || !d_offs || !d_dig || !d_dig_offs) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    if (((uintptr_t)d_raw & 15) != 0) {
        set_error("d_seqs must be 16-byte aligned (and readable for round_up(total_chars, 16) + 16 bytes)");
        return SPX_E_ARG;
    }
    if (cap < spx_digest_capacity(kind, k, total_in)) {
        set_error("the buffer for the digested reads must hold spx_digest_capacity() = %llu bytes",
                  (unsigned long long)spx_digest_capacity(kind, k, total_in));
        return SPX_E_ARG;
    }
    static const bool old_walk = getenv("SPX_OLD_WALK") != nullptr;
    // (SPX_DIGEST_PARKED: the "digest_parked" option from the environment, for callers that have no handle on it -- the CLI's tests)
    static const int env_parked = getenv("SPX_DIGEST_PARKED") ? atoi(getenv("SPX_DIGEST_PARKED")) : -1;
    const int parked_mode = env_parked >= 0 ? env_parked : ix->digest_parked;
    std::lock_guard<std::mutex> g(ix->mu);
    SPX_HIP(hipSetDevice(ix->device));
    if (ix->num_cus == 0) {
        hipDeviceProp_t prop;
        SPX_HIP(hipGetDeviceProperties(&prop, ix->device));
        ix->num_cus = prop.multiProcessorCount;
    }
    // (automatic: only batches that fill the device's lanes with reads -- the others may be long reads that the chunked
    // walk should get, and that one takes its reads by their offsets)
    bool park = parked_mode != 1 && ix->rows != nullptr && ix->view.compact && !old_walk && nreads > 0 && nreads < (1ull << 31) &&
                !(mode == SPX_MODE_MS && ms_lengths) && ix->force_lanes_per_wave == 0 &&
                (parked_mode == 2 || nreads * 2 > (uint64_t)ix->num_cus * 20 * 64);
    const int rc = launch_digest(ix, kind, k, w, d_raw, d_offs, nreads, total_in, d_dig, d_dig_offs, (hipStream_t)stream, &park);
    if (rc == SPX_OK && park) *in_starts = d_offs;
    return rc;
}

// grow-only device scratch owned by the index (no hipMalloc/hipFree per call); callers hold host_mu
static int ensure_scratch(spx_index* ix, int slot, size_t bytes, void** out) {
    spx_index::Scratch& sc = ix->scratch[slot];
    if (sc.cap < bytes) {
        if (sc.p) (void)hipFree(sc.p);
        sc.p = nullptr;
        sc.cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        SPX_HIP(hipMalloc(&sc.p, want));
        sc.cap = want;
    }
    *out = sc.p;
    return SPX_OK;
}

// The host-buffer entry points enqueue their copies on the handle's stream and leave early on any error after that.  A copy
// that is still reading the caller's seqs / offsets / gap (or writing its outputs) when the call has already failed would
// leave work in flight on memory the caller may free: every way out that is not SPX_OK waits for the stream first
// (before the copies became asynchronous a failed call never left anything behind).
namespace {
struct QuietOnError {
    hipStream_t st;
    bool device_wide;  // the pipelined batches run on three streams of the handle: wait for the device
    bool ok = false;
    explicit QuietOnError(hipStream_t s, bool wide = false) : st(s), device_wide(wide) {}
    int done(int rc) {
        ok = rc == SPX_OK;
        return rc;
    }
    ~QuietOnError() {
        if (ok) return;
        if (device_wide)
            (void)hipDeviceSynchronize();
        else
            (void)hipStreamSynchronize(st);
    }
};
}  // namespace

int spx_digest_batch(spx_index* ix, int kind, uint32_t k, uint32_t w, const uint8_t* seqs, const uint64_t* offsets,
                     uint64_t nreads, uint8_t* out_seqs, uint64_t out_capacity, uint64_t* out_offsets) {
    if (!ix || !seqs || !offsets || !out_offsets) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> hg(ix->host_mu);
    SPX_HIP(hipSetDevice(ix->device));
    const uint64_t total = nreads ? offsets[nreads] : 0;
    const uint64_t cap = spx_digest_capacity(kind, k, total);
    void *dseq = nullptr, *doff = nullptr, *dout = nullptr, *dooff = nullptr;
    int rc;
    if ((rc = ensure_scratch(ix, 6, ((total + 15) & ~15ull) + 16, &dseq)) != SPX_OK) return rc;
    if ((rc = ensure_scratch(ix, 1, (nreads + 1) * 8, &doff)) != SPX_OK) return rc;
    if ((rc = ensure_scratch(ix, 0, cap, &dout)) != SPX_OK) return rc;
    if ((rc = ensure_scratch(ix, 7, (nreads + 1) * 8, &dooff)) != SPX_OK) return rc;
    hipStream_t st = nullptr;
    if ((rc = ctx_stream_of(ix, &st)) != SPX_OK) return rc;
    QuietOnError quiet(st);
    SPX_HIP(hipMemcpyAsync(dseq, seqs, total, hipMemcpyHostToDevice, st));
    SPX_HIP(hipMemcpyAsync(doff, offsets, (nreads + 1) * 8, hipMemcpyHostToDevice, st));
    rc = spx_digest_batch_device(ix, kind, k, w, (const uint8_t*)dseq, (const uint64_t*)doff, nreads, total,
                                 (uint8_t*)dout, cap, (uint64_t*)dooff, st);
    if (rc != SPX_OK) return rc;
    SPX_HIP(hipMemcpyAsync(out_offsets, dooff, (nreads + 1) * 8, hipMemcpyDeviceToHost, st));
    SPX_HIP(hipStreamSynchronize(st));
    const uint64_t dtotal = out_offsets[nreads];
    if (dtotal > out_capacity || (dtotal && !out_seqs)) {
        set_error("out_seqs holds %llu bytes, the digested reads need %llu", (unsigned long long)out_capacity,
                  (unsigned long long)dtotal);
        return SPX_E_ARG;
    }
    if (dtotal) {
        SPX_HIP(hipMemcpyAsync(out_seqs, dout, dtotal, hipMemcpyDeviceToHost, st));
        SPX_HIP(hipStreamSynchronize(st));
    }
    return quiet.done(SPX_OK);
}

static int check_query(spx_index* ix, int mode, const void* seqs, const void* offs,
                       uint32_t* out_lengths, uint64_t* out_pointers, uint32_t* out_docs,
                       spx_class* out_class, uint64_t bin_width) {
    if (!ix || !seqs || !offs) {
        set_error("index, seqs and offsets must be non-null");
        return SPX_E_ARG;
    }
    if (mode != SPX_MODE_PML && mode != SPX_MODE_MS) {
        set_error("mode must be SPX_MODE_PML or SPX_MODE_MS");
        return SPX_E_ARG;
    }
    if (mode == SPX_MODE_PML && !out_lengths && !out_class) {
        set_error("PML mode needs out_lengths (or out_class alone: classification without the per-character values)");
        return SPX_E_ARG;
    }
    if (mode == SPX_MODE_PML && out_pointers) {
        set_error("out_pointers is only produced in MS mode");
        return SPX_E_ARG;
    }
    if (mode == SPX_MODE_MS) {
        if (!ix->has_samples) {
            set_error("MS mode needs an index built with SA samples (.ssa/.esa)");
            return SPX_E_ARG;
        }
        if (!out_pointers) {
            set_error("MS mode needs out_pointers");
            return SPX_E_ARG;
        }
        if (out_lengths && !ix->text) {
            set_error("MS lengths need the text: call spx_index_set_text first");
            return SPX_E_ARG;
        }
        if (out_class && !out_lengths) {
            set_error("MS classification needs out_lengths");
            return SPX_E_ARG;
        }
    }
    if (out_docs && !ix->has_docs) {
        set_error("document ids requested but the index has no document array");
        return SPX_E_ARG;
    }
    if (out_class && bin_width == 0) {
        set_error("bin_width must be > 0");
        return SPX_E_ARG;
    }
    return SPX_OK;
}

// narrow: d_out_lengths / d_out_docs are uint16_t arrays (the 16-bit entry points)
static int query_device_impl(spx_index* ix, int mode, const uint8_t* d_seqs, const uint64_t* d_offsets,
                             uint64_t nreads, uint64_t total_chars, uint32_t* d_out_lengths,
                             uint64_t* d_out_pointers, uint32_t* d_out_docs, spx_class* d_out_class,
                             uint64_t bin_width, uint64_t max_value_thr, void* stream, bool narrow,
                             const uint64_t* d_in_starts = nullptr, uint64_t geom_chars = 0) {
    int rc = check_query(ix, mode, d_seqs, d_offsets, d_out_lengths, d_out_pointers, d_out_docs,
                         d_out_class, bin_width);
    if (rc != SPX_OK) return rc;
    if (((uintptr_t)d_seqs & 15) != 0) {
        set_error("d_seqs must be 16-byte aligned (and readable for round_up(total_chars, 4) + 32 bytes)");
        return SPX_E_ARG;
    }
    if ((((uintptr_t)d_out_lengths | (uintptr_t)d_out_pointers | (uintptr_t)d_out_docs) & 15) != 0) {
        set_error("output buffers must be 16-byte aligned (results are written as 16-byte vectors)");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> g(ix->mu);
    SPX_HIP(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;
    // the counters / events belong to the index: a query enqueued on another stream waits for
    // the previous one (queries on one index are serialised, as the header promises)
    if (ix->have_timing && ix->last_stream != st) SPX_HIP(hipStreamWaitEvent(st, ix->ev_done, 0));
    SPX_HIP(hipMemsetAsync(ix->counters, 0, sizeof(WalkCounters), st));
    BatchArgs a{};
    a.seqs = d_seqs;
    a.offs = d_offsets;
    a.nreads = nreads;
    a.total_chars = total_chars;
    a.out_lengths = d_out_lengths;
    a.out_pointers = d_out_pointers;
    a.out_docs = d_out_docs;
    a.out_class = (mode == SPX_MODE_PML) ? d_out_class : nullptr;
    a.bin_width = bin_width;
    a.bin_magic = bin_width > 1 ? (uint64_t)(~0ull / bin_width) + 1 : 0;
    a.max_value_thr = max_value_thr;
    a.counters = ix->counters;
    a.narrow = narrow ? 1 : 0;
    a.in_starts = d_in_starts;  // (reads parked by the digestion: digest_for_walk below made sure the plain fast walk takes them)
    if ((rc = prepare_len_mask(ix, mode, a)) != SPX_OK) return rc;
    SPX_HIP(hipEventRecord(ix->ev0, st));
    bool chunked = false, wrote = false;
    if (nreads > 0) {
        // (geom_chars: what the batch is expected to hold when total_chars is only an upper bound -- reads digested on the
        // device a moment ago; it shapes the chunks, total_chars sizes the scratch)
        if (!d_in_starts && (rc = launch_walk_chunked(ix, mode, a, total_chars, st, &chunked, geom_chars)) != SPX_OK) return rc;
        if (!chunked && (rc = launch_walk(ix, mode, a, total_chars, st, &wrote)) != SPX_OK) return rc;
    }
    SPX_HIP(hipEventRecord(ix->ev1, st));
    // PML, state-machine walk: it left one bit per character; the lengths are written from them here
    if (nreads > 0 && !chunked && !wrote && (rc = launch_len_expand(ix, a, st)) != SPX_OK) return rc;
    if (mode == SPX_MODE_MS && d_out_lengths && nreads > 0) {
        a.out_class = d_out_class;
        rc = launch_ms_extend(ix, a, st);
        if (rc != SPX_OK) return rc;
    }
    SPX_HIP(hipEventRecord(ix->ev_done, st));
    ix->have_timing = true;
    ix->last_stream = st;
    return SPX_OK;
}

int spx_query_batch_device(spx_index* ix, int mode, const uint8_t* d_seqs, const uint64_t* d_offsets,
                           uint64_t nreads, uint64_t total_chars, uint32_t* d_out_lengths,
                           uint64_t* d_out_pointers, uint32_t* d_out_docs, spx_class* d_out_class,
                           uint64_t bin_width, uint64_t max_value_thr, void* stream) {
    return query_device_impl(ix, mode, d_seqs, d_offsets, nreads, total_chars, d_out_lengths, d_out_pointers,
                             d_out_docs, d_out_class, bin_width, max_value_thr, stream, false);
}

int spx_query_batch_device16(spx_index* ix, int mode, const uint8_t* d_seqs, const uint64_t* d_offsets,
                             uint64_t nreads, uint64_t total_chars, uint16_t* d_out_lengths,
                             uint64_t* d_out_pointers, uint16_t* d_out_docs, spx_class* d_out_class,
                             uint64_t bin_width, uint64_t max_value_thr, void* stream) {
    return query_device_impl(ix, mode, d_seqs, d_offsets, nreads, total_chars, (uint32_t*)d_out_lengths,
                             d_out_pointers, (uint32_t*)d_out_docs, d_out_class, bin_width, max_value_thr, stream,
                             true);
}

// device side of the host-buffer queries: d_seq / d_off are resident (scratch), results go
// through scratch slots 2..5 to the host buffers.  Caller holds host_mu.  width: bytes per length /
// doc value in the host buffers (4, or 2 for the 16-bit entry points)
static int run_and_fetch(spx_index* ix, int mode, const uint8_t* d_seq, const uint64_t* d_off, uint64_t nreads,
                         uint64_t total, void* out_lengths, uint64_t* out_pointers, void* out_docs,
                         spx_class* out_class, uint64_t bin_width, uint64_t max_value_thr, size_t width = 4,
                         const uint64_t* d_in_starts = nullptr) {
    void *dlen = nullptr, *dptr = nullptr, *ddoc = nullptr, *dcls = nullptr;
    int rc;
    if (out_lengths && (rc = ensure_scratch(ix, 2, (total + 1) * 4, &dlen)) != SPX_OK) return rc;
    if (out_pointers && (rc = ensure_scratch(ix, 3, (total + 1) * 8, &dptr)) != SPX_OK) return rc;
    if (out_docs && (rc = ensure_scratch(ix, 4, (total + 1) * 4, &ddoc)) != SPX_OK) return rc;
    if (out_class && (rc = ensure_scratch(ix, 5, (nreads + 1) * sizeof(spx_class), &dcls)) != SPX_OK) return rc;
    hipStream_t st = nullptr;
    if ((rc = ctx_stream_of(ix, &st)) != SPX_OK) return rc;
    rc = query_device_impl(ix, mode, d_seq, d_off, nreads, total, (uint32_t*)dlen, (uint64_t*)dptr,
                           (uint32_t*)ddoc, (spx_class*)dcls, bin_width, max_value_thr, st, width == 2, d_in_starts);
    if (rc != SPX_OK) return rc;
    if (out_lengths) SPX_HIP(hipMemcpyAsync(out_lengths, dlen, total * width, hipMemcpyDeviceToHost, st));
    if (out_pointers) SPX_HIP(hipMemcpyAsync(out_pointers, dptr, total * 8, hipMemcpyDeviceToHost, st));
    if (out_docs) SPX_HIP(hipMemcpyAsync(out_docs, ddoc, total * width, hipMemcpyDeviceToHost, st));
    if (out_class) SPX_HIP(hipMemcpyAsync(out_class, dcls, nreads * sizeof(spx_class), hipMemcpyDeviceToHost, st));
    WalkCounters wc;
    SPX_HIP(hipMemcpyAsync(&wc, ix->counters, sizeof wc, hipMemcpyDeviceToHost, st));
    SPX_HIP(hipStreamSynchronize(st));
    if (wc.error) {
        set_error("the walk hit %llu undefined steps (predecessor jump without a predecessor run: "
                  "thresholds are inconsistent with the BWT%s)", wc.error,
                  width == 2 ? "; or a read of 65536 characters or more with 16-bit outputs" : "");
        return SPX_E_FORMAT;
    }
    return SPX_OK;
}

// Large host batches: chunks of reads go through copy-in / walk / copy-out on three streams, so the
// PCIe transfers of one chunk overlap the kernel of another.  Offsets are absolute, so every chunk
// is launched on the same device buffers with the offsets pointer advanced.  Caller holds host_mu.
// 16-bit outputs hold values below 65536: a read that long is refused.  first_long_read: its index, or q1 (the error
// text is thread-local: the caller's thread reports)
static uint64_t first_long_read(const uint64_t* offsets, uint64_t q0, uint64_t q1) {
    for (uint64_t q = q0; q < q1; ++q)
        if (offsets[q + 1] - offsets[q] >= 65536) return q;
    return q1;
}
static int check_narrow_reads(const uint64_t* offsets, uint64_t q0, uint64_t q1) {
    const uint64_t q = first_long_read(offsets, q0, q1);
    if (q == q1) return SPX_OK;
    set_error("read %llu has 65536 characters or more: use the 32-bit entry point", (unsigned long long)q);
    return SPX_E_ARG;
}

static int run_pipelined(spx_index* ix, int mode, const uint8_t* seqs, const uint64_t* offsets, uint64_t nreads,
                         uint8_t* d_seq, uint64_t* d_off, uint64_t padded, void* out_lengths, uint64_t* out_pointers,
                         void* out_docs, spx_class* out_class, uint64_t bin_width, uint64_t max_value_thr,
                         size_t width) {
    const uint64_t total = offsets[nreads];
    void *dlen = nullptr, *dptr = nullptr, *ddoc = nullptr, *dcls = nullptr;
    int rc;
    if (out_lengths && (rc = ensure_scratch(ix, 2, (total + 1) * 4, &dlen)) != SPX_OK) return rc;
    if (out_pointers && (rc = ensure_scratch(ix, 3, (total + 1) * 8, &dptr)) != SPX_OK) return rc;
    if (out_docs && (rc = ensure_scratch(ix, 4, (total + 1) * 4, &ddoc)) != SPX_OK) return rc;
    if (out_class && (rc = ensure_scratch(ix, 5, (nreads + 1) * sizeof(spx_class), &dcls)) != SPX_OK) return rc;
    std::lock_guard<std::mutex> g(ix->mu);
    constexpr int NCH = spx_index::PIPE_CHUNKS;
    if (!ix->pipe_s[0]) {
        for (auto& st : ix->pipe_s) SPX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (int c = 0; c < NCH; ++c) {
            SPX_HIP(hipEventCreateWithFlags(&ix->pipe_in[c], hipEventDisableTiming));
            SPX_HIP(hipEventCreateWithFlags(&ix->pipe_k[c], hipEventDisableTiming));
        }
    }
    hipStream_t s_in = ix->pipe_s[0], s_k = ix->pipe_s[1], s_out = ix->pipe_s[2];
    // The pieces GROW.  The call is as long as its copy-out (the results are twice the bytes of the reads, and the
    // walk is faster than either copy) plus whatever passes before the first result can leave: so the first piece is
    // small (1/64 of the reads: copied in, walked and on its way out after ~0.4 ms instead of the ~3.7 ms an eighth of
    // the batch and the whole offsets array took), and every piece is 1.5 x the one before -- less than the 1.7 x by
    // which the walk outruns the copy-out, so the copy-out stream never waits for a walk.  SPX_PIPE_EVEN=1: equal pieces.
    uint64_t cut[NCH + 1];
    {
        static const bool even = getenv("SPX_PIPE_EVEN") != nullptr;
        double acc = 0, piece = 1.0 / 64.0;
        cut[0] = 0;
        for (int c = 0; c < NCH; ++c) {
            acc += even ? 1.0 / NCH : piece;
            piece *= 1.5;
            // (the series reaches the whole before the last piece: what is left then is one smaller piece)
            cut[c + 1] = (c + 1 == NCH || acc >= 1.0) ? nreads : (uint64_t)((double)nreads * acc);
        }
    }
    if (mode == SPX_MODE_PML && out_lengths) {
        // the length-bit scratch is sized for the largest piece BEFORE the pipeline starts: growing it between
        // pieces would hipFree (an implicit device synchronisation) in the middle of the copy / compute overlap
        uint64_t worst = 0;
        for (int c = 0; c < NCH; ++c) {
            const uint64_t q0 = cut[c], q1 = cut[c + 1];
            const uint64_t pairs = ((offsets[q1] - offsets[q0]) >> 7) + (q1 - q0) + 2;
            worst = pairs > worst ? pairs : worst;
        }
        void* unused = nullptr;
        if ((rc = chunk_scratch(ix, 8, worst * 16, &unused)) != SPX_OK) return rc;
    }
    if (ix->have_timing && ix->last_stream != s_k) SPX_HIP(hipStreamWaitEvent(s_k, ix->ev_done, 0));
    SPX_HIP(hipMemsetAsync(ix->counters, 0, sizeof(WalkCounters), s_k));
    SPX_HIP(hipEventRecord(ix->ev0, s_k));
    // SPX_PIPE_TRACE=1: when every piece's copy-in, walk and copy-out ended, on stderr (timed events of their own)
    static const bool trace = getenv("SPX_PIPE_TRACE") != nullptr;
    const auto h0 = std::chrono::steady_clock::now();
    hipEvent_t tr0 = nullptr, tr[NCH][3] = {};
    if (trace) {
        SPX_HIP(hipEventCreate(&tr0));
        for (auto& row : tr)
            for (auto& e : row) SPX_HIP(hipEventCreate(&e));
        SPX_HIP(hipEventRecord(tr0, s_in));
    }
    // 16-bit outputs: the reads' lengths are checked by a thread of its own while this one enqueues (10^7 offsets are
    // 3-4 ms of one core; in front of the pipeline that is 15 % of the call, and spread between the pieces' enqueues it
    // made the copy-out of the later pieces slow: profiles/r03_host_path_pipeline.txt)
    uint64_t long_read = nreads;
    std::thread narrow_check;
    if (width == 2) narrow_check = std::thread([&] { long_read = first_long_read(offsets, 0, nreads); });
    struct Joiner {
        std::thread& t;
        ~Joiner() {
            if (t.joinable()) t.join();
        }
    } joiner{narrow_check};
    // the read-ahead padding first, on the copy-in stream: every piece's event covers it; a piece's offsets travel
    // with the piece (the whole array up front is 80 MB for 10^7 reads: 1.4 ms before anything else could start)
    SPX_HIP(hipMemsetAsync(d_seq + total, 0, padded - total, s_in));
    for (int c = 0; c < NCH; ++c) {
        const uint64_t q0 = cut[c], q1 = cut[c + 1];
        if (q1 == q0) continue;
        const uint64_t a = offsets[q0], b = offsets[q1];
        SPX_HIP(hipMemcpyAsync(d_off + q0, offsets + q0, (q1 - q0 + 1) * 8, hipMemcpyHostToDevice, s_in));
        SPX_HIP(hipMemcpyAsync(d_seq + a, seqs + a, b - a, hipMemcpyHostToDevice, s_in));
        SPX_HIP(hipEventRecord(ix->pipe_in[c], s_in));
        if (trace) SPX_HIP(hipEventRecord(tr[c][0], s_in));
        SPX_HIP(hipStreamWaitEvent(s_k, ix->pipe_in[c], 0));
        BatchArgs args{};
        args.seqs = d_seq;
        args.offs = d_off + q0;
        args.nreads = q1 - q0;
        args.total_chars = b - a;
        args.out_lengths = (uint32_t*)dlen;
        args.out_pointers = (uint64_t*)dptr;
        args.out_docs = (uint32_t*)ddoc;
        args.out_class = (mode == SPX_MODE_PML && dcls) ? (spx_class*)dcls + q0 : nullptr;
        args.bin_width = bin_width;
        args.bin_magic = bin_width > 1 ? (uint64_t)(~0ull / bin_width) + 1 : 0;
        args.max_value_thr = max_value_thr;
        args.counters = ix->counters;
        args.narrow = width == 2 ? 1 : 0;
        if ((rc = prepare_len_mask(ix, mode, args)) != SPX_OK) return rc;
        {  // a chunk of few, long reads is cut further and walked chunk-wise (spx_walk.hip)
            bool chunked = false, wrote = false;
            if ((rc = launch_walk_chunked(ix, mode, args, b - a, s_k, &chunked)) != SPX_OK) return rc;
            if (!chunked && ((rc = launch_walk(ix, mode, args, b - a, s_k, &wrote)) != SPX_OK ||
                             (!wrote && (rc = launch_len_expand(ix, args, s_k)) != SPX_OK)))
                return rc;
        }
        if (mode == SPX_MODE_MS && dlen) {
            args.out_class = dcls ? (spx_class*)dcls + q0 : nullptr;
            if ((rc = launch_ms_extend(ix, args, s_k)) != SPX_OK) return rc;
        }
        SPX_HIP(hipEventRecord(ix->pipe_k[c], s_k));
        if (trace) SPX_HIP(hipEventRecord(tr[c][1], s_k));
        SPX_HIP(hipStreamWaitEvent(s_out, ix->pipe_k[c], 0));
        if (out_lengths)
            SPX_HIP(hipMemcpyAsync((char*)out_lengths + a * width, (char*)dlen + a * width, (b - a) * width,
                                   hipMemcpyDeviceToHost, s_out));
        if (out_pointers)
            SPX_HIP(hipMemcpyAsync(out_pointers + a, (uint64_t*)dptr + a, (b - a) * 8, hipMemcpyDeviceToHost, s_out));
        if (out_docs)
            SPX_HIP(hipMemcpyAsync((char*)out_docs + a * width, (char*)ddoc + a * width, (b - a) * width,
                                   hipMemcpyDeviceToHost, s_out));
        if (out_class)
            SPX_HIP(hipMemcpyAsync(out_class + q0, (spx_class*)dcls + q0, (q1 - q0) * sizeof(spx_class),
                                   hipMemcpyDeviceToHost, s_out));
        if (trace) SPX_HIP(hipEventRecord(tr[c][2], s_out));
    }
    SPX_HIP(hipEventRecord(ix->ev1, s_k));
    SPX_HIP(hipEventRecord(ix->ev_done, s_k));
    ix->have_timing = true;
    ix->last_stream = s_k;
    const double h_enq = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    SPX_HIP(hipStreamSynchronize(s_out));
    SPX_HIP(hipStreamSynchronize(s_k));
    if (narrow_check.joinable()) narrow_check.join();
    if (long_read != nreads) return check_narrow_reads(offsets, long_read, nreads);  // (the walk counted it as an error, too)
    if (trace) {
        std::fprintf(stderr, "spx pipeline: host: all pieces enqueued after %.2f ms, streams drained after %.2f ms\n", h_enq, 
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count());
        std::fprintf(stderr, "spx pipeline: piece reads | copy-in done, walk done, copy-out done (ms after the first copy was enqueued)\n");
        for (int c = 0; c < NCH; ++c) {
            float t[3] = {0, 0, 0};
            if (cut[c + 1] > cut[c])
                for (int j = 0; j < 3; ++j) (void)hipEventElapsedTime(&t[j], tr0, tr[c][j]);
            std::fprintf(stderr, "  %2d %9llu | %7.2f %7.2f %7.2f\n", c, (unsigned long long)(cut[c + 1] - cut[c]), t[0], t[1], t[2]);
            for (auto& e : tr[c]) (void)hipEventDestroy(e);
        }
        (void)hipEventDestroy(tr0);
    }
    WalkCounters wc;
    SPX_HIP(hipMemcpy(&wc, ix->counters, sizeof wc, hipMemcpyDeviceToHost));
    if (wc.error) {
        set_error("the walk hit %llu undefined steps (predecessor jump without a predecessor run: "
                  "thresholds are inconsistent with the BWT%s)", wc.error,
                  width == 2 ? "; or a read of 65536 characters or more with 16-bit outputs" : "");
        return SPX_E_FORMAT;
    }
    return SPX_OK;
}

static int query_host_impl(spx_index* ix, int mode, const uint8_t* seqs, const uint64_t* offsets, uint64_t nreads,
                           void* out_lengths, uint64_t* out_pointers, void* out_docs, spx_class* out_class,
                           uint64_t bin_width, uint64_t max_value_thr, size_t width) {
    int rc = check_query(ix, mode, seqs, offsets, (uint32_t*)out_lengths, out_pointers, (uint32_t*)out_docs,
                         out_class, bin_width);
    if (rc != SPX_OK) return rc;
    // (a pipelined batch checks its pieces as it enqueues them: 10^7 offsets are 3 ms of one core, spent beside the
    // device's work there instead of in front of it)
    const bool pipelined = nreads >= (1ull << 18) && nreads && offsets[nreads] >= (64u << 20);
    if (width == 2 && !pipelined && (rc = check_narrow_reads(offsets, 0, nreads)) != SPX_OK) return rc;
    std::lock_guard<std::mutex> hg(ix->host_mu);  // one host-buffer query at a time per index
    SPX_HIP(hipSetDevice(ix->device));
    const uint64_t total = nreads ? offsets[nreads] : 0;
    void *dseq = nullptr, *doff = nullptr;
    const uint64_t padded = ((total + 3) / 4) * 4 + 32;
    if ((rc = ensure_scratch(ix, 0, padded, &dseq)) != SPX_OK) return rc;
    if ((rc = ensure_scratch(ix, 1, (nreads + 1) * 8, &doff)) != SPX_OK) return rc;
    if (pipelined) {
        QuietOnError quiet(nullptr, true);
        return quiet.done(run_pipelined(ix, mode, seqs, offsets, nreads, (uint8_t*)dseq, (uint64_t*)doff, padded, out_lengths,
                                        out_pointers, out_docs, out_class, bin_width, max_value_thr, width));
    }
    hipStream_t st = nullptr;
    if ((rc = ctx_stream_of(ix, &st)) != SPX_OK) return rc;
    QuietOnError quiet(st);
    SPX_HIP(hipMemcpyAsync(doff, offsets, (nreads + 1) * 8, hipMemcpyHostToDevice, st));
    SPX_HIP(hipMemsetAsync((char*)dseq + total, 0, padded - total, st));
    SPX_HIP(hipMemcpyAsync(dseq, seqs, total, hipMemcpyHostToDevice, st));
    return quiet.done(run_and_fetch(ix, mode, (const uint8_t*)dseq, (const uint64_t*)doff, nreads, total, out_lengths,
                                    out_pointers, out_docs, out_class, bin_width, max_value_thr, width));
}

int spx_query_batch(spx_index* ix, int mode, const uint8_t* seqs, const uint64_t* offsets,
                    uint64_t nreads, uint32_t* out_lengths, uint64_t* out_pointers,
                    uint32_t* out_docs, spx_class* out_class, uint64_t bin_width,
                    uint64_t max_value_thr) {
    return query_host_impl(ix, mode, seqs, offsets, nreads, out_lengths, out_pointers, out_docs, out_class,
                           bin_width, max_value_thr, 4);
}

int spx_query_batch16(spx_index* ix, int mode, const uint8_t* seqs, const uint64_t* offsets,
                      uint64_t nreads, uint16_t* out_lengths, uint64_t* out_pointers,
                      uint16_t* out_docs, spx_class* out_class, uint64_t bin_width,
                      uint64_t max_value_thr) {
    return query_host_impl(ix, mode, seqs, offsets, nreads, out_lengths, out_pointers, out_docs, out_class,
                           bin_width, max_value_thr, 2);
}

int spx_digest_query_batch(spx_index* ix, int mode, int kind, uint32_t k, uint32_t w, const uint8_t* seqs,
                           const uint64_t* offsets, uint64_t nreads, uint64_t* out_offsets, uint64_t out_capacity,
                           uint32_t* out_lengths, uint64_t* out_pointers, uint32_t* out_docs, spx_class* out_class,
                           uint64_t bin_width, uint64_t max_value_thr) {
    int rc = check_query(ix, mode, seqs, offsets, out_lengths, out_pointers, out_docs, out_class, bin_width);
    if (rc != SPX_OK) return rc;
    if (!out_offsets) {
        set_error("out_offsets must be non-null");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> hg(ix->host_mu);
    SPX_HIP(hipSetDevice(ix->device));
    const uint64_t total = nreads ? offsets[nreads] : 0;
    const uint64_t cap = spx_digest_capacity(kind, k, total);
    void *draw = nullptr, *doff = nullptr, *dseq = nullptr, *dooff = nullptr;
    if ((rc = ensure_scratch(ix, 6, ((total + 15) & ~15ull) + 16, &draw)) != SPX_OK) return rc;
    if ((rc = ensure_scratch(ix, 1, (nreads + 1) * 8, &doff)) != SPX_OK) return rc;
    if ((rc = ensure_scratch(ix, 0, cap, &dseq)) != SPX_OK) return rc;
    if ((rc = ensure_scratch(ix, 7, (nreads + 1) * 8, &dooff)) != SPX_OK) return rc;
    hipStream_t st = nullptr;
    if ((rc = ctx_stream_of(ix, &st)) != SPX_OK) return rc;
    QuietOnError quiet(st);
    SPX_HIP(hipMemcpyAsync(draw, seqs, total, hipMemcpyHostToDevice, st));
    SPX_HIP(hipMemcpyAsync(doff, offsets, (nreads + 1) * 8, hipMemcpyHostToDevice, st));
    const uint64_t* in_starts = nullptr;
    rc = digest_for_walk(ix, mode, out_lengths != nullptr, kind, k, w, (const uint8_t*)draw, (const uint64_t*)doff, nreads, total,
                         (uint8_t*)dseq, cap, (uint64_t*)dooff, st, &in_starts);
    if (rc != SPX_OK) return rc;
    SPX_HIP(hipMemcpyAsync(out_offsets, dooff, (nreads + 1) * 8, hipMemcpyDeviceToHost, st));
    SPX_HIP(hipStreamSynchronize(st));
    const uint64_t dtotal = out_offsets[nreads];
    if (dtotal > out_capacity) {
        set_error("output buffers hold %llu entries, the digested reads have %llu characters",
                  (unsigned long long)out_capacity, (unsigned long long)dtotal);
        return SPX_E_ARG;
    }
    // the digested reads never leave the device: the walk starts from the scratch buffers
    return quiet.done(run_and_fetch(ix, mode, (const uint8_t*)dseq, (const uint64_t*)dooff, nreads, dtotal, out_lengths,
                                    out_pointers, out_docs, out_class, bin_width, max_value_thr, 4, in_starts));
}

// The same with everything resident in HBM and asynchronous on `stream`: DNA reads in, results at the digested reads'
// offsets (d_out_offsets) out.  What `run -m / -a` does to a read before matching_statistics (compute_ms_pml.cpp:919-923)
// and the query itself, one call.
static int digest_query_device_impl(spx_index* ix, int mode, int kind, uint32_t k, uint32_t w, const uint8_t* d_seqs,
                                    const uint64_t* d_offsets, uint64_t nreads, uint64_t total_chars, uint8_t* d_digested,
                                    uint64_t digested_capacity, uint64_t* d_out_offsets, uint32_t* d_out_lengths,
                                    uint64_t* d_out_pointers, uint32_t* d_out_docs, spx_class* d_out_class, uint64_t bin_width,
                                    uint64_t max_value_thr, void* stream, bool narrow) {
    const uint64_t* in_starts = nullptr;
    // everything the walk would refuse is refused BEFORE the digestion is enqueued (a misaligned d_digested
    // used to take the digestion's dword stores first and the error afterwards)
    int rc = check_query(ix, mode, d_seqs, d_offsets, d_out_lengths, d_out_pointers, d_out_docs, d_out_class, bin_width);
    if (rc != SPX_OK) return rc;
    if ((((uintptr_t)d_digested | (uintptr_t)d_out_lengths | (uintptr_t)d_out_pointers | (uintptr_t)d_out_docs) & 15) != 0) {
        set_error("d_digested and the output buffers must be 16-byte aligned");
        return SPX_E_ARG;
    }
    rc = digest_for_walk(ix, mode, d_out_lengths != nullptr, kind, k, w, d_seqs, d_offsets, nreads, total_chars, d_digested,
                         digested_capacity, d_out_offsets, stream, &in_starts);
    if (rc != SPX_OK) return rc;
    // total_chars bounds the digested characters and sizes the walk's scratch; the chunked walk's geometry (is the batch
    // long reads at all, how long is a chunk) goes by what a digestion leaves: about two minimizers per window of
    // w - k + 1 k-mers, k letters each with -a
    const uint64_t per = (kind == SPX_DIGEST_DNA ? (uint64_t)k : 1ull) * 2;
    const uint64_t est = std::min<uint64_t>(total_chars, total_chars * per / (uint64_t)(w - k + 2) + nreads);
    return query_device_impl(ix, mode, d_digested, d_out_offsets, nreads, total_chars, d_out_lengths, d_out_pointers, d_out_docs,
                             d_out_class, bin_width, max_value_thr, stream, narrow, in_starts, est);
}

int spx_digest_query_batch_device(spx_index* ix, int mode, int kind, uint32_t k, uint32_t w, const uint8_t* d_seqs,
                                  const uint64_t* d_offsets, uint64_t nreads, uint64_t total_chars, uint8_t* d_digested,
                                  uint64_t digested_capacity, uint64_t* d_out_offsets, uint32_t* d_out_lengths,
                                  uint64_t* d_out_pointers, uint32_t* d_out_docs, spx_class* d_out_class, uint64_t bin_width,
                                  uint64_t max_value_thr, void* stream) {
    return digest_query_device_impl(ix, mode, kind, k, w, d_seqs, d_offsets, nreads, total_chars, d_digested, digested_capacity,
                                    d_out_offsets, d_out_lengths, d_out_pointers, d_out_docs, d_out_class, bin_width, max_value_thr,
                                    stream, false);
}

int spx_digest_query_batch_device16(spx_index* ix, int mode, int kind, uint32_t k, uint32_t w, const uint8_t* d_seqs,
                                    const uint64_t* d_offsets, uint64_t nreads, uint64_t total_chars, uint8_t* d_digested,
                                    uint64_t digested_capacity, uint64_t* d_out_offsets, uint16_t* d_out_lengths,
                                    uint64_t* d_out_pointers, uint16_t* d_out_docs, spx_class* d_out_class, uint64_t bin_width,
                                    uint64_t max_value_thr, void* stream) {
    return digest_query_device_impl(ix, mode, kind, k, w, d_seqs, d_offsets, nreads, total_chars, d_digested, digested_capacity,
                                    d_out_offsets, (uint32_t*)d_out_lengths, d_out_pointers, (uint32_t*)d_out_docs, d_out_class,
                                    bin_width, max_value_thr, stream, true);
}

// ---------------------------------------------------------------------------------------------
// The vectors as the text the reference writes (compute_ms_pml.cpp:1001-1010, 1182-1205), produced on the device
// (spx_text.hip): spx_query_text_begin runs [digestion +] the walk [+ the MS extension] and formats; the sizes of
// the streams come back, the caller sizes its (page-locked) buffers, spx_query_text_fetch copies the text.
// ---------------------------------------------------------------------------------------------
int spx_query_text_begin(spx_index* ix, int mode, int digest_kind, uint32_t k, uint32_t w, const uint8_t* seqs,
                         const uint64_t* offsets, uint64_t nreads, const uint32_t* gap, uint32_t streams,
                         spx_class* out_class, uint64_t bin_width, uint64_t max_value_thr, uint64_t out_bytes[3]) {
    if (!ix || !seqs || !offsets || !out_bytes) {
        set_error("index, seqs, offsets and out_bytes must be non-null");
        return SPX_E_ARG;
    }
    const bool want_len = streams & SPX_TEXT_LENGTHS, want_ptr = streams & SPX_TEXT_POINTERS, want_doc = streams & SPX_TEXT_DOCS;
    // (check_query looks at which outputs are asked for, not at the pointers' targets)
    int rc = check_query(ix, mode, seqs, offsets, (want_len || (mode == SPX_MODE_MS && out_class)) ? (uint32_t*)1 : nullptr, (mode == SPX_MODE_MS) ? (uint64_t*)1 : nullptr,
                         want_doc ? (uint32_t*)1 : nullptr, out_class, bin_width);
    if (rc != SPX_OK) return rc;
    if (want_ptr && mode != SPX_MODE_MS) {
        set_error("the pointers stream is only produced in MS mode");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> hg(ix->host_mu);
    ix->text_ready = false;
    SPX_HIP(hipSetDevice(ix->device));
    static const bool timing = getenv("SPX_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now();
    hipStream_t st = nullptr;
    if ((rc = ctx_stream_of(ix, &st)) != SPX_OK) return rc;
    auto lap = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const double t = now();
        fprintf(stderr, "[spx] text_begin: %-28s %.2f ms\n", what, (t - t_mark) * 1e3);
        t_mark = t;
    };
    // SPX_PHASE_TRACE=1: the device's own times of a call's phases (events on the handle's stream, read after the call's one
    // synchronisation: nothing is added to the stream's work) -- how long the copy in / the kernels / the copy out of one
    // query context take while another context of the same device is at work (profiles/r05_cli_overlap.txt)
    static const bool phase_trace = getenv("SPX_PHASE_TRACE") != nullptr;
    static thread_local hipEvent_t pe[4] = {nullptr, nullptr, nullptr, nullptr};
    auto mark = [&](int i) {
        if (!phase_trace) return;
        if (!pe[i]) (void)hipEventCreate(&pe[i]);
        (void)hipEventRecord(pe[i], st);
    };
    mark(0);
    const uint64_t total_in = nreads ? offsets[nreads] : 0;
    void *dseq = nullptr, *doff = nullptr, *dgap = nullptr;
    const uint64_t padded = ((total_in + 3) / 4) * 4 + 32;
    if ((rc = ensure_scratch(ix, digest_kind ? 6 : 0, padded, &dseq)) != SPX_OK) return rc;
    if ((rc = ensure_scratch(ix, 1, (nreads + 1) * 8, &doff)) != SPX_OK) return rc;
    if ((rc = ensure_scratch(ix, 8, (nreads + 1) * 4, &dgap)) != SPX_OK) return rc;
    QuietOnError quiet(st);  // (a failed call leaves nothing reading seqs / offsets / gap)
    SPX_HIP(hipMemcpyAsync(dseq, seqs, total_in, hipMemcpyHostToDevice, st));
    SPX_HIP(hipMemsetAsync((char*)dseq + total_in, 0, padded - total_in, st));
    SPX_HIP(hipMemcpyAsync(doff, offsets, (nreads + 1) * 8, hipMemcpyHostToDevice, st));
    if (gap) SPX_HIP(hipMemcpyAsync(dgap, gap, nreads * 4, hipMemcpyHostToDevice, st));
    mark(1);
    lap("copy in");
    const uint8_t* wseq = (const uint8_t*)dseq;
    const uint64_t* woff = (const uint64_t*)doff;
    uint64_t total = total_in;
    bool narrow = true;
    for (uint64_t q = 0; q < nreads && narrow; ++q) narrow = offsets[q + 1] - offsets[q] < 65536;
    const uint64_t* in_starts = nullptr;  // set when the digested reads stay where the digestion parked them
    if (digest_kind) {
        // perform_minimizer_digestion / perform_dna_minimizer_digestion (compute_ms_pml.cpp:919-923): the digested
        // reads never leave the device; the vectors are laid out at the digested offsets
        const uint64_t cap = spx_digest_capacity(digest_kind, k, total_in);
        void *dd = nullptr, *ddo = nullptr;
        if ((rc = ensure_scratch(ix, 0, cap, &dd)) != SPX_OK) return rc;
        if ((rc = ensure_scratch(ix, 7, (nreads + 1) * 8, &ddo)) != SPX_OK) return rc;
        rc = digest_for_walk(ix, mode, want_len || out_class != nullptr, digest_kind, k, w, (const uint8_t*)dseq, (const uint64_t*)doff,
                             nreads, total_in, (uint8_t*)dd, cap, (uint64_t*)ddo, st, &in_starts);
        if (rc != SPX_OK) return rc;
        {
            PubSrc ps{{(const uint64_t*)ddo + nreads, nullptr, nullptr, nullptr}, nullptr, 0};
            if ((rc = publish(ix, ps, st)) != SPX_OK) return rc;
            if ((rc = ctx_wait(ix, st)) != SPX_OK) return rc;
            total = ix->h_pub[0];
        }
        wseq = (const uint8_t*)dd;
        woff = (const uint64_t*)ddo;
    }
    void *dlen = nullptr, *dptr = nullptr, *ddoc = nullptr, *dcls = nullptr;
    const bool need_len = want_len || (mode == SPX_MODE_MS && out_class);
    if (need_len && (rc = ensure_scratch(ix, 2, (total + 8) * 4, &dlen)) != SPX_OK) return rc;
    if (mode == SPX_MODE_MS && (rc = ensure_scratch(ix, 3, (total + 1) * 8, &dptr)) != SPX_OK) return rc;
    if (want_doc && (rc = ensure_scratch(ix, 4, (total + 8) * 4, &ddoc)) != SPX_OK) return rc;
    if (out_class && (rc = ensure_scratch(ix, 5, (nreads + 1) * sizeof(spx_class), &dcls)) != SPX_OK) return rc;
    rc = query_device_impl(ix, mode, wseq, woff, nreads, total, (uint32_t*)dlen, (uint64_t*)dptr, (uint32_t*)ddoc,
                           (spx_class*)dcls, bin_width, max_value_thr, st, narrow, in_starts);
    if (rc != SPX_OK) return rc;
    mark(2);
    lap("[digest +] walk");
    // count + scan per stream, then ONE read-back of the three sizes
    const size_t cub = text_scan_bytes(nreads);
    void* dcub = nullptr;
    if ((rc = ensure_scratch(ix, 9, cub + 256, &dcub)) != SPX_OK) return rc;
    const void* vals[3] = {want_len ? dlen : nullptr, want_ptr ? dptr : nullptr, want_doc ? ddoc : nullptr};
    const int vbytes[3] = {narrow ? 2 : 4, 8, narrow ? 2 : 4};
    void *lb[3] = {nullptr, nullptr, nullptr}, *ls[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < 3; ++i) {
        out_bytes[i] = 0;
        ix->text_bytes[i] = 0;
        if (!vals[i]) continue;
        if ((rc = ensure_scratch(ix, 10 + i, (nreads + 2) * 8, &lb[i])) != SPX_OK) return rc;
        if ((rc = ensure_scratch(ix, 13 + i, (nreads + 2) * 8, &ls[i])) != SPX_OK) return rc;
        if ((rc = launch_text_count(vals[i], vbytes[i], woff, gap ? (const uint32_t*)dgap : nullptr, nreads, (uint64_t*)lb[i],
                                    (uint64_t*)ls[i], dcub, cub, st)) != SPX_OK)
            return rc;
    }
    // the three sizes and the walk's counters: published, not copied (h_pub); the class records travel with the text
    WalkCounters wc;
    {
        PubSrc ps{{nullptr, nullptr, nullptr, nullptr}, (const uint64_t*)ix->counters, (int)(sizeof wc / 8)};
        for (int i = 0; i < 3; ++i)
            if (vals[i]) ps.one[i] = (const uint64_t*)ls[i] + nreads;
        if ((rc = publish(ix, ps, st)) != SPX_OK) return rc;
        mark(3);
        if ((rc = ctx_wait(ix, st)) != SPX_OK) return rc;
        if (phase_trace) {
            float a = 0, b = 0, c = 0;
            (void)hipEventElapsedTime(&a, pe[0], pe[1]);
            (void)hipEventElapsedTime(&b, pe[1], pe[2]);
            (void)hipEventElapsedTime(&c, pe[2], pe[3]);
            fprintf(stderr, "[phases] %p begin: copy in %.3f  walk %.3f  sizes %.3f ms\n", (void*)ix, a, b, c);
        }
        for (int i = 0; i < 3; ++i) out_bytes[i] = vals[i] ? ix->h_pub[i] : 0;
        std::memcpy(&wc, ix->h_pub + 4, sizeof wc);
    }
    ix->text_cls_host = out_class;
    lap("count + scan");
    for (int i = 0; i < 3; ++i) {
        if (!vals[i]) continue;
        void* dtext = nullptr;
        if ((rc = ensure_scratch(ix, 16 + i, out_bytes[i] + 64, &dtext)) != SPX_OK) return rc;
        if ((rc = launch_text_write(vals[i], vbytes[i], woff, gap ? (const uint32_t*)dgap : nullptr, nreads,
                                    (const uint64_t*)ls[i], (char*)dtext, st)) != SPX_OK)
            return rc;
        ix->text_bytes[i] = out_bytes[i];
    }
    // (not waited for: spx_query_text_fetch copies on the same stream, behind the digits)
    lap("digits");
    if (wc.error) {
        set_error("the walk hit %llu undefined steps (predecessor jump without a predecessor run: "
                  "thresholds are inconsistent with the BWT)", wc.error);
        return SPX_E_FORMAT;
    }
    ix->text_nreads = nreads;
    ix->text_ready = true;
    return quiet.done(SPX_OK);
}

int spx_query_text_fetch(spx_index* ix, char* text[3], uint64_t* line_start[3]) {
    if (!ix || !text) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> hg(ix->host_mu);
    if (!ix->text_ready) {
        set_error("spx_query_text_fetch without a successful spx_query_text_begin");
        return SPX_E_ARG;
    }
    SPX_HIP(hipSetDevice(ix->device));
    hipStream_t st = nullptr;
    {
        const int rc = ctx_stream_of(ix, &st);
        if (rc != SPX_OK) return rc;
    }
    static const bool timing = getenv("SPX_TIMING") != nullptr;
    static const bool phase_trace = getenv("SPX_PHASE_TRACE") != nullptr;
    static thread_local hipEvent_t fe[2] = {nullptr, nullptr};
    if (phase_trace) {
        if (!fe[0]) (void)hipEventCreate(&fe[0]), (void)hipEventCreate(&fe[1]);
        (void)hipEventRecord(fe[0], st);
    }
    const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    for (int i = 0; i < 3; ++i) {
        if (ix->text_bytes[i] == 0) continue;
        if (text[i]) SPX_HIP(hipMemcpyAsync(text[i], ix->scratch[16 + i].p, ix->text_bytes[i], hipMemcpyDeviceToHost, st));
        if (line_start && line_start[i])
            SPX_HIP(hipMemcpyAsync(line_start[i], ix->scratch[13 + i].p, (ix->text_nreads + 1) * 8, hipMemcpyDeviceToHost, st));
    }
    if (ix->text_cls_host)
        SPX_HIP(hipMemcpyAsync(ix->text_cls_host, ix->scratch[5].p, ix->text_nreads * sizeof(spx_class), hipMemcpyDeviceToHost, st));
    if (phase_trace) (void)hipEventRecord(fe[1], st);
    {
        const int rc = ctx_wait(ix, st);
        if (rc != SPX_OK) return rc;
    }
    if (phase_trace) {
        float a = 0;
        (void)hipEventElapsedTime(&a, fe[0], fe[1]);
        fprintf(stderr, "[phases] %p fetch: copy out %.3f ms (behind the digits' kernels)\n", (void*)ix, a);
    }
    ix->text_cls_host = nullptr;
    ix->text_ready = false;
    if (timing)
        fprintf(stderr, "[spx] text_fetch: %.2f ms for %.1f MB\n",
                (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0) * 1e3,
                (ix->text_bytes[0] + ix->text_bytes[1] + ix->text_bytes[2]) / 1e6);
    return SPX_OK;
}

// The scratch a spx_query_text_begin / _fetch pair of up to max_chars characters in max_reads reads will ask for, allocated now:
// the first super-batch of a run otherwise pays for it (~300 MB of hipMalloc at ~30 ms / GB: 10 ms of a 1.5 ms call, and the
// workers of one device queue behind each other for it), and a scratch buffer that has to GROW mid-run is freed first -- a
// device-wide synchronisation under every other worker's feet.  A hint: whatever turns out larger still grows on demand.
int spx_query_text_reserve(spx_index* ix, int mode, int digest_kind, uint32_t k, uint64_t max_chars, uint64_t max_reads,
                           uint32_t streams, int with_class, const uint64_t text_bytes[3]) {
    if (!ix || (mode != SPX_MODE_PML && mode != SPX_MODE_MS)) {
        set_error("spx_query_text_reserve: index and a mode");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> hg(ix->host_mu);  // (the order every host-buffer query takes them in: host_mu, then mu)
    std::lock_guard<std::mutex> g(ix->mu);
    SPX_HIP(hipSetDevice(ix->device));
    hipStream_t st = nullptr;
    int rc;
    if ((rc = ctx_stream_of(ix, &st)) != SPX_OK) return rc;
    const bool want_len = streams & SPX_TEXT_LENGTHS, want_ptr = streams & SPX_TEXT_POINTERS, want_doc = streams & SPX_TEXT_DOCS;
    void* p = nullptr;
    const uint64_t padded = ((max_chars + 3) / 4) * 4 + 32;
    if ((rc = ensure_scratch(ix, digest_kind ? 6 : 0, padded, &p)) != SPX_OK) return rc;
    if ((rc = ensure_scratch(ix, 1, (max_reads + 1) * 8, &p)) != SPX_OK) return rc;
    if ((rc = ensure_scratch(ix, 8, (max_reads + 1) * 4, &p)) != SPX_OK) return rc;
    uint64_t total = max_chars;
    if (digest_kind) {
        if ((rc = ensure_scratch(ix, 0, spx_digest_capacity(digest_kind, k, max_chars), &p)) != SPX_OK) return rc;
        if ((rc = ensure_scratch(ix, 7, (max_reads + 1) * 8, &p)) != SPX_OK) return rc;
        // (the vectors are sized from the digested total, known only per batch: what an undigested batch would need is the bound)
    }
    const bool need_len = want_len || (mode == SPX_MODE_MS && with_class);
    if (need_len && (rc = ensure_scratch(ix, 2, (total + 8) * 4, &p)) != SPX_OK) return rc;
    if (mode == SPX_MODE_MS && (rc = ensure_scratch(ix, 3, (total + 1) * 8, &p)) != SPX_OK) return rc;
    if (want_doc && (rc = ensure_scratch(ix, 4, (total + 8) * 4, &p)) != SPX_OK) return rc;
    if (with_class && (rc = ensure_scratch(ix, 5, (max_reads + 1) * sizeof(spx_class), &p)) != SPX_OK) return rc;
    if ((rc = ensure_scratch(ix, 9, text_scan_bytes(max_reads) + 256, &p)) != SPX_OK) return rc;
    const bool on[3] = {want_len, want_ptr && mode == SPX_MODE_MS, want_doc};
    for (int i = 0; i < 3; ++i) {
        if (!on[i]) continue;
        if ((rc = ensure_scratch(ix, 10 + i, (max_reads + 2) * 8, &p)) != SPX_OK) return rc;
        if ((rc = ensure_scratch(ix, 13 + i, (max_reads + 2) * 8, &p)) != SPX_OK) return rc;
        if (text_bytes && text_bytes[i] && (rc = ensure_scratch(ix, 16 + i, text_bytes[i] + 64, &p)) != SPX_OK) return rc;
    }
    if (mode == SPX_MODE_PML && need_len) {  // the walk's reset bits (prepare_len_mask)
        const uint64_t pairs = (total >> 7) + max_reads + 2;
        if ((rc = chunk_scratch(ix, 8, pairs * 16, &p)) != SPX_OK) return rc;
    }
    if (!ix->h_pub) {
        PubSrc none{{nullptr, nullptr, nullptr, nullptr}, nullptr, 0};
        if ((rc = publish(ix, none, st)) != SPX_OK) return rc;
    }
    SPX_HIP(hipStreamSynchronize(st));
    return SPX_OK;
}

// ---------------------------------------------------------------------------------------------
// flat-layout cache (.spx) and replication: the device arrays of an index as they are
// ---------------------------------------------------------------------------------------------
namespace {

struct SpxFileHeader {
    char magic[8];        // "SPXFLAT\0"
    char layout[56];      // spx_version(): a cache written by another layout is refused
    uint64_t header_bytes;
    uint64_t n, r;
    uint32_t has_samples, has_docs;
    uint64_t n_text;
    uint64_t arr_bytes[spx_index::NARR];
    uint64_t arr_offset[spx_index::NARR];  // file offsets, 4096-aligned
    spx::DevIndex view;   // scalars; the pointers inside are rebound on load
    uint64_t device_bytes;
    char source_tag[128]; // spx_index_set_source_tag(): what the index was built from, as the caller names it
};

constexpr size_t STAGE = 16u << 20;

// file -> device through two page-locked staging buffers (read of chunk i+1 overlaps copy of chunk i)
int read_to_device(FILE* f, uint64_t off, void* dst, uint64_t bytes, void* stage[2], hipStream_t st, hipEvent_t ev[2]) {
    if (fseeko(f, (off_t)off, SEEK_SET) != 0) {
        set_error("seek failed");
        return SPX_E_IO;
    }
    int b = 0;
    for (uint64_t done = 0; done < bytes; b ^= 1) {
        const size_t take = (size_t)std::min<uint64_t>(STAGE, bytes - done);
        SPX_HIP(hipEventSynchronize(ev[b]));  // the copy that last used this buffer
        if (fread(stage[b], 1, take, f) != take) {
            set_error("cache file is truncated");
            return SPX_E_IO;
        }
        SPX_HIP(hipMemcpyAsync((char*)dst + done, stage[b], take, hipMemcpyHostToDevice, st));
        SPX_HIP(hipEventRecord(ev[b], st));
        done += take;
    }
    return SPX_OK;
}

int write_from_device(FILE* f, const void* src, uint64_t bytes, void* stage[2], hipStream_t st, hipEvent_t ev[2]) {
    // device -> host copy of chunk i+1 overlaps the fwrite of chunk i
    uint64_t issued = 0, written = 0;
    size_t len[2] = {0, 0};
    int b = 0;
    auto issue = [&](int buf) -> int {
        len[buf] = (size_t)std::min<uint64_t>(STAGE, bytes - issued);
        SPX_HIP(hipMemcpyAsync(stage[buf], (const char*)src + issued, len[buf], hipMemcpyDeviceToHost, st));
        SPX_HIP(hipEventRecord(ev[buf], st));
        issued += len[buf];
        return SPX_OK;
    };
    if (bytes == 0) return SPX_OK;
    int rc = issue(0);
    if (rc != SPX_OK) return rc;
    while (written < bytes) {
        if (issued < bytes && (rc = issue(b ^ 1)) != SPX_OK) return rc;
        SPX_HIP(hipEventSynchronize(ev[b]));
        if (fwrite(stage[b], 1, len[b], f) != len[b]) {
            set_error("write failed (disk full?)");
            return SPX_E_IO;
        }
        written += len[b];
        b ^= 1;
    }
    return SPX_OK;
}

struct Staging {  // two pinned buffers + a stream + two events, released on scope exit
    void* stage[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int init() {
        for (int i = 0; i < 2; ++i) {
            SPX_HIP(hipHostMalloc(&stage[i], STAGE, hipHostMallocDefault));
            SPX_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        SPX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        return SPX_OK;
    }
    ~Staging() {
        for (int i = 0; i < 2; ++i) {
            if (stage[i]) (void)hipHostFree(stage[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
        if (st) (void)hipStreamDestroy(st);
    }
};

// One array between a file and the device, in up to IO_THREADS slices: every slice has its own descriptor position,
// staging buffers and stream (a single reader does 6-7 GB/s from tmpfs, PCIe takes several times that).
constexpr int IO_THREADS = 4;
// Page-locked staging is expensive to allocate and more so to release (seconds for a few hundred MB): the pool
// (8 x 16 MB) is made once per process and kept; one save / load at a time uses it.
struct IoPool {
    Staging sg[IO_THREADS];
    bool ready = false;
    std::mutex mu;
    int init() {
        if (ready) return SPX_OK;
        for (auto& g : sg) {
            const int rc = g.init();
            if (rc != SPX_OK) return rc;
        }
        ready = true;
        return SPX_OK;
    }
};
static IoPool& io_pool() {
    static IoPool* p = new IoPool();  // never destroyed: the HIP runtime may be gone by the time statics are
    return *p;
}
int transfer_array(IoPool& pool, const std::string& path, uint64_t off, void* dev, uint64_t bytes, bool to_device,
                   int device) {
    if (bytes == 0) return SPX_OK;
    const int nt = bytes >= (256ull << 20) ? IO_THREADS : 1;
    std::vector<int> rc(nt, SPX_OK);
    std::vector<std::string> msg(nt);
    auto work = [&](int t) {
        const uint64_t lo = (bytes * t / nt) & ~4095ull, hi = t + 1 == nt ? bytes : (bytes * (t + 1) / nt) & ~4095ull;
        auto run = [&]() -> int {
            SPX_HIP(hipSetDevice(device));
            Staging& sg = pool.sg[t];
            int r = SPX_OK;
            FILE* f = fopen(path.c_str(), to_device ? "rb" : "r+b");
            if (!f) {
                set_error("cannot open %s", path.c_str());
                return SPX_E_IO;
            }
            if (to_device) {
                r = read_to_device(f, off + lo, (char*)dev + lo, hi - lo, sg.stage, sg.st, sg.ev);
                if (r == SPX_OK && hipStreamSynchronize(sg.st) != hipSuccess) r = SPX_E_HIP;
            } else {
                r = fseeko(f, (off_t)(off + lo), SEEK_SET) == 0 ? SPX_OK : SPX_E_IO;
                if (r == SPX_OK) r = write_from_device(f, (const char*)dev + lo, hi - lo, sg.stage, sg.st, sg.ev);
            }
            if (fclose(f) != 0 && r == SPX_OK && !to_device) {
                set_error("write failed (disk full?)");
                r = SPX_E_IO;
            }
            return r;
        };
        rc[t] = run();
        if (rc[t] != SPX_OK) msg[t] = spx_last_error();  // the error text is thread-local: carry it over
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (int t = 0; t < nt; ++t)
        if (rc[t] != SPX_OK) {
            set_error("%s", msg[t].c_str());
            return rc[t];
        }
    return SPX_OK;
}

}  // namespace

const char* spx_version(void) { return SPX_LAYOUT_VERSION; }

int spx_index_save(spx_index* ix, const char* path) {
    if (!ix || !path) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> g(ix->mu);
    SPX_HIP(hipSetDevice(ix->device));
    SPX_HIP(hipDeviceSynchronize());
    SpxFileHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "SPXFLAT", 8);
    snprintf(h.layout, sizeof h.layout, "%s", SPX_LAYOUT_VERSION);
    h.header_bytes = sizeof h;
    h.n = ix->n;
    h.r = ix->r;
    h.has_samples = ix->has_samples;
    h.has_docs = ix->has_docs;
    h.n_text = ix->n_text;
    h.view = ix->view;
    {  // the file holds no addresses: the pointers are rebound on load (bind_view)
        spx_index blank;
        blank.view = h.view;
        blank.n_text = ix->n_text;
        bind_view(&blank);
        h.view = blank.view;
    }
    h.device_bytes = ix->device_bytes;
    memcpy(h.source_tag, ix->source_tag, sizeof h.source_tag);
    // the fat table and fat_js are not written: spx_index_load_flat rebuilds them from the other arrays (build_fat)
    uint64_t off = (sizeof h + 4095) & ~4095ull;
    for (int i = 0; i < spx_index::NARR; ++i) {
        const bool skip = i == A_FAT || i == A_FATJ;
        h.arr_bytes[i] = ix->arr_bytes[i];
        h.arr_offset[i] = skip ? 0 : off;
        if (!skip) off = (off + ix->arr_bytes[i] + 4095) & ~4095ull;
    }
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) {
        set_error("cannot create %s", tmp.c_str());
        return SPX_E_IO;
    }
    IoPool& pool = io_pool();
    std::lock_guard<std::mutex> pg(pool.mu);
    int rc = pool.init();
    if (rc == SPX_OK && (fwrite(&h, sizeof h, 1, f) != 1 || ftruncate(fileno(f), (off_t)off) != 0)) {
        set_error("write failed");
        rc = SPX_E_IO;
    }
    if (fclose(f) != 0 && rc == SPX_OK) {
        set_error("write failed (disk full?)");
        rc = SPX_E_IO;
    }
    void** arr[spx_index::NARR];
    index_arrays(ix, arr);
    for (int i = 0; i < spx_index::NARR && rc == SPX_OK; ++i)
        if (h.arr_offset[i]) rc = transfer_array(pool, tmp, h.arr_offset[i], *arr[i], h.arr_bytes[i], false, ix->device);
    if (rc == SPX_OK && rename(tmp.c_str(), path) != 0) {
        set_error("cannot rename %s to %s", tmp.c_str(), path);
        rc = SPX_E_IO;
    }
    if (rc != SPX_OK) remove(tmp.c_str());
    return rc;
}

spx_index* spx_index_load_flat(const char* path, int device) {
    if (!path) {
        set_error("path is null");
        return nullptr;
    }
    if (select_device(device) != SPX_OK) return nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) {
        set_error("cannot open %s", path);
        return nullptr;
    }
    SpxFileHeader h;
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "SPXFLAT", 8) != 0 || h.header_bytes != sizeof h) {
        set_error("%s is not a flat-layout cache of this library", path);
        fclose(f);
        return nullptr;
    }
    h.layout[sizeof h.layout - 1] = 0;
    if (strcmp(h.layout, SPX_LAYOUT_VERSION) != 0) {
        set_error("%s was written by layout '%s', this library is '%s': rebuild the cache", path, h.layout,
                  SPX_LAYOUT_VERSION);
        fclose(f);
        return nullptr;
    }
    {   // the header's fields against each other and against the file: a damaged cache must not size device
        // arrays the kernels then run past
        struct stat stf;
        const uint64_t fsize = fstat(fileno(f), &stf) == 0 ? (uint64_t)stf.st_size : 0;
        const uint64_t r = h.view.r;  // runs of the flat layout (pieces of long runs count); h.r is the file's r
        const uint64_t row_bytes = h.view.compact ? sizeof(spx::Row32) : sizeof(spx::Row);
        const bool aux = h.has_samples || h.has_docs;
        uint64_t want[spx_index::NARR] = {};
        want[A_ROWS] = (r + ROW_PAD) * row_bytes;
        want[A_DIRROWS] = (r + ROW_PAD) * sizeof(spx::JumpRow);
        want[A_FAT] = (h.view.nfat + 2) * (uint64_t)h.view.fat_stride;
        want[A_FATJ] = spx::fatjs_count(h.view.nfat) * 4 + 64;
        want[A_Q] = (r + 1 + Q_PAD) * 4;
        want[A_AUX] = aux ? (r + 2) * sizeof(spx::Aux) : 0;
        want[A_SSRUN] = h.has_samples ? (r + 4) * 8 : 0;
        want[A_RUNDOCS] = h.has_docs ? (r + ROW_PAD) * 4 : 0;
        want[A_LETTERS] = 256 * sizeof(spx::LetterInfo);
        want[A_TEXT] = h.n_text ? h.n_text + 16 : 0;
        bool ok = r > 0 && r < 0xfffffff0ull && h.r > 0 && h.r <= r && h.n > 0 && h.view.n == h.n &&
                  (h.view.fat_stride == 32u || (!aux && h.view.fat_stride == 16u)) && (h.view.fat_stride == 16u || aux || h.view.compact) &&
                  (h.n_text == 0 || h.n_text + 1 == h.n || h.n_text < h.n);
        for (int i = 0; ok && i < spx_index::NARR; ++i) {
            ok = h.arr_bytes[i] == want[i];
            const bool stored = i != A_FAT && i != A_FATJ && h.arr_bytes[i] != 0;
            if (ok && stored) ok = h.arr_offset[i] >= sizeof h && h.arr_offset[i] + h.arr_bytes[i] <= fsize;
        }
        if (!ok) {
            set_error("%s: the header does not describe a consistent index (array sizes / offsets against r = %llu and the "
                      "file's %llu bytes): rebuild the cache", path, (unsigned long long)r, (unsigned long long)fsize);
            fclose(f);
            return nullptr;
        }
    }
    spx_index* ix = new spx_index();
    ix->device = device;
    h.source_tag[sizeof h.source_tag - 1] = 0;
    memcpy(ix->source_tag, h.source_tag, sizeof ix->source_tag);
    ix->n = h.n;
    ix->r = h.r;
    ix->has_samples = h.has_samples != 0;
    ix->has_docs = h.has_docs != 0;
    ix->n_text = h.n_text;
    ix->view = h.view;
    ix->device_bytes = h.device_bytes;
    const bool timing = getenv("SPX_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto body = [&]() -> int {
        double t0 = now();
        IoPool& pool = io_pool();
        std::lock_guard<std::mutex> pg(pool.mu);
        int rc = pool.init();
        if (rc != SPX_OK) return rc;
        if (timing) fprintf(stderr, "[spx] load_flat: staging pool %.3f s\n", now() - t0);
        void** arr[spx_index::NARR];
        index_arrays(ix, arr);
        for (int i = 0; i < spx_index::NARR; ++i) {
            ix->arr_bytes[i] = h.arr_bytes[i];
            if (h.arr_bytes[i] == 0 || h.arr_offset[i] == 0) continue;  // absent, or rebuilt below
            SPX_HIP(hipMalloc(arr[i], h.arr_bytes[i]));
            t0 = now();
            if ((rc = transfer_array(pool, path, h.arr_offset[i], *arr[i], h.arr_bytes[i], true, device)) != SPX_OK) return rc;
            if (timing) fprintf(stderr, "[spx] load_flat: array %d, %.2f GB in %.3f s\n", i, h.arr_bytes[i] / 1e9, now() - t0);
        }
        bind_view(ix);
        t0 = now();
        if ((rc = build_fat(ix)) != SPX_OK) return rc;
        if (timing) fprintf(stderr, "[spx] load_flat: fat table rebuilt in %.3f s\n", now() - t0);
        return init_runtime(ix);
    };
    const int rc = body();
    fclose(f);
    if (rc != SPX_OK) {
        spx_index_free(ix);
        return nullptr;
    }
    return ix;
}

spx_index* spx_index_clone(spx_index* src, int device) {
    if (!src) {
        set_error("index is null");
        return nullptr;
    }
    if (select_device(device) != SPX_OK) return nullptr;
    spx_index* ix = new spx_index();
    ix->device = device;
    auto body = [&]() -> int {
        std::lock_guard<std::mutex> g(src->mu);
        ix->n = src->n;
        ix->r = src->r;
        ix->has_samples = src->has_samples;
        ix->has_docs = src->has_docs;
        ix->n_text = src->n_text;
        ix->view = src->view;
        ix->device_bytes = src->device_bytes;
        if (device != src->device) {  // xGMI peer copies when the devices can reach each other
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device, src->device) == hipSuccess && can)
                (void)hipDeviceEnablePeerAccess(src->device, 0);
            (void)hipGetLastError();  // "already enabled" is fine
        }
        void** from[spx_index::NARR];
        void** to[spx_index::NARR];
        index_arrays(src, from);
        index_arrays(ix, to);
        if (device == src->device) {
            // the same device: a second query context over the same arrays (nothing is copied; the arrays are read-only
            // once built and go when the last handle is freed) -- what lets two host threads keep one device's copy engines
            // and compute units busy at the same time without a second 200 GB replica
            if (!src->owner) {
                src->owner = std::make_shared<ArrayOwner>();
                src->owner->device = src->device;
                for (int i = 0; i < spx_index::NARR; ++i) src->owner->p[i] = *from[i];
            }
            ix->owner = src->owner;
            for (int i = 0; i < spx_index::NARR; ++i) {
                ix->arr_bytes[i] = src->arr_bytes[i];
                *to[i] = *from[i];
            }
        } else {
            for (int i = 0; i < spx_index::NARR; ++i) {
                ix->arr_bytes[i] = src->arr_bytes[i];
                if (src->arr_bytes[i] == 0) continue;
                SPX_HIP(hipMalloc(to[i], src->arr_bytes[i]));
                SPX_HIP(hipMemcpyPeerAsync(*to[i], device, *from[i], src->device, src->arr_bytes[i], nullptr));
            }
            SPX_HIP(hipDeviceSynchronize());
        }
        bind_view(ix);
        const int rc = init_runtime(ix);
        memcpy(ix->charhash, src->charhash, sizeof ix->charhash);
        memcpy(ix->source_tag, src->source_tag, sizeof ix->source_tag);
        ix->waves_per_cu = src->waves_per_cu;
        ix->chunk_mode = src->chunk_mode;
        ix->chunk_shift = src->chunk_shift;
        ix->chunk_len = src->chunk_len;
        ix->force_lanes_per_wave = src->force_lanes_per_wave;
        ix->force_digest_kernel = src->force_digest_kernel;
        ix->digest_parked = src->digest_parked;
        return rc;
    };
    if (body() != SPX_OK) {
        spx_index_free(ix);
        return nullptr;
    }
    return ix;
}

int spx_index_describe(const spx_index* ix, char* buf, size_t cap) {
    if (!ix || !buf || cap == 0) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    const DevIndex& v = ix->view;
    // (SPX_DESCRIBE_ADDRESSES: also where the three big arrays lie -- tools/c5_regimes.py; not part of the description proper,
    // which is equal for an index and its copy)
    char where[160] = "";
    if (getenv("SPX_DESCRIBE_ADDRESSES"))
        snprintf(where, sizeof where, ", \"rows_at\": \"%p\", \"dirrows_at\": \"%p\", \"fat_at\": \"%p\"", (void*)ix->rows, (void*)ix->dirrows, (void*)ix->fat);
    snprintf(buf, cap,
             "{\"layout\": \"%s\", \"n\": %llu, \"r\": %llu, \"flat_runs\": %u, \"letters\": %u, \"compact_rows\": %u, "
             "\"fat_slots\": %llu, \"fat_slots_per_run\": %.4f, \"fat_stride\": %u, \"has_samples\": %d, "
             "\"has_docs\": %d, \"n_text\": %llu, \"device_bytes\": %llu%s}",
             SPX_LAYOUT_VERSION, (unsigned long long)ix->n, (unsigned long long)ix->r, v.r, v.nletters, v.compact,
             (unsigned long long)v.nfat, (double)v.nfat / (double)(v.r ? v.r : 1), v.fat_stride,
             (int)ix->has_samples, (int)ix->has_docs, (unsigned long long)ix->n_text,
             (unsigned long long)(ix->device_bytes + ix->n_text), where);
    return SPX_OK;
}

int spx_last_chunk_stats(spx_index* ix, uint64_t out[4]) {
    if (!ix || !out) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> g(ix->mu);
    if (!ix->have_timing) {
        set_error("no query has run on this index yet");
        return SPX_E_ARG;
    }
    SPX_HIP(hipSetDevice(ix->device));
    SPX_HIP(hipStreamSynchronize(ix->last_stream));
    WalkCounters wc;
    SPX_HIP(hipMemcpy(&wc, ix->counters, sizeof wc, hipMemcpyDeviceToHost));
    out[0] = ix->last_chunk_len;
    out[1] = ix->last_chunk_bound;
    out[2] = wc.reserved0;
    out[3] = wc.pad_;
    return SPX_OK;
}

int spx_last_walk_stats(spx_index* ix, spx_walk_stats* out) {
    if (!ix || !out) {
        set_error("null argument");
        return SPX_E_ARG;
    }
    std::lock_guard<std::mutex> g(ix->mu);
    if (!ix->have_timing) {
        set_error("no query has run on this index yet");
        return SPX_E_ARG;
    }
    SPX_HIP(hipSetDevice(ix->device));
    SPX_HIP(hipEventSynchronize(ix->ev1));
    SPX_HIP(hipStreamSynchronize(ix->last_stream));
    WalkCounters wc;
    SPX_HIP(hipMemcpy(&wc, ix->counters, sizeof wc, hipMemcpyDeviceToHost));
    float ms = 0;
    SPX_HIP(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
    out->steps = wc.steps;
    out->jumps = wc.jumps;
    out->pred_jumps = wc.pred_jumps;
    out->row_loads = wc.row_loads;
    out->dir_loads = wc.dir_loads;
    out->kernel_ms = ms;
    if (wc.error) {
        set_error("the walk hit %llu undefined steps (inconsistent thresholds; or a read of 65536 characters or more with "
                  "16-bit outputs; or a batch that holds more characters than total_chars)", wc.error);
        return SPX_E_FORMAT;
    }
    return SPX_OK;
}

}  // extern "C"
