// context.hip — context, error text, cached device allocator, operator handles, timers, checksums.
#include "common.h"
#include <thread>
#include <chrono>
#include <sys/mman.h>
#include <stdint.h>
#include <vector>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <vector>
#include <mutex>

namespace symgpu {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    set_error("HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return e == hipErrorOutOfMemory ? SYMGPU_E_NOMEM : SYMGPU_E_HIP;
}

// One context per DEVICE, all of them in this process (round 5: single-process multi-device mode, SURVEY 8b "one host process ... driving
// <= 8 devices").  A host thread has a CURRENT device — the one it last selected with symgpu_set_device / symgpu_init, else the
// first device initialised in the process — and everything that is not handed a handle works on it.  A call that IS handed a handle runs
// on the handle's device for its duration (DeviceScope).
static Context g_ctxs[SYMGPU_MAX_DEVICES];
static int g_default_dev = -1;                    // the first device initialised in this process
static thread_local int t_cur_dev = -1;           // this thread's selection (-1: the default)
static thread_local int t_bound_dev = -1;         // the device hipSetDevice was last called with on this thread
i64 g_counters[16] = {0};   // symgpu_debug_counter 1..10 (0 is g_hash_reseeds, cleanup.hip)
static int cur_index() { return t_cur_dev >= 0 ? t_cur_dev : (g_default_dev >= 0 ? g_default_dev : 0); }
Context &ctx() { return g_ctxs[cur_index()]; }
Context *ctx_of_device(int device) { return device >= 0 && device < SYMGPU_MAX_DEVICES ? &g_ctxs[device] : nullptr; }
void select_device(int device) { t_cur_dev = device; }
int selected_device() { return t_cur_dev; }

static std::mutex g_deg_mu;
static std::string g_degraded;
void note_degraded(const char *what) {
    std::lock_guard<std::mutex> lk(g_deg_mu);
    if (g_degraded.find(what) != std::string::npos) return;
    if (!g_degraded.empty()) g_degraded += "; ";
    g_degraded += what;
    fprintf(stderr, "symgpu: warning: %s (results are unaffected; symgpu_degraded() lists what was switched off)\n", what);
}

int require_ctx() {
    Context &c = ctx();
    if (!c.ready) {
        set_error("symgpu_init() has not been called for this device (or no HIP device)");
        return SYMGPU_E_NODEVICE;
    }
    // HIP's current device is per host thread and starts at 0: a thread other than the one that called symgpu_init (e.g. the
    // watchdog thread of symmer_amd/parallel.py), or one that has moved on to another context, must be bound to the context's device
    // before it touches the runtime
    if (t_bound_dev != c.device) {
        HIP_TRY(hipSetDevice(c.device));
        t_bound_dev = c.device;
    }
    return SYMGPU_OK;
}
void forget_bound_device() { t_bound_dev = -1; }



int DeviceScope::enter(const symgpu_op_s *a, const symgpu_op_s *b, const symgpu_op_s *c) {
    saved = t_cur_dev;
    active = true;
    const symgpu_op_s *first = a ? a : (b ? b : c);
    if (first) {
        if ((b && b->device != first->device) || (c && c->device != first->device)) {
            set_error("operands live on different devices (%d and %d): copy one over with symgpu_op_copy_rows first", first->device,
                      (b && b->device != first->device) ? b->device : c->device);
            return SYMGPU_E_INVALID;
        }
        t_cur_dev = first->device;
    }
    return require_ctx();
}
DeviceScope::~DeviceScope() { if (active) t_cur_dev = saved; }

// ---- cached allocator ------------------------------------------------------------------------------------------------------
// Size classes (power-of-two-ish), freed blocks parked per class until shutdown / release.  A class that has no parked block is
// carved from an ARENA — 4 GiB chunks, bump pointer, blocks up to 1 GiB — instead of going to hipMalloc (100 us .. 10 ms per call):
// a chain of rotations whose term count grows meets a new size class with every step, and with the arena its first pass costs what
// every later pass costs (round 2 needed a warm-up pass in the bench for that).  Carved blocks are never returned to the runtime
// one by one; a chunk is released as a whole when none of its blocks is in use (dev_cache_release).
static std::mutex g_alloc_mu;
struct LiveBlock { size_t cls; int chunk; int dev; size_t req = 0; };   // req: requested bytes (canary mode only)   // chunk: index into the device's chunks, -1 = its own hipMalloc
static std::map<void *, LiveBlock> g_live;        // block in use -> class / origin (device pointers are unique across the devices)
struct Chunk { char *base; size_t size, used; i64 live; };
struct DevAlloc {                                 // the allocator's state of ONE device
    std::multimap<size_t, void *> free_;          // size class -> parked block
    std::map<void *, int> parked_chunk;           // parked block -> origin (only arena blocks)
    std::vector<Chunk> chunks;
    size_t cached_bytes = 0;
    size_t cache_limit = (size_t)64 << 30;        // parked blocks: at most 64 GiB, raised to half of the device memory at init
                                                  // (hipMalloc / hipFree of multi-GB blocks cost ~10 ms per GB)
    bool arena_on = true;
};
static DevAlloc g_alloc[SYMGPU_MAX_DEVICES];
static const size_t ARENA_CHUNK = (size_t)4 << 30, ARENA_MAX_BLOCK = (size_t)1 << 30;
// the names the allocator below was written with, now the CURRENT device's state
#define g_free (g_alloc[cur_index()].free_)
#define g_parked_chunk (g_alloc[cur_index()].parked_chunk)
#define g_chunks (g_alloc[cur_index()].chunks)
#define g_cached_bytes (g_alloc[cur_index()].cached_bytes)
#define g_cache_limit (g_alloc[cur_index()].cache_limit)
#define g_arena_on (g_alloc[cur_index()].arena_on)

static size_t size_class(size_t b) {
    if (b < 256) b = 256;
    if (b <= ((size_t)1 << 20)) {           // <= 1 MiB: next power of two
        size_t c = 256;
        while (c < b) c <<= 1;
        return c;
    }
    size_t g = (size_t)1 << 20;             // above: 1 MiB granularity rounded to 1/8 of the leading power
    size_t p = g;
    while ((p << 1) <= b) p <<= 1;
    size_t step = p >> 3;
    if (step < g) step = g;
    return (b + step - 1) / step * step;
}

// carve `c` bytes (a multiple of 256) from the arena; nullptr if the arena is off, the block is too large or memory is short
static void *arena_carve(size_t c, int *chunk) {
    if (!g_arena_on || c > ARENA_MAX_BLOCK) return nullptr;
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = (int)g_chunks.size() - 1; i >= 0; --i) {
            Chunk &ch = g_chunks[i];
            if (ch.base && ch.size - ch.used >= c) {
                void *p = ch.base + ch.used;
                ch.used += c;
                ++ch.live;
                *chunk = i;
                return p;
            }
        }
        if (pass == 1) break;
        void *base = nullptr;
        ++g_counters[3];
        if (hipMalloc(&base, ARENA_CHUNK) != hipSuccess) { (void)hipGetLastError(); g_arena_on = false; return nullptr; }
        g_chunks.push_back(Chunk{static_cast<char *>(base), ARENA_CHUNK, 0, 0});
    }
    return nullptr;
}

// Debug aid (tuning build, SYMGPU_ALLOC_CANARY=1): every block gets 256 bytes of 0xA5 behind the bytes that were asked for, checked when
// the block is freed — a kernel that writes past the end of a buffer is reported on stderr with the block's size (size classes round up, so
// such a write otherwise lands in padding and goes unnoticed).  Reads past the end cannot be caught this way.
constexpr size_t CANARY = 256;
static bool canary_on() { static const bool on = [] { const char *e = SG_TUNE("SYMGPU_ALLOC_CANARY"); return e && e[0] == '1'; }(); return on; }
static void canary_set(void *p, size_t bytes) {
    (void)hipMemsetAsync(static_cast<char *>(p) + bytes, 0xA5, CANARY, ctx().stream);
}
static void canary_check(void *p, size_t bytes) {
    unsigned char h[CANARY];
    (void)hipDeviceSynchronize();
    if (hipMemcpy(h, static_cast<char *>(p) + bytes, CANARY, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
    for (size_t k = 0; k < CANARY; ++k)
        if (h[k] != 0xA5) {
            fprintf(stderr, "symgpu CANARY: block of %zu bytes overwritten at +%zu behind its end (value 0x%02x)\n", bytes, k, h[k]);
            ++g_counters[11];
            return;
        }
}
int dev_alloc(size_t bytes, void **ptr) {
    SG_TRY(require_ctx());
    const size_t req = bytes;
    if (canary_on()) bytes += CANARY;
    size_t c = size_class(bytes);
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        auto it = g_free.find(c);
        if (it != g_free.end()) {
            *ptr = it->second;
            g_free.erase(it);
            g_cached_bytes -= c;
            int chunk = -1;
            auto pc = g_parked_chunk.find(*ptr);
            if (pc != g_parked_chunk.end()) { chunk = pc->second; g_parked_chunk.erase(pc); ++g_chunks[chunk].live; }
            g_live[*ptr] = LiveBlock{c, chunk, cur_index(), req};
            if (canary_on()) canary_set(*ptr, req);
            return SYMGPU_OK;
        }
        int chunk = -1;
        if (void *p = arena_carve(c, &chunk)) {
            *ptr = p;
            g_live[p] = LiveBlock{c, chunk, cur_index(), req};
            if (canary_on()) canary_set(p, req);
            return SYMGPU_OK;
        }
    }
    hipError_t e = hipMalloc(ptr, c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dev_cache_release();
        e = hipMalloc(ptr, c);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            // Last resort: a parked block of a LARGER class.  dev_cache_release cannot return arena blocks whose chunk still holds
            // a live block (one long-lived handle pins its 4 GiB chunk), so their memory would otherwise be lost to this request.
            // The block keeps its own class and goes back to it when freed.
            std::lock_guard<std::mutex> lk(g_alloc_mu);
            auto it = g_free.lower_bound(c);
            if (it != g_free.end()) {
                const size_t cls = it->first;
                *ptr = it->second;
                g_free.erase(it);
                g_cached_bytes -= cls;
                int chunk = -1;
                auto pc = g_parked_chunk.find(*ptr);
                if (pc != g_parked_chunk.end()) { chunk = pc->second; g_parked_chunk.erase(pc); ++g_chunks[chunk].live; }
                g_live[*ptr] = LiveBlock{cls, chunk, cur_index(), req};
                if (canary_on()) canary_set(*ptr, req);
                return SYMGPU_OK;
            }
            set_error("device allocation of %zu bytes failed: %s", c, hipGetErrorString(e));
            *ptr = nullptr;
            return SYMGPU_E_NOMEM;
        }
    }
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    ++g_counters[3];
    g_live[*ptr] = LiveBlock{c, -1, cur_index(), req};
    if (canary_on()) canary_set(*ptr, req);
    return SYMGPU_OK;
}

int dev_free(void *ptr) {
    if (!ptr) return SYMGPU_OK;
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    auto it = g_live.find(ptr);
    if (it == g_live.end()) {
        set_error("dev_free: unknown pointer");
        return SYMGPU_E_INVALID;
    }
    const LiveBlock blk = it->second;
    g_live.erase(it);
    if (canary_on()) canary_check(ptr, blk.req);
    DevAlloc &A = g_alloc[blk.dev];                    // the OWNING device's lists (a handle may be dropped while another device is current)
    if (blk.chunk >= 0) {                              // arena block: parked, whatever the limit says (it cannot go back on its own)
        --A.chunks[blk.chunk].live;
        A.parked_chunk[ptr] = blk.chunk;
        A.free_.insert({blk.cls, ptr});
        A.cached_bytes += blk.cls;
    } else if (A.cached_bytes + blk.cls > A.cache_limit) {
        // stream-ordered safety: everything runs on one stream per device, but hipFree synchronises anyway
        (void)hipFree(ptr);
    } else {
        A.free_.insert({blk.cls, ptr});
        A.cached_bytes += blk.cls;
    }
    return SYMGPU_OK;
}

void dev_cache_release() {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
    for (auto it = g_free.begin(); it != g_free.end();) {
        auto pc = g_parked_chunk.find(it->second);
        if (pc == g_parked_chunk.end()) {              // its own hipMalloc
            (void)hipFree(it->second);
            g_cached_bytes -= it->first;
            it = g_free.erase(it);
        } else if (g_chunks[pc->second].live == 0) {   // arena block of a chunk nobody uses: goes with its chunk below
            g_cached_bytes -= it->first;
            g_parked_chunk.erase(pc);
            it = g_free.erase(it);
        } else {
            ++it;
        }
    }
    for (Chunk &ch : g_chunks)
        if (ch.base && ch.live == 0) { (void)hipFree(ch.base); ch.base = nullptr; ch.size = ch.used = 0; }
}

// ---- per-launch profiling ---------------------------------------------------------------------------
struct ProfClass { bool on = false; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; };
static ProfClass g_prof_all[SYMGPU_MAX_DEVICES][SYMGPU_PROF_CLASSES];
#define g_prof (g_prof_all[cur_index()])

ProfScope::ProfScope(int kernel_class) : cls(kernel_class), on(false) {
    if (cls < 0 || cls >= SYMGPU_PROF_CLASSES || !g_prof[cls].on) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { (void)hipGetLastError(); return; }
    on = true;
    (void)hipEventRecord(a, ctx().stream);
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(b, ctx().stream);
    g_prof[cls].ev.push_back({a, b});
}

// ---- small kernels -----------------------------------------------------------------------------
__global__ void k_xor_fold(const u64 *__restrict__ rows, i64 T, int W, u64 *__restrict__ out) {
    // out[w] ^= XOR over rows; one block per grid-stride chunk, lanes over (row, word) pairs
    extern __shared__ u64 s_fold[];
    for (int w = threadIdx.x; w < W; w += blockDim.x) s_fold[w] = 0;
    __syncthreads();
    i64 total = T * (i64)W;
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (i64)gridDim.x * blockDim.x) {
        int w = (int)(idx % W);
        atomicXor((unsigned long long *)&s_fold[w], (unsigned long long)rows[idx]);
    }
    __syncthreads();
    for (int w = threadIdx.x; w < W; w += blockDim.x)
        if (s_fold[w]) atomicXor((unsigned long long *)&out[w], (unsigned long long)s_fold[w]);
}

__global__ void k_sum_f64x2(const double *__restrict__ c, i64 T, double *__restrict__ out) {
    double re = 0, im = 0;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) {
        re += c[2 * t];
        im += c[2 * t + 1];
    }
    for (int off = 32; off > 0; off >>= 1) {
        re += __shfl_down(re, off);
        im += __shfl_down(im, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], re);
        atomicAdd(&out[1], im);
    }
}

__global__ void k_sum_u8(const uint8_t *__restrict__ p, i64 n, unsigned long long *__restrict__ out) {
    unsigned long long s = 0;
    i64 n16 = n / 16;
    const uint4 *p4 = reinterpret_cast<const uint4 *>(p);
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (i64)gridDim.x * blockDim.x) {
        uint4 v = p4[i];
        // bytes are 0/1: popcount counts them
        s += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (i64 i = n16 * 16; i < n; ++i) s += p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

__global__ void k_popc_u64(const u64 *__restrict__ p, i64 n, unsigned long long *__restrict__ out) {
    unsigned long long s = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) s += __popcll(p[i]);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

__device__ __forceinline__ u64 splitmix64(u64 x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}

// synthetic operator: each bit set with probability `density` (compared on 16-bit slices of a counter hash),
// coefficients from a Box-Muller pair; padding bits zero.
__global__ void k_random_op(u64 *__restrict__ rows, double *__restrict__ coeff, i64 T, int n, int Wq, u32 thresh16, u64 seed) {
    i64 total = T * (i64)(2 * Wq);
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (i64)gridDim.x * blockDim.x) {
        int w = (int)(idx % (2 * Wq));
        int wq = w % Wq;
        u64 word = 0;
        for (int g = 0; g < 16; ++g) {
            u64 r = splitmix64(seed ^ (u64)idx * 16 + g);
            for (int k = 0; k < 4; ++k) {
                int bit = g * 4 + k;
                if (((r >> (16 * k)) & 0xffff) < thresh16 && wq * 64 + bit < n) word |= 1ULL << bit;
            }
        }
        rows[idx] = word;
    }
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) {
        u64 a = splitmix64(seed ^ 0xabcdef12345ULL ^ (u64)t * 2), b = splitmix64(seed ^ 0xabcdef12345ULL ^ ((u64)t * 2 + 1));
        double u1 = ((a >> 11) + 1.0) * (1.0 / 9007199254740993.0), u2 = (b >> 11) * (1.0 / 9007199254740992.0);
        double r = sqrt(-2.0 * log(u1));
        coeff[2 * t] = r * cos(6.283185307179586 * u2);
        coeff[2 * t + 1] = r * sin(6.283185307179586 * u2);
    }
}

// coefficients in place: c <- (conjugate_first ? conj(c) : c) * (re + i im), plain IEEE products (no contraction: NumPy's complex multiply)
__global__ __launch_bounds__(256) void k_scale_coeff(double *__restrict__ c, i64 T, double re, double im, int conjugate_first) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    double2 v = reinterpret_cast<double2 *>(c)[t];
    if (conjugate_first) v.y = -v.y;
    double2 o;
    o.x = __dsub_rn(__dmul_rn(v.x, re), __dmul_rn(v.y, im));
    o.y = __dadd_rn(__dmul_rn(v.x, im), __dmul_rn(v.y, re));
    reinterpret_cast<double2 *>(c)[t] = o;
}

// reference layout (np.bool_ [T][2n], X columns then Z columns, base.py:42-74) <-> packed rows: one wavefront per (term, word); lane l owns
// qubit 64 w + l, so the packed word IS the wavefront's ballot (and a word's 64 bytes are one coalesced store on the way back)
__global__ __launch_bounds__(256) void k_pack_bool(const uint8_t *__restrict__ symp, i64 T, int n, int Wq, u64 *__restrict__ rows) {
    const int lane = threadIdx.x & 63;
    const i64 n_words = T * 2 * Wq;
    for (i64 idx = (i64)blockIdx.x * 4 + (threadIdx.x >> 6); idx < n_words; idx += (i64)gridDim.x * 4) {
        const i64 t = idx / (2 * Wq);
        const int w = (int)(idx % (2 * Wq));
        const int half = w >= Wq, q = (half ? w - Wq : w) * 64 + lane;
        const bool bit = q < n && symp[t * 2 * n + (half ? n : 0) + q] != 0;
        const u64 word = __ballot(bit);
        if (lane == 0) rows[idx] = word;
    }
}
__global__ __launch_bounds__(256) void k_unpack_bool(const u64 *__restrict__ rows, i64 T, int n, int Wq, uint8_t *__restrict__ symp) {
    const int lane = threadIdx.x & 63;
    const i64 n_words = T * 2 * Wq;
    for (i64 idx = (i64)blockIdx.x * 4 + (threadIdx.x >> 6); idx < n_words; idx += (i64)gridDim.x * 4) {
        const i64 t = idx / (2 * Wq);
        const int w = (int)(idx % (2 * Wq));
        const int half = w >= Wq, q = (half ? w - Wq : w) * 64 + lane;
        const u64 word = rows[idx];
        if (q < n) symp[t * 2 * n + (half ? n : 0) + q] = (uint8_t)((word >> lane) & 1);
    }
}

// on-box bandwidth ceilings for the roofline (one 16-byte store / load+store per thread, one-shot grid)
typedef unsigned int u32x4p __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_probe_fill(u32x4p *out, i64 n, u32 v) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (u32x4p)(v);
}
__global__ __launch_bounds__(256) void k_probe_copy(const u32x4p *in, u32x4p *out, i64 n) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// Large device -> host copies into FRESH pageable memory (np.empty) are bound by first-touch page faults taken inside the
// runtime's pinning path: 9-18 GB/s, against 43-55 GB/s once the pages exist (tools/ubench_d2h.hip, MI355X box).  Touch the
// destination pages first, from a few threads; the copy overwrites the whole range anyway.
static void touch_pages(char *base, size_t lo, size_t hi) {
    for (size_t o = lo; o < hi; o += 4096) reinterpret_cast<volatile char *>(base)[o] = 0;
    if (hi > lo) reinterpret_cast<volatile char *>(base)[hi - 1] = 0;
}

static int prefault_threads() {
    const unsigned hw = std::thread::hardware_concurrency();
    return hw >= 16 ? 8 : (hw >= 4 ? 4 : 1);
}

// first touch of [lo, hi) of a large D2H destination from several threads (the runtime's own pinning path takes the first-touch faults
// at 9-18 GB/s, tools/ubench_d2h.hip); the workers are returned running, the caller joins them
static std::vector<std::thread> prefault_start(char *base, size_t lo, size_t hi) {
    std::vector<std::thread> workers;
    const size_t page = 4096;
    const int n_threads = prefault_threads();
    const size_t chunk = ((hi - lo) / n_threads + page - 1) / page * page;
    for (int k = 0; k < n_threads && chunk; ++k) {
        const size_t a = lo + (size_t)k * chunk, b = a + chunk < hi ? a + chunk : hi;
        if (a >= b) break;
        workers.emplace_back(touch_pages, base, a, b);
    }
    return workers;
}

static void ask_for_huge_pages(void *dst, size_t bytes) {
    // transparent huge pages on the page-aligned interior (honoured where THP is 'always' or 'madvise'): 512x fewer faults
    const size_t page = 4096;
    const uintptr_t a = (reinterpret_cast<uintptr_t>(dst) + page - 1) & ~(uintptr_t)(page - 1);
    const uintptr_t b = (reinterpret_cast<uintptr_t>(dst) + bytes) & ~(uintptr_t)(page - 1);
    if (b > a) (void)madvise(reinterpret_cast<void *>(a), b - a, MADV_HUGEPAGE);
}

void prefault_host(void *dst, size_t bytes) {
    if (!dst || bytes < ((size_t)64 << 20)) return;
    ask_for_huge_pages(dst, bytes);
    for (auto &w : prefault_start(static_cast<char *>(dst), 0, bytes)) w.join();
}

// A large device-to-host copy into pageable memory, in pieces: while piece k travels (the copy call blocks its thread) the pages of
// piece k + 1 are touched by the worker threads, so the first-touch faults of a fresh destination (a 40 GB commutation table: ~0.2 s) hide
// behind the PCIe transfer; the first piece is touched while the kernels queued ahead of the copy are still running.
static int download_pipelined(const char *dev, char *host, size_t bytes) {
    const size_t piece = (size_t)1 << 30;
    ask_for_huge_pages(host, bytes);
    for (auto &w : prefault_start(host, 0, piece < bytes ? piece : bytes)) w.join();
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t n = bytes - off < piece ? bytes - off : piece;
        std::vector<std::thread> next;
        if (off + piece < bytes) next = prefault_start(host, off + piece, off + 2 * piece < bytes ? off + 2 * piece : bytes);
        const hipError_t e1 = hipMemcpyAsync(host + off, dev + off, n, hipMemcpyDeviceToHost, ctx().stream);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamSynchronize(ctx().stream) : e1;
        for (auto &w : next) w.join();
        HIP_TRY(e2);
    }
    return SYMGPU_OK;
}

// device -> pageable host memory, any size: small copies as one call, large ones pipelined (see above); returns with the data on the host
static int download_any(const void *dev, void *host, size_t bytes) {
    if (bytes >= ((size_t)2 << 30)) return download_pipelined(static_cast<const char *>(dev), static_cast<char *>(host), bytes);
    prefault_host(host, bytes);
    HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx().stream));
    HIP_TRY(hipStreamSynchronize(ctx().stream));
    return SYMGPU_OK;
}

// ---- a few words back to the host in the middle of a call ---------------------------------------------------------------------------
// hipMemcpyAsync of a few bytes + hipStreamSynchronize costs two host round trips on this runtime (the stream is drained, THEN a blit
// kernel is queued, then drained again: 35 + 25 us of idle GPU per read-back, rocprofv3 timeline of cfg3).  Instead a one-wavefront kernel
// at the end of the queue stores the words into mapped, coherent host memory and a sequence number behind them (system-scope release);
// the host polls the sequence number.  Nothing else of the runtime is involved; after 2 s without an answer (a kernel fault upstream)
// the stream is synchronised the ordinary way and its error reported.
__global__ void k_mail_words(const u32 *__restrict__ a, int n_a, const u32 *__restrict__ b, int n_b, const u32 *__restrict__ c1, u32 *__restrict__ mail, u32 seq) {
    const int t = threadIdx.x;
    if (t < n_a) __hip_atomic_store(mail + 1 + t, a[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (t < n_a + n_b) __hip_atomic_store(mail + 1 + t, b[t - n_a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (t == n_a + n_b && c1) __hip_atomic_store(mail + 1 + t, *c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __builtin_amdgcn_s_barrier();
    if (t == 0) __hip_atomic_store(mail, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static bool mail_ready() {
    Context &c = ctx();
    const char *plain = getenv("SYMGPU_READBACK_PLAIN");
    if (plain && plain[0] == '1') return false;
    if (!c.mail_host && !c.mail_failed) {
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            memset(h, 0, 64);
            c.mail_host = static_cast<u32 *>(h);
            c.mail_dev = static_cast<u32 *>(d);
        } else {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
            c.mail_failed = true;
            note_degraded("read-back through mapped host memory unavailable: every mid-call read-back is a copy + stream synchronisation");
        }
    }
    return c.mail_host != nullptr;
}
// post: queue the words' way home (more work may be queued behind it before the wait); wait: poll, copy out.  One read-back in flight per
// context.
int read_back_post(const u32 *a, int n_a, const u32 *b, int n_b, ReadBack *rb, const u32 *c1) {
    Context &c = ctx();
    rb->n = n_a + n_b + (c1 ? 1 : 0); rb->seq = 0;
    if (rb->n > 8) { set_error("read_back: %d words", rb->n); return SYMGPU_E_INVALID; }
    if (!mail_ready()) {                                                // the copies are queued here, the synchronisation is the wait
        if (n_a) HIP_TRY(hipMemcpyAsync(rb->plain, a, (size_t)n_a * 4, hipMemcpyDeviceToHost, c.stream));
        if (n_b) HIP_TRY(hipMemcpyAsync(rb->plain + n_a, b, (size_t)n_b * 4, hipMemcpyDeviceToHost, c.stream));
        if (c1) HIP_TRY(hipMemcpyAsync(rb->plain + n_a + n_b, c1, 4, hipMemcpyDeviceToHost, c.stream));
        return SYMGPU_OK;
    }
    ++c.mail_seq;
    if (c.mail_seq == 0) ++c.mail_seq;                                  // never 0
    rb->seq = c.mail_seq;
    hipLaunchKernelGGL(k_mail_words, dim3(1), dim3(64), 0, c.stream, a, n_a, b, n_b, c1, c.mail_dev, rb->seq);
    KERNEL_CHECK();
    return SYMGPU_OK;
}
int read_back_wait(ReadBack *rb, u32 *host_out) {
    Context &c = ctx();
    const int n = rb->n;
    if (rb->seq == 0) {
        HIP_TRY(hipStreamSynchronize(c.stream));
        for (int k = 0; k < n; ++k) host_out[k] = rb->plain[k];
        return SYMGPU_OK;
    }
    volatile u32 *mail = c.mail_host;
    const auto t0 = std::chrono::steady_clock::now();
    for (u64 spin = 0;; ++spin) {
        if (__atomic_load_n(&mail[0], __ATOMIC_ACQUIRE) == rb->seq) break;
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            HIP_TRY(hipStreamSynchronize(c.stream));                    // reports a fault upstream; otherwise the words are there now
            if (__atomic_load_n(&mail[0], __ATOMIC_ACQUIRE) != rb->seq) { set_error("read_back: no answer from the device"); return SYMGPU_E_HIP; }
            break;
        }
    }
    for (int k = 0; k < n; ++k) host_out[k] = mail[1 + k];
    return SYMGPU_OK;
}
int read_back_words(const u32 *a, int n_a, const u32 *b, int n_b, u32 *host_out, const u32 *c1) {
    ReadBack rb;
    SG_TRY(read_back_post(a, n_a, b, n_b, &rb, c1));
    return read_back_wait(&rb, host_out);
}

}  // namespace symgpu

using namespace symgpu;

extern "C" {

const char *symgpu_last_error(void) { return g_err; }

int symgpu_device_count(int *n) {
    if (!n) return SYMGPU_E_INVALID;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { (void)hipGetLastError(); c = 0; }
    *n = c;
    return SYMGPU_OK;
}

static int init_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return SYMGPU_E_NODEVICE;
    }
    if (device < 0 || device >= n || device >= SYMGPU_MAX_DEVICES) {
        set_error("device %d out of range (have %d, at most %d per process)", device, n, SYMGPU_MAX_DEVICES);
        return SYMGPU_E_INVALID;
    }
    Context &c = g_ctxs[device];
    if (c.ready) return SYMGPU_OK;
    HIP_TRY(hipSetDevice(device));
    t_bound_dev = device;
    HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c.stream2, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&c.ev0));
    HIP_TRY(hipEventCreate(&c.ev1));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c.device = device;
    {
        size_t f = 0, t = 0;
        if (hipMemGetInfo(&f, &t) == hipSuccess && t / 2 > g_alloc[device].cache_limit) g_alloc[device].cache_limit = t / 2;
    }
    if (const char *e = SG_TUNE("SYMGPU_ARENA")) g_alloc[device].arena_on = !(e[0] == '0');       // 0: every size class straight from hipMalloc (round 2's allocator)
    c.ready = true;
    if (g_default_dev < 0) g_default_dev = device;
    return SYMGPU_OK;
}

int symgpu_init(int device) {
    // creates the device's context if it does not exist yet and makes the device this thread's current one (one process per GPU: the
    // only call a rank makes; single-process multi-device: symgpu_init_all, then symgpu_set_device)
    SG_TRY(init_device(device));
    t_cur_dev = device;
    return require_ctx();
}

int symgpu_init_all(int n_devices) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return SYMGPU_E_NODEVICE;
    }
    if (n_devices <= 0 || n_devices > n) n_devices = n;
    if (n_devices > SYMGPU_MAX_DEVICES) n_devices = SYMGPU_MAX_DEVICES;
    for (int d = 0; d < n_devices; ++d) SG_TRY(init_device(d));
    // peer access both ways between every pair: symgpu_op_copy_rows between devices and RCCL's transports use it (xGMI)
    for (int a = 0; a < n_devices; ++a) {
        HIP_TRY(hipSetDevice(a));
        for (int b = 0; b < n_devices; ++b) {
            if (a == b) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
                const hipError_t pe = hipDeviceEnablePeerAccess(b, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
    }
    t_bound_dev = -1;
    if (t_cur_dev < 0) t_cur_dev = g_default_dev;
    return require_ctx();
}

int symgpu_set_device(int device) {
    SG_TRY(init_device(device));
    t_cur_dev = device;
    return require_ctx();
}

int symgpu_current_device(int *device) {
    SG_REQUIRE(device, "current_device: null argument");
    *device = ctx().ready ? ctx().device : -1;
    return SYMGPU_OK;
}

int symgpu_n_initialised(int *n) {
    SG_REQUIRE(n, "n_initialised: null argument");
    int k = 0;
    for (int d = 0; d < SYMGPU_MAX_DEVICES; ++d) k += g_ctxs[d].ready ? 1 : 0;
    *n = k;
    return SYMGPU_OK;
}

static void shutdown_device(int device) {
    Context &c = g_ctxs[device];
    if (!c.ready) return;
    const int saved = t_cur_dev;
    t_cur_dev = device;
    if (hipSetDevice(device) == hipSuccess) t_bound_dev = device;
    (void)hipStreamSynchronize(c.stream);
    if (c.rot_table) { dev_free(c.rot_table); c.rot_table = nullptr; c.rot_table_cap = 0; c.rot_gen = 0; }
    dev_cache_release();
    if (c.hash_tab) { (void)hipFree(c.hash_tab); c.hash_tab = nullptr; }
    if (c.xs_pow) { (void)hipFree(c.xs_pow); c.xs_pow = nullptr; }
    if (c.rot_flags) { (void)hipFree(c.rot_flags); c.rot_flags = nullptr; }
    if (c.rot_partner) { (void)hipFree(c.rot_partner); c.rot_partner = nullptr; c.rot_partner_cap = 0; }
    if (c.sort_state) { (void)hipFree(c.sort_state); c.sort_state = nullptr; c.sort_bar_base = 0; }
    if (c.sort_scan_ticket) { (void)hipFree(c.sort_scan_ticket); c.sort_scan_ticket = nullptr; }
    if (c.m7_flags) { (void)hipFree(c.m7_flags); c.m7_flags = nullptr; }
    for (auto &q : c.emit_probe) {
        for (auto &ev : q.ev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
        q = EmitProbe();
    }
    if (c.res_table) { (void)hipFree(c.res_table); c.res_table = nullptr; c.res_table_cap = 0; }
    if (c.res_state) { (void)hipFree(c.res_state); c.res_state = nullptr; c.res_epoch = 0; }
    if (c.rot_host_cnt) { (void)hipHostFree(c.rot_host_cnt); c.rot_host_cnt = nullptr; c.rot_host_cnt_dev = nullptr; }
    (void)hipEventDestroy(c.ev0);
    (void)hipEventDestroy(c.ev1);
    (void)hipStreamSynchronize(c.stream2);
    if (c.mail_host) { (void)hipHostFree(c.mail_host); c.mail_host = nullptr; c.mail_dev = nullptr; }
    (void)hipEventDestroy(c.ev_fork);
    (void)hipEventDestroy(c.ev_join);
    (void)hipStreamDestroy(c.stream2);
    (void)hipStreamDestroy(c.stream);
    c = Context();
    t_cur_dev = saved;
}

int symgpu_shutdown(void) {
    for (int d = 0; d < SYMGPU_MAX_DEVICES; ++d) shutdown_device(d);
    g_default_dev = -1;
    t_cur_dev = -1;
    t_bound_dev = -1;
    return SYMGPU_OK;
}

int symgpu_sync(void) {
    SG_TRY(require_ctx());
    HIP_TRY(hipStreamSynchronize(ctx().stream));
    return SYMGPU_OK;
}

int symgpu_device_sync(void) {
    SG_TRY(require_ctx());
    HIP_TRY(hipDeviceSynchronize());
    return SYMGPU_OK;
}

int symgpu_device_name(char *buf, int len) {
    SG_TRY(require_ctx());
    if (!buf || len <= 0) return SYMGPU_E_INVALID;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx().device));
    // (some driver stacks leave the marketing name empty: say so instead of printing nothing)
    snprintf(buf, (size_t)len, "%s (%s, %d CUs)", prop.name[0] ? prop.name : "AMD GPU, name not reported by the driver", prop.gcnArchName, prop.multiProcessorCount);
    return SYMGPU_OK;
}

int symgpu_mem_info(int64_t *free_bytes, int64_t *total_bytes) {
    SG_TRY(require_ctx());
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return SYMGPU_OK;
}

int symgpu_timer_start(void) {
    SG_TRY(require_ctx());
    HIP_TRY(hipEventRecord(ctx().ev0, ctx().stream));
    return SYMGPU_OK;
}

int symgpu_timer_stop(float *ms) {
    SG_TRY(require_ctx());
    HIP_TRY(hipEventRecord(ctx().ev1, ctx().stream));
    HIP_TRY(hipEventSynchronize(ctx().ev1));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, ctx().ev0, ctx().ev1));
    if (ms) *ms = t;
    return SYMGPU_OK;
}

int symgpu_membw_probe(int64_t bytes, double *fill_GBps, double *copy_GBps) {
    SG_TRY(require_ctx());
    SG_REQUIRE(bytes >= (1 << 20), "membw_probe: at least 1 MiB");
    const i64 n = bytes / 16;
    Scratch a, b;
    SG_TRY(a.alloc((size_t)n * 16));
    SG_TRY(b.alloc((size_t)n * 16));
    Context &c = ctx();
    const unsigned grid = (unsigned)((n + 255) / 256);
    float best_fill = 1e30f, best_copy = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        float ms = 0;
        HIP_TRY(hipEventRecord(c.ev0, c.stream));
        hipLaunchKernelGGL(k_probe_fill, dim3(grid), dim3(256), 0, c.stream, a.as<u32x4p>(), n, 7u + rep);
        HIP_TRY(hipEventRecord(c.ev1, c.stream));
        HIP_TRY(hipEventSynchronize(c.ev1));
        HIP_TRY(hipEventElapsedTime(&ms, c.ev0, c.ev1));
        if (rep && ms < best_fill) best_fill = ms;
        HIP_TRY(hipEventRecord(c.ev0, c.stream));
        hipLaunchKernelGGL(k_probe_copy, dim3(grid), dim3(256), 0, c.stream, a.as<u32x4p>(), b.as<u32x4p>(), n);
        HIP_TRY(hipEventRecord(c.ev1, c.stream));
        HIP_TRY(hipEventSynchronize(c.ev1));
        HIP_TRY(hipEventElapsedTime(&ms, c.ev0, c.ev1));
        if (rep && ms < best_copy) best_copy = ms;
    }
    KERNEL_CHECK();
    if (fill_GBps) *fill_GBps = (double)n * 16 / (best_fill * 1e-3) / 1e9;
    if (copy_GBps) *copy_GBps = 2.0 * (double)n * 16 / (best_copy * 1e-3) / 1e9;
    return SYMGPU_OK;
}

int symgpu_prof_enable(int kernel_class, int on) {
    SG_REQUIRE(kernel_class >= 0 && kernel_class < SYMGPU_PROF_CLASSES, "prof_enable: class");
    g_prof[kernel_class].on = on != 0;
    return SYMGPU_OK;
}

int symgpu_debug_counter(int which, int64_t *value) {
    SG_REQUIRE(value && which >= 0 && which <= 10, "debug_counter: 0 = row-hash reseeds, 1 = rotations done by the one-launch kernel, 2 = its failures (verification / time-out), 3 = device allocations that went to hipMalloc, 4-6 = host nanoseconds of the one-launch rotation (preparation, launch call, wait), 7 / 8 = payload bytes host -> device / device -> host, 9 / 10 = operator uploads / downloads");
    *value = which == 0 ? g_hash_reseeds : g_counters[which];
    return SYMGPU_OK;
}

int symgpu_degraded(char *buf, int len) {
    if (!buf || len <= 0) return SYMGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(g_deg_mu);
    snprintf(buf, (size_t)len, "%s", g_degraded.c_str());
    return SYMGPU_OK;
}

int symgpu_prof_read(int kernel_class, int64_t *n_launches, double *total_ms) {
    SG_TRY(require_ctx());
    SG_REQUIRE(kernel_class >= 0 && kernel_class < SYMGPU_PROF_CLASSES, "prof_read: class");
    HIP_TRY(hipStreamSynchronize(ctx().stream));
    double tot = 0;
    for (auto &p : g_prof[kernel_class].ev) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) tot += ms;
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    if (n_launches) *n_launches = (int64_t)g_prof[kernel_class].ev.size();
    if (total_ms) *total_ms = tot;
    g_prof[kernel_class].ev.clear();
    return SYMGPU_OK;
}

// ---- raw device buffers ------------------------------------------------------------------------
int symgpu_dev_alloc(int64_t bytes, void **ptr) {
    SG_REQUIRE(ptr && bytes >= 0, "dev_alloc");
    return dev_alloc((size_t)bytes, ptr);
}
int symgpu_dev_free(void *ptr) { return dev_free(ptr); }

int symgpu_dev_download(const void *dev, void *host, int64_t bytes) {
    SG_TRY(require_ctx());
    SG_REQUIRE(dev && host && bytes >= 0, "dev_download");
    SG_TRY(download_any(dev, host, (size_t)bytes));
    count_d2h((size_t)bytes);
    return SYMGPU_OK;
}

int symgpu_dev_upload(void *dev, const void *host, int64_t bytes) {
    SG_TRY(require_ctx());
    SG_REQUIRE(dev && host && bytes >= 0, "dev_upload");
    HIP_TRY(hipMemcpyAsync(dev, host, (size_t)bytes, hipMemcpyHostToDevice, ctx().stream));
    HIP_TRY(hipStreamSynchronize(ctx().stream));
    count_h2d((size_t)bytes);
    return SYMGPU_OK;
}

static int reduce_to_host_u64(void (*launch)(const void *, i64, unsigned long long *, hipStream_t), const void *p, i64 n, uint64_t *sum) {
    Scratch acc;
    SG_TRY(acc.alloc(sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(acc.p, 0, sizeof(unsigned long long), ctx().stream));
    launch(p, n, acc.as<unsigned long long>(), ctx().stream);
    KERNEL_CHECK();
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, acc.p, sizeof(h), hipMemcpyDeviceToHost, ctx().stream));
    HIP_TRY(hipStreamSynchronize(ctx().stream));
    *sum = h;
    return SYMGPU_OK;
}

int symgpu_dev_checksum_u8(const uint8_t *dev, int64_t n, uint64_t *sum) {
    SG_TRY(require_ctx());
    SG_REQUIRE(dev && sum && n >= 0, "dev_checksum_u8");
    SG_REQUIRE(((uintptr_t)dev & 15) == 0, "dev_checksum_u8: pointer must be 16-byte aligned");
    return reduce_to_host_u64([](const void *p, i64 n_, unsigned long long *o, hipStream_t s) {
        hipLaunchKernelGGL(k_sum_u8, dim3(2048), dim3(256), 0, s, (const uint8_t *)p, n_, o); }, dev, n, sum);
}

int symgpu_dev_popcount_u64(const uint64_t *dev, int64_t n_words, uint64_t *sum) {
    SG_TRY(require_ctx());
    SG_REQUIRE(dev && sum && n_words >= 0, "dev_popcount_u64");
    return reduce_to_host_u64([](const void *p, i64 n_, unsigned long long *o, hipStream_t s) {
        hipLaunchKernelGGL(k_popc_u64, dim3(2048), dim3(256), 0, s, (const u64 *)p, n_, o); }, dev, n_words, sum);
}

// ---- operator handles ----------------------------------------------------------------------------
int symgpu_op_alloc(int64_t capacity_rows, int Wq, int with_coeff, symgpu_op_t *out) {
    SG_TRY(require_ctx());
    SG_REQUIRE(out && capacity_rows >= 0 && Wq >= 1, "op_alloc");
    symgpu_op_s *op = new symgpu_op_s();
    op->device = ctx().device;
    op->Wq = Wq;
    op->capacity = capacity_rows;
    op->T = 0;
    int rc = dev_alloc((size_t)capacity_rows * 2 * Wq * sizeof(u64), (void **)&op->rows);
    if (rc == SYMGPU_OK && with_coeff) rc = dev_alloc((size_t)capacity_rows * 2 * sizeof(double), (void **)&op->coeff);
    if (rc != SYMGPU_OK) {
        if (op->rows) dev_free(op->rows);
        delete op;
        return rc;
    }
    *out = op;
    return SYMGPU_OK;
}

int symgpu_op_free(symgpu_op_t op) {
    if (!op) return SYMGPU_OK;
    op_invalidate(op);
    if (op->rows) dev_free(op->rows);
    if (op->coeff) dev_free(op->coeff);
    delete op;
    return SYMGPU_OK;
}

int symgpu_op_set_rows(symgpu_op_t op, int64_t T) {
    SG_REQUIRE(op && T >= 0 && T <= op->capacity, "op_set_rows");
    if (T != op->T) op_invalidate(op);
    op->T = T;
    return SYMGPU_OK;
}

int symgpu_op_write(symgpu_op_t op, int64_t row_offset, const uint64_t *rows, const double *coeff, int64_t count) {
    SG_ENTER(op);
    SG_REQUIRE(op && row_offset >= 0 && count >= 0 && row_offset + count <= op->capacity, "op_write: row range exceeds the capacity");
    SG_REQUIRE(count == 0 || rows, "op_write: null rows");
    const size_t W = (size_t)2 * op->Wq;
    if (count > 0) {
        HIP_TRY(hipMemcpyAsync(op->rows + (size_t)row_offset * W, rows, (size_t)count * W * 8, hipMemcpyHostToDevice, ctx().stream));
        if (coeff && op->coeff)
            HIP_TRY(hipMemcpyAsync(op->coeff + 2 * (size_t)row_offset, coeff, (size_t)count * 16, hipMemcpyHostToDevice, ctx().stream));
        HIP_TRY(hipStreamSynchronize(ctx().stream));
        count_h2d((size_t)count * W * 8 + ((coeff && op->coeff) ? (size_t)count * 16 : 0));
        ++g_counters[9];
    }
    op_invalidate(op);
    if (row_offset + count > op->T) op->T = row_offset + count;
    return SYMGPU_OK;
}

int symgpu_op_copy_rows(symgpu_op_t dst, int64_t dst_offset, symgpu_op_t src, int64_t src_offset, int64_t count) {
    SG_REQUIRE(dst && src && dst != src && dst->Wq == src->Wq, "op_copy_rows: handles");
    SG_ENTER(dst);                                                     // runs on the destination's device
    SG_REQUIRE(count >= 0 && dst_offset >= 0 && src_offset >= 0 && dst_offset + count <= dst->capacity && src_offset + count <= src->T,
               "op_copy_rows: row range");
    const size_t W = (size_t)2 * dst->Wq;
    if (count > 0 && src->device != dst->device) {
        // the one call that crosses devices: a peer copy (xGMI) on the destination's stream, after the source's stream has drained
        Context *sc = ctx_of_device(src->device);
        SG_REQUIRE(sc && sc->ready, "op_copy_rows: the source's device has no context");
        HIP_TRY(hipStreamSynchronize(sc->stream));
        HIP_TRY(hipMemcpyPeerAsync(dst->rows + (size_t)dst_offset * W, dst->device, src->rows + (size_t)src_offset * W, src->device, (size_t)count * W * 8, ctx().stream));
        if (dst->coeff && src->coeff)
            HIP_TRY(hipMemcpyPeerAsync(dst->coeff + 2 * (size_t)dst_offset, dst->device, src->coeff + 2 * (size_t)src_offset, src->device, (size_t)count * 16, ctx().stream));
    } else if (count > 0) {
        HIP_TRY(hipMemcpyAsync(dst->rows + (size_t)dst_offset * W, src->rows + (size_t)src_offset * W, (size_t)count * W * 8, hipMemcpyDeviceToDevice, ctx().stream));
        if (dst->coeff && src->coeff)
            HIP_TRY(hipMemcpyAsync(dst->coeff + 2 * (size_t)dst_offset, src->coeff + 2 * (size_t)src_offset, (size_t)count * 16, hipMemcpyDeviceToDevice, ctx().stream));
    }
    op_invalidate(dst);
    if (dst_offset + count > dst->T) dst->T = dst_offset + count;
    return SYMGPU_OK;
}

int symgpu_op_upload(const uint64_t *rows, const double *coeff, int64_t T, int Wq, symgpu_op_t *out) {
    SG_TRY(require_ctx());
    SG_REQUIRE(out && T >= 0 && Wq >= 1 && (rows || T == 0), "op_upload");
    symgpu_op_t op = nullptr;
    SG_TRY(symgpu_op_alloc(T, Wq, coeff != nullptr, &op));
    op->T = T;
    if (T > 0) {
        hipError_t e = hipMemcpyAsync(op->rows, rows, (size_t)T * 2 * Wq * sizeof(u64), hipMemcpyHostToDevice, ctx().stream);
        if (e == hipSuccess && coeff) e = hipMemcpyAsync(op->coeff, coeff, (size_t)T * 2 * sizeof(double), hipMemcpyHostToDevice, ctx().stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx().stream);   // host buffers are not retained past the call
        if (e != hipSuccess) { symgpu_op_free(op); return hip_fail(e, "op_upload memcpy", __FILE__, __LINE__); }
        count_h2d((size_t)T * 2 * Wq * sizeof(u64) + (coeff ? (size_t)T * 16 : 0));
        ++g_counters[9];
    }
    *out = op;
    return SYMGPU_OK;
}

int symgpu_op_download(symgpu_op_t op, uint64_t *rows, double *coeff, int64_t capacity_rows) {
    SG_ENTER(op);
    SG_REQUIRE(op, "op_download: null handle");
    if (capacity_rows < op->T) {
        set_error("op_download: capacity %lld < %lld rows", (long long)capacity_rows, (long long)op->T);
        return SYMGPU_E_CAPACITY;
    }
    if (op->T > 0) {
        if (rows) {
            SG_TRY(download_any(op->rows, rows, (size_t)op->T * 2 * op->Wq * sizeof(u64)));
            count_d2h((size_t)op->T * 2 * op->Wq * sizeof(u64));
        }
        if (coeff) {
            SG_REQUIRE(op->coeff, "op_download: operator has no coefficients");
            prefault_host(coeff, (size_t)op->T * 2 * sizeof(double));
            HIP_TRY(hipMemcpyAsync(coeff, op->coeff, (size_t)op->T * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
            count_d2h((size_t)op->T * 16);
        }
        if (rows || coeff) ++g_counters[10];
    }
    HIP_TRY(hipStreamSynchronize(ctx().stream));
    return SYMGPU_OK;
}

// ---- handle-level primitives behind the device-resident drop-in classes (symmer_amd/operators/base.py) -----------------------------------
int symgpu_op_clone(symgpu_op_t in, symgpu_op_t *out) {
    SG_ENTER(in);
    SG_REQUIRE(in && out, "op_clone: null argument");
    symgpu_op_t op = nullptr;
    SG_TRY(symgpu_op_alloc(in->T, in->Wq, in->coeff != nullptr, &op));
    op->T = in->T;
    if (in->T > 0) {
        hipError_t e = hipMemcpyAsync(op->rows, in->rows, (size_t)in->T * 2 * in->Wq * sizeof(u64), hipMemcpyDeviceToDevice, ctx().stream);
        if (e == hipSuccess && in->coeff) e = hipMemcpyAsync(op->coeff, in->coeff, (size_t)in->T * 16, hipMemcpyDeviceToDevice, ctx().stream);
        if (e != hipSuccess) { symgpu_op_free(op); return hip_fail(e, "op_clone memcpy", __FILE__, __LINE__); }
    }
    op->dup_free = in->dup_free;          // the rows are the same rows
    *out = op;
    return SYMGPU_OK;
}

int symgpu_op_set_coeff(symgpu_op_t op, const double *coeff_host) {
    SG_ENTER(op);
    SG_REQUIRE(op && (coeff_host || op->T == 0), "op_set_coeff: null argument");
    if (!op->coeff) SG_TRY(dev_alloc((size_t)(op->capacity > 0 ? op->capacity : 1) * 16, (void **)&op->coeff));
    if (op->T > 0) {
        HIP_TRY(hipMemcpyAsync(op->coeff, coeff_host, (size_t)op->T * 16, hipMemcpyHostToDevice, ctx().stream));
        HIP_TRY(hipStreamSynchronize(ctx().stream));              // host buffers are not retained past the call
        count_h2d((size_t)op->T * 16);
        ++g_counters[9];
    }
    return SYMGPU_OK;                                             // the rows did not change: per-handle caches stay
}

int symgpu_op_scale(symgpu_op_t op, double re, double im, int conjugate_first) {
    SG_ENTER(op);
    SG_REQUIRE(op && (op->coeff || op->T == 0), "op_scale: operator has no coefficients");
    if (op->T > 0) {
        hipLaunchKernelGGL(k_scale_coeff, dim3((unsigned)((op->T + 255) / 256)), dim3(256), 0, ctx().stream, op->coeff, op->T, re, im, conjugate_first);
        KERNEL_CHECK();
    }
    return SYMGPU_OK;
}

int symgpu_op_ycount(symgpu_op_t op, int64_t *out_host) {
    SG_ENTER(op);
    SG_REQUIRE(op && (out_host || op->T == 0), "op_ycount: null argument");
    if (op->T == 0) return SYMGPU_OK;
    const int *yc = nullptr;
    SG_TRY(op_ycount(op, &yc));
    int *h = (int *)malloc((size_t)op->T * sizeof(int));
    if (!h) { set_error("host allocation failed"); return SYMGPU_E_NOMEM; }
    hipError_t e = hipMemcpyAsync(h, yc, (size_t)op->T * sizeof(int), hipMemcpyDeviceToHost, ctx().stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx().stream);
    if (e != hipSuccess) { free(h); return hip_fail(e, "op_ycount download", __FILE__, __LINE__); }
    count_d2h((size_t)op->T * sizeof(int));
    for (i64 t = 0; t < op->T; ++t) out_host[t] = h[t];
    free(h);
    return SYMGPU_OK;
}

int symgpu_op_upload_bool(const uint8_t *symp, const double *coeff, int64_t T, int n_qubits, symgpu_op_t *out) {
    SG_TRY(require_ctx());
    SG_REQUIRE(out && T >= 0 && n_qubits >= 1 && (symp || T == 0), "op_upload_bool");
    const int Wq = (n_qubits + 63) / 64;
    symgpu_op_t op = nullptr;
    SG_TRY(symgpu_op_alloc(T, Wq, coeff != nullptr, &op));
    op->T = T;
    if (T > 0) {
        const size_t nb = (size_t)T * 2 * n_qubits;
        Scratch stage;
        int rc = stage.alloc(nb);
        if (rc != SYMGPU_OK) { symgpu_op_free(op); return rc; }
        hipError_t e = hipMemcpyAsync(stage.p, symp, nb, hipMemcpyHostToDevice, ctx().stream);
        if (e == hipSuccess && coeff) e = hipMemcpyAsync(op->coeff, coeff, (size_t)T * 16, hipMemcpyHostToDevice, ctx().stream);
        if (e == hipSuccess) {
            // one wavefront per (row, word): lane l reads the byte of qubit 64 w + l, the ballot is the packed word
            const i64 n_words = T * 2 * Wq;
            const unsigned grid = (unsigned)((n_words + 3) / 4 < 65536 * 16 ? (n_words + 3) / 4 : 65536 * 16);
            hipLaunchKernelGGL(k_pack_bool, dim3(grid), dim3(256), 0, ctx().stream, stage.as<uint8_t>(), T, n_qubits, Wq, op->rows);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx().stream);   // host buffers are not retained past the call (and the staging buffer goes)
        if (e != hipSuccess) { symgpu_op_free(op); return hip_fail(e, "op_upload_bool", __FILE__, __LINE__); }
        count_h2d(nb + (coeff ? (size_t)T * 16 : 0));
        ++g_counters[9];
    }
    *out = op;
    return SYMGPU_OK;
}

int symgpu_op_download_bool(symgpu_op_t op, int n_qubits, uint8_t *symp_out, int64_t capacity_rows) {
    SG_ENTER(op);
    SG_REQUIRE(op && n_qubits >= 1 && (n_qubits + 63) / 64 == op->Wq, "op_download_bool: qubit count does not match the packed width");
    if (capacity_rows < op->T) {
        set_error("op_download_bool: capacity %lld < %lld rows", (long long)capacity_rows, (long long)op->T);
        return SYMGPU_E_CAPACITY;
    }
    if (op->T == 0) return SYMGPU_OK;
    SG_REQUIRE(symp_out, "op_download_bool: null output");
    const size_t nb = (size_t)op->T * 2 * n_qubits;
    Scratch stage;
    SG_TRY(stage.alloc(nb));
    const i64 n_words = op->T * 2 * op->Wq;
    const unsigned grid = (unsigned)((n_words + 3) / 4 < 65536 * 16 ? (n_words + 3) / 4 : 65536 * 16);
    hipLaunchKernelGGL(k_unpack_bool, dim3(grid), dim3(256), 0, ctx().stream, op->rows, op->T, n_qubits, op->Wq, stage.as<uint8_t>());
    KERNEL_CHECK();
    SG_TRY(download_any(stage.p, symp_out, nb));
    count_d2h(nb);
    ++g_counters[10];
    return SYMGPU_OK;
}

int symgpu_op_info(symgpu_op_t op, int64_t *T, int *Wq, int64_t *capacity_rows) {
    SG_REQUIRE(op, "op_info: null handle");
    if (T) *T = op->T;
    if (Wq) *Wq = op->Wq;
    if (capacity_rows) *capacity_rows = op->capacity;
    return SYMGPU_OK;
}

int symgpu_op_random(int64_t T, int n_qubits, double density, uint64_t seed, symgpu_op_t *out) {
    SG_TRY(require_ctx());
    SG_REQUIRE(out && T >= 0 && n_qubits >= 1 && density >= 0.0 && density <= 1.0, "op_random");
    int Wq = (n_qubits + 63) / 64;
    symgpu_op_t op = nullptr;
    SG_TRY(symgpu_op_alloc(T, Wq, 1, &op));
    op->T = T;
    if (T > 0) {
        u32 th = (u32)(density * 65536.0 + 0.5);
        hipLaunchKernelGGL(k_random_op, dim3(4096), dim3(256), 0, ctx().stream, op->rows, op->coeff, T, n_qubits, Wq, th, seed);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { symgpu_op_free(op); return hip_fail(e, "k_random_op", __FILE__, __LINE__); }
    }
    *out = op;
    return SYMGPU_OK;
}

int symgpu_op_popcount(symgpu_op_t op, uint64_t *sum) {
    SG_ENTER(op);
    SG_REQUIRE(op && sum, "op_popcount: null argument");
    if (op->T == 0) { *sum = 0; return SYMGPU_OK; }
    return symgpu_dev_popcount_u64(op->rows, op->T * 2 * op->Wq, sum);
}

int symgpu_op_checksum(symgpu_op_t op, uint64_t *xor_words, double *coeff_sum) {
    SG_ENTER(op);
    SG_REQUIRE(op, "op_checksum: null handle");
    int W = 2 * op->Wq;
    Scratch acc;
    SG_TRY(acc.alloc((size_t)W * sizeof(u64) + 2 * sizeof(double)));
    HIP_TRY(hipMemsetAsync(acc.p, 0, (size_t)W * sizeof(u64) + 2 * sizeof(double), ctx().stream));
    u64 *dx = acc.as<u64>();
    double *dc = reinterpret_cast<double *>(dx + W);
    if (op->T > 0) {
        if (xor_words) {
            hipLaunchKernelGGL(k_xor_fold, dim3(1024), dim3(256), (size_t)W * sizeof(u64), ctx().stream, op->rows, op->T, W, dx);
            KERNEL_CHECK();
        }
        if (coeff_sum && op->coeff) {
            hipLaunchKernelGGL(k_sum_f64x2, dim3(1024), dim3(256), 0, ctx().stream, op->coeff, op->T, dc);
            KERNEL_CHECK();
        }
    }
    if (xor_words) HIP_TRY(hipMemcpyAsync(xor_words, dx, (size_t)W * sizeof(u64), hipMemcpyDeviceToHost, ctx().stream));
    if (coeff_sum) HIP_TRY(hipMemcpyAsync(coeff_sum, dc, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
    HIP_TRY(hipStreamSynchronize(ctx().stream));
    return SYMGPU_OK;
}

}  // extern "C"
