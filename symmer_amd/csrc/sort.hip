// sort.hip — device-wide exclusive scan and stable LSD radix sort of (u64 key, u32 value) pairs or bare u64 keys.
//
// Used by term cleanup (reference: symplectic_cleanup, symmer/operators/utils.py:230-279) to group equal
// rows: keys are 64-bit GF(2)-linear row hashes, values are input indices.  The sort is STABLE, so inside
// a run of equal keys the values stay in ascending input order — which is exactly the order in which the
// reference's `np.add.at` accumulates duplicate coefficients (utils.py:273-274).
//
// HBM-bound: per 8-bit pass each element is read twice (histogram + scatter: 8 B + 12 B) and written once
// (12 B).  Tiles of 4096 elements are ranked with 64-wide wavefront ballots (one match mask per lane from 8
// __ballot calls), staged through LDS in digit order and written out as contiguous per-digit runs.
#include "common.h"
#include <stdlib.h>

namespace symgpu {

typedef int32_t i32;

// ------------------------------------------------------------------------------------------------
// exclusive scan (u32), 2048 elements per block, recursive over block sums
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;

__device__ __forceinline__ u32 wave_incl_scan(u32 v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        u32 n = __shfl_up(v, off);
        if (lane >= off) v += n;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix, *total = block sum
__device__ __forceinline__ u32 block_excl_scan_256(u32 v, u32 *s_wave /* [4] */, u32 *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = wave_incl_scan(v, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    u32 off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        u32 s = s_wave[k];
        if (k < wave) off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return off + incl - v;
}

__global__ __launch_bounds__(256) void k_scan_block(const u32 *__restrict__ in, u32 *__restrict__ out, i64 n, u32 *__restrict__ block_sums) {
    __shared__ u32 s_wave[4];
    const i64 base = (i64)blockIdx.x * SCAN_BLOCK + (i64)threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS];
    u32 sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        sum += v[k];
    }
    u32 total;
    u32 excl = block_excl_scan_256(sum, s_wave, &total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = excl;
        excl += v[k];
    }
    if (threadIdx.x == 0 && block_sums) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_scan_add(u32 *__restrict__ out, i64 n, const u32 *__restrict__ block_offsets) {
    const i64 base = (i64)blockIdx.x * SCAN_BLOCK + (i64)threadIdx.x * SCAN_ITEMS;
    const u32 off = block_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) out[base + k] += off;
}

__global__ void k_store_total(const u32 *__restrict__ last_in, const u32 *__restrict__ last_out, u32 *__restrict__ total) {
    *total = *last_in + *last_out;
}

// Two launches for up to 2^23 elements (every caller but the very largest): (1) block sums, the LAST workgroup to finish scans them
// (ticket; the sums travel write-through) and files the total, (2) every block scans its 2,048 elements on top of its offset.  The
// recursive form below (block scan, scan of the sums, add, plus a copy and a store for the total) was five or six launches of 4-6 us each
// around a few hundred KB of data.
constexpr int SCAN2_MAX_BLOCKS = 4096;
__global__ __launch_bounds__(256) void k_scan_sums(const u32 *__restrict__ in, i64 n, u32 *__restrict__ block_sums, u32 *__restrict__ total, u32 *__restrict__ ticket) {
    __shared__ u32 s_wave[4];
    __shared__ u32 s_last;
    const i64 base = (i64)blockIdx.x * SCAN_BLOCK + (i64)threadIdx.x * SCAN_ITEMS;
    u32 sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) sum += (base + k < n) ? in[base + k] : 0u;
    u32 tot;
    (void)block_excl_scan_256(sum, s_wave, &tot);
    if (threadIdx.x == 0) {
        __hip_atomic_store(block_sums + blockIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // exclusive scan of the block sums in place: sixteen per thread
    constexpr int PER = SCAN2_MAX_BLOCKS / 256;
    u32 v[PER], mine = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int b = threadIdx.x * PER + k;
        v[k] = b < (int)gridDim.x ? __hip_atomic_load(block_sums + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        mine += v[k];
    }
    u32 all;
    u32 excl = block_excl_scan_256(mine, s_wave, &all);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int b = threadIdx.x * PER + k;
        if (b < (int)gridDim.x) block_sums[b] = excl;
        excl += v[k];
    }
    if (threadIdx.x == 0) { if (total) *total = all; *ticket = 0; }
}
__global__ __launch_bounds__(256) void k_scan_final(const u32 *__restrict__ in, u32 *__restrict__ out, i64 n, const u32 *__restrict__ block_offsets) {
    __shared__ u32 s_wave[4];
    const i64 base = (i64)blockIdx.x * SCAN_BLOCK + (i64)threadIdx.x * SCAN_ITEMS;
    const u32 off = block_offsets[blockIdx.x];
    u32 v[SCAN_ITEMS];
    u32 sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        sum += v[k];
    }
    u32 total;
    u32 excl = off + block_excl_scan_256(sum, s_wave, &total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = excl;
        excl += v[k];
    }
}

// Population counts of the 64-bit words of a bitmap and their exclusive prefix in ONE launch, for bitmaps of up to 2^13 words (512K bits):
// the count kernel + the two scan kernels are 5 us each on an idle queue, which is most of a small product's prefix.  One workgroup of
// 1,024 lanes, a lane takes up to eight consecutive words (loaded together).  n32 >= 0: the bitmap was written in 32-bit words, the upper
// half of the last 64-bit word is unwritten when n32 is odd.
__global__ __launch_bounds__(1024) void k_popc_scan_small(const u64 *__restrict__ bits, i64 n, i64 n32, u32 *__restrict__ out, u32 *__restrict__ total) {
    __shared__ u32 s_wave[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 b0 = (i64)threadIdx.x * 8;
    u32 cnt[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const i64 w = b0 + k;
        u64 b = w < n ? bits[w] : 0ULL;
        if (n32 >= 0 && w == n - 1 && (n32 & 1)) b &= 0xFFFFFFFFULL;
        cnt[k] = (u32)__popcll(b);
        sum += cnt[k];
    }
    const u32 incl = wave_incl_scan(sum, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    u32 off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const u32 sv = s_wave[k];
        if (k < wave) off += sv;
        tot += sv;
    }
    u32 run = off + incl - sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (b0 + k < n) out[b0 + k] = run;
        run += cnt[k];
    }
    if (threadIdx.x == 0 && total) *total = tot;
}
int popc_scan_small(const u64 *bits, i64 n, i64 n32, u32 *out, u32 *total_dev) {
    hipLaunchKernelGGL(k_popc_scan_small, dim3(1), dim3(1024), 0, ctx().stream, bits, n, n32, out, total_dev);
    KERNEL_CHECK();
    return SYMGPU_OK;
}

// out may alias in.  total_dev (optional, device) receives the sum of all n inputs.
int exclusive_scan_u32(const u32 *in, u32 *out, i64 n, u32 *total_dev) {
    Context &c = ctx();
    hipStream_t st = c.stream;
    if (n <= 0) {
        if (total_dev) HIP_TRY(hipMemsetAsync(total_dev, 0, sizeof(u32), st));
        return SYMGPU_OK;
    }
    const i64 nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    const bool force_recursive = [] { const char *e = getenv("SYMGPU_SCAN_RECURSIVE"); return e && e[0] == '1'; }();     // tests: the form of the largest inputs
    if (nb <= SCAN2_MAX_BLOCKS && !force_recursive) {
        if (!c.sort_scan_ticket) {
            HIP_TRY(hipMalloc((void **)&c.sort_scan_ticket, 256));
            HIP_TRY(hipMemsetAsync(c.sort_scan_ticket, 0, 256, st));
        }
        Scratch sums;
        SG_TRY(sums.alloc((size_t)nb * sizeof(u32)));
        hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(256), 0, st, in, n, sums.as<u32>(), total_dev, c.sort_scan_ticket + 1);
        hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(256), 0, st, in, out, n, sums.as<u32>());
        KERNEL_CHECK();
        return SYMGPU_OK;
    }
    Scratch last;   // keep in[n-1] before it is overwritten (aliasing) to form the total
    if (total_dev) {
        SG_TRY(last.alloc(sizeof(u32)));
        HIP_TRY(hipMemcpyAsync(last.p, in + (n - 1), sizeof(u32), hipMemcpyDeviceToDevice, st));
    }
    if (nb == 1) {
        hipLaunchKernelGGL(k_scan_block, dim3(1), dim3(256), 0, st, in, out, n, (u32 *)nullptr);
        KERNEL_CHECK();
    } else {
        Scratch sums;
        SG_TRY(sums.alloc((size_t)nb * sizeof(u32)));
        hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nb), dim3(256), 0, st, in, out, n, sums.as<u32>());
        KERNEL_CHECK();
        SG_TRY(exclusive_scan_u32(sums.as<u32>(), sums.as<u32>(), nb, nullptr));
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(256), 0, st, out, n, sums.as<u32>());
        KERNEL_CHECK();
    }
    if (total_dev) {
        hipLaunchKernelGGL(k_store_total, dim3(1), dim3(1), 0, st, last.as<u32>(), out + (n - 1), total_dev);
        KERNEL_CHECK();
    }
    return SYMGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// radix sort
// ------------------------------------------------------------------------------------------------
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = 256 * RS_ITEMS;   // 4096
constexpr int RS_WSEG = 64 * RS_ITEMS;    // keys per wave segment

// the lanes of the wavefront that hold the same 8-bit digit as this lane (and are valid): per bit one sign-extended field, one
// compare for the ballot and one three-input op per mask half (m & ~(ballot ^ sel)); the plain `m &= bit ? bal : ~bal` compiled to
// 95 VALU instructions per key, this to 45
__device__ __forceinline__ u64 match_digit(u32 d, bool valid) {
    const u64 m0 = __ballot(valid);
    u32 lo = (u32)m0, hi = (u32)(m0 >> 32);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int sel = __builtin_amdgcn_sbfe((int)d, b, 1);                 // 0 or -1
        const u64 bal = __ballot(sel != 0);
        lo = __builtin_amdgcn_bitop3_b32(lo, (u32)bal, (u32)sel, 0x90);       // lo & ~(bal ^ sel)
        hi = __builtin_amdgcn_bitop3_b32(hi, (u32)(bal >> 32), (u32)sel, 0x90);
    }
    return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void k_rs_hist(const u64 *__restrict__ keys, i64 n, int shift, i64 n_tiles, u32 *__restrict__ tile_hist) {
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const i64 base = (i64)blockIdx.x * RS_TILE;
    u64 kk[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {                               // sixteen loads in flight
        const i64 idx = base + k * 256 + threadIdx.x;
        kk[k] = idx < n ? keys[idx] : 0ULL;
    }
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k)
        if (base + k * 256 + threadIdx.x < n) atomicAdd(&h[(kk[k] >> shift) & 255], 1u);
    __syncthreads();
    tile_hist[(i64)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];       // tile-major: one coalesced KB per tile, here and in the scatter
}

// Exclusive scan of the tile histograms over the tiles, per digit, ONE launch per pass.  The histograms are TILE-major (256 counters of a
// tile side by side: every access of k_rs_hist, of this kernel and of the scatter is a coalesced KB; the digit-major layout of rounds 2-3
// made the scatter gather its 256 offsets from 256 different lines).  Workgroup c scans a chunk of `chunk_len` tiles in place (thread d
// walks column d, sixteen rows in flight) and files the chunk's totals write-through; the LAST workgroup to finish (ticket) turns the
// chunk totals into chunk_excl[c][d] and row_total[d].  The scatter adds tile_off[tile][d] + chunk_excl[tile / chunk_len][d].
constexpr int RSC_MAX_CHUNKS = 128;
constexpr int RSC_MIN_LEN = 16;
__global__ __launch_bounds__(256) void k_rs_scan_cols(u32 *__restrict__ tile_hist, i64 n_tiles, i64 chunk_len, u32 *__restrict__ chunk_tot,
                                                       u32 *__restrict__ chunk_excl, u32 *__restrict__ row_total, u32 *__restrict__ ticket) {
    __shared__ u32 s_last;
    const int d = threadIdx.x;
    const i64 t0 = (i64)blockIdx.x * chunk_len, t1 = t0 + chunk_len < n_tiles ? t0 + chunk_len : n_tiles;
    u32 run = 0;
    for (i64 t = t0; t < t1; t += 16) {
        u32 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = t + k < t1 ? tile_hist[(t + k) * 256 + d] : 0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (t + k < t1) tile_hist[(t + k) * 256 + d] = run;
            run += v[k];
        }
    }
    __hip_atomic_store(chunk_tot + (i64)blockIdx.x * 256 + d, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the write-through stores have left before the ticket is taken
    __syncthreads();
    if (d == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    u32 acc = 0;
    for (int c = 0; c < (int)gridDim.x; c += 32) {
        u32 v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k)
            v[k] = c + k < (int)gridDim.x ? __hip_atomic_load(chunk_tot + (i64)(c + k) * 256 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (c + k < (int)gridDim.x) chunk_excl[(i64)(c + k) * 256 + d] = acc;
            acc += v[k];
        }
    }
    row_total[d] = acc;
    if (d == 0) *ticket = 0;                                           // for the next pass (stream order)
}

template <bool HAS_VALS>
__global__ __launch_bounds__(256) void k_rs_scatter(const u64 *__restrict__ keys, const u32 *__restrict__ vals, i64 n, int shift,
                                                     i64 n_tiles, const u32 *__restrict__ tile_off /* [tile][256], scanned inside its chunk */,
                                                     const u32 *__restrict__ chunk_excl /* [chunk][256] */, i64 chunk_len,
                                                     const u32 *__restrict__ row_total /* [256]: keys per digit */,
                                                     u64 *__restrict__ out_keys, u32 *__restrict__ out_vals) {
    __shared__ u64 s_key[RS_TILE];
    __shared__ u32 s_val[HAS_VALS ? RS_TILE : 1];
    __shared__ u32 s_cnt[4][256];      // per-wave running digit counters, then per-wave exclusive offsets.  NOT volatile and not
                                       // through a pointer: both turn every access into a flat, system-scope load / store with a
                                       // wait behind it; wavefront-scope fences order the lanes' reads and writes instead
    __shared__ u32 s_dig_off[256];     // exclusive offset of each digit inside the tile
    __shared__ u32 s_gbase[256];       // global base of each digit for this tile
    __shared__ u32 s_wave[4];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware tile order: workgroups go round-robin over the 8 XCDs, and consecutive tiles append to the same 256 digit streams — the
    // runs of one tile end in the middle of cache lines that the next tile completes.  Every XCD therefore takes a CONTIGUOUS eighth of
    // the tiles, in order: the partial lines meet in ONE L2 instead of leaving two XCDs as masked partial writes.
    const i64 per_xcd = (n_tiles + 7) / 8;
    const i64 tile = (i64)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const i64 tile_base = tile * RS_TILE;
    const u32 goff = tile_off[tile * 256 + threadIdx.x] + chunk_excl[tile / chunk_len * 256 + threadIdx.x];     // issued first, used behind the key loads
    const u32 gtot = row_total[threadIdx.x];
    u64 key[RS_ITEMS];
    u32 val[RS_ITEMS];
    u32 pos[RS_ITEMS];
    const u64 lt_mask = (1ULL << lane) - 1ULL;
    // all sixteen loads in flight before the first key is used: the wavefront fences of the ranking loop otherwise pin every load
    // behind its own s_waitcnt vmcnt(0) — sixteen dependent round trips per tile (round 3: 331 -> see DESIGN §3.3 per pass at cfg3)
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const i64 idx = tile_base + (i64)wave * RS_WSEG + r * 64 + lane;
        const bool valid = idx < n;
        key[r] = valid ? keys[idx] : ~0ULL;
        val[r] = (HAS_VALS && valid) ? vals[idx] : 0u;
    }
    asm volatile("" : "+v"(key[0]), "+v"(key[1]), "+v"(key[2]), "+v"(key[3]), "+v"(key[4]), "+v"(key[5]), "+v"(key[6]), "+v"(key[7]));
    asm volatile("" : "+v"(key[8]), "+v"(key[9]), "+v"(key[10]), "+v"(key[11]), "+v"(key[12]), "+v"(key[13]), "+v"(key[14]), "+v"(key[15]));
    if (HAS_VALS) {
        asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]), "+v"(val[5]), "+v"(val[6]), "+v"(val[7]));
        asm volatile("" : "+v"(val[8]), "+v"(val[9]), "+v"(val[10]), "+v"(val[11]), "+v"(val[12]), "+v"(val[13]), "+v"(val[14]), "+v"(val[15]));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s_cnt[k][threadIdx.x] = 0;
    {   // global base of digit d for this tile = keys of smaller digits + keys of digit d in earlier tiles
        u32 all;
        s_gbase[threadIdx.x] = goff + block_excl_scan_256(gtot, s_wave, &all);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const i64 idx = tile_base + (i64)wave * RS_WSEG + r * 64 + lane;
        const bool valid = idx < n;
        const u32 d = (u32)(key[r] >> shift) & 255u;
        const u64 m = match_digit(d, valid);
        const u32 rank = __popcll(m & lt_mask);
        const u32 count = __popcll(m);
        u32 base = 0;
        if (valid) base = s_cnt[wave][d];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) s_cnt[wave][d] = base + count;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        pos[r] = base + rank;
    }
    __syncthreads();
    // per-digit: exclusive offsets over waves, tile digit counts, exclusive scan over digits
    {
        const int d = threadIdx.x;
        u32 run = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u32 c = s_cnt[k][d];
            s_cnt[k][d] = run;
            run += c;
        }
        u32 total;
        u32 excl = block_excl_scan_256(run, s_wave, &total);
        s_dig_off[d] = excl;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const i64 idx = tile_base + (i64)wave * RS_WSEG + r * 64 + lane;
        if (idx < n) {
            const u32 d = (u32)(key[r] >> shift) & 255u;
            const u32 slot = s_dig_off[d] + s_cnt[wave][d] + pos[r];
            s_key[slot] = key[r];
            if (HAS_VALS) s_val[slot] = val[r];
        }
    }
    __syncthreads();
    const i64 n_valid = (n - tile_base < RS_TILE) ? (n - tile_base) : RS_TILE;
#pragma unroll 4
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int s = k * 256 + threadIdx.x;
        if (s < n_valid) {
            const u64 kk = s_key[s];
            const u32 d = (u32)(kk >> shift) & 255u;
            const i64 g = (i64)s_gbase[d] + (s - (int)s_dig_off[d]);
            out_keys[g] = kk;
            if (HAS_VALS) out_vals[g] = s_val[s];
        }
    }
}

// Sort n pairs by key bits [begin_bit, end_bit) (multiple of 8 wide), stable.  Ping-pongs between the
// given buffers; *result_in_tmp tells where the sorted data ended up.  vals == nullptr: keys only.
static_assert(RS_TILE == SORT_TILE, "the tile size callers form their first histogram with");
int radix_sort_pairs_u64_u32(u64 *keys, u32 *vals, u64 *keys_tmp, u32 *vals_tmp, i64 n, int begin_bit, int end_bit,
                             bool *result_in_tmp, u32 *first_hist) {
    *result_in_tmp = false;
    if (n <= 1) return SYMGPU_OK;
    if (n >= ((i64)1 << 32)) {
        set_error("radix sort: n = %lld exceeds 2^32 - 1", (long long)n);
        return SYMGPU_E_INVALID;
    }
    hipStream_t st = ctx().stream;
    const i64 n_tiles = (n + RS_TILE - 1) / RS_TILE;
    Context &c = ctx();
    if (!c.sort_scan_ticket) {
        HIP_TRY(hipMalloc((void **)&c.sort_scan_ticket, 256));
        HIP_TRY(hipMemsetAsync(c.sort_scan_ticket, 0, 256, st));
    }
    const i64 chunk_len = (n_tiles + RSC_MAX_CHUNKS - 1) / RSC_MAX_CHUNKS > RSC_MIN_LEN ? (n_tiles + RSC_MAX_CHUNKS - 1) / RSC_MAX_CHUNKS : RSC_MIN_LEN;
    const unsigned n_chunks = (unsigned)((n_tiles + chunk_len - 1) / chunk_len);
    Scratch hist_own, scan;                                            // scan: row_total[256] | chunk_tot[n_chunks][256] | chunk_excl[n_chunks][256]
    if (!first_hist) SG_TRY(hist_own.alloc((size_t)n_tiles * 256 * sizeof(u32)));
    SG_TRY(scan.alloc((size_t)(1 + 2 * n_chunks) * 256 * sizeof(u32)));
    u32 *row_total = scan.as<u32>(), *chunk_tot = row_total + 256, *chunk_excl = chunk_tot + (size_t)n_chunks * 256;
    u32 *hist_p = first_hist ? first_hist : hist_own.as<u32>();
    u64 *ksrc = keys, *kdst = keys_tmp;
    u32 *vsrc = vals, *vdst = vals_tmp;
    bool in_tmp = false;
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        if (!(first_hist && shift == begin_bit)) {
            hipLaunchKernelGGL(k_rs_hist, dim3((unsigned)n_tiles), dim3(256), 0, st, ksrc, n, shift, n_tiles, hist_p);
            KERNEL_CHECK();
        }
        hipLaunchKernelGGL(k_rs_scan_cols, dim3(n_chunks), dim3(256), 0, st, hist_p, n_tiles, chunk_len, chunk_tot, chunk_excl, row_total, c.sort_scan_ticket);
        if (vals)
            hipLaunchKernelGGL(k_rs_scatter<true>, dim3((unsigned)((n_tiles + 7) / 8 * 8)), dim3(256), 0, st, ksrc, vsrc, n, shift, n_tiles, hist_p, chunk_excl,
                               chunk_len, row_total, kdst, vdst);
        else
            hipLaunchKernelGGL(k_rs_scatter<false>, dim3((unsigned)((n_tiles + 7) / 8 * 8)), dim3(256), 0, st, ksrc, (const u32 *)nullptr, n, shift, n_tiles,
                               hist_p, chunk_excl, chunk_len, row_total, kdst, (u32 *)nullptr);
        KERNEL_CHECK();
        u64 *tk = ksrc; ksrc = kdst; kdst = tk;
        u32 *tv = vsrc; vsrc = vdst; vdst = tv;
        in_tmp = !in_tmp;
    }
    *result_in_tmp = in_tmp;
    return SYMGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// The same sort as ONE persistent launch (keys only, up to COOP_MAX_TILES tiles): at 10^5 keys a pass of the multi-launch form is
// four or five launches of a few microseconds each — launch bound (26 us per pass); here the passes are separated by two counter
// barriers each.  Everything that crosses workgroups (keys, tile histograms, the barrier counter) moves through agent-scope
// relaxed loads and stores (write-through / L1-bypassing: cdna_hip_programming.md Guideline 16), so the barriers need no fences.
// The workgroups must be co-resident: <= COOP_MAX_TILES blocks of 256 threads and 35 KB of LDS.
// ------------------------------------------------------------------------------------------------
constexpr int COOP_MAX_TILES = 128;
// where the one launch beats a launch per step (round 6, P * P and plain cleanups of 100 qubits): keys alone up to ~2e5 (245,350 keys: 182
// against 178 us; 125,250: 158 against 184), keys with values up to ~1.3e5 (100,000 rows: 116 against 121 us; 200,000: 140 against 127)
constexpr int COOP_TILES_KEYS = 48, COOP_TILES_PAIRS = 32;

// returns false if the other workgroups did not arrive within ~2^22 polls (not co-resident): the caller's kernel gives up, bar[1] says so
__device__ __forceinline__ bool coop_barrier(u32 *bar, u32 target, int G, u32 *s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // this wavefront's write-through stores have left
    __syncthreads();
    if (G > 1 && threadIdx.x == 0) {
        atomicAdd(bar, 1u);
        u32 ok = 1;
        for (u32 spin = 0; (i32)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0; ++spin) {
            if (spin >= (1u << 22) || __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = ok;
    }
    __syncthreads();
    return G == 1 || *s_flag != 0;
}

// HAS_VALS: a 32-bit value travels with every key (round 6: the index sort of a plain cleanup — `A + B`, `cleanup()` — of up to 5e5 rows:
// three passes were nine launches, launch bound)
template <bool HAS_VALS>
__global__ __launch_bounds__(256) void k_rs_coop(u64 *__restrict__ buf_a, u64 *__restrict__ buf_b, u32 *__restrict__ val_a, u32 *__restrict__ val_b, i64 n, int begin_bit,
                                                  int n_passes, u32 *__restrict__ tile_hist /* [256][G] */, u32 *__restrict__ bar, u32 bar_base) {
    __shared__ u64 s_key[RS_TILE];
    __shared__ u32 s_val[HAS_VALS ? RS_TILE : 1];
    __shared__ u32 s_cnt[4][256];                           // see k_rs_scatter: plain LDS accesses ordered by wavefront-scope fences
    __shared__ u32 s_dig_off[256];
    __shared__ u32 s_gbase[256];
    __shared__ u32 s_wave[4];
    __shared__ u32 s_ok;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = gridDim.x, tile = blockIdx.x;
    const i64 tile_base = (i64)tile * RS_TILE;
    const u64 lt_mask = (1ULL << lane) - 1ULL;
    u32 n_bar = 0;
    for (int p = 0; p < n_passes; ++p) {
        const int shift = begin_bit + 8 * p;
        const u64 *src = (p & 1) ? buf_b : buf_a;
        u64 *dst = (p & 1) ? buf_a : buf_b;
        const u32 *vsrc = (p & 1) ? val_b : val_a;
        u32 *vdst = (p & 1) ? val_a : val_b;
#pragma unroll
        for (int k = 0; k < 4; ++k) s_cnt[k][threadIdx.x] = 0;
        __syncthreads();
        u64 key[RS_ITEMS];
        u32 pos[RS_ITEMS], val[HAS_VALS ? RS_ITEMS : 1];
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {                                   // all sixteen loads in flight before the first one is used
            const i64 idx = tile_base + (i64)wave * RS_WSEG + r * 64 + lane;
            key[r] = idx < n ? __hip_atomic_load(src + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ULL;
            if (HAS_VALS) val[r] = idx < n ? __hip_atomic_load(vsrc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
        asm volatile("" : "+v"(key[0]), "+v"(key[1]), "+v"(key[2]), "+v"(key[3]), "+v"(key[4]), "+v"(key[5]), "+v"(key[6]), "+v"(key[7]));
        asm volatile("" : "+v"(key[8]), "+v"(key[9]), "+v"(key[10]), "+v"(key[11]), "+v"(key[12]), "+v"(key[13]), "+v"(key[14]), "+v"(key[15]));
        static_assert(RS_ITEMS == 16, "the two statements above name sixteen keys");
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            const i64 idx = tile_base + (i64)wave * RS_WSEG + r * 64 + lane;
            const bool valid = idx < n;
            const u32 d = (u32)(key[r] >> shift) & 255u;
            const u64 m = match_digit(d, valid);
            const u32 rank = __popcll(m & lt_mask);
            const u32 count = __popcll(m);
            u32 base = 0;
            if (valid) base = s_cnt[wave][d];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) s_cnt[wave][d] = base + count;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            pos[r] = base + rank;
        }
        __syncthreads();
        u32 my_count;
        {   // per digit: exclusive offsets over the wavefronts, the tile's count, exclusive scan over the digits
            const int d = threadIdx.x;
            u32 run = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const u32 c = s_cnt[k][d]; s_cnt[k][d] = run; run += c; }
            my_count = run;
            u32 total;
            s_dig_off[d] = block_excl_scan_256(run, s_wave, &total);
            __hip_atomic_store(&tile_hist[(i64)d * G + tile], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!coop_barrier(bar, bar_base + (u32)G * (++n_bar), G, &s_ok)) return;
        {   // global base of every digit for this tile: keys of smaller digits (all tiles) + keys of this digit in earlier tiles
            const int d = threadIdx.x;
            u32 all = 0, before = 0;
            if (G == 1) all = my_count;
            else
                for (int t0 = 0; t0 < G; t0 += 16) {                           // sixteen independent loads per round
                    u32 x[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        x[k] = t0 + k < G ? __hip_atomic_load(&tile_hist[(i64)d * G + t0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
                    asm volatile("" : "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
#pragma unroll
                    for (int k = 0; k < 16; ++k) { all += x[k]; if (t0 + k < tile) before += x[k]; }
                }
            u32 total;
            s_gbase[d] = block_excl_scan_256(all, s_wave, &total) + before;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            const i64 idx = tile_base + (i64)wave * RS_WSEG + r * 64 + lane;
            if (idx < n) {
                const u32 d = (u32)(key[r] >> shift) & 255u;
                s_key[s_dig_off[d] + s_cnt[wave][d] + pos[r]] = key[r];
                if (HAS_VALS) s_val[s_dig_off[d] + s_cnt[wave][d] + pos[r]] = val[r];
            }
        }
        __syncthreads();
        const i64 n_valid = (n - tile_base < RS_TILE) ? (n - tile_base) : RS_TILE;
#pragma unroll 4
        for (int k = 0; k < RS_ITEMS; ++k) {
            const int sidx = k * 256 + threadIdx.x;
            if (sidx < n_valid) {
                const u64 kk = s_key[sidx];
                const u32 d = (u32)(kk >> shift) & 255u;
                __hip_atomic_store(dst + ((i64)s_gbase[d] + (sidx - (int)s_dig_off[d])), kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (HAS_VALS) __hip_atomic_store(vdst + ((i64)s_gbase[d] + (sidx - (int)s_dig_off[d])), s_val[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!coop_barrier(bar, bar_base + (u32)G * (++n_bar), G, &s_ok)) return;
    }
}

// returns SYMGPU_OK with *done = false if the array is too large for the one-launch form
static int radix_sort_coop_launch(u64 *keys, u32 *vals, u64 *keys_tmp, u32 *vals_tmp, i64 n, int begin_bit, int end_bit, bool *result_in_tmp, bool *done, bool auto_size) {
    *done = false;
    *result_in_tmp = false;
    if (n <= 1) { *done = true; return SYMGPU_OK; }
    const i64 n_tiles = (n + RS_TILE - 1) / RS_TILE;
    if (n_tiles > COOP_MAX_TILES || (auto_size && n_tiles > (vals ? COOP_TILES_PAIRS : COOP_TILES_KEYS))) return SYMGPU_OK;
    if (const char *e = SG_TUNE("SYMGPU_SORT_COOP")) if (e[0] == '0') return SYMGPU_OK;
    Context &c = ctx();
    if (c.sort_coop_disabled) return SYMGPU_OK;
    hipStream_t st = c.stream;
    // c.sort_state: [0] barrier counter (monotonic over the launches; cleared before it can wrap), [1] time-out flag, [64 ..] tile histograms
    if (!c.sort_state) {
        HIP_TRY(hipMalloc((void **)&c.sort_state, (64 + 256 * COOP_MAX_TILES) * sizeof(u32)));
        HIP_TRY(hipMemsetAsync(c.sort_state, 0, 64 * sizeof(u32), st));
        c.sort_bar_base = 0;
    }
    const int n_passes = (end_bit - begin_bit + 7) / 8;
    const u32 arrivals = (u32)n_tiles * 2u * (u32)n_passes;
    if (c.sort_bar_base > 0x70000000u - arrivals) {
        HIP_TRY(hipMemsetAsync(c.sort_state, 0, 64 * sizeof(u32), st));
        c.sort_bar_base = 0;
    }
    if (vals)
        hipLaunchKernelGGL((k_rs_coop<true>), dim3((unsigned)n_tiles), dim3(256), 0, st, keys, keys_tmp, vals, vals_tmp, n, begin_bit, n_passes, c.sort_state + 64, c.sort_state,
                           c.sort_bar_base);
    else
        hipLaunchKernelGGL((k_rs_coop<false>), dim3((unsigned)n_tiles), dim3(256), 0, st, keys, keys_tmp, (u32 *)nullptr, (u32 *)nullptr, n, begin_bit, n_passes, c.sort_state + 64,
                           c.sort_state, c.sort_bar_base);
    KERNEL_CHECK();
    c.sort_bar_base += n_tiles > 1 ? arrivals : 0u;
    *result_in_tmp = (n_passes & 1) != 0;
    *done = true;
    return SYMGPU_OK;
}
int radix_sort_keys_u64_coop(u64 *keys, u64 *keys_tmp, i64 n, int begin_bit, int end_bit, bool *result_in_tmp, bool *done) {
    return radix_sort_coop_launch(keys, nullptr, keys_tmp, nullptr, n, begin_bit, end_bit, result_in_tmp, done, false);
}
// ... only where it beats the launches per step (small products: the caller has no better use for the answer than "sort these keys")
int radix_sort_keys_u64_small(u64 *keys, u64 *keys_tmp, i64 n, int begin_bit, int end_bit, bool *result_in_tmp, bool *done) {
    return radix_sort_coop_launch(keys, nullptr, keys_tmp, nullptr, n, begin_bit, end_bit, result_in_tmp, done, true);
}
// the same with a 32-bit value per key (plain cleanups of up to COOP_MAX_TILES tiles)
int radix_sort_pairs_u64_u32_coop(u64 *keys, u32 *vals, u64 *keys_tmp, u32 *vals_tmp, i64 n, int begin_bit, int end_bit, bool *result_in_tmp, bool *done) {
    return radix_sort_coop_launch(keys, vals, keys_tmp, vals_tmp, n, begin_bit, end_bit, result_in_tmp, done, true);
}

// After the stream has been synchronised: did a one-launch sort since the last check give up at a barrier?  Then its output is garbage;
// the form is switched off for the rest of the process and the caller reports the failure.
// the device word a one-launch sort raises when it gives up (null: no such sort has run), and what to do with its value once it is on the
// host — for callers that read it back together with their own status words
const u32 *radix_sort_coop_flag() {
    Context &c = ctx();
    return (!c.sort_state || c.sort_coop_disabled) ? nullptr : c.sort_state + 1;
}
void radix_sort_coop_note(u32 flag, bool *timed_out) {
    *timed_out = false;
    if (!flag) return;
    *timed_out = true;
    ctx().sort_coop_disabled = true;
    note_degraded("one-launch radix sort (k_rs_coop) off: an in-kernel barrier timed out (workgroups not co-resident?); sorts take one launch per pass");
}
int radix_sort_coop_check(bool *timed_out) {
    *timed_out = false;
    const u32 *p = radix_sort_coop_flag();
    if (!p) return SYMGPU_OK;
    u32 flag = 0;
    HIP_TRY(hipMemcpy(&flag, p, 4, hipMemcpyDeviceToHost));
    radix_sort_coop_note(flag, timed_out);
    return SYMGPU_OK;
}

int radix_sort_keys_u64(u64 *keys, u64 *keys_tmp, i64 n, int begin_bit, int end_bit, bool *result_in_tmp, u32 *first_hist) {
    return radix_sort_pairs_u64_u32(keys, nullptr, keys_tmp, nullptr, n, begin_bit, end_bit, result_in_tmp, first_hist);
}

}  // namespace symgpu
