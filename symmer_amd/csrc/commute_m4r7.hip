// commute_m4r7.hip — the Four-Russians commutation kernel: TWO 7-bit tables per step folded with v_bitop3 (round 5), run by persistent
// workgroups over the (tile, step) space (round 6).
// (reference: symmer/operators/base.py:938-971 -> matmul_GF2 / numba_dot_matmal_GF2, utils.py:9-78: f64 dgemm, then % 2)
//
// A step covers 14 contraction bits: two 128-entry tables of XOR-combinations of B's bit-rows (2 x 32 KiB, double buffered: 128 KiB of
// LDS); a row of A reads one 256-byte entry of EACH (a lane: 16 bytes, one ds_read_b128) and folds both into its accumulator with one
// v_bitop3 (a ^ b ^ c) per dword: 2 reads + 2 v_perm + 4 v_bitop3 per row and step.  A workgroup of 8 waves holds the accumulators of
// 1,536 rows x 2,048 columns in registers (R = 48 rows per 16-lane slot: 192 of a wave's 256 VGPRs).
//
// Round 6 (200,000^2 terms at n = 2000, one launch: 37.1 -> 31.2 ms; one rank's 25,000-row share of the 8-GPU run: 5.5 -> 4.2 ms):
//   * stream-K: a launch is num_cu persistent workgroups, each owning a contiguous range of the (tile, step) space — the same number of
//     steps (+-1) whatever the tile count (1,666 tiles = 6.51 per CU cost 7 rounds as one-tile workgroups).  A range starts and ends
//     inside a tile.  The owner of a tile's LAST steps meets it as its first job and leaves the raw accumulators in scratch; the owner of
//     its FIRST steps meets it as its last job, adds the neighbour's part (published a tile time or more ago) and writes the tile.  Nobody
//     waits: a part that is not published yet (workgroups queued behind each other on a shared GPU) sends the tile to k_m7_fixup.
//     The ranges start at different steps of their tiles, so the 3 MB tile epilogues no longer arrive from all CUs at once.
//   * index bytes come from global memory into registers (vmcnt; chunks of CH rows, NS register sets in flight) instead of through LDS: no
//     LDS reads for them, no dependent LDS round trip at the start of a step, and ONE rolling window of table reads through the whole
//     step.  Measured with s_memtime stamps (-DSYMGPU_M7_STAMPS): the look-ups of a step take 2,900-3,200 cycles for 768 wave reads =
//     the LDS pipe's 4 cycles per ds_read_b128.
//   * waves 4..7 (one per SIMD) build the next tables BEFORE their look-ups, waves 0..3 after: a step no longer starts with all eight waves
//     waiting for their first table entries at the same moment.
//   * tables are written with ds_write_addtid_b32 (a wave stores one 256-byte entry per instruction, no address register).
//   * every scalar of a step (index offsets, BT row offsets) comes from a table computed once (k_m7_klist) and is fetched one step
//     ahead; the step's global requests are issued before the barrier that precedes it.  (Round 5 recomputed them at the top of every
//     step behind four scalar loads: ~800 cycles of a 6,000-cycle step.)
// What a step still costs beyond its 3,072 cycles of table reads: the builds (a wave's 32 entries take ~800 cycles whatever the chain
// structure: one wave issues a DS instruction every ~25 cycles next to four streaming waves) and ~300 cycles around the barrier.
//
// Measured and dropped (round 5): look-ups and the next table's build as one interleaved instruction stream; a staggered start of every
// CU's first workgroup; the index dwords in a rolling window of their own through LDS.  Round 6: uneven build shares between the two wave
// groups (200,000^2: 31.9 ms even; 33.1 / 34.2 / 36.0 ms with the early builders at 3 : 5 / 2 : 6 / 1 : 7 of a table, 33.0 / 33.5 / 34.9 ms
// the other way round); all waves building after their look-ups (31.5 against 31.3 ms); the tables written four entries at a time with
// ds_write_b128 (eight stores per wave and step instead of 32: 34.0 against 30.4 ms — the wide stores and the 16-byte reads of the bit rows
// keep the pipe from the look-ups: builds 960-1,200 cycles against ~800, look-ups 3,100-3,560 against 2,900-3,200).
//
// Layouts prepared per call:
//   A7[g][i]   bits [7g, 7g + 7) of packed row i (values 0..127), group-major, i zero padded to Npad; one more all-zero group at index
//              NG7 pads an odd number of non-zero groups to whole pairs (entry 0 of any table is the XOR of no rows = 0).
//   BT[c][jw]  bit c of B rows 64jw..64jw+63 (the bit-major copy of commute_m4r.hip, cached on the operator); contraction bit c of A pairs
//              with row c + 64Wq (c in the X half) or c - 64Wq (Z half).
//   klist      the groups in which A has any non-zero value, ascending, padded to an even count with the zero group.
//   steptab    per pair of groups: index offsets and BT row offsets (second half of k_m7_klist).
#include "common.h"
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

namespace symgpu {

typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

constexpr int M7_TILE_W = 32;                          // 64-bit words per column tile: 2048 columns
constexpr int M7_ENTRY_BYTES = M7_TILE_W * 8;          // 256
constexpr int M7_STREAM_MIN_WORK = 125;                // tile-steps per persistent workgroup from which it pays
constexpr int M7_STREAM_MIN_STEPS = 32;                // steps a tile (pairs of 7-bit groups of the padded row: 28 at n <= 192, 37 above) from which the stream-K launch pays
constexpr int M7_TABLE_BYTES = 128 * M7_ENTRY_BYTES;   // 32 KiB: one 7-bit group
constexpr int M7_BUF_BYTES = 2 * M7_TABLE_BYTES;       // the two tables of a step
constexpr int M7_LDS = 2 * M7_BUF_BYTES;               // double buffered: 128 KiB
constexpr int M7_BT_STAGE = 14 * M7_ENTRY_BYTES;       // the 14 bit-rows of a step: 3.5 KiB
constexpr int M7_WAVES = 8;

__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// group-major copy of A (zero padded to Npad rows, plus the all-zero group NG7) + "group has a non-zero value" flags.
// A workgroup turns 256 rows x 16 words (+ 1: a 7-bit group may straddle a word) round through LDS: rows arrive as contiguous 136-byte
// pieces, a thread then owns one row and every group is stored as 256 consecutive bytes.  (Round 5 read each row word by word from its own
// thread: 0.14 ms for 25,000 rows of 64 words — 3 % of one rank's share of the 8-GPU adjacency run; now 0.02 ms.)
constexpr int A7_CW = 16;                                             // words per chunk
__global__ __launch_bounds__(256) void k_m7_a7(const u64 *__restrict__ rows, i64 N, int W, int ng7, uint8_t *__restrict__ A7, i64 Npad, u32 *__restrict__ flags) {
    __shared__ u64 tile[A7_CW + 1][257];                              // [word][row], pitch 257: the transposing writes spread over the banks
    const i64 i0 = (i64)blockIdx.x * 256;
    const int c0 = blockIdx.y * A7_CW;                                // first word of this chunk
    const int nw = (W - c0 < A7_CW + 1) ? W - c0 : A7_CW + 1;         // words present (the extra one only if it exists)
    const int lane = threadIdx.x & 63;
    for (int idx = threadIdx.x; idx < 256 * (A7_CW + 1); idx += 256) {
        const int r = idx / (A7_CW + 1), k = idx - r * (A7_CW + 1);
        tile[k][r] = (k < nw && i0 + r < N) ? rows[(i0 + r) * W + c0 + k] : 0ULL;
    }
    __syncthreads();
    const int g0 = (64 * c0 + 6) / 7;                                 // groups whose first bit lies in this chunk
    const int g1 = (64 * (c0 + A7_CW) + 6) / 7 < ng7 ? (64 * (c0 + A7_CW) + 6) / 7 : ng7;
    const int r = threadIdx.x;
    for (int g = g0; g < g1; ++g) {
        const int bit0 = 7 * g - 64 * c0, w = bit0 >> 6, sh = bit0 & 63;
        u64 v = tile[w][r] >> sh;
        if (sh > 57) v |= tile[w + 1][r] << (64 - sh);
        const u32 val = (u32)v & 0x7Fu;
        A7[(i64)g * Npad + i0 + r] = (uint8_t)val;
        const u64 any = __ballot(val != 0);
        if (any && lane == 0) flags[g] = 1u;                          // benign race: every writer stores 1
    }
    if (blockIdx.y == 0) A7[(i64)ng7 * Npad + i0 + r] = 0;            // the padding group
}

// compact the flagged groups (single wave; ascending) and pad to an even count with the zero group; then the per-step scalars of the
// stream-K kernel, computed once: entry p (16 u64) = { byte offset of group ga's index bytes in A7, same for gb, then for w = 0..6 the word
// offsets of the BT rows that bit w of ga / of gb pairs with }.  Two entries behind the last pair point at the zero group / row 0, so the
// kernel fetches entries t + 1 and t + 2 without asking whether they exist.
__global__ __launch_bounds__(64) void k_m7_klist(const u32 *__restrict__ flags, int ng7, u32 *__restrict__ klist, u32 *__restrict__ n_pairs, int max_pairs, i64 Npad,
                                                 i64 Mw_pad, int Wq, u64 *__restrict__ tab) {
    const int lane = threadIdx.x;
    u32 count = 0;
    for (int base = 0; base < ng7; base += 64) {
        const int g = base + lane;
        const bool on = g < ng7 && flags[g] != 0;
        const u64 m = __ballot(on);
        if (on) klist[count + __popcll(m & ((1ULL << lane) - 1))] = (u32)g;
        count += (u32)__popcll(m);
    }
    if (lane == 0) {
        if (count & 1u) klist[count] = (u32)ng7;
        if (count == 0) { klist[0] = klist[1] = (u32)ng7; count = 1; }   // an all-identity left operand: one step on the zero group (every workgroup owns >= 1 step)
        *n_pairs = (count + 1) / 2;
    }
    __threadfence_block();
    __syncthreads();                                                  // (one wave: orders the klist stores above before the loads below)
    const u32 np = (count + 1) / 2 + (count == 0 ? 1u : 0u);
    const i64 half_bits = (i64)64 * Wq;
    auto bt_row = [&](u32 g, int r) -> i64 {                          // BT row that contraction bit 7g + r of A pairs with
        const i64 c = 7 * (i64)g + r;
        return c < half_bits ? c + half_bits : (c < 2 * half_bits ? c - half_bits : 0);   // (padding bits of the last group: A has zeros there)
    };
    for (int p = lane; p < max_pairs + 2; p += 64) {
        const bool live = (u32)p < np;
        const u32 ga = live ? klist[2 * p] : (u32)ng7, gb = live ? klist[2 * p + 1] : (u32)ng7;
        u64 *e = tab + (i64)p * 16;
        e[0] = (u64)ga * (u64)Npad;
        e[1] = (u64)gb * (u64)Npad;
        for (int w = 0; w < 7; ++w) {
            e[2 + 2 * w] = (u64)(bt_row(ga, w) * Mw_pad);
            e[3 + 2 * w] = (u64)(bt_row(gb, w) * Mw_pad);
        }
    }
}

constexpr int M7S_LDS = M7_LDS + 2 * M7_BT_STAGE;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// inline-asm pieces as functions (operands of an asm statement inside a generic lambda are not captured by this compiler)
template <int OFF> __device__ __forceinline__ void m7_gload(u32 &d, u32 voff, u64 sbase) { asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF)); }
// a base that may have just left a v_readfirstlane: VALU-written SGPR -> VMEM address needs 5 wait states, which nobody inserts inside asm
template <int OFF> __device__ __forceinline__ void m7_gload_fresh(u32 &d, u32 voff, u64 sbase) { asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF)); }
template <int N> __device__ __forceinline__ void m7_vmwait(u32 &a, u32 &b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int N> __device__ __forceinline__ void m7_vmwait(u32 &a, u32 &b, u32 &c, u32 &d) { asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N)); }
template <int N> __device__ __forceinline__ void m7_ldswait(u32x4 &a, u32x4 &b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int OFF> __device__ __forceinline__ void m7_write_addtid(u32 e) { asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(e), "n"(OFF) : "memory"); }
__device__ __forceinline__ void m7_read2(u32x4 &a, u32x4 &c, u32 addr_a, u32 addr_b) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(a) : "v"(addr_a));
    asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(c) : "v"(addr_b));
}
__device__ __forceinline__ void m7_pin(u32x4 &x) { asm volatile("" : "+v"(x)); }
// 16 np.bool_ bytes of a row: one non-temporal 16-byte store wherever the row length and the base put it (an unaligned store costs a second
// transaction where it straddles a cache line, not correctness), the columns past the end of the row byte by byte.  Round 6: rows that
// are not a multiple of 16 bytes used to leave the kernel bit-packed for a second, byte-by-byte pass (100,008^2: 13 ms against 2.6 ms).
__device__ __forceinline__ void m7_store16(uint8_t *dst, u32x4 v, i64 cols_left) {
    typedef u32x4 u32x4_any __attribute__((aligned(1)));
    if (cols_left >= 16) {
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4_any *>(dst));
    } else {
        for (int b = 0; b < (int)cols_left; ++b) dst[b] = (uint8_t)(v[b >> 2] >> (8 * (b & 3)));
    }
}
// a lane index the optimiser cannot see through: what is computed from it stays where it is used (the epilogues' per-lane constants would
// otherwise be hoisted out of the job loop and held in registers through the look-up stream, which has none to spare)
__device__ __forceinline__ int m7_opaque(int x) { asm volatile("" : "+v"(x)); return x; }

template <int R, int LOOKP, int CH, int NS>
__global__ __launch_bounds__(64 * M7_WAVES) void k_commutes_m4r7s(const uint8_t *__restrict__ A7, i64 Npad, i64 N, const u64 *__restrict__ BT, i64 Mw_pad,
                                                                   const u64 *__restrict__ steptab, const u32 *__restrict__ np_ptr,
                                                                   void *__restrict__ out_v, i64 out_stride, i64 m_cols, int bytes,
                                                                   u64 *__restrict__ part, i64 n_rt, i64 n_tiles, int stream, int force_fixup, u64 *dbg, u32 *__restrict__ flags, u32 epoch) {
    constexpr int WAVES = M7_WAVES;
    constexpr int WG_ROWS = 4 * WAVES * R;
    constexpr int NCH = R / CH, QC = CH / 4, LPC = 2 * QC;             // chunks per step; index dwords per chunk and table; loads per chunk
    static_assert(R % CH == 0 && CH % 4 == 0 && NCH % NS == 0 && NS >= 2 && LOOKP <= CH && LOOKP <= 6 && QC <= 2, "chunk c of every step lives in register set c % NS");
    constexpr i64 PART_WORDS = (i64)WG_ROWS * M7_TILE_W;
    extern __shared__ __attribute__((aligned(16))) uint8_t m7_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = lane >> 4;
    const u32 S = *np_ptr;
    // the look-up address is assembled bytewise (v_perm_b32), which needs the tables at LDS offset 0: m7_lds is the kernel's only LDS object
    if ((u32)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)m7_lds != 0) __builtin_trap();
    uint8_t *const bt_stage = m7_lds + M7_LDS;                        // [2][14 rows][256 B]
    const bool build_first = wave >= 4;                                // uniform

    i64 s_lo, s_hi;
    if (stream) {
        const i64 L = n_tiles * (i64)S;
        s_lo = (i64)blockIdx.x * L / gridDim.x;
        s_hi = ((i64)blockIdx.x + 1) * L / gridDim.x;
    } else {
        s_lo = (i64)blockIdx.x * S;
        s_hi = s_lo + S;
    }

    auto scalar_u64 = [](u64 a) -> u64 {                              // uniform by construction; tell the register allocator
        const u32 lo = __builtin_amdgcn_readfirstlane((u32)a), hi = __builtin_amdgcn_readfirstlane((u32)(a >> 32));
        return ((u64)hi << 32) | lo;
    };
    typedef u64 u64x2s __attribute__((ext_vector_type(2)));
    const u64x2s *const tab2 = reinterpret_cast<const u64x2s *>(steptab);   // entry p = tab2[8 p .. 8 p + 7]: {a7 offsets}, then {BT offsets of row w} for w = 0..6
    const int wrow = wave < 7 ? wave : 6;                                    // (wave 7 stages nothing; it reads wave 6's pair to stay in bounds)

    for (i64 cur = s_lo; cur < s_hi;) {
        const i64 tile = cur / S;
        const u32 k0 = (u32)(cur - tile * S);
        const u32 k1 = (s_hi - cur < (i64)(S - k0)) ? k0 + (u32)(s_hi - cur) : S;
        cur += k1 - k0;
        const i64 rt = tile % n_rt, ct = tile / n_rt;
        const i64 tile_row0 = rt * WG_ROWS;
        const i64 tile_w0 = ct * M7_TILE_W;
        const u32 voff = (u32)(tile_row0 + (i64)wave * (4 * R) + slot * R);   // this slot's first row (A7 holds a byte per row)

        u32x4 acc[R];                                                  // 128 result bits of a row: four consecutive registers
#pragma unroll
        for (int j = 0; j < R; ++j) acc[j] = u32x4{0, 0, 0, 0};
        u32 ia[NS][QC], ib[NS][QC];

        // staging: wave w < 7 fetches row w of both tables' seven BT rows, a dword per lane (the row is uniform: scalar base + lane offset).
        // The loads are asm like the index loads (the compiler would add a 64-bit per-lane base and drain vmcnt around them); they are the
        // oldest loads of their step, so they have landed once at most the (NS - 1) LPC index loads requested last are outstanding.
        auto stage_load = [&](u32 (&v)[2], u64x2s bt_off) {            // (unconditional: wave 7 and steps past the end read rows that exist)
            const u32 lane4 = (u32)m7_opaque(lane) * 4;
            m7_gload_fresh<0>(v[0], lane4, scalar_u64(reinterpret_cast<u64>(BT + bt_off.x + tile_w0)));
            m7_gload_fresh<0>(v[1], lane4, scalar_u64(reinterpret_cast<u64>(BT + bt_off.y + tile_w0)));
        };
        // (the wait is unconditional: a register whose load is still in flight must not look dead to the compiler, which would hand it out
        // as a temporary and have the returning load overwrite that)
        auto stage_store = [&](u32 (&v)[2], u32 bt_slot, bool have_bt, auto in_loop) {
            if constexpr (decltype(in_loop)::value) m7_vmwait<(NS - 1) * LPC>(v[0], v[1]);
            else m7_vmwait<0>(v[0], v[1]);                            // job prologue: these loads are the youngest
            if (wave < 7 && have_bt) {
                const u32 lane4 = (u32)m7_opaque(lane) * 4;
                u32 *d = reinterpret_cast<u32 *>(bt_stage + bt_slot * M7_BT_STAGE + wave * 256 + lane4);
                d[0] = v[0];
                d[7 * 64] = v[1];
            }
        };
        // Table builder: waves 0-3 table A, 4-7 table B; a wave writes the 32 entries (wave & 3) * 32 + g in Gray order; a lane holds dword
        // `lane` of an entry and a wave stores one whole entry per ds_write_addtid_b32 (LDS address = M0[15:0] + 16-bit immediate + 4 lane:
        // the run-time part — table, quarter: <= 56 KiB — goes into M0; the second buffer, + 64 KiB, is reached with M0 + 8188 and an
        // immediate of 57348 + 256 g <= 65,284).  Measured: giving the waves that build before their look-ups fewer (or more) entries than the
        // others is slower in both directions (200,000^2: 31.9 ms even, 33.1 / 34.2 / 36.0 ms at 3 : 5 / 2 : 6 / 1 : 7, 33.0 / 33.5 / 34.9 ms
        // the other way round) — a wave's build is a dependent chain, eight even shares are the shortest.
        const int tb = wave >> 2;
        auto build = [&](u32 buf, u32 bt_slot) {
            u32 brow[7];
            const u32 lane4 = (u32)m7_opaque(lane) * 4;
#pragma unroll
            for (int r = 0; r < 7; ++r) brow[r] = *reinterpret_cast<const u32 *>(bt_stage + bt_slot * M7_BT_STAGE + (tb * 7 + r) * 256 + lane4);
            const u32 hi = (u32)(wave & 3);
            // four independent chains (entry bits 3, 4 = chain), interleaved: a single chain of 32 dependent xor -> store steps took 23 cycles
            // per entry from one wave
            u32 e[4];
            e[0] = ((hi & 1u) ? brow[5] : 0u) ^ ((hi & 2u) ? brow[6] : 0u);
            e[1] = e[0] ^ brow[3];
            e[2] = e[0] ^ brow[4];
            e[3] = e[1] ^ brow[4];
            const u32 rt_part = (u32)tb * M7_TABLE_BYTES + hi * (32 * M7_ENTRY_BYTES);
            if (buf == 0) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 1" ::"s"(rt_part));
                static_for<0, 32>([&](auto sc) {
                    constexpr int s = decltype(sc)::value / 4, c = decltype(sc)::value % 4;
                    if constexpr (s > 0) e[c] = (u32)m7_opaque((int)(e[c] ^ brow[__builtin_ctz(s)]));   // (opaque: the chains are not to be turned into a tree of 32 live values)
                    m7_write_addtid<(c * 8 + (s ^ (s >> 1))) * M7_ENTRY_BYTES>(e[c]);
                });
            } else {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 1" ::"s"(rt_part + 8188u));
                static_for<0, 32>([&](auto sc) {
                    constexpr int s = decltype(sc)::value / 4, c = decltype(sc)::value % 4;
                    if constexpr (s > 0) e[c] = (u32)m7_opaque((int)(e[c] ^ brow[__builtin_ctz(s)]));
                    m7_write_addtid<M7_BUF_BYTES - 8188 + (c * 8 + (s ^ (s >> 1))) * M7_ENTRY_BYTES>(e[c]);
                });
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the barrier that follows does not know about these stores
        };
        // index bytes of chunk c (rows c CH .. c CH + CH - 1 of every slot) of the step whose groups sit at sa / sb -> register set c % NS
        auto prefetch = [&](auto cc, u64 sa, u64 sb) {
            constexpr int c = decltype(cc)::value;
            static_for<0, QC>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                m7_gload<c * CH + 4 * q>(ia[c % NS][q], voff, sa);
                m7_gload<c * CH + 4 * q>(ib[c % NS][q], voff, sb);
            });
        };
        auto prefetch_fresh = [&](auto cc, u64 sa, u64 sb) {           // the job's first requests: the bases have just been computed
            constexpr int c = decltype(cc)::value;
            static_for<0, QC>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                m7_gload_fresh<c * CH + 4 * q>(ia[c % NS][q], voff, sa);
                m7_gload_fresh<c * CH + 4 * q>(ib[c % NS][q], voff, sb);
            });
        };
        // chunk k has landed when at most the NS - 2 chunks requested after it are outstanding (loads return in order).  The count holds
        // in every step because every step requests its successor's first chunks — the last step of a job requests its own again (the
        // loads are drained behind the loop).  There is ONE wait per chunk on purpose: a wait on either side of a branch makes the
        // compiler merge the register sets with v_mov copies placed in front of the wait, i.e. copies of registers whose loads are
        // still in flight.  For the same reason no load sits inside a branch: a wave that has nothing to fetch reads a valid dummy.
        auto chunk_wait = [&](auto kc) {
            constexpr int set = decltype(kc)::value % NS;
            if constexpr (QC == 2) m7_vmwait<(NS - 2) * LPC>(ia[set][0], ia[set][1], ib[set][0], ib[set][1]);
            else m7_vmwait<(NS - 2) * LPC>(ia[set][0], ib[set][0]);
        };
        // One step's look-ups.  The reads and the waits for them are written by hand: LDS operations complete in order, so "at most
        // 2 (LOOKP - 1) outstanding" means the reads of the row being folded have landed; anything the compiler issues itself around this
        // block only makes the waits stricter.  E(k): wait for chunk k's index bytes, then request chunk k + NS - 1 (of this step, or of the
        // next one — its set was last used by chunk k - 1, whose reads have all been issued).
        auto lookups = [&](u32 buf, u64 sa, u64 sb, u64 sa_next, u64 sb_next) {
            const u32 base = (((u32)m7_opaque(lane) & 15u) << 4) | (buf << 16);
            auto issue = [&](auto jc, u32x4 &a, u32x4 &c) {
                constexpr int j = decltype(jc)::value, set = (j / CH) % NS, q = (j % CH) / 4;
                // address byte 0 = lane offset, byte 1 = table index (entry stride 256 B), byte 2 = buffer, byte 3 = 0; table B: + 32 KiB (immediate)
                const u32 addr_a = __builtin_amdgcn_perm(ia[set][q], base, 0x0c020000u | ((4u + (j % 4)) << 8));
                const u32 addr_b = __builtin_amdgcn_perm(ib[set][q], base, 0x0c020000u | ((4u + (j % 4)) << 8));
                m7_read2(a, c, addr_a, addr_b);
            };
            auto event = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                chunk_wait(kc);
                if constexpr (k + NS - 1 < NCH) prefetch(std::integral_constant<int, k + NS - 1>{}, sa, sb);
                else prefetch(std::integral_constant<int, k + NS - 1 - NCH>{}, sa_next, sb_next);
            };
            event(std::integral_constant<int, 0>{});
            u32x4 va[LOOKP], vb[LOOKP];
            static_for<0, LOOKP>([&](auto bc) { issue(bc, va[decltype(bc)::value], vb[decltype(bc)::value]); });
            static_for<0, R>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int newer = (R - 1 - j < LOOKP - 1) ? R - 1 - j : LOOKP - 1;   // row pairs issued after row j that may still be in flight
                m7_ldswait<2 * newer>(va[j % LOOKP], vb[j % LOOKP]);
                const u32x4 a = va[j % LOOKP], c = vb[j % LOOKP];
                u32x4 &x = acc[j];
                x = u32x4{xor3(x.x, a.x, c.x), xor3(x.y, a.y, c.y), xor3(x.z, a.z, c.z), xor3(x.w, a.w, c.w)};
                m7_pin(x);
                if constexpr (j + LOOKP < R) {
                    if constexpr ((j + LOOKP) % CH == 0) event(std::integral_constant<int, (j + LOOKP) / CH>{});
                    issue(std::integral_constant<int, j + LOOKP>{}, va[j % LOOKP], vb[j % LOOKP]);
                }
            });
        };

        // Step u of the job (pair t = k0 + u): look-ups on the tables in buffer u & 1 while the tables of pair t + 1 are built from BT slot
        // (u + 1) & 1 into the other buffer; the staging threads fetch the BT rows of pair t + 2 -> slot u & 1, read only after the barrier
        // that ends the step (and last read before the barrier that started it).
        {
            const u64 a7 = reinterpret_cast<u64>(A7);
            auto tab_row = [&](u32 p) -> i64 { return (i64)(p < S + 2 ? p : S + 1) * 8; };   // entries S and S + 1 exist (zero group)
            const u64x2s a7_0 = tab2[(i64)k0 * 8];
            u64 sa = scalar_u64(a7 + a7_0.x), sb = scalar_u64(a7 + a7_0.y);
            static_for<0, NS - 1>([&](auto cc) { prefetch_fresh(cc, sa, sb); });
            u32 st[2] = {0, 0};
            stage_load(st, tab2[(i64)k0 * 8 + 1 + wrow]);
            stage_store(st, 0, true, std::false_type{});
            __syncthreads();
            build(0, 0);
            const bool has1 = k0 + 1 < k1;
            stage_load(st, tab2[tab_row(k0 + 1) + 1 + wrow]);
            stage_store(st, 1, has1, std::false_type{});
            // What a step needs from outside is requested BEFORE the barrier that precedes it, where the waves wait for each other anyway: the BT
            // rows of pair t + 2 (registers -> LDS at the end of step t) and the index offsets of pair t + 1; the table entries behind them
            // one step earlier still (a scalar load takes ~300 cycles).  Entries S and S + 1 of the table exist (zero group).
            const u64x2s a7_1 = tab2[tab_row(k0 + 1)];
            u64 sa_next = scalar_u64(a7 + a7_1.x), sb_next = scalar_u64(a7 + a7_1.y);
            stage_load(st, tab2[tab_row(k0 + 2) + 1 + wrow]);
            u64x2s a7_next = tab2[tab_row(k0 + 2)], bt_next = tab2[tab_row(k0 + 3) + 1 + wrow];
            __syncthreads();
#ifdef SYMGPU_M7_STAMPS
            u64 tm[6] = {0, 0, 0, 0, 0, 0};
            const bool stamp = dbg != nullptr && blockIdx.x == 3;
#define M7_STAMP(c) const u64 c = stamp ? __builtin_amdgcn_s_memtime() : 0
#else
#define M7_STAMP(c)
#endif
            for (u32 t = k0, u = 0; t < k1; ++t, ++u) {
                M7_STAMP(c0);
                const bool more = t + 1 < k1, more2 = t + 2 < k1;     // uniform
                M7_STAMP(c1);
                if (more && build_first) build((u + 1) & 1u, (u + 1) & 1u);
                M7_STAMP(c2);
                lookups(u & 1u, sa, sb, more ? sa_next : sa, more ? sb_next : sb);
                M7_STAMP(c3);
                if (more && !build_first) build((u + 1) & 1u, (u + 1) & 1u);
                M7_STAMP(c4);
                stage_store(st, u & 1u, more2, std::true_type{});
                // the next step's requests
                sa = sa_next; sb = sb_next;
                sa_next = scalar_u64(a7 + a7_next.x); sb_next = scalar_u64(a7 + a7_next.y);
                stage_load(st, bt_next);
                a7_next = tab2[tab_row(t + 3)];
                bt_next = tab2[tab_row(t + 4) + 1 + wrow];
                M7_STAMP(c5);
                __syncthreads();
#ifdef SYMGPU_M7_STAMPS
                if (stamp) {
                    const u64 c6 = __builtin_amdgcn_s_memtime();
                    tm[0] += c1 - c0; tm[1] += c2 - c1; tm[2] += c3 - c2; tm[3] += c4 - c3; tm[4] += c5 - c4; tm[5] += c6 - c5;
                }
#endif
            }
            // the last step's requests (its successor's chunks, the BT rows of a pair past the end) are never used: let them land before
            // their registers go to the epilogue
            static_for<0, NS>([&](auto sc) {
                static_for<0, QC>([&](auto qc) { m7_vmwait<0>(ia[decltype(sc)::value][decltype(qc)::value], ib[decltype(sc)::value][decltype(qc)::value]); });
            });
            m7_vmwait<0>(st[0], st[1]);
#ifdef SYMGPU_M7_STAMPS
            if (stamp && lane == 0 && k0 == 0 && k1 == S) {
                for (int q = 0; q < 6; ++q) dbg[wave * 8 + q] = tm[q];
                dbg[wave * 8 + 6] = S;
                dbg[wave * 8 + 7] = __builtin_amdgcn_s_memrealtime();
            }
#endif
        }

        // A split tile: the workgroup that owns its steps [k, S) met it as its FIRST job and leaves the raw accumulators (no NOT) in scratch
        // [workgroup][row of the tile][32 words], then publishes flags[workgroup] = epoch.  The owner of steps [0, k) meets the tile as its
        // LAST job: when the neighbour's part is published (always, while all workgroups are resident: it was written a tile time or more
        // ago) it adds that part to its accumulators and finishes the tile like a whole one; otherwise (workgroups queued behind each other
        // on a shared GPU) it leaves its own part in scratch and asks k_m7_fixup for the tile — nobody ever waits for anybody.
        bool whole = (k0 == 0 && k1 == S);
        const int lane_e = m7_opaque(lane), slot_e = lane_e >> 4, wp_e = lane_e & 15;
        const i64 part_off = ((i64)wave * (4 * R) + slot_e * R) * M7_TILE_W + 2 * wp_e;
        if (!whole && k0 > 0) {
            u64 *dst = part + ((i64)blockIdx.x * 2) * PART_WORDS + part_off;
#pragma unroll
            for (int j = 0; j < R; ++j) *reinterpret_cast<u32x4 *>(dst + (i64)j * M7_TILE_W) = acc[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (!whole) {
            u32 *const seen = reinterpret_cast<u32 *>(bt_stage);       // (free: the job's steps are over)
            if (threadIdx.x == 0) *seen = __hip_atomic_load(flags + blockIdx.x + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool published = (*seen == epoch) && !force_fixup;
            __syncthreads();
            if (published) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const u64 *src = part + ((i64)(blockIdx.x + 1) * 2) * PART_WORDS + part_off;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + (i64)j * M7_TILE_W));
                    acc[j] ^= v;
                }
                whole = true;
            } else {
                u64 *dst = part + ((i64)blockIdx.x * 2 + 1) * PART_WORDS + part_off;
#pragma unroll
                for (int j = 0; j < R; ++j) *reinterpret_cast<u32x4 *>(dst + (i64)j * M7_TILE_W) = acc[j];
                if (threadIdx.x == 0) flags[gridDim.x + blockIdx.x] = epoch;   // read by the fix-up launch that follows this one
            }
        }
        if (!whole) {
        } else if (bytes) {
            // One byte per pair, 16-byte stores, a wave store = 1 KiB of one output row.  A lane holds 128 result bits of a row but must write 16
            // columns of it: the rows are turned round through the (now free) table area, 128 KiB / WAVES per wave = 4 * RO rows x 2048 bits per pass.
            uint8_t *const out = reinterpret_cast<uint8_t *>(out_v);
            constexpr int RO = 128 / WAVES;                              // rows per slot and pass
            uint8_t *const region = m7_lds + wave * (4 * RO * M7_ENTRY_BYTES);
            const i64 wave_row0 = tile_row0 + (i64)wave * (4 * R);
#pragma unroll
            for (int p0 = 0; p0 < R; p0 += RO) {
#pragma unroll
                for (int jj = 0; jj < RO; ++jj) {
                    if (p0 + jj < R) {
                        *reinterpret_cast<u32x4 *>(region + (slot_e * RO + jj) * M7_ENTRY_BYTES + wp_e * 16) = ~acc[p0 + jj];   // commute = NOT parity
                    }
                }
                __syncthreads();
                const int rows_here = (R - p0 < RO) ? R - p0 : RO;
                for (int q = 0; q < 4 * rows_here * 2; ++q) {
                    const int half = q & 1, rl = q >> 1, s = rl / rows_here, jj = rl - s * rows_here;
                    const i64 i = wave_row0 + s * R + p0 + jj;
                    const i64 col = (tile_w0 << 6) + half * 1024 + lane_e * 16;
                    if (i < N && col < m_cols) {
                        const u32 b16 = *reinterpret_cast<const uint16_t *>(region + (s * RO + jj) * M7_ENTRY_BYTES + half * 128 + lane_e * 2);
                        u32x4 v;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const u32 x = (b16 >> (4 * k)) & 0xFu;
                            v[k] = (x | (x << 7) | (x << 14) | (x << 21)) & 0x01010101u;
                        }
                        m7_store16(out + i * out_stride + col, v, m_cols - col);
                    }
                }
                __syncthreads();
            }
        } else {
            // commute = NOT parity; columns >= M stay zero
            u64 *const out_bits = reinterpret_cast<u64 *>(out_v);
            const i64 Mw = (m_cols + 63) >> 6;
            const u64 last_mask = (m_cols & 63) ? ((1ULL << (m_cols & 63)) - 1) : ~0ULL;
            const i64 row0 = tile_row0 + (i64)wave * (4 * R) + slot_e * R;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const i64 i = row0 + j;
                if (i < N) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const i64 jw = tile_w0 + 2 * wp_e + h;
                        if (jw < Mw) {
                            u64 v = ~(h ? ((u64)acc[j].w << 32) | acc[j].z : ((u64)acc[j].y << 32) | acc[j].x);
                            if (jw == Mw - 1) v &= last_mask;
                            out_bits[i * out_stride + jw] = v;
                        }
                    }
                }
            }
        }
    }
}

// the split tiles of a stream-K launch that were NOT finished inside it: boundary b lies between workgroups b and b + 1; tail part of b XOR
// head part of b + 1 -> output
template <int R>
__global__ __launch_bounds__(256) void k_m7_fixup(const u64 *__restrict__ part, const u32 *__restrict__ np_ptr, i64 n_rt, i64 n_tiles, int P, i64 N,
                                                  void *__restrict__ out_v, i64 out_stride, i64 m_cols, int bytes, const u32 *__restrict__ flags, u32 epoch) {
    if (flags[P + blockIdx.x] != epoch) return;                       // (the usual case: the tile was finished inside the main launch)
    constexpr int WG_ROWS = 4 * M7_WAVES * R;
    constexpr i64 PART_WORDS = (i64)WG_ROWS * M7_TILE_W;
    const u32 S = *np_ptr;
    const i64 L = n_tiles * (i64)S;
    const int b = blockIdx.x;
    const i64 s1 = ((i64)b + 1) * L / P;
    if (s1 % S == 0) return;
    const i64 tile = s1 / S, rt = tile % n_rt, ct = tile / n_rt;
    const u64 *pt = part + ((i64)b * 2 + 1) * PART_WORDS;
    const u64 *ph = part + ((i64)(b + 1) * 2) * PART_WORDS;
    if (bytes) {
        uint8_t *const out = reinterpret_cast<uint8_t *>(out_v);
        for (int idx = blockIdx.y * 256 + threadIdx.x; idx < WG_ROWS * 128; idx += gridDim.y * 256) {
            const int lr = idx >> 7, c = idx & 127;
            const i64 i = rt * WG_ROWS + lr, col = ct * (64 * M7_TILE_W) + c * 16;
            if (i >= N || col >= m_cols) continue;
            const u64 w = ~(pt[lr * M7_TILE_W + (c >> 2)] ^ ph[lr * M7_TILE_W + (c >> 2)]);
            const u32 b16 = (u32)(w >> (16 * (c & 3))) & 0xFFFFu;
            u32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 x = (b16 >> (4 * k)) & 0xFu;
                v[k] = (x | (x << 7) | (x << 14) | (x << 21)) & 0x01010101u;
            }
            m7_store16(out + i * out_stride + col, v, m_cols - col);
        }
    } else {
        u64 *const out_bits = reinterpret_cast<u64 *>(out_v);
        const i64 Mw = (m_cols + 63) >> 6;
        const u64 last_mask = (m_cols & 63) ? ((1ULL << (m_cols & 63)) - 1) : ~0ULL;
        for (int idx = blockIdx.y * 256 + threadIdx.x; idx < WG_ROWS * M7_TILE_W; idx += gridDim.y * 256) {
            const int lr = idx / M7_TILE_W, wd = idx % M7_TILE_W;
            const i64 i = rt * WG_ROWS + lr, jw = ct * M7_TILE_W + wd;
            if (i >= N || jw >= Mw) continue;
            u64 v = ~(pt[idx] ^ ph[idx]);
            if (jw == Mw - 1) v &= last_mask;
            out_bits[i * out_stride + jw] = v;
        }
    }
}

template <int R, int LOOKP, int CH, int NS>
static int launch_m7s(const uint8_t *A7, i64 Npad, i64 N, const u64 *BT, i64 Mw_pad, const u64 *steptab, const u32 *np, void *out, i64 stride, i64 M, bool bytes, int max_steps) {
    const bool attr = SG_DEVICE_ONCE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_commutes_m4r7s<R, LOOKP, CH, NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         M7S_LDS) == hipSuccess);
    if (!attr) { set_error("commutes_m4r7: %d bytes of LDS refused", M7S_LDS); return SYMGPU_E_HIP; }
    constexpr i64 WG_ROWS = 4 * M7_WAVES * R;
    const i64 n_rt = Npad / WG_ROWS, n_ct = Mw_pad / M7_TILE_W, n_tiles = n_rt * n_ct;
    const int P = ctx().num_cu;
    // one tile per workgroup where the persistent launch's published parts only cost: short operators (below M7_STREAM_MIN_STEPS steps a
    // tile the table is bound by its own bytes — 30,000^2 terms of 20 / 100 / 150 qubits, R = 16: 0.339 / 0.370 / 0.372 ms streamed against
    // 0.284 / 0.307 / 0.339 ms tile by tile), launches of little work (20,000^2 terms of 200 / 300 / 400 qubits: 0.220 / 0.238 / 0.259 against
    // 0.189 / 0.219 / 0.254 ms; the streamed launch wins from about 125 tile-steps per workgroup) and fewer tiles than compute units
    // (profiles/r06_m4r_pick.txt).  Runtime switches ("Environment switches"): both force a path the kernel takes by itself.
    bool stream = n_tiles >= P && max_steps >= M7_STREAM_MIN_STEPS && n_tiles * max_steps >= (i64)M7_STREAM_MIN_WORK * P;
    if (const char *e = getenv("SYMGPU_M4R_STREAM")) stream = n_tiles >= P && atoi(e) != 0;
    const int force_fixup = getenv("SYMGPU_M4R_FIXUP") ? 1 : 0;
    Scratch part, dbgbuf;
    Context &c = ctx();
    if (!c.m7_flags) {                                                // [P] "head part published" + [P] "tile left to the fix-up launch", compared with the launch's epoch
        HIP_TRY(hipMalloc((void **)&c.m7_flags, (size_t)2 * 1024 * 4));
        HIP_TRY(hipMemsetAsync(c.m7_flags, 0, (size_t)2 * 1024 * 4, c.stream));
    }
    if (++c.m7_epoch == 0) c.m7_epoch = 1;
    SG_REQUIRE(P <= 1024, "commutes_m4r7: more than 1024 compute units");
    if (stream) SG_TRY(part.alloc((size_t)P * 2 * WG_ROWS * M7_TILE_W * 8));
    u64 *dbg = nullptr;
#ifdef SYMGPU_M7_STAMPS
    if (getenv("SYMGPU_M4R_STAMPS")) { SG_TRY(dbgbuf.alloc(64 * 8)); HIP_TRY(hipMemsetAsync(dbgbuf.p, 0, 64 * 8, ctx().stream)); dbg = dbgbuf.as<u64>(); }
#endif
    SG_REQUIRE(n_tiles < (i64)1 << 31, "commutes_m4r7: tile count");
    hipLaunchKernelGGL((k_commutes_m4r7s<R, LOOKP, CH, NS>), dim3((unsigned)(stream ? P : n_tiles)), dim3(64 * M7_WAVES), M7S_LDS, ctx().stream, A7, Npad, N, BT, Mw_pad,
                       steptab, np, out, stride, M, bytes ? 1 : 0, part.as<u64>(), n_rt, n_tiles, stream ? 1 : 0, force_fixup, dbg, c.m7_flags, c.m7_epoch);
    KERNEL_CHECK();
#ifdef SYMGPU_M7_STAMPS
    if (dbg) {
        u64 h[64];
        HIP_TRY(hipMemcpyAsync(h, dbg, sizeof h, hipMemcpyDeviceToHost, ctx().stream));
        HIP_TRY(hipStreamSynchronize(ctx().stream));
        for (int w = 0; w < 8; ++w)
            fprintf(stderr, "m7 timing wave %d (cycles per step, S=%llu): stage %.0f build-first %.0f lookups %.0f build-after %.0f store %.0f barrier %.0f\n", w, (unsigned long long)h[w * 8 + 6],
                    (double)h[w * 8] / (double)(h[w * 8 + 6] ? h[w * 8 + 6] : 1), (double)h[w * 8 + 1] / (double)(h[w * 8 + 6] ? h[w * 8 + 6] : 1), (double)h[w * 8 + 2] / (double)(h[w * 8 + 6] ? h[w * 8 + 6] : 1),
                    (double)h[w * 8 + 3] / (double)(h[w * 8 + 6] ? h[w * 8 + 6] : 1), (double)h[w * 8 + 4] / (double)(h[w * 8 + 6] ? h[w * 8 + 6] : 1), (double)h[w * 8 + 5] / (double)(h[w * 8 + 6] ? h[w * 8 + 6] : 1));
    }
#endif
    if (stream) {
        hipLaunchKernelGGL((k_m7_fixup<R>), dim3((unsigned)(P - 1), 24), dim3(256), 0, ctx().stream, part.as<u64>(), np, n_rt, n_tiles, P, N, out, stride, M, bytes ? 1 : 0, c.m7_flags, c.m7_epoch);
        KERNEL_CHECK();
    }
    return SYMGPU_OK;
}

// prepared operands of one call (A7, flags, klist) and the launch; bt_p = bit-major copy of B (built / cached by commutes_m4r_dev)
int commutes_m4r7_launch(const u64 *A, i64 N, i64 M, int Wq, const u64 *bt_p, i64 Mw_pad, int R, bool bytes, void *dst, i64 stride) {
    hipStream_t st = ctx().stream;
    const int W = 2 * Wq, ng7 = (128 * Wq + 6) / 7;
    const i64 Npad = (N + (i64)4 * M7_WAVES * R - 1) / ((i64)4 * M7_WAVES * R) * ((i64)4 * M7_WAVES * R);   // multiples of 256
    Scratch a7, flags, klist;
    SG_TRY(a7.alloc((size_t)(ng7 + 1) * Npad));
    const size_t flag_bytes = ((size_t)(ng7 + 1) * 4 + 255) / 256 * 256;   // (a whole number of 256-byte pieces: one fill kernel, not a body and a tail)
    SG_TRY(flags.alloc(flag_bytes));
    SG_TRY(klist.alloc((size_t)(ng7 + 2) * 4));
    HIP_TRY(hipMemsetAsync(flags.p, 0, flag_bytes, st));
    hipLaunchKernelGGL(k_m7_a7, dim3((unsigned)(Npad / 256), (unsigned)((W + A7_CW - 1) / A7_CW)), dim3(256), 0, st, A, N, W, ng7, a7.as<uint8_t>(), Npad, flags.as<u32>());
    KERNEL_CHECK();
    u32 *np = flags.as<u32>() + ng7;
    Scratch steptab;
    const int max_pairs = (ng7 + 1) / 2;
    SG_TRY(steptab.alloc((size_t)(max_pairs + 2) * 16 * 8));
    hipLaunchKernelGGL(k_m7_klist, dim3(1), dim3(64), 0, st, flags.as<u32>(), ng7, klist.as<u32>(), np, max_pairs, Npad, Mw_pad, Wq, steptab.as<u64>());
    KERNEL_CHECK();
    ProfScope prof(1);
#define M7S_ARGS a7.as<uint8_t>(), Npad, N, bt_p, Mw_pad, steptab.as<u64>(), np, dst, stride, M, bytes, max_pairs
    if (R == 48) SG_TRY((launch_m7s<48, 3, 8, 3>(M7S_ARGS)));
    else if (R == 24) SG_TRY((launch_m7s<24, 4, 8, 3>(M7S_ARGS)));
    else SG_TRY((launch_m7s<16, 4, 4, 4>(M7S_ARGS)));
#undef M7S_ARGS
    return SYMGPU_OK;
}

}  // namespace symgpu
