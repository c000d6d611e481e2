// commute_m4r7.hip — round 5: the Four-Russians commutation kernel with TWO 7-bit tables per step, folded with v_bitop3 (XOR3).
// (reference: symmer/operators/base.py:938-971 -> matmul_GF2 / numba_dot_matmal_GF2, utils.py:9-78: f64 dgemm, then % 2)
//
// commute_m4r.hip cuts the contraction axis into bytes: one 256-entry table (64 KiB) per step, one ds_read_b128 + 4 v_xor + 1 v_perm per
// row and 8 contraction bits.  Its look-up stream is VALU bound (5 instructions per read: 4.97 cycles per read and CU against the LDS
// pipe's 4.0, tools/ubench_lds.hip), the table build is 1,170 cycles per step, and the step ends in a barrier.  Here a step covers 14
// contraction bits: two 128-entry tables (2 x 32 KiB, double buffered: the same 128 KiB of LDS), a row reads one entry of EACH and folds
// both into its accumulator with ONE v_bitop3 (a ^ b ^ c) per dword — 2 reads + 2 v_perm + 4 v_bitop3 = 3 VALU instructions per read,
// and the stream runs at the LDS pipe's 4.02 cycles per read (measured).  Per contraction bit: look-ups 4.02 / 7 = 0.57 cycles per row-wave
// (8-bit tables: 4.97 / 8 = 0.62), table entries written 256 / 14 = 18.3 (32), barriers 1 / 14 (1 / 8).
// Measured (200,000^2 terms at n = 2000, one launch): 36.1 ms against 40.1 ms with one 8-bit table per step (R = 48); R = 40: 38.1 / 44.4,
// R = 24: 50.8 / 58.7, R = 16: 60.0 / 75.3.  Without the table builds the launch takes 31.8 ms: the look-ups still cost ~5.5 cycles per read in
// the kernel — every step starts with the index bytes and the first table entries still on their way (two dependent LDS round trips
// behind a workgroup barrier, twice per step at R = 48), which the depth of the read window does not change (2 / 3 / 4 pairs: 37.4 / 36.1 /
// 36.4 ms).  Measured and dropped this round: look-ups and the next table's build as one interleaved instruction stream (8-bit kernel, R = 40:
// 45.8 against 44.7 ms — the LDS pipe serves reads and writes from one queue, the interleaved writes delay the reads the folds wait for) and
// a staggered start of the first workgroup of every CU so that the 3 MB tile epilogues do not meet in the memory system (35.9 / 35.8 ms);
// the index dwords in a rolling window of their own (read 2 LOOKP + 1 rows ahead inside the stream: one pass over all rows at R = 48 and one
// round trip less per step, but 38.3 against 36.5 ms — the index reads queue in front of the table reads the folds wait for).
//
// Layouts prepared per call:
//   A7[g][i]   bits [7g, 7g + 7) of packed row i (values 0..127), group-major, i zero padded to Npad; one more all-zero group at index
//              NG7 pads an odd number of non-zero groups to whole pairs (entry 0 of any table is the XOR of no rows = 0).
//   BT[c][jw]  bit c of B rows 64jw..64jw+63 (the bit-major copy of commute_m4r.hip, cached on the operator); contraction bit c of A pairs
//              with row c + 64Wq (c in the X half) or c - 64Wq (Z half).
//   klist      the groups in which A has any non-zero value, ascending, padded to an even count with the zero group.
#include "common.h"
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

namespace symgpu {

typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u64x2 lds_u64x2;            // raw LDS address -> ds_read_b128 without a base add

constexpr int M7_TILE_W = 32;                          // 64-bit words per column tile: 2048 columns
constexpr int M7_ENTRY_BYTES = M7_TILE_W * 8;          // 256
constexpr int M7_TABLE_BYTES = 128 * M7_ENTRY_BYTES;   // 32 KiB: one 7-bit group
constexpr int M7_BUF_BYTES = 2 * M7_TABLE_BYTES;       // the two tables of a step
constexpr int M7_LDS = 2 * M7_BUF_BYTES;               // double buffered: 128 KiB
constexpr int M7_BT_STAGE = 14 * M7_ENTRY_BYTES;       // the 14 bit-rows of a step: 3.5 KiB
constexpr int M7_WAVES = 8;
constexpr int m7_lds_bytes(int wg_rows) { return M7_LDS + 2 * M7_BT_STAGE + 4 * wg_rows; }

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

// compact the flagged groups (single wave; ascending) and pad to an even count with the zero group
__global__ __launch_bounds__(64) void k_m7_klist(const u32 *__restrict__ flags, int ng7, u32 *__restrict__ klist, u32 *__restrict__ n_pairs) {
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
}

// R = rows per 16-lane slot (wave = 4 slots), LOOKP = row PAIRS of reads in flight per wave, BYTES: np.bool_ output from the epilogue
template <int R, int LOOKP, bool BYTES>
__global__ __launch_bounds__(64 * M7_WAVES) void k_commutes_m4r7(const uint8_t *__restrict__ A7, i64 Npad, i64 N, const u64 *__restrict__ BT, i64 Mw_pad,
                                                                  int Wq, const u32 *__restrict__ klist, const u32 *__restrict__ np_ptr,
                                                                  u64 *__restrict__ out_bits, i64 out_stride, i64 m_cols) {
    constexpr int WAVES = M7_WAVES;
    constexpr int WG_ROWS = 4 * WAVES * R;
    constexpr int PASSES = R > 40 ? 2 : 1, RP = R / PASSES;            // index registers for RP rows at a time (R = 48: two passes of 24)
    static_assert(RP % 4 == 0 && RP > LOOKP && LOOKP <= 6, "index bytes arrive as dwords; the wait counts are immediates up to 10");
    extern __shared__ __attribute__((aligned(16))) uint8_t m7_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = lane >> 4, wp = lane & 15;
    const i64 row0 = (i64)blockIdx.x * WG_ROWS + (i64)wave * (4 * R) + slot * R;
    const i64 tile_w0 = (i64)blockIdx.y * M7_TILE_W;
    const u32 n_pairs = *np_ptr;

    u64 acc[R][2];
#pragma unroll
    for (int j = 0; j < R; ++j) acc[j][0] = acc[j][1] = 0;

    // the look-up address is assembled bytewise (v_perm_b32), which needs the tables at LDS offset 0: m7_lds is the kernel's only LDS object
    if ((u32)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)m7_lds != 0) __builtin_trap();
    uint8_t *const bt_stage = m7_lds + M7_LDS;                        // [2][14 rows][256 B]
    uint8_t *const ix_stage = bt_stage + 2 * M7_BT_STAGE;             // [2][2 tables][WG_ROWS] bytes
    const int tid = threadIdx.x;
    const i64 half_bits = (i64)64 * Wq;
    // staging threads: 0..223 one 16-byte piece of the 14 BT rows; 224.. one 16-byte piece of the 2 x WG_ROWS index bytes
    constexpr int IX_THREADS = 2 * WG_ROWS / 16;
    static_assert(224 + IX_THREADS <= 64 * WAVES, "staging fits the workgroup");
    const bool stage_bt = tid < 224, stage_ix = tid >= 224 && tid < 224 + IX_THREADS;
    const int bt_r = tid >> 4;                                        // 0..13: row r of table bt_r / 7
    const int ix_k = (tid - 224) / (WG_ROWS / 16), ix_off = 16 * ((tid - 224) % (WG_ROWS / 16));

    auto bt_row = [&](u32 g, int r) -> i64 {                          // BT row that contraction bit 7g + r of A pairs with
        const i64 c = 7 * (i64)g + r;
        return c < half_bits ? c + half_bits : (c < 2 * half_bits ? c - half_bits : 0);   // (padding bits of the last group: A has zeros there)
    };
    auto stage_load = [&](u32 ga_bt, u32 gb_bt, bool have_bt, u32 ga_ix, u32 gb_ix, bool have_ix) -> u32x4 {
        u32x4 v = {0, 0, 0, 0};
        if (stage_bt && have_bt) v = *reinterpret_cast<const u32x4 *>(BT + bt_row(bt_r < 7 ? ga_bt : gb_bt, bt_r % 7) * Mw_pad + tile_w0 + 2 * (tid & 15));
        if (stage_ix && have_ix) v = *reinterpret_cast<const u32x4 *>(A7 + (i64)(ix_k ? gb_ix : ga_ix) * Npad + (i64)blockIdx.x * WG_ROWS + ix_off);
        return v;
    };
    auto stage_store = [&](u32x4 v, u32 bt_slot, bool have_bt, u32 ix_slot, bool have_ix) {
        if (stage_bt && have_bt) *reinterpret_cast<u32x4 *>(bt_stage + bt_slot * M7_BT_STAGE + 16 * tid) = v;
        if (stage_ix && have_ix) *reinterpret_cast<u32x4 *>(ix_stage + ix_slot * (2 * WG_ROWS) + ix_k * WG_ROWS + ix_off) = v;
    };
    // table builder: waves 0-3 table A, 4-7 table B; lane (h, w) writes word w of the 16 entries ent_hi * 16 + g (Gray order over g)
    const int bh = lane >> 5, bw = lane & 31;
    const int tb = wave >> 2;
    const u32 ent_hi = (u32)(wave & 3) * 2 + bh;                      // entry bits 4..6
    auto build = [&](u32 buf, u32 bt_slot) {
        u64 brow[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) brow[r] = *reinterpret_cast<const u64 *>(bt_stage + bt_slot * M7_BT_STAGE + (tb * 7 + r) * 256 + bw * 8);
        u64 e = 0;
#pragma unroll
        for (int r = 4; r < 7; ++r) e ^= ((ent_hi >> (r - 4)) & 1u) ? brow[r] : 0ULL;
        uint8_t *dst = m7_lds + buf * M7_BUF_BYTES + tb * M7_TABLE_BYTES + (ent_hi * 16) * M7_ENTRY_BYTES + bw * 8;
        constexpr int flip[15] = {0, 1, 0, 2, 0, 1, 0, 3, 0, 1, 0, 2, 0, 1, 0};
        int g = 0;
        *reinterpret_cast<u64 *>(dst) = e;
#pragma unroll
        for (int s = 0; s < 15; ++s) {
            g ^= 1 << flip[s];
            e ^= brow[flip[s]];
            *reinterpret_cast<u64 *>(dst + g * M7_ENTRY_BYTES) = e;
        }
    };
    const u32 look_base = (u32)wp * 16;
    auto lookups = [&](u32 buf, u32 ix_slot) {
        const u32 base = look_base | (buf << 16);
        const uint8_t *ixa = ix_stage + ix_slot * (2 * WG_ROWS) + wave * (4 * R) + slot * R;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            u32 ia[RP / 4], ib[RP / 4];
#pragma unroll
            for (int q = 0; q < RP / 4; ++q) {
                ia[q] = *reinterpret_cast<const u32 *>(ixa + p * RP + 4 * q);
                ib[q] = *reinterpret_cast<const u32 *>(ixa + WG_ROWS + p * RP + 4 * q);
            }
            // address byte 0 = lane offset, byte 1 = table index (entry stride 256 B), byte 2 = buffer, byte 3 = 0; table B: + 32 KiB (immediate offset).
            // The reads are written as instructions (and the waits for them by hand): left to the compiler, the rolling window of LOOKP row
            // pairs in flight collapsed to one pair (s_waitcnt lgkmcnt(1) in front of every fold: 7 cycles per read instead of 4).  LDS
            // operations complete in order, so "at most 2 (LOOKP - 1) outstanding" means the reads of the row being folded have landed; LDS
            // operations the compiler issues itself around this block only make these waits stricter, never weaker.
            auto issue = [&](int j, u64x2 &a, u64x2 &c) {
                const u32 addr_a = __builtin_amdgcn_perm(ia[j / 4], base, 0x0c020000u | ((4u + (j % 4)) << 8));
                const u32 addr_b = __builtin_amdgcn_perm(ib[j / 4], base, 0x0c020000u | ((4u + (j % 4)) << 8));
                asm volatile("ds_read_b128 %0, %1" : "=v"(a) : "v"(addr_a));
                asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(c) : "v"(addr_b));
            };
            // every index dword is complete before the first hand-written read: a compiler wait for one of them later on would count the
            // hand-written reads as its own and drain the window
#pragma unroll
            for (int q = 0; q < RP / 4; ++q) asm volatile("" : "+v"(ia[q]), "+v"(ib[q]));
            u64x2 va[LOOKP], vb[LOOKP];
#pragma unroll
            for (int b = 0; b < LOOKP; ++b) issue(b, va[b], vb[b]);
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const int newer = (RP - 1 - j < LOOKP - 1) ? RP - 1 - j : LOOKP - 1;     // row pairs issued after row j that may still be in flight
                switch (2 * newer) {                                                    // (the count is an immediate of the instruction)
                    case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va[j % LOOKP]), "+v"(vb[j % LOOKP])); break;
                    case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(va[j % LOOKP]), "+v"(vb[j % LOOKP])); break;
                    case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(va[j % LOOKP]), "+v"(vb[j % LOOKP])); break;
                    case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(va[j % LOOKP]), "+v"(vb[j % LOOKP])); break;
                    case 8: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(va[j % LOOKP]), "+v"(vb[j % LOOKP])); break;
                    default: asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(va[j % LOOKP]), "+v"(vb[j % LOOKP])); break;
                }
                const u64x2 a = va[j % LOOKP], c = vb[j % LOOKP];
                u64 &x0 = acc[p * RP + j][0], &x1 = acc[p * RP + j][1];
                const u32 l0 = xor3((u32)x0, (u32)a.x, (u32)c.x), h0 = xor3((u32)(x0 >> 32), (u32)(a.x >> 32), (u32)(c.x >> 32));
                const u32 l1 = xor3((u32)x1, (u32)a.y, (u32)c.y), h1 = xor3((u32)(x1 >> 32), (u32)(a.y >> 32), (u32)(c.y >> 32));
                x0 = ((u64)h0 << 32) | l0;
                x1 = ((u64)h1 << 32) | l1;
                asm volatile("" : "+v"(x0), "+v"(x1));
                if (j + LOOKP < RP) issue(j + LOOKP, va[j % LOOKP], vb[j % LOOKP]);
            }
        }
    };

    // Step t: look-ups on the tables of pair t (buffer t&1, indices in slot t&1) while the tables of pair t+1 are built from BT slot (t+1)&1; the
    // staging threads fetch the BT rows of pair t+2 -> slot t&1 and the indices of pair t+1 -> slot (t+1)&1, both read only after the barrier
    // that ends the step (and last read before the barrier that started it).
    if (n_pairs > 0) {
        auto ga = [&](u32 p) -> u32 { return klist[2 * p]; };
        auto gb = [&](u32 p) -> u32 { return klist[2 * p + 1]; };
        stage_store(stage_load(ga(0), gb(0), true, ga(0), gb(0), true), 0, true, 0, true);
        __syncthreads();
        build(0, 0);
        const bool has1 = n_pairs > 1;
        stage_store(stage_load(has1 ? ga(1) : 0, has1 ? gb(1) : 0, has1, 0, 0, false), 1, has1, 0, false);
        __syncthreads();
        for (u32 t = 0; t < n_pairs; ++t) {
            const bool more = t + 1 < n_pairs, more2 = t + 2 < n_pairs;   // uniform
            const u32x4 st = stage_load(more2 ? ga(t + 2) : 0, more2 ? gb(t + 2) : 0, more2, more ? ga(t + 1) : 0, more ? gb(t + 1) : 0, more);
            lookups(t & 1u, t & 1u);
            if (more) build((t + 1) & 1u, (t + 1) & 1u);
            stage_store(st, t & 1u, more2, (t + 1) & 1u, more);
            __syncthreads();
        }
    }

    if constexpr (BYTES) {
        // One byte per pair, 16-byte stores, a wave store = 1 KiB of one output row.  A lane holds 128 result bits of a row but must write 16
        // columns of it: the rows are turned round through the (now free) table area, 128 KiB / WAVES per wave = 4 * RO rows x 2048 bits per pass.
        uint8_t *const out = reinterpret_cast<uint8_t *>(out_bits);
        constexpr int RO = 128 / WAVES;                              // rows per slot and pass
        uint8_t *const region = m7_lds + wave * (4 * RO * M7_ENTRY_BYTES);
        const i64 wave_row0 = (i64)blockIdx.x * WG_ROWS + (i64)wave * (4 * R);
        __syncthreads();                                             // every wave is done with the tables
#pragma unroll
        for (int p0 = 0; p0 < R; p0 += RO) {
#pragma unroll
            for (int jj = 0; jj < RO; ++jj) {
                if (p0 + jj < R) {
                    const u64x2 v = {~acc[p0 + jj][0], ~acc[p0 + jj][1]};                       // commute = NOT parity
                    *reinterpret_cast<u64x2 *>(region + (slot * RO + jj) * M7_ENTRY_BYTES + wp * 16) = v;
                }
            }
            __syncthreads();
            const int rows_here = (R - p0 < RO) ? R - p0 : RO;
            for (int q = 0; q < 4 * rows_here * 2; ++q) {
                const int half = q & 1, rl = q >> 1, s = rl / rows_here, jj = rl - s * rows_here;
                const i64 i = wave_row0 + s * R + p0 + jj;
                const i64 col = (tile_w0 << 6) + half * 1024 + lane * 16;           // m_cols % 16 == 0: whole chunks in or out
                if (i < N && col < m_cols) {
                    const u32 b16 = *reinterpret_cast<const uint16_t *>(region + (s * RO + jj) * M7_ENTRY_BYTES + half * 128 + lane * 2);
                    u32x4 v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u32 x = (b16 >> (4 * k)) & 0xFu;
                        v[k] = (x | (x << 7) | (x << 14) | (x << 21)) & 0x01010101u;
                    }
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out + i * out_stride + col));
                }
            }
            __syncthreads();
        }
    } else {
        // commute = NOT parity; columns >= M stay zero
        const i64 Mw = (m_cols + 63) >> 6;
        const u64 last_mask = (m_cols & 63) ? ((1ULL << (m_cols & 63)) - 1) : ~0ULL;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const i64 i = row0 + j;
            if (i < N) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const i64 jw = tile_w0 + 2 * wp + h;
                    if (jw < Mw) {
                        u64 v = ~acc[j][h];
                        if (jw == Mw - 1) v &= last_mask;
                        out_bits[i * out_stride + jw] = v;
                    }
                }
            }
        }
    }
}

template <int R, int LOOKP, bool BYTES>
static int launch_m7(const uint8_t *A7, i64 Npad, i64 N, const u64 *BT, i64 Mw_pad, int Wq, const u32 *klist, const u32 *np, void *out, i64 stride, i64 M) {
    constexpr int lds = m7_lds_bytes(4 * M7_WAVES * R);
    const bool attr = SG_DEVICE_ONCE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_commutes_m4r7<R, LOOKP, BYTES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         lds) == hipSuccess);
    if (!attr) { set_error("commutes_m4r7: %d bytes of LDS refused", lds); return SYMGPU_E_HIP; }
    const i64 gx = Npad / (4 * M7_WAVES * R), gy = Mw_pad / M7_TILE_W;
    // blockIdx.x (fast) walks the row blocks: workgroups that run together share the BT column tile in L2
    for (i64 y0 = 0; y0 < gy; y0 += 65535) {
        const i64 ny = gy - y0 < 65535 ? gy - y0 : 65535;
        hipLaunchKernelGGL((k_commutes_m4r7<R, LOOKP, BYTES>), dim3((unsigned)gx, (unsigned)ny), dim3(64 * M7_WAVES), lds, ctx().stream, A7, Npad, N,
                           BT + y0 * M7_TILE_W, Mw_pad, Wq, klist, np,
                           BYTES ? reinterpret_cast<u64 *>(static_cast<uint8_t *>(out) + y0 * M7_TILE_W * 64) : static_cast<u64 *>(out) + y0 * M7_TILE_W,
                           stride, M - y0 * M7_TILE_W * 64);
        KERNEL_CHECK();
    }
    return SYMGPU_OK;
}


// Per-step scalars of the stream-K kernel, computed once: entry p (16 u64) = { byte offset of group ga's index bytes in A7, same for gb,
// then for w = 0..6 the word offsets of the BT rows that bit w of ga / of gb pairs with }.  Two entries behind the last pair point at
// the zero group / row 0, so the kernel fetches entries t + 1 and t + 2 without asking whether they exist.
__global__ __launch_bounds__(64) void k_m7_steptab(const u32 *__restrict__ klist, const u32 *__restrict__ n_pairs, int ng7, int max_pairs, i64 Npad, i64 Mw_pad, int Wq,
                                                   u64 *__restrict__ tab) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= max_pairs + 2) return;
    const bool live = (u32)p < *n_pairs;
    const u32 ga = live ? klist[2 * p] : (u32)ng7, gb = live ? klist[2 * p + 1] : (u32)ng7;
    const i64 half_bits = (i64)64 * Wq;
    auto bt_row = [&](u32 g, int r) -> i64 {                          // BT row that contraction bit 7g + r of A pairs with
        const i64 c = 7 * (i64)g + r;
        return c < half_bits ? c + half_bits : (c < 2 * half_bits ? c - half_bits : 0);   // (padding bits of the last group: A has zeros there)
    };
    u64 *e = tab + (i64)p * 16;
    e[0] = (u64)ga * (u64)Npad;
    e[1] = (u64)gb * (u64)Npad;
    for (int w = 0; w < 7; ++w) {
        e[2 + 2 * w] = (u64)(bt_row(ga, w) * Mw_pad);
        e[3 + 2 * w] = (u64)(bt_row(gb, w) * Mw_pad);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the same two-table step inside a PERSISTENT workgroup that owns a contiguous range of the (tile, step) space ("stream-K").
//   * Every workgroup gets the same number of steps (+-1), whatever the tile count: a 25,000 x 200,000 slab is 1,666 tiles = 6.51 per CU,
//     which cost 7 rounds as one-tile workgroups.  A range starts and ends inside a tile; the two parts of such a tile leave their raw
//     accumulators in scratch and k_m7_fixup adds them (XOR) and writes the result — at most one split tile per workgroup boundary.
//   * The ranges start at different steps of their tiles, so the tile epilogues (3 MB of np.bool_ each) are spread over the launch instead
//     of arriving from all 256 CUs at once.
//   * Index bytes come from global memory into registers (vmcnt; chunks of CH rows, NS register sets in flight) instead of through LDS:
//     no LDS reads for them, no dependent LDS round trip at the start of a step, and the rolling window of table reads runs through the
//     whole step instead of draining between two passes.
//   * Half of the waves (4..7, one per SIMD) build the next tables BEFORE their look-ups, the other half after: a step no longer starts
//     with all eight waves waiting for their first table entries at the same moment.
//   * Tables are written with ds_write_addtid_b32 (a wave stores one 256-byte entry per instruction, no address register: 2 cycles
//     against 6 for the ds_write_b64 form, MI355X_MICROARCH.md LDS table).
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
// a lane index the optimiser cannot see through: what is computed from it stays where it is used (the epilogues' per-lane constants would
// otherwise be hoisted out of the job loop and held in registers through the look-up stream, which has none to spare)
__device__ __forceinline__ int m7_opaque(int x) { asm volatile("" : "+v"(x)); return x; }

template <int R, int LOOKP, int CH, int NS>
__global__ __launch_bounds__(64 * M7_WAVES) void k_commutes_m4r7s(const uint8_t *__restrict__ A7, i64 Npad, i64 N, const u64 *__restrict__ BT, i64 Mw_pad,
                                                                   const u64 *__restrict__ steptab, const u32 *__restrict__ np_ptr,
                                                                   void *__restrict__ out_v, i64 out_stride, i64 m_cols, int bytes,
                                                                   u64 *__restrict__ part, i64 n_rt, i64 n_tiles, int stream, int order, u64 *dbg) {
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
    const bool build_first = (order & 1) != 0 && wave >= 4;           // uniform

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
        const u32 voff = (u32)(((order & 2) ? 0 : tile_row0) + (i64)wave * (4 * R) + slot * R);   // this slot's first row (A7 holds a byte per row)

        u32x4 acc[R];                                                  // 128 result bits of a row: four consecutive registers
#pragma unroll
        for (int j = 0; j < R; ++j) acc[j] = u32x4{0, 0, 0, 0};
        u32 ia[NS][QC], ib[NS][QC];

        // staging: wave w < 7 fetches row w of both tables' seven BT rows, a dword per lane (the row is uniform: scalar base + lane offset).
        // The loads are asm like the index loads (the compiler would add a 64-bit per-lane base and drain vmcnt around them); they are the
        // oldest loads of their step, so they have landed once at most the (NS - 1) LPC index loads requested last are outstanding.
        auto stage_load = [&](u32 (&v)[2], u64x2s bt_off, bool have_bt) {
            if (wave < 7 && have_bt) {
                const u32 lane4 = (u32)m7_opaque(lane) * 4;
                const i64 tw = (order & 4) ? 0 : tile_w0;
                m7_gload_fresh<0>(v[0], lane4, scalar_u64(reinterpret_cast<u64>(BT + bt_off.x + tw)));
                m7_gload_fresh<0>(v[1], lane4, scalar_u64(reinterpret_cast<u64>(BT + bt_off.y + tw)));
            }
        };
        auto stage_store = [&](u32 (&v)[2], u32 bt_slot, bool have_bt, auto in_loop) {
            if (wave < 7 && have_bt) {
                if constexpr (decltype(in_loop)::value) m7_vmwait<(NS - 1) * LPC>(v[0], v[1]);
                else m7_vmwait<0>(v[0], v[1]);                        // job prologue: these loads are the youngest
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
            u32 e = ((hi & 1u) ? brow[5] : 0u) ^ ((hi & 2u) ? brow[6] : 0u);
            const u32 rt_part = (u32)tb * M7_TABLE_BYTES + hi * (32 * M7_ENTRY_BYTES);
            if (buf == 0) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 1" ::"s"(rt_part));
                static_for<0, 32>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if constexpr (s > 0) e = (u32)m7_opaque((int)(e ^ brow[__builtin_ctz(s)]));   // one live value: the chain is not to be turned into a tree
                    m7_write_addtid<(s ^ (s >> 1)) * M7_ENTRY_BYTES>(e);
                });
            } else {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 1" ::"s"(rt_part + 8188u));
                static_for<0, 32>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if constexpr (s > 0) e = (u32)m7_opaque((int)(e ^ brow[__builtin_ctz(s)]));
                    m7_write_addtid<M7_BUF_BYTES - 8188 + (s ^ (s >> 1)) * M7_ENTRY_BYTES>(e);
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
        // chunk k has landed when at most the NS - 2 chunks requested after it are outstanding (loads return in order)
        auto chunk_wait = [&](auto kc) {
            constexpr int set = decltype(kc)::value % NS;
            if constexpr (QC == 2) m7_vmwait<(NS - 2) * LPC>(ia[set][0], ia[set][1], ib[set][0], ib[set][1]);
            else m7_vmwait<(NS - 2) * LPC>(ia[set][0], ib[set][0]);
        };
        // One step's look-ups.  The reads and the waits for them are written by hand: LDS operations complete in order, so "at most
        // 2 (LOOKP - 1) outstanding" means the reads of the row being folded have landed; anything the compiler issues itself around this
        // block only makes the waits stricter.  E(k): wait for chunk k's index bytes, then request chunk k + NS - 1 (of this step, or of the
        // next one — its set was last used by chunk k - 1, whose reads have all been issued).
        auto lookups = [&](u32 buf, u64 sa, u64 sb, u64 sa_next, u64 sb_next, bool more) {
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
                else if (more) prefetch(std::integral_constant<int, k + NS - 1 - NCH>{}, sa_next, sb_next);
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
            const u64x2s a7_0 = tab2[(i64)k0 * 8];
            u64 sa = scalar_u64(a7 + a7_0.x), sb = scalar_u64(a7 + a7_0.y);
            static_for<0, NS - 1>([&](auto cc) { prefetch_fresh(cc, sa, sb); });
            u32 st[2] = {0, 0};
            stage_load(st, tab2[(i64)k0 * 8 + 1 + wrow], true);
            stage_store(st, 0, true, std::false_type{});
            __syncthreads();
            build(0, 0);
            const bool has1 = k0 + 1 < k1;
            stage_load(st, tab2[(i64)(k0 + 1) * 8 + 1 + wrow], has1);
            stage_store(st, 1, has1, std::false_type{});
            // the scalars of a step are fetched one step ahead (a scalar load takes ~300 cycles): step t needs the index offsets of pair t + 1 and
            // this wave's BT offsets of pair t + 2
            u64x2s a7_next = tab2[(i64)(k0 + 1) * 8], bt_next = tab2[(i64)(k0 + 2) * 8 + 1 + wrow];
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
                const u64x2s a7_1 = a7_next, bt_2 = bt_next;
                a7_next = tab2[(i64)(t + 2) * 8];                     // for the next step; entries S and S + 1 exist (zero group)
                bt_next = tab2[(i64)(t + 3 < S + 2 ? t + 3 : S + 1) * 8 + 1 + wrow];
                u64 sa_next = scalar_u64(a7 + a7_1.x), sb_next = scalar_u64(a7 + a7_1.y);
                stage_load(st, bt_2, more2);
                M7_STAMP(c1);
                if (more && build_first) build((u + 1) & 1u, (u + 1) & 1u);
                M7_STAMP(c2);
                lookups(u & 1u, sa, sb, sa_next, sb_next, more);
                M7_STAMP(c3);
                if (more && !build_first) build((u + 1) & 1u, (u + 1) & 1u);
                M7_STAMP(c4);
                stage_store(st, u & 1u, more2, std::true_type{});
                sa = sa_next; sb = sb_next;
                M7_STAMP(c5);
                __syncthreads();
#ifdef SYMGPU_M7_STAMPS
                if (stamp) {
                    const u64 c6 = __builtin_amdgcn_s_memtime();
                    tm[0] += c1 - c0; tm[1] += c2 - c1; tm[2] += c3 - c2; tm[3] += c4 - c3; tm[4] += c5 - c4; tm[5] += c6 - c5;
                }
#endif
            }
#ifdef SYMGPU_M7_STAMPS
            if (stamp && lane == 0 && k0 == 0 && k1 == S) {
                for (int q = 0; q < 6; ++q) dbg[wave * 8 + q] = tm[q];
                dbg[wave * 8 + 6] = S;
                dbg[wave * 8 + 7] = __builtin_amdgcn_s_memrealtime();
            }
#endif
        }

        const bool whole = (k0 == 0 && k1 == S);
        const int lane_e = m7_opaque(lane), slot_e = lane_e >> 4, wp_e = lane_e & 15;
        if (!whole) {
            // part of a split tile: the raw accumulators (no NOT) -> scratch [workgroup][head 0 / tail 1][row of the tile][32 words]
            u64 *dst = part + ((i64)blockIdx.x * 2 + (k0 == 0 ? 1 : 0)) * PART_WORDS + ((i64)wave * (4 * R) + slot_e * R) * M7_TILE_W + 2 * wp_e;
#pragma unroll
            for (int j = 0; j < R; ++j) *reinterpret_cast<u32x4 *>(dst + (i64)j * M7_TILE_W) = acc[j];
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
                    const i64 col = (tile_w0 << 6) + half * 1024 + lane_e * 16;           // m_cols % 16 == 0: whole chunks in or out
                    if (i < N && col < m_cols) {
                        const u32 b16 = *reinterpret_cast<const uint16_t *>(region + (s * RO + jj) * M7_ENTRY_BYTES + half * 128 + lane_e * 2);
                        u32x4 v;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const u32 x = (b16 >> (4 * k)) & 0xFu;
                            v[k] = (x | (x << 7) | (x << 14) | (x << 21)) & 0x01010101u;
                        }
                        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out + i * out_stride + col));
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

// the split tiles of a stream-K launch: boundary b lies between workgroups b and b + 1; tail part of b XOR head part of b + 1 -> output
template <int R>
__global__ __launch_bounds__(256) void k_m7_fixup(const u64 *__restrict__ part, const u32 *__restrict__ np_ptr, i64 n_rt, i64 n_tiles, int P, i64 N,
                                                  void *__restrict__ out_v, i64 out_stride, i64 m_cols, int bytes) {
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
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out + i * out_stride + col));
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
static int launch_m7s(const uint8_t *A7, i64 Npad, i64 N, const u64 *BT, i64 Mw_pad, const u64 *steptab, const u32 *np, void *out, i64 stride, i64 M, bool bytes) {
    const bool attr = SG_DEVICE_ONCE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_commutes_m4r7s<R, LOOKP, CH, NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         M7S_LDS) == hipSuccess);
    if (!attr) { set_error("commutes_m4r7: %d bytes of LDS refused", M7S_LDS); return SYMGPU_E_HIP; }
    constexpr i64 WG_ROWS = 4 * M7_WAVES * R;
    const i64 n_rt = Npad / WG_ROWS, n_ct = Mw_pad / M7_TILE_W, n_tiles = n_rt * n_ct;
    const int P = ctx().num_cu;
    int order = 1;
    if (const char *e = getenv("SYMGPU_M4R_ORDER")) order = atoi(e);
    bool stream = n_tiles >= P;
    if (const char *e = getenv("SYMGPU_M4R_STREAM")) stream = stream && atoi(e) != 0;
    Scratch part, dbgbuf;
    if (stream) SG_TRY(part.alloc((size_t)P * 2 * WG_ROWS * M7_TILE_W * 8));
    u64 *dbg = nullptr;
    if (getenv("SYMGPU_M4R_DBG2")) { SG_TRY(dbgbuf.alloc(64 * 8)); HIP_TRY(hipMemsetAsync(dbgbuf.p, 0, 64 * 8, ctx().stream)); dbg = dbgbuf.as<u64>(); }
    SG_REQUIRE(n_tiles < (i64)1 << 31, "commutes_m4r7: tile count");
    hipLaunchKernelGGL((k_commutes_m4r7s<R, LOOKP, CH, NS>), dim3((unsigned)(stream ? P : n_tiles)), dim3(64 * M7_WAVES), M7S_LDS, ctx().stream, A7, Npad, N, BT, Mw_pad,
                       steptab, np, out, stride, M, bytes ? 1 : 0, part.as<u64>(), n_rt, n_tiles, stream ? 1 : 0, order, dbg);
    KERNEL_CHECK();
    if (dbg) {
        u64 h[64];
        HIP_TRY(hipMemcpyAsync(h, dbg, sizeof h, hipMemcpyDeviceToHost, ctx().stream));
        HIP_TRY(hipStreamSynchronize(ctx().stream));
        for (int w = 0; w < 8; ++w)
            fprintf(stderr, "m7 timing wave %d (cycles per step, S=%llu): stage %.0f build-first %.0f lookups %.0f build-after %.0f store %.0f barrier %.0f\n", w, (unsigned long long)h[w * 8 + 6],
                    (double)h[w * 8] / (double)(h[w * 8 + 6] ? h[w * 8 + 6] : 1), (double)h[w * 8 + 1] / (double)(h[w * 8 + 6] ? h[w * 8 + 6] : 1), (double)h[w * 8 + 2] / (double)(h[w * 8 + 6] ? h[w * 8 + 6] : 1),
                    (double)h[w * 8 + 3] / (double)(h[w * 8 + 6] ? h[w * 8 + 6] : 1), (double)h[w * 8 + 4] / (double)(h[w * 8 + 6] ? h[w * 8 + 6] : 1), (double)h[w * 8 + 5] / (double)(h[w * 8 + 6] ? h[w * 8 + 6] : 1));
    }
    if (stream) {
        hipLaunchKernelGGL((k_m7_fixup<R>), dim3((unsigned)(P - 1), 24), dim3(256), 0, ctx().stream, part.as<u64>(), np, n_rt, n_tiles, P, N, out, stride, M, bytes ? 1 : 0);
        KERNEL_CHECK();
    }
    return SYMGPU_OK;
}

// prepared operands of one call (A7, flags, klist) and the launch; bt_p = bit-major copy of B (built / cached by commutes_m4r_dev)
int commutes_m4r7_launch(const u64 *A, i64 N, i64 M, int Wq, const u64 *bt_p, i64 Mw_pad, int R, bool bytes, void *dst, i64 stride) {
    hipStream_t st = ctx().stream;
    if (!getenv("SYMGPU_M4R_OLD") && R == 40) R = 48;
    const int W = 2 * Wq, ng7 = (128 * Wq + 6) / 7;
    const i64 Npad = (N + (i64)4 * M7_WAVES * R - 1) / ((i64)4 * M7_WAVES * R) * ((i64)4 * M7_WAVES * R);   // multiples of 256
    Scratch a7, flags, klist;
    SG_TRY(a7.alloc((size_t)(ng7 + 1) * Npad));
    SG_TRY(flags.alloc((size_t)(ng7 + 1) * 4));
    SG_TRY(klist.alloc((size_t)(ng7 + 2) * 4));
    HIP_TRY(hipMemsetAsync(flags.p, 0, (size_t)(ng7 + 1) * 4, st));
    hipLaunchKernelGGL(k_m7_a7, dim3((unsigned)(Npad / 256), (unsigned)((W + A7_CW - 1) / A7_CW)), dim3(256), 0, st, A, N, W, ng7, a7.as<uint8_t>(), Npad, flags.as<u32>());
    KERNEL_CHECK();
    u32 *np = flags.as<u32>() + ng7;
    hipLaunchKernelGGL(k_m7_klist, dim3(1), dim3(64), 0, st, flags.as<u32>(), ng7, klist.as<u32>(), np);
    KERNEL_CHECK();
    Scratch steptab;
    const int max_pairs = (ng7 + 1) / 2;
    SG_TRY(steptab.alloc((size_t)(max_pairs + 2) * 16 * 8));
    hipLaunchKernelGGL(k_m7_steptab, dim3((unsigned)((max_pairs + 2 + 63) / 64)), dim3(64), 0, st, klist.as<u32>(), np, ng7, max_pairs, Npad, Mw_pad, Wq, steptab.as<u64>());
    KERNEL_CHECK();
    ProfScope prof(1);
    if (!getenv("SYMGPU_M4R_OLD")) {
#define M7S_ARGS a7.as<uint8_t>(), Npad, N, bt_p, Mw_pad, steptab.as<u64>(), np, dst, stride, M, bytes
        if (R == 48) SG_TRY((launch_m7s<48, 3, 8, 3>(M7S_ARGS)));
        else if (R == 24) SG_TRY((launch_m7s<24, 4, 8, 3>(M7S_ARGS)));
        else SG_TRY((launch_m7s<16, 4, 4, 4>(M7S_ARGS)));
#undef M7S_ARGS
        return SYMGPU_OK;
    }
#define M7_ARGS a7.as<uint8_t>(), Npad, N, bt_p, Mw_pad, Wq, klist.as<u32>(), np, dst, stride, M
#define M7_LAUNCH(BY)                                                  \
    if (R == 48) SG_TRY((launch_m7<48, 3, BY>(M7_ARGS)));              \
    else if (R == 40) SG_TRY((launch_m7<40, 4, BY>(M7_ARGS)));         \
    else if (R == 24) SG_TRY((launch_m7<24, 4, BY>(M7_ARGS)));         \
    else SG_TRY((launch_m7<16, 4, BY>(M7_ARGS)));
    if (bytes) { M7_LAUNCH(true) } else { M7_LAUNCH(false) }
#undef M7_LAUNCH
#undef M7_ARGS
    return SYMGPU_OK;
}

}  // namespace symgpu
