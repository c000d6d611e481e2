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

namespace symgpu {

typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
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

// group-major copy of A (zero padded to Npad rows, plus the all-zero group NG7) + "group has a non-zero value" flags
__global__ __launch_bounds__(256) void k_m7_a7(const u64 *__restrict__ rows, i64 N, int W, int ng7, uint8_t *__restrict__ A7, i64 Npad, u32 *__restrict__ flags) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= Npad) return;                                            // Npad is a multiple of 256: whole waves leave together
    const int lane = threadIdx.x & 63;
    for (int g = 0; g < ng7; ++g) {
        const int bit0 = 7 * g, w = bit0 >> 6, sh = bit0 & 63;
        u64 v = 0;
        if (i < N) {
            v = rows[i * W + w] >> sh;
            if (sh > 57 && w + 1 < W) v |= rows[i * W + w + 1] << (64 - sh);
        }
        const u32 val = (u32)v & 0x7Fu;
        A7[(i64)g * Npad + i] = (uint8_t)val;
        const u64 any = __ballot(val != 0);
        if (any && lane == 0) flags[g] = 1u;                          // benign race: every writer stores 1
    }
    A7[(i64)ng7 * Npad + i] = 0;                                      // the padding group
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

// prepared operands of one call (A7, flags, klist) and the launch; bt_p = bit-major copy of B (built / cached by commutes_m4r_dev)
int commutes_m4r7_launch(const u64 *A, i64 N, i64 M, int Wq, const u64 *bt_p, i64 Mw_pad, int R, bool bytes, void *dst, i64 stride) {
    hipStream_t st = ctx().stream;
    const int W = 2 * Wq, ng7 = (128 * Wq + 6) / 7;
    const i64 Npad = (N + (i64)4 * M7_WAVES * R - 1) / ((i64)4 * M7_WAVES * R) * ((i64)4 * M7_WAVES * R);   // multiples of 256
    Scratch a7, flags, klist;
    SG_TRY(a7.alloc((size_t)(ng7 + 1) * Npad));
    SG_TRY(flags.alloc((size_t)(ng7 + 1) * 4));
    SG_TRY(klist.alloc((size_t)(ng7 + 2) * 4));
    HIP_TRY(hipMemsetAsync(flags.p, 0, (size_t)(ng7 + 1) * 4, st));
    hipLaunchKernelGGL(k_m7_a7, dim3((unsigned)(Npad / 256)), dim3(256), 0, st, A, N, W, ng7, a7.as<uint8_t>(), Npad, flags.as<u32>());
    KERNEL_CHECK();
    u32 *np = flags.as<u32>() + ng7;
    hipLaunchKernelGGL(k_m7_klist, dim3(1), dim3(64), 0, st, flags.as<u32>(), ng7, klist.as<u32>(), np);
    KERNEL_CHECK();
    ProfScope prof(1);
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
