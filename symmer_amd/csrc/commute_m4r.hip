// commute_m4r.hip — termwise commutation as a GF(2) matrix product by the Method of Four Russians, tables in LDS.
// (reference: symmer/operators/base.py:938-971 -> matmul_GF2 / numba_dot_matmal_GF2, utils.py:9-78: f64 dgemm, then % 2)
//
//   C = NOT( A . Omega . B^T  mod 2 ),  A: N x 2n bits, B: M x 2n bits       (True = commute)
//
// The register-tile kernel of commute.hip spends 4 VALU instructions per pair and 64-bit word (256 lane-ops per pair at
// n = 2000) and is VALU-issue bound.  Here the contraction axis is cut into groups of 8 bits ("k-blocks" = one byte of an
// A row).  For a k-block and a tile of 2048 columns a workgroup tabulates all 256 XOR-combinations of the 8 bit-rows of
// the (bit-transposed, X/Z-swapped) right operand in LDS: 256 entries x 256 B = 64 KiB.  A row of A then needs ONE
// ds_read_b128 per lane (entry = its byte; a wave instruction serves 4 rows x 2048 columns) and 4 v_xor + 1 v_perm per
// 8 contraction bits and 128 pairs per lane — 1/13 of the VALU work per pair.  The next table is built into the second
// 64 KiB buffer in the same iteration: one barrier per k-block.
//
// Bounds (tools/ubench_lds.hip, MI355X): the look-up stream runs at 5.1-5.5 cycles per ds_read_b128 wave instruction and CU
// (LDS peak 4.0; without the v_perm 4.3: the 5 VALU instructions per read are the co-limit), a 64 KiB table costs 890
// cycles of ds_write_b64 (1180 with its XOR chain).  Per k-block and 1536-row workgroup: 384 x 5.1 + 1100 = 3050 cycles;
// measured in the kernel 3600-3860.  cfg5 slice (25,000 x 200,000, n = 2000): 5.8 ms = 8.6e11 pairs/s against 17.4 ms on
// the register-tile kernel; 52 TB/s of table reads = 33 % of the 157 TB/s LDS read peak.
//
// Layouts prepared per call (all tiny next to the N x M output):
//   A8[kb][i]   byte kb of packed row i (byte-major copy, i padded with zeros): the index bytes of a workgroup's rows for ONE
//               k-block are contiguous.
//   BT[c][jw]   bit c of B rows 64jw..64jw+63 (bit-major copy); row c of the contraction pairs A bit c with B bit c +- 64Wq
//               (x with z', z with x'), which is just a row offset into BT.
//   klist       the k-blocks in which A has any non-zero byte (all-zero bytes — padding above n, untouched qubits — look up
//               entry 0 = 0 and are skipped for the whole launch).
//
// Wave = 4 row slots (16 lanes each) x R rows per slot; lane = (slot, word pair): accumulators acc[R][2] u64 in VGPRs
// (R = 48: 192 VGPRs; 8 waves per CU x 256 VGPRs = the whole register file).  ds_read_b128's four 16-lane service groups
// take lanes from different slots at disjoint word pairs, and the entry stride is exactly 256 B = 64 banks, so look-ups of
// four different entries are conflict free (SQ_LDS_IDX_ACTIVE = the ideal 4 cycles per read + 8 per ds_write2_b64).
// Table index -> LDS address is ONE v_perm_b32 (byte of the index dword -> byte 1 of the address, lane offset in byte 0,
// buffer select in byte 2).
//
// Measured and rejected: per-wave global loads of the BT rows / index bytes instead of the LDS hand-off (7.0 ms instead of
// 5.8: 13 vector loads per wave and k-block, all L1 hits, cost more than the look-ups), 16 waves x 16 rows (7.1 ms: less
// amortisation of the table), the two halves of the workgroup running look-ups and build in opposite order (6.7 ms), a
// separate bits -> bytes expansion kernel (+1.2 ms: now the epilogue), two tables per iteration read together and folded with one
// v_bitop3 XOR3 per dword (3 instead of 5 VALU instructions per read, but no double buffering and a second barrier: 6.0 ms).
#include "common.h"
#include <stdlib.h>

namespace symgpu {

typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u64x2 lds_u64x2;            // raw LDS address -> ds_read_b128 without a base add

constexpr int MK_TILE_W = 32;                         // 64-bit words per column tile: 2048 columns
constexpr int MK_ENTRY_BYTES = MK_TILE_W * 8;         // 256
constexpr int MK_TABLE_BYTES = 256 * MK_ENTRY_BYTES;  // 64 KiB per k-block
constexpr int MK_LDS = 2 * MK_TABLE_BYTES;            // double buffered
constexpr int MK_BT_STAGE = 8 * MK_ENTRY_BYTES;       // the 8 bit-rows of one k-block: 2 KiB
constexpr int mk_lds_bytes(int wg_rows) { return MK_LDS + 2 * MK_BT_STAGE + 2 * wg_rows; }

// ---- operand preparation ---------------------------------------------------------------------------------------------
// byte-major copy of A (zero padded to Npad rows) + "k-block has a non-zero byte" flags
__global__ __launch_bounds__(256) void k_m4r_a8(const u64 *__restrict__ rows, i64 N, int W, uint8_t *__restrict__ A8, i64 Npad, u32 *__restrict__ flags) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= Npad) return;                                            // Npad is a multiple of 256: whole waves leave together
    const int lane = threadIdx.x & 63;
    for (int w = 0; w < W; ++w) {
        const u64 v = (i < N) ? rows[i * W + w] : 0ULL;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const u32 byte = (u32)(v >> (8 * b)) & 0xFFu;
            A8[(i64)(8 * w + b) * Npad + i] = (uint8_t)byte;
            const u64 any = __ballot(byte != 0);
            if (any && lane == 0) flags[8 * w + b] = 1u;              // benign race: every writer stores 1
        }
    }
}

// compact the flagged k-blocks (single wave; order ascending)
__global__ __launch_bounds__(64) void k_m4r_klist(const u32 *__restrict__ flags, int nkb, u32 *__restrict__ klist, u32 *__restrict__ nk) {
    const int lane = threadIdx.x;
    u32 count = 0;
    for (int base = 0; base < nkb; base += 64) {
        const int kb = base + lane;
        const bool on = kb < nkb && flags[kb] != 0;
        const u64 m = __ballot(on);
        if (on) klist[count + __popcll(m & ((1ULL << lane) - 1))] = (u32)kb;
        count += (u32)__popcll(m);
    }
    if (lane == 0) *nk = count;
}

// bit-major copy of B: BT[c][jw], c = 64*sw + bit.  One wave transposes eight 64 x 64 bit blocks of one source word and
// writes 64 contiguous bytes per bit-row.
__global__ __launch_bounds__(256) void k_m4r_bt(const u64 *__restrict__ rows, i64 M, int W, u64 *__restrict__ BT, i64 Mw_pad) {
    const int lane = threadIdx.x & 63;
    const i64 group = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);       // group of 8 column words
    const int sw = blockIdx.y;
    if (group * 8 >= Mw_pad) return;
    u64 mine[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const i64 t = (group * 8 + g) * 64 + lane;
        const u64 word = (t < M) ? rows[t * W + sw] : 0ULL;
        u64 m = 0;
        for (int b = 0; b < 64; ++b) {
            const u64 bal = __ballot((word >> b) & 1ULL);
            if (lane == b) m = bal;
        }
        mine[g] = m;
    }
    u64 *dst = BT + ((i64)64 * sw + lane) * Mw_pad + group * 8;
#pragma unroll
    for (int g = 0; g < 8; ++g) dst[g] = mine[g];
}

// ---- main kernel ---------------------------------------------------------------------------------------------------
// R = rows per 16-lane slot, WAVES = waves per workgroup (8: 256 VGPRs each, 16: 128), LOOK = ds_read_b128 in flight per wave
// BYTES: the epilogue expands the result bits to np.bool_ bytes itself (out_bits is then the uint8 output, out_stride = M,
// M % 16 == 0 and a 16-byte aligned base); otherwise it stores bit-packed rows.
template <int R, int WAVES, int LOOK, bool BYTES>
__global__ __launch_bounds__(64 * WAVES) void k_commutes_m4r(const uint8_t *__restrict__ A8, i64 Npad, i64 N, const u64 *__restrict__ BT, i64 Mw_pad,
                                                              int Wq, const u32 *__restrict__ klist, const u32 *__restrict__ nk_ptr,
                                                              u64 *__restrict__ out_bits, i64 out_stride, i64 m_cols) {
    // out_bits / out_stride: output base at this launch's first column and row stride (u64 words, or bytes if BYTES);
    // m_cols: valid columns from there
    static_assert(R % 8 == 0, "indices arrive as dwordx2 = 8 rows");
    extern __shared__ __attribute__((aligned(16))) uint8_t m4r_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = lane >> 4, wp = lane & 15;
    const i64 row0 = (i64)blockIdx.x * (4 * WAVES * R) + (i64)wave * (4 * R) + slot * R;
    const i64 tile_w0 = (i64)blockIdx.y * MK_TILE_W;
    const u32 nk = *nk_ptr;

    u64 acc[R][2];
#pragma unroll
    for (int j = 0; j < R; ++j) acc[j][0] = acc[j][1] = 0;

    // table builder: lane (h, w) writes word w of the NG = 128 / WAVES entries (2*wave + h) * NG + g (Gray order over g)
    constexpr int NG = 128 / WAVES, GB = (NG == 16) ? 4 : 3;         // 8 waves: 16 entries per lane, 16 waves: 8
    static_assert(WAVES == 8 || WAVES == 16, "256 entries over 2 * WAVES half waves");
    constexpr int WG_ROWS = 4 * WAVES * R;
    const int bh = lane >> 5, bw = lane & 31;
    const u32 ent_hi = (u32)wave * 2 + bh;                           // entry bits GB..7
    const i64 half_bits = (i64)64 * Wq;
    const u32 look_base = (u32)wp * 16;
    // the look-up address is assembled bytewise (v_perm_b32), which needs the tables at LDS offset 0: m4r_lds is the kernel's only LDS object
    if ((u32)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)m4r_lds != 0) __builtin_trap();

    // Per k-block a workgroup needs 8 bit-rows x 256 B of BT and one index byte per row of A8: 2 KiB + WG_ROWS bytes.  Every
    // wave needs all of it, so it is fetched ONCE (waves 0..3, one 16-byte global load per lane) and handed round through two
    // small double-buffered LDS slots behind the tables — per-wave global loads of the same bytes (8 + R/8 vector loads per
    // wave and k-block, all L1 hits) cost more time than the look-ups themselves.
    uint8_t *const bt_stage = m4r_lds + MK_LDS;                      // [2][8 rows][256 B]
    uint8_t *const ix_stage = bt_stage + 2 * MK_BT_STAGE;            // [2][WG_ROWS] bytes
    const int tid = threadIdx.x;
    const bool stage_bt = tid < 128, stage_ix = tid >= 128 && tid < 128 + WG_ROWS / 16;   // wave-uniform except the last idx wave
    const u64 *bt_src = BT + tile_w0 + 2 * (tid & 15);               // + (c0 + tid / 16) * Mw_pad
    const uint8_t *ix_src = A8 + (i64)blockIdx.x * WG_ROWS + 16 * (tid - 128);

    auto bt_row0 = [&](u32 kb) -> i64 {
        // contraction byte kb of A pairs with the swapped half of B: x bits with z', z bits with x'
        return (kb < 8u * (u32)Wq) ? half_bits + 8 * (i64)kb : 8 * ((i64)kb - 8 * Wq);
    };
    auto stage_load = [&](u32 kb_bt, bool have_bt, u32 kb_ix, bool have_ix) -> u32x4 {
        u32x4 v = {0, 0, 0, 0};
        if (stage_bt && have_bt) v = *reinterpret_cast<const u32x4 *>(bt_src + (bt_row0(kb_bt) + (tid >> 4)) * Mw_pad);
        if (stage_ix && have_ix) v = *reinterpret_cast<const u32x4 *>(ix_src + (i64)kb_ix * Npad);
        return v;
    };
    auto stage_store = [&](u32x4 v, u32 bt_slot, bool have_bt, u32 ix_slot, bool have_ix) {
        if (stage_bt && have_bt) *reinterpret_cast<u32x4 *>(bt_stage + bt_slot * MK_BT_STAGE + 16 * tid) = v;
        if (stage_ix && have_ix) *reinterpret_cast<u32x4 *>(ix_stage + ix_slot * WG_ROWS + 16 * (tid - 128)) = v;
    };
    auto build = [&](u32 buf, u32 bt_slot) {
        u64 brow[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) brow[r] = *reinterpret_cast<const u64 *>(bt_stage + bt_slot * MK_BT_STAGE + r * 256 + bw * 8);
        u64 e = 0;
#pragma unroll
        for (int r = GB; r < 8; ++r) e ^= ((ent_hi >> (r - GB)) & 1u) ? brow[r] : 0ULL;
        uint8_t *dst = m4r_lds + buf * MK_TABLE_BYTES + (ent_hi * NG) * MK_ENTRY_BYTES + bw * 8;
        // Gray code over the low GB bits: one row flipped per step
        constexpr int flip[15] = {0, 1, 0, 2, 0, 1, 0, 3, 0, 1, 0, 2, 0, 1, 0};
        int g = 0;
        *reinterpret_cast<u64 *>(dst) = e;
#pragma unroll
        for (int s = 0; s < NG - 1; ++s) {
            g ^= 1 << flip[s];
            e ^= brow[flip[s]];
            *reinterpret_cast<u64 *>(dst + g * MK_ENTRY_BYTES) = e;
        }
    };
    auto lookups = [&](u32 buf, u32 ix_slot) {
        const u32 base = look_base | (buf << 16);
        u32x2 idx[R / 8];
        const uint8_t *ix = ix_stage + ix_slot * WG_ROWS + wave * (4 * R) + slot * R;
#pragma unroll
        for (int q = 0; q < R / 8; ++q) idx[q] = *reinterpret_cast<const u32x2 *>(ix + 8 * q);
        // rolling window of LOOK reads in flight (4 VGPRs each): read j + LOOK is issued right after the sum of row j, so
        // the LDS pipeline sees a continuous stream; the pins keep the optimiser from sinking the sums below all the reads
        // (which spills) and keep program order between sums and reads
        auto read = [&](int j) -> u64x2 {
            const u32 word = idx[j / 8][(j / 4) % 2];
            // address byte 0 = lane offset, byte 1 = table index (entry stride 256 B), byte 2 = buffer, byte 3 = 0
            const u32 addr = __builtin_amdgcn_perm(word, base, 0x0c020000u | ((4u + (j % 4)) << 8));
            return *reinterpret_cast<const lds_u64x2 *>((uintptr_t)addr);
        };
        u64x2 v[LOOK];
#pragma unroll
        for (int b = 0; b < LOOK; ++b) v[b] = read(b);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            acc[j][0] ^= v[j % LOOK].x;
            acc[j][1] ^= v[j % LOOK].y;
            asm volatile("" : "+v"(acc[j][0]), "+v"(acc[j][1]));
            if (j + LOOK < R) v[j % LOOK] = read(j + LOOK);
        }
        // tell the machine scheduler to keep that order (it would otherwise batch the reads again): the index reads, LOOK
        // reads, then {4 v_xor + 1 v_perm, 1 read} per row
        __builtin_amdgcn_sched_group_barrier(0x100, R / 8, 0);
#pragma unroll
        for (int b = 0; b < LOOK; ++b) { __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
        for (int j = 0; j < R - LOOK; ++j) { __builtin_amdgcn_sched_group_barrier(0x002, 5, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x002, 4 * LOOK, 0);
    };

    // Iteration t: look-ups on table t (buffer t&1, indices in slot t&1) while table t+1 is built from BT slot (t+1)&1; the
    // staging waves fetch BT rows of k-block t+2 -> slot t&1 and the indices of k-block t+1 -> slot (t+1)&1, both read
    // only after the barrier that ends the iteration (and last read before the barrier that started it).
    if (nk > 0) {
        const u32 k0 = klist[0], k1 = nk > 1 ? klist[1] : 0;
        stage_store(stage_load(k0, true, k0, true), 0, true, 0, true);
        __syncthreads();
        build(0, 0);
        stage_store(stage_load(k1, nk > 1, 0, false), 1, nk > 1, 0, false);
        __syncthreads();
        u32 kb_ix = k1, kb_bt = nk > 2 ? klist[2] : 0;               // k-blocks t+1 and t+2
        for (u32 t = 0; t < nk; ++t) {
            const bool more = t + 1 < nk, more2 = t + 2 < nk;        // uniform
            const u32x4 st = stage_load(kb_bt, more2, kb_ix, more);
            const u32 kb_next = t + 3 < nk ? klist[t + 3] : 0;
            // (measured: running the two phases in opposite order on the two halves of the workgroup, so that the look-ups of
            // one half overlap the table build of the other, is slower — 6.7 ms instead of 5.85 ms on the cfg5 slice)
            lookups(t & 1u, t & 1u);
            if (more) build((t + 1) & 1u, (t + 1) & 1u);
            stage_store(st, t & 1u, more2, (t + 1) & 1u, more);
            kb_ix = kb_bt;
            kb_bt = kb_next;
            __syncthreads();
        }
    }

    if constexpr (BYTES) {
        // One byte per pair, 16-byte stores, a wave store = 1 KiB of one output row.  A lane holds 128 result bits of a row but
        // must write 16 columns of it: the rows are turned round through the (now free) table area, 128 KiB / WAVES per wave =
        // 4 * RP rows x 2048 bits per pass.
        uint8_t *const out = reinterpret_cast<uint8_t *>(out_bits);
        constexpr int RP = 128 / WAVES;                              // rows per slot and pass: WAVES x 4 RP x 256 B = 128 KiB
        uint8_t *const region = m4r_lds + wave * (4 * RP * MK_ENTRY_BYTES);
        const i64 wave_row0 = (i64)blockIdx.x * WG_ROWS + (i64)wave * (4 * R);
        __syncthreads();                                             // every wave is done with the tables
#pragma unroll
        for (int p0 = 0; p0 < R; p0 += RP) {
#pragma unroll
            for (int jj = 0; jj < RP; ++jj) {
                if (p0 + jj < R) {
                    const u64x2 v = {~acc[p0 + jj][0], ~acc[p0 + jj][1]};                       // commute = NOT parity
                    *reinterpret_cast<u64x2 *>(region + (slot * RP + jj) * MK_ENTRY_BYTES + wp * 16) = v;
                }
            }
            __syncthreads();
            const int rows_here = (R - p0 < RP) ? R - p0 : RP;
            for (int q = 0; q < 4 * rows_here * 2; ++q) {
                const int half = q & 1, rl = q >> 1, s = rl / rows_here, jj = rl - s * rows_here;
                const i64 i = wave_row0 + s * R + p0 + jj;
                const i64 col = (tile_w0 << 6) + half * 1024 + lane * 16;           // m_cols % 16 == 0: whole chunks in or out
                if (i < N && col < m_cols) {
                    const u32 b16 = *reinterpret_cast<const uint16_t *>(region + (s * RP + jj) * MK_ENTRY_BYTES + half * 128 + lane * 2);
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

// bit-packed rows -> np.bool_ bytes: one thread per 16 columns
template <bool VEC>
__global__ __launch_bounds__(256) void k_bits_to_bytes(const u64 *__restrict__ bits, i64 stride_words, i64 N, i64 M, uint8_t *__restrict__ out) {
    const i64 n16 = (M + 15) / 16;
    const i64 total = N * n16;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
        const i64 i = idx / n16, c = idx - i * n16;
        const u32 b16 = (u32)(bits[i * stride_words + (c >> 2)] >> (16 * (c & 3))) & 0xFFFFu;
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 x = (b16 >> (4 * q)) & 0xFu;
            v[q] = (x | (x << 7) | (x << 14) | (x << 21)) & 0x01010101u;
        }
        uint8_t *dst = out + i * M + 16 * c;
        if (VEC) {
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst));
        } else {
#pragma unroll
            for (int b = 0; b < 16; ++b)
                if (16 * c + b < M) dst[b] = (uint8_t)(v[b / 4] >> (8 * (b % 4)));
        }
    }
}

static i64 round_up_i64(i64 x, i64 m) { return (x + m - 1) / m * m; }

template <int R, int WAVES, int LOOK, bool BYTES>
static int launch_m4r(const uint8_t *A8, i64 Npad, i64 N, const u64 *BT, i64 Mw_pad, int Wq, const u32 *klist, const u32 *nk, void *out, i64 stride,
                      i64 M) {
    constexpr int lds = mk_lds_bytes(4 * WAVES * R);
    const bool attr = SG_DEVICE_ONCE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_commutes_m4r<R, WAVES, LOOK, BYTES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         lds) == hipSuccess);
    if (!attr) { set_error("commutes_m4r: %d bytes of LDS refused", lds); return SYMGPU_E_HIP; }
    const i64 gx = Npad / (4 * WAVES * R), gy = Mw_pad / MK_TILE_W;
    // blockIdx.x (fast) walks the row blocks: workgroups that run together share the BT column tile in L2
    for (i64 y0 = 0; y0 < gy; y0 += 65535) {
        const i64 ny = gy - y0 < 65535 ? gy - y0 : 65535;
        hipLaunchKernelGGL((k_commutes_m4r<R, WAVES, LOOK, BYTES>), dim3((unsigned)gx, (unsigned)ny), dim3(64 * WAVES), lds, ctx().stream, A8, Npad, N,
                           BT + y0 * MK_TILE_W, Mw_pad, Wq, klist, nk,
                           BYTES ? reinterpret_cast<u64 *>(static_cast<uint8_t *>(out) + y0 * MK_TILE_W * 64) : static_cast<u64 *>(out) + y0 * MK_TILE_W,
                           stride, M - y0 * MK_TILE_W * 64);
        KERNEL_CHECK();
    }
    return SYMGPU_OK;
}

// tile variants: {rows per slot, waves} -> rows per workgroup = 4 * waves * R.  Taller tiles amortise the 64 KiB table over
// more rows (cfg5 slice, n = 2000: R = 16 / 24 / 40 / 48 -> 9.8 / 7.8 / 6.2 / 5.8 ms) but need enough workgroups for 256 CUs.
struct M4rVariant { int R, waves; };
static i64 m4r_workgroups(i64 N, i64 M, int R) {
    const i64 Mw = (M + 63) / 64;
    return ((N + 32 * R - 1) / (32 * R)) * ((Mw + MK_TILE_W - 1) / MK_TILE_W);
}
static M4rVariant m4r_pick(i64 N, i64 M) {
    if (const char *e = getenv("SYMGPU_M4R_R")) {
        const int r = atoi(e);
        if (r == 16 || r == 24 || r == 40 || r == 48) return {r, 8};
        if (r == 116) return {16, 16};                               // 16 waves x 16 rows: measured slower than 8 x 40, kept for the tests
    }
    const int cand[3] = {48, 40, 24};
    for (int R : cand)
        if (m4r_workgroups(N, M, R) >= 3 * ctx().num_cu) return {R, 8};
    return {16, 8};
}
// enough 512 x 2048 tiles to occupy most of the chip (below that the register-tile kernel wins: 1024 x 16384 at n = 2000 takes
// 0.10 ms there and 0.6 ms here; 4096 x 65536: 1.03 ms against 0.66 ms)
// — and tiles that are at least half full in both directions: a 512 x 2048 tile costs the same whether it holds 1 row or 512
// (100,000 x 1 at n = 1000: 0.14 ms on the register-tile kernel, 0.34 ms here)
bool commutes_m4r_worthwhile(i64 N, i64 M) { return N >= 256 && M >= 1024 && m4r_workgroups(N, M, 16) >= (3 * ctx().num_cu) / 4; }

// Same contract as commutes_dev (commute.hip): exactly one of out / out_bits is non-null.
int commutes_m4r_dev(const u64 *A, i64 N, const u64 *B, i64 M, int Wq, uint8_t *out, u64 *out_bits, symgpu_op_s *b_owner) {
    if (N == 0 || M == 0) return SYMGPU_OK;
    hipStream_t st = ctx().stream;
    const int W = 2 * Wq, nkb = 16 * Wq;
    const M4rVariant var = m4r_pick(N, M);
    const int R = var.R;
    const i64 Npad = round_up_i64(N, (i64)4 * var.waves * R);            // 512 .. 1280: multiples of 256
    const i64 Mw = (M + 63) / 64, Mw_pad = round_up_i64(Mw, MK_TILE_W);
    Scratch a8, bt, flags, klist, bits;
    SG_TRY(a8.alloc((size_t)nkb * Npad));
    // the bit-major copy of B is cached on its operator (an adjacency matrix computed slab by slab transposes B once)
    const u64 *bt_p = nullptr;
    if (b_owner && b_owner->rows == B && b_owner->T == M) {
        if (!b_owner->bt || b_owner->bt_T != M || b_owner->bt_pad != Mw_pad) {
            if (b_owner->bt) { dev_free(b_owner->bt); b_owner->bt = nullptr; }
            SG_TRY(dev_alloc((size_t)64 * W * Mw_pad * 8, (void **)&b_owner->bt));
            hipLaunchKernelGGL(k_m4r_bt, dim3((unsigned)((Mw_pad / 8 + 3) / 4), (unsigned)W), dim3(256), 0, st, B, M, W, b_owner->bt, Mw_pad);
            KERNEL_CHECK();
            b_owner->bt_pad = Mw_pad;
            b_owner->bt_T = M;
        }
        bt_p = b_owner->bt;
    } else {
        SG_TRY(bt.alloc((size_t)64 * W * Mw_pad * 8));
        hipLaunchKernelGGL(k_m4r_bt, dim3((unsigned)((Mw_pad / 8 + 3) / 4), (unsigned)W), dim3(256), 0, st, B, M, W, bt.as<u64>(), Mw_pad);
        KERNEL_CHECK();
        bt_p = bt.p ? bt.as<u64>() : nullptr;
    }
    SG_TRY(flags.alloc((size_t)(nkb + 1) * 4));
    SG_TRY(klist.alloc((size_t)nkb * 4));
    HIP_TRY(hipMemsetAsync(flags.p, 0, (size_t)(nkb + 1) * 4, st));
    hipLaunchKernelGGL(k_m4r_a8, dim3((unsigned)(Npad / 256)), dim3(256), 0, st, A, N, W, a8.as<uint8_t>(), Npad, flags.as<u32>());
    KERNEL_CHECK();
    u32 *nk = flags.as<u32>() + nkb;
    hipLaunchKernelGGL(k_m4r_klist, dim3(1), dim3(64), 0, st, flags.as<u32>(), nkb, klist.as<u32>(), nk);
    KERNEL_CHECK();
    // np.bool_ output: expanded by the kernel's own epilogue when rows can be written with aligned 16-byte stores, otherwise
    // bit-packed rows to scratch + a separate expansion with byte stores
    const bool fused_bytes = out && (M % 16 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && !SG_TUNE("SYMGPU_M4R_UNFUSED");
    void *dst = out_bits;
    i64 stride = Mw;
    if (fused_bytes) {
        dst = out;
        stride = M;
    } else if (out) {
        SG_TRY(bits.alloc((size_t)N * Mw * 8));
        dst = bits.p;
    }
    {
        ProfScope prof(1);
#define M4R_ARGS a8.as<uint8_t>(), Npad, N, bt_p, Mw_pad, Wq, klist.as<u32>(), nk, dst, stride, M
#define M4R_LAUNCH(BY)                                                        \
        if (var.waves == 16) SG_TRY((launch_m4r<16, 16, 6, BY>(M4R_ARGS)));   \
        else if (R == 48) SG_TRY((launch_m4r<48, 8, 6, BY>(M4R_ARGS)));       \
        else if (R == 40) SG_TRY((launch_m4r<40, 8, 8, BY>(M4R_ARGS)));       \
        else if (R == 24) SG_TRY((launch_m4r<24, 8, 8, BY>(M4R_ARGS)));       \
        else SG_TRY((launch_m4r<16, 8, 8, BY>(M4R_ARGS)));
        if (fused_bytes) { M4R_LAUNCH(true) } else { M4R_LAUNCH(false) }
#undef M4R_LAUNCH
#undef M4R_ARGS
    }
    if (out && !fused_bytes) {
        const i64 total = N * ((M + 15) / 16);
        i64 g = (total + 255) / 256;
        if (g > 65536) g = 65536;
        const bool vec = (M % 16 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
        if (vec) hipLaunchKernelGGL(k_bits_to_bytes<true>, dim3((unsigned)g), dim3(256), 0, st, bits.as<u64>(), stride, N, M, out);
        else hipLaunchKernelGGL(k_bits_to_bytes<false>, dim3((unsigned)g), dim3(256), 0, st, bits.as<u64>(), stride, N, M, out);
        KERNEL_CHECK();
    }
    return SYMGPU_OK;
}

}  // namespace symgpu
