// commute_m4r.hip — termwise commutation as a GF(2) matrix product by the Method of Four Russians, tables in LDS: operand preparation of
// the right operand, kernel choice and output handling.  The kernel itself is commute_m4r7.hip.
// (reference: symmer/operators/base.py:938-971 -> matmul_GF2 / numba_dot_matmal_GF2, utils.py:9-78: f64 dgemm, then % 2)
//
//   C = NOT( A . Omega . B^T  mod 2 ),  A: N x 2n bits, B: M x 2n bits       (True = commute)
//
// The register-tile kernel of commute.hip spends 4 VALU instructions per pair and 64-bit word (256 lane-ops per pair at n = 2000) and is
// VALU-issue bound.  Here the contraction axis is cut into groups of bits; for a group and a tile of 2048 columns a workgroup tabulates all
// XOR-combinations of the group's bit-rows of the (bit-transposed, X/Z-swapped) right operand in LDS, and a row of A then needs ONE
// ds_read_b128 per lane and group (a wave instruction serves 4 rows x 2048 columns) — 1/13 of the VALU work per pair.
// Rounds 2-4 used one 256-entry table (8 bits, 64 KiB) per step: 40.1 ms for the 200,000^2 adjacency matrix of cfg5, its look-up stream
// VALU bound at 5 instructions per read.  Round 5 (commute_m4r7.hip): two 128-entry tables per step folded with v_bitop3, 36.1 ms; round 6:
// persistent workgroups over the (tile, step) space, 31.2 ms.
//
//   BT[c][jw]   bit c of B rows 64jw..64jw+63 (bit-major copy, cached on the operator); row c of the contraction pairs A bit c with B bit
//               c +- 64Wq (x with z', z with x'), which is just a row offset into BT.
#include "common.h"
#include <stdlib.h>

namespace symgpu {

typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u64x2 lds_u64x2;            // raw LDS address -> ds_read_b128 without a base add

constexpr int MK_TILE_W = 32;                         // 64-bit words per column tile: 2048 columns

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

// bit-packed rows -> np.bool_ bytes for ANY row length and base alignment: the output [N][M] is treated as one flat run of N * M bytes and
// written in aligned 16-byte chunks (non-temporal), a chunk taking its 16 bits from one row or — across a row boundary — from two or
// more.  (Rounds 2-5 wrote rows whose length is not a multiple of 16 byte by byte: 13 ms for a 100,008^2 table against 2.6 ms for
// 100,000^2 — and 15 of 16 term counts are not multiples of 16.)  A block owns a contiguous span; row and column of its first byte
// come from one 64-bit division, the chunks inside from 32-bit ones.
constexpr int B2B_CHUNKS_PER_THREAD = 8;
__device__ __forceinline__ u32 b2b_bits16(const u64 *__restrict__ row, i64 j) {          // bits [j, j + 16) of a bit-packed row
    const i64 w = j >> 6;
    const int sh = (int)(j & 63);
    u64 v = row[w] >> sh;
    if (sh > 48) v |= row[w + 1] << (64 - sh);
    return (u32)v & 0xFFFFu;
}
__global__ __launch_bounds__(256) void k_bits_to_bytes_flat(const u64 *__restrict__ bits, i64 stride_words, i64 N, i64 M, uint8_t *__restrict__ out) {
    const i64 total = N * M;
    const i64 head = (i64)((16 - (reinterpret_cast<uintptr_t>(out) & 15)) & 15) < total ? (i64)((16 - (reinterpret_cast<uintptr_t>(out) & 15)) & 15) : total;
    const i64 n_chunks = (total - head) >> 4;
    const i64 span = (i64)256 * B2B_CHUNKS_PER_THREAD;                 // chunks per block
    const i64 c_block = (i64)blockIdx.x * span;
    auto bit_at = [&](i64 f) -> u32 { const i64 i = f / M, j = f - i * M; return (u32)(bits[i * stride_words + (j >> 6)] >> (j & 63)) & 1u; };
    if (blockIdx.x == 0) {                                             // the unaligned head and tail: a few bytes, one at a time
        for (i64 f = threadIdx.x; f < head; f += 256) out[f] = (uint8_t)bit_at(f);
        for (i64 f = head + (n_chunks << 4) + threadIdx.x; f < total; f += 256) out[f] = (uint8_t)bit_at(f);
    }
    if (c_block >= n_chunks) return;
    const i64 f_block = head + (c_block << 4);
    const i64 i_block = f_block / M, j_block = f_block - i_block * M;  // uniform
    u32x4 *dst = reinterpret_cast<u32x4 *>(out + head) + c_block;
#pragma unroll
    for (int k = 0; k < B2B_CHUNKS_PER_THREAD; ++k) {
        const i64 c = (i64)k * 256 + threadIdx.x;                      // chunk inside the span: consecutive lanes, consecutive chunks
        if (c_block + c >= n_chunks) break;
        const i64 d = j_block + (c << 4);                              // column of the chunk's first byte, counted from row i_block
        i64 i, j;
        if (d < ((i64)1 << 31) && M < ((i64)1 << 31)) { const u32 q = (u32)d / (u32)M; i = i_block + q; j = d - (i64)q * M; }
        else { const i64 q = d / M; i = i_block + q; j = d - q * M; }
        u32 b16;
        if (j + 16 <= M) {
            b16 = b2b_bits16(bits + i * stride_words, j);
        } else {                                                       // the chunk runs over the end of row i (rows shorter than 16: of several rows)
            b16 = 0;
            i64 ii = i, jj = j;
            for (int t = 0; t < 16; ++t) {
                if (ii < N) b16 |= ((u32)(bits[ii * stride_words + (jj >> 6)] >> (jj & 63)) & 1u) << t;
                if (++jj == M) { jj = 0; ++ii; }
            }
        }
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 x = (b16 >> (4 * q)) & 0xFu;
            v[q] = (x | (x << 7) | (x << 14) | (x << 21)) & 0x01010101u;
        }
        __builtin_nontemporal_store(v, dst + c);
    }
}

int bits_to_bytes_dev(const u64 *bits, i64 stride_words, i64 N, i64 M, uint8_t *out) {
    if (N <= 0 || M <= 0) return SYMGPU_OK;
    const i64 n_chunks = (N * M) / 16 + 1;
    const i64 span = (i64)256 * B2B_CHUNKS_PER_THREAD;
    const i64 g = (n_chunks + span - 1) / span;
    SG_REQUIRE(g < ((i64)1 << 31), "bits_to_bytes: table too large for one launch");
    hipLaunchKernelGGL(k_bits_to_bytes_flat, dim3((unsigned)g), dim3(256), 0, ctx().stream, bits, stride_words, N, M, out);
    KERNEL_CHECK();
    return SYMGPU_OK;
}

static i64 round_up_i64(i64 x, i64 m) { return (x + m - 1) / m * m; }

// tile heights: R rows per 16-lane slot -> 32 R rows per workgroup.  Taller tiles amortise the tables over more rows (round 5, 200,000^2
// terms at n = 2000: R = 16 / 24 / 48 -> 60.0 / 50.8 / 36.1 ms), but a workgroup finishes a tile with a store phase no other work on its CU
// hides, so a launch wants several tiles per CU for the stores of one workgroup to fall under the lookups of the others — the more, the
// shorter the tile's lookup phase is.  Measured (round 6, profiles/r06_m4r_pick.txt, n in 20..2000 x N in 20,000..100,000): the tallest
// height with  tiles x steps-per-tile >= 200 x CUs  is within 5 % of the best of the three everywhere; below 32 steps (n <= 192: rows of at
// most three words a half) the table is bound by its own bytes and R = 16 is never beaten.  SYMGPU_M4R_R forces one (tests).
static i64 m4r_workgroups(i64 N, i64 M, int R) {
    const i64 Mw = (M + 63) / 64;
    return ((N + 32 * R - 1) / (32 * R)) * ((Mw + MK_TILE_W - 1) / MK_TILE_W);
}
static int m4r_pick(i64 N, i64 M, int Wq) {
    if (const char *e = getenv("SYMGPU_M4R_R")) {
        const int r = atoi(e);
        if (r == 16 || r == 24 || r == 48) return r;
    }
    const i64 steps = ((128 * (i64)Wq + 6) / 7 + 1) / 2;
    if (steps < 32) return 16;
    const int cand[2] = {48, 24};
    for (int R : cand)
        if (m4r_workgroups(N, M, R) * steps >= (i64)200 * ctx().num_cu) return R;
    return 16;
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
    const int W = 2 * Wq;
    const int R = m4r_pick(N, M, Wq);
    const i64 Mw = (M + 63) / 64, Mw_pad = round_up_i64(Mw, MK_TILE_W);
    Scratch bt, bits;
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
    // np.bool_ output: expanded by the kernel's own epilogue, rows of any length at any base (unaligned 16-byte stores where they have to be).
    // SYMGPU_M4R_UNFUSED=1: bit-packed rows to scratch + the flat expansion kernel (a second pass: 2.8 against 2.3 ms at 100,000^2 terms of 20
    // qubits, but 0.30 against 0.36 ms at 30,000^2) — kept as the tested alternative.
    const bool fused_bytes = out && !getenv("SYMGPU_M4R_UNFUSED");
    void *dst = out_bits;
    i64 stride = Mw;
    if (fused_bytes) {
        dst = out;
        stride = M;
    } else if (out) {
        SG_TRY(bits.alloc((size_t)N * Mw * 8));
        dst = bits.p;
    }
    SG_TRY(commutes_m4r7_launch(A, N, M, Wq, bt_p, Mw_pad, R, fused_bytes, dst, stride));
    if (out && !fused_bytes) SG_TRY(bits_to_bytes_dev(bits.as<u64>(), stride, N, M, out));
    return SYMGPU_OK;
}

}  // namespace symgpu
