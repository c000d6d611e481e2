// gf2.hip — GF(2) row reduction without row swaps (reference: _rref_binary, symmer/operators/utils.py:292-315)
// and the symmetry-generator kernel built on it (IndependentOp.symmetry_generators,
// symmer/operators/independent_op.py:124-126).
//
// Reference loop: for i = 0..R-1: if row i != 0: pivot = leftmost set column of row i; XOR row i into every
// OTHER row that has that column set.  Sequential in i.  Blocked form used here (bit-exact by construction), up to
// 64 consecutive rows per block:
//   lead    — first non-zero word of each block row (one wave per row, coalesced, early exit).
//   panel   — ONE wavefront, lane = block row.  It holds a WINDOW of 4 words (256 columns) of every block row,
//             starting at the leading word of the block's first non-zero row, runs the reference loop on the window
//             (pivot row broadcast with v_readlane, pivot-column flags of all 64 rows with ONE __ballot, row updates
//             lane-predicated: ~50 instructions per pivot) and records, besides the pivots, the transformation
//             T (new_row_r = XOR_{i in T_r} old_row_i).  The window is exact as long as every processed row has its
//             leading word inside it and does not cancel to zero inside it; the first row that violates this ENDS the
//             block (it opens the next one, whose window starts at its own leading word), so any matrix is handled.
//   select  — for every row r outside the block: f(r) = its bits at the block's pivot columns BEFORE the block is
//             applied.  The reduced block is the identity on its pivot columns, so the unique combination of reduced
//             block rows that clears those bits is f(r) itself — exactly the row the sequential loop produces; in terms
//             of the OLD block rows the selector is g = f*T.  Block rows use g = T_r (minus themselves).
//   sweep   — row ^= XOR_{i in g(row)} old_block_row_i for ALL rows, one pass over the matrix per block (old block
//             rows snapshotted first; 64 of them live in VGPRs; wave-uniform selector -> scalar branches).
// Row-XORs are COUNTED as the reference performs them: with mask_j = set of block rows that held pivot j's column at
// time j, the sequential-time selector of an outside row is t_j = f_j ^ parity(f & mask_j & (2^j-1)), so the count is
// sum_j |mask_j| + sum_r |t(r)|  (derivation in DESIGN.md §3.5).
#include "common.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

namespace symgpu {

constexpr int WK = 64;        // max rows per block = lanes of the panel wave
constexpr int WN = 4;         // window width in 64-bit words
constexpr int NOLEAD = 0x7fffffff;
constexpr int SPEC_W = 8;     // words of the closed-column frontier (k_gf2_spec)

struct BlockInfo {
    i64 i0;                   // first row of the block
    int kk;                   // rows in the block (0: nothing left)
    int pivw[WK];             // absolute pivot word per block row, -1 = no pivot (zero row)
    int pivb[WK];
    u64 mask[WK];             // block rows (bit r < kk, r != j) holding pivot j's column at time j
    u64 T[WK];                // new_row_r = XOR_{i in T[r]} old_row_i
    int w_next;               // the largest pivot word of the block: where the NEXT block's window most likely starts (-1: no guess)
};

struct SweepState {
    i64 next_i0;              // first row not yet processed
};

// lead[r] = index of the first non-zero word of row next_i0 + r (NOLEAD if the row is zero, -1 if beyond the matrix)
__global__ __launch_bounds__(256) void k_lead(const u64 *__restrict__ rows, i64 R, i64 Wc, const SweepState *__restrict__ st, int *__restrict__ lead) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const i64 row = st->next_i0 + r;
    if (r >= WK) return;
    if (row >= R) { if (lane == 0) lead[r] = -1; return; }
    const u64 *p = rows + row * Wc;
    int found = NOLEAD;
    for (i64 w0 = 0; w0 < Wc; w0 += 64) {
        const i64 w = w0 + lane;
        const u64 nz = __ballot(w < Wc && p[w] != 0);
        if (nz) { found = (int)(w0 + __builtin_ctzll(nz)); break; }
    }
    if (lane == 0) lead[r] = found;
}

__global__ void k_sum_u32(const u32 *__restrict__ p, i64 n, unsigned long long *__restrict__ out) {
    unsigned long long s = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) s += p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

__device__ __forceinline__ u64 readlane64(u64 v, int l) {
    const u32 lo = __builtin_amdgcn_readlane((u32)v, l), hi = __builtin_amdgcn_readlane((u32)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

// Where the 4-word window starts: at the smallest leading word of the block's rows, but never so far left that the first non-zero
// row (lane jf) falls out of it.  (Round 2 started it AT the first row's leading word: when the pivots cross a word boundary some
// rows still lead in the word before — a dense matrix then lost a one-row block every 64 columns.)
__device__ __forceinline__ int window_start(int a, bool valid, int jf) {
    const int a_first = __builtin_amdgcn_readlane(a, jf);
    int lo = (valid && a != NOLEAD) ? a : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(lo, off); lo = o < lo ? o : lo; }
    const int floor_w = a_first - (WN - 1);
    return lo > floor_w ? lo : floor_w;
}

// the reference loop on a window of WNT words per block row (registers of ONE wavefront, lane = block row).
// Rows still to process, in order (genuinely zero rows have no pivot and are never modified: skipped).  ONE exit test per pivot:
// the row's leading word lies outside the window, or the row cancelled to zero inside it -> it opens the next block (re-windowed).
template <int WNT>
__device__ __forceinline__ void panel_loop(const u64 *__restrict__ rows, i64 Wc, i64 i0, int lane, bool valid, int w_lo, u64 in_m, u64 todo,
                                           int &kk, int &pw, int &pb, u64 &my_mask, u64 &tv, const u64 *spec, int w_spec, int &w_max) {
    // spec: the window words loaded speculatively at w_spec (the previous block's guess), beside the leads instead of behind them
    u64 C[WNT];
    if (w_spec == w_lo) {
#pragma unroll
        for (int k = 0; k < WNT; ++k) C[k] = spec[k];
    } else {
#pragma unroll
        for (int k = 0; k < WNT; ++k) C[k] = (valid && (i64)w_lo + k < Wc) ? rows[(i0 + lane) * Wc + w_lo + k] : 0ULL;
    }
    while (todo) {
        const int j = __builtin_ctzll(todo);
        u64 p[WNT];
#pragma unroll
        for (int k = 0; k < WNT; ++k) p[k] = readlane64(C[k], j);
        if (!((in_m >> j) & 1ULL)) { kk = j; break; }
        int k0 = 0, b;
        u64 mk;
        if (p[0] != 0) {                                            // common case: the pivot sits in the first window word
            b = __builtin_ctzll(p[0]);
            mk = __ballot((C[0] >> b) & 1ULL);
        } else {
            k0 = -1;
#pragma unroll
            for (int k = WNT - 1; k >= 1; --k) if (p[k] != 0) k0 = k;
            if (k0 < 0) { kk = j; break; }
            u64 pk = p[1], ck = C[1];
#pragma unroll
            for (int k = 2; k < WNT; ++k) if (k == k0) { pk = p[k]; ck = C[k]; }
            b = __builtin_ctzll(pk);
            mk = __ballot((ck >> b) & 1ULL);
        }
        mk &= ~(1ULL << j);
        todo &= todo - 1;
        if (lane == j) { pw = w_lo + k0; pb = b; my_mask = mk; }
        w_max = w_lo + k0 > w_max ? w_lo + k0 : w_max;
        const u64 tj = readlane64(tv, j);
        if ((mk >> lane) & 1ULL) {
#pragma unroll
            for (int k = 0; k < WNT; ++k) C[k] ^= p[k];
            tv ^= tj;
        }
    }
}

// The same loop on a TWO-word window, written for its DEPENDENT CHAIN (round 4: the panel is the critical path of every block — 64 pivots,
// one after the other, 560 cycles each in the generic loop, nearly all of it pipeline latency between the vector and the scalar unit:
// readlane -> scalar find -> vector test -> ballot -> scalar mask -> EXEC -> vector update -> readlane ...).  Here the chain of a pivot is
// six v_readlane (issued together) -> branch-free scalar arithmetic (s_ff1 on the two window words, one-hot masks) -> ONE vector block:
// the holders of the pivot column as an all-ones / zero word per lane (four and/or, compare, select) and the update as six v_bitop3
// x ^= p & m — no EXEC-masked update, no ballot on the way to the next pivot.  Row j itself is kept out with EXEC (it holds the column it
// pivots on); the holder mask for the records is the compare's VCC, filed into lane j afterwards (off the chain).
__device__ __forceinline__ void panel_loop_narrow(const u64 *__restrict__ rows, i64 Wc, i64 i0, int lane, bool valid, int w_lo, u64 in_m, u64 todo,
                                                  int &kk, int &pw, int &pb, u64 &my_mask, u64 &tv, const u64 *spec, int w_spec, int &w_max) {
    u64 C[2];
    if (w_spec == w_lo) { C[0] = spec[0]; C[1] = spec[1]; }
    else {
#pragma unroll
        for (int k = 0; k < 2; ++k) C[k] = (valid && (i64)w_lo + k < Wc) ? rows[(i0 + lane) * Wc + w_lo + k] : 0ULL;
    }
    u32 c0 = (u32)C[0], c1 = (u32)(C[0] >> 32), c2 = (u32)C[1], c3 = (u32)(C[1] >> 32), t0 = (u32)tv, t1 = (u32)(tv >> 32);
    u32 pw_v = (u32)pw, pb_v = (u32)pb, mlo_v = (u32)my_mask, mhi_v = (u32)(my_mask >> 32);
    const int w_lo_s = __builtin_amdgcn_readfirstlane(w_lo);
    while (todo) {
        const int j = __builtin_ctzll(todo);
        const u32 p0 = __builtin_amdgcn_readlane(c0, j), p1 = __builtin_amdgcn_readlane(c1, j), p2 = __builtin_amdgcn_readlane(c2, j),
                  p3 = __builtin_amdgcn_readlane(c3, j), q0 = __builtin_amdgcn_readlane(t0, j), q1 = __builtin_amdgcn_readlane(t1, j);
        const u64 P0 = ((u64)p1 << 32) | p0, P1 = ((u64)p3 << 32) | p2;
        // (rare exits, one test: the row leads outside the window, or it cancelled to zero inside it)
        if (!((in_m >> j) & 1ULL) || (P0 | P1) == 0ULL) { kk = j; break; }
        const int hiw = P0 == 0ULL ? 1 : 0;                                     // the pivot sits in the second window word
        const int b = __builtin_ctzll(hiw ? P1 : P0);
        const u64 oh = 1ULL << b, M0 = hiw ? 0ULL : oh, M1 = hiw ? oh : 0ULL;   // one-hot over the window
        const u64 onej = 1ULL << j;
        u64 mk;
        u32 t;
        asm volatile("s_andn2_b64 exec, -1, %[onej]\n\t"
                     "v_and_b32 %[t], %[m0], %[c0]\n\t"
                     "v_and_or_b32 %[t], %[c1], %[m1], %[t]\n\t"
                     "v_and_or_b32 %[t], %[c2], %[m2], %[t]\n\t"
                     "v_and_or_b32 %[t], %[c3], %[m3], %[t]\n\t"
                     "v_cmp_ne_u32 vcc, 0, %[t]\n\t"
                     "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
                     "v_bitop3_b32 %[c0], %[c0], %[p0], %[t] bitop3:0x78\n\t"
                     "v_bitop3_b32 %[c1], %[c1], %[p1], %[t] bitop3:0x78\n\t"
                     "v_bitop3_b32 %[c2], %[c2], %[p2], %[t] bitop3:0x78\n\t"
                     "v_bitop3_b32 %[c3], %[c3], %[p3], %[t] bitop3:0x78\n\t"
                     "v_bitop3_b32 %[t0], %[t0], %[q0], %[t] bitop3:0x78\n\t"
                     "v_bitop3_b32 %[t1], %[t1], %[q1], %[t] bitop3:0x78\n\t"
                     "s_mov_b64 %[mk], vcc\n\t"
                     "s_mov_b64 exec, -1"
                     : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [t0] "+v"(t0), [t1] "+v"(t1), [t] "=&v"(t), [mk] "=&s"(mk)
                     : [onej] "s"(onej), [m0] "s"((u32)M0), [m1] "s"((u32)(M0 >> 32)), [m2] "s"((u32)M1), [m3] "s"((u32)(M1 >> 32)),
                       [p0] "s"(p0), [p1] "s"(p1), [p2] "s"(p2), [p3] "s"(p3), [q0] "s"(q0), [q1] "s"(q1)
                     : "vcc");
        todo &= todo - 1;
        const int wabs = w_lo_s + hiw;
        w_max = wabs > w_max ? wabs : w_max;
        // the pivot's records into lane j (EXEC = {j}); nothing of the next pivot depends on them
        asm volatile("s_mov_b64 exec, %[onej]\n\t"
                     "v_mov_b32 %[pw], %[vw]\n\t"
                     "v_mov_b32 %[pb], %[vb]\n\t"
                     "v_mov_b32 %[ml], %[vl]\n\t"
                     "v_mov_b32 %[mh], %[vh]\n\t"
                     "s_mov_b64 exec, -1"
                     : [pw] "+v"(pw_v), [pb] "+v"(pb_v), [ml] "+v"(mlo_v), [mh] "+v"(mhi_v)
                     : [onej] "s"(onej), [vw] "s"(wabs), [vb] "s"(b), [vl] "s"((u32)mk), [vh] "s"((u32)(mk >> 32)));
    }
    pw = (int)pw_v; pb = (int)pb_v; my_mask = ((u64)mhi_v << 32) | mlo_v; tv = ((u64)t1 << 32) | t0;
}

// the panel proper: ONE wavefront (lane = block row), `a` = this lane's leading word (lead[] semantics), block starts at i0
__device__ __forceinline__ void panel_wave(const u64 *__restrict__ rows, i64 R, i64 Wc, i64 i0, int a, int lane, SweepState *__restrict__ st,
                                           BlockInfo *__restrict__ info, i64 *__restrict__ pivots, unsigned long long *__restrict__ xor_count,
                                           const u64 *spec = nullptr, int w_spec = -1, int lean = 1, int *kk_out = nullptr, int *pw_out = nullptr,
                                           int *pb_out = nullptr) {
    if (kk_out) { *kk_out = 0; *pw_out = -1; *pb_out = 0; }
    if (i0 >= R) { if (lane == 0) { info->i0 = i0; info->kk = 0; info->w_next = -1; } return; }
    int w_max = -1;
    u64 no_spec[WN] = {0, 0, 0, 0};
    if (!spec) { spec = no_spec; w_spec = -1; }
    const bool valid = a >= 0;
    const int n_valid = __popcll(__ballot(valid));                 // rows i0 .. i0+n_valid-1 exist
    const u64 zero_m = __ballot(valid && a == NOLEAD);
    const u64 fin_m = __ballot(valid && a != NOLEAD);
    int kk = n_valid;
    int pw = -1, pb = 0;                                            // this lane's (= block row's) pivot
    u64 my_mask = 0;                                                // mask_j for j = lane
    u64 tv = 1ULL << lane;                                          // T row of this lane
    if (fin_m != 0) {
        const int jf = __builtin_ctzll(fin_m);
        const int w_lo = window_start(a, valid, jf);
        // narrow window (round 3): when every row of the block leads inside the first TWO words — a dense matrix, whose 64 pivots
        // are 64 consecutive columns — the panel keeps two words per row instead of four: 4 v_readlane + 4 v_xor less per pivot.
        // A row that cancels to zero inside the two words ends the block (as it does with four), so the result is unchanged.
        const bool narrow = __ballot(valid && a != NOLEAD && !(a >= w_lo && a < w_lo + 2)) == 0ULL;
        const int wn = narrow ? 2 : WN;
        const u64 in_m = __ballot(valid && a != NOLEAD && a >= w_lo && a < w_lo + wn);
        const u64 todo0 = (n_valid >= 64 ? ~0ULL : ((1ULL << n_valid) - 1ULL)) & ~zero_m;
        static_assert(WN >= 2, "narrow window");
        if (narrow && lean) panel_loop_narrow(rows, Wc, i0, lane, valid, w_lo, in_m, todo0, kk, pw, pb, my_mask, tv, spec, w_spec, w_max);
        else if (narrow) panel_loop<2>(rows, Wc, i0, lane, valid, w_lo, in_m, todo0, kk, pw, pb, my_mask, tv, spec, w_spec, w_max);
        else panel_loop<WN>(rows, Wc, i0, lane, valid, w_lo, in_m, todo0, kk, pw, pb, my_mask, tv, spec, w_spec, w_max);
    }
    // publish: only rows < kk belong to the block
    const u64 low = (kk >= 64) ? ~0ULL : ((1ULL << kk) - 1ULL);
    const bool mine = lane < kk;
    if (kk_out) { *kk_out = kk; *pw_out = mine ? pw : -1; *pb_out = pb; }
    info->pivw[lane] = mine ? pw : -1;
    info->pivb[lane] = mine ? pb : 0;
    info->mask[lane] = mine ? (my_mask & low) : 0ULL;
    info->T[lane] = mine ? tv : 0ULL;
    if (mine && pivots) pivots[i0 + lane] = pw < 0 ? -1 : (i64)pw * 64 + pb;
    unsigned long long c = mine ? (unsigned long long)__popcll(my_mask & low) : 0ULL;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if (lane == 0) {
        info->i0 = i0;
        info->kk = kk;
        info->w_next = w_max;                                        // w_max is wave-uniform (scalar running maximum)
        st->next_i0 = i0 + kk;
        if (c) atomicAdd(xor_count, c);
    }
}

// ---- full-row panel (round 3): sparse rows lead at scattered words, and the 4-word window then ends a block after a row or two
// (700 x 700 at density 0.003: 450 blocks instead of 11, 8.7x the dense time).  When the window would end the block early and the
// rows are at most FULL_WC words long, the whole workgroup runs the reference loop on the 64 FULL rows in LDS (<= 128 KiB, the
// sweep's table area): wavefront 0 finds the pivot of row j and the block rows that hold its column and keeps T, everybody XORs
// row j into those rows.  Two barriers per pivot (~0.5 us) instead of ~0.25 us in registers, but the block never ends early.
constexpr int FULL_WC = 256;
__device__ __forceinline__ void panel_full(const u64 *__restrict__ rows, i64 R, i64 Wc, i64 i0, int a, u64 *__restrict__ m /* LDS [64][Wc] */,
                                           u64 *__restrict__ s_bc /* LDS [4] */, SweepState *__restrict__ st, BlockInfo *__restrict__ info,
                                           i64 *__restrict__ pivots, unsigned long long *__restrict__ xor_count) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nt = blockDim.x;
    const int W = (int)Wc;
    const int n_valid = (int)(R - i0 < WK ? R - i0 : WK);
    for (int x = threadIdx.x; x < n_valid * W; x += nt) m[x] = rows[i0 * Wc + x];
    __syncthreads();
    int pw = -1, pb = 0;                                            // wavefront 0, lane = block row
    u64 my_mask = 0, tv = 1ULL << lane;
    for (int j = 0; j < n_valid; ++j) {
        if (wave == 0) {
            int jw = -1, jb = 0;
            if (__builtin_amdgcn_readlane(a, j) != NOLEAD) {        // rows that were zero when phase 0 looked stay zero: nothing touches them
                for (int w0 = 0; w0 < W; w0 += 64) {
                    const int w = w0 + lane;
                    const u64 v = w < W ? m[j * W + w] : 0ULL;
                    const u64 nz = __ballot(v != 0);
                    if (nz) {
                        const int l = __builtin_ctzll(nz);
                        jw = w0 + l;
                        jb = __builtin_ctzll(readlane64(v, l));
                        break;
                    }
                }
            }
            u64 mk = 0;
            if (jw >= 0) mk = __ballot(lane < n_valid && lane != j && ((m[lane * W + jw] >> jb) & 1ULL));
            if (lane == j) { pw = jw; pb = jb; my_mask = mk; }
            const u64 tj = readlane64(tv, j);
            if ((mk >> lane) & 1ULL) tv ^= tj;
            if (lane == 0) s_bc[0] = mk;
        }
        __syncthreads();
        const u64 mk = s_bc[0];
        // a wavefront per flagged row (wave-uniform test: unflagged rows cost nothing), lanes over the words
        for (int r = wave; r < n_valid; r += nt / 64)
            if ((mk >> r) & 1ULL)
                for (int w = lane; w < W; w += 64) m[r * W + w] ^= m[j * W + w];
        __syncthreads();
    }
    if (wave == 0) {
        const bool mine = lane < n_valid;
        info->pivw[lane] = mine ? pw : -1;
        info->pivb[lane] = mine ? pb : 0;
        info->mask[lane] = mine ? my_mask : 0ULL;
        info->T[lane] = mine ? tv : 0ULL;
        if (mine && pivots) pivots[i0 + lane] = pw < 0 ? -1 : (i64)pw * 64 + pb;
        unsigned long long c = mine ? (unsigned long long)__popcll(my_mask) : 0ULL;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
        if (lane == 0) {
            info->i0 = i0;
            info->kk = n_valid;
            info->w_next = -1;
            st->next_i0 = i0 + n_valid;
            if (c) atomicAdd(xor_count, c);
        }
    }
}

__global__ __launch_bounds__(64) void k_wpanel(const u64 *__restrict__ rows, i64 R, i64 Wc, SweepState *__restrict__ st, const int *__restrict__ lead,
                                                BlockInfo *__restrict__ info, i64 *__restrict__ pivots, unsigned long long *__restrict__ xor_count, int lean) {
    const i64 i0 = st->next_i0;
    panel_wave(rows, R, Wc, i0, i0 < R ? lead[threadIdx.x] : -1, threadIdx.x, st, info, pivots, xor_count, nullptr, -1, lean);
}

// selectors of all rows (in terms of the OLD block rows), reference-order XOR count, snapshot of the old block rows.
// One wavefront per row, lane j <-> pivot j: the 64 pivot-column bits are fetched in parallel and f is ONE __ballot.
// one wavefront, one row: its selector in terms of the OLD block rows, and the row's share of the reference-order XOR count
__device__ __forceinline__ u64 select_row(const u64 *__restrict__ rows, i64 Wc, i64 r, int lane, i64 i0, int kk, int pw, int pb, u64 mj, u64 Tj,
                                          u32 *__restrict__ rowcnt) {
    if (r >= i0 && r < i0 + kk) {
        // new_r = XOR_{i in T_r} old_i = old_r ^ XOR_{i in T_r xor {r}} old_i
        return readlane64(Tj, (int)(r - i0)) ^ (1ULL << (r - i0));
    }
    const bool bit = (lane < kk && pw >= 0) ? ((rows[r * Wc + pw] >> pb) & 1ULL) : false;
    const u64 f = __ballot(bit);
    // sequential-time selector t_j = f_j ^ parity(f & mask_j & (2^j-1)): |t| row-XORs in the reference loop
    const bool tj = bit ^ (bool)(__popcll(f & mj) & 1);
    const u64 t = __ballot(tj);
    if (lane == 0 && rowcnt) atomicAdd(&rowcnt[r], (u32)__popcll(t));  // fire and forget (a load + store pair puts a round trip in front of the selector)
    u64 x = bit ? Tj : 0ULL;                                          // g = XOR_{j in f} T_j
    for (int off = 32; off > 0; off >>= 1) x ^= __shfl_xor(x, off);
    return x;
}

__global__ __launch_bounds__(256) void k_select(const u64 *__restrict__ rows, i64 R, i64 Wc, const BlockInfo *__restrict__ info,
                                                 u64 *__restrict__ sel, u64 *__restrict__ snap, u32 *__restrict__ rowcnt) {
    const int kk = info->kk;
    if (kk == 0) return;
    const int lane = threadIdx.x & 63;
    const i64 i0 = info->i0;
    const int pw = info->pivw[lane], pb = info->pivb[lane];
    const u64 mj = info->mask[lane] & ((1ULL << lane) - 1ULL);      // earlier block rows that held pivot `lane`'s column
    const u64 Tj = info->T[lane];
    const i64 r = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r < R) {
        const u64 g = select_row(rows, Wc, r, lane, i0, kk, pw, pb, mj, Tj, rowcnt);
        if (lane == 0) sel[r] = g;
    }
    // snapshot of the old block rows (the sweep overwrites them while other workgroups still read them)
    const i64 total = (i64)kk * Wc;
    for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < total; k += (i64)gridDim.x * 256) snap[k] = rows[i0 * Wc + k];
}


// Each lane owns one word column of SW_ROWS rows (kept in registers); the old block rows stream past once
// (independent loads, no dependent load->xor->store chain per row) and are XORed in under wave-uniform selector bits.
template <int SW_ROWS, int UNR>
__device__ __forceinline__ void sweep_tile(u64 *__restrict__ rows, i64 R, i64 Wc, int kk, const u64 *__restrict__ sel, const u64 *__restrict__ snap,
                                           i64 col_block, i64 row_group) {
    if (kk == 0) return;
    const i64 w = col_block * 256 + threadIdx.x;
    const bool live = w < Wc;
    const i64 wl = live ? w : Wc - 1;                               // dead lanes load a valid column and never store
    const i64 rb = row_group * SW_ROWS;
    const bool full = rb + SW_ROWS <= R;                             // uniform: branch-free loads and stores for full tiles
    u32 slo[SW_ROWS], shi[SW_ROWS];
    u64 x[SW_ROWS];
    u32 any = 0;
#pragma unroll
    for (int k = 0; k < SW_ROWS; ++k) {
        const i64 r = (full || rb + k < R) ? rb + k : R - 1;          // clamped: rows past the end are loaded but never stored
        const u64 sv = (rb + k < R) ? sel[r] : 0ULL;
        slo[k] = __builtin_amdgcn_readfirstlane((u32)sv);
        shi[k] = __builtin_amdgcn_readfirstlane((u32)(sv >> 32));
        any |= slo[k] | shi[k];
        x[k] = rows[r * Wc + wl];
    }
    if (any == 0) return;                                            // uniform
    const u64 *sp = snap + wl;
#pragma unroll UNR
    for (int j = 0; j < kk; ++j) {
        const u64 b = sp[(i64)j * Wc];
        const u32 blo = (u32)b, bhi = (u32)(b >> 32);
#pragma unroll
        for (int k = 0; k < SW_ROWS; ++k) {
            // all-ones / zero from selector bit j in ONE VALU op (v_bfe_i32): the CU's single scalar unit would otherwise
            // be the bottleneck (3 SALU per (j, row))
            const u32 sbits = (j < 32) ? slo[k] : shi[k];
            const u32 m = (u32)__builtin_amdgcn_sbfe((int)sbits, j & 31, 1);
            const u32 lo = __builtin_amdgcn_bitop3_b32((u32)x[k], blo, m, 0x78);
            const u32 hi = __builtin_amdgcn_bitop3_b32((u32)(x[k] >> 32), bhi, m, 0x78);
            x[k] = ((u64)hi << 32) | lo;
        }
    }
    if (!live) return;
    if (full) {
        // unconditional back-to-back stores (a conditional store per row made the compiler wait for the previous store:
        // s_waitcnt vmcnt(0) before each of the 16 stores serialised them)
        u64 *dst = rows + rb * Wc + w;
#pragma unroll
        for (int k = 0; k < SW_ROWS; ++k) dst[(i64)k * Wc] = x[k];
    } else {
#pragma unroll
        for (int k = 0; k < SW_ROWS; ++k)
            if (rb + k < R) rows[(rb + k) * Wc + w] = x[k];
    }
}

template <int SW_ROWS, int UNR>
__global__ __launch_bounds__(256) void k_sweep(u64 *__restrict__ rows, i64 R, i64 Wc, const BlockInfo *__restrict__ info,
                                                const u64 *__restrict__ sel, const u64 *__restrict__ snap) {
    sweep_tile<SW_ROWS, UNR>(rows, R, Wc, info->kk, sel, snap, blockIdx.x, blockIdx.y);
}

// ---- Method-of-Four-Russians sweep ------------------------------------------------------------------------------
// The flag-per-block-row sweep above costs 3 VALU instructions per (block row, matrix row, word): ~200 per word of a matrix
// row for a 64-row block, which makes the sweep VALU-bound.  Here a workgroup owns a 64-word column tile and FIRST tabulates,
// in LDS, all 16 XOR combinations of every group of 4 old block rows (16 groups x 16 entries x 64 words x 8 B = 128 KiB of the
// CU's 160 KiB).  A matrix row then needs 16 table look-ups (ds_read_b64, wave-uniform entry index taken from 4 selector bits,
// consecutive lanes -> consecutive words: conflict free) and 16 XORs per word instead of 64 conditional ones.
// PHASE as in k_sweep_lookahead: 0 = only the rows of the next block, 1 = workgroup 0 runs the panel of the next block and the
// others sweep all remaining rows.
constexpr int M4_TW = 64;                       // words per column tile = lanes
constexpr int M4_NT = 1024;                     // threads per workgroup (16 waves; one workgroup per CU because of the table)
constexpr int M4_U = 4;                         // rows in flight per wave
constexpr size_t M4_LDS = (size_t)16 * 16 * M4_TW * sizeof(u64);

// PHASE 3 (round 3) = phase 0 and the selector launch in ONE grid: blocks 0..3 compute the selectors of the next block's 64 rows and
// publish them (agent-scope stores, one tagged flag per row), the next n_tiles blocks are phase 0's tile workgroups — they build their
// tables straight from the old block rows (nobody writes those in this launch) and wait for the 64 flags before they touch the next
// block's rows — and the remaining blocks compute the selectors of all other rows (which phase 0 never touches) and the snapshot.
// One kernel boundary less on the critical path of every block: select 5 us -> hidden behind phase 0.
constexpr int SEL_PRI = 4;                      // priority blocks: 16 rows (wavefronts) each = the next block's 64 rows
struct FusedSelect {
    u64 *sel;                                   // writable view of the selectors
    u64 *snap;
    u32 *rowcnt;
    u64 *ready;                                 // [64][2] granules {epoch 32 | half of the row's selector 32}: the data is the flag
    u32 epoch;
    u32 *fail;                                  // a tile workgroup gave up waiting
    int full_panel;                             // 1: the panel may switch to the full rows in LDS (panel_full)
    int lean_panel;                             // 1: two-word windows run panel_loop_narrow (0: the generic loop; tests)
};
template <int PHASE>
__global__ __launch_bounds__(M4_NT) void k_sweep_m4r(u64 *__restrict__ rows, i64 R, i64 Wc, const BlockInfo *__restrict__ info,
                                                      const u64 *__restrict__ sel, const u64 *__restrict__ snap, int n_tiles, int n_chunks,
                                                      BlockInfo *__restrict__ info_next, SweepState *__restrict__ st, i64 *__restrict__ pivots,
                                                      unsigned long long *__restrict__ xor_count, int *__restrict__ lead, FusedSelect fs) {
    extern __shared__ u64 tab[];                                    // [16 groups][16 entries][64 words]
    __shared__ u64 s_sel[WK];
    __shared__ int s_ok;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = info->kk;
    const i64 i0n = info->i0 + kk;                                  // first row of the next block
    // rows of the next block: [nb, ne)   (PHASE 2: plain sweep of all rows, no lookahead)
    const i64 nb = PHASE == 2 ? 0 : (i0n < R ? i0n : R), ne = PHASE == 2 ? 0 : (i0n + WK < R ? i0n + WK : R);
    int k = blockIdx.x;
    if (PHASE == 3) {
        if (k < SEL_PRI || k >= SEL_PRI + n_tiles) {
            // ---- selector role: one wavefront per row ----
            if (kk == 0) return;
            const i64 i0 = info->i0;
            const int pw = info->pivw[lane], pb = info->pivb[lane];
            const u64 mj = info->mask[lane] & ((1ULL << lane) - 1ULL);
            const u64 Tj = info->T[lane];
            i64 r = -1;
            if (k < SEL_PRI) { r = nb + (i64)k * 16 + wave; if (r >= ne) r = -1; }
            else {
                const i64 v = (i64)(k - SEL_PRI - n_tiles) * 16 + wave, n_other = R - (ne - nb);
                if (v < n_other) r = v < nb ? v : v + (ne - nb);
            }
            if (r >= 0) {
                const u64 g = select_row(rows, Wc, r, lane, i0, kk, pw, pb, mj, Tj, fs.rowcnt);
                if (lane == 0) {
                    if (k < SEL_PRI) {
                        __hip_atomic_store(&fs.ready[2 * (r - nb)], ((u64)fs.epoch << 32) | (u32)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&fs.ready[2 * (r - nb) + 1], ((u64)fs.epoch << 32) | (u32)(g >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else fs.sel[r] = g;
                }
            }
            // snapshot of the old block rows for phase 1 (which overwrites them while other workgroups still read them)
            const i64 n_sel = (i64)gridDim.x - n_tiles, me = k < SEL_PRI ? k : k - n_tiles;
            const i64 total = (i64)kk * Wc;
            for (i64 x = me * M4_NT + threadIdx.x; x < total; x += n_sel * M4_NT) fs.snap[x] = rows[i0 * Wc + x];
            return;
        }
        k -= SEL_PRI;
    }
    if (PHASE == 1) {
        if (k == 0) {
            // ---- panel workgroup; the leading words were collected by phase 0 (reset for the next block).  One wavefront on a 4-word
            //      window — or, when the window would end the block early and the rows fit, the whole workgroup on the full rows ----
            int a = -1;
            u64 spec[WN] = {0, 0, 0, 0};
            const int w_spec = info->w_next;                         // the block that was just panelled guessed this block's window
            if (wave == 0 && i0n + lane < R) {
                a = lead[lane];
                if (w_spec >= 0) {                                   // the window words at the guess: loaded beside the leads, not behind them
#pragma unroll
                    for (int k = 0; k < WN; ++k) spec[k] = (i64)w_spec + k < Wc ? rows[(i0n + lane) * Wc + w_spec + k] : 0ULL;
                }
            }
            if (wave == 0) {
                bool full = false;
                if (fs.full_panel && Wc <= FULL_WC && i0n < R) {
                    const bool valid = a >= 0;
                    const u64 fin_m = __ballot(valid && a != NOLEAD);
                    if (fin_m != 0) {
                        const int w_lo = window_start(a, valid, __builtin_ctzll(fin_m));
                        const u64 bad = __ballot(valid && a != NOLEAD && !(a >= w_lo && a < w_lo + WN));   // rows that lead outside the window
                        const int n_valid = __popcll(__ballot(valid));
                        full = bad != 0 && __builtin_ctzll(bad) < (n_valid < 32 ? n_valid : 32);
                    }
                }
                if (lane == 0) { s_ok = full ? 1 : 0; if (full) atomicAdd(fs.fail + 1, 1u); }
                if (i0n + lane < R) lead[lane] = NOLEAD;
            }
            __syncthreads();
            if (s_ok) {
                if (wave != 0) a = 0;
                panel_full(rows, R, Wc, i0n, a, tab, s_sel, st, info_next, pivots, xor_count);
            } else if (wave == 0) panel_wave(rows, R, Wc, i0n, a, lane, st, info_next, pivots, xor_count, spec, w_spec, fs.lean_panel);
            return;
        }
        --k;
    }
    constexpr bool P0 = PHASE == 0 || PHASE == 3;
    if (kk == 0 && !P0) return;
    const int tile = k % n_tiles, chunk = k / n_tiles;
    // rows of this chunk: PHASE 0 the next block's rows, PHASE 1 / 2 the virtual index space of all OTHER rows
    const i64 n_rows = P0 ? ne - nb : R - (ne - nb);
    const i64 per = (n_rows + n_chunks - 1) / n_chunks;
    const i64 v_lo = (i64)chunk * per, v_hi = v_lo + per < n_rows ? v_lo + per : n_rows;
    if (v_lo >= v_hi) return;
    const i64 w = (i64)tile * M4_TW + lane;
    const bool live = w < Wc;
    const i64 wl = live ? w : Wc - 1;
    // ---- stream the rows: M4_U per wave and step, software pipelined (the loads of step i+1 are in flight while step i
    //      does its table look-ups: a wave only runs a handful of steps, so nothing else would hide the load latency) ----
    const i64 shift = ne - nb;
    const i64 step = M4_U * (M4_NT / 64);
    i64 rn[M4_U];
    u64 xn[M4_U], sn[M4_U];
    auto fetch = [&](i64 v0, bool with_sel) {
#pragma unroll
        for (int u = 0; u < M4_U; ++u) {
            const i64 v = v0 + u < v_hi ? v0 + u : (v0 < v_hi ? v0 : v_lo);   // tail: surplus slots repeat a valid row, never stored
            rn[u] = P0 ? nb + v : (v < nb ? v : v + shift);
            if (with_sel) sn[u] = kk != 0 ? (PHASE == 3 ? s_sel[rn[u] - nb] : sel[rn[u]]) : 0ULL;
            xn[u] = rows[rn[u] * Wc + wl];
        }
    };
    i64 v0 = v_lo + M4_U * wave;
    // phases 1 / 2: the first rows and their selectors are on their way while the tables are built (round 4: the loads used to start behind
    // the table build and its barrier, 2-3 us of every 15 us launch with nothing in flight)
    if (PHASE != 3 && v0 < v_hi) fetch(v0, true);
    // ---- tabulate the XOR combinations of the old block rows ----
    if (kk != 0) {
        for (int g = wave; g < 16; g += M4_NT / 64) {
            u64 sv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                sv[i] = (4 * g + i < kk) ? (PHASE == 3 ? rows[(info->i0 + 4 * g + i) * Wc + wl] : snap[(i64)(4 * g + i) * Wc + wl]) : 0ULL;
            u64 t[16];
            t[0] = 0; t[1] = sv[0]; t[2] = sv[1]; t[3] = sv[0] ^ sv[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[4 + e] = t[e] ^ sv[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[8 + e] = t[e] ^ sv[3];
#pragma unroll
            for (int e = 0; e < 16; ++e) tab[(g * 16 + e) * M4_TW + lane] = t[e];
        }
    }
    if (PHASE == 3) {
        // the rows themselves do not depend on the selectors: their loads are issued before the wait
        if (v0 < v_hi) fetch(v0, false);
        if (kk != 0) {
            // the selectors of the next block's rows come from blocks 0..3 of this launch: wait for their flags (bounded), then read them
            if (wave == 0) {
                const int nr = (int)(ne - nb);
                bool ok = true;
                u64 g0 = 0, g1 = 0;
                for (u32 spins = 0;; ++spins) {
                    const u64 tagged = (u64)fs.epoch << 32;
                    g0 = lane < nr ? __hip_atomic_load(&fs.ready[2 * lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tagged;
                    g1 = lane < nr ? __hip_atomic_load(&fs.ready[2 * lane + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tagged;
                    if (__ballot((u32)(g0 >> 32) == fs.epoch && (u32)(g1 >> 32) == fs.epoch) == ~0ULL) break;
                    if (spins >= (1u << 22)) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                s_sel[lane] = (ok && lane < nr) ? (((u64)(u32)g1 << 32) | (u32)g0) : 0ULL;
                if (lane == 0) { s_ok = ok ? 1 : 0; if (!ok) atomicOr(fs.fail, 1u); }
            }
        }
    }
    __syncthreads();
    if (PHASE == 3 && kk != 0 && !s_ok) return;                      // flagged: the call fails loudly
    if (PHASE == 3) {
#pragma unroll
        for (int u = 0; u < M4_U; ++u) sn[u] = (kk != 0 && v0 < v_hi) ? s_sel[rn[u] - nb] : 0ULL;
    }
    for (; v0 < v_hi; v0 += step) {
        i64 r[M4_U];
        u64 x[M4_U];
        u32 slo[M4_U], shi[M4_U];
#pragma unroll
        for (int u = 0; u < M4_U; ++u) {
            r[u] = rn[u]; x[u] = xn[u];
            slo[u] = __builtin_amdgcn_readfirstlane((u32)sn[u]);
            shi[u] = __builtin_amdgcn_readfirstlane((u32)(sn[u] >> 32));
        }
        if (v0 + step < v_hi) fetch(v0 + step, true);                // uniform
#pragma unroll
        for (int u = 0; u < M4_U; ++u) {
            const bool mine = (u == 0 || v0 + u < v_hi);
            if ((slo[u] | shi[u]) != 0u) {                           // uniform; untouched rows are not rewritten
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const u32 e = ((g < 8 ? slo[u] : shi[u]) >> (4 * (g & 7))) & 15u;
                    x[u] ^= tab[(g * 16 + (int)e) * M4_TW + lane];
                }
                if (live && mine) rows[r[u] * Wc + w] = x[u];
            }
            if (P0 && mine) {
                // leading word of the next block's rows for its panel: minimum over the column tiles
                const u64 nz = __ballot(live && x[u] != 0);
                if (nz && lane == 0) atomicMin(&lead[r[u] - nb], tile * M4_TW + (int)__builtin_ctzll(nz));
            }
        }
    }
}

// ---- ONE launch per block (round 4) --------------------------------------------------------------------------------------------------
// Launch A of the two-launch schedule (selectors of every row for the block just panelled, update of the next block's 64 rows, their
// leading words, snapshot of the block's old rows) only exists because the selectors of a row must be read before ANY column tile of that
// row is updated.  Here it becomes the TAIL of the launch that panels the block: workgroup 0 panels block k+1 while the tile workgroups sweep
// block k over all other rows (as phase 1); then every tile workgroup arrives at a counter, waits until all of them and the panel have
// (everything block k writes is in memory, the pivots of block k+1 are known, nobody writes a row any more), and the tail runs:
//   1. selectors of ALL rows for block k+1 — one wavefront per row, the rows shared out over the tile workgroups (sel[] is rewritten: its old
//      contents were consumed before the counter);
//   2. the n_tiles workgroups of chunk 0: selectors of the 64 rows after block k+1 once more into LDS (they are about to overwrite the pivot words
//      they are read from: a second, small counter separates every workgroup's reads from everybody's writes), tables of block k+1 from
//      its rows (written out as the snapshot the next launch builds its tables from), update of those 64 rows, their leading words.
// The next launch starts exactly where launch B of the two-launch schedule starts.  Two in-launch waits, both bounded; a time-out raises the
// same flag as launch A's and rref_dev restores the matrix and falls back to separate launches.  Slower than the two launches it replaces
// (see rref_dev_impl): kept behind SYMGPU_GF2_MERGED=1.
struct MergedSync { u32 *ctr; u32 epoch; u32 n_wg; };                  // ctr[0]: tile workgroups past their sweep, ctr[1]: panel done (epoch), ctr[2]: chunk-0 workgroups past their selector reads
typedef int32_t i32;
__device__ __forceinline__ bool merged_wait(const u32 *p, u32 target, bool at_least) {
    for (u32 spins = 0;; ++spins) {
        const u32 v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (at_least ? (i32)(v - target) >= 0 : v == target) return true;
        if (spins >= (1u << 22)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}
__global__ __launch_bounds__(M4_NT) void k_gf2_merged(u64 *__restrict__ rows, i64 R, i64 Wc, const BlockInfo *__restrict__ info, u64 *__restrict__ sel,
                                                       u64 *__restrict__ snap, int n_tiles, int n_chunks, BlockInfo *__restrict__ info_next,
                                                       SweepState *__restrict__ st, i64 *__restrict__ pivots, unsigned long long *__restrict__ xor_count,
                                                       int *__restrict__ lead, FusedSelect fs, MergedSync ms) {
    extern __shared__ u64 tab[];                                    // [16 groups][16 entries][64 words]
    __shared__ u64 s_sel[WK];
    __shared__ int s_ok;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = info->kk;
    const i64 i0n = info->i0 + kk;                                  // first row of the next block
    const i64 nb = i0n < R ? i0n : R, ne = i0n + WK < R ? i0n + WK : R;
    if (blockIdx.x == 0) {
        // ---- panel workgroup (as phase 1 of k_sweep_m4r) ----
        int a = -1;
        u64 spec[WN] = {0, 0, 0, 0};
        const int w_spec = info->w_next;
        if (wave == 0 && i0n + lane < R) {
            a = lead[lane];
            if (w_spec >= 0) {
#pragma unroll
                for (int k = 0; k < WN; ++k) spec[k] = (i64)w_spec + k < Wc ? rows[(i0n + lane) * Wc + w_spec + k] : 0ULL;
            }
        }
        if (wave == 0) {
            bool full = false;
            if (fs.full_panel && Wc <= FULL_WC && i0n < R) {
                const bool valid = a >= 0;
                const u64 fin_m = __ballot(valid && a != NOLEAD);
                if (fin_m != 0) {
                    const int w_lo = window_start(a, valid, __builtin_ctzll(fin_m));
                    const u64 bad = __ballot(valid && a != NOLEAD && !(a >= w_lo && a < w_lo + WN));
                    const int n_valid = __popcll(__ballot(valid));
                    full = bad != 0 && __builtin_ctzll(bad) < (n_valid < 32 ? n_valid : 32);
                }
            }
            if (lane == 0) { s_ok = full ? 1 : 0; if (full) atomicAdd(fs.fail + 1, 1u); }
            if (i0n + lane < R) lead[lane] = NOLEAD;
        }
        __syncthreads();
        if (s_ok) {
            if (wave != 0) a = 0;
            panel_full(rows, R, Wc, i0n, a, tab, s_sel, st, info_next, pivots, xor_count);
        } else if (wave == 0) panel_wave(rows, R, Wc, i0n, a, lane, st, info_next, pivots, xor_count, spec, w_spec, fs.lean_panel);
        // publish: the block's info is in memory before the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&ms.ctr[1], ms.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const int k = blockIdx.x - 1;
    const int tile = k % n_tiles, chunk = k / n_tiles;
    const i64 w = (i64)tile * M4_TW + lane;
    const bool live = w < Wc;
    const i64 wl = live ? w : Wc - 1;
    const i64 step = M4_U * (M4_NT / 64);
    // one pass of "rows ^= table look-ups under their selectors" over the virtual rows [v_lo, v_hi): PRI = the 64 rows after the block
    // (selectors from LDS, leading words recorded), otherwise all rows but those of [x_b, x_e) (selectors from sel[])
    auto sweep_rows = [&](i64 v_lo, i64 v_hi, bool pri, i64 x_b, i64 x_e, bool have_table) {
        i64 rn[M4_U];
        u64 xn[M4_U], sn[M4_U];
        const i64 shift = x_e - x_b;
        auto fetch = [&](i64 v0) {
#pragma unroll
            for (int u = 0; u < M4_U; ++u) {
                const i64 v = v0 + u < v_hi ? v0 + u : (v0 < v_hi ? v0 : v_lo);
                rn[u] = pri ? x_b + v : (v < x_b ? v : v + shift);
                sn[u] = have_table ? (pri ? s_sel[rn[u] - x_b] : sel[rn[u]]) : 0ULL;
                xn[u] = rows[rn[u] * Wc + wl];
            }
        };
        i64 v0 = v_lo + M4_U * wave;
        if (v0 < v_hi) fetch(v0);
        for (; v0 < v_hi; v0 += step) {
            i64 r[M4_U];
            u64 x[M4_U];
            u32 slo[M4_U], shi[M4_U];
#pragma unroll
            for (int u = 0; u < M4_U; ++u) {
                r[u] = rn[u]; x[u] = xn[u];
                slo[u] = __builtin_amdgcn_readfirstlane((u32)sn[u]);
                shi[u] = __builtin_amdgcn_readfirstlane((u32)(sn[u] >> 32));
            }
            if (v0 + step < v_hi) fetch(v0 + step);
#pragma unroll
            for (int u = 0; u < M4_U; ++u) {
                const bool mine = (u == 0 || v0 + u < v_hi);
                if ((slo[u] | shi[u]) != 0u) {
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const u32 e = ((g < 8 ? slo[u] : shi[u]) >> (4 * (g & 7))) & 15u;
                        x[u] ^= tab[(g * 16 + (int)e) * M4_TW + lane];
                    }
                    if (live && mine) rows[r[u] * Wc + w] = x[u];
                }
                if (pri && mine) {
                    const u64 nz = __ballot(live && x[u] != 0);
                    if (nz && lane == 0) atomicMin(&lead[r[u] - x_b], tile * M4_TW + (int)__builtin_ctzll(nz));
                }
            }
        }
    };
    // tables of the XOR combinations of a block's old rows (src: the snapshot, or the rows themselves + snapshot written on the way)
    auto build_table = [&](const u64 *src, i64 src_stride_rows_base, int kkb, u64 *snap_out) {
        for (int g = wave; g < 16; g += M4_NT / 64) {
            u64 sv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sv[i] = (4 * g + i < kkb) ? src[(src_stride_rows_base + 4 * g + i) * Wc + wl] : 0ULL;
                if (snap_out && live && 4 * g + i < kkb) snap_out[(i64)(4 * g + i) * Wc + w] = sv[i];
            }
            u64 t[16];
            t[0] = 0; t[1] = sv[0]; t[2] = sv[1]; t[3] = sv[0] ^ sv[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[4 + e] = t[e] ^ sv[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[8 + e] = t[e] ^ sv[3];
#pragma unroll
            for (int e = 0; e < 16; ++e) tab[(g * 16 + e) * M4_TW + lane] = t[e];
        }
    };
    // ---- the sweep of block k over all rows but the next block's (phase 1) ----
    if (kk != 0) {
        const i64 n_rows = R - (ne - nb);
        const i64 per = (n_rows + n_chunks - 1) / n_chunks;
        const i64 v_lo = (i64)chunk * per, v_hi = v_lo + per < n_rows ? v_lo + per : n_rows;
        if (v_lo < v_hi) {
            build_table(snap, 0, kk, nullptr);
            __syncthreads();
            sweep_rows(v_lo, v_hi, false, nb, ne, true);
        }
    }
    // ---- everything this workgroup writes for block k is in memory; wait for the others and for the panel ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicAdd(&ms.ctr[0], 1u);
        bool ok = merged_wait(&ms.ctr[0], ms.epoch * ms.n_wg, true) && merged_wait(&ms.ctr[1], ms.epoch, false);
        if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        else atomicOr(fs.fail, 1u);
        s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    if (!s_ok) return;
    // ---- tail 1: selectors of all rows for the block that was just panelled ----
    const int kk2 = info_next->kk;
    const i64 i02 = info_next->i0;
    const int pw2 = info_next->pivw[lane], pb2 = info_next->pivb[lane];
    const u64 mj2 = info_next->mask[lane] & ((1ULL << lane) - 1ULL);
    const u64 T2 = info_next->T[lane];
    const i64 nb2 = i02 + kk2 < R ? i02 + kk2 : R, ne2 = i02 + kk2 + WK < R ? i02 + kk2 + WK : R;
    if (kk2 != 0) {
        // (the 64 rows after the new block are left to tail 2: the chunk-0 workgroups overwrite their pivot words, and their selectors in
        // sel[] are never read — the next launch's sweep skips those rows)
        for (i64 r = (i64)k * (M4_NT / 64) + wave; r < R; r += (i64)ms.n_wg * (M4_NT / 64)) {
            if (r >= nb2 && r < ne2) continue;                       // wave-uniform
            const u64 g = select_row(rows, Wc, r, lane, i02, kk2, pw2, pb2, mj2, T2, fs.rowcnt);
            if (lane == 0) sel[r] = g;
        }
    }
    if (chunk != 0) return;
    // ---- tail 2 (one workgroup per column tile): the 64 rows after the new block ----
    if (kk2 != 0) {
        for (i64 r = nb2 + wave; r < ne2; r += M4_NT / 64) {
            const u64 g = select_row(rows, Wc, r, lane, i02, kk2, pw2, pb2, mj2, T2, tile == 0 ? fs.rowcnt : nullptr);   // counted once
            if (lane == 0) s_sel[r - nb2] = g;
        }
    }
    // every chunk-0 workgroup has READ the pivot words of those rows before any of them writes one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&ms.ctr[2], 1u);
        const bool ok = merged_wait(&ms.ctr[2], ms.epoch * (u32)n_tiles, true);
        if (!ok) atomicOr(fs.fail, 1u);
        s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    if (!s_ok || kk2 == 0) return;
    build_table(rows, i02, kk2, snap);                              // + the snapshot the next launch builds its tables from
    __syncthreads();
    sweep_rows(0, ne2 - nb2, true, nb2, ne2, true);
}

// ---- ONE launch per block and NO in-launch wait (round 4): the column-window schedule ------------------------------------------------
// What made launch A necessary: (1) the selector of a row must be read from the row's bits at the block's pivot columns before any column
// tile of the row is rewritten, (2) the next block's panel needs that block's 64 rows up to date (window + leading words).  Both are
// answered from a COLUMN WINDOW instead: every launch files, for ALL rows, the CWW words starting at the first word column that is not
// closed yet (closed = each of its 64 columns is the pivot column of a panelled block, or certified dead) as they are after the launch's
// update — twelve 8-byte stores per row from the tile workgroup that owns those words.  In the next launch
//   * every tile workgroup computes the selectors of its rows itself from that window (the pivots of a block panelled from the window lie
//     inside it by construction) — no selector launch, no sel[] array;
//   * the panel workgroup forms the next block's 64 candidate rows AS THEY WILL BE after this launch's sweep on those twelve words
//     (win_r = window_r ^ XOR_{j in g(r)} window_{old row j}): left of the window every unprocessed row is zero (closed columns), so a
//     row's leading word is its first non-zero window word, and the panel runs at once — beside the sweep, as before;
//   * the old block rows the tables are built from were filed by the previous launch as well (snap, the 64 candidate rows).
// Rows the window says nothing about: a candidate no old row goes into (g = 0) is unchanged by this launch, and its true leading word was
// filed by the previous one (leadkey: an atomic max per row and column tile over the 128 rows that can become candidates); a candidate
// that changes and is zero on the window is UNKNOWN and ends the block in front of it.  A block that starts with a row that leads outside
// the window (a "far" block: dependent rows lead in the identity columns) is panelled from the matrix itself — only unchanged rows can
// belong to it — and because its pivot columns are not in the column window, the next launch is a SELECTOR launch (mode F1: sel[] of all
// rows read from the matrix, nothing else) and the one after sweeps with sel[] (F2).  If even the first candidate is unknown, the next
// launch sweeps nothing (IDLE) and panels from the matrix with everything known.  The launches are generic steps: what a step does is read
// from a control block the previous step wrote (double buffered by launch parity), so the host just enqueues steps.
// Hole columns (left behind the pivots, never to be led in: dependent on the closed ones) would pin the window: the tile workgroup that owns
// the first unclosed word ORs, over all unprocessed rows, its bits that are not pivot columns yet; what stays zero is dead for good and
// the next panel closes it.
constexpr int CWW = 12;
enum { SPEC_N = 0, SPEC_F1 = 1, SPEC_F2 = 2, SPEC_IDLE = 3, SPEC_DONE = 4 };
struct SpecCtrl { u32 ep, mode, blk, pad; };                        // launch L reads ctrl[L & 1] and writes ctrl[(L + 1) & 1]
struct SpecState {                                                  // data epoch e reads st[e & 1], its panel writes st[(e + 1) & 1]
    i64 next_i0;                                                    // first row that is not a block row yet
    int closed_base;                                                // word columns [0, closed_base) are closed; cm[k]: closed columns of word closed_base + k
    int pad;
    u64 cm[SPEC_W];
};
struct SpecShared {
    SpecCtrl ctrl[2];
    u32 broken;                                                     // != 0: handed back to the two-launch schedule (too many far blocks)
    u32 n_far;
    i64 handoff_i0;                                                 // first row that is not reduced (nothing is pending)
    unsigned long long n_spec, n_idle;                              // statistics
    u64 alive[2], alive_hm[2];                                      // per data parity: candidate hole columns (hm) and those seen set (alive)
    int alive_w[2];                                                 //                  the word they belong to
    int colbase[2];                                                 //                  first word of the column window the epoch files
    u64 tstamp[16];                                                 // SYMGPU_GF2_DEBUG
};
struct SpecArgs {
    SpecState *st;                                                  // [2]
    SpecShared *sh;
    BlockInfo *binfo;                                               // [2]
    u64 *colwin;                                                    // [2][R][CWW]
    u64 *snap;                                                      // [2][WK][Wc]
    u64 *leadkey;                                                   // [R]: (epoch << 32) | (0xffffffff - leading word) where the epoch left the row non-zero
    u64 *sel;                                                       // [R]: far blocks
    u32 *rowcnt;
    u32 launch;                                                     // from 1
    int lean_panel;
};
// g = f * T without a cross-lane reduction: lane i holds column i of T (Tt), g_i = parity(f & Tt_i), g = ballot — a handful of VALU
// instructions per row instead of twelve ds_bpermute
__device__ __forceinline__ u64 transpose_T(u64 Tj, int lane) {
    u64 tt = 0;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
        const u64 col = __ballot((Tj >> i) & 1ULL);
        if (lane == i) tt = col;
    }
    return tt;
}
// selector of row r in terms of the old block rows from its pivot-column bit (lane j <-> pivot j); the row's share of the XOR count
__device__ __forceinline__ u64 select_from_bit(bool bit, int lane, i64 r, i64 i0, int kk, u64 mj, u64 Tj, u64 Tt, u32 *__restrict__ rowcnt) {
    if (r >= i0 && r < i0 + kk) return readlane64(Tj, (int)(r - i0)) ^ (1ULL << (r - i0));
    const u64 f = __ballot(bit);
    if (rowcnt) {
        const bool tj = bit ^ (bool)(__popcll(f & mj) & 1);
        const u64 t = __ballot(tj);
        if (lane == 0) atomicAdd(&rowcnt[r], (u32)__popcll(t));
    }
    return __ballot(__popcll(f & Tt) & 1);
}
__global__ __launch_bounds__(M4_NT) void k_gf2_spec(u64 *__restrict__ rows, i64 R, i64 Wc, int n_tiles, int n_chunks, i64 *__restrict__ pivots,
                                                     unsigned long long *__restrict__ xor_count, SpecArgs sa) {
    extern __shared__ u64 tab[];                                    // tile workgroups: [16 groups][16 entries][64 words]; panel: staging
    __shared__ u64 s_g[WK];
    __shared__ int s_lead[WK];
    __shared__ u64 s_cm[SPEC_W];
    __shared__ int s_flag;
    SpecShared *__restrict__ sh = sa.sh;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- everything a step needs to know, loaded together (one round trip, not a chain) ----
    const SpecCtrl c = sh->ctrl[sa.launch & 1u];
    SpecCtrl *__restrict__ c_out = &sh->ctrl[(sa.launch + 1u) & 1u];
    const u32 broken = sh->broken;
    const u32 ep = c.ep, mode = c.mode;
    const int pc = (int)(ep & 1u), pp = pc ^ 1;                     // parity of what this epoch files / of what the previous one filed
    const SpecState *__restrict__ s_in = sa.st + pc;
    SpecState *__restrict__ s_out = sa.st + pp;
    const BlockInfo *__restrict__ info = sa.binfo + ((c.blk + 1u) & 1u);
    BlockInfo *__restrict__ info_next = sa.binfo + (c.blk & 1u);
    const int kk_info = info->kk;
    const i64 i0 = info->i0;
    const i64 i0n = s_in->next_i0;                                  // candidates of the next block: [i0n, i0n + 64)
    const int cbp = sh->colbase[pp];                                // first word of the previous epoch's column window
    u64 cmv[SPEC_W];
#pragma unroll
    for (int q = 0; q < SPEC_W; ++q) cmv[q] = s_in->cm[q];
    const int closed_base = s_in->closed_base;
    int pw = info->pivw[lane], pb = info->pivb[lane];              // lane j <-> pivot j of the block to sweep
    u64 mj = info->mask[lane] & ((1ULL << lane) - 1ULL), Tj = info->T[lane];
    if (broken != 0u || mode == SPEC_DONE) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *c_out = c;
        return;
    }
    const int kk = mode == SPEC_IDLE ? 0 : kk_info;
    if (kk == 0) { pw = -1; pb = 0; mj = 0; Tj = 0; }
    // first word column that is not closed (this epoch's column window starts there) and the closed columns of that word
    int cw_now = closed_base;
    u64 cw_mask = ~0ULL;
    {
        bool open = false;
#pragma unroll
        for (int q = 0; q < SPEC_W; ++q) {
            if (!open && cmv[q] != ~0ULL) { open = true; cw_mask = cmv[q]; }
            if (!open) ++cw_now;
        }
    }
    if (mode == SPEC_F1) {
        // ---- selector step: sel[] of every row for a block whose pivot columns are outside the column window ----
        const i64 n_wave = (i64)gridDim.x * (M4_NT / 64);
        for (i64 r = (i64)blockIdx.x * (M4_NT / 64) + wave; r < R; r += n_wave) {
            const u64 g = select_row(rows, Wc, r, lane, i0, kk, pw, pb, mj, Tj, sa.rowcnt);
            if (lane == 0) sa.sel[r] = g;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) { SpecCtrl o = c; o.mode = SPEC_F2; *c_out = o; }
        return;
    }
    const bool from_sel = mode == SPEC_F2;
    const u64 *__restrict__ CWp = sa.colwin + (size_t)pp * (size_t)R * CWW;
    u64 *__restrict__ CWn = sa.colwin + (size_t)pc * (size_t)R * CWW;
    const u64 *__restrict__ snp = sa.snap + (size_t)pp * (size_t)WK * Wc;
    u64 *__restrict__ snn = sa.snap + (size_t)pc * (size_t)WK * Wc;
    const int pwr = (!from_sel && lane < kk && pw >= 0) ? pw - cbp : -1;     // the pivot's word inside the previous column window
    if (blockIdx.x == 0) {
        // ================= panel workgroup =================
        u64 *s_old = tab, *s_win = tab + WK * CWW, *s_raw = tab + 2 * WK * CWW;     // [64][CWW] each
        const bool stamp = sa.launch == 20u && threadIdx.x == 0;
        if (stamp) sh->tstamp[0] = wall_clock64();
        const bool direct = mode == SPEC_IDLE;                      // nothing is swept in this launch: the matrix itself is the truth
        const int base = direct ? cw_now : cbp;                     // first word of the window the candidates are looked at through
        const int nr = (int)(R - i0n < WK ? (R - i0n > 0 ? R - i0n : 0) : WK);
        // 1. the candidates' own window words, the old block rows' window words; selectors and filed leading words of the candidates —
        //    four candidates per wavefront, their loads issued together
        for (int x = threadIdx.x; x < WK * CWW; x += M4_NT) {
            const int r = x / CWW, q = x - r * CWW;
            u64 raw = 0, old = 0;
            const bool inside = (i64)base + q < Wc;                 // (words past the end of the rows were never filed)
            if (r < nr && inside) raw = direct ? rows[(i0n + r) * Wc + base + q] : CWp[(i0n + r) * CWW + q];
            if (r < kk && inside) old = CWp[(i0 + r) * CWW + q];
            s_raw[x] = raw; s_old[x] = old;
        }
        {
            u64 cbw[4], lk[4], sg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = wave * 4 + u;
                cbw[u] = (r < nr && pwr >= 0) ? CWp[(i0n + r) * CWW + pwr] : 0ULL;
                lk[u] = r < nr ? sa.leadkey[i0n + r] : 0ULL;
                sg[u] = (r < nr && from_sel && kk != 0) ? sa.sel[i0n + r] : 0ULL;
            }
            const u64 Tt = (kk != 0 && !from_sel) ? transpose_T(Tj, lane) : 0ULL;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = wave * 4 + u;
                u64 g = sg[u];
                if (r < nr && kk != 0 && !from_sel) g = select_from_bit(pwr >= 0 && ((cbw[u] >> pb) & 1ULL), lane, i0n + r, i0, kk, mj, Tj, Tt, nullptr);
                // the row's true leading word where it is known without the window: filed by the previous epoch and still true (g == 0)
                int tl = -2;                                        // -2: not known
                if (r < nr && direct) {
                    int fw = NOLEAD;
                    for (i64 x0 = 0; x0 < Wc; x0 += 64) {
                        const i64 x = x0 + lane;
                        const u64 nz = __ballot(x < Wc && rows[(i0n + r) * Wc + x] != 0ULL);
                        if (nz) { fw = (int)(x0 + __builtin_ctzll(nz)); break; }
                    }
                    tl = fw;
                } else if (r < nr && g == 0ULL) tl = (u32)(lk[u] >> 32) == ep - 1u ? (int)(0xffffffffu - (u32)lk[u]) : NOLEAD;
                if (lane == 0 && r < WK) { s_g[r] = g; s_lead[r] = tl; }
            }
        }
        __syncthreads();
        // 2. the candidates as the sweep of this launch leaves them, on the window words
        for (int x = threadIdx.x; x < WK * CWW; x += M4_NT) {
            const int r = x / CWW, q = x - r * CWW;
            u64 v = s_raw[x], g = s_g[r];
            while (g) { const int j = __builtin_ctzll(g); g &= g - 1; v ^= s_old[j * CWW + q]; }
            s_win[x] = v;
        }
        __syncthreads();
        if (wave != 0) return;
        if (stamp) sh->tstamp[1] = wall_clock64();
        // 3. leading words, decision, panel
        const bool valid = lane < nr;
        const int q0 = cw_now - base;                               // words in front of it are closed: zero in every unprocessed row
        const int UNKNOWN = 0x7ffffff0;
        int a = -1;
        if (valid) {
            const int tl = s_lead[lane];
            a = tl != -2 ? tl : UNKNOWN;
            if (tl == -2) for (int q = CWW - 1; q >= 0; --q) if (q >= q0 && s_win[lane * CWW + q] != 0ULL) a = base + q;
        }
        if (nr == 0) {                                              // no row left: this launch's sweep is the last thing that happens
            if (lane == 0) {
                SpecCtrl o = c; o.ep = ep + 1u; o.mode = SPEC_DONE; *c_out = o;
                info_next->i0 = i0n; info_next->kk = 0; info_next->w_next = -1;
            }
            return;
        }
        const u64 fin_m = __ballot(valid && a != NOLEAD);
        int w_lo = -1;                                              // the panel's window start (-1: a block of zero rows)
        bool stuck = false, far = false;
        if (fin_m != 0ULL) {
            const int jf = __builtin_ctzll(fin_m);
            if (__builtin_amdgcn_readlane(a, jf) == UNKNOWN) stuck = true;
            else {
                w_lo = window_start(a, valid, jf);
                far = w_lo < base || w_lo + WN > base + CWW;
            }
        }
        int kkn = 0, pwn = -1, pbn = 0;                             // the block panelled in this step (none when stuck)
        if (stuck) {
            // nothing is known about the first row: the next step sweeps nothing and looks at the matrix itself
            if (lane == 0) {
                SpecCtrl o = c; o.ep = ep + 1u; o.mode = SPEC_IDLE; *c_out = o;
                s_out->next_i0 = i0n;
                atomicAdd(&sh->n_idle, 1ULL);
            }
        } else if (far && sh->n_far >= 64u && sh->n_far * 2u > c.blk) {
            // mostly far blocks (sparse rows): the two-launch schedule with its full-row panels does these better; nothing is pending after
            // this launch's sweep
            if (lane == 0) {
                sh->handoff_i0 = i0n;
                SpecCtrl o = c; o.ep = ep + 1u; o.mode = SPEC_DONE; *c_out = o;
                __hip_atomic_store(&sh->broken, sa.launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        } else {
            const bool from_win = fin_m != 0ULL && !far;
            u64 spec[WN] = {0, 0, 0, 0};
            if (valid && from_win) {
#pragma unroll
                for (int q = 0; q < WN; ++q) spec[q] = s_win[lane * CWW + (w_lo - base) + q];
            }
            // a far block is read from the matrix: rows that change in this launch cannot belong to it (they lead near the frontier, far
            // from its window, so the panel's window test ends the block in front of them anyway; made explicit)
            if (far && !direct && valid && s_g[lane] != 0ULL && a != NOLEAD) a = UNKNOWN;
            if (stamp) sh->tstamp[2] = wall_clock64();
            panel_wave(rows, R, Wc, i0n, a, lane, reinterpret_cast<SweepState *>(s_out), info_next, pivots, xor_count, from_win ? spec : nullptr, from_win ? w_lo : -1,
                       sa.lean_panel, &kkn, &pwn, &pbn);
            if (stamp) { sh->tstamp[3] = wall_clock64(); sh->tstamp[8] = (u64)kkn; }
            // the next step: the block's pivots inside the column window this launch files -> N, else selectors from the matrix first
            const bool outside = __ballot(lane < kkn && pwn >= 0 && (pwn < cw_now || pwn >= cw_now + CWW)) != 0ULL;
            if (lane == 0) {
                SpecCtrl o = c;
                o.ep = ep + 1u; o.mode = outside ? SPEC_F1 : SPEC_N; o.blk = c.blk + 1u;
                *c_out = o;
                if (outside) atomicAdd(&sh->n_far, 1u); else atomicAdd(&sh->n_spec, 1ULL);
            }
        }
        // 4. the closed-column frontier after this step: carried over, holes certified dead by the previous epoch, words that filled up
        //    shifted out, this block's pivot columns filed
        if (lane == 0) {
            u64 cm2[SPEC_W];
#pragma unroll
            for (int q = 0; q < SPEC_W; ++q) cm2[q] = cmv[q];
            const u64 dead = sh->alive_hm[pp] & ~sh->alive[pp];
            const int aw = sh->alive_w[pp];
            if (dead != 0ULL && aw >= closed_base && aw < closed_base + SPEC_W) {
#pragma unroll
                for (int q = 0; q < SPEC_W; ++q) if (q == aw - closed_base) cm2[q] |= dead;
            }
            sh->alive[pp] = 0ULL; sh->alive_hm[pp] = 0ULL;
            int full = 0;
#pragma unroll
            for (int q = 0; q < SPEC_W; ++q) { if (cm2[q] != ~0ULL) break; ++full; }
#pragma unroll
            for (int q = 0; q < SPEC_W; ++q) {
                u64 v = 0;
#pragma unroll
                for (int q2 = 0; q2 < SPEC_W; ++q2) if (q2 == q + full) v = cm2[q2];
                s_cm[q] = v;
            }
            s_flag = closed_base + full;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nb2 = s_flag;
        if (lane < kkn && pwn >= nb2 && pwn < nb2 + SPEC_W) atomicOr((unsigned long long *)&s_cm[pwn - nb2], 1ULL << pbn);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < SPEC_W) s_out->cm[lane] = s_cm[lane];
        if (lane == 0) s_out->closed_base = nb2;
        if (stamp) sh->tstamp[4] = wall_clock64();
        return;
    }
    // ================= tile workgroups: sweep of the current block over their rows =================
    const int k = (int)blockIdx.x - 1;
    const bool tstamp = sa.launch == 20u && threadIdx.x == 0 && (k == 0 || k == (int)gridDim.x - 2);
    const int tso = k == 0 ? 5 : 10;
    if (tstamp) sh->tstamp[tso] = wall_clock64();
    const int tile = k % n_tiles, chunk = k / n_tiles;
    const i64 per = (R + n_chunks - 1) / n_chunks;
    const i64 v_lo = (i64)chunk * per, v_hi = v_lo + per < R ? v_lo + per : R;
    if (v_lo >= v_hi) return;
    const i64 w = (i64)tile * M4_TW + lane;
    const bool live = w < Wc;
    const i64 wl = live ? w : Wc - 1;
    // what this tile files besides the rows (all wave-uniform tests): the column window, the hole candidates of the first unclosed word
    const u64 hm = ~cw_mask;
    const bool tile_cw = (i64)tile * M4_TW < (i64)cw_now + CWW && (i64)tile * M4_TW + M4_TW > (i64)cw_now;
    const bool tile_hole = hm != 0ULL && (i64)tile * M4_TW <= (i64)cw_now && (i64)cw_now < (i64)tile * M4_TW + M4_TW;
    const bool hole_lane = live && w == (i64)cw_now;
    const i64 qn = w - (i64)cw_now;                                 // this lane's word inside the column window this epoch files
    const bool cw_lane = live && qn >= 0 && qn < CWW;
    if (k == 0 && threadIdx.x == 0) { sh->colbase[pc] = cw_now; sh->alive_w[pc] = cw_now; sh->alive_hm[pc] = hm; }
    u32 *const rowcnt = (tile == 0 && !from_sel) ? sa.rowcnt : nullptr;
    const i64 step = M4_U * (M4_NT / 64);
    i64 rn[M4_U];
    u64 xn[M4_U], cn[M4_U];
    auto fetch = [&](i64 v0) {
#pragma unroll
        for (int u = 0; u < M4_U; ++u) {
            rn[u] = v0 + u < v_hi ? v0 + u : (v0 < v_hi ? v0 : v_lo);   // tail: surplus slots repeat a valid row, never stored
            xn[u] = rows[rn[u] * Wc + wl];
            cn[u] = from_sel ? (kk != 0 ? sa.sel[rn[u]] : 0ULL) : (pwr >= 0 ? CWp[rn[u] * CWW + pwr] : 0ULL);
        }
    };
    i64 v0 = v_lo + M4_U * wave;
    if (v0 < v_hi) fetch(v0);
    const u64 Tt = (kk != 0 && !from_sel) ? transpose_T(Tj, lane) : 0ULL;
    if (kk != 0) {
        for (int g = wave; g < 16; g += M4_NT / 64) {
            u64 sv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) sv[i] = (4 * g + i < kk) ? snp[(i64)(4 * g + i) * Wc + wl] : 0ULL;
            u64 t[16];
            t[0] = 0; t[1] = sv[0]; t[2] = sv[1]; t[3] = sv[0] ^ sv[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[4 + e] = t[e] ^ sv[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[8 + e] = t[e] ^ sv[3];
#pragma unroll
            for (int e = 0; e < 16; ++e) tab[(g * 16 + e) * M4_TW + lane] = t[e];
        }
    }
    __syncthreads();
    if (tstamp) sh->tstamp[tso + 1] = wall_clock64();
    u64 alive_acc = 0;
    for (; v0 < v_hi; v0 += step) {
        i64 r[M4_U];
        u64 x[M4_U], cb[M4_U];
#pragma unroll
        for (int u = 0; u < M4_U; ++u) { r[u] = rn[u]; x[u] = xn[u]; cb[u] = cn[u]; }
        if (v0 + step < v_hi) fetch(v0 + step);                      // uniform
#pragma unroll
        for (int u = 0; u < M4_U; ++u) {
            const bool mine = (u == 0 || v0 + u < v_hi);
            u64 g = 0;
            if (kk != 0) g = from_sel ? cb[u] : select_from_bit(pwr >= 0 && ((cb[u] >> pb) & 1ULL), lane, r[u], i0, kk, mj, Tj, Tt, mine ? rowcnt : nullptr);
            const u32 slo = __builtin_amdgcn_readfirstlane((u32)g), shi = __builtin_amdgcn_readfirstlane((u32)(g >> 32));
            if ((slo | shi) != 0u) {                                 // uniform; untouched rows are not rewritten
#pragma unroll
                for (int gg = 0; gg < 16; ++gg) {
                    const u32 e = ((gg < 8 ? slo : shi) >> (4 * (gg & 7))) & 15u;
                    x[u] ^= tab[(gg * 16 + (int)e) * M4_TW + lane];
                }
                if (live && mine) rows[r[u] * Wc + w] = x[u];
            }
            if (mine) {
                const i64 ru = __builtin_amdgcn_readfirstlane((u32)r[u]) | ((i64)__builtin_amdgcn_readfirstlane((u32)(r[u] >> 32)) << 32);
                if (tile_cw && cw_lane) CWn[ru * CWW + qn] = x[u];
                if (ru >= i0n && ru < i0n + 2 * WK) {                // the rows that can be candidates of the next step: true leading word; old rows
                    const u64 nzb = __ballot(live && x[u] != 0ULL);
                    if (nzb != 0ULL && lane == 0)
                        atomicMax((unsigned long long *)&sa.leadkey[ru], ((u64)ep << 32) | (u64)(0xffffffffu - (u32)(tile * M4_TW + (int)__builtin_ctzll(nzb))));
                    if (ru < i0n + WK && live) snn[(ru - i0n) * Wc + w] = x[u];
                }
                if (tile_hole && hole_lane && ru >= i0n) alive_acc |= x[u] & hm;
            }
        }
    }
    if (tile_hole && alive_acc != 0ULL) atomicOr((unsigned long long *)&sh->alive[pc], alive_acc);
    if (tstamp) sh->tstamp[tso + 2] = wall_clock64();
}

// ---- small matrices: the whole reduction in ONE workgroup -------------------------------------------------------------
// R <= 64 rows and Wc <= 64 words (32 KiB of LDS): the reference loop verbatim — for every row in order: leftmost set column,
// flags of the rows holding it (one ballot), XOR — without the 3 launches per block of the blocked path.  Stabiliser sets,
// generator reconstructions and the symmetry matrices of molecules with <= 32 qubits land here (their rows are sparse and lead
// at scattered columns, which the windowed panel handles poorly).
constexpr int SMALL_R = 64, SMALL_WC = 64;     // wider dense matrices are faster on the blocked path (measured)

__global__ __launch_bounds__(256) void k_rref_small(u64 *__restrict__ rows, int R, int Wc, i64 *__restrict__ pivots,
                                                    unsigned long long *__restrict__ xor_count) {
    extern __shared__ u64 m[];                                      // [R][Wc]
    __shared__ u64 s_mask;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = R * Wc;
    for (int k = threadIdx.x; k < total; k += 256) m[k] = rows[k];
    __syncthreads();
    unsigned long long count = 0;                                   // kept by thread 0
    for (int j = 0; j < R; ++j) {
        if (wave == 0) {
            int pw = -1, pb = 0;
            for (int w0 = 0; w0 < Wc; w0 += 64) {
                const int w = w0 + lane;
                const u64 v = w < Wc ? m[j * Wc + w] : 0ULL;
                const u64 nz = __ballot(v != 0);
                if (nz) {
                    const int l = __builtin_ctzll(nz);
                    pw = w0 + l;
                    pb = __builtin_ctzll(__shfl(v, l));
                    break;
                }
            }
            u64 mask = 0;
            if (pw >= 0) mask = __ballot(lane < R && lane != j && ((m[lane * Wc + pw] >> pb) & 1ULL));
            if (lane == 0) {
                s_mask = mask;
                count += (unsigned long long)__popcll(mask);
                if (pivots) pivots[j] = pw < 0 ? -1 : (i64)pw * 64 + pb;
            }
        }
        __syncthreads();
        const u64 mask = s_mask;
        if (mask) {
            for (int k = threadIdx.x; k < total; k += 256) {
                const int r = k / Wc, w = k - r * Wc;
                if ((mask >> r) & 1ULL) m[k] ^= m[j * Wc + w];
            }
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < total; k += 256) rows[k] = m[k];
    if (threadIdx.x == 0 && xor_count) *xor_count = count;
}

// ---- in-place reduction of a device matrix -------------------------------------------------------
// fused_select: launch A carries the selectors of the next block's rows to its tile workgroups through in-launch flags (one launch less per
// block).  *timed_out: a tile workgroup gave up waiting (its workgroups were not co-resident) — the matrix is then partly updated: the caller
// restores it and runs the schedule with separate launches.
static int rref_dev_impl(u64 *rows, i64 R, i64 Wc, i64 *xor_count, i64 *pivots_host, bool fused_select_allowed, bool *timed_out) {
    hipStream_t st = ctx().stream;
    *timed_out = false;
    if (xor_count) *xor_count = 0;
    if (R <= 0 || Wc <= 0) return SYMGPU_OK;
    if (Wc >= ((i64)1 << 31) - 64) { set_error("rref: Wc too large"); return SYMGPU_E_INVALID; }
    {   // one-workgroup path for small matrices (SYMGPU_GF2_SMALL=0 disables it: tests)
        const char *env_small = getenv("SYMGPU_GF2_SMALL");
        static const bool small_attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rref_small), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                           SMALL_R * SMALL_WC * 8) == hipSuccess;
        if (R <= SMALL_R && Wc <= SMALL_WC && small_attr && !(env_small && env_small[0] == '0')) {
            Scratch piv, count;
            SG_TRY(piv.alloc((size_t)R * 8));
            SG_TRY(count.alloc(16));
            HIP_TRY(hipMemsetAsync(count.p, 0, 16, st));
            hipLaunchKernelGGL(k_rref_small, dim3(1), dim3(256), (size_t)R * Wc * 8, st, rows, (int)R, (int)Wc, piv.as<i64>(),
                               count.as<unsigned long long>());
            KERNEL_CHECK();
            unsigned long long h = 0;
            HIP_TRY(hipMemcpyAsync(&h, count.p, 8, hipMemcpyDeviceToHost, st));
            if (pivots_host) HIP_TRY(hipMemcpyAsync(pivots_host, piv.p, (size_t)R * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (xor_count) *xor_count = (i64)h;
            return SYMGPU_OK;
        }
    }
    Scratch info, state, lead, sel, snap, count, piv, rowcnt, ready;
    SG_TRY(info.alloc(2 * sizeof(BlockInfo)));
    SG_TRY(state.alloc(sizeof(SweepState)));
    SG_TRY(lead.alloc(WK * sizeof(int)));
    SG_TRY(sel.alloc((size_t)R * 8));
    SG_TRY(snap.alloc((size_t)WK * Wc * 8));
    SG_TRY(count.alloc(16));
    SG_TRY(piv.alloc((size_t)R * 8));
    SG_TRY(rowcnt.alloc((size_t)R * 4));
    HIP_TRY(hipMemsetAsync(rowcnt.p, 0, (size_t)R * 4, st));
    HIP_TRY(hipMemsetAsync(count.p, 0, 16, st));
    SG_TRY(ready.alloc(2 * WK * sizeof(u64)));
    HIP_TRY(hipMemsetAsync(ready.p, 0, 2 * WK * sizeof(u64), st));
    HIP_TRY(hipMemsetAsync(state.p, 0, sizeof(SweepState), st));
    HIP_TRY(hipMemsetAsync(info.p, 0, 2 * sizeof(BlockInfo), st));   // {i0 = 0, kk = 0}: "nothing swept yet, next block starts at row 0"
    constexpr int SW_ROWS = 16;                                     // rows per sweep workgroup, held in VGPRs
    const unsigned gx = (unsigned)((Wc + 255) / 256), gy = (unsigned)((R + SW_ROWS - 1) / SW_ROWS);
    const unsigned gsel = (unsigned)((R + 3) / 4);
    BlockInfo *binfo = info.as<BlockInfo>();
    // Four-Russians sweep (128 KiB of LDS per workgroup) unless disabled or refused by the runtime
    static const bool m4r_attr = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_m4r<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)M4_LDS) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_m4r<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)M4_LDS) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_m4r<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)M4_LDS) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_m4r<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)M4_LDS) == hipSuccess;
    }();
    const char *env_m4r = getenv("SYMGPU_GF2_M4R"), *env_la = getenv("SYMGPU_GF2_LOOKAHEAD");   // read per call: the tests switch paths
    const bool m4r = m4r_attr && !(env_m4r && env_m4r[0] == '0'), m4r_plain = m4r;
    const int m4_tiles = (int)((Wc + M4_TW - 1) / M4_TW);
    int m4_chunks = 256 / m4_tiles;                              // one workgroup per CU: about one round of workgroups
    if ((i64)m4_chunks > (R + 127) / 128) m4_chunks = (int)((R + 127) / 128);   // the table costs about 100 rows of work
    if (m4_chunks < 1) m4_chunks = 1;
    const bool lookahead = !(env_la && env_la[0] == '0');
    // SYMGPU_GF2_FUSED_SELECT=0: the selector launch on its own in front of phase 0 (three launches per block instead of two)
    const bool fused_select = fused_select_allowed;
    FusedSelect fs;
    fs.sel = sel.as<u64>(); fs.snap = snap.as<u64>(); fs.rowcnt = rowcnt.as<u32>(); fs.ready = ready.as<u64>(); fs.epoch = 0;
    fs.fail = reinterpret_cast<u32 *>(count.p) + 2;
    fs.full_panel = [] { const char *e = getenv("SYMGPU_GF2_FULL_PANEL"); return !(e && e[0] == '0'); }() ? 1 : 0;
    fs.lean_panel = [] { const char *e = getenv("SYMGPU_GF2_LEAN_PANEL"); return !(e && e[0] == '0'); }() ? 1 : 0;
    // one launch per block: needs every tile workgroup and the panel co-resident (<= one per CU) and the in-launch waits allowed
    static const bool merged_attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gf2_merged), hipFuncAttributeMaxDynamicSharedMemorySize, (int)M4_LDS) == hipSuccess;
    // MEASURED AND NOT THE DEFAULT (cfg4, round 4): 37.9 us per block against 9.7 + 14.3 us for the two launches — the grid-wide wait has to
    // write the sweep's 27 MB back from the XCDs' L2s (release) before the tail may read other workgroups' rows, which a kernel boundary does
    // in 2-4 us and an in-launch release + acquire + counter does in ~10, and the tail's two dependent steps (selectors, then 64 rows) run on
    // a chip that is otherwise idle.  SYMGPU_GF2_MERGED=1 selects it (tests keep it exercised).
    const bool merged_env = [] { const char *e = getenv("SYMGPU_GF2_MERGED"); return e && e[0] == '1'; }();
    const bool merged = fused_select && merged_env && merged_attr && lookahead && m4r && m4_tiles * m4_chunks + 1 <= ctx().num_cu;
    // the speculative-window schedule (k_gf2_spec): one launch per block, no in-launch wait; SYMGPU_GF2_SPEC=0 keeps to the two launches
    static const bool spec_attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gf2_spec), hipFuncAttributeMaxDynamicSharedMemorySize, (int)M4_LDS) == hipSuccess;
    // MEASURED AND NOT THE DEFAULT (cfg4, round 4): a step takes 24 us — the panel still waits for the selectors of its 64 candidates and
    // their updated window (two memory round trips and two LDS stages, 9.6 us in front of the panel's 12), exactly what launch A cost, and
    // far / unknown first rows add steps (87 launches for 73 blocks): 2.4 ms against 2.0.  SYMGPU_GF2_SPEC=1 selects it (tests keep it exercised).
    const bool spec_env = [] { const char *e = getenv("SYMGPU_GF2_SPEC"); return e && e[0] == '1'; }();
    bool spec_sched = spec_env && spec_attr && fused_select && !merged && lookahead && m4r && R * CWW < ((i64)1 << 40);
    Scratch spec_st, spec_sh, spec_cw, spec_snap, spec_lk;
    SpecArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.rowcnt = rowcnt.as<u32>(); sa.lean_panel = fs.lean_panel; sa.binfo = binfo; sa.sel = sel.as<u64>();
    if (spec_sched) {
        SG_TRY(spec_st.alloc(2 * sizeof(SpecState)));
        SG_TRY(spec_sh.alloc(sizeof(SpecShared)));
        SG_TRY(spec_cw.alloc((size_t)2 * R * CWW * sizeof(u64)));
        SG_TRY(spec_snap.alloc((size_t)2 * WK * Wc * sizeof(u64)));
        SG_TRY(spec_lk.alloc((size_t)R * sizeof(u64)));
        HIP_TRY(hipMemsetAsync(spec_lk.p, 0, (size_t)R * sizeof(u64), st));
        HIP_TRY(hipMemsetAsync(spec_st.p, 0, 2 * sizeof(SpecState), st));
        SpecShared h0;
        memset(&h0, 0, sizeof(h0));
        h0.ctrl[1].ep = 1; h0.ctrl[1].mode = SPEC_IDLE; h0.ctrl[1].blk = 0;     // launch 1: nothing to sweep, panel from the matrix
        HIP_TRY(hipMemcpyAsync(spec_sh.p, &h0, sizeof(h0), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));                           // (stack buffer)
        sa.st = spec_st.as<SpecState>(); sa.sh = spec_sh.as<SpecShared>(); sa.colwin = spec_cw.as<u64>(); sa.snap = spec_snap.as<u64>();
        sa.leadkey = spec_lk.as<u64>();
    }
    Scratch msync;
    if (merged) {
        SG_TRY(msync.alloc(64));
        HIP_TRY(hipMemsetAsync(msync.p, 0, 64, st));
    }
    if (lookahead && m4r && (i64)m4_tiles * m4_chunks + 1 < ((i64)1 << 31)) {
        // Pipeline, three launches per block: select(b) -> phase 0: sweep of the rows of block b+1 + their leading words ->
        // phase 1: panel of block b+1 (-> the other info buffer) inside the sweep of all remaining rows.  The very first
        // iteration has nothing to sweep (zeroed info): it only collects the leading words of rows 0..63 and panels block 0.
        // `it` keeps counting across batches.
        {
            int h_lead[WK];
            for (int k = 0; k < WK; ++k) h_lead[k] = NOLEAD;
            HIP_TRY(hipMemcpyAsync(lead.p, h_lead, sizeof(h_lead), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));                      // h_lead is a stack buffer
        }
        i64 it = 0, done = 0, prev = -1;
        bool finished = false;
        if (spec_sched) {
            // the column-window schedule: generic steps, the device decides what each one does (k_gf2_spec)
            const bool dbg = getenv("SYMGPU_GF2_DEBUG") != nullptr;
            u32 launch = 0;
            i64 spec_done = 0;
            for (int batch = 0;; ++batch) {
                i64 n_steps = (R - spec_done + WK - 1) / WK + 2;
                if (batch == 0 && n_steps > 12) n_steps = 12;         // a short first batch: a matrix the schedule hands back shows at once
                if (n_steps > 4096) n_steps = 4096;
                for (i64 k = 0; k < n_steps; ++k) {
                    sa.launch = ++launch;
                    ProfScope prof(2);
                    hipLaunchKernelGGL(k_gf2_spec, dim3((unsigned)(m4_tiles * m4_chunks + 1)), dim3(M4_NT), M4_LDS, st, rows, R, Wc, m4_tiles, m4_chunks, piv.as<i64>(),
                                       count.as<unsigned long long>(), sa);
                    KERNEL_CHECK();
                }
                SpecShared hsh;
                SpecState hst[2];
                HIP_TRY(hipMemcpyAsync(&hsh, spec_sh.p, sizeof(hsh), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(hst, spec_st.p, sizeof(hst), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                const SpecCtrl cc = hsh.ctrl[(launch + 1u) & 1u];
                const i64 reached = hst[cc.ep & 1u].next_i0;
                if (dbg) {
                    const u64 t0 = hsh.tstamp[0];
                    fprintf(stderr, "rref steps: %u launches, %llu blocks from the column window, %u far, %llu idle steps, row %lld, mode %u%s\n", launch, hsh.n_spec, hsh.n_far,
                            hsh.n_idle, (long long)reached, cc.mode, hsh.broken ? " (handed back to the two-launch schedule)" : "");
                    if (launch >= 20u && t0)
                        fprintf(stderr, "  launch 20 (10 ns ticks from the panel's start): staged %lld, decided %lld, panelled %lld (kk %llu), frontier %lld; tile wg 0: start %lld tables %lld done %lld; last: start %lld tables %lld done %lld\n",
                                (long long)(hsh.tstamp[1] - t0), (long long)(hsh.tstamp[2] - t0), (long long)(hsh.tstamp[3] - t0), (unsigned long long)hsh.tstamp[8],
                                (long long)(hsh.tstamp[4] - t0), (long long)(hsh.tstamp[5] - t0), (long long)(hsh.tstamp[6] - t0), (long long)(hsh.tstamp[7] - t0),
                                (long long)(hsh.tstamp[10] - t0), (long long)(hsh.tstamp[11] - t0), (long long)(hsh.tstamp[12] - t0));
                }
                if (hsh.broken != 0u) {
                    // everything in front of row handoff_i0 is reduced and swept, nothing is pending: the state the two-launch schedule starts
                    // from (an empty "current" block at that row)
                    BlockInfo hb;
                    memset(&hb, 0, sizeof(hb));
                    hb.i0 = hsh.handoff_i0; hb.kk = 0; hb.w_next = -1;
                    HIP_TRY(hipMemcpyAsync(binfo, &hb, sizeof(hb), hipMemcpyHostToDevice, st));
                    HIP_TRY(hipMemcpyAsync(binfo + 1, &hb, sizeof(hb), hipMemcpyHostToDevice, st));
                    SweepState hs0;
                    hs0.next_i0 = hsh.handoff_i0;
                    HIP_TRY(hipMemcpyAsync(state.p, &hs0, sizeof(hs0), hipMemcpyHostToDevice, st));
                    HIP_TRY(hipStreamSynchronize(st));               // (stack buffers)
                    done = hsh.handoff_i0;
                    prev = hsh.handoff_i0 - 1;
                    break;
                }
                if (cc.mode == SPEC_DONE) { finished = true; break; }
                if (batch > 0 && reached <= spec_done && n_steps >= 4) { set_error("rref: no progress (internal error, column-window schedule)"); return SYMGPU_E_INVALID; }
                spec_done = reached;
            }
        }
        while (!finished) {
            i64 n_iter = (R - done + WK - 1) / WK + 1;
            if (n_iter > 4096) n_iter = 4096;
            for (i64 k = 0; k < n_iter; ++k, ++it) {
                BlockInfo *cur = binfo + ((it + 1) & 1), *next = binfo + (it & 1);      // cur: block it-1 (to sweep), next: block it (to panel)
                fs.epoch = (u32)(it + 1);
                if (merged) {
                    // one launch per block (k_gf2_merged); the very first iteration still needs the leading words of rows 0..63: launch A once
                    if (it == 0)
                        hipLaunchKernelGGL(k_sweep_m4r<3>, dim3((unsigned)(SEL_PRI + m4_tiles + (R + 15) / 16)), dim3(M4_NT), M4_LDS, st, rows, R, Wc, cur, sel.as<u64>(),
                                           snap.as<u64>(), m4_tiles, 1, next, state.as<SweepState>(), piv.as<i64>(), count.as<unsigned long long>(), lead.as<int>(), fs);
                    MergedSync ms;
                    ms.ctr = msync.as<u32>(); ms.epoch = (u32)(it + 1); ms.n_wg = (u32)(m4_tiles * m4_chunks);
                    ProfScope prof(2);
                    hipLaunchKernelGGL(k_gf2_merged, dim3((unsigned)(m4_tiles * m4_chunks + 1)), dim3(M4_NT), M4_LDS, st, rows, R, Wc, cur, sel.as<u64>(), snap.as<u64>(),
                                       m4_tiles, m4_chunks, next, state.as<SweepState>(), piv.as<i64>(), count.as<unsigned long long>(), lead.as<int>(), fs, ms);
                    KERNEL_CHECK();
                    continue;
                }
                if (fused_select) {
                    // selectors of block it-1 and phase 0 in one grid (the very first iteration has no block to select for: kk == 0)
                    const unsigned g3 = (unsigned)(SEL_PRI + m4_tiles + (R + 15) / 16);
                    hipLaunchKernelGGL(k_sweep_m4r<3>, dim3(g3), dim3(M4_NT), M4_LDS, st, rows, R, Wc, cur, sel.as<u64>(), snap.as<u64>(),
                                       m4_tiles, 1, next, state.as<SweepState>(), piv.as<i64>(), count.as<unsigned long long>(), lead.as<int>(), fs);
                } else {
                    if (it > 0)
                        hipLaunchKernelGGL(k_select, dim3(gsel), dim3(256), 0, st, rows, R, Wc, cur, sel.as<u64>(), snap.as<u64>(), rowcnt.as<u32>());
                    hipLaunchKernelGGL(k_sweep_m4r<0>, dim3(m4_tiles), dim3(M4_NT), M4_LDS, st, rows, R, Wc, cur, sel.as<u64>(), snap.as<u64>(),
                                       m4_tiles, 1, next, state.as<SweepState>(), piv.as<i64>(), count.as<unsigned long long>(), lead.as<int>(), fs);
                }
                ProfScope prof(2);
                hipLaunchKernelGGL(k_sweep_m4r<1>, dim3(m4_tiles * m4_chunks + 1), dim3(M4_NT), M4_LDS, st, rows, R, Wc, cur, sel.as<u64>(),
                                   snap.as<u64>(), m4_tiles, m4_chunks, next, state.as<SweepState>(), piv.as<i64>(), count.as<unsigned long long>(),
                                   lead.as<int>(), fs);
                KERNEL_CHECK();
            }
            // the block that has been panelled but not swept yet: kk == 0 means the matrix is exhausted
            struct { i64 i0; int kk; } pending;
            HIP_TRY(hipMemcpyAsync(&pending, binfo + ((it + 1) & 1), sizeof(pending), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (pending.kk == 0) finished = true;
            else if (pending.i0 <= prev) { set_error("rref: no progress (internal error)"); return SYMGPU_E_INVALID; }
            prev = pending.i0;
            done = pending.i0;
        }
    } else {
    i64 done = 0;
    while (done < R) {
        // optimistic batch: every block consumes up to 64 rows; blocks that end early are caught by the read-back
        i64 n_iter = (R - done + WK - 1) / WK;
        if (n_iter > 4096) n_iter = 4096;
        for (i64 it = 0; it < n_iter; ++it) {
            hipLaunchKernelGGL(k_lead, dim3(WK / 4), dim3(256), 0, st, rows, R, Wc, state.as<SweepState>(), lead.as<int>());
            hipLaunchKernelGGL(k_wpanel, dim3(1), dim3(64), 0, st, rows, R, Wc, state.as<SweepState>(), lead.as<int>(), binfo,
                               piv.as<i64>(), count.as<unsigned long long>(), fs.lean_panel);
            hipLaunchKernelGGL(k_select, dim3(gsel), dim3(256), 0, st, rows, R, Wc, binfo, sel.as<u64>(), snap.as<u64>(),
                               rowcnt.as<u32>());
            ProfScope prof(2);
            if (m4r_plain)
                hipLaunchKernelGGL(k_sweep_m4r<2>, dim3(m4_tiles * m4_chunks), dim3(M4_NT), M4_LDS, st, rows, R, Wc, binfo, sel.as<u64>(), snap.as<u64>(),
                                   m4_tiles, m4_chunks, binfo, state.as<SweepState>(), piv.as<i64>(), count.as<unsigned long long>(), lead.as<int>(), fs);
            else
            hipLaunchKernelGGL((k_sweep<SW_ROWS, 4>), dim3(gx, gy), dim3(256), 0, st, rows, R, Wc, binfo, sel.as<u64>(), snap.as<u64>());
            KERNEL_CHECK();
        }
        SweepState hs;
        HIP_TRY(hipMemcpyAsync(&hs, state.p, sizeof(hs), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (hs.next_i0 <= done) { set_error("rref: no progress (internal error)"); return SYMGPU_E_INVALID; }
        done = hs.next_i0;
    }
    }
    hipLaunchKernelGGL(k_sum_u32, dim3(256), dim3(256), 0, st, rowcnt.as<u32>(), R, count.as<unsigned long long>());
    KERNEL_CHECK();
    unsigned long long hb[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(hb, count.p, 16, hipMemcpyDeviceToHost, st));
    if (pivots_host) HIP_TRY(hipMemcpyAsync(pivots_host, piv.p, (size_t)R * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (getenv("SYMGPU_GF2_DEBUG")) fprintf(stderr, "rref %lld x %lld words: full-row panels %u\n", (long long)R, (long long)Wc, (u32)(hb[1] >> 32));
    if ((u32)hb[1] != 0) { *timed_out = true; return SYMGPU_OK; }
    if (xor_count) *xor_count = (i64)hb[0];
    return SYMGPU_OK;
}

static bool g_gf2_fused_off = false;         // a launch-A wait timed out once: the process keeps to the separate-launch schedule

int rref_dev(u64 *rows, i64 R, i64 Wc, i64 *xor_count, i64 *pivots_host) {
    hipStream_t st = ctx().stream;
    if (xor_count) *xor_count = 0;
    if (R <= 0 || Wc <= 0) return SYMGPU_OK;
    const bool env_fused = [] { const char *e = getenv("SYMGPU_GF2_FUSED_SELECT"); return !(e && e[0] == '0'); }();
    const bool inject = getenv("SYMGPU_GF2_INJECT_TIMEOUT") != nullptr;      // tests: pretend the first attempt timed out
    const bool fused = env_fused && !g_gf2_fused_off;
    // The fused schedule waits inside a launch for flags of other workgroups (bounded, ~1 s).  Should that wait ever give up, the matrix
    // is half updated in place — so a copy of the input is kept (2 x 27 MB at 5 TB/s = 11 us of a 2 ms call at cfg4) and the reduction is
    // redone from it with separate launches; the fused form stays off for the rest of the process.
    Scratch orig;
    const bool big = R > SMALL_R || Wc > SMALL_WC;
    if (fused && big) {
        SG_TRY(orig.alloc((size_t)R * Wc * 8));
        HIP_TRY(hipMemcpyAsync(orig.p, rows, (size_t)R * Wc * 8, hipMemcpyDeviceToDevice, st));
    }
    bool timed_out = false;
    SG_TRY(rref_dev_impl(rows, R, Wc, xor_count, pivots_host, fused, &timed_out));
    if (inject && fused && big) timed_out = true;
    if (!timed_out) return SYMGPU_OK;
    if (!orig.p) { set_error("rref: an in-launch wait timed out on a schedule that has none (internal error)"); return SYMGPU_E_HIP; }
    g_gf2_fused_off = !inject;
    HIP_TRY(hipMemcpyAsync(rows, orig.p, (size_t)R * Wc * 8, hipMemcpyDeviceToDevice, st));
    if (xor_count) *xor_count = 0;
    SG_TRY(rref_dev_impl(rows, R, Wc, xor_count, pivots_host, false, &timed_out));
    if (timed_out) { set_error("rref: time-out on the separate-launch schedule (internal error)"); return SYMGPU_E_HIP; }
    return SYMGPU_OK;
}

// ---- symmetry-generator matrix build / read-out ----------------------------------------------------
// mat is (2n) x Wc, Wc = Wm + 2*Wq, Wm = ceil(M/64):  row c < n  = [ Z[:,c] | e_c ],  row n+c = [ X[:,c] | e_{n+c} ]
// (the transpose of independent_op.py:124's  vstack([hstack([Z, X]), eye(2n)])  with zero padding columns,
// which can never become pivots).  One wave transposes a 64-term x 64-qubit bit tile with 64 ballots.
__global__ __launch_bounds__(256) void k_build_symmat(const u64 *__restrict__ H, i64 M, int n, int Wq, u64 *__restrict__ mat, i64 Wc) {
    const int lane = threadIdx.x & 63;
    const i64 tile = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);   // 64-term tile index
    const int sw = blockIdx.y;                                   // source word 0..2Wq-1
    const i64 n_tiles = (M + 63) / 64;
    if (tile >= n_tiles) return;
    const i64 t = tile * 64 + lane;
    const u64 word = (t < M) ? H[t * 2 * Wq + sw] : 0ULL;
    u64 mine = 0;
    for (int b = 0; b < 64; ++b) {
        const u64 m = __ballot((word >> b) & 1ULL);
        if (lane == b) mine = m;
    }
    const int q = 64 * (sw % Wq) + lane;
    if (q < n) {
        const i64 c = (sw >= Wq) ? q : (i64)n + q;   // Z words feed rows 0..n-1, X words rows n..2n-1
        mat[c * Wc + tile] = mine;
    }
}

__global__ void k_set_identity(u64 *__restrict__ mat, int n, int Wq, i64 Wc, i64 Wm) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * n) return;
    const int q = c < n ? c : c - n;
    const i64 w = Wm + (c < n ? 0 : Wq) + q / 64;
    mat[(i64)c * Wc + w] |= 1ULL << (q % 64);
}

// flag[c] = 1 iff the first Wm words of row c are all zero (one wave per row)
__global__ __launch_bounds__(256) void k_rowzero_flags(const u64 *__restrict__ mat, i64 R, i64 Wc, i64 Wm, u32 *__restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const i64 r = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    bool nz = false;
    for (i64 w = lane; w < Wm; w += 64) nz |= (mat[r * Wc + w] != 0);
    const u64 any = __ballot(nz);
    if (lane == 0) flag[r] = any ? 0u : 1u;
}

__global__ void k_copy_generators(const u64 *__restrict__ mat, i64 R, i64 Wc, i64 Wm, int W, const u32 *__restrict__ pos, u32 total,
                                  u64 *__restrict__ out) {
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * W) return;
    const i64 r = idx / W;
    const int w = (int)(idx - r * W);
    const u32 p = pos[r];
    const u32 nxt = (r + 1 < R) ? pos[r + 1] : total;
    if (nxt == p + 1) out[(i64)p * W + w] = mat[r * Wc + Wm + w];
}

}  // namespace symgpu

using namespace symgpu;

extern "C" {

int symgpu_rref_dev(uint64_t *rows_dev, int64_t R, int64_t Wc, int64_t *xor_count, int64_t *pivots_host) {
    SG_TRY(require_ctx());
    SG_REQUIRE(R >= 0 && Wc >= 0 && (rows_dev || R * Wc == 0), "rref_dev");
    return rref_dev(rows_dev, R, Wc, xor_count, pivots_host);
}

int symgpu_rref(uint64_t *rows, int64_t R, int64_t Wc, int64_t *xor_count, int64_t *pivots) {
    SG_TRY(require_ctx());
    SG_REQUIRE(R >= 0 && Wc >= 0 && (rows || R * Wc == 0), "rref");
    if (xor_count) *xor_count = 0;
    if (R == 0 || Wc == 0) {
        if (pivots) for (i64 r = 0; r < R; ++r) pivots[r] = -1;
        return SYMGPU_OK;
    }
    Scratch d;
    SG_TRY(d.alloc((size_t)R * Wc * 8));
    HIP_TRY(hipMemcpyAsync(d.p, rows, (size_t)R * Wc * 8, hipMemcpyHostToDevice, ctx().stream));
    count_h2d((size_t)R * Wc * 8); count_d2h((size_t)R * Wc * 8);
    SG_TRY(rref_dev(d.as<u64>(), R, Wc, xor_count, pivots));
    HIP_TRY(hipMemcpyAsync(rows, d.p, (size_t)R * Wc * 8, hipMemcpyDeviceToHost, ctx().stream));
    HIP_TRY(hipStreamSynchronize(ctx().stream));
    return SYMGPU_OK;
}

int symgpu_symmetry_kernel_dev(symgpu_op_t H, int n_qubits, uint64_t *out, int64_t capacity, int64_t *k, int64_t *xor_count) {
    SG_TRY(require_ctx());
    SG_REQUIRE(H && k && n_qubits >= 1, "symmetry_kernel_dev");
    SG_REQUIRE((n_qubits + 63) / 64 == H->Wq, "symmetry_kernel_dev: n_qubits does not match Wq");
    hipStream_t st = ctx().stream;
    const int n = n_qubits, Wq = H->Wq, W = 2 * Wq;
    const i64 M = H->T, Wm = (M + 63) / 64, Wc = Wm + W, R = 2 * (i64)n;
    Scratch mat, flag, total, gens;
    SG_TRY(mat.alloc((size_t)R * Wc * 8));
    HIP_TRY(hipMemsetAsync(mat.p, 0, (size_t)R * Wc * 8, st));
    if (M > 0) {
        dim3 grid((unsigned)((Wm + 3) / 4), (unsigned)W);
        hipLaunchKernelGGL(k_build_symmat, grid, dim3(256), 0, st, H->rows, M, n, Wq, mat.as<u64>(), Wc);
        KERNEL_CHECK();
    }
    hipLaunchKernelGGL(k_set_identity, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, mat.as<u64>(), n, Wq, Wc, Wm);
    KERNEL_CHECK();
    SG_TRY(rref_dev(mat.as<u64>(), R, Wc, xor_count, nullptr));
    SG_TRY(flag.alloc((size_t)R * 4));
    SG_TRY(total.alloc(16));
    hipLaunchKernelGGL(k_rowzero_flags, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, mat.as<u64>(), R, Wc, Wm, flag.as<u32>());
    KERNEL_CHECK();
    SG_TRY(exclusive_scan_u32(flag.as<u32>(), flag.as<u32>(), R, total.as<u32>()));
    u32 kcount = 0;
    HIP_TRY(hipMemcpyAsync(&kcount, total.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *k = kcount;
    if ((i64)kcount > capacity) {
        set_error("symmetry_kernel: capacity %lld < %u generators", (long long)capacity, kcount);
        return SYMGPU_E_CAPACITY;
    }
    if (kcount == 0) return SYMGPU_OK;
    SG_REQUIRE(out, "symmetry_kernel: null output");
    SG_TRY(gens.alloc((size_t)kcount * W * 8));
    hipLaunchKernelGGL(k_copy_generators, dim3((unsigned)((R * W + 255) / 256)), dim3(256), 0, st, mat.as<u64>(), R, Wc, Wm, W,
                       flag.as<u32>(), kcount, gens.as<u64>());
    KERNEL_CHECK();
    HIP_TRY(hipMemcpyAsync(out, gens.p, (size_t)kcount * W * 8, hipMemcpyDeviceToHost, st));
    count_d2h((size_t)kcount * W * 8);
    HIP_TRY(hipStreamSynchronize(st));
    return SYMGPU_OK;
}

int symgpu_symmetry_kernel(const uint64_t *H, int64_t M, int n_qubits, int Wq, uint64_t *out, int64_t capacity, int64_t *k,
                           int64_t *xor_count) {
    SG_TRY(require_ctx());
    SG_REQUIRE(M >= 0 && n_qubits >= 1 && Wq == (n_qubits + 63) / 64 && k, "symmetry_kernel: sizes");
    SG_REQUIRE(H || M == 0, "symmetry_kernel: null input");
    symgpu_op_t op = nullptr;
    SG_TRY(symgpu_op_upload(H, nullptr, M, Wq, &op));
    int rc = symgpu_symmetry_kernel_dev(op, n_qubits, out, capacity, k, xor_count);
    symgpu_op_free(op);
    return rc;
}

}  // extern "C"
