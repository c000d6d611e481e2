// product.hip — all-pairs Pauli product (reference: PauliwordOp._multiply_by_operator,
// symmer/operators/base.py:764-794; operand swap of __mul__ base.py:847-852 folded into `inner_is_left`).
//
//   row  o*Ni + i  =  inner[i] xor outer[o]
//   coeff          =  c_i * c_o * i^e ,  e = (3(Y_i+Y_o) + Y_out + 2|x_left & z_right|) mod 4
//
// Two kernels, each with its own bound:
//  * k_mul_coeff  — VALU: 8 instructions per pair per 64-bit word (xor, bitop3, bcnt, bitop3 per 32-bit
//    half).  Word-major operands: 8 outer terms per wave arrive in SGPRs via s_load_dwordx16, 4 inner terms
//    per lane via coalesced 512-byte loads.  Writes 16 B/pair, coalesced along the inner index.
//  * k_mul_rows   — HBM-write stream: 16*Wq B/pair.  Every lane owns 16-byte chunks of inner rows (held in
//    VGPRs across the outer loop) and streams  chunk ^ outer[o][chunk % Wq]  with non-temporal 16-byte
//    stores, 1 KiB per wave instruction, perfectly coalesced.  Inputs stay in L2; algorithmic bytes == HBM
//    bytes.  This is the kernel the north-star roofline is quoted on.
//  * k_mul_rows_e + k_mul_coeff_expand (round 2, default when the product carries coefficients and a row is a power-of-two
//    number of 16-byte chunks: n in 65..128, 193..256, 449..512, 961..1024, 1985..2048, 4033..4096, or n <= 64) — the row
//    stream has its VALU idle, and in the row-major layout the X and the Z words of a term sit in the two halves of an aligned
//    lane group: the phase sum  Y_out + 2|x_left & z_right|  of every output row is formed on the way (DPP lane exchange,
//    v_bcnt, three DPP adds: measured free, tools/ubench_fused.hip) and leaves as ONE byte per pair; a purely streaming second
//    kernel expands bytes to coefficients at HBM speed.  The word-major VALU-bound k_mul_coeff (0.158 ms per 2.56e7 pairs)
//    is replaced by 0.075 ms of streaming: 1.083 -> 1.0 ms per slab.
#include "common.h"
#include <stdlib.h>
#include <stdio.h>

namespace symgpu {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32 xor_and(u32 acc, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(acc, b, c, 0x78); }   // a ^ (b & c)
__device__ __forceinline__ u32 to_vgpr(u32 s) { u32 v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v; }
__device__ __forceinline__ u32 and_xor(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x60); }     // a & (b ^ c)
__device__ __forceinline__ u32 bcnt_acc(u32 x, u32 acc) { u32 r; asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc)); return r; }   // popc(x) + acc

constexpr int PO = 8;   // outer terms per wave (SGPR operand)
constexpr int PJ = 4;   // inner terms per lane: i = ibase + 64*b + lane
constexpr int PW = 4;   // waves per block, stacked along o

// KM = 0: coefficients.  KM = 1 (keys): instead of the coefficient the kernel emits the packed cleanup key of every pair
// (hash | phase exponent e | o | i): the 16-byte coefficient is never materialised, the cleanup rebuilds c_i * c_o * i^e from
// e and the two operand tables.  KM = 2 (keys of a squared operator, both operands the same array): the twins (i, o) / (o, i)
// of P * P are the same row with the same coefficient magnitude — equal if the two terms commute (e even), opposite if they
// anticommute (e odd) — and the twin with i > o comes first in pair-index order.  Only the pairs with i >= o get a key, half
// of them, written compacted in index order (slot = o*Ni - o(o-1)/2 + i - o); the cleanup weights them 1 (i == o), 2
// (commuting) or 0 (anticommuting: the pair still fixes the first-occurrence position of its row).
template <bool INNER_LEFT, int KM>
__global__ __launch_bounds__(256) void k_mul_coeff(const u64 *__restrict__ It, i64 Ipad, i64 Ni, const double *__restrict__ ci,
                                                    const u64 *__restrict__ Ot, i64 Opad, i64 No, const double *__restrict__ co,
                                                    int Wq, double *__restrict__ out /* [(o)*Ni + i][2], o relative to slab */,
                                                    PairKeyArgs ka) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool KEYS = KM != 0;
    const i64 o0 = ((i64)blockIdx.y * PW + wave) * PO;   // wave-uniform, relative to the slab
    const i64 ibase = (i64)blockIdx.x * (64 * PJ);
    if (ibase >= Ipad) return;                                         // surplus block of the grid padded to a multiple of 8 (below)
    if (KM == 2 && ibase + 64 * PJ - 1 < o0 + ka.o_base) return;       // tile strictly below the diagonal: no pair with i >= o

    u32 cnt[PO][PJ], flip[PO][PJ];
    u32 yi[PJ];
    int yo[PO];
#pragma unroll
    for (int a = 0; a < PO; ++a) {
        yo[a] = 0;
#pragma unroll
        for (int b = 0; b < PJ; ++b) cnt[a][b] = flip[a][b] = 0;
    }
#pragma unroll
    for (int b = 0; b < PJ; ++b) yi[b] = 0;

    const u64 *pox = Ot + o0, *poz = Ot + (i64)Wq * Opad + o0;
    const u64 *pix = It + ibase + lane, *piz = It + (i64)Wq * Ipad + ibase + lane;

    // The words of step w + 1 are fetched before the arithmetic of step w; one parity accumulator for both halves of a word frees the
    // registers the second set of operand words needs.
    // Two register sets used in turn (no copies between them): the loads of step w + 1 are issued at the top of step w and first used a
    // whole step later — the scalar loads of the outer words too, which the single-set form waited for right where it issued them.
    struct Words { u64 xi[PJ], zi[PJ], xo[PO], zo[PO]; };
    Words A, B;
    auto fetch = [&](Words &d, int w) {
#pragma unroll
        for (int b = 0; b < PJ; ++b) {
            d.xi[b] = pix[(i64)w * Ipad + 64 * b];
            d.zi[b] = piz[(i64)w * Ipad + 64 * b];
        }
#pragma unroll
        for (int a = 0; a < PO; ++a) {
            d.xo[a] = pox[(i64)w * Opad + a];      // wave-uniform -> s_load
            d.zo[a] = poz[(i64)w * Opad + a];
        }
    };
    auto step = [&](const Words &c) {
#pragma unroll
        for (int b = 0; b < PJ; ++b) yi[b] += __popcll(c.xi[b] & c.zi[b]);
#pragma unroll
        for (int a = 0; a < PO; ++a) yo[a] += __popcll(c.xo[a] & c.zo[a]);
#pragma unroll
        for (int a = 0; a < PO; ++a) {
            // SGPR sources cost ~40 % VALU issue rate on gfx950 (tools/ubench_bitop.hip): copy the uniform words to VGPRs once
            // (the words that only feed a two-operand v_xor stay scalar: an SGPR source is free there)
            const u32 xol = INNER_LEFT ? (u32)c.xo[a] : to_vgpr((u32)c.xo[a]), xoh = INNER_LEFT ? (u32)(c.xo[a] >> 32) : to_vgpr((u32)(c.xo[a] >> 32));
            const u32 zol = to_vgpr((u32)c.zo[a]), zoh = to_vgpr((u32)(c.zo[a] >> 32));
#pragma unroll
            for (int b = 0; b < PJ; ++b) {
                const u32 xil = (u32)c.xi[b], xih = (u32)(c.xi[b] >> 32), zil = (u32)c.zi[b], zih = (u32)(c.zi[b] >> 32);
                // Y_out += |(xi^xo) & (zi^zo)|   (v_bcnt with its accumulator operand: the compiler adds two counts with a third instruction)
                cnt[a][b] = bcnt_acc(and_xor(xil ^ xol, zil, zol), cnt[a][b]);
                cnt[a][b] = bcnt_acc(and_xor(xih ^ xoh, zih, zoh), cnt[a][b]);
                // flip ^= x_left & z_right
                if (INNER_LEFT) {
                    flip[a][b] = xor_and(flip[a][b], xil, zol);
                    flip[a][b] = xor_and(flip[a][b], xih, zoh);
                } else {
                    flip[a][b] = xor_and(flip[a][b], zil, xol);
                    flip[a][b] = xor_and(flip[a][b], zih, xoh);
                }
            }
        }
    };
    fetch(A, 0);
    int w = 0;
    for (; w + 1 < Wq; w += 2) {
        fetch(B, w + 1);
        step(A);
        fetch(A, w + 2 < Wq ? w + 2 : w + 1);
        step(B);
    }
    if (w < Wq) step(A);

    // Epilogue.  The stores of a full 8-outer-term tile are issued unconditionally back to back: a conditional store per
    // pair made the compiler drain the memory counter (s_waitcnt vmcnt(0)) before every single store.
    const bool full_o = o0 + PO <= No;                                   // wave-uniform
    u64 ho[PO];
    double cor[PO], coi[PO];
#pragma unroll
    for (int a = 0; a < PO; ++a) {
        const i64 o = (o0 + a < No) ? o0 + a : (No > 0 ? No - 1 : 0);    // clamped: rows past the end are computed, never stored
        if (KEYS) ho[a] = ka.hO[o];
        else { cor[a] = co[2 * o]; coi[a] = co[2 * o + 1]; }
    }
    if (KM == 2) {
        const int F = ka.bi + ka.bo + 2;
        const u64 hmask = ~((1ULL << F) - 1ULL);
#pragma unroll
        for (int a = 0; a < PO; ++a) {
            const i64 o = o0 + a + ka.o_base;                               // absolute outer index
            if (o0 + a >= No) break;                                        // wave-uniform
            u64 *dst = ka.keys + (o * Ni - o * (o - 1) / 2 - o);            // + i
#pragma unroll
            for (int b = 0; b < PJ; ++b) {
                const i64 i = ibase + 64 * b + lane;
                const u64 e = (3u * (yi[b] + (u32)yo[a]) + cnt[a][b] + 2u * (__popc(flip[a][b]) & 1u)) & 3u;
                if (i < Ni && i >= o) {
                    if (ka.ebytes) ka.ebytes[(o * Ni - o * (o - 1) / 2 - o) + i] = (unsigned char)(e | (i == o ? 4u : 0u));      // (uniform choice)
                    else dst[i] = ((ka.hI[i] ^ ho[a]) & hmask) | (e << (ka.bi + ka.bo)) | ((u64)o << ka.bi) | (u64)i;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < PJ; ++b) {
        const i64 i = ibase + 64 * b + lane;
        if (i >= Ni) continue;
        if (KEYS) {
            const u64 hi = ka.hI[i];
            const int F = ka.bi + ka.bo + 2;
            const u64 hmask = ~((1ULL << F) - 1ULL);
            u64 key[PO];
#pragma unroll
            for (int a = 0; a < PO; ++a) {
                const u64 e = (3u * (yi[b] + (u32)yo[a]) + cnt[a][b] + 2u * (__popc(flip[a][b]) & 1u)) & 3u;
                key[a] = ((hi ^ ho[a]) & hmask) | (e << (ka.bi + ka.bo)) | ((u64)(o0 + a + ka.o_base) << ka.bi) | (u64)i;
            }
            if (ka.ebytes) {                                                // (uniform) one byte per pair: the phase exponent
                unsigned char *db = ka.ebytes + o0 * Ni + i;
                if (full_o) {
#pragma unroll
                    for (int a = 0; a < PO; ++a) db[(i64)a * Ni] = (unsigned char)((key[a] >> (ka.bi + ka.bo)) & 3u);
                } else {
#pragma unroll
                    for (int a = 0; a < PO; ++a)
                        if (o0 + a < No) db[(i64)a * Ni] = (unsigned char)((key[a] >> (ka.bi + ka.bo)) & 3u);
                }
                continue;
            }
            u64 *dst = ka.keys + o0 * Ni + i;
            if (full_o) {
#pragma unroll
                for (int a = 0; a < PO; ++a) dst[(i64)a * Ni] = key[a];
            } else {
#pragma unroll
                for (int a = 0; a < PO; ++a)
                    if (o0 + a < No) dst[(i64)a * Ni] = key[a];
            }
            continue;
        }
        const double ar = ci[2 * i], ai = ci[2 * i + 1];
        double2 v[PO];
#pragma unroll
        for (int a = 0; a < PO; ++a) {
            const int e = (int)((3u * (yi[b] + (u32)yo[a]) + cnt[a][b] + 2u * (__popc(flip[a][b]) & 1u)) & 3u);
            pair_coefficient(ar, ai, cor[a], coi[a], e, v[a].x, v[a].y);
        }
        double2 *dst = reinterpret_cast<double2 *>(out) + o0 * Ni + i;
        if (full_o) {
#pragma unroll
            for (int a = 0; a < PO; ++a) {
                typedef double f64x2 __attribute__((ext_vector_type(2)));
                const f64x2 w = {v[a].x, v[a].y};
                __builtin_nontemporal_store(w, reinterpret_cast<f64x2 *>(dst + (i64)a * Ni));   // streamed out: keep the operands in L2
            }
        } else {
#pragma unroll
            for (int a = 0; a < PO; ++a)
                if (o0 + a < No) dst[(i64)a * Ni] = v[a];
        }
    }
}

// ---- the HBM-write stream ------------------------------------------------------------------------
// RCT = 16-byte chunks per lane, NT = non-temporal stores, rto = outer rows per block (runtime).
template <int RCT, bool NT>
__global__ __launch_bounds__(1024) void k_mul_rows(const u32x4 *__restrict__ inner, i64 n_chunks, const u32x4 *__restrict__ outer,
                                                    int Wq, i64 o_count, u32x4 *__restrict__ out, int rto, i64 out_stride) {
    const int BT = blockDim.x;                                      // 256 (default) .. 1024 threads: BT * 16 contiguous bytes per row
    const i64 c0 = (i64)blockIdx.x * (BT * RCT) + threadIdx.x;
    u32x4 v[RCT];
    int wq[RCT];
    bool ok[RCT];
#pragma unroll
    for (int k = 0; k < RCT; ++k) {
        const i64 c = c0 + BT * k;
        ok[k] = c < n_chunks;
        v[k] = ok[k] ? inner[c] : (u32x4)(0u);
        wq[k] = ok[k] ? (int)(c % Wq) : 0;
    }
    const i64 ob = (i64)blockIdx.y * rto;
    const i64 oe = ob + rto < o_count ? ob + rto : o_count;
    for (i64 o = ob; o < oe; ++o) {
        const u32x4 *orow = outer + o * Wq;
        u32x4 *dst = out + o * out_stride + c0;
#pragma unroll
        for (int k = 0; k < RCT; ++k) {
            if (ok[k]) {
                u32x4 r = v[k] ^ orow[wq[k]];
                if (NT) __builtin_nontemporal_store(r, dst + BT * k);
                else dst[BT * k] = r;
            }
        }
    }
}


// ---- the HBM-write stream that also leaves the phase sums ----------------------------------------
// WQ = 16-byte chunks per row (= words per X block), a power of two <= 64: a row occupies an aligned group of WQ lanes, its X
// words in the lower half and its Z words in the upper half, so lane ^ WQ/2 holds the other half of the same qubits.
template <int CTRL> __device__ __forceinline__ u32 dpp(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR8 = 0x128;
template <int WQ> __device__ __forceinline__ u32 other_half(u32 v) {
    if (WQ == 2) return dpp<DPP_XOR1>(v);
    if (WQ == 4) return dpp<DPP_XOR2>(v);
    if (WQ == 16) return dpp<DPP_ROR8>(v);
    return (u32)__shfl_xor((int)v, WQ / 2);
}
// sum over the WQ/2 lanes of the half of the row the lane sits in (butterfly; every lane of the half gets the sum)
template <int WQ> __device__ __forceinline__ u32 half_row_sum(u32 s) {
    if (WQ >= 4) s += dpp<DPP_XOR1>(s);
    if (WQ >= 8) s += dpp<DPP_XOR2>(s);
    if (WQ >= 16) s += dpp<DPP_HALF_MIRROR>(s);     // quads hold equal values: lane 7-L supplies the other quad's
    if (WQ >= 32) s += dpp<DPP_MIRROR>(s);
    if (WQ >= 64) s += (u32)__shfl_xor((int)s, 16);
    return s;
}

// One output row segment per block (the sequential write pattern of k_mul_rows<1, true>, rto = 1) plus, per output row, the byte
// (Y_out + 2 |x_left & z_right|) mod 4.  The 256/WQ bytes of a block are gathered in LDS and leave as one store of wave 0.
// e-byte index: o * gx * R + ((bx % 8) * (gx / 8) + bx / 8) * R + row in block, R = 256 / WQ: workgroups go to the XCDs round
// robin, so the bytes of consecutive blocks OF ONE XCD are adjacent and fill whole lines in that XCD's L2 (plain stores) —
// bytes of neighbouring blocks interleaved from 8 different L2s cost partial-line write-backs.
template <int WQ, bool INNER_LEFT>
__global__ __launch_bounds__(256) void k_mul_rows_e(const u32x4 *__restrict__ inner, i64 n_chunks, const u32x4 *__restrict__ outer,
                                                     u32x4 *__restrict__ out, unsigned char *__restrict__ eb, i64 out_stride) {
    constexpr int R = 256 / WQ;
    __shared__ __attribute__((aligned(16))) unsigned char sb[R];
    const i64 cb = (i64)blockIdx.x * 256;
    if (cb >= n_chunks) return;                                      // surplus block of the padded grid
    const i64 c0 = cb + threadIdx.x;
    const bool ok = c0 < n_chunks;
    const u32x4 v = ok ? inner[c0] : (u32x4)(0u);
    const i64 o = blockIdx.y;
    const u32x4 r = outer[o * WQ + (threadIdx.x & (WQ - 1))];
    const u32x4 x = v ^ r;
    if (ok) __builtin_nontemporal_store(x, out + o * out_stride + c0);
    u32 s;
    if constexpr (WQ == 1) {                                         // the chunk is the whole row: x word, z word
        const u32 cy = __popc(x.x & x.z) + __popc(x.y & x.w);
        const u32 cf = INNER_LEFT ? __popc(v.x & r.z) + __popc(v.y & r.w) : __popc(v.z & r.x) + __popc(v.w & r.y);
        s = cy + 2u * cf;
    } else {
        // Y_out: both halves of the row form the same x & z words.  flip: lower half x_inner & z_outer, upper half z_inner & x_outer.
        u32 cy = __popc(x.x & other_half<WQ>(x.x));
        cy += __popc(x.y & other_half<WQ>(x.y));
        cy += __popc(x.z & other_half<WQ>(x.z));
        cy += __popc(x.w & other_half<WQ>(x.w));
        u32 cf = __popc(v.x & other_half<WQ>(r.x));
        cf += __popc(v.y & other_half<WQ>(r.y));
        cf += __popc(v.z & other_half<WQ>(r.z));
        cf += __popc(v.w & other_half<WQ>(r.w));
        s = half_row_sum<WQ>(cy + 2u * cf);
    }
    // the first lane of the half that holds x_left & z_right files the byte
    if ((threadIdx.x & (WQ - 1)) == (INNER_LEFT ? 0 : WQ / 2)) sb[threadIdx.x / WQ] = (unsigned char)(s & 3u);
    __syncthreads();
    if (threadIdx.x < R / 4) {
        const i64 gx = gridDim.x;
        unsigned char *dst = eb + (o * gx + (i64)(blockIdx.x & 7) * (gx >> 3) + (i64)(blockIdx.x >> 3)) * R;
        reinterpret_cast<u32 *>(dst)[threadIdx.x] = reinterpret_cast<const u32 *>(sb)[threadIdx.x];
    }
}

// bytes -> coefficients: c_i * c_o * i^e, e = (3 (Y_i + Y_o) + byte) mod 4.  One lane per inner term, EO outer rows per block,
// 16-byte non-temporal stores, 1 KiB contiguous per wave instruction.
constexpr int EO = 4;
__global__ __launch_bounds__(256) void k_mul_coeff_expand(const unsigned char *__restrict__ eb, i64 gx, int rshift, const int *__restrict__ yi,
                                                           const int *__restrict__ yo, const double *__restrict__ ci, const double *__restrict__ co,
                                                           i64 Ni, i64 No, double *__restrict__ out, i64 out_stride) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= Ni) return;
    const i64 R = 1LL << rshift, bx = i >> rshift;
    const unsigned char *src = eb + ((bx & 7) * (gx >> 3) + (bx >> 3)) * R + (i & (R - 1));
    const double ar = ci[2 * i], ai = ci[2 * i + 1];
    const u32 y = (u32)yi[i];
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2 *dst = reinterpret_cast<f64x2 *>(out) + i;
    const i64 ob = (i64)blockIdx.y * EO;
    if (ob + EO <= No) {
        u32 b[EO];
#pragma unroll
        for (int k = 0; k < EO; ++k) b[k] = src[(ob + k) * gx * R];
#pragma unroll
        for (int k = 0; k < EO; ++k) {
            const i64 o = ob + k;
            const int e = (int)((3u * (y + (u32)yo[o]) + b[k]) & 3u);
            double re, im;
            pair_coefficient(ar, ai, co[2 * o], co[2 * o + 1], e, re, im);
            const f64x2 w = {re, im};
            __builtin_nontemporal_store(w, dst + o * out_stride);
        }
    } else {
        for (i64 o = ob; o < No; ++o) {
            const int e = (int)((3u * (y + (u32)yo[o]) + src[o * gx * R]) & 3u);
            double re, im;
            pair_coefficient(ar, ai, co[2 * o], co[2 * o + 1], e, re, im);
            const f64x2 w = {re, im};
            __builtin_nontemporal_store(w, dst + o * out_stride);
        }
    }
}

// tuning knobs (defaults are the measured best on MI355X; SYMGPU_ROWS_VARIANT="rc,rto,nt[,threads[,pad8]]" overrides for experiments):
// 16-byte chunks per lane, outer rows per block, non-temporal stores, block size, grid.x padded to a multiple of 8
struct RowsVariant { int rc = 1, rto = 1, nt = 1, threads = 256, pad8 = 1; bool parsed = false; };
static RowsVariant g_rv;
static const RowsVariant &rows_variant() {
    if (!g_rv.parsed) {
        g_rv.parsed = true;
        const char *e = SG_TUNE("SYMGPU_ROWS_VARIANT");
        if (e) {
            int a = 0, b2 = 0, c = 0, d = 256, p8 = 1;
            const int got = sscanf(e, "%d,%d,%d,%d,%d", &a, &b2, &c, &d, &p8);
            if (got >= 3 && (a == 1 || a == 2 || a == 4 || a == 8) && b2 >= 1) { g_rv.rc = a; g_rv.rto = b2; g_rv.nt = c != 0; }
            if (got >= 4 && (d == 64 || d == 128 || d == 256 || d == 512 || d == 1024)) g_rv.threads = d;
            if (got >= 5) g_rv.pad8 = p8 != 0;
        }
    }
    return g_rv;
}

static i64 round_up(i64 x, i64 m) { return (x + m - 1) / m * m; }

// coefficients of the slab of outer rows [o_begin, o_end): out_coeff[(o-o_begin)*Ni + i].
// It = word-major inner operand (padded to Ipad, a multiple of 64*PJ); kernels go to stream `st`.
static int mul_coeff_launch(const u64 *It, i64 Ipad, const double *ci, i64 Ni, const u64 *outer, const double *co, i64 o_begin, i64 o_end,
                            int Wq, int inner_is_left, double *out_coeff, hipStream_t st, Scratch &ot,
                            const PairKeyArgs *keys = nullptr) {
    const i64 No = o_end - o_begin;
    const int W = 2 * Wq;
    // P * P in key mode: the word-major copy of the inner operand IS the outer one (its padding is the wider of the two)
    const bool same_operand = keys && keys->squared && o_begin == 0 && No == Ni;
    const i64 Opad = same_operand ? Ipad : round_up(No, PO * PW);
    const u64 *Ot = It;
    if (!same_operand) {
        SG_TRY(ot.alloc((size_t)Opad * W * sizeof(u64)));
        SG_TRY(to_wordmajor(outer + o_begin * W, No, W, ot.as<u64>(), Opad, st));
        Ot = ot.as<u64>();
    }
    // grid.x a multiple of 8: inner tile bx is then always read by XCD bx % 8, whose L2 keeps its eighth of the word-major inner
    // operand across the outer row blocks (same reasoning as in mul_rows_dev)
    const i64 gx = ((Ni + 64 * PJ - 1) / (64 * PJ) + 7) / 8 * 8;
    const i64 gy_total = round_up(No, PO * PW) / (PO * PW);
    const i64 max_gy = 65535;
    for (i64 y0 = 0; y0 < gy_total; y0 += max_gy) {
        const i64 ny = gy_total - y0 < max_gy ? gy_total - y0 : max_gy;
        const i64 ooff = y0 * PO * PW;
        dim3 grid((unsigned)gx, (unsigned)ny);
#define LAUNCH_COEFF(L) hipLaunchKernelGGL((k_mul_coeff<L, 0>), grid, dim3(256), 0, st, It, Ipad, Ni, ci, Ot + ooff, Opad, \
                                              No - ooff, co + 2 * (o_begin + ooff), Wq, out_coeff + 2 * ooff * Ni, PairKeyArgs())
        if (keys) {
            // key mode runs over the whole outer operand (o_begin == 0); the o field stays absolute through o_base
            PairKeyArgs ka = *keys;
            ka.hO += ooff;
            if (!ka.squared) ka.keys += ooff * Ni;                          // dense keys: slot o*Ni + i; squared: compacted, absolute slots
            ka.o_base = ooff;
#define LAUNCH_KEYS(L, M) hipLaunchKernelGGL((k_mul_coeff<L, M>), grid, dim3(256), 0, st, It, Ipad, Ni, (const double *)nullptr, Ot + ooff, Opad, \
                                                No - ooff, (const double *)nullptr, Wq, (double *)nullptr, ka)
            if (ka.squared) { LAUNCH_KEYS(true, 2); }                       // P * P: left and right are the same operand
            else if (inner_is_left) { LAUNCH_KEYS(true, 1); }
            else { LAUNCH_KEYS(false, 1); }
#undef LAUNCH_KEYS
        } else {
            if (inner_is_left) LAUNCH_COEFF(true); else LAUNCH_COEFF(false);
        }
#undef LAUNCH_COEFF
        KERNEL_CHECK();
    }
    return SYMGPU_OK;
}

int mul_coeff_dev(const u64 *inner, const double *ci, i64 Ni, const u64 *outer, const double *co, i64 o_begin, i64 o_end,
                  int Wq, int inner_is_left, double *out_coeff) {
    if (Ni == 0 || o_end - o_begin <= 0) return SYMGPU_OK;
    if (wide_pairs_worthwhile(Ni, o_end - o_begin, Wq))           // few pairs of very long rows: parallel over the words (wide.hip)
        return wide_mul_coeff_dev(inner, ci, Ni, outer, co, o_begin, o_end, Wq, inner_is_left, out_coeff, nullptr);
    const i64 Ipad = round_up(Ni, 64 * PJ);
    Scratch it, ot;
    SG_TRY(it.alloc((size_t)Ipad * 2 * Wq * sizeof(u64)));
    SG_TRY(to_wordmajor(inner, Ni, 2 * Wq, it.as<u64>(), Ipad));
    return mul_coeff_launch(it.as<u64>(), Ipad, ci, Ni, outer, co, o_begin, o_end, Wq, inner_is_left, out_coeff, ctx().stream, ot);
}

int mul_keys_dev(const u64 *inner, i64 Ni, const u64 *outer, i64 No, int Wq, int inner_is_left, PairKeyArgs ka) {
    if (Ni == 0 || No <= 0) return SYMGPU_OK;
    if (wide_pairs_worthwhile(Ni, No, Wq)) return wide_mul_coeff_dev(inner, nullptr, Ni, outer, nullptr, 0, No, Wq, inner_is_left, nullptr, &ka);
    const i64 Ipad = round_up(Ni, 64 * PJ);
    Scratch it, ot;
    SG_TRY(it.alloc((size_t)Ipad * 2 * Wq * sizeof(u64)));
    SG_TRY(to_wordmajor(inner, Ni, 2 * Wq, it.as<u64>(), Ipad));
    return mul_coeff_launch(it.as<u64>(), Ipad, nullptr, Ni, outer, nullptr, 0, No, Wq, inner_is_left, nullptr, ctx().stream, ot, &ka);
}

// Chunks (16 B) of the inner operand per launch of a row stream: all of it while an eighth fits an XCD's L2 with room to spare (3.5 MiB of
// 4), else equal tiles of at most 3 MiB per XCD — whole rows, whole 256-chunk blocks, gx a multiple of 8.  SYMGPU_PRODUCT_TILE_MB (tests)
// sets the tile size in MiB of inner operand.
static i64 inner_tile_chunks(i64 n_chunks, int Wq) {
    i64 budget = (i64)24 << 20, whole = (i64)28 << 20;
    if (const char *e = getenv("SYMGPU_PRODUCT_TILE_MB")) {
        const double mb = atof(e);
        if (mb > 0) budget = whole = (i64)(mb * 1048576.0);
    }
    if (n_chunks * 16 <= whole) return n_chunks;
    const i64 n_tiles = (n_chunks * 16 + budget - 1) / budget;
    i64 unit = 2048;                                                // 8 blocks of 256 chunks ...
    while (unit % Wq) unit += 2048;                                 // ... and whole rows (a power-of-two Wq <= 64 divides 2048)
    const i64 per = (n_chunks + n_tiles - 1) / n_tiles;
    return (per + unit - 1) / unit * unit;
}

// rows of the slab: out_rows[((o-o_begin)*Ni + i)*W + w]
int mul_rows_dev(const u64 *inner, i64 Ni, const u64 *outer, i64 o_begin, i64 o_end, int Wq, u64 *out_rows) {
    const i64 No = o_end - o_begin;
    if (Ni == 0 || No <= 0) return SYMGPU_OK;
    const i64 n_chunks = Ni * Wq;
    const RowsVariant &rv = rows_variant();
    // Workgroups go to the 8 XCDs round-robin by linear id (= by * gx + bx): with gx a multiple of 8 the inner chunk bx is ALWAYS
    // read by XCD bx % 8, so each XCD's 4 MB L2 only ever sees its own eighth of the inner operand (3.2 MB of 25.6 MB at the
    // benchmark size) and keeps it.  The re-reads of the inner operand then stop at the L2 instead of crossing the fabric, which
    // makes ONE output row per block affordable — and that is the sequential write pattern the HBM likes (tools/ubench_rows.hip:
    // 12 rows per block 6.27 TB/s either way; 1 row per block 3.76 TB/s unpadded, 7.14 TB/s padded).  The surplus blocks exit.
    // An inner operand whose eighth does NOT fit (round 6 sweep: 10^5 terms of 2,000 qubits = 6.4 MB per XCD, 0.47 of the HBM peak
    // against 0.8 for 10^4 terms) is cut into tiles that do (inner_tiles): every tile is a launch over all outer rows.
    const i64 max_gy = 65535;
    const i64 gy_total = (No + rv.rto - 1) / rv.rto;
    const i64 tile = inner_tile_chunks(n_chunks, Wq);
    for (i64 c_lo = 0; c_lo < n_chunks; c_lo += tile) {
        const i64 nc = n_chunks - c_lo < tile ? n_chunks - c_lo : tile;
        i64 gx = (nc + (i64)rv.threads * rv.rc - 1) / ((i64)rv.threads * rv.rc);
        if (rv.pad8) gx = (gx + 7) / 8 * 8;
        for (i64 y0 = 0; y0 < gy_total; y0 += max_gy) {
            const i64 ny = gy_total - y0 < max_gy ? gy_total - y0 : max_gy;
            const i64 ooff = y0 * rv.rto;
            dim3 grid((unsigned)gx, (unsigned)ny);
            const u32x4 *pi = reinterpret_cast<const u32x4 *>(inner) + c_lo;
            const u32x4 *po = reinterpret_cast<const u32x4 *>(outer + (o_begin + ooff) * 2 * Wq);
            u32x4 *pd = reinterpret_cast<u32x4 *>(out_rows) + ooff * n_chunks + c_lo;
            ProfScope prof(0);
#define LAUNCH_ROWS(RCV, NTV) hipLaunchKernelGGL((k_mul_rows<RCV, NTV>), grid, dim3(rv.threads), 0, ctx().stream, pi, nc, po, Wq, No - ooff, pd, rv.rto, n_chunks)
            if (rv.nt) {
                if (rv.rc == 1) LAUNCH_ROWS(1, true); else if (rv.rc == 2) LAUNCH_ROWS(2, true); else if (rv.rc == 8) LAUNCH_ROWS(8, true); else LAUNCH_ROWS(4, true);
            } else {
                if (rv.rc == 1) LAUNCH_ROWS(1, false); else if (rv.rc == 2) LAUNCH_ROWS(2, false); else if (rv.rc == 8) LAUNCH_ROWS(8, false); else LAUNCH_ROWS(4, false);
            }
#undef LAUNCH_ROWS
            KERNEL_CHECK();
        }
    }
    return SYMGPU_OK;
}


// rows AND coefficients of the slab [o_begin, o_end) through the phase-byte row stream (k_mul_rows_e + k_mul_coeff_expand)
static bool fused_rows_supported(int Wq) { return Wq >= 1 && Wq <= 64 && (Wq & (Wq - 1)) == 0; }
static int mul_rows_coeff_fused(symgpu_op_s *inner, symgpu_op_s *outer, i64 o_begin, i64 o_end, int inner_is_left, symgpu_op_s *out) {
    const i64 Ni = inner->T, No = o_end - o_begin;
    const int Wq = inner->Wq;
    const i64 n_chunks = Ni * Wq;
    const int *yi = nullptr, *yo = nullptr;
    SG_TRY(op_ycount(inner, &yi));
    SG_TRY(op_ycount(outer, &yo));
    const i64 R = 256 / Wq;
    int rshift = 0;
    while ((1LL << rshift) < R) ++rshift;
    hipStream_t st = ctx().stream;
    const i64 max_gy = 65535 / EO * EO;
    const i64 tile = inner_tile_chunks(n_chunks, Wq);                  // see mul_rows_dev: an inner operand beyond the L2s goes tile by tile
    const i64 gx_max = (((tile < n_chunks ? tile : n_chunks) + 255) / 256 + 7) / 8 * 8;
    Scratch eb;
    SG_TRY(eb.alloc((size_t)(No < max_gy ? No : max_gy) * gx_max * R));
    for (i64 c_lo = 0; c_lo < n_chunks; c_lo += tile) {
        const i64 nc = n_chunks - c_lo < tile ? n_chunks - c_lo : tile;
        const i64 i_lo = c_lo / Wq, ni = nc / Wq;
        const i64 gx = ((nc + 255) / 256 + 7) / 8 * 8;                 // a multiple of 8: see mul_rows_dev
        for (i64 y0 = 0; y0 < No; y0 += max_gy) {
            const i64 ny = No - y0 < max_gy ? No - y0 : max_gy;
            const u32x4 *pi = reinterpret_cast<const u32x4 *>(inner->rows) + c_lo;
            const u32x4 *po = reinterpret_cast<const u32x4 *>(outer->rows + (o_begin + y0) * 2 * Wq);
            u32x4 *pd = reinterpret_cast<u32x4 *>(out->rows) + y0 * n_chunks + c_lo;
            dim3 grid((unsigned)gx, (unsigned)ny);
            {
                ProfScope prof(0);
#define LAUNCH_E(W) do { if (inner_is_left) hipLaunchKernelGGL((k_mul_rows_e<W, true>), grid, dim3(256), 0, st, pi, nc, po, pd, eb.as<unsigned char>(), n_chunks); \
                         else hipLaunchKernelGGL((k_mul_rows_e<W, false>), grid, dim3(256), 0, st, pi, nc, po, pd, eb.as<unsigned char>(), n_chunks); } while (0)
                switch (Wq) {
                    case 1: LAUNCH_E(1); break;
                    case 2: LAUNCH_E(2); break;
                    case 4: LAUNCH_E(4); break;
                    case 8: LAUNCH_E(8); break;
                    case 16: LAUNCH_E(16); break;
                    case 32: LAUNCH_E(32); break;
                    default: LAUNCH_E(64); break;
                }
#undef LAUNCH_E
                KERNEL_CHECK();
            }
            dim3 ge((unsigned)((ni + 255) / 256), (unsigned)((ny + EO - 1) / EO));
            hipLaunchKernelGGL(k_mul_coeff_expand, ge, dim3(256), 0, st, eb.as<unsigned char>(), gx, rshift, yi + i_lo, yo + o_begin + y0, inner->coeff + 2 * i_lo,
                               outer->coeff + 2 * (o_begin + y0), ni, ny, out->coeff + 2 * (y0 * Ni + i_lo), Ni);
            KERNEL_CHECK();
        }
    }
    return SYMGPU_OK;
}

}  // namespace symgpu

using namespace symgpu;

extern "C" {

int symgpu_mul_allpairs_dev(symgpu_op_t inner, symgpu_op_t outer, int64_t o_begin, int64_t o_end, int inner_is_left,
                            symgpu_op_t out) {
    SG_ENTER(inner, outer, out);
    SG_REQUIRE(inner && outer && out, "mul_allpairs_dev: null handle");
    SG_REQUIRE(inner->Wq == outer->Wq && out->Wq == inner->Wq, "mul_allpairs_dev: operands must share Wq");
    SG_REQUIRE(0 <= o_begin && o_begin <= o_end && o_end <= outer->T, "mul_allpairs_dev: bad outer range");
    const i64 rows = (o_end - o_begin) * inner->T;
    if (rows > out->capacity) {
        set_error("mul_allpairs_dev: output capacity %lld < %lld rows", (long long)out->capacity, (long long)rows);
        return SYMGPU_E_CAPACITY;
    }
    op_invalidate(out);
    if (out->coeff && rows > 0) {
        SG_REQUIRE(inner->coeff && outer->coeff, "mul_allpairs_dev: operands have no coefficients");
        // Coefficient kernel (VALU-bound, 16 B/pair) and row stream (HBM-bound, 16*Wq B/pair) run ONE AFTER THE OTHER.  Round 1
        // overlapped them on two streams, which paid while the row stream wrote 12 rows per block (6.1 TB/s either way); the
        // one-row-per-block stream lives on the inner operand staying in each XCD's L2 (mul_rows_dev), and the coefficient kernel's
        // 436 MB of traffic per slab running beside it evicts that: overlapped 1.34 ms per slab, in turn 0.95 + 0.15 = 1.10 ms.
        // SYMGPU_PRODUCT_OVERLAP=1 brings the side stream back for experiments.  (ONE launch doing both was slower still in
        // round 1, 1.86e10 pairs/s: the long VALU prologue of every block delays its stores.)
        // Default since round 2 where the row length allows it: the row stream forms the phase sums on the way (its VALU is idle)
        // and a streaming kernel expands them (SYMGPU_PRODUCT_FUSED=0 selects the two kernels below).
        const char *fe = getenv("SYMGPU_PRODUCT_FUSED");                // read per call: the tests run both paths in one process
        const bool fused = !(fe && fe[0] == '0');
        if (fused && fused_rows_supported(inner->Wq)) {
            SG_TRY(mul_rows_coeff_fused(inner, outer, o_begin, o_end, inner_is_left, out));
            out->T = rows;
            return SYMGPU_OK;
        }
        if (wide_pairs_worthwhile(inner->T, o_end - o_begin, inner->Wq)) {   // few pairs of very long rows (wide.hip)
            SG_TRY(wide_mul_coeff_dev(inner->rows, inner->coeff, inner->T, outer->rows, outer->coeff, o_begin, o_end, inner->Wq, inner_is_left,
                                      out->coeff, nullptr));
            SG_TRY(mul_rows_dev(inner->rows, inner->T, outer->rows, o_begin, o_end, inner->Wq, out->rows));
            out->T = rows;
            return SYMGPU_OK;
        }
        Context &c = ctx();
        const u64 *It = nullptr;
        i64 Ipad = 0;
        SG_TRY(op_wordmajor(inner, 64 * PJ, &It, &Ipad));          // cached across slabs of the same inner operand
        Scratch ot;
        const bool overlap = [] { const char *e = SG_TUNE("SYMGPU_PRODUCT_OVERLAP"); return e && e[0] == '1'; }();
        if (overlap) {
            HIP_TRY(hipEventRecord(c.ev_fork, c.stream));
            HIP_TRY(hipStreamWaitEvent(c.stream2, c.ev_fork, 0));
            int rc = mul_coeff_launch(It, Ipad, inner->coeff, inner->T, outer->rows, outer->coeff, o_begin, o_end, inner->Wq,
                                      inner_is_left, out->coeff, c.stream2, ot);
            hipError_t e1 = hipEventRecord(c.ev_join, c.stream2);
            int rc2 = mul_rows_dev(inner->rows, inner->T, outer->rows, o_begin, o_end, inner->Wq, out->rows);
            hipError_t e2 = hipStreamWaitEvent(c.stream, c.ev_join, 0);
            if (rc != SYMGPU_OK) return rc;
            if (rc2 != SYMGPU_OK) return rc2;
            if (e1 != hipSuccess) return hip_fail(e1, "event record (join)", __FILE__, __LINE__);
            if (e2 != hipSuccess) return hip_fail(e2, "stream wait (join)", __FILE__, __LINE__);
        } else {
            SG_TRY(mul_coeff_launch(It, Ipad, inner->coeff, inner->T, outer->rows, outer->coeff, o_begin, o_end, inner->Wq, inner_is_left, out->coeff,
                                    c.stream, ot));
            SG_TRY(mul_rows_dev(inner->rows, inner->T, outer->rows, o_begin, o_end, inner->Wq, out->rows));
        }
    } else {
        SG_TRY(mul_rows_dev(inner->rows, inner->T, outer->rows, o_begin, o_end, inner->Wq, out->rows));
    }
    out->T = rows;
    return SYMGPU_OK;
}

int symgpu_mul_allpairs(const uint64_t *inner, const double *ci, int64_t Ni, const uint64_t *outer, const double *co,
                        int64_t No, int Wq, int inner_is_left, uint64_t *out_rows, double *out_coeff) {
    SG_TRY(require_ctx());
    SG_REQUIRE(Ni >= 0 && No >= 0 && Wq >= 1, "mul_allpairs: sizes");
    if (Ni == 0 || No == 0) return SYMGPU_OK;
    SG_REQUIRE(inner && outer && ci && co && out_rows && out_coeff, "mul_allpairs: null pointer");
    symgpu_op_t a = nullptr, b = nullptr, o = nullptr;
    int rc = symgpu_op_upload(inner, ci, Ni, Wq, &a);
    if (rc == SYMGPU_OK) rc = symgpu_op_upload(outer, co, No, Wq, &b);
    if (rc == SYMGPU_OK) rc = symgpu_op_alloc(Ni * No, Wq, 1, &o);
    if (rc == SYMGPU_OK) rc = symgpu_mul_allpairs_dev(a, b, 0, No, inner_is_left, o);
    if (rc == SYMGPU_OK) rc = symgpu_op_download(o, out_rows, out_coeff, Ni * No);
    symgpu_op_free(a); symgpu_op_free(b); symgpu_op_free(o);
    return rc;
}

}  // extern "C"
