// commute.hip — termwise commutation / adjacency (reference: symmer/operators/base.py:938-971 via
// matmul_GF2, utils.py:9-78) and Y_count (base.py:604-615).
//
//   C[i][j] = NOT parity( |x_i & z'_j| + |z_i & x'_j| )      (True = commute)
//
// Roofline: integer-VALU-bound, not HBM-bound.  Per pair and per 64-bit word the kernel issues 4 VALU
// instructions (two v_bitop3_b32 per 32-bit half: acc ^= xa&zb ; acc ^= za&xb) and one popcount-parity
// at the very end (parity of a sum of popcounts == parity of the popcount of the XOR).  Operands are
// tiny (N*16*Wq bytes) and stay in L2/MALL; the only HBM stream is the 1 B/pair (or 1 bit/pair) output.
//
// Mapping (64-wide wavefronts): both operands are first re-laid word-major (layout.hip).  One wave owns
// 8 rows of A x 512 columns of B: the 8 A-words of a word index arrive in SGPRs through a single
// s_load_dwordx16 (wave-uniform address) and are copied to VGPRs, every lane owns 8 ADJACENT columns of B
// (64 contiguous bytes per word) and keeps 8x8 32-bit XOR accumulators in VGPRs.  No LDS is needed:
// nothing is shared between lanes.
#include "common.h"
#include <stdlib.h>
#include <stdint.h>

namespace symgpu {

constexpr int CI = 8;    // A rows per wave (SGPR operand)
constexpr int WAVES = 4; // waves per block, stacked along i

// gfx950 v_bitop3_b32 with truth table 0x78: acc ^ (b & c) in ONE VALU instruction
__device__ __forceinline__ u32 xor_and(u32 acc, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(acc, b, c, 0x78); }
// force a wave-uniform value into a VGPR (the compiler would otherwise fold the SGPR into every consumer)
__device__ __forceinline__ u32 to_vgpr(u32 s) { u32 v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v; }

// ---- 8 x 8 register tile ---------------------------------------------------------------------------------------------
// Wave = 8 rows of A x 512 columns of B, lane = 8 ADJACENT columns (j = jbase + 8*lane + b): the B words of a lane are 64
// contiguous bytes (4 dwordx4 loads per block), the 8 result bytes of a row leave as ONE 8-byte store (512 B per wave
// store instead of 64 B), and the 4 v_mov that bring an A word into VGPRs are shared by 32 bitop3 instead of 16
// (VALU mix 256 bitop3 : 32 mov per word step = 89 % useful issue slots instead of 80 %).  The low and high halves of a
// word feed ONE 32-bit accumulator (only the parity of the total popcount matters): 64 accumulator VGPRs for 64 pairs.
constexpr int DJ = 8;    // adjacent B columns per lane

template <bool BITS, bool VEC>
__global__ __launch_bounds__(256) void k_commutes(const u64 *__restrict__ At, i64 Npad, i64 N,
                                                    const u64 *__restrict__ Bt, i64 Mpad, i64 M, int Wq,
                                                    uint8_t *__restrict__ out, i64 out_stride, uint8_t *__restrict__ out_bits, i64 bits_stride_bytes) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i64 i0 = ((i64)blockIdx.x * WAVES + wave) * CI;   // wave-uniform
    const i64 jbase = (i64)blockIdx.y * (64 * DJ);
    if (i0 >= Npad) return;

    u32 acc[CI][DJ];
#pragma unroll
    for (int a = 0; a < CI; ++a)
#pragma unroll
        for (int b = 0; b < DJ; ++b) acc[a][b] = 0;

    const u64 *pax = At + i0;
    const u64 *paz = At + (i64)Wq * Npad + i0;
    const u64 *pbx = Bt + jbase + DJ * lane;
    const u64 *pbz = Bt + (i64)Wq * Mpad + jbase + DJ * lane;

    for (int w = 0; w < Wq; ++w) {
        u64 xb[DJ], zb[DJ];
#pragma unroll
        for (int b = 0; b < DJ; ++b) {
            xb[b] = pbx[(i64)w * Mpad + b];
            zb[b] = pbz[(i64)w * Mpad + b];
        }
        u64 xa[CI], za[CI];
#pragma unroll
        for (int a = 0; a < CI; ++a) {
            xa[a] = pax[(i64)w * Npad + a];   // uniform -> scalar loads
            za[a] = paz[(i64)w * Npad + a];
        }
#pragma unroll
        for (int a = 0; a < CI; ++a) {
            // Measured on gfx950 (tools/ubench_bitop.hip): a VALU instruction with an SGPR source issues at ~60 % of the
            // all-VGPR rate, so the wave-uniform A words are copied into VGPRs once (4 v_mov) and reused by 4*DJ bitop3.
            // Tried and rejected (slower on MI355X, 25,000 x 200,000 block at n=2000; this form: 17.4 ms): 8x4 tile with
            // strided columns and two accumulators per pair (19.0 ms), per-lane broadcast vector loads (34.8 ms), A tile
            // in LDS (19.7 ms), A+B tiles in LDS with barriers (25.3 ms), 4x8 tile with explicit prefetch (35.4 ms),
            // 8x16 tile (36.0 ms: register pressure).
            const u32 xal = to_vgpr((u32)xa[a]), xah = to_vgpr((u32)(xa[a] >> 32));
            const u32 zal = to_vgpr((u32)za[a]), zah = to_vgpr((u32)(za[a] >> 32));
#pragma unroll
            for (int b = 0; b < DJ; ++b) {
                u32 t = xor_and(xor_and(acc[a][b], xal, (u32)zb[b]), zal, (u32)xb[b]);
                acc[a][b] = xor_and(xor_and(t, xah, (u32)(zb[b] >> 32)), zah, (u32)(xb[b] >> 32));
            }
        }
    }

    const i64 j0 = jbase + DJ * lane;
#pragma unroll
    for (int a = 0; a < CI; ++a) {
        const i64 i = i0 + a;
        if (i >= N) break;                                      // wave-uniform
        if (BITS) {
#pragma unroll
            for (int q = 0; q < DJ / 8; ++q) {
                u32 byte = 0;
#pragma unroll
                for (int b = 0; b < 8; ++b) byte |= ((!(__popc(acc[a][8 * q + b]) & 1) && j0 + 8 * q + b < M) ? 1u : 0u) << b;
                if (j0 + 8 * q < ((M + 63) & ~(i64)63)) out_bits[i * bits_stride_bytes + (j0 >> 3) + q] = (uint8_t)byte;   // zero padding up to the word end
            }
        } else {
            u64 v[DJ / 8];
#pragma unroll
            for (int q = 0; q < DJ / 8; ++q) {
                v[q] = 0;
#pragma unroll
                for (int b = 0; b < 8; ++b) v[q] |= (u64)(!(__popc(acc[a][8 * q + b]) & 1)) << (8 * b);
            }
            uint8_t *dst = out + i * out_stride + j0;
            if (VEC && j0 + DJ <= M) {
                // rows of any length at any base: the 8-byte stores may be unaligned (global memory takes them; a store that straddles a
                // cache line costs a second transaction, not correctness)
                typedef u64 u64_any __attribute__((aligned(1)));
#pragma unroll
                for (int q = 0; q < DJ / 8; ++q) __builtin_nontemporal_store(v[q], reinterpret_cast<u64_any *>(dst) + q);
            } else {
#pragma unroll
                for (int b = 0; b < DJ; ++b)
                    if (j0 + b < M) dst[b] = (uint8_t)(v[b / 8] >> (8 * (b % 8)));
            }
        }
    }
}

// Y_count: one lane group per row on row-major packed rows (tiny, O(T*Wq))
__global__ void k_ycount(const u64 *__restrict__ rows, i64 T, int Wq, int *__restrict__ out) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) {
        const u64 *r = rows + t * 2 * Wq;
        int c = 0;
        for (int w = 0; w < Wq; ++w) c += __popcll(r[w] & r[Wq + w]);
        out[t] = c;
    }
}

// long rows: one block per row, the words spread over its threads (one thread walking 1.5 million words took 0.2 s)
__global__ __launch_bounds__(256) void k_ycount_long(const u64 *__restrict__ rows, i64 t_base, int Wq, int *__restrict__ out) {
    __shared__ int red[4];
    const i64 t = t_base + blockIdx.x;
    const u64 *r = rows + t * 2 * Wq;
    int c = 0;
    for (int w = threadIdx.x; w < Wq; w += 256) c += __popcll(r[w] & r[Wq + w]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) out[t] = red[0] + red[1] + red[2] + red[3];
}

int ycount_dev(const u64 *rows, i64 T, int Wq, int *out) {
    if (T == 0) return SYMGPU_OK;
    if (Wq >= 512) {
        for (i64 t0 = 0; t0 < T; t0 += 0x7fffffff) {
            const i64 nt = T - t0 < 0x7fffffff ? T - t0 : 0x7fffffff;
            hipLaunchKernelGGL(k_ycount_long, dim3((unsigned)nt), dim3(256), 0, ctx().stream, rows, t0, Wq, out);
            KERNEL_CHECK();
        }
        return SYMGPU_OK;
    }
    int grid = (int)((T + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_ycount, dim3(grid), dim3(256), 0, ctx().stream, rows, T, Wq, out);
    KERNEL_CHECK();
    return SYMGPU_OK;
}

static i64 round_up(i64 x, i64 m) { return (x + m - 1) / m * m; }

// A: N rows, B: M rows, row-major packed device pointers.  Exactly one of out / out_bits is non-null.
// Which kernel: the Four-Russians kernel (commute_m4r.hip) does 1/16 of the VALU work per pair but pays a fixed price per
// workgroup (operand transposes, 64 KiB table per k-block shared by >= 512 rows), so it needs enough rows and columns to fill
// the chip with its 512..1536 x 2048 tiles; the register-tile kernel below serves everything smaller.  SYMGPU_COMMUTE_M4R=1 / 0 forces
// one or the other (the tests run both on every case).
static bool use_m4r(i64 N, i64 M, int Wq) {
    if (const char *e = getenv("SYMGPU_COMMUTE_M4R")) {
        if (e[0] == '1') return true;
        if (e[0] == '0') return false;
    }
    return commutes_m4r_worthwhile(N, M);
}

int commutes_dev(const u64 *A, i64 N, const u64 *B, i64 M, int Wq, uint8_t *out, u64 *out_bits, symgpu_op_s *b_owner) {
    if (N == 0 || M == 0) return SYMGPU_OK;
    if (use_m4r(N, M, Wq)) return commutes_m4r_dev(A, N, B, M, Wq, out, out_bits, b_owner);
    if (wide_pairs_worthwhile(N, M, Wq)) return wide_commutes_dev(A, N, B, M, Wq, out, out_bits);      // few pairs of very long rows
    const int W = 2 * Wq;
    const int cj = DJ;
    const bool same = (B == A && M == N);                 // adjacency: one word-major copy serves both sides
    const i64 Mpad = round_up(M, 64 * cj), Npad = same ? Mpad : round_up(N, CI * WAVES);
    Scratch at, bt;
    SG_TRY(at.alloc((size_t)Npad * W * sizeof(u64)));
    SG_TRY(to_wordmajor(A, N, W, at.as<u64>(), Npad));
    const u64 *Bt = nullptr;
    if (B == A && M == N && Mpad == Npad) {
        Bt = at.as<u64>();
    } else {
        SG_TRY(bt.alloc((size_t)Mpad * W * sizeof(u64)));
        SG_TRY(to_wordmajor(B, M, W, bt.as<u64>(), Mpad));
        Bt = bt.as<u64>();
    }
    // blockIdx.x walks along i (fast) so that consecutive blocks reuse the same B column tile from L2
    const i64 gx = Npad / (CI * WAVES), gy = Mpad / (64 * cj);
    // np.bool_ output: 8-byte stores whatever the row length and the base address (unaligned where they have to be; the last columns of a
    // row that do not fill a lane's 16 go out byte by byte).  Round 6: rows that are not a multiple of 8 bytes used to be written byte by
    // byte altogether — 0.083 ms for a 10,001^2 table against 0.041 ms for 10,000^2.
    u64 *bits_dst = out_bits;
    const i64 Mw = (M + 63) / 64;
    // grid.y is limited to 65535: loop over column super-tiles if needed
    const i64 max_gy = 65535;
    for (i64 y0 = 0; y0 < gy; y0 += max_gy) {
        i64 ny = gy - y0 < max_gy ? gy - y0 : max_gy;
        const i64 joff = y0 * 64 * cj;
        dim3 grid((unsigned)gx, (unsigned)ny);
        ProfScope prof(1);
        if (bits_dst) {
            const i64 stride_bytes = Mw * 8;
            hipLaunchKernelGGL((k_commutes<true, false>), grid, dim3(256), 0, ctx().stream, at.as<u64>(), Npad, N, Bt + joff, Mpad, M - joff, Wq,
                               (uint8_t *)nullptr, (i64)0, reinterpret_cast<uint8_t *>(bits_dst) + (joff >> 3), stride_bytes);
        } else {
            // The output base is shifted so that column j of this launch maps to joff + j of the full row.
            hipLaunchKernelGGL((k_commutes<false, true>), grid, dim3(256), 0, ctx().stream, at.as<u64>(), Npad, N, Bt + joff, Mpad, M - joff, Wq,
                               out + joff, M, (uint8_t *)nullptr, (i64)0);
        }
        KERNEL_CHECK();
    }
    return SYMGPU_OK;
}

}  // namespace symgpu

using namespace symgpu;

extern "C" {

int symgpu_ycount(const uint64_t *rows, int64_t T, int Wq, int64_t *out) {
    SG_TRY(require_ctx());
    SG_REQUIRE(T >= 0 && Wq >= 1 && (T == 0 || (rows && out)), "ycount");
    if (T == 0) return SYMGPU_OK;
    Scratch d, o;
    SG_TRY(d.alloc((size_t)T * 2 * Wq * sizeof(u64)));
    SG_TRY(o.alloc((size_t)T * sizeof(int)));
    HIP_TRY(hipMemcpyAsync(d.p, rows, (size_t)T * 2 * Wq * sizeof(u64), hipMemcpyHostToDevice, ctx().stream));
    count_h2d((size_t)T * 2 * Wq * sizeof(u64)); count_d2h((size_t)T * sizeof(int));
    SG_TRY(ycount_dev(d.as<u64>(), T, Wq, o.as<int>()));
    int *h = (int *)malloc((size_t)T * sizeof(int));
    if (!h) { set_error("host allocation failed"); return SYMGPU_E_NOMEM; }
    hipError_t e = hipMemcpyAsync(h, o.p, (size_t)T * sizeof(int), hipMemcpyDeviceToHost, ctx().stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx().stream);
    if (e != hipSuccess) { free(h); return hip_fail(e, "ycount download", __FILE__, __LINE__); }
    for (i64 t = 0; t < T; ++t) out[t] = h[t];
    free(h);
    return SYMGPU_OK;
}

int symgpu_commutes_dev(symgpu_op_t A, int64_t a_begin, int64_t a_end, symgpu_op_t B, uint8_t *out_dev) {
    SG_ENTER(A, B);
    SG_REQUIRE(A && B && A->Wq == B->Wq, "commutes_dev: operands must share Wq");
    SG_REQUIRE(0 <= a_begin && a_begin <= a_end && a_end <= A->T, "commutes_dev: bad row range");
    SG_REQUIRE(out_dev || a_end == a_begin || B->T == 0, "commutes_dev: null output");
    return commutes_dev(A->rows + a_begin * 2 * A->Wq, a_end - a_begin, B->rows, B->T, A->Wq, out_dev, nullptr, B);
}

int symgpu_commutes_bits_dev(symgpu_op_t A, int64_t a_begin, int64_t a_end, symgpu_op_t B, uint64_t *out_bits_dev) {
    SG_ENTER(A, B);
    SG_REQUIRE(A && B && A->Wq == B->Wq, "commutes_bits_dev: operands must share Wq");
    SG_REQUIRE(0 <= a_begin && a_begin <= a_end && a_end <= A->T, "commutes_bits_dev: bad row range");
    SG_REQUIRE(out_bits_dev || a_end == a_begin || B->T == 0, "commutes_bits_dev: null output");
    return commutes_dev(A->rows + a_begin * 2 * A->Wq, a_end - a_begin, B->rows, B->T, A->Wq, nullptr, out_bits_dev, B);
}

int symgpu_commutes(const uint64_t *A, int64_t N, const uint64_t *B, int64_t M, int Wq, uint8_t *out) {
    SG_TRY(require_ctx());
    SG_REQUIRE(N >= 0 && M >= 0 && Wq >= 1, "commutes: sizes");
    if (N == 0 || M == 0) return SYMGPU_OK;
    SG_REQUIRE(A && B && out, "commutes: null pointer");
    const size_t rb = (size_t)2 * Wq * sizeof(u64);
    Scratch da, db, dout;
    SG_TRY(da.alloc((size_t)N * rb));
    HIP_TRY(hipMemcpyAsync(da.p, A, (size_t)N * rb, hipMemcpyHostToDevice, ctx().stream));
    count_h2d((size_t)N * rb); count_d2h((size_t)N * (size_t)M);
    const u64 *pb = da.as<u64>();
    if (!(B == A && M == N)) {
        SG_TRY(db.alloc((size_t)M * rb));
        HIP_TRY(hipMemcpyAsync(db.p, B, (size_t)M * rb, hipMemcpyHostToDevice, ctx().stream));
        count_h2d((size_t)M * rb);
        pb = db.as<u64>();
    }
    SG_TRY(dout.alloc((size_t)N * (size_t)M));
    SG_TRY(commutes_dev(da.as<u64>(), N, pb, M, Wq, dout.as<uint8_t>(), nullptr));
    prefault_host(out, (size_t)N * (size_t)M);
    HIP_TRY(hipMemcpyAsync(out, dout.p, (size_t)N * (size_t)M, hipMemcpyDeviceToHost, ctx().stream));
    HIP_TRY(hipStreamSynchronize(ctx().stream));
    return SYMGPU_OK;
}

}  // extern "C"
