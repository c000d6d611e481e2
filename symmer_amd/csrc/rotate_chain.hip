// rotate_chain.hip — a RUN of Clifford rotations of a clean operator (reference: PauliwordOp.perform_rotations,
// symmer/operators/base.py:1163-1186, every step _rotate_by_single_Pword :1139-1154 followed by cleanup() :1185) with the operator held
// in REGISTERS for the whole run.
//
// For a clean operator (no duplicate rows, every |c| above the threshold — what cleanup() leaves and what a Clifford rotation
// preserves) one rotation by angle k pi/2 is:  rows that anticommute with Q get  row ^= Q, c *= i^e (-i)  (odd k) and  c = -c
// (k in {2, 3});  then the operator is re-ordered  [anticommuting rows | commuting rows], both parts in their previous order — a
// STABLE PARTITION.  The multi-launch forms of rotate.hip materialise that partition after every rotation (two launches and 54 MB of
// traffic per rotation at 10^5 terms of 1,000 qubits: 20 us).  But nothing in the next rotation depends on the ORDER of the rows —
// flags, phases and coefficients are functions of the row alone.  K stable partitions by the bits b_1 .. b_K (b = 0: anticommuting)
// are one stable LSD radix sort by the K-bit number b_K ... b_1.  So:
//
//   k_cchain_reg   every lane keeps its 16-byte chunks of the rows in registers for the whole run (a row = an aligned group of WQ
//                  lanes, DPP lane exchange as in product.hip's row stream); per rotation it forms the flag and the phase
//                  exponent, flips the row, rotates the coefficient and shifts the row's partition bit into a 64-bit key — no LDS,
//                  no barrier, no communication between workgroups, HBM touched once at the start and once at the end.  VALU
//                  bound: ~45 instructions per chunk and rotation.
//   radix sort     of the keys [partition bits : <= 40 | original index : 22] (sort.hip, stable, 8 bits per pass)
//   k_cchain_permute  rows and coefficients to their final places (one gather of whole rows).
//
// Runs longer than 40 rotations are cut into segments.  Used by symgpu_rotate_clifford_chain_dev (rotate.hip) above 128 terms for
// rows of a power-of-two number of 16-byte chunks (<= 32); SYMGPU_CHAIN_REG=0 keeps the multi-launch forms (the tests run both).
#include "common.h"
#include "rotate_common.h"
#include <stdlib.h>

namespace symgpu {

typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int CHAIN_SEG = 40;                     // rotations per segment: 5 sort passes of 8 bits
constexpr int CHAIN_IDX_BITS = 22;

// the clifford_k of the segment's rotations, two bits each (scalar registers: the rotation loop must not start with a load)
struct ChainKs { u64 lo; u32 hi; };

__device__ __forceinline__ u32 popc4(u32x4 a) { return __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w); }

// lane exchanges inside the aligned group of WQ lanes that holds a row; every lane of the group is active, so the DPP forms need no
// defined `old` value (no v_mov in front of them)
template <int CTRL> __device__ __forceinline__ u32 ch_dpp(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int WQ> __device__ __forceinline__ u32 ch_other_half(u32 v) {
    if (WQ == 2) return ch_dpp<0xB1>(v);
    if (WQ == 4) return ch_dpp<0x4E>(v);
    if (WQ == 16) return ch_dpp<0x128>(v);
    return (u32)__shfl_xor((int)v, WQ / 2);
}
template <int WQ> __device__ __forceinline__ u32 ch_row_sum(u32 s) {
    if (WQ >= 2) s += ch_dpp<0xB1>(s);
    if (WQ >= 4) s += ch_dpp<0x4E>(s);
    if (WQ >= 8) s += ch_dpp<0x141>(s);
    if (WQ >= 16) s += ch_dpp<0x140>(s);
    if (WQ >= 32) s += (u32)__shfl_xor((int)s, 16);
    return s;
}
template <int WQ> __device__ __forceinline__ u32 ch_row_xor(u32 s) {
    if (WQ >= 2) s ^= ch_dpp<0xB1>(s);
    if (WQ >= 4) s ^= ch_dpp<0x4E>(s);
    if (WQ >= 8) s ^= ch_dpp<0x141>(s);
    if (WQ >= 16) s ^= ch_dpp<0x140>(s);
    if (WQ >= 32) s ^= (u32)__shfl_xor((int)s, 16);
    return s;
}

template <int WQ, int NCH>
__global__ __launch_bounds__(256, NCH == 4 ? 4 : 6) void k_cchain_reg(const u32x4 *__restrict__ rows, const double *__restrict__ coeff, const u64 *__restrict__ perm, i64 T,
                                                     const u32x4 *__restrict__ qs, const ChainKs ks, int K, u32x4 *__restrict__ out_rows,
                                                     double *__restrict__ out_coeff, u64 *__restrict__ keys) {
    const int tid = threadIdx.x;
    const int c = tid & (WQ - 1);
    const bool xhalf = WQ == 1 || c < WQ / 2;
    const u32 xm = xhalf ? ~0u : 0u;
    const i64 total = T * WQ;
    u32x4 v[NCH];
    u32 klo[NCH], khi[NCH];                                                    // partition bits of rotations 0..9 (bits 22..31) and 10..39
    u32 yp[NCH], ex[NCH];                                                      // Y count of the row; accumulated phase exponent of its coefficient
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const i64 i = ((i64)blockIdx.x * NCH + j) * 256 + tid;
        // perm (the sorted keys of the previous segment): row t of this segment is row perm[t] of the buffer the previous one wrote —
        // the re-ordering is done by this gather instead of by a pass of its own
        i64 src = i / WQ;
        if (perm && i < total) { const i64 s2 = (i64)(perm[src] & (((u64)1 << CHAIN_IDX_BITS) - 1)); src = s2 < T ? s2 : src; }
        v[j] = i < total ? __builtin_nontemporal_load(rows + src * WQ + c) : (u32x4)(0u);
        klo[j] = 0; khi[j] = 0;
        ex[j] = 0;
    }
    // Y count of every row (afterwards maintained: a flipped row's Y count is the Y_out of that rotation)
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if constexpr (WQ == 1) {
            yp[j] = __popc(v[j].x & v[j].z) + __popc(v[j].y & v[j].w);
        } else {
            const u32x4 o = {ch_other_half<WQ>(v[j].x), ch_other_half<WQ>(v[j].y), ch_other_half<WQ>(v[j].z), ch_other_half<WQ>(v[j].w)};
            yp[j] = ch_row_sum<WQ>(popc4(v[j] & o) & xm);
        }
    }
    // the segment's Q rows go to LDS first (<= 40 x 512 bytes): a small operator's rotation is shorter than the latency of a load
    // from L2, so fetching Q(r + 1) during rotation r would leave every rotation waiting for its Q
    __shared__ u32x4 s_q[CHAIN_SEG * WQ];
    for (int i = tid; i < K * WQ; i += 256) s_q[i] = qs[i];
    __syncthreads();
    for (int r = 0; r < K; ++r) {
        const u32x4 q_s = s_q[r * WQ + c], q_o = WQ == 1 ? q_s : s_q[r * WQ + (c ^ (WQ / 2))];     // this lane's chunk of Q and the other half's
        const u32 k = r < 32 ? (u32)(ks.lo >> (2 * r)) & 3u : (ks.hi >> (2 * (r - 32))) & 3u;
        const u32 kneg = (k == 2 || k == 3) ? 2u : 0u;
        const bool low = CHAIN_IDX_BITS + r < 32;
        const u32 bit = 1u << ((CHAIN_IDX_BITS + r) & 31);
        if (k & 1) {
            // odd multiple of pi/2: anticommuting rows become row ^ Q with coefficient c * i^e * (-i) = c * i^(e + 3)
            u32 yq;
            if constexpr (WQ == 1) yq = __popc(q_s.x & q_s.z) + __popc(q_s.y & q_s.w);
            else yq = ch_row_sum<WQ>(popc4(q_s & q_o) & xm);
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const u32x4 x = v[j];
                const u32x4 y = x ^ q_s;                                       // the rotated row; its other half is (other half of x) ^ q_o
                u32 pf, yo;                                                    // pf: parity of |x & zq| + |z & xq| (bit 0) and of |x & zq| (bit 1)
                if constexpr (WQ == 1) {
                    const u32 f = __popc(x.x & q_s.z) + __popc(x.y & q_s.w);
                    pf = ((f + __popc(x.z & q_s.x) + __popc(x.w & q_s.y)) & 1u) | ((f & 1u) << 1);
                    yo = __popc(y.x & y.z) + __popc(y.y & y.w);
                } else {
                    const u32 t = (x.x & q_o.x) ^ (x.y & q_o.y) ^ (x.z & q_o.z) ^ (x.w & q_o.w);
                    const u32 p1 = __popc(t) & 1u;
                    pf = ch_row_xor<WQ>(p1 | ((p1 << 1) & xm));
                    const u32x4 oy = {ch_other_half<WQ>(y.x), ch_other_half<WQ>(y.y), ch_other_half<WQ>(y.z), ch_other_half<WQ>(y.w)};
                    yo = ch_row_sum<WQ>(popc4(y & oy) & xm);
                }
                const u32 am = 0u - (pf & 1u);                                 // all ones if the row anticommutes with Q
                v[j] = u32x4{(y.x & am) | (x.x & ~am), (y.y & am) | (x.y & ~am), (y.z & am) | (x.z & ~am), (y.w & am) | (x.w & ~am)};
                // the row's books (meaningful in its chunk-0 lane): only the exponent is kept, the coefficient is multiplied by i^ex
                // once, at the end (exact: swap / negate)
                const u32 e = 3u * (yp[j] + yq) + yo + (pf & 2u);
                ex[j] += (e + 3u + kneg) & am;
                yp[j] = (yo & am) | (yp[j] & ~am);
                if (low) klo[j] |= bit & ~am; else khi[j] |= bit & ~am;
            }
        } else {
            // even multiple: rows stay, anticommuting coefficients change sign for k = 2
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const u32x4 x = v[j];
                u32 pf;
                if constexpr (WQ == 1) {
                    pf = (__popc(x.x & q_s.z) + __popc(x.y & q_s.w) + __popc(x.z & q_s.x) + __popc(x.w & q_s.y)) & 1u;
                } else {
                    const u32 t = (x.x & q_o.x) ^ (x.y & q_o.y) ^ (x.z & q_o.z) ^ (x.w & q_o.w);
                    pf = ch_row_xor<WQ>(__popc(t) & 1u);
                }
                const u32 am = 0u - (pf & 1u);
                ex[j] += kneg & am;
                if (low) klo[j] |= bit & ~am; else khi[j] |= bit & ~am;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const i64 i = ((i64)blockIdx.x * NCH + j) * 256 + tid;
        if (i < total) {
            __builtin_nontemporal_store(v[j], out_rows + i);
            if (c == 0) {
                const i64 t = i / WQ;
                i64 src = t;
                if (perm) { const i64 s2 = (i64)(perm[t] & (((u64)1 << CHAIN_IDX_BITS) - 1)); src = s2 < T ? s2 : t; }
                const f64x2 cc = reinterpret_cast<const f64x2 *>(coeff)[src];
                double re, im;
                phase_mul(cc.x, cc.y, (int)(ex[j] & 3u), re, im);
                reinterpret_cast<f64x2 *>(out_coeff)[t] = f64x2{re, im};
                keys[t] = ((u64)khi[j] << 32) | (u64)klo[j] | (u64)t;
            }
        }
    }
}

// rows and coefficients from the run's original order to the order of the sorted keys
__global__ __launch_bounds__(256) void k_cchain_permute(const u32x4 *__restrict__ rows, const double *__restrict__ coeff, const u64 *__restrict__ keys, i64 T,
                                                         int Wq, int wsh, u32x4 *__restrict__ out_rows, double *__restrict__ out_coeff) {
    const i64 total = T * Wq;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < total; i += (i64)gridDim.x * 256) {
        const i64 d = i >> wsh;
        const int c = (int)(i - (d << wsh));
        i64 src = (i64)(keys[d] & (((u64)1 << CHAIN_IDX_BITS) - 1));
        if (src >= T) src = d;                                               // (only after a failed sort: stay inside the buffer)
        __builtin_nontemporal_store(rows[src * Wq + c], out_rows + i);
        if (c == 0) reinterpret_cast<f64x2 *>(out_coeff)[d] = reinterpret_cast<const f64x2 *>(coeff)[src];
    }
}

bool clifford_chain_registers_applicable(i64 T, int Wq) {
    if (const char *e = getenv("SYMGPU_CHAIN_REG")) if (e[0] == '0') return false;
    return T >= 1 && T <= ((i64)1 << CHAIN_IDX_BITS) && Wq <= 32 && (Wq & (Wq - 1)) == 0;
}

// K rotations (Q rows qs_dev[K][2 Wq] on the device, ks_host[K] in 0..3) of the clean operator in `a` (T rows); `b` is a second
// operator of the same capacity.  *in_b tells where the result ends up.
int clifford_chain_registers(symgpu_op_t a, symgpu_op_t b, i64 T, const u64 *qs_dev, const int *ks_host, i64 K, int *in_b) {
    hipStream_t st = ctx().stream;
    const int Wq = a->Wq;
    *in_b = 0;
    int wsh = 0;
    while ((1 << wsh) < Wq) ++wsh;
    Scratch kbuf[4];                                                           // keys / sort scratch of the even and of the odd segments
    for (int i = 0; i < 4; ++i) SG_TRY(kbuf[i].alloc((size_t)T * 8));
    const i64 total = T * Wq;
    // chunks per lane: 2 (74 VGPRs: six wavefronts per SIMD, no spills; the kernel is VALU bound and its wavefronts are independent, so
    // many short ones balance best: 4 chunks per lane need 128 VGPRs and run 1.5 rounds of wavefronts at 10^5 terms, 8 spill) unless
    // that leaves fewer than ~4,096 wavefronts
    // (rows of 32 chunks exchange halves through ds_bpermute instead of DPP: their per-rotation overhead wants 4 chunks per lane)
    const int nch = total / 128 >= 8192 ? (Wq == 32 ? 4 : 2) : 1;
    const unsigned grid = (unsigned)((total + 256 * (i64)nch - 1) / (256 * (i64)nch));
    symgpu_op_t cur = a, other = b;
    const u64 *perm = nullptr;
    int seg = 0;
    for (i64 r0 = 0; r0 < K; r0 += CHAIN_SEG, ++seg) {
        const int n = (int)(K - r0 < CHAIN_SEG ? K - r0 : CHAIN_SEG);
        ChainKs ks = {0, 0};
        for (int i = 0; i < n; ++i) {
            if (i < 32) ks.lo |= (u64)(ks_host[r0 + i] & 3) << (2 * i);
            else ks.hi |= (u32)(ks_host[r0 + i] & 3) << (2 * (i - 32));
        }
        const u32x4 *rin = reinterpret_cast<const u32x4 *>(cur->rows), *q4 = reinterpret_cast<const u32x4 *>(qs_dev + r0 * 2 * Wq);
        u32x4 *rout = reinterpret_cast<u32x4 *>(other->rows);
        u64 *knew = kbuf[2 * (seg & 1)].as<u64>(), *ktmp = kbuf[2 * (seg & 1) + 1].as<u64>();
#define CH_LAUNCH(WQV, NCHV) hipLaunchKernelGGL((k_cchain_reg<WQV, NCHV>), dim3(grid), dim3(256), 0, st, rin, cur->coeff, perm, T, q4, ks, n, rout, other->coeff, knew)
#define CH_NCH(WQV) do { if (nch == 1) CH_LAUNCH(WQV, 1); else if (nch == 2) CH_LAUNCH(WQV, 2); else CH_LAUNCH(WQV, 4); } while (0)
        ProfScope *prof = new ProfScope(5);
        switch (Wq) {
            case 1: CH_NCH(1); break;
            case 2: CH_NCH(2); break;
            case 4: CH_NCH(4); break;
            case 8: CH_NCH(8); break;
            case 16: CH_NCH(16); break;
            default: CH_NCH(32); break;
        }
#undef CH_NCH
#undef CH_LAUNCH
        delete prof;
        KERNEL_CHECK();
        bool in_tmp = false, coop = false;
        const int end_bit = CHAIN_IDX_BITS + 8 * ((n + 7) / 8);
        SG_TRY(radix_sort_keys_u64_coop(knew, ktmp, T, CHAIN_IDX_BITS, end_bit, &in_tmp, &coop));     // one launch for all passes up to 2^19 keys
        if (!coop) SG_TRY(radix_sort_keys_u64(knew, ktmp, T, CHAIN_IDX_BITS, end_bit, &in_tmp));
        perm = in_tmp ? ktmp : knew;
        symgpu_op_t t2 = cur; cur = other; other = t2;
    }
    if (perm) {
        // the last segment's rows, still in that segment's input order, to their final places
        i64 pg = (total + 255) / 256;
        if (pg > 16384) pg = 16384;
        hipLaunchKernelGGL(k_cchain_permute, dim3((unsigned)pg), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(cur->rows), cur->coeff, perm, T, Wq, wsh,
                           reinterpret_cast<u32x4 *>(other->rows), other->coeff);
        KERNEL_CHECK();
        cur = other;
    }
    *in_b = cur == b ? 1 : 0;
    HIP_TRY(hipStreamSynchronize(st));                                         // the scratch buffers go back to the allocator on return
    bool timed_out = false;
    SG_TRY(radix_sort_coop_check(&timed_out));
    // the one-launch sort gave up at a barrier (its workgroups were not co-resident, e.g. a shared GPU): radix_sort_coop_check has
    // switched the form off for the process; the caller restores `a` from the untouched input and runs the chain again on the
    // multi-launch sort
    if (timed_out) return CHAIN_RETRY;
    return SYMGPU_OK;
}

}  // namespace symgpu
