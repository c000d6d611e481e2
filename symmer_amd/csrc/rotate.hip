// rotate.hip — single-Pauli rotation (reference: PauliwordOp._rotate_by_single_Pword,
// symmer/operators/base.py:1090-1161) as ONE fused device pass over a device-resident operator.
//
//   R(t) P R(t)^+ = P                        if [P, Q] = 0
//                 = cos(t) P + sin(t)(-i P Q) if {P, Q} = 0
//
// analyze : per row, anticommutation parity with Q and the phase exponent e of P*Q
//           (e = (3(Y_P+Y_Q) + Y_out + 2|x_P & z_Q|) mod 4), G lanes per row, xor/popcount + shuffles.
// scan    : positions of anticommuting rows (exclusive scan of flags).
// build   : non-Clifford: stack [commuting | cos * anticommuting | (-i sin) i^e * (anticommuting ^ Q)] in the
//           reference's order (base.py:1158-1161), then first-occurrence cleanup (cleanup.hip) merges P^Q
//           partners; Clifford (angle = k*pi/2): [rotated anticommuting | commuting], no merge (base.py:1139-1154);
//           odd k: row ^ Q with c * i^e * (-i), k in {2,3}: negated (k is NOT reduced mod 4, base.py:1148).
// Equivalent to the reference's three intermediate cleanups when the input has no duplicate rows
// (every operator that left cleanup() qualifies).
#include "common.h"

namespace symgpu {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// flags[t] = 1 iff row t anticommutes with q;  ph[t] = phase exponent e of (row_t * q)
__global__ __launch_bounds__(256) void k_rot_analyze(const u64 *__restrict__ rows, i64 T, int Wq, int G, const u64 *__restrict__ q,
                                                      u32 *__restrict__ flags, uint8_t *__restrict__ ph) {
    const int rows_per_block = 256 / G;
    const int g = threadIdx.x % G, rsub = threadIdx.x / G;
    // Y count of q (every lane redundantly; Wq is small)
    int yq = 0;
    for (int w = 0; w < Wq; ++w) yq += __popcll(q[w] & q[Wq + w]);
    for (i64 t0 = (i64)blockIdx.x * rows_per_block; t0 < T; t0 += (i64)gridDim.x * rows_per_block) {
        const i64 t = t0 + rsub;
        u64 par = 0, flip = 0;
        int yp = 0, yout = 0;
        if (t < T) {
            const u64 *r = rows + t * 2 * Wq;
            for (int w = g; w < Wq; w += G) {
                const u64 x = r[w], z = r[Wq + w], xq = q[w], zq = q[Wq + w];
                par ^= (x & zq) ^ (z & xq);
                flip ^= x & zq;
                yp += __popcll(x & z);
                yout += __popcll((x ^ xq) & (z ^ zq));
            }
        }
        int pp = __popcll(par) & 1, fp = __popcll(flip) & 1;
        for (int off = G >> 1; off > 0; off >>= 1) {
            pp ^= __shfl_xor(pp, off);
            fp ^= __shfl_xor(fp, off);
            yp += __shfl_xor(yp, off);
            yout += __shfl_xor(yout, off);
        }
        if (g == 0 && t < T) {
            flags[t] = (u32)pp;
            ph[t] = (uint8_t)((3 * (yp + yq) + yout + 2 * fp) & 3);
        }
    }
}

// Clifford odd-k: product rows whose coefficient is <= thr are dropped by the reference's `*` (cleanup inside
// _multiply_by_operator, base.py:789-793): fold that into the flag that is scanned.
__global__ void k_rot_keepflags(const u32 *__restrict__ anti, const double *__restrict__ coeff, i64 T, double thr, int drop_small,
                                u32 *__restrict__ keep) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) {
        u32 k = anti[t];
        if (k && drop_small && !(hypot(coeff[2 * t], coeff[2 * t + 1]) > thr)) k = 0;
        keep[t] = k;
    }
}

__device__ __forceinline__ void phase_mul(double re, double im, int e, double &ore, double &oim) {
    switch (e & 3) {
        case 0: ore = re; oim = im; break;
        case 1: ore = -im; oim = re; break;
        case 2: ore = -re; oim = -im; break;
        default: ore = im; oim = -re; break;
    }
}

// coefficients of the stacked operator.  MODE 0: non-Clifford; MODE 1: Clifford.
// apos = exclusive scan of `sel` (anticommuting-and-kept flags); n_sel = its total.
template <int MODE>
__global__ void k_rot_coeff(const double *__restrict__ coeff, const u32 *__restrict__ anti, const u32 *__restrict__ sel,
                            const u32 *__restrict__ apos, const u32 *__restrict__ cpos, const uint8_t *__restrict__ ph, i64 T, i64 n_sel,
                            i64 n_comm, double cos_t, double sin_t, int k, double *__restrict__ out_coeff, u32 *__restrict__ dst_main,
                            u32 *__restrict__ dst_prod) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) {
        const double re = coeff[2 * t], im = coeff[2 * t + 1];
        u32 d_main = 0xffffffffu, d_prod = 0xffffffffu;
        if (MODE == 0) {
            if (!anti[t]) {
                d_main = cpos[t];                                   // commuting rows first, input order
                out_coeff[2 * (i64)d_main] = re; out_coeff[2 * (i64)d_main + 1] = im;
            } else {
                d_main = (u32)(n_comm + apos[t]);                   // cos * P
                out_coeff[2 * (i64)d_main] = __dmul_rn(re, cos_t); out_coeff[2 * (i64)d_main + 1] = __dmul_rn(im, cos_t);
                d_prod = (u32)(n_comm + n_sel + apos[t]);           // (-i sin) * i^e * c   on row P^Q
                double pr, pi;
                phase_mul(re, im, ph[t], pr, pi);
                out_coeff[2 * (i64)d_prod] = __dmul_rn(pi, sin_t); out_coeff[2 * (i64)d_prod + 1] = -__dmul_rn(pr, sin_t);
            }
        } else {
            if (!anti[t]) {
                d_main = (u32)(n_sel + cpos[t]);                    // commuting rows after the rotated ones
                out_coeff[2 * (i64)d_main] = re; out_coeff[2 * (i64)d_main + 1] = im;
            } else if (sel[t]) {
                double pr = re, pi = im;
                if (k & 1) {                                        // c * i^e * (-i)
                    double a, b;
                    phase_mul(re, im, ph[t], a, b);
                    pr = b; pi = -a;
                    d_prod = apos[t];
                } else {
                    d_main = apos[t];
                }
                if (k == 2 || k == 3) { pr = -pr; pi = -pi; }
                const u32 d = (k & 1) ? d_prod : d_main;
                out_coeff[2 * (i64)d] = pr; out_coeff[2 * (i64)d + 1] = pi;
            }
        }
        dst_main[t] = d_main;
        dst_prod[t] = d_prod;
    }
}

// rows of the stacked operator as 16-byte chunks
__global__ void k_rot_rows(const u32x4 *__restrict__ rows, const u32x4 *__restrict__ q, i64 T, int Wq, const u32 *__restrict__ dst_main,
                           const u32 *__restrict__ dst_prod, u32x4 *__restrict__ out) {
    const i64 total = T * Wq;
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (i64)gridDim.x * blockDim.x) {
        const i64 t = idx / Wq;
        const int c = (int)(idx - t * Wq);
        const u32x4 v = rows[idx];
        const u32 dm = dst_main[t], dp = dst_prod[t];
        if (dm != 0xffffffffu) out[(i64)dm * Wq + c] = v;
        if (dp != 0xffffffffu) out[(i64)dp * Wq + c] = v ^ q[c];
    }
}

__global__ void k_not_flags(const u32 *__restrict__ a, i64 T, u32 *__restrict__ out) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) out[t] = a[t] ? 0u : 1u;
}

static int grid_for(i64 n, int block = 256, int cap = 8192) {
    i64 g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace symgpu

using namespace symgpu;

extern "C" {

int symgpu_rotate_single_dev(symgpu_op_t in, const uint64_t *q_row_host, double cos_t, double sin_t, int clifford_k, double thr,
                             symgpu_op_t *out, int *all_commute) {
    SG_TRY(require_ctx());
    SG_REQUIRE(in && q_row_host && out && all_commute, "rotate_single_dev: null argument");
    SG_REQUIRE(in->coeff || in->T == 0, "rotate_single_dev: operator has no coefficients");
    hipStream_t st = ctx().stream;
    const i64 T = in->T;
    const int Wq = in->Wq, W = 2 * Wq;
    *out = nullptr;
    *all_commute = 1;
    if (T == 0) return SYMGPU_OK;
    SG_REQUIRE(T < ((i64)1 << 31), "rotate_single_dev: too many rows");
    Scratch q, anti, sel, apos, cpos, ph, totals, dmain, dprod;
    SG_TRY(q.alloc((size_t)W * 8));
    SG_TRY(anti.alloc((size_t)T * 4));
    SG_TRY(sel.alloc((size_t)T * 4));
    SG_TRY(apos.alloc((size_t)T * 4));
    SG_TRY(cpos.alloc((size_t)T * 4));
    SG_TRY(ph.alloc((size_t)T));
    SG_TRY(totals.alloc(16));
    SG_TRY(dmain.alloc((size_t)T * 4));
    SG_TRY(dprod.alloc((size_t)T * 4));
    HIP_TRY(hipMemcpyAsync(q.p, q_row_host, (size_t)W * 8, hipMemcpyHostToDevice, st));
    int G = 1;
    while (G < Wq && G < 64) G <<= 1;
    {
        const int rpb = 256 / G;
        i64 g = (T + rpb - 1) / rpb;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(k_rot_analyze, dim3((unsigned)g), dim3(256), 0, st, in->rows, T, Wq, G, q.as<u64>(), anti.as<u32>(), ph.as<uint8_t>());
        KERNEL_CHECK();
    }
    const bool clifford = clifford_k >= 0;
    const int drop_small = clifford && (clifford_k & 1);
    hipLaunchKernelGGL(k_rot_keepflags, dim3(grid_for(T)), dim3(256), 0, st, anti.as<u32>(), in->coeff, T, thr, drop_small, sel.as<u32>());
    KERNEL_CHECK();
    u32 *tot = totals.as<u32>();
    SG_TRY(exclusive_scan_u32(sel.as<u32>(), apos.as<u32>(), T, tot));            // positions among selected anticommuting rows
    hipLaunchKernelGGL(k_not_flags, dim3(grid_for(T)), dim3(256), 0, st, anti.as<u32>(), T, cpos.as<u32>());
    KERNEL_CHECK();
    SG_TRY(exclusive_scan_u32(cpos.as<u32>(), cpos.as<u32>(), T, tot + 1));        // positions among commuting rows
    u32 h[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(h, tot, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const i64 n_sel = h[0], n_comm = h[1];
    if (n_comm == T) return SYMGPU_OK;        // every term commutes: identity action (base.py:1131-1133)
    *all_commute = 0;
    const i64 n_stack = clifford ? (n_sel + n_comm) : (n_comm + 2 * n_sel);
    symgpu_op_t stack = nullptr;
    SG_TRY(symgpu_op_alloc(n_stack > 0 ? n_stack : 1, Wq, 1, &stack));
    stack->T = n_stack;
    if (clifford)
        hipLaunchKernelGGL(k_rot_coeff<1>, dim3(grid_for(T)), dim3(256), 0, st, in->coeff, anti.as<u32>(), sel.as<u32>(), apos.as<u32>(),
                           cpos.as<u32>(), ph.as<uint8_t>(), T, n_sel, n_comm, cos_t, sin_t, clifford_k, stack->coeff, dmain.as<u32>(), dprod.as<u32>());
    else
        hipLaunchKernelGGL(k_rot_coeff<0>, dim3(grid_for(T)), dim3(256), 0, st, in->coeff, anti.as<u32>(), sel.as<u32>(), apos.as<u32>(),
                           cpos.as<u32>(), ph.as<uint8_t>(), T, n_sel, n_comm, cos_t, sin_t, clifford_k, stack->coeff, dmain.as<u32>(), dprod.as<u32>());
    hipLaunchKernelGGL(k_rot_rows, dim3(grid_for(T * Wq)), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(in->rows),
                       reinterpret_cast<const u32x4 *>(q.p), T, Wq, dmain.as<u32>(), dprod.as<u32>(), reinterpret_cast<u32x4 *>(stack->rows));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { symgpu_op_free(stack); return hip_fail(e, "rotate build", __FILE__, __LINE__); }
    if (clifford) {
        HIP_TRY(hipStreamSynchronize(st));
        *out = stack;
        return SYMGPU_OK;
    }
    symgpu_op_t res = nullptr;
    int rc = cleanup_core(stack->rows, stack->coeff, n_stack, W, nullptr, 0, nullptr, 0, thr, 1, &res, Wq);
    symgpu_op_free(stack);
    if (rc != SYMGPU_OK) return rc;
    *out = res;
    return SYMGPU_OK;
}

int symgpu_rotate_single(const uint64_t *rows, const double *coeff, int64_t N, int Wq, const uint64_t *q_row, double cos_t, double sin_t,
                         int clifford_k, double thr, uint64_t *out_rows, double *out_coeff, int64_t capacity, int64_t *n_out,
                         int *all_commute) {
    SG_TRY(require_ctx());
    SG_REQUIRE(N >= 0 && Wq >= 1 && q_row && all_commute && n_out, "rotate_single: arguments");
    SG_REQUIRE(N == 0 || (rows && coeff), "rotate_single: null input");
    symgpu_op_t in = nullptr, res = nullptr;
    SG_TRY(symgpu_op_upload(rows, coeff, N, Wq, &in));
    int rc = symgpu_rotate_single_dev(in, q_row, cos_t, sin_t, clifford_k, thr, &res, all_commute);
    symgpu_op_free(in);
    if (rc != SYMGPU_OK) return rc;
    if (*all_commute || !res) { *n_out = N; if (res) symgpu_op_free(res); return SYMGPU_OK; }
    *n_out = res->T;
    if (res->T > capacity) {
        set_error("rotate_single: capacity %lld < %lld rows", (long long)capacity, (long long)res->T);
        rc = SYMGPU_E_CAPACITY;
    } else {
        rc = symgpu_op_download(res, out_rows, out_coeff, capacity);
    }
    symgpu_op_free(res);
    return rc;
}

}  // extern "C"
