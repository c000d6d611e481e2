// gf2.hip — GF(2) row reduction without row swaps (reference: _rref_binary, symmer/operators/utils.py:292-315)
// and the symmetry-generator kernel built on it (IndependentOp.symmetry_generators,
// symmer/operators/independent_op.py:124-126).
//
// Reference loop: for i = 0..R-1: if row i != 0: pivot = leftmost set column of row i; XOR row i into every
// OTHER row that has that column set.  Sequential in i.  Blocked form used here (bit-exact by construction):
//   panel   — ONE workgroup holds K <= 32 consecutive rows in LDS and runs the reference loop restricted to
//             those rows (Gauss-Jordan inside the block): 3 barriers per pivot, no HBM/L2 round trip.
//             Afterwards the block is the identity on its own pivot columns.
//   select  — for every row r outside the block: f(r) = its bits at the block's pivot columns (taken BEFORE
//             the block is applied).  Because the reduced block is the identity there, the unique combination
//             of block rows that clears those bits is exactly f(r) — the same row the sequential loop yields.
//   sweep   — r ^= XOR_{j in f(r)} block_row_j for all rows, one pass over the matrix per K pivots
//             (pivot rows broadcast from registers; wave-uniform selector -> scalar branches).
// Row-XORs are COUNTED as the reference performs them: with mask_j = set of block rows that held pivot j's
// column at time j, the sequential-time selector of an outside row is t_j = f_j ^ parity(f & mask_j & (2^j-1)),
// so the count is sum_j |mask_j| + sum_r |t(r)|  (derivation in DESIGN.md).
#include "common.h"

namespace symgpu {

constexpr int GK = 32;           // max pivots per block
constexpr int PANEL_THREADS = 1024;

struct PanelInfo {
    int pivw[GK];     // pivot word index or -1
    int pivb[GK];     // pivot bit
    u32 mask[GK];     // block rows (bit r) that held the pivot column at time j, r != j
};

template <bool IN_LDS>
__global__ __launch_bounds__(PANEL_THREADS) void k_panel(u64 *__restrict__ rows, i64 Wc, i64 i0, int kk, PanelInfo *__restrict__ info,
                                                          i64 *__restrict__ pivots, unsigned long long *__restrict__ xor_count) {
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    __shared__ int s_min[2];
    __shared__ u32 s_mask[GK];
    __shared__ int s_pw[GK], s_pb[GK];
    const int tid = threadIdx.x;
    u64 *blk = IN_LDS ? smem : rows + i0 * Wc;
    if (IN_LDS) {
        for (i64 k = tid; k < (i64)kk * Wc; k += PANEL_THREADS) blk[k] = rows[i0 * Wc + k];
    }
    if (tid < 2) s_min[tid] = 0x7fffffff;
    __syncthreads();
    unsigned long long cnt = 0;
    for (int j = 0; j < kk; ++j) {
        u64 *rj = blk + (i64)j * Wc;
        // leftmost non-zero word of row j
        for (i64 w = tid; w < Wc; w += PANEL_THREADS) {
            if (rj[w] != 0) { atomicMin(&s_min[j & 1], (int)w); break; }
        }
        __syncthreads();
        const int w0 = s_min[j & 1];
        if (tid == 0) s_min[(j + 1) & 1] = 0x7fffffff;   // reset the other slot for the next pivot
        if (w0 == 0x7fffffff) {                           // zero row: no pivot
            if (tid == 0) { s_pw[j] = -1; s_pb[j] = 0; s_mask[j] = 0; }
            __syncthreads();
            continue;
        }
        const int b = __builtin_ctzll(rj[w0]);
        u32 mask = 0;
        for (int r = 0; r < kk; ++r)
            if (r != j && ((blk[(i64)r * Wc + w0] >> b) & 1ULL)) mask |= 1u << r;
        if (tid == 0) { s_pw[j] = w0; s_pb[j] = b; s_mask[j] = mask; cnt += __popc(mask); }
        __syncthreads();   // every thread has read the flags before any row changes
        if (mask) {
            for (i64 w = w0 + tid; w < Wc; w += PANEL_THREADS) {   // words left of the pivot word are zero in row j
                const u64 x = rj[w];
                if (x) {
                    u32 m = mask;
                    while (m) {
                        const int r = __builtin_ctz(m);
                        m &= m - 1;
                        blk[(i64)r * Wc + w] ^= x;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (IN_LDS) {
        for (i64 k = tid; k < (i64)kk * Wc; k += PANEL_THREADS) rows[i0 * Wc + k] = blk[k];
    }
    if (tid < GK) {
        const bool live = tid < kk;
        info->pivw[tid] = live ? s_pw[tid] : -1;
        info->pivb[tid] = live ? s_pb[tid] : 0;
        info->mask[tid] = live ? s_mask[tid] : 0u;
        if (live && pivots) pivots[i0 + tid] = s_pw[tid] < 0 ? -1 : (i64)s_pw[tid] * 64 + s_pb[tid];
    }
    if (tid == 0 && cnt) atomicAdd(xor_count, cnt);
}

// ---- register-resident panel ------------------------------------------------------------------------
// 256 threads (one wave per SIMD); thread t holds words t, t+256, ... (WPT of them) of all KB block rows in VGPRs
// (KB*WPT*2 registers, up to 256).  Per pivot: the pivot row is picked with a wave-uniform if-chain, its leftmost
// non-zero word is found with WPT ballots + one 4-entry LDS exchange, the owner lane of that word extracts the pivot
// bit of every block row into a KB-bit mask (second LDS exchange), and every thread XORs the pivot row into the
// flagged rows in registers under scalar branches.  Two barriers per pivot, no LDS atomics, no LDS read-modify-write.
template <int WPT, int KB>
__global__ __launch_bounds__(256) void k_panel_reg(u64 *__restrict__ rows, i64 Wc, i64 i0, int kk, PanelInfo *__restrict__ info,
                                                    i64 *__restrict__ pivots, unsigned long long *__restrict__ xor_count) {
    __shared__ int s_cand[2][4];
    __shared__ u32 s_mask[2];
    __shared__ u32 s_allmask[GK];
    __shared__ int s_pw[GK], s_pb[GK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32 Rl[KB][WPT], Rh[KB][WPT];      // 32-bit halves: no register-pair constraints across the loop back-edge
#pragma unroll
    for (int r = 0; r < KB; ++r)
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const i64 w = tid + 256 * k;
            const u64 v = (r < kk && w < Wc) ? rows[(i0 + r) * Wc + w] : 0ULL;
            Rl[r][k] = (u32)v;
            Rh[r][k] = (u32)(v >> 32);
        }
    u32 cnt = 0;
#pragma unroll 1
    for (int j = 0; j < kk; ++j) {
        const int slot = j & 1;
        u32 pl[WPT], ph[WPT];
#pragma unroll
        for (int k = 0; k < WPT; ++k) pl[k] = ph[k] = 0;
#pragma unroll
        for (int r = 0; r < KB; ++r)
            if (r == j) {                       // wave-uniform: scalar branch, one copy executed
#pragma unroll
                for (int k = 0; k < WPT; ++k) { pl[k] = Rl[r][k]; ph[k] = Rh[r][k]; }
            }
        // leftmost non-zero word of row j: word index = 256*k + tid, increasing in k first
        int cand = 0x7fffffff;
#pragma unroll
        for (int k = WPT - 1; k >= 0; --k) {
            const u64 b = __ballot((pl[k] | ph[k]) != 0);
            if (b) cand = 256 * k + 64 * wave + (int)__builtin_ctzll(b);
        }
        if (lane == 0) s_cand[slot][wave] = cand;
        __syncthreads();
        int w0 = s_cand[slot][0];
#pragma unroll
        for (int q = 1; q < 4; ++q) w0 = min(w0, s_cand[slot][q]);
        if (w0 == 0x7fffffff) {                 // zero row: no pivot (uniform)
            if (tid == 0) { s_pw[j] = -1; s_pb[j] = 0; s_allmask[j] = 0; }
            continue;
        }
        const int k0 = w0 >> 8, t0 = w0 & 255;
        if (tid == t0) {                        // owner of the pivot word: pivot bit + flags of all block rows
            u32 mask = 0;
            int b = 0;
#pragma unroll
            for (int k = 0; k < WPT; ++k)
                if (k == k0) {
                    const bool in_lo = pl[k] != 0;
                    const int bb = in_lo ? __builtin_ctz(pl[k]) : __builtin_ctz(ph[k]);
                    b = in_lo ? bb : 32 + bb;
#pragma unroll
                    for (int r = 0; r < KB; ++r) mask |= (((in_lo ? Rl[r][k] : Rh[r][k]) >> bb) & 1u) << r;
                }
            mask &= ~(1u << j);
            s_mask[slot] = mask;
            s_pw[j] = w0; s_pb[j] = b; s_allmask[j] = mask;
            cnt += __popc(mask);
        }
        __syncthreads();
        const u32 mask = __builtin_amdgcn_readfirstlane(s_mask[slot]);
#pragma unroll
        for (int r = 0; r < KB; ++r) {
            // branch-free: sel is a wave-uniform all-ones/zero word (SGPR); R ^= p & sel is one v_bitop3_b32 per half
            const u32 sel = ((mask >> r) & 1u) ? 0xffffffffu : 0u;
#pragma unroll
            for (int k = 0; k < WPT; ++k) {
                Rl[r][k] = __builtin_amdgcn_bitop3_b32(Rl[r][k], pl[k], sel, 0x78);
                Rh[r][k] = __builtin_amdgcn_bitop3_b32(Rh[r][k], ph[k], sel, 0x78);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < KB; ++r)
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const i64 w = tid + 256 * k;
            if (r < kk && w < Wc) rows[(i0 + r) * Wc + w] = ((u64)Rh[r][k] << 32) | Rl[r][k];
        }
    if (tid < GK) {
        const bool live = tid < kk;
        info->pivw[tid] = live ? s_pw[tid] : -1;
        info->pivb[tid] = live ? s_pb[tid] : 0;
        info->mask[tid] = live ? s_allmask[tid] : 0u;
        if (live && pivots) pivots[i0 + tid] = s_pw[tid] < 0 ? -1 : (i64)s_pw[tid] * 64 + s_pb[tid];
    }
    // the owner lane differs per pivot: reduce the per-thread counts
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if (lane == 0 && cnt) atomicAdd(xor_count, (unsigned long long)cnt);
}

// selector of every row outside the block + reference-order XOR count
__global__ __launch_bounds__(256) void k_select(const u64 *__restrict__ rows, i64 R, i64 Wc, i64 i0, int kk, const PanelInfo *__restrict__ info,
                                                 u32 *__restrict__ sel, unsigned long long *__restrict__ xor_count) {
    __shared__ PanelInfo s_info;
    if (threadIdx.x < GK) {
        s_info.pivw[threadIdx.x] = info->pivw[threadIdx.x];
        s_info.pivb[threadIdx.x] = info->pivb[threadIdx.x];
        s_info.mask[threadIdx.x] = info->mask[threadIdx.x];
    }
    __syncthreads();
    const i64 r = (i64)blockIdx.x * 256 + threadIdx.x;
    u32 f = 0;
    int c = 0;
    if (r < R && (r < i0 || r >= i0 + kk)) {
        const u64 *row = rows + r * Wc;
        for (int j = 0; j < kk; ++j) {
            const int pw = s_info.pivw[j];
            if (pw >= 0) f |= (u32)((row[pw] >> s_info.pivb[j]) & 1ULL) << j;
        }
        for (int j = 0; j < kk; ++j) {
            const u32 tj = ((f >> j) & 1u) ^ (__popc(f & s_info.mask[j] & ((1u << j) - 1u)) & 1u);
            c += (int)tj;
        }
    }
    if (r < R) sel[r] = f;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(xor_count, (unsigned long long)c);
}

constexpr int SW_ROWS = 16;   // rows per sweep block

__global__ __launch_bounds__(256) void k_sweep(u64 *__restrict__ rows, i64 R, i64 Wc, i64 i0, int kk, const u32 *__restrict__ sel) {
    const i64 w = (i64)blockIdx.x * 256 + threadIdx.x;
    const bool live = w < Wc;
    u64 bw[GK];
#pragma unroll
    for (int j = 0; j < GK; ++j) bw[j] = (live && j < kk) ? rows[(i0 + j) * Wc + w] : 0ULL;
    const i64 rb = (i64)blockIdx.y * SW_ROWS;
    for (int k = 0; k < SW_ROWS; ++k) {
        const i64 r = rb + k;
        if (r >= R) break;
        const u32 s = __builtin_amdgcn_readfirstlane(sel[r]);   // uniform
        if (s == 0) continue;
        if (live) {
            u64 x = rows[r * Wc + w];
#pragma unroll
            for (int j = 0; j < GK; ++j)
                if ((s >> j) & 1u) x ^= bw[j];
            rows[r * Wc + w] = x;
        }
    }
}

// ---- in-place reduction of a device matrix -------------------------------------------------------
int rref_dev(u64 *rows, i64 R, i64 Wc, i64 *xor_count, i64 *pivots_host) {
    hipStream_t st = ctx().stream;
    if (xor_count) *xor_count = 0;
    if (R <= 0 || Wc <= 0) return SYMGPU_OK;
    if (Wc >= ((i64)1 << 31)) { set_error("rref: Wc too large"); return SYMGPU_E_INVALID; }
    Scratch info, sel, count, piv;
    SG_TRY(info.alloc(sizeof(PanelInfo)));
    SG_TRY(sel.alloc((size_t)R * 4));
    SG_TRY(count.alloc(16));
    SG_TRY(piv.alloc((size_t)R * 8));
    HIP_TRY(hipMemsetAsync(count.p, 0, 16, st));
    // Panel variant: rows up to 2048 words stay in VGPRs (k_panel_reg); wider rows use the LDS panel (as many rows as fit
    // in LDS, <= 32), and rows wider than the LDS budget run the same code on global memory.
    const size_t lds_budget = 144 * 1024;
    int K = GK, variant = 0;        // 0: registers, 1: LDS, 2: global
    if (Wc <= 1024) K = 32;
    else if (Wc <= 2048) K = 16;
    else {
        K = (int)(lds_budget / ((size_t)Wc * 8));
        variant = 1;
        if (K < 1) { K = 8; variant = 2; }
        if (K > GK) K = GK;
    }
    if (variant == 1) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_panel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget));
    }
    const unsigned gx = (unsigned)((Wc + 255) / 256), gy = (unsigned)((R + SW_ROWS - 1) / SW_ROWS);
    PanelInfo *pinfo = info.as<PanelInfo>();
    i64 *ppiv = piv.as<i64>();
    unsigned long long *pcount = count.as<unsigned long long>();
    for (i64 i0 = 0; i0 < R; i0 += K) {
        const int kk = (int)((R - i0 < K) ? (R - i0) : K);
        if (variant == 0) {
            if (Wc <= 256) hipLaunchKernelGGL((k_panel_reg<1, 32>), dim3(1), dim3(256), 0, st, rows, Wc, i0, kk, pinfo, ppiv, pcount);
            else if (Wc <= 512) hipLaunchKernelGGL((k_panel_reg<2, 32>), dim3(1), dim3(256), 0, st, rows, Wc, i0, kk, pinfo, ppiv, pcount);
            else if (Wc <= 1024) hipLaunchKernelGGL((k_panel_reg<4, 32>), dim3(1), dim3(256), 0, st, rows, Wc, i0, kk, pinfo, ppiv, pcount);
            else hipLaunchKernelGGL((k_panel_reg<8, 16>), dim3(1), dim3(256), 0, st, rows, Wc, i0, kk, pinfo, ppiv, pcount);
        } else if (variant == 1)
            hipLaunchKernelGGL(k_panel<true>, dim3(1), dim3(PANEL_THREADS), (size_t)kk * Wc * 8, st, rows, Wc, i0, kk, pinfo, ppiv, pcount);
        else
            hipLaunchKernelGGL(k_panel<false>, dim3(1), dim3(PANEL_THREADS), 0, st, rows, Wc, i0, kk, pinfo, ppiv, pcount);
        KERNEL_CHECK();
        if (R > kk) {
            hipLaunchKernelGGL(k_select, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, rows, R, Wc, i0, kk, info.as<PanelInfo>(),
                               sel.as<u32>(), count.as<unsigned long long>());
            KERNEL_CHECK();
            ProfScope prof(2);
            hipLaunchKernelGGL(k_sweep, dim3(gx, gy), dim3(256), 0, st, rows, R, Wc, i0, kk, sel.as<u32>());
            KERNEL_CHECK();
        }
    }
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, count.p, 8, hipMemcpyDeviceToHost, st));
    if (pivots_host) HIP_TRY(hipMemcpyAsync(pivots_host, piv.p, (size_t)R * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (xor_count) *xor_count = (i64)h;
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
