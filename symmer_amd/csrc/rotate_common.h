// rotate_common.h — declarations shared by rotate.hip (multi-launch rotation paths) and rotate_resident.hip (one persistent launch).
#pragma once
#include "common.h"

namespace symgpu {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Generation-tagged entries of the persistent join table (Context::rot_table): [tag = hash >> 32 | generation : 10 | row + 1 : 22].
struct JoinTable {
    u64 *slots;
    u32 mask;          // capacity - 1
    u32 gen;           // 1 .. 1023
    u32 *flags;        // [0] = gen when a duplicate input row was seen
};
__device__ __forceinline__ u64 mix64(u64 h) { h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 29; return h; }
__device__ __forceinline__ u32 jt_gen(u64 v) { return (u32)(v >> 22) & 1023u; }
__device__ __forceinline__ i64 jt_row(u64 v) { return (i64)(v & 0x3FFFFFULL) - 1; }


// The rotation's Pauli row Q BY VALUE in the kernel arguments (rows of <= 64 words): no host-to-device copy in front of a rotation.
struct QArg { u64 w[64]; };

__device__ __forceinline__ void phase_mul(double re, double im, int e, double &ore, double &oim) {
    switch (e & 3) {
        case 0: ore = re; oim = im; break;
        case 1: ore = -im; oim = re; break;
        case 2: ore = -re; oim = -im; break;
        default: ore = im; oim = -re; break;
    }
}


// lane exchange inside an aligned group of WQ lanes that holds one row, 16 bytes per lane (X words in the lower, Z words in the upper half)
template <int CTRL> __device__ __forceinline__ u32 rot_dpp(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
template <int WQ> __device__ __forceinline__ u32 rot_other_half(u32 v) {
    if (WQ == 2) return rot_dpp<0xB1>(v);
    if (WQ == 4) return rot_dpp<0x4E>(v);
    if (WQ == 16) return rot_dpp<0x128>(v);
    return (u32)__shfl_xor((int)v, WQ / 2);
}
template <int WQ> __device__ __forceinline__ u32 rot_row_sum(u32 s) {          // over the WQ lanes of the row (butterfly)
    if (WQ >= 2) s += rot_dpp<0xB1>(s);
    if (WQ >= 4) s += rot_dpp<0x4E>(s);
    if (WQ >= 8) s += rot_dpp<0x141>(s);
    if (WQ >= 16) s += rot_dpp<0x140>(s);
    if (WQ >= 32) s += (u32)__shfl_xor((int)s, 16);
    if (WQ >= 64) s += (u32)__shfl_xor((int)s, 32);
    return s;
}

// counts of one rotation; dup: a duplicate input row was seen by this call's join-table insert (multi-launch path); the resident
// kernel reports a failed row verification or a barrier time-out there (2 / 3).
struct RotCounts { u32 nC, nA, nN, nAnti, dup; };

int host_counts(RotCounts **host, RotCounts **dev);          // pinned, device-mapped host copy of the counts (one per context)
int join_table_for(i64 T, JoinTable *jt);                     // the persistent join table with a fresh generation

// rotate_resident.hip: the whole rotation as ONE persistent launch with the operator's rows resident in LDS.  *done = 0: not
// applicable (operator too large, duplicate status unknown, no cached hashes ...) or verification failed — take the other paths.
int rotate_resident_try(symgpu_op_t in, const u64 *q_host, double cos_t, double sin_t, int clifford_k, double thr, symgpu_op_t *out,
                        int *all_commute, int *done);

// rotate_chain.hip: a run of Clifford rotations of a clean operator with the rows in registers; *in_b: the result is in `b`
constexpr int CHAIN_RETRY = 1;        // clifford_chain_registers: the one-launch sort timed out, buffers invalid, run again
bool clifford_chain_registers_applicable(i64 T, int Wq);
int clifford_chain_registers(symgpu_op_t a, symgpu_op_t b, i64 T, const u64 *qs_dev, const int *ks_host, i64 K, int *in_b);
int rotate_resident_trace(u64 *out, int max_wgs, int *n_wgs);   // phase stamps of the last traced launch (SYMGPU_RES_TRACE=1)

}  // namespace symgpu
