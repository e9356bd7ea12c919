// 32x32x16 MFMA fragment helpers of the attention kernels (attention.hip) and the fused LN1 + qkv + attention kernel (fused_block.hip).
#pragma once
#include "common.h"

__device__ __forceinline__ int slot_key(int s2, int h2, int j) {   // key (or query) index of k-slot (s2, lane-half, j)
    return (j & 3) + 8 * (2 * s2 + (j >> 2)) + 4 * h2;
}
__device__ __forceinline__ int acc_row(int r, int h2) { return (r & 3) + 8 * (r >> 2) + 4 * h2; }

// Transposed operand fragment: the 8 k-slots of this lane are tokens slot_key(s2, h2, 0..7) (two runs of four consecutive
// rows of the staged [32 tokens][HD] tile), all at column `col` = 32*block + (lane & 31).  gfx950's LDS transpose read does the
// gather: per 16-lane group, lane t passes the address of row (t >> 2), columns 4*(t & 3) .. +3 of a [4 rows][16 columns]
// block (any row pitch) and receives column t of that block, rows 0..3 (probed: tools/probes/tr_probe.hip).  Two reads per
// fragment instead of eight ds_read_u16 + packing; at cfg-3 the u16 gathers had made the backward kernels LDS-bound.
template <int HD, int PITCH = HD>
__device__ __forceinline__ bf16x8 gather_frag(const bf16_t* lds, int s2, int h2, int col) {
    const int lane = threadIdx.x & 63, t = lane & 15, g = lane >> 4;          // h2 == g >> 1, col & 31 == lane & 31
    int c = (col & ~31) + 16 * (g & 1) + 4 * (t & 3);
    if (HD % 32 != 0) c = min(c, HD - 4);                                     // partial last d-block: those outputs are dropped
    const int row = 16 * s2 + 4 * h2 + (t >> 2);
    const unsigned addr = (unsigned)(uintptr_t)(lds + row * PITCH + c);
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(addr), "n"(8 * PITCH * 2)
                 : "memory");
    U128 u;
    u.u = u32x4{lo[0], lo[1], hi[0], hi[1]};
    return u.v;
}

// Both k-halves (s2 = 0, 1) of ONE staged tile, or of TWO tiles, with a single wait: the four / eight transpose reads are in flight
// together (rows 16*s2 + .. sit at compile-time offsets from one address).  gather_frag waits after every pair of reads; the
// value-times-probability loops issued 12 - 24 of those exposed LDS round trips per key tile.
template <int HD, int PITCH = HD>
__device__ __forceinline__ void gather_frag_s2(const bf16_t* lds, int h2, int col, bf16x8 (&f)[2]) {
    const int lane = threadIdx.x & 63, t = lane & 15, g = lane >> 4;
    int c = (col & ~31) + 16 * (g & 1) + 4 * (t & 3);
    if (HD % 32 != 0) c = min(c, HD - 4);
    const unsigned addr = (unsigned)(uintptr_t)(lds + (4 * h2 + (t >> 2)) * PITCH + c);
    u32x2 r0, r1, r2, r3;
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:%5\n\t"
                 "ds_read_b64_tr_b16 %2, %4 offset:%6\n\tds_read_b64_tr_b16 %3, %4 offset:%7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                 : "v"(addr), "n"(8 * PITCH * 2), "n"(16 * PITCH * 2), "n"(24 * PITCH * 2)
                 : "memory");
    U128 u0, u1;
    u0.u = u32x4{r0[0], r0[1], r1[0], r1[1]};
    u1.u = u32x4{r2[0], r2[1], r3[0], r3[1]};
    f[0] = u0.v; f[1] = u1.v;
}
template <int HD, int PITCH = HD>
__device__ __forceinline__ void gather_frag_2x2(const bf16_t* ldsA, int colA, const bf16_t* ldsB, int colB, int h2, bf16x8 (&fa)[2],
                                                bf16x8 (&fb)[2]) {
    const int lane = threadIdx.x & 63, t = lane & 15, g = lane >> 4;
    int cA = (colA & ~31) + 16 * (g & 1) + 4 * (t & 3), cB = (colB & ~31) + 16 * (g & 1) + 4 * (t & 3);
    if (HD % 32 != 0) { cA = min(cA, HD - 4); cB = min(cB, HD - 4); }
    const int e = (4 * h2 + (t >> 2)) * PITCH;
    const unsigned aA = (unsigned)(uintptr_t)(ldsA + e + cA), aB = (unsigned)(uintptr_t)(ldsB + e + cB);
    u32x2 r0, r1, r2, r3, q0, q1, q2, q3;
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:%10\n\t"
                 "ds_read_b64_tr_b16 %2, %8 offset:%11\n\tds_read_b64_tr_b16 %3, %8 offset:%12\n\t"
                 "ds_read_b64_tr_b16 %4, %9\n\tds_read_b64_tr_b16 %5, %9 offset:%10\n\t"
                 "ds_read_b64_tr_b16 %6, %9 offset:%11\n\tds_read_b64_tr_b16 %7, %9 offset:%12\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
                 : "v"(aA), "v"(aB), "n"(8 * PITCH * 2), "n"(16 * PITCH * 2), "n"(24 * PITCH * 2)
                 : "memory");
    U128 u;
    u.u = u32x4{r0[0], r0[1], r1[0], r1[1]}; fa[0] = u.v;
    u.u = u32x4{r2[0], r2[1], r3[0], r3[1]}; fa[1] = u.v;
    u.u = u32x4{q0[0], q0[1], q1[0], q1[1]}; fb[0] = u.v;
    u.u = u32x4{q2[0], q2[1], q3[0], q3[1]}; fb[1] = u.v;
}

// The same eight transpose reads WITHOUT the wait: the caller overlaps them with MFMAs on the previous fragments and calls gather_wait()
// before the first use (the operands tie the wait to the registers, so hipcc cannot move a use above it).
struct Frag2x2 { u32x2 r[8]; };
template <int HD, int PITCH = HD>
__device__ __forceinline__ void gather_issue_2x2(const bf16_t* ldsA, int colA, const bf16_t* ldsB, int colB, int h2, Frag2x2& f) {
    const int lane = threadIdx.x & 63, t = lane & 15, g = lane >> 4;
    int cA = (colA & ~31) + 16 * (g & 1) + 4 * (t & 3), cB = (colB & ~31) + 16 * (g & 1) + 4 * (t & 3);
    if (HD % 32 != 0) { cA = min(cA, HD - 4); cB = min(cB, HD - 4); }
    const int e = (4 * h2 + (t >> 2)) * PITCH;
    const unsigned aA = (unsigned)(uintptr_t)(ldsA + e + cA), aB = (unsigned)(uintptr_t)(ldsB + e + cB);
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:%10\n\t"
                 "ds_read_b64_tr_b16 %2, %8 offset:%11\n\tds_read_b64_tr_b16 %3, %8 offset:%12\n\t"
                 "ds_read_b64_tr_b16 %4, %9\n\tds_read_b64_tr_b16 %5, %9 offset:%10\n\t"
                 "ds_read_b64_tr_b16 %6, %9 offset:%11\n\tds_read_b64_tr_b16 %7, %9 offset:%12"
                 : "=&v"(f.r[0]), "=&v"(f.r[1]), "=&v"(f.r[2]), "=&v"(f.r[3]), "=&v"(f.r[4]), "=&v"(f.r[5]), "=&v"(f.r[6]), "=&v"(f.r[7])
                 : "v"(aA), "v"(aB), "n"(8 * PITCH * 2), "n"(16 * PITCH * 2), "n"(24 * PITCH * 2)
                 : "memory");
}
__device__ __forceinline__ void gather_wait(Frag2x2& f, bf16x8 (&fa)[2], bf16x8 (&fb)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.r[0]), "+v"(f.r[1]), "+v"(f.r[2]), "+v"(f.r[3]), "+v"(f.r[4]), "+v"(f.r[5]), "+v"(f.r[6]), "+v"(f.r[7])
                 :
                 : "memory");
    U128 u;
    u.u = u32x4{f.r[0][0], f.r[0][1], f.r[1][0], f.r[1][1]}; fa[0] = u.v;
    u.u = u32x4{f.r[2][0], f.r[2][1], f.r[3][0], f.r[3][1]}; fa[1] = u.v;
    u.u = u32x4{f.r[4][0], f.r[4][1], f.r[5][0], f.r[5][1]}; fb[0] = u.v;
    u.u = u32x4{f.r[6][0], f.r[6][1], f.r[7][0], f.r[7][1]}; fb[1] = u.v;
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

