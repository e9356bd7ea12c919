// k-major ("transposed") LDS tiles of the LDS-DMA GEMM kernels: slot swizzle + gfx950 transpose reads (ds_read_b64_tr_b16), and the
// fragment read of a k-contiguous tile.  Shared by gemm.hip and bwd_gemm.hip (moved out of gemm.hip in round 5, unchanged).
#pragma once
#include "common.h"
#include "dma_tile.h"

// fragment (8 consecutive k of tile row r) of a k-contiguous [rows][64] tile written by the DMA with the dma_swz64 slot swizzle
__device__ __forceinline__ bf16x8 read_frag_dma(const unsigned char* lds, int r, int kc) {      // 128-byte rows
    return *reinterpret_cast<const bf16x8*>(lds + r * 128 + ((kc ^ dma_swz64(r)) << 4));
}

// slot swizzle of a k-major tile row: 256-byte rows (128 columns) put all four rows of a transpose-read block on the same banks
// -> xor (r & 3) << 1; 128-byte rows (64 columns) alias rows two apart -> xor ((r >> 1) & 1) << 1.
// A transpose read is served 32 lanes at a time (MI355X_MICROARCH.md, LDS table): lanes 0-15 take rows kq..kq+3 and lanes 16-31 rows
// kq+8..kq+11 of the same columns, so rows 8 apart must not share banks either (PMC round 2: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE =
// 0.50 on the TN wgrads with the round-1 function, which only separated the four rows of one block) -> one more slot bit from r >> 3.
template <int COLS> __device__ __forceinline__ int kmajor_swz(int r) {
    return COLS >= 128 ? (((r & 3) << 1) | (((r >> 3) & 1) << 3)) : ((((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2));   // 512-byte rows alias like 256-byte ones
}

template <int COLS = 128>
__device__ __forceinline__ bf16x8 frag_kmajor(const unsigned char* tile, int col16, int kq8, int lane) {
    const int t = lane & 15;
    const int row = kq8 + (t >> 2), col = col16 + 4 * (t & 3);
    const int slot = (col >> 3) ^ kmajor_swz<COLS>(row);
    const unsigned addr = (unsigned)(uintptr_t)(tile + row * (COLS * 2) + slot * 16 + (col & 7) * 2);
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(addr), "n"(4 * COLS * 2)
                 : "memory");
    U128 u;
    u.u = u32x4{lo[0], lo[1], hi[0], hi[1]};
    return u.v;
}

// NF fragments (columns col16, col16 + 16, ..) of one k-major tile -- or of an A and a B tile -- with ONE wait: all transpose
// reads of a k-step are in flight together.  frag_kmajor waits per fragment: FM + FN exposed LDS round trips for FM * FN MFMAs.
template <int COLS>
__device__ __forceinline__ unsigned kmajor_addr(const unsigned char* tile, int col16, int kq8, int lane) {
    const int t = lane & 15;
    const int row = kq8 + (t >> 2), col = col16 + 4 * (t & 3);
    const int slot = (col >> 3) ^ kmajor_swz<COLS>(row);
    return (unsigned)(uintptr_t)(tile + row * (COLS * 2) + slot * 16 + (col & 7) * 2);
}
__device__ __forceinline__ bf16x8 tr_pack(const u32x2 lo, const u32x2 hi) {
    U128 u;
    u.u = u32x4{lo[0], lo[1], hi[0], hi[1]};
    return u.v;
}
template <int COLS, int NF>
__device__ __forceinline__ void frags_kmajor(const unsigned char* tile, int col16, int kq8, int lane, bf16x8 (&out)[NF]) {
    static_assert(NF == 2 || NF == 4, "two or four fragments per call");
    unsigned ad[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) ad[i] = kmajor_addr<COLS>(tile, col16 + 16 * i, kq8, lane);
    u32x2 l[NF], h[NF];
    if constexpr (NF == 2) {
        asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
                     "ds_read_b64_tr_b16 %2, %5\n\tds_read_b64_tr_b16 %3, %5 offset:%6\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1])
                     : "v"(ad[0]), "v"(ad[1]), "n"(4 * COLS * 2)
                     : "memory");
    } else {
        asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:%12\n\t"
                     "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:%12\n\t"
                     "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %10 offset:%12\n\t"
                     "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %11 offset:%12\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1]), "=&v"(l[2]), "=&v"(h[2]), "=&v"(l[3]), "=&v"(h[3])
                     : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "n"(4 * COLS * 2)
                     : "memory");
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) out[i] = tr_pack(l[i], h[i]);
}
// A and B tiles of a wgrad k-step together (2 * NF fragments, one wait)
template <int COLS, int NF>
__device__ __forceinline__ void frags_kmajor_ab(const unsigned char* tA, int colA, const unsigned char* tB, int colB, int kq8, int lane,
                                                bf16x8 (&fa)[NF], bf16x8 (&fb)[NF]) {
    static_assert(NF == 2 || NF == 4, "two or four fragments per operand");
    unsigned aa[NF], ab[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        aa[i] = kmajor_addr<COLS>(tA, colA + 16 * i, kq8, lane);
        ab[i] = kmajor_addr<COLS>(tB, colB + 16 * i, kq8, lane);
    }
    u32x2 la[NF], ha[NF], lb[NF], hb[NF];
    if constexpr (NF == 2) {
        asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:%12\n\t"
                     "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:%12\n\t"
                     "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %10 offset:%12\n\t"
                     "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %11 offset:%12\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(la[0]), "=&v"(ha[0]), "=&v"(la[1]), "=&v"(ha[1]), "=&v"(lb[0]), "=&v"(hb[0]), "=&v"(lb[1]), "=&v"(hb[1])
                     : "v"(aa[0]), "v"(aa[1]), "v"(ab[0]), "v"(ab[1]), "n"(4 * COLS * 2)
                     : "memory");
    } else {
        asm volatile("ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:%24\n\t"
                     "ds_read_b64_tr_b16 %2, %17\n\tds_read_b64_tr_b16 %3, %17 offset:%24\n\t"
                     "ds_read_b64_tr_b16 %4, %18\n\tds_read_b64_tr_b16 %5, %18 offset:%24\n\t"
                     "ds_read_b64_tr_b16 %6, %19\n\tds_read_b64_tr_b16 %7, %19 offset:%24\n\t"
                     "ds_read_b64_tr_b16 %8, %20\n\tds_read_b64_tr_b16 %9, %20 offset:%24\n\t"
                     "ds_read_b64_tr_b16 %10, %21\n\tds_read_b64_tr_b16 %11, %21 offset:%24\n\t"
                     "ds_read_b64_tr_b16 %12, %22\n\tds_read_b64_tr_b16 %13, %22 offset:%24\n\t"
                     "ds_read_b64_tr_b16 %14, %23\n\tds_read_b64_tr_b16 %15, %23 offset:%24\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(la[0]), "=&v"(ha[0]), "=&v"(la[1]), "=&v"(ha[1]), "=&v"(la[2]), "=&v"(ha[2]), "=&v"(la[3]), "=&v"(ha[3]),
                       "=&v"(lb[0]), "=&v"(hb[0]), "=&v"(lb[1]), "=&v"(hb[1]), "=&v"(lb[2]), "=&v"(hb[2]), "=&v"(lb[3]), "=&v"(hb[3])
                     : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ab[0]), "v"(ab[1]), "v"(ab[2]), "v"(ab[3]), "n"(4 * COLS * 2)
                     : "memory");
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) { fa[i] = tr_pack(la[i], ha[i]); fb[i] = tr_pack(lb[i], hb[i]); }
}

