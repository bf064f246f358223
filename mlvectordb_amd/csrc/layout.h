// Corpus layout in HBM ("panels"), shared by host and device code.
//
// Rows are fp32, padded to ld = round_up(dim, 16) floats.  16 consecutive rows form a
// PANEL of 16*ld floats.  Inside a panel the 16-column groups follow each other (1 KiB
// each); inside a group the order is [g = (col%16)/4][r = row%16][col%4]:
//
//     offset(row, col) = (row/16)*16*ld + (col/16)*256 + ((col%16)/4)*64 + (row%16)*4 + col%4
//
// Why: lane l = 16*g + r of a wavefront that loads a float4 at group_base + 4*l touches
// one fully contiguous, lane-linear 1 KiB burst, and what it receives -- 4 consecutive
// columns of row r -- is exactly its share of the A operand of v_mfma_f32_16x16x32_bf16
// (row = lane & 15, k-slice = lane >> 4).  The exact (fp64) kernels use the same mapping,
// so a row's dot product needs a 2-step cross-lane reduction instead of 6.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MLVDB_HD __host__ __device__ __forceinline__
#else
#define MLVDB_HD inline
#endif

namespace mlvdb {

constexpr int kPanelRows = 16;
constexpr int kGroupFloats = 256;  // one 16-row x 16-col group = 1 KiB
constexpr int kTileRows = 768;     // capacity granule: a multiple of every kernel's row tile (192, 256)

MLVDB_HD int32_t layout_ld(int32_t dim) { return (dim + 15) & ~15; }

MLVDB_HD int64_t layout_offset(int64_t row, int32_t col, int32_t ld) {
    return (row >> 4) * (int64_t)(kPanelRows * ld) + (int64_t)(col >> 4) * kGroupFloats + ((col & 15) >> 2) * 64 +
           (row & 15) * 4 + (col & 3);
}

// bf16 shadow of the corpus (filter scans only): same 16-row panels, 32-column groups of 1 KiB,
// inside a group [g = (col%32)/8][r = row%16][col%8]: lane 16*g + r of a wave reads 16 contiguous
// bytes = its whole A fragment of one v_mfma_f32_16x16x32_bf16 k-step, in natural k order.
// Offset in bf16 elements.
MLVDB_HD int64_t layout_offset_b(int64_t row, int32_t col, int32_t ld) {
    return (row >> 4) * (int64_t)(kPanelRows * ld) + (int64_t)(col >> 5) * 512 + ((col & 31) >> 3) * 128 +
           (row & 15) * 8 + (col & 7);
}

MLVDB_HD int64_t round_up_rows(int64_t rows) { return (rows + kTileRows - 1) / kTileRows * kTileRows; }

// distance spaces (include/mlvdb_hip.h)
constexpr int kSpaceL2 = 0;
constexpr int kSpaceCosine = 1;
constexpr int kSpaceIp = 2;

}  // namespace mlvdb
