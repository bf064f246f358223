// Corpus layout in HBM ("panels"), shared by host and device code.
//
// Rows are fp32, padded to ld = round_up(dim, 16) floats.  16 consecutive rows form a
// PANEL of 16*ld floats.  Inside a panel the 16-column groups follow each other (1 KiB
// each); inside a group the rows follow each other, 16 columns = 64 bytes per row:
//
//     offset(row, col) = (row/16)*16*ld + (col/16)*256 + (row%16)*16 + col%16
//
// Why: lane l = 16*g + r of a wavefront that loads a float4 at group_base + lane_group_offset(l)
// takes part in one fully contiguous 1 KiB burst, and what it receives -- columns 4g..4g+3 of
// row r -- is exactly its share of the A operand of v_mfma_f32_16x16x32_bf16 (row = lane & 15,
// k-slice = lane >> 4).  The exact (fp64) kernels use the same mapping, so a row's dot product
// needs a 2-step cross-lane reduction instead of 6.  A single row is a sequence of whole 64-byte
// pieces (one per group, 1 KiB apart), so the rescoring kernels' row gathers use every byte of
// the sectors they touch (with the groups split by column quarter they used a quarter).
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
    return (row >> 4) * (int64_t)(kPanelRows * ld) + (int64_t)(col >> 4) * kGroupFloats + (row & 15) * 16 + (col & 15);
}

// float offset, inside a group, of the float4 that lane l = 16*g + r owns (row r, columns 4g..4g+3)
MLVDB_HD int32_t lane_group_offset(int32_t lane) { return (lane & 15) * 16 + (lane >> 4) * 4; }

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
