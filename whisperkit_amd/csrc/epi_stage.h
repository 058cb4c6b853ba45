// Index maps of gemm256_kernel's LDS-staged epilogues (csrc/gemm.hip), shared with tests/native/epi_stage_check.cpp, which replays them
// for all 64 lanes on the CPU and checks that every element of a 128 x 64 wave tile reaches its row-major place exactly once.
//
// Why: in the accumulator layout of v_mfma_f32_32x32x16 (operands swapped) a lane owns ONE output row and 4 consecutive columns per
// register group, so neighbouring lanes hold neighbouring ROWS: a direct store instruction touches 32 different 128-byte lines with
// 8 - 16 bytes each, and a line is completed by 8 separate instructions.  After the K loop the workgroup's 128 KB of operand stages are
// dead; each wave turns its tile through a private 16 KB slice of them (no workgroup barrier: LDS operations of one wave execute in
// order) so that neighbouring lanes hold neighbouring 16-byte pieces of the SAME row and a store instruction writes whole lines.
// The values, and the order of the additions that make them, are those of the direct epilogues: the outputs are bit-identical.
#pragma once

#if defined(__HIPCC__)
#define WH_EPI_FN __host__ __device__ inline __attribute__((always_inline))
#else
#define WH_EPI_FN inline
#endif

namespace wh {
namespace epi {

constexpr int kWaveRegion = 16384;   // bytes of LDS per wave (8 waves x 16 KB = the two operand stages)

// ---- f16 row-major outputs (EPI_F16, EPI_GELU_F16, the q / k columns of EPI_QKV_ENC): two passes of 64 rows x 64 columns.
// Swapped accumulators: acc[i][j][4 g + e] is row i * 32 + (lane & 31), column j * 32 + 8 g + 4 (lane >> 5) + e.
constexpr int kRow16 = 144;          // 64 columns x 2 bytes + 16: 16-byte aligned rows, 36-dword stride spreads the banks
WH_EPI_FN int f16_write_off(int lane, int i2, int j, int g) { return (i2 * 32 + (lane & 31)) * kRow16 + j * 64 + g * 16 + (lane >> 5) * 8; }   // 8 bytes: 4 columns
WH_EPI_FN int f16_read_row(int lane, int it) { return it * 8 + (lane >> 3); }          // 0 .. 63 within the pass, it = 0 .. 7
WH_EPI_FN int f16_read_col(int lane) { return (lane & 7) * 8; }                         // first of 8 columns (16 bytes): 8 lanes = one 128-byte line
WH_EPI_FN int f16_read_off(int lane, int it) { return f16_read_row(lane, it) * kRow16 + f16_read_col(lane) * 2; }

// ---- fp32 row-major read-modify-write (EPI_RESID_F32): four passes of 32 rows x 64 columns.
constexpr int kRow32 = 272;          // 64 columns x 4 bytes + 16
WH_EPI_FN int f32_write_off(int lane, int j, int g) { return (lane & 31) * kRow32 + j * 128 + g * 32 + (lane >> 5) * 16; }                     // 16 bytes: 4 columns
WH_EPI_FN int f32_read_row(int lane, int it) { return it * 4 + (lane >> 4); }          // 0 .. 31 within the pass, it = 0 .. 7
WH_EPI_FN int f32_read_col(int lane) { return (lane & 15) * 4; }                        // first of 4 columns (16 bytes): 16 lanes = two 128-byte lines
WH_EPI_FN int f32_read_off(int lane, int it) { return f32_read_row(lane, it) * kRow32 + f32_read_col(lane) * 4; }

// ---- V^T of EPI_QKV_ENC (output contiguous along the ROWS t of the product): two passes (j) of 32 columns c x 128 rows.
// Unswapped accumulators: acc[i][j][4 g + r] is row i * 32 + 8 g + 4 (lane >> 5) + r, column j * 32 + (lane & 31).
constexpr int kRowT = 272;           // 128 rows x 2 bytes + 16
WH_EPI_FN int vt_write_off(int lane, int i, int g) { return (lane & 31) * kRowT + (i * 32 + g * 8 + (lane >> 5) * 4) * 2; }                     // 8 bytes: 4 rows t
WH_EPI_FN int vt_read_col(int lane, int it) { return it * 4 + (lane >> 4); }           // column c, 0 .. 31 within the pass, it = 0 .. 7
WH_EPI_FN int vt_read_row(int lane) { return (lane & 15) * 8; }                         // first of 8 rows t (16 bytes): 16 lanes = the tile's 128 rows
WH_EPI_FN int vt_read_off(int lane, int it) { return vt_read_col(lane, it) * kRowT + vt_read_row(lane) * 2; }

static_assert(64 * kRow16 <= kWaveRegion && 32 * kRow32 <= kWaveRegion && 32 * kRowT <= kWaveRegion, "a pass fits the wave's slice");

}  // namespace epi
}  // namespace wh
