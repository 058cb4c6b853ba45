// CPU replay of gemm256_kernel's LDS-staged epilogues (whisperkit_amd/csrc/epi_stage.h), built and run by tests/test_kernel_index_math.py.
// Every accumulator element of a wave's 128 x 64 tile is tagged with its (row, column), pushed through the write map into a 16 KB byte
// image and pulled out through the read map; the check is that each row-major destination receives exactly its own tag, once, that no
// access leaves the wave's slice, that 16-byte reads are 16-byte aligned, and that the lanes of a store instruction cover whole lines.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>

#include "epi_stage.h"

using namespace wh::epi;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static uint16_t tag16(int row, int col) { return (uint16_t)(row * 64 + col); }            // 128 x 64 = 8192 distinct tags
static uint32_t tag32(int row, int col) { return 0x10000u + (uint32_t)(row * 64 + col); }

int main() {
    std::vector<uint8_t> lds(kWaveRegion);
    // ---- f16 row-major, swapped accumulators
    {
        std::vector<int> hits(128 * 64, 0);
        for (int hh = 0; hh < 2; ++hh) {
            std::memset(lds.data(), 0xff, lds.size());
            for (int lane = 0; lane < 64; ++lane)
                for (int i2 = 0; i2 < 2; ++i2)
                    for (int j = 0; j < 2; ++j)
                        for (int g = 0; g < 4; ++g) {
                            const int off = f16_write_off(lane, i2, j, g);
                            CHECK(off >= 0 && off + 8 <= kWaveRegion && off % 8 == 0);
                            for (int e = 0; e < 4; ++e) {
                                const int row = (2 * hh + i2) * 32 + (lane & 31), col = j * 32 + 8 * g + 4 * (lane >> 5) + e;
                                const uint16_t t = tag16(row, col);
                                std::memcpy(&lds[off + 2 * e], &t, 2);
                            }
                        }
            for (int it = 0; it < 8; ++it) {
                std::set<int> lines;
                for (int lane = 0; lane < 64; ++lane) {
                    const int off = f16_read_off(lane, it);
                    CHECK(off >= 0 && off + 16 <= kWaveRegion && off % 16 == 0);
                    const int row = hh * 64 + f16_read_row(lane, it), col = f16_read_col(lane);
                    CHECK(col % 8 == 0 && col + 8 <= 64);
                    for (int e = 0; e < 8; ++e) {
                        uint16_t t;
                        std::memcpy(&t, &lds[off + 2 * e], 2);
                        CHECK(t == tag16(row, col + e));
                        ++hits[row * 64 + col + e];
                    }
                    lines.insert(row);       // a 64-column f16 row of the wave tile is one 128-byte line
                }
                CHECK(lines.size() == 8);    // one store instruction = 8 whole lines
            }
        }
        for (int h : hits) CHECK(h == 1);
    }
    // ---- fp32 row-major, swapped accumulators
    {
        std::vector<int> hits(128 * 64, 0);
        for (int i = 0; i < 4; ++i) {
            std::memset(lds.data(), 0xff, lds.size());
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 2; ++j)
                    for (int g = 0; g < 4; ++g) {
                        const int off = f32_write_off(lane, j, g);
                        CHECK(off >= 0 && off + 16 <= kWaveRegion && off % 16 == 0);
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t t = tag32(i * 32 + (lane & 31), j * 32 + 8 * g + 4 * (lane >> 5) + e);
                            std::memcpy(&lds[off + 4 * e], &t, 4);
                        }
                    }
            for (int it = 0; it < 8; ++it) {
                std::set<int> rows;
                for (int lane = 0; lane < 64; ++lane) {
                    const int off = f32_read_off(lane, it);
                    CHECK(off >= 0 && off + 16 <= kWaveRegion && off % 16 == 0);
                    const int row = i * 32 + f32_read_row(lane, it), col = f32_read_col(lane);
                    CHECK(col % 4 == 0 && col + 4 <= 64);
                    for (int e = 0; e < 4; ++e) {
                        uint32_t t;
                        std::memcpy(&t, &lds[off + 4 * e], 4);
                        CHECK(t == tag32(row, col + e));
                        ++hits[row * 64 + col + e];
                    }
                    rows.insert(row);
                }
                CHECK(rows.size() == 4);     // one instruction = 4 rows x 256 bytes = 8 whole lines
            }
        }
        for (int h : hits) CHECK(h == 1);
    }
    // ---- V^T, unswapped accumulators: destination [column c][row t]
    {
        std::vector<int> hits(128 * 64, 0);
        for (int j = 0; j < 2; ++j) {
            std::memset(lds.data(), 0xff, lds.size());
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i)
                    for (int g = 0; g < 4; ++g) {
                        const int off = vt_write_off(lane, i, g);
                        CHECK(off >= 0 && off + 8 <= kWaveRegion && off % 8 == 0);
                        for (int r = 0; r < 4; ++r) {
                            const uint16_t t = tag16(i * 32 + 8 * g + 4 * (lane >> 5) + r, j * 32 + (lane & 31));
                            std::memcpy(&lds[off + 2 * r], &t, 2);
                        }
                    }
            for (int it = 0; it < 8; ++it) {
                std::set<int> cols;
                for (int lane = 0; lane < 64; ++lane) {
                    const int off = vt_read_off(lane, it);
                    CHECK(off >= 0 && off + 16 <= kWaveRegion && off % 16 == 0);
                    const int col = j * 32 + vt_read_col(lane, it), row = vt_read_row(lane);
                    CHECK(row % 8 == 0 && row + 8 <= 128);
                    for (int e = 0; e < 8; ++e) {
                        uint16_t t;
                        std::memcpy(&t, &lds[off + 2 * e], 2);
                        CHECK(t == tag16(row + e, col));
                        ++hits[(row + e) * 64 + col];
                    }
                    cols.insert(col);
                }
                CHECK(cols.size() == 4);     // one instruction = 4 V^T rows x 256 contiguous bytes
            }
        }
        for (int h : hits) CHECK(h == 1);
    }
    std::printf(fails ? "EPI_STAGE_FAILED %d\n" : "EPI_STAGE_OK\n", fails);
    return fails ? 1 : 0;
}
