// Device-side mirror of hs_colstat (include/hairsplitter_hip.h) and shared constants.
#pragma once
#include <stdint.h>
namespace hsdev {
struct alignas(16) hs_colstat_dev {
    uint8_t key[4];
    uint16_t cnt[5];
    uint16_t depth;
};
static_assert(sizeof(hs_colstat_dev) == 16, "hs_colstat must be 16 bytes");
// result of k_column_top3 == hs_coltop (include/hairsplitter_hip.h)
struct alignas(16) hs_coltop_dev {
    int32_t c0, c1, c2;
    uint8_t k0, k1, tie, pad;
};
static_assert(sizeof(hs_coltop_dev) == 16, "hs_coltop must be 16 bytes");
}  // namespace hsdev
