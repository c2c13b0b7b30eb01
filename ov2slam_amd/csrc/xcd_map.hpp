// xcd_map.hpp -- XCD-aware 1-D work-group map, shared by lk3.hip, pyramid.hip and clahe.hip (and compiled on the host by
// tests/test_xcd_map.py).
//
// MI355X deals consecutive work-group ids of a launch round-robin over its 8 XCDs, each with its own 4 MB L2.  Work-groups
// that share data (the tiles / keypoint blocks / LUTs of ONE image) should therefore carry ids of the same residue mod 8 and
// of consecutive rank, so that the shared lines are fetched into one L2 only, and at about the same time.
//
// A launch has `per_item * batch` work-groups, `per_item` for each of `batch` items.  ov2_xcd_map turns the hardware id into
// (item, k), k in [0, per_item): items are taken in groups of 8 -- item q*8 + r owns the ids {8 * (q * per_item + k) + r} --
// and the batch % 8 items left over take the remaining ids in plain order.  The map is a bijection for every per_item >= 1,
// batch >= 1 (checked exhaustively for small sizes by the test).
#pragma once

#if defined(__HIPCC__)
#define OV2_XCD_HD __host__ __device__ __forceinline__
#else
#define OV2_XCD_HD static inline
#endif

OV2_XCD_HD void ov2_xcd_map(int id, int per_item, int batch, int *item, int *k)
{
    const int b8 = batch & ~7;
    if (id < per_item * b8) {
        const int idx = id >> 3, q = idx / per_item;
        *item = q * 8 + (id & 7);
        *k = idx - q * per_item;
    } else {
        const int r = id - per_item * b8, q = r / per_item;
        *item = b8 + q;
        *k = r - q * per_item;
    }
}
