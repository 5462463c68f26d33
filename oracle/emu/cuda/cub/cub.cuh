// cub::DeviceRadixSort::SortPairs / cub::DeviceScan::ExclusiveSum as the reference calls them
// (aabb_culling.h:235-241, tile_ops.h): the documented semantics -- stable ascending sort on
// all key bits of a signed key, exclusive prefix sum -- with the two-phase temp-storage protocol.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>
#include "../cuda_runtime.h"
namespace cub {
struct DeviceRadixSort {
  template <typename K, typename V>
  static cudaError_t SortPairs(void *d_temp, size_t &temp_bytes, const K *keys_in, K *keys_out, const V *vals_in,
                               V *vals_out, int num, int = 0, int = sizeof(K) * 8, cudaStream_t = nullptr) {
    if (d_temp == nullptr) { temp_bytes = 16; return 0; }
    std::vector<int> idx(num);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return keys_in[a] < keys_in[b]; });
    for (int i = 0; i < num; ++i) { keys_out[i] = keys_in[idx[i]]; vals_out[i] = vals_in[idx[i]]; }
    return 0;
  }
};
struct DeviceScan {
  template <typename I, typename O>
  static cudaError_t ExclusiveSum(void *d_temp, size_t &temp_bytes, I in, O out, int num, cudaStream_t = nullptr) {
    if (d_temp == nullptr) { temp_bytes = 16; return 0; }
    auto run = decltype(in[0] + in[0])(0);
    for (int i = 0; i < num; ++i) { auto v = in[i]; out[i] = run; run += v; }
    return 0;
  }
};
}  // namespace cub
