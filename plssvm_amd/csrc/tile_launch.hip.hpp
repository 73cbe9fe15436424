/*
 * tile_launch.hip.hpp -- host-side helpers shared by the two tile-kernel translation units (tile_launch_f32.hip,
 * tile_launch_f64.hip): the kernels are compiled in two units so that the fp32 and fp64 instantiations build in parallel.
 */
#pragma once

#include "lssvm_problem.hip.hpp"

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

namespace lssvm {

/* dynamic LDS above 64 KiB must be opted into once per kernel AND per device; the note is taken only after the call succeeded */
inline void ensure_dynamic_lds_impl(const void *kernel, size_t bytes) {
    static std::mutex m;
    static std::map<std::pair<int, const void *>, size_t> done;
    int device = 0;
    LSSVM_HIP_CHECK(hipGetDevice(&device));
    const std::lock_guard<std::mutex> lock(m);
    const auto key = std::make_pair(device, kernel);
    const auto it = done.find(key);
    if (it != done.end() && it->second == bytes) return;
    LSSVM_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)));
    done[key] = bytes;
}
template <typename K>
static void ensure_dynamic_lds(K kernel, size_t bytes) {
    ensure_dynamic_lds_impl(reinterpret_cast<const void *>(kernel), bytes);
}

/* fills the block -> work item mapping fields and returns the grid size */
template <typename T>
static unsigned finish_mapping(TileArgs<T> &a, int num_jc) {
    a.num_jc = num_jc;  // (dbg and mfma_shape were filled in by the caller from ITS options: set_launch_options)
    return static_cast<unsigned>(a.num_ib) * static_cast<unsigned>(num_jc);
}

/* grid of a launch over the work-item list of the symmetric variant: one workgroup per item, or -- TileArgs::queue, for_each_work_item -- persistent workgroups that draw the
 * items from the problem's counters: as many as the device runs at once (`per_cu` resident workgroups per CU; a kernel that fits only one lets the surplus find the list empty) */
template <typename T>
static dim3 sym_grid(const TileArgs<T> &a, int per_cu = 2) {
    return dim3(static_cast<unsigned>(a.queue != nullptr ? std::min(a.num_items, std::max(per_cu * a.queue_grid, 8)) : a.num_items));
}

/* fp32 "bf16x6" split kernel (tile_launch_f32s.hip); `grid` is used by the full-square variant only */
void launch_split_tile_kernel(const TileArgs<float> &a, int kernel_type, dim3 grid, hipStream_t s);
/* fp32 "f16x3" split kernel (tile_launch_f32h.hip) */
void launch_f16_tile_kernel(const TileArgs<float> &a, int kernel_type, dim3 grid, hipStream_t s);
/* both split kernels with 256-row workgroups on block pairs (tile_launch_f32d.hip): symmetric variant, <= 128 features per pass */
void launch_pair_tile_kernel(const TileArgs<float> &a, int kernel_type, hipStream_t s);
/* rbf / polynomial on more features than a row panel in registers holds (tile_launch_f32x.hip): feature panels inside a tile */
void launch_wide_tile_kernel(const TileArgs<float> &a, int kernel_type, dim3 grid, hipStream_t s);
/* fp64 rbf / polynomial on more than 256 features (tile_launch_f64x.hip): feature panels of 64 inside a sub-tile */
void launch_wide_tile_kernel_f64(const TileArgs<double> &a, int kernel_type, dim3 grid, hipStream_t s);

}  // namespace lssvm
