/* one half of tile_launch_f32.hip as a translation unit of its own (see there) */
#define LSSVM_TU_HALF 2
#include "tile_launch_f32.hip"
