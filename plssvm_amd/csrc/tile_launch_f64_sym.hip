/* one half of tile_launch_f64.hip as a translation unit of its own (see there) */
#define LSSVM_TU_HALF 1
#include "tile_launch_f64.hip"
