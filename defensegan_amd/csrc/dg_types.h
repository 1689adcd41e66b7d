// POD tables shared by the host planner (dg_plan.cpp, no HIP dependency) and the device kernels.
#pragma once
#include <stdint.h>

namespace dg {

struct PosEntry {        // one (output position, column tile); sorted by descending work
    int out_off;         // float offset of the position inside an output row
    int n0;              // first output column of this tile
    int tap_begin;       // index into the tap table
    int tap_count;
};
struct TapEntry {
    int a_off;           // float offset inside an input row
    int w_off;           // float offset into the weight buffer
};

}  // namespace dg
