// Internal interface of the dataflow SOR sweep (p1_sor_dataflow.hip), used by the C-ABI entry points in p1_sor.hip.
#pragma once

#include "common.hpp"

namespace hyteg_hip {

constexpr int kSorDataflowMinLevel = 3; // tiles are 8 x 8 rows

// One in-place SOR sweep over the inner points of `ncells` macro-cells of one level in ONE launch.
// Weights: either `stencils_dev` (device table [cell][15][15], row 14 = inner stencil, batched form) or `w` (15 host
// doubles, ncells == 1).
int launch_sor_dataflow( int                  ncells,
                         double* const*       u,
                         const double* const* rhs,
                         int                  level,
                         const double*        stencils_dev,
                         const double*        w,
                         double               relax,
                         int                  backwards,
                         hipStream_t          stream );

} // namespace hyteg_hip
