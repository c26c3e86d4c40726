// hyteg_host.hpp -- C++ host layer above the C-ABI (include/hyteg_hip.h).
//
// Mirrors the names and call signatures of the reference classes on the P1 hot path so that code written
// against HyTeG reads the same here:
//   MeshInfo                     src/hyteg/mesh/MeshInfo.hpp:221,512-585
//   PrimitiveStorage             src/hyteg/primitivestorage/PrimitiveStorage.hpp, SetupPrimitiveStorage.cpp
//   P1Function< double >         src/hyteg/p1functionspace/VertexDoFFunction.hpp/.cpp (interpolate, assign, add,
//                                multElementwise, dotLocal, dotGlobal)
//   P1ConstantOperator< Form >   src/constant_stencil_operator/P1ConstantOperator.hpp:33-168 + P1Operator.hpp:192-447
//   P1toP1LinearRestriction / P1toP1LinearProlongation   src/hyteg/gridtransferoperators/
//   WeightedJacobiSmoother, GaussSeidelSmoother, SORSmoother, CGSolver, GeometricMultigridSolver  src/hyteg/solvers/
//
// Data model: one device array per (macro-cell, level) in HyTeG's cell layout.  The DoFs
// of macro-faces/edges/vertices are the boundary entries of every adjacent cell array and are kept bit-identical
// in all copies.  Operators evaluate them as per-cell partial results + one additive exchange ("sumShared"),
// which is what HyTeG itself does for elementwise operators and grid transfer
// (P1ElementwiseOperator.cpp:186-188, P1toP1LinearRestriction.cpp:343-345), instead of the six directed ghost
// phases of P1Operator.hpp:201-207.  All device work goes through the C-ABI; this file launches nothing itself.
#pragma once
#pragma once

#include "types.hpp"
#include "mesh.hpp"
#include "storage.hpp"
#include "p1function.hpp"
#include "p2function.hpp"
#include "forms.hpp"
#include "p1operator.hpp"
#include "p1elementwise.hpp"
#include "p2operator.hpp"
#include "p2gridtransfer.hpp"
#include "gridtransfer.hpp"
#include "solvers.hpp"
#include "stokes.hpp"
#include "taylorhood.hpp"
