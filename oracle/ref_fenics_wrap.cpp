// TEST INFRASTRUCTURE (oracle/_ref recipe), NOT PRODUCT CODE.
//
// extern "C" wrapper (our own text) around the reference's FEniCS/UFC-generated element-matrix
// routines, compiled IN PLACE from /root/reference:
//   src/hyteg/forms/form_fenics_generated/p1_tet_diffusion.h:4093-4251  (cell_integral::tabulate_tensor)
//   src/hyteg/forms/form_fenics_generated/p1_tet_mass.h                 (same class shape)
// These headers depend only on the C++ standard library and the in-tree src/hyteg/fenics/ufc.h,
// so no stand-in header is involved.  Everything else on the hot path (the pystencils-generated
// kernels) includes waLBerla/Eigen headers that are empty submodules in the reference snapshot and
// is therefore NOT built (see DESIGN.md "Oracle").
#include "hyteg/forms/form_fenics_generated/p1_tet_diffusion.h"
#include "hyteg/forms/form_fenics_generated/p1_tet_mass.h"
#include "hyteg/forms/form_fenics_generated/p2_tet_diffusion.h"

extern "C" {

// A: 16 doubles, written exactly as tabulate_tensor writes them; coords: 4 vertices x 3.
__attribute__( ( visibility( "default" ) ) ) void ref_p1_tet_diffusion( double* A, const double* coords )
{
   p1_tet_diffusion_cell_integral_0_otherwise gen;
   gen.tabulate_tensor( A, nullptr, coords, 0 );
}

// A: 100 doubles (10 x 10, FEniCS ordering), as P2FenicsForm::computeLocalStiffnessMatrix calls it
// (src/hyteg/forms/form_fenics_base/P2FenicsForm.cpp:160-175)
__attribute__( ( visibility( "default" ) ) ) void ref_p2_tet_diffusion( double* A, const double* coords )
{
   p2_tet_diffusion_cell_integral_0_otherwise gen;
   gen.tabulate_tensor( A, nullptr, coords, 0 );
}

__attribute__( ( visibility( "default" ) ) ) void ref_p1_tet_mass( double* A, const double* coords )
{
   p1_tet_mass_cell_integral_0_otherwise gen;
   gen.tabulate_tensor( A, nullptr, coords, 0 );
}
}
