// TEST INFRASTRUCTURE (oracle/_ref recipe), NOT PRODUCT CODE.
//
// extern "C" wrapper (our own text) around the reference's FEniCS/UFC-generated element-matrix
// routines, compiled IN PLACE from /root/reference:
//   src/hyteg/forms/form_fenics_generated/p1_tet_diffusion.h:4093-4251  (cell_integral::tabulate_tensor)
//   src/hyteg/forms/form_fenics_generated/p1_tet_mass.h                 (same class shape)
//   src/hyteg/forms/form_fenics_generated/p1_tet_{div,divt,pspg}_tet.h  (blocks of the P1-P1 Stokes operator)
//   src/hyteg/forms/form_fenics_generated/p2_to_p1_tet_div_tet.h, p1_to_p2_tet_divt_tet.h  (mixed blocks of the Taylor-Hood operator)
// These headers depend only on the C++ standard library and the in-tree src/hyteg/fenics/ufc.h,
// so no stand-in header is involved.  Everything else on the hot path (the pystencils-generated
// kernels) includes waLBerla/Eigen headers that are empty submodules in the reference snapshot and
// is therefore NOT built (see DESIGN.md "Oracle").
#include "hyteg/forms/form_fenics_generated/p1_tet_diffusion.h"
#include "hyteg/forms/form_fenics_generated/p1_tet_mass.h"
#include "hyteg/forms/form_fenics_generated/p1_tet_div_tet.h"
#include "hyteg/forms/form_fenics_generated/p1_tet_divt_tet.h"
#include "hyteg/forms/form_fenics_generated/p1_tet_pspg_tet.h"
#include "hyteg/forms/form_fenics_generated/p2_tet_diffusion.h"
#include "hyteg/forms/form_fenics_generated/p2_to_p1_tet_div_tet.h"
#include "hyteg/forms/form_fenics_generated/p1_to_p2_tet_divt_tet.h"

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

// blocks of P1P1StokesOperator (src/constant_stencil_operator/P1ConstantOperator.hpp:178-210): k = 0, 1, 2 -> x, y, z
__attribute__( ( visibility( "default" ) ) ) void ref_p1_tet_div( double* A, const double* coords, int k )
{
   if ( k == 0 )
   {
      p1_tet_div_tet_cell_integral_0_otherwise gen;
      gen.tabulate_tensor( A, nullptr, coords, 0 );
   }
   else if ( k == 1 )
   {
      p1_tet_div_tet_cell_integral_1_otherwise gen;
      gen.tabulate_tensor( A, nullptr, coords, 0 );
   }
   else
   {
      p1_tet_div_tet_cell_integral_2_otherwise gen;
      gen.tabulate_tensor( A, nullptr, coords, 0 );
   }
}
__attribute__( ( visibility( "default" ) ) ) void ref_p1_tet_divt( double* A, const double* coords, int k )
{
   if ( k == 0 )
   {
      p1_tet_divt_tet_cell_integral_0_otherwise gen;
      gen.tabulate_tensor( A, nullptr, coords, 0 );
   }
   else if ( k == 1 )
   {
      p1_tet_divt_tet_cell_integral_1_otherwise gen;
      gen.tabulate_tensor( A, nullptr, coords, 0 );
   }
   else
   {
      p1_tet_divt_tet_cell_integral_2_otherwise gen;
      gen.tabulate_tensor( A, nullptr, coords, 0 );
   }
}
__attribute__( ( visibility( "default" ) ) ) void ref_p1_tet_pspg( double* A, const double* coords )
{
   p1_tet_pspg_tet_cell_integral_0_otherwise gen;
   gen.tabulate_tensor( A, nullptr, coords, 0 );
}

// mixed blocks of P2P1TaylorHoodStokesOperator (src/mixed_operator/P2ToP1ConstantOperator.hpp:90-97, P1ToP2ConstantOperator.hpp):
// A: 40 doubles as tabulate_tensor writes them (div: 4 rows of P1 test functions x 10 P2 columns; divT: 10 P2 rows x 4 P1 columns)
__attribute__( ( visibility( "default" ) ) ) void ref_p2_to_p1_tet_div( double* A, const double* coords, int k )
{
   if ( k == 0 )
   {
      p2_to_p1_tet_div_tet_cell_integral_0_otherwise gen;
      gen.tabulate_tensor( A, nullptr, coords, 0 );
   }
   else if ( k == 1 )
   {
      p2_to_p1_tet_div_tet_cell_integral_1_otherwise gen;
      gen.tabulate_tensor( A, nullptr, coords, 0 );
   }
   else
   {
      p2_to_p1_tet_div_tet_cell_integral_2_otherwise gen;
      gen.tabulate_tensor( A, nullptr, coords, 0 );
   }
}
__attribute__( ( visibility( "default" ) ) ) void ref_p1_to_p2_tet_divt( double* A, const double* coords, int k )
{
   if ( k == 0 )
   {
      p1_to_p2_tet_divt_tet_cell_integral_0_otherwise gen;
      gen.tabulate_tensor( A, nullptr, coords, 0 );
   }
   else if ( k == 1 )
   {
      p1_to_p2_tet_divt_tet_cell_integral_1_otherwise gen;
      gen.tabulate_tensor( A, nullptr, coords, 0 );
   }
   else
   {
      p1_to_p2_tet_divt_tet_cell_integral_2_otherwise gen;
      gen.tabulate_tensor( A, nullptr, coords, 0 );
   }
}
}
