// forms.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// P1 forms (closed-form element matrices) and stencil assembly (src/hyteg/p1functionspace/P1Elements.hpp)
#pragma once

#include "storage.hpp"

namespace hyteg {

// =====================================================================================================
// Forms: first row of the P1 element matrix of a tetrahedron (kernel INPUT, setup only).
// P1FenicsForm< ..., p1_tet_diffusion_cell_integral_0_otherwise >  src/hyteg/forms/form_fenics_base/P1FenicsForm.hpp:96-124
// -> src/hyteg/forms/form_fenics_generated/p1_tet_diffusion.h:4113-4240: K_0j = |det J|/6 grad(lambda_0).grad(lambda_j);
// p1_tet_mass.h: M_0j = |det J|/120 (1 + delta_0j).
// =====================================================================================================
namespace forms {
inline double det3( const double J[3][3] )
{
   return J[0][0] * ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) - J[0][1] * ( J[1][0] * J[2][2] - J[1][2] * J[2][0] ) +
          J[0][2] * ( J[1][0] * J[2][1] - J[1][1] * J[2][0] );
}
struct P1LaplaceForm
{
   static void integrateRow0( const std::array< Point3D, 4 >& c, double row[4] )
   {
      double J[3][3];
      for ( int r = 0; r < 3; ++r )
         for ( int k = 0; k < 3; ++k )
            J[r][k] = c[k + 1][r] - c[0][r];
      const double det = det3( J );
      double       Ji[3][3];
      Ji[0][0] = ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) / det;
      Ji[0][1] = ( J[0][2] * J[2][1] - J[0][1] * J[2][2] ) / det;
      Ji[0][2] = ( J[0][1] * J[1][2] - J[0][2] * J[1][1] ) / det;
      Ji[1][0] = ( J[1][2] * J[2][0] - J[1][0] * J[2][2] ) / det;
      Ji[1][1] = ( J[0][0] * J[2][2] - J[0][2] * J[2][0] ) / det;
      Ji[1][2] = ( J[0][2] * J[1][0] - J[0][0] * J[1][2] ) / det;
      Ji[2][0] = ( J[1][0] * J[2][1] - J[1][1] * J[2][0] ) / det;
      Ji[2][1] = ( J[0][1] * J[2][0] - J[0][0] * J[2][1] ) / det;
      Ji[2][2] = ( J[0][0] * J[1][1] - J[0][1] * J[1][0] ) / det;
      double g[4][3];
      for ( int r = 0; r < 3; ++r )
      {
         g[1][r] = Ji[0][r];
         g[2][r] = Ji[1][r];
         g[3][r] = Ji[2][r];
         g[0][r] = -( Ji[0][r] + Ji[1][r] + Ji[2][r] );
      }
      const double vol6 = std::fabs( det ) / 6.0;
      for ( int j = 0; j < 4; ++j )
         row[j] = vol6 * ( g[0][0] * g[j][0] + g[0][1] * g[j][1] + g[0][2] * g[j][2] );
   }
};
// barycentric gradients g[4][3] and volume of a tetrahedron
inline double tetGradients( const std::array< Point3D, 4 >& c, double g[4][3] )
{
   double J[3][3];
   for ( int r = 0; r < 3; ++r )
      for ( int k = 0; k < 3; ++k )
         J[r][k] = c[k + 1][r] - c[0][r];
   const double det = det3( J );
   double       Ji[3][3];
   Ji[0][0] = ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) / det;
   Ji[0][1] = ( J[0][2] * J[2][1] - J[0][1] * J[2][2] ) / det;
   Ji[0][2] = ( J[0][1] * J[1][2] - J[0][2] * J[1][1] ) / det;
   Ji[1][0] = ( J[1][2] * J[2][0] - J[1][0] * J[2][2] ) / det;
   Ji[1][1] = ( J[0][0] * J[2][2] - J[0][2] * J[2][0] ) / det;
   Ji[1][2] = ( J[0][2] * J[1][0] - J[0][0] * J[1][2] ) / det;
   Ji[2][0] = ( J[1][0] * J[2][1] - J[1][1] * J[2][0] ) / det;
   Ji[2][1] = ( J[0][1] * J[2][0] - J[0][0] * J[2][1] ) / det;
   Ji[2][2] = ( J[0][0] * J[1][1] - J[0][1] * J[1][0] ) / det;
   for ( int r = 0; r < 3; ++r )
   {
      g[1][r] = Ji[0][r];
      g[2][r] = Ji[1][r];
      g[3][r] = Ji[2][r];
      g[0][r] = -( Ji[0][r] + Ji[1][r] + Ji[2][r] );
   }
   return std::fabs( det ) / 6.0;
}
// Blocks of the P1-P1 Stokes operator (src/constant_stencil_operator/P1ConstantOperator.hpp:178-210; UFL sources
// data/operators/form_{div,divt,pspg}_tet.ufl).  Row 0 = test function of vertex 0, entry j = trial function of vertex j.
// P1Div{x,y,z}Operator: -u.dx(K) q dx
template < int K >
struct P1DivForm
{
   static void integrateRow0( const std::array< Point3D, 4 >& c, double row[4] )
   {
      double       g[4][3];
      const double V = tetGradients( c, g );
      for ( int j = 0; j < 4; ++j )
         row[j] = -g[j][K] * V / 4.0;
   }
};
// P1DivT{x,y,z}Operator: -v.dx(K) p dx
template < int K >
struct P1DivTForm
{
   static void integrateRow0( const std::array< Point3D, 4 >& c, double row[4] )
   {
      double       g[4][3];
      const double V = tetGradients( c, g );
      for ( int j = 0; j < 4; ++j )
         row[j] = -g[0][K] * V / 4.0;
   }
};
// P1PSPGOperator: -tau grad p . grad q dx, tau = (cell volume)^(2/3) / 12
struct P1PSPGForm
{
   static void integrateRow0( const std::array< Point3D, 4 >& c, double row[4] )
   {
      double       g[4][3];
      const double V   = tetGradients( c, g );
      const double tau = std::pow( V, 2.0 / 3.0 ) / 12.0;
      for ( int j = 0; j < 4; ++j )
         row[j] = -tau * V * ( g[0][0] * g[j][0] + g[0][1] * g[j][1] + g[0][2] * g[j][2] );
   }
};
struct P1MassForm
{
   static void integrateRow0( const std::array< Point3D, 4 >& c, double row[4] )
   {
      double J[3][3];
      for ( int r = 0; r < 3; ++r )
         for ( int k = 0; k < 3; ++k )
            J[r][k] = c[k + 1][r] - c[0][r];
      const double d = std::fabs( det3( J ) ) / 120.0;
      row[0]         = 2.0 * d;
      row[1] = row[2] = row[3] = d;
   }
};
} // namespace forms

// stencil slots in the C-ABI order (include/hyteg_hip.h) and the 24 micro-tetrahedra around an inner micro-vertex
// (src/hyteg/p1functionspace/P1Elements.hpp:93-143; slot numbers instead of stencilDirection names)
namespace stencil {
static const int kOffsets[15][3] = { { 0, 0, -1 }, { 1, 0, -1 }, { -1, 1, -1 }, { 0, 1, -1 }, { 0, -1, 0 },
                                     { 1, -1, 0 }, { -1, 0, 0 }, { 0, 0, 0 },   { 1, 0, 0 },  { -1, 1, 0 },
                                     { 0, 1, 0 },  { 0, -1, 1 }, { 1, -1, 1 },  { -1, 0, 1 }, { 0, 0, 1 } };
enum
{
   BC = 0, BE, BNW, BN, S, SE, W, C, E, NW, N, TS, TSE, TW, TC
};
static const int kMicroTets[24][4] = {
    { C, BC, BE, BN }, { C, S, SE, TS },   { C, W, NW, TW },   { C, N, E, TC },    { C, W, BC, S },    { C, E, SE, BE },
    { C, N, NW, BN },  { C, TS, TC, TW },  { C, BC, BN, BNW }, { C, W, S, TS },    { C, E, SE, TSE },  { C, NW, N, TC },
    { C, BC, S, SE },  { C, W, NW, BNW },  { C, E, BN, N },    { C, TC, TS, TSE }, { C, W, BC, BNW },  { C, E, BE, BN },
    { C, TC, TW, NW }, { C, SE, TS, TSE }, { C, BC, BE, SE },  { C, BN, BNW, NW }, { C, E, TSE, TC },  { C, W, TS, TW } };
// which cell faces a slot's points lie on: edges 0-5, faces 0-3, vertices 0-3
static const int kSlotFaces[14][4] = { { 1, 1, 0, 0 }, { 1, 0, 1, 0 }, { 1, 0, 0, 1 }, { 0, 1, 1, 0 }, { 0, 1, 0, 1 },
                                       { 0, 0, 1, 1 }, { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 },
                                       { 1, 1, 1, 0 }, { 1, 1, 0, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 } };
inline bool directionStaysInCell( int slot, const int* d )
{
   const int* f = kSlotFaces[slot];
   return !( ( f[0] && d[2] < 0 ) || ( f[1] && d[1] < 0 ) || ( f[2] && d[0] < 0 ) || ( f[3] && d[0] + d[1] + d[2] > 0 ) );
}

struct CellStencils
{
   double inner[15];     // stencil at an inner micro-vertex (P1ConstantOperator.cpp:680-693 assembles it at (1,1,1))
   double slots[14][15]; // this cell's share of the stencil at a micro-vertex on edge 0-5 / face 0-3 / vertex 0-3
};

// P1Elements3D::calculateStencilInMacroCell( index, cell, level, form ), P1Elements.hpp:303-380, for all 15 point classes.
// Affine cells: the 24 element matrices do not depend on the micro-vertex, so they are computed once.
template < class Form >
CellStencils assemble( const MacroCell& cell, uint_t level )
{
   const double step = 1.0 / double( int64_t( 1 ) << level );
   Point3D      xs, ys, zs;
   for ( int r = 0; r < 3; ++r )
   {
      xs[r] = ( cell.coords[1][r] - cell.coords[0][r] ) * step;
      ys[r] = ( cell.coords[2][r] - cell.coords[0][r] ) * step;
      zs[r] = ( cell.coords[3][r] - cell.coords[0][r] ) * step;
   }
   double rows[24][4];
   for ( int t = 0; t < 24; ++t )
   {
      std::array< Point3D, 4 > c;
      for ( int v = 0; v < 4; ++v )
      {
         const int* o = kOffsets[kMicroTets[t][v]];
         for ( int r = 0; r < 3; ++r )
            c[v][r] = cell.coords[0][r] + xs[r] * double( 1 + o[0] ) + ys[r] * double( 1 + o[1] ) + zs[r] * double( 1 + o[2] );
      }
      Form::integrateRow0( c, rows[t] );
   }
   CellStencils S{};
   for ( int t = 0; t < 24; ++t )
      for ( int v = 0; v < 4; ++v )
         S.inner[kMicroTets[t][v]] += rows[t][v];
   for ( int s = 0; s < 14; ++s )
      for ( int t = 0; t < 24; ++t )
      {
         bool inside = true;
         for ( int v = 1; v < 4; ++v )
            inside = inside && directionStaysInCell( s, kOffsets[kMicroTets[t][v]] );
         if ( inside )
            for ( int v = 0; v < 4; ++v )
               S.slots[s][kMicroTets[t][v]] += rows[t][v];
      }
   return S;
}
// Tables of the SOR / Gauss-Seidel sweep over the macro-vertices, -edges and -faces around a cell
// (hyteg_hip_p1_sor_shell_cell): total weights over all neighbour cells, sweep orientations = the macro-primitives'
// own orientations (vertex ids ascending, MeshInfo.cpp:37-72), and the cell's partial stencils without the weights
// that the sweep handles itself (`rest`).
struct CellSorTables
{
   double rest[14][15];
   int    edgeVerts[6][2];
   double edgeW[6][3];
   int    faceVerts[4][3];
   double faceW[4][7];
   double vertexW[4];
};
inline int offsetIndex( int dx, int dy, int dz )
{
   for ( int k = 0; k < 15; ++k )
      if ( kOffsets[k][0] == dx && kOffsets[k][1] == dy && kOffsets[k][2] == dz )
         return k;
   throw std::runtime_error( "offsetIndex: not a stencil direction" );
}
static const int kFaceDirs[6][2] = { { -1, 0 }, { 1, 0 }, { 0, -1 }, { 0, 1 }, { 1, -1 }, { -1, 1 } };
static const int kUnit[4][3]     = { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
} // namespace stencil

} // namespace hyteg
