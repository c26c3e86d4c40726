// Host-side element matrices and stencil assembly for the operators whose C-ABI entry points take macro-cell COORDINATES
// instead of weights: the seam of the generated hyteg_operators (apply_macro_3D( dst, src, 12 macro-vertex coordinates,
// micro_edges_per_macro_edge, ... ), e.g. apps/2023-zikeli-mt/MT-apps/operators-used/
// P1ElementwiseDiffusion_cubes_const_float64.hpp:95-130).  Those kernels compute, per micro-cell type, the element matrix of
// the affine micro-cell (constant per type) and scatter elMat * (local source vector) into dst; the arithmetic below yields
// the same matrices (P1: |T| grad l_i . grad l_j, exact for the one-point rules the generator uses; P2: closed form of the
// quadratic Lagrange basis), from which the constant stencils of the macro-cell's point classes are summed.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>

namespace hyteg_hip {
namespace elmat {

// celldof::macrocell::getMicroVerticesFromMicroCell (src/hyteg/volumedofspace/CellDoFIndexing.hpp:155-198), cell types in
// the order of celldof::allCellTypes: WHITE_UP, BLUE_UP, GREEN_UP, WHITE_DOWN, BLUE_DOWN, GREEN_DOWN
static const int kMicroCellVerts[6][4][3] = {
    { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } }, { { 1, 0, 0 }, { 1, 1, 0 }, { 0, 1, 0 }, { 1, 0, 1 } },
    { { 1, 0, 0 }, { 0, 1, 0 }, { 1, 0, 1 }, { 0, 0, 1 } }, { { 1, 1, 0 }, { 1, 1, 1 }, { 0, 1, 1 }, { 1, 0, 1 } },
    { { 1, 0, 1 }, { 0, 1, 1 }, { 0, 0, 1 }, { 0, 1, 0 } }, { { 0, 1, 0 }, { 1, 1, 0 }, { 1, 0, 1 }, { 0, 1, 1 } } };

// the 15 stencil directions in the C-ABI's weight order (std::map< indexing::Index > order: z, then y, then x)
static const int kDirs[15][3] = { { 0, 0, -1 }, { 1, 0, -1 }, { -1, 1, -1 }, { 0, 1, -1 }, { 0, -1, 0 },
                                  { 1, -1, 0 }, { -1, 0, 0 }, { 0, 0, 0 },   { 1, 0, 0 },  { -1, 1, 0 },
                                  { 0, 1, 0 },  { 0, -1, 1 }, { 1, -1, 1 },  { -1, 0, 1 }, { 0, 0, 1 } };
inline int dir_index( int dx, int dy, int dz )
{
   for ( int k = 0; k < 15; ++k )
      if ( kDirs[k][0] == dx && kDirs[k][1] == dy && kDirs[k][2] == dz )
         return k;
   return -1;
}

// barycentric gradients and volume of the tetrahedron c[4][3]
inline double gradients( const double c[4][3], double g[4][3] )
{
   double J[3][3];
   for ( int r = 0; r < 3; ++r )
      for ( int k = 0; k < 3; ++k )
         J[r][k] = c[k + 1][r] - c[0][r];
   const double det = J[0][0] * ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) - J[0][1] * ( J[1][0] * J[2][2] - J[1][2] * J[2][0] ) +
                      J[0][2] * ( J[1][0] * J[2][1] - J[1][1] * J[2][0] );
   double Ji[3][3];
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
      g[1][r] = Ji[0][r], g[2][r] = Ji[1][r], g[3][r] = Ji[2][r];
      g[0][r] = -( Ji[0][r] + Ji[1][r] + Ji[2][r] );
   }
   return std::fabs( det ) / 6.0;
}

// coordinates of the four vertices of the micro-cell of type t at micro-index (0,0,0) of the macro-cell cc[4][3] refined
// n = micro_edges_per_macro_edge times (affine cell: every micro-cell of a type is a translate of this one)
inline void micro_cell_coords( const double cc[4][3], int64_t n, int t, double c[4][3] )
{
   const double h = 1.0 / (double) n;
   for ( int k = 0; k < 4; ++k )
      for ( int r = 0; r < 3; ++r )
         c[k][r] = cc[0][r] + h * ( ( cc[1][r] - cc[0][r] ) * kMicroCellVerts[t][k][0] + ( cc[2][r] - cc[0][r] ) * kMicroCellVerts[t][k][1] +
                                    ( cc[3][r] - cc[0][r] ) * kMicroCellVerts[t][k][2] );
}

// P1 diffusion: A_ij = |T| grad l_i . grad l_j
inline void p1_diffusion( const double c[4][3], double A[4][4] )
{
   double       g[4][3];
   const double V = gradients( c, g );
   for ( int i = 0; i < 4; ++i )
      for ( int j = 0; j < 4; ++j )
         A[i][j] = V * ( g[i][0] * g[j][0] + g[i][1] * g[j][1] + g[i][2] * g[j][2] );
}

// P2 diffusion in FEniCS ordering (4 vertices, then the edges (2,3)(1,3)(1,2)(0,3)(0,2)(0,1)): phi_a = l_a ( 2 l_a - 1 ),
// phi_ab = 4 l_a l_b; grad phi_i = sum_a ( sum_p C[i][a][p] l_p + D[i][a] ) grad l_a with
// int l_p l_q = V ( 1 + delta_pq ) / 20, int l_p = V / 4
inline void p2_diffusion( const double c[4][3], double A[100] )
{
   double       g[4][3];
   const double V = gradients( c, g );
   double       G[4][4];
   for ( int a = 0; a < 4; ++a )
      for ( int b = 0; b < 4; ++b )
         G[a][b] = g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2];
   static const int pairs[6][2] = { { 2, 3 }, { 1, 3 }, { 1, 2 }, { 0, 3 }, { 0, 2 }, { 0, 1 } };
   double           C[10][4][4] = {}, D[10][4] = {};
   for ( int a = 0; a < 4; ++a )
      C[a][a][a] = 4.0, D[a][a] = -1.0;
   for ( int k = 0; k < 6; ++k )
      C[4 + k][pairs[k][0]][pairs[k][1]] = 4.0, C[4 + k][pairs[k][1]][pairs[k][0]] = 4.0;
   for ( int i = 0; i < 10; ++i )
      for ( int j = 0; j < 10; ++j )
      {
         double s = 0.0;
         for ( int a = 0; a < 4; ++a )
            for ( int b = 0; b < 4; ++b )
            {
               double t = D[i][a] * D[j][b] * V;
               for ( int p = 0; p < 4; ++p )
               {
                  t += ( C[i][a][p] * D[j][b] + D[i][a] * C[j][b][p] ) * V / 4.0;
                  for ( int q = 0; q < 4; ++q )
                     t += C[i][a][p] * C[j][b][q] * V * ( p == q ? 2.0 : 1.0 ) / 20.0;
               }
               s += t * G[a][b];
            }
         A[10 * i + j] = s;
      }
}

// The constant stencils the scatter over micro-cells implies on one affine macro-cell: at an inner micro-vertex (inner[15])
// and, for a micro-vertex on macro-edge 0-5 / macro-face 0-3 / macro-vertex 0-3 (slots 0-13, the C-ABI's point classes), this
// cell's share, i.e. the sum over the adjacent micro-cells that lie inside the macro-cell.  Which micro-cells those are
// depends on the class only; they are read off a representative point of a width-9 cell.
struct P1Stencils
{
   double inner[15];
   double slots[14][15];
};
inline P1Stencils p1_stencils_from_element_matrices( const double A[6][4][4] )
{
   constexpr int NW = 9; // width of the representative cell (level 3)
   static const int rep[15][3] = { { 4, 0, 0 }, { 0, 4, 0 }, { 4, 4, 0 }, { 0, 0, 4 }, { 4, 0, 4 }, { 0, 4, 4 },           // edges 0-5
                                   { 2, 2, 0 }, { 2, 0, 2 }, { 0, 2, 2 }, { 2, 2, 4 },                                     // faces 0-3
                                   { 0, 0, 0 }, { 8, 0, 0 }, { 0, 8, 0 }, { 0, 0, 8 },                                     // vertices 0-3
                                   { 2, 2, 2 } };                                                                         // inner
   P1Stencils S{};
   for ( int cls = 0; cls < 15; ++cls )
   {
      double* w = cls == 14 ? S.inner : S.slots[cls];
      const int* p = rep[cls];
      for ( int t = 0; t < 6; ++t )
         for ( int k = 0; k < 4; ++k )
         {
            // the micro-cell of type t that has p as its local vertex k
            const int m[3] = { p[0] - kMicroCellVerts[t][k][0], p[1] - kMicroCellVerts[t][k][1], p[2] - kMicroCellVerts[t][k][2] };
            bool      inside = true;
            for ( int j = 0; j < 4; ++j )
            {
               const int v[3] = { m[0] + kMicroCellVerts[t][j][0], m[1] + kMicroCellVerts[t][j][1], m[2] + kMicroCellVerts[t][j][2] };
               inside = inside && v[0] >= 0 && v[1] >= 0 && v[2] >= 0 && v[0] + v[1] + v[2] <= NW - 1;
            }
            if ( !inside )
               continue;
            for ( int j = 0; j < 4; ++j )
            {
               const int d = dir_index( m[0] + kMicroCellVerts[t][j][0] - p[0], m[1] + kMicroCellVerts[t][j][1] - p[1],
                                        m[2] + kMicroCellVerts[t][j][2] - p[2] );
               w[d] += A[t][k][j];
            }
         }
   }
   return S;
}

inline P1Stencils p1_diffusion_stencils( const double cc[4][3], int64_t micro_edges_per_macro_edge )
{
   double A[6][4][4];
   for ( int t = 0; t < 6; ++t )
   {
      double c[4][3];
      micro_cell_coords( cc, micro_edges_per_macro_edge, t, c );
      p1_diffusion( c, A[t] );
   }
   return p1_stencils_from_element_matrices( A );
}

} // namespace elmat
} // namespace hyteg_hip
