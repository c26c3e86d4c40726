// p2elements.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// Stencil assembly of the constant-stencil P2 operator, the reference's way: P2Elements::P2Elements3D
// (src/hyteg/p2functionspace/P2Elements3D.hpp:185-420).  For a micro-vertex (vertex DoF) or a micro-edge (edge DoF) of a
// macro-cell the functions walk over the micro-cells around it that lie inside the macro-cell -- the elements of
// P1Elements3D::allCellsAtInnerVertex (P1Elements.hpp:93-143; for an edge: those at its first end point that also contain the
// second one) --, integrate the P2 form on each and add one entry of the local matrix per (centre DoF, leaf DoF) pair to a map
// keyed by the leaf's index offset.  "Also works for indices on the boundary of a macro-cell. In this case the stencil map
// simply contains less elements" (P2Elements3D.hpp:176): there the maps hold this cell's share.
// Index conventions: edgedof::calcEdgeDoFOrientation / calcEdgeDoFIndex / calcNeighboringVertexDoFIndices
// (src/hyteg/edgedofspace/EdgeDoFIndexing.hpp:89-210); local DoF of a vertex pair: fenics::P2DoFMap (src/hyteg/fenics/fenics.hpp:120-124).
#pragma once

#include <array>
#include <map>
#include <vector>

#include "forms.hpp"

namespace hyteg {
namespace P2Elements {
namespace P2Elements3D {

using Idx = std::array< int, 3 >;
inline Idx operator+( const Idx& a, const Idx& b ) { return { a[0] + b[0], a[1] + b[1], a[2] + b[2] }; }
inline Idx operator-( const Idx& a, const Idx& b ) { return { a[0] - b[0], a[1] - b[1], a[2] - b[2] }; }

// kinds: 0 = vertex DoFs, 1..7 = edge DoFs of orientation X, Y, Z, XY, XZ, YZ, XYZ (EdgeDoFOrientation.hpp:29-39, + 1)
enum EdgeDoFOrientation
{
   X = 0, Y, Z, XY, XZ, YZ, XYZ
};
// a stencil leaf: (source kind, index offset from the centre DoF); ordered like the reference's nested maps
// (orientation, then indexing::Index: z, y, x)
struct Key
{
   int  kind;
   Idx  off;
   bool operator<( const Key& o ) const
   {
      if ( kind != o.kind )
         return kind < o.kind;
      if ( off[2] != o.off[2] )
         return off[2] < o.off[2];
      if ( off[1] != o.off[1] )
         return off[1] < o.off[1];
      return off[0] < o.off[0];
   }
};
using StencilMap   = std::map< Key, double >;
using KindStencils = std::array< StencilMap, 8 >; // by destination kind: [0] = v2v + e2v leaves, [1 + o] = v2e + e2e leaves of orientation o

inline EdgeDoFOrientation calcEdgeDoFOrientation( const Idx& a, const Idx& b )
{
   const int d0 = std::abs( b[0] - a[0] ), d1 = std::abs( b[1] - a[1] ), d2 = std::abs( b[2] - a[2] );
   if ( d0 && !d1 && !d2 ) return X;
   if ( !d0 && d1 && !d2 ) return Y;
   if ( !d0 && !d1 && d2 ) return Z;
   if ( d0 && d1 && !d2 ) return XY;
   if ( d0 && !d1 && d2 ) return XZ;
   if ( !d0 && d1 && d2 ) return YZ;
   return XYZ;
}
inline Idx calcEdgeDoFIndex( const Idx& a, const Idx& b )
{
   auto lower = [&]( int axis ) { return a[axis] < b[axis] ? a : b; };
   switch ( calcEdgeDoFOrientation( a, b ) )
   {
   case X: return lower( 0 );
   case Y: return lower( 1 );
   case Z: return lower( 2 );
   case XY: { const Idx l = lower( 0 ); return { l[0], l[1] - 1, l[2] }; }
   case XZ: { const Idx l = lower( 0 ); return { l[0], l[1], l[2] - 1 }; }
   case YZ: { const Idx l = lower( 1 ); return { l[0], l[1], l[2] - 1 }; }
   default: { const Idx l = lower( 0 ); return { l[0], l[1] - 1, l[2] }; }
   }
}
inline std::array< Idx, 2 > calcNeighboringVertexDoFIndices( EdgeDoFOrientation o )
{
   switch ( o )
   {
   case X: return { Idx{ 0, 0, 0 }, Idx{ 1, 0, 0 } };
   case Y: return { Idx{ 0, 0, 0 }, Idx{ 0, 1, 0 } };
   case Z: return { Idx{ 0, 0, 0 }, Idx{ 0, 0, 1 } };
   case XY: return { Idx{ 1, 0, 0 }, Idx{ 0, 1, 0 } };
   case XZ: return { Idx{ 1, 0, 0 }, Idx{ 0, 0, 1 } };
   case YZ: return { Idx{ 0, 1, 0 }, Idx{ 0, 0, 1 } };
   default: return { Idx{ 0, 1, 0 }, Idx{ 1, 0, 1 } };
   }
}
static const int kP2DoFMap[4][4] = { { 0, 9, 8, 7 }, { 9, 1, 6, 5 }, { 8, 6, 2, 4 }, { 7, 5, 4, 3 } };

// the elements at a micro-vertex as index offsets (element[0] = the vertex itself)
using Element = std::array< Idx, 4 >;
inline bool insideCell( const Idx& p, int N ) { return p[0] >= 0 && p[1] >= 0 && p[2] >= 0 && p[0] + p[1] + p[2] <= N - 1; }
// P1Elements3D::getNeighboringElements( microVertexIndex, level ), P1Elements.hpp:215-301: the elements whose vertices lie in the cell
inline std::vector< Element > getNeighboringElements( const Idx& v, int N )
{
   std::vector< Element > out;
   for ( int t = 0; t < 24; ++t )
   {
      Element e;
      bool    in = true;
      for ( int k = 0; k < 4; ++k )
      {
         const int* o = stencil::kOffsets[stencil::kMicroTets[t][k]];
         e[k]         = Idx{ o[0], o[1], o[2] };
         in           = in && insideCell( v + e[k], N );
      }
      if ( in )
         out.push_back( e );
   }
   return out;
}
// edgeWithOrientationFromElement, P2Elements3D.hpp:37-49
inline bool edgeWithOrientationFromElement( const Element& e, EdgeDoFOrientation o, std::array< int, 2 >& edge )
{
   for ( int v0 = 0; v0 < 4; ++v0 )
      for ( int v1 = 0; v1 < v0; ++v1 )
         if ( calcEdgeDoFOrientation( e[v0], e[v1] ) == o )
         {
            edge = { v1, v0 };
            return true;
         }
   return false;
}

// local matrix of the element with vertices `base + e[k]` of the macro-cell refined to `level`
template < class P2Form >
inline void integrateElement( const MacroCell& cell, uint_t level, const Idx& base, const Element& e, double elMat[100] )
{
   const double             step = 1.0 / double( int64_t( 1 ) << level );
   std::array< Point3D, 4 > c;
   for ( int k = 0; k < 4; ++k )
   {
      const Idx p = base + e[k];
      for ( int r = 0; r < 3; ++r )
         c[k][r] = cell.coords[0][r] + step * ( ( cell.coords[1][r] - cell.coords[0][r] ) * p[0] + ( cell.coords[2][r] - cell.coords[0][r] ) * p[1] +
                                                ( cell.coords[3][r] - cell.coords[0][r] ) * p[2] );
   }
   P2Form::integrateAll( c, elMat );
}

// vertex DoF at micro-vertex v: the vertex-to-vertex leaves (P1Elements3D::calculateStencilInMacroCell, P1Elements.hpp:303-380)
// and the edge-to-vertex leaves of every orientation (calculateEdgeToVertexStencilInMacroCell, P2Elements3D.hpp:185-238)
template < class P2Form >
inline StencilMap calculateVertexStencilInMacroCell( const Idx& v, const MacroCell& cell, uint_t level, int N )
{
   StencilMap S;
   for ( const Element& e : getNeighboringElements( v, N ) )
   {
      double M[100];
      integrateElement< P2Form >( cell, level, v, e, M );
      for ( int k = 0; k < 4; ++k )
         S[Key{ 0, e[k] }] += M[10 * kP2DoFMap[0][0] + kP2DoFMap[k][k]];
      for ( int o = X; o <= XYZ; ++o )
      {
         std::array< int, 2 > edge;
         if ( edgeWithOrientationFromElement( e, EdgeDoFOrientation( o ), edge ) )
            S[Key{ 1 + o, calcEdgeDoFIndex( e[edge[0]], e[edge[1]] ) }] += M[10 * kP2DoFMap[0][0] + kP2DoFMap[edge[0]][edge[1]]];
      }
   }
   return S;
}
// edge DoF of orientation `center` at micro-edge index m: the vertex-to-edge leaves (calculateVertexToEdgeStencilInMacroCell,
// P2Elements3D.hpp:256-336) and the edge-to-edge leaves of every orientation (calculateEdgeToEdgeStencilInMacroCell, :359-420)
template < class P2Form >
inline StencilMap calculateEdgeStencilInMacroCell( const Idx& m, EdgeDoFOrientation center, const MacroCell& cell, uint_t level, int N )
{
   StencilMap  S;
   const auto  nv     = calcNeighboringVertexDoFIndices( center );
   const Idx   second = nv[1] - nv[0];
   const Idx   base   = m + nv[0];
   for ( const Element& e : getNeighboringElements( base, N ) )
   {
      bool hasSecond = false;
      for ( int k = 0; k < 4; ++k )
         hasSecond = hasSecond || e[k] == second;
      if ( !hasSecond )
         continue;
      double M[100];
      integrateElement< P2Form >( cell, level, base, e, M );
      std::array< int, 2 > ce;
      edgeWithOrientationFromElement( e, center, ce );
      const int row = kP2DoFMap[ce[0]][ce[1]];
      for ( int k = 0; k < 4; ++k )
         S[Key{ 0, nv[0] + e[k] }] += M[10 * row + kP2DoFMap[k][k]];
      for ( int o = X; o <= XYZ; ++o )
      {
         std::array< int, 2 > le;
         if ( edgeWithOrientationFromElement( e, EdgeDoFOrientation( o ), le ) )
            S[Key{ 1 + o, calcEdgeDoFIndex( nv[0] + e[le[0]], nv[0] + e[le[1]] ) }] += M[10 * row + kP2DoFMap[le[0]][le[1]]];
      }
   }
   return S;
}

// point class (0..13 macro-edge / -face / -vertex slot, 14 inner) of the macro-primitive that contains ALL the given points
inline int classOfPoints( const std::vector< Idx >& pts, int N )
{
   int f[4] = { 1, 1, 1, 1 };
   for ( const Idx& p : pts )
      f[0] &= p[2] == 0, f[1] &= p[1] == 0, f[2] &= p[0] == 0, f[3] &= p[0] + p[1] + p[2] == N - 1;
   const int cnt = f[0] + f[1] + f[2] + f[3];
   if ( cnt == 0 )
      return 14;
   if ( cnt == 1 )
      return 6 + ( f[0] ? 0 : f[1] ? 1 : f[2] ? 2 : 3 );
   if ( cnt == 2 )
   {
      if ( f[0] )
         return f[1] ? 0 : ( f[2] ? 1 : 2 );
      if ( f[1] )
         return f[2] ? 3 : 4;
      return 5;
   }
   if ( f[0] && f[1] && f[2] )
      return 10;
   if ( f[0] && f[1] && f[3] )
      return 11;
   if ( f[0] && f[2] && f[3] )
      return 12;
   return 13;
}

// the stencils of every DoF kind at a DoF of point class `cls` (14 = inner): which micro-cells exist around such a DoF depends
// on the class only (levels >= 2), so the DoF is looked up in a width-9 copy of the cell's index space; the element geometry
// is the cell's at `level` (translates of the six micro-cell types).  A kind without a DoF of that class yields an empty map.
template < class P2Form >
inline KindStencils assembleAtClass( const MacroCell& cell, uint_t level, int cls )
{
   constexpr int N = 9, n = 8;
   KindStencils  out;
   for ( int kind = 0; kind < 8; ++kind )
   {
      const int W     = kind == 0 ? N : ( kind == 1 + XYZ ? n - 1 : n );
      bool      found = false;
      for ( int z = 0; z < W && !found; ++z )
         for ( int y = 0; y < W - z && !found; ++y )
            for ( int x = 0; x < W - z - y && !found; ++x )
            {
               const Idx          p{ x, y, z };
               std::vector< Idx > pts;
               if ( kind == 0 )
                  pts = { p };
               else
               {
                  const auto nv = calcNeighboringVertexDoFIndices( EdgeDoFOrientation( kind - 1 ) );
                  pts           = { p + nv[0], p + nv[1] };
               }
               if ( classOfPoints( pts, N ) != cls )
                  continue;
               found     = true;
               out[kind] = kind == 0 ? calculateVertexStencilInMacroCell< P2Form >( p, cell, level, N ) :
                                       calculateEdgeStencilInMacroCell< P2Form >( p, EdgeDoFOrientation( kind - 1 ), cell, level, N );
            }
   }
   return out;
}

} // namespace P2Elements3D
} // namespace P2Elements
} // namespace hyteg
