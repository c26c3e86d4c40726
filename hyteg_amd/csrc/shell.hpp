// Shell of a macro-cell (the points on its macro-faces / -edges / -vertices): enumeration, the macro-primitive slot of a
// point, and this cell's share of the stencil sum at a shell point.  Shared by p1_boundary.hip (one launch per step) and
// p1_apply_rank.hip (shares, pack and interior of a rank's cell in one launch).
#pragma once

#include "common.hpp"

namespace hyteg_hip {
namespace shell {

struct Slots14x15
{
   double w[14][15];
};

static __constant__ int kOffs[15][3] = { { 0, 0, -1 }, { 1, 0, -1 }, { -1, 1, -1 }, { 0, 1, -1 }, { 0, -1, 0 },
                                  { 1, -1, 0 }, { -1, 0, 0 }, { 0, 0, 0 },   { 1, 0, 0 },  { -1, 1, 0 },
                                  { 0, 1, 0 },  { 0, -1, 1 }, { 1, -1, 1 },  { -1, 0, 1 }, { 0, 0, 1 } };

// slot of the macro-primitive a point lies on (see p1_transfer.hip / MacroCellIndexing.cpp:36-91), -1 interior
__device__ inline int shell_slot( int N, int x, int y, int z )
{
   const int f0 = ( z == 0 ), f1 = ( y == 0 ), f2 = ( x == 0 ), f3 = ( x + y + z == N - 1 );
   const int cnt = f0 + f1 + f2 + f3;
   if ( cnt == 0 )
      return -1;
   if ( cnt == 1 )
      return 6 + ( f0 ? 0 : f1 ? 1 : f2 ? 2 : 3 );
   if ( cnt == 2 )
   {
      if ( f0 )
         return f1 ? 0 : ( f2 ? 1 : 2 );
      if ( f1 )
         return f2 ? 3 : 4;
      return 5;
   }
   if ( f0 && f1 && f2 )
      return 10;
   if ( f0 && f1 && f3 )
      return 11;
   if ( f0 && f2 && f3 )
      return 12;
   return 13;
}

// Enumerates every shell point exactly once: q in [0, 4 tri(N)) -> (x,y,z,slot); returns false for the
// duplicates (a point on an edge/vertex is visited through its lowest-numbered face only) and padding.
__device__ inline bool shell_point( int N, int q, int& x, int& y, int& z, int& slot )
{
   const int T = tri( N );
   if ( q >= 4 * T )
      return false;
   const int f = q / T, r = q - f * T;
   const int j = row_of( N, r );
   const int i = r - row_start( N, j );
   switch ( f )
   {
   case 0:
      x = i, y = j, z = 0;
      break;
   case 1:
      x = i, y = 0, z = j;
      break;
   case 2:
      x = 0, y = i, z = j;
      break;
   default:
      x = i, y = j, z = N - 1 - i - j;
      break;
   }
   const int lowest = ( z == 0 ) ? 0 : ( y == 0 ) ? 1 : ( x == 0 ) ? 2 : 3;
   if ( lowest != f )
      return false;
   slot = shell_slot( N, x, y, z );
   return true;
}

// this cell's share of ( A src ) at shell point (x, y, z) of macro-primitive `slot`: the neighbours inside the cell, in the
// order of the weights.  All fifteen values and weights are loaded before the first FMA (a loop that skips the neighbours
// outside the cell pays one dependent memory round trip per neighbour: 7 us for a launch over one macro-face); a neighbour
// outside the cell reads the point itself and is skipped by a select, so the sum is the same FMA chain as before.
// Neighbour indices by layout algebra from the point's own index (w = length of row 0 of slice z):
//   same slice: row y -> y + 1: + ( w - y ), y -> y - 1: - ( w - y + 1 );  slice z - 1: - tri( w + 1 ) + y';  z + 1: + tri( w ) - y'
// T: value type of the arrays and of the arithmetic (double; float for the float instantiations of the elementwise seam)
template < typename T = double >
__device__ inline T share( const Slots14x15& S, const T* __restrict__ src, int N, int x, int y, int z, int slot )
{
   const int w  = N - z;
   const int i0 = cell_index( N, x, y, z );
   T         v[15], c[15];
   bool      ok[15];
#pragma unroll
   for ( int k = 0; k < 15; ++k )
   {
      const int dx = kOffs[k][0], dy = kOffs[k][1], dz = kOffs[k][2];
      const int nx = x + dx, ny = y + dy, nz = z + dz;
      ok[k]        = !( nx < 0 || ny < 0 || nz < 0 || nx + ny + nz > N - 1 );
      const int rowDelta   = dy == 0 ? 0 : ( dy > 0 ? ( w - y ) : -( w - y + 1 ) );
      const int sliceDelta = dz == 0 ? 0 : ( dz > 0 ? tri( w ) - ny : ny - tri( w + 1 ) );
      v[k]                 = src[ok[k] ? i0 + sliceDelta + rowDelta + dx : i0];
      c[k]                 = (T) S.w[slot][k];
   }
   T acc = (T) 0;
#pragma unroll
   for ( int k = 0; k < 15; ++k )
      acc = ok[k] ? fma( c[k], v[k], acc ) : acc;
   return acc;
}

} // namespace shell
} // namespace hyteg_hip
