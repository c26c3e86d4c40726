// P2 elementwise operator on one macro-cell: dst = alpha * A * src for vertex + edge DoFs, where A is
// given by the 10 x 10 element matrices of the six micro-cell types (constant over an affine macro-cell).
// Reference: P2ElementwiseOperator::gemv / localMatrixVectorMultiply3D (src/hyteg/elementwiseoperators/
// P2ElementwiseOperator.cpp:66-223): a loop over micro-cells that SCATTERS ten sums into the destination arrays.
// Here the operation is a GATHER (no atomics, fixed summation order = the order the reference's scatter loop produces):
// a destination DoF of kind c (vertex, or edge orientation X/Y/Z/XY/XZ/YZ/XYZ) is local DoF k of the micro-cell
// (type t, index dof - offset[t][k]) for a fixed list of (t, k); each such cell contributes row k of its element matrix
// times its ten source values.  The lists and offsets follow from celldof::macrocell::getMicroVerticesFromMicroCell
// (volumedofspace/CellDoFIndexing.hpp:155-198) and edgedof::calcEdgeDoFIndex / calcEdgeDoFOrientation
// (edgedofspace/EdgeDoFIndexing.hpp:89-165); they are built once on the host.
// Kernels in this file, by level: levels 0-1 the table-driven micro-cell gather (p2_elementwise_kernel, one thread group per DoF);
// level 2 and kind-restricted applies below level 6: compile-time stencils, thread per DoF (p2_inner_body, p2_boundary_body) or by
// rows (p2_rows_body_dpp); from level 3: p2_class_rows_kernel -- row waves that compute the inner DoFs and every boundary class
// (round 3, the section "Row kernel with every point class" below).
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <tuple>
#include <vector>

#include <cstdlib>

#include "common.hpp"

using namespace hyteg_hip;

namespace {

constexpr int kThreads = 256;

struct LocalDof
{
   signed char kind; // 0 vertex array, 1..7 edge array block X, Y, Z, XY, XZ, YZ, XYZ
   signed char ox, oy, oz;
};
struct Entry
{
   signed char type, row, ox, oy, oz, pad[3];
};
struct P2Tables
{
   LocalDof    local[6][10];
   Entry       entries[8][24];
   signed char nentries[8];
};

const int kMicroVerts[6][4][3] = { { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } }, { { 1, 0, 0 }, { 1, 1, 0 }, { 0, 1, 0 }, { 1, 0, 1 } },
                                   { { 1, 0, 0 }, { 0, 1, 0 }, { 1, 0, 1 }, { 0, 0, 1 } }, { { 1, 1, 0 }, { 1, 1, 1 }, { 0, 1, 1 }, { 1, 0, 1 } },
                                   { { 1, 0, 1 }, { 0, 1, 1 }, { 0, 0, 1 }, { 0, 1, 0 } }, { { 0, 1, 0 }, { 1, 1, 0 }, { 1, 0, 1 }, { 0, 1, 1 } } };
const int kEdgePairs[6][2]      = { { 2, 3 }, { 1, 3 }, { 1, 2 }, { 0, 3 }, { 0, 2 }, { 0, 1 } };
// logical edge index = (lower end point by the orientation's rule) + shift; orientation from the difference vector
void edge_of( const int* a, const int* b, int& kind, int* e )
{
   const int  d[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
   const int* lo;
   if ( d[1] == 0 && d[2] == 0 )
   {
      kind = 1, lo = a[0] < b[0] ? a : b;
      e[0] = lo[0], e[1] = lo[1], e[2] = lo[2];
   }
   else if ( d[0] == 0 && d[2] == 0 )
   {
      kind = 2, lo = a[1] < b[1] ? a : b;
      e[0] = lo[0], e[1] = lo[1], e[2] = lo[2];
   }
   else if ( d[0] == 0 && d[1] == 0 )
   {
      kind = 3, lo = a[2] < b[2] ? a : b;
      e[0] = lo[0], e[1] = lo[1], e[2] = lo[2];
   }
   else if ( d[2] == 0 )
   {
      kind = 4, lo = a[0] < b[0] ? a : b;
      e[0] = lo[0], e[1] = lo[1] - 1, e[2] = lo[2];
   }
   else if ( d[1] == 0 )
   {
      kind = 5, lo = a[0] < b[0] ? a : b;
      e[0] = lo[0], e[1] = lo[1], e[2] = lo[2] - 1;
   }
   else if ( d[0] == 0 )
   {
      kind = 6, lo = a[1] < b[1] ? a : b;
      e[0] = lo[0], e[1] = lo[1], e[2] = lo[2] - 1;
   }
   else
   {
      kind = 7, lo = a[0] < b[0] ? a : b;
      e[0] = lo[0], e[1] = lo[1] - 1, e[2] = lo[2];
   }
}

const P2Tables& tables()
{
   static P2Tables       T;
   static std::once_flag once;
   std::call_once( once, [] {
      for ( int t = 0; t < 6; ++t )
      {
         for ( int k = 0; k < 4; ++k )
            T.local[t][k] = LocalDof{ 0, (signed char) kMicroVerts[t][k][0], (signed char) kMicroVerts[t][k][1], (signed char) kMicroVerts[t][k][2] };
         for ( int k = 0; k < 6; ++k )
         {
            int kind, e[3];
            edge_of( kMicroVerts[t][kEdgePairs[k][0]], kMicroVerts[t][kEdgePairs[k][1]], kind, e );
            T.local[t][4 + k] = LocalDof{ (signed char) kind, (signed char) e[0], (signed char) e[1], (signed char) e[2] };
         }
      }
      for ( int c = 0; c < 8; ++c )
      {
         int n = 0;
         for ( int t = 0; t < 6; ++t )
         {
            // cells of one type contribute in micro-cell iteration order (z, y, x ascending), i.e. offsets descending
            int first = n;
            for ( int k = 0; k < 10; ++k )
               if ( T.local[t][k].kind == c )
                  T.entries[c][n++] = Entry{ (signed char) t, (signed char) k, T.local[t][k].ox, T.local[t][k].oy, T.local[t][k].oz, { 0, 0, 0 } };
            for ( int a = first; a < n; ++a )
               for ( int b = a + 1; b < n; ++b )
               {
                  const Entry &A = T.entries[c][a], &B = T.entries[c][b];
                  const bool   swap = B.oz > A.oz || ( B.oz == A.oz && ( B.oy > A.oy || ( B.oy == A.oy && B.ox > A.ox ) ) );
                  if ( swap )
                     std::swap( T.entries[c][a], T.entries[c][b] );
               }
         }
         T.nentries[c] = (signed char) n;
      }
   } );
   return T;
}

struct P2Args
{
   double*       dstV;
   double*       dstE;
   const double* srcV;
   const double* srcE;
   const double* elmat; // device, [6][10][10]
   double        alpha;
   int           N, update;
   unsigned      mask;
   unsigned      kinds; // destination kinds to compute: bit 0 vertex DoFs, 1..7 edge DoFs X, Y, Z, XY, XZ, YZ, XYZ
   P2Tables      T;
};

__device__ inline int class_from_flags( int f0, int f1, int f2, int f3 )
{
   const int cnt = f0 + f1 + f2 + f3;
   if ( cnt == 0 )
      return 14;
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

// slice z of entry i of a tetrahedral array of width W (largest z with slice_start(W,z) <= i): cube-root estimate + fix-up
// (a binary search with 64-bit products was a quarter of the instructions of the inner kernels)
__device__ inline int slice_of( int W, int64_t i )
{
   const int64_t rest = tet64( W ) - i; // entries from i to the end: tet(W - z) >= rest > tet(W - z - 1)
   int           m    = (int) cbrtf( 6.0f * (float) rest );
   m                  = m < 1 ? 1 : ( m > W ? W : m );
   while ( m > 1 && tet64( m - 1 ) >= rest )
      --m;
   while ( tet64( m ) < rest )
      ++m;
   return W - m;
}

// end points of an edge DoF relative to its logical index, by orientation X, Y, Z, XY, XZ, YZ, XYZ
__constant__ int kEdgeEnds[7][2][3] = { { { 0, 0, 0 }, { 1, 0, 0 } }, { { 0, 0, 0 }, { 0, 1, 0 } }, { { 0, 0, 0 }, { 0, 0, 1 } },
                                        { { 1, 0, 0 }, { 0, 1, 0 } }, { { 1, 0, 0 }, { 0, 0, 1 } }, { { 0, 1, 0 }, { 0, 0, 1 } },
                                        { { 0, 1, 0 }, { 1, 0, 1 } } };
__constant__ int kRowDeficit[6]     = { 0, 1, 1, 2, 1, 1 }; // numCellsPerRowByType: n - deficit

__device__ inline int64_t edge_block_start( int n, int kind ) { return (int64_t) ( kind - 1 ) * tet64( n ); }

// Boundary DoFs only, densely enumerated: the non-inner DoFs of a kind lie on faces of that kind's own tetrahedral array
// (all four for vertex DoFs, two for X .. YZ edge DoFs, none for XYZ), so thread q walks the four triangular faces
// (q -> face, (i,j)) and keeps a point at its lowest-numbered face.
template < int G >
__global__ __launch_bounds__( kThreads ) void p2_elementwise_kernel( const P2Args A )
{
   const int c = blockIdx.y; // destination kind
   const int N = A.N, n = N - 1;
   const int W = c == 0 ? N : ( c == 7 ? n - 1 : n );
   if ( W <= 0 || !( ( A.kinds >> c ) & 1u ) )
      return;
   // G lanes per DoF share its (at most 24) adjacent micro-cells, 24 / G each.  One thread per DoF walks 24 dependent memory
   // round trips: pure latency (39 us at level 5, 47 us at level 7); 32 lanes per DoF repeat the decode 32 times
   // (13 us at level 5 but 115 us at level 7); G = 8 keeps three round trips and a 8-fold decode.
   constexpr bool LANES = G > 1;
   const int      T     = tri( W );
   const int      lane  = threadIdx.x & ( G - 1 );
   const int      q     = blockIdx.x * ( kThreads / G ) + (int) threadIdx.x / G;
   if ( q >= 4 * T )
      return;
   int x, y, z;
   {
      const int f = q / T, r = q - f * T;
      const int j = row_of( W, r );
      const int k = r - row_start( W, j );
      switch ( f )
      {
      case 0:
         x = k, y = j, z = 0;
         break;
      case 1:
         x = k, y = 0, z = j;
         break;
      case 2:
         x = 0, y = k, z = j;
         break;
      default:
         x = k, y = j, z = W - 1 - k - j;
         break;
      }
      const int lowest = ( z == 0 ) ? 0 : ( y == 0 ) ? 1 : ( x == 0 ) ? 2 : 3;
      if ( lowest != f )
         return;
   }
   const int64_t i = (int64_t) cell_index( W, x, y, z );
   int           cls;
   if ( c == 0 )
      cls = class_from_flags( z == 0, y == 0, x == 0, x + y + z == N - 1 );
   else
   {
      int f0 = 1, f1 = 1, f2 = 1, f3 = 1;
#pragma unroll
      for ( int e = 0; e < 2; ++e )
      {
         const int px = x + kEdgeEnds[c - 1][e][0], py = y + kEdgeEnds[c - 1][e][1], pz = z + kEdgeEnds[c - 1][e][2];
         f0 &= pz == 0, f1 &= py == 0, f2 &= px == 0, f3 &= px + py + pz == N - 1;
      }
      cls = class_from_flags( f0, f1, f2, f3 );
   }
   if ( !( ( A.mask >> cls ) & 1u ) )
      return;
   // one adjacent micro-cell: alpha * (row of its element matrix) . (its ten source values)
   auto contribution = [&]( int l, double& part ) -> bool {
      const Entry en = A.T.entries[c][l];
      const int   t = en.type, mx = x - en.ox, my = y - en.oy, mz = z - en.oz;
      const int   rows = n - kRowDeficit[t];
      if ( mx < 0 || my < 0 || mz < 0 || mx + my + mz > rows - 1 )
         return false;
      const double* M = A.elmat + 100 * t + 10 * en.row;
      double        s = 0.0;
#pragma unroll
      for ( int k = 0; k < 10; ++k )
      {
         const LocalDof ld = A.T.local[t][k];
         const int      px = mx + ld.ox, py = my + ld.oy, pz = mz + ld.oz;
         const double   v  = ld.kind == 0 ? A.srcV[(int64_t) cell_index( N, px, py, pz )] :
                                            A.srcE[edge_block_start( n, ld.kind ) + cell_index( ld.kind == 7 ? n - 1 : n, px, py, pz )];
         s                 = s + M[k] * v;
      }
      part = A.alpha * s;
      return true;
   };
   double acc = 0.0;
   if constexpr ( LANES )
   {
      // lane l evaluates micro-cells l, l + G, l + 2G, ...; every lane then adds all contributions in the reference's loop order
      constexpr int      kSlots = 24 / G;
      double             part[kSlots];
      unsigned long long vmask[kSlots];
#pragma unroll
      for ( int k = 0; k < kSlots; ++k )
      {
         part[k]          = 0.0;
         const int  l     = lane + k * G;
         const bool valid = l < A.T.nentries[c] && contribution( l, part[k] );
         vmask[k]         = __ballot( valid );
      }
      const int base = ( threadIdx.x & 63 ) & ~( G - 1 ); // first lane of this DoF's group inside the wave
#pragma unroll
      for ( int l = 0; l < 24; ++l )
      {
         const double p = __shfl( part[l / G], base + ( l % G ), 64 );
         if ( ( vmask[l / G] >> ( base + ( l % G ) ) ) & 1ull )
            acc += p;
      }
   }
   else
   {
      for ( int l = 0; l < A.T.nentries[c]; ++l )
      {
         double part;
         if ( contribution( l, part ) )
            acc += part;
      }
   }
   if ( lane != 0 )
      return;
   double* out = c == 0 ? A.dstV + i : A.dstE + edge_block_start( n, c ) + i;
   *out        = A.update == HYTEG_HIP_ADD ? *out + acc : acc;
}

// ---- vector operations and dot product on the edge-DoF array (EdgeDoFFunction::assign / add / dotLocal on a macro-cell,
// src/hyteg/edgedofspace/EdgeDoFFunction.cpp; generic loops in EdgeDoFMacroCell.hpp), masked by point class ----
__device__ inline bool edge_entry( int n, int64_t i, int& x, int& y, int& z, int& o )
{
   const int64_t blk = tet64( n );
   o                 = (int) ( i / blk );
   if ( o > 6 )
      return false;
   const int     W = o == 6 ? n - 1 : n;
   const int64_t r = i - (int64_t) o * blk;
   if ( W <= 0 || r >= tet64( W ) )
      return false;
   z           = slice_of( W, r );
   const int j = (int) ( r - ( tet64( W ) - tet64( W - z ) ) );
   y           = row_of( W - z, j );
   x           = j - row_start( W - z, y );
   return true;
}
__device__ inline int edge_class( int N, int x, int y, int z, int o )
{
   int f0 = 1, f1 = 1, f2 = 1, f3 = 1;
#pragma unroll
   for ( int e = 0; e < 2; ++e )
   {
      const int px = x + kEdgeEnds[o][e][0], py = y + kEdgeEnds[o][e][1], pz = z + kEdgeEnds[o][e][2];
      f0 &= pz == 0, f1 &= py == 0, f2 &= px == 0, f3 &= px + py + pz == N - 1;
   }
   return class_from_flags( f0, f1, f2, f3 );
}

struct EdgeVecArgs
{
   double*       dst;
   const double* src[HYTEG_HIP_MAX_SRCS];
   double        c[HYTEG_HIP_MAX_SRCS];
   int64_t       size;
   int           N, nsrc, op; // 0 assign, 1 add, 2 mult, 3 set constant c[0]
   unsigned      mask;
   unsigned      kinds; // bit k (1..7): edge DoFs of orientation k - 1 take part
};
__global__ __launch_bounds__( kThreads ) void p2_edge_vector_kernel( const EdgeVecArgs A )
{
   const int64_t i = (int64_t) blockIdx.x * kThreads + threadIdx.x;
   int           x, y, z, o;
   if ( i >= A.size || !edge_entry( A.N - 1, i, x, y, z, o ) || !( ( A.kinds >> ( o + 1 ) ) & 1u ) ||
        !( ( A.mask >> edge_class( A.N, x, y, z, o ) ) & 1u ) )
      return;
   double tmp;
   if ( A.op == 3 )
      tmp = A.c[0];
   else if ( A.op == 2 )
   {
      tmp = A.src[0][i];
      for ( int k = 1; k < A.nsrc; ++k )
         tmp *= A.src[k][i];
   }
   else
   {
      tmp = A.c[0] * A.src[0][i];
      for ( int k = 1; k < A.nsrc; ++k )
         tmp += A.c[k] * A.src[k][i];
      if ( A.op == 1 )
         tmp = A.dst[i] + tmp;
   }
   A.dst[i] = tmp;
}

// the same for up to HYTEG_HIP_MAX_BATCH macro-cells in one launch (blockIdx.y = cell): at the small levels of a multigrid cycle a
// launch per (cell, operation) is pure launch latency -- a Taylor-Hood V(3,3) cycle on 24 cells issued 54,000 of them (round 3)
struct EdgeVecBatchArgs
{
   double*       dst[HYTEG_HIP_MAX_BATCH];
   const double* src[HYTEG_HIP_MAX_SRCS][HYTEG_HIP_MAX_BATCH];
   unsigned      mask[HYTEG_HIP_MAX_BATCH];
   double        c[HYTEG_HIP_MAX_SRCS];
   int64_t       size;
   int           N, nsrc, op;
   unsigned      kinds;
};
__global__ __launch_bounds__( kThreads ) void p2_edge_vector_batch_kernel( const EdgeVecBatchArgs A )
{
   const int      cell = blockIdx.y;
   const unsigned mask = A.mask[cell];
   const int64_t  i    = (int64_t) blockIdx.x * kThreads + threadIdx.x;
   int            x, y, z, o;
   if ( mask == 0 || i >= A.size || !edge_entry( A.N - 1, i, x, y, z, o ) || !( ( A.kinds >> ( o + 1 ) ) & 1u ) ||
        !( ( mask >> edge_class( A.N, x, y, z, o ) ) & 1u ) )
      return;
   double* dst = A.dst[cell];
   double  tmp;
   if ( A.op == 3 )
      tmp = A.c[0];
   else if ( A.op == 2 )
   {
      tmp = A.src[0][cell][i];
      for ( int k = 1; k < A.nsrc; ++k )
         tmp *= A.src[k][cell][i];
   }
   else
   {
      tmp = A.c[0] * A.src[0][cell][i];
      for ( int k = 1; k < A.nsrc; ++k )
         tmp += A.c[k] * A.src[k][cell][i];
      if ( A.op == 1 )
         tmp = dst[i] + tmp;
   }
   dst[i] = tmp;
}

constexpr int kEdgeDotBlocks = 1024;
__global__ __launch_bounds__( kThreads ) void p2_edge_dot_kernel( const double* __restrict__ a, const double* __restrict__ b, int64_t size, int N,
                                                                   unsigned mask, double* partial )
{
   __shared__ double sh[kThreads / 64];
   double            acc = 0.0;
   // fixed entry -> thread assignment: deterministic
   for ( int64_t i = (int64_t) blockIdx.x * kThreads + threadIdx.x; i < size; i += (int64_t) gridDim.x * kThreads )
   {
      int x, y, z, o;
      if ( edge_entry( N - 1, i, x, y, z, o ) && ( ( mask >> edge_class( N, x, y, z, o ) ) & 1u ) )
         acc = fma( a[i], b[i], acc );
   }
#pragma unroll
   for ( int off = 32; off > 0; off >>= 1 )
      acc += __shfl_down( acc, off, 64 );
   if ( ( threadIdx.x & 63 ) == 0 )
      sh[threadIdx.x >> 6] = acc;
   __syncthreads();
   if ( threadIdx.x == 0 )
   {
      double r = 0.0;
      for ( int k = 0; k < kThreads / 64; ++k )
         r += sh[k];
      partial[blockIdx.x] = r;
   }
}
// the masked dot product of up to HYTEG_HIP_MAX_BATCH macro-cells in one launch: one workgroup per cell walks the cell's edge-DoF array
// in a fixed order (deterministic); for the small levels of a cycle, where two launches per cell and dot product were 40 % of a
// Taylor-Hood cycle's kernel time (round 3)
struct EdgeDotBatchArgs
{
   const double* a[HYTEG_HIP_MAX_BATCH];
   const double* b[HYTEG_HIP_MAX_BATCH];
   unsigned      mask[HYTEG_HIP_MAX_BATCH];
   int64_t       size;
   int           N;
   double*       result; // [ncells]
};
__global__ __launch_bounds__( kThreads ) void p2_edge_dot_batch_kernel( const EdgeDotBatchArgs A )
{
   __shared__ double sh[kThreads / 64];
   const int         cell = blockIdx.x;
   const unsigned    mask = A.mask[cell];
   const double*     a    = A.a[cell];
   const double*     b    = A.b[cell];
   double            acc  = 0.0;
   if ( mask != 0 )
      for ( int64_t i = threadIdx.x; i < A.size; i += kThreads )
      {
         int x, y, z, o;
         if ( edge_entry( A.N - 1, i, x, y, z, o ) && ( ( mask >> edge_class( A.N, x, y, z, o ) ) & 1u ) )
            acc = fma( a[i], b[i], acc );
      }
#pragma unroll
   for ( int off = 32; off > 0; off >>= 1 )
      acc += __shfl_down( acc, off, 64 );
   if ( ( threadIdx.x & 63 ) == 0 )
      sh[threadIdx.x >> 6] = acc;
   __syncthreads();
   if ( threadIdx.x == 0 )
   {
      double r = 0.0;
      for ( int k = 0; k < kThreads / 64; ++k )
         r += sh[k];
      A.result[cell] = r;
   }
}
__global__ __launch_bounds__( kThreads ) void p2_sum_partials_kernel( const double* partial, int n, double* result )
{
   __shared__ double sh[kThreads / 64];
   double            acc = 0.0;
   for ( int k = threadIdx.x; k < n; k += kThreads )
      acc += partial[k];
#pragma unroll
   for ( int off = 32; off > 0; off >>= 1 )
      acc += __shfl_down( acc, off, 64 );
   if ( ( threadIdx.x & 63 ) == 0 )
      sh[threadIdx.x >> 6] = acc;
   __syncthreads();
   if ( threadIdx.x == 0 )
   {
      double r = 0.0;
      for ( int k = 0; k < kThreads / 64; ++k )
         r += sh[k];
      *result = r;
   }
}


// =====================================================================================================================
// Fast path for INNER DoFs: on an affine macro-cell every inner DoF of one kind sees the same neighbourhood, so the sum
// over its adjacent micro-cells collapses to a constant stencil  sum_q w[q] * src_{kind_q}( dof + d_q )  (what the reference's
// P2ConstantOperator assembles into its vertex-to-vertex, edge-to-vertex, vertex-to-edge and edge-to-edge stencils).  The list
// of (source kind, offset) pairs per destination kind is a geometric fact and is built at COMPILE time from the micro-cell
// tables, so the kernel is straight-line code with constant offsets; the weights are summed from the element matrices on
// the host (hyteg_hip_p2_build_operator_table) and read through scalar loads.
// =====================================================================================================================
struct CLocal
{
   int kind, ox, oy, oz;
};
constexpr int cMicroVerts[6][4][3] = { { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } }, { { 1, 0, 0 }, { 1, 1, 0 }, { 0, 1, 0 }, { 1, 0, 1 } },
                                       { { 1, 0, 0 }, { 0, 1, 0 }, { 1, 0, 1 }, { 0, 0, 1 } }, { { 1, 1, 0 }, { 1, 1, 1 }, { 0, 1, 1 }, { 1, 0, 1 } },
                                       { { 1, 0, 1 }, { 0, 1, 1 }, { 0, 0, 1 }, { 0, 1, 0 } }, { { 0, 1, 0 }, { 1, 1, 0 }, { 1, 0, 1 }, { 0, 1, 1 } } };
constexpr int cEdgePairs[6][2]      = { { 2, 3 }, { 1, 3 }, { 1, 2 }, { 0, 3 }, { 0, 2 }, { 0, 1 } };

constexpr CLocal c_local( int t, int k )
{
   if ( k < 4 )
      return CLocal{ 0, cMicroVerts[t][k][0], cMicroVerts[t][k][1], cMicroVerts[t][k][2] };
   const int* a = cMicroVerts[t][cEdgePairs[k - 4][0]];
   const int* b = cMicroVerts[t][cEdgePairs[k - 4][1]];
   const int  d0 = b[0] - a[0], d1 = b[1] - a[1], d2 = b[2] - a[2];
   if ( d1 == 0 && d2 == 0 )
   {
      const int* lo = a[0] < b[0] ? a : b;
      return CLocal{ 1, lo[0], lo[1], lo[2] };
   }
   if ( d0 == 0 && d2 == 0 )
   {
      const int* lo = a[1] < b[1] ? a : b;
      return CLocal{ 2, lo[0], lo[1], lo[2] };
   }
   if ( d0 == 0 && d1 == 0 )
   {
      const int* lo = a[2] < b[2] ? a : b;
      return CLocal{ 3, lo[0], lo[1], lo[2] };
   }
   if ( d2 == 0 )
   {
      const int* lo = a[0] < b[0] ? a : b;
      return CLocal{ 4, lo[0], lo[1] - 1, lo[2] };
   }
   if ( d1 == 0 )
   {
      const int* lo = a[0] < b[0] ? a : b;
      return CLocal{ 5, lo[0], lo[1], lo[2] - 1 };
   }
   if ( d0 == 0 )
   {
      const int* lo = a[1] < b[1] ? a : b;
      return CLocal{ 6, lo[0], lo[1], lo[2] - 1 };
   }
   const int* lo = a[0] < b[0] ? a : b;
   return CLocal{ 7, lo[0], lo[1] - 1, lo[2] };
}

constexpr int kMaxStencil = 96;
struct KindStencil
{
   int n;
   int kind[kMaxStencil], dx[kMaxStencil], dy[kMaxStencil], dz[kMaxStencil];
};
// unique (source kind, offset) pairs of destination kind c, in first-seen order over (type, local row, local column)
constexpr KindStencil build_kind_stencil( int c )
{
   KindStencil S{};
   for ( int t = 0; t < 6; ++t )
      for ( int k = 0; k < 10; ++k )
      {
         const CLocal row = c_local( t, k );
         if ( row.kind != c )
            continue;
         for ( int j = 0; j < 10; ++j )
         {
            const CLocal col = c_local( t, j );
            const int    dx = col.ox - row.ox, dy = col.oy - row.oy, dz = col.oz - row.oz;
            bool         found = false;
            for ( int q = 0; q < S.n; ++q )
               found = found || ( S.kind[q] == col.kind && S.dx[q] == dx && S.dy[q] == dy && S.dz[q] == dz );
            if ( !found )
            {
               S.kind[S.n] = col.kind, S.dx[S.n] = dx, S.dy[S.n] = dy, S.dz[S.n] = dz;
               ++S.n;
            }
         }
      }
   return S;
}
template < int C >
struct KindStencilOf
{
   static constexpr KindStencil value = build_kind_stencil( C );
};
constexpr int stencil_count( int c ) { return build_kind_stencil( c ).n; }
constexpr int stencil_offset( int c )
{
   int o = 600; // the element matrices come first in the operator table
   for ( int k = 0; k < c; ++k )
      o += stencil_count( k );
   return o;
}
// after the inner stencils: per destination kind, 14 boundary point classes x the same entry list (weights of neighbours
// whose micro-cells do not exist for that class are exactly zero)
constexpr int class_offset( int c )
{
   int o = stencil_offset( 8 );
   for ( int k = 0; k < c; ++k )
      o += 14 * stencil_count( k );
   return o;
}
constexpr int kOperatorTableSize = class_offset( 8 );

struct P2FastArgs
{
   double*       dstV;
   double*       dstE;
   const double* srcV;
   const double* srcE;
   const double* table; // device: [600 element matrices | stencil weights of kind 0 | kind 1 | ... ]
   double        alpha;
   int           N, update;
   unsigned      kinds; // destination kinds to compute (bit per kind), as in P2Args
};

// Row bases: every stencil entry of destination kind C reads source kind K at (x + dx, y + dy, z + dz) with compile-time
// (K, dx, dy, dz); the array index of (x, y + dy, z + dz) in kind K's block is computed once per USED (K, dy, dz) and the
// entries add dx.  32-bit index arithmetic: the largest index at level 9 is 6 tet(512) + tet(511) < 2^31.
template < int C >
constexpr bool row_used( int K, int dy, int dz )
{
   constexpr KindStencil S = KindStencilOf< C >::value;
   for ( int q = 0; q < S.n; ++q )
      if ( S.kind[q] == K && S.dy[q] == dy && S.dz[q] == dz )
         return true;
   return false;
}
struct RowBases
{
   int b[8][3][3]; // [source kind][dy + 1][dz + 1]
};
// index of (x, y + DY, z + DZ) from the index i0 of (x, y, z) in a tetrahedral array whose slice z has first-row length Wz:
// (x,y,z) -> (x,y+1,z): + (Wz - y);  (x,y,z) -> (x,y,z+1): + tri(Wz) - y  (the layout algebra of the P1 kernels)
template < int DY, int DZ >
__device__ inline int p2_neighbour_row( int i0, int Wz, int y )
{
   int i = i0, w = Wz;
   if constexpr ( DZ == 1 )
   {
      i += tri( w ) - y;
      w -= 1;
   }
   else if constexpr ( DZ == -1 )
   {
      i -= tri( w + 1 ) - y;
      w += 1;
   }
   if constexpr ( DY == 1 )
      i += w - y;
   else if constexpr ( DY == -1 )
      i -= w - y + 1;
   return i;
}
template < int C, int K, int DY, int DZ >
__device__ inline void p2_row_base( RowBases& R, int i0, int Wz, int y )
{
   if constexpr ( row_used< C >( K, DY, DZ ) )
      R.b[K][DY + 1][DZ + 1] = p2_neighbour_row< DY, DZ >( i0, Wz, y );
}
template < int C, int K >
constexpr bool kind_used()
{
   for ( int dy = -1; dy <= 1; ++dy )
      for ( int dz = -1; dz <= 1; ++dz )
         if ( row_used< C >( K, dy, dz ) )
            return true;
   return false;
}
template < int C, int K >
__device__ inline void p2_row_bases_of_kind( RowBases& R, int N, int n, int x, int y, int z )
{
   if constexpr ( kind_used< C, K >() )
   {
      const int W  = K == 0 ? N : ( K == 7 ? n - 1 : n );
      const int i0 = ( K == 0 ? 0 : ( K - 1 ) * (int) tet32( (unsigned) n ) ) + cell_index( W, x, y, z );
      const int Wz = W - z;
      p2_row_base< C, K, -1, -1 >( R, i0, Wz, y );
      p2_row_base< C, K, 0, -1 >( R, i0, Wz, y );
      p2_row_base< C, K, 1, -1 >( R, i0, Wz, y );
      p2_row_base< C, K, -1, 0 >( R, i0, Wz, y );
      p2_row_base< C, K, 0, 0 >( R, i0, Wz, y );
      p2_row_base< C, K, 1, 0 >( R, i0, Wz, y );
      p2_row_base< C, K, -1, 1 >( R, i0, Wz, y );
      p2_row_base< C, K, 0, 1 >( R, i0, Wz, y );
      p2_row_base< C, K, 1, 1 >( R, i0, Wz, y );
   }
}

template < int C, int Q >
__device__ inline void p2_term( const P2FastArgs& A, const double* __restrict__ w, const RowBases& R, double& acc )
{
   constexpr int K = KindStencilOf< C >::value.kind[Q], DX = KindStencilOf< C >::value.dx[Q], DY = KindStencilOf< C >::value.dy[Q],
                 DZ = KindStencilOf< C >::value.dz[Q];
   static_assert( DY >= -1 && DY <= 1 && DZ >= -1 && DZ <= 1, "stencil offsets" );
   const int idx = R.b[K][DY + 1][DZ + 1] + DX;
   acc           = fma( w[Q], K == 0 ? A.srcV[idx] : A.srcE[idx], acc );
}

// inner DoFs of kind C: inner vertex DoFs x,y,z >= 1, x+y+z <= N-2; inner edge DoFs by EdgeDoFIndexing.hpp:987-1020
template < int C >
__device__ inline bool p2_inner( int N, int x, int y, int z )
{
   const int n = N - 1, s = x + y + z;
   if constexpr ( C == 0 )
      return x >= 1 && y >= 1 && z >= 1 && s <= N - 2;
   else if constexpr ( C == 1 )
      return y > 0 && z > 0 && s < n;
   else if constexpr ( C == 2 )
      return x > 0 && z > 0 && s < n;
   else if constexpr ( C == 3 )
      return x > 0 && y > 0 && s < n;
   else if constexpr ( C == 4 )
      return z > 0 && s < n - 1;
   else if constexpr ( C == 5 )
      return y > 0 && s < n - 1;
   else if constexpr ( C == 6 )
      return x > 0 && s < n - 1;
   else
      return s < n - 1;
}

template < int C >
__device__ inline void p2_inner_body( const P2FastArgs& A )
{
   constexpr int NQ  = KindStencilOf< C >::value.n; // forced constant evaluation: none of the table code may run on the device
   constexpr int OFF = stencil_offset( C );
   const int     N = A.N, n = N - 1;
   const int     W = C == 0 ? N : ( C == 7 ? n - 1 : n );
   const int64_t         i = (int64_t) blockIdx.x * kThreads + threadIdx.x;
   if ( W <= 0 || i >= tet64( W ) )
      return;
   const int z = slice_of( W, i );
   const int j = (int) ( i - ( tet64( W ) - tet64( W - z ) ) );
   const int y = row_of( W - z, j );
   const int x = j - row_start( W - z, y );
   if ( !p2_inner< C >( N, x, y, z ) )
      return;
   const double* __restrict__ w = A.table + OFF;
   double acc                   = 0.0;
   RowBases R;
   [&]< int... K >( std::integer_sequence< int, K... > ) { ( p2_row_bases_of_kind< C, K >( R, N, n, x, y, z ), ... ); }
   ( std::make_integer_sequence< int, 8 >{} );
   [&]< int... Q >( std::integer_sequence< int, Q... > ) { ( p2_term< C, Q >( A, w, R, acc ), ... ); }
   ( std::make_integer_sequence< int, NQ >{} );
   acc         = A.alpha * acc;
   double* out = C == 0 ? A.dstV + i : A.dstE + edge_block_start( n, C ) + i;
   *out        = A.update == HYTEG_HIP_ADD ? *out + acc : acc;
}

// all eight destination kinds in one launch (blockIdx.y = kind): one ramp-up instead of eight, kinds overlap
__device__ inline void p2_inner_dispatch( const P2FastArgs& A, int kind )
{
   if ( !( ( A.kinds >> kind ) & 1u ) )
      return;
   switch ( kind )
   {
   case 0: p2_inner_body< 0 >( A ); break;
   case 1: p2_inner_body< 1 >( A ); break;
   case 2: p2_inner_body< 2 >( A ); break;
   case 3: p2_inner_body< 3 >( A ); break;
   case 4: p2_inner_body< 4 >( A ); break;
   case 5: p2_inner_body< 5 >( A ); break;
   case 6: p2_inner_body< 6 >( A ); break;
   default: p2_inner_body< 7 >( A ); break;
   }
}
__global__ __launch_bounds__( kThreads ) void p2_inner_kernel( const P2FastArgs A ) { p2_inner_dispatch( A, (int) blockIdx.y ); }
// the same for up to HYTEG_HIP_MAX_BATCH macro-cells of one level (blockIdx.z = cell): the cells' arrays, operator tables and point
// masks travel as pointer lists in the kernel arguments
struct P2BatchPtrs
{
   double*       dstV[HYTEG_HIP_MAX_BATCH];
   double*       dstE[HYTEG_HIP_MAX_BATCH];
   const double* srcV[HYTEG_HIP_MAX_BATCH];
   const double* srcE[HYTEG_HIP_MAX_BATCH];
   const double* table[HYTEG_HIP_MAX_BATCH];
   unsigned      mask[HYTEG_HIP_MAX_BATCH];
};
__device__ inline P2FastArgs p2_batch_view( const P2FastArgs& F, const P2BatchPtrs& P, int cell )
{
   P2FastArgs A = F;
   A.dstV = P.dstV[cell], A.dstE = P.dstE[cell], A.srcV = P.srcV[cell], A.srcE = P.srcE[cell], A.table = P.table[cell];
   return A;
}
__global__ __launch_bounds__( kThreads ) void p2_inner_batch_kernel( const P2FastArgs F, const P2BatchPtrs P )
{
   const int cell = blockIdx.z;
   if ( !( P.mask[cell] & HYTEG_HIP_MASK_INNER ) )
      return;
   p2_inner_dispatch( p2_batch_view( F, P, cell ), (int) blockIdx.y );
}

// =====================================================================================================================
// Row form of the inner stencils (levels >= 3).  p2_inner_kernel above spends ~90 % of its ~900 instructions per
// DoF on index arithmetic (decoding (x, y, z) from the flat index, eight array indices, 64-bit addresses).  Here ONE WAVE
// owns a run of <= 64 consecutive micro-vertex positions x of one row (y, z) -- a TILES_ROWS tile of the vertex array -- and
// produces ALL EIGHT destination kinds at those positions:
//   * y, z are wave-uniform, so every row base is scalar arithmetic: the index of (x0, y, z) in the three array widths
//     (N, N-1, N-2) comes with the tile, the nine neighbour rows (y+dy, z+dz) of each width are layout-algebra deltas;
//   * the union of the sources of all eight stencils (kSrc: distinct (kind, dx, dy, dz); 230 stencil entries share them) is
//     loaded ONCE into registers by buffer loads whose whole byte offset sits in the vector offset -- a row that does not
//     exist or a position beyond the end of a row gives an offset that is either out of range (the descriptor returns 0) or
//     inside the array (a wrong value that only lanes use whose result is not stored): no clamping, no faults;
//   * each destination kind sums its entries in the same order with the same FMAs as p2_inner_kernel (bit-identical
//     results) and stores where p2_inner< C > holds.
// =====================================================================================================================
struct SrcList
{
   int n;
   int kind[160], dx[160], dy[160], dz[160];
};
constexpr SrcList build_src_list()
{
   SrcList U{};
   for ( int c = 0; c < 8; ++c )
   {
      const KindStencil S = build_kind_stencil( c );
      for ( int q = 0; q < S.n; ++q )
      {
         bool found = false;
         for ( int i = 0; i < U.n; ++i )
            found = found || ( U.kind[i] == S.kind[q] && U.dx[i] == S.dx[q] && U.dy[i] == S.dy[q] && U.dz[i] == S.dz[q] );
         if ( !found )
         {
            U.kind[U.n] = S.kind[q], U.dx[U.n] = S.dx[q], U.dy[U.n] = S.dy[q], U.dz[U.n] = S.dz[q];
            ++U.n;
         }
      }
   }
   return U;
}
constexpr SrcList kSrc = build_src_list();
static_assert( kSrc.n <= 160, "source list" );
template < int C >
struct SrcIndexOf
{
   int idx[kMaxStencil];
};
template < int C >
constexpr SrcIndexOf< C > build_src_index()
{
   SrcIndexOf< C >       R{};
   constexpr KindStencil S = KindStencilOf< C >::value;
   for ( int q = 0; q < S.n; ++q )
      for ( int i = 0; i < kSrc.n; ++i )
         if ( kSrc.kind[i] == S.kind[q] && kSrc.dx[i] == S.dx[q] && kSrc.dy[i] == S.dy[q] && kSrc.dz[i] == S.dz[q] )
            R.idx[q] = i;
   return R;
}
template < int C >
struct SrcIndex
{
   static constexpr SrcIndexOf< C > value = build_src_index< C >();
};

// which destination kinds use source i (bit per kind): a kind-restricted apply (the per-type sweeps of the P2 Gauss-Seidel
// smoother) loads only the sources of the kinds it computes
struct SrcUsers
{
   unsigned m[160];
};
constexpr SrcUsers build_src_users()
{
   SrcUsers          R{};
   const KindStencil S8[8] = { KindStencilOf< 0 >::value, KindStencilOf< 1 >::value, KindStencilOf< 2 >::value, KindStencilOf< 3 >::value,
                               KindStencilOf< 4 >::value, KindStencilOf< 5 >::value, KindStencilOf< 6 >::value, KindStencilOf< 7 >::value };
   for ( int i = 0; i < kSrc.n; ++i )
      for ( int c = 0; c < 8; ++c )
         for ( int q = 0; q < S8[c].n; ++q )
            if ( kSrc.kind[i] == S8[c].kind[q] && kSrc.dx[i] == S8[c].dx[q] && kSrc.dy[i] == S8[c].dy[q] && kSrc.dz[i] == S8[c].dz[q] )
               R.m[i] |= 1u << c;
   return R;
}
constexpr SrcUsers kSrcUsers = build_src_users();

struct P2RowsArgs
{
   P2FastArgs  F;
   const Tile* tiles; // TILES_ROWS of the vertex array, capacity 64; pad[0], pad[1] = the tile's first index at widths N-1, N-2
   int         ntiles;
   unsigned    vbytes, ebytes; // sizes of the vertex- and edge-DoF arrays
   int         xcd_chunk;      // row blocks per XCD: block b works on chunk b % 8 (0: blocks in launch order)
};
constexpr int kRowsWaves = 4;

typedef int p2_v2i __attribute__( ( ext_vector_type( 2 ) ) );

// byte offset (without the lane part, biased by -8 so that dx = -1, 0, 1 become the instruction offsets 0, 8, 16) of row
// (y + DY, z + DZ) of source kind K, from the tile's indices i0[width class] of (x0, y, z)
template < int K, int DY, int DZ >
__device__ inline int p2_rows_base( const int ( &i0 )[3], int N, int y, int z )
{
   constexpr int c  = K == 0 ? 0 : ( K == 7 ? 2 : 1 );
   const int     n  = N - 1;
   const int     W  = N - c;
   const int     bk = K == 0 ? 0 : ( K - 1 ) * (int) tet32( (unsigned) n );
   return ( bk + p2_neighbour_row< DY, DZ >( i0[c], W - z, y ) - 1 ) * 8;
}

template < int C, int UPDATE >
__device__ inline void p2_rows_kind( const P2RowsArgs& A, const double ( &U )[kSrc.n], const int ( &i0 )[3], int lane, int x, int y, int z,
                                     int cnt, __amdgpu_buffer_rsrc_t rdV, __amdgpu_buffer_rsrc_t rdE )
{
   constexpr int NQ  = KindStencilOf< C >::value.n;
   constexpr int OFF = stencil_offset( C );
   // constant address space: the weights are read by scalar loads and enter the FMAs as SGPR operands
   typedef const __attribute__( ( address_space( 4 ) ) ) double* cptr_t;
   const cptr_t w   = (cptr_t) ( A.F.table + OFF );
   double       acc = 0.0;
   [&]< int... Q >( std::integer_sequence< int, Q... > ) { ( ( acc = fma( w[Q], U[SrcIndex< C >::value.idx[Q]], acc ) ), ... ); }
   ( std::make_integer_sequence< int, NQ >{} );
   acc                 = A.F.alpha * acc;
   const int  N        = A.F.N, n = N - 1;
   constexpr int c     = C == 0 ? 0 : ( C == 7 ? 2 : 1 );
   const int  bk       = C == 0 ? 0 : ( C - 1 ) * (int) tet32( (unsigned) n );
   const bool on       = lane < cnt && p2_inner< C >( N, x, y, z );
   const int  voff     = on ? ( bk + i0[c] + lane ) * 8 : -8;
   const __amdgpu_buffer_rsrc_t rd = C == 0 ? rdV : rdE;
   if constexpr ( UPDATE == HYTEG_HIP_ADD ) // compile-time: a run-time branch made every kind wait for the previous kind's store
   {
      const p2_v2i o = __builtin_amdgcn_raw_buffer_load_b64( rd, voff, 0, 0 );
      acc            = __hiloint2double( o.y, o.x ) + acc;
   }
   __builtin_amdgcn_raw_buffer_store_b64( p2_v2i{ __double2loint( acc ), __double2hiint( acc ) }, rd, voff, 0, 0 );
}

// RESTRICTED: only some destination kinds are computed (A.F.kinds) and only their sources are loaded; the unrestricted form
// keeps its loads free of branches (with one wave-uniform branch per load the full apply was 16 % slower)
template < int UPDATE, bool RESTRICTED = false >
__device__ inline void p2_rows_body( const P2RowsArgs& A, const Tile* tiles, int ntiles, int xcd_chunk, int block )
{
   // Workgroups b, b + 8, ... run on the same XCD: they take consecutive row groups of ONE chunk of the cell, so that the
   // source rows neighbouring destination rows share (every source row serves ~7 destination rows) are found in that XCD's
   // L2 instead of being fetched by up to four L2s
   if ( xcd_chunk > 0 )
   {
      if ( ( block >> 3 ) >= xcd_chunk )
         return;
      block = ( block & 7 ) * xcd_chunk + ( block >> 3 );
   }
   const int t = __builtin_amdgcn_readfirstlane( block * kRowsWaves + ( (int) threadIdx.x >> 6 ) );
   if ( t >= ntiles )
      return;
   const Tile tl   = tiles[t];
   const int  lane = threadIdx.x & 63;
   const int  N    = A.F.N;
   const int  y = tl.ya, z = tl.z, x = tl.yb + lane;
   const int  i0[3] = { tl.a, tl.pad[0], tl.pad[1] };
   const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc( const_cast< double* >( A.F.srcV ), 0, A.vbytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc( const_cast< double* >( A.F.srcE ), 0, A.ebytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rdV = __builtin_amdgcn_make_buffer_rsrc( A.F.dstV, 0, A.vbytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rdE = __builtin_amdgcn_make_buffer_rsrc( A.F.dstE, 0, A.ebytes, 0x00020000 );
   const int lane8 = lane * 8;

   double U[kSrc.n];
   [&]< int... I >( std::integer_sequence< int, I... > ) {
      ( ( [&] {
           constexpr int K = kSrc.kind[I], DX = kSrc.dx[I], DY = kSrc.dy[I], DZ = kSrc.dz[I];
           double        u = 0.0;
           if ( !RESTRICTED || ( kSrcUsers.m[I] & A.F.kinds ) ) // wave-uniform: sources of kinds that are not computed are not loaded
           {
              const int    voff = p2_rows_base< K, DY, DZ >( i0, N, y, z ) + lane8 + ( DX + 1 ) * 8;
              const p2_v2i v    = __builtin_amdgcn_raw_buffer_load_b64( K == 0 ? rsV : rsE, voff, 0, 0 );
              u                 = __hiloint2double( v.y, v.x );
           }
           U[I] = u;
        }() ),
        ... );
   }
   ( std::make_integer_sequence< int, kSrc.n >{} );

   [&]< int... C >( std::integer_sequence< int, C... > ) {
      ( ( ( !RESTRICTED || ( ( A.F.kinds >> C ) & 1u ) ) ? p2_rows_kind< C, UPDATE >( A, U, i0, lane, x, y, z, tl.cnt, rdV, rdE ) : (void) 0 ), ... );
   }
   ( std::make_integer_sequence< int, 8 >{} );
}
// ---- the same with each source ROW loaded once (round 2, after the counters: 1.0 M load instructions per level-7 launch at
// ~16 cycles each in the CU's address / L1 path are what the row kernel above is bound by -- TCP_TOTAL_CACHE_ACCESSES 18.4 M,
// L1 hit rate 95 %, L1 -> L2 latency 260 cycles, 56 % of the wave cycles waiting for instructions).  The 89 sources are 44
// distinct rows (kind, dy, dz) read at dx = -1, 0, +1: a wave loads each row once, lane l holding x0 - 1 + l, and takes the
// x-neighbours from the neighbouring lanes (DPP wave shifts, as the P1 apply does); it produces 62 positions (lanes 1..62).
// Same sources at the same addresses, same FMA order: bit-identical to p2_rows_body.
struct RowList
{
   int      n;
   int      kind[64], dy[64], dz[64];
   unsigned users[64]; // destination kinds that read the row
   int      ofSrc[160]; // row of source i
};
constexpr RowList build_row_list()
{
   RowList R{};
   for ( int i = 0; i < kSrc.n; ++i )
   {
      int r = -1;
      for ( int k = 0; k < R.n; ++k )
         if ( R.kind[k] == kSrc.kind[i] && R.dy[k] == kSrc.dy[i] && R.dz[k] == kSrc.dz[i] )
            r = k;
      if ( r < 0 )
      {
         r         = R.n++;
         R.kind[r] = kSrc.kind[i], R.dy[r] = kSrc.dy[i], R.dz[r] = kSrc.dz[i];
      }
      R.users[r] |= kSrcUsers.m[i];
      R.ofSrc[i] = r;
   }
   return R;
}
constexpr RowList kRows = build_row_list();
static_assert( kRows.n <= 64, "row list" );

__device__ inline double p2_lane_minus_1( double v )
{
   int lo = __double2loint( v ), hi = __double2hiint( v );
   lo     = __builtin_amdgcn_mov_dpp( lo, 0x138, 0xf, 0xf, true ); // wave_shr:1
   hi     = __builtin_amdgcn_mov_dpp( hi, 0x138, 0xf, 0xf, true );
   return __hiloint2double( hi, lo );
}
__device__ inline double p2_lane_plus_1( double v )
{
   int lo = __double2loint( v ), hi = __double2hiint( v );
   lo     = __builtin_amdgcn_mov_dpp( lo, 0x130, 0xf, 0xf, true ); // wave_shl:1
   hi     = __builtin_amdgcn_mov_dpp( hi, 0x130, 0xf, 0xf, true );
   return __hiloint2double( hi, lo );
}

constexpr int kRowsDppCapacity = 62;

template < int C, int UPDATE >
__device__ inline void p2_rows_kind_dpp( const P2RowsArgs& A, const double ( &R )[kRows.n], const int ( &i0 )[3], int lane, int x, int y, int z,
                                         int cnt, __amdgpu_buffer_rsrc_t rdV, __amdgpu_buffer_rsrc_t rdE )
{
   constexpr int NQ  = KindStencilOf< C >::value.n;
   constexpr int OFF = stencil_offset( C );
   typedef const __attribute__( ( address_space( 4 ) ) ) double* cptr_t;
   const cptr_t w   = (cptr_t) ( A.F.table + OFF );
   double       acc = 0.0;
   [&]< int... Q >( std::integer_sequence< int, Q... > ) {
      ( ( [&] {
           constexpr int I  = SrcIndex< C >::value.idx[Q];
           constexpr int DX = kSrc.dx[I];
           const double  r  = R[kRows.ofSrc[I]];
           const double  u  = DX == 0 ? r : ( DX > 0 ? p2_lane_plus_1( r ) : p2_lane_minus_1( r ) );
           acc              = fma( w[Q], u, acc );
        }() ),
        ... );
   }
   ( std::make_integer_sequence< int, NQ >{} );
   acc                 = A.F.alpha * acc;
   const int  N        = A.F.N, n = N - 1;
   constexpr int c     = C == 0 ? 0 : ( C == 7 ? 2 : 1 );
   const int  bk       = C == 0 ? 0 : ( C - 1 ) * (int) tet32( (unsigned) n );
   const bool on       = lane >= 1 && lane <= cnt && p2_inner< C >( N, x, y, z );
   const int  voff     = on ? ( bk + i0[c] + lane - 1 ) * 8 : -8;
   const __amdgpu_buffer_rsrc_t rd = C == 0 ? rdV : rdE;
   if constexpr ( UPDATE == HYTEG_HIP_ADD )
   {
      const p2_v2i o = __builtin_amdgcn_raw_buffer_load_b64( rd, voff, 0, 0 );
      acc            = __hiloint2double( o.y, o.x ) + acc;
   }
   __builtin_amdgcn_raw_buffer_store_b64( p2_v2i{ __double2loint( acc ), __double2hiint( acc ) }, rd, voff, 0, 0 );
}

template < int UPDATE, bool RESTRICTED = false >
__device__ inline void p2_rows_body_dpp( const P2RowsArgs& A, const Tile* tiles, int ntiles, int xcd_chunk, int block )
{
   if ( xcd_chunk > 0 )
   {
      if ( ( block >> 3 ) >= xcd_chunk )
         return;
      block = ( block & 7 ) * xcd_chunk + ( block >> 3 );
   }
   const int t = __builtin_amdgcn_readfirstlane( block * kRowsWaves + ( (int) threadIdx.x >> 6 ) );
   if ( t >= ntiles )
      return;
   const Tile tl   = tiles[t]; // capacity 62
   const int  lane = threadIdx.x & 63;
   const int  N    = A.F.N;
   const int  y = tl.ya, z = tl.z, x = tl.yb - 1 + lane;
   const int  i0[3] = { tl.a, tl.pad[0], tl.pad[1] };
   const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc( const_cast< double* >( A.F.srcV ), 0, A.vbytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc( const_cast< double* >( A.F.srcE ), 0, A.ebytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rdV = __builtin_amdgcn_make_buffer_rsrc( A.F.dstV, 0, A.vbytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rdE = __builtin_amdgcn_make_buffer_rsrc( A.F.dstE, 0, A.ebytes, 0x00020000 );
   const int lane8 = lane * 8;

   double R[kRows.n];
   [&]< int... I >( std::integer_sequence< int, I... > ) {
      ( ( [&] {
           constexpr int K = kRows.kind[I], DY = kRows.dy[I], DZ = kRows.dz[I];
           double        u = 0.0;
           if ( !RESTRICTED || ( kRows.users[I] & A.F.kinds ) )
           {
              // p2_rows_base is biased by one element: + lane8 addresses x0 - 1 + lane
              const int    voff = p2_rows_base< K, DY, DZ >( i0, N, y, z ) + lane8;
              const p2_v2i v    = __builtin_amdgcn_raw_buffer_load_b64( K == 0 ? rsV : rsE, voff, 0, 0 );
              u                 = __hiloint2double( v.y, v.x );
           }
           R[I] = u;
        }() ),
        ... );
   }
   ( std::make_integer_sequence< int, kRows.n >{} );

   [&]< int... C >( std::integer_sequence< int, C... > ) {
      ( ( ( !RESTRICTED || ( ( A.F.kinds >> C ) & 1u ) ) ? p2_rows_kind_dpp< C, UPDATE >( A, R, i0, lane, x, y, z, tl.cnt, rdV, rdE ) : (void) 0 ), ... );
   }
   ( std::make_integer_sequence< int, 8 >{} );
}

// the three values a wave needs before it can fetch its tile are leading scalar arguments: the command processor preloads them
// into SGPRs (-amdgpu-kernarg-preload-count=4), so the tile load does not wait for a kernel-argument load (as in the P1 apply;
// here without a measurable difference: 41.0 vs 40.7 us at level 7)
template < int UPDATE, bool RESTRICTED, bool DPP = false >
__global__ __launch_bounds__( 64 * kRowsWaves, 2 ) void p2_rows_kernel( const Tile* tiles, int ntiles, int xcd_chunk, const P2RowsArgs A )
{
   if constexpr ( DPP )
      p2_rows_body_dpp< UPDATE, RESTRICTED >( A, tiles, ntiles, xcd_chunk, (int) blockIdx.x );
   else
      p2_rows_body< UPDATE, RESTRICTED >( A, tiles, ntiles, xcd_chunk, (int) blockIdx.x );
}

// Boundary DoFs in stencil form (levels >= 2): which adjacent micro-cells exist depends only on the macro-primitive the DoF
// lies on, so every (destination kind, point class) has its own weight row over the SAME compile-time entry list; entries
// whose weight is zero (neighbour outside the macro-cell, or a genuinely vanishing coupling) are skipped.  Dense enumeration
// over the four faces of each kind's tetrahedral array as in p2_elementwise_kernel.
template < int C, int Q >
__device__ inline void p2_term_class( const P2FastArgs& A, const double* __restrict__ w, const RowBases& R, double& acc )
{
   constexpr int K = KindStencilOf< C >::value.kind[Q], DX = KindStencilOf< C >::value.dx[Q], DY = KindStencilOf< C >::value.dy[Q],
                 DZ = KindStencilOf< C >::value.dz[Q];
   // unconditional load from a safe index instead of a branch: all loads of a thread stay in flight together
   const double wq  = w[Q];
   const int    idx = wq != 0.0 ? R.b[K][DY + 1][DZ + 1] + DX : 0;
   acc              = fma( wq, K == 0 ? A.srcV[idx] : A.srcE[idx], acc );
}
struct P2ClassArgs
{
   P2FastArgs F;
   unsigned   mask;
};
template < int C >
__device__ inline void p2_boundary_body( const P2ClassArgs& B, int bx )
{
   constexpr int     NQ  = KindStencilOf< C >::value.n;
   constexpr int     OFF = class_offset( C );
   const P2FastArgs& A   = B.F;
   const int         N = A.N, n = N - 1;
   const int         W = C == 0 ? N : ( C == 7 ? n - 1 : n );
   if ( W <= 0 )
      return;
   const int T = tri( W );
   const int q = bx * kThreads + threadIdx.x;
   if ( q >= 4 * T )
      return;
   int x, y, z;
   {
      const int f = q / T, r = q - f * T;
      const int j = row_of( W, r );
      const int k = r - row_start( W, j );
      switch ( f )
      {
      case 0:
         x = k, y = j, z = 0;
         break;
      case 1:
         x = k, y = 0, z = j;
         break;
      case 2:
         x = 0, y = k, z = j;
         break;
      default:
         x = k, y = j, z = W - 1 - k - j;
         break;
      }
      const int lowest = ( z == 0 ) ? 0 : ( y == 0 ) ? 1 : ( x == 0 ) ? 2 : 3;
      if ( lowest != f )
         return;
   }
   int cls;
   if constexpr ( C == 0 )
      cls = class_from_flags( z == 0, y == 0, x == 0, x + y + z == N - 1 );
   else
      cls = edge_class( N, x, y, z, C - 1 );
   if ( cls == 14 || !( ( B.mask >> cls ) & 1u ) )
      return;
   const double* __restrict__ w = A.table + OFF + cls * NQ;
   double   acc                 = 0.0;
   RowBases R;
   [&]< int... K >( std::integer_sequence< int, K... > ) { ( p2_row_bases_of_kind< C, K >( R, N, n, x, y, z ), ... ); }
   ( std::make_integer_sequence< int, 8 >{} );
   [&]< int... Q >( std::integer_sequence< int, Q... > ) { ( p2_term_class< C, Q >( A, w, R, acc ), ... ); }
   ( std::make_integer_sequence< int, NQ >{} );
   acc            = A.alpha * acc;
   const int i    = cell_index( W, x, y, z );
   double*   out  = C == 0 ? A.dstV + i : A.dstE + edge_block_start( n, C ) + i;
   *out           = A.update == HYTEG_HIP_ADD ? *out + acc : acc;
}
__device__ inline void p2_boundary_dispatch( const P2ClassArgs& B, int kind, int bx )
{
   if ( !( ( B.F.kinds >> kind ) & 1u ) )
      return;
   switch ( kind )
   {
   case 0: p2_boundary_body< 0 >( B, bx ); break;
   case 1: p2_boundary_body< 1 >( B, bx ); break;
   case 2: p2_boundary_body< 2 >( B, bx ); break;
   case 3: p2_boundary_body< 3 >( B, bx ); break;
   case 4: p2_boundary_body< 4 >( B, bx ); break;
   case 5: p2_boundary_body< 5 >( B, bx ); break;
   case 6: p2_boundary_body< 6 >( B, bx ); break;
   default: p2_boundary_body< 7 >( B, bx ); break;
   }
}
__global__ __launch_bounds__( kThreads ) void p2_boundary_kernel( const P2ClassArgs B ) { p2_boundary_dispatch( B, blockIdx.y, blockIdx.x ); }
__global__ __launch_bounds__( kThreads ) void p2_boundary_batch_kernel( const P2FastArgs F, const P2BatchPtrs P )
{
   const int      cell  = blockIdx.z;
   const unsigned shell = P.mask[cell] & HYTEG_HIP_MASK_SHELL;
   if ( shell == 0 )
      return;
   P2ClassArgs B;
   B.F    = p2_batch_view( F, P, cell );
   B.mask = shell;
   p2_boundary_dispatch( B, blockIdx.y, blockIdx.x );
}

// inner rows and boundary DoFs in ONE launch (they write disjoint DoFs and read the same sources): the boundary workgroups
// -- thread per DoF, a long chain of index arithmetic and dependent loads -- come first and run beside the row waves
// instead of after them
static_assert( kThreads == 64 * kRowsWaves, "the fused launch uses one block shape" );
template < int UPDATE, bool DPP = false >
__global__ __launch_bounds__( kThreads, 2 ) void p2_apply_fused_kernel( const Tile* tiles, int ntiles, int xcd_chunk, const P2RowsArgs A,
                                                                        unsigned shellMask, int nbx )
{
   if ( (int) blockIdx.x < 8 * nbx )
   {
      P2ClassArgs B;
      B.F    = A.F;
      B.mask = shellMask;
      p2_boundary_dispatch( B, (int) blockIdx.x / nbx, (int) blockIdx.x % nbx );
      return;
   }
   if constexpr ( DPP )
      p2_rows_body_dpp< UPDATE >( A, tiles, ntiles, xcd_chunk, (int) blockIdx.x - 8 * nbx );
   else
      p2_rows_body< UPDATE >( A, tiles, ntiles, xcd_chunk, (int) blockIdx.x - 8 * nbx );
}

// =====================================================================================================================
// Row kernel with every point class (round 3; levels >= 3, all destination kinds, masks that include the inner DoFs): replaces the
// launch of p2_rows_body_dpp + p2_boundary_body.  One wave owns a run of 62 positions of a row (y, z) of the vertex array (lanes
// 1..62; lanes 0 and 63 hold the x-neighbours), loads the 44 source rows once and produces all eight kinds, as p2_rows_body_dpp does.
// What is new:
//   * BOUNDARY DoFs are computed by the same waves.  The point class of a DoF -- which adjacent micro-cells exist -- depends on
//     four flags (z = 0, y = 0, x = 0, x + y + z = n; for an edge DoF: both end points).  The first two are wave-uniform, and within
//     a row only its FIRST DoF can have x = 0 and only its LAST one x + y + z = n.  So three passes, each with ONE class per wave and
//     therefore a wave-uniform weight row of the operator table (scalar loads, weights as SGPR operands): pass 0 all DoFs of the run
//     off those two planes, pass 1 the DoF at x = 0 (tiles with x0 = 0, the four kinds that can lie in that plane), pass 2 the last
//     DoF of the row (the tile that holds it, the four kinds that can lie on x + y + z = n).  Passes 1 and 2 run the whole wave for
//     one lane's DoF (138 FMAs each) -- far cheaper than the thread-per-DoF kernel, whose 65-96 loads per DoF hit a cache line each
//     on these two faces: level 7 (all DoFs) 44.4 -> 30.1 us, level 8 256 -> 156 us (profiles/r03_p2_class_rows.txt).
//   * The sum of a DoF runs in three partial sums (entries with dx = 0, +1, -1, each in the order of the entry list); the two
//     x-neighbour sums move by one lane at the end (two wave shifts per DoF instead of one per entry: 357 instead of 546 vector
//     instructions per wave, 99 VGPRs, 4 waves per SIMD).  Results agree with the other kernels to rounding, not bit for bit.
//   * Rows below y = 0 / z = 0 do not exist and are read as 0 (their base is moved beyond every array): the class weights of the
//     neighbours outside the macro-cell are exactly 0 and never meet a stray value.  Positions beyond the ends of a row (lane 0 of
//     the first tile, lanes past the last entry) read whatever the layout holds there, finite for finite input, and meet either a
//     zero weight or a lane that stores nothing -- as in p2_term_class, which reads entry 0 for its zero weights.
//
// Measured and not kept (same file): two positions per lane with 16-byte loads (NP = 2; range-checked dword by dword, so the half
// of a pair beyond the end of the array reads as 0): 196 VGPRs, slower at every level (level 7: 32.2 us inner DoFs against 25.0);
// a z-march (rows of slices z-1 .. z+2 in four register slots, 20 row loads per slice instead of 44): 184 VGPRs, 160 spilled
// SGPRs, 3.7 us per slice and wave, 40.7 us at level 7; the launch with every load and store forced out of range and no FMAs
// still takes 17 of 27 us -- the instruction stream of a wave, not the memory, is what these kernels are bound by.
// =====================================================================================================================
constexpr int      kClassRowsMinLevel = 3;
#ifndef HYTEG_P2_CLASS_ROWS_WAVES
#define HYTEG_P2_CLASS_ROWS_WAVES 4
#endif
constexpr int      kClassRowsWaves    = HYTEG_P2_CLASS_ROWS_WAVES; // waves per workgroup
#ifndef HYTEG_P2_DST_AUX
#define HYTEG_P2_DST_AUX 0
#endif
constexpr int      kClassRowsDstAux   = HYTEG_P2_DST_AUX; // cache policy of the destination arrays: 0 = plain; 2 = nontemporal measured: level 7 30.1 -> 29.1 us, level 8 156 -> 162, levels 4-5 +5 %
typedef int p2_v4i __attribute__( ( ext_vector_type( 4 ) ) );

template < int C >
struct DxUse
{
   bool plus, minus;
};
template < int C >
constexpr DxUse< C > build_dx_use()
{
   DxUse< C >            U{};
   constexpr KindStencil S = KindStencilOf< C >::value;
   for ( int q = 0; q < S.n; ++q )
   {
      U.plus  = U.plus || S.dx[q] > 0;
      U.minus = U.minus || S.dx[q] < 0;
   }
   return U;
}

// destination kind C at the NP positions xa .. xa + NP - 1 of row (y, z) a lane holds; R[row] = its NP source values in that row
// PASS 0: the DoFs off the planes x = 0 and x + y + z = n (one class per row: flags z == 0, y == 0).  PASS 1: the DoF at x = 0 of the row
// (kinds that can lie in that plane; tiles with x0 = 0).  PASS 2: the last DoF of the row, on x + y + z = n (kinds that can lie in that
// plane; the tile that holds it), unless it is the one at x = 0.  Every pass has ONE point class per wave, so its weights are a
// wave-uniform row of the operator table; passes 1 and 2 run the whole wave for one lane's DoF.
template < int C, int UPDATE, int NP, int PASS, bool RESTRICTED >
__device__ __forceinline__ void p2_classrows_kind( const P2RowsArgs& A, const double ( &R )[kRows.n][NP], const int ( &i0 )[3], int lane, int xa, int x0,
                                              int y, int z, unsigned mask, __amdgpu_buffer_rsrc_t rdV, __amdgpu_buffer_rsrc_t rdE )
{
   constexpr int  NQ = KindStencilOf< C >::value.n;
   constexpr bool F0 = C == 0 || C == 1 || C == 2 || C == 4; // kinds whose DoFs can lie in the plane z = 0 / y = 0 / x = 0 / x + y + z = n
   constexpr bool F1 = C == 0 || C == 1 || C == 3 || C == 5; // (both end points of the edge)
   constexpr bool F2 = C == 0 || C == 2 || C == 3 || C == 6;
   constexpr bool F3 = C == 0 || C == 4 || C == 5 || C == 6;
   if constexpr ( ( PASS == 1 && !F2 ) || ( PASS == 2 && !F3 ) )
      return;
   if ( RESTRICTED && !( ( A.F.kinds >> C ) & 1u ) ) // wave-uniform: a kind-restricted apply (the per-type sweeps of the P2 Gauss-Seidel smoother)
      return;
   const int  Nn = A.F.N, nn = Nn - 1;
   const int  top = ( C == 0 ? Nn - 1 : ( C == 7 ? nn - 2 : nn - 1 ) ) - y - z; // x of the last entry of the row in the kind's array
   const bool f0 = F0 && z == 0, f1 = F1 && y == 0;
   int        cls, xOnly = 0;
   if constexpr ( PASS == 0 )
      cls = f0 ? ( f1 ? 0 : 6 ) : ( f1 ? 7 : 14 );
   else if constexpr ( PASS == 1 )
   {
      if ( x0 != 0 || top < 0 )
         return;
      cls = class_from_flags( f0, f1, 1, F3 && top == 0 );
   }
   else
   {
      xOnly = top;
      if ( xOnly < ( F2 ? 1 : 0 ) || xOnly < x0 || xOnly >= x0 + 62 * NP )
         return;
      cls = class_from_flags( f0, f1, 0, 1 );
   }
   if ( !( ( mask >> cls ) & 1u ) ) // wave-uniform
      return;
   constexpr int OFF_INNER = stencil_offset( C ), OFF_CLASS = class_offset( C ); // forced constant evaluation (none of the table code on the device)
   const int     woff      = cls == 14 ? OFF_INNER : OFF_CLASS + cls * NQ;
   typedef const __attribute__( ( address_space( 4 ) ) ) double* cptr_t;
   const cptr_t w = (cptr_t) ( A.F.table + woff );
   double       a0[NP] = {}, ap[NP] = {}, am[NP] = {};
   [&]< int... Q >( std::integer_sequence< int, Q... > ) {
      ( ( [&] {
           constexpr int I   = SrcIndex< C >::value.idx[Q];
           constexpr int DX  = kSrc.dx[I];
           constexpr int row = kRows.ofSrc[I];
           const double  wq  = w[Q];
           for ( int p = 0; p < NP; ++p )
              if constexpr ( DX == 0 )
                 a0[p] = fma( wq, R[row][p], a0[p] );
              else if constexpr ( DX > 0 )
                 ap[p] = fma( wq, R[row][p], ap[p] );
              else
                 am[p] = fma( wq, R[row][p], am[p] );
        }() ),
        ... );
   }
   ( std::make_integer_sequence< int, NQ >{} );
   // position p takes the dx = +1 sum formed at position p + 1 and the dx = -1 sum formed at position p - 1 (in the next / previous lane
   // at the ends of the lane's run)
   constexpr DxUse< C > U = build_dx_use< C >();
   double               acc[NP];
   for ( int p = 0; p < NP; ++p )
      acc[p] = a0[p];
   if constexpr ( U.plus )
   {
      const double next = p2_lane_plus_1( ap[0] );
      for ( int p = 0; p < NP; ++p )
         acc[p] += p + 1 < NP ? ap[p + 1 < NP ? p + 1 : 0] : next;
   }
   if constexpr ( U.minus )
   {
      const double prev = p2_lane_minus_1( am[NP - 1] );
      for ( int p = 0; p < NP; ++p )
         acc[p] += p >= 1 ? am[p >= 1 ? p - 1 : 0] : prev;
   }
   const int     N  = A.F.N, n = N - 1;
   constexpr int c  = C == 0 ? 0 : ( C == 7 ? 2 : 1 );
   const int     bk = C == 0 ? 0 : ( C - 1 ) * (int) tet32( (unsigned) n );
   const __amdgpu_buffer_rsrc_t rd = C == 0 ? rdV : rdE;
   [&]< int... P >( std::integer_sequence< int, P... > ) {
      ( ( [&] {
           const int x = xa + P, s = x + y + z;
           // the DoF exists in its array and lies neither on x = 0 nor on x + y + z = n: p2_inner< C > without its conditions on y and z
           bool here;
           if constexpr ( C == 0 )
              here = x >= 1 && s <= N - 2;
           else if constexpr ( C == 1 )
              here = s < n;
           else if constexpr ( C == 2 || C == 3 )
              here = x > 0 && s < n;
           else if constexpr ( C == 6 )
              here = x > 0 && s < n - 1;
           else
              here = s < n - 1;
           bool on = lane >= 1 && lane <= 62;
           if constexpr ( PASS == 0 )
              on = on && here;
           else
              on = on && x == xOnly;
           const int  voff = on ? ( bk + i0[c] + NP * ( lane - 1 ) + P ) * 8 : -8;
           double     v    = A.F.alpha * acc[P];
           if constexpr ( UPDATE == HYTEG_HIP_ADD )
           {
              const p2_v2i o = __builtin_amdgcn_raw_buffer_load_b64( rd, voff, 0, kClassRowsDstAux );
              v              = __hiloint2double( o.y, o.x ) + v;
           }
           __builtin_amdgcn_raw_buffer_store_b64( p2_v2i{ __double2loint( v ), __double2hiint( v ) }, rd, voff, 0, kClassRowsDstAux );
        }() ),
        ... );
   }
   ( std::make_integer_sequence< int, NP >{} );
}

// RESTRICTED: only the destination kinds of A.F.kinds are computed and only the rows they read are loaded (the others' bases are moved
// beyond the arrays: the load is issued and returns 0 without touching memory)
template < int UPDATE, int NP, bool RESTRICTED = false >
__device__ __forceinline__ void p2_classrows_body( const P2RowsArgs& A, const Tile* tiles, int ntiles, int xcd_chunk, int block, unsigned mask )
{
   if ( xcd_chunk > 0 )
   {
      if ( ( block >> 3 ) >= xcd_chunk )
         return;
      block = ( block & 7 ) * xcd_chunk + ( block >> 3 );
   }
   const int t = __builtin_amdgcn_readfirstlane( block * kClassRowsWaves + ( (int) threadIdx.x >> 6 ) );
   if ( t >= ntiles )
      return;
   const Tile tl    = tiles[t]; // a, pad[0], pad[1]: index of (x0, y, z) at widths N, N-1, N-2; ya = y, yb = x0
   const int  lane  = threadIdx.x & 63;
   const int  N     = A.F.N;
   const int  y = tl.ya, z = tl.z, xa = tl.yb + NP * ( lane - 1 ); // lane 0 holds the NP positions in front of x0
   if ( !( mask & HYTEG_HIP_MASK_INNER ) )
   {
      // boundary classes only: a tile off the planes y = 0, z = 0 that holds neither the first nor (one of) the last entries of its rows
      // has nothing to compute
      const bool ends = tl.yb == 0 || tl.yb + 62 * NP > N - 2 - y - z;
      if ( !( mask & HYTEG_HIP_MASK_SHELL ) || !( ends || y == 0 || z == 0 ) )
         return;
   }
   const int  i0[3] = { tl.a, tl.pad[0], tl.pad[1] };
   const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc( const_cast< double* >( A.F.srcV ), 0, A.vbytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc( const_cast< double* >( A.F.srcE ), 0, A.ebytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rdV = __builtin_amdgcn_make_buffer_rsrc( A.F.dstV, 0, A.vbytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rdE = __builtin_amdgcn_make_buffer_rsrc( A.F.dstE, 0, A.ebytes, 0x00020000 );
   const int  laneBytes = lane * 8 * NP;
   // rows below y = 0 / z = 0 do not exist: their base is moved beyond every array (two scalar flags, one select per such row); rows
   // beyond the top of a kind's array are read only by lanes whose results are not stored
   const bool rowBelow = y >= 1, sliceBelow = z >= 1;
   constexpr int kNowhere = (int) 0x80000000u;

   double R[kRows.n][NP];
   [&]< int... I >( std::integer_sequence< int, I... > ) {
      ( ( [&] {
           constexpr int K = kRows.kind[I], DY = kRows.dy[I], DZ = kRows.dz[I];
           int           base = p2_rows_base< K, DY, DZ >( i0, N, y, z ) + 8 - 8 * NP; // p2_rows_base is biased by one element
           if ( RESTRICTED && !( kRows.users[I] & A.F.kinds ) )
              base = kNowhere;
           if constexpr ( DY < 0 && DZ < 0 )
              base = ( rowBelow && sliceBelow ) ? base : kNowhere;
           else if constexpr ( DY < 0 )
              base = rowBelow ? base : kNowhere;
           else if constexpr ( DZ < 0 )
              base = sliceBelow ? base : kNowhere;
           if constexpr ( NP == 2 )
           {
              // a 16-byte load is range-checked dword by dword: the half of a pair that lies beyond the end of the array reads as 0
              const p2_v4i v = __builtin_amdgcn_raw_buffer_load_b128( K == 0 ? rsV : rsE, base + laneBytes, 0, 0 );
              R[I][0]        = __hiloint2double( v.y, v.x );
              R[I][NP - 1]   = __hiloint2double( v.w, v.z );
           }
           else
           {
              const p2_v2i v = __builtin_amdgcn_raw_buffer_load_b64( K == 0 ? rsV : rsE, base + laneBytes, 0, 0 );
              R[I][0]        = __hiloint2double( v.y, v.x );
           }
        }() ),
        ... );
   }
   ( std::make_integer_sequence< int, kRows.n >{} );

   [&]< int... C >( std::integer_sequence< int, C... > ) {
      ( p2_classrows_kind< C, UPDATE, NP, 0, RESTRICTED >( A, R, i0, lane, xa, tl.yb, y, z, mask, rdV, rdE ), ... );
      if ( mask & HYTEG_HIP_MASK_SHELL ) // wave-uniform: the DoFs on x = 0 and on x + y + z = n
      {
         ( p2_classrows_kind< C, UPDATE, NP, 1, RESTRICTED >( A, R, i0, lane, xa, tl.yb, y, z, mask, rdV, rdE ), ... );
         ( p2_classrows_kind< C, UPDATE, NP, 2, RESTRICTED >( A, R, i0, lane, xa, tl.yb, y, z, mask, rdV, rdE ), ... );
      }
   }
   ( std::make_integer_sequence< int, 8 >{} );
}

template < int UPDATE, int NP, bool RESTRICTED >
__global__ __launch_bounds__( 64 * kClassRowsWaves ) void p2_class_rows_kernel( const Tile* tiles, int ntiles, int xcd_chunk, const P2RowsArgs A, unsigned mask )
{
   p2_classrows_body< UPDATE, NP, RESTRICTED >( A, tiles, ntiles, xcd_chunk, (int) blockIdx.x, mask );
}

// the same for up to HYTEG_HIP_MAX_BATCH macro-cells of one level (blockIdx.y = cell), as p2_inner_batch_kernel
template < int UPDATE, bool RESTRICTED >
__global__ __launch_bounds__( 64 * kClassRowsWaves ) void p2_class_rows_batch_kernel( const Tile* tiles, int ntiles, const P2RowsArgs A, const P2BatchPtrs P )
{
   const int      cell = blockIdx.y;
   const unsigned mask = P.mask[cell];
   if ( mask == 0 )
      return;
   P2RowsArgs B = A;
   B.F          = p2_batch_view( A.F, P, cell );
   p2_classrows_body< UPDATE, 1, RESTRICTED >( B, tiles, ntiles, 0, (int) blockIdx.x, mask );
}

// first level the row kernel with every point class is used at (HYTEG_HIP_P2_CLASS_ROWS_MIN_LEVEL, hyteg_hip_p2_set_class_rows_min_level: tests run it at small
// levels, 99 = the row kernel of round 2 at every level)
std::atomic< int >& class_rows_min_level()
{
   static std::atomic< int > v( [] {
      const char* e = std::getenv( "HYTEG_HIP_P2_CLASS_ROWS_MIN_LEVEL" );
      return e ? std::atoi( e ) : kClassRowsMinLevel;
   }() );
   return v;
}

// tiles of the row kernel with every point class: (x0, y, z) with x0 a multiple of the capacity (62 positions per lane position), over the positions (x, y, z) with x + y + z <= N - 2 (where some kind
// has a DoF off the planes x = 0 and x + y + z = n)
int get_class_rows_tiles( int level, int capacity, TileTable* out )
{
   static std::mutex                                         mtx;
   static std::map< std::tuple< int, int, int >, TileTable > cache;
   int                                                  dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   std::lock_guard< std::mutex > lock( mtx );
   const auto                    key = std::make_tuple( dev, level, capacity );
   auto                          it  = cache.find( key );
   if ( it != cache.end() )
   {
      *out = it->second;
      return HYTEG_HIP_OK;
   }
   const int           N = ( 1 << level ) + 1;
   std::vector< Tile > host;
   for ( int z = 0; z <= N - 1; ++z )
      for ( int y = 0; y <= N - 1 - z; ++y )
         for ( int x0 = 0; x0 <= N - 1 - y - z; x0 += capacity )
         {
            Tile tl{};
            tl.a      = cell_index( N, x0, y, z );
            tl.pad[0] = cell_index( N - 1, x0, y, z );
            tl.pad[1] = cell_index( N - 2, x0, y, z );
            tl.ya = y, tl.yb = x0, tl.z = z;
            tl.cnt = std::min( capacity, N - y - z - x0 );
            host.push_back( tl );
         }
   TileTable tt;
   tt.count = (int) host.size();
   void* p  = nullptr;
   HH_CHECK_HIP( hipMalloc( &p, host.size() * sizeof( Tile ) ) );
   HH_CHECK_HIP( hipMemcpy( p, host.data(), host.size() * sizeof( Tile ), hipMemcpyHostToDevice ) );
   tt.dev     = static_cast< const Tile* >( p );
   cache[key] = tt;
   *out       = tt;
   return HYTEG_HIP_OK;
}

// host: does micro-cell (type t, index m) lie inside a macro-cell of width N?
bool micro_cell_inside( int t, int mx, int my, int mz, int N )
{
   for ( int v = 0; v < 4; ++v )
   {
      const int x = mx + cMicroVerts[t][v][0], y = my + cMicroVerts[t][v][1], z = mz + cMicroVerts[t][v][2];
      if ( x < 0 || y < 0 || z < 0 || x + y + z > N - 1 )
         return false;
   }
   return true;
}
int host_class_of( int c, int x, int y, int z, int N )
{
   static const int ends[7][2][3] = { { { 0, 0, 0 }, { 1, 0, 0 } }, { { 0, 0, 0 }, { 0, 1, 0 } }, { { 0, 0, 0 }, { 0, 0, 1 } },
                                      { { 1, 0, 0 }, { 0, 1, 0 } }, { { 1, 0, 0 }, { 0, 0, 1 } }, { { 0, 1, 0 }, { 0, 0, 1 } },
                                      { { 0, 1, 0 }, { 1, 0, 1 } } };
   int f[4] = { 1, 1, 1, 1 };
   const int npts = c == 0 ? 1 : 2;
   for ( int e = 0; e < npts; ++e )
   {
      const int px = x + ( c == 0 ? 0 : ends[c - 1][e][0] ), py = y + ( c == 0 ? 0 : ends[c - 1][e][1] ), pz = z + ( c == 0 ? 0 : ends[c - 1][e][2] );
      f[0] &= pz == 0, f[1] &= py == 0, f[2] &= px == 0, f[3] &= px + py + pz == N - 1;
   }
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

} // namespace

extern "C" {

HYTEG_HIP_API size_t hyteg_hip_p2_operator_table_size( void ) { return (size_t) kOperatorTableSize; }

HYTEG_HIP_API int hyteg_hip_p2_set_class_rows_min_level( int level )
{
   const int before = class_rows_min_level().load();
   class_rows_min_level().store( level < 3 ? 3 : level );
   return before;
}

HYTEG_HIP_API int hyteg_hip_p2_build_operator_table( const double* elmat_host, double* table_host )
{
   HH_REQUIRE( elmat_host && table_host, "p2_build_operator_table: null pointer" );
   for ( int k = 0; k < 600; ++k )
      table_host[k] = elmat_host[k];
   for ( int c = 0; c < 8; ++c )
   {
      const KindStencil S   = build_kind_stencil( c );
      double*           w   = table_host + stencil_offset( c );
      for ( int q = 0; q < S.n; ++q )
         w[q] = 0.0;
      for ( int t = 0; t < 6; ++t )
         for ( int k = 0; k < 10; ++k )
         {
            const CLocal row = c_local( t, k );
            if ( row.kind != c )
               continue;
            for ( int j = 0; j < 10; ++j )
            {
               const CLocal col = c_local( t, j );
               const int    dx = col.ox - row.ox, dy = col.oy - row.oy, dz = col.oz - row.oz;
               for ( int q = 0; q < S.n; ++q )
                  if ( S.kind[q] == col.kind && S.dx[q] == dx && S.dy[q] == dy && S.dz[q] == dz )
                     w[q] += elmat_host[100 * t + 10 * k + j];
            }
         }
      // boundary classes: the adjacent micro-cells that exist are read off a representative DoF of that class at level 3
      const int Nr = 9, nr = 8, Wr = c == 0 ? Nr : ( c == 7 ? nr - 1 : nr );
      double*   wc = table_host + class_offset( c );
      for ( int k = 0; k < 14 * S.n; ++k )
         wc[k] = 0.0;
      for ( int cls = 0; cls < 14; ++cls )
      {
         bool found = false;
         for ( int z = 0; z < Wr && !found; ++z )
            for ( int y = 0; y < Wr - z && !found; ++y )
               for ( int x = 0; x < Wr - z - y && !found; ++x )
               {
                  if ( host_class_of( c, x, y, z, Nr ) != cls )
                     continue;
                  found = true;
                  for ( int t = 0; t < 6; ++t )
                     for ( int k = 0; k < 10; ++k )
                     {
                        const CLocal row = c_local( t, k );
                        if ( row.kind != c || !micro_cell_inside( t, x - row.ox, y - row.oy, z - row.oz, Nr ) )
                           continue;
                        for ( int j = 0; j < 10; ++j )
                        {
                           const CLocal col = c_local( t, j );
                           const int    dx = col.ox - row.ox, dy = col.oy - row.oy, dz = col.oz - row.oz;
                           for ( int q = 0; q < S.n; ++q )
                              if ( S.kind[q] == col.kind && S.dx[q] == dx && S.dy[q] == dy && S.dz[q] == dz )
                                 wc[cls * S.n + q] += elmat_host[100 * t + 10 * k + j];
                        }
                     }
               }
      }
   }
   return HYTEG_HIP_OK;
}

// ---- f4: the constant-stencil operator's kernel seam ----------------------------------------------------------------
// P2ConstantOperator::apply = four sub-operators (P2ConstantOperator.cpp:100-112) whose macro-cell kernels take stencil maps:
//   vertex->vertex  std::map< Index, real_t >                                                    (P1ConstantOperator)
//   edge->vertex    std::map< EdgeDoFOrientation, std::map< Index, real_t > >                    e2vStencilMap[ leaf orientation ][ offset ]
//   vertex->edge    std::map< EdgeDoFOrientation, std::map< Index, real_t > >                    v2eStencilMap[ centre orientation ][ offset ]
//   edge->edge      std::map< EdgeDoFOrientation, std::map< EdgeDoFOrientation, std::map< Index, real_t > > >   [ centre ][ leaf ][ offset ]
// (mixedoperators/EdgeDoFToVertexDoFOperator/generatedKernels/apply_3D_macrocell_edgedof_to_vertexdof_replace.hpp:36,
//  mixedoperators/VertexDoFToEdgeDoFOperator/generatedKernels/apply_3D_macrocell_vertexdof_to_edgedof_replace.hpp:36,
//  constant_stencil_operator/EdgeDoFGeneratedKernels/apply_3D_macrocell_edgedof_to_edgedof_replace.hpp:37).
// The key sets of those maps are a geometric fact -- the (source kind, offset) lists of KindStencil above -- so a binding passes
// only the VALUES, flattened in the maps' own iteration order (orientations in enum order X, Y, Z, XY, XZ, YZ, XYZ; offsets in
// indexing::Index order z, y, x), the four maps one after the other.  hyteg_hip_p2_constant_stencil_layout returns the keys in
// that order so that a binding can check its maps against them.
} // extern "C"
namespace {
struct CanonKey
{
   int c, s, dx, dy, dz; // destination kind (0 vertex, 1..7 edge X..XYZ), source kind, offset source index - destination index
};
inline int canon_group( const CanonKey& k ) { return k.c == 0 ? ( k.s == 0 ? 0 : 1 ) : ( k.s == 0 ? 2 : 3 ); }
const std::vector< CanonKey >& canonical_keys()
{
   static const std::vector< CanonKey > keys = [] {
      std::vector< CanonKey > v;
      for ( int c = 0; c < 8; ++c )
      {
         const KindStencil S = build_kind_stencil( c );
         for ( int q = 0; q < S.n; ++q )
            v.push_back( CanonKey{ c, S.kind[q], S.dx[q], S.dy[q], S.dz[q] } );
      }
      std::sort( v.begin(), v.end(), []( const CanonKey& a, const CanonKey& b ) {
         const int ga = canon_group( a ), gb = canon_group( b );
         if ( ga != gb )
            return ga < gb;
         if ( a.c != b.c )
            return a.c < b.c;
         if ( a.s != b.s )
            return a.s < b.s;
         if ( a.dz != b.dz )
            return a.dz < b.dz;
         if ( a.dy != b.dy )
            return a.dy < b.dy;
         return a.dx < b.dx;
      } );
      return v;
   }();
   return keys;
}
// position of table entry (c, q) in the canonical list
int canonical_position( int c, int q )
{
   static const std::vector< std::vector< int > > pos = [] {
      const auto&                       keys = canonical_keys();
      std::vector< std::vector< int > > p( 8 );
      for ( int c2 = 0; c2 < 8; ++c2 )
      {
         const KindStencil S = build_kind_stencil( c2 );
         p[c2].assign( S.n, -1 );
         for ( int q2 = 0; q2 < S.n; ++q2 )
            for ( size_t i = 0; i < keys.size(); ++i )
               if ( keys[i].c == c2 && keys[i].s == S.kind[q2] && keys[i].dx == S.dx[q2] && keys[i].dy == S.dy[q2] && keys[i].dz == S.dz[q2] )
                  p[c2][q2] = (int) i;
      }
      return p;
   }();
   return pos[c][q];
}
// device copies of operator tables built inside this file (the sub-operator entry points), by content
int cached_table( const std::vector< double >& host, const double** dev_out )
{
   static std::mutex                                                   mtx;
   static std::map< std::pair< int, std::vector< double > >, double* > cache;
   int                                                                 dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   std::lock_guard< std::mutex > lock( mtx );
   auto                          it = cache.find( { dev, host } );
   if ( it == cache.end() )
   {
      void* p = nullptr;
      HH_CHECK_HIP( hipMalloc( &p, host.size() * sizeof( double ) ) );
      HH_CHECK_HIP( hipMemcpy( p, host.data(), host.size() * sizeof( double ), hipMemcpyHostToDevice ) );
      it = cache.emplace( std::make_pair( dev, host ), static_cast< double* >( p ) ).first;
   }
   *dev_out = it->second;
   return HYTEG_HIP_OK;
}
// the seven edge-DoF block pointers of the reference's kernels (alphabetical: X, XY, XYZ, XZ, Y, YZ, Z) must be the blocks of ONE
// edge-DoF array (EdgeDoFIndexing.hpp:920-985: X, Y, Z, XY, XZ, YZ blocks of tet(2^level) entries, then XYZ)
template < typename P >
bool blocks_of_one_array( P x, P xy, P xyz, P xz, P y, P yz, P z, int level )
{
   const int64_t b = tet64( (int64_t) 1 << level );
   return y == x + b && z == x + 2 * b && xy == x + 3 * b && xz == x + 4 * b && yz == x + 5 * b && xyz == x + 6 * b;
}
} // namespace
extern "C" {

HYTEG_HIP_API int hyteg_hip_p2_constant_stencil_layout( int* counts, int* keys )
{
   HH_REQUIRE( counts, "p2_constant_stencil_layout: null pointer" );
   const auto& K = canonical_keys();
   counts[0] = counts[1] = counts[2] = counts[3] = 0;
   for ( size_t i = 0; i < K.size(); ++i )
   {
      ++counts[canon_group( K[i] )];
      if ( keys )
         keys[5 * i] = K[i].c, keys[5 * i + 1] = K[i].s, keys[5 * i + 2] = K[i].dx, keys[5 * i + 3] = K[i].dy, keys[5 * i + 4] = K[i].dz;
   }
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2_build_operator_table_from_stencils( const double* inner, const double* classes, double* table_host )
{
   HH_REQUIRE( inner && table_host, "p2_build_operator_table_from_stencils: null pointer" );
   const int total = (int) canonical_keys().size();
   for ( int k = 0; k < kOperatorTableSize; ++k )
      table_host[k] = 0.0; // no element matrices: such a table serves levels >= 2 (levels 0, 1 gather micro-cell by micro-cell)
   for ( int c = 0; c < 8; ++c )
   {
      const int n = stencil_count( c );
      for ( int q = 0; q < n; ++q )
      {
         const int pos                       = canonical_position( c, q );
         table_host[stencil_offset( c ) + q] = inner[pos];
         if ( classes )
            for ( int cls = 0; cls < 14; ++cls )
               table_host[class_offset( c ) + cls * n + q] = classes[(size_t) cls * total + pos];
      }
   }
   return HYTEG_HIP_OK;
}

// one sub-operator on the INNER DoFs of a macro-cell (what the reference's macro-cell kernels update): a table that carries only
// that sub-operator's weights, the destination kinds it writes
static int apply_sub_operator( double* dst_vertex, double* dst_edge, const double* src_vertex, const double* src_edge, int level, int group,
                               const double* values, int update, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( level >= 2 && level <= HYTEG_HIP_P2_MAX_LEVEL, "p2 constant sub-operator: level out of range [2,9]" );
   const auto&           K = canonical_keys();
   std::vector< double > inner( K.size(), 0.0 );
   int                   first = 0;
   for ( size_t i = 0; i < K.size() && canon_group( K[i] ) < group; ++i )
      ++first;
   for ( size_t i = first; i < K.size() && canon_group( K[i] ) == group; ++i )
      inner[i] = values[i - first];
   std::vector< double > table( kOperatorTableSize );
   hyteg_hip_p2_build_operator_table_from_stencils( inner.data(), nullptr, table.data() );
   const double* table_dev = nullptr;
   const int     rc        = cached_table( table, &table_dev );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   return hyteg_hip_p2_elementwise_apply_cell_kinds( dst_vertex, dst_edge, src_vertex, src_edge, level, table_dev, 1.0, update, HYTEG_HIP_MASK_INNER,
                                                     group <= 1 ? 0x01u : 0xFEu, stream );
}

HYTEG_HIP_API int hyteg_hip_p2_apply_cell_edgedof_to_vertexdof( const double* src_x, const double* src_xy, const double* src_xyz, const double* src_xz,
                                                                const double* src_y, const double* src_yz, const double* src_z, double* dst_vertex,
                                                                const double* e2v_stencil, int level, int update, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( src_x && dst_vertex && e2v_stencil, "p2_apply_cell_edgedof_to_vertexdof: null pointer" );
   HH_REQUIRE( blocks_of_one_array( src_x, src_xy, src_xyz, src_xz, src_y, src_yz, src_z, level ),
               "p2_apply_cell_edgedof_to_vertexdof: the seven source pointers are not the blocks of one edge-DoF array" );
   // the vertex source of the fused kernel carries zero weights here and its edge destination is masked: the edge source (a
   // genuine, finite source, at least as long as a vertex array) stands in for the former, an unwritten pointer for the latter
   return apply_sub_operator( dst_vertex, const_cast< double* >( src_x ) + 1, src_x, src_x, level, 1, e2v_stencil, update, stream );
}

HYTEG_HIP_API int hyteg_hip_p2_apply_cell_vertexdof_to_edgedof( double* dst_x, double* dst_xy, double* dst_xyz, double* dst_xz, double* dst_y,
                                                                double* dst_yz, double* dst_z, const double* src_vertex, int level,
                                                                const double* v2e_stencil, int update, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst_x && src_vertex && v2e_stencil, "p2_apply_cell_vertexdof_to_edgedof: null pointer" );
   HH_REQUIRE( blocks_of_one_array( dst_x, dst_xy, dst_xyz, dst_xz, dst_y, dst_yz, dst_z, level ),
               "p2_apply_cell_vertexdof_to_edgedof: the seven destination pointers are not the blocks of one edge-DoF array" );
   // edge source: zero weights, but 0 * x is only 0 for finite x -- a zero-filled array of the level's edge-DoF size stands in
   // (kept per device and level); vertex destination: masked, never written
   static std::mutex                              mtx;
   static std::map< std::pair< int, int >, double* > zeros;
   int                                            dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   double* zero_edges = nullptr;
   {
      std::lock_guard< std::mutex > lock( mtx );
      auto                          it = zeros.find( { dev, level } );
      if ( it == zeros.end() )
      {
         void*        p     = nullptr;
         const size_t bytes = ( hyteg_hip_p2_edge_array_size( level ) + 1 ) * sizeof( double );
         HH_CHECK_HIP( hipMalloc( &p, bytes ) );
         HH_CHECK_HIP( hipMemset( p, 0, bytes ) );
         it = zeros.emplace( std::make_pair( dev, level ), static_cast< double* >( p ) ).first;
      }
      zero_edges = it->second;
   }
   return apply_sub_operator( const_cast< double* >( src_vertex ) + 1, dst_x, src_vertex, zero_edges, level, 2, v2e_stencil, update, stream );
}

HYTEG_HIP_API int hyteg_hip_p2_apply_cell_edgedof_to_edgedof( double* dst_x, double* dst_xy, double* dst_xyz, double* dst_xz, double* dst_y, double* dst_yz,
                                                              double* dst_z, const double* src_x, const double* src_xy, const double* src_xyz,
                                                              const double* src_xz, const double* src_y, const double* src_yz, const double* src_z,
                                                              const double* e2e_stencil, int level, int update, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst_x && src_x && e2e_stencil, "p2_apply_cell_edgedof_to_edgedof: null pointer" );
   HH_REQUIRE( blocks_of_one_array( dst_x, dst_xy, dst_xyz, dst_xz, dst_y, dst_yz, dst_z, level ) &&
                   blocks_of_one_array( src_x, src_xy, src_xyz, src_xz, src_y, src_yz, src_z, level ),
               "p2_apply_cell_edgedof_to_edgedof: the seven pointers are not the blocks of one edge-DoF array" );
   return apply_sub_operator( dst_x + 1, dst_x, src_x, src_x, level, 3, e2e_stencil, update, stream );
}

HYTEG_HIP_API int hyteg_hip_p2_edge_vector_cell_masked( int                  op,
                                                        double*              dst,
                                                        int                  nsrc,
                                                        const double* const* srcs,
                                                        const double*        scalars,
                                                        int                  level,
                                                        unsigned             mask,
                                                        hyteg_hip_stream_t   stream )
{
   return hyteg_hip_p2_edge_vector_cell_kinds( op, dst, nsrc, srcs, scalars, level, mask, 0xFEu, stream );
}

HYTEG_HIP_API int hyteg_hip_p2_edge_vector_cell_kinds( int                  op,
                                                       double*              dst,
                                                       int                  nsrc,
                                                       const double* const* srcs,
                                                       const double*        scalars,
                                                       int                  level,
                                                       unsigned             mask,
                                                       unsigned             kind_mask,
                                                       hyteg_hip_stream_t   stream )
{
   HH_REQUIRE( dst && op >= 0 && op <= 3, "p2_edge_vector_cell_masked: null dst or bad op" );
   HH_REQUIRE( level >= 0 && level <= HYTEG_HIP_P2_MAX_LEVEL, "p2_edge_vector_cell_masked: level out of range [0,9]" );
   HH_REQUIRE( op == 3 ? scalars != nullptr : ( nsrc >= 1 && nsrc <= HYTEG_HIP_MAX_SRCS && srcs ), "p2_edge_vector_cell_masked: bad sources" );
   HH_REQUIRE( op == 2 || scalars, "p2_edge_vector_cell_masked: null scalars" );
   if ( ( mask & HYTEG_HIP_MASK_ALL ) == 0 || ( kind_mask & 0xFEu ) == 0 )
      return HYTEG_HIP_OK;
   EdgeVecArgs A{};
   A.dst = dst, A.N = ( 1 << level ) + 1, A.nsrc = nsrc, A.op = op, A.mask = mask & HYTEG_HIP_MASK_ALL, A.kinds = kind_mask & 0xFEu;
   A.size = (int64_t) hyteg_hip_p2_edge_array_size( level );
   if ( op == 3 )
      A.c[0] = scalars[0];
   else
      for ( int k = 0; k < nsrc; ++k )
      {
         HH_REQUIRE( srcs[k], "p2_edge_vector_cell_masked: null source" );
         A.src[k] = srcs[k];
         A.c[k]   = scalars ? scalars[k] : 1.0;
      }
   if ( A.size == 0 )
      return HYTEG_HIP_OK;
   hipLaunchKernelGGL( p2_edge_vector_kernel, dim3( (unsigned) ( ( A.size + kThreads - 1 ) / kThreads ) ), dim3( kThreads ), 0,
                       as_stream( stream ), A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2_edge_vector_cells_kinds( int op, int ncells, double* const* dst, int nsrc, const double* const* srcs,
                                                        const double* scalars, int level, const unsigned* masks, unsigned kind_mask,
                                                        hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && masks && op >= 0 && op <= 3, "p2_edge_vector_cells_kinds: null pointer or bad op" );
   HH_REQUIRE( ncells >= 1 && ncells <= HYTEG_HIP_MAX_BATCH, "p2_edge_vector_cells_kinds: 1 <= ncells <= HYTEG_HIP_MAX_BATCH" );
   HH_REQUIRE( level >= 0 && level <= HYTEG_HIP_P2_MAX_LEVEL, "p2_edge_vector_cells_kinds: level out of range [0,9]" );
   HH_REQUIRE( op == 3 ? scalars != nullptr : ( nsrc >= 1 && nsrc <= HYTEG_HIP_MAX_SRCS && srcs ), "p2_edge_vector_cells_kinds: bad sources" );
   HH_REQUIRE( op == 2 || scalars, "p2_edge_vector_cells_kinds: null scalars" );
   if ( ( kind_mask & 0xFEu ) == 0 )
      return HYTEG_HIP_OK;
   EdgeVecBatchArgs A{};
   A.N = ( 1 << level ) + 1, A.nsrc = nsrc, A.op = op, A.kinds = kind_mask & 0xFEu;
   A.size = (int64_t) hyteg_hip_p2_edge_array_size( level );
   if ( A.size == 0 )
      return HYTEG_HIP_OK;
   bool any = false;
   for ( int c = 0; c < ncells; ++c )
   {
      HH_REQUIRE( dst[c], "p2_edge_vector_cells_kinds: null destination" );
      A.dst[c]  = dst[c];
      A.mask[c] = masks[c] & HYTEG_HIP_MASK_ALL;
      any       = any || A.mask[c] != 0;
   }
   if ( !any )
      return HYTEG_HIP_OK;
   if ( op == 3 )
      A.c[0] = scalars[0];
   else
      for ( int k = 0; k < nsrc; ++k )
      {
         A.c[k] = scalars ? scalars[k] : 1.0;
         for ( int c = 0; c < ncells; ++c )
         {
            HH_REQUIRE( srcs[(size_t) k * ncells + c], "p2_edge_vector_cells_kinds: null source" );
            A.src[k][c] = srcs[(size_t) k * ncells + c];
         }
      }
   hipLaunchKernelGGL( p2_edge_vector_batch_kernel, dim3( (unsigned) ( ( A.size + kThreads - 1 ) / kThreads ), (unsigned) ncells ), dim3( kThreads ), 0,
                       as_stream( stream ), A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2_edge_dot_cells_masked( int ncells, const double* const* a, const double* const* b, int level, const unsigned* masks,
                                                      double* results_dev, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( a && b && masks && results_dev, "p2_edge_dot_cells_masked: null pointer" );
   HH_REQUIRE( ncells >= 1 && ncells <= HYTEG_HIP_MAX_BATCH, "p2_edge_dot_cells_masked: 1 <= ncells <= HYTEG_HIP_MAX_BATCH" );
   HH_REQUIRE( level >= 0 && level <= HYTEG_HIP_P2_MAX_LEVEL, "p2_edge_dot_cells_masked: level out of range [0,9]" );
   EdgeDotBatchArgs A{};
   A.size = (int64_t) hyteg_hip_p2_edge_array_size( level ), A.N = ( 1 << level ) + 1, A.result = results_dev;
   for ( int c = 0; c < ncells; ++c )
   {
      HH_REQUIRE( a[c] && b[c], "p2_edge_dot_cells_masked: null array" );
      A.a[c] = a[c], A.b[c] = b[c], A.mask[c] = masks[c] & HYTEG_HIP_MASK_ALL;
   }
   hipLaunchKernelGGL( p2_edge_dot_batch_kernel, dim3( (unsigned) ncells ), dim3( kThreads ), 0, as_stream( stream ), A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2_edge_dot_cell_masked( const double*      a,
                                                     const double*      b,
                                                     int                level,
                                                     unsigned           mask,
                                                     double*            result_dev,
                                                     void*              workspace_dev,
                                                     hyteg_hip_stream_t stream )
{
   HH_REQUIRE( a && b && result_dev && workspace_dev, "p2_edge_dot_cell_masked: null pointer" );
   HH_REQUIRE( level >= 0 && level <= HYTEG_HIP_P2_MAX_LEVEL, "p2_edge_dot_cell_masked: level out of range [0,9]" );
   const int64_t size   = (int64_t) hyteg_hip_p2_edge_array_size( level );
   int64_t       blocks = ( size + kThreads - 1 ) / kThreads;
   blocks               = blocks < 1 ? 1 : ( blocks > kEdgeDotBlocks ? kEdgeDotBlocks : blocks );
   double* partial      = static_cast< double* >( workspace_dev );
   hipLaunchKernelGGL( p2_edge_dot_kernel, dim3( (unsigned) blocks ), dim3( kThreads ), 0, as_stream( stream ), a, b, size, ( 1 << level ) + 1,
                       mask & HYTEG_HIP_MASK_ALL, partial );
   hipLaunchKernelGGL( p2_sum_partials_kernel, dim3( 1 ), dim3( kThreads ), 0, as_stream( stream ), partial, (int) blocks, result_dev );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API size_t hyteg_hip_p2_edge_array_size( int level )
{
   if ( level < 0 || level > HYTEG_HIP_P2_MAX_LEVEL )
      return 0;
   const int64_t n = (int64_t) 1 << level;
   return (size_t) ( 6 * tet64( n ) + tet64( n - 1 ) );
}

HYTEG_HIP_API int hyteg_hip_p2_elementwise_apply_cell( double*            dst_vertex,
                                                       double*            dst_edge,
                                                       const double*      src_vertex,
                                                       const double*      src_edge,
                                                       int                level,
                                                       const double*      optable_dev,
                                                       double             alpha,
                                                       int                update,
                                                       unsigned           mask,
                                                       hyteg_hip_stream_t stream )
{
   return hyteg_hip_p2_elementwise_apply_cell_kinds( dst_vertex, dst_edge, src_vertex, src_edge, level, optable_dev, alpha, update, mask, 0xFFu,
                                                     stream );
}

HYTEG_HIP_API int hyteg_hip_p2_elementwise_apply_cell_kinds( double*            dst_vertex,
                                                             double*            dst_edge,
                                                             const double*      src_vertex,
                                                             const double*      src_edge,
                                                             int                level,
                                                             const double*      optable_dev,
                                                             double             alpha,
                                                             int                update,
                                                             unsigned           mask,
                                                             unsigned           kind_mask,
                                                             hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst_vertex && dst_edge && src_vertex && src_edge && optable_dev, "p2_elementwise_apply_cell: null pointer" );
   HH_REQUIRE( level >= 0 && level <= HYTEG_HIP_P2_MAX_LEVEL, "p2_elementwise_apply_cell: level out of range [0,9]" );
   HH_REQUIRE( dst_vertex != src_vertex && dst_edge != src_edge, "p2_elementwise_apply_cell: src and dst must not alias" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p2_elementwise_apply_cell: bad update type" );
   mask &= HYTEG_HIP_MASK_ALL;
   kind_mask &= 0xFFu;
   if ( mask == 0 || kind_mask == 0 )
      return HYTEG_HIP_OK;
   hipStream_t       s = as_stream( stream );
   static const bool perThread = [] {
      const char* e = std::getenv( "HYTEG_HIP_P2_INNER_THREADS" ); // measurement switch: round 1's thread-per-DoF kernel, two launches
      return e && e[0] == '1';
   }();
   P2FastArgs F;
   F.dstV = dst_vertex, F.dstE = dst_edge, F.srcV = src_vertex, F.srcE = src_edge, F.table = optable_dev, F.alpha = alpha;
   F.N = ( 1 << level ) + 1, F.update = update, F.kinds = kind_mask;
   const int  faces = 4 * tri( F.N );
   const int  nbx   = ( faces + kThreads - 1 ) / kThreads;
   const bool rows  = ( mask & HYTEG_HIP_MASK_INNER ) && level >= 3 && !perThread;
   // a kind-restricted apply (the per-type sweeps of the Gauss-Seidel smoother) takes the class-rows kernel from level 6: below, where the
   // launches are latency-bound, the kernels of round 2 were 5 % faster on the sweep (profiles/r03_p2_class_rows.txt (D))
   if ( level >= 3 && !perThread && level >= class_rows_min_level().load( std::memory_order_relaxed ) && ( kind_mask == 0xFFu || level >= 6 ) )
   {
      // one launch of row waves for the inner DoFs and every boundary class (all kinds, or the kinds of kind_mask)
      TileTable tt;
      const int rc = get_class_rows_tiles( level, 62, &tt );
      if ( rc != HYTEG_HIP_OK )
         return rc;
      P2RowsArgs R;
      R.F = F, R.tiles = tt.dev, R.ntiles = tt.count;
      const int n = F.N - 1;
      R.vbytes    = (unsigned) ( tet64( F.N ) * 8 );
      R.ebytes    = (unsigned) ( ( 6 * tet64( n ) + tet64( n - 1 ) ) * 8 );
      unsigned waveBlocks = (unsigned) ( ( tt.count + kClassRowsWaves - 1 ) / kClassRowsWaves );
      R.xcd_chunk         = 0;
      if ( waveBlocks >= 64 )
      {
         R.xcd_chunk = (int) ( ( waveBlocks + 7 ) / 8 );
         waveBlocks  = 8u * (unsigned) R.xcd_chunk;
      }
#define P2_LAUNCH_CLASS_ROWS( UPD, RES )                                                                                                       \
   hipLaunchKernelGGL( ( p2_class_rows_kernel< UPD, 1, RES > ), dim3( waveBlocks ), dim3( 64 * kClassRowsWaves ), 0, s, R.tiles, R.ntiles, R.xcd_chunk, R, mask )
      if ( kind_mask != 0xFFu && update == HYTEG_HIP_ADD )
         P2_LAUNCH_CLASS_ROWS( HYTEG_HIP_ADD, true );
      else if ( kind_mask != 0xFFu )
         P2_LAUNCH_CLASS_ROWS( HYTEG_HIP_REPLACE, true );
      else if ( update == HYTEG_HIP_ADD )
         P2_LAUNCH_CLASS_ROWS( HYTEG_HIP_ADD, false );
      else
         P2_LAUNCH_CLASS_ROWS( HYTEG_HIP_REPLACE, false );
#undef P2_LAUNCH_CLASS_ROWS
      HH_CHECK_HIP( hipGetLastError() );
      return HYTEG_HIP_OK;
   }
   if ( rows )
   {
      // inner DoFs by rows (p2_rows_body_dpp: every source row loaded once, 62 positions per wave; HYTEG_HIP_P2_ROWS_DPP=0 selects
      // p2_rows_body: every source loaded, 64 positions); the boundary DoFs, if asked for, in the same launch
      static const bool dpp = [] {
         const char* e = std::getenv( "HYTEG_HIP_P2_ROWS_DPP" );
         return !( e && e[0] == '0' );
      }();
      TileTable tt;
      const int rc = get_tiles( level, TILES_ROWS, dpp ? kRowsDppCapacity : 64, &tt );
      if ( rc != HYTEG_HIP_OK )
         return rc;
      P2RowsArgs R;
      R.F = F, R.tiles = tt.dev, R.ntiles = tt.count;
      const int n = F.N - 1;
      R.vbytes    = (unsigned) ( tet64( F.N ) * 8 );
      R.ebytes    = (unsigned) ( ( 6 * tet64( n ) + tet64( n - 1 ) ) * 8 );
      unsigned rowBlocks = (unsigned) ( ( tt.count + kRowsWaves - 1 ) / kRowsWaves );
      R.xcd_chunk        = 0;
      static const bool xcdRows = [] {
         const char* e = std::getenv( "HYTEG_HIP_P2_XCD_ROWS" );
         return !( e && e[0] == '0' );
      }();
      if ( xcdRows && rowBlocks >= 64 )
      {
         R.xcd_chunk = (int) ( ( rowBlocks + 7 ) / 8 );
         rowBlocks   = 8u * (unsigned) R.xcd_chunk;
      }
      const unsigned shell = mask & HYTEG_HIP_MASK_SHELL;
      const int      nb    = shell ? nbx : 0;
      const dim3     block( kThreads );
#define P2_LAUNCH_ROWS( UPD, RES )                                                                                                            \
   do                                                                                                                                          \
   {                                                                                                                                           \
      if ( dpp )                                                                                                                               \
         hipLaunchKernelGGL( ( p2_rows_kernel< UPD, RES, true > ), dim3( rowBlocks ), block, 0, s, R.tiles, R.ntiles, R.xcd_chunk, R );      \
      else                                                                                                                                     \
         hipLaunchKernelGGL( ( p2_rows_kernel< UPD, RES, false > ), dim3( rowBlocks ), block, 0, s, R.tiles, R.ntiles, R.xcd_chunk, R );     \
   } while ( 0 )
#define P2_LAUNCH_FUSED( UPD )                                                                                                                 \
   do                                                                                                                                          \
   {                                                                                                                                           \
      if ( dpp )                                                                                                                               \
         hipLaunchKernelGGL( ( p2_apply_fused_kernel< UPD, true > ), dim3( 8 * nb + rowBlocks ), block, 0, s, R.tiles, R.ntiles, R.xcd_chunk, \
                             R, shell, nb );                                                                                                   \
      else                                                                                                                                     \
         hipLaunchKernelGGL( ( p2_apply_fused_kernel< UPD, false > ), dim3( 8 * nb + rowBlocks ), block, 0, s, R.tiles, R.ntiles,            \
                             R.xcd_chunk, R, shell, nb );                                                                                      \
   } while ( 0 )
      if ( kind_mask != 0xFFu )
      {
         // some kinds only: the boundary DoFs (their kernel skips the other kinds) in their own launch, the rows restricted
         if ( nb )
         {
            P2ClassArgs B;
            B.F = F, B.mask = shell;
            hipLaunchKernelGGL( p2_boundary_kernel, dim3( (unsigned) nbx, 8 ), dim3( kThreads ), 0, s, B );
         }
         if ( update == HYTEG_HIP_ADD )
            P2_LAUNCH_ROWS( HYTEG_HIP_ADD, true );
         else
            P2_LAUNCH_ROWS( HYTEG_HIP_REPLACE, true );
      }
      else if ( nb == 0 )
      {
         if ( update == HYTEG_HIP_ADD )
            P2_LAUNCH_ROWS( HYTEG_HIP_ADD, false );
         else
            P2_LAUNCH_ROWS( HYTEG_HIP_REPLACE, false );
      }
      else if ( update == HYTEG_HIP_ADD )
         P2_LAUNCH_FUSED( HYTEG_HIP_ADD );
      else
         P2_LAUNCH_FUSED( HYTEG_HIP_REPLACE );
#undef P2_LAUNCH_ROWS
#undef P2_LAUNCH_FUSED
      HH_CHECK_HIP( hipGetLastError() );
      return HYTEG_HIP_OK;
   }
   if ( mask & HYTEG_HIP_MASK_INNER )
   {
      // inner DoFs: compile-time stencils, one thread per DoF, all destination kinds in one launch
      const int64_t largest = tet64( F.N );
      hipLaunchKernelGGL( p2_inner_kernel, dim3( (unsigned) ( ( largest + kThreads - 1 ) / kThreads ), 8 ), dim3( kThreads ), 0, s, F );
   }
   if ( ( mask & HYTEG_HIP_MASK_SHELL ) && level >= 2 )
   {
      // DoFs on the macro-cell boundary: per-class constant stencils
      P2ClassArgs B;
      B.F = F, B.mask = mask & HYTEG_HIP_MASK_SHELL;
      hipLaunchKernelGGL( p2_boundary_kernel, dim3( (unsigned) nbx, 8 ), dim3( kThreads ), 0, s, B );
   }
   else if ( mask & HYTEG_HIP_MASK_SHELL )
   {
      // levels 0 and 1: a DoF can be next to several macro-faces at once; micro-cell by micro-cell gather in the reference's order
      P2Args A;
      A.dstV = dst_vertex, A.dstE = dst_edge, A.srcV = src_vertex, A.srcE = src_edge, A.elmat = optable_dev, A.alpha = alpha;
      A.N = ( 1 << level ) + 1, A.update = update, A.mask = mask & HYTEG_HIP_MASK_SHELL, A.kinds = kind_mask, A.T = tables();
      const int     faces = 4 * tri( A.N ); // candidates of the widest kind
      constexpr int G     = 8;
      hipLaunchKernelGGL( p2_elementwise_kernel< G >, dim3( (unsigned) ( ( faces + kThreads / G - 1 ) / ( kThreads / G ) ), 8 ), dim3( kThreads ), 0,
                          s, A );
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}


// ---- Gauss-Seidel / SOR on the macro-edges and macro-faces shared between macro-cells, in the reference's order ------------------
// P2ConstantOperator::smooth_sor (src/constant_stencil_operator/P2ConstantOperator.cpp:1267-1330) sweeps macro-vertices, -edges,
// -faces, -cells one class after the other; a primitive's sweep sees current values on itself and its boundary and ghost-layer
// values for everything else.  Cell-centric form (as for P1, p1_sor_shell.hip): the ghost-layer part of every row is ONE apply
// with the operator table whose weights for sources ON the primitive's closure are zeroed (summed over the cells by the additive
// exchange); the closure part is evaluated with the complementary tables, and the only sequential piece -- the edge DoFs inside
// a macro-face, which couple with each other -- is swept on every cell's copy with the total weights by the kernel below.
} // extern "C"
namespace {
// faces of the cell (bit g; 0: z = 0, 1: y = 0, 2: x = 0, 3: x + y + z = n) that contain the macro-primitive of point class cls
int class_face_flags( int cls )
{
   static const int edges[6] = { 0x3, 0x5, 0x9, 0x6, 0xA, 0xC }, verts[4] = { 0x7, 0xB, 0xD, 0xE };
   return cls < 6 ? edges[cls] : ( cls < 10 ? 1 << ( cls - 6 ) : verts[cls - 10] );
}
int plane_fn( int g, const int* p ) { return g == 0 ? p[2] : ( g == 1 ? p[1] : ( g == 2 ? p[0] : -( p[0] + p[1] + p[2] ) ) ); }
// the micro-vertices a DoF of a kind sits on, relative to its logical index (vertex DoF: one; edge DoF: its two end points)
int kind_points( int kind, int pts[2][3] )
{
   static const int ends[7][2][3] = { { { 0, 0, 0 }, { 1, 0, 0 } }, { { 0, 0, 0 }, { 0, 1, 0 } }, { { 0, 0, 0 }, { 0, 0, 1 } },
                                      { { 1, 0, 0 }, { 0, 1, 0 } }, { { 1, 0, 0 }, { 0, 0, 1 } }, { { 0, 1, 0 }, { 0, 0, 1 } },
                                      { { 0, 1, 0 }, { 1, 0, 1 } } };
   if ( kind == 0 )
   {
      pts[0][0] = pts[0][1] = pts[0][2] = 0;
      return 1;
   }
   for ( int e = 0; e < 2; ++e )
      for ( int r = 0; r < 3; ++r )
         pts[e][r] = ends[kind - 1][e][r];
   return 2;
}
// does the source DoF (kind K at offset d from a destination DoF of kind C and point class cls) lie on the closure of the
// destination's macro-primitive, i.e. on every cell face that contains it?  A question about offsets only.
bool source_on_closure( int C, int cls, int K, int dx, int dy, int dz )
{
   int       pd[2][3], ps[2][3];
   const int nd = kind_points( C, pd ), ns = kind_points( K, ps ), flags = class_face_flags( cls );
   for ( int g = 0; g < 4; ++g )
   {
      if ( !( ( flags >> g ) & 1 ) )
         continue;
      const int h = plane_fn( g, pd[0] );
      for ( int e = 1; e < nd; ++e )
         if ( plane_fn( g, pd[e] ) != h )
            return false; // a DoF of this kind cannot lie on that face at all: the class row is never used
      for ( int e = 0; e < ns; ++e )
      {
         const int q[3] = { dx + ps[e][0], dy + ps[e][1], dz + ps[e][2] };
         if ( plane_fn( g, q ) != h )
            return false;
      }
   }
   return true;
}

// a macro-face in the cell's index space: its vertices in the order of their global ids are the cell-local vertices l0, l1, l2;
// micro-vertex (i, j) of the face = O + i a + j b; edge DoF types of the face: X (i,j)-(i+1,j), XY (i+1,j)-(i,j+1), Y (i,j)-(i,j+1)
struct P2FaceFrame
{
   int    O[3], a[3], b[3]; // O is filled per level (n * unit vector of l0)
   int    kind[3];          // cell edge-DoF kind (1..6) of the face types X, XY, Y
   int    off[3][3];        // logical index of face edge (t, i, j) in the cell = O + i a + j b + off[t]
   double w[3][5];          // diagonal, then the four in-face neighbours of kFaceNb
};
// in-face neighbours of an edge DoF: (type, di, dj), the other edges of the two face triangles that share it
const int kFaceNbHost[3][4][3] = { { { 1, 0, 0 }, { 2, 0, 0 }, { 1, 0, -1 }, { 2, 1, -1 } },
                                   { { 0, 0, 0 }, { 2, 0, 0 }, { 0, 0, 1 }, { 2, 1, 0 } },
                                   { { 0, 0, 0 }, { 1, 0, 0 }, { 0, -1, 1 }, { 1, -1, 0 } } };
__constant__ int kFaceNb[3][4][3] = { { { 1, 0, 0 }, { 2, 0, 0 }, { 1, 0, -1 }, { 2, 1, -1 } },
                                      { { 0, 0, 0 }, { 2, 0, 0 }, { 0, 0, 1 }, { 2, 1, 0 } },
                                      { { 0, 0, 0 }, { 1, 0, 0 }, { 0, -1, 1 }, { 1, -1, 0 } } };
bool face_frame( const int lv[3], P2FaceFrame& F, int& faceClass )
{
   static const int unit[4][3] = { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
   static const int dirs[6][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 }, { -1, 1, 0 }, { -1, 0, 1 }, { 0, -1, 1 } };
   static const int e0[6][3]   = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 1, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 } };
   for ( int k = 0; k < 3; ++k )
      if ( lv[k] < 0 || lv[k] > 3 )
         return false;
   if ( lv[0] == lv[1] || lv[0] == lv[2] || lv[1] == lv[2] )
      return false;
   const int missing = 6 - lv[0] - lv[1] - lv[2];
   faceClass         = 6 + ( missing == 3 ? 0 : ( missing == 2 ? 1 : ( missing == 1 ? 2 : 3 ) ) );
   for ( int r = 0; r < 3; ++r )
   {
      F.O[r] = unit[lv[0]][r]; // scaled by n by the caller
      F.a[r] = unit[lv[1]][r] - unit[lv[0]][r];
      F.b[r] = unit[lv[2]][r] - unit[lv[0]][r];
   }
   for ( int t = 0; t < 3; ++t )
   {
      int D[3], S[3]; // direction and start point (relative to micro-vertex (i, j)) of face type t
      for ( int r = 0; r < 3; ++r )
      {
         D[r] = t == 0 ? F.a[r] : ( t == 1 ? F.b[r] - F.a[r] : F.b[r] );
         S[r] = t == 1 ? F.a[r] : 0;
      }
      F.kind[t] = 0;
      for ( int k = 0; k < 6; ++k )
      {
         const bool plus  = D[0] == dirs[k][0] && D[1] == dirs[k][1] && D[2] == dirs[k][2];
         const bool minus = D[0] == -dirs[k][0] && D[1] == -dirs[k][1] && D[2] == -dirs[k][2];
         if ( !plus && !minus )
            continue;
         F.kind[t] = k + 1;
         for ( int r = 0; r < 3; ++r )
            F.off[t][r] = ( plus ? S[r] : S[r] + D[r] ) - e0[k][r];
      }
      if ( F.kind[t] == 0 )
         return false;
   }
   return true;
}

struct P2FaceSorArgs
{
   double*       u;
   const double* q;
   P2FaceFrame   F[4];
   unsigned      mask;
   int           N, backwards;
   double        relax;
};
__device__ inline int64_t face_edge_index( const P2FaceFrame& F, int n, int t, int i, int j )
{
   const int x = F.O[0] + i * F.a[0] + j * F.b[0] + F.off[t][0], y = F.O[1] + i * F.a[1] + j * F.b[1] + F.off[t][1],
             z = F.O[2] + i * F.a[2] + j * F.b[2] + F.off[t][2];
   return edge_block_start( n, F.kind[t] ) + cell_index( n, x, y, z );
}
// P2::macroface::generated::sor_3D_macroface_P2_update_edgedofs[_backwards] on this cell's copy of the face: rows ascending, x
// ascending, at every index X, XY, Y in place (backwards: everything reversed).  The order only matters between coupled DoFs,
// and the stage 3 ( x + 2 y ) + type puts every DoF after the neighbours the loop visits before it and before the others: one
// workgroup per face walks the stages, all DoFs of a stage at once.
__global__ __launch_bounds__( 256 ) void p2_sor_face_edges_kernel( const P2FaceSorArgs A )
{
   const int f = blockIdx.x;
   if ( !( ( A.mask >> ( 6 + f ) ) & 1u ) )
      return;
   const P2FaceFrame& F = A.F[f];
   const int          n = A.N - 1, stages = 3 * ( 2 * n - 1 );
   for ( int step = 0; step < stages; ++step )
   {
      const int s = A.backwards ? stages - 1 - step : step;
      const int t = s % 3, qq = s / 3;
      const int ylo = qq - n + 1 > 0 ? qq - n + 1 : 0, yhi = qq / 2;
      for ( int y = ylo + (int) threadIdx.x; y <= yhi; y += (int) blockDim.x )
      {
         const int  x     = qq - 2 * y;
         const bool inner = t == 0 ? y >= 1 : ( t == 1 ? x + y <= n - 2 : x >= 1 );
         if ( !inner || x + y > n - 1 )
            continue;
         const int64_t i   = face_edge_index( F, n, t, x, y );
         double        sum = A.q[i];
#pragma unroll
         for ( int k = 0; k < 4; ++k )
            sum -= F.w[t][1 + k] * A.u[face_edge_index( F, n, kFaceNb[t][k][0], x + kFaceNb[t][k][1], y + kFaceNb[t][k][2] )];
         A.u[i] = ( 1.0 - A.relax ) * A.u[i] + A.relax / F.w[t][0] * sum;
      }
      __syncthreads();
   }
}
// the same with the face's edge DoFs staged in LDS in the FACE's layout -- type t, row j, position i at t tri(n) + row_start(n, j) + i --
// (levels <= 6: 3 tri(64) doubles = 50 KB): a stage then costs an LDS round trip and a barrier instead of dependent global loads
// (48 us per call at level 3 with the kernel above, where 45 stages move a few hundred values)
__device__ inline int face_lds_index( int n, int t, int i, int j ) { return t * tri( n ) + row_start( n, j ) + i; }
__device__ inline void p2_sor_face_edges_lds_body( double* u, const double* q, const P2FaceFrame& F, int N, int backwards, double relax )
{
   extern __shared__ double lu[]; // [3][tri(n)]
   const int n = N - 1, T = tri( n ), stages = 3 * ( 2 * n - 1 );
   // every edge DoF of the face plane (inner ones and those on its boundary edges): (t, i, j) with i + j <= n - 1
   for ( int e = threadIdx.x; e < 3 * T; e += (int) blockDim.x )
   {
      const int t = e / T, r = e - t * T;
      const int j = row_of( n, r ), i = r - row_start( n, j );
      lu[e]       = u[face_edge_index( F, n, t, i, j )];
   }
   __syncthreads();
   for ( int step = 0; step < stages; ++step )
   {
      const int s = backwards ? stages - 1 - step : step;
      const int t = s % 3, qq = s / 3;
      const int ylo = qq - n + 1 > 0 ? qq - n + 1 : 0, yhi = qq / 2;
      for ( int y = ylo + (int) threadIdx.x; y <= yhi; y += (int) blockDim.x )
      {
         const int  x     = qq - 2 * y;
         const bool inner = t == 0 ? y >= 1 : ( t == 1 ? x + y <= n - 2 : x >= 1 );
         if ( !inner || x + y > n - 1 )
            continue;
         const int l   = face_lds_index( n, t, x, y );
         double    sum = q[face_edge_index( F, n, t, x, y )];
#pragma unroll
         for ( int k = 0; k < 4; ++k )
            sum -= F.w[t][1 + k] * lu[face_lds_index( n, kFaceNb[t][k][0], x + kFaceNb[t][k][1], y + kFaceNb[t][k][2] )];
         lu[l] = ( 1.0 - relax ) * lu[l] + relax / F.w[t][0] * sum;
      }
      __syncthreads();
   }
   for ( int e = threadIdx.x; e < 3 * T; e += (int) blockDim.x )
   {
      const int  t = e / T, r = e - t * T;
      const int  j = row_of( n, r ), i = r - row_start( n, j );
      const bool inner = t == 0 ? j >= 1 : ( t == 1 ? i + j <= n - 2 : i >= 1 );
      if ( inner )
         u[face_edge_index( F, n, t, i, j )] = lu[e];
   }
}
__global__ __launch_bounds__( 256 ) void p2_sor_face_edges_lds_kernel( const P2FaceSorArgs A )
{
   const int f = blockIdx.x;
   if ( !( ( A.mask >> ( 6 + f ) ) & 1u ) )
      return;
   p2_sor_face_edges_lds_body( A.u, A.q, A.F[f], A.N, A.backwards, A.relax );
}
// up to HYTEG_HIP_MAX_BATCH macro-cells in one launch (blockIdx.y = cell): the cells' face frames (with the faces' total weights) come
// from a device table the caller built once per level (hyteg_hip_p2_sor_face_frames)
struct P2FaceSorBatchArgs
{
   double*            u[HYTEG_HIP_MAX_BATCH];
   const double*      q[HYTEG_HIP_MAX_BATCH];
   unsigned           mask[HYTEG_HIP_MAX_BATCH];
   const P2FaceFrame* frames; // [cell][4]
   int                N, backwards;
   double             relax;
};
__global__ __launch_bounds__( 256 ) void p2_sor_face_edges_lds_batch_kernel( const P2FaceSorBatchArgs A )
{
   const int f = blockIdx.x, cell = blockIdx.y;
   if ( !( ( A.mask[cell] >> ( 6 + f ) ) & 1u ) )
      return;
   __shared__ P2FaceFrame F;
   if ( threadIdx.x == 0 )
      F = A.frames[4 * cell + f];
   __syncthreads();
   p2_sor_face_edges_lds_body( A.u[cell], A.q[cell], F, A.N, A.backwards, A.relax );
}
} // namespace
extern "C" {

HYTEG_HIP_API int hyteg_hip_p2_elementwise_apply_cells_kinds( int ncells, double* const* dst_vertex, double* const* dst_edge, const double* const* src_vertex,
                                                              const double* const* src_edge, int level, const double* const* optables_dev, double alpha,
                                                              int update, const unsigned* masks, unsigned kind_mask, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst_vertex && dst_edge && src_vertex && src_edge && optables_dev && masks, "p2_elementwise_apply_cells_kinds: null pointer" );
   HH_REQUIRE( ncells >= 1 && ncells <= HYTEG_HIP_MAX_BATCH, "p2_elementwise_apply_cells_kinds: 1 <= ncells <= HYTEG_HIP_MAX_BATCH" );
   HH_REQUIRE( level >= 2 && level <= 6, "p2_elementwise_apply_cells_kinds: levels 2..6 (below: the micro-cell gather per cell; above: the row kernels per cell)" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p2_elementwise_apply_cells_kinds: bad update" );
   kind_mask &= 0xFFu;
   if ( kind_mask == 0 )
      return HYTEG_HIP_OK;
   P2BatchPtrs P{};
   unsigned    any = 0;
   for ( int c = 0; c < ncells; ++c )
   {
      HH_REQUIRE( dst_vertex[c] && dst_edge[c] && src_vertex[c] && src_edge[c] && optables_dev[c], "p2_elementwise_apply_cells_kinds: null array" );
      HH_REQUIRE( dst_vertex[c] != src_vertex[c] && dst_edge[c] != src_edge[c], "p2_elementwise_apply_cells_kinds: dst and src must differ" );
      P.dstV[c] = dst_vertex[c], P.dstE[c] = dst_edge[c], P.srcV[c] = src_vertex[c], P.srcE[c] = src_edge[c], P.table[c] = optables_dev[c];
      P.mask[c] = masks[c] & HYTEG_HIP_MASK_ALL;
      any |= P.mask[c];
   }
   if ( any == 0 )
      return HYTEG_HIP_OK;
   P2FastArgs F{};
   F.alpha = alpha, F.N = ( 1 << level ) + 1, F.update = update, F.kinds = kind_mask;
   hipStream_t s = as_stream( stream );
   if ( level >= class_rows_min_level().load( std::memory_order_relaxed ) && ( kind_mask == 0xFFu || level >= 6 ) )
   {
      // row waves for the inner DoFs and every boundary class of every cell, one launch (all kinds; a kind mask from level 6, as above)
      TileTable tt;
      const int rc = get_class_rows_tiles( level, 62, &tt );
      if ( rc != HYTEG_HIP_OK )
         return rc;
      P2RowsArgs R;
      R.F = F, R.tiles = tt.dev, R.ntiles = tt.count, R.xcd_chunk = 0;
      const int n = F.N - 1;
      R.vbytes    = (unsigned) ( tet64( F.N ) * 8 );
      R.ebytes    = (unsigned) ( ( 6 * tet64( n ) + tet64( n - 1 ) ) * 8 );
      const dim3 grid( (unsigned) ( ( tt.count + kClassRowsWaves - 1 ) / kClassRowsWaves ), (unsigned) ncells );
#define P2_LAUNCH_CLASS_ROWS( UPD, RES ) \
   hipLaunchKernelGGL( ( p2_class_rows_batch_kernel< UPD, RES > ), grid, dim3( 64 * kClassRowsWaves ), 0, s, R.tiles, R.ntiles, R, P )
      if ( kind_mask != 0xFFu && update == HYTEG_HIP_ADD )
         P2_LAUNCH_CLASS_ROWS( HYTEG_HIP_ADD, true );
      else if ( kind_mask != 0xFFu )
         P2_LAUNCH_CLASS_ROWS( HYTEG_HIP_REPLACE, true );
      else if ( update == HYTEG_HIP_ADD )
         P2_LAUNCH_CLASS_ROWS( HYTEG_HIP_ADD, false );
      else
         P2_LAUNCH_CLASS_ROWS( HYTEG_HIP_REPLACE, false );
#undef P2_LAUNCH_CLASS_ROWS
      HH_CHECK_HIP( hipGetLastError() );
      return HYTEG_HIP_OK;
   }
   if ( any & HYTEG_HIP_MASK_INNER )
   {
      const int64_t largest = tet64( F.N );
      hipLaunchKernelGGL( p2_inner_batch_kernel, dim3( (unsigned) ( ( largest + kThreads - 1 ) / kThreads ), 8, (unsigned) ncells ), dim3( kThreads ), 0, s, F, P );
   }
   if ( any & HYTEG_HIP_MASK_SHELL )
   {
      const int nbx = ( 4 * tri( F.N ) + kThreads - 1 ) / kThreads;
      hipLaunchKernelGGL( p2_boundary_batch_kernel, dim3( (unsigned) nbx, 8, (unsigned) ncells ), dim3( kThreads ), 0, s, F, P );
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2_operator_table_closure_split( const double* table_host, double* outside, double* closure_vertex, double* closure_edge )
{
   HH_REQUIRE( table_host && outside && closure_vertex && closure_edge, "p2_operator_table_closure_split: null pointer" );
   for ( int k = 0; k < kOperatorTableSize; ++k )
      outside[k] = closure_vertex[k] = closure_edge[k] = 0.0;
   for ( int c = 0; c < 8; ++c )
   {
      const KindStencil S = build_kind_stencil( c );
      for ( int cls = 0; cls < 14; ++cls )
         for ( int q = 0; q < S.n; ++q )
         {
            const int    at = class_offset( c ) + cls * S.n + q;
            const double w  = table_host[at];
            if ( source_on_closure( c, cls, S.kind[q], S.dx[q], S.dy[q], S.dz[q] ) )
               ( S.kind[q] == 0 ? closure_vertex : closure_edge )[at] = w;
            else
               outside[at] = w;
         }
   }
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2_operator_table_face_edge_weights( const double* table_host, const int* face_verts, double* w )
{
   HH_REQUIRE( table_host && face_verts && w, "p2_operator_table_face_edge_weights: null pointer" );
   P2FaceFrame F;
   int         cls = 0;
   HH_REQUIRE( face_frame( face_verts, F, cls ), "p2_operator_table_face_edge_weights: face_verts must be three different cell-local vertex ids" );
   for ( int t = 0; t < 3; ++t )
   {
      const KindStencil S = build_kind_stencil( F.kind[t] );
      for ( int k = 0; k < 5; ++k )
      {
         int kind = F.kind[t], d[3] = { 0, 0, 0 };
         if ( k > 0 )
         {
            const int* nb = kFaceNbHost[t][k - 1];
            kind          = F.kind[nb[0]];
            for ( int r = 0; r < 3; ++r )
               d[r] = nb[1] * F.a[r] + nb[2] * F.b[r] + F.off[nb[0]][r] - F.off[t][r];
         }
         double v = 0.0;
         bool   found = false;
         for ( int q = 0; q < S.n && !found; ++q )
            if ( S.kind[q] == kind && S.dx[q] == d[0] && S.dy[q] == d[1] && S.dz[q] == d[2] )
               v = table_host[class_offset( F.kind[t] ) + ( cls - 0 ) * S.n + q], found = true;
         HH_REQUIRE( found, "p2_operator_table_face_edge_weights: a face neighbour is not in the stencil list (internal error)" );
         w[5 * t + k] = v;
      }
   }
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2_sor_face_edgedofs_cell( double* dst_edge, const double* q_edge, int level, const int* face_verts, const double* face_w,
                                                       double relax, unsigned mask, int backwards, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst_edge && q_edge && face_verts && face_w, "p2_sor_face_edgedofs_cell: null pointer" );
   HH_REQUIRE( level >= 2 && level <= HYTEG_HIP_P2_MAX_LEVEL, "p2_sor_face_edgedofs_cell: level out of range (2..9)" );
   HH_REQUIRE( dst_edge != q_edge, "p2_sor_face_edgedofs_cell: dst and q must differ" );
   mask &= 0xFu << 6;
   if ( mask == 0 )
      return HYTEG_HIP_OK;
   P2FaceSorArgs A;
   A.u = dst_edge, A.q = q_edge, A.mask = mask, A.N = ( 1 << level ) + 1, A.backwards = backwards ? 1 : 0, A.relax = relax;
   const int n = A.N - 1;
   for ( int f = 0; f < 4; ++f )
   {
      int cls = 0;
      if ( !( ( mask >> ( 6 + f ) ) & 1u ) )
      {
         A.F[f] = P2FaceFrame{};
         continue;
      }
      HH_REQUIRE( face_frame( face_verts + 3 * f, A.F[f], cls ) && cls == 6 + f,
                  "p2_sor_face_edgedofs_cell: face_verts[f] must be the three cell-local vertex ids of face f" );
      for ( int r = 0; r < 3; ++r )
         A.F[f].O[r] *= n;
      for ( int t = 0; t < 3; ++t )
      {
         HH_REQUIRE( face_w[15 * f + 5 * t] != 0.0, "p2_sor_face_edgedofs_cell: zero diagonal weight" );
         for ( int k = 0; k < 5; ++k )
            A.F[f].w[t][k] = face_w[15 * f + 5 * t + k];
      }
   }
   const size_t lds = (size_t) 3 * tri( n ) * sizeof( double );
   if ( level <= 6 )
   {
      if ( lds > 48 * 1024 )
         HH_CHECK_HIP( hipFuncSetAttribute( reinterpret_cast< const void* >( p2_sor_face_edges_lds_kernel ), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int) lds ) );
      hipLaunchKernelGGL( p2_sor_face_edges_lds_kernel, dim3( 4 ), dim3( 256 ), lds, as_stream( stream ), A );
   }
   else
      hipLaunchKernelGGL( p2_sor_face_edges_kernel, dim3( 4 ), dim3( 256 ), 0, as_stream( stream ), A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API size_t hyteg_hip_p2_sor_face_frames_bytes( void ) { return 4 * sizeof( P2FaceFrame ); }
HYTEG_HIP_API int    hyteg_hip_p2_sor_face_frames( int level, const int* face_verts, const double* face_w, void* frames_host )
{
   HH_REQUIRE( face_verts && face_w && frames_host, "p2_sor_face_frames: null pointer" );
   HH_REQUIRE( level >= 2 && level <= HYTEG_HIP_P2_MAX_LEVEL, "p2_sor_face_frames: level out of range (2..9)" );
   P2FaceFrame* F = static_cast< P2FaceFrame* >( frames_host );
   const int    n = 1 << level;
   for ( int f = 0; f < 4; ++f )
   {
      int cls = 0;
      F[f]    = P2FaceFrame{};
      HH_REQUIRE( face_frame( face_verts + 3 * f, F[f], cls ) && cls == 6 + f, "p2_sor_face_frames: face_verts[f] must be the three cell-local vertex ids of face f" );
      for ( int r = 0; r < 3; ++r )
         F[f].O[r] *= n;
      for ( int t = 0; t < 3; ++t )
         for ( int k = 0; k < 5; ++k )
            F[f].w[t][k] = face_w[15 * f + 5 * t + k];
   }
   return HYTEG_HIP_OK;
}
HYTEG_HIP_API int hyteg_hip_p2_sor_face_edgedofs_cells( int ncells, double* const* dst_edge, const double* const* q_edge, int level, const void* frames_dev,
                                                        double relax, const unsigned* masks, int backwards, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst_edge && q_edge && frames_dev && masks, "p2_sor_face_edgedofs_cells: null pointer" );
   HH_REQUIRE( ncells >= 1 && ncells <= HYTEG_HIP_MAX_BATCH, "p2_sor_face_edgedofs_cells: 1 <= ncells <= HYTEG_HIP_MAX_BATCH" );
   HH_REQUIRE( level >= 2 && level <= 6, "p2_sor_face_edgedofs_cells: levels 2..6 (the face's edge DoFs are staged in LDS)" );
   P2FaceSorBatchArgs A{};
   A.frames = static_cast< const P2FaceFrame* >( frames_dev ), A.N = ( 1 << level ) + 1, A.backwards = backwards ? 1 : 0, A.relax = relax;
   unsigned any = 0;
   for ( int c = 0; c < ncells; ++c )
   {
      HH_REQUIRE( dst_edge[c] && q_edge[c] && dst_edge[c] != q_edge[c], "p2_sor_face_edgedofs_cells: null array, or dst and q are the same" );
      A.u[c] = dst_edge[c], A.q[c] = q_edge[c], A.mask[c] = masks[c] & ( 0xFu << 6 );
      any |= A.mask[c];
   }
   if ( any == 0 )
      return HYTEG_HIP_OK;
   const size_t lds = (size_t) 3 * tri( A.N - 1 ) * sizeof( double );
   if ( lds > 48 * 1024 )
      HH_CHECK_HIP( hipFuncSetAttribute( reinterpret_cast< const void* >( p2_sor_face_edges_lds_batch_kernel ), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int) lds ) );
   hipLaunchKernelGGL( p2_sor_face_edges_lds_batch_kernel, dim3( 4, (unsigned) ncells ), dim3( 256 ), lds, as_stream( stream ), A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

} // extern "C"
