// Batched forms of the P1 kernels: ONE launch covers all macro-cells a rank owns (grid.y = cell) and both the inner and
// the boundary points of every cell (the 15-bit point mask selects, the point's class selects the stencil).
// They exist for the coarse and middle levels of a multi-cell V-cycle, where a level-2..6 cell has 35..48k points and
// the per-cell kernels of the other files spend their time in launch latency: regular_octahedron_8el needed ~5,500
// launches of 2-6 us kernels per V(3,3) cycle.  The fine levels keep the tuned per-cell kernels.
// One thread per array entry over FULL tiles; neighbours are direct global loads (the arrays of these levels live in L2).
#include <map>
#include <mutex>

#include "common.hpp"

using namespace hyteg_hip;

namespace {

constexpr int kTile    = 256; // one entry per thread: the batched levels are small, a thread looping over 4 entries pays 4 x the dependent-load latency
constexpr int kThreads = 256;
constexpr int kPer     = kTile / kThreads;
constexpr int kMaxB    = HYTEG_HIP_MAX_BATCH;

__constant__ int kOffsB[15][3] = { { 0, 0, -1 }, { 1, 0, -1 }, { -1, 1, -1 }, { 0, 1, -1 }, { 0, -1, 0 },
                                   { 1, -1, 0 }, { -1, 0, 0 }, { 0, 0, 0 },   { 1, 0, 0 },  { -1, 1, 0 },
                                   { 0, 1, 0 },  { 0, -1, 1 }, { 1, -1, 1 },  { -1, 0, 1 }, { 0, 0, 1 } };

// class of a point: 0..13 = slot of the macro-primitive it lies on, 14 = inner point (bit numbers of the point mask)
__device__ inline int point_class( int N, int x, int y, int z )
{
   const int f0 = ( z == 0 ), f1 = ( y == 0 ), f2 = ( x == 0 ), f3 = ( x + y + z == N - 1 );
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

struct Point
{
   int  i, x, y, z, cls;
   bool ok;
};
// u-th entry of this thread in tile tl
__device__ inline Point decode( const Tile& tl, int N, int u )
{
   Point     p;
   const int e = (int) threadIdx.x + u * kThreads;
   p.ok        = e < tl.cnt;
   p.i         = tl.a + ( p.ok ? e : tl.cnt - 1 );
   const int W = N - tl.z, j = p.i - slice_start( N, tl.z );
   p.z         = tl.z;
   p.y         = row_of( W, j );
   p.x         = j - row_start( W, p.y );
   p.cls       = point_class( N, p.x, p.y, p.z );
   return p;
}

template < int NP >
struct Cells
{
   double*  p[NP][kMaxB]; // p[0] = destination (or first operand), the others sources
   unsigned mask[kMaxB];
};

// ---- vector ops -------------------------------------------------------------------------------------------
struct VecB
{
   Cells< 1 + HYTEG_HIP_MAX_SRCS > c;
   double                          s[HYTEG_HIP_MAX_SRCS];
   const double*                   sp[HYTEG_HIP_MAX_SRCS]; // non-null: the coefficient is read from device memory
   const Tile*                     tiles;
   int                             N, nsrc, op; // op 0 assign, 1 add, 2 mult, 3 set constant s[0]
};

template < int NSRC >
__global__ __launch_bounds__( kThreads ) void batch_vector_kernel( const VecB A )
{
   const Tile     tl   = A.tiles[blockIdx.x];
   const int      cell = blockIdx.y;
   const unsigned mask = A.c.mask[cell];
   double*        dst  = A.c.p[0][cell];
   double         s[NSRC];
#pragma unroll
   for ( int k = 0; k < NSRC; ++k )
      s[k] = A.sp[k] ? *A.sp[k] : A.s[k];
#pragma unroll
   for ( int u = 0; u < kPer; ++u )
   {
      const Point p = decode( tl, A.N, u );
      if ( !p.ok || !( ( mask >> p.cls ) & 1u ) )
         continue;
      double tmp;
      if ( A.op == 3 )
         tmp = s[0];
      else if ( A.op == 2 )
      {
         tmp = A.c.p[1][cell][p.i];
#pragma unroll
         for ( int k = 1; k < NSRC; ++k )
            tmp *= A.c.p[1 + k][cell][p.i];
      }
      else
      {
         tmp = s[0] * A.c.p[1][cell][p.i];
#pragma unroll
         for ( int k = 1; k < NSRC; ++k )
            tmp += s[k] * A.c.p[1 + k][cell][p.i];
         if ( A.op == 1 )
            tmp = dst[p.i] + tmp;
      }
      dst[p.i] = tmp;
   }
}

// ---- dot ---------------------------------------------------------------------------------------------------
struct DotB
{
   Cells< 2 >  c;
   const Tile* tiles;
   int         N, ntiles, ncells;
   double*     partial;
   double*     result;
   unsigned*   counter; // zero between launches: the last workgroup to finish reduces the partial sums and resets it
   double*     cg;      // non-null: run phase `cgPhase` of the conjugate gradient recurrences after the reduction
   int         cgPhase;
   double      relTol, absTol;
};

// scalar recurrences of the conjugate gradient iteration (see hyteg_hip_cg_scalars in include/hyteg_hip.h)
__device__ inline void cg_scalars_update( double* s, int phase, double relTol, double absTol )
{
   const bool done = s[HYTEG_HIP_CG_DONE] != 0.0;
   if ( phase == 0 )
   {
      s[HYTEG_HIP_CG_PRSOLD]     = s[HYTEG_HIP_CG_RR];
      s[HYTEG_HIP_CG_RES_START]  = sqrt( s[HYTEG_HIP_CG_RR] );
      s[HYTEG_HIP_CG_DONE]       = s[HYTEG_HIP_CG_RES_START] < absTol ? 1.0 : 0.0;
      s[HYTEG_HIP_CG_ITERATIONS] = 0.0;
      s[HYTEG_HIP_CG_ONE]        = 1.0;
      s[HYTEG_HIP_CG_ALPHA] = s[HYTEG_HIP_CG_NEG_ALPHA] = s[HYTEG_HIP_CG_BETA] = 0.0;
   }
   else if ( phase == 1 )
   {
      const double alpha        = done ? 0.0 : s[HYTEG_HIP_CG_PRSOLD] / s[HYTEG_HIP_CG_PAP];
      s[HYTEG_HIP_CG_ALPHA]     = alpha;
      s[HYTEG_HIP_CG_NEG_ALPHA] = -alpha;
   }
   else if ( !done )
   {
      const double rsnew = s[HYTEG_HIP_CG_RR], sq = sqrt( rsnew );
      s[HYTEG_HIP_CG_ITERATIONS] += 1.0;
      if ( sq / s[HYTEG_HIP_CG_RES_START] < relTol || sq < absTol )
         s[HYTEG_HIP_CG_DONE] = 1.0;
      else
      {
         s[HYTEG_HIP_CG_BETA]   = rsnew / s[HYTEG_HIP_CG_PRSOLD];
         s[HYTEG_HIP_CG_PRSOLD] = rsnew;
      }
   }
}

__device__ inline double wave_sum_b( double v )
{
#pragma unroll
   for ( int off = 32; off > 0; off >>= 1 )
      v += __shfl_down( v, off, 64 );
   return v;
}
__device__ inline double block_sum_b( double v, double* sh )
{
   v = wave_sum_b( v );
   if ( ( threadIdx.x & 63 ) == 0 )
      sh[threadIdx.x >> 6] = v;
   __syncthreads();
   double r = 0.0;
   if ( threadIdx.x == 0 )
      for ( int k = 0; k < kThreads / 64; ++k )
         r += sh[k];
   return r;
}

// fixed (cell, tile) -> workgroup assignment and fixed reduction trees: the result does not depend on timing
__global__ __launch_bounds__( kThreads ) void batch_dot_kernel( const DotB A )
{
   __shared__ double sh[kThreads / 64];
   double            acc   = 0.0;
   const int         total = A.ntiles * A.ncells;
   for ( int w = blockIdx.x; w < total; w += gridDim.x )
   {
      const int      cell = w / A.ntiles;
      const Tile     tl   = A.tiles[w - cell * A.ntiles];
      const unsigned mask = A.c.mask[cell];
      const double * a = A.c.p[0][cell], *b = A.c.p[1][cell];
#pragma unroll
      for ( int u = 0; u < kPer; ++u )
      {
         const Point p = decode( tl, A.N, u );
         if ( p.ok && ( ( mask >> p.cls ) & 1u ) )
            acc = fma( a[p.i], b[p.i], acc );
      }
   }
   const double r = block_sum_b( acc, sh );
   if ( A.counter == nullptr )
   {
      // many workgroups: batch_dot_final_kernel reduces the partial sums (tickets on one address would serialise)
      if ( threadIdx.x == 0 )
         A.partial[blockIdx.x] = r;
      return;
   }
   // one launch: the workgroup that finishes last reduces the partial sums, in the same fixed order whichever it is
   __shared__ bool last;
   if ( threadIdx.x == 0 )
   {
      // written through to memory (agent-scope store: the XCDs do not share an L2) and acknowledged before the ticket is taken;
      // no release fence -- that is an L2 write-back per workgroup on this architecture (see dot_finish in p1_vector.hip)
      __hip_atomic_store( A.partial + blockIdx.x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
      last = __hip_atomic_fetch_add( A.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) == gridDim.x - 1;
   }
   __syncthreads();
   if ( !last )
      return;
   double sum = 0.0;
   for ( int k = threadIdx.x; k < (int) gridDim.x; k += kThreads )
      sum += __hip_atomic_load( A.partial + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
   __syncthreads(); // sh is reused
   const double grand = block_sum_b( sum, sh );
   if ( threadIdx.x == 0 )
   {
      *A.result = grand;
      __hip_atomic_store( A.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      if ( A.cg )
         cg_scalars_update( A.cg, A.cgPhase, A.relTol, A.absTol );
   }
}

__global__ __launch_bounds__( kThreads ) void batch_dot_final_kernel( const double* partial, int n, double* result, double* cg, int cgPhase,
                                                                       double relTol, double absTol )
{
   __shared__ double sh[kThreads / 64];
   double            acc = 0.0;
   for ( int k = threadIdx.x; k < n; k += kThreads )
      acc += partial[k];
   const double r = block_sum_b( acc, sh );
   if ( threadIdx.x == 0 )
   {
      *result = r;
      if ( cg )
         cg_scalars_update( cg, cgPhase, relTol, absTol );
   }
}

// ---- apply / Jacobi ------------------------------------------------------------------------------------------
struct ApplyB
{
   Cells< 4 >    c; // dst, src, rhs, invdiag
   const double* stencils; // [ncells][15 classes][15 weights]: classes 0..13 the cell's shares, 14 the inner stencil
   const Tile*   tiles;
   int           N, mode; // 0 replace, 1 add, 2 Jacobi phase 0, 3 Jacobi phase 1 (shell update after the exchange)
   double        relax;
};

// All fifteen values and weights are loaded before the first FMA; a neighbour outside the cell reads the point itself and
// is skipped by a select (same FMA chain as a loop that skips it, without one dependent memory round trip per neighbour).
// Neighbour indices by layout algebra from p.i, as in shell.hpp.
__device__ inline double stencil_sum( const double* __restrict__ w, const double* __restrict__ src, int N, const Point& p )
{
   const int W = N - p.z;
   double    v[15], c[15];
   bool      ok[15];
#pragma unroll
   for ( int k = 0; k < 15; ++k )
   {
      const int dx = kOffsB[k][0], dy = kOffsB[k][1], dz = kOffsB[k][2];
      const int nx = p.x + dx, ny = p.y + dy, nz = p.z + dz;
      ok[k]        = !( nx < 0 || ny < 0 || nz < 0 || nx + ny + nz > N - 1 );
      const int rowDelta   = dy == 0 ? 0 : ( dy > 0 ? ( W - p.y ) : -( W - p.y + 1 ) );
      const int sliceDelta = dz == 0 ? 0 : ( dz > 0 ? tri( W ) - ny : ny - tri( W + 1 ) );
      v[k]                 = src[ok[k] ? p.i + sliceDelta + rowDelta + dx : p.i];
      c[k]                 = w[k];
   }
   double acc = 0.0;
#pragma unroll
   for ( int k = 0; k < 15; ++k )
      acc = ok[k] ? fma( c[k], v[k], acc ) : acc;
   return acc;
}

__global__ __launch_bounds__( kThreads ) void batch_apply_kernel( const ApplyB A )
{
   const Tile     tl   = A.tiles[blockIdx.x];
   const int      cell = blockIdx.y;
   const unsigned mask = A.c.mask[cell];
   double*        dst  = A.c.p[0][cell];
   const double*  src  = A.c.p[1][cell];
   const double*  tbl  = A.stencils + (size_t) cell * 225;
#pragma unroll
   for ( int u = 0; u < kPer; ++u )
   {
      const Point p = decode( tl, A.N, u );
      if ( !p.ok || !( ( mask >> p.cls ) & 1u ) )
         continue;
      if ( A.mode == 3 )
      {
         // shell points after the shares were summed into dst: dst = src + relax * invdiag * ( rhs - dst )
         if ( p.cls != 14 )
            dst[p.i] = src[p.i] + A.relax * ( A.c.p[3][cell][p.i] * ( A.c.p[2][cell][p.i] - dst[p.i] ) );
         continue;
      }
      const double acc = stencil_sum( tbl + p.cls * 15, src, A.N, p );
      if ( A.mode == 0 )
         dst[p.i] = acc;
      else if ( A.mode == 1 )
         dst[p.i] = dst[p.i] + acc;
      else if ( p.cls == 14 )
         dst[p.i] = src[p.i] + A.relax * ( A.c.p[3][cell][p.i] * ( A.c.p[2][cell][p.i] - acc ) );
      else
         dst[p.i] = acc; // this cell's share, summed over cells before phase 1
   }
}

// ---- grid transfer ---------------------------------------------------------------------------------------------
struct TransferB
{
   Cells< 2 >    c;      // p[0] = written array, p[1] = read array
   const double* nncInv; // [ncells][14]
   const Tile*   tiles;  // FULL tiles of the written level
   int           Nc, update;
};

__constant__ int kNB14B[14][3] = { { -1, 0, 0 }, { -1, 0, 1 }, { -1, 1, -1 }, { -1, 1, 0 }, { 0, -1, 0 },
                                   { 0, -1, 1 }, { 0, 0, -1 }, { 0, 0, 1 },   { 0, 1, -1 }, { 0, 1, 0 },
                                   { 1, -1, 0 }, { 1, -1, 1 }, { 1, 0, -1 },  { 1, 0, 0 } };
__constant__ int kAxisB[8][3]  = { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 1, -1, 0 },
                                  { 0, 0, 1 }, { 1, 0, -1 }, { 0, 1, -1 }, { 1, -1, 1 } };
__constant__ int kLoFirstB[8] = { 1, 1, 1, 0, 1, 0, 0, 1 };

// same terms and summation order as p1_restrict_kernel (p1_transfer.hip)
__global__ __launch_bounds__( kThreads ) void batch_restrict_kernel( const TransferB A )
{
   const Tile     tl     = A.tiles[blockIdx.x];
   const int      cell   = blockIdx.y, Nc = A.Nc, Nf = 2 * Nc - 1;
   const unsigned mask   = A.c.mask[cell];
   double*        coarse = A.c.p[0][cell];
   const double*  fine   = A.c.p[1][cell];
   const double*  inv    = A.nncInv + cell * 14;
#pragma unroll
   for ( int u = 0; u < kPer; ++u )
   {
      const Point p = decode( tl, Nc, u );
      if ( !p.ok || !( ( mask >> p.cls ) & 1u ) )
         continue;
      // the fourteen fine values and their scalings are loaded before the first sum (a loop that skips the neighbours outside
      // the cell pays dependent round trips per neighbour); the sum is the same sequence of separately rounded terms
      const int centre = cell_index( Nf, 2 * p.x, 2 * p.y, 2 * p.z );
      double    fv[14], sc[14];
      bool      ok[14];
#pragma unroll
      for ( int k = 0; k < 14; ++k )
      {
         const int fx = 2 * p.x + kNB14B[k][0], fy = 2 * p.y + kNB14B[k][1], fz = 2 * p.z + kNB14B[k][2];
         ok[k]        = !( fx < 0 || fy < 0 || fz < 0 || fx + fy + fz > Nf - 1 );
         const int fc = ok[k] ? point_class( Nf, fx, fy, fz ) : 14;
         fv[k]        = fine[ok[k] ? cell_index( Nf, fx, fy, fz ) : centre];
         // products and sums rounded separately (no FMA contraction): the per-cell kernels of p1_transfer.hip do the same, so
         // that both give the same bits as the reference's scalar loops
         sc[k] = ( fc == 14 ? 1.0 : inv[fc] ) * 0.5;
      }
      double acc   = 0.0;
      bool   first = true;
#pragma unroll
      for ( int k = 0; k < 14; ++k )
      {
         const double term = mul_rn( sc[k], fv[k] );
         acc               = ok[k] ? ( first ? term : add_rn( acc, term ) ) : acc;
         first             = first && !ok[k];
      }
      const int    fc   = point_class( Nf, 2 * p.x, 2 * p.y, 2 * p.z );
      const double term = mul_rn( fc == 14 ? 1.0 : inv[fc], fine[cell_index( Nf, 2 * p.x, 2 * p.y, 2 * p.z )] );
      coarse[p.i]       = first ? term : add_rn( acc, term );
   }
}

// same terms and order as p1_prolongate_kernel; tiles of the FINE level, A.Nc = coarse width
__global__ __launch_bounds__( kThreads ) void batch_prolongate_kernel( const TransferB A )
{
   const Tile     tl     = A.tiles[blockIdx.x];
   const int      cell   = blockIdx.y, Nc = A.Nc, Nf = 2 * Nc - 1;
   const unsigned mask   = A.c.mask[cell];
   double*        fine   = A.c.p[0][cell];
   const double*  coarse = A.c.p[1][cell];
   const double*  inv    = A.nncInv + cell * 14;
#pragma unroll
   for ( int u = 0; u < kPer; ++u )
   {
      const Point p = decode( tl, Nf, u );
      if ( !p.ok || !( ( mask >> p.cls ) & 1u ) )
         continue;
      const double sc   = p.cls == 14 ? 1.0 : inv[p.cls];
      const int    code = ( p.x & 1 ) | ( ( p.y & 1 ) << 1 ) | ( ( p.z & 1 ) << 2 );
      const double old  = ( A.update == HYTEG_HIP_ADD && p.cls == 14 ) ? fine[p.i] : 0.0;
      double       v;
      if ( code == 0 )
         v = add_rn( old, mul_rn( sc, coarse[cell_index( Nc, p.x >> 1, p.y >> 1, p.z >> 1 )] ) );
      else
      {
         const int    ex = kAxisB[code][0], ey = kAxisB[code][1], ez = kAxisB[code][2];
         const double lo = coarse[cell_index( Nc, ( p.x - ex ) >> 1, ( p.y - ey ) >> 1, ( p.z - ez ) >> 1 )];
         const double hi = coarse[cell_index( Nc, ( p.x + ex ) >> 1, ( p.y + ey ) >> 1, ( p.z + ez ) >> 1 )];
         const double h  = sc * 0.5;
         const double tl = mul_rn( h, lo ), th = mul_rn( h, hi );
         v               = kLoFirstB[code] ? add_rn( add_rn( old, tl ), th ) : add_rn( add_rn( old, th ), tl );
      }
      fine[p.i] = v;
   }
}

inline bool batch_level_ok( int level ) { return level >= 0 && level <= HYTEG_HIP_MAX_LEVEL; }

// tiles that cover the whole array also exist for levels 0 and 1 (get_tiles builds them for any level)
int full_tiles( int level, TileTable* tt ) { return get_tiles( level, TILES_FULL, kTile, tt ); }

} // namespace

extern "C" {

#define BATCH_CHECKS( name )                                                                                   \
   HH_REQUIRE( ncells >= 1 && ncells <= kMaxB, name ": ncells must be 1..HYTEG_HIP_MAX_BATCH" );              \
   HH_REQUIRE( batch_level_ok( level ), name ": level out of range [0,11]" );                                 \
   HH_REQUIRE( masks != nullptr, name ": null masks" );

static int launch_vector_b( int                  op,
                            int                  ncells,
                            double* const*       dst,
                            int                  nsrc,
                            const double* const* srcs,
                            const double*        scalars,
                            const double* const* scalar_ptrs,
                            int                  level,
                            const unsigned*      masks,
                            hyteg_hip_stream_t   stream )
{
   BATCH_CHECKS( "p1_vector_cells" );
   HH_REQUIRE( op >= 0 && op <= 3 && dst, "p1_vector_cells: bad op or null dst" );
   HH_REQUIRE( op == 3 ? scalars != nullptr : ( nsrc >= 1 && nsrc <= HYTEG_HIP_MAX_SRCS && srcs ), "p1_vector_cells: bad sources" );
   HH_REQUIRE( op == 2 || op == 3 || scalars || scalar_ptrs, "p1_vector_cells: null scalars" );
   TileTable tt;
   int       rc = full_tiles( level, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   VecB A{};
   bool any = false;
   for ( int c = 0; c < ncells; ++c )
   {
      HH_REQUIRE( dst[c], "p1_vector_cells: null destination" );
      A.c.p[0][c]  = dst[c];
      A.c.mask[c]  = masks[c] & HYTEG_HIP_MASK_ALL;
      any          = any || A.c.mask[c];
      if ( op != 3 )
         for ( int k = 0; k < nsrc; ++k )
         {
            HH_REQUIRE( srcs[k * ncells + c], "p1_vector_cells: null source" );
            A.c.p[1 + k][c] = const_cast< double* >( srcs[k * ncells + c] );
         }
   }
   if ( !any || tt.count == 0 )
      return HYTEG_HIP_OK;
   if ( op == 3 )
      A.s[0] = scalars[0];
   else
      for ( int k = 0; k < nsrc; ++k )
      {
         A.s[k]  = scalars ? scalars[k] : 1.0;
         A.sp[k] = scalar_ptrs ? scalar_ptrs[k] : nullptr;
      }
   A.tiles = tt.dev, A.N = ( 1 << level ) + 1, A.nsrc = nsrc, A.op = op;
   const dim3 grid( tt.count, ncells );
   switch ( op == 3 ? 1 : nsrc )
   {
   case 1:
      hipLaunchKernelGGL( batch_vector_kernel< 1 >, grid, dim3( kThreads ), 0, as_stream( stream ), A );
      break;
   case 2:
      hipLaunchKernelGGL( batch_vector_kernel< 2 >, grid, dim3( kThreads ), 0, as_stream( stream ), A );
      break;
   case 3:
      hipLaunchKernelGGL( batch_vector_kernel< 3 >, grid, dim3( kThreads ), 0, as_stream( stream ), A );
      break;
   default:
      hipLaunchKernelGGL( batch_vector_kernel< 4 >, grid, dim3( kThreads ), 0, as_stream( stream ), A );
      break;
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_vector_cells( int                  op,
                                             int                  ncells,
                                             double* const*       dst,
                                             int                  nsrc,
                                             const double* const* srcs,
                                             const double*        scalars,
                                             int                  level,
                                             const unsigned*      masks,
                                             hyteg_hip_stream_t   stream )
{
   return launch_vector_b( op, ncells, dst, nsrc, srcs, scalars, nullptr, level, masks, stream );
}

HYTEG_HIP_API int hyteg_hip_p1_vector_cells_dev( int                  op,
                                                 int                  ncells,
                                                 double* const*       dst,
                                                 int                  nsrc,
                                                 const double* const* srcs,
                                                 const double* const* scalar_ptrs,
                                                 int                  level,
                                                 const unsigned*      masks,
                                                 hyteg_hip_stream_t   stream )
{
   HH_REQUIRE( ( op == 0 || op == 1 ) && scalar_ptrs, "p1_vector_cells_dev: op must be 0 (assign) or 1 (add), scalar pointers non-null" );
   for ( int k = 0; k < nsrc && k < HYTEG_HIP_MAX_SRCS; ++k )
      HH_REQUIRE( scalar_ptrs[k], "p1_vector_cells_dev: null scalar pointer" );
   return launch_vector_b( op, ncells, dst, nsrc, srcs, nullptr, scalar_ptrs, level, masks, stream );
}

namespace {
// scalar recurrences of the conjugate gradient iteration (CGSolver.hpp:91-140 of the reference keeps them on the host),
// one thread; s: HYTEG_HIP_CG_* slots
__global__ void cg_scalars_kernel( double* s, int phase, double relTol, double absTol )
{
   if ( threadIdx.x == 0 && blockIdx.x == 0 )
      cg_scalars_update( s, phase, relTol, absTol );
}
} // namespace

namespace {
// ---- conjugate gradients for problems that fit one workgroup ------------------------------------------------------------
// The coarsest level of a multigrid hierarchy (8 macro-cells at level 2: 280 array entries) needs ~17 CG iterations per
// cycle; as separate launches an iteration costs ~35 us of pure latency (apply, sum over shared copies, two reductions,
// three updates).  Here ONE workgroup runs the whole solve: p, A p and r live in LDS, an iteration is a handful of
// workgroup barriers.  Same recurrences as CGSolver.hpp:91-140 (identity preconditioner), same operator (per-cell shares
// of the stencil on shared points, summed over the copies of a point), same masks as the separate kernels.
constexpr int kCgThreads  = 1024;
constexpr int kCgMaxTotal = 4096; // entries of all cell arrays together
constexpr int kCgPer      = kCgMaxTotal / kCgThreads;

struct CgSmallArgs
{
   double*         x[kMaxB];
   const double*   b[kMaxB];
   unsigned        mask[kMaxB], owned[kMaxB];
   const double*   stencils;
   const int*      groupPtr[2];
   const int*      entryCell[2];
   const int*      entryOff[2];
   int             ngroups[2];
   int             N, cs, ncells, maxIter;
   double          relTol, absTol;
   double*         info; // [0] iterations, [1] sqrt(<r,r>) at exit
};

__device__ inline double cg_block_sum( double v, double* sh )
{
   v = wave_sum_b( v );
   __syncthreads(); // sh may still be read from the previous reduction
   if ( ( threadIdx.x & 63 ) == 0 )
      sh[threadIdx.x >> 6] = v;
   __syncthreads();
   double r = 0.0;
#pragma unroll
   for ( int k = 0; k < kCgThreads / 64; ++k )
      r += sh[k];
   return r; // the same value in every thread
}

__global__ __launch_bounds__( kCgThreads ) void p1_cg_small_kernel( const CgSmallArgs A )
{
   extern __shared__ double lds[];
   __shared__ double        sh[kCgThreads / 64];
   const int                total = A.ncells * A.cs;
   double *                 P = lds, *AP = lds + total, *R = lds + 2 * total;

   // this thread's entries: cell, coordinates, class, whether the flag selects them / the dot products count them
   int  idx[kCgPer], cell[kCgPer], px[kCgPer], py[kCgPer], pz[kCgPer], cls[kCgPer];
   bool sel[kCgPer], own[kCgPer];
#pragma unroll
   for ( int u = 0; u < kCgPer; ++u )
   {
      const int i = (int) threadIdx.x + u * kCgThreads;
      idx[u]      = i < total ? i : -1;
      sel[u] = own[u] = false;
      cell[u] = px[u] = py[u] = pz[u] = 0;
      cls[u]                           = 14;
      if ( i < total )
      {
         const int c = i / A.cs;
         int       j = i - c * A.cs, z = 0;
         while ( j >= tri( A.N - z ) )
         {
            j -= tri( A.N - z );
            ++z;
         }
         int y = 0;
         while ( j >= A.N - z - y )
         {
            j -= A.N - z - y;
            ++y;
         }
         cell[u] = c, px[u] = j, py[u] = y, pz[u] = z;
         cls[u] = point_class( A.N, j, y, z );
         sel[u] = ( A.mask[c] >> cls[u] ) & 1u;
         own[u] = ( A.owned[c] >> cls[u] ) & 1u;
      }
   }
   auto applyTo = [&]( const double* srcBase, bool srcGlobal ) {
      // AP = this cell's part of A * src on the selected points, then the sum over the copies of shared points
#pragma unroll
      for ( int u = 0; u < kCgPer; ++u )
         if ( idx[u] >= 0 && sel[u] )
         {
            Point p;
            p.i = idx[u] - cell[u] * A.cs, p.x = px[u], p.y = py[u], p.z = pz[u], p.cls = cls[u], p.ok = true;
            const double* src = srcGlobal ? A.x[cell[u]] : srcBase + cell[u] * A.cs;
            AP[idx[u]]        = stencil_sum( A.stencils + (size_t) cell[u] * 225 + cls[u] * 15, src, A.N, p );
         }
      __syncthreads();
      for ( int k = 0; k < 2; ++k )
         for ( int g = threadIdx.x; g < A.ngroups[k]; g += kCgThreads )
         {
            const int lo = A.groupPtr[k][g], hi = A.groupPtr[k][g + 1];
            double    s  = 0.0;
            for ( int e = lo; e < hi; ++e )
            {
               const double v = AP[A.entryCell[k][e] * A.cs + A.entryOff[k][e]];
               s              = e == lo ? v : s + v;
            }
            for ( int e = lo; e < hi; ++e )
               AP[A.entryCell[k][e] * A.cs + A.entryOff[k][e]] = s;
         }
      __syncthreads();
   };

   // r = b - A x ; p = r on the selected points, p = 0 elsewhere
   applyTo( nullptr, true );
   double rr = 0.0;
#pragma unroll
   for ( int u = 0; u < kCgPer; ++u )
      if ( idx[u] >= 0 )
      {
         const double r = sel[u] ? A.b[cell[u]][idx[u] - cell[u] * A.cs] - AP[idx[u]] : 0.0;
         R[idx[u]] = r, P[idx[u]] = r;
         if ( own[u] && sel[u] )
            rr = fma( r, r, rr );
      }
   rr                    = cg_block_sum( rr, sh );
   double       prsold   = rr;
   const double resStart = sqrt( rr );
   int          its      = 0;
   double       resNow   = resStart;
   __syncthreads();
   if ( !( resStart < A.absTol ) )
      for ( int it = 0; it < A.maxIter; ++it )
      {
         applyTo( P, false );
         double pap = 0.0;
#pragma unroll
         for ( int u = 0; u < kCgPer; ++u )
            if ( idx[u] >= 0 && own[u] && sel[u] )
               pap = fma( P[idx[u]], AP[idx[u]], pap );
         pap                = cg_block_sum( pap, sh );
         const double alpha = prsold / pap;
         double       rsnew = 0.0;
#pragma unroll
         for ( int u = 0; u < kCgPer; ++u )
            if ( idx[u] >= 0 && sel[u] )
            {
               double* xc = A.x[cell[u]] + ( idx[u] - cell[u] * A.cs );
               *xc        = *xc + alpha * P[idx[u]];
               const double r = R[idx[u]] + ( -alpha ) * AP[idx[u]];
               R[idx[u]]      = r;
               if ( own[u] )
                  rsnew = fma( r, r, rsnew );
            }
         rsnew  = cg_block_sum( rsnew, sh );
         its    = it + 1;
         resNow = sqrt( rsnew );
         if ( resNow / resStart < A.relTol || resNow < A.absTol )
            break;
         const double beta = rsnew / prsold;
#pragma unroll
         for ( int u = 0; u < kCgPer; ++u )
            if ( idx[u] >= 0 && sel[u] )
               P[idx[u]] = R[idx[u]] + beta * P[idx[u]];
         prsold = rsnew;
         __syncthreads();
      }
   if ( threadIdx.x == 0 && A.info )
   {
      A.info[0] = (double) its;
      A.info[1] = resNow;
   }
}
} // namespace

HYTEG_HIP_API int hyteg_hip_p1_cg_small_max_entries( void ) { return kCgMaxTotal; }

HYTEG_HIP_API int hyteg_hip_p1_cg_small_cells( int                  ncells,
                                               double* const*       x,
                                               const double* const* b,
                                               int                  level,
                                               const double*        stencils_dev,
                                               const unsigned*      masks,
                                               const unsigned*      owned_masks,
                                               const int* const*    group_ptr_dev,
                                               const int* const*    entry_cell_dev,
                                               const int* const*    entry_off_dev,
                                               const int*           ngroups,
                                               int                  max_iter,
                                               double               rel_tol,
                                               double               abs_tol,
                                               double*              info_dev,
                                               hyteg_hip_stream_t   stream )
{
   BATCH_CHECKS( "p1_cg_small_cells" );
   HH_REQUIRE( x && b && stencils_dev && owned_masks && ngroups, "p1_cg_small_cells: null pointer" );
   const int N = ( 1 << level ) + 1, cs = (int) tet64( N );
   HH_REQUIRE( (int64_t) ncells * cs <= kCgMaxTotal, "p1_cg_small_cells: the cell arrays together exceed hyteg_hip_p1_cg_small_max_entries()" );
   CgSmallArgs A{};
   for ( int c = 0; c < ncells; ++c )
   {
      HH_REQUIRE( x[c] && b[c], "p1_cg_small_cells: null array" );
      A.x[c] = x[c], A.b[c] = b[c];
      A.mask[c] = masks[c] & HYTEG_HIP_MASK_ALL, A.owned[c] = owned_masks[c] & HYTEG_HIP_MASK_ALL;
   }
   for ( int k = 0; k < 2; ++k )
   {
      A.ngroups[k] = ngroups[k];
      if ( ngroups[k] > 0 )
      {
         HH_REQUIRE( group_ptr_dev && entry_cell_dev && entry_off_dev && group_ptr_dev[k] && entry_cell_dev[k] && entry_off_dev[k],
                     "p1_cg_small_cells: null group table" );
         A.groupPtr[k] = group_ptr_dev[k], A.entryCell[k] = entry_cell_dev[k], A.entryOff[k] = entry_off_dev[k];
      }
   }
   A.stencils = stencils_dev, A.N = N, A.cs = cs, A.ncells = ncells, A.maxIter = max_iter;
   A.relTol = rel_tol, A.absTol = abs_tol, A.info = info_dev;
   const size_t lds = (size_t) 3 * ncells * cs * sizeof( double );
   if ( lds > 48 * 1024 )
      HH_CHECK_HIP( hipFuncSetAttribute( reinterpret_cast< const void* >( p1_cg_small_kernel ), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int) lds ) );
   hipLaunchKernelGGL( p1_cg_small_kernel, dim3( 1 ), dim3( kCgThreads ), lds, as_stream( stream ), A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_cg_scalars( double* s_dev, int phase, double rel_tol, double abs_tol, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( s_dev && phase >= 0 && phase <= 2, "cg_scalars: null pointer or phase not in 0..2" );
   hipLaunchKernelGGL( cg_scalars_kernel, dim3( 1 ), dim3( 64 ), 0, as_stream( stream ), s_dev, phase, rel_tol, abs_tol );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}


namespace {
int launch_dot_b( int                  ncells,
                  const double* const* a,
                  const double* const* b,
                  int                  level,
                  const unsigned*      masks,
                  double*              result_dev,
                  void*                workspace_dev,
                  double*              cg,
                  int                  cgPhase,
                  double               relTol,
                  double               absTol,
                  hyteg_hip_stream_t   stream )
{
   BATCH_CHECKS( "p1_dot_cells" );
   HH_REQUIRE( a && b && result_dev && workspace_dev, "p1_dot_cells: null pointer" );
   TileTable tt;
   int       rc = full_tiles( level, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   DotB A{};
   for ( int c = 0; c < ncells; ++c )
   {
      HH_REQUIRE( a[c] && b[c], "p1_dot_cells: null operand" );
      A.c.p[0][c] = const_cast< double* >( a[c] );
      A.c.p[1][c] = const_cast< double* >( b[c] );
      A.c.mask[c] = masks[c] & HYTEG_HIP_MASK_ALL;
   }
   A.tiles = tt.dev, A.N = ( 1 << level ) + 1, A.ntiles = tt.count, A.ncells = ncells;
   A.partial = static_cast< double* >( workspace_dev );
   A.result  = result_dev;
   A.cg = cg, A.cgPhase = cgPhase, A.relTol = relTol, A.absTol = absTol;
   const int total  = tt.count * ncells;
   const int blocks = total < 1024 ? ( total > 0 ? total : 1 ) : 1024; // workspace holds 1024 + 256 doubles
   if ( blocks <= 64 )
   {
      rc = dot_counter( as_stream( stream ), &A.counter );
      if ( rc != HYTEG_HIP_OK )
         return rc;
      hipLaunchKernelGGL( batch_dot_kernel, dim3( blocks ), dim3( kThreads ), 0, as_stream( stream ), A );
   }
   else
   {
      A.counter = nullptr;
      hipLaunchKernelGGL( batch_dot_kernel, dim3( blocks ), dim3( kThreads ), 0, as_stream( stream ), A );
      hipLaunchKernelGGL( batch_dot_final_kernel, dim3( 1 ), dim3( kThreads ), 0, as_stream( stream ), A.partial, blocks, result_dev, cg, cgPhase,
                          relTol, absTol );
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}
} // namespace

HYTEG_HIP_API int hyteg_hip_p1_dot_cells( int                  ncells,
                                          const double* const* a,
                                          const double* const* b,
                                          int                  level,
                                          const unsigned*      masks,
                                          double*              result_dev,
                                          void*                workspace_dev,
                                          hyteg_hip_stream_t   stream )
{
   return launch_dot_b( ncells, a, b, level, masks, result_dev, workspace_dev, nullptr, 0, 0.0, 0.0, stream );
}

HYTEG_HIP_API int hyteg_hip_p1_dot_cells_cg( int                  ncells,
                                             const double* const* a,
                                             const double* const* b,
                                             int                  level,
                                             const unsigned*      masks,
                                             double*              s_dev,
                                             int                  slot,
                                             int                  phase,
                                             double               rel_tol,
                                             double               abs_tol,
                                             void*                workspace_dev,
                                             hyteg_hip_stream_t   stream )
{
   HH_REQUIRE( s_dev && ( slot == HYTEG_HIP_CG_PAP || slot == HYTEG_HIP_CG_RR ) && phase >= 0 && phase <= 2,
               "p1_dot_cells_cg: slot must be PAP or RR, phase 0..2" );
   return launch_dot_b( ncells, a, b, level, masks, s_dev + slot, workspace_dev, s_dev, phase, rel_tol, abs_tol, stream );
}

static int launch_apply_b( int                  mode,
                           int                  ncells,
                           double* const*       dst,
                           const double* const* src,
                           const double* const* rhs,
                           const double* const* invdiag,
                           int                  level,
                           const double*        stencils_dev,
                           double               relax,
                           const unsigned*      masks,
                           hyteg_hip_stream_t   stream )
{
   TileTable tt;
   int       rc = full_tiles( level, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   ApplyB A{};
   bool   any = false;
   for ( int c = 0; c < ncells; ++c )
   {
      HH_REQUIRE( dst[c] && src[c] && dst[c] != src[c], "batched apply: null or aliased dst / src" );
      A.c.p[0][c] = dst[c];
      A.c.p[1][c] = const_cast< double* >( src[c] );
      if ( mode >= 2 )
      {
         HH_REQUIRE( rhs && invdiag && rhs[c] && invdiag[c], "p1_jacobi_cells: null rhs / inverse diagonal" );
         A.c.p[2][c] = const_cast< double* >( rhs[c] );
         A.c.p[3][c] = const_cast< double* >( invdiag[c] );
      }
      A.c.mask[c] = masks[c] & HYTEG_HIP_MASK_ALL;
      any         = any || A.c.mask[c];
   }
   if ( !any || tt.count == 0 )
      return HYTEG_HIP_OK;
   A.stencils = stencils_dev, A.tiles = tt.dev, A.N = ( 1 << level ) + 1, A.mode = mode, A.relax = relax;
   hipLaunchKernelGGL( batch_apply_kernel, dim3( tt.count, ncells ), dim3( kThreads ), 0, as_stream( stream ), A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_apply_cells( int                  ncells,
                                            double* const*       dst,
                                            const double* const* src,
                                            int                  level,
                                            const double*        stencils_dev,
                                            const unsigned*      masks,
                                            int                  update,
                                            hyteg_hip_stream_t   stream )
{
   BATCH_CHECKS( "p1_apply_cells" );
   HH_REQUIRE( dst && src && stencils_dev, "p1_apply_cells: null pointer" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p1_apply_cells: bad update type" );
   return launch_apply_b( update == HYTEG_HIP_ADD ? 1 : 0, ncells, dst, src, nullptr, nullptr, level, stencils_dev, 0.0, masks, stream );
}

HYTEG_HIP_API int hyteg_hip_p1_jacobi_cells( int                  ncells,
                                             double* const*       dst,
                                             const double* const* rhs,
                                             const double* const* src,
                                             const double* const* invdiag,
                                             int                  level,
                                             const double*        stencils_dev,
                                             double               relax,
                                             const unsigned*      masks,
                                             int                  phase,
                                             hyteg_hip_stream_t   stream )
{
   BATCH_CHECKS( "p1_jacobi_cells" );
   HH_REQUIRE( dst && rhs && src && invdiag && stencils_dev, "p1_jacobi_cells: null pointer" );
   HH_REQUIRE( phase == 0 || phase == 1, "p1_jacobi_cells: phase must be 0 or 1" );
   return launch_apply_b( 2 + phase, ncells, dst, src, rhs, invdiag, level, stencils_dev, relax, masks, stream );
}

static int launch_transfer_b( bool                 restrict_,
                              int                  ncells,
                              double* const*       out,
                              const double* const* in,
                              int                  coarse_level,
                              const double*        nnc_inv_dev,
                              const unsigned*      masks,
                              int                  update,
                              hyteg_hip_stream_t   stream )
{
   TileTable tt;
   int       rc = full_tiles( restrict_ ? coarse_level : coarse_level + 1, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   TransferB A{};
   bool      any = false;
   for ( int c = 0; c < ncells; ++c )
   {
      HH_REQUIRE( out[c] && in[c], "batched grid transfer: null array" );
      A.c.p[0][c] = out[c];
      A.c.p[1][c] = const_cast< double* >( in[c] );
      A.c.mask[c] = masks[c] & HYTEG_HIP_MASK_ALL;
      any         = any || A.c.mask[c];
   }
   if ( !any || tt.count == 0 )
      return HYTEG_HIP_OK;
   A.nncInv = nnc_inv_dev, A.tiles = tt.dev, A.Nc = ( 1 << coarse_level ) + 1, A.update = update;
   if ( restrict_ )
      hipLaunchKernelGGL( batch_restrict_kernel, dim3( tt.count, ncells ), dim3( kThreads ), 0, as_stream( stream ), A );
   else
      hipLaunchKernelGGL( batch_prolongate_kernel, dim3( tt.count, ncells ), dim3( kThreads ), 0, as_stream( stream ), A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_restrict_cells( int                  ncells,
                                               double* const*       coarse,
                                               const double* const* fine,
                                               int                  coarse_level,
                                               const double*        nnc_inv_dev,
                                               const unsigned*      masks,
                                               hyteg_hip_stream_t   stream )
{
   const int level = coarse_level;
   BATCH_CHECKS( "p1_restrict_cells" );
   HH_REQUIRE( coarse && fine && nnc_inv_dev && coarse_level + 1 <= HYTEG_HIP_MAX_LEVEL, "p1_restrict_cells: null pointer or level" );
   return launch_transfer_b( true, ncells, coarse, fine, coarse_level, nnc_inv_dev, masks, 0, stream );
}

HYTEG_HIP_API int hyteg_hip_p1_prolongate_cells( int                  ncells,
                                                 const double* const* coarse,
                                                 double* const*       fine,
                                                 int                  coarse_level,
                                                 const double*        nnc_inv_dev,
                                                 const unsigned*      masks,
                                                 int                  update,
                                                 hyteg_hip_stream_t   stream )
{
   const int level = coarse_level;
   BATCH_CHECKS( "p1_prolongate_cells" );
   HH_REQUIRE( coarse && fine && nnc_inv_dev && coarse_level + 1 <= HYTEG_HIP_MAX_LEVEL, "p1_prolongate_cells: null pointer or level" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p1_prolongate_cells: bad update type" );
   return launch_transfer_b( false, ncells, fine, coarse, coarse_level, nnc_inv_dev, masks, update, stream );
}

} // extern "C"
