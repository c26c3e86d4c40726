// C-ABI entry points: assign / add / multElementwise / dot on the interior of one macro-cell.
// Pure streaming kernels: one workgroup per inner tile, 8-byte coalesced accesses, the only extra
// work being the row decode that masks the two boundary entries of every row.
#include <algorithm>

#include "common.hpp"
#include "kernels_apply_zmarch.hpp" // BrickTask

using namespace hyteg_hip;

namespace {

constexpr int kTile    = 1024;
constexpr int kThreads = 256;

struct VecArgs
{
   double*       dst;
   const double* src[HYTEG_HIP_MAX_SRCS];
   double        c[HYTEG_HIP_MAX_SRCS];
   const Tile*   tiles;
   int           ntiles;
   int           N;
   int           nsrc;
};

enum VecOp
{
   OP_ASSIGN = 0,
   OP_ADD    = 1,
   OP_MULT   = 2
};

__device__ inline bool inner_entry( int N, const Tile& tl, int s0, int i )
{
   const int W = N - tl.z;
   const int j = i - s0;
   const int y = row_of( W, j );
   const int x = j - row_start( W, y );
   return x >= 1 && x <= W - y - 2;
}

constexpr int kPerThread = kTile / kThreads; // entries per thread, all loads of a thread issued before the first use

template < int OP, int NSRC >
__global__ __launch_bounds__( kThreads ) void p1_vector_kernel( const VecArgs A )
{
   const int t = blockIdx.x;
   if ( t >= A.ntiles )
      return;
   const Tile tl = A.tiles[t];
   const int  s0 = slice_start( A.N, tl.z );
   double     v[kPerThread][NSRC + 1];
   bool       in[kPerThread];
#pragma unroll
   for ( int u = 0; u < kPerThread; ++u )
   {
      const int e = (int) threadIdx.x + u * kThreads;
      const int i = tl.a + ( e < tl.cnt ? e : tl.cnt - 1 ); // clamped: the load is always legal, the store is predicated
      in[u]       = e < tl.cnt && inner_entry( A.N, tl, s0, i );
#pragma unroll
      for ( int k = 0; k < NSRC; ++k )
         v[u][k] = A.src[k][i];
      if ( OP == OP_ADD )
         v[u][NSRC] = A.dst[i];
   }
#pragma unroll
   for ( int u = 0; u < kPerThread; ++u )
   {
      double tmp;
      if ( OP == OP_MULT )
      {
         tmp = v[u][0];
#pragma unroll
         for ( int k = 1; k < NSRC; ++k )
            tmp *= v[u][k];
      }
      else
      {
         tmp = A.c[0] * v[u][0];
#pragma unroll
         for ( int k = 1; k < NSRC; ++k )
            tmp += A.c[k] * v[u][k];
         if ( OP == OP_ADD )
            tmp = v[u][NSRC] + tmp;
      }
      if ( in[u] )
         __builtin_nontemporal_store( tmp, &A.dst[tl.a + (int) threadIdx.x + u * kThreads] );
   }
}

template < int OP >
int launch_vec( double* dst, int nsrc, const double* const* srcs, const double* scalars, int level, hipStream_t stream )
{
   TileTable tt;
   int       rc = get_tiles( level, TILES_INNER, kTile, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   if ( tt.count == 0 )
      return HYTEG_HIP_OK;
   VecArgs A{};
   A.dst = dst;
   for ( int k = 0; k < nsrc; ++k )
   {
      A.src[k] = srcs[k];
      A.c[k]   = scalars ? scalars[k] : 1.0;
   }
   A.tiles  = tt.dev;
   A.ntiles = tt.count;
   A.N      = ( 1 << level ) + 1;
   A.nsrc   = nsrc;
   switch ( nsrc )
   {
   case 1:
      hipLaunchKernelGGL( ( p1_vector_kernel< OP, 1 > ), dim3( tt.count ), dim3( kThreads ), 0, stream, A );
      break;
   case 2:
      hipLaunchKernelGGL( ( p1_vector_kernel< OP, 2 > ), dim3( tt.count ), dim3( kThreads ), 0, stream, A );
      break;
   case 3:
      hipLaunchKernelGGL( ( p1_vector_kernel< OP, 3 > ), dim3( tt.count ), dim3( kThreads ), 0, stream, A );
      break;
   default:
      hipLaunchKernelGGL( ( p1_vector_kernel< OP, 4 > ), dim3( tt.count ), dim3( kThreads ), 0, stream, A );
      break;
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

// ---- dot ---------------------------------------------------------------------------------------
constexpr int kDotBlocks = 1024; // partial sums; also the workspace size in doubles
constexpr int kDotSingleLaunchBlocks = 64;

__device__ inline double wave_sum( double v )
{
#pragma unroll
   for ( int off = 32; off > 0; off >>= 1 )
      v += __shfl_down( v, off, 64 );
   return v;
}

__device__ inline double block_sum( double v, double* sh /* >= kThreads/64 */ )
{
   v = wave_sum( v );
   if ( ( threadIdx.x & 63 ) == 0 )
      sh[threadIdx.x >> 6] = v;
   __syncthreads();
   double r = 0.0;
   if ( threadIdx.x == 0 )
   {
#pragma unroll
      for ( int k = 0; k < kThreads / 64; ++k )
         r += sh[k];
   }
   return r; // valid in thread 0
}

// One launch: every workgroup writes its partial sum through to memory (agent-scope store: the 8 XCDs do not share an L2),
// waits until the store is acknowledged and takes a ticket; the workgroup that draws the last ticket reduces the partial sums,
// in the same fixed order whichever it is (the order of p1_dot_final_kernel).  No release fence: on this architecture a
// device-scope fence is an L2 write-back per workgroup (round 1 measured 38.5 instead of 12.8 us for a level-8 dot with ~1000
// workgroups in one launch).  Without fences the one-launch form at that size costs 15.2 against 11.6 us for two launches, so it
// stays reserved for few workgroups.
__device__ inline void dot_finish( double r, double* partial, unsigned* counter, double* result, double* sh )
{
   __shared__ bool last;
   if ( threadIdx.x == 0 )
   {
      __hip_atomic_store( partial + blockIdx.x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
      last = __hip_atomic_fetch_add( counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) == gridDim.x - 1;
   }
   __syncthreads();
   if ( !last )
      return;
   double sum = 0.0;
   for ( int k = threadIdx.x; k < (int) gridDim.x; k += kThreads )
      sum += __hip_atomic_load( partial + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
   __syncthreads(); // sh is reused
   const double grand = block_sum( sum, sh );
   if ( threadIdx.x == 0 )
   {
      *result = grand;
      __hip_atomic_store( counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
   }
}

__global__ __launch_bounds__( kThreads ) void p1_dot_partial_kernel( const double* __restrict__ a,
                                                                      const double* __restrict__ b,
                                                                      const Tile* tiles,
                                                                      int         ntiles,
                                                                      int         N,
                                                                      double*     partial,
                                                                      unsigned*   counter,
                                                                      double*     result )
{
   __shared__ double sh[kThreads / 64];
   double            acc = 0.0;
   // fixed tile -> workgroup assignment: deterministic summation order
   for ( int t = blockIdx.x; t < ntiles; t += gridDim.x )
   {
      const Tile tl = tiles[t];
      const int  s0 = slice_start( N, tl.z );
      double     va[kPerThread], vb[kPerThread];
      bool       in[kPerThread];
#pragma unroll
      for ( int u = 0; u < kPerThread; ++u )
      {
         const int e = (int) threadIdx.x + u * kThreads;
         const int i = tl.a + ( e < tl.cnt ? e : tl.cnt - 1 );
         in[u]       = e < tl.cnt && inner_entry( N, tl, s0, i );
         va[u]       = a[i];
         vb[u]       = b[i];
      }
#pragma unroll
      for ( int u = 0; u < kPerThread; ++u )
         acc = in[u] ? fma( va[u], vb[u], acc ) : acc;
   }
   const double r = block_sum( acc, sh );
   if ( result == nullptr )
   {
      // the caller reduces the partial sums together with others (masked dot)
      if ( threadIdx.x == 0 )
         partial[blockIdx.x] = r;
      return;
   }
   dot_finish( r, partial, counter, result, sh );
}

// Dot product over the inner points enumerated by the brick tasks of the z-march apply (NY rows x nz slices x 62 lanes):
// no index decoding per entry, coalesced 512-byte row segments, only inner points are touched.  Fixed task -> wave
// assignment and fixed reduction trees: deterministic.
template < int NY >
__global__ __launch_bounds__( kThreads ) void p1_dot_brick_kernel( const double* __restrict__ a,
                                                                    const double* __restrict__ b,
                                                                    const BrickTask* __restrict__ tasks,
                                                                    int       ntasks,
                                                                    double*   partial,
                                                                    unsigned* counter,
                                                                    double*   result )
{
   __shared__ double sh[kThreads / 64];
   const int         lane = threadIdx.x & 63;
   constexpr int     kWaves = kThreads / 64;
   double            acc  = 0.0;
   for ( int task = blockIdx.x * kWaves + ( threadIdx.x >> 6 ); task < ntasks; task += gridDim.x * kWaves )
   {
      const BrickTask t  = tasks[task];
      const int       ym = t.y0 - 1;
      int             base = t.i0, Wq = t.W0; // (xb, ym, z0 - 1), row-0 length of that slice
      for ( int s = 0; s < t.nz; ++s )
      {
         base += tri( Wq ) - ym; // (xb, ym, z0 + s)
         Wq -= 1;
         int    io = base + ( Wq - ym ); // (xb, y0, z0 + s)
         double va[NY], vb[NY];
         bool   ok[NY];
#pragma unroll
         for ( int j = 0; j < NY; ++j )
         {
            const int R   = Wq - ( t.y0 + j );       // length of row y0 + j
            const int cnt = min( 62, R - 2 - t.xb ); // inner points of the row held by lanes 1 .. cnt
            ok[j]         = (unsigned) ( lane - 1 ) < (unsigned) max( cnt, 0 );
            const int i   = ok[j] ? io + lane : t.i0;
            va[j]         = a[i];
            vb[j]         = b[i];
            io += R;
         }
#pragma unroll
         for ( int j = 0; j < NY; ++j )
            acc = ok[j] ? fma( va[j], vb[j], acc ) : acc;
      }
   }
   const double r = block_sum( acc, sh );
   if ( result == nullptr )
   {
      if ( threadIdx.x == 0 )
         partial[blockIdx.x] = r;
      return;
   }
   dot_finish( r, partial, counter, result, sh );
}

__global__ __launch_bounds__( kThreads ) void p1_dot_final_kernel( const double* partial, int n, double* result )
{
   __shared__ double sh[kThreads / 64];
   double            acc = 0.0;
   for ( int k = threadIdx.x; k < n; k += kThreads )
      acc += partial[k];
   const double r = block_sum( acc, sh );
   if ( threadIdx.x == 0 )
      *result = r;
}

} // namespace

namespace hyteg_hip {
int launch_vec_inner( int op, double* dst, int nsrc, const double* const* srcs, const double* scalars, int level, hipStream_t stream )
{
   if ( op == 0 )
      return launch_vec< OP_ASSIGN >( dst, nsrc, srcs, scalars, level, stream );
   if ( op == 1 )
      return launch_vec< OP_ADD >( dst, nsrc, srcs, scalars, level, stream );
   return launch_vec< OP_MULT >( dst, nsrc, srcs, nullptr, level, stream );
}

int launch_dot_inner_partial( const double* a, const double* b, int level, double* partial, int* nblocks, hipStream_t stream )
{
   TileTable tt;
   int       rc = get_tiles( level, TILES_INNER, kTile, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   *nblocks = tt.count < kDotBlocks ? tt.count : kDotBlocks;
   if ( *nblocks > 0 )
      hipLaunchKernelGGL( p1_dot_partial_kernel, dim3( *nblocks ), dim3( kThreads ), 0, stream, a, b, tt.dev, tt.count,
                          ( 1 << level ) + 1, partial, (unsigned*) nullptr, (double*) nullptr );
   return HYTEG_HIP_OK;
}
} // namespace hyteg_hip

extern "C" {

#define VEC_COMMON_CHECKS( name )                                                                          \
   HH_REQUIRE( dst && srcs, name ": null pointer" );                                                       \
   HH_REQUIRE( level_ok( level ), name ": level out of range [2,11]" );                                    \
   HH_REQUIRE( nsrc >= 1 && nsrc <= HYTEG_HIP_MAX_SRCS, name ": nsrc must be 1..HYTEG_HIP_MAX_SRCS" );     \
   for ( int k = 0; k < nsrc; ++k )                                                                        \
      HH_REQUIRE( srcs[k] != nullptr, name ": null source pointer" );

HYTEG_HIP_API int hyteg_hip_p1_assign_cell( double*              dst,
                                            int                  nsrc,
                                            const double* const* srcs,
                                            const double*        scalars,
                                            int                  level,
                                            hyteg_hip_stream_t   stream )
{
   VEC_COMMON_CHECKS( "p1_assign_cell" );
   HH_REQUIRE( scalars != nullptr, "p1_assign_cell: null scalars" );
   return launch_vec< OP_ASSIGN >( dst, nsrc, srcs, scalars, level, as_stream( stream ) );
}

HYTEG_HIP_API int hyteg_hip_p1_add_cell( double*              dst,
                                         int                  nsrc,
                                         const double* const* srcs,
                                         const double*        scalars,
                                         int                  level,
                                         hyteg_hip_stream_t   stream )
{
   VEC_COMMON_CHECKS( "p1_add_cell" );
   HH_REQUIRE( scalars != nullptr, "p1_add_cell: null scalars" );
   return launch_vec< OP_ADD >( dst, nsrc, srcs, scalars, level, as_stream( stream ) );
}

HYTEG_HIP_API int
    hyteg_hip_p1_mult_cell( double* dst, int nsrc, const double* const* srcs, int level, hyteg_hip_stream_t stream )
{
   VEC_COMMON_CHECKS( "p1_mult_cell" );
   return launch_vec< OP_MULT >( dst, nsrc, srcs, nullptr, level, as_stream( stream ) );
}

HYTEG_HIP_API size_t hyteg_hip_dot_workspace_bytes( void ) { return ( kDotBlocks + 256 ) * sizeof( double ); }

HYTEG_HIP_API int hyteg_hip_p1_dot_cell( const double*      a,
                                         const double*      b,
                                         int                level,
                                         double*            result_dev,
                                         void*              workspace_dev,
                                         hyteg_hip_stream_t stream )
{
   HH_REQUIRE( a && b && result_dev && workspace_dev, "p1_dot_cell: null pointer" );
   HH_REQUIRE( level_ok( level ), "p1_dot_cell: level out of range [2,11]" );
   TileTable tt;
   int       rc = get_tiles( level, TILES_INNER, kTile, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   double* partial = static_cast< double* >( workspace_dev );
   if ( level >= 8 && level <= 10 )
   {
      // large arrays: the brick tasks of the apply enumerate the inner points without any per-entry index decoding
      BrickTable bt;
      rc = get_bricks( level, 4, 8, &bt );
      if ( rc != HYTEG_HIP_OK )
         return rc;
      const int nb = std::min( kDotBlocks, ( bt.count + kThreads / 64 - 1 ) / ( kThreads / 64 ) );
      // two launches: with ~500 workgroups the one-launch form (dot_finish) is slower -- 15.2 vs 11.6 us at level 8: every
      // workgroup ends with a write-through store, its acknowledgement and a ticket, and the last one reads 500 partial sums
      hipLaunchKernelGGL( p1_dot_brick_kernel< 4 >, dim3( nb ), dim3( kThreads ), 0, as_stream( stream ), a, b, bt.dev, bt.count, partial,
                          (unsigned*) nullptr, (double*) nullptr );
      hipLaunchKernelGGL( p1_dot_final_kernel, dim3( 1 ), dim3( kThreads ), 0, as_stream( stream ), partial, nb, result_dev );
      HH_CHECK_HIP( hipGetLastError() );
      return HYTEG_HIP_OK;
   }
   const int nblocks = tt.count < kDotBlocks ? ( tt.count > 0 ? tt.count : 1 ) : kDotBlocks;
   if ( nblocks <= kDotSingleLaunchBlocks )
   {
      // few workgroups (coarse levels): the one that finishes last reduces the partial sums -- one launch (dot_finish)
      unsigned* counter = nullptr;
      rc                = dot_counter( as_stream( stream ), &counter );
      if ( rc != HYTEG_HIP_OK )
         return rc;
      hipLaunchKernelGGL( p1_dot_partial_kernel, dim3( nblocks ), dim3( kThreads ), 0, as_stream( stream ), a, b, tt.dev, tt.count,
                          ( 1 << level ) + 1, partial, counter, result_dev );
   }
   else
   {
      hipLaunchKernelGGL( p1_dot_partial_kernel, dim3( nblocks ), dim3( kThreads ), 0, as_stream( stream ), a, b, tt.dev, tt.count,
                          ( 1 << level ) + 1, partial, (unsigned*) nullptr, (double*) nullptr );
      hipLaunchKernelGGL( p1_dot_final_kernel, dim3( 1 ), dim3( kThreads ), 0, as_stream( stream ), partial, nblocks, result_dev );
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_prepare_level( int level )
{
   HH_REQUIRE( level_ok( level ), "prepare_level: level out of range [2,11]" );
   TileTable tt;
   int       rc = get_tiles( level, TILES_INNER, kTile, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   rc = get_tiles( level, TILES_FULL, kTile, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   rc = get_tiles( level, TILES_ROWS, 64, &tt ); // restriction onto this level (p1_transfer.hip, kRestrictRow)
   if ( rc != HYTEG_HIP_OK )
      return rc;
   BrickTable bt; // same shapes as p1_apply.hip: 4 x 8 for Replace from level 8 on, 4 x 4 otherwise
   rc = get_bricks( level, 4, 4, &bt );
   if ( rc != HYTEG_HIP_OK || level < 8 )
      return rc;
   return get_bricks( level, 4, 8, &bt );
}
}
