// Developer micro-benchmark (not part of the product library): times variants of the macro-cell apply kernel
// on rotating buffer pairs (working set > 256 MiB Infinity Cache) and a copy kernel of the same traffic.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHYTEG_HIP_BUILDING -o apply_bench apply_bench.hip ../runtime.hip
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../kernels_apply.hpp"
#include "kernels_apply_rowmarch.hpp"
#include "kernels_apply_strip2.hpp"
#include "kernels_apply_zmarch_r01.hpp"

using namespace hyteg_hip;

#define CK( e )                                                                                   \
   do                                                                                             \
   {                                                                                              \
      hipError_t _e = ( e );                                                                      \
      if ( _e != hipSuccess )                                                                     \
      {                                                                                           \
         fprintf( stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString( _e ) );    \
         exit( 1 );                                                                               \
      }                                                                                           \
   } while ( 0 )

// naive reference: one wave per (y,z) row, 15 direct global loads per point
__global__ __launch_bounds__( 64 ) void apply_rows_naive( double* dst, const double* src, int N, Stencil15 st )
{
   const int n = N - 1;
   // decode (z,y) from the row counter: rows of slice z: y = 1 .. n-z-1 (n - z - 1 rows, last may be empty)
   int r = blockIdx.x, z = 1;
   // nrows(z) = n - 1 - z ; cumulative -> find z by float sqrt then fix
   {
      // rows before slice z: sum_{k=1}^{z-1} (n-1-k) = (z-1)(n-1) - z(z-1)/2
      float a  = (float) ( 2 * n - 1 );
      float zz = ( a - sqrtf( a * a - 8.0f * (float) r ) ) * 0.5f;
      z        = (int) zz + 1;
      auto before = [n]( int z ) { return ( z - 1 ) * ( n - 1 ) - ( z * ( z - 1 ) ) / 2; };
      while ( z > 1 && before( z ) > r )
         --z;
      while ( before( z + 1 ) <= r )
         ++z;
      r -= before( z );
   }
   const int y = 1 + r;
   const int W = N - z, R = W - y, S0 = tri( W ), Sm = tri( W + 1 );
   const int base = slice_start( N, z ) + row_start( W, y );
   const double* w = st.w;
   for ( int x = 1 + threadIdx.x; x <= R - 2; x += 64 )
   {
      const int i = base + x;
      double    acc;
      acc = w[6] * src[i - 1];
      acc = fma( w[3], src[i - Sm + W + 1], acc );
      acc = fma( w[10], src[i + R], acc );
      acc = fma( w[5], src[i - R], acc );
      acc = fma( w[12], src[i + S0 - W + 1], acc );
      acc = fma( w[1], src[i - Sm + y + 1], acc );
      acc = fma( w[8], src[i + 1], acc );
      acc = fma( w[13], src[i + S0 - y - 1], acc );
      acc = fma( w[2], src[i - Sm + W], acc );
      acc = fma( w[9], src[i + R - 1], acc );
      acc = fma( w[4], src[i - R - 1], acc );
      acc = fma( w[11], src[i + S0 - W], acc );
      acc = fma( w[0], src[i - Sm + y], acc );
      acc = fma( w[7], src[i], acc );
      acc = fma( w[14], src[i + S0 - y], acc );
      dst[i] = acc;
   }
}

__global__ __launch_bounds__( 256 ) void copy_kernel( double2* dst, const double2* src, int npairs )
{
   for ( int k = blockIdx.x * 256 + threadIdx.x; k < npairs; k += gridDim.x * 256 )
      dst[k] = src[k];
}

template < int U >
__global__ __launch_bounds__( 256 ) void copy_batched_kernel( double2* __restrict__ dst, const double2* __restrict__ src, int npairs )
{
   // U independent 16-byte loads issued back to back per thread (clamped index: no control flow), then U stores
   const int base = blockIdx.x * ( U * 256 ) + threadIdx.x;
   double2   v0, v1, v2, v3, v4, v5, v6, v7;
#define CL( u ) src[min( base + ( u ) * 256, npairs - 1 )]
   v0 = CL( 0 );
   if ( U > 1 ) v1 = CL( 1 );
   if ( U > 2 ) v2 = CL( 2 );
   if ( U > 3 ) v3 = CL( 3 );
   if ( U > 4 ) v4 = CL( 4 );
   if ( U > 5 ) v5 = CL( 5 );
   if ( U > 6 ) v6 = CL( 6 );
   if ( U > 7 ) v7 = CL( 7 );
#undef CL
#define ST( u, v ) if ( U > ( u ) && base + ( u ) * 256 < npairs ) dst[base + ( u ) * 256] = v
   ST( 0, v0 );
   ST( 1, v1 );
   ST( 2, v2 );
   ST( 3, v3 );
   ST( 4, v4 );
   ST( 5, v5 );
   ST( 6, v6 );
   ST( 7, v7 );
#undef ST
}

template < int U >
__global__ __launch_bounds__( 256 ) void copy_unrolled_kernel( double2* __restrict__ dst, const double2* __restrict__ src, int npairs )
{
   // each workgroup copies a contiguous chunk of U*256 pairs; U loads in flight per thread
   const int base = blockIdx.x * ( U * 256 ) + threadIdx.x;
   double2   v[U];
#pragma unroll
   for ( int u = 0; u < U; ++u )
      if ( base + u * 256 < npairs )
         v[u] = src[base + u * 256];
#pragma unroll
   for ( int u = 0; u < U; ++u )
      if ( base + u * 256 < npairs )
         dst[base + u * 256] = v[u];
}

template < typename T, bool NT >
__global__ __launch_bounds__( 256 ) void copy_w_kernel( T* __restrict__ dst, const T* __restrict__ src, int n )
{
   for ( int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256 )
   {
      T v = src[k];
      if ( NT )
         __builtin_nontemporal_store( v, &dst[k] );
      else
         dst[k] = v;
   }
}
// copy where every wave reads an unaligned 64-element window and writes only lanes 1..62 of it (the z-march kernel's
// access shape: 496-byte stores that start and end inside a 64-byte sector); SHIFT = offset of the windows in elements
template < bool NT, int ACTIVE >
__global__ __launch_bounds__( 256 ) void copy_seg_kernel( double* __restrict__ dst, const double* __restrict__ src, int n, int shift )
{
   const int lane = threadIdx.x & 63;
   const int nw   = gridDim.x * 4;
   for ( int w = blockIdx.x * 4 + ( threadIdx.x >> 6 ); w * ACTIVE + shift < n; w += nw )
   {
      const int k = w * ACTIVE + shift + lane - ( 64 - ACTIVE ) / 2;
      if ( k < 0 || k >= n )
         continue;
      const double v = src[k];
      if ( lane >= ( 64 - ACTIVE ) / 2 && lane < ( 64 - ACTIVE ) / 2 + ACTIVE )
      {
         if ( NT )
            __builtin_nontemporal_store( v, &dst[k] );
         else
            dst[k] = v;
      }
   }
}
template < typename T, bool NT >
__global__ __launch_bounds__( 256 ) void fill_w_kernel( T* __restrict__ dst, T v, int n )
{
   for ( int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256 )
   {
      if ( NT )
         __builtin_nontemporal_store( v, &dst[k] );
      else
         dst[k] = v;
   }
}
template < typename T >
__global__ __launch_bounds__( 256 ) void read_w_kernel( double* __restrict__ out, const T* __restrict__ src, int n )
{
   double acc = 0;
   for ( int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256 )
   {
      T v = src[k];
      acc += *reinterpret_cast< double* >( &v );
   }
   if ( acc == 1.2345e-300 )
      out[0] = acc;
}

struct Variant
{
   const char* name;
   int         T;
   bool        xcd;
};

int main( int argc, char** argv )
{
   const int level = argc > 1 ? atoi( argv[1] ) : 8;
   const int reps  = argc > 2 ? atoi( argv[2] ) : 200;
   const int nbuf  = argc > 3 ? atoi( argv[3] ) : 8;
   const char* filter = argc > 4 ? argv[4] : ""; // only run variants whose name contains this
   const int N     = ( 1 << level ) + 1;
   const int total = (int) tet64( N );
   const int64_t inner = hyteg_hip_cell_inner_size( level );
   printf( "level %d  N %d  entries %d  inner %lld  buffers %d pairs (%.1f MB total)\n", level, N, total, (long long) inner,
           nbuf, nbuf * 2.0 * total * 8 / 1e6 );

   std::vector< double > h( total );
   std::mt19937_64       gen( 42 );
   std::uniform_real_distribution< double > U( 0, 1 );
   for ( auto& v : h )
      v = U( gen );
   Stencil15 st;
   for ( int k = 0; k < 15; ++k )
      st.w[k] = U( gen ) - 0.5;

   std::vector< double* > src( nbuf ), dst( nbuf );
   for ( int b = 0; b < nbuf; ++b )
   {
      CK( hipMalloc( &src[b], (size_t) total * 8 + 16 ) );
      CK( hipMalloc( &dst[b], (size_t) total * 8 + 16 ) );
      CK( hipMemcpy( src[b], h.data(), (size_t) total * 8, hipMemcpyHostToDevice ) );
      CK( hipMemset( dst[b], 0, (size_t) total * 8 ) );
   }
   double* refout;
   CK( hipMalloc( &refout, (size_t) total * 8 ) );
   CK( hipMemset( refout, 0, (size_t) total * 8 ) );
   const int n     = N - 1;
   const int nrows = ( n - 2 ) * ( n - 1 ) / 2; // sum_{z=1}^{n-2} (n-1-z)
   hipLaunchKernelGGL( apply_rows_naive, dim3( nrows ), dim3( 64 ), 0, 0, refout, src[0], N, st );
   CK( hipDeviceSynchronize() );
   std::vector< double > href( total ), hout( total );
   CK( hipMemcpy( href.data(), refout, (size_t) total * 8, hipMemcpyDeviceToHost ) );

   auto want = [&]( const char* name ) { return filter[0] == 0 || strstr( name, filter ) != nullptr; };
   hipEvent_t e0, e1;
   CK( hipEventCreate( &e0 ) );
   CK( hipEventCreate( &e1 ) );
   auto report = [&]( const char* name, float ms, double maxdiff ) {
      const double us = ms * 1e3 / reps;
      printf( "%-34s %9.2f us/launch  %8.1f GDoF/s  %8.1f GB/s(16B/DoF)  maxdiff %.2e\n", name, us,
              inner / us * 1e-3, 16.0 * inner / us * 1e-3, maxdiff );
      fflush( stdout );
   };

   // copy kernel (same bytes as an apply: read total, write total)
   for ( int grid : { 1024, 2048, 4096 } )
   {
      if ( !want( "copy" ) && strstr( filter, "copy" ) != filter )
         break;
      if ( strstr( filter, "copy " ) == filter )
         break;
      for ( int r = 0; r < 10; ++r )
         hipLaunchKernelGGL( copy_kernel, dim3( grid ), dim3( 256 ), 0, 0, (double2*) dst[r % nbuf], (const double2*) src[r % nbuf], total / 2 );
      CK( hipEventRecord( e0 ) );
      for ( int r = 0; r < reps; ++r )
         hipLaunchKernelGGL( copy_kernel, dim3( grid ), dim3( 256 ), 0, 0, (double2*) dst[r % nbuf], (const double2*) src[r % nbuf], total / 2 );
      CK( hipEventRecord( e1 ) );
      CK( hipEventSynchronize( e1 ) );
      float ms;
      CK( hipEventElapsedTime( &ms, e0, e1 ) );
      char nm[64];
      snprintf( nm, 64, "copy double2 grid=%d", grid );
      report( nm, ms, 0.0 );
   }

   if ( want( "copy" ) || strstr( filter, "copy" ) == filter )
   {
      typedef double __attribute__( ( ext_vector_type( 2 ) ) ) d2;
      auto timeit = [&]( const char* nm, auto&& launch ) {
         if ( strstr( filter, "copy " ) == filter && strcmp( filter, nm ) != 0 && strstr( nm, filter ) == nullptr )
            return;
         for ( int r = 0; r < 10; ++r )
            launch( r % nbuf );
         CK( hipEventRecord( e0 ) );
         for ( int r = 0; r < reps; ++r )
            launch( r % nbuf );
         CK( hipEventRecord( e1 ) );
         CK( hipEventSynchronize( e1 ) );
         float ms;
         CK( hipEventElapsedTime( &ms, e0, e1 ) );
         report( nm, ms, 0.0 );
      };
      for ( int grid : { 1024, 2048 } )
      {
         char nm[96];
         snprintf( nm, 96, "copy 8B/lane grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_w_kernel< double, false > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], src[b], total ); } );
         snprintf( nm, 96, "copy 8B/lane nt-store grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_w_kernel< double, true > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], src[b], total ); } );
         snprintf( nm, 96, "copy 16B/lane grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_w_kernel< d2, false > ), dim3( grid ), dim3( 256 ), 0, 0, (d2*) dst[b], (const d2*) src[b], total / 2 ); } );
         snprintf( nm, 96, "copy 16B/lane nt-store grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_w_kernel< d2, true > ), dim3( grid ), dim3( 256 ), 0, 0, (d2*) dst[b], (const d2*) src[b], total / 2 ); } );
         snprintf( nm, 96, "copy seg 62-of-64 lanes nt shift=1 grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_seg_kernel< true, 62 > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], src[b], total, 1 ); } );
         snprintf( nm, 96, "copy seg 64-of-64 lanes nt shift=0 grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_seg_kernel< true, 64 > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], src[b], total, 0 ); } );
         snprintf( nm, 96, "copy seg 64-of-64 lanes nt shift=3 grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_seg_kernel< true, 64 > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], src[b], total, 3 ); } );
         snprintf( nm, 96, "copy seg 56-of-64 lanes nt shift=0 grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_seg_kernel< true, 56 > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], src[b], total, 4 ); } );
         snprintf( nm, 96, "copy seg 62-of-64 lanes plain shift=1 grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_seg_kernel< false, 62 > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], src[b], total, 1 ); } );
         snprintf( nm, 96, "fill 8B/lane grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( fill_w_kernel< double, false > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], 1.5, total ); } );
         snprintf( nm, 96, "fill 16B/lane grid=%d", grid );
         d2 vv = { 1.5, 2.5 };
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( fill_w_kernel< d2, false > ), dim3( grid ), dim3( 256 ), 0, 0, (d2*) dst[b], vv, total / 2 ); } );
         snprintf( nm, 96, "fill 16B/lane nt grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( fill_w_kernel< d2, true > ), dim3( grid ), dim3( 256 ), 0, 0, (d2*) dst[b], vv, total / 2 ); } );
         snprintf( nm, 96, "read 8B/lane grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( read_w_kernel< double > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], src[b], total ); } );
         snprintf( nm, 96, "read 16B/lane grid=%d", grid );
         timeit( nm, [&]( int b ) { hipLaunchKernelGGL( ( read_w_kernel< d2 > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], (const d2*) src[b], total / 2 ); } );
      }
      auto run = [&]( const char* nm, auto kern, int U ) {
         const int npairs = total / 2;
         const int grid   = ( npairs + U * 256 - 1 ) / ( U * 256 );
         for ( int r = 0; r < 10; ++r )
            hipLaunchKernelGGL( kern, dim3( grid ), dim3( 256 ), 0, 0, (double2*) dst[r % nbuf], (const double2*) src[r % nbuf], npairs );
         CK( hipEventRecord( e0 ) );
         for ( int r = 0; r < reps; ++r )
            hipLaunchKernelGGL( kern, dim3( grid ), dim3( 256 ), 0, 0, (double2*) dst[r % nbuf], (const double2*) src[r % nbuf], npairs );
         CK( hipEventRecord( e1 ) );
         CK( hipEventSynchronize( e1 ) );
         float ms;
         CK( hipEventElapsedTime( &ms, e0, e1 ) );
         report( nm, ms, 0.0 );
      };
      run( "copy batched U=1", copy_batched_kernel< 1 >, 1 );
      run( "copy batched U=2", copy_batched_kernel< 2 >, 2 );
      run( "copy batched U=3", copy_batched_kernel< 3 >, 3 );
      run( "copy batched U=4", copy_batched_kernel< 4 >, 4 );
      run( "copy batched U=6", copy_batched_kernel< 6 >, 6 );
      run( "copy batched U=8", copy_batched_kernel< 8 >, 8 );
   }

   // naive
   if ( want( "naive" ) )
   {
      for ( int r = 0; r < 5; ++r )
         hipLaunchKernelGGL( apply_rows_naive, dim3( nrows ), dim3( 64 ), 0, 0, dst[r % nbuf], src[r % nbuf], N, st );
      CK( hipEventRecord( e0 ) );
      for ( int r = 0; r < reps; ++r )
         hipLaunchKernelGGL( apply_rows_naive, dim3( nrows ), dim3( 64 ), 0, 0, dst[r % nbuf], src[r % nbuf], N, st );
      CK( hipEventRecord( e1 ) );
      CK( hipEventSynchronize( e1 ) );
      float ms;
      CK( hipEventElapsedTime( &ms, e0, e1 ) );
      report( "naive rows (15 global loads)", ms, 0.0 );
   }

   const Variant variants[] = { { "tiled T=256", 256, false },   { "tiled T=256 xcd", 256, true },
                                { "tiled T=512", 512, false },   { "tiled T=512 xcd", 512, true },
                                { "tiled T=1024", 1024, false }, { "tiled T=1024 xcd", 1024, true },
                                { "tiled T=2048", 2048, false }, { "tiled T=2048 xcd", 2048, true },
                                { "tiled T=4096 xcd", 4096, true } };
   for ( const Variant& v : variants )
   {
      if ( !want( v.name ) )
         continue;
      TileTable tt;
      if ( get_tiles( level, TILES_INNER, v.T, &tt ) != HYTEG_HIP_OK )
      {
         fprintf( stderr, "tiles failed: %s\n", hyteg_hip_last_error() );
         return 1;
      }
      ApplyArgs A{};
      A.tiles  = tt.dev;
      A.ntiles = tt.count;
      A.N      = N;
      A.total  = total;
      A.st     = st;
      const int nblocks = ( tt.count + 7 ) & ~7;
      A.xcd_chunk       = v.xcd ? nblocks / 8 : 0;
      const size_t lds  = (size_t) apply_lds_doubles( v.T, N ) * 8;
      auto         kern = p1_apply_tiled_kernel< APPLY_REPLACE, true >;
      if ( lds > 48 * 1024 )
         CK( hipFuncSetAttribute( reinterpret_cast< const void* >( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds ) );
      auto launch = [&]( int b ) {
         A.dst = dst[b];
         A.src = src[b];
         hipLaunchKernelGGL( kern, dim3( nblocks ), dim3( kApplyThreads ), lds, 0, A );
      };
      CK( hipMemset( dst[0], 0, (size_t) total * 8 ) );
      launch( 0 );
      CK( hipDeviceSynchronize() );
      CK( hipMemcpy( hout.data(), dst[0], (size_t) total * 8, hipMemcpyDeviceToHost ) );
      double maxdiff = 0;
      for ( int i = 0; i < total; ++i )
         maxdiff = std::max( maxdiff, std::fabs( hout[i] - href[i] ) );
      for ( int r = 0; r < 10; ++r )
         launch( r % nbuf );
      CK( hipEventRecord( e0 ) );
      for ( int r = 0; r < reps; ++r )
         launch( r % nbuf );
      CK( hipEventRecord( e1 ) );
      CK( hipEventSynchronize( e1 ) );
      float ms;
      CK( hipEventElapsedTime( &ms, e0, e1 ) );
      char nm[96];
      snprintf( nm, 96, "%s (tiles %d, lds %zu B)", v.name, tt.count, lds );
      report( nm, ms, maxdiff );
   }
   // ---- row-marching register kernel ----
   for ( int depth : { 1, 2, 4 } )
   for ( int rows : { 8, 16 } )
      for ( int xcd = 1; xcd < 2; ++xcd )
      {
         {
            char nm0[96];
            snprintf( nm0, 96, "rowmarch D=%d rows=%d%s", depth, rows, xcd ? " xcd" : "" );
            if ( !want( nm0 ) )
               continue;
         }
         std::vector< RowTask > tasks;
         build_row_tasks( level, rows, tasks );
         RowTask* dtasks;
         CK( hipMalloc( &dtasks, tasks.size() * sizeof( RowTask ) ) );
         CK( hipMemcpy( dtasks, tasks.data(), tasks.size() * sizeof( RowTask ), hipMemcpyHostToDevice ) );
         RowMarchArgs A{};
         A.tasks  = dtasks;
         A.ntasks = (int) tasks.size();
         A.N      = N;
         A.total  = total;
         A.st     = st;
         int nblocks = ( A.ntasks + kRowMarchWavesPerBlock - 1 ) / kRowMarchWavesPerBlock;
         nblocks     = ( nblocks + 7 ) & ~7;
         A.xcd_chunk = xcd ? nblocks / 8 : 0;
         auto launch = [&]( int b ) {
            A.dst = dst[b];
            A.src = src[b];
            auto kern = depth == 1 ? p1_apply_rowmarch_kernel< APPLY_REPLACE, 1 >
                      : depth == 2 ? p1_apply_rowmarch_kernel< APPLY_REPLACE, 2 >
                      : depth == 3 ? p1_apply_rowmarch_kernel< APPLY_REPLACE, 3 >
                      : depth == 4 ? p1_apply_rowmarch_kernel< APPLY_REPLACE, 4 >
                                   : p1_apply_rowmarch_kernel< APPLY_REPLACE, 6 >;
            hipLaunchKernelGGL( kern, dim3( nblocks ), dim3( 64 * kRowMarchWavesPerBlock ), 0, 0, A );
         };
         CK( hipMemset( dst[0], 0, (size_t) total * 8 ) );
         launch( 0 );
         CK( hipDeviceSynchronize() );
         CK( hipMemcpy( hout.data(), dst[0], (size_t) total * 8, hipMemcpyDeviceToHost ) );
         double maxdiff = 0;
         for ( int i = 0; i < total; ++i )
            maxdiff = std::max( maxdiff, std::fabs( hout[i] - href[i] ) );
         for ( int r = 0; r < 10; ++r )
            launch( r % nbuf );
         CK( hipEventRecord( e0 ) );
         for ( int r = 0; r < reps; ++r )
            launch( r % nbuf );
         CK( hipEventRecord( e1 ) );
         CK( hipEventSynchronize( e1 ) );
         float ms;
         CK( hipEventElapsedTime( &ms, e0, e1 ) );
         char nm[96];
         snprintf( nm, 96, "rowmarch D=%d rows=%d%s (tasks %d)", depth, rows, xcd ? " xcd" : "", A.ntasks );
         report( nm, ms, maxdiff );
         CK( hipFree( dtasks ) );
      }
   // ---- fully unrolled row-strip kernel ----
   for ( int rows : { 2, 3, 4, 6, 8, 12 } )
      for ( int xcd = 0; xcd < 2; ++xcd )
      {
         char nm0[96];
         snprintf( nm0, 96, "rowstrip NY=%d%s", rows, xcd ? " xcd" : "" );
         if ( !want( nm0 ) )
            continue;
         std::vector< RowTask > tasks;
         build_row_tasks( level, rows, tasks );
         RowTask* dtasks;
         CK( hipMalloc( &dtasks, tasks.size() * sizeof( RowTask ) ) );
         CK( hipMemcpy( dtasks, tasks.data(), tasks.size() * sizeof( RowTask ), hipMemcpyHostToDevice ) );
         RowMarchArgs A{};
         A.tasks  = dtasks;
         A.ntasks = (int) tasks.size();
         A.N      = N;
         A.total  = total;
         A.st     = st;
         int nblocks = ( A.ntasks + kRowMarchWavesPerBlock - 1 ) / kRowMarchWavesPerBlock;
         nblocks     = ( nblocks + 7 ) & ~7;
         A.xcd_chunk = xcd ? nblocks / 8 : 0;
         auto kern   = rows == 2   ? p1_apply_rowstrip_kernel< APPLY_REPLACE, 2 >
                       : rows == 3 ? p1_apply_rowstrip_kernel< APPLY_REPLACE, 3 >
                       : rows == 4 ? p1_apply_rowstrip_kernel< APPLY_REPLACE, 4 >
                       : rows == 6 ? p1_apply_rowstrip_kernel< APPLY_REPLACE, 6 >
                       : rows == 8 ? p1_apply_rowstrip_kernel< APPLY_REPLACE, 8 >
                                   : p1_apply_rowstrip_kernel< APPLY_REPLACE, 12 >;
         auto launch = [&]( int b ) {
            A.dst = dst[b];
            A.src = src[b];
            hipLaunchKernelGGL( kern, dim3( nblocks ), dim3( 64 * kRowMarchWavesPerBlock ), 0, 0, A );
         };
         CK( hipMemset( dst[0], 0, (size_t) total * 8 ) );
         launch( 0 );
         CK( hipDeviceSynchronize() );
         CK( hipMemcpy( hout.data(), dst[0], (size_t) total * 8, hipMemcpyDeviceToHost ) );
         double maxdiff = 0;
         for ( int i = 0; i < total; ++i )
            maxdiff = std::max( maxdiff, std::fabs( hout[i] - href[i] ) );
         for ( int r = 0; r < 10; ++r )
            launch( r % nbuf );
         CK( hipEventRecord( e0 ) );
         for ( int r = 0; r < reps; ++r )
            launch( r % nbuf );
         CK( hipEventRecord( e1 ) );
         CK( hipEventSynchronize( e1 ) );
         float ms;
         CK( hipEventElapsedTime( &ms, e0, e1 ) );
         char nm[96];
         snprintf( nm, 96, "%s (tasks %d)", nm0, A.ntasks );
         report( nm, ms, maxdiff );
         CK( hipFree( dtasks ) );
      }
   // ---- ablations of rowstrip NY=6 xcd ----
   if ( want( "ablate" ) )
   {
      std::vector< RowTask > tasks;
      build_row_tasks( level, 6, tasks );
      RowTask* dtasks;
      CK( hipMalloc( &dtasks, tasks.size() * sizeof( RowTask ) ) );
      CK( hipMemcpy( dtasks, tasks.data(), tasks.size() * sizeof( RowTask ), hipMemcpyHostToDevice ) );
      RowMarchArgs A{};
      A.tasks  = dtasks;
      A.ntasks = (int) tasks.size();
      A.N      = N;
      A.total  = total;
      A.st     = st;
      int nblocks = ( A.ntasks + kRowMarchWavesPerBlock - 1 ) / kRowMarchWavesPerBlock;
      nblocks     = ( nblocks + 7 ) & ~7;
      A.xcd_chunk = nblocks / 8;
      auto run = [&]( const char* nm, auto kern ) {
         auto launch = [&]( int b ) {
            A.dst = dst[b];
            A.src = src[b];
            hipLaunchKernelGGL( kern, dim3( nblocks ), dim3( 64 * kRowMarchWavesPerBlock ), 0, 0, A );
         };
         for ( int r = 0; r < 10; ++r )
            launch( r % nbuf );
         CK( hipEventRecord( e0 ) );
         for ( int r = 0; r < reps; ++r )
            launch( r % nbuf );
         CK( hipEventRecord( e1 ) );
         CK( hipEventSynchronize( e1 ) );
         float ms;
         CK( hipEventElapsedTime( &ms, e0, e1 ) );
         report( nm, ms, -1.0 );
      };
      run( "ablate 0 (full kernel)", p1_apply_rowstrip_kernel< APPLY_REPLACE, 6, 0 > );
      run( "ablate 1 (no up/down loads)", p1_apply_rowstrip_kernel< APPLY_REPLACE, 6, 1 > );
      run( "ablate 2 (unmasked stores)", p1_apply_rowstrip_kernel< APPLY_REPLACE, 6, 2 > );
      run( "ablate 4 (no stores)", p1_apply_rowstrip_kernel< APPLY_REPLACE, 6, 4 > );
      run( "ablate 8 (no arithmetic)", p1_apply_rowstrip_kernel< APPLY_REPLACE, 6, 8 > );
      run( "ablate 16 (aligned segments)", p1_apply_rowstrip_kernel< APPLY_REPLACE, 6, 16 > );
      run( "ablate 1+8", p1_apply_rowstrip_kernel< APPLY_REPLACE, 6, 9 > );
      run( "ablate 1+2+8", p1_apply_rowstrip_kernel< APPLY_REPLACE, 6, 11 > );
      run( "ablate 1+4+8 (loads only)", p1_apply_rowstrip_kernel< APPLY_REPLACE, 6, 13 > );
      run( "ablate 4+8 (all loads, no st)", p1_apply_rowstrip_kernel< APPLY_REPLACE, 6, 12 > );
      run( "ablate 2+16 (aligned+unmasked)", p1_apply_rowstrip_kernel< APPLY_REPLACE, 6, 18 > );
      run( "ablate 1+2+8+16", p1_apply_rowstrip_kernel< APPLY_REPLACE, 6, 27 > );
      CK( hipFree( dtasks ) );
   }

   // ---- occupancy-limited rowstrip (dynamic LDS as an occupancy throttle) ----
   if ( want( "occ" ) )
   {
      for ( int rows : { 4, 8 } )
      {
         std::vector< RowTask > tasks;
         build_row_tasks( level, rows, tasks );
         RowTask* dtasks;
         CK( hipMalloc( &dtasks, tasks.size() * sizeof( RowTask ) ) );
         CK( hipMemcpy( dtasks, tasks.data(), tasks.size() * sizeof( RowTask ), hipMemcpyHostToDevice ) );
         RowMarchArgs A{};
         A.tasks  = dtasks;
         A.ntasks = (int) tasks.size();
         A.N      = N;
         A.total  = total;
         A.st     = st;
         int nblocks = ( A.ntasks + kRowMarchWavesPerBlock - 1 ) / kRowMarchWavesPerBlock;
         nblocks     = ( nblocks + 7 ) & ~7;
         A.xcd_chunk = nblocks / 8;
         auto kern = rows == 4 ? p1_apply_rowstrip_kernel< APPLY_REPLACE, 4, 0 > : p1_apply_rowstrip_kernel< APPLY_REPLACE, 8, 0 >;
         CK( hipFuncSetAttribute( reinterpret_cast< const void* >( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 ) );
         for ( int blocks_per_cu : { 8, 6, 4, 3, 2, 1 } )
         {
            const size_t lds = blocks_per_cu >= 8 ? 0 : ( 160 * 1024 / blocks_per_cu ) & ~255;
            auto launch = [&]( int b ) {
               A.dst = dst[b];
               A.src = src[b];
               hipLaunchKernelGGL( kern, dim3( nblocks ), dim3( 64 * kRowMarchWavesPerBlock ), lds, 0, A );
            };
            for ( int r = 0; r < 10; ++r )
               launch( r % nbuf );
            CK( hipEventRecord( e0 ) );
            for ( int r = 0; r < reps; ++r )
               launch( r % nbuf );
            CK( hipEventRecord( e1 ) );
            CK( hipEventSynchronize( e1 ) );
            float ms;
            CK( hipEventElapsedTime( &ms, e0, e1 ) );
            char nm[96];
            snprintf( nm, 96, "occ rowstrip NY=%d, %d blocks/CU", rows, blocks_per_cu );
            report( nm, ms, -1.0 );
         }
         CK( hipFree( dtasks ) );
      }
   }

   // ---- strip2: two doubles per lane, buffer addressing ----
   for ( int rows : { 2, 3, 4, 6 } )
      for ( int xcd = 0; xcd < 2; ++xcd )
      {
         char nm0[96];
         snprintf( nm0, 96, "strip2 NY=%d%s", rows, xcd ? " xcd" : "" );
         if ( !want( nm0 ) )
            continue;
         std::vector< StripTask > tasks;
         build_strip_tasks( level, rows, tasks );
         StripTask* dtasks;
         CK( hipMalloc( &dtasks, tasks.size() * sizeof( StripTask ) ) );
         CK( hipMemcpy( dtasks, tasks.data(), tasks.size() * sizeof( StripTask ), hipMemcpyHostToDevice ) );
         Strip2Args A{};
         A.tasks  = dtasks;
         A.ntasks = (int) tasks.size();
         A.bytes  = (unsigned) total * 8u;
         A.st     = st;
         int nblocks = ( A.ntasks + kStripWavesPerBlock - 1 ) / kStripWavesPerBlock;
         nblocks     = ( nblocks + 7 ) & ~7;
         A.xcd_chunk = xcd ? nblocks / 8 : 0;
         auto kern   = rows == 2   ? p1_apply_strip2_kernel< APPLY_REPLACE, 2 >
                       : rows == 3 ? p1_apply_strip2_kernel< APPLY_REPLACE, 3 >
                       : rows == 4 ? p1_apply_strip2_kernel< APPLY_REPLACE, 4 >
                                   : p1_apply_strip2_kernel< APPLY_REPLACE, 6 >;
         auto launch = [&]( int b ) {
            A.dst = dst[b];
            A.src = src[b];
            hipLaunchKernelGGL( kern, dim3( nblocks ), dim3( 64 * kStripWavesPerBlock ), 0, 0, A );
         };
         CK( hipMemset( dst[0], 0, (size_t) total * 8 ) );
         launch( 0 );
         CK( hipDeviceSynchronize() );
         CK( hipMemcpy( hout.data(), dst[0], (size_t) total * 8, hipMemcpyDeviceToHost ) );
         double maxdiff = 0;
         for ( int i = 0; i < total; ++i )
            maxdiff = std::max( maxdiff, std::fabs( hout[i] - href[i] ) );
         for ( int r = 0; r < 10; ++r )
            launch( r % nbuf );
         CK( hipEventRecord( e0 ) );
         for ( int r = 0; r < reps; ++r )
            launch( r % nbuf );
         CK( hipEventRecord( e1 ) );
         CK( hipEventSynchronize( e1 ) );
         float ms;
         CK( hipEventElapsedTime( &ms, e0, e1 ) );
         char nm[96];
         snprintf( nm, 96, "%s (tasks %d)", nm0, A.ntasks );
         report( nm, ms, maxdiff );
         CK( hipFree( dtasks ) );
      }
   // ---- z-march bricks ----
   {
      auto run = [&]( int NY, int LZ, int xcd, auto kern ) {
         char nm0[96];
         snprintf( nm0, 96, "zmarch NY=%d LZ=%d%s", NY, LZ, xcd ? " xcd" : "" );
         if ( !want( nm0 ) )
            return;
         std::vector< BrickTask > tasks;
         build_brick_tasks( level, NY, LZ, tasks );
         BrickTask* dtasks;
         CK( hipMalloc( &dtasks, tasks.size() * sizeof( BrickTask ) ) );
         CK( hipMemcpy( dtasks, tasks.data(), tasks.size() * sizeof( BrickTask ), hipMemcpyHostToDevice ) );
         ZMarchArgs A{};
         A.tasks  = dtasks;
         A.ntasks = (int) tasks.size();
         A.bytes  = (unsigned) total * 8u;
         A.st     = st;
         int nblocks = ( A.ntasks + kZMarchWavesPerBlock - 1 ) / kZMarchWavesPerBlock;
         nblocks     = ( nblocks + 7 ) & ~7;
         A.xcd_chunk = xcd ? nblocks / 8 : 0;
         A.relax = 0.66;
         auto launch = [&]( int b ) {
            A.dst = dst[b];
            A.src = src[b];
            A.rhs = src[( b + 1 ) % nbuf]; // only read by the Jacobi instantiations
            hipLaunchKernelGGL( kern, dim3( nblocks ), dim3( 64 * kZMarchWavesPerBlock ), 0, 0, A );
         };
         CK( hipMemset( dst[0], 0, (size_t) total * 8 ) );
         launch( 0 );
         CK( hipDeviceSynchronize() );
         CK( hipMemcpy( hout.data(), dst[0], (size_t) total * 8, hipMemcpyDeviceToHost ) );
         double maxdiff = 0;
         for ( int i = 0; i < total; ++i )
            maxdiff = std::max( maxdiff, std::fabs( hout[i] - href[i] ) );
         for ( int r = 0; r < 10; ++r )
            launch( r % nbuf );
         CK( hipEventRecord( e0 ) );
         for ( int r = 0; r < reps; ++r )
            launch( r % nbuf );
         CK( hipEventRecord( e1 ) );
         CK( hipEventSynchronize( e1 ) );
         float ms;
         CK( hipEventElapsedTime( &ms, e0, e1 ) );
         char nm[96];
         snprintf( nm, 96, "%s (tasks %d)", nm0, A.ntasks );
         report( nm, ms, maxdiff );
         CK( hipFree( dtasks ) );
      };
      for ( int xcd = 1; xcd < 2; ++xcd )
      {
         run( 2, 4, xcd, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 4 > );
         run( 2, 8, xcd, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 8 > );
         run( 3, 6, xcd, p1_apply_zmarch_kernel< APPLY_REPLACE, 3, 6 > );
         run( 4, 4, xcd, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4 > );
         run( 4, 8, xcd, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8 > );
         run( 4, 16, xcd, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 16 > );
         run( 6, 4, xcd, p1_apply_zmarch_kernel< APPLY_REPLACE, 6, 4 > );
         run( 6, 8, xcd, p1_apply_zmarch_kernel< APPLY_REPLACE, 6, 8 > );
         run( 8, 8, xcd, p1_apply_zmarch_kernel< APPLY_REPLACE, 8, 8 > );
      }
      run( 4, 8, 0, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8 > );
      if ( want( "zfact" ) )
      {
         filter = "";
         printf( "unfactorised shifts (8 per output): NY,LZ = 4,4  4,8  2,8\n" );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 2, 0, 0 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 0 > );
         run( 2, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 8, 0, 2, 0, 0 > );
         printf( "factorised, ablations (none / no stores / no arithmetic / neither) at 4,4:\n" );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0 > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 4 > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 8 > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 12 > );
         filter = "zfact";
      }
      if ( want( "zpf" ) )
      {
         filter = "";
         printf( "all loads before the first store: NY,LZ = 4,4  4,2  2,4  4,3  2,8  4,6  (then 4,4 without stores)\n" );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 2, 0, 1, true > );
         run( 4, 2, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 2, 0, 2, 0, 1, true > );
         run( 2, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 4, 0, 2, 0, 1, true > );
         run( 4, 3, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 3, 0, 2, 0, 1, true > );
         run( 2, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 8, 0, 2, 0, 1, true > );
         run( 4, 6, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 6, 0, 2, 0, 1, true > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 4, 2, 0, true, true > );
         printf( "same with plain (aux 0) stores: 4,4  2,4\n" );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 0, 0, 1, true > );
         run( 2, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 4, 0, 0, 0, 1, true > );
         filter = "zpf";
      }
      if ( want( "zmask" ) )
      {
         filter = "";
         printf( "loads masked beyond the row end: NY,LZ = 4,4  4,8  2,8  2,4  6,4 (then 4,4: no stores / neither)\n" );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 2, 0, 1, false, true > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 1, false, true > );
         run( 2, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 8, 0, 2, 0, 1, false, true > );
         run( 2, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 4, 0, 2, 0, 1, false, true > );
         run( 6, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 6, 4, 0, 2, 0, 1, false, true > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 4, 2, 0, true, false, true > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 12, 2, 0, true, false, true > );
         filter = "zmask";
      }
      if ( want( "zilp" ) )
      {
         filter = "";
         printf( "rows evaluated side by side (FACT=2): NY,LZ = 4,4  4,8  2,8  2,4  6,4  8,4 | 4,4 masked loads | 4,4 serial (FACT=1)\n" );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 2, 0, 2 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 2 > );
         run( 2, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 8, 0, 2, 0, 2 > );
         run( 2, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 4, 0, 2, 0, 2 > );
         run( 6, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 6, 4, 0, 2, 0, 2 > );
         run( 8, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 8, 4, 0, 2, 0, 2 > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 2, 0, 2, false, true > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 2, 0, 1 > );
         filter = "zilp";
      }
      if ( want( "zfinal" ) )
      {
         filter = "";
         printf( "candidates: 4,4,F0 | 4,8,F1,mask | 4,8,F2,mask | 4,4,F1,mask | 2,8,F1,mask | 4,8,F1 | 4,6,F1,mask | 3,6,F1,mask\n" );
         for ( int rep = 0; rep < 2; ++rep )
         {
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 2, 0, 0 > );
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 1, false, true > );
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 2, false, true > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 2, 0, 1, false, true > );
            run( 2, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 8, 0, 2, 0, 1, false, true > );
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 1 > );
            run( 4, 6, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 6, 0, 2, 0, 1, false, true > );
            run( 3, 6, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 3, 6, 0, 2, 0, 1, false, true > );
         }
         filter = "zfinal";
      }
      if ( want( "zmodes" ) )
      {
         filter = "";
         printf( "ADD: 4,4,F0 | 4,4,F1,mask | 4,4,F1 | 4,8,F1,mask | 2,4,F1,mask | 4,2,F1,mask ; then the same for JACOBI (maxdiff not meaningful)\n" );
         for ( int rep = 0; rep < 2; ++rep )
         {
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 4, 0, 2, 0, 0 > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 4, 0, 2, 0, 1, false, true > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 4, 0, 2, 0, 1 > );
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 8, 0, 2, 0, 1, false, true > );
            run( 2, 4, 1, p1_apply_zmarch_kernel< APPLY_ADD, 2, 4, 0, 2, 0, 1, false, true > );
            run( 4, 2, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 2, 0, 2, 0, 1, false, true > );
         }
         for ( int rep = 0; rep < 2; ++rep )
         {
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_JACOBI, 4, 4, 0, 2, 0, 0 > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_JACOBI, 4, 4, 0, 2, 0, 1, false, true > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_JACOBI, 4, 4, 0, 2, 0, 1 > );
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_JACOBI, 4, 8, 0, 2, 0, 1, false, true > );
            run( 2, 4, 1, p1_apply_zmarch_kernel< APPLY_JACOBI, 2, 4, 0, 2, 0, 1, false, true > );
            run( 4, 2, 1, p1_apply_zmarch_kernel< APPLY_JACOBI, 4, 2, 0, 2, 0, 1, false, true > );
         }
         filter = "zmodes";
      }
      if ( want( "zaddaux" ) )
      {
         filter = "";
         printf( "ADD 4,4,F1: st nt/ld plain | st plain/ld plain | st nt/ld nt | st plain/ld nt | st sc1/ld plain | st nt/ld sc1 | st nt+sc0 / ld sc0\n" );
         for ( int rep = 0; rep < 2; ++rep )
         {
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 4, 0, 2, 0, 1, false, false, 0 > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 4, 0, 0, 0, 1, false, false, 0 > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 4, 0, 2, 0, 1, false, false, 2 > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 4, 0, 0, 0, 1, false, false, 2 > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 4, 0, 16, 0, 1, false, false, 0 > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 4, 0, 2, 0, 1, false, false, 16 > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 4, 0, 3, 0, 1, false, false, 1 > );
         }
         filter = "zaddaux";
      }
      if ( want( "zsoff" ) )
      {
         filter = "";
         printf( "instruction diet (SOFF): 4,8 shipped | 4,8 SOFF | 4,4 SOFF | 2,8 SOFF | 4,6 SOFF | 6,4 SOFF | 4,8 SOFF F2 ; twice\n" );
         for ( int rep = 0; rep < 2; ++rep )
         {
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 1, false, true > );
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 1, false, true, 0, true > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 2, 0, 1, false, true, 0, true > );
            run( 2, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 8, 0, 2, 0, 1, false, true, 0, true > );
            run( 4, 6, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 6, 0, 2, 0, 1, false, true, 0, true > );
            run( 6, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 6, 4, 0, 2, 0, 1, false, true, 0, true > );
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 2, false, true, 0, true > );
         }
         filter = "zsoff";
      }
      if ( want( "zpfd" ) )
      {
         filter = "";
         printf( "prefetch distance (slices ahead): 4,8 PFD 1 (shipped) | 2 | 3 ; 4,4 PFD 1 | 2 ; 2,8 PFD 2 ; twice\n" );
         for ( int rep = 0; rep < 2; ++rep )
         {
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 1, false, true, 0, true, 1 > );
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 1, false, true, 0, true, 2 > );
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 1, false, true, 0, true, 3 > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 2, 0, 1, false, true, 0, true, 1 > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 2, 0, 1, false, true, 0, true, 2 > );
            run( 2, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 8, 0, 2, 0, 1, false, true, 0, true, 2 > );
         }
         filter = "zpfd";
      }
      if ( want( "zabl" ) )
      {
         filter = "";
         printf( "shipped 4,8,F1,SOFF: full | no stores | no arithmetic | neither ; then the same with all loads first (PFALL); then 4,4\n" );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 1, false, true, 0, true > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 4, 2, 0, 1, false, true, 0, true > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 8, 2, 0, 1, false, true, 0, true > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 12, 2, 0, 1, false, true, 0, true > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0, 1, true, true, 0, true > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 4, 2, 0, 1, true, true, 0, true > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 8, 2, 0, 1, true, true, 0, true > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 12, 2, 0, 1, true, true, 0, true > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, 2, 0, 1, false, true, 0, true > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 4, 2, 0, 1, false, true, 0, true > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 8, 2, 0, 1, false, true, 0, true > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 12, 2, 0, 1, false, true, 0, true > );
         run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 12, 2, 0, 1, true, true, 0, true > );
         run( 2, 4, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 4, 12, 2, 0, 1, true, true, 0, true > );
         run( 2, 2, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 2, 12, 2, 0, 1, true, true, 0, true > );
         filter = "zabl";
      }
      if ( want( "zmodes2" ) )
      {
         filter = "";
         printf( "SOFF kernels, EX_AUX=2: ADD 4,4 | 4,8 | 2,8 | 4,6 ; JACOBI 4,4 | 4,8 | 2,8 | 4,6 (rhs = another src buffer: partly cache-resident)\n" );
         for ( int rep = 0; rep < 2; ++rep )
         {
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 4, 0, 2, 0, 1, false, true, 2, true > );
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 8, 0, 2, 0, 1, false, true, 2, true > );
            run( 2, 8, 1, p1_apply_zmarch_kernel< APPLY_ADD, 2, 8, 0, 2, 0, 1, false, true, 2, true > );
            run( 4, 6, 1, p1_apply_zmarch_kernel< APPLY_ADD, 4, 6, 0, 2, 0, 1, false, true, 2, true > );
            run( 4, 4, 1, p1_apply_zmarch_kernel< APPLY_JACOBI, 4, 4, 0, 2, 0, 1, false, true, 2, true > );
            run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_JACOBI, 4, 8, 0, 2, 0, 1, false, true, 2, true > );
            run( 2, 8, 1, p1_apply_zmarch_kernel< APPLY_JACOBI, 2, 8, 0, 2, 0, 1, false, true, 2, true > );
            run( 4, 6, 1, p1_apply_zmarch_kernel< APPLY_JACOBI, 4, 6, 0, 2, 0, 1, false, true, 2, true > );
         }
         filter = "zmodes2";
      }
      if ( want( "zaux" ) )
      {
         filter = "";
         printf( "store aux 0,1,2,3,16,17,18,19 (load aux 0):\n" );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 0, 0 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 1, 0 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 0 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 3, 0 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 16, 0 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 17, 0 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 18, 0 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 19, 0 > );
         printf( "load aux 1,2,16,18 (store aux 2):\n" );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 1 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 2 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 16 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, 2, 18 > );
         printf( "ablations with nt stores: none / no stores / no arithmetic / neither\n" );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 4 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 8 > );
         run( 4, 8, 1, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 12 > );
         filter = "zaux";
      }
   }
   return 0;
}
