// Developer micro-benchmark (not part of the product library): times variants of the macro-cell apply kernel
// on rotating buffer pairs (working set > 256 MiB Infinity Cache) and a copy kernel of the same traffic.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHYTEG_HIP_BUILDING -o apply_bench apply_bench.hip ../runtime.hip
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <random>
#include <vector>

#include "../kernels_apply.hpp"

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

   // naive
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
   return 0;
}
