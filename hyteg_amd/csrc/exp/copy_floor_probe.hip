// Which copy kernel is the fastest way to move one level-8 cell array (22.9 MB in + 22.9 MB out) on this device?
// Variants: bytes per lane (8 / 16), load policy (plain / nontemporal), store policy (plain / nontemporal), grid size,
// and hipMemcpyAsync device-to-device.  Ring of 9 pairs (412 MB > Infinity Cache), 200 launches per variant between two events.
// Build: hipcc --offload-arch=gfx950 -O3 -o copy_floor_probe copy_floor_probe.hip      Run: ./copy_floor_probe
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

typedef double v2d __attribute__( ( ext_vector_type( 2 ) ) );

template < typename V, bool NTL, bool NTS >
__global__ __launch_bounds__( 256 ) void copy_kernel( V* __restrict__ dst, const V* __restrict__ src, long n )
{
   const long stride = (long) gridDim.x * 256;
   for ( long k = (long) blockIdx.x * 256 + threadIdx.x; k < n; k += stride )
   {
      V v;
      if constexpr ( NTL )
         v = __builtin_nontemporal_load( &src[k] );
      else
         v = src[k];
      if constexpr ( NTS )
         __builtin_nontemporal_store( v, &dst[k] );
      else
         dst[k] = v;
   }
}

#define CK( x ) do { hipError_t e = ( x ); if ( e != hipSuccess ) { printf( "%s: %s\n", #x, hipGetErrorString( e ) ); return 1; } } while ( 0 )

int main()
{
   const long n = 2862209, nbuf = 9;
   const int  reps = 200;
   double *   src[nbuf], *dst[nbuf];
   for ( int b = 0; b < nbuf; ++b )
   {
      CK( hipMalloc( &src[b], ( n + 1 ) * 8 ) );
      CK( hipMalloc( &dst[b], ( n + 1 ) * 8 ) );
      CK( hipMemset( src[b], 1, n * 8 ) );
      CK( hipMemset( dst[b], 0, n * 8 ) );
   }
   hipEvent_t e0, e1;
   CK( hipEventCreate( &e0 ) );
   CK( hipEventCreate( &e1 ) );
   auto run = [&]( const char* name, auto launch ) {
      for ( int r = 0; r < 30; ++r )
         launch( r % nbuf );
      hipDeviceSynchronize();
      float best = 1e9f, sum = 0;
      for ( int rep = 0; rep < 5; ++rep )
      {
         hipEventRecord( e0, 0 );
         for ( int r = 0; r < reps; ++r )
            launch( r % nbuf );
         hipEventRecord( e1, 0 );
         hipEventSynchronize( e1 );
         float ms;
         hipEventElapsedTime( &ms, e0, e1 );
         best = ms < best ? ms : best;
         sum += ms;
      }
      printf( "%-46s  %.3f us (best region %.3f)\n", name, sum / 5 * 1e3 / reps, best * 1e3 / reps );
   };
   for ( int grid : { 512, 1024, 2048, 4096, 11181 } )
   {
      char nm[128];
      snprintf( nm, 128, "8B/lane  plain ld, plain st, grid %d", grid );
      run( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_kernel< double, false, false > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], src[b], n ); } );
      snprintf( nm, 128, "8B/lane  plain ld, nt st,    grid %d", grid );
      run( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_kernel< double, false, true > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], src[b], n ); } );
      snprintf( nm, 128, "8B/lane  nt ld,    nt st,    grid %d", grid );
      run( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_kernel< double, true, true > ), dim3( grid ), dim3( 256 ), 0, 0, dst[b], src[b], n ); } );
      snprintf( nm, 128, "16B/lane plain ld, plain st, grid %d", grid );
      run( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_kernel< v2d, false, false > ), dim3( grid ), dim3( 256 ), 0, 0, (v2d*) dst[b], (const v2d*) src[b], n / 2 ); } );
      snprintf( nm, 128, "16B/lane plain ld, nt st,    grid %d", grid );
      run( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_kernel< v2d, false, true > ), dim3( grid ), dim3( 256 ), 0, 0, (v2d*) dst[b], (const v2d*) src[b], n / 2 ); } );
      snprintf( nm, 128, "16B/lane nt ld,    nt st,    grid %d", grid );
      run( nm, [&]( int b ) { hipLaunchKernelGGL( ( copy_kernel< v2d, true, true > ), dim3( grid ), dim3( 256 ), 0, 0, (v2d*) dst[b], (const v2d*) src[b], n / 2 ); } );
   }
   run( "hipMemcpyAsync device to device", [&]( int b ) { hipMemcpyAsync( dst[b], src[b], n * 8, hipMemcpyDeviceToDevice, 0 ); } );
   return 0;
}
