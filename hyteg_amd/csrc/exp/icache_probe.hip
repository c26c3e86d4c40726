// Developer probe (round 2): what does straight-line code cost when every launch starts with a cold instruction cache?
// A kernel of K independent f64 FMAs per wave (8 accumulators, no memory traffic), one wave per SIMD, is timed
//   (a) back to back (its code stays in L2 between launches),
//   (b) alternating with a 46 MB nontemporal copy that streams through L2 (the situation of the apply kernel in a
//       smoother or in bench.py: by the next launch the code has been evicted from L2),
// for K = 256 ... 8192 (2 KB ... 64 KB of code), straight-line and as a loop over a 256-instruction body.
// Ideal: K x 4 cycles (a wave64 f64 FMA occupies the SIMD for 4 cycles).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -o icache_probe icache_probe.hip
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime.h>

#define CK( e )                                                                                \
   do                                                                                          \
   {                                                                                           \
      hipError_t _e = ( e );                                                                   \
      if ( _e != hipSuccess )                                                                  \
      {                                                                                        \
         fprintf( stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString( _e ) ); \
         exit( 1 );                                                                            \
      }                                                                                        \
   } while ( 0 )

template < int K, bool LOOP >
__global__ __launch_bounds__( 256 ) void code_kernel( double* out, double a, double b )
{
   double acc[8];
#pragma unroll
   for ( int i = 0; i < 8; ++i )
      acc[i] = a + i;
   if constexpr ( LOOP )
   {
#pragma nounroll
      for ( int it = 0; it < K / 256; ++it )
      {
#pragma unroll
         for ( int k = 0; k < 256; ++k )
            asm volatile( "v_fma_f64 %0, %0, %1, %2" : "+v"( acc[k & 7] ) : "v"( b ), "v"( a ) );
      }
   }
   else
   {
#pragma unroll
      for ( int k = 0; k < K; ++k )
         asm volatile( "v_fma_f64 %0, %0, %1, %2" : "+v"( acc[k & 7] ) : "v"( b ), "v"( a ) );
   }
   double s = 0;
#pragma unroll
   for ( int i = 0; i < 8; ++i )
      s += acc[i];
   if ( s == 1.2345e-300 )
      out[threadIdx.x] = s;
}

typedef double __attribute__( ( ext_vector_type( 2 ) ) ) d2;
__global__ __launch_bounds__( 256 ) void flush_copy( d2* __restrict__ dst, const d2* __restrict__ src, int n )
{
   for ( int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256 )
   {
      d2 v = src[k];
      __builtin_nontemporal_store( v, &dst[k] );
   }
}

int main()
{
   const int n = 2862209 / 2;
   const int nbuf = 9;
   std::vector< d2* > src( nbuf ), dst( nbuf );
   for ( int b = 0; b < nbuf; ++b )
   {
      CK( hipMalloc( &src[b], (size_t) n * 16 ) );
      CK( hipMalloc( &dst[b], (size_t) n * 16 ) );
      CK( hipMemset( src[b], 0, (size_t) n * 16 ) );
   }
   double* out;
   CK( hipMalloc( &out, 4096 ) );
   hipEvent_t e0, e1;
   CK( hipEventCreate( &e0 ) );
   CK( hipEventCreate( &e1 ) );
   const int reps = 200;
   auto      timeit = [&]( auto&& body ) {
      for ( int r = 0; r < 10; ++r )
         body( r );
      CK( hipEventRecord( e0 ) );
      for ( int r = 0; r < reps; ++r )
         body( r );
      CK( hipEventRecord( e1 ) );
      CK( hipEventSynchronize( e1 ) );
      float ms;
      CK( hipEventElapsedTime( &ms, e0, e1 ) );
      return ms * 1e3 / reps;
   };
   auto flush = [&]( int r ) {
      hipLaunchKernelGGL( flush_copy, dim3( 2048 ), dim3( 256 ), 0, 0, dst[r % nbuf], src[r % nbuf], n );
   };
   const double t_flush = timeit( flush );
   printf( "flush copy (46 MB, nontemporal stores) alone: %.2f us\n", t_flush );
   printf( "%-28s %10s %12s %12s %12s\n", "kernel (256 WGs x 4 waves)", "code KB", "ideal us", "hot us", "cold us" );
   auto probe = [&]( const char* name, int K, auto kern ) {
      auto code = [&]( int ) { hipLaunchKernelGGL( kern, dim3( 256 ), dim3( 256 ), 0, 0, out, 1.0, 1.0000001 ); };
      const double hot  = timeit( code );
      const double both = timeit( [&]( int r ) {
         flush( r );
         code( r );
      } );
      printf( "%-28s %10.1f %12.2f %12.2f %12.2f\n", name, K * 8 / 1024.0, K * 4 / 2400.0, hot, both - t_flush );
      fflush( stdout );
   };
   probe( "straight K=256", 256, code_kernel< 256, false > );
   probe( "straight K=512", 512, code_kernel< 512, false > );
   probe( "straight K=1024", 1024, code_kernel< 1024, false > );
   probe( "straight K=2048", 2048, code_kernel< 2048, false > );
   probe( "straight K=4096", 4096, code_kernel< 4096, false > );
   probe( "straight K=8192", 8192, code_kernel< 8192, false > );
   probe( "loop 256 x K/256, K=1024", 1024, code_kernel< 1024, true > );
   probe( "loop 256 x K/256, K=2048", 2048, code_kernel< 2048, true > );
   probe( "loop 256 x K/256, K=4096", 4096, code_kernel< 4096, true > );
   probe( "loop 256 x K/256, K=8192", 8192, code_kernel< 8192, true > );
   return 0;
}
