// Probe: cost per dependent kernel launch, stream launches vs one captured hipGraph (development only).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#define CK( e )                                                                    \
   do                                                                              \
   {                                                                               \
      hipError_t _e = ( e );                                                       \
      if ( _e != hipSuccess )                                                      \
      {                                                                            \
         printf( "%s: %s\n", #e, hipGetErrorString( _e ) );                        \
         return 1;                                                                 \
      }                                                                            \
   } while ( 0 )

__global__ void bump( double* x, int n )
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if ( i < n )
      x[i] = x[i] * 1.0000001 + 1.0;
}

int main()
{
   hipStream_t s;
   CK( hipStreamCreate( &s ) );
   for ( int n : { 64, 6545, 366145 } )
   {
      double* x;
      CK( hipMalloc( &x, n * sizeof( double ) ) );
      CK( hipMemset( x, 0, n * sizeof( double ) ) );
      const int chain = 200, reps = 20;
      const int blocks = ( n + 255 ) / 256;
      auto      run    = [&]() {
         for ( int k = 0; k < chain; ++k )
            hipLaunchKernelGGL( bump, dim3( blocks ), dim3( 256 ), 0, s, x, n );
      };
      run();
      CK( hipStreamSynchronize( s ) );
      auto t0 = std::chrono::steady_clock::now();
      for ( int r = 0; r < reps; ++r )
         run();
      CK( hipStreamSynchronize( s ) );
      const double us_stream = std::chrono::duration< double, std::micro >( std::chrono::steady_clock::now() - t0 ).count() / ( reps * chain );

      hipGraph_t     g;
      hipGraphExec_t ge;
      CK( hipStreamBeginCapture( s, hipStreamCaptureModeThreadLocal ) );
      run();
      CK( hipStreamEndCapture( s, &g ) );
      CK( hipGraphInstantiate( &ge, g, nullptr, nullptr, 0 ) );
      CK( hipGraphLaunch( ge, s ) );
      CK( hipStreamSynchronize( s ) );
      t0 = std::chrono::steady_clock::now();
      for ( int r = 0; r < reps; ++r )
         CK( hipGraphLaunch( ge, s ) );
      CK( hipStreamSynchronize( s ) );
      const double us_graph = std::chrono::duration< double, std::micro >( std::chrono::steady_clock::now() - t0 ).count() / ( reps * chain );
      printf( "n = %7d (%4d blocks): %.2f us per dependent launch on a stream, %.2f us per node in a graph of %d\n", n, blocks, us_stream,
              us_graph, chain );
      CK( hipGraphExecDestroy( ge ) );
      CK( hipGraphDestroy( g ) );
      CK( hipFree( x ) );
   }
   return 0;
}
