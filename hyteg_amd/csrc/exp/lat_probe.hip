// Developer probe (round 2): what ONE wave per SIMD can issue on gfx950 -- the numbers behind the step time of the
// blocked SOR kernel.  Shader cycles (s_memtime) per operation for
//   * f64 FMA with K = 1, 2, 4, 8 independent accumulators (one wave; and 2 / 4 waves on the same SIMD),
//   * a dependent chain of ds_read_b64 (LDS latency), K independent ds_read_b64 per iteration,
//   * ds_write -> s_barrier -> ds_read round trips of a 256-thread workgroup.
// Build: hipcc --offload-arch=gfx950 -O3 -o lat_probe lat_probe.hip     Run: lat_probe
#include <cstdio>
#include <cstdlib>
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

template < int K >
__global__ void fma_probe( double* out, unsigned long long* cyc, int iters, double a, double b )
{
   double acc[K];
#pragma unroll
   for ( int k = 0; k < K; ++k )
      acc[k] = threadIdx.x + k;
   const unsigned long long t0 = __builtin_readcyclecounter();
   for ( int i = 0; i < iters; ++i )
   {
#pragma unroll
      for ( int r = 0; r < 8; ++r )
#pragma unroll
         for ( int k = 0; k < K; ++k )
            acc[k] = fma( acc[k], a, b );
   }
   const unsigned long long t1 = __builtin_readcyclecounter();
   double                   s  = 0;
#pragma unroll
   for ( int k = 0; k < K; ++k )
      s += acc[k];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
   if ( threadIdx.x == 0 && blockIdx.x == 0 )
      cyc[0] = t1 - t0;
}

template < int K >
__global__ void fma32_probe( float* out, unsigned long long* cyc, int iters, float a, float b )
{
   float acc[K];
#pragma unroll
   for ( int k = 0; k < K; ++k )
      acc[k] = threadIdx.x + k;
   const unsigned long long t0 = __builtin_readcyclecounter();
   for ( int i = 0; i < iters; ++i )
   {
#pragma unroll
      for ( int r = 0; r < 8; ++r )
#pragma unroll
         for ( int k = 0; k < K; ++k )
            acc[k] = fmaf( acc[k], a, b );
   }
   const unsigned long long t1 = __builtin_readcyclecounter();
   float                    s  = 0;
#pragma unroll
   for ( int k = 0; k < K; ++k )
      s += acc[k];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
   if ( threadIdx.x == 0 && blockIdx.x == 0 )
      cyc[0] = t1 - t0;
}

// dependent LDS reads: lds[i] holds the next index
__global__ void lds_chase( int* out, unsigned long long* cyc, int iters )
{
   __shared__ long long lds[1024];
   for ( int i = threadIdx.x; i < 1024; i += blockDim.x )
      lds[i] = ( i * 17 + 5 ) & 1023;
   __syncthreads();
   long long                idx = threadIdx.x;
   const unsigned long long t0  = __builtin_readcyclecounter();
   for ( int i = 0; i < iters; ++i )
   {
#pragma unroll
      for ( int r = 0; r < 8; ++r )
         idx = lds[idx];
   }
   const unsigned long long t1 = __builtin_readcyclecounter();
   out[threadIdx.x]            = (int) idx;
   if ( threadIdx.x == 0 )
      cyc[0] = t1 - t0;
}

// K independent b64 reads per iteration, consumed by an integer add (no f64 arithmetic)
template < int K >
__global__ void lds_tput( long long* out, unsigned long long* cyc, int iters )
{
   __shared__ long long lds[4096];
   for ( int i = threadIdx.x; i < 4096; i += blockDim.x )
      lds[i] = i;
   __syncthreads();
   long long                s  = 0;
   int                      o  = threadIdx.x * 17;
   const unsigned long long t0 = __builtin_readcyclecounter();
   for ( int i = 0; i < iters; ++i )
   {
#pragma unroll
      for ( int k = 0; k < K; ++k )
         s += lds[( o + k * 331 ) & 4095];
      o += 7;
   }
   const unsigned long long t1 = __builtin_readcyclecounter();
   out[threadIdx.x]            = s;
   if ( threadIdx.x == 0 )
      cyc[0] = t1 - t0;
}

// write -> barrier -> read of the neighbour's value: the exchange of one SOR step without its arithmetic
__global__ void barrier_roundtrip( double* out, unsigned long long* cyc, int iters )
{
   __shared__ double lds[512];
   lds[threadIdx.x] = threadIdx.x;
   __syncthreads();
   double                   v  = threadIdx.x;
   const unsigned long long t0 = __builtin_readcyclecounter();
   for ( int i = 0; i < iters; ++i )
   {
      lds[threadIdx.x] = v;
      __syncthreads();
      v = lds[( threadIdx.x + 65 ) % blockDim.x];
      __syncthreads();
   }
   const unsigned long long t1 = __builtin_readcyclecounter();
   out[threadIdx.x]            = v;
   if ( threadIdx.x == 0 )
      cyc[0] = t1 - t0;
}

int main()
{
   double*             out;
   unsigned long long* cyc;
   CK( hipMalloc( &out, 1 << 20 ) );
   CK( hipMalloc( &cyc, 64 ) );
   const int iters = 2000;
   auto      get   = [&]() {
      unsigned long long h;
      CK( hipDeviceSynchronize() );
      CK( hipMemcpy( &h, cyc, 8, hipMemcpyDeviceToHost ) );
      return (double) h;
   };
#define RUN_FMA( K, threads )                                                                                           \
   hipLaunchKernelGGL( fma_probe< K >, dim3( 1 ), dim3( threads ), 0, 0, out, cyc, iters, 1.0000001, 1e-9 );            \
   printf( "f64 fma, %d independent chains, %3d threads (%d wave(s) per SIMD): %6.2f cycles per wave-FMA\n", K, threads, \
           ( threads + 255 ) / 256, get() / ( iters * 8.0 * K ) );
   RUN_FMA( 1, 64 )
   RUN_FMA( 2, 64 )
   RUN_FMA( 4, 64 )
   RUN_FMA( 8, 64 )
   RUN_FMA( 1, 256 )
   RUN_FMA( 4, 256 )
   RUN_FMA( 1, 512 )
   RUN_FMA( 4, 512 )
   RUN_FMA( 1, 1024 )
   RUN_FMA( 4, 1024 )
#define RUN_FMA32( K, threads )                                                                                          \
   hipLaunchKernelGGL( fma32_probe< K >, dim3( 1 ), dim3( threads ), 0, 0, (float*) out, cyc, iters, 1.0000001f, 1e-9f ); \
   printf( "f32 fma, %d independent chains, %3d threads: %6.2f cycles per wave-FMA\n", K, threads, get() / ( iters * 8.0 * K ) );
   RUN_FMA32( 1, 64 )
   RUN_FMA32( 4, 64 )
   RUN_FMA32( 8, 64 )
   hipLaunchKernelGGL( lds_chase, dim3( 1 ), dim3( 64 ), 0, 0, (int*) out, cyc, iters );
   printf( "dependent ds_read_b64 (1 wave): %6.1f cycles per read\n", get() / ( iters * 8.0 ) );
   hipLaunchKernelGGL( lds_chase, dim3( 1 ), dim3( 256 ), 0, 0, (int*) out, cyc, iters );
   printf( "dependent ds_read_b64 (4 waves, one per SIMD): %6.1f cycles per read\n", get() / ( iters * 8.0 ) );
   hipLaunchKernelGGL( lds_tput< 4 >, dim3( 1 ), dim3( 256 ), 0, 0, (long long*) out, cyc, iters );
   printf( "4 independent ds_read_b64 per iteration (4 waves): %6.1f cycles per iteration\n", get() / iters );
   hipLaunchKernelGGL( lds_tput< 14 >, dim3( 1 ), dim3( 256 ), 0, 0, (long long*) out, cyc, iters );
   printf( "14 independent ds_read_b64 per iteration (4 waves): %6.1f cycles per iteration\n", get() / iters );
   hipLaunchKernelGGL( barrier_roundtrip, dim3( 1 ), dim3( 256 ), 0, 0, out, cyc, iters );
   printf( "ds_write, barrier, ds_read, barrier (256 threads): %6.1f cycles per round\n", get() / iters );
   hipLaunchKernelGGL( barrier_roundtrip, dim3( 1 ), dim3( 64 ), 0, 0, out, cyc, iters );
   printf( "ds_write, barrier, ds_read, barrier (64 threads): %6.1f cycles per round\n", get() / iters );
   return 0;
}
