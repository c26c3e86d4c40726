// Developer probe: effective shader clock under light load (few workgroups) vs full load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin( double* out, unsigned long long* stamps, int iters )
{
   unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
   double a = threadIdx.x * 1e-9, b = 1.0000001;
   for ( int i = 0; i < iters; ++i )
      a = fma( a, b, 1e-12 );
   unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
   if ( threadIdx.x == 0 )
   {
      stamps[2 * blockIdx.x]     = t1 - t0;
      stamps[2 * blockIdx.x + 1] = r1 - r0;
   }
   if ( a == 12345.678 )
      out[0] = a;
}
int main()
{
   double* out;
   unsigned long long* st;
   hipMalloc( &out, 8 );
   hipMalloc( &st, 16 * 8192 );
   for ( int rep = 0; rep < 2; ++rep )
      for ( int blocks : { 1, 16, 256, 2048 } )
         for ( int iters : { 2000, 20000 } )
         {
            hipEvent_t e0, e1;
            hipEventCreate( &e0 );
            hipEventCreate( &e1 );
            hipEventRecord( e0 );
            for ( int k = 0; k < 20; ++k )
               hipLaunchKernelGGL( spin, dim3( blocks ), dim3( 256 ), 0, 0, out, st, iters );
            hipEventRecord( e1 );
            hipEventSynchronize( e1 );
            float ms;
            hipEventElapsedTime( &ms, e0, e1 );
            std::vector< unsigned long long > h( 2 );
            hipMemcpy( h.data(), st, 16, hipMemcpyDeviceToHost );
            printf( "blocks %5d iters %6d: %8.2f us/launch, block0: %llu shader cycles in %llu x10ns => %.0f MHz, %.2f cycles/fma\n", blocks,
                    iters, ms * 1e3 / 20, h[0], h[1], h[0] / ( h[1] * 0.01 ), (double) h[0] / iters );
         }
   return 0;
}
