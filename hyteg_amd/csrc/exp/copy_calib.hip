// Calibration kernels for the PMC traffic passes (tools/profile_bench.sh): copies of known size with the apply kernel's
// access width (8 B per lane), plain and nontemporal stores.  FETCH_SIZE / WRITE_SIZE of these launches fix the
// counter-to-bytes factors on gfx950 (MI355X_MICROARCH.md: FETCH_SIZE reports half of a coalesced stream).
// Build: hipcc --offload-arch=gfx950 -O3 -o copy_calib copy_calib.hip      Run: copy_calib [reps]
#include <cstdio>
#include <cstdlib>

#include <hip/hip_runtime.h>

template < bool NT >
__global__ __launch_bounds__( 256 ) void calib_copy_kernel( double* __restrict__ dst, const double* __restrict__ src, int n )
{
   for ( int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256 )
   {
      const double v = src[k];
      if ( NT )
         __builtin_nontemporal_store( v, &dst[k] );
      else
         dst[k] = v;
   }
}

int main( int argc, char** argv )
{
   const int reps = argc > 1 ? atoi( argv[1] ) : 20;
   const int n    = 2862209; // one level-8 cell array: 22,897,672 B each way
   const int nbuf = 9;
   double *  src[nbuf], *dst[nbuf];
   for ( int b = 0; b < nbuf; ++b )
   {
      if ( hipMalloc( &src[b], (size_t) n * 8 ) != hipSuccess || hipMalloc( &dst[b], (size_t) n * 8 ) != hipSuccess )
         return 1;
      hipMemset( src[b], 0, (size_t) n * 8 );
   }
   for ( int r = 0; r < reps; ++r )
      hipLaunchKernelGGL( calib_copy_kernel< false >, dim3( 1024 ), dim3( 256 ), 0, 0, dst[r % nbuf], src[r % nbuf], n );
   for ( int r = 0; r < reps; ++r )
      hipLaunchKernelGGL( calib_copy_kernel< true >, dim3( 1024 ), dim3( 256 ), 0, 0, dst[r % nbuf], src[r % nbuf], n );
   if ( hipDeviceSynchronize() != hipSuccess )
      return 1;
   printf( "%d plain + %d nontemporal copies of %d doubles\n", reps, reps, n );
   return 0;
}
