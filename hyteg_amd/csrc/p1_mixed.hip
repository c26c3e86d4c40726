// Mixed-precision support of the C-ABI: conversions between double and float arrays and the correction update
// y (double) += alpha * x (float).  The reference converts between function precisions with
// VertexDoFFunction< ValueType >::copyFrom( VertexDoFFunction< otherValueType > ) (VertexDoFFunction.hpp:598-650, used by
// tests/hyteg/mixedPrecision/basicMixedPrecisionTest.cpp); here the arrays are flat device arrays.
#include "common.hpp"

using namespace hyteg_hip;

namespace {
constexpr int kThreads = 256;

template < typename D, typename S >
__global__ __launch_bounds__( kThreads ) void convert_kernel( D* __restrict__ dst, const S* __restrict__ src, size_t n )
{
   for ( size_t k = (size_t) blockIdx.x * kThreads + threadIdx.x; k < n; k += (size_t) gridDim.x * kThreads )
      dst[k] = (D) src[k];
}
__global__ __launch_bounds__( kThreads ) void axpy_f32_f64_kernel( double* __restrict__ y, const float* __restrict__ x, double alpha, size_t n )
{
   for ( size_t k = (size_t) blockIdx.x * kThreads + threadIdx.x; k < n; k += (size_t) gridDim.x * kThreads )
      y[k] = fma( alpha, (double) x[k], y[k] );
}
inline int blocks_for( size_t n )
{
   const size_t b = ( n + kThreads - 1 ) / kThreads;
   return (int) ( b < 1 ? 1 : ( b > 4096 ? 4096 : b ) );
}
} // namespace

extern "C" {

HYTEG_HIP_API int hyteg_hip_convert_f64_to_f32( float* dst, const double* src, size_t n, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && src, "convert_f64_to_f32: null pointer" );
   if ( n == 0 )
      return HYTEG_HIP_OK;
   hipLaunchKernelGGL( ( convert_kernel< float, double > ), dim3( blocks_for( n ) ), dim3( kThreads ), 0, as_stream( stream ), dst, src, n );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_convert_f32_to_f64( double* dst, const float* src, size_t n, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && src, "convert_f32_to_f64: null pointer" );
   if ( n == 0 )
      return HYTEG_HIP_OK;
   hipLaunchKernelGGL( ( convert_kernel< double, float > ), dim3( blocks_for( n ) ), dim3( kThreads ), 0, as_stream( stream ), dst, src, n );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_axpy_f32_into_f64( double* y, const float* x, double alpha, size_t n, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( y && x, "axpy_f32_into_f64: null pointer" );
   if ( n == 0 )
      return HYTEG_HIP_OK;
   hipLaunchKernelGGL( axpy_f32_f64_kernel, dim3( blocks_for( n ) ), dim3( kThreads ), 0, as_stream( stream ), y, x, alpha, n );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}
}
