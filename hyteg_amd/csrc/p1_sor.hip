// C-ABI entry point: in-place SOR / Gauss-Seidel sweep on one macro-cell in the reference's
// lexicographic order, executed as hyperplanes t = x + 2y + 3z.
//
// Why this is exact: the reference updates points in (z, y, x) order.  A point reads already-updated
// values from W(-1,0,0) S(0,-1,0) SE(1,-1,0) BC(0,0,-1) BE(1,0,-1) BN(0,1,-1) BNW(-1,1,-1), whose
// t is smaller by 1,2,1,3,2,1,2, and not-yet-updated values from the 7 opposite neighbours, whose t is
// larger by the same amounts.  Points on one plane never read each other, so all points of a plane can
// be updated concurrently and planes in increasing t reproduce the sequential sweep bit for bit (up to
// FMA contraction).  The backward sweep is the same planes in decreasing t.
#include "common.hpp"

using namespace hyteg_hip;

namespace {

constexpr int kThreads = 64;

struct SorArgs
{
   double*       u;
   const double* rhs;
   int           N; // width
   int           t; // hyperplane
   double        relax;
   double        one_minus_relax;
   double        invc;
   Stencil15     st;
};

// one workgroup per z; threads over the y range of the plane inside that slice
__global__ __launch_bounds__( kThreads ) void p1_sor_plane_kernel( const SorArgs A )
{
   const int n = A.N - 1; // 2^level
   const int z = 1 + blockIdx.x;
   const int t = A.t;
   // x = t - 2y - 3z >= 1  and  x + y + z = t - y - 2z <= n - 1
   int       ylo = t - 2 * z - n + 1;
   ylo           = ylo < 1 ? 1 : ylo;
   const int num = t - 3 * z - 1;
   if ( num < 2 )
      return;
   const int yhi = num >> 1;
   const int W   = A.N - z;
   const int S0  = tri( W );
   const int Sm  = tri( W + 1 );
   const int s0  = slice_start( A.N, z );
   const double* w = A.st.w;
   for ( int y = ylo + threadIdx.x; y <= yhi; y += kThreads )
   {
      const int x = t - 2 * y - 3 * z;
      const int R = W - y;
      const int i = s0 + row_start( W, y ) + x;
      double*   u = A.u;
      // 14 terms -(w_k u_k) in the reference's order (sor_3D_macrocell_P1.cpp:74), rhs last
      double acc = -w[3] * u[i - Sm + W + 1];       // BN
      acc        = fma( -w[10], u[i + R], acc );    // N
      acc        = fma( -w[5], u[i - R], acc );     // SE
      acc        = fma( -w[12], u[i + S0 - W + 1], acc ); // TSE
      acc        = fma( -w[1], u[i - Sm + y + 1], acc );  // BE
      acc        = fma( -w[8], u[i + 1], acc );     // E
      acc        = fma( -w[6], u[i - 1], acc );     // W
      acc        = fma( -w[13], u[i + S0 - y - 1], acc ); // TW
      acc        = fma( -w[2], u[i - Sm + W], acc );      // BNW
      acc        = fma( -w[9], u[i + R - 1], acc );       // NW
      acc        = fma( -w[4], u[i - R - 1], acc );       // S
      acc        = fma( -w[11], u[i + S0 - W], acc );     // TS
      acc        = fma( -w[0], u[i - Sm + y], acc );      // BC
      acc        = fma( -w[14], u[i + S0 - y], acc );     // TC
      acc        = acc + A.rhs[i];
      u[i]       = A.relax * A.invc * acc + A.one_minus_relax * u[i];
   }
}

} // namespace

extern "C" {

HYTEG_HIP_API int hyteg_hip_p1_sor_cell( double*            u,
                                         const double*      rhs,
                                         int                level,
                                         const double*      w,
                                         double             relax,
                                         int                backwards,
                                         hyteg_hip_stream_t stream )
{
   HH_REQUIRE( u && rhs && w, "p1_sor_cell: null pointer" );
   HH_REQUIRE( level_ok( level ), "p1_sor_cell: level out of range [2,11]" );
   HH_REQUIRE( u != rhs, "p1_sor_cell: u and rhs must not alias" );
   HH_REQUIRE( w[7] != 0.0, "p1_sor_cell: zero centre weight" );
   SorArgs A;
   A.u               = u;
   A.rhs             = rhs;
   A.N               = ( 1 << level ) + 1;
   A.relax           = relax;
   A.one_minus_relax = 1.0 + ( -relax );
   A.invc            = 1.0 / w[7];
   for ( int k = 0; k < 15; ++k )
      A.st.w[k] = w[k];
   const int n = 1 << level;
   // interior: x,y,z >= 1, x+y+z <= n-1  =>  t from 6 to max over the interior of x+2y+3z = 3(n-1)-3 (x=y=1, z=n-3)
   const int tmin = 6, tmax = 1 + 2 + 3 * ( n - 3 );
   if ( tmax < tmin )
      return HYTEG_HIP_OK;
   for ( int k = 0; k <= tmax - tmin; ++k )
   {
      A.t = backwards ? tmax - k : tmin + k;
      // slices that can hold a point of this plane: 3z <= t - 3
      int nz = ( A.t - 3 ) / 3;
      nz     = nz > n - 3 ? n - 3 : nz;
      if ( nz < 1 )
         continue;
      hipLaunchKernelGGL( p1_sor_plane_kernel, dim3( nz ), dim3( kThreads ), 0, as_stream( stream ), A );
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}
}
