// Developer harness (round 2; not part of the product library): the production z-march apply kernel at level 8,
//   * timed per variant (brick table vs decoded bricks, prefetch distance) over rotating buffer pairs, and
//   * with -DZM_DO_TRACE: per-wave timestamps (s_memrealtime, 100 MHz) at kernel entry, after the brick is known,
//     after the prologue loads are issued, when they have arrived, and at the end, plus XCC / HW ids, to see where a
//     launch spends its time (dispatch ramp, table load, first loads, march, tail).
// Build: see build_trace.sh.   Run: apply_trace [level] [reps] [nbuf]
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include <hip/hip_runtime.h>

#ifdef ZM_DO_TRACE
__device__ unsigned long long* g_trace;
#define ZM_TRACE( slot )                                                                                              \
   do                                                                                                                 \
   {                                                                                                                  \
      if ( ( slot ) == 3 )                                                                                            \
         asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );                                                           \
      const unsigned long long _t = __builtin_amdgcn_s_memrealtime();                                                 \
      if ( ( threadIdx.x & 63 ) == 0 )                                                                                \
      {                                                                                                               \
         unsigned long long* _p = g_trace + ( (size_t) blockIdx.x * kZMarchWavesPerBlock + ( threadIdx.x >> 6 ) ) * 8; \
         _p[slot]               = _t;                                                                                 \
         if ( ( slot ) == 0 )                                                                                         \
         {                                                                                                            \
            _p[6] = __builtin_amdgcn_s_getreg( 20 | ( 0 << 6 ) | ( 31 << 11 ) ); /* XCC_ID */                         \
            _p[7] = __builtin_amdgcn_s_getreg( 4 | ( 0 << 6 ) | ( 31 << 11 ) );  /* HW_ID */                          \
         }                                                                                                            \
      }                                                                                                               \
   } while ( 0 )
#endif

#include "../kernels_apply_zmarch.hpp"
#include "kernels_apply_zloop.hpp"

using namespace hyteg_hip;

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

// reference: one wave per (y,z) row, 15 direct loads per point, the same summation order as the z-march kernel is NOT
// required: compare with a tolerance
__global__ __launch_bounds__( 64 ) void apply_rows_naive( double* dst, const double* src, int N, Stencil15 st )
{
   const int z = 1 + blockIdx.y, y = 1 + blockIdx.x;
   const int W = N - z, R = W - y;
   if ( y > W - 3 )
      return;
   const int     S0 = tri( W ), Sm = tri( W + 1 );
   const int     base = slice_start( N, z ) + row_start( W, y );
   const double* w    = st.w;
   for ( int x = 1 + threadIdx.x; x <= R - 2; x += 64 )
   {
      const int i = base + x;
      double    acc;
      acc    = w[6] * src[i - 1];
      acc    = fma( w[3], src[i - Sm + W + 1], acc );
      acc    = fma( w[10], src[i + R], acc );
      acc    = fma( w[5], src[i - R], acc );
      acc    = fma( w[12], src[i + S0 - W + 1], acc );
      acc    = fma( w[1], src[i - Sm + y + 1], acc );
      acc    = fma( w[8], src[i + 1], acc );
      acc    = fma( w[13], src[i + S0 - y - 1], acc );
      acc    = fma( w[2], src[i - Sm + W], acc );
      acc    = fma( w[9], src[i + R - 1], acc );
      acc    = fma( w[4], src[i - R - 1], acc );
      acc    = fma( w[11], src[i + S0 - W], acc );
      acc    = fma( w[0], src[i - Sm + y], acc );
      acc    = fma( w[7], src[i], acc );
      acc    = fma( w[14], src[i + S0 - y], acc );
      dst[i] = acc;
   }
}

template < typename T >
static T pct( std::vector< T > v, double p )
{
   std::sort( v.begin(), v.end() );
   return v[std::min( v.size() - 1, (size_t) ( p * ( v.size() - 1 ) + 0.5 ) )];
}

int main( int argc, char** argv )
{
   const int     level = argc > 1 ? atoi( argv[1] ) : 8;
   const int     reps  = argc > 2 ? atoi( argv[2] ) : 300;
   const int     nbuf  = argc > 3 ? atoi( argv[3] ) : 9;
   const int     N     = ( 1 << level ) + 1;
   const int     total = (int) tet64( N );
   const int64_t inner = hyteg_hip_cell_inner_size( level );
   printf( "level %d  N %d  entries %d  inner %lld  buffers %d pairs (%.1f MB total)\n", level, N, total, (long long) inner,
           nbuf, nbuf * 2.0 * total * 8 / 1e6 );

   std::vector< double >                    h( total );
   std::mt19937_64                          gen( 42 );
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
   hipLaunchKernelGGL( apply_rows_naive, dim3( N, N ), dim3( 64 ), 0, 0, refout, src[0], N, st );
   CK( hipDeviceSynchronize() );
   std::vector< double > href( total ), hout( total );
   CK( hipMemcpy( href.data(), refout, (size_t) total * 8, hipMemcpyDeviceToHost ) );

   hipEvent_t e0, e1;
   CK( hipEventCreate( &e0 ) );
   CK( hipEventCreate( &e1 ) );

   auto run = [&]( const char* name, int NY, int LZ, bool dec, auto kern ) {
      std::vector< BrickTask > tasks;
      std::vector< int >       zs;
      build_brick_tasks( level, NY, LZ, tasks, &zs );
      if ( dec && ( (int) zs.size() - 1 > kZMarchMaxZChunks || N - 4 > 62 * kZMarchMaxStairs ) )
      {
         printf( "%-40s not decodable at this level\n", name );
         return;
      }
      BrickTask* dtasks;
      CK( hipMalloc( &dtasks, tasks.size() * sizeof( BrickTask ) ) );
      CK( hipMemcpy( dtasks, tasks.data(), tasks.size() * sizeof( BrickTask ), hipMemcpyHostToDevice ) );
      ZMarchArgs A{};
      A.tasks  = dtasks;
      A.ntasks = (int) tasks.size();
      A.bytes  = (unsigned) total * 8u;
      A.N      = N;
      A.st     = st;
      A.relax  = 0.66;
      for ( int k = 0; k < kZMarchMaxZChunks; ++k )
         A.zs[k] = k + 1 < (int) zs.size() ? zs[k] : A.ntasks;
      int nblocks = ( A.ntasks + kZMarchWavesPerBlock - 1 ) / kZMarchWavesPerBlock;
      nblocks     = ( nblocks + 7 ) & ~7;
      A.xcd_chunk = nblocks / 8;
#ifdef ZM_DO_TRACE
      const size_t        nw = (size_t) nblocks * kZMarchWavesPerBlock;
      unsigned long long* dtr;
      CK( hipMalloc( &dtr, nw * 64 ) );
      CK( hipMemset( dtr, 0, nw * 64 ) );
      CK( hipMemcpyToSymbol( HIP_SYMBOL( g_trace ), &dtr, sizeof( dtr ) ) );
#endif
      auto launch = [&]( int b ) {
         A.dst = dst[b];
         A.src = src[b];
         A.rhs = src[( b + 1 ) % nbuf];
         hipLaunchKernelGGL( kern, dim3( nblocks ), dim3( 64 * kZMarchWavesPerBlock ), 0, 0, A );
      };
      CK( hipMemset( dst[0], 0, (size_t) total * 8 ) );
      launch( 0 );
      CK( hipDeviceSynchronize() );
      CK( hipMemcpy( hout.data(), dst[0], (size_t) total * 8, hipMemcpyDeviceToHost ) );
      double maxdiff = 0;
      for ( int i = 0; i < total; ++i )
         maxdiff = std::max( maxdiff, std::fabs( hout[i] - href[i] ) );
      for ( int r = 0; r < 2 * nbuf; ++r )
         launch( r % nbuf );
      CK( hipEventRecord( e0 ) );
      for ( int r = 0; r < reps; ++r )
         launch( r % nbuf );
      CK( hipEventRecord( e1 ) );
      CK( hipEventSynchronize( e1 ) );
      float ms;
      CK( hipEventElapsedTime( &ms, e0, e1 ) );
      const double us = ms * 1e3 / reps;
      printf( "%-40s tasks %5d  %7.2f us/launch  %7.1f GDoF/s  %7.1f GB/s(16B/DoF)  maxdiff %.2e\n", name, A.ntasks, us,
              inner / us * 1e-3, 16.0 * inner / us * 1e-3, maxdiff );
      fflush( stdout );
#ifdef ZM_DO_TRACE
      // the last launch of the timed loop left its timestamps in dtr (every launch overwrites them)
      std::vector< unsigned long long > tr( nw * 8 );
      CK( hipMemcpy( tr.data(), dtr, nw * 64, hipMemcpyDeviceToHost ) );
      unsigned long long t00 = ~0ull, tend = 0;
      for ( size_t w = 0; w < nw; ++w )
         if ( tr[w * 8 + 4] )
         {
            t00  = std::min( t00, tr[w * 8] );
            tend = std::max( tend, tr[w * 8 + 4] );
         }
      std::vector< double > start, dtask, dissue, darrive, dmarch, dtotal, endt;
      std::vector< double > perx_end[8], perx_start[8];
      for ( size_t w = 0; w < nw; ++w )
      {
         const unsigned long long* p = &tr[w * 8];
         if ( !p[4] )
            continue;
         start.push_back( ( p[0] - t00 ) * 0.01 );
         dtask.push_back( ( p[1] - p[0] ) * 0.01 );
         dissue.push_back( ( p[2] - p[1] ) * 0.01 );
         darrive.push_back( ( p[3] - p[2] ) * 0.01 );
         dmarch.push_back( ( p[4] - p[3] ) * 0.01 );
         dtotal.push_back( ( p[4] - p[0] ) * 0.01 );
         endt.push_back( ( p[4] - t00 ) * 0.01 );
         perx_end[p[6] & 7].push_back( ( p[4] - t00 ) * 0.01 );
         perx_start[p[6] & 7].push_back( ( p[0] - t00 ) * 0.01 );
      }
      auto line = [&]( const char* what, std::vector< double >& v ) {
         printf( "     %-34s min %6.2f  p10 %6.2f  med %6.2f  p90 %6.2f  max %6.2f us   (%zu waves)\n", what, pct( v, 0.0 ),
                 pct( v, 0.1 ), pct( v, 0.5 ), pct( v, 0.9 ), pct( v, 1.0 ), v.size() );
      };
      printf( "   trace of one launch: first wave start -> last wave end %.2f us\n", ( tend - t00 ) * 0.01 );
      line( "wave start after first start", start );
      line( "entry -> brick known", dtask );
      line( "brick known -> prologue issued", dissue );
      line( "prologue issued -> arrived", darrive );
      line( "arrived -> last store issued", dmarch );
      line( "wave lifetime", dtotal );
      line( "wave end after first start", endt );
      for ( int x = 0; x < 8; ++x )
         if ( !perx_end[x].empty() )
            printf( "     XCC %d: %4zu waves, starts med %5.2f max %5.2f, ends med %5.2f max %5.2f\n", x, perx_end[x].size(),
                    pct( perx_start[x], 0.5 ), pct( perx_start[x], 1.0 ), pct( perx_end[x], 0.5 ), pct( perx_end[x], 1.0 ) );
      // waves per CU (HW_ID: cu_id bits 8-11, sh_id 12, se_id 13-15 on gfx9) -- how evenly the launch fills the chip
      {
         std::vector< int > percu( 8 * 64, 0 );
         for ( size_t w = 0; w < nw; ++w )
            if ( tr[w * 8 + 4] )
               percu[( tr[w * 8 + 6] & 7 ) * 64 + ( ( tr[w * 8 + 7] >> 8 ) & 63 )]++;
         int used = 0, mx = 0, mn = 1 << 30;
         for ( int c : percu )
            if ( c )
            {
               ++used;
               mx = std::max( mx, c );
               mn = std::min( mn, c );
            }
         printf( "     %d distinct (XCC, HW_ID[13:8]) slots hold waves: min %d max %d waves per slot\n", used, mn, mx );
      }
      // timeline: waves running at time t
      {
         const double T = ( tend - t00 ) * 0.01;
         printf( "     running waves at t = " );
         for ( double t = 0.5; t < T; t += 0.5 )
         {
            int n = 0;
            for ( size_t i = 0; i < start.size(); ++i )
               n += ( start[i] <= t && endt[i] > t );
            printf( "%.1f:%d ", t, n );
         }
         printf( "\n" );
      }
      CK( hipFree( dtr ) );
#endif
      CK( hipFree( dtasks ) );
   };

   run( "zmarch 4x8 table PFD1", 4, 8, false, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, false, 1 > );
   run( "zmarch 4x8 table PFD2", 4, 8, false, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, false, 2 > );
   run( "zmarch 4x8 decode PFD1", 4, 8, true, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 8, 0, true, 1 > );
   run( "zmarch 4x4 table PFD1", 4, 4, false, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 4, 0, false, 1 > );
   run( "zmarch 4x6 table PFD1", 4, 6, false, p1_apply_zmarch_kernel< APPLY_REPLACE, 4, 6, 0, false, 1 > );
   run( "zmarch 2x8 table PFD1", 2, 8, false, p1_apply_zmarch_kernel< APPLY_REPLACE, 2, 8, 0, false, 1 > );
   run( "zmarch 3x8 table PFD1", 3, 8, false, p1_apply_zmarch_kernel< APPLY_REPLACE, 3, 8, 0, false, 1 > );
   run( "zloop 4x8 renaming(4)", 4, 8, false, p1_apply_zloop_kernel< APPLY_REPLACE, 4, 4 > );
   return 0;
}
