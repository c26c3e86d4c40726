// Developer harness (round 2; not part of the product library): the production blocked SOR kernel with per-block
// timestamps (s_memrealtime, 100 MHz) at entry, after the row bases, after the staging loads have arrived, when the
// block is in LDS, after the 46 plane steps and after the write-back.  Build: build_trace.sh.  Run: sor_trace [level]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime.h>

__device__ unsigned long long* g_sor_trace;
#define SOR_TRACE( slot )                                                                          \
   do                                                                                              \
   {                                                                                               \
      if ( ( slot ) == 2 )                                                                         \
         asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );                                        \
      const unsigned long long _t = __builtin_amdgcn_s_memrealtime();                              \
      if ( threadIdx.x == 0 )                                                                      \
         g_sor_trace[( ( (size_t) blk.P * 32 + blk.Q ) * 32 + blk.R ) * 8 + ( slot )] = _t;        \
   } while ( 0 )

// with -DSOR_STEPS: where the cycles of one plane step go (wave 0 of every block; s_memtime, shader clock).  The stamps
// wait for all outstanding LDS traffic, so they serialise what the production step overlaps: read them as the cost of
// the pieces, not as a decomposition of the production step.
#ifdef SOR_STEPS
__device__ unsigned long long g_step_acc[8];
#define SOR_STEP_STAMP( k )                                                        \
   do                                                                              \
   {                                                                               \
      if ( ( k ) == 4 )                                                            \
         _st_prev = __builtin_readcyclecounter();                                  \
      else                                                                         \
      {                                                                            \
         asm volatile( "s_waitcnt lgkmcnt(0)" ::: "memory" );                      \
         const unsigned long long _n = __builtin_readcyclecounter();               \
         if ( _st_prev )                                                           \
            _st_acc[k] += _n - _st_prev;                                           \
         if ( _st_prev )                                                           \
            _st_prev = _n;                                                         \
      }                                                                            \
   } while ( 0 )
#define SOR_STEP_DECL                                                    \
   unsigned long long _st_prev = 0, _st_acc[4] = { 0, 0, 0, 0 };         \
   const unsigned long long _st_t0 = __builtin_readcyclecounter(), _st_r0 = __builtin_amdgcn_s_memrealtime();
#define SOR_STEP_FLUSH                                                 \
   if ( threadIdx.x == 0 && blk.P == 8 && blk.Q == 8 && blk.R == 4 )  \
   {                                                                   \
      for ( int _k = 0; _k < 4; ++_k )                                 \
         g_step_acc[_k] = _st_acc[_k];                                 \
      g_step_acc[4] = __builtin_readcyclecounter() - _st_t0;           \
      g_step_acc[5] = __builtin_amdgcn_s_memrealtime() - _st_r0;       \
   }
#endif

// -DSOR_LOOKAHEAD: the variant of exp/p1_sor_lookahead.hip (register-carried neighbours, look-ahead sum, branch-free
// step, blocks from the grid index) instead of the production file -- measured slower, see its header
#ifdef SOR_LOOKAHEAD
#include "p1_sor_lookahead.hip"
#else
#include "../p1_sor.hip"
#endif

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

int main( int argc, char** argv )
{
   const int    level = argc > 1 ? atoi( argv[1] ) : 8;
   const int    N     = ( 1 << level ) + 1;
   const size_t n     = (size_t) N * ( N + 1 ) * ( N + 2 ) / 6;
   std::vector< double > hu( n ), hr( n );
   for ( size_t i = 0; i < n; ++i )
      hu[i] = (double) ( ( i * 2654435761u ) % 1000 ) * 1e-3, hr[i] = (double) ( ( i * 40503u ) % 1000 ) * 1e-3;
   double *u, *r;
   CK( hipMalloc( &u, n * 8 ) );
   CK( hipMalloc( &r, n * 8 ) );
   CK( hipMemcpy( u, hu.data(), n * 8, hipMemcpyHostToDevice ) );
   CK( hipMemcpy( r, hr.data(), n * 8, hipMemcpyHostToDevice ) );
   const size_t        nslots = (size_t) 32 * 32 * 32 * 8;
   unsigned long long* tr;
   CK( hipMalloc( &tr, nslots * 8 ) );
   CK( hipMemset( tr, 0, nslots * 8 ) );
   CK( hipMemcpyToSymbol( HIP_SYMBOL( g_sor_trace ), &tr, sizeof( tr ) ) );
   double w[15] = { -0.1, -0.05, -0.02, -0.07, -0.03, -0.04, -0.11, 1.0, -0.11, -0.04, -0.03, -0.07, -0.02, -0.05, -0.1 };
   hipStream_t s;
   CK( hipStreamCreate( &s ) );
   hipEvent_t e0, e1;
   CK( hipEventCreate( &e0 ) );
   CK( hipEventCreate( &e1 ) );
   for ( int bw = 0; bw < 2; ++bw )
   {
      for ( int k = 0; k < 3; ++k )
         if ( hyteg_hip_p1_sor_cell( u, r, level, w, 1.0, bw, s ) != 0 )
            return 1;
      CK( hipStreamSynchronize( s ) );
      const int reps = 10;
      CK( hipEventRecord( e0, s ) );
      for ( int k = 0; k < reps; ++k )
         hyteg_hip_p1_sor_cell( u, r, level, w, 1.0, bw, s );
      CK( hipEventRecord( e1, s ) );
      CK( hipStreamSynchronize( s ) );
      float ms;
      CK( hipEventElapsedTime( &ms, e0, e1 ) );
      printf( "level %d %s sweep: %.1f us\n", level, bw ? "backward" : "forward", ms * 1e3 / reps );
#ifdef SOR_STEPS
      {
         unsigned long long acc[8];
         CK( hipMemcpyFromSymbol( acc, HIP_SYMBOL( g_step_acc ), sizeof( acc ) ) );
         printf( "  block (8,8,4), wave 0, shader cycles per step: reads issued -> all arrived %.0f | arithmetic + write + register moves -> write done %.0f | "
                 "loop overhead %.0f | barrier %.0f\n",
                 acc[0] / 46.0, acc[1] / 46.0, acc[2] / 46.0, acc[3] / 46.0 );
         printf( "  the 46 + 3 steps: %llu shader cycles in %llu x 10 ns -> counter at %.0f MHz\n", acc[4], acc[5], acc[4] / ( acc[5] * 0.01 ) );
      }
#endif
      std::vector< unsigned long long > h( nslots );
      CK( hipMemcpy( h.data(), tr, nslots * 8, hipMemcpyDeviceToHost ) );
      // per block wavefront: first entry, medians of the phases, last end
      const int nb = ( ( 1 << level ) + 15 ) / 16;
      printf( "  wavefront blocks | start->end of the launch (us) | medians per block: rowbase, loads, to LDS, steps, write-back (us)\n" );
      unsigned long long prevEnd = 0;
      for ( int T = 0; T < 3 * nb; ++T )
      {
         std::vector< double > ph[5];
         unsigned long long     first = ~0ull, last = 0;
         int                    cnt   = 0;
         for ( int R = 0; R < nb; ++R )
            for ( int Q = R; Q < nb; ++Q )
            {
               const int P = T - Q - R;
               if ( P < Q || P >= nb )
                  continue;
               const unsigned long long* t = &h[( ( (size_t) P * 32 + Q ) * 32 + R ) * 8];
               if ( t[0] == 0 )
                  continue;
               ++cnt;
               first = std::min( first, t[0] ), last = std::max( last, t[5] );
               for ( int k = 0; k < 5; ++k )
                  ph[k].push_back( ( t[k + 1] - t[k] ) * 0.01 );
            }
         if ( !cnt )
            continue;
         for ( auto& v : ph )
            std::sort( v.begin(), v.end() );
         const int wv = bw ? 3 * nb - T : T;
         if ( T % 4 == 0 || cnt > 100 )
            printf( "  T %3d  %4d blocks | %6.2f (gap to previous %6.2f) | %5.2f %5.2f %5.2f %6.2f %5.2f\n", T, cnt, ( last - first ) * 0.01,
                    prevEnd ? ( (double) first - (double) prevEnd ) * 0.01 : 0.0, ph[0][cnt / 2], ph[1][cnt / 2], ph[2][cnt / 2],
                    ph[3][cnt / 2], ph[4][cnt / 2] );
         (void) wv;
         prevEnd = last;
      }
   }
   return 0;
}
