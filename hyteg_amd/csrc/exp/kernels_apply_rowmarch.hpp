// Row-marching register kernel for the 15-point constant-stencil apply / fused Jacobi on one macro-cell.
//
// One WAVE owns a strip of one z-slice: 64 consecutive x positions (lanes 1..62 produce outputs, lanes 0
// and 63 are halo lanes) times `ny` consecutive rows.  It marches in +y keeping a register window of the
// seven source rows an output row needs,
//     slice z  : rows y-1, y, y+1        slice z+1 : rows y-1, y        slice z-1 : rows y, y+1
// so that advancing one row costs three new coalesced 8-byte loads (the new y+1 rows), the x-1 / x+1
// neighbours come from wave-wide DPP shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1), and there is no LDS,
// no barrier and no per-point index decode.  Loads for the next row are issued before the current row is
// evaluated: a ring of D prefetched row triples keeps D rows of loads in flight per wave, because a wave
// that waits one full memory round trip per row is latency-bound (measured: D=1 runs at 2x the copy time).
//
// Index algebra (W = N-z, R = W-y = length of row y in slice z):
//   row y-1 -> y   : +R+1        row y -> y+1 : +R           (slice z)
//   up   row y-1 -> y : +R       down row y -> y+1 : +R+1
#pragma once

#include <algorithm>
#include <vector>

#include "../common.hpp"

namespace hyteg_hip {

struct RowTask
{
   int z;  // slice
   int y0; // first row
   int ny; // number of rows
   int x0; // x of lane 1 (lane l holds x0 - 1 + l)
};
static_assert( sizeof( RowTask ) == 16, "RowTask must be 16 bytes" );

struct RowMarchArgs
{
   double*        dst;
   const double*  src;
   const double*  rhs;     // JACOBI only
   const double*  invdiag; // JACOBI only, may be null
   const RowTask* tasks;
   int            ntasks;
   int            N;
   int            total;
   int            xcd_chunk; // workgroups per XCD group (0: identity map)
   double         relax;
   Stencil15      st;
};

constexpr int kRowMarchWavesPerBlock = 4;

__device__ inline double lane_minus_1( double v ) // value held by lane-1 (x-1)
{
   int lo = __double2loint( v ), hi = __double2hiint( v );
   lo     = __builtin_amdgcn_mov_dpp( lo, 0x138, 0xf, 0xf, true ); // wave_shr:1
   hi     = __builtin_amdgcn_mov_dpp( hi, 0x138, 0xf, 0xf, true );
   return __hiloint2double( hi, lo );
}
__device__ inline double lane_plus_1( double v ) // value held by lane+1 (x+1)
{
   int lo = __double2loint( v ), hi = __double2hiint( v );
   lo     = __builtin_amdgcn_mov_dpp( lo, 0x130, 0xf, 0xf, true ); // wave_shl:1
   hi     = __builtin_amdgcn_mov_dpp( hi, 0x130, 0xf, 0xf, true );
   return __hiloint2double( hi, lo );
}

template < int MODE, int D >
__global__ __launch_bounds__( 64 * kRowMarchWavesPerBlock ) void p1_apply_rowmarch_kernel( const RowMarchArgs A )
{
   int b = blockIdx.x;
   if ( A.xcd_chunk > 0 )
      b = ( blockIdx.x & 7 ) * A.xcd_chunk + ( blockIdx.x >> 3 );
   const int task = __builtin_amdgcn_readfirstlane( b * kRowMarchWavesPerBlock + ( threadIdx.x >> 6 ) );
   if ( task >= A.ntasks )
      return;
   const RowTask t    = A.tasks[task];
   const int     lane = threadIdx.x & 63;
   const int     N    = A.N;
   const int     W    = N - t.z;
   const int     S0   = tri( W );
   const int     Sm   = tri( W + 1 );
   const int     s0   = slice_start( N, t.z );
   const int     x    = t.x0 - 1 + lane;
   const int     top  = A.total - 1;
   const int     y    = t.y0;
   int           R    = W - y; // length of the current output row

   const int im  = s0 + row_start( W, y - 1 ) + x;          // (x, y-1, z)
   int       i0  = im + R + 1;                              // (x, y,   z)
   int       ip  = i0 + R;                                  // (x, y+1, z)
   const int ium = s0 + S0 + row_start( W - 1, y - 1 ) + x; // (x, y-1, z+1)
   int       iu0 = ium + R;                                 // (x, y,   z+1)
   const int id0 = s0 - Sm + row_start( W + 1, y ) + x;     // (x, y,   z-1)
   int       idp = id0 + R + 1;                             // (x, y+1, z-1)

   const double* __restrict__ src = A.src;
#define LD( idx ) src[min( max( ( idx ), 0 ), top )]
   double a_m = LD( im ), a_0 = LD( i0 ), a_p = LD( ip );
   double u_m = LD( ium ), u_0 = LD( iu0 );
   double d_0 = LD( id0 ), d_p = LD( idp );

   // ring of D prefetched row triples: slot j holds the rows that enter the window after output row k*D + j
   double pa[D], pu[D], pd[D];
   int    Rn = R; // length of the output row whose successor triple is loaded next
#pragma unroll
   for ( int j = 0; j < D; ++j )
   {
      ip += Rn - 1;
      iu0 += Rn - 1;
      idp += Rn;
      pa[j] = LD( ip );
      pu[j] = LD( iu0 );
      pd[j] = LD( idp );
      Rn -= 1;
   }

   const double* w       = A.st.w;
   const double  invc    = 1.0 / w[7];
   const bool    lane_ok = lane >= 1 && lane <= 62;

   // straight-line body of D rows (no control flow inside, so the compiler's vmcnt accounting stays exact and
   // the D row triples really are in flight together); rows beyond ny are computed but not stored
   for ( int k0 = 0; k0 < t.ny; k0 += D )
   {
#pragma unroll
      for ( int j = 0; j < D; ++j )
      {
         {
            const bool active = lane_ok && x <= R - 2 && k0 + j < t.ny;
            double     rhs_v = 0.0, invd = invc;
            if ( MODE == APPLY_JACOBI )
            {
               const int ic = min( i0, top );
               rhs_v        = A.rhs[ic];
               if ( A.invdiag )
                  invd = A.invdiag[ic];
            }

            double acc;
            acc = w[6] * lane_minus_1( a_0 );             // W
            acc = fma( w[3], d_p, acc );                  // BN
            acc = fma( w[10], a_p, acc );                 // N
            acc = fma( w[5], lane_plus_1( a_m ), acc );   // SE
            acc = fma( w[12], lane_plus_1( u_m ), acc );  // TSE
            acc = fma( w[1], lane_plus_1( d_0 ), acc );   // BE
            acc = fma( w[8], lane_plus_1( a_0 ), acc );   // E
            acc = fma( w[13], lane_minus_1( u_0 ), acc ); // TW
            acc = fma( w[2], lane_minus_1( d_p ), acc );  // BNW
            acc = fma( w[9], lane_minus_1( a_p ), acc );  // NW
            acc = fma( w[4], a_m, acc );                  // S
            acc = fma( w[11], u_m, acc );                 // TS
            acc = fma( w[0], d_0, acc );                  // BC
            acc = fma( w[7], a_0, acc );                  // C
            acc = fma( w[14], u_0, acc );                 // TC

            if ( active )
            {
               if ( MODE == APPLY_REPLACE )
                  A.dst[i0] = acc;
               else if ( MODE == APPLY_ADD )
                  A.dst[i0] = acc + A.dst[i0];
               else
                  A.dst[i0] = a_0 + A.relax * ( invd * ( rhs_v - acc ) );
            }

            // advance the window by one row, refill the ring slot D rows ahead
            a_m = a_0;
            a_0 = a_p;
            a_p = pa[j];
            u_m = u_0;
            u_0 = pu[j];
            d_0 = d_p;
            d_p = pd[j];
            i0 += R;
            R -= 1;
            ip += Rn - 1;
            iu0 += Rn - 1;
            idp += Rn;
            Rn -= 1;
            pa[j] = LD( ip );
            pu[j] = LD( iu0 );
            pd[j] = LD( idp );
         }
      }
   }
#undef LD
}

// Fully unrolled variant: every task has exactly NY rows (rows past the interior are masked).  All
// 3*NY+4 row segments the strip needs are loaded up front into registers with compile-time indices (one
// memory round trip per task, maximal memory-level parallelism, exact s_waitcnt accounting because there is
// no loop), then the NY output rows are evaluated from registers.
// ABL (developer ablation switches, 0 in production): 1 = no up/down loads, 2 = unmasked stores (wrong
// results, same bytes), 4 = no stores, 8 = no stencil arithmetic, 16 = 16-byte-aligned row segments (wrong data)
template < int MODE, int NY, int ABL = 0 >
__global__ __launch_bounds__( 64 * kRowMarchWavesPerBlock ) void p1_apply_rowstrip_kernel( const RowMarchArgs A )
{
   int b = blockIdx.x;
   if ( A.xcd_chunk > 0 )
      b = ( blockIdx.x & 7 ) * A.xcd_chunk + ( blockIdx.x >> 3 );
   const int task = __builtin_amdgcn_readfirstlane( b * kRowMarchWavesPerBlock + ( threadIdx.x >> 6 ) );
   if ( task >= A.ntasks )
      return;
   const RowTask t    = A.tasks[task];
   const int     lane = threadIdx.x & 63;
   const int     N    = A.N;
   const int     W    = N - t.z;
   const int     S0   = tri( W );
   const int     Sm   = tri( W + 1 );
   const int     s0   = slice_start( N, t.z );
   const int     x    = t.x0 - 1 + lane;
   const int     top  = A.total - 1;
   const int     y    = t.y0;
   const int     R0   = W - y; // length of the first output row

   const double* __restrict__ src = A.src;
#define LD( idx ) src[min( max( ( ABL & 16 ) ? ( ( idx ) & ~15 ) + lane : ( idx ), 0 ), top )]
   double a[NY + 2], u[NY + 1], d[NY + 1];
   {
      int ia = s0 + row_start( W, y - 1 ) + x;          // (x, y-1, z); next row: + (R0+1), then R0, R0-1, ...
      int iu = s0 + S0 + row_start( W - 1, y - 1 ) + x; // (x, y-1, z+1); next row: + R0, R0-1, ...
      int id = s0 - Sm + row_start( W + 1, y ) + x;     // (x, y, z-1); next row: + (R0+1), R0, ...
#pragma unroll
      for ( int j = 0; j < NY + 2; ++j )
      {
         a[j] = LD( ia );
         ia += R0 + 1 - j;
         if ( j < NY + 1 )
         {
            if ( ABL & 1 )
            {
               u[j] = a[j];
               d[j] = a[j];
            }
            else
            {
               u[j] = LD( iu );
               d[j] = LD( id );
            }
            iu += R0 - j;
            id += R0 + 1 - j;
         }
      }
   }
#undef LD

   const double* w       = A.st.w;
   const double  invc    = 1.0 / w[7];
   const bool    lane_ok = lane >= 1 && lane <= 62;
   int           i0      = s0 + row_start( W, y ) + x;

#pragma unroll
   for ( int j = 0; j < NY; ++j )
   {
      const int  R      = R0 - j;
      const bool active = ( ABL & 2 ) ? true : ( lane_ok && x <= R - 2 && j < t.ny );
      double     acc;
      if ( ABL & 8 )
      {
         acc = a[j] + a[j + 1] + a[j + 2] + u[j] + u[j + 1] + d[j] + d[j + 1];
      }
      else
      {
      acc = w[6] * lane_minus_1( a[j + 1] );             // W
      acc = fma( w[3], d[j + 1], acc );                  // BN
      acc = fma( w[10], a[j + 2], acc );                 // N
      acc = fma( w[5], lane_plus_1( a[j] ), acc );       // SE
      acc = fma( w[12], lane_plus_1( u[j] ), acc );      // TSE
      acc = fma( w[1], lane_plus_1( d[j] ), acc );       // BE
      acc = fma( w[8], lane_plus_1( a[j + 1] ), acc );   // E
      acc = fma( w[13], lane_minus_1( u[j + 1] ), acc ); // TW
      acc = fma( w[2], lane_minus_1( d[j + 1] ), acc );  // BNW
      acc = fma( w[9], lane_minus_1( a[j + 2] ), acc );  // NW
      acc = fma( w[4], a[j], acc );                      // S
      acc = fma( w[11], u[j], acc );                     // TS
      acc = fma( w[0], d[j], acc );                      // BC
      acc = fma( w[7], a[j + 1], acc );                  // C
      acc = fma( w[14], u[j + 1], acc );                 // TC
      }
      if ( ABL & 4 )
      {
         if ( acc == 1.2345e-300 )
            A.dst[i0] = acc;
      }
      else if ( active )
      {
         if ( MODE == APPLY_REPLACE )
            A.dst[( ABL & 16 ) ? min( ( i0 & ~15 ) + lane, top ) : i0] = acc;
         else if ( MODE == APPLY_ADD )
            A.dst[i0] = acc + A.dst[i0];
         else
         {
            const double invd = A.invdiag ? A.invdiag[i0] : invc;
            A.dst[i0]         = a[j + 1] + A.relax * ( invd * ( A.rhs[i0] - acc ) );
         }
      }
      i0 += R;
   }
}

// host: build the task list for (level, rows per task).  Order: z, then y-chunk, then x-chunk (memory order).
inline void build_row_tasks( int level, int rows_per_task, std::vector< RowTask >& out )
{
   const int N = ( 1 << level ) + 1;
   out.clear();
   for ( int z = 1; z <= N - 4; ++z )
   {
      const int W = N - z;
      for ( int y0 = 1; y0 <= W - 3; y0 += rows_per_task )
      {
         const int ny   = std::min( rows_per_task, W - 3 - y0 + 1 );
         const int xmax = W - y0 - 2; // last interior x of the first (longest) row
         for ( int x0 = 1; x0 <= xmax; x0 += 62 )
            out.push_back( RowTask{ z, y0, ny, x0 } );
      }
   }
}

} // namespace hyteg_hip
