// "strip2" kernel for the 15-point constant-stencil apply / fused Jacobi on one macro-cell.
//
// Why this shape (measured on MI355X, level 8): the apply moves only 46 MB, a plain copy of
// that size takes ~10 us, and every earlier variant (direct loads, LDS tiles with per-point index decode,
// one-double-per-lane row marching) sat at 17-25 us because it was bound by VALU/SALU instruction issue, not
// by HBM.  This kernel minimises instructions per output:
//   * one WAVE owns a strip of one z-slice: NY consecutive rows x 128 consecutive x positions, TWO doubles
//     per lane (lane l holds x = xb+2l and xb+2l+1), lanes 1..62 produce 124 outputs per row;
//   * all 3*NY+4 source row segments the strip needs (slice z rows y0-1..y0+NY, slice z+1 rows y0-1..y0+NY-1,
//     slice z-1 rows y0..y0+NY) are fetched up front with 16-byte buffer loads (one memory round trip per
//     task, addresses = wave-uniform row base + lane*16, i.e. one VALU add per load; the buffer descriptor's
//     range check returns 0 beyond the array, so there is no clamping and no exec masking);
//   * x-1 / x+1 neighbours: within a lane for half of the cases, otherwise one wave-wide DPP shift
//     (v_mov_b32_dpp wave_shr:1 / wave_shl:1) of the neighbouring lane's odd / even element;
//   * stores are 16-byte buffer stores whose offset is forced out of range for inactive lanes.
// No LDS, no barrier, no per-point index decode, no loop (fully unrolled over NY).
//
// Index algebra (W = N-z, R = W-y = length of row y in slice z), element index of a fixed x:
//   slice z   : row y-1 -> y : +R+1,   y -> y+1 : +R
//   slice z+1 : row y-1 -> y : +R,     slice z-1 : row y -> y+1 : +R+1
#pragma once

#include <algorithm>
#include <vector>

#include "../common.hpp"

namespace hyteg_hip {

struct StripTask
{
   int ia;  // element index of (xb, y0-1, z)
   int iu;  // element index of (xb, y0-1, z+1)
   int id;  // element index of (xb, y0,   z-1)
   int io;  // element index of (xb, y0,   z)   (outputs)
   int R0;  // length of row y0 in slice z
   int ny;  // rows of this strip that exist (<= NY)
   int xb;  // x held in lane 0's first element (= x0 - 2, x0 = first output x)
   int pad;
};
static_assert( sizeof( StripTask ) == 32, "StripTask must be 32 bytes" );

struct Strip2Args
{
   double*          dst;
   const double*    src;
   const double*    rhs;     // JACOBI only
   const double*    invdiag; // JACOBI only, may be null
   const StripTask* tasks;
   int              ntasks;
   unsigned         bytes;     // size of the cell array in bytes (buffer range)
   int              xcd_chunk; // workgroups per XCD group (0: identity map)
   int              pad;
   double           relax;
   Stencil15        st;
};

constexpr int kStripWavesPerBlock = 4;

typedef int v4i_t __attribute__( ( ext_vector_type( 4 ) ) );
typedef int v2i_t __attribute__( ( ext_vector_type( 2 ) ) );

__device__ inline double2 buf_load2( __amdgpu_buffer_rsrc_t r, int byte_off )
{
   v4i_t v = __builtin_amdgcn_raw_buffer_load_b128( r, byte_off, 0, 0 );
   return *reinterpret_cast< double2* >( &v );
}
__device__ inline void buf_store2( __amdgpu_buffer_rsrc_t r, int byte_off, double2 d )
{
   __builtin_amdgcn_raw_buffer_store_b128( *reinterpret_cast< v4i_t* >( &d ), r, byte_off, 0, 0 );
}
__device__ inline void buf_store1( __amdgpu_buffer_rsrc_t r, int byte_off, double d )
{
   __builtin_amdgcn_raw_buffer_store_b64( *reinterpret_cast< v2i_t* >( &d ), r, byte_off, 0, 0 );
}

__device__ inline double dpp_lane_minus_1( double v ) // value held by lane-1
{
   int lo = __double2loint( v ), hi = __double2hiint( v );
   lo     = __builtin_amdgcn_mov_dpp( lo, 0x138, 0xf, 0xf, true ); // wave_shr:1
   hi     = __builtin_amdgcn_mov_dpp( hi, 0x138, 0xf, 0xf, true );
   return __hiloint2double( hi, lo );
}
__device__ inline double dpp_lane_plus_1( double v ) // value held by lane+1
{
   int lo = __double2loint( v ), hi = __double2hiint( v );
   lo     = __builtin_amdgcn_mov_dpp( lo, 0x130, 0xf, 0xf, true ); // wave_shl:1
   hi     = __builtin_amdgcn_mov_dpp( hi, 0x130, 0xf, 0xf, true );
   return __hiloint2double( hi, lo );
}

template < int MODE, int NY >
__global__ __launch_bounds__( 64 * kStripWavesPerBlock ) void p1_apply_strip2_kernel( const Strip2Args A )
{
   int b = blockIdx.x;
   if ( A.xcd_chunk > 0 )
      b = ( blockIdx.x & 7 ) * A.xcd_chunk + ( blockIdx.x >> 3 );
   const int task = __builtin_amdgcn_readfirstlane( b * kStripWavesPerBlock + ( threadIdx.x >> 6 ) );
   if ( task >= A.ntasks )
      return;
   const StripTask t    = A.tasks[task];
   const int       lane = threadIdx.x & 63;

   const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc( const_cast< double* >( A.src ), 0, A.bytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc( A.dst, 0, A.bytes, 0x00020000 );

   const int lane_off = lane * 16;
   const int R0       = t.R0;

   // ---- all loads up front ----
   double2 a[NY + 2], u[NY + 1], d[NY + 1];
   {
      int ia = t.ia * 8, iu = t.iu * 8, id = t.id * 8; // wave-uniform byte offsets of the row segments
#pragma unroll
      for ( int j = 0; j < NY + 2; ++j )
      {
         a[j] = buf_load2( rs, ia + lane_off );
         ia += ( R0 + 1 - j ) * 8;
         if ( j < NY + 1 )
         {
            u[j] = buf_load2( rs, iu + lane_off );
            iu += ( R0 - j ) * 8;
            d[j] = buf_load2( rs, id + lane_off );
            id += ( R0 + 1 - j ) * 8;
         }
      }
   }

   const double* w       = A.st.w;
   const double  invc    = 1.0 / w[7];
   const bool    lane_ok = lane >= 1 && lane <= 62;
   const int     x       = t.xb + 2 * lane; // x of this lane's first element
   int           io      = t.io * 8;        // wave-uniform byte offset of the output row

   // wave-wide shifts, each computed once: L*[j] = element x-1 of row j (lane-1's odd element),
   // R*[j] = element x+2 of row j (lane+1's even element)
   double La[NY + 2], Ra[NY + 2], Lu[NY + 1], Ru[NY + 1], Ld[NY + 1], Rd[NY + 1];
#pragma unroll
   for ( int j = 0; j < NY + 2; ++j )
   {
      if ( j >= 1 )
         La[j] = dpp_lane_minus_1( a[j].y );
      if ( j <= NY )
         Ra[j] = dpp_lane_plus_1( a[j].x );
      if ( j >= 1 && j <= NY )
      {
         Lu[j] = dpp_lane_minus_1( u[j].y );
         Ld[j] = dpp_lane_minus_1( d[j].y );
      }
      if ( j < NY )
      {
         Ru[j] = dpp_lane_plus_1( u[j].x );
         Rd[j] = dpp_lane_plus_1( d[j].x );
      }
   }

#pragma unroll
   for ( int j = 0; j < NY; ++j )
   {
      const int     R  = R0 - j;
      const double2 am = a[j], a0 = a[j + 1], ap = a[j + 2];
      const double2 um = u[j], u0 = u[j + 1];
      const double2 d0 = d[j], dp = d[j + 1];
      const double  La0 = La[j + 1], Lap = La[j + 2], Lu0 = Lu[j + 1], Ldp = Ld[j + 1];
      const double  Ra0 = Ra[j + 1], Ram = Ra[j], Rum = Ru[j], Rd0 = Rd[j];
      double e0, e1;
      // first output (x): neighbours x-1 -> L*, x -> .x, x+1 -> .y
      e0 = w[6] * La0;           // W
      e0 = fma( w[3], dp.x, e0 );  // BN
      e0 = fma( w[10], ap.x, e0 ); // N
      e0 = fma( w[5], am.y, e0 );  // SE
      e0 = fma( w[12], um.y, e0 ); // TSE
      e0 = fma( w[1], d0.y, e0 );  // BE
      e0 = fma( w[8], a0.y, e0 );  // E
      e0 = fma( w[13], Lu0, e0 );  // TW
      e0 = fma( w[2], Ldp, e0 );   // BNW
      e0 = fma( w[9], Lap, e0 );   // NW
      e0 = fma( w[4], am.x, e0 );  // S
      e0 = fma( w[11], um.x, e0 ); // TS
      e0 = fma( w[0], d0.x, e0 );  // BC
      e0 = fma( w[7], a0.x, e0 );  // C
      e0 = fma( w[14], u0.x, e0 ); // TC
      // second output (x+1): neighbours x -> .x, x+1 -> .y, x+2 -> R*
      e1 = w[6] * a0.x;
      e1 = fma( w[3], dp.y, e1 );
      e1 = fma( w[10], ap.y, e1 );
      e1 = fma( w[5], Ram, e1 );
      e1 = fma( w[12], Rum, e1 );
      e1 = fma( w[1], Rd0, e1 );
      e1 = fma( w[8], Ra0, e1 );
      e1 = fma( w[13], u0.x, e1 );
      e1 = fma( w[2], dp.x, e1 );
      e1 = fma( w[9], ap.x, e1 );
      e1 = fma( w[4], am.y, e1 );
      e1 = fma( w[11], um.y, e1 );
      e1 = fma( w[0], d0.y, e1 );
      e1 = fma( w[7], a0.y, e1 );
      e1 = fma( w[14], u0.y, e1 );

      const bool row_ok = lane_ok && j < t.ny;
      const bool act0   = row_ok && x <= R - 2;     // first output is an interior point
      const bool act1   = row_ok && x + 1 <= R - 2; // second output is an interior point
      const int  off    = io + lane_off;
      double2    out;
      if ( MODE == APPLY_REPLACE )
      {
         out.x = e0;
         out.y = e1;
      }
      else if ( MODE == APPLY_ADD )
      {
         const double2 old = buf_load2( rd, off );
         out.x             = e0 + old.x;
         out.y             = e1 + old.y;
      }
      else
      {
         const __amdgpu_buffer_rsrc_t rr =
             __builtin_amdgcn_make_buffer_rsrc( const_cast< double* >( A.rhs ), 0, A.bytes, 0x00020000 );
         const double2 rv = buf_load2( rr, off );
         double2       iv;
         iv.x = invc;
         iv.y = invc;
         if ( A.invdiag )
         {
            const __amdgpu_buffer_rsrc_t ri =
                __builtin_amdgcn_make_buffer_rsrc( const_cast< double* >( A.invdiag ), 0, A.bytes, 0x00020000 );
            iv = buf_load2( ri, off );
         }
         out.x = a0.x + A.relax * ( iv.x * ( rv.x - e0 ) );
         out.y = a0.y + A.relax * ( iv.y * ( rv.y - e1 ) );
      }
      // both active: one 16-byte store; only the first active (row end): one 8-byte store; offset -1 = dropped
      buf_store2( rd, act1 ? off : -16, out );
      buf_store1( rd, ( act0 && !act1 ) ? off : -16, out.x );
      io += R * 8;
   }
}

// host: strips of NY rows x 124 outputs, in memory order (z, y-chunk, x-chunk)
inline void build_strip_tasks( int level, int NY, std::vector< StripTask >& out )
{
   const int N = ( 1 << level ) + 1;
   out.clear();
   for ( int z = 1; z <= N - 4; ++z )
   {
      const int W  = N - z;
      const int s0 = slice_start( N, z );
      const int S0 = tri( W ), Sm = tri( W + 1 );
      for ( int y0 = 1; y0 <= W - 3; y0 += NY )
      {
         const int ny   = std::min( NY, W - 3 - y0 + 1 );
         const int R0   = W - y0;
         const int xmax = R0 - 2; // last interior x of the first (longest) row
         for ( int x0 = 1; x0 <= xmax; x0 += 124 )
         {
            StripTask t{};
            t.xb = x0 - 2;
            t.ia = s0 + row_start( W, y0 - 1 ) + t.xb;
            t.iu = s0 + S0 + row_start( W - 1, y0 - 1 ) + t.xb;
            t.id = s0 - Sm + row_start( W + 1, y0 ) + t.xb;
            t.io = s0 + row_start( W, y0 ) + t.xb;
            t.R0 = R0;
            t.ny = ny;
            out.push_back( t );
         }
      }
   }
}

} // namespace hyteg_hip
