// "z-march" kernel for the 15-point constant-stencil apply / fused Jacobi on one macro-cell.
//
// Measured facts that shape it (MI355X, level 8; DESIGN.md has the numbers):
//   * the apply moves only 46 MB; a plain copy of that size takes ~10 us, of which ~1.7 us is launch/ramp;
//   * arithmetic, index decode, DPP shifts are NOT the bound (removing all arithmetic changes nothing);
//   * what costs time is (a) re-reading every source row ~4x through L1/L2 and (b) issuing all loads first
//     and all stores last, so that HBM reads and the write-back of dst never overlap.
// So: one WAVE owns a brick of NY rows x 64 x-positions (lanes 1..62 produce outputs) x LZ slices and marches
// in +z.  Slice z+1's rows are loaded ONCE and serve as "up" rows for slice z, centre rows for z+1 and "down"
// rows for z+2 straight from registers ((LZ(NY+2)+2(NY+1))/(LZ NY) ~ 1.8 row loads per output row instead of
// 4), and the loads of slice z+2 are issued before slice z is evaluated and stored, so loads and stores are
// in flight together in every wave.  Everything is fully unrolled (compile-time NY, LZ): no loop-carried
// register copies, exact s_waitcnt accounting.  x-1/x+1 neighbours come from wave-wide DPP shifts; addresses
// are wave-uniform row bases + lane*8 through a buffer descriptor whose range check returns 0 past the array
// end and drops stores whose offset is forced out of range (no exec masking, no clamping).
//
// Index algebra: W = N-z; element (x,y,z) -> (x,y,z+1): + tri(W) - y;  (x,y,z) -> (x,y+1,z): + (W-y).
#pragma once

#include <algorithm>
#include <vector>

#include "../common.hpp"

namespace hyteg_hip {

struct BrickTask
{
   int i0;  // element index of (xb, y0-1, z0-1): first row segment of the first slice
   int W0;  // N - (z0-1): row-0 length of slice z0-1
   int y0;  // first output row
   int xb;  // x held by lane 0 (= x0 - 1, x0 = first output x)
   int nz;  // slices of this brick that exist (<= LZ)
   int pad[3];
};
static_assert( sizeof( BrickTask ) == 32, "BrickTask must be 32 bytes" );

struct ZMarchArgs
{
   double*          dst;
   const double*    src;
   const double*    rhs;     // JACOBI only
   const double*    invdiag; // JACOBI only, may be null
   const BrickTask* tasks;
   int              ntasks;
   unsigned         bytes;     // size of the cell array in bytes (buffer range)
   int              xcd_chunk; // workgroups per XCD group (0: identity map)
   int              pad;
   double           relax;
   Stencil15        st;
};

#ifndef HYTEG_ZM_WAVES_PER_BLOCK
#define HYTEG_ZM_WAVES_PER_BLOCK 4
#endif
constexpr int kZMarchWavesPerBlock = HYTEG_ZM_WAVES_PER_BLOCK; // 1, 2, 8 measured within noise of 4 at level 8

typedef int zm_v2i_t __attribute__( ( ext_vector_type( 2 ) ) );

template < int AUX = 0 >
__device__ inline double zm_load( __amdgpu_buffer_rsrc_t r, int byte_off )
{
   zm_v2i_t v = __builtin_amdgcn_raw_buffer_load_b64( r, byte_off, 0, AUX );
   return *reinterpret_cast< double* >( &v );
}
// wave-uniform row base in the scalar offset, lane part in the vector offset (the range check sees the vector offset only)
template < int AUX = 0 >
__device__ inline double zm_load2( __amdgpu_buffer_rsrc_t r, int voff, int soff )
{
   zm_v2i_t v = __builtin_amdgcn_raw_buffer_load_b64( r, voff, soff, AUX );
   return *reinterpret_cast< double* >( &v );
}
template < int AUX = 0 >
__device__ inline void zm_store2( __amdgpu_buffer_rsrc_t r, int voff, int soff, double d )
{
   __builtin_amdgcn_raw_buffer_store_b64( *reinterpret_cast< zm_v2i_t* >( &d ), r, voff, soff, AUX );
}
template < int AUX = 0 >
__device__ inline void zm_store( __amdgpu_buffer_rsrc_t r, int byte_off, double d )
{
   __builtin_amdgcn_raw_buffer_store_b64( *reinterpret_cast< zm_v2i_t* >( &d ), r, byte_off, 0, AUX );
}
__device__ inline double zm_lane_minus_1( double v )
{
   int lo = __double2loint( v ), hi = __double2hiint( v );
   lo     = __builtin_amdgcn_mov_dpp( lo, 0x138, 0xf, 0xf, true ); // wave_shr:1
   hi     = __builtin_amdgcn_mov_dpp( hi, 0x138, 0xf, 0xf, true );
   return __hiloint2double( hi, lo );
}
__device__ inline double zm_lane_plus_1( double v )
{
   int lo = __double2loint( v ), hi = __double2hiint( v );
   lo     = __builtin_amdgcn_mov_dpp( lo, 0x130, 0xf, 0xf, true ); // wave_shl:1
   hi     = __builtin_amdgcn_mov_dpp( hi, 0x130, 0xf, 0xf, true );
   return __hiloint2double( hi, lo );
}

// Cache policy (gfx950 "aux" bits: 1 = sc0, 2 = nt, 16 = sc1).  dst is written once and never re-read by this
// kernel: nontemporal stores took the level-8 apply from 14.5 to 10.4 us (they do not leave 22 MB of dirty
// lines for the end-of-kernel L2 write-back).  ABL: developer ablation switches, 0 in production
// (4 = no stores, 8 = no stencil arithmetic).
constexpr int kStoreAuxDefault = 2;
// FACT: the eight x-shifted stencil terms are summed per shift direction BEFORE the lane shift (two wave shifts per
// output instead of eight; the shifts were half of the kernel's VALU time).  Changes the summation order, not the terms.
// FACT == 2 additionally evaluates the NY rows of a step side by side (16 independent FMA chains instead of one 8-deep
// chain per row): measured equal to FACT == 1 within noise at levels 6..9, kept for the harness only.
// PFALL: all source loads of the brick are issued before the first store (tests whether loads queue behind the
// nontemporal stores in the wave's in-order vmcnt): measured 4-5% SLOWER at level 8, harness only.
// SOFF: instruction diet.  The kernel issues ~45 instructions per output row and a wave issues one every 4-5 cycles, which
// at level 8 (62 wave-rows per SIMD) is ~5 us of issue time per SIMD, as much as the HBM time.  Row bases go into the
// buffer instructions' scalar offset (no per-lane address add), masked loads become one v_min (lanes beyond the row end
// re-read the row's last entry: same cache line), the store predicate one unsigned compare against a wave-uniform limit.
// MASKLD: lanes whose x lies beyond the end of the row being loaded get an out-of-range offset (the buffer range check
// returns 0 without touching the cache) instead of fetching the next row's entries: ~30% of all lanes at level 8.
// Also means the kernel never reads past the end of the source array.  Level 8: -1..2%.
template < int MODE, int NY, int LZ, int ABL = 0, int ST_AUX = kStoreAuxDefault, int LD_AUX = 0, int FACT = 1, bool PFALL = false,
           bool MASKLD = false, int EX_AUX = 0, bool SOFF = false, int PFD = 1 >
__global__ __launch_bounds__( 64 * kZMarchWavesPerBlock ) void p1_apply_zmarch_kernel( const ZMarchArgs A )
{
   int b = blockIdx.x;
   if ( A.xcd_chunk > 0 )
      b = ( blockIdx.x & 7 ) * A.xcd_chunk + ( blockIdx.x >> 3 );
   const int task = __builtin_amdgcn_readfirstlane( b * kZMarchWavesPerBlock + ( threadIdx.x >> 6 ) );
   if ( task >= A.ntasks )
      return;
   const BrickTask t    = A.tasks[task];
   const int       lane = threadIdx.x & 63;

   const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc( const_cast< double* >( A.src ), 0, A.bytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc( A.dst, 0, A.bytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rr =
       __builtin_amdgcn_make_buffer_rsrc( const_cast< double* >( MODE == APPLY_JACOBI ? A.rhs : A.src ), 0, A.bytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
       const_cast< double* >( ( MODE == APPLY_JACOBI && A.invdiag ) ? A.invdiag : A.src ), 0, A.bytes, 0x00020000 );

   const int lane_off = lane * 8;
   const int ym       = t.y0 - 1; // first row held per slice

   // S[q][r]: slice z0-1+q, row ym+r (r = 0..NY+1), x = xb + lane.  q = 0..LZ+1.
   double S[LZ + 2][NY + 2];
   // wave-uniform element index of (xb, ym, z0-1+q) and row-0 length of that slice
   int base = t.i0;
   int Wq   = t.W0;

   auto load_slice = [&]( auto qc, int base_q, int W_q ) {
      constexpr int q  = decltype( qc )::value;
      int           ix = base_q;
#pragma unroll
      for ( int r = 0; r < NY + 2; ++r )
      {
         // slice 0 is only ever a "down" slice (rows y0..y0+NY), the last one only an "up" slice (rows ym..)
         const bool need = ( q == 0 ) ? ( r >= 1 ) : ( q == LZ + 1 ? ( r <= NY ) : true );
         if ( need )
         {
            if constexpr ( SOFF )
            {
               const int last8 = ( W_q - ( ym + r ) - 1 - t.xb ) * 8; // byte offset of the row's last entry from lane 0's
               S[q][r]         = zm_load2< LD_AUX >( rs, min( lane_off, last8 ), ix * 8 ); // negative: out of range, returns 0
            }
            else if constexpr ( MASKLD )
            {
               const int last = W_q - ( ym + r ) - 1 - t.xb; // lane holding the last entry of this row
               S[q][r]        = zm_load< LD_AUX >( rs, lane <= last ? ix * 8 + lane_off : -8 );
            }
            else
               S[q][r] = zm_load< LD_AUX >( rs, ix * 8 + lane_off );
         }
         ix += W_q - ( ym + r ); // next row of the same slice
      }
   };

   // prologue: slices 0, 1, 2
   int baseq[LZ + 2], Wqs[LZ + 2];
#pragma unroll
   for ( int q = 0; q < LZ + 2; ++q )
   {
      baseq[q] = base;
      Wqs[q]   = Wq;
      base += tri( Wq ) - ym; // (x, ym, z) -> (x, ym, z+1)
      Wq -= 1;
   }
   load_slice( std::integral_constant< int, 0 >{}, baseq[0], Wqs[0] );
   load_slice( std::integral_constant< int, 1 >{}, baseq[1], Wqs[1] );
   load_slice( std::integral_constant< int, 2 >{}, baseq[2], Wqs[2] );
   // PFD: how many slices ahead of the one being computed the loads run (1: the next slice is loaded while this one is
   // computed).  All LZ+2 slices have their own registers, so a longer distance costs no registers, only earlier issue.
   if constexpr ( !PFALL && PFD >= 2 && LZ + 1 >= 3 )
      load_slice( std::integral_constant< int, 3 >{}, baseq[3 <= LZ + 1 ? 3 : 0], Wqs[3 <= LZ + 1 ? 3 : 0] );
   if constexpr ( !PFALL && PFD >= 3 && LZ + 1 >= 4 )
      load_slice( std::integral_constant< int, 4 >{}, baseq[4 <= LZ + 1 ? 4 : 0], Wqs[4 <= LZ + 1 ? 4 : 0] );
   if constexpr ( PFALL )
   {
      [&]< int... Is >( std::integer_sequence< int, Is... > ) {
         ( load_slice( std::integral_constant< int, Is + 3 >{}, baseq[Is + 3], Wqs[Is + 3] ), ... );
      }
      ( std::make_integer_sequence< int, LZ - 1 >{} );
   }

   const double* w       = A.st.w;
   const double  invc    = 1.0 / w[7];
   const bool    lane_ok = lane >= 1 && lane <= 62;
   const int     x       = t.xb + lane;

   auto step = [&]( auto sc ) {
      constexpr int s = decltype( sc )::value; // output slice z0 + s, centre q = s+1
      constexpr int q = s + 1;
      if constexpr ( q + 1 + PFD <= LZ + 1 && !PFALL )
         load_slice( std::integral_constant< int, q + 1 + PFD >{}, baseq[q + 1 + PFD], Wqs[q + 1 + PFD] );

      const int W  = Wqs[q];
      int       io = baseq[q] + ( W - ym ); // (xb, y0, z)
      // ADD / JACOBI read a second (and third) array at the output points: issue those loads now, ahead of the
      // slice's arithmetic, instead of one dependent round trip per row right before the store
      double ex0[NY], ex1[NY];
      if constexpr ( MODE != APPLY_REPLACE )
      {
         int ie = io;
#pragma unroll
         for ( int j = 0; j < NY; ++j )
         {
            if constexpr ( SOFF )
            {
               const int last8 = ( W - ( t.y0 + j ) - 1 - t.xb ) * 8;
               const int vo    = min( lane_off, last8 );
               ex0[j]          = MODE == APPLY_ADD ? zm_load2< EX_AUX >( rd, vo, ie * 8 ) : zm_load2< EX_AUX >( rr, vo, ie * 8 );
               ex1[j]          = ( MODE == APPLY_JACOBI && A.invdiag ) ? zm_load2( ri, vo, ie * 8 ) : invc;
            }
            else
            {
            const int off = ie * 8 + lane_off;
            ex0[j]        = MODE == APPLY_ADD ? zm_load< EX_AUX >( rd, off ) : zm_load< EX_AUX >( rr, off );
            ex1[j]        = ( MODE == APPLY_JACOBI && A.invdiag ) ? zm_load( ri, off ) : invc; // nt here: 18.8 -> 21.0 us
            }
            ie += W - ( t.y0 + j );
         }
      }
      double accs[NY];
      if constexpr ( FACT == 2 && ( ABL & 8 ) == 0 )
      {
         double pe[NY], pw[NY], s1[NY], s2[NY];
#define ZM_ROWS( expr )               \
   _Pragma( "unroll" ) for ( int j = 0; j < NY; ++j ) \
   {                                  \
      const double am = S[q][j], a0 = S[q][j + 1], ap = S[q][j + 2];          \
      const double um = S[q + 1][j], u0 = S[q + 1][j + 1];                    \
      const double d0 = S[q - 1][j + 1], dp = S[q - 1][j + 2];                \
      (void) am, (void) a0, (void) ap, (void) um, (void) u0, (void) d0, (void) dp; \
      expr;                           \
   }
         // pe: what the lane to the left needs from this lane (E, SE, TSE, BE); pw: what the lane to the right needs
         // (W, TW, BNW, NW); s1, s2: the seven unshifted terms
         ZM_ROWS( pe[j] = w[8] * a0 )
         ZM_ROWS( pw[j] = w[6] * a0 )
         ZM_ROWS( s1[j] = w[3] * dp )
         ZM_ROWS( s2[j] = w[4] * am )
         ZM_ROWS( pe[j] = fma( w[5], am, pe[j] ) )
         ZM_ROWS( pw[j] = fma( w[13], u0, pw[j] ) )
         ZM_ROWS( s1[j] = fma( w[10], ap, s1[j] ) )
         ZM_ROWS( s2[j] = fma( w[11], um, s2[j] ) )
         ZM_ROWS( pe[j] = fma( w[12], um, pe[j] ) )
         ZM_ROWS( pw[j] = fma( w[2], dp, pw[j] ) )
         ZM_ROWS( s1[j] = fma( w[0], d0, s1[j] ) )
         ZM_ROWS( s2[j] = fma( w[7], a0, s2[j] ) )
         ZM_ROWS( pe[j] = fma( w[1], d0, pe[j] ) )
         ZM_ROWS( pw[j] = fma( w[9], ap, pw[j] ) )
         ZM_ROWS( s1[j] = fma( w[14], u0, s1[j] ) )
         ZM_ROWS( s1[j] = s1[j] + s2[j] )
         ZM_ROWS( pe[j] = zm_lane_plus_1( pe[j] ) + zm_lane_minus_1( pw[j] ) )
         ZM_ROWS( accs[j] = pe[j] + s1[j] )
#undef ZM_ROWS
      }
      else
      {
#pragma unroll
      for ( int j = 0; j < NY; ++j )
      {
         const double am = S[q][j], a0 = S[q][j + 1], ap = S[q][j + 2];
         const double um = S[q + 1][j], u0 = S[q + 1][j + 1];
         const double d0 = S[q - 1][j + 1], dp = S[q - 1][j + 2];
         double       acc;
         if constexpr ( ( ABL & 8 ) != 0 )
            acc = am + a0 + ap + um + u0 + d0 + dp;
         else if constexpr ( FACT == 1 )
         {
            double pe = w[8] * a0; // what the lane to the left needs from this lane: E, SE, TSE, BE
            pe        = fma( w[5], am, pe );
            pe        = fma( w[12], um, pe );
            pe        = fma( w[1], d0, pe );
            double pw = w[6] * a0; // what the lane to the right needs: W, TW, BNW, NW
            pw        = fma( w[13], u0, pw );
            pw        = fma( w[2], dp, pw );
            pw        = fma( w[9], ap, pw );
            acc       = zm_lane_plus_1( pe ) + zm_lane_minus_1( pw );
            acc       = fma( w[3], dp, acc );  // BN
            acc       = fma( w[10], ap, acc ); // N
            acc       = fma( w[4], am, acc );  // S
            acc       = fma( w[11], um, acc ); // TS
            acc       = fma( w[0], d0, acc );  // BC
            acc       = fma( w[7], a0, acc );  // C
            acc       = fma( w[14], u0, acc ); // TC
         }
         else
         {
            // the reference's summation order (apply_3D_macrocell_vertexdof_to_vertexdof_replace)
            acc = w[6] * zm_lane_minus_1( a0 );             // W
            acc = fma( w[3], dp, acc );                     // BN
            acc = fma( w[10], ap, acc );                    // N
            acc = fma( w[5], zm_lane_plus_1( am ), acc );   // SE
            acc = fma( w[12], zm_lane_plus_1( um ), acc );  // TSE
            acc = fma( w[1], zm_lane_plus_1( d0 ), acc );   // BE
            acc = fma( w[8], zm_lane_plus_1( a0 ), acc );   // E
            acc = fma( w[13], zm_lane_minus_1( u0 ), acc ); // TW
            acc = fma( w[2], zm_lane_minus_1( dp ), acc );  // BNW
            acc = fma( w[9], zm_lane_minus_1( ap ), acc );  // NW
            acc = fma( w[4], am, acc );                     // S
            acc = fma( w[11], um, acc );                    // TS
            acc = fma( w[0], d0, acc );                     // BC
            acc = fma( w[7], a0, acc );                     // C
            acc = fma( w[14], u0, acc );                    // TC
         }
         accs[j] = acc;
      }
      }
#pragma unroll
      for ( int j = 0; j < NY; ++j )
      {
         const int    R      = W - ( t.y0 + j );
         const double a0     = S[q][j + 1];
         const double acc    = accs[j];
         const bool   active = lane_ok && x <= R - 2 && s < t.nz;
         const int    off    = io * 8 + lane_off;
         double       out;
         if ( MODE == APPLY_REPLACE )
            out = acc;
         else if ( MODE == APPLY_ADD )
            out = acc + ex0[j];
         else
            out = a0 + A.relax * ( ex1[j] * ( ex0[j] - acc ) );
         if constexpr ( ( ABL & 4 ) != 0 )
         {
            if ( out == 1.2345e-300 )
               zm_store< 0 >( rd, off, out );
         }
         else if constexpr ( SOFF )
         {
            // outputs are lanes 1 .. min( 62, R - 2 - xb ) of slices that exist: one unsigned compare of (lane - 1)
            const int      cnt = s < t.nz ? min( 62, R - 2 - t.xb ) : 0; // wave-uniform
            const unsigned lm1 = (unsigned) ( lane - 1 );
            zm_store2< ST_AUX >( rd, lm1 < (unsigned) max( cnt, 0 ) ? lane_off : -8, io * 8, out );
         }
         else
            zm_store< ST_AUX >( rd, active ? off : -8, out );
         io += R;
      }
   };

   // fully unrolled march
   [&]< int... Is >( std::integer_sequence< int, Is... > ) { ( step( std::integral_constant< int, Is >{} ), ... ); }
   ( std::make_integer_sequence< int, LZ >{} );
}

// host: bricks of NY rows x 62 outputs x LZ slices, ordered z-chunk, y-chunk, x-chunk (memory order)
inline void build_brick_tasks( int level, int NY, int LZ, std::vector< BrickTask >& out )
{
   const int N = ( 1 << level ) + 1;
   out.clear();
   for ( int z0 = 1; z0 <= N - 4; z0 += LZ )
   {
      const int W = N - z0; // row-0 length of the first output slice
      for ( int y0 = 1; y0 <= W - 3; y0 += NY )
      {
         const int xmax = W - y0 - 2; // last interior x of the brick's longest row
         for ( int x0 = 1; x0 <= xmax; x0 += 62 )
         {
            BrickTask t{};
            t.xb = x0 - 1;
            t.y0 = y0;
            t.W0 = W + 1;
            t.i0 = slice_start( N, z0 - 1 ) + row_start( W + 1, y0 - 1 ) + t.xb;
            t.nz = std::min( LZ, N - 4 - z0 + 1 );
            out.push_back( t );
         }
      }
   }
}

} // namespace hyteg_hip
