// Rolled form of the z-march apply kernel (kernels_apply_zmarch.hpp): same bricks, same loads, same arithmetic and
// summation order, but the march over the brick's slices is a LOOP whose body is compiled once (UNR = 1: the three
// live slices are rotated with register copies; UNR = 4: four steps per trip, the slice registers rotate by
// renaming) instead of LZ fully unrolled steps.
//
// Hypothesis tested (round 2): at level 8 every wave runs exactly one brick, so every instruction of the unrolled
// kernel (10 KB of code) is executed once per wave; if the instruction cache were cold at every launch (the 46 MB that
// stream through L2 between two launches evict the code), a loop body fetched once should win.
// Result: it does not.  exp/icache_probe.hip: straight-line code costs 4.6 cycles per f64 FMA whether or not a
// 46 MB copy runs between the launches (2 ... 64 KB of code), i.e. instruction fetch keeps up with a cold cache; and
// this kernel is 4-8% SLOWER than the unrolled one at levels 7-9 (copies / renaming: level 8 13.96 / 13.21 vs 12.75 us on
// the same box; profiles/r02_apply_zloop_and_icache_probe.txt).  Kept for the harness only.
#pragma once

#include "../kernels_apply_zmarch.hpp"

namespace hyteg_hip {

template < int MODE, int NY, int UNR, int EX_AUX = 0 >
__global__ __launch_bounds__( 64 * kZMarchWavesPerBlock ) void p1_apply_zloop_kernel( const ZMarchArgs A )
{
   static_assert( UNR == 1 || UNR == 4, "UNR: 1 (register copies) or 4 (rotation by renaming)" );
   ZM_TRACE( 0 );
   int b = blockIdx.x;
   if ( A.xcd_chunk > 0 )
      b = ( blockIdx.x & 7 ) * A.xcd_chunk + ( blockIdx.x >> 3 );
   const int task = __builtin_amdgcn_readfirstlane( b * kZMarchWavesPerBlock + ( threadIdx.x >> 6 ) );
   if ( task >= A.ntasks )
      return;
   const BrickTask t    = A.tasks[task];
   const int lane = threadIdx.x & 63;
   ZM_TRACE( 1 );

   constexpr int kStAux = 2; // nontemporal
   const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc( static_cast< double* >( const_cast< void* >( A.src ) ), 0, A.bytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc( static_cast< double* >( A.dst ), 0, A.bytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rr =
       __builtin_amdgcn_make_buffer_rsrc( static_cast< double* >( const_cast< void* >( MODE == APPLY_JACOBI ? A.rhs : A.src ) ), 0, A.bytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
       static_cast< double* >( const_cast< void* >( ( MODE == APPLY_JACOBI && A.invdiag ) ? A.invdiag : A.src ) ), 0, A.bytes, 0x00020000 );

   const int lane_off = lane * 8;
   const int ym       = t.y0 - 1; // first row held per slice

   // S[k][r]: row ym+r (r = 0..NY+1) of one slice, x = xb + lane; which slice a k holds rotates with the march
   double S[4][NY + 2];

   auto load_slice = [&]( double( &Sq )[NY + 2], int base_q, int W_q ) {
      int ix = base_q;
#pragma unroll
      for ( int r = 0; r < NY + 2; ++r )
      {
         const int last8 = ( W_q - ( ym + r ) - 1 - t.xb ) * 8; // byte offset of the row's last entry from lane 0's
         Sq[r]           = zm_load2< double, 0 >( rs, min( lane_off, last8 ), ix * 8 );
         ix += W_q - ( ym + r ); // next row of the same slice
      }
   };

   const double* w    = A.st.w;
   const double  invc = 1.0 / w[7];

   // output slice with centre rows Sc (slice of row-0 length W, (xb, ym, z) at element basec), Sd below, Su above
   auto step = [&]( const double( &Sd )[NY + 2], const double( &Sc )[NY + 2], const double( &Su )[NY + 2], int basec, int W ) {
      int    io = basec + ( W - ym ); // (xb, y0, z)
      double ex0[NY], ex1[NY];
      if constexpr ( MODE != APPLY_REPLACE )
      {
         int ie = io;
#pragma unroll
         for ( int j = 0; j < NY; ++j )
         {
            const int last8 = ( W - ( t.y0 + j ) - 1 - t.xb ) * 8;
            const int vo    = min( lane_off, last8 );
            ex0[j]          = MODE == APPLY_ADD ? zm_load2< EX_AUX >( rd, vo, ie * 8 ) : zm_load2< EX_AUX >( rr, vo, ie * 8 );
            ex1[j]          = ( MODE == APPLY_JACOBI && A.invdiag ) ? zm_load2< double >( ri, vo, ie * 8 ) : invc;
            ie += W - ( t.y0 + j );
         }
      }
#pragma unroll
      for ( int j = 0; j < NY; ++j )
      {
         const double am = Sc[j], a0 = Sc[j + 1], ap = Sc[j + 2];
         const double um = Su[j], u0 = Su[j + 1];
         const double d0 = Sd[j + 1], dp = Sd[j + 2];
         double       pe = w[8] * a0; // what the lane to the left needs from this lane: E, SE, TSE, BE
         pe              = fma( w[5], am, pe );
         pe              = fma( w[12], um, pe );
         pe              = fma( w[1], d0, pe );
         double pw       = w[6] * a0; // what the lane to the right needs: W, TW, BNW, NW
         pw              = fma( w[13], u0, pw );
         pw              = fma( w[2], dp, pw );
         pw              = fma( w[9], ap, pw );
         double acc      = zm_lane_plus_1( pe ) + zm_lane_minus_1( pw );
         acc             = fma( w[3], dp, acc );  // BN
         acc             = fma( w[10], ap, acc ); // N
         acc             = fma( w[4], am, acc );  // S
         acc             = fma( w[11], um, acc ); // TS
         acc             = fma( w[0], d0, acc );  // BC
         acc             = fma( w[7], a0, acc );  // C
         acc             = fma( w[14], u0, acc ); // TC

         const int R = W - ( t.y0 + j );
         double    out;
         if ( MODE == APPLY_REPLACE )
            out = acc;
         else if ( MODE == APPLY_ADD )
            out = acc + ex0[j];
         else
            out = a0 + A.relax * ( ex1[j] * ( ex0[j] - acc ) );
         const int      cnt = min( 62, R - 2 - t.xb ); // outputs: lanes 1 .. cnt (wave-uniform)
         const unsigned lm1 = (unsigned) ( lane - 1 );
         zm_store2< double, kStAux >( rd, lm1 < (unsigned) max( cnt, 0 ) ? lane_off : -8, io * 8, out );
         io += R;
      }
   };

   // (basen, Wn): the next slice to load; (basec, Wc): the centre slice of the next output slice
   int basen = t.i0, Wn = t.W0;
   load_slice( S[0], basen, Wn );
   basen += tri( Wn ) - ym, Wn -= 1;
   int basec = basen, Wc = Wn;
   load_slice( S[1], basen, Wn );
   basen += tri( Wn ) - ym, Wn -= 1;
   load_slice( S[2], basen, Wn );
   basen += tri( Wn ) - ym, Wn -= 1;
   ZM_TRACE( 2 );
   ZM_TRACE( 3 );

   if constexpr ( UNR == 1 )
   {
#pragma nounroll
      for ( int s = 0; s < t.nz; ++s )
      {
         if ( s + 1 < t.nz )
            load_slice( S[3], basen, Wn ); // "up" slice of the next output slice
         step( S[0], S[1], S[2], basec, Wc );
#pragma unroll
         for ( int r = 0; r < NY + 2; ++r )
         {
            S[0][r] = S[1][r];
            S[1][r] = S[2][r];
            S[2][r] = S[3][r];
         }
         basec += tri( Wc ) - ym, Wc -= 1;
         basen += tri( Wn ) - ym, Wn -= 1;
      }
   }
   else
   {
#pragma nounroll
      for ( int s0 = 0; s0 < t.nz; s0 += 4 )
      {
         [&]< int... U >( std::integer_sequence< int, U... > ) {
            ( [&] {
               if ( s0 + U < t.nz )
               {
                  if ( s0 + U + 1 < t.nz )
                     load_slice( S[( U + 3 ) & 3], basen, Wn );
                  step( S[U & 3], S[( U + 1 ) & 3], S[( U + 2 ) & 3], basec, Wc );
                  basec += tri( Wc ) - ym, Wc -= 1;
                  basen += tri( Wn ) - ym, Wn -= 1;
               }
            }(),
              ... );
         }
         ( std::make_integer_sequence< int, 4 >{} );
      }
   }
   ZM_TRACE( 4 );
}

} // namespace hyteg_hip
