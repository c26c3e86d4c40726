// "z-march" kernel for the 15-point constant-stencil apply / fused Jacobi on one macro-cell.
// Replaces apply_3D_macrocell_vertexdof_to_vertexdof_{replace,add} (reference:
// src/constant_stencil_operator/P1generatedKernels/apply_3D_macrocell_vertexdof_to_vertexdof_replace.cpp:34-78)
// and the 1 apply + 3 vector passes of P1Operator::smooth_jac (src/hyteg/p1functionspace/P1Operator.hpp:429-447).
//
// One WAVE owns a brick of NY rows x 64 x-positions (lanes 1..62 produce outputs) x LZ slices and marches in +z.
// Slice z+1's rows are loaded ONCE and serve as "up" rows for slice z, centre rows for z+1 and "down" rows for z+2
// straight from registers ((LZ(NY+2)+2(NY+1))/(LZ NY) ~ 1.8 row loads per output row instead of 4), and the loads
// of slice z+1+PFD are issued before slice z is evaluated and stored, so loads and stores are in flight together in
// every wave.  Everything is fully unrolled (compile-time NY, LZ): no loop-carried register copies, exact s_waitcnt
// accounting.  x-1/x+1 neighbours come from wave-wide DPP shifts of per-direction partial sums (two shifts per
// output); addresses are wave-uniform row bases in the buffer instructions' scalar offset + lane*8 in the vector
// offset; the descriptor's range check returns 0 for / drops accesses whose vector offset is forced out of range
// (no exec masking).  dst is written with nontemporal stores (plain stores leave 22 MB of dirty lines for the
// end-of-kernel L2 write-back: 14.5 -> 10.4 us at level 8).
//
// The superseded variants and ablation switches of round 1 (reference summation order, all-loads-first, unmasked
// loads, per-lane addressing ...) and their harness are in the history (commit ceeab28, hyteg_amd/csrc/exp/apply_bench.hip);
// their measurements are profiles/r01_apply_*.txt.
//
// Index algebra: W = N-z; element (x,y,z) -> (x,y,z+1): + tri(W) - y;  (x,y,z) -> (x,y+1,z): + (W-y).
#pragma once

#include <algorithm>
#include <utility>
#include <vector>

#include "common.hpp"

namespace hyteg_hip {

constexpr int kBrickMaxSlices = 10; // LZ + 2 for the tallest brick shape in use (LZ = 8)
struct BrickTask
{
   int i0;  // element index of (xb, y0-1, z0-1): first row segment of the first slice
   int W0;  // N - (z0-1): row-0 length of slice z0-1
   int y0;  // first output row
   int xb;  // x held by lane 0 (= x0 - 1, x0 = first output x)
   int nz;  // slices of this brick that exist (<= LZ)
   int pad;
   // element index of (xb, y0-1, z0-1+q), q = 0 .. LZ+1 (base[0] = i0): the table carries them so that a wave does not
   // spend ~75 scalar instructions on triangular numbers between reading its task and issuing its first loads -- the 8
   // waves of a CU share one scalar unit and at level 8 they all do this at the same moment
   int base[kBrickMaxSlices];
};
static_assert( sizeof( BrickTask ) == 64, "BrickTask must be 64 bytes (one s_load_dwordx16)" );

constexpr int kZMarchMaxZChunks = 32; // decode mode: number of z-chunks whose first task index fits the kernel arguments
constexpr int kZMarchMaxStairs  = 5;  // decode mode: x-chunks of the longest row (N - 4 <= 310: up to level 8)

struct ZMarchArgs
{
   // cell arrays of the kernel's value type (double, or float: the reference instantiates its generated apply kernels for
   // both, apply_3D_macrocell_vertexdof_to_vertexdof_replace.cpp:96-97)
   void*            dst;
   const void*      src;
   const void*      rhs;     // JACOBI / RESIDUAL modes
   const void*      invdiag; // JACOBI only, may be null
   void*            dst2;    // RESIDUAL_F32OUT: second float output (the first Jacobi iterate of the error equation)
   double*          xacc;    // JACOBI_ACCUM: the double array the last float sweep is added to
   const BrickTask* tasks;   // table mode (DEC == false)
   int              ntasks;
   unsigned         bytes;     // size of the cell array in bytes (buffer range): entries x sizeof( value type )
   int              xcd_chunk; // workgroups per XCD group (0: identity map)
   int              N;         // 2^level + 1
   double           relax;
   Stencil15        st;
   int              zs[kZMarchMaxZChunks]; // decode mode: first task of z-chunk k (entries past the last chunk = ntasks)
};

#ifndef HYTEG_ZM_WAVES_PER_BLOCK
#define HYTEG_ZM_WAVES_PER_BLOCK 4
#endif
constexpr int kZMarchWavesPerBlock = HYTEG_ZM_WAVES_PER_BLOCK; // 1, 2, 8 measured within noise of 4 at level 8

// developer hook: the trace harness (exp/apply_trace.hip) defines ZM_TRACE( slot ) to record a timestamp per wave
#ifndef ZM_TRACE
#define ZM_TRACE( slot )
#endif

typedef int zm_v2i_t __attribute__( ( ext_vector_type( 2 ) ) );

// wave-uniform row base in the scalar offset, lane part in the vector offset (the range check sees the vector offset only)
template < typename T, int AUX = 0 >
__device__ inline T zm_load2( __amdgpu_buffer_rsrc_t r, int voff, int soff )
{
   if constexpr ( sizeof( T ) == 8 )
   {
      zm_v2i_t v = __builtin_amdgcn_raw_buffer_load_b64( r, voff, soff, AUX );
      return *reinterpret_cast< T* >( &v );
   }
   else
   {
      int v = __builtin_amdgcn_raw_buffer_load_b32( r, voff, soff, AUX );
      return *reinterpret_cast< T* >( &v );
   }
}
template < typename T, int AUX = 0 >
__device__ inline void zm_store2( __amdgpu_buffer_rsrc_t r, int voff, int soff, T d )
{
   if constexpr ( sizeof( T ) == 8 )
      __builtin_amdgcn_raw_buffer_store_b64( *reinterpret_cast< zm_v2i_t* >( &d ), r, voff, soff, AUX );
   else
      __builtin_amdgcn_raw_buffer_store_b32( *reinterpret_cast< int* >( &d ), r, voff, soff, AUX );
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
__device__ inline float zm_lane_minus_1( float v )
{
   return __int_as_float( __builtin_amdgcn_mov_dpp( __float_as_int( v ), 0x138, 0xf, 0xf, true ) );
}
__device__ inline float zm_lane_plus_1( float v )
{
   return __int_as_float( __builtin_amdgcn_mov_dpp( __float_as_int( v ), 0x130, 0xf, 0xf, true ) );
}

__host__ __device__ inline void zm_fill_bases( BrickTask& t, int LZ )
{
   int base = t.i0, Wq = t.W0;
   for ( int q = 0; q < kBrickMaxSlices; ++q )
   {
      t.base[q] = q < LZ + 2 ? base : 0;
      base += tri( Wq ) - ( t.y0 - 1 ); // (x, ym, z) -> (x, ym, z+1)
      Wq -= 1;
   }
}

// Bricks are enumerated z-chunk, y-chunk, x-chunk (memory order).  z-chunk k: z0 = 1 + LZ k, M = N - 4 - LZ k (>= 1);
// its y-chunk yc (y0 = 1 + NY yc <= M) has ( M - NY yc + 61 ) / 62 x-chunks of 62 outputs.
template < int NY, int LZ >
__host__ __device__ inline BrickTask zm_make_task( int N, int k, int yc, int xc )
{
   const int z0 = 1 + LZ * k, W = N - z0, y0 = 1 + NY * yc;
   BrickTask t{};
   t.xb = 62 * xc;
   t.y0 = y0;
   t.W0 = W + 1;
   t.i0 = slice_start( N, z0 - 1 ) + row_start( W + 1, y0 - 1 ) + t.xb;
   t.nz = LZ < N - 3 - z0 ? LZ : N - 3 - z0;
   return t; // base[] is not filled: the decode mode recomputes the slice bases in the kernel
}

// Decode mode: the brick of a task index from the z-chunk starts in the kernel arguments, scalar arithmetic only —
// no dependent table load in front of the brick's first source loads (at level 8 every wave runs ONE brick, so
// that round trip is on the critical path of the whole launch).  Within a z-chunk the number of x-chunks per
// y-chunk is a staircase K, K-1, ..., 1 (K <= 5 up to level 8); walk the stairs.
template < int NY, int LZ >
__device__ inline BrickTask zm_decode_task( const ZMarchArgs& A, int task )
{
   int k = 0, first = 0; // zs is non-decreasing: the last z-chunk whose first task is <= task
#pragma unroll
   for ( int i = 1; i < kZMarchMaxZChunks; ++i )
   {
      const bool ge = task >= A.zs[i];
      k             = ge ? i : k;
      first         = ge ? A.zs[i] : first;
   }
   int       t = task - first;
   const int M = A.N - 4 - LZ * k;
   const int K = ( M + 61 ) / 62; // x-chunks of the z-chunk's first y-chunk
   int       yc = 0, xc = 0;
   bool      found = false;
#pragma unroll
   for ( int c = kZMarchMaxStairs; c >= 1; --c )
   {
      if ( !found && c <= K )
      {
         // y-chunks with exactly c x-chunks: 62(c-1) < M - NY yc <= 62 c
         const int lo  = M - 62 * c;
         const int ylo = lo > 0 ? (int) ( (unsigned) ( lo + NY - 1 ) / (unsigned) NY ) : 0;
         const int yhi = (int) ( (unsigned) ( M - 62 * ( c - 1 ) - 1 ) / (unsigned) NY );
         const int cnt = ( yhi - ylo + 1 ) * c;
         if ( t < cnt )
         {
            const int q = (int) ( (unsigned) t / (unsigned) c ); // c is a compile-time constant here
            yc          = ylo + q;
            xc          = t - q * c;
            found       = true;
         }
         else
            t -= cnt;
      }
   }
   return zm_make_task< NY, LZ >( A.N, k, yc, xc );
}

// EX_AUX: cache policy of the second array of ADD (dst, read once: nontemporal) / JACOBI (rhs: plain, the next sweep
// re-reads it).  gfx950 "aux" bits: 1 = sc0, 2 = nt, 16 = sc1.
// PFD: how many slices ahead of the one being computed the loads run.
// T: value type of the arrays and of the arithmetic (double or float; the weights travel as doubles and are converted).
// XS: x-stride between neighbouring bricks = outputs per row segment.  62: lanes 1..62 of every row are stored (segments begin
// at x = 1 + 62 k whatever the row's address).  56: ALIGNED store windows -- every row segment of a brick begins at the first
// boundary of 8 entries of dst (doubles: 64 bytes) inside lanes 1..7 (at a boundary of 4 entries in lane 4 where it would fall
// on lane 8) and is 56 entries long (doubles: 448 bytes), so that no 64-byte line of dst is written by two waves except the
// first and the last line of a row (round 3: the PMC write traffic of the 62-wide form is 1.11 x the bytes of dst).
template < int MODE, int NY, int LZ, int EX_AUX, bool DEC, int PFD, typename T, int XS = 62, int ST_AUX = 2, int SRC_AUX = 0 >
__device__ inline void zmarch_body( const ZMarchArgs& A, const BrickTask* tasks, int ntasks, int xcd_chunk )
{
   static_assert( XS == 62 || ( XS == 56 && !DEC ), "x-stride: 62 (plain) or 56 (aligned store windows, table mode)" );
   constexpr int SZ = (int) sizeof( T );
   ZM_TRACE( 0 );
   int b = blockIdx.x;
   if ( xcd_chunk > 0 )
      b = ( blockIdx.x & 7 ) * xcd_chunk + ( blockIdx.x >> 3 );
   const int task = __builtin_amdgcn_readfirstlane( b * kZMarchWavesPerBlock + ( threadIdx.x >> 6 ) );
   if ( task >= ntasks )
      return;
   BrickTask t;
   if constexpr ( DEC )
      t = zm_decode_task< NY, LZ >( A, task );
   else
      t = tasks[task];
   const int lane = threadIdx.x & 63;
   ZM_TRACE( 1 );

   constexpr int kStAux = ST_AUX; // 2 = nontemporal (the default; 1 = sc0, 16 = sc1: measured variants)
   const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc( const_cast< void* >( A.src ), 0, A.bytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc( A.dst, 0, MODE == APPLY_RESIDUAL_F32OUT ? A.bytes / 2 : A.bytes, 0x00020000 );
   constexpr bool kHasRhs = MODE == APPLY_JACOBI || MODE == APPLY_RESIDUAL || MODE == APPLY_RESIDUAL_F32OUT || MODE == APPLY_JACOBI_ACCUM;
   const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc( const_cast< void* >( kHasRhs ? A.rhs : A.src ), 0, A.bytes, 0x00020000 );
   // mixed-precision modes: float outputs of a double kernel (half the bytes), the double accumulator of a float kernel (twice)
   const __amdgpu_buffer_rsrc_t rd2 =
       __builtin_amdgcn_make_buffer_rsrc( MODE == APPLY_RESIDUAL_F32OUT ? A.dst2 : A.dst, 0, MODE == APPLY_RESIDUAL_F32OUT ? A.bytes / 2 : A.bytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
       MODE == APPLY_JACOBI_ACCUM ? (void*) A.xacc : A.dst, 0, MODE == APPLY_JACOBI_ACCUM ? A.bytes * 2 : A.bytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
       const_cast< void* >( ( MODE == APPLY_JACOBI && A.invdiag ) ? A.invdiag : A.src ), 0, A.bytes, 0x00020000 );

   const int lane_off = lane * SZ;
   const int ym       = t.y0 - 1; // first row held per slice
   // aligned windows: entry index of dst's first byte in units of the value type, modulo one 64-byte line
   const int dst_phase = XS == 56 ? (int) ( ( reinterpret_cast< uintptr_t >( A.dst ) / SZ ) & 7 ) : 0;

   // S[q][r]: slice z0-1+q, row ym+r (r = 0..NY+1), x = xb + lane.  q = 0..LZ+1.
   T S[LZ + 2][NY + 2];

   // Lanes whose x lies beyond the end of the row being loaded re-read the row's last entry (same cache line; a
   // negative bound puts the whole row out of range: the range check returns 0 without touching the cache), so the
   // kernel never reads past the end of the source array and fetches no entries of the next row (~30% of all lanes).
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
            const int last8 = ( W_q - ( ym + r ) - 1 - t.xb ) * SZ; // byte offset of the row's last entry from lane 0's
            S[q][r]         = zm_load2< T, SRC_AUX >( rs, min( lane_off, last8 ), ix * SZ );
         }
         ix += W_q - ( ym + r ); // next row of the same slice
      }
   };

   // wave-uniform element index of (xb, ym, z0-1+q) and row-0 length of that slice
   // (bricks taller than the table's base array -- experimental shapes -- compute their slice bases here)
   constexpr bool kBasesFromTable = !DEC && LZ + 2 <= kBrickMaxSlices;
   int baseq[LZ + 2], Wqs[LZ + 2];
   {
      int base = t.i0, Wq = t.W0;
#pragma unroll
      for ( int q = 0; q < LZ + 2; ++q )
      {
         baseq[q] = !kBasesFromTable ? base : t.base[q < kBrickMaxSlices ? q : 0];
         Wqs[q]   = Wq;
         base += tri( Wq ) - ym; // (x, ym, z) -> (x, ym, z+1)
         Wq -= 1;
      }
   }
   // prologue: slices 0 .. 1+PFD
   [&]< int... Is >( std::integer_sequence< int, Is... > ) {
      ( load_slice( std::integral_constant< int, Is >{}, baseq[Is], Wqs[Is] ), ... );
   }
   ( std::make_integer_sequence < int, ( 2 + PFD <= LZ + 2 ? 2 + PFD : LZ + 2 ) > {} );
   ZM_TRACE( 2 );

   T w[15]; // the stencil in the arithmetic's precision
#pragma unroll
   for ( int i = 0; i < 15; ++i )
      w[i] = (T) A.st.w[i];
   const T invc  = (T) ( 1.0 / A.st.w[7] );
   const T relax = (T) A.relax;

   // ADD / JACOBI read a second (and third) array at the output points.  Those loads run PFD slices ahead of their use as
   // well, like the source slices (round 1 issued them at the top of the step that consumes them: the wave then waited a
   // whole memory round trip at its first store of every slice -- 16.9 us for the fused Jacobi against 9.7 us for the apply,
   // i.e. 7 us for 23 MB more).
   T          EX0[LZ][NY], EX1[LZ][NY];
   double     EXD[LZ][NY]; // JACOBI_ACCUM: the accumulator's old values
   const bool hasInv = MODE == APPLY_JACOBI && A.invdiag != nullptr;
   auto       load_extra = [&]( auto sc ) {
      constexpr int s = decltype( sc )::value;
      constexpr int q = s + 1;
      if constexpr ( MODE != APPLY_REPLACE )
      {
         const int W  = Wqs[q];
         int       ie = baseq[q] + ( W - ym ); // (xb, y0, z)
#pragma unroll
         for ( int j = 0; j < NY; ++j )
         {
            const int last8 = ( W - ( t.y0 + j ) - 1 - t.xb ) * SZ;
            const int vo    = min( lane_off, last8 );
            EX0[s][j]       = MODE == APPLY_ADD ? zm_load2< T, EX_AUX >( rd, vo, ie * SZ ) : zm_load2< T, EX_AUX >( rr, vo, ie * SZ );
            if constexpr ( MODE == APPLY_JACOBI_ACCUM )
               EXD[s][j] = zm_load2< double, 2 >( rx, min( lane * 8, last8 * 2 ), ie * 8 ); // read once, rewritten right after: nontemporal
            if constexpr ( MODE == APPLY_JACOBI )
            {
               T v = invc; // scalar inverse diagonal unless a function was given (wave-uniform branch)
               if ( hasInv )
                  v = zm_load2< T >( ri, vo, ie * SZ );
               EX1[s][j] = v;
            }
            ie += W - ( t.y0 + j );
         }
      }
   };
   if constexpr ( MODE != APPLY_REPLACE )
   {
      [&]< int... Is >( std::integer_sequence< int, Is... > ) { ( load_extra( std::integral_constant< int, Is >{} ), ... ); }
      ( std::make_integer_sequence < int, ( PFD < LZ ? PFD : LZ ) > {} );
   }

   auto step = [&]( auto sc ) {
      constexpr int s = decltype( sc )::value; // output slice z0 + s, centre q = s+1
      constexpr int q = s + 1;
      if constexpr ( q + 1 + PFD <= LZ + 1 )
         load_slice( std::integral_constant< int, q + 1 + PFD >{}, baseq[q + 1 + PFD], Wqs[q + 1 + PFD] );
      if constexpr ( s + PFD < LZ )
         load_extra( std::integral_constant< int, s + PFD >{} );

      const int W  = Wqs[q];
      int       io = baseq[q] + ( W - ym ); // (xb, y0, z)
      if constexpr ( s == 0 )
         ZM_TRACE( 3 );
#pragma unroll
      for ( int j = 0; j < NY; ++j )
      {
         const T am = S[q][j], a0 = S[q][j + 1], ap = S[q][j + 2];
         const T um = S[q + 1][j], u0 = S[q + 1][j + 1];
         const T d0 = S[q - 1][j + 1], dp = S[q - 1][j + 2];
         // the eight x-shifted terms are summed per shift direction BEFORE the lane shift (two wave shifts per output
         // instead of eight); same terms as the reference, different summation order
         T pe = w[8] * a0; // what the lane to the left needs from this lane: E, SE, TSE, BE
         pe        = fma( w[5], am, pe );
         pe        = fma( w[12], um, pe );
         pe        = fma( w[1], d0, pe );
         T pw = w[6] * a0; // what the lane to the right needs: W, TW, BNW, NW
         pw        = fma( w[13], u0, pw );
         pw        = fma( w[2], dp, pw );
         pw        = fma( w[9], ap, pw );
         T acc = zm_lane_plus_1( pe ) + zm_lane_minus_1( pw );
         acc        = fma( w[3], dp, acc );  // BN
         acc        = fma( w[10], ap, acc ); // N
         acc        = fma( w[4], am, acc );  // S
         acc        = fma( w[11], um, acc ); // TS
         acc        = fma( w[0], d0, acc );  // BC
         acc        = fma( w[7], a0, acc );  // C
         acc        = fma( w[14], u0, acc ); // TC

         const int R = W - ( t.y0 + j );
         T         out;
         if ( MODE == APPLY_REPLACE )
            out = acc;
         else if ( MODE == APPLY_ADD )
            out = acc + EX0[s][j];
         else if ( MODE == APPLY_RESIDUAL || MODE == APPLY_RESIDUAL_F32OUT )
            out = EX0[s][j] - acc; // the bits of assign( { 1, -1 }, { rhs, A src } ): one rounding of rhs - acc either way
         else if ( MODE == APPLY_JACOBI_ACCUM )
            out = a0 + relax * ( invc * ( EX0[s][j] - acc ) );
         else
            out = a0 + relax * ( EX1[s][j] * ( EX0[s][j] - acc ) );
         if constexpr ( XS == 62 )
         {
            // outputs are lanes 1 .. min( 62, R - 2 - xb ) of slices that exist: one unsigned compare of (lane - 1)
            const int      cnt = s < t.nz ? min( 62, R - 2 - t.xb ) : 0; // wave-uniform
            const unsigned lm1 = (unsigned) ( lane - 1 );
            const bool     on  = lm1 < (unsigned) max( cnt, 0 );
            if constexpr ( MODE == APPLY_RESIDUAL_F32OUT )
            {
               // r as float, and the first Jacobi iterate of A e = r from e = 0: relax * r / centre (float arithmetic)
               const float rf = (float) out;
               zm_store2< float, 0 >( rd, on ? lane * 4 : -8, io * 4, rf ); // re-read by the following float sweeps: not nontemporal
               zm_store2< float, 0 >( rd2, on ? lane * 4 : -8, io * 4, (float) relax * ( (float) invc * rf ) );
            }
            else if constexpr ( MODE == APPLY_JACOBI_ACCUM )
               zm_store2< double, kStAux >( rx, on ? lane * 8 : -8, io * 8, EXD[s][j] + (double) out );
            else
               zm_store2< T, kStAux >( rd, on ? lane_off : -8, io * SZ, out );
         }
         else
         {
            // lanes [lo, hi): lo = the lane in 1..7 whose entry begins a 64-byte line of dst (lane 4 = a 32-byte boundary when
            // that lane would be 8); the row's first brick starts at x = 1 instead; hi = lo + 56 clipped to the last inner x.
            // A neighbour brick (xb + 56) computes the same lo for this row, so the windows tile the row.
            int lo = ( ( -( io + dst_phase ) - 1 ) & 7 ) + 1; // 1 .. 8: first lane whose entry index is a multiple of 8
            lo     = lo == 8 ? 4 : lo;
            const int hi  = s < t.nz ? min( lo + 56, R - 1 - t.xb ) : 0;
            const int lo1 = t.xb == 0 ? 1 : lo;
            zm_store2< T, kStAux >( rd, (unsigned) ( lane - lo1 ) < (unsigned) max( hi - lo1, 0 ) ? lane_off : -8, io * SZ, out );
         }
         io += R;
      }
   };

   // fully unrolled march
   [&]< int... Is >( std::integer_sequence< int, Is... > ) { ( step( std::integral_constant< int, Is >{} ), ... ); }
   ( std::make_integer_sequence< int, LZ >{} );
   ZM_TRACE( 4 );
}

template < int MODE, int NY, int LZ, int EX_AUX = 0, bool DEC = false, int PFD = 1, typename T = double, int XS = 62 >
__global__ __launch_bounds__( 64 * kZMarchWavesPerBlock ) void p1_apply_zmarch_kernel( const ZMarchArgs A )
{
   zmarch_body< MODE, NY, LZ, EX_AUX, DEC, PFD, T, XS >( A, A.tasks, A.ntasks, A.xcd_chunk );
}

// The same kernel with the three values a wave needs before it can fetch its brick -- table pointer, task count, XCD chunk --
// as leading scalar arguments: built with -mllvm -amdgpu-kernarg-preload-count=4 the command processor places them in SGPRs
// at wave launch, so the task load does not wait for a kernel-argument load first (one scalar round trip less in the start-up
// chain of DESIGN 3.1).  The rest of the arguments stay in the struct.
template < int MODE, int NY, int LZ, int EX_AUX = 0, bool DEC = false, int PFD = 1, typename T = double, int XS = 62, int ST_AUX = 2, int SRC_AUX = 0 >
__global__ __launch_bounds__( 64 * kZMarchWavesPerBlock ) void p1_apply_zmarch_preload_kernel( const BrickTask* tasks, int ntasks, int xcd_chunk,
                                                                                              const ZMarchArgs A )
{
   zmarch_body< MODE, NY, LZ, EX_AUX, DEC, PFD, T, XS, ST_AUX, SRC_AUX >( A, tasks, ntasks, xcd_chunk );
}

// host: bricks of NY rows x XS outputs x LZ slices, ordered z-chunk, y-chunk, x-chunk (memory order); zs (optional)
// receives the first task index of every z-chunk followed by the total
inline void build_brick_tasks( int level, int NY, int LZ, std::vector< BrickTask >& out, std::vector< int >* zs = nullptr, int XS = 62 )
{
   const int N = ( 1 << level ) + 1;
   out.clear();
   if ( zs )
      zs->clear();
   for ( int z0 = 1; z0 <= N - 4; z0 += LZ )
   {
      if ( zs )
         zs->push_back( (int) out.size() );
      const int W = N - z0; // row-0 length of the first output slice
      for ( int y0 = 1; y0 <= W - 3; y0 += NY )
      {
         const int xmax = W - y0 - 2; // last interior x of the brick's longest row
         for ( int x0 = 1; x0 <= xmax; x0 += XS )
         {
            BrickTask t{};
            t.xb = x0 - 1;
            t.y0 = y0;
            t.W0 = W + 1;
            t.i0 = slice_start( N, z0 - 1 ) + row_start( W + 1, y0 - 1 ) + t.xb;
            t.nz = std::min( LZ, N - 4 - z0 + 1 );
            zm_fill_bases( t, LZ );
            out.push_back( t );
         }
      }
   }
   if ( zs )
      zs->push_back( (int) out.size() );
}

} // namespace hyteg_hip
