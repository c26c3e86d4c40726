// Device kernels for the 15-point constant-stencil apply / fused Jacobi on one macro-cell.
//
// Mapping (see DESIGN.md "apply kernel"): the macro-cell array is linear (x fastest, then y, then z),
// so a run of T consecutive entries of slice z ("tile") needs exactly three CONTIGUOUS spans of the
// source array: rows y-1..y+1 of slice z, and the matching rows of slices z-1 and z+1.  One workgroup
// stages the three spans into LDS with coalesced 16-byte loads (each source element is read from
// L2/HBM once per workgroup that needs it), then every thread evaluates its outputs with 15
// conflict-free ds_read_b64 and writes dst coalesced.
//
// For an entry with global index i in row y of slice z (W = N-z, R = W-y, S0 = tri(W), Sm = tri(W+1)):
//   slice z  : W i-1   E i+1   N i+R   NW i+R-1   S i-R-1   SE i-R
//   slice z+1: TC i+S0-y   TW TC-1   TS i+S0-W   TSE TS+1
//   slice z-1: BC i-Sm+y   BE BC+1   BN i-Sm+W+1 BNW BN-1
// All of these are monotone in i, so the spans are fixed by the tile's first and last entry.
#pragma once

#include "common.hpp"

namespace hyteg_hip {

constexpr int kApplyThreads = 256;

struct ApplyArgs
{
   double*       dst;
   const double* src;
   const double* rhs;     // JACOBI only
   const double* invdiag; // JACOBI only, may be null
   const Tile*   tiles;
   int           ntiles;
   int           N;     // width = 2^level + 1
   int           total; // number of entries of the cell array
   int           xcd_chunk; // tiles per XCD group (0: identity block->tile map)
   double        relax;
   Stencil15     st;
};

// number of doubles of dynamic LDS the tiled kernel needs for tile capacity T at width N
__host__ __device__ inline int apply_lds_doubles( int T, int N ) { return 3 * T + 4 * N + 32; }

// copy src[lo .. lo+len) to lds[0 .. len): lo and len even, src 16-byte aligned => 16-byte accesses
__device__ inline void stage_span( double* lds, const double* __restrict__ src, int lo, int len, int total, bool vec_ok )
{
   if ( vec_ok )
   {
      const int      npairs = len >> 1;
      const double2* s2     = reinterpret_cast< const double2* >( src + lo );
      double2*       l2     = reinterpret_cast< double2* >( lds );
      // the last pair may straddle the end of the array when `total` is odd
      const int safe_pairs = ( lo + len <= total ) ? npairs : npairs - 1;
      for ( int k = threadIdx.x; k < safe_pairs; k += kApplyThreads )
         l2[k] = s2[k];
      if ( safe_pairs < npairs && threadIdx.x == 0 )
      {
         const int k = 2 * safe_pairs;
         lds[k]      = src[lo + k];
         lds[k + 1]  = 0.0;
      }
   }
   else
   {
      for ( int k = threadIdx.x; k < len; k += kApplyThreads )
         lds[k] = ( lo + k < total ) ? src[lo + k] : 0.0;
   }
}

template < int MODE, bool VEC >
__global__ __launch_bounds__( kApplyThreads ) void p1_apply_tiled_kernel( const ApplyArgs A )
{
   extern __shared__ __attribute__( ( aligned( 16 ) ) ) double lds[];

   int t = blockIdx.x;
   if ( A.xcd_chunk > 0 )
      t = ( blockIdx.x & 7 ) * A.xcd_chunk + ( blockIdx.x >> 3 ); // blocks b, b+8, .. share an XCD: give them adjacent tiles
   if ( t >= A.ntiles )
      return;

   const Tile tl   = A.tiles[t];
   const int  N    = A.N;
   const int  W    = N - tl.z;
   const int  S0   = tri( W );
   const int  Sm   = tri( W + 1 );
   const int  Ra   = W - tl.ya;
   const int  Rb   = W - tl.yb;
   const int  last = tl.a + tl.cnt - 1;

   // spans (inclusive bounds), rounded to even/odd so that every span starts 16-byte aligned
   int mid_lo = tl.a - ( Ra + 1 ), mid_hi = last + Rb;
   int up_lo = tl.a + S0 - W, up_hi = last + S0 - tl.yb;
   int dn_lo = tl.a - Sm + tl.ya, dn_hi = last - Sm + W + 1;
   mid_lo &= ~1;
   up_lo &= ~1;
   dn_lo &= ~1;
   const int top = ( A.total - 1 ) | 1; // last index, rounded up to odd (stage_span guards the overhang)
   mid_hi        = min( mid_hi | 1, top );
   up_hi         = min( up_hi | 1, top );
   dn_hi         = min( dn_hi | 1, top );
   const int mid_len = mid_hi - mid_lo + 1, up_len = up_hi - up_lo + 1, dn_len = dn_hi - dn_lo + 1;

   double* lmid = lds;
   double* lup  = lmid + mid_len;
   double* ldn  = lup + up_len;

   stage_span( lmid, A.src, mid_lo, mid_len, A.total, VEC );
   stage_span( lup, A.src, up_lo, up_len, A.total, VEC );
   stage_span( ldn, A.src, dn_lo, dn_len, A.total, VEC );
   __syncthreads();

   const int     s0   = slice_start( N, tl.z );
   const double* w    = A.st.w;
   const double  invc = 1.0 / w[7];
   for ( int e = threadIdx.x; e < tl.cnt; e += kApplyThreads )
   {
      const int i = tl.a + e;
      const int j = i - s0;
      const int y = row_of( W, j );
      const int x = j - row_start( W, y );
      const int R = W - y;
      if ( x < 1 || x > R - 2 )
         continue;
      const double* m = lmid + ( i - mid_lo );
      const double* u = lup + ( i + S0 - up_lo );
      const double* d = ldn + ( i - Sm - dn_lo );
      const double  c = m[0];
      double        acc;
      acc = w[6] * m[-1];
      acc = fma( w[3], d[W + 1], acc );
      acc = fma( w[10], m[R], acc );
      acc = fma( w[5], m[-R], acc );
      acc = fma( w[12], u[-W + 1], acc );
      acc = fma( w[1], d[y + 1], acc );
      acc = fma( w[8], m[1], acc );
      acc = fma( w[13], u[-y - 1], acc );
      acc = fma( w[2], d[W], acc );
      acc = fma( w[9], m[R - 1], acc );
      acc = fma( w[4], m[-R - 1], acc );
      acc = fma( w[11], u[-W], acc );
      acc = fma( w[0], d[y], acc );
      acc = fma( w[7], c, acc );
      acc = fma( w[14], u[-y], acc );
      if ( MODE == APPLY_REPLACE )
         A.dst[i] = acc;
      else if ( MODE == APPLY_ADD )
         A.dst[i] = acc + A.dst[i];
      else
      {
         const double invd = A.invdiag ? A.invdiag[i] : invc;
         A.dst[i]          = c + A.relax * ( invd * ( A.rhs[i] - acc ) );
      }
   }
}

} // namespace hyteg_hip
