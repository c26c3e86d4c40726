// NEGATIVE RESULT of round 2 (kept so that the measurement can be repeated; built only by exp/build_trace.sh into
// sor_trace_lookahead, never into the library).  A copy of ../p1_sor.hip whose blocked kernel was rewritten along the
// lines "shorten the critical path of a plane step":
//   * neighbours carried along p in registers (8 LDS reads per update instead of 16),
//   * everything but the three terms written on the previous plane summed one step ahead,
//   * a branch-free step (idle threads read clamped addresses and store to a sink) so that the compiler interleaves both sums,
//   * blocks of a wavefront taken from the grid index instead of a table, rhs loads issued together with the u loads.
// Level 8 on one box, same run: 1000 / 1034 us per forward / backward sweep against 844 / 844 us of the production
// kernel (levels 6, 7: 210 / 470 against 172 / 390).  Why (exp/sor_trace.hip -DSOR_STEPS, exp/lat_probe.hip): a plane
// step is not bound by the length of its dependent f64 chain (8.6 cycles per dependent FMA, not the 32 that round 1
// wrote down) but by what four waves -- one per SIMD -- can push through the LDS and the issue port between two barriers:
// ~330 cycles from the first address computation until the reads have arrived, ~125 for arithmetic and the write, ~80
// for the barrier.  The production kernel lets whole waves skip a step when none of their points lies on the plane (only 35 %
// of the thread-steps of a 16^3 block are active); the branch-free step makes every wave do the full work in every step.
// C-ABI entry point: in-place SOR / Gauss-Seidel sweep on one macro-cell in the reference's
// lexicographic order, executed as hyperplanes t = x + 2y + 3z.
//
// Why this is exact: the reference updates points in (z, y, x) order.  A point reads already-updated
// values from W(-1,0,0) S(0,-1,0) SE(1,-1,0) BC(0,0,-1) BE(1,0,-1) BN(0,1,-1) BNW(-1,1,-1), whose
// t is smaller by 1,2,1,3,2,1,2, and not-yet-updated values from the 7 opposite neighbours, whose t is
// larger by the same amounts.  Points on one plane never read each other, so all points of a plane can
// be updated concurrently and planes in increasing t reproduce the sequential sweep bit for bit (up to
// FMA contraction) -- that holds for the PLANE kernel.  The blocked form (levels >= 5, the default) visits the points in an
// order that respects the same dependencies, so every update reads exactly the values the sequential sweep reads, but it sums
// the 14 neighbour terms of an update in three partial chains: order-exact in its updates, reassociated within an update
// (relative differences of a few ulp; tests/test_gpu_parity.py compares at 1e-12, level 8 included).
// The backward sweep is the same planes in decreasing t.
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include <atomic>

#include "../common.hpp"
#include "../sor_dataflow.hpp"

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


// ---------------------------------------------------------------------------------------------------------
// Blocked form of the same sweep.  In the skewed coordinates (p,q,r) = (x+y+z, y+z, z) every already-updated
// neighbour of a point has all three coordinates <= the point's and every not-yet-updated one >=:
//   W(-1,0,0)  S(-1,-1,0)  SE(0,-1,0)  BC(-1,-1,-1)  BE(0,-1,-1)  BN(0,0,-1)  BNW(-1,0,-1)   and their negatives.
// So rectangular blocks of B^3 points in (p,q,r), visited in block-wavefront order P+Q+R (one launch per wavefront,
// all blocks of a wavefront concurrently), with the points inside a block visited in hyperplanes p+q+r by ONE
// workgroup from LDS, reproduce the sequential (z,y,x) sweep exactly.  Thread (ql,rl) of a workgroup owns the row
// (q,r) of its block and marches along p (= along x), staggered by ql+rl steps.
// ---------------------------------------------------------------------------------------------------------
constexpr int kB  = 16;     // block edge
constexpr int kBH = kB + 2; // with halo
constexpr int kBP = kB + 6; // LDS row length along p: halo + two padding entries per side (addresses of idle steps, never used)

struct SorBlock
{
   short P, Q, R, pad;
};

struct SorBlockArgs
{
   double*       u;
   const double* rhs;
   int           N;
   int           T, Q0, R0; // block wavefront P + Q + R of this launch; Q = Q0 + blockIdx.x, R = R0 + blockIdx.y
   double        relax, one_minus_relax, invc;
   Stencil15     st;
};
// batched form: blockIdx.z = cell; weights from the device table [cell][15][15] (row 14 = inner stencil)
struct SorBlockBatchArgs
{
   double*       u[HYTEG_HIP_MAX_BATCH];
   const double* rhs[HYTEG_HIP_MAX_BATCH];
   const double* stencils;
   int           N;
   int           T, Q0, R0;
   double        relax, one_minus_relax;
};

__host__ __device__ inline int lds_index( int pl, int ql, int rl ) { return ( ( rl + 1 ) * kBH + ( ql + 1 ) ) * kBP + ( pl + 3 ); }

// does block (P,Q,R) of a cell with n = 2^level hold an interior point?  (r >= 1, q >= r + 1, p >= q + 1, p <= n - 1)
__host__ __device__ inline bool sor_block_nonempty( int n, int P, int Q, int R )
{
   const int nb = ( n + kB - 1 ) / kB;
   if ( R < 0 || Q < R || P < Q || P >= nb )
      return false;
   const int rmin = max( 1, R * kB ), rmax = min( n - 3, R * kB + kB - 1 );
   if ( rmin > rmax )
      return false;
   const int qmin = max( rmin + 1, Q * kB ), qmax = min( n - 2, Q * kB + kB - 1 );
   if ( qmin > qmax )
      return false;
   const int pmin = max( qmin + 1, P * kB ), pmax = min( n - 1, P * kB + kB - 1 );
   return pmin <= pmax;
}

struct SorBlockView
{
   double*       u;
   const double* rhs;
   const double* w;
   double        relax, one_minus_relax, invc;
   int           N;
};
// timestamp hook of the developer harness exp/sor_trace.hip; expands to nothing in the library
#ifndef SOR_TRACE
#define SOR_TRACE( slot )
#endif
#ifndef SOR_STEP_STAMP
#define SOR_STEP_STAMP( k )
#define SOR_STEP_DECL
#define SOR_STEP_FLUSH
#endif

template < bool BW >
__device__ inline void sor_block_body( const SorBlockView A, const SorBlock blk );

// The blocks of a wavefront are the (Q,R) of a 2-D grid with P = T - Q - R: no block table to load before the first
// useful instruction; the workgroups of empty or non-existent blocks leave at once.
template < bool BW >
__global__ __launch_bounds__( kB* kB ) void p1_sor_block_kernel( const SorBlockArgs A )
{
   const int Q = A.Q0 + blockIdx.x, R = A.R0 + blockIdx.y, P = A.T - Q - R;
   if ( !sor_block_nonempty( A.N - 1, P, Q, R ) )
      return;
   sor_block_body< BW >( SorBlockView{ A.u, A.rhs, A.st.w, A.relax, A.one_minus_relax, A.invc, A.N }, SorBlock{ (short) P, (short) Q, (short) R, 0 } );
}
template < bool BW >
__global__ __launch_bounds__( kB* kB ) void p1_sor_block_batch_kernel( const SorBlockBatchArgs A )
{
   const int Q = A.Q0 + blockIdx.x, R = A.R0 + blockIdx.y, P = A.T - Q - R;
   if ( !sor_block_nonempty( A.N - 1, P, Q, R ) )
      return;
   const int     cell = blockIdx.z;
   const double* w    = A.stencils + (size_t) cell * 225 + 14 * 15;
   sor_block_body< BW >( SorBlockView{ A.u[cell], A.rhs[cell], w, A.relax, A.one_minus_relax, 1.0 / w[7], A.N },
                         SorBlock{ (short) P, (short) Q, (short) R, 0 } );
}

template < bool BW >
__device__ inline void sor_block_body( const SorBlockView A, const SorBlock blk )
{
   __shared__ double lu[kBP * kBH * kBH];
   __shared__ double lr[kB * kB * kB]; // rhs of the block (a per-thread register row would be runtime-indexed -> scratch)
   __shared__ double sink[kB * kB];    // where idle threads store (keeps the step free of branches)
   const int         N = A.N, n = N - 1;
   const int         p0 = blk.P * kB, q0 = blk.Q * kB, r0 = blk.R * kB;
   double*           u = A.u;
   SOR_TRACE( 0 );

   // Rows of constant (q,r) are contiguous in memory along p (x = p - q).  The array index of p = 0 of each of the
   // 18 x 18 rows of the block + halo is computed once (a per-element index polynomial costs ~70 instructions; with
   // 39 + 16 elements per thread that was a third of the block time); -1: the row is outside the array.
   __shared__ int rowBase[kBH * kBH];
   for ( int t = threadIdx.x; t < kBH * kBH; t += kB * kB )
   {
      const int q = q0 + t % kBH - 1, r = r0 + t / kBH - 1;
      rowBase[t]  = ( r >= 0 && q >= r && q <= n ) ? slice_start( N, r ) + row_start( N - r, q - r ) - q : -1;
   }
   __syncthreads();
   SOR_TRACE( 1 );

   // stage the block with its halo, and its right-hand side
   // (constant trip counts + full unrolling: ALL loads of a thread -- 23 of u, 16 of rhs -- are in flight together; a
   //  rolled loop waits one memory round trip per iteration, and so did the rhs when it was loaded after u was in LDS)
   constexpr int kStageU = ( kBH * kBH * kBH + kB * kB - 1 ) / ( kB * kB );
   double        stage[kStageU];
#pragma unroll
   for ( int it = 0; it < kStageU; ++it )
   {
      const int idx = it * ( kB * kB ) + threadIdx.x;
      const int row = idx / kBH; // (rl + 1) * kBH + (ql + 1)
      const int p = p0 + idx % kBH - 1, q = q0 + row % kBH - 1;
      const int base = idx < kBH * kBH * kBH ? rowBase[row] : -1;
      stage[it]      = ( base >= 0 && p >= q && p <= n ) ? u[base + p] : 0.0;
   }
   // rhs and write-back: thread handles the entries idx = it * 256 + tid of the 16^3 block: pl = tid % 16, row (tid / 16, it)
   const int pw = p0 + threadIdx.x % kB, qw = q0 + threadIdx.x / kB;
   double    stageR[kB];
#pragma unroll
   for ( int it = 0; it < kB; ++it )
   {
      const int  r2   = r0 + it;
      const int  base = rowBase[( it + 1 ) * kBH + threadIdx.x / kB + 1];
      const bool ok   = r2 >= 1 && qw >= r2 + 1 && pw >= qw + 1 && pw <= n - 1;
      stageR[it]      = ok ? A.rhs[base + pw] : 0.0;
   }
   SOR_TRACE( 2 );
#pragma unroll
   for ( int it = 0; it < kStageU; ++it )
   {
      const int idx = it * ( kB * kB ) + threadIdx.x;
      if ( idx < kBH * kBH * kBH )
         lu[( idx / kBH ) * kBP + idx % kBH + 2] = stage[it];
   }
#pragma unroll
   for ( int it = 0; it < kB; ++it )
      lr[it * ( kB * kB ) + threadIdx.x] = stageR[it];
   const int ql = threadIdx.x % kB, rl = threadIdx.x / kB;
   const int q = q0 + ql, r = r0 + rl;
   const int y = q - r, z = r;
   // interior row? (y >= 1, z >= 1, and at least x = 1 fits: 1 + y + z <= n - 2)
   const bool row_ok = y >= 1 && z >= 1 && y + z <= n - 2;
   // the interior points of the row inside this block: pl in [plo, phi]   (x = p - q >= 1, x + y + z = p <= n - 1)
   const int plo = row_ok ? max( 0, q + 1 - p0 ) : 1;
   const int phi = row_ok ? min( kB - 1, n - 1 - p0 ) : 0;
   __syncthreads();
   SOR_TRACE( 3 );

   // One hyperplane per step; thread (ql,rl) updates the point C = (s - ql - rl, ql, rl) of plane s.  Two things keep
   // the step short (measured with exp/sor_trace.hip and exp/lat_probe.hip: the step is bound by LDS
   // bandwidth -- 64 lanes x 8 B = 4 clocks per read instruction per wave -- and by the write -> barrier -> read turn):
   //  * values travel along p in registers.  In every (dq,dr) column of the stencil the two neighbours of C are adjacent
   //    in p, and the leading one of step s is the trailing one of step s+1 (same version: a value is read either
   //    before or after its one update, never across it), so a step reads only the 7 leading neighbours + rhs: 8 LDS
   //    reads instead of 16;
   //  * only W (the thread's own previous result), SE and BN were written on the previous plane.  Everything else of the
   //    sum is formed one step AHEAD ("base" of the point X = C + 1), so the critical path of a step is
   //    2 reads -> fma, fma, add -> write -> barrier.
   // The step is branch-free (idle threads read clamped addresses and store to `sink`), so the compiler interleaves the
   // two computations.  Backward sweep = the mirror image: all offsets negated, weight k <-> 14 - k.
   // relax / centre weight are folded into the weights: u_new = (1-relax) u + sum_k ( -relax w_k / w_c ) u_k +
   // ( relax / w_c ) rhs.  The ORDER OF UPDATES is the reference's; the order of the additions inside one update is not.
   const double* w  = A.w;
   const double  kk = A.relax * A.invc;
#define WT( k ) ( -kk * w[BW ? 14 - ( k ) : ( k )] )
   const double kW = WT( 6 ), kSE = WT( 5 ), kBN = WT( 3 );
   const double kE = WT( 8 ), kN = WT( 10 ), kNW = WT( 9 ), kTC = WT( 14 ), kTW = WT( 13 ), kTSE = WT( 12 ), kTS = WT( 11 );
   const double kS = WT( 4 ), kBC = WT( 0 ), kBE = WT( 1 ), kBNW = WT( 2 );
#undef WT
   constexpr int SG = BW ? -1 : 1;
#define OFF( dp, dq, dr ) ( SG * ( ( dp ) + ( dq ) * kBP + ( dr ) * kBP * kBH ) )
   const int cRow = lds_index( 0, ql, rl ), rRow = ( rl * kB + ql ) * kB;
   double    base = 0.0, uW = 0.0, ownC = 0.0, ownX = 0.0, nw = 0.0, tw = 0.0, ts = 0.0, bc = 0.0;
   SOR_STEP_DECL
   auto      step = [&]( int s ) {
      const int    C = s - ql - rl, X = C + SG;
      const bool   act = C >= plo && C <= phi;
      const int    cC = cRow + min( max( C, -3 ), kB + 2 ), cX = cRow + min( max( X, -2 ), kB + 1 );
      const double vSE = lu[cC + OFF( 0, -1, 0 )], vBN = lu[cC + OFF( 0, 0, -1 )];
      const double vE = lu[cX + OFF( 1, 0, 0 )], vN = lu[cX + OFF( 1, 1, 0 )], vTC = lu[cX + OFF( 1, 1, 1 )];
      const double vTSE = lu[cX + OFF( 1, 0, 1 )], vBE = lu[cX + OFF( 0, -1, -1 )];
      const double vR   = lr[rRow + min( max( X, 0 ), kB - 1 )];
      SOR_STEP_STAMP( 0 );
      // the point of this step
      const double unew = fma( kW, uW, base ) + fma( kBN, vBN, kSE * vSE );
      *( act ? &lu[cC] : &sink[threadIdx.x] ) = unew;
      // everything of the next point that is known already
      double a0 = A.one_minus_relax * ownX;
      double a1 = kk * vR;
      double a2 = kE * vE;
      double a3 = kN * vN;
      a0        = fma( kNW, nw, a0 );
      a1        = fma( kTC, vTC, a1 );
      a2        = fma( kTW, tw, a2 );
      a3        = fma( kTSE, vTSE, a3 );
      a0        = fma( kTS, ts, a0 );
      a1        = fma( kS, vSE, a1 );
      a2        = fma( kBC, bc, a2 );
      a3        = fma( kBE, vBE, a3 );
      a0        = fma( kBNW, vBN, a0 );
      base      = ( a0 + a1 ) + ( a2 + a3 );
      uW        = act ? unew : ownC;
      ownC = ownX, ownX = vE, nw = vN, tw = vTC, ts = vTSE, bc = vBE;
      SOR_STEP_STAMP( 1 );
   };
#undef OFF
   // three idle steps fill the registers (every thread is idle in them: nothing is written, no barrier needed)
   constexpr int sFirst = BW ? 3 * kB - 3 : 0;
   step( sFirst - 3 * SG );
   step( sFirst - 2 * SG );
   step( sFirst - SG );
#pragma unroll 1
   for ( int k = 0; k < 3 * kB - 2; ++k )
   {
      SOR_STEP_STAMP( 4 );
      step( sFirst + SG * k );
      SOR_STEP_STAMP( 2 );
#if defined( SOR_ABLATE ) && ( SOR_ABLATE & 1 )
      asm volatile( "s_waitcnt lgkmcnt(0)" ::: "memory" );
#else
      __syncthreads();
#endif
      SOR_STEP_STAMP( 3 );
   }
   SOR_STEP_FLUSH
   SOR_TRACE( 4 );

   // write the updated interior points back (coalesced along p = along x)
#pragma unroll
   for ( int it = 0; it < kB; ++it )
   {
      const int r2 = r0 + it;
      if ( r2 >= 1 && qw >= r2 + 1 && pw >= qw + 1 && pw <= n - 1 )
         u[rowBase[( it + 1 ) * kBH + threadIdx.x / kB + 1] + pw] = lu[lds_index( threadIdx.x % kB, threadIdx.x / kB, it )];
   }
   SOR_TRACE( 5 );
}

// per level: the non-empty block wavefronts T = P + Q + R with the bounding box of their (Q,R) (host-side only)
struct SorWavefront
{
   int T, Q0, nQ, R0, nR;
};

const std::vector< SorWavefront >& sor_wavefronts( int level )
{
   static std::mutex                                   mtx;
   static std::map< int, std::vector< SorWavefront > > cache;
   std::lock_guard< std::mutex >                       lock( mtx );
   auto                                                it = cache.find( level );
   if ( it == cache.end() )
   {
      const int                   n  = 1 << level;
      const int                   nb = ( n + kB - 1 ) / kB;
      std::vector< SorWavefront > wf;
      for ( int T = 0; T <= 3 * ( nb - 1 ); ++T )
      {
         int qlo = nb, qhi = -1, rlo = nb, rhi = -1;
         for ( int R = 0; R < nb; ++R )
            for ( int Q = R; Q < nb; ++Q )
               if ( sor_block_nonempty( n, T - Q - R, Q, R ) )
                  qlo = std::min( qlo, Q ), qhi = std::max( qhi, Q ), rlo = std::min( rlo, R ), rhi = std::max( rhi, R );
         if ( qhi >= 0 )
            wf.push_back( SorWavefront{ T, qlo, qhi - qlo + 1, rlo, rhi - rlo + 1 } );
      }
      it = cache.emplace( level, std::move( wf ) ).first;
   }
   return it->second;
}


// ---------------------------------------------------------------------------------------------------------
// Levels <= 5 of a batch: the whole cell array (<= 6,545 entries) lives in LDS and ONE workgroup runs all hyperplanes
// of the sweep, blockIdx.x = cell.  Same update order and the same summation order as p1_sor_plane_kernel.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSmallThreads = 256;
struct SorSmallArgs
{
   double*       u[HYTEG_HIP_MAX_BATCH];
   const double* rhs[HYTEG_HIP_MAX_BATCH];
   const double* stencils;
   int           N, size, backwards;
   double        relax, one_minus_relax;
};
__global__ __launch_bounds__( kSmallThreads ) void p1_sor_small_kernel( const SorSmallArgs A )
{
   extern __shared__ double lu[];
   const int                cell = blockIdx.x, N = A.N, n = N - 1;
   double*                  ug   = A.u[cell];
   const double*            rhs  = A.rhs[cell];
   const double*            w    = A.stencils + (size_t) cell * 225 + 14 * 15;
   const double             invc = 1.0 / w[7];
   for ( int i = threadIdx.x; i < A.size; i += kSmallThreads )
      lu[i] = ug[i];
   __syncthreads();
   const int tmin = 6, tmax = 1 + 2 + 3 * ( n - 3 );
   const int nzy  = ( n - 3 ) * ( n - 2 ); // candidate (z, y) pairs: z in [1, n-3], y in [1, n-2]
   for ( int k = 0; k <= tmax - tmin; ++k )
   {
      const int t = A.backwards ? tmax - k : tmin + k;
      for ( int c = threadIdx.x; c < nzy; c += kSmallThreads )
      {
         const int z = 1 + c / ( n - 2 ), y = 1 + c % ( n - 2 );
         const int x = t - 2 * y - 3 * z;
         if ( x < 1 || x + y + z > n - 1 )
            continue;
         const int W = N - z, R = W - y, S0 = tri( W ), Sm = tri( W + 1 );
         const int i = slice_start( N, z ) + row_start( W, y ) + x;
         double    acc = -w[3] * lu[i - Sm + W + 1];          // BN
         acc           = fma( -w[10], lu[i + R], acc );       // N
         acc           = fma( -w[5], lu[i - R], acc );        // SE
         acc           = fma( -w[12], lu[i + S0 - W + 1], acc ); // TSE
         acc           = fma( -w[1], lu[i - Sm + y + 1], acc );  // BE
         acc           = fma( -w[8], lu[i + 1], acc );        // E
         acc           = fma( -w[6], lu[i - 1], acc );        // W
         acc           = fma( -w[13], lu[i + S0 - y - 1], acc ); // TW
         acc           = fma( -w[2], lu[i - Sm + W], acc );      // BNW
         acc           = fma( -w[9], lu[i + R - 1], acc );       // NW
         acc           = fma( -w[4], lu[i - R - 1], acc );       // S
         acc           = fma( -w[11], lu[i + S0 - W], acc );     // TS
         acc           = fma( -w[0], lu[i - Sm + y], acc );      // BC
         acc           = fma( -w[14], lu[i + S0 - y], acc );     // TC
         acc           = acc + rhs[i];
         lu[i]         = A.relax * invc * acc + A.one_minus_relax * lu[i];
      }
      __syncthreads();
   }
   for ( int i = threadIdx.x; i < A.size; i += kSmallThreads )
      ug[i] = lu[i];
}

// which form of the cell sweep the entry points use: 0 = by level (default), 1 = one launch per hyperplane,
// 2 = blocks (one launch per block wavefront; levels <= 5 of a batch: one workgroup per cell), 3 = dataflow (one launch)
std::atomic< int > g_sorAlgorithm{ 0 };

inline bool use_dataflow( int level )
{
   const int a = g_sorAlgorithm.load( std::memory_order_relaxed );
   // opt-in only: measured slower than the blocked form at every level (p1_sor_dataflow.hip)
   return a == HYTEG_HIP_SOR_DATAFLOW && level >= kSorDataflowMinLevel;
}

} // namespace

extern "C" {

HYTEG_HIP_API int hyteg_hip_set_sor_algorithm( int algorithm )
{
   HH_REQUIRE( algorithm >= HYTEG_HIP_SOR_AUTO && algorithm <= HYTEG_HIP_SOR_DATAFLOW, "set_sor_algorithm: unknown algorithm" );
   g_sorAlgorithm.store( algorithm, std::memory_order_relaxed );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_sor_cells( int                  ncells,
                                          double* const*       u,
                                          const double* const* rhs,
                                          int                  level,
                                          const double*        stencils_dev,
                                          double               relax,
                                          int                  backwards,
                                          const unsigned*      masks,
                                          hyteg_hip_stream_t   stream )
{
   HH_REQUIRE( ncells >= 1 && ncells <= HYTEG_HIP_MAX_BATCH, "p1_sor_cells: ncells must be 1..HYTEG_HIP_MAX_BATCH" );
   HH_REQUIRE( u && rhs && stencils_dev && masks, "p1_sor_cells: null pointer" );
   HH_REQUIRE( level >= 0 && level <= HYTEG_HIP_MAX_LEVEL, "p1_sor_cells: level out of range [0,11]" );
   if ( level < 2 )
      return HYTEG_HIP_OK; // no inner points
   // cells whose inner points are not selected are left out of the batch
   int           sel[HYTEG_HIP_MAX_BATCH], m = 0;
   for ( int c = 0; c < ncells; ++c )
      if ( masks[c] & HYTEG_HIP_MASK_INNER )
      {
         HH_REQUIRE( u[c] && rhs[c] && u[c] != rhs[c], "p1_sor_cells: null or aliased arrays" );
         sel[m++] = c;
      }
   if ( m == 0 )
      return HYTEG_HIP_OK;
   const int N = ( 1 << level ) + 1;
   if ( use_dataflow( level ) )
   {
      // one launch per run of consecutive selected cells (the stencil table is indexed by the position in the batch)
      int k = 0;
      while ( k < m )
      {
         int e = k;
         while ( e + 1 < m && sel[e + 1] == sel[e] + 1 )
            ++e;
         double*       uu[HYTEG_HIP_MAX_BATCH];
         const double* rr[HYTEG_HIP_MAX_BATCH];
         for ( int j = k; j <= e; ++j )
            uu[j - k] = u[sel[j]], rr[j - k] = rhs[sel[j]];
         const int rc = launch_sor_dataflow( e - k + 1, uu, rr, level, stencils_dev + (size_t) sel[k] * 225, nullptr, relax, backwards,
                                             as_stream( stream ) );
         if ( rc != HYTEG_HIP_OK )
            return rc;
         k = e + 1;
      }
      return HYTEG_HIP_OK;
   }
   if ( level <= 5 && g_sorAlgorithm.load( std::memory_order_relaxed ) != HYTEG_HIP_SOR_PLANES )
   {
      // the stencil table is indexed by the position in the batch: compact batches need the original cell index,
      // so the kernel gets one launch per run of consecutive selected cells
      int k = 0;
      while ( k < m )
      {
         int e = k;
         while ( e + 1 < m && sel[e + 1] == sel[e] + 1 )
            ++e;
         SorSmallArgs A{};
         for ( int j = k; j <= e; ++j )
            A.u[j - k] = u[sel[j]], A.rhs[j - k] = rhs[sel[j]];
         A.stencils = stencils_dev + (size_t) sel[k] * 225;
         A.N = N, A.size = (int) tet64( N ), A.backwards = backwards ? 1 : 0;
         A.relax = relax, A.one_minus_relax = 1.0 + ( -relax );
         const size_t lds = (size_t) A.size * sizeof( double );
         if ( lds > 48 * 1024 )
            HH_CHECK_HIP( hipFuncSetAttribute( reinterpret_cast< const void* >( p1_sor_small_kernel ),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds ) );
         hipLaunchKernelGGL( p1_sor_small_kernel, dim3( e - k + 1 ), dim3( kSmallThreads ), lds, as_stream( stream ), A );
         k = e + 1;
      }
      HH_CHECK_HIP( hipGetLastError() );
      return HYTEG_HIP_OK;
   }
   const std::vector< SorWavefront >& wf = sor_wavefronts( level );
   int                                k  = 0;
   while ( k < m )
   {
      int e = k;
      while ( e + 1 < m && sel[e + 1] == sel[e] + 1 )
         ++e;
      SorBlockBatchArgs B{};
      for ( int j = k; j <= e; ++j )
         B.u[j - k] = u[sel[j]], B.rhs[j - k] = rhs[sel[j]];
      B.stencils = stencils_dev + (size_t) sel[k] * 225;
      B.N = N, B.relax = relax, B.one_minus_relax = 1.0 + ( -relax );
      const int nw = (int) wf.size();
      for ( int q = 0; q < nw; ++q )
      {
         const SorWavefront& W = wf[backwards ? nw - 1 - q : q];
         B.T = W.T, B.Q0 = W.Q0, B.R0 = W.R0;
         const dim3 grid( W.nQ, W.nR, e - k + 1 );
         if ( backwards )
            hipLaunchKernelGGL( p1_sor_block_batch_kernel< true >, grid, dim3( kB * kB ), 0, as_stream( stream ), B );
         else
            hipLaunchKernelGGL( p1_sor_block_batch_kernel< false >, grid, dim3( kB * kB ), 0, as_stream( stream ), B );
      }
      k = e + 1;
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

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
   if ( use_dataflow( level ) )
      return launch_sor_dataflow( 1, &u, &rhs, level, nullptr, w, relax, backwards, as_stream( stream ) );
   SorArgs A;
   A.u               = u;
   A.rhs             = rhs;
   A.N               = ( 1 << level ) + 1;
   A.relax           = relax;
   A.one_minus_relax = 1.0 + ( -relax );
   A.invc            = 1.0 / w[7];
   for ( int k = 0; k < 15; ++k )
      A.st.w[k] = w[k];
   if ( level >= 5 && g_sorAlgorithm.load( std::memory_order_relaxed ) != HYTEG_HIP_SOR_PLANES )
   {
      // blocked sweep: one launch per block wavefront
      const std::vector< SorWavefront >& wf = sor_wavefronts( level );
      SorBlockArgs                       B;
      B.u               = u;
      B.rhs             = rhs;
      B.N               = A.N;
      B.relax           = relax;
      B.one_minus_relax = A.one_minus_relax;
      B.invc            = A.invc;
      B.st              = A.st;
      const int nw      = (int) wf.size();
      for ( int k = 0; k < nw; ++k )
      {
         const SorWavefront& W = wf[backwards ? nw - 1 - k : k];
         B.T = W.T, B.Q0 = W.Q0, B.R0 = W.R0;
         if ( backwards )
            hipLaunchKernelGGL( p1_sor_block_kernel< true >, dim3( W.nQ, W.nR ), dim3( kB * kB ), 0, as_stream( stream ), B );
         else
            hipLaunchKernelGGL( p1_sor_block_kernel< false >, dim3( W.nQ, W.nR ), dim3( kB * kB ), 0, as_stream( stream ), B );
      }
      HH_CHECK_HIP( hipGetLastError() );
      return HYTEG_HIP_OK;
   }
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
