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
#include <tuple>
#include <vector>

#include <atomic>

#include <cstdlib>

#include "common.hpp"
#include "sor_dataflow.hpp"

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

struct SorBlock
{
   short P, Q, R, pad;
};

struct SorBlockArgs
{
   double*         u;
   const double*   rhs;
   const SorBlock* blocks; // blocks of this wavefront
   int             N;
   int             backwards;
   double          relax, one_minus_relax, invc;
   Stencil15       st;
};
// batched form: blockIdx.y = cell; weights from the device table [cell][15][15] (row 14 = inner stencil)
struct SorBlockBatchArgs
{
   double*         u[HYTEG_HIP_MAX_BATCH];
   const double*   rhs[HYTEG_HIP_MAX_BATCH];
   const double*   stencils;
   const SorBlock* blocks;
   int             N;
   int             backwards;
   double          relax, one_minus_relax;
};

__device__ inline int lds_index( int pl, int ql, int rl ) { return ( ( rl + 1 ) * kBH + ( ql + 1 ) ) * kBH + ( pl + 1 ); }

struct SorBlockView
{
   double*       u;
   const double* rhs;
   const double* w;
   double        relax, one_minus_relax, invc;
   int           N, backwards;
};
__device__ inline void sor_block_body( const SorBlockView A, const SorBlock blk );
// timestamp hook of the developer harness exp/sor_trace.hip; expands to nothing in the library
#ifndef SOR_TRACE
#define SOR_TRACE( slot )
#endif

__global__ __launch_bounds__( kB* kB ) void p1_sor_block_kernel( const SorBlockArgs A )
{
   sor_block_body( SorBlockView{ A.u, A.rhs, A.st.w, A.relax, A.one_minus_relax, A.invc, A.N, A.backwards }, A.blocks[blockIdx.x] );
}
__global__ __launch_bounds__( kB* kB ) void p1_sor_block_batch_kernel( const SorBlockBatchArgs A )
{
   const int     cell = blockIdx.y;
   const double* w    = A.stencils + (size_t) cell * 225 + 14 * 15;
   sor_block_body( SorBlockView{ A.u[cell], A.rhs[cell], w, A.relax, A.one_minus_relax, 1.0 / w[7], A.N, A.backwards }, A.blocks[blockIdx.x] );
}

__device__ inline void sor_block_body( const SorBlockView A, const SorBlock blk )
{
   __shared__ double lu[kBH * kBH * kBH];
   __shared__ double lr[kB * kB * kB]; // rhs of the block (a per-thread register row would be runtime-indexed -> scratch)
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

   // stage the block and its halo
   // (constant trip counts + full unrolling: all loads of a thread are in flight together; a rolled loop waits one
   //  memory round trip per iteration, which made staging 2/3 of the block time)
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
   SOR_TRACE( 2 );
#pragma unroll
   for ( int it = 0; it < kStageU; ++it )
   {
      const int idx = it * ( kB * kB ) + threadIdx.x;
      if ( idx < kBH * kBH * kBH )
         lu[idx] = stage[it];
   }
   const int ql = threadIdx.x % kB, rl = threadIdx.x / kB;
   const int q = q0 + ql, r = r0 + rl;
   const int y = q - r, z = r;
   // interior row? (y >= 1, z >= 1, and at least x = 1 fits: 1 + y + z <= n - 2)
   const bool row_ok = y >= 1 && z >= 1 && y + z <= n - 2;
   // rhs and write-back: thread handles the entries idx = it * 256 + tid of the 16^3 block: pl = tid % 16, row (tid / 16, it)
   const int  pw = p0 + threadIdx.x % kB, qw = q0 + threadIdx.x / kB;
   {
      double stageR[kB];
#pragma unroll
      for ( int it = 0; it < kB; ++it )
      {
         const int  r2   = r0 + it;
         const int  base = rowBase[( it + 1 ) * kBH + threadIdx.x / kB + 1];
         const bool ok   = r2 >= 1 && qw >= r2 + 1 && pw >= qw + 1 && pw <= n - 1;
         stageR[it]      = ok ? A.rhs[base + pw] : 0.0;
      }
#pragma unroll
      for ( int it = 0; it < kB; ++it )
         lr[it * ( kB * kB ) + threadIdx.x] = stageR[it];
   }
   __syncthreads();
   SOR_TRACE( 3 );

   const double* w = A.w;
#pragma unroll 1
   for ( int step = 0; step < 3 * kB - 2; ++step )
   {
      const int s  = A.backwards ? ( 3 * kB - 3 - step ) : step;
      const int pl = s - ql - rl;
      if ( pl >= 0 && pl < kB )
      {
         const int x = p0 + pl - q;
         if ( row_ok && x >= 1 && x + y + z <= n - 1 )
         {
            const int c = lds_index( pl, ql, rl );
#define LU( dp, dq, dr ) lu[c + ( dp ) + ( dq ) * kBH + ( dr ) * kBH * kBH]
            // 14 terms -(w_k u_k) in the reference's order (sor_3D_macrocell_P1.cpp:74), rhs last
            // three partial sums (a 15-deep dependent FMA chain per step is latency on the critical path of the
            // whole sweep); the order of the UPDATES is the reference's, the order of additions inside one update is not
            double a0 = -w[3] * LU( 0, 0, -1 );          // BN  ( 0, 1,-1)
            double a1 = -w[10] * LU( 1, 1, 0 );          // N   ( 0, 1, 0)
            double a2 = -w[5] * LU( 0, -1, 0 );          // SE  ( 1,-1, 0)
            a0        = fma( -w[12], LU( 1, 0, 1 ), a0 );   // TSE ( 1,-1, 1)
            a1        = fma( -w[1], LU( 0, -1, -1 ), a1 );  // BE  ( 1, 0,-1)
            a2        = fma( -w[8], LU( 1, 0, 0 ), a2 );    // E   ( 1, 0, 0)
            a0        = fma( -w[6], LU( -1, 0, 0 ), a0 );   // W   (-1, 0, 0)
            a1        = fma( -w[13], LU( 0, 1, 1 ), a1 );   // TW  (-1, 0, 1)
            a2        = fma( -w[2], LU( -1, 0, -1 ), a2 );  // BNW (-1, 1,-1)
            a0        = fma( -w[9], LU( 0, 1, 0 ), a0 );    // NW  (-1, 1, 0)
            a1        = fma( -w[4], LU( -1, -1, 0 ), a1 );  // S   ( 0,-1, 0)
            a2        = fma( -w[11], LU( 0, 0, 1 ), a2 );   // TS  ( 0,-1, 1)
            a0        = fma( -w[0], LU( -1, -1, -1 ), a0 ); // BC  ( 0, 0,-1)
            a1        = fma( -w[14], LU( 1, 1, 1 ), a1 );   // TC  ( 0, 0, 1)
            a2        = a2 + lr[( rl * kB + ql ) * kB + pl];
            const double acc = ( a0 + a1 ) + a2;
            lu[c]      = A.relax * A.invc * acc + A.one_minus_relax * lu[c];
#undef LU
         }
      }
      __syncthreads();
   }
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

// per level: the non-empty blocks sorted by wavefront, and the wavefront offsets
struct SorBlockTable
{
   const SorBlock*    dev = nullptr;
   std::vector< int > wavefrontStart; // size nwavefronts + 1
};

// the non-empty blocks of a level by block wavefront T = P + Q + R (entries of byT may be empty)
static std::vector< std::vector< SorBlock > > sor_blocks_by_wavefront( int level )
{
   const int n  = 1 << level;
   const int nb = ( n + kB - 1 ) / kB;
   std::vector< std::vector< SorBlock > > byT( 3 * nb );
   for ( int R = 0; R < nb; ++R )
      for ( int Q = R; Q < nb; ++Q )
         for ( int P = Q; P < nb; ++P )
         {
            // non-empty iff some interior point: r>=1, q>=r+1, p>=q+1, p<=n-1 within the block ranges
            const int rmin = std::max( 1, R * kB ), rmax = std::min( n - 3, R * kB + kB - 1 );
            if ( rmin > rmax )
               continue;
            const int qmin = std::max( rmin + 1, Q * kB ), qmax = std::min( n - 2, Q * kB + kB - 1 );
            if ( qmin > qmax )
               continue;
            const int pmin = std::max( qmin + 1, P * kB ), pmax = std::min( n - 1, P * kB + kB - 1 );
            if ( pmin > pmax )
               continue;
            byT[P + Q + R].push_back( SorBlock{ (short) P, (short) Q, (short) R, 0 } );
         }
   return byT;
}

int get_sor_blocks( int level, const SorBlockTable** out )
{
   static std::mutex                                   mtx;
   static std::map< std::pair< int, int >, SorBlockTable > cache;
   int                                                 dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   std::lock_guard< std::mutex > lock( mtx );
   auto                          key = std::make_pair( dev, level );
   auto                          it  = cache.find( key );
   if ( it == cache.end() )
   {
      const std::vector< std::vector< SorBlock > > byT = sor_blocks_by_wavefront( level );
      SorBlockTable           tab;
      std::vector< SorBlock > flat;
      tab.wavefrontStart.push_back( 0 );
      for ( auto& v : byT )
      {
         if ( v.empty() )
            continue;
         flat.insert( flat.end(), v.begin(), v.end() );
         tab.wavefrontStart.push_back( (int) flat.size() );
      }
      if ( !flat.empty() )
      {
         void* p = nullptr;
         HH_CHECK_HIP( hipMalloc( &p, flat.size() * sizeof( SorBlock ) ) );
         HH_CHECK_HIP( hipMemcpy( p, flat.data(), flat.size() * sizeof( SorBlock ), hipMemcpyHostToDevice ) );
         tab.dev = static_cast< const SorBlock* >( p );
      }
      it = cache.emplace( key, std::move( tab ) ).first;
   }
   *out = &it->second;
   return HYTEG_HIP_OK;
}


// Several consecutive sweeps of the same direction as ONE pipeline of launches.  A sweep alone is 46 launches at level 8, each
// with at most 36 of the 256 CUs busy.  Sweep s + 1 may process a block as soon as sweep s is done with every block it reads old
// values from -- the blocks B + (a,b,c), a,b,c in {0,1}, i.e. up to wavefront T(B) + 3 -- and sweep s never reads a block that
// sweep s + 1 has already touched as long as sweep s + 1 stays at least 4 wavefronts behind.  So step tau of the pipeline
// launches the blocks of wavefront tau - 4 s of every sweep s in one grid: S sweeps take 46 + 4 ( S - 1 ) launches instead of
// 46 S, with the same arithmetic in the same order per point (bit-identical to S calls of the sweep).
constexpr int kSorSweepLag = 4;
struct SorPipelineTable
{
   const SorBlock*    dev = nullptr;
   std::vector< int > stepStart; // size nsteps + 1
};
int get_sor_pipeline( int level, int nsweeps, bool backwards, const SorPipelineTable** out )
{
   static std::mutex                                                    mtx;
   static std::map< std::tuple< int, int, int, bool >, SorPipelineTable > cache;
   int                                                                  dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   std::lock_guard< std::mutex > lock( mtx );
   auto                          key = std::make_tuple( dev, level, nsweeps, backwards );
   auto                          it  = cache.find( key );
   if ( it == cache.end() )
   {
      const std::vector< std::vector< SorBlock > > byT = sor_blocks_by_wavefront( level );
      const int                                    nT  = (int) byT.size();
      SorPipelineTable                             tab;
      std::vector< SorBlock >                      flat;
      tab.stepStart.push_back( 0 );
      for ( int tau = 0; tau < nT + kSorSweepLag * ( nsweeps - 1 ); ++tau )
      {
         for ( int sw = 0; sw < nsweeps; ++sw )
         {
            const int k = tau - kSorSweepLag * sw; // position of sweep sw in its own order
            if ( k < 0 || k >= nT )
               continue;
            const auto& v = byT[(size_t) ( backwards ? nT - 1 - k : k )];
            flat.insert( flat.end(), v.begin(), v.end() );
         }
         if ( (int) flat.size() > tab.stepStart.back() )
            tab.stepStart.push_back( (int) flat.size() );
      }
      if ( !flat.empty() )
      {
         void* p = nullptr;
         HH_CHECK_HIP( hipMalloc( &p, flat.size() * sizeof( SorBlock ) ) );
         HH_CHECK_HIP( hipMemcpy( p, flat.data(), flat.size() * sizeof( SorBlock ), hipMemcpyHostToDevice ) );
         tab.dev = static_cast< const SorBlock* >( p );
      }
      it = cache.emplace( key, std::move( tab ) ).first;
   }
   *out = &it->second;
   return HYTEG_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Levels <= 5 of a batch: the whole cell array (<= 6,545 entries) lives in LDS and ONE workgroup runs all hyperplanes
// of the sweep, blockIdx.x = cell.  Same update order and the same summation order as p1_sor_plane_kernel.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSmallThreads = 256;
// the inner points of a cell sorted by hyperplane t = x + 2 y + 3 z (array order inside a plane), with what the neighbour
// offsets need; round 3: the kernel used to scan all (z, y) candidates of every plane -- 96 planes x 930 candidates at level 5,
// one integer division each, most of them off the plane: 1.5 us per plane, 137 us per level-5 sweep
struct SorPlanePoint
{
   int   i;    // array index
   short W, y; // slice width N - z, row
};
struct SorPlaneTable
{
   const SorPlanePoint* pts = nullptr;
   const int*           off = nullptr; // [nplanes + 1]
   int                  nplanes = 0;
};
int get_sor_planes( int level, const SorPlaneTable** out )
{
   static std::mutex                                     mtx;
   static std::map< std::pair< int, int >, SorPlaneTable > cache;
   int                                                   dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   std::lock_guard< std::mutex > lock( mtx );
   auto                          it = cache.find( { dev, level } );
   if ( it == cache.end() )
   {
      const int N = ( 1 << level ) + 1, n = N - 1;
      const int tmin = 6, tmax = 1 + 2 + 3 * ( n - 3 );
      SorPlaneTable tab;
      tab.nplanes = tmax >= tmin ? tmax - tmin + 1 : 0;
      std::vector< std::vector< SorPlanePoint > > byT( (size_t) std::max( tab.nplanes, 0 ) );
      for ( int z = 1; z <= n - 3; ++z )
         for ( int y = 1; y <= n - 2 - z; ++y )
            for ( int x = 1; x <= n - 1 - y - z; ++x )
               byT[(size_t) ( x + 2 * y + 3 * z - tmin )].push_back(
                   SorPlanePoint{ (int) ( slice_start( N, z ) + row_start( N - z, y ) + x ), (short) ( N - z ), (short) y } );
      std::vector< SorPlanePoint > flat;
      std::vector< int >           off( 1, 0 );
      for ( const auto& v : byT )
      {
         flat.insert( flat.end(), v.begin(), v.end() );
         off.push_back( (int) flat.size() );
      }
      void *pp = nullptr, *po = nullptr;
      HH_CHECK_HIP( hipMalloc( &pp, std::max< size_t >( 1, flat.size() ) * sizeof( SorPlanePoint ) ) );
      HH_CHECK_HIP( hipMalloc( &po, off.size() * sizeof( int ) ) );
      if ( !flat.empty() )
         HH_CHECK_HIP( hipMemcpy( pp, flat.data(), flat.size() * sizeof( SorPlanePoint ), hipMemcpyHostToDevice ) );
      HH_CHECK_HIP( hipMemcpy( po, off.data(), off.size() * sizeof( int ), hipMemcpyHostToDevice ) );
      tab.pts = static_cast< const SorPlanePoint* >( pp ), tab.off = static_cast< const int* >( po );
      it = cache.emplace( std::make_pair( dev, level ), tab ).first;
   }
   *out = &it->second;
   return HYTEG_HIP_OK;
}
struct SorSmallArgs
{
   double*              u[HYTEG_HIP_MAX_BATCH];
   const double*        rhs[HYTEG_HIP_MAX_BATCH];
   const double*        stencils; // device table [cell][15][15], or null: the one stencil in `w` (single-cell entry point)
   const SorPlanePoint* pts;
   const int*           off;
   int                  N, size, backwards, nplanes;
   double               relax, one_minus_relax;
   double               w[15];
};
__global__ __launch_bounds__( kSmallThreads ) void p1_sor_small_kernel( const SorSmallArgs A )
{
   extern __shared__ double lu[];
   const int                cell = blockIdx.x;
   double*                  ug   = A.u[cell];
   const double*            rhs  = A.rhs[cell];
   const double*            w    = A.stencils ? A.stencils + (size_t) cell * 225 + 14 * 15 : A.w;
   const double             invc = 1.0 / w[7];
   for ( int i = threadIdx.x; i < A.size; i += kSmallThreads )
      lu[i] = ug[i];
   __syncthreads();
   for ( int k = 0; k < A.nplanes; ++k )
   {
      const int t = A.backwards ? A.nplanes - 1 - k : k;
      for ( int c = A.off[t] + (int) threadIdx.x; c < A.off[t + 1]; c += kSmallThreads )
      {
         const SorPlanePoint P = A.pts[c];
         const int i = P.i, W = P.W, y = P.y;
         const int R = W - y, S0 = tri( W ), Sm = tri( W + 1 );
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
      const SorPlaneTable* planes = nullptr;
      const int            prc    = get_sor_planes( level, &planes );
      if ( prc != HYTEG_HIP_OK )
         return prc;
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
         A.pts = planes->pts, A.off = planes->off, A.nplanes = planes->nplanes;
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
   const SorBlockTable* tab = nullptr;
   int                  rc  = get_sor_blocks( level, &tab );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   int k = 0;
   while ( k < m )
   {
      int e = k;
      while ( e + 1 < m && sel[e + 1] == sel[e] + 1 )
         ++e;
      SorBlockBatchArgs B{};
      for ( int j = k; j <= e; ++j )
         B.u[j - k] = u[sel[j]], B.rhs[j - k] = rhs[sel[j]];
      B.stencils = stencils_dev + (size_t) sel[k] * 225;
      B.N = N, B.backwards = backwards ? 1 : 0, B.relax = relax, B.one_minus_relax = 1.0 + ( -relax );
      const int nw = (int) tab->wavefrontStart.size() - 1;
      for ( int q = 0; q < nw; ++q )
      {
         const int wv = backwards ? nw - 1 - q : q;
         const int lo = tab->wavefrontStart[wv], hi = tab->wavefrontStart[wv + 1];
         B.blocks     = tab->dev + lo;
         hipLaunchKernelGGL( p1_sor_block_batch_kernel, dim3( hi - lo, e - k + 1 ), dim3( kB * kB ), 0, as_stream( stream ), B );
      }
      k = e + 1;
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_sor_cell_sweeps( double*            u,
                                                const double*      rhs,
                                                int                level,
                                                const double*      w,
                                                double             relax,
                                                int                backwards,
                                                int                nsweeps,
                                                hyteg_hip_stream_t stream )
{
   HH_REQUIRE( u && rhs && w, "p1_sor_cell_sweeps: null pointer" );
   HH_REQUIRE( level_ok( level ), "p1_sor_cell_sweeps: level out of range [2,11]" );
   HH_REQUIRE( u != rhs, "p1_sor_cell_sweeps: u and rhs must not alias" );
   HH_REQUIRE( w[7] != 0.0, "p1_sor_cell_sweeps: zero centre weight" );
   HH_REQUIRE( nsweeps >= 0 && nsweeps <= 64, "p1_sor_cell_sweeps: nsweeps must be 0..64" );
   static const bool pipeline = [] {
      const char* e = std::getenv( "HYTEG_HIP_SOR_PIPELINE" ); // 0: one sweep after the other
      return !( e && e[0] == '0' );
   }();
   const int algo = g_sorAlgorithm.load( std::memory_order_relaxed );
   if ( nsweeps <= 1 || !pipeline || level < 5 || use_dataflow( level ) || algo == HYTEG_HIP_SOR_PLANES )
   {
      for ( int k = 0; k < nsweeps; ++k )
      {
         const int rc = hyteg_hip_p1_sor_cell( u, rhs, level, w, relax, backwards, stream );
         if ( rc != HYTEG_HIP_OK )
            return rc;
      }
      return HYTEG_HIP_OK;
   }
   const SorPipelineTable* tab = nullptr;
   int                     rc  = get_sor_pipeline( level, nsweeps, backwards != 0, &tab );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   SorBlockArgs B;
   B.u               = u;
   B.rhs             = rhs;
   B.N               = ( 1 << level ) + 1;
   B.backwards       = backwards ? 1 : 0;
   B.relax           = relax;
   B.one_minus_relax = 1.0 + ( -relax );
   B.invc            = 1.0 / w[7];
   for ( int k = 0; k < 15; ++k )
      B.st.w[k] = w[k];
   const int nsteps = (int) tab->stepStart.size() - 1;
   for ( int k = 0; k < nsteps; ++k )
   {
      const int lo = tab->stepStart[k], hi = tab->stepStart[k + 1];
      B.blocks     = tab->dev + lo;
      hipLaunchKernelGGL( p1_sor_block_kernel, dim3( hi - lo ), dim3( kB * kB ), 0, as_stream( stream ), B );
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
   static const int smallMax = [] {
      const char* e = std::getenv( "HYTEG_HIP_SOR_SMALL_MAX_LEVEL" ); // measurement switch
      return e ? std::atoi( e ) : 4; // level 5: 53 us here against 65 us blocked, but the pipelined sweeps (blocked) would no longer be
                                     // bit-identical to consecutive single sweeps there: the two forms sum a row in different orders
   }();
   if ( level <= smallMax && level <= 5 && g_sorAlgorithm.load( std::memory_order_relaxed ) == HYTEG_HIP_SOR_AUTO )
   {
      // the whole cell array fits into LDS: ONE workgroup runs all hyperplanes of the sweep in one launch (the kernel of the
      // batched entry point with one cell; same update order and summation order as the plane kernel).  Measured against one
      // launch per plane: 5 / 11 / 28 us instead of 5 / 33 / 125 us at levels 2 / 3 / 4 (round 1, scanning candidates; with
      // the plane table of round 3: 4.7 / 8.0 / 21 us, profiles/r03_sor_small_planes.txt).
      SorSmallArgs S{};
      const SorPlaneTable* planes = nullptr;
      const int            prc    = get_sor_planes( level, &planes );
      if ( prc != HYTEG_HIP_OK )
         return prc;
      S.u[0] = u, S.rhs[0] = rhs, S.stencils = nullptr;
      S.pts = planes->pts, S.off = planes->off, S.nplanes = planes->nplanes;
      S.N = A.N, S.size = (int) tet64( A.N ), S.backwards = backwards ? 1 : 0;
      S.relax = relax, S.one_minus_relax = A.one_minus_relax;
      for ( int k = 0; k < 15; ++k )
         S.w[k] = w[k];
      const size_t lds = (size_t) S.size * sizeof( double );
      if ( lds > 48 * 1024 )
         HH_CHECK_HIP( hipFuncSetAttribute( reinterpret_cast< const void* >( p1_sor_small_kernel ), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int) lds ) );
      hipLaunchKernelGGL( p1_sor_small_kernel, dim3( 1 ), dim3( kSmallThreads ), lds, as_stream( stream ), S );
      HH_CHECK_HIP( hipGetLastError() );
      return HYTEG_HIP_OK;
   }
   if ( level >= 5 && g_sorAlgorithm.load( std::memory_order_relaxed ) != HYTEG_HIP_SOR_PLANES )
   {
      // blocked sweep: one launch per block wavefront
      const SorBlockTable* tab = nullptr;
      int                  rc  = get_sor_blocks( level, &tab );
      if ( rc != HYTEG_HIP_OK )
         return rc;
      SorBlockArgs B;
      B.u               = u;
      B.rhs             = rhs;
      B.N               = A.N;
      B.backwards       = backwards ? 1 : 0;
      B.relax           = relax;
      B.one_minus_relax = A.one_minus_relax;
      B.invc            = A.invc;
      B.st              = A.st;
      const int nw      = (int) tab->wavefrontStart.size() - 1;
      for ( int k = 0; k < nw; ++k )
      {
         const int wv = backwards ? nw - 1 - k : k;
         const int lo = tab->wavefrontStart[wv], hi = tab->wavefrontStart[wv + 1];
         B.blocks     = tab->dev + lo;
         hipLaunchKernelGGL( p1_sor_block_kernel, dim3( hi - lo ), dim3( kB * kB ), 0, as_stream( stream ), B );
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
