// Dataflow form of the in-place SOR / Gauss-Seidel sweep on macro-cells (sor_3D_macrocell_P1.cpp:32-90 and
// sor_3D_macrocell_P1_backwards.cpp:52-57 of the reference): ONE launch per sweep.
//
// Geometry (see p1_sor.hip for the argument): in the skewed coordinates (p,q,r) = (x+y+z, y+z, z) every neighbour a
// point reads already-updated has all coordinates <= the point's, every other neighbour >=.  Any execution order
// that respects this componentwise order reproduces the reference's lexicographic (z,y,x) sweep.
//
// Decomposition: a COLUMN is the set of points with q in [8Q,8Q+8), r in [8R,8R+8), all p.  One wave (= one
// workgroup) owns a column: lane (ql,rl) owns the row (q,r) -- contiguous in memory along p -- and the wave marches
// along p, lane (ql,rl) staggered by ql+rl steps, so that in step S the lane updates p = S - ql - rl.  The current
// values of the column and of the rows around it live in an LDS ring indexed by (p+ql+rl) mod 32, which makes every
// neighbour access of a step a wave-uniform ring slot (S-3 .. S+3) plus a compile-time row offset.
//
// Columns depend on each other like the points do: (Q,R) reads updated values of (Q-1,R), (Q,R-1), (Q-1,R-1) and
// not-yet-updated values of (Q+1,R), (Q,R+1), (Q+1,R+1) (mirrored for the backward sweep).  Instead of one launch per
// block wavefront (p1_sor.hip), all columns run concurrently and hand their results over through global memory in
// chunks of 8 steps:
//   producer:  8 steps -> store the 8 x 64 new values -> wait for the stores -> publish progress[column]
//   consumer:  poll progress of its (<= 3) predecessors -> load their rows into the ring -> 8 steps ...
// A column starts its chunk c as soon as its predecessors have finished chunk c+1 (the stagger inside a tile is up
// to 14 steps), so the critical path is (3 * 2^level) steps plus one hand-over latency per column boundary crossed
// (2 * 2^level / 8), instead of (number of block wavefronts) x (3 x block edge) steps plus one launch each.
//
// Progress (no co-residency assumption): workgroups take their column from an atomic ticket; tickets run through the
// columns in an order in which every predecessor has a smaller ticket.  A wave that is running therefore only ever
// waits for waves that took their ticket before it, and the wave with the smallest unfinished ticket never waits.
// Every spin is bounded: a wave that exceeds the bound raises the abort word, which makes every wave leave
// (publishing "finished") -- the launch always drains; the host reports the abort at the next call.
//
// Measured on MI355X (profiles/r01_sor_dataflow_vs_blocks.txt): correct, but NOT faster than the blocked form -- per
// 8-step chunk a column spends ~2 us in the steps, ~2.7 us until its write-through stores are acknowledged and ~1.7 us in
// polls and halo loads, and a consumer runs two chunks behind its producer, so a sweep costs ~ (2 * 2^level / 8) x 15 us.
// The form is therefore opt-in (hyteg_hip_set_sor_algorithm( HYTEG_HIP_SOR_DATAFLOW )); AUTO keeps the blocked form.
//
// Coherence: the 8 XCDs have separate L2s.  Everything that crosses columns (the array values and the progress
// words) is accessed with agent-scope atomic loads / stores (sc1: coherent in device memory), the stores of a chunk
// are complete (s_waitcnt vmcnt(0), plus an agent-scope release fence unless HYTEG_DF_LIGHT_FENCE) before the progress
// word is published, and the rows of a predecessor are loaded only after its progress word has been observed.
#include <climits>
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "sor_dataflow.hpp"

#ifndef HYTEG_DF_LIGHT_FENCE
#define HYTEG_DF_LIGHT_FENCE 1
#endif

namespace hyteg_hip {
namespace {

constexpr int kT         = 8;   // tile edge: kT x kT rows per column = one wave
constexpr int kXS        = kT + 2;
constexpr int kRows      = kXS * kXS; // rows of the ring cross-section (tile + halo)
constexpr int kRing      = 32;  // ring slots (4 groups of 8)
constexpr int kRhsRing   = 16;
constexpr int kSpinLimit = 1 << 21;
constexpr int kCtrlAbort = 0, kCtrlTicket = 4, kCtrlFlags = 8;

struct DfColumn
{
   short Q, R;
};

struct DfArgs
{
   double*         u[HYTEG_HIP_MAX_BATCH];
   const double*   rhs[HYTEG_HIP_MAX_BATCH];
   const double*   stencils;  // batched form: device table [cell][15][15], row 14 = inner stencil; else nullptr
   const DfColumn* cols;      // all (Q,R), 0 <= R <= Q < nb, sorted by Q+R
   int*            ctrl;      // [0] abort, [4] ticket, [8 + cell*nb*nb + Q*nb + R] progress
   int*            hostAbort; // mapped host word, set when a spin bound was exceeded
   int             N, nb, ncols, ncells, backwards;
   double          relax, one_minus_relax;
   Stencil15       st;
};

__device__ inline double ld_co( const double* p ) { return __hip_atomic_load( p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); }
__device__ inline void   st_co( double* p, double v ) { __hip_atomic_store( p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); }
__device__ inline int    ld_flag( const int* p ) { return __hip_atomic_load( p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); }

struct DfWave
{
   double*       u;
   const double* rhs;
   double*       ring;
   double*       rring;
   int           N, n, Q, R, back, lane;
};

// Ring fills.  8 consecutive lanes read the 8 consecutive entries (64 B) of one row and one slot group; a fill is split
// into issuing the loads (values stay in registers while other work proceeds) and committing them to the ring.
//   upwind rows : the 18 halo rows whose values the predecessors produce (forward: q or r index -1, backward: 8)
//   other rows  : the remaining 9 x 9 rows (the tile and the downwind halo)
constexpr int kUpIts    = 3;  // 24 >= 18 rows
constexpr int kOtherIts = 11; // 88 >= 81 rows

__device__ inline bool df_up_row( const DfWave& W, int m, int& qh, int& rh )
{
   const int e = W.back ? kT : -1;
   if ( m < kXS )
   {
      qh = m - 1, rh = e;
      return qh != ( W.back ? -1 : kT ); // the corner rows (-1,8) / (8,-1) are never read
   }
   qh = e, rh = m - kXS;
   return m < kXS + kT;
}
__device__ inline bool df_other_row( const DfWave& W, int m, int& qh, int& rh )
{
   const int o = W.back ? -1 : 0;
   qh = m % ( kT + 1 ) + o, rh = m / ( kT + 1 ) + o;
   return m < ( kT + 1 ) * ( kT + 1 );
}

// Everything of a fill that depends only on the lane is computed once per column (a single wave issues one
// instruction every 4-5 cycles: index polynomials per chunk would cost more than the 8 steps themselves).
template < int ITS, bool UP >
struct DfFill
{
   double val[ITS];
   int    base[ITS];    // array index of p = 0 of the row (the row is contiguous in p)
   int    pmin[ITS];    // first p of the row inside the array (x >= 0); INT_MAX if the row is not loaded
   int    off[ITS];     // qh + rh: ring slot = p + off
   int    ringRow[ITS]; // row of the ring cross-section, -1: nothing to do

   __device__ inline void init( const DfWave& W )
   {
#pragma unroll
      for ( int it = 0; it < ITS; ++it )
      {
         const int  m = it * 8 + ( W.lane >> 3 );
         int        qh, rh;
         const bool doit = UP ? df_up_row( W, m, qh, rh ) : df_other_row( W, m, qh, rh );
         const int  q = kT * W.Q + qh, r = kT * W.R + rh;
         const bool rowIn = doit && r >= 0 && q >= r && q <= W.n;
         off[it]     = qh + rh;
         ringRow[it] = doit ? ( qh + 1 ) + kXS * ( rh + 1 ) : -1;
         pmin[it]    = rowIn ? q : INT_MAX;
         base[it]    = rowIn ? slice_start( W.N, r ) + row_start( W.N - r, q - r ) - q : 0;
      }
   }
   __device__ inline void issue( const DfWave& W, int g )
   {
      const int t = 8 * g + ( W.lane & 7 );
#pragma unroll
      for ( int it = 0; it < ITS; ++it )
      {
         const int p = t - off[it];
         val[it]     = ( p >= pmin[it] && p <= W.n ) ? ld_co( W.u + base[it] + p ) : 0.0;
      }
   }
   __device__ inline void commit( const DfWave& W, int g ) const
   {
      const int slot = ( ( 8 * g + ( W.lane & 7 ) ) & ( kRing - 1 ) ) * kRows;
#pragma unroll
      for ( int it = 0; it < ITS; ++it )
         if ( ringRow[it] >= 0 )
            W.ring[slot + ringRow[it]] = val[it];
   }
};
using DfFillUp    = DfFill< kUpIts, true >;
using DfFillOther = DfFill< kOtherIts, false >;

// rhs loads and result stores of the 64 tile rows, same lane mapping
struct DfTileRows
{
   double val[kT];
   int    base[kT], pmin[kT], off[kT];

   __device__ inline void init( const DfWave& W )
   {
#pragma unroll
      for ( int it = 0; it < kT; ++it )
      {
         const int  ri = it * 8 + ( W.lane >> 3 );
         const int  ql = ri & 7, rl = ri >> 3;
         const int  q = kT * W.Q + ql, r = kT * W.R + rl;
         const bool rowOk = r >= 1 && q >= r + 1 && q <= W.n - 2;
         off[it]  = ql + rl;
         pmin[it] = rowOk ? q + 1 : INT_MAX;
         base[it] = rowOk ? slice_start( W.N, r ) + row_start( W.N - r, q - r ) - q : 0;
      }
   }
   __device__ inline void issueRhs( const DfWave& W, int g )
   {
      const int t = 8 * g + ( W.lane & 7 );
#pragma unroll
      for ( int it = 0; it < kT; ++it )
      {
         const int p = t - off[it];
         val[it]     = ( p >= pmin[it] && p <= W.n - 1 ) ? W.rhs[base[it] + p] : 0.0;
      }
   }
   __device__ inline void commitRhs( const DfWave& W, int g ) const
   {
      const int slot = ( ( 8 * g + ( W.lane & 7 ) ) & ( kRhsRing - 1 ) ) * 64 + ( W.lane >> 3 );
#pragma unroll
      for ( int it = 0; it < kT; ++it )
         W.rring[slot + it * 8] = val[it];
   }
   // stores the values the 8 steps of slot group g produced
   __device__ inline void store( const DfWave& W, int g ) const
   {
      const int t    = 8 * g + ( W.lane & 7 );
      const int slot = ( t & ( kRing - 1 ) ) * kRows + ( W.lane >> 3 ) % 8 + 1;
#pragma unroll
      for ( int it = 0; it < kT; ++it )
      {
         // ri = it * 8 + (lane >> 3):  ql = lane >> 3,  rl = it
         const int p = t - off[it];
         if ( p >= pmin[it] && p <= W.n - 1 )
            st_co( W.u + base[it] + p, W.ring[slot + kXS * ( it + 1 )] );
      }
   }
};

template < bool BATCH >
__global__ __launch_bounds__( 64 ) void p1_sor_dataflow_kernel( const DfArgs A )
{
   __shared__ double ring[kRing * kRows];
   __shared__ double rring[kRhsRing * 64];
   const int         lane   = threadIdx.x;
   int               ticket = 0;
   if ( lane == 0 )
      ticket = atomicAdd( A.ctrl + kCtrlTicket, 1 );
   ticket = __builtin_amdgcn_readfirstlane( ticket );

   const int      cell = BATCH ? ticket % A.ncells : 0;
   const int      ci   = BATCH ? ticket / A.ncells : ticket;
   const int      nb   = A.nb;
   const DfColumn cq   = A.cols[A.backwards ? A.ncols - 1 - ci : ci];
   const int      Q = cq.Q, R = cq.R;
   int*           flags  = A.ctrl + kCtrlFlags + cell * nb * nb;
   int*           myflag = flags + Q * nb + R;
   const int      dir    = A.backwards ? -1 : 1;

   // predecessors in sweep direction: (Qp,R), (Q,Rp), (Qp,Rp)
   const int  Qp = Q - dir, Rp = R - dir;
   const bool hasA = Qp >= R && Qp < nb;
   const bool hasB = Rp >= 0 && Rp <= Q;
   const bool hasC = Rp >= 0 && Qp >= Rp && Qp < nb;
   const int* flagA = flags + Qp * nb + R;
   const int* flagB = flags + Q * nb + Rp;
   const int* flagC = flags + Qp * nb + Rp;

   double w[15];
   if constexpr ( BATCH )
   {
      const double* ws = A.stencils + (size_t) cell * 225 + 14 * 15;
#pragma unroll
      for ( int k = 0; k < 15; ++k )
         w[k] = ws[k];
   }
   else
   {
#pragma unroll
      for ( int k = 0; k < 15; ++k )
         w[k] = A.st.w[k];
   }
   const double scale = A.relax * ( 1.0 / w[7] );

   DfWave W;
   W.u = A.u[BATCH ? cell : 0], W.rhs = A.rhs[BATCH ? cell : 0];
   W.ring = ring, W.rring = rring;
   W.N = A.N, W.n = A.N - 1, W.Q = Q, W.R = R, W.back = A.backwards, W.lane = lane;

   const int  ql = lane & 7, rl = lane >> 3;
   const int  q = kT * Q + ql, r = kT * R + rl;
   const bool rowOk = r >= 1 && q >= r + 1 && q <= W.n - 2;
   const int  myrow = ( ql + 1 ) + kXS * ( rl + 1 );

   // chunks of this column: slot groups c = Q .. nb+1 (steps S = p + ql + rl, p in [q+1, n-1]); numbered in sweep
   // direction by ct (the same numbering in every column): forward ct = c, backward ct = ctop - c
   const int ctop = nb + 1;
   const int nch  = nb + 2 - Q;
   bool      ok   = true;

   // progress words of the predecessors, as last seen (a predecessor that does not exist counts as finished); the
   // three words are polled together, one memory round trip per poll
   int  seenA = hasA ? 0 : INT_MAX, seenB = hasB ? 0 : INT_MAX, seenC = hasC ? 0 : INT_MAX;
   auto waitPreds = [&]( int ct ) -> bool {
      const int needAB = 8 * ct + 16, needC = 8 * ct + 24;
      int       it     = 0;
      while ( seenA < needAB || seenB < needAB || seenC < needC )
      {
         const int a = hasA ? ld_flag( flagA ) : INT_MAX;
         const int b = hasB ? ld_flag( flagB ) : INT_MAX;
         const int c = hasC ? ld_flag( flagC ) : INT_MAX;
         seenA = a, seenB = b, seenC = c;
         if ( ( ++it & 31 ) == 0 )
         {
            if ( ld_flag( A.ctrl + kCtrlAbort ) != 0 )
               return false;
            if ( it > kSpinLimit )
            {
               __hip_atomic_store( A.ctrl + kCtrlAbort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
               __hip_atomic_store( A.hostAbort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
               return false;
            }
         }
      }
#if !HYTEG_DF_LIGHT_FENCE
      __builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "agent" );
#endif
      return true;
   };

   DfFillUp    fu;
   DfFillOther fo;
   DfTileRows  fr;
   fu.init( W ), fo.init( W ), fr.init( W );
   {
      // prologue: the two slot groups behind the first chunk and the group of the chunk itself
      const int c0 = A.backwards ? ctop : Q;
      fo.issue( W, c0 - dir );
      fr.issueRhs( W, c0 );
      fo.commit( W, c0 - dir );
      fo.issue( W, c0 );
      fo.commit( W, c0 );
      fo.issue( W, c0 + dir ); // stays in flight: committed at the top of the first chunk
      ok = waitPreds( A.backwards ? 0 : Q );
      if ( ok )
      {
         fu.issue( W, c0 - dir );
         fu.commit( W, c0 - dir );
      }
   }

   for ( int k = 0; k < nch && ok; ++k )
   {
      const int c  = A.backwards ? ctop - k : Q + k; // slot group of this chunk
      const int ct = A.backwards ? k : Q + k;
      ok           = waitPreds( ct );
      if ( !ok )
         break;
      fu.issue( W, c );
      fo.commit( W, c + dir ); // the other rows' group c + dir and the rhs of group c were issued one chunk ago
      fr.commitRhs( W, c );
      fu.commit( W, c );
      if ( k + 1 < nch )
      {
         // prefetch for the next chunk; in flight during the 8 steps below
         fo.issue( W, c + 2 * dir );
         fr.issueRhs( W, c + dir );
      }
      __syncthreads(); // one wave per workgroup: orders the LDS writes above before the reads below

#pragma unroll 1
      for ( int j = 0; j < 8; ++j )
      {
         const int  S      = 8 * c + ( A.backwards ? 7 - j : j );
         const int  p      = S - ql - rl;
         const bool active = rowOk && p >= q + 1 && p <= W.n - 1;
         // ring slot bases of S-3 .. S+3 (wave-uniform)
         const double* b0 = ring + ( ( S ) & ( kRing - 1 ) ) * kRows + myrow;
         const double* bm1 = ring + ( ( S - 1 ) & ( kRing - 1 ) ) * kRows + myrow;
         const double* bm2 = ring + ( ( S - 2 ) & ( kRing - 1 ) ) * kRows + myrow;
         const double* bm3 = ring + ( ( S - 3 ) & ( kRing - 1 ) ) * kRows + myrow;
         const double* bp1 = ring + ( ( S + 1 ) & ( kRing - 1 ) ) * kRows + myrow;
         const double* bp2 = ring + ( ( S + 2 ) & ( kRing - 1 ) ) * kRows + myrow;
         const double* bp3 = ring + ( ( S + 3 ) & ( kRing - 1 ) ) * kRows + myrow;
         // neighbour (dp,dq,dr): slot S+dp+dq+dr, row offset dq + kXS dr; weights in the order of the reference's map:
         // 0 BC 1 BE 2 BNW 3 BN 4 S 5 SE 6 W 7 C 8 E 9 NW 10 N 11 TS 12 TSE 13 TW 14 TC  (skewed offsets: p1_sor.hip)
         double a0 = -w[3] * bm1[-kXS];                    // BN  ( 0, 0,-1)
         double a1 = -w[10] * bp2[1];                      // N   ( 1, 1, 0)
         double a2 = -w[5] * bm1[-1];                      // SE  ( 0,-1, 0)
         a0        = fma( -w[12], bp2[kXS], a0 );          // TSE ( 1, 0, 1)
         a1        = fma( -w[1], bm2[-1 - kXS], a1 );      // BE  ( 0,-1,-1)
         a2        = fma( -w[8], bp1[0], a2 );             // E   ( 1, 0, 0)
         a0        = fma( -w[6], bm1[0], a0 );             // W   (-1, 0, 0)
         a1        = fma( -w[13], bp2[1 + kXS], a1 );      // TW  ( 0, 1, 1)
         a2        = fma( -w[2], bm2[-kXS], a2 );          // BNW (-1, 0,-1)
         a0        = fma( -w[9], bp1[1], a0 );             // NW  ( 0, 1, 0)
         a1        = fma( -w[4], bm2[-1], a1 );            // S   (-1,-1, 0)
         a2        = fma( -w[11], bp1[kXS], a2 );          // TS  ( 0, 0, 1)
         a0        = fma( -w[0], bm3[-1 - kXS], a0 );      // BC  (-1,-1,-1)
         a1        = fma( -w[14], bp3[1 + kXS], a1 );      // TC  ( 1, 1, 1)
         a2        = a2 + rring[( S & ( kRhsRing - 1 ) ) * 64 + lane];
         const double acc = ( a0 + a1 ) + a2;
         const double nv  = scale * acc + A.one_minus_relax * b0[0];
         if ( active )
            ring[( S & ( kRing - 1 ) ) * kRows + myrow] = nv;
         __syncthreads();
      }

      fr.store( W, c );
#if HYTEG_DF_LIGHT_FENCE
      asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
#else
      __builtin_amdgcn_fence( __ATOMIC_RELEASE, "agent" );
#endif
      if ( lane == 0 )
         __hip_atomic_store( myflag, 8 * ct + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
   }
   if ( lane == 0 )
      __hip_atomic_store( myflag, INT_MAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
}

// ---- host side ------------------------------------------------------------------------------------------
struct DfColumnTable
{
   const DfColumn* dev   = nullptr;
   int             count = 0;
};
struct DfControl
{
   int*   ctrl         = nullptr;
   size_t capInts      = 0;
   int*   hostAbort    = nullptr; // mapped pinned host word
   int*   hostAbortDev = nullptr;
};

std::mutex                                        g_mtx;
std::map< std::pair< int, int >, DfColumnTable >   g_cols;    // (device, nb)
std::map< std::pair< int, hipStream_t >, DfControl > g_control; // (device, stream)

int get_columns( int dev, int nb, DfColumnTable* out )
{
   auto it = g_cols.find( { dev, nb } );
   if ( it == g_cols.end() )
   {
      std::vector< DfColumn > cols;
      for ( int t = 0; t <= 2 * ( nb - 1 ); ++t )
         for ( int R = 0; R < nb; ++R )
         {
            const int Q = t - R;
            if ( Q >= R && Q < nb )
               cols.push_back( DfColumn{ (short) Q, (short) R } );
         }
      void* p = nullptr;
      HH_CHECK_HIP( hipMalloc( &p, cols.size() * sizeof( DfColumn ) ) );
      HH_CHECK_HIP( hipMemcpy( p, cols.data(), cols.size() * sizeof( DfColumn ), hipMemcpyHostToDevice ) );
      DfColumnTable tab;
      tab.dev = static_cast< const DfColumn* >( p ), tab.count = (int) cols.size();
      it = g_cols.emplace( std::make_pair( dev, nb ), tab ).first;
   }
   *out = it->second;
   return HYTEG_HIP_OK;
}

int get_control( int dev, hipStream_t stream, size_t ints, DfControl** out )
{
   DfControl& C = g_control[{ dev, stream }];
   if ( !C.hostAbort )
   {
      void* h = nullptr;
      HH_CHECK_HIP( hipHostMalloc( &h, sizeof( int ), hipHostMallocMapped ) );
      C.hostAbort  = static_cast< int* >( h );
      *C.hostAbort = 0;
      void* d      = nullptr;
      HH_CHECK_HIP( hipHostGetDevicePointer( &d, h, 0 ) );
      C.hostAbortDev = static_cast< int* >( d );
   }
   if ( C.capInts < ints )
   {
      // a launch on this stream may still use the old buffer: it is released only after the stream has drained
      if ( C.ctrl )
      {
         HH_CHECK_HIP( hipStreamSynchronize( stream ) );
         HH_CHECK_HIP( hipFree( C.ctrl ) );
         C.ctrl = nullptr, C.capInts = 0;
      }
      void* p = nullptr;
      HH_CHECK_HIP( hipMalloc( &p, ints * sizeof( int ) ) );
      HH_CHECK_HIP( hipMemset( p, 0, ints * sizeof( int ) ) );
      C.ctrl = static_cast< int* >( p ), C.capInts = ints;
   }
   *out = &C;
   return HYTEG_HIP_OK;
}

} // namespace

int launch_sor_dataflow( int                  ncells,
                         double* const*       u,
                         const double* const* rhs,
                         int                  level,
                         const double*        stencils_dev,
                         const double*        w,
                         double               relax,
                         int                  backwards,
                         hipStream_t          stream )
{
   HH_REQUIRE( level >= kSorDataflowMinLevel && level <= HYTEG_HIP_MAX_LEVEL, "sor dataflow: level out of range" );
   HH_REQUIRE( ncells >= 1 && ncells <= HYTEG_HIP_MAX_BATCH, "sor dataflow: ncells out of range" );
   HH_REQUIRE( ( stencils_dev != nullptr ) != ( w != nullptr ), "sor dataflow: exactly one of the stencil arguments" );
   int dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   const int N = ( 1 << level ) + 1, nb = ( N - 1 ) / kT;
   std::lock_guard< std::mutex > lock( g_mtx );
   DfColumnTable                 cols;
   int                           rc = get_columns( dev, nb, &cols );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   DfControl*   C    = nullptr;
   const size_t ints = (size_t) kCtrlFlags + (size_t) ncells * nb * nb;
   rc                = get_control( dev, stream, ints, &C );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   if ( *static_cast< volatile int* >( C->hostAbort ) != 0 )
      return fail( HYTEG_HIP_ELAUNCH, "sor dataflow: an earlier sweep on this stream exceeded its spin bound and was abandoned" );

   DfArgs A{};
   for ( int c = 0; c < ncells; ++c )
      A.u[c] = u[c], A.rhs[c] = rhs[c];
   A.stencils = stencils_dev, A.cols = cols.dev, A.ctrl = C->ctrl, A.hostAbort = C->hostAbortDev;
   A.N = N, A.nb = nb, A.ncols = cols.count, A.ncells = ncells, A.backwards = backwards ? 1 : 0;
   A.relax = relax, A.one_minus_relax = 1.0 + ( -relax );
   if ( w )
      for ( int k = 0; k < 15; ++k )
         A.st.w[k] = w[k];
   // ticket and progress words back to "not started" (the abort word at [0] is sticky)
   HH_CHECK_HIP( hipMemsetAsync( C->ctrl + kCtrlTicket, 0, ( ints - kCtrlTicket ) * sizeof( int ), stream ) );
   if ( stencils_dev )
      hipLaunchKernelGGL( p1_sor_dataflow_kernel< true >, dim3( cols.count * ncells ), dim3( 64 ), 0, stream, A );
   else
      hipLaunchKernelGGL( p1_sor_dataflow_kernel< false >, dim3( cols.count ), dim3( 64 ), 0, stream, A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

} // namespace hyteg_hip
