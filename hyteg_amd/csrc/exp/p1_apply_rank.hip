// NEGATIVE RESULT (round 2), kept so that the measurement can be repeated: NOT part of libhyteg_hip.so.  To rebuild the
// experiment add this file to HIP_SOURCES in __graft_entry__.py, declare hyteg_hip_p1_apply_cell_rank in include/hyteg_hip.h and
// hyteg_amd/capi.py (signature below) and run exp/rank_kernel_probe.py.  MI355X, level 8, loop-back peers, one process
// (microseconds per apply until the device has drained; interior kernel alone 9.4):
//                                                               1 shared face   3 shared faces
//   four launches: shares, p2p pack, interior, wait + reduce         25.3            36.1      <- what the host layer does
//   this kernel + wait + reduce                                      43.0            77.8
//   this kernel without the pack workgroups' wait for the shares     20.5 (wrong results)
//   shares + pack in one launch (no bricks), interior, reduce        40.2            69.8
//   the same without the wait                                        28.5            35.0 (wrong results)
// The dependency inside a launch (pack workgroups polling a counter the share workgroups increment, values handed over through
// memory with agent-scope stores / loads because the XCDs do not share an L2) costs 12-30 us -- more than the two launches it
// saves -- and under the interior kernel's HBM load every link of the shares -> counter -> poll -> gather -> store chain is a
// loaded-latency round trip.  Results were bit-identical to the four-launch path (tests/test_gpu_distributed.py ran with it).
//
// One launch for everything a rank's macro-cell contributes to an operator application before the reduce kernel:
//   P1Operator::apply (src/hyteg/p1functionspace/P1Operator.hpp:192-320) of a storage distributed over several ranks is
//   boundary shares -> pack -> [exchange] -> interior stencil -> reduce.  Four launches cost a rank ~24 us at level 8 where the
//   interior kernel alone takes 9.9 (Multi-GPU status).  Here the first three share ONE grid:
//     workgroups [0, B)          the z-march bricks of the interior (kernels_apply_zmarch.hpp, unchanged; they come first and
//                                keep their XCD-aware mapping),
//     workgroups [B, B + S)      this cell's shares at its shared shell points (shell.hpp), written through to memory
//                                (agent-scope stores: the 8 XCDs do not share an L2) and reported in a counter,
//     workgroups [B + S, ...)    the pack of the peer-to-peer exchange (p2p_device.hpp): they wait for the S share workgroups
//                                -- which have smaller indices, i.e. were dispatched before them and wait for nobody --,
//                                gather the shares with agent-scope loads, store them into the peers' arenas and publish
//                                the sequence number.
//   The values travel while the bricks are still streaming; the reduce kernel (hyteg_hip_reduce_shared_after_p2p) follows.
// Shares and interior are computed by the same code as in their own launches: results are bit-identical to the four-launch path.
#include <cstdlib>

#include "../kernels_apply_zmarch.hpp"
#include "../p2p_device.hpp"
#include "../shell.hpp"

using namespace hyteg_hip;

namespace {

constexpr int kBrickNY = 4; // as in p1_apply.hip
inline int   brick_lz( int level ) { return level >= 8 ? 8 : 4; }
constexpr int kRankThreads = 64 * kZMarchWavesPerBlock;

struct RankArgs
{
   shell::Slots14x15           S;
   unsigned                    mask;
   int                         update;
   int                         brickBlocks, shellBlocks;
   const hyteg_hip_p2p_peer_t* peers;
   int                         npeers;
   double* const*              bases;
   const int*                  entry_buf;
   const int*                  entry_off;
   int                         n;
   unsigned long long          seq;
   unsigned*                   counters; // [0] pack workgroups done, [1] share workgroups done; both zero between launches
   unsigned*                   status;
   unsigned long long          timeout_ticks;
   int                         dbg;
};

template < int MODE, int LZ >
__global__ __launch_bounds__( kRankThreads ) void p1_apply_rank_kernel( const BrickTask* tasks, int ntasks, int xcd_chunk, const ZMarchArgs A,
                                                                        const RankArgs R )
{
   const int b = blockIdx.x;
   if ( b < R.brickBlocks )
   {
      zmarch_body< MODE, kBrickNY, LZ, MODE == APPLY_ADD ? 2 : 0, false, 2, double >( A, tasks, ntasks, xcd_chunk );
      return;
   }
   double*       dst = static_cast< double* >( A.dst );
   const double* src = static_cast< const double* >( A.src );
   if ( b < R.brickBlocks + R.shellBlocks )
   {
      const int q = ( b - R.brickBlocks ) * kRankThreads + (int) threadIdx.x;
      int       x, y, z, slot;
      if ( !( R.dbg & 4 ) && shell::shell_point( A.N, q, x, y, z, slot ) && ( ( R.mask >> slot ) & 1u ) )
      {
         const double acc = shell::share( R.S, src, A.N, x, y, z, slot );
         const int    i   = cell_index( A.N, x, y, z );
         if ( R.dbg & 2 )
            dst[i] = acc;
         else
            __hip_atomic_store( dst + i, R.update == HYTEG_HIP_ADD ? acc + dst[i] : acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      }
      p2p::stores_acknowledged();
      __syncthreads();
      if ( threadIdx.x == 0 )
         __hip_atomic_fetch_add( R.counters + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      return;
   }
   // pack: wait for the shares (bounded, like every wait of the transport)
   if ( threadIdx.x == 0 && !( R.dbg & 1 ) )
   {
      const unsigned long long t0 = wall_clock64();
      while ( __hip_atomic_load( R.counters + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) < (unsigned) R.shellBlocks )
      {
         if ( wall_clock64() - t0 > R.timeout_ticks )
         {
            __hip_atomic_store( R.status, 0x40000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
            break;
         }
         __builtin_amdgcn_s_sleep( 4 );
      }
   }
   __syncthreads();
   const int first = R.brickBlocks + R.shellBlocks;
   const int k     = ( b - first ) * kRankThreads + (int) threadIdx.x;
   if ( k < R.n && !( R.dbg & 8 ) )
      p2p::send_value( R.peers, R.npeers, k, R.seq,
                       __hip_atomic_load( R.bases[R.entry_buf[k]] + R.entry_off[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) );
   p2p::stores_acknowledged();
   __syncthreads();
   if ( threadIdx.x == 0 )
   {
      const unsigned done = __hip_atomic_fetch_add( R.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      if ( done == gridDim.x - first - 1 )
      {
         // every pack workgroup has passed its wait: both counters are free for the next launch
         __hip_atomic_store( R.counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
         __hip_atomic_store( R.counters + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
         p2p::publish( R.peers, R.npeers, R.seq );
      }
   }
}

template < int MODE, int LZ >
int launch_rank( double* dst, const double* src, int level, const double* w, RankArgs& R, hipStream_t stream )
{
   BrickTable bt;
   int        rc = get_bricks( level, kBrickNY, LZ, &bt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   ZMarchArgs A{};
   A.dst    = dst;
   A.src    = src;
   A.tasks  = bt.dev;
   A.ntasks = bt.count;
   A.N      = ( 1 << level ) + 1;
   A.bytes  = (unsigned) ( tet64( A.N ) * (int64_t) sizeof( double ) );
   for ( int k = 0; k < 15; ++k )
      A.st.w[k] = w[k];
   for ( int k = 0; k < kZMarchMaxZChunks; ++k )
      A.zs[k] = bt.zs[k];
   int nblocks   = ( bt.count + kZMarchWavesPerBlock - 1 ) / kZMarchWavesPerBlock;
   nblocks       = ( nblocks + 7 ) & ~7;
   if ( R.dbg & 16 ) // shares + pack only; the caller launches the interior kernel
      nblocks = 0;
   A.xcd_chunk   = nblocks / 8;
   R.brickBlocks = nblocks;
   R.shellBlocks = ( 4 * tri( A.N ) + kRankThreads - 1 ) / kRankThreads;
   const int packBlocks = R.n > 0 ? ( R.n + kRankThreads - 1 ) / kRankThreads : 1; // an empty message still signals
   hipLaunchKernelGGL( ( p1_apply_rank_kernel< MODE, LZ > ), dim3( nblocks + R.shellBlocks + packBlocks ), dim3( kRankThreads ), 0, stream, A.tasks,
                       A.ntasks, A.xcd_chunk, A, R );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

} // namespace

extern "C" {

HYTEG_HIP_API int hyteg_hip_p1_apply_cell_rank( double*                     dst,
                                                const double*               src,
                                                int                         level,
                                                const double*               w,
                                                const double*               w_slots,
                                                unsigned                    mask,
                                                int                         update,
                                                const hyteg_hip_p2p_peer_t* peers,
                                                int                         npeers,
                                                double* const*              bases,
                                                const int*                  entry_buf,
                                                const int*                  entry_off,
                                                int                         n,
                                                unsigned long long          seq,
                                                unsigned*                   counters,
                                                unsigned*                   status,
                                                unsigned                    timeout_ms,
                                                hyteg_hip_stream_t          stream )
{
   HH_REQUIRE( dst && src && w && w_slots, "p1_apply_cell_rank: null pointer" );
   HH_REQUIRE( dst != src, "p1_apply_cell_rank: src and dst must not alias" );
   HH_REQUIRE( level >= HYTEG_HIP_MIN_LEVEL && level <= 10, "p1_apply_cell_rank: level out of range [2,10]" );
   // Add with shared points goes through a temporary in the host layer (shares are summed over cells before they are added)
   HH_REQUIRE( update == HYTEG_HIP_REPLACE, "p1_apply_cell_rank: Replace only" );
   HH_REQUIRE( ( mask & HYTEG_HIP_MASK_INNER ) && ( mask & HYTEG_HIP_MASK_SHELL ), "p1_apply_cell_rank: needs inner and shared points" );
   HH_REQUIRE( npeers > 0 && peers && counters && status && seq > 0 && n >= 0, "p1_apply_cell_rank: bad exchange arguments" );
   HH_REQUIRE( n == 0 || ( bases && entry_buf && entry_off ), "p1_apply_cell_rank: null pointer" );
   RankArgs R{};
   for ( int s = 0; s < 14; ++s )
      for ( int k = 0; k < 15; ++k )
         R.S.w[s][k] = w_slots[15 * s + k];
   R.mask = mask & HYTEG_HIP_MASK_SHELL, R.update = update;
   R.peers = peers, R.npeers = npeers, R.bases = bases, R.entry_buf = entry_buf, R.entry_off = entry_off, R.n = n;
   R.seq = seq, R.counters = counters, R.status = status;
   R.timeout_ticks = (unsigned long long) ( timeout_ms ? timeout_ms : 20000u ) * 100000ull;
   if ( const char* e = std::getenv( "HYTEG_HIP_RANK_DBG" ) )
      R.dbg = std::atoi( e );
   hipStream_t s = as_stream( stream );
   return brick_lz( level ) == 8 ? launch_rank< APPLY_REPLACE, 8 >( dst, src, level, w, R, s ) : launch_rank< APPLY_REPLACE, 4 >( dst, src, level, w, R, s );
}
}
