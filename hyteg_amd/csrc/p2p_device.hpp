// Device side of the peer-to-peer exchange (see comm_p2p.hip for the protocol): how a value reaches a peer's receive slot and
// how the arrival is published.  Shared by the pack kernel of comm_p2p.hip and the one-launch rank apply (p1_apply_rank.hip).
#pragma once

#include "common.hpp"

namespace hyteg_hip {
namespace p2p {

// No release fence in the pack kernel: on gfx942/gfx950 a system- (or agent-) scope release is an L2 write-back
// (buffer_wbl2), and one per workgroup made the kernel last 31 us instead of 3.4 (exp/p2p_probe.py).  Instead the values are
// written with system-scope write-through stores (sc0 sc1: they do not stay in this GPU's L2, and the arena is uncached on
// the owner's side), a wave waits until its stores have been acknowledged (s_waitcnt vmcnt(0)) before its workgroup
// reports in, and the last workgroup -- which has observed every other workgroup's report -- writes the flag words.
__device__ __forceinline__ void store_through( double* p, double v )
{
   __hip_atomic_store( p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
}
__device__ __forceinline__ void stores_acknowledged()
{
   asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
}

// value k of the send enumeration goes to the peer whose segment [start, start + count) holds k (segments are concatenated
// per peer), slot [seq & 1]
__device__ __forceinline__ void send_value( const hyteg_hip_p2p_peer_t* __restrict__ peers, int npeers, int k, unsigned long long seq, double v )
{
   int p = 0;
   while ( p + 1 < npeers && k >= peers[p + 1].start )
      ++p;
   store_through( peers[p].slot[seq & 1ull] + ( k - peers[p].start ), v );
}
// called by ONE thread once every workgroup's stores have been acknowledged: the sequence number into every peer's flag word
__device__ __forceinline__ void publish( const hyteg_hip_p2p_peer_t* __restrict__ peers, int npeers, unsigned long long seq )
{
   for ( int p = 0; p < npeers; ++p )
      __hip_atomic_store( peers[p].flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
}

} // namespace p2p
} // namespace hyteg_hip
