// Peer-to-peer part of the C-ABI: the shared-point exchange without a library call per exchange.
//
// Same semantics as comm_rccl.hip (src/hyteg/communication/BufferedCommunication.cpp:181-470: pack, send to / receive from
// every neighbour rank, wait, unpack; payloads of src/hyteg/p1functionspace/VertexDoFAdditivePackInfo.hpp:676-745), other
// mechanism: every rank owns one ARENA of device memory that the other ranks of the node map through HIP IPC (xGMI
// peer access).  The pack kernel gathers the shared values and stores them straight into the receive slots inside the
// peers' arenas; its last workgroup then publishes a sequence number in every peer's flag word.  The receiving rank
// enqueues a one-wave wait kernel in front of its reduce kernel.  Per exchange that is two kernel launches on the compute
// stream and no host synchronisation, where ncclSend / ncclRecv cost ~25 us of host time per group (Multi-GPU).
//
// Visibility: arenas are UNCACHED device memory (hipDeviceMallocUncached), so neither the writer's nor the owner's L2
// holds lines of them; values and flags are written with system-scope write-through stores, the flag only after every
// wave's value stores have been acknowledged (see the pack kernel), and polled with system-scope loads.  Slot reuse: a receive
// segment has two slots used alternately (sequence parity).  A sender overwrites slot s of exchange k only in exchange
// k + 2, i.e. after its own reduce of exchange k + 1, which waited for the receiver's pack k + 1, which the receiver's
// stream ran after its reduce of exchange k -- exchanges between two ranks are symmetric, so no acknowledgement travels.
// Every poll loop is bounded (timeout -> status word, the launch drains, the host layer raises).
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "p2p_device.hpp"

namespace hyteg_hip {
namespace {

constexpr int kPackThreads = 256;

__global__ __launch_bounds__( kPackThreads ) void p2p_pack_kernel( const hyteg_hip_p2p_peer_t* __restrict__ peers,
                                                                   int                        npeers,
                                                                   double* const* __restrict__ bases,
                                                                   const int* __restrict__ entry_buf,
                                                                   const int* __restrict__ entry_off,
                                                                   int                n,
                                                                   unsigned long long seq,
                                                                   unsigned*          counter )
{
   const int k = blockIdx.x * kPackThreads + threadIdx.x;
   if ( k < n )
      p2p::send_value( peers, npeers, k, seq, bases[entry_buf[k]][entry_off[k]] );
   p2p::stores_acknowledged();
   __syncthreads();
   if ( threadIdx.x == 0 )
   {
      const unsigned done = __hip_atomic_fetch_add( counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      if ( done == gridDim.x - 1 )
      {
         __hip_atomic_store( counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); // ready for the next exchange
         p2p::publish( peers, npeers, seq );
      }
   }
}

// one wave: lane p polls the flag word of peer p, p + 64, ... until it has reached seq
__global__ __launch_bounds__( 64 ) void p2p_wait_kernel( const unsigned long long* flags,
                                                          int                       npeers,
                                                          int                       stride,
                                                          unsigned long long        seq,
                                                          unsigned*                 status,
                                                          unsigned long long        timeout_ticks )
{
   const unsigned long long t0 = wall_clock64();
   for ( int p = threadIdx.x; p < npeers; p += 64 )
   {
      const unsigned long long* f = flags + (size_t) p * stride;
      while ( __hip_atomic_load( f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM ) < seq )
      {
         if ( wall_clock64() - t0 > timeout_ticks )
         {
            __hip_atomic_store( status, 1u + (unsigned) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
            break;
         }
         __builtin_amdgcn_s_sleep( 8 );
      }
   }
   // acquire once (cache invalidate, no write-back); the reduce kernel that follows reads the uncached slots
   __builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "" );
}

} // namespace
} // namespace hyteg_hip

using namespace hyteg_hip;

extern "C" {

HYTEG_HIP_API int hyteg_hip_p2p_arena_create( size_t bytes, void** base, unsigned char* handle, int* kind )
{
   HH_REQUIRE( base && handle && bytes > 0, "p2p_arena_create: null pointer / empty arena" );
   static_assert( sizeof( hipIpcMemHandle_t ) == HYTEG_HIP_P2P_HANDLE_BYTES, "IPC handle size" );
   // HYTEG_HIP_P2P_ARENA = uncached (default) | finegrained | default -- the last two only to study the difference
   const char*  e     = std::getenv( "HYTEG_HIP_P2P_ARENA" );
   const std::string want = e ? e : "uncached";
   void*        p     = nullptr;
   int          k     = 0;
   if ( want == "uncached" )
      HH_CHECK_HIP( hipExtMallocWithFlags( &p, bytes, hipDeviceMallocUncached ) );
   else if ( want == "finegrained" )
   {
      HH_CHECK_HIP( hipExtMallocWithFlags( &p, bytes, hipDeviceMallocFinegrained ) );
      k = 1;
   }
   else if ( want == "default" )
   {
      HH_CHECK_HIP( hipMalloc( &p, bytes ) );
      k = 2;
   }
   else
      return fail( HYTEG_HIP_EINVAL, "p2p_arena_create: HYTEG_HIP_P2P_ARENA must be uncached, finegrained or default" );
   hipError_t err = hipMemset( p, 0, bytes );
   if ( err == hipSuccess )
      err = hipDeviceSynchronize();
   hipIpcMemHandle_t h;
   if ( err == hipSuccess )
      err = hipIpcGetMemHandle( &h, p );
   if ( err != hipSuccess )
   {
      (void) hipFree( p );
      return fail( HYTEG_HIP_ELAUNCH, std::string( "p2p_arena_create: " ) + hipGetErrorString( err ) );
   }
   std::memcpy( handle, &h, sizeof( h ) );
   *base = p;
   if ( kind )
      *kind = k;
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2p_arena_destroy( void* base )
{
   if ( base )
      HH_CHECK_HIP( hipFree( base ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2p_arena_open( const unsigned char* handle, void** mapped )
{
   HH_REQUIRE( handle && mapped, "p2p_arena_open: null pointer" );
   hipIpcMemHandle_t h;
   std::memcpy( &h, handle, sizeof( h ) );
   HH_CHECK_HIP( hipIpcOpenMemHandle( mapped, h, hipIpcMemLazyEnablePeerAccess ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2p_arena_close( void* mapped )
{
   if ( mapped )
      HH_CHECK_HIP( hipIpcCloseMemHandle( mapped ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2p_pack( const hyteg_hip_p2p_peer_t* peers,
                                      int                         npeers,
                                      double* const*              bases,
                                      const int*                  entry_buf,
                                      const int*                  entry_off,
                                      int                         n,
                                      unsigned long long          seq,
                                      unsigned*                   counter,
                                      hyteg_hip_stream_t          stream )
{
   if ( npeers <= 0 )
      return HYTEG_HIP_OK;
   HH_REQUIRE( peers && counter && n >= 0, "p2p_pack: null pointer" );
   HH_REQUIRE( n == 0 || ( bases && entry_buf && entry_off ), "p2p_pack: null pointer" );
   HH_REQUIRE( seq > 0, "p2p_pack: sequence numbers start at 1" );
   const int blocks = n > 0 ? ( n + kPackThreads - 1 ) / kPackThreads : 1; // an empty message still signals
   hipLaunchKernelGGL( p2p_pack_kernel, dim3( blocks ), dim3( kPackThreads ), 0, as_stream( stream ), peers, npeers, bases,
                       entry_buf, entry_off, n, seq, counter );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2p_wait( const unsigned long long* flags,
                                      int                       npeers,
                                      int                       stride,
                                      unsigned long long        seq,
                                      unsigned*                 status,
                                      unsigned                  timeout_ms,
                                      hyteg_hip_stream_t        stream )
{
   if ( npeers <= 0 )
      return HYTEG_HIP_OK;
   HH_REQUIRE( flags && status && stride >= 1, "p2p_wait: null pointer" );
   // wall_clock64 counts at 100 MHz on gfx9
   const unsigned long long ticks = (unsigned long long) ( timeout_ms ? timeout_ms : 20000u ) * 100000ull;
   hipLaunchKernelGGL( p2p_wait_kernel, dim3( 1 ), dim3( 64 ), 0, as_stream( stream ), flags, npeers, stride, seq, status, ticks );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}
}
