// Communication part of the C-ABI: events, and the neighbour exchange / all-reduce over RCCL (xGMI).
//
// Replaces, for one process per GPU, what the reference does with waLBerla's BufferSystem over MPI:
//   src/hyteg/communication/BufferedCommunication.cpp:181-470   (start/endCommunication: pack, Isend/Irecv, wait, unpack)
//   src/hyteg/p1functionspace/VertexDoFAdditivePackInfo.hpp:676-745 (payloads of the additive exchange)
//   src/hyteg/p1functionspace/VertexDoFFunction.cpp:1710-1717   (dotGlobal: allReduceInplace( SUM ) of one scalar)
// The message pattern is sparse neighbour point-to-point, so an exchange is ONE group of ncclSend / ncclRecv pairs
// (one per peer rank) on the caller's communication stream -- no host synchronisation, no staging; the caller orders
// it against its pack / reduce kernels with events (hyteg_hip_event_*).
//
// librccl is resolved at run time with dlopen: a process that already holds one (PyTorch ships its own copy and loads
// it with torch) must not map a second one, and a process that never communicates does not need any.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

#include <mutex>

#include <rccl/rccl.h>

#include "common.hpp"

namespace hyteg_hip {
namespace {

struct RcclApi
{
   void*        handle = nullptr;
   std::string  origin;
   ncclResult_t ( *GetUniqueId )( ncclUniqueId* )                                                                    = nullptr;
   ncclResult_t ( *CommInitRank )( ncclComm_t*, int, ncclUniqueId, int )                                             = nullptr;
   ncclResult_t ( *CommDestroy )( ncclComm_t )                                                                       = nullptr;
   ncclResult_t ( *GroupStart )()                                                                                    = nullptr;
   ncclResult_t ( *GroupEnd )()                                                                                      = nullptr;
   ncclResult_t ( *Send )( const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t )                       = nullptr;
   ncclResult_t ( *Recv )( void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t )                             = nullptr;
   ncclResult_t ( *AllReduce )( const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t )   = nullptr;
   const char* ( *GetErrorString )( ncclResult_t )                                                                   = nullptr;
   ncclResult_t ( *GetVersion )( int* )                                                                              = nullptr;
};

RcclApi     g_rccl;
std::mutex  g_rccl_mutex;
std::string g_rccl_error;

bool load_rccl()
{
   std::lock_guard< std::mutex > lock( g_rccl_mutex );
   if ( g_rccl.handle )
      return true;
   // 1. a copy that is already mapped (torch's is NEEDED as "librccl.so", ROCm's has the soname "librccl.so.1")
   const char* loaded[] = { "librccl.so", "librccl.so.1" };
   for ( const char* n : loaded )
      if ( void* h = dlopen( n, RTLD_NOW | RTLD_NOLOAD ) )
      {
         g_rccl.handle = h;
         g_rccl.origin = std::string( "already loaded: " ) + n;
         break;
      }
   // 2. HYTEG_HIP_RCCL_LIB, then the loader's search path, then ROCm's default location
   if ( !g_rccl.handle )
   {
      const char* env     = getenv( "HYTEG_HIP_RCCL_LIB" );
      const char* names[] = { env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so" };
      for ( const char* n : names )
         if ( n && n[0] )
            if ( void* h = dlopen( n, RTLD_NOW | RTLD_GLOBAL ) )
            {
               g_rccl.handle = h;
               g_rccl.origin = std::string( "dlopen: " ) + n;
               break;
            }
   }
   if ( !g_rccl.handle )
   {
      g_rccl_error = std::string( "librccl not found: " ) + ( dlerror() ? dlerror() : "" );
      return false;
   }
   bool ok   = true;
   auto need = [&]( auto& fn, const char* sym ) {
      fn = reinterpret_cast< std::remove_reference_t< decltype( fn ) > >( dlsym( g_rccl.handle, sym ) );
      if ( !fn )
      {
         ok           = false;
         g_rccl_error = std::string( "librccl lacks " ) + sym;
      }
   };
   need( g_rccl.GetUniqueId, "ncclGetUniqueId" );
   need( g_rccl.CommInitRank, "ncclCommInitRank" );
   need( g_rccl.CommDestroy, "ncclCommDestroy" );
   need( g_rccl.GroupStart, "ncclGroupStart" );
   need( g_rccl.GroupEnd, "ncclGroupEnd" );
   need( g_rccl.Send, "ncclSend" );
   need( g_rccl.Recv, "ncclRecv" );
   need( g_rccl.AllReduce, "ncclAllReduce" );
   need( g_rccl.GetErrorString, "ncclGetErrorString" );
   need( g_rccl.GetVersion, "ncclGetVersion" );
   if ( !ok )
      g_rccl.handle = nullptr;
   return ok;
}

struct Comm
{
   ncclComm_t comm   = nullptr;
   int        nranks = 0, rank = 0;
};

#define HH_CHECK_RCCL( expr )                                                                                       \
   do                                                                                                               \
   {                                                                                                                \
      ncclResult_t _r = ( expr );                                                                                   \
      if ( _r != ncclSuccess )                                                                                      \
         return ::hyteg_hip::fail( HYTEG_HIP_ELAUNCH, std::string( #expr ) + ": " + g_rccl.GetErrorString( _r ) );  \
   } while ( 0 )

} // namespace
} // namespace hyteg_hip

using namespace hyteg_hip;

static_assert( sizeof( ncclUniqueId ) == HYTEG_HIP_COMM_ID_BYTES, "ncclUniqueId size" );

extern "C" {

// ---- events ---------------------------------------------------------------------------------------------------------
HYTEG_HIP_API int hyteg_hip_event_create( hyteg_hip_event_t* event )
{
   HH_REQUIRE( event != nullptr, "event_create: null out pointer" );
   hipEvent_t e;
   HH_CHECK_HIP( hipEventCreateWithFlags( &e, hipEventDisableTiming ) );
   *event = reinterpret_cast< hyteg_hip_event_t >( e );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_event_create_timing( hyteg_hip_event_t* event )
{
   HH_REQUIRE( event != nullptr, "event_create_timing: null out pointer" );
   // hipEventDisableSystemFence: "for events that are only used to measure timing ... avoids the cost of cache writeback and
   // invalidation, and the performance impact of those actions on the execution of following work" (hip_runtime_api.h); results
   // are read after a stream / device synchronisation, never through these events.  HYTEG_HIP_TIMING_EVENT_FENCE=1 keeps the fence.
   const char*    env   = std::getenv( "HYTEG_HIP_TIMING_EVENT_FENCE" );
   const unsigned flags = ( env && env[0] == '1' ) ? hipEventDefault : hipEventDisableSystemFence;
   hipEvent_t     e;
   HH_CHECK_HIP( hipEventCreateWithFlags( &e, flags ) );
   *event = reinterpret_cast< hyteg_hip_event_t >( e );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_event_elapsed_ms( hyteg_hip_event_t start, hyteg_hip_event_t stop, float* ms )
{
   HH_REQUIRE( start && stop && ms, "event_elapsed_ms: null argument" );
   HH_CHECK_HIP( hipEventSynchronize( reinterpret_cast< hipEvent_t >( stop ) ) );
   HH_CHECK_HIP( hipEventElapsedTime( ms, reinterpret_cast< hipEvent_t >( start ), reinterpret_cast< hipEvent_t >( stop ) ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_event_destroy( hyteg_hip_event_t event )
{
   HH_CHECK_HIP( hipEventDestroy( reinterpret_cast< hipEvent_t >( event ) ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_event_record( hyteg_hip_event_t event, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( event != nullptr, "event_record: null event" );
   HH_CHECK_HIP( hipEventRecord( reinterpret_cast< hipEvent_t >( event ), as_stream( stream ) ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_stream_wait_event( hyteg_hip_stream_t stream, hyteg_hip_event_t event )
{
   HH_REQUIRE( event != nullptr, "stream_wait_event: null event" );
   HH_CHECK_HIP( hipStreamWaitEvent( as_stream( stream ), reinterpret_cast< hipEvent_t >( event ), 0 ) );
   return HYTEG_HIP_OK;
}

// ---- RCCL communicator ----------------------------------------------------------------------------------------------
HYTEG_HIP_API int hyteg_hip_comm_available( char* origin, size_t buflen )
{
   if ( !load_rccl() )
      return fail( HYTEG_HIP_EINVAL, g_rccl_error );
   if ( origin && buflen > 0 )
   {
      int v = 0;
      g_rccl.GetVersion( &v );
      snprintf( origin, buflen, "%s (version code %d)", g_rccl.origin.c_str(), v );
   }
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_comm_unique_id( unsigned char* id )
{
   HH_REQUIRE( id != nullptr, "comm_unique_id: null pointer" );
   if ( !load_rccl() )
      return fail( HYTEG_HIP_EINVAL, g_rccl_error );
   ncclUniqueId uid;
   HH_CHECK_RCCL( g_rccl.GetUniqueId( &uid ) );
   std::memcpy( id, &uid, sizeof( uid ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_comm_create( hyteg_hip_comm_t* comm, int nranks, int rank, const unsigned char* id )
{
   HH_REQUIRE( comm && id, "comm_create: null pointer" );
   HH_REQUIRE( nranks >= 1 && rank >= 0 && rank < nranks, "comm_create: bad rank / number of ranks" );
   if ( !load_rccl() )
      return fail( HYTEG_HIP_EINVAL, g_rccl_error );
   ncclUniqueId uid;
   std::memcpy( &uid, id, sizeof( uid ) );
   auto* c   = new Comm;
   c->nranks = nranks;
   c->rank   = rank;
   ncclResult_t r = g_rccl.CommInitRank( &c->comm, nranks, uid, rank ); // collective over all ranks, on the current device
   if ( r != ncclSuccess )
   {
      delete c;
      return fail( HYTEG_HIP_ELAUNCH, std::string( "ncclCommInitRank: " ) + g_rccl.GetErrorString( r ) );
   }
   *comm = reinterpret_cast< hyteg_hip_comm_t >( c );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_comm_destroy( hyteg_hip_comm_t comm )
{
   auto* c = reinterpret_cast< Comm* >( comm );
   if ( !c )
      return HYTEG_HIP_OK;
   if ( c->comm )
      HH_CHECK_RCCL( g_rccl.CommDestroy( c->comm ) );
   delete c;
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_comm_exchange( hyteg_hip_comm_t   comm,
                                           int                npeers,
                                           const int*         peers,
                                           const double*      send,
                                           const int*         send_count,
                                           double*            recv,
                                           const int*         recv_count,
                                           hyteg_hip_stream_t stream )
{
   auto* c = reinterpret_cast< Comm* >( comm );
   HH_REQUIRE( c && c->comm, "comm_exchange: null communicator" );
   HH_REQUIRE( npeers >= 0, "comm_exchange: negative number of peers" );
   if ( npeers == 0 )
      return HYTEG_HIP_OK;
   HH_REQUIRE( peers && send_count && recv_count, "comm_exchange: null pointer" );
   size_t so = 0, ro = 0;
   HH_CHECK_RCCL( g_rccl.GroupStart() );
   for ( int k = 0; k < npeers; ++k )
   {
      HH_REQUIRE( peers[k] >= 0 && peers[k] < c->nranks, "comm_exchange: bad peer rank" ); // own rank: allowed (loop-back)
      if ( send_count[k] > 0 )
         HH_CHECK_RCCL( g_rccl.Send( send + so, (size_t) send_count[k], ncclDouble, peers[k], c->comm, as_stream( stream ) ) );
      if ( recv_count[k] > 0 )
         HH_CHECK_RCCL( g_rccl.Recv( recv + ro, (size_t) recv_count[k], ncclDouble, peers[k], c->comm, as_stream( stream ) ) );
      so += (size_t) send_count[k];
      ro += (size_t) recv_count[k];
   }
   HH_CHECK_RCCL( g_rccl.GroupEnd() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_comm_allreduce_sum( hyteg_hip_comm_t comm, double* values, int n, hyteg_hip_stream_t stream )
{
   auto* c = reinterpret_cast< Comm* >( comm );
   HH_REQUIRE( c && c->comm, "comm_allreduce_sum: null communicator" );
   HH_REQUIRE( values && n > 0, "comm_allreduce_sum: nothing to reduce" );
   HH_CHECK_RCCL( g_rccl.AllReduce( values, values, (size_t) n, ncclDouble, ncclSum, c->comm, as_stream( stream ) ) );
   return HYTEG_HIP_OK;
}
}
