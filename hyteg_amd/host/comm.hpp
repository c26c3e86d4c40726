// comm.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// Exchange plans and the transports behind the shared-point exchange of a storage distributed over several ranks:
// the semantics of src/hyteg/communication/BufferedCommunication.cpp:181-470 (startCommunication / endCommunication:
// pack, non-blocking send/receive per neighbour rank, wait, unpack) for one process per GPU.
#pragma once

#include "types.hpp"

namespace hyteg {

// Who holds which copy of the DoFs that a boundary class of macro-primitives shares between cells, and what travels
// to which rank.  Built by PrimitiveStorage::buildPlan (rank-independent enumeration order on both sides).
struct ExchangePlan
{
   // host copies
   std::vector< int > groupPtr, entryBuf, entryOff; // entryBuf < nLocal: local cell; else nLocal + peer slot
   std::vector< int > peers;                        // ranks we exchange with, ascending
   std::vector< int > sendCount, recvCount;         // per peer
   std::vector< int > sendBuf, sendOff;             // concatenated per peer: (local cell, offset)
   // device copies
   int *dGroupPtr = nullptr, *dEntryBuf = nullptr, *dEntryOff = nullptr, *dSendBuf = nullptr, *dSendOff = nullptr;
   // communication buffers (device), registered by the application for multi-rank runs or allocated here
   double *sendBuffer = nullptr, *recvBuffer = nullptr;
   bool    ownsBuffers = false;
   bool    onDevice    = false;
   int     ngroups() const { return (int) groupPtr.size() - 1; }
   int     totalSend() const { return (int) sendBuf.size(); }
   int     totalRecv() const
   {
      int t = 0;
      for ( int r : recvCount )
         t += r;
      return t;
   }
};

// Moves plan.sendBuffer (segments per peer) into the peers' plan.recvBuffer.  `key` = cls + 2 * dofKind identifies the
// plan of a level (boundary class 0 / 1; vertex-DoF / edge-DoF arrays).
class Transport
{
 public:
   virtual ~Transport() = default;
   // The pack kernel has been enqueued on `compute`.  Starts the transfer; may return before it has completed, so that
   // kernels enqueued next (the interior stencil) overlap it.
   virtual void exchangeBegin( const ExchangePlan& plan, int level, int key, hyteg_hip_stream_t compute ) = 0;
   // Work enqueued on `compute` after this call sees the received values in plan.recvBuffer.
   virtual void exchangeEnd( const ExchangePlan& plan, int level, int key, hyteg_hip_stream_t compute ) = 0;
   // in-place sum over all ranks of n doubles in HOST memory (walberla::mpi::allReduceInplace, VertexDoFFunction.cpp:1717)
   virtual void allreduceSum( double* values, int n ) = 0;
   // A collective transport (all-to-all) must be entered by every rank, also by ranks without peers in that plan;
   // a point-to-point transport is only called by ranks that have peers.
   virtual bool collective() const = 0;
   virtual const char* name() const = 0;
};

// callbacks set by the embedding application (hyteg_amd/host.py: torch.distributed over gloo for the CPU tests); a
// non-zero return value means the callback failed: the host layer throws instead of reducing stale receive buffers
struct CommHooks
{
   int ( *exchangeBegin )( void* user, int level, int key ) = nullptr;
   int ( *exchangeEnd )( void* user, int level, int key )   = nullptr;
   int ( *allreduceSum )( void* user, double* values, int n ) = nullptr;
   void* user                                                  = nullptr;
};

class HookTransport : public Transport
{
 public:
   explicit HookTransport( const CommHooks& h )
   : h_( h )
   {
      if ( !h.exchangeBegin || !h.exchangeEnd || !h.allreduceSum )
         throw std::runtime_error( "HookTransport: all three hooks must be set" );
   }
   void exchangeBegin( const ExchangePlan&, int level, int key, hyteg_hip_stream_t ) override
   {
      if ( int rc = h_.exchangeBegin( h_.user, level, key ) )
         throw std::runtime_error( "exchange hook (begin) failed with code " + std::to_string( rc ) + " at level " +
                                   std::to_string( level ) + ", plan " + std::to_string( key ) );
   }
   void exchangeEnd( const ExchangePlan&, int level, int key, hyteg_hip_stream_t ) override
   {
      if ( int rc = h_.exchangeEnd( h_.user, level, key ) )
         throw std::runtime_error( "exchange hook (end) failed with code " + std::to_string( rc ) + " at level " +
                                   std::to_string( level ) + ", plan " + std::to_string( key ) );
   }
   void allreduceSum( double* values, int n ) override
   {
      if ( int rc = h_.allreduceSum( h_.user, values, n ) )
         throw std::runtime_error( "all-reduce hook failed with code " + std::to_string( rc ) );
   }
   bool        collective() const override { return true; }
   const char* name() const override { return "hooks"; }

 private:
   CommHooks h_;
};

// RCCL over xGMI, issued from here: one group of ncclSend / ncclRecv pairs per exchange on a communication stream of
// its own, ordered against the compute stream with events only (no host synchronisation, no callbacks).
//   Begin:  [compute] pack kernel -> event "packed";  [comm] wait "packed", send/recv group, event "arrived"
//   End:    [compute] wait "arrived" -> reduce kernel
// Between Begin and End the compute stream runs the interior stencil kernel.  Buffer reuse is safe by stream order: the
// next pack of the same plan comes after the reduce kernel that waited for "arrived" (the sends are part of that group),
// and the next receive is enqueued behind the next "packed", which the compute stream records after that reduce kernel.
class RcclTransport : public Transport
{
 public:
   RcclTransport( int nranks, int rank, const unsigned char* uniqueId )
   {
      hipCheck( hyteg_hip_comm_create( &comm_, nranks, rank, uniqueId ), "RcclTransport: comm_create" );
      hipCheck( hyteg_hip_stream_create( &commStream_ ), "RcclTransport: stream_create" );
      void* p = nullptr;
      hipCheck( hyteg_hip_malloc( &p, kScalars * sizeof( double ) ), "RcclTransport: malloc" );
      scalars_ = static_cast< double* >( p );
   }
   ~RcclTransport() override
   {
      hyteg_hip_stream_synchronize( commStream_ );
      for ( auto& e : events_ )
      {
         hyteg_hip_event_destroy( e.second.packed );
         hyteg_hip_event_destroy( e.second.arrived );
      }
      hyteg_hip_free( scalars_ );
      hyteg_hip_comm_destroy( comm_ );
      hyteg_hip_stream_destroy( commStream_ );
   }
   void exchangeBegin( const ExchangePlan& plan, int level, int key, hyteg_hip_stream_t compute ) override
   {
      if ( plan.peers.empty() )
         return;
      Events& e = eventsFor( level, key );
      hipCheck( hyteg_hip_event_record( e.packed, compute ), "RcclTransport: record packed" );
      hipCheck( hyteg_hip_stream_wait_event( commStream_, e.packed ), "RcclTransport: wait packed" );
      hipCheck( hyteg_hip_comm_exchange( comm_, (int) plan.peers.size(), plan.peers.data(), plan.sendBuffer, plan.sendCount.data(),
                                         plan.recvBuffer, plan.recvCount.data(), commStream_ ),
                "RcclTransport: exchange" );
      hipCheck( hyteg_hip_event_record( e.arrived, commStream_ ), "RcclTransport: record arrived" );
   }
   void exchangeEnd( const ExchangePlan& plan, int level, int key, hyteg_hip_stream_t compute ) override
   {
      if ( plan.peers.empty() )
         return;
      hipCheck( hyteg_hip_stream_wait_event( compute, eventsFor( level, key ).arrived ), "RcclTransport: wait arrived" );
   }
   void allreduceSum( double* values, int n ) override
   {
      if ( n > kScalars )
         throw std::runtime_error( "RcclTransport::allreduceSum: too many values" );
      hipCheck( hyteg_hip_upload( scalars_, values, n * sizeof( double ), commStream_ ), "RcclTransport: upload" );
      hipCheck( hyteg_hip_comm_allreduce_sum( comm_, scalars_, n, commStream_ ), "RcclTransport: allreduce" );
      hipCheck( hyteg_hip_download( values, scalars_, n * sizeof( double ), commStream_ ), "RcclTransport: download" );
      hipCheck( hyteg_hip_stream_synchronize( commStream_ ), "RcclTransport: sync" );
   }
   bool        collective() const override { return false; }
   const char* name() const override { return "rccl"; }

 private:
   struct Events
   {
      hyteg_hip_event_t packed = nullptr, arrived = nullptr;
   };
   Events& eventsFor( int level, int key )
   {
      auto it = events_.find( { level, key } );
      if ( it == events_.end() )
      {
         Events e;
         hipCheck( hyteg_hip_event_create( &e.packed ), "RcclTransport: event_create" );
         hipCheck( hyteg_hip_event_create( &e.arrived ), "RcclTransport: event_create" );
         it = events_.emplace( std::make_pair( level, key ), e ).first;
      }
      return it->second;
   }
   static constexpr int                         kScalars = 16;
   hyteg_hip_comm_t                             comm_       = nullptr;
   hyteg_hip_stream_t                           commStream_ = nullptr;
   double*                                      scalars_    = nullptr;
   std::map< std::pair< int, int >, Events >    events_;
};

} // namespace hyteg
