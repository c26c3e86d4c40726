// comm.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// Exchange plans and the transports behind the shared-point exchange of a storage distributed over several ranks:
// the semantics of src/hyteg/communication/BufferedCommunication.cpp:181-470 (startCommunication / endCommunication:
// pack, non-blocking send/receive per neighbour rank, wait, unpack) for one process per GPU.
#pragma once

#include <set>

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

// what the reduce kernel of an exchange has to wait for itself (hyteg_hip_reduce_shared_after_p2p)
struct ArrivalWait
{
   const unsigned long long* flags = nullptr;
   int                       npeers = 0, stride = 0;
   unsigned long long        seq    = 0;
   unsigned*                 status = nullptr;
   unsigned                  timeoutMs = 0;
};

// what a kernel needs to deliver the values of an exchange itself, as it computes them (hyteg_hip_p1_apply_cell_boundary_p2p)
struct PackArgs
{
   const hyteg_hip_p2p_peer_t* peers   = nullptr;
   int                         npeers  = 0;
   unsigned long long          seq     = 0;
   unsigned*                   counter = nullptr;
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
   // A transport that delivers the packed values itself (straight into the peers' receive slots) enqueues the pack here and
   // returns true; otherwise the storage gathers them into plan.sendBuffer.  Called right before exchangeBegin.
   // bases / entries: the arguments of hyteg_hip_gather_entries for this plan.
   virtual bool pack( const ExchangePlan&, int /*level*/, int /*key*/, double* const* /*bases*/, hyteg_hip_stream_t ) { return false; }
   // where the reduce kernel finds the received segments (concatenated per peer) of the exchange that exchangeEnd has just
   // completed; valid until the next exchangeBegin of the plan
   virtual double* recvBase( const ExchangePlan& plan, int /*level*/, int /*key*/ ) { return plan.recvBuffer; }
   // Instead of pack(): true = the caller's own kernel delivers the values of this exchange with `a` (the exchange has
   // begun; exchangeBegin must still be called, exchangeEnd as usual); false = not supported, nothing has happened.
   virtual bool packArgs( const ExchangePlan&, int /*level*/, int /*key*/, PackArgs& /*a*/, hyteg_hip_stream_t ) { return false; }
   // true: exchangeEnd has NOT made the compute stream wait for the arrival; the reduce kernel does it itself with `w`
   virtual bool arrivalWait( const ExchangePlan&, int /*level*/, int /*key*/, ArrivalWait& /*w*/ ) { return false; }
   // true: the exchange of this plan is ordered against whatever stream `compute` is at each call (not against a stream
   // fixed at set-up), so the caller may run it on a side stream next to other kernels
   virtual bool anyStream( int /*level*/, int /*key*/ ) const { return false; }
   // raises if a device-side wait of this transport has timed out since the last call (synchronises the stream)
   virtual void check( hyteg_hip_stream_t ) {}
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
   bool        anyStream( int, int ) const override { return true; }
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

// Peer to peer over xGMI without a library call per exchange (C-ABI hyteg_hip_p2p_*, comm_p2p.hip): the pack kernel stores
// into receive slots inside the peers' IPC-mapped arenas and publishes a sequence number, a one-wave kernel in front of the
// reduce kernel waits for the peers' numbers.  Set-up (once, through whatever channel the application has -- DistributedContext
// uses torch.distributed): exchange of the arena handles, then per plan every rank lays out the receive side in its own
// arena (layoutPlan) and learns where its peers expect its values (connectPlan).  Plans that were not connected, and the
// all-reduce of dot products, go through the transport this one wraps (RCCL in production, hooks in the gloo tests).
class P2PTransport : public Transport
{
 public:
   static constexpr int    kFlagStride = 8;   // flag words of a plan are 64 bytes apart
   static constexpr size_t kAlign      = 256; // of every piece of the arena

   P2PTransport( int nranks, int rank, size_t arenaBytes, std::shared_ptr< Transport > inner )
   : nranks_( nranks )
   , rank_( rank )
   , arenaBytes_( arenaBytes )
   , inner_( std::move( inner ) )
   , mapped_( (size_t) nranks, nullptr )
   {
      if ( !inner_ )
         throw std::runtime_error( "P2PTransport: needs an inner transport (set-up of unconnected plans, all-reduce)" );
      handle_.resize( HYTEG_HIP_P2P_HANDLE_BYTES );
      hipCheck( hyteg_hip_p2p_arena_create( arenaBytes, &arena_, handle_.data(), &arenaKind_ ), "P2PTransport: arena" );
      void* p = nullptr;
      hipCheck( hyteg_hip_malloc( &p, sizeof( unsigned ) ), "P2PTransport: malloc" );
      dStatus_ = static_cast< unsigned* >( p );
      const unsigned zero = 0;
      hipCheck( hyteg_hip_upload( dStatus_, &zero, sizeof( zero ), nullptr ), "P2PTransport: upload" );
      hipCheck( hyteg_hip_stream_synchronize( nullptr ), "P2PTransport: sync" );
      if ( const char* e = std::getenv( "HYTEG_HIP_P2P_TIMEOUT_MS" ) )
         timeoutMs_ = (unsigned) std::atoi( e );
      if ( const char* e = std::getenv( "HYTEG_HIP_P2P_FUSED_WAIT" ) ) // 0: a wait kernel of its own in front of the reduce kernel
         fusedWait_ = std::atoi( e ) != 0;
      live().insert( this );
   }
   // An exception between the begin and the end of an exchange (a failed launch, a Python hook that raised) would leave the plan
   // marked "in flight" and every later exchange of it would be refused: the host layer's C facade calls this from its catch
   // block, for every live transport (the failed operation's results are invalid either way).
   static void abortExchangesOfAllTransports()
   {
      for ( P2PTransport* t : live() )
         for ( auto& s : t->plans_ )
            s.second.inFlight = false;
   }
   ~P2PTransport() override
   {
      live().erase( this );
      if ( exchanged_ )
         hyteg_hip_stream_synchronize( lastCompute_ );
      hyteg_hip_stream_synchronize( nullptr );
      for ( auto& s : plans_ )
      {
         hyteg_hip_free( s.second.dPeers );
         hyteg_hip_free( s.second.dCounter );
      }
      for ( void* m : mapped_ )
         hyteg_hip_p2p_arena_close( m );
      hyteg_hip_free( dStatus_ );
      hyteg_hip_p2p_arena_destroy( arena_ );
   }
   const std::vector< unsigned char >& handle() const { return handle_; }
   int                                 arenaKind() const { return arenaKind_; }
   Transport&                          inner() const { return *inner_; }
   std::shared_ptr< Transport >        innerShared() const { return inner_; }

   // handles: nranks * HYTEG_HIP_P2P_HANDLE_BYTES bytes, rank by rank (the own entry is ignored)
   void openPeers( const unsigned char* handles )
   {
      for ( int r = 0; r < nranks_; ++r )
         if ( r != rank_ && !mapped_[(size_t) r] )
            hipCheck( hyteg_hip_p2p_arena_open( handles + (size_t) r * HYTEG_HIP_P2P_HANDLE_BYTES, &mapped_[(size_t) r] ),
                      "P2PTransport: open the arena of a peer" );
      opened_ = true;
   }
   // Receive side of a plan inside this rank's arena.  Returns, per peer of the plan, the byte offsets { slot 0, slot 1,
   // flag word } of that peer's segment: the application delivers triple k to rank plan.peers[k].
   std::vector< long long > layoutPlan( const ExchangePlan& plan, int level, int key )
   {
      PlanState& S = plans_[{ level, key }];
      if ( !S.laidOut )
      {
         const size_t bytes = (size_t) plan.totalRecv() * sizeof( double );
         S.recvOff[0]       = take( bytes );
         S.recvOff[1]       = take( bytes );
         S.flagOff          = take( plan.peers.size() * kFlagStride * sizeof( unsigned long long ) );
         S.laidOut          = true;
      }
      std::vector< long long > out;
      size_t                   seg = 0;
      for ( size_t k = 0; k < plan.peers.size(); ++k )
      {
         out.push_back( (long long) ( S.recvOff[0] + seg ) );
         out.push_back( (long long) ( S.recvOff[1] + seg ) );
         out.push_back( (long long) ( S.flagOff + k * kFlagStride * sizeof( unsigned long long ) ) );
         seg += (size_t) plan.recvCount[k] * sizeof( double );
      }
      return out;
   }
   // offsets: per peer of the plan the triple that peer's layoutPlan returned for THIS rank (offsets into its arena)
   void connectPlan( const ExchangePlan& plan, int level, int key, const long long* offsets )
   {
      if ( !opened_ )
         throw std::runtime_error( "P2PTransport::connectPlan: openPeers has not been called" );
      PlanState& S = plans_[{ level, key }];
      if ( !S.laidOut )
         throw std::runtime_error( "P2PTransport::connectPlan: layoutPlan comes first" );
      std::vector< hyteg_hip_p2p_peer_t > peers( plan.peers.size() );
      int                                 start = 0;
      for ( size_t k = 0; k < plan.peers.size(); ++k )
      {
         const int q = plan.peers[k];
         if ( q < 0 || q >= nranks_ || q == rank_ || !mapped_[(size_t) q] )
            throw std::runtime_error( "P2PTransport::connectPlan: peer rank " + std::to_string( q ) + " has no mapped arena" );
         char* base = static_cast< char* >( mapped_[(size_t) q] );
         for ( int s = 0; s < 3; ++s )
            if ( offsets[3 * k + s] < 0 || (size_t) offsets[3 * k + s] + ( s < 2 ? (size_t) plan.sendCount[k] * sizeof( double ) : 8 ) > arenaBytes_ )
               throw std::runtime_error( "P2PTransport::connectPlan: offset outside the peer's arena" );
         peers[k].slot[0] = reinterpret_cast< double* >( base + offsets[3 * k + 0] );
         peers[k].slot[1] = reinterpret_cast< double* >( base + offsets[3 * k + 1] );
         peers[k].flag    = reinterpret_cast< unsigned long long* >( base + offsets[3 * k + 2] );
         peers[k].start   = start;
         peers[k].count   = plan.sendCount[k];
         start += plan.sendCount[k];
      }
      if ( !peers.empty() )
      {
         void* d = nullptr;
         hipCheck( hyteg_hip_malloc( &d, peers.size() * sizeof( hyteg_hip_p2p_peer_t ) ), "P2PTransport: malloc" );
         hipCheck( hyteg_hip_upload( d, peers.data(), peers.size() * sizeof( hyteg_hip_p2p_peer_t ), nullptr ), "P2PTransport: upload" );
         S.dPeers = static_cast< hyteg_hip_p2p_peer_t* >( d );
         hipCheck( hyteg_hip_malloc( &d, sizeof( unsigned ) ), "P2PTransport: malloc" );
         const unsigned zero = 0;
         hipCheck( hyteg_hip_upload( d, &zero, sizeof( zero ), nullptr ), "P2PTransport: upload" );
         S.dCounter = static_cast< unsigned* >( d );
         hipCheck( hyteg_hip_stream_synchronize( nullptr ), "P2PTransport: sync" );
      }
      S.connected = true;
   }
   bool connected( int level, int key ) const
   {
      auto it = plans_.find( { level, key } );
      return it != plans_.end() && it->second.connected;
   }

   bool pack( const ExchangePlan& plan, int level, int key, double* const* bases, hyteg_hip_stream_t compute ) override
   {
      PlanState* S = find( level, key );
      if ( !S )
         return inner_->pack( plan, level, key, bases, compute );
      if ( S->inFlight )
         throw std::runtime_error( "P2PTransport: second exchange of a plan begun before the first one has ended" );
      S->inFlight  = true;
      lastCompute_ = compute, exchanged_ = true;
      ++S->seq;
      hipCheck( hyteg_hip_p2p_pack( S->dPeers, (int) plan.peers.size(), bases, plan.dSendBuf, plan.dSendOff, plan.totalSend(), S->seq,
                                    S->dCounter, compute ),
                "P2PTransport: pack" );
      return true;
   }
   bool packArgs( const ExchangePlan& plan, int level, int key, PackArgs& a, hyteg_hip_stream_t compute ) override
   {
      PlanState* S = find( level, key );
      if ( !S || plan.peers.empty() )
         return false;
      if ( S->inFlight )
         throw std::runtime_error( "P2PTransport: second exchange of a plan begun before the first one has ended" );
      S->inFlight  = true;
      lastCompute_ = compute, exchanged_ = true;
      ++S->seq;
      a.peers   = S->dPeers;
      a.npeers  = (int) plan.peers.size();
      a.seq     = S->seq;
      a.counter = S->dCounter;
      return true;
   }
   void exchangeBegin( const ExchangePlan& plan, int level, int key, hyteg_hip_stream_t compute ) override
   {
      if ( !find( level, key ) )
         inner_->exchangeBegin( plan, level, key, compute );
   }
   void exchangeEnd( const ExchangePlan& plan, int level, int key, hyteg_hip_stream_t compute ) override
   {
      PlanState* S = find( level, key );
      if ( !S )
         return inner_->exchangeEnd( plan, level, key, compute );
      if ( plan.peers.empty() )
         return;
      if ( !S->inFlight )
         throw std::runtime_error( "P2PTransport: exchangeEnd without exchangeBegin" );
      S->inFlight = false;
      if ( fusedWait_ )
         return; // the reduce kernel waits (arrivalWait)
      hipCheck( hyteg_hip_p2p_wait( reinterpret_cast< const unsigned long long* >( static_cast< char* >( arena_ ) + S->flagOff ),
                                    (int) plan.peers.size(), kFlagStride, S->seq, dStatus_, timeoutMs_, compute ),
                "P2PTransport: wait" );
   }
   bool arrivalWait( const ExchangePlan& plan, int level, int key, ArrivalWait& w ) override
   {
      PlanState* S = find( level, key );
      if ( !S )
         return inner_->arrivalWait( plan, level, key, w );
      if ( !fusedWait_ || plan.peers.empty() )
         return false;
      w.flags     = reinterpret_cast< const unsigned long long* >( static_cast< char* >( arena_ ) + S->flagOff );
      w.npeers    = (int) plan.peers.size();
      w.stride    = kFlagStride;
      w.seq       = S->seq;
      w.status    = dStatus_;
      w.timeoutMs = timeoutMs_;
      return true;
   }
   double* recvBase( const ExchangePlan& plan, int level, int key ) override
   {
      PlanState* S = find( level, key );
      if ( !S )
         return inner_->recvBase( plan, level, key );
      return reinterpret_cast< double* >( static_cast< char* >( arena_ ) + S->recvOff[S->seq & 1ull] );
   }
   void check( hyteg_hip_stream_t compute ) override
   {
      unsigned st = 0;
      hipCheck( hyteg_hip_download( &st, dStatus_, sizeof( st ), compute ), "P2PTransport: download" );
      hipCheck( hyteg_hip_stream_synchronize( compute ), "P2PTransport: sync" );
      if ( st )
      {
         const unsigned zero = 0;
         hipCheck( hyteg_hip_upload( dStatus_, &zero, sizeof( zero ), compute ), "P2PTransport: upload" );
         hipCheck( hyteg_hip_stream_synchronize( compute ), "P2PTransport: sync" );
         throw std::runtime_error( "P2PTransport: a wait for the values of peer slot " + std::to_string( st - 1 ) +
                                   " timed out -- the results since the last check are not valid" );
      }
      inner_->check( compute );
   }
   // A global sum is a point where the host waits for the device anyway (the local sums have just been read back): the status word
   // of the arrival waits is read there as well, so that a solver that only ever calls dot products (CG, MINRES, the multigrid
   // residual norms) cannot continue on values a timed-out wait let through.
   void allreduceSum( double* values, int n ) override
   {
      if ( exchanged_ ) // (the stream handle itself may be null: the default stream)
         check( lastCompute_ );
      inner_->allreduceSum( values, n );
   }
   bool        collective() const override { return inner_->collective(); }
   bool        anyStream( int level, int key ) const override { return connected( level, key ) || inner_->anyStream( level, key ); }
   const char* name() const override { return "p2p"; }

 private:
   struct PlanState
   {
      bool                  laidOut = false, connected = false, inFlight = false;
      size_t                recvOff[2] = { 0, 0 }, flagOff = 0;
      hyteg_hip_p2p_peer_t* dPeers   = nullptr;
      unsigned*             dCounter = nullptr;
      unsigned long long    seq      = 0;
   };
   PlanState* find( int level, int key )
   {
      auto it = plans_.find( { level, key } );
      return it != plans_.end() && it->second.connected ? &it->second : nullptr;
   }
   size_t take( size_t bytes )
   {
      const size_t off = used_;
      used_ += ( std::max< size_t >( bytes, 1 ) + kAlign - 1 ) / kAlign * kAlign;
      if ( used_ > arenaBytes_ )
         throw std::runtime_error( "P2PTransport: arena of " + std::to_string( arenaBytes_ ) + " bytes is too small" );
      return off;
   }
   int                                          nranks_, rank_;
   size_t                                       arenaBytes_, used_ = 0;
   std::shared_ptr< Transport >                 inner_;
   std::vector< void* >                         mapped_;
   std::vector< unsigned char >                 handle_;
   void*                                        arena_     = nullptr;
   int                                          arenaKind_ = 0;
   bool                                         opened_    = false;
   unsigned*                                    dStatus_   = nullptr;
   static std::set< P2PTransport* >& live()
   {
      static std::set< P2PTransport* > s;
      return s;
   }
   unsigned                                     timeoutMs_ = 0;
   bool                                         fusedWait_ = true;
   hyteg_hip_stream_t                           lastCompute_ = nullptr;
   bool                                         exchanged_ = false; // an exchange has been packed on lastCompute_
   std::map< std::pair< int, int >, PlanState > plans_;
};

} // namespace hyteg
