// storage.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// PrimitiveStorage: macro-primitives, neighbourhood, boundary flags, rank assignment, exchange plans,
// communication hooks (src/hyteg/primitivestorage/, src/hyteg/communication/BufferedCommunication.cpp)
#pragma once

#include "comm.hpp"
#include "mesh.hpp"
#include "timing.hpp"

namespace hyteg {

// =====================================================================================================
// PrimitiveStorage: macro-vertices / edges / faces / cells, neighbourhood, boundary flags, rank assignment.
// Cell-local numbering (src/hyteg/primitives/Cell.hpp, src/hyteg/indexing/MacroCellIndexing.cpp:36-91):
// faces 0:(0,1,2) 1:(0,1,3) 2:(0,2,3) 3:(1,2,3); edges 0:(0,1) 1:(0,2) 2:(1,2) 3:(0,3) 4:(1,3) 5:(2,3).
// Slot order of all per-cell 14-arrays: { edge0..5, face0..3, vertex0..3 } (the grid-transfer kernels' order).
// =====================================================================================================
struct MacroCell
{
   int                      id;
   std::array< int, 4 >     v;      // global vertex ids, local order = mesh order
   std::array< Point3D, 4 > coords; // getCoordinates()
   std::array< int, 6 >     edges;  // global edge ids by local edge
   std::array< int, 4 >     faces;  // global face ids by local face
   int                      rank;
   int                      localIndex; // index among this rank's cells, -1 if remote
};
struct MacroPrimitive
{
   std::vector< int > v;     // sorted global vertex ids (1, 2 or 3)
   std::vector< int > cells; // adjacent global cell ids, ascending
   bool               onBoundary = false;
   uint_t             getNumNeighborCells() const { return cells.size(); }
};

static const int kCellFaceVerts[4][3] = { { 0, 1, 2 }, { 0, 1, 3 }, { 0, 2, 3 }, { 1, 2, 3 } };
static const int kCellEdgeVerts[6][2] = { { 0, 1 }, { 0, 2 }, { 1, 2 }, { 0, 3 }, { 1, 3 }, { 2, 3 } };

class PrimitiveStorage
{
 public:
   PrimitiveStorage( const MeshInfo& mesh, int rank = 0, int nranks = 1 )
   : rank_( rank )
   , nranks_( nranks )
   {
      if ( nranks < 1 || rank < 0 || rank >= nranks )
         throw std::runtime_error( "PrimitiveStorage: bad rank / number of ranks" );
      std::map< std::vector< int >, int > edgeId, faceId;
      vertices_.resize( mesh.vertices.size() );
      for ( uint_t i = 0; i < vertices_.size(); ++i )
         vertices_[i].v = { (int) i };
      for ( uint_t c = 0; c < mesh.cells.size(); ++c )
      {
         MacroCell cell;
         cell.id = (int) c;
         cell.v  = mesh.cells[c];
         for ( int k = 0; k < 4; ++k )
         {
            if ( cell.v[k] < 0 || cell.v[k] >= (int) mesh.vertices.size() )
               throw std::runtime_error( "PrimitiveStorage: cell refers to a missing vertex" );
            cell.coords[k] = mesh.vertices[cell.v[k]];
            vertices_[cell.v[k]].cells.push_back( (int) c );
         }
         for ( int e = 0; e < 6; ++e )
         {
            std::vector< int > key = { cell.v[kCellEdgeVerts[e][0]], cell.v[kCellEdgeVerts[e][1]] };
            std::sort( key.begin(), key.end() );
            auto it = edgeId.find( key );
            if ( it == edgeId.end() )
            {
               it = edgeId.emplace( key, (int) edges_.size() ).first;
               edges_.push_back( MacroPrimitive{ key, {}, false } );
            }
            cell.edges[e] = it->second;
            edges_[it->second].cells.push_back( (int) c );
         }
         for ( int f = 0; f < 4; ++f )
         {
            std::vector< int > key = { cell.v[kCellFaceVerts[f][0]], cell.v[kCellFaceVerts[f][1]], cell.v[kCellFaceVerts[f][2]] };
            std::sort( key.begin(), key.end() );
            auto it = faceId.find( key );
            if ( it == faceId.end() )
            {
               it = faceId.emplace( key, (int) faces_.size() ).first;
               faces_.push_back( MacroPrimitive{ key, {}, false } );
            }
            cell.faces[f] = it->second;
            faces_[it->second].cells.push_back( (int) c );
         }
         // SetupPrimitiveStorage's default balancing is round robin over ranks (loadbalancing/SimpleBalancer.cpp: roundRobin)
         cell.rank       = (int) ( c % (uint_t) nranks );
         cell.localIndex = -1;
         cells_.push_back( cell );
      }
      // setMeshBoundaryFlagsOnBoundary( 1, 0, true ): a face with one neighbour cell is on the boundary, and so is
      // every edge / vertex of such a face (SetupPrimitiveStorage.cpp, onBoundary())
      for ( auto& f : faces_ )
      {
         if ( f.cells.size() > 2 )
            throw std::runtime_error( "PrimitiveStorage: face with more than two neighbour cells" );
         f.onBoundary = f.cells.size() == 1;
         if ( f.onBoundary )
         {
            for ( int a = 0; a < 3; ++a )
            {
               vertices_[f.v[a]].onBoundary = true;
               for ( int b = a + 1; b < 3; ++b )
               {
                  std::vector< int > key = { f.v[a], f.v[b] };
                  edges_[edgeId.at( key )].onBoundary = true;
               }
            }
         }
      }
      for ( auto& c : cells_ )
         if ( c.rank == rank_ )
         {
            c.localIndex = (int) localCells_.size();
            localCells_.push_back( c.id );
         }
      for ( auto& p : vertices_ )
         std::sort( p.cells.begin(), p.cells.end() );
      // no device work here: topology and exchange plans can be built (and tested) without a GPU
   }
   ~PrimitiveStorage()
   {
      if ( dotResult_ )
         hyteg_hip_free( dotResult_ );
      if ( dotWorkspace_ )
         hyteg_hip_free( dotWorkspace_ );
      for ( void* p : scratchAll_ )
         hyteg_hip_free( p );
      if ( sideStream_ )
      {
         hyteg_hip_stream_synchronize( sideStream_ );
         hyteg_hip_event_destroy( forked_ );
         hyteg_hip_event_destroy( joined_ );
         hyteg_hip_stream_destroy( sideStream_ );
      }
   }
   PrimitiveStorage( const PrimitiveStorage& )            = delete;
   PrimitiveStorage& operator=( const PrimitiveStorage& ) = delete;

   bool hasGlobalCells() const { return !cells_.empty(); }
   int  rank() const { return rank_; }
   int  numRanks() const { return nranks_; }

   const std::vector< MacroCell >&      getCells() const { return cells_; }
   const std::vector< MacroPrimitive >& getFaces() const { return faces_; }
   const std::vector< MacroPrimitive >& getEdges() const { return edges_; }
   const std::vector< MacroPrimitive >& getVertices() const { return vertices_; }
   const std::vector< int >&            getLocalCellIDs() const { return localCells_; }
   uint_t                               getNumberOfLocalCells() const { return localCells_.size(); }
   const MacroCell&                     getLocalCell( uint_t i ) const { return cells_[localCells_.at( i )]; }

   // boundary type of every primitive on the domain boundary (BoundaryCondition::create0123BC maps flag 1 -> Dirichlet)
   void    setBoundaryType( DoFType t ) { boundaryType_ = t; }
   DoFType boundaryTypeOf( bool onBoundary ) const { return onBoundary ? boundaryType_ : Inner; }

   // the macro-primitive behind slot s (0..13) of a cell
   const MacroPrimitive& primitiveOfSlot( const MacroCell& c, int s ) const
   {
      if ( s < 6 )
         return edges_[c.edges[s]];
      if ( s < 10 )
         return faces_[c.faces[s - 6]];
      return vertices_[c.v[s - 10]];
   }

   // point mask of a cell for a DoFType flag: the per-primitive test of P1Operator.hpp:213-303
   unsigned maskFor( const MacroCell& c, DoFType flag ) const
   {
      unsigned m = testFlag( Inner, flag ) ? HYTEG_HIP_MASK_INNER : 0u; // a macro-cell is never on the mesh boundary
      for ( int s = 0; s < 14; ++s )
         if ( testFlag( boundaryTypeOf( primitiveOfSlot( c, s ).onBoundary ), flag ) )
            m |= 1u << s;
      return m;
   }
   // like maskFor but a shared primitive is counted by its lowest-numbered neighbour cell only (dot products)
   unsigned ownedMaskFor( const MacroCell& c, DoFType flag ) const
   {
      unsigned m = maskFor( c, flag );
      for ( int s = 0; s < 14; ++s )
         if ( primitiveOfSlot( c, s ).cells.front() != c.id )
            m &= ~( 1u << s );
      return m;
   }
   // numNeighborCells of the 14 primitives around a cell, in the grid-transfer kernels' argument order
   std::array< double, 14 > numNeighborCells( const MacroCell& c ) const
   {
      std::array< double, 14 > n{};
      for ( int s = 0; s < 14; ++s )
         n[s] = (double) primitiveOfSlot( c, s ).cells.size();
      return n;
   }

   // ---- batched launches (p1_batch.hip): one launch for all local cells on the levels where a cell is small ----
   // Default: levels <= 6 whenever the rank owns more than one cell (a single cell is served better by the tuned per-cell
   // kernels: measured 1.03 vs 1.32 ms per V(3,3) Jacobi cycle); HYTEG_AMD_BATCH_MAX_LEVEL overrides (-1 disables).
   bool useBatch( uint_t level ) const
   {
      if ( batchMaxLevel_ == -2 )
      {
         const char* e  = std::getenv( "HYTEG_AMD_BATCH_MAX_LEVEL" );
         batchMaxLevel_ = e ? std::atoi( e ) : 6;
         const char* s  = std::getenv( "HYTEG_AMD_BATCH_SINGLE_MAX_LEVEL" );
         batchSingleMaxLevel_ = s ? std::atoi( s ) : kBatchSingleMaxLevelDefault;
      }
      if ( localCells_.size() == 1 )
         return (int) level <= std::min( batchMaxLevel_, batchSingleMaxLevel_ );
      return localCells_.size() > 1 && (int) level <= batchMaxLevel_;
   }
   // the one-workgroup Gauss-Seidel sweep of small cells (levels <= 5) also pays off for a single cell: 1 launch instead of ~3n
   // The macro-cell sweep of SOR / Gauss-Seidel: also above the batch level the cells of a rank share their launches -- a cell's
   // sweep is 46 dependent launches at level 8 with at most 36 workgroups each, so eight cells one after the other leave the GPU
   // idle eight times as long as eight cells per launch (hyteg_hip_p1_sor_cells: grid.y = cell).  HYTEG_AMD_BATCH_SOR_ALL=0: only
   // up to the batch level, as for the other kernels.
   bool useBatchSor( uint_t level ) const
   {
      // the one-workgroup sweep of small cells: up to level 5 for several cells (one workgroup each, all at once), up to level 4
      // for a single cell (at level 5 the blocked sweep is faster there: 65 against 120 us)
      const uint_t smallMax = localCells_.size() == 1 ? 4 : 5;
      if ( useBatch( level ) || ( !localCells_.empty() && level <= smallMax && batchMaxLevel_ >= 0 && (int) level <= batchMaxLevel_ ) )
         return true;
      static const bool all = [] {
         const char* e = std::getenv( "HYTEG_AMD_BATCH_SOR_ALL" );
         return !( e && e[0] == '0' );
      }();
      useBatch( level ); // reads HYTEG_AMD_BATCH_MAX_LEVEL on first use
      return all && localCells_.size() > 1 && batchMaxLevel_ >= 0 && level <= 10;
   }
   void setBatchMaxLevel( int l ) { batchMaxLevel_ = l; }
   std::vector< unsigned > masksFor( DoFType flag, bool owned = false, unsigned keep = HYTEG_HIP_MASK_ALL ) const
   {
      std::vector< unsigned > m;
      for ( int id : localCells_ )
         m.push_back( ( owned ? ownedMaskFor( cells_[id], flag ) : maskFor( cells_[id], flag ) ) & keep );
      return m;
   }
   // device table [local cell][14] of 1 / numNeighborCells (grid transfer)
   const double* nncInvDevice() const
   {
      if ( !nncInv_ && !localCells_.empty() )
      {
         std::vector< double > h;
         for ( int id : localCells_ )
            for ( double n : numNeighborCells( cells_[id] ) )
               h.push_back( 1.0 / n );
         nncInv_ = uploadTable( h );
      }
      return nncInv_;
   }
   // small read-only device table owned by the storage (freed with it)
   void* uploadBytes( const void* h, size_t bytes ) const
   {
      void* p = nullptr;
      hipCheck( hyteg_hip_malloc( &p, std::max< size_t >( 8, bytes ) ), "uploadTable: malloc" );
      hipCheck( hyteg_hip_upload( p, h, bytes, stream_ ), "uploadTable: upload" );
      hipCheck( hyteg_hip_stream_synchronize( stream_ ), "uploadTable: sync" );
      scratchAll_.push_back( p );
      return p;
   }
   double* uploadTable( const std::vector< double >& h ) const
   {
      return static_cast< double* >( uploadBytes( h.data(), h.size() * sizeof( double ) ) );
   }
   // calls fn( first, count ) for chunks of at most HYTEG_HIP_MAX_BATCH local cells
   template < typename F >
   void forCellChunks( F&& fn ) const
   {
      const int n = (int) localCells_.size();
      for ( int first = 0; first < n; first += HYTEG_HIP_MAX_BATCH )
         fn( first, std::min( HYTEG_HIP_MAX_BATCH, n - first ) );
   }

   void              setStream( hyteg_hip_stream_t s )
   {
      stream_ = s;
      if ( timingTree_ )
         timingTree_->setStream( s );
   }
   // PrimitiveStorage::getTimingTree(): null unless enabled (a disabled tree costs one pointer test per range).
   // synchronize: every range waits for the stream before it stops, so that it measures execution, not enqueueing.
   void enableTiming( bool on, bool synchronize = false )
   {
      if ( !on )
      {
         timingTree_.reset();
         return;
      }
      if ( !timingTree_ )
         timingTree_ = std::make_shared< TimingTree >();
      timingTree_->setSynchronize( synchronize, stream_ );
   }
   TimingTree* getTimingTree() const { return timingTree_.get(); }
   hyteg_hip_stream_t stream() const { return stream_; }

   // ---- side stream for the shared-point chain of an operator application -------------------------------------------
   // In apply() the chain boundary shares -> pack -> [exchange] -> reduce touches only shared points of dst and the interior
   // kernel only interior points.  With a transport that orders the exchange against the stream it is called on (RCCL, peer
   // to peer) the chain runs on a second stream next to the interior kernel: SideChain forks it off the storage's stream
   // (event), toMain() / toSide() select where the next launches go, join() makes the storage's stream wait for the chain.
   // OFF by default (HYTEG_AMD_SIDE_STREAM=1 enables it; =2 also on one rank, to measure): the two cross-stream event
   // dependencies per apply cost more than the overlap gains -- two level-8 cells on one rank 39.4 -> 49.3 us per apply,
   // two ranks sharing a GPU 39.7 -> 150-170 us (profiles/r02_p2p_probe.txt).
   bool sideChainUsable( int level, DoFType flag, int dofKind ) const
   {
      if ( sideEnabled_ < 0 )
      {
         const char* e = std::getenv( "HYTEG_AMD_SIDE_STREAM" );
         sideEnabled_  = e ? std::atoi( e ) : 0;
      }
      if ( sideEnabled_ == 2 && !mainStream_ ) // measurement aid: also on one rank (local shared points only)
         return true;
      if ( nranks_ == 1 || !transport_ )
         return false;
      if ( !sideEnabled_ || mainStream_ )
         return false;
      bool any = false;
      for ( int cls = 0; cls < 2; ++cls )
         if ( testFlag( boundaryTypeOf( cls == 1 ), flag ) && !exchangePlan( level, cls, dofKind ).peers.empty() )
         {
            if ( !transport_->anyStream( level, cls + 2 * dofKind ) )
               return false;
            any = true;
         }
      return any;
   }
   class SideChain
   {
    public:
      SideChain( const PrimitiveStorage& st, bool use )
      : st_( st )
      , open_( use )
      {
         if ( !open_ )
            return;
         if ( !st_.sideStream_ )
         {
            hipCheck( hyteg_hip_stream_create( &st_.sideStream_ ), "side chain: stream_create" );
            hipCheck( hyteg_hip_event_create( &st_.forked_ ), "side chain: event_create" );
            hipCheck( hyteg_hip_event_create( &st_.joined_ ), "side chain: event_create" );
         }
         hipCheck( hyteg_hip_event_record( st_.forked_, st_.stream_ ), "side chain: record" );
         hipCheck( hyteg_hip_stream_wait_event( st_.sideStream_, st_.forked_ ), "side chain: wait" );
         st_.mainStream_ = st_.stream_;
         st_.stream_     = st_.sideStream_;
      }
      void toMain() const
      {
         if ( open_ )
            st_.stream_ = st_.mainStream_;
      }
      void toSide() const
      {
         if ( open_ )
            st_.stream_ = st_.sideStream_;
      }
      void join()
      {
         if ( !open_ )
            return;
         open_                  = false;
         hyteg_hip_stream_t main = st_.mainStream_;
         st_.stream_             = main;
         st_.mainStream_         = nullptr;
         hipCheck( hyteg_hip_event_record( st_.joined_, st_.sideStream_ ), "side chain: record" );
         hipCheck( hyteg_hip_stream_wait_event( main, st_.joined_ ), "side chain: wait" );
      }
      ~SideChain()
      {
         if ( open_ ) // left by an exception: back to the storage's stream, which then waits for whatever the chain holds
         {
            st_.stream_     = st_.mainStream_;
            st_.mainStream_ = nullptr;
            if ( hyteg_hip_event_record( st_.joined_, st_.sideStream_ ) == HYTEG_HIP_OK )
               hyteg_hip_stream_wait_event( st_.stream_, st_.joined_ );
         }
      }

    private:
      const PrimitiveStorage& st_;
      bool                    open_;
   };
   // ---- transport of the shared-point exchange (storages distributed over several ranks) ----
   void setCommHooks( const CommHooks& h ) { transport_ = std::make_shared< HookTransport >( h ); }
   void setTransport( std::shared_ptr< Transport > t ) { transport_ = std::move( t ); }
   // RCCL over xGMI issued from this layer; uniqueId: HYTEG_HIP_COMM_ID_BYTES bytes created by rank 0
   // (hyteg_hip_comm_unique_id) and distributed by the application.  Collective over all ranks.
   void       useRccl( const unsigned char* uniqueId ) { transport_ = std::make_shared< RcclTransport >( nranks_, rank_, uniqueId ); }
   // Peer-to-peer transport on top of the current one (which keeps the all-reduce and every plan that is not connected):
   // see P2PTransport in comm.hpp for the set-up sequence.
   P2PTransport& useP2P( size_t arenaBytes )
   {
      auto p     = std::make_shared< P2PTransport >( nranks_, rank_, arenaBytes, transport_ );
      transport_ = p;
      return *p;
   }
   P2PTransport& p2p() const
   {
      auto* p = dynamic_cast< P2PTransport* >( transport_.get() );
      if ( !p )
         throw std::runtime_error( "PrimitiveStorage: the transport is not the peer-to-peer one" );
      return *p;
   }
   // back to the transport the peer-to-peer one wraps (used when its self-check fails on some rank)
   void dropP2P()
   {
      if ( auto* p = dynamic_cast< P2PTransport* >( transport_.get() ) )
      {
         std::shared_ptr< Transport > inner = p->innerShared();
         transport_                         = inner;
      }
   }
   void       checkTransport() const
   {
      if ( transport_ )
         transport_->check( stream_ );
   }
   // at the points where the host reads device results anyway (downloads of cell arrays): only the peer-to-peer transport has
   // something to check there (the status word of its device-side arrival waits)
   void checkTransportAtHostRead() const
   {
      if ( transport_ && std::string( transport_->name() ) == "p2p" )
         transport_->check( stream_ );
   }
   Transport* transport() const { return transport_.get(); }
   Transport& requireTransport( const char* what ) const
   {
      if ( !transport_ )
         throw std::runtime_error( std::string( what ) + ": storage is distributed but no transport (RCCL or hooks) is set" );
      return *transport_;
   }
   double allreduceSum( double v, const char* what ) const
   {
      if ( nranks_ > 1 )
         requireTransport( what ).allreduceSum( &v, 1 );
      return v;
   }

   // Shared-point exchange of the arrays `arrays` (one device array per local cell: the vertex-DoF or the edge-DoF arrays
   // of one function at `level`): pack the partial values other ranks need + start the transfer / wait + reduce local and
   // received copies in a fixed order (additive: every copy := sum of all copies, VertexDoFAdditivePackInfo.hpp:676-745;
   // otherwise every copy := the copy of the lowest-numbered neighbour cell).  Kernels that do not touch shared points
   // may be enqueued between Begin and End: they overlap the transfer.
   void sharedExchangeBegin( const std::vector< double* >& arrays, int level, DoFType flag, int dofKind ) const
   {
      if ( nranks_ == 1 )
         return;
      Transport& T = requireTransport( "exchange" );
      for ( int cls = 0; cls < 2; ++cls )
      {
         if ( !testFlag( boundaryTypeOf( cls == 1 ), flag ) )
            continue;
         const ExchangePlan& host = exchangePlan( level, cls, dofKind );
         if ( host.peers.empty() && !T.collective() )
            continue;
         const ExchangePlan& plan = devicePlan( level, cls, dofKind );
         const int key = cls + 2 * dofKind;
         if ( !plan.peers.empty() )
         {
            double** bases = basesTable( arrays, plan, plan.recvBuffer );
            if ( !T.pack( plan, level, key, bases, stream_ ) )
               hipCheck( hyteg_hip_gather_entries( plan.sendBuffer, bases, plan.dSendBuf, plan.dSendOff, plan.totalSend(), stream_ ),
                         "exchange: pack" );
         }
         T.exchangeBegin( plan, level, key, stream_ );
      }
   }
   // sharedExchangeBegin for a rank with ONE macro-cell whose boundary-share kernel delivers the shares itself
   // (hyteg_hip_p1_apply_cell_boundary_p2p).  Possible if exactly one boundary class under `flag` has peers and the transport
   // hands out pack arguments (peer to peer).  true: the exchange has begun -- the caller MUST launch that kernel with `out`;
   // false: nothing has happened, use the boundary kernel + sharedExchangeBegin.  Taken for exchanges of at least
   // HYTEG_AMD_SHARE_SEND_MIN values (default 60000; 0 = never): with loop-back peers on one GPU an apply with three shared
   // level-8 macro-faces (97k values) takes 29.9 instead of 33.6 us, one with a single face (32k) 22.7 instead of 21.0 --
   // every workgroup of the share kernel then waits for write-through stores (profiles/r02_p2p_probe.txt).
   struct ShareSend
   {
      PackArgs   a;
      const int *first = nullptr, *list = nullptr; // CSR over the shell enumeration of the boundary kernel
   };
   bool sharedExchangeBeginByShares( int level, DoFType flag, ShareSend& out ) const
   {
      if ( nranks_ == 1 || !transport_ || localCells_.size() != 1 )
         return false;
      static const int minValues = [] {
         const char* e = std::getenv( "HYTEG_AMD_SHARE_SEND_MIN" );
         return e ? std::atoi( e ) : 60000;
      }();
      if ( minValues <= 0 )
         return false;
      int active = -1;
      for ( int cls = 0; cls < 2; ++cls )
         if ( testFlag( boundaryTypeOf( cls == 1 ), flag ) && !exchangePlan( level, cls, 0 ).peers.empty() )
         {
            if ( active >= 0 )
               return false;
            active = cls;
         }
      if ( active < 0 || exchangePlan( level, active, 0 ).totalSend() < minValues )
         return false;
      const ExchangePlan& plan = devicePlan( level, active, 0 );
      const auto&         tbl  = shareSendTable( level, active, plan );
      if ( !transport_->packArgs( plan, level, active, out.a, stream_ ) )
         return false;
      out.first = tbl.first, out.list = tbl.list;
      // as sharedExchangeBegin: a collective transport is entered for the class without peers as well
      for ( int cls = 0; cls < 2; ++cls )
      {
         if ( !testFlag( boundaryTypeOf( cls == 1 ), flag ) )
            continue;
         if ( cls == active || transport_->collective() )
            transport_->exchangeBegin( devicePlan( level, cls, 0 ), level, cls, stream_ );
      }
      return true;
   }
   void sharedExchangeEnd( const std::vector< double* >& arrays, int level, DoFType flag, int dofKind, bool additive ) const
   {
      for ( int cls = 0; cls < 2; ++cls )
      {
         if ( !testFlag( boundaryTypeOf( cls == 1 ), flag ) )
            continue;
         const ExchangePlan& host = exchangePlan( level, cls, dofKind );
         const int           key  = cls + 2 * dofKind;
         if ( nranks_ > 1 && ( !host.peers.empty() || requireTransport( "exchange" ).collective() ) )
            transport_->exchangeEnd( devicePlan( level, cls, dofKind ), level, key, stream_ );
         if ( host.ngroups() == 0 )
            continue;
         const ExchangePlan& plan  = devicePlan( level, cls, dofKind );
         const bool          remote = nranks_ > 1 && !plan.peers.empty();
         double**            bases  = basesTable( arrays, plan, remote ? transport_->recvBase( plan, level, key ) : plan.recvBuffer );
         ArrivalWait         w;
         if ( remote && transport_->arrivalWait( plan, level, key, w ) )
            hipCheck( hyteg_hip_reduce_shared_after_p2p( bases, plan.dGroupPtr, plan.dEntryBuf, plan.dEntryOff, plan.ngroups(), (int) arrays.size(),
                                                         additive ? 1 : 0, w.flags, w.npeers, w.stride, w.seq, w.status, w.timeoutMs, stream_ ),
                      "exchange: wait + reduce" );
         else
            hipCheck( additive ? hyteg_hip_sum_shared( bases, plan.dGroupPtr, plan.dEntryBuf, plan.dEntryOff, plan.ngroups(),
                                                       (int) arrays.size(), stream_ )
                               : hyteg_hip_copy_shared( bases, plan.dGroupPtr, plan.dEntryBuf, plan.dEntryOff, plan.ngroups(),
                                                        (int) arrays.size(), stream_ ),
                      "exchange: reduce" );
      }
   }
   // For the ONE local cell of this rank: which entries of the plan's send enumeration the shell point q of the boundary kernel
   // (p1_boundary.hip shell_point: q = face * tri( N ) + row_start( N, j ) + i, a point on several faces through its lowest one)
   // feeds, as CSR arrays on the device.
   struct ShareSendTable
   {
      int *first = nullptr, *list = nullptr;
   };
   const ShareSendTable& shareSendTable( int level, int cls, const ExchangePlan& plan ) const
   {
      auto it = shareSendTables_.find( { level, cls } );
      if ( it != shareSendTables_.end() )
         return it->second;
      const int64_t N = layout::width( level ), T = N * ( N + 1 ) / 2;
      std::multimap< int, int > byOffset; // array offset -> send entry
      for ( int k = 0; k < plan.totalSend(); ++k )
      {
         if ( plan.sendBuf[(size_t) k] != 0 )
            throw std::runtime_error( "shareSendTable: more than one local cell" );
         byOffset.emplace( plan.sendOff[(size_t) k], k );
      }
      std::vector< int > first( (size_t) ( 4 * T + 1 ), 0 ), list;
      for ( int64_t f = 0; f < 4; ++f )
         for ( int64_t j = 0; j < N; ++j )
            for ( int64_t i = 0; i < N - j; ++i )
            {
               const int64_t q = f * T + ( j * N - j * ( j - 1 ) / 2 ) + i;
               const int64_t x = f == 2 ? 0 : i, y = f == 0 || f == 3 ? j : ( f == 1 ? 0 : i ), z = f == 0 ? 0 : ( f == 3 ? N - 1 - i - j : j );
               const int64_t lowest = z == 0 ? 0 : ( y == 0 ? 1 : ( x == 0 ? 2 : 3 ) );
               if ( lowest == f )
               {
                  auto range = byOffset.equal_range( (int) layout::cellIndex( N, x, y, z ) );
                  for ( auto e = range.first; e != range.second; ++e )
                     list.push_back( e->second );
               }
               first[(size_t) q + 1] = (int) list.size();
            }
      // rows are visited in increasing q, so first[] is already cumulative -- but q runs face by face: make sure of it
      for ( size_t q = 1; q < first.size(); ++q )
         if ( first[q] < first[q - 1] )
            throw std::runtime_error( "shareSendTable: enumeration out of order" );
      if ( (int) list.size() != plan.totalSend() )
         throw std::runtime_error( "shareSendTable: " + std::to_string( plan.totalSend() - (int) list.size() ) +
                                   " send entries are not shell points of the cell" );
      ShareSendTable t;
      t.first = uploadVector( first );
      list.push_back( 0 ); // never empty
      t.list  = uploadVector( list );
      return shareSendTables_.emplace( std::make_pair( level, cls ), t ).first->second;
   }
   // device table [ local cell arrays ..., receive segment of peer 0, peer 1, ... ] (cached by content)
   double** basesTable( const std::vector< double* >& arrays, const ExchangePlan& plan, double* recvBase ) const
   {
      std::vector< double* > host( arrays );
      double*                seg = recvBase;
      for ( uint_t s = 0; s < plan.peers.size(); ++s )
      {
         host.push_back( seg );
         seg += plan.recvCount[s];
      }
      return pointerTable( host );
   }

   // pool of scratch device arrays keyed by size, so that operators can use temporaries without hipMalloc/hipFree
   // in the hot path (the role of hyteg::getTemporaryFunction, src/hyteg/memory/TempFunctionManager.hpp)
   double* acquireScratch( size_t doubles ) const
   {
      auto& freeList = scratchFree_[doubles];
      if ( !freeList.empty() )
      {
         double* p = freeList.back();
         freeList.pop_back();
         return p;
      }
      void* p = nullptr;
      hipCheck( hyteg_hip_malloc( &p, doubles * sizeof( double ) ), "scratch: malloc" );
      scratchAll_.push_back( p );
      return static_cast< double* >( p );
   }
   void releaseScratch( size_t doubles, double* p ) const { scratchFree_[doubles].push_back( p ); }

   // device copy of a list of device pointers (the "bases" argument of the exchange kernels), cached by content: scratch
   // functions get the same arrays from the pool again and again, so after the first cycle nothing is allocated or
   // uploaded in the hot path (and the path can be recorded into a launch graph)
   double** pointerTable( const std::vector< double* >& host ) const
   {
      auto it = pointerTables_.find( host );
      if ( it != pointerTables_.end() )
         return it->second;
      void* d = nullptr;
      hipCheck( hyteg_hip_malloc( &d, std::max< size_t >( 1, host.size() ) * sizeof( double* ) ), "bases: malloc" );
      // on the null stream and complete on return: valid for whatever stream uses the table next, and legal while the
      // storage's stream is being recorded into a launch graph
      hipCheck( hyteg_hip_upload( d, host.data(), host.size() * sizeof( double* ), nullptr ), "bases: upload" );
      hipCheck( hyteg_hip_stream_synchronize( nullptr ), "bases: sync" );
      scratchAll_.push_back( d );
      pointerTables_[host] = static_cast< double** >( d );
      return static_cast< double** >( d );
   }

   double* dotResult() const
   {
      if ( !dotResult_ )
      {
         hipCheck( hyteg_hip_malloc( &dotResult_, sizeof( double ) * std::max< size_t >( 1, localCells_.size() ) ), "PrimitiveStorage: malloc" );
         hipCheck( hyteg_hip_malloc( &dotWorkspace_, hyteg_hip_dot_workspace_bytes() ), "PrimitiveStorage: malloc" );
      }
      return static_cast< double* >( dotResult_ );
   }
   void* dotWorkspace() const
   {
      dotResult();
      return dotWorkspace_;
   }

   // ---------------------------------------------------------------------------------------------------
   // Additive exchange plan of one (level, boundary class): every DoF on a macro-face/edge/vertex with at least
   // two neighbour cells, at least one of them local, is a group; its entries are its copies in ascending global
   // cell order.  cls 0: primitives in the interior of the domain, cls 1: primitives on the domain boundary.
   // ---------------------------------------------------------------------------------------------------
   using ExchangePlan = hyteg::ExchangePlan;

   // host part of the plan (no GPU needed)
   // dofKind 0: vertex DoFs (P1 arrays); 1: edge DoFs (the edge-DoF arrays of P2 functions)
   const ExchangePlan& exchangePlan( int level, int cls, int dofKind = 0 ) const
   {
      auto key = std::make_pair( level, cls + 2 * dofKind );
      auto it  = plans_.find( key );
      if ( it == plans_.end() )
         it = plans_.emplace( key, buildPlan( level, cls, dofKind ) ).first;
      return it->second;
   }
   // plan with its index arrays (and default communication buffers) resident on the device
   const ExchangePlan& devicePlan( int level, int cls, int dofKind = 0 ) const
   {
      auto& P = const_cast< ExchangePlan& >( exchangePlan( level, cls, dofKind ) );
      if ( !P.onDevice )
      {
         P.dGroupPtr = uploadVector( P.groupPtr );
         P.dEntryBuf = uploadVector( P.entryBuf );
         P.dEntryOff = uploadVector( P.entryOff );
         P.dSendBuf  = uploadVector( P.sendBuf );
         P.dSendOff  = uploadVector( P.sendOff );
         if ( !P.sendBuffer && ( P.totalSend() > 0 || P.totalRecv() > 0 ) )
         {
            void *s = nullptr, *r = nullptr;
            hipCheck( hyteg_hip_malloc( &s, std::max( 1, P.totalSend() ) * sizeof( double ) ), "plan: malloc" );
            hipCheck( hyteg_hip_malloc( &r, std::max( 1, P.totalRecv() ) * sizeof( double ) ), "plan: malloc" );
            P.sendBuffer  = static_cast< double* >( s );
            P.recvBuffer  = static_cast< double* >( r );
            P.ownsBuffers = true;
         }
         P.onDevice = true;
      }
      return P;
   }
   // multi-rank: the application owns the communication buffers (e.g. torch tensors) and registers them here
   void registerCommBuffers( int level, int key, double* send, double* recv ) const
   {
      auto& p = const_cast< ExchangePlan& >( exchangePlan( level, key & 1, key >> 1 ) ); // key = cls + 2 * dofKind
      if ( p.ownsBuffers )
      {
         hyteg_hip_free( p.sendBuffer );
         hyteg_hip_free( p.recvBuffer );
         p.ownsBuffers = false;
      }
      p.sendBuffer = send;
      p.recvBuffer = recv;
   }

 private:
   template < typename T >
   static T* uploadVector( const std::vector< T >& v )
   {
      if ( v.empty() )
         return nullptr;
      void* d = nullptr;
      hipCheck( hyteg_hip_malloc( &d, v.size() * sizeof( T ) ), "upload: malloc" );
      hipCheck( hyteg_hip_upload( d, v.data(), v.size() * sizeof( T ), nullptr ), "upload: copy" );
      hipCheck( hyteg_hip_stream_synchronize( nullptr ), "upload: sync" );
      return static_cast< T* >( d );
   }

   // array index inside cell `c` of the point with barycentric weights w[k] on the primitive's vertices p.v[k]
   static int64_t indexInCell( const MacroCell& c, const MacroPrimitive& p, const int* w, int64_t N )
   {
      int64_t bary[4] = { 0, 0, 0, 0 };
      for ( uint_t k = 0; k < p.v.size(); ++k )
      {
         int l = -1;
         for ( int q = 0; q < 4; ++q )
            if ( c.v[q] == p.v[k] )
               l = q;
         if ( l < 0 )
            throw std::runtime_error( "indexInCell: primitive is not part of the cell" );
         bary[l] = w[k];
      }
      return layout::cellIndex( N, bary[1], bary[2], bary[3] );
   }

   // array index in the edge-DoF array of cell `c` of the edge DoF between the points with barycentric weights wa, wb on the
   // primitive's vertices (edgedof::calcEdgeDoFIndex / calcEdgeDoFOrientation, EdgeDoFIndexing.hpp:89-165, + macrocell::index)
   static int64_t edgeIndexInCell( const MacroCell& c, const MacroPrimitive& p, const int* wa, const int* wb, int level )
   {
      int64_t a[4] = { 0, 0, 0, 0 }, b[4] = { 0, 0, 0, 0 };
      for ( uint_t k = 0; k < p.v.size(); ++k )
      {
         int l = -1;
         for ( int q = 0; q < 4; ++q )
            if ( c.v[q] == p.v[k] )
               l = q;
         if ( l < 0 )
            throw std::runtime_error( "edgeIndexInCell: primitive is not part of the cell" );
         a[l] = wa[k], b[l] = wb[k];
      }
      const int64_t* A  = a + 1; // (x, y, z) = weights of cell vertices 1, 2, 3
      const int64_t* B  = b + 1;
      const int64_t  d0 = B[0] - A[0], d1 = B[1] - A[1], d2 = B[2] - A[2];
      const int64_t  n  = int64_t( 1 ) << level;
      int            o;
      int64_t        e[3];
      auto           lower = [&]( int axis ) { return A[axis] < B[axis] ? A : B; };
      if ( d1 == 0 && d2 == 0 )
         o = 0, e[0] = lower( 0 )[0], e[1] = lower( 0 )[1], e[2] = lower( 0 )[2];
      else if ( d0 == 0 && d2 == 0 )
         o = 1, e[0] = lower( 1 )[0], e[1] = lower( 1 )[1], e[2] = lower( 1 )[2];
      else if ( d0 == 0 && d1 == 0 )
         o = 2, e[0] = lower( 2 )[0], e[1] = lower( 2 )[1], e[2] = lower( 2 )[2];
      else if ( d2 == 0 )
         o = 3, e[0] = lower( 0 )[0], e[1] = lower( 0 )[1] - 1, e[2] = lower( 0 )[2];
      else if ( d1 == 0 )
         o = 4, e[0] = lower( 0 )[0], e[1] = lower( 0 )[1], e[2] = lower( 0 )[2] - 1;
      else if ( d0 == 0 )
         o = 5, e[0] = lower( 1 )[0], e[1] = lower( 1 )[1], e[2] = lower( 1 )[2] - 1;
      else
         o = 6, e[0] = lower( 0 )[0], e[1] = lower( 0 )[1] - 1, e[2] = lower( 0 )[2];
      return o * layout::tet( n ) + layout::cellIndex( o == 6 ? n - 1 : n, e[0], e[1], e[2] );
   }

   ExchangePlan buildPlan( int level, int cls, int dofKind = 0 ) const
   {
      ExchangePlan  P;
      const int64_t N = layout::width( level ), n = N - 1;
      const int     nLocal = (int) localCells_.size();
      // peers: ranks of remote cells sharing a primitive of this class with a local cell
      std::set< int > peerSet;
      auto            involves = [&]( const MacroPrimitive& p, bool& local ) {
         local = false;
         if ( p.cells.size() < 2 || ( p.onBoundary ? 1 : 0 ) != cls )
            return false;
         for ( int c : p.cells )
            local = local || cells_[c].rank == rank_;
         return true;
      };
      auto forAllPrimitives = [&]( auto&& fn ) {
         for ( const auto& p : faces_ )
            fn( p );
         for ( const auto& p : edges_ )
            fn( p );
         for ( const auto& p : vertices_ )
            fn( p );
      };
      forAllPrimitives( [&]( const MacroPrimitive& p ) {
         bool local;
         if ( involves( p, local ) && local )
            for ( int c : p.cells )
               if ( cells_[c].rank != rank_ )
                  peerSet.insert( cells_[c].rank );
      } );
      P.peers.assign( peerSet.begin(), peerSet.end() );
      std::map< int, int > peerSlot;
      for ( uint_t i = 0; i < P.peers.size(); ++i )
         peerSlot[P.peers[i]] = (int) i;
      P.sendCount.assign( P.peers.size(), 0 );
      P.recvCount.assign( P.peers.size(), 0 );
      std::vector< std::vector< int > > sendBufPer( P.peers.size() ), sendOffPer( P.peers.size() );

      // enumerate the DoFs that belong to a primitive in a rank-independent order: fn( wa, wb ) with the barycentric
      // weights of the point (vertex DoF, wb unused) or of the two end points of the micro-edge (edge DoF)
      auto pointsOf = [&]( const MacroPrimitive& p, auto&& fn ) {
         if ( dofKind == 1 )
         {
            if ( p.v.size() == 3 )
            {
               // micro-edges in the plane of the face whose end points do not lie on one and the same macro-edge of the face
               auto emit = [&]( int64_t i0, int64_t j0, int64_t i1, int64_t j1 ) {
                  const int wa[3] = { (int) ( n - i0 - j0 ), (int) i0, (int) j0 }, wb[3] = { (int) ( n - i1 - j1 ), (int) i1, (int) j1 };
                  for ( int k = 0; k < 3; ++k )
                     if ( wa[k] == 0 && wb[k] == 0 )
                        return;
                  fn( wa, wb );
               };
               for ( int64_t j = 0; j <= n; ++j )
                  for ( int64_t i = 0; i + j <= n; ++i )
                  {
                     if ( i + j + 1 <= n )
                     {
                        emit( i, j, i + 1, j );
                        emit( i, j, i, j + 1 );
                        emit( i + 1, j, i, j + 1 );
                     }
                  }
            }
            else if ( p.v.size() == 2 )
            {
               for ( int64_t i = 0; i <= n - 1; ++i )
               {
                  const int wa[2] = { (int) ( n - i ), (int) i }, wb[2] = { (int) ( n - i - 1 ), (int) ( i + 1 ) };
                  fn( wa, wb );
               }
            }
            return;
         }
         if ( p.v.size() == 3 )
         {
            for ( int64_t j = 1; j <= n - 2; ++j )
               for ( int64_t i = 1; i + j <= n - 1; ++i )
               {
                  const int w[3] = { (int) ( n - i - j ), (int) i, (int) j };
                  fn( w, w );
               }
         }
         else if ( p.v.size() == 2 )
         {
            for ( int64_t i = 1; i <= n - 1; ++i )
            {
               const int w[2] = { (int) ( n - i ), (int) i };
               fn( w, w );
            }
         }
         else
         {
            const int w[1] = { (int) n };
            fn( w, w );
         }
      };

      // first pass: receive offsets.  The data a peer sends us is ordered by (primitive, point, entry) over all
      // groups that involve both ranks -- the same loop the peer runs to fill its send buffer.
      std::vector< int > recvCursor( P.peers.size(), 0 );
      P.groupPtr.push_back( 0 );
      forAllPrimitives( [&]( const MacroPrimitive& p ) {
         bool local;
         if ( !involves( p, local ) || !local )
            return;
         pointsOf( p, [&]( const int* w, const int* wb ) {
            for ( int c : p.cells )
            {
               const MacroCell& cell = cells_[c];
               if ( cell.rank == rank_ )
               {
                  const int off = dofKind == 1 ? (int) edgeIndexInCell( cell, p, w, wb, level ) : (int) indexInCell( cell, p, w, N );
                  P.entryBuf.push_back( cell.localIndex );
                  P.entryOff.push_back( off );
                  // this value goes to every peer that shares the group
                  std::set< int > dests;
                  for ( int c2 : p.cells )
                     if ( cells_[c2].rank != rank_ )
                        dests.insert( cells_[c2].rank );
                  for ( int d : dests )
                  {
                     sendBufPer[peerSlot[d]].push_back( cell.localIndex );
                     sendOffPer[peerSlot[d]].push_back( off );
                  }
               }
               else
               {
                  const int s = peerSlot[cell.rank];
                  P.entryBuf.push_back( nLocal + s );
                  P.entryOff.push_back( recvCursor[s]++ );
               }
            }
            P.groupPtr.push_back( (int) P.entryBuf.size() );
         } );
      } );
      // receive buffer = concatenation over peers: turn per-peer offsets into offsets relative to the peer's segment;
      // bases[nLocal + s] points at the start of peer s's segment, so the offsets stay as they are.
      for ( uint_t s = 0; s < P.peers.size(); ++s )
      {
         P.recvCount[s] = recvCursor[s];
         P.sendCount[s] = (int) sendBufPer[s].size();
         P.sendBuf.insert( P.sendBuf.end(), sendBufPer[s].begin(), sendBufPer[s].end() );
         P.sendOff.insert( P.sendOff.end(), sendOffPer[s].begin(), sendOffPer[s].end() );
      }
      return P;
   }

   int                                                     rank_, nranks_;
   std::vector< MacroCell >                                cells_;
   std::vector< MacroPrimitive >                           faces_, edges_, vertices_;
   std::vector< int >                                      localCells_;
   DoFType                                                 boundaryType_ = DirichletBoundary;
   mutable hyteg_hip_stream_t                              stream_       = nullptr;
   mutable hyteg_hip_stream_t                              sideStream_ = nullptr, mainStream_ = nullptr;
   mutable hyteg_hip_event_t                               forked_ = nullptr, joined_ = nullptr;
   mutable int                                             sideEnabled_ = -1;
   std::shared_ptr< Transport >                            transport_;
   std::shared_ptr< TimingTree >                           timingTree_;
   mutable void *                                          dotResult_ = nullptr, *dotWorkspace_ = nullptr;
   mutable std::map< size_t, std::vector< double* > >      scratchFree_;
   mutable std::vector< void* >                            scratchAll_;
   mutable std::map< std::vector< double* >, double** >    pointerTables_;
   mutable std::map< std::pair< int, int >, ShareSendTable > shareSendTables_;
   mutable double*                                         nncInv_        = nullptr;
   mutable int                                             batchMaxLevel_ = -2; // -2: read HYTEG_AMD_BATCH_MAX_LEVEL on first use
   // a rank with ONE macro-cell: levels up to this one use the generic batched kernels as well
   static constexpr int                                    kBatchSingleMaxLevelDefault = -1;
   mutable int                                             batchSingleMaxLevel_        = kBatchSingleMaxLevelDefault;
   mutable std::map< std::pair< int, int >, ExchangePlan > plans_;
};

} // namespace hyteg
