// p2function.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// P2Function< double > = vertex DoFs + edge DoFs (src/hyteg/p2functionspace/P2Function.hpp)
#pragma once

#include "p1function.hpp"

namespace hyteg {

// =====================================================================================================
// P2Function< double >  ( src/hyteg/p2functionspace/P2Function.hpp ): a VertexDoF function plus an EdgeDoF function.
// First version: storages with ONE macro-cell (the shared edge DoFs of several cells need their own
// exchange plans, which do not exist yet).  Edge-DoF arrays: layout of edgedofspace/EdgeDoFIndexing.hpp:920-985.
// =====================================================================================================
template < typename ValueType >
class P2Function
{
 public:
   using valueType = ValueType;
   P2Function( const std::string& name, const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : name_( name )
   , storage_( storage )
   , minLevel_( minLevel )
   , maxLevel_( maxLevel )
   , vertexDoFFunction_( name + "_VertexDoF", storage, minLevel, maxLevel )
   {
      if ( maxLevel > HYTEG_HIP_P2_MAX_LEVEL )
         throw std::runtime_error( "P2Function: level out of range" );
      edge_.resize( storage->getNumberOfLocalCells() );
      for ( auto& perCell : edge_ )
         for ( uint_t l = minLevel; l <= maxLevel; ++l )
         {
            const size_t bytes = std::max< size_t >( 1, hyteg_hip_p2_edge_array_size( (int) l ) ) * sizeof( double );
            void*        q     = nullptr;
            hipCheck( hyteg_hip_malloc( &q, bytes ), "P2Function: malloc" );
            hipCheck( hyteg_hip_memset_zero( q, bytes, storage->stream() ), "P2Function: memset" );
            perCell.push_back( static_cast< double* >( q ) );
         }
   }
   ~P2Function()
   {
      for ( auto& perCell : edge_ )
         for ( double* q : perCell )
            hyteg_hip_free( q );
   }
   P2Function( const P2Function& )            = delete;
   P2Function& operator=( const P2Function& ) = delete;

   const P1Function< ValueType >&      getVertexDoFFunction() const { return vertexDoFFunction_; }
   std::shared_ptr< PrimitiveStorage > getStorage() const { return storage_; }
   // device pointer of the edge-DoF array of local cell c
   double* getEdgeCellPointer( uint_t c, uint_t level ) const
   {
      if ( c >= edge_.size() || level < minLevel_ || level > maxLevel_ )
         throw std::runtime_error( "P2Function '" + name_ + "': bad cell or level" );
      return edge_[c][level - minLevel_];
   }
   uint_t getNumberOfEdgeDoFs( uint_t level ) const { return hyteg_hip_p2_edge_array_size( (int) level ); }

   void interpolate( ValueType constant, uint_t level, DoFType flag = All ) const
   {
      vertexDoFFunction_.interpolate( constant, level, flag );
      if ( storage_->useBatch( level ) )
      {
         const auto masks = storage_->masksFor( flag );
         storage_->forCellChunks( [&]( int first, int count ) {
            std::vector< double* > dst;
            for ( int c = first; c < first + count; ++c )
               dst.push_back( getEdgeCellPointer( (uint_t) c, level ) );
            hipCheck( hyteg_hip_p2_edge_vector_cells_kinds( 3, count, dst.data(), 0, nullptr, &constant, (int) level, masks.data() + first, 0xFEu,
                                                            storage_->stream() ),
                      "P2Function::interpolate (batched)" );
         } );
         return;
      }
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         hipCheck( hyteg_hip_p2_edge_vector_cell_masked( 3, getEdgeCellPointer( c, level ), 0, nullptr, &constant, (int) level,
                                                         storage_->maskFor( cell, flag ), storage_->stream() ),
                   "P2Function::interpolate" );
      } );
   }
   // expression evaluated at the micro-vertices and at the edge midpoints (EdgeDoFFunction::interpolate)
   void interpolate( const std::function< ValueType( const Point3D& ) >& expr, uint_t level, DoFType flag = All ) const
   {
      vertexDoFFunction_.interpolate( expr, level, flag );
      static const int ends[7][2][3] = { { { 0, 0, 0 }, { 1, 0, 0 } }, { { 0, 0, 0 }, { 0, 1, 0 } }, { { 0, 0, 0 }, { 0, 0, 1 } },
                                         { { 1, 0, 0 }, { 0, 1, 0 } }, { { 1, 0, 0 }, { 0, 0, 1 } }, { { 0, 1, 0 }, { 0, 0, 1 } },
                                         { { 0, 1, 0 }, { 1, 0, 1 } } };
      const int64_t n    = int64_t( 1 ) << level;
      const double  step = 1.0 / double( n );
      const size_t  ne   = std::max< size_t >( 1, getNumberOfEdgeDoFs( level ) );
      std::vector< double* > tmp;
      forCells( [&]( uint_t, const MacroCell& cell ) {
         std::vector< double > host;
         host.reserve( ne );
         for ( int o = 0; o < 7; ++o )
         {
            const int64_t W = o == 6 ? n - 1 : n;
            for ( int64_t z = 0; z < W; ++z )
               for ( int64_t y = 0; y < W - z; ++y )
                  for ( int64_t x = 0; x < W - z - y; ++x )
                  {
                     const double mx = double( x ) + 0.5 * ( ends[o][0][0] + ends[o][1][0] ), my = double( y ) + 0.5 * ( ends[o][0][1] + ends[o][1][1] ),
                                  mz = double( z ) + 0.5 * ( ends[o][0][2] + ends[o][1][2] );
                     Point3D      q;
                     for ( int r = 0; r < 3; ++r )
                        q[r] = cell.coords[0][r] + ( cell.coords[1][r] - cell.coords[0][r] ) * step * mx +
                               ( cell.coords[2][r] - cell.coords[0][r] ) * step * my + ( cell.coords[3][r] - cell.coords[0][r] ) * step * mz;
                     host.push_back( expr( q ) );
                  }
         }
         double* t = storage_->acquireScratch( ne );
         hipCheck( hyteg_hip_upload( t, host.data(), host.size() * sizeof( double ), storage_->stream() ), "P2Function::interpolate: upload" );
         hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "P2Function::interpolate: sync" );
         tmp.push_back( t );
      } );
      // the copies of a shared edge DoF were evaluated from different cells' coordinates: make them bit-identical
      exchangeEdges( tmp, level, All, false );
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const double* srcs[1] = { tmp[c] };
         const double  one[1]  = { 1.0 };
         hipCheck( hyteg_hip_p2_edge_vector_cell_masked( 0, getEdgeCellPointer( c, level ), 1, srcs, one, (int) level,
                                                         storage_->maskFor( cell, flag ), storage_->stream() ),
                   "P2Function::interpolate: assign" );
      } );
      hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "P2Function::interpolate: sync" );
      for ( double* t : tmp )
         storage_->releaseScratch( ne, t );
   }
   void setToZero( uint_t level ) const { interpolate( ValueType( 0 ), level, All ); }

   void assign( const std::vector< ValueType >&                                           scalars,
                const std::vector< std::reference_wrapper< const P2Function< ValueType > > >& functions,
                uint_t                                                                    level,
                DoFType                                                                   flag = All ) const
   {
      vectorOp( 0, scalars, functions, level, flag );
   }
   void add( const std::vector< ValueType >&                                           scalars,
             const std::vector< std::reference_wrapper< const P2Function< ValueType > > >& functions,
             uint_t                                                                    level,
             DoFType                                                                   flag = All ) const
   {
      vectorOp( 1, scalars, functions, level, flag );
   }
   // P2Function::multElementwise (P2Function.cpp: vertex- and edge-DoF parts separately)
   void multElementwise( const std::vector< std::reference_wrapper< const P2Function< ValueType > > >& functions,
                         uint_t                                                                    level,
                         DoFType                                                                   flag = All ) const
   {
      vectorOp( 2, std::vector< ValueType >( functions.size(), ValueType( 1 ) ), functions, level, flag );
   }
   uint64_t uid() const { return uid_; }
   // assign / add / multElementwise restricted to DoF kinds (bit 0: vertex DoFs, 1..7: edge DoFs X, Y, Z, XY, XZ, YZ, XYZ): the
   // per-type sweeps of the P2 Gauss-Seidel smoother
   // `keep`: further restriction to point classes (HYTEG_HIP_MASK_INNER: inside the macro-cells, HYTEG_HIP_MASK_SHELL: on shared primitives)
   void assignKinds( const std::vector< ValueType >& scalars, const std::vector< std::reference_wrapper< const P2Function< ValueType > > >& functions,
                     uint_t level, DoFType flag, unsigned kinds, unsigned keep = HYTEG_HIP_MASK_ALL ) const
   {
      vectorOp( 0, scalars, functions, level, flag, kinds, keep );
   }
   void addKinds( const std::vector< ValueType >& scalars, const std::vector< std::reference_wrapper< const P2Function< ValueType > > >& functions,
                  uint_t level, DoFType flag, unsigned kinds, unsigned keep = HYTEG_HIP_MASK_ALL ) const
   {
      vectorOp( 1, scalars, functions, level, flag, kinds, keep );
   }
   void multElementwiseKinds( const std::vector< std::reference_wrapper< const P2Function< ValueType > > >& functions, uint_t level, DoFType flag,
                              unsigned kinds, unsigned keep = HYTEG_HIP_MASK_ALL ) const
   {
      vectorOp( 2, std::vector< ValueType >( functions.size(), ValueType( 1 ) ), functions, level, flag, kinds, keep );
   }
   // a shared DoF is counted by its lowest-numbered neighbour cell only
   ValueType dotLocal( const P2Function< ValueType >& rhs, uint_t level, DoFType flag = All ) const
   {
      double       sum = vertexDoFFunction_.dotLocal( rhs.vertexDoFFunction_, level, flag );
      const uint_t nl  = storage_->getNumberOfLocalCells();
      if ( storage_->useBatch( level ) )
      {
         const auto masks = storage_->masksFor( flag, true );
         storage_->forCellChunks( [&]( int first, int count ) {
            std::vector< const double* > a, b;
            for ( int c = first; c < first + count; ++c )
               a.push_back( getEdgeCellPointer( (uint_t) c, level ) ), b.push_back( rhs.getEdgeCellPointer( (uint_t) c, level ) );
            hipCheck( hyteg_hip_p2_edge_dot_cells_masked( count, a.data(), b.data(), (int) level, masks.data() + first, storage_->dotResult() + first,
                                                          storage_->stream() ),
                      "P2Function::dotLocal (batched)" );
         } );
      }
      else
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         hipCheck( hyteg_hip_p2_edge_dot_cell_masked( getEdgeCellPointer( c, level ), rhs.getEdgeCellPointer( c, level ), (int) level,
                                                      storage_->ownedMaskFor( cell, flag ), storage_->dotResult() + c, storage_->dotWorkspace(),
                                                      storage_->stream() ),
                   "P2Function::dotLocal" );
      } );
      std::vector< double > parts( nl, 0.0 );
      hipCheck( hyteg_hip_download( parts.data(), storage_->dotResult(), nl * sizeof( double ), storage_->stream() ), "P2Function::dotLocal: download" );
      for ( double v : parts )
         sum += v;
      return sum;
   }
   ValueType dotGlobal( const P2Function< ValueType >& rhs, uint_t level, DoFType flag = All ) const
   {
      return storage_->allreduceSum( dotLocal( rhs, level, flag ), "P2Function::dotGlobal" );
   }

   // every copy of a shared edge DoF := sum of all copies (communicateAdditively< Cell, Face / Edge > of the EdgeDoFFunction)
   void sumSharedEdgeCopies( uint_t level, DoFType flag = All ) const
   {
      std::vector< double* > arrays;
      forCells( [&]( uint_t c, const MacroCell& ) { arrays.push_back( getEdgeCellPointer( c, level ) ); } );
      exchangeEdges( arrays, level, flag, true );
   }

   // every copy := the copy held by the lowest-numbered neighbour cell
   void syncSharedEdgeCopies( uint_t level, DoFType flag = All ) const
   {
      std::vector< double* > arrays;
      forCells( [&]( uint_t c, const MacroCell& ) { arrays.push_back( getEdgeCellPointer( c, level ) ); } );
      exchangeEdges( arrays, level, flag, false );
   }

   void copyEdgeToHost( uint_t c, uint_t level, double* host ) const
   {
      hipCheck( hyteg_hip_download( host, getEdgeCellPointer( c, level ), getNumberOfEdgeDoFs( level ) * sizeof( double ), storage_->stream() ),
                "P2Function::copyEdgeToHost" );
   }
   void copyEdgeFromHost( uint_t c, uint_t level, const double* host ) const
   {
      hipCheck( hyteg_hip_upload( getEdgeCellPointer( c, level ), host, getNumberOfEdgeDoFs( level ) * sizeof( double ), storage_->stream() ),
                "P2Function::copyEdgeFromHost" );
      hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "P2Function::copyEdgeFromHost: sync" );
   }

 private:
   template < typename F >
   void forCells( F&& fn ) const
   {
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         fn( c, storage_->getLocalCell( c ) );
   }
   // additive (or copy) exchange of the shared edge DoFs held in `arrays` (one edge-DoF array per local cell), over all
   // ranks: the edge-DoF plans (dofKind 1) of the storage travel through the same transport as the vertex-DoF ones
   void exchangeEdges( const std::vector< double* >& arrays, uint_t level, DoFType flag, bool additive ) const
   {
      storage_->sharedExchangeBegin( arrays, (int) level, flag, 1 );
      storage_->sharedExchangeEnd( arrays, (int) level, flag, 1, additive );
   }
   void vectorOp( int                                                                       op,
                  const std::vector< ValueType >&                                           scalars,
                  const std::vector< std::reference_wrapper< const P2Function< ValueType > > >& functions,
                  uint_t                                                                    level,
                  DoFType                                                                   flag,
                  unsigned                                                                  kinds = 0xFFu,
                  unsigned                                                                  keep  = HYTEG_HIP_MASK_ALL ) const
   {
      if ( functions.empty() || functions.size() > HYTEG_HIP_MAX_SRCS || scalars.size() != functions.size() )
         throw std::runtime_error( "P2Function::assign/add/multElementwise: bad number of functions or scalars" );
      std::vector< std::reference_wrapper< const P1Function< ValueType > > > vs;
      for ( uint_t k = 0; k < functions.size(); ++k )
         vs.push_back( functions[k].get().vertexDoFFunction_ );
      if ( kinds & 1u )
      {
         if ( op == 0 )
            vertexDoFFunction_.assign( scalars, vs, level, flag );
         else if ( op == 1 )
            vertexDoFFunction_.add( scalars, vs, level, flag );
         else
            vertexDoFFunction_.multElementwise( vs, level, flag );
      }
      if ( ( kinds & 0xFEu ) == 0 )
         return;
      if ( storage_->useBatch( level ) )
      {
         // all local cells in one launch per chunk (as the vertex-DoF part above)
         const auto masks = storage_->masksFor( flag, false, keep );
         storage_->forCellChunks( [&]( int first, int count ) {
            std::vector< double* >       dst;
            std::vector< const double* > srcs; // [function][cell]
            for ( int c = first; c < first + count; ++c )
               dst.push_back( getEdgeCellPointer( (uint_t) c, level ) );
            for ( const auto& f : functions )
               for ( int c = first; c < first + count; ++c )
                  srcs.push_back( f.get().getEdgeCellPointer( (uint_t) c, level ) );
            hipCheck( hyteg_hip_p2_edge_vector_cells_kinds( op, count, dst.data(), (int) functions.size(), srcs.data(), scalars.data(), (int) level,
                                                            masks.data() + first, kinds, storage_->stream() ),
                      "P2Function vector op (batched)" );
         } );
         return;
      }
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const double* es[HYTEG_HIP_MAX_SRCS];
         for ( uint_t k = 0; k < functions.size(); ++k )
            es[k] = functions[k].get().getEdgeCellPointer( c, level );
         hipCheck( hyteg_hip_p2_edge_vector_cell_kinds( op, getEdgeCellPointer( c, level ), (int) functions.size(), es, scalars.data(), (int) level,
                                                        storage_->maskFor( cell, flag ) & keep, kinds, storage_->stream() ),
                   "P2Function vector op" );
      } );
   }

   std::string                                            name_;
   std::shared_ptr< PrimitiveStorage >                    storage_;
   uint_t                                                 minLevel_, maxLevel_;
   P1Function< ValueType >                                vertexDoFFunction_;
   std::vector< std::vector< double* > >                  edge_; // [local cell][level - minLevel]
   uint64_t                                               uid_ = nextUid();
};

} // namespace hyteg
