// p1function.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// P1Function< double > (= vertexdof::VertexDoFFunction< double >, src/hyteg/p1functionspace/VertexDoFFunction.cpp)
#pragma once

#include "storage.hpp"

namespace hyteg {

// =====================================================================================================
// P1Function< double > (= vertexdof::VertexDoFFunction< double >)
// =====================================================================================================
template < typename ValueType >
class P1Function
{
   static_assert( std::is_same< ValueType, double >::value,
                  "only double is supported, like the reference's generated 3D kernels (P1ConstantOperator.cpp:417-420)" );

 public:
   using valueType = ValueType;

   uint64_t uid() const { return uid_; }

   // BoundaryCondition::createAllInnerBC() (the pressure of P1StokesFunction, composites/P1StokesFunction.hpp:41-42): every
   // point of this function counts as Inner, whatever the storage says about the macro-primitive it lies on.  Default:
   // the storage's boundary types (create0123BC).  Operators, grid transfer and the function's own methods translate the
   // DoFType flag they are given with the destination function's boundary condition, as the reference does with
   // dst.getBoundaryCondition().getBoundaryType( primitive.getMeshBoundaryFlag() ) (P1Operator.hpp:301-303).
   void    setBoundaryConditionAllInner( bool on = true ) { allInner_ = on; }
   bool    hasAllInnerBoundaryCondition() const { return allInner_; }
   DoFType effectiveFlag( DoFType flag ) const { return allInner_ ? ( testFlag( flag, Inner ) ? All : DoFType( 0 ) ) : flag; }

   // scratch = true: arrays come from (and return to) the storage's scratch pool and are NOT zero-initialised
   P1Function( const std::string& name, const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel,
               bool scratch = false )
   : name_( name )
   , storage_( storage )
   , minLevel_( minLevel )
   , maxLevel_( maxLevel )
   , scratch_( scratch )
   {
      if ( maxLevel > HYTEG_HIP_MAX_LEVEL || minLevel > maxLevel )
         throw std::runtime_error( "P1Function: bad level range" );
      const uint_t nLocal = storage->getNumberOfLocalCells();
      data_.resize( nLocal );
      for ( uint_t c = 0; c < nLocal; ++c )
         for ( uint_t l = minLevel; l <= maxLevel; ++l )
         {
            const size_t doubles = (size_t) layout::cellSize( (int) l );
            if ( scratch )
            {
               data_[c].push_back( storage->acquireScratch( doubles ) );
               continue;
            }
            void* p = nullptr;
            hipCheck( hyteg_hip_malloc( &p, doubles * sizeof( double ) ), "P1Function: malloc" );
            hipCheck( hyteg_hip_memset_zero( p, doubles * sizeof( double ), storage->stream() ), "P1Function: memset" );
            data_[c].push_back( static_cast< double* >( p ) );
         }
   }
   ~P1Function()
   {
      for ( auto& c : data_ )
         for ( uint_t l = 0; l < c.size(); ++l )
         {
            if ( scratch_ )
               storage_->releaseScratch( (size_t) layout::cellSize( (int) ( minLevel_ + l ) ), c[l] );
            else
               hyteg_hip_free( c[l] );
         }
   }
   P1Function( const P1Function& )            = delete;
   P1Function& operator=( const P1Function& ) = delete;

   const std::string&                  getFunctionName() const { return name_; }
   std::shared_ptr< PrimitiveStorage > getStorage() const { return storage_; }
   uint_t                              getMinLevel() const { return minLevel_; }
   uint_t                              getMaxLevel() const { return maxLevel_; }

   // device pointer of the array of local cell `c` at `level` (FunctionMemory::getPointer, FunctionMemory.hpp:109-113)
   double* getCellPointer( uint_t c, uint_t level ) const
   {
      checkLevel( level );
      return data_.at( c )[level - minLevel_];
   }

   // device pointers of local cells [first, first + count) at `level`
   std::vector< double* > cellPointers( uint_t level, int first, int count ) const
   {
      std::vector< double* > p;
      for ( int c = first; c < first + count; ++c )
         p.push_back( getCellPointer( (uint_t) c, level ) );
      return p;
   }

   // ---- interpolate ( VertexDoFFunction.cpp:380-392, :395-470 ) ----
   void interpolate( ValueType constant, uint_t level, DoFType flagIn = All ) const
   {
      ScopedTimer timerFn( storage_->getTimingTree(), "P1Function" ), timerOp( storage_->getTimingTree(), "Interpolate" );
      const DoFType flag = effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      if ( storage_->useBatch( level ) )
      {
         const auto masks = storage_->masksFor( flag );
         storage_->forCellChunks( [&]( int first, int count ) {
            const auto dst = cellPointers( level, first, count );
            hipCheck( hyteg_hip_p1_vector_cells( 3, count, dst.data(), 0, nullptr, &constant, (int) level, masks.data() + first,
                                                 storage_->stream() ),
                      "interpolate (batched)" );
         } );
         return;
      }
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         hipCheck( hyteg_hip_p1_set_cell_masked( getCellPointer( c, level ), constant, (int) level, storage_->maskFor( cell, flag ),
                                                 storage_->stream() ),
                   "interpolate" );
      } );
   }
   void interpolate( const std::function< ValueType( const Point3D& ) >& expr, uint_t level, DoFType flagIn = All ) const
   {
      ScopedTimer timerFn( storage_->getTimingTree(), "P1Function" ), timerOp( storage_->getTimingTree(), "Interpolate" );
      const DoFType flag = effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      // evaluated on the host at the micro-vertex coordinates of VertexDoFMacroCell.hpp:70-77, then uploaded
      const int64_t N = layout::width( (int) level ), size = layout::cellSize( (int) level );
      P1Function    tmp( "interpolate_tmp", storage_, level, level, true );
      std::vector< double > host( (size_t) size );
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const double step = 1.0 / double( N - 1 );
         int64_t      k    = 0;
         for ( int64_t z = 0; z < N; ++z )
            for ( int64_t y = 0; y < N - z; ++y )
               for ( int64_t x = 0; x < N - z - y; ++x )
               {
                  Point3D p;
                  for ( int r = 0; r < 3; ++r )
                  {
                     const double xs = ( cell.coords[1][r] - cell.coords[0][r] ) * step;
                     const double ys = ( cell.coords[2][r] - cell.coords[0][r] ) * step;
                     const double zs = ( cell.coords[3][r] - cell.coords[0][r] ) * step;
                     p[r]            = cell.coords[0][r] + xs * double( x ) + ys * double( y ) + zs * double( z );
                  }
                  host[(size_t) k++] = expr( p );
               }
         hipCheck( hyteg_hip_upload( tmp.getCellPointer( c, level ), host.data(), (size_t) size * sizeof( double ), storage_->stream() ),
                   "interpolate: upload" );
         hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "interpolate: sync" );
      } );
      // the copies of a shared DoF are evaluated from different cells' coordinates: make them bit-identical
      tmp.syncSharedCopies( level );
      assign( { 1.0 }, { tmp }, level, flag );
   }

   // ---- assign / add / multElementwise ( VertexDoFFunction.cpp:1130-1221, :1408-1484, :1487-1563 ) ----
   void assign( const std::vector< ValueType >&                                           scalars,
                const std::vector< std::reference_wrapper< const P1Function< ValueType > > >& functions,
                uint_t                                                                    level,
                DoFType                                                                   flag = All ) const
   {
      vectorOp( 0, scalars, functions, level, flag );
   }
   void add( const std::vector< ValueType >&                                           scalars,
             const std::vector< std::reference_wrapper< const P1Function< ValueType > > >& functions,
             uint_t                                                                    level,
             DoFType                                                                   flag = All ) const
   {
      vectorOp( 1, scalars, functions, level, flag );
   }
   void multElementwise( const std::vector< std::reference_wrapper< const P1Function< ValueType > > >& functions,
                         uint_t                                                                    level,
                         DoFType                                                                   flag = All ) const
   {
      vectorOp( 2, {}, functions, level, flag );
   }
   void setToZero( uint_t level ) const { interpolate( ValueType( 0 ), level, All ); }

   // ---- dot ( VertexDoFFunction.cpp:1710-1793 ) ----
   ValueType dotLocal( const P1Function< ValueType >& rhs, uint_t level, DoFType flagIn = All ) const
   {
      ScopedTimer timerFn( storage_->getTimingTree(), "P1Function" ), timerOp( storage_->getTimingTree(), "Dot (local)" );
      const DoFType flag = effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      // one result slot per local cell, a single download (= one host synchronisation) per dot product; the
      // workspace is reused cell after cell, which is safe because all launches are ordered on one stream
      const uint_t nLocal = storage_->getNumberOfLocalCells();
      if ( storage_->useBatch( level ) )
      {
         // one partial + one final launch per chunk of cells, one number per chunk comes back
         const auto masks  = storage_->masksFor( flag, true );
         int        nchunk = 0;
         storage_->forCellChunks( [&]( int first, int count ) {
            const auto a = cellPointers( level, first, count ), b = rhs.cellPointers( level, first, count );
            hipCheck( hyteg_hip_p1_dot_cells( count, a.data(), b.data(), (int) level, masks.data() + first, storage_->dotResult() + nchunk,
                                              storage_->dotWorkspace(), storage_->stream() ),
                      "dotLocal (batched)" );
            ++nchunk;
         } );
         std::vector< double > parts( (size_t) nchunk, 0.0 );
         hipCheck( hyteg_hip_download( parts.data(), storage_->dotResult(), parts.size() * sizeof( double ), storage_->stream() ),
                   "dotLocal: download" );
         double sum = 0.0;
         for ( double v : parts )
            sum += v;
         return sum;
      }
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         hipCheck( hyteg_hip_p1_dot_cell_masked( getCellPointer( c, level ), rhs.getCellPointer( c, level ), (int) level,
                                                 storage_->ownedMaskFor( cell, flag ), storage_->dotResult() + c, storage_->dotWorkspace(),
                                                 storage_->stream() ),
                   "dotLocal" );
      } );
      std::vector< double > parts( nLocal, 0.0 );
      if ( nLocal > 0 )
         hipCheck( hyteg_hip_download( parts.data(), storage_->dotResult(), nLocal * sizeof( double ), storage_->stream() ),
                   "dotLocal: download" );
      double sum = 0.0;
      for ( double v : parts )
         sum += v; // cells in ascending order: deterministic
      return sum;
   }
   ValueType dotGlobal( const P1Function< ValueType >& rhs, uint_t level, DoFType flag = All ) const
   {
      const double local = dotLocal( rhs, level, flag );
      ScopedTimer  timerFn( storage_->getTimingTree(), "P1Function" ), timerOp( storage_->getTimingTree(), "Dot (reduce)" );
      return storage_->allreduceSum( local, "dotGlobal" );
   }

   // ---- shared-point exchange (the cell-centric replacement of communicate<> / communicateAdditively<>) ----
   // additive: every copy of a shared DoF := sum of all copies (VertexDoFAdditivePackInfo.hpp:676-745 + copy back)
   void sumSharedCopies( uint_t level, DoFType flag = All ) const
   {
      exchangeBegin( level, flag );
      exchangeEnd( level, flag, true );
   }
   // every copy := the copy held by the lowest-numbered neighbour cell
   void syncSharedCopies( uint_t level, DoFType flag = All ) const
   {
      exchangeBegin( level, flag );
      exchangeEnd( level, flag, false );
   }
   // split form: pack + start the transfer / wait + reduce.  Kernels that do not touch shared points may be
   // launched in between (the interior apply overlaps the halo exchange).
   void beginSumSharedCopies( uint_t level, DoFType flag = All ) const { exchangeBegin( level, flag ); }
   void endSumSharedCopies( uint_t level, DoFType flag = All ) const { exchangeEnd( level, flag, true ); }
   // beginSumSharedCopies for a caller whose boundary-share kernel delivers the shares itself
   // (PrimitiveStorage::sharedExchangeBeginByShares); flag: effective flag
   bool beginSumSharedCopiesByShares( uint_t level, DoFType flag, PrimitiveStorage::ShareSend& out ) const
   {
      checkLevel( level );
      return storage_->sharedExchangeBeginByShares( (int) level, flag, out );
   }

   void copyCellToHost( uint_t c, uint_t level, double* host ) const
   {
      hipCheck( hyteg_hip_download( host, getCellPointer( c, level ), (size_t) layout::cellSize( (int) level ) * sizeof( double ),
                                    storage_->stream() ),
                "copyCellToHost" );
      // the host is about to use values that may have come through a peer-to-peer exchange: a timed-out arrival wait fails here
      storage_->checkTransportAtHostRead();
   }
   void copyCellFromHost( uint_t c, uint_t level, const double* host ) const
   {
      hipCheck( hyteg_hip_upload( getCellPointer( c, level ), host, (size_t) layout::cellSize( (int) level ) * sizeof( double ),
                                  storage_->stream() ),
                "copyCellFromHost" );
      hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "copyCellFromHost: sync" );
   }

 private:
   void checkLevel( uint_t level ) const
   {
      if ( level < minLevel_ || level > maxLevel_ )
         throw std::runtime_error( "P1Function '" + name_ + "': level " + std::to_string( level ) + " not allocated" );
   }
   template < typename F >
   void forCells( F&& fn ) const
   {
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         fn( c, storage_->getLocalCell( c ) );
   }
   void vectorOp( int                                                                       op,
                  const std::vector< ValueType >&                                           scalars,
                  const std::vector< std::reference_wrapper< const P1Function< ValueType > > >& functions,
                  uint_t                                                                    level,
                  DoFType                                                                   flagIn ) const
   {
      const DoFType flag = effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      static const char* kOpNames[3] = { "Assign", "Add", "Multiply elementwise" };
      ScopedTimer        timerFn( storage_->getTimingTree(), "P1Function" ), timerOp( storage_->getTimingTree(), kOpNames[op] );
      if ( functions.empty() || functions.size() > HYTEG_HIP_MAX_SRCS || ( op != 2 && scalars.size() != functions.size() ) )
         throw std::runtime_error( "P1Function::assign/add/multElementwise: bad number of functions or scalars" );
      if ( storage_->useBatch( level ) )
      {
         const auto masks = storage_->masksFor( flag );
         storage_->forCellChunks( [&]( int first, int count ) {
            const auto             dst = cellPointers( level, first, count );
            std::vector< double* > srcs; // [function][cell]
            for ( const auto& f : functions )
               for ( double* q : f.get().cellPointers( level, first, count ) )
                  srcs.push_back( q );
            hipCheck( hyteg_hip_p1_vector_cells( op, count, dst.data(), (int) functions.size(), srcs.data(),
                                                 op == 2 ? nullptr : scalars.data(), (int) level, masks.data() + first, storage_->stream() ),
                      "P1Function vector op (batched)" );
         } );
         return;
      }
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const double* srcs[HYTEG_HIP_MAX_SRCS];
         for ( uint_t k = 0; k < functions.size(); ++k )
            srcs[k] = functions[k].get().getCellPointer( c, level );
         hipCheck( hyteg_hip_p1_vector_cell_masked( op, getCellPointer( c, level ), (int) functions.size(), srcs,
                                                    op == 2 ? nullptr : scalars.data(), (int) level, storage_->maskFor( cell, flag ),
                                                    storage_->stream() ),
                   "P1Function vector op" );
      } );
   }

 public:
   // assign (op 0) / add (op 1) with coefficients read from device memory when the kernels run; storages of one rank
   // with at most HYTEG_HIP_MAX_BATCH local cells (one launch)
   void vectorOpDeviceScalars( int op, const std::vector< const double* >& scalarPtrs,
                               const std::vector< std::reference_wrapper< const P1Function< ValueType > > >& functions, uint_t level,
                               DoFType flagIn ) const
   {
      const DoFType flag = effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      const int  count = (int) storage_->getNumberOfLocalCells();
      const auto masks = storage_->masksFor( flag );
      const auto dst   = cellPointers( level, 0, count );
      std::vector< double* > srcs;
      for ( const auto& f : functions )
         for ( double* q : f.get().cellPointers( level, 0, count ) )
            srcs.push_back( q );
      hipCheck( hyteg_hip_p1_vector_cells_dev( op, count, dst.data(), (int) functions.size(), srcs.data(), scalarPtrs.data(), (int) level,
                                               masks.data(), storage_->stream() ),
                "P1Function vector op (device scalars)" );
   }
   // cgScalars[slot] = <this, rhs> over the points `flag` selects (each shared point counted once), then phase `phase` of
   // the conjugate gradient recurrences (hyteg_hip_cg_scalars), in one launch; no host synchronisation
   void dotLocalToCgScalars( const P1Function< ValueType >& rhs, uint_t level, DoFType flagIn, double* cgScalars, int slot, int phase,
                             double relTol, double absTol ) const
   {
      const DoFType flag = effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      const int  count = (int) storage_->getNumberOfLocalCells();
      const auto masks = storage_->masksFor( flag, true );
      const auto a = cellPointers( level, 0, count ), b = rhs.cellPointers( level, 0, count );
      hipCheck( hyteg_hip_p1_dot_cells_cg( count, a.data(), b.data(), (int) level, masks.data(), cgScalars, slot, phase, relTol, absTol,
                                           storage_->dotWorkspace(), storage_->stream() ),
                "dotLocalToCgScalars" );
   }

 private:
   std::vector< double* > cellArrays( uint_t level ) const
   {
      std::vector< double* > a;
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         a.push_back( getCellPointer( c, level ) );
      return a;
   }
   void exchangeBegin( uint_t level, DoFType flagIn ) const
   {
      const DoFType flag = effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      checkLevel( level );
      storage_->sharedExchangeBegin( cellArrays( level ), (int) level, flag, 0 );
   }
   void exchangeEnd( uint_t level, DoFType flagIn, bool additive ) const
   {
      const DoFType flag = effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      storage_->sharedExchangeEnd( cellArrays( level ), (int) level, flag, 0, additive );
   }

   std::string                                                           name_;
   std::shared_ptr< PrimitiveStorage >                                   storage_;
   uint_t                                                                minLevel_, maxLevel_;
   bool                                                                  scratch_ = false;
   bool                                                                  allInner_ = false;
   uint64_t                                                              uid_     = nextUid();
   std::vector< std::vector< double* > >                                 data_;
};

} // namespace hyteg
