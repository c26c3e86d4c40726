// p1operator.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// P1ConstantOperator< Form > (src/constant_stencil_operator/P1ConstantOperator.hpp,
// src/hyteg/p1functionspace/P1Operator.hpp)
#pragma once

#include "forms.hpp"
#include "p1function.hpp"

namespace hyteg {

// =====================================================================================================
// P1ConstantOperator< Form >  ( src/constant_stencil_operator/P1ConstantOperator.hpp:33-168 )
// =====================================================================================================
template < class Form >
class P1ConstantOperator
{
 public:
   using srcType = P1Function< double >;
   using dstType = P1Function< double >;

   P1ConstantOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : storage_( storage )
   , minLevel_( minLevel )
   , maxLevel_( maxLevel )
   {
      // assembleStencils(), P1ConstantOperator.cpp:680-732: per level and cell
      for ( uint_t l = minLevel; l <= maxLevel; ++l )
      {
         std::vector< stencil::CellStencils > perCell;
         for ( const auto& cell : storage->getCells() ) // all cells: inverse diagonals need the neighbours' shares
            perCell.push_back( stencil::assemble< Form >( cell, l ) );
         stencils_[l] = perCell;
         sorTables_[l] = buildSorTables( perCell );
         if ( l >= HYTEG_HIP_MIN_LEVEL )
            hipCheck( hyteg_hip_prepare_level( (int) l ), "P1ConstantOperator: prepare_level" );
      }
   }

   std::shared_ptr< PrimitiveStorage > getStorage() const { return storage_; }
   uint64_t                            uid() const { return uid_; }
   uint_t                              getMinLevel() const { return minLevel_; }
   uint_t                              getMaxLevel() const { return maxLevel_; }
   const stencil::CellStencils&        getCellStencils( int globalCellID, uint_t level ) const { return stencils_.at( level ).at( globalCellID ); }

   // Operator::apply, P1Operator.hpp:192-320
   void apply( const P1Function< double >& src, const P1Function< double >& dst, uint_t level, DoFType flagIn, UpdateType updateType = Replace ) const
   {
      ScopedTimer timerOp( storage_->getTimingTree(), "Operator P1Function to P1Function" ), timerApply( storage_->getTimingTree(), "Apply" );
      const DoFType flag = dst.effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      if ( &src == &dst )
         throw std::runtime_error( "P1ConstantOperator::apply: src and dst must differ (P1Operator.hpp:198)" );
      if ( storage_->useBatch( level ) )
      {
         applyBatched( src, dst, level, flag, updateType );
         return;
      }
      const P1Function< double >* shellDst = &dst;
      std::unique_ptr< P1Function< double > > tmp;
      if ( updateType == Add && hasSharedPoints( level, flag ) )
      {
         // partial results of shared DoFs are summed over cells before they are added to dst
         tmp.reset( new P1Function< double >( "apply_tmp", storage_, level, level, true ) );
         tmp->interpolate( 0.0, level, All );
         shellDst = tmp.get();
      }
      // 1. this cell's share of the shared macro-face/edge/vertex DoFs (tiny kernels), 2. start the halo exchange,
      // 3. the interior stencil while the exchange is in flight, 4. reduce the shares.  1, 2 and 4 only touch shared points:
      // with a stream-agnostic transport they can form a chain on a side stream next to 3 (PrimitiveStorage::SideChain;
      // opt-in, measured slower than one stream).
      PrimitiveStorage::SideChain chain( *storage_, storage_->sideChainUsable( (int) level, flag, 0 ) );
      // a rank with ONE macro-cell that exchanges peer to peer: the share kernel stores the shares into the peers' slots itself
      PrimitiveStorage::ShareSend send;
      const bool bySharesKernel = shellDst->beginSumSharedCopiesByShares( level, flag, send );
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const auto& S = getCellStencils( cell.id, level );
         if ( bySharesKernel )
         {
            hipCheck( hyteg_hip_p1_apply_cell_boundary_p2p( shellDst->getCellPointer( c, level ), src.getCellPointer( c, level ), (int) level,
                                                            &S.slots[0][0], storage_->maskFor( cell, flag ),
                                                            ( updateType == Add && shellDst == &dst ) ? HYTEG_HIP_ADD : HYTEG_HIP_REPLACE,
                                                            send.first, send.list, send.a.peers, send.a.npeers, send.a.seq, send.a.counter,
                                                            storage_->stream() ),
                      "apply: boundary + send" );
            return;
         }
         hipCheck( hyteg_hip_p1_apply_cell_boundary( shellDst->getCellPointer( c, level ), src.getCellPointer( c, level ), (int) level,
                                                     &S.slots[0][0], storage_->maskFor( cell, flag ),
                                                     ( updateType == Add && shellDst == &dst ) ? HYTEG_HIP_ADD : HYTEG_HIP_REPLACE,
                                                     storage_->stream() ),
                   "apply: boundary" );
      } );
      if ( !bySharesKernel )
         shellDst->beginSumSharedCopies( level, flag );
      chain.toMain();
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const unsigned mask = storage_->maskFor( cell, flag );
         if ( ( mask & HYTEG_HIP_MASK_INNER ) && level >= HYTEG_HIP_MIN_LEVEL )
            hipCheck( hyteg_hip_p1_apply_cell( dst.getCellPointer( c, level ), src.getCellPointer( c, level ), (int) level,
                                               getCellStencils( cell.id, level ).inner,
                                               updateType == Replace ? HYTEG_HIP_REPLACE : HYTEG_HIP_ADD, storage_->stream() ),
                      "apply: cell" );
      } );
      chain.toSide();
      shellDst->endSumSharedCopies( level, flag );
      chain.join();
      if ( shellDst != &dst )
      {
         // dst += tmp on the shell points selected by flag
         forCells( [&]( uint_t c, const MacroCell& cell ) {
            const double* srcs[1] = { shellDst->getCellPointer( c, level ) };
            const double  one[1]  = { 1.0 };
            hipCheck( hyteg_hip_p1_vector_cell_masked( 1, dst.getCellPointer( c, level ), 1, srcs, one, (int) level,
                                                       storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_SHELL, storage_->stream() ),
                      "apply: add shell" );
         } );
      }
   }

   // r = b - A x on the points `flag` selects: apply followed by assign( { 1, -1 }, { b, r } ) as the multigrid cycle writes it
   // (GeometricMultigridSolver.hpp:240-246); where no shell point is selected (one macro-cell with fixed boundary values) the
   // interior kernel forms the difference itself -- one launch, the same bits
   void residual( const P1Function< double >& x, const P1Function< double >& b, const P1Function< double >& r, uint_t level, DoFType flagIn ) const
   {
      const DoFType flag     = r.effectiveFlag( flagIn );
      bool          anyShell = storage_->numRanks() > 1 || storage_->useBatch( level ) || level < HYTEG_HIP_MIN_LEVEL || level > 10 || &x == &r;
      forCells( [&]( uint_t, const MacroCell& cell ) { anyShell = anyShell || ( storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_SHELL ); } );
      if ( anyShell )
      {
         apply( x, r, level, flagIn );
         r.assign( { 1.0, -1.0 }, { b, r }, level, flagIn );
         return;
      }
      ScopedTimer timerOp( storage_->getTimingTree(), "Operator P1Function to P1Function" ), timerApply( storage_->getTimingTree(), "Apply" );
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         if ( storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_INNER )
            hipCheck( hyteg_hip_p1_residual_cell( r.getCellPointer( c, level ), b.getCellPointer( c, level ), x.getCellPointer( c, level ),
                                                  (int) level, getCellStencils( cell.id, level ).inner, storage_->stream() ),
                      "residual: cell" );
      } );
   }

   // P1Operator::smooth_jac, P1Operator.hpp:429-447
   void smooth_jac( const P1Function< double >& dst, const P1Function< double >& rhs, const P1Function< double >& src, double relax,
                    uint_t level, DoFType flagIn ) const
   {
      ScopedTimer timerOp( storage_->getTimingTree(), "Operator P1Function to P1Function" ), timerJac( storage_->getTimingTree(), "smooth_jac" );
      const DoFType flag = dst.effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      if ( &src == &dst )
         throw std::runtime_error( "smooth_jac: src and dst must differ" );
      const auto& invDiag = *getInverseDiagonalValues();
      if ( storage_->useBatch( level ) )
      {
         // phase 0: inner points complete, shell points this cell's share; exchange; phase 1: shell update
         const auto masks = storage_->masksFor( flag );
         for ( int phase = 0; phase < 2; ++phase )
         {
            storage_->forCellChunks( [&]( int first, int count ) {
               const auto d = dst.cellPointers( level, first, count ), r = rhs.cellPointers( level, first, count ),
                          u = src.cellPointers( level, first, count ), iv = invDiag.cellPointers( level, first, count );
               hipCheck( hyteg_hip_p1_jacobi_cells( count, d.data(), r.data(), u.data(), iv.data(), (int) level,
                                                    stencilTable( level ) + (size_t) first * 225, relax, masks.data() + first, phase,
                                                    storage_->stream() ),
                         "smooth_jac (batched)" );
            } );
            if ( phase == 0 )
               dst.sumSharedCopies( level, flag );
         }
         return;
      }
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const auto&    S    = getCellStencils( cell.id, level );
         const unsigned mask = storage_->maskFor( cell, flag );
         if ( ( mask & HYTEG_HIP_MASK_INNER ) && level >= HYTEG_HIP_MIN_LEVEL )
            hipCheck( hyteg_hip_p1_jacobi_cell( dst.getCellPointer( c, level ), rhs.getCellPointer( c, level ), src.getCellPointer( c, level ),
                                                nullptr, (int) level, S.inner, relax, storage_->stream() ),
                      "smooth_jac: cell" );
         hipCheck( hyteg_hip_p1_apply_cell_boundary( dst.getCellPointer( c, level ), src.getCellPointer( c, level ), (int) level,
                                                     &S.slots[0][0], mask, HYTEG_HIP_REPLACE, storage_->stream() ),
                   "smooth_jac: boundary" );
      } );
      dst.sumSharedCopies( level, flag );
      // on the shell: dst = rhs - dst ; dst = invDiag .* dst ; dst = src + relax * dst  (the reference's three passes)
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const unsigned shell = storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_SHELL;
         if ( !shell )
            return;
         double*       d = dst.getCellPointer( c, level );
         const double* a[2] = { rhs.getCellPointer( c, level ), d };
         const double  s1[2] = { 1.0, -1.0 };
         hipCheck( hyteg_hip_p1_vector_cell_masked( 0, d, 2, a, s1, (int) level, shell, storage_->stream() ), "smooth_jac: residual" );
         const double* m[2] = { invDiag.getCellPointer( c, level ), d };
         hipCheck( hyteg_hip_p1_vector_cell_masked( 2, d, 2, m, nullptr, (int) level, shell, storage_->stream() ), "smooth_jac: scale" );
         const double* u[2] = { src.getCellPointer( c, level ), d };
         const double  s2[2] = { 1.0, relax };
         hipCheck( hyteg_hip_p1_vector_cell_masked( 0, d, 2, u, s2, (int) level, shell, storage_->stream() ), "smooth_jac: update" );
      } );
   }

   // `steps` consecutive sweeps of smooth_sor (what a multigrid cycle's pre- / post-smoothing loop does,
   // GeometricMultigridSolver.hpp:209-215).  Where no shared or boundary point is swept (every macro-cell's shell is fixed: one
   // macro-cell with Dirichlet values, or cells whose common faces are not selected by the flag) the sweeps of a cell do not
   // see anybody else's updates in between, and from level 5 on they run as one pipeline of block wavefronts
   // (hyteg_hip_p1_sor_cell_sweeps: bit-identical to the loop, 46 + 4 ( steps - 1 ) launches instead of 46 steps at level 8).
   void smooth_sor_steps( const P1Function< double >& dst, const P1Function< double >& rhs, double relax, uint_t level, DoFType flagIn,
                          uint_t steps, bool backwards = false ) const
   {
      const DoFType flag     = dst.effectiveFlag( flagIn );
      bool          anyShell = false;
      forCells( [&]( uint_t, const MacroCell& cell ) { anyShell = anyShell || ( storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_SHELL ); } );
      if ( steps <= 1 || anyShell || storage_->numRanks() != 1 || storage_->useBatchSor( level ) || level < 5 )
      {
         for ( uint_t k = 0; k < steps; ++k )
            smooth_sor( dst, rhs, relax, level, flagIn, backwards );
         return;
      }
      ScopedTimer timerOp( storage_->getTimingTree(), "Operator P1Function to P1Function" ), timerSor( storage_->getTimingTree(), backwards ? "SOR backwards" : "SOR" );
      if ( &dst == &rhs )
         throw std::runtime_error( "smooth_sor: dst and rhs must differ" );
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         if ( storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_INNER )
            hipCheck( hyteg_hip_p1_sor_cell_sweeps( dst.getCellPointer( c, level ), rhs.getCellPointer( c, level ), (int) level,
                                                    getCellStencils( cell.id, level ).inner, relax, backwards ? 1 : 0, (int) steps,
                                                    storage_->stream() ),
                      "smooth_sor_steps: cell" );
      } );
   }
   // P1Operator::smooth_sor / smooth_gs, P1Operator.hpp:322-418: macro-vertices, -edges, -faces, -cells (reversed for
   // backwards), each class with the values the reference's communication schedule gives it.  Cell-centric form:
   //  rest  = (stencil sum over the neighbours outside the primitive's closure), summed over cells by ONE exchange,
   //          taken from the pre-sweep state (forward) -- the reference's ghost layers are not refreshed in between;
   //  sweep = every cell runs the vertex / edge / face sweeps on its own copies with the total weights (bit-identical
   //          copies, no further exchange), then the lexicographic macro-cell sweep.
   // Backwards the reference communicates before every class, so `rest` is rebuilt (and exchanged) per class.
   void smooth_sor( const P1Function< double >& dst, const P1Function< double >& rhs, double relax, uint_t level, DoFType flagIn,
                    bool backwards = false ) const
   {
      ScopedTimer timerOp( storage_->getTimingTree(), "Operator P1Function to P1Function" ), timerSor( storage_->getTimingTree(), backwards ? "SOR backwards" : "SOR" );
      const DoFType flag = dst.effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      if ( &dst == &rhs )
         throw std::runtime_error( "smooth_sor: dst and rhs must differ" );
      bool anyShell = false;
      forCells( [&]( uint_t, const MacroCell& cell ) { anyShell = anyShell || ( storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_SHELL ); } );
      auto sweepCells = [&]() { launchSorCells( dst, rhs, relax, level, flag, backwards ); };
      if ( !anyShell && storage_->numRanks() == 1 )
      {
         sweepCells();
         return;
      }
      auto& restSlot = sorRest_[level];
      if ( !restSlot )
         restSlot.reset( new P1Function< double >( "sor_rest", storage_, level, level ) );
      P1Function< double >& rest = *restSlot;
      auto                 sweepShell = [&]( unsigned bits ) {
         if ( storage_->useBatchSor( level ) )
         {
            const auto masks = storage_->masksFor( flag, false, bits & HYTEG_HIP_MASK_SHELL );
            storage_->forCellChunks( [&]( int first, int count ) {
               const auto r = rest.cellPointers( level, first, count ), u = dst.cellPointers( level, first, count );
               hipCheck( hyteg_hip_p1_apply_cells( count, r.data(), u.data(), (int) level, restTable( level ) + (size_t) first * 225,
                                                   masks.data() + first, HYTEG_HIP_REPLACE, storage_->stream() ),
                         "smooth_sor: rest (batched)" );
            } );
         }
         else
         forCells( [&]( uint_t c, const MacroCell& cell ) {
            const auto&    T    = sorTables_.at( level ).at( cell.id );
            const unsigned mask = storage_->maskFor( cell, flag ) & bits;
            hipCheck( hyteg_hip_p1_apply_cell_boundary( rest.getCellPointer( c, level ), dst.getCellPointer( c, level ), (int) level,
                                                        &T.rest[0][0], mask, HYTEG_HIP_REPLACE, storage_->stream() ),
                      "smooth_sor: rest" );
         } );
         rest.sumSharedCopies( level, flag );
         launchSorShell( dst, rhs, rest, relax, level, flag, bits, backwards );
      };
      if ( !backwards )
      {
         sweepShell( HYTEG_HIP_MASK_SHELL );
         sweepCells();
      }
      else
      {
         sweepCells();
         sweepShell( 0xFu << 6 );  // macro-faces
         sweepShell( 0x3Fu );      // macro-edges
         sweepShell( 0xFu << 10 ); // macro-vertices
      }
   }
   // The two halves of smooth_sor for callers that provide the stencil sum over the neighbours outside a primitive's closure
   // themselves (P2 operators: the vertex-to-vertex sweeps of P2ConstantOperator::smooth_sor_macro_{vertices,edges,faces,cells}
   // see the edge DoFs as well): the sweeps of the point classes in `bits` on every cell's copies with the total weights, `rest`
   // already summed over the cells; and the lexicographic macro-cell sweeps.
   void smooth_sor_shell_given_rest( const P1Function< double >& dst, const P1Function< double >& rhs, const P1Function< double >& rest,
                                     double relax, uint_t level, DoFType flagIn, unsigned bits, bool backwards = false ) const
   {
      launchSorShell( dst, rhs, rest, relax, level, dst.effectiveFlag( flagIn ), bits, backwards );
   }
   void smooth_sor_cells_only( const P1Function< double >& dst, const P1Function< double >& rhs, double relax, uint_t level, DoFType flagIn,
                               bool backwards = false ) const
   {
      launchSorCells( dst, rhs, relax, level, dst.effectiveFlag( flagIn ), backwards );
   }

 private:
   void launchSorCells( const P1Function< double >& dst, const P1Function< double >& rhs, double relax, uint_t level, DoFType flag,
                        bool backwards ) const
   {
      if ( storage_->useBatchSor( level ) )
      {
         const auto masks = storage_->masksFor( flag );
         storage_->forCellChunks( [&]( int first, int count ) {
            const auto u = dst.cellPointers( level, first, count ), r = rhs.cellPointers( level, first, count );
            hipCheck( hyteg_hip_p1_sor_cells( count, u.data(), r.data(), (int) level, stencilTable( level ) + (size_t) first * 225, relax,
                                              backwards ? 1 : 0, masks.data() + first, storage_->stream() ),
                      "smooth_sor: cells (batched)" );
         } );
         return;
      }
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const unsigned mask = storage_->maskFor( cell, flag );
         if ( ( mask & HYTEG_HIP_MASK_INNER ) && level >= HYTEG_HIP_MIN_LEVEL )
            hipCheck( hyteg_hip_p1_sor_cell( dst.getCellPointer( c, level ), rhs.getCellPointer( c, level ), (int) level,
                                             getCellStencils( cell.id, level ).inner, relax, backwards ? 1 : 0, storage_->stream() ),
                      "smooth_sor: cell" );
      } );
   }
   void launchSorShell( const P1Function< double >& dst, const P1Function< double >& rhs, const P1Function< double >& rest, double relax,
                        uint_t level, DoFType flag, unsigned bits, bool backwards ) const
   {
      if ( storage_->useBatchSor( level ) )
      {
         const auto masks = storage_->masksFor( flag, false, bits & HYTEG_HIP_MASK_SHELL );
         storage_->forCellChunks( [&]( int first, int count ) {
            const auto u = dst.cellPointers( level, first, count ), r = rhs.cellPointers( level, first, count ),
                       q = rest.cellPointers( level, first, count );
            hipCheck( hyteg_hip_p1_sor_shell_cells( count, u.data(), r.data(), q.data(), (int) level, shellTable( level ) + first, relax,
                                                    masks.data() + first, backwards ? 1 : 0, storage_->stream() ),
                      "smooth_sor: shell (batched)" );
         } );
         return;
      }
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const auto&    T    = sorTables_.at( level ).at( cell.id );
         const unsigned mask = storage_->maskFor( cell, flag ) & bits;
         hipCheck( hyteg_hip_p1_sor_shell_cell( dst.getCellPointer( c, level ), rhs.getCellPointer( c, level ),
                                                rest.getCellPointer( c, level ), (int) level, &T.edgeVerts[0][0], &T.edgeW[0][0],
                                                &T.faceVerts[0][0], &T.faceW[0][0], T.vertexW, relax, mask, backwards ? 1 : 0,
                                                storage_->stream() ),
                   "smooth_sor: shell" );
      } );
   }

 public:
   // Several functions swept by the SAME launches (no counterpart in the reference, which sweeps one function at a time): the
   // Gauss-Seidel phases are chains of small dependent kernels whose duration does not depend on how many cells a launch
   // covers, and the batched kernels take any list of cell arrays -- so the three velocity components of the Stokes smoother
   // (StokesVelocityBlockBlockDiagonalPreconditioner) share one chain instead of running three.  The batch is ordered
   // [function][cell]; every function sees exactly the kernels, tables and order of smooth_sor: results are bit-identical.
   // Falls back to one smooth_sor per function where the batched path does not apply (large levels, different boundary
   // conditions).
   void smooth_sor_many( const std::vector< std::reference_wrapper< const P1Function< double > > >& dsts,
                         const std::vector< std::reference_wrapper< const P1Function< double > > >& rhss, double relax, uint_t level,
                         DoFType flagIn, bool backwards = false ) const
   {
      const uint_t nf = dsts.size();
      if ( nf == 0 || rhss.size() != nf )
         throw std::runtime_error( "smooth_sor_many: need as many right-hand sides as functions" );
      const DoFType flag = dsts[0].get().effectiveFlag( flagIn );
      bool          together = nf > 1 && storage_->useBatch( level ) && storage_->useBatchSor( level );
      for ( uint_t k = 0; k < nf; ++k )
         together = together && dsts[k].get().effectiveFlag( flagIn ) == flag && &dsts[k].get() != &rhss[k].get();
      if ( !together )
      {
         for ( uint_t k = 0; k < nf; ++k )
            smooth_sor( dsts[k].get(), rhss[k].get(), relax, level, flagIn, backwards );
         return;
      }
      ScopedTimer timerOp( storage_->getTimingTree(), "Operator P1Function to P1Function" ), timerSor( storage_->getTimingTree(), backwards ? "SOR backwards" : "SOR" );
      const int nc    = (int) storage_->getNumberOfLocalCells();
      const int total = nc * (int) nf;
      // chunks of the concatenated [function][cell] list
      auto forChunks = [&]( auto&& fn ) {
         for ( int first = 0; first < total; first += HYTEG_HIP_MAX_BATCH )
            fn( first, std::min( HYTEG_HIP_MAX_BATCH, total - first ) );
      };
      auto pointers = [&]( const std::vector< std::reference_wrapper< const P1Function< double > > >& fs, int first, int count ) {
         std::vector< double* > p;
         for ( int e = first; e < first + count; ++e )
            p.push_back( fs[(uint_t) ( e / nc )].get().getCellPointer( (uint_t) ( e % nc ), level ) );
         return p;
      };
      auto repeatMasks = [&]( const std::vector< unsigned >& once ) {
         std::vector< unsigned > m;
         for ( uint_t k = 0; k < nf; ++k )
            m.insert( m.end(), once.begin(), once.end() );
         return m;
      };
      bool anyShell = false;
      forCells( [&]( uint_t, const MacroCell& cell ) { anyShell = anyShell || ( storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_SHELL ); } );
      auto sweepCells = [&]() {
         const auto    masks = repeatMasks( storage_->masksFor( flag ) );
         const double* table = repeatedTable( stencilTablesRep_, stencilTableHost( level ), level, nf );
         forChunks( [&]( int first, int count ) {
            const auto u = pointers( dsts, first, count ), r = pointers( rhss, first, count );
            hipCheck( hyteg_hip_p1_sor_cells( count, u.data(), r.data(), (int) level, table + (size_t) first * 225, relax, backwards ? 1 : 0,
                                              masks.data() + first, storage_->stream() ),
                      "smooth_sor_many: cells" );
         } );
      };
      if ( !anyShell && storage_->numRanks() == 1 )
      {
         sweepCells();
         return;
      }
      std::vector< std::reference_wrapper< const P1Function< double > > > rests;
      for ( uint_t k = 0; k < nf; ++k )
      {
         auto& slot = sorRestMany_[std::make_pair( level, k )];
         if ( !slot )
            slot.reset( new P1Function< double >( "sor_rest_many", storage_, level, level ) );
         rests.push_back( std::cref( *slot ) );
      }
      auto sweepShell = [&]( unsigned bits ) {
         const auto    masks = repeatMasks( storage_->masksFor( flag, false, bits & HYTEG_HIP_MASK_SHELL ) );
         const double* rt    = repeatedTable( restTablesRep_, restTableHost( level ), level, nf );
         forChunks( [&]( int first, int count ) {
            const auto r = pointers( rests, first, count ), u = pointers( dsts, first, count );
            hipCheck( hyteg_hip_p1_apply_cells( count, r.data(), u.data(), (int) level, rt + (size_t) first * 225, masks.data() + first,
                                                HYTEG_HIP_REPLACE, storage_->stream() ),
                      "smooth_sor_many: rest" );
         } );
         for ( uint_t k = 0; k < nf; ++k )
            rests[k].get().sumSharedCopies( level, flag );
         const hyteg_hip_sor_shell_tables* st = repeatedShellTable( level, nf );
         forChunks( [&]( int first, int count ) {
            const auto u = pointers( dsts, first, count ), r = pointers( rhss, first, count ), q = pointers( rests, first, count );
            hipCheck( hyteg_hip_p1_sor_shell_cells( count, u.data(), r.data(), q.data(), (int) level, st + first, relax, masks.data() + first,
                                                    backwards ? 1 : 0, storage_->stream() ),
                      "smooth_sor_many: shell" );
         } );
      };
      if ( !backwards )
      {
         sweepShell( HYTEG_HIP_MASK_SHELL );
         sweepCells();
      }
      else
      {
         sweepCells();
         sweepShell( 0xFu << 6 );  // macro-faces
         sweepShell( 0x3Fu );      // macro-edges
         sweepShell( 0xFu << 10 ); // macro-vertices
      }
   }

   void smooth_gs( const P1Function< double >& dst, const P1Function< double >& rhs, uint_t level, DoFType flag ) const
   {
      smooth_sor( dst, rhs, 1.0, level, flag, false );
   }
   void smooth_sor_backwards( const P1Function< double >& dst, const P1Function< double >& rhs, double relax, uint_t level, DoFType flag ) const
   {
      smooth_sor( dst, rhs, relax, level, flag, true );
   }

   // P1Operator::computeInverseDiagonalOperatorValues, P1Operator.hpp:461-465, 636-906
   void computeInverseDiagonalOperatorValues()
   {
      inverseDiagonalValues_.reset( new P1Function< double >( "inverse diagonal entries", storage_, minLevel_, maxLevel_ ) );
      for ( uint_t l = minLevel_; l <= maxLevel_; ++l )
         for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         {
            const MacroCell& cell = storage_->getLocalCell( c );
            double*          d    = inverseDiagonalValues_->getCellPointer( c, l );
            hipCheck( hyteg_hip_p1_set_cell_masked( d, 1.0 / getCellStencils( cell.id, l ).inner[stencil::C], (int) l, HYTEG_HIP_MASK_INNER,
                                                    storage_->stream() ),
                      "inverse diagonal" );
            for ( int s = 0; s < 14; ++s )
            {
               // centre weight of a shared DoF = sum of the neighbour cells' shares (the reference adds the per-cell
               // centre entries of faceStencil3D / edgeStencil3D, P1Operator.hpp:700-870)
               const MacroPrimitive& p     = storage_->primitiveOfSlot( cell, s );
               double                total = 0.0;
               for ( int nc : p.cells )
               {
                  const MacroCell& other = storage_->getCells()[nc];
                  total += getCellStencils( nc, l ).slots[slotOf( other, p )][stencil::C];
               }
               hipCheck( hyteg_hip_p1_set_cell_masked( d, 1.0 / total, (int) l, 1u << s, storage_->stream() ), "inverse diagonal" );
            }
         }
   }
   std::shared_ptr< P1Function< double > > getInverseDiagonalValues() const
   {
      if ( !inverseDiagonalValues_ )
         throw std::runtime_error( "Inverse diagonal values have not been assembled, call computeInverseDiagonalOperatorValues() "
                                   "to set up this function." );
      return inverseDiagonalValues_;
   }

   // slot (0..13) under which primitive p appears in cell c
   static int slotOf( const MacroCell& c, const MacroPrimitive& p )
   {
      auto local = [&]( int g ) {
         for ( int q = 0; q < 4; ++q )
            if ( c.v[q] == g )
               return q;
         throw std::runtime_error( "slotOf: primitive is not part of the cell" );
      };
      if ( p.v.size() == 1 )
         return 10 + local( p.v[0] );
      if ( p.v.size() == 2 )
      {
         int a = local( p.v[0] ), b = local( p.v[1] );
         if ( a > b )
            std::swap( a, b );
         for ( int e = 0; e < 6; ++e )
            if ( kCellEdgeVerts[e][0] == a && kCellEdgeVerts[e][1] == b )
               return e;
      }
      int l[3] = { local( p.v[0] ), local( p.v[1] ), local( p.v[2] ) };
      std::sort( l, l + 3 );
      for ( int f = 0; f < 4; ++f )
         if ( kCellFaceVerts[f][0] == l[0] && kCellFaceVerts[f][1] == l[1] && kCellFaceVerts[f][2] == l[2] )
            return 6 + f;
      throw std::runtime_error( "slotOf: not found" );
   }

 private:
   template < typename F >
   void forCells( F&& fn ) const
   {
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         fn( c, storage_->getLocalCell( c ) );
   }
   // Operator::apply with one launch for all local cells: every selected point gets (this cell's share of) its stencil sum,
   // then the shares of the shared points are summed.  Add needs the summed shares in a temporary first.
   void applyBatched( const P1Function< double >& src, const P1Function< double >& dst, uint_t level, DoFType flag, UpdateType updateType ) const
   {
      const bool sharedAdd = updateType == Add && hasSharedPoints( level, flag );
      auto       run       = [&]( const P1Function< double >& out, unsigned keep, int update ) {
         const auto masks = storage_->masksFor( flag, false, keep );
         storage_->forCellChunks( [&]( int first, int count ) {
            const auto d = out.cellPointers( level, first, count ), u = src.cellPointers( level, first, count );
            hipCheck( hyteg_hip_p1_apply_cells( count, d.data(), u.data(), (int) level, stencilTable( level ) + (size_t) first * 225,
                                                masks.data() + first, update, storage_->stream() ),
                      "apply (batched)" );
         } );
      };
      if ( !sharedAdd )
      {
         run( dst, HYTEG_HIP_MASK_ALL, updateType == Add ? HYTEG_HIP_ADD : HYTEG_HIP_REPLACE );
         dst.sumSharedCopies( level, flag );
         return;
      }
      P1Function< double > tmp( "apply_tmp", storage_, level, level, true );
      tmp.interpolate( 0.0, level, All );
      run( dst, HYTEG_HIP_MASK_INNER, HYTEG_HIP_ADD );
      run( tmp, HYTEG_HIP_MASK_SHELL, HYTEG_HIP_REPLACE );
      tmp.sumSharedCopies( level, flag );
      const auto masks = storage_->masksFor( flag, false, HYTEG_HIP_MASK_SHELL );
      storage_->forCellChunks( [&]( int first, int count ) {
         const auto   d = dst.cellPointers( level, first, count ), t = tmp.cellPointers( level, first, count );
         const double one = 1.0;
         hipCheck( hyteg_hip_p1_vector_cells( 1, count, d.data(), 1, t.data(), &one, (int) level, masks.data() + first, storage_->stream() ),
                   "apply: add shell (batched)" );
      } );
   }
 public:
   // ---- the whole CG solve in one launch for problems that fit one workgroup (hyteg_hip_p1_cg_small_cells) ----
   bool canCgSolveSmall( uint_t level ) const
   {
      const size_t n = storage_->getNumberOfLocalCells();
      return storage_->numRanks() == 1 && n >= 1 && n <= HYTEG_HIP_MAX_BATCH &&
             (int64_t) n * layout::cellSize( (int) level ) <= hyteg_hip_p1_cg_small_max_entries();
   }
   void cgSolveSmall( const P1Function< double >& x, const P1Function< double >& b, uint_t level, DoFType flagIn, uint_t maxIter, double relTol,
                      double absTol, double* infoDev ) const
   {
      const DoFType flag = x.effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      const int  count = (int) storage_->getNumberOfLocalCells();
      const auto masks = storage_->masksFor( flag ), owned = storage_->masksFor( flag, true );
      const auto xs = x.cellPointers( level, 0, count ), bs = b.cellPointers( level, 0, count );
      const int* gp[2] = { nullptr, nullptr }, *ec[2] = { nullptr, nullptr }, *eo[2] = { nullptr, nullptr };
      int        ng[2] = { 0, 0 };
      for ( int cls = 0; cls < 2; ++cls )
      {
         if ( !testFlag( storage_->boundaryTypeOf( cls == 1 ), flag ) || storage_->exchangePlan( (int) level, cls ).ngroups() == 0 )
            continue;
         const auto& plan = storage_->devicePlan( (int) level, cls );
         gp[cls] = plan.dGroupPtr, ec[cls] = plan.dEntryBuf, eo[cls] = plan.dEntryOff, ng[cls] = plan.ngroups();
      }
      hipCheck( hyteg_hip_p1_cg_small_cells( count, xs.data(), bs.data(), (int) level, stencilTable( level ), masks.data(), owned.data(), gp, ec,
                                             eo, ng, (int) maxIter, relTol, absTol, infoDev, storage_->stream() ),
                "cgSolveSmall" );
   }

 private:
   // the per-cell tables of the batched kernels repeated nf times (smooth_sor_many: the batch is [function][cell])
   const double* repeatedTable( std::map< std::pair< uint_t, uint_t >, const double* >& cache, const std::vector< double >& once, uint_t level,
                                uint_t nf ) const
   {
      auto key = std::make_pair( level, nf );
      auto it  = cache.find( key );
      if ( it != cache.end() )
         return it->second;
      std::vector< double > h;
      for ( uint_t k = 0; k < nf; ++k )
         h.insert( h.end(), once.begin(), once.end() );
      return cache[key] = storage_->uploadTable( h );
   }
   std::vector< double > stencilTableHost( uint_t level ) const
   {
      std::vector< double > h;
      for ( int id : storage_->getLocalCellIDs() )
      {
         const auto& S = getCellStencils( id, level );
         h.insert( h.end(), &S.slots[0][0], &S.slots[0][0] + 14 * 15 );
         h.insert( h.end(), S.inner, S.inner + 15 );
      }
      return h;
   }
   std::vector< double > restTableHost( uint_t level ) const
   {
      std::vector< double > h;
      for ( int id : storage_->getLocalCellIDs() )
      {
         const auto& T = sorTables_.at( level ).at( id );
         h.insert( h.end(), &T.rest[0][0], &T.rest[0][0] + 14 * 15 );
         h.insert( h.end(), 15, 0.0 );
      }
      return h;
   }
   std::vector< hyteg_hip_sor_shell_tables > shellTableHost( uint_t level ) const
   {
      std::vector< hyteg_hip_sor_shell_tables > h;
      for ( int id : storage_->getLocalCellIDs() )
      {
         const auto&                T = sorTables_.at( level ).at( id );
         hyteg_hip_sor_shell_tables r{};
         std::memcpy( r.edge_verts, T.edgeVerts, sizeof( r.edge_verts ) );
         std::memcpy( r.face_verts, T.faceVerts, sizeof( r.face_verts ) );
         std::memcpy( r.edge_w, T.edgeW, sizeof( r.edge_w ) );
         std::memcpy( r.face_w, T.faceW, sizeof( r.face_w ) );
         std::memcpy( r.vertex_w, T.vertexW, sizeof( r.vertex_w ) );
         h.push_back( r );
      }
      return h;
   }
   const hyteg_hip_sor_shell_tables* repeatedShellTable( uint_t level, uint_t nf ) const
   {
      auto key = std::make_pair( level, nf );
      auto it  = shellTablesRep_.find( key );
      if ( it != shellTablesRep_.end() )
         return it->second;
      const auto                                once = shellTableHost( level );
      std::vector< hyteg_hip_sor_shell_tables > h;
      for ( uint_t k = 0; k < nf; ++k )
         h.insert( h.end(), once.begin(), once.end() );
      return shellTablesRep_[key] =
                 static_cast< const hyteg_hip_sor_shell_tables* >( storage_->uploadBytes( h.data(), h.size() * sizeof( h[0] ) ) );
   }
   // device tables [local cell][15 point classes][15 weights] for the batched kernels: classes 0..13 the cell's shares, 14 inner
   const double* stencilTable( uint_t level ) const
   {
      auto it = stencilTables_.find( level );
      if ( it != stencilTables_.end() )
         return it->second;
      std::vector< double > h;
      for ( int id : storage_->getLocalCellIDs() )
      {
         const auto& S = getCellStencils( id, level );
         h.insert( h.end(), &S.slots[0][0], &S.slots[0][0] + 14 * 15 );
         h.insert( h.end(), S.inner, S.inner + 15 );
      }
      return stencilTables_[level] = storage_->uploadTable( h );
   }
   const hyteg_hip_sor_shell_tables* shellTable( uint_t level ) const
   {
      auto it = shellTables_.find( level );
      if ( it != shellTables_.end() )
         return it->second;
      std::vector< hyteg_hip_sor_shell_tables > h;
      for ( int id : storage_->getLocalCellIDs() )
      {
         const auto&                T = sorTables_.at( level ).at( id );
         hyteg_hip_sor_shell_tables r;
         std::memcpy( r.edge_verts, T.edgeVerts, sizeof( r.edge_verts ) );
         std::memcpy( r.face_verts, T.faceVerts, sizeof( r.face_verts ) );
         std::memcpy( r.edge_w, T.edgeW, sizeof( r.edge_w ) );
         std::memcpy( r.face_w, T.faceW, sizeof( r.face_w ) );
         std::memcpy( r.vertex_w, T.vertexW, sizeof( r.vertex_w ) );
         h.push_back( r );
      }
      return shellTables_[level] =
                 static_cast< const hyteg_hip_sor_shell_tables* >( storage_->uploadBytes( h.data(), h.size() * sizeof( h[0] ) ) );
   }
   const double* restTable( uint_t level ) const
   {
      auto it = restTables_.find( level );
      if ( it != restTables_.end() )
         return it->second;
      std::vector< double > h;
      for ( int id : storage_->getLocalCellIDs() )
      {
         const auto& T = sorTables_.at( level ).at( id );
         h.insert( h.end(), &T.rest[0][0], &T.rest[0][0] + 14 * 15 );
         h.insert( h.end(), 15, 0.0 );
      }
      return restTables_[level] = storage_->uploadTable( h );
   }
   // total weights and sweep orientations of every macro-primitive, handed to each adjacent cell in its local numbering
   std::vector< stencil::CellSorTables > buildSorTables( const std::vector< stencil::CellStencils >& S ) const
   {
      using namespace stencil;
      const auto&                   cells = storage_->getCells();
      std::vector< CellSorTables >  T( cells.size() );
      std::vector< std::array< double, 3 > > edgeTot( storage_->getEdges().size(), std::array< double, 3 >{} );
      std::vector< std::array< double, 7 > > faceTot( storage_->getFaces().size(), std::array< double, 7 >{} );
      std::vector< double >                  vertTot( storage_->getVertices().size(), 0.0 );
      for ( const auto& c : cells )
      {
         CellSorTables& t = T[c.id];
         for ( int s = 0; s < 14; ++s )
            for ( int k = 0; k < 15; ++k )
               t.rest[s][k] = k == C ? 0.0 : S[c.id].slots[s][k];
         for ( int k = 0; k < 4; ++k )
            vertTot[c.v[k]] += S[c.id].slots[10 + k][C];
         for ( int e = 0; e < 6; ++e )
         {
            int lo = kCellEdgeVerts[e][0], hi = kCellEdgeVerts[e][1];
            if ( c.v[lo] > c.v[hi] )
               std::swap( lo, hi );
            const int kp = offsetIndex( kUnit[hi][0] - kUnit[lo][0], kUnit[hi][1] - kUnit[lo][1], kUnit[hi][2] - kUnit[lo][2] );
            const int km = offsetIndex( kUnit[lo][0] - kUnit[hi][0], kUnit[lo][1] - kUnit[hi][1], kUnit[lo][2] - kUnit[hi][2] );
            t.edgeVerts[e][0] = lo, t.edgeVerts[e][1] = hi;
            auto& tot = edgeTot[c.edges[e]];
            tot[0] += S[c.id].slots[e][C], tot[1] += S[c.id].slots[e][km], tot[2] += S[c.id].slots[e][kp];
            t.rest[e][km] = t.rest[e][kp] = 0.0;
         }
         for ( int f = 0; f < 4; ++f )
         {
            int l[3] = { kCellFaceVerts[f][0], kCellFaceVerts[f][1], kCellFaceVerts[f][2] };
            std::sort( l, l + 3, [&]( int a, int b ) { return c.v[a] < c.v[b]; } );
            auto& tot = faceTot[c.faces[f]];
            tot[0] += S[c.id].slots[6 + f][C];
            for ( int d = 0; d < 6; ++d )
            {
               int o[3];
               for ( int r = 0; r < 3; ++r )
                  o[r] = kFaceDirs[d][0] * ( kUnit[l[1]][r] - kUnit[l[0]][r] ) + kFaceDirs[d][1] * ( kUnit[l[2]][r] - kUnit[l[0]][r] );
               const int k = offsetIndex( o[0], o[1], o[2] );
               tot[1 + d] += S[c.id].slots[6 + f][k];
               t.rest[6 + f][k] = 0.0;
            }
            for ( int r = 0; r < 3; ++r )
               t.faceVerts[f][r] = l[r];
         }
      }
      for ( const auto& c : cells )
      {
         CellSorTables& t = T[c.id];
         for ( int k = 0; k < 4; ++k )
            t.vertexW[k] = vertTot[c.v[k]];
         for ( int e = 0; e < 6; ++e )
            for ( int k = 0; k < 3; ++k )
               t.edgeW[e][k] = edgeTot[c.edges[e]][k];
         for ( int f = 0; f < 4; ++f )
            for ( int k = 0; k < 7; ++k )
               t.faceW[f][k] = faceTot[c.faces[f]][k];
      }
      return T;
   }
   bool hasSharedPoints( uint_t level, DoFType flag ) const
   {
      for ( int cls = 0; cls < 2; ++cls )
         if ( testFlag( storage_->boundaryTypeOf( cls == 1 ), flag ) && storage_->exchangePlan( (int) level, cls ).ngroups() > 0 )
            return true;
      return false;
   }

   std::shared_ptr< PrimitiveStorage >                        storage_;
   uint_t                                                     minLevel_, maxLevel_;
   uint64_t                                                   uid_ = nextUid();
   std::map< uint_t, std::vector< stencil::CellStencils > >   stencils_;
   std::map< uint_t, std::vector< stencil::CellSorTables > >  sorTables_;
   mutable std::map< uint_t, std::unique_ptr< P1Function< double > > > sorRest_;
   mutable std::map< std::pair< uint_t, uint_t >, std::unique_ptr< P1Function< double > > > sorRestMany_; // (level, k)
   mutable std::map< std::pair< uint_t, uint_t >, const double* >                         stencilTablesRep_, restTablesRep_;
   mutable std::map< std::pair< uint_t, uint_t >, const hyteg_hip_sor_shell_tables* >     shellTablesRep_;
   mutable std::map< uint_t, const double* >                  stencilTables_, restTables_;
   mutable std::map< uint_t, const hyteg_hip_sor_shell_tables* > shellTables_;
   std::shared_ptr< P1Function< double > >                    inverseDiagonalValues_;
};

using P1ConstantLaplaceOperator = P1ConstantOperator< forms::P1LaplaceForm >; // P1ConstantOperator.hpp:167-168
using P1ConstantMassOperator    = P1ConstantOperator< forms::P1MassForm >;

} // namespace hyteg
