// gridtransfer.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// P1toP1LinearRestriction / P1toP1LinearProlongation (src/hyteg/gridtransferoperators/)
#pragma once

#include "p1function.hpp"

namespace hyteg {

// =====================================================================================================
// Grid transfer ( src/hyteg/gridtransferoperators/P1toP1LinearRestriction.cpp:169-346, P1toP1LinearProlongation.cpp:194-410 )
// =====================================================================================================
class P1toP1LinearRestriction
{
 public:
   void restrict( const P1Function< double >& function, const uint_t& sourceLevel, const DoFType& flagIn ) const
   {
      restrictInto( function, function, sourceLevel, flagIn );
   }
   // restriction of `fineFunction` at sourceLevel into `function` at sourceLevel - 1: what a multigrid cycle does with
   // restrict( tmp ) followed by b.assign( { 1 }, { tmp }, sourceLevel - 1 ) (GeometricMultigridSolver.hpp:262-266), without the copy
   void restrictInto( const P1Function< double >& fineFunction, const P1Function< double >& function, const uint_t& sourceLevel,
                      const DoFType& flagIn ) const
   {
      ScopedTimer timerRestrict( function.getStorage()->getTimingTree(), "P1toP1LinearRestriction" );
      const DoFType flag = function.effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      auto         storage = function.getStorage();
      const uint_t dstLevel = sourceLevel - 1;
      if ( storage->useBatch( sourceLevel ) )
      {
         const auto masks = storage->masksFor( flag );
         storage->forCellChunks( [&]( int first, int count ) {
            const auto co = function.cellPointers( dstLevel, first, count ), fi = fineFunction.cellPointers( sourceLevel, first, count );
            hipCheck( hyteg_hip_p1_restrict_cells( count, co.data(), fi.data(), (int) dstLevel, storage->nncInvDevice() + (size_t) first * 14,
                                                   masks.data() + first, storage->stream() ),
                      "restrict (batched)" );
         } );
         function.sumSharedCopies( dstLevel, flag );
         return;
      }
      for ( uint_t c = 0; c < storage->getNumberOfLocalCells(); ++c )
      {
         const MacroCell& cell = storage->getLocalCell( c );
         const auto       nnc  = storage->numNeighborCells( cell );
         hipCheck( hyteg_hip_p1_restrict_cell_masked( function.getCellPointer( c, dstLevel ), fineFunction.getCellPointer( c, sourceLevel ),
                                                      (int) dstLevel, nnc.data(), storage->maskFor( cell, flag ), storage->stream() ),
                   "restrict" );
      }
      // communicateAdditively< Cell, {Vertex,Edge,Face} >( dstLevel, flag ^ All, ... ) (:343-345)
      function.sumSharedCopies( dstLevel, flag );
   }
};

class P1toP1LinearProlongation
{
 public:
   void prolongate( const P1Function< double >& function, const uint_t& sourceLevel, const DoFType& flag ) const
   {
      run( function, function, sourceLevel, flag );
   }
   void prolongateAndAdd( const P1Function< double >& function, const uint_t& sourceLevel, const DoFType& flagIn ) const
   {
      const DoFType& flag = flagIn;
      // the prolongated correction is formed in a temporary (Replace), summed over cells on shared points, then added.
      // The temporary needs no initialisation: the masked kernel writes every point `flag` selects, the sum over
      // shared copies and the add read only those.
      auto                 storage = function.getStorage();
      // no shell point selected (a macro-cell whose boundary values are fixed): nothing is shared, the kernel adds in place
      {
         const DoFType flag     = function.effectiveFlag( flagIn );
         bool          anyShell = storage->numRanks() > 1 || storage->useBatch( sourceLevel + 1 );
         for ( uint_t c = 0; c < storage->getNumberOfLocalCells() && !anyShell; ++c )
            anyShell = ( storage->maskFor( storage->getLocalCell( c ), flag ) & HYTEG_HIP_MASK_SHELL ) != 0;
         if ( !anyShell )
         {
            ScopedTimer timerProlongate( storage->getTimingTree(), "P1toP1LinearProlongation" );
            for ( uint_t c = 0; c < storage->getNumberOfLocalCells(); ++c )
            {
               const MacroCell& cell = storage->getLocalCell( c );
               const auto       nnc  = storage->numNeighborCells( cell );
               hipCheck( hyteg_hip_p1_prolongate_cell_masked_update( function.getCellPointer( c, sourceLevel ),
                                                                     function.getCellPointer( c, sourceLevel + 1 ), (int) sourceLevel,
                                                                     nnc.data(), storage->maskFor( cell, flag ), HYTEG_HIP_ADD, storage->stream() ),
                         "prolongateAndAdd" );
            }
            return;
         }
      }
      P1Function< double > tmp( "prolongate_tmp", storage, sourceLevel + 1, sourceLevel + 1, true );
      tmp.setBoundaryConditionAllInner( function.hasAllInnerBoundaryCondition() );
      run( function, tmp, sourceLevel, flag );
      function.add( { 1.0 }, { tmp }, sourceLevel + 1, flag );
   }

 private:
   static void run( const P1Function< double >& src, const P1Function< double >& dst, uint_t sourceLevel, DoFType flagIn )
   {
      ScopedTimer timerProlongate( src.getStorage()->getTimingTree(), "P1toP1LinearProlongation" );
      const DoFType flag = dst.effectiveFlag( flagIn ); // the function's boundary condition decides what `Inner` means
      auto storage = src.getStorage();
      if ( storage->useBatch( sourceLevel + 1 ) )
      {
         const auto masks = storage->masksFor( flag );
         storage->forCellChunks( [&]( int first, int count ) {
            const auto co = src.cellPointers( sourceLevel, first, count ), fi = dst.cellPointers( sourceLevel + 1, first, count );
            hipCheck( hyteg_hip_p1_prolongate_cells( count, co.data(), fi.data(), (int) sourceLevel,
                                                     storage->nncInvDevice() + (size_t) first * 14, masks.data() + first, HYTEG_HIP_REPLACE,
                                                     storage->stream() ),
                      "prolongate (batched)" );
         } );
         dst.sumSharedCopies( sourceLevel + 1, flag );
         return;
      }
      for ( uint_t c = 0; c < storage->getNumberOfLocalCells(); ++c )
      {
         const MacroCell& cell = storage->getLocalCell( c );
         const auto       nnc  = storage->numNeighborCells( cell );
         hipCheck( hyteg_hip_p1_prolongate_cell_masked( src.getCellPointer( c, sourceLevel ), dst.getCellPointer( c, sourceLevel + 1 ),
                                                        (int) sourceLevel, nnc.data(), storage->maskFor( cell, flag ), storage->stream() ),
                   "prolongate" );
      }
      dst.sumSharedCopies( sourceLevel + 1, flag );
   }
};

} // namespace hyteg
