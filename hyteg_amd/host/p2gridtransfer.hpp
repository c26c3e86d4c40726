// p2gridtransfer.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// P2toP2QuadraticProlongation / P2toP2QuadraticRestriction (src/hyteg/gridtransferoperators/P2toP2QuadraticProlongation.cpp:
// 37-64,217-424; P2toP2QuadraticRestriction.cpp:35-47,131-286)
#pragma once

#include <map>
#include <memory>

#include "p2function.hpp"

namespace hyteg {

class P2toP2QuadraticProlongation
{
 public:
   // fine := quadratic interpolant of the coarse function on the DoFs `flag` selects
   void prolongate( const P2Function< double >& function, const uint_t& sourceLevel, const DoFType& flag ) const
   {
      run( function, function, sourceLevel, flag );
   }
   // fine += interpolant: formed in a temporary (Replace), made bit-identical on shared DoFs, then added
   void prolongateAndAdd( const P2Function< double >& function, const uint_t& sourceLevel, const DoFType& flag ) const
   {
      // the temporary is kept per (storage, level): creating it allocates and clears an array pair per macro-cell, and a multigrid
      // cycle comes through here once per level and function
      auto& slot = tmp_[std::make_pair( function.getStorage().get(), sourceLevel + 1 )];
      if ( !slot )
         slot.reset( new P2Function< double >( "p2_prolongate_tmp", function.getStorage(), sourceLevel + 1, sourceLevel + 1 ) );
      run( function, *slot, sourceLevel, flag );
      function.add( { 1.0 }, { *slot }, sourceLevel + 1, flag );
   }

 private:
   mutable std::map< std::pair< const PrimitiveStorage*, uint_t >, std::unique_ptr< P2Function< double > > > tmp_;
   static void run( const P2Function< double >& src, const P2Function< double >& dst, uint_t sourceLevel, DoFType flag )
   {
      auto        storage = src.getStorage();
      ScopedTimer timer( storage->getTimingTree(), "P2toP2QuadraticProlongation" );
      for ( uint_t c = 0; c < storage->getNumberOfLocalCells(); ++c )
      {
         const MacroCell& cell = storage->getLocalCell( c );
         hipCheck( hyteg_hip_p2_prolongate_cell( dst.getVertexDoFFunction().getCellPointer( c, sourceLevel + 1 ),
                                                 dst.getEdgeCellPointer( c, sourceLevel + 1 ),
                                                 src.getVertexDoFFunction().getCellPointer( c, sourceLevel ),
                                                 src.getEdgeCellPointer( c, sourceLevel ), (int) sourceLevel, HYTEG_HIP_REPLACE,
                                                 storage->maskFor( cell, flag ), storage->stream() ),
                   "P2toP2QuadraticProlongation" );
      }
      // every cell has computed the complete value of the DoFs on its boundary from its own coarse DoFs; the values of the
      // neighbour cells agree up to rounding (different containing micro-cells): one copy wins, so that all copies of a
      // shared DoF stay bit-identical (the reference reaches the same state with scaled contributions + additive exchange)
      dst.getVertexDoFFunction().syncSharedCopies( sourceLevel + 1, flag );
      dst.syncSharedEdgeCopies( sourceLevel + 1, flag );
   }
};

class P2toP2QuadraticRestriction
{
 public:
   // coarse := P^T fine on the DoFs `flag` selects (level sourceLevel -> sourceLevel - 1)
   void restrict( const P2Function< double >& function, const uint_t& sourceLevel, const DoFType& flag ) const
   {
      auto         storage  = function.getStorage();
      ScopedTimer  timer( storage->getTimingTree(), "P2toP2QuadraticRestriction" );
      const uint_t dstLevel = sourceLevel - 1;
      for ( uint_t c = 0; c < storage->getNumberOfLocalCells(); ++c )
      {
         const MacroCell& cell = storage->getLocalCell( c );
         const auto       nnc  = storage->numNeighborCells( cell );
         hipCheck( hyteg_hip_p2_restrict_cell( function.getVertexDoFFunction().getCellPointer( c, dstLevel ),
                                               function.getEdgeCellPointer( c, dstLevel ),
                                               function.getVertexDoFFunction().getCellPointer( c, sourceLevel ),
                                               function.getEdgeCellPointer( c, sourceLevel ), (int) dstLevel, nnc.data(),
                                               storage->maskFor( cell, flag ), storage->stream() ),
                   "P2toP2QuadraticRestriction" );
      }
      // communicateAdditively< Cell, {Face, Edge, Vertex} > of both DoF kinds (P2toP2QuadraticRestriction.cpp:276-285)
      function.getVertexDoFFunction().sumSharedCopies( dstLevel, flag );
      function.sumSharedEdgeCopies( dstLevel, flag );
   }
};

} // namespace hyteg
