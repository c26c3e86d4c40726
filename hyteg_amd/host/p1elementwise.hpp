// p1elementwise.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// P1ElementwiseDiffusion: the operator class the hyteg_operators generator emits (module hyteg_operators; sample in the tree:
// apps/2023-zikeli-mt/MT-apps/operators-used/P1ElementwiseDiffusion_cubes_const_float64.hpp:64-93), which derives from
// Operator< P1Function, P1Function > and OperatorWithInverseDiagonal< P1Function > and whose apply() loops over the macro-cells
// and hands each one's arrays, its twelve vertex coordinates and micro_edges_per_macro_edge to apply_macro_3D (.cpp:76-165).
// Here that per-cell call is the C-ABI seam hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_masked: this class never sees a
// stencil.  The in-tree equivalent is P1ElementwiseLaplaceOperator (src/hyteg/elementwiseoperators/P1ElementwiseOperator.hpp);
// its smooth_jac (P1ElementwiseOperator.cpp:288-331) is mirrored below.
#pragma once

#include "p1function.hpp"

namespace hyteg {
namespace operatorgeneration {

class P1ElementwiseDiffusion
{
 public:
   using srcType = P1Function< double >;
   using dstType = P1Function< double >;

   P1ElementwiseDiffusion( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : storage_( storage )
   , minLevel_( minLevel )
   , maxLevel_( maxLevel )
   {
      for ( uint_t l = std::max< uint_t >( minLevel, HYTEG_HIP_MIN_LEVEL ); l <= maxLevel; ++l )
         hipCheck( hyteg_hip_prepare_level( (int) l ), "P1ElementwiseDiffusion: prepare_level" );
   }
   std::shared_ptr< PrimitiveStorage > getStorage() const { return storage_; }
   uint_t                              getMinLevel() const { return minLevel_; }
   uint_t                              getMaxLevel() const { return maxLevel_; }

   // P1ElementwiseDiffusion_cubes_const_float64::apply, .cpp:55-165.  The reference synchronises the halos of src, zeroes dst
   // (flagged points and the cells' halos), runs the kernel on every cell and communicates additively with the BC exclusion
   // All ^ flag.  Cell-centric storage: every copy of a shared DoF holds the same value (nothing to synchronise), a cell's
   // boundary entries are the DoFs themselves, so the kernel is restricted to the flagged point classes (Replace writes them,
   // so no zeroing pass), the shares of shared points are summed by the additive exchange while the inner points are computed.
   void apply( const P1Function< double >& src, const P1Function< double >& dst, uint_t level, DoFType flagIn, UpdateType updateType = Replace ) const
   {
      ScopedTimer timerOp( storage_->getTimingTree(), "Operator P1Function to P1Function" ), timerApply( storage_->getTimingTree(), "apply" );
      const DoFType flag = dst.effectiveFlag( flagIn );
      if ( &src == &dst )
         throw std::runtime_error( "P1ElementwiseDiffusion::apply: src and dst must differ" );
      const int64_t n = int64_t( 1 ) << level;
      bool          anyShell = false;
      forCells( [&]( uint_t, const MacroCell& cell ) { anyShell = anyShell || ( storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_SHELL ); } );
      const P1Function< double >*             shellDst = &dst;
      std::unique_ptr< P1Function< double > > tmp;
      if ( updateType == Add && anyShell && ( storage_->getCells().size() > 1 ) )
      {
         // the shares of a shared DoF are summed over the cells before the sum is added to dst
         tmp.reset( new P1Function< double >( "apply_tmp", storage_, level, level, true ) );
         tmp->interpolate( 0.0, level, All );
         shellDst = tmp.get();
      }
      auto kernel = [&]( const P1Function< double >& out, uint_t c, const MacroCell& cell, unsigned mask, int update ) {
         if ( mask == 0 )
            return;
         double cc[12];
         for ( int v = 0; v < 4; ++v )
            for ( int r = 0; r < 3; ++r )
               cc[3 * v + r] = cell.coords[v][r];
         hipCheck( hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_masked( out.getCellPointer( c, level ), src.getCellPointer( c, level ), cc, n, mask,
                                                                             update, storage_->stream() ),
                   "P1ElementwiseDiffusion::apply: apply_macro_3D" );
      };
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         kernel( *shellDst, c, cell, storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_SHELL,
                 ( updateType == Add && shellDst == &dst ) ? HYTEG_HIP_ADD : HYTEG_HIP_REPLACE );
      } );
      shellDst->beginSumSharedCopies( level, flag );
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         kernel( dst, c, cell, storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_INNER, updateType == Replace ? HYTEG_HIP_REPLACE : HYTEG_HIP_ADD );
      } );
      shellDst->endSumSharedCopies( level, flag );
      if ( shellDst != &dst )
         forCells( [&]( uint_t c, const MacroCell& cell ) {
            const double* srcs[1] = { shellDst->getCellPointer( c, level ) };
            const double  one[1]  = { 1.0 };
            hipCheck( hyteg_hip_p1_vector_cell_masked( 1, dst.getCellPointer( c, level ), 1, srcs, one, (int) level,
                                                       storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_SHELL, storage_->stream() ),
                      "P1ElementwiseDiffusion::apply: add shell" );
         } );
   }

   // computeInverseDiagonalOperatorValues, .cpp (same file): invDiag_ = 0; per cell
   // computeInverseDiagonalOperatorValues_macro_3D( invDiag_, coordinates, micro_edges ); additive communication; invertElementwise
   void computeInverseDiagonalOperatorValues()
   {
      invDiag_.reset( new P1Function< double >( "inverse diagonal entries", storage_, minLevel_, maxLevel_ ) );
      for ( uint_t l = minLevel_; l <= maxLevel_; ++l )
      {
         invDiag_->interpolate( 0.0, l, All );
         forCells( [&]( uint_t c, const MacroCell& cell ) {
            double cc[12];
            for ( int v = 0; v < 4; ++v )
               for ( int r = 0; r < 3; ++r )
                  cc[3 * v + r] = cell.coords[v][r];
            const int64_t n = int64_t( 1 ) << l;
            hipCheck( hyteg_hip_p1_elementwise_diffusion_diagonal_macro_3d( invDiag_->getCellPointer( c, l ), cc, n, double( n ), storage_->stream() ),
                      "P1ElementwiseDiffusion: diagonal" );
         } );
         invDiag_->sumSharedCopies( l, All );
         // invertElementwise: set-up time, on the host
         std::vector< double > h( (size_t) hyteg_hip_cell_size( (int) l ) );
         forCells( [&]( uint_t c, const MacroCell& ) {
            invDiag_->copyCellToHost( c, l, h.data() );
            for ( double& v : h )
               v = v != 0.0 ? 1.0 / v : 0.0;
            invDiag_->copyCellFromHost( c, l, h.data() );
         } );
      }
   }
   std::shared_ptr< P1Function< double > > getInverseDiagonalValues() const
   {
      if ( !invDiag_ )
         throw std::runtime_error( "Inverse diagonal values have not been assembled, call computeInverseDiagonalOperatorValues() "
                                   "to set up this function." );
      return invDiag_;
   }

   // P1ElementwiseOperator::smooth_jac, src/hyteg/elementwiseoperators/P1ElementwiseOperator.cpp:288-331:
   //   dst = A src ; dst = rhs - dst ; dst = invDiag .* dst ; dst = src + omega dst      (four passes, as the reference writes it)
   void smooth_jac( const P1Function< double >& dst, const P1Function< double >& rhs, const P1Function< double >& src, double omega, uint_t level,
                    DoFType flag ) const
   {
      ScopedTimer timerOp( storage_->getTimingTree(), "Operator P1Function to P1Function" ), timerJac( storage_->getTimingTree(), "smooth_jac" );
      apply( src, dst, level, flag, Replace );
      dst.assign( { 1.0, -1.0 }, { rhs, dst }, level, flag );
      dst.multElementwise( { *getInverseDiagonalValues(), dst }, level, flag );
      dst.assign( { 1.0, omega }, { src, dst }, level, flag );
   }

 private:
   template < typename F >
   void forCells( F&& fn ) const
   {
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         fn( c, storage_->getLocalCell( c ) );
   }
   std::shared_ptr< PrimitiveStorage >     storage_;
   uint_t                                  minLevel_, maxLevel_;
   std::shared_ptr< P1Function< double > > invDiag_;
};

} // namespace operatorgeneration
} // namespace hyteg
