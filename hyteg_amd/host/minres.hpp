// minres.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// MinResSolver (src/hyteg/solvers/MinresSolver.hpp:33-290) and the preconditioners stokesSphere composes around it
// (apps/stokesSphere/StokesSphere.cpp:200-260): StokesPressureBlockPreconditioner
// (solvers/preconditioners/stokes/StokesPressureBlockPreconditioner.hpp:29-48), StokesBlockDiagonalPreconditioner
// (…/StokesBlockDiagonalPreconditioner.hpp), JacobiPreconditioner (solvers/preconditioners/JacobiPreconditioner.hpp) and the
// pressure-block operator P1LumpedInvMassOperator (constant_stencil_operator/P1ConstantOperator.hpp:197-202).
// MINRES needs apply, assign / add and global dot products only, so it runs on any number of ranks: it is the coarse-grid
// solver of the distributed Stokes V-cycle (BASELINE config 5), where the dense LU stand-in of stokes.hpp is single-rank.
#pragma once

#include <cmath>
#include <limits>

#include "solvers.hpp"

namespace hyteg {

// IdentityPreconditioner (solvers/preconditioners/IdentityPreconditioner.hpp): x = b
template < class OperatorType >
class IdentityPreconditioner : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType;
   void solve( const OperatorType&, const FunctionType& x, const FunctionType& b, uint_t level ) override { x.assign( { 1.0 }, { b }, level, All ); }
};

// JacobiPreconditioner (solvers/preconditioners/JacobiPreconditioner.hpp:31-58): x = b, then `iterations` Jacobi steps
template < class OperatorType >
class JacobiPreconditioner : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType;
   JacobiPreconditioner( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel, uint_t iterations )
   : iterations_( iterations )
   , tmp_( "jac_tmp", storage, minLevel, maxLevel )
   , flag_( Inner | NeumannBoundary )
   {}
   void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) override
   {
      x.assign( { 1.0 }, { b }, level, flag_ );
      for ( uint_t i = 0; i < iterations_; ++i )
      {
         tmp_.assign( { 1.0 }, { x }, level, flag_ );
         A.smooth_jac( x, b, tmp_, 1.0, level, flag_ );
      }
   }

 private:
   uint_t       iterations_;
   FunctionType tmp_;
   DoFType      flag_;
};

// P1LumpedInvMassOperator = P1ConstantOperator< mass form, Diagonal = false, Lumped = true, InvertDiagonal = true >
// (P1ConstantOperator.hpp:197-202): the stencil's row sum on the centre, inverted (P1Operator.hpp:2128-2158; on points shared
// between macro-cells the inverse of the SUM of the cells' row sums).  = pointwise multiplication with 1 / ( M 1 ).
class P1LumpedInvMassOperator
{
 public:
   P1LumpedInvMassOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : invLumped_( "lumped_inv_mass", storage, minLevel, maxLevel )
   , tmp_( "lumped_inv_mass_tmp", storage, minLevel, maxLevel )
   {
      P1ConstantMassOperator mass( storage, minLevel, maxLevel );
      P1Function< double >   one( "lumped_inv_mass_one", storage, minLevel, maxLevel );
      std::vector< double >  h;
      for ( uint_t l = minLevel; l <= maxLevel; ++l )
      {
         one.interpolate( 1.0, l, All );
         mass.apply( one, invLumped_, l, All, Replace ); // row sums; the shares of shared points are summed by the apply
         h.resize( (size_t) hyteg_hip_cell_size( (int) l ) );
         for ( uint_t c = 0; c < storage->getNumberOfLocalCells(); ++c )
         {
            invLumped_.copyCellToHost( c, l, h.data() );
            for ( double& v : h )
               v = v != 0.0 ? 1.0 / v : 0.0;
            invLumped_.copyCellFromHost( c, l, h.data() );
         }
      }
   }
   void apply( const P1Function< double >& src, const P1Function< double >& dst, uint_t level, DoFType flag, UpdateType updateType = Replace ) const
   {
      if ( updateType == Replace )
      {
         dst.multElementwise( { invLumped_, src }, level, flag );
         return;
      }
      tmp_.multElementwise( { invLumped_, src }, level, flag );
      dst.add( { 1.0 }, { tmp_ }, level, flag );
   }
   const P1Function< double >& getInverseLumpedMass() const { return invLumped_; }

 private:
   P1Function< double > invLumped_, tmp_;
};

// StokesPressureBlockPreconditioner.hpp:29-48: x = b, then x.p = pressureBlockPreconditioner( b.p )
template < class OperatorType, class pressureBlockPreconditionerType >
class StokesPressureBlockPreconditioner : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType;
   StokesPressureBlockPreconditioner( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : pressureBlockPreconditioner_( std::make_shared< pressureBlockPreconditionerType >( storage, minLevel, maxLevel ) )
   , flag_( Inner | NeumannBoundary | FreeslipBoundary )
   {}
   void solve( const OperatorType&, const FunctionType& x, const FunctionType& b, uint_t level ) override
   {
      x.assign( { 1.0 }, { b }, level, flag_ );
      pressureBlockPreconditioner_->apply( b.p(), x.p(), level, flag_, Replace );
   }

 private:
   std::shared_ptr< pressureBlockPreconditionerType > pressureBlockPreconditioner_;
   DoFType                                            flag_;
};

// StokesBlockDiagonalPreconditioner.hpp:28-69: `velocityPreconditionSteps` times: the velocity block preconditioner (a multigrid
// cycle of the scalar Laplace operator) on every velocity component, then the pressure block preconditioner on p
template < class OperatorType, class pressureBlockPreconditionerType >
class StokesBlockDiagonalPreconditioner : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType;
   StokesBlockDiagonalPreconditioner( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel, uint_t velocityPreconditionSteps,
                                      std::shared_ptr< Solver< typename OperatorType::VelocityOperator_T > > velocityBlockPreconditioner =
                                          std::make_shared< IdentityPreconditioner< typename OperatorType::VelocityOperator_T > >() )
   : velocityPreconditionSteps_( velocityPreconditionSteps )
   , flag_( Inner | NeumannBoundary )
   , velocityBlockPreconditioner_( std::move( velocityBlockPreconditioner ) )
   , pressureBlockPreconditioner_( std::make_shared< pressureBlockPreconditionerType >( storage, minLevel, maxLevel ) )
   {}
   void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) override
   {
      for ( uint_t steps = 0; steps < velocityPreconditionSteps_; ++steps )
      {
         for ( uint_t k = 0; k < x.uvw().getDimension(); ++k )
            velocityBlockPreconditioner_->solve( A.getA(), x.uvw()[k], b.uvw()[k], level );
         pressureBlockPreconditioner_->apply( b.p(), x.p(), level, flag_, Replace );
      }
   }

 private:
   uint_t                                                                 velocityPreconditionSteps_;
   DoFType                                                                flag_;
   std::shared_ptr< Solver< typename OperatorType::VelocityOperator_T > > velocityBlockPreconditioner_;
   std::shared_ptr< pressureBlockPreconditionerType >                     pressureBlockPreconditioner_;
};

// MinResSolver, MinresSolver.hpp:88-250, statement by statement.  The reference rotates its ten work functions by swapping
// their data ( swap( …, level ) ); here the same rotation is done on pointers to them.
template < class OperatorType >
class MinResSolver : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType;
   MinResSolver( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel,
                 uint_t maxIter = std::numeric_limits< uint_t >::max(), double relativeTolerance = 1e-16, double absoluteTolerance = 1e-16,
                 std::shared_ptr< Solver< OperatorType > > preconditioner = std::make_shared< IdentityPreconditioner< OperatorType > >() )
   : maxIter_( maxIter )
   , iterations_( maxIter )
   , relativeTolerance_( relativeTolerance )
   , absoluteTolerance_( absoluteTolerance )
   , flag_( Inner | NeumannBoundary | FreeslipBoundary )
   , preconditioner_( std::move( preconditioner ) )
   , storage_( storage )
   {
      static const char* names[10] = { "minres_vm", "minres_v", "minres_vp", "minres_z", "minres_zp", "minres_wm", "minres_w", "minres_wp", "minres_tmp", "minres_r" };
      for ( const char* n : names )
         work_.push_back( std::make_shared< FunctionType >( n, storage, minLevel, maxLevel ) );
   }

   void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) override
   {
      ScopedTimer timer( storage_->getTimingTree(), "MinRes Solver" );
      const FunctionType *vm = work_[0].get(), *v = work_[1].get(), *vp = work_[2].get(), *z = work_[3].get(), *zp = work_[4].get(),
                         *wm = work_[5].get(), *w = work_[6].get(), *wp = work_[7].get(), *r = work_[9].get();
      for ( auto& f : work_ )
         f->interpolate( 0.0, level, All ); // setToZero

      A.apply( x, *r, level, flag_ );
      v->assign( { 1.0, -1.0 }, { b, *r }, level, flag_ );
      preconditioner_->solve( A, *z, *v, level );

      double       gamma_old = 1.0;
      double       gamma_new = std::sqrt( z->dotGlobal( *v, level, flag_ ) );
      const double res_start = gamma_new;
      double       eta = gamma_new, s_old = 0.0, s_new = 0.0, c_old = 1.0, c_new = 1.0;
      iterations_ = 0;
      if ( gamma_new < absoluteTolerance_ )
         return;

      iterations_ = maxIter_;
      for ( uint_t i = 0; i < maxIter_; ++i )
      {
         z->assign( { 1.0 / gamma_new }, { *z }, level, flag_ );
         A.apply( *z, *vp, level, flag_ );
         const double delta = vp->dotGlobal( *z, level, flag_ );

         vp->assign( { 1.0, -delta / gamma_new, -gamma_new / gamma_old }, { *vp, *v, *vm }, level, flag_ );

         zp->interpolate( 0.0, level, flag_ );
         preconditioner_->solve( A, *zp, *vp, level );

         gamma_old = gamma_new;
         gamma_new = std::sqrt( zp->dotGlobal( *vp, level, flag_ ) );

         const double alpha0 = c_new * delta - c_old * s_new * gamma_old;
         const double alpha1 = std::sqrt( alpha0 * alpha0 + gamma_new * gamma_new );
         const double alpha2 = s_new * delta + c_old * c_new * gamma_old;
         const double alpha3 = s_old * gamma_old;

         c_old = c_new;
         c_new = alpha0 / alpha1;
         s_old = s_new;
         s_new = gamma_new / alpha1;

         wp->assign( { 1.0 / alpha1, -alpha3 / alpha1, -alpha2 / alpha1 }, { *z, *wm, *w }, level, flag_ );
         x.add( { c_new * eta }, { *wp }, level, flag_ );

         eta = -s_new * eta;

         // the reference's three rotations by swap( ..., level ): (vm, v, vp) <- (v, vp, vm), (wm, w, wp) <- (w, wp, wm), (z, zp) <- (zp, z)
         {
            const FunctionType* t = vm;
            vm = v, v = vp, vp = t;
            t  = wm;
            wm = w, w = wp, wp = t;
            t  = z;
            z = zp, zp = t;
         }

         if ( std::fabs( eta ) / res_start < relativeTolerance_ || std::fabs( eta ) < absoluteTolerance_ || !std::isfinite( eta ) )
         {
            iterations_ = i;
            break;
         }
      }
   }
   uint_t getIterations() const { return iterations_; }
   void   setAbsoluteTolerance( double t ) { absoluteTolerance_ = t; }
   void   setRelativeTolerance( double t ) { relativeTolerance_ = t; }

 private:
   uint_t                                         maxIter_, iterations_;
   double                                         relativeTolerance_, absoluteTolerance_;
   DoFType                                        flag_;
   std::shared_ptr< Solver< OperatorType > >      preconditioner_;
   std::vector< std::shared_ptr< FunctionType > > work_;
   std::shared_ptr< PrimitiveStorage >            storage_;
};

} // namespace hyteg
