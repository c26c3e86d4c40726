// solvers.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// smoother wrappers, CGSolver, GeometricMultigridSolver (src/hyteg/solvers/)
#pragma once

#include "p1operator.hpp"
#include "p2operator.hpp"
#include "gridtransfer.hpp"
#include "p2gridtransfer.hpp"

namespace hyteg {

// =====================================================================================================
// Solvers ( src/hyteg/solvers/ )
// =====================================================================================================
template < class OperatorType >
class Solver
{
 public:
   using FunctionType = typename OperatorType::srcType;
   virtual ~Solver()  = default;
   virtual void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) = 0;
   // `steps` consecutive solve() calls (the pre-/post-smoothing loops of GeometricMultigridSolver.hpp:228-233,300-305);
   // a smoother may override it with something equivalent but cheaper
   virtual void solveSteps( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level, uint_t steps )
   {
      for ( uint_t i = 0; i < steps; ++i )
         solve( A, x, b, level );
   }
   // one solve() for each of several independent (x, b) pairs of the same operator (the velocity components of a block-diagonal
   // preconditioner); a solver may override it with something equivalent that shares launches between the pairs
   virtual void solveMany( const OperatorType& A, const std::vector< std::reference_wrapper< const FunctionType > >& xs,
                           const std::vector< std::reference_wrapper< const FunctionType > >& bs, uint_t level )
   {
      for ( size_t k = 0; k < xs.size(); ++k )
         solve( A, xs[k].get(), bs[k].get(), level );
   }
};

// WeightedJacobiSmoother.hpp:46-62
template < class OperatorType >
class WeightedJacobiSmoother : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType; // P1Function or P2Function
   WeightedJacobiSmoother( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel, double relax )
   : relax_( relax )
   , tmp_( "weighted_jacobi_tmp", storage, minLevel, maxLevel )
   , flag_( Inner | NeumannBoundary )
   {}
   void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) override
   {
      tmp_.assign( { 1.0 }, { x }, level, All );
      A.smooth_jac( x, b, tmp_, relax_, level, flag_ );
   }
   // n steps with ONE copy instead of n: after tmp = x (all points) a Jacobi step may just as well write into tmp
   // reading x, since a step only writes the points `flag_` selects and all other entries of the two functions agree.
   // Every step computes exactly what solve() computes (same kernel, same operands): results are bit-identical.
   void solveSteps( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level, uint_t steps ) override
   {
      if ( steps == 0 )
         return;
      tmp_.assign( { 1.0 }, { x }, level, All );
      for ( uint_t i = 0; i < steps; ++i )
      {
         if ( i % 2 == 0 )
            A.smooth_jac( x, b, tmp_, relax_, level, flag_ );
         else
            A.smooth_jac( tmp_, b, x, relax_, level, flag_ );
      }
      if ( steps % 2 == 0 )
         x.assign( { 1.0 }, { tmp_ }, level, flag_ );
   }

 private:
   double       relax_;
   FunctionType tmp_;
   DoFType      flag_;
};

// Mixed-precision weighted Jacobi: the "fp32 smoother" of BASELINE config 5.  The reference instantiates its generated apply
// kernels for float (apply_3D_macrocell_vertexdof_to_vertexdof_replace.cpp:96-97) and converts between function
// precisions with copyFrom (VertexDoFFunction.hpp:598-650); this smoother uses those pieces in defect-correction form, so
// that the iterate and the right-hand side stay in double.  n Jacobi sweeps on A x = b are, in exact arithmetic,
//      r = b - A x;   e_1 = relax r / c;   e_{k+1} = e_k + relax ( r - A e_k ) / c;   x += e_n          (c: centre weight)
// -- the sweeps on the error equation run in float (half the bytes), the first one fused with the residual and the last one
// with the update of x (hyteg_hip_p1_residual_jacobi_start_f32 / hyteg_hip_p1_jacobi_accumulate_f32): n launches for n sweeps.
//  * solveSteps( n ) on macro-cells whose shell is not selected by the flag (one macro-cell with fixed boundary values): exactly
//    that -- n Jacobi sweeps, corrections rounded to float relative to the CORRECTION, not to the iterate (round 2 needed
//    residual + conversion + sweeps + axpy + copy + a double sweep per step: 3-4 x slower than the double smoother);
//  * otherwise (points shared between macro-cells are smoothed): per step fp32Sweeps float sweeps on the cell interiors with
//    e = 0 on the cell boundary (block Jacobi over the macro-cells), then one double smooth_jac over all selected points.
template < class OperatorType >
class MixedPrecisionJacobiSmoother : public Solver< OperatorType >
{
 public:
   MixedPrecisionJacobiSmoother( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel, double relax, uint_t fp32Sweeps = 2 )
   : storage_( storage )
   , relax_( relax )
   , fp32Sweeps_( fp32Sweeps )
   , tmp_( "mixed_jacobi_tmp", storage, minLevel, maxLevel )
   , flag_( Inner | NeumannBoundary )
   {}
   ~MixedPrecisionJacobiSmoother() override
   {
      for ( auto& kv : buffers_ )
         for ( float* q : kv.second )
            hyteg_hip_free( q );
   }
   void solve( const OperatorType& A, const P1Function< double >& x, const P1Function< double >& b, uint_t level ) override
   {
      if ( floatLevel( level ) && fp32Sweeps_ >= 2 )
         floatSweeps( A, x, b, level, fp32Sweeps_ );
      tmp_.assign( { 1.0 }, { x }, level, All );
      A.smooth_jac( x, b, tmp_, relax_, level, flag_ );
   }
   void solveSteps( const OperatorType& A, const P1Function< double >& x, const P1Function< double >& b, uint_t level, uint_t steps ) override
   {
      bool anyShell = storage_->numRanks() > 1;
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         anyShell = anyShell || ( storage_->maskFor( storage_->getLocalCell( c ), x.effectiveFlag( flag_ ) ) & HYTEG_HIP_MASK_SHELL );
      if ( anyShell || !floatLevel( level ) || steps < 2 )
      {
         for ( uint_t i = 0; i < steps; ++i )
            solve( A, x, b, level );
         return;
      }
      floatSweeps( A, x, b, level, steps ); // `steps` Jacobi sweeps: every selected point is an inner point of its macro-cell
   }

 private:
   static bool floatLevel( uint_t level ) { return level >= HYTEG_HIP_MIN_LEVEL && level <= 10; }
   // x += e_n on the inner points of every local cell, e_n = n float Jacobi sweeps on A e = b - A x from e = 0
   void floatSweeps( const OperatorType& A, const P1Function< double >& x, const P1Function< double >& b, uint_t level, uint_t n )
   {
      const DoFType flag = x.effectiveFlag( flag_ );
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
      {
         const MacroCell& cell = storage_->getLocalCell( c );
         if ( !( storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_INNER ) )
            continue;
         float**     f  = buffersFor( c, level ); // r, e0, e1: boundary entries of e0 / e1 are zero since their allocation
         const auto& st = A.getCellStencils( cell.id, level );
         auto        s  = storage_->stream();
         hipCheck( hyteg_hip_p1_residual_jacobi_start_f32( f[0], f[1], b.getCellPointer( c, level ), x.getCellPointer( c, level ), (int) level, st.inner,
                                                           relax_, s ),
                   "mixed Jacobi: residual + first sweep" );
         float *src = f[1], *dst = f[2];
         for ( uint_t k = 2; k < n; ++k )
         {
            hipCheck( hyteg_hip_p1_jacobi_cell_f32( dst, f[0], src, nullptr, (int) level, st.inner, relax_, s ), "mixed Jacobi: float sweep" );
            std::swap( src, dst );
         }
         hipCheck( hyteg_hip_p1_jacobi_accumulate_f32( x.getCellPointer( c, level ), f[0], src, (int) level, st.inner, relax_, s ),
                   "mixed Jacobi: last sweep + correction" );
      }
   }
   float** buffersFor( uint_t c, uint_t level )
   {
      auto key = std::make_pair( c, level );
      auto it  = buffers_.find( key );
      if ( it == buffers_.end() )
      {
         const size_t          bytes = (size_t) layout::cellSize( (int) level ) * sizeof( float );
         std::vector< float* > v;
         for ( int k = 0; k < 3; ++k )
         {
            void* q = nullptr;
            hipCheck( hyteg_hip_malloc( &q, bytes ), "mixed Jacobi: malloc" );
            hipCheck( hyteg_hip_memset_zero( q, bytes, storage_->stream() ), "mixed Jacobi: zero" );
            v.push_back( static_cast< float* >( q ) );
         }
         it = buffers_.emplace( key, v ).first;
      }
      return it->second.data();
   }
   std::shared_ptr< PrimitiveStorage >                            storage_;
   double                                                         relax_;
   uint_t                                                         fp32Sweeps_;
   P1Function< double >                                           tmp_;
   DoFType                                                        flag_;
   std::map< std::pair< uint_t, uint_t >, std::vector< float* > > buffers_;
};

// GaussSeidelSmoother.hpp:38-50, SORSmoother.hpp:33-46
template < class OperatorType >
class SORSmoother : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType; // P1Function or P2Function
   explicit SORSmoother( double relax )
   : relax_( relax )
   , flag_( Inner | NeumannBoundary )
   {}
   void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) override
   {
      A.smooth_sor( x, b, relax_, level, flag_ );
   }
   void solveMany( const OperatorType& A, const std::vector< std::reference_wrapper< const FunctionType > >& xs,
                   const std::vector< std::reference_wrapper< const FunctionType > >& bs, uint_t level ) override
   {
      if constexpr ( std::is_same< FunctionType, P1Function< double > >::value )
         A.smooth_sor_many( xs, bs, relax_, level, flag_ );
      else
         Solver< OperatorType >::solveMany( A, xs, bs, level );
   }
   // the `steps` sweeps of a pre- / post-smoothing phase at once: pipelined where the operator can (smooth_sor_steps)
   void solveSteps( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level, uint_t steps ) override
   {
      if constexpr ( std::is_same< FunctionType, P1Function< double > >::value )
         A.smooth_sor_steps( x, b, relax_, level, flag_, steps );
      else
         Solver< OperatorType >::solveSteps( A, x, b, level, steps );
   }

 private:
   double  relax_;
   DoFType flag_;
};
template < class OperatorType >
class GaussSeidelSmoother : public SORSmoother< OperatorType >
{
 public:
   GaussSeidelSmoother()
   : SORSmoother< OperatorType >( 1.0 )
   {}
};

// CGSolver.hpp:88-205 (no preconditioner: IdentityPreconditioner)
template < class OperatorType >
class CGSolver : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType;
   CGSolver( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel, uint_t maxIter = 1000,
             double relativeTolerance = 1e-16, double absoluteTolerance = 1e-16 )
   : p_( "p", storage, minLevel, maxLevel )
   , z_( "z", storage, minLevel, maxLevel )
   , ap_( "ap", storage, minLevel, maxLevel )
   , r_( "r", storage, minLevel, maxLevel )
   , flag_( Inner | NeumannBoundary )
   , maxIter_( maxIter )
   , relTol_( relativeTolerance )
   , absTol_( absoluteTolerance )
   {}
   ~CGSolver() override
   {
      if ( scalars_ )
         hyteg_hip_free( scalars_ );
   }
   // Device-resident scalars (no counterpart in the reference, whose loop reads every dot product on the host): on
   // coarse levels a CG iteration is ~10 launches of a few microseconds, and three host round trips per iteration
   // cost more than the launches.  alpha, beta and the convergence test are computed by a one-thread kernel
   // (hyteg_hip_cg_scalars), the vector updates read them from device memory, and the host looks at the convergence
   // flag every 4 iterations; iterations enqueued after convergence are exact no-ops (alpha = 0).  Same recurrences,
   // same arithmetic as the loop below.  Used for storages of one rank, up to level 5; off: HYTEG_AMD_DEVICE_CG=0.
   void setUseDeviceScalars( bool on ) { useDeviceScalars_ = on; }

   void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) override
   {
      iterationsOnDevice_ = false;
      if ( deviceScalarsUsable( x, level ) )
      {
         solveWithDeviceScalars( A, x, b, level );
         return;
      }
      p_.setToZero( level );
      z_.setToZero( level );
      ap_.setToZero( level );
      r_.setToZero( level );
      // init(): r = b - A x ; z = r ; p = z ; prsold = <r,z>
      A.apply( x, p_, level, flag_, Replace );
      r_.assign( { 1.0, -1.0 }, { b, p_ }, level, flag_ );
      z_.assign( { 1.0 }, { r_ }, level, flag_ );
      p_.assign( { 1.0 }, { z_ }, level, flag_ );
      double       prsold    = r_.dotGlobal( z_, level, flag_ );
      const double res_start = std::sqrt( r_.dotGlobal( r_, level, flag_ ) );
      iterations_            = 0;
      if ( res_start < absTol_ )
         return;
      for ( uint_t i = 0; i < maxIter_; ++i )
      {
         A.apply( p_, ap_, level, flag_, Replace );
         const double pAp   = p_.dotGlobal( ap_, level, flag_ );
         const double alpha = prsold / pAp;
         x.add( { alpha }, { p_ }, level, flag_ );
         r_.add( { -alpha }, { ap_ }, level, flag_ );
         const double rsnew   = r_.dotGlobal( r_, level, flag_ );
         const double sqrsnew = std::sqrt( rsnew );
         iterations_          = i + 1;
         if ( sqrsnew / res_start < relTol_ || sqrsnew < absTol_ )
            break;
         // identity preconditioner: z = r, so <r,z> is the <r,r> just computed (the reference copies and reduces again)
         const double prsnew = rsnew;
         const double beta   = prsnew / prsold;
         p_.assign( { 1.0, beta }, { r_, p_ }, level, flag_ );
         prsold = prsnew;
      }
   }
   uint_t getIterations() const
   {
      if ( iterationsOnDevice_ )
      {
         // the one-launch solve leaves its iteration count on the device; fetched (one synchronisation) only when asked for
         double h = 0.0;
         hipCheck( hyteg_hip_download( &h, scalars_, sizeof( double ), iterationsStream_ ), "CGSolver: iterations" );
         iterations_         = (uint_t) h;
         iterationsOnDevice_ = false;
      }
      return iterations_;
   }
   // levels whose cell arrays together fit one workgroup are solved by ONE launch (hyteg_hip_p1_cg_small_cells); off: false
   void setUseSingleLaunch( bool on ) { useSingleLaunch_ = on; }

 private:
   template < typename F >
   bool deviceScalarsUsable( const F&, uint_t ) const
   {
      return false; // P2 functions: host scalars
   }
   bool deviceScalarsUsable( const P1Function< double >& x, uint_t level ) const
   {
      static const bool envOn = [] {
         const char* e = std::getenv( "HYTEG_AMD_DEVICE_CG" );
         return !( e && e[0] == '0' );
      }();
      const auto& st = *x.getStorage();
      return envOn && useDeviceScalars_ && st.numRanks() == 1 && level <= 5 && st.getNumberOfLocalCells() >= 1 &&
             st.getNumberOfLocalCells() <= HYTEG_HIP_MAX_BATCH;
   }
   template < typename F >
   void solveWithDeviceScalars( const OperatorType&, const F&, const F&, uint_t )
   {}
   void solveWithDeviceScalars( const OperatorType& A, const P1Function< double >& x, const P1Function< double >& b, uint_t level )
   {
      const auto& st = *x.getStorage();
      if ( !scalars_ )
      {
         void* d = nullptr;
         hipCheck( hyteg_hip_malloc( &d, HYTEG_HIP_CG_SLOTS * sizeof( double ) ), "CGSolver: scalars" );
         scalars_ = static_cast< double* >( d );
      }
      double* const S = scalars_;
      if ( useSingleLaunch_ && A.canCgSolveSmall( level ) )
      {
         A.cgSolveSmall( x, b, level, flag_, maxIter_, relTol_, absTol_, S );
         iterationsOnDevice_ = true;
         iterationsStream_   = st.stream(); // the download has to be ordered after the solve on ITS stream
         return;
      }
      p_.setToZero( level ); // apply( p ) reads p on every point; assign below writes only the points flag_ selects
      A.apply( x, p_, level, flag_, Replace );
      r_.assign( { 1.0, -1.0 }, { b, p_ }, level, flag_ );
      p_.assign( { 1.0 }, { r_ }, level, flag_ );
      hipCheck( hyteg_hip_memset_zero( S, HYTEG_HIP_CG_SLOTS * sizeof( double ), st.stream() ), "CGSolver: scalars reset" );
      r_.dotLocalToCgScalars( r_, level, flag_, S, HYTEG_HIP_CG_RR, 0, relTol_, absTol_ );
      iterations_ = 0;
      for ( uint_t i = 0; i < maxIter_; ++i )
      {
         A.apply( p_, ap_, level, flag_, Replace );
         p_.dotLocalToCgScalars( ap_, level, flag_, S, HYTEG_HIP_CG_PAP, 1, relTol_, absTol_ );
         x.vectorOpDeviceScalars( 1, { S + HYTEG_HIP_CG_ALPHA }, { p_ }, level, flag_ );
         r_.vectorOpDeviceScalars( 1, { S + HYTEG_HIP_CG_NEG_ALPHA }, { ap_ }, level, flag_ );
         r_.dotLocalToCgScalars( r_, level, flag_, S, HYTEG_HIP_CG_RR, 2, relTol_, absTol_ );
         p_.vectorOpDeviceScalars( 0, { S + HYTEG_HIP_CG_ONE, S + HYTEG_HIP_CG_BETA }, { r_, p_ }, level, flag_ );
         if ( ( i + 1 ) % 4 == 0 || i + 1 == maxIter_ )
         {
            double h[2];
            hipCheck( hyteg_hip_download( h, S + HYTEG_HIP_CG_DONE, 2 * sizeof( double ), st.stream() ), "CGSolver: convergence flag" );
            iterations_ = (uint_t) h[1];
            if ( h[0] != 0.0 )
               break;
         }
      }
   }

   bool                 useDeviceScalars_ = true, useSingleLaunch_ = true;
   mutable bool         iterationsOnDevice_ = false;
   hyteg_hip_stream_t   iterationsStream_   = nullptr;
   double*              scalars_          = nullptr;
   FunctionType         p_, z_, ap_, r_;
   DoFType              flag_;
   uint_t               maxIter_;
   double               relTol_, absTol_;
   mutable uint_t       iterations_ = 0;
};

// GeometricMultigridSolver.hpp:40-330
// Generic over the operator's function type and the grid-transfer operators, like the reference (which takes
// RestrictionOperator< FunctionType > / ProlongationOperator< FunctionType >): P1 functions with the linear transfer by
// default, P1StokesFunction with P1P1StokesToP1P1Stokes{Restriction,Prolongation} (stokes.hpp).
// does the operator offer residual( x, b, r, level, flag )?
template < class OperatorType, class FunctionType, class = void >
struct HasResidual : std::false_type
{};
template < class OperatorType, class FunctionType >
struct HasResidual< OperatorType, FunctionType,
                    std::void_t< decltype( std::declval< const OperatorType& >().residual( std::declval< const FunctionType& >(), std::declval< const FunctionType& >(),
                                                                                           std::declval< const FunctionType& >(), uint_t( 0 ), All ) ) > >
: std::true_type
{};

template < class OperatorType, class RestrictionType = P1toP1LinearRestriction, class ProlongationType = P1toP1LinearProlongation >
class GeometricMultigridSolver : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType;
   GeometricMultigridSolver( const std::shared_ptr< PrimitiveStorage >&         storage,
                             std::shared_ptr< Solver< OperatorType > >          smoother,
                             std::shared_ptr< Solver< OperatorType > >          coarseSolver,
                             std::shared_ptr< RestrictionType >                 restrictionOperator,
                             std::shared_ptr< ProlongationType >                prolongationOperator,
                             uint_t                                             minLevel,
                             uint_t                                             maxLevel,
                             uint_t                                             preSmoothSteps  = 3,
                             uint_t                                             postSmoothSteps = 3,
                             uint_t                                             smoothIncrement = 0,
                             CycleType                                          cycleType       = CycleType::VCYCLE )
   : minLevel_( minLevel )
   , maxLevel_( maxLevel )
   , preSmoothSteps_( preSmoothSteps )
   , postSmoothSteps_( postSmoothSteps )
   , smoothIncrement_( smoothIncrement )
   , flag_( Inner | NeumannBoundary )
   , cycleType_( cycleType )
   , smoother_( smoother )
   , coarseSolver_( coarseSolver )
   , restrictionOperator_( restrictionOperator )
   , prolongationOperator_( prolongationOperator )
   , tmp_( "gmg_tmp", storage, minLevel, maxLevel )
   , storage_( storage )
   {}

   ~GeometricMultigridSolver() override
   {
      for ( auto& kv : recordings_ )
         kv.second.destroy();
      if ( captureStream_ )
         hyteg_hip_stream_destroy( captureStream_ );
   }

   // Launch graphs (no counterpart in the reference, whose cycle is host loops): the launches of a cycle -- ~20 per
   // level, most of them a few microseconds on the coarse levels -- are recorded once per (operator, x, b, level) and
   // replayed as one graph launch per segment between coarse-grid solves (the coarse solver reads dot products on the
   // host and stays outside).  The first cycle with given arguments runs with ordinary launches (it creates every lazily
   // built table and scratch array), the second records (nothing executes while recording) and replays, later cycles
   // replay.  Same kernels, same order, same arguments: results are identical to ordinary launches.
   // Opt-in: setUseGraphs( true ) or HYTEG_AMD_GRAPHS=1.  Measured on MI355X the replay saves only 1-8 % of a cycle (the
   // cycle is bound by the ~3 us a dependent small kernel takes on the GPU, not by the host's launch rate), while
   // recording and instantiating costs a few milliseconds once -- it pays for solves of many cycles only.  Never
   // used for storages distributed over several ranks (the exchange hooks are host callbacks).
   void setUseGraphs( bool on ) { useGraphs_ = on; }
   std::shared_ptr< Solver< OperatorType > > getCoarseSolver() const { return coarseSolver_; }
   bool usesGraphs() const { return graphsUsable(); }
   // number of cycles that were replayed from a recording (tests)
   uint_t replayedCycles() const { return replayed_; }

   void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) override
   {
      ScopedTimer timerGmg( storage_->getTimingTree(), "Geometric Multigrid Solver" );
      invokedLevel_ = level;
      if ( !graphsUsable() )
      {
         solveRecursively( A, x, b, level );
         return;
      }
      const Key key{ A.uid(), x.uid(), b.uid(), level };
      auto      it = recordings_.find( key );
      if ( it == recordings_.end() )
      {
         if ( recordings_.size() >= 8 )
         {
            for ( auto& kv : recordings_ )
               kv.second.destroy();
            recordings_.clear();
         }
         recordings_[key] = Recording{};
         solveRecursively( A, x, b, level );
         return;
      }
      Recording& rec = it->second;
      if ( !rec.recorded && !rec.failed )
         record( rec, A, x, b, level );
      if ( !rec.recorded )
      {
         solveRecursively( A, x, b, level );
         return;
      }
      for ( size_t k = 0; k < rec.segments.size(); ++k )
      {
         hipCheck( hyteg_hip_graph_launch( rec.segments[k], storage_->stream() ), "GeometricMultigridSolver: graph launch" );
         if ( k + 1 < rec.segments.size() )
            coarseSolver_->solve( A, x, b, minLevel_ );
      }
      ++replayed_;
   }

 private:
   using Key = std::tuple< uint64_t, uint64_t, uint64_t, uint_t >;
   struct Recording
   {
      std::vector< hyteg_hip_graph_t > segments; // separated by coarse-grid solves
      bool                             recorded = false, failed = false;
      int                              attempts = 0;
      void                             destroy()
      {
         for ( auto g : segments )
            hyteg_hip_graph_destroy( g );
         segments.clear();
      }
   };

   bool graphsUsable() const
   {
      static const bool envOn = [] {
         const char* e = std::getenv( "HYTEG_AMD_GRAPHS" );
         return e && e[0] == '1';
      }();
      // recorded cycles exist for P1 functions only: the Stokes composition uploads tables and synchronises inside its smoother
      // (not legal while a stream is being recorded), the P2 cycle has not been tried
      return ( useGraphs_ || envOn ) && storage_->numRanks() == 1 && std::is_same< FunctionType, P1Function< double > >::value;
   }

   void endSegment( Recording& rec )
   {
      hyteg_hip_graph_t g = nullptr;
      capturing_          = false;
      hipCheck( hyteg_hip_graph_end_capture( captureStream_, &g ), "GeometricMultigridSolver: end capture" );
      rec.segments.push_back( g );
   }
   void beginSegment()
   {
      hipCheck( hyteg_hip_graph_begin_capture( captureStream_ ), "GeometricMultigridSolver: begin capture" );
      capturing_ = true;
   }

   void record( Recording& rec, const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level )
   {
      const hyteg_hip_stream_t user = storage_->stream();
      try
      {
         if ( !captureStream_ )
            hipCheck( hyteg_hip_stream_create( &captureStream_ ), "GeometricMultigridSolver: stream" );
         storage_->setStream( captureStream_ );
         recording_ = &rec;
         beginSegment();
         solveRecursively( A, x, b, level );
         endSegment( rec );
         rec.recorded = true;
      } catch ( const std::exception& e )
      {
         // something in the cycle cannot be recorded: nothing has executed, fall back to ordinary launches for good
         if ( rec.attempts >= 2 )
            std::fprintf( stderr, "hyteg_amd: multigrid cycle not recordable (%s); using ordinary launches\n", e.what() );
         if ( capturing_ )
            hyteg_hip_graph_abort_capture( captureStream_ );
         capturing_ = false;
         rec.destroy();
         rec.failed = ++rec.attempts >= 3; // a table built lazily in this very cycle: the next cycle tries again
      }
      recording_ = nullptr;
      storage_->setStream( user );
   }

   // timer names and nesting of GeometricMultigridSolver.hpp:200-300
   void solveRecursively( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level )
   {
      TimingTree*       tt        = recording_ ? nullptr : storage_->getTimingTree();
      const std::string levelName = "Level " + std::to_string( level );
      if ( level == minLevel_ )
      {
         if ( recording_ )
         {
            endSegment( *recording_ );
            beginSegment();
         }
         else
         {
            ScopedTimer tl( tt, levelName ), tc( tt, "Coarse Grid Solver" );
            coarseSolver_->solve( A, x, b, minLevel_ );
         }
         return;
      }
      const uint_t pre = preSmoothSteps_ + smoothIncrement_ * ( invokedLevel_ - level );
      {
         ScopedTimer tl( tt, levelName ), ts( tt, "Smoother" );
         smoother_->solveSteps( A, x, b, level, pre );
      }
      {
         ScopedTimer tl( tt, levelName ), tr( tt, "Residual" );
         if constexpr ( HasResidual< OperatorType, FunctionType >::value )
            A.residual( x, b, tmp_, level, flag_ ); // apply + assign, in one launch where the operator can
         else
         {
            A.apply( x, tmp_, level, flag_ );
            tmp_.assign( { 1.0, -1.0 }, { b, tmp_ }, level, flag_ );
         }
      }
      {
         ScopedTimer tl( tt, levelName ), tr( tt, "Restriction" );
         if constexpr ( std::is_same< RestrictionType, P1toP1LinearRestriction >::value )
            restrictionOperator_->restrictInto( tmp_, b, level, flag_ ); // restrict + assign without the copy
         else
         {
            restrictionOperator_->restrict( tmp_, level, flag_ );
            b.assign( { 1.0 }, { tmp_ }, level - 1, flag_ );
         }
         x.interpolate( 0.0, level - 1 );
      }
      solveRecursively( A, x, b, level - 1 );
      if ( cycleType_ == CycleType::WCYCLE )
         solveRecursively( A, x, b, level - 1 );
      {
         ScopedTimer tl( tt, levelName ), tp( tt, "Prolongation" );
         prolongationOperator_->prolongateAndAdd( x, level - 1, flag_ );
      }
      const uint_t post = postSmoothSteps_ + smoothIncrement_ * ( invokedLevel_ - level );
      {
         ScopedTimer tl( tt, levelName ), ts( tt, "Smoother" );
         smoother_->solveSteps( A, x, b, level, post );
      }
   }

   uint_t                                       minLevel_, maxLevel_, preSmoothSteps_, postSmoothSteps_, smoothIncrement_;
   uint_t                                       invokedLevel_ = 0;
   DoFType                                      flag_;
   CycleType                                    cycleType_;
   std::shared_ptr< Solver< OperatorType > >    smoother_, coarseSolver_;
   std::shared_ptr< RestrictionType >           restrictionOperator_;
   std::shared_ptr< ProlongationType >          prolongationOperator_;
   FunctionType                                 tmp_;
   std::shared_ptr< PrimitiveStorage >          storage_;
   bool                                         useGraphs_ = false, capturing_ = false;
   hyteg_hip_stream_t                           captureStream_ = nullptr;
   Recording*                                   recording_     = nullptr;
   std::map< Key, Recording >                   recordings_;
   uint_t                                       replayed_ = 0;
};

} // namespace hyteg
