// stokes.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// The stabilised P1-P1 Stokes operator as a composition of P1 constant-stencil operators, its Uzawa smoother and
// grid transfer: src/mixed_operator/P1P1StokesOperator.hpp, VectorLaplaceOperator.hpp, VectorToScalarOperator.hpp,
// ScalarToVectorOperator.hpp, src/hyteg/composites/P1StokesFunction.hpp, src/hyteg/solvers/UzawaSmoother.hpp,
// src/hyteg/solvers/preconditioners/stokes/StokesVelocityBlockBlockDiagonalPreconditioner.hpp,
// src/hyteg/gridtransferoperators/P1P1StokesToP1P1Stokes{Restriction,Prolongation}.hpp.
// Every block is a P1ConstantOperator< Form > (p1operator.hpp) with another form: the device work is the same 15-point
// stencil kernels as for the Laplace operator, with other weights.
#pragma once

#include "minres.hpp"
#include "solvers.hpp"

namespace hyteg {

// P1ConstantOperator.hpp:178-210
using P1DivxOperator  = P1ConstantOperator< forms::P1DivForm< 0 > >;
using P1DivyOperator  = P1ConstantOperator< forms::P1DivForm< 1 > >;
using P1DivzOperator  = P1ConstantOperator< forms::P1DivForm< 2 > >;
using P1DivTxOperator = P1ConstantOperator< forms::P1DivTForm< 0 > >;
using P1DivTyOperator = P1ConstantOperator< forms::P1DivTForm< 1 > >;
using P1DivTzOperator = P1ConstantOperator< forms::P1DivTForm< 2 > >;
using P1PSPGOperator  = P1ConstantOperator< forms::P1PSPGForm >;

// =====================================================================================================
// P1VectorFunction (src/hyteg/p1functionspace/P1VectorFunction.hpp), three components in 3D
// =====================================================================================================
template < typename ValueType >
class P1VectorFunction
{
 public:
   P1VectorFunction( const std::string& name, const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   {
      static const char* suffix[3] = { "_u", "_v", "_w" };
      for ( int k = 0; k < 3; ++k )
         comp_.push_back( std::make_shared< P1Function< ValueType > >( name + suffix[k], storage, minLevel, maxLevel ) );
   }
   uint_t                         getDimension() const { return 3; }
   const P1Function< ValueType >& operator[]( uint_t k ) const { return *comp_.at( k ); }
   const P1Function< ValueType >& component( uint_t k ) const { return *comp_.at( k ); }

   void interpolate( ValueType constant, uint_t level, DoFType flag = All ) const
   {
      for ( auto& c : comp_ )
         c->interpolate( constant, level, flag );
   }
   void interpolate( const std::vector< std::function< ValueType( const Point3D& ) > >& expr, uint_t level, DoFType flag = All ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         comp_[k]->interpolate( expr.at( k ), level, flag );
   }
   void assign( const std::vector< ValueType >&                                                        scalars,
                const std::vector< std::reference_wrapper< const P1VectorFunction< ValueType > > >& functions,
                uint_t                                                                               level,
                DoFType                                                                              flag = All ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         comp_[k]->assign( scalars, componentRefs( functions, k ), level, flag );
   }
   void add( const std::vector< ValueType >&                                                        scalars,
             const std::vector< std::reference_wrapper< const P1VectorFunction< ValueType > > >& functions,
             uint_t                                                                               level,
             DoFType                                                                              flag = All ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         comp_[k]->add( scalars, componentRefs( functions, k ), level, flag );
   }
   ValueType dotGlobal( const P1VectorFunction< ValueType >& rhs, uint_t level, DoFType flag = All ) const
   {
      ValueType s = 0;
      for ( uint_t k = 0; k < 3; ++k )
         s += comp_[k]->dotGlobal( rhs[k], level, flag );
      return s;
   }

 private:
   static std::vector< std::reference_wrapper< const P1Function< ValueType > > >
       componentRefs( const std::vector< std::reference_wrapper< const P1VectorFunction< ValueType > > >& functions, uint_t k )
   {
      std::vector< std::reference_wrapper< const P1Function< ValueType > > > r;
      for ( const auto& f : functions )
         r.push_back( std::cref( f.get()[k] ) );
      return r;
   }
   std::vector< std::shared_ptr< P1Function< ValueType > > > comp_;
};

// =====================================================================================================
// P1StokesFunction (composites/P1StokesFunction.hpp): velocity with the storage's boundary types (create0123BC),
// pressure with createAllInnerBC
// =====================================================================================================
template < typename ValueType >
class P1StokesFunction
{
 public:
   using valueType = ValueType;
   P1StokesFunction( const std::string& name, const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : uvw_( name + "_uvw", storage, minLevel, maxLevel )
   , p_( name + "_p", storage, minLevel, maxLevel )
   {
      p_.setBoundaryConditionAllInner();
   }
   const P1VectorFunction< ValueType >& uvw() const { return uvw_; }
   const P1Function< ValueType >&       p() const { return p_; }
   uint64_t                             uid() const { return p_.uid(); }

   void interpolate( ValueType constant, uint_t level, DoFType flag = All ) const
   {
      uvw_.interpolate( constant, level, flag );
      p_.interpolate( constant, level, flag );
   }
   void assign( const std::vector< ValueType >&                                                        scalars,
                const std::vector< std::reference_wrapper< const P1StokesFunction< ValueType > > >& functions,
                uint_t                                                                               level,
                DoFType                                                                              flag = All ) const
   {
      std::vector< std::reference_wrapper< const P1VectorFunction< ValueType > > > v;
      std::vector< std::reference_wrapper< const P1Function< ValueType > > >       q;
      for ( const auto& f : functions )
      {
         v.push_back( std::cref( f.get().uvw() ) );
         q.push_back( std::cref( f.get().p() ) );
      }
      uvw_.assign( scalars, v, level, flag );
      p_.assign( scalars, q, level, flag );
   }
   void add( const std::vector< ValueType >&                                                        scalars,
             const std::vector< std::reference_wrapper< const P1StokesFunction< ValueType > > >& functions,
             uint_t                                                                               level,
             DoFType                                                                              flag = All ) const
   {
      std::vector< std::reference_wrapper< const P1VectorFunction< ValueType > > > v;
      std::vector< std::reference_wrapper< const P1Function< ValueType > > >       q;
      for ( const auto& f : functions )
      {
         v.push_back( std::cref( f.get().uvw() ) );
         q.push_back( std::cref( f.get().p() ) );
      }
      uvw_.add( scalars, v, level, flag );
      p_.add( scalars, q, level, flag );
   }
   ValueType dotGlobal( const P1StokesFunction< ValueType >& rhs, uint_t level, DoFType flag = All ) const
   {
      return uvw_.dotGlobal( rhs.uvw(), level, flag ) + p_.dotGlobal( rhs.p(), level, flag );
   }

 private:
   P1VectorFunction< ValueType > uvw_;
   P1Function< ValueType >       p_;
};

// vertexdof::projectMean (VertexDoFFunction.hpp:586-592): subtract the mean over ALL DoFs (every shared DoF counted once)
inline void projectMean( const P1Function< double >& pressure, uint_t level )
{
   auto                 storage = pressure.getStorage();
   P1Function< double > one( "projectMean_one", storage, level, level );
   one.interpolate( 1.0, level, All );
   const double count = one.dotGlobal( one, level, All );
   const double sum   = pressure.dotGlobal( one, level, All );
   pressure.add( { -sum / count }, { one }, level, All );
}

// =====================================================================================================
// Block operators
// =====================================================================================================
// VectorLaplaceOperator.hpp:122  P1ConstantVectorLaplaceOperator: block diagonal of one scalar Laplace operator
class P1ConstantVectorLaplaceOperator
{
 public:
   using srcType = P1VectorFunction< double >;
   using dstType = P1VectorFunction< double >;
   P1ConstantVectorLaplaceOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : lapl_( std::make_shared< P1ConstantLaplaceOperator >( storage, minLevel, maxLevel ) )
   {
      lapl_->computeInverseDiagonalOperatorValues(); // Jacobi-type velocity smoothers ask the scalar operator for it
   }
   void apply( const srcType& src, const dstType& dst, uint_t level, DoFType flag, UpdateType updateType = Replace ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         lapl_->apply( src[k], dst[k], level, flag, updateType );
   }
   const P1ConstantLaplaceOperator&                   getSubOperator( uint_t, uint_t ) const { return *lapl_; }
   std::shared_ptr< const P1ConstantLaplaceOperator > scalar() const { return lapl_; }

 private:
   std::shared_ptr< P1ConstantLaplaceOperator > lapl_;
};

// VectorToScalarOperator.hpp: P1ConstantDivOperator = ( Divx, Divy, Divz ): first component with the caller's update type,
// the others added (:63-71)
class P1ConstantDivOperator
{
 public:
   P1ConstantDivOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : x_( storage, minLevel, maxLevel )
   , y_( storage, minLevel, maxLevel )
   , z_( storage, minLevel, maxLevel )
   {}
   void apply( const P1VectorFunction< double >& src, const P1Function< double >& dst, uint_t level, DoFType flag, UpdateType updateType = Replace ) const
   {
      x_.apply( src[0], dst, level, flag, updateType );
      y_.apply( src[1], dst, level, flag, Add );
      z_.apply( src[2], dst, level, flag, Add );
   }
   const P1DivxOperator& x() const { return x_; }
   const P1DivyOperator& y() const { return y_; }
   const P1DivzOperator& z() const { return z_; }

 private:
   P1DivxOperator x_;
   P1DivyOperator y_;
   P1DivzOperator z_;
};

// ScalarToVectorOperator.hpp: P1ConstantDivTOperator = ( DivTx, DivTy, DivTz )^T
class P1ConstantDivTOperator
{
 public:
   P1ConstantDivTOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : x_( storage, minLevel, maxLevel )
   , y_( storage, minLevel, maxLevel )
   , z_( storage, minLevel, maxLevel )
   {}
   void apply( const P1Function< double >& src, const P1VectorFunction< double >& dst, uint_t level, DoFType flag, UpdateType updateType = Replace ) const
   {
      x_.apply( src, dst[0], level, flag, updateType );
      y_.apply( src, dst[1], level, flag, updateType );
      z_.apply( src, dst[2], level, flag, updateType );
   }
   const P1DivTxOperator& x() const { return x_; }
   const P1DivTyOperator& y() const { return y_; }
   const P1DivTzOperator& z() const { return z_; }

 private:
   P1DivTxOperator x_;
   P1DivTyOperator y_;
   P1DivTzOperator z_;
};

// P1PSPGInvDiagOperator (P1ConstantOperator.hpp:206-210: Diagonal, InvertDiagonal): the stencil keeps only the centre
// weight, inverted (P1Operator.hpp:2149-2158; on shared points the inverse of the SUM over the neighbour cells, :2110-2117)
// = pointwise multiplication with the inverse diagonal of the PSPG operator
class P1PSPGInvDiagOperator
{
 public:
   P1PSPGInvDiagOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : pspg_( storage, minLevel, maxLevel )
   , tmp_( "pspg_inv_diag_tmp", storage, minLevel, maxLevel )
   {
      pspg_.computeInverseDiagonalOperatorValues();
      tmp_.setBoundaryConditionAllInner();
   }
   void apply( const P1Function< double >& src, const P1Function< double >& dst, uint_t level, DoFType flag, UpdateType updateType = Replace ) const
   {
      const auto& invDiag = *pspg_.getInverseDiagonalValues();
      if ( updateType == Replace )
      {
         dst.multElementwise( { invDiag, src }, level, flag );
         return;
      }
      tmp_.multElementwise( { invDiag, src }, level, flag );
      dst.add( { 1.0 }, { tmp_ }, level, flag );
   }

 private:
   P1PSPGOperator       pspg_;
   P1Function< double > tmp_;
};

// P1P1StokesOperator.hpp:32-99
class P1P1StokesOperator
{
 public:
   using srcType            = P1StokesFunction< double >;
   using dstType            = P1StokesFunction< double >;
   using VelocityOperator_T = P1ConstantLaplaceOperator;
   static constexpr bool hasPspgBlock = true; // has_pspg_block< P1P1StokesOperator >, P1P1StokesOperator.hpp:95

   P1P1StokesOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : lapl( storage, minLevel, maxLevel )
   , div( storage, minLevel, maxLevel )
   , divT( storage, minLevel, maxLevel )
   , pspg( storage, minLevel, maxLevel )
   , pspg_inv_diag_( storage, minLevel, maxLevel )
   , storage_( storage )
   {}

   // :51-64
   void apply( const srcType& src, const dstType& dst, uint_t level, DoFType flag, UpdateType = Replace ) const
   {
      if ( &src == &dst )
         throw std::runtime_error( "P1P1StokesOperator::apply: src and dst must differ" );
      lapl.apply( src.uvw(), dst.uvw(), level, flag, Replace );
      divT.apply( src.p(), dst.uvw(), level, flag, Add );
      div.apply( src.uvw(), dst.p(), level, flag, Replace );
      pspg.apply( src.p(), dst.p(), level, flag, Add );
   }
   const P1ConstantLaplaceOperator&    getA() const { return lapl.getSubOperator( 0, 0 ); }
   std::shared_ptr< PrimitiveStorage > getStorage() const { return storage_; }
   uint64_t                            uid() const { return uid_; }

   P1ConstantVectorLaplaceOperator lapl;
   P1ConstantDivOperator           div;
   P1ConstantDivTOperator          divT;
   P1PSPGOperator                  pspg;
   P1PSPGInvDiagOperator           pspg_inv_diag_;

 private:
   std::shared_ptr< PrimitiveStorage > storage_;
   uint64_t                            uid_ = nextUid();
};

// =====================================================================================================
// Smoothers and grid transfer
// =====================================================================================================
// StokesVelocityBlockBlockDiagonalPreconditioner.hpp:34-56: the scalar smoother on every velocity component
template < class OperatorType >
class StokesVelocityBlockBlockDiagonalPreconditioner : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType;
   StokesVelocityBlockBlockDiagonalPreconditioner( const std::shared_ptr< PrimitiveStorage >&,
                                                   std::shared_ptr< Solver< typename OperatorType::VelocityOperator_T > > scalar )
   : scalar_( std::move( scalar ) )
   {}
   void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) override
   {
      // the components are independent: one solveMany instead of a loop, so that a smoother can sweep them with shared launches
      std::vector< std::reference_wrapper< const P1Function< double > > > xs, bs;
      for ( uint_t k = 0; k < x.uvw().getDimension(); ++k )
      {
         xs.push_back( std::cref( x.uvw()[k] ) );
         bs.push_back( std::cref( b.uvw()[k] ) );
      }
      scalar_->solveMany( A.getA(), xs, bs, level );
   }

 private:
   std::shared_ptr< Solver< typename OperatorType::VelocityOperator_T > > scalar_;
};

// UzawaSmoother.hpp:99-330, block-Laplace variant with PSPG stabilisation (:262-288):
//   r.uvw = b.uvw - divT x.p;   numGSIterationsVelocity x velocitySmoother( A, x, r );
//   r.p   = relax * ( b.p - ( pspg x.p + div x.uvw ) );   x.p += pspg_inv_diag r.p
template < class OperatorType >
class UzawaSmoother : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType;
   UzawaSmoother( const std::shared_ptr< PrimitiveStorage >& storage,
                  std::shared_ptr< Solver< OperatorType > >  velocitySmoother,
                  uint_t                                     minLevel,
                  uint_t                                     maxLevel,
                  double                                     relaxParam,
                  DoFType                                    flag                    = Inner | NeumannBoundary | FreeslipBoundary,
                  uint_t                                     numGSIterationsVelocity = 2 )
   : velocitySmoother_( std::move( velocitySmoother ) )
   , flag_( flag )
   , relaxParam_( relaxParam )
   , numGSIterationsVelocity_( numGSIterationsVelocity )
   , r_( "uzawa_smoother_r", storage, minLevel, maxLevel )
   {}
   void setRelaxationParameter( double omega ) { relaxParam_ = omega; }

   void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) override
   {
      A.divT.apply( x.p(), r_.uvw(), level, flag_, Replace );
      r_.uvw().assign( { 1.0, -1.0 }, { b.uvw(), r_.uvw() }, level, flag_ );
      for ( uint_t i = 0; i < numGSIterationsVelocity_; ++i )
         velocitySmoother_->solve( A, x, r_, level );
      // has_pspg_block (composites/StokesOperatorTraits.hpp): with a stabilised operator the pressure residual contains C p
      // (UzawaSmoother.hpp:262-288), without one -- Taylor-Hood -- it is b.p - div u (:390-450); both scale it with the inverse
      // diagonal of the PSPG operator
      if constexpr ( OperatorType::hasPspgBlock )
      {
         A.pspg.apply( x.p(), r_.p(), level, flag_, Replace );
         A.div.apply( x.uvw(), r_.p(), level, flag_, Add );
      }
      else
         A.div.apply( x.uvw(), r_.p(), level, flag_, Replace );
      r_.p().assign( { 1.0, -1.0 }, { b.p(), r_.p() }, level, flag_ );
      r_.p().assign( { relaxParam_ }, { r_.p() }, level, flag_ );
      A.pspg_inv_diag_.apply( r_.p(), x.p(), level, flag_, Add );
   }

 private:
   std::shared_ptr< Solver< OperatorType > > velocitySmoother_;
   DoFType                                   flag_;
   double                                    relaxParam_;
   uint_t                                    numGSIterationsVelocity_;
   FunctionType                              r_;
};

// P1P1StokesToP1P1StokesRestriction.hpp:34-58 / ...Prolongation.hpp
class P1P1StokesToP1P1StokesRestriction
{
 public:
   explicit P1P1StokesToP1P1StokesRestriction( bool projectMeanAfterRestriction = false )
   : projectMean_( projectMeanAfterRestriction )
   {}
   void restrict( const P1StokesFunction< double >& f, uint_t sourceLevel, DoFType flag ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         r_.restrict( f.uvw()[k], sourceLevel, flag );
      r_.restrict( f.p(), sourceLevel, flag );
      if ( projectMean_ )
         projectMean( f.p(), sourceLevel - 1 );
   }

 private:
   P1toP1LinearRestriction r_;
   bool                    projectMean_;
};
class P1P1StokesToP1P1StokesProlongation
{
 public:
   void prolongate( const P1StokesFunction< double >& f, uint_t sourceLevel, DoFType flag ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         p_.prolongate( f.uvw()[k], sourceLevel, flag );
      p_.prolongate( f.p(), sourceLevel, flag );
   }
   void prolongateAndAdd( const P1StokesFunction< double >& f, uint_t sourceLevel, DoFType flag ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         p_.prolongateAndAdd( f.uvw()[k], sourceLevel, flag );
      p_.prolongateAndAdd( f.p(), sourceLevel, flag );
   }

 private:
   P1toP1LinearProlongation p_;
};

// Coarse-grid solver of the saddle-point system, standing in for PETScLUSolver< P1P1StokesOperator >
// (src/hyteg/petsc/PETScLUSolver.hpp; the convergence tests use it on level 2): the level's operator is assembled into a
// dense matrix by applying it to unit vectors (setup, once per level: 4 x #points applications of a few hundred points
// each), factorised on the host with partial pivoting, and every solve is a download, two triangular solves and an
// upload.  The pressure is determined up to a constant: the matrix is bordered with the zero-mean constraint.
// Single rank only (a coarse-grid direct solve across ranks would gather the system; not needed by the tests here).
template < class OperatorType >
class DenseCoarseGridSolver : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType;
   DenseCoarseGridSolver( const std::shared_ptr< PrimitiveStorage >& storage, uint_t level )
   : storage_( storage )
   , level_( level )
   , e_( "dense_coarse_e", storage, level, level )
   , Ae_( "dense_coarse_Ae", storage, level, level )
   {
      if ( storage->numRanks() != 1 )
         throw std::runtime_error( "DenseCoarseGridSolver: single rank only" );
   }

   void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) override
   {
      if ( level != level_ )
         throw std::runtime_error( "DenseCoarseGridSolver: built for another level" );
      if ( !factorised_ )
         factorise( A );
      // right-hand side: b on free DoFs; Dirichlet velocity DoFs keep the value x has there (identity rows)
      std::vector< double > rhs( n_ + 1, 0.0 ), xv = gather( x ), bv = gather( b );
      for ( uint_t i = 0; i < n_; ++i )
         rhs[i] = free_[i] ? bv[i] : xv[i];
      // forward / backward substitution with the stored LU factors
      std::vector< double > y( n_ + 1 );
      for ( uint_t i = 0; i <= n_; ++i )
      {
         double s = rhs[perm_[i]];
         for ( uint_t j = 0; j < i; ++j )
            s -= lu_[i * ( n_ + 1 ) + j] * y[j];
         y[i] = s;
      }
      for ( uint_t ii = n_ + 1; ii-- > 0; )
      {
         double s = y[ii];
         for ( uint_t j = ii + 1; j <= n_; ++j )
            s -= lu_[ii * ( n_ + 1 ) + j] * y[j];
         y[ii] = s / lu_[ii * ( n_ + 1 ) + ii];
      }
      scatter( x, y );
   }

 private:
   // global numbering: component c (0..2 velocity, 3 pressure) x global point; a global point = one representative
   // (local cell, array index) per group of copies, found through the exchange-free property that copies agree
   struct PointRef
   {
      uint_t cell;
      int    index;
   };
   void enumeratePoints()
   {
      const int64_t N = layout::width( (int) level_ ), total = layout::cellSize( (int) level_ );
      const uint_t  nCells = storage_->getNumberOfLocalCells();
      // identify copies by their physical coordinates (exact in binary for the dyadic refinement of one macro-vertex set:
      // use the barycentric integer coordinates with the GLOBAL macro-vertex ids, sorted)
      std::map< std::vector< int64_t >, uint_t > ids;
      pointOf_.assign( nCells, std::vector< int >( (size_t) total, -1 ) );
      for ( uint_t c = 0; c < nCells; ++c )
      {
         const MacroCell& cell = storage_->getLocalCell( c );
         for ( int64_t z = 0; z < N; ++z )
            for ( int64_t y = 0; y < N - z; ++y )
               for ( int64_t x = 0; x < N - z - y; ++x )
               {
                  const int64_t bary[4] = { N - 1 - x - y - z, x, y, z };
                  std::vector< std::pair< int, int64_t > > key;
                  for ( int k = 0; k < 4; ++k )
                     if ( bary[k] != 0 )
                        key.push_back( { cell.v[k], bary[k] } );
                  std::sort( key.begin(), key.end() );
                  std::vector< int64_t > flat;
                  for ( auto& kv : key )
                  {
                     flat.push_back( kv.first );
                     flat.push_back( kv.second );
                  }
                  auto it = ids.find( flat );
                  if ( it == ids.end() )
                  {
                     it = ids.emplace( flat, (uint_t) reps_.size() ).first;
                     reps_.push_back( { c, (int) layout::cellIndex( N, x, y, z ) } );
                  }
                  pointOf_[c][(size_t) layout::cellIndex( N, x, y, z )] = (int) it->second;
               }
      }
      np_ = reps_.size();
      n_  = 4 * np_;
   }
   const P1Function< double >& comp( const FunctionType& f, uint_t c ) const { return c < 3 ? f.uvw()[c] : f.p(); }
   // device-side packing (round 3; the first version moved every cell array through the host by itself: 281,000 synchronous copies
   // at set-up and 288 per coarse-grid solve): index tables built once, one pack launch + one download per gather, one upload +
   // one pack launch per cell array per scatter -- hyteg_hip_gather_entries( out, bases, buf, off, n ): out[k] = bases[buf[k]][off[k]]
   void buildIndexTables() const
   {
      if ( gatherBuf_ )
         return;
      const uint_t      nCells = storage_->getNumberOfLocalCells();
      const int64_t     total  = layout::cellSize( (int) level_ );
      std::vector< int > gb( n_ ), go( n_ );
      for ( uint_t c = 0; c < 4; ++c )
         for ( uint_t g = 0; g < np_; ++g )
         {
            gb[c * np_ + g] = (int) ( c * nCells + reps_[g].cell ); // position of the array in basesOf( f )
            go[c * np_ + g] = reps_[g].index;
         }
      gatherBuf_ = static_cast< const int* >( storage_->uploadBytes( gb.data(), gb.size() * sizeof( int ) ) );
      gatherOff_ = static_cast< const int* >( storage_->uploadBytes( go.data(), go.size() * sizeof( int ) ) );
      // scatter: entry i of the array of (component c, cell) reads V[ c np + pointOf[cell][i] ] -- one table for all components, the
      // component's offset comes through the base pointer V + c np
      std::vector< int > so( nCells * (size_t) total ), zeros( (size_t) total, 0 );
      for ( uint_t cell = 0; cell < nCells; ++cell )
         for ( int64_t i = 0; i < total; ++i )
            so[cell * (size_t) total + (size_t) i] = pointOf_[cell][(size_t) i];
      scatterOff_  = static_cast< const int* >( storage_->uploadBytes( so.data(), so.size() * sizeof( int ) ) );
      scatterZero_ = static_cast< const int* >( storage_->uploadBytes( zeros.data(), zeros.size() * sizeof( int ) ) );
      std::vector< double > none( n_ + 1, 0.0 );
      packed_ = storage_->uploadTable( none );
      std::vector< double* > vb;
      for ( uint_t c = 0; c < 4; ++c )
         vb.push_back( packed_ + c * np_ );
      packedBases_ = storage_->pointerTable( vb );
   }
   double** basesOf( const FunctionType& f ) const
   {
      std::vector< double* > b;
      for ( uint_t c = 0; c < 4; ++c )
         for ( uint_t cell = 0; cell < storage_->getNumberOfLocalCells(); ++cell )
            b.push_back( comp( f, c ).getCellPointer( cell, level_ ) );
      return storage_->pointerTable( b );
   }
   std::vector< double > gather( const FunctionType& f ) const
   {
      buildIndexTables();
      std::vector< double > v( n_ );
      hipCheck( hyteg_hip_gather_entries( packed_, basesOf( f ), gatherBuf_, gatherOff_, (int) n_, storage_->stream() ), "dense coarse solver: gather" );
      hipCheck( hyteg_hip_download( v.data(), packed_, n_ * sizeof( double ), storage_->stream() ), "dense coarse solver: download" );
      hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "dense coarse solver: sync" );
      return v;
   }
   // components [cFirst, cLast] of f := v
   void scatter( const FunctionType& f, const std::vector< double >& v, uint_t cFirst = 0, uint_t cLast = 3 ) const
   {
      buildIndexTables();
      const int64_t total = layout::cellSize( (int) level_ );
      hipCheck( hyteg_hip_upload( packed_, v.data(), n_ * sizeof( double ), storage_->stream() ), "dense coarse solver: upload" );
      for ( uint_t c = cFirst; c <= cLast; ++c )
         for ( uint_t cell = 0; cell < storage_->getNumberOfLocalCells(); ++cell )
            hipCheck( hyteg_hip_gather_entries( comp( f, c ).getCellPointer( cell, level_ ), packedBases_ + c, scatterZero_,
                                                scatterOff_ + cell * (size_t) total, (int) total, storage_->stream() ),
                      "dense coarse solver: scatter" );
      hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "dense coarse solver: sync" ); // v may go out of scope
   }
   void factorise( const OperatorType& A )
   {
      enumeratePoints();
      const DoFType flag = Inner | NeumannBoundary | FreeslipBoundary;
      // free DoFs: where the operator writes (velocity: not on the Dirichlet boundary; pressure: everywhere)
      {
         e_.interpolate( 0.0, level_, All );
         e_.interpolate( 1.0, level_, flag );
         const auto m = gather( e_ );
         free_.assign( n_, false );
         for ( uint_t i = 0; i < n_; ++i )
            free_[i] = m[i] != 0.0;
      }
      const uint_t          m = n_ + 1;
      std::vector< double > M( m * m, 0.0 );
      std::vector< double > unit( n_, 0.0 );
      scatter( e_, unit );
      for ( uint_t j = 0; j < n_; ++j )
      {
         // only the component that holds DoF j changes (and the previous one when j is a component's first DoF)
         const uint_t c = j / np_;
         if ( j > 0 )
            unit[j - 1] = 0.0;
         unit[j] = 1.0;
         scatter( e_, unit, ( j % np_ == 0 && c > 0 ) ? c - 1 : c, c );
         Ae_.interpolate( 0.0, level_, All );
         A.apply( e_, Ae_, level_, flag );
         const auto col = gather( Ae_ );
         for ( uint_t i = 0; i < n_; ++i )
            M[i * m + j] = free_[i] ? col[i] : ( i == j ? 1.0 : 0.0 );
      }
      // zero-mean pressure: Lagrange multiplier row / column over the pressure DoFs
      for ( uint_t k = 0; k < np_; ++k )
      {
         M[n_ * m + 3 * np_ + k] = 1.0;
         M[( 3 * np_ + k ) * m + n_] = 1.0;
      }
      // LU with partial pivoting; perm_[i] = original row that ended up in row i
      perm_.resize( m );
      for ( uint_t i = 0; i < m; ++i )
         perm_[i] = i;
      for ( uint_t k = 0; k < m; ++k )
      {
         uint_t piv = k;
         for ( uint_t i = k + 1; i < m; ++i )
            if ( std::fabs( M[i * m + k] ) > std::fabs( M[piv * m + k] ) )
               piv = i;
         if ( M[piv * m + k] == 0.0 )
            throw std::runtime_error( "DenseCoarseGridSolver: singular matrix" );
         if ( piv != k )
         {
            for ( uint_t j = 0; j < m; ++j )
               std::swap( M[k * m + j], M[piv * m + j] );
            std::swap( perm_[k], perm_[piv] );
         }
         for ( uint_t i = k + 1; i < m; ++i )
         {
            const double f = M[i * m + k] / M[k * m + k];
            M[i * m + k]   = f;
            if ( f != 0.0 )
               for ( uint_t j = k + 1; j < m; ++j )
                  M[i * m + j] -= f * M[k * m + j];
         }
      }
      lu_.swap( M );
      factorised_ = true;
   }

   std::shared_ptr< PrimitiveStorage > storage_;
   uint_t                              level_;
   FunctionType                        e_, Ae_;
   bool                                factorised_ = false;
   uint_t                              np_ = 0, n_ = 0;
   std::vector< PointRef >             reps_;
   std::vector< std::vector< int > >   pointOf_;
   std::vector< bool >                 free_;
   std::vector< double >               lu_;
   std::vector< uint_t >               perm_;
   mutable const int *                 gatherBuf_ = nullptr, *gatherOff_ = nullptr, *scatterOff_ = nullptr, *scatterZero_ = nullptr;
   mutable double*                     packed_      = nullptr;
   mutable double**                    packedBases_ = nullptr;
};

} // namespace hyteg
