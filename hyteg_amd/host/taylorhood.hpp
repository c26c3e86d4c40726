// taylorhood.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// P2-P1 Taylor-Hood Stokes composition (BASELINE config 5's "P2-P1 Stokes block operator"):
//   P2P1TaylorHoodFunction            src/hyteg/composites/P2P1TaylorHoodFunction.hpp
//   P2ToP1 / P1ToP2 mixed operators   src/mixed_operator/P2ToP1ConstantOperator.hpp:47-101, P1ToP2ConstantOperator.hpp
//   P2P1TaylorHoodStokesOperator      src/mixed_operator/P2P1TaylorHoodStokesOperator.hpp:34-110
//   transfer                          src/hyteg/gridtransferoperators/P2P1StokesToP2P1Stokes{Restriction,Prolongation}.hpp
// No new device code: a P2 -> P1 block is the P2 apply kernel asked for its vertex-DoF rows only, with an element matrix whose
// edge rows are zero; a P1 -> P2 block is the same kernel with zero edge columns and a zero edge-DoF source array.
#pragma once

#include "minres.hpp"
#include "p2operator.hpp"
#include "p2gridtransfer.hpp"
#include "stokes.hpp"

namespace hyteg {

namespace forms {
// int_T lambda_i d_k phi_j (i: P1 shape function, j: P2 shape function in FEniCS ordering: vertices 0-3, edges (2,3) (1,3) (1,2) (0,3)
// (0,2) (0,1)) in closed form: phi_a = lambda_a ( 2 lambda_a - 1 ), phi_ab = 4 lambda_a lambda_b,
// int lambda_i = V / 4, int lambda_i lambda_p = V ( 1 + delta_ip ) / 20.
inline void p2GradientAgainstP1( const std::array< Point3D, 4 >& c, int k, double B[4][10] )
{
   double J[3][3];
   for ( int r = 0; r < 3; ++r )
      for ( int q = 0; q < 3; ++q )
         J[r][q] = c[q + 1][r] - c[0][r];
   const double det = det3( J );
   double       Ji[3][3];
   Ji[0][0] = ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) / det;
   Ji[0][1] = ( J[0][2] * J[2][1] - J[0][1] * J[2][2] ) / det;
   Ji[0][2] = ( J[0][1] * J[1][2] - J[0][2] * J[1][1] ) / det;
   Ji[1][0] = ( J[1][2] * J[2][0] - J[1][0] * J[2][2] ) / det;
   Ji[1][1] = ( J[0][0] * J[2][2] - J[0][2] * J[2][0] ) / det;
   Ji[1][2] = ( J[0][2] * J[1][0] - J[0][0] * J[1][2] ) / det;
   Ji[2][0] = ( J[1][0] * J[2][1] - J[1][1] * J[2][0] ) / det;
   Ji[2][1] = ( J[0][1] * J[2][0] - J[0][0] * J[2][1] ) / det;
   Ji[2][2] = ( J[0][0] * J[1][1] - J[0][1] * J[1][0] ) / det;
   double g[4]; // d_k lambda_a
   g[1] = Ji[0][k], g[2] = Ji[1][k], g[3] = Ji[2][k];
   g[0] = -( g[1] + g[2] + g[3] );
   const double     V           = std::fabs( det ) / 6.0;
   static const int pairs[6][2] = { { 2, 3 }, { 1, 3 }, { 1, 2 }, { 0, 3 }, { 0, 2 }, { 0, 1 } };
   for ( int i = 0; i < 4; ++i )
   {
      for ( int a = 0; a < 4; ++a ) // grad phi_a = ( 4 lambda_a - 1 ) grad lambda_a
         B[i][a] = g[a] * V * ( i == a ? 3.0 / 20.0 : -1.0 / 20.0 );
      for ( int e = 0; e < 6; ++e ) // grad phi_ab = 4 ( lambda_b grad lambda_a + lambda_a grad lambda_b )
      {
         const int a = pairs[e][0], b = pairs[e][1];
         B[i][4 + e] = 4.0 * V / 20.0 * ( g[a] * ( i == b ? 2.0 : 1.0 ) + g[b] * ( i == a ? 2.0 : 1.0 ) );
      }
   }
}
// p2_to_p1_tet_div_tet_cell_integral_K_otherwise ( - int q d_K u ) as a 10 x 10 matrix whose edge ROWS are zero
template < int K >
struct P2ToP1DivForm
{
   static void integrateAll( const std::array< Point3D, 4 >& c, double elMat[100] )
   {
      double B[4][10];
      p2GradientAgainstP1( c, K, B );
      for ( int k = 0; k < 100; ++k )
         elMat[k] = 0.0;
      for ( int i = 0; i < 4; ++i )
         for ( int j = 0; j < 10; ++j )
            elMat[10 * i + j] = -B[i][j];
   }
};
// p1_to_p2_tet_divt_tet_cell_integral_K_otherwise ( - int p d_K v ) as a 10 x 10 matrix whose edge COLUMNS are zero
template < int K >
struct P1ToP2DivTForm
{
   static void integrateAll( const std::array< Point3D, 4 >& c, double elMat[100] )
   {
      double B[4][10];
      p2GradientAgainstP1( c, K, B );
      for ( int k = 0; k < 100; ++k )
         elMat[k] = 0.0;
      for ( int j = 0; j < 10; ++j )
         for ( int i = 0; i < 4; ++i )
            elMat[10 * j + i] = -B[i][j];
   }
};
} // namespace forms

// P2ToP1ConstantOperator< Form > (P2ToP1ConstantOperator.hpp:47): dst( P1 ) = / += A src( P2 )
template < class Form >
class P2ToP1Operator : public P2ElementwiseOperator< Form >
{
   using Base = P2ElementwiseOperator< Form >;

 public:
   P2ToP1Operator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : Base( storage, minLevel, maxLevel )
   , tmp_( "p2_to_p1_tmp", storage, minLevel, maxLevel )
   , tmpP1_( "p2_to_p1_tmp_p1", storage, minLevel, maxLevel )
   {}
   void apply( const P2Function< double >& src, const P1Function< double >& dst, uint_t level, DoFType flagIn, UpdateType updateType = Replace ) const
   {
      const DoFType flag = dst.effectiveFlag( flagIn );
      tmpP1_.setBoundaryConditionAllInner( dst.hasAllInnerBoundaryCondition() );
      std::vector< double* >       dv, de;
      std::vector< const double* > sv, se;
      for ( uint_t c = 0; c < this->storage_->getNumberOfLocalCells(); ++c )
      {
         dv.push_back( tmpP1_.getCellPointer( c, level ) ), de.push_back( tmp_.getEdgeCellPointer( c, level ) );
         sv.push_back( src.getVertexDoFFunction().getCellPointer( c, level ) ), se.push_back( src.getEdgeCellPointer( c, level ) );
      }
      this->launchPointers( this->elementMatrices_.at( level ), 1.0, sv, se, dv, de, level, this->storage_->masksFor( flag ), HYTEG_HIP_REPLACE, 1u );
      if ( this->storage_->getCells().size() > 1 )
         tmpP1_.sumSharedCopies( level, flagIn );
      if ( updateType == Replace )
         dst.assign( { 1.0 }, { tmpP1_ }, level, flagIn );
      else
         dst.add( { 1.0 }, { tmpP1_ }, level, flagIn );
   }

 private:
   P2Function< double >         tmp_; // its edge array is the (never written) edge destination the kernel's signature asks for
   mutable P1Function< double > tmpP1_;
};

// P1ToP2ConstantOperator< Form > (P1ToP2ConstantOperator.hpp): dst( P2 ) = / += A src( P1 )
template < class Form >
class P1ToP2Operator : public P2ElementwiseOperator< Form >
{
   using Base = P2ElementwiseOperator< Form >;

 public:
   P1ToP2Operator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : Base( storage, minLevel, maxLevel )
   , tmp_( "p1_to_p2_tmp", storage, minLevel, maxLevel )
   , zero_( "p1_to_p2_zero", storage, minLevel, maxLevel )
   {}
   void apply( const P1Function< double >& src, const P2Function< double >& dst, uint_t level, DoFType flag, UpdateType updateType = Replace ) const
   {
      std::vector< double* >       dv, de;
      std::vector< const double* > sv, se;
      for ( uint_t c = 0; c < this->storage_->getNumberOfLocalCells(); ++c )
      {
         dv.push_back( tmp_.getVertexDoFFunction().getCellPointer( c, level ) ), de.push_back( tmp_.getEdgeCellPointer( c, level ) );
         sv.push_back( src.getCellPointer( c, level ) ), se.push_back( zero_.getEdgeCellPointer( c, level ) );
      }
      this->launchPointers( this->elementMatrices_.at( level ), 1.0, sv, se, dv, de, level, this->storage_->masksFor( flag ), HYTEG_HIP_REPLACE, 0xFFu );
      if ( this->storage_->getCells().size() > 1 )
      {
         tmp_.getVertexDoFFunction().sumSharedCopies( level, flag );
         tmp_.sumSharedEdgeCopies( level, flag );
      }
      if ( updateType == Replace )
         dst.assign( { 1.0 }, { tmp_ }, level, flag );
      else
         dst.add( { 1.0 }, { tmp_ }, level, flag );
   }

 private:
   P2Function< double > tmp_;
   P2Function< double > zero_; // zero edge-DoF source (the element matrix has zero edge columns; the kernel still reads the array)
};

// P2VectorFunction (src/hyteg/p2functionspace/P2VectorFunction.hpp), three components
template < typename ValueType >
class P2VectorFunction
{
 public:
   P2VectorFunction( const std::string& name, const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   {
      static const char* suffix[3] = { "_u", "_v", "_w" };
      for ( int k = 0; k < 3; ++k )
         comp_.push_back( std::make_shared< P2Function< ValueType > >( name + suffix[k], storage, minLevel, maxLevel ) );
   }
   uint_t                         getDimension() const { return 3; }
   const P2Function< ValueType >& operator[]( uint_t k ) const { return *comp_.at( k ); }

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
   void assign( const std::vector< ValueType >& scalars, const std::vector< std::reference_wrapper< const P2VectorFunction< ValueType > > >& functions,
                uint_t level, DoFType flag = All ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         comp_[k]->assign( scalars, componentRefs( functions, k ), level, flag );
   }
   void add( const std::vector< ValueType >& scalars, const std::vector< std::reference_wrapper< const P2VectorFunction< ValueType > > >& functions,
             uint_t level, DoFType flag = All ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         comp_[k]->add( scalars, componentRefs( functions, k ), level, flag );
   }
   ValueType dotGlobal( const P2VectorFunction< ValueType >& rhs, uint_t level, DoFType flag = All ) const
   {
      ValueType s = 0;
      for ( uint_t k = 0; k < 3; ++k )
         s += comp_[k]->dotGlobal( rhs[k], level, flag );
      return s;
   }

 private:
   static std::vector< std::reference_wrapper< const P2Function< ValueType > > >
       componentRefs( const std::vector< std::reference_wrapper< const P2VectorFunction< ValueType > > >& functions, uint_t k )
   {
      std::vector< std::reference_wrapper< const P2Function< ValueType > > > r;
      for ( const auto& f : functions )
         r.push_back( std::cref( f.get()[k] ) );
      return r;
   }
   std::vector< std::shared_ptr< P2Function< ValueType > > > comp_;
};

// P2P1TaylorHoodFunction (composites/P2P1TaylorHoodFunction.hpp): P2 velocity with the storage's boundary types, P1 pressure with
// createAllInnerBC
template < typename ValueType >
class P2P1TaylorHoodFunction
{
 public:
   using valueType = ValueType;
   P2P1TaylorHoodFunction( const std::string& name, const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : uvw_( name + "_uvw", storage, minLevel, maxLevel )
   , p_( name + "_p", storage, minLevel, maxLevel )
   {
      p_.setBoundaryConditionAllInner();
   }
   const P2VectorFunction< ValueType >& uvw() const { return uvw_; }
   const P1Function< ValueType >&       p() const { return p_; }
   uint64_t                             uid() const { return p_.uid(); }

   void interpolate( ValueType constant, uint_t level, DoFType flag = All ) const
   {
      uvw_.interpolate( constant, level, flag );
      p_.interpolate( constant, level, flag );
   }
   void assign( const std::vector< ValueType >&                                                              scalars,
                const std::vector< std::reference_wrapper< const P2P1TaylorHoodFunction< ValueType > > >& functions, uint_t level,
                DoFType flag = All ) const
   {
      std::vector< std::reference_wrapper< const P2VectorFunction< ValueType > > > v;
      std::vector< std::reference_wrapper< const P1Function< ValueType > > >       q;
      for ( const auto& f : functions )
         v.push_back( std::cref( f.get().uvw() ) ), q.push_back( std::cref( f.get().p() ) );
      uvw_.assign( scalars, v, level, flag );
      p_.assign( scalars, q, level, flag );
   }
   void add( const std::vector< ValueType >&                                                              scalars,
             const std::vector< std::reference_wrapper< const P2P1TaylorHoodFunction< ValueType > > >& functions, uint_t level,
             DoFType flag = All ) const
   {
      std::vector< std::reference_wrapper< const P2VectorFunction< ValueType > > > v;
      std::vector< std::reference_wrapper< const P1Function< ValueType > > >       q;
      for ( const auto& f : functions )
         v.push_back( std::cref( f.get().uvw() ) ), q.push_back( std::cref( f.get().p() ) );
      uvw_.add( scalars, v, level, flag );
      p_.add( scalars, q, level, flag );
   }
   ValueType dotGlobal( const P2P1TaylorHoodFunction< ValueType >& rhs, uint_t level, DoFType flag = All ) const
   {
      return uvw_.dotGlobal( rhs.uvw(), level, flag ) + p_.dotGlobal( rhs.p(), level, flag );
   }

 private:
   P2VectorFunction< ValueType > uvw_;
   P1Function< ValueType >       p_;
};

// the three div / divT blocks as one operator (VectorToScalarOperator / ScalarToVectorOperator, src/mixed_operator/)
class P2ToP1DivOperator
{
 public:
   P2ToP1DivOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : x_( storage, minLevel, maxLevel )
   , y_( storage, minLevel, maxLevel )
   , z_( storage, minLevel, maxLevel )
   {}
   void apply( const P2VectorFunction< double >& src, const P1Function< double >& dst, uint_t level, DoFType flag, UpdateType updateType = Replace ) const
   {
      x_.apply( src[0], dst, level, flag, updateType );
      y_.apply( src[1], dst, level, flag, Add );
      z_.apply( src[2], dst, level, flag, Add );
   }

 private:
   P2ToP1Operator< forms::P2ToP1DivForm< 0 > > x_;
   P2ToP1Operator< forms::P2ToP1DivForm< 1 > > y_;
   P2ToP1Operator< forms::P2ToP1DivForm< 2 > > z_;
};
class P1ToP2DivTOperator
{
 public:
   P1ToP2DivTOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : x_( storage, minLevel, maxLevel )
   , y_( storage, minLevel, maxLevel )
   , z_( storage, minLevel, maxLevel )
   {}
   void apply( const P1Function< double >& src, const P2VectorFunction< double >& dst, uint_t level, DoFType flag, UpdateType updateType = Replace ) const
   {
      x_.apply( src, dst[0], level, flag, updateType );
      y_.apply( src, dst[1], level, flag, updateType );
      z_.apply( src, dst[2], level, flag, updateType );
   }

 private:
   P1ToP2Operator< forms::P1ToP2DivTForm< 0 > > x_;
   P1ToP2Operator< forms::P1ToP2DivTForm< 1 > > y_;
   P1ToP2Operator< forms::P1ToP2DivTForm< 2 > > z_;
};
// P2ConstantVectorLaplaceOperator (VectorLaplaceOperator.hpp): the scalar operator on every component
class P2ConstantVectorLaplaceOperator
{
 public:
   P2ConstantVectorLaplaceOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : lapl_( storage, minLevel, maxLevel )
   {
      lapl_.computeInverseDiagonalOperatorValues();
   }
   void apply( const P2VectorFunction< double >& src, const P2VectorFunction< double >& dst, uint_t level, DoFType flag, UpdateType updateType = Replace ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         lapl_.apply( src[k], dst[k], level, flag, updateType );
   }
   const P2ConstantLaplaceOperator& getSubOperator( uint_t, uint_t ) const { return lapl_; }

 private:
   P2ConstantLaplaceOperator lapl_;
};

// P2P1TaylorHoodStokesOperator.hpp:34-110 (apply :55-64; the PSPG members serve the Uzawa smoother as Schur-complement approximation)
class P2P1TaylorHoodStokesOperator
{
 public:
   using srcType            = P2P1TaylorHoodFunction< double >;
   using dstType            = P2P1TaylorHoodFunction< double >;
   using VelocityOperator_T = P2ConstantLaplaceOperator;
   static constexpr bool hasPspgBlock = false; // no specialisation of has_pspg_block for this operator (StokesOperatorTraits.hpp:25-29)

   P2P1TaylorHoodStokesOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : lapl( storage, minLevel, maxLevel )
   , div( storage, minLevel, maxLevel )
   , divT( storage, minLevel, maxLevel )
   , pspg( storage, minLevel, maxLevel )
   , pspg_inv_diag_( storage, minLevel, maxLevel )
   , storage_( storage )
   {}
   void apply( const srcType& src, const dstType& dst, uint_t level, DoFType flag, UpdateType = Replace ) const
   {
      if ( &src == &dst )
         throw std::runtime_error( "P2P1TaylorHoodStokesOperator::apply: src and dst must differ" );
      lapl.apply( src.uvw(), dst.uvw(), level, flag, Replace );
      divT.apply( src.p(), dst.uvw(), level, flag, Add );
      div.apply( src.uvw(), dst.p(), level, flag, Replace );
   }
   const P2ConstantLaplaceOperator&    getA() const { return lapl.getSubOperator( 0, 0 ); }
   std::shared_ptr< PrimitiveStorage > getStorage() const { return storage_; }
   uint64_t                            uid() const { return uid_; }

   P2ConstantVectorLaplaceOperator lapl;
   P2ToP1DivOperator               div;
   P1ToP2DivTOperator              divT;
   P1PSPGOperator                  pspg;
   P1PSPGInvDiagOperator           pspg_inv_diag_;

 private:
   std::shared_ptr< PrimitiveStorage > storage_;
   uint64_t                            uid_ = nextUid();
};

// StokesVelocityBlockBlockDiagonalPreconditioner for the Taylor-Hood operator: the scalar P2 smoother on every velocity component
class TaylorHoodVelocityBlockPreconditioner : public Solver< P2P1TaylorHoodStokesOperator >
{
 public:
   explicit TaylorHoodVelocityBlockPreconditioner( std::shared_ptr< Solver< P2ConstantLaplaceOperator > > scalar )
   : scalar_( std::move( scalar ) )
   {}
   void solve( const P2P1TaylorHoodStokesOperator& A, const P2P1TaylorHoodFunction< double >& x, const P2P1TaylorHoodFunction< double >& b,
               uint_t level ) override
   {
      for ( uint_t k = 0; k < 3; ++k )
         scalar_->solve( A.getA(), x.uvw()[k], b.uvw()[k], level );
   }

 private:
   std::shared_ptr< Solver< P2ConstantLaplaceOperator > > scalar_;
};

// P2P1StokesToP2P1StokesRestriction.hpp / ...Prolongation.hpp: quadratic transfer on the velocity, linear on the pressure
class P2P1StokesToP2P1StokesRestriction
{
 public:
   explicit P2P1StokesToP2P1StokesRestriction( bool projectMeanAfterRestriction = false )
   : projectMean_( projectMeanAfterRestriction )
   {}
   void restrict( const P2P1TaylorHoodFunction< double >& f, uint_t sourceLevel, DoFType flag ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         rv_.restrict( f.uvw()[k], sourceLevel, flag );
      rp_.restrict( f.p(), sourceLevel, flag );
      if ( projectMean_ )
         projectMean( f.p(), sourceLevel - 1 );
   }

 private:
   P2toP2QuadraticRestriction rv_;
   P1toP1LinearRestriction    rp_;
   bool                       projectMean_;
};
class P2P1StokesToP2P1StokesProlongation
{
 public:
   void prolongate( const P2P1TaylorHoodFunction< double >& f, uint_t sourceLevel, DoFType flag ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         pv_.prolongate( f.uvw()[k], sourceLevel, flag );
      pp_.prolongate( f.p(), sourceLevel, flag );
   }
   void prolongateAndAdd( const P2P1TaylorHoodFunction< double >& f, uint_t sourceLevel, DoFType flag ) const
   {
      for ( uint_t k = 0; k < 3; ++k )
         pv_.prolongateAndAdd( f.uvw()[k], sourceLevel, flag );
      pp_.prolongateAndAdd( f.p(), sourceLevel, flag );
   }

 private:
   P2toP2QuadraticProlongation pv_;
   P1toP1LinearProlongation    pp_;
};

} // namespace hyteg
