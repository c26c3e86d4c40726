// p2operator.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// P2 forms and P2ElementwiseOperator (src/hyteg/elementwiseoperators/P2ElementwiseOperator.cpp)
#pragma once

#include <cstdlib>

#include "p2elements.hpp"
#include "p2function.hpp"
#include "forms.hpp"
#include "p1operator.hpp"

namespace hyteg {

// =====================================================================================================
// P2ElementwiseOperator< P2Form >  ( src/hyteg/elementwiseoperators/P2ElementwiseOperator.hpp:454, .cpp:110-223 ), affine cells:
// the six element matrices per (cell, level) are computed once (kernel INPUT) and kept on the device.
// =====================================================================================================
namespace forms {
// P2 diffusion element matrix in FEniCS ordering (vertices 0-3, edges (2,3) (1,3) (1,2) (0,3) (0,2) (0,1)); what
// P2FenicsForm< ..., p2_tet_diffusion_cell_integral_0_otherwise >::integrateAll returns (form_fenics_base/P2FenicsForm.cpp:160-175).
// Closed form: phi_a = l_a (2 l_a - 1), phi_ab = 4 l_a l_b; int l_a = V/4, int l_a l_b = V (1 + delta_ab) / 20.
struct P2LaplaceForm
{
   static void integrateAll( const std::array< Point3D, 4 >& c, double elMat[100] )
   {
      double J[3][3];
      for ( int r = 0; r < 3; ++r )
         for ( int k = 0; k < 3; ++k )
            J[r][k] = c[k + 1][r] - c[0][r];
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
      double g[4][3];
      for ( int r = 0; r < 3; ++r )
      {
         g[1][r] = Ji[0][r], g[2][r] = Ji[1][r], g[3][r] = Ji[2][r];
         g[0][r] = -( Ji[0][r] + Ji[1][r] + Ji[2][r] );
      }
      const double V = std::fabs( det ) / 6.0;
      double       G[4][4];
      for ( int a = 0; a < 4; ++a )
         for ( int b = 0; b < 4; ++b )
            G[a][b] = g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2];
      // grad phi_i = sum_a ( sum_p C[i][a][p] l_p + D[i][a] ) grad l_a
      static const int pairs[6][2] = { { 2, 3 }, { 1, 3 }, { 1, 2 }, { 0, 3 }, { 0, 2 }, { 0, 1 } };
      double           C[10][4][4] = {}, D[10][4] = {};
      for ( int a = 0; a < 4; ++a )
         C[a][a][a] = 4.0, D[a][a] = -1.0;
      for ( int k = 0; k < 6; ++k )
         C[4 + k][pairs[k][0]][pairs[k][1]] = 4.0, C[4 + k][pairs[k][1]][pairs[k][0]] = 4.0;
      for ( int i = 0; i < 10; ++i )
         for ( int j = 0; j < 10; ++j )
         {
            double s = 0.0;
            for ( int a = 0; a < 4; ++a )
               for ( int b = 0; b < 4; ++b )
               {
                  double t = D[i][a] * D[j][b] * V;
                  for ( int p = 0; p < 4; ++p )
                  {
                     t += ( C[i][a][p] * D[j][b] + D[i][a] * C[j][b][p] ) * V / 4.0;
                     for ( int q = 0; q < 4; ++q )
                        t += C[i][a][p] * C[j][b][q] * V * ( p == q ? 2.0 : 1.0 ) / 20.0;
                  }
                  s += t * G[a][b];
               }
            elMat[10 * i + j] = s;
         }
   }
};
} // namespace forms

namespace forms {
// the vertex-vertex block of a P2 form as a P1-style form (first row of the element matrix, vertex columns): the
// vertex-to-vertex sub-operator of P2ConstantOperator (P2ConstantOperator.hpp: vertexToVertex), whose Gauss-Seidel sweep is the
// P1 one
template < class P2Form >
struct P2VertexToVertexForm
{
   static void integrateRow0( const std::array< Point3D, 4 >& c, double row[4] )
   {
      double h[100];
      P2Form::integrateAll( c, h );
      for ( int j = 0; j < 4; ++j )
         row[j] = h[j];
   }
};
} // namespace forms

template < class P2Form >
class P2ElementwiseOperator
{
 public:
   using srcType = P2Function< double >;
   using dstType = P2Function< double >;
   virtual ~P2ElementwiseOperator() = default;

 protected:
   // for derived operators that bring their own operator tables (P2ConstantOperator: from stencils it assembles itself)
   struct NoTables
   {};
   P2ElementwiseOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel, NoTables )
   : storage_( storage )
   , minLevel_( minLevel )
   , maxLevel_( maxLevel )
   {}
   // the operator table whose apply to the function 1 gives the operator's diagonal (computeInverseDiagonalOperatorValues)
   virtual std::vector< double > diagonalTable( uint_t level, uint_t localCell ) const
   {
      std::vector< double > h( 600, 0.0 );
      const auto&           full = hostMatrices_.at( level ).at( localCell );
      for ( int t = 0; t < 6; ++t )
         for ( int k = 0; k < 10; ++k )
            h[100 * t + 11 * k] = full[100 * t + 11 * k];
      std::vector< double > table( hyteg_hip_p2_operator_table_size() );
      hipCheck( hyteg_hip_p2_build_operator_table( h.data(), table.data() ), "P2ElementwiseOperator: diagonal table" );
      return table;
   }

 public:
   P2ElementwiseOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : storage_( storage )
   , minLevel_( minLevel )
   , maxLevel_( maxLevel )
   {
      for ( uint_t l = minLevel; l <= maxLevel; ++l )
         for ( uint_t lc = 0; lc < storage->getNumberOfLocalCells(); ++lc )
            addElementMatrixTable( l, lc );
   }

 protected:
   // the six element matrices of a macro-cell at level l (kernel INPUT)
   static std::vector< double > elementMatricesOf( const MacroCell& cell, uint_t l )
   {
      // micro-cell vertex offsets of the six cell types, celldof::macrocell::getMicroVerticesFromMicroCell (CellDoFIndexing.hpp:155-198)
      static const int verts[6][4][3] = {
          { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } }, { { 1, 0, 0 }, { 1, 1, 0 }, { 0, 1, 0 }, { 1, 0, 1 } },
          { { 1, 0, 0 }, { 0, 1, 0 }, { 1, 0, 1 }, { 0, 0, 1 } }, { { 1, 1, 0 }, { 1, 1, 1 }, { 0, 1, 1 }, { 1, 0, 1 } },
          { { 1, 0, 1 }, { 0, 1, 1 }, { 0, 0, 1 }, { 0, 1, 0 } }, { { 0, 1, 0 }, { 1, 1, 0 }, { 1, 0, 1 }, { 0, 1, 1 } } };
      const double          step = 1.0 / double( int64_t( 1 ) << l );
      std::vector< double > h( 600 );
      for ( int t = 0; t < 6; ++t )
      {
         std::array< Point3D, 4 > c;
         for ( int k = 0; k < 4; ++k )
            for ( int r = 0; r < 3; ++r )
               c[k][r] = cell.coords[0][r] + ( cell.coords[1][r] - cell.coords[0][r] ) * step * verts[t][k][0] +
                         ( cell.coords[2][r] - cell.coords[0][r] ) * step * verts[t][k][1] +
                         ( cell.coords[3][r] - cell.coords[0][r] ) * step * verts[t][k][2];
         P2Form::integrateAll( c, h.data() + 100 * t );
      }
      return h;
   }
   // the operator table (host copy) of ANY macro-cell of the mesh, local or not: the Gauss-Seidel tables need the shares of all
   // cells at a macro-face
   virtual std::vector< double > hostTableOf( const MacroCell& cell, uint_t l ) const
   {
      const std::vector< double > h = elementMatricesOf( cell, l );
      std::vector< double >       table( hyteg_hip_p2_operator_table_size() );
      hipCheck( hyteg_hip_p2_build_operator_table( h.data(), table.data() ), "P2ElementwiseOperator: operator table" );
      return table;
   }
   // the six element matrices of local cell lc at level l and the operator table built from them, uploaded
   void addElementMatrixTable( uint_t l, uint_t lc )
   {
      const std::vector< double > h = elementMatricesOf( storage_->getLocalCell( lc ), l );
      std::vector< double >       table( hyteg_hip_p2_operator_table_size() );
      hipCheck( hyteg_hip_p2_build_operator_table( h.data(), table.data() ), "P2ElementwiseOperator: operator table" );
      elementMatrices_[l].push_back( storage_->uploadTable( table ) );
      hostMatrices_[l].push_back( h );
   }

 public:
   uint64_t uid() const { return uid_; }

   // P2ElementwiseOperator::computeInverseDiagonalOperatorValues (P2ElementwiseOperator.hpp:110, computeDiagonalOperatorValues .cpp:420-520:
   // every micro-cell adds the diagonal of its element matrix to its ten DoFs, shared DoFs are summed over the macro-cells, then the
   // entries are inverted).  Here: the apply kernel with element matrices whose off-diagonal entries are zeroed, applied to the
   // function 1 -- every DoF receives exactly the sum of the elMat[k][k] of its adjacent micro-cells, through the same kernels and
   // the same additive exchange as apply() -- and a reciprocal on the host (set-up time, once per operator).
   void computeInverseDiagonalOperatorValues()
   {
      inverseDiagonalValues_.reset( new P2Function< double >( "inverse diagonal entries", storage_, minLevel_, maxLevel_ ) );
      P2Function< double > ones( "p2_diag_ones", storage_, minLevel_, maxLevel_ );
      for ( uint_t l = minLevel_; l <= maxLevel_; ++l )
      {
         ones.interpolate( 1.0, l, All );
         std::vector< const double* > diagTables;
         for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
            diagTables.push_back( storage_->uploadTable( diagonalTable( l, c ) ) );
         const P2Function< double >& d = *inverseDiagonalValues_;
         launchWith( diagTables, 1.0, ones, d, l, All, HYTEG_HIP_MASK_ALL, HYTEG_HIP_REPLACE );
         if ( storage_->getCells().size() > 1 )
         {
            d.getVertexDoFFunction().sumSharedCopies( l, All );
            d.sumSharedEdgeCopies( l, All );
         }
         const size_t          nv = (size_t) layout::cellSize( (int) l ), ne = std::max< size_t >( 1, d.getNumberOfEdgeDoFs( l ) );
         std::vector< double > hv( nv ), he( ne );
         for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         {
            d.getVertexDoFFunction().copyCellToHost( c, l, hv.data() );
            d.copyEdgeToHost( c, l, he.data() );
            hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "P2ElementwiseOperator: inverse diagonal" );
            for ( auto& v : hv )
               v = v != 0.0 ? 1.0 / v : 0.0;
            for ( auto& v : he )
               v = v != 0.0 ? 1.0 / v : 0.0;
            d.getVertexDoFFunction().copyCellFromHost( c, l, hv.data() );
            d.copyEdgeFromHost( c, l, he.data() );
         }
      }
   }
   std::shared_ptr< P2Function< double > > getInverseDiagonalValues() const
   {
      if ( !inverseDiagonalValues_ )
         throw std::runtime_error( "Inverse diagonal values have not been assembled, call computeInverseDiagonalOperatorValues() "
                                   "to set up this function." );
      return inverseDiagonalValues_;
   }

   // apply restricted to destination kinds (bit 0: vertex DoFs, 1..7: edge DoFs X .. XYZ) and point classes (`keep`); the other
   // entries of dst are not meaningful afterwards (use a temporary)
   void applyKinds( const P2Function< double >& src, const P2Function< double >& dst, uint_t level, DoFType flag, unsigned kinds,
                    unsigned keep = HYTEG_HIP_MASK_ALL ) const
   {
      if ( &src == &dst )
         throw std::runtime_error( "P2ElementwiseOperator::applyKinds: src and dst must differ" );
      launchWith( elementMatrices_.at( level ), 1.0, src, dst, level, flag, keep, HYTEG_HIP_REPLACE, kinds );
      if ( storage_->getCells().size() > 1 && ( keep & HYTEG_HIP_MASK_SHELL ) )
      {
         if ( kinds & 1u )
            dst.getVertexDoFFunction().sumSharedCopies( level, flag );
         if ( kinds & 0xFEu )
            dst.sumSharedEdgeCopies( level, flag );
      }
   }

   // P2ConstantOperator::smooth_sor / smooth_gs (src/constant_stencil_operator/P2ConstantOperator.cpp:113-153, macro-cell part
   // :913-1200).  On a macro-cell the reference sweeps the vertex DoFs in lexicographic order
   // (sor_3D_macrocell_P2_update_vertexdofs[_backwards]: vertex-to-vertex and edge-to-vertex stencils, edge DoFs as they are) and
   // then the edge DoFs type by type, X, Y, Z, XY, XZ, YZ, XYZ (sor_3D_macrocell_P2_update_edgedofs_by_type_*; backwards: the
   // reverse, edge types first) -- no element holds two edges of one type, so the DoFs of a type do not couple and a type's
   // sweep is order-free.  Here:
   //   * vertex DoFs:  rhs' = rhs_v - ( A u )_v + ( A_vv u_v )_v  moves the edge contributions to the right-hand side; then the
   //     P1 sweep of the vertex-to-vertex operator (P1ConstantOperator< P2VertexToVertexForm >::smooth_sor: exact lexicographic
   //     order inside the cells, the reference's vertex / edge / face order on shared primitives);
   //   * edge DoFs INSIDE the macro-cells, type T:  e_T += relax D^-1 ( rhs - A u )_T  with the apply restricted to that type --
   //     this IS the reference's macro-cell sweep;
   //   * edge DoFs on primitives shared between macro-cells (before the cell sweeps, as in the reference, which treats
   //     macro-edges and -faces before the cells): the edge type of such a DoF differs from neighbour cell to neighbour cell
   //     (every cell has its own local frame), so the colour of a shared DoF is its type in the lowest-numbered neighbour cell:
   //     per colour every cell applies the operator on its shell (all types), the shares are summed, every cell updates its
   //     shell DoFs of that type, and the copies are then overwritten with the owner's -- each shared DoF is updated exactly
   //     once per sweep.  DoFs of one colour with different owner cells can couple: there the sweep is a Jacobi step between
   //     them.  A Gauss-Seidel ordering of the same splitting, not the reference's iterates on multi-cell meshes.
   void smooth_sor( const P2Function< double >& dst, const P2Function< double >& rhs, double relax, uint_t level, DoFType flag,
                    bool backwards = false ) const
   {
      if ( &dst == &rhs )
         throw std::runtime_error( "P2ElementwiseOperator::smooth_sor: dst and rhs must differ" );
      if ( !v2v_ )
      {
         v2v_.reset( new P1ConstantOperator< forms::P2VertexToVertexForm< P2Form > >( storage_, minLevel_, maxLevel_ ) );
         sorTmp_.reset( new P2Function< double >( "p2_sor_tmp", storage_, minLevel_, maxLevel_ ) );
         sorTmpV_.reset( new P1Function< double >( "p2_sor_tmp_v", storage_, minLevel_, maxLevel_ ) );
      }
      const P2Function< double >& t  = *sorTmp_;
      const P1Function< double >& tv = t.getVertexDoFFunction();
      const P1Function< double >& sv = *sorTmpV_;
      const P1Function< double >& uv = dst.getVertexDoFFunction();
      const bool                  shared = storage_->getCells().size() > 1;
      bool                        anyShell = false;
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         anyShell = anyShell || ( storage_->maskFor( storage_->getLocalCell( c ), flag ) & HYTEG_HIP_MASK_SHELL );
      auto update = [&]( unsigned k, unsigned keep ) {
         t.assignKinds( { 1.0, -1.0 }, { rhs, t }, level, flag, k, keep );
         t.multElementwiseKinds( { *getInverseDiagonalValues(), t }, level, flag, k, keep );
         dst.addKinds( { relax }, { t }, level, flag, k, keep );
      };
      auto cellEdgeSweep = [&]( int type ) {
         applyKinds( dst, t, level, flag, 1u << type, HYTEG_HIP_MASK_INNER );
         update( 1u << type, HYTEG_HIP_MASK_INNER );
      };
      if ( level >= 2 && ( anyShell || storage_->numRanks() > 1 ) )
      {
         smoothSorReferenceOrder( dst, rhs, relax, level, flag, backwards, cellEdgeSweep );
         return;
      }
      // levels 0 and 1 (a DoF can lie on several macro-faces at once: no stencil per point class), or nothing shared is swept
      auto vertexSweep = [&]() {
         applyKinds( dst, t, level, flag, 1u );
         v2v_->apply( uv, sv, level, flag );
         tv.assign( { 1.0, -1.0, 1.0 }, { rhs.getVertexDoFFunction(), tv, sv }, level, flag );
         v2v_->smooth_sor( uv, tv, relax, level, flag, backwards );
      };
      auto sharedEdgeSweep = [&]( int type ) {
         applyKinds( dst, t, level, flag, 0xFEu, HYTEG_HIP_MASK_SHELL ); // every cell's share of every shared edge DoF, summed
         update( 1u << type, HYTEG_HIP_MASK_SHELL );
         dst.syncSharedEdgeCopies( level, flag ); // the lowest-numbered neighbour cell's copy (and its frame) decides
      };
      if ( !backwards )
      {
         vertexSweep();
         if ( shared )
            for ( int type = 1; type <= 7; ++type )
               sharedEdgeSweep( type );
         for ( int type = 1; type <= 7; ++type )
            cellEdgeSweep( type );
      }
      else
      {
         for ( int type = 7; type >= 1; --type )
            cellEdgeSweep( type );
         if ( shared )
            for ( int type = 7; type >= 1; --type )
               sharedEdgeSweep( type );
         vertexSweep();
      }
   }

 private:
   // tables of the sweeps over shared macro-primitives, per level (built on first use)
   struct SorTables
   {
      std::vector< const double* >             outside, closureVertex, closureEdge; // device tables per local cell
      std::vector< std::array< int, 12 > >     faceVerts;                            // per local cell: [face][3] local vertex ids by global id
      std::vector< std::array< double, 60 > >  faceW;                                // per local cell: [face][type][5] total weights
      const double*                            framesDev = nullptr;                  // levels <= 6: the records of hyteg_hip_p2_sor_face_frames, cell after cell
   };
   const SorTables& sorTables( uint_t level ) const
   {
      auto it = sorTables_.find( level );
      if ( it != sorTables_.end() )
         return it->second;
      SorTables                                T;
      std::vector< std::array< double, 15 > >  faceTot( storage_->getFaces().size(), std::array< double, 15 >{} );
      auto                                     faceVertsOf = []( const MacroCell& c, int f, int l[3] ) {
         for ( int r = 0; r < 3; ++r )
            l[r] = kCellFaceVerts[f][r];
         std::sort( l, l + 3, [&]( int a, int b ) { return c.v[a] < c.v[b]; } );
      };
      for ( const auto& c : storage_->getCells() ) // all cells of the mesh: a face's weights are the sum of its cells' shares
      {
         const std::vector< double > table = hostTableOf( c, level );
         for ( int f = 0; f < 4; ++f )
         {
            int    l[3];
            double w[15];
            faceVertsOf( c, f, l );
            hipCheck( hyteg_hip_p2_operator_table_face_edge_weights( table.data(), l, w ), "P2 smooth_sor: face weights" );
            for ( int k = 0; k < 15; ++k )
               faceTot[c.faces[f]][k] += w[k];
         }
         if ( c.localIndex < 0 )
            continue;
         std::vector< double > o( table.size() ), cv( table.size() ), ce( table.size() );
         hipCheck( hyteg_hip_p2_operator_table_closure_split( table.data(), o.data(), cv.data(), ce.data() ), "P2 smooth_sor: closure split" );
         if ( (size_t) c.localIndex >= T.outside.size() )
            T.outside.resize( c.localIndex + 1 ), T.closureVertex.resize( c.localIndex + 1 ), T.closureEdge.resize( c.localIndex + 1 ),
                T.faceVerts.resize( c.localIndex + 1 ), T.faceW.resize( c.localIndex + 1 );
         T.outside[c.localIndex]       = storage_->uploadTable( o );
         T.closureVertex[c.localIndex] = storage_->uploadTable( cv );
         T.closureEdge[c.localIndex]   = storage_->uploadTable( ce );
      }
      for ( const auto& c : storage_->getCells() )
      {
         if ( c.localIndex < 0 )
            continue;
         for ( int f = 0; f < 4; ++f )
         {
            int l[3];
            faceVertsOf( c, f, l );
            for ( int r = 0; r < 3; ++r )
               T.faceVerts[c.localIndex][3 * f + r] = l[r];
            for ( int k = 0; k < 15; ++k )
               T.faceW[c.localIndex][15 * f + k] = faceTot[c.faces[f]][k];
         }
      }
      if ( level <= 6 && !T.faceVerts.empty() )
      {
         const size_t          words = hyteg_hip_p2_sor_face_frames_bytes() / sizeof( double );
         std::vector< double > frames( words * T.faceVerts.size() );
         for ( size_t c = 0; c < T.faceVerts.size(); ++c )
            hipCheck( hyteg_hip_p2_sor_face_frames( (int) level, T.faceVerts[c].data(), T.faceW[c].data(), frames.data() + c * words ), "P2 smooth_sor: face frames" );
         T.framesDev = storage_->uploadTable( frames );
      }
      return sorTables_.emplace( level, std::move( T ) ).first->second;
   }
   // P2ConstantOperator::smooth_sor (P2ConstantOperator.cpp:1267-1330) on a mesh with shared macro-primitives, levels >= 2.
   // Forward: macro-vertices (vertex DoF), macro-edges (their vertex DoFs along the edge, then their edge DoFs), macro-faces (vertex
   // DoFs rows ascending, then edge DoFs rows ascending with X, XY, Y at every index), macro-cells (vertex DoFs, edge DoFs by type);
   // backwards the classes in reverse order and every loop reversed (a macro-edge still vertex DoFs first, P2MacroEdge.cpp:617-680).
   // A row of a DoF on primitive P splits into
   //    rest     sources outside the closure of P: ghost-layer values in the reference -- forward the values at the START of the
   //             sweep (nothing is communicated downwards in between), backwards those at the start of the class;
   //             ONE apply with the `outside` tables, summed over the cells by the additive exchange;
   //    closure  sources on P or its boundary, current values: edge-DoF sources of a vertex DoF and vertex-DoF sources of an edge
   //             DoF do not change while their counterpart is swept -- one apply with the closure tables per half-class;
   //             vertex-vertex couplings are swept by the P1 kernels of the vertex-to-vertex operator on every cell's copy
   //             (p1_sor_shell.hip), edge-edge couplings inside a macro-face by hyteg_hip_p2_sor_face_edgedofs_cell; the edge DoFs
   //             of one macro-edge do not couple.
   template < typename CellEdgeSweep >
   void smoothSorReferenceOrder( const P2Function< double >& dst, const P2Function< double >& rhs, double relax, uint_t level, DoFType flag,
                                 bool backwards, CellEdgeSweep&& cellEdgeSweep ) const
   {
      const SorTables& T = sorTables( level );
      if ( !sorRest_ )
         sorRest_.reset( new P2Function< double >( "p2_sor_rest", storage_, minLevel_, maxLevel_ ) );
      const P2Function< double >& t     = *sorTmp_;
      const P2Function< double >& rest  = *sorRest_;
      const P1Function< double >& tv    = t.getVertexDoFFunction();
      const P1Function< double >& restV = rest.getVertexDoFFunction();
      const P1Function< double >& sv    = *sorTmpV_;
      const P1Function< double >& uv    = dst.getVertexDoFFunction();
      const P1Function< double >& bv    = rhs.getVertexDoFFunction();
      const unsigned              V = 0xFu << 10, E = 0x3Fu, F = 0xFu << 6;
      auto computeRest = [&]( unsigned classes ) {
         launchWith( T.outside, 1.0, dst, rest, level, flag, classes, HYTEG_HIP_REPLACE );
         restV.sumSharedCopies( level, flag );
         rest.sumSharedEdgeCopies( level, flag );
      };
      auto vertexDoFs = [&]( unsigned classes ) {
         if ( classes == V )
         {
            v2v_->smooth_sor_shell_given_rest( uv, bv, restV, relax, level, flag, classes, backwards );
            return;
         }
         launchWith( T.closureEdge, 1.0, dst, t, level, flag, classes, HYTEG_HIP_REPLACE, 1u );
         tv.sumSharedCopies( level, flag );
         sv.assign( { 1.0, 1.0 }, { restV, tv }, level, flag );
         v2v_->smooth_sor_shell_given_rest( uv, bv, sv, relax, level, flag, classes, backwards );
      };
      // q = rhs - rest - (vertex-DoF sources on the closure), at the edge DoFs of the given classes
      auto edgeRightHandSide = [&]( unsigned classes ) {
         launchWith( T.closureVertex, 1.0, dst, t, level, flag, classes, HYTEG_HIP_REPLACE, 0xFEu );
         t.sumSharedEdgeCopies( level, flag );
         t.assignKinds( { 1.0, -1.0, -1.0 }, { rhs, rest, t }, level, flag, 0xFEu, classes );
      };
      auto edgeDoFsOfMacroEdges = [&]() {
         edgeRightHandSide( E );
         t.multElementwiseKinds( { *getInverseDiagonalValues(), t }, level, flag, 0xFEu, E );
         dst.assignKinds( { 1.0 - relax, relax }, { dst, t }, level, flag, 0xFEu, E );
      };
      auto edgeDoFsOfMacroFaces = [&]() {
         edgeRightHandSide( F );
         if ( T.framesDev && storage_->useBatch( level ) )
         {
            const auto   masks = storage_->masksFor( flag, false, F );
            const size_t bytes = hyteg_hip_p2_sor_face_frames_bytes();
            storage_->forCellChunks( [&]( int first, int count ) {
               std::vector< double* >       u;
               std::vector< const double* > q;
               for ( int c = first; c < first + count; ++c )
                  u.push_back( dst.getEdgeCellPointer( (uint_t) c, level ) ), q.push_back( t.getEdgeCellPointer( (uint_t) c, level ) );
               hipCheck( hyteg_hip_p2_sor_face_edgedofs_cells( count, u.data(), q.data(), (int) level,
                                                               reinterpret_cast< const char* >( T.framesDev ) + (size_t) first * bytes, relax,
                                                               masks.data() + first, backwards ? 1 : 0, storage_->stream() ),
                         "P2 smooth_sor: face edge DoFs (batched)" );
            } );
            return;
         }
         for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
            hipCheck( hyteg_hip_p2_sor_face_edgedofs_cell( dst.getEdgeCellPointer( c, level ), t.getEdgeCellPointer( c, level ), (int) level,
                                                           T.faceVerts[c].data(), T.faceW[c].data(), relax,
                                                           storage_->maskFor( storage_->getLocalCell( c ), flag ) & F, backwards ? 1 : 0,
                                                           storage_->stream() ),
                      "P2 smooth_sor: face edge DoFs" );
      };
      auto vertexDoFsOfCells = [&]() {
         applyKinds( dst, t, level, flag, 1u, HYTEG_HIP_MASK_INNER );
         v2v_->apply( uv, sv, level, flag );
         tv.assign( { 1.0, -1.0, 1.0 }, { bv, tv, sv }, level, flag );
         v2v_->smooth_sor_cells_only( uv, tv, relax, level, flag, backwards );
      };
      if ( !backwards )
      {
         computeRest( HYTEG_HIP_MASK_SHELL );
         vertexDoFs( V );
         vertexDoFs( E );
         edgeDoFsOfMacroEdges();
         vertexDoFs( F );
         edgeDoFsOfMacroFaces();
         vertexDoFsOfCells();
         for ( int type = 1; type <= 7; ++type )
            cellEdgeSweep( type );
      }
      else
      {
         for ( int type = 7; type >= 1; --type )
            cellEdgeSweep( type );
         vertexDoFsOfCells();
         computeRest( F );
         edgeDoFsOfMacroFaces();
         vertexDoFs( F );
         computeRest( E );
         vertexDoFs( E );
         edgeDoFsOfMacroEdges();
         computeRest( V );
         vertexDoFs( V );
      }
   }

 public:
   void smooth_gs( const P2Function< double >& dst, const P2Function< double >& rhs, uint_t level, DoFType flag ) const
   {
      smooth_sor( dst, rhs, 1.0, level, flag );
   }

   // P2ElementwiseOperator::smooth_jac (P2ElementwiseOperator.cpp:344-374), statement by statement:
   //   dst = A src;  dst = rhs - dst;  dst = D^-1 dst;  dst = src + relax dst
   void smooth_jac( const P2Function< double >& dst, const P2Function< double >& rhs, const P2Function< double >& src, double relax,
                    uint_t level, DoFType flag ) const
   {
      if ( &dst == &src )
         throw std::runtime_error( "P2ElementwiseOperator::smooth_jac: src and dst must differ" );
      apply( src, dst, level, flag );
      dst.assign( { 1.0, -1.0 }, { rhs, dst }, level, flag );
      dst.multElementwise( { *getInverseDiagonalValues(), dst }, level, flag );
      dst.assign( { 1.0, relax }, { src, dst }, level, flag );
   }

   std::shared_ptr< PrimitiveStorage > getStorage() const { return storage_; }
   const std::vector< double >&        getElementMatrices( uint_t level, uint_t localCell = 0 ) const { return hostMatrices_.at( level ).at( localCell ); }

   // Operator::apply = gemv( 1, src, updateType == Replace ? 0 : 1, dst ), P2ElementwiseOperator.hpp:60-75
   void apply( const P2Function< double >& src, const P2Function< double >& dst, uint_t level, DoFType flag, UpdateType updateType = Replace ) const
   {
      gemv( 1.0, src, updateType == Replace ? 0.0 : 1.0, dst, level, flag );
   }
   // every cell adds the contributions of its own micro-cells; on DoFs shared by several cells these are partial sums that the
   // additive exchange completes (communicateAdditively< Cell, ... > at the end of the reference's gemv, :225-235)
   void gemv( double alpha, const P2Function< double >& src, double beta, const P2Function< double >& dst, uint_t level, DoFType flag ) const
   {
      if ( &src == &dst )
         throw std::runtime_error( "P2ElementwiseOperator::gemv: src and dst must differ" );
      const bool shared = storage_->getCells().size() > 1; // also a rank with one cell shares DoFs with cells of other ranks
      if ( !shared )
      {
         if ( beta != 0.0 && beta != 1.0 )
            dst.assign( { beta }, { dst }, level, flag );
         launch( alpha, src, dst, level, flag, HYTEG_HIP_MASK_ALL, beta == 0.0 ? HYTEG_HIP_REPLACE : HYTEG_HIP_ADD );
         return;
      }
      if ( beta == 0.0 )
      {
         launch( alpha, src, dst, level, flag, HYTEG_HIP_MASK_ALL, HYTEG_HIP_REPLACE );
         dst.getVertexDoFFunction().sumSharedCopies( level, flag );
         dst.sumSharedEdgeCopies( level, flag );
         return;
      }
      // beta != 0: the summed shares of the shared DoFs are formed in a temporary and then added
      P2Function< double > tmp( "p2_gemv_tmp", storage_, level, level );
      launch( alpha, src, tmp, level, flag, HYTEG_HIP_MASK_ALL, HYTEG_HIP_REPLACE );
      tmp.getVertexDoFFunction().sumSharedCopies( level, flag );
      tmp.sumSharedEdgeCopies( level, flag );
      dst.assign( { beta, 1.0 }, { dst, tmp }, level, flag );
   }

 protected:
   void launch( double alpha, const P2Function< double >& src, const P2Function< double >& dst, uint_t level, DoFType flag, unsigned keep,
                int update ) const
   {
      launchWith( elementMatrices_.at( level ), alpha, src, dst, level, flag, keep, update );
   }
   void launchWith( const std::vector< const double* >& tables, double alpha, const P2Function< double >& src, const P2Function< double >& dst,
                    uint_t level, DoFType flag, unsigned keep, int update, unsigned kinds = 0xFFu ) const
   {
      std::vector< double* >       dv, de;
      std::vector< const double* > sv, se;
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
      {
         dv.push_back( dst.getVertexDoFFunction().getCellPointer( c, level ) ), de.push_back( dst.getEdgeCellPointer( c, level ) );
         sv.push_back( src.getVertexDoFFunction().getCellPointer( c, level ) ), se.push_back( src.getEdgeCellPointer( c, level ) );
      }
      launchPointers( tables, alpha, sv, se, dv, de, level, storage_->masksFor( flag, false, keep ), update, kinds );
   }
   // the kernel launches of one operator application over the local cells, given the cells' arrays (mixed operators pass a P1
   // function's arrays as vertex part)
   void launchPointers( const std::vector< const double* >& tables, double alpha, const std::vector< const double* >& sv,
                        const std::vector< const double* >& se, const std::vector< double* >& dv, const std::vector< double* >& de, uint_t level,
                        const std::vector< unsigned >& masks, int update, unsigned kinds ) const
   {
      const uint_t nLocal = storage_->getNumberOfLocalCells();
      if ( storage_->useBatch( level ) && level >= 2 && level <= 6 )
      {
         // small levels, several cells: all local cells in two launches per chunk (inner DoFs, boundary DoFs) instead of two or three per
         // cell -- a Taylor-Hood cycle on 24 cells issued 62,000 apply launches of 3-5 us each (round 3)
         storage_->forCellChunks( [&]( int first, int count ) {
            hipCheck( hyteg_hip_p2_elementwise_apply_cells_kinds( count, dv.data() + first, de.data() + first, sv.data() + first, se.data() + first, (int) level,
                                                                  tables.data() + first, alpha, update, masks.data() + first, kinds, storage_->stream() ),
                      "P2ElementwiseOperator::gemv (batched)" );
         } );
         return;
      }
      // the cells of a rank are independent launches: alternating between the storage's stream and a side stream lets a launch start
      // while the last waves of the previous one drain -- level 7, 24 cells: 1110 -> 1033 us per apply, 6 cells: no difference
      // (profiles/r03_cell_streams.txt); HYTEG_AMD_CELL_STREAMS=1 keeps one stream
      static const bool oneStream = [] {
         const char* e = std::getenv( "HYTEG_AMD_CELL_STREAMS" );
         return e && e[0] == '1';
      }();
      PrimitiveStorage::SideChain chain( *storage_, !oneStream && nLocal >= 8 && level >= 5 );
      for ( uint_t c = 0; c < nLocal; ++c )
      {
         ( c & 1u ) ? chain.toSide() : chain.toMain();
         hipCheck( hyteg_hip_p2_elementwise_apply_cell_kinds( dv[c], de[c], sv[c], se[c], (int) level, tables.at( c ), alpha, update, masks[c], kinds,
                                                              storage_->stream() ),
                   "P2ElementwiseOperator::gemv" );
      }
      chain.join();
   }
   std::shared_ptr< PrimitiveStorage >                          storage_;
   uint_t                                                       minLevel_, maxLevel_;
   std::map< uint_t, std::vector< const double* > >             elementMatrices_;
   std::map< uint_t, std::vector< std::vector< double > > >     hostMatrices_;
   std::shared_ptr< P2Function< double > >                      inverseDiagonalValues_;
   mutable std::unique_ptr< P1ConstantOperator< forms::P2VertexToVertexForm< P2Form > > > v2v_; // smooth_sor
   mutable std::unique_ptr< P2Function< double > >                                        sorTmp_;
   mutable std::unique_ptr< P1Function< double > >                                        sorTmpV_;
   mutable std::unique_ptr< P2Function< double > >                                        sorRest_;
   mutable std::map< uint_t, SorTables >                                                  sorTables_;
   uint64_t                                                     uid_ = nextUid();
};
using P2ElementwiseLaplaceOperator = P2ElementwiseOperator< forms::P2LaplaceForm >; // P2ElementwiseOperator.hpp:454

// P2ConstantOperator< P2Form > (src/constant_stencil_operator/P2ConstantOperator.hpp; apply = the four sub-operators
// VertexToVertex, EdgeToVertex, VertexToEdge, EdgeToEdge, P2ConstantOperator.cpp:100-112, each a constant stencil per
// macro-primitive).  This class ASSEMBLES those stencils itself, the reference's way -- P2Elements3D::calculate*StencilInMacroCell
// (p2elements.hpp), for an inner DoF of every kind and for a DoF of each of the 14 boundary point classes (there the maps hold
// this cell's share, as the reference's functions do "for indices on the boundary of a macro-cell") -- and hands the VALUES,
// flattened in the iteration order of the reference's std::map types, to the C-ABI's kernel seam
// (hyteg_hip_p2_build_operator_table_from_stencils; the layout of the keys comes from hyteg_hip_p2_constant_stencil_layout and
// is checked against the assembled maps).  From level 2 on no element matrix reaches the kernels.  apply / gemv / smoothers
// are the base class's: all four sub-operators in one pass over the DoFs.
template < class P2Form >
class P2ConstantOperator : public P2ElementwiseOperator< P2Form >
{
   using Base = P2ElementwiseOperator< P2Form >;

 public:
   P2ConstantOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : Base( storage, minLevel, maxLevel, typename Base::NoTables{} )
   {
      int counts[4];
      hipCheck( hyteg_hip_p2_constant_stencil_layout( counts, nullptr ), "P2ConstantOperator: layout" );
      total_ = counts[0] + counts[1] + counts[2] + counts[3];
      keys_.resize( 5 * (size_t) total_ );
      hipCheck( hyteg_hip_p2_constant_stencil_layout( counts, keys_.data() ), "P2ConstantOperator: layout" );
      for ( uint_t l = minLevel; l <= maxLevel; ++l )
         for ( uint_t lc = 0; lc < storage->getNumberOfLocalCells(); ++lc )
         {
            if ( l < 2 )
            {
               // levels 0 and 1: a DoF can lie on several macro-faces at once, there is no stencil per point class; the kernels
               // gather micro-cell by micro-cell from the element matrices there (a handful of DoFs)
               this->addElementMatrixTable( l, lc );
               stencils_[l].push_back( CellStencils{} );
               continue;
            }
            CellStencils          S = assembleStencils( storage->getLocalCell( lc ), l );
            std::vector< double > table( hyteg_hip_p2_operator_table_size() );
            hipCheck( hyteg_hip_p2_build_operator_table_from_stencils( S.inner.data(), S.classes.data(), table.data() ),
                      "P2ConstantOperator: operator table" );
            this->elementMatrices_[l].push_back( storage->uploadTable( table ) );
            stencils_[l].push_back( std::move( S ) );
         }
   }
   // the flattened inner stencils of a local cell (v2v | e2v | v2e | e2e in the reference's map order) and their keys
   const std::vector< double >& getInnerStencils( uint_t level, uint_t localCell = 0 ) const { return stencils_.at( level ).at( localCell ).inner; }
   const std::vector< int >&    getStencilKeys() const { return keys_; }

 protected:
   std::vector< double > hostTableOf( const MacroCell& cell, uint_t l ) const override
   {
      if ( l < 2 )
         return Base::hostTableOf( cell, l );
      const CellStencils    S = assembleStencils( cell, l );
      std::vector< double > table( hyteg_hip_p2_operator_table_size() );
      hipCheck( hyteg_hip_p2_build_operator_table_from_stencils( S.inner.data(), S.classes.data(), table.data() ), "P2ConstantOperator: operator table" );
      return table;
   }
   // diagonal: the weights with source kind = destination kind and offset 0
   std::vector< double > diagonalTable( uint_t level, uint_t localCell ) const override
   {
      if ( level < 2 )
         return Base::diagonalTable( level, localCell );
      const auto&           S = stencils_.at( level ).at( localCell );
      std::vector< double > inner( (size_t) total_, 0.0 ), classes( 14 * (size_t) total_, 0.0 );
      for ( int i = 0; i < total_; ++i )
      {
         const int* k = &keys_[5 * (size_t) i];
         if ( k[0] != k[1] || k[2] != 0 || k[3] != 0 || k[4] != 0 )
            continue;
         inner[i] = S.inner[i];
         for ( int cls = 0; cls < 14; ++cls )
            classes[(size_t) cls * total_ + i] = S.classes[(size_t) cls * total_ + i];
      }
      std::vector< double > table( hyteg_hip_p2_operator_table_size() );
      hipCheck( hyteg_hip_p2_build_operator_table_from_stencils( inner.data(), classes.data(), table.data() ), "P2ConstantOperator: diagonal table" );
      return table;
   }

 private:
   struct CellStencils
   {
      std::vector< double > inner, classes;
   };
   // inner DoFs: the stencil maps must have exactly the keys of the layout; boundary classes: a subset of them
   CellStencils assembleStencils( const MacroCell& cell, uint_t l ) const
   {
      CellStencils S;
      S.inner = flatten( P2Elements::P2Elements3D::assembleAtClass< P2Form >( cell, l, 14 ), true );
      S.classes.assign( 14 * (size_t) total_, 0.0 );
      for ( int cls = 0; cls < 14; ++cls )
      {
         const auto v = flatten( P2Elements::P2Elements3D::assembleAtClass< P2Form >( cell, l, cls ), false );
         std::copy( v.begin(), v.end(), S.classes.begin() + (size_t) cls * total_ );
      }
      return S;
   }
   // values of the assembled maps in the order of the layout's keys; `complete`: every key of the layout must be present
   std::vector< double > flatten( const P2Elements::P2Elements3D::KindStencils& maps, bool complete ) const
   {
      std::vector< double > v( (size_t) total_, 0.0 );
      size_t                found = 0;
      for ( int i = 0; i < total_; ++i )
      {
         const int* k  = &keys_[5 * (size_t) i];
         auto       it = maps[k[0]].find( P2Elements::P2Elements3D::Key{ k[1], { k[2], k[3], k[4] } } );
         if ( it != maps[k[0]].end() )
            v[i] = it->second, ++found;
         else if ( complete )
            throw std::runtime_error( "P2ConstantOperator: the assembled stencil lacks a key of hyteg_hip_p2_constant_stencil_layout" );
      }
      size_t have = 0;
      for ( const auto& m : maps )
         have += m.size();
      if ( found != have )
         throw std::runtime_error( "P2ConstantOperator: the assembled stencil has keys hyteg_hip_p2_constant_stencil_layout does not list" );
      return v;
   }
   int                                            total_ = 0;
   std::vector< int >                             keys_;
   std::map< uint_t, std::vector< CellStencils > > stencils_;
};
using P2ConstantLaplaceOperator = P2ConstantOperator< forms::P2LaplaceForm >; // P2ConstantOperator.hpp

} // namespace hyteg
