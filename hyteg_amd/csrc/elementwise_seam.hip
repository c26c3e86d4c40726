// C-ABI entry points at the seam of the generated elementwise operators (module hyteg_operators; the shape of the seam is
// visible in the vendored sample apps/2023-zikeli-mt/MT-apps/operators-used/P1ElementwiseDiffusion_cubes_const_float64.{hpp,cpp}):
//   apply_macro_3D( dst*, src*, macro_vertex_coord_id_{0..3}comp{0..2}, int64 micro_edges_per_macro_edge, float same )
// adds, for every micro-cell of the macro-cell, elMat( micro-cell type ) * ( the cell's source values ) to the cell's
// destination values -- at ALL points of the cell array, the points on the macro-cell's boundary included (they receive this
// cell's share; the caller zeroes them before and sums the shares of the neighbour cells afterwards,
// P1ElementwiseDiffusion_cubes_const_float64.cpp:76-165).
//
// Here: the element matrices are constant per micro-cell type on an affine macro-cell, so the scatter equals one constant
// stencil per point class; the host part of the call sums them from the coordinates (element_matrices.hpp), the device part is
// the z-march kernel in Add mode for the inner points plus the boundary-share kernel for the shell.  No new device code.
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "element_matrices.hpp"
#include "shell.hpp"

using namespace hyteg_hip;

namespace {

// micro_edges_per_macro_edge = 2^level
int level_of( int64_t micro_edges, int* level )
{
   if ( micro_edges < 1 || ( micro_edges & ( micro_edges - 1 ) ) != 0 )
      return HYTEG_HIP_EINVAL;
   int l = 0;
   while ( ( (int64_t) 1 << l ) < micro_edges )
      ++l;
   *level = l;
   return HYTEG_HIP_OK;
}

bool degenerate( const double cc[4][3] )
{
   double g[4][3];
   const double V = elmat::gradients( cc, g );
   return !( V > 0.0 ) || !std::isfinite( V );
}

// diag[i] += c[ class of point i ] (14 shell classes, then the inner points); set-up work, one thread per array entry
struct ClassConstants
{
   double c[15];
};
__global__ __launch_bounds__( 256 ) void p1_add_class_constants_kernel( double* diag, const Tile* tiles, int ntiles, int N, const ClassConstants C )
{
   const int t = blockIdx.x;
   if ( t >= ntiles )
      return;
   const Tile tl = tiles[t];
   const int  W = N - tl.z, s0 = slice_start( N, tl.z );
   for ( int e = threadIdx.x; e < tl.cnt; e += 256 )
   {
      const int i = tl.a + e, j = i - s0;
      const int y = row_of( W, j ), x = j - row_start( W, y );
      const int slot = shell::shell_slot( N, x, y, tl.z );
      diag[i] += C.c[slot < 0 ? 14 : slot];
   }
}

template < typename C >
void unpack_coords( const C* coords, double cc[4][3] )
{
   for ( int v = 0; v < 4; ++v )
      for ( int r = 0; r < 3; ++r )
         cc[v][r] = (double) coords[3 * v + r];
}

} // namespace

extern "C" {

HYTEG_HIP_API int hyteg_hip_p1_elementwise_diffusion_stencils( const double* macro_vertex_coords,
                                                               int64_t       micro_edges_per_macro_edge,
                                                               double*       w_inner,
                                                               double*       w_slots )
{
   HH_REQUIRE( macro_vertex_coords && w_inner && w_slots, "p1_elementwise_diffusion_stencils: null pointer" );
   int level = 0;
   HH_REQUIRE( level_of( micro_edges_per_macro_edge, &level ) == HYTEG_HIP_OK && level <= HYTEG_HIP_MAX_LEVEL,
               "p1_elementwise_diffusion_stencils: micro_edges_per_macro_edge must be a power of two, at most 2^11" );
   double cc[4][3];
   unpack_coords( macro_vertex_coords, cc );
   HH_REQUIRE( !degenerate( cc ), "p1_elementwise_diffusion_stencils: degenerate macro-cell" );
   const elmat::P1Stencils S = elmat::p1_diffusion_stencils( cc, micro_edges_per_macro_edge );
   for ( int k = 0; k < 15; ++k )
      w_inner[k] = S.inner[k];
   for ( int s = 0; s < 14; ++s )
      for ( int k = 0; k < 15; ++k )
         w_slots[15 * s + k] = S.slots[s][k];
   return HYTEG_HIP_OK;
}

// the kernel restricted to the point classes of `mask` (bit k < 14: points on macro-edge / -face / -vertex slot k, bit 14: inner
// points), Replace or Add: what the host layer's operator needs, because in the cell-centric storage the boundary entries of a
// cell array are the DoFs themselves (not a halo that may be zeroed) and points outside the flag must stay untouched
HYTEG_HIP_API int hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_masked( double*            dst,
                                                                            const double*      src,
                                                                            const double*      macro_vertex_coords,
                                                                            int64_t            micro_edges_per_macro_edge,
                                                                            unsigned           mask,
                                                                            int                update,
                                                                            hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && src && macro_vertex_coords, "p1_elementwise_diffusion_apply_macro_3d: null pointer" );
   HH_REQUIRE( dst != src, "p1_elementwise_diffusion_apply_macro_3d: src and dst must not alias" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p1_elementwise_diffusion_apply_macro_3d: bad update type" );
   double w_inner[15], w_slots[14 * 15];
   int    rc = hyteg_hip_p1_elementwise_diffusion_stencils( macro_vertex_coords, micro_edges_per_macro_edge, w_inner, w_slots );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   int level = 0;
   level_of( micro_edges_per_macro_edge, &level );
   // points on the macro-cell's faces / edges / vertices: this cell's share; inner points (levels >= 2): the full stencil
   if ( mask & HYTEG_HIP_MASK_SHELL )
   {
      rc = hyteg_hip_p1_apply_cell_boundary( dst, src, level, w_slots, mask & HYTEG_HIP_MASK_SHELL, update, stream );
      if ( rc != HYTEG_HIP_OK )
         return rc;
   }
   if ( ( mask & HYTEG_HIP_MASK_INNER ) && level >= HYTEG_HIP_MIN_LEVEL )
      return hyteg_hip_p1_apply_cell( dst, src, level, w_inner, update, stream );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_elementwise_diffusion_apply_macro_3d( double*            dst,
                                                                     const double*      src,
                                                                     const double*      macro_vertex_coords,
                                                                     int64_t            micro_edges_per_macro_edge,
                                                                     double             micro_edges_per_macro_edge_float,
                                                                     hyteg_hip_stream_t stream )
{
   HH_REQUIRE( micro_edges_per_macro_edge_float == (double) micro_edges_per_macro_edge,
               "p1_elementwise_diffusion_apply_macro_3d: the two micro_edges_per_macro_edge arguments differ" );
   return hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_masked( dst, src, macro_vertex_coords, micro_edges_per_macro_edge, HYTEG_HIP_MASK_ALL,
                                                                     HYTEG_HIP_ADD, stream );
}

HYTEG_HIP_API int hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_f32( float*             dst,
                                                                         const float*       src,
                                                                         const float*       macro_vertex_coords,
                                                                         int64_t            micro_edges_per_macro_edge,
                                                                         float              micro_edges_per_macro_edge_float,
                                                                         hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && src && macro_vertex_coords, "p1_elementwise_diffusion_apply_macro_3d_f32: null pointer" );
   HH_REQUIRE( dst != src, "p1_elementwise_diffusion_apply_macro_3d_f32: src and dst must not alias" );
   HH_REQUIRE( micro_edges_per_macro_edge_float == (float) micro_edges_per_macro_edge,
               "p1_elementwise_diffusion_apply_macro_3d_f32: the two micro_edges_per_macro_edge arguments differ" );
   int level = 0;
   HH_REQUIRE( level_of( micro_edges_per_macro_edge, &level ) == HYTEG_HIP_OK && level <= 10,
               "p1_elementwise_diffusion_apply_macro_3d_f32: micro_edges_per_macro_edge must be a power of two, at most 2^10" );
   double cd[12];
   for ( int k = 0; k < 12; ++k )
      cd[k] = (double) macro_vertex_coords[k];
   double w_inner[15], w_slots[14 * 15];
   int    rc = hyteg_hip_p1_elementwise_diffusion_stencils( cd, micro_edges_per_macro_edge, w_inner, w_slots );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   rc = hyteg_hip_p1_apply_cell_boundary_f32( dst, src, level, w_slots, HYTEG_HIP_MASK_SHELL, HYTEG_HIP_ADD, stream );
   if ( rc != HYTEG_HIP_OK || level < HYTEG_HIP_MIN_LEVEL )
      return rc;
   return hyteg_hip_p1_apply_cell_f32( dst, src, level, w_inner, HYTEG_HIP_ADD, stream );
}

// computeInverseDiagonalOperatorValues_macro_3D of the generated operators: diag += the diagonal entries of the element
// matrices of the adjacent micro-cells, at all points of the cell array (the caller sums the shares of neighbour cells and
// inverts, P1ElementwiseDiffusion_cubes_const_float64.cpp: computeInverseDiagonalOperatorValues)
HYTEG_HIP_API int hyteg_hip_p1_elementwise_diffusion_diagonal_macro_3d( double*            diag,
                                                                        const double*      macro_vertex_coords,
                                                                        int64_t            micro_edges_per_macro_edge,
                                                                        double             micro_edges_per_macro_edge_float,
                                                                        hyteg_hip_stream_t stream )
{
   HH_REQUIRE( diag && macro_vertex_coords, "p1_elementwise_diffusion_diagonal_macro_3d: null pointer" );
   HH_REQUIRE( micro_edges_per_macro_edge_float == (double) micro_edges_per_macro_edge,
               "p1_elementwise_diffusion_diagonal_macro_3d: the two micro_edges_per_macro_edge arguments differ" );
   double w_inner[15], w_slots[14 * 15];
   int    rc = hyteg_hip_p1_elementwise_diffusion_stencils( macro_vertex_coords, micro_edges_per_macro_edge, w_inner, w_slots );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   int level = 0;
   level_of( micro_edges_per_macro_edge, &level );
   HH_REQUIRE( level <= 10, "p1_elementwise_diffusion_diagonal_macro_3d: levels 0..10" );
   ClassConstants C;
   for ( int s = 0; s < 14; ++s )
      C.c[s] = w_slots[15 * s + 7];
   C.c[14] = w_inner[7];
   TileTable tt;
   rc = get_tiles( level, TILES_FULL, 1024, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   if ( tt.count > 0 )
      hipLaunchKernelGGL( p1_add_class_constants_kernel, dim3( tt.count ), dim3( 256 ), 0, as_stream( stream ), diag, tt.dev, tt.count,
                          ( 1 << level ) + 1, C );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

// ---- P2: the same seam for the generated P2ElementwiseDiffusion operator (module hyteg_operators; source absent from the
// snapshot, call sites src/hyteg_operators_composites/viscousblock/P2ViscousBlockLaplaceOperator.hpp:29,66).  Its
// apply_macro_3D takes the vertex- and edge-DoF arrays of dst and src and the same geometry arguments; the ORDER of the four
// array arguments in the generated code could not be read (parity unpinned for the argument order): vertex before edge here,
// as everywhere in this header.  dst += A_cell src on ALL DoFs of the macro-cell (boundary DoFs receive this cell's share).
// The element matrices of the six micro-cell types are computed from the coordinates and turned into the kernel's operator
// table once per (coordinates, micro_edges_per_macro_edge); the tables stay cached on the device.
HYTEG_HIP_API int hyteg_hip_p2_elementwise_diffusion_element_matrices( const double* macro_vertex_coords,
                                                                       int64_t       micro_edges_per_macro_edge,
                                                                       double*       elmat /* 600 */ )
{
   HH_REQUIRE( macro_vertex_coords && elmat, "p2_elementwise_diffusion_element_matrices: null pointer" );
   int level = 0;
   HH_REQUIRE( level_of( micro_edges_per_macro_edge, &level ) == HYTEG_HIP_OK, "p2_elementwise_diffusion_element_matrices: micro_edges_per_macro_edge must be a power of two" );
   double cc[4][3];
   unpack_coords( macro_vertex_coords, cc );
   HH_REQUIRE( !degenerate( cc ), "p2_elementwise_diffusion_element_matrices: degenerate macro-cell" );
   for ( int t = 0; t < 6; ++t )
   {
      double c[4][3];
      elmat::micro_cell_coords( cc, micro_edges_per_macro_edge, t, c );
      elmat::p2_diffusion( c, elmat + 100 * t );
   }
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2_elementwise_diffusion_apply_macro_3d( double*            dst_vertex,
                                                                     double*            dst_edge,
                                                                     const double*      src_vertex,
                                                                     const double*      src_edge,
                                                                     const double*      macro_vertex_coords,
                                                                     int64_t            micro_edges_per_macro_edge,
                                                                     double             micro_edges_per_macro_edge_float,
                                                                     hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst_vertex && dst_edge && src_vertex && src_edge && macro_vertex_coords, "p2_elementwise_diffusion_apply_macro_3d: null pointer" );
   HH_REQUIRE( micro_edges_per_macro_edge_float == (double) micro_edges_per_macro_edge,
               "p2_elementwise_diffusion_apply_macro_3d: the two micro_edges_per_macro_edge arguments differ" );
   int level = 0;
   HH_REQUIRE( level_of( micro_edges_per_macro_edge, &level ) == HYTEG_HIP_OK && level <= HYTEG_HIP_P2_MAX_LEVEL,
               "p2_elementwise_diffusion_apply_macro_3d: micro_edges_per_macro_edge must be a power of two, at most 2^9" );
   // operator table of this (cell, level), built and uploaded on first use
   static std::mutex                                            mtx;
   static std::map< std::pair< int, std::vector< double > >, double* > cache;
   int dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   std::vector< double > key( macro_vertex_coords, macro_vertex_coords + 12 );
   key.push_back( (double) micro_edges_per_macro_edge );
   const double* table_dev = nullptr;
   {
      std::lock_guard< std::mutex > lock( mtx );
      auto it = cache.find( { dev, key } );
      if ( it == cache.end() )
      {
         double elm[600];
         int    rc = hyteg_hip_p2_elementwise_diffusion_element_matrices( macro_vertex_coords, micro_edges_per_macro_edge, elm );
         if ( rc != HYTEG_HIP_OK )
            return rc;
         std::vector< double > table( hyteg_hip_p2_operator_table_size() );
         rc = hyteg_hip_p2_build_operator_table( elm, table.data() );
         if ( rc != HYTEG_HIP_OK )
            return rc;
         void* p = nullptr;
         HH_CHECK_HIP( hipMalloc( &p, table.size() * sizeof( double ) ) );
         HH_CHECK_HIP( hipMemcpy( p, table.data(), table.size() * sizeof( double ), hipMemcpyHostToDevice ) );
         it = cache.emplace( std::make_pair( dev, key ), static_cast< double* >( p ) ).first;
      }
      table_dev = it->second;
   }
   return hyteg_hip_p2_elementwise_apply_cell( dst_vertex, dst_edge, src_vertex, src_edge, level, table_dev, 1.0, HYTEG_HIP_ADD, HYTEG_HIP_MASK_ALL, stream );
}
}
