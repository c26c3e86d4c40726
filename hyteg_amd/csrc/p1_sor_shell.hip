// SOR / Gauss-Seidel on the macro-vertices, macro-edges and macro-faces of one macro-cell, in the cell's own array.
// Reference: P1Operator::smooth_sor_macro_vertices / _edges / _faces (src/hyteg/p1functionspace/P1Operator.hpp:908-1007),
// smooth_sor_edge (:1352-1421), smooth_sor_face3D (:1424-1503).
//
// The reference sweeps every macro-primitive in its own memory: new values inside the primitive (and on its
// lower-dimensional boundary), ghost-layer values from the last communication for everything else.  Here the
// "everything else" part of the stencil sum arrives pre-computed in `rest` (summed over all neighbour cells by the
// additive exchange), and the sweep itself runs redundantly on every cell's copy of the primitive with the TOTAL
// weights, so all copies end up bit-identical without a further exchange.
//   vertices : pointwise
//   edges    : first-order linear recurrence along the edge (one wave per edge, 64 points per round)
//   faces    : lexicographic 2-D sweep as hyperplane wavefront; one workgroup per face, one thread per row, the three
//              already-updated neighbours travel through LDS / a register, everything else is folded into `a` by a
//              fully parallel preparation kernel
#include "common.hpp"

using namespace hyteg_hip;

namespace {

// per-cell descriptors; same layout as hyteg_hip_sor_shell_tables (include/hyteg_hip.h)
struct SorShellDesc
{
   int    edgeV[6][2];
   int    faceV[4][3];
   double edgeW[6][3];
   double faceW[4][7];
   double vertexW[4];
};
static_assert( sizeof( SorShellDesc ) == sizeof( hyteg_hip_sor_shell_tables ), "descriptor layout" );

// one launch covers `ncells` cells (the cell index is a grid dimension); the single-cell entry point passes its
// descriptor by value in `one`, the batched one a device table
struct SorShellArgs
{
   double*             dst[HYTEG_HIP_MAX_BATCH];
   const double*       rhs[HYTEG_HIP_MAX_BATCH];
   double*             rest[HYTEG_HIP_MAX_BATCH];
   unsigned            mask[HYTEG_HIP_MAX_BATCH];
   const SorShellDesc* table;
   SorShellDesc        one;
   int                 N;
   int                 backwards;
   double              relax;
};
// view of one cell
struct SorShellCell
{
   double*             dst;
   const double*       rhs;
   double*             rest;
   unsigned            mask;
   const SorShellDesc* S;
   int                 N, backwards;
   double              relax;
   __device__ inline SorShellCell( const SorShellArgs& A, int cell )
   : dst( A.dst[cell] ), rhs( A.rhs[cell] ), rest( A.rest[cell] ), mask( A.mask[cell] ), S( A.table ? A.table + cell : &A.one ), N( A.N ),
     backwards( A.backwards ), relax( A.relax )
   {}
};

// index-space position of cell-local vertex k scaled by n (n = N-1 gives the vertex, n = 1 its unit vector)
__device__ inline void vtx( int k, int n, int& x, int& y, int& z )
{
   x = k == 1 ? n : 0;
   y = k == 2 ? n : 0;
   z = k == 3 ? n : 0;
}

__global__ __launch_bounds__( 64 ) void p1_sor_vertices_kernel( const SorShellArgs AA )
{
   const SorShellCell A( AA, blockIdx.x );
   const int          k = threadIdx.x;
   if ( k >= 4 || !( ( A.mask >> ( 10 + k ) ) & 1u ) )
      return;
   int x, y, z;
   vtx( k, A.N - 1, x, y, z );
   const int i = cell_index( A.N, x, y, z );
   A.dst[i]    = ( 1.0 - A.relax ) * A.dst[i] + A.relax * ( A.rhs[i] - A.rest[i] ) / A.S->vertexW[k];
}

__global__ __launch_bounds__( 64 ) void p1_sor_edges_kernel( const SorShellArgs AA )
{
   const SorShellCell A( AA, blockIdx.y );
   const int          e = blockIdx.x;
   if ( !( ( A.mask >> e ) & 1u ) )
      return;
   const int N = A.N, n = N - 1;
   int       first = A.S->edgeV[e][0], last = A.S->edgeV[e][1];
   double    wPrev = A.S->edgeW[e][1], wNext = A.S->edgeW[e][2];
   if ( A.backwards )
   {
      const int    t = first;
      const double w = wPrev;
      first = last, last = t;
      wPrev = wNext, wNext = w;
   }
   int ox, oy, oz, fx, fy, fz, lx, ly, lz;
   vtx( first, n, ox, oy, oz );
   vtx( first, 1, fx, fy, fz );
   vtx( last, 1, lx, ly, lz );
   const int    dx = lx - fx, dy = ly - fy, dz = lz - fz;
   const double sc = A.relax / A.S->edgeW[e][0];
   const double b  = -sc * wPrev;

   __shared__ double sa[64], sr[64];
   double            carry = A.dst[cell_index( N, ox, oy, oz )];
   const int         cnt   = n - 1; // interior points s = 1 .. n-1, s counted from `first`
   for ( int base = 0; base < cnt; base += 64 )
   {
      const int s   = base + (int) threadIdx.x + 1;
      int       idx = 0;
      if ( s <= cnt )
      {
         idx             = cell_index( N, ox + s * dx, oy + s * dy, oz + s * dz );
         const int nxt   = cell_index( N, ox + ( s + 1 ) * dx, oy + ( s + 1 ) * dy, oz + ( s + 1 ) * dz );
         sa[threadIdx.x] = ( 1.0 - A.relax ) * A.dst[idx] + sc * ( A.rhs[idx] - A.rest[idx] - wNext * A.dst[nxt] );
      }
      __syncthreads();
      if ( threadIdx.x == 0 )
      {
         const int m = cnt - base < 64 ? cnt - base : 64;
         for ( int k = 0; k < m; ++k )
         {
            carry = fma( b, carry, sa[k] );
            sr[k] = carry;
         }
      }
      __syncthreads();
      if ( s <= cnt )
         A.dst[idx] = sr[threadIdx.x];
      __syncthreads();
   }
}

// face coordinates (i, j) -> cell coordinates; in-plane neighbour directions in the order of faceW[.][1..6]
struct FaceFrame
{
   int ox, oy, oz, ax, ay, az, bx, by, bz;
   __device__ inline FaceFrame( const SorShellCell& A, int f )
   {
      int ux, uy, uz, vx, vy, vz, wx, wy, wz;
      vtx( A.S->faceV[f][0], A.N - 1, ox, oy, oz );
      vtx( A.S->faceV[f][0], 1, ux, uy, uz );
      vtx( A.S->faceV[f][1], 1, vx, vy, vz );
      vtx( A.S->faceV[f][2], 1, wx, wy, wz );
      ax = vx - ux, ay = vy - uy, az = vz - uz;
      bx = wx - ux, by = wy - uy, bz = wz - uz;
   }
   __device__ inline int index( int N, int i, int j ) const
   {
      return cell_index( N, ox + i * ax + j * bx, oy + i * ay + j * by, oz + i * az + j * bz );
   }
};
__constant__ int kFaceDirs[6][2] = { { -1, 0 }, { 1, 0 }, { 0, -1 }, { 0, 1 }, { 1, -1 }, { -1, 1 } }; // W E S N SE NW

__device__ inline bool face_interior( int N, int i, int j ) { return i >= 1 && j >= 1 && i + j <= N - 2; }

constexpr int kPrepThreads = 128;

// rest[p] <- a_p = (1-relax) u_p + relax/c ( rhs_p - rest_p - sum_{not-yet-updated in-plane neighbours} w u
//                                            - sum_{already-final in-plane neighbours on the face boundary} w u )
__global__ __launch_bounds__( kPrepThreads ) void p1_sor_face_prep_kernel( const SorShellArgs AA )
{
   const SorShellCell A( AA, blockIdx.z );
   const int          f = blockIdx.y;
   if ( !( ( A.mask >> ( 6 + f ) ) & 1u ) )
      return;
   const int       N = A.N, j = blockIdx.x + 1;
   const FaceFrame F( A, f );
   const double    sc = A.relax / A.S->faceW[f][0];
   for ( int i = 1 + (int) threadIdx.x; i <= N - 2 - j; i += kPrepThreads )
   {
      const int idx = F.index( N, i, j );
      double    t   = A.rhs[idx] - A.rest[idx];
#pragma unroll
      for ( int d = 0; d < 6; ++d )
      {
         // forward: W, S, SE are updated before (i,j); backwards: E, N, NW
         const bool prev = A.backwards ? ( d & 1 ) : !( d & 1 );
         const int  ni = i + kFaceDirs[d][0], nj = j + kFaceDirs[d][1];
         if ( !prev || !face_interior( N, ni, nj ) )
            t -= A.S->faceW[f][1 + d] * A.dst[F.index( N, ni, nj )];
      }
      A.rest[idx] = ( 1.0 - A.relax ) * A.dst[idx] + sc * t;
   }
}

constexpr int kSweepDepth = 8; // software prefetch distance of the a-values along a row

template < int RPT >
__global__ __launch_bounds__( 1024 ) void p1_sor_face_sweep_kernel( const SorShellArgs AA )
{
   const SorShellCell A( AA, blockIdx.y );
   const int          f = blockIdx.x;
   if ( !( ( A.mask >> ( 6 + f ) ) & 1u ) )
      return;
   extern __shared__ double val[]; // [3][stride]
   const int                N = A.N, R = N - 3, bw = A.backwards;
   const int                stride = RPT * (int) blockDim.x + 2;
   const FaceFrame          F( A, f );
   const double             sc    = A.relax / A.S->faceW[f][0];
   const int                kappa = bw ? 1 : 2;
   // neighbour in the same row, neighbour of step tau-1, neighbour of step tau-2
   const int    dL = bw ? 1 : 0, d1 = bw ? 5 : 4, d2 = bw ? 3 : 2;
   const double bL = -sc * A.S->faceW[f][1 + dL], b1 = -sc * A.S->faceW[f][1 + d1], b2 = -sc * A.S->faceW[f][1 + d2];
   const int    tauMin = 1 + kappa, tauMax = bw ? 2 * N - 6 : 2 * N - 5;

   double q[RPT][kSweepDepth];
   double left[RPT];
   int    rr[RPT], jj[RPT], len[RPT];
#pragma unroll
   for ( int u = 0; u < RPT; ++u )
   {
      rr[u]   = (int) threadIdx.x + 1 + u * (int) blockDim.x; // ordinal of the row in sweep order
      jj[u]   = bw ? R + 1 - rr[u] : rr[u];
      len[u]  = N - 2 - jj[u];
      left[u] = 0.0;
#pragma unroll
      for ( int k = 0; k < kSweepDepth; ++k )
      {
         const int s = tauMin + k - kappa * rr[u];
         q[u][k]     = ( rr[u] <= R && s >= 1 && s <= len[u] ) ? A.rest[F.index( N, bw ? len[u] + 1 - s : s, jj[u] )] : 0.0;
      }
   }
   int p0 = tauMin % 3; // LDS ring slot of the current step
   for ( int tau0 = tauMin; tau0 <= tauMax; tau0 += kSweepDepth )
   {
#pragma unroll
      for ( int k = 0; k < kSweepDepth; ++k )
      {
         const int tau = tau0 + k;
         const int pm1 = p0 == 0 ? 2 : p0 - 1, pm2 = pm1 == 0 ? 2 : pm1 - 1;
#pragma unroll
         for ( int u = 0; u < RPT; ++u )
         {
            const int s = tau - kappa * rr[u];
            if ( rr[u] <= R && s >= 1 && s <= len[u] )
            {
               const int i = bw ? len[u] + 1 - s : s, j = jj[u];
               if ( s == 1 )
                  left[u] = 0.0;
               const double n1 = face_interior( N, i + kFaceDirs[d1][0], j + kFaceDirs[d1][1] ) ? val[pm1 * stride + rr[u] - 1] : 0.0;
               const double n2 = face_interior( N, i + kFaceDirs[d2][0], j + kFaceDirs[d2][1] ) ? val[pm2 * stride + rr[u] - 1] : 0.0;
               const double v  = fma( bL, left[u], fma( b1, n1, fma( b2, n2, q[u][k] ) ) );
               left[u]                    = v;
               val[p0 * stride + rr[u]]   = v;
               A.dst[F.index( N, i, j )]  = v;
            }
            const int sn = s + kSweepDepth;
            q[u][k]      = ( rr[u] <= R && sn >= 1 && sn <= len[u] ) ? A.rest[F.index( N, bw ? len[u] + 1 - sn : sn, jj[u] )] : 0.0;
         }
         __syncthreads();
         p0 = p0 == 2 ? 0 : p0 + 1;
      }
   }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same sweep in HyTeG's OWN macro-face layout [ tri(N) face DoFs | ghost layer of cell 0 | ghost layer of cell 1 ]
// (hyteg_hip_p1_sor_face3d; replaces sor_3D_macroface_P1{,_one_sided}{,_backwards}, reference loop
// P1Operator::smooth_sor_face3D, P1Operator.hpp:1424-1503).  Which face-array entry a stencil leaf of a neighbour cell lands on
// is a property of the leaf and the cell's vertex map alone (the map is linear), so the host folds the 2 x 14 leaves into
// seven in-plane weights (centre + W E S N SE NW, summed over the cells) and a list of ghost-layer leaves; the preparation
// kernel forms  a = (1 - relax) u + relax / c ( rhs - ghost leaves - in-plane neighbours that are not updated before the
// point ), the sweep kernel is the hyperplane wavefront above with the face array's own index function.
// ---------------------------------------------------------------------------------------------------------------------
struct FaceSorArgs
{
   double*       dst;
   const double* rhs;
   double*       work; // tri(N) doubles
   int           N, backwards;
   double        relax;
   double        W[7]; // centre, then W E S N SE NW (kFaceDirs order), each summed over the neighbour cells
   int           nghost;
   int           gk[32], gdx[32], gdy[32]; // ghost-layer leaves: neighbour cell, offset in face coordinates
   double        gw[32];
};

__global__ __launch_bounds__( kPrepThreads ) void p1_sor_face3d_prep_kernel( const FaceSorArgs A )
{
   const int    N = A.N, j = blockIdx.x + 1;
   const double sc = A.relax / A.W[0];
   for ( int i = 1 + (int) threadIdx.x; i <= N - 2 - j; i += kPrepThreads )
   {
      const int idx = row_start( N, j ) + i;
      double    t   = A.rhs[idx];
      for ( int g = 0; g < A.nghost; ++g )
         t -= A.gw[g] * A.dst[tri( N ) + A.gk[g] * tri( N - 1 ) + row_start( N - 1, j + A.gdy[g] ) + i + A.gdx[g]];
#pragma unroll
      for ( int d = 0; d < 6; ++d )
      {
         // forward: W, S, SE are updated before (i,j); backwards: E, N, NW
         const bool prev = A.backwards ? ( d & 1 ) : !( d & 1 );
         const int  ni = i + kFaceDirs[d][0], nj = j + kFaceDirs[d][1];
         if ( !prev || !face_interior( N, ni, nj ) )
            t -= A.W[1 + d] * A.dst[row_start( N, nj ) + ni];
      }
      A.work[idx] = ( 1.0 - A.relax ) * A.dst[idx] + sc * t;
   }
}

template < int RPT >
__global__ __launch_bounds__( 1024 ) void p1_sor_face3d_sweep_kernel( const FaceSorArgs A )
{
   extern __shared__ double val[]; // [3][stride]
   const int                N = A.N, R = N - 3, bw = A.backwards;
   const int                stride = RPT * (int) blockDim.x + 2;
   const double             sc     = A.relax / A.W[0];
   const int                kappa  = bw ? 1 : 2;
   // neighbour in the same row, neighbour of step tau-1, neighbour of step tau-2
   const int    dL = bw ? 1 : 0, d1 = bw ? 5 : 4, d2 = bw ? 3 : 2;
   const double bL = -sc * A.W[1 + dL], b1 = -sc * A.W[1 + d1], b2 = -sc * A.W[1 + d2];
   const int    tauMin = 1 + kappa, tauMax = bw ? 2 * N - 6 : 2 * N - 5;

   double q[RPT][kSweepDepth];
   double left[RPT];
   int    rr[RPT], jj[RPT], len[RPT];
#pragma unroll
   for ( int u = 0; u < RPT; ++u )
   {
      rr[u]   = (int) threadIdx.x + 1 + u * (int) blockDim.x; // ordinal of the row in sweep order
      jj[u]   = bw ? R + 1 - rr[u] : rr[u];
      len[u]  = N - 2 - jj[u];
      left[u] = 0.0;
#pragma unroll
      for ( int k = 0; k < kSweepDepth; ++k )
      {
         const int s = tauMin + k - kappa * rr[u];
         q[u][k]     = ( rr[u] <= R && s >= 1 && s <= len[u] ) ? A.work[row_start( N, jj[u] ) + ( bw ? len[u] + 1 - s : s )] : 0.0;
      }
   }
   int p0 = tauMin % 3; // LDS ring slot of the current step
   for ( int tau0 = tauMin; tau0 <= tauMax; tau0 += kSweepDepth )
   {
#pragma unroll
      for ( int k = 0; k < kSweepDepth; ++k )
      {
         const int tau = tau0 + k;
         const int pm1 = p0 == 0 ? 2 : p0 - 1, pm2 = pm1 == 0 ? 2 : pm1 - 1;
#pragma unroll
         for ( int u = 0; u < RPT; ++u )
         {
            const int s = tau - kappa * rr[u];
            if ( rr[u] <= R && s >= 1 && s <= len[u] )
            {
               const int i = bw ? len[u] + 1 - s : s, j = jj[u];
               if ( s == 1 )
                  left[u] = 0.0;
               const double n1 = face_interior( N, i + kFaceDirs[d1][0], j + kFaceDirs[d1][1] ) ? val[pm1 * stride + rr[u] - 1] : 0.0;
               const double n2 = face_interior( N, i + kFaceDirs[d2][0], j + kFaceDirs[d2][1] ) ? val[pm2 * stride + rr[u] - 1] : 0.0;
               const double v  = fma( bL, left[u], fma( b1, n1, fma( b2, n2, q[u][k] ) ) );
               left[u]                       = v;
               val[p0 * stride + rr[u]]      = v;
               A.dst[row_start( N, j ) + i]  = v;
            }
            const int sn = s + kSweepDepth;
            q[u][k]      = ( rr[u] <= R && sn >= 1 && sn <= len[u] ) ? A.work[row_start( N, jj[u] ) + ( bw ? len[u] + 1 - sn : sn )] : 0.0;
         }
         __syncthreads();
         p0 = p0 == 2 ? 0 : p0 + 1;
      }
   }
}

bool is_perm( const int* v, int n, const int* ref )
{
   for ( int a = 0; a < n; ++a )
   {
      bool found = false;
      for ( int b = 0; b < n; ++b )
         found = found || v[a] == ref[b];
      if ( !found )
         return false;
      for ( int b = a + 1; b < n; ++b )
         if ( v[a] == v[b] )
            return false;
   }
   return true;
}

const int kEdgeVerts[6][2] = { { 0, 1 }, { 0, 2 }, { 1, 2 }, { 0, 3 }, { 1, 3 }, { 2, 3 } };
const int kFaceVerts[4][3] = { { 0, 1, 2 }, { 0, 1, 3 }, { 0, 2, 3 }, { 1, 2, 3 } };

} // namespace

// launches for the cells already placed in A (ncells of them), classes selected by the union of their masks
static int launch_sor_shell( SorShellArgs& A, int ncells, unsigned any_mask, int backwards, hipStream_t s )
{
   const int  N        = A.N;
   const bool vertices = ( any_mask >> 10 ) & 0xFu, edges = ( any_mask & 0x3Fu ) && N >= 3, faces = ( ( any_mask >> 6 ) & 0xFu ) && N >= 5;
   auto       doVertices = [&]() { hipLaunchKernelGGL( p1_sor_vertices_kernel, dim3( ncells ), dim3( 64 ), 0, s, A ); };
   auto       doEdges    = [&]() { hipLaunchKernelGGL( p1_sor_edges_kernel, dim3( 6, ncells ), dim3( 64 ), 0, s, A ); };
   auto       doFaces    = [&]() {
      const int R = N - 3;
      hipLaunchKernelGGL( p1_sor_face_prep_kernel, dim3( R, 4, ncells ), dim3( kPrepThreads ), 0, s, A );
      const int    rpt     = R > 512 ? 2 : 1;
      const int    threads = ( ( ( R + rpt - 1 ) / rpt + 63 ) / 64 ) * 64;
      const size_t lds     = size_t( 3 ) * ( size_t( rpt ) * threads + 2 ) * sizeof( double );
      if ( rpt == 2 )
         hipLaunchKernelGGL( p1_sor_face_sweep_kernel< 2 >, dim3( 4, ncells ), dim3( threads ), lds, s, A );
      else
         hipLaunchKernelGGL( p1_sor_face_sweep_kernel< 1 >, dim3( 4, ncells ), dim3( threads ), lds, s, A );
   };
   if ( !backwards )
   {
      if ( vertices )
         doVertices();
      if ( edges )
         doEdges();
      if ( faces )
         doFaces();
   }
   else
   {
      if ( faces )
         doFaces();
      if ( edges )
         doEdges();
      if ( vertices )
         doVertices();
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

extern "C" {

HYTEG_HIP_API int hyteg_hip_p1_sor_shell_cell( double*            dst,
                                               const double*      rhs,
                                               double*            rest,
                                               int                level,
                                               const int*         edge_verts,
                                               const double*      edge_w,
                                               const int*         face_verts,
                                               const double*      face_w,
                                               const double*      vertex_w,
                                               double             relax,
                                               unsigned           mask,
                                               int                backwards,
                                               hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && rhs && rest && edge_verts && edge_w && face_verts && face_w && vertex_w, "p1_sor_shell_cell: null pointer" );
   HH_REQUIRE( level >= 0 && level <= HYTEG_HIP_MAX_LEVEL, "p1_sor_shell_cell: level out of range [0,11]" );
   HH_REQUIRE( dst != rhs && dst != rest && rhs != rest, "p1_sor_shell_cell: dst, rhs and rest must not alias" );
   mask &= HYTEG_HIP_MASK_SHELL;
   if ( mask == 0 )
      return HYTEG_HIP_OK;
   SorShellArgs A{};
   A.dst[0] = dst, A.rhs[0] = rhs, A.rest[0] = rest, A.mask[0] = mask;
   A.table = nullptr, A.N = ( 1 << level ) + 1, A.relax = relax, A.backwards = backwards ? 1 : 0;
   SorShellDesc& S = A.one;
   for ( int e = 0; e < 6; ++e )
   {
      HH_REQUIRE( is_perm( edge_verts + 2 * e, 2, kEdgeVerts[e] ), "p1_sor_shell_cell: edge_verts[e] must name the two vertices of cell edge e" );
      for ( int k = 0; k < 2; ++k )
         S.edgeV[e][k] = edge_verts[2 * e + k];
      for ( int k = 0; k < 3; ++k )
         S.edgeW[e][k] = edge_w[3 * e + k];
      HH_REQUIRE( !( ( mask >> e ) & 1u ) || S.edgeW[e][0] != 0.0, "p1_sor_shell_cell: zero centre weight on an edge" );
   }
   for ( int f = 0; f < 4; ++f )
   {
      HH_REQUIRE( is_perm( face_verts + 3 * f, 3, kFaceVerts[f] ), "p1_sor_shell_cell: face_verts[f] must name the three vertices of cell face f" );
      for ( int k = 0; k < 3; ++k )
         S.faceV[f][k] = face_verts[3 * f + k];
      for ( int k = 0; k < 7; ++k )
         S.faceW[f][k] = face_w[7 * f + k];
      HH_REQUIRE( !( ( mask >> ( 6 + f ) ) & 1u ) || S.faceW[f][0] != 0.0, "p1_sor_shell_cell: zero centre weight on a face" );
   }
   for ( int k = 0; k < 4; ++k )
   {
      S.vertexW[k] = vertex_w[k];
      HH_REQUIRE( !( ( mask >> ( 10 + k ) ) & 1u ) || S.vertexW[k] != 0.0, "p1_sor_shell_cell: zero centre weight on a vertex" );
   }
   return launch_sor_shell( A, 1, mask, backwards, as_stream( stream ) );
}

HYTEG_HIP_API int hyteg_hip_p1_sor_shell_cells( int                               ncells,
                                                double* const*                    dst,
                                                const double* const*              rhs,
                                                double* const*                    rest,
                                                int                               level,
                                                const hyteg_hip_sor_shell_tables* tables_dev,
                                                double                            relax,
                                                const unsigned*                   masks,
                                                int                               backwards,
                                                hyteg_hip_stream_t                stream )
{
   HH_REQUIRE( ncells >= 1 && ncells <= HYTEG_HIP_MAX_BATCH, "p1_sor_shell_cells: ncells must be 1..HYTEG_HIP_MAX_BATCH" );
   HH_REQUIRE( dst && rhs && rest && tables_dev && masks, "p1_sor_shell_cells: null pointer" );
   HH_REQUIRE( level >= 0 && level <= HYTEG_HIP_MAX_LEVEL, "p1_sor_shell_cells: level out of range [0,11]" );
   SorShellArgs A{};
   unsigned     any = 0;
   for ( int c = 0; c < ncells; ++c )
   {
      HH_REQUIRE( dst[c] && rhs[c] && rest[c] && dst[c] != rhs[c] && dst[c] != rest[c] && rhs[c] != rest[c],
                  "p1_sor_shell_cells: null or aliased arrays" );
      A.dst[c] = dst[c], A.rhs[c] = rhs[c], A.rest[c] = rest[c], A.mask[c] = masks[c] & HYTEG_HIP_MASK_SHELL;
      any |= A.mask[c];
   }
   if ( any == 0 )
      return HYTEG_HIP_OK;
   A.table = reinterpret_cast< const SorShellDesc* >( tables_dev );
   A.N = ( 1 << level ) + 1, A.relax = relax, A.backwards = backwards ? 1 : 0;
   return launch_sor_shell( A, ncells, any, backwards, as_stream( stream ) );
}

HYTEG_HIP_API size_t hyteg_hip_p1_sor_face3d_workspace( int level )
{
   const int N = ( 1 << level ) + 1;
   return (size_t) tri( N ) * sizeof( double );
}

HYTEG_HIP_API int hyteg_hip_p1_sor_face3d( double*            dst_face,
                                           const double*      rhs_face,
                                           double*            work,
                                           int                level,
                                           int                ncells,
                                           const int*         vmaps,
                                           const double*      w,
                                           double             relax,
                                           int                backwards,
                                           hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst_face && rhs_face && work && vmaps && w, "p1_sor_face3d: null pointer" );
   HH_REQUIRE( level >= 1 && level <= HYTEG_HIP_MAX_LEVEL, "p1_sor_face3d: level out of range [1,11]" );
   HH_REQUIRE( ncells == 1 || ncells == 2, "p1_sor_face3d: a macro-face has 1 or 2 neighbour cells" );
   HH_REQUIRE( dst_face != rhs_face, "p1_sor_face3d: dst and rhs must not alias" );
   static const int offs[15][3] = { { 0, 0, -1 }, { 1, 0, -1 }, { -1, 1, -1 }, { 0, 1, -1 }, { 0, -1, 0 }, { 1, -1, 0 }, { -1, 0, 0 }, { 0, 0, 0 },
                                    { 1, 0, 0 },  { -1, 1, 0 }, { 0, 1, 0 },   { 0, -1, 1 }, { 1, -1, 1 }, { -1, 0, 1 }, { 0, 0, 1 } };
   static const int dirs[6][2]   = { { -1, 0 }, { 1, 0 }, { 0, -1 }, { 0, 1 }, { 1, -1 }, { -1, 1 } }; // = kFaceDirs
   FaceSorArgs A{};
   A.dst = dst_face, A.rhs = rhs_face, A.work = work, A.N = ( 1 << level ) + 1, A.backwards = backwards ? 1 : 0, A.relax = relax;
   for ( int k = 0; k < ncells; ++k )
   {
      const int v0 = vmaps[3 * k], v1 = vmaps[3 * k + 1], v2 = vmaps[3 * k + 2];
      HH_REQUIRE( v0 >= 0 && v0 < 4 && v1 >= 0 && v1 < 4 && v2 >= 0 && v2 < 4 && v0 != v1 && v0 != v2 && v1 != v2, "p1_sor_face3d: bad vertex map" );
      const int v3 = 6 - v0 - v1 - v2;
      for ( int s = 0; s < 15; ++s )
      {
         // a cell offset in barycentric coordinates, read off in the face's frame (the vertex map is linear)
         const int bary[4] = { -offs[s][0] - offs[s][1] - offs[s][2], offs[s][0], offs[s][1], offs[s][2] };
         const int dfx = bary[v1], dfy = bary[v2], dfz = bary[v3];
         const double ws = w[15 * k + s];
         if ( dfz < 0 )
            continue; // the leaf lies outside this cell: not part of its share
         if ( dfz == 0 )
         {
            if ( dfx == 0 && dfy == 0 )
            {
               A.W[0] += ws;
               continue;
            }
            int d = -1;
            for ( int e = 0; e < 6; ++e )
               if ( dirs[e][0] == dfx && dirs[e][1] == dfy )
                  d = e;
            HH_REQUIRE( d >= 0, "p1_sor_face3d: in-plane leaf off the face stencil" );
            A.W[1 + d] += ws;
            continue;
         }
         HH_REQUIRE( A.nghost < 32, "p1_sor_face3d: too many ghost leaves" );
         A.gk[A.nghost] = k, A.gdx[A.nghost] = dfx, A.gdy[A.nghost] = dfy, A.gw[A.nghost] = ws;
         ++A.nghost;
      }
   }
   HH_REQUIRE( A.W[0] != 0.0, "p1_sor_face3d: zero centre weight" );
   const int R = A.N - 3;
   if ( R < 1 )
      return HYTEG_HIP_OK; // no inner face DoFs below level 2
   hipStream_t st = as_stream( stream );
   hipLaunchKernelGGL( p1_sor_face3d_prep_kernel, dim3( R ), dim3( kPrepThreads ), 0, st, A );
   const int    rpt     = R > 512 ? ( R > 1024 ? 4 : 2 ) : 1;
   const int    threads = ( ( ( R + rpt - 1 ) / rpt + 63 ) / 64 ) * 64;
   const size_t lds     = size_t( 3 ) * ( size_t( rpt ) * threads + 2 ) * sizeof( double );
   if ( rpt == 4 )
      hipLaunchKernelGGL( p1_sor_face3d_sweep_kernel< 4 >, dim3( 1 ), dim3( threads ), lds, st, A );
   else if ( rpt == 2 )
      hipLaunchKernelGGL( p1_sor_face3d_sweep_kernel< 2 >, dim3( 1 ), dim3( threads ), lds, st, A );
   else
      hipLaunchKernelGGL( p1_sor_face3d_sweep_kernel< 1 >, dim3( 1 ), dim3( threads ), lds, st, A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

} // extern "C"
