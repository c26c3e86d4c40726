// Quadratic (P2) grid transfer on one macro-cell:
//   P2toP2QuadraticProlongation::prolongateAdditively3D  (src/hyteg/gridtransferoperators/P2toP2QuadraticProlongation.cpp:217-424,
//       generatedKernels/prolongate_3D_macrocell_P2_push_from_{vertexdofs,edgedofs}.cpp: a scatter over the coarse DoFs)
//   P2toP2QuadraticRestriction::restrictAdditively3D    (P2toP2QuadraticRestriction.cpp:131-286,
//       generatedKernels/restrict_3D_macrocell_P2_update_{vertexdofs,edgedofs}.cpp)
// Both are GATHERS here (no atomics, fixed summation order), driven by two small tables that do not depend on the level:
//  * prolongation: a fine DoF (vertex DoF, or edge DoF of one of the seven orientations) with index parities (px,py,pz)
//    lies in the relative interior of exactly one coarse micro-simplex (a coarse micro-vertex, -edge, -face or -cell);
//    its value is the coarse function's quadratic interpolant there: sum of shape-function values
//    lambda_i ( 2 lambda_i - 1 ) (coarse vertex DoFs) and 4 lambda_i lambda_j (coarse edge DoFs) over that simplex' DoFs,
//    at most ten terms, with offsets relative to ( x/2, y/2, z/2 ).  64 patterns (8 kinds x 8 parities).  All DoFs of
//    that simplex belong to this macro-cell whenever the fine DoF does, so no neighbour data is needed;
//  * restriction = the transpose: a coarse DoF sums the fine DoFs of its basis function's support (125 for a vertex
//    DoF, 27 for an edge DoF inside the cell) with the same weights; fine DoFs outside the macro-cell are skipped and
//    fine DoFs on a macro-face / -edge / -vertex shared by k cells are scaled by 1/k, so that the additive exchange over
//    the cells counts every fine DoF once (the numNeighborCells* arguments of the reference kernels).
// The tables are built on the host from the micro-cell geometry (six micro-cell types, seven edge orientations) and the
// shape functions -- nothing is transcribed from the generated kernels.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.hpp"

using namespace hyteg_hip;

namespace {

constexpr int kThreads = 256;

struct TEntry
{
   signed char kind; // 0: vertex array, 1..7: edge array block X, Y, Z, XY, XZ, YZ, XYZ
   signed char dx, dy, dz;
   float       pad;
   double      w;
};
static_assert( sizeof( TEntry ) == 16, "TEntry" );
constexpr int kMaxProlong  = 10;
constexpr int kMaxRestrict = 128;
struct TransferTables
{
   TEntry      prolong[64][kMaxProlong]; // pattern = fine kind * 8 + px + 2 py + 4 pz; offsets relative to ( x>>1, y>>1, z>>1 )
   int         nprolong[64];
   TEntry      restrict_[8][kMaxRestrict]; // per coarse kind; offsets relative to 2 * ( coarse index )
   int         nrestrict[8];
};

// micro-vertices of the six micro-cell types (celldof::macrocell::getMicroVerticesFromMicroCell, CellDoFIndexing.hpp:155-198)
const int kMicroVerts[6][4][3] = { { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } }, { { 1, 0, 0 }, { 1, 1, 0 }, { 0, 1, 0 }, { 1, 0, 1 } },
                                   { { 1, 0, 0 }, { 0, 1, 0 }, { 1, 0, 1 }, { 0, 0, 1 } }, { { 1, 1, 0 }, { 1, 1, 1 }, { 0, 1, 1 }, { 1, 0, 1 } },
                                   { { 1, 0, 1 }, { 0, 1, 1 }, { 0, 0, 1 }, { 0, 1, 0 } }, { { 0, 1, 0 }, { 1, 1, 0 }, { 1, 0, 1 }, { 0, 1, 1 } } };
// end points of an edge DoF relative to its logical index, by orientation X, Y, Z, XY, XZ, YZ, XYZ (EdgeDoFIndexing.hpp)
const int            kEnds[7][2][3] = { { { 0, 0, 0 }, { 1, 0, 0 } }, { { 0, 0, 0 }, { 0, 1, 0 } }, { { 0, 0, 0 }, { 0, 0, 1 } },
                                        { { 1, 0, 0 }, { 0, 1, 0 } }, { { 1, 0, 0 }, { 0, 0, 1 } }, { { 0, 1, 0 }, { 0, 0, 1 } },
                                        { { 0, 1, 0 }, { 1, 0, 1 } } };
__constant__ int kEndsDev[7][2][3] = { { { 0, 0, 0 }, { 1, 0, 0 } }, { { 0, 0, 0 }, { 0, 1, 0 } }, { { 0, 0, 0 }, { 0, 0, 1 } },
                                        { { 1, 0, 0 }, { 0, 1, 0 } }, { { 1, 0, 0 }, { 0, 0, 1 } }, { { 0, 1, 0 }, { 0, 0, 1 } },
                                        { { 0, 1, 0 }, { 1, 0, 1 } } };

// the edge DoF between the micro-vertices a and b: orientation (1..7) and logical index
bool edge_between( const int* a, const int* b, int& kind, int* idx )
{
   for ( int o = 0; o < 7; ++o )
   {
      const int d[3] = { kEnds[o][1][0] - kEnds[o][0][0], kEnds[o][1][1] - kEnds[o][0][1], kEnds[o][1][2] - kEnds[o][0][2] };
      for ( int s = 0; s < 2; ++s )
      {
         const int* p = s ? b : a;
         const int* q = s ? a : b;
         if ( q[0] - p[0] == d[0] && q[1] - p[1] == d[1] && q[2] - p[2] == d[2] )
         {
            kind = o + 1;
            for ( int r = 0; r < 3; ++r )
               idx[r] = p[r] - kEnds[o][0][r];
            return true;
         }
      }
   }
   return false;
}

void build_tables( TransferTables& T )
{
   std::memset( &T, 0, sizeof( T ) );
   for ( int kf = 0; kf < 8; ++kf )
      for ( int par = 0; par < 8; ++par )
      {
         const int p[3] = { par & 1, ( par >> 1 ) & 1, ( par >> 2 ) & 1 };
         // position of the fine DoF relative to the coarse index ( x>>1, y>>1, z>>1 ), in quarters of a coarse cell
         int Q[3];
         for ( int r = 0; r < 3; ++r )
            Q[r] = 2 * p[r] + ( kf == 0 ? 0 : kEnds[kf - 1][0][r] + kEnds[kf - 1][1][r] );
         // a coarse micro-cell that contains it (any: the interpolant is continuous, and the terms with non-zero weight are
         // those of the smallest sub-simplex containing the point)
         bool found = false;
         for ( int dz = -1; dz <= 1 && !found; ++dz )
            for ( int dy = -1; dy <= 1 && !found; ++dy )
               for ( int dx = -1; dx <= 1 && !found; ++dx )
                  for ( int t = 0; t < 6 && !found; ++t )
                  {
                     int V[4][3];
                     for ( int k = 0; k < 4; ++k )
                     {
                        V[k][0] = dx + kMicroVerts[t][k][0];
                        V[k][1] = dy + kMicroVerts[t][k][1];
                        V[k][2] = dz + kMicroVerts[t][k][2];
                     }
                     // barycentric coordinates of Q / 4 in the micro-cell V: solve with Cramer's rule (small integers: exact)
                     double M[3][3], rhs[3];
                     for ( int r = 0; r < 3; ++r )
                     {
                        for ( int k = 0; k < 3; ++k )
                           M[r][k] = 4.0 * ( V[k + 1][r] - V[0][r] );
                        rhs[r] = Q[r] - 4.0 * V[0][r];
                     }
                     auto det3 = []( const double A[3][3] ) {
                        return A[0][0] * ( A[1][1] * A[2][2] - A[1][2] * A[2][1] ) - A[0][1] * ( A[1][0] * A[2][2] - A[1][2] * A[2][0] ) +
                               A[0][2] * ( A[1][0] * A[2][1] - A[1][1] * A[2][0] );
                     };
                     const double D = det3( M );
                     double       lam[4];
                     for ( int k = 0; k < 3; ++k )
                     {
                        double Mk[3][3];
                        for ( int r = 0; r < 3; ++r )
                           for ( int c = 0; c < 3; ++c )
                              Mk[r][c] = c == k ? rhs[r] : M[r][c];
                        lam[k + 1] = det3( Mk ) / D;
                     }
                     lam[0] = 1.0 - lam[1] - lam[2] - lam[3];
                     if ( lam[0] < -1e-12 || lam[1] < -1e-12 || lam[2] < -1e-12 || lam[3] < -1e-12 )
                        continue;
                     found         = true;
                     const int pat = kf * 8 + par;
                     int&      n   = T.nprolong[pat];
                     for ( int k = 0; k < 4; ++k )
                     {
                        const double w = lam[k] * ( 2.0 * lam[k] - 1.0 );
                        if ( std::fabs( w ) > 1e-14 )
                           T.prolong[pat][n++] = TEntry{ 0, (signed char) V[k][0], (signed char) V[k][1], (signed char) V[k][2], 0.0f, w };
                     }
                     for ( int a = 0; a < 4; ++a )
                        for ( int b = a + 1; b < 4; ++b )
                        {
                           const double w = 4.0 * lam[a] * lam[b];
                           if ( std::fabs( w ) <= 1e-14 )
                              continue;
                           int kind, idx[3];
                           if ( !edge_between( V[a], V[b], kind, idx ) )
                              continue; // cannot happen: every pair of micro-cell vertices is joined by a micro-edge
                           T.prolong[pat][n++] = TEntry{ (signed char) kind, (signed char) idx[0], (signed char) idx[1], (signed char) idx[2], 0.0f, w };
                        }
                  }
      }
   // transpose: the fine DoF ( kf, 2 b + p ) takes w from the coarse DoF ( kc, b + d )  =>  the coarse DoF ( kc, C ) gives w to
   // the fine DoF ( kf, 2 C - 2 d + p )
   for ( int kf = 0; kf < 8; ++kf )
      for ( int par = 0; par < 8; ++par )
      {
         const int pat = kf * 8 + par;
         for ( int e = 0; e < T.nprolong[pat]; ++e )
         {
            const TEntry& s = T.prolong[pat][e];
            int&          n = T.nrestrict[(int) s.kind];
            if ( n >= kMaxRestrict )
               continue; // checked by the caller (support sizes are 125 / <= 45)
            T.restrict_[(int) s.kind][n++] = TEntry{ (signed char) kf,
                                                     (signed char) ( -2 * s.dx + ( par & 1 ) ),
                                                     (signed char) ( -2 * s.dy + ( ( par >> 1 ) & 1 ) ),
                                                     (signed char) ( -2 * s.dz + ( ( par >> 2 ) & 1 ) ),
                                                     0.0f,
                                                     s.w };
         }
      }
}

// device copy of the tables, one per device, built on first use
int get_tables( const TransferTables** out )
{
   static std::mutex                                 mtx;
   static std::vector< std::pair< int, TransferTables* > > cache;
   int                                               dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   std::lock_guard< std::mutex > lock( mtx );
   for ( auto& kv : cache )
      if ( kv.first == dev )
      {
         *out = kv.second;
         return HYTEG_HIP_OK;
      }
   static TransferTables host;
   build_tables( host );
   for ( int k = 0; k < 8; ++k )
      if ( host.nrestrict[k] >= kMaxRestrict )
         return fail( HYTEG_HIP_EINVAL, "p2 transfer: restriction table overflow" );
   void* d = nullptr;
   HH_CHECK_HIP( hipMalloc( &d, sizeof( TransferTables ) ) );
   HH_CHECK_HIP( hipMemcpy( d, &host, sizeof( TransferTables ), hipMemcpyHostToDevice ) );
   cache.push_back( { dev, static_cast< TransferTables* >( d ) } );
   *out = static_cast< TransferTables* >( d );
   return HYTEG_HIP_OK;
}

// ---- layout helpers (vertex array of width N; edge array: seven tetrahedral blocks of width n = N - 1, n - 1 for XYZ) ----
__device__ inline int width_of_kind( int N, int kind ) { return kind == 0 ? N : ( kind == 7 ? N - 2 : N - 1 ); }
__device__ inline bool dof_exists( int N, int kind, int x, int y, int z )
{
   const int W = width_of_kind( N, kind );
   return x >= 0 && y >= 0 && z >= 0 && x + y + z <= W - 1;
}
// 32-bit index arithmetic throughout (round 3; levels <= 9: the largest index is 6 tet(512) + tet(511) < 2^31): the 64-bit
// products of the first version were a large part of the ~600 instructions a fine DoF cost
__device__ inline int dof_offset( int N, int kind, int x, int y, int z )
{
   const int W = width_of_kind( N, kind );
   return ( kind == 0 ? 0 : ( kind - 1 ) * (int) tet32( (unsigned) ( N - 1 ) ) ) + cell_index( W, x, y, z );
}
// slice z of entry i of a tetrahedral array of width W
__device__ inline int slice_of( int W, int i )
{
   const unsigned rest = tet32( (unsigned) W ) - (unsigned) i;
   int            m    = (int) cbrtf( 6.0f * (float) rest );
   m                   = m < 1 ? 1 : ( m > W ? W : m );
   while ( m > 1 && tet32( (unsigned) ( m - 1 ) ) >= rest )
      --m;
   while ( tet32( (unsigned) m ) < rest )
      ++m;
   return W - m;
}
__device__ inline void decode( int W, int i, int& x, int& y, int& z )
{
   z           = slice_of( W, i );
   const int j = i - (int) ( tet32( (unsigned) W ) - tet32( (unsigned) ( W - z ) ) );
   y           = row_of( W - z, j );
   x           = j - row_start( W - z, y );
}
// point class 0..13 (slot of the macro-primitive: edge0..5, face0..3, vertex0..3) or 14 (inside the cell)
__device__ inline int class_from_flags( int f0, int f1, int f2, int f3 )
{
   const int cnt = f0 + f1 + f2 + f3;
   if ( cnt == 0 )
      return 14;
   if ( cnt == 1 )
      return 6 + ( f0 ? 0 : f1 ? 1 : f2 ? 2 : 3 );
   if ( cnt == 2 )
   {
      if ( f0 )
         return f1 ? 0 : ( f2 ? 1 : 2 );
      if ( f1 )
         return f2 ? 3 : 4;
      return 5;
   }
   if ( f0 && f1 && f2 )
      return 10;
   if ( f0 && f1 && f3 )
      return 11;
   if ( f0 && f2 && f3 )
      return 12;
   return 13;
}
__device__ inline int dof_class( int N, int kind, int x, int y, int z )
{
   if ( kind == 0 )
      return class_from_flags( z == 0, y == 0, x == 0, x + y + z == N - 1 );
   int f0 = 1, f1 = 1, f2 = 1, f3 = 1;
#pragma unroll
   for ( int e = 0; e < 2; ++e )
   {
      const int px = x + kEndsDev[kind - 1][e][0], py = y + kEndsDev[kind - 1][e][1], pz = z + kEndsDev[kind - 1][e][2];
      f0 &= pz == 0, f1 &= py == 0, f2 &= px == 0, f3 &= px + py + pz == N - 1;
   }
   return class_from_flags( f0, f1, f2, f3 );
}

struct P2TransferArgs
{
   double*               dstV;
   double*               dstE;
   const double*         srcV;
   const double*         srcE;
   const TransferTables* T;
   int                   Nc, Nf; // widths of the coarse / fine vertex arrays
   int                   update;
   unsigned              mask;
   Nnc14                 nncInv;
};

// Which kind and which block of that kind's array a workgroup takes.  Grid ( blocks, 8 ): kind = blockIdx.y -- all blocks of kind 0 run
// before those of kind 1, and every kind streams the whole source again (the restriction reads 95 MB for 23 MB of fine DoFs,
// profiles/r03_p2_transfer_rows.txt).  Grid ( 64 * groups ), a measurement switch that is off: workgroups g, g + 8, ... run on the same XCD; XCD j takes block 8 r + j of
// kind 0, then of kind 1, ... then block 8 ( r + 1 ) + j: the eight kinds of one region follow each other in the same L2.
__device__ inline void kind_and_block( int& kind, int& block )
{
   if ( gridDim.y == 8 )
   {
      kind = blockIdx.y, block = blockIdx.x;
      return;
   }
   const int g = blockIdx.x, q = g >> 3;
   kind  = q & 7;
   block = ( q >> 3 ) * 8 + ( g & 7 );
}

// one thread per fine DoF of a kind
__global__ __launch_bounds__( kThreads ) void p2_prolongate_kernel( const P2TransferArgs A )
{
   int kind, block;
   kind_and_block( kind, block );
   const int     W    = width_of_kind( A.Nf, kind );
   const int i    = (int) ( block * kThreads + threadIdx.x );
   if ( W <= 0 || i >= (int) tet32( (unsigned) W ) )
      return;
   int x, y, z;
   decode( W, i, x, y, z );
   if ( !( ( A.mask >> dof_class( A.Nf, kind, x, y, z ) ) & 1u ) )
      return;
   const int     pat = kind * 8 + ( x & 1 ) + 2 * ( y & 1 ) + 4 * ( z & 1 );
   const int     bx = x >> 1, by = y >> 1, bz = z >> 1;
   const TEntry* e = A.T->prolong[pat];
   const int n   = A.T->nprolong[pat];
   double    acc = 0.0;
   // one term per iteration: neighbouring lanes have different patterns (1 to 10 terms, 5 on average); padding every lane to ten
   // terms with all loads in flight (50.7 us) or to groups of four (49.0 us) costs more loads than it saves waiting (44.0 us)
   for ( int k = 0; k < n; ++k )
   {
      const int kc  = e[k].kind;
      const int off = dof_offset( A.Nc, kc, bx + e[k].dx, by + e[k].dy, bz + e[k].dz );
      acc           = fma( e[k].w, kc == 0 ? A.srcV[off] : A.srcE[off], acc );
   }
   double* out = kind == 0 ? A.dstV + i : A.dstE + ( kind - 1 ) * (int) tet32( (unsigned) ( A.Nf - 1 ) ) + i;
   if ( A.update == HYTEG_HIP_ADD )
      acc += *out;
   *out = acc;
}

// one thread per coarse DoF of a kind
__global__ __launch_bounds__( kThreads ) void p2_restrict_kernel( const P2TransferArgs A )
{
   int kind, block;
   kind_and_block( kind, block );
   const int     W    = width_of_kind( A.Nc, kind );
   const int i    = (int) ( block * kThreads + threadIdx.x );
   if ( W <= 0 || i >= (int) tet32( (unsigned) W ) )
      return;
   int x, y, z;
   decode( W, i, x, y, z );
   if ( !( ( A.mask >> dof_class( A.Nc, kind, x, y, z ) ) & 1u ) )
      return;
   const TEntry* e   = A.T->restrict_[kind];
   const int     n   = A.T->nrestrict[kind];
   double        acc = 0.0;
   // chunks of kChunk terms with all their loads in flight together (round 3; the table is padded with zero entries -- weight 0,
   // the fine vertex ( 2x, 2y, 2z ) -- up to kMaxRestrict, a multiple of kChunk).  A fine DoF outside the macro-cell contributes
   // weight 0 times entry 0 of the array: an exact 0, the bits of the loop that skipped it.
   constexpr int kChunk = 16;
   static_assert( kMaxRestrict % kChunk == 0, "restriction table padding" );
   for ( int k0 = 0; k0 < n; k0 += kChunk )
   {
      double v[kChunk], ws[kChunk];
#pragma unroll
      for ( int j = 0; j < kChunk; ++j )
      {
         const TEntry t  = e[k0 + j];
         const int    kf = t.kind;
         const int    fx = 2 * x + t.dx, fy = 2 * y + t.dy, fz = 2 * z + t.dz;
         const bool   ex = dof_exists( A.Nf, kf, fx, fy, fz );
         const int    cls   = dof_class( A.Nf, kf, fx, fy, fz );
         const double scale = cls == 14 ? 1.0 : A.nncInv.inv[cls];
         const int    off   = ex ? dof_offset( A.Nf, kf, fx, fy, fz ) : 0;
         v[j]               = kf == 0 ? A.srcV[off] : A.srcE[off];
         ws[j]              = ex ? t.w * scale : 0.0;
      }
#pragma unroll
      for ( int j = 0; j < kChunk; ++j )
         acc = fma( ws[j], v[j], acc );
   }
   double* out = kind == 0 ? A.dstV + i : A.dstE + ( kind - 1 ) * (int) tet32( (unsigned) ( A.Nc - 1 ) ) + i;
   *out        = acc;
}

// ---- round 3: both transfers by ROWS ----------------------------------------------------------------------------------------
// The thread-per-DoF kernels above spend ~400 instructions per fine DoF on index decoding (cube roots), 64-bit offsets, the class
// of every touched DoF and table look-ups (level 6 -> 7: 51 us / 76 us = 0.06 / 0.04 of the roofline).  By rows everything that
// does not depend on x is wave-uniform: a wave owns a run of one row (kind, y, z) -- a TILES_ROWS tile --, the parities of y
// and z, hence the list of (source kind, offset, weight) terms, are the wave's, the source row bases are scalar arithmetic, and
// a lane adds its x to them.
//
// Prolongation: lane l produces the fine DoFs x = x0 + 2 l and x0 + 2 l + 1 (x0 even): the two parities of x are two passes over
// wave-uniform term lists, and the coarse values of a term are read at ( x0 >> 1 ) + l + dx -- consecutive lanes, consecutive
// addresses.
__device__ inline int row_base32( int N, int kind, int y, int z ) // index of ( 0, y, z ) of `kind` in its (vertex or edge) array
{
   const int W = width_of_kind( N, kind );
   return ( kind == 0 ? 0 : ( kind - 1 ) * (int) tet64( N - 1 ) ) + cell_index( W, 0, y, z );
}
__global__ __launch_bounds__( 256 ) void p2_prolongate_rows_kernel( const P2TransferArgs A, const Tile* __restrict__ tiles, int ntiles )
{
   const int kind = blockIdx.y;
   const int t    = __builtin_amdgcn_readfirstlane( (int) ( blockIdx.x * 4 + ( threadIdx.x >> 6 ) ) );
   if ( t >= ntiles )
      return;
   const Tile tl = tiles[t];
   const int  y = tl.ya, z = tl.z, x0 = tl.yb; // x0 is a multiple of the tile capacity (128): even
   const int  W = width_of_kind( A.Nf, kind );
   const int  R = W - z - y; // length of the fine row of this kind (<= 0: the row does not exist for this kind)
   if ( x0 >= R )
      return;
   const int  lane = threadIdx.x & 63;
   const int  by = y >> 1, bz = z >> 1, bx0 = x0 >> 1;
   const int  fbase = row_base32( A.Nf, kind, y, z );
   double*    dst   = kind == 0 ? A.dstV : A.dstE;
   const bool all   = ( A.mask & HYTEG_HIP_MASK_ALL ) == HYTEG_HIP_MASK_ALL;
#pragma unroll
   for ( int px = 0; px < 2; ++px )
   {
      const int  x      = x0 + 2 * lane + px;
      const bool exists = x < R;
      const int  pat    = kind * 8 + px + 2 * ( y & 1 ) + 4 * ( z & 1 );
      // all kMaxProlong terms, unrolled: the entries behind the pattern's last term have weight 0 and offset 0 (a coarse vertex
      // that exists whenever the fine DoF does), so the ten table entries, ten row bases and ten loads are independent of each
      // other and in flight together (a loop over the pattern's own term count made every term wait for the previous one's
      // load: 88 us instead of 51 us for the thread-per-DoF kernel)
      double v[kMaxProlong], w[kMaxProlong];
#pragma unroll
      for ( int k = 0; k < kMaxProlong; ++k )
      {
         const TEntry  e   = A.T->prolong[pat][k]; // wave-uniform
         const int     kc  = e.kind;
         const int     cb  = row_base32( A.Nc, kc, by + e.dy, bz + e.dz ) + bx0 + e.dx;
         const double* src = kc == 0 ? A.srcV : A.srcE;
         v[k]              = src[exists ? cb + lane : 0]; // lanes past the row end read entry 0 and store nothing
         w[k]              = e.w;
      }
      double acc = 0.0;
#pragma unroll
      for ( int k = 0; k < kMaxProlong; ++k )
         acc = fma( w[k], v[k], acc ); // a zero weight adds an exact 0: the bits of the term-count loop
      if ( exists && ( all || ( ( A.mask >> dof_class( A.Nf, kind, x, y, z ) ) & 1u ) ) )
      {
         double* out = dst + fbase + x;
         *out        = A.update == HYTEG_HIP_ADD ? *out + acc : acc;
      }
   }
}

// Restriction: lane l owns the coarse DoF x = x0 + l of the wave's coarse row; the terms (fine kind, offset, weight) of the coarse
// kind are wave-uniform, a fine row exists or not for the whole wave, its base is scalar, and the lane reads 2 x + dx in it.
// The scale 1 / numNeighbourCells of a fine DoF on the macro-cell's boundary depends on x only through "on the face x = 0" and
// "on the face x + y + z = N - 1": four wave-uniform scales per term, selected per lane.
__device__ inline double class_scale( const Nnc14& inv, int f0, int f1, int f2, int f3 )
{
   const int cls = class_from_flags( f0, f1, f2, f3 );
   return cls == 14 ? 1.0 : inv.inv[cls];
}
__global__ __launch_bounds__( 256 ) void p2_restrict_rows_kernel( const P2TransferArgs A, const Tile* __restrict__ tiles, int ntiles )
{
   const int kind = blockIdx.y;
   const int t    = __builtin_amdgcn_readfirstlane( (int) ( blockIdx.x * 4 + ( threadIdx.x >> 6 ) ) );
   if ( t >= ntiles )
      return;
   const Tile tl = tiles[t];
   const int  y = tl.ya, z = tl.z, x0 = tl.yb;
   const int  W = width_of_kind( A.Nc, kind );
   const int  R = W - z - y;
   if ( x0 >= R )
      return;
   const int  lane   = threadIdx.x & 63;
   const int  x      = x0 + lane;
   const bool exists = x < R;
   const int  n      = A.T->nrestrict[kind];
   const int  Nf     = A.Nf;
   double     acc    = 0.0;
   for ( int k = 0; k < n; ++k )
   {
      const TEntry e  = A.T->restrict_[kind][k]; // wave-uniform
      const int    kf = e.kind;
      const int    fy = 2 * y + e.dy, fz = 2 * z + e.dz;
      const int    Wf = width_of_kind( Nf, kf );
      const int    Rf = Wf - fz - fy; // length of the fine row
      if ( fy < 0 || fz < 0 || Rf <= 0 )
         continue; // the whole fine row lies outside the macro-cell
      const int fb = row_base32( Nf, kf, fy, fz );
      const int fx = 2 * x + e.dx;
      // class of the fine DoF: end points p_e = ( fx, fy, fz ) + ends[e] (vertex DoFs: the point itself)
      int  f0 = 1, f1 = 1, ex0 = 0, ex1 = 0, s0 = 0, s1 = 0;
      if ( kf == 0 )
      {
         f0 = fz == 0, f1 = fy == 0;
         s0 = s1 = fy + fz;
      }
      else
      {
         const int( *E )[3] = kEndsDev[kf - 1];
         f0  = ( fz + E[0][2] == 0 ) && ( fz + E[1][2] == 0 );
         f1  = ( fy + E[0][1] == 0 ) && ( fy + E[1][1] == 0 );
         ex0 = E[0][0], ex1 = E[1][0];
         s0  = fy + fz + E[0][0] + E[0][1] + E[0][2];
         s1  = fy + fz + E[1][0] + E[1][1] + E[1][2];
      }
      const double sc00 = class_scale( A.nncInv, f0, f1, 0, 0 ), sc10 = class_scale( A.nncInv, f0, f1, 1, 0 ),
                   sc01 = class_scale( A.nncInv, f0, f1, 0, 1 ), sc11 = class_scale( A.nncInv, f0, f1, 1, 1 );
      const bool in  = exists && fx >= 0 && fx < Rf;
      const bool f2  = ( fx + ex0 == 0 ) && ( fx + ex1 == 0 );
      const bool f3  = ( fx + s0 == Nf - 1 ) && ( fx + s1 == Nf - 1 );
      const double sc = f2 ? ( f3 ? sc11 : sc10 ) : ( f3 ? sc01 : sc00 );
      const double* src = kf == 0 ? A.srcV : A.srcE;
      const double  v   = src[in ? fb + fx : fb];
      acc               = in ? fma( e.w * sc, v, acc ) : acc;
   }
   if ( exists && ( ( A.mask >> dof_class( A.Nc, kind, x, y, z ) ) & 1u ) )
   {
      double* out = kind == 0 ? A.dstV : A.dstE;
      out[row_base32( A.Nc, kind, y, z ) + x] = acc;
   }
}

// measurement switch: HYTEG_HIP_P2_TRANSFER_ROWS=1 selects the row kernels of round 3 -- correct (tests/test_gpu_p2_transfer.py
// passes with them) but SLOWER than the thread-per-DoF kernels at level 6 -> 7: prolongation 88 us with a loop over the pattern's
// term count, 136 us with all ten terms unrolled, against 51 us; restriction 166 us against 77 us.  The row bases of up to
// 2 x 10 / 125 wave-uniform terms are scalar arithmetic (triangular and tetrahedral numbers, a division by 3 each), and the CU's
// ONE scalar unit serialises them for all its waves: ~2000 scalar instructions per wave x 80,000 waves.  What would pay is term
// lists known at compile time (as in the P2 apply), so that the bases become layout algebra with constant offsets.
inline bool transfer_by_threads()
{
   static const bool v = [] {
      const char* e = std::getenv( "HYTEG_HIP_P2_TRANSFER_ROWS" );
      return !( e && e[0] == '1' );
   }();
   return v;
}

} // namespace

// HYTEG_HIP_P2_TRANSFER_INTERLEAVE=1: the kinds of one region follow each other on one XCD instead of grid ( blocks, 8 ), kind by kind.
// Measured (profiles/r03_p2_transfer_rows.txt): the HBM reads of the restriction fall from 95 to 70 MB, but the launch takes 101 us
// instead of 56 (prolongation 48.5 instead of 44.0) -- the kinds' very different term counts no longer balance over the XCDs.  Off.
static bool kinds_interleaved()
{
   static const bool v = [] {
      const char* e = std::getenv( "HYTEG_HIP_P2_TRANSFER_INTERLEAVE" );
      return e && e[0] == '1';
   }();
   return v;
}

extern "C" {

HYTEG_HIP_API int hyteg_hip_p2_prolongate_cell( double*            fine_vertex,
                                                double*            fine_edge,
                                                const double*      coarse_vertex,
                                                const double*      coarse_edge,
                                                int                coarse_level,
                                                int                update,
                                                unsigned           mask,
                                                hyteg_hip_stream_t stream )
{
   HH_REQUIRE( fine_vertex && fine_edge && coarse_vertex && coarse_edge, "p2_prolongate_cell: null pointer" );
   HH_REQUIRE( coarse_level >= 0 && coarse_level + 1 <= 10, "p2_prolongate_cell: coarse level out of range [0,9]" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p2_prolongate_cell: bad update type" );
   P2TransferArgs A{};
   int            rc = get_tables( &A.T );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   A.dstV = fine_vertex, A.dstE = fine_edge, A.srcV = coarse_vertex, A.srcE = coarse_edge;
   A.Nc = ( 1 << coarse_level ) + 1, A.Nf = ( 1 << ( coarse_level + 1 ) ) + 1;
   A.update = update, A.mask = mask;
   if ( !transfer_by_threads() && coarse_level + 1 <= 9 )
   {
      TileTable tt;
      rc = get_tiles( coarse_level + 1, TILES_ROWS, 128, &tt );
      if ( rc != HYTEG_HIP_OK )
         return rc;
      hipLaunchKernelGGL( p2_prolongate_rows_kernel, dim3( (unsigned) ( ( tt.count + 3 ) / 4 ), 8 ), dim3( 256 ), 0, as_stream( stream ), A, tt.dev, tt.count );
      HH_CHECK_HIP( hipGetLastError() );
      return HYTEG_HIP_OK;
   }
   const int64_t most = tet64( A.Nf );
   {
      const unsigned blocks = (unsigned) ( ( most + kThreads - 1 ) / kThreads );
      if ( kinds_interleaved() && blocks >= 16 )
         hipLaunchKernelGGL( p2_prolongate_kernel, dim3( 64u * ( ( blocks + 7 ) / 8 ) ), dim3( kThreads ), 0, as_stream( stream ), A );
      else
         hipLaunchKernelGGL( p2_prolongate_kernel, dim3( blocks, 8 ), dim3( kThreads ), 0, as_stream( stream ), A );
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p2_restrict_cell( double*            coarse_vertex,
                                              double*            coarse_edge,
                                              const double*      fine_vertex,
                                              const double*      fine_edge,
                                              int                coarse_level,
                                              const double*      nnc,
                                              unsigned           mask,
                                              hyteg_hip_stream_t stream )
{
   HH_REQUIRE( fine_vertex && fine_edge && coarse_vertex && coarse_edge && nnc, "p2_restrict_cell: null pointer" );
   HH_REQUIRE( coarse_level >= 0 && coarse_level + 1 <= 10, "p2_restrict_cell: coarse level out of range [0,9]" );
   P2TransferArgs A{};
   int            rc = get_tables( &A.T );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   A.dstV = coarse_vertex, A.dstE = coarse_edge, A.srcV = fine_vertex, A.srcE = fine_edge;
   A.Nc = ( 1 << coarse_level ) + 1, A.Nf = ( 1 << ( coarse_level + 1 ) ) + 1;
   A.mask = mask;
   for ( int k = 0; k < 14; ++k )
   {
      HH_REQUIRE( nnc[k] > 0.0, "p2_restrict_cell: neighbour counts must be positive" );
      A.nncInv.inv[k] = 1.0 / nnc[k];
   }
   if ( !transfer_by_threads() && coarse_level + 1 <= 9 )
   {
      TileTable tt;
      rc = get_tiles( coarse_level, TILES_ROWS, 64, &tt );
      if ( rc != HYTEG_HIP_OK )
         return rc;
      hipLaunchKernelGGL( p2_restrict_rows_kernel, dim3( (unsigned) ( ( tt.count + 3 ) / 4 ), 8 ), dim3( 256 ), 0, as_stream( stream ), A, tt.dev, tt.count );
      HH_CHECK_HIP( hipGetLastError() );
      return HYTEG_HIP_OK;
   }
   const int64_t most = tet64( A.Nc );
   {
      const unsigned blocks = (unsigned) ( ( most + kThreads - 1 ) / kThreads );
      if ( kinds_interleaved() && blocks >= 16 )
         hipLaunchKernelGGL( p2_restrict_kernel, dim3( 64u * ( ( blocks + 7 ) / 8 ) ), dim3( kThreads ), 0, as_stream( stream ), A );
      else
         hipLaunchKernelGGL( p2_restrict_kernel, dim3( blocks, 8 ), dim3( kThreads ), 0, as_stream( stream ), A );
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}
}
