// C-ABI entry points: P1 linear restriction / prolongation on one macro-cell (gather forms).
#include "common.hpp"

using namespace hyteg_hip;

namespace {

constexpr int kTile    = 1024;
constexpr int kThreads = 256;

// slot in nnc[14] = { edge0..5, face0..3, vertex0..3 } of the macro-primitive the point lies on, or -1
// (src/hyteg/indexing/MacroCellIndexing.cpp:36-91)
__device__ inline int prim_slot( int N, int x, int y, int z )
{
   const int f0 = ( z == 0 ), f1 = ( y == 0 ), f2 = ( x == 0 ), f3 = ( x + y + z == N - 1 );
   const int cnt = f0 + f1 + f2 + f3;
   if ( cnt == 0 )
      return -1;
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

__device__ inline double prim_scale( const Nnc14& s, int N, int x, int y, int z )
{
   const int slot = prim_slot( N, x, y, z );
   return slot < 0 ? 1.0 : s.inv[slot];
}

__constant__ int kNB14[14][3] = { { -1, 0, 0 }, { -1, 0, 1 }, { -1, 1, -1 }, { -1, 1, 0 }, { 0, -1, 0 },
                                  { 0, -1, 1 }, { 0, 0, -1 }, { 0, 0, 1 },   { 0, 1, -1 }, { 0, 1, 0 },
                                  { 1, -1, 0 }, { 1, -1, 1 }, { 1, 0, -1 },  { 1, 0, 0 } };

// restriction: one thread per coarse entry (FULL tiles of the coarse level, kRestrictTile entries per workgroup so
// that even level 7 fills the chip).  Summation order = the 14 neighbours in kNB14 order, then the centre.
// Inner coarse points (the bulk) only have inner fine neighbours: no scaling, no range checks, and the 15 fine
// addresses come from 7 row bases.
constexpr int kRestrictTile = 256;

__global__ __launch_bounds__( kThreads ) void p1_restrict_kernel( double* __restrict__ coarse,
                                                                   const double* __restrict__ fine,
                                                                   const Tile* tiles,
                                                                   int         ntiles,
                                                                   int         Nc,
                                                                   unsigned    mask,
                                                                   const Nnc14 s )
{
   const int t = blockIdx.x;
   if ( t >= ntiles )
      return;
   const Tile tl = tiles[t];
   const int  e  = threadIdx.x;
   if ( e >= tl.cnt )
      return;
   const int Nf = 2 * Nc - 1;
   const int Wc = Nc - tl.z;
   const int z  = tl.z;
   const int i  = tl.a + e;
   const int j  = i - slice_start( Nc, z );
   const int y  = row_of( Wc, j );
   const int x  = j - row_start( Wc, y );
   const int cs = prim_slot( Nc, x, y, z );
   if ( !( ( mask >> ( cs < 0 ? 14 : cs ) ) & 1u ) )
      return;
   if ( cs < 0 )
   {
      const int fz = 2 * z, fy = 2 * y, fx = 2 * x;
      const int sm = slice_start( Nf, fz - 1 ), s0 = slice_start( Nf, fz ), sp = slice_start( Nf, fz + 1 );
      const int Wm = Nf - fz + 1, W0 = Nf - fz, Wp = Nf - fz - 1;
      // rows (dy, dz) that occur in kNB14
      const double* r_0m = fine + sm + row_start( Wm, fy ) + fx;     // ( 0,-1)
      const double* r_pm = fine + sm + row_start( Wm, fy + 1 ) + fx; // (+1,-1)
      const double* r_m0 = fine + s0 + row_start( W0, fy - 1 ) + fx; // (-1, 0)
      const double* r_00 = fine + s0 + row_start( W0, fy ) + fx;     // ( 0, 0)
      const double* r_p0 = fine + s0 + row_start( W0, fy + 1 ) + fx; // (+1, 0)
      const double* r_mp = fine + sp + row_start( Wp, fy - 1 ) + fx; // (-1,+1)
      const double* r_0p = fine + sp + row_start( Wp, fy ) + fx;     // ( 0,+1)
      // kNB14 order: (-1,0,0) (-1,0,1) (-1,1,-1) (-1,1,0) (0,-1,0) (0,-1,1) (0,0,-1) (0,0,1) (0,1,-1) (0,1,0) (1,-1,0) (1,-1,1) (1,0,-1) (1,0,0)
      const double v0 = r_00[-1], v1 = r_0p[-1], v2 = r_pm[-1], v3 = r_p0[-1], v4 = r_m0[0], v5 = r_mp[0], v6 = r_0m[0], v7 = r_0p[0],
                   v8 = r_pm[0], v9 = r_p0[0], v10 = r_m0[1], v11 = r_mp[1], v12 = r_0m[1], v13 = r_00[1], vc = r_00[0];
      double acc = 0.5 * v0;
      acc        = acc + 0.5 * v1;
      acc        = acc + 0.5 * v2;
      acc        = acc + 0.5 * v3;
      acc        = acc + 0.5 * v4;
      acc        = acc + 0.5 * v5;
      acc        = acc + 0.5 * v6;
      acc        = acc + 0.5 * v7;
      acc        = acc + 0.5 * v8;
      acc        = acc + 0.5 * v9;
      acc        = acc + 0.5 * v10;
      acc        = acc + 0.5 * v11;
      acc        = acc + 0.5 * v12;
      acc        = acc + 0.5 * v13;
      coarse[i]  = acc + vc;
      return;
   }
   double acc   = 0.0;
   bool   first = true;
#pragma unroll
   for ( int k = 0; k < 14; ++k )
   {
      const int fx = 2 * x + kNB14[k][0], fy = 2 * y + kNB14[k][1], fz = 2 * z + kNB14[k][2];
      if ( fx < 0 || fy < 0 || fz < 0 || fx + fy + fz > Nf - 1 )
         continue;
      const double term = prim_scale( s, Nf, fx, fy, fz ) * 0.5 * fine[cell_index( Nf, fx, fy, fz )];
      acc               = first ? term : acc + term;
      first             = false;
   }
   const double term = prim_scale( s, Nf, 2 * x, 2 * y, 2 * z ) * fine[cell_index( Nf, 2 * x, 2 * y, 2 * z )];
   coarse[i]         = first ? term : acc + term;
}

// prolongation: one thread per fine entry (FULL tiles of the fine level).  A fine point with all-even
// coordinates copies its coarse twin; any other fine point is the midpoint of exactly one of the 7
// stencil axes, selected by its parity pattern, and receives half of each of the two end points.
// `lo_first` tells which end point the reference's scatter loop (lexicographic over coarse points)
// would have added first.
__constant__ int kAxis[8][3]  = { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 1, -1, 0 },
                                 { 0, 0, 1 }, { 1, 0, -1 }, { 0, 1, -1 }, { 1, -1, 1 } };
__constant__ int kLoFirst[8] = { 1, 1, 1, 0, 1, 0, 0, 1 };

template < int UPDATE >
__global__ __launch_bounds__( kThreads ) void p1_prolongate_kernel( const double* __restrict__ coarse,
                                                                     double* __restrict__ fine,
                                                                     const Tile* tiles,
                                                                     int         ntiles,
                                                                     int         Nf,
                                                                     unsigned    mask,
                                                                     const Nnc14 s )
{
   const int t = blockIdx.x;
   if ( t >= ntiles )
      return;
   const Tile tl = tiles[t];
   const int  Nc = ( Nf + 1 ) / 2;
   const int  Wf = Nf - tl.z;
   const int  s0 = slice_start( Nf, tl.z );
   const int  z  = tl.z;
   constexpr int kPer = kTile / kThreads;
   double        lo[kPer], hi[kPer], old[kPer], sc[kPer];
   int           code[kPer];
   bool          on[kPer];
   // all gathers of a thread are issued before the first store
#pragma unroll
   for ( int u = 0; u < kPer; ++u )
   {
      const int e    = (int) threadIdx.x + u * kThreads;
      const int i    = tl.a + ( e < tl.cnt ? e : tl.cnt - 1 );
      const int j    = i - s0;
      const int y    = row_of( Wf, j );
      const int x    = j - row_start( Wf, y );
      const int slot = prim_slot( Nf, x, y, z );
      on[u]          = e < tl.cnt && ( ( mask >> ( slot < 0 ? 14 : slot ) ) & 1u );
      sc[u]          = slot < 0 ? 1.0 : s.inv[slot];
      code[u]        = ( x & 1 ) | ( ( y & 1 ) << 1 ) | ( ( z & 1 ) << 2 );
      // Replace zeroes everything first; Add zeroes only the boundary shell (P1toP1LinearProlongation.cpp:214-238)
      old[u] = ( UPDATE == HYTEG_HIP_ADD && slot < 0 ) ? fine[i] : 0.0;
      const int ex = kAxis[code[u]][0], ey = kAxis[code[u]][1], ez = kAxis[code[u]][2];
      lo[u] = coarse[cell_index( Nc, ( x - ex ) >> 1, ( y - ey ) >> 1, ( z - ez ) >> 1 )];
      hi[u] = coarse[cell_index( Nc, ( x + ex ) >> 1, ( y + ey ) >> 1, ( z + ez ) >> 1 )];
   }
#pragma unroll
   for ( int u = 0; u < kPer; ++u )
   {
      double v;
      if ( code[u] == 0 )
         v = old[u] + sc[u] * lo[u]; // the coarse twin (lo == hi)
      else
      {
         const double h = sc[u] * 0.5;
         v              = kLoFirst[code[u]] ? ( old[u] + h * lo[u] ) + h * hi[u] : ( old[u] + h * hi[u] ) + h * lo[u];
      }
      if ( on[u] )
         __builtin_nontemporal_store( v, &fine[tl.a + (int) threadIdx.x + u * kThreads] );
   }
}

} // namespace

extern "C" {

HYTEG_HIP_API int hyteg_hip_p1_restrict_cell( double*            coarse,
                                              const double*      fine,
                                              int                coarse_level,
                                              const double*      nnc,
                                              hyteg_hip_stream_t stream )
{
   return hyteg_hip_p1_restrict_cell_masked( coarse, fine, coarse_level, nnc, HYTEG_HIP_MASK_ALL, stream );
}

HYTEG_HIP_API int hyteg_hip_p1_restrict_cell_masked( double*            coarse,
                                                     const double*      fine,
                                                     int                coarse_level,
                                                     const double*      nnc,
                                                     unsigned           mask,
                                                     hyteg_hip_stream_t stream )
{
   HH_REQUIRE( coarse && fine && nnc, "p1_restrict_cell: null pointer" );
   HH_REQUIRE( coarse_level >= 0 && coarse_level + 1 <= HYTEG_HIP_MAX_LEVEL, "p1_restrict_cell: level out of range" );
   Nnc14 s;
   for ( int k = 0; k < 14; ++k )
   {
      HH_REQUIRE( nnc[k] >= 1.0, "p1_restrict_cell: neighbour-cell counts must be >= 1" );
      s.inv[k] = 1.0 / nnc[k];
   }
   TileTable tt;
   int       rc = get_tiles( coarse_level, TILES_FULL, kRestrictTile, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   hipLaunchKernelGGL( p1_restrict_kernel,
                       dim3( tt.count ),
                       dim3( kThreads ),
                       0,
                       as_stream( stream ),
                       coarse,
                       fine,
                       tt.dev,
                       tt.count,
                       ( 1 << coarse_level ) + 1,
                       mask,
                       s );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

static int prolongate_impl( const double* coarse, double* fine, int coarse_level, const double* nnc, int update, unsigned mask,
                            hyteg_hip_stream_t stream );

HYTEG_HIP_API int hyteg_hip_p1_prolongate_cell( const double*      coarse,
                                                double*            fine,
                                                int                coarse_level,
                                                const double*      nnc,
                                                int                update,
                                                hyteg_hip_stream_t stream )
{
   return prolongate_impl( coarse, fine, coarse_level, nnc, update, HYTEG_HIP_MASK_ALL, stream );
}

HYTEG_HIP_API int hyteg_hip_p1_prolongate_cell_masked( const double*      coarse,
                                                       double*            fine,
                                                       int                coarse_level,
                                                       const double*      nnc,
                                                       unsigned           mask,
                                                       hyteg_hip_stream_t stream )
{
   return prolongate_impl( coarse, fine, coarse_level, nnc, HYTEG_HIP_REPLACE, mask, stream );
}

static int prolongate_impl( const double* coarse, double* fine, int coarse_level, const double* nnc, int update, unsigned mask,
                            hyteg_hip_stream_t stream )
{
   HH_REQUIRE( coarse && fine && nnc, "p1_prolongate_cell: null pointer" );
   HH_REQUIRE( coarse_level >= 0 && coarse_level + 1 <= HYTEG_HIP_MAX_LEVEL, "p1_prolongate_cell: level out of range" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p1_prolongate_cell: bad update type" );
   Nnc14 s;
   for ( int k = 0; k < 14; ++k )
   {
      HH_REQUIRE( nnc[k] >= 1.0, "p1_prolongate_cell: neighbour-cell counts must be >= 1" );
      s.inv[k] = 1.0 / nnc[k];
   }
   TileTable tt;
   int       rc = get_tiles( coarse_level + 1, TILES_FULL, kTile, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   const int Nf = ( 1 << ( coarse_level + 1 ) ) + 1;
   if ( update == HYTEG_HIP_REPLACE )
      hipLaunchKernelGGL( ( p1_prolongate_kernel< HYTEG_HIP_REPLACE > ),
                          dim3( tt.count ),
                          dim3( kThreads ),
                          0,
                          as_stream( stream ),
                          coarse,
                          fine,
                          tt.dev,
                          tt.count,
                          Nf,
                          mask,
                          s );
   else
      hipLaunchKernelGGL( ( p1_prolongate_kernel< HYTEG_HIP_ADD > ),
                          dim3( tt.count ),
                          dim3( kThreads ),
                          0,
                          as_stream( stream ),
                          coarse,
                          fine,
                          tt.dev,
                          tt.count,
                          Nf,
                          mask,
                          s );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}
}
