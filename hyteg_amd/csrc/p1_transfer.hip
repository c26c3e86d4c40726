// C-ABI entry points: P1 linear restriction / prolongation on one macro-cell (gather forms).
#include "common.hpp"

using namespace hyteg_hip;

namespace {


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

// Enumerates every point on the cell boundary exactly once: q in [0, 4 tri(N)) -> (x,y,z,slot); false for the duplicates (a
// point on a cell edge / vertex is visited through its lowest-numbered face only) and for padding.
__device__ inline bool shell_point( int N, int q, int& x, int& y, int& z, int& slot )
{
   const int T = tri( N );
   if ( q >= 4 * T )
      return false;
   const int f = q / T, r = q - f * T;
   const int j = row_of( N, r );
   const int k = r - row_start( N, j );
   switch ( f )
   {
   case 0: x = k, y = j, z = 0; break;
   case 1: x = k, y = 0, z = j; break;
   case 2: x = 0, y = k, z = j; break;
   default: x = k, y = j, z = N - 1 - k - j; break;
   }
   const int lowest = ( z == 0 ) ? 0 : ( y == 0 ) ? 1 : ( x == 0 ) ? 2 : 3;
   if ( lowest != f )
      return false;
   slot = prim_slot( N, x, y, z );
   return true;
}

__device__ inline double prim_scale( const Nnc14& s, int N, int x, int y, int z )
{
   const int slot = prim_slot( N, x, y, z );
   return slot < 0 ? 1.0 : s.inv[slot];
}

__constant__ int kNB14[14][3] = { { -1, 0, 0 }, { -1, 0, 1 }, { -1, 1, -1 }, { -1, 1, 0 }, { 0, -1, 0 },
                                  { 0, -1, 1 }, { 0, 0, -1 }, { 0, 0, 1 },   { 0, 1, -1 }, { 0, 1, 0 },
                                  { 1, -1, 0 }, { 1, -1, 1 }, { 1, 0, -1 },  { 1, 0, 0 } };

// restriction: one WAVE per run of 64 consecutive coarse entries of one row (TILES_ROWS of the coarse level), four waves per
// workgroup.  Row and slice are wave-uniform (no per-thread index decoding), and for an inner coarse point (the bulk: only
// inner fine neighbours, no scaling, no range checks) the 15 fine values come from 7 fine rows as 7 16-byte loads + one
// 8-byte load per lane -- rows that contribute the offsets (-1, 0) are loaded at 2x - 1, rows that contribute (0, +1) at
// 2x -- which a wave issues as contiguous 1 KiB requests (the first version issued 15 8-byte loads with stride 16 per
// thread and decoded (x, y, z) per point: 15.8 us for level 8 -> 7).  Summation order = the 14 neighbours in kNB14 order,
// then the centre, as before.  Coarse points on the cell boundary (scaled, range-checked sums: ~500 instructions) are NOT
// handled by the row waves -- two lanes of every wave would take that path and the other 62 would wait for them, which made
// the first row-mapped version slower than the tile-mapped one (22.7 vs 15.8 us) -- but by extra workgroups at the end of
// the same launch that enumerate the shell points densely.
constexpr int kRestrictRow   = 64;
constexpr int kRestrictWaves = 4;
typedef double tr_d2 __attribute__( ( ext_vector_type( 2 ) ) );
__device__ inline tr_d2 load2( const double* p )
{
   tr_d2 v;
   __builtin_memcpy( &v, p, sizeof( v ) ); // 8-byte aligned 16-byte load
   return v;
}

// inner coarse point: 15 unscaled fine values from 7 rows (y, z wave-uniform when called from a row wave)
__device__ inline double restrict_inner( const double* __restrict__ fine, int Nf, int x, int y, int z )
{
   const int fz = 2 * z, fy = 2 * y, fx = 2 * x;
   const int sm = slice_start( Nf, fz - 1 ), s0 = slice_start( Nf, fz ), sp = slice_start( Nf, fz + 1 );
   const int Wm = Nf - fz + 1, W0 = Nf - fz, Wp = Nf - fz - 1;
   const double* r_0m = fine + sm + row_start( Wm, fy ) + fx;     // ( 0,-1): offsets 0, +1
   const double* r_pm = fine + sm + row_start( Wm, fy + 1 ) + fx; // (+1,-1): -1, 0
   const double* r_m0 = fine + s0 + row_start( W0, fy - 1 ) + fx; // (-1, 0): 0, +1
   const double* r_00 = fine + s0 + row_start( W0, fy ) + fx;     // ( 0, 0): -1, 0, +1
   const double* r_p0 = fine + s0 + row_start( W0, fy + 1 ) + fx; // (+1, 0): -1, 0
   const double* r_mp = fine + sp + row_start( Wp, fy - 1 ) + fx; // (-1,+1): 0, +1
   const double* r_0p = fine + sp + row_start( Wp, fy ) + fx;     // ( 0,+1): -1, 0
   const tr_d2  a00 = load2( r_00 - 1 ), a0p = load2( r_0p - 1 ), apm = load2( r_pm - 1 ), ap0 = load2( r_p0 - 1 );
   const tr_d2  am0 = load2( r_m0 ), amp = load2( r_mp ), a0m = load2( r_0m );
   const double e00 = r_00[1];
   // kNB14 order: (-1,0,0) (-1,0,1) (-1,1,-1) (-1,1,0) (0,-1,0) (0,-1,1) (0,0,-1) (0,0,1) (0,1,-1) (0,1,0) (1,-1,0) (1,-1,1) (1,0,-1) (1,0,0)
   double acc = 0.5 * a00.x;
   acc        = acc + 0.5 * a0p.x;
   acc        = acc + 0.5 * apm.x;
   acc        = acc + 0.5 * ap0.x;
   acc        = acc + 0.5 * am0.x;
   acc        = acc + 0.5 * amp.x;
   acc        = acc + 0.5 * a0m.x;
   acc        = acc + 0.5 * a0p.y;
   acc        = acc + 0.5 * apm.y;
   acc        = acc + 0.5 * ap0.y;
   acc        = acc + 0.5 * am0.y;
   acc        = acc + 0.5 * amp.y;
   acc        = acc + 0.5 * a0m.y;
   acc        = acc + 0.5 * e00;
   return acc + a00.y;
}
// coarse point on the cell boundary: the fine neighbours that exist, each scaled by 1 / numNeighborCells of its primitive
__device__ inline double restrict_shell( const double* __restrict__ fine, int Nf, int x, int y, int z, const Nnc14& s )
{
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
   return first ? term : acc + term;
}

__global__ __launch_bounds__( 64 * kRestrictWaves ) void p1_restrict_kernel( double* __restrict__ coarse,
                                                                              const double* __restrict__ fine,
                                                                              const Tile* tiles,
                                                                              int         ntiles,
                                                                              int         Nc,
                                                                              unsigned    mask,
                                                                              const Nnc14 s )
{
   const int Nf        = 2 * Nc - 1;
   const int rowBlocks = ( ntiles + kRestrictWaves - 1 ) / kRestrictWaves;
   if ( (int) blockIdx.x < rowBlocks )
   {
      const int t = __builtin_amdgcn_readfirstlane( blockIdx.x * kRestrictWaves + ( threadIdx.x >> 6 ) );
      if ( t >= ntiles || !( ( mask >> 14 ) & 1u ) )
         return;
      const Tile tl   = tiles[t];
      const int  lane = threadIdx.x & 63;
      const int  x    = tl.yb + lane;
      if ( lane >= tl.cnt || prim_slot( Nc, x, tl.ya, tl.z ) >= 0 )
         return; // shell points: the workgroups behind the row waves
      coarse[tl.a + lane] = restrict_inner( fine, Nf, x, tl.ya, tl.z );
      return;
   }
   const int q = ( (int) blockIdx.x - rowBlocks ) * 64 * kRestrictWaves + (int) threadIdx.x;
   int       x, y, z, slot;
   if ( !shell_point( Nc, q, x, y, z, slot ) || !( ( mask >> slot ) & 1u ) )
      return;
   coarse[cell_index( Nc, x, y, z )] = restrict_shell( fine, Nf, x, y, z, s );
}

// prolongation: one thread per fine entry (FULL tiles of the fine level).  A fine point with all-even
// coordinates copies its coarse twin; any other fine point is the midpoint of exactly one of the 7
// stencil axes, selected by its parity pattern, and receives half of each of the two end points.
// `lo_first` tells which end point the reference's scatter loop (lexicographic over coarse points)
// would have added first.
__constant__ int kAxis[8][3]  = { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 1, -1, 0 },
                                 { 0, 0, 1 }, { 1, 0, -1 }, { 0, 1, -1 }, { 1, -1, 1 } };
__constant__ int kLoFirst[8] = { 1, 1, 1, 0, 1, 0, 0, 1 };

// (A row-mapped form -- one wave per 128 consecutive fine entries of a row, wave-uniform coarse rows, 16-byte stores -- was
// measured SLOWER than this tile-mapped one, 17.3 vs 14.0 us for level 7 -> 8: with two outputs per lane the per-wave
// scalar work, four row bases and a table entry, outweighs the per-thread index decoding it saves.)
constexpr int kTile    = 1024;
constexpr int kThreads = 256;
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
   int       rc = get_tiles( coarse_level, TILES_ROWS, kRestrictRow, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   const int Ncw         = ( 1 << coarse_level ) + 1;
   const int rowBlocks   = ( tt.count + kRestrictWaves - 1 ) / kRestrictWaves;
   const int shellBlocks = ( 4 * tri( Ncw ) + 64 * kRestrictWaves - 1 ) / ( 64 * kRestrictWaves );
   hipLaunchKernelGGL( p1_restrict_kernel,
                       dim3( rowBlocks + shellBlocks ),
                       dim3( 64 * kRestrictWaves ),
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
   const int  Nf = ( 1 << ( coarse_level + 1 ) ) + 1;
   const dim3 pgrid( tt.count ), pblock( kThreads );
   if ( update == HYTEG_HIP_REPLACE )
      hipLaunchKernelGGL( ( p1_prolongate_kernel< HYTEG_HIP_REPLACE > ),
                          pgrid,
                          pblock,
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
                          pgrid,
                          pblock,
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
