// C-ABI entry points: P1 linear restriction / prolongation on one macro-cell (gather forms).
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

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

__device__ inline double prim_scale( const Nnc14& s, int N, int x, int y, int z )
{
   const int slot = prim_slot( N, x, y, z );
   return slot < 0 ? 1.0 : s.inv[slot];
}

// ---- the cell boundary, organised for waves: a "face row" is a run of <= 64 consecutive points k of row j of face f
// (f = 0: z = 0, (k, j, 0);  1: y = 0, (k, 0, j);  2: x = 0, (0, k, j);  3: x + y + z = N - 1, (k, j, N-1-k-j)), so that face
// and row are wave-uniform and no thread divides or takes a square root to find its point (round 1 enumerated the boundary by one flat index per thread).
// A point on a cell edge or vertex is visited through its lowest-numbered face only.
struct FaceRow
{
   short f, j, k0, cnt;
};
__device__ inline void face_point( int N, int f, int j, int k, int& x, int& y, int& z )
{
   switch ( f )
   {
   case 0: x = k, y = j, z = 0; break;
   case 1: x = k, y = 0, z = j; break;
   case 2: x = 0, y = k, z = j; break;
   default: x = k, y = j, z = N - 1 - k - j; break;
   }
}
__device__ inline int lowest_face( int x, int y, int z ) { return ( z == 0 ) ? 0 : ( y == 0 ) ? 1 : ( x == 0 ) ? 2 : 3; }

constexpr int kNB14c[14][3] = { { -1, 0, 0 }, { -1, 0, 1 }, { -1, 1, -1 }, { -1, 1, 0 }, { 0, -1, 0 },
                                { 0, -1, 1 }, { 0, 0, -1 }, { 0, 0, 1 },   { 0, 1, -1 }, { 0, 1, 0 },
                                { 1, -1, 0 }, { 1, -1, 1 }, { 1, 0, -1 },  { 1, 0, 0 } }; // = kNB14, for compile-time use
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
// the first row-mapped version slower than the tile-mapped one (22.7 vs 15.8 us) -- but by extra workgroups at the FRONT of
// the same launch: face rows (restrict_face: straight-line, 10 loads + centre) and one thread per edge / vertex point
// (restrict_shell, the generic sum).  Round 2, level 8 -> 7: 14.6 us with those points behind the row waves and a skip per
// missing neighbour (the compiler serialised the 15 loads: a 7 us tail at every level), 6.6 us without any boundary point,
// 10.0 us now; consecutive row workgroups are kept on one XCD (they share fine rows): 14.7 -> 13.3 us by itself.
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
// coarse point on the cell boundary: the fine neighbours that exist, each scaled by 1 / numNeighborCells of its primitive.
// Branch-free: all 15 loads are issued before the first addition (a neighbour outside the cell reads entry 0 and contributes
// an exact 0.0) -- with a skip per missing neighbour the compiler serialised the loads, and this path (cell edges and
// vertices only, 6 N threads) was a 7 us chain of dependent round trips at EVERY level.
__device__ inline double restrict_shell( const double* __restrict__ fine, int Nf, int x, int y, int z, const Nnc14& s )
{
   double v[14], sc[14];
#pragma unroll
   for ( int k = 0; k < 14; ++k )
   {
      const int  fx = 2 * x + kNB14c[k][0], fy = 2 * y + kNB14c[k][1], fz = 2 * z + kNB14c[k][2];
      const bool in = fx >= 0 && fy >= 0 && fz >= 0 && fx + fy + fz <= Nf - 1;
      sc[k]         = in ? prim_scale( s, Nf, fx, fy, fz ) * 0.5 : 0.0;
      v[k]          = fine[in ? cell_index( Nf, fx, fy, fz ) : 0];
   }
   // products and sums rounded separately (mul_rn / add_rn of common.hpp: no FMA contraction), here and in the batched kernels
   // (p1_batch.hip): the same bits as the reference's scalar loops
   const double c = mul_rn( prim_scale( s, Nf, 2 * x, 2 * y, 2 * z ), fine[cell_index( Nf, 2 * x, 2 * y, 2 * z )] );
   double       acc = 0.0;
#pragma unroll
   for ( int k = 0; k < 14; ++k )
      acc = add_rn( acc, sc[k] != 0.0 ? mul_rn( sc[k], v[k] ) : 0.0 );
   return add_rn( acc, c );
}

// coarse point in the interior of face F (not on a cell edge): its fine neighbours are either in the face (scaled by the
// face's 1 / numNeighborCells), one layer inside the cell (unscaled) or outside (skipped) -- which, is a property of the
// offset and the face alone, so the sum is straight-line code: 10 loads + the centre, same terms in the same order as
// restrict_shell
template < int F >
__device__ inline double restrict_face( const double* __restrict__ fine, int Nf, int x, int y, int z, double inv_f )
{
   double acc = 0.0;
#pragma unroll
   for ( int k = 0; k < 14; ++k )
   {
      constexpr auto inward = []( int kk ) {
         const int dx = kNB14c[kk][0], dy = kNB14c[kk][1], dz = kNB14c[kk][2];
         return F == 0 ? dz : F == 1 ? dy : F == 2 ? dx : -( dx + dy + dz );
      };
      const int c = inward( k );
      if ( c < 0 )
         continue;
      const double v    = fine[cell_index( Nf, 2 * x + kNB14c[k][0], 2 * y + kNB14c[k][1], 2 * z + kNB14c[k][2] )];
      const double term = mul_rn( ( c == 0 ? inv_f : 1.0 ) * 0.5, v );
      acc               = add_rn( acc, term );
   }
   return add_rn( acc, mul_rn( inv_f, fine[cell_index( Nf, 2 * x, 2 * y, 2 * z )] ) );
}

// the points of the six cell edges (t = 0 .. N-1 along each; the four vertices are visited three times with equal results)
__device__ inline void edge_point( int N, int e, int t, int& x, int& y, int& z )
{
   const int n = N - 1;
   switch ( e )
   {
   case 0: x = t, y = 0, z = 0; break;
   case 1: x = 0, y = t, z = 0; break;
   case 2: x = t, y = n - t, z = 0; break;
   case 3: x = 0, y = 0, z = t; break;
   case 4: x = t, y = 0, z = n - t; break;
   default: x = 0, y = t, z = n - t; break;
   }
}

__global__ __launch_bounds__( 64 * kRestrictWaves ) void p1_restrict_kernel( double* __restrict__ coarse,
                                                                              const double* __restrict__ fine,
                                                                              const Tile* tiles,
                                                                              int         ntiles,
                                                                              const FaceRow* faceRows,
                                                                              int         nfaceRows,
                                                                              int         Nc,
                                                                              int         edgeBlocks,
                                                                              int         shellBlocks, // edge + face-row workgroups
                                                                              int         xcd_chunk,
                                                                              unsigned    mask,
                                                                              const Nnc14 s )
{
   const int Nf = 2 * Nc - 1;
   // the boundary points come FIRST in the grid (behind the row waves they were the tail of the launch: 6 of 12.6 us at
   // level 8 -> 7): edge and vertex points, one thread each, through the generic range-checked sum (~500 instructions);
   // then the face rows
   if ( (int) blockIdx.x < edgeBlocks )
   {
      const int q = (int) blockIdx.x * 64 * kRestrictWaves + (int) threadIdx.x;
      if ( q >= 6 * Nc )
         return;
      int x, y, z;
      edge_point( Nc, q / Nc, q % Nc, x, y, z );
      const int slot = prim_slot( Nc, x, y, z );
      if ( ( mask >> slot ) & 1u )
         coarse[cell_index( Nc, x, y, z )] = restrict_shell( fine, Nf, x, y, z, s );
      return;
   }
   if ( (int) blockIdx.x < shellBlocks )
   {
      const int r = __builtin_amdgcn_readfirstlane( ( (int) blockIdx.x - edgeBlocks ) * kRestrictWaves + ( (int) threadIdx.x >> 6 ) );
      if ( r >= nfaceRows )
         return;
      const FaceRow fr   = faceRows[r];
      const int     lane = threadIdx.x & 63;
      int           x, y, z;
      face_point( Nc, fr.f, fr.j, fr.k0 + lane, x, y, z );
      if ( lane >= fr.cnt || prim_slot( Nc, x, y, z ) != 6 + fr.f || !( ( mask >> ( 6 + fr.f ) ) & 1u ) )
         return; // edge / vertex points: the threads in front
      const double inv_f = s.inv[6 + fr.f];
      double       v;
      switch ( fr.f )
      {
      case 0: v = restrict_face< 0 >( fine, Nf, x, y, z, inv_f ); break;
      case 1: v = restrict_face< 1 >( fine, Nf, x, y, z, inv_f ); break;
      case 2: v = restrict_face< 2 >( fine, Nf, x, y, z, inv_f ); break;
      default: v = restrict_face< 3 >( fine, Nf, x, y, z, inv_f ); break;
      }
      coarse[cell_index( Nc, x, y, z )] = v;
      return;
   }
   // consecutive rows share fine rows: keep them on one XCD (workgroups go round-robin over the 8 XCDs, each with its own L2)
   int b = (int) blockIdx.x - shellBlocks;
   if ( xcd_chunk > 0 )
      b = ( b & 7 ) * xcd_chunk + ( b >> 3 );
   const int t = __builtin_amdgcn_readfirstlane( b * kRestrictWaves + ( threadIdx.x >> 6 ) );
   if ( t >= ntiles || !( ( mask >> 14 ) & 1u ) )
      return;
   const Tile tl   = tiles[t];
   const int  lane = threadIdx.x & 63;
   const int  x    = tl.yb + lane;
   if ( lane >= tl.cnt || prim_slot( Nc, x, tl.ya, tl.z ) >= 0 )
      return; // boundary points: the workgroups in front
   coarse[tl.a + lane] = restrict_inner( fine, Nf, x, tl.ya, tl.z );
}

// prolongation: one thread per fine entry (FULL tiles of the fine level).  A fine point with all-even
// coordinates copies its coarse twin; any other fine point is the midpoint of exactly one of the 7
// stencil axes, selected by its parity pattern, and receives half of each of the two end points.
// `lo_first` tells which end point the reference's scatter loop (lexicographic over coarse points)
// would have added first.
__constant__ int kAxis[8][3]  = { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 1, -1, 0 },
                                 { 0, 0, 1 }, { 1, 0, -1 }, { 0, 1, -1 }, { 1, -1, 1 } };
__constant__ int kLoFirst[8] = { 1, 1, 1, 0, 1, 0, 0, 1 };
constexpr bool   kLoFirstC[8] = { 1, 1, 1, 0, 1, 0, 0, 1 }; // the same, for compile-time parities
__device__ inline double zm_ld( __amdgpu_buffer_rsrc_t r, int voff, int soff )
{
   typedef int v2i __attribute__( ( ext_vector_type( 2 ) ) );
   const v2i v = __builtin_amdgcn_raw_buffer_load_b64( r, voff, soff, 2 ); // nontemporal: read once, overwritten
   return __hiloint2double( v.y, v.x );
}

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
         v = add_rn( old[u], mul_rn( sc[u], lo[u] ) ); // the coarse twin (lo == hi)
      else
      {
         const double h  = sc[u] * 0.5;
         const double tl = mul_rn( h, lo[u] ), th = mul_rn( h, hi[u] );
         v               = kLoFirst[code[u]] ? add_rn( add_rn( old[u], tl ), th ) : add_rn( add_rn( old[u], th ), tl );
      }
      if ( on[u] )
         __builtin_nontemporal_store( v, &fine[tl.a + (int) threadIdx.x + u * kThreads] );
   }
}

// ---------------------------------------------------------------------------------------------------------------------
// Prolongation, brick form (fine level >= 4).  The tile kernel above decodes (x, y, z) per fine entry
// (~100 instructions each: 2.9 M entries at level 8 make it instruction-bound, 14 us for 26 MB).  Here a WAVE owns a
// brick of 4 rows x 64 x-positions x 8 slices of INNER fine points -- the same shape as the apply kernel's bricks, rows and
// slices wave-uniform, y0 and z0 odd -- so the parity pattern of every row of the brick, hence the stencil axis and the one
// or two coarse rows it reads, is known at compile time:
//     fine row (y, z), Y = y >> 1, Z = z >> 1:   (y even, z even)  Ra = Rb = row( Y, Z )
//                                                (y odd,  z even)  Ra = row( Y, Z ),     Rb = row( Y + 1, Z )
//                                                (y even, z odd )  Ra = row( Y, Z ),     Rb = row( Y, Z + 1 )
//                                                (y odd,  z odd )  Ra = row( Y, Z + 1 ), Rb = row( Y + 1, Z )
//     even x: lo = Ra[x/2], hi = Rb[x/2];      odd x: lo = Rb[(x-1)/2], hi = Ra[(x+1)/2]
// (kAxis / kLoFirst above, resolved per parity).  A lane loads Ra[x/2], Ra[x/2 + 1] as one 16-byte buffer load and Rb[x/2] as
// an 8-byte one, row bases in the scalar offset; stores are nontemporal.  The task carries the index of the brick's first
// entry in each of its 8 fine slices and of its first coarse row in the 6 coarse slices it touches (64 bytes, one
// s_load_dwordx16); row bases inside a slice are running sums of row lengths.  Lanes outside a row's inner range get an
// out-of-range vector offset: the buffer range check drops their loads and stores.  The fine points on the cell boundary
// (scaled by 1 / numNeighborCells, masked per primitive, zeroed in Add mode) take the generic per-point path in extra
// workgroups at the FRONT of the same launch, so that they overlap the bricks instead of trailing them.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPB_NY = 4, kPB_LZ = 8, kPB_Waves = 4;
struct ProlTask
{
   int fbase[kPB_LZ]; // fine index of (xb, y0, z0 + s)
   int cbase[6];      // coarse index of (xb / 2, y0 >> 1, (z0 >> 1) + q)
   int y0;
   int xb_z0; // xb << 16 | z0
};
static_assert( sizeof( ProlTask ) == 64, "ProlTask must be 64 bytes (one s_load_dwordx16)" );

typedef int tr_v4i __attribute__( ( ext_vector_type( 4 ) ) );
typedef int tr_v2i __attribute__( ( ext_vector_type( 2 ) ) );

// generic path of one fine point (any primitive): the body of p1_prolongate_kernel for a single (x, y, z)
template < int UPDATE >
__device__ inline void prolongate_point( const double* __restrict__ coarse, double* __restrict__ fine, int Nf, int x, int y, int z,
                                         int slot, const Nnc14& s )
{
   const int    Nc   = ( Nf + 1 ) / 2;
   const int    i    = cell_index( Nf, x, y, z );
   const double sc   = slot < 0 ? 1.0 : s.inv[slot];
   const int    code = ( x & 1 ) | ( ( y & 1 ) << 1 ) | ( ( z & 1 ) << 2 );
   const double old  = ( UPDATE == HYTEG_HIP_ADD && slot < 0 ) ? fine[i] : 0.0;
   const int    ex = kAxis[code][0], ey = kAxis[code][1], ez = kAxis[code][2];
   const double lo = coarse[cell_index( Nc, ( x - ex ) >> 1, ( y - ey ) >> 1, ( z - ez ) >> 1 )];
   const double hi = coarse[cell_index( Nc, ( x + ex ) >> 1, ( y + ey ) >> 1, ( z + ez ) >> 1 )];
   double       v;
   if ( code == 0 )
      v = add_rn( old, mul_rn( sc, lo ) );
   else
   {
      const double h  = sc * 0.5;
      const double tl = mul_rn( h, lo ), th = mul_rn( h, hi );
      v               = kLoFirst[code] ? add_rn( add_rn( old, tl ), th ) : add_rn( add_rn( old, th ), tl );
   }
   fine[i] = v;
}

template < int UPDATE, int PFD >
__global__ __launch_bounds__( 64 * kPB_Waves ) void p1_prolongate_brick_kernel( const double* __restrict__ coarse,
                                                                                 double* __restrict__ fine,
                                                                                 const ProlTask* tasks,
                                                                                 int         ntasks,
                                                                                 const FaceRow* faceRows,
                                                                                 int         nfaceRows,
                                                                                 int         shellBlocks,
                                                                                 int         brickBlocks,
                                                                                 int         xcd_chunk,
                                                                                 int         Nf,
                                                                                 unsigned    mask,
                                                                                 const Nnc14 sN )
{
   if ( (int) blockIdx.x >= brickBlocks )
   {
      const int r = __builtin_amdgcn_readfirstlane( ( (int) blockIdx.x - brickBlocks ) * kPB_Waves + ( (int) threadIdx.x >> 6 ) );
      if ( r >= nfaceRows )
         return;
      const FaceRow fr   = faceRows[r];
      const int     lane = threadIdx.x & 63;
      int           x, y, z;
      face_point( Nf, fr.f, fr.j, fr.k0 + lane, x, y, z );
      if ( lane >= fr.cnt || lowest_face( x, y, z ) != fr.f )
         return;
      if ( fr.f >= 2 && y >= 1 && z >= 1 && y + z <= Nf - 3 )
         return; // first / last point of a brick row
      const int slot = prim_slot( Nf, x, y, z );
      if ( ( mask >> slot ) & 1u )
         prolongate_point< UPDATE >( coarse, fine, Nf, x, y, z, slot, sN );
      return;
   }
   // consecutive bricks (neighbours in memory) on the same XCD: the workgroups of a launch go round-robin over the 8 XCDs
   int b = blockIdx.x;
   if ( xcd_chunk > 0 )
      b = ( blockIdx.x & 7 ) * xcd_chunk + ( blockIdx.x >> 3 );
   const int task = __builtin_amdgcn_readfirstlane( b * kPB_Waves + ( (int) threadIdx.x >> 6 ) );
   // the rows of a brick begin with a point of face 2 (x = 0) and end with one of face 3 (x + y + z = N - 1), neither on a
   // cell edge: the brick writes them too (slots 8 and 9) -- as separate 8-byte stores they were 66 k partial cache lines
   const bool m14 = ( mask >> 14 ) & 1u, m8 = ( mask >> 8 ) & 1u, m9 = ( mask >> 9 ) & 1u;
   if ( task >= ntasks || !( m14 || m8 || m9 ) )
      return;
   const ProlTask t    = tasks[task];
   const int      lane = threadIdx.x & 63;
   const int      xb = t.xb_z0 >> 16, z0 = t.xb_z0 & 0xffff, y0 = t.y0;
   const int      Nc = ( Nf + 1 ) / 2;
   const int      Y0 = y0 >> 1, Z0 = z0 >> 1;
   const unsigned fbytes = (unsigned) ( tet64( Nf ) * 8 ), cbytes = (unsigned) ( tet64( Nc ) * 8 );
   const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc( const_cast< double* >( coarse ), 0, cbytes, 0x00020000 );
   const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc( fine, 0, fbytes, 0x00020000 );
   const double inv8 = sN.inv[8], inv9 = sN.inv[9];
   const int  nz   = min( kPB_LZ, Nf - 3 - z0 );
   const bool odd  = lane & 1;
   const bool isF2 = xb + lane == 0;

   // coarse row (Y0 + jy) of coarse slice Z0 + q: running sums of row lengths from the task's per-slice bases
   auto crow = [&]( int q, int jy ) {
      const int W = Nc - ( Z0 + q ) - Y0; // length of row Y0 of that slice
      return t.cbase[q] + jy * W - ( jy * ( jy - 1 ) ) / 2;
   };

   tr_v4i A[kPB_LZ][kPB_NY];
   tr_v2i B[kPB_LZ][kPB_NY];
   double O[kPB_LZ][kPB_NY];
   int    voffF[kPB_LZ][kPB_NY]; // lane * 8, or out of range

   auto load_slice = [&]( auto sc ) {
      constexpr int s  = decltype( sc )::value;
      constexpr int pz = ( 1 + s ) & 1, q = ( 1 + s ) >> 1; // z0 is odd
      const int     Wf = Nf - ( z0 + s ) - y0;              // length of fine row y0 of this slice
#pragma unroll
      for ( int j = 0; j < kPB_NY; ++j )
      {
         const int py = ( 1 + j ) & 1, jy = ( 1 + j ) >> 1; // y0 is odd
         // x of the row: 0 .. Nf - 1 - y - z, first and last on faces 2 and 3 (rows without inner points are not brick rows)
         const int  hi_l   = Nf - 1 - ( y0 + j ) - ( z0 + s ) - xb;
         const bool on     = isF2 ? m8 : ( lane == hi_l ? m9 : m14 );
         const bool active = s < nz && hi_l >= 2 - xb && lane <= hi_l && on;
         const int  ra = ( py && pz ) ? crow( q + 1, jy ) : crow( q, jy );
         const int  rb = py ? crow( q, jy + 1 ) : ( pz ? crow( q + 1, jy ) : crow( q, jy ) );
         const int  vc = active ? ( lane >> 1 ) * 8 : -16;
         A[s][j]       = __builtin_amdgcn_raw_buffer_load_b128( rc, vc, ra * 8, 0 );
         B[s][j]       = __builtin_amdgcn_raw_buffer_load_b64( rc, vc, rb * 8, 0 );
         voffF[s][j]   = active ? lane * 8 : -8;
         if constexpr ( UPDATE == HYTEG_HIP_ADD )
         {
            const int fr = t.fbase[s] + j * Wf - ( j * ( j - 1 ) ) / 2;
            O[s][j]      = zm_ld( rf, voffF[s][j], fr * 8 );
         }
      }
   };
   auto store_slice = [&]( auto sc ) {
      constexpr int s  = decltype( sc )::value;
      constexpr int pz = ( 1 + s ) & 1;
      const int     Wf = Nf - ( z0 + s ) - y0;
#pragma unroll
      for ( int j = 0; j < kPB_NY; ++j )
      {
         const int    py = ( 1 + j ) & 1;
         const tr_v4i a  = A[s][j];
         const tr_v2i b  = B[s][j];
         const double a0 = __hiloint2double( a.y, a.x ), a1 = __hiloint2double( a.w, a.z ), b0 = __hiloint2double( b.y, b.x );
         const double lo = odd ? b0 : a0, hi = odd ? a1 : b0;
         const int    hi_l = Nf - 1 - ( y0 + j ) - ( z0 + s ) - xb;
         const bool   bnd  = isF2 || lane == hi_l;
         const double sc   = isF2 ? inv8 : ( lane == hi_l ? inv9 : 1.0 );
         // Add keeps the old value of inner points only (the boundary is zeroed first, P1toP1LinearProlongation.cpp:214-238)
         const double old = ( UPDATE == HYTEG_HIP_ADD && !bnd ) ? O[s][j] : 0.0;
         // kLoFirst[ px | py << 1 | pz << 2 ]
         const bool lf_even = kLoFirstC[( py << 1 ) | ( pz << 2 )], lf_odd = kLoFirstC[1 | ( py << 1 ) | ( pz << 2 )];
         const bool lf      = odd ? lf_odd : lf_even;
         const double first = lf ? lo : hi, second = lf ? hi : lo;
         const double h     = sc * 0.5;
         double       v     = add_rn( add_rn( old, mul_rn( h, first ) ), mul_rn( h, second ) );
         if ( !py && !pz )
            v = odd ? v : add_rn( old, mul_rn( sc, a0 ) ); // the coarse twin
         const int fr = t.fbase[s] + j * Wf - ( j * ( j - 1 ) ) / 2;
         __builtin_amdgcn_raw_buffer_store_b64( tr_v2i{ __double2loint( v ), __double2hiint( v ) }, rf, voffF[s][j], fr * 8, 2 );
      }
   };
   // loads run PFD slices ahead of the stores
   [&]< int... Is >( std::integer_sequence< int, Is... > ) { ( load_slice( std::integral_constant< int, Is >{} ), ... ); }
   ( std::make_integer_sequence < int, PFD < kPB_LZ ? PFD : kPB_LZ > {} );
   [&]< int... Is >( std::integer_sequence< int, Is... > ) {
      ( ( [&] {
           if constexpr ( Is + PFD < kPB_LZ )
              load_slice( std::integral_constant< int, Is + PFD >{} );
           store_slice( std::integral_constant< int, Is >{} );
        }() ),
        ... );
   }
   ( std::make_integer_sequence< int, kPB_LZ >{} );
}

struct ProlTable
{
   const ProlTask* dev   = nullptr;
   int             count = 0;
};
int get_prolongation_bricks( int fine_level, ProlTable* out )
{
   static std::mutex                                     mtx;
   static std::map< std::pair< int, int >, ProlTable > cache;
   int                                                   dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   std::lock_guard< std::mutex > lock( mtx );
   auto                          key = std::make_pair( dev, fine_level );
   auto                          it  = cache.find( key );
   if ( it == cache.end() )
   {
      const int               Nf = ( 1 << fine_level ) + 1, Nc = ( Nf + 1 ) / 2;
      std::vector< ProlTask > host;
      for ( int z0 = 1; z0 <= Nf - 4; z0 += kPB_LZ )
         for ( int y0 = 1; y0 <= Nf - 3 - z0; y0 += kPB_NY )
            for ( int xb = 0; xb <= Nf - 1 - y0 - z0; xb += 64 )
            {
               ProlTask t{};
               t.y0    = y0;
               t.xb_z0 = ( xb << 16 ) | z0;
               for ( int s = 0; s < kPB_LZ; ++s )
                  t.fbase[s] = z0 + s <= Nf - 1 - y0 ? cell_index( Nf, xb, y0, z0 + s ) : 0;
               for ( int q = 0; q < 6; ++q )
               {
                  const int Z = ( z0 >> 1 ) + q, Y = y0 >> 1;
                  t.cbase[q]  = ( Z <= Nc - 1 && Y <= Nc - 1 - Z ) ? cell_index( Nc, 0, Y, Z ) + xb / 2 : 0;
               }
               host.push_back( t );
            }
      ProlTable tab;
      tab.count = (int) host.size();
      if ( !host.empty() )
      {
         void* p = nullptr;
         HH_CHECK_HIP( hipMalloc( &p, host.size() * sizeof( ProlTask ) ) );
         HH_CHECK_HIP( hipMemcpy( p, host.data(), host.size() * sizeof( ProlTask ), hipMemcpyHostToDevice ) );
         tab.dev = static_cast< const ProlTask* >( p );
      }
      it = cache.emplace( key, tab ).first;
   }
   *out = it->second;
   return HYTEG_HIP_OK;
}

struct FaceRowTable
{
   const FaceRow* dev   = nullptr;
   int            count = 0;
};
int get_face_rows( int level, FaceRowTable* out )
{
   static std::mutex                                        mtx;
   static std::map< std::pair< int, int >, FaceRowTable > cache;
   int                                                      dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   std::lock_guard< std::mutex > lock( mtx );
   auto                          key = std::make_pair( dev, level );
   auto                          it  = cache.find( key );
   if ( it == cache.end() )
   {
      const int              N = ( 1 << level ) + 1;
      std::vector< FaceRow > host;
      for ( int f = 0; f < 4; ++f )
         for ( int j = 0; j < N; ++j )
            for ( int k0 = 0; k0 < N - j; k0 += 64 )
               host.push_back( FaceRow{ (short) f, (short) j, (short) k0, (short) std::min( 64, N - j - k0 ) } );
      FaceRowTable tab;
      tab.count = (int) host.size();
      void* p   = nullptr;
      HH_CHECK_HIP( hipMalloc( &p, host.size() * sizeof( FaceRow ) ) );
      HH_CHECK_HIP( hipMemcpy( p, host.data(), host.size() * sizeof( FaceRow ), hipMemcpyHostToDevice ) );
      tab.dev = static_cast< const FaceRow* >( p );
      it      = cache.emplace( key, tab ).first;
   }
   *out = it->second;
   return HYTEG_HIP_OK;
}

// measurement switch: HYTEG_HIP_TRANSFER_TILES=1 keeps the round-1 tile kernels
bool transfer_use_tiles()
{
   static const bool v = [] {
      const char* e = std::getenv( "HYTEG_HIP_TRANSFER_TILES" );
      return e && e[0] == '1';
   }();
   return v;
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
   const int        Ncw       = ( 1 << coarse_level ) + 1;
   const int        xcd_chunk = ( ( tt.count + kRestrictWaves - 1 ) / kRestrictWaves + 7 ) / 8;
   const int        rowBlocks = 8 * xcd_chunk;
   FaceRowTable ft;
   rc = get_face_rows( coarse_level, &ft );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   const int edgeBlocks  = ( 6 * Ncw + 64 * kRestrictWaves - 1 ) / ( 64 * kRestrictWaves );
   const int shellBlocks = edgeBlocks + ( ft.count + kRestrictWaves - 1 ) / kRestrictWaves;
   hipLaunchKernelGGL( p1_restrict_kernel,
                       dim3( rowBlocks + shellBlocks ),
                       dim3( 64 * kRestrictWaves ),
                       0,
                       as_stream( stream ),
                       coarse,
                       fine,
                       tt.dev,
                       tt.count,
                       ft.dev,
                       ft.count,
                       ( 1 << coarse_level ) + 1,
                       edgeBlocks,
                       shellBlocks,
                       xcd_chunk,
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

HYTEG_HIP_API int hyteg_hip_p1_prolongate_cell_masked_update( const double*      coarse,
                                                              double*            fine,
                                                              int                coarse_level,
                                                              const double*      nnc,
                                                              unsigned           mask,
                                                              int                update,
                                                              hyteg_hip_stream_t stream )
{
   return prolongate_impl( coarse, fine, coarse_level, nnc, update, mask, stream );
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
   const int Nf = ( 1 << ( coarse_level + 1 ) ) + 1;
   // level 11: an 11.5 GB array is beyond 32-bit buffer offsets and int indices -> the tile kernel (64-bit pointers)
   if ( coarse_level + 1 >= 4 && coarse_level + 1 <= 10 && !transfer_use_tiles() )
   {
      ProlTable pt;
      int       rcb = get_prolongation_bricks( coarse_level + 1, &pt );
      if ( rcb != HYTEG_HIP_OK )
         return rcb;
      FaceRowTable ft;
      rcb = get_face_rows( coarse_level + 1, &ft );
      if ( rcb != HYTEG_HIP_OK )
         return rcb;
      const int  shellBlocks = ( ft.count + kPB_Waves - 1 ) / kPB_Waves;
      const int  xcd_chunk   = ( ( pt.count + kPB_Waves - 1 ) / kPB_Waves + 7 ) / 8;
      const int  brickBlocks = 8 * xcd_chunk;
      const dim3 grid( shellBlocks + brickBlocks ), block( 64 * kPB_Waves );
      if ( update == HYTEG_HIP_REPLACE )
         hipLaunchKernelGGL( ( p1_prolongate_brick_kernel< HYTEG_HIP_REPLACE, 2 > ), grid, block, 0, as_stream( stream ), coarse, fine, pt.dev,
                             pt.count, ft.dev, ft.count, shellBlocks, brickBlocks, xcd_chunk, Nf, mask, s );
      else
         hipLaunchKernelGGL( ( p1_prolongate_brick_kernel< HYTEG_HIP_ADD, 2 > ), grid, block, 0, as_stream( stream ), coarse, fine, pt.dev,
                             pt.count, ft.dev, ft.count, shellBlocks, brickBlocks, xcd_chunk, Nf, mask, s );
      HH_CHECK_HIP( hipGetLastError() );
      return HYTEG_HIP_OK;
   }
   TileTable tt;
   int       rc = get_tiles( coarse_level + 1, TILES_FULL, kTile, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
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
