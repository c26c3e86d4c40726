// Shared host/device helpers of libhyteg_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/hyteg_hip.h"

namespace hyteg_hip {

// ---- error plumbing -------------------------------------------------------------------------
void set_error( const std::string& msg );
int  fail( int code, const std::string& msg );

#define HH_CHECK_HIP( expr )                                                                                         \
   do                                                                                                                \
   {                                                                                                                 \
      hipError_t _e = ( expr );                                                                                      \
      if ( _e != hipSuccess )                                                                                        \
         return ::hyteg_hip::fail( _e == hipErrorOutOfMemory ? HYTEG_HIP_ENOMEM : HYTEG_HIP_ELAUNCH,                 \
                                   std::string( #expr ) + ": " + hipGetErrorString( _e ) );                          \
   } while ( 0 )

#define HH_REQUIRE( cond, msg )                                         \
   do                                                                   \
   {                                                                    \
      if ( !( cond ) )                                                  \
         return ::hyteg_hip::fail( HYTEG_HIP_EINVAL, std::string( msg ) ); \
   } while ( 0 )

inline bool level_ok( int level ) { return level >= HYTEG_HIP_MIN_LEVEL && level <= HYTEG_HIP_MAX_LEVEL; }

// ---- HyTeG macro-cell layout (src/hyteg/indexing/MacroCellIndexing.hpp:40-52) ----------------
// a product / a sum rounded on its own: the compiler contracts a * b + c into an FMA wherever it sees one (-ffp-contract=fast
// is hipcc's default, and HIP's __dmul_rn / __dadd_rn are plain operators that contract as well).  Kernels that promise the
// bits of the reference's scalar loops, or of each other, build their sums from these two.
__host__ __device__ inline double mul_rn( double a, double b )
{
#pragma clang fp contract( off )
   return a * b;
}
__host__ __device__ inline double add_rn( double a, double b )
{
#pragma clang fp contract( off )
   return a + b;
}
__host__ __device__ inline int tri( int w ) { return ( w * ( w + 1 ) ) / 2; }
__host__ __device__ inline int64_t tet64( int64_t w ) { return ( w * ( w + 1 ) * ( w + 2 ) ) / 6; }
// start of slice z in a cell array of width N
// tet(w) in 32-bit arithmetic: (w(w+1)/2)(w+2) = 3 tet(w) stays below 2^32 for w <= 1290 (levels <= 10)
__host__ __device__ inline unsigned tet32( unsigned w ) { return ( ( w * ( w + 1u ) ) >> 1 ) * ( w + 2u ) / 3u; }
__host__ __device__ inline int slice_start( int N, int z )
{
   if ( N <= 1290 )
      return (int) ( tet32( (unsigned) N ) - tet32( (unsigned) ( N - z ) ) );
   return (int) ( tet64( N ) - tet64( N - z ) );
}
// start of row y inside a slice whose row 0 has length W
__host__ __device__ inline int row_start( int W, int y ) { return y * W - ( y * ( y - 1 ) ) / 2; }
__host__ __device__ inline int cell_index( int N, int x, int y, int z )
{
   return slice_start( N, z ) + row_start( N - z, y ) + x;
}

// Row of the slice-local offset j in a slice whose row 0 has length W: largest y with row_start(W,y) <= j.
__device__ inline int row_of( int W, int j )
{
   const float b    = (float) ( 2 * W + 1 );
   const float disc = b * b - 8.0f * (float) j;
   int         y    = (int) ( ( b - __builtin_sqrtf( disc > 0.0f ? disc : 0.0f ) ) * 0.5f );
   y                = y < 0 ? 0 : ( y > W - 1 ? W - 1 : y );
   while ( row_start( W, y ) > j )
      --y;
   while ( y + 1 < W && row_start( W, y + 1 ) <= j )
      ++y;
   return y;
}

// ---- tile tables ------------------------------------------------------------------------------
// A tile is a run of `cnt` consecutive array entries inside ONE z-slice.  Kernels map one workgroup
// to one tile; because the layout is linear, everything a tile's stencils touch is three contiguous
// spans of the source array (slices z-1, z, z+1).
struct Tile
{
   int a;   // global index of the first entry
   int cnt; // number of entries (<= tile capacity)
   int z;   // slice
   int ya;  // row of the first entry
   int yb;  // row of the last entry
   int pad[3];
};
static_assert( sizeof( Tile ) == 32, "Tile must be 32 bytes" );

enum TileKind
{
   TILES_INNER = 0, // cover rows 1..W-3 of slices 1..N-3 (the range the interior kernels loop over)
   TILES_FULL  = 1, // cover every entry of the array
   TILES_ROWS  = 2  // one tile = up to `capacity` consecutive entries of ONE row: a = index of (x0, y, z), ya = y, yb = x0
};

struct TileTable
{
   const Tile* dev   = nullptr;
   int         count = 0;
};

// Returns the (cached, device-resident) tile table for (current device, level, kind, capacity).
// First use allocates and uploads synchronously; later uses are lookup only.
int get_tiles( int level, TileKind kind, int capacity, TileTable* out );

// Brick task tables of the z-march apply kernel (kernels_apply_zmarch.hpp), cached like tile tables.
struct BrickTask;
struct BrickTable
{
   const BrickTask* dev   = nullptr;
   int              count = 0;
   bool             decodable = false; // z-chunk starts fit the kernel arguments (decode mode of the z-march kernel)
   int              zs[32]    = {};    // first task of z-chunk k, padded with count
};
int get_bricks( int level, int NY, int LZ, BrickTable* out, int XS = 62 ); // XS: x-stride of the bricks (62, or 56 = aligned windows)

// ticket counter of the single-launch reductions (p1_batch.hip), one per (device, stream); zero between launches
int dot_counter( hipStream_t stream, unsigned** out );

inline hipStream_t as_stream( hyteg_hip_stream_t s ) { return reinterpret_cast< hipStream_t >( s ); }

enum ApplyMode
{
   APPLY_REPLACE = 0,
   APPLY_ADD     = 1,
   APPLY_JACOBI  = 2,
   APPLY_RESIDUAL = 3, // dst = rhs - A src
   // the two fused steps of the mixed-precision Jacobi smoother (double arrays in, float arrays out, and back):
   APPLY_RESIDUAL_F32OUT = 4, // double arithmetic: r = rhs - A src; dst (float) = r, dst2 (float) = relax * r / centre  (= the first Jacobi
                              //   sweep on A e = r from e = 0)
   APPLY_JACOBI_ACCUM = 5     // float arithmetic: e = src + relax * ( rhs - A src ) / centre (src, rhs float); xacc (double) += e; e is not stored
};

struct Stencil15
{
   double w[15];
};
struct Nnc14
{
   double inv[14]; // 1 / numNeighborCells: edge0..5, face0..3, vertex0..3
};

} // namespace hyteg_hip
