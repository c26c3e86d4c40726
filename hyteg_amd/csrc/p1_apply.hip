// C-ABI entry points: constant-stencil apply and fused weighted Jacobi on one macro-cell.
#include <cstdlib>
#include <type_traits>

#include "kernels_apply.hpp"
#include "kernels_apply_zmarch.hpp"

using namespace hyteg_hip;

namespace {

constexpr int kTile = 1024;

// Brick shape of the z-march kernel: rows x slices per wave.  Level 8 has only ~64k wave-rows for 1024 SIMDs, so the
// shape trades reuse (taller bricks re-read fewer halo slices) against the number of waves: 4 x 8 is best from
// level 8 on, 4 x 4 below (MI355X: level 7 4.3 vs 5.9 us, level 8 9.9 vs 9.7 us, level 9 equal).
constexpr int kBrickNY = 4;
inline int   brick_lz( int level ) { return level >= 8 ? 8 : 4; }

// developer switches (measurement only).  HYTEG_HIP_APPLY_DECODE=1: bricks decoded from the task index instead of read from
// the table (measured SLOWER: 12.0 vs 10.0 us at level 8 -- ~200 scalar instructions per wave on the CU's one scalar
// unit cost more than the table's round trip; profiles/r02_apply_wave_trace_table_vs_decode.txt).
// HYTEG_HIP_APPLY_PFD=1: loads run one slice ahead of the arithmetic instead of two (default 2: 9.45 vs 9.6-9.7 us at level 8
// in three A/B pairs of bench.py on one box, gpurun_out r02e; round 1 had measured no difference).
inline bool apply_decode_enabled()
{
   static const bool on = [] {
      const char* e = getenv( "HYTEG_HIP_APPLY_DECODE" );
      return e && e[0] == '1';
   }();
   return on;
}
inline int apply_prefetch_distance()
{
   static const int pfd = [] {
      const char* e = getenv( "HYTEG_HIP_APPLY_PFD" );
      return ( e && e[0] == '1' ) ? 1 : 2;
   }();
   return pfd;
}

// HYTEG_HIP_APPLY_PRELOAD=0: the variant whose arguments all travel in the struct instead of the one whose first three are
// preloaded into SGPRs (default on: 9.90-10.18 -> 9.60-9.74 us in four A/B pairs of bench.py on one box)
inline bool apply_preload_enabled()
{
   static const bool v = [] {
      const char* e = getenv( "HYTEG_HIP_APPLY_PRELOAD" );
      return !( e && e[0] == '0' );
   }();
   return v;
}

// HYTEG_HIP_APPLY_ALIGNED=1: bricks 56 outputs wide whose store windows begin on 64-byte boundaries of dst (XS = 56 of
// kernels_apply_zmarch.hpp) instead of 62 outputs wide at x = 1 + 62 k
inline bool apply_aligned_enabled()
{
   static const bool v = [] {
      const char* e = getenv( "HYTEG_HIP_APPLY_ALIGNED" );
      return e && e[0] == '1';
   }();
   return v;
}

template < int MODE, int LZ, typename T = double, int XS = 62 >
int launch_zmarch_lz( T* dst, const T* src, const T* rhs, const T* invdiag, int level, const double* w, double relax, hipStream_t stream )
{
   BrickTable bt;
   int        rc = get_bricks( level, kBrickNY, LZ, &bt, XS );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   if ( bt.count == 0 )
      return HYTEG_HIP_OK;
   ZMarchArgs A{};
   A.dst     = dst;
   A.src     = src;
   A.rhs     = rhs;
   A.invdiag = invdiag;
   A.tasks   = bt.dev;
   A.ntasks  = bt.count;
   A.N       = ( 1 << level ) + 1;
   A.bytes   = (unsigned) ( tet64( A.N ) * (int64_t) sizeof( T ) );
   A.relax   = relax;
   for ( int k = 0; k < 15; ++k )
      A.st.w[k] = w[k];
   static_assert( sizeof( bt.zs ) == sizeof( A.zs ), "z-chunk table sizes" );
   for ( int k = 0; k < kZMarchMaxZChunks; ++k )
      A.zs[k] = bt.zs[k];
   int nblocks = ( bt.count + kZMarchWavesPerBlock - 1 ) / kZMarchWavesPerBlock;
   nblocks     = ( nblocks + 7 ) & ~7;
   A.xcd_chunk = nblocks / 8;
   // dst of Add is read exactly once per element and written right after: nontemporal load (18.6 -> 17.2 us).  rhs /
   // inverse diagonal of Jacobi are re-read by the next sweep of the smoother and stay plain (nontemporal: 12.4 -> 17.6 us
   // when they are still in the Infinity Cache, -2% when they are not).
   constexpr int kExAux = MODE == APPLY_ADD ? 2 : 0; // (the right-hand side of the residual mode is re-read by the cycle: plain)
   const dim3 grid( nblocks ), block( 64 * kZMarchWavesPerBlock );
   if constexpr ( XS != 62 )
      hipLaunchKernelGGL( ( p1_apply_zmarch_preload_kernel< MODE, kBrickNY, LZ, kExAux, false, 2, T, XS > ), grid, block, 0, stream, A.tasks,
                          A.ntasks, A.xcd_chunk, A );
   else if constexpr ( !std::is_same< T, double >::value )
   {
      if ( apply_preload_enabled() )
         hipLaunchKernelGGL( ( p1_apply_zmarch_preload_kernel< MODE, kBrickNY, LZ, kExAux, false, 2, T > ), grid, block, 0, stream, A.tasks,
                             A.ntasks, A.xcd_chunk, A );
      else
         hipLaunchKernelGGL( ( p1_apply_zmarch_kernel< MODE, kBrickNY, LZ, kExAux, false, 2, T > ), grid, block, 0, stream, A );
   }
   else if ( bt.decodable && apply_decode_enabled() )
      hipLaunchKernelGGL( ( p1_apply_zmarch_kernel< MODE, kBrickNY, LZ, kExAux, true > ), grid, block, 0, stream, A );
   else if ( apply_prefetch_distance() == 1 )
      hipLaunchKernelGGL( ( p1_apply_zmarch_kernel< MODE, kBrickNY, LZ, kExAux, false, 1 > ), grid, block, 0, stream, A );
   else if ( apply_preload_enabled() )
      hipLaunchKernelGGL( ( p1_apply_zmarch_preload_kernel< MODE, kBrickNY, LZ, kExAux, false, 2 > ), grid, block, 0, stream, A.tasks, A.ntasks,
                          A.xcd_chunk, A );
   else
      hipLaunchKernelGGL( ( p1_apply_zmarch_kernel< MODE, kBrickNY, LZ, kExAux, false, 2 > ), grid, block, 0, stream, A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

template < int MODE, typename T = double >
int launch_zmarch( T* dst, const T* src, const T* rhs, const T* invdiag, int level, const double* w, double relax, hipStream_t stream )
{
   if constexpr ( MODE == APPLY_REPLACE && std::is_same< T, double >::value )
      if ( brick_lz( level ) == 8 && apply_aligned_enabled() )
         return launch_zmarch_lz< MODE, 8, T, 56 >( dst, src, rhs, invdiag, level, w, relax, stream );
   if ( brick_lz( level ) == 8 )
      return launch_zmarch_lz< MODE, 8, T >( dst, src, rhs, invdiag, level, w, relax, stream );
   return launch_zmarch_lz< MODE, 4, T >( dst, src, rhs, invdiag, level, w, relax, stream );
}

template < int MODE >
int launch_apply( double* dst, const double* src, const double* rhs, const double* invdiag, int level, const double* w,
                  double relax, hipStream_t stream )
{
   // z-march register kernel whenever byte offsets fit the 32-bit buffer addressing (level <= 10);
   // the LDS-tiled kernel (pointer addressing) covers level 11
   if ( tet64( ( 1 << level ) + 1 ) * 8 < ( (int64_t) 1 << 31 ) )
      return launch_zmarch< MODE >( dst, src, rhs, invdiag, level, w, relax, stream );

   TileTable tt;
   int       rc = get_tiles( level, TILES_INNER, kTile, &tt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   if ( tt.count == 0 )
      return HYTEG_HIP_OK;

   ApplyArgs A;
   A.dst     = dst;
   A.src     = src;
   A.rhs     = rhs;
   A.invdiag = invdiag;
   A.tiles   = tt.dev;
   A.ntiles  = tt.count;
   A.N       = ( 1 << level ) + 1;
   A.total   = (int) tet64( A.N );
   A.relax   = relax;
   for ( int k = 0; k < 15; ++k )
      A.st.w[k] = w[k];
   const int nblocks = ( tt.count + 7 ) & ~7;
   A.xcd_chunk       = nblocks / 8;

   const size_t lds_bytes = (size_t) apply_lds_doubles( kTile, A.N ) * sizeof( double );
   const bool   vec       = ( reinterpret_cast< uintptr_t >( src ) & 15 ) == 0;
   auto         kern      = vec ? p1_apply_tiled_kernel< MODE, true > : p1_apply_tiled_kernel< MODE, false >;
   if ( lds_bytes > 48 * 1024 )
      HH_CHECK_HIP( hipFuncSetAttribute( reinterpret_cast< const void* >( kern ),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int) lds_bytes ) );
   hipLaunchKernelGGL( kern, dim3( nblocks ), dim3( kApplyThreads ), lds_bytes, stream, A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

} // namespace

extern "C" {

HYTEG_HIP_API int hyteg_hip_p1_apply_cell( double*            dst,
                                           const double*      src,
                                           int                level,
                                           const double*      w,
                                           int                update,
                                           hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && src && w, "p1_apply_cell: null pointer" );
   HH_REQUIRE( level_ok( level ), "p1_apply_cell: level out of range [2,11]" );
   HH_REQUIRE( dst != src, "p1_apply_cell: src and dst must not alias" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p1_apply_cell: bad update type" );
   if ( update == HYTEG_HIP_REPLACE )
      return launch_apply< APPLY_REPLACE >( dst, src, nullptr, nullptr, level, w, 0.0, as_stream( stream ) );
   return launch_apply< APPLY_ADD >( dst, src, nullptr, nullptr, level, w, 0.0, as_stream( stream ) );
}

// ---- float instantiations (the reference instantiates its generated apply kernels for float as well:
// apply_3D_macrocell_vertexdof_to_vertexdof_replace.cpp:96-97); levels 2..10 (32-bit buffer addressing) ----
HYTEG_HIP_API int hyteg_hip_p1_apply_cell_f32( float* dst, const float* src, int level, const double* w, int update, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && src && w, "p1_apply_cell_f32: null pointer" );
   HH_REQUIRE( level >= HYTEG_HIP_MIN_LEVEL && level <= 10, "p1_apply_cell_f32: level out of range [2,10]" );
   HH_REQUIRE( dst != src, "p1_apply_cell_f32: src and dst must not alias" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p1_apply_cell_f32: bad update type" );
   if ( update == HYTEG_HIP_REPLACE )
      return launch_zmarch< APPLY_REPLACE, float >( dst, src, nullptr, nullptr, level, w, 0.0, as_stream( stream ) );
   return launch_zmarch< APPLY_ADD, float >( dst, src, nullptr, nullptr, level, w, 0.0, as_stream( stream ) );
}

HYTEG_HIP_API int hyteg_hip_p1_jacobi_cell_f32( float*             dst,
                                                const float*       rhs,
                                                const float*       src,
                                                const float*       invdiag,
                                                int                level,
                                                const double*      w,
                                                double             relax,
                                                hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && rhs && src && w, "p1_jacobi_cell_f32: null pointer" );
   HH_REQUIRE( level >= HYTEG_HIP_MIN_LEVEL && level <= 10, "p1_jacobi_cell_f32: level out of range [2,10]" );
   HH_REQUIRE( dst != src, "p1_jacobi_cell_f32: src and dst must not alias" );
   HH_REQUIRE( w[7] != 0.0, "p1_jacobi_cell_f32: zero centre weight" );
   return launch_zmarch< APPLY_JACOBI, float >( dst, src, rhs, invdiag, level, w, relax, as_stream( stream ) );
}

HYTEG_HIP_API int hyteg_hip_p1_residual_cell( double*            dst,
                                              const double*      rhs,
                                              const double*      src,
                                              int                level,
                                              const double*      w,
                                              hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && rhs && src && w, "p1_residual_cell: null pointer" );
   HH_REQUIRE( level >= HYTEG_HIP_MIN_LEVEL && level <= 10, "p1_residual_cell: level out of range [2,10]" );
   HH_REQUIRE( dst != src, "p1_residual_cell: src and dst must not alias" );
   return launch_zmarch< APPLY_RESIDUAL >( dst, src, rhs, (const double*) nullptr, level, w, 0.0, as_stream( stream ) );
}

HYTEG_HIP_API int hyteg_hip_p1_apply_kernel_name( int level, int update, char* buf, size_t buflen )
{
   HH_REQUIRE( buf && buflen > 0, "p1_apply_kernel_name: null buffer" );
   HH_REQUIRE( level_ok( level ), "p1_apply_kernel_name: level out of range [2,11]" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p1_apply_kernel_name: bad update type" );
   const int mode = update == HYTEG_HIP_REPLACE ? APPLY_REPLACE : APPLY_ADD;
   if ( tet64( ( 1 << level ) + 1 ) * 8 >= ( (int64_t) 1 << 31 ) )
   {
      snprintf( buf, buflen, "p1_apply_tiled_kernel<MODE=%d>", mode );
      return HYTEG_HIP_OK;
   }
   const int  lz = brick_lz( level );
   BrickTable bt;
   int        rc = lz == 8 ? get_bricks( level, kBrickNY, 8, &bt ) : get_bricks( level, kBrickNY, 4, &bt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   const bool dec = bt.decodable && apply_decode_enabled();
   const int  pfd = dec ? 1 : apply_prefetch_distance();
   const bool pre = !dec && pfd == 2 && apply_preload_enabled(); // as in launch_zmarch_lz
   if ( mode == APPLY_REPLACE && lz == 8 && apply_aligned_enabled() )
   {
      snprintf( buf, buflen, "p1_apply_zmarch_preload_kernel<MODE=%d,NY=%d,LZ=%d,EX_AUX=0,DEC=0,PFD=2,XS=56>", mode, kBrickNY, lz );
      return HYTEG_HIP_OK;
   }
   snprintf( buf, buflen, "%s<MODE=%d,NY=%d,LZ=%d,EX_AUX=%d,DEC=%d,PFD=%d>", pre ? "p1_apply_zmarch_preload_kernel" : "p1_apply_zmarch_kernel", mode,
             kBrickNY, lz, mode == APPLY_ADD ? 2 : 0, dec ? 1 : 0, pfd );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_jacobi_cell( double*            dst,
                                            const double*      rhs,
                                            const double*      src,
                                            const double*      invdiag,
                                            int                level,
                                            const double*      w,
                                            double             relax,
                                            hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && rhs && src && w, "p1_jacobi_cell: null pointer" );
   HH_REQUIRE( level_ok( level ), "p1_jacobi_cell: level out of range [2,11]" );
   HH_REQUIRE( dst != src, "p1_jacobi_cell: src and dst must not alias" );
   HH_REQUIRE( w[7] != 0.0, "p1_jacobi_cell: zero centre weight" );
   return launch_apply< APPLY_JACOBI >( dst, src, rhs, invdiag, level, w, relax, as_stream( stream ) );
}
}
