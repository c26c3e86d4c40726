// C-ABI entry points: constant-stencil apply and fused weighted Jacobi on one macro-cell.
#include <cstdlib>
#include <string>
#include <type_traits>

#include "kernels_apply.hpp"
#include "kernels_apply_zmarch.hpp"

using namespace hyteg_hip;

namespace {

constexpr int kTile = 1024;

// Brick shape of the z-march kernel: rows x slices per wave, and how many slices the loads run ahead of the arithmetic.
// Level 8 has only ~64k wave-rows for 1024 SIMDs, so the shape trades reuse (taller / wider bricks re-read fewer halo rows
// and slices) against the number of waves and against the share of a wave's loads that precede its first store (the
// prologue: (2 + PFD) of LZ + 2 slices).  Round 3 swept the shapes with bench.py (profiles/r03_apply_shape_sweep.txt):
// at level 8 more, shorter waves with a short prologue win -- 2 x 8, one slice ahead: 9.17-9.26 us against 9.53-9.57 us for
// round 2's 4 x 8, two slices ahead; everything taller or wider is slower (4 x 16: 13.3, 6 x 8: 11.1, 4 x 12: 11.0 us).
struct BrickShape
{
   int  ny, lz, pfd;
   bool operator==( const BrickShape& o ) const { return ny == o.ny && lz == o.lz && pfd == o.pfd; }
};
BrickShape g_shape_override{ 0, 0, 0 }; // hyteg_hip_set_apply_shape (tuning knob; 0 = the defaults below)

// defaults read off the sweep of every compiled shape x mode x level (tools/apply_shape_sweep.py,
// profiles/r03_apply_shape_sweep.txt): levels <= 7 (few hundred bricks, cache-resident) want many short waves; level 8
// (one generation of waves, HBM-bound) wants 8 slices and a short prologue, two rows for the two-stream kernels and four for
// the three-stream ones; from level 9 on (several generations) 4 x 4, two slices ahead
inline BrickShape default_shape( int mode, int level, bool f32 )
{
   // Chosen from tools/apply_shape_sweep.py with every operand class 2.2 x larger than the Infinity Cache (HBM regime,
   // profiles/r03_apply_shape_sweep_hbm.txt).  The first sweep of round 3 ran on rings whose SOURCE arrays fitted into that cache
   // and preferred 2 x 8, one slice ahead, for level-8 Replace (9.15 against 9.50 us); read from HBM that shape is the slower
   // one (13.3 against 12.5 us).  Levels <= 7: arrays of a few MB, cache-resident in any cycle; levels >= 9: HBM either way.
   if ( level <= 7 )
      return ( mode == APPLY_REPLACE || f32 ) ? BrickShape{ 2, 4, 1 } : BrickShape{ 4, 4, 1 };
   if ( level == 8 )
      return ( mode == APPLY_ADD || mode == APPLY_RESIDUAL ) ? BrickShape{ 4, 8, 1 } : BrickShape{ 4, 8, 2 };
   return BrickShape{ 4, 4, 2 };
}

// `extra`: the second float output of APPLY_RESIDUAL_F32OUT / the double accumulator of APPLY_JACOBI_ACCUM
template < int MODE, int NY, int LZ, int PFD, typename T >
int launch_zmarch_shape( void* dst, const T* src, const T* rhs, const T* invdiag, int level, const double* w, double relax, hipStream_t stream,
                         void* extra = nullptr )
{
   BrickTable bt;
   int        rc = get_bricks( level, NY, LZ, &bt );
   if ( rc != HYTEG_HIP_OK )
      return rc;
   if ( bt.count == 0 )
      return HYTEG_HIP_OK;
   ZMarchArgs A{};
   A.dst     = dst;
   A.src     = src;
   A.rhs     = rhs;
   A.invdiag = invdiag;
   A.dst2    = extra;
   A.xacc    = static_cast< double* >( extra );
   A.tasks   = bt.dev;
   A.ntasks  = bt.count;
   A.N       = ( 1 << level ) + 1;
   A.bytes   = (unsigned) ( tet64( A.N ) * (int64_t) sizeof( T ) );
   A.relax   = relax;
   for ( int k = 0; k < 15; ++k )
      A.st.w[k] = w[k];
   int nblocks = ( bt.count + kZMarchWavesPerBlock - 1 ) / kZMarchWavesPerBlock;
   nblocks     = ( nblocks + 7 ) & ~7;
   A.xcd_chunk = nblocks / 8;
   static const bool xcdSlabs = [] {
      const char* e = getenv( "HYTEG_HIP_APPLY_XCD_SLABS" ); // measurement switch: 0 = workgroups in launch order (XCDs interleaved brick by brick)
      return !( e && e[0] == '0' );
   }();
   if ( !xcdSlabs )
      A.xcd_chunk = 0;
   // dst of Add is read exactly once per element and written right after: nontemporal load (18.6 -> 17.2 us).  rhs /
   // inverse diagonal of Jacobi are re-read by the next sweep of the smoother and stay plain (nontemporal: 12.4 -> 17.6 us
   // when they are still in the Infinity Cache, -2% when they are not).
   constexpr int kExAux = MODE == APPLY_ADD ? 2 : 0; // (the right-hand side of the residual mode is re-read by the cycle: plain)
   // the first three arguments are preloaded into SGPRs (-mllvm -amdgpu-kernarg-preload-count=4): the task load does not wait
   // for a kernel-argument load (round 2: 9.90-10.18 -> 9.60-9.74 us)
   hipLaunchKernelGGL( ( p1_apply_zmarch_preload_kernel< MODE, NY, LZ, kExAux, false, PFD, T > ), dim3( nblocks ), dim3( 64 * kZMarchWavesPerBlock ), 0,
                       stream, A.tasks, A.ntasks, A.xcd_chunk, A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

// the shapes compiled in: the defaults and the runners-up of the sweep (so that the sweep can be repeated on another box)
#define HYTEG_ZM_SHAPES( X ) X( 2, 8, 1 ) X( 4, 8, 2 ) X( 4, 8, 1 ) X( 4, 4, 2 ) X( 4, 4, 1 ) X( 2, 4, 1 ) X( 8, 4, 2 )

inline bool shape_compiled( const BrickShape& s )
{
#define HH_X( NY_, LZ_, PFD_ ) \
   if ( s == BrickShape{ NY_, LZ_, PFD_ } ) \
      return true;
   HYTEG_ZM_SHAPES( HH_X )
#undef HH_X
   return false;
}

// the shape of a launch: hyteg_hip_set_apply_shape, else HYTEG_HIP_APPLY_SHAPE=NYxLZxPFD from the environment (read once;
// lets tools/gpu/r03_shapes.sh A/B whole bench.py runs), else the default of the mode and level
inline BrickShape current_shape( int mode, int level, bool f32 )
{
   static const BrickShape env = [] {
      BrickShape  e{ 0, 0, 0 };
      const char* v = getenv( "HYTEG_HIP_APPLY_SHAPE" );
      if ( v && sscanf( v, "%dx%dx%d", &e.ny, &e.lz, &e.pfd ) != 3 )
         e = BrickShape{ -1, -1, -1 }; // unparsable: every launch fails with "not compiled in"
      return e;
   }();
   if ( g_shape_override.ny )
      return g_shape_override;
   if ( env.ny )
      return env;
   return default_shape( mode, level, f32 );
}

template < int MODE, typename T = double >
int launch_zmarch( void* dst, const T* src, const T* rhs, const T* invdiag, int level, const double* w, double relax, hipStream_t stream,
                   void* extra = nullptr )
{
   // the fused mixed-precision steps take the shapes of the kernels they replace (residual / float Jacobi)
   const int        shapeMode = MODE == APPLY_RESIDUAL_F32OUT ? APPLY_RESIDUAL : ( MODE == APPLY_JACOBI_ACCUM ? APPLY_JACOBI : MODE );
   const BrickShape s         = current_shape( shapeMode, level, !std::is_same< T, double >::value );
#define HH_X( NY_, LZ_, PFD_ ) \
   if ( s == BrickShape{ NY_, LZ_, PFD_ } ) \
      return launch_zmarch_shape< MODE, NY_, LZ_, PFD_, T >( dst, src, rhs, invdiag, level, w, relax, stream, extra );
   HYTEG_ZM_SHAPES( HH_X )
#undef HH_X
   return fail( HYTEG_HIP_EINVAL, "apply: brick shape not compiled in" );
}

template < int MODE >
int launch_apply( double* dst, const double* src, const double* rhs, const double* invdiag, int level, const double* w,
                  double relax, hipStream_t stream )
{
   // z-march register kernel whenever byte offsets fit the 32-bit buffer addressing (level <= 10);
   // the LDS-tiled kernel (pointer addressing) covers level 11
   static const bool forceTiled = [] {
      const char* e = getenv( "HYTEG_HIP_APPLY_LDS_TILED" ); // measurement switch: the LDS-tiled kernel of round 1 at every level
      return e && e[0] == '1';
   }();
   if ( !forceTiled && tet64( ( 1 << level ) + 1 ) * 8 < ( (int64_t) 1 << 31 ) )
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

// ---- the two fused steps of the mixed-precision Jacobi smoother (host/solvers.hpp MixedPrecisionJacobiSmoother::solveSteps) ----
HYTEG_HIP_API int hyteg_hip_p1_residual_jacobi_start_f32( float*             r_f32,
                                                          float*             e_f32,
                                                          const double*      rhs,
                                                          const double*      src,
                                                          int                level,
                                                          const double*      w,
                                                          double             relax,
                                                          hyteg_hip_stream_t stream )
{
   HH_REQUIRE( r_f32 && e_f32 && rhs && src && w, "p1_residual_jacobi_start_f32: null pointer" );
   HH_REQUIRE( level >= HYTEG_HIP_MIN_LEVEL && level <= 10, "p1_residual_jacobi_start_f32: level out of range [2,10]" );
   HH_REQUIRE( r_f32 != e_f32, "p1_residual_jacobi_start_f32: the two outputs must differ" );
   HH_REQUIRE( w[7] != 0.0, "p1_residual_jacobi_start_f32: zero centre weight" );
   return launch_zmarch< APPLY_RESIDUAL_F32OUT >( r_f32, src, rhs, (const double*) nullptr, level, w, relax, as_stream( stream ), e_f32 );
}

HYTEG_HIP_API int hyteg_hip_p1_jacobi_accumulate_f32( double*            x,
                                                      const float*       rhs_f32,
                                                      const float*       e_f32,
                                                      int                level,
                                                      const double*      w,
                                                      double             relax,
                                                      hyteg_hip_stream_t stream )
{
   HH_REQUIRE( x && rhs_f32 && e_f32 && w, "p1_jacobi_accumulate_f32: null pointer" );
   HH_REQUIRE( level >= HYTEG_HIP_MIN_LEVEL && level <= 10, "p1_jacobi_accumulate_f32: level out of range [2,10]" );
   HH_REQUIRE( w[7] != 0.0, "p1_jacobi_accumulate_f32: zero centre weight" );
   return launch_zmarch< APPLY_JACOBI_ACCUM, float >( nullptr, e_f32, rhs_f32, (const float*) nullptr, level, w, relax, as_stream( stream ), x );
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
   const BrickShape s = current_shape( mode, level, false );
   snprintf( buf, buflen, "p1_apply_zmarch_preload_kernel<MODE=%d,NY=%d,LZ=%d,EX_AUX=%d,DEC=0,PFD=%d>", mode, s.ny, s.lz, mode == APPLY_ADD ? 2 : 0, s.pfd );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_set_apply_shape( int ny, int lz, int pfd )
{
   if ( ny == 0 && lz == 0 && pfd == 0 )
   {
      g_shape_override = BrickShape{ 0, 0, 0 };
      return HYTEG_HIP_OK;
   }
   HH_REQUIRE( shape_compiled( BrickShape{ ny, lz, pfd } ), "set_apply_shape: this brick shape is not compiled in (p1_apply.hip: HYTEG_ZM_SHAPES)" );
   g_shape_override = BrickShape{ ny, lz, pfd };
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
