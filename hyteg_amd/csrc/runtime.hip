// Runtime part of the C-ABI: errors, device/memory/stream helpers, layout helpers, tile tables.
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.hpp"
#include "kernels_apply_zmarch.hpp"

namespace hyteg_hip {

static thread_local std::string g_last_error;

void set_error( const std::string& msg ) { g_last_error = msg; }
int  fail( int code, const std::string& msg )
{
   g_last_error = msg;
   return code;
}

// ---- tile tables --------------------------------------------------------------------------------
static std::vector< Tile > build_tiles( int level, TileKind kind, int capacity )
{
   const int           N = ( 1 << level ) + 1;
   std::vector< Tile > tiles;
   if ( kind == TILES_ROWS )
   {
      for ( int z = 0; z < N; ++z )
         for ( int y = 0; y < N - z; ++y )
         {
            const int R = N - z - y;
            for ( int x0 = 0; x0 < R; x0 += capacity )
            {
               Tile tl{};
               tl.a   = cell_index( N, x0, y, z );
               tl.cnt = std::min( capacity, R - x0 );
               tl.z   = z;
               tl.ya  = y;
               tl.yb  = x0;
               // the same (x0, y, z) in tetrahedral arrays of width N-1 and N-2 (the P2 edge-DoF kinds; plain index algebra,
               // meaningful only where the point exists there)
               tl.pad[0] = N >= 2 ? cell_index( N - 1, x0, y, z ) : 0;
               tl.pad[1] = N >= 3 ? cell_index( N - 2, x0, y, z ) : 0;
               tiles.push_back( tl );
            }
         }
      return tiles;
   }
   const int           zlo = kind == TILES_INNER ? 1 : 0;
   const int           zhi = kind == TILES_INNER ? N - 4 : N - 1;
   for ( int z = zlo; z <= zhi; ++z )
   {
      const int W  = N - z;
      const int s0 = slice_start( N, z );
      int       first, last; // slice-local offsets, inclusive
      if ( kind == TILES_INNER )
      {
         first = row_start( W, 1 ) + 1;
         last  = row_start( W, W - 3 ) + 1;
      }
      else
      {
         first = 0;
         last  = tri( W ) - 1;
      }
      // balance the tiles of a slice: equal sizes rather than full tiles plus a short tail
      const int total  = last - first + 1;
      const int ntiles = ( total + capacity - 1 ) / capacity;
      int       y      = 0;
      for ( int t = 0; t < ntiles; ++t )
      {
         const int lo = first + (int) ( ( (int64_t) total * t ) / ntiles );
         const int hi = first + (int) ( ( (int64_t) total * ( t + 1 ) ) / ntiles ) - 1;
         Tile      tl{};
         tl.a   = s0 + lo;
         tl.cnt = hi - lo + 1;
         tl.z   = z;
         while ( y + 1 < W && row_start( W, y + 1 ) <= lo )
            ++y;
         tl.ya  = y;
         int yb = y;
         while ( yb + 1 < W && row_start( W, yb + 1 ) <= hi )
            ++yb;
         tl.yb = yb;
         tiles.push_back( tl );
      }
   }
   return tiles;
}

int get_tiles( int level, TileKind kind, int capacity, TileTable* out )
{
   static std::mutex                                                  mtx;
   static std::map< std::tuple< int, int, int, int >, TileTable >    cache;
   int                                                                dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   std::lock_guard< std::mutex > lock( mtx );
   auto                          key = std::make_tuple( dev, level, (int) kind, capacity );
   auto                          it  = cache.find( key );
   if ( it != cache.end() )
   {
      *out = it->second;
      return HYTEG_HIP_OK;
   }
   std::vector< Tile > host = build_tiles( level, kind, capacity );
   TileTable           tt;
   tt.count = (int) host.size();
   if ( tt.count > 0 )
   {
      void* p = nullptr;
      HH_CHECK_HIP( hipMalloc( &p, host.size() * sizeof( Tile ) ) );
      HH_CHECK_HIP( hipMemcpy( p, host.data(), host.size() * sizeof( Tile ), hipMemcpyHostToDevice ) );
      tt.dev = static_cast< const Tile* >( p );
   }
   cache[key] = tt;
   *out       = tt;
   return HYTEG_HIP_OK;
}

int get_bricks( int level, int NY, int LZ, BrickTable* out, int XS )
{
   // hot path: the same (level, shape) is asked for on every launch
   thread_local int        lastKey[4] = { -1, -1, -1, -1 };
   thread_local BrickTable lastVal;
   int                     dev0 = 0;
   HH_CHECK_HIP( hipGetDevice( &dev0 ) );
   NY = NY * 1000 + XS; // the key's shape entry carries the x-stride
   if ( lastKey[0] == dev0 && lastKey[1] == level && lastKey[2] == NY && lastKey[3] == LZ )
   {
      *out = lastVal;
      return HYTEG_HIP_OK;
   }
   static std::mutex                                             mtx;
   static std::map< std::tuple< int, int, int, int >, BrickTable > cache;
   int                                                           dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   std::lock_guard< std::mutex > lock( mtx );
   auto                          key = std::make_tuple( dev, level, NY, LZ );
   auto                          it  = cache.find( key );
   if ( it != cache.end() )
   {
      *out       = it->second;
      lastKey[0] = dev, lastKey[1] = level, lastKey[2] = NY, lastKey[3] = LZ;
      lastVal    = it->second;
      return HYTEG_HIP_OK;
   }
   std::vector< BrickTask > host;
   std::vector< int >       zs;
   build_brick_tasks( level, NY / 1000, LZ, host, &zs, XS );
   BrickTable bt;
   bt.count = (int) host.size();
   // decode mode: zs = starts of the z-chunks followed by the total
   bt.decodable = XS == 62 && (int) zs.size() - 1 <= kZMarchMaxZChunks && ( 1 << level ) - 3 <= 62 * kZMarchMaxStairs;
   for ( int k = 0; k < kZMarchMaxZChunks; ++k )
      bt.zs[k] = k + 1 < (int) zs.size() ? zs[k] : bt.count;
   if ( bt.count > 0 )
   {
      void* p = nullptr;
      HH_CHECK_HIP( hipMalloc( &p, host.size() * sizeof( BrickTask ) ) );
      HH_CHECK_HIP( hipMemcpy( p, host.data(), host.size() * sizeof( BrickTask ), hipMemcpyHostToDevice ) );
      bt.dev = static_cast< const BrickTask* >( p );
   }
   cache[key] = bt;
   *out       = bt;
   lastKey[0] = dev, lastKey[1] = level, lastKey[2] = NY, lastKey[3] = LZ;
   lastVal    = bt;
   return HYTEG_HIP_OK;
}

} // namespace hyteg_hip

using namespace hyteg_hip;

namespace hyteg_hip {
// ticket counter of the single-launch reductions, one per (device, stream); zero between launches
int dot_counter( hipStream_t stream, unsigned** out )
{
   static std::mutex                                         mtx;
   static std::map< std::pair< int, hipStream_t >, unsigned* > counters;
   int                                                       dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   std::lock_guard< std::mutex > lock( mtx );
   auto                          it = counters.find( { dev, stream } );
   if ( it == counters.end() )
   {
      void* p = nullptr;
      HH_CHECK_HIP( hipMalloc( &p, sizeof( unsigned ) ) );
      HH_CHECK_HIP( hipMemset( p, 0, sizeof( unsigned ) ) );
      it = counters.emplace( std::make_pair( dev, stream ), static_cast< unsigned* >( p ) ).first;
   }
   *out = it->second;
   return HYTEG_HIP_OK;
}
} // namespace hyteg_hip

extern "C" {

HYTEG_HIP_API const char* hyteg_hip_version( void ) { return "hyteg_hip 0.1 (gfx950)"; }
HYTEG_HIP_API const char* hyteg_hip_last_error( void ) { return g_last_error.c_str(); }

HYTEG_HIP_API int hyteg_hip_device_count( int* count )
{
   HH_REQUIRE( count != nullptr, "device_count: null out pointer" );
   hipError_t e = hipGetDeviceCount( count );
   if ( e != hipSuccess )
   {
      *count = 0;
      return fail( HYTEG_HIP_ENODEV, std::string( "hipGetDeviceCount: " ) + hipGetErrorString( e ) );
   }
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_set_device( int device )
{
   HH_CHECK_HIP( hipSetDevice( device ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_device_name( char* buf, size_t buflen )
{
   HH_REQUIRE( buf != nullptr && buflen > 0, "device_name: bad buffer" );
   int dev = 0;
   HH_CHECK_HIP( hipGetDevice( &dev ) );
   hipDeviceProp_t prop;
   HH_CHECK_HIP( hipGetDeviceProperties( &prop, dev ) );
   snprintf( buf, buflen, "%s (%s)", prop.name, prop.gcnArchName );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_malloc( void** dev_ptr, size_t bytes )
{
   HH_REQUIRE( dev_ptr != nullptr, "malloc: null out pointer" );
   *dev_ptr = nullptr;
   // arrays of 1 MiB and more are rounded up to whole 2 MiB fragments, as torch's allocator does (their addresses are 2 MiB-aligned
   // already).  Measured in round 3 (tools/gpu/scratch/alloc_probe.py): no effect on the kernels' speed, on any box met --
   // hipMalloc per array, rounded sizes, one arena and torch's allocator all run the apply at the same rate; kept because it
   // costs nothing and keeps an array's last page fragment to itself.
   constexpr size_t kFragment = size_t( 2 ) << 20;
   if ( bytes >= ( size_t( 1 ) << 20 ) )
      bytes = ( bytes + kFragment - 1 ) / kFragment * kFragment;
   HH_CHECK_HIP( hipMalloc( dev_ptr, bytes ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_free( void* dev_ptr )
{
   HH_CHECK_HIP( hipFree( dev_ptr ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_memset_zero( void* dev_ptr, size_t bytes, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dev_ptr != nullptr || bytes == 0, "memset_zero: null pointer" );
   HH_CHECK_HIP( hipMemsetAsync( dev_ptr, 0, bytes, as_stream( stream ) ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_upload( void* dev_dst, const void* host_src, size_t bytes, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( ( dev_dst && host_src ) || bytes == 0, "upload: null pointer" );
   HH_CHECK_HIP( hipMemcpyAsync( dev_dst, host_src, bytes, hipMemcpyHostToDevice, as_stream( stream ) ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_download( void* host_dst, const void* dev_src, size_t bytes, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( ( host_dst && dev_src ) || bytes == 0, "download: null pointer" );
   HH_CHECK_HIP( hipMemcpyAsync( host_dst, dev_src, bytes, hipMemcpyDeviceToHost, as_stream( stream ) ) );
   HH_CHECK_HIP( hipStreamSynchronize( as_stream( stream ) ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_copy( void* dev_dst, const void* dev_src, size_t bytes, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( ( dev_dst && dev_src ) || bytes == 0, "copy: null pointer" );
   HH_CHECK_HIP( hipMemcpyAsync( dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice, as_stream( stream ) ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_stream_create( hyteg_hip_stream_t* stream )
{
   HH_REQUIRE( stream != nullptr, "stream_create: null out pointer" );
   hipStream_t s;
   HH_CHECK_HIP( hipStreamCreateWithFlags( &s, hipStreamNonBlocking ) );
   *stream = reinterpret_cast< hyteg_hip_stream_t >( s );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_stream_destroy( hyteg_hip_stream_t stream )
{
   HH_CHECK_HIP( hipStreamDestroy( as_stream( stream ) ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_stream_synchronize( hyteg_hip_stream_t stream )
{
   HH_CHECK_HIP( hipStreamSynchronize( as_stream( stream ) ) );
   return HYTEG_HIP_OK;
}

// ---- graphs -----------------------------------------------------------------------------------------
// Launch-bound sequences (the coarse levels of a multigrid cycle: ~100 dependent launches of a few microseconds) are
// recorded once from the stream they are issued on and replayed as ONE graph launch; every entry point of this
// library is capturable once its lazily built tables exist (hyteg_hip_prepare_level or one ordinary call).
HYTEG_HIP_API int hyteg_hip_graph_begin_capture( hyteg_hip_stream_t stream )
{
   HH_REQUIRE( stream != nullptr, "graph_begin_capture: the null stream cannot be captured" );
   HH_CHECK_HIP( hipStreamBeginCapture( as_stream( stream ), hipStreamCaptureModeRelaxed ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_graph_end_capture( hyteg_hip_stream_t stream, hyteg_hip_graph_t* graph )
{
   HH_REQUIRE( stream != nullptr && graph != nullptr, "graph_end_capture: null argument" );
   *graph       = nullptr;
   hipGraph_t g = nullptr;
   HH_CHECK_HIP( hipStreamEndCapture( as_stream( stream ), &g ) );
   if ( g == nullptr )
      return HYTEG_HIP_OK; // nothing was recorded
   size_t nodes = 0;
   HH_CHECK_HIP( hipGraphGetNodes( g, nullptr, &nodes ) );
   if ( nodes == 0 )
   {
      HH_CHECK_HIP( hipGraphDestroy( g ) );
      return HYTEG_HIP_OK;
   }
   hipGraphExec_t exec = nullptr;
   hipError_t     e    = hipGraphInstantiate( &exec, g, nullptr, nullptr, 0 );
   (void) hipGraphDestroy( g );
   if ( e != hipSuccess )
      return fail( HYTEG_HIP_ELAUNCH, std::string( "hipGraphInstantiate: " ) + hipGetErrorString( e ) );
   *graph = reinterpret_cast< hyteg_hip_graph_t >( exec );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_graph_abort_capture( hyteg_hip_stream_t stream )
{
   hipGraph_t g = nullptr;
   (void) hipStreamEndCapture( as_stream( stream ), &g );
   if ( g )
      (void) hipGraphDestroy( g );
   (void) hipGetLastError();
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_graph_launch( hyteg_hip_graph_t graph, hyteg_hip_stream_t stream )
{
   if ( graph == nullptr )
      return HYTEG_HIP_OK; // an empty recording
   HH_CHECK_HIP( hipGraphLaunch( reinterpret_cast< hipGraphExec_t >( graph ), as_stream( stream ) ) );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_graph_destroy( hyteg_hip_graph_t graph )
{
   if ( graph )
      HH_CHECK_HIP( hipGraphExecDestroy( reinterpret_cast< hipGraphExec_t >( graph ) ) );
   return HYTEG_HIP_OK;
}

// ---- calibration: streaming copy (the practical floor a stencil sweep is compared with) ---------------
}

namespace hyteg_hip {
// one double per lane and step, plain loads, nontemporal stores, 2048 workgroups: the fastest of the copy kernels measured on
// MI355X for one level-8 cell array (hyteg_amd/csrc/exp/copy_floor_probe.hip, profiles/r03_copy_floor_probe.txt: 7.5 us for
// 22.9 MB in + 22.9 MB out; 16 bytes per lane 7.6-7.8; nontemporal loads 8.9-9.0; plain stores 9.7-10; hipMemcpyAsync 10.5)
template < bool NT >
__global__ __launch_bounds__( 256 ) void calib_copy_kernel( double* __restrict__ dst, const double* __restrict__ src, int64_t n )
{
   const int64_t stride = (int64_t) gridDim.x * 256;
   for ( int64_t k = (int64_t) blockIdx.x * 256 + threadIdx.x; k < n; k += stride )
   {
      if constexpr ( NT )
         __builtin_nontemporal_store( src[k], &dst[k] );
      else
         dst[k] = src[k];
   }
}
} // namespace hyteg_hip

extern "C" {
HYTEG_HIP_API int hyteg_hip_calib_copy( double* dst, const double* src, int64_t n, int nontemporal, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && src && n > 0, "calib_copy: null pointer or empty array" );
   HH_REQUIRE( ( reinterpret_cast< uintptr_t >( dst ) & 7 ) == 0 && ( reinterpret_cast< uintptr_t >( src ) & 7 ) == 0,
               "calib_copy: arrays of doubles must be 8-byte aligned" );
   int64_t blocks = ( n + 255 ) / 256;
   blocks         = blocks > 2048 ? 2048 : blocks; // 8 workgroups per CU, grid-stride beyond that
   if ( nontemporal )
      hipLaunchKernelGGL( calib_copy_kernel< true >, dim3( (unsigned) blocks ), dim3( 256 ), 0, as_stream( stream ), dst, src, n );
   else
      hipLaunchKernelGGL( calib_copy_kernel< false >, dim3( (unsigned) blocks ), dim3( 256 ), 0, as_stream( stream ), dst, src, n );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_calib_copy_ring( double* const* dsts, const double* const* srcs, int npairs, int64_t n, int nontemporal, int first,
                                             int count, hyteg_hip_stream_t stream, hyteg_hip_event_t start, hyteg_hip_event_t stop )
{
   HH_REQUIRE( dsts && srcs && npairs > 0 && count >= 0 && first >= 0, "calib_copy_ring: bad ring" );
   if ( start )
      HH_CHECK_HIP( hipEventRecord( reinterpret_cast< hipEvent_t >( start ), as_stream( stream ) ) );
   for ( int k = first; k < first + count; ++k )
   {
      const int rc = hyteg_hip_calib_copy( dsts[k % npairs], srcs[k % npairs], n, nontemporal, stream );
      if ( rc != HYTEG_HIP_OK )
         return rc;
   }
   if ( stop )
      HH_CHECK_HIP( hipEventRecord( reinterpret_cast< hipEvent_t >( stop ), as_stream( stream ) ) );
   return HYTEG_HIP_OK;
}

// ---- layout ---------------------------------------------------------------------------------------
HYTEG_HIP_API int64_t hyteg_hip_cell_width( int level ) { return ( (int64_t) 1 << level ) + 1; }
HYTEG_HIP_API int64_t hyteg_hip_cell_size( int level ) { return tet64( hyteg_hip_cell_width( level ) ); }
HYTEG_HIP_API int64_t hyteg_hip_cell_inner_size( int level )
{
   const int64_t n = ( (int64_t) 1 << level ) - 1;
   return n < 3 ? 0 : n * ( n - 1 ) * ( n - 2 ) / 6;
}
HYTEG_HIP_API int64_t hyteg_hip_cell_index( int level, int x, int y, int z )
{
   const int64_t N = hyteg_hip_cell_width( level ), W = N - z;
   return tet64( N ) - tet64( W ) + (int64_t) y * W - ( (int64_t) y * ( y - 1 ) ) / 2 + x;
}
}
