// C-ABI entry points for the cell-centric multi-cell model: kernels on the boundary shell of a macro-cell
// (points on its macro-faces/edges/vertices), masked vector kernels, masked dot, and the additive exchange of
// shared points.  The shell has O(4^L) points, so these kernels are latency- rather than bandwidth-relevant.
#include "common.hpp"
#include "p2p_device.hpp"
#include "shell.hpp"

using namespace hyteg_hip;

namespace {

constexpr int kThreads = 256;

using namespace hyteg_hip::shell;

template < typename T >
__global__ __launch_bounds__( kThreads ) void p1_apply_shell_kernel( T* __restrict__ dst,
                                                                      const T* __restrict__ src,
                                                                      int              N,
                                                                      unsigned         mask,
                                                                      int              update,
                                                                      const Slots14x15 S )
{
   const int q = blockIdx.x * kThreads + threadIdx.x;
   int       x, y, z, slot;
   if ( !shell_point( N, q, x, y, z, slot ) || !( ( mask >> slot ) & 1u ) )
      return;
   const T   acc = share< T >( S, src, N, x, y, z, slot );
   const int i   = cell_index( N, x, y, z );
   dst[i]        = update == HYTEG_HIP_ADD ? acc + dst[i] : acc;
}

// The shares of a rank's ONE macro-cell, delivered as they are computed: a point whose share other ranks need (send_first /
// send_list: its indices in the send enumeration of the exchange plan, CSR over the shell enumeration q) stores it straight into
// the peers' receive slots, and the last workgroup publishes the sequence number -- hyteg_hip_p1_apply_cell_boundary and
// hyteg_hip_p2p_pack in one launch, without the pack kernel's gather.
__global__ __launch_bounds__( kThreads ) void p1_apply_shell_send_kernel( double* __restrict__ dst,
                                                                           const double* __restrict__ src,
                                                                           int              N,
                                                                           unsigned         mask,
                                                                           int              update,
                                                                           const Slots14x15 S,
                                                                           const int* __restrict__ send_first,
                                                                           const int* __restrict__ send_list,
                                                                           const hyteg_hip_p2p_peer_t* __restrict__ peers,
                                                                           int                npeers,
                                                                           unsigned long long seq,
                                                                           unsigned*          counter )
{
   const int q = blockIdx.x * kThreads + threadIdx.x;
   int       x, y, z, slot;
   if ( shell_point( N, q, x, y, z, slot ) && ( ( mask >> slot ) & 1u ) )
   {
      const double acc = share( S, src, N, x, y, z, slot );
      const int    i   = cell_index( N, x, y, z );
      const double val = update == HYTEG_HIP_ADD ? acc + dst[i] : acc;
      dst[i]           = val;
      for ( int j = send_first[q]; j < send_first[q + 1]; ++j )
         p2p::send_value( peers, npeers, send_list[j], seq, val );
   }
   p2p::stores_acknowledged();
   __syncthreads();
   if ( threadIdx.x == 0 )
   {
      const unsigned done = __hip_atomic_fetch_add( counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      if ( done == gridDim.x - 1 )
      {
         __hip_atomic_store( counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
         p2p::publish( peers, npeers, seq );
      }
   }
}

struct VecArgsM
{
   double*       dst;
   const double* src[HYTEG_HIP_MAX_SRCS];
   double        c[HYTEG_HIP_MAX_SRCS];
   int           N;
   int           nsrc;
   unsigned      mask;
   int           op; // 0 assign, 1 add, 2 mult, 3 set constant (c[0])
};

__global__ __launch_bounds__( kThreads ) void p1_vector_shell_kernel( const VecArgsM A )
{
   const int q = blockIdx.x * kThreads + threadIdx.x;
   int       x, y, z, slot;
   if ( !shell_point( A.N, q, x, y, z, slot ) || !( ( A.mask >> slot ) & 1u ) )
      return;
   const int i = cell_index( A.N, x, y, z );
   double    tmp;
   if ( A.op == 3 )
      tmp = A.c[0];
   else if ( A.op == 2 )
   {
      tmp = A.src[0][i];
      for ( int k = 1; k < A.nsrc; ++k )
         tmp *= A.src[k][i];
   }
   else
   {
      tmp = A.c[0] * A.src[0][i];
      for ( int k = 1; k < A.nsrc; ++k )
         tmp += A.c[k] * A.src[k][i];
      if ( A.op == 1 )
         tmp = A.dst[i] + tmp;
   }
   A.dst[i] = tmp;
}

// All points of the array selected (flag All on a cell without Dirichlet parts): the layout does not matter, one flat
// launch instead of an inner-point and a shell kernel.  Same expressions as the two kernels it replaces (bit-identical).
constexpr int kFlatPerThread = 4;
__global__ __launch_bounds__( kThreads ) void p1_vector_flat_kernel( const VecArgsM A, int n )
{
   const int base = blockIdx.x * ( kThreads * kFlatPerThread ) + threadIdx.x;
   double    v[kFlatPerThread][HYTEG_HIP_MAX_SRCS + 1];
#pragma unroll
   for ( int u = 0; u < kFlatPerThread; ++u )
   {
      const int i = base + u * kThreads;
      if ( i < n && A.op != 3 )
      {
#pragma unroll
         for ( int k = 0; k < HYTEG_HIP_MAX_SRCS; ++k )
            if ( k < A.nsrc )
               v[u][k] = A.src[k][i];
         if ( A.op == 1 )
            v[u][HYTEG_HIP_MAX_SRCS] = A.dst[i];
      }
   }
#pragma unroll
   for ( int u = 0; u < kFlatPerThread; ++u )
   {
      const int i = base + u * kThreads;
      if ( i >= n )
         continue;
      double tmp;
      if ( A.op == 3 )
         tmp = A.c[0];
      else if ( A.op == 2 )
      {
         tmp = v[u][0];
#pragma unroll
         for ( int k = 1; k < HYTEG_HIP_MAX_SRCS; ++k )
            if ( k < A.nsrc )
               tmp *= v[u][k];
      }
      else
      {
         tmp = A.c[0] * v[u][0];
#pragma unroll
         for ( int k = 1; k < HYTEG_HIP_MAX_SRCS; ++k )
            if ( k < A.nsrc )
               tmp += A.c[k] * v[u][k];
         if ( A.op == 1 )
            tmp = v[u][HYTEG_HIP_MAX_SRCS] + tmp;
      }
      __builtin_nontemporal_store( tmp, &A.dst[i] );
   }
}
inline void launch_flat( const VecArgsM& A, int level, hipStream_t stream )
{
   const int n = (int) tet64( ( 1 << level ) + 1 );
   hipLaunchKernelGGL( p1_vector_flat_kernel, dim3( ( n + kThreads * kFlatPerThread - 1 ) / ( kThreads * kFlatPerThread ) ), dim3( kThreads ), 0,
                       stream, A, n );
}

__global__ __launch_bounds__( kThreads ) void p1_set_inner_kernel( double* dst, double value, const Tile* tiles, int ntiles, int N )
{
   const int t = blockIdx.x;
   if ( t >= ntiles )
      return;
   const Tile tl = tiles[t];
   const int  W = N - tl.z, s0 = slice_start( N, tl.z );
   for ( int e = threadIdx.x; e < tl.cnt; e += kThreads )
   {
      const int i = tl.a + e, j = i - s0;
      const int y = row_of( W, j ), x = j - row_start( W, y );
      if ( x >= 1 && x <= W - y - 2 )
         dst[i] = value;
   }
}

__device__ inline double wave_sum( double v )
{
#pragma unroll
   for ( int off = 32; off > 0; off >>= 1 )
      v += __shfl_down( v, off, 64 );
   return v;
}

// partial sums of a.b over the masked shell points; fixed point -> workgroup assignment (deterministic)
__global__ __launch_bounds__( kThreads ) void p1_dot_shell_kernel( const double* __restrict__ a,
                                                                    const double* __restrict__ b,
                                                                    int      N,
                                                                    unsigned mask,
                                                                    int      npoints,
                                                                    double*  partial )
{
   __shared__ double sh[kThreads / 64];
   double            acc = 0.0;
   for ( int q = blockIdx.x * kThreads + threadIdx.x; q < npoints; q += gridDim.x * kThreads )
   {
      int x, y, z, slot;
      if ( shell_point( N, q, x, y, z, slot ) && ( ( mask >> slot ) & 1u ) )
      {
         const int i = cell_index( N, x, y, z );
         acc         = fma( a[i], b[i], acc );
      }
   }
   acc = wave_sum( acc );
   if ( ( threadIdx.x & 63 ) == 0 )
      sh[threadIdx.x >> 6] = acc;
   __syncthreads();
   if ( threadIdx.x == 0 )
   {
      double r = 0.0;
      for ( int k = 0; k < kThreads / 64; ++k )
         r += sh[k];
      partial[blockIdx.x] = r;
   }
}

__global__ __launch_bounds__( kThreads ) void zero_partials_kernel( double* p, int n )
{
   const int k = blockIdx.x * kThreads + threadIdx.x;
   if ( k < n )
      p[k] = 0.0;
}

__global__ __launch_bounds__( kThreads ) void sum_partials_kernel( const double* partial, int n, double* result )
{
   __shared__ double sh[kThreads / 64];
   double            acc = 0.0;
   for ( int k = threadIdx.x; k < n; k += kThreads )
      acc += partial[k];
   acc = wave_sum( acc );
   if ( ( threadIdx.x & 63 ) == 0 )
      sh[threadIdx.x >> 6] = acc;
   __syncthreads();
   if ( threadIdx.x == 0 )
   {
      double r = 0.0;
      for ( int k = 0; k < kThreads / 64; ++k )
         r += sh[k];
      *result = r;
   }
}

// what a reduce kernel waits for when the values of other ranks arrive peer to peer (comm_p2p.hip): the flag words of this
// rank's arena must have reached seq
struct ArrivalWait
{
   const unsigned long long* flags;
   int                       npeers, stride;
   unsigned long long        seq;
   unsigned*                 status;
   unsigned long long        timeout_ticks;
};

// plans with at least this many groups take the 2-slot form of the reduce kernel
constexpr int kSmallBatchFrom = 4096;

template < bool SUM, bool WAIT = false, int KB = 8 >
__global__ __launch_bounds__( kThreads ) void sum_shared_kernel( double* const* __restrict__ bases,
                                                                  const int* __restrict__ group_ptr,
                                                                  const int* __restrict__ entry_buf,
                                                                  const int* __restrict__ entry_off,
                                                                  int ngroups,
                                                                  int n_writable,
                                                                  ArrivalWait W = ArrivalWait{} )
{
   if constexpr ( WAIT )
   {
      // every workgroup polls for itself (first wave, lane p takes peer p, p + 64, ...); bounded like hyteg_hip_p2p_wait
      if ( threadIdx.x < 64 )
      {
         const unsigned long long t0 = wall_clock64();
         for ( int p = threadIdx.x; p < W.npeers; p += 64 )
         {
            const unsigned long long* f = W.flags + (size_t) p * W.stride;
            while ( __hip_atomic_load( f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM ) < W.seq )
            {
               if ( wall_clock64() - t0 > W.timeout_ticks )
               {
                  __hip_atomic_store( W.status, 1u + (unsigned) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
                  break;
               }
               __builtin_amdgcn_s_sleep( 8 );
            }
         }
         __builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "" );
      }
      __syncthreads();
   }
   const int g = blockIdx.x * kThreads + threadIdx.x;
   if ( g >= ngroups )
      return;
   const int lo = group_ptr[g], hi = group_ptr[g + 1];
   // The copies of a group are read in batches of 8 with all index loads, then all base-pointer loads, then all value
   // loads in flight together (a rolled loop pays three dependent memory round trips per copy: 6 us for a launch that
   // moves a few KB); the sum itself runs in the order of the entries, as before.  KB = 2 for the plans of fine levels, whose
   // groups are almost all macro-face points with two copies (8 slots per group would quadruple their index and value loads).
   constexpr int kBatch = KB;
   double        s      = 0.0;
   for ( int e0 = lo; e0 < ( SUM ? hi : lo + 1 ); e0 += kBatch )
   {
      int     off[kBatch];
      double* b[kBatch];
      double  v[kBatch];
#pragma unroll
      for ( int k = 0; k < kBatch; ++k )
      {
         const int e = e0 + k < hi ? e0 + k : lo;
         off[k]      = entry_off[e];
         b[k]        = bases[entry_buf[e]];
      }
#pragma unroll
      for ( int k = 0; k < kBatch; ++k )
         v[k] = b[k][off[k]];
#pragma unroll
      for ( int k = 0; k < kBatch; ++k )
         if ( e0 + k < ( SUM ? hi : lo + 1 ) )
            s = ( e0 + k == lo ) ? v[k] : s + v[k];
   }
   for ( int e0 = lo; e0 < hi; e0 += kBatch )
   {
      int     off[kBatch], buf[kBatch];
      double* b[kBatch];
#pragma unroll
      for ( int k = 0; k < kBatch; ++k )
      {
         const int e = e0 + k < hi ? e0 + k : lo;
         off[k]      = entry_off[e];
         buf[k]      = entry_buf[e];
         b[k]        = bases[buf[k]];
      }
#pragma unroll
      for ( int k = 0; k < kBatch; ++k )
         if ( e0 + k < hi && buf[k] < n_writable )
            b[k][off[k]] = s;
   }
}

__global__ __launch_bounds__( kThreads ) void gather_entries_kernel( double* __restrict__ out,
                                                                      double* const* __restrict__ bases,
                                                                      const int* __restrict__ entry_buf,
                                                                      const int* __restrict__ entry_off,
                                                                      int n )
{
   const int k = blockIdx.x * kThreads + threadIdx.x;
   if ( k < n )
      out[k] = bases[entry_buf[k]][entry_off[k]];
}


// ---- HyTeG macro-face layout: ghost copies and the one-/two-sided face apply (generic in the vertex map) ----
struct FaceMap
{
   int v[3];
   int v3;
};
struct FaceApplyArgs
{
   double*       dst;
   const double* src;
   int           N;
   int           ncells;
   int           update;
   FaceMap       map[2];
   double        w[2][15];
};

__device__ inline void face_to_cell( int n, const FaceMap& m, int fx, int fy, int fz, int& cx, int& cy, int& cz )
{
   int bary[4] = { 0, 0, 0, 0 };
   bary[m.v[0]] = n - fx - fy - fz;
   bary[m.v[1]] = fx;
   bary[m.v[2]] = fy;
   bary[m.v3]   = fz;
   cx = bary[1], cy = bary[2], cz = bary[3];
}

// dir 0: face -> cell boundary layer (all tri(N) face DoFs); dir 1: cell layer at distance 1 -> ghost layer
__global__ __launch_bounds__( kThreads ) void p1_face_cell_copy_kernel( double* dst, const double* src, int N, FaceMap m, int dir, int ghost_offset )
{
   const int Wf = dir == 0 ? N : N - 1; // width of the triangle that is iterated
   const int r  = blockIdx.x * kThreads + threadIdx.x;
   if ( r >= tri( Wf ) )
      return;
   const int y = row_of( Wf, r ), x = r - row_start( Wf, y );
   int       cx, cy, cz;
   face_to_cell( N - 1, m, x, y, dir, cx, cy, cz );
   const int ci = cell_index( N, cx, cy, cz );
   if ( dir == 0 )
      dst[ci] = src[r];
   else
      dst[ghost_offset + r] = src[ci];
}

__global__ __launch_bounds__( kThreads ) void p1_apply_face3d_kernel( const FaceApplyArgs A )
{
   const int N = A.N, n = N - 1;
   const int r = blockIdx.x * kThreads + threadIdx.x;
   if ( r >= tri( N ) )
      return;
   const int y = row_of( N, r ), x = r - row_start( N, y );
   if ( x < 1 || y < 1 || x + y > n - 1 )
      return; // inner face DoFs only (macroface::Iterator( level, 1 ))
   double tmp = 0.0;
   for ( int k = 0; k < A.ncells; ++k )
   {
      const FaceMap& m = A.map[k];
      int            cx, cy, cz;
      face_to_cell( n, m, x, y, 0, cx, cy, cz );
#pragma unroll
      for ( int s = 0; s < 15; ++s )
      {
         const int lx = cx + kOffs[s][0], ly = cy + kOffs[s][1], lz = cz + kOffs[s][2];
         if ( lx < 0 || ly < 0 || lz < 0 || lx + ly + lz > n )
            continue;
         const int bary[4] = { n - lx - ly - lz, lx, ly, lz };
         const int fx = bary[m.v[1]], fy = bary[m.v[2]], fz = bary[m.v3];
         if ( fz > 1 )
            continue;
         const int idx = fz == 0 ? row_start( N, fy ) + fx : tri( N ) + k * tri( N - 1 ) + row_start( N - 1, fy ) + fx;
         tmp           = fma( A.w[k][s], A.src[idx], tmp );
      }
   }
   A.dst[r] = A.update == HYTEG_HIP_ADD ? A.dst[r] + tmp : tmp;
}

inline bool vmap_ok( int v0, int v1, int v2 )
{
   return v0 >= 0 && v0 < 4 && v1 >= 0 && v1 < 4 && v2 >= 0 && v2 < 4 && v0 != v1 && v0 != v2 && v1 != v2;
}

inline bool shell_level_ok( int level ) { return level >= 0 && level <= HYTEG_HIP_MAX_LEVEL; }
inline int  shell_blocks( int N ) { return ( 4 * tri( N ) + kThreads - 1 ) / kThreads; }

} // namespace

// implemented in p1_vector.hip / p1_transfer.hip
namespace hyteg_hip {
int launch_vec_inner( int op, double* dst, int nsrc, const double* const* srcs, const double* scalars, int level, hipStream_t stream );
int launch_dot_inner_partial( const double* a, const double* b, int level, double* partial, int* nblocks, hipStream_t stream );
} // namespace hyteg_hip

extern "C" {

HYTEG_HIP_API int hyteg_hip_p1_apply_cell_boundary( double*            dst,
                                                    const double*      src,
                                                    int                level,
                                                    const double*      w_slots,
                                                    unsigned           mask,
                                                    int                update,
                                                    hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && src && w_slots, "p1_apply_cell_boundary: null pointer" );
   HH_REQUIRE( shell_level_ok( level ), "p1_apply_cell_boundary: level out of range [0,11]" );
   HH_REQUIRE( dst != src, "p1_apply_cell_boundary: src and dst must not alias" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p1_apply_cell_boundary: bad update type" );
   if ( ( mask & HYTEG_HIP_MASK_SHELL ) == 0 )
      return HYTEG_HIP_OK;
   Slots14x15 S;
   for ( int s = 0; s < 14; ++s )
      for ( int k = 0; k < 15; ++k )
         S.w[s][k] = w_slots[15 * s + k];
   const int N = ( 1 << level ) + 1;
   hipLaunchKernelGGL( p1_apply_shell_kernel< double >, dim3( shell_blocks( N ) ), dim3( kThreads ), 0, as_stream( stream ), dst, src, N,
                       mask & HYTEG_HIP_MASK_SHELL, update, S );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

// float instantiation (the generated elementwise operators exist for float32 as well:
// apps/2023-zikeli-mt/MT-apps/operators-used/P1ElementwiseDiffusion_cubes_const_float32.hpp); weights arrive as doubles
HYTEG_HIP_API int hyteg_hip_p1_apply_cell_boundary_f32( float*             dst,
                                                        const float*       src,
                                                        int                level,
                                                        const double*      w_slots,
                                                        unsigned           mask,
                                                        int                update,
                                                        hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst && src && w_slots, "p1_apply_cell_boundary_f32: null pointer" );
   HH_REQUIRE( shell_level_ok( level ), "p1_apply_cell_boundary_f32: level out of range [0,11]" );
   HH_REQUIRE( dst != src, "p1_apply_cell_boundary_f32: src and dst must not alias" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p1_apply_cell_boundary_f32: bad update type" );
   if ( ( mask & HYTEG_HIP_MASK_SHELL ) == 0 )
      return HYTEG_HIP_OK;
   Slots14x15 S;
   for ( int s = 0; s < 14; ++s )
      for ( int k = 0; k < 15; ++k )
         S.w[s][k] = w_slots[15 * s + k];
   const int N = ( 1 << level ) + 1;
   hipLaunchKernelGGL( p1_apply_shell_kernel< float >, dim3( shell_blocks( N ) ), dim3( kThreads ), 0, as_stream( stream ), dst, src, N,
                       mask & HYTEG_HIP_MASK_SHELL, update, S );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_apply_cell_boundary_p2p( double*                     dst,
                                                        const double*               src,
                                                        int                         level,
                                                        const double*               w_slots,
                                                        unsigned                    mask,
                                                        int                         update,
                                                        const int*                  send_first,
                                                        const int*                  send_list,
                                                        const hyteg_hip_p2p_peer_t* peers,
                                                        int                         npeers,
                                                        unsigned long long          seq,
                                                        unsigned*                   counter,
                                                        hyteg_hip_stream_t          stream )
{
   HH_REQUIRE( dst && src && w_slots, "p1_apply_cell_boundary_p2p: null pointer" );
   HH_REQUIRE( shell_level_ok( level ), "p1_apply_cell_boundary_p2p: level out of range [0,11]" );
   HH_REQUIRE( dst != src, "p1_apply_cell_boundary_p2p: src and dst must not alias" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p1_apply_cell_boundary_p2p: bad update type" );
   HH_REQUIRE( send_first && send_list && peers && npeers > 0 && counter && seq > 0, "p1_apply_cell_boundary_p2p: bad exchange arguments" );
   Slots14x15 S;
   for ( int s = 0; s < 14; ++s )
      for ( int k = 0; k < 15; ++k )
         S.w[s][k] = w_slots[15 * s + k];
   const int N = ( 1 << level ) + 1;
   // launched even if the mask selects nothing: the peers wait for the sequence number
   hipLaunchKernelGGL( p1_apply_shell_send_kernel, dim3( shell_blocks( N ) ), dim3( kThreads ), 0, as_stream( stream ), dst, src, N,
                       mask & HYTEG_HIP_MASK_SHELL, update, S, send_first, send_list, peers, npeers, seq, counter );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_vector_cell_masked( int                  op,
                                                   double*              dst,
                                                   int                  nsrc,
                                                   const double* const* srcs,
                                                   const double*        scalars,
                                                   int                  level,
                                                   unsigned             mask,
                                                   hyteg_hip_stream_t   stream )
{
   HH_REQUIRE( dst && srcs, "p1_vector_cell_masked: null pointer" );
   HH_REQUIRE( op >= 0 && op <= 2, "p1_vector_cell_masked: op must be 0 (assign), 1 (add) or 2 (mult)" );
   HH_REQUIRE( shell_level_ok( level ), "p1_vector_cell_masked: level out of range [0,11]" );
   HH_REQUIRE( nsrc >= 1 && nsrc <= HYTEG_HIP_MAX_SRCS, "p1_vector_cell_masked: nsrc must be 1..HYTEG_HIP_MAX_SRCS" );
   HH_REQUIRE( op == 2 || scalars != nullptr, "p1_vector_cell_masked: null scalars" );
   for ( int k = 0; k < nsrc; ++k )
      HH_REQUIRE( srcs[k] != nullptr, "p1_vector_cell_masked: null source pointer" );
   if ( ( mask & HYTEG_HIP_MASK_ALL ) == HYTEG_HIP_MASK_ALL && level <= 10 )
   {
      VecArgsM A{};
      A.dst = dst;
      for ( int k = 0; k < nsrc; ++k )
      {
         A.src[k] = srcs[k];
         A.c[k]   = scalars ? scalars[k] : 1.0;
      }
      A.N = ( 1 << level ) + 1, A.nsrc = nsrc, A.mask = mask, A.op = op;
      launch_flat( A, level, as_stream( stream ) );
      HH_CHECK_HIP( hipGetLastError() );
      return HYTEG_HIP_OK;
   }
   if ( ( mask & HYTEG_HIP_MASK_INNER ) && level >= HYTEG_HIP_MIN_LEVEL )
   {
      int rc = launch_vec_inner( op, dst, nsrc, srcs, scalars, level, as_stream( stream ) );
      if ( rc != HYTEG_HIP_OK )
         return rc;
   }
   if ( mask & HYTEG_HIP_MASK_SHELL )
   {
      VecArgsM A{};
      A.dst = dst;
      for ( int k = 0; k < nsrc; ++k )
      {
         A.src[k] = srcs[k];
         A.c[k]   = scalars ? scalars[k] : 1.0;
      }
      A.N    = ( 1 << level ) + 1;
      A.nsrc = nsrc;
      A.mask = mask & HYTEG_HIP_MASK_SHELL;
      A.op   = op;
      hipLaunchKernelGGL( p1_vector_shell_kernel, dim3( shell_blocks( A.N ) ), dim3( kThreads ), 0, as_stream( stream ), A );
      HH_CHECK_HIP( hipGetLastError() );
   }
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int
    hyteg_hip_p1_set_cell_masked( double* dst, double value, int level, unsigned mask, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst, "p1_set_cell_masked: null pointer" );
   HH_REQUIRE( shell_level_ok( level ), "p1_set_cell_masked: level out of range [0,11]" );
   const int N = ( 1 << level ) + 1;
   if ( ( mask & HYTEG_HIP_MASK_ALL ) == HYTEG_HIP_MASK_ALL && level <= 10 )
   {
      VecArgsM A{};
      A.dst = dst, A.c[0] = value, A.N = N, A.nsrc = 0, A.mask = mask, A.op = 3;
      launch_flat( A, level, as_stream( stream ) );
      HH_CHECK_HIP( hipGetLastError() );
      return HYTEG_HIP_OK;
   }
   if ( ( mask & HYTEG_HIP_MASK_INNER ) && level >= HYTEG_HIP_MIN_LEVEL )
   {
      TileTable tt;
      int       rc = get_tiles( level, TILES_INNER, 1024, &tt );
      if ( rc != HYTEG_HIP_OK )
         return rc;
      if ( tt.count > 0 )
         hipLaunchKernelGGL( p1_set_inner_kernel, dim3( tt.count ), dim3( kThreads ), 0, as_stream( stream ), dst, value, tt.dev,
                             tt.count, N );
   }
   if ( mask & HYTEG_HIP_MASK_SHELL )
   {
      VecArgsM A{};
      A.dst  = dst;
      A.c[0] = value;
      A.N    = N;
      A.nsrc = 0;
      A.mask = mask & HYTEG_HIP_MASK_SHELL;
      A.op   = 3;
      hipLaunchKernelGGL( p1_vector_shell_kernel, dim3( shell_blocks( N ) ), dim3( kThreads ), 0, as_stream( stream ), A );
   }
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_dot_cell_masked( const double*      a,
                                                const double*      b,
                                                int                level,
                                                unsigned           mask,
                                                double*            result_dev,
                                                void*              workspace_dev,
                                                hyteg_hip_stream_t stream )
{
   HH_REQUIRE( a && b && result_dev && workspace_dev, "p1_dot_cell_masked: null pointer" );
   HH_REQUIRE( shell_level_ok( level ), "p1_dot_cell_masked: level out of range [0,11]" );
   double*   partial = static_cast< double* >( workspace_dev ); // [0,1024): interior, [1024,1024+256): shell
   const int N       = ( 1 << level ) + 1;
   int       n_inner = 0;
   if ( ( mask & HYTEG_HIP_MASK_INNER ) && level >= HYTEG_HIP_MIN_LEVEL )
   {
      int rc = launch_dot_inner_partial( a, b, level, partial, &n_inner, as_stream( stream ) );
      if ( rc != HYTEG_HIP_OK )
         return rc;
   }
   int n_shell = 0;
   if ( mask & HYTEG_HIP_MASK_SHELL )
   {
      const int npoints = 4 * tri( N );
      n_shell           = ( npoints + kThreads - 1 ) / kThreads;
      n_shell           = n_shell > 256 ? 256 : n_shell;
      hipLaunchKernelGGL( p1_dot_shell_kernel, dim3( n_shell ), dim3( kThreads ), 0, as_stream( stream ), a, b, N,
                          mask & HYTEG_HIP_MASK_SHELL, npoints, partial + n_inner );
   }
   if ( n_inner + n_shell == 0 )
   {
      hipLaunchKernelGGL( zero_partials_kernel, dim3( 1 ), dim3( kThreads ), 0, as_stream( stream ), result_dev, 1 );
   }
   else
      hipLaunchKernelGGL( sum_partials_kernel, dim3( 1 ), dim3( kThreads ), 0, as_stream( stream ), partial, n_inner + n_shell,
                          result_dev );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_sum_shared( double* const*     bases,
                                        const int*         group_ptr,
                                        const int*         entry_buf,
                                        const int*         entry_off,
                                        int                ngroups,
                                        int                n_writable,
                                        hyteg_hip_stream_t stream )
{
   if ( ngroups <= 0 )
      return HYTEG_HIP_OK;
   HH_REQUIRE( bases && group_ptr && entry_buf && entry_off, "sum_shared: null pointer" );
   if ( ngroups >= kSmallBatchFrom )
      hipLaunchKernelGGL( ( sum_shared_kernel< true, false, 2 > ), dim3( ( ngroups + kThreads - 1 ) / kThreads ), dim3( kThreads ), 0,
                          as_stream( stream ), bases, group_ptr, entry_buf, entry_off, ngroups, n_writable, ArrivalWait{} );
   else
      hipLaunchKernelGGL( ( sum_shared_kernel< true, false, 8 > ), dim3( ( ngroups + kThreads - 1 ) / kThreads ), dim3( kThreads ), 0,
                          as_stream( stream ), bases, group_ptr, entry_buf, entry_off, ngroups, n_writable, ArrivalWait{} );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_copy_shared( double* const*     bases,
                                         const int*         group_ptr,
                                         const int*         entry_buf,
                                         const int*         entry_off,
                                         int                ngroups,
                                         int                n_writable,
                                         hyteg_hip_stream_t stream )
{
   if ( ngroups <= 0 )
      return HYTEG_HIP_OK;
   HH_REQUIRE( bases && group_ptr && entry_buf && entry_off, "copy_shared: null pointer" );
   if ( ngroups >= kSmallBatchFrom )
      hipLaunchKernelGGL( ( sum_shared_kernel< false, false, 2 > ), dim3( ( ngroups + kThreads - 1 ) / kThreads ), dim3( kThreads ), 0,
                          as_stream( stream ), bases, group_ptr, entry_buf, entry_off, ngroups, n_writable, ArrivalWait{} );
   else
      hipLaunchKernelGGL( ( sum_shared_kernel< false, false, 8 > ), dim3( ( ngroups + kThreads - 1 ) / kThreads ), dim3( kThreads ), 0,
                          as_stream( stream ), bases, group_ptr, entry_buf, entry_off, ngroups, n_writable, ArrivalWait{} );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_reduce_shared_after_p2p( double* const*            bases,
                                                     const int*                group_ptr,
                                                     const int*                entry_buf,
                                                     const int*                entry_off,
                                                     int                       ngroups,
                                                     int                       n_writable,
                                                     int                       additive,
                                                     const unsigned long long* flags,
                                                     int                       npeers,
                                                     int                       stride,
                                                     unsigned long long        seq,
                                                     unsigned*                 status,
                                                     unsigned                  timeout_ms,
                                                     hyteg_hip_stream_t        stream )
{
   if ( ngroups <= 0 )
      return HYTEG_HIP_OK;
   HH_REQUIRE( bases && group_ptr && entry_buf && entry_off, "reduce_shared_after_p2p: null pointer" );
   HH_REQUIRE( npeers >= 0 && ( npeers == 0 || ( flags && status && stride >= 1 ) ), "reduce_shared_after_p2p: bad wait arguments" );
   const ArrivalWait W{ flags, npeers, stride, seq, status, (unsigned long long) ( timeout_ms ? timeout_ms : 20000u ) * 100000ull };
   const dim3        grid( ( ngroups + kThreads - 1 ) / kThreads ), block( kThreads );
   const bool small = ngroups >= kSmallBatchFrom;
   if ( additive && small )
      hipLaunchKernelGGL( ( sum_shared_kernel< true, true, 2 > ), grid, block, 0, as_stream( stream ), bases, group_ptr, entry_buf, entry_off,
                          ngroups, n_writable, W );
   else if ( additive )
      hipLaunchKernelGGL( ( sum_shared_kernel< true, true, 8 > ), grid, block, 0, as_stream( stream ), bases, group_ptr, entry_buf, entry_off,
                          ngroups, n_writable, W );
   else if ( small )
      hipLaunchKernelGGL( ( sum_shared_kernel< false, true, 2 > ), grid, block, 0, as_stream( stream ), bases, group_ptr, entry_buf, entry_off,
                          ngroups, n_writable, W );
   else
      hipLaunchKernelGGL( ( sum_shared_kernel< false, true, 8 > ), grid, block, 0, as_stream( stream ), bases, group_ptr, entry_buf, entry_off,
                          ngroups, n_writable, W );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_gather_entries( double*            out,
                                            double* const*     bases,
                                            const int*         entry_buf,
                                            const int*         entry_off,
                                            int                n,
                                            hyteg_hip_stream_t stream )
{
   if ( n <= 0 )
      return HYTEG_HIP_OK;
   HH_REQUIRE( out && bases && entry_buf && entry_off, "gather_entries: null pointer" );
   hipLaunchKernelGGL( gather_entries_kernel, dim3( ( n + kThreads - 1 ) / kThreads ), dim3( kThreads ), 0, as_stream( stream ), out,
                       bases, entry_buf, entry_off, n );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_copy_face_to_cell( double* cell, const double* face, int level, int v0, int v1, int v2, hyteg_hip_stream_t stream )
{
   HH_REQUIRE( cell && face, "p1_copy_face_to_cell: null pointer" );
   HH_REQUIRE( shell_level_ok( level ), "p1_copy_face_to_cell: level out of range [0,11]" );
   HH_REQUIRE( vmap_ok( v0, v1, v2 ), "p1_copy_face_to_cell: v0,v1,v2 must be distinct cell-local vertex ids" );
   const int N = ( 1 << level ) + 1;
   FaceMap   m{ { v0, v1, v2 }, 6 - v0 - v1 - v2 };
   hipLaunchKernelGGL( p1_face_cell_copy_kernel, dim3( ( tri( N ) + kThreads - 1 ) / kThreads ), dim3( kThreads ), 0, as_stream( stream ),
                       cell, face, N, m, 0, 0 );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_copy_cell_to_face( double* face, const double* cell, int level, int v0, int v1, int v2, int neighbor,
                                                  hyteg_hip_stream_t stream )
{
   HH_REQUIRE( cell && face, "p1_copy_cell_to_face: null pointer" );
   HH_REQUIRE( level >= 1 && level <= HYTEG_HIP_MAX_LEVEL, "p1_copy_cell_to_face: level out of range [1,11]" );
   HH_REQUIRE( vmap_ok( v0, v1, v2 ), "p1_copy_cell_to_face: v0,v1,v2 must be distinct cell-local vertex ids" );
   HH_REQUIRE( neighbor == 0 || neighbor == 1, "p1_copy_cell_to_face: neighbor must be 0 or 1" );
   const int N = ( 1 << level ) + 1;
   FaceMap   m{ { v0, v1, v2 }, 6 - v0 - v1 - v2 };
   hipLaunchKernelGGL( p1_face_cell_copy_kernel, dim3( ( tri( N - 1 ) + kThreads - 1 ) / kThreads ), dim3( kThreads ), 0,
                       as_stream( stream ), face, cell, N, m, 1, tri( N ) + neighbor * tri( N - 1 ) );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}

HYTEG_HIP_API int hyteg_hip_p1_apply_face3d( double*            dst_face,
                                             const double*      src_face,
                                             int                level,
                                             int                ncells,
                                             const int*         vmaps,
                                             const double*      w,
                                             int                update,
                                             hyteg_hip_stream_t stream )
{
   HH_REQUIRE( dst_face && src_face && vmaps && w, "p1_apply_face3d: null pointer" );
   HH_REQUIRE( level >= 1 && level <= HYTEG_HIP_MAX_LEVEL, "p1_apply_face3d: level out of range [1,11]" );
   HH_REQUIRE( ncells == 1 || ncells == 2, "p1_apply_face3d: a macro-face has 1 or 2 neighbour cells" );
   HH_REQUIRE( dst_face != src_face, "p1_apply_face3d: src and dst must not alias" );
   HH_REQUIRE( update == HYTEG_HIP_REPLACE || update == HYTEG_HIP_ADD, "p1_apply_face3d: bad update type" );
   FaceApplyArgs A{};
   A.dst    = dst_face;
   A.src    = src_face;
   A.N      = ( 1 << level ) + 1;
   A.ncells = ncells;
   A.update = update;
   for ( int k = 0; k < ncells; ++k )
   {
      HH_REQUIRE( vmap_ok( vmaps[3 * k], vmaps[3 * k + 1], vmaps[3 * k + 2] ), "p1_apply_face3d: bad vertex map" );
      A.map[k] = FaceMap{ { vmaps[3 * k], vmaps[3 * k + 1], vmaps[3 * k + 2] }, 6 - vmaps[3 * k] - vmaps[3 * k + 1] - vmaps[3 * k + 2] };
      for ( int s = 0; s < 15; ++s )
         A.w[k][s] = w[15 * k + s];
   }
   hipLaunchKernelGGL( p1_apply_face3d_kernel, dim3( ( tri( A.N ) + kThreads - 1 ) / kThreads ), dim3( kThreads ), 0, as_stream( stream ), A );
   HH_CHECK_HIP( hipGetLastError() );
   return HYTEG_HIP_OK;
}
}
