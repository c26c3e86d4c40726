// hyteg_host.hpp -- C++ host layer above the C-ABI (include/hyteg_hip.h).
//
// Mirrors the names and call signatures of the reference classes on the P1 hot path so that code written
// against HyTeG reads the same here:
//   MeshInfo                     src/hyteg/mesh/MeshInfo.hpp:221,512-585
//   PrimitiveStorage             src/hyteg/primitivestorage/PrimitiveStorage.hpp, SetupPrimitiveStorage.cpp
//   P1Function< double >         src/hyteg/p1functionspace/VertexDoFFunction.hpp/.cpp (interpolate, assign, add,
//                                multElementwise, dotLocal, dotGlobal)
//   P1ConstantOperator< Form >   src/constant_stencil_operator/P1ConstantOperator.hpp:33-168 + P1Operator.hpp:192-447
//   P1toP1LinearRestriction / P1toP1LinearProlongation   src/hyteg/gridtransferoperators/
//   WeightedJacobiSmoother, GaussSeidelSmoother, SORSmoother, CGSolver, GeometricMultigridSolver  src/hyteg/solvers/
//
// Data model: one device array per (macro-cell, level) in HyTeG's cell layout.  The DoFs
// of macro-faces/edges/vertices are the boundary entries of every adjacent cell array and are kept bit-identical
// in all copies.  Operators evaluate them as per-cell partial results + one additive exchange ("sumShared"),
// which is what HyTeG itself does for elementwise operators and grid transfer
// (P1ElementwiseOperator.cpp:186-188, P1toP1LinearRestriction.cpp:343-345), instead of the six directed ghost
// phases of P1Operator.hpp:201-207.  All device work goes through the C-ABI; this file launches nothing itself.
#pragma once

#include <algorithm>
#include <atomic>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/hyteg_hip.h"

namespace hyteg {

using real_t = double;
using uint_t = std::size_t;
using Point3D = std::array< double, 3 >;

// ---- src/hyteg/types/types.hpp:29-77 ------------------------------------------------------------------
enum UpdateType
{
   Replace = 0,
   Add     = 1
};
enum DoFType : std::size_t
{
   None              = 0,
   All               = 1 + 2 + 4 + 8,
   Boundary          = 2 + 4 + 8,
   Inner             = 1,
   DirichletBoundary = 2,
   NeumannBoundary   = 4,
   FreeslipBoundary  = 8
};
inline DoFType operator|( DoFType a, DoFType b ) { return DoFType( std::size_t( a ) | std::size_t( b ) ); }
inline DoFType operator&( DoFType a, DoFType b ) { return DoFType( std::size_t( a ) & std::size_t( b ) ); }
inline DoFType operator^( DoFType a, DoFType b ) { return DoFType( std::size_t( a ) ^ std::size_t( b ) ); }
inline bool    testFlag( DoFType a, DoFType b ) { return ( a & b ) != 0; }
enum class CycleType
{
   VCYCLE,
   WCYCLE
};

// the reference aborts on failure (WALBERLA_ABORT); the host layer throws, the C facade turns it into a code
inline void hipCheck( int rc, const char* what )
{
   if ( rc != HYTEG_HIP_OK )
      throw std::runtime_error( std::string( what ) + ": " + hyteg_hip_last_error() );
}

// identity of functions and operators that outlives address reuse (keys of recorded launch graphs)
inline uint64_t nextUid()
{
   static std::atomic< uint64_t > counter{ 1 };
   return counter.fetch_add( 1 );
}

namespace layout {
inline int64_t width( int level ) { return ( int64_t( 1 ) << level ) + 1; }
inline int64_t tet( int64_t w ) { return w * ( w + 1 ) * ( w + 2 ) / 6; }
inline int64_t cellSize( int level ) { return tet( width( level ) ); }
inline int64_t cellIndex( int64_t N, int64_t x, int64_t y, int64_t z )
{
   const int64_t W = N - z;
   return tet( N ) - tet( W ) + y * W - y * ( y - 1 ) / 2 + x;
}
} // namespace layout

// =====================================================================================================
// MeshInfo: vertices + tetrahedra.  Readers for Gmsh ASCII 2.2 and 4.1 (tetrahedra = element type 4).
// =====================================================================================================
class MeshInfo
{
 public:
   std::vector< Point3D >              vertices;
   std::vector< std::array< int, 4 > > cells; // indices into vertices

   static MeshInfo singleTetrahedron( const std::array< Point3D, 4 >& c )
   {
      MeshInfo m;
      m.vertices.assign( c.begin(), c.end() );
      m.cells.push_back( { 0, 1, 2, 3 } );
      return m;
   }

   static MeshInfo fromArrays( int nv, const double* v, int nc, const int* c )
   {
      MeshInfo m;
      for ( int i = 0; i < nv; ++i )
         m.vertices.push_back( { v[3 * i], v[3 * i + 1], v[3 * i + 2] } );
      for ( int i = 0; i < nc; ++i )
         m.cells.push_back( { c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3] } );
      return m;
   }

   static MeshInfo fromGmshFile( const std::string& path )
   {
      std::ifstream in( path );
      if ( !in )
         throw std::runtime_error( "MeshInfo::fromGmshFile: cannot open " + path );
      std::string line;
      double      version = 0;
      MeshInfo    m;
      std::map< long, int > nodeIndex;
      while ( std::getline( in, line ) )
      {
         if ( line.rfind( "$MeshFormat", 0 ) == 0 )
         {
            int ft, ds;
            in >> version >> ft >> ds;
            if ( ft != 0 )
               throw std::runtime_error( "MeshInfo::fromGmshFile: only ASCII meshes are supported" );
         }
         else if ( line.rfind( "$Nodes", 0 ) == 0 )
         {
            if ( version < 4.0 )
            {
               long n;
               in >> n;
               for ( long i = 0; i < n; ++i )
               {
                  long   id;
                  double x, y, z;
                  in >> id >> x >> y >> z;
                  nodeIndex[id] = (int) m.vertices.size();
                  m.vertices.push_back( { x, y, z } );
               }
            }
            else
            {
               long nblocks, nnodes, minTag, maxTag;
               in >> nblocks >> nnodes >> minTag >> maxTag;
               for ( long b = 0; b < nblocks; ++b )
               {
                  int  dim, tag, parametric;
                  long nb;
                  in >> dim >> tag >> parametric >> nb;
                  std::vector< long > ids( nb );
                  for ( auto& id : ids )
                     in >> id;
                  for ( long i = 0; i < nb; ++i )
                  {
                     double x, y, z;
                     in >> x >> y >> z;
                     nodeIndex[ids[i]] = (int) m.vertices.size();
                     m.vertices.push_back( { x, y, z } );
                  }
               }
            }
         }
         else if ( line.rfind( "$Elements", 0 ) == 0 )
         {
            if ( version < 4.0 )
            {
               long n;
               in >> n;
               std::getline( in, line );
               for ( long i = 0; i < n; ++i )
               {
                  std::getline( in, line );
                  std::istringstream ls( line );
                  long               id;
                  int                type, ntags;
                  ls >> id >> type >> ntags;
                  for ( int t = 0; t < ntags; ++t )
                  {
                     long tag;
                     ls >> tag;
                  }
                  if ( type == 4 )
                  {
                     long a, b, c, d;
                     ls >> a >> b >> c >> d;
                     m.cells.push_back( { nodeIndex.at( a ), nodeIndex.at( b ), nodeIndex.at( c ), nodeIndex.at( d ) } );
                  }
               }
            }
            else
            {
               long nblocks, nel, minTag, maxTag;
               in >> nblocks >> nel >> minTag >> maxTag;
               for ( long b = 0; b < nblocks; ++b )
               {
                  int  dim, tag, type;
                  long nb;
                  in >> dim >> tag >> type >> nb;
                  const int nn = type == 15 ? 1 : type == 1 ? 2 : type == 2 ? 3 : type == 4 ? 4 : -1;
                  if ( nn < 0 )
                     throw std::runtime_error( "MeshInfo::fromGmshFile: unsupported element type" );
                  for ( long i = 0; i < nb; ++i )
                  {
                     long id, nd[4];
                     in >> id;
                     for ( int k = 0; k < nn; ++k )
                        in >> nd[k];
                     if ( type == 4 )
                        m.cells.push_back(
                            { nodeIndex.at( nd[0] ), nodeIndex.at( nd[1] ), nodeIndex.at( nd[2] ), nodeIndex.at( nd[3] ) } );
                  }
               }
            }
         }
      }
      if ( m.cells.empty() )
         throw std::runtime_error( "MeshInfo::fromGmshFile: no tetrahedra in " + path );
      return m;
   }
};

// =====================================================================================================
// PrimitiveStorage: macro-vertices / edges / faces / cells, neighbourhood, boundary flags, rank assignment.
// Cell-local numbering (src/hyteg/primitives/Cell.hpp, src/hyteg/indexing/MacroCellIndexing.cpp:36-91):
// faces 0:(0,1,2) 1:(0,1,3) 2:(0,2,3) 3:(1,2,3); edges 0:(0,1) 1:(0,2) 2:(1,2) 3:(0,3) 4:(1,3) 5:(2,3).
// Slot order of all per-cell 14-arrays: { edge0..5, face0..3, vertex0..3 } (the grid-transfer kernels' order).
// =====================================================================================================
struct MacroCell
{
   int                      id;
   std::array< int, 4 >     v;      // global vertex ids, local order = mesh order
   std::array< Point3D, 4 > coords; // getCoordinates()
   std::array< int, 6 >     edges;  // global edge ids by local edge
   std::array< int, 4 >     faces;  // global face ids by local face
   int                      rank;
   int                      localIndex; // index among this rank's cells, -1 if remote
};
struct MacroPrimitive
{
   std::vector< int > v;     // sorted global vertex ids (1, 2 or 3)
   std::vector< int > cells; // adjacent global cell ids, ascending
   bool               onBoundary = false;
   uint_t             getNumNeighborCells() const { return cells.size(); }
};

// callbacks for storages distributed over several ranks (set by the embedding application; see hyteg_amd/host.py)
struct CommHooks
{
   // all-to-all of the packed partial values of one (level, boundary class); buffers were registered before.
   // Begin may return before the data has arrived (so that interior kernels overlap the transfer); End waits.
   void ( *exchangeBegin )( void* user, int level, int cls ) = nullptr;
   void ( *exchangeEnd )( void* user, int level, int cls )   = nullptr;
   // in-place sum over all ranks of n doubles in host memory (walberla::mpi::allReduceInplace, VertexDoFFunction.cpp:1717)
   void ( *allreduceSum )( void* user, double* values, int n ) = nullptr;
   void* user                                                   = nullptr;
};

static const int kCellFaceVerts[4][3] = { { 0, 1, 2 }, { 0, 1, 3 }, { 0, 2, 3 }, { 1, 2, 3 } };
static const int kCellEdgeVerts[6][2] = { { 0, 1 }, { 0, 2 }, { 1, 2 }, { 0, 3 }, { 1, 3 }, { 2, 3 } };

class PrimitiveStorage
{
 public:
   PrimitiveStorage( const MeshInfo& mesh, int rank = 0, int nranks = 1 )
   : rank_( rank )
   , nranks_( nranks )
   {
      if ( nranks < 1 || rank < 0 || rank >= nranks )
         throw std::runtime_error( "PrimitiveStorage: bad rank / number of ranks" );
      std::map< std::vector< int >, int > edgeId, faceId;
      vertices_.resize( mesh.vertices.size() );
      for ( uint_t i = 0; i < vertices_.size(); ++i )
         vertices_[i].v = { (int) i };
      for ( uint_t c = 0; c < mesh.cells.size(); ++c )
      {
         MacroCell cell;
         cell.id = (int) c;
         cell.v  = mesh.cells[c];
         for ( int k = 0; k < 4; ++k )
         {
            if ( cell.v[k] < 0 || cell.v[k] >= (int) mesh.vertices.size() )
               throw std::runtime_error( "PrimitiveStorage: cell refers to a missing vertex" );
            cell.coords[k] = mesh.vertices[cell.v[k]];
            vertices_[cell.v[k]].cells.push_back( (int) c );
         }
         for ( int e = 0; e < 6; ++e )
         {
            std::vector< int > key = { cell.v[kCellEdgeVerts[e][0]], cell.v[kCellEdgeVerts[e][1]] };
            std::sort( key.begin(), key.end() );
            auto it = edgeId.find( key );
            if ( it == edgeId.end() )
            {
               it = edgeId.emplace( key, (int) edges_.size() ).first;
               edges_.push_back( MacroPrimitive{ key, {}, false } );
            }
            cell.edges[e] = it->second;
            edges_[it->second].cells.push_back( (int) c );
         }
         for ( int f = 0; f < 4; ++f )
         {
            std::vector< int > key = { cell.v[kCellFaceVerts[f][0]], cell.v[kCellFaceVerts[f][1]], cell.v[kCellFaceVerts[f][2]] };
            std::sort( key.begin(), key.end() );
            auto it = faceId.find( key );
            if ( it == faceId.end() )
            {
               it = faceId.emplace( key, (int) faces_.size() ).first;
               faces_.push_back( MacroPrimitive{ key, {}, false } );
            }
            cell.faces[f] = it->second;
            faces_[it->second].cells.push_back( (int) c );
         }
         // SetupPrimitiveStorage's default balancing is round robin over ranks (loadbalancing/SimpleBalancer.cpp: roundRobin)
         cell.rank       = (int) ( c % (uint_t) nranks );
         cell.localIndex = -1;
         cells_.push_back( cell );
      }
      // setMeshBoundaryFlagsOnBoundary( 1, 0, true ): a face with one neighbour cell is on the boundary, and so is
      // every edge / vertex of such a face (SetupPrimitiveStorage.cpp, onBoundary())
      for ( auto& f : faces_ )
      {
         if ( f.cells.size() > 2 )
            throw std::runtime_error( "PrimitiveStorage: face with more than two neighbour cells" );
         f.onBoundary = f.cells.size() == 1;
         if ( f.onBoundary )
         {
            for ( int a = 0; a < 3; ++a )
            {
               vertices_[f.v[a]].onBoundary = true;
               for ( int b = a + 1; b < 3; ++b )
               {
                  std::vector< int > key = { f.v[a], f.v[b] };
                  edges_[edgeId.at( key )].onBoundary = true;
               }
            }
         }
      }
      for ( auto& c : cells_ )
         if ( c.rank == rank_ )
         {
            c.localIndex = (int) localCells_.size();
            localCells_.push_back( c.id );
         }
      for ( auto& p : vertices_ )
         std::sort( p.cells.begin(), p.cells.end() );
      // no device work here: topology and exchange plans can be built (and tested) without a GPU
   }
   ~PrimitiveStorage()
   {
      if ( dotResult_ )
         hyteg_hip_free( dotResult_ );
      if ( dotWorkspace_ )
         hyteg_hip_free( dotWorkspace_ );
      for ( void* p : scratchAll_ )
         hyteg_hip_free( p );
   }
   PrimitiveStorage( const PrimitiveStorage& )            = delete;
   PrimitiveStorage& operator=( const PrimitiveStorage& ) = delete;

   bool hasGlobalCells() const { return !cells_.empty(); }
   int  rank() const { return rank_; }
   int  numRanks() const { return nranks_; }

   const std::vector< MacroCell >&      getCells() const { return cells_; }
   const std::vector< MacroPrimitive >& getFaces() const { return faces_; }
   const std::vector< MacroPrimitive >& getEdges() const { return edges_; }
   const std::vector< MacroPrimitive >& getVertices() const { return vertices_; }
   const std::vector< int >&            getLocalCellIDs() const { return localCells_; }
   uint_t                               getNumberOfLocalCells() const { return localCells_.size(); }
   const MacroCell&                     getLocalCell( uint_t i ) const { return cells_[localCells_.at( i )]; }

   // boundary type of every primitive on the domain boundary (BoundaryCondition::create0123BC maps flag 1 -> Dirichlet)
   void    setBoundaryType( DoFType t ) { boundaryType_ = t; }
   DoFType boundaryTypeOf( bool onBoundary ) const { return onBoundary ? boundaryType_ : Inner; }

   // the macro-primitive behind slot s (0..13) of a cell
   const MacroPrimitive& primitiveOfSlot( const MacroCell& c, int s ) const
   {
      if ( s < 6 )
         return edges_[c.edges[s]];
      if ( s < 10 )
         return faces_[c.faces[s - 6]];
      return vertices_[c.v[s - 10]];
   }

   // point mask of a cell for a DoFType flag: the per-primitive test of P1Operator.hpp:213-303
   unsigned maskFor( const MacroCell& c, DoFType flag ) const
   {
      unsigned m = testFlag( Inner, flag ) ? HYTEG_HIP_MASK_INNER : 0u; // a macro-cell is never on the mesh boundary
      for ( int s = 0; s < 14; ++s )
         if ( testFlag( boundaryTypeOf( primitiveOfSlot( c, s ).onBoundary ), flag ) )
            m |= 1u << s;
      return m;
   }
   // like maskFor but a shared primitive is counted by its lowest-numbered neighbour cell only (dot products)
   unsigned ownedMaskFor( const MacroCell& c, DoFType flag ) const
   {
      unsigned m = maskFor( c, flag );
      for ( int s = 0; s < 14; ++s )
         if ( primitiveOfSlot( c, s ).cells.front() != c.id )
            m &= ~( 1u << s );
      return m;
   }
   // numNeighborCells of the 14 primitives around a cell, in the grid-transfer kernels' argument order
   std::array< double, 14 > numNeighborCells( const MacroCell& c ) const
   {
      std::array< double, 14 > n{};
      for ( int s = 0; s < 14; ++s )
         n[s] = (double) primitiveOfSlot( c, s ).cells.size();
      return n;
   }

   // ---- batched launches (p1_batch.hip): one launch for all local cells on the levels where a cell is small ----
   // Default: levels <= 6 whenever the rank owns more than one cell (a single cell is served better by the tuned per-cell
   // kernels: measured 1.03 vs 1.32 ms per V(3,3) Jacobi cycle); HYTEG_AMD_BATCH_MAX_LEVEL overrides (-1 disables).
   bool useBatch( uint_t level ) const
   {
      if ( batchMaxLevel_ == -2 )
      {
         const char* e  = std::getenv( "HYTEG_AMD_BATCH_MAX_LEVEL" );
         batchMaxLevel_ = e ? std::atoi( e ) : 6;
         const char* s  = std::getenv( "HYTEG_AMD_BATCH_SINGLE_MAX_LEVEL" );
         batchSingleMaxLevel_ = s ? std::atoi( s ) : kBatchSingleMaxLevelDefault;
      }
      if ( localCells_.size() == 1 )
         return (int) level <= std::min( batchMaxLevel_, batchSingleMaxLevel_ );
      return localCells_.size() > 1 && (int) level <= batchMaxLevel_;
   }
   // the one-workgroup Gauss-Seidel sweep of small cells (levels <= 5) also pays off for a single cell: 1 launch instead of ~3n
   bool useBatchSor( uint_t level ) const
   {
      return useBatch( level ) || ( !localCells_.empty() && level <= 5 && batchMaxLevel_ >= 0 && (int) level <= batchMaxLevel_ );
   }
   void setBatchMaxLevel( int l ) { batchMaxLevel_ = l; }
   std::vector< unsigned > masksFor( DoFType flag, bool owned = false, unsigned keep = HYTEG_HIP_MASK_ALL ) const
   {
      std::vector< unsigned > m;
      for ( int id : localCells_ )
         m.push_back( ( owned ? ownedMaskFor( cells_[id], flag ) : maskFor( cells_[id], flag ) ) & keep );
      return m;
   }
   // device table [local cell][14] of 1 / numNeighborCells (grid transfer)
   const double* nncInvDevice() const
   {
      if ( !nncInv_ && !localCells_.empty() )
      {
         std::vector< double > h;
         for ( int id : localCells_ )
            for ( double n : numNeighborCells( cells_[id] ) )
               h.push_back( 1.0 / n );
         nncInv_ = uploadTable( h );
      }
      return nncInv_;
   }
   // small read-only device table owned by the storage (freed with it)
   void* uploadBytes( const void* h, size_t bytes ) const
   {
      void* p = nullptr;
      hipCheck( hyteg_hip_malloc( &p, std::max< size_t >( 8, bytes ) ), "uploadTable: malloc" );
      hipCheck( hyteg_hip_upload( p, h, bytes, stream_ ), "uploadTable: upload" );
      hipCheck( hyteg_hip_stream_synchronize( stream_ ), "uploadTable: sync" );
      scratchAll_.push_back( p );
      return p;
   }
   double* uploadTable( const std::vector< double >& h ) const
   {
      return static_cast< double* >( uploadBytes( h.data(), h.size() * sizeof( double ) ) );
   }
   // calls fn( first, count ) for chunks of at most HYTEG_HIP_MAX_BATCH local cells
   template < typename F >
   void forCellChunks( F&& fn ) const
   {
      const int n = (int) localCells_.size();
      for ( int first = 0; first < n; first += HYTEG_HIP_MAX_BATCH )
         fn( first, std::min( HYTEG_HIP_MAX_BATCH, n - first ) );
   }

   void              setStream( hyteg_hip_stream_t s ) { stream_ = s; }
   hyteg_hip_stream_t stream() const { return stream_; }
   void              setCommHooks( const CommHooks& h ) { hooks_ = h; }
   const CommHooks&  hooks() const { return hooks_; }

   // pool of scratch device arrays keyed by size, so that operators can use temporaries without hipMalloc/hipFree
   // in the hot path (the role of hyteg::getTemporaryFunction, src/hyteg/memory/TempFunctionManager.hpp)
   double* acquireScratch( size_t doubles ) const
   {
      auto& freeList = scratchFree_[doubles];
      if ( !freeList.empty() )
      {
         double* p = freeList.back();
         freeList.pop_back();
         return p;
      }
      void* p = nullptr;
      hipCheck( hyteg_hip_malloc( &p, doubles * sizeof( double ) ), "scratch: malloc" );
      scratchAll_.push_back( p );
      return static_cast< double* >( p );
   }
   void releaseScratch( size_t doubles, double* p ) const { scratchFree_[doubles].push_back( p ); }

   // device copy of a list of device pointers (the "bases" argument of the exchange kernels), cached by content: scratch
   // functions get the same arrays from the pool again and again, so after the first cycle nothing is allocated or
   // uploaded in the hot path (and the path can be recorded into a launch graph)
   double** pointerTable( const std::vector< double* >& host ) const
   {
      auto it = pointerTables_.find( host );
      if ( it != pointerTables_.end() )
         return it->second;
      void* d = nullptr;
      hipCheck( hyteg_hip_malloc( &d, std::max< size_t >( 1, host.size() ) * sizeof( double* ) ), "bases: malloc" );
      // on the null stream and complete on return: valid for whatever stream uses the table next, and legal while the
      // storage's stream is being recorded into a launch graph
      hipCheck( hyteg_hip_upload( d, host.data(), host.size() * sizeof( double* ), nullptr ), "bases: upload" );
      hipCheck( hyteg_hip_stream_synchronize( nullptr ), "bases: sync" );
      scratchAll_.push_back( d );
      pointerTables_[host] = static_cast< double** >( d );
      return static_cast< double** >( d );
   }

   double* dotResult() const
   {
      if ( !dotResult_ )
      {
         hipCheck( hyteg_hip_malloc( &dotResult_, sizeof( double ) * std::max< size_t >( 1, localCells_.size() ) ), "PrimitiveStorage: malloc" );
         hipCheck( hyteg_hip_malloc( &dotWorkspace_, hyteg_hip_dot_workspace_bytes() ), "PrimitiveStorage: malloc" );
      }
      return static_cast< double* >( dotResult_ );
   }
   void* dotWorkspace() const
   {
      dotResult();
      return dotWorkspace_;
   }

   // ---------------------------------------------------------------------------------------------------
   // Additive exchange plan of one (level, boundary class): every DoF on a macro-face/edge/vertex with at least
   // two neighbour cells, at least one of them local, is a group; its entries are its copies in ascending global
   // cell order.  cls 0: primitives in the interior of the domain, cls 1: primitives on the domain boundary.
   // ---------------------------------------------------------------------------------------------------
   struct ExchangePlan
   {
      // host copies
      std::vector< int > groupPtr, entryBuf, entryOff; // entryBuf < nLocal: local cell; else nLocal + peer slot
      std::vector< int > peers;                        // ranks we exchange with, ascending
      std::vector< int > sendCount, recvCount;         // per peer
      std::vector< int > sendBuf, sendOff;             // concatenated per peer: (local cell, offset)
      // device copies
      int *dGroupPtr = nullptr, *dEntryBuf = nullptr, *dEntryOff = nullptr, *dSendBuf = nullptr, *dSendOff = nullptr;
      // communication buffers (device), registered by the application for multi-rank runs or allocated here
      double *sendBuffer = nullptr, *recvBuffer = nullptr;
      bool    ownsBuffers = false;
      bool    onDevice    = false;
      int     ngroups() const { return (int) groupPtr.size() - 1; }
      int     totalSend() const { return (int) sendBuf.size(); }
      int     totalRecv() const
      {
         int t = 0;
         for ( int r : recvCount )
            t += r;
         return t;
      }
   };

   // host part of the plan (no GPU needed)
   // dofKind 0: vertex DoFs (P1 arrays); 1: edge DoFs (the edge-DoF arrays of P2 functions)
   const ExchangePlan& exchangePlan( int level, int cls, int dofKind = 0 ) const
   {
      auto key = std::make_pair( level, cls + 2 * dofKind );
      auto it  = plans_.find( key );
      if ( it == plans_.end() )
         it = plans_.emplace( key, buildPlan( level, cls, dofKind ) ).first;
      return it->second;
   }
   // plan with its index arrays (and default communication buffers) resident on the device
   const ExchangePlan& devicePlan( int level, int cls, int dofKind = 0 ) const
   {
      auto& P = const_cast< ExchangePlan& >( exchangePlan( level, cls, dofKind ) );
      if ( !P.onDevice )
      {
         P.dGroupPtr = uploadVector( P.groupPtr );
         P.dEntryBuf = uploadVector( P.entryBuf );
         P.dEntryOff = uploadVector( P.entryOff );
         P.dSendBuf  = uploadVector( P.sendBuf );
         P.dSendOff  = uploadVector( P.sendOff );
         if ( !P.sendBuffer && ( P.totalSend() > 0 || P.totalRecv() > 0 ) )
         {
            void *s = nullptr, *r = nullptr;
            hipCheck( hyteg_hip_malloc( &s, std::max( 1, P.totalSend() ) * sizeof( double ) ), "plan: malloc" );
            hipCheck( hyteg_hip_malloc( &r, std::max( 1, P.totalRecv() ) * sizeof( double ) ), "plan: malloc" );
            P.sendBuffer  = static_cast< double* >( s );
            P.recvBuffer  = static_cast< double* >( r );
            P.ownsBuffers = true;
         }
         P.onDevice = true;
      }
      return P;
   }
   // multi-rank: the application owns the communication buffers (e.g. torch tensors) and registers them here
   void registerCommBuffers( int level, int cls, double* send, double* recv ) const
   {
      auto& p = const_cast< ExchangePlan& >( exchangePlan( level, cls ) );
      if ( p.ownsBuffers )
      {
         hyteg_hip_free( p.sendBuffer );
         hyteg_hip_free( p.recvBuffer );
         p.ownsBuffers = false;
      }
      p.sendBuffer = send;
      p.recvBuffer = recv;
   }

 private:
   template < typename T >
   static T* uploadVector( const std::vector< T >& v )
   {
      if ( v.empty() )
         return nullptr;
      void* d = nullptr;
      hipCheck( hyteg_hip_malloc( &d, v.size() * sizeof( T ) ), "upload: malloc" );
      hipCheck( hyteg_hip_upload( d, v.data(), v.size() * sizeof( T ), nullptr ), "upload: copy" );
      hipCheck( hyteg_hip_stream_synchronize( nullptr ), "upload: sync" );
      return static_cast< T* >( d );
   }

   // array index inside cell `c` of the point with barycentric weights w[k] on the primitive's vertices p.v[k]
   static int64_t indexInCell( const MacroCell& c, const MacroPrimitive& p, const int* w, int64_t N )
   {
      int64_t bary[4] = { 0, 0, 0, 0 };
      for ( uint_t k = 0; k < p.v.size(); ++k )
      {
         int l = -1;
         for ( int q = 0; q < 4; ++q )
            if ( c.v[q] == p.v[k] )
               l = q;
         if ( l < 0 )
            throw std::runtime_error( "indexInCell: primitive is not part of the cell" );
         bary[l] = w[k];
      }
      return layout::cellIndex( N, bary[1], bary[2], bary[3] );
   }

   // array index in the edge-DoF array of cell `c` of the edge DoF between the points with barycentric weights wa, wb on the
   // primitive's vertices (edgedof::calcEdgeDoFIndex / calcEdgeDoFOrientation, EdgeDoFIndexing.hpp:89-165, + macrocell::index)
   static int64_t edgeIndexInCell( const MacroCell& c, const MacroPrimitive& p, const int* wa, const int* wb, int level )
   {
      int64_t a[4] = { 0, 0, 0, 0 }, b[4] = { 0, 0, 0, 0 };
      for ( uint_t k = 0; k < p.v.size(); ++k )
      {
         int l = -1;
         for ( int q = 0; q < 4; ++q )
            if ( c.v[q] == p.v[k] )
               l = q;
         if ( l < 0 )
            throw std::runtime_error( "edgeIndexInCell: primitive is not part of the cell" );
         a[l] = wa[k], b[l] = wb[k];
      }
      const int64_t* A  = a + 1; // (x, y, z) = weights of cell vertices 1, 2, 3
      const int64_t* B  = b + 1;
      const int64_t  d0 = B[0] - A[0], d1 = B[1] - A[1], d2 = B[2] - A[2];
      const int64_t  n  = int64_t( 1 ) << level;
      int            o;
      int64_t        e[3];
      auto           lower = [&]( int axis ) { return A[axis] < B[axis] ? A : B; };
      if ( d1 == 0 && d2 == 0 )
         o = 0, e[0] = lower( 0 )[0], e[1] = lower( 0 )[1], e[2] = lower( 0 )[2];
      else if ( d0 == 0 && d2 == 0 )
         o = 1, e[0] = lower( 1 )[0], e[1] = lower( 1 )[1], e[2] = lower( 1 )[2];
      else if ( d0 == 0 && d1 == 0 )
         o = 2, e[0] = lower( 2 )[0], e[1] = lower( 2 )[1], e[2] = lower( 2 )[2];
      else if ( d2 == 0 )
         o = 3, e[0] = lower( 0 )[0], e[1] = lower( 0 )[1] - 1, e[2] = lower( 0 )[2];
      else if ( d1 == 0 )
         o = 4, e[0] = lower( 0 )[0], e[1] = lower( 0 )[1], e[2] = lower( 0 )[2] - 1;
      else if ( d0 == 0 )
         o = 5, e[0] = lower( 1 )[0], e[1] = lower( 1 )[1], e[2] = lower( 1 )[2] - 1;
      else
         o = 6, e[0] = lower( 0 )[0], e[1] = lower( 0 )[1] - 1, e[2] = lower( 0 )[2];
      return o * layout::tet( n ) + layout::cellIndex( o == 6 ? n - 1 : n, e[0], e[1], e[2] );
   }

   ExchangePlan buildPlan( int level, int cls, int dofKind = 0 ) const
   {
      ExchangePlan  P;
      const int64_t N = layout::width( level ), n = N - 1;
      const int     nLocal = (int) localCells_.size();
      // peers: ranks of remote cells sharing a primitive of this class with a local cell
      std::set< int > peerSet;
      auto            involves = [&]( const MacroPrimitive& p, bool& local ) {
         local = false;
         if ( p.cells.size() < 2 || ( p.onBoundary ? 1 : 0 ) != cls )
            return false;
         for ( int c : p.cells )
            local = local || cells_[c].rank == rank_;
         return true;
      };
      auto forAllPrimitives = [&]( auto&& fn ) {
         for ( const auto& p : faces_ )
            fn( p );
         for ( const auto& p : edges_ )
            fn( p );
         for ( const auto& p : vertices_ )
            fn( p );
      };
      forAllPrimitives( [&]( const MacroPrimitive& p ) {
         bool local;
         if ( involves( p, local ) && local )
            for ( int c : p.cells )
               if ( cells_[c].rank != rank_ )
                  peerSet.insert( cells_[c].rank );
      } );
      P.peers.assign( peerSet.begin(), peerSet.end() );
      std::map< int, int > peerSlot;
      for ( uint_t i = 0; i < P.peers.size(); ++i )
         peerSlot[P.peers[i]] = (int) i;
      P.sendCount.assign( P.peers.size(), 0 );
      P.recvCount.assign( P.peers.size(), 0 );
      std::vector< std::vector< int > > sendBufPer( P.peers.size() ), sendOffPer( P.peers.size() );

      // enumerate the DoFs that belong to a primitive in a rank-independent order: fn( wa, wb ) with the barycentric
      // weights of the point (vertex DoF, wb unused) or of the two end points of the micro-edge (edge DoF)
      auto pointsOf = [&]( const MacroPrimitive& p, auto&& fn ) {
         if ( dofKind == 1 )
         {
            if ( p.v.size() == 3 )
            {
               // micro-edges in the plane of the face whose end points do not lie on one and the same macro-edge of the face
               auto emit = [&]( int64_t i0, int64_t j0, int64_t i1, int64_t j1 ) {
                  const int wa[3] = { (int) ( n - i0 - j0 ), (int) i0, (int) j0 }, wb[3] = { (int) ( n - i1 - j1 ), (int) i1, (int) j1 };
                  for ( int k = 0; k < 3; ++k )
                     if ( wa[k] == 0 && wb[k] == 0 )
                        return;
                  fn( wa, wb );
               };
               for ( int64_t j = 0; j <= n; ++j )
                  for ( int64_t i = 0; i + j <= n; ++i )
                  {
                     if ( i + j + 1 <= n )
                     {
                        emit( i, j, i + 1, j );
                        emit( i, j, i, j + 1 );
                        emit( i + 1, j, i, j + 1 );
                     }
                  }
            }
            else if ( p.v.size() == 2 )
            {
               for ( int64_t i = 0; i <= n - 1; ++i )
               {
                  const int wa[2] = { (int) ( n - i ), (int) i }, wb[2] = { (int) ( n - i - 1 ), (int) ( i + 1 ) };
                  fn( wa, wb );
               }
            }
            return;
         }
         if ( p.v.size() == 3 )
         {
            for ( int64_t j = 1; j <= n - 2; ++j )
               for ( int64_t i = 1; i + j <= n - 1; ++i )
               {
                  const int w[3] = { (int) ( n - i - j ), (int) i, (int) j };
                  fn( w, w );
               }
         }
         else if ( p.v.size() == 2 )
         {
            for ( int64_t i = 1; i <= n - 1; ++i )
            {
               const int w[2] = { (int) ( n - i ), (int) i };
               fn( w, w );
            }
         }
         else
         {
            const int w[1] = { (int) n };
            fn( w, w );
         }
      };

      // first pass: receive offsets.  The data a peer sends us is ordered by (primitive, point, entry) over all
      // groups that involve both ranks -- the same loop the peer runs to fill its send buffer.
      std::vector< int > recvCursor( P.peers.size(), 0 );
      P.groupPtr.push_back( 0 );
      forAllPrimitives( [&]( const MacroPrimitive& p ) {
         bool local;
         if ( !involves( p, local ) || !local )
            return;
         pointsOf( p, [&]( const int* w, const int* wb ) {
            for ( int c : p.cells )
            {
               const MacroCell& cell = cells_[c];
               if ( cell.rank == rank_ )
               {
                  const int off = dofKind == 1 ? (int) edgeIndexInCell( cell, p, w, wb, level ) : (int) indexInCell( cell, p, w, N );
                  P.entryBuf.push_back( cell.localIndex );
                  P.entryOff.push_back( off );
                  // this value goes to every peer that shares the group
                  std::set< int > dests;
                  for ( int c2 : p.cells )
                     if ( cells_[c2].rank != rank_ )
                        dests.insert( cells_[c2].rank );
                  for ( int d : dests )
                  {
                     sendBufPer[peerSlot[d]].push_back( cell.localIndex );
                     sendOffPer[peerSlot[d]].push_back( off );
                  }
               }
               else
               {
                  const int s = peerSlot[cell.rank];
                  P.entryBuf.push_back( nLocal + s );
                  P.entryOff.push_back( recvCursor[s]++ );
               }
            }
            P.groupPtr.push_back( (int) P.entryBuf.size() );
         } );
      } );
      // receive buffer = concatenation over peers: turn per-peer offsets into offsets relative to the peer's segment;
      // bases[nLocal + s] points at the start of peer s's segment, so the offsets stay as they are.
      for ( uint_t s = 0; s < P.peers.size(); ++s )
      {
         P.recvCount[s] = recvCursor[s];
         P.sendCount[s] = (int) sendBufPer[s].size();
         P.sendBuf.insert( P.sendBuf.end(), sendBufPer[s].begin(), sendBufPer[s].end() );
         P.sendOff.insert( P.sendOff.end(), sendOffPer[s].begin(), sendOffPer[s].end() );
      }
      return P;
   }

   int                                                     rank_, nranks_;
   std::vector< MacroCell >                                cells_;
   std::vector< MacroPrimitive >                           faces_, edges_, vertices_;
   std::vector< int >                                      localCells_;
   DoFType                                                 boundaryType_ = DirichletBoundary;
   hyteg_hip_stream_t                                      stream_       = nullptr;
   CommHooks                                               hooks_;
   mutable void *                                          dotResult_ = nullptr, *dotWorkspace_ = nullptr;
   mutable std::map< size_t, std::vector< double* > >      scratchFree_;
   mutable std::vector< void* >                            scratchAll_;
   mutable std::map< std::vector< double* >, double** >    pointerTables_;
   mutable double*                                         nncInv_        = nullptr;
   mutable int                                             batchMaxLevel_ = -2; // -2: read HYTEG_AMD_BATCH_MAX_LEVEL on first use
   // a rank with ONE macro-cell: levels up to this one use the generic batched kernels as well
   static constexpr int                                    kBatchSingleMaxLevelDefault = -1;
   mutable int                                             batchSingleMaxLevel_        = kBatchSingleMaxLevelDefault;
   mutable std::map< std::pair< int, int >, ExchangePlan > plans_;
};

// =====================================================================================================
// P1Function< double > (= vertexdof::VertexDoFFunction< double >)
// =====================================================================================================
template < typename ValueType >
class P1Function
{
   static_assert( std::is_same< ValueType, double >::value,
                  "only double is supported, like the reference's generated 3D kernels (P1ConstantOperator.cpp:417-420)" );

 public:
   using valueType = ValueType;

   uint64_t uid() const { return uid_; }

   // scratch = true: arrays come from (and return to) the storage's scratch pool and are NOT zero-initialised
   P1Function( const std::string& name, const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel,
               bool scratch = false )
   : name_( name )
   , storage_( storage )
   , minLevel_( minLevel )
   , maxLevel_( maxLevel )
   , scratch_( scratch )
   {
      if ( maxLevel > HYTEG_HIP_MAX_LEVEL || minLevel > maxLevel )
         throw std::runtime_error( "P1Function: bad level range" );
      const uint_t nLocal = storage->getNumberOfLocalCells();
      data_.resize( nLocal );
      for ( uint_t c = 0; c < nLocal; ++c )
         for ( uint_t l = minLevel; l <= maxLevel; ++l )
         {
            const size_t doubles = (size_t) layout::cellSize( (int) l );
            if ( scratch )
            {
               data_[c].push_back( storage->acquireScratch( doubles ) );
               continue;
            }
            void* p = nullptr;
            hipCheck( hyteg_hip_malloc( &p, doubles * sizeof( double ) ), "P1Function: malloc" );
            hipCheck( hyteg_hip_memset_zero( p, doubles * sizeof( double ), storage->stream() ), "P1Function: memset" );
            data_[c].push_back( static_cast< double* >( p ) );
         }
   }
   ~P1Function()
   {
      for ( auto& c : data_ )
         for ( uint_t l = 0; l < c.size(); ++l )
         {
            if ( scratch_ )
               storage_->releaseScratch( (size_t) layout::cellSize( (int) ( minLevel_ + l ) ), c[l] );
            else
               hyteg_hip_free( c[l] );
         }
   }
   P1Function( const P1Function& )            = delete;
   P1Function& operator=( const P1Function& ) = delete;

   const std::string&                  getFunctionName() const { return name_; }
   std::shared_ptr< PrimitiveStorage > getStorage() const { return storage_; }
   uint_t                              getMinLevel() const { return minLevel_; }
   uint_t                              getMaxLevel() const { return maxLevel_; }

   // device pointer of the array of local cell `c` at `level` (FunctionMemory::getPointer, FunctionMemory.hpp:109-113)
   double* getCellPointer( uint_t c, uint_t level ) const
   {
      checkLevel( level );
      return data_.at( c )[level - minLevel_];
   }

   // device pointers of local cells [first, first + count) at `level`
   std::vector< double* > cellPointers( uint_t level, int first, int count ) const
   {
      std::vector< double* > p;
      for ( int c = first; c < first + count; ++c )
         p.push_back( getCellPointer( (uint_t) c, level ) );
      return p;
   }

   // ---- interpolate ( VertexDoFFunction.cpp:380-392, :395-470 ) ----
   void interpolate( ValueType constant, uint_t level, DoFType flag = All ) const
   {
      if ( storage_->useBatch( level ) )
      {
         const auto masks = storage_->masksFor( flag );
         storage_->forCellChunks( [&]( int first, int count ) {
            const auto dst = cellPointers( level, first, count );
            hipCheck( hyteg_hip_p1_vector_cells( 3, count, dst.data(), 0, nullptr, &constant, (int) level, masks.data() + first,
                                                 storage_->stream() ),
                      "interpolate (batched)" );
         } );
         return;
      }
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         hipCheck( hyteg_hip_p1_set_cell_masked( getCellPointer( c, level ), constant, (int) level, storage_->maskFor( cell, flag ),
                                                 storage_->stream() ),
                   "interpolate" );
      } );
   }
   void interpolate( const std::function< ValueType( const Point3D& ) >& expr, uint_t level, DoFType flag = All ) const
   {
      // evaluated on the host at the micro-vertex coordinates of VertexDoFMacroCell.hpp:70-77, then uploaded
      const int64_t N = layout::width( (int) level ), size = layout::cellSize( (int) level );
      P1Function    tmp( "interpolate_tmp", storage_, level, level, true );
      std::vector< double > host( (size_t) size );
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const double step = 1.0 / double( N - 1 );
         int64_t      k    = 0;
         for ( int64_t z = 0; z < N; ++z )
            for ( int64_t y = 0; y < N - z; ++y )
               for ( int64_t x = 0; x < N - z - y; ++x )
               {
                  Point3D p;
                  for ( int r = 0; r < 3; ++r )
                  {
                     const double xs = ( cell.coords[1][r] - cell.coords[0][r] ) * step;
                     const double ys = ( cell.coords[2][r] - cell.coords[0][r] ) * step;
                     const double zs = ( cell.coords[3][r] - cell.coords[0][r] ) * step;
                     p[r]            = cell.coords[0][r] + xs * double( x ) + ys * double( y ) + zs * double( z );
                  }
                  host[(size_t) k++] = expr( p );
               }
         hipCheck( hyteg_hip_upload( tmp.getCellPointer( c, level ), host.data(), (size_t) size * sizeof( double ), storage_->stream() ),
                   "interpolate: upload" );
         hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "interpolate: sync" );
      } );
      // the copies of a shared DoF are evaluated from different cells' coordinates: make them bit-identical
      tmp.syncSharedCopies( level );
      assign( { 1.0 }, { tmp }, level, flag );
   }

   // ---- assign / add / multElementwise ( VertexDoFFunction.cpp:1130-1221, :1408-1484, :1487-1563 ) ----
   void assign( const std::vector< ValueType >&                                           scalars,
                const std::vector< std::reference_wrapper< const P1Function< ValueType > > >& functions,
                uint_t                                                                    level,
                DoFType                                                                   flag = All ) const
   {
      vectorOp( 0, scalars, functions, level, flag );
   }
   void add( const std::vector< ValueType >&                                           scalars,
             const std::vector< std::reference_wrapper< const P1Function< ValueType > > >& functions,
             uint_t                                                                    level,
             DoFType                                                                   flag = All ) const
   {
      vectorOp( 1, scalars, functions, level, flag );
   }
   void multElementwise( const std::vector< std::reference_wrapper< const P1Function< ValueType > > >& functions,
                         uint_t                                                                    level,
                         DoFType                                                                   flag = All ) const
   {
      vectorOp( 2, {}, functions, level, flag );
   }
   void setToZero( uint_t level ) const { interpolate( ValueType( 0 ), level, All ); }

   // ---- dot ( VertexDoFFunction.cpp:1710-1793 ) ----
   ValueType dotLocal( const P1Function< ValueType >& rhs, uint_t level, DoFType flag = All ) const
   {
      // one result slot per local cell, a single download (= one host synchronisation) per dot product; the
      // workspace is reused cell after cell, which is safe because all launches are ordered on one stream
      const uint_t nLocal = storage_->getNumberOfLocalCells();
      if ( storage_->useBatch( level ) )
      {
         // one partial + one final launch per chunk of cells, one number per chunk comes back
         const auto masks  = storage_->masksFor( flag, true );
         int        nchunk = 0;
         storage_->forCellChunks( [&]( int first, int count ) {
            const auto a = cellPointers( level, first, count ), b = rhs.cellPointers( level, first, count );
            hipCheck( hyteg_hip_p1_dot_cells( count, a.data(), b.data(), (int) level, masks.data() + first, storage_->dotResult() + nchunk,
                                              storage_->dotWorkspace(), storage_->stream() ),
                      "dotLocal (batched)" );
            ++nchunk;
         } );
         std::vector< double > parts( (size_t) nchunk, 0.0 );
         hipCheck( hyteg_hip_download( parts.data(), storage_->dotResult(), parts.size() * sizeof( double ), storage_->stream() ),
                   "dotLocal: download" );
         double sum = 0.0;
         for ( double v : parts )
            sum += v;
         return sum;
      }
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         hipCheck( hyteg_hip_p1_dot_cell_masked( getCellPointer( c, level ), rhs.getCellPointer( c, level ), (int) level,
                                                 storage_->ownedMaskFor( cell, flag ), storage_->dotResult() + c, storage_->dotWorkspace(),
                                                 storage_->stream() ),
                   "dotLocal" );
      } );
      std::vector< double > parts( nLocal, 0.0 );
      if ( nLocal > 0 )
         hipCheck( hyteg_hip_download( parts.data(), storage_->dotResult(), nLocal * sizeof( double ), storage_->stream() ),
                   "dotLocal: download" );
      double sum = 0.0;
      for ( double v : parts )
         sum += v; // cells in ascending order: deterministic
      return sum;
   }
   ValueType dotGlobal( const P1Function< ValueType >& rhs, uint_t level, DoFType flag = All ) const
   {
      double v = dotLocal( rhs, level, flag );
      if ( storage_->numRanks() > 1 )
      {
         if ( !storage_->hooks().allreduceSum )
            throw std::runtime_error( "dotGlobal: storage is distributed but no allreduce hook is set" );
         storage_->hooks().allreduceSum( storage_->hooks().user, &v, 1 );
      }
      return v;
   }

   // ---- shared-point exchange (the cell-centric replacement of communicate<> / communicateAdditively<>) ----
   // additive: every copy of a shared DoF := sum of all copies (VertexDoFAdditivePackInfo.hpp:676-745 + copy back)
   void sumSharedCopies( uint_t level, DoFType flag = All ) const
   {
      exchangeBegin( level, flag );
      exchangeEnd( level, flag, true );
   }
   // every copy := the copy held by the lowest-numbered neighbour cell
   void syncSharedCopies( uint_t level, DoFType flag = All ) const
   {
      exchangeBegin( level, flag );
      exchangeEnd( level, flag, false );
   }
   // split form: pack + start the transfer / wait + reduce.  Kernels that do not touch shared points may be
   // launched in between (the interior apply overlaps the halo exchange).
   void beginSumSharedCopies( uint_t level, DoFType flag = All ) const { exchangeBegin( level, flag ); }
   void endSumSharedCopies( uint_t level, DoFType flag = All ) const { exchangeEnd( level, flag, true ); }

   void copyCellToHost( uint_t c, uint_t level, double* host ) const
   {
      hipCheck( hyteg_hip_download( host, getCellPointer( c, level ), (size_t) layout::cellSize( (int) level ) * sizeof( double ),
                                    storage_->stream() ),
                "copyCellToHost" );
   }
   void copyCellFromHost( uint_t c, uint_t level, const double* host ) const
   {
      hipCheck( hyteg_hip_upload( getCellPointer( c, level ), host, (size_t) layout::cellSize( (int) level ) * sizeof( double ),
                                  storage_->stream() ),
                "copyCellFromHost" );
      hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "copyCellFromHost: sync" );
   }

 private:
   void checkLevel( uint_t level ) const
   {
      if ( level < minLevel_ || level > maxLevel_ )
         throw std::runtime_error( "P1Function '" + name_ + "': level " + std::to_string( level ) + " not allocated" );
   }
   template < typename F >
   void forCells( F&& fn ) const
   {
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         fn( c, storage_->getLocalCell( c ) );
   }
   void vectorOp( int                                                                       op,
                  const std::vector< ValueType >&                                           scalars,
                  const std::vector< std::reference_wrapper< const P1Function< ValueType > > >& functions,
                  uint_t                                                                    level,
                  DoFType                                                                   flag ) const
   {
      if ( functions.empty() || functions.size() > HYTEG_HIP_MAX_SRCS || ( op != 2 && scalars.size() != functions.size() ) )
         throw std::runtime_error( "P1Function::assign/add/multElementwise: bad number of functions or scalars" );
      if ( storage_->useBatch( level ) )
      {
         const auto masks = storage_->masksFor( flag );
         storage_->forCellChunks( [&]( int first, int count ) {
            const auto             dst = cellPointers( level, first, count );
            std::vector< double* > srcs; // [function][cell]
            for ( const auto& f : functions )
               for ( double* q : f.get().cellPointers( level, first, count ) )
                  srcs.push_back( q );
            hipCheck( hyteg_hip_p1_vector_cells( op, count, dst.data(), (int) functions.size(), srcs.data(),
                                                 op == 2 ? nullptr : scalars.data(), (int) level, masks.data() + first, storage_->stream() ),
                      "P1Function vector op (batched)" );
         } );
         return;
      }
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const double* srcs[HYTEG_HIP_MAX_SRCS];
         for ( uint_t k = 0; k < functions.size(); ++k )
            srcs[k] = functions[k].get().getCellPointer( c, level );
         hipCheck( hyteg_hip_p1_vector_cell_masked( op, getCellPointer( c, level ), (int) functions.size(), srcs,
                                                    op == 2 ? nullptr : scalars.data(), (int) level, storage_->maskFor( cell, flag ),
                                                    storage_->stream() ),
                   "P1Function vector op" );
      } );
   }

 public:
   // assign (op 0) / add (op 1) with coefficients read from device memory when the kernels run; storages of one rank
   // with at most HYTEG_HIP_MAX_BATCH local cells (one launch)
   void vectorOpDeviceScalars( int op, const std::vector< const double* >& scalarPtrs,
                               const std::vector< std::reference_wrapper< const P1Function< ValueType > > >& functions, uint_t level,
                               DoFType flag ) const
   {
      const int  count = (int) storage_->getNumberOfLocalCells();
      const auto masks = storage_->masksFor( flag );
      const auto dst   = cellPointers( level, 0, count );
      std::vector< double* > srcs;
      for ( const auto& f : functions )
         for ( double* q : f.get().cellPointers( level, 0, count ) )
            srcs.push_back( q );
      hipCheck( hyteg_hip_p1_vector_cells_dev( op, count, dst.data(), (int) functions.size(), srcs.data(), scalarPtrs.data(), (int) level,
                                               masks.data(), storage_->stream() ),
                "P1Function vector op (device scalars)" );
   }
   // cgScalars[slot] = <this, rhs> over the points `flag` selects (each shared point counted once), then phase `phase` of
   // the conjugate gradient recurrences (hyteg_hip_cg_scalars), in one launch; no host synchronisation
   void dotLocalToCgScalars( const P1Function< ValueType >& rhs, uint_t level, DoFType flag, double* cgScalars, int slot, int phase,
                             double relTol, double absTol ) const
   {
      const int  count = (int) storage_->getNumberOfLocalCells();
      const auto masks = storage_->masksFor( flag, true );
      const auto a = cellPointers( level, 0, count ), b = rhs.cellPointers( level, 0, count );
      hipCheck( hyteg_hip_p1_dot_cells_cg( count, a.data(), b.data(), (int) level, masks.data(), cgScalars, slot, phase, relTol, absTol,
                                           storage_->dotWorkspace(), storage_->stream() ),
                "dotLocalToCgScalars" );
   }

 private:
   // device table [ local cell arrays at `level` ..., receive segment of peer 0, peer 1, ... ]
   double** basesFor( uint_t level, int cls ) const
   {
      const auto& plan = storage_->devicePlan( (int) level, cls );
      std::vector< double* > host;
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         host.push_back( getCellPointer( c, level ) );
      double* seg = plan.recvBuffer;
      for ( uint_t s = 0; s < plan.peers.size(); ++s )
      {
         host.push_back( seg );
         seg += plan.recvCount[s];
      }
      return storage_->pointerTable( host );
   }

   // The hooks are called by EVERY rank for every boundary class the flag selects, also by a rank that shares nothing
   // with anybody in that class: the transport behind them is a collective (all_to_all), and a rank that skipped the
   // call would dead-lock the others.  The hook itself decides (globally) whether there is anything to exchange.
   void exchangeBegin( uint_t level, DoFType flag ) const
   {
      checkLevel( level );
      if ( storage_->numRanks() == 1 )
         return;
      if ( !storage_->hooks().exchangeBegin || !storage_->hooks().exchangeEnd )
         throw std::runtime_error( "exchange: storage is distributed but no exchange hooks are set" );
      for ( int cls = 0; cls < 2; ++cls )
      {
         if ( !testFlag( storage_->boundaryTypeOf( cls == 1 ), flag ) )
            continue;
         if ( !storage_->exchangePlan( (int) level, cls ).peers.empty() )
         {
            const auto& plan  = storage_->devicePlan( (int) level, cls );
            double**    bases = basesFor( level, cls );
            hipCheck( hyteg_hip_gather_entries( plan.sendBuffer, bases, plan.dSendBuf, plan.dSendOff, plan.totalSend(), storage_->stream() ),
                      "exchange: pack" );
         }
         storage_->hooks().exchangeBegin( storage_->hooks().user, (int) level, cls );
      }
   }
   void exchangeEnd( uint_t level, DoFType flag, bool additive ) const
   {
      for ( int cls = 0; cls < 2; ++cls )
      {
         if ( !testFlag( storage_->boundaryTypeOf( cls == 1 ), flag ) )
            continue;
         if ( storage_->numRanks() > 1 )
            storage_->hooks().exchangeEnd( storage_->hooks().user, (int) level, cls );
         if ( storage_->exchangePlan( (int) level, cls ).ngroups() == 0 )
            continue;
         const auto& plan  = storage_->devicePlan( (int) level, cls );
         double**    bases = basesFor( level, cls );
         hipCheck( additive ? hyteg_hip_sum_shared( bases, plan.dGroupPtr, plan.dEntryBuf, plan.dEntryOff, plan.ngroups(),
                                                    (int) storage_->getNumberOfLocalCells(), storage_->stream() )
                            : hyteg_hip_copy_shared( bases, plan.dGroupPtr, plan.dEntryBuf, plan.dEntryOff, plan.ngroups(),
                                                     (int) storage_->getNumberOfLocalCells(), storage_->stream() ),
                   "exchange: reduce" );
      }
   }

   std::string                                                           name_;
   std::shared_ptr< PrimitiveStorage >                                   storage_;
   uint_t                                                                minLevel_, maxLevel_;
   bool                                                                  scratch_ = false;
   uint64_t                                                              uid_     = nextUid();
   std::vector< std::vector< double* > >                                 data_;
};

// =====================================================================================================
// P2Function< double >  ( src/hyteg/p2functionspace/P2Function.hpp ): a VertexDoF function plus an EdgeDoF function.
// First version: storages with ONE macro-cell (the shared edge DoFs of several cells need their own
// exchange plans, which do not exist yet).  Edge-DoF arrays: layout of edgedofspace/EdgeDoFIndexing.hpp:920-985.
// =====================================================================================================
template < typename ValueType >
class P2Function
{
 public:
   using valueType = ValueType;
   P2Function( const std::string& name, const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : name_( name )
   , storage_( storage )
   , minLevel_( minLevel )
   , maxLevel_( maxLevel )
   , vertexDoFFunction_( name + "_VertexDoF", storage, minLevel, maxLevel )
   {
      if ( storage->numRanks() != 1 )
         throw std::runtime_error( "P2Function: storages distributed over several ranks are not supported in this version" );
      if ( maxLevel > HYTEG_HIP_P2_MAX_LEVEL )
         throw std::runtime_error( "P2Function: level out of range" );
      edge_.resize( storage->getNumberOfLocalCells() );
      for ( auto& perCell : edge_ )
         for ( uint_t l = minLevel; l <= maxLevel; ++l )
         {
            const size_t bytes = std::max< size_t >( 1, hyteg_hip_p2_edge_array_size( (int) l ) ) * sizeof( double );
            void*        q     = nullptr;
            hipCheck( hyteg_hip_malloc( &q, bytes ), "P2Function: malloc" );
            hipCheck( hyteg_hip_memset_zero( q, bytes, storage->stream() ), "P2Function: memset" );
            perCell.push_back( static_cast< double* >( q ) );
         }
   }
   ~P2Function()
   {
      for ( auto& perCell : edge_ )
         for ( double* q : perCell )
            hyteg_hip_free( q );
      for ( auto& kv : edgeBases_ )
         hyteg_hip_free( kv.second );
   }
   P2Function( const P2Function& )            = delete;
   P2Function& operator=( const P2Function& ) = delete;

   const P1Function< ValueType >&      getVertexDoFFunction() const { return vertexDoFFunction_; }
   std::shared_ptr< PrimitiveStorage > getStorage() const { return storage_; }
   // device pointer of the edge-DoF array of local cell c
   double* getEdgeCellPointer( uint_t c, uint_t level ) const
   {
      if ( c >= edge_.size() || level < minLevel_ || level > maxLevel_ )
         throw std::runtime_error( "P2Function '" + name_ + "': bad cell or level" );
      return edge_[c][level - minLevel_];
   }
   uint_t getNumberOfEdgeDoFs( uint_t level ) const { return hyteg_hip_p2_edge_array_size( (int) level ); }

   void interpolate( ValueType constant, uint_t level, DoFType flag = All ) const
   {
      vertexDoFFunction_.interpolate( constant, level, flag );
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         hipCheck( hyteg_hip_p2_edge_vector_cell_masked( 3, getEdgeCellPointer( c, level ), 0, nullptr, &constant, (int) level,
                                                         storage_->maskFor( cell, flag ), storage_->stream() ),
                   "P2Function::interpolate" );
      } );
   }
   // expression evaluated at the micro-vertices and at the edge midpoints (EdgeDoFFunction::interpolate)
   void interpolate( const std::function< ValueType( const Point3D& ) >& expr, uint_t level, DoFType flag = All ) const
   {
      vertexDoFFunction_.interpolate( expr, level, flag );
      static const int ends[7][2][3] = { { { 0, 0, 0 }, { 1, 0, 0 } }, { { 0, 0, 0 }, { 0, 1, 0 } }, { { 0, 0, 0 }, { 0, 0, 1 } },
                                         { { 1, 0, 0 }, { 0, 1, 0 } }, { { 1, 0, 0 }, { 0, 0, 1 } }, { { 0, 1, 0 }, { 0, 0, 1 } },
                                         { { 0, 1, 0 }, { 1, 0, 1 } } };
      const int64_t n    = int64_t( 1 ) << level;
      const double  step = 1.0 / double( n );
      const size_t  ne   = std::max< size_t >( 1, getNumberOfEdgeDoFs( level ) );
      std::vector< double* > tmp;
      forCells( [&]( uint_t, const MacroCell& cell ) {
         std::vector< double > host;
         host.reserve( ne );
         for ( int o = 0; o < 7; ++o )
         {
            const int64_t W = o == 6 ? n - 1 : n;
            for ( int64_t z = 0; z < W; ++z )
               for ( int64_t y = 0; y < W - z; ++y )
                  for ( int64_t x = 0; x < W - z - y; ++x )
                  {
                     const double mx = double( x ) + 0.5 * ( ends[o][0][0] + ends[o][1][0] ), my = double( y ) + 0.5 * ( ends[o][0][1] + ends[o][1][1] ),
                                  mz = double( z ) + 0.5 * ( ends[o][0][2] + ends[o][1][2] );
                     Point3D      q;
                     for ( int r = 0; r < 3; ++r )
                        q[r] = cell.coords[0][r] + ( cell.coords[1][r] - cell.coords[0][r] ) * step * mx +
                               ( cell.coords[2][r] - cell.coords[0][r] ) * step * my + ( cell.coords[3][r] - cell.coords[0][r] ) * step * mz;
                     host.push_back( expr( q ) );
                  }
         }
         double* t = storage_->acquireScratch( ne );
         hipCheck( hyteg_hip_upload( t, host.data(), host.size() * sizeof( double ), storage_->stream() ), "P2Function::interpolate: upload" );
         hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "P2Function::interpolate: sync" );
         tmp.push_back( t );
      } );
      // the copies of a shared edge DoF were evaluated from different cells' coordinates: make them bit-identical
      exchangeEdges( tmp, level, All, false );
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const double* srcs[1] = { tmp[c] };
         const double  one[1]  = { 1.0 };
         hipCheck( hyteg_hip_p2_edge_vector_cell_masked( 0, getEdgeCellPointer( c, level ), 1, srcs, one, (int) level,
                                                         storage_->maskFor( cell, flag ), storage_->stream() ),
                   "P2Function::interpolate: assign" );
      } );
      hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "P2Function::interpolate: sync" );
      for ( double* t : tmp )
         storage_->releaseScratch( ne, t );
   }
   void setToZero( uint_t level ) const { interpolate( ValueType( 0 ), level, All ); }

   void assign( const std::vector< ValueType >&                                           scalars,
                const std::vector< std::reference_wrapper< const P2Function< ValueType > > >& functions,
                uint_t                                                                    level,
                DoFType                                                                   flag = All ) const
   {
      vectorOp( 0, scalars, functions, level, flag );
   }
   void add( const std::vector< ValueType >&                                           scalars,
             const std::vector< std::reference_wrapper< const P2Function< ValueType > > >& functions,
             uint_t                                                                    level,
             DoFType                                                                   flag = All ) const
   {
      vectorOp( 1, scalars, functions, level, flag );
   }
   // a shared DoF is counted by its lowest-numbered neighbour cell only
   ValueType dotLocal( const P2Function< ValueType >& rhs, uint_t level, DoFType flag = All ) const
   {
      double       sum = vertexDoFFunction_.dotLocal( rhs.vertexDoFFunction_, level, flag );
      const uint_t nl  = storage_->getNumberOfLocalCells();
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         hipCheck( hyteg_hip_p2_edge_dot_cell_masked( getEdgeCellPointer( c, level ), rhs.getEdgeCellPointer( c, level ), (int) level,
                                                      storage_->ownedMaskFor( cell, flag ), storage_->dotResult() + c, storage_->dotWorkspace(),
                                                      storage_->stream() ),
                   "P2Function::dotLocal" );
      } );
      std::vector< double > parts( nl, 0.0 );
      hipCheck( hyteg_hip_download( parts.data(), storage_->dotResult(), nl * sizeof( double ), storage_->stream() ), "P2Function::dotLocal: download" );
      for ( double v : parts )
         sum += v;
      return sum;
   }
   ValueType dotGlobal( const P2Function< ValueType >& rhs, uint_t level, DoFType flag = All ) const { return dotLocal( rhs, level, flag ); }

   // every copy of a shared edge DoF := sum of all copies (communicateAdditively< Cell, Face / Edge > of the EdgeDoFFunction)
   void sumSharedEdgeCopies( uint_t level, DoFType flag = All ) const
   {
      std::vector< double* > arrays;
      forCells( [&]( uint_t c, const MacroCell& ) { arrays.push_back( getEdgeCellPointer( c, level ) ); } );
      exchangeEdges( arrays, level, flag, true );
   }

   void copyEdgeToHost( uint_t c, uint_t level, double* host ) const
   {
      hipCheck( hyteg_hip_download( host, getEdgeCellPointer( c, level ), getNumberOfEdgeDoFs( level ) * sizeof( double ), storage_->stream() ),
                "P2Function::copyEdgeToHost" );
   }
   void copyEdgeFromHost( uint_t c, uint_t level, const double* host ) const
   {
      hipCheck( hyteg_hip_upload( getEdgeCellPointer( c, level ), host, getNumberOfEdgeDoFs( level ) * sizeof( double ), storage_->stream() ),
                "P2Function::copyEdgeFromHost" );
      hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "P2Function::copyEdgeFromHost: sync" );
   }

 private:
   template < typename F >
   void forCells( F&& fn ) const
   {
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         fn( c, storage_->getLocalCell( c ) );
   }
   // additive (or copy) exchange of the shared edge DoFs held in `arrays` (one edge-DoF array per local cell)
   void exchangeEdges( const std::vector< double* >& arrays, uint_t level, DoFType flag, bool additive ) const
   {
      for ( int cls = 0; cls < 2; ++cls )
      {
         if ( !testFlag( storage_->boundaryTypeOf( cls == 1 ), flag ) || storage_->exchangePlan( (int) level, cls, 1 ).ngroups() == 0 )
            continue;
         const auto& plan = storage_->devicePlan( (int) level, cls, 1 );
         // device table of the array pointers; cached per (first pointer) because temporaries come and go
         auto   key = std::make_pair( arrays[0], cls );
         auto   it  = edgeBases_.find( key );
         if ( it == edgeBases_.end() )
         {
            void* d = nullptr;
            hipCheck( hyteg_hip_malloc( &d, arrays.size() * sizeof( double* ) ), "edge bases: malloc" );
            it = edgeBases_.emplace( key, static_cast< double** >( d ) ).first;
         }
         hipCheck( hyteg_hip_upload( it->second, arrays.data(), arrays.size() * sizeof( double* ), storage_->stream() ), "edge bases: upload" );
         hipCheck( hyteg_hip_stream_synchronize( storage_->stream() ), "edge bases: sync" );
         hipCheck( additive ? hyteg_hip_sum_shared( it->second, plan.dGroupPtr, plan.dEntryBuf, plan.dEntryOff, plan.ngroups(),
                                                    (int) arrays.size(), storage_->stream() )
                            : hyteg_hip_copy_shared( it->second, plan.dGroupPtr, plan.dEntryBuf, plan.dEntryOff, plan.ngroups(),
                                                     (int) arrays.size(), storage_->stream() ),
                   "edge exchange" );
      }
   }
   void vectorOp( int                                                                       op,
                  const std::vector< ValueType >&                                           scalars,
                  const std::vector< std::reference_wrapper< const P2Function< ValueType > > >& functions,
                  uint_t                                                                    level,
                  DoFType                                                                   flag ) const
   {
      if ( functions.empty() || functions.size() > HYTEG_HIP_MAX_SRCS || scalars.size() != functions.size() )
         throw std::runtime_error( "P2Function::assign/add: bad number of functions or scalars" );
      std::vector< std::reference_wrapper< const P1Function< ValueType > > > vs;
      for ( uint_t k = 0; k < functions.size(); ++k )
         vs.push_back( functions[k].get().vertexDoFFunction_ );
      if ( op == 0 )
         vertexDoFFunction_.assign( scalars, vs, level, flag );
      else
         vertexDoFFunction_.add( scalars, vs, level, flag );
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const double* es[HYTEG_HIP_MAX_SRCS];
         for ( uint_t k = 0; k < functions.size(); ++k )
            es[k] = functions[k].get().getEdgeCellPointer( c, level );
         hipCheck( hyteg_hip_p2_edge_vector_cell_masked( op, getEdgeCellPointer( c, level ), (int) functions.size(), es, scalars.data(), (int) level,
                                                         storage_->maskFor( cell, flag ), storage_->stream() ),
                   "P2Function vector op" );
      } );
   }

   std::string                                            name_;
   std::shared_ptr< PrimitiveStorage >                    storage_;
   uint_t                                                 minLevel_, maxLevel_;
   P1Function< ValueType >                                vertexDoFFunction_;
   std::vector< std::vector< double* > >                  edge_; // [local cell][level - minLevel]
   mutable std::map< std::pair< double*, int >, double** > edgeBases_;
};

// =====================================================================================================
// Forms: first row of the P1 element matrix of a tetrahedron (kernel INPUT, setup only).
// P1FenicsForm< ..., p1_tet_diffusion_cell_integral_0_otherwise >  src/hyteg/forms/form_fenics_base/P1FenicsForm.hpp:96-124
// -> src/hyteg/forms/form_fenics_generated/p1_tet_diffusion.h:4113-4240: K_0j = |det J|/6 grad(lambda_0).grad(lambda_j);
// p1_tet_mass.h: M_0j = |det J|/120 (1 + delta_0j).
// =====================================================================================================
namespace forms {
inline double det3( const double J[3][3] )
{
   return J[0][0] * ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) - J[0][1] * ( J[1][0] * J[2][2] - J[1][2] * J[2][0] ) +
          J[0][2] * ( J[1][0] * J[2][1] - J[1][1] * J[2][0] );
}
struct P1LaplaceForm
{
   static void integrateRow0( const std::array< Point3D, 4 >& c, double row[4] )
   {
      double J[3][3];
      for ( int r = 0; r < 3; ++r )
         for ( int k = 0; k < 3; ++k )
            J[r][k] = c[k + 1][r] - c[0][r];
      const double det = det3( J );
      double       Ji[3][3];
      Ji[0][0] = ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) / det;
      Ji[0][1] = ( J[0][2] * J[2][1] - J[0][1] * J[2][2] ) / det;
      Ji[0][2] = ( J[0][1] * J[1][2] - J[0][2] * J[1][1] ) / det;
      Ji[1][0] = ( J[1][2] * J[2][0] - J[1][0] * J[2][2] ) / det;
      Ji[1][1] = ( J[0][0] * J[2][2] - J[0][2] * J[2][0] ) / det;
      Ji[1][2] = ( J[0][2] * J[1][0] - J[0][0] * J[1][2] ) / det;
      Ji[2][0] = ( J[1][0] * J[2][1] - J[1][1] * J[2][0] ) / det;
      Ji[2][1] = ( J[0][1] * J[2][0] - J[0][0] * J[2][1] ) / det;
      Ji[2][2] = ( J[0][0] * J[1][1] - J[0][1] * J[1][0] ) / det;
      double g[4][3];
      for ( int r = 0; r < 3; ++r )
      {
         g[1][r] = Ji[0][r];
         g[2][r] = Ji[1][r];
         g[3][r] = Ji[2][r];
         g[0][r] = -( Ji[0][r] + Ji[1][r] + Ji[2][r] );
      }
      const double vol6 = std::fabs( det ) / 6.0;
      for ( int j = 0; j < 4; ++j )
         row[j] = vol6 * ( g[0][0] * g[j][0] + g[0][1] * g[j][1] + g[0][2] * g[j][2] );
   }
};
struct P1MassForm
{
   static void integrateRow0( const std::array< Point3D, 4 >& c, double row[4] )
   {
      double J[3][3];
      for ( int r = 0; r < 3; ++r )
         for ( int k = 0; k < 3; ++k )
            J[r][k] = c[k + 1][r] - c[0][r];
      const double d = std::fabs( det3( J ) ) / 120.0;
      row[0]         = 2.0 * d;
      row[1] = row[2] = row[3] = d;
   }
};
} // namespace forms

// stencil slots in the C-ABI order (include/hyteg_hip.h) and the 24 micro-tetrahedra around an inner micro-vertex
// (src/hyteg/p1functionspace/P1Elements.hpp:93-143; slot numbers instead of stencilDirection names)
namespace stencil {
static const int kOffsets[15][3] = { { 0, 0, -1 }, { 1, 0, -1 }, { -1, 1, -1 }, { 0, 1, -1 }, { 0, -1, 0 },
                                     { 1, -1, 0 }, { -1, 0, 0 }, { 0, 0, 0 },   { 1, 0, 0 },  { -1, 1, 0 },
                                     { 0, 1, 0 },  { 0, -1, 1 }, { 1, -1, 1 },  { -1, 0, 1 }, { 0, 0, 1 } };
enum
{
   BC = 0, BE, BNW, BN, S, SE, W, C, E, NW, N, TS, TSE, TW, TC
};
static const int kMicroTets[24][4] = {
    { C, BC, BE, BN }, { C, S, SE, TS },   { C, W, NW, TW },   { C, N, E, TC },    { C, W, BC, S },    { C, E, SE, BE },
    { C, N, NW, BN },  { C, TS, TC, TW },  { C, BC, BN, BNW }, { C, W, S, TS },    { C, E, SE, TSE },  { C, NW, N, TC },
    { C, BC, S, SE },  { C, W, NW, BNW },  { C, E, BN, N },    { C, TC, TS, TSE }, { C, W, BC, BNW },  { C, E, BE, BN },
    { C, TC, TW, NW }, { C, SE, TS, TSE }, { C, BC, BE, SE },  { C, BN, BNW, NW }, { C, E, TSE, TC },  { C, W, TS, TW } };
// which cell faces a slot's points lie on: edges 0-5, faces 0-3, vertices 0-3
static const int kSlotFaces[14][4] = { { 1, 1, 0, 0 }, { 1, 0, 1, 0 }, { 1, 0, 0, 1 }, { 0, 1, 1, 0 }, { 0, 1, 0, 1 },
                                       { 0, 0, 1, 1 }, { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 },
                                       { 1, 1, 1, 0 }, { 1, 1, 0, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 } };
inline bool directionStaysInCell( int slot, const int* d )
{
   const int* f = kSlotFaces[slot];
   return !( ( f[0] && d[2] < 0 ) || ( f[1] && d[1] < 0 ) || ( f[2] && d[0] < 0 ) || ( f[3] && d[0] + d[1] + d[2] > 0 ) );
}

struct CellStencils
{
   double inner[15];     // stencil at an inner micro-vertex (P1ConstantOperator.cpp:680-693 assembles it at (1,1,1))
   double slots[14][15]; // this cell's share of the stencil at a micro-vertex on edge 0-5 / face 0-3 / vertex 0-3
};

// P1Elements3D::calculateStencilInMacroCell( index, cell, level, form ), P1Elements.hpp:303-380, for all 15 point classes.
// Affine cells: the 24 element matrices do not depend on the micro-vertex, so they are computed once.
template < class Form >
CellStencils assemble( const MacroCell& cell, uint_t level )
{
   const double step = 1.0 / double( int64_t( 1 ) << level );
   Point3D      xs, ys, zs;
   for ( int r = 0; r < 3; ++r )
   {
      xs[r] = ( cell.coords[1][r] - cell.coords[0][r] ) * step;
      ys[r] = ( cell.coords[2][r] - cell.coords[0][r] ) * step;
      zs[r] = ( cell.coords[3][r] - cell.coords[0][r] ) * step;
   }
   double rows[24][4];
   for ( int t = 0; t < 24; ++t )
   {
      std::array< Point3D, 4 > c;
      for ( int v = 0; v < 4; ++v )
      {
         const int* o = kOffsets[kMicroTets[t][v]];
         for ( int r = 0; r < 3; ++r )
            c[v][r] = cell.coords[0][r] + xs[r] * double( 1 + o[0] ) + ys[r] * double( 1 + o[1] ) + zs[r] * double( 1 + o[2] );
      }
      Form::integrateRow0( c, rows[t] );
   }
   CellStencils S{};
   for ( int t = 0; t < 24; ++t )
      for ( int v = 0; v < 4; ++v )
         S.inner[kMicroTets[t][v]] += rows[t][v];
   for ( int s = 0; s < 14; ++s )
      for ( int t = 0; t < 24; ++t )
      {
         bool inside = true;
         for ( int v = 1; v < 4; ++v )
            inside = inside && directionStaysInCell( s, kOffsets[kMicroTets[t][v]] );
         if ( inside )
            for ( int v = 0; v < 4; ++v )
               S.slots[s][kMicroTets[t][v]] += rows[t][v];
      }
   return S;
}
// Tables of the SOR / Gauss-Seidel sweep over the macro-vertices, -edges and -faces around a cell
// (hyteg_hip_p1_sor_shell_cell): total weights over all neighbour cells, sweep orientations = the macro-primitives'
// own orientations (vertex ids ascending, MeshInfo.cpp:37-72), and the cell's partial stencils without the weights
// that the sweep handles itself (`rest`).
struct CellSorTables
{
   double rest[14][15];
   int    edgeVerts[6][2];
   double edgeW[6][3];
   int    faceVerts[4][3];
   double faceW[4][7];
   double vertexW[4];
};
inline int offsetIndex( int dx, int dy, int dz )
{
   for ( int k = 0; k < 15; ++k )
      if ( kOffsets[k][0] == dx && kOffsets[k][1] == dy && kOffsets[k][2] == dz )
         return k;
   throw std::runtime_error( "offsetIndex: not a stencil direction" );
}
static const int kFaceDirs[6][2] = { { -1, 0 }, { 1, 0 }, { 0, -1 }, { 0, 1 }, { 1, -1 }, { -1, 1 } };
static const int kUnit[4][3]     = { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
} // namespace stencil

// =====================================================================================================
// P1ConstantOperator< Form >  ( src/constant_stencil_operator/P1ConstantOperator.hpp:33-168 )
// =====================================================================================================
template < class Form >
class P1ConstantOperator
{
 public:
   using srcType = P1Function< double >;
   using dstType = P1Function< double >;

   P1ConstantOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : storage_( storage )
   , minLevel_( minLevel )
   , maxLevel_( maxLevel )
   {
      // assembleStencils(), P1ConstantOperator.cpp:680-732: per level and cell
      for ( uint_t l = minLevel; l <= maxLevel; ++l )
      {
         std::vector< stencil::CellStencils > perCell;
         for ( const auto& cell : storage->getCells() ) // all cells: inverse diagonals need the neighbours' shares
            perCell.push_back( stencil::assemble< Form >( cell, l ) );
         stencils_[l] = perCell;
         sorTables_[l] = buildSorTables( perCell );
         if ( l >= HYTEG_HIP_MIN_LEVEL )
            hipCheck( hyteg_hip_prepare_level( (int) l ), "P1ConstantOperator: prepare_level" );
      }
   }

   std::shared_ptr< PrimitiveStorage > getStorage() const { return storage_; }
   uint64_t                            uid() const { return uid_; }
   uint_t                              getMinLevel() const { return minLevel_; }
   uint_t                              getMaxLevel() const { return maxLevel_; }
   const stencil::CellStencils&        getCellStencils( int globalCellID, uint_t level ) const { return stencils_.at( level ).at( globalCellID ); }

   // Operator::apply, P1Operator.hpp:192-320
   void apply( const P1Function< double >& src, const P1Function< double >& dst, uint_t level, DoFType flag, UpdateType updateType = Replace ) const
   {
      if ( &src == &dst )
         throw std::runtime_error( "P1ConstantOperator::apply: src and dst must differ (P1Operator.hpp:198)" );
      if ( storage_->useBatch( level ) )
      {
         applyBatched( src, dst, level, flag, updateType );
         return;
      }
      const P1Function< double >* shellDst = &dst;
      std::unique_ptr< P1Function< double > > tmp;
      if ( updateType == Add && hasSharedPoints( level, flag ) )
      {
         // partial results of shared DoFs are summed over cells before they are added to dst
         tmp.reset( new P1Function< double >( "apply_tmp", storage_, level, level, true ) );
         tmp->interpolate( 0.0, level, All );
         shellDst = tmp.get();
      }
      // 1. this cell's share of the shared macro-face/edge/vertex DoFs (tiny kernels), 2. start the halo exchange,
      // 3. the interior stencil while the exchange is in flight, 4. reduce the shares
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const auto& S = getCellStencils( cell.id, level );
         hipCheck( hyteg_hip_p1_apply_cell_boundary( shellDst->getCellPointer( c, level ), src.getCellPointer( c, level ), (int) level,
                                                     &S.slots[0][0], storage_->maskFor( cell, flag ),
                                                     ( updateType == Add && shellDst == &dst ) ? HYTEG_HIP_ADD : HYTEG_HIP_REPLACE,
                                                     storage_->stream() ),
                   "apply: boundary" );
      } );
      shellDst->beginSumSharedCopies( level, flag );
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const unsigned mask = storage_->maskFor( cell, flag );
         if ( ( mask & HYTEG_HIP_MASK_INNER ) && level >= HYTEG_HIP_MIN_LEVEL )
            hipCheck( hyteg_hip_p1_apply_cell( dst.getCellPointer( c, level ), src.getCellPointer( c, level ), (int) level,
                                               getCellStencils( cell.id, level ).inner,
                                               updateType == Replace ? HYTEG_HIP_REPLACE : HYTEG_HIP_ADD, storage_->stream() ),
                      "apply: cell" );
      } );
      shellDst->endSumSharedCopies( level, flag );
      if ( shellDst != &dst )
      {
         // dst += tmp on the shell points selected by flag
         forCells( [&]( uint_t c, const MacroCell& cell ) {
            const double* srcs[1] = { shellDst->getCellPointer( c, level ) };
            const double  one[1]  = { 1.0 };
            hipCheck( hyteg_hip_p1_vector_cell_masked( 1, dst.getCellPointer( c, level ), 1, srcs, one, (int) level,
                                                       storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_SHELL, storage_->stream() ),
                      "apply: add shell" );
         } );
      }
   }

   // P1Operator::smooth_jac, P1Operator.hpp:429-447
   void smooth_jac( const P1Function< double >& dst, const P1Function< double >& rhs, const P1Function< double >& src, double relax,
                    uint_t level, DoFType flag ) const
   {
      if ( &src == &dst )
         throw std::runtime_error( "smooth_jac: src and dst must differ" );
      const auto& invDiag = *getInverseDiagonalValues();
      if ( storage_->useBatch( level ) )
      {
         // phase 0: inner points complete, shell points this cell's share; exchange; phase 1: shell update
         const auto masks = storage_->masksFor( flag );
         for ( int phase = 0; phase < 2; ++phase )
         {
            storage_->forCellChunks( [&]( int first, int count ) {
               const auto d = dst.cellPointers( level, first, count ), r = rhs.cellPointers( level, first, count ),
                          u = src.cellPointers( level, first, count ), iv = invDiag.cellPointers( level, first, count );
               hipCheck( hyteg_hip_p1_jacobi_cells( count, d.data(), r.data(), u.data(), iv.data(), (int) level,
                                                    stencilTable( level ) + (size_t) first * 225, relax, masks.data() + first, phase,
                                                    storage_->stream() ),
                         "smooth_jac (batched)" );
            } );
            if ( phase == 0 )
               dst.sumSharedCopies( level, flag );
         }
         return;
      }
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const auto&    S    = getCellStencils( cell.id, level );
         const unsigned mask = storage_->maskFor( cell, flag );
         if ( ( mask & HYTEG_HIP_MASK_INNER ) && level >= HYTEG_HIP_MIN_LEVEL )
            hipCheck( hyteg_hip_p1_jacobi_cell( dst.getCellPointer( c, level ), rhs.getCellPointer( c, level ), src.getCellPointer( c, level ),
                                                nullptr, (int) level, S.inner, relax, storage_->stream() ),
                      "smooth_jac: cell" );
         hipCheck( hyteg_hip_p1_apply_cell_boundary( dst.getCellPointer( c, level ), src.getCellPointer( c, level ), (int) level,
                                                     &S.slots[0][0], mask, HYTEG_HIP_REPLACE, storage_->stream() ),
                   "smooth_jac: boundary" );
      } );
      dst.sumSharedCopies( level, flag );
      // on the shell: dst = rhs - dst ; dst = invDiag .* dst ; dst = src + relax * dst  (the reference's three passes)
      forCells( [&]( uint_t c, const MacroCell& cell ) {
         const unsigned shell = storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_SHELL;
         if ( !shell )
            return;
         double*       d = dst.getCellPointer( c, level );
         const double* a[2] = { rhs.getCellPointer( c, level ), d };
         const double  s1[2] = { 1.0, -1.0 };
         hipCheck( hyteg_hip_p1_vector_cell_masked( 0, d, 2, a, s1, (int) level, shell, storage_->stream() ), "smooth_jac: residual" );
         const double* m[2] = { invDiag.getCellPointer( c, level ), d };
         hipCheck( hyteg_hip_p1_vector_cell_masked( 2, d, 2, m, nullptr, (int) level, shell, storage_->stream() ), "smooth_jac: scale" );
         const double* u[2] = { src.getCellPointer( c, level ), d };
         const double  s2[2] = { 1.0, relax };
         hipCheck( hyteg_hip_p1_vector_cell_masked( 0, d, 2, u, s2, (int) level, shell, storage_->stream() ), "smooth_jac: update" );
      } );
   }

   // P1Operator::smooth_sor / smooth_gs, P1Operator.hpp:322-418: macro-vertices, -edges, -faces, -cells (reversed for
   // backwards), each class with the values the reference's communication schedule gives it.  Cell-centric form:
   //  rest  = (stencil sum over the neighbours outside the primitive's closure), summed over cells by ONE exchange,
   //          taken from the pre-sweep state (forward) -- the reference's ghost layers are not refreshed in between;
   //  sweep = every cell runs the vertex / edge / face sweeps on its own copies with the total weights (bit-identical
   //          copies, no further exchange), then the lexicographic macro-cell sweep.
   // Backwards the reference communicates before every class, so `rest` is rebuilt (and exchanged) per class.
   void smooth_sor( const P1Function< double >& dst, const P1Function< double >& rhs, double relax, uint_t level, DoFType flag,
                    bool backwards = false ) const
   {
      if ( &dst == &rhs )
         throw std::runtime_error( "smooth_sor: dst and rhs must differ" );
      bool anyShell = false;
      forCells( [&]( uint_t, const MacroCell& cell ) { anyShell = anyShell || ( storage_->maskFor( cell, flag ) & HYTEG_HIP_MASK_SHELL ); } );
      auto sweepCells = [&]() {
         if ( storage_->useBatchSor( level ) )
         {
            const auto masks = storage_->masksFor( flag );
            storage_->forCellChunks( [&]( int first, int count ) {
               const auto u = dst.cellPointers( level, first, count ), r = rhs.cellPointers( level, first, count );
               hipCheck( hyteg_hip_p1_sor_cells( count, u.data(), r.data(), (int) level, stencilTable( level ) + (size_t) first * 225, relax,
                                                 backwards ? 1 : 0, masks.data() + first, storage_->stream() ),
                         "smooth_sor: cells (batched)" );
            } );
            return;
         }
         forCells( [&]( uint_t c, const MacroCell& cell ) {
            const unsigned mask = storage_->maskFor( cell, flag );
            if ( ( mask & HYTEG_HIP_MASK_INNER ) && level >= HYTEG_HIP_MIN_LEVEL )
               hipCheck( hyteg_hip_p1_sor_cell( dst.getCellPointer( c, level ), rhs.getCellPointer( c, level ), (int) level,
                                                getCellStencils( cell.id, level ).inner, relax, backwards ? 1 : 0, storage_->stream() ),
                         "smooth_sor: cell" );
         } );
      };
      if ( !anyShell && storage_->numRanks() == 1 )
      {
         sweepCells();
         return;
      }
      auto& restSlot = sorRest_[level];
      if ( !restSlot )
         restSlot.reset( new P1Function< double >( "sor_rest", storage_, level, level ) );
      P1Function< double >& rest = *restSlot;
      auto                 sweepShell = [&]( unsigned bits ) {
         if ( storage_->useBatch( level ) )
         {
            const auto masks = storage_->masksFor( flag, false, bits & HYTEG_HIP_MASK_SHELL );
            storage_->forCellChunks( [&]( int first, int count ) {
               const auto r = rest.cellPointers( level, first, count ), u = dst.cellPointers( level, first, count );
               hipCheck( hyteg_hip_p1_apply_cells( count, r.data(), u.data(), (int) level, restTable( level ) + (size_t) first * 225,
                                                   masks.data() + first, HYTEG_HIP_REPLACE, storage_->stream() ),
                         "smooth_sor: rest (batched)" );
            } );
         }
         else
         forCells( [&]( uint_t c, const MacroCell& cell ) {
            const auto&    T    = sorTables_.at( level ).at( cell.id );
            const unsigned mask = storage_->maskFor( cell, flag ) & bits;
            hipCheck( hyteg_hip_p1_apply_cell_boundary( rest.getCellPointer( c, level ), dst.getCellPointer( c, level ), (int) level,
                                                        &T.rest[0][0], mask, HYTEG_HIP_REPLACE, storage_->stream() ),
                      "smooth_sor: rest" );
         } );
         rest.sumSharedCopies( level, flag );
         if ( storage_->useBatch( level ) )
         {
            const auto masks = storage_->masksFor( flag, false, bits & HYTEG_HIP_MASK_SHELL );
            storage_->forCellChunks( [&]( int first, int count ) {
               const auto u = dst.cellPointers( level, first, count ), r = rhs.cellPointers( level, first, count ),
                          q = rest.cellPointers( level, first, count );
               hipCheck( hyteg_hip_p1_sor_shell_cells( count, u.data(), r.data(), q.data(), (int) level, shellTable( level ) + first, relax,
                                                       masks.data() + first, backwards ? 1 : 0, storage_->stream() ),
                         "smooth_sor: shell (batched)" );
            } );
            return;
         }
         forCells( [&]( uint_t c, const MacroCell& cell ) {
            const auto&    T    = sorTables_.at( level ).at( cell.id );
            const unsigned mask = storage_->maskFor( cell, flag ) & bits;
            hipCheck( hyteg_hip_p1_sor_shell_cell( dst.getCellPointer( c, level ), rhs.getCellPointer( c, level ),
                                                   rest.getCellPointer( c, level ), (int) level, &T.edgeVerts[0][0], &T.edgeW[0][0],
                                                   &T.faceVerts[0][0], &T.faceW[0][0], T.vertexW, relax, mask, backwards ? 1 : 0,
                                                   storage_->stream() ),
                      "smooth_sor: shell" );
         } );
      };
      if ( !backwards )
      {
         sweepShell( HYTEG_HIP_MASK_SHELL );
         sweepCells();
      }
      else
      {
         sweepCells();
         sweepShell( 0xFu << 6 );  // macro-faces
         sweepShell( 0x3Fu );      // macro-edges
         sweepShell( 0xFu << 10 ); // macro-vertices
      }
   }
   void smooth_gs( const P1Function< double >& dst, const P1Function< double >& rhs, uint_t level, DoFType flag ) const
   {
      smooth_sor( dst, rhs, 1.0, level, flag, false );
   }
   void smooth_sor_backwards( const P1Function< double >& dst, const P1Function< double >& rhs, double relax, uint_t level, DoFType flag ) const
   {
      smooth_sor( dst, rhs, relax, level, flag, true );
   }

   // P1Operator::computeInverseDiagonalOperatorValues, P1Operator.hpp:461-465, 636-906
   void computeInverseDiagonalOperatorValues()
   {
      inverseDiagonalValues_.reset( new P1Function< double >( "inverse diagonal entries", storage_, minLevel_, maxLevel_ ) );
      for ( uint_t l = minLevel_; l <= maxLevel_; ++l )
         for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         {
            const MacroCell& cell = storage_->getLocalCell( c );
            double*          d    = inverseDiagonalValues_->getCellPointer( c, l );
            hipCheck( hyteg_hip_p1_set_cell_masked( d, 1.0 / getCellStencils( cell.id, l ).inner[stencil::C], (int) l, HYTEG_HIP_MASK_INNER,
                                                    storage_->stream() ),
                      "inverse diagonal" );
            for ( int s = 0; s < 14; ++s )
            {
               // centre weight of a shared DoF = sum of the neighbour cells' shares (the reference adds the per-cell
               // centre entries of faceStencil3D / edgeStencil3D, P1Operator.hpp:700-870)
               const MacroPrimitive& p     = storage_->primitiveOfSlot( cell, s );
               double                total = 0.0;
               for ( int nc : p.cells )
               {
                  const MacroCell& other = storage_->getCells()[nc];
                  total += getCellStencils( nc, l ).slots[slotOf( other, p )][stencil::C];
               }
               hipCheck( hyteg_hip_p1_set_cell_masked( d, 1.0 / total, (int) l, 1u << s, storage_->stream() ), "inverse diagonal" );
            }
         }
   }
   std::shared_ptr< P1Function< double > > getInverseDiagonalValues() const
   {
      if ( !inverseDiagonalValues_ )
         throw std::runtime_error( "Inverse diagonal values have not been assembled, call computeInverseDiagonalOperatorValues() "
                                   "to set up this function." );
      return inverseDiagonalValues_;
   }

   // slot (0..13) under which primitive p appears in cell c
   static int slotOf( const MacroCell& c, const MacroPrimitive& p )
   {
      auto local = [&]( int g ) {
         for ( int q = 0; q < 4; ++q )
            if ( c.v[q] == g )
               return q;
         throw std::runtime_error( "slotOf: primitive is not part of the cell" );
      };
      if ( p.v.size() == 1 )
         return 10 + local( p.v[0] );
      if ( p.v.size() == 2 )
      {
         int a = local( p.v[0] ), b = local( p.v[1] );
         if ( a > b )
            std::swap( a, b );
         for ( int e = 0; e < 6; ++e )
            if ( kCellEdgeVerts[e][0] == a && kCellEdgeVerts[e][1] == b )
               return e;
      }
      int l[3] = { local( p.v[0] ), local( p.v[1] ), local( p.v[2] ) };
      std::sort( l, l + 3 );
      for ( int f = 0; f < 4; ++f )
         if ( kCellFaceVerts[f][0] == l[0] && kCellFaceVerts[f][1] == l[1] && kCellFaceVerts[f][2] == l[2] )
            return 6 + f;
      throw std::runtime_error( "slotOf: not found" );
   }

 private:
   template < typename F >
   void forCells( F&& fn ) const
   {
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
         fn( c, storage_->getLocalCell( c ) );
   }
   // Operator::apply with one launch for all local cells: every selected point gets (this cell's share of) its stencil sum,
   // then the shares of the shared points are summed.  Add needs the summed shares in a temporary first.
   void applyBatched( const P1Function< double >& src, const P1Function< double >& dst, uint_t level, DoFType flag, UpdateType updateType ) const
   {
      const bool sharedAdd = updateType == Add && hasSharedPoints( level, flag );
      auto       run       = [&]( const P1Function< double >& out, unsigned keep, int update ) {
         const auto masks = storage_->masksFor( flag, false, keep );
         storage_->forCellChunks( [&]( int first, int count ) {
            const auto d = out.cellPointers( level, first, count ), u = src.cellPointers( level, first, count );
            hipCheck( hyteg_hip_p1_apply_cells( count, d.data(), u.data(), (int) level, stencilTable( level ) + (size_t) first * 225,
                                                masks.data() + first, update, storage_->stream() ),
                      "apply (batched)" );
         } );
      };
      if ( !sharedAdd )
      {
         run( dst, HYTEG_HIP_MASK_ALL, updateType == Add ? HYTEG_HIP_ADD : HYTEG_HIP_REPLACE );
         dst.sumSharedCopies( level, flag );
         return;
      }
      P1Function< double > tmp( "apply_tmp", storage_, level, level, true );
      tmp.interpolate( 0.0, level, All );
      run( dst, HYTEG_HIP_MASK_INNER, HYTEG_HIP_ADD );
      run( tmp, HYTEG_HIP_MASK_SHELL, HYTEG_HIP_REPLACE );
      tmp.sumSharedCopies( level, flag );
      const auto masks = storage_->masksFor( flag, false, HYTEG_HIP_MASK_SHELL );
      storage_->forCellChunks( [&]( int first, int count ) {
         const auto   d = dst.cellPointers( level, first, count ), t = tmp.cellPointers( level, first, count );
         const double one = 1.0;
         hipCheck( hyteg_hip_p1_vector_cells( 1, count, d.data(), 1, t.data(), &one, (int) level, masks.data() + first, storage_->stream() ),
                   "apply: add shell (batched)" );
      } );
   }
 public:
   // ---- the whole CG solve in one launch for problems that fit one workgroup (hyteg_hip_p1_cg_small_cells) ----
   bool canCgSolveSmall( uint_t level ) const
   {
      const size_t n = storage_->getNumberOfLocalCells();
      return storage_->numRanks() == 1 && n >= 1 && n <= HYTEG_HIP_MAX_BATCH &&
             (int64_t) n * layout::cellSize( (int) level ) <= hyteg_hip_p1_cg_small_max_entries();
   }
   void cgSolveSmall( const P1Function< double >& x, const P1Function< double >& b, uint_t level, DoFType flag, uint_t maxIter, double relTol,
                      double absTol, double* infoDev ) const
   {
      const int  count = (int) storage_->getNumberOfLocalCells();
      const auto masks = storage_->masksFor( flag ), owned = storage_->masksFor( flag, true );
      const auto xs = x.cellPointers( level, 0, count ), bs = b.cellPointers( level, 0, count );
      const int* gp[2] = { nullptr, nullptr }, *ec[2] = { nullptr, nullptr }, *eo[2] = { nullptr, nullptr };
      int        ng[2] = { 0, 0 };
      for ( int cls = 0; cls < 2; ++cls )
      {
         if ( !testFlag( storage_->boundaryTypeOf( cls == 1 ), flag ) || storage_->exchangePlan( (int) level, cls ).ngroups() == 0 )
            continue;
         const auto& plan = storage_->devicePlan( (int) level, cls );
         gp[cls] = plan.dGroupPtr, ec[cls] = plan.dEntryBuf, eo[cls] = plan.dEntryOff, ng[cls] = plan.ngroups();
      }
      hipCheck( hyteg_hip_p1_cg_small_cells( count, xs.data(), bs.data(), (int) level, stencilTable( level ), masks.data(), owned.data(), gp, ec,
                                             eo, ng, (int) maxIter, relTol, absTol, infoDev, storage_->stream() ),
                "cgSolveSmall" );
   }

 private:
   // device tables [local cell][15 point classes][15 weights] for the batched kernels: classes 0..13 the cell's shares, 14 inner
   const double* stencilTable( uint_t level ) const
   {
      auto it = stencilTables_.find( level );
      if ( it != stencilTables_.end() )
         return it->second;
      std::vector< double > h;
      for ( int id : storage_->getLocalCellIDs() )
      {
         const auto& S = getCellStencils( id, level );
         h.insert( h.end(), &S.slots[0][0], &S.slots[0][0] + 14 * 15 );
         h.insert( h.end(), S.inner, S.inner + 15 );
      }
      return stencilTables_[level] = storage_->uploadTable( h );
   }
   const hyteg_hip_sor_shell_tables* shellTable( uint_t level ) const
   {
      auto it = shellTables_.find( level );
      if ( it != shellTables_.end() )
         return it->second;
      std::vector< hyteg_hip_sor_shell_tables > h;
      for ( int id : storage_->getLocalCellIDs() )
      {
         const auto&                T = sorTables_.at( level ).at( id );
         hyteg_hip_sor_shell_tables r;
         std::memcpy( r.edge_verts, T.edgeVerts, sizeof( r.edge_verts ) );
         std::memcpy( r.face_verts, T.faceVerts, sizeof( r.face_verts ) );
         std::memcpy( r.edge_w, T.edgeW, sizeof( r.edge_w ) );
         std::memcpy( r.face_w, T.faceW, sizeof( r.face_w ) );
         std::memcpy( r.vertex_w, T.vertexW, sizeof( r.vertex_w ) );
         h.push_back( r );
      }
      return shellTables_[level] =
                 static_cast< const hyteg_hip_sor_shell_tables* >( storage_->uploadBytes( h.data(), h.size() * sizeof( h[0] ) ) );
   }
   const double* restTable( uint_t level ) const
   {
      auto it = restTables_.find( level );
      if ( it != restTables_.end() )
         return it->second;
      std::vector< double > h;
      for ( int id : storage_->getLocalCellIDs() )
      {
         const auto& T = sorTables_.at( level ).at( id );
         h.insert( h.end(), &T.rest[0][0], &T.rest[0][0] + 14 * 15 );
         h.insert( h.end(), 15, 0.0 );
      }
      return restTables_[level] = storage_->uploadTable( h );
   }
   // total weights and sweep orientations of every macro-primitive, handed to each adjacent cell in its local numbering
   std::vector< stencil::CellSorTables > buildSorTables( const std::vector< stencil::CellStencils >& S ) const
   {
      using namespace stencil;
      const auto&                   cells = storage_->getCells();
      std::vector< CellSorTables >  T( cells.size() );
      std::vector< std::array< double, 3 > > edgeTot( storage_->getEdges().size(), std::array< double, 3 >{} );
      std::vector< std::array< double, 7 > > faceTot( storage_->getFaces().size(), std::array< double, 7 >{} );
      std::vector< double >                  vertTot( storage_->getVertices().size(), 0.0 );
      for ( const auto& c : cells )
      {
         CellSorTables& t = T[c.id];
         for ( int s = 0; s < 14; ++s )
            for ( int k = 0; k < 15; ++k )
               t.rest[s][k] = k == C ? 0.0 : S[c.id].slots[s][k];
         for ( int k = 0; k < 4; ++k )
            vertTot[c.v[k]] += S[c.id].slots[10 + k][C];
         for ( int e = 0; e < 6; ++e )
         {
            int lo = kCellEdgeVerts[e][0], hi = kCellEdgeVerts[e][1];
            if ( c.v[lo] > c.v[hi] )
               std::swap( lo, hi );
            const int kp = offsetIndex( kUnit[hi][0] - kUnit[lo][0], kUnit[hi][1] - kUnit[lo][1], kUnit[hi][2] - kUnit[lo][2] );
            const int km = offsetIndex( kUnit[lo][0] - kUnit[hi][0], kUnit[lo][1] - kUnit[hi][1], kUnit[lo][2] - kUnit[hi][2] );
            t.edgeVerts[e][0] = lo, t.edgeVerts[e][1] = hi;
            auto& tot = edgeTot[c.edges[e]];
            tot[0] += S[c.id].slots[e][C], tot[1] += S[c.id].slots[e][km], tot[2] += S[c.id].slots[e][kp];
            t.rest[e][km] = t.rest[e][kp] = 0.0;
         }
         for ( int f = 0; f < 4; ++f )
         {
            int l[3] = { kCellFaceVerts[f][0], kCellFaceVerts[f][1], kCellFaceVerts[f][2] };
            std::sort( l, l + 3, [&]( int a, int b ) { return c.v[a] < c.v[b]; } );
            auto& tot = faceTot[c.faces[f]];
            tot[0] += S[c.id].slots[6 + f][C];
            for ( int d = 0; d < 6; ++d )
            {
               int o[3];
               for ( int r = 0; r < 3; ++r )
                  o[r] = kFaceDirs[d][0] * ( kUnit[l[1]][r] - kUnit[l[0]][r] ) + kFaceDirs[d][1] * ( kUnit[l[2]][r] - kUnit[l[0]][r] );
               const int k = offsetIndex( o[0], o[1], o[2] );
               tot[1 + d] += S[c.id].slots[6 + f][k];
               t.rest[6 + f][k] = 0.0;
            }
            for ( int r = 0; r < 3; ++r )
               t.faceVerts[f][r] = l[r];
         }
      }
      for ( const auto& c : cells )
      {
         CellSorTables& t = T[c.id];
         for ( int k = 0; k < 4; ++k )
            t.vertexW[k] = vertTot[c.v[k]];
         for ( int e = 0; e < 6; ++e )
            for ( int k = 0; k < 3; ++k )
               t.edgeW[e][k] = edgeTot[c.edges[e]][k];
         for ( int f = 0; f < 4; ++f )
            for ( int k = 0; k < 7; ++k )
               t.faceW[f][k] = faceTot[c.faces[f]][k];
      }
      return T;
   }
   bool hasSharedPoints( uint_t level, DoFType flag ) const
   {
      for ( int cls = 0; cls < 2; ++cls )
         if ( testFlag( storage_->boundaryTypeOf( cls == 1 ), flag ) && storage_->exchangePlan( (int) level, cls ).ngroups() > 0 )
            return true;
      return false;
   }

   std::shared_ptr< PrimitiveStorage >                        storage_;
   uint_t                                                     minLevel_, maxLevel_;
   uint64_t                                                   uid_ = nextUid();
   std::map< uint_t, std::vector< stencil::CellStencils > >   stencils_;
   std::map< uint_t, std::vector< stencil::CellSorTables > >  sorTables_;
   mutable std::map< uint_t, std::unique_ptr< P1Function< double > > > sorRest_;
   mutable std::map< uint_t, const double* >                  stencilTables_, restTables_;
   mutable std::map< uint_t, const hyteg_hip_sor_shell_tables* > shellTables_;
   std::shared_ptr< P1Function< double > >                    inverseDiagonalValues_;
};

using P1ConstantLaplaceOperator = P1ConstantOperator< forms::P1LaplaceForm >; // P1ConstantOperator.hpp:167-168
using P1ConstantMassOperator    = P1ConstantOperator< forms::P1MassForm >;

// =====================================================================================================
// P2ElementwiseOperator< P2Form >  ( src/hyteg/elementwiseoperators/P2ElementwiseOperator.hpp:454, .cpp:110-223 ), affine cells:
// the six element matrices per (cell, level) are computed once (kernel INPUT) and kept on the device.
// =====================================================================================================
namespace forms {
// P2 diffusion element matrix in FEniCS ordering (vertices 0-3, edges (2,3) (1,3) (1,2) (0,3) (0,2) (0,1)); what
// P2FenicsForm< ..., p2_tet_diffusion_cell_integral_0_otherwise >::integrateAll returns (form_fenics_base/P2FenicsForm.cpp:160-175).
// Closed form: phi_a = l_a (2 l_a - 1), phi_ab = 4 l_a l_b; int l_a = V/4, int l_a l_b = V (1 + delta_ab) / 20.
struct P2LaplaceForm
{
   static void integrateAll( const std::array< Point3D, 4 >& c, double elMat[100] )
   {
      double J[3][3];
      for ( int r = 0; r < 3; ++r )
         for ( int k = 0; k < 3; ++k )
            J[r][k] = c[k + 1][r] - c[0][r];
      const double det = det3( J );
      double       Ji[3][3];
      Ji[0][0] = ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) / det;
      Ji[0][1] = ( J[0][2] * J[2][1] - J[0][1] * J[2][2] ) / det;
      Ji[0][2] = ( J[0][1] * J[1][2] - J[0][2] * J[1][1] ) / det;
      Ji[1][0] = ( J[1][2] * J[2][0] - J[1][0] * J[2][2] ) / det;
      Ji[1][1] = ( J[0][0] * J[2][2] - J[0][2] * J[2][0] ) / det;
      Ji[1][2] = ( J[0][2] * J[1][0] - J[0][0] * J[1][2] ) / det;
      Ji[2][0] = ( J[1][0] * J[2][1] - J[1][1] * J[2][0] ) / det;
      Ji[2][1] = ( J[0][1] * J[2][0] - J[0][0] * J[2][1] ) / det;
      Ji[2][2] = ( J[0][0] * J[1][1] - J[0][1] * J[1][0] ) / det;
      double g[4][3];
      for ( int r = 0; r < 3; ++r )
      {
         g[1][r] = Ji[0][r], g[2][r] = Ji[1][r], g[3][r] = Ji[2][r];
         g[0][r] = -( Ji[0][r] + Ji[1][r] + Ji[2][r] );
      }
      const double V = std::fabs( det ) / 6.0;
      double       G[4][4];
      for ( int a = 0; a < 4; ++a )
         for ( int b = 0; b < 4; ++b )
            G[a][b] = g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2];
      // grad phi_i = sum_a ( sum_p C[i][a][p] l_p + D[i][a] ) grad l_a
      static const int pairs[6][2] = { { 2, 3 }, { 1, 3 }, { 1, 2 }, { 0, 3 }, { 0, 2 }, { 0, 1 } };
      double           C[10][4][4] = {}, D[10][4] = {};
      for ( int a = 0; a < 4; ++a )
         C[a][a][a] = 4.0, D[a][a] = -1.0;
      for ( int k = 0; k < 6; ++k )
         C[4 + k][pairs[k][0]][pairs[k][1]] = 4.0, C[4 + k][pairs[k][1]][pairs[k][0]] = 4.0;
      for ( int i = 0; i < 10; ++i )
         for ( int j = 0; j < 10; ++j )
         {
            double s = 0.0;
            for ( int a = 0; a < 4; ++a )
               for ( int b = 0; b < 4; ++b )
               {
                  double t = D[i][a] * D[j][b] * V;
                  for ( int p = 0; p < 4; ++p )
                  {
                     t += ( C[i][a][p] * D[j][b] + D[i][a] * C[j][b][p] ) * V / 4.0;
                     for ( int q = 0; q < 4; ++q )
                        t += C[i][a][p] * C[j][b][q] * V * ( p == q ? 2.0 : 1.0 ) / 20.0;
                  }
                  s += t * G[a][b];
               }
            elMat[10 * i + j] = s;
         }
   }
};
} // namespace forms

template < class P2Form >
class P2ElementwiseOperator
{
 public:
   using srcType = P2Function< double >;
   using dstType = P2Function< double >;
   P2ElementwiseOperator( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel )
   : storage_( storage )
   , minLevel_( minLevel )
   , maxLevel_( maxLevel )
   {
      if ( storage->numRanks() != 1 )
         throw std::runtime_error( "P2ElementwiseOperator: storages distributed over several ranks are not supported in this version" );
      // micro-cell vertex offsets of the six cell types, celldof::macrocell::getMicroVerticesFromMicroCell (CellDoFIndexing.hpp:155-198)
      static const int verts[6][4][3] = {
          { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } }, { { 1, 0, 0 }, { 1, 1, 0 }, { 0, 1, 0 }, { 1, 0, 1 } },
          { { 1, 0, 0 }, { 0, 1, 0 }, { 1, 0, 1 }, { 0, 0, 1 } }, { { 1, 1, 0 }, { 1, 1, 1 }, { 0, 1, 1 }, { 1, 0, 1 } },
          { { 1, 0, 1 }, { 0, 1, 1 }, { 0, 0, 1 }, { 0, 1, 0 } }, { { 0, 1, 0 }, { 1, 1, 0 }, { 1, 0, 1 }, { 0, 1, 1 } } };
      for ( uint_t l = minLevel; l <= maxLevel; ++l )
         for ( uint_t lc = 0; lc < storage->getNumberOfLocalCells(); ++lc )
         {
            const MacroCell&      cell = storage->getLocalCell( lc );
            const double          step = 1.0 / double( int64_t( 1 ) << l );
            std::vector< double > h( 600 );
            for ( int t = 0; t < 6; ++t )
            {
               std::array< Point3D, 4 > c;
               for ( int k = 0; k < 4; ++k )
                  for ( int r = 0; r < 3; ++r )
                     c[k][r] = cell.coords[0][r] + ( cell.coords[1][r] - cell.coords[0][r] ) * step * verts[t][k][0] +
                               ( cell.coords[2][r] - cell.coords[0][r] ) * step * verts[t][k][1] +
                               ( cell.coords[3][r] - cell.coords[0][r] ) * step * verts[t][k][2];
               P2Form::integrateAll( c, h.data() + 100 * t );
            }
            std::vector< double > table( hyteg_hip_p2_operator_table_size() );
            hipCheck( hyteg_hip_p2_build_operator_table( h.data(), table.data() ), "P2ElementwiseOperator: operator table" );
            elementMatrices_[l].push_back( storage->uploadTable( table ) );
            hostMatrices_[l].push_back( h );
         }
   }
   std::shared_ptr< PrimitiveStorage > getStorage() const { return storage_; }
   const std::vector< double >&        getElementMatrices( uint_t level, uint_t localCell = 0 ) const { return hostMatrices_.at( level ).at( localCell ); }

   // Operator::apply = gemv( 1, src, updateType == Replace ? 0 : 1, dst ), P2ElementwiseOperator.hpp:60-75
   void apply( const P2Function< double >& src, const P2Function< double >& dst, uint_t level, DoFType flag, UpdateType updateType = Replace ) const
   {
      gemv( 1.0, src, updateType == Replace ? 0.0 : 1.0, dst, level, flag );
   }
   // every cell adds the contributions of its own micro-cells; on DoFs shared by several cells these are partial sums that the
   // additive exchange completes (communicateAdditively< Cell, ... > at the end of the reference's gemv, :225-235)
   void gemv( double alpha, const P2Function< double >& src, double beta, const P2Function< double >& dst, uint_t level, DoFType flag ) const
   {
      if ( &src == &dst )
         throw std::runtime_error( "P2ElementwiseOperator::gemv: src and dst must differ" );
      const bool shared = storage_->getNumberOfLocalCells() > 1;
      if ( !shared )
      {
         if ( beta != 0.0 && beta != 1.0 )
            dst.assign( { beta }, { dst }, level, flag );
         launch( alpha, src, dst, level, flag, HYTEG_HIP_MASK_ALL, beta == 0.0 ? HYTEG_HIP_REPLACE : HYTEG_HIP_ADD );
         return;
      }
      if ( beta == 0.0 )
      {
         launch( alpha, src, dst, level, flag, HYTEG_HIP_MASK_ALL, HYTEG_HIP_REPLACE );
         dst.getVertexDoFFunction().sumSharedCopies( level, flag );
         dst.sumSharedEdgeCopies( level, flag );
         return;
      }
      // beta != 0: the summed shares of the shared DoFs are formed in a temporary and then added
      P2Function< double > tmp( "p2_gemv_tmp", storage_, level, level );
      launch( alpha, src, tmp, level, flag, HYTEG_HIP_MASK_ALL, HYTEG_HIP_REPLACE );
      tmp.getVertexDoFFunction().sumSharedCopies( level, flag );
      tmp.sumSharedEdgeCopies( level, flag );
      dst.assign( { beta, 1.0 }, { dst, tmp }, level, flag );
   }

 private:
   void launch( double alpha, const P2Function< double >& src, const P2Function< double >& dst, uint_t level, DoFType flag, unsigned keep,
                int update ) const
   {
      for ( uint_t c = 0; c < storage_->getNumberOfLocalCells(); ++c )
      {
         const MacroCell& cell = storage_->getLocalCell( c );
         hipCheck( hyteg_hip_p2_elementwise_apply_cell( dst.getVertexDoFFunction().getCellPointer( c, level ), dst.getEdgeCellPointer( c, level ),
                                                        src.getVertexDoFFunction().getCellPointer( c, level ), src.getEdgeCellPointer( c, level ),
                                                        (int) level, elementMatrices_.at( level ).at( c ), alpha, update,
                                                        storage_->maskFor( cell, flag ) & keep, storage_->stream() ),
                   "P2ElementwiseOperator::gemv" );
      }
   }
   std::shared_ptr< PrimitiveStorage >                          storage_;
   uint_t                                                       minLevel_, maxLevel_;
   std::map< uint_t, std::vector< const double* > >             elementMatrices_;
   std::map< uint_t, std::vector< std::vector< double > > >     hostMatrices_;
};
using P2ElementwiseLaplaceOperator = P2ElementwiseOperator< forms::P2LaplaceForm >; // P2ElementwiseOperator.hpp:454

// =====================================================================================================
// Grid transfer ( src/hyteg/gridtransferoperators/P1toP1LinearRestriction.cpp:169-346, P1toP1LinearProlongation.cpp:194-410 )
// =====================================================================================================
class P1toP1LinearRestriction
{
 public:
   void restrict( const P1Function< double >& function, const uint_t& sourceLevel, const DoFType& flag ) const
   {
      auto         storage = function.getStorage();
      const uint_t dstLevel = sourceLevel - 1;
      if ( storage->useBatch( sourceLevel ) )
      {
         const auto masks = storage->masksFor( flag );
         storage->forCellChunks( [&]( int first, int count ) {
            const auto co = function.cellPointers( dstLevel, first, count ), fi = function.cellPointers( sourceLevel, first, count );
            hipCheck( hyteg_hip_p1_restrict_cells( count, co.data(), fi.data(), (int) dstLevel, storage->nncInvDevice() + (size_t) first * 14,
                                                   masks.data() + first, storage->stream() ),
                      "restrict (batched)" );
         } );
         function.sumSharedCopies( dstLevel, flag );
         return;
      }
      for ( uint_t c = 0; c < storage->getNumberOfLocalCells(); ++c )
      {
         const MacroCell& cell = storage->getLocalCell( c );
         const auto       nnc  = storage->numNeighborCells( cell );
         hipCheck( hyteg_hip_p1_restrict_cell_masked( function.getCellPointer( c, dstLevel ), function.getCellPointer( c, sourceLevel ),
                                                      (int) dstLevel, nnc.data(), storage->maskFor( cell, flag ), storage->stream() ),
                   "restrict" );
      }
      // communicateAdditively< Cell, {Vertex,Edge,Face} >( dstLevel, flag ^ All, ... ) (:343-345)
      function.sumSharedCopies( dstLevel, flag );
   }
};

class P1toP1LinearProlongation
{
 public:
   void prolongate( const P1Function< double >& function, const uint_t& sourceLevel, const DoFType& flag ) const
   {
      run( function, function, sourceLevel, flag );
   }
   void prolongateAndAdd( const P1Function< double >& function, const uint_t& sourceLevel, const DoFType& flag ) const
   {
      // the prolongated correction is formed in a temporary (Replace), summed over cells on shared points, then added.
      // The temporary needs no initialisation: the masked kernel writes every point `flag` selects, the sum over
      // shared copies and the add read only those.
      auto                 storage = function.getStorage();
      P1Function< double > tmp( "prolongate_tmp", storage, sourceLevel + 1, sourceLevel + 1, true );
      run( function, tmp, sourceLevel, flag );
      function.add( { 1.0 }, { tmp }, sourceLevel + 1, flag );
   }

 private:
   static void run( const P1Function< double >& src, const P1Function< double >& dst, uint_t sourceLevel, DoFType flag )
   {
      auto storage = src.getStorage();
      if ( storage->useBatch( sourceLevel + 1 ) )
      {
         const auto masks = storage->masksFor( flag );
         storage->forCellChunks( [&]( int first, int count ) {
            const auto co = src.cellPointers( sourceLevel, first, count ), fi = dst.cellPointers( sourceLevel + 1, first, count );
            hipCheck( hyteg_hip_p1_prolongate_cells( count, co.data(), fi.data(), (int) sourceLevel,
                                                     storage->nncInvDevice() + (size_t) first * 14, masks.data() + first, HYTEG_HIP_REPLACE,
                                                     storage->stream() ),
                      "prolongate (batched)" );
         } );
         dst.sumSharedCopies( sourceLevel + 1, flag );
         return;
      }
      for ( uint_t c = 0; c < storage->getNumberOfLocalCells(); ++c )
      {
         const MacroCell& cell = storage->getLocalCell( c );
         const auto       nnc  = storage->numNeighborCells( cell );
         hipCheck( hyteg_hip_p1_prolongate_cell_masked( src.getCellPointer( c, sourceLevel ), dst.getCellPointer( c, sourceLevel + 1 ),
                                                        (int) sourceLevel, nnc.data(), storage->maskFor( cell, flag ), storage->stream() ),
                   "prolongate" );
      }
      dst.sumSharedCopies( sourceLevel + 1, flag );
   }
};

// =====================================================================================================
// Solvers ( src/hyteg/solvers/ )
// =====================================================================================================
template < class OperatorType >
class Solver
{
 public:
   using FunctionType = typename OperatorType::srcType;
   virtual ~Solver()  = default;
   virtual void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) = 0;
   // `steps` consecutive solve() calls (the pre-/post-smoothing loops of GeometricMultigridSolver.hpp:228-233,300-305);
   // a smoother may override it with something equivalent but cheaper
   virtual void solveSteps( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level, uint_t steps )
   {
      for ( uint_t i = 0; i < steps; ++i )
         solve( A, x, b, level );
   }
};

// WeightedJacobiSmoother.hpp:46-62
template < class OperatorType >
class WeightedJacobiSmoother : public Solver< OperatorType >
{
 public:
   WeightedJacobiSmoother( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel, double relax )
   : relax_( relax )
   , tmp_( "weighted_jacobi_tmp", storage, minLevel, maxLevel )
   , flag_( Inner | NeumannBoundary )
   {}
   void solve( const OperatorType& A, const P1Function< double >& x, const P1Function< double >& b, uint_t level ) override
   {
      tmp_.assign( { 1.0 }, { x }, level, All );
      A.smooth_jac( x, b, tmp_, relax_, level, flag_ );
   }
   // n steps with ONE copy instead of n: after tmp = x (all points) a Jacobi step may just as well write into tmp
   // reading x, since a step only writes the points `flag_` selects and all other entries of the two functions agree.
   // Every step computes exactly what solve() computes (same kernel, same operands): results are bit-identical.
   void solveSteps( const OperatorType& A, const P1Function< double >& x, const P1Function< double >& b, uint_t level, uint_t steps ) override
   {
      if ( steps == 0 )
         return;
      tmp_.assign( { 1.0 }, { x }, level, All );
      for ( uint_t i = 0; i < steps; ++i )
      {
         if ( i % 2 == 0 )
            A.smooth_jac( x, b, tmp_, relax_, level, flag_ );
         else
            A.smooth_jac( tmp_, b, x, relax_, level, flag_ );
      }
      if ( steps % 2 == 0 )
         x.assign( { 1.0 }, { tmp_ }, level, flag_ );
   }

 private:
   double               relax_;
   P1Function< double > tmp_;
   DoFType              flag_;
};

// GaussSeidelSmoother.hpp:38-50, SORSmoother.hpp:33-46
template < class OperatorType >
class SORSmoother : public Solver< OperatorType >
{
 public:
   explicit SORSmoother( double relax )
   : relax_( relax )
   , flag_( Inner | NeumannBoundary )
   {}
   void solve( const OperatorType& A, const P1Function< double >& x, const P1Function< double >& b, uint_t level ) override
   {
      A.smooth_sor( x, b, relax_, level, flag_ );
   }

 private:
   double  relax_;
   DoFType flag_;
};
template < class OperatorType >
class GaussSeidelSmoother : public SORSmoother< OperatorType >
{
 public:
   GaussSeidelSmoother()
   : SORSmoother< OperatorType >( 1.0 )
   {}
};

// CGSolver.hpp:88-205 (no preconditioner: IdentityPreconditioner)
template < class OperatorType >
class CGSolver : public Solver< OperatorType >
{
 public:
   using FunctionType = typename OperatorType::srcType;
   CGSolver( const std::shared_ptr< PrimitiveStorage >& storage, uint_t minLevel, uint_t maxLevel, uint_t maxIter = 1000,
             double relativeTolerance = 1e-16, double absoluteTolerance = 1e-16 )
   : p_( "p", storage, minLevel, maxLevel )
   , z_( "z", storage, minLevel, maxLevel )
   , ap_( "ap", storage, minLevel, maxLevel )
   , r_( "r", storage, minLevel, maxLevel )
   , flag_( Inner | NeumannBoundary )
   , maxIter_( maxIter )
   , relTol_( relativeTolerance )
   , absTol_( absoluteTolerance )
   {}
   ~CGSolver() override
   {
      if ( scalars_ )
         hyteg_hip_free( scalars_ );
   }
   // Device-resident scalars (no counterpart in the reference, whose loop reads every dot product on the host): on
   // coarse levels a CG iteration is ~10 launches of a few microseconds, and three host round trips per iteration
   // cost more than the launches.  alpha, beta and the convergence test are computed by a one-thread kernel
   // (hyteg_hip_cg_scalars), the vector updates read them from device memory, and the host looks at the convergence
   // flag every 4 iterations; iterations enqueued after convergence are exact no-ops (alpha = 0).  Same recurrences,
   // same arithmetic as the loop below.  Used for storages of one rank, up to level 5; off: HYTEG_AMD_DEVICE_CG=0.
   void setUseDeviceScalars( bool on ) { useDeviceScalars_ = on; }

   void solve( const OperatorType& A, const FunctionType& x, const FunctionType& b, uint_t level ) override
   {
      iterationsOnDevice_ = false;
      if ( deviceScalarsUsable( x, level ) )
      {
         solveWithDeviceScalars( A, x, b, level );
         return;
      }
      p_.setToZero( level );
      z_.setToZero( level );
      ap_.setToZero( level );
      r_.setToZero( level );
      // init(): r = b - A x ; z = r ; p = z ; prsold = <r,z>
      A.apply( x, p_, level, flag_, Replace );
      r_.assign( { 1.0, -1.0 }, { b, p_ }, level, flag_ );
      z_.assign( { 1.0 }, { r_ }, level, flag_ );
      p_.assign( { 1.0 }, { z_ }, level, flag_ );
      double       prsold    = r_.dotGlobal( z_, level, flag_ );
      const double res_start = std::sqrt( r_.dotGlobal( r_, level, flag_ ) );
      iterations_            = 0;
      if ( res_start < absTol_ )
         return;
      for ( uint_t i = 0; i < maxIter_; ++i )
      {
         A.apply( p_, ap_, level, flag_, Replace );
         const double pAp   = p_.dotGlobal( ap_, level, flag_ );
         const double alpha = prsold / pAp;
         x.add( { alpha }, { p_ }, level, flag_ );
         r_.add( { -alpha }, { ap_ }, level, flag_ );
         const double rsnew   = r_.dotGlobal( r_, level, flag_ );
         const double sqrsnew = std::sqrt( rsnew );
         iterations_          = i + 1;
         if ( sqrsnew / res_start < relTol_ || sqrsnew < absTol_ )
            break;
         // identity preconditioner: z = r, so <r,z> is the <r,r> just computed (the reference copies and reduces again)
         const double prsnew = rsnew;
         const double beta   = prsnew / prsold;
         p_.assign( { 1.0, beta }, { r_, p_ }, level, flag_ );
         prsold = prsnew;
      }
   }
   uint_t getIterations() const
   {
      if ( iterationsOnDevice_ )
      {
         // the one-launch solve leaves its iteration count on the device; fetched (one synchronisation) only when asked for
         double h = 0.0;
         hipCheck( hyteg_hip_download( &h, scalars_, sizeof( double ), iterationsStream_ ), "CGSolver: iterations" );
         iterations_         = (uint_t) h;
         iterationsOnDevice_ = false;
      }
      return iterations_;
   }
   // levels whose cell arrays together fit one workgroup are solved by ONE launch (hyteg_hip_p1_cg_small_cells); off: false
   void setUseSingleLaunch( bool on ) { useSingleLaunch_ = on; }

 private:
   template < typename F >
   bool deviceScalarsUsable( const F&, uint_t ) const
   {
      return false; // P2 functions: host scalars
   }
   bool deviceScalarsUsable( const P1Function< double >& x, uint_t level ) const
   {
      static const bool envOn = [] {
         const char* e = std::getenv( "HYTEG_AMD_DEVICE_CG" );
         return !( e && e[0] == '0' );
      }();
      const auto& st = *x.getStorage();
      return envOn && useDeviceScalars_ && st.numRanks() == 1 && level <= 5 && st.getNumberOfLocalCells() >= 1 &&
             st.getNumberOfLocalCells() <= HYTEG_HIP_MAX_BATCH;
   }
   template < typename F >
   void solveWithDeviceScalars( const OperatorType&, const F&, const F&, uint_t )
   {}
   void solveWithDeviceScalars( const OperatorType& A, const P1Function< double >& x, const P1Function< double >& b, uint_t level )
   {
      const auto& st = *x.getStorage();
      if ( !scalars_ )
      {
         void* d = nullptr;
         hipCheck( hyteg_hip_malloc( &d, HYTEG_HIP_CG_SLOTS * sizeof( double ) ), "CGSolver: scalars" );
         scalars_ = static_cast< double* >( d );
      }
      double* const S = scalars_;
      if ( useSingleLaunch_ && A.canCgSolveSmall( level ) )
      {
         A.cgSolveSmall( x, b, level, flag_, maxIter_, relTol_, absTol_, S );
         iterationsOnDevice_ = true;
         iterationsStream_   = st.stream(); // the download has to be ordered after the solve on ITS stream
         return;
      }
      p_.setToZero( level ); // apply( p ) reads p on every point; assign below writes only the points flag_ selects
      A.apply( x, p_, level, flag_, Replace );
      r_.assign( { 1.0, -1.0 }, { b, p_ }, level, flag_ );
      p_.assign( { 1.0 }, { r_ }, level, flag_ );
      hipCheck( hyteg_hip_memset_zero( S, HYTEG_HIP_CG_SLOTS * sizeof( double ), st.stream() ), "CGSolver: scalars reset" );
      r_.dotLocalToCgScalars( r_, level, flag_, S, HYTEG_HIP_CG_RR, 0, relTol_, absTol_ );
      iterations_ = 0;
      for ( uint_t i = 0; i < maxIter_; ++i )
      {
         A.apply( p_, ap_, level, flag_, Replace );
         p_.dotLocalToCgScalars( ap_, level, flag_, S, HYTEG_HIP_CG_PAP, 1, relTol_, absTol_ );
         x.vectorOpDeviceScalars( 1, { S + HYTEG_HIP_CG_ALPHA }, { p_ }, level, flag_ );
         r_.vectorOpDeviceScalars( 1, { S + HYTEG_HIP_CG_NEG_ALPHA }, { ap_ }, level, flag_ );
         r_.dotLocalToCgScalars( r_, level, flag_, S, HYTEG_HIP_CG_RR, 2, relTol_, absTol_ );
         p_.vectorOpDeviceScalars( 0, { S + HYTEG_HIP_CG_ONE, S + HYTEG_HIP_CG_BETA }, { r_, p_ }, level, flag_ );
         if ( ( i + 1 ) % 4 == 0 || i + 1 == maxIter_ )
         {
            double h[2];
            hipCheck( hyteg_hip_download( h, S + HYTEG_HIP_CG_DONE, 2 * sizeof( double ), st.stream() ), "CGSolver: convergence flag" );
            iterations_ = (uint_t) h[1];
            if ( h[0] != 0.0 )
               break;
         }
      }
   }

   bool                 useDeviceScalars_ = true, useSingleLaunch_ = true;
   mutable bool         iterationsOnDevice_ = false;
   hyteg_hip_stream_t   iterationsStream_   = nullptr;
   double*              scalars_          = nullptr;
   FunctionType         p_, z_, ap_, r_;
   DoFType              flag_;
   uint_t               maxIter_;
   double               relTol_, absTol_;
   mutable uint_t       iterations_ = 0;
};

// GeometricMultigridSolver.hpp:40-330
template < class OperatorType >
class GeometricMultigridSolver : public Solver< OperatorType >
{
 public:
   GeometricMultigridSolver( const std::shared_ptr< PrimitiveStorage >&         storage,
                             std::shared_ptr< Solver< OperatorType > >          smoother,
                             std::shared_ptr< Solver< OperatorType > >          coarseSolver,
                             std::shared_ptr< P1toP1LinearRestriction >         restrictionOperator,
                             std::shared_ptr< P1toP1LinearProlongation >        prolongationOperator,
                             uint_t                                             minLevel,
                             uint_t                                             maxLevel,
                             uint_t                                             preSmoothSteps  = 3,
                             uint_t                                             postSmoothSteps = 3,
                             uint_t                                             smoothIncrement = 0,
                             CycleType                                          cycleType       = CycleType::VCYCLE )
   : minLevel_( minLevel )
   , maxLevel_( maxLevel )
   , preSmoothSteps_( preSmoothSteps )
   , postSmoothSteps_( postSmoothSteps )
   , smoothIncrement_( smoothIncrement )
   , flag_( Inner | NeumannBoundary )
   , cycleType_( cycleType )
   , smoother_( smoother )
   , coarseSolver_( coarseSolver )
   , restrictionOperator_( restrictionOperator )
   , prolongationOperator_( prolongationOperator )
   , tmp_( "gmg_tmp", storage, minLevel, maxLevel )
   , storage_( storage )
   {}

   ~GeometricMultigridSolver() override
   {
      for ( auto& kv : recordings_ )
         kv.second.destroy();
      if ( captureStream_ )
         hyteg_hip_stream_destroy( captureStream_ );
   }

   // Launch graphs (no counterpart in the reference, whose cycle is host loops): the launches of a cycle -- ~20 per
   // level, most of them a few microseconds on the coarse levels -- are recorded once per (operator, x, b, level) and
   // replayed as one graph launch per segment between coarse-grid solves (the coarse solver reads dot products on the
   // host and stays outside).  The first cycle with given arguments runs with ordinary launches (it creates every lazily
   // built table and scratch array), the second records (nothing executes while recording) and replays, later cycles
   // replay.  Same kernels, same order, same arguments: results are identical to ordinary launches.
   // Opt-in: setUseGraphs( true ) or HYTEG_AMD_GRAPHS=1.  Measured on MI355X the replay saves only 1-8 % of a cycle (the
   // cycle is bound by the ~3 us a dependent small kernel takes on the GPU, not by the host's launch rate), while
   // recording and instantiating costs a few milliseconds once -- it pays for solves of many cycles only.  Never
   // used for storages distributed over several ranks (the exchange hooks are host callbacks).
   void setUseGraphs( bool on ) { useGraphs_ = on; }
   std::shared_ptr< Solver< OperatorType > > getCoarseSolver() const { return coarseSolver_; }
   bool usesGraphs() const { return graphsUsable(); }
   // number of cycles that were replayed from a recording (tests)
   uint_t replayedCycles() const { return replayed_; }

   void solve( const OperatorType& A, const P1Function< double >& x, const P1Function< double >& b, uint_t level ) override
   {
      invokedLevel_ = level;
      if ( !graphsUsable() )
      {
         solveRecursively( A, x, b, level );
         return;
      }
      const Key key{ A.uid(), x.uid(), b.uid(), level };
      auto      it = recordings_.find( key );
      if ( it == recordings_.end() )
      {
         if ( recordings_.size() >= 8 )
         {
            for ( auto& kv : recordings_ )
               kv.second.destroy();
            recordings_.clear();
         }
         recordings_[key] = Recording{};
         solveRecursively( A, x, b, level );
         return;
      }
      Recording& rec = it->second;
      if ( !rec.recorded && !rec.failed )
         record( rec, A, x, b, level );
      if ( !rec.recorded )
      {
         solveRecursively( A, x, b, level );
         return;
      }
      for ( size_t k = 0; k < rec.segments.size(); ++k )
      {
         hipCheck( hyteg_hip_graph_launch( rec.segments[k], storage_->stream() ), "GeometricMultigridSolver: graph launch" );
         if ( k + 1 < rec.segments.size() )
            coarseSolver_->solve( A, x, b, minLevel_ );
      }
      ++replayed_;
   }

 private:
   using Key = std::tuple< uint64_t, uint64_t, uint64_t, uint_t >;
   struct Recording
   {
      std::vector< hyteg_hip_graph_t > segments; // separated by coarse-grid solves
      bool                             recorded = false, failed = false;
      int                              attempts = 0;
      void                             destroy()
      {
         for ( auto g : segments )
            hyteg_hip_graph_destroy( g );
         segments.clear();
      }
   };

   bool graphsUsable() const
   {
      static const bool envOn = [] {
         const char* e = std::getenv( "HYTEG_AMD_GRAPHS" );
         return e && e[0] == '1';
      }();
      return ( useGraphs_ || envOn ) && storage_->numRanks() == 1;
   }

   void endSegment( Recording& rec )
   {
      hyteg_hip_graph_t g = nullptr;
      capturing_          = false;
      hipCheck( hyteg_hip_graph_end_capture( captureStream_, &g ), "GeometricMultigridSolver: end capture" );
      rec.segments.push_back( g );
   }
   void beginSegment()
   {
      hipCheck( hyteg_hip_graph_begin_capture( captureStream_ ), "GeometricMultigridSolver: begin capture" );
      capturing_ = true;
   }

   void record( Recording& rec, const OperatorType& A, const P1Function< double >& x, const P1Function< double >& b, uint_t level )
   {
      const hyteg_hip_stream_t user = storage_->stream();
      try
      {
         if ( !captureStream_ )
            hipCheck( hyteg_hip_stream_create( &captureStream_ ), "GeometricMultigridSolver: stream" );
         storage_->setStream( captureStream_ );
         recording_ = &rec;
         beginSegment();
         solveRecursively( A, x, b, level );
         endSegment( rec );
         rec.recorded = true;
      } catch ( const std::exception& e )
      {
         // something in the cycle cannot be recorded: nothing has executed, fall back to ordinary launches for good
         if ( rec.attempts >= 2 )
            std::fprintf( stderr, "hyteg_amd: multigrid cycle not recordable (%s); using ordinary launches\n", e.what() );
         if ( capturing_ )
            hyteg_hip_graph_abort_capture( captureStream_ );
         capturing_ = false;
         rec.destroy();
         rec.failed = ++rec.attempts >= 3; // a table built lazily in this very cycle: the next cycle tries again
      }
      recording_ = nullptr;
      storage_->setStream( user );
   }

   void solveRecursively( const OperatorType& A, const P1Function< double >& x, const P1Function< double >& b, uint_t level )
   {
      if ( level == minLevel_ )
      {
         if ( recording_ )
         {
            endSegment( *recording_ );
            beginSegment();
         }
         else
            coarseSolver_->solve( A, x, b, minLevel_ );
         return;
      }
      const uint_t pre = preSmoothSteps_ + smoothIncrement_ * ( invokedLevel_ - level );
      smoother_->solveSteps( A, x, b, level, pre );
      A.apply( x, tmp_, level, flag_ );
      tmp_.assign( { 1.0, -1.0 }, { b, tmp_ }, level, flag_ );
      restrictionOperator_->restrict( tmp_, level, flag_ );
      b.assign( { 1.0 }, { tmp_ }, level - 1, flag_ );
      x.interpolate( 0.0, level - 1 );
      solveRecursively( A, x, b, level - 1 );
      if ( cycleType_ == CycleType::WCYCLE )
         solveRecursively( A, x, b, level - 1 );
      prolongationOperator_->prolongateAndAdd( x, level - 1, flag_ );
      const uint_t post = postSmoothSteps_ + smoothIncrement_ * ( invokedLevel_ - level );
      smoother_->solveSteps( A, x, b, level, post );
   }

   uint_t                                       minLevel_, maxLevel_, preSmoothSteps_, postSmoothSteps_, smoothIncrement_;
   uint_t                                       invokedLevel_ = 0;
   DoFType                                      flag_;
   CycleType                                    cycleType_;
   std::shared_ptr< Solver< OperatorType > >    smoother_, coarseSolver_;
   std::shared_ptr< P1toP1LinearRestriction >   restrictionOperator_;
   std::shared_ptr< P1toP1LinearProlongation >  prolongationOperator_;
   P1Function< double >                         tmp_;
   std::shared_ptr< PrimitiveStorage >          storage_;
   bool                                         useGraphs_ = false, capturing_ = false;
   hyteg_hip_stream_t                           captureStream_ = nullptr;
   Recording*                                   recording_     = nullptr;
   std::map< Key, Recording >                   recordings_;
   uint_t                                       replayed_ = 0;
};

} // namespace hyteg
