// C facade of the C++ host layer (include/hyteg_host.h).  Exceptions become return codes.
#include "hyteg_host.hpp"

#include "../../include/hyteg_host.h"

using namespace hyteg;

namespace {
thread_local std::string g_err;

struct StorageH
{
   std::shared_ptr< PrimitiveStorage > p;
};
struct FunctionH
{
   std::shared_ptr< P1Function< double > > p;
};
// form ids of hyteg_host_operator_create (include/hyteg_host.h): 0 Laplace, 1 mass, 2-4 div x/y/z, 5-7 divT x/y/z, 8 PSPG
struct OperatorH
{
   int                                          form;
   std::shared_ptr< P1ConstantLaplaceOperator > laplace;
   std::shared_ptr< P1ConstantMassOperator >    mass;
   std::shared_ptr< P1DivxOperator >            divx;
   std::shared_ptr< P1DivyOperator >            divy;
   std::shared_ptr< P1DivzOperator >            divz;
   std::shared_ptr< P1DivTxOperator >           divtx;
   std::shared_ptr< P1DivTyOperator >           divty;
   std::shared_ptr< P1DivTzOperator >           divtz;
   std::shared_ptr< P1PSPGOperator >            pspg;
   FunctionH                                    invDiag; // borrowed view
};
struct StokesFunctionH
{
   std::shared_ptr< P1StokesFunction< double > > p;
   FunctionH                                     comp[4]; // borrowed views of u, v, w, p
};
struct StokesOperatorH
{
   std::shared_ptr< P1P1StokesOperator > p;
};
struct StokesSolverH
{
   std::shared_ptr< Solver< P1P1StokesOperator > > p;
};
struct SolverH
{
   std::shared_ptr< Solver< P1ConstantLaplaceOperator > > p;
};

template < typename F >
int guarded( F&& fn )
{
   try
   {
      fn();
      return 0;
   } catch ( const std::exception& e )
   {
      g_err = e.what();
      P2PTransport::abortExchangesOfAllTransports(); // no plan stays "in flight" behind a failed operation
      return 1;
   } catch ( ... )
   {
      g_err = "unknown error";
      P2PTransport::abortExchangesOfAllTransports();
      return 1;
   }
}
PrimitiveStorage&      S( hh_storage_t s ) { return *static_cast< StorageH* >( s )->p; }
P1Function< double >&  F( hh_function_t f ) { return *static_cast< FunctionH* >( f )->p; }
std::vector< std::reference_wrapper< const P1Function< double > > > refs( int n, const hh_function_t* fs )
{
   std::vector< std::reference_wrapper< const P1Function< double > > > r;
   for ( int i = 0; i < n; ++i )
      r.push_back( std::cref( F( fs[i] ) ) );
   return r;
}
} // namespace

extern "C" {

HYTEG_HOST_API const char* hyteg_host_last_error( void ) { return g_err.c_str(); }

HYTEG_HOST_API int hyteg_host_storage_from_gmsh( const char* path, int rank, int nranks, hh_storage_t* out )
{
   return guarded( [&] { *out = new StorageH{ std::make_shared< PrimitiveStorage >( MeshInfo::fromGmshFile( path ), rank, nranks ) }; } );
}
HYTEG_HOST_API int hyteg_host_storage_from_arrays( int nv, const double* xyz, int nc, const int* cells, int rank, int nranks, hh_storage_t* out )
{
   return guarded( [&] { *out = new StorageH{ std::make_shared< PrimitiveStorage >( MeshInfo::fromArrays( nv, xyz, nc, cells ), rank, nranks ) }; } );
}
HYTEG_HOST_API int hyteg_host_storage_destroy( hh_storage_t s )
{
   return guarded( [&] { delete static_cast< StorageH* >( s ); } );
}
HYTEG_HOST_API int hyteg_host_storage_counts( hh_storage_t s, int* c )
{
   return guarded( [&] {
      auto& st = S( s );
      c[0]     = (int) st.getCells().size();
      c[1]     = (int) st.getFaces().size();
      c[2]     = (int) st.getEdges().size();
      c[3]     = (int) st.getVertices().size();
      c[4]     = (int) st.getNumberOfLocalCells();
      c[5]     = st.numRanks();
   } );
}
HYTEG_HOST_API int hyteg_host_storage_local_cell( hh_storage_t s, int li, int* gid, double* coords12, double* nnc14 )
{
   return guarded( [&] {
      const MacroCell& c = S( s ).getLocalCell( (uint_t) li );
      if ( gid )
         *gid = c.id;
      if ( coords12 )
         for ( int v = 0; v < 4; ++v )
            for ( int r = 0; r < 3; ++r )
               coords12[3 * v + r] = c.coords[v][r];
      if ( nnc14 )
      {
         auto n = S( s ).numNeighborCells( c );
         std::copy( n.begin(), n.end(), nnc14 );
      }
   } );
}
HYTEG_HOST_API int hyteg_host_storage_mask( hh_storage_t s, int li, int flag, int owned, unsigned* mask )
{
   return guarded( [&] {
      const MacroCell& c = S( s ).getLocalCell( (uint_t) li );
      *mask              = owned ? S( s ).ownedMaskFor( c, DoFType( flag ) ) : S( s ).maskFor( c, DoFType( flag ) );
   } );
}
HYTEG_HOST_API int hyteg_host_storage_set_boundary_type( hh_storage_t s, int t )
{
   return guarded( [&] { S( s ).setBoundaryType( DoFType( t ) ); } );
}
HYTEG_HOST_API int hyteg_host_storage_set_stream( hh_storage_t s, void* stream )
{
   return guarded( [&] { S( s ).setStream( stream ); } );
}
HYTEG_HOST_API int hyteg_host_storage_set_batch_max_level( hh_storage_t s, int level )
{
   return guarded( [&] { S( s ).setBatchMaxLevel( level ); } );
}
HYTEG_HOST_API int hyteg_host_storage_use_rccl( hh_storage_t s, const unsigned char* unique_id )
{
   return guarded( [&] {
      if ( !unique_id )
         throw std::runtime_error( "storage_use_rccl: null unique id" );
      S( s ).useRccl( unique_id );
   } );
}
HYTEG_HOST_API int hyteg_host_storage_use_p2p( hh_storage_t s, size_t arena_bytes, unsigned char* handle, int* arena_kind )
{
   return guarded( [&] {
      if ( !handle )
         throw std::runtime_error( "storage_use_p2p: null handle buffer" );
      P2PTransport& T = S( s ).useP2P( arena_bytes );
      std::copy( T.handle().begin(), T.handle().end(), handle );
      if ( arena_kind )
         *arena_kind = T.arenaKind();
   } );
}
HYTEG_HOST_API int hyteg_host_storage_p2p_open( hh_storage_t s, const unsigned char* handles )
{
   return guarded( [&] {
      if ( !handles )
         throw std::runtime_error( "storage_p2p_open: null handles" );
      S( s ).p2p().openPeers( handles );
   } );
}
HYTEG_HOST_API int hyteg_host_storage_p2p_layout( hh_storage_t s, int level, int key, long long* offsets )
{
   return guarded( [&] {
      const auto& plan = S( s ).devicePlan( level, key & 1, key >> 1 );
      const auto  o    = S( s ).p2p().layoutPlan( plan, level, key );
      if ( !o.empty() && !offsets )
         throw std::runtime_error( "storage_p2p_layout: null offsets" );
      std::copy( o.begin(), o.end(), offsets );
   } );
}
HYTEG_HOST_API int hyteg_host_storage_p2p_connect( hh_storage_t s, int level, int key, const long long* offsets )
{
   return guarded( [&] {
      const auto& plan = S( s ).devicePlan( level, key & 1, key >> 1 );
      if ( !plan.peers.empty() && !offsets )
         throw std::runtime_error( "storage_p2p_connect: null offsets" );
      S( s ).p2p().connectPlan( plan, level, key, offsets );
   } );
}
HYTEG_HOST_API int hyteg_host_storage_drop_p2p( hh_storage_t s )
{
   return guarded( [&] { S( s ).dropP2P(); } );
}
HYTEG_HOST_API int hyteg_host_storage_check_transport( hh_storage_t s )
{
   return guarded( [&] { S( s ).checkTransport(); } );
}
HYTEG_HOST_API int hyteg_host_storage_allreduce_sum( hh_storage_t s, double* values, int n )
{
   return guarded( [&] {
      if ( S( s ).numRanks() > 1 )
         S( s ).requireTransport( "allreduce_sum" ).allreduceSum( values, n );
   } );
}
HYTEG_HOST_API int hyteg_host_storage_enable_timing( hh_storage_t s, int on, int synchronize )
{
   return guarded( [&] { S( s ).enableTiming( on != 0, synchronize != 0 ); } );
}
HYTEG_HOST_API int hyteg_host_storage_timing_json( hh_storage_t s, char* buf, size_t buflen, size_t* needed )
{
   return guarded( [&] {
      if ( !S( s ).getTimingTree() )
         throw std::runtime_error( "storage_timing_json: timing is not enabled" );
      const std::string j = S( s ).getTimingTree()->toJSON();
      if ( needed )
         *needed = j.size() + 1;
      if ( buf && buflen > 0 )
         snprintf( buf, buflen, "%s", j.c_str() );
   } );
}
HYTEG_HOST_API int hyteg_host_storage_timing_reset( hh_storage_t s )
{
   return guarded( [&] {
      if ( S( s ).getTimingTree() )
         S( s ).getTimingTree()->reset();
   } );
}
HYTEG_HOST_API int hyteg_host_storage_transport_name( hh_storage_t s, char* buf, int buflen )
{
   return guarded( [&] {
      const char* n = S( s ).transport() ? S( s ).transport()->name() : "none";
      snprintf( buf, (size_t) buflen, "%s", n );
   } );
}
HYTEG_HOST_API int hyteg_host_storage_set_hooks( hh_storage_t s, int ( *exb )( void*, int, int ), int ( *exe )( void*, int, int ),
                                                 int ( *ar )( void*, double*, int ), void* user )
{
   return guarded( [&] {
      CommHooks h;
      h.exchangeBegin = exb;
      h.exchangeEnd   = exe;
      h.allreduceSum = ar;
      h.user         = user;
      S( s ).setCommHooks( h );
   } );
}
// key = cls + 2 * dof_kind (dof_kind 0: vertex DoFs, 1: edge DoFs of P2 functions)
HYTEG_HOST_API int hyteg_host_plan_sizes( hh_storage_t s, int level, int key, int* sizes )
{
   return guarded( [&] {
      const auto& P = S( s ).exchangePlan( level, key & 1, key >> 1 );
      sizes[0]      = P.ngroups();
      sizes[1]      = (int) P.entryBuf.size();
      sizes[2]      = (int) P.peers.size();
      sizes[3]      = P.totalSend();
      sizes[4]      = P.totalRecv();
   } );
}
HYTEG_HOST_API int hyteg_host_plan_export( hh_storage_t s, int level, int key, int* gp, int* eb, int* eo, int* peers, int* sc, int* rc, int* sb, int* so )
{
   return guarded( [&] {
      const auto& P = S( s ).exchangePlan( level, key & 1, key >> 1 );
      std::copy( P.groupPtr.begin(), P.groupPtr.end(), gp );
      std::copy( P.entryBuf.begin(), P.entryBuf.end(), eb );
      std::copy( P.entryOff.begin(), P.entryOff.end(), eo );
      std::copy( P.peers.begin(), P.peers.end(), peers );
      std::copy( P.sendCount.begin(), P.sendCount.end(), sc );
      std::copy( P.recvCount.begin(), P.recvCount.end(), rc );
      std::copy( P.sendBuf.begin(), P.sendBuf.end(), sb );
      std::copy( P.sendOff.begin(), P.sendOff.end(), so );
   } );
}
HYTEG_HOST_API int hyteg_host_plan_register_buffers( hh_storage_t s, int level, int cls, double* send, double* recv )
{
   return guarded( [&] { S( s ).registerCommBuffers( level, cls, send, recv ); } );
}

HYTEG_HOST_API int hyteg_host_function_create( hh_storage_t s, const char* name, int minL, int maxL, hh_function_t* out )
{
   return guarded( [&] {
      *out = new FunctionH{ std::make_shared< P1Function< double > >( name, static_cast< StorageH* >( s )->p, (uint_t) minL, (uint_t) maxL ) };
   } );
}
HYTEG_HOST_API int hyteg_host_function_destroy( hh_function_t f )
{
   return guarded( [&] { delete static_cast< FunctionH* >( f ); } );
}
HYTEG_HOST_API int hyteg_host_function_cell_pointer( hh_function_t f, int c, int level, double** p )
{
   return guarded( [&] { *p = F( f ).getCellPointer( (uint_t) c, (uint_t) level ); } );
}
HYTEG_HOST_API int hyteg_host_function_upload_cell( hh_function_t f, int c, int level, const double* host )
{
   return guarded( [&] { F( f ).copyCellFromHost( (uint_t) c, (uint_t) level, host ); } );
}
HYTEG_HOST_API int hyteg_host_function_download_cell( hh_function_t f, int c, int level, double* host )
{
   return guarded( [&] { F( f ).copyCellToHost( (uint_t) c, (uint_t) level, host ); } );
}
HYTEG_HOST_API int hyteg_host_function_interpolate_constant( hh_function_t f, double v, int level, int flag )
{
   return guarded( [&] { F( f ).interpolate( v, (uint_t) level, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_function_assign( hh_function_t dst, int n, const double* sc, const hh_function_t* srcs, int level, int flag )
{
   return guarded( [&] { F( dst ).assign( std::vector< double >( sc, sc + n ), refs( n, srcs ), (uint_t) level, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_function_add( hh_function_t dst, int n, const double* sc, const hh_function_t* srcs, int level, int flag )
{
   return guarded( [&] { F( dst ).add( std::vector< double >( sc, sc + n ), refs( n, srcs ), (uint_t) level, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_function_mult_elementwise( hh_function_t dst, int n, const hh_function_t* srcs, int level, int flag )
{
   return guarded( [&] { F( dst ).multElementwise( refs( n, srcs ), (uint_t) level, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_function_dot( hh_function_t a, hh_function_t b, int level, int flag, int global, double* result )
{
   return guarded( [&] {
      *result = global ? F( a ).dotGlobal( F( b ), (uint_t) level, DoFType( flag ) ) : F( a ).dotLocal( F( b ), (uint_t) level, DoFType( flag ) );
   } );
}
HYTEG_HOST_API int hyteg_host_function_sum_shared( hh_function_t f, int level, int flag )
{
   return guarded( [&] { F( f ).sumSharedCopies( (uint_t) level, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_function_set_all_inner( hh_function_t f, int on )
{
   return guarded( [&] { F( f ).setBoundaryConditionAllInner( on != 0 ); } );
}
HYTEG_HOST_API int hyteg_host_function_sync_shared( hh_function_t f, int level, int flag )
{
   return guarded( [&] { F( f ).syncSharedCopies( (uint_t) level, DoFType( flag ) ); } );
}

HYTEG_HOST_API int hyteg_host_operator_create( hh_storage_t s, int minL, int maxL, int form, hh_operator_t* out )
{
   return guarded( [&] {
      if ( form < 0 || form > 8 )
         throw std::runtime_error( "operator_create: unknown form id" );
      auto* h      = new OperatorH{};
      h->form      = form;
      auto storage = static_cast< StorageH* >( s )->p;
      const uint_t a = (uint_t) minL, b = (uint_t) maxL;
      switch ( form )
      {
      case 0: h->laplace = std::make_shared< P1ConstantLaplaceOperator >( storage, a, b ); break;
      case 1: h->mass = std::make_shared< P1ConstantMassOperator >( storage, a, b ); break;
      case 2: h->divx = std::make_shared< P1DivxOperator >( storage, a, b ); break;
      case 3: h->divy = std::make_shared< P1DivyOperator >( storage, a, b ); break;
      case 4: h->divz = std::make_shared< P1DivzOperator >( storage, a, b ); break;
      case 5: h->divtx = std::make_shared< P1DivTxOperator >( storage, a, b ); break;
      case 6: h->divty = std::make_shared< P1DivTyOperator >( storage, a, b ); break;
      case 7: h->divtz = std::make_shared< P1DivTzOperator >( storage, a, b ); break;
      default: h->pspg = std::make_shared< P1PSPGOperator >( storage, a, b ); break;
      }
      *out = h;
   } );
}
HYTEG_HOST_API int hyteg_host_operator_destroy( hh_operator_t op )
{
   return guarded( [&] { delete static_cast< OperatorH* >( op ); } );
}
#define WITH_OP_CASE( member ) \
   {                           \
      auto& A = *_h->member;   \
      expr;                    \
   }                           \
   break
#define WITH_OP( op, expr )                       \
   do                                             \
   {                                              \
      auto* _h = static_cast< OperatorH* >( op ); \
      switch ( _h->form )                         \
      {                                           \
      case 0: { auto& A = *_h->laplace; expr; } break; \
      case 1: { auto& A = *_h->mass; expr; } break;    \
      case 2: { auto& A = *_h->divx; expr; } break;    \
      case 3: { auto& A = *_h->divy; expr; } break;    \
      case 4: { auto& A = *_h->divz; expr; } break;    \
      case 5: { auto& A = *_h->divtx; expr; } break;   \
      case 6: { auto& A = *_h->divty; expr; } break;   \
      case 7: { auto& A = *_h->divtz; expr; } break;   \
      default: { auto& A = *_h->pspg; expr; } break;   \
      }                                           \
   } while ( 0 )
HYTEG_HOST_API int hyteg_host_operator_stencils( hh_operator_t op, int gc, int level, double* inner15, double* slots210 )
{
   return guarded( [&] {
      WITH_OP( op, {
         const auto& st = A.getCellStencils( gc, (uint_t) level );
         std::copy( st.inner, st.inner + 15, inner15 );
         std::copy( &st.slots[0][0], &st.slots[0][0] + 210, slots210 );
      } );
   } );
}
HYTEG_HOST_API int hyteg_host_operator_apply( hh_operator_t op, hh_function_t src, hh_function_t dst, int level, int flag, int update )
{
   return guarded( [&] { WITH_OP( op, A.apply( F( src ), F( dst ), (uint_t) level, DoFType( flag ), update ? Add : Replace ) ); } );
}
HYTEG_HOST_API int hyteg_host_operator_apply_cycle( hh_operator_t op, int npairs, const hh_function_t* srcs, const hh_function_t* dsts, int level,
                                                    int flag, int update, int first, int steps )
{
   return guarded( [&] {
      for ( int k = 0; k < steps; ++k )
      {
         const int j = ( first + k ) % npairs;
         WITH_OP( op, A.apply( F( srcs[j] ), F( dsts[j] ), (uint_t) level, DoFType( flag ), update ? Add : Replace ) );
      }
   } );
}
HYTEG_HOST_API int hyteg_host_operator_apply_cycle_timed( hh_operator_t op, int npairs, const hh_function_t* srcs, const hh_function_t* dsts,
                                                          int level, int flag, int update, int first, int steps, void* evStart, void* evStop )
{
   // the same loop between two timing events of the C-ABI, recorded on the storage's stream directly before the first and after
   // the last apply (a measurement harness's bracket without its own call overhead inside)
   return guarded( [&] {
      if ( steps <= 0 || npairs <= 0 )
         return;
      hyteg_hip_stream_t stream = F( srcs[0] ).getStorage()->stream();
      if ( evStart )
         hipCheck( hyteg_hip_event_record( evStart, stream ), "apply_cycle_timed: record" );
      for ( int k = 0; k < steps; ++k )
      {
         const int j = ( first + k ) % npairs;
         WITH_OP( op, A.apply( F( srcs[j] ), F( dsts[j] ), (uint_t) level, DoFType( flag ), update ? Add : Replace ) );
      }
      if ( evStop )
         hipCheck( hyteg_hip_event_record( evStop, stream ), "apply_cycle_timed: record" );
   } );
}
HYTEG_HOST_API int hyteg_host_operator_smooth_jac( hh_operator_t op, hh_function_t dst, hh_function_t rhs, hh_function_t src, double relax, int level, int flag )
{
   return guarded( [&] { WITH_OP( op, A.smooth_jac( F( dst ), F( rhs ), F( src ), relax, (uint_t) level, DoFType( flag ) ) ); } );
}
HYTEG_HOST_API int hyteg_host_operator_smooth_sor( hh_operator_t op, hh_function_t dst, hh_function_t rhs, double relax, int level, int flag, int backwards )
{
   return guarded( [&] { WITH_OP( op, A.smooth_sor( F( dst ), F( rhs ), relax, (uint_t) level, DoFType( flag ), backwards != 0 ) ); } );
}
HYTEG_HOST_API int hyteg_host_operator_compute_inverse_diagonal( hh_operator_t op )
{
   return guarded( [&] { WITH_OP( op, A.computeInverseDiagonalOperatorValues() ); } );
}
HYTEG_HOST_API int hyteg_host_operator_inverse_diagonal( hh_operator_t op, hh_function_t* out )
{
   return guarded( [&] {
      auto* h = static_cast< OperatorH* >( op );
      WITH_OP( op, h->invDiag.p = A.getInverseDiagonalValues() );
      *out = &h->invDiag;
   } );
}

// ---- the generated elementwise operator class (module hyteg_operators): P1ElementwiseDiffusion ----
struct ElementwiseH
{
   std::shared_ptr< operatorgeneration::P1ElementwiseDiffusion > p;
   FunctionH                                                     invDiag; // borrowed view
};
HYTEG_HOST_API int hyteg_host_elementwise_create( hh_storage_t s, int minL, int maxL, hh_elementwise_t* out )
{
   return guarded( [&] {
      auto* h = new ElementwiseH{};
      h->p    = std::make_shared< operatorgeneration::P1ElementwiseDiffusion >( static_cast< StorageH* >( s )->p, (uint_t) minL, (uint_t) maxL );
      *out    = h;
   } );
}
HYTEG_HOST_API int hyteg_host_elementwise_destroy( hh_elementwise_t op )
{
   return guarded( [&] { delete static_cast< ElementwiseH* >( op ); } );
}
HYTEG_HOST_API int hyteg_host_elementwise_apply( hh_elementwise_t op, hh_function_t src, hh_function_t dst, int level, int flag, int update )
{
   return guarded( [&] { static_cast< ElementwiseH* >( op )->p->apply( F( src ), F( dst ), (uint_t) level, DoFType( flag ), update ? Add : Replace ); } );
}
HYTEG_HOST_API int hyteg_host_elementwise_compute_inverse_diagonal( hh_elementwise_t op )
{
   return guarded( [&] { static_cast< ElementwiseH* >( op )->p->computeInverseDiagonalOperatorValues(); } );
}
HYTEG_HOST_API int hyteg_host_elementwise_inverse_diagonal( hh_elementwise_t op, hh_function_t* out )
{
   return guarded( [&] {
      auto* h      = static_cast< ElementwiseH* >( op );
      h->invDiag.p = h->p->getInverseDiagonalValues();
      *out         = &h->invDiag;
   } );
}
HYTEG_HOST_API int hyteg_host_elementwise_smooth_jac( hh_elementwise_t op, hh_function_t dst, hh_function_t rhs, hh_function_t src, double relax,
                                                      int level, int flag )
{
   return guarded( [&] { static_cast< ElementwiseH* >( op )->p->smooth_jac( F( dst ), F( rhs ), F( src ), relax, (uint_t) level, DoFType( flag ) ); } );
}

HYTEG_HOST_API int hyteg_host_restrict( hh_function_t f, int sourceLevel, int flag )
{
   return guarded( [&] { P1toP1LinearRestriction().restrict( F( f ), (uint_t) sourceLevel, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_prolongate( hh_function_t f, int sourceLevel, int flag )
{
   return guarded( [&] { P1toP1LinearProlongation().prolongate( F( f ), (uint_t) sourceLevel, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_prolongate_and_add( hh_function_t f, int sourceLevel, int flag )
{
   return guarded( [&] { P1toP1LinearProlongation().prolongateAndAdd( F( f ), (uint_t) sourceLevel, DoFType( flag ) ); } );
}

HYTEG_HOST_API int hyteg_host_gmg_create( hh_storage_t s, int minL, int maxL, int smoother, double relax, int pre, int post, int wcycle,
                                          int cgMaxIter, double cgTol, hh_solver_t* out )
{
   return guarded( [&] {
      using Op     = P1ConstantLaplaceOperator;
      auto storage = static_cast< StorageH* >( s )->p;
      std::shared_ptr< Solver< Op > > sm;
      if ( smoother == 0 )
         sm = std::make_shared< WeightedJacobiSmoother< Op > >( storage, (uint_t) minL, (uint_t) maxL, relax );
      else if ( smoother == 1 )
         sm = std::make_shared< GaussSeidelSmoother< Op > >();
      else if ( smoother == 3 )
         sm = std::make_shared< MixedPrecisionJacobiSmoother< Op > >( storage, (uint_t) minL, (uint_t) maxL, relax );
      else
         sm = std::make_shared< SORSmoother< Op > >( relax );
      auto coarse = std::make_shared< CGSolver< Op > >( storage, (uint_t) minL, (uint_t) minL, (uint_t) cgMaxIter, cgTol, cgTol );
      *out        = new SolverH{ std::make_shared< GeometricMultigridSolver< Op > >(
          storage, sm, coarse, std::make_shared< P1toP1LinearRestriction >(), std::make_shared< P1toP1LinearProlongation >(), (uint_t) minL,
          (uint_t) maxL, (uint_t) pre, (uint_t) post, 0, wcycle ? CycleType::WCYCLE : CycleType::VCYCLE ) };
   } );
}
HYTEG_HOST_API int hyteg_host_gmg_set_use_graphs( hh_solver_t solver, int on )
{
   return guarded( [&] {
      auto g = std::dynamic_pointer_cast< GeometricMultigridSolver< P1ConstantLaplaceOperator > >( static_cast< SolverH* >( solver )->p );
      if ( !g )
         throw std::runtime_error( "gmg_set_use_graphs: not a geometric multigrid solver" );
      g->setUseGraphs( on != 0 );
   } );
}
HYTEG_HOST_API int hyteg_host_cg_set_use_device_scalars( hh_solver_t solver, int on )
{
   return guarded( [&] {
      using Op = P1ConstantLaplaceOperator;
      auto sp  = static_cast< SolverH* >( solver )->p;
      if ( auto g = std::dynamic_pointer_cast< GeometricMultigridSolver< Op > >( sp ) )
         sp = g->getCoarseSolver();
      auto cg = std::dynamic_pointer_cast< CGSolver< Op > >( sp );
      if ( !cg )
         throw std::runtime_error( "cg_set_use_device_scalars: neither a CG solver nor a multigrid solver with a CG coarse solver" );
      cg->setUseDeviceScalars( ( on & 1 ) != 0 );
      cg->setUseSingleLaunch( ( on & 2 ) != 0 );
   } );
}
HYTEG_HOST_API int hyteg_host_cg_iterations( hh_solver_t solver, int* iterations )
{
   return guarded( [&] {
      auto cg = std::dynamic_pointer_cast< CGSolver< P1ConstantLaplaceOperator > >( static_cast< SolverH* >( solver )->p );
      if ( !cg )
         throw std::runtime_error( "cg_iterations: not a CG solver" );
      *iterations = (int) cg->getIterations();
   } );
}
HYTEG_HOST_API int hyteg_host_gmg_replayed_cycles( hh_solver_t solver, int* count )
{
   return guarded( [&] {
      auto g = std::dynamic_pointer_cast< GeometricMultigridSolver< P1ConstantLaplaceOperator > >( static_cast< SolverH* >( solver )->p );
      if ( !g )
         throw std::runtime_error( "gmg_replayed_cycles: not a geometric multigrid solver" );
      *count = (int) g->replayedCycles();
   } );
}
HYTEG_HOST_API int hyteg_host_cg_create( hh_storage_t s, int minL, int maxL, int maxIter, double tol, hh_solver_t* out )
{
   return guarded( [&] {
      *out = new SolverH{ std::make_shared< CGSolver< P1ConstantLaplaceOperator > >( static_cast< StorageH* >( s )->p, (uint_t) minL, (uint_t) maxL,
                                                                                     (uint_t) maxIter, tol, tol ) };
   } );
}
HYTEG_HOST_API int hyteg_host_solver_solve( hh_solver_t solver, hh_operator_t laplace, hh_function_t x, hh_function_t b, int level )
{
   return guarded( [&] {
      auto* h = static_cast< OperatorH* >( laplace );
      if ( h->form != 0 )
         throw std::runtime_error( "solver_solve: the solvers are instantiated for the Laplace operator" );
      static_cast< SolverH* >( solver )->p->solve( *h->laplace, F( x ), F( b ), (uint_t) level );
   } );
}
HYTEG_HOST_API int hyteg_host_solver_destroy( hh_solver_t solver )
{
   return guarded( [&] { delete static_cast< SolverH* >( solver ); } );
}

/* ---- P1-P1 Stokes (src/mixed_operator/P1P1StokesOperator.hpp, src/hyteg/solvers/UzawaSmoother.hpp) ---- */
HYTEG_HOST_API int hyteg_host_stokes_function_create( hh_storage_t s, const char* name, int minL, int maxL, hh_stokes_function_t* out )
{
   return guarded( [&] {
      auto* h = new StokesFunctionH{};
      h->p    = std::make_shared< P1StokesFunction< double > >( name, static_cast< StorageH* >( s )->p, (uint_t) minL, (uint_t) maxL );
      *out    = h;
   } );
}
HYTEG_HOST_API int hyteg_host_stokes_function_destroy( hh_stokes_function_t f )
{
   return guarded( [&] { delete static_cast< StokesFunctionH* >( f ); } );
}
HYTEG_HOST_API int hyteg_host_stokes_function_component( hh_stokes_function_t f, int k, hh_function_t* out )
{
   return guarded( [&] {
      auto* h = static_cast< StokesFunctionH* >( f );
      if ( k < 0 || k > 3 )
         throw std::runtime_error( "stokes_function_component: k = 0, 1, 2 (velocity) or 3 (pressure)" );
      // aliasing pointer: shares ownership with the Stokes function, points at the component
      const P1Function< double >& c = k < 3 ? h->p->uvw()[(uint_t) k] : h->p->p();
      h->comp[k].p                  = std::shared_ptr< P1Function< double > >( h->p, const_cast< P1Function< double >* >( &c ) );
      *out                          = &h->comp[k];
   } );
}
HYTEG_HOST_API int hyteg_host_stokes_function_assign( hh_stokes_function_t dst, int n, const double* scalars, const hh_stokes_function_t* fs, int level, int flag )
{
   return guarded( [&] {
      std::vector< std::reference_wrapper< const P1StokesFunction< double > > > r;
      for ( int i = 0; i < n; ++i )
         r.push_back( std::cref( *static_cast< StokesFunctionH* >( fs[i] )->p ) );
      static_cast< StokesFunctionH* >( dst )->p->assign( std::vector< double >( scalars, scalars + n ), r, (uint_t) level, DoFType( flag ) );
   } );
}
HYTEG_HOST_API int hyteg_host_stokes_function_dot( hh_stokes_function_t a, hh_stokes_function_t b, int level, int flag, double* out )
{
   return guarded( [&] {
      *out = static_cast< StokesFunctionH* >( a )->p->dotGlobal( *static_cast< StokesFunctionH* >( b )->p, (uint_t) level, DoFType( flag ) );
   } );
}
HYTEG_HOST_API int hyteg_host_project_mean( hh_function_t pressure, int level )
{
   return guarded( [&] { projectMean( F( pressure ), (uint_t) level ); } );
}
HYTEG_HOST_API int hyteg_host_stokes_operator_create( hh_storage_t s, int minL, int maxL, hh_stokes_operator_t* out )
{
   return guarded( [&] {
      *out = new StokesOperatorH{ std::make_shared< P1P1StokesOperator >( static_cast< StorageH* >( s )->p, (uint_t) minL, (uint_t) maxL ) };
   } );
}
HYTEG_HOST_API int hyteg_host_stokes_operator_destroy( hh_stokes_operator_t op )
{
   return guarded( [&] { delete static_cast< StokesOperatorH* >( op ); } );
}
HYTEG_HOST_API int hyteg_host_stokes_operator_apply( hh_stokes_operator_t op, hh_stokes_function_t src, hh_stokes_function_t dst, int level, int flag )
{
   return guarded( [&] {
      static_cast< StokesOperatorH* >( op )->p->apply( *static_cast< StokesFunctionH* >( src )->p, *static_cast< StokesFunctionH* >( dst )->p,
                                                       (uint_t) level, DoFType( flag ) );
   } );
}
HYTEG_HOST_API int hyteg_host_stokes_uzawa_create( hh_storage_t s, int minL, int maxL, double relax, int velocity_iterations, int velocity_smoother,
                                                   double velocity_relax, hh_stokes_solver_t* out )
{
   return guarded( [&] {
      using Op     = P1P1StokesOperator;
      auto storage = static_cast< StorageH* >( s )->p;
      std::shared_ptr< Solver< P1ConstantLaplaceOperator > > scalar;
      if ( velocity_smoother == 0 )
         scalar = std::make_shared< WeightedJacobiSmoother< P1ConstantLaplaceOperator > >( storage, (uint_t) minL, (uint_t) maxL, velocity_relax );
      else if ( velocity_smoother == 1 )
         scalar = std::make_shared< GaussSeidelSmoother< P1ConstantLaplaceOperator > >();
      else if ( velocity_smoother == 2 )
         scalar = std::make_shared< SORSmoother< P1ConstantLaplaceOperator > >( velocity_relax );
      else if ( velocity_smoother == 3 ) // BASELINE config 5's "fp32 smoother": float Jacobi sweeps inside the Uzawa smoother
         scalar = std::make_shared< MixedPrecisionJacobiSmoother< P1ConstantLaplaceOperator > >( storage, (uint_t) minL, (uint_t) maxL, velocity_relax );
      else
         throw std::runtime_error( "stokes_uzawa_create: unknown velocity smoother" );
      auto velocity = std::make_shared< StokesVelocityBlockBlockDiagonalPreconditioner< Op > >( storage, scalar );
      *out          = new StokesSolverH{ std::make_shared< UzawaSmoother< Op > >( storage, velocity, (uint_t) minL, (uint_t) maxL, relax,
                                                                                  Inner | NeumannBoundary | FreeslipBoundary,
                                                                                  (uint_t) velocity_iterations ) };
   } );
}
// coarse: 0 = dense LU on the host (stand-in for PETScLUSolver, single rank), 1 = MinResSolver preconditioned with
// StokesPressureBlockPreconditioner< P1P1StokesOperator, P1LumpedInvMassOperator > as apps/stokesSphere/StokesSphere.cpp:227-237
// composes it (any number of ranks)
HYTEG_HOST_API int hyteg_host_stokes_gmg_create_with_coarse( hh_storage_t s, hh_stokes_solver_t smoother, int minL, int maxL, int pre, int post,
                                                             int increment, int project_mean_after_restriction, int coarse, int coarse_max_iter,
                                                             double coarse_rel_tol, hh_stokes_solver_t* out )
{
   return guarded( [&] {
      using Op     = P1P1StokesOperator;
      auto storage = static_cast< StorageH* >( s )->p;
      std::shared_ptr< Solver< Op > > coarseSolver;
      if ( coarse == 0 )
         coarseSolver = std::make_shared< DenseCoarseGridSolver< Op > >( storage, (uint_t) minL );
      else if ( coarse == 1 )
      {
         auto prec    = std::make_shared< StokesPressureBlockPreconditioner< Op, P1LumpedInvMassOperator > >( storage, (uint_t) minL, (uint_t) minL );
         coarseSolver = std::make_shared< MinResSolver< Op > >( storage, (uint_t) minL, (uint_t) minL, (uint_t) coarse_max_iter, coarse_rel_tol, 1e-16, prec );
      }
      else
         throw std::runtime_error( "stokes_gmg_create: unknown coarse-grid solver" );
      *out = new StokesSolverH{
          std::make_shared< GeometricMultigridSolver< Op, P1P1StokesToP1P1StokesRestriction, P1P1StokesToP1P1StokesProlongation > >(
              storage, static_cast< StokesSolverH* >( smoother )->p, coarseSolver,
              std::make_shared< P1P1StokesToP1P1StokesRestriction >( project_mean_after_restriction != 0 ),
              std::make_shared< P1P1StokesToP1P1StokesProlongation >(), (uint_t) minL, (uint_t) maxL, (uint_t) pre, (uint_t) post,
              (uint_t) increment ) };
   } );
}
HYTEG_HOST_API int hyteg_host_stokes_gmg_create( hh_storage_t s, hh_stokes_solver_t smoother, int minL, int maxL, int pre, int post, int increment,
                                                 int project_mean_after_restriction, hh_stokes_solver_t* out )
{
   return hyteg_host_stokes_gmg_create_with_coarse( s, smoother, minL, maxL, pre, post, increment, project_mean_after_restriction, 0, 0, 0.0, out );
}
// MinResSolver< P1P1StokesOperator > on its own: preconditioner 0 identity, 1 pressure block (lumped inverse mass),
// 2 StokesBlockDiagonalPreconditioner with `velocity_steps` V(2,2) cycles of the scalar Laplace operator per velocity component
// (apps/stokesSphere/StokesSphere.cpp:198-222, tests/hyteg/convergence/P1Stokes3DMinResConvergenceTest.cpp:166-183)
HYTEG_HOST_API int hyteg_host_stokes_minres_create( hh_storage_t s, int minL, int maxL, int max_iter, double rel_tol, int preconditioner,
                                                    int velocity_steps, hh_stokes_solver_t* out )
{
   return guarded( [&] {
      using Op     = P1P1StokesOperator;
      auto storage = static_cast< StorageH* >( s )->p;
      std::shared_ptr< Solver< Op > > prec;
      if ( preconditioner == 0 )
         prec = std::make_shared< IdentityPreconditioner< Op > >();
      else if ( preconditioner == 1 )
         prec = std::make_shared< StokesPressureBlockPreconditioner< Op, P1LumpedInvMassOperator > >( storage, (uint_t) minL, (uint_t) maxL );
      else if ( preconditioner == 2 )
      {
         using L       = P1ConstantLaplaceOperator;
         auto smoother = std::make_shared< GaussSeidelSmoother< L > >();
         auto coarse   = std::make_shared< CGSolver< L > >( storage, (uint_t) minL, (uint_t) maxL );
         auto gmg      = std::make_shared< GeometricMultigridSolver< L > >( storage, smoother, coarse, std::make_shared< P1toP1LinearRestriction >(),
                                                                       std::make_shared< P1toP1LinearProlongation >(), (uint_t) minL, (uint_t) maxL, 2, 2 );
         prec = std::make_shared< StokesBlockDiagonalPreconditioner< Op, P1LumpedInvMassOperator > >( storage, (uint_t) minL, (uint_t) maxL,
                                                                                                       (uint_t) velocity_steps, gmg );
      }
      else
         throw std::runtime_error( "stokes_minres_create: unknown preconditioner" );
      auto solver = std::make_shared< MinResSolver< Op > >( storage, (uint_t) minL, (uint_t) maxL, (uint_t) max_iter, rel_tol, 1e-16, prec );
      *out        = new StokesSolverH{ solver };
   } );
}
HYTEG_HOST_API int hyteg_host_stokes_minres_iterations( hh_stokes_solver_t solver, int* iterations )
{
   return guarded( [&] {
      auto* m = dynamic_cast< MinResSolver< P1P1StokesOperator >* >( static_cast< StokesSolverH* >( solver )->p.get() );
      if ( !m )
         throw std::runtime_error( "stokes_minres_iterations: not a MinResSolver" );
      *iterations = (int) m->getIterations();
   } );
}
// MinResSolver< P1ConstantLaplaceOperator > with JacobiPreconditioner( jacobi_iterations ) (0: identity), as
// tests/hyteg/convergence/P1MinResConvergenceTest.cpp:68-72
HYTEG_HOST_API int hyteg_host_solver_create_minres( hh_storage_t s, int minL, int maxL, int max_iter, double rel_tol, int jacobi_iterations, hh_solver_t* out )
{
   return guarded( [&] {
      using L      = P1ConstantLaplaceOperator;
      auto storage = static_cast< StorageH* >( s )->p;
      std::shared_ptr< Solver< L > > prec;
      if ( jacobi_iterations > 0 )
         prec = std::make_shared< JacobiPreconditioner< L > >( storage, (uint_t) minL, (uint_t) maxL, (uint_t) jacobi_iterations );
      else
         prec = std::make_shared< IdentityPreconditioner< L > >();
      *out = new SolverH{ std::make_shared< MinResSolver< L > >( storage, (uint_t) minL, (uint_t) maxL, (uint_t) max_iter, rel_tol, 1e-16, prec ) };
   } );
}
HYTEG_HOST_API int hyteg_host_stokes_solver_solve( hh_stokes_solver_t solver, hh_stokes_operator_t op, hh_stokes_function_t x, hh_stokes_function_t b, int level )
{
   return guarded( [&] {
      static_cast< StokesSolverH* >( solver )->p->solve( *static_cast< StokesOperatorH* >( op )->p, *static_cast< StokesFunctionH* >( x )->p,
                                                         *static_cast< StokesFunctionH* >( b )->p, (uint_t) level );
   } );
}
HYTEG_HOST_API int hyteg_host_stokes_solver_destroy( hh_stokes_solver_t solver )
{
   return guarded( [&] { delete static_cast< StokesSolverH* >( solver ); } );
}

/* ---- P2 (single macro-cell) ---- */
namespace {
struct P2FunctionH
{
   std::shared_ptr< P2Function< double > > p;
};
struct P2OperatorH
{
   std::shared_ptr< P2ElementwiseLaplaceOperator > p;
};
P2Function< double >& F2( hh_p2function_t f ) { return *static_cast< P2FunctionH* >( f )->p; }
std::vector< std::reference_wrapper< const P2Function< double > > > refs2( int n, const hh_p2function_t* fs )
{
   std::vector< std::reference_wrapper< const P2Function< double > > > r;
   for ( int i = 0; i < n; ++i )
      r.push_back( std::cref( F2( fs[i] ) ) );
   return r;
}
} // namespace

HYTEG_HOST_API int hyteg_host_p2function_create( hh_storage_t s, const char* name, int minL, int maxL, hh_p2function_t* out )
{
   return guarded( [&] {
      *out = new P2FunctionH{ std::make_shared< P2Function< double > >( name, static_cast< StorageH* >( s )->p, (uint_t) minL, (uint_t) maxL ) };
   } );
}
HYTEG_HOST_API int hyteg_host_p2function_destroy( hh_p2function_t f )
{
   return guarded( [&] { delete static_cast< P2FunctionH* >( f ); } );
}
HYTEG_HOST_API int hyteg_host_p2function_pointers( hh_p2function_t f, int local_cell, int level, double** vertex_dev, double** edge_dev )
{
   return guarded( [&] {
      *vertex_dev = F2( f ).getVertexDoFFunction().getCellPointer( (uint_t) local_cell, (uint_t) level );
      *edge_dev   = F2( f ).getEdgeCellPointer( (uint_t) local_cell, (uint_t) level );
   } );
}
HYTEG_HOST_API int hyteg_host_p2function_upload( hh_p2function_t f, int local_cell, int level, const double* vertex_host, const double* edge_host )
{
   return guarded( [&] {
      F2( f ).getVertexDoFFunction().copyCellFromHost( (uint_t) local_cell, (uint_t) level, vertex_host );
      F2( f ).copyEdgeFromHost( (uint_t) local_cell, (uint_t) level, edge_host );
   } );
}
HYTEG_HOST_API int hyteg_host_p2function_download( hh_p2function_t f, int local_cell, int level, double* vertex_host, double* edge_host )
{
   return guarded( [&] {
      F2( f ).getVertexDoFFunction().copyCellToHost( (uint_t) local_cell, (uint_t) level, vertex_host );
      F2( f ).copyEdgeToHost( (uint_t) local_cell, (uint_t) level, edge_host );
   } );
}
HYTEG_HOST_API int hyteg_host_p2function_interpolate_constant( hh_p2function_t f, double value, int level, int flag )
{
   return guarded( [&] { F2( f ).interpolate( value, (uint_t) level, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_p2function_assign( hh_p2function_t dst, int n, const double* scalars, const hh_p2function_t* srcs, int level, int flag )
{
   return guarded( [&] { F2( dst ).assign( std::vector< double >( scalars, scalars + n ), refs2( n, srcs ), (uint_t) level, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_p2function_add( hh_p2function_t dst, int n, const double* scalars, const hh_p2function_t* srcs, int level, int flag )
{
   return guarded( [&] { F2( dst ).add( std::vector< double >( scalars, scalars + n ), refs2( n, srcs ), (uint_t) level, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_p2function_dot( hh_p2function_t a, hh_p2function_t b, int level, int flag, double* result )
{
   return guarded( [&] { *result = F2( a ).dotGlobal( F2( b ), (uint_t) level, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_p2operator_create( hh_storage_t s, int minL, int maxL, hh_p2operator_t* out )
{
   return guarded( [&] {
      *out = new P2OperatorH{ std::make_shared< P2ElementwiseLaplaceOperator >( static_cast< StorageH* >( s )->p, (uint_t) minL, (uint_t) maxL ) };
   } );
}
HYTEG_HOST_API int hyteg_host_p2_prolongate( hh_p2function_t f, int sourceLevel, int flag, int add )
{
   return guarded( [&] {
      auto& fn = *static_cast< P2FunctionH* >( f )->p;
      if ( add )
         P2toP2QuadraticProlongation().prolongateAndAdd( fn, (uint_t) sourceLevel, DoFType( flag ) );
      else
         P2toP2QuadraticProlongation().prolongate( fn, (uint_t) sourceLevel, DoFType( flag ) );
   } );
}
HYTEG_HOST_API int hyteg_host_p2_restrict( hh_p2function_t f, int sourceLevel, int flag )
{
   return guarded( [&] { P2toP2QuadraticRestriction().restrict( *static_cast< P2FunctionH* >( f )->p, (uint_t) sourceLevel, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_p2operator_create_constant( hh_storage_t s, int minL, int maxL, hh_p2operator_t* out )
{
   return guarded( [&] {
      *out = new P2OperatorH{ std::make_shared< P2ConstantLaplaceOperator >( static_cast< StorageH* >( s )->p, (uint_t) minL, (uint_t) maxL ) };
   } );
}
HYTEG_HOST_API int hyteg_host_p2operator_constant_stencils( hh_p2operator_t op, int local_cell, int level, double* out, int capacity, int* count )
{
   return guarded( [&] {
      auto* c = dynamic_cast< P2ConstantLaplaceOperator* >( static_cast< P2OperatorH* >( op )->p.get() );
      if ( !c )
         throw std::runtime_error( "p2operator_constant_stencils: not a P2ConstantLaplaceOperator" );
      const auto& v = c->getInnerStencils( (uint_t) level, (uint_t) local_cell );
      *count        = (int) v.size();
      if ( capacity < (int) v.size() )
         throw std::runtime_error( "p2operator_constant_stencils: buffer too small" );
      std::copy( v.begin(), v.end(), out );
   } );
}
HYTEG_HOST_API int hyteg_host_p2operator_destroy( hh_p2operator_t op )
{
   return guarded( [&] { delete static_cast< P2OperatorH* >( op ); } );
}
HYTEG_HOST_API int hyteg_host_p2operator_element_matrices( hh_p2operator_t op, int local_cell, int level, double* out600 )
{
   return guarded( [&] {
      const auto& h = static_cast< P2OperatorH* >( op )->p->getElementMatrices( (uint_t) level, (uint_t) local_cell );
      std::copy( h.begin(), h.end(), out600 );
   } );
}
HYTEG_HOST_API int hyteg_host_p2operator_apply( hh_p2operator_t op, hh_p2function_t src, hh_p2function_t dst, int level, int flag, int update )
{
   return guarded( [&] { static_cast< P2OperatorH* >( op )->p->apply( F2( src ), F2( dst ), (uint_t) level, DoFType( flag ), UpdateType( update ) ); } );
}
/* P2 Jacobi smoother and geometric multigrid (P2ElementwiseOperator::smooth_jac, WeightedJacobiSmoother,
 * GeometricMultigridSolver with P2toP2Quadratic{Restriction,Prolongation}, CG on the coarsest level) */
HYTEG_HOST_API int hyteg_host_p2operator_compute_inverse_diagonal( hh_p2operator_t op )
{
   return guarded( [&] { static_cast< P2OperatorH* >( op )->p->computeInverseDiagonalOperatorValues(); } );
}
HYTEG_HOST_API int hyteg_host_p2operator_inverse_diagonal_copy( hh_p2operator_t op, hh_p2function_t dst, int level )
{
   return guarded( [&] { F2( dst ).assign( { 1.0 }, { *static_cast< P2OperatorH* >( op )->p->getInverseDiagonalValues() }, (uint_t) level, All ); } );
}
HYTEG_HOST_API int hyteg_host_p2operator_smooth_jac( hh_p2operator_t op, hh_p2function_t dst, hh_p2function_t rhs, hh_p2function_t src, double relax,
                                                     int level, int flag )
{
   return guarded( [&] { static_cast< P2OperatorH* >( op )->p->smooth_jac( F2( dst ), F2( rhs ), F2( src ), relax, (uint_t) level, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_p2operator_smooth_sor( hh_p2operator_t op, hh_p2function_t dst, hh_p2function_t rhs, double relax, int level, int flag,
                                                     int backwards )
{
   return guarded( [&] {
      static_cast< P2OperatorH* >( op )->p->smooth_sor( F2( dst ), F2( rhs ), relax, (uint_t) level, DoFType( flag ), backwards != 0 );
   } );
}
struct P2SolverH
{
   std::shared_ptr< Solver< P2ElementwiseLaplaceOperator > > p;
};
HYTEG_HOST_API int hyteg_host_p2_gmg_create( hh_storage_t s, int minL, int maxL, int smootherKind, double relax, int pre, int post, int wcycle,
                                             int cgMaxIter, double cgTol, hh_p2solver_t* out )
{
   return guarded( [&] {
      using Op     = P2ElementwiseLaplaceOperator;
      auto storage = static_cast< StorageH* >( s )->p;
      std::shared_ptr< Solver< Op > > smoother;
      if ( smootherKind == 0 )
         smoother = std::make_shared< WeightedJacobiSmoother< Op > >( storage, (uint_t) minL, (uint_t) maxL, relax );
      else if ( smootherKind == 1 )
         smoother = std::make_shared< GaussSeidelSmoother< Op > >();
      else
         smoother = std::make_shared< SORSmoother< Op > >( relax );
      auto coarse   = std::make_shared< CGSolver< Op > >( storage, (uint_t) minL, (uint_t) minL, (uint_t) cgMaxIter, cgTol, cgTol );
      *out          = new P2SolverH{ std::make_shared< GeometricMultigridSolver< Op, P2toP2QuadraticRestriction, P2toP2QuadraticProlongation > >(
          storage, smoother, coarse, std::make_shared< P2toP2QuadraticRestriction >(), std::make_shared< P2toP2QuadraticProlongation >(),
          (uint_t) minL, (uint_t) maxL, (uint_t) pre, (uint_t) post, 0, wcycle ? CycleType::WCYCLE : CycleType::VCYCLE ) };
   } );
}
HYTEG_HOST_API int hyteg_host_p2_solver_solve( hh_p2solver_t solver, hh_p2operator_t op, hh_p2function_t x, hh_p2function_t b, int level )
{
   return guarded( [&] { static_cast< P2SolverH* >( solver )->p->solve( *static_cast< P2OperatorH* >( op )->p, F2( x ), F2( b ), (uint_t) level ); } );
}
HYTEG_HOST_API int hyteg_host_p2_solver_destroy( hh_p2solver_t solver )
{
   return guarded( [&] { delete static_cast< P2SolverH* >( solver ); } );
}
HYTEG_HOST_API int hyteg_host_p2_cg_solve( hh_storage_t s, hh_p2operator_t op, hh_p2function_t x, hh_p2function_t b, int level, int maxIter,
                                           double tol, int* iterations )
{
   return guarded( [&] {
      CGSolver< P2ElementwiseLaplaceOperator > cg( static_cast< StorageH* >( s )->p, (uint_t) level, (uint_t) level, (uint_t) maxIter, tol, tol );
      cg.solve( *static_cast< P2OperatorH* >( op )->p, F2( x ), F2( b ), (uint_t) level );
      if ( iterations )
         *iterations = (int) cg.getIterations();
   } );
}
}


/* ---- P2-P1 Taylor-Hood Stokes (taylorhood.hpp) ---- */
namespace {
struct THFunctionH
{
   std::shared_ptr< P2P1TaylorHoodFunction< double > > p;
   P2FunctionH                                         comp[3]; // aliasing views of the velocity components
   FunctionH                                           pressure;
};
struct THOperatorH
{
   std::shared_ptr< P2P1TaylorHoodStokesOperator > p;
};
struct THSolverH
{
   std::shared_ptr< Solver< P2P1TaylorHoodStokesOperator > > p;
};
P2P1TaylorHoodFunction< double >& FT( hh_th_function_t f ) { return *static_cast< THFunctionH* >( f )->p; }
} // namespace

HYTEG_HOST_API int hyteg_host_th_function_create( hh_storage_t s, const char* name, int minL, int maxL, hh_th_function_t* out )
{
   return guarded( [&] {
      auto* h = new THFunctionH{};
      h->p    = std::make_shared< P2P1TaylorHoodFunction< double > >( name, static_cast< StorageH* >( s )->p, (uint_t) minL, (uint_t) maxL );
      *out    = h;
   } );
}
HYTEG_HOST_API int hyteg_host_th_function_destroy( hh_th_function_t f ) { return guarded( [&] { delete static_cast< THFunctionH* >( f ); } ); }
HYTEG_HOST_API int hyteg_host_th_function_velocity( hh_th_function_t f, int k, hh_p2function_t* out )
{
   return guarded( [&] {
      auto* h = static_cast< THFunctionH* >( f );
      if ( k < 0 || k > 2 )
         throw std::runtime_error( "th_function_velocity: k = 0, 1, 2" );
      const P2Function< double >& c = h->p->uvw()[(uint_t) k];
      h->comp[k].p                  = std::shared_ptr< P2Function< double > >( h->p, const_cast< P2Function< double >* >( &c ) );
      *out                          = &h->comp[k];
   } );
}
HYTEG_HOST_API int hyteg_host_th_function_pressure( hh_th_function_t f, hh_function_t* out )
{
   return guarded( [&] {
      auto* h       = static_cast< THFunctionH* >( f );
      h->pressure.p = std::shared_ptr< P1Function< double > >( h->p, const_cast< P1Function< double >* >( &h->p->p() ) );
      *out          = &h->pressure;
   } );
}
HYTEG_HOST_API int hyteg_host_th_function_assign( hh_th_function_t dst, int n, const double* scalars, const hh_th_function_t* fs, int level, int flag )
{
   return guarded( [&] {
      std::vector< std::reference_wrapper< const P2P1TaylorHoodFunction< double > > > r;
      for ( int i = 0; i < n; ++i )
         r.push_back( std::cref( FT( fs[i] ) ) );
      FT( dst ).assign( std::vector< double >( scalars, scalars + n ), r, (uint_t) level, DoFType( flag ) );
   } );
}
HYTEG_HOST_API int hyteg_host_th_function_interpolate_constant( hh_th_function_t f, double value, int level, int flag )
{
   return guarded( [&] { FT( f ).interpolate( value, (uint_t) level, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_th_function_dot( hh_th_function_t a, hh_th_function_t b, int level, int flag, double* out )
{
   return guarded( [&] { *out = FT( a ).dotGlobal( FT( b ), (uint_t) level, DoFType( flag ) ); } );
}
HYTEG_HOST_API int hyteg_host_th_operator_create( hh_storage_t s, int minL, int maxL, hh_th_operator_t* out )
{
   return guarded( [&] {
      *out = new THOperatorH{ std::make_shared< P2P1TaylorHoodStokesOperator >( static_cast< StorageH* >( s )->p, (uint_t) minL, (uint_t) maxL ) };
   } );
}
HYTEG_HOST_API int hyteg_host_th_operator_destroy( hh_th_operator_t op ) { return guarded( [&] { delete static_cast< THOperatorH* >( op ); } ); }
HYTEG_HOST_API int hyteg_host_th_operator_apply( hh_th_operator_t op, hh_th_function_t src, hh_th_function_t dst, int level, int flag )
{
   return guarded( [&] { static_cast< THOperatorH* >( op )->p->apply( FT( src ), FT( dst ), (uint_t) level, DoFType( flag ) ); } );
}
// single blocks, for the parity tests: which = 0 div (velocity of src -> pressure of dst), 1 divT (pressure of src -> velocity of dst)
HYTEG_HOST_API int hyteg_host_th_operator_apply_block( hh_th_operator_t op, int which, hh_th_function_t src, hh_th_function_t dst, int level, int flag )
{
   return guarded( [&] {
      auto& A = *static_cast< THOperatorH* >( op )->p;
      if ( which == 0 )
         A.div.apply( FT( src ).uvw(), FT( dst ).p(), (uint_t) level, DoFType( flag ), Replace );
      else if ( which == 1 )
         A.divT.apply( FT( src ).p(), FT( dst ).uvw(), (uint_t) level, DoFType( flag ), Replace );
      else
         throw std::runtime_error( "th_operator_apply_block: which = 0 (div) or 1 (divT)" );
   } );
}
// GeometricMultigridSolver< P2P1TaylorHoodStokesOperator > as tests/hyteg/convergence/P2P1Stokes3DUzawaConvergenceTest.cpp:150-163 composes it:
// UzawaSmoother( relax ) over StokesVelocityBlockBlockDiagonalPreconditioner( GaussSeidelSmoother ), P2P1StokesToP2P1Stokes transfer with
// projectMean after the restriction, V( pre, post ) + increment; coarse grid: MinResSolver preconditioned with
// StokesPressureBlockPreconditioner< ., P1LumpedInvMassOperator > (the reference's test uses PETSc's LU there)
HYTEG_HOST_API int hyteg_host_th_gmg_create( hh_storage_t s, int minL, int maxL, double uzawa_relax, int pre, int post, int increment,
                                             int coarse_max_iter, double coarse_rel_tol, hh_th_solver_t* out )
{
   return guarded( [&] {
      using Op     = P2P1TaylorHoodStokesOperator;
      auto storage = static_cast< StorageH* >( s )->p;
      auto gs      = std::make_shared< GaussSeidelSmoother< P2ConstantLaplaceOperator > >();
      auto vel     = std::make_shared< TaylorHoodVelocityBlockPreconditioner >( gs );
      auto uzawa   = std::make_shared< UzawaSmoother< Op > >( storage, vel, (uint_t) minL, (uint_t) maxL, uzawa_relax );
      auto prec    = std::make_shared< StokesPressureBlockPreconditioner< Op, P1LumpedInvMassOperator > >( storage, (uint_t) minL, (uint_t) minL );
      auto coarse  = std::make_shared< MinResSolver< Op > >( storage, (uint_t) minL, (uint_t) minL, (uint_t) coarse_max_iter, coarse_rel_tol, 1e-16, prec );
      *out = new THSolverH{ std::make_shared< GeometricMultigridSolver< Op, P2P1StokesToP2P1StokesRestriction, P2P1StokesToP2P1StokesProlongation > >(
          storage, uzawa, coarse, std::make_shared< P2P1StokesToP2P1StokesRestriction >( true ), std::make_shared< P2P1StokesToP2P1StokesProlongation >(),
          (uint_t) minL, (uint_t) maxL, (uint_t) pre, (uint_t) post, (uint_t) increment ) };
   } );
}
HYTEG_HOST_API int hyteg_host_th_minres_create( hh_storage_t s, int minL, int maxL, int max_iter, double rel_tol, hh_th_solver_t* out )
{
   return guarded( [&] {
      using Op     = P2P1TaylorHoodStokesOperator;
      auto storage = static_cast< StorageH* >( s )->p;
      auto prec    = std::make_shared< StokesPressureBlockPreconditioner< Op, P1LumpedInvMassOperator > >( storage, (uint_t) minL, (uint_t) maxL );
      *out = new THSolverH{ std::make_shared< MinResSolver< Op > >( storage, (uint_t) minL, (uint_t) maxL, (uint_t) max_iter, rel_tol, 1e-16, prec ) };
   } );
}
HYTEG_HOST_API int hyteg_host_th_solver_solve( hh_th_solver_t solver, hh_th_operator_t op, hh_th_function_t x, hh_th_function_t b, int level )
{
   return guarded( [&] { static_cast< THSolverH* >( solver )->p->solve( *static_cast< THOperatorH* >( op )->p, FT( x ), FT( b ), (uint_t) level ); } );
}
HYTEG_HOST_API int hyteg_host_th_solver_destroy( hh_th_solver_t solver ) { return guarded( [&] { delete static_cast< THSolverH* >( solver ); } ); }
HYTEG_HOST_API int hyteg_host_th_project_pressure_mean( hh_th_function_t f, int level )
{
   return guarded( [&] { projectMean( FT( f ).p(), (uint_t) level ); } );
}
// the 10 x 10 element matrix (FEniCS ordering) a mixed block hands to the P2 kernel: which = 0 div (edge rows zero), 1 divT (edge
// columns zero), component k, tetrahedron coords[4][3] -- for the pins against the reference's FEniCS forms
HYTEG_HOST_API int hyteg_host_th_form_element_matrix( int which, int k, const double* coords12, double* out100 )
{
   return guarded( [&] {
      std::array< Point3D, 4 > c;
      for ( int v = 0; v < 4; ++v )
         for ( int r = 0; r < 3; ++r )
            c[v][r] = coords12[3 * v + r];
      if ( k < 0 || k > 2 || which < 0 || which > 1 )
         throw std::runtime_error( "th_form_element_matrix: which = 0, 1; k = 0, 1, 2" );
      if ( which == 0 )
         k == 0 ? forms::P2ToP1DivForm< 0 >::integrateAll( c, out100 ) : ( k == 1 ? forms::P2ToP1DivForm< 1 >::integrateAll( c, out100 ) : forms::P2ToP1DivForm< 2 >::integrateAll( c, out100 ) );
      else
         k == 0 ? forms::P1ToP2DivTForm< 0 >::integrateAll( c, out100 ) : ( k == 1 ? forms::P1ToP2DivTForm< 1 >::integrateAll( c, out100 ) : forms::P1ToP2DivTForm< 2 >::integrateAll( c, out100 ) );
   } );
}
