/*
 * hyteg_host.h -- C facade over the C++ host layer (hyteg_amd/host/hyteg_host.hpp) for non-C++ callers
 * (the pytest suite, bench.py and the torch.distributed driver use it through ctypes).  The C++ classes
 * carry the reference's names (PrimitiveStorage, P1Function, P1ConstantLaplaceOperator, P1toP1Linear*,
 * GeometricMultigridSolver ...); this header only flattens them to handles.  Library: libhyteg_host.so,
 * which itself calls nothing but libhyteg_hip.so (include/hyteg_hip.h).
 *
 * All functions return 0 on success, non-zero on failure (hyteg_host_last_error() has the message); the
 * reference would WALBERLA_ABORT.  Handles are opaque pointers.  DoFType flags and UpdateType are the integer
 * values of src/hyteg/types/types.hpp:29-44 (Inner = 1, DirichletBoundary = 2, NeumannBoundary = 4, All = 15).
 */
#ifndef HYTEG_HOST_H
#define HYTEG_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#if defined( HYTEG_HOST_BUILDING )
#define HYTEG_HOST_API __attribute__( ( visibility( "default" ) ) )
#else
#define HYTEG_HOST_API
#endif

typedef void* hh_storage_t;
typedef void* hh_function_t;
typedef void* hh_operator_t;
typedef void* hh_solver_t;
typedef void* hh_p2function_t;
typedef void* hh_p2operator_t;
typedef void* hh_elementwise_t;
typedef void* hh_stokes_function_t;
typedef void* hh_stokes_operator_t;
typedef void* hh_stokes_solver_t;

HYTEG_HOST_API const char* hyteg_host_last_error( void );

/* ---- PrimitiveStorage ---- */
HYTEG_HOST_API int hyteg_host_storage_from_gmsh( const char* path, int rank, int nranks, hh_storage_t* out );
HYTEG_HOST_API int hyteg_host_storage_from_arrays( int nvertices, const double* xyz, int ncells, const int* cells, int rank, int nranks, hh_storage_t* out );
HYTEG_HOST_API int hyteg_host_storage_destroy( hh_storage_t s );
/* counts[0..5] = global cells, faces, edges, vertices, local cells, ranks */
HYTEG_HOST_API int hyteg_host_storage_counts( hh_storage_t s, int* counts );
HYTEG_HOST_API int hyteg_host_storage_local_cell( hh_storage_t s, int local_index, int* global_id, double* coords12, double* nnc14 );
HYTEG_HOST_API int hyteg_host_storage_mask( hh_storage_t s, int local_index, int flag, int owned, unsigned* mask );
HYTEG_HOST_API int hyteg_host_storage_set_boundary_type( hh_storage_t s, int dof_type );
HYTEG_HOST_API int hyteg_host_storage_set_stream( hh_storage_t s, void* stream );
/* levels <= `level` use the batched kernels (one launch for all local cells, inner and boundary points together) when the
 * rank owns more than one cell; -1 disables batching.  Default 6, or the environment variable HYTEG_AMD_BATCH_MAX_LEVEL. */
HYTEG_HOST_API int hyteg_host_storage_set_batch_max_level( hh_storage_t s, int level );
/* Timing tree with the reference's timer names (walberla::WcTimingTree behind PrimitiveStorage::getTimingTree():
 * "Operator P1Function to P1Function" / "Apply", "smooth_jac", "SOR"; "P1Function" / "Assign", "Dot (local)" ...;
 * "Geometric Multigrid Solver" / "Level L" / "Smoother" ...; src/hyteg/operators/Operator.hpp:148-166,
 * GeometricMultigridSolver.hpp:200-300) and its JSON dump (src/hyteg/dataexport/TimingOutput.hpp:46-60).  Off by default.
 * synchronize != 0: every range waits for the device before it stops (measures execution instead of enqueueing).
 * HYTEG_AMD_ROCTX=1 in the environment additionally emits roctx ranges with the same names.
 * timing_json: writes at most buflen bytes (NUL-terminated), *needed receives the size of the complete text. */
HYTEG_HOST_API int hyteg_host_storage_enable_timing( hh_storage_t s, int on, int synchronize );
HYTEG_HOST_API int hyteg_host_storage_timing_json( hh_storage_t s, char* buf, size_t buflen, size_t* needed );
HYTEG_HOST_API int hyteg_host_storage_timing_reset( hh_storage_t s );
/* Transport of the shared-point exchange of a storage distributed over several ranks (one process per GPU).
 * (a) RCCL over xGMI issued from the host layer itself: neighbour send/recv groups on a communication stream, ordered
 *     against the compute stream with events; all-reduce of dot products.  unique_id: HYTEG_HIP_COMM_ID_BYTES bytes from
 *     hyteg_hip_comm_unique_id on rank 0, distributed to all ranks by the application; collective over all ranks; call
 *     it with the rank's device current.  Semantics: BufferedCommunication.cpp:181-470 of the reference.
 * (b) hooks: exchange_begin( user, level, key ) starts an all-to-all of the registered send buffer of plan
 *     key = cls + 2 * dof_kind (may return before completion), exchange_end( user, level, key ) waits for it,
 *     allreduce_sum( user, values, n ) sums host values in place.  A hook returns 0 on success; any other value makes
 *     the calling operation fail (the host layer never reduces a receive buffer whose transfer failed). */
HYTEG_HOST_API int hyteg_host_storage_use_rccl( hh_storage_t s, const unsigned char* unique_id );
HYTEG_HOST_API int hyteg_host_storage_transport_name( hh_storage_t s, char* buf, int buflen );
/* (c) peer to peer on top of (a) or (b) (which keep the all-reduce and every plan that is not connected): the pack kernel
 *     stores into receive slots inside the peers' IPC-mapped arenas, a one-wave kernel waits for their sequence numbers
 *     (include/hyteg_hip.h, hyteg_hip_p2p_*) -- no library call per exchange.  Set-up, collective, any channel for the bytes:
 *       use_p2p( arena_bytes ) -> this rank's HYTEG_HIP_P2P_HANDLE_BYTES handle;  p2p_open( handles of all ranks, rank by rank );
 *       per plan: p2p_layout -> 3 byte offsets { slot 0, slot 1, flag } per peer of the plan, deliver triple k to rank
 *       peers[k];  p2p_connect( the triples the peers laid out for this rank, in the order of peers[] ).
 *     check_transport synchronises and fails if a device-side wait has timed out; drop_p2p returns to the wrapped transport. */
HYTEG_HOST_API int hyteg_host_storage_use_p2p( hh_storage_t s, size_t arena_bytes, unsigned char* handle, int* arena_kind );
HYTEG_HOST_API int hyteg_host_storage_p2p_open( hh_storage_t s, const unsigned char* handles );
HYTEG_HOST_API int hyteg_host_storage_p2p_layout( hh_storage_t s, int level, int key, long long* offsets );
HYTEG_HOST_API int hyteg_host_storage_p2p_connect( hh_storage_t s, int level, int key, const long long* offsets );
HYTEG_HOST_API int hyteg_host_storage_drop_p2p( hh_storage_t s );
HYTEG_HOST_API int hyteg_host_storage_check_transport( hh_storage_t s );
/* in-place sum over all ranks of n doubles in host memory through the storage's transport (no-op on one rank):
 * walberla::mpi::allReduceInplace( ..., SUM ) of the reference's norms and dot products */
HYTEG_HOST_API int hyteg_host_storage_allreduce_sum( hh_storage_t s, double* values, int n );
HYTEG_HOST_API int hyteg_host_storage_set_hooks( hh_storage_t s, int ( *exchange_begin )( void*, int, int ),
                                                 int ( *exchange_end )( void*, int, int ),
                                                 int ( *allreduce_sum )( void*, double*, int ), void* user );
/* exchange plan of (level, key = cls + 2 * dof_kind; dof_kind 0: vertex DoFs, 1: edge DoFs of P2 functions) -- host
 * data, works without a GPU.  sizes[0..4] = ngroups, nentries, npeers, total_send, total_recv */
HYTEG_HOST_API int hyteg_host_plan_sizes( hh_storage_t s, int level, int key, int* sizes );
HYTEG_HOST_API int hyteg_host_plan_export( hh_storage_t s, int level, int key, int* group_ptr, int* entry_buf, int* entry_off,
                                           int* peers, int* send_count, int* recv_count, int* send_buf, int* send_off );
HYTEG_HOST_API int hyteg_host_plan_register_buffers( hh_storage_t s, int level, int key, double* send_dev, double* recv_dev );

/* ---- P1Function<double> ---- */
HYTEG_HOST_API int hyteg_host_function_create( hh_storage_t s, const char* name, int min_level, int max_level, hh_function_t* out );
HYTEG_HOST_API int hyteg_host_function_destroy( hh_function_t f );
HYTEG_HOST_API int hyteg_host_function_cell_pointer( hh_function_t f, int local_cell, int level, double** dev_ptr );
HYTEG_HOST_API int hyteg_host_function_upload_cell( hh_function_t f, int local_cell, int level, const double* host );
HYTEG_HOST_API int hyteg_host_function_download_cell( hh_function_t f, int local_cell, int level, double* host );
HYTEG_HOST_API int hyteg_host_function_interpolate_constant( hh_function_t f, double value, int level, int flag );
HYTEG_HOST_API int hyteg_host_function_assign( hh_function_t dst, int n, const double* scalars, const hh_function_t* srcs, int level, int flag );
HYTEG_HOST_API int hyteg_host_function_add( hh_function_t dst, int n, const double* scalars, const hh_function_t* srcs, int level, int flag );
HYTEG_HOST_API int hyteg_host_function_mult_elementwise( hh_function_t dst, int n, const hh_function_t* srcs, int level, int flag );
HYTEG_HOST_API int hyteg_host_function_dot( hh_function_t a, hh_function_t b, int level, int flag, int global, double* result );
HYTEG_HOST_API int hyteg_host_function_sum_shared( hh_function_t f, int level, int flag );
HYTEG_HOST_API int hyteg_host_function_sync_shared( hh_function_t f, int level, int flag );
/* BoundaryCondition::createAllInnerBC() for this function (the pressure of a Stokes function): every point counts as Inner */
HYTEG_HOST_API int hyteg_host_function_set_all_inner( hh_function_t f, int on );

/* ---- P1ConstantOperator< Form > (src/constant_stencil_operator/P1ConstantOperator.hpp:165-210)
 * form 0: Laplace, 1: mass, 2-4: P1Div{x,y,z}Operator, 5-7: P1DivT{x,y,z}Operator, 8: P1PSPGOperator ---- */
HYTEG_HOST_API int hyteg_host_operator_create( hh_storage_t s, int min_level, int max_level, int form, hh_operator_t* out );
HYTEG_HOST_API int hyteg_host_operator_destroy( hh_operator_t op );
/* inner[15], slots[14*15] of a GLOBAL cell id */
HYTEG_HOST_API int hyteg_host_operator_stencils( hh_operator_t op, int global_cell, int level, double* inner15, double* slots210 );
HYTEG_HOST_API int hyteg_host_operator_apply( hh_operator_t op, hh_function_t src, hh_function_t dst, int level, int flag, int update );
/* `steps` consecutive applies cycling over `npairs` (src, dst) function pairs: step k uses pair (first + k) % npairs.
 * The loop a C++ application would write around Operator::apply (apps/benchmarks/ApplyBenchmark/ApplyBenchmark.cpp:95-101);
 * lets a non-C++ driver time K applies without paying its own per-call overhead K times. */
HYTEG_HOST_API int hyteg_host_operator_apply_cycle( hh_operator_t op, int npairs, const hh_function_t* srcs, const hh_function_t* dsts,
                                                    int level, int flag, int update, int first, int steps );
/* the same loop between two timing events of the C-ABI (hyteg_hip_event_create_timing; NULL = none), recorded on the storage's
 * stream directly before the first and after the last apply: the device time of exactly these `steps` applies */
HYTEG_HOST_API int hyteg_host_operator_apply_cycle_timed( hh_operator_t op, int npairs, const hh_function_t* srcs, const hh_function_t* dsts,
                                                          int level, int flag, int update, int first, int steps, void* ev_start, void* ev_stop );
HYTEG_HOST_API int hyteg_host_operator_smooth_jac( hh_operator_t op, hh_function_t dst, hh_function_t rhs, hh_function_t src, double relax, int level, int flag );
HYTEG_HOST_API int hyteg_host_operator_smooth_sor( hh_operator_t op, hh_function_t dst, hh_function_t rhs, double relax, int level, int flag, int backwards );
HYTEG_HOST_API int hyteg_host_operator_compute_inverse_diagonal( hh_operator_t op );
HYTEG_HOST_API int hyteg_host_operator_inverse_diagonal( hh_operator_t op, hh_function_t* out /* borrowed */ );

/* hyteg::operatorgeneration::P1ElementwiseDiffusion (module hyteg_operators; sample class
 * apps/2023-zikeli-mt/MT-apps/operators-used/P1ElementwiseDiffusion_cubes_const_float64.hpp:64-93): apply() hands every
 * macro-cell's arrays, vertex coordinates and micro_edges_per_macro_edge to the C-ABI seam
 * hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_masked; computeInverseDiagonalOperatorValues / getInverseDiagonalValues;
 * smooth_jac as P1ElementwiseOperator.cpp:288-331 */
HYTEG_HOST_API int hyteg_host_elementwise_create( hh_storage_t s, int min_level, int max_level, hh_elementwise_t* out );
HYTEG_HOST_API int hyteg_host_elementwise_destroy( hh_elementwise_t op );
HYTEG_HOST_API int hyteg_host_elementwise_apply( hh_elementwise_t op, hh_function_t src, hh_function_t dst, int level, int flag, int update );
HYTEG_HOST_API int hyteg_host_elementwise_compute_inverse_diagonal( hh_elementwise_t op );
HYTEG_HOST_API int hyteg_host_elementwise_inverse_diagonal( hh_elementwise_t op, hh_function_t* out /* borrowed */ );
HYTEG_HOST_API int hyteg_host_elementwise_smooth_jac( hh_elementwise_t op, hh_function_t dst, hh_function_t rhs, hh_function_t src, double relax,
                                                      int level, int flag );

/* ---- grid transfer ---- */
HYTEG_HOST_API int hyteg_host_restrict( hh_function_t f, int source_level, int flag );
HYTEG_HOST_API int hyteg_host_prolongate( hh_function_t f, int source_level, int flag );
HYTEG_HOST_API int hyteg_host_prolongate_and_add( hh_function_t f, int source_level, int flag );

/* ---- solvers on the Laplace operator ----
 * smoother: 0 = weighted Jacobi (relax), 1 = Gauss-Seidel, 2 = SOR (relax).  Coarse grid: CG. */
HYTEG_HOST_API int hyteg_host_gmg_create( hh_storage_t s, int min_level, int max_level, int smoother, double relax, int pre, int post,
                                          int wcycle, int cg_max_iter, double cg_tol, hh_solver_t* out );
/* launch graphs of the cycle (hyteg_host.hpp, GeometricMultigridSolver::setUseGraphs): opt-in, storages of one rank;
 * replayed_cycles counts the cycles that ran from a recording */
HYTEG_HOST_API int hyteg_host_gmg_set_use_graphs( hh_solver_t solver, int on );
HYTEG_HOST_API int hyteg_host_gmg_replayed_cycles( hh_solver_t solver, int* count );
/* CGSolver::setUseDeviceScalars / setUseSingleLaunch (hyteg_host.hpp); on: bit 0 = alpha, beta and the convergence test stay
 * on the device (storages of one rank up to level 5), bit 1 = problems that fit one workgroup are solved by one launch
 * (both default on); `solver` is a CG solver or a multigrid solver whose coarse solver is one */
HYTEG_HOST_API int hyteg_host_cg_set_use_device_scalars( hh_solver_t solver, int on );
/* CGSolver::getIterations of the last solve */
HYTEG_HOST_API int hyteg_host_cg_iterations( hh_solver_t solver, int* iterations );
HYTEG_HOST_API int hyteg_host_cg_create( hh_storage_t s, int min_level, int max_level, int max_iter, double tol, hh_solver_t* out );
HYTEG_HOST_API int hyteg_host_solver_solve( hh_solver_t solver, hh_operator_t laplace, hh_function_t x, hh_function_t b, int level );
HYTEG_HOST_API int hyteg_host_solver_destroy( hh_solver_t solver );

/* ---- P1-P1 Stokes: P1StokesFunction (composites/P1StokesFunction.hpp: velocity with the storage's boundary types, pressure
 * with createAllInnerBC), P1P1StokesOperator::apply (mixed_operator/P1P1StokesOperator.hpp:51-64), UzawaSmoother
 * (solvers/UzawaSmoother.hpp:262-288) over StokesVelocityBlockBlockDiagonalPreconditioner, geometric multigrid with
 * P1P1StokesToP1P1Stokes{Restriction,Prolongation} and a dense direct coarse-grid solver standing in for PETScLUSolver ---- */
HYTEG_HOST_API int hyteg_host_stokes_function_create( hh_storage_t s, const char* name, int min_level, int max_level, hh_stokes_function_t* out );
HYTEG_HOST_API int hyteg_host_stokes_function_destroy( hh_stokes_function_t f );
/* k = 0, 1, 2: velocity components, 3: pressure; the handle is a view owned by the Stokes function (do not destroy it) */
HYTEG_HOST_API int hyteg_host_stokes_function_component( hh_stokes_function_t f, int k, hh_function_t* out );
HYTEG_HOST_API int hyteg_host_stokes_function_assign( hh_stokes_function_t dst, int n, const double* scalars, const hh_stokes_function_t* fs, int level, int flag );
HYTEG_HOST_API int hyteg_host_stokes_function_dot( hh_stokes_function_t a, hh_stokes_function_t b, int level, int flag, double* out );
/* vertexdof::projectMean (VertexDoFFunction.hpp:586-592) */
HYTEG_HOST_API int hyteg_host_project_mean( hh_function_t pressure, int level );
HYTEG_HOST_API int hyteg_host_stokes_operator_create( hh_storage_t s, int min_level, int max_level, hh_stokes_operator_t* out );
HYTEG_HOST_API int hyteg_host_stokes_operator_destroy( hh_stokes_operator_t op );
HYTEG_HOST_API int hyteg_host_stokes_operator_apply( hh_stokes_operator_t op, hh_stokes_function_t src, hh_stokes_function_t dst, int level, int flag );
/* velocity_smoother: 0 weighted Jacobi, 1 Gauss-Seidel, 2 SOR (relaxation velocity_relax) on every velocity component */
HYTEG_HOST_API int hyteg_host_stokes_uzawa_create( hh_storage_t s, int min_level, int max_level, double relax, int velocity_iterations,
                                                   int velocity_smoother, double velocity_relax, hh_stokes_solver_t* out );
HYTEG_HOST_API int hyteg_host_stokes_gmg_create( hh_storage_t s, hh_stokes_solver_t smoother, int min_level, int max_level, int pre, int post,
                                                 int increment, int project_mean_after_restriction, hh_stokes_solver_t* out );
/* GeometricMultigridSolver< P1P1StokesOperator > with a choice of coarse-grid solver: 0 = dense LU on the host (stand-in for
 * PETScLUSolver, single rank), 1 = MinResSolver preconditioned with StokesPressureBlockPreconditioner< P1P1StokesOperator,
 * P1LumpedInvMassOperator > (apps/stokesSphere/StokesSphere.cpp:227-237; any number of ranks) */
HYTEG_HOST_API int hyteg_host_stokes_gmg_create_with_coarse( hh_storage_t s, hh_stokes_solver_t smoother, int min_level, int max_level, int pre, int post,
                                                             int increment, int project_mean_after_restriction, int coarse, int coarse_max_iter,
                                                             double coarse_rel_tol, hh_stokes_solver_t* out );
/* MinResSolver< P1P1StokesOperator > (src/hyteg/solvers/MinresSolver.hpp): preconditioner 0 identity, 1 pressure block (lumped
 * inverse mass), 2 StokesBlockDiagonalPreconditioner with `velocity_steps` V(2,2) Laplace cycles per velocity component */
HYTEG_HOST_API int hyteg_host_stokes_minres_create( hh_storage_t s, int min_level, int max_level, int max_iter, double rel_tol, int preconditioner,
                                                    int velocity_steps, hh_stokes_solver_t* out );
HYTEG_HOST_API int hyteg_host_stokes_minres_iterations( hh_stokes_solver_t solver, int* iterations );
/* MinResSolver< P1ConstantLaplaceOperator > with JacobiPreconditioner( jacobi_iterations ) (0: identity) */
HYTEG_HOST_API int hyteg_host_solver_create_minres( hh_storage_t s, int min_level, int max_level, int max_iter, double rel_tol, int jacobi_iterations,
                                                    hh_solver_t* out );
HYTEG_HOST_API int hyteg_host_stokes_solver_solve( hh_stokes_solver_t solver, hh_stokes_operator_t op, hh_stokes_function_t x, hh_stokes_function_t b, int level );
HYTEG_HOST_API int hyteg_host_stokes_solver_destroy( hh_stokes_solver_t solver );

/* ---- P2Function / P2ElementwiseLaplaceOperator (first version): any number of macro-cells on ONE rank ----
 * vertex part: the P1 cell array; edge part: hyteg_hip_p2_edge_array_size( level ) doubles (EdgeDoFIndexing.hpp:920-985) */
HYTEG_HOST_API int hyteg_host_p2function_create( hh_storage_t s, const char* name, int min_level, int max_level, hh_p2function_t* out );
HYTEG_HOST_API int hyteg_host_p2function_destroy( hh_p2function_t f );
HYTEG_HOST_API int hyteg_host_p2function_pointers( hh_p2function_t f, int local_cell, int level, double** vertex_dev, double** edge_dev );
HYTEG_HOST_API int hyteg_host_p2function_upload( hh_p2function_t f, int local_cell, int level, const double* vertex_host, const double* edge_host );
HYTEG_HOST_API int hyteg_host_p2function_download( hh_p2function_t f, int local_cell, int level, double* vertex_host, double* edge_host );
HYTEG_HOST_API int hyteg_host_p2function_interpolate_constant( hh_p2function_t f, double value, int level, int flag );
HYTEG_HOST_API int hyteg_host_p2function_assign( hh_p2function_t dst, int n, const double* scalars, const hh_p2function_t* srcs, int level, int flag );
HYTEG_HOST_API int hyteg_host_p2function_add( hh_p2function_t dst, int n, const double* scalars, const hh_p2function_t* srcs, int level, int flag );
HYTEG_HOST_API int hyteg_host_p2function_dot( hh_p2function_t a, hh_p2function_t b, int level, int flag, double* result );
HYTEG_HOST_API int hyteg_host_p2operator_create( hh_storage_t s, int min_level, int max_level, hh_p2operator_t* out );
/* P2toP2QuadraticProlongation::prolongate / prolongateAndAdd (add != 0) from source_level to source_level + 1 and
 * P2toP2QuadraticRestriction::restrict from source_level to source_level - 1 (src/hyteg/gridtransferoperators/) */
HYTEG_HOST_API int hyteg_host_p2_prolongate( hh_p2function_t f, int source_level, int flag, int add );
HYTEG_HOST_API int hyteg_host_p2_restrict( hh_p2function_t f, int source_level, int flag );
/* P2ConstantLaplaceOperator (src/constant_stencil_operator/P2ConstantOperator.hpp): same handle type and calls as the
 * elementwise operator; on affine macro-cells the assembled constant stencils ARE what the kernel's operator table holds */
HYTEG_HOST_API int hyteg_host_p2operator_create_constant( hh_storage_t s, int min_level, int max_level, hh_p2operator_t* out );
/* the inner-DoF stencils a P2ConstantLaplaceOperator assembled for a local cell (levels >= 2): the values of the four maps
 * v2v | e2v | v2e | e2e in the order of hyteg_hip_p2_constant_stencil_layout */
HYTEG_HOST_API int hyteg_host_p2operator_constant_stencils( hh_p2operator_t op, int local_cell, int level, double* out, int capacity, int* count );
HYTEG_HOST_API int hyteg_host_p2operator_destroy( hh_p2operator_t op );
/* the six 10 x 10 element matrices (FEniCS ordering) of a local cell at `level` */
HYTEG_HOST_API int hyteg_host_p2operator_element_matrices( hh_p2operator_t op, int local_cell, int level, double* out600 );
HYTEG_HOST_API int hyteg_host_p2operator_apply( hh_p2operator_t op, hh_p2function_t src, hh_p2function_t dst, int level, int flag, int update );
/* P2 Jacobi smoother and multigrid.  compute_inverse_diagonal: P2ElementwiseOperator::computeInverseDiagonalOperatorValues
 * (P2ElementwiseOperator.hpp:110, .cpp:420-520); smooth_jac: .cpp:344-374 (dst != src);  gmg: GeometricMultigridSolver
 * (src/hyteg/solvers/GeometricMultigridSolver.hpp) over WeightedJacobiSmoother, P2toP2QuadraticRestriction / Prolongation and a CG
 * coarse-grid solver -- the composition of tests/hyteg/P2/P2GMG3DConvergenceTest.cpp with the Jacobi smoother in place of its
 * Gauss-Seidel one (the edge-DoF Gauss-Seidel kernels are not built). */
typedef void* hh_p2solver_t;
HYTEG_HOST_API int hyteg_host_p2operator_compute_inverse_diagonal( hh_p2operator_t op );
HYTEG_HOST_API int hyteg_host_p2operator_inverse_diagonal_copy( hh_p2operator_t op, hh_p2function_t dst, int level );
HYTEG_HOST_API int hyteg_host_p2operator_smooth_jac( hh_p2operator_t op, hh_p2function_t dst, hh_p2function_t rhs, hh_p2function_t src, double relax,
                                                     int level, int flag );
/* smooth_sor: P2ConstantOperator::smooth_sor (P2ConstantOperator.cpp:113-153; macro-cell part :913-1200: vertex DoFs in
 * lexicographic order, then the edge DoFs type by type; backwards: reversed).  Exactly the reference's sweep inside a macro-cell;
 * on DoFs shared between macro-cells a different Gauss-Seidel ordering of the same splitting.  Needs
 * compute_inverse_diagonal.  gmg smoother: 0 weighted Jacobi( relax ), 1 Gauss-Seidel, 2 SOR( relax ). */
HYTEG_HOST_API int hyteg_host_p2operator_smooth_sor( hh_p2operator_t op, hh_p2function_t dst, hh_p2function_t rhs, double relax, int level, int flag,
                                                     int backwards );
HYTEG_HOST_API int hyteg_host_p2_gmg_create( hh_storage_t s, int min_level, int max_level, int smoother, double relax, int pre, int post, int wcycle,
                                             int cg_max_iter, double cg_tol, hh_p2solver_t* out );
HYTEG_HOST_API int hyteg_host_p2_solver_solve( hh_p2solver_t solver, hh_p2operator_t op, hh_p2function_t x, hh_p2function_t b, int level );
HYTEG_HOST_API int hyteg_host_p2_solver_destroy( hh_p2solver_t solver );
/* CGSolver< P2ElementwiseLaplaceOperator > on one level, flags Inner | NeumannBoundary as in the reference's CGSolver */
HYTEG_HOST_API int hyteg_host_p2_cg_solve( hh_storage_t s, hh_p2operator_t op, hh_p2function_t x, hh_p2function_t b, int level, int max_iter,
                                           double tol, int* iterations );

/* ---- P2-P1 Taylor-Hood Stokes (BASELINE config 5's "P2-P1 Stokes block operator"; hyteg_amd/host/taylorhood.hpp) ----
 * P2P1TaylorHoodFunction (composites/P2P1TaylorHoodFunction.hpp): three P2 velocity components + a P1 pressure (all-inner boundary
 * condition); P2P1TaylorHoodStokesOperator::apply (mixed_operator/P2P1TaylorHoodStokesOperator.hpp:55-64): Laplace on the velocity,
 * divT (P1 -> P2) added, div (P2 -> P1) into the pressure.  The handles returned by _velocity / _pressure are views that live as
 * long as the Taylor-Hood function.  gmg: the composition of tests/hyteg/convergence/P2P1Stokes3DUzawaConvergenceTest.cpp:150-163
 * (Uzawa over Gauss-Seidel, quadratic / linear transfer, projectMean after the restriction) with pressure-preconditioned MINRES on the
 * coarsest level in place of PETSc's LU. */
typedef void* hh_th_function_t;
typedef void* hh_th_operator_t;
typedef void* hh_th_solver_t;
HYTEG_HOST_API int hyteg_host_th_function_create( hh_storage_t s, const char* name, int min_level, int max_level, hh_th_function_t* out );
HYTEG_HOST_API int hyteg_host_th_function_destroy( hh_th_function_t f );
HYTEG_HOST_API int hyteg_host_th_function_velocity( hh_th_function_t f, int k, hh_p2function_t* out );
HYTEG_HOST_API int hyteg_host_th_function_pressure( hh_th_function_t f, hh_function_t* out );
HYTEG_HOST_API int hyteg_host_th_function_assign( hh_th_function_t dst, int n, const double* scalars, const hh_th_function_t* fs, int level, int flag );
HYTEG_HOST_API int hyteg_host_th_function_interpolate_constant( hh_th_function_t f, double value, int level, int flag );
HYTEG_HOST_API int hyteg_host_th_function_dot( hh_th_function_t a, hh_th_function_t b, int level, int flag, double* out );
HYTEG_HOST_API int hyteg_host_th_operator_create( hh_storage_t s, int min_level, int max_level, hh_th_operator_t* out );
HYTEG_HOST_API int hyteg_host_th_operator_destroy( hh_th_operator_t op );
HYTEG_HOST_API int hyteg_host_th_operator_apply( hh_th_operator_t op, hh_th_function_t src, hh_th_function_t dst, int level, int flag );
/* which: 0 div (velocity of src -> pressure of dst), 1 divT (pressure of src -> velocity of dst) */
HYTEG_HOST_API int hyteg_host_th_operator_apply_block( hh_th_operator_t op, int which, hh_th_function_t src, hh_th_function_t dst, int level, int flag );
HYTEG_HOST_API int hyteg_host_th_gmg_create( hh_storage_t s, int min_level, int max_level, double uzawa_relax, int pre, int post, int increment,
                                             int coarse_max_iter, double coarse_rel_tol, hh_th_solver_t* out );
HYTEG_HOST_API int hyteg_host_th_minres_create( hh_storage_t s, int min_level, int max_level, int max_iter, double rel_tol, hh_th_solver_t* out );
HYTEG_HOST_API int hyteg_host_th_solver_solve( hh_th_solver_t solver, hh_th_operator_t op, hh_th_function_t x, hh_th_function_t b, int level );
HYTEG_HOST_API int hyteg_host_th_solver_destroy( hh_th_solver_t solver );
HYTEG_HOST_API int hyteg_host_th_project_pressure_mean( hh_th_function_t f, int level );
/* the 10 x 10 element matrix (FEniCS ordering) of a mixed block as the P2 kernel gets it: which 0 = div (p2_to_p1_tet_div_tet, edge rows
 * zero), 1 = divT (p1_to_p2_tet_divt_tet, edge columns zero); component k; coords[4][3] */
HYTEG_HOST_API int hyteg_host_th_form_element_matrix( int which, int k, const double* coords12, double* out100 );

#ifdef __cplusplus
}
#endif
#endif
