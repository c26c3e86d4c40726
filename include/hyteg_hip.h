/*
 * hyteg_hip.h -- C-ABI of libhyteg_hip.so: MI355X (gfx950) kernels for HyTeG's matrix-free P1
 * hot path (constant-stencil apply, smoothers, vector kernels, grid transfer on macro-cells).
 *
 * This is the drop-in boundary.  HyTeG has no plugin/FFI mechanism; the seam is the set of free
 * functions in namespace hyteg::vertexdof::macrocell::generated that P1ConstantOperator and
 * VertexDoFFunction call with raw pointers.  Each entry point below names
 * the reference function (file:line under /root/reference/) it replaces.  INTEGRATION.md shows the
 * binding a HyTeG maintainer would add on the reference side.
 *
 * Conventions
 *  - plain C, no C++/torch types.  All array pointers are DEVICE pointers (hipMalloc'ed or any
 *    pointer valid on the current HIP device, e.g. a torch tensor's data_ptr()).
 *  - arrays use HyTeG's linear tetrahedral macro-cell layout
 *    (src/hyteg/indexing/MacroCellIndexing.hpp:40-52): width = 2^level + 1, x fastest, then y,
 *    then z; size = width (width+1) (width+2) / 6.
 *  - stencil weights w[15] are passed BY VALUE from host memory in the iteration order of the
 *    reference's std::map<indexing::Index, real_t> (sorted by z, then y, then x;
 *    src/hyteg/indexing/Common.hpp:67-71):
 *       0:( 0, 0,-1)  1:( 1, 0,-1)  2:(-1, 1,-1)  3:( 0, 1,-1)
 *       4:( 0,-1, 0)  5:( 1,-1, 0)  6:(-1, 0, 0)  7:( 0, 0, 0)  8:( 1, 0, 0)  9:(-1, 1, 0) 10:( 0, 1, 0)
 *      11:( 0,-1, 1) 12:( 1,-1, 1) 13:(-1, 0, 1) 14:( 0, 0, 1)
 *    so a binding fills it with   for (auto& e : stencilMap) w[k++] = e.second;
 *  - neighbour-cell counts nnc[14] = { edge0..5, face0..3, vertex0..3 }, the argument order of the
 *    reference's grid-transfer kernels.
 *  - every kernel launch is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *    legacy default stream) and stream-ordered with respect to other work on that stream.
 *  - return value: HYTEG_HIP_OK or an error code; hyteg_hip_last_error() gives a message.
 *    The reference aborts the process on failure (WALBERLA_ABORT); a binding should do the same.
 *  - fp64 only, like the reference's generated 3-D kernels (non-double aborts there:
 *    src/constant_stencil_operator/P1ConstantOperator.cpp:417-420).
 *  - levels 2..11 (int32 indexing, as in the reference's generated kernels).  The operator loops
 *    in the reference only run for level >= 2 (src/hyteg/p1functionspace/P1Operator.hpp:293).
 */
#ifndef HYTEG_HIP_H
#define HYTEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined( HYTEG_HIP_BUILDING )
#define HYTEG_HIP_API __attribute__( ( visibility( "default" ) ) )
#else
#define HYTEG_HIP_API
#endif

enum
{
   HYTEG_HIP_OK      = 0,
   HYTEG_HIP_EINVAL  = 1, /* bad argument (null pointer, level out of range, aliasing src==dst ...) */
   HYTEG_HIP_ELAUNCH = 2, /* HIP runtime reported an error for a launch / memcpy */
   HYTEG_HIP_ENOMEM  = 3,
   HYTEG_HIP_ENODEV  = 4 /* no usable gfx950 device */
};

/* UpdateType, src/hyteg/types/types.hpp:29-33 */
enum
{
   HYTEG_HIP_REPLACE = 0,
   HYTEG_HIP_ADD     = 1
};

#define HYTEG_HIP_MIN_LEVEL 2
#define HYTEG_HIP_MAX_LEVEL 11
#define HYTEG_HIP_MAX_SRCS 4 /* assign/add/mult take at most this many source functions */

typedef void* hyteg_hip_stream_t;

/* ---- runtime -------------------------------------------------------------------------------- */
HYTEG_HIP_API const char* hyteg_hip_version( void );
HYTEG_HIP_API const char* hyteg_hip_last_error( void );
HYTEG_HIP_API int         hyteg_hip_device_count( int* count );
HYTEG_HIP_API int         hyteg_hip_set_device( int device );
/* name + gcnArchName of the current device, e.g. "AMD Instinct MI355X (gfx950:sramecc+:xnack-)" */
HYTEG_HIP_API int hyteg_hip_device_name( char* buf, size_t buflen );

/* device memory standing in for FunctionMemory<double>'s per-level std::vector
 * (src/hyteg/memory/FunctionMemory.hpp:109-113,216) */
HYTEG_HIP_API int hyteg_hip_malloc( void** dev_ptr, size_t bytes );
HYTEG_HIP_API int hyteg_hip_free( void* dev_ptr );
HYTEG_HIP_API int hyteg_hip_memset_zero( void* dev_ptr, size_t bytes, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_upload( void* dev_dst, const void* host_src, size_t bytes, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_download( void* host_dst, const void* dev_src, size_t bytes, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_copy( void* dev_dst, const void* dev_src, size_t bytes, hyteg_hip_stream_t stream );
/* Calibration: a streaming copy kernel of n doubles (plain loads, nontemporal stores if `nontemporal`),
 * the practical floor a sweep over the same bytes is compared with (bench.py: roofline.copy_us).  No reference
 * counterpart (the reference's counterpart of this measurement is the STREAM table of its kerncraft machine files,
 * data/kerncraftMachineFiles/SkylakeSP_Platinum-8147_2.7GHz.yml:412-420). */
HYTEG_HIP_API int hyteg_hip_calib_copy( double* dst, const double* src, int64_t n, int nontemporal, hyteg_hip_stream_t stream );
/* `count` such copies, copy k on pair (first + k) % npairs, between two timing events (hyteg_hip_event_create_timing; NULL =
 * none) recorded on `stream` directly before the first and after the last launch. */
typedef void* hyteg_hip_event_t;
HYTEG_HIP_API int hyteg_hip_calib_copy_ring( double* const* dsts, const double* const* srcs, int npairs, int64_t n, int nontemporal, int first,
                                             int count, hyteg_hip_stream_t stream, hyteg_hip_event_t start, hyteg_hip_event_t stop );
HYTEG_HIP_API int hyteg_hip_stream_create( hyteg_hip_stream_t* stream );
HYTEG_HIP_API int hyteg_hip_stream_destroy( hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_stream_synchronize( hyteg_hip_stream_t stream );

/* Graphs: record the launches issued on `stream` between begin and end (nothing executes while recording) and replay
 * them with one call.  No counterpart in the reference (its kernels are host loops); this is what keeps the coarse
 * levels of GeometricMultigridSolver::solveRecursively (GeometricMultigridSolver.hpp:180-319), ~100 dependent launches
 * of a few microseconds, from being bound by the host's launch rate.  The null stream cannot be recorded; a recording
 * with no launches yields a NULL graph, which launches as a no-op.  abort_capture ends a recording after an error. */
typedef void* hyteg_hip_graph_t;
HYTEG_HIP_API int hyteg_hip_graph_begin_capture( hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_graph_end_capture( hyteg_hip_stream_t stream, hyteg_hip_graph_t* graph );
HYTEG_HIP_API int hyteg_hip_graph_abort_capture( hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_graph_launch( hyteg_hip_graph_t graph, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_graph_destroy( hyteg_hip_graph_t graph );

/* Builds and uploads the per-level launch tables (tile descriptors) so that later launches at that
 * level allocate nothing (HIP-graph capturable).  Launches do this lazily on first use otherwise. */
HYTEG_HIP_API int hyteg_hip_prepare_level( int level );

/* ---- layout (host-side helpers; same arithmetic as the kernels) ---------------------------------
 * src/hyteg/indexing/MacroCellIndexing.hpp:40-52, src/hyteg/Levelinfo.hpp:36-115 */
HYTEG_HIP_API int64_t hyteg_hip_cell_width( int level );      /* 2^level + 1 */
HYTEG_HIP_API int64_t hyteg_hip_cell_size( int level );       /* number of doubles in a cell array */
HYTEG_HIP_API int64_t hyteg_hip_cell_inner_size( int level ); /* DoFs the interior kernels update */
HYTEG_HIP_API int64_t hyteg_hip_cell_index( int level, int x, int y, int z );

/* ---- a2: constant-stencil apply on one macro-cell ---------------------------------------------------
 * replaces hyteg::vertexdof::macrocell::generated::apply_3D_macrocell_vertexdof_to_vertexdof_replace / _add
 *   src/constant_stencil_operator/P1generatedKernels/apply_3D_macrocell_vertexdof_to_vertexdof_replace.cpp:34-101
 *   src/constant_stencil_operator/P1generatedKernels/apply_3D_macrocell_vertexdof_to_vertexdof_add.cpp
 *   (caller src/constant_stencil_operator/P1ConstantOperator.cpp:392-428)
 * dst_i (=|+=) sum_k w_k src_{i+o_k} on the cell interior; boundary entries of dst are not touched.
 * src and dst must not alias (src/hyteg/p1functionspace/P1Operator.hpp:198). */
HYTEG_HIP_API int hyteg_hip_p1_apply_cell( double*            dst,
                                           const double*      src,
                                           int                level,
                                           const double*      w /* host, 15 */,
                                           int                update,
                                           hyteg_hip_stream_t stream );

/* ---- float instantiations of a2 / a4 and mixed-precision support -------------------------------------------------
 * The reference instantiates its generated apply kernels for float32 as well
 *   src/constant_stencil_operator/P1generatedKernels/apply_3D_macrocell_vertexdof_to_vertexdof_replace.cpp:96-97
 * (its generated SOR / Gauss-Seidel kernels are double only, sor_3D_macrocell_P1.cpp).  Arrays of float in the same
 * macro-cell layout; weights are given as doubles and converted; the arithmetic is float.  Levels 2..10. */
HYTEG_HIP_API int hyteg_hip_p1_apply_cell_f32( float* dst, const float* src, int level, const double* w /* host, 15 */, int update, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_p1_jacobi_cell_f32( float*             dst,
                                                const float*       rhs,
                                                const float*       src,
                                                const float*       invdiag /* device or NULL */,
                                                int                level,
                                                const double*      w /* host, 15 */,
                                                double             relax,
                                                hyteg_hip_stream_t stream );
/* flat conversions and y (double) += alpha * x (float): VertexDoFFunction::copyFrom between value types
 * (src/hyteg/p1functionspace/VertexDoFFunction.hpp:598-650) on device arrays of n entries */
HYTEG_HIP_API int hyteg_hip_convert_f64_to_f32( float* dst, const double* src, size_t n, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_convert_f32_to_f64( double* dst, const float* src, size_t n, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_axpy_f32_into_f64( double* y, const float* x, double alpha, size_t n, hyteg_hip_stream_t stream );

/* The two fused steps of a mixed-precision Jacobi smoother that keeps iterate and right-hand side in double and sweeps the error
 * equation in float (no reference counterpart as a kernel; the reference's pieces are the float instantiation of its apply kernel
 * and VertexDoFFunction::copyFrom between precisions, see above).  n Jacobi sweeps on A x = b equal
 *    r = b - A x;  e_1 = relax r / c;  e_{k+1} = e_k + relax ( r - A e_k ) / c  (k = 1 .. n-1);  x += e_n      (c = centre weight)
 * on the inner points of a macro-cell whose boundary values are fixed (e = 0 there):
 *   hyteg_hip_p1_residual_jacobi_start_f32: r (double arithmetic) rounded to float into r_f32, and e_1 into e_f32 -- one launch;
 *   hyteg_hip_p1_jacobi_cell_f32 (above): the sweeps in between;
 *   hyteg_hip_p1_jacobi_accumulate_f32: the last sweep, added to x (double) instead of being stored -- one launch.
 * n launches for n sweeps, 60 instead of 72 bytes per DoF for n = 3.  Levels 2..10; only inner points are read-modified. */
HYTEG_HIP_API int hyteg_hip_p1_residual_jacobi_start_f32( float* r_f32, float* e_f32, const double* rhs, const double* src, int level, const double* w,
                                                          double relax, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_p1_jacobi_accumulate_f32( double* x, const float* rhs_f32, const float* e_f32, int level, const double* w, double relax,
                                                      hyteg_hip_stream_t stream );

/* Name of the kernel instantiation hyteg_hip_p1_apply_cell( ..., level, ..., update, ... ) launches on the current device,
 * with its template arguments, e.g. "p1_apply_zmarch_preload_kernel<MODE=0,NY=2,LZ=8,EX_AUX=0,DEC=0,PFD=1>" — what profiler
 * output and recorded counter files are matched against (no reference counterpart: measurement support). */
HYTEG_HIP_API int hyteg_hip_p1_apply_kernel_name( int level, int update, char* buf, size_t buflen );
/* Tuning knob (measurement support, like hyteg_hip_set_sor_algorithm): brick shape of the z-march apply / Jacobi / residual
 * kernels for all later launches of this process -- ny rows x lz slices per wave, loads pfd slices ahead of the arithmetic.
 * Only shapes compiled into the library are accepted (EINVAL otherwise); ( 0, 0, 0 ) restores the per-level defaults.
 * Results do not depend on the shape (same terms in the same order at every point). */
HYTEG_HIP_API int hyteg_hip_set_apply_shape( int ny, int lz, int pfd );

/* ---- a4: weighted Jacobi sweep, fused ------------------------------------------------------------------
 * replaces the 1 apply + 3 vector passes of P1Operator::smooth_jac
 *   src/hyteg/p1functionspace/P1Operator.hpp:429-447
 * dst = src + relax * invDiag .* ( rhs - A src )  on the cell interior.
 * invdiag: cell array of inverse diagonal values (getInverseDiagonalValues()), or NULL to use the
 * constant 1/w[7] (what computeInverseDiagonalOperatorValues() stores on a constant-stencil cell,
 * P1Operator.hpp:636-906).  dst must not alias src. */
HYTEG_HIP_API int hyteg_hip_p1_jacobi_cell( double*            dst,
                                            const double*      rhs,
                                            const double*      src,
                                            const double*      invdiag /* device or NULL */,
                                            int                level,
                                            const double*      w /* host, 15 */,
                                            double             relax,
                                            hyteg_hip_stream_t stream );

/* Selects how hyteg_hip_p1_sor_cell / hyteg_hip_p1_sor_cells execute the sweep.  Every form visits the points in an
 * order that respects the reference's lexicographic (z,y,x) dependencies (sor_3D_macrocell_P1.cpp:48-88), so the results
 * agree up to the rounding of the 15-term sum.  Process-wide; meant for tests and benchmarks.
 *   AUTO      levels >= 5: BLOCKS; below: PLANES (batches at levels <= 5: one workgroup per cell)
 *   PLANES    one launch per hyperplane x + 2y + 3z
 *   BLOCKS    16^3 blocks in skewed coordinates, one launch per block wavefront
 *   DATAFLOW  one launch; 8 x 8-row columns, one wave each, hand results over through progress words (levels >= 3);
 *             measured slower than BLOCKS on MI355X, kept as an opt-in */
enum hyteg_hip_sor_algorithm
{
   HYTEG_HIP_SOR_AUTO     = 0,
   HYTEG_HIP_SOR_PLANES   = 1,
   HYTEG_HIP_SOR_BLOCKS   = 2,
   HYTEG_HIP_SOR_DATAFLOW = 3
};
HYTEG_HIP_API int hyteg_hip_set_sor_algorithm( int algorithm );

/* ---- a3: SOR / Gauss-Seidel sweep, in place, in the reference's lexicographic order -----------------
 * replaces sor_3D_macrocell_P1, sor_3D_macrocell_P1_backwards, gaussseidel_3D_macrocell_P1 (relax = 1)
 *   src/constant_stencil_operator/P1generatedKernels/sor_3D_macrocell_P1.cpp:32-90
 *   src/constant_stencil_operator/P1generatedKernels/sor_3D_macrocell_P1_backwards.cpp:52-57
 *   src/constant_stencil_operator/P1generatedKernels/gaussseidel_3D_macrocell_P1.cpp:32-88
 *   (caller P1ConstantOperator.cpp:639-677)
 * The sweep is executed as hyperplanes x + 2y + 3z = const, which reproduces the sequential
 * (z, y, x) update order of the reference exactly (every already-updated neighbour lies on an
 * earlier plane, every not-yet-updated one on a later plane). */
HYTEG_HIP_API int hyteg_hip_p1_sor_cell( double*            u,
                                         const double*      rhs,
                                         int                level,
                                         const double*      w /* host, 15 */,
                                         double             relax,
                                         int                backwards,
                                         hyteg_hip_stream_t stream );
/* nsweeps consecutive sweeps of the same direction: what a smoother's pre- / post-smoothing loop does to a macro-cell
 * whose boundary values do not change between the sweeps (GeometricMultigridSolver.hpp:209-215 calling
 * P1ConstantOperator::smooth_sor).  Same result, bit for bit, as nsweeps calls of hyteg_hip_p1_sor_cell; from level 5 on
 * the sweeps run as one pipeline of block wavefronts (sweep s + 1 four wavefronts behind sweep s), 46 + 4 ( nsweeps - 1 )
 * launches instead of 46 nsweeps at level 8. */
HYTEG_HIP_API int hyteg_hip_p1_sor_cell_sweeps( double*            u,
                                                const double*      rhs,
                                                int                level,
                                                const double*      w /* host, 15 */,
                                                double             relax,
                                                int                backwards,
                                                int                nsweeps,
                                                hyteg_hip_stream_t stream );

/* dst = rhs - A src on the cell interior in one launch: what a multigrid cycle computes with apply + assign( { 1, -1 } )
 * (GeometricMultigridSolver.hpp:240-246: A.apply( x, tmp ); tmp.assign( { 1.0, -1.0 }, { b, tmp } )); the same bits as the two
 * calls.  dst may be rhs, not src.  Levels 2..10. */
HYTEG_HIP_API int hyteg_hip_p1_residual_cell( double*            dst,
                                              const double*      rhs,
                                              const double*      src,
                                              int                level,
                                              const double*      w /* host, 15 */,
                                              hyteg_hip_stream_t stream );

/* ---- a6: vector kernels on the cell interior ---------------------------------------------------------
 * assign:  dst = sum_k c_k src_k      replaces assign_3D_macrocell_vertexdof_{1,2,3}_rhs_function(s)
 *   src/hyteg/p1functionspace/generatedKernels/assign_3D_macrocell_vertexdof_*.cpp
 *   (dispatch src/hyteg/p1functionspace/VertexDoFFunction.cpp:1088-1128; generic
 *    src/hyteg/p1functionspace/VertexDoFMacroCell.hpp:376-402)
 * add:     dst += sum_k c_k src_k     VertexDoFMacroCell.hpp:456-481, add_3D_macrocell_vertexdof_1_rhsfunction.cpp
 * mult:    dst = prod_k src_k         VertexDoFMacroCell.hpp:483-508 (multElementwise)
 * srcs: HOST array of nsrc device pointers (1 <= nsrc <= HYTEG_HIP_MAX_SRCS); sources may alias dst. */
HYTEG_HIP_API int hyteg_hip_p1_assign_cell( double*              dst,
                                            int                  nsrc,
                                            const double* const* srcs,
                                            const double*        scalars /* host, nsrc */,
                                            int                  level,
                                            hyteg_hip_stream_t   stream );
HYTEG_HIP_API int hyteg_hip_p1_add_cell( double*              dst,
                                         int                  nsrc,
                                         const double* const* srcs,
                                         const double*        scalars /* host, nsrc */,
                                         int                  level,
                                         hyteg_hip_stream_t   stream );
HYTEG_HIP_API int
    hyteg_hip_p1_mult_cell( double* dst, int nsrc, const double* const* srcs, int level, hyteg_hip_stream_t stream );

/* dot: sum over the cell interior of a_i * b_i   replaces vertexdof::macrocell::dot
 *   src/hyteg/p1functionspace/VertexDoFMacroCell.hpp:511-529
 * Two deterministic passes (per-workgroup partial sums, then one workgroup adds them in fixed
 * order); the result is written to *result_dev (device memory, stream-ordered).  workspace_dev must
 * hold hyteg_hip_dot_workspace_bytes() bytes and must not be shared by dots in flight on different
 * streams.  The reference's plain sequential sum differs by reassociation only. */
HYTEG_HIP_API size_t hyteg_hip_dot_workspace_bytes( void );
HYTEG_HIP_API int    hyteg_hip_p1_dot_cell( const double*      a,
                                            const double*      b,
                                            int                level,
                                            double*            result_dev,
                                            void*              workspace_dev,
                                            hyteg_hip_stream_t stream );

/* ---- a7 / a8: grid transfer on one macro-cell ----------------------------------------------------------
 * restrict: replaces restrict_3D_macrocell_P1_pull_additive<double>
 *   src/hyteg/gridtransferoperators/generatedKernels/restrict_3D_macrocell_P1_pull_additive.cpp:33-1690
 *   (caller src/hyteg/gridtransferoperators/P1toP1LinearRestriction.cpp:169-243)
 *   writes EVERY coarse point (boundary included) with this cell's partial sum.
 * prolongate: replaces the zeroing at P1toP1LinearProlongation.cpp:214-238 plus
 *   prolongate_3D_macrocell_P1_push_additive<double>
 *   src/hyteg/gridtransferoperators/generatedKernels/prolongate_3D_macrocell_P1_push_additive.cpp
 *   The reference scatters from coarse points; here every fine point gathers its (at most two)
 *   coarse parents, adding them in the order the scatter would have.  update = REPLACE overwrites
 *   the whole fine array; update = ADD overwrites the boundary shell and adds to the interior. */
HYTEG_HIP_API int hyteg_hip_p1_restrict_cell( double*            coarse,
                                              const double*      fine,
                                              int                coarse_level,
                                              const double*      nnc /* host, 14 */,
                                              hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_p1_prolongate_cell( const double*      coarse,
                                                double*            fine,
                                                int                coarse_level,
                                                const double*      nnc /* host, 14 */,
                                                int                update,
                                                hyteg_hip_stream_t stream );

/* ---- cell-centric multi-cell support -------------------------------------------------------------------
 * On meshes with several macro-cells this library keeps the DoFs of macro-faces/edges/vertices in the boundary
 * entries of every adjacent cell array (HyTeG's cell arrays hold the same copies after communicate<Face,Cell>,
 * src/hyteg/p1functionspace/VertexDoFPackInfo.hpp:410-478).  Points are addressed by a 15-bit mask:
 * bit k (k = 0..13) selects the points on the macro-primitive in slot k of { edge0..5, face0..3, vertex0..3 }
 * (the cell-local numbering of src/hyteg/indexing/MacroCellIndexing.cpp:36-91), bit 14 the cell interior.
 * A host layer derives the mask from the DoFType flag and the primitives' boundary conditions, which is the
 * test `testFlag( bc.getBoundaryType( prim.getMeshBoundaryFlag() ), flag )` of P1Operator.hpp:213-303. */
#define HYTEG_HIP_SLOT_INNER 14
#define HYTEG_HIP_MASK_INNER ( 1u << 14 )
#define HYTEG_HIP_MASK_SHELL 0x3FFFu
#define HYTEG_HIP_MASK_ALL 0x7FFFu

/* a9: apply on the points of the cell boundary with this cell's PARTIAL stencils
 * (what P1Elements3D::calculateStencilInMacroCell gives for a micro-vertex on that face/edge/vertex,
 *  src/hyteg/p1functionspace/P1Elements.hpp:215-380; the reference stores them per neighbour cell in
 *  faceStencil3D / edgeStencil3D, src/constant_stencil_operator/P1ConstantOperator.cpp:239-357).
 * w_slots: host, 14 x 15 weights (slot-major, stencil order as above; weights of neighbours outside the cell
 * are ignored).  dst_i (=|+=) sum over the neighbours inside the cell.  Summing the results of all cells that
 * share a point gives the reference's value of apply() on that macro-face/edge/vertex DoF. */
HYTEG_HIP_API int hyteg_hip_p1_apply_cell_boundary( double*            dst,
                                                    const double*      src,
                                                    int                level,
                                                    const double*      w_slots /* host, 14*15 */,
                                                    unsigned           mask,
                                                    int                update,
                                                    hyteg_hip_stream_t stream );

/* a6 with a point mask: op 0 = assign, 1 = add, 2 = multElementwise (VertexDoFFunction.cpp:1130-1221,
 * 1408-1484, 1487-1563 loop over vertices, edges, faces and cells with the same flag test). */
HYTEG_HIP_API int hyteg_hip_p1_vector_cell_masked( int                  op,
                                                   double*              dst,
                                                   int                  nsrc,
                                                   const double* const* srcs,
                                                   const double*        scalars /* host, nsrc; ignored for op 2 */,
                                                   int                  level,
                                                   unsigned             mask,
                                                   hyteg_hip_stream_t   stream );
/* dst = value on the masked points (VertexDoFFunction::interpolate( constant, level, flag )) */
HYTEG_HIP_API int
    hyteg_hip_p1_set_cell_masked( double* dst, double value, int level, unsigned mask, hyteg_hip_stream_t stream );
/* dot over the masked points (dotLocal sums vertices + edges + faces + cells, VertexDoFFunction.cpp:1720-1793;
 * the host passes, per cell, the mask of the primitives that cell "owns" so that every DoF counts once). */
HYTEG_HIP_API int hyteg_hip_p1_dot_cell_masked( const double*      a,
                                                const double*      b,
                                                int                level,
                                                unsigned           mask,
                                                double*            result_dev,
                                                void*              workspace_dev,
                                                hyteg_hip_stream_t stream );
/* grid transfer writing only the masked coarse / fine points (interior bit must be set) */
HYTEG_HIP_API int hyteg_hip_p1_restrict_cell_masked( double*            coarse,
                                                     const double*      fine,
                                                     int                coarse_level,
                                                     const double*      nnc /* host, 14 */,
                                                     unsigned           mask,
                                                     hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_p1_prolongate_cell_masked( const double*      coarse,
                                                       double*            fine,
                                                       int                coarse_level,
                                                       const double*      nnc /* host, 14 */,
                                                       unsigned           mask,
                                                       hyteg_hip_stream_t stream );
/* the same with an update type: ADD adds the interpolant to the selected interior points (bit 14) and, like
 * hyteg_hip_p1_prolongate_cell, overwrites the selected shell points -- with an interior-only mask it is
 * P1toP1LinearProlongation::prolongateAndAdd on a macro-cell whose shell is not selected, in one launch */
HYTEG_HIP_API int hyteg_hip_p1_prolongate_cell_masked_update( const double*      coarse,
                                                              double*            fine,
                                                              int                coarse_level,
                                                              const double*      nnc /* host, 14 */,
                                                              unsigned           mask,
                                                              int                update,
                                                              hyteg_hip_stream_t stream );

/* a3 on the macro-cell boundary: SOR / Gauss-Seidel on the macro-vertices, macro-edges and macro-faces around one cell
 * (vertexdof::macrovertex::smooth_sor, VertexDoFMacroVertex.hpp:231-251; P1Operator::smooth_sor_edge,
 * P1Operator.hpp:1352-1421; P1Operator::smooth_sor_face3D / the generated sor_3D_macroface_P1, :1424-1503), in the order
 * of P1Operator::smooth_sor (:348-418): vertices, edges, faces -- faces, edges, vertices if `backwards`.
 * The reference sweeps each macro-primitive in its own memory with ghost layers; here the sweep runs in the cell's
 * array on the points selected by `mask` (bits 0..13), with
 *   rest       : for every selected point the stencil sum over all neighbours that are NOT on the same macro-primitive
 *                (nor on its lower-dimensional boundary), already summed over all neighbour cells -- the ghost-layer
 *                part of the reference's sum.  Overwritten on macro-face points (scratch).
 *   edge_verts : [6][2] cell-local vertex numbers (a, b): edge e is swept from a to b (the macro-edge's orientation)
 *   edge_w     : [6][3] total weights: centre, neighbour towards a, neighbour towards b
 *   face_verts : [4][3] cell-local numbers of the macro-face's vertices 0, 1, 2 (rows y along 0->2, x along 0->1)
 *   face_w     : [4][7] total weights: centre, then (-1,0) (1,0) (0,-1) (0,1) (1,-1) (-1,1) in face coordinates
 *   vertex_w   : [4]    total centre weights
 * (host arrays).  Every cell around a shared primitive runs the same sweep on its own copy, so the copies stay
 * bit-identical without an exchange.  Levels 0..11. */
HYTEG_HIP_API int hyteg_hip_p1_sor_shell_cell( double*            dst,
                                               const double*      rhs,
                                               double*            rest,
                                               int                level,
                                               const int*         edge_verts,
                                               const double*      edge_w,
                                               const int*         face_verts,
                                               const double*      face_w,
                                               const double*      vertex_w,
                                               double             relax,
                                               unsigned           mask,
                                               int                backwards,
                                               hyteg_hip_stream_t stream );

/* ---- batched forms: ONE launch for up to HYTEG_HIP_MAX_BATCH macro-cells of the same level, inner and
 * boundary points together.  They replace the per-primitive loops of the reference (`for ( cell : storage->getCells() )`,
 * P1Operator.hpp:286-309; VertexDoFFunction.cpp:1130-1221, 1710-1793; P1toP1LinearRestriction.cpp:193-243) on the levels
 * where a cell is too small to fill the GPU.  `dst`, `src`, ... are HOST arrays of `ncells` device pointers, `masks` a host
 * array of `ncells` point masks (bits 0..13 macro-primitive slots, bit 14 inner points); tables ending in `_dev` are device
 * memory.  Levels 0..11. */
#define HYTEG_HIP_MAX_BATCH 80
/* op 0 assign, 1 add, 2 multElementwise, 3 set to scalars[0]; srcs = [nsrc][ncells] */
HYTEG_HIP_API int hyteg_hip_p1_vector_cells( int                  op,
                                             int                  ncells,
                                             double* const*       dst,
                                             int                  nsrc,
                                             const double* const* srcs,
                                             const double*        scalars,
                                             int                  level,
                                             const unsigned*      masks,
                                             hyteg_hip_stream_t   stream );
/* assign / add whose coefficients are read from device memory when the kernel runs (scalar_ptrs: host array of nsrc
 * device pointers): lets an iteration whose scalars are themselves results of device reductions proceed without a
 * host round trip per scalar (the conjugate gradient recurrences below) */
HYTEG_HIP_API int hyteg_hip_p1_vector_cells_dev( int                  op,
                                                 int                  ncells,
                                                 double* const*       dst,
                                                 int                  nsrc,
                                                 const double* const* srcs,
                                                 const double* const* scalar_ptrs,
                                                 int                  level,
                                                 const unsigned*      masks,
                                                 hyteg_hip_stream_t   stream );
/* Scalar recurrences of CGSolver::solve (src/hyteg/solvers/CGSolver.hpp:91-140; identity preconditioner) on the device.
 * s_dev: HYTEG_HIP_CG_SLOTS doubles.  The caller writes the dot products into RR / PAP (hyteg_hip_p1_dot_cells) and calls
 *   phase 0  after RR = <r,r> of the initial residual: prsold = RR, res_start = sqrt(RR), done = res_start < abs_tol
 *   phase 1  after PAP = <p,Ap>:  alpha = prsold / PAP, NEG_ALPHA = -alpha        (both 0 once done)
 *   phase 2  after RR = <r,r>:    iterations += 1; done if sqrt(RR)/res_start < rel_tol or sqrt(RR) < abs_tol,
 *                                 else beta = RR / prsold, prsold = RR            (nothing once done)
 * With alpha = 0 the updates x += alpha p, r -= alpha Ap leave x and r as they are, so iterations enqueued beyond
 * convergence change nothing and the host only needs to look at DONE every few iterations. */
enum hyteg_hip_cg_slot
{
   HYTEG_HIP_CG_PRSOLD     = 0,
   HYTEG_HIP_CG_PAP        = 1,
   HYTEG_HIP_CG_RR         = 2,
   HYTEG_HIP_CG_ALPHA      = 3,
   HYTEG_HIP_CG_NEG_ALPHA  = 4,
   HYTEG_HIP_CG_BETA       = 5,
   HYTEG_HIP_CG_RES_START  = 6,
   HYTEG_HIP_CG_DONE       = 7,
   HYTEG_HIP_CG_ITERATIONS = 8,
   HYTEG_HIP_CG_ONE        = 9,
   HYTEG_HIP_CG_SLOTS      = 16
};
HYTEG_HIP_API int hyteg_hip_cg_scalars( double* s_dev, int phase, double rel_tol, double abs_tol, hyteg_hip_stream_t stream );
/* The whole conjugate gradient solve of CGSolver::solve (CGSolver.hpp:91-140, identity preconditioner) in ONE launch of one
 * workgroup, for problems whose cell arrays together have at most hyteg_hip_p1_cg_small_max_entries() entries (the
 * coarsest level of a multigrid hierarchy).  x: initial guess in, solution out (entries the masks do not select are read
 * as boundary values); operator and masks as in hyteg_hip_p1_apply_cells; owned_masks select the points a dot product
 * counts (each shared point once).  The copies of shared points are summed inside the kernel: group tables of up to two
 * exchange classes (group_ptr[k]: ngroups[k] + 1 offsets into entry_cell[k] / entry_off[k], all device arrays; entries
 * name a cell of this batch and an array offset).  info_dev (may be NULL): [0] iterations, [1] final residual norm. */
HYTEG_HIP_API int hyteg_hip_p1_cg_small_max_entries( void );
HYTEG_HIP_API int hyteg_hip_p1_cg_small_cells( int                  ncells,
                                               double* const*       x,
                                               const double* const* b,
                                               int                  level,
                                               const double*        stencils_dev,
                                               const unsigned*      masks,
                                               const unsigned*      owned_masks,
                                               const int* const*    group_ptr_dev,
                                               const int* const*    entry_cell_dev,
                                               const int* const*    entry_off_dev,
                                               const int*           ngroups,
                                               int                  max_iter,
                                               double               rel_tol,
                                               double               abs_tol,
                                               double*              info_dev,
                                               hyteg_hip_stream_t   stream );
/* hyteg_hip_p1_dot_cells into s_dev[slot] (slot = HYTEG_HIP_CG_PAP or HYTEG_HIP_CG_RR) followed by hyteg_hip_cg_scalars( phase )
 * in the same launch */
HYTEG_HIP_API int hyteg_hip_p1_dot_cells_cg( int                  ncells,
                                             const double* const* a,
                                             const double* const* b,
                                             int                  level,
                                             const unsigned*      masks,
                                             double*              s_dev,
                                             int                  slot,
                                             int                  phase,
                                             double               rel_tol,
                                             double               abs_tol,
                                             void*                workspace_dev,
                                             hyteg_hip_stream_t   stream );
/* *result_dev = sum over all cells and masked points of a.b; up to 64 workgroups: one launch (the workgroup that finishes
 * last reduces the partial sums), more: a second launch does; fixed order either way: the result does not depend on timing;
 * workspace: hyteg_hip_dot_workspace_bytes() */
HYTEG_HIP_API int hyteg_hip_p1_dot_cells( int                  ncells,
                                          const double* const* a,
                                          const double* const* b,
                                          int                  level,
                                          const unsigned*      masks,
                                          double*              result_dev,
                                          void*                workspace_dev,
                                          hyteg_hip_stream_t   stream );
/* stencils_dev: [ncells][15][15]; rows 0..13 = the cell's share of the stencil at a point of macro-primitive slot s
 * (w_slots of hyteg_hip_p1_apply_cell_boundary), row 14 = the inner stencil (w of hyteg_hip_p1_apply_cell) */
HYTEG_HIP_API int hyteg_hip_p1_apply_cells( int                  ncells,
                                            double* const*       dst,
                                            const double* const* src,
                                            int                  level,
                                            const double*        stencils_dev,
                                            const unsigned*      masks,
                                            int                  update,
                                            hyteg_hip_stream_t   stream );
/* smooth_jac in two phases around the additive exchange of dst: phase 0 writes the complete update on inner points and the
 * cell's share of A.src on boundary points; phase 1 turns the summed shares into dst = src + relax*invdiag*( rhs - dst ) */
HYTEG_HIP_API int hyteg_hip_p1_jacobi_cells( int                  ncells,
                                             double* const*       dst,
                                             const double* const* rhs,
                                             const double* const* src,
                                             const double* const* invdiag,
                                             int                  level,
                                             const double*        stencils_dev,
                                             double               relax,
                                             const unsigned*      masks,
                                             int                  phase,
                                             hyteg_hip_stream_t   stream );
/* nnc_inv_dev: [ncells][14] = 1 / numNeighborCells in the slot order of hyteg_hip_p1_restrict_cell */
HYTEG_HIP_API int hyteg_hip_p1_restrict_cells( int                  ncells,
                                               double* const*       coarse,
                                               const double* const* fine,
                                               int                  coarse_level,
                                               const double*        nnc_inv_dev,
                                               const unsigned*      masks,
                                               hyteg_hip_stream_t   stream );
HYTEG_HIP_API int hyteg_hip_p1_prolongate_cells( int                  ncells,
                                                 const double* const* coarse,
                                                 double* const*       fine,
                                                 int                  coarse_level,
                                                 const double*        nnc_inv_dev,
                                                 const unsigned*      masks,
                                                 int                  update,
                                                 hyteg_hip_stream_t   stream );

/* in-place SOR / GS sweep over the inner points of every cell whose mask has bit 14 (hyteg_hip_p1_sor_cell for a batch);
 * levels <= 5: one workgroup per cell with the array in LDS, above: the blocked sweep with the cell as second grid dimension.
 * stencils_dev as for hyteg_hip_p1_apply_cells (row 14 is used). */
HYTEG_HIP_API int hyteg_hip_p1_sor_cells( int                  ncells,
                                          double* const*       u,
                                          const double* const* rhs,
                                          int                  level,
                                          const double*        stencils_dev,
                                          double               relax,
                                          int                  backwards,
                                          const unsigned*      masks,
                                          hyteg_hip_stream_t   stream );
/* per-cell tables of hyteg_hip_p1_sor_shell_cell as one record (device array of `ncells` records for the batched form) */
typedef struct hyteg_hip_sor_shell_tables
{
   int    edge_verts[6][2];
   int    face_verts[4][3];
   double edge_w[6][3];
   double face_w[4][7];
   double vertex_w[4];
} hyteg_hip_sor_shell_tables;
HYTEG_HIP_API int hyteg_hip_p1_sor_shell_cells( int                               ncells,
                                                double* const*                    dst,
                                                const double* const*              rhs,
                                                double* const*                    rest,
                                                int                               level,
                                                const hyteg_hip_sor_shell_tables* tables_dev,
                                                double                            relax,
                                                const unsigned*                   masks,
                                                int                               backwards,
                                                hyteg_hip_stream_t                stream );

/* ---- f1: P2 (vertex + edge DoFs) elementwise operator on one macro-cell ----
 * P2ElementwiseOperator::gemv, src/hyteg/elementwiseoperators/P2ElementwiseOperator.cpp:110-223 (cell loop) with
 * localMatrixVectorMultiply3D (:66-107).  Arrays: the vertex-DoF array of the P1 kernels and the edge-DoF array of
 * src/hyteg/edgedofspace/EdgeDoFIndexing.hpp:920-985: blocks X, Y, Z, XY, XZ, YZ of tet(2^level) entries and XYZ of
 * tet(2^level - 1).  The operator of an affine macro-cell is given by the six 10 x 10 element matrices of the micro-cell
 * types WHITE_UP, BLUE_UP, GREEN_UP, WHITE_DOWN, BLUE_DOWN, GREEN_DOWN (celldof::allCellTypes) in FEniCS ordering
 * (P2Form::integrateAll, kernel INPUT): hyteg_hip_p2_build_operator_table turns elmat[6][10][10] (host) into the table the
 * kernel reads -- the matrices themselves, used micro-cell by micro-cell for DoFs on the macro-cell boundary, followed by
 * the constant stencils they imply for inner DoFs (hyteg_hip_p2_operator_table_size() doubles); the caller uploads it
 * once per (cell, level).  Every DoF whose point class is in `mask` (vertex DoFs as for the P1 kernels; an
 * edge DoF belongs to the macro-primitive that contains both its end points) receives alpha * (A src) (REPLACE) or has
 * it added (ADD).  Boundary DoFs: the sum over micro-cells is taken in the reference's loop order; inner DoFs: in stencil order.
 * Levels 0..9. */
#define HYTEG_HIP_P2_MAX_LEVEL 9
HYTEG_HIP_API size_t hyteg_hip_p2_edge_array_size( int level );
HYTEG_HIP_API size_t hyteg_hip_p2_operator_table_size( void );
HYTEG_HIP_API int    hyteg_hip_p2_build_operator_table( const double* elmat_host /* 600 */, double* table_host );
/* EdgeDoFFunction assign / add / multElementwise / interpolate( constant ) and dotLocal on the edge-DoF array of one
 * macro-cell (src/hyteg/edgedofspace/EdgeDoFFunction.cpp), restricted to the DoFs whose point class is in `mask`;
 * op and workspace as for hyteg_hip_p1_vector_cell_masked / hyteg_hip_p1_dot_cell_masked */
HYTEG_HIP_API int hyteg_hip_p2_edge_vector_cell_masked( int                  op,
                                                        double*              dst,
                                                        int                  nsrc,
                                                        const double* const* srcs,
                                                        const double*        scalars,
                                                        int                  level,
                                                        unsigned             mask,
                                                        hyteg_hip_stream_t   stream );
/* the same restricted further to the edge-DoF orientations in kind_mask (bit 1..7 = X, Y, Z, XY, XZ, YZ, XYZ): the sweeps
 * "by type" of the P2 Gauss-Seidel smoother (sor_3D_macrocell_P2_update_edgedofs_by_type_*, P2ConstantOperator.cpp:913-1200) */
HYTEG_HIP_API int hyteg_hip_p2_edge_vector_cell_kinds( int                  op,
                                                       double*              dst,
                                                       int                  nsrc,
                                                       const double* const* srcs,
                                                       const double*        scalars,
                                                       int                  level,
                                                       unsigned             mask,
                                                       unsigned             kind_mask,
                                                       hyteg_hip_stream_t   stream );
HYTEG_HIP_API int hyteg_hip_p2_edge_dot_cell_masked( const double*      a,
                                                     const double*      b,
                                                     int                level,
                                                     unsigned           mask,
                                                     double*            result_dev,
                                                     void*              workspace_dev,
                                                     hyteg_hip_stream_t stream );
HYTEG_HIP_API int    hyteg_hip_p2_elementwise_apply_cell( double*            dst_vertex,
                                                          double*            dst_edge,
                                                          const double*      src_vertex,
                                                          const double*      src_edge,
                                                          int                level,
                                                          const double*      optable_dev,
                                                          double             alpha,
                                                          int                update,
                                                          unsigned           mask,
                                                          hyteg_hip_stream_t stream );
/* ... only the destination kinds in kind_mask (bit 0: vertex DoFs, 1..7: edge DoFs X, Y, Z, XY, XZ, YZ, XYZ) are computed and
 * written; the others are left alone */
HYTEG_HIP_API int    hyteg_hip_p2_elementwise_apply_cell_kinds( double*            dst_vertex,
                                                          double*            dst_edge,
                                                          const double*      src_vertex,
                                                          const double*      src_edge,
                                                          int                level,
                                                          const double*      optable_dev,
                                                          double             alpha,
                                                          int                update,
                                                          unsigned           mask,
                                                          unsigned           kind_mask,
                                                          hyteg_hip_stream_t stream );

/* The level from which hyteg_hip_p2_elementwise_apply_cell( .., kind mask 0xFF, mask with the inner DoFs ) uses the row kernel that
 * computes every point class (default 3, its lowest; 99 = never: the row kernel for the inner DoFs + the thread-per-DoF boundary kernel
 * of rounds 1-2).  Returns the previous value.  Results of the two agree to rounding (different partial sums), so this is a tuning and
 * test switch, not part of the operator interface (no reference counterpart). */
HYTEG_HIP_API int hyteg_hip_p2_set_class_rows_min_level( int level );

/* hyteg_hip_p2_edge_vector_cell_kinds for up to HYTEG_HIP_MAX_BATCH macro-cells of one level in ONE launch (the EdgeDoFFunction loops over
 * the macro-cells of a rank, src/hyteg/edgedofspace/EdgeDoFFunction.cpp): dst and masks are HOST arrays of ncells entries, srcs a HOST
 * array [nsrc][ncells] of device pointers */
HYTEG_HIP_API int hyteg_hip_p2_edge_vector_cells_kinds( int op, int ncells, double* const* dst, int nsrc, const double* const* srcs,
                                                        const double* scalars, int level, const unsigned* masks, unsigned kind_mask,
                                                        hyteg_hip_stream_t stream );

/* hyteg_hip_p2_edge_dot_cell_masked for up to HYTEG_HIP_MAX_BATCH macro-cells in one launch: results_dev[c] = the masked dot product of cell c
 * (a, b, masks: HOST arrays of ncells entries; one workgroup per cell, fixed summation order) */
HYTEG_HIP_API int hyteg_hip_p2_edge_dot_cells_masked( int ncells, const double* const* a, const double* const* b, int level, const unsigned* masks,
                                                      double* results_dev, hyteg_hip_stream_t stream );
/* hyteg_hip_p2_elementwise_apply_cell_kinds for up to HYTEG_HIP_MAX_BATCH macro-cells of one level in two launches (inner DoFs, boundary
 * DoFs; P2ElementwiseOperator::gemv's loop over the macro-cells, P2ElementwiseOperator.cpp:131-223): all arguments per cell as HOST
 * arrays of ncells entries.  Levels 2..6 (thread-per-DoF kernels: where a launch per cell is pure launch latency). */
HYTEG_HIP_API int hyteg_hip_p2_elementwise_apply_cells_kinds( int ncells, double* const* dst_vertex, double* const* dst_edge, const double* const* src_vertex,
                                                              const double* const* src_edge, int level, const double* const* optables_dev, double alpha,
                                                              int update, const unsigned* masks, unsigned kind_mask, hyteg_hip_stream_t stream );

/* ---- P2 Gauss-Seidel / SOR on macro-primitives shared between macro-cells, in the reference's order ----
 * P2ConstantOperator::smooth_sor (src/constant_stencil_operator/P2ConstantOperator.cpp:1267-1330): macro-vertices (:157-201),
 * macro-edges (:205-266, P2MacroEdge.cpp:617-680), macro-faces (:269-880, kernels
 * P2generatedKernels/sor_3D_macroface_P2_update_{vertexdofs,edgedofs}*.cpp), macro-cells.  A primitive's sweep sees current
 * values on itself and on its boundary and ghost-layer values elsewhere.  Pieces for a cell-centric implementation:
 *  - hyteg_hip_p2_operator_table_closure_split (host arrays of hyteg_hip_p2_operator_table_size() doubles): splits the rows of
 *    the 14 boundary point classes of an operator table by where the SOURCE lies: `outside` keeps the weights of sources that do
 *    not lie on the closure of the destination's macro-primitive (what the reference reads from ghost layers), `closure_vertex` /
 *    `closure_edge` the weights of vertex- / edge-DoF sources on it.  Inner rows and element matrices are zero in all three:
 *    such tables serve the boundary classes at levels >= 2.
 *  - hyteg_hip_p2_operator_table_face_edge_weights: this cell's share of the couplings between the edge DoFs INSIDE one macro-face,
 *    in the face's own frame -- face_verts[3] = the cell-local vertex ids of the face's vertices in the order of their global ids
 *    (x runs from the first to the second, y from the first to the third); w[3][5]: for the face types X, XY, Y the diagonal
 *    and the four neighbours  X: XY(i,j) Y(i,j) XY(i,j-1) Y(i+1,j-1);  XY: X(i,j) Y(i,j) X(i,j+1) Y(i+1,j);
 *    Y: X(i,j) XY(i,j) X(i-1,j+1) XY(i-1,j)  (the other edges of the two face triangles that share the edge).
 *  - hyteg_hip_p2_sor_face_edgedofs_cell: sor_3D_macroface_P2_update_edgedofs[_backwards] on this cell's copy of its faces in
 *    `mask` (bits 6..9): rows ascending, x ascending, X, XY, Y at every index, each in place,
 *        u_i = (1 - relax) u_i + relax / w_ii ( q_i - sum_{in-face neighbours} w_ij u_j ),
 *    q = rhs - (everything that does not change during the sweep), face_w[4][3][5] = the TOTAL weights (sum of the shares of
 *    the cells at the face), face_verts[4][3] as above per face.  Every cell at a face runs the same sweep on its own copy.
 *    Levels 2..9. */
HYTEG_HIP_API int hyteg_hip_p2_operator_table_closure_split( const double* table_host, double* outside, double* closure_vertex, double* closure_edge );
HYTEG_HIP_API int hyteg_hip_p2_operator_table_face_edge_weights( const double* table_host, const int* face_verts /* 3 */, double* w /* 15 */ );
/* the same for up to HYTEG_HIP_MAX_BATCH macro-cells in one launch (levels 2..6): hyteg_hip_p2_sor_face_frames turns a cell's face_verts[4][3]
 * and face_w[4][3][5] into hyteg_hip_p2_sor_face_frames_bytes() bytes (host; a multiple of 8); the caller keeps the records of its cells, one
 * after the other, in device memory (frames_dev) */
HYTEG_HIP_API size_t hyteg_hip_p2_sor_face_frames_bytes( void );
HYTEG_HIP_API int    hyteg_hip_p2_sor_face_frames( int level, const int* face_verts /* 12 */, const double* face_w /* 60 */, void* frames_host );
HYTEG_HIP_API int    hyteg_hip_p2_sor_face_edgedofs_cells( int ncells, double* const* dst_edge, const double* const* q_edge, int level, const void* frames_dev,
                                                           double relax, const unsigned* masks, int backwards, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_p2_sor_face_edgedofs_cell( double* dst_edge, const double* q_edge, int level, const int* face_verts /* 12 */,
                                                       const double* face_w /* 60 */, double relax, unsigned mask, int backwards,
                                                       hyteg_hip_stream_t stream );

/* a10: additive exchange of shared points (the reduce-into-owner of VertexDoFAdditivePackInfo.hpp:676-745,
 * followed by the copy back into every adjacent cell).  A group is one physical DoF; its entries are the
 * places that hold a partial value of it: (buffer index into `bases`, element offset).
 *   out[e] = init + sum_{e in group} bases[buf_e][off_e]   in entry order (fixed, so every rank gets the same bits),
 * written back to every entry whose buffer index is < n_writable (local cell arrays; receive buffers follow).
 * All arrays are device memory except the scalars. */
HYTEG_HIP_API int hyteg_hip_sum_shared( double* const*     bases /* device table of device pointers */,
                                        const int*         group_ptr /* device, ngroups+1 */,
                                        const int*         entry_buf /* device, nentries */,
                                        const int*         entry_off /* device, nentries */,
                                        int                ngroups,
                                        int                n_writable,
                                        hyteg_hip_stream_t stream );
/* same groups, but every writable entry := the value of the group's FIRST entry (owner copy wins); used after
 * host-side interpolation, where the copies of a shared DoF may differ in the last bit */
HYTEG_HIP_API int hyteg_hip_copy_shared( double* const*     bases,
                                         const int*         group_ptr,
                                         const int*         entry_buf,
                                         const int*         entry_off,
                                         int                ngroups,
                                         int                n_writable,
                                         hyteg_hip_stream_t stream );
/* pack: out[k] = bases[buf[k]][off[k]]  (send buffer for one peer) */
HYTEG_HIP_API int hyteg_hip_gather_entries( double*            out,
                                            double* const*     bases,
                                            const int*         entry_buf,
                                            const int*         entry_off,
                                            int                n,
                                            hyteg_hip_stream_t stream );

/* ---- a9 / a10 in HyTeG's own macro-face layout (drop-in for the face seam) -----------------------------------
 * Macro-face array of width N = 2^level + 1:  [ tri(N) face DoFs | tri(N-1) ghost layer of neighbour cell 0 |
 * tri(N-1) ghost layer of neighbour cell 1 ]  (src/hyteg/p1functionspace/VertexDoFMemory.hpp:59-63,
 * VertexDoFIndexing.cpp:219-226).  v0,v1,v2 = cell.getFaceLocalVertexToCellLocalVertexMaps()[localFaceID], the
 * arguments of the reference's generated copy kernels.
 *
 * copy_face_to_cell replaces vertexdof::comm::generated::communicate_directly_vertexdof_face_to_cell
 *   (src/hyteg/p1functionspace/generatedKernels/communicate_directly_vertexdof_face_to_cell*.cpp; caller
 *    VertexDoFPackInfo.hpp:438-478): all tri(N) face DoFs onto the cell's boundary layer.
 * copy_cell_to_face replaces communicate_directly_vertexdof_cell_to_face (caller VertexDoFPackInfo.hpp:552-615):
 *   the cell layer at distance 1 from the face into ghost layer `neighbor` (the face's cell_index of that cell).
 * apply_face3d replaces apply_3D_macroface_one_sided_vertexdof_to_vertexdof_{replace,add} called once per
 *   neighbour cell (src/constant_stencil_operator/P1ConstantOperator.cpp:239-357; spec P1Operator.hpp:1111-1173):
 *   dst_i (=|+=) sum over the ncells (1 or 2) neighbour cells of that cell's share of the stencil.
 *   vmaps: host, ncells x 3; w: host, ncells x 15 weights in the CELL's stencil directions (w[15] order; entries
 *   for directions that leave the cell are ignored) -- faceStencil3D[cell] of the reference. */
HYTEG_HIP_API int hyteg_hip_p1_copy_face_to_cell( double* cell, const double* face, int level, int v0, int v1, int v2, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_p1_copy_cell_to_face( double* face, const double* cell, int level, int v0, int v1, int v2, int neighbor,
                                                  hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_p1_apply_face3d( double*            dst_face,
                                             const double*      src_face,
                                             int                level,
                                             int                ncells,
                                             const int*         vmaps /* host, ncells*3 */,
                                             const double*      w /* host, ncells*15 */,
                                             int                update,
                                             hyteg_hip_stream_t stream );
/* sor_face3d replaces vertexdof::macroface::generated::sor_3D_macroface_P1 / _one_sided and their _backwards variants
 *   (src/constant_stencil_operator/P1generatedKernels/sor_3D_macroface_P1*.cpp; call site P1ConstantOperator.cpp:432-571;
 *   the loop they unroll: P1Operator::smooth_sor_face3D, P1Operator.hpp:1424-1503): in-place SOR / Gauss-Seidel sweep over
 *   the inner DoFs of ONE macro-face in lexicographic (y, x) order (backwards: reversed), ghost layers read as they are.
 *   dst_face: face array with ncells ghost layers; rhs_face: at least the tri(N) face DoFs; vmaps, w as for apply_face3d
 *   (w[7] of every cell = its share of the centre weight).  work: hyteg_hip_p1_sor_face3d_workspace( level ) bytes of
 *   device memory (the sweep is split into a parallel preparation pass and a wavefront pass; the values in between live
 *   there).  The updates are made in the reference's order; the terms of one update are summed in a different order
 *   (tests compare at 1e-12). */
HYTEG_HIP_API size_t hyteg_hip_p1_sor_face3d_workspace( int level );
HYTEG_HIP_API int hyteg_hip_p1_sor_face3d( double*            dst_face,
                                           const double*      rhs_face,
                                           double*            work,
                                           int                level,
                                           int                ncells,
                                           const int*         vmaps /* host, ncells*3 */,
                                           const double*      w /* host, ncells*15 */,
                                           double             relax,
                                           int                backwards,
                                           hyteg_hip_stream_t stream );

/* ---- f1: quadratic (P2) grid transfer on one macro-cell -----------------------------------------------------------
 * replaces P2toP2QuadraticProlongation::prolongateAdditively3D and P2toP2QuadraticRestriction::restrictAdditively3D
 *   src/hyteg/gridtransferoperators/P2toP2QuadraticProlongation.cpp:217-424,
 *   generatedKernels/prolongate_3D_macrocell_P2_push_from_{vertexdofs,edgedofs}.cpp
 *   src/hyteg/gridtransferoperators/P2toP2QuadraticRestriction.cpp:131-286,
 *   generatedKernels/restrict_3D_macrocell_P2_update_{vertexdofs,edgedofs}.cpp
 * vertex arrays: macro-cell layout of width 2^level + 1; edge arrays: hyteg_hip_p2_edge_array_size( level ) entries, seven
 * orientation blocks (X, Y, Z, XY, XZ, YZ, XYZ).  mask: 15-bit point-class mask of the DESTINATION DoFs (bits 0..13: the
 * macro-primitive slots edge0..5, face0..3, vertex0..3; bit 14: inside the cell), as for the elementwise apply.
 * prolongate: fine = (update == ADD ? fine : 0) + quadratic interpolant of the coarse function at the fine DoFs.  Every
 *   cell computes the complete value of the fine DoFs on its boundary (the interpolant is continuous), so the copies of
 *   several cells agree up to rounding; the host layer makes them bit-identical with a copy, not a sum.
 * restrict: coarse = P^T fine, where fine DoFs on a macro-primitive shared by nnc[slot] cells are scaled by 1 / nnc[slot]
 *   (nnc: host, 14 values in slot order, the numNeighborCells* arguments of the reference kernels); the additive exchange
 *   over the cells completes the coarse DoFs on shared primitives. */
HYTEG_HIP_API int hyteg_hip_p2_prolongate_cell( double*            fine_vertex,
                                                double*            fine_edge,
                                                const double*      coarse_vertex,
                                                const double*      coarse_edge,
                                                int                coarse_level,
                                                int                update,
                                                unsigned           mask,
                                                hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_p2_restrict_cell( double*            coarse_vertex,
                                              double*            coarse_edge,
                                              const double*      fine_vertex,
                                              const double*      fine_edge,
                                              int                coarse_level,
                                              const double*      nnc /* host, 14 */,
                                              unsigned           mask,
                                              hyteg_hip_stream_t stream );

/* ---- events and the neighbour exchange over RCCL (xGMI) --------------------------------------------------------
 * One process per GPU.  Replaces, on this path, what the reference does through waLBerla's BufferSystem over MPI:
 *   src/hyteg/communication/BufferedCommunication.cpp:181-470   start/endCommunication: pack, Isend/Irecv, wait, unpack
 *   src/hyteg/p1functionspace/VertexDoFAdditivePackInfo.hpp:676-745   payloads of the additive exchange
 *   src/hyteg/p1functionspace/VertexDoFFunction.cpp:1710-1717   dotGlobal: allReduceInplace( SUM ) of one scalar
 * The message pattern is sparse neighbour point-to-point: hyteg_hip_comm_exchange is ONE group of ncclSend / ncclRecv
 * pairs (one per peer rank) enqueued on `stream`, no host synchronisation, no staging; the caller orders it against its
 * pack / reduce kernels with events.  librccl is resolved at run time (a copy the process already holds -- PyTorch
 * ships and loads its own -- is reused; HYTEG_HIP_RCCL_LIB overrides the search). */
HYTEG_HIP_API int hyteg_hip_event_create( hyteg_hip_event_t* event );
/* events that carry a device timestamp: elapsed_ms waits for `stop` and returns the time between the two (hipEventElapsedTime) */
HYTEG_HIP_API int hyteg_hip_event_create_timing( hyteg_hip_event_t* event );
HYTEG_HIP_API int hyteg_hip_event_elapsed_ms( hyteg_hip_event_t start, hyteg_hip_event_t stop, float* ms );
HYTEG_HIP_API int hyteg_hip_event_destroy( hyteg_hip_event_t event );
HYTEG_HIP_API int hyteg_hip_event_record( hyteg_hip_event_t event, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_stream_wait_event( hyteg_hip_stream_t stream, hyteg_hip_event_t event );

#define HYTEG_HIP_COMM_ID_BYTES 128 /* sizeof( ncclUniqueId ) */
typedef void* hyteg_hip_comm_t;
/* HYTEG_HIP_OK if librccl could be resolved; origin (optional) receives where it came from and its version code */
HYTEG_HIP_API int hyteg_hip_comm_available( char* origin, size_t buflen );
/* rank 0 creates the id and distributes its HYTEG_HIP_COMM_ID_BYTES bytes to all ranks by whatever means the
 * application has (MPI_Bcast in HyTeG, a torch.distributed store in bench.py) */
HYTEG_HIP_API int hyteg_hip_comm_unique_id( unsigned char* id );
/* collective over all ranks; binds the communicator to the calling thread's current device */
HYTEG_HIP_API int hyteg_hip_comm_create( hyteg_hip_comm_t* comm, int nranks, int rank, const unsigned char* id );
HYTEG_HIP_API int hyteg_hip_comm_destroy( hyteg_hip_comm_t comm );
/* send[ sum of send_count[0..k) ... ) goes to peers[k], recv[ sum of recv_count[0..k) ... ) comes from peers[k]
 * (device buffers, counts in doubles, host arrays of length npeers); stream-ordered.  A peer may be the calling rank
 * itself (loop-back inside the group; what the single-GPU tests use). */
HYTEG_HIP_API int hyteg_hip_comm_exchange( hyteg_hip_comm_t   comm,
                                           int                npeers,
                                           const int*         peers,
                                           const double*      send,
                                           const int*         send_count,
                                           double*            recv,
                                           const int*         recv_count,
                                           hyteg_hip_stream_t stream );
/* in-place sum over all ranks of n doubles in device memory; stream-ordered */
HYTEG_HIP_API int hyteg_hip_comm_allreduce_sum( hyteg_hip_comm_t comm, double* values, int n, hyteg_hip_stream_t stream );

/* ---- the same exchange peer to peer: stores into IPC-mapped arenas of the neighbour GPUs (comm_p2p.hip) -----------
 * Same reference semantics as hyteg_hip_comm_exchange (BufferedCommunication.cpp:181-470), no library call per exchange:
 * every rank creates one arena (uncached device memory), hands its HYTEG_HIP_P2P_HANDLE_BYTES handle to the other ranks
 * of the node (any channel; the handle is a plain byte string) and maps theirs.  Inside its arena a rank lays out, per
 * exchange plan and peer, two receive slots (used alternately, by sequence parity) and one 8-byte flag word; it tells
 * each peer where they are, and the peer describes them, as pointers into ITS mapping of that arena, in a
 * hyteg_hip_p2p_peer_t.  hyteg_hip_p2p_pack gathers n values like hyteg_hip_gather_entries and stores value k into slot
 * [seq & 1] of the peer whose segment [start, start + count) holds k, then writes seq to every peer's flag word;
 * hyteg_hip_p2p_wait enqueues a kernel that returns once the npeers flag words at flags[p * stride] (this rank's own
 * arena) have reached seq, or sets *status to 1 + p after timeout_ms (0: 20 s) -- it never spins unbounded.
 * Sequence numbers start at 1 and grow by one per exchange of a plan on both sides. */
#define HYTEG_HIP_P2P_HANDLE_BYTES 64 /* sizeof( hipIpcMemHandle_t ) */
typedef struct
{
   double*             slot[2]; /* receive slots of this rank's segment in the peer's arena (mapped here) */
   unsigned long long* flag;    /* this rank's flag word in the peer's arena (mapped here) */
   int                 start;   /* first index of the peer's segment in the send enumeration */
   int                 count;
} hyteg_hip_p2p_peer_t;
/* kind (optional) receives 0 uncached / 1 fine-grained / 2 default memory (env HYTEG_HIP_P2P_ARENA; default uncached) */
HYTEG_HIP_API int hyteg_hip_p2p_arena_create( size_t bytes, void** base, unsigned char* handle, int* kind );
HYTEG_HIP_API int hyteg_hip_p2p_arena_destroy( void* base );
HYTEG_HIP_API int hyteg_hip_p2p_arena_open( const unsigned char* handle, void** mapped );
HYTEG_HIP_API int hyteg_hip_p2p_arena_close( void* mapped );
/* peers: device array of npeers descriptors; counter: device word, zero before the first call, owned by the plan */
HYTEG_HIP_API int hyteg_hip_p2p_pack( const hyteg_hip_p2p_peer_t* peers,
                                      int                         npeers,
                                      double* const*              bases,
                                      const int*                  entry_buf,
                                      const int*                  entry_off,
                                      int                         n,
                                      unsigned long long          seq,
                                      unsigned*                   counter,
                                      hyteg_hip_stream_t          stream );
/* hyteg_hip_p1_apply_cell_boundary + hyteg_hip_p2p_pack in one launch, for a rank that owns ONE macro-cell: a shell point q
 * (the enumeration of the boundary kernel: q in [0, 4 tri(N)), face by face, row by row) whose share other ranks need stores it
 * into the peers' slots as entries send_list[ send_first[q] .. send_first[q+1] ) of the plan's send enumeration; the last
 * workgroup publishes seq.  send_first has 4 tri(N) + 1 entries.  Same values, bit for bit, as the two calls. */
HYTEG_HIP_API int hyteg_hip_p1_apply_cell_boundary_p2p( double*                     dst,
                                                        const double*               src,
                                                        int                         level,
                                                        const double*               w_slots /* host, 14 x 15 */,
                                                        unsigned                    mask,
                                                        int                         update,
                                                        const int*                  send_first,
                                                        const int*                  send_list,
                                                        const hyteg_hip_p2p_peer_t* peers,
                                                        int                         npeers,
                                                        unsigned long long          seq,
                                                        unsigned*                   counter,
                                                        hyteg_hip_stream_t          stream );
/* hyteg_hip_sum_shared ( additive != 0 ) / hyteg_hip_copy_shared whose workgroups first wait like hyteg_hip_p2p_wait: the
 * wait kernel and the reduce kernel of an exchange in one launch */
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
                                                     hyteg_hip_stream_t        stream );
HYTEG_HIP_API int hyteg_hip_p2p_wait( const unsigned long long* flags,
                                      int                       npeers,
                                      int                       stride,
                                      unsigned long long        seq,
                                      unsigned*                 status,
                                      unsigned                  timeout_ms,
                                      hyteg_hip_stream_t        stream );

/* ---- f4: the constant-stencil P2 operator at its kernel seam -------------------------------------------------------------
 * P2ConstantOperator::apply (src/constant_stencil_operator/P2ConstantOperator.cpp:100-112) = four sub-operators whose
 * macro-cell kernels take stencil MAPS and update the macro-cell's INNER DoFs:
 *   vertex->vertex  hyteg_hip_p1_apply_cell (above)
 *   edge->vertex    apply_3D_macrocell_edgedof_to_vertexdof_{replace,add}( src X, XY, XYZ, XZ, Y, YZ, Z, dst vertex, e2vStencilMap, level )
 *                   src/hyteg/mixedoperators/EdgeDoFToVertexDoFOperator/generatedKernels/apply_3D_macrocell_edgedof_to_vertexdof_replace.hpp:36
 *   vertex->edge    apply_3D_macrocell_vertexdof_to_edgedof_{replace,add}( dst X, ..., Z, src vertex, level, v2eStencilMap )
 *                   src/hyteg/mixedoperators/VertexDoFToEdgeDoFOperator/generatedKernels/apply_3D_macrocell_vertexdof_to_edgedof_replace.hpp:36
 *   edge->edge      apply_3D_macrocell_edgedof_to_edgedof_{replace,add}( dst X, ..., Z, src X, ..., Z, e2eStencilMap, level )
 *                   src/constant_stencil_operator/EdgeDoFGeneratedKernels/apply_3D_macrocell_edgedof_to_edgedof_replace.hpp:37
 * The entry points keep the reference's pointer lists (the seven pointers must be the blocks of ONE edge-DoF array, which is
 * what the reference passes: &data[ index( level, 0, 0, 0, orientation ) ]) and take the stencil map's VALUES flattened in the
 * map's own iteration order -- orientations in the order of the enum (X, Y, Z, XY, XZ, YZ, XYZ), offsets in indexing::Index
 * order (z, y, x):
 *     for ( auto& a : e2eStencilMap ) for ( auto& b : a.second ) for ( auto& c : b.second ) w[k++] = c.second;
 * hyteg_hip_p2_constant_stencil_layout gives the number of values of the four maps (counts[4]: v2v, e2v, v2e, e2e) and, if
 * keys != NULL, for every value of the concatenation v2v | e2v | v2e | e2e the five integers { destination kind, source kind,
 * dx, dy, dz } (kind 0 = vertex DoFs, 1..7 = edge DoFs X, Y, Z, XY, XZ, YZ, XYZ; offset = source index - destination index),
 * so that a binding can check its maps' keys.  Levels 2..9.  These three calls run the fused kernel with all other weights
 * zero; the fast path is ONE pass for all four sub-operators: hyteg_hip_p2_build_operator_table_from_stencils turns the
 * concatenated values (inner DoFs) and, optionally, the cell's share stencils of the 14 boundary point classes
 * (classes[14][total], zero where a neighbour lies outside the cell) into the table hyteg_hip_p2_elementwise_apply_cell takes. */
HYTEG_HIP_API int hyteg_hip_p2_constant_stencil_layout( int* counts /* 4 */, int* keys /* NULL or 5 * total */ );
HYTEG_HIP_API int hyteg_hip_p2_build_operator_table_from_stencils( const double* inner /* total */, const double* classes /* NULL or 14 * total */,
                                                                   double* table_host );
HYTEG_HIP_API int hyteg_hip_p2_apply_cell_edgedof_to_vertexdof( const double* src_x, const double* src_xy, const double* src_xyz, const double* src_xz,
                                                                const double* src_y, const double* src_yz, const double* src_z, double* dst_vertex,
                                                                const double* e2v_stencil, int level, int update, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_p2_apply_cell_vertexdof_to_edgedof( double* dst_x, double* dst_xy, double* dst_xyz, double* dst_xz, double* dst_y,
                                                                double* dst_yz, double* dst_z, const double* src_vertex, int level,
                                                                const double* v2e_stencil, int update, hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_p2_apply_cell_edgedof_to_edgedof( double* dst_x, double* dst_xy, double* dst_xyz, double* dst_xz, double* dst_y, double* dst_yz,
                                                              double* dst_z, const double* src_x, const double* src_xy, const double* src_xyz,
                                                              const double* src_xz, const double* src_y, const double* src_yz, const double* src_z,
                                                              const double* e2e_stencil, int level, int update, hyteg_hip_stream_t stream );

/* ---- b-3: the seam of the generated elementwise operators (module hyteg_operators) ---------------------------------------
 * The generated operators call, per macro-cell,
 *    apply_macro_3D( dst*, src*, macro_vertex_coord_id_{0..3}comp{0..2} (12 scalars), int64 micro_edges_per_macro_edge,
 *                    (value type) micro_edges_per_macro_edge_float )
 * (apps/2023-zikeli-mt/MT-apps/operators-used/P1ElementwiseDiffusion_cubes_const_float64.hpp:95-130, kernel .cpp:782-;
 * call site .cpp:76-165: halo synchronised and dst zeroed before, additive communication afterwards).  The entry points
 * below have that argument list (the twelve coordinates as one array [vertex][component]) and that effect:
 *    dst += ( operator of THIS macro-cell ) src   at ALL points of the cell array,
 * i.e. the full stencil at inner points and this cell's share at points on its macro-faces / -edges / -vertices.
 * The element matrices are computed on the host from the coordinates (constant per micro-cell type on an affine cell) and
 * summed into the constant stencils of the 15 point classes; results agree with the reference's element-by-element
 * scatter to rounding (the reference's own criterion for constant-stencil vs elementwise: < 1e-13,
 * tests/hyteg/convergence/P1JacobiConvergenceTest.cpp:117).  micro_edges_per_macro_edge = 2^level, levels 0..11
 * (float: 0..10). */
HYTEG_HIP_API int hyteg_hip_p1_elementwise_diffusion_apply_macro_3d( double*            dst,
                                                                     const double*      src,
                                                                     const double*      macro_vertex_coords /* 12: [vertex][component] */,
                                                                     int64_t            micro_edges_per_macro_edge,
                                                                     double             micro_edges_per_macro_edge_float,
                                                                     hyteg_hip_stream_t stream );
/* The same kernel restricted to the point classes of `mask` (as hyteg_hip_p1_apply_cell_boundary: bit k < 14 = points on slot k,
 * HYTEG_HIP_MASK_INNER = inner points) with REPLACE or ADD: for callers whose cell arrays hold the shared DoFs themselves (the
 * host layer here) instead of a halo that may be zeroed.  ( HYTEG_HIP_MASK_ALL, HYTEG_HIP_ADD ) is the call above. */
HYTEG_HIP_API int hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_masked( double*            dst,
                                                                            const double*      src,
                                                                            const double*      macro_vertex_coords,
                                                                            int64_t            micro_edges_per_macro_edge,
                                                                            unsigned           mask,
                                                                            int                update,
                                                                            hyteg_hip_stream_t stream );
/* float32 sibling (P1ElementwiseDiffusion_cubes_const_float32.hpp); stencils are summed in double and rounded to float */
HYTEG_HIP_API int hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_f32( float*             dst,
                                                                         const float*       src,
                                                                         const float*       macro_vertex_coords,
                                                                         int64_t            micro_edges_per_macro_edge,
                                                                         float              micro_edges_per_macro_edge_float,
                                                                         hyteg_hip_stream_t stream );
/* computeInverseDiagonalOperatorValues_macro_3D of the same operators: diag += the diagonal entries of the element matrices
 * of the adjacent micro-cells at all points of the cell array (levels 0..10) */
HYTEG_HIP_API int hyteg_hip_p1_elementwise_diffusion_diagonal_macro_3d( double*            diag,
                                                                        const double*      macro_vertex_coords,
                                                                        int64_t            micro_edges_per_macro_edge,
                                                                        double             micro_edges_per_macro_edge_float,
                                                                        hyteg_hip_stream_t stream );
/* host helper: the stencils the calls above use -- w_inner[15] and w_slots[14][15] in the conventions of
 * hyteg_hip_p1_apply_cell / hyteg_hip_p1_apply_cell_boundary */
HYTEG_HIP_API int hyteg_hip_p1_elementwise_diffusion_stencils( const double* macro_vertex_coords,
                                                               int64_t       micro_edges_per_macro_edge,
                                                               double*       w_inner,
                                                               double*       w_slots );
/* float instantiation of hyteg_hip_p1_apply_cell_boundary (used by the float seam above) */
HYTEG_HIP_API int hyteg_hip_p1_apply_cell_boundary_f32( float*             dst,
                                                        const float*       src,
                                                        int                level,
                                                        const double*      w_slots,
                                                        unsigned           mask,
                                                        int                update,
                                                        hyteg_hip_stream_t stream );
/* P2 form of the seam (generated P2ElementwiseDiffusion; its source is absent from the snapshot -- call sites
 * src/hyteg_operators_composites/viscousblock/P2ViscousBlockLaplaceOperator.hpp:29,66 -- so the ORDER of the four array
 * arguments is this header's convention, vertex before edge: parity unpinned for the argument order).
 * dst += A_cell src on all vertex and edge DoFs of the macro-cell; levels 0..9.  The six 10 x 10 element matrices
 * (FEniCS ordering, micro-cell types in the order of celldof::allCellTypes) are available through the second call. */
HYTEG_HIP_API int hyteg_hip_p2_elementwise_diffusion_apply_macro_3d( double*            dst_vertex,
                                                                     double*            dst_edge,
                                                                     const double*      src_vertex,
                                                                     const double*      src_edge,
                                                                     const double*      macro_vertex_coords,
                                                                     int64_t            micro_edges_per_macro_edge,
                                                                     double             micro_edges_per_macro_edge_float,
                                                                     hyteg_hip_stream_t stream );
HYTEG_HIP_API int hyteg_hip_p2_elementwise_diffusion_element_matrices( const double* macro_vertex_coords,
                                                                       int64_t       micro_edges_per_macro_edge,
                                                                       double*       elmat /* 600 */ );

#ifdef __cplusplus
}
#endif
#endif /* HYTEG_HIP_H */
