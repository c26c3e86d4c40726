/*
 * p1_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the HyTeG algorithms on the P1 matrix-free
 * hot path.  It is the checker the HIP kernels are compared against.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (hyteg_amd/, include/) never links, imports or calls anything in oracle/.
 *
 * Every function cites the reference file:line (relative to /root/reference/) it restates.
 * Nothing here is copied from the reference: the generated kernels there spell out the index
 * polynomial per access; here the same arithmetic is expressed through cell_index().
 *
 * Pinning: the reference ships no golden vectors for this path.  The oracle is pinned by the
 * reference's own known-answer tests and properties (tests/test_oracle_pins.py): index tables
 * (tests/hyteg/Indexing/VertexDoFMacroCellIndexingTest.cpp:61-108), array sizes
 * (tests/hyteg/Indexing/CommonIndexingTest.cpp:166-170), Laplace annihilates constants/linears
 * (tests/hyteg/P1/P1LaplaceOperator3DTest.cpp), stencil row-sum/halving
 * (tests/hyteg/vertexdofspace/VertexDoFStencilAssemblyTest.cpp), prolongation exact on linears
 * (tests/hyteg/vertexdofspace/VertexDoFLinearProlongation3DTest.cpp), and the FEniCS element
 * matrix compiled in place from the reference (oracle/_ref, see oracle/Makefile).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared  (contraction off: the summation order below
 * is the reference's, so results are reproducible across compilers).
 *
 * Stencil weight order w[15] (the C-ABI order, include/hyteg_hip.h): iteration order of the
 * reference's std::map<indexing::Index,real_t>, i.e. sorted by (z, y, x)
 * (src/hyteg/indexing/Common.hpp:67-71):
 *   0:( 0, 0,-1) 1:( 1, 0,-1) 2:(-1, 1,-1) 3:( 0, 1,-1)
 *   4:( 0,-1, 0) 5:( 1,-1, 0) 6:(-1, 0, 0) 7:( 0, 0, 0) 8:( 1, 0, 0) 9:(-1, 1, 0) 10:( 0, 1, 0)
 *  11:( 0,-1, 1) 12:( 1,-1, 1) 13:(-1, 0, 1) 14:( 0, 0, 1)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HO_API __attribute__( ( visibility( "default" ) ) )

/* ---- stencil slots (see header comment) ---- */
enum
{
   W_BC  = 0,  /* ( 0, 0,-1) */
   W_BE  = 1,  /* ( 1, 0,-1) */
   W_BNW = 2,  /* (-1, 1,-1) */
   W_BN  = 3,  /* ( 0, 1,-1) */
   W_S   = 4,  /* ( 0,-1, 0) */
   W_SE  = 5,  /* ( 1,-1, 0) */
   W_W   = 6,  /* (-1, 0, 0) */
   W_C   = 7,  /* ( 0, 0, 0) */
   W_E   = 8,  /* ( 1, 0, 0) */
   W_NW  = 9,  /* (-1, 1, 0) */
   W_N   = 10, /* ( 0, 1, 0) */
   W_TS  = 11, /* ( 0,-1, 1) */
   W_TSE = 12, /* ( 1,-1, 1) */
   W_TW  = 13, /* (-1, 0, 1) */
   W_TC  = 14  /* ( 0, 0, 1) */
};

static const int OFFS[15][3] = { { 0, 0, -1 }, { 1, 0, -1 }, { -1, 1, -1 }, { 0, 1, -1 }, { 0, -1, 0 },
                                 { 1, -1, 0 }, { -1, 0, 0 }, { 0, 0, 0 },   { 1, 0, 0 },  { -1, 1, 0 },
                                 { 0, 1, 0 },  { 0, -1, 1 }, { 1, -1, 1 },  { -1, 0, 1 }, { 0, 0, 1 } };

/* ---------------------------------------------------------------------------------------------
 * Layout.  src/hyteg/indexing/MacroCellIndexing.hpp:40-52 (linearMacroCellSize / -Index),
 * src/hyteg/Levelinfo.hpp:36-115 (num_microvertices_per_edge = 2^level + 1).
 * ------------------------------------------------------------------------------------------- */
static inline int64_t tet_size( int64_t width ) { return ( ( width + 2 ) * ( width + 1 ) * width ) / 6; }

static inline int64_t cell_index_w( int64_t width, int64_t x, int64_t y, int64_t z )
{
   const int64_t wms         = width - z;
   const int64_t sliceOffset = tet_size( width ) - tet_size( wms );
   const int64_t rowOffset   = y * ( wms + 1 ) - ( ( y + 1 ) * y ) / 2;
   return sliceOffset + rowOffset + x;
}

HO_API int64_t ho_width( int level ) { return ( (int64_t) 1 << level ) + 1; }
HO_API int64_t ho_cell_size( int level ) { return tet_size( ho_width( level ) ); }
HO_API int64_t ho_cell_index( int level, int x, int y, int z ) { return cell_index_w( ho_width( level ), x, y, z ); }
/* number of points the interior loops visit: C(2^L-1,3); CellIterator(level,1),
 * src/hyteg/indexing/MacroCellIndexing.cpp:92-104 */
HO_API int64_t ho_cell_inner_size( int level )
{
   const int64_t n = ( (int64_t) 1 << level ) - 1;
   return n < 3 ? 0 : n * ( n - 1 ) * ( n - 2 ) / 6;
}
/* macro-face (triangle) layout: src/hyteg/indexing/MacroFaceIndexing.hpp:40-50 */
HO_API int64_t ho_face_size_w( int64_t width ) { return ( ( width + 1 ) * width ) / 2; }
HO_API int64_t ho_face_index_w( int64_t width, int64_t x, int64_t y )
{
   return y * ( width + 1 ) - ( ( y + 1 ) * y ) / 2 + x;
}

/* ---------------------------------------------------------------------------------------------
 * a2: 15-point constant-stencil apply on the cell interior.
 * src/constant_stencil_operator/P1generatedKernels/apply_3D_macrocell_vertexdof_to_vertexdof_replace.cpp:34-78
 * (sum order xi_18+...+xi_32 at :73) and ..._add.cpp (same products, old dst value added LAST).
 * update: 0 = Replace, 1 = Add  (src/hyteg/types/types.hpp:29-33).
 * ------------------------------------------------------------------------------------------- */
HO_API void ho_apply_cell( double* dst, const double* src, int level, const double* w, int update )
{
   const int64_t N = ho_width( level );
   const int     n = 1 << level;
   for ( int z = 1; z < n; ++z )
      for ( int y = 1; y < n - z; ++y )
         for ( int x = 1; x < n - y - z; ++x )
         {
#define S( dx, dy, dz ) src[cell_index_w( N, x + ( dx ), y + ( dy ), z + ( dz ) )]
            /* the reference's summation order (replace.cpp:73): */
            double acc = w[W_W] * S( -1, 0, 0 );
            acc        = acc + w[W_BN] * S( 0, 1, -1 );
            acc        = acc + w[W_N] * S( 0, 1, 0 );
            acc        = acc + w[W_SE] * S( 1, -1, 0 );
            acc        = acc + w[W_TSE] * S( 1, -1, 1 );
            acc        = acc + w[W_BE] * S( 1, 0, -1 );
            acc        = acc + w[W_E] * S( 1, 0, 0 );
            acc        = acc + w[W_TW] * S( -1, 0, 1 );
            acc        = acc + w[W_BNW] * S( -1, 1, -1 );
            acc        = acc + w[W_NW] * S( -1, 1, 0 );
            acc        = acc + w[W_S] * S( 0, -1, 0 );
            acc        = acc + w[W_TS] * S( 0, -1, 1 );
            acc        = acc + w[W_BC] * S( 0, 0, -1 );
            acc        = acc + w[W_C] * S( 0, 0, 0 );
            acc        = acc + w[W_TC] * S( 0, 0, 1 );
#undef S
            const int64_t i = cell_index_w( N, x, y, z );
            dst[i]          = update ? acc + dst[i] : acc;
         }
}

/* float32 instantiation of the same kernel (apply_3D_macrocell_vertexdof_to_vertexdof_replace.cpp:96-97 instantiates the
 * template for walberla::float32; the stencil map then holds floats): same term order, float arithmetic.  The weights
 * arrive as doubles and are rounded to float first, as a std::map< Index, float > filled from real_t values would hold them. */
HO_API void ho_apply_cell_f32( float* dst, const float* src, int level, const double* wd, int update )
{
   const int64_t N = ho_width( level );
   const int     n = 1 << level;
   float         w[15];
   for ( int k = 0; k < 15; ++k )
      w[k] = (float) wd[k];
   for ( int z = 1; z < n; ++z )
      for ( int y = 1; y < n - z; ++y )
         for ( int x = 1; x < n - y - z; ++x )
         {
#define S( dx, dy, dz ) src[cell_index_w( N, x + ( dx ), y + ( dy ), z + ( dz ) )]
            float acc = w[W_W] * S( -1, 0, 0 );
            acc       = acc + w[W_BN] * S( 0, 1, -1 );
            acc       = acc + w[W_N] * S( 0, 1, 0 );
            acc       = acc + w[W_SE] * S( 1, -1, 0 );
            acc       = acc + w[W_TSE] * S( 1, -1, 1 );
            acc       = acc + w[W_BE] * S( 1, 0, -1 );
            acc       = acc + w[W_E] * S( 1, 0, 0 );
            acc       = acc + w[W_TW] * S( -1, 0, 1 );
            acc       = acc + w[W_BNW] * S( -1, 1, -1 );
            acc       = acc + w[W_NW] * S( -1, 1, 0 );
            acc       = acc + w[W_S] * S( 0, -1, 0 );
            acc       = acc + w[W_TS] * S( 0, -1, 1 );
            acc       = acc + w[W_BC] * S( 0, 0, -1 );
            acc       = acc + w[W_C] * S( 0, 0, 0 );
            acc       = acc + w[W_TC] * S( 0, 0, 1 );
#undef S
            const int64_t i = cell_index_w( N, x, y, z );
            dst[i]          = update ? acc + dst[i] : acc;
         }
}
/* smooth_jac (P1Operator.hpp:429-447) in float: apply, rhs - dst, * invdiag, src + relax * ( . ) */
HO_API void ho_jacobi_cell_f32( float* dst, const float* rhs, const float* src, const float* invdiag, int level, const double* wd,
                                double relax )
{
   ho_apply_cell_f32( dst, src, level, wd, 0 );
   const int64_t N    = ho_width( level );
   const int     n    = 1 << level;
   const float   invC = (float) ( 1.0 / wd[W_C] ), rx = (float) relax;
   for ( int z = 1; z < n; ++z )
      for ( int y = 1; y < n - z; ++y )
         for ( int x = 1; x < n - y - z; ++x )
         {
            const int64_t i = cell_index_w( N, x, y, z );
            float         t = rhs[i] - dst[i];
            t               = ( invdiag ? invdiag[i] : invC ) * t;
            dst[i]          = src[i] + rx * t;
         }
}

/* ---------------------------------------------------------------------------------------------
 * a3: in-place lexicographic SOR / Gauss-Seidel sweeps on the cell interior.
 * src/constant_stencil_operator/P1generatedKernels/sor_3D_macrocell_P1.cpp:32-90 (update at :74):
 *     u_i = relax * (1/w_c) * ( -sum_14 w_k u_k + rhs_i ) + ( 1 + (-relax) ) * u_i
 * with sum order xi_21..xi_35 (rhs last); sor_3D_macrocell_P1_backwards.cpp:52-57 (reversed loops);
 * gaussseidel_3D_macrocell_P1.cpp:32-88 (u_i = (1/w_c) * ( -sum + rhs_i ), same term order, :72).
 * ------------------------------------------------------------------------------------------- */
static inline double residual_sum( const double* u, const double* rhs, const double* w, int64_t N, int x, int y, int z )
{
#define U( dx, dy, dz ) u[cell_index_w( N, x + ( dx ), y + ( dy ), z + ( dz ) )]
   /* 14 terms -(w_k) * u_k in the reference's order, then rhs LAST (sor: xi_21..xi_35; gs: xi_19..xi_33) */
   double acc = -w[W_BN] * U( 0, 1, -1 );
   acc        = acc + -w[W_N] * U( 0, 1, 0 );
   acc        = acc + -w[W_SE] * U( 1, -1, 0 );
   acc        = acc + -w[W_TSE] * U( 1, -1, 1 );
   acc        = acc + -w[W_BE] * U( 1, 0, -1 );
   acc        = acc + -w[W_E] * U( 1, 0, 0 );
   acc        = acc + -w[W_W] * U( -1, 0, 0 );
   acc        = acc + -w[W_TW] * U( -1, 0, 1 );
   acc        = acc + -w[W_BNW] * U( -1, 1, -1 );
   acc        = acc + -w[W_NW] * U( -1, 1, 0 );
   acc        = acc + -w[W_S] * U( 0, -1, 0 );
   acc        = acc + -w[W_TS] * U( 0, -1, 1 );
   acc        = acc + -w[W_BC] * U( 0, 0, -1 );
   acc        = acc + -w[W_TC] * U( 0, 0, 1 );
   acc        = acc + rhs[cell_index_w( N, x, y, z )];
#undef U
   return acc;
}

static inline double sor_point( double* u, const double* rhs, const double* w, int64_t N, int x, int y, int z, double invC,
                                double relax, double oneMinusRelax )
{
   return relax * invC * residual_sum( u, rhs, w, N, x, y, z ) + oneMinusRelax * u[cell_index_w( N, x, y, z )];
}

HO_API void ho_sor_cell( double* u, const double* rhs, int level, const double* w, double relax, int backwards )
{
   const int64_t N    = ho_width( level );
   const int     n    = 1 << level;
   const double  invC = 1 / w[W_C];
   const double  omr  = 1.0 + ( -relax );
   if ( !backwards )
   {
      for ( int z = 1; z < n; ++z )
         for ( int y = 1; y < n - z; ++y )
            for ( int x = 1; x < n - y - z; ++x )
               u[cell_index_w( N, x, y, z )] = sor_point( u, rhs, w, N, x, y, z, invC, relax, omr );
   }
   else
   {
      for ( int z = n - 1; z >= 1; --z )
         for ( int y = n - z - 1; y >= 1; --y )
            for ( int x = n - y - z - 1; x >= 1; --x )
               u[cell_index_w( N, x, y, z )] = sor_point( u, rhs, w, N, x, y, z, invC, relax, omr );
   }
}

HO_API void ho_gs_cell( double* u, const double* rhs, int level, const double* w )
{
   /* gaussseidel_3D_macrocell_P1.cpp:72: u_i = xi_17 * ( xi_19 + ... + xi_33 ), xi_17 = 1/w_c, xi_33 = rhs */
   const int64_t N    = ho_width( level );
   const int     n    = 1 << level;
   const double  invC = 1 / w[W_C];
   for ( int z = 1; z < n; ++z )
      for ( int y = 1; y < n - z; ++y )
         for ( int x = 1; x < n - y - z; ++x )
            u[cell_index_w( N, x, y, z )] = invC * residual_sum( u, rhs, w, N, x, y, z );
}

/* ---------------------------------------------------------------------------------------------
 * a6: vector kernels on the cell interior.
 * assign: src/hyteg/p1functionspace/generatedKernels/assign_3D_macrocell_vertexdof_{1,2,3}_rhsfunction(s).cpp
 *         (dispatch src/hyteg/p1functionspace/VertexDoFFunction.cpp:1088-1128); generic n-ary form
 *         src/hyteg/p1functionspace/VertexDoFMacroCell.hpp:376-402.
 * add:    VertexDoFMacroCell.hpp:456-481 (dst += sum_k c_k src_k), scalar add :441-453.
 * multElementwise: VertexDoFMacroCell.hpp:483-508.   dot: :511-529 (plain sequential +=).
 * ------------------------------------------------------------------------------------------- */
#define FOR_INNER( level )                    \
   const int64_t N = ho_width( level );       \
   const int     n = 1 << ( level );          \
   for ( int z = 1; z < n; ++z )              \
      for ( int y = 1; y < n - z; ++y )       \
         for ( int x = 1; x < n - y - z; ++x )

HO_API void ho_assign( double* dst, int nsrc, const double* const* srcs, const double* scalars, int level )
{
   FOR_INNER( level )
   {
      const int64_t i   = cell_index_w( N, x, y, z );
      double        tmp = scalars[0] * srcs[0][i];
      for ( int k = 1; k < nsrc; ++k )
         tmp = tmp + scalars[k] * srcs[k][i];
      dst[i] = tmp;
   }
}

HO_API void ho_add( double* dst, int nsrc, const double* const* srcs, const double* scalars, int level )
{
   FOR_INNER( level )
   {
      const int64_t i   = cell_index_w( N, x, y, z );
      double        tmp = scalars[0] * srcs[0][i];
      for ( int k = 1; k < nsrc; ++k )
         tmp = tmp + scalars[k] * srcs[k][i];
      dst[i] = dst[i] + tmp;
   }
}

HO_API void ho_add_scalar( double* dst, double scalar, int level )
{
   FOR_INNER( level ) { dst[cell_index_w( N, x, y, z )] += scalar; }
}

HO_API void ho_mult_elementwise( double* dst, int nsrc, const double* const* srcs, int level )
{
   FOR_INNER( level )
   {
      const int64_t i   = cell_index_w( N, x, y, z );
      double        tmp = srcs[0][i];
      for ( int k = 1; k < nsrc; ++k )
         tmp = tmp * srcs[k][i];
      dst[i] = tmp;
   }
}

HO_API double ho_dot( const double* a, const double* b, int level )
{
   double sp = 0;
   FOR_INNER( level )
   {
      const int64_t i = cell_index_w( N, x, y, z );
      sp              = sp + a[i] * b[i];
   }
   return sp;
}

/* interpolate a constant onto the cell interior: VertexDoFMacroCell.hpp:79-93 */
HO_API void ho_set_inner( double* dst, double value, int level )
{
   FOR_INNER( level ) { dst[cell_index_w( N, x, y, z )] = value; }
}

/* ---------------------------------------------------------------------------------------------
 * a4: weighted Jacobi exactly as composed by P1Operator::smooth_jac,
 * src/hyteg/p1functionspace/P1Operator.hpp:429-447:
 *     apply(src,dst) ; dst = 1*rhs + (-1)*dst ; dst = invDiag .* dst ; dst = 1*src + relax*dst
 * On a constant-stencil macro-cell invDiag is the constant 1/w_c in the interior
 * (P1Operator.hpp:636-906 copies the centre weight, then invertElementwise).  invdiag may be
 * NULL (use 1/w[7]) or a full cell array.
 * ------------------------------------------------------------------------------------------- */
HO_API void ho_jacobi_cell( double* dst, const double* rhs, const double* src, const double* invdiag, int level,
                            const double* w, double relax )
{
   ho_apply_cell( dst, src, level, w, 0 );
   const double invC = 1.0 / w[W_C];
   FOR_INNER( level )
   {
      const int64_t i = cell_index_w( N, x, y, z );
      double        t = 1.0 * rhs[i] + ( -1.0 ) * dst[i];
      t               = ( invdiag ? invdiag[i] : invC ) * t;
      dst[i]          = 1.0 * src[i] + relax * t;
   }
}

/* ---------------------------------------------------------------------------------------------
 * Classification of a point of the cell array by the macro-primitive it lies on.
 * src/hyteg/indexing/MacroCellIndexing.cpp:36-91 (isOnCellFace / isOnCellEdge / isOnCellVertex):
 * faces: 0: z==0, 1: y==0, 2: x==0, 3: x+y+z==width-1;  edges by face pairs
 * (0,1)->0 (0,2)->1 (0,3)->2 (1,2)->3 (1,3)->4 (2,3)->5; vertices 0:(0,0,0) 1:(w-1,0,0) 2:(0,w-1,0) 3:(0,0,w-1).
 * Returns the slot in nnc[14] = { edge0..5, face0..3, vertex0..3 } (the argument order of the
 * reference's grid-transfer kernels) or -1 for an interior point.
 * ------------------------------------------------------------------------------------------- */
static int prim_slot( int64_t width, int64_t x, int64_t y, int64_t z )
{
   const int f0 = ( z == 0 ), f1 = ( y == 0 ), f2 = ( x == 0 ), f3 = ( x + y + z == width - 1 );
   const int cnt = f0 + f1 + f2 + f3;
   if ( cnt == 0 )
      return -1;
   if ( cnt == 1 )
      return 6 + ( f0 ? 0 : f1 ? 1 : f2 ? 2 : 3 );
   if ( cnt == 2 )
   {
      if ( f0 && f1 ) return 0;
      if ( f0 && f2 ) return 1;
      if ( f0 && f3 ) return 2;
      if ( f1 && f2 ) return 3;
      if ( f1 && f3 ) return 4;
      return 5;
   }
   /* vertex */
   if ( f0 && f1 && f2 ) return 10;
   if ( f0 && f1 && f3 ) return 11;
   if ( f0 && f2 && f3 ) return 12;
   return 13;
}

HO_API int ho_prim_slot( int level, int x, int y, int z ) { return prim_slot( ho_width( level ), x, y, z ); }

/* ---------------------------------------------------------------------------------------------
 * a7: restriction, pull form.
 * src/hyteg/gridtransferoperators/generatedKernels/restrict_3D_macrocell_P1_pull_additive.cpp:33-1690
 * (15 regions; caller src/hyteg/gridtransferoperators/P1toP1LinearRestriction.cpp:169-243).
 * For EVERY coarse point i (boundary included):
 *    coarse[i] = sum over d in {centre, 14 neighbours}, fine point f = 2i+d inside the cell, of
 *                wt(d) * (1/nnc(primitive f lies on)) * fine[f],     wt(centre)=1, wt(else)=1/2,
 * where nnc = number of macro-cells adjacent to that macro-primitive (1 for interior points).
 * The per-cell partial sums are later added up across cells (additive communication,
 * P1toP1LinearRestriction.cpp:343-345), which is why shared fine values are pre-divided.
 * nnc[14] = { edge0..5, face0..3, vertex0..3 }.
 * Term order: centre last (as in every region of the reference); neighbours in the fixed order
 * of the table below.  The reference's per-region term order differs by reassociation only.
 * ------------------------------------------------------------------------------------------- */
static const int NB14[14][3] = { { -1, 0, 0 }, { -1, 0, 1 }, { -1, 1, -1 }, { -1, 1, 0 }, { 0, -1, 0 },
                                 { 0, -1, 1 }, { 0, 0, -1 }, { 0, 0, 1 },   { 0, 1, -1 }, { 0, 1, 0 },
                                 { 1, -1, 0 }, { 1, -1, 1 }, { 1, 0, -1 },  { 1, 0, 0 } };

static inline int inside( int64_t width, int64_t x, int64_t y, int64_t z )
{
   return x >= 0 && y >= 0 && z >= 0 && x + y + z <= width - 1;
}

HO_API void ho_restrict_cell( double* coarse, const double* fine, int coarse_level, const double* nnc )
{
   const int64_t Nc = ho_width( coarse_level ), Nf = ho_width( coarse_level + 1 );
   double        inv[14];
   for ( int k = 0; k < 14; ++k )
      inv[k] = 1 / nnc[k];
   for ( int64_t z = 0; z < Nc; ++z )
      for ( int64_t y = 0; y < Nc - z; ++y )
         for ( int64_t x = 0; x < Nc - z - y; ++x )
         {
            double acc   = 0;
            int    first = 1;
            for ( int k = 0; k < 14; ++k )
            {
               const int64_t fx = 2 * x + NB14[k][0], fy = 2 * y + NB14[k][1], fz = 2 * z + NB14[k][2];
               if ( !inside( Nf, fx, fy, fz ) )
                  continue;
               const int    slot = prim_slot( Nf, fx, fy, fz );
               const double s    = slot < 0 ? 1.0 : inv[slot];
               const double t    = s * 0.5 * fine[cell_index_w( Nf, fx, fy, fz )];
               acc               = first ? t : acc + t;
               first             = 0;
            }
            const int    slot = prim_slot( Nf, 2 * x, 2 * y, 2 * z );
            const double s    = slot < 0 ? 1.0 : inv[slot];
            const double t    = 1.0 * s * fine[cell_index_w( Nf, 2 * x, 2 * y, 2 * z )];
            acc               = first ? t : acc + t;
            coarse[cell_index_w( Nc, x, y, z )] = acc;
         }
}

/* ---------------------------------------------------------------------------------------------
 * a8: prolongation, push (scatter-add) form.
 * src/hyteg/gridtransferoperators/generatedKernels/prolongate_3D_macrocell_P1_push_additive.cpp
 * (caller + readable generic branch: src/hyteg/gridtransferoperators/P1toP1LinearProlongation.cpp:194-410).
 * ho_prolongate_prepare restates :214-238 (Replace: zero the whole fine array; Add: zero only the
 * points on the cell boundary).  ho_prolongate_cell then visits every coarse point i in
 * lexicographic (z,y,x) order and does
 *    fine[2i]   += (1/nnc(prim(2i)))       * coarse[i]
 *    fine[2i+d] += 1/2 * (1/nnc(prim(2i+d))) * coarse[i]      for the 14 d with 2i+d inside the cell.
 * ------------------------------------------------------------------------------------------- */
HO_API void ho_prolongate_prepare( double* fine, int fine_level, int update )
{
   const int64_t Nf = ho_width( fine_level );
   for ( int64_t z = 0; z < Nf; ++z )
      for ( int64_t y = 0; y < Nf - z; ++y )
         for ( int64_t x = 0; x < Nf - z - y; ++x )
            if ( update == 0 || prim_slot( Nf, x, y, z ) >= 0 )
               fine[cell_index_w( Nf, x, y, z )] = 0.0;
}

HO_API void ho_prolongate_cell( const double* coarse, double* fine, int coarse_level, const double* nnc )
{
   const int64_t Nc = ho_width( coarse_level ), Nf = ho_width( coarse_level + 1 );
   double        inv[14];
   for ( int k = 0; k < 14; ++k )
      inv[k] = 1 / nnc[k];
   for ( int64_t z = 0; z < Nc; ++z )
      for ( int64_t y = 0; y < Nc - z; ++y )
         for ( int64_t x = 0; x < Nc - z - y; ++x )
         {
            const double c = coarse[cell_index_w( Nc, x, y, z )];
            {
               const int    slot = prim_slot( Nf, 2 * x, 2 * y, 2 * z );
               const double s    = slot < 0 ? 1.0 : inv[slot];
               fine[cell_index_w( Nf, 2 * x, 2 * y, 2 * z )] += 1.0 * s * c;
            }
            for ( int k = 0; k < 14; ++k )
            {
               const int64_t fx = 2 * x + NB14[k][0], fy = 2 * y + NB14[k][1], fz = 2 * z + NB14[k][2];
               if ( !inside( Nf, fx, fy, fz ) )
                  continue;
               const int    slot = prim_slot( Nf, fx, fy, fz );
               const double s    = slot < 0 ? 1.0 : inv[slot];
               fine[cell_index_w( Nf, fx, fy, fz )] += s * 0.5 * c;
            }
         }
}

/* ---------------------------------------------------------------------------------------------
 * a11: stencil assembly (kernel INPUT).
 * Element matrix: P1 stiffness matrix of one tetrahedron, the quantity
 * src/hyteg/forms/form_fenics_generated/p1_tet_diffusion.h:4113-4240 (tabulate_tensor) computes:
 *    K_ij = |det J| / 6 * ( grad lambda_i . grad lambda_j ).
 * Here by the closed form via the inverse Jacobian (the generated code evaluates the same
 * expression through 80 temporaries and the literal 0.1666666666666667; agreement is to rounding,
 * checked against the header compiled in place, oracle/_ref).  coords = 4 vertices x 3, row-major.
 * ------------------------------------------------------------------------------------------- */
HO_API void ho_p1_tet_diffusion( double* A /*16, row-major*/, const double* c /*12*/ )
{
   double J[3][3]; /* J[r][k] = d x_r / d xi_k = c[k+1][r] - c[0][r] */
   for ( int r = 0; r < 3; ++r )
      for ( int k = 0; k < 3; ++k )
         J[r][k] = c[3 * ( k + 1 ) + r] - c[r];
   const double det = J[0][0] * ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) - J[0][1] * ( J[1][0] * J[2][2] - J[1][2] * J[2][0] ) +
                      J[0][2] * ( J[1][0] * J[2][1] - J[1][1] * J[2][0] );
   double Ji[3][3]; /* inverse: Ji[k][r] = d xi_k / d x_r */
   Ji[0][0] = ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) / det;
   Ji[0][1] = ( J[0][2] * J[2][1] - J[0][1] * J[2][2] ) / det;
   Ji[0][2] = ( J[0][1] * J[1][2] - J[0][2] * J[1][1] ) / det;
   Ji[1][0] = ( J[1][2] * J[2][0] - J[1][0] * J[2][2] ) / det;
   Ji[1][1] = ( J[0][0] * J[2][2] - J[0][2] * J[2][0] ) / det;
   Ji[1][2] = ( J[0][2] * J[1][0] - J[0][0] * J[1][2] ) / det;
   Ji[2][0] = ( J[1][0] * J[2][1] - J[1][1] * J[2][0] ) / det;
   Ji[2][1] = ( J[0][1] * J[2][0] - J[0][0] * J[2][1] ) / det;
   Ji[2][2] = ( J[0][0] * J[1][1] - J[0][1] * J[1][0] ) / det;
   double g[4][3]; /* physical gradients of the barycentric functions */
   for ( int r = 0; r < 3; ++r )
   {
      g[1][r] = Ji[0][r];
      g[2][r] = Ji[1][r];
      g[3][r] = Ji[2][r];
      g[0][r] = -( Ji[0][r] + Ji[1][r] + Ji[2][r] );
   }
   const double vol6 = fabs( det ) / 6.0;
   for ( int i = 0; i < 4; ++i )
      for ( int j = 0; j < 4; ++j )
         A[4 * i + j] = vol6 * ( g[i][0] * g[j][0] + g[i][1] * g[j][1] + g[i][2] * g[j][2] );
}

/* P1 mass matrix of one tetrahedron (src/hyteg/forms/form_fenics_generated/p1_tet_mass.h):
 * M_ij = |det J| / 120 * (1 + delta_ij). */
HO_API void ho_p1_tet_mass( double* A, const double* c )
{
   double J[3][3];
   for ( int r = 0; r < 3; ++r )
      for ( int k = 0; k < 3; ++k )
         J[r][k] = c[3 * ( k + 1 ) + r] - c[r];
   const double det = J[0][0] * ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) - J[0][1] * ( J[1][0] * J[2][2] - J[1][2] * J[2][0] ) +
                      J[0][2] * ( J[1][0] * J[2][1] - J[1][1] * J[2][0] );
   for ( int i = 0; i < 4; ++i )
      for ( int j = 0; j < 4; ++j )
         A[4 * i + j] = fabs( det ) / 120.0 * ( i == j ? 2.0 : 1.0 );
}

/* barycentric gradients g[4][3] and volume of a tetrahedron (shared by the first-order forms below) */
static double tet_gradients( double g[4][3], const double* c )
{
   double J[3][3];
   for ( int r = 0; r < 3; ++r )
      for ( int k = 0; k < 3; ++k )
         J[r][k] = c[3 * ( k + 1 ) + r] - c[r];
   const double det = J[0][0] * ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) - J[0][1] * ( J[1][0] * J[2][2] - J[1][2] * J[2][0] ) +
                      J[0][2] * ( J[1][0] * J[2][1] - J[1][1] * J[2][0] );
   double Ji[3][3];
   Ji[0][0] = ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) / det;
   Ji[0][1] = ( J[0][2] * J[2][1] - J[0][1] * J[2][2] ) / det;
   Ji[0][2] = ( J[0][1] * J[1][2] - J[0][2] * J[1][1] ) / det;
   Ji[1][0] = ( J[1][2] * J[2][0] - J[1][0] * J[2][2] ) / det;
   Ji[1][1] = ( J[0][0] * J[2][2] - J[0][2] * J[2][0] ) / det;
   Ji[1][2] = ( J[0][2] * J[1][0] - J[0][0] * J[1][2] ) / det;
   Ji[2][0] = ( J[1][0] * J[2][1] - J[1][1] * J[2][0] ) / det;
   Ji[2][1] = ( J[0][1] * J[2][0] - J[0][0] * J[2][1] ) / det;
   Ji[2][2] = ( J[0][0] * J[1][1] - J[0][1] * J[1][0] ) / det;
   for ( int r = 0; r < 3; ++r )
   {
      g[1][r] = Ji[0][r];
      g[2][r] = Ji[1][r];
      g[3][r] = Ji[2][r];
      g[0][r] = -( Ji[0][r] + Ji[1][r] + Ji[2][r] );
   }
   return fabs( det ) / 6.0;
}

/* Blocks of the P1-P1 Stokes operator (src/mixed_operator/P1P1StokesOperator.hpp:51-64), element matrices A[test i][trial j]
 * as the FEniCS-generated code writes them (row-major; P1FenicsForm.hpp:96-124 reads row 0 of the row-major hyteg::Matrix):
 *   div_k   data/operators/form_div_tet.ufl   -u.dx(k) q dx     -> -(g_j)_k V/4   (p1_tet_div_tet.h, cell_integral_k)
 *   divT_k  data/operators/form_divt_tet.ufl  -v.dx(k) p dx     -> -(g_i)_k V/4   (p1_tet_divt_tet.h, cell_integral_k)
 *   pspg    data/operators/form_pspg_tet.ufl  -tau grad p.grad q dx, tau = V^(2/3)/12  (p1_tet_pspg_tet.h) */
HO_API void ho_p1_tet_div( double* A, const double* c, int k )
{
   double       g[4][3];
   const double V = tet_gradients( g, c );
   for ( int i = 0; i < 4; ++i )
      for ( int j = 0; j < 4; ++j )
         A[4 * i + j] = -g[j][k] * V / 4.0;
}
HO_API void ho_p1_tet_divt( double* A, const double* c, int k )
{
   double       g[4][3];
   const double V = tet_gradients( g, c );
   for ( int i = 0; i < 4; ++i )
      for ( int j = 0; j < 4; ++j )
         A[4 * i + j] = -g[i][k] * V / 4.0;
}
HO_API void ho_p1_tet_pspg( double* A, const double* c )
{
   double       g[4][3];
   const double V   = tet_gradients( g, c );
   const double tau = pow( V, 2.0 / 3.0 ) / 12.0;
   for ( int i = 0; i < 4; ++i )
      for ( int j = 0; j < 4; ++j )
         A[4 * i + j] = -tau * V * ( g[i][0] * g[j][0] + g[i][1] * g[j][1] + g[i][2] * g[j][2] );
}
/* form ids shared with the host layer's facade: 0 Laplace, 1 mass, 2-4 div x/y/z, 5-7 divT x/y/z, 8 PSPG */
static void element_matrix( int form, double* A, const double* coords )
{
   if ( form == 0 )
      ho_p1_tet_diffusion( A, coords );
   else if ( form == 1 )
      ho_p1_tet_mass( A, coords );
   else if ( form >= 2 && form <= 4 )
      ho_p1_tet_div( A, coords, form - 2 );
   else if ( form >= 5 && form <= 7 )
      ho_p1_tet_divt( A, coords, form - 5 );
   else
      ho_p1_tet_pspg( A, coords );
}

/* The 24 micro-tetrahedra around an interior micro-vertex, each as 4 stencil slots with the
 * centre first.  src/hyteg/p1functionspace/P1Elements.hpp:93-143 (white/blue/green up/down). */
static const int MICRO_TETS[24][4] = {
    /* whiteUp   */ { W_C, W_BC, W_BE, W_BN }, { W_C, W_S, W_SE, W_TS }, { W_C, W_W, W_NW, W_TW }, { W_C, W_N, W_E, W_TC },
    /* whiteDown */ { W_C, W_W, W_BC, W_S },   { W_C, W_E, W_SE, W_BE }, { W_C, W_N, W_NW, W_BN }, { W_C, W_TS, W_TC, W_TW },
    /* blueUp    */ { W_C, W_BC, W_BN, W_BNW }, { W_C, W_W, W_S, W_TS }, { W_C, W_E, W_SE, W_TSE }, { W_C, W_NW, W_N, W_TC },
    /* blueDown  */ { W_C, W_BC, W_S, W_SE },  { W_C, W_W, W_NW, W_BNW }, { W_C, W_E, W_BN, W_N }, { W_C, W_TC, W_TS, W_TSE },
    /* greenUp   */ { W_C, W_W, W_BC, W_BNW }, { W_C, W_E, W_BE, W_BN }, { W_C, W_TC, W_TW, W_NW }, { W_C, W_SE, W_TS, W_TSE },
    /* greenDown */ { W_C, W_BC, W_BE, W_SE }, { W_C, W_BN, W_BNW, W_NW }, { W_C, W_E, W_TSE, W_TC }, { W_C, W_W, W_TS, W_TW } };

/* Cell stencil at an interior micro-vertex (the reference assembles it at index (1,1,1),
 * src/constant_stencil_operator/P1ConstantOperator.cpp:680-693 -> P1Operator.hpp:2123-2136 ->
 * P1Elements.hpp:466-528; geometry src/hyteg/p1functionspace/VertexDoFMacroCell.hpp:70-77).
 * form: 0 = Laplace (diffusion), 1 = mass, 2-4 div x/y/z, 5-7 divT x/y/z, 8 PSPG.  cellcoords = 4 macro-vertices x 3. */
HO_API void ho_assemble_cell_stencil( double* w, const double* cc, int level, int form )
{
   const double step = 1.0 / (double) ( (int64_t) 1 << level );
   double       xs[3], ys[3], zs[3];
   for ( int r = 0; r < 3; ++r )
   {
      xs[r] = ( cc[3 + r] - cc[r] ) * step;
      ys[r] = ( cc[6 + r] - cc[r] ) * step;
      zs[r] = ( cc[9 + r] - cc[r] ) * step;
   }
   for ( int k = 0; k < 15; ++k )
      w[k] = 0.0;
   for ( int t = 0; t < 24; ++t )
   {
      double coords[12], A[16];
      for ( int v = 0; v < 4; ++v )
      {
         const int* o = OFFS[MICRO_TETS[t][v]];
         for ( int r = 0; r < 3; ++r )
            coords[3 * v + r] = cc[r] + xs[r] * (double) ( 1 + o[0] ) + ys[r] * (double) ( 1 + o[1] ) + zs[r] * (double) ( 1 + o[2] );
      }
      element_matrix( form, A, coords );
      for ( int v = 0; v < 4; ++v )
         w[MICRO_TETS[t][v]] += A[v]; /* first row of the element matrix */
   }
}

/* physical coordinate of a micro-vertex: VertexDoFMacroCell.hpp:70-77 */
HO_API void ho_coordinate_from_index( double* out, const double* cc, int level, int x, int y, int z )
{
   const double step = 1.0 / (double) ( (int64_t) 1 << level );
   for ( int r = 0; r < 3; ++r )
   {
      const double xs = ( cc[3 + r] - cc[r] ) * step, ys = ( cc[6 + r] - cc[r] ) * step, zs = ( cc[9 + r] - cc[r] ) * step;
      out[r] = cc[r] + xs * (double) x + ys * (double) y + zs * (double) z;
   }
}

/* ---------------------------------------------------------------------------------------------
 * a9 (cell-centric form): per-cell PARTIAL stencils on the cell boundary and the apply that uses them.
 * src/hyteg/p1functionspace/P1Elements.hpp:215-301 (getNeighboringElements: a micro-vertex on a cell
 * face keeps the 12 micro-tets of allCellsAtFace[f] (:145-205), on a cell edge the intersection of two
 * such lists, on a cell vertex a single micro-tet) and :303-380 (calculateStencilInMacroCell).
 * Those lists are exactly the micro-tets of the 24 whose four vertices stay inside the cell, i.e. whose
 * directions d satisfy, for every cell face the point lies on:
 *    face 0 (z=0): dz >= 0,  face 1 (y=0): dy >= 0,  face 2 (x=0): dx >= 0,  face 3 (x+y+z=n): dx+dy+dz <= 0.
 * The reference keeps these weights per neighbour cell in faceStencil3D / edgeStencil3D
 * (src/constant_stencil_operator/P1ConstantOperator.cpp:239-357) and sums the per-cell results.
 * Slots: { edge0..5, face0..3, vertex0..3 } (ho_prim_slot).  w_slots: 14 x 15, stencil order as w[15].
 * ------------------------------------------------------------------------------------------- */
static const int SLOT_FACES[14][4] = {
    /* edges  */ { 1, 1, 0, 0 }, { 1, 0, 1, 0 }, { 1, 0, 0, 1 }, { 0, 1, 1, 0 }, { 0, 1, 0, 1 }, { 0, 0, 1, 1 },
    /* faces  */ { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 },
    /* verts  */ { 1, 1, 1, 0 }, { 1, 1, 0, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 } };

static int dir_allowed( int slot, const int* d )
{
   const int* f = SLOT_FACES[slot];
   if ( f[0] && d[2] < 0 ) return 0;
   if ( f[1] && d[1] < 0 ) return 0;
   if ( f[2] && d[0] < 0 ) return 0;
   if ( f[3] && d[0] + d[1] + d[2] > 0 ) return 0;
   return 1;
}

HO_API void ho_assemble_cell_slot_stencils( double* w_slots, const double* cc, int level, int form )
{
   const double step = 1.0 / (double) ( (int64_t) 1 << level );
   double       xs[3], ys[3], zs[3];
   for ( int r = 0; r < 3; ++r )
   {
      xs[r] = ( cc[3 + r] - cc[r] ) * step;
      ys[r] = ( cc[6 + r] - cc[r] ) * step;
      zs[r] = ( cc[9 + r] - cc[r] ) * step;
   }
   double rows[24][4];
   for ( int t = 0; t < 24; ++t )
   {
      double coords[12], A[16];
      for ( int v = 0; v < 4; ++v )
      {
         const int* o = OFFS[MICRO_TETS[t][v]];
         for ( int r = 0; r < 3; ++r )
            coords[3 * v + r] = cc[r] + xs[r] * (double) ( 1 + o[0] ) + ys[r] * (double) ( 1 + o[1] ) + zs[r] * (double) ( 1 + o[2] );
      }
      element_matrix( form, A, coords );
      for ( int v = 0; v < 4; ++v )
         rows[t][v] = A[v];
   }
   for ( int s = 0; s < 14; ++s )
   {
      double* w = w_slots + 15 * s;
      for ( int k = 0; k < 15; ++k )
         w[k] = 0.0;
      for ( int t = 0; t < 24; ++t )
      {
         int ok = 1;
         for ( int v = 1; v < 4; ++v )
            ok = ok && dir_allowed( s, OFFS[MICRO_TETS[t][v]] );
         if ( !ok )
            continue;
         for ( int v = 0; v < 4; ++v )
            w[MICRO_TETS[t][v]] += rows[t][v];
      }
   }
}

/* mask bit k (0..13): points on slot k; bit 14: interior */
static inline int point_selected( unsigned mask, int slot ) { return ( mask >> ( slot < 0 ? 14 : slot ) ) & 1u; }

HO_API void ho_apply_cell_boundary( double* dst, const double* src, int level, const double* w_slots, unsigned mask, int update )
{
   const int64_t N = ho_width( level );
   for ( int64_t z = 0; z < N; ++z )
      for ( int64_t y = 0; y < N - z; ++y )
         for ( int64_t x = 0; x < N - z - y; ++x )
         {
            const int slot = prim_slot( N, x, y, z );
            if ( slot < 0 || !point_selected( mask, slot ) )
               continue;
            const double* w   = w_slots + 15 * slot;
            double        acc = 0.0;
            for ( int k = 0; k < 15; ++k )
            {
               const int64_t nx = x + OFFS[k][0], ny = y + OFFS[k][1], nz = z + OFFS[k][2];
               if ( !inside( N, nx, ny, nz ) )
                  continue;
               acc = acc + w[k] * src[cell_index_w( N, nx, ny, nz )];
            }
            const int64_t i = cell_index_w( N, x, y, z );
            dst[i]          = update ? acc + dst[i] : acc;
         }
}

/* masked vector kernels: op 0 assign, 1 add, 2 mult, 3 set constant scalars[0]
 * (VertexDoFFunction.cpp:1130-1221, 1408-1484, 1487-1563, interpolate( constant ) :380-392) */
HO_API void ho_vector_cell_masked( int op, double* dst, int nsrc, const double* const* srcs, const double* scalars, int level,
                                   unsigned mask )
{
   const int64_t N = ho_width( level );
   for ( int64_t z = 0; z < N; ++z )
      for ( int64_t y = 0; y < N - z; ++y )
         for ( int64_t x = 0; x < N - z - y; ++x )
         {
            if ( !point_selected( mask, prim_slot( N, x, y, z ) ) )
               continue;
            const int64_t i = cell_index_w( N, x, y, z );
            double        tmp;
            if ( op == 3 )
               tmp = scalars[0];
            else if ( op == 2 )
            {
               tmp = srcs[0][i];
               for ( int k = 1; k < nsrc; ++k )
                  tmp = tmp * srcs[k][i];
            }
            else
            {
               tmp = scalars[0] * srcs[0][i];
               for ( int k = 1; k < nsrc; ++k )
                  tmp = tmp + scalars[k] * srcs[k][i];
               if ( op == 1 )
                  tmp = dst[i] + tmp;
            }
            dst[i] = tmp;
         }
}

HO_API double ho_dot_cell_masked( const double* a, const double* b, int level, unsigned mask )
{
   const int64_t N  = ho_width( level );
   double        sp = 0;
   for ( int64_t z = 0; z < N; ++z )
      for ( int64_t y = 0; y < N - z; ++y )
         for ( int64_t x = 0; x < N - z - y; ++x )
            if ( point_selected( mask, prim_slot( N, x, y, z ) ) )
            {
               const int64_t i = cell_index_w( N, x, y, z );
               sp              = sp + a[i] * b[i];
            }
   return sp;
}

/* ---------------------------------------------------------------------------------------------
 * a9 / a10 in HyTeG's own macro-face layout (drop-in seam).
 * Macro-face array of width N: [ tri(N) face DoFs | tri(N-1) ghost layer of neighbour cell 0 | ... of cell 1 ]
 * (src/hyteg/p1functionspace/VertexDoFMemory.hpp:59-63, VertexDoFIndexing.cpp:219-226).
 * A face point (x,y) / ghost point (x,y,1) in the face's basis (v0,v1,v2,v3) -- v0..v2 the cell-local vertices of
 * the face's vertices 0..2 (Cell::getFaceLocalVertexToCellLocalVertexMaps), v3 the remaining one -- is the cell
 * point with weights n-x-y-z, x, y, z on cell vertices v0, v1, v2, v3
 * (indexing::basisConversion, src/hyteg/indexing/DistanceCoordinateSystem.hpp:107-150;
 *  getIndexInNeighboringMacroCell, src/hyteg/p1functionspace/VertexDoFMacroFace.hpp:53-70).
 * ------------------------------------------------------------------------------------------- */
static void face_to_cell_coords( int64_t n, const int* v, int64_t fx, int64_t fy, int64_t fz, int64_t* c )
{
   int64_t bary[4] = { 0, 0, 0, 0 };
   int     v3      = 6 - v[0] - v[1] - v[2];
   bary[v[0]]      = n - fx - fy - fz;
   bary[v[1]]      = fx;
   bary[v[2]]      = fy;
   bary[v3]        = fz;
   c[0] = bary[1], c[1] = bary[2], c[2] = bary[3];
}

/* communicateLocalFaceToCell, src/hyteg/p1functionspace/VertexDoFPackInfo.hpp:438-478 */
HO_API void ho_copy_face_to_cell( double* cell, const double* face, int level, int v0, int v1, int v2 )
{
   const int64_t N = ho_width( level ), n = N - 1;
   const int     v[3] = { v0, v1, v2 };
   for ( int64_t y = 0; y < N; ++y )
      for ( int64_t x = 0; x < N - y; ++x )
      {
         int64_t c[3];
         face_to_cell_coords( n, v, x, y, 0, c );
         cell[cell_index_w( N, c[0], c[1], c[2] )] = face[ho_face_index_w( N, x, y )];
      }
}

/* communicateLocalCellToFace, VertexDoFPackInfo.hpp:552-615: the cell layer at distance 1 from the face */
HO_API void ho_copy_cell_to_face( double* face, const double* cell, int level, int v0, int v1, int v2, int neighbor )
{
   const int64_t N = ho_width( level ), n = N - 1;
   const int     v[3] = { v0, v1, v2 };
   double*       ghost = face + ho_face_size_w( N ) + neighbor * ho_face_size_w( N - 1 );
   for ( int64_t y = 0; y < N - 1; ++y )
      for ( int64_t x = 0; x < N - 1 - y; ++x )
      {
         int64_t c[3];
         face_to_cell_coords( n, v, x, y, 1, c );
         ghost[ho_face_index_w( N - 1, x, y )] = cell[cell_index_w( N, c[0], c[1], c[2] )];
      }
}

/* P1Operator::apply_face3D, src/hyteg/p1functionspace/P1Operator.hpp:1111-1173: for every inner face DoF and every
 * neighbour cell, the cell's share of the stencil (weights in the CELL's stencil directions, w15 order) applied to
 * leaves that are converted back into the face's array (in-plane leaves) or the cell's ghost layer (distance 1).
 * vmaps: ncells x 3 cell-local vertex ids; w: ncells x 15. */
HO_API void ho_apply_face3d( double* dst, const double* src, int level, int ncells, const int* vmaps, const double* w, int update )
{
   const int64_t N = ho_width( level ), n = N - 1;
   for ( int64_t y = 1; y < N - 2; ++y )
      for ( int64_t x = 1; x < N - 1 - y - 0; ++x )
      {
         if ( x + y > n - 1 )
            continue;
         double tmp = 0.0;
         for ( int k = 0; k < ncells; ++k )
         {
            const int* v  = vmaps + 3 * k;
            const int  v3 = 6 - v[0] - v[1] - v[2];
            int64_t    c[3];
            face_to_cell_coords( n, v, x, y, 0, c );
            for ( int s = 0; s < 15; ++s )
            {
               const double weight = w[15 * k + s];
               const int64_t lx = c[0] + OFFS[s][0], ly = c[1] + OFFS[s][1], lz = c[2] + OFFS[s][2];
               if ( !inside( N, lx, ly, lz ) )
                  continue; /* direction leaves the cell: not part of this cell's share */
               int64_t bary[4] = { n - lx - ly - lz, lx, ly, lz };
               const int64_t fx = bary[v[1]], fy = bary[v[2]], fz = bary[v3];
               if ( fz > 1 )
                  continue;
               const int64_t idx = fz == 0 ? ho_face_index_w( N, fx, fy ) :
                                             ho_face_size_w( N ) + k * ho_face_size_w( N - 1 ) + ho_face_index_w( N - 1, fx, fy );
               tmp += weight * src[idx];
            }
         }
         const int64_t i = ho_face_index_w( N, x, y );
         dst[i]          = update ? dst[i] + tmp : tmp;
      }
}

/* SOR / Gauss-Seidel sweep over the inner DoFs of ONE macro-face in HyTeG's own face layout
 * [ tri(N) face DoFs | ghost layer of neighbour cell 0 | ghost layer of neighbour cell 1 ], in place.
 * P1Operator::smooth_sor_face3D, src/hyteg/p1functionspace/P1Operator.hpp:1424-1503 (the loop the constant-stencil operator's
 * generated kernels sor_3D_macroface_P1{,_one_sided}{,_backwards} unroll; call site P1ConstantOperator.cpp:432-571): for every
 * inner face DoF in lexicographic (y, x) order -- backwards: the reverse order -- tmp = rhs - sum over the neighbour cells of
 * that cell's off-centre stencil leaves (in-plane leaves from the face DoFs, leaves at distance 1 from the cell's ghost
 * layer), dst = (1 - relax) dst + relax tmp / (sum of the cells' centre weights).  vmaps, w as in ho_apply_face3d. */
HO_API void ho_sor_face3d( double* dst, const double* rhs, int level, int ncells, const int* vmaps, const double* w, double relax, int backwards )
{
   const int64_t N = ho_width( level ), n = N - 1;
   double        centre = 0.0;
   for ( int k = 0; k < ncells; ++k )
      centre += w[15 * k + 7];
   const double  inv   = 1.0 / centre;
   const int64_t count = ( N - 3 ) * ( N - 2 ) / 2; /* inner face DoFs: y = 1..N-3, x = 1..N-2-y */
   for ( int64_t q = 0; q < count; ++q )
   {
      /* the q-th inner DoF in (y, x) order, or in the reverse of that order */
      int64_t r = backwards ? count - 1 - q : q, y = 1;
      while ( r >= N - 2 - y )
      {
         r -= N - 2 - y;
         ++y;
      }
      const int64_t x   = 1 + r;
      double        tmp = rhs[ho_face_index_w( N, x, y )];
      for ( int k = 0; k < ncells; ++k )
      {
         const int* v  = vmaps + 3 * k;
         const int  v3 = 6 - v[0] - v[1] - v[2];
         int64_t    c[3];
         face_to_cell_coords( n, v, x, y, 0, c );
         for ( int s = 0; s < 15; ++s )
         {
            if ( s == 7 )
               continue;
            const int64_t lx = c[0] + OFFS[s][0], ly = c[1] + OFFS[s][1], lz = c[2] + OFFS[s][2];
            if ( !inside( N, lx, ly, lz ) )
               continue;
            int64_t       bary[4] = { n - lx - ly - lz, lx, ly, lz };
            const int64_t fx = bary[v[1]], fy = bary[v[2]], fz = bary[v3];
            if ( fz > 1 )
               continue;
            const int64_t idx = fz == 0 ? ho_face_index_w( N, fx, fy ) :
                                          ho_face_size_w( N ) + k * ho_face_size_w( N - 1 ) + ho_face_index_w( N - 1, fx, fy );
            tmp -= w[15 * k + s] * dst[idx];
         }
      }
      const int64_t i = ho_face_index_w( N, x, y );
      dst[i]          = ( 1.0 - relax ) * dst[i] + relax * tmp * inv;
   }
}

/* ---- SOR / Gauss-Seidel on the macro-vertices, -edges and -faces around one cell, cell-centric restatement ----
 * vertexdof::macrovertex::smooth_sor (src/hyteg/p1functionspace/VertexDoFMacroVertex.hpp:231-251),
 * P1Operator::smooth_sor_edge (src/hyteg/p1functionspace/P1Operator.hpp:1352-1421) and
 * P1Operator::smooth_sor_face3D (:1424-1503), called in this order by smooth_sor (:348-418; backwards: reversed).
 * A macro-primitive is swept in its own memory there: its own points (and the lower-dimensional primitives on its
 * boundary) carry current values, the ghost layers carry what the last communication delivered.  In the cell-centric
 * storage the ghost-layer part of the sum, taken over ALL neighbour cells, is the input `rest`; the weights of the
 * primitive's own points are the totals over all neighbour cells:
 *   vertex_w[k]            centre weight of cell-local vertex k
 *   edge_verts[e] = (a,b)  the edge runs from cell-local vertex a to b (the macro-edge's own orientation),
 *   edge_w[e]     = centre, weight towards a, weight towards b
 *   face_verts[f] = cell-local numbers of the macro-face's vertices 0, 1, 2 (x runs 0->1, y runs 0->2),
 *   face_w[f]     = centre, then the in-plane neighbours (-1,0) (1,0) (0,-1) (0,1) (1,-1) (-1,1) in face coordinates.
 * Order inside a primitive: edge index ascending, face rows y ascending and x ascending inside a row
 * (vertexdof::macroface::Iterator( level, 1 )); descending for backwards. */
static void unit_of_vertex( int k, int64_t n, int64_t* c )
{
   c[0] = k == 1 ? n : 0;
   c[1] = k == 2 ? n : 0;
   c[2] = k == 3 ? n : 0;
}

HO_API void ho_sor_shell_cell( double* dst, const double* rhs, const double* rest, int level, const int* edge_verts,
                               const double* edge_w, const int* face_verts, const double* face_w, const double* vertex_w,
                               double relax, unsigned mask, int backwards )
{
   const int64_t N = ho_width( level ), n = N - 1;
   static const int FD[6][2] = { { -1, 0 }, { 1, 0 }, { 0, -1 }, { 0, 1 }, { 1, -1 }, { -1, 1 } };
   for ( int phase = 0; phase < 3; ++phase )
   {
      const int cls = backwards ? 2 - phase : phase; /* 0 vertices, 1 edges, 2 faces */
      if ( cls == 0 )
      {
         for ( int k = 0; k < 4; ++k )
         {
            if ( !point_selected( mask, 10 + k ) )
               continue;
            int64_t c[3];
            unit_of_vertex( k, n, c );
            const int64_t i   = cell_index_w( N, c[0], c[1], c[2] );
            double        tmp = rhs[i];
            tmp -= rest[i];
            dst[i] = ( 1.0 - relax ) * dst[i] + relax * tmp / vertex_w[k];
         }
      }
      else if ( cls == 1 )
      {
         for ( int e = 0; e < 6; ++e )
         {
            if ( !point_selected( mask, e ) )
               continue;
            int64_t o[3], ua[3], ub[3];
            unit_of_vertex( edge_verts[2 * e], n, o );
            unit_of_vertex( edge_verts[2 * e], 1, ua );
            unit_of_vertex( edge_verts[2 * e + 1], 1, ub );
            const int64_t d[3]            = { ub[0] - ua[0], ub[1] - ua[1], ub[2] - ua[2] };
            const double  invCenterWeight = 1.0 / edge_w[3 * e];
            for ( int64_t ii = backwards ? n - 1 : 1; ii != ( backwards ? 0 : n ); ii += backwards ? -1 : 1 )
            {
               const int64_t iC  = cell_index_w( N, o[0] + ii * d[0], o[1] + ii * d[1], o[2] + ii * d[2] );
               const int64_t iW  = cell_index_w( N, o[0] + ( ii - 1 ) * d[0], o[1] + ( ii - 1 ) * d[1], o[2] + ( ii - 1 ) * d[2] );
               const int64_t iE  = cell_index_w( N, o[0] + ( ii + 1 ) * d[0], o[1] + ( ii + 1 ) * d[1], o[2] + ( ii + 1 ) * d[2] );
               double        tmp = rhs[iC];
               tmp -= edge_w[3 * e + 1] * dst[iW] + edge_w[3 * e + 2] * dst[iE];
               tmp -= rest[iC];
               dst[iC] = ( 1.0 - relax ) * dst[iC] + relax * invCenterWeight * tmp;
            }
         }
      }
      else
      {
         for ( int f = 0; f < 4; ++f )
         {
            if ( !point_selected( mask, 6 + f ) )
               continue;
            int64_t o[3], u0[3], u1[3], u2[3];
            unit_of_vertex( face_verts[3 * f], n, o );
            unit_of_vertex( face_verts[3 * f], 1, u0 );
            unit_of_vertex( face_verts[3 * f + 1], 1, u1 );
            unit_of_vertex( face_verts[3 * f + 2], 1, u2 );
            const double invCenterWeight = 1.0 / face_w[7 * f];
            const int64_t count = ( n - 2 ) * ( n - 1 ) / 2; /* inner points of the face */
            for ( int64_t q = 0; q < count; ++q )
            {
               /* q-th inner point in (y outer, x inner) order, reversed for backwards */
               int64_t r = backwards ? count - 1 - q : q, y = 1, x;
               while ( r >= n - 1 - y )
               {
                  r -= n - 1 - y;
                  ++y;
               }
               x = 1 + r;
               int64_t p[3];
               for ( int a = 0; a < 3; ++a )
                  p[a] = o[a] + x * ( u1[a] - u0[a] ) + y * ( u2[a] - u0[a] );
               const int64_t iC  = cell_index_w( N, p[0], p[1], p[2] );
               double        tmp = rhs[iC];
               for ( int k = 0; k < 6; ++k )
               {
                  int64_t l[3];
                  for ( int a = 0; a < 3; ++a )
                     l[a] = p[a] + FD[k][0] * ( u1[a] - u0[a] ) + FD[k][1] * ( u2[a] - u0[a] );
                  tmp -= face_w[7 * f + 1 + k] * dst[cell_index_w( N, l[0], l[1], l[2] )];
               }
               tmp -= rest[iC];
               dst[iC] = ( 1.0 - relax ) * dst[iC] + relax * tmp * invCenterWeight;
            }
         }
      }
   }
}

/* =====================================================================================================================
 * P2 on one macro-cell: vertex DoFs (the P1 array) + edge DoFs, P2ElementwiseOperator::gemv.
 * Edge-DoF array of a macro-cell (src/hyteg/edgedofspace/EdgeDoFIndexing.hpp:920-985): seven blocks, one per orientation
 * X, Y, Z, XY, XZ, YZ (tetrahedral arrays of width n = 2^level) and XYZ (width n - 1), in this order.
 * ===================================================================================================================== */
enum { EO_X = 0, EO_Y, EO_Z, EO_XY, EO_XZ, EO_YZ, EO_XYZ };

HO_API int64_t ho_edge_array_size( int level )
{
   const int64_t n = (int64_t) 1 << level;
   return 6 * tet_size( n ) + tet_size( n - 1 );
}
HO_API int64_t ho_edge_index( int level, int64_t x, int64_t y, int64_t z, int orientation )
{
   const int64_t n = (int64_t) 1 << level;
   return orientation * tet_size( n ) + cell_index_w( orientation == EO_XYZ ? n - 1 : n, x, y, z );
}

/* calcEdgeDoFOrientation / calcEdgeDoFIndex, EdgeDoFIndexing.hpp:89-165: v0, v1 logical micro-vertex indices */
static int edge_between( const int64_t* v0, const int64_t* v1, int64_t* e )
{
   const int64_t  d[3] = { v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2] };
   const int64_t* lo;
   int            o;
   if ( d[1] == 0 && d[2] == 0 )
      o = EO_X;
   else if ( d[0] == 0 && d[2] == 0 )
      o = EO_Y;
   else if ( d[0] == 0 && d[1] == 0 )
      o = EO_Z;
   else if ( d[2] == 0 )
      o = EO_XY;
   else if ( d[1] == 0 )
      o = EO_XZ;
   else if ( d[0] == 0 )
      o = EO_YZ;
   else
      o = EO_XYZ;
   switch ( o )
   {
   case EO_X:
      lo = v0[0] < v1[0] ? v0 : v1;
      e[0] = lo[0], e[1] = lo[1], e[2] = lo[2];
      break;
   case EO_Y:
      lo = v0[1] < v1[1] ? v0 : v1;
      e[0] = lo[0], e[1] = lo[1], e[2] = lo[2];
      break;
   case EO_Z:
      lo = v0[2] < v1[2] ? v0 : v1;
      e[0] = lo[0], e[1] = lo[1], e[2] = lo[2];
      break;
   case EO_XY:
      lo = v0[0] < v1[0] ? v0 : v1;
      e[0] = lo[0], e[1] = lo[1] - 1, e[2] = lo[2];
      break;
   case EO_XZ:
      lo = v0[0] < v1[0] ? v0 : v1;
      e[0] = lo[0], e[1] = lo[1], e[2] = lo[2] - 1;
      break;
   case EO_YZ:
      lo = v0[1] < v1[1] ? v0 : v1;
      e[0] = lo[0], e[1] = lo[1], e[2] = lo[2] - 1;
      break;
   default:
      lo = v0[0] < v1[0] ? v0 : v1;
      e[0] = lo[0], e[1] = lo[1] - 1, e[2] = lo[2];
      break;
   }
   return o;
}

/* celldof::macrocell::getMicroVerticesFromMicroCell, src/hyteg/volumedofspace/CellDoFIndexing.hpp:155-198;
 * cell types in the order of celldof::allCellTypes (:55-60): WHITE_UP, BLUE_UP, GREEN_UP, WHITE_DOWN, BLUE_DOWN, GREEN_DOWN */
static const int MICRO_CELL_VERTS[6][4][3] = {
    { { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } }, { { 1, 0, 0 }, { 1, 1, 0 }, { 0, 1, 0 }, { 1, 0, 1 } },
    { { 1, 0, 0 }, { 0, 1, 0 }, { 1, 0, 1 }, { 0, 0, 1 } }, { { 1, 1, 0 }, { 1, 1, 1 }, { 0, 1, 1 }, { 1, 0, 1 } },
    { { 1, 0, 1 }, { 0, 1, 1 }, { 0, 0, 1 }, { 0, 1, 0 } }, { { 0, 1, 0 }, { 1, 1, 0 }, { 1, 0, 1 }, { 0, 1, 1 } } };
/* numCellsPerRowByType, CellDoFIndexing.hpp:64-83: n, n-1, n-1, n-2, n-1, n-1 */
static const int MICRO_CELL_ROW_DEFICIT[6] = { 0, 1, 1, 2, 1, 1 };
/* FEniCS ordering of the six edges of a micro-cell, EdgeDoFIndexing.hpp:1370-1375 */
static const int FENICS_EDGE_PAIRS[6][2] = { { 2, 3 }, { 1, 3 }, { 1, 2 }, { 0, 3 }, { 0, 2 }, { 0, 1 } };

/* the ten array indices of micro-cell (type, mx, my, mz): 0..3 into the vertex array, 4..9 into the edge array */
HO_API void ho_p2_micro_cell_dofs( int level, int type, int64_t mx, int64_t my, int64_t mz, int64_t* idx10 )
{
   const int64_t N = ho_width( level );
   int64_t       v[4][3];
   for ( int k = 0; k < 4; ++k )
   {
      v[k][0]  = mx + MICRO_CELL_VERTS[type][k][0];
      v[k][1]  = my + MICRO_CELL_VERTS[type][k][1];
      v[k][2]  = mz + MICRO_CELL_VERTS[type][k][2];
      idx10[k] = cell_index_w( N, v[k][0], v[k][1], v[k][2] );
   }
   for ( int k = 0; k < 6; ++k )
   {
      int64_t   e[3];
      const int o   = edge_between( v[FENICS_EDGE_PAIRS[k][0]], v[FENICS_EDGE_PAIRS[k][1]], e );
      idx10[4 + k] = ho_edge_index( level, e[0], e[1], e[2], o );
   }
}

/* P2 diffusion element matrix of a tetrahedron in FEniCS ordering (4 vertices, then the edges (2,3)(1,3)(1,2)(0,3)(0,2)(0,1)),
 * closed form: phi_a = l_a(2 l_a - 1), phi_ab = 4 l_a l_b, int l_a l_b = V(1 + delta_ab)/20, int l_a = V/4.
 * The reference uses the generated p2_tet_diffusion.h (forms/form_fenics_base/P2FenicsForm.cpp:160-175); pinned against it. */
HO_API void ho_p2_tet_diffusion( double* A, const double* coords )
{
   double J[3][3], Ji[3][3], g[4][3];
   for ( int r = 0; r < 3; ++r )
      for ( int k = 0; k < 3; ++k )
         J[r][k] = coords[3 * ( k + 1 ) + r] - coords[r];
   const double det = J[0][0] * ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) - J[0][1] * ( J[1][0] * J[2][2] - J[1][2] * J[2][0] ) +
                      J[0][2] * ( J[1][0] * J[2][1] - J[1][1] * J[2][0] );
   Ji[0][0] = ( J[1][1] * J[2][2] - J[1][2] * J[2][1] ) / det;
   Ji[0][1] = ( J[0][2] * J[2][1] - J[0][1] * J[2][2] ) / det;
   Ji[0][2] = ( J[0][1] * J[1][2] - J[0][2] * J[1][1] ) / det;
   Ji[1][0] = ( J[1][2] * J[2][0] - J[1][0] * J[2][2] ) / det;
   Ji[1][1] = ( J[0][0] * J[2][2] - J[0][2] * J[2][0] ) / det;
   Ji[1][2] = ( J[0][2] * J[1][0] - J[0][0] * J[1][2] ) / det;
   Ji[2][0] = ( J[1][0] * J[2][1] - J[1][1] * J[2][0] ) / det;
   Ji[2][1] = ( J[0][1] * J[2][0] - J[0][0] * J[2][1] ) / det;
   Ji[2][2] = ( J[0][0] * J[1][1] - J[0][1] * J[1][0] ) / det;
   for ( int r = 0; r < 3; ++r )
   {
      g[1][r] = Ji[0][r], g[2][r] = Ji[1][r], g[3][r] = Ji[2][r];
      g[0][r] = -( Ji[0][r] + Ji[1][r] + Ji[2][r] );
   }
   const double V = fabs( det ) / 6.0;
   double       G[4][4]; /* grad l_a . grad l_b */
   for ( int a = 0; a < 4; ++a )
      for ( int b = 0; b < 4; ++b )
         G[a][b] = g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2];
   /* grad phi_i = sum_a c_i[a][.] : every basis gradient is a combination  sum_{a,b} C_i[a][b] l_b grad l_a  + D_i[a] grad l_a */
   double C[10][4][4], D[10][4];
   memset( C, 0, sizeof( C ) );
   memset( D, 0, sizeof( D ) );
   for ( int a = 0; a < 4; ++a )
   {
      C[a][a][a] = 4.0; /* grad( l_a (2 l_a - 1) ) = (4 l_a - 1) grad l_a */
      D[a][a]    = -1.0;
   }
   for ( int k = 0; k < 6; ++k )
   {
      const int a = FENICS_EDGE_PAIRS[k][0], b = FENICS_EDGE_PAIRS[k][1];
      C[4 + k][a][b] = 4.0; /* grad( 4 l_a l_b ) = 4 l_b grad l_a + 4 l_a grad l_b */
      C[4 + k][b][a] = 4.0;
   }
   for ( int i = 0; i < 10; ++i )
      for ( int j = 0; j < 10; ++j )
      {
         double s = 0.0;
         for ( int a = 0; a < 4; ++a )
            for ( int b = 0; b < 4; ++b )
            {
               /* ( sum_p C_i[a][p] l_p + D_i[a] ) ( sum_q C_j[b][q] l_q + D_j[b] ) G[a][b] integrated */
               double t = D[i][a] * D[j][b] * V;
               for ( int p = 0; p < 4; ++p )
               {
                  t += C[i][a][p] * D[j][b] * V / 4.0 + D[i][a] * C[j][b][p] * V / 4.0;
                  for ( int q = 0; q < 4; ++q )
                     t += C[i][a][p] * C[j][b][q] * V * ( p == q ? 2.0 : 1.0 ) / 20.0;
               }
               s += t * G[a][b];
            }
         A[10 * i + j] = s;
      }
}

/* element matrices of the six micro-cell types of an affine macro-cell (assembleLocalElementMatrix3D,
 * elementwiseoperators/P2ElementwiseOperator.cpp + coordinateFromIndex): elmat[6][100] */
HO_API void ho_p2_cell_element_matrices( double* elmat, const double* cell_coords12, int level )
{
   for ( int t = 0; t < 6; ++t )
   {
      double c[12];
      for ( int k = 0; k < 4; ++k )
         ho_coordinate_from_index( c + 3 * k, cell_coords12, level, MICRO_CELL_VERTS[t][k][0], MICRO_CELL_VERTS[t][k][1],
                                   MICRO_CELL_VERTS[t][k][2] );
      ho_p2_tet_diffusion( elmat + 100 * t, c );
   }
}

/* The generated elementwise P1 diffusion operator's macro-cell kernel,
 * apps/2023-zikeli-mt/MT-apps/operators-used/P1ElementwiseDiffusion_cubes_const_float64.cpp:782- (apply_macro_3D: per
 * micro-cell type the element matrix of the affine micro-cell from the twelve macro-vertex coordinates, then for every
 * micro-cell of that type dst[v_i] += sum_j elMat_ij src[v_j]), which is the loop of the in-tree
 * P1ElementwiseOperator::apply (src/hyteg/elementwiseoperators/P1ElementwiseOperator.cpp:158-178 with
 * localMatrixVectorMultiply3D, P1LocalOperations.hpp:85-114) with the element matrix hoisted out of the micro-cell loop.
 * Literal restatement: micro-cell types in the order of celldof::allCellTypes, micro-cells in iterator order, the element
 * matrix from ho_p1_tet_diffusion (pinned to the reference's p1_tet_diffusion.h through oracle/_ref); dst is ADDED to at every
 * point of the cell array, boundary points included. */
HO_API void ho_p1_elementwise_apply_macro_3d( double* dst, const double* src, const double* cell_coords12, int64_t micro_edges )
{
   int level = 0;
   while ( ( (int64_t) 1 << level ) < micro_edges )
      ++level;
   const int64_t N = ho_width( level ), n = N - 1;
   for ( int t = 0; t < 6; ++t )
   {
      double c[12], M[16];
      for ( int k = 0; k < 4; ++k )
         ho_coordinate_from_index( c + 3 * k, cell_coords12, level, MICRO_CELL_VERTS[t][k][0], MICRO_CELL_VERTS[t][k][1],
                                   MICRO_CELL_VERTS[t][k][2] );
      ho_p1_tet_diffusion( M, c );
      const int64_t rows = n - MICRO_CELL_ROW_DEFICIT[t];
      for ( int64_t z = 0; z < rows; ++z )
         for ( int64_t y = 0; y < rows - z; ++y )
            for ( int64_t x = 0; x < rows - z - y; ++x )
            {
               int64_t idx[4];
               double  old[4];
               for ( int k = 0; k < 4; ++k )
               {
                  idx[k] = cell_index_w( N, x + MICRO_CELL_VERTS[t][k][0], y + MICRO_CELL_VERTS[t][k][1], z + MICRO_CELL_VERTS[t][k][2] );
                  old[k] = src[idx[k]];
               }
               for ( int k = 0; k < 4; ++k )
               {
                  double sum = 0.0;
                  for ( int j = 0; j < 4; ++j )
                     sum = sum + M[4 * k + j] * old[j];
                  dst[idx[k]] += sum;
               }
            }
   }
}

/* computeInverseDiagonalOperatorValues_macro_3D of the same operator: diag[v_i] += elMat_ii over all micro-cells */
HO_API void ho_p1_elementwise_diagonal_macro_3d( double* diag, const double* cell_coords12, int64_t micro_edges )
{
   int level = 0;
   while ( ( (int64_t) 1 << level ) < micro_edges )
      ++level;
   const int64_t N = ho_width( level ), n = N - 1;
   for ( int t = 0; t < 6; ++t )
   {
      double c[12], M[16];
      for ( int k = 0; k < 4; ++k )
         ho_coordinate_from_index( c + 3 * k, cell_coords12, level, MICRO_CELL_VERTS[t][k][0], MICRO_CELL_VERTS[t][k][1],
                                   MICRO_CELL_VERTS[t][k][2] );
      ho_p1_tet_diffusion( M, c );
      const int64_t rows = n - MICRO_CELL_ROW_DEFICIT[t];
      for ( int64_t z = 0; z < rows; ++z )
         for ( int64_t y = 0; y < rows - z; ++y )
            for ( int64_t x = 0; x < rows - z - y; ++x )
               for ( int k = 0; k < 4; ++k )
                  diag[cell_index_w( N, x + MICRO_CELL_VERTS[t][k][0], y + MICRO_CELL_VERTS[t][k][1], z + MICRO_CELL_VERTS[t][k][2] )] +=
                      M[4 * k + k];
   }
}

/* point class (slot 0..13, 14 inner) of the edge DoF with logical index (x,y,z) and orientation o: the macro-primitive that
 * contains both end points */
static const int EDGE_END_OFFSETS[7][2][3] = { { { 0, 0, 0 }, { 1, 0, 0 } }, { { 0, 0, 0 }, { 0, 1, 0 } }, { { 0, 0, 0 }, { 0, 0, 1 } },
                                               { { 1, 0, 0 }, { 0, 1, 0 } }, { { 1, 0, 0 }, { 0, 0, 1 } }, { { 0, 1, 0 }, { 0, 0, 1 } },
                                               { { 0, 1, 0 }, { 1, 0, 1 } } };
static int slot_from_face_flags( int f0, int f1, int f2, int f3 )
{
   const int cnt = f0 + f1 + f2 + f3;
   if ( cnt == 0 )
      return 14;
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
HO_API int ho_edge_dof_class( int level, int64_t x, int64_t y, int64_t z, int o )
{
   const int64_t N = ho_width( level );
   int           f[4] = { 1, 1, 1, 1 };
   for ( int e = 0; e < 2; ++e )
   {
      const int64_t px = x + EDGE_END_OFFSETS[o][e][0], py = y + EDGE_END_OFFSETS[o][e][1], pz = z + EDGE_END_OFFSETS[o][e][2];
      f[0] &= pz == 0, f[1] &= py == 0, f[2] &= px == 0, f[3] &= px + py + pz == N - 1;
   }
   return slot_from_face_flags( f[0], f[1], f[2], f[3] );
}

/* P2ElementwiseOperator::gemv on one macro-cell (P2ElementwiseOperator.cpp:110-223 with localMatrixVectorMultiply3D :66-107):
 * every micro-cell, type by type in iterator order, adds alpha * elMat * (its ten source values) to its ten destination DoFs.
 * Restated for the cell-centric storage: the scatter runs into zeroed temporaries (the reference zeroes dst where flagged and on
 * the whole halo first); then every DoF whose point class is in `mask` receives its sum (update 0) or has it added (update 1). */
HO_API void ho_p2_elementwise_apply_cell( double* dstV, double* dstE, const double* srcV, const double* srcE, int level,
                                          const double* elmat, double alpha, int update, unsigned mask )
{
   const int64_t N = ho_width( level ), n = N - 1;
   const int64_t nv = tet_size( N ), ne = ho_edge_array_size( level );
   double*       tv = (double*) calloc( (size_t) nv, sizeof( double ) );
   double*       te = (double*) calloc( (size_t) ( ne > 0 ? ne : 1 ), sizeof( double ) );
   for ( int t = 0; t < 6; ++t )
   {
      const int64_t rows = n - MICRO_CELL_ROW_DEFICIT[t];
      const double* M    = elmat + 100 * t;
      for ( int64_t z = 0; z < rows; ++z )
         for ( int64_t y = 0; y < rows - z; ++y )
            for ( int64_t x = 0; x < rows - z - y; ++x )
            {
               int64_t idx[10];
               double  old[10];
               ho_p2_micro_cell_dofs( level, t, x, y, z, idx );
               for ( int k = 0; k < 4; ++k )
                  old[k] = srcV[idx[k]];
               for ( int k = 4; k < 10; ++k )
                  old[k] = srcE[idx[k]];
               for ( int k = 0; k < 10; ++k )
               {
                  double s = 0.0;
                  for ( int j = 0; j < 10; ++j )
                     s = s + M[10 * k + j] * old[j];
                  if ( k < 4 )
                     tv[idx[k]] += alpha * s;
                  else
                     te[idx[k]] += alpha * s;
               }
            }
   }
   for ( int64_t z = 0; z < N; ++z )
      for ( int64_t y = 0; y < N - z; ++y )
         for ( int64_t x = 0; x < N - z - y; ++x )
         {
            const int slot = prim_slot( N, x, y, z );
            if ( !point_selected( mask, slot ) )
               continue;
            const int64_t i = cell_index_w( N, x, y, z );
            dstV[i]         = update ? dstV[i] + tv[i] : tv[i];
         }
   for ( int o = 0; o < 7; ++o )
   {
      const int64_t W = o == EO_XYZ ? n - 1 : n;
      for ( int64_t z = 0; z < W; ++z )
         for ( int64_t y = 0; y < W - z; ++y )
            for ( int64_t x = 0; x < W - z - y; ++x )
            {
               if ( !( ( mask >> ho_edge_dof_class( level, x, y, z, o ) ) & 1u ) )
                  continue;
               const int64_t i = ho_edge_index( level, x, y, z, o );
               dstE[i]         = update ? dstE[i] + te[i] : te[i];
            }
   }
   free( tv );
   free( te );
}
