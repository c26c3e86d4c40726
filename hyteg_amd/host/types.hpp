// types.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// enums of src/hyteg/types/types.hpp, error plumbing, the macro-cell layout algebra
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

} // namespace hyteg
