// timing.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// Timing tree of the hot path with the reference's timer names and nesting, and roctx ranges of the same names.
//   walberla::WcTimingTree threaded through PrimitiveStorage::getTimingTree(); Operator::startTiming / stopTiming nest
//   "Operator <Src> to <Dst>" / "Apply" / "Macro-Cell" ... (src/hyteg/operators/Operator.hpp:148-166,
//   src/hyteg/p1functionspace/P1Operator.hpp:200-319,366-417,436-446); functions: "Assign", "Add", "Dot (local)",
//   "Dot (reduce)", "Interpolate", "Multiply elementwise" (VertexDoFFunction.cpp:386,1137,1226,1492,1713-1722); multigrid:
//   "Geometric Multigrid Solver" / "Level L" / "Smoother" | "Restriction" | "Prolongation" | "Coarse Grid Solver"
//   (GeometricMultigridSolver.hpp:200-300); JSON dump: src/hyteg/dataexport/TimingOutput.hpp:46-60.
// The JSON layout follows walberla::timing::to_json of core/timing/TimingJSON.h (every node: "total", "average",
// "count", "min", "max", "variance", children keyed by timer name); waLBerla is an empty submodule in the reference
// snapshot, so that layout is restated from its documentation, not checked against its code ("parity unpinned").
//
// Device work is asynchronous: by default a timer measures the time the host needs to ENQUEUE the work of a range.
// With setSynchronize( true ) every stop() first waits for the storage's stream, so the ranges measure execution
// (profiling mode: it serialises host and device).  HYTEG_AMD_ROCTX=1 additionally emits roctxRangePush / Pop with the
// timer names, for `rocprofv3 --marker-trace`.
#pragma once

#include <chrono>
#include <dlfcn.h>

#include "types.hpp"

namespace hyteg {

class TimingTree
{
 public:
   struct Node
   {
      uint_t                        count = 0;
      double                        total = 0.0, minimum = 0.0, maximum = 0.0, sumOfSquares = 0.0;
      std::map< std::string, Node > children;
   };

   void setSynchronize( bool on, hyteg_hip_stream_t stream )
   {
      sync_   = on;
      stream_ = stream;
   }
   void setStream( hyteg_hip_stream_t stream ) { stream_ = stream; }

   void start( const std::string& name )
   {
      Node& n = current().children[name];
      stack_.push_back( { &n, name, Clock::now() } );
      if ( roctx().push )
         roctx().push( name.c_str() );
   }
   void stop( const std::string& name )
   {
      if ( stack_.empty() || stack_.back().name != name )
         throw std::runtime_error( "TimingTree::stop( \"" + name + "\" ): not the innermost running timer" );
      if ( sync_ )
         hyteg_hip_stream_synchronize( stream_ );
      if ( roctx().pop )
         roctx().pop();
      const double dt = std::chrono::duration< double >( Clock::now() - stack_.back().t0 ).count();
      Node&        n  = *stack_.back().node;
      n.minimum       = n.count == 0 ? dt : std::min( n.minimum, dt );
      n.maximum       = n.count == 0 ? dt : std::max( n.maximum, dt );
      n.count += 1;
      n.total += dt;
      n.sumOfSquares += dt * dt;
      stack_.pop_back();
   }
   void reset()
   {
      if ( !stack_.empty() )
         throw std::runtime_error( "TimingTree::reset: timers are running" );
      root_ = Node{};
   }
   const Node& root() const { return root_; }

   // walberla::timing::to_json layout (see the header comment), indented by 4 like TimingOutput.hpp:56
   std::string toJSON() const
   {
      std::ostringstream os;
      os.precision( 17 );
      os << "{\n";
      writeChildren( os, root_, 1 );
      os << "\n}\n";
      return os.str();
   }

 private:
   using Clock = std::chrono::steady_clock;
   struct Running
   {
      Node*             node;
      std::string       name;
      Clock::time_point t0;
   };
   struct Roctx
   {
      int ( *push )( const char* ) = nullptr;
      int ( *pop )()               = nullptr;
   };
   static Roctx& roctx()
   {
      static Roctx r = [] {
         Roctx       x;
         const char* e = std::getenv( "HYTEG_AMD_ROCTX" );
         if ( e && e[0] == '1' )
            for ( const char* lib : { "libroctx64.so.4", "libroctx64.so", "librocprofiler-sdk-roctx.so.1" } )
               if ( void* h = dlopen( lib, RTLD_NOW | RTLD_GLOBAL ) )
               {
                  x.push = reinterpret_cast< int ( * )( const char* ) >( dlsym( h, "roctxRangePushA" ) );
                  x.pop  = reinterpret_cast< int ( * )() >( dlsym( h, "roctxRangePop" ) );
                  if ( x.push && x.pop )
                     break;
                  x = Roctx{};
               }
         return x;
      }();
      return r;
   }
   Node& current() { return stack_.empty() ? root_ : *stack_.back().node; }
   static void writeEscaped( std::ostringstream& os, const std::string& s )
   {
      os << '"';
      for ( char c : s )
      {
         if ( c == '"' || c == '\\' )
            os << '\\';
         os << c;
      }
      os << '"';
   }
   static void writeNode( std::ostringstream& os, const Node& n, int depth )
   {
      const std::string pad( 4 * (size_t) depth, ' ' );
      const double      avg = n.count ? n.total / (double) n.count : 0.0;
      const double      var = n.count ? std::max( 0.0, n.sumOfSquares / (double) n.count - avg * avg ) : 0.0;
      os << pad << "\"total\": " << n.total << ",\n"
         << pad << "\"average\": " << avg << ",\n"
         << pad << "\"count\": " << n.count << ",\n"
         << pad << "\"min\": " << n.minimum << ",\n"
         << pad << "\"max\": " << n.maximum << ",\n"
         << pad << "\"variance\": " << var;
      if ( !n.children.empty() )
      {
         os << ",\n";
         writeChildren( os, n, depth );
      }
   }
   static void writeChildren( std::ostringstream& os, const Node& n, int depth )
   {
      const std::string pad( 4 * (size_t) depth, ' ' );
      bool              first = true;
      for ( const auto& kv : n.children )
      {
         if ( !first )
            os << ",\n";
         first = false;
         os << pad;
         writeEscaped( os, kv.first );
         os << ": {\n";
         writeNode( os, kv.second, depth + 1 );
         os << "\n" << pad << "}";
      }
   }

   Node                   root_;
   std::vector< Running > stack_;
   bool                   sync_   = false;
   hyteg_hip_stream_t     stream_ = nullptr;
};

// start / stop pair tied to a scope; a null tree (timing disabled, the default) costs one pointer test
class ScopedTimer
{
 public:
   ScopedTimer( TimingTree* tree, const char* name )
   : tree_( tree )
   , name_( name )
   {
      if ( tree_ )
         tree_->start( name_ );
   }
   ScopedTimer( TimingTree* tree, const std::string& name )
   : tree_( tree )
   , owned_( name )
   , name_( owned_.c_str() )
   {
      if ( tree_ )
         tree_->start( owned_ );
   }
   ~ScopedTimer()
   {
      if ( tree_ )
      {
         try
         {
            tree_->stop( name_ );
         } catch ( ... )
         {}
      }
   }
   ScopedTimer( const ScopedTimer& )            = delete;
   ScopedTimer& operator=( const ScopedTimer& ) = delete;

 private:
   TimingTree* tree_;
   std::string owned_;
   const char* name_;
};

} // namespace hyteg
