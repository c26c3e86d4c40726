// mesh.hpp -- part of the C++ host layer above the C-ABI (see hyteg_host.hpp for the data model).
// MeshInfo: vertices + tetrahedra, Gmsh readers (src/hyteg/mesh/MeshInfo.hpp:221,512-585)
#pragma once

#include "types.hpp"

namespace hyteg {

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

} // namespace hyteg
