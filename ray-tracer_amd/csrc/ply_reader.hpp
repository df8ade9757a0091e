// ply_reader.hpp -- Stanford PLY reader with the reference loader's behaviour
// (sceneObjects/TriangleMesh.java:23-94,128-197).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mi355rt {

// The arrays are malloc()ed (what mi355rt_ply_load hands to its caller, who frees them with mi355rt_free: no copy of
// half a gigabyte in between); the struct frees what is still its own.
struct PlyMesh {
   float *vertices = nullptr;     // n_vertices * 3
   float *normals = nullptr;      // n_vertices * 3 (zeros when the file has none)
   int32_t *indices = nullptr;    // n_faces * 3
   size_t n_vertices = 0, n_faces = 0;
   bool has_normals = false;
   PlyMesh() = default;
   PlyMesh(const PlyMesh &) = delete;
   PlyMesh &operator=(const PlyMesh &) = delete;
   ~PlyMesh() { clear(); }
   void clear();
};

// Returns an empty string on success, else what went wrong.
std::string read_ply(const char *path, PlyMesh &out);

}   // namespace mi355rt
