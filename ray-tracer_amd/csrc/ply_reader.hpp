// ply_reader.hpp -- Stanford PLY reader with the reference loader's behaviour
// (sceneObjects/TriangleMesh.java:23-94,128-197).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mi355rt {

struct PlyMesh {
   std::vector<float> vertices;   // n * 3
   std::vector<float> normals;    // n * 3 (zeros when the file has none)
   std::vector<int32_t> indices;  // f * 3
   bool has_normals = false;
};

// Returns an empty string on success, else what went wrong.
std::string read_ply(const char *path, PlyMesh &out);

}   // namespace mi355rt
