// ply_reader.cpp -- whole-file, cursor-based PLY parse that yields what the reference loader yields
// (sceneObjects/TriangleMesh.java).  Behaviours kept on purpose:
//   * header: "element vertex N", "element face N"; a property's slot is its rank among ALL non-list
//     property lines; slots are recognised by line suffix, " nx"/" ny"/" nz" first, then " x"/" y"/" z"
//     (TriangleMesh.java:41-70);
//   * ASCII body: fields separated by exactly one space (String.split(" ")), faces use fields 1..3
//     and run to end of file (:128-153);
//   * binary body: big-endian only (DataInputStream), every vertex property a 4-byte float, one
//     list-count byte skipped per face, three 4-byte indices (:155-197).
// Where the reference would die with an exception (short file, unparsable field) this returns an
// error string instead of exiting the process (TriangleMesh.java:90-93).
#include "ply_reader.hpp"

#include "host_parallel.hpp"

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace mi355rt {

namespace {

struct Cursor {
   const char *p, *end;
   // one text line as BufferedReader.readLine sees it; false at end of data
   bool line(const char *&b, const char *&e)
   {
      if (p >= end) return false;
      b = p;
      while (p < end && *p != '\n' && *p != '\r') ++p;
      e = p;
      if (p < end) {
         if (*p == '\r' && p + 1 < end && p[1] == '\n') p += 2;
         else ++p;
      }
      return true;
   }
};

bool has_prefix(const char *b, const char *e, const char *lit)
{
   const size_t n = std::strlen(lit);
   return (size_t) (e - b) >= n && std::memcmp(b, lit, n) == 0;
}
bool has_suffix(const char *b, const char *e, const char *lit)
{
   const size_t n = std::strlen(lit);
   return (size_t) (e - b) >= n && std::memcmp(e - n, lit, n) == 0;
}

// fields of a line split on single spaces, trailing empty fields dropped
void fields(const char *b, const char *e, std::vector<std::string> &out)
{
   out.clear();
   const char *s = b;
   for (const char *q = b;; ++q) {
      if (q == e || *q == ' ') {
         out.emplace_back(s, q);
         if (q == e) break;
         s = q + 1;
      }
   }
   while (!out.empty() && out.back().empty()) out.pop_back();
}

bool to_float(const std::string &s, float &v)
{
   if (s.empty()) return false;
   char *endp = nullptr;
   v = std::strtof(s.c_str(), &endp);
   return endp && *endp == 0;
}
bool to_int(const std::string &s, int32_t &v)
{
   if (s.empty() || s[0] == ' ' || s[0] == '\t') return false;
   char *endp = nullptr;
   errno = 0;
   const long x = std::strtol(s.c_str(), &endp, 10);
   if (!endp || *endp != 0 || errno || x < INT32_MIN || x > INT32_MAX) return false;
   v = (int32_t) x;
   return true;
}

uint32_t be32(const unsigned char *q)
{
   uint32_t v;
   std::memcpy(&v, q, 4);
   return __builtin_bswap32(v);
}

// The file as one read-only range: mapped (a 533 MB Lucy-class file is neither read into a buffer nor copied while one
// grows), or read whole where mapping is refused (pipes, odd file systems).
struct FileBytes {
   const char *p = nullptr;
   size_t n = 0;
   void *map = nullptr;
   std::string owned;
   ~FileBytes()
   {
      if (map) munmap(map, n);
   }
   bool open(const char *path)
   {
      const int fd = ::open(path, O_RDONLY);
      if (fd < 0) return false;
      struct stat sb;
      if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
         void *m = mmap(nullptr, (size_t) sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
         if (m != MAP_FAILED) {
            (void) madvise(m, (size_t) sb.st_size, MADV_SEQUENTIAL);
            map = m;
            p = (const char *) m;
            n = (size_t) sb.st_size;
            ::close(fd);
            return true;
         }
      }
      char chunk[1 << 16];
      ssize_t got;
      while ((got = ::read(fd, chunk, sizeof(chunk))) > 0) owned.append(chunk, (size_t) got);
      ::close(fd);
      p = owned.data();
      n = owned.size();
      return true;
   }
};

}   // namespace

void PlyMesh::clear()
{
   std::free(vertices);
   std::free(normals);
   std::free(indices);
   vertices = normals = nullptr;
   indices = nullptr;
   n_vertices = n_faces = 0;
   has_normals = false;
}

std::string read_ply(const char *path, PlyMesh &out)
{
   FileBytes file;
   if (!file.open(path)) return "cannot open file";
   struct {
      const char *p;
      size_t n;
      const char *data() const { return p; }
      size_t size() const { return n; }
   } data { file.p, file.n };

   Cursor cur { data.data(), data.data() + data.size() };
   const char *b, *e;
   int32_t n_vertices = 0, n_faces = 0;
   int slot = 0, sx = -1, sy = -1, sz = -1, snx = -1, sny = -1, snz = -1;
   bool ascii = true, in_body = false;
   std::vector<std::string> fl;

   while (!in_body && cur.line(b, e)) {
      if (has_prefix(b, e, "element vertex")) {
         fields(b, e, fl);
         if (fl.size() < 3 || !to_int(fl[2], n_vertices) || n_vertices < 0) return "bad 'element vertex' line";
      } else if (has_prefix(b, e, "element face")) {
         fields(b, e, fl);
         if (fl.size() < 3 || !to_int(fl[2], n_faces) || n_faces < 0) return "bad 'element face' line";
      } else if (has_prefix(b, e, "property")) {
         if (has_suffix(b, e, " nx")) snx = slot;
         else if (has_suffix(b, e, " ny")) sny = slot;
         else if (has_suffix(b, e, " nz")) snz = slot;
         else if (has_suffix(b, e, " x")) sx = slot;
         else if (has_suffix(b, e, " y")) sy = slot;
         else if (has_suffix(b, e, " z")) sz = slot;
         if (!has_prefix(b, e, "property list")) slot++;
      } else if (has_prefix(b, e, "format binary")) {
         ascii = false;
      } else if ((size_t) (e - b) == 10 && std::memcmp(b, "end_header", 10) == 0) {
         in_body = true;
      }
   }
   if (sx < 0 || sy < 0 || sz < 0) return "header declares no x/y/z vertex properties";
   const bool has_n = snx >= 0 && sny >= 0 && snz >= 0;

   // The header's counts size the arrays, so they are checked against what the file can hold BEFORE anything is
   // allocated (a 100-byte file declaring two billion vertices must be an error, not a 48 GB request).  Binary:
   // the exact block sizes.  ASCII: every vertex line and every face line is at least one byte long; a file with
   // fewer face lines than declared keeps zero rows for the rest, as the reference's int[numFaces][3] does.
   const char *body = nullptr, *body_end = data.data() + data.size();
   if (ascii) {
      body = cur.p;
   } else {
      // the reference re-opens the file and skips '\n'-terminated lines up to the one that starts with "end_header"
      const char *q = data.data();
      for (;;) {
         const char *nl = (const char *) std::memchr(q, '\n', (size_t) (body_end - q));
         if (!nl) return "binary file has no end_header line";
         const bool last = has_prefix(q, nl, "end_header");
         q = nl + 1;
         if (last) break;
      }
      body = q;
   }
   const size_t body_bytes = (size_t) (body_end - body);
   if (ascii) {
      if ((size_t) n_vertices > body_bytes) return "'element vertex' declares more vertices than the file has bytes left";
      if ((size_t) n_faces > body_bytes) return "'element face' declares more faces than the file has bytes left";
   } else {
      if ((size_t) n_vertices * (size_t) slot * 4 > body_bytes) return "file ends inside the vertex block";
      if ((size_t) n_faces * 13 > body_bytes - (size_t) n_vertices * (size_t) slot * 4) return "file ends inside the face block";
   }

   // (binary: every element is written below; ASCII: face rows the file does not have stay zero; normals the file does not
   // have are zeros: calloc's untouched pages cost nothing)
   out.clear();
   out.has_normals = has_n;
   out.n_vertices = (size_t) n_vertices;
   out.n_faces = (size_t) n_faces;
   const size_t nv3 = std::max<size_t>(3 * (size_t) n_vertices, 1), nf3 = std::max<size_t>(3 * (size_t) n_faces, 1);
   out.vertices = (float *) (ascii ? std::calloc(nv3, sizeof(float)) : std::malloc(nv3 * sizeof(float)));
   out.normals = (float *) (ascii || !has_n ? std::calloc(nv3, sizeof(float)) : std::malloc(nv3 * sizeof(float)));
   out.indices = (int32_t *) (ascii ? std::calloc(nf3, sizeof(int32_t)) : std::malloc(nf3 * sizeof(int32_t)));
   if (!out.vertices || !out.normals || !out.indices) {
      out.clear();
      return "out of host memory";
   }

   if (ascii) {
      for (int32_t i = 0; i < n_vertices; i++) {
         if (!cur.line(b, e)) return "file ends inside the vertex list";
         fields(b, e, fl);
         const int need = has_n ? std::max(std::max(sx, sy), std::max(std::max(sz, snx), std::max(sny, snz))) : std::max(sx, std::max(sy, sz));
         if ((int) fl.size() <= need) return "vertex line " + std::to_string(i) + " has too few fields";
         float *v = &out.vertices[3 * (size_t) i];
         if (!to_float(fl[(size_t) sx], v[0]) || !to_float(fl[(size_t) sy], v[1]) || !to_float(fl[(size_t) sz], v[2]))
            return "unparsable coordinate on vertex line " + std::to_string(i);
         if (has_n) {
            float *n = &out.normals[3 * (size_t) i];
            if (!to_float(fl[(size_t) snx], n[0]) || !to_float(fl[(size_t) sny], n[1]) || !to_float(fl[(size_t) snz], n[2]))
               return "unparsable normal on vertex line " + std::to_string(i);
         }
      }
      int32_t row = 0;
      while (cur.line(b, e)) {
         fields(b, e, fl);
         if (row >= n_faces) return "more face lines than 'element face' declares";
         if (fl.size() < 4) return "face line " + std::to_string(row) + " has fewer than 4 fields";
         int32_t *t = &out.indices[3 * (size_t) row];
         if (!to_int(fl[1], t[0]) || !to_int(fl[2], t[1]) || !to_int(fl[3], t[2])) return "unparsable index on face line " + std::to_string(row);
         row++;
      }
      return std::string();
   }

   // binary body (sizes checked above)
   const char *q = body, *qe = body_end;
   const size_t vbytes = (size_t) n_vertices * (size_t) slot * 4;
   const size_t fbytes = (size_t) n_faces * 13;
   if ((size_t) (qe - q) < vbytes) return "file ends inside the vertex block";
   const unsigned char *u0 = (const unsigned char *) q;
   const size_t stride = (size_t) slot * 4;
   float *vout = out.vertices, *nout = out.normals;
   in_ranges((size_t) n_vertices, (size_t) 1 << 18, [=](size_t, size_t first, size_t last) {
      const int src[3] = { sx, sy, sz }, nsrc[3] = { snx, sny, snz };
      const unsigned char *u = u0 + first * stride;
      for (size_t i = first; i < last; i++, u += stride) {
         for (int c = 0; c < 3; c++) {
            const uint32_t bits = be32(u + 4 * (size_t) src[c]);
            std::memcpy(vout + 3 * i + c, &bits, 4);
         }
         if (has_n) {
            for (int c = 0; c < 3; c++) {
               const uint32_t bits = be32(u + 4 * (size_t) nsrc[c]);
               std::memcpy(nout + 3 * i + c, &bits, 4);
            }
         }
      }
   });
   const unsigned char *f0 = u0 + vbytes;
   if ((size_t) ((const unsigned char *) qe - f0) < fbytes) return "file ends inside the face block";
   int32_t *iout = out.indices;
   in_ranges((size_t) n_faces, (size_t) 1 << 18, [=](size_t, size_t first, size_t last) {
      const unsigned char *u = f0 + first * 13;
      for (size_t i = first; i < last; i++, u += 13)
         for (int c = 0; c < 3; c++) iout[3 * i + c] = (int32_t) be32(u + 1 + 4 * c);
   });
   return std::string();
}

}   // namespace mi355rt
