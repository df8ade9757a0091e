// scene_build.cpp -- the reference's spatial-median hierarchy, built straight into the flattened
// pre-order layout of layout.h.
//
// What is reproduced (bounds/KDNode.java:29-90, RTStatics.java:24-25,50-52,244-314):
//   * split while depth < MAX_DEPTH and faces > MAX_CHILDREN; plane = midpoint of the node's box
//     on the axis cycling X -> Y -> Z; a face goes left if its max <= plane, right if its
//     min >= plane, otherwise to BOTH sides; faces keep their stored order;
//   * a child's box is the tight box of its faces' vertices (not clipped to the plane); an empty
//     side is a null child;
//   * the root box covers ALL vertices, referenced or not (bounds/KDTree.java:35-44), and
//     shadowDistance = max span / 2000 (KDTree.java:46).
// What is different: the tree is never materialised as linked nodes.  Per-face bounds are computed
// once; each internal node becomes one record holding both children's boxes (80 bytes on the device), emitted in
// DFS pre-order, and leaf faces are appended to one slot array in DFS order.
#include "scene_build.hpp"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <utility>

#include "host_math.hpp"

namespace mi355rt {

namespace {

struct FaceBounds {
   float lo[3], hi[3];
};

class Flattener {
 public:
   Flattener(const std::vector<FaceBounds> &fb, const BuildOptions &opts, BuiltObject &out) : fb_(fb), opts_(opts), out_(out) {}

   // Emits the subtree holding `faces` (consumed) whose box is `box`; returns its link/cnt pair.
   void emit(std::vector<int32_t> &faces, const float lo[3], const float hi[3], int axis, int depth, int32_t &link, int32_t &cnt)
   {
      out_.max_depth = std::max(out_.max_depth, (int32_t) depth);
      const bool split = opts_.use_kdtree && depth < opts_.max_depth && faces.size() > (size_t) opts_.max_children;
      if (!split) {
         link = (int32_t) out_.slot_face.size();
         cnt = (int32_t) faces.size();
         out_.slot_face.insert(out_.slot_face.end(), faces.begin(), faces.end());
         for (size_t k = 0; k < faces.size(); k++) out_.slot_remaining.push_back((int32_t) (faces.size() - k));
         out_.n_leaves++;
         return;
      }
      const int32_t me = (int32_t) out_.nodes.size();
      out_.nodes.emplace_back();
      link = me;
      cnt = 0;

      const float plane = (hi[axis] - lo[axis]) / 2.0f + lo[axis];
      std::vector<int32_t> left, right;
      left.reserve(faces.size() / 2 + 8);
      right.reserve(faces.size() / 2 + 8);
      for (int32_t f : faces) {
         const FaceBounds &b = fb_[(size_t) f];
         if (b.hi[axis] <= plane) {
            left.push_back(f);
         } else if (b.lo[axis] >= plane) {
            right.push_back(f);
         } else {
            left.push_back(f);
            right.push_back(f);
         }
      }
      std::vector<int32_t>().swap(faces);

      NodeRec rec;
      for (int c = 0; c < 3; c++) {
         rec.lmin[c] = rec.rmin[c] = jm::kFloatMax;
         rec.lmax[c] = rec.rmax[c] = -jm::kFloatMax;
      }
      rec.llink = rec.rlink = LINK_NONE;
      rec.lcnt = rec.rcnt = 0;
      const int next_axis = (axis + 1) % 3;
      if (!left.empty()) {
         tight_box(left, rec.lmin, rec.lmax);
         emit(left, rec.lmin, rec.lmax, next_axis, depth + 1, rec.llink, rec.lcnt);
      }
      if (!right.empty()) {
         tight_box(right, rec.rmin, rec.rmax);
         emit(right, rec.rmin, rec.rmax, next_axis, depth + 1, rec.rlink, rec.rcnt);
      }
      out_.nodes[(size_t) me] = rec;
   }

 private:
   void tight_box(const std::vector<int32_t> &faces, float lo[3], float hi[3]) const
   {
      for (int32_t f : faces) {
         const FaceBounds &b = fb_[(size_t) f];
         for (int c = 0; c < 3; c++) {
            lo[c] = jm::fmin_java(lo[c], b.lo[c]);
            hi[c] = jm::fmax_java(hi[c], b.hi[c]);
         }
      }
   }

   const std::vector<FaceBounds> &fb_;
   const BuildOptions &opts_;
   BuiltObject &out_;
};

}   // namespace

namespace {

// Both record forms (NodeRec, DevNodeRec) carry { llink, rlink, lcnt, rcnt } at byte 48: the grouper sees only those.
struct Links {
   int32_t llink, rlink, lcnt, rcnt;
};

class Grouper {
 public:
   Grouper(void *records, size_t n_records, size_t stride, const int32_t *slot_face, GroupedLeaves &out)
      : rec_((unsigned char *) records), stride_(stride), slot_face_(slot_face), out_(out)
   {
      range_.resize(n_records);
   }

   // slot range [begin, end) of every subtree (leaves are laid out in DFS order, so a subtree's slots are one run)
   void ranges(int32_t link, int32_t cnt, int32_t &b, int32_t &e)
   {
      if (cnt > 0) {
         b = link;
         e = link + cnt;
         return;
      }
      const Links r = *at(link);
      int32_t lb = -1, le = -1, rb = -1, re = -1;
      if (r.llink != LINK_NONE) ranges(r.llink, r.lcnt, lb, le);
      if (r.rlink != LINK_NONE) ranges(r.rlink, r.rcnt, rb, re);
      b = lb >= 0 ? lb : rb;
      e = re >= 0 ? re : le;
      range_[(size_t) link] = { b, e };
   }

   // Walks the subtree under (link, cnt) in DFS order; returns the link the parent must store.
   int32_t assign(int32_t link, int32_t cnt)
   {
      if (cnt > 0) return leaf_alone(link, cnt);
      const int32_t b = range_[(size_t) link].first, e = range_[(size_t) link].second;
      if (e - b <= GROUP_SLOTS) {
         ids_.assign(slot_face_ + b, slot_face_ + e);
         std::sort(ids_.begin(), ids_.end());
         ids_.erase(std::unique(ids_.begin(), ids_.end()), ids_.end());
         if ((int) ids_.size() <= GROUP_FACES) {
            const int32_t gbase = (int32_t) out_.gface.size();
            out_.gface.insert(out_.gface.end(), ids_.begin(), ids_.end());
            out_.n_groups++;
            out_.n_shared++;
            in_group(link, gbase);
            return link;
         }
      }
      Links *r = at(link);
      if (r->llink != LINK_NONE) r->llink = assign(r->llink, r->lcnt);
      if (r->rlink != LINK_NONE) r->rlink = assign(r->rlink, r->rcnt);
      return link;
   }

 private:
   Links *at(int32_t node) const { return (Links *) (rec_ + stride_ * (size_t) node + 48); }

   // a leaf that is its own group: GROUP_FACES faces per descriptor, in slot order
   int32_t leaf_alone(int32_t first_slot, int32_t cnt)
   {
      const int32_t first_desc = (int32_t) out_.descs.size();
      for (int32_t done = 0; done < cnt; done += GROUP_FACES) {
         const int32_t n = std::min(GROUP_FACES, cnt - done);
         LeafDesc d;
         d.key = first_slot;
         d.gbase = (int32_t) out_.gface.size() | DESC_IDENT | (done + n < cnt ? DESC_MORE : 0);
         d.mask = n == 32 ? 0xffffffffu : ((1u << n) - 1u);
         d.pad = 0;
         out_.descs.push_back(d);
         out_.gface.insert(out_.gface.end(), slot_face_ + first_slot + done, slot_face_ + first_slot + done + n);
         out_.n_groups++;
      }
      return first_desc;
   }

   // every leaf under `link` becomes one descriptor: a mask over the group's faces (ids_, sorted), in DFS order
   void in_group(int32_t link, int32_t gbase)
   {
      Links *r = at(link);
      for (int side = 0; side < 2; side++) {
         int32_t &l = side ? r->rlink : r->llink;
         const int32_t c = side ? r->rcnt : r->lcnt;
         if (l == LINK_NONE) continue;
         if (c == 0) {
            in_group(l, gbase);
            continue;
         }
         LeafDesc d;
         d.key = l;
         d.gbase = gbase;
         d.mask = 0;
         d.pad = 0;
         for (int32_t k = 0; k < c; k++) {
            const int32_t f = slot_face_[(size_t) (l + k)];
            d.mask |= 1u << (uint32_t) (std::lower_bound(ids_.begin(), ids_.end(), f) - ids_.begin());
         }
         l = (int32_t) out_.descs.size();
         out_.descs.push_back(d);
      }
   }

   unsigned char *rec_;
   size_t stride_;
   const int32_t *slot_face_;
   GroupedLeaves &out_;
   std::vector<std::pair<int32_t, int32_t>> range_;
   std::vector<int32_t> ids_;
};

}   // namespace

void group_leaves(void *records, size_t n_records, size_t stride, const int32_t *slot_face, int32_t &root_link, int32_t root_cnt, GroupedLeaves &out)
{
   static_assert(offsetof(NodeRec, llink) == 48 && offsetof(DevNodeRec, llink) == 48, "the grouper reads the links at byte 48 of a record");
   out = GroupedLeaves();
   if (root_cnt == 0 && n_records == 0) return;   // an empty mesh: no leaf at all
   Grouper g(records, n_records, stride, slot_face, out);
   int32_t b, e;
   g.ranges(root_link, root_cnt, b, e);
   root_link = g.assign(root_link, root_cnt);
}

void synthesise_normals(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces, float *out)
{
   std::fill(out, out + 3 * (size_t) n_vertices, 0.0f);
   for (uint32_t f = 0; f < n_faces; f++) {
      const int32_t *face = indices + 3 * (size_t) f;
      const float *a = vertices + 3 * (size_t) face[0];
      const float *b = vertices + 3 * (size_t) face[1];
      const float *c = vertices + 3 * (size_t) face[2];
      // RTStatics.computeNormal (RTStatics.java:323-336)
      const jm::Vec3 e1 { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
      const jm::Vec3 e2 { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
      const jm::Vec3 n = jm::normalized(jm::cross(e1, e2));
      for (int k = 0; k < 3; k++) {   // unweighted sum, in face order
         float *dst = out + 3 * (size_t) face[k];
         dst[0] += n.x;
         dst[1] += n.y;
         dst[2] += n.z;
      }
   }
   for (uint32_t v = 0; v < n_vertices; v++) {
      float *d = out + 3 * (size_t) v;
      const float sq = (0.0f - d[0]) * (0.0f - d[0]) + (0.0f - d[1]) * (0.0f - d[1]) + (0.0f - d[2]) * (0.0f - d[2]);
      const float len = (float) std::sqrt((double) sq);
      d[0] /= len;
      d[1] /= len;
      d[2] /= len;
   }
}

std::string build_object(const float *vertices, uint32_t n_vertices, const float *normals, const int32_t *indices,
                         uint32_t n_faces, const BuildOptions &opts, BuiltObject &out)
{
   if ((n_vertices && !vertices) || (n_faces && !indices)) return "null vertex or index array";
   if (opts.max_depth < 0 || opts.max_depth > MAX_TREE_DEPTH) return "max_depth outside [0, " + std::to_string(MAX_TREE_DEPTH) + "]";
   if (opts.max_children < 0) return "max_children must be >= 0";
   for (size_t i = 0; i < 3 * (size_t) n_faces; i++) {
      if (indices[i] < 0 || (uint32_t) indices[i] >= n_vertices) return "face index out of range at element " + std::to_string(i);
   }

   out = BuiltObject();
   out.normals.resize(3 * (size_t) n_vertices);
   if (normals) std::copy(normals, normals + 3 * (size_t) n_vertices, out.normals.begin());
   else synthesise_normals(vertices, n_vertices, indices, n_faces, out.normals.data());

   for (int c = 0; c < 3; c++) {
      out.bmin[c] = jm::kFloatMax;
      out.bmax[c] = -jm::kFloatMax;
   }
   for (uint32_t v = 0; v < n_vertices; v++) {
      for (int c = 0; c < 3; c++) {
         out.bmin[c] = jm::fmin_java(vertices[3 * (size_t) v + c], out.bmin[c]);
         out.bmax[c] = jm::fmax_java(vertices[3 * (size_t) v + c], out.bmax[c]);
      }
   }
   out.shadow_distance =
       jm::fmax_java(out.bmax[0] - out.bmin[0], jm::fmax_java(out.bmax[1] - out.bmin[1], out.bmax[2] - out.bmin[2])) / 2000.0f;

   if (!opts.build_tree) return std::string();

   std::vector<FaceBounds> fb((size_t) n_faces);
   for (uint32_t f = 0; f < n_faces; f++) {
      FaceBounds &b = fb[f];
      for (int c = 0; c < 3; c++) {
         b.lo[c] = jm::kFloatMax;
         b.hi[c] = -jm::kFloatMax;
      }
      for (int k = 0; k < 3; k++) {
         const float *p = vertices + 3 * (size_t) indices[3 * (size_t) f + k];
         for (int c = 0; c < 3; c++) {
            b.lo[c] = jm::fmin_java(b.lo[c], p[c]);
            b.hi[c] = jm::fmax_java(b.hi[c], p[c]);
         }
      }
   }

   std::vector<int32_t> all((size_t) n_faces);
   for (uint32_t f = 0; f < n_faces; f++) all[f] = (int32_t) f;
   Flattener fl(fb, opts, out);
   fl.emit(all, out.bmin, out.bmax, 0, 0, out.root_link, out.root_cnt);
   out.root_is_leaf = out.nodes.empty() ? 1 : 0;
   if (out.slot_face.size() > (size_t) 0x7fffffff / 3) return "hierarchy too large: leaf slots exceed 32-bit addressing";
   return std::string();
}

}   // namespace mi355rt
