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

#include "host_parallel.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <exception>
#include <mutex>
#include <thread>
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

// slot range [begin, end) of every subtree (leaves are laid out in DFS order, so a subtree's slots are one run)
struct Ranges {
   Ranges(const unsigned char *rec, size_t stride, size_t n_records) : rec_(rec), stride_(stride), range(n_records) {}
   void find(int32_t link, int32_t cnt, int32_t &b, int32_t &e)
   {
      if (cnt > 0) {
         b = link;
         e = link + cnt;
         return;
      }
      const Links r = *(const Links *) (rec_ + stride_ * (size_t) link + 48);
      int32_t lb = -1, le = -1, rb = -1, re = -1;
      if (r.llink != LINK_NONE) find(r.llink, r.lcnt, lb, le);
      if (r.rlink != LINK_NONE) find(r.rlink, r.rcnt, rb, re);
      b = lb >= 0 ? lb : rb;
      e = re >= 0 ? re : le;
      range[(size_t) link] = { b, e };
   }
   const unsigned char *rec_;
   size_t stride_;
   std::vector<std::pair<int32_t, int32_t>> range;
};

// Groups the leaves of one subtree.  Run twice: `dry` (nothing is written: how many descriptors and group faces the
// subtree makes), then for real with the places those counts give it in the object's arrays -- so that disjoint
// subtrees can be done by different threads and the arrays still come out in DFS order.
class Grouper {
 public:
   Grouper(void *records, size_t stride, const int32_t *slot_face, const Ranges &ranges, bool dry, LeafDesc *descs, int32_t *gface,
           int32_t desc0, int32_t gface0)
      : n_desc(desc0), n_gface(gface0), rec_((unsigned char *) records), stride_(stride), slot_face_(slot_face), range_(ranges.range), dry_(dry),
        descs_(descs), gface_(gface)
   {
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
            const int32_t gbase = n_gface;
            if (!dry_) std::copy(ids_.begin(), ids_.end(), gface_ + n_gface);
            n_gface += (int32_t) ids_.size();
            n_groups++;
            n_shared++;
            in_group(link, gbase);
            return link;
         }
      }
      Links *r = at(link);
      if (r->llink != LINK_NONE) {
         const int32_t l = assign(r->llink, r->lcnt);
         if (!dry_) r->llink = l;
      }
      if (r->rlink != LINK_NONE) {
         const int32_t l = assign(r->rlink, r->rcnt);
         if (!dry_) r->rlink = l;
      }
      return link;
   }

   int32_t n_desc, n_gface;   // next free descriptor / group-face entry
   uint64_t n_groups = 0, n_shared = 0;

 private:
   Links *at(int32_t node) const { return (Links *) (rec_ + stride_ * (size_t) node + 48); }

   // a leaf that is its own group: GROUP_FACES faces per descriptor, in slot order
   int32_t leaf_alone(int32_t first_slot, int32_t cnt)
   {
      const int32_t first_desc = n_desc;
      for (int32_t done = 0; done < cnt; done += GROUP_FACES) {
         const int32_t n = std::min(GROUP_FACES, cnt - done);
         if (!dry_) {
            LeafDesc d;
            d.key = first_slot;
            d.gbase = n_gface | DESC_IDENT | (done + n < cnt ? DESC_MORE : 0);
            d.mask = n == 32 ? 0xffffffffu : ((1u << n) - 1u);
            d.pad = 0;
            descs_[n_desc] = d;
            std::copy(slot_face_ + first_slot + done, slot_face_ + first_slot + done + n, gface_ + n_gface);
         }
         n_desc++;
         n_gface += n;
         n_groups++;
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
         if (!dry_) {
            LeafDesc d;
            d.key = l;
            d.gbase = gbase;
            d.mask = 0;
            d.pad = 0;
            for (int32_t k = 0; k < c; k++) {
               const int32_t f = slot_face_[(size_t) (l + k)];
               d.mask |= 1u << (uint32_t) (std::lower_bound(ids_.begin(), ids_.end(), f) - ids_.begin());
            }
            descs_[n_desc] = d;
            l = n_desc;
         }
         n_desc++;
      }
   }

   unsigned char *rec_;
   size_t stride_;
   const int32_t *slot_face_;
   const std::vector<std::pair<int32_t, int32_t>> &range_;
   bool dry_;
   LeafDesc *descs_;
   int32_t *gface_;
   std::vector<int32_t> ids_;
};

// a subtree small enough to be one thread's piece of work, and where its (possibly rewritten) link goes
struct GroupTask {
   int32_t link, cnt;
   int32_t *parent_slot;
   int32_t n_desc = 0, n_gface = 0;
   uint64_t n_groups = 0, n_shared = 0;
};

void collect_tasks(unsigned char *rec, size_t stride, const Ranges &ranges, int32_t piece, int32_t link, int32_t cnt, int32_t *parent_slot,
                   std::vector<GroupTask> &tasks)
{
   if (cnt > 0 || ranges.range[(size_t) link].second - ranges.range[(size_t) link].first <= piece) {
      GroupTask t;
      t.link = link;
      t.cnt = cnt;
      t.parent_slot = parent_slot;
      tasks.push_back(t);
      return;
   }
   Links *r = (Links *) (rec + stride * (size_t) link + 48);   // (more slots than a group may span: never a group itself)
   if (r->llink != LINK_NONE) collect_tasks(rec, stride, ranges, piece, r->llink, r->lcnt, &r->llink, tasks);
   if (r->rlink != LINK_NONE) collect_tasks(rec, stride, ranges, piece, r->rlink, r->rcnt, &r->rlink, tasks);
}

// body(i) for i in [0, n) on up to 16 threads.  What a body throws (an allocation failure) is rethrown here, on the
// caller's thread, after every worker has stopped: nothing may leave a worker thread.
template <typename F> void run_tasks(size_t n, F &&body)
{
   const unsigned hw = std::thread::hardware_concurrency();
   const size_t n_threads = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 16), n / 4 + 1);
   if (n_threads <= 1) {
      for (size_t i = 0; i < n; i++) body(i);
      return;
   }
   std::atomic<size_t> next(0);
   std::atomic<bool> failed(false);
   std::exception_ptr error;
   std::mutex error_lock;
   auto work = [&]() {
      try {
         for (size_t i = next.fetch_add(1); i < n && !failed.load(); i = next.fetch_add(1)) body(i);
      } catch (...) {
         std::lock_guard<std::mutex> hold(error_lock);
         if (!error) error = std::current_exception();
         failed.store(true);
      }
   };
   std::vector<std::thread> pool;
   try {
      for (size_t t = 1; t < n_threads; t++) pool.emplace_back(work);
   } catch (...) {   // no more threads to be had: the ones there are, and this one, do the work
   }
   work();
   for (std::thread &t : pool) t.join();
   if (error) std::rethrow_exception(error);
}

}   // namespace

void group_leaves(void *records, size_t n_records, size_t stride, const int32_t *slot_face, int32_t &root_link, int32_t root_cnt, GroupedLeaves &out)
{
   static_assert(offsetof(NodeRec, llink) == 48 && offsetof(DevNodeRec, llink) == 48, "the grouper reads the links at byte 48 of a record");
   out = GroupedLeaves();
   if (root_cnt == 0 && n_records == 0) return;   // an empty mesh: no leaf at all
   unsigned char *rec = (unsigned char *) records;
   Ranges ranges(rec, stride, n_records);
   int32_t b = 0, e = 0;
   ranges.find(root_link, root_cnt, b, e);
   // Disjoint subtrees are grouped independently (a group never spans more than GROUP_SLOTS slots, so cutting the tree
   // into pieces of at least that many changes nothing): a dry run per piece for its counts, a prefix sum for its place
   // in the arrays, then the real run.  The pieces depend on the tree alone, not on the number of threads.
   const int32_t piece = std::max<int32_t>(GROUP_SLOTS, (e - b) / 1024);
   std::vector<GroupTask> tasks;
   collect_tasks(rec, stride, ranges, piece, root_link, root_cnt, &root_link, tasks);
   run_tasks(tasks.size(), [&](size_t i) {
      GroupTask &t = tasks[i];
      Grouper g(records, stride, slot_face, ranges, true, nullptr, nullptr, 0, 0);
      (void) g.assign(t.link, t.cnt);
      t.n_desc = g.n_desc;
      t.n_gface = g.n_gface;
   });
   std::vector<int32_t> desc0(tasks.size() + 1, 0), gface0(tasks.size() + 1, 0);
   for (size_t i = 0; i < tasks.size(); i++) {
      desc0[i + 1] = desc0[i] + tasks[i].n_desc;
      gface0[i + 1] = gface0[i] + tasks[i].n_gface;
   }
   out.descs.resize((size_t) desc0.back());
   out.gface.resize((size_t) gface0.back());
   run_tasks(tasks.size(), [&](size_t i) {
      GroupTask &t = tasks[i];
      Grouper g(records, stride, slot_face, ranges, false, out.descs.data(), out.gface.data(), desc0[i], gface0[i]);
      *t.parent_slot = g.assign(t.link, t.cnt);
      t.n_groups = g.n_groups;
      t.n_shared = g.n_shared;
   });
   for (const GroupTask &t : tasks) {
      out.n_groups += t.n_groups;
      out.n_shared += t.n_shared;
   }
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
   {   // (flat passes over tens of millions of elements: on the host threads there are; the first bad element is reported
       // and the bounds are combined with Java's min / max, so the answers are those of the one-thread loops)
      constexpr size_t NONE = ~(size_t) 0;
      std::vector<size_t> bad(host_threads_for(3 * (size_t) n_faces, (size_t) 1 << 20), NONE);
      in_ranges(3 * (size_t) n_faces, (size_t) 1 << 20, [&](size_t t, size_t first, size_t last) {
         for (size_t i = first; i < last; i++)
            if (indices[i] < 0 || (uint32_t) indices[i] >= n_vertices) {
               bad[t] = i;
               return;
            }
      });
      for (size_t b : bad)
         if (b != NONE) return "face index out of range at element " + std::to_string(b);
   }

   out = BuiltObject();
   // (a part the GPU builds gets or keeps its normals there: opts.synth_normals is off and nothing here reads them)
   if (opts.synth_normals) {
      out.normals.resize(3 * (size_t) n_vertices);
      if (normals) std::copy(normals, normals + 3 * (size_t) n_vertices, out.normals.begin());
      else synthesise_normals(vertices, n_vertices, indices, n_faces, out.normals.data());
   }

   {
      struct Box {
         float lo[3], hi[3];
      };
      std::vector<Box> boxes(host_threads_for((size_t) n_vertices, (size_t) 1 << 19));
      in_ranges((size_t) n_vertices, (size_t) 1 << 19, [&](size_t t, size_t first, size_t last) {
         Box b;
         for (int c = 0; c < 3; c++) {
            b.lo[c] = jm::kFloatMax;
            b.hi[c] = -jm::kFloatMax;
         }
         for (size_t v = first; v < last; v++) {
            for (int c = 0; c < 3; c++) {
               b.lo[c] = jm::fmin_java(vertices[3 * v + c], b.lo[c]);
               b.hi[c] = jm::fmax_java(vertices[3 * v + c], b.hi[c]);
            }
         }
         boxes[t] = b;
      });
      for (int c = 0; c < 3; c++) {
         out.bmin[c] = jm::kFloatMax;
         out.bmax[c] = -jm::kFloatMax;
      }
      for (const Box &b : boxes) {   // in range order: min / max of the same values, NaN sticky as in the one loop
         for (int c = 0; c < 3; c++) {
            out.bmin[c] = jm::fmin_java(b.lo[c], out.bmin[c]);
            out.bmax[c] = jm::fmax_java(b.hi[c], out.bmax[c]);
         }
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
