// api.cpp -- the C ABI of libmi355rt.so (include/mi355rt.h): handles, error reporting, host-side
// set-up (camera constants, BRDF table remap) and HIP memory / launch plumbing.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl.so is loaded on demand (multi_gpu.inl), never linked

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mi355rt.h"
#include "host_math.hpp"
#include "gpu_build.hpp"
#include "kernels.h"
#include "layout.h"
#include "ply_reader.hpp"
#include "scene_build.hpp"
#include "host_parallel.hpp"
#include "switches.hpp"

using namespace mi355rt;

// The frame scratch of mi355rt_render_device_noalloc (include/mi355rt.h): a file-scope device array instead of
// hipMalloc'ed buffers.  Every GPU the library runs on holds its own copy at its own address until the process ends.
// In host code the NAME is only a stand-in for the symbol, never a device address: noalloc_scratch_ptr() resolves it
// once per device with hipGetSymbolAddress, and host code uses that pointer wherever it needs the memory.
static __device__ __align__(256) unsigned char scratch[MI355RT_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
void *g_scratch_ptr[MAX_DEVICES] = { nullptr };

// the address of `scratch` on `device`, which must be the current device; resolved on first use per device
hipError_t noalloc_scratch_ptr(int device, void **out)
{
   *out = nullptr;
   if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
   std::lock_guard<std::mutex> lock(g_scratch_mutex);
   if (!g_scratch_ptr[device]) {
      void *ptr = nullptr;
      const hipError_t e = hipGetSymbolAddress((void **) &ptr, HIP_SYMBOL(scratch));
      if (e != hipSuccess) return e;
      if (!ptr) return hipErrorInvalidSymbol;
      g_scratch_ptr[device] = ptr;
   }
   *out = g_scratch_ptr[device];
   return hipSuccess;
}

}   // namespace

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg)
{
   g_error = msg;
   return code;
}

int hip_fail(hipError_t e, const char *what)
{
   return fail(e == hipErrorOutOfMemory ? MI355RT_E_MEMORY : MI355RT_E_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

// No exception may leave an extern "C" function (it would terminate the host process -- a JVM behind the JNI shim):
// every entry point's body runs inside guarded().
template <typename F> int guarded(F &&body) noexcept
{
   int code = MI355RT_E_ARG;
   const char *what = "unexpected C++ exception";
   try {
      return body();
   } catch (const std::bad_alloc &) {
      code = MI355RT_E_MEMORY;
      what = "out of host memory";
   } catch (const std::length_error &) {
      code = MI355RT_E_MEMORY;
      what = "out of host memory (size beyond what a container holds)";
   } catch (const std::exception &) {
      what = "unexpected C++ exception (std::exception)";
   } catch (...) {
   }
   try {
      g_error = what;
   } catch (...) {
      g_error.clear();
   }
   return code;
}

#define HIP_TRY(expr)                                  \
   do {                                                \
      hipError_t e_ = (expr);                          \
      if (e_ != hipSuccess) return hip_fail(e_, #expr); \
   } while (0)

// makes `device` current for the scope of one API call
class DeviceScope {
 public:
   explicit DeviceScope(int device) : prev_(-1), ok_(true)
   {
      if (hipGetDevice(&prev_) != hipSuccess) { ok_ = false; return; }
      if (prev_ != device && hipSetDevice(device) != hipSuccess) ok_ = false;
      changed_ = ok_ && prev_ != device;
   }
   ~DeviceScope()
   {
      if (changed_) (void) hipSetDevice(prev_);
   }
   bool ok() const { return ok_; }

 private:
   int prev_;
   bool ok_;
   bool changed_ = false;
};

struct HostObject {
   float bmin[3], bmax[3];
   float shadow_distance;
   uint64_t n_nodes, n_leaves, n_slots;
   int32_t max_depth;
   bool is_mesh;
};

}   // namespace

// device scratch of one frame pass (layout.h "FRAME SCRATCH")
struct FrameScratch {
   size_t cap = 0;        // sample slots
   bool with_mirror = false, with_refraction = false;
   float *refr_parent = nullptr, *refr_terms = nullptr;
   float *samples = nullptr, *rays0 = nullptr, *hits0 = nullptr, *mat0 = nullptr;
   unsigned long long *queued = nullptr;
   int32_t *hit_obj0 = nullptr;
   uint32_t *shadow = nullptr;
   float *rays1[2] = { nullptr, nullptr };
   float *hits1 = nullptr;
   int32_t *hit_obj1 = nullptr;
   uint64_t bytes = 0;
};

namespace {
struct Multi;   // multi_gpu.inl
}

struct mi355rt_scene {
   int device = 0;
   int n_cus = 256;
   std::vector<HostObject> objects;
   std::vector<mi355rt_material> materials;
   DevObject *d_objs = nullptr;
   DevObjectExt *d_ext = nullptr;   // parallel to d_objs: sphere / plane / refraction / checkerboard parameters
   int n_analytic = 0;              // spheres + planes
   bool has_refraction = false;
   bool boxes_finite = true;        // see SceneArgs.boxes_finite
   bool big_leaves = false;         // see SceneArgs.big_leaves
   bool any_gpu_build = false;
   size_t n_node_recs = 0, n_descs = 0, n_gface_floats = 0, n_slots = 0, n_fnrm_floats = 0;   // sizes of the arrays below
   void *d_nodes = nullptr, *d_descs = nullptr, *d_gfaces = nullptr, *d_fnrm = nullptr;   // all objects, concatenated (layout.h)
   int32_t *d_slot_ids = nullptr;
   uint64_t n_groups = 0, n_shared_groups = 0;
   float *d_brdf_dirs = nullptr, *d_brdf_weights = nullptr;
   int32_t n_dirs = 0, n_materials = 0;
   float smin[3], smax[3];   // scene bounds (Scene.java:21-33)
   float abs_max = 0.f;
   float max_normal_len = 0.f;
   bool has_mirror = false;
   bool plain_materials = true;   // ColorInformation only
   uint64_t device_bytes = 0;
   uint32_t *d_counters = nullptr;   // CNT_WORDS
   unsigned long long *h_readback = nullptr;   // page-locked, N_STATS words: where run_frame reads queue sizes and counters back to
   void *noalloc_scratch = nullptr;  // this device's copy of the static scratch array (mi355rt_render_device_noalloc)
   bool stats_valid = false;
   mi355rt_stats stats_accum {};
   unsigned long long diag[N_STATS] = { 0 };
   FrameScratch scratch;
   // MI355RT_COLLECT_TIMES: HIP events around each launch, resolved lazily (mi355rt_kernel_profile)
   struct TimedLaunch {
      hipEvent_t begin, end;
      int kernel;
   };
   std::vector<TimedLaunch> pending_times;
   std::vector<hipEvent_t> event_pool;
   double kernel_ms[MI355RT_N_KERNELS] = { 0 };
   uint64_t kernel_launches[MI355RT_N_KERNELS] = { 0 };
   // frame buffers of the host-buffer render entry point
   float *d_rgb = nullptr;
   int32_t *d_hit_id = nullptr;
   int32_t *d_hit_obj = nullptr;
   float *d_hit_w = nullptr;
   uint32_t *d_argb8 = nullptr;
   size_t frame_pixels = 0;
   hipStream_t copy_stream = nullptr;   // mi355rt_render_host: device-to-host copies run beside each other
   int n_bouncing = 0;                  // objects with a ReflectionMaterial or RefractionMaterial
   // seconds of scene creation: [0] validate + bounds (+ host normals and hierarchy), GPU builder: [1] upload + normals,
   // [2] hierarchy, [3] leaf groups, [4] records; [5] host builder: records + leaf groups; [6] copies into the scene's arrays
   double build_seconds[7] = { 0, 0, 0, 0, 0, 0, 0 };
   Multi *multi = nullptr;             // several GPUs behind this handle (mi355rt_scene_set_devices)
   // mi355rt_scene_view: a view shares the scene arrays of `parent` (everything a frame only reads) and owns what a frame
   // writes (scratch, counters, read-back block, statistics, timers, host-render frame buffers)
   mi355rt_scene *parent = nullptr;
   int n_views = 0;      // views alive on this scene
   bool dying = false;   // destroyed while views were alive: released with the last of them
};

namespace {

void free_scratch(FrameScratch &fs)
{
   void *all[] = { fs.queued, fs.samples, fs.rays0, fs.hits0, fs.mat0, fs.hit_obj0, fs.shadow, fs.rays1[0], fs.rays1[1], fs.hits1, fs.hit_obj1,
                   fs.refr_parent, fs.refr_terms };
   for (void *q : all)
      if (q) (void) hipFree(q);
   fs = FrameScratch();
}

// what a handle owns per FRAME (scratch queues, counters, read-back block, host-render frame buffers, timers): a view's
// own, and a scene's too -- a scene destroyed while views are alive gives these back at once (up to 14-46 GB of scratch)
void free_frame_resources(mi355rt_scene *s)
{
   if (s->d_counters) (void) hipFree(s->d_counters);
   if (s->h_readback) (void) hipHostFree(s->h_readback);
   if (s->d_rgb) (void) hipFree(s->d_rgb);
   if (s->d_argb8) (void) hipFree(s->d_argb8);
   if (s->d_hit_id) (void) hipFree(s->d_hit_id);
   if (s->d_hit_obj) (void) hipFree(s->d_hit_obj);
   if (s->d_hit_w) (void) hipFree(s->d_hit_w);
   s->d_counters = nullptr;
   s->h_readback = nullptr;
   s->d_rgb = nullptr;
   s->d_argb8 = nullptr;
   s->d_hit_id = s->d_hit_obj = nullptr;
   s->d_hit_w = nullptr;
   s->frame_pixels = 0;
   free_scratch(s->scratch);
   for (const mi355rt_scene::TimedLaunch &t : s->pending_times) {
      (void) hipEventDestroy(t.begin);
      (void) hipEventDestroy(t.end);
   }
   s->pending_times.clear();
   for (hipEvent_t e : s->event_pool) (void) hipEventDestroy(e);
   s->event_pool.clear();
   if (s->copy_stream) (void) hipStreamDestroy(s->copy_stream);
   s->copy_stream = nullptr;
}

void release_multi(mi355rt_scene *s);   // the other GPUs' replicas, streams and blocks of a handle (below)

// the calling thread's current device must be s->device
void release_scene_impl(mi355rt_scene *s)
{
   if (!s) return;
   release_multi(s);
   if (!s->parent && s->n_views > 0) {   // its views still read the arrays: the last of them releases the scene
      free_frame_resources(s);
      s->dying = true;
      return;
   }
   if (!s->parent) {
      if (s->d_nodes) (void) hipFree(s->d_nodes);
      if (s->d_descs) (void) hipFree(s->d_descs);
      if (s->d_gfaces) (void) hipFree(s->d_gfaces);
      if (s->d_slot_ids) (void) hipFree(s->d_slot_ids);
      if (s->d_fnrm) (void) hipFree(s->d_fnrm);
      if (s->d_objs) (void) hipFree(s->d_objs);
      if (s->d_ext) (void) hipFree(s->d_ext);
      if (s->d_brdf_dirs) (void) hipFree(s->d_brdf_dirs);
      if (s->d_brdf_weights) (void) hipFree(s->d_brdf_weights);
   }
   free_frame_resources(s);
   mi355rt_scene *root = s->parent;
   delete s;
   if (root && --root->n_views == 0 && root->dying) release_scene_impl(root);
}

// waits for the pending timed launches and folds their durations into kernel_ms / kernel_launches
int resolve_times(mi355rt_scene *s)
{
   for (const mi355rt_scene::TimedLaunch &t : s->pending_times) {
      float ms = 0.0f;
      hipError_t e = hipEventSynchronize(t.end);
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, t.begin, t.end);
      s->event_pool.push_back(t.begin);
      s->event_pool.push_back(t.end);
      if (e != hipSuccess) {
         s->pending_times.clear();
         return hip_fail(e, "hipEventElapsedTime");
      }
      s->kernel_ms[t.kernel] += (double) ms;
      s->kernel_launches[t.kernel] += 1;
   }
   s->pending_times.clear();
   return MI355RT_OK;
}

// brackets one launch with events when the frame collects times
class LaunchTimer {
 public:
   LaunchTimer(mi355rt_scene *s, bool on, hipStream_t stream) : s_(s), on_(on), stream_(stream) {}
   template <typename F> hipError_t run(int kernel, F &&launch)
   {
      if (!on_) return launch();
      hipEvent_t ev[2];
      for (hipEvent_t &e : ev) {
         if (!s_->event_pool.empty()) {
            e = s_->event_pool.back();
            s_->event_pool.pop_back();
         } else if (hipError_t rc = hipEventCreate(&e); rc != hipSuccess) {
            return rc;
         }
      }
      hipError_t rc = hipEventRecord(ev[0], stream_);
      if (rc == hipSuccess) rc = launch();
      if (rc == hipSuccess) rc = hipEventRecord(ev[1], stream_);
      if (rc != hipSuccess) {
         s_->event_pool.push_back(ev[0]);
         s_->event_pool.push_back(ev[1]);
         return rc;
      }
      s_->pending_times.push_back({ ev[0], ev[1], kernel });
      return hipSuccess;
   }

 private:
   mi355rt_scene *s_;
   bool on_;
   hipStream_t stream_;
};

int upload(const void *src, size_t bytes, void **dst, mi355rt_scene *s)
{
   *dst = nullptr;
   const size_t alloc = bytes ? bytes : 64;
   HIP_TRY(hipMalloc(dst, alloc));
   s->device_bytes += alloc;
   if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
   return MI355RT_OK;
}

int check_settings(const mi355rt_settings *st)
{
   if (!st) return fail(MI355RT_E_ARG, "settings is null");
   if (st->multi_samples < 1) return fail(MI355RT_E_ARG, "multi_samples must be >= 1");
   if (st->brdf_samples < 0) return fail(MI355RT_E_ARG, "brdf_samples must be >= 0");
   if (st->lighting_model != MI355RT_LIGHTING_PLAIN && st->lighting_model != MI355RT_LIGHTING_PHONG)
      return fail(MI355RT_E_ARG, "unknown lighting_model");
   if (st->max_mirror_bounces < 0) return fail(MI355RT_E_ARG, "max_mirror_bounces must be >= 0");
   if (st->max_mirror_bounces > 2047) return fail(MI355RT_E_LIMIT, "max_mirror_bounces must be <= 2047");
   return MI355RT_OK;
}

void set_material(DevObject &d, const mi355rt_material &m)
{
   d.mat.kind = m.kind;
   std::memcpy(d.mat.diffuse, m.diffuse, sizeof(d.mat.diffuse));
   std::memcpy(d.mat.specular, m.specular, sizeof(d.mat.specular));
   d.mat.shininess = m.shininess;
   d.mat.brdf_id = m.brdf_id;
}

// Sphere ctor (sceneObjects/Sphere.java:15-22) with BoundingSphere.getMinMax (bounds/BoundingSphere.java:105-111), or the
// Plane ctor (sceneObjects/Plane.java:28-76).  Returns an error text, empty on success.
std::string add_analytic_object(mi355rt_scene *s, const mi355rt_object_desc &od, DevObject &d, DevObjectExt &ext)
{
   HostObject ho;
   std::memset(&ho, 0, sizeof(ho));
   ho.is_mesh = false;
   if (od.kind == MI355RT_OBJ_SPHERE) {
      for (int c = 0; c < 3; c++) {
         ext.center[c] = od.center[c];
         ho.bmin[c] = od.center[c] - od.radius;
         ho.bmax[c] = od.center[c] + od.radius;
      }
      ext.radius = od.radius;
      d.kind = OBJ_SPHERE;
   } else {
      float c[4][3];
      for (int i = 0; i < 4; i++)
         for (int k = 0; k < 3; k++) c[i][k] = od.corners[3 * ((uint32_t) i < od.n_corners ? i : 2) + k];   // :60-63 a triangle repeats its third corner
      ext.pA = c[0][1] * (c[1][2] - c[2][2]) + c[1][1] * (c[2][2] - c[0][2]) + c[2][1] * (c[0][2] - c[1][2]);
      ext.pB = c[0][2] * (c[1][0] - c[2][0]) + c[1][2] * (c[2][0] - c[0][0]) + c[2][2] * (c[0][0] - c[1][0]);
      ext.pC = c[0][0] * (c[1][1] - c[2][1]) + c[1][0] * (c[2][1] - c[0][1]) + c[2][0] * (c[0][1] - c[1][1]);
      ext.pD = -(c[0][0] * (c[1][1] * c[2][2] - c[2][1] * c[1][2]) + c[1][0] * (c[2][1] * c[0][2] - c[0][1] * c[2][2]) +
                 c[2][0] * (c[0][1] * c[1][2] - c[1][1] * c[0][2]));
      if ((ext.pA == 0) && (ext.pB == 0) && (ext.pC == 0)) return "the first three corners of the plane are colinear (Plane.java:56-58)";
      const float len2 = ext.pA * ext.pA + ext.pB * ext.pB + ext.pC * ext.pC;
      const float rn = (float) (1.0 / std::sqrt((double) len2));   // Vector3f.normalize
      ext.pnormal[0] = ext.pA * rn;
      ext.pnormal[1] = ext.pB * rn;
      ext.pnormal[2] = ext.pC * rn;
      for (int k = 0; k < 3; k++) {
         ho.bmin[k] = jm::fmin_java(c[0][k], jm::fmin_java(c[1][k], jm::fmin_java(c[2][k], c[3][k])));
         ho.bmax[k] = jm::fmax_java(c[0][k], jm::fmax_java(c[1][k], jm::fmax_java(c[2][k], c[3][k])));
      }
      d.kind = OBJ_PLANE;
   }
   std::memcpy(d.bmin, ho.bmin, sizeof(d.bmin));
   std::memcpy(d.bmax, ho.bmax, sizeof(d.bmax));
   d.root_entry = STACK_EMPTY;   // the walker passes over it
   set_material(d, od.mesh.material);
   s->objects.push_back(ho);
   s->materials.push_back(od.mesh.material);
   s->n_analytic++;
   if (s->max_normal_len < 1.0001f) s->max_normal_len = 1.0001f;   // analytic normals are unit length (a NaN from a mesh stays)
   return std::string();
}

// One mesh's share of the scene's arrays, on the host: the hierarchy comes from the host builder (scene_build.cpp) or
// from the GPU builder (gpu_build.hip, downloaded), its leaves are grouped (scene_build.hpp "leaf groups"), and every
// part is copied to its place once all objects' sizes are known.
struct ScenePart {
   std::vector<DevNodeRec> nodes;    // links relative to the part: node index / index of a leaf's first descriptor
   std::vector<int32_t> slot_face;   // face id per leaf slot, the reference's order
   GroupedLeaves groups;             // descs (key / gbase relative to the part), group-face ids
   std::vector<float> fnrm;          // 12 floats per face: its three vertex normals
   bool on_gpu = false;              // built by gpu_build.hip: everything above is empty, `gpu` holds the part's arrays on the device
   GpuBuiltObject gpu;
   size_t n_nodes = 0, n_slots = 0, n_descs = 0, n_gfaces = 0, face_base = 0, n_faces = 0;
   ~ScenePart() { gpu_build_free(gpu); }
   ScenePart() = default;
   ScenePart(const ScenePart &) = delete;
   ScenePart &operator=(const ScenePart &) = delete;
   size_t node_base = 0, slot_base = 0, desc_base = 0, gface_base = 0;
   int32_t root_link = 0, root_cnt = 0;
};

// the host builder's records in the uploaded form (both children interleaved per coordinate)
void host_part_records(const BuiltObject &built, ScenePart &part)
{
   part.nodes.reserve(built.nodes.size());
   for (const NodeRec &rec : built.nodes) {
      DevNodeRec d;
      const float *lo[2] = { rec.lmin, rec.rmin }, *hi[2] = { rec.lmax, rec.rmax };
      for (int side = 0; side < 2; side++) {
         d.lo_x[side] = lo[side][0];
         d.lo_y[side] = lo[side][1];
         d.lo_z[side] = lo[side][2];
         d.hi_x[side] = hi[side][0];
         d.hi_y[side] = hi[side][1];
         d.hi_z[side] = hi[side][2];
      }
      d.llink = rec.llink;
      d.rlink = rec.rlink;
      d.lcnt = rec.lcnt;
      d.rcnt = rec.rcnt;
      d.lgbase = d.rgbase = 0;
      d.lmask = d.rmask = 0u;
      part.nodes.push_back(d);
   }
   part.slot_face = built.slot_face;
   part.root_link = built.root_link;
   part.root_cnt = built.root_cnt;
}

// 48 bytes per face: the three vertex normals, fetched once per final hit
void face_normals_of(std::vector<float> &fnrm, const mi355rt_mesh_desc &md, const std::vector<float> &normals)
{
   fnrm.assign(12 * (size_t) md.n_faces, 0.0f);
   for (size_t f = 0; f < md.n_faces; f++) {
      float *dst = fnrm.data() + 12 * f;
      for (int kx = 0; kx < 3; kx++) {
         const float *nrm = normals.data() + 3 * (size_t) md.indices[3 * f + kx];
         dst[4 * kx + 0] = nrm[0];
         dst[4 * kx + 1] = nrm[1];
         dst[4 * kx + 2] = nrm[2];
      }
   }
}

// Groups the part's leaves, makes its links absolute and notes whether every box is finite.
void finish_part(ScenePart &part, bool &boxes_finite)
{
   group_leaves(part.nodes.data(), part.nodes.size(), sizeof(DevNodeRec), part.slot_face.data(), part.root_link, part.root_cnt, part.groups);
   for (DevNodeRec &d : part.nodes) {
      // a leaf child's descriptor words ride in its parent's record (layout.h); the group base made absolute like the descriptor's own
      d.lgbase = d.rgbase = 0;
      d.lmask = d.rmask = 0u;
      if (d.llink != LINK_NONE && d.lcnt > 0) {
         d.lgbase = part.groups.descs[(size_t) d.llink].gbase + (int32_t) part.gface_base;
         d.lmask = part.groups.descs[(size_t) d.llink].mask;
      }
      if (d.rlink != LINK_NONE && d.rcnt > 0) {
         d.rgbase = part.groups.descs[(size_t) d.rlink].gbase + (int32_t) part.gface_base;
         d.rmask = part.groups.descs[(size_t) d.rlink].mask;
      }
      if (d.llink != LINK_NONE) d.llink += (int32_t) (d.lcnt > 0 ? part.desc_base : part.node_base);
      if (d.rlink != LINK_NONE) d.rlink += (int32_t) (d.rcnt > 0 ? part.desc_base : part.node_base);
      d.lcnt = d.lcnt > 0 ? -1 : 0;   // the uploaded form keeps "is a leaf" as a mask (layout.h)
      d.rcnt = d.rcnt > 0 ? -1 : 0;
      for (int side = 0; side < 2; side++)
         boxes_finite = boxes_finite && std::isfinite(d.lo_x[side]) && std::isfinite(d.lo_y[side]) && std::isfinite(d.lo_z[side]) &&
                        std::isfinite(d.hi_x[side]) && std::isfinite(d.hi_y[side]) && std::isfinite(d.hi_z[side]);
   }
   for (LeafDesc &d : part.groups.descs) {
      d.key += (int32_t) part.slot_base;
      d.gbase += (int32_t) part.gface_base;   // the flag bits sit above DESC_BASE_MASK
   }
}

// Allocates the scene's node, descriptor, group-face and slot-id arrays and fills each part's share.  GFACE_PAD zero
// faces follow the last group: a lane that sits a trip out reads entry 0, a leaf's last odd face reads one entry on.
hipError_t upload_parts(mi355rt_scene *s, const std::vector<ScenePart> &parts, const mi355rt_object_desc *objects, size_t node_total, size_t slot_total,
                        size_t desc_total, size_t gface_total, size_t face_total)
{
   const size_t node_bytes = std::max<size_t>(node_total, 1) * sizeof(DevNodeRec);
   const size_t desc_bytes = (desc_total + DESC_PAD) * sizeof(LeafDesc);
   const size_t gface_floats = 12 * (gface_total + GFACE_PAD);
   const size_t slot_bytes = std::max<size_t>(slot_total, 1) * sizeof(int32_t);
   const size_t fnrm_bytes = std::max<size_t>(12 * face_total * sizeof(float), 64);
   hipError_t e = hipMalloc(&s->d_nodes, node_bytes);
   if (e == hipSuccess) e = hipMalloc(&s->d_descs, desc_bytes);
   if (e == hipSuccess) e = hipMalloc(&s->d_gfaces, gface_floats * sizeof(float));
   if (e == hipSuccess) e = hipMalloc((void **) &s->d_slot_ids, slot_bytes);
   if (e == hipSuccess) e = hipMalloc(&s->d_fnrm, fnrm_bytes);
   if (e != hipSuccess) return e;
   s->device_bytes += node_bytes + desc_bytes + gface_floats * sizeof(float) + slot_bytes + fnrm_bytes;
   s->n_node_recs = node_total;
   s->n_descs = desc_total;
   s->n_gface_floats = gface_floats;
   s->n_slots = slot_total;
   s->n_fnrm_floats = 12 * face_total;
   e = hipMemset((float *) s->d_gfaces + 12 * gface_total, 0, 12 * (size_t) GFACE_PAD * sizeof(float));
   if (e == hipSuccess) e = hipMemset((LeafDesc *) s->d_descs + desc_total, 0, DESC_PAD * sizeof(LeafDesc));
   for (size_t i = 0; i < parts.size() && e == hipSuccess; i++) {
      const ScenePart &part = parts[i];
      const hipMemcpyKind kind = part.on_gpu ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;   // a part built on the GPU never left it
      const void *nodes = part.on_gpu ? (const void *) part.gpu.nodes : part.nodes.data();
      const void *descs = part.on_gpu ? (const void *) part.gpu.descs : part.groups.descs.data();
      const void *slots = part.on_gpu ? (const void *) part.gpu.slot_ids : part.slot_face.data();
      const void *fnrm = part.on_gpu ? (const void *) part.gpu.fnrm : part.fnrm.data();
      if (part.n_nodes) e = hipMemcpy((DevNodeRec *) s->d_nodes + part.node_base, nodes, part.n_nodes * sizeof(DevNodeRec), kind);
      if (e == hipSuccess && part.n_descs) e = hipMemcpy((LeafDesc *) s->d_descs + part.desc_base, descs, part.n_descs * sizeof(LeafDesc), kind);
      if (e == hipSuccess && part.n_slots) e = hipMemcpy(s->d_slot_ids + part.slot_base, slots, part.n_slots * sizeof(int32_t), kind);
      if (e == hipSuccess && part.n_faces) e = hipMemcpy((float *) s->d_fnrm + 12 * part.face_base, fnrm, 12 * part.n_faces * sizeof(float), kind);
      if (e == hipSuccess && part.n_gfaces) {
         if (part.on_gpu)
            e = hipMemcpy((float *) s->d_gfaces + 12 * part.gface_base, part.gpu.gfaces, 12 * part.n_gfaces * sizeof(float), hipMemcpyDeviceToDevice);
         else   // the faces' corners, gathered on the device
            e = gpu_emit_gfaces(objects[i].mesh.vertices, objects[i].mesh.n_vertices, objects[i].mesh.indices, objects[i].mesh.n_faces,
                                part.groups.gface.data(), part.groups.gface.size(), (float *) s->d_gfaces + 12 * part.gface_base);
      }
   }
   if (e == hipSuccess) e = hipDeviceSynchronize();
   return e;
}

// a pass renders at most this many sample slots at once (bounds the scratch: ~104 B per slot,
// ~204 B with mirrors, 144 B more with a RefractionMaterial)
// Every pass ends on its longest rays with the chip draining, so passes are as large as memory allows: 2^27 slots (a 4K frame
// at 16 spp in one pass: 14 GB of scratch for a mesh scene, 27 GB with a mirror) unless that is more than a quarter of the
// HBM that is free when the scratch has to grow; measured 4K x 16 spp, 2^24 against 2^27: Thai-class 15.2 -> 11.9 ms,
// Lucy-class 51.5 -> 48.3.  MI355RT_PASS_SLOTS_LOG2 forces a size (the tests use it to run many small passes).
constexpr size_t PASS_SLOTS_DEFAULT_LOG2 = 27, PASS_SLOTS_MIN_LOG2 = 22, PASS_SLOTS_MAX_LOG2 = 28;
size_t scratch_bytes_per_slot(bool with_mirror, bool with_refraction)
{
   return 104 + 1 + (with_mirror ? 100 : 0) + (with_refraction ? 16 * (1 + MAX_REFRACTION_TERMS) : 0);
}
size_t pass_slots(const mi355rt_scene *s, size_t frame_slots, bool with_mirror, bool with_refraction)
{
   if (const long l = switches().pass_slots_log2; l > 0)
      return (size_t) 1 << (l < 12 ? 12 : (l > (long) PASS_SLOTS_MAX_LOG2 ? (long) PASS_SLOTS_MAX_LOG2 : l));
   size_t slots = (size_t) 1 << PASS_SLOTS_DEFAULT_LOG2;
   const FrameScratch &fs = s->scratch;
   const bool fits = (fs.with_mirror || !with_mirror) && (fs.with_refraction || !with_refraction);
   if (fits && fs.cap >= std::min(frame_slots, slots)) return slots;   // nothing to allocate
   size_t free_b = 0, total_b = 0;
   if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t) 1 << PASS_SLOTS_MIN_LOG2;
   free_b += fs.bytes;   // what the scratch holds now is given back first
   while (slots > ((size_t) 1 << PASS_SLOTS_MIN_LOG2) && slots / 2 >= frame_slots) slots >>= 1;   // (no larger than the frame needs)
   while (slots > ((size_t) 1 << PASS_SLOTS_MIN_LOG2) && slots * scratch_bytes_per_slot(with_mirror, with_refraction) > free_b / 4) slots >>= 1;
   return slots;
}
constexpr int BOUNCE_UNCONDITIONAL = 3;   // bounces enqueued without a read-back of the bounce queue's size (run_frame)
constexpr unsigned long long FUSE_MIN_SLOTS = 100000ull;   // passes of at least this many samples walk shadow rays inside the primary launch (with short passes dealt out spread the fused launch wins on every share of the bench frame measured: 8.3 / 2.1 / 1.0 / 0.26 M samples 3.69 / 1.86 / 1.18 / 0.58 ms against 4.06 / 2.06 / 1.37 / 0.72)

int ensure_scratch(mi355rt_scene *s, size_t cap, bool with_mirror, bool with_refraction)
{
   FrameScratch &fs = s->scratch;
   if (fs.cap >= cap && (fs.with_mirror || !with_mirror) && (fs.with_refraction || !with_refraction)) return MI355RT_OK;
   free_scratch(fs);
   struct Want {
      void **dst;
      size_t bytes;
   };
   const Want wants[] = {
      { (void **) &fs.queued, cap / 8 },
      { (void **) &fs.samples, cap * 16 },  { (void **) &fs.rays0, cap * 32 },    { (void **) &fs.hits0, cap * 32 },
      { (void **) &fs.mat0, cap * 16 },     { (void **) &fs.hit_obj0, cap * 4 },  { (void **) &fs.shadow, cap * 4 },
      { (void **) &fs.rays1[0], with_mirror ? cap * 32 : 0 }, { (void **) &fs.rays1[1], with_mirror ? cap * 32 : 0 },
      { (void **) &fs.hits1, with_mirror ? cap * 32 : 0 },    { (void **) &fs.hit_obj1, with_mirror ? cap * 4 : 0 },
      { (void **) &fs.refr_parent, with_refraction ? cap * 16 : 0 },
      { (void **) &fs.refr_terms, with_refraction ? cap * 16 * MAX_REFRACTION_TERMS : 0 },
   };
   for (const Want &w : wants) {
      if (!w.bytes) continue;
      hipError_t e = hipMalloc(w.dst, w.bytes);
      if (e != hipSuccess) {
         free_scratch(fs);
         return hip_fail(e, "hipMalloc(frame scratch)");
      }
      fs.bytes += w.bytes;
   }
   fs.cap = cap;
   fs.with_mirror = with_mirror;
   fs.with_refraction = with_refraction;
   return MI355RT_OK;
}

// RTStatics.shadowIntersects does a full closest-hit query per object and then asks whether that
// hit lies before the light (RTStatics.java:626-633).  When the light is farther from the scene's
// box than the box's own diagonal (with slack), EVERY accepted candidate (w > shadowDistance)
// satisfies "w > EPSILON && w < distance(hit, light) + EPSILON", so "occluded" is simply "some
// candidate exists" and the walk may stop at the first one.  Conditions checked here:
//   * finite scene box; every object's shadowDistance >= EPSILON (so w > shadowDistance => w > EPSILON);
//   * w <= diag(box) + |N|*EPSILON for any origin on a surface and any hit point in the box, and
//     distance(hit, light) >= distance(light, box); require distance(light, box) > 1.01 * that bound.
bool shadow_any_hit_is_exact(const mi355rt_scene *s, const mi355rt_settings *st)
{
   if (s->objects.empty()) return false;
   double diag2 = 0.0, dl2 = 0.0;
   for (int c = 0; c < 3; c++) {
      if (!std::isfinite(s->smin[c]) || !std::isfinite(s->smax[c]) || !std::isfinite(st->light_origin[c])) return false;
      const double e = (double) s->smax[c] - (double) s->smin[c];
      diag2 += e * e;
      const double l = st->light_origin[c];
      const double d = std::fmax(std::fmax((double) s->smin[c] - l, l - (double) s->smax[c]), 0.0);
      dl2 += d * d;
   }
   if (!(st->epsilon >= 0.0f) || !std::isfinite(s->max_normal_len)) return false;
   for (const HostObject &o : s->objects)   // mesh candidates only: spheres and planes are tested in full by analytic_shadow_kernel
      if (o.is_mesh && !(o.shadow_distance >= st->epsilon)) return false;
   const double reach = std::sqrt(diag2) + 2.0 * (double) s->max_normal_len * (double) st->epsilon + (double) st->epsilon;
   return std::sqrt(dl2) > 1.01 * reach + 1e-6 * (double) s->abs_max;
}

// layout.h FrameArgs::FastDiv.  d >= 1.  For d not a power of two 2^s < d < 2^(s+1), so m = ceil(2^(32+s) / d) lies in
// (2^31, 2^32); mulhi(n, m) >> s = floor(n m / 2^(32+s)) and n (m - 2^(32+s)/d) < n <= 2^31 < 2^(32+s)/d, which is what makes
// the floor exact.
FrameArgs::FastDiv fast_div_of(uint32_t d)
{
   FrameArgs::FastDiv f;
   f.d = d;
   f.pad = 0;
   uint32_t s = 0;
   while ((2u << s) <= d && s < 31) ++s;   // floor(log2 d)
   f.s = s;
   if ((d & (d - 1)) == 0) {
      f.m = 0;
   } else {
      const unsigned long long two = 1ull << (32 + s);
      f.m = (uint32_t) ((two + d - 1) / d);
   }
   return f;
}

int fill_frame_args(const mi355rt_scene *s, const mi355rt_camera *cam, const mi355rt_settings *st, FrameArgs &p, long long &n_micro)
{
   if (!cam) return fail(MI355RT_E_ARG, "camera is null");
   if (int rc = check_settings(st)) return rc;
   if (cam->width < 1 || cam->height < 1) return fail(MI355RT_E_ARG, "image size must be positive");
   if ((uint64_t) cam->width * (uint64_t) cam->height > (uint64_t) 1 << 30) return fail(MI355RT_E_LIMIT, "image larger than 2^30 pixels");
   if ((size_t) st->multi_samples * 64 > ((size_t) 1 << 24)) return fail(MI355RT_E_LIMIT, "multi_samples too large");
   std::memset(&p, 0, sizeof(p));
   p.sc.objs = s->d_objs;
   p.sc.nodes = s->d_nodes;
   p.sc.gfaces = s->d_gfaces;
   p.sc.descs = s->d_descs;
   p.sc.slot_ids = s->d_slot_ids;
   p.sc.fnrm = s->d_fnrm;
   p.sc.n_objs = (int32_t) s->objects.size();
   p.sc.epsilon = st->epsilon;
   p.sc.near_plane = st->near_plane;
   p.sc.far_plane = st->far_plane;
   p.sc.scene_abs_max = s->abs_max;
   p.sc.max_normal_len = s->max_normal_len;
   p.sc.ext = s->d_ext;
   p.sc.n_analytic = s->n_analytic;
   p.sc.boxes_finite = s->boxes_finite ? 1 : 0;
   p.sc.big_leaves = s->big_leaves ? 1 : 0;
   p.max_recursion_depth = st->max_recursion_depth;
   std::memcpy(p.origin, cam->origin, sizeof(p.origin));
   std::memcpy(p.rot, cam->rotation, sizeof(p.rot));
   // Camera.createImage constants (Camera.java:123-126)
   p.x_min = -(cam->viewport_w / 2.0f);
   p.y_min = cam->viewport_h / 2.0f;
   p.x_inc = cam->viewport_w / (float) cam->width;
   p.y_inc = -cam->viewport_h / (float) cam->height;
   p.near_plane_cam = cam->near_plane;
   p.width = cam->width;
   p.height = cam->height;
   p.multi_samples = st->multi_samples;
   p.brdf_samples = st->brdf_samples;
   p.lighting_model = st->lighting_model;
   p.max_mirror_bounces = st->max_mirror_bounces;
   p.rng_seed = st->rng_seed;
   std::memcpy(p.light_origin, st->light_origin, sizeof(p.light_origin));
   std::memcpy(p.light_emission, st->light_emission, sizeof(p.light_emission));
   std::memcpy(p.light_ambient, st->light_ambient, sizeof(p.light_ambient));
   p.shadow_any_hit = shadow_any_hit_is_exact(s, st) ? 1 : 0;
   // One launch for primary and shadow rays (walk_kernel MODE_FUSED) is possible when the shadow answer is "any accepted
   // face" and the primary hit is final once the mesh walk ends (no spheres or planes to compare it with afterwards).
   // Here: 0 never, 1 for passes long enough to gain from it (run_frame), 2 always.  The switch MI355RT_FUSE_SHADOW=0/1 forces it.
   const int fuse_forced = switches().fuse_shadow;
   const bool can_fuse = p.shadow_any_hit && st->lighting_model == MI355RT_LIGHTING_PHONG && s->n_analytic == 0;
   p.fused_shadow = !can_fuse || fuse_forced == 0 ? 0 : (fuse_forced == 1 ? 2 : 1);
   p.brdf_dirs = s->d_brdf_dirs;
   p.brdf_weights = s->d_brdf_weights;
   p.n_dirs = s->n_dirs;
   p.n_materials = s->n_materials;
   for (size_t i = 0; i < s->materials.size(); i++) {
      const mi355rt_material &m = s->materials[i];
      if (m.kind == MI355RT_MAT_BRDF && (m.brdf_id < 0 || m.brdf_id >= s->n_materials))
         return fail(MI355RT_E_ARG, "object " + std::to_string(i) + ": BRDF material row " + std::to_string(m.brdf_id) +
                                        " is outside the loaded tables (call mi355rt_brdf_tables first)");
   }
   int tile = st->tile_size == 0 ? 16 : st->tile_size;
   int world = st->tile_world == 0 ? 1 : st->tile_world;
   if (tile < 8 || tile % 8 != 0 || tile > 4096) return fail(MI355RT_E_ARG, "tile_size must be a multiple of 8 in [8, 4096]");
   if (world < 1 || st->tile_rank < 0 || st->tile_rank >= world) return fail(MI355RT_E_ARG, "need 0 <= tile_rank < tile_world");
   p.tile_size = tile;
   p.tile_rank = st->tile_rank;
   p.tile_world = world;
   p.tiles_x = (cam->width + tile - 1) / tile;
   p.tiles_y = (cam->height + tile - 1) / tile;
   const long long n_tiles = (long long) p.tiles_x * p.tiles_y;
   const long long owned = n_tiles > st->tile_rank ? (n_tiles - st->tile_rank + world - 1) / world : 0;
   p.micro_per_tile = (tile / 8) * (tile / 8);
   p.div_ms = fast_div_of((uint32_t) st->multi_samples);
   p.div_micro_per_tile = fast_div_of((uint32_t) p.micro_per_tile);
   p.div_tiles_x = fast_div_of((uint32_t) p.tiles_x);
   p.div_mpr = fast_div_of((uint32_t) (tile / 8));
   n_micro = owned * p.micro_per_tile;
   if (n_micro > 0x3fffffffLL) return fail(MI355RT_E_LIMIT, "too many 8x8 pixel groups");
   p.tile_compact = st->tile_compact != 0 ? 1 : 0;
   p.stats = (st->collect_stats & (MI355RT_COLLECT_COUNTS | MI355RT_COLLECT_DIAG)) != 0;
   p.diag = (st->collect_stats & MI355RT_COLLECT_DIAG) != 0;
   return MI355RT_OK;
}

// one frame = passes of at most PASS_SLOTS sample slots, each a run of the kernel pipeline
// `fixed`: the frame runs as ONE pass in these buffers (with `fixed_counters` as its counter block) and nothing is
// allocated, created or freed on the way -- mi355rt_render_device_noalloc; else the scene's own scratch, grown on demand.
int run_frame(mi355rt_scene *s, FrameArgs &p, long long n_micro, bool collect_times, hipStream_t stream, const FrameScratch *fixed = nullptr,
              uint32_t *fixed_counters = nullptr)
{
   if (n_micro == 0) return MI355RT_OK;
   if (s->pending_times.size() > 4096)
      if (int rc = resolve_times(s)) return rc;
   LaunchTimer timer(s, collect_times, stream);
   const size_t slots_per_micro = (size_t) 64 * (size_t) p.multi_samples;
   const bool with_mirror = s->has_mirror || s->has_refraction;
   const long long micro_per_pass =
      fixed ? n_micro : (long long) std::max<size_t>(1, pass_slots(s, (size_t) n_micro * slots_per_micro, with_mirror, s->has_refraction) / slots_per_micro);
   const long long first_pass = std::min(n_micro, micro_per_pass);
   const size_t first_slots = (size_t) ((first_pass + N_BANDS - 1) / N_BANDS) * N_BANDS * slots_per_micro;
   if (fixed) {
      if (fixed->cap < first_slots) return fail(MI355RT_E_LIMIT, "frame does not fit the fixed scratch");
   } else if (int rc = ensure_scratch(s, first_slots, with_mirror, s->has_refraction)) {
      return rc;
   }
   const FrameScratch &fs = fixed ? *fixed : s->scratch;
   p.samples = fs.samples;
   p.queued = fs.queued;
   p.rays0 = fs.rays0;
   p.hits0 = fs.hits0;
   p.hit_obj0 = fs.hit_obj0;
   p.mat0 = fs.mat0;
   p.shadow = fs.shadow;
   p.rays1[0] = fs.rays1[0];
   p.rays1[1] = fs.rays1[1];
   p.hits1 = fs.hits1;
   p.hit_obj1 = fs.hit_obj1;
   p.refr_parent = fs.refr_parent;
   p.refr_terms = fs.refr_terms;
   p.counters = fixed ? fixed_counters : s->d_counters;
   if (p.stats) {
      s->stats_accum = mi355rt_stats {};
      s->stats_valid = true;
   }

   const int fuse_policy = p.fused_shadow;
   const bool bounce_sync = switches().bounce_sync != 0;
   for (long long m0 = 0; m0 < n_micro; m0 += micro_per_pass) {
      const long long m1 = std::min(n_micro, m0 + micro_per_pass);
      p.micro_begin = (int32_t) m0;
      p.micro_end = (int32_t) m1;
      p.band_cap = (uint32_t) ((((unsigned long long) (m1 - m0) + N_BANDS - 1) / N_BANDS) * slots_per_micro);
      const unsigned long long slots = (unsigned long long) p.band_cap * N_BANDS;   // queue capacity of the pass
      // (round 1's walker lost with the fused launch below 3 M samples: a short launch is mostly its end, and shadow rays
      // dealt out afresh over all waves ended sooner than each wave's own; dealt out spread, short fused launches win too)
      p.fused_shadow = fuse_policy == 2 || (fuse_policy == 1 && slots >= FUSE_MIN_SLOTS) ? 1 : 0;
      p.phong_in_shade = p.fused_shadow && !s->has_mirror && !s->has_refraction ? 1 : 0;
      p.plain_materials = s->plain_materials ? 1 : 0;
      HIP_TRY(hipMemsetAsync(p.counters, 0, CNT_WORDS * sizeof(uint32_t), stream));
      HIP_TRY(timer.run(MI355RT_K_GENERATE, [&] { return launch_generate(p, stream); }));
      int cursor_set = 0;   // each persistent launch owns 8 cursor words (zeroed with the rest at pass start)
      auto next_cursor = [&]() -> int { return CNT_CURSOR0 + 8 * (cursor_set++); };
      const int cw0 = next_cursor();
      HIP_TRY(timer.run(MI355RT_K_WALK_PRIMARY, [&] {
         return launch_trace(p.sc, p.rays0, p.hits0, p.hit_obj0, p.counters, CNT_Q0, cw0, p.band_cap, slots, s->n_cus, p.stats ? 2 : 0, p.diag, stream,
                             p.fused_shadow ? p.light_origin : nullptr, p.fused_shadow ? p.mat0 : nullptr);
      }));
      if (s->n_analytic) HIP_TRY(timer.run(MI355RT_K_ANALYTIC, [&] {
         return launch_analytic_closest(p.sc, p.rays0, p.hits0, p.hit_obj0, p.counters, CNT_Q0, p.band_cap, slots, s->n_cus, p.stats, stream);
      }));
      HIP_TRY(timer.run(MI355RT_K_SHADE, [&] { return launch_shade(p, 0, p.rays0, p.hits0, p.hit_obj0, CNT_Q0, p.rays1[0], CNT_SEC_A, slots, s->n_cus, stream); }));
      if (s->has_mirror || s->has_refraction) {
         // Mirror / refraction bounces: ping-pong queues until one comes up empty.  A bounce ray skips the object it left
         // (ReflectionMaterial.java:16, RefractionMaterial.java:33: mirrorObject), so in a scene with ONE such object a chain
         // is one bounce long and its single walk is simply enqueued.  With several, the first BOUNCE_UNCONDITIONAL bounces
         // are enqueued without asking (every launch finds its queue's size on the device: the walker trims itself to it,
         // shade and the analytic kernels stride over it, so a bounce that does not exist costs a few empty launches), and
         // only chains longer than that cost one read-back of the queue's size per further bounce -- the host has to know
         // when to stop.  (The switch MI355RT_BOUNCE_SYNC=1 asks after every bounce, as round 2 did: the tests compare.)
         const int max_walks = s->n_bouncing <= 1 ? std::min(1, p.max_mirror_bounces) : p.max_mirror_bounces;
         const int unconditional = bounce_sync ? 0 : (s->n_bouncing <= 1 ? 1 : BOUNCE_UNCONDITIONAL);
         int cur = 0;
         for (int bounce = 0; bounce < max_walks; bounce++) {
            unsigned long long n_sec = slots;   // sizes the launches when the queue's length stays on the device
            if (bounce >= unconditional) {
               uint32_t *n_band = (uint32_t *) s->h_readback;   // page-locked: a plain DMA, no staging buffer
               HIP_TRY(hipMemcpyAsync(n_band, p.counters + (cur == 0 ? CNT_SEC_A : CNT_SEC_B), N_BANDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
               HIP_TRY(hipStreamSynchronize(stream));
               n_sec = 0;
               for (int b = 0; b < N_BANDS; b++) n_sec += n_band[b];
               if (n_sec == 0) break;
            }
            const int cur_cnt = cur == 0 ? CNT_SEC_A : CNT_SEC_B, nxt_cnt = cur == 0 ? CNT_SEC_B : CNT_SEC_A;
            if (cursor_set >= CNT_CURSOR_SETS - 1) cursor_set = 1;   // keep the last set for the shadow launch
            const int cw = next_cursor();
            HIP_TRY(hipMemsetAsync(p.counters + cw, 0, 8 * sizeof(uint32_t), stream));
            HIP_TRY(hipMemsetAsync(p.counters + nxt_cnt, 0, N_BANDS * sizeof(uint32_t), stream));
            HIP_TRY(timer.run(MI355RT_K_WALK_SECONDARY, [&] {
               return launch_trace(p.sc, p.rays1[cur], p.hits1, p.hit_obj1, p.counters, cur_cnt, cw, p.band_cap, n_sec, s->n_cus, p.stats ? 1 : 0, p.diag, stream);
            }));
            if (s->n_analytic) HIP_TRY(timer.run(MI355RT_K_ANALYTIC, [&] {
               return launch_analytic_closest(p.sc, p.rays1[cur], p.hits1, p.hit_obj1, p.counters, cur_cnt, p.band_cap, n_sec, s->n_cus, 0, stream);
            }));
            HIP_TRY(timer.run(MI355RT_K_SHADE, [&] { return launch_shade(p, 1, p.rays1[cur], p.hits1, p.hit_obj1, cur_cnt, p.rays1[cur ^ 1], nxt_cnt, n_sec, s->n_cus, stream); }));
            cur ^= 1;
         }
      }
      if (p.lighting_model == MI355RT_LIGHTING_PHONG) {
         const int cw = CNT_CURSOR0 + 8 * (CNT_CURSOR_SETS - 1);
         if (!p.fused_shadow) HIP_TRY(timer.run(MI355RT_K_WALK_SHADOW, [&] { return launch_shadow(p, cw, slots, s->n_cus, stream); }));
         if (s->n_analytic) HIP_TRY(timer.run(MI355RT_K_ANALYTIC, [&] { return launch_analytic_shadow(p, slots, s->n_cus, stream); }));
         if (!p.phong_in_shade) HIP_TRY(timer.run(MI355RT_K_PHONG, [&] { return launch_phong(p, slots, s->n_cus, stream); }));
      }
      HIP_TRY(timer.run(MI355RT_K_RESOLVE, [&] { return launch_resolve(p, stream); }));
      if (p.stats) {
         unsigned long long *h = s->h_readback;
         HIP_TRY(hipMemcpyAsync(h, p.counters + CNT_STATS, N_STATS * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
         HIP_TRY(hipStreamSynchronize(stream));
         for (int q = 0; q < N_STATS; q++) s->diag[q] = (m0 == 0 ? 0ull : s->diag[q]) + h[q];
         s->stats_accum.primary_rays += h[0];
         s->stats_accum.shadow_rays += h[1];
         s->stats_accum.secondary_rays += h[2];
         s->stats_accum.box_tests += h[3];
         s->stats_accum.tri_tests += h[4];
         s->stats_accum.hits += h[5];
      }
   }
   return MI355RT_OK;
}

#include "multi_gpu.inl"

// One device's view of `root` (a scene, never a view): shares the arrays, owns what a frame writes.  root->device is current.
int make_view(mi355rt_scene *root, mi355rt_scene **view_out)
{
   *view_out = nullptr;
   mi355rt_scene *v = new mi355rt_scene(*root);   // the array pointers, sizes, objects, bounds and flags
   v->parent = root;
   v->n_views = 0;
   v->dying = false;
   v->multi = nullptr;
   v->d_counters = nullptr;
   v->h_readback = nullptr;
   v->scratch = FrameScratch();
   v->pending_times.clear();
   v->event_pool.clear();
   for (int k = 0; k < MI355RT_N_KERNELS; k++) {
      v->kernel_ms[k] = 0.0;
      v->kernel_launches[k] = 0;
   }
   v->stats_valid = false;
   v->stats_accum = mi355rt_stats {};
   v->d_rgb = nullptr;
   v->d_argb8 = nullptr;
   v->d_hit_id = v->d_hit_obj = nullptr;
   v->d_hit_w = nullptr;
   v->frame_pixels = 0;
   v->copy_stream = nullptr;
   v->device_bytes = 0;   // what the view itself holds
   root->n_views++;       // from here on release_scene_impl(v) undoes everything
   hipError_t e = hipMalloc((void **) &v->d_counters, CNT_WORDS * sizeof(uint32_t));
   if (e == hipSuccess) e = hipHostMalloc((void **) &v->h_readback, N_STATS * sizeof(unsigned long long), hipHostMallocDefault);
   if (e == hipSuccess) e = hipMemset(v->d_counters, 0, CNT_WORDS * sizeof(uint32_t));
   if (e != hipSuccess) {
      release_scene_impl(v);
      return hip_fail(e, "scene view");
   }
   v->device_bytes += CNT_WORDS * sizeof(uint32_t);
   *view_out = v;
   return MI355RT_OK;
}

// mi355rt_render_device's frame on whatever the handle spans: one GPU, or the GPUs of its device list
int render_any(mi355rt_scene *scene, const mi355rt_camera *cam, const mi355rt_settings *settings, const mi355rt_frame *frame, hipStream_t stream)
{
   if (scene->multi) return render_multi(scene, cam, settings, frame, stream);
   FrameArgs p;
   long long n_micro = 0;
   if (int rc = fill_frame_args(scene, cam, settings, p, n_micro)) return rc;
   p.rgb = frame->rgb;
   p.argb8 = frame->argb8;
   p.hit_id = frame->hit_id;
   p.hit_obj = frame->hit_obj;
   p.hit_w = frame->hit_w;
   return run_frame(scene, p, n_micro, (settings->collect_stats & MI355RT_COLLECT_TIMES) != 0, stream);
}

}   // namespace

extern "C" {

const char *mi355rt_last_error(void) { return g_error.c_str(); }
int mi355rt_abi_version(void) { return MI355RT_ABI_VERSION; }

int mi355rt_device_count(void)
{
   return guarded([&]() -> int {
      int n = 0;
      if (hipGetDeviceCount(&n) != hipSuccess) return 0;
      return n;
   });
}

int mi355rt_scene_create_objects(const mi355rt_object_desc *objects, uint32_t n_meshes, const mi355rt_build_opts *opts, mi355rt_scene **out)
{
   return guarded([&]() -> int {
      if (!out) return fail(MI355RT_E_ARG, "out is null");
      *out = nullptr;
      if (!opts) return fail(MI355RT_E_ARG, "build options are null");
      if (n_meshes && !objects) return fail(MI355RT_E_ARG, "objects is null");
      if (n_meshes > (uint32_t) MAX_OBJECTS) return fail(MI355RT_E_LIMIT, "more than " + std::to_string(MAX_OBJECTS) + " objects");
      for (uint32_t i = 0; i < n_meshes; i++) {
         const mi355rt_object_desc &od = objects[i];
         if (od.kind > MI355RT_OBJ_PLANE) return fail(MI355RT_E_ARG, "unknown kind of object " + std::to_string(i));
         if (od.mesh.material.kind > MI355RT_MAT_CHECKERBOARD) return fail(MI355RT_E_ARG, "unknown material kind on object " + std::to_string(i));
         if (od.kind == MI355RT_OBJ_PLANE && (od.n_corners < 3 || od.n_corners > 4))
            return fail(MI355RT_E_ARG, "object " + std::to_string(i) + ": a plane takes 3 or 4 corners");
      }
      int n_dev = 0;
      if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return fail(MI355RT_E_DEVICE, "no HIP device available (libmi355rt has no CPU path)");
      int device = opts->device;
      if (device < 0) {
         if (hipGetDevice(&device) != hipSuccess) return fail(MI355RT_E_DEVICE, "hipGetDevice failed");
      }
      if (device >= n_dev) return fail(MI355RT_E_ARG, "device ordinal out of range");
      DeviceScope scope(device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device");

      BuildOptions bo;
      bo.max_depth = opts->max_depth;
      bo.max_children = opts->max_children;
      bo.use_kdtree = opts->use_kdtree != 0;

      mi355rt_scene *s = new (std::nothrow) mi355rt_scene();
      if (!s) return fail(MI355RT_E_MEMORY, "out of host memory");
      struct SceneGuard {   // an exception on the way (std::bad_alloc from a vector) must not leak the scene or its device memory
         mi355rt_scene *s;
         ~SceneGuard() { release_scene_impl(s); }
      } guard { s };
      auto release_scene = [&guard](mi355rt_scene *) {   // the early returns below
         release_scene_impl(guard.s);
         guard.s = nullptr;
      };
      s->device = device;
      for (int c = 0; c < 3; c++) {
         s->smin[c] = jm::kFloatMax;
         s->smax[c] = -jm::kFloatMax;
      }
      std::vector<DevObject> dev_objs(n_meshes ? n_meshes : 1);
      std::memset(dev_objs.data(), 0, sizeof(DevObject) * dev_objs.size());
      std::vector<DevObjectExt> dev_ext(dev_objs.size());
      std::memset(dev_ext.data(), 0, sizeof(DevObjectExt) * dev_ext.size());
      // all objects share one array of each kind (links absolute); each object's part is built on the host (scene_build.cpp)
      // or on the GPU (gpu_build.hip), grouped on the host and copied to its place at the end
      std::vector<ScenePart> parts(n_meshes ? n_meshes : 1);
      bool boxes_finite = true;
      size_t slot_total = 0, face_total = 0, node_total = 0, desc_total = 0, gface_total = 0;
      const Switches sw = switches();
      const bool allow_gpu_build = !sw.builder_host;   // MI355RT_BUILDER=host: never build on the GPU (tuning / tests); default: GPU where it applies

      try {
         for (uint32_t i = 0; i < n_meshes; i++) {
            const mi355rt_object_desc &od = objects[i];
            const mi355rt_mesh_desc &md = od.mesh;
            DevObjectExt &ext = dev_ext[i];
            ext.refraction_index = od.ext.refraction_index;
            std::memcpy(ext.color1, od.ext.checker_color1, sizeof(ext.color1));
            std::memcpy(ext.color2, od.ext.checker_color2, sizeof(ext.color2));
            std::memcpy(ext.cell, od.ext.checker_cell, sizeof(ext.cell));
            if (md.material.kind == MI355RT_MAT_REFLECTION) s->has_mirror = true;
            if (md.material.kind == MI355RT_MAT_REFRACTION) s->has_refraction = true;
            if (md.material.kind == MI355RT_MAT_REFLECTION || md.material.kind == MI355RT_MAT_REFRACTION) s->n_bouncing++;
            if (md.material.kind != MI355RT_MAT_COLOR) s->plain_materials = false;
            if (od.kind != MI355RT_OBJ_MESH) {
               const std::string err = add_analytic_object(s, od, dev_objs[i], ext);
               if (!err.empty()) {
                                    release_scene(s);
                  return fail(MI355RT_E_ARG, "object " + std::to_string(i) + ": " + err);
               }
               continue;
            }
            ScenePart &part = parts[i];
            // The GPU builder takes every mesh whose root splits and whose vertices are all finite (Java's Math.min / max
            // treat NaN specially; the host builder restates that, the GPU's integer-key min / max do not).
            BuildOptions bo_obj = bo;   // computeKDTree is per TriangleMesh in the reference (TriangleMesh.java:23,89)
            if (md.compute_kdtree != 0) bo_obj.use_kdtree = md.compute_kdtree > 0;
            bool gpu_build = allow_gpu_build && bo_obj.use_kdtree && bo.max_depth >= 1 && md.n_faces > (uint32_t) std::max(bo.max_children, 0) &&
                             md.vertices && md.indices;
            if (gpu_build) {
               std::vector<char> finite(host_threads_for(3 * (size_t) md.n_vertices, (size_t) 1 << 20), 1);
               in_ranges(3 * (size_t) md.n_vertices, (size_t) 1 << 20, [&](size_t t, size_t first, size_t last) {
                  bool ok = true;
                  for (size_t v = first; ok && v < last; v++) ok = std::isfinite(md.vertices[v]);
                  finite[t] = ok ? 1 : 0;
               });
               for (char ok : finite) gpu_build = gpu_build && ok != 0;
            }
            bo_obj.build_tree = !gpu_build;
            bo_obj.synth_normals = !gpu_build;   // (a part built on the GPU gets its normals there too)
            BuiltObject bo_out;
            const auto t_prep = std::chrono::steady_clock::now();
            const std::string err = build_object(md.vertices, md.n_vertices, md.normals, md.indices, md.n_faces, bo_obj, bo_out);
            if (!err.empty()) {
               release_scene(s);
               return fail(err.find("max_depth") != std::string::npos ? MI355RT_E_LIMIT : MI355RT_E_ARG, "mesh " + std::to_string(i) + ": " + err);
            }
            const size_t face_base = face_total;
            uint64_t n_leaves = bo_out.n_leaves;
            int32_t tree_depth = bo_out.max_depth;
            const auto t_tree = std::chrono::steady_clock::now();
            part.node_base = node_total;
            part.slot_base = slot_total;
            part.desc_base = desc_total;
            part.gface_base = gface_total;
            part.face_base = face_base;
            part.n_faces = md.n_faces;
            float part_max_normal = 0.0f;
            if (gpu_build) {
               // normals, hierarchy, leaf groups and records all on the device, in their final form (absolute links): the
               // part is copied into the scene's arrays device to device
               std::string gerr;
               PartBases bases;
               bases.node = node_total;
               bases.slot = slot_total;
               bases.desc = desc_total;
               bases.gface = gface_total;
               const hipError_t ge = gpu_build_object(md.vertices, md.n_vertices, md.normals, md.indices, md.n_faces, bo_out.bmin, bo_out.bmax, bo_obj,
                                                      bases, part.gpu, gerr);
               if (ge != hipSuccess) {
                  release_scene(s);
                  if (!gerr.empty()) return fail(MI355RT_E_LIMIT, "mesh " + std::to_string(i) + ": " + gerr);
                  return hip_fail(ge, "GPU hierarchy build");
               }
               n_leaves = part.gpu.n_leaves;
               tree_depth = part.gpu.max_depth;
               part.on_gpu = true;
               s->any_gpu_build = true;
               part.root_link = 0;   // the GPU builder only takes meshes whose root splits: node 0
               part.root_cnt = 0;
               part.n_nodes = (size_t) part.gpu.n_nodes;
               part.n_slots = (size_t) part.gpu.n_slots;
               part.n_descs = (size_t) part.gpu.n_descs;
               part.n_gfaces = (size_t) part.gpu.n_gfaces;
               s->n_groups += part.gpu.n_groups;
               s->n_shared_groups += part.gpu.n_shared;
               part_max_normal = part.gpu.max_normal_len;   // (every box of such a part is finite: its vertices are)
            } else {
               host_part_records(bo_out, part);
            }
            const bool root_is_leaf = !gpu_build && bo_out.root_is_leaf;
            const int32_t root_faces = part.root_cnt;
            if (!gpu_build) {
               finish_part(part, boxes_finite);
               part.n_nodes = part.nodes.size();
               part.n_slots = part.slot_face.size();
               part.n_descs = part.groups.descs.size();
               part.n_gfaces = part.groups.gface.size();
               s->n_groups += part.groups.n_groups;
               s->n_shared_groups += part.groups.n_shared;
               face_normals_of(part.fnrm, md, bo_out.normals);
               for (size_t v = 0; v + 2 < bo_out.normals.size(); v += 3) {
                  const float *nv = bo_out.normals.data() + v;
                  const float len = std::sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
                  if (!(len <= part_max_normal)) part_max_normal = len;   // a NaN length sticks (and disables the any-hit shortcut)
               }
            }
            if (node_total + part.n_nodes > (size_t) 0x3ffffff0 || slot_total + part.n_slots > (size_t) 0x3ffffff0 ||
                face_base + md.n_faces > (size_t) 0x3ffffff0 || gface_total + part.n_gfaces > (size_t) 0x3ffffff0) {
               release_scene(s);
               return fail(MI355RT_E_LIMIT, "scene exceeds 2^30 nodes, leaf slots or faces");
            }
            const size_t node_base = node_total, slot_base = slot_total;
            node_total += part.n_nodes;
            slot_total += part.n_slots;
            desc_total += part.n_descs;
            gface_total += part.n_gfaces;
            const auto t_end = std::chrono::steady_clock::now();
            s->build_seconds[0] += std::chrono::duration<double>(t_tree - t_prep).count();
            if (gpu_build) {
               for (int q = 0; q < 4; q++) s->build_seconds[1 + q] += part.gpu.seconds[q];
            } else {
               s->build_seconds[5] += std::chrono::duration<double>(t_end - t_tree).count();
            }
            if (sw.build_timing) {   // profiling aid
               std::fprintf(stderr, "mi355rt build: object %u, %u faces: validate + bounds%s %.3f s, %s %.3f s\n", i, md.n_faces,
                            gpu_build ? "" : " + normals + host hierarchy", std::chrono::duration<double>(t_tree - t_prep).count(),
                            gpu_build ? "GPU normals + hierarchy + leaf groups + records" : "records + leaf groups", std::chrono::duration<double>(t_end - t_tree).count());
            }
            face_total += md.n_faces;

            s->objects.emplace_back();
            s->materials.push_back(md.material);
            if (!(part_max_normal <= s->max_normal_len)) s->max_normal_len = part_max_normal;   // (a NaN sticks)
            HostObject &ho = s->objects.back();
            std::memcpy(ho.bmin, bo_out.bmin, sizeof(ho.bmin));
            std::memcpy(ho.bmax, bo_out.bmax, sizeof(ho.bmax));
            ho.shadow_distance = bo_out.shadow_distance;
            ho.n_nodes = part.n_nodes;
            ho.n_leaves = n_leaves;
            ho.n_slots = part.n_slots;
            ho.max_depth = tree_depth;
            ho.is_mesh = true;
            DevObject &d = dev_objs[i];
            d.kind = OBJ_MESH;
            std::memcpy(d.bmin, ho.bmin, sizeof(d.bmin));
            std::memcpy(d.bmax, ho.bmax, sizeof(d.bmax));
            // a leaf root: its first descriptor (group_leaves left it in root_link, relative to the part)
            d.root_entry = !root_is_leaf ? (int32_t) node_base : (root_faces > 0 ? ~(int32_t) (part.desc_base + (size_t) part.root_link) : STACK_EMPTY);
            d.shadow_distance = bo_out.shadow_distance;
            d.face_base = (int32_t) face_base;
            d.node_base = (int32_t) node_base;
            d.slot_base = (int32_t) slot_base;
            set_material(d, md.material);
         }
         for (const HostObject &ho : s->objects) {   // Scene.java:21-33
            for (int c = 0; c < 3; c++) {
               s->smin[c] = jm::fmin_java(s->smin[c], ho.bmin[c]);
               s->smax[c] = jm::fmax_java(s->smax[c], ho.bmax[c]);
            }
         }
      } catch (const std::bad_alloc &) {
                  release_scene(s);
         return fail(MI355RT_E_MEMORY, "out of host memory while building the hierarchy");
      }
      {
         const auto t_up = std::chrono::steady_clock::now();
         const hipError_t e = upload_parts(s, parts, objects, node_total, slot_total, desc_total, gface_total, face_total);
         if (e != hipSuccess) {
            release_scene(s);
            return hip_fail(e, "hierarchy upload");
         }
         for (ScenePart &part : parts) gpu_build_free(part.gpu);   // (copied: the parts' own arrays go)
         s->build_seconds[6] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_up).count();
      }
      float amax = 0.f;
      for (int c = 0; c < 3; c++) {
         if (std::isfinite(s->smin[c])) amax = std::fmax(amax, std::fabs(s->smin[c]));
         if (std::isfinite(s->smax[c])) amax = std::fmax(amax, std::fabs(s->smax[c]));
      }
      s->abs_max = n_meshes ? amax : 0.f;
      s->boxes_finite = boxes_finite;
      {   // leaves of a dozen faces or more on average (a mesh of millions of faces under the reference's depth limit)
         uint64_t leaves = 0, slots = 0;
         for (const HostObject &o : s->objects) {
            leaves += o.n_leaves;
            slots += o.n_slots;
         }
         s->big_leaves = leaves > 0 && slots >= 12 * leaves;
      }
      int rc;
      if ((rc = upload(dev_objs.data(), sizeof(DevObject) * dev_objs.size(), (void **) &s->d_objs, s)) ||
          (rc = upload(dev_ext.data(), sizeof(DevObjectExt) * dev_ext.size(), (void **) &s->d_ext, s))) {
         release_scene(s);
         return rc;
      }
      hipError_t e = hipMalloc((void **) &s->d_counters, CNT_WORDS * sizeof(uint32_t));
      if (e == hipSuccess) e = hipHostMalloc((void **) &s->h_readback, N_STATS * sizeof(unsigned long long), hipHostMallocDefault);
      if (e != hipSuccess) {
         release_scene(s);
         return hip_fail(e, "hipMalloc(counters)");
      }
      // set-up for mi355rt_render_device_noalloc: this device's address of the static scratch array, while the device is current
      // (a device that cannot resolve it -- too short of memory for the array -- only loses that entry point, which reports it)
      if (noalloc_scratch_ptr(device, &s->noalloc_scratch) != hipSuccess) {
         s->noalloc_scratch = nullptr;
         (void) hipGetLastError();   // clear the sticky error
      }
      int cus = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) s->n_cus = cus;
      guard.s = nullptr;
      *out = s;
      return MI355RT_OK;
   });
}

int mi355rt_scene_create(const mi355rt_mesh_desc *meshes, uint32_t n_meshes, const mi355rt_build_opts *opts, mi355rt_scene **out)
{
   return guarded([&]() -> int {
      if (!out) return fail(MI355RT_E_ARG, "out is null");
      *out = nullptr;
      if (n_meshes && !meshes) return fail(MI355RT_E_ARG, "meshes is null");
      for (uint32_t i = 0; i < n_meshes; i++) {
         if (meshes[i].material.kind > MI355RT_MAT_REFLECTION) return fail(MI355RT_E_ARG, "unknown material kind on mesh " + std::to_string(i));
      }
      std::vector<mi355rt_object_desc> objects(n_meshes ? n_meshes : 1);
      std::memset(objects.data(), 0, sizeof(mi355rt_object_desc) * objects.size());
      for (uint32_t i = 0; i < n_meshes; i++) {
         objects[i].kind = MI355RT_OBJ_MESH;
         objects[i].mesh = meshes[i];
      }
      return mi355rt_scene_create_objects(objects.data(), n_meshes, opts, out);
   });
}

void mi355rt_scene_destroy(mi355rt_scene *scene)
{
   if (!scene) return;
   DeviceScope scope(scene->device);
   release_scene_impl(scene);
}

int mi355rt_scene_view(mi355rt_scene *scene, mi355rt_scene **view_out)
{
   return guarded([&]() -> int {
      if (!scene || !view_out) return fail(MI355RT_E_ARG, "null argument");
      *view_out = nullptr;
      mi355rt_scene *root = scene->parent ? scene->parent : scene;
      if (root->dying) return fail(MI355RT_E_ARG, "the scene has been destroyed");
      DeviceScope scope(root->device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device");
      mi355rt_scene *v = nullptr;
      if (int rc = make_view(root, &v)) return rc;
      if (root->multi) {   // a view of a scene with a device list spans the same devices
         if (int rc = view_multi(root, v)) {
            release_scene_impl(v);
            return rc;
         }
      }
      *view_out = v;
      return MI355RT_OK;
   });
}

int mi355rt_scene_set_devices(mi355rt_scene *scene, const int32_t *devices, int32_t n_devices)
{
   return guarded([&]() -> int {
      if (!scene || !devices) return fail(MI355RT_E_ARG, "null argument");
      DeviceScope scope(scene->device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device");
      return set_devices_impl(scene, devices, n_devices);
   });
}

int mi355rt_scene_multi_info(mi355rt_scene *scene, mi355rt_multi_info *out)
{
   return guarded([&]() -> int {
      if (!scene || !out) return fail(MI355RT_E_ARG, "null argument");
      std::memset(out, 0, sizeof(*out));
      out->n_ranks = 1;
      out->transport = TRANSPORT_NONE;
      out->devices[0] = scene->device;
      const Multi *m = scene->multi;
      if (!m) return MI355RT_OK;
      out->n_ranks = (int32_t) m->ranks.size();
      out->transport = m->shared->transport;
      for (size_t r = 0; r < m->ranks.size(); r++) {
         const RankCtx &rk = m->ranks[r];
         out->devices[r] = rk.device;
         if (!rk.timed) continue;
         DeviceScope scope(rk.device);
         float ms = 0.0f;
         HIP_TRY(hipEventSynchronize(rk.t_end));
         HIP_TRY(hipEventElapsedTime(&ms, rk.t_begin, rk.t_end));
         out->render_ms[r] = ms;
      }
      return MI355RT_OK;
   });
}

int mi355rt_debug_switch(const char *name, const char *value)
{
   return guarded([&]() -> int {
      if (!set_switch(name, value)) return fail(MI355RT_E_ARG, std::string("unknown switch ") + (name ? name : ""));
      return MI355RT_OK;
   });
}

int mi355rt_scene_build_seconds(const mi355rt_scene *scene, double out[7])
{
   return guarded([&]() -> int {
      if (!scene || !out) return fail(MI355RT_E_ARG, "null argument");
      const mi355rt_scene *root = scene->parent ? scene->parent : scene;
      for (int q = 0; q < 7; q++) out[q] = root->build_seconds[q];
      return MI355RT_OK;
   });
}

int mi355rt_debug_scene_layout(mi355rt_scene *scene, uint64_t out[8])
{
   return guarded([&]() -> int {
      if (!scene || !out) return fail(MI355RT_E_ARG, "null argument");
      out[0] = scene->n_node_recs;
      out[1] = scene->n_descs;
      out[2] = scene->n_gface_floats / 12;
      out[3] = scene->n_slots;
      out[4] = scene->n_groups;
      out[5] = scene->n_shared_groups;
      out[6] = scene->big_leaves ? 1 : 0;
      out[7] = (scene->parent ? scene->parent : scene)->device_bytes;
      return MI355RT_OK;
   });
}

int mi355rt_brdf_tables(mi355rt_scene *scene, const float *angles, int32_t n_dirs, const float *weights, int32_t n_materials)
{
   return guarded([&]() -> int {
      if (!scene || !angles || !weights) return fail(MI355RT_E_ARG, "null argument");
      if (scene->parent || scene->n_views > 0) return fail(MI355RT_E_ARG, "set the BRDF tables on the scene before creating views of it");
      if (scene->multi) return fail(MI355RT_E_ARG, "set the BRDF tables on the scene before giving it a device list (the replicas copy them)");
      if (n_dirs < 1 || n_materials < 1) return fail(MI355RT_E_ARG, "table sizes must be positive");
      DeviceScope scope(scene->device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device");
      // BRDFMaterial.remapDirections -> PBRTMath.getRemappedDirection(float[], float[]) PBRTMath.java:74-97
      const float f_pi = (float) jm::kPi, f_2pi = 2.0f * (float) jm::kPi;
      std::vector<float> dirs(3 * (size_t) n_dirs);
      for (int32_t i = 0; i < n_dirs; i++) {
         const float theta_v = angles[4 * i], phi_v = angles[4 * i + 1], theta_i = angles[4 * i + 2], phi_i = angles[4 * i + 3];
         const float cosi = (float) std::cos((double) theta_v), coso = (float) std::cos((double) theta_i);
         const float sini = (float) std::sin((double) theta_v), sino = (float) std::sin((double) theta_i);
         const float phii = ((double) phi_v < jm::kPi / 2.0f) ? phi_v + 2.0f * f_pi : phi_v;   // PBRTMath.normalizePhi
         const float phio = ((double) phi_i < jm::kPi / 2.0f) ? phi_i + 2.0f * f_pi : phi_i;
         float dphi = phii - phio;
         if (dphi < 0.0f) dphi += f_2pi;
         if (dphi > f_2pi) dphi -= f_2pi;
         if (dphi > f_pi) dphi = f_2pi - dphi;
         dirs[3 * (size_t) i + 0] = sini * sino;
         dirs[3 * (size_t) i + 1] = dphi / f_pi;
         dirs[3 * (size_t) i + 2] = cosi * coso;
      }
      if (scene->d_brdf_dirs) (void) hipFree(scene->d_brdf_dirs);
      if (scene->d_brdf_weights) (void) hipFree(scene->d_brdf_weights);
      scene->d_brdf_dirs = scene->d_brdf_weights = nullptr;
      scene->n_dirs = scene->n_materials = 0;
      int rc;
      if ((rc = upload(dirs.data(), dirs.size() * sizeof(float), (void **) &scene->d_brdf_dirs, scene))) return rc;
      if ((rc = upload(weights, sizeof(float) * (size_t) n_dirs * (size_t) n_materials, (void **) &scene->d_brdf_weights, scene))) return rc;
      scene->n_dirs = n_dirs;
      scene->n_materials = n_materials;
      return MI355RT_OK;
   });
}

int mi355rt_scene_epsilon(const mi355rt_scene *scene, float *epsilon_out)
{
   return guarded([&]() -> int {
      if (!scene || !epsilon_out) return fail(MI355RT_E_ARG, "null argument");
      const float span = jm::fmax_java(scene->smax[0] - scene->smin[0],
                                       jm::fmax_java(scene->smax[1] - scene->smin[1], scene->smax[2] - scene->smin[2]));
      *epsilon_out = (span < 10) ? 1e-15f : 1e-3f;   // Scene.java:35-36
      return MI355RT_OK;
   });
}

int mi355rt_scene_object_info(const mi355rt_scene *scene, uint32_t object, float minmax[6], float *shadow_distance,
                              uint64_t *n_nodes, uint64_t *n_leaves, uint64_t *n_leaf_slots, int32_t *max_depth)
{
   return guarded([&]() -> int {
      if (!scene) return fail(MI355RT_E_ARG, "scene is null");
      if (object >= scene->objects.size()) return fail(MI355RT_E_ARG, "object index out of range");
      const HostObject &o = scene->objects[object];
      if (minmax) {
         std::memcpy(minmax, o.bmin, sizeof(o.bmin));
         std::memcpy(minmax + 3, o.bmax, sizeof(o.bmax));
      }
      if (shadow_distance) *shadow_distance = o.shadow_distance;
      if (n_nodes) *n_nodes = o.n_nodes;
      if (n_leaves) *n_leaves = o.n_leaves;
      if (n_leaf_slots) *n_leaf_slots = o.n_slots;
      if (max_depth) *max_depth = o.max_depth;
      return MI355RT_OK;
   });
}

int mi355rt_scene_device_bytes(const mi355rt_scene *scene, uint64_t *bytes_out)
{
   return guarded([&]() -> int {
      if (!scene || !bytes_out) return fail(MI355RT_E_ARG, "null argument");
      *bytes_out = scene->device_bytes;
      return MI355RT_OK;
   });
}

int mi355rt_camera_setup(const mi355rt_scene *scene, const float orientation_deg[3], float fov_deg, float near_plane, int32_t width,
                         int32_t height, mi355rt_camera *out)
{
   return guarded([&]() -> int {
      if (!scene || !orientation_deg || !out) return fail(MI355RT_E_ARG, "null argument");
      if (width < 1 || height < 1) return fail(MI355RT_E_ARG, "image size must be positive");
      out->near_plane = near_plane;
      out->width = width;
      out->height = height;
      // Camera.java:50-51
      out->viewport_w = (width >= height ? (float) width / (float) height : 1.0f) * near_plane;
      out->viewport_h = (width >= height ? 1.0f : (float) height / (float) width) * near_plane;
      // RTStatics.initializeQuat4f(float[]) RTStatics.java:345-354: yaw(-o0) * pitch(o1) * roll(-o2)
      const jm::Quat yaw = jm::quat_from_axis_angle({ 0.f, 1.f, 0.f }, -orientation_deg[0]);
      const jm::Quat pitch = jm::quat_from_axis_angle({ 1.f, 0.f, 0.f }, orientation_deg[1]);
      const jm::Quat roll = jm::quat_from_axis_angle({ 0.f, 0.f, 1.f }, -orientation_deg[2]);
      jm::quat_to_rows(jm::quat_mul(jm::quat_mul(yaw, pitch), roll), out->rotation);
      // Camera.java:72-81
      const float w = scene->smax[0] - scene->smin[0];
      const float h = scene->smax[1] - scene->smin[1];
      const float d = scene->smax[2] - scene->smin[2];
      const float cx = w / 2.0f + scene->smin[0];
      const float cy = h / 2.0f + scene->smin[1];
      const float cz = d / 2.0f + scene->smin[2];
      const float distance = w / 2.0f / (float) std::tan(jm::to_radians((double) fov_deg));
      const jm::Vec3 o = jm::rotate(out->rotation, { cx, cy, cz + distance });
      out->origin[0] = o.x;
      out->origin[1] = o.y;
      out->origin[2] = o.z;
      return MI355RT_OK;
   });
}

int mi355rt_render_device(mi355rt_scene *scene, const mi355rt_camera *cam, const mi355rt_settings *settings, const mi355rt_frame *frame,
                          void *hip_stream)
{
   return guarded([&]() -> int {
      if (!scene || !frame || !settings) return fail(MI355RT_E_ARG, "null argument");
      DeviceScope scope(scene->device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device");
      return render_any(scene, cam, settings, frame, (hipStream_t) hip_stream);
   });
}

int mi355rt_render_device_noalloc(mi355rt_scene *scene, const mi355rt_camera *cam, const mi355rt_settings *settings, const mi355rt_frame *frame,
                                  void *hip_stream)
{
   return guarded([&]() -> int {
      if (!scene || !frame || !settings) return fail(MI355RT_E_ARG, "null argument");
      if (settings->collect_stats != 0) return fail(MI355RT_E_ARG, "mi355rt_render_device_noalloc takes collect_stats == 0 (counting and timing create events and read back)");
      FrameArgs p;
      long long n_micro = 0;
      if (int rc = fill_frame_args(scene, cam, settings, p, n_micro)) return rc;
      DeviceScope scope(scene->device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device");
      unsigned char *base = (unsigned char *) scene->noalloc_scratch;   // resolved at scene creation (hipGetSymbolAddress), per device
      if (!base) return fail(MI355RT_E_DEVICE, "the static scratch array was not resolved on this device");
      // the whole frame is one pass: every band's sub-queue of every record array must fit the array
      const bool with_mirror = scene->has_mirror || scene->has_refraction, with_refraction = scene->has_refraction;
      const unsigned long long slots = (unsigned long long) ((n_micro + N_BANDS - 1) / N_BANDS) * N_BANDS * 64ull * (unsigned long long) settings->multi_samples;
      const unsigned long long per_slot = 104ull + (with_mirror ? 100ull : 0ull) + (with_refraction ? 16ull * (1 + MAX_REFRACTION_TERMS) : 0ull);
      constexpr unsigned long long HEAD = 4096, ALIGN_SLACK = 13 * 256;   // counters; each of up to 13 arrays starts on a 256-byte boundary
      if (slots * per_slot + slots / 8 + HEAD + ALIGN_SLACK > (unsigned long long) MI355RT_NOALLOC_SCRATCH_BYTES)
         return fail(MI355RT_E_LIMIT, "mi355rt_render_device_noalloc: " + std::to_string(slots) + " sample slots x " + std::to_string(per_slot) +
                                          " bytes exceed the static scratch of " + std::to_string((unsigned long long) MI355RT_NOALLOC_SCRATCH_BYTES) +
                                          " bytes; use mi355rt_render_device for this frame");
      static_assert(CNT_WORDS * sizeof(uint32_t) <= 4096, "the counter block sits in the first 4 KiB of the scratch");
      FrameScratch fs;
      size_t off = HEAD;
      auto carve = [&](size_t bytes) -> void * {
         void *q = base + off;
         off = (off + bytes + 255) & ~(size_t) 255;
         return q;
      };
      fs.cap = (size_t) slots;
      fs.with_mirror = with_mirror;
      fs.with_refraction = with_refraction;
      fs.queued = (unsigned long long *) carve(slots / 8);
      fs.samples = (float *) carve(slots * 16);
      fs.rays0 = (float *) carve(slots * 32);
      fs.hits0 = (float *) carve(slots * 32);
      fs.mat0 = (float *) carve(slots * 16);
      fs.hit_obj0 = (int32_t *) carve(slots * 4);
      fs.shadow = (uint32_t *) carve(slots * 4);
      if (with_mirror) {
         fs.rays1[0] = (float *) carve(slots * 32);
         fs.rays1[1] = (float *) carve(slots * 32);
         fs.hits1 = (float *) carve(slots * 32);
         fs.hit_obj1 = (int32_t *) carve(slots * 4);
      }
      if (with_refraction) {
         fs.refr_parent = (float *) carve(slots * 16);
         fs.refr_terms = (float *) carve(slots * 16 * MAX_REFRACTION_TERMS);
      }
      p.rgb = frame->rgb;
      p.argb8 = frame->argb8;
      p.hit_id = frame->hit_id;
      p.hit_obj = frame->hit_obj;
      p.hit_w = frame->hit_w;
      // the array keeps the previous call's contents: run_frame clears the counter block at the start of the pass (the
      // only part that must start at zero); every queue record is written before it is read
      return run_frame(scene, p, n_micro, false, (hipStream_t) hip_stream, &fs, (uint32_t *) base);
   });
}

int mi355rt_tiles_unpack_device(const float *blocks_dev, int32_t world, int32_t slots_per_rank, int32_t tile_size, int32_t width, int32_t height,
                                const mi355rt_frame *frame, void *hip_stream)
{
   return guarded([&]() -> int {
      if (!blocks_dev || !frame) return fail(MI355RT_E_ARG, "null argument");
      if (world < 1 || slots_per_rank < 0 || width < 1 || height < 1) return fail(MI355RT_E_ARG, "sizes must be positive");
      if (tile_size < 8 || tile_size % 8 != 0 || tile_size > 4096) return fail(MI355RT_E_ARG, "tile_size must be a multiple of 8 in [8, 4096]");
      const long long tiles = (long long) ((width + tile_size - 1) / tile_size) * ((height + tile_size - 1) / tile_size);
      if ((long long) slots_per_rank != (tiles + world - 1) / world)
         return fail(MI355RT_E_ARG, "slots_per_rank must be ceil(tiles / world) = " + std::to_string((tiles + world - 1) / world));
      HIP_TRY(launch_tiles_unpack(blocks_dev, world, slots_per_rank, tile_size, width, height, frame->rgb, frame->argb8, frame->hit_id,
                                  frame->hit_obj, frame->hit_w, (hipStream_t) hip_stream));
      return MI355RT_OK;
   });
}

int mi355rt_render_host(mi355rt_scene *scene, const mi355rt_camera *cam, const mi355rt_settings *settings, const mi355rt_host_frame *out)
{
   return guarded([&]() -> int {
      if (!scene || !cam || !settings || !out) return fail(MI355RT_E_ARG, "null argument");
      if (cam->width < 1 || cam->height < 1) return fail(MI355RT_E_ARG, "image size must be positive");
      if (settings->tile_compact) return fail(MI355RT_E_ARG, "tile_compact is for mi355rt_render_device (this call returns whole frames in image order)");
      DeviceScope scope(scene->device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device");
      const size_t pixels = (size_t) cam->width * (size_t) cam->height;
      if (pixels > scene->frame_pixels) {
         void **all[] = { (void **) &scene->d_rgb, (void **) &scene->d_argb8, (void **) &scene->d_hit_id, (void **) &scene->d_hit_obj, (void **) &scene->d_hit_w };
         for (void **q : all) {
            if (*q) (void) hipFree(*q);
            *q = nullptr;
         }
         scene->frame_pixels = 0;
         HIP_TRY(hipMalloc((void **) &scene->d_rgb, pixels * 3 * sizeof(float)));
         HIP_TRY(hipMalloc((void **) &scene->d_argb8, pixels * sizeof(uint32_t)));
         HIP_TRY(hipMalloc((void **) &scene->d_hit_id, pixels * sizeof(int32_t)));
         HIP_TRY(hipMalloc((void **) &scene->d_hit_obj, pixels * sizeof(int32_t)));
         HIP_TRY(hipMalloc((void **) &scene->d_hit_w, pixels * sizeof(float)));
         scene->frame_pixels = pixels;
      }
      // an output nobody asked for is neither written nor copied
      struct Out {
         void *host, *dev;
         size_t bytes;
      } outs[] = { { out->rgb, scene->d_rgb, pixels * 3 * sizeof(float) },       { out->argb8, scene->d_argb8, pixels * sizeof(uint32_t) },
                   { out->hit_id, scene->d_hit_id, pixels * sizeof(int32_t) },   { out->hit_obj, scene->d_hit_obj, pixels * sizeof(int32_t) },
                   { out->hit_w, scene->d_hit_w, pixels * sizeof(float) } };
      if (settings->tile_world > 1)   // tiles owned by other ranks stay zero
         for (const Out &o : outs)
            if (o.host) HIP_TRY(hipMemsetAsync(o.dev, 0, o.bytes, nullptr));
      mi355rt_frame fr;
      std::memset(&fr, 0, sizeof(fr));
      fr.rgb = out->rgb ? scene->d_rgb : nullptr;
      fr.argb8 = out->argb8 ? scene->d_argb8 : nullptr;
      fr.hit_id = out->hit_id ? scene->d_hit_id : nullptr;
      fr.hit_obj = out->hit_obj ? scene->d_hit_obj : nullptr;
      fr.hit_w = out->hit_w ? scene->d_hit_w : nullptr;
      if (int rc = render_any(scene, cam, settings, &fr, nullptr)) return rc;
      // the copies queue behind the frame on the same stream, one after the other, and the host waits once
      for (const Out &o : outs)
         if (o.host) HIP_TRY(hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, nullptr));
      HIP_TRY(hipStreamSynchronize(nullptr));
      return MI355RT_OK;
   });
}

int mi355rt_render(mi355rt_scene *scene, const mi355rt_camera *cam, const mi355rt_settings *settings, float *rgb_out,
                   int32_t *hit_id_out, int32_t *hit_obj_out, float *hit_w_out)
{
   mi355rt_host_frame out;
   std::memset(&out, 0, sizeof(out));
   out.rgb = rgb_out;
   out.hit_id = hit_id_out;
   out.hit_obj = hit_obj_out;
   out.hit_w = hit_w_out;
   return mi355rt_render_host(scene, cam, settings, &out);
}

int mi355rt_host_alloc(uint64_t bytes, void **ptr_out)
{
   return guarded([&]() -> int {
      if (!ptr_out || !bytes) return fail(MI355RT_E_ARG, "null pointer or zero size");
      *ptr_out = nullptr;
      HIP_TRY(hipHostMalloc(ptr_out, (size_t) bytes, hipHostMallocDefault));
      return MI355RT_OK;
   });
}

int mi355rt_host_free(void *ptr)
{
   return guarded([&]() -> int {
      if (!ptr) return MI355RT_OK;
      HIP_TRY(hipHostFree(ptr));
      return MI355RT_OK;
   });
}

int mi355rt_frame_stats(mi355rt_scene *scene, mi355rt_stats *out)
{
   return guarded([&]() -> int {
      if (!scene || !out) return fail(MI355RT_E_ARG, "null argument");
      if (!scene->stats_valid) return fail(MI355RT_E_ARG, "no render with collect_stats has run on this scene");
      *out = scene->stats_accum;   // run_frame reads the counters back at the end of each pass
      if (scene->multi) {          // a frame over several GPUs: every rank counted its own tiles
         for (size_t r = 1; r < scene->multi->ranks.size(); r++) {
            const mi355rt_stats &q = scene->multi->ranks[r].s->stats_accum;
            out->primary_rays += q.primary_rays;
            out->shadow_rays += q.shadow_rays;
            out->secondary_rays += q.secondary_rays;
            out->box_tests += q.box_tests;
            out->tri_tests += q.tri_tests;
            out->hits += q.hits;
         }
      }
      return MI355RT_OK;
   });
}

int mi355rt_image_argb8_device(const float *rgb_dev, int32_t width, int32_t height, int normalized, uint32_t *argb_dev,
                               uint32_t *minmax_scratch, void *hip_stream)
{
   return guarded([&]() -> int {
      if (!rgb_dev || !argb_dev || (normalized && !minmax_scratch)) return fail(MI355RT_E_ARG, "null argument");
      if (width < 1 || height < 1) return fail(MI355RT_E_ARG, "image size must be positive");
      int device = 0, cus = 256;
      HIP_TRY(hipGetDevice(&device));
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
      HIP_TRY(launch_image(rgb_dev, (unsigned long long) width * (unsigned long long) height, normalized, minmax_scratch, argb_dev, cus,
                           (hipStream_t) hip_stream));
      return MI355RT_OK;
   });
}

int mi355rt_image_argb8(const float *rgb, int32_t width, int32_t height, int normalized, uint32_t *argb_out, int32_t device)
{
   return guarded([&]() -> int {
      if (!rgb || !argb_out) return fail(MI355RT_E_ARG, "null argument");
      if (width < 1 || height < 1) return fail(MI355RT_E_ARG, "image size must be positive");
      int n_dev = 0;
      if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return fail(MI355RT_E_DEVICE, "no HIP device available (libmi355rt has no CPU path)");
      if (device < 0 && hipGetDevice(&device) != hipSuccess) return fail(MI355RT_E_DEVICE, "hipGetDevice failed");
      if (device >= n_dev) return fail(MI355RT_E_ARG, "device ordinal out of range");
      DeviceScope scope(device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device");
      const size_t n = (size_t) width * (size_t) height;
      float *d_rgb = nullptr;
      uint32_t *d_out = nullptr;   // n pixels + 2 words of min / max
      hipError_t e = hipMalloc((void **) &d_rgb, n * 3 * sizeof(float));
      if (e == hipSuccess) e = hipMalloc((void **) &d_out, (n + 2) * sizeof(uint32_t));
      if (e == hipSuccess) e = hipMemcpy(d_rgb, rgb, n * 3 * sizeof(float), hipMemcpyHostToDevice);
      int rc = MI355RT_OK;
      if (e == hipSuccess) rc = mi355rt_image_argb8_device(d_rgb, width, height, normalized, d_out, d_out + n, nullptr);
      if (e == hipSuccess && rc == MI355RT_OK) e = hipMemcpy(argb_out, d_out, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
      if (d_rgb) (void) hipFree(d_rgb);
      if (d_out) (void) hipFree(d_out);
      if (e != hipSuccess) return hip_fail(e, "mi355rt_image_argb8");
      return rc;
   });
}

int mi355rt_kernel_profile(mi355rt_scene *scene, mi355rt_kernel_times *out, int reset)
{
   return guarded([&]() -> int {
      if (!scene || !out) return fail(MI355RT_E_ARG, "null argument");
      DeviceScope scope(scene->device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device");
      if (int rc = resolve_times(scene)) return rc;
      std::memset(out, 0, sizeof(*out));
      for (int k = 0; k < MI355RT_N_KERNELS; k++) {
         out->ms[k] = scene->kernel_ms[k];
         out->launches[k] = scene->kernel_launches[k];
      }
      if (scene->stats_valid) {
         const int kinds[3] = { MI355RT_K_WALK_PRIMARY, MI355RT_K_WALK_SECONDARY, MI355RT_K_WALK_SHADOW };
         for (int q = 0; q < 3; q++) {   // trace_kernels.hip: ST_KIND_RAYS = 16, ST_KIND_BOX = 19, ST_KIND_TRI = 22
            out->rays[kinds[q]] = scene->diag[16 + q];
            out->box_tests[kinds[q]] = scene->diag[19 + q];
            out->tri_tests[kinds[q]] = scene->diag[22 + q];
         }
      }
      if (reset) {
         for (int k = 0; k < MI355RT_N_KERNELS; k++) {
            scene->kernel_ms[k] = 0.0;
            scene->kernel_launches[k] = 0;
         }
      }
      return MI355RT_OK;
   });
}

int mi355rt_debug_counters(mi355rt_scene *scene, uint64_t out[64])
{
   return guarded([&]() -> int {
      if (!scene || !out) return fail(MI355RT_E_ARG, "null argument");
      if (!scene->stats_valid) return fail(MI355RT_E_ARG, "no render with collect_stats has run on this scene");
      static_assert(N_STATS == 64, "mi355rt_debug_counters hands out N_STATS values");
      for (int q = 0; q < N_STATS; q++) out[q] = scene->diag[q];
      return MI355RT_OK;
   });
}

int mi355rt_debug_scene_digest(mi355rt_scene *scene, uint64_t out[4])
{
   return guarded([&]() -> int {
      if (!scene || !out) return fail(MI355RT_E_ARG, "null argument");
      DeviceScope scope(scene->device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device");
      // out[1] runs over the three leaf arrays one after the other: descriptors, group faces, slot ids
      const void *src[5] = { scene->d_nodes, scene->d_descs, scene->d_gfaces, scene->d_slot_ids, scene->d_fnrm };
      const size_t bytes[5] = { scene->n_node_recs * sizeof(DevNodeRec), scene->n_descs * sizeof(LeafDesc), scene->n_gface_floats * sizeof(float),
                                scene->n_slots * sizeof(int32_t), scene->n_fnrm_floats * sizeof(float) };
      const int dst[5] = { 0, 1, 1, 1, 2 };
      for (int k = 0; k < 3; k++) out[k] = 1469598103934665603ull;   // FNV-1a
      for (int k = 0; k < 5; k++) {
         std::vector<unsigned char> host(bytes[k]);
         if (bytes[k]) HIP_TRY(hipMemcpy(host.data(), src[k], bytes[k], hipMemcpyDeviceToHost));
         uint64_t h = out[dst[k]];
         for (unsigned char c : host) h = (h ^ c) * 1099511628211ull;
         out[dst[k]] = h;
      }
      out[3] = scene->any_gpu_build ? 1u : 0u;
      return MI355RT_OK;
   });
}

int mi355rt_trace_rays(mi355rt_scene *scene, const mi355rt_settings *settings, uint64_t n, const float *origins,
                       const float *directions, int32_t *hit_id, int32_t *hit_obj, float *hit_w, float *hit_point, float *hit_normal)
{
   return guarded([&]() -> int {
      if (!scene || !settings) return fail(MI355RT_E_ARG, "null argument");
      if (n == 0) return MI355RT_OK;
      if (!origins || !directions) return fail(MI355RT_E_ARG, "null ray arrays");
      if (n > ((uint64_t) 1 << 30)) return fail(MI355RT_E_LIMIT, "more than 2^30 rays in one call");
      DeviceScope scope(scene->device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device");

      struct Buffers {
         void *p[10] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
         ~Buffers()
         {
            for (void *q : p)
               if (q) (void) hipFree(q);
         }
      } buf;
      // 0 origins, 1 dirs, 2 rays, 3 hits, 4 hit_obj, 5 out_id, 6 out_obj, 7 out_w, 8 out_point, 9 out_normal
      const size_t sizes[10] = { n * 12, n * 12, n * 32, n * 32, n * 4, n * 4, n * 4, n * 4, n * 12, n * 12 };
      for (int i = 0; i < 10; i++) HIP_TRY(hipMalloc(&buf.p[i], sizes[i]));
      HIP_TRY(hipMemcpy(buf.p[0], origins, sizes[0], hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(buf.p[1], directions, sizes[1], hipMemcpyHostToDevice));
      TraceBatchArgs a;
      std::memset(&a, 0, sizeof(a));
      a.sc.objs = scene->d_objs;
      a.sc.nodes = scene->d_nodes;
      a.sc.gfaces = scene->d_gfaces;
      a.sc.descs = scene->d_descs;
      a.sc.slot_ids = scene->d_slot_ids;
      a.sc.fnrm = scene->d_fnrm;
      a.sc.n_objs = (int32_t) scene->objects.size();
      a.sc.epsilon = settings->epsilon;
      a.sc.near_plane = settings->near_plane;
      a.sc.far_plane = settings->far_plane;
      a.sc.scene_abs_max = scene->abs_max;
      a.sc.max_normal_len = scene->max_normal_len;
      a.sc.ext = scene->d_ext;
      a.sc.n_analytic = scene->n_analytic;
      a.sc.boxes_finite = scene->boxes_finite ? 1 : 0;
      a.sc.big_leaves = scene->big_leaves ? 1 : 0;
      a.n = (uint32_t) n;
      a.band_cap = (uint32_t) ((n + N_BANDS - 1) / N_BANDS);
      a.origins = (const float *) buf.p[0];
      a.dirs = (const float *) buf.p[1];
      a.rays = (float *) buf.p[2];
      a.hits = (float *) buf.p[3];
      a.hit_obj = (int32_t *) buf.p[4];
      a.counters = scene->d_counters;
      a.out_id = (int32_t *) buf.p[5];
      a.out_obj = (int32_t *) buf.p[6];
      a.out_w = (float *) buf.p[7];
      a.out_point = (float *) buf.p[8];
      a.out_normal = (float *) buf.p[9];
      HIP_TRY(hipMemsetAsync(scene->d_counters, 0, CNT_WORDS * sizeof(uint32_t), nullptr));
      HIP_TRY(launch_batch_pack(a, nullptr));
      HIP_TRY(launch_trace(a.sc, a.rays, a.hits, a.hit_obj, a.counters, CNT_Q0, CNT_CURSOR0, a.band_cap, n, scene->n_cus, 0, 0, nullptr));
      HIP_TRY(launch_analytic_closest(a.sc, a.rays, a.hits, a.hit_obj, a.counters, CNT_Q0, a.band_cap, n, scene->n_cus, 0, nullptr));
      HIP_TRY(launch_batch_unpack(a, nullptr));
      HIP_TRY(hipStreamSynchronize(nullptr));
      if (hit_id) HIP_TRY(hipMemcpy(hit_id, buf.p[5], sizes[5], hipMemcpyDeviceToHost));
      if (hit_obj) HIP_TRY(hipMemcpy(hit_obj, buf.p[6], sizes[6], hipMemcpyDeviceToHost));
      if (hit_w) HIP_TRY(hipMemcpy(hit_w, buf.p[7], sizes[7], hipMemcpyDeviceToHost));
      if (hit_point) HIP_TRY(hipMemcpy(hit_point, buf.p[8], sizes[8], hipMemcpyDeviceToHost));
      if (hit_normal) HIP_TRY(hipMemcpy(hit_normal, buf.p[9], sizes[9], hipMemcpyDeviceToHost));
      return MI355RT_OK;
   });
}

int mi355rt_debug_arith(uint64_t n, const float *a, const float *b, float *out_div, float *out_sqrt, float *out_rnorm, float *out_pow)
{
   return guarded([&]() -> int {
      if (!a || !b || !out_div || !out_sqrt || !out_rnorm || !out_pow) return fail(MI355RT_E_ARG, "null argument");
      if (n == 0) return MI355RT_OK;
      int n_dev = 0;
      if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return fail(MI355RT_E_DEVICE, "no HIP device available");
      float *d[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
      int rc = MI355RT_OK;
      for (int i = 0; i < 6 && rc == MI355RT_OK; i++) {
         hipError_t e = hipMalloc((void **) &d[i], n * sizeof(float));
         if (e != hipSuccess) rc = hip_fail(e, "hipMalloc");
      }
      if (rc == MI355RT_OK) {
         hipError_t e = hipMemcpy(d[0], a, n * sizeof(float), hipMemcpyHostToDevice);
         if (e == hipSuccess) e = hipMemcpy(d[1], b, n * sizeof(float), hipMemcpyHostToDevice);
         if (e == hipSuccess) e = launch_arith_probe(n, d[0], d[1], d[2], d[3], d[4], d[5], nullptr);
         if (e == hipSuccess) e = hipDeviceSynchronize();
         if (e == hipSuccess) e = hipMemcpy(out_div, d[2], n * sizeof(float), hipMemcpyDeviceToHost);
         if (e == hipSuccess) e = hipMemcpy(out_sqrt, d[3], n * sizeof(float), hipMemcpyDeviceToHost);
         if (e == hipSuccess) e = hipMemcpy(out_rnorm, d[4], n * sizeof(float), hipMemcpyDeviceToHost);
         if (e == hipSuccess) e = hipMemcpy(out_pow, d[5], n * sizeof(float), hipMemcpyDeviceToHost);
         if (e != hipSuccess) rc = hip_fail(e, "arith probe");
      }
      for (float *q : d)
         if (q) (void) hipFree(q);
      return rc;
   });
}

int mi355rt_ply_load(const char *path, uint32_t *n_vertices, uint32_t *n_faces, float **vertices, float **normals, int32_t **indices,
                     int32_t *has_normals)
{
   return guarded([&]() -> int {
      if (!path || !n_vertices || !n_faces || !vertices || !normals || !indices || !has_normals) return fail(MI355RT_E_ARG, "null argument");
      PlyMesh m;
      const std::string err = read_ply(path, m);
      if (!err.empty()) return fail(MI355RT_E_IO, std::string(path) + ": " + err);
      // the reader's arrays are malloc()ed: they go to the caller as they are (mi355rt_free)
      *n_vertices = (uint32_t) m.n_vertices;
      *n_faces = (uint32_t) m.n_faces;
      *vertices = m.vertices;
      *normals = m.normals;
      *indices = m.indices;
      m.vertices = m.normals = nullptr;
      m.indices = nullptr;
      *has_normals = m.has_normals ? 1 : 0;
      return MI355RT_OK;
   });
}

int mi355rt_compute_normals(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces, float *normals_out)
{
   return guarded([&]() -> int {
      if ((n_vertices && !vertices) || (n_faces && !indices) || !normals_out) return fail(MI355RT_E_ARG, "null argument");
      for (size_t i = 0; i < 3 * (size_t) n_faces; i++) {
         if (indices[i] < 0 || (uint32_t) indices[i] >= n_vertices) return fail(MI355RT_E_ARG, "face index out of range");
      }
      // on the device (face-order sums per vertex: the host's bits) when there is one; the loader half of the library works without
      int n_dev = 0;
      if (n_faces > 50000 && hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0) {
         if (gpu_compute_normals(vertices, n_vertices, indices, n_faces, normals_out) == hipSuccess) return MI355RT_OK;
         (void) hipGetLastError();
      }
      synthesise_normals(vertices, n_vertices, indices, n_faces, normals_out);
      return MI355RT_OK;
   });
}

void mi355rt_free(void *p) { std::free(p); }

}   // extern "C"
