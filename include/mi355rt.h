/*
 * mi355rt.h -- C ABI of libmi355rt.so, the MI355X (gfx950) implementation of the per-pixel trace
 * loop of sranger/Ray-Tracer.
 *
 * The reference has no plugin / FFI interface; the seam is the body of
 * Camera.createImage() (Camera.java:119-185).  Every entry point below names the reference
 * code it replaces; paths are relative to /root/reference/src/stephen/ranger/ar/.
 *
 * Conventions: plain C, POD structs, no C++ or torch types.  Every function returns 0 on success
 * or a negative MI355RT_E_* code; the message is available from mi355rt_last_error() (thread
 * local).  Callers own every buffer they pass; the library copies inputs during scene_create and
 * keeps no caller pointer past a call.  There is no CPU fallback: without a HIP device every
 * compute entry point fails with MI355RT_E_DEVICE.
 */
#ifndef MI355RT_H
#define MI355RT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355RT_ABI_VERSION 8   /* 2: mi355rt_scene_create_objects, analytic objects, refraction / checkerboard materials; 3: mi355rt_host_alloc / _free;
                                 * 4: mi355rt_render_device_noalloc; 5: mi355rt_render returns hit_obj, mi355rt_mesh_desc.compute_kdtree;
                                 * 6: mi355rt_settings.tile_compact, mi355rt_tiles_unpack_device;
                                 * 7: mi355rt_scene_view;
                                 * 8: mi355rt_scene_set_devices / mi355rt_scene_multi_info (several GPUs behind one handle), mi355rt_render_host
                                 *    (8-bit image as a host output), mi355rt_debug_switch, mi355rt_tiles_unpack_device writes argb8 */

enum {
   MI355RT_OK = 0,
   MI355RT_E_ARG = -1,      /* bad argument (null pointer, negative size, index out of range) */
   MI355RT_E_DEVICE = -2,   /* no HIP device / HIP runtime error */
   MI355RT_E_MEMORY = -3,   /* host or device allocation failed */
   MI355RT_E_LIMIT = -4,    /* a documented limit was exceeded (tree depth, scratch size ...) */
   MI355RT_E_IO = -5        /* PLY file unreadable or malformed */
};

/* materials/ColorInformation.java:15-37, materials/BRDFMaterial.java:47-51,158-167,
 * materials/ReflectionMaterial.java:10-24 */
enum { MI355RT_MAT_COLOR = 0, MI355RT_MAT_BRDF = 1, MI355RT_MAT_REFLECTION = 2,
       MI355RT_MAT_REFRACTION = 3,     /* materials/RefractionMaterial.java:29-95 (mi355rt_material_ext.refraction_index) */
       MI355RT_MAT_CHECKERBOARD = 4 }; /* materials/CheckerboardMaterial.java:21-49 (mi355rt_material_ext.checker_*) */
/* Scene.objects entries (next row: SURVEY.md 8f rank 2): bounds/KDTree.java over a TriangleMesh (0),
 * sceneObjects/Sphere.java:29-112 inside its bounds/BoundingSphere.java:29-102 (1),
 * sceneObjects/Plane.java:28-154 inside its bounds/AxisAlignedBoundingBox.java (2) */
enum { MI355RT_OBJ_MESH = 0, MI355RT_OBJ_SPHERE = 1, MI355RT_OBJ_PLANE = 2 };
/* Guard with no counterpart in the reference (like max_mirror_bounces): a path carries at most this many pending
 * RefractionMaterial absorption terms (RefractionMaterial.java:43-53); a further RefractionMaterial hit on that
 * path answers with its diffuse colour, as SceneObject.getColor does at its depth cut-off. */
#define MI355RT_MAX_REFRACTION_TERMS 8
/* lighting/LightingModel.java:13-15 (0), lighting/PhongLightingModel.java:18-54 (1) */
enum { MI355RT_LIGHTING_PLAIN = 0, MI355RT_LIGHTING_PHONG = 1 };

typedef struct {
   uint32_t kind;
   float emission[3], ambient[3], diffuse[3], specular[3]; /* Color.getColorComponents() values */
   float shininess;                                        /* 20 by default, 100 for ReflectionMaterial */
   int32_t brdf_id;                                        /* row of the weights table (BRDFMaterial.java:50) */
} mi355rt_material;

/* the three arrays TriangleMesh hands to KDTree (sceneObjects/TriangleMesh.java:19-21,89), flattened */
typedef struct {
   const float *vertices;  /* n_vertices * 3 */
   const float *normals;   /* n_vertices * 3, or NULL: synthesised as TriangleMesh.computeNormals (:96-126) */
   const int32_t *indices; /* n_faces * 3 */
   uint32_t n_vertices, n_faces;
   mi355rt_material material;
   int32_t compute_kdtree; /* this TriangleMesh's own computeKDTree argument (TriangleMesh.java:23,89): 0 = build_opts.use_kdtree
                            * decides, > 0 build the hierarchy, < 0 keep every face in one leaf */
} mi355rt_mesh_desc;

typedef struct {
   float refraction_index;                     /* RefractionMaterial.refractionIndex (INDEX_OF_GLASS 1.52f ...) */
   float checker_color1[3], checker_color2[3]; /* CheckerboardMaterial.color1 / color2 components */
   float checker_cell[3];                      /* cellWidth, cellHeight, cellDepth */
} mi355rt_material_ext;

/* one entry of Scene.objects */
typedef struct {
   uint32_t kind;             /* MI355RT_OBJ_* */
   mi355rt_mesh_desc mesh;    /* MESH: the arrays.  Every kind: mesh.material is the object's ColorInformation */
   mi355rt_material_ext ext;  /* read for MI355RT_MAT_REFRACTION / MI355RT_MAT_CHECKERBOARD */
   float center[3], radius;   /* SPHERE: new Sphere(radius, origin, colorInfo) (Sphere.java:15-22) */
   float corners[12];         /* PLANE: new Plane(corners, colorInfo) (Plane.java:28-76), 3 floats per corner */
   uint32_t n_corners;        /* 3 or 4 */
} mi355rt_object_desc;

typedef struct {
   int32_t max_depth;    /* RTStatics.MAX_DEPTH = 20    (RTStatics.java:25) */
   int32_t max_children; /* RTStatics.MAX_CHILDREN = 5  (RTStatics.java:24) */
   int32_t use_kdtree;   /* the GUI's KD-tree checkbox / computeKDTree (bounds/KDNode.java:40) */
   int32_t device;       /* HIP device ordinal; -1 = the calling thread's current device */
} mi355rt_build_opts;

/* what the Camera constructor derives (Camera.java:40-94) */
typedef struct {
   float origin[3];
   float rotation[9]; /* row-major 3x3 of Camera.rotation (m00 m01 m02 m10 ...) */
   float viewport_w, viewport_h, near_plane;
   int32_t width, height;
} mi355rt_camera;

/* RTStatics settings + Camera/Scene fields that stay on the Java side and are passed per render */
typedef struct {
   float epsilon;               /* RTStatics.EPSILON as set by the Scene ctor (Scene.java:35-36) */
   float near_plane, far_plane; /* RTStatics.NEAR_PLANE 0.01f / FAR_PLANE 3000f (RTStatics.java:19-20) */
   int32_t max_recursion_depth; /* RTStatics.MAX_RECURSION_DEPTH = 5: SceneObject.getColor's cut-off for spheres and planes
                                 * (SceneObject.java:65-67); unused by mesh objects: KDTree.getColor has no cut-off */
   int32_t multi_samples;       /* Camera.multiSamples */
   int32_t brdf_samples;        /* Camera.brdfSamples */
   int32_t lighting_model;      /* MI355RT_LIGHTING_* */
   int32_t max_mirror_bounces;  /* guard where the reference would recurse without bound (bounds/KDTree.java:77-80) */
   uint64_t rng_seed;           /* the reference's jitter RNG is unseeded (RenderThread.java:32); see DESIGN.md */
   float light_origin[3], light_emission[3], light_ambient[3]; /* lighting/Light.java:7-16 */
   int32_t tile_size;           /* image-tile sharding: square tiles of this many pixels (multiple of 8; 0 = 16) */
   int32_t tile_rank, tile_world; /* this process renders tiles t with t % tile_world == tile_rank */
   int32_t collect_stats;       /* bit mask: MI355RT_COLLECT_COUNTS (count rays / box tests / triangle tests; slower; see
                                 * mi355rt_frame_stats), MI355RT_COLLECT_TIMES (HIP events around every kernel; see
                                 * mi355rt_kernel_profile) */
   int32_t tile_compact;        /* != 0 (mi355rt_render_device only): the frame's arrays are in tile-compact order -- pixel (x, y)
                                 * of this rank's k-th owned tile (tile t = tile_rank + k * tile_world) is element
                                 * k * tile_size^2 + (y mod tile_size) * tile_size + (x mod tile_size); pixels of a tile that hang
                                 * off the image are not written.  It is the form the one exchange step of a multi-GPU frame
                                 * sends as it is (mi355rt_tiles_unpack_device puts it back in image order) */
} mi355rt_settings;

typedef struct {
   uint64_t primary_rays, shadow_rays, secondary_rays;
   uint64_t box_tests, tri_tests, hits;
} mi355rt_stats;

#define MI355RT_COLLECT_COUNTS 1
#define MI355RT_COLLECT_TIMES 2
#define MI355RT_COLLECT_DIAG 4   /* counts plus the walker's lane-utilisation diagnostics (mi355rt_debug_counters); slower */

/* the kernels of one frame (the wavefront pipeline of DESIGN.md), as indices into mi355rt_kernel_times */
enum {
   MI355RT_K_GENERATE = 0,       /* primary rays: Camera.createImage / RenderThread.run ray set-up */
   MI355RT_K_WALK_PRIMARY = 1,   /* closest hit of the primary rays (Camera.getClosestIntersection) */
   MI355RT_K_SHADE = 2,          /* material colour, mirror bounce generation, shadow list */
   MI355RT_K_WALK_SECONDARY = 3, /* closest hit of mirror-bounce rays */
   MI355RT_K_WALK_SHADOW = 4,    /* RTStatics.shadowIntersects */
   MI355RT_K_PHONG = 5,          /* PhongLightingModel.getPixelColor */
   MI355RT_K_RESOLVE = 6,        /* RenderThread's average over the samples, pixel write */
   MI355RT_K_ANALYTIC = 7,       /* Sphere / Plane intersections folded into the walkers' results (launched only when the scene has any) */
   MI355RT_N_KERNELS = 8
};

typedef struct {
   double ms[MI355RT_N_KERNELS];          /* summed HIP-event time per kernel since the last reset */
   uint64_t launches[MI355RT_N_KERNELS];
   /* of the most recent MI355RT_COLLECT_COUNTS frame, per walker kernel (other entries 0) */
   uint64_t rays[MI355RT_N_KERNELS], box_tests[MI355RT_N_KERNELS], tri_tests[MI355RT_N_KERNELS];
} mi355rt_kernel_times;

/* device-resident frame outputs; any pointer may be NULL.  Pixel p = y*width + x (row-major);
 * Camera.pixels[x][y] is the transpose. */
typedef struct {
   float *rgb;        /* width*height*3, unclamped f32 after the NaN scrub (Camera.java:96-101) */
   uint32_t *argb8;   /* width*height, Color(r,g,b).getRGB() as in Camera.image (Camera.java:103-107) */
   int32_t *hit_id;   /* width*height, face index within its mesh for sample 0 (0 for a sphere or plane), -1 = miss */
   int32_t *hit_obj;  /* width*height, object index for sample 0, -1 = miss */
   float *hit_w;      /* width*height, IntersectionInformation.w for sample 0 */
} mi355rt_frame;

typedef struct mi355rt_scene mi355rt_scene;

/* ---- scene ------------------------------------------------------------------------------- */

/* Replaces `new KDTree(mesh, vertices, normals, indices, colorInfo, computeKDTree)` per mesh
 * (bounds/KDTree.java:24-56, bounds/KDNode.java:29-90) and `new Scene(volumes, ...)`'s bounds
 * (Scene.java:21-33): builds the reference's hierarchy, flattened, in HBM.  The hierarchy of a mesh whose root
 * splits and whose vertices are all finite is built on the GPU, level by level; any other
 * mesh, and every mesh when the switch MI355RT_BUILDER=host is set, by the equivalent host builder.  Both give the same bytes.
 * For a mesh the GPU builder takes, the vertex normals (when the mesh brings none), the leaf groups and the node records are
 * made on the device as well: nothing but the caller's arrays crosses PCIe. */
int mi355rt_scene_create(const mi355rt_mesh_desc *meshes, uint32_t n_meshes, const mi355rt_build_opts *opts,
                         mi355rt_scene **out);
/* The same for any mix of meshes, spheres and planes in Scene.objects order (RayTracer.getWhittedObjects /
 * getCornellBox, RayTracer.java:155-235).  Spheres and planes are intersected analytically
 * (Sphere.java:29-112, Plane.java:87-154) behind their own bounding tests (BoundingSphere.java:29-102,
 * RTStatics.aabbIntersection); planes with colinear corners fail with MI355RT_E_ARG as Plane.java:56-58 throws. */
int mi355rt_scene_create_objects(const mi355rt_object_desc *objects, uint32_t n_objects, const mi355rt_build_opts *opts,
                                 mi355rt_scene **out);
void mi355rt_scene_destroy(mi355rt_scene *scene);
/* Frames in flight.  Camera.createImage() (Camera.java:119-185) renders one frame and waits for it; a caller that
 * renders a sequence (the GUI's re-render on every settings change, a turntable, the tiles of consecutive frames on
 * every GPU of a node) can have frame k + 1 on the chip while frame k ends on its longest rays and leaves the chip
 * half empty.  A view is a second frame context on the same scene: it shares everything a frame only reads (hierarchy,
 * faces, normals, BRDF tables: no second copy in HBM, no second build) and owns what a frame writes (scratch queues,
 * counters, statistics, timers), so mi355rt_render_device calls on different views with different streams overlap on
 * the GPU.  A view is used wherever a scene is (camera_setup, render*, frame_stats, ...) and destroyed with
 * mi355rt_scene_destroy, in any order (the arrays live until the scene and all its views are destroyed).  Set the
 * BRDF tables before creating views.  Calls on ONE handle are still one at a time; creating and destroying views of
 * one scene is not thread safe. */
int mi355rt_scene_view(mi355rt_scene *scene, mi355rt_scene **view_out);

/* Several GPUs behind ONE handle.  The reference's fan-out is inside the one call: Camera.createImage() scatters the pixels
 * over availableProcessors() threads that share the read-only scene (Camera.java:129-182).  Here: give the scene a device
 * list once, and every later mi355rt_render / mi355rt_render_host / mi355rt_render_device on it (and on views created
 * afterwards) spreads the frame over those GPUs -- the scene's arrays are copied to each device (nothing is rebuilt), rank r
 * (the r-th device of the list; one host thread and one stream per device inside the library) renders the tiles t with
 * t % n_devices == r (settings.tile_size squares, 16 by default) into one tile-compact block, the blocks travel to
 * devices[0] in the frame's one exchange step -- RCCL ncclSend / ncclRecv over xGMI (librccl.so.1 is loaded on first use),
 * or peer copies when a device is named twice, librccl is missing or the switch MI355RT_TRANSPORT=peer is set -- and one
 * kernel there writes the frame in image order.  Frames are bit-identical to the one-GPU frame's.
 * devices[0] must be the scene's own device (or -1 for it); a device may appear several times (each entry is a rank with a
 * replica and a stream of its own: how the path is rehearsed on one GPU).  Call it after mi355rt_brdf_tables and before
 * mi355rt_scene_view, once per scene.  settings.tile_rank / tile_world / tile_compact must be unset on such a scene (the
 * library shards the frame itself); mi355rt_frame_stats sums the ranks' counts; mi355rt_kernel_profile, mi355rt_trace_rays
 * and mi355rt_render_device_noalloc use devices[0] alone.  mi355rt_render_device stays asynchronous on hip_stream (a stream
 * of devices[0]): the ranks' streams wait for it, and it waits for the unpacked frame. */
#define MI355RT_MAX_RANKS 16
int mi355rt_scene_set_devices(mi355rt_scene *scene, const int32_t *devices, int32_t n_devices);

enum { MI355RT_TRANSPORT_NONE = 0,   /* one GPU: no exchange step */
       MI355RT_TRANSPORT_PEER = 1,   /* hipMemcpyPeerAsync from each rank's stream */
       MI355RT_TRANSPORT_RCCL = 2 }; /* ncclSend / ncclRecv */
typedef struct {
   int32_t n_ranks;
   int32_t transport;                     /* MI355RT_TRANSPORT_* */
   int32_t devices[MI355RT_MAX_RANKS];
   float render_ms[MI355RT_MAX_RANKS];    /* each rank's kernels of the most recent frame (HIP events on the rank's stream) */
} mi355rt_multi_info;
/* Blocking (waits for the most recent frame's events).  No counterpart in the reference: it is what bench.py reports per rank. */
int mi355rt_scene_multi_info(mi355rt_scene *scene, mi355rt_multi_info *out);

/* Replaces BRDFMaterial.brdfDirections / brdfWeights (materials/BRDFMaterial.java:33,43): takes the
 * PARSED tables (rows of table.txt: theta_v phi_v theta_i phi_i; rows of abrdf.dat) and applies
 * BRDFMaterial.remapDirections (:148-156, PBRTMath.java:74-97). */
int mi355rt_brdf_tables(mi355rt_scene *scene, const float *angles_nx4, int32_t n_dirs, const float *weights,
                        int32_t n_materials);

/* Scene ctor's RTStatics.EPSILON choice (Scene.java:35-36) for this scene's bounds. */
int mi355rt_scene_epsilon(const mi355rt_scene *scene, float *epsilon_out);

/* object i: minmax = {minx,miny,minz,maxx,maxy,maxz} (KDTree.getMinMax), KDTree.shadowDistance,
 * and sizes of the flattened hierarchy. */
int mi355rt_scene_object_info(const mi355rt_scene *scene, uint32_t object, float minmax[6], float *shadow_distance,
                              uint64_t *n_nodes, uint64_t *n_leaves, uint64_t *n_leaf_slots, int32_t *max_depth);
/* Where scene creation spent its time, in seconds (the reference prints its own: TriangleMesh.java:84-85,123-124, KDTree.java:49-55):
 * out[0] validation and bounds on the host (for a mesh the GPU builder does not take: its normals and hierarchy too); GPU builder:
 * out[1] upload + vertex normals (TriangleMesh.computeNormals), out[2] hierarchy (KDNode), out[3] leaf groups, out[4] node records;
 * out[5] host builder: records + leaf groups; out[6] copies into the scene's arrays. */
int mi355rt_scene_build_seconds(const mi355rt_scene *scene, double out[7]);
/* bytes of HBM held by the scene */
int mi355rt_scene_device_bytes(const mi355rt_scene *scene, uint64_t *bytes_out);

/* Replaces the Camera constructor's derived fields (Camera.java:50-51,56-57,59-81,
 * RTStatics.initializeQuat4f RTStatics.java:345-361).  A Java caller may instead fill
 * mi355rt_camera from its own Camera object. */
int mi355rt_camera_setup(const mi355rt_scene *scene, const float orientation_deg[3], float fov_deg, float near_plane,
                         int32_t width, int32_t height, mi355rt_camera *out);

/* ---- render ------------------------------------------------------------------------------ */

/* Replaces Camera.createImage() + RenderThread.run() + Camera.setPixel() (Camera.java:119-185,
 * RenderThread.java:30-69, Camera.java:96-108): blocking; host output buffers (any may be NULL):
 * rgb W*H*3 f32 row-major; hit_id / hit_obj / hit_w W*H for sample 0 (face index within its mesh, index of the
 * object in Scene.objects -- IntersectionInformation.intersectionObject --, IntersectionInformation.w; -1, -1, 0 = miss). */
int mi355rt_render(mi355rt_scene *scene, const mi355rt_camera *cam, const mi355rt_settings *settings, float *rgb_out,
                   int32_t *hit_id_out, int32_t *hit_obj_out, float *hit_w_out);

/* The same call with the 8-bit image among its host outputs: argb8 = Camera.image, W*H Color.getRGB() values written by the
 * frame's own kernels (Camera.setPixel, Camera.java:103-107) -- a third of the bytes of rgb across PCIe, and what a caller
 * that only displays the frame needs.  Any pointer may be NULL; only what is asked for is written on the device and copied
 * (the copies queue behind the frame and the host waits once). */
typedef struct {
   float *rgb;        /* W*H*3 */
   uint32_t *argb8;   /* W*H */
   int32_t *hit_id;   /* W*H */
   int32_t *hit_obj;  /* W*H */
   float *hit_w;      /* W*H */
} mi355rt_host_frame;
int mi355rt_render_host(mi355rt_scene *scene, const mi355rt_camera *cam, const mi355rt_settings *settings, const mi355rt_host_frame *out);

/* Page-locked host memory for the buffers mi355rt_render writes to: its device-to-host copies then run at PCIe
 * speed instead of through the runtime's staging buffers (figures in DESIGN.md section 5).  No counterpart in the
 * reference, which has no device: it is for hosts that keep one output buffer per camera, as Camera.pixels is one
 * array (Camera.java:38) -- a Java shim wraps the pointer once with NewDirectByteBuffer.  Ordinary memory works
 * too, only slower.  Free with mi355rt_host_free, not free(). */
int mi355rt_host_alloc(uint64_t bytes, void **ptr_out);
int mi355rt_host_free(void *ptr);

/* Same frame, outputs left in device memory (pointers in `frame` are device pointers on the
 * scene's device).  Asynchronous on `hip_stream` (a hipStream_t, NULL = the null stream); the
 * caller synchronises.  With tile_world > 1 only this rank's tiles are written.
 * The call BLOCKS the host (it is still ordered on hip_stream) in three cases: the scene holds TWO OR MORE objects with a
 * ReflectionMaterial or RefractionMaterial and a path bounces between them more than three times (the first three bounces
 * are enqueued without asking -- their launches find the bounce queue's size on the device; from the fourth on one 32-byte
 * read-back per bounce tells the host when to stop.  A scene with ONE such object never blocks: a bounce ray skips the object
 * it left, ReflectionMaterial.java:16, so its chains are one bounce long); settings.collect_stats has MI355RT_COLLECT_COUNTS /
 * _DIAG (one read-back of the counters per pass); and a call that needs more frame scratch than the scene holds so far (the
 * first one, or a larger frame) frees and allocates device memory, which synchronises the device.  Frames of a size the scene
 * has rendered before return at once: that is what frames in flight on several streams rely on.
 * Frame scratch: about 104 bytes per sample slot of a pass (204 with a mirror, 348 with refraction), a pass being the
 * whole frame up to 2^27 slots -- a 3840x2160 frame at 16 spp is one pass and 14 GB -- or less when that would take
 * over a quarter of the device memory that is free when the scratch has to grow (MI355RT_PASS_SLOTS_LOG2 forces a
 * size); a frame of more slots runs as several passes.  How a frame is cut into passes never changes its pixels. */
int mi355rt_render_device(mi355rt_scene *scene, const mi355rt_camera *cam, const mi355rt_settings *settings,
                          const mi355rt_frame *frame, void *hip_stream);

/* The same frame as mi355rt_render_device for callers that cannot allocate: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference (it has no device).  The call
 * allocates, creates and frees nothing (no hipMalloc / hipFree / event creation): its frame scratch -- the queues and
 * records mi355rt_render_device keeps in buffers it allocates on demand -- is a static device array of
 * MI355RT_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends.  The frame must fit
 * it as ONE pass: ceil(8x8 pixel groups / 8) * 8 * 64 * multi_samples sample slots at 104 bytes each (204 with a
 * ReflectionMaterial in the scene, 348 with a RefractionMaterial): 1,288,000 slots for a mesh-only scene, e.g.
 * 1280x960 at 1 spp or 640x480 at 4 spp; anything larger fails with MI355RT_E_LIMIT and changes nothing.
 * settings.collect_stats must be 0.  Inputs and outputs are as for mi355rt_render_device (device pointers in `frame`,
 * asynchronous on hip_stream except for bounce chains longer than three, as there; NOT capturable into a HIP graph when
 * the scene holds two or more mirror / refraction objects, whose bounce loop may synchronise the stream).
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two streams, two threads, two
 * scenes) race; calls in sequence on one stream are safe, whatever the previous call left behind. */
#define MI355RT_NOALLOC_SCRATCH_BYTES (128u << 20)
int mi355rt_render_device_noalloc(mi355rt_scene *scene, const mi355rt_camera *cam, const mi355rt_settings *settings,
                                  const mi355rt_frame *frame, void *hip_stream);

/* Multi-GPU frames (the reference's own fan-out is Camera.java:129-182, pixels scattered over threads).
 * Each rank renders its tiles with settings.tile_compact into ONE block of 6 n words, n = slots_per_rank * tile_size^2,
 * laid out [rgb: 3 n][hit_id: n][hit_obj: n][hit_w: n] (mi355rt_frame pointers at those offsets; slots_per_rank =
 * ceil(number of tiles / world), the same on every rank); the blocks are gathered to one rank (RCCL); this call, on that
 * rank, scatters `world` consecutive blocks at blocks_dev back into image order.  Device pointers on the current
 * device, asynchronous on hip_stream; any pointer of `frame` may be NULL (argb8 is packed from the blocks' rgb). */
int mi355rt_tiles_unpack_device(const float *blocks_dev, int32_t world, int32_t slots_per_rank, int32_t tile_size, int32_t width,
                                int32_t height, const mi355rt_frame *frame, void *hip_stream);

/* Counters of the most recent render on this scene that ran with settings.collect_stats != 0
 * (blocking: synchronises the scene's device). */
int mi355rt_frame_stats(mi355rt_scene *scene, mi355rt_stats *out);

/* Per-kernel HIP-event times of the renders that ran with MI355RT_COLLECT_TIMES since the last reset,
 * and the per-kernel counts of the most recent MI355RT_COLLECT_COUNTS render (blocking: waits for the
 * pending events).  reset != 0 clears the times afterwards.  No counterpart in the reference: it is
 * what bench.py's roofline line is computed from. */
int mi355rt_kernel_profile(mi355rt_scene *scene, mi355rt_kernel_times *out, int reset);

/* Camera.getClosestIntersection(null, origin, direction, null, 0) for a batch of rays
 * (Camera.java:193-215): host buffers, n*3 origins / directions in; any output may be NULL. */
int mi355rt_trace_rays(mi355rt_scene *scene, const mi355rt_settings *settings, uint64_t n, const float *origins,
                       const float *directions, int32_t *hit_id, int32_t *hit_obj, float *hit_w, float *hit_point,
                       float *hit_normal);

/* ---- 8-bit images (next row: SURVEY.md 8f rank 3) ----------------------------------------- */

/* Replaces the image half of Camera.setPixel (Camera.java:103-107; normalized == 0) and
 * Camera.updateNormalizedImage (Camera.java:217-263; normalized != 0) for a finished f32 frame: rgb is
 * row-major width*height*3 (what mi355rt_render returns), argb_out gets width*height Color.getRGB() values.
 * Host buffers; runs on `device` (-1: current). */
int mi355rt_image_argb8(const float *rgb, int32_t width, int32_t height, int normalized, uint32_t *argb_out, int32_t device);
/* The same on device buffers of the current device, asynchronous on hip_stream; minmax_scratch = 2 device words. */
int mi355rt_image_argb8_device(const float *rgb_dev, int32_t width, int32_t height, int normalized, uint32_t *argb_dev,
                               uint32_t *minmax_scratch, void *hip_stream);

/* ---- loader (next row: SURVEY.md 8f rank 1) ---------------------------------------------- */

/* Replaces the TriangleMesh constructor's parse (sceneObjects/TriangleMesh.java:23-94,128-197).
 * Arrays are malloc'ed by the library; release each with mi355rt_free. */
int mi355rt_ply_load(const char *path, uint32_t *n_vertices, uint32_t *n_faces, float **vertices, float **normals,
                     int32_t **indices, int32_t *has_normals);
/* TriangleMesh.computeNormals (sceneObjects/TriangleMesh.java:96-126) into normals_out (n_vertices*3). */
int mi355rt_compute_normals(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces,
                            float *normals_out);
void mi355rt_free(void *p);

/* ---- misc -------------------------------------------------------------------------------- */

/* Diagnostic: evaluates on the device the arithmetic primitives bit-exact parity rests on, per element:
 * a/b (f32), sqrt(a) (f32), (float)(1.0/sqrt((double)a)) (Vector3f.normalize's scale) and
 * (float)pow((double)b, 20.0) (the Phong exponent).  No counterpart in the reference. */
int mi355rt_debug_arith(uint64_t n, const float *a, const float *b, float *out_div, float *out_sqrt, float *out_rnorm,
                        float *out_pow);

/* Diagnostic: the 64 raw 64-bit counters of the most recent collect_stats render: [0..5] = mi355rt_stats,
 * [6..15] = lane-trips / wave-trips of the walker's node and leaf loops, outer trips, refills, steals, wave
 * lifetimes; [16..24] per-walker rays / box tests / triangle tests; [25..42] per-walker timeline, [43..58] rays by trips taken
 * (trace_kernels.hip ST_*).  No counterpart in the reference. */
int mi355rt_debug_counters(mi355rt_scene *scene, uint64_t out[64]);

/* Diagnostic: sizes of the scene's device arrays -- out[0] node records, [1] leaf descriptors,
 * [2] group faces, [3] leaf slots (the reference's duplicated leaf entries), [4] leaf groups, [5] groups shared by
 * several leaves, [6] 1 if the walker takes its form for leaves of tens of faces, [7] bytes resident on the device.
 * No counterpart in the reference. */
int mi355rt_debug_scene_layout(mi355rt_scene *scene, uint64_t out[8]);

/* Diagnostic: FNV-1a digests of the scene's device arrays as uploaded -- out[0] node records, out[1] leaf data
 * (descriptors, group faces, slot ids), out[2] face normals -- and out[3] = 1 if any mesh's hierarchy was built on the GPU.  The tests use it to
 * show that the GPU builder and the host builder (environment MI355RT_BUILDER=host) produce the same bytes.
 * No counterpart in the reference. */
int mi355rt_debug_scene_digest(mi355rt_scene *scene, uint64_t out[4]);

/* Diagnostic / tests: the library's tuning switches.  They are read from the MI355RT_* environment variables ONCE, when the
 * library first needs one (never on the frame path, and never again: a host that calls setenv later changes nothing);
 * this call sets one afterwards.  name: with or without the MI355RT_ prefix (BUILDER, FUSE_SHADOW, PASS_SLOTS_LOG2,
 * BIG_LEAVES, BLOCKS_PER_CU[_SHADOW], TRIM, TRIM_PRIMARY_X2, TRIM_SHADOW_X2, SPREAD_SL, SPREAD_BL, SPREAD_PRIMARY_MAX,
 * TRANSPORT, BOUNCE_SYNC, GPU_GROUP, BUILD_TIMING); value NULL = the built-in default; name NULL = every switch back to what
 * the environment said.  Not for use while another thread renders.  No counterpart in the reference. */
int mi355rt_debug_switch(const char *name, const char *value);

const char *mi355rt_last_error(void);
int mi355rt_abi_version(void);
int mi355rt_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
