/*
 * rt_oracle.h -- CPU restatement of the sranger/Ray-Tracer per-pixel trace loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 *  and there is no JVM in this pipeline, so this restatement is
 * pinned only by hand-derived known answers (tests/test_oracle_known_answers.py)
 * and by a second, independent numpy-f32 restatement of the core arithmetic
 * (tests/np_restatement.py).
 *
 * All file:line citations are relative to /root/reference/src/stephen/ranger/ar/.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* material kinds: materials/ColorInformation.java, BRDFMaterial.java, ReflectionMaterial.java */
enum { ORC_MAT_COLOR = 0, ORC_MAT_BRDF = 1, ORC_MAT_REFLECTION = 2,
       ORC_MAT_REFRACTION = 3,    /* materials/RefractionMaterial.java */
       ORC_MAT_CHECKERBOARD = 4 };/* materials/CheckerboardMaterial.java */
/* object kinds: bounds/KDTree.java over a TriangleMesh (0), sceneObjects/Sphere.java inside a
 * bounds/BoundingSphere.java (1), sceneObjects/Plane.java inside a bounds/AxisAlignedBoundingBox.java (2) */
enum { ORC_OBJ_MESH = 0, ORC_OBJ_SPHERE = 1, ORC_OBJ_PLANE = 2 };
/* guard (no counterpart in the reference, like max_mirror_bounces): a path carries at most this many pending
 * RefractionMaterial absorption terms; a further RefractionMaterial hit answers with its diffuse colour */
#define ORC_MAX_REFRACTION_TERMS 8
/* lighting models: lighting/LightingModel.java (0), PhongLightingModel.java (1) */
enum { ORC_LIGHTING_PLAIN = 0, ORC_LIGHTING_PHONG = 1 };

typedef struct {
   uint32_t kind;
   float emission[3], ambient[3], diffuse[3], specular[3];
   float shininess;
   int32_t brdf_id;
} orc_material;

typedef struct {
   const float *vertices;   /* n_vertices * 3 */
   const float *normals;    /* n_vertices * 3, or NULL => TriangleMesh.computeNormals */
   const int32_t *indices;  /* n_faces * 3 */
   uint32_t n_vertices, n_faces;
   orc_material material;
   int32_t compute_kdtree;  /* TriangleMesh's own computeKDTree (TriangleMesh.java:89): 0 = opts->use_kdtree decides, > 0 yes, < 0 no */
} orc_mesh_desc;

typedef struct {
   float refraction_index;                      /* RefractionMaterial.refractionIndex (RefractionMaterial.java:21) */
   float checker_color1[3], checker_color2[3];  /* CheckerboardMaterial.color1 / color2 */
   float checker_cell[3];                       /* cellWidth, cellHeight, cellDepth */
} orc_material_ext;

typedef struct {
   uint32_t kind;            /* ORC_OBJ_* */
   orc_mesh_desc mesh;       /* MESH: the arrays; every kind: mesh.material is the object's ColorInformation */
   orc_material_ext ext;
   float center[3], radius;  /* SPHERE: Sphere(radius, origin, colorInfo) Sphere.java:15-22 */
   float corners[12];        /* PLANE: Plane(corners, colorInfo) Plane.java:28-76 */
   uint32_t n_corners;       /* 3 or 4 */
} orc_object_desc;

typedef struct {
   int32_t max_depth;     /* RTStatics.MAX_DEPTH    (RTStatics.java:25) */
   int32_t max_children;  /* RTStatics.MAX_CHILDREN (RTStatics.java:24) */
   int32_t use_kdtree;    /* computeKDTree flag     (KDNode.java:40)    */
   int32_t device;        /* unused by the oracle; keeps the layout equal to the product's */
} orc_build_opts;

typedef struct {
   float origin[3];
   float rotation[9];     /* row-major m00 m01 m02 m10 ... */
   float viewport_w, viewport_h, near_plane;
   int32_t width, height;
} orc_camera;

typedef struct {
   float epsilon;               /* RTStatics.EPSILON at render time (Scene.java:36) */
   float near_plane, far_plane; /* RTStatics.NEAR_PLANE / FAR_PLANE (RTStatics.java:19-20) */
   int32_t max_recursion_depth; /* RTStatics.MAX_RECURSION_DEPTH (RTStatics.java:21) */
   int32_t multi_samples;       /* Camera.multiSamples */
   int32_t brdf_samples;        /* Camera.brdfSamples  */
   int32_t lighting_model;
   int32_t max_mirror_bounces;  /* guard for KDTree.getColor's missing cut-off (KDTree.java:77-80) */
   uint64_t rng_seed;
   float light_origin[3], light_emission[3], light_ambient[3];
   int32_t tile_size, tile_rank, tile_world; /* image-tile sharding (ignored by the oracle) */
   int32_t collect_stats;
} orc_settings;

typedef struct {
   uint64_t primary_rays, shadow_rays, secondary_rays;
   uint64_t box_tests, tri_tests, hits;
} orc_stats;

typedef struct orc_scene orc_scene;

int orc_scene_create(const orc_mesh_desc *meshes, uint32_t n_meshes, const orc_build_opts *opts, orc_scene **out);
/* any mix of meshes, spheres and planes, in the order of Scene.objects */
int orc_scene_create_objects(const orc_object_desc *objects, uint32_t n_objects, const orc_build_opts *opts, orc_scene **out);
void orc_scene_destroy(orc_scene *s);
int orc_scene_set_brdf(orc_scene *s, const float *angles_nx4, int32_t n_dirs, const float *weights, int32_t n_materials);
/* object i: 6 floats min xyz, max xyz; shadowDistance; node / leaf / leaf-slot counts, depth */
int orc_scene_object_info(const orc_scene *s, uint32_t obj, float minmax[6], float *shadow_distance,
                          uint64_t *n_nodes, uint64_t *n_leaves, uint64_t *n_leaf_slots, int32_t *max_depth);
/* the synthesised / copied per-vertex normals of object i (n_vertices*3) */
const float *orc_scene_object_normals(const orc_scene *s, uint32_t obj);
float orc_scene_epsilon(const orc_scene *s);
void orc_camera_setup(const orc_scene *s, const float orientation[3], float fov, float near_plane,
                      int32_t width, int32_t height, orc_camera *out);

/* full frame; rgb is row-major [y][x][3]; hit_* describe sample 0 of each pixel; any output may be NULL.
 * pixel_stride k > 1 renders only pixels whose linear index (y*W+x) is a multiple of k. */
int orc_render(const orc_scene *s, const orc_camera *cam, const orc_settings *st, int32_t n_threads,
               int32_t pixel_stride, float *rgb, int32_t *hit_id, int32_t *hit_obj, float *hit_w,
               orc_stats *stats);

/* closest hit of arbitrary rays: Camera.getClosestIntersection(null, o, d, null, 0) */
int orc_trace_rays(const orc_scene *s, const orc_settings *st, uint64_t n, const float *origins,
                   const float *dirs, int32_t *hit_id, int32_t *hit_obj, float *hit_w,
                   float *hit_point, float *hit_normal);

/* the same, followed by lightingModel.getPixelColor(info, 0) of each hit (all outputs required); see tools/java/GoldenDumper.java */
int orc_shade_rays(const orc_scene *s, const orc_settings *st, uint64_t n, const float *origins, const float *dirs,
                   int32_t *hit_obj, float *hit_w, float *hit_point, float *hit_normal, float *rgb);

/* unit pieces for known-answer tests */
int orc_aabb_intersection(const float o[3], const float d[3], const float minmax[6], float near_plane, float far_plane);
int orc_intersects_triangle(const float o[3], const float d[3], const float p[9], const float n[9], float eps, float out[7]);
void orc_reflection_direction(const float n[3], const float dir[3], float out[3]);
/* Sphere.getIntersection Sphere.java:29-112 (bounding != 0: BoundingSphere.intersects BoundingSphere.java:29-102, out untouched);
 * out = point xyz, normal xyz, w; returns 1 on a hit */
int orc_sphere_intersection(const float o[3], const float d[3], const float center[3], float radius, float eps, int bounding, float out[7]);
/* Plane ctor + Plane.getIntersection Plane.java:28-154; out as above; returns 1 on a hit, -1 for colinear corners */
int orc_plane_intersection(const float o[3], const float d[3], const float *corners, uint32_t n_corners, float eps, float out[7]);
/* RefractionMaterial.getRefractionDirection RefractionMaterial.java:61-95 */
void orc_refraction_direction(const float n[3], const float dir[3], float refraction_index, float out[3]);
/* CheckerboardMaterial.getMaterialColor CheckerboardMaterial.java:21-49 */
void orc_checker_color(const float minmax[6], const float point[3], const orc_material_ext *ext, float out[3]);
void orc_normalize(float v[3]);
void orc_image_argb8(const float *rgb, int32_t width, int32_t height, int normalized, uint32_t *out);
void orc_rgb_to_hsv(const float rgb[3], float hsv[3]);
void orc_hsv_to_rgb(const float hsv[3], float rgb[3]);
void orc_remap_angles(const float in2[2], const float out2[2], float out3[3]);
void orc_remap_vectors(const float normal[3], const float in_dir[3], const float out_dir[3], float out3[3]);
float orc_brdf_luminocity(const orc_scene *s, int32_t brdf_id, int32_t brdf_samples, const float ray_dir[3],
                          const float point[3], const float normal[3], const float light[3]);
void orc_phong(const float color_in[3], const float kd[3], const float ks[3], float shininess,
               const float light_emission[3], const float light_ambient[3], const float light_origin[3],
               const float point[3], const float normal[3], const float ray_dir[3], int32_t shadowed,
               float out[3]);
void orc_quat_rotation(const float orientation[3], float rot9[9]);
/* per-pixel sample jitter stream: java.util.Random restated, seeded per pixel */
uint64_t orc_rng_init(uint64_t seed, uint64_t pixel_index);
float orc_rng_next_float(uint64_t *state);
void orc_compute_normals(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces, float *normals_out);

#ifdef __cplusplus
}
#endif
#endif
