// mi355rt.hpp -- C++ host-side mirror of the reference's classes for the trace path, over the C ABI of
// include/mi355rt.h (header only; link with -lmi355rt).
//
// The reference's host language is Java and no JDK exists in this pipeline, so this is the compiled-language
// host side: the same class names, constructor arguments and call sequence as the reference
// (stephen.ranger.ar.*), so that a caller reads like RayTracer.java / RayTracerInterface.java:
//
//    Light light({0, 100, 100}, {0.3f, 0.3f, 0.3f}, {0.5f, 0.5f, 0.9f});                 // RayTracer.java:106
//    TriangleMesh mesh("dragon.ply", ColorInformation({0.9f, 0.9f, 0.9f}), true);        // RayTracer.java:122-123
//    PhongLightingModel model(light, {&mesh});
//    Scene scene({&mesh}, light, {0, 0, 0}, model, 23.0f);                                // Scene.java:14
//    Camera camera(scene, 4, 1, RTStatics::NEAR_PLANE, 1920, 1080);                       // Camera.java:40
//    camera.addActionListener([](int id, const char *what) { ... });
//    camera.createImage();                                                                // Camera.java:119
//    const float *rgb = camera.getPixel(x, y);                                            // Camera.pixels[x][y]
//
// Error behaviour: the reference prints a stack trace and exits (TriangleMesh.java:90-93); here every failed
// ABI call throws mi355rt::Error carrying the status code and mi355rt_last_error().
#ifndef MI355RT_HPP
#define MI355RT_HPP

#include <array>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mi355rt.h"

namespace mi355rt {

struct Error : std::runtime_error {
   int code;
   Error(int c, const std::string &what) : std::runtime_error("mi355rt error " + std::to_string(c) + ": " + what), code(c) {}
};

inline void check(int rc)
{
   if (rc != MI355RT_OK) throw Error(rc, mi355rt_last_error());
}

using Color3 = std::array<float, 3>;

// RTStatics.java:19-25 and the EPSILON the Scene constructor sets (Scene.java:35-36)
struct RTStatics {
   static constexpr float NEAR_PLANE = 0.01f;
   static constexpr float FAR_PLANE = 3000.0f;
   static constexpr int MAX_RECURSION_DEPTH = 5;
   static constexpr int MAX_DEPTH = 20;
   static constexpr int MAX_CHILDREN = 5;
   static constexpr int MAX_MIRROR_BOUNCES = 64;   // no counterpart: bounds/KDTree.java:77-80 has no cut-off
   static float &EPSILON()
   {
      static float value = 1e-15f;
      return value;
   }
};

// lighting/Light.java:7-16
struct Light {
   Color3 origin, emission, ambient;
   Light(Color3 o = { 100.f, 100.f, 100.f }, Color3 e = { 1.f, 1.f, 1.f }, Color3 a = { 1.f, 1.f, 1.f }) : origin(o), emission(e), ambient(a) {}
};

// materials/ColorInformation.java:15-37
class ColorInformation {
 public:
   Color3 emission { 0.f, 0.f, 0.f }, ambient, diffuse, specular;
   float shininess;
   explicit ColorInformation(Color3 d = { 1.f, 1.f, 1.f }, float shine = 20.0f) : ambient(d), diffuse(d), specular(d), shininess(shine) {}
   ColorInformation(Color3 e, Color3 a, Color3 d, Color3 s, float shine) : emission(e), ambient(a), diffuse(d), specular(s), shininess(shine) {}
   virtual ~ColorInformation() = default;
   virtual uint32_t kind() const { return MI355RT_MAT_COLOR; }
   virtual int32_t brdfId() const { return 0; }
   virtual mi355rt_material_ext abiExt() const { return mi355rt_material_ext {}; }
   mi355rt_material abi() const
   {
      mi355rt_material m {};
      m.kind = kind();
      for (int c = 0; c < 3; c++) {
         m.emission[c] = emission[c];
         m.ambient[c] = ambient[c];
         m.diffuse[c] = diffuse[c];
         m.specular[c] = specular[c];
      }
      m.shininess = shininess;
      m.brdf_id = brdfId();
      return m;
   }
};

// materials/BRDFMaterial.java:33-167: the parsed CUReT tables are handed over once (the Java statics)
class BRDFMaterial : public ColorInformation {
 public:
   int materialID;
   BRDFMaterial(int id, Color3 color) : ColorInformation(color), materialID(id) {}
   uint32_t kind() const override { return MI355RT_MAT_BRDF; }
   int32_t brdfId() const override { return materialID; }
   struct Tables {
      std::vector<float> angles;    // n_dirs x 4: theta_v phi_v theta_i phi_i (rows of table.txt)
      std::vector<float> weights;   // n_materials x n_dirs (rows of abrdf.dat)
      int n_dirs = 0, n_materials = 0;
   };
   static Tables &tables()
   {
      static Tables t;
      return t;
   }
   static void setTables(std::vector<float> angles, int n_dirs, std::vector<float> weights, int n_materials)
   {
      if ((int) angles.size() != n_dirs * 4 || (int) weights.size() != n_materials * n_dirs) throw std::invalid_argument("need angles[n,4] and weights[m,n]");
      tables() = Tables { std::move(angles), std::move(weights), n_dirs, n_materials };
   }
};

// materials/ReflectionMaterial.java:10-24
class ReflectionMaterial : public ColorInformation {
 public:
   explicit ReflectionMaterial(Color3 d) : ColorInformation(d, 100.0f) {}
   uint32_t kind() const override { return MI355RT_MAT_REFLECTION; }
};

// materials/RefractionMaterial.java:11-59
class RefractionMaterial : public ColorInformation {
 public:
   static constexpr float INDEX_OF_AIR = 1.00029f, INDEX_OF_GLASS = 1.52f, INDEX_OF_ICE = 1.31f, INDEX_OF_WATER = 1.33f,
                          INDEX_OF_QUARTZ = 1.46f, INDEX_OF_SAPPHIRE = 1.77f, INDEX_OF_DIAMOND = 2.417f;
   float refractionIndex;
   RefractionMaterial(Color3 d, float index) : ColorInformation(d, 100.0f), refractionIndex(index) {}
   uint32_t kind() const override { return MI355RT_MAT_REFRACTION; }
   mi355rt_material_ext abiExt() const override
   {
      mi355rt_material_ext e {};
      e.refraction_index = refractionIndex;
      return e;
   }
};

// materials/CheckerboardMaterial.java:8-49: ColorInformation() (black, shininess 20) with two cell colours
class CheckerboardMaterial : public ColorInformation {
 public:
   Color3 color1, color2;
   float cellWidth, cellHeight, cellDepth;
   CheckerboardMaterial(Color3 c1, Color3 c2, float w, float h, float d)
      : ColorInformation(Color3 { 0.f, 0.f, 0.f }), color1(c1), color2(c2), cellWidth(w), cellHeight(h), cellDepth(d)
   {
   }
   uint32_t kind() const override { return MI355RT_MAT_CHECKERBOARD; }
   mi355rt_material_ext abiExt() const override
   {
      mi355rt_material_ext e {};
      for (int c = 0; c < 3; c++) {
         e.checker_color1[c] = color1[c];
         e.checker_color2[c] = color2[c];
      }
      e.checker_cell[0] = cellWidth;
      e.checker_cell[1] = cellHeight;
      e.checker_cell[2] = cellDepth;
      return e;
   }
};

// bounds/BoundingVolume.java: what Scene.objects holds.  Here it is the description handed to the library, which
// builds the KDTree / BoundingSphere / AxisAlignedBoundingBox behind it.
class BoundingVolume {
 public:
   mi355rt_material material;
   mi355rt_material_ext materialExt;
   bool computeKDTree = true;
   virtual ~BoundingVolume() = default;
   virtual void fill(mi355rt_object_desc &d) const = 0;
   const BoundingVolume *getBoundingVolume() const { return this; }

 protected:
   explicit BoundingVolume(const ColorInformation &colorInfo) : material(colorInfo.abi()), materialExt(colorInfo.abiExt()) {}
};

// sceneObjects/Sphere.java:15-22
class Sphere : public BoundingVolume {
 public:
   float radius;
   Color3 origin;
   Sphere(float r, Color3 o, const ColorInformation &colorInfo) : BoundingVolume(colorInfo), radius(r), origin(o) {}
   void fill(mi355rt_object_desc &d) const override
   {
      d.kind = MI355RT_OBJ_SPHERE;
      for (int c = 0; c < 3; c++) d.center[c] = origin[c];
      d.radius = radius;
   }
};

// sceneObjects/Plane.java:28-76: three or four corners
class Plane : public BoundingVolume {
 public:
   std::vector<Color3> corners;
   Plane(std::vector<Color3> c, const ColorInformation &colorInfo) : BoundingVolume(colorInfo), corners(std::move(c))
   {
      if (corners.size() < 3) throw std::invalid_argument("The array of corner values must contain three or more vertices");   // Plane.java:33-35
      if (corners.size() > 4) corners.resize(4);
   }
   void fill(mi355rt_object_desc &d) const override
   {
      d.kind = MI355RT_OBJ_PLANE;
      d.n_corners = (uint32_t) corners.size();
      for (size_t i = 0; i < corners.size(); i++)
         for (int c = 0; c < 3; c++) d.corners[3 * i + c] = corners[i][c];
   }
};

// sceneObjects/TriangleMesh.java:23-94: a PLY path (parsed with the reference loader's rules) or arrays
class TriangleMesh : public BoundingVolume {
 public:
   std::vector<float> vertices, normals;   // normals empty: synthesised as TriangleMesh.computeNormals
   std::vector<int32_t> indices;

   TriangleMesh(const std::string &modelLocation, const ColorInformation &colorInfo, bool kdtree = true) : BoundingVolume(colorInfo)
   {
      computeKDTree = kdtree;
      uint32_t nv = 0, nf = 0;
      float *v = nullptr, *n = nullptr;
      int32_t *i = nullptr, has_normals = 0;
      check(mi355rt_ply_load(modelLocation.c_str(), &nv, &nf, &v, &n, &i, &has_normals));
      vertices.assign(v, v + 3 * (size_t) nv);
      if (has_normals) normals.assign(n, n + 3 * (size_t) nv);
      indices.assign(i, i + 3 * (size_t) nf);
      mi355rt_free(v);
      mi355rt_free(n);
      mi355rt_free(i);
   }
   TriangleMesh(std::vector<float> v, std::vector<int32_t> i, const ColorInformation &colorInfo, bool kdtree = true,
                std::vector<float> n = {})
      : BoundingVolume(colorInfo), vertices(std::move(v)), normals(std::move(n)), indices(std::move(i))
   {
      computeKDTree = kdtree;
   }
   void fill(mi355rt_object_desc &d) const override
   {
      d.kind = MI355RT_OBJ_MESH;
      d.mesh.vertices = vertices.data();
      d.mesh.normals = normals.empty() ? nullptr : normals.data();
      d.mesh.indices = indices.data();
      d.mesh.n_vertices = (uint32_t) (vertices.size() / 3);
      d.mesh.n_faces = (uint32_t) (indices.size() / 3);
   }
};

// lighting/LightingModel.java:13-15, lighting/PhongLightingModel.java:18-54
class LightingModel {
 public:
   virtual ~LightingModel() = default;
   virtual int kind() const { return MI355RT_LIGHTING_PLAIN; }
};
class PhongLightingModel : public LightingModel {
 public:
   PhongLightingModel(const Light &, const std::vector<const BoundingVolume *> &) {}
   int kind() const override { return MI355RT_LIGHTING_PHONG; }
};

// Scene.java:14-38
class Scene {
 public:
   std::vector<const BoundingVolume *> objects;
   Light light;
   std::array<float, 3> cameraOrientation;
   int lightingModel;
   float fov;

   Scene(std::vector<const BoundingVolume *> objs, const Light &l, std::array<float, 3> orientation, const LightingModel &model, float fieldOfView,
         int device = -1)
      : objects(std::move(objs)), light(l), cameraOrientation(orientation), lightingModel(model.kind()), fov(fieldOfView)
   {
      std::vector<mi355rt_object_desc> descs(objects.size(), mi355rt_object_desc {});
      bool brdf = false;
      for (size_t k = 0; k < objects.size(); k++) {   // TriangleMesh, Sphere or Plane, in Scene.objects order
         const BoundingVolume &m = *objects[k];
         m.fill(descs[k]);
         descs[k].mesh.material = m.material;
         descs[k].ext = m.materialExt;
         descs[k].mesh.compute_kdtree = m.computeKDTree ? 1 : -1;   // per mesh, as TriangleMesh.java:89 passes it on
         brdf = brdf || m.material.kind == MI355RT_MAT_BRDF;
      }
      mi355rt_build_opts opts { RTStatics::MAX_DEPTH, RTStatics::MAX_CHILDREN, 1, device };
      check(mi355rt_scene_create_objects(descs.data(), (uint32_t) descs.size(), &opts, &handle_));
      if (brdf) {
         const BRDFMaterial::Tables &t = BRDFMaterial::tables();
         if (t.n_dirs == 0) throw std::invalid_argument("BRDFMaterial::setTables must be called before building a BRDF scene");
         check(mi355rt_brdf_tables(handle_, t.angles.data(), t.n_dirs, t.weights.data(), t.n_materials));
      }
      check(mi355rt_scene_epsilon(handle_, &RTStatics::EPSILON()));   // the global the reference's Scene ctor sets
   }
   Scene(const Scene &) = delete;
   Scene &operator=(const Scene &) = delete;
   ~Scene() { mi355rt_scene_destroy(handle_); }
   mi355rt_scene *handle() const { return handle_; }

   // mi355rt_scene_set_devices: every later frame of this scene (and of its views) is spread over these GPUs inside the one
   // render call, as Camera.createImage spreads pixels over availableProcessors() threads (Camera.java:129-182).
   // devices[0] must be the device the scene was created on; naming a device twice is allowed (two ranks on it).
   void setDevices(const std::vector<int32_t> &devices) { check(mi355rt_scene_set_devices(handle_, devices.data(), (uint32_t) devices.size())); }
   mi355rt_multi_info multiInfo() const
   {
      mi355rt_multi_info info {};
      check(mi355rt_scene_multi_info(handle_, &info));
      return info;
   }
   // seconds of the phases of scene creation (what TriangleMesh.java:84-85 and KDTree.java:49-55 print)
   std::array<double, 7> buildSeconds() const
   {
      std::array<double, 7> t {};
      check(mi355rt_scene_build_seconds(handle_, t.data()));
      return t;
   }

 private:
   mi355rt_scene *handle_ = nullptr;
};

// mi355rt_scene_view: a second frame context on a scene (shared arrays in HBM, own scratch), for frames in flight.
// Its handle is used wherever a scene's is.
class SceneView {
 public:
   explicit SceneView(const Scene &scene) { check(mi355rt_scene_view(scene.handle(), &handle_)); }
   SceneView(const SceneView &) = delete;
   SceneView &operator=(const SceneView &) = delete;
   ~SceneView() { mi355rt_scene_destroy(handle_); }
   mi355rt_scene *handle() const { return handle_; }

 private:
   mi355rt_scene *handle_ = nullptr;
};

// Camera.java:25-215
class Camera {
 public:
   using ActionListener = std::function<void(int id, const char *command)>;
   const Scene &scene;
   int multiSamples, brdfSamples, screenWidth, screenHeight;
   float nearPlaneDistance;
   uint64_t rngSeed = 0;   // the reference's jitter RNG is unseeded (RenderThread.java:32)
   mi355rt_camera cam {};  // origin, rotation, viewportWidth / viewportHeight as the reference's constructor derives them
   std::vector<int32_t> hitId;
   std::vector<int32_t> hitObj;   // index into Scene.objects, -1 = miss
   std::vector<float> hitW;
   std::vector<uint32_t> image;   // Camera.image (Camera.java:103-107): 0xFFrrggbb, row-major

   Camera(const Scene &s, int ms, int brdf, float nearPlane, int width, int height)
      : scene(s), multiSamples(ms), brdfSamples(brdf), screenWidth(width), screenHeight(height), nearPlaneDistance(nearPlane)
   {
      check(mi355rt_camera_setup(scene.handle(), scene.cameraOrientation.data(), scene.fov, nearPlane, width, height, &cam));
   }
   void addActionListener(ActionListener l) { listeners_.push_back(std::move(l)); }

   mi355rt_settings settings() const
   {
      mi355rt_settings st {};
      st.epsilon = RTStatics::EPSILON();
      st.near_plane = RTStatics::NEAR_PLANE;
      st.far_plane = RTStatics::FAR_PLANE;
      st.max_recursion_depth = RTStatics::MAX_RECURSION_DEPTH;
      st.multi_samples = multiSamples;
      st.brdf_samples = brdfSamples;
      st.lighting_model = scene.lightingModel;
      st.max_mirror_bounces = RTStatics::MAX_MIRROR_BOUNCES;
      st.rng_seed = rngSeed;
      for (int c = 0; c < 3; c++) {
         st.light_origin[c] = scene.light.origin[c];
         st.light_emission[c] = scene.light.emission[c];
         st.light_ambient[c] = scene.light.ambient[c];
      }
      st.tile_world = 1;
      return st;
   }

   // Camera.createImage (Camera.java:119-185); blocks until the frame is done, then fires "finished" (id 1)
   void createImage()
   {
      const size_t n = (size_t) screenWidth * (size_t) screenHeight;
      rows_.assign(3 * n, 0.0f);
      hitId.assign(n, -1);
      hitObj.assign(n, -1);
      hitW.assign(n, 0.0f);
      image.assign(n, 0u);
      const mi355rt_settings st = settings();
      mi355rt_host_frame out {};
      out.rgb = rows_.data();
      out.argb8 = image.data();
      out.hit_id = hitId.data();
      out.hit_obj = hitObj.data();
      out.hit_w = hitW.data();
      check(mi355rt_render_host(scene.handle(), &cam, &st, &out));
      std::vector<ActionListener> fire;
      fire.swap(listeners_);
      for (ActionListener &l : fire) l(1, "finished");
   }
   // Camera.pixels[x][y]: unclamped f32 RGB after the NaN scrub (Camera.java:96-101)
   const float *getPixel(int x, int y) const { return rows_.data() + 3 * ((size_t) y * (size_t) screenWidth + (size_t) x); }
   const std::vector<float> &rowMajorPixels() const { return rows_; }

   // Camera.getClosestIntersection(null, origin, direction, null, 0) for arrays of rays (Camera.java:193-215)
   void getClosestIntersection(const std::vector<float> &origins, const std::vector<float> &directions, std::vector<int32_t> &face,
                               std::vector<int32_t> &object, std::vector<float> &w, std::vector<float> &point, std::vector<float> &normal) const
   {
      const size_t n = origins.size() / 3;
      face.assign(n, -1);
      object.assign(n, -1);
      w.assign(n, 0.0f);
      point.assign(3 * n, 0.0f);
      normal.assign(3 * n, 0.0f);
      const mi355rt_settings st = settings();
      check(mi355rt_trace_rays(scene.handle(), &st, n, origins.data(), directions.data(), face.data(), object.data(), w.data(), point.data(),
                               normal.data()));
   }

 private:
   std::vector<float> rows_;
   std::vector<ActionListener> listeners_;
};

}   // namespace mi355rt
#endif
