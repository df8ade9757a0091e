// analytic.h -- device forms of the reference's analytic scene objects and of the two materials that come with
// them.  Every comparison and every float / double step is the reference's own, so NaNs
// and ties fall where they fall there.  Paths are relative to /root/reference/src/stephen/ranger/ar/.
#pragma once
#include "device_util.h"

namespace mi355rt {

namespace {

// RTStatics.leastPositive RTStatics.java:82-100
__device__ __forceinline__ float least_positive(float i, float j)
{
   if (i < 0 && j < 0) return -1.0f;
   if (i < 0 && j > 0) return j;
   if (i > 0 && j < 0) return i;
   return (i < j) ? i : j;
}

// The quadratic shared by Sphere.getIntersection (sceneObjects/Sphere.java:38-99) and BoundingSphere.intersects
// (bounds/BoundingSphere.java:34-95): they differ in the first near-origin threshold only (0.01 against
// RTStatics.EPSILON).  Math.pow(x, 2) of a float is exact in double, so the squares are taken in double.
// Returns w, or -1 where the reference returns null / false.
__device__ __forceinline__ float sphere_quadratic(V3 o, V3 d, V3 c, float radius, double first_threshold)
{
   const float a = (float) ((double) d.x * (double) d.x + (double) d.y * (double) d.y + (double) d.z * (double) d.z);
   const float ex = o.x - c.x, ey = o.y - c.y, ez = o.z - c.z;
   const float b = 2.0f * (d.x * ex + d.y * ey + d.z * ez);
   const float cc = (float) ((double) ex * (double) ex + (double) ey * (double) ey + (double) ez * (double) ez - (double) radius * (double) radius);
   const float b24c = (float) ((double) b * (double) b) - 4.0f * cc;
   if (b24c < 0) return -1.0f;
   const double root = sqrt((double) b24c);
   const float wplus = (float) ((double) -b + root) / (2.0f * a);
   const float wminus = (float) ((double) -b - root) / (2.0f * a);
   float w = least_positive(wplus, wminus);
   if (w <= 0) return -1.0f;
   if (w == wplus) {
      const float xn = o.x + d.x * wplus, yn = o.y + d.y * wplus, zn = o.z + d.z * wplus;
      const float xd = o.x - xn, yd = o.y - yn, zd = o.z - zn;
      const float n_dist = __builtin_sqrtf(xd * xd + yd * yd + zd * zd);
      if (((double) n_dist < first_threshold) && (wminus > 0)) w = wminus;
      else if (((double) n_dist < first_threshold) && (wminus < 0)) return -1.0f;
   } else if (w == wminus) {
      const float xm = o.x + d.x * wminus, ym = o.y + d.y * wminus, zm = o.z + d.z * wminus;
      const float xd = o.x - xm, yd = o.y - ym, zd = o.z - zm;
      const float m_dist = __builtin_sqrtf(xd * xd + yd * yd + zd * zd);
      if (((double) m_dist < 0.01) && (wplus > 0)) w = wplus;
      else if (((double) m_dist < 0.01) && (wplus < 0)) return -1.0f;
   }
   return (w > 0) ? w : -1.0f;
}

// BoundingSphere.intersects bounds/BoundingSphere.java:29-102 (the ray is not advanced)
__device__ __forceinline__ bool bounding_sphere_intersects(V3 o, V3 d, const DevObjectExt &e, float eps)
{
   return sphere_quadratic(o, d, mk3(e.center[0], e.center[1], e.center[2]), e.radius, (double) eps) > 0;
}

// Sphere.getIntersection sceneObjects/Sphere.java:29-112
__device__ __forceinline__ bool sphere_intersection(V3 o, V3 d, const DevObjectExt &e, float eps, V3 &P, V3 &N, float &w)
{
   const V3 c = mk3(e.center[0], e.center[1], e.center[2]);
   const V3 origin = add3(o, scale3(d, eps * 2.0f));   // :30-33
   w = sphere_quadratic(origin, d, c, e.radius, 0.01);
   if (!(w > 0)) return false;
   P = add3(scale3(d, w), origin);   // :102-104
   N = normalize3(sub3(P, c));
   return true;
}

// Plane.getIntersection sceneObjects/Plane.java:87-154; lo / hi = the plane's bounds (Plane.java:65-70)
__device__ __forceinline__ bool plane_intersection(V3 o, V3 d, const DevObjectExt &e, const float lo[3], const float hi[3], float eps,
                                                   V3 &P, V3 &N, float &w)
{
   const V3 origin = add3(o, scale3(d, eps * 2.0f));   // :88-91
   V3 pn = mk3(e.pnormal[0], e.pnormal[1], e.pnormal[2]);
   float vD = dot3(pn, d);
   if (vD > eps) {
      pn = scale3(pn, -1.0f);
      vD = dot3(pn, d);
   }
   if (!((vD <= -eps) || (vD >= eps))) return false;
   const float t = -(e.pA * origin.x + e.pB * origin.y + e.pC * origin.z + e.pD) / (e.pA * d.x + e.pB * d.y + e.pC * d.z);
   if (!(t > -eps)) return false;
   V3 rD = d;
   if (vD > -eps) rD = mk3(-rD.x, -rD.y, -rD.z);
   const V3 pI = mk3(origin.x + rD.x * t, origin.y + rD.y * t, origin.z + rD.z * t);
   // Plane.pointInPolygon :143-145
   if (!((pI.x >= lo[0] - eps) && (pI.x <= hi[0] + eps) && (pI.y >= lo[1] - eps) && (pI.y <= hi[1] + eps) && (pI.z >= lo[2] - eps) &&
         (pI.z <= hi[2] + eps)))
      return false;
   const V3 temp = sub3(pI, origin);
   P = pI;
   N = pn;
   w = __builtin_sqrtf(temp.x * temp.x + temp.y * temp.y + temp.z * temp.z);   // Vector3f.length()
   return true;
}

// BoundingVolume.intersects for a sphere or plane object (Camera.java:205)
__device__ __forceinline__ bool analytic_intersects(const DevObject &ob, const DevObjectExt &e, const Ray &r, float eps, float nearp, float farp)
{
   if (ob.kind == OBJ_SPHERE) return bounding_sphere_intersects(r.o, r.d, e, eps);
   float te;
   return slab_test(r, nearp, farp, ob.bmin[0], ob.bmin[1], ob.bmin[2], ob.bmax[0], ob.bmax[1], ob.bmax[2], te);
}

// BoundingVolume.getChildIntersection: BoundingSphere.java:24-26 tests itself again, AxisAlignedBoundingBox.java:26-28 does not
__device__ __forceinline__ bool analytic_child_intersection(const DevObject &ob, const DevObjectExt &e, const Ray &r, float eps, V3 &P,
                                                            V3 &N, float &w)
{
   if (ob.kind == OBJ_SPHERE) {
      if (!bounding_sphere_intersects(r.o, r.d, e, eps)) return false;
      return sphere_intersection(r.o, r.d, e, eps, P, N, w);
   }
   return plane_intersection(r.o, r.d, e, ob.bmin, ob.bmax, eps, P, N, w);
}

// RefractionMaterial.getRefractionDirection materials/RefractionMaterial.java:61-95
__device__ __forceinline__ V3 refraction_direction(V3 normal, V3 in_dir, float refraction_index)
{
   const float aether = 1.00029f;   // INDEX_OF_AIR
   V3 sn = normal;
   double n;
   double cos_i = (double) dot3(sn, in_dir);
   if (cos_i <= 0) {
      n = (double) (refraction_index / aether);
      cos_i = -cos_i;
   } else {
      n = (double) (aether / refraction_index);
      sn = scale3(sn, -1.0f);
   }
   const double snell_root = 1.0 - n * n * (1.0 - cos_i * cos_i);
   if (snell_root < 0) return reflect3(sn, in_dir);
   V3 out = scale3(in_dir, (float) n);
   out = add3(out, scale3(sn, (float) (n * cos_i - sqrt(snell_root))));
   return normalize3(out);
}

// (int) Math.floor(x) for a double
__device__ __forceinline__ int java_d2i(double v)
{
   if (v != v) return 0;
   if (v >= 2147483647.0) return 2147483647;
   if (v <= -2147483648.0) return (int) 0x80000000;
   return (int) v;
}

// CheckerboardMaterial.getMaterialColor materials/CheckerboardMaterial.java:21-49
__device__ __forceinline__ void checker_color(const float lo[3], const float hi[3], V3 p, const DevObjectExt &e, float out[3])
{
   const float x_span = hi[0] - lo[0], y_span = hi[1] - lo[1], z_span = hi[2] - lo[2];
   const float x_dist = p.x - lo[0], y_dist = p.y - lo[1], z_dist = p.z - lo[2];
   if ((x_dist < 0) || (y_dist < 0) || (z_dist < 0) || (x_dist > x_span) || (y_dist > y_span) || (z_dist > z_span)) {
      out[0] = out[1] = out[2] = 0.0f;
      return;
   }
   const int xc = java_d2i(floor((double) (x_dist / e.cell[0])));
   const int yc = java_d2i(floor((double) (y_dist / e.cell[1])));
   const int zc = java_d2i(floor((double) (z_dist / e.cell[2])));
   const bool one = ((xc % 2 == 0) && (yc % 2 == 0) && (zc % 2 == 0)) || ((xc % 2 == 1) && (yc % 2 == 0) && (zc % 2 == 1)) ||
                    ((xc % 2 == 0) && (yc % 2 == 1) && (zc % 2 == 1));
   const float *c = one ? e.color1 : e.color2;
   out[0] = c[0];
   out[1] = c[1];
   out[2] = c[2];
}

}   // namespace

}   // namespace mi355rt
