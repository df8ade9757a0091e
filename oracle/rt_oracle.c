/*
 * rt_oracle.c -- CPU restatement (C99, strict IEEE) of the sranger/Ray-Tracer per-pixel
 * trace loop.  TEST INFRASTRUCTURE ONLY -- see rt_oracle.h.  PARITY UNPINNED (no reference
 * tests or golden vectors exist; no JVM here).
 *
 * Build flags that matter: -ffp-contract=off -fno-fast-math (Java never fuses a*b+c and
 * evaluates float expressions in float).  The structure deliberately mirrors the Java
 * (pointer tree, recursive both-children traversal, per-face redundant box test) so that it
 * is the reference's algorithm, cost included -- it doubles as the CPU baseline.
 *
 * Citations are file:line under /root/reference/src/stephen/ranger/ar/ unless stated.
 * javax.vecmath 1.5.2 (lib/vecmath.jar, class files only) arithmetic follows SURVEY.md 8(c).
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- Java arithmetic helpers */

/* java.lang.Math.min/max(float,float): NaN-propagating, -0.0 < +0.0 */
static float jmin(float a, float b)
{
   if (a != a) return a;
   if (a == 0.0f && b == 0.0f && signbit(b)) return b;
   return (a <= b) ? a : b;
}
static float jmax(float a, float b)
{
   if (a != a) return a;
   if (a == 0.0f && b == 0.0f && signbit(a)) return b;
   return (a >= b) ? a : b;
}
/* (int) of a double, Java semantics (saturating, NaN -> 0) */
static int32_t jd2i(double d)
{
   if (d != d) return 0;
   if (d >= 2147483647.0) return 2147483647;
   if (d <= -2147483648.0) return (-2147483647 - 1);
   return (int32_t) d;
}
/* Math.toRadians: angdeg / 180.0 * PI (JDK <= 8 form; JDK >= 9 multiplies by a constant, which
 * can differ in the last double bit -- invisible after the (float) casts that follow) */
static double jtoradians(double deg) { return deg / 180.0 * 3.14159265358979323846; }

typedef struct { float x, y, z; } v3;

/* Vector3f.dot: (x*x' + y*y') + z*z', f32 */
static float v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* Vector3f.cross(a,b): note operand order in y */
static v3 v3cross(v3 a, v3 b)
{
   v3 r;
   r.x = a.y * b.z - a.z * b.y;
   r.y = b.x * a.z - b.z * a.x;
   r.z = a.x * b.y - a.y * b.x;
   return r;
}
static v3 v3sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static v3 v3add(v3 a, v3 b) { v3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static v3 v3scale(v3 a, float s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }
/* Vector3f.normalize: norm = (float)(1.0/Math.sqrt(x*x+y*y+z*z)); three f32 multiplies */
static v3 v3normalize(v3 a)
{
   float s = a.x * a.x + a.y * a.y + a.z * a.z;
   float norm = (float) (1.0 / sqrt((double) s));
   v3 r = { a.x * norm, a.y * norm, a.z * norm };
   return r;
}
/* RTStatics.getDistance(Vector3f,Vector3f) RTStatics.java:193-208 (and the float[] twins :167-182, :217-232) */
static float distance3(v3 p1, v3 p2)
{
   float d2 = (p1.x - p2.x) * (p1.x - p2.x) + (p1.y - p2.y) * (p1.y - p2.y) + (p1.z - p2.z) * (p1.z - p2.z);
   return (float) sqrt((double) d2);
}

/* ---------------------------------------------------------------- scene data */

typedef struct kdnode {
   struct kdnode *left, *right;
   float mm[2][3];
   int32_t *faces; /* leaf: face ids in stored order */
   uint32_t n_faces;
   int depth;
} kdnode;

typedef struct {
   uint32_t kind;         /* ORC_OBJ_* */
   float *vertices, *normals;
   int32_t *indices;
   uint32_t n_vertices, n_faces;
   orc_material mat;
   orc_material_ext ext;
   v3 center;             /* Sphere.origin / BoundingSphere.origin */
   float radius;
   float pA, pB, pC, pD;  /* Plane.A..D (Plane.java:37-51) */
   v3 pnormal;            /* Plane.normal (Plane.java:53-54) */
   float mm[2][3];        /* KDTree.minMax (KDTree.java:32-44); BoundingSphere.getMinMax; the Plane's AxisAlignedBoundingBox */
   float shadow_distance; /* KDTree.java:46 */
   kdnode *root;
   uint64_t n_nodes, n_leaves, n_leaf_slots;
   int32_t max_depth;
} object;

struct orc_scene {
   object *objs;
   uint32_t n_objs;
   float *brdf_dirs;    /* n_dirs * 3 remapped (BRDFMaterial.brdfDirections) */
   float *brdf_weights; /* n_materials * n_dirs */
   int32_t n_dirs, n_materials;
};

typedef struct {
   v3 origin, dir;
} ray;

typedef struct {
   v3 point, normal;
   float w;
   int32_t obj, face;
   ray r;
} hitinfo;

typedef struct {
   const orc_scene *scene;
   const orc_settings *st;
   orc_stats stats;
} ctx;

/* ---------------------------------------------------------------- RTStatics pieces */

/* RTStatics.aabbIntersection(Ray,float[][]) RTStatics.java:102-158 */
static int aabb_intersection(const ray *r, const float mm[2][3], float near_plane, float far_plane)
{
   float txmin, txmax, tymin, tymax, tzmin, tzmax;
   const float divx = 1.0f / r->dir.x;
   const float divy = 1.0f / r->dir.y;
   const float divz = 1.0f / r->dir.z;

   if (divx >= 0) {
      txmin = (mm[0][0] - r->origin.x) * divx;
      txmax = (mm[1][0] - r->origin.x) * divx;
   } else {
      txmin = (mm[1][0] - r->origin.x) * divx;
      txmax = (mm[0][0] - r->origin.x) * divx;
   }
   if (divy >= 0) {
      tymin = (mm[0][1] - r->origin.y) * divy;
      tymax = (mm[1][1] - r->origin.y) * divy;
   } else {
      tymin = (mm[1][1] - r->origin.y) * divy;
      tymax = (mm[0][1] - r->origin.y) * divy;
   }
   if (txmin > tymax || tymin > txmax) return 0;
   if (tymin > txmin) txmin = tymin;
   if (tymax < txmax) txmax = tymax;
   if (divz >= 0) {
      tzmin = (mm[0][2] - r->origin.z) * divz;
      tzmax = (mm[1][2] - r->origin.z) * divz;
   } else {
      tzmin = (mm[1][2] - r->origin.z) * divz;
      tzmax = (mm[0][2] - r->origin.z) * divz;
   }
   if (txmin > tzmax || tzmin > txmax) return 0;
   if (tzmin > txmin) txmin = tzmin;
   if (tzmax < txmax) txmax = tzmax;
   return txmin < far_plane && txmax > near_plane;
}

/* RTStatics.getReflectionDirection(n, dir) RTStatics.java:67-76 */
static v3 reflection_direction(v3 n, v3 dir)
{
   v3 N = v3scale(n, 2.0f);
   N = v3scale(N, v3dot(dir, n));
   return v3normalize(v3sub(dir, N));
}

/* RTStatics.offsetPosition RTStatics.java:608-615 */
static v3 offset_position(v3 p, v3 n, float eps) { return v3add(p, v3scale(n, eps)); }

/* RTStatics.getMinMax(vertices, int[] face, out) RTStatics.java:296-314 */
static void face_minmax(const float *verts, const int32_t *face, float out[2][3])
{
   int k, c;
   for (c = 0; c < 3; c++) { out[0][c] = 3.4028234663852886e38f; out[1][c] = -3.4028234663852886e38f; }
   for (k = 0; k < 3; k++) {
      const float *v = verts + 3 * (size_t) face[k];
      for (c = 0; c < 3; c++) {
         out[0][c] = jmin(out[0][c], v[c]);
         out[1][c] = jmax(out[1][c], v[c]);
      }
   }
}

/* RTStatics.computeNormal RTStatics.java:323-336 */
static v3 compute_face_normal(const float *verts, const int32_t *face)
{
   const float *a = verts + 3 * (size_t) face[0], *b = verts + 3 * (size_t) face[1], *c = verts + 3 * (size_t) face[2];
   v3 e1 = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
   v3 e2 = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
   return v3normalize(v3cross(e1, e2));
}

/* TriangleMesh.computeNormals (no-normals branch) sceneObjects/TriangleMesh.java:96-126 */
void orc_compute_normals(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces, float *nrm)
{
   uint32_t f, i;
   int k;
   memset(nrm, 0, sizeof(float) * 3 * (size_t) n_vertices);
   for (f = 0; f < n_faces; f++) {
      const int32_t *face = indices + 3 * (size_t) f;
      v3 n = compute_face_normal(vertices, face);
      for (k = 0; k < 3; k++) {
         float *d = nrm + 3 * (size_t) face[k];
         d[0] += n.x;
         d[1] += n.y;
         d[2] += n.z;
      }
   }
   for (i = 0; i < n_vertices; i++) {
      float *d = nrm + 3 * (size_t) i;
      /* RTStatics.getDistance(float[] zero, float[] n): (0-nx)*(0-nx)+... */
      float d2 = (0.0f - d[0]) * (0.0f - d[0]) + (0.0f - d[1]) * (0.0f - d[1]) + (0.0f - d[2]) * (0.0f - d[2]);
      float length = (float) sqrt((double) d2);
      d[0] /= length;
      d[1] /= length;
      d[2] /= length;
   }
}

/* Triangle.intersectsTriangle sceneObjects/Triangle.java:85-133; out = {x,y,z,nx,ny,nz,w} */
static int intersects_triangle(v3 rO, v3 rD, const float *P0, const float *P1, const float *P2, const float *N0,
                               const float *N1, const float *N2, float eps, float out[7])
{
   v3 p0 = { P0[0], P0[1], P0[2] }, p1 = { P1[0], P1[1], P1[2] }, p2 = { P2[0], P2[1], P2[2] };
   v3 e1 = v3sub(p1, p0);
   v3 e2 = v3sub(p2, p0);
   v3 p = v3cross(rD, e2);
   float divisor = v3dot(p, e1);
   v3 t, q;
   float u, v, w, dx, dy, dz;

   if ((divisor < eps) && (divisor > -eps)) return 0;
   t = v3sub(rO, p0);
   q = v3cross(t, e1);
   u = v3dot(p, t) / divisor;
   if ((u < 0) || (u > 1)) return 0;
   v = v3dot(q, rD) / divisor;
   if ((v < 0) || (v + u > 1)) return 0;
   w = 1.0f - u - v;
   out[0] = w * p0.x + u * p1.x + v * p2.x;
   out[1] = w * p0.y + u * p1.y + v * p2.y;
   out[2] = w * p0.z + u * p1.z + v * p2.z;
   out[3] = w * N0[0] + u * N1[0] + v * N2[0];
   out[4] = w * N0[1] + u * N1[1] + v * N2[1];
   out[5] = w * N0[2] + u * N1[2] + v * N2[2];
   /* RTStatics.getDistance(float[]{rO}, returnValue) RTStatics.java:167-182 */
   dx = rO.x - out[0];
   dy = rO.y - out[1];
   dz = rO.z - out[2];
   out[6] = (float) sqrt((double) (dx * dx + dy * dy + dz * dz));
   return 1;
}

/* ---------------------------------------------------------------- analytic primitives */

/* RTStatics.leastPositive RTStatics.java:82-100 */
static float least_positive(float i, float j)
{
   float ret;
   if (i < 0 && j < 0) ret = -1;
   else if (i < 0 && j > 0) ret = j;
   else if (i > 0 && j < 0) ret = i;
   else if (i < j) ret = i;
   else ret = j;
   return ret;
}

/* The quadratic of Sphere.getIntersection (Sphere.java:38-99) and BoundingSphere.intersects
 * (BoundingSphere.java:34-95); they differ only in the first near-origin threshold (0.01 against
 * RTStatics.EPSILON).  Math.pow(x, 2) of a float x is exact in double, so x * x in double restates it.
 * Returns w, or -1 where the reference returns null / false. */
static float sphere_quadratic(v3 origin, v3 direction, v3 center, float radius, double first_threshold)
{
   const float a = (float) ((double) direction.x * (double) direction.x + (double) direction.y * (double) direction.y +
                            (double) direction.z * (double) direction.z);
   const float b = 2 * (direction.x * (origin.x - center.x) + direction.y * (origin.y - center.y) + direction.z * (origin.z - center.z));
   const float ex = origin.x - center.x, ey = origin.y - center.y, ez = origin.z - center.z;
   const float c = (float) ((double) ex * (double) ex + (double) ey * (double) ey + (double) ez * (double) ez - (double) radius * (double) radius);
   const float b24c = (float) ((double) b * (double) b) - 4 * c;
   const float wplus = (float) ((double) -b + sqrt((double) b24c)) / (2 * a);
   const float wminus = (float) ((double) -b - sqrt((double) b24c)) / (2 * a);
   float w;
   if (b24c < 0) return -1;
   w = least_positive(wplus, wminus);
   if (w <= 0) return -1;
   if (w == wplus) {
      const float xn = origin.x + direction.x * wplus, yn = origin.y + direction.y * wplus, zn = origin.z + direction.z * wplus;
      const float xd = origin.x - xn, yd = origin.y - yn, zd = origin.z - zn;
      const float nDist = (float) sqrt((double) (xd * xd + yd * yd + zd * zd));
      if (((double) nDist < first_threshold) && (wminus > 0)) w = wminus;
      else if (((double) nDist < first_threshold) && (wminus < 0)) return -1;
   } else if (w == wminus) {
      const float xm = origin.x + direction.x * wminus, ym = origin.y + direction.y * wminus, zm = origin.z + direction.z * wminus;
      const float xd = origin.x - xm, yd = origin.y - ym, zd = origin.z - zm;
      const float mDist = (float) sqrt((double) (xd * xd + yd * yd + zd * zd));
      if (((double) mDist < 0.01) && (wplus > 0)) w = wplus;
      else if (((double) mDist < 0.01) && (wplus < 0)) return -1;
   }
   return (w > 0) ? w : -1;
}

/* BoundingSphere.intersects bounds/BoundingSphere.java:29-102 (the ray is not advanced) */
static int bounding_sphere_intersects(const ray *r, v3 center, float radius, float eps)
{
   return sphere_quadratic(r->origin, r->dir, center, radius, (double) eps) > 0;
}

/* Sphere.getIntersection sceneObjects/Sphere.java:29-112; out = {x,y,z,nx,ny,nz,w} */
static int sphere_intersection(const ray *r, v3 center, float radius, float eps, float out[7])
{
   const v3 origin = v3add(r->origin, v3scale(r->dir, eps * 2)); /* Sphere.java:30-33 */
   const float w = sphere_quadratic(origin, r->dir, center, radius, 0.01);
   v3 p, n;
   if (!(w > 0)) return 0;
   p = v3add(v3scale(r->dir, w), origin); /* Sphere.java:102-104 */
   n = v3normalize(v3sub(p, center));
   out[0] = p.x; out[1] = p.y; out[2] = p.z;
   out[3] = n.x; out[4] = n.y; out[5] = n.z;
   out[6] = w; /* offsetDistance is 0 (Sphere.java:35) */
   return 1;
}

/* Plane constructor sceneObjects/Plane.java:28-76; returns 0, or -1 for colinear corners (:56-58) */
static int plane_setup(object *o, const float *cn, uint32_t n_corners)
{
   float c[4][3];
   uint32_t i;
   int k;
   for (i = 0; i < 4; i++)
      for (k = 0; k < 3; k++) c[i][k] = cn[3 * (i < n_corners ? i : 2) + k]; /* :60-63 a triangle repeats its third corner */
   o->pA = c[0][1] * (c[1][2] - c[2][2]) + c[1][1] * (c[2][2] - c[0][2]) + c[2][1] * (c[0][2] - c[1][2]);
   o->pB = c[0][2] * (c[1][0] - c[2][0]) + c[1][2] * (c[2][0] - c[0][0]) + c[2][2] * (c[0][0] - c[1][0]);
   o->pC = c[0][0] * (c[1][1] - c[2][1]) + c[1][0] * (c[2][1] - c[0][1]) + c[2][0] * (c[0][1] - c[1][1]);
   o->pD = -(c[0][0] * (c[1][1] * c[2][2] - c[2][1] * c[1][2]) + c[1][0] * (c[2][1] * c[0][2] - c[0][1] * c[2][2]) +
             c[2][0] * (c[0][1] * c[1][2] - c[1][1] * c[0][2]));
   o->pnormal.x = o->pA; o->pnormal.y = o->pB; o->pnormal.z = o->pC;
   o->pnormal = v3normalize(o->pnormal);
   if ((o->pA == 0) && (o->pB == 0) && (o->pC == 0)) return -1;
   for (k = 0; k < 3; k++) {
      o->mm[0][k] = jmin(c[0][k], jmin(c[1][k], jmin(c[2][k], c[3][k])));
      o->mm[1][k] = jmax(c[0][k], jmax(c[1][k], jmax(c[2][k], c[3][k])));
   }
   return 0;
}

/* Plane.getIntersection sceneObjects/Plane.java:87-154 */
static int plane_intersection(const object *o, const ray *r, float eps, float out[7])
{
   const v3 origin = v3add(r->origin, v3scale(r->dir, eps * 2)); /* :88-91 */
   const v3 dir = r->dir;
   v3 pn = o->pnormal;
   float vD = v3dot(pn, dir);
   if (vD > eps) {
      pn = v3scale(pn, -1);
      vD = v3dot(pn, dir);
   }
   if ((vD <= -eps) || (vD >= eps)) {
      const float t = -(o->pA * origin.x + o->pB * origin.y + o->pC * origin.z + o->pD) / (o->pA * dir.x + o->pB * dir.y + o->pC * dir.z);
      if (t > -eps) {
         v3 rD = dir, pI;
         if (vD > -eps) { rD.x = -rD.x; rD.y = -rD.y; rD.z = -rD.z; }
         pI.x = origin.x + rD.x * t;
         pI.y = origin.y + rD.y * t;
         pI.z = origin.z + rD.z * t;
         /* Plane.pointInPolygon :143-145 */
         if ((pI.x >= o->mm[0][0] - eps) && (pI.x <= o->mm[1][0] + eps) && (pI.y >= o->mm[0][1] - eps) && (pI.y <= o->mm[1][1] + eps) &&
             (pI.z >= o->mm[0][2] - eps) && (pI.z <= o->mm[1][2] + eps)) {
            const v3 temp = v3sub(pI, origin);
            out[0] = pI.x; out[1] = pI.y; out[2] = pI.z;
            out[3] = pn.x; out[4] = pn.y; out[5] = pn.z;
            out[6] = (float) sqrt((double) (temp.x * temp.x + temp.y * temp.y + temp.z * temp.z)); /* Vector3f.length() */
            return 1;
         }
      }
   }
   return 0;
}

/* RefractionMaterial.getRefractionDirection materials/RefractionMaterial.java:61-95 */
static v3 refraction_direction(v3 normal, v3 in_dir, float refraction_index)
{
   const float aether = 1.00029f; /* INDEX_OF_AIR :12,20 */
   v3 surface_normal = normal, out;
   double n;
   double cosI = (double) v3dot(surface_normal, in_dir);
   double snell_root;
   if (cosI <= 0) {
      n = (double) (refraction_index / aether);
      cosI = -cosI;
   } else {
      n = (double) (aether / refraction_index);
      surface_normal = v3scale(surface_normal, -1.0f);
   }
   snell_root = 1.0 - n * n * (1.0 - cosI * cosI);
   if (snell_root < 0) {
      out = reflection_direction(surface_normal, in_dir);
   } else {
      out = v3scale(in_dir, (float) n);
      out = v3add(out, v3scale(surface_normal, (float) (n * cosI - sqrt(snell_root))));
      out = v3normalize(out);
   }
   return out;
}

/* CheckerboardMaterial.getMaterialColor materials/CheckerboardMaterial.java:21-49 */
static void checker_color(const float mm[2][3], v3 p, const orc_material_ext *e, float out[3])
{
   const float xSpan = mm[1][0] - mm[0][0], ySpan = mm[1][1] - mm[0][1], zSpan = mm[1][2] - mm[0][2];
   const float xDist = p.x - mm[0][0], yDist = p.y - mm[0][1], zDist = p.z - mm[0][2];
   int32_t xCell, yCell, zCell;
   if ((xDist < 0) || (yDist < 0) || (zDist < 0) || (xDist > xSpan) || (yDist > ySpan) || (zDist > zSpan)) {
      out[0] = out[1] = out[2] = 0.0f;
      return;
   }
   xCell = jd2i(floor((double) (xDist / e->checker_cell[0])));
   yCell = jd2i(floor((double) (yDist / e->checker_cell[1])));
   zCell = jd2i(floor((double) (zDist / e->checker_cell[2])));
   if (((xCell % 2 == 0) && (yCell % 2 == 0) && (zCell % 2 == 0)) || ((xCell % 2 == 1) && (yCell % 2 == 0) && (zCell % 2 == 1)) ||
       ((xCell % 2 == 0) && (yCell % 2 == 1) && (zCell % 2 == 1)))
      memcpy(out, e->checker_color1, sizeof(float) * 3);
   else
      memcpy(out, e->checker_color2, sizeof(float) * 3);
}

/* ---------------------------------------------------------------- hierarchy build */

typedef struct {
   const object *o;
   const orc_build_opts *opts;
   object *stats_o;
} buildctx;

/* KDNode constructor bounds/KDNode.java:29-90.  `faces` is owned by the callee. */
static kdnode *kdnode_new(buildctx *b, int32_t *faces, uint32_t n, const float mm[2][3], int axis, int depth)
{
   kdnode *node = (kdnode *) calloc(1, sizeof(kdnode));
   const object *o = b->o;
   memcpy(node->mm, mm, sizeof(node->mm));
   node->depth = depth;
   b->stats_o->n_nodes++;
   if (depth > b->stats_o->max_depth) b->stats_o->max_depth = depth;

   if (b->opts->use_kdtree && depth < b->opts->max_depth && n > (uint32_t) b->opts->max_children) {
      /* KDNode.java:45-51 */
      const float median = (mm[1][axis] - mm[0][axis]) / 2.0f + mm[0][axis];
      int32_t *lf = (int32_t *) malloc(sizeof(int32_t) * (n ? n : 1));
      int32_t *rf = (int32_t *) malloc(sizeof(int32_t) * (n ? n : 1));
      uint32_t nl = 0, nr = 0, i;
      float lmm[2][3], rmm[2][3], cmm[2][3];
      int c;
      const int next_axis = axis == 0 ? 1 : axis == 1 ? 2 : 0; /* RTStatics.java:50-52 */

      for (c = 0; c < 3; c++) {
         lmm[0][c] = rmm[0][c] = 3.4028234663852886e38f;
         lmm[1][c] = rmm[1][c] = -3.4028234663852886e38f;
      }
      /* KDNode.java:58-69 */
      for (i = 0; i < n; i++) {
         const int32_t f = faces[i];
         face_minmax(o->vertices, o->indices + 3 * (size_t) f, cmm);
         if (cmm[1][axis] <= median) {
            lf[nl++] = f;
         } else if (cmm[0][axis] >= median) {
            rf[nr++] = f;
         } else {
            lf[nl++] = f;
            rf[nr++] = f;
         }
      }
      free(faces);
      faces = NULL;
      /* KDNode.java:71-85; RTStatics.getMinMax(vertices, int[][] faces, out) RTStatics.java:244-264 */
      if (nl > 0) {
         for (i = 0; i < nl; i++) {
            face_minmax(o->vertices, o->indices + 3 * (size_t) lf[i], cmm);
            for (c = 0; c < 3; c++) {
               lmm[0][c] = jmin(lmm[0][c], cmm[0][c]);
               lmm[1][c] = jmax(lmm[1][c], cmm[1][c]);
            }
         }
         node->left = kdnode_new(b, lf, nl, (const float(*)[3]) lmm, next_axis, depth + 1);
      } else {
         free(lf);
      }
      if (nr > 0) {
         for (i = 0; i < nr; i++) {
            face_minmax(o->vertices, o->indices + 3 * (size_t) rf[i], cmm);
            for (c = 0; c < 3; c++) {
               rmm[0][c] = jmin(rmm[0][c], cmm[0][c]);
               rmm[1][c] = jmax(rmm[1][c], cmm[1][c]);
            }
         }
         node->right = kdnode_new(b, rf, nr, (const float(*)[3]) rmm, next_axis, depth + 1);
      } else {
         free(rf);
      }
   } else {
      node->faces = faces;
      node->n_faces = n;
      b->stats_o->n_leaves++;
      b->stats_o->n_leaf_slots += n;
   }
   return node;
}

static void kdnode_free(kdnode *n)
{
   if (!n) return;
   kdnode_free(n->left);
   kdnode_free(n->right);
   free(n->faces);
   free(n);
}

/* ---------------------------------------------------------------- traversal */

typedef struct {
   float v[7];
   int32_t face;
} leafhit;

/* KDNode.getChildIntersection bounds/KDNode.java:92-123 */
static int kdnode_child_intersection(ctx *c, const object *o, const kdnode *node, const ray *r, leafhit *out)
{
   const float nearp = c->st->near_plane, farp = c->st->far_plane;
   if (node->left != NULL || node->right != NULL) {
      leafhit tl, tr;
      int hl = 0, hr = 0;
      if (node->left != NULL) {
         c->stats.box_tests++;
         if (aabb_intersection(r, (const float(*)[3]) node->left->mm, nearp, farp)) hl = kdnode_child_intersection(c, o, node->left, r, &tl);
      }
      if (node->right != NULL) {
         c->stats.box_tests++;
         if (aabb_intersection(r, (const float(*)[3]) node->right->mm, nearp, farp)) hr = kdnode_child_intersection(c, o, node->right, r, &tr);
      }
      /* KDNode.java:105 -- tie => right */
      if (!hl && !hr) return 0;
      if (!hl) { *out = tr; return 1; }
      if (!hr) { *out = tl; return 1; }
      *out = (tl.v[6] < tr.v[6]) ? tl : tr;
      return 1;
   } else {
      leafhit temp;
      int have = 0;
      uint32_t i;
      for (i = 0; i < node->n_faces; i++) {
         /* KDNode.java:111 -- the node's own box, re-tested per face */
         c->stats.box_tests++;
         if (aabb_intersection(r, (const float(*)[3]) node->mm, nearp, farp)) {
            const int32_t f = node->faces[i];
            const int32_t *idx = o->indices + 3 * (size_t) f;
            c->stats.tri_tests++;
            if (intersects_triangle(r->origin, r->dir, o->vertices + 3 * (size_t) idx[0], o->vertices + 3 * (size_t) idx[1],
                                    o->vertices + 3 * (size_t) idx[2], o->normals + 3 * (size_t) idx[0],
                                    o->normals + 3 * (size_t) idx[1], o->normals + 3 * (size_t) idx[2], c->st->epsilon, temp.v)) {
               /* KDNode.java:114 -- first face wins ties */
               if (temp.v[6] > o->shadow_distance && (!have || temp.v[6] < out->v[6])) {
                  temp.face = f;
                  *out = temp;
                  have = 1;
               }
            }
         }
      }
      return have;
   }
}

/* KDTree.getChildIntersection bounds/KDTree.java:59-65 */
static int kdtree_child_intersection(ctx *c, uint32_t obj, const ray *r, hitinfo *out)
{
   const object *o = &c->scene->objs[obj];
   leafhit h;
   c->stats.box_tests++;
   if (!aabb_intersection(r, (const float(*)[3]) o->mm, c->st->near_plane, c->st->far_plane)) return 0;
   if (!kdnode_child_intersection(c, o, o->root, r, &h)) return 0;
   out->point.x = h.v[0]; out->point.y = h.v[1]; out->point.z = h.v[2];
   out->normal.x = h.v[3]; out->normal.y = h.v[4]; out->normal.z = h.v[5];
   out->w = h.v[6];
   out->obj = (int32_t) obj;
   out->face = h.face;
   out->r = *r;
   return 1;
}

/* BoundingVolume.intersects: KDTree / AxisAlignedBoundingBox -> RTStatics.aabbIntersection, BoundingSphere.java:29-102 */
static int object_intersects(ctx *c, const object *o, const ray *r)
{
   if (o->kind == ORC_OBJ_SPHERE) return bounding_sphere_intersects(r, o->center, o->radius, c->st->epsilon);
   if (o->kind == ORC_OBJ_MESH) c->stats.box_tests++;
   return aabb_intersection(r, (const float(*)[3]) o->mm, c->st->near_plane, c->st->far_plane);
}

/* BoundingVolume.getChildIntersection: KDTree.java:59-65, BoundingSphere.java:24-26 (tests itself again),
 * AxisAlignedBoundingBox.java:26-28 (no test) */
static int object_child_intersection(ctx *c, uint32_t obj, const ray *r, hitinfo *out)
{
   const object *o = &c->scene->objs[obj];
   float v[7];
   if (o->kind == ORC_OBJ_MESH) return kdtree_child_intersection(c, obj, r, out);
   if (o->kind == ORC_OBJ_SPHERE) {
      if (!bounding_sphere_intersects(r, o->center, o->radius, c->st->epsilon)) return 0;
      if (!sphere_intersection(r, o->center, o->radius, c->st->epsilon, v)) return 0;
   } else {
      if (!plane_intersection(o, r, c->st->epsilon, v)) return 0;
   }
   out->point.x = v[0]; out->point.y = v[1]; out->point.z = v[2];
   out->normal.x = v[3]; out->normal.y = v[4]; out->normal.z = v[5];
   out->w = v[6];
   out->obj = (int32_t) obj;
   out->face = 0;
   out->r = *r;
   return 1;
}

/* Camera.getClosestIntersection Camera.java:193-215 (Ray ctor Ray.java:9-13 normalises direction again) */
static int closest_intersection(ctx *c, int32_t mirror_obj, v3 origin, v3 direction, const v3 *normal, hitinfo *closest)
{
   ray r;
   uint32_t i;
   int have = 0;
   hitinfo temp;
   if (normal != NULL) origin = offset_position(origin, *normal, c->st->epsilon);
   r.origin = origin;
   r.dir = v3normalize(direction);
   for (i = 0; i < c->scene->n_objs; i++) {
      if (mirror_obj >= 0 && (uint32_t) mirror_obj == i) continue;
      if (!object_intersects(c, &c->scene->objs[i], &r)) continue;
      if (object_child_intersection(c, i, &r, &temp)) {
         if ((temp.w > c->st->epsilon) && (!have || (temp.w < closest->w))) {
            *closest = temp;
            have = 1;
         }
      }
   }
   return have;
}

/* RTStatics.shadowIntersects RTStatics.java:617-639 */
static int shadow_intersects(ctx *c, const hitinfo *info)
{
   const v3 light = { c->st->light_origin[0], c->st->light_origin[1], c->st->light_origin[2] };
   ray r;
   uint32_t i;
   hitinfo sh;
   v3 d = v3normalize(v3sub(light, info->point));
   r.origin = offset_position(info->point, info->normal, c->st->epsilon);
   r.dir = v3normalize(d); /* Ray ctor */
   c->stats.shadow_rays++;
   for (i = 0; i < c->scene->n_objs; i++) {
      if (object_child_intersection(c, i, &r, &sh) && sh.w > c->st->epsilon) {
         const float light_distance = distance3(sh.point, light);
         if (sh.w < light_distance + c->st->epsilon) return 1;
      }
   }
   return 0;
}

/* ---------------------------------------------------------------- colour helpers */

/* RTStatics.convertRGBtoHSV RTStatics.java:442-460 */
static void rgb_to_hsv(const float rgb[3], float hsv[3])
{
   const float R = rgb[0], G = rgb[1], B = rgb[2];
   float v, x, f;
   int i;
   x = jmin(R, jmin(G, B));
   v = jmax(R, jmax(G, B));
   if (v == x) {
      hsv[0] = -1;
      hsv[1] = 0;
      hsv[2] = v;
      return;
   }
   f = R == x ? G - B : G == x ? B - R : R - G;
   i = R == x ? 3 : G == x ? 5 : 1;
   hsv[0] = i - f / (v - x);
   hsv[1] = (v - x) / v;
   hsv[2] = v;
}

/* RTStatics.convertHSVtoRGB RTStatics.java:462-512 */
static void hsv_to_rgb(const float hsv[3], float rgb[3])
{
   const float h = hsv[0], s = hsv[1], v = hsv[2];
   float m, n, f;
   int i;
   if (h == -1) {
      rgb[0] = rgb[1] = rgb[2] = v;
      return;
   }
   i = jd2i(floor((double) h));
   f = h - i;
   if (i % 2 == 0) f = 1 - f;
   m = v * (1 - s);
   n = v * (1 - s * f);
   rgb[0] = rgb[1] = rgb[2] = 0.0f;
   switch (i) {
   case 6:
   case 0: rgb[0] = v; rgb[1] = n; rgb[2] = m; break;
   case 1: rgb[0] = n; rgb[1] = v; rgb[2] = m; break;
   case 2: rgb[0] = m; rgb[1] = v; rgb[2] = n; break;
   case 3: rgb[0] = m; rgb[1] = n; rgb[2] = v; break;
   case 4: rgb[0] = n; rgb[1] = m; rgb[2] = v; break;
   case 5: rgb[0] = v; rgb[1] = m; rgb[2] = n; break;
   default: break;
   }
   rgb[0] = jmin(1.0f, jmax(0.0f, rgb[0]));
   rgb[1] = jmin(1.0f, jmax(0.0f, rgb[1]));
   rgb[2] = jmin(1.0f, jmax(0.0f, rgb[2]));
}

/* ---------------------------------------------------------------- PBRTMath / BRDFMaterial */

#define F_PI ((float) 3.14159265358979323846)
#define F_2_PI (2.0f * (float) 3.14159265358979323846)

/* PBRTMath.normalizePhi PBRTMath.java:99-101 (compare against the double PI/2f) */
static float normalize_phi(float phi) { return ((double) phi < 3.14159265358979323846 / 2.0f) ? phi + 2.0f * F_PI : phi; }

/* PBRTMath.getRemappedDirection(float[] in, float[] out) PBRTMath.java:74-97 */
static void remap_angles(const float in[2], const float out[2], float res[3])
{
   const float cosi = (float) cos((double) in[0]);
   const float coso = (float) cos((double) out[0]);
   const float sini = (float) sin((double) in[0]);
   const float sino = (float) sin((double) out[0]);
   const float phii = normalize_phi(in[1]);
   const float phio = normalize_phi(out[1]);
   float dphi = phii - phio;
   if (dphi < 0.0f) dphi += F_2_PI;
   if (dphi > F_2_PI) dphi -= F_2_PI;
   if (dphi > F_PI) dphi = F_2_PI - dphi;
   res[0] = sini * sino;
   res[1] = dphi / F_PI;
   res[2] = cosi * coso;
}

/* PBRTMath.getThetaPhi PBRTMath.java:103-112 */
static void theta_phi(v3 n, v3 s, v3 t, v3 dir, float out[2])
{
   const float z = v3dot(dir, n);
   const float y = v3dot(dir, s);
   const float x = v3dot(dir, t);
   out[0] = (float) acos((double) z);
   out[1] = (float) atan2((double) y, (double) x);
}

/* PBRTMath.getNormalTangent PBRTMath.java:118-130 */
static v3 normal_tangent(v3 normal)
{
   v3 rv = { 0, 1, 0 };
   const float dot = fabsf(v3dot(normal, rv));
   if (dot > 0.9f) { rv.x = -1; rv.y = 0; rv.z = 0; }
   return v3cross(normal, rv);
}

/* PBRTMath.getRemappedDirection(normal, tangent, inDir, outDir) PBRTMath.java:59-72 */
static void remap_vectors(v3 normal, v3 tangent, v3 in_dir, v3 out_dir, float res[3])
{
   v3 n = v3normalize(normal);
   v3 t = v3normalize(tangent);
   v3 s = v3normalize(v3cross(n, t));
   float in[2], out[2];
   theta_phi(n, s, t, in_dir, in);
   theta_phi(n, s, t, out_dir, out);
   remap_angles(in, out, res);
}

/* PBRTMath.getDistance2 PBRTMath.java:55-57 */
static float distance2_abs(const float p1[3], const float p2[3])
{
   return fabsf((p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]));
}

/* BRDFMaterial.getBRDFLuminocity materials/BRDFMaterial.java:53-92 */
static float brdf_luminocity(const orc_scene *sc, int32_t brdf_id, int32_t brdf_samples, v3 ray_dir, v3 point, v3 normal, v3 light)
{
   float luminocity = 0.0f, weight = 0.0f;
   int ctr = 0, i, j;
   const float *material = sc->brdf_weights + (size_t) brdf_id * (size_t) sc->n_dirs;
   v3 neg_ray = v3normalize(v3scale(ray_dir, -1.0f));
   v3 temp_dir = v3normalize(v3sub(light, point));
   v3 tangent = normal_tangent(normal);

   for (i = 0; i < brdf_samples; i++) {
      float remapped[3];
      float last_max_dist2 = 0.001f;
      remap_vectors(normal, tangent, neg_ray, reflection_direction(normal, temp_dir), remapped);
      while ((ctr < 4) && (last_max_dist2 < 1.5f)) {
         for (j = 0; j < sc->n_dirs; j++) {
            const float dist2 = distance2_abs(remapped, sc->brdf_dirs + 3 * (size_t) j);
            if (dist2 < last_max_dist2) {
               const float temp = (float) exp(-100.0 * (double) dist2);
               luminocity += material[j] * temp;
               weight += temp;
               ctr++;
            }
         }
         last_max_dist2 *= 2.0f;
      }
   }
   if (ctr == 0) return 0.0f;
   return luminocity / weight * 6.0f;
}

/* ---------------------------------------------------------------- materials / lighting */

static void material_color(ctx *c, const hitinfo *info, int bounces, int n_terms, float out[3]);

/* BoundingVolume.getColor: KDTree.getColor bounds/KDTree.java:77-80 has no depth cut-off; BoundingSphere /
 * AxisAlignedBoundingBox forward to SceneObject.getColor sceneObjects/SceneObject.java:64-70, which has one.
 * `bounces` is the reference's `depth`; n_terms counts the RefractionMaterial terms pending on this path. */
static void material_color(ctx *c, const hitinfo *info, int bounces, int n_terms, float out[3])
{
   const object *o = &c->scene->objs[info->obj];
   const v3 light = { c->st->light_origin[0], c->st->light_origin[1], c->st->light_origin[2] };
   if (o->kind != ORC_OBJ_MESH && bounces >= c->st->max_recursion_depth) { /* SceneObject.java:65-67 */
      memcpy(out, o->mat.diffuse, sizeof(float) * 3);
      return;
   }
   switch (o->mat.kind) {
   case ORC_MAT_BRDF: {
      /* BRDFMaterial.getMaterialColor materials/BRDFMaterial.java:158-167 */
      float hsv[3];
      const float lum = brdf_luminocity(c->scene, o->mat.brdf_id, c->st->brdf_samples, info->r.dir, info->point, info->normal, light);
      rgb_to_hsv(o->mat.diffuse, hsv);
      hsv[2] = lum;
      hsv_to_rgb(hsv, out);
      return;
   }
   case ORC_MAT_REFLECTION: {
      /* ReflectionMaterial.getMaterialColor materials/ReflectionMaterial.java:14-24 */
      hitinfo m;
      v3 refl;
      if (bounces >= c->st->max_mirror_bounces) { /* guard: the reference would recurse on */
         memcpy(out, o->mat.diffuse, sizeof(float) * 3);
         return;
      }
      refl = reflection_direction(info->normal, info->r.dir);
      c->stats.secondary_rays++;
      if (!closest_intersection(c, info->obj, info->point, refl, &info->normal, &m)) {
         memcpy(out, c->st->light_ambient, sizeof(float) * 3);
      } else {
         material_color(c, &m, bounces + 1, n_terms, out);
      }
      return;
   }
   case ORC_MAT_REFRACTION: {
      /* RefractionMaterial.getMaterialColor materials/RefractionMaterial.java:29-59 */
      hitinfo m;
      v3 dir;
      int entering, k;
      if (bounces >= c->st->max_mirror_bounces || n_terms >= ORC_MAX_REFRACTION_TERMS) { /* guards */
         memcpy(out, o->mat.diffuse, sizeof(float) * 3);
         return;
      }
      dir = refraction_direction(info->normal, info->r.dir, o->ext.refraction_index);
      c->stats.secondary_rays++;
      if (!closest_intersection(c, info->obj, info->point, dir, &info->normal, &m)) {
         memcpy(out, c->st->light_ambient, sizeof(float) * 3);
         return;
      }
      entering = v3dot(info->normal, info->r.dir) <= 0.0f;
      material_color(c, &m, bounces + 1, n_terms + (entering ? 1 : 0), out);
      if (entering) {
         const float distance = distance3(info->point, m.point);
         for (k = 0; k < 3; k++) {
            float col = o->mat.diffuse[k];
            col *= 0.15f * -distance;
            out[k] = (float) ((double) out[k] + exp((double) col)); /* returnColor[k] += Math.exp(color[k]) */
         }
      }
      return;
   }
   case ORC_MAT_CHECKERBOARD:
      checker_color((const float(*)[3]) o->mm, info->point, &o->ext, out);
      return;
   default:
      /* ColorInformation.getMaterialColor materials/ColorInformation.java:35-37 */
      memcpy(out, o->mat.diffuse, sizeof(float) * 3);
      return;
   }
}

/* the arithmetic of PhongLightingModel.getPixelColor lighting/PhongLightingModel.java:23-53 */
static void phong_combine(const float color_in[3], const float kd[3], const float ks[3], float a, const float is_[3],
                          const float ia[3], v3 light, v3 point, v3 normal, v3 ray_dir, int shadowed, float out[3])
{
   const float *id = is_; /* light.emission used for both (PhongLightingModel.java:28-29) */
   v3 L = v3normalize(v3sub(light, point));
   v3 N = normal;
   v3 zero = { 0, 0, 0 };
   v3 V = v3sub(zero, ray_dir); /* V.sub(direction) on a zero vector */
   v3 R = reflection_direction(normal, L);
   const double LdotN = (double) v3dot(L, N);
   const double RdotVexpA = pow((double) v3dot(V, R), (double) a);
   const double spec = shadowed ? 0.0f : 1.0f;
   const double shade = shadowed ? 0.6f : 1.0f;
   int k;
   for (k = 0; k < 3; k++) {
      const double term = shade * (kd[k] * LdotN * id[k] + spec * ks[k] * RdotVexpA * is_[k] + 0.4f * ia[k]);
      out[k] = (float) ((double) color_in[k] * term); /* color[k] *= (double) */
   }
}

/* LightingModel.getPixelColor / PhongLightingModel.getPixelColor */
static void pixel_color(ctx *c, const hitinfo *info, float out[3])
{
   float color[3];
   material_color(c, info, 0, 0, color);
   if (c->st->lighting_model == ORC_LIGHTING_PHONG) {
      const object *o = &c->scene->objs[info->obj];
      const v3 light = { c->st->light_origin[0], c->st->light_origin[1], c->st->light_origin[2] };
      const int shadowed = shadow_intersects(c, info);
      phong_combine(color, o->mat.diffuse, o->mat.specular, o->mat.shininess, c->st->light_emission, c->st->light_ambient,
                    light, info->point, info->normal, info->r.dir, shadowed, out);
   } else {
      memcpy(out, color, sizeof(color));
   }
}

/* ---------------------------------------------------------------- sample jitter RNG */

/* The reference draws jitter from an unseeded java.util.Random per render thread and scatters
 * pixels to threads at random (RenderThread.java:32, Camera.java:131-143), so samples >= 1 are not
 * reproducible even Java-vs-Java.  Here each pixel owns a java.util.Random stream seeded from
 * (rng_seed, pixel index) through a splitmix64 finaliser; the generator itself is Java's LCG. */
uint64_t orc_rng_init(uint64_t seed, uint64_t pixel_index)
{
   uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (pixel_index + 1ULL);
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
   z = z ^ (z >> 31);
   return (z ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1); /* Random(long seed) scramble */
}
float orc_rng_next_float(uint64_t *state)
{
   /* Random.next(24) / (float)(1 << 24) */
   *state = (*state * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
   return (float) (int32_t) (*state >> 24) / (float) (1 << 24);
}

/* ---------------------------------------------------------------- render */

typedef struct {
   const orc_scene *scene;
   const orc_camera *cam;
   const orc_settings *st;
   int tid, n_threads, pixel_stride;
   float *rgb, *hit_w;
   int32_t *hit_id, *hit_obj;
   orc_stats stats;
} job;

/* RenderThread.run RenderThread.java:30-69 + Camera.createImage constants Camera.java:123-126
 * + RTStatics.computeColorAverage RTStatics.java:363-373 + Camera.setPixel Camera.java:96-101 */
static void *render_job(void *arg)
{
   job *jb = (job *) arg;
   const orc_camera *cam = jb->cam;
   const orc_settings *st = jb->st;
   const int W = cam->width, H = cam->height;
   const float xMin = -(cam->viewport_w / 2.0f);
   const float yMin = cam->viewport_h / 2.0f;
   const float xInc = cam->viewport_w / W;
   const float yInc = -cam->viewport_h / H;
   const v3 origin = { cam->origin[0], cam->origin[1], cam->origin[2] };
   const float *m = cam->rotation;
   const int ms = st->multi_samples < 1 ? 1 : st->multi_samples;
   ctx c;
   uint64_t p, k;
   const uint64_t total = (uint64_t) W * (uint64_t) H;
   memset(&c, 0, sizeof(c));
   c.scene = jb->scene;
   c.st = st;

   for (k = (uint64_t) jb->tid; k * (uint64_t) jb->pixel_stride < total; k += (uint64_t) jb->n_threads) {
      int px, py, j;
      float x, y, r = 0, g = 0, b = 0;
      uint64_t rng;
      p = k * (uint64_t) jb->pixel_stride;
      px = (int) (p % (uint64_t) W);
      py = (int) (p / (uint64_t) W);
      x = px * xInc + xMin;
      y = py * yInc + yMin;
      rng = orc_rng_init(st->rng_seed, p);
      for (j = 0; j < ms; j++) {
         v3 vd, t;
         hitinfo closest;
         float color[3];
         vd.x = j == 0 ? x : x + (orc_rng_next_float(&rng) * 2.0f - 1.0f) * xInc / 2.0f;
         vd.y = j == 0 ? y : y + +(orc_rng_next_float(&rng) * 2.0f - 1.0f) * yInc / 2.0f;
         vd.z = -cam->near_plane;
         /* Matrix4f.transform(Vector3f): rotation only */
         t.x = m[0] * vd.x + m[1] * vd.y + m[2] * vd.z;
         t.y = m[3] * vd.x + m[4] * vd.y + m[5] * vd.z;
         t.z = m[6] * vd.x + m[7] * vd.y + m[8] * vd.z;
         vd = v3normalize(t);
         c.stats.primary_rays++;
         if (closest_intersection(&c, -1, origin, vd, NULL, &closest)) {
            c.stats.hits++;
            pixel_color(&c, &closest, color);
            if (j == 0) {
               if (jb->hit_id) jb->hit_id[p] = closest.face;
               if (jb->hit_obj) jb->hit_obj[p] = closest.obj;
               if (jb->hit_w) jb->hit_w[p] = closest.w;
            }
         } else {
            color[0] = st->light_ambient[0];
            color[1] = st->light_ambient[1];
            color[2] = st->light_ambient[2];
            if (j == 0) {
               if (jb->hit_id) jb->hit_id[p] = -1;
               if (jb->hit_obj) jb->hit_obj[p] = -1;
               if (jb->hit_w) jb->hit_w[p] = 0.0f;
            }
         }
         r += color[0];
         g += color[1];
         b += color[2];
      }
      r = r / ms;
      g = g / ms;
      b = b / ms;
      if (r != r || g != g || b != b) r = g = b = 0.0f; /* Camera.setPixel NaN scrub */
      if (jb->rgb) {
         jb->rgb[3 * p + 0] = r;
         jb->rgb[3 * p + 1] = g;
         jb->rgb[3 * p + 2] = b;
      }
   }
   jb->stats = c.stats;
   return NULL;
}

int orc_render(const orc_scene *s, const orc_camera *cam, const orc_settings *st, int32_t n_threads, int32_t pixel_stride,
               float *rgb, int32_t *hit_id, int32_t *hit_obj, float *hit_w, orc_stats *stats)
{
   job *jobs;
   pthread_t *th;
   int i;
   if (!s || !cam || !st || cam->width < 1 || cam->height < 1) return -1;
   if (n_threads < 1) n_threads = 1;
   if (pixel_stride < 1) pixel_stride = 1;
   jobs = (job *) calloc((size_t) n_threads, sizeof(job));
   th = (pthread_t *) calloc((size_t) n_threads, sizeof(pthread_t));
   for (i = 0; i < n_threads; i++) {
      jobs[i].scene = s;
      jobs[i].cam = cam;
      jobs[i].st = st;
      jobs[i].tid = i;
      jobs[i].n_threads = n_threads;
      jobs[i].pixel_stride = pixel_stride;
      jobs[i].rgb = rgb;
      jobs[i].hit_id = hit_id;
      jobs[i].hit_obj = hit_obj;
      jobs[i].hit_w = hit_w;
      if (n_threads == 1) render_job(&jobs[i]);
      else pthread_create(&th[i], NULL, render_job, &jobs[i]);
   }
   if (stats) memset(stats, 0, sizeof(*stats));
   for (i = 0; i < n_threads; i++) {
      if (n_threads > 1) pthread_join(th[i], NULL);
      if (stats) {
         stats->primary_rays += jobs[i].stats.primary_rays;
         stats->shadow_rays += jobs[i].stats.shadow_rays;
         stats->secondary_rays += jobs[i].stats.secondary_rays;
         stats->box_tests += jobs[i].stats.box_tests;
         stats->tri_tests += jobs[i].stats.tri_tests;
         stats->hits += jobs[i].stats.hits;
      }
   }
   free(jobs);
   free(th);
   return 0;
}

int orc_trace_rays(const orc_scene *s, const orc_settings *st, uint64_t n, const float *origins, const float *dirs,
                   int32_t *hit_id, int32_t *hit_obj, float *hit_w, float *hit_point, float *hit_normal)
{
   ctx c;
   uint64_t i;
   if (!s || !st) return -1;
   memset(&c, 0, sizeof(c));
   c.scene = s;
   c.st = st;
   for (i = 0; i < n; i++) {
      v3 o = { origins[3 * i], origins[3 * i + 1], origins[3 * i + 2] };
      v3 d = { dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2] };
      hitinfo h;
      if (closest_intersection(&c, -1, o, d, NULL, &h)) {
         if (hit_id) hit_id[i] = h.face;
         if (hit_obj) hit_obj[i] = h.obj;
         if (hit_w) hit_w[i] = h.w;
         if (hit_point) { hit_point[3 * i] = h.point.x; hit_point[3 * i + 1] = h.point.y; hit_point[3 * i + 2] = h.point.z; }
         if (hit_normal) { hit_normal[3 * i] = h.normal.x; hit_normal[3 * i + 1] = h.normal.y; hit_normal[3 * i + 2] = h.normal.z; }
      } else {
         if (hit_id) hit_id[i] = -1;
         if (hit_obj) hit_obj[i] = -1;
         if (hit_w) hit_w[i] = 0.0f;
         if (hit_point) hit_point[3 * i] = hit_point[3 * i + 1] = hit_point[3 * i + 2] = 0.0f;
         if (hit_normal) hit_normal[3 * i] = hit_normal[3 * i + 1] = hit_normal[3 * i + 2] = 0.0f;
      }
   }
   return 0;
}

/* closest hit of each ray, then camera.lightingModel.getPixelColor(info, 0) for it (RenderThread.java:49-52 for an
 * arbitrary ray): what tools/java/GoldenDumper.java prints from the reference itself */
int orc_shade_rays(const orc_scene *s, const orc_settings *st, uint64_t n, const float *origins, const float *dirs,
                   int32_t *hit_obj, float *hit_w, float *hit_point, float *hit_normal, float *rgb)
{
   ctx c;
   uint64_t i;
   if (!s || !st || !hit_obj || !hit_w || !hit_point || !hit_normal || !rgb) return -1;
   memset(&c, 0, sizeof(c));
   c.scene = s;
   c.st = st;
   for (i = 0; i < n; i++) {
      v3 o = { origins[3 * i], origins[3 * i + 1], origins[3 * i + 2] };
      v3 d = { dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2] };
      hitinfo h;
      float col[3] = { 0.0f, 0.0f, 0.0f };
      if (closest_intersection(&c, -1, o, d, NULL, &h)) {
         pixel_color(&c, &h, col);
         hit_obj[i] = h.obj;
         hit_w[i] = h.w;
         hit_point[3 * i] = h.point.x; hit_point[3 * i + 1] = h.point.y; hit_point[3 * i + 2] = h.point.z;
         hit_normal[3 * i] = h.normal.x; hit_normal[3 * i + 1] = h.normal.y; hit_normal[3 * i + 2] = h.normal.z;
      } else {
         hit_obj[i] = -1;
         hit_w[i] = 0.0f;
         hit_point[3 * i] = hit_point[3 * i + 1] = hit_point[3 * i + 2] = 0.0f;
         hit_normal[3 * i] = hit_normal[3 * i + 1] = hit_normal[3 * i + 2] = 0.0f;
      }
      rgb[3 * i] = col[0]; rgb[3 * i + 1] = col[1]; rgb[3 * i + 2] = col[2];
   }
   return 0;
}

/* ---------------------------------------------------------------- scene create / camera */

int orc_scene_create_objects(const orc_object_desc *objects, uint32_t n_objects, const orc_build_opts *opts, orc_scene **out)
{
   orc_scene *s;
   uint32_t i;
   if ((!objects && n_objects) || !opts || !out) return -1;
   s = (orc_scene *) calloc(1, sizeof(orc_scene));
   s->objs = (object *) calloc(n_objects ? n_objects : 1, sizeof(object));
   s->n_objs = n_objects;
   for (i = 0; i < n_objects; i++) {
      const orc_object_desc *od = &objects[i];
      const orc_mesh_desc *md = &od->mesh;
      object *o = &s->objs[i];
      uint32_t v, f;
      int c;
      int32_t *faces;
      buildctx b;
      orc_build_opts obj_opts;
      o->kind = od->kind;
      o->mat = md->material;
      o->ext = od->ext;
      if (od->kind == ORC_OBJ_SPHERE) {
         /* Sphere ctor Sphere.java:15-22, BoundingSphere.getMinMax BoundingSphere.java:105-111 */
         o->center.x = od->center[0]; o->center.y = od->center[1]; o->center.z = od->center[2];
         o->radius = od->radius;
         for (c = 0; c < 3; c++) {
            o->mm[0][c] = od->center[c] - od->radius;
            o->mm[1][c] = od->center[c] + od->radius;
         }
         continue;
      }
      if (od->kind == ORC_OBJ_PLANE) {
         if (od->n_corners < 3 || od->n_corners > 4 || plane_setup(o, od->corners, od->n_corners) != 0) { orc_scene_destroy(s); return -3; }
         continue;
      }
      if (od->kind != ORC_OBJ_MESH) { orc_scene_destroy(s); return -1; }
      o->n_vertices = md->n_vertices;
      o->n_faces = md->n_faces;
      o->vertices = (float *) malloc(sizeof(float) * 3 * (size_t) (md->n_vertices ? md->n_vertices : 1));
      o->normals = (float *) malloc(sizeof(float) * 3 * (size_t) (md->n_vertices ? md->n_vertices : 1));
      o->indices = (int32_t *) malloc(sizeof(int32_t) * 3 * (size_t) (md->n_faces ? md->n_faces : 1));
      memcpy(o->vertices, md->vertices, sizeof(float) * 3 * (size_t) md->n_vertices);
      memcpy(o->indices, md->indices, sizeof(int32_t) * 3 * (size_t) md->n_faces);
      for (f = 0; f < 3 * md->n_faces; f++) {
         if (o->indices[f] < 0 || (uint32_t) o->indices[f] >= md->n_vertices) { orc_scene_destroy(s); return -2; }
      }
      if (md->normals) memcpy(o->normals, md->normals, sizeof(float) * 3 * (size_t) md->n_vertices);
      else orc_compute_normals(o->vertices, o->n_vertices, o->indices, o->n_faces, o->normals);
      /* KDTree ctor bounds/KDTree.java:32-46: bounds over ALL vertices */
      for (c = 0; c < 3; c++) { o->mm[0][c] = 3.4028234663852886e38f; o->mm[1][c] = -3.4028234663852886e38f; }
      for (v = 0; v < md->n_vertices; v++) {
         for (c = 0; c < 3; c++) {
            o->mm[0][c] = jmin(o->vertices[3 * (size_t) v + c], o->mm[0][c]);
            o->mm[1][c] = jmax(o->vertices[3 * (size_t) v + c], o->mm[1][c]);
         }
      }
      o->shadow_distance = jmax(o->mm[1][0] - o->mm[0][0], jmax(o->mm[1][1] - o->mm[0][1], o->mm[1][2] - o->mm[0][2])) / 2000.0f;
      faces = (int32_t *) malloc(sizeof(int32_t) * (size_t) (md->n_faces ? md->n_faces : 1));
      for (f = 0; f < md->n_faces; f++) faces[f] = (int32_t) f;
      b.o = o;
      obj_opts = *opts;   /* computeKDTree is per TriangleMesh in the reference (TriangleMesh.java:89) */
      if (md->compute_kdtree != 0) obj_opts.use_kdtree = md->compute_kdtree > 0;
      b.opts = &obj_opts;
      b.stats_o = o;
      o->root = kdnode_new(&b, faces, md->n_faces, (const float(*)[3]) o->mm, 0, 0);
   }
   *out = s;
   return 0;
}

int orc_scene_create(const orc_mesh_desc *meshes, uint32_t n_meshes, const orc_build_opts *opts, orc_scene **out)
{
   orc_object_desc *od;
   uint32_t i;
   int rc;
   if (!meshes || !opts || !out) return -1;
   od = (orc_object_desc *) calloc(n_meshes ? n_meshes : 1, sizeof(orc_object_desc));
   for (i = 0; i < n_meshes; i++) {
      od[i].kind = ORC_OBJ_MESH;
      od[i].mesh = meshes[i];
   }
   rc = orc_scene_create_objects(od, n_meshes, opts, out);
   free(od);
   return rc;
}

void orc_scene_destroy(orc_scene *s)
{
   uint32_t i;
   if (!s) return;
   for (i = 0; i < s->n_objs; i++) {
      kdnode_free(s->objs[i].root);
      free(s->objs[i].vertices);
      free(s->objs[i].normals);
      free(s->objs[i].indices);
   }
   free(s->objs);
   free(s->brdf_dirs);
   free(s->brdf_weights);
   free(s);
}

/* BRDFMaterial.remapDirections materials/BRDFMaterial.java:148-156 over parsed table.txt rows
 * {theta_v, phi_v, theta_i, phi_i}; weights = parsed abrdf.dat rows */
int orc_scene_set_brdf(orc_scene *s, const float *angles, int32_t n_dirs, const float *weights, int32_t n_materials)
{
   int32_t i;
   if (!s || !angles || !weights || n_dirs < 0 || n_materials < 0) return -1;
   free(s->brdf_dirs);
   free(s->brdf_weights);
   s->brdf_dirs = (float *) malloc(sizeof(float) * 3 * (size_t) (n_dirs ? n_dirs : 1));
   s->brdf_weights = (float *) malloc(sizeof(float) * (size_t) (n_dirs ? n_dirs : 1) * (size_t) (n_materials ? n_materials : 1));
   for (i = 0; i < n_dirs; i++) {
      const float in[2] = { angles[4 * i + 0], angles[4 * i + 1] };
      const float out[2] = { angles[4 * i + 2], angles[4 * i + 3] };
      remap_angles(in, out, s->brdf_dirs + 3 * (size_t) i);
   }
   memcpy(s->brdf_weights, weights, sizeof(float) * (size_t) n_dirs * (size_t) n_materials);
   s->n_dirs = n_dirs;
   s->n_materials = n_materials;
   return 0;
}

int orc_scene_object_info(const orc_scene *s, uint32_t obj, float minmax[6], float *shadow_distance, uint64_t *n_nodes,
                          uint64_t *n_leaves, uint64_t *n_leaf_slots, int32_t *max_depth)
{
   const object *o;
   if (!s || obj >= s->n_objs) return -1;
   o = &s->objs[obj];
   if (minmax) memcpy(minmax, o->mm, sizeof(float) * 6);
   if (shadow_distance) *shadow_distance = o->shadow_distance;
   if (n_nodes) *n_nodes = o->n_nodes;
   if (n_leaves) *n_leaves = o->n_leaves;
   if (n_leaf_slots) *n_leaf_slots = o->n_leaf_slots;
   if (max_depth) *max_depth = o->max_depth;
   return 0;
}

const float *orc_scene_object_normals(const orc_scene *s, uint32_t obj)
{
   if (!s || obj >= s->n_objs) return NULL;
   return s->objs[obj].normals;
}

/* scene bounds as both Scene.java:21-33 and Camera.java:59-70 compute them */
static void scene_bounds(const orc_scene *s, float mm[2][3])
{
   uint32_t i;
   int c;
   for (c = 0; c < 3; c++) { mm[0][c] = 3.4028234663852886e38f; mm[1][c] = -3.4028234663852886e38f; }
   for (i = 0; i < s->n_objs; i++) {
      for (c = 0; c < 3; c++) {
         mm[0][c] = jmin(mm[0][c], s->objs[i].mm[0][c]);
         mm[1][c] = jmax(mm[1][c], s->objs[i].mm[1][c]);
      }
   }
}

/* Scene ctor Scene.java:35-36 */
float orc_scene_epsilon(const orc_scene *s)
{
   float mm[2][3];
   float max_span;
   scene_bounds(s, mm);
   max_span = jmax(mm[1][0] - mm[0][0], jmax(mm[1][1] - mm[0][1], mm[1][2] - mm[0][2]));
   return (max_span < 10) ? 1e-15f : 1e-3f;
}

typedef struct { float x, y, z, w; } quat;

/* RTStatics.initializeQuat4f(axis, angle) RTStatics.java:356-361 + Quat4f.set(AxisAngle4f) */
static quat quat_axis_angle(float ax, float ay, float az, float angle_deg)
{
   const float angle = (float) jtoradians((double) angle_deg);
   quat q;
   float amag = (float) sqrt((double) (ax * ax + ay * ay + az * az));
   if ((double) amag < 0.000001) {
      q.w = q.x = q.y = q.z = 0.0f;
   } else {
      float mag;
      amag = 1.0f / amag;
      mag = (float) sin((double) angle / 2.0);
      q.w = (float) cos((double) angle / 2.0);
      q.x = ax * amag * mag;
      q.y = ay * amag * mag;
      q.z = az * amag * mag;
   }
   return q;
}
/* Quat4f.mul(q1): this = this * q1 */
static quat quat_mul(quat a, quat b)
{
   quat r;
   r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
   r.x = a.w * b.x + b.w * a.x + a.y * b.z - a.z * b.y;
   r.y = a.w * b.y + b.w * a.y - a.x * b.z + a.z * b.x;
   r.z = a.w * b.z + b.w * a.z + a.x * b.y - a.y * b.x;
   return r;
}

/* RTStatics.initializeQuat4f(float[]) RTStatics.java:345-354 + Matrix4f.set(Quat4f) */
void orc_quat_rotation(const float orientation[3], float m[9])
{
   quat qx = quat_axis_angle(0, 1, 0, -orientation[0]);
   quat qy = quat_axis_angle(1, 0, 0, orientation[1]);
   quat qz = quat_axis_angle(0, 0, 1, -orientation[2]);
   quat q = quat_mul(quat_mul(qx, qy), qz);
   m[0] = (1.0f - 2.0f * q.y * q.y - 2.0f * q.z * q.z);
   m[3] = (2.0f * (q.x * q.y + q.w * q.z));
   m[6] = (2.0f * (q.x * q.z - q.w * q.y));
   m[1] = (2.0f * (q.x * q.y - q.w * q.z));
   m[4] = (1.0f - 2.0f * q.x * q.x - 2.0f * q.z * q.z);
   m[7] = (2.0f * (q.y * q.z + q.w * q.x));
   m[2] = (2.0f * (q.x * q.z + q.w * q.y));
   m[5] = (2.0f * (q.y * q.z - q.w * q.x));
   m[8] = (1.0f - 2.0f * q.x * q.x - 2.0f * q.y * q.y);
}

/* Camera ctor Camera.java:40-94 */
void orc_camera_setup(const orc_scene *s, const float orientation[3], float fov, float near_plane, int32_t W, int32_t H,
                      orc_camera *out)
{
   float mm[2][3];
   float width, height, depth, cx, cy, cz, distance;
   float ox, oy, oz;
   const float *m;
   out->near_plane = near_plane;
   out->width = W;
   out->height = H;
   out->viewport_w = (W >= H ? (float) W / (float) H : 1.0f) * near_plane;
   out->viewport_h = (W >= H ? 1.0f : (float) H / (float) W) * near_plane;
   orc_quat_rotation(orientation, out->rotation);
   scene_bounds(s, mm);
   width = mm[1][0] - mm[0][0];
   height = mm[1][1] - mm[0][1];
   depth = mm[1][2] - mm[0][2];
   cx = width / 2.0f + mm[0][0];
   cy = height / 2.0f + mm[0][1];
   cz = depth / 2.0f + mm[0][2];
   distance = width / 2.0f / (float) tan(jtoradians((double) fov));
   ox = cx;
   oy = cy;
   oz = cz + distance;
   m = out->rotation;
   out->origin[0] = m[0] * ox + m[1] * oy + m[2] * oz;
   out->origin[1] = m[3] * ox + m[4] * oy + m[5] * oz;
   out->origin[2] = m[6] * ox + m[7] * oy + m[8] * oz;
}

/* ---------------------------------------------------------------- unit entry points */

int orc_aabb_intersection(const float o[3], const float d[3], const float minmax[6], float near_plane, float far_plane)
{
   ray r;
   float mm[2][3];
   r.origin.x = o[0]; r.origin.y = o[1]; r.origin.z = o[2];
   r.dir.x = d[0]; r.dir.y = d[1]; r.dir.z = d[2];
   memcpy(mm, minmax, sizeof(mm));
   return aabb_intersection(&r, (const float(*)[3]) mm, near_plane, far_plane);
}

int orc_intersects_triangle(const float o[3], const float d[3], const float p[9], const float n[9], float eps, float out[7])
{
   v3 rO = { o[0], o[1], o[2] }, rD = { d[0], d[1], d[2] };
   return intersects_triangle(rO, rD, p, p + 3, p + 6, n, n + 3, n + 6, eps, out);
}

void orc_reflection_direction(const float n[3], const float dir[3], float out[3])
{
   v3 N = { n[0], n[1], n[2] }, D = { dir[0], dir[1], dir[2] };
   v3 r = reflection_direction(N, D);
   out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

int orc_sphere_intersection(const float o[3], const float d[3], const float center[3], float radius, float eps, int bounding, float out[7])
{
   ray r;
   v3 c = { center[0], center[1], center[2] };
   r.origin.x = o[0]; r.origin.y = o[1]; r.origin.z = o[2];
   r.dir.x = d[0]; r.dir.y = d[1]; r.dir.z = d[2];
   if (bounding) return bounding_sphere_intersects(&r, c, radius, eps);
   return sphere_intersection(&r, c, radius, eps, out);
}

int orc_plane_intersection(const float o[3], const float d[3], const float *corners, uint32_t n_corners, float eps, float out[7])
{
   object ob;
   ray r;
   memset(&ob, 0, sizeof(ob));
   if (n_corners < 3 || n_corners > 4 || plane_setup(&ob, corners, n_corners) != 0) return -1;
   r.origin.x = o[0]; r.origin.y = o[1]; r.origin.z = o[2];
   r.dir.x = d[0]; r.dir.y = d[1]; r.dir.z = d[2];
   return plane_intersection(&ob, &r, eps, out);
}

void orc_refraction_direction(const float n[3], const float dir[3], float refraction_index, float out[3])
{
   v3 N = { n[0], n[1], n[2] }, D = { dir[0], dir[1], dir[2] };
   v3 r = refraction_direction(N, D, refraction_index);
   out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void orc_checker_color(const float minmax[6], const float point[3], const orc_material_ext *ext, float out[3])
{
   float mm[2][3];
   v3 p = { point[0], point[1], point[2] };
   memcpy(mm, minmax, sizeof(mm));
   checker_color((const float(*)[3]) mm, p, ext, out);
}

void orc_normalize(float v[3])
{
   v3 a = { v[0], v[1], v[2] };
   a = v3normalize(a);
   v[0] = a.x; v[1] = a.y; v[2] = a.z;
}

/* new Color(bound(0,1,r), bound(0,1,g), bound(0,1,b)).getRGB(): java.awt.Color(float,float,float) rounds each
 * component with (int) (c * 255 + 0.5); RTStatics.bound (RTStatics.java:384-386) */
static uint32_t pack_color(float r, float g, float b)
{
   const float c[3] = { jmin(1.0f, jmax(0.0f, r)), jmin(1.0f, jmax(0.0f, g)), jmin(1.0f, jmax(0.0f, b)) };
   uint32_t out = 0xFF000000u;
   int k;
   for (k = 0; k < 3; k++) out |= ((uint32_t) (int) ((double) (c[k] * 255.0f) + 0.5) & 255u) << (16 - 8 * k);
   return out;
}

/* normalized == 0: the 8-bit image Camera.setPixel keeps (Camera.java:103-107).
 * normalized != 0: Camera.updateNormalizedImage (Camera.java:217-263): min / max of the HSV value over the pixels
 * without a NaN, then value := (value - min) / (max - min), back to RGB, packed.  rgb is row-major W*H*3. */
void orc_image_argb8(const float *rgb, int32_t width, int32_t height, int normalized, uint32_t *out)
{
   const size_t n = (size_t) width * (size_t) height;
   float lo = FLT_MAX, hi = -FLT_MAX;
   size_t i;
   if (normalized) {
      for (i = 0; i < n; i++) {
         const float *c = rgb + 3 * i;
         float hsv[3];
         if (c[0] != c[0] || c[1] != c[1] || c[2] != c[2]) continue;
         rgb_to_hsv(c, hsv);
         lo = jmin(lo, hsv[2]);
         hi = jmax(hi, hsv[2]);
      }
   }
   for (i = 0; i < n; i++) {
      const float *c = rgb + 3 * i;
      if (normalized) {
         float hsv[3], o[3];
         rgb_to_hsv(c, hsv);
         hsv[2] = (hsv[2] - lo) / (hi - lo);
         hsv_to_rgb(hsv, o);
         out[i] = pack_color(o[0], o[1], o[2]);
      } else {
         out[i] = pack_color(c[0], c[1], c[2]);
      }
   }
}

void orc_rgb_to_hsv(const float rgb[3], float hsv[3]) { rgb_to_hsv(rgb, hsv); }
void orc_hsv_to_rgb(const float hsv[3], float rgb[3]) { hsv_to_rgb(hsv, rgb); }
void orc_remap_angles(const float in2[2], const float out2[2], float out3[3]) { remap_angles(in2, out2, out3); }

void orc_remap_vectors(const float normal[3], const float in_dir[3], const float out_dir[3], float out3[3])
{
   v3 n = { normal[0], normal[1], normal[2] };
   v3 i = { in_dir[0], in_dir[1], in_dir[2] };
   v3 o = { out_dir[0], out_dir[1], out_dir[2] };
   remap_vectors(n, normal_tangent(n), i, o, out3);
}

float orc_brdf_luminocity(const orc_scene *s, int32_t brdf_id, int32_t brdf_samples, const float ray_dir[3], const float point[3],
                          const float normal[3], const float light[3])
{
   v3 d = { ray_dir[0], ray_dir[1], ray_dir[2] };
   v3 p = { point[0], point[1], point[2] };
   v3 n = { normal[0], normal[1], normal[2] };
   v3 l = { light[0], light[1], light[2] };
   return brdf_luminocity(s, brdf_id, brdf_samples, d, p, n, l);
}

void orc_phong(const float color_in[3], const float kd[3], const float ks[3], float shininess, const float light_emission[3],
               const float light_ambient[3], const float light_origin[3], const float point[3], const float normal[3],
               const float ray_dir[3], int32_t shadowed, float out[3])
{
   v3 l = { light_origin[0], light_origin[1], light_origin[2] };
   v3 p = { point[0], point[1], point[2] };
   v3 n = { normal[0], normal[1], normal[2] };
   v3 d = { ray_dir[0], ray_dir[1], ray_dir[2] };
   phong_combine(color_in, kd, ks, shininess, light_emission, light_ambient, l, p, n, d, shadowed, out);
}
