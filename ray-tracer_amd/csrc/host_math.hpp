// host_math.hpp -- Java / javax.vecmath arithmetic used by the host side of libmi355rt
// (camera set-up, BRDF table remap, normal synthesis, hierarchy build).
// Compiled with -ffp-contract=off: Java never fuses a*b+c.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace mi355rt {
namespace jm {

constexpr float kFloatMax = 3.4028234663852886e38f;
constexpr double kPi = 3.14159265358979323846;

// java.lang.Math.min / max on floats: a NaN operand wins, and -0.0f orders below +0.0f.
inline float fmin_java(float a, float b)
{
   if (std::isnan(a)) return a;
   if (std::isnan(b)) return b;
   if (a == b) return std::signbit(a) ? a : b;   // equal values: prefer the negative zero
   return a < b ? a : b;
}
inline float fmax_java(float a, float b)
{
   if (std::isnan(a)) return a;
   if (std::isnan(b)) return b;
   if (a == b) return std::signbit(a) ? b : a;   // equal values: prefer the positive zero
   return a > b ? a : b;
}

// Math.toRadians (JDK <= 8 formula; see DESIGN.md "camera constants")
inline double to_radians(double deg) { return deg / 180.0 * kPi; }

struct Vec3 {
   float x, y, z;
};
inline Vec3 operator-(Vec3 a, Vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// Vector3f.cross (vecmath 1.5.2): y = b.x*a.z - b.z*a.x
inline Vec3 cross(Vec3 a, Vec3 b) { return { a.y * b.z - a.z * b.y, b.x * a.z - b.z * a.x, a.x * b.y - a.y * b.x }; }
// Vector3f.normalize: scale by (float)(1.0 / sqrt((double) len2))
inline Vec3 normalized(Vec3 a)
{
   const float len2 = a.x * a.x + a.y * a.y + a.z * a.z;
   const float inv = (float) (1.0 / std::sqrt((double) len2));
   return { a.x * inv, a.y * inv, a.z * inv };
}

struct Quat {
   float x, y, z, w;
};
// Quat4f.set(AxisAngle4f) for the axis/angle RTStatics.initializeQuat4f builds (RTStatics.java:356-361)
inline Quat quat_from_axis_angle(Vec3 axis, float angle_deg)
{
   const float angle = (float) to_radians((double) angle_deg);
   Quat q {};
   float amag = (float) std::sqrt((double) (axis.x * axis.x + axis.y * axis.y + axis.z * axis.z));
   if ((double) amag < 0.000001) return Quat { 0.f, 0.f, 0.f, 0.f };
   amag = 1.0f / amag;
   const float mag = (float) std::sin((double) angle / 2.0);
   q.w = (float) std::cos((double) angle / 2.0);
   q.x = axis.x * amag * mag;
   q.y = axis.y * amag * mag;
   q.z = axis.z * amag * mag;
   return q;
}
// Quat4f.mul(q1)
inline Quat quat_mul(Quat a, Quat b)
{
   Quat r;
   r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
   r.x = a.w * b.x + b.w * a.x + a.y * b.z - a.z * b.y;
   r.y = a.w * b.y + b.w * a.y - a.x * b.z + a.z * b.x;
   r.z = a.w * b.z + b.w * a.z + a.x * b.y - a.y * b.x;
   return r;
}
// Matrix4f.set(Quat4f), upper 3x3, row-major
inline void quat_to_rows(Quat q, float m[9])
{
   m[0] = (1.0f - 2.0f * q.y * q.y - 2.0f * q.z * q.z);
   m[1] = (2.0f * (q.x * q.y - q.w * q.z));
   m[2] = (2.0f * (q.x * q.z + q.w * q.y));
   m[3] = (2.0f * (q.x * q.y + q.w * q.z));
   m[4] = (1.0f - 2.0f * q.x * q.x - 2.0f * q.z * q.z);
   m[5] = (2.0f * (q.y * q.z - q.w * q.x));
   m[6] = (2.0f * (q.x * q.z - q.w * q.y));
   m[7] = (2.0f * (q.y * q.z + q.w * q.x));
   m[8] = (1.0f - 2.0f * q.x * q.x - 2.0f * q.y * q.y);
}
// Matrix4f.transform(Vector3f): rotation part only
inline Vec3 rotate(const float m[9], Vec3 v)
{
   return { m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z };
}

}   // namespace jm
}   // namespace mi355rt
