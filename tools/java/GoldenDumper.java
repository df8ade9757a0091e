// GoldenDumper.java -- for a maintainer with a JDK: prints what the REFERENCE ITSELF computes for a fixed list of rays, so
// that the oracle of this repository (oracle/rt_oracle.c, a restatement read from the same source text) can be pinned
// against the real program.  It cannot be compiled or run in this pipeline (no JDK) and nothing depends on it:
// tests/test_java_golden.py compares the oracle with tests/golden/java_golden.txt when that file exists and is skipped
// otherwise.
//
//   cd <sranger/Ray-Tracer checkout>
//   javac -cp lib/vecmath.jar -d bin src/stephen/ranger/ar/*.java src/stephen/ranger/ar/*/*.java <this repo>/tools/java/GoldenDumper.java
//   java -cp bin:lib/vecmath.jar GoldenDumper <this repo>/tests/golden/java_inputs/mesh.ply \
//        <this repo>/tests/golden/java_inputs/rays.txt > <this repo>/tests/golden/java_golden.txt
//
// Output, one line per item, every float as the 8 hex digits of Float.floatToIntBits (so nothing is lost in printing):
//   epsilon <bits>
//   camera <origin xyz bits> <rotation m00 m01 m02 m10 ... m22 bits>       (Camera ctor, Camera.java:40-94, 64 x 48, fov 15)
//   ray <index> miss
//   ray <index> hit <w> <point xyz> <normal xyz> <rgb of PhongLightingModel.getPixelColor(info, 0)>
import java.awt.Color;
import java.io.BufferedReader;
import java.io.File;
import java.io.FileReader;

import javax.vecmath.Vector3f;

import stephen.ranger.ar.Camera;
import stephen.ranger.ar.IntersectionInformation;
import stephen.ranger.ar.RTStatics;
import stephen.ranger.ar.Scene;
import stephen.ranger.ar.bounds.BoundingVolume;
import stephen.ranger.ar.lighting.Light;
import stephen.ranger.ar.lighting.PhongLightingModel;
import stephen.ranger.ar.materials.ColorInformation;
import stephen.ranger.ar.sceneObjects.TriangleMesh;

public class GoldenDumper {
   static String h(final float f) {
      return String.format("%08x", Float.floatToIntBits(f));
   }

   public static void main(final String[] args) throws Exception {
      final java.io.PrintStream out = System.out;
      System.setOut(System.err);   // the reference prints its own progress lines; keep them out of the result
      final BoundingVolume[] volumes = new BoundingVolume[] {
            new TriangleMesh(new File(args[0]), new ColorInformation(new Color(0.9f, 0.9f, 0.9f, 1f)), true).getBoundingVolume() };
      final Light light = new Light(new Vector3f(0, 100, 100), new Color(0.3f, 0.3f, 0.3f, 1.0f), new Color(0.5f, 0.5f, 0.9f, 1.0f));
      final PhongLightingModel model = new PhongLightingModel(light, volumes);
      final Scene scene = new Scene(volumes, light, new float[] { 0, 0, 0 }, model, 15f);
      final Camera camera = new Camera(scene, 1, 1, RTStatics.NEAR_PLANE, 64, 48);
      out.println("epsilon " + h(RTStatics.EPSILON));
      final StringBuilder c = new StringBuilder("camera");
      c.append(' ').append(h(camera.origin.x)).append(' ').append(h(camera.origin.y)).append(' ').append(h(camera.origin.z));
      final float[] m = { camera.rotation.m00, camera.rotation.m01, camera.rotation.m02, camera.rotation.m10, camera.rotation.m11,
            camera.rotation.m12, camera.rotation.m20, camera.rotation.m21, camera.rotation.m22 };
      for (final float v : m) {
         c.append(' ').append(h(v));
      }
      out.println(c);
      final BufferedReader rays = new BufferedReader(new FileReader(args[1]));
      String line;
      int index = 0;
      while ((line = rays.readLine()) != null) {
         final String[] t = line.trim().split(" ");
         if (t.length < 6) {
            continue;
         }
         // the ray file holds the bit patterns too: the inputs are the same floats on both sides
         final float[] f = new float[6];
         for (int i = 0; i < 6; i++) {
            f[i] = Float.intBitsToFloat((int) Long.parseLong(t[i], 16));
         }
         final IntersectionInformation info = camera.getClosestIntersection(null, new Vector3f(f[0], f[1], f[2]), new Vector3f(f[3], f[4], f[5]), null, 0);
         if (info == null) {
            out.println("ray " + index + " miss");
         } else {
            final float[] rgb = model.getPixelColor(info, 0);
            out.println("ray " + index + " hit " + h(info.w) + ' ' + h(info.intersection.x) + ' ' + h(info.intersection.y) + ' ' + h(info.intersection.z) + ' '
                  + h(info.normal.x) + ' ' + h(info.normal.y) + ' ' + h(info.normal.z) + ' ' + h(rgb[0]) + ' ' + h(rgb[1]) + ' ' + h(rgb[2]));
         }
         index++;
      }
      rays.close();
   }
}
