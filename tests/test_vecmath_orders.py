"""javax.vecmath 1.5.2's operation orders, re-checked from its bytecode.

The reference's float arithmetic below the hot path is javax.vecmath (lib/vecmath.jar: class files only, SURVEY.md
section 8c); the oracle and tests/np_restatement.py restate its methods from a disassembly.  This test repeats that
disassembly whenever the jar is there (this container; it is skipped where /root/reference does not exist): the class
files are READ as data (tests/classfile_exprs.py: a class-file reader and a symbolic evaluator of straight-line float
bytecode; nothing is loaded into a JVM or run), every method the path calls is turned into the expressions it assigns,
and those are compared (1) with the order of operations written down here, the one the oracle implements
(oracle/rt_oracle.c: v3_dot, v3_cross, v3_normalize, quat_*, camera rows), and (2) numerically, bit for bit, with
tests/np_restatement.py on random inputs.  The last test runs the oracle's own exports (orc_normalize,
orc_quat_rotation) beside them.  CPU only."""
import math
import os
import types
import zipfile

import numpy as np
import pytest

import np_restatement as npr
from classfile_exprs import ClassFile, method_exprs, strip

JAR = "/root/reference/lib/vecmath.jar"
pytestmark = pytest.mark.skipif(not os.path.exists(JAR), reason="the reference's lib/vecmath.jar is not on this machine")

V3 = "Ljavax/vecmath/Vector3f;"
T3 = "Ljavax/vecmath/Tuple3f;"
Q4 = "Ljavax/vecmath/Quat4f;"
AA = "Ljavax/vecmath/AxisAngle4f;"

SUMSQ = "((a.x*a.x)+(a.y*a.y))+(a.z*a.z)"


def _sumsq(obj):
    return SUMSQ.replace("a.", obj + ".")


# (class, method, descriptor, conditionals whose jump is taken) -> assignments in program order, then what it returns
EXPECTED = {
    ("Vector3f", "dot", "(%s)F" % V3, ()): ([], "((this.x*a1.x)+(this.y*a1.y))+(this.z*a1.z)"),
    ("Vector3f", "cross", "(%s%s)V" % (V3, V3), ()): ([
        ("this.z", "(a1.x*a2.y)-(a1.y*a2.x)"),
        ("this.x", "(a1.y*a2.z)-(a1.z*a2.y)"),
        ("this.y", "(a2.x*a1.z)-(a2.z*a1.x)")], None),   # (note the operand order in y)
    ("Vector3f", "normalize", "()V", ()): ([
        ("this.%s" % c, "this.%s*(float)(1.0/Math.sqrt((double)(%s)))" % (c, _sumsq("this"))) for c in "xyz"], None),
    ("Vector3f", "length", "()F", ()): ([], "(float)Math.sqrt((double)(%s))" % _sumsq("this")),
    ("Tuple3f", "sub", "(%s%s)V" % (T3, T3), ()): ([("this.%s" % c, "a1.%s-a2.%s" % (c, c)) for c in "xyz"], None),
    ("Tuple3f", "sub", "(%s)V" % T3, ()): ([("this.%s" % c, "this.%s-a1.%s" % (c, c)) for c in "xyz"], None),
    ("Tuple3f", "add", "(%s%s)V" % (T3, T3), ()): ([("this.%s" % c, "a1.%s+a2.%s" % (c, c)) for c in "xyz"], None),
    ("Tuple3f", "add", "(%s)V" % T3, ()): ([("this.%s" % c, "this.%s+a1.%s" % (c, c)) for c in "xyz"], None),
    ("Tuple3f", "scale", "(F)V", ()): ([("this.%s" % c, "this.%s*a1" % c) for c in "xyz"], None),
    ("Tuple3f", "scale", "(F%s)V" % T3, ()): ([("this.%s" % c, "a1*a2.%s" % c) for c in "xyz"], None),
    ("Tuple3f", "negate", "()V", ()): ([("this.%s" % c, "-this.%s" % c) for c in "xyz"], None),
    # no translation for a Vector3f; x and y are worked out before z is stored
    ("Matrix4f", "transform", "(%s)V" % V3, ()): ([
        ("a1.z", "((this.m20*a1.x)+(this.m21*a1.y))+(this.m22*a1.z)"),
        ("a1.x", "((this.m00*a1.x)+(this.m01*a1.y))+(this.m02*a1.z)"),
        ("a1.y", "((this.m10*a1.x)+(this.m11*a1.y))+(this.m12*a1.z)")], None),
    ("Matrix4f", "set", "(%s)V" % Q4, ()): ([
        ("this.m00", "(1.0f-((2.0f*a1.y)*a1.y))-((2.0f*a1.z)*a1.z)"),
        ("this.m10", "2.0f*((a1.x*a1.y)+(a1.w*a1.z))"),
        ("this.m20", "2.0f*((a1.x*a1.z)-(a1.w*a1.y))"),
        ("this.m01", "2.0f*((a1.x*a1.y)-(a1.w*a1.z))"),
        ("this.m11", "(1.0f-((2.0f*a1.x)*a1.x))-((2.0f*a1.z)*a1.z)"),
        ("this.m21", "2.0f*((a1.y*a1.z)+(a1.w*a1.x))"),
        ("this.m02", "2.0f*((a1.x*a1.z)+(a1.w*a1.y))"),
        ("this.m12", "2.0f*((a1.y*a1.z)-(a1.w*a1.x))"),
        ("this.m22", "(1.0f-((2.0f*a1.x)*a1.x))-((2.0f*a1.y)*a1.y)"),
        ("this.m03", "0.0f"), ("this.m13", "0.0f"), ("this.m23", "0.0f"),
        ("this.m30", "0.0f"), ("this.m31", "0.0f"), ("this.m32", "0.0f"), ("this.m33", "1.0f")], None),
    ("Quat4f", "mul", "(%s)V" % Q4, ()): ([
        ("this.z", "(((this.w*a1.z)+(a1.w*this.z))+(this.x*a1.y))-(this.y*a1.x)"),
        ("this.w", "(((this.w*a1.w)-(this.x*a1.x))-(this.y*a1.y))-(this.z*a1.z)"),
        ("this.x", "(((this.w*a1.x)+(a1.w*this.x))+(this.y*a1.z))-(this.z*a1.y)"),
        ("this.y", "(((this.w*a1.y)+(a1.w*this.y))-(this.x*a1.z))+(this.z*a1.x)")], None),
    # the axis too short: the zero quaternion ...
    ("Quat4f", "set", "(%s)V" % AA, ()): ([
        ("if", "cmp((double)(float)Math.sqrt((double)(%s)),1e-06)>=0 not taken" % _sumsq("a1")),
        ("this.w", "0.0f"), ("this.x", "0.0f"), ("this.y", "0.0f"), ("this.z", "0.0f")], None),
    # ... else 1 / |axis| in f32, the half angle in f64, sin and cos narrowed to f32
    ("Quat4f", "set", "(%s)V" % AA, (0,)): ([
        ("if", "cmp((double)(float)Math.sqrt((double)(%s)),1e-06)>=0 taken" % _sumsq("a1")),
        ("this.w", "(float)Math.cos(((double)a1.angle/2.0))")] + [
        ("this.%s" % c, "(a1.%s*(1.0f/(float)Math.sqrt((double)(%s))))*(float)Math.sin(((double)a1.angle/2.0))" % (c, _sumsq("a1"))) for c in "xyz"], None),
}


@pytest.fixture(scope="module")
def classes():
    with zipfile.ZipFile(JAR) as z:
        return {n: ClassFile(z.read("javax/vecmath/%s.class" % n)) for n in ("Vector3f", "Tuple3f", "Matrix4f", "Quat4f")}


def test_the_jar_is_the_version_the_survey_names():
    with zipfile.ZipFile(JAR) as z:
        manifest = z.read("META-INF/MANIFEST.MF").decode("utf-8", "replace")
    assert "Implementation-Version: 1.5.2" in manifest


@pytest.mark.parametrize("key", sorted(EXPECTED, key=str), ids=lambda k: "%s.%s%s%s" % (k[0], k[1], k[2].replace("Ljavax/vecmath/", "").replace(";", ""), "+jump" if k[3] else ""))
def test_operation_order_is_the_restated_one(classes, key):
    cls, name, desc, take = key
    out, ret = method_exprs(classes[cls], name, desc, take=take)
    want_out, want_ret = EXPECTED[key]
    assert [(t, strip(e)) for t, e in out] == want_out
    assert (strip(ret) if ret is not None else None) == want_ret


# ---- the same expressions evaluated in numpy scalars against the second restatement

def _env(**objs):
    env = {"f32": np.float32, "f64": np.float64,
           "Math_sqrt": lambda x: np.sqrt(np.float64(x)), "Math_sin": lambda x: np.float64(math.sin(float(x))),
           "Math_cos": lambda x: np.float64(math.cos(float(x)))}
    env.update(objs)
    return env


def _obj(**fields):
    return types.SimpleNamespace(**{k: np.float32(v) for k, v in fields.items()})


def _run(cf, name, desc, take=(), **objs):
    """the method's assignments evaluated on the given objects: {target: value}, and the returned value"""
    out, ret = method_exprs(cf, name, desc, take=take)
    env = _env(**objs)
    with np.errstate(all="ignore"):
        vals = {t: eval(e.py, {"__builtins__": {}}, env) for t, e in out if t != "if"}   # (expressions built by classfile_exprs from opcodes)
        r = eval(ret.py, {"__builtins__": {}}, env) if ret is not None else None
    return vals, r


def _bits(x):
    return np.float32(x).view(np.uint32)


def test_vector_methods_equal_the_numpy_restatement_bit_for_bit(classes):
    rng = np.random.default_rng(20241111)
    V, T = classes["Vector3f"], classes["Tuple3f"]
    for _ in range(200):
        a = (rng.standard_normal(3) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        b = (rng.standard_normal(3) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        A, B = _obj(x=a[0], y=a[1], z=a[2]), _obj(x=b[0], y=b[1], z=b[2])
        _, r = _run(V, "dot", "(%s)F" % V3, this=A, a1=B)
        assert _bits(r) == _bits(npr.vdot(a, b))
        vals, _ = _run(V, "cross", "(%s%s)V" % (V3, V3), this=_obj(x=0, y=0, z=0), a1=A, a2=B)
        want = npr.vcross(a, b)
        assert [_bits(vals["this." + c]) for c in "xyz"] == [_bits(w) for w in want]
        vals, _ = _run(V, "normalize", "()V", this=A)
        want = npr.vnormalize(a)
        assert [_bits(vals["this." + c]) for c in "xyz"] == [_bits(w) for w in want]
        vals, _ = _run(T, "sub", "(%s%s)V" % (T3, T3), this=_obj(x=0, y=0, z=0), a1=A, a2=B)
        assert [_bits(vals["this." + c]) for c in "xyz"] == [_bits(w) for w in npr.vsub(a, b)]


def test_camera_rotation_equals_the_numpy_restatement_bit_for_bit(classes):
    """RTStatics.initializeQuat4f (RTStatics.java:345-361) = three Quat4f.set(AxisAngle4f), two Quat4f.mul, then
    Matrix4f.set(Quat4f) (Camera.java:56-57): the bytecode's expressions, chained, against np_restatement.rotation_rows."""
    rng = np.random.default_rng(7)
    Q, M = classes["Quat4f"], classes["Matrix4f"]

    def axis_angle(axis, angle_deg):
        angle = np.float32(np.float64(np.float32(angle_deg)) / 180.0 * math.pi)   # RTStatics.java:356-361
        vals, _ = _run(Q, "set", "(%s)V" % AA, take=(0,), this=_obj(x=0, y=0, z=0, w=0), a1=_obj(x=axis[0], y=axis[1], z=axis[2], angle=angle))
        return _obj(**{c: vals["this." + c] for c in "xyzw"})

    def mul(a, b):
        vals, _ = _run(Q, "mul", "(%s)V" % Q4, this=a, a1=b)
        return _obj(**{c: vals["this." + c] for c in "xyzw"})

    for _ in range(100):
        o = (rng.uniform(-180.0, 180.0, 3)).astype(np.float32)
        q = mul(mul(axis_angle((0, 1, 0), -o[0]), axis_angle((1, 0, 0), o[1])), axis_angle((0, 0, 1), -o[2]))
        vals, _ = _run(M, "set", "(%s)V" % Q4, this=types.SimpleNamespace(), a1=q)
        got = [vals["this.m%d%d" % (r, c)] for r in range(3) for c in range(3)]
        want = npr.rotation_rows(o)
        assert [_bits(g) for g in got] == [_bits(w) for w in want]
        for c, v in zip("xyzw", npr.quat_axis_angle([np.float32(0), np.float32(1), np.float32(0)], -o[0])):
            assert _bits(getattr(axis_angle((0, 1, 0), -o[0]), c)) == _bits(v)


def test_the_oracle_itself_equals_the_bytecode_bit_for_bit(classes, orc):
    """oracle/rt_oracle.c's v3_normalize and its camera rotation (orc_normalize, orc_quat_rotation) against the expressions
    read from the class files -- the C restatement tied to the jar directly, not only through the numpy one."""
    rng = np.random.default_rng(99)
    V, Q, M = classes["Vector3f"], classes["Quat4f"], classes["Matrix4f"]
    for _ in range(200):
        a = (rng.standard_normal(3) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        vals, _ = _run(V, "normalize", "()V", this=_obj(x=a[0], y=a[1], z=a[2]))
        v = a.copy()
        orc.lib().orc_normalize(orc.P(v))
        assert [_bits(vals["this." + c]) for c in "xyz"] == [_bits(x) for x in v]

    def axis_angle(axis, angle_deg):
        angle = np.float32(np.float64(np.float32(angle_deg)) / 180.0 * math.pi)
        vals, _ = _run(Q, "set", "(%s)V" % AA, take=(0,), this=_obj(x=0, y=0, z=0, w=0), a1=_obj(x=axis[0], y=axis[1], z=axis[2], angle=angle))
        return _obj(**{c: vals["this." + c] for c in "xyzw"})

    def mul(a, b):
        vals, _ = _run(Q, "mul", "(%s)V" % Q4, this=a, a1=b)
        return _obj(**{c: vals["this." + c] for c in "xyzw"})

    for _ in range(100):
        o = (rng.uniform(-180.0, 180.0, 3)).astype(np.float32)
        q = mul(mul(axis_angle((0, 1, 0), -o[0]), axis_angle((1, 0, 0), o[1])), axis_angle((0, 0, 1), -o[2]))
        vals, _ = _run(M, "set", "(%s)V" % Q4, this=types.SimpleNamespace(), a1=q)
        got = [vals["this.m%d%d" % (r, c)] for r in range(3) for c in range(3)]
        m = np.zeros(9, dtype=np.float32)
        orc.lib().orc_quat_rotation(orc.P(o), orc.P(m))
        assert [_bits(g) for g in got] == [_bits(x) for x in m]
