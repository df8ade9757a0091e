"""A small reader of Java class files (JVMS chapter 4) and a symbolic evaluator of the straight-line float arithmetic in a
method's bytecode (JVMS chapter 6): test infrastructure for tests/test_vecmath_orders.py.

It READS class files as data -- nothing is loaded into a JVM or run -- and turns a method into the expressions it assigns
or returns, fully parenthesised in evaluation order, e.g. Vector3f.dot -> "((this.x*a1.x)+(this.y*a1.y))+(this.z*a1.z)".
Only what javax.vecmath's float methods use is implemented; anything else raises NotImplementedError."""
import struct


class ClassFile:
    def __init__(self, data):
        self.d = data
        self.p = 0
        if self.u4() != 0xCAFEBABE:
            raise ValueError("not a class file")
        self.u2(), self.u2()   # minor, major
        n = self.u2()
        self.cp = [None] * n
        i = 1
        while i < n:
            tag = self.u1()
            if tag == 1:
                ln = self.u2()
                self.cp[i] = ("utf8", self.d[self.p:self.p + ln].decode("utf-8", "replace"))
                self.p += ln
            elif tag == 3:
                self.cp[i] = ("int", struct.unpack(">i", self.take(4))[0])
            elif tag == 4:
                self.cp[i] = ("float", struct.unpack(">f", self.take(4))[0])
            elif tag == 5:
                self.cp[i] = ("long", struct.unpack(">q", self.take(8))[0])
                i += 1
            elif tag == 6:
                self.cp[i] = ("double", struct.unpack(">d", self.take(8))[0])
                i += 1
            elif tag in (7, 8, 16, 19, 20):
                self.cp[i] = ({7: "class", 8: "string", 16: "mtype", 19: "module", 20: "package"}[tag], self.u2())
            elif tag in (9, 10, 11, 12, 17, 18):
                self.cp[i] = ({9: "field", 10: "method", 11: "imethod", 12: "nat", 17: "dynamic", 18: "indy"}[tag], self.u2(), self.u2())
            elif tag == 15:
                self.cp[i] = ("mhandle", self.u1(), self.u2())
            else:
                raise ValueError("constant pool tag %d" % tag)
            i += 1
        self.u2()   # access
        self.this_class = self.class_name(self.u2())
        self.u2()   # super
        for _ in range(self.u2()):
            self.u2()
        for _ in range(self.u2()):   # fields
            self.u2(), self.u2(), self.u2()
            self.skip_attributes()
        self.methods = {}   # (name, descriptor) -> (access, code bytes)
        for _ in range(self.u2()):
            access, name, desc = self.u2(), self.utf8(self.u2()), self.utf8(self.u2())
            code = None
            for _ in range(self.u2()):
                aname, alen = self.utf8(self.u2()), self.u4()
                body = self.take(alen)
                if aname == "Code":
                    clen = struct.unpack(">I", body[4:8])[0]
                    code = body[8:8 + clen]
            self.methods[(name, desc)] = (access, code)

    def take(self, n):
        b = self.d[self.p:self.p + n]
        self.p += n
        return b

    def u1(self):
        return self.take(1)[0]

    def u2(self):
        return struct.unpack(">H", self.take(2))[0]

    def u4(self):
        return struct.unpack(">I", self.take(4))[0]

    def skip_attributes(self):
        for _ in range(self.u2()):
            self.u2()
            self.take(self.u4())

    def utf8(self, i):
        return self.cp[i][1]

    def class_name(self, i):
        return self.utf8(self.cp[i][1])

    def member(self, i):
        """(class, name, descriptor) of a Fieldref / Methodref"""
        _, c, nat = self.cp[i]
        return self.class_name(c), self.utf8(self.cp[nat][1]), self.utf8(self.cp[nat][2])


def _arg_slots(desc):
    """slot width of each argument of a method descriptor"""
    out, i = [], 1
    while desc[i] != ")":
        if desc[i] in "DJ":
            out.append(2)
            i += 1
        elif desc[i] == "L":
            out.append(1)
            i = desc.index(";", i) + 1
        elif desc[i] == "[":
            while desc[i] == "[":
                i += 1
            i = desc.index(";", i) + 1 if desc[i] == "L" else i + 1
            out.append(1)
        else:
            out.append(1)
            i += 1
    return out


def _num(v):
    return repr(float(v))


class Sym(str):
    """an expression as text (Java-like, fully parenthesised); .py is the same expression as Python over numpy scalars
    (f32 / f64 round every operation the way the JVM's fadd / dadd ... do)"""
    def __new__(cls, text, py=None):
        o = super().__new__(cls, text)
        o.py = py if py is not None else text
        return o


def method_exprs(cf, name, desc, take=()):
    """Symbolic run of a method without loops: returns (assignments, returned), assignments = [(target, expression)] in
    program order for every putfield, returned = the expression of an *return (None for void).  Arguments are a1, a2, ...;
    a field read is what the method last stored there, else "<object>.<field>".  A forward conditional is noted as
    ("if", condition) and followed along its fall-through path, or along the jump for the conditionals whose index
    (0, 1, ... in the order met) is in `take`."""
    access, code = cf.methods[(name, desc)]
    static = bool(access & 0x0008)
    loc = {}
    slot = 0
    if not static:
        loc[0] = Sym("this")
        slot = 1
    for k, w in enumerate(_arg_slots(desc)):
        loc[slot] = Sym("a%d" % (k + 1))
        slot += w
    fields = {}   # (object expr, field) -> stored expression
    stack, out, ret = [], [], None
    pc = 0
    n_if = 0
    binop = {0x62: ("+", "f32"), 0x63: ("+", "f64"), 0x66: ("-", "f32"), 0x67: ("-", "f64"), 0x6a: ("*", "f32"), 0x6b: ("*", "f64"),
             0x6e: ("/", "f32"), 0x6f: ("/", "f64")}
    while pc < len(code):
        op = code[pc]
        if 0x2a <= op <= 0x2d:      # aload_n
            stack.append(loc[op - 0x2a])
        elif 0x22 <= op <= 0x25:    # fload_n
            stack.append(loc[op - 0x22])
        elif 0x26 <= op <= 0x29:    # dload_n
            stack.append(loc[op - 0x26])
        elif op in (0x19, 0x17, 0x18):   # aload, fload, dload
            pc += 1
            stack.append(loc[code[pc]])
        elif 0x43 <= op <= 0x46:    # fstore_n
            loc[op - 0x43] = stack.pop()
        elif 0x47 <= op <= 0x4a:    # dstore_n
            loc[op - 0x47] = stack.pop()
        elif op in (0x38, 0x39):    # fstore, dstore
            pc += 1
            loc[code[pc]] = stack.pop()
        elif op == 0xb4:            # getfield
            _, fname, _ = cf.member(struct.unpack(">H", code[pc + 1:pc + 3])[0])
            pc += 2
            obj = stack.pop()
            stack.append(fields.get((str(obj), fname), Sym("%s.%s" % (obj, fname))))
        elif op == 0xb5:            # putfield
            _, fname, _ = cf.member(struct.unpack(">H", code[pc + 1:pc + 3])[0])
            pc += 2
            val, obj = stack.pop(), stack.pop()
            fields[(str(obj), fname)] = val
            out.append(("%s.%s" % (obj, fname), val))
        elif op in binop:
            b, a = stack.pop(), stack.pop()
            sym, rnd = binop[op]
            stack.append(Sym("(%s%s%s)" % (a, sym, b), "%s(%s%s%s)" % (rnd, a.py, sym, b.py)))
        elif op in (0x76, 0x77):    # fneg, dneg
            a = stack.pop()
            stack.append(Sym("(-%s)" % a, "(-%s)" % a.py))
        elif op == 0x8d:            # f2d
            a = stack.pop()
            stack.append(Sym("(double)%s" % a, "f64(%s)" % a.py))
        elif op == 0x90:            # d2f
            a = stack.pop()
            stack.append(Sym("(float)%s" % a, "f32(%s)" % a.py))
        elif 0x0b <= op <= 0x0d:    # fconst_n
            stack.append(Sym(_num(op - 0x0b) + "f", "f32(%s)" % _num(op - 0x0b)))
        elif op in (0x0e, 0x0f):    # dconst_n
            stack.append(Sym(_num(op - 0x0e), "f64(%s)" % _num(op - 0x0e)))
        elif op in (0x12, 0x13, 0x14):   # ldc, ldc_w, ldc2_w
            if op == 0x12:
                pc += 1
                idx = code[pc]
            else:
                idx = struct.unpack(">H", code[pc + 1:pc + 3])[0]
                pc += 2
            kind, val = cf.cp[idx][0], cf.cp[idx][1]
            stack.append(Sym(_num(val) + ("f" if kind == "float" else ""), "%s(%s)" % ("f32" if kind == "float" else "f64", _num(val))))
        elif op == 0xb8:            # invokestatic
            cls, mname, mdesc = cf.member(struct.unpack(">H", code[pc + 1:pc + 3])[0])
            pc += 2
            n = len(_arg_slots(mdesc))
            args = [stack.pop() for _ in range(n)][::-1]
            short = cls.split("/")[-1]
            stack.append(Sym("%s.%s(%s)" % (short, mname, ",".join(args)), "%s_%s(%s)" % (short, mname, ",".join(a.py for a in args))))
        elif op == 0x59:            # dup
            stack.append(stack[-1])
        elif op in (0x95, 0x96, 0x97, 0x98):   # fcmpl, fcmpg, dcmpl, dcmpg
            b, a = stack.pop(), stack.pop()
            stack.append(Sym("cmp(%s,%s)" % (a, b)))
        elif 0x99 <= op <= 0x9e:    # if<cond>
            cond = ["==0", "!=0", "<0", ">=0", ">0", "<=0"][op - 0x99]
            jump = n_if in take
            out.append(("if", "%s%s %s" % (stack.pop(), cond, "taken" if jump else "not taken")))
            n_if += 1
            if jump:
                off = struct.unpack(">h", code[pc + 1:pc + 3])[0]
                if off <= 0:
                    raise NotImplementedError("loop in %s%s" % (name, desc))
                pc += off
                continue
            pc += 2
        elif op == 0xa7:            # goto: a forward jump ends a conditional block; follow it
            off = struct.unpack(">h", code[pc + 1:pc + 3])[0]
            if off <= 0:
                raise NotImplementedError("loop in %s%s" % (name, desc))
            pc += off
            continue
        elif op in (0xae, 0xaf):    # freturn, dreturn
            ret = stack.pop()
            break
        elif op == 0xb1:            # return
            break
        else:
            raise NotImplementedError("opcode 0x%02x at %d in %s%s" % (op, pc, name, desc))
        pc += 1
    return out, ret


def strip(expr):
    """the expression's text without one pair of outer parentheses"""
    expr = str(expr)
    if expr.startswith("(") and expr.endswith(")"):
        depth = 0
        for i, ch in enumerate(expr):
            depth += ch == "("
            depth -= ch == ")"
            if depth == 0 and i < len(expr) - 1:
                return expr
        return expr[1:-1]
    return expr
