/*
 * qe_oracle.c -- CPU ORACLE (test infrastructure, see qe_oracle.h).
 *
 * Row-at-a-time, boxed-value restatement of the reference evaluator.  It is
 * deliberately shaped like the reference (tree walk per row, nullable boxed
 * values, one reused scan row buffer), not like the GPU engine: it shares no
 * code, no program encoding and no data layout with libqe_hip.so.
 */
#include "qe_oracle.h"

#include <math.h>
#include <string.h>

/* ---------- JDK primitives ------------------------------------------------ */

/* java.lang.Double.doubleToLongBits: all NaNs collapse to 0x7ff8000000000000 */
static int64_t double_to_long_bits(double d) {
    if (d != d) return (int64_t)0x7ff8000000000000LL;
    int64_t b;
    memcpy(&b, &d, sizeof b);
    return b;
}

/* java.lang.Double.compare(double,double): total order, -0.0 < 0.0, NaN greatest */
int32_t qo_double_compare(double a, double b) {
    if (a < b) return -1;
    if (a > b) return 1;
    int64_t x = double_to_long_bits(a), y = double_to_long_bits(b);
    return x == y ? 0 : (x < y ? -1 : 1);
}

/* java.lang.Double.equals */
int32_t qo_double_equals(double a, double b) { return double_to_long_bits(a) == double_to_long_bits(b); }

/* java.lang.String.compareTo (Interpreter.kt:104-107, BytecodeCompiler.kt:303): lexicographic order of the UTF-16 code
 * units.  UTF-8 byte order agrees with it except that a supplementary character (a surrogate pair 0xD800.. in UTF-16)
 * sorts BEFORE U+E000..U+FFFF; next_unit() walks a UTF-8 string one UTF-16 unit at a time. */
typedef struct { const unsigned char *p; uint32_t pending; } u16_iter;
static int next_unit(u16_iter *it, uint32_t *unit) {
    if (it->pending) { *unit = it->pending; it->pending = 0; return 1; }
    const unsigned char *p = it->p;
    if (!*p) return 0;
    uint32_t c = p[0], cp = c;
    int extra = c < 0x80 ? 0 : (c >> 5) == 0x6 ? 1 : (c >> 4) == 0xe ? 2 : (c >> 3) == 0x1e ? 3 : -1, ok = extra >= 0;
    for (int k = 1; ok && k <= extra; k++) ok = (p[k] & 0xc0) == 0x80;   /* a NUL ends the check: never reads past the end */
    if (ok) {
        if (extra > 0) cp = c & (0xffu >> (extra + 2));
        for (int k = 1; k <= extra; k++) cp = (cp << 6) | (p[k] & 0x3f);
        it->p += extra + 1;
    } else {
        it->p += 1;
    }
    if (cp >= 0x10000 && cp <= 0x10ffff) {
        cp -= 0x10000;
        *unit = 0xd800 + (cp >> 10);
        it->pending = 0xdc00 + (cp & 0x3ff);
    } else {
        *unit = cp & 0xffff;
    }
    return 1;
}
static int32_t string_compare(const char *a, const char *b) {
    u16_iter x = {(const unsigned char *)a, 0}, y = {(const unsigned char *)b, 0};
    for (;;) {
        uint32_t ux = 0, uy = 0;
        const int hx = next_unit(&x, &ux), hy = next_unit(&y, &uy);
        if (!hx || !hy) return hx == hy ? 0 : (hx ? 1 : -1);
        if (ux != uy) return ux < uy ? -1 : 1;
    }
}

/* Math.min / Math.max (Accumulators.kt:62,80): NaN wins, -0.0 < 0.0 */
static double java_min(double a, double b) {
    if (a != a) return a;
    if (b != b) return b;
    if (a == 0.0 && b == 0.0) return signbit(a) ? a : b;
    return a < b ? a : b;
}
static double java_max(double a, double b) {
    if (a != a) return a;
    if (b != b) return b;
    if (a == 0.0 && b == 0.0) return signbit(a) ? b : a;
    return a > b ? a : b;
}

static uint64_t mix64_fwd(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

/* ---------- boxed values --------------------------------------------------- */

static qo_value v_null(void) { qo_value v; memset(&v, 0, sizeof v); v.tag = QO_T_NULL; return v; }
static qo_value v_f64(double d) { qo_value v = v_null(); v.tag = QO_T_F64; v.u.d = d; return v; }
static qo_value v_bool(int b) { qo_value v = v_null(); v.tag = QO_T_BOOL; v.u.b = b ? 1 : 0; return v; }
static qo_value v_str(const char *s) { qo_value v = v_null(); v.tag = QO_T_STR; v.u.s = s; return v; }
static qo_value v_i64(int64_t l) { qo_value v = v_null(); v.tag = QO_T_I64; v.u.l = l; return v; }
static qo_value v_i32(int32_t i) { qo_value v = v_null(); v.tag = QO_T_I32; v.u.i = i; return v; }

static int is_numeric(const qo_value *v) { return v->tag == QO_T_F64 || v->tag == QO_T_I64 || v->tag == QO_T_I32; }

/* binary numeric promotion (JLS 5.6.2) for the INT64/INT32 extension */
static int promoted_tag(const qo_value *a, const qo_value *b) {
    if (a->tag == QO_T_F64 || b->tag == QO_T_F64) return QO_T_F64;
    if (a->tag == QO_T_I64 || b->tag == QO_T_I64) return QO_T_I64;
    return QO_T_I32;
}
static double as_f64(const qo_value *v) {
    return v->tag == QO_T_F64 ? v->u.d : (v->tag == QO_T_I64 ? (double)v->u.l : (double)v->u.i);
}
static int64_t as_i64(const qo_value *v) { return v->tag == QO_T_I64 ? v->u.l : (int64_t)v->u.i; }

/* ---------- arithmetic ------------------------------------------------------ */

/* Interpreter.kt:94-100 / BytecodeCompiler.kt:328-345 (DADD DSUB DMUL DDIV DREM DNEG).
 * Extension: long/int operands use Java two's-complement wrap; integer
 * division or remainder by zero yields NULL (build decision). */
static int32_t arith(int fn, const qo_value *a, const qo_value *b, qo_value *out) {
    if (!is_numeric(a) || (b && !is_numeric(b))) return QO_THROWN; /* ClassCastException */
    if (fn == QO_UNARY_PLUS) { *out = *a; return QO_OK; }
    if (fn == QO_UNARY_MINUS) {
        if (a->tag == QO_T_F64) *out = v_f64(-a->u.d);
        else if (a->tag == QO_T_I64) *out = v_i64((int64_t)(0ULL - (uint64_t)a->u.l));
        else *out = v_i32((int32_t)(0U - (uint32_t)a->u.i));
        return QO_OK;
    }
    int t = promoted_tag(a, b);
    if (t == QO_T_F64) {
        double x = as_f64(a), y = as_f64(b), r;
        switch (fn) {
        case QO_ADD: r = x + y; break;
        case QO_SUB: r = x - y; break;
        case QO_MUL: r = x * y; break;
        case QO_DIV: r = x / y; break;
        case QO_MOD: r = fmod(x, y); break; /* DREM == C fmod (JLS 15.17.3) */
        default: return QO_BAD_ARG;
        }
        *out = v_f64(r);
        return QO_OK;
    }
    if (t == QO_T_I64) {
        uint64_t x = (uint64_t)as_i64(a), y = (uint64_t)as_i64(b);
        int64_t sx = (int64_t)x, sy = (int64_t)y, r;
        switch (fn) {
        case QO_ADD: r = (int64_t)(x + y); break;
        case QO_SUB: r = (int64_t)(x - y); break;
        case QO_MUL: r = (int64_t)(x * y); break;
        case QO_DIV:
            if (sy == 0) { *out = v_null(); return QO_OK; }
            r = (sx == INT64_MIN && sy == -1) ? INT64_MIN : sx / sy; break;
        case QO_MOD:
            if (sy == 0) { *out = v_null(); return QO_OK; }
            r = (sy == -1) ? 0 : sx % sy; break;
        default: return QO_BAD_ARG;
        }
        *out = v_i64(r);
        return QO_OK;
    }
    {
        uint32_t x = (uint32_t)a->u.i, y = (uint32_t)b->u.i;
        int32_t sx = (int32_t)x, sy = (int32_t)y, r;
        switch (fn) {
        case QO_ADD: r = (int32_t)(x + y); break;
        case QO_SUB: r = (int32_t)(x - y); break;
        case QO_MUL: r = (int32_t)(x * y); break;
        case QO_DIV:
            if (sy == 0) { *out = v_null(); return QO_OK; }
            r = (sx == INT32_MIN && sy == -1) ? INT32_MIN : sx / sy; break;
        case QO_MOD:
            if (sy == 0) { *out = v_null(); return QO_OK; }
            r = (sy == -1) ? 0 : sx % sy; break;
        default: return QO_BAD_ARG;
        }
        *out = v_i32(r);
        return QO_OK;
    }
}

/* ---------- comparison ------------------------------------------------------ */

/* compareTo / compare() of two non-null boxed values of one kind */
static int32_t compare_values(const qo_value *a, const qo_value *b, int32_t *cmp) {
    if (is_numeric(a) && is_numeric(b)) {
        int t = promoted_tag(a, b);
        if (t == QO_T_F64) { *cmp = qo_double_compare(as_f64(a), as_f64(b)); return QO_OK; }
        int64_t x = as_i64(a), y = as_i64(b);
        *cmp = x < y ? -1 : (x > y ? 1 : 0);
        return QO_OK;
    }
    if (a->tag == QO_T_STR && b->tag == QO_T_STR) { *cmp = string_compare(a->u.s, b->u.s); return QO_OK; }
    if (a->tag == QO_T_BOOL && b->tag == QO_T_BOOL) { *cmp = a->u.b - b->u.b; return QO_OK; } /* Boolean.compare */
    return QO_THROWN; /* ClassCastException in compareTo */
}

/* Any.equals for CMP_EQ/CMP_NE in INTERPRETER and CLOSURE (Interpreter.kt:102-103, ClosureCompiler.kt:125-126) */
static int32_t equals_values(const qo_value *a, const qo_value *b, int32_t *eq) {
    if (is_numeric(a) && is_numeric(b)) {
        int t = promoted_tag(a, b);
        if (t == QO_T_F64) *eq = qo_double_equals(as_f64(a), as_f64(b));
        else *eq = as_i64(a) == as_i64(b);
        return QO_OK;
    }
    if (a->tag == QO_T_STR && b->tag == QO_T_STR) { *eq = strcmp(a->u.s, b->u.s) == 0; return QO_OK; }
    if (a->tag == QO_T_BOOL && b->tag == QO_T_BOOL) { *eq = a->u.b == b->u.b; return QO_OK; }
    *eq = 0; /* equals() of different classes is false, no exception */
    return QO_OK;
}

static int32_t comparison(int fn, const qo_value *a, const qo_value *b, int32_t mode, qo_value *out) {
    int32_t c = 0, eq = 0, st;
    if (mode == QO_BYTECODE_COMPILER) {
        /* BytecodeCompiler.kt:286-322: compare()/compareTo() then ifICmp against 0 for all six */
        st = compare_values(a, b, &c);
        if (st) return st;
        switch (fn) {
        case QO_CMP_LT: *out = v_bool(c < 0); break;
        case QO_CMP_LE: *out = v_bool(c <= 0); break;
        case QO_CMP_GE: *out = v_bool(c >= 0); break;
        case QO_CMP_GT: *out = v_bool(c > 0); break;
        case QO_CMP_EQ: *out = v_bool(c == 0); break;
        case QO_CMP_NE: *out = v_bool(c != 0); break;
        default: return QO_BAD_ARG;
        }
        return QO_OK;
    }
    if (fn == QO_CMP_EQ || fn == QO_CMP_NE) {
        st = equals_values(a, b, &eq);
        if (st) return st;
        *out = v_bool(fn == QO_CMP_EQ ? eq : !eq);
        return QO_OK;
    }
    if (mode == QO_CLOSURE_COMPILER) {
        /* ClosureCompiler.kt:127-130: (a as Double) < (b as Double), IEEE-754 primitive compare.
         * Extension: long/int operands compare as promoted primitives (no NaN possible). */
        if (!is_numeric(a) || !is_numeric(b)) return QO_THROWN;
        if (promoted_tag(a, b) == QO_T_F64) {
            double x = as_f64(a), y = as_f64(b);
            switch (fn) {
            case QO_CMP_LT: *out = v_bool(x < y); break;
            case QO_CMP_LE: *out = v_bool(x <= y); break;
            case QO_CMP_GE: *out = v_bool(x >= y); break;
            case QO_CMP_GT: *out = v_bool(x > y); break;
            default: return QO_BAD_ARG;
            }
            return QO_OK;
        }
    }
    /* INTERPRETER: Comparable.compareTo (Interpreter.kt:104-107) */
    st = compare_values(a, b, &c);
    if (st) return st;
    switch (fn) {
    case QO_CMP_LT: *out = v_bool(c < 0); break;
    case QO_CMP_LE: *out = v_bool(c <= 0); break;
    case QO_CMP_GE: *out = v_bool(c >= 0); break;
    case QO_CMP_GT: *out = v_bool(c > 0); break;
    default: return QO_BAD_ARG;
    }
    return QO_OK;
}

/* ---------- the tree walk --------------------------------------------------- */

static int32_t as_boolean_nullable(const qo_value *v, int *isnull, int *b) {
    if (v->tag == QO_T_NULL) { *isnull = 1; *b = 0; return QO_OK; }
    if (v->tag != QO_T_BOOL) return QO_THROWN; /* `as Boolean?` ClassCastException */
    *isnull = 0; *b = v->u.b;
    return QO_OK;
}

int32_t qo_eval(const qo_node *nodes, int32_t root, const qo_value *row, int32_t mode, qo_value *out) {
    const qo_node *n = &nodes[root];
    switch (n->kind) {
    case QO_COLUMN: *out = row[n->col]; return QO_OK;                 /* Interpreter.kt:25-27 */
    case QO_NUMERIC_LITERAL: *out = v_f64(n->num); return QO_OK;      /* :13-15 */
    case QO_BOOLEAN_LITERAL: *out = v_bool(n->bval); return QO_OK;    /* :17-19 */
    case QO_STRING_LITERAL: *out = v_str(n->str); return QO_OK;       /* :21-23 */
    case QO_FUNCTION: break;
    default: return QO_BAD_ARG;
    }

    int32_t st;
    qo_value p, q;
    int pn, pb, qn, qb;

    switch (n->fn) {
    case QO_IF: /* Interpreter.kt:46-53 */
        if ((st = qo_eval(nodes, n->ops[0], row, mode, &p))) return st;
        if ((st = as_boolean_nullable(&p, &pn, &pb))) return st;
        if (pn) { *out = v_null(); return QO_OK; }
        if ((st = qo_eval(nodes, pb ? n->ops[1] : n->ops[2], row, mode, out))) return st;
        /* extension only: branches of different numeric types are promoted to the node's static
         * type, like Java's `c ? long : double` (the reference requires equal types, TypeCheck.kt:89-91) */
        if (is_numeric(out)) {
            if (n->dtype == QO_DOUBLE && out->tag != QO_T_F64) *out = v_f64(as_f64(out));
            else if (n->dtype == QO_INT64 && out->tag == QO_T_I32) *out = v_i64(as_i64(out));
        }
        return QO_OK;

    case QO_AND: /* Interpreter.kt:54-72: lazy Kleene AND */
        if ((st = qo_eval(nodes, n->ops[0], row, mode, &p))) return st;
        if ((st = as_boolean_nullable(&p, &pn, &pb))) return st;
        if (pn) {
            if ((st = qo_eval(nodes, n->ops[1], row, mode, &q))) return st;
            if ((st = as_boolean_nullable(&q, &qn, &qb))) return st;
            *out = (qn || qb) ? v_null() : v_bool(0);
            return QO_OK;
        }
        if (pb) {
            if ((st = qo_eval(nodes, n->ops[1], row, mode, &q))) return st;
            if ((st = as_boolean_nullable(&q, &qn, &qb))) return st;
            *out = qn ? v_null() : v_bool(qb);
            return QO_OK;
        }
        *out = v_bool(0);
        return QO_OK;

    case QO_OR: /* Interpreter.kt:73-91: lazy Kleene OR */
        if ((st = qo_eval(nodes, n->ops[0], row, mode, &p))) return st;
        if ((st = as_boolean_nullable(&p, &pn, &pb))) return st;
        if (pn) {
            if ((st = qo_eval(nodes, n->ops[1], row, mode, &q))) return st;
            if ((st = as_boolean_nullable(&q, &qn, &qb))) return st;
            *out = (!qn && qb) ? v_bool(1) : v_null();
            return QO_OK;
        }
        if (pb) { *out = v_bool(1); return QO_OK; }
        if ((st = qo_eval(nodes, n->ops[1], row, mode, &q))) return st;
        if ((st = as_boolean_nullable(&q, &qn, &qb))) return st;
        *out = qn ? v_null() : v_bool(qb);
        return QO_OK;

    case QO_NOT:
        if ((st = qo_eval(nodes, n->ops[0], row, mode, &p))) return st;
        if (p.tag == QO_T_NULL) {
            /* Interpreter.kt:92 `as Boolean` on null throws; ClosureCompiler.kt:115 and
             * BytecodeCompiler.kt:346-351 propagate null */
            if (mode == QO_INTERPRETER) return QO_THROWN;
            *out = v_null();
            return QO_OK;
        }
        if (p.tag != QO_T_BOOL) return QO_THROWN;
        *out = v_bool(!p.u.b);
        return QO_OK;
    default: break;
    }

    /* ARITHMETIC / COMPARISON: operands evaluated up front, null if any is null
     * (Interpreter.kt:35-42; ClosureCompiler.kt:28-38; BytecodeCompiler.kt:229-258) */
    qo_value a = v_null(), b = v_null();
    if (n->nops < 1 || n->nops > 2) return QO_BAD_ARG;
    if ((st = qo_eval(nodes, n->ops[0], row, mode, &a))) return st;
    if (n->nops == 2) {
        /* CLOSURE returns null before evaluating op2 when op1 is null; no side effects, same value */
        if ((st = qo_eval(nodes, n->ops[1], row, mode, &b))) return st;
    }
    if (a.tag == QO_T_NULL || (n->nops == 2 && b.tag == QO_T_NULL)) { *out = v_null(); return QO_OK; }

    switch (n->fn) {
    case QO_UNARY_MINUS: case QO_UNARY_PLUS:
        return arith(n->fn, &a, NULL, out);
    case QO_MUL: case QO_DIV: case QO_MOD: case QO_ADD: case QO_SUB:
        return arith(n->fn, &a, &b, out);
    case QO_CMP_LT: case QO_CMP_LE: case QO_CMP_GE: case QO_CMP_GT: case QO_CMP_EQ: case QO_CMP_NE:
        return comparison(n->fn, &a, &b, mode, out);
    default:
        return QO_BAD_ARG;
    }
}

/* ---------- operators -------------------------------------------------------- */

/* MemorySourceOperator.next (MemorySourceOperator.kt:18-36): box row i into the reused buffer */
static void scan_row(const qo_column *cols, int32_t ncols, int64_t i, qo_value *row) {
    for (int32_t j = 0; j < ncols; j++) {
        const qo_column *c = &cols[j];
        if (c->valid && !c->valid[i]) { row[j] = v_null(); continue; }
        switch (c->dtype) {
        case QO_DOUBLE: row[j] = v_f64(((const double *)c->data)[i]); break;
        case QO_INT64: row[j] = v_i64(((const int64_t *)c->data)[i]); break;
        case QO_INT32: row[j] = v_i32(((const int32_t *)c->data)[i]); break;
        case QO_BOOLEAN: row[j] = v_bool(((const uint8_t *)c->data)[i]); break;
        case QO_STRING: row[j] = v_str(c->dict[((const int32_t *)c->data)[i]]); break;
        default: row[j] = v_null(); break;
        }
    }
}

static int32_t store_out(qo_out_column *o, int64_t pos, const qo_value *v) {
    if (v->tag == QO_T_NULL) {
        o->valid[pos] = 0;
        switch (o->dtype) {
        case QO_DOUBLE: ((double *)o->data)[pos] = 0.0; break;
        case QO_INT64: ((int64_t *)o->data)[pos] = 0; break;
        case QO_INT32: ((int32_t *)o->data)[pos] = 0; break;
        case QO_BOOLEAN: ((uint8_t *)o->data)[pos] = 0; break;
        case QO_STRING: ((const char **)o->data)[pos] = NULL; break;
        default: return QO_BAD_ARG;
        }
        return QO_OK;
    }
    o->valid[pos] = 1;
    switch (o->dtype) {
    case QO_DOUBLE: if (v->tag != QO_T_F64) return QO_BAD_ARG; ((double *)o->data)[pos] = v->u.d; break;
    case QO_INT64: if (v->tag != QO_T_I64) return QO_BAD_ARG; ((int64_t *)o->data)[pos] = v->u.l; break;
    case QO_INT32: if (v->tag != QO_T_I32) return QO_BAD_ARG; ((int32_t *)o->data)[pos] = v->u.i; break;
    case QO_BOOLEAN: if (v->tag != QO_T_BOOL) return QO_BAD_ARG; ((uint8_t *)o->data)[pos] = (uint8_t)v->u.b; break;
    case QO_STRING: if (v->tag != QO_T_STR) return QO_BAD_ARG; ((const char **)o->data)[pos] = v->u.s; break;
    default: return QO_BAD_ARG;
    }
    return QO_OK;
}

#define QO_MAX_COLS 64

int64_t qo_filter_project(const qo_node *nodes, int32_t filter_root, const int32_t *proj_roots, int32_t nproj,
                          const qo_column *cols, int32_t ncols, int64_t nrows, int32_t mode,
                          qo_out_column *outs, int32_t *err) {
    qo_value row[QO_MAX_COLS];
    int64_t nout = 0;
    *err = QO_OK;
    if (ncols > QO_MAX_COLS) { *err = QO_BAD_ARG; return -1; }
    for (int64_t i = 0; i < nrows; i++) {
        scan_row(cols, ncols, i, row);
        if (filter_root >= 0) {
            /* FilterOperator.next (FilterOperator.kt:17-22) */
            qo_value res;
            int32_t st = qo_eval(nodes, filter_root, row, mode, &res);
            if (st) { *err = st; return -1; }
            if (res.tag == QO_T_NULL) continue;
            if (res.tag != QO_T_BOOL) { *err = QO_THROWN; return -1; }
            if (!res.u.b) continue;
        }
        /* ProjectionOperator.next (ProjectionOperator.kt:15-19) */
        for (int32_t k = 0; k < nproj; k++) {
            qo_value v;
            int32_t st = qo_eval(nodes, proj_roots[k], row, mode, &v);
            if (st) { *err = st; return -1; }
            st = store_out(&outs[k], nout, &v);
            if (st) { *err = st; return -1; }
        }
        nout++;
    }
    return nout;
}

int64_t qo_filter_aggregate(const qo_node *nodes, int32_t filter_root, const int32_t *expr_roots,
                            const int32_t *agg_fns, int32_t nagg,
                            const qo_column *cols, int32_t ncols, int64_t nrows, int32_t mode,
                            double *out_values, uint8_t *out_valid, int32_t *err) {
    qo_value row[QO_MAX_COLS];
    double acc[QO_MAX_COLS];
    int64_t cnt[QO_MAX_COLS];
    int64_t nsel = 0;
    *err = QO_OK;
    if (ncols > QO_MAX_COLS || nagg > QO_MAX_COLS) { *err = QO_BAD_ARG; return -1; }
    for (int32_t k = 0; k < nagg; k++) {
        cnt[k] = 0;
        acc[k] = agg_fns[k] == QO_MIN ? INFINITY : (agg_fns[k] == QO_MAX ? -INFINITY : 0.0); /* Accumulators.kt:57,75 */
    }
    for (int64_t i = 0; i < nrows; i++) {
        scan_row(cols, ncols, i, row);
        if (filter_root >= 0) {
            qo_value res;
            int32_t st = qo_eval(nodes, filter_root, row, mode, &res);
            if (st) { *err = st; return -1; }
            if (res.tag != QO_T_BOOL || !res.u.b) continue;
        }
        nsel++;
        for (int32_t k = 0; k < nagg; k++) {
            qo_value v;
            int32_t st = qo_eval(nodes, expr_roots[k], row, mode, &v);
            if (st) { *err = st; return -1; }
            if (v.tag == QO_T_NULL) continue;                  /* GlobalAggregationOperator.kt:17-20 */
            cnt[k]++;
            if (agg_fns[k] == QO_COUNT) continue;              /* Accumulators.kt:29-31 */
            if (!is_numeric(&v)) { *err = QO_THROWN; return -1; } /* `value as Double` */
            double d = as_f64(&v);
            switch (agg_fns[k]) {
            case QO_SUM: case QO_AVG: acc[k] += d; break;      /* :42-45, :96-99: sequential in row order */
            case QO_MIN: acc[k] = java_min(acc[k], d); break;
            case QO_MAX: acc[k] = java_max(acc[k], d); break;
            default: *err = QO_BAD_ARG; return -1;
            }
        }
    }
    for (int32_t k = 0; k < nagg; k++) {
        if (agg_fns[k] == QO_COUNT) { out_values[k] = (double)cnt[k]; out_valid[k] = 1; continue; }
        out_valid[k] = cnt[k] != 0;
        out_values[k] = cnt[k] == 0 ? 0.0 : (agg_fns[k] == QO_AVG ? acc[k] / (double)cnt[k] : acc[k]);
    }
    return nsel;
}

/* ---------- group by ------------------------------------------------------------ */

#include <stdlib.h>

/* Any?.equals for group keys: Array.contentEquals (GroupByAggregationOperator.kt:9-11) */
static int key_value_equals(const qo_value *a, const qo_value *b) {
    if (a->tag != b->tag) return 0;
    switch (a->tag) {
    case QO_T_NULL: return 1;
    case QO_T_F64: return qo_double_equals(a->u.d, b->u.d);
    case QO_T_BOOL: return a->u.b == b->u.b;
    case QO_T_STR: return strcmp(a->u.s, b->u.s) == 0;
    case QO_T_I64: return a->u.l == b->u.l;
    case QO_T_I32: return a->u.i == b->u.i;
    default: return 0;
    }
}

static uint64_t key_value_hash(const qo_value *v) {
    switch (v->tag) {
    case QO_T_F64: return mix64_fwd((uint64_t)double_to_long_bits(v->u.d) + 1);
    case QO_T_BOOL: return mix64_fwd((uint64_t)v->u.b + 11);
    case QO_T_I64: return mix64_fwd((uint64_t)v->u.l + 3);
    case QO_T_I32: return mix64_fwd((uint64_t)(int64_t)v->u.i + 5);
    case QO_T_STR: {
        uint64_t h = 1469598103934665603ULL;
        for (const unsigned char *c = (const unsigned char *)v->u.s; *c; c++) { h ^= *c; h *= 1099511628211ULL; }
        return h;
    }
    default: return 0x9E3779B97F4A7C15ULL;
    }
}

int64_t qo_filter_groupby(const qo_node *nodes, int32_t filter_root, const int32_t *key_roots, int32_t nkeys,
                          const int32_t *expr_roots, const int32_t *agg_fns, int32_t nagg,
                          const qo_column *cols, int32_t ncols, int64_t nrows, int32_t mode, int64_t max_groups,
                          qo_out_column *out_keys, double *out_values, uint8_t *out_valid, int32_t *err) {
    qo_value row[QO_MAX_COLS], key[QO_MAX_COLS];
    *err = QO_OK;
    if (ncols > QO_MAX_COLS || nagg > QO_MAX_COLS || nkeys > QO_MAX_COLS || max_groups < 0) { *err = QO_BAD_ARG; return -1; }
    /* insertion-ordered map: groups[] in first-appearance order + open-addressing index */
    int64_t cap = 16;
    while (cap < 2 * max_groups + 2) cap <<= 1;
    int64_t *index = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    qo_value *gkeys = (qo_value *)malloc((size_t)(max_groups + 1) * (size_t)(nkeys ? nkeys : 1) * sizeof(qo_value));
    double *acc = (double *)malloc((size_t)(max_groups + 1) * (size_t)(nagg ? nagg : 1) * sizeof(double));
    int64_t *cnt = (int64_t *)malloc((size_t)(max_groups + 1) * (size_t)(nagg ? nagg : 1) * sizeof(int64_t));
    int64_t ngroups = 0;
    if (!index || !gkeys || !acc || !cnt) { *err = QO_BAD_ARG; ngroups = -1; goto done; }
    for (int64_t i = 0; i < cap; i++) index[i] = -1;
    for (int64_t i = 0; i < nrows; i++) {
        scan_row(cols, ncols, i, row);
        if (filter_root >= 0) {
            qo_value res;
            int32_t st = qo_eval(nodes, filter_root, row, mode, &res);
            if (st) { *err = st; ngroups = -1; goto done; }
            if (res.tag != QO_T_BOOL || !res.u.b) continue;
        }
        uint64_t h = 0x243F6A8885A308D3ULL;
        for (int32_t k = 0; k < nkeys; k++) {
            int32_t st = qo_eval(nodes, key_roots[k], row, mode, &key[k]);
            if (st) { *err = st; ngroups = -1; goto done; }
            h = mix64_fwd(h ^ key_value_hash(&key[k]));
        }
        int64_t slot = (int64_t)(h & (uint64_t)(cap - 1)), g = -1;
        for (;;) {   /* map.computeIfAbsent(key) (:35-37) */
            int64_t cand = index[slot];
            if (cand < 0) break;
            int eq = 1;
            for (int32_t k = 0; k < nkeys && eq; k++) eq = key_value_equals(&gkeys[cand * nkeys + k], &key[k]);
            if (eq) { g = cand; break; }
            slot = (slot + 1) & (cap - 1);
        }
        if (g < 0) {
            if (ngroups >= max_groups) { *err = QO_BAD_ARG; ngroups = -1; goto done; }
            g = ngroups++;
            index[slot] = g;
            for (int32_t k = 0; k < nkeys; k++) gkeys[g * nkeys + k] = key[k];
            for (int32_t a = 0; a < nagg; a++) {
                cnt[g * nagg + a] = 0;
                acc[g * nagg + a] = agg_fns[a] == QO_MIN ? INFINITY : (agg_fns[a] == QO_MAX ? -INFINITY : 0.0);
            }
        }
        for (int32_t a = 0; a < nagg; a++) {   /* :39-44 */
            qo_value v;
            int32_t st = qo_eval(nodes, expr_roots[a], row, mode, &v);
            if (st) { *err = st; ngroups = -1; goto done; }
            if (v.tag == QO_T_NULL) continue;
            cnt[g * nagg + a]++;
            if (agg_fns[a] == QO_COUNT) continue;
            if (!is_numeric(&v)) { *err = QO_THROWN; ngroups = -1; goto done; }
            double d = as_f64(&v);
            switch (agg_fns[a]) {
            case QO_SUM: case QO_AVG: acc[g * nagg + a] += d; break;
            case QO_MIN: acc[g * nagg + a] = java_min(acc[g * nagg + a], d); break;
            case QO_MAX: acc[g * nagg + a] = java_max(acc[g * nagg + a], d); break;
            default: *err = QO_BAD_ARG; ngroups = -1; goto done;
            }
        }
    }
    for (int64_t g = 0; g < ngroups; g++) {
        for (int32_t k = 0; k < nkeys; k++) {
            int32_t st = store_out(&out_keys[k], g, &gkeys[g * nkeys + k]);
            if (st) { *err = st; ngroups = -1; goto done; }
        }
        for (int32_t a = 0; a < nagg; a++) {
            int64_t c = cnt[g * nagg + a];
            double x = acc[g * nagg + a];
            if (agg_fns[a] == QO_COUNT) { out_values[g * nagg + a] = (double)c; out_valid[g * nagg + a] = 1; continue; }
            out_valid[g * nagg + a] = c != 0;
            out_values[g * nagg + a] = c == 0 ? 0.0 : (agg_fns[a] == QO_AVG ? x / (double)c : x);
        }
    }
done:
    free(index); free(gkeys); free(acc); free(cnt);
    return ngroups;
}

/* ---------- synthetic generator ----------------------- */

static uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

#define QO_GOLDEN 0x9E3779B97F4A7C15ULL

uint64_t qo_gen_raw(uint64_t seed, int32_t col_id, uint64_t row) {
    return mix64(QO_GOLDEN * (uint64_t)(col_id + 1) + row * QO_GOLDEN + seed);
}

void qo_generate(const qo_gen_spec *s, uint64_t seed, int64_t row_begin, int64_t nrows, void *data, uint8_t *valid) {
    for (int64_t k = 0; k < nrows; k++) {
        uint64_t i = (uint64_t)(row_begin + k);
        uint64_t x = qo_gen_raw(seed, s->col_id, i);
        switch (s->kind) {
        case QO_GEN_I64_MOD: ((int64_t *)data)[k] = (int64_t)(x % s->modulus) + s->offset; break;
        case QO_GEN_I64_ROWID: ((int64_t *)data)[k] = (int64_t)i; break;
        case QO_GEN_I32_MOD: ((int32_t *)data)[k] = (int32_t)((int64_t)(x % s->modulus) + s->offset); break;
        case QO_GEN_F64_UNIT: ((double *)data)[k] = (double)(x >> 11) * 0x1.0p-53; break;
        case QO_GEN_F64_MOD: ((double *)data)[k] = (double)((int64_t)(x % s->modulus) + s->offset); break;
        case QO_GEN_F64_STEP: ((double *)data)[k] = (double)((int64_t)(x % s->modulus) + s->offset) * s->step; break;
        case QO_GEN_F64_PRICE: {
            uint64_t q = qo_gen_raw(seed, s->aux_col_id, i) % 50 + 1;
            uint64_t cents = 90000 + x % 120000;
            ((double *)data)[k] = (double)(int64_t)(q * cents) / 100.0;
            break;
        }
        default: break;
        }
        if (valid) valid[k] = s->null_pct > 0 ? (uint8_t)(mix64(x + QO_GOLDEN) % 100 >= (uint64_t)s->null_pct) : 1;
    }
}
