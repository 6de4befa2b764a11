/* rmock.c — a small stand-in for the part of R's C API that matrixextra_amd/csrc/r_shim.cpp calls, so that the `.Call`
 * shim can be EXECUTED (tests/test_gpu_r_shim_exec.py) in images that have no R.
 *
 * TEST INFRASTRUCTURE ONLY.  It implements the declarations of tests/r_api_decls/ (Rinternals.h, R_ext/Rdynload.h) with
 * the semantics "Writing R Extensions" §5 / R 4.x gives them, restricted to what marshalling code can observe:
 *
 *   * SEXPs are tagged heap records (LGLSXP / INTSXP / REALSXP / STRSXP / VECSXP / CHARSXP / SYMSXP / S4SXP) with an
 *     attribute list (names, dim, class) and, for S4 objects, a slot list;
 *   * PROTECT / UNPROTECT are a real stack, and a `gctorture` mode runs a mark & sweep on EVERY allocation from the roots R
 *     has (protect stack, R_PreserveObject list, the arguments of the running .Call, objects the driver holds): anything
 *     else is collected — its payload poisoned, every later access recorded as a violation.  A missing PROTECT in the
 *     shim therefore shows up as a failed test, as it would under R's own gctorture(TRUE);
 *   * fresh vectors are filled with a poison pattern (R does not zero Rf_allocVector memory either);
 *   * Rf_error() formats the message, runs the pending R_ExecWithCleanup handlers, resets the protect stack to the depth
 *     of the running .Call and long-jumps to the trampoline (rmock_dotcall), which reports an R error to the driver;
 *   * R_ExecWithCleanup calls the cleanup function on normal return AND on a jump (R's begincontext/cend behaviour);
 *   * an allocation can be made to fail on demand (rmock_fail_alloc_at) — R's "cannot allocate vector of size" long-jump;
 *   * R_registerRoutines captures the R_CallMethodDef table; rmock_dotcall looks a routine up BY NAME, checks the number
 *     of arguments as R does for registered routines, and checks the protect stack balance on return
 *     (R: "stack imbalance in .Call").
 *
 * Nothing here knows about MatrixExtra. */
#define _GNU_SOURCE
#include <limits.h>
#include <math.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#define CHARSXP 9
#define SYMSXP 1
#define S4SXP 25
#define MAGIC_LIVE 0x5EC5EC5Eu
#define MAGIC_DEAD 0xDEADDEADu
#define MAX_ATTR 8
#define MAX_SLOTS 16

struct SEXPREC {
    uint32_t magic;
    int type;
    R_xlen_t length;
    void *data;               /* payload: int / double / SEXP[] / char[] */
    size_t bytes;
    int nattr;
    SEXP attr_sym[MAX_ATTR], attr_val[MAX_ATTR];
    int nslots;
    SEXP slot_sym[MAX_SLOTS], slot_val[MAX_SLOTS];
    int mark;
    int held;                 /* the driver holds it (an input, or a result it has not released) */
    uint64_t id;
    struct SEXPREC *next;     /* all objects */
};

static struct SEXPREC nil_rec = {MAGIC_LIVE, NILSXP, 0, NULL, 0, 0, {0}, {0}, 0, {0}, {0}, 0, 1, 0, NULL};
SEXP R_NilValue = &nil_rec;
SEXP R_NamesSymbol, R_DimSymbol, R_ClassSymbol;

static SEXP all_objects;
static uint64_t next_id = 1;
static long n_allocs, n_collected;
static int gctorture;
static long fail_alloc_countdown = -1;

#define PSTACK_MAX 10000
static SEXP pstack[PSTACK_MAX];
static int pdepth;

static SEXP *preserved;
static int n_preserved, cap_preserved;

static char violation[512];
static long n_violations;
static char last_error[1024];

/* the running .Call */
static jmp_buf *call_jmp;
static int call_pbase;
static SEXP *call_args;
static int call_nargs;
struct cleanup { void (*fn)(void *); void *data; };
static struct cleanup cleanups[64];
static int n_cleanups;
static void **transient;
static int n_transient, cap_transient;

static void violate(const char *fmt, ...)
{
    n_violations++;
    if (violation[0]) return;               /* keep the first */
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(violation, sizeof violation, fmt, ap);
    va_end(ap);
}

static int alive(SEXP x, const char *who)
{
    if (!x) { violate("%s: NULL SEXP", who); return 0; }
    if (x->magic == MAGIC_DEAD) { violate("%s: object #%llu was garbage-collected (missing PROTECT)", who, (unsigned long long)x->id); return 0; }
    if (x->magic != MAGIC_LIVE) { violate("%s: not a SEXP", who); return 0; }
    return 1;
}

/* ------------------------------------------------------------------------------------------- garbage collection */
static void mark(SEXP x)
{
    if (!x || x->magic != MAGIC_LIVE || x->mark) return;
    x->mark = 1;
    for (int k = 0; k < x->nattr; k++) { mark(x->attr_sym[k]); mark(x->attr_val[k]); }
    for (int k = 0; k < x->nslots; k++) { mark(x->slot_sym[k]); mark(x->slot_val[k]); }
    if (x->type == VECSXP || x->type == STRSXP)
        for (R_xlen_t k = 0; k < x->length; k++) mark(((SEXP *)x->data)[k]);
}

static void collect(void)
{
    for (SEXP o = all_objects; o; o = o->next) o->mark = 0;
    for (SEXP o = all_objects; o; o = o->next)
        if (o->magic == MAGIC_LIVE && (o->held || o->type == SYMSXP)) mark(o);
    for (int k = 0; k < pdepth; k++) mark(pstack[k]);
    for (int k = 0; k < n_preserved; k++) mark(preserved[k]);
    for (int k = 0; k < call_nargs; k++) mark(call_args[k]);
    for (SEXP o = all_objects; o; o = o->next)
        if (o->magic == MAGIC_LIVE && !o->mark) {
            o->magic = MAGIC_DEAD;
            if (o->data && o->bytes) memset(o->data, 0xDD, o->bytes);     /* use-after-collect reads garbage, not old data */
            n_collected++;
        }
}

static SEXP new_object(int type, R_xlen_t n, size_t elt)
{
    if (fail_alloc_countdown >= 0 && fail_alloc_countdown-- == 0)
        Rf_error("cannot allocate vector of size %.1f Mb (rmock: injected allocation failure)", (double)n * (double)elt / 1048576.0);
    if (gctorture) collect();
    SEXP x = (SEXP)calloc(1, sizeof(struct SEXPREC));
    if (!x) abort();
    x->magic = MAGIC_LIVE;
    x->type = type;
    x->length = n;
    x->bytes = (size_t)n * elt;
    if (x->bytes) {
        x->data = malloc(x->bytes);
        if (!x->data) abort();
        /* R does not clear new vectors (VECSXP / STRSXP slots are initialised: R_NilValue / R_BlankString) */
        if (type == VECSXP || type == STRSXP) for (R_xlen_t k = 0; k < n; k++) ((SEXP *)x->data)[k] = R_NilValue;
        else memset(x->data, 0xA5, x->bytes);
    }
    x->id = next_id++;
    x->next = all_objects;
    all_objects = x;
    n_allocs++;
    return x;
}

/* ------------------------------------------------------------------------------------------- the R API subset */
int TYPEOF(SEXP x) { return alive(x, "TYPEOF") ? x->type : NILSXP; }
R_xlen_t XLENGTH(SEXP x) { return alive(x, "XLENGTH") ? x->length : 0; }
R_xlen_t Rf_xlength(SEXP x)                      /* any SEXP: vectors their length, NULL 0, everything else 1 */
{
    if (!x || x == R_NilValue || !alive(x, "Rf_xlength")) return 0;
    switch (x->type) {
    case LGLSXP: case INTSXP: case REALSXP: case STRSXP: case VECSXP: return x->length;
    default: return 1;
    }
}

static void *payload(SEXP x, int t1, int t2, const char *who)
{
    static double scratch[8];
    if (!alive(x, who)) return scratch;
    if (x->type != t1 && x->type != t2) {                       /* R: "INTEGER() can only be applied to a 'integer', not a ..." */
        Rf_error("%s can only be applied to type %d, not %d", who, t1, x->type);
    }
    return x->data ? x->data : (void *)scratch;                 /* zero-length vectors have a valid, non-NULL pointer in R */
}
int *INTEGER(SEXP x) { return (int *)payload(x, INTSXP, LGLSXP, "INTEGER()"); }
int *LOGICAL(SEXP x) { return (int *)payload(x, LGLSXP, LGLSXP, "LOGICAL()"); }
double *REAL(SEXP x) { return (double *)payload(x, REALSXP, REALSXP, "REAL()"); }

SEXP VECTOR_ELT(SEXP x, R_xlen_t i)
{
    if (!alive(x, "VECTOR_ELT")) return R_NilValue;
    if (x->type != VECSXP) Rf_error("VECTOR_ELT() can only be applied to a 'list', not type %d", x->type);
    if (i < 0 || i >= x->length) Rf_error("attempt to access index %ld/%ld in VECTOR_ELT", (long)i, (long)x->length);
    return ((SEXP *)x->data)[i];
}
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v)
{
    if (!alive(x, "SET_VECTOR_ELT") || !alive(v, "SET_VECTOR_ELT value")) return v;
    if (x->type != VECSXP) Rf_error("SET_VECTOR_ELT() can only be applied to a 'list', not type %d", x->type);
    if (i < 0 || i >= x->length) Rf_error("attempt to set index %ld/%ld in SET_VECTOR_ELT", (long)i, (long)x->length);
    ((SEXP *)x->data)[i] = v;
    return v;
}
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v)
{
    if (!alive(x, "SET_STRING_ELT") || !alive(v, "SET_STRING_ELT value")) return;
    if (x->type != STRSXP) Rf_error("SET_STRING_ELT() can only be applied to a 'character vector', not type %d", x->type);
    if (v->type != CHARSXP) Rf_error("Value of SET_STRING_ELT() must be a 'CHARSXP' not type %d", v->type);
    if (i < 0 || i >= x->length) Rf_error("attempt to set index %ld/%ld in SET_STRING_ELT", (long)i, (long)x->length);
    ((SEXP *)x->data)[i] = v;
}
SEXP STRING_ELT(SEXP x, R_xlen_t i)
{
    if (!alive(x, "STRING_ELT")) return R_NilValue;
    if (x->type != STRSXP) Rf_error("STRING_ELT() can only be applied to a 'character vector', not type %d", x->type);
    if (i < 0 || i >= x->length) Rf_error("attempt to access index %ld/%ld in STRING_ELT", (long)i, (long)x->length);
    return ((SEXP *)x->data)[i];
}
const char *R_CHAR(SEXP x)
{
    if (!alive(x, "R_CHAR")) return "";
    if (x->type != CHARSXP && x->type != SYMSXP) Rf_error("CHAR() can only be applied to a 'CHARSXP', not type %d", x->type);
    return x->data ? (const char *)x->data : "";
}

SEXP Rf_protect(SEXP s)
{
    if (pdepth >= PSTACK_MAX) Rf_error("protect(): protection stack overflow");
    alive(s, "PROTECT");
    pstack[pdepth++] = s;
    return s;
}
void Rf_unprotect(int n)
{
    if (n > pdepth - (call_jmp ? call_pbase : 0)) { violate("UNPROTECT(%d): only %d protected in this .Call (stack imbalance)", n, pdepth - call_pbase); n = pdepth - call_pbase; }
    pdepth -= n;
}

SEXP Rf_allocVector(SEXPTYPE t, R_xlen_t n)
{
    if (n < 0) Rf_error("negative length vectors are not allowed");
    switch (t) {
    case LGLSXP: case INTSXP: return new_object((int)t, n, sizeof(int));
    case REALSXP: return new_object(REALSXP, n, sizeof(double));
    case STRSXP: case VECSXP: return new_object((int)t, n, sizeof(SEXP));
    default: Rf_error("allocVector: type %u is not implemented in rmock", t);
    }
    return R_NilValue;
}

static SEXP get_attr(SEXP x, SEXP sym)
{
    for (int k = 0; k < x->nattr; k++) if (x->attr_sym[k] == sym) return x->attr_val[k];
    return R_NilValue;
}
SEXP Rf_setAttrib(SEXP x, SEXP sym, SEXP val)
{
    if (!alive(x, "setAttrib") || !alive(sym, "setAttrib name") || !alive(val, "setAttrib value")) return val;
    if (sym == R_NamesSymbol && val != R_NilValue && (val->type != STRSXP || val->length != x->length))
        Rf_error("'names' attribute [%ld] must be the same length as the vector [%ld]", (long)val->length, (long)x->length);
    for (int k = 0; k < x->nattr; k++) if (x->attr_sym[k] == sym) { x->attr_val[k] = val; return val; }
    if (x->nattr >= MAX_ATTR) Rf_error("rmock: too many attributes");
    x->attr_sym[x->nattr] = sym;
    x->attr_val[x->nattr++] = val;
    return val;
}

SEXP Rf_allocMatrix(SEXPTYPE t, int nr, int nc)
{
    if (nr < 0 || nc < 0) Rf_error("negative extents to matrix");
    if ((double)nr * (double)nc > (double)INT_MAX && 0) Rf_error("allocMatrix: too many elements specified");   /* long vectors are allowed */
    SEXP x = Rf_protect(Rf_allocVector(t, (R_xlen_t)nr * nc));
    SEXP dim = Rf_protect(Rf_allocVector(INTSXP, 2));
    INTEGER(dim)[0] = nr;
    INTEGER(dim)[1] = nc;
    Rf_setAttrib(x, R_DimSymbol, dim);
    Rf_unprotect(2);
    return x;
}

#define NA_INT INT_MIN
static double na_real(void)
{
    union { double d; uint32_t w[2]; } u;
    u.w[1] = 0x7FF00000u;       /* little endian: high word */
    u.w[0] = 1954;
    return u.d;
}

SEXP Rf_coerceVector(SEXP x, SEXPTYPE t)
{
    if (!alive(x, "coerceVector")) return R_NilValue;
    if ((SEXPTYPE)x->type == t) return x;
    if (x->type == NILSXP) return Rf_allocVector(t, 0);           /* as.integer(NULL) is integer(0) */
    const int from = x->type;
    if ((from != LGLSXP && from != INTSXP && from != REALSXP) || (t != LGLSXP && t != INTSXP && t != REALSXP))
        Rf_error("cannot coerce type %d to vector of type %u", from, t);
    SEXP out = Rf_protect(Rf_allocVector(t, x->length));
    for (R_xlen_t k = 0; k < x->length; k++) {
        if (from == REALSXP) {
            const double v = ((double *)x->data)[k];
            int r;
            if (isnan(v)) r = NA_INT;
            else if (t == LGLSXP) r = v != 0;
            else if (v >= 2147483648.0 || v <= -2147483649.0) r = NA_INT;     /* R warns "NAs introduced by coercion to integer range" */
            else r = (int)v;
            ((int *)out->data)[k] = r;
        } else {
            const int v = ((int *)x->data)[k];
            if (t == REALSXP) ((double *)out->data)[k] = v == NA_INT ? na_real() : (double)v;
            else if (t == LGLSXP) ((int *)out->data)[k] = v == NA_INT ? NA_INT : (v != 0);
            else ((int *)out->data)[k] = v;                         /* logical -> integer keeps 0 / 1 / NA */
        }
    }
    for (int k = 0; k < x->nattr; k++) Rf_setAttrib(out, x->attr_sym[k], x->attr_val[k]);     /* dim, names survive */
    Rf_unprotect(1);
    return out;
}

static SEXP mk_text(int type, const char *s)
{
    const size_t n = strlen(s);
    SEXP x = new_object(type, (R_xlen_t)n, 1);
    free(x->data);
    x->data = malloc(n + 1);
    memcpy(x->data, s, n + 1);
    x->bytes = n + 1;
    return x;
}
SEXP Rf_mkChar(const char *s) { return mk_text(CHARSXP, s); }
SEXP Rf_install(const char *name)
{
    for (SEXP o = all_objects; o; o = o->next)
        if (o->type == SYMSXP && o->magic == MAGIC_LIVE && !strcmp((const char *)o->data, name)) return o;
    /* symbols are never collected; creating one must not run the collector either (R's symbol table is a root, and
     * shim code like `R_do_slot(o, Rf_install("p"))` is legal) */
    const int g = gctorture;
    const long f = fail_alloc_countdown;
    gctorture = 0; fail_alloc_countdown = -1;
    SEXP s = mk_text(SYMSXP, name);
    gctorture = g; fail_alloc_countdown = f;
    return s;
}
SEXP Rf_ScalarLogical(int v)
{
    SEXP x = Rf_allocVector(LGLSXP, 1);
    ((int *)x->data)[0] = v == NA_INT ? NA_INT : (v != 0);
    return x;
}

int Rf_asInteger(SEXP x)
{
    if (!alive(x, "asInteger") || x->length < 1) return NA_INT;
    if (x->type == INTSXP || x->type == LGLSXP) return ((int *)x->data)[0];
    if (x->type == REALSXP) {
        const double v = ((double *)x->data)[0];
        if (isnan(v) || v >= 2147483648.0 || v <= -2147483649.0) return NA_INT;
        return (int)v;
    }
    return NA_INT;
}
int Rf_asLogical(SEXP x)
{
    if (!alive(x, "asLogical") || x->length < 1) return NA_INT;
    if (x->type == LGLSXP) return ((int *)x->data)[0];
    if (x->type == INTSXP) { const int v = ((int *)x->data)[0]; return v == NA_INT ? NA_INT : v != 0; }
    if (x->type == REALSXP) { const double v = ((double *)x->data)[0]; return isnan(v) ? NA_INT : v != 0; }
    return NA_INT;
}
int Rf_nrows(SEXP x)
{
    if (!alive(x, "nrows")) return 0;
    SEXP d = get_attr(x, R_DimSymbol);
    if (d == R_NilValue) return (int)x->length;                   /* R: a vector is a one-column matrix */
    return ((int *)d->data)[0];
}
int Rf_ncols(SEXP x)
{
    if (!alive(x, "ncols")) return 0;
    SEXP d = get_attr(x, R_DimSymbol);
    if (d == R_NilValue || d->length < 2) return 1;
    return ((int *)d->data)[1];
}
Rboolean Rf_inherits(SEXP x, const char *name)
{
    if (!alive(x, "inherits")) return FALSE;
    SEXP c = get_attr(x, R_ClassSymbol);
    if (c == R_NilValue) return FALSE;
    for (R_xlen_t k = 0; k < c->length; k++) if (!strcmp(R_CHAR(((SEXP *)c->data)[k]), name)) return TRUE;
    return FALSE;
}
int R_has_slot(SEXP obj, SEXP name)
{
    if (!alive(obj, "R_has_slot") || !alive(name, "R_has_slot name")) return 0;
    for (int k = 0; k < obj->nslots; k++) if (obj->slot_sym[k] == name) return 1;
    return 0;
}
SEXP R_do_slot(SEXP obj, SEXP name)
{
    if (!alive(obj, "R_do_slot") || !alive(name, "R_do_slot name")) return R_NilValue;
    for (int k = 0; k < obj->nslots; k++) if (obj->slot_sym[k] == name) return obj->slot_val[k];
    SEXP c = get_attr(obj, R_ClassSymbol);
    Rf_error("no slot of name \"%s\" for this object of class \"%s\"", R_CHAR(name),
             c != R_NilValue && c->length ? R_CHAR(((SEXP *)c->data)[0]) : "?");
    return R_NilValue;
}

void R_PreserveObject(SEXP x)
{
    alive(x, "R_PreserveObject");
    if (n_preserved == cap_preserved) { cap_preserved = cap_preserved ? 2 * cap_preserved : 16; preserved = (SEXP *)realloc(preserved, sizeof(SEXP) * (size_t)cap_preserved); }
    preserved[n_preserved++] = x;
}
void R_ReleaseObject(SEXP x)
{
    for (int k = n_preserved - 1; k >= 0; k--)
        if (preserved[k] == x) { preserved[k] = preserved[--n_preserved]; return; }
    violate("R_ReleaseObject: object #%llu was not preserved", (unsigned long long)(x ? x->id : 0));
}

SEXP R_ExecWithCleanup(SEXP (*fun)(void *), void *data, void (*cleanfun)(void *), void *cleandata)
{
    if (n_cleanups >= 64) Rf_error("rmock: too many nested R_ExecWithCleanup");
    cleanups[n_cleanups].fn = cleanfun;
    cleanups[n_cleanups++].data = cleandata;
    SEXP r = fun(data);
    n_cleanups--;
    cleanfun(cleandata);                       /* on normal return too (R: cntxt.cend runs in endcontext's place) */
    return r;
}

char *R_alloc(size_t n, int size)
{
    void *p = malloc(n * (size_t)size + 1);
    if (!p) Rf_error("cannot allocate memory block");
    if (n_transient == cap_transient) { cap_transient = cap_transient ? 2 * cap_transient : 16; transient = (void **)realloc(transient, sizeof(void *) * (size_t)cap_transient); }
    transient[n_transient++] = p;
    return (char *)p;
}

void Rf_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error, sizeof last_error, fmt, ap);
    va_end(ap);
    if (!call_jmp) { fprintf(stderr, "rmock: Rf_error outside a .Call: %s\n", last_error); abort(); }
    while (n_cleanups > 0) { n_cleanups--; cleanups[n_cleanups].fn(cleanups[n_cleanups].data); }     /* innermost first */
    pdepth = call_pbase;                       /* R unwinds the protect stack to the context's depth */
    longjmp(*call_jmp, 1);
}

/* ------------------------------------------------------------------------------------------- registration */
struct _DllInfo { const R_CallMethodDef *table; int n; int dynamic_symbols; };
static DllInfo the_dll = {NULL, 0, 1};

int R_registerRoutines(DllInfo *info, const void *c, const R_CallMethodDef *call, const void *f, const void *e)
{
    (void)c; (void)f; (void)e;
    info->table = call;
    info->n = 0;
    if (call) while (call[info->n].name) info->n++;
    return 1;
}
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value)
{
    const Rboolean old = info->dynamic_symbols ? TRUE : FALSE;
    info->dynamic_symbols = value;
    return old;
}

/* R_FindSymbol(name, "MatrixExtra", NULL): MatrixExtra's own DLL beside the shim.  Off by default (the shim loaded on its own:
 * every call is served by the backend); rmock_set_host_routines(1) makes every "_MatrixExtra_*" name resolve to one stub that
 * counts its calls and returns the string "host" — the shim must hand small operands to it and large ones to the backend. */
static int host_routines_on = 0;
static long host_calls = 0;
static char host_last[128];
static SEXP host_stub(void)
{
    host_calls++;
    SEXP s = Rf_allocVector(STRSXP, 1);
    PROTECT(s);
    ((SEXP *)s->data)[0] = Rf_mkChar("host");
    UNPROTECT(1);
    return s;
}
DL_FUNC R_FindSymbol(char const *name, char const *pkg, struct Rf_RegisteredNativeSymbol *symbol)
{
    (void)symbol;
    if (!host_routines_on || !pkg || strcmp(pkg, "MatrixExtra") != 0 || strncmp(name, "_MatrixExtra_", 13) != 0) return NULL;
    snprintf(host_last, sizeof host_last, "%s", name);
    return (DL_FUNC)host_stub;
}
void rmock_set_host_routines(int on) { host_routines_on = on; }
long rmock_host_calls(void) { return host_calls; }
const char *rmock_host_last(void) { return host_last; }

/* ------------------------------------------------------------------------------------------- driver interface (ctypes) */
static void boot(void)
{
    if (R_NamesSymbol) return;
    R_NamesSymbol = Rf_install("names");
    R_DimSymbol = Rf_install("dim");
    R_ClassSymbol = Rf_install("class");
}

DllInfo *rmock_dllinfo(void) { boot(); return &the_dll; }
int rmock_n_routines(void) { return the_dll.n; }
const char *rmock_routine_name(int k) { return k >= 0 && k < the_dll.n ? the_dll.table[k].name : NULL; }
int rmock_routine_arity(int k) { return k >= 0 && k < the_dll.n ? the_dll.table[k].numArgs : -1; }
int rmock_dynamic_symbols(void) { return the_dll.dynamic_symbols; }

SEXP rmock_nil(void) { return R_NilValue; }
SEXP rmock_alloc(int type, int64_t n)            /* an object the driver holds (an argument of a later .Call) */
{
    boot();
    const int g = gctorture; const long f = fail_alloc_countdown;
    gctorture = 0; fail_alloc_countdown = -1;
    SEXP x = type == S4SXP ? new_object(S4SXP, 0, 0) : Rf_allocVector((SEXPTYPE)type, (R_xlen_t)n);
    x->held = 1;
    if (x->bytes && type != VECSXP && type != STRSXP) memset(x->data, 0, x->bytes);
    gctorture = g; fail_alloc_countdown = f;
    return x;
}
void *rmock_dataptr(SEXP x) { return x && x->magic == MAGIC_LIVE ? x->data : NULL; }
int64_t rmock_length(SEXP x) { return x && x->magic == MAGIC_LIVE ? (int64_t)x->length : -1; }
int rmock_typeof(SEXP x) { return x && x->magic == MAGIC_LIVE ? x->type : -1; }
int rmock_is_live(SEXP x) { return x && x->magic == MAGIC_LIVE; }
uint64_t rmock_id(SEXP x) { return x ? x->id : 0; }
void rmock_hold(SEXP x) { if (x && x != R_NilValue) x->held = 1; }
void rmock_release(SEXP x) { if (x && x != R_NilValue) x->held = 0; }
void rmock_set_dim(SEXP x, int nr, int nc)
{
    const int g = gctorture; gctorture = 0;
    SEXP d = Rf_allocVector(INTSXP, 2);
    ((int *)d->data)[0] = nr; ((int *)d->data)[1] = nc;
    Rf_setAttrib(x, R_DimSymbol, d);
    gctorture = g;
}
int rmock_get_dim(SEXP x, int *nr, int *nc)
{
    SEXP d = get_attr(x, R_DimSymbol);
    if (d == R_NilValue) return 0;
    *nr = ((int *)d->data)[0]; *nc = ((int *)d->data)[1];
    return 1;
}
static SEXP held_strvec(int n, const char *const *s)
{
    SEXP v = Rf_allocVector(STRSXP, n);
    for (int k = 0; k < n; k++) ((SEXP *)v->data)[k] = Rf_mkChar(s[k]);
    return v;
}
void rmock_set_class(SEXP x, const char *cls)
{
    const int g = gctorture; gctorture = 0;
    Rf_setAttrib(x, R_ClassSymbol, held_strvec(1, &cls));
    gctorture = g;
}
void rmock_set_slot(SEXP obj, const char *name, SEXP v)
{
    SEXP s = Rf_install(name);
    for (int k = 0; k < obj->nslots; k++) if (obj->slot_sym[k] == s) { obj->slot_val[k] = v; return; }
    if (obj->nslots >= MAX_SLOTS) abort();
    obj->slot_sym[obj->nslots] = s;
    obj->slot_val[obj->nslots++] = v;
}
SEXP rmock_get_slot(SEXP obj, const char *name)
{
    SEXP s = Rf_install(name);
    for (int k = 0; k < obj->nslots; k++) if (obj->slot_sym[k] == s) return obj->slot_val[k];
    return NULL;
}
SEXP rmock_mkstring(const char *s)
{
    const int g = gctorture; gctorture = 0;
    SEXP v = held_strvec(1, &s);
    v->held = 1;
    gctorture = g;
    return v;
}
SEXP rmock_list_elt(SEXP x, int64_t k) { return x && x->type == VECSXP && k >= 0 && k < x->length ? ((SEXP *)x->data)[k] : NULL; }
void rmock_set_list_elt(SEXP x, int64_t k, SEXP v) { ((SEXP *)x->data)[k] = v; }
const char *rmock_string_elt(SEXP x, int64_t k)
{
    if (!x || x->type != STRSXP || k < 0 || k >= x->length) return NULL;
    SEXP c = ((SEXP *)x->data)[k];
    return c && c->magic == MAGIC_LIVE && c->data ? (const char *)c->data : "";
}
SEXP rmock_names(SEXP x) { SEXP n = get_attr(x, R_NamesSymbol); return n == R_NilValue ? NULL : n; }

void rmock_set_gctorture(int on) { gctorture = on; }
void rmock_fail_alloc_at(long k) { fail_alloc_countdown = k; }       /* the k-th allocation from now (0 = the next) fails; -1 = off */
void rmock_gc(void) { collect(); }
int rmock_protect_depth(void) { return pdepth; }
int rmock_preserved_count(void) { return n_preserved; }
long rmock_violations(void) { return n_violations; }
const char *rmock_violation_msg(void) { return violation; }
void rmock_clear_violations(void) { n_violations = 0; violation[0] = 0; }
const char *rmock_last_error(void) { return last_error; }
long rmock_alloc_count(void) { return n_allocs; }
long rmock_collected_count(void) { return n_collected; }
long rmock_live_count(void) { long n = 0; for (SEXP o = all_objects; o; o = o->next) n += o->magic == MAGIC_LIVE; return n; }

/* free every collected object's memory (the records of live ones stay) */
void rmock_sweep_dead(void)
{
    SEXP *link = &all_objects;
    while (*link) {
        SEXP o = *link;
        if (o->magic == MAGIC_DEAD) { *link = o->next; free(o->data); o->magic = 0; free(o); }
        else link = &o->next;
    }
}

typedef SEXP (*fn0)(void);
typedef SEXP (*fn1)(SEXP);
typedef SEXP (*fn2)(SEXP, SEXP);
typedef SEXP (*fn3)(SEXP, SEXP, SEXP);
typedef SEXP (*fn4)(SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn5)(SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn6)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn7)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn8)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn9)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn10)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn11)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);

/* .Call(name, args...) for a REGISTERED routine.  0 = value in *result (held by the driver until rmock_release);
 * 1 = the routine raised an R error (rmock_last_error()); 2 = no such routine; 3 = wrong number of arguments. */
int rmock_dotcall(const char *name, int nargs, SEXP *args, SEXP *result)
{
    boot();
    *result = NULL;
    const R_CallMethodDef *def = NULL;
    for (int k = 0; k < the_dll.n; k++) if (!strcmp(the_dll.table[k].name, name)) { def = &the_dll.table[k]; break; }
    if (!def) { snprintf(last_error, sizeof last_error, "C symbol name \"%s\" not in DLL", name); return 2; }
    if (def->numArgs != nargs) {
        snprintf(last_error, sizeof last_error, "Incorrect number of arguments (%d), expecting %d for '%s'", nargs, def->numArgs, name);
        return 3;
    }
    jmp_buf jb;
    volatile int rc = 0;
    call_jmp = &jb;
    call_pbase = pdepth;
    call_args = args;
    call_nargs = nargs;
    n_cleanups = 0;
    last_error[0] = 0;
    SEXP a[11] = {0};
    for (int k = 0; k < nargs && k < 11; k++) a[k] = args[k];
    SEXP r = NULL;
    if (setjmp(jb) == 0) {
        DL_FUNC f = def->fun;
        switch (nargs) {
        case 0: r = ((fn0)f)(); break;
        case 1: r = ((fn1)f)(a[0]); break;
        case 2: r = ((fn2)f)(a[0], a[1]); break;
        case 3: r = ((fn3)f)(a[0], a[1], a[2]); break;
        case 4: r = ((fn4)f)(a[0], a[1], a[2], a[3]); break;
        case 5: r = ((fn5)f)(a[0], a[1], a[2], a[3], a[4]); break;
        case 6: r = ((fn6)f)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
        case 7: r = ((fn7)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
        case 8: r = ((fn8)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
        case 9: r = ((fn9)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
        case 10: r = ((fn10)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]); break;
        case 11: r = ((fn11)f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10]); break;
        default: snprintf(last_error, sizeof last_error, "rmock: %d arguments not supported", nargs); rc = 3;
        }
        if (rc == 0) {
            if (pdepth != call_pbase) {
                violate("stack imbalance in '.Call' of %s, %d then %d", name, call_pbase, pdepth);     /* R warns with these words */
                pdepth = call_pbase;
            }
            if (!r) violate("%s returned a NULL pointer instead of a SEXP", name);
            else if (alive(r, "the .Call result")) { if (r != R_NilValue) r->held = 1; *result = r; }
        }
    } else {
        rc = 1;
    }
    call_jmp = NULL;
    call_args = NULL;
    call_nargs = 0;
    for (int k = 0; k < n_transient; k++) free(transient[k]);      /* R_alloc memory lives until the .Call returns */
    n_transient = 0;
    return rc;
}

/* Self-test of the collector (tests/test_r_shim_exec.py): a routine that forgets to PROTECT its first allocation must be
 * reported under gctorture; with the PROTECT in place it must not.  Returns the number of violations recorded. */
static SEXP selftest_body(SEXP with_protect)
{
    const int prot = Rf_asLogical(with_protect);
    SEXP a = Rf_allocVector(INTSXP, 4);
    if (prot) PROTECT(a);
    SEXP b = PROTECT(Rf_allocVector(REALSXP, 4));       /* under gctorture this allocation collects an unprotected `a` */
    INTEGER(a)[0] = 7;
    REAL(b)[0] = (double)INTEGER(a)[0];
    UNPROTECT(prot ? 2 : 1);
    return b;
}
long rmock_selftest_missing_protect(int with_protect)
{
    boot();
    static const R_CallMethodDef t[] = {{"selftest", (DL_FUNC)&selftest_body, 1}, {NULL, NULL, 0}};
    DllInfo saved = the_dll;
    R_registerRoutines(&the_dll, NULL, t, NULL, NULL);
    const int g = gctorture;
    gctorture = 1;
    const long v0 = n_violations;
    SEXP flag = rmock_alloc(LGLSXP, 1);
    ((int *)flag->data)[0] = with_protect;
    SEXP args[1] = {flag}, out = NULL;
    rmock_dotcall("selftest", 1, args, &out);
    rmock_release(flag);
    if (out) rmock_release(out);
    gctorture = g;
    the_dll = saved;
    return n_violations - v0;
}
