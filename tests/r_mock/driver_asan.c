/* driver_asan.c — drives the .Call shim + the mock R runtime from plain C, for an AddressSanitizer + UBSan build that runs
 * WITHOUT a GPU (tests/test_r_shim_exec.py::test_shim_and_mock_under_asan): every routine of the registered table is
 * called by name with well-formed small arguments and with coercions forced (indices / nthreads as doubles), with the
 * collector running on every allocation and with R allocation failures injected at every position.  On a box without a
 * GPU the library refuses each call, i.e. what runs is the shim's marshalling up to the C-ABI call and its error exit
 * (Rf_error long-jump with the protect stack unwound) — the part no GPU test can put under a sanitizer (GPU ASan is not
 * available on the pool).  TEST INFRASTRUCTURE ONLY. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

/* rmock.c's driver interface */
DllInfo *rmock_dllinfo(void);
int rmock_n_routines(void);
const char *rmock_routine_name(int k);
int rmock_routine_arity(int k);
SEXP rmock_alloc(int type, int64_t n);
void *rmock_dataptr(SEXP x);
void rmock_set_dim(SEXP x, int nr, int nc);
void rmock_set_class(SEXP x, const char *cls);
void rmock_set_slot(SEXP obj, const char *name, SEXP v);
void rmock_set_list_elt(SEXP x, int64_t k, SEXP v);
void rmock_release(SEXP x);
void rmock_set_gctorture(int on);
void rmock_fail_alloc_at(long k);
int rmock_protect_depth(void);
int rmock_preserved_count(void);
long rmock_violations(void);
const char *rmock_violation_msg(void);
const char *rmock_last_error(void);
int rmock_dotcall(const char *name, int nargs, SEXP *args, SEXP *result);
long rmock_selftest_missing_protect(int with_protect);
void rmock_clear_violations(void);
void rmock_sweep_dead(void);
void R_init_mxgpu_r(DllInfo *);

static SEXP ints(int n, const int *v) { SEXP s = rmock_alloc(INTSXP, n); if (n) memcpy(rmock_dataptr(s), v, sizeof(int) * (size_t)n); return s; }
static SEXP lgls(int n, const int *v) { SEXP s = rmock_alloc(LGLSXP, n); if (n) memcpy(rmock_dataptr(s), v, sizeof(int) * (size_t)n); return s; }
static SEXP reals(int n, const double *v) { SEXP s = rmock_alloc(REALSXP, n); if (n) memcpy(rmock_dataptr(s), v, sizeof(double) * (size_t)n); return s; }

int main(void)
{
    R_init_mxgpu_r(rmock_dllinfo());
    if (rmock_n_routines() < 60) { fprintf(stderr, "only %d routines registered\n", rmock_n_routines()); return 1; }
    if (rmock_selftest_missing_protect(1) != 0 || rmock_selftest_missing_protect(0) == 0) { fprintf(stderr, "collector self-test failed\n"); return 1; }
    rmock_clear_violations();
    rmock_set_gctorture(1);

    /* a 3 x 4 CSR matrix (the README's), its twin with double-typed index vectors, dense operands */
    const int p[] = {0, 2, 3, 3}, j[] = {2, 3, 1}, l[] = {1, 0, INT32_MIN};
    const double x[] = {2, 1, 3}, pd[] = {0, 2, 3, 3}, jd[] = {2, 3, 1};
    double dense[12];
    for (int k = 0; k < 12; k++) dense[k] = k * 0.5 - 1;
    SEXP sp = ints(4, p), sj = ints(3, j), sx = reals(3, x), sl = lgls(3, l), spd = reals(4, pd), sjd = reals(3, jd);
    SEXP Y = reals(12, dense);
    rmock_set_dim(Y, 3, 4);                                     /* n x K for tcrossprod_csr_dense; also X (nr x nc) */
    int yi[12];
    for (int k = 0; k < 12; k++) yi[k] = k;
    SEXP Yi = ints(12, yi);
    rmock_set_dim(Yi, 3, 4);
    const int one = 1, four = 4, rows[] = {1, 0, 1}, cols[] = {1, 2};
    const double onef = 1.0;
    SEXP s1 = ints(1, &one), s4 = ints(1, &four), s1d = reals(1, &onef), T = lgls(1, &one), F = lgls(1, &(int){0});
    SEXP srows = ints(3, rows), scols = ints(2, cols), v4 = reals(4, dense), vi4 = ints(4, yi), d3 = reals(3, dense);
    SEXP out = rmock_alloc(25, 0);                               /* S4 */
    rmock_set_class(out, "dgRMatrix");
    rmock_set_slot(out, "p", ints(4, p)); rmock_set_slot(out, "j", ints(3, j)); rmock_set_slot(out, "x", reals(3, x));
    SEXP A = rmock_alloc(25, 0);
    rmock_set_class(A, "dgRMatrix");
    const int dim[] = {3, 4};
    rmock_set_slot(A, "p", sp); rmock_set_slot(A, "j", sj); rmock_set_slot(A, "x", sx); rmock_set_slot(A, "Dim", ints(2, dim));
    SEXP objs = rmock_alloc(VECSXP, 1);
    rmock_set_list_elt(objs, 0, A);

    /* arguments by arity: the first ones are the CSR triple wherever a routine takes one; whatever the routine makes of the
     * rest, it must neither crash nor leave the protect stack unbalanced — on a box without a GPU every call ends in an R error */
    long calls = 0, errors = 0;
    for (int k = 0; k < rmock_n_routines(); k++) {
        const char *name = rmock_routine_name(k);
        const int ar = rmock_routine_arity(k);
        for (int variant = 0; variant < 3; variant++) {
            SEXP a[11];
            const int dbl = variant == 1;                        /* variant 1: index vectors / scalars as doubles (coercion paths) */
            SEXP P = dbl ? spd : sp, J = dbl ? sjd : sj, ONE = dbl ? s1d : s1;
            for (int q = 0; q < 11; q++) a[q] = ONE;
            if (strstr(name, "dense_csc") || strstr(name, "tcrossprod_dense_csr")) { a[0] = strstr(name, "float32") ? Yi : Y; a[1] = P; a[2] = J; a[3] = sx; a[4] = ONE; a[5] = s4; }
            else if (strstr(name, "tcrossprod_csr_dense")) { a[0] = P; a[1] = J; a[2] = sx; a[3] = strstr(name, "float32") ? Yi : Y; a[4] = ONE; }
            else if (strstr(name, "concat_csr_batch")) { a[0] = objs; a[1] = out; }
            else if (strstr(name, "check_is_")) { a[0] = dbl ? sjd : sj; }
            else if (strstr(name, "rowvec_by_csc")) { a[0] = vi4; a[1] = P; a[2] = J; a[3] = sx; }
            else if (strstr(name, "_mxgpu_set_option")) { a[0] = sj; a[1] = ONE; }          /* a non-string name: refused */
            else if (strstr(name, "_mxgpu_set_devices")) { a[0] = variant == 2 ? s4 : rmock_alloc(INTSXP, 0); }
            else if (strstr(name, "elemwise") && !strstr(name, "dense")) { a[0] = P; a[1] = P; a[2] = J; a[3] = J; a[4] = strstr(name, "logical") ? sl : sx; a[5] = a[4]; a[6] = variant == 2 ? T : F; }
            else if (strstr(name, "cbind")) { a[0] = P; a[1] = J; a[2] = strstr(name, "binary") ? P : (strstr(name, "logical") ? sl : sx); a[3] = strstr(name, "binary") ? J : P; a[4] = J; a[5] = strstr(name, "logical") ? sl : sx; }
            else {                                               /* CSR triple first, then vectors / scalars */
                a[0] = P; a[1] = J;
                const int binary = strstr(name, "binary") != NULL && !strstr(name, "svec") && !strstr(name, "reverse_columns");
                int q = 2;
                if (!binary) a[q++] = strstr(name, "logical") && !strstr(name, "svec") && !strstr(name, "dvec_logical") ? sl : sx;
                if (strstr(name, "copy_csr") || strstr(name, "svec")) { a[q++] = srows; if (q < 11) a[q++] = strstr(name, "svec") ? d3 : scols; }
                else if (strstr(name, "dvec") && !strstr(name, "by_dvec")) a[q++] = strstr(name, "integer") || strstr(name, "logical") || strstr(name, "float32") ? vi4 : v4;
                else if (strstr(name, "by_dvec")) { a[q++] = strstr(name, "logicaland") ? lgls(3, l) : d3; a[q++] = s4; a[q++] = T; for (; q < 11; q++) a[q] = F; a[10] = T; }
                else if (strstr(name, "by_dense")) a[q++] = strstr(name, "double") ? Y : Yi;
                else if (strstr(name, "check_valid")) { a[2] = s4; a[3] = s4; }
            }
            for (long fail = -1; fail < (variant == 0 ? 6 : 0); fail++) {
                SEXP res = NULL;
                rmock_fail_alloc_at(fail);
                const int depth = rmock_protect_depth(), kept = rmock_preserved_count();
                const int rc = rmock_dotcall(name, ar, a, &res);
                rmock_fail_alloc_at(-1);
                calls++;
                errors += rc == 1;
                if (rc > 1) { fprintf(stderr, "%s: rc %d (%s)\n", name, rc, rmock_last_error()); return 1; }
                if (rmock_protect_depth() != depth || rmock_preserved_count() != kept) { fprintf(stderr, "%s: protect / preserve imbalance\n", name); return 1; }
                if (rmock_violations()) { fprintf(stderr, "%s: %s\n", name, rmock_violation_msg()); return 1; }
                if (res) rmock_release(res);
            }
        }
        rmock_sweep_dead();
    }
    printf("rmock driver ok: %ld calls through %d routines, %ld R errors, no violations\n", calls, rmock_n_routines(), errors);
    return 0;
}
