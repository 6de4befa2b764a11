/* Hand-written declarations of the part of R's C API that matrixextra_amd/csrc/r_shim.cpp uses — for
 * `g++ -fsyntax-only` in tests/test_r_shim_syntax.py ONLY (R is not installed in the development image; nothing links
 * against this).  Signatures follow "Writing R Extensions" §5-6 / R 4.x Rinternals.h: types and signedness matter to the
 * check (R_xlen_t is ptrdiff_t, LOGICAL() is int*, DL_FUNC is a generic function pointer). */
#ifndef MX_TEST_RINTERNALS_H
#define MX_TEST_RINTERNALS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
typedef enum { FALSE = 0, TRUE } Rboolean;
#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
extern SEXP R_NilValue, R_NamesSymbol;
int TYPEOF(SEXP x);
R_xlen_t XLENGTH(SEXP x);
R_xlen_t Rf_xlength(SEXP x);
int *INTEGER(SEXP x);
int *LOGICAL(SEXP x);
double *REAL(SEXP x);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
const char *R_CHAR(SEXP x);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
SEXP Rf_coerceVector(SEXP, SEXPTYPE);
SEXP Rf_mkChar(const char *);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
SEXP Rf_install(const char *);
SEXP Rf_ScalarLogical(int);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
Rboolean Rf_inherits(SEXP, const char *);
SEXP R_do_slot(SEXP obj, SEXP name);
int R_has_slot(SEXP obj, SEXP name);
void R_PreserveObject(SEXP);
void R_ReleaseObject(SEXP);
SEXP R_ExecWithCleanup(SEXP (*fun)(void *), void *data, void (*cleanfun)(void *), void *cleandata);
char *R_alloc(size_t, int);
#if defined(__GNUC__)
void Rf_error(const char *, ...) __attribute__((noreturn, format(printf, 1, 2)));
#else
void Rf_error(const char *, ...);
#endif
#ifdef __cplusplus
}
#endif
#endif
