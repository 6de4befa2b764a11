/* see ../Rinternals.h: syntax-check declarations only (R_ext/Rdynload.h of R 4.x) */
#ifndef MX_TEST_RDYNLOAD_H
#define MX_TEST_RDYNLOAD_H
#include "../Rinternals.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
struct R_CMethodDef_;
int R_registerRoutines(DllInfo *info, const void *cMethods, const R_CallMethodDef *callMethods, const void *fortranMethods,
                       const void *externalMethods);
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value);
struct Rf_RegisteredNativeSymbol;
DL_FUNC R_FindSymbol(char const *name, char const *pkg, struct Rf_RegisteredNativeSymbol *symbol);
#ifdef __cplusplus
}
#endif
#endif
