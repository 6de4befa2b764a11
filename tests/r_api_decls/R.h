/* see Rinternals.h in this directory: syntax-check declarations only */
#ifndef MX_TEST_R_H
#define MX_TEST_R_H
#include <stddef.h>
#endif
