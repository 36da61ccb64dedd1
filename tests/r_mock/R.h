/* MOCK of R's <R.h> for tests/test_r_shim_compiles.py ONLY: just enough declarations for r/ccgp_shim.c to go
 * through the C front end, so that a drift between the shim's calls and include/ccgp.h is a compile error in
 * the CPU test-suite.  It says nothing about R's real ABI and is never linked or executed. */
#ifndef CCGP_MOCK_R_H
#define CCGP_MOCK_R_H
#include <stddef.h>
typedef enum { FALSE = 0, TRUE } Rboolean;
void Rf_error(const char* fmt, ...);
void Rf_warning(const char* fmt, ...);
#endif
