/* FUNCTIONAL MOCK of R's <R.h> -- see Rinternals.h in this directory. */
#ifndef CCGP_MOCK_R_H
#define CCGP_MOCK_R_H
#include <stddef.h>
typedef enum { FALSE = 0, TRUE } Rboolean;
void Rf_error(const char* fmt, ...) __attribute__((noreturn));
void Rf_warning(const char* fmt, ...);
char* R_alloc(size_t n, int size);
#endif
