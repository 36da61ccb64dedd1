/* FUNCTIONAL MOCK of <Rinternals.h> for the test-suite (tests/r_mock/rmock.c implements it).
 *
 * The build container and the GPU boxes have no R.  r/ccgp_shim.c is nevertheless meant to be EXECUTED by the
 * tests, so this mock implements the dozen R-API entry points the shim uses with R's semantics where they matter
 * to the shim: typed vectors with a length, the `dim` and `names` attributes, Rf_nrows / Rf_ncols on matrices and
 * plain vectors, NA_REAL as R's NaN payload 1954, NA_INTEGER = INT_MIN, coercion in Rf_asInteger / Rf_asReal,
 * VECSXP / STRSXP containers, a PROTECT stack whose balance is checked per .Call, an rchk-style check that every
 * object is protected (or owned by a protected container) whenever a later allocation could collect it,
 * Rf_warning captured instead of printed, Rf_error as a longjmp back to the .Call trampoline, and
 * R_registerRoutines feeding that trampoline (so the registered argument counts are what dispatches a call).
 * It says nothing about R's real binary layout: the shim is compiled against THIS header for the tests and
 * against R's own headers by a maintainer. */
#ifndef CCGP_MOCK_RINTERNALS_H
#define CCGP_MOCK_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
enum { NILSXP = 0, SYMSXP = 1, CHARSXP = 9, LGLSXP = 10, INTSXP = 13, REALSXP = 14, STRSXP = 16, VECSXP = 19 };
extern double R_NaReal;
extern int R_NaInt;
extern SEXP R_NilValue;
extern SEXP R_NamesSymbol;
extern SEXP R_DimSymbol;
#define NA_REAL R_NaReal
#define NA_INTEGER R_NaInt
#define NA_LOGICAL R_NaInt
#define ISNAN(x) ((x) != (x))
double* REAL(SEXP x);
int* INTEGER(SEXP x);
int* LOGICAL(SEXP x);
int TYPEOF(SEXP x);
R_xlen_t Rf_xlength(SEXP x);
int Rf_nrows(SEXP x);
int Rf_ncols(SEXP x);
int Rf_length(SEXP x);
int Rf_isNull(SEXP x);
int Rf_asInteger(SEXP x);
double Rf_asReal(SEXP x);
SEXP Rf_allocVector(unsigned int type, R_xlen_t n);
SEXP Rf_allocMatrix(unsigned int type, int nrow, int ncol);
SEXP Rf_ScalarReal(double x);
SEXP Rf_ScalarInteger(int x);
SEXP Rf_ScalarLogical(int x);
SEXP Rf_mkChar(const char* s);
SEXP Rf_getAttrib(SEXP x, SEXP name);
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
const char* R_CHAR(SEXP x);
#define CHAR(x) R_CHAR(x)
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
#endif
