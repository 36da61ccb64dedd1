/* MOCK of <Rinternals.h> -- see R.h in this directory. */
#ifndef CCGP_MOCK_RINTERNALS_H
#define CCGP_MOCK_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
enum { LGLSXP = 10, INTSXP = 13, REALSXP = 14, STRSXP = 16, VECSXP = 19 };
extern double R_NaReal;
extern int R_NaInt;
extern SEXP R_NamesSymbol;
#define NA_REAL R_NaReal
#define NA_LOGICAL R_NaInt
#define ISNAN(x) ((x) != (x))
double* REAL(SEXP x);
int* INTEGER(SEXP x);
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
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
#endif
