/* Functional mock of the R C API used by r/ccgp_shim.c (see Rinternals.h in this directory).  TEST
 * INFRASTRUCTURE: linked with the shim into tests/r_mock/ccgpR_mock.so and driven from Python through ctypes
 * (tests/r_mock/rmock.py).  It implements R's semantics where the shim depends on them and records what a
 * maintainer would otherwise only find out under R: unbalanced PROTECT stacks, objects left unprotected across
 * an allocation, typed accessors applied to the wrong vector type, .Call argument counts that differ from the
 * registration. */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#include <limits.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct SEXPREC {
  int type;
  R_xlen_t length;
  void* data;        /* double / int / SEXP* / char* */
  SEXP dim, names;   /* the two attributes the shim touches */
  int protect;       /* times on the PROTECT stack */
  int owned;         /* stored into a container (SET_VECTOR_ELT / SET_STRING_ELT / setAttrib) */
  int input;         /* created by the test harness (an argument of the .Call: R protects those) */
  struct SEXPREC* next;
};

static struct SEXPREC nil_rec = {NILSXP, 0, NULL, NULL, NULL, 0, 1, 1, NULL};
static struct SEXPREC names_sym = {SYMSXP, 0, NULL, NULL, NULL, 0, 1, 1, NULL};
static struct SEXPREC dim_sym = {SYMSXP, 0, NULL, NULL, NULL, 0, 1, 1, NULL};
SEXP R_NilValue = &nil_rec;
SEXP R_NamesSymbol = &names_sym;
SEXP R_DimSymbol = &dim_sym;
double R_NaReal;   /* set in rmock_init: NaN with low word 1954, as R does */
int R_NaInt = INT_MIN;

static SEXP all_objects = NULL;
static int harness_mode = 1;        /* allocations made by the harness are inputs */
static SEXP protect_stack[4096];
static int protect_top = 0;
static int unbalanced = 0, gc_hazards = 0, type_errors = 0, underflows = 0;
static char warnings[16][512];
static int n_warnings = 0;
static char last_error[512];
static jmp_buf* error_jmp = NULL;

static void init_na(void) {
  union { double d; uint32_t w[2]; } u;
  u.w[1] = 0x7ff00000u;   /* little endian: high word */
  u.w[0] = 1954u;
  R_NaReal = u.d;
}
__attribute__((constructor)) static void rmock_ctor(void) { init_na(); }

int rmock_is_na_real(double x) {
  union { double d; uint32_t w[2]; } u;
  u.d = x;
  return x != x && u.w[0] == 1954u;
}

/* rchk-style: an allocation may trigger a collection; every object of THIS call must then be on the PROTECT stack
 * or owned by something that is */
static void check_hazards(void) {
  for (SEXP o = all_objects; o; o = o->next)
    if (!o->input && o->protect == 0 && !o->owned) ++gc_hazards;
}

static SEXP new_obj(int type, R_xlen_t n) {
  if (!harness_mode) check_hazards();
  SEXP o = (SEXP)calloc(1, sizeof(struct SEXPREC));
  o->type = type;
  o->length = n;
  o->dim = o->names = R_NilValue;
  o->input = harness_mode;
  size_t el = type == REALSXP ? sizeof(double) : (type == INTSXP || type == LGLSXP) ? sizeof(int)
              : (type == VECSXP || type == STRSXP) ? sizeof(SEXP) : 1;
  o->data = calloc((size_t)(n > 0 ? n : 1) + (type == CHARSXP ? 1 : 0), el);
  if (type == VECSXP || type == STRSXP)
    for (R_xlen_t i = 0; i < n; ++i) ((SEXP*)o->data)[i] = R_NilValue;
  o->next = all_objects;
  all_objects = o;
  return o;
}

SEXP Rf_allocVector(unsigned int type, R_xlen_t n) { return new_obj((int)type, n); }

/* R_alloc: transient storage that R reclaims when the .Call returns (no PROTECT needed, never a GC hazard); here an owned
 * byte object freed with everything else at rmock_reset */
char* R_alloc(size_t n, int size) {
  SEXP o = new_obj(CHARSXP, (R_xlen_t)(n * (size_t)size) + 16);
  o->owned = 1;
  return (char*)o->data;
}

SEXP Rf_allocMatrix(unsigned int type, int nrow, int ncol) {
  SEXP o = new_obj((int)type, (R_xlen_t)nrow * ncol);
  /* R protects the matrix while it allocates the dim vector; mirror that so allocMatrix itself is no hazard */
  o->protect++;
  SEXP d = new_obj(INTSXP, 2);
  o->protect--;
  ((int*)d->data)[0] = nrow;
  ((int*)d->data)[1] = ncol;
  d->owned = 1;
  o->dim = d;
  return o;
}

static void* typed(SEXP x, int type, const char* who) {
  if (!x || x->type != type) {
    ++type_errors;
    snprintf(last_error, sizeof last_error, "%s() applied to an object of type %d", who, x ? x->type : -1);
    if (error_jmp) longjmp(*error_jmp, 2);
    abort();
  }
  return x->data;
}
double* REAL(SEXP x) { return (double*)typed(x, REALSXP, "REAL"); }
int* INTEGER(SEXP x) { return (int*)typed(x, INTSXP, "INTEGER"); }
int* LOGICAL(SEXP x) { return (int*)typed(x, LGLSXP, "LOGICAL"); }
int TYPEOF(SEXP x) { return x->type; }
R_xlen_t Rf_xlength(SEXP x) { return x->length; }
int Rf_length(SEXP x) { return (int)x->length; }
int Rf_isNull(SEXP x) { return x == R_NilValue || x->type == NILSXP; }

/* nrows / ncols as in R: from `dim` when there is one, else length and 1 */
int Rf_nrows(SEXP x) { return x->dim != R_NilValue ? ((int*)x->dim->data)[0] : (int)x->length; }
int Rf_ncols(SEXP x) { return x->dim != R_NilValue && x->dim->length >= 2 ? ((int*)x->dim->data)[1] : 1; }

int Rf_asInteger(SEXP x) {
  if (x->length < 1) return NA_INTEGER;
  if (x->type == INTSXP || x->type == LGLSXP) return ((int*)x->data)[0];
  if (x->type == REALSXP) {
    const double v = ((double*)x->data)[0];
    if (v != v || v >= 2147483648.0 || v <= -2147483649.0) return NA_INTEGER;
    return (int)v;
  }
  return NA_INTEGER;
}
double Rf_asReal(SEXP x) {
  if (x->length < 1) return NA_REAL;
  if (x->type == REALSXP) return ((double*)x->data)[0];
  if (x->type == INTSXP || x->type == LGLSXP) {
    const int v = ((int*)x->data)[0];
    return v == NA_INTEGER ? NA_REAL : (double)v;
  }
  return NA_REAL;
}

SEXP Rf_ScalarReal(double v) { SEXP o = new_obj(REALSXP, 1); ((double*)o->data)[0] = v; return o; }
SEXP Rf_ScalarInteger(int v) { SEXP o = new_obj(INTSXP, 1); ((int*)o->data)[0] = v; return o; }
SEXP Rf_ScalarLogical(int v) { SEXP o = new_obj(LGLSXP, 1); ((int*)o->data)[0] = v; return o; }
SEXP Rf_mkChar(const char* s) {
  SEXP o = new_obj(CHARSXP, (R_xlen_t)strlen(s));
  memcpy(o->data, s, strlen(s) + 1);
  return o;
}
const char* R_CHAR(SEXP x) { return (const char*)typed(x, CHARSXP, "CHAR"); }

SEXP Rf_getAttrib(SEXP x, SEXP name) {
  if (name == R_DimSymbol) return x->dim;
  if (name == R_NamesSymbol) return x->names;
  return R_NilValue;
}
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val) {
  if (name == R_DimSymbol) x->dim = val;
  else if (name == R_NamesSymbol) x->names = val;
  val->owned = 1;
  return val;
}
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) {
  SEXP* p = (SEXP*)typed(x, VECSXP, "SET_VECTOR_ELT");
  if (i < 0 || i >= x->length) { ++type_errors; return v; }
  p[i] = v;
  v->owned = 1;
  return v;
}
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { return ((SEXP*)typed(x, VECSXP, "VECTOR_ELT"))[i]; }
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) {
  SEXP* p = (SEXP*)typed(x, STRSXP, "SET_STRING_ELT");
  if (i < 0 || i >= x->length || v->type != CHARSXP) { ++type_errors; return; }
  p[i] = v;
  v->owned = 1;
}
SEXP STRING_ELT(SEXP x, R_xlen_t i) { return ((SEXP*)typed(x, STRSXP, "STRING_ELT"))[i]; }

SEXP Rf_protect(SEXP x) {
  if (protect_top < 4096) protect_stack[protect_top++] = x;
  x->protect++;
  return x;
}
void Rf_unprotect(int n) {
  while (n-- > 0) {
    if (protect_top == 0) { ++underflows; return; }
    protect_stack[--protect_top]->protect--;
  }
}

void Rf_warning(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  if (n_warnings < 16) vsnprintf(warnings[n_warnings], sizeof warnings[0], fmt, ap);
  ++n_warnings;
  va_end(ap);
}
void Rf_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error, sizeof last_error, fmt, ap);
  va_end(ap);
  if (error_jmp) longjmp(*error_jmp, 1);
  fprintf(stderr, "Rf_error outside a .Call: %s\n", last_error);
  abort();
}

/* ---- registration and the .Call trampoline ------------------------------------------------------------- */
static const R_CallMethodDef* call_table = NULL;
static int dynamic_symbols = 1;
int R_registerRoutines(DllInfo* info, const void* c, const R_CallMethodDef* call, const void* f, const void* e) {
  (void)info; (void)c; (void)f; (void)e;
  call_table = call;
  return 1;
}
int R_useDynamicSymbols(DllInfo* info, int value) { (void)info; dynamic_symbols = value; return 1; }

extern void R_init_ccgpR(DllInfo* dll);
extern void R_unload_ccgpR(DllInfo* dll);

int rmock_load(void) {   /* what dyn.load() does after dlopen */
  R_init_ccgpR(NULL);
  if (!call_table) return -1;
  int n = 0;
  while (call_table[n].name) ++n;
  return n;
}
void rmock_unload(void) { R_unload_ccgpR(NULL); }
const char* rmock_routine_name(int i) { return call_table[i].name; }
int rmock_routine_nargs(int i) { return call_table[i].numArgs; }
int rmock_dynamic_symbols(void) { return dynamic_symbols; }

typedef SEXP (*F0)(void);
typedef SEXP (*F1)(SEXP);
typedef SEXP (*F2)(SEXP, SEXP);
typedef SEXP (*F3)(SEXP, SEXP, SEXP);
typedef SEXP (*F4)(SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*F5)(SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*F6)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*F7)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*F8)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);

/* .Call(name, args...): looked up in the REGISTERED table (R_useDynamicSymbols(FALSE) forbids anything else), the
 * argument count must match the registration.  Returns NULL after Rf_error (message in rmock_last_error) or when
 * the routine is not registered / the count differs. */
SEXP rmock_dot_call(const char* name, int nargs, SEXP* a) {
  last_error[0] = 0;
  if (!call_table) { snprintf(last_error, sizeof last_error, "no routines registered"); return NULL; }
  const R_CallMethodDef* m = NULL;
  for (int i = 0; call_table[i].name; ++i)
    if (strcmp(call_table[i].name, name) == 0) m = &call_table[i];
  if (!m) { snprintf(last_error, sizeof last_error, "\"%s\" not available for .Call()", name); return NULL; }
  if (m->numArgs != nargs) {
    snprintf(last_error, sizeof last_error, "Incorrect number of arguments (%d), expecting %d for '%s'", nargs, m->numArgs, name);
    return NULL;
  }
  const int top0 = protect_top;
  jmp_buf jb;
  error_jmp = &jb;
  harness_mode = 0;
  SEXP r = NULL;
  if (setjmp(jb) == 0) {
    switch (nargs) {
      case 0: r = ((F0)m->fun)(); break;
      case 1: r = ((F1)m->fun)(a[0]); break;
      case 2: r = ((F2)m->fun)(a[0], a[1]); break;
      case 3: r = ((F3)m->fun)(a[0], a[1], a[2]); break;
      case 4: r = ((F4)m->fun)(a[0], a[1], a[2], a[3]); break;
      case 5: r = ((F5)m->fun)(a[0], a[1], a[2], a[3], a[4]); break;
      case 6: r = ((F6)m->fun)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
      case 7: r = ((F7)m->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
      case 8: r = ((F8)m->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
      default: snprintf(last_error, sizeof last_error, "mock trampoline takes at most 8 arguments"); break;
    }
    if (protect_top != top0) ++unbalanced;   /* R: "stack imbalance in .Call" */
  } else {
    /* Rf_error: R unwinds the protect stack to the context's level */
    while (protect_top > top0) protect_stack[--protect_top]->protect--;
    r = NULL;
  }
  harness_mode = 1;
  error_jmp = NULL;
  /* the value of the call now belongs to the caller (in R: bound to a variable, hence reachable); what the call
   * left behind unreferenced is garbage, not a hazard for LATER calls */
  for (SEXP o = all_objects; o; o = o->next) o->input = 1;
  return r;
}

/* ---- harness helpers (Python side: tests/r_mock/rmock.py) ------------------------------------------------- */
SEXP rmock_real(const double* v, R_xlen_t n, int nrow, int ncol) {   /* nrow < 0: plain vector */
  SEXP o = nrow >= 0 ? Rf_allocMatrix(REALSXP, nrow, ncol) : Rf_allocVector(REALSXP, n);
  if (n) memcpy(o->data, v, sizeof(double) * (size_t)n);
  return o;
}
SEXP rmock_int(const int* v, R_xlen_t n) {
  SEXP o = Rf_allocVector(INTSXP, n);
  if (n) memcpy(o->data, v, sizeof(int) * (size_t)n);
  return o;
}
SEXP rmock_nil(void) { return R_NilValue; }
/* a data frame as .Call sees it: a VECSXP of columns with a `names` attribute (n_names = 0: unnamed list) */
SEXP rmock_list(R_xlen_t n) { return Rf_allocVector(VECSXP, n); }
void rmock_list_set(SEXP l, R_xlen_t i, SEXP v) { SET_VECTOR_ELT(l, i, v); }
void rmock_list_names(SEXP l, const char** names, int n_names) {
  SEXP s = Rf_allocVector(STRSXP, n_names);
  for (int i = 0; i < n_names; ++i) SET_STRING_ELT(s, i, Rf_mkChar(names[i]));
  Rf_setAttrib(l, R_NamesSymbol, s);
}
int rmock_typeof(SEXP x) { return x->type; }
long rmock_length(SEXP x) { return (long)x->length; }
int rmock_dim(SEXP x, int which) { return x->dim == R_NilValue ? -1 : ((int*)x->dim->data)[which]; }
void* rmock_data(SEXP x) { return x->data; }
SEXP rmock_elt(SEXP x, long i) { return ((SEXP*)x->data)[i]; }
SEXP rmock_names(SEXP x) { return x->names; }
const char* rmock_char(SEXP x) { return (const char*)x->data; }
double rmock_na_real(void) { return R_NaReal; }
int rmock_n_warnings(void) { return n_warnings; }
const char* rmock_warning(int i) { return i < 16 && i < n_warnings ? warnings[i] : ""; }
const char* rmock_last_error(void) { return last_error; }
int rmock_counter(int which) {   /* 0 unbalanced PROTECT, 1 gc hazards, 2 type errors, 3 UNPROTECT underflows, 4 protect depth */
  switch (which) {
    case 0: return unbalanced;
    case 1: return gc_hazards;
    case 2: return type_errors;
    case 3: return underflows;
    default: return protect_top;
  }
}
/* free every object and clear the records (between tests; the shim's handles survive) */
void rmock_reset(void) {
  while (all_objects) {
    SEXP n = all_objects->next;
    free(all_objects->data);
    free(all_objects);
    all_objects = n;
  }
  protect_top = 0;
  unbalanced = gc_hazards = type_errors = underflows = 0;
  n_warnings = 0;
  last_error[0] = 0;
}
