/* FUNCTIONAL MOCK of <R_ext/Rdynload.h> -- see ../Rinternals.h. */
#ifndef CCGP_MOCK_RDYNLOAD_H
#define CCGP_MOCK_RDYNLOAD_H
typedef void* (*DL_FUNC)(void);
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo* info, const void* c, const R_CallMethodDef* call, const void* f, const void* e);
int R_useDynamicSymbols(DllInfo* info, int value);
#endif
