/* Declarations-only stand-in for MATLAB's mex.h -- TEST INFRASTRUCTURE, not a MEX runtime.
 *
 * The image has no MATLAB, so mex/dbat_hip_mex.cpp can never be linked here.  This header carries
 * just the prototypes the gateway uses (interleaved-complex API of R2018a, as the reference's own
 * MEX file uses them: /root/reference/code/test/postcov/icpc_mex.c:495-611, dumpsparse.c:25-27), so
 * that tests/test_mex_gateway.py can run the compiler's front end over the gateway
 * (g++ -fsyntax-only) and catch signature / type errors.  Nothing is defined, nothing links. */
#ifndef DBAT_TEST_MEX_STUB_H
#define DBAT_TEST_MEX_STUB_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxUNKNOWN_CLASS = 0, mxDOUBLE_CLASS = 6, mxINT32_CLASS = 12, mxUINT8_CLASS = 9 } mxClassID;

bool mxIsStruct(const mxArray *pa);
mxArray *mxGetField(const mxArray *pa, mwIndex index, const char *fieldname);
double mxGetScalar(const mxArray *pa);
void *mxGetData(const mxArray *pa);
double *mxGetDoubles(const mxArray *pa);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double value);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID classid, mxComplexity flag);
void mxSetN(mxArray *pa, mwSize n);
mxArray *mxCreateSparse(mwSize m, mwSize n, mwSize nzmax, mxComplexity flag);
mwIndex *mxGetIr(const mxArray *pa);
mwIndex *mxGetJc(const mxArray *pa);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag);
size_t mxGetNumberOfElements(const mxArray *pa);
bool mxIsChar(const mxArray *pa);
bool mxIsUint8(const mxArray *pa);
int mxGetString(const mxArray *pa, char *str, mwSize strlen);
void mxDestroyArray(mxArray *pa);
bool mxIsEmpty(const mxArray *pa);
bool mxIsClass(const mxArray *pa, const char *classname);
int mexCallMATLAB(int nlhs, mxArray *plhs[], int nrhs, mxArray *prhs[], const char *functionName);
mxArray *mexCallMATLABWithTrap(int nlhs, mxArray *plhs[], int nrhs, mxArray *prhs[], const char *functionName);
void mexErrMsgIdAndTxt(const char *identifier, const char *fmt, ...);
int mexPrintf(const char *fmt, ...);
void mexLock(void);
void mexUnlock(void);
bool mexIsLocked(void);
int mexAtExit(void (*exit_fcn)(void));
void mexWarnMsgIdAndTxt(const char *identifier, const char *fmt, ...);

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);

#ifdef __cplusplus
}
#endif
#endif
