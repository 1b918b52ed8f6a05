/*
 * rectdetect-mi355x: the reference's OpenCL dispatch helper surface (reference oclhelper.h:12-39),
 * served by a thin HIP runtime layer (rectdetect_amd/csrc/rd_runtime.hip) instead of an OpenCL ICD.
 *
 * Opaque-handle mapping:  cl_device_id -> HIP device ordinal, cl_context -> device binding,
 * cl_command_queue -> hipStream_t, cl_mem -> hipMalloc / hipHostMalloc allocation, cl_event -> hipEvent_t.
 * Kernels are compiled ahead of time for gfx950, so there is no program build at run time and work-group
 * sizes are fixed per kernel: the "plan" (local-work-size autotuner, reference oclhelper.c:312-605) functions
 * keep their signatures and file format but do not influence launches.
 *
 * Include <CL/cl.h> (types only) before this header, exactly as the reference's users do.
 */
#ifndef RD_COMPAT_OCLHELPER_H
#define RD_COMPAT_OCLHELPER_H
#if defined(__cplusplus)
extern "C" {
#endif

const char *clStrError(int c);                         /* reference oclhelper.c:107-111 */
cl_int checkError(cl_int ret, const char *s);          /* fatal unless CL_SUCCESS, :113-119 */
cl_int ce(cl_int ret);                                 /* fatal unless CL_SUCCESS, :133-138 */

char *getDeviceName(cl_device_id device);              /* malloc'd "<name>, <version>", :143-158 */
cl_device_id simpleGetDevice(int did);                 /* flat index over all devices; bad index lists devices and exit(-1), :171-196 */
int simpleGetDevices(cl_device_id *devices, int maxDevices);
cl_context simpleCreateContext(cl_device_id device);   /* :225-233 */
int simpleBuildProgram(cl_program program, cl_device_id device, const char *optionString);
void simpleSetKernelArg(cl_kernel kernel, const char *format, ...);   /* 'i' int, 'l' long, 'f' float, 'd' double, 'M' cl_mem; :254-308 */
cl_event runKernel1D(cl_command_queue queue, cl_kernel kernel, int kernelID, size_t ws1, int nev, ...);
cl_event runKernel2D(cl_command_queue queue, cl_kernel kernel, int kernelID, size_t ws1, size_t ws2, int nev, ...);
cl_event runKernel1Dx(cl_command_queue queue, cl_kernel kernel, int kernelID, size_t ws1, const cl_event *events);
cl_event runKernel2Dx(cl_command_queue queue, cl_kernel kernel, int kernelID, size_t ws1, size_t ws2, const cl_event *events);

void waitForEvent(cl_event ev);                        /* blocks until the event completed, :799-817 */

void clearPlan();
int loadPlan(const char *fn, cl_device_id device);     /* 0 if lines for this device were found, else -1, :394-441 */
void savePlan(const char *fn, cl_device_id device);    /* :464-527 */
void startProfiling(size_t ws1, size_t ws2, size_t ws3);
void finishProfiling();
void showPlan();

void *allocatePinnedMemory(size_t z, cl_context context, cl_command_queue queue);   /* page-locked host memory, :837-853 */
void freePinnedMemory(void *p, cl_context context, cl_command_queue queue);

int getNextKernelID();

#if defined(__cplusplus)
}
#endif
#endif
